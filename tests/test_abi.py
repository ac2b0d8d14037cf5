"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/nlps_gpu.h declares, and fails loudly (no CPU fallback) when there is no GPU."""
import ctypes
import os
import re

import pytest

from util import ROOT, make_case, nlps


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "nlps_gpu.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nlps_(?:gpu|host)_\w+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    n = nlps()
    L = n.lib()
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(L, s), "missing export " + s
    assert sorted(n.SYMBOLS) == syms, "python binding list out of sync with the header"


def test_header_is_plain_c():
    import subprocess
    import tempfile
    src = '#include "nlps_gpu.h"\nint main(void){ nlps_grid g; nlps_particles p; (void)g; (void)p; return 0; }\n'
    with tempfile.TemporaryDirectory() as d:
        f = os.path.join(d, "t.c")
        open(f, "w").write(src)
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                               "-c", f, "-o", os.path.join(d, "t.o")])


def test_product_does_not_reference_the_oracle():
    """The product path must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "nl-partsol_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert not re.search(r"import\s+oracle|from\s+oracle|nlps_oracle|oracle/|orc_\w+\(", txt), \
                    fn + " references the oracle"
    import subprocess
    out = subprocess.check_output(["ldd", os.path.join(pkg, "csrc", "libnlps_gpu.so")]).decode()
    assert "nlps_oracle" not in out


def test_no_gpu_means_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    n = nlps()
    case = make_case(2, [8, 8], [2, 2], [3, 3])
    with pytest.raises(n.NlpsError):
        n.Solver(2, case["grid_n"], case["origin"], case["h"], case["cloud"], case["materials"])


def test_glue_compiles_against_the_reference_headers(tmp_path):
    """integration/nlps_glue.c is the binding a maintainer adds to the reference.  It includes the
    reference's own Types.h / Globals.h, so wherever the reference tree is present (this container; not the GPU
    box) it is COMPILED TO AN OBJECT against the real Particle / Mesh / Material / Boundaries declarations, in 2-D
    and 3-D, and every nlps_* symbol the object leaves undefined must be an export of the library (the reference
    itself cannot be linked here: PETSc, LAPACK)."""
    import shutil
    import subprocess
    import pytest
    ref = "/root/reference/nl-partsol/src"
    if not os.path.isdir(ref) or shutil.which("gcc") is None:
        pytest.skip("reference tree or gcc not available here")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = nlps()
    for k, dim in enumerate((["-DUSE_PLAINSTRAIN"], [])):
        obj = str(tmp_path / ("nlps_glue_%d.o" % k))
        cmd = ["gcc", "-std=gnu99", "-c", "-Werror=implicit-function-declaration",
               "-Werror=incompatible-pointer-types", "-Werror=int-conversion"] + dim + \
              ["-I" + ref, "-I" + os.path.join(root, "include"), os.path.join(root, "integration", "nlps_glue.c"), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        syms = subprocess.run(["nm", "-u", obj], capture_output=True, text=True).stdout.split()
        used = sorted(x for x in syms if x.startswith("nlps_"))
        assert "nlps_gpu_create" in used and "nlps_host_lattice_from_nodes" in used and "nlps_gpu_update_kinetics" in used
        for name in used:
            assert name in n.SYMBOLS, name + " is used by the glue but not exported by the library"
        defined = subprocess.run(["nm", "--defined-only", obj], capture_output=True, text=True).stdout
        for name in ("nlps_glue_create", "nlps_glue_boundaries", "nlps_glue_masks", "nlps_glue_download",
                     "nlps_glue_update_particles_static", "nlps_glue_free"):
            assert name in defined


def test_c_caller_links_against_the_library(tmp_path):
    """tests/c/abi_step.c (plain C99, no Python in the way) compiles and LINKS against libnlps_gpu.so; the GPU test
    test_gpu_c_caller.py runs it."""
    import shutil
    import subprocess
    import pytest
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "nl-partsol_amd", "csrc")
    nlps().lib()
    exe = str(tmp_path / "abi_step")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(root, "include"),
                        os.path.join(root, "tests", "c", "abi_step.c"), "-o", exe, "-L" + libdir, "-lnlps_gpu",
                        "-Wl,-rpath," + libdir, "-lm"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    assert os.path.exists(os.path.join(root, "tests", "golden", "nh3d_abi.bin"))
