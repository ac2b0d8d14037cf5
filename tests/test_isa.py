"""The ISA of the product kernels: no workgroup barrier that a wave can skip.  An `s_barrier` directly behind an
`s_cbranch_execz` is jumped over by a wave whose exec mask is empty there while its sibling waves execute it -- the
workgroup never meets again (a GPU hang; the first revision of round 3's k_step_fused held one).  Compiles
the library's device code to assembly with the product flags (hipcc cross-compiles without a GPU) and checks every kernel."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_no_kernel_holds_a_barrier_a_wave_can_skip(tmp_path):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = str(tmp_path / "dev.s")
    subprocess.check_call([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics",
                           "-fvisibility=hidden", "-fvisibility-inlines-hidden", "--cuda-device-only", "-S", "-o", out,
                           os.path.join(ROOT, "nl-partsol_amd", "csrc", "nlps_gpu.hip")], stderr=subprocess.DEVNULL)
    txt = open(out).read()
    kernels = re.findall(r"\.amdhsa_kernel (\S+)", txt)
    assert len(kernels) > 50
    nbar = 0
    for name in kernels:
        body = [l.strip() for l in txt[txt.find("\n" + name + ":"):txt.find(".amdhsa_kernel " + name)].split("\n")]
        ins = [l for l in body if l and not l.startswith((";", "."))]
        for k, l in enumerate(ins):
            if l.startswith("s_barrier"):
                nbar += 1
                assert k == 0 or not ins[k - 1].startswith(("s_cbranch_execz", "s_cbranch_execnz")), \
                    "%s: s_barrier behind %s" % (name, ins[k - 1])
    assert nbar > 20  # (the tile kernels do hold barriers: the check looked at something)
