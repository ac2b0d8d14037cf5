"""A C caller that links: tests/c/abi_step.c is built with gcc against libnlps_gpu.so and run as a fresh child process
(no Python, no ctypes between the caller and the C-ABI); it replays the "nh3d" golden case from the committed binary
fixture and compares with the oracle's end state itself."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu


def test_c_caller_reproduces_the_golden_case(tmp_path):
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "nl-partsol_amd", "csrc")
    assert os.path.exists(os.path.join(libdir, "libnlps_gpu.so")), "libnlps_gpu.so is not built"
    exe = str(tmp_path / "abi_step")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-I" + os.path.join(root, "include"),
                        os.path.join(root, "tests", "c", "abi_step.c"), "-o", exe, "-L" + libdir, "-lnlps_gpu",
                        "-Wl,-rpath," + libdir, "-lm"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe, os.path.join(root, "tests", "golden", "nh3d_abi.bin")], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and "abi_step: PASS" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
