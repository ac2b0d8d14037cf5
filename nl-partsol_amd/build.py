"""Builds the gfx950 shared library (hipcc cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "nlps_gpu.hip")
IO_SRC = os.path.join(HERE, "csrc", "nlps_io.cpp")  # host-only input formats (GiD meshes, lattice, particles)
DEPS = [SRC, IO_SRC] + [os.path.join(HERE, "csrc", f) for f in ("nlps_device.hpp", "nlps_tables.hpp", "nlps_tile_kernels.hpp",
                                                         "nlps_tangent_kernels.hpp")] + \
       [os.path.join(HERE, "..", "include", "nlps_gpu.h")]
# NLPS_GPU_LIB: developer hook, loads another BUILD of this same library (kernel experiments: tools/kbench.py)
LIB = os.environ.get("NLPS_GPU_LIB") or os.path.join(HERE, "csrc", "libnlps_gpu.so")
# -ffp-contract=off: index-deciding arithmetic (closest node, cut-off radius) must round exactly like
# the CPU path; hot loops that may fuse use explicit fma().  -munsafe-fp-atomics: hardware
# global_atomic_add_f64 instead of a CAS loop.  -fvisibility=hidden: only the C-ABI of include/nlps_gpu.h is exported
# (the header switches the default back on for its declarations), so the template instantiations of hipcub /
# rocPRIM inside this library can never be interposed by another copy in the host process (torch, PETSc ...).
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off",
         "-munsafe-fp-atomics", "-fvisibility=hidden", "-fvisibility-inlines-hidden", "-Wall"]


def build(force=False):
    if os.environ.get("NLPS_GPU_LIB"):
        return LIB
    if not force and os.path.exists(LIB) and all(
            os.path.getmtime(LIB) >= os.path.getmtime(d) for d in DEPS if os.path.exists(d)):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc] + FLAGS + ["-o", LIB, SRC, IO_SRC])
    return LIB


if __name__ == "__main__":
    print(build(force=True))
