"""Writes tests/golden/nh3d_abi.bin: the inputs of the "nh3d" golden case (make_golden.py) and its end state after four
explicit steps (from nh3d.npz, i.e. from the oracle) as one flat little-endian file that a plain C program can read
(tests/c/abi_step.c).  Layout: int32 header {magic 0x4e4c5053, ndim, n0, n1, n2, np, nsteps, nbc}, float64 {h, dt,
E, nu, g0, g1, g2}, then float64 arrays x[np*d], vel[np*d], mass[np], vol0[np], rho[np], int32 bc_nodes[nbc],
then the expected end state: int32 I0[np], float64 x[np*d], vel[np*d], F_n[np*9], stress[np*9].

    python tests/golden/make_abi_fixture.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def main():
    case = mg.build("nh3d")
    g = np.load(os.path.join(HERE, "nh3d.npz"))
    c = case["cloud"]
    d, npart = 3, c["x"].shape[0]
    bc = mg.bcs_for(case)[0]["nodes"].astype(np.int32)
    m = case["materials"][0]
    with open(os.path.join(HERE, "nh3d_abi.bin"), "wb") as f:
        np.array([0x4E4C5053, d] + list(case["grid_n"]) + [npart, mg.NSTEPS, bc.size], dtype=np.int32).tofile(f)
        np.array([case["h"], mg.dt_for(case), m["E"], m["nu"], 0.0, 0.0, -9.81], dtype=np.float64).tofile(f)
        for k in ("x", "vel", "mass", "vol0", "rho"):
            np.ascontiguousarray(c[k], dtype=np.float64).tofile(f)
        bc.tofile(f)
        g["end_I0"].astype(np.int32).tofile(f)
        for k in ("end_x", "end_vel", "end_F_n", "end_stress"):
            np.ascontiguousarray(g[k], dtype=np.float64).tofile(f)
    print("nh3d_abi.bin", npart, "particles,", bc.size, "Dirichlet nodes")


if __name__ == "__main__":
    main()
