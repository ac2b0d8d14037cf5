"""Generates tests/golden/*.npz: small input/output vectors for the hot path.

IMPORTANT: these vectors come from the build's own CPU oracle (oracle/nlps_oracle.c), NOT from the
reference: the reference cannot be compiled in the build image (its path needs <lapacke.h> + LAPACK,
DESIGN.md "oracle") and ships no golden vectors of its own.  They pin the oracle and
the HIP path against regressions and let the GPU box check the HIP path without the oracle's sources.

    OMP_NUM_THREADS=1 python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
os.environ.setdefault("OMP_NUM_THREADS", "1")
import util  # noqa: E402

CASES = {
    "nh2d": dict(ndim=2, cells=[12, 11], lo=[3, 3], blk=[6, 5], material=util.NH, velocity=[0.0, -10.0]),
    "nh3d": dict(ndim=3, cells=[9, 9, 8], lo=[3, 3, 3], blk=[3, 3, 2], material=util.NH, velocity=[0.0, 0.0, -10.0]),
    "hencky2d": dict(ndim=2, cells=[12, 11], lo=[3, 3], blk=[6, 5], material=util.HENCKY, velocity=[0.0, -10.0]),
    "dp3d": dict(ndim=3, cells=[9, 9, 8], lo=[3, 3, 3], blk=[3, 3, 2], material=util.DP, velocity=[0.0, 0.0, -0.2]),
    "dp2d": dict(ndim=2, cells=[12, 11], lo=[3, 3], blk=[6, 5], material=util.DP, velocity=[0.0, -0.2]),
}
NSTEPS = 4


def build(name):
    c = CASES[name]
    case = util.make_case(c["ndim"], c["cells"], c["lo"], c["blk"], material=c["material"], velocity=c["velocity"])
    return case


def bcs_for(case):
    return [util.dirichlet_plane(case, case["ndim"] - 1, 2, NSTEPS)]


def dt_for(case):
    return 0.1 * case["h"] / np.sqrt(case["materials"][0]["E"] / 1000.0)


def run_oracle(case):
    o = util.orc()
    M, P, prm, mats = util.oracle_setup(case)
    out = {"init_I0": P["I0"].copy(), "init_nn": P["nn"].copy(), "init_list": P["list"].copy(),
           "init_lambda": P["lambda"].copy(), "init_beta": P["beta"].copy(), "init_active": M.active().copy()}
    grav = [0.0] * (case["ndim"] - 1) + [-9.81]
    st = o.ExplicitStepper(P, M, mats, prm, o.BccSet(bcs_for(case)), NSTEPS, gravity=grav)
    for t in range(NSTEPS):
        assert st.step(t, dt_for(case)) == 0
    for k in ("x", "dis", "vel", "acc", "F_n", "stress", "J_n", "rho", "W", "lambda", "b_e_n", "kappa_n", "eps_n"):
        out["end_" + k] = P[k].copy()
    out["end_I0"] = P["I0"].copy()
    out["end_nn"] = P["nn"].copy()
    out["end_list"] = P["list"].copy()
    for k in ("mass", "dU", "force", "accel", "reaction"):
        out["nodal_" + k] = st.nodal(k).copy()
    out["nactive"] = np.array([st.out.nactive])
    return out


if __name__ == "__main__":
    for name in CASES:
        case = build(name)
        out = run_oracle(case)
        out["in_x"] = case["cloud"]["x"]
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, case["cloud"]["x"].shape, "->", name + ".npz")
