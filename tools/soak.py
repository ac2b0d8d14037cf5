#!/usr/bin/env python3
"""Developer soak: the bench cloud (1 M particles) falls 5 cells onto the floor and keeps deforming, N steps with the
library's own re-sort policy; the same run with the riding search in its old form (defer_ranks = 0) must stay on the same
index maps and within rounding of the same state."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

nlps = importlib.import_module("nl-partsol_amd.nlps")
synth = importlib.import_module("nl-partsol_amd.synth")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
cells = int(sys.argv[2]) if len(sys.argv) > 2 else 50
out = []
for defer in (1, 0):
    case = bench.build_case(0, 1, cells)
    case["materials"] = [{"type": 0, "E": 1.0e6, "nu": 0.3}]
    S = nlps.Solver(3, case["grid_n"], case["origin"], case["h"], case["cloud"], case["materials"], nsteps=1)
    S.debug_option("defer_ranks", defer)
    nodes = synth.plane_nodes(case["grid_n"], 2, 0)
    bcs = nlps.BccSet([{"nodes": nodes, "dim": 3, "dir": np.ones((3, 1), dtype=np.int32), "value": np.zeros((3, 1))}])
    S.initialise_shapefun()
    m0 = case["cloud"]["mass"].sum()
    t0 = time.perf_counter()
    for t in range(N):
        S.explicit_step(bcs, 0, 1e-3, 0.5, [0.0, 0.0, -9.81])
        if t % 200 == 199:
            nod = S.explicit_nodal()
            print("defer %d step %d: flags %x, nodal mass / particle mass - 1 = %.2e, %.3f ms/step" % (
                defer, t + 1, S.status_flags(), nod["mass"].reshape(-1, 3)[:, 0].sum() / m0 - 1.0,
                1e3 * (time.perf_counter() - t0) / (t + 1)), flush=True)
    st = S.download_state(["x", "vel", "F_n", "J_n", "I0"])
    assert S.status_flags() == 0
    out.append(st)
    S.close()
a, b = out
print("I0 equal:", np.array_equal(a["I0"], b["I0"]), " max |dx| %.3e  max |dF| %.3e  z range %.2f..%.2f  J range %.3f..%.3f" % (
    np.abs(a["x"] - b["x"]).max(), np.abs(a["F_n"] - b["F_n"]).max(), a["x"][:, 2].min(), a["x"][:, 2].max(), a["J_n"].min(), a["J_n"].max()))
