#!/usr/bin/env python3
"""Developer probe: how much of a step is kernel tail?  Two independent clouds of N/2 particles stepped on two streams at
once against one cloud of N on one stream: what the second form loses is what overlapping the tails of K2 / K3 / K5 with
the next kernel's start could recover at most ("the tail")."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

nlps = importlib.import_module("nl-partsol_amd.nlps")
synth = importlib.import_module("nl-partsol_amd.synth")


def make(cells_xy, cells_z, stream, nst):
    margin = 5
    gc = [cells_xy + 2 * margin, cells_xy + 2 * margin, cells_z + 2 * margin]
    cloud = synth.make_cloud(3, gc, [margin] * 3, [cells_xy, cells_xy, cells_z], h=1.0, jitter=0.05, seed=12345,
                             velocity=[0.0, 0.0, -10.0])
    gn = synth.grid_nodes(gc)
    S = nlps.Solver(3, gn, [0.0] * 3, 1.0, cloud, [{"type": 0, "E": 1.0e7, "nu": 0.3}], nsteps=nst, stream=stream)
    nodes = synth.plane_nodes(gn, 2, 0)
    bcs = nlps.BccSet([{"nodes": nodes, "dim": 3, "dir": np.ones((3, nst), dtype=np.int32), "value": np.zeros((3, nst))}])
    S.initialise_shapefun()
    S.set_resort_interval(0)
    return S, bcs


nst, steps = 40, 20
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
dt = 1e-3
for label, sets in (("one cloud 50x50x50", [(50, 50, s1)]), ("two clouds 50x50x25, two streams", [(50, 25, s1), (50, 25, s2)]),
                    ("two clouds 50x50x25, ONE stream", [(50, 25, s1), (50, 25, s1)])):
    solvers = [make(cxy, cz, st.cuda_stream, nst) for cxy, cz, st in sets]
    for t in range(5):
        for S, b in solvers:
            S.explicit_step(b, t, dt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(5, 5 + steps):
        for S, b in solvers:
            S.explicit_step(b, t, dt)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    print("%-40s %.3f ms per step of %d particles" % (label, ms, sum(S.np for S, _ in solvers)), flush=True)
    for S, _ in solvers:
        S.close()
