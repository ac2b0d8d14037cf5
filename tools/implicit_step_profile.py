#!/usr/bin/env python3
"""Developer tool: the calls of ONE implicit Newmark time step as the maintained driver makes them
(U-Newmark-beta.c:192-409: search, masks, lumped mass, nodal field, initial guess, then per Newton iterate the residual and
the tangent, then kinetic increments, particle update and roll), device-resident vectors, no linear solve: where the
library's share of such a step goes.    python tools/implicit_step_profile.py [cells=50] [newton=3] [residuals_per_iterate=2]"""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

nlps = importlib.import_module("nl-partsol_amd.nlps")
synth = importlib.import_module("nl-partsol_amd.synth")
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 50
newton = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nres = int(sys.argv[3]) if len(sys.argv) > 3 else 2
case = bench.build_case(0, 1, cells)
nst = 4
S = nlps.Solver(3, case["grid_n"], case["origin"], case["h"], case["cloud"], case["materials"], nsteps=nst)
S.initialise_shapefun()
nodes = synth.plane_nodes(case["grid_n"], 2, 0)
gb = nlps.BccSet([{"nodes": nodes, "dim": 3, "dir": np.ones((3, nst), dtype=np.int32), "value": np.zeros((3, nst))}])
beta, gamma, dt = 0.25, 0.5, 1.0e-3
a = [1 / (beta * dt * dt), 1 / (beta * dt), (1 - 2 * beta) / (2 * beta), gamma / (beta * dt), 1 - gamma / beta,
     (1 - gamma / (2 * beta)) * dt]
T = {}


def timed(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    S.synchronize()
    torch.cuda.synchronize()
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
    return r


for step in range(1, 3):
    T.clear()
    timed("local_search", S.local_search)
    timed("active_masks", lambda: S.active_masks(gb, step, download=False))
    n = S.nactive * 3
    dev = lambda: torch.zeros(n, dtype=torch.float64, device="cuda")  # noqa: E731
    M, V, A, R = dev(), dev(), dev(), dev()
    timed("lumped_mass", lambda: S.compute_nodal_lumped_mass(out=M))
    timed("nodal_field_n", lambda: S.get_nodal_field_n(M, V, A))
    dU = torch.from_numpy(timed("form_initial_guess", lambda: S.form_initial_guess(V, A, dt, gb, step))).cuda()
    for it in range(newton):
        for q in range(nres):  # (SNES: the iterate and its line-search trials)
            timed("residual", lambda: S.lagrangian_evaluation(dU, V, A, M, a, [0.0, 0.0, -9.81], None, step, 1.0, None, out=R))
        timed("tangent", lambda: S.jacobian_evaluation(a[0], M, True, on_device=True))
        dU = dU * (1.0 - 1e-3)  # (stands in for the linear solve's update)
    dV, dA = dev(), dev()
    dVn, dAn = timed("kinetic_increments", lambda: S.compute_nodal_kinetic_increments(dU, V, A, a))
    dV, dA = torch.from_numpy(dVn).cuda(), torch.from_numpy(dAn).cuda()
    timed("update_kinetics", lambda: S.update_particles_kinetics_FLIP_PIC(1.0, dU, V, dV, dA))
    timed("roll_state", S.update_particles_internal_variables)
    tot = sum(T.values())
    print("step %d, %d particles, %d Newton iterates x (%d residuals + 1 tangent): %.1f ms in the library, flags %x" % (
        step, case["cloud"]["x"].shape[0], newton, nres, 1e3 * tot, S.status_flags()))
    for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
        print("   %-20s %9.3f ms  %5.1f %%" % (k, 1e3 * v, 100 * v / tot))
