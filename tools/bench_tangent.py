#!/usr/bin/env python3
"""Measurement of the tangent-assembly row: nlps_gpu_tangent_assemble + nlps_gpu_tangent_coo on
synthetic Neo-Hookean clouds, with the oracle's dense restatement timed beside it on a bounded sample.
usage: python tools/bench_tangent.py [--reps 5]      (needs an MI355X; prints one JSON line per case)"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run_case(name, ndim, cells, lo, blk, reps):
    import util
    nlps = importlib.import_module("nl-partsol_amd.nlps")
    case = util.make_case(ndim, cells, lo, blk, material=util.NH, velocity=[0.0] * (ndim - 1) + [-1.0])
    S = util.gpu_setup(case, nsteps=1)
    gb = nlps.BccSet([util.dirichlet_plane(case, ndim - 1, lo[ndim - 1], 1)])
    S.active_masks(gb, 0)
    rng = np.random.default_rng(0)
    dU = 1e-3 * rng.normal(size=S.nactive * ndim)
    S.local_compatibility_conditions(dU)
    S.constitutive_update()
    Mv = S.compute_nodal_lumped_mass()
    S.jacobian_evaluation(1.0, Mv, True)  # warm-up (allocates the stencil array)
    S.synchronize()
    t_asm, t_all = [], []
    import ctypes as C
    for _ in range(reps):
        nnz = C.c_longlong(0)
        t0 = time.perf_counter()
        S._chk(S.L.nlps_gpu_tangent_assemble(S.h, C.byref(nnz)))  # synchronises (status check)
        t1 = time.perf_counter()
        t_asm.append(t1 - t0)
        t0 = time.perf_counter()
        rows, cols, vals = S.jacobian_evaluation(1.0, Mv, True)
        t_all.append(time.perf_counter() - t0)
    nn, _ = S.download_lists()
    pairs = float((nn.astype(np.int64) ** 2).sum())
    np_ = case["cloud"]["x"].shape[0]
    ta = float(np.median(t_asm))
    out = {"metric": "tangent node-pair blocks/s (assembly kernel, incl. clearing the stencil array)",
           "workload": name, "ndim": ndim, "particles": int(np_), "mean_neighbours": float(nn.mean()),
           "pair_blocks": pairs, "nnz": int(rows.size), "assemble_ms": 1e3 * ta,
           "assemble_plus_coo_download_ms": 1e3 * float(np.median(t_all)), "value": pairs / ta,
           "unit": "blocks/s", "dtype": "f64"}
    S.close()
    return out, case


def cpu_sample(ndim, budget_particles):
    """oracle restatement (dense, serial like the reference's omp-critical MatSetValues) on a small cloud"""
    import util
    o = util.orc()
    if ndim == 2:
        case = util.make_case(2, [16, 16], [4, 4], [8, 8], material=util.NH)
    else:
        case = util.make_case(3, [9, 9, 9], [3, 3, 3], [3, 3, 3], material=util.NH)
    M, P, prm, mats = util.oracle_setup(case)
    n2m, na = o.active_nodes(M)
    o.compatibility(np.zeros(na * ndim), None, P, M, n2m)
    o.constitutive(P, mats, prm)
    t0 = time.perf_counter()
    K, pat, st = o.tangent_matrix(P, M, mats, n2m, None, na, with_pattern=False)
    dt = time.perf_counter() - t0
    pairs = float((P["nn"].astype(np.int64) ** 2).sum())
    return {"value": pairs / dt, "unit": "blocks/s", "cores": 1, "kind": "port",
            "sample": "%d-D, %d particles, dense matrix %d^2" % (ndim, P.np, na * ndim)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("needs an MI355X (no CPU fallback)")
    cases = [("2-D bar 10k particles (BASELINE configs[0] shape)", 2, [60, 60], [5, 5], [50, 50]),
             ("2-D 250k particles", 2, [260, 260], [5, 5], [250, 250]),
             ("3-D 27k particles", 3, [25, 25, 25], [5, 5, 5], [15, 15, 15])]
    cpu = {2: cpu_sample(2, 0), 3: cpu_sample(3, 0)}
    for name, ndim, cells, lo, blk in cases:
        out, _ = run_case(name, ndim, cells, lo, blk, a.reps)
        out["cpu_baseline"] = cpu[ndim]
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
