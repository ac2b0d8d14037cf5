#!/usr/bin/env python3
"""Developer tool: per-kernel times of the explicit step on the bench cloud for one build of the library.

    [NLPS_GPU_LIB=build/exp/lib_X.so] python tools/kbench.py [--cells 50] [--law nh|hencky|dp|mixed] [--phases]

--phases needs a build with -DNLPS_PHASE_TIMING=1 (tools/build_variant.sh ph -DNLPS_PHASE_TIMING=1)."""
import argparse
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, default=50)
ap.add_argument("--law", default="nh")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--phases", action="store_true")
ap.add_argument("--no-order", action="store_true")
ap.add_argument("--det", action="store_true", help="deterministic mode")
ap.add_argument("--stir", type=int, default=0, help="untimed shear steps first (DESIGN.md stirred cloud)")
ap.add_argument("--resort", action="store_true", help="one periodic re-sort of the fused step before the timed steps")
ap.add_argument("--adaptive", type=float, default=0.0, help="adaptive re-sort budget (also during the stir and the timed steps)")
ap.add_argument("--opt", action="append", default=[], help="NAME=VALUE for nlps_gpu_debug_option (fuse_search, lazy_nodal, ...)")
ap.add_argument("--tag", default=os.path.basename(os.environ.get("NLPS_GPU_LIB", "product")))
a = ap.parse_args()
nlps = importlib.import_module("nl-partsol_amd.nlps")
synth = importlib.import_module("nl-partsol_amd.synth")
case = bench.build_case(0, 1, a.cells)
grav = None
if a.law != "nh":
    dp = synth.drucker_prager_material()
    mats = {"hencky": [{"type": 1, "E": 1.0e7, "nu": 0.3}], "dp": [dp],
            "mixed": [{"type": 0, "E": 2.0e4, "nu": 0.3}, {"type": 1, "E": 1.0e4, "nu": 0.25}, dp],
            "layers": [{"type": 0, "E": 2.0e4, "nu": 0.3}, {"type": 1, "E": 1.0e4, "nu": 0.25}, dp]}[a.law]
    case["materials"] = mats
    if a.law in ("dp", "mixed", "layers"):
        case["cloud"]["kappa_n"][:] = dp["kappa_0"]
        case["cloud"]["vel"][:] = 0.0
        grav = [0.0, 0.0, -9.81]
    if a.law == "mixed":  # laws interleaved particle by particle (worst case)
        case["cloud"]["matidx"] = (np.arange(case["cloud"]["x"].shape[0]) % 3).astype(np.int32)
    if a.law == "layers":  # three horizontal layers of different materials
        z = case["cloud"]["x"][:, 2]
        case["cloud"]["matidx"] = np.minimum(2, ((z - z.min()) / (z.max() - z.min() + 1e-9) * 3).astype(np.int32))
if a.stir:
    x = case["cloud"]["x"]
    c = x.mean(axis=0)
    v = np.zeros_like(x)
    v[:, 0] = 10.0 * (x[:, 2] - c[2]) / (0.5 * a.cells)
    v[:, 1] = 10.0 * (x[:, 0] - c[0]) / (0.5 * a.cells)
    v[:, 2] = -3.0
    case["cloud"]["vel"] = v
    case["materials"] = [{"type": 0, "E": 1.0e5, "nu": 0.3}]
nst = a.steps + 10
S = nlps.Solver(3, case["grid_n"], case["origin"], case["h"], case["cloud"], case["materials"], nsteps=nst)
nodes = synth.plane_nodes(case["grid_n"], 2, 0)
bcs = nlps.BccSet([{"nodes": nodes, "dim": 3, "dir": np.ones((3, nst), dtype=np.int32), "value": np.zeros((3, nst))}])
if a.no_order:
    S.L.nlps_gpu_debug_set_tile_ordering.argtypes = [C.c_void_p, C.c_int]
    S.L.nlps_gpu_debug_set_tile_ordering(S.h, 0)
if a.det:
    S.set_deterministic(True)
for kv in a.opt:
    S.debug_option(kv.split("=")[0], float(kv.split("=")[1]))
S.initialise_shapefun()
E = max(m["E"] for m in case["materials"])
dt = 0.1 * case["h"] / np.sqrt(E / 1000.0)
if a.stir:
    bcs = nlps.BccSet([])
    dt = 2e-3
    S.set_resort_interval(0)
    if a.adaptive:
        S.set_resort_interval(1000000)
        S.set_adaptive_resort(a.adaptive)
    for t in range(a.stir):
        S.explicit_step(bcs, 0, dt)
        if a.adaptive and t % 8 == 7:
            c_, d_ = S.debug_displaced()
            print("   stir step %d: displaced %.3f debt %.3f" % (t, c_ / case["cloud"]["x"].shape[0], d_), flush=True)
for t in range(5):
    S.explicit_step(bcs, t, dt, 0.5, grav)
if a.resort:
    S.set_resort_interval(1)
    for t in range(2):
        S.explicit_step(bcs, 4, dt, 0.5, grav)
    S.set_resort_interval(0)
if a.phases:
    S.L.nlps_gpu_debug_phases.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    out = np.zeros(32, dtype=np.uint64)
    S.L.nlps_gpu_debug_phases(S.h, out.ctypes.data, 1)
S.set_timing(True)
k = np.zeros(8)
for t in range(5, 5 + a.steps):
    S.explicit_step(bcs, t, dt, 0.5, grav)
    k += np.array(S.get_timing())
k /= a.steps
k[:4] = np.maximum(k[:4] - k[5], 0)
if True:
    print("%s law=%s cells=%d: search %.3f K2 %.3f K3 %.3f K5 %.3f nodal %.3f sum %.3f ms | flags %x" %
          (a.tag, a.law, a.cells, k[0], k[1], k[2], k[3], k[4], k[:5].sum(), S.status_flags()), flush=True)
if a.phases:
    S.L.nlps_gpu_debug_phases(S.h, out.ctypes.data, 1)
    o = out.astype(np.float64) / a.steps
    nw = case["cloud"]["x"].shape[0] / 64
    names = {0: "K2 prologue", 1: "K2 loads+mask", 2: "K2 newton", 3: "K2 predictor+scatter", 4: "K2 end barrier", 5: "K2 flush",
             8: "K3 prologue", 9: "K3 loads+factors", 10: "K3 moments+gather",
             11: "K3 F+stress+B", 12: "K3 scatter", 13: "K3 end barrier", 14: "K3 flush"}
    print("  K2 Newton evaluations per particle %.3f" % (o[7] / case["cloud"]["x"].shape[0]))
    for kk, nm in names.items():
        print("  %-30s %10.0f cycles/wave" % (nm, o[kk] / nw))
