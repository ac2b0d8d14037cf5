#!/usr/bin/env python3
"""bench.py — particle-steps/s of the explicit P2G + stress + G2P step (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one fused explicit predictor-corrector particle step (search + LME Newton, P2G of mass
and momentum, G2P gradient + F-update + Kirchhoff stress, P2G of the internal force, G2P kinematic
update) over the synthetic cloud.  N = 1 runs BASELINE configs[1]: 3-D elastic cube impact,
1 M particles (50^3 cells x 8), LME, Neo-Hookean, inside a 60^3-cell grid with a rigid floor.
N > 1 is WEAK scaling by default: every rank owns one such 1 M-particle block, stacked along z (the slab axis),
ghost-node layers exchanged with the two z-neighbours over RCCL (the library's own ncclSend / ncclRecv path);
`--scaling strong --particles-total 8000000` splits ONE cube (BASELINE configs[3], 100^3 cells x 8) into N z-slabs
instead, so N = 1, 2, 4, 8 all run the same 8 M-particle job.  The line carries BOTH records (`weak`, `strong`; the
headline fields are the one --scaling names), for N > 1 a partitioned-vs-whole check over the real communicator
(`partition_check`), the rank count RCCL reports, the overlap form that survived the warm-up and per-rank kernel and
exchange-wait times.
Inputs are resident in HBM before the timed region; the timed region holds ONE physical re-sort of the particle
arrays (the library's housekeeping, default cadence one per 50 steps: charged here at one per K steps).
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic bytes per particle per stage, FP64, 3-D (SURVEY.md §8d / BASELINE.md §3.4)
BYTES_3D = {"S1": 100, "S2": 123, "S3": 371, "S4": 215, "S5": 408, "step": 1217}
BYTES_2D = {"S1": 76, "S2": 94, "S3": 228, "S4": 136, "S5": 276, "step": 810}  # SURVEY 8d, the 2-D column
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PARTITION_TOL = 1e-9        # partitioned vs whole cloud (max relative field error): ONE constant for the step-down and the abort
FP64_VEC_PEAK_TFLOPS = 78.6  # half the 157.3 TF FP32 vector peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cells", type=int, default=50, help="cells per axis of one rank's block (50 -> 1 M particles)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-cells", type=int, default=20, help="cells per axis of the CPU-baseline sample (20 -> 64 000 particles, ~20 s of CPU work)")
    ap.add_argument("--halo", choices=["p2p", "allreduce"], default="p2p")
    ap.add_argument("--halo-impl", choices=["c", "torch"], default="c",
                    help="N > 1: c = the library's own RCCL exchange (nlps_gpu_rccl_attach: ncclSend/ncclRecv on a "
                         "library-owned stream, no Python in the step); torch = the callback through torch.distributed")
    ap.add_argument("--migrate-every", type=int, default=0,
                    help="N > 1: hand particles whose closest node left the rank's slab to the neighbour every k "
                         "steps (0 = never: the default 25 steps move the cloud by 0.25 cells)")
    ap.add_argument("--workload", choices=["step", "tangent", "residual", "step2d"], default="step",
                    help="step: the explicit particle step (the headline metric); tangent: the Neo-Hookean tangent "
                         "assembly of the implicit driver (SURVEY 8f n1), one JSON line per case; residual: the implicit "
                         "driver's residual callback (nlps_gpu_lagrangian_evaluation), fused vs separate stages, one "
                         "JSON line per law (Neo-Hookean, Drucker-Prager); step2d: the 2-D explicit step alone (the "
                         "`secondary` record of the default run as its own line, for the profiler)")
    ap.add_argument("--residual-laws", default="nh,dp")
    ap.add_argument("--no-implicit", action="store_true", help="skip the `implicit` record of the default run (N = 1 only)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --cells^3 x 8 particles per rank; strong: --particles-total split into z-slabs")
    ap.add_argument("--particles-total", type=int, default=8000000, help="--scaling strong: size of the one job")
    ap.add_argument("--no-stirred", action="store_true", help="skip the stirred-cloud figure (N = 1 only)")
    ap.add_argument("--overlap", type=int, choices=[0, 1], default=1,
                    help="N > 1: run the halo exchanges behind the interior tiles (1: the fastest overlapped form that "
                         "survives the warm-up, one launch per stage, else split launches) or blocking in place (0)")
    ap.add_argument("--no-second-scaling", action="store_true",
                    help="only the record --scaling asks for (default: the weak AND the strong record in the one line)")
    ap.add_argument("--no-partition-check", action="store_true", help="N > 1: skip the partitioned-vs-whole check")
    ap.add_argument("--no-secondary", action="store_true", help="skip the 2-D 1 M-particle secondary figure (N = 1 only)")
    return ap.parse_args()


def build_case(rank, world, cells, margin=5, cells_z=None):
    """Rank's block: cells x cells x cells_z[rank] cells (8 particles each), stacked along z in a grid that holds the
    blocks of all ranks plus a margin.  cells_z: one thickness for all ranks or a list (unequal slabs)."""
    synth = importlib.import_module("nl-partsol_amd.synth")
    cells_z = cells if cells_z is None else cells_z
    cz = list(cells_z) if hasattr(cells_z, "__len__") else [int(cells_z)] * world
    gc = [cells + 2 * margin, cells + 2 * margin, sum(cz) + 2 * margin]
    lo = [margin, margin, margin + sum(cz[:rank])]
    cloud = synth.make_cloud(3, gc, lo, [cells, cells, cz[rank]], h=1.0, jitter=0.05, seed=12345 + rank,
                             velocity=[0.0, 0.0, -10.0])
    return {"ndim": 3, "cells": gc, "grid_n": synth.grid_nodes(gc), "origin": [0.0, 0.0, 0.0], "h": 1.0,
            "cloud": cloud, "materials": [{"type": 0, "E": 1.0e7, "nu": 0.3}], "block_lo": lo}


def cpu_baseline(cells, budget_s=30.0):
    """Times the oracle's explicit step (a from-scratch CPU port with OpenMP and the reference's omp-critical nodal
    accumulation, U-Newmark-beta.c:582-586), built with the reference's release flags (-Ofast -fopenmp,
    CMakeLists.txt:42,86), on a bounded 3-D sample of the same workload: one warm-up step, then the median of fifteen
    steps at 1 thread (about 10 s) and of five at this GPU's share of the host cores.  kind = "port": the reference's own path cannot be built in this image
    (LAPACK), so no port / reference ratio exists."""
    os.environ.pop("OMP_NUM_THREADS", None)
    from oracle import orc
    if getattr(orc, "_LIB", None) is None:
        orc.use_fast_build(True)
    synth = importlib.import_module("nl-partsol_amd.synth")
    margin = 5
    gc = [cells + 2 * margin] * 3
    gn = synth.grid_nodes(gc)
    M = orc.OracleMesh(3, gn, [0.0] * 3, 1.0)
    prm = orc.default_params()
    mats = orc.make_materials([{"type": 0, "E": 1.0e7, "nu": 0.3}])
    nsteps = 20
    nodes = synth.plane_nodes(gn, 2, 0)
    bcs = orc.BccSet([{"nodes": nodes, "dim": 3, "dir": np.ones((3, nsteps), dtype=np.int32),
                       "value": np.zeros((3, nsteps))}])
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    # "all cores" = this process's share of the host: a one-GPU box hands out 16 cores per GPU however many it has
    nall = min(ncpu, 16)
    rates = {}
    t_begin = time.perf_counter()
    npart = 0
    for nthr in sorted({1, nall}):
        orc.set_num_threads(nthr)
        cloud = synth.make_cloud(3, gc, [margin] * 3, [cells] * 3, velocity=[0.0, 0.0, -10.0])
        P = orc.OracleParticles(cloud)
        npart = P.np
        assert orc.initialize_lme(P, M, prm) == 0
        st = orc.ExplicitStepper(P, M, mats, prm, bcs, nsteps)
        t0 = time.perf_counter()
        assert st.step(0, 1e-3) == 0  # warm-up (first touch, caches)
        t_warm = time.perf_counter() - t0
        ts = []
        for t in range(1, 16 if nthr == 1 else 6):
            if ts and time.perf_counter() - t_begin + max(ts) > budget_s * (0.5 if nthr == 1 else 1.0):
                break  # bounded sample: the omp-critical accumulation can make a many-thread step very slow
            t0 = time.perf_counter()
            assert st.step(t, 1e-3) == 0
            ts.append(time.perf_counter() - t0)
        if not ts:
            ts = [t_warm]
        rates[nthr] = (P.np / float(np.median(ts)), len(ts))
    best = max(rates, key=lambda k: rates[k][0])
    return {"value": rates[best][0], "unit": "particle-steps/s", "cores": best, "kind": "port",
            "cores_note": "%d thread(s) gave the best rate; tried 1 and %d (this GPU's share) of the %d host cores "
                          "this process may use" % (best, nall, ncpu),
            "threads_1": rates[1][0], "threads_all": rates[nall][0], "threads_all_count": nall, "host_cores": ncpu,
            "flags": "-Ofast -fopenmp (the reference's CMAKE_C_FLAGS_RELEASE + OpenMP)",
            "port_over_reference": None,
            "sample": "%d^3 cells x 8 = %d particles, 3-D LME Neo-Hookean explicit step, median of %d steps after one "
                      "warm-up step, at 1 thread and at %d threads (this GPU's share of the host cores); oracle/nlps_oracle.c keeps the "
                      "reference's omp-critical nodal accumulation, which is why more threads do not help; no "
                      "port/reference ratio exists (the reference path needs LAPACK, absent from the image)"
                      % (cells, npart, rates[best][1], nall)}


# ---------------------------------------------------------------------------------------------------
# --workload tangent: nlps_gpu_tangent_assemble + nlps_gpu_tangent_coo on synthetic Neo-Hookean clouds, with the
# oracle's dense restatement timed beside it on a bounded sample (the cpu_baseline leg)
# ---------------------------------------------------------------------------------------------------
def _nh_case(ndim, cells, lo, blk):
    synth = importlib.import_module("nl-partsol_amd.synth")
    cloud = synth.make_cloud(ndim, cells, lo, blk, h=1.0, jitter=0.05, seed=12345, velocity=[0.0] * (ndim - 1) + [-1.0])
    return {"ndim": ndim, "cells": cells, "grid_n": synth.grid_nodes(cells), "origin": [0.0] * ndim, "h": 1.0,
            "cloud": cloud, "materials": [{"type": 0, "E": 1.0e7, "nu": 0.3}]}


def tangent_cpu_baseline(ndim):
    from oracle import orc
    case = _nh_case(2, [16, 16], [4, 4], [8, 8]) if ndim == 2 else _nh_case(3, [9, 9, 9], [3, 3, 3], [3, 3, 3])
    M = orc.OracleMesh(ndim, case["grid_n"], case["origin"], case["h"])
    P = orc.OracleParticles(case["cloud"])
    prm = orc.default_params()
    mats = orc.make_materials(case["materials"])
    assert orc.initialize_lme(P, M, prm) == 0
    n2m, na = orc.active_nodes(M)
    orc.compatibility(np.zeros(na * ndim), None, P, M, n2m)
    orc.constitutive(P, mats, prm)
    t0 = time.perf_counter()
    K, pat, st = orc.tangent_matrix(P, M, mats, n2m, None, na, with_pattern=False)
    dt = time.perf_counter() - t0
    pairs = float((P["nn"].astype(np.int64) ** 2).sum())
    return {"value": pairs / dt, "unit": "blocks/s", "cores": 1, "kind": "port",
            "sample": "%d-D, %d particles, dense matrix %d^2 (oracle/nlps_oracle.c, serial like the reference's "
                      "omp-critical MatSetValues)" % (ndim, P.np, na * ndim)}


def bench_tangent(a):
    import ctypes as C
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (no CPU fallback)")
    nlps = importlib.import_module("nl-partsol_amd.nlps")
    synth = importlib.import_module("nl-partsol_amd.synth")
    cases = [("2-D bar 10k particles (BASELINE configs[0] shape)", 2, [60, 60], [5, 5], [50, 50]),
             ("2-D 250k particles", 2, [260, 260], [5, 5], [250, 250]),
             ("3-D 27k particles", 3, [25, 25, 25], [5, 5, 5], [15, 15, 15])]
    cpu = {} if a.no_cpu_baseline else {2: tangent_cpu_baseline(2), 3: tangent_cpu_baseline(3)}
    for name, ndim, cells, lo, blk in cases:
        case = _nh_case(ndim, cells, lo, blk)
        S = nlps.Solver(ndim, case["grid_n"], case["origin"], case["h"], case["cloud"], case["materials"], nsteps=1)
        S.initialise_shapefun()
        nodes = synth.plane_nodes(case["grid_n"], ndim - 1, lo[ndim - 1])
        gb = nlps.BccSet([{"nodes": nodes, "dim": ndim, "dir": np.ones((ndim, 1), dtype=np.int32),
                           "value": np.zeros((ndim, 1))}])
        S.active_masks(gb, 0)
        rng = np.random.default_rng(0)
        S.local_compatibility_conditions(1e-3 * rng.normal(size=S.nactive * ndim))
        S.constitutive_update()
        Mv = S.compute_nodal_lumped_mass()
        S.jacobian_evaluation(1.0, Mv, True)  # warm-up (allocates the stencil array)
        t_asm, t_all, t_dev = [], [], []
        S.jacobian_evaluation(1.0, Mv, True, on_device=True)  # (warm-up of the device-output form: torch's allocator)
        for _ in range(max(3, a.steps // 4)):
            nnz = C.c_longlong(0)
            t0 = time.perf_counter()
            S._chk(S.L.nlps_gpu_tangent_assemble(S.h, C.byref(nnz)))  # synchronises (status check)
            t_asm.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            rows, cols, vals = S.jacobian_evaluation(1.0, Mv, True)
            t_all.append(time.perf_counter() - t0)
            t0 = time.perf_counter()  # assembly + triplets written into device arrays (MatSetValuesCOO of a GPU matrix type)
            S.jacobian_evaluation(1.0, Mv, True, on_device=True)
            t_dev.append(time.perf_counter() - t0)
        nn, _ = S.download_lists()
        pairs = float((nn.astype(np.int64) ** 2).sum())
        ta = float(np.median(t_asm))
        out = {"metric": "tangent particle-pair blocks/s (assembly kernels, incl. clearing the stencil array)",
               "workload": name, "ndim": ndim, "particles": int(case["cloud"]["x"].shape[0]),
               "mean_neighbours": float(nn.mean()), "pair_blocks": pairs, "nnz": int(rows.size),
               "assemble_ms": 1e3 * ta, "assemble_plus_coo_download_ms": 1e3 * float(np.median(t_all)),
               "assemble_plus_coo_on_device_ms": 1e3 * float(np.median(t_dev)),
               "value": pairs / ta, "unit": "blocks/s", "dtype": "f64", "data": "synthetic", "n_gpus": 1}
        if cpu:
            out["cpu_baseline"] = cpu[ndim]
        print(json.dumps(out), flush=True)
        S.close()


# ---------------------------------------------------------------------------------------------------
# --workload residual: nlps_gpu_lagrangian_evaluation, the residual callback of the maintained implicit driver
# (__lagrangian_evaluation, U-Newmark-beta.c:970-1058: SNES runs it at every Newton iterate and line-search trial), as
# one fused device call against the composition of the separate stage entries, with device-resident and with host
# (VecGetArray) vectors; the oracle's composition of the same stages is the cpu_baseline leg
# ---------------------------------------------------------------------------------------------------
RESIDUAL_BYTES_3D = {"nh": BYTES_3D["S3"] + BYTES_3D["S4"], "dp": BYTES_3D["S3"] + BYTES_3D["S4"] + 176}  # S3 + S4 (+ S3' for D-P)


def residual_cpu_baseline(cells, law):
    """The oracle's composition of the stages of __lagrangian_evaluation (:1018-1036) on a bounded sample of the same
    cloud: compatibility + constitutive + internal forces (OpenMP, the reference's omp-critical accumulation) + the
    inertial term (numpy), 1 thread and this GPU's share of the host cores."""
    os.environ.pop("OMP_NUM_THREADS", None)
    from oracle import orc
    if getattr(orc, "_LIB", None) is None:
        orc.use_fast_build(True)
    synth = importlib.import_module("nl-partsol_amd.synth")
    margin = 5
    gc = [cells + 2 * margin] * 3
    gn = synth.grid_nodes(gc)
    M = orc.OracleMesh(3, gn, [0.0] * 3, 1.0)
    prm = orc.default_params()
    mat = {"type": 0, "E": 1.0e7, "nu": 0.3} if law == "nh" else synth.drucker_prager_material()
    mats = orc.make_materials([mat])
    cloud = synth.make_cloud(3, gc, [margin] * 3, [cells] * 3, velocity=[0.0, 0.0, -10.0])
    if law == "dp":
        cloud["kappa_n"][:] = mat["kappa_0"]
    P = orc.OracleParticles(cloud)
    assert orc.initialize_lme(P, M, prm) == 0
    n2m, na = orc.active_nodes(M)
    nodes = synth.plane_nodes(gn, 2, margin)
    bcs = orc.BccSet([{"nodes": nodes, "dim": 3, "dir": np.ones((3, 1), dtype=np.int32), "value": np.zeros((3, 1))}])
    d2m, _ = orc.active_dofs(n2m, na, 3, bcs, 0, 1)
    Mv = orc.lumped_mass(P, M, n2m, na)
    V, A = orc.nodal_field_n(Mv, P, M, n2m, d2m, na)
    rng = np.random.default_rng(0)
    dU = (2e-2 if law == "dp" else 1e-3) * rng.normal(size=na * 3)
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    nall = min(ncpu, 16)
    rates = {}
    for nthr in sorted({1, nall}):
        orc.set_num_threads(nthr)
        ts = []
        for it in range(4):
            t0 = time.perf_counter()
            assert orc.compatibility(dU, None, P, M, n2m) == 0
            assert orc.constitutive(P, mats, prm) == 0
            R, st = orc.internal_forces(P, M, n2m, d2m, na)
            R[d2m != -1] += (Mv * (4.0e6 * dU - 4.0e3 * V - A))[d2m != -1]
            if it > 0:
                ts.append(time.perf_counter() - t0)
        rates[nthr] = P.np / float(np.median(ts))
    best = max(rates, key=rates.get)
    return {"value": rates[best], "unit": "particle-residuals/s", "cores": best, "kind": "port",
            "threads_1": rates[1], "threads_all": rates[nall], "threads_all_count": nall, "host_cores": ncpu,
            "sample": "%d^3 cells x 8 = %d particles, 3-D LME %s: oracle compatibility + constitutive + internal forces + "
                      "inertial term, median of 3 evaluations after one warm-up, -Ofast -fopenmp"
                      % (cells, P.np, "Neo-Hookean" if law == "nh" else "Drucker-Prager")}


def bench_residual(a, stream=None, laws=("nh", "dp"), cpu=True, emit=True):
    """-> list of records (one per law); with emit, one JSON line per law"""
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (no CPU fallback)")
    nlps = importlib.import_module("nl-partsol_amd.nlps")
    synth = importlib.import_module("nl-partsol_amd.synth")
    if stream is None:
        tstream = torch.cuda.Stream()
        torch.cuda.set_stream(tstream)
        stream = tstream.cuda_stream
    recs = []
    for law in laws:
        case = build_case(0, 1, a.cells)
        grav = [0.0, 0.0, -9.81]
        amp = 1e-3
        if law == "dp":
            dp = synth.drucker_prager_material()
            case["materials"] = [dp]
            case["cloud"]["kappa_n"][:] = dp["kappa_0"]
            amp = 2e-2  # large enough that part of the cloud yields (the return mapping is what configs[4] is about)
        S = nlps.Solver(3, case["grid_n"], case["origin"], case["h"], case["cloud"], case["materials"], nsteps=1, stream=stream)
        S.initialise_shapefun()
        S.local_search()
        nodes = synth.plane_nodes(case["grid_n"], 2, case["block_lo"][2])
        gb = nlps.BccSet([{"nodes": nodes, "dim": 3, "dir": np.ones((3, 1), dtype=np.int32), "value": np.zeros((3, 1))}])
        S.active_masks(gb, 0)
        n = S.nactive * 3
        Mv = S.compute_nodal_lumped_mass()
        V, A = S.get_nodal_field_n(Mv)
        dt = 1.0e-3  # (an implicit step: 10 x the explicit limit of this cloud)
        alpha = [1 / (0.25 * dt * dt), 1 / (0.25 * dt), 1.0, 0.5 / (0.25 * dt), -1.0, 0.0]
        rng = np.random.default_rng(0)
        dU = amp * rng.normal(size=n)
        host = [dU, V, A, Mv]
        dev = [torch.from_numpy(np.ascontiguousarray(v)).cuda() for v in host]
        R_h = np.zeros(n)
        R_d = torch.zeros(n, dtype=torch.float64, device="cuda")
        S.set_timing(True)

        spikes = {}

        def run(vecs, R, flags, name=""):
            ktimes, walls = [], []
            for _ in range(a.warmup):
                S.lagrangian_evaluation(vecs[0], vecs[1], vecs[2], vecs[3], alpha, grav, out=R, flags=flags)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                t1 = time.perf_counter()
                S.lagrangian_evaluation(vecs[0], vecs[1], vecs[2], vecs[3], alpha, grav, out=R, flags=flags)
                walls.append(time.perf_counter() - t1)  # (every evaluation ends synchronised: it returns a status)
                ktimes.append(S.get_timing()[2])
            torch.cuda.synchronize()
            mean = 1e3 * (time.perf_counter() - t0) / a.steps
            spikes[name] = {"median_ms": 1e3 * float(np.median(walls)), "max_ms": 1e3 * float(np.max(walls))}
            return mean, float(np.mean(ktimes))

        fused_dev, k_ms = run(dev, R_d, 0, "fused_device")
        fused_host, _ = run(host, R_h, 0, "fused_host")
        fused_host_same, _ = run(host, R_h, S.LAGR_SAME_STEP)  # Un_dt, Un_dt2, M staged once per SNES solve
        S.set_timing(False)
        sep_dev, _ = run(dev, R_d, S.LAGR_SEPARATE, "separate_device")
        sep_host, _ = run(host, R_h, S.LAGR_SEPARATE)
        # the two forms agree (the parity tests hold them against the oracle; this is the bench checking itself)
        S.lagrangian_evaluation(dev[0], dev[1], dev[2], dev[3], alpha, grav, out=R_d, flags=0)
        r1 = R_d.cpu().numpy().copy()
        S.lagrangian_evaluation(dev[0], dev[1], dev[2], dev[3], alpha, grav, out=R_d, flags=S.LAGR_SEPARATE)
        agree = float(np.max(np.abs(r1 - R_d.cpu().numpy())) / max(np.max(np.abs(r1)), 1e-300))
        st = S.download_state(fields=["EPS_n1", "EPS_n"])
        yielding = float(np.mean(st["EPS_n1"] > st["EPS_n"]))
        # the other half of a Newton iterate: __jacobian_evaluation from the state this evaluation left, triplets written
        # into device arrays (MatSetValuesCOO of a GPU matrix type); a few repetitions, the first one allocates
        tan = []
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rr, cc, vv = S.jacobian_evaluation(alpha[0], dev[3], True, on_device=True)
            torch.cuda.synchronize()
            tan.append(1e3 * (time.perf_counter() - t0))
            nnz_t = int(vv.numel())
            del rr, cc, vv
        flags = S.status_flags()
        npart = S.np
        S.close()
        alg = RESIDUAL_BYTES_3D[law]
        achieved = npart * alg / (k_ms * 1e-3) / 1e9
        rec = {"metric": "particle-residuals/sec (G2P grad + F-update + stress + P2G internal force, one SNES residual evaluation)",
               "value": npart / (fused_dev * 1e-3), "unit": "particle-residuals/s", "n_gpus": 1, "steps": a.steps,
               "warmup": a.warmup, "ms_per_step": fused_dev, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f64", "data": "synthetic",
               "config": {"workload": "3-D cube, %d particles (%d^3 cells x 8), LME gamma=3, %s, one evaluation of "
                                      "__lagrangian_evaluation (U-Newmark-beta.c:970-1058) per step; value = fused entry, "
                                      "device-resident vectors" % (npart, a.cells, "Neo-Hookean E=1e7 nu=0.3" if law == "nh"
                                                                   else "Drucker-Prager (%.0f %% of the particles yielding)" % (100 * yielding)),
                          "active_dofs": int(n)},
               "fused_ms": {"device_vectors": fused_dev, "host_vectors_VecGetArray": fused_host,
                            "host_vectors_same_step": fused_host_same},
               "separate_stages_ms": {"device_vectors": sep_dev, "host_vectors_VecGetArray": sep_host},
               "per_call_wall_ms": spikes, "fused_over_separate": sep_dev / fused_dev, "fused_vs_separate_max_rel_diff": agree, "status_flags": flags,
               "tangent": {"ms_per_assembly_with_triplets_on_device": float(np.median(tan[1:])), "nnz": nnz_t,
                           "note": "nlps_gpu_tangent_assemble + nlps_gpu_tangent_coo into device arrays at this state "
                                   "(__jacobian_evaluation, U-Newmark-beta.c:1646-1830); python bench.py --workload tangent "
                                   "has the per-case lines"},
               "roofline": {"bound": "hbm", "kernel": "k3_tile<3,%d,3> (MODE 3)" % (0 if law == "nh" else 2), "achieved": achieved,
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                            "kernel_ms": k_ms, "algorithmic_bytes_per_particle": alg,
                            "note": "kernel_ms: HIP events on the handle's stream around the MODE-3 launch inside the timed "
                                    "evaluations (nlps_gpu_get_timing slot 2); algorithmic bytes = SURVEY 8d S3 + S4%s; "
                                    "counter traffic: profiles/r04_residual_*" % (" + S3'" if law == "dp" else "")}}
        if cpu and not a.no_cpu_baseline:
            rec["cpu_baseline"] = residual_cpu_baseline(a.cpu_cells, law)
        recs.append(rec)
        if emit:
            print(json.dumps(rec), flush=True)
    return recs


def stirred_figure(nlps, synth, a, stream):
    """The same cloud after 65 steps of a sheared velocity field (1.3 cells of relative drift): particles
    have changed closest node and tile, the memory runs of the tile lists are broken.  Timed like the headline: K steps
    with one physical re-sort in the middle."""
    import torch
    case = build_case(0, 1, a.cells)
    x = case["cloud"]["x"]
    c = x.mean(axis=0)
    v = np.zeros_like(x)
    v[:, 0] = 10.0 * (x[:, 2] - c[2]) / (0.5 * a.cells)
    v[:, 1] = 10.0 * (x[:, 0] - c[0]) / (0.5 * a.cells)
    v[:, 2] = -3.0
    case["cloud"]["vel"] = v
    soft = {"type": 0, "E": 1.0e5, "nu": 0.3}
    bcs = nlps.BccSet([])
    dt = 2e-3

    def run(library_policy):
        S = nlps.Solver(3, case["grid_n"], case["origin"], case["h"], case["cloud"], [soft], nsteps=1, stream=stream)
        if not library_policy:
            S.set_adaptive_resort(0.0)
            S.set_resort_interval(0)
        S.initialise_shapefun()
        for _ in range(65):
            S.explicit_step(bcs, 0, dt)
        torch.cuda.synchronize()
        if not library_policy:
            S.set_resort_interval(a.steps // 2 + 1)
        t0 = time.perf_counter()
        for i in range(a.steps):
            S.explicit_step(bcs, 0, dt)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / a.steps
        flags = S.status_flags()
        S.close()
        return ms, flags

    ms, flags = run(False)
    ms_lib, flags_lib = run(True)
    return {"ms_per_step": ms, "steps": a.steps, "shear_steps_before": 65, "resorts_in_timed_region": 1,
            "status_flags": flags | flags_lib,
            "workload": "bench cloud, Neo-Hookean E=1e5, sheared velocity field, no re-sort during the 65 shear steps",
            "library_policy_ms_per_step": ms_lib,
            "library_policy": "the library's own re-sort policy from the first step on (interval 50 + adaptive re-sort, "
                              "nlps_gpu_set_adaptive_resort budget 0.8): the re-sorts it decides on are inside the timed steps"}


# ---------------------------------------------------------------------------------------------------
# the explicit step on N ranks
# ---------------------------------------------------------------------------------------------------
ST_HALO = 16  # particle outside the rank's node window, or an exchange that never arrived (include/nlps_gpu.h)


class Ctx:
    """process-wide pieces: rank / world, torch.distributed, the shared HIP stream"""

    def __init__(self, a):
        import torch
        import torch.distributed as dist
        self.a, self.torch, self.dist = a, torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != a.gpus and self.world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % a.gpus)
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: no HIP device visible (no CPU fallback)")
        # rehearsal knobs (tests only): several ranks on the one GPU of a test box, gloo standing in for RCCL
        self.backend = os.environ.get("NLPS_BENCH_BACKEND", "nccl")
        local = int(os.environ.get("NLPS_BENCH_DEVICE", local))
        torch.cuda.set_device(local)
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=torch.device("cuda", local))
            else:
                dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)
        self.nlps = importlib.import_module("nl-partsol_amd.nlps")
        self.synth = importlib.import_module("nl-partsol_amd.synth")
        self.overlap_cap = 2  # the fastest exchange form the runs may start from (partition_check lowers it)
        self.halo_mod = importlib.import_module("nl-partsol_amd.halo")
        # One real (non-default) HIP stream shared by the library's kernels and the torch ops of the halo callback:
        # on the legacy default stream every torch op would synchronise with the library's own stream across queues
        # (measured: ~50 us per op).
        self.work_stream = torch.cuda.Stream()
        torch.cuda.set_stream(self.work_stream)
        self.stream = self.work_stream.cuda_stream
        self.dev = "cuda" if self.backend == "nccl" else "cpu"

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def reduce(self, value, op):
        """scalar over all ranks (op: "max" / "min" / "sum")"""
        if self.world == 1:
            return value
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op={"max": self.dist.ReduceOp.MAX, "min": self.dist.ReduceOp.MIN,
                                    "sum": self.dist.ReduceOp.SUM}[op])
        return float(t.item())

    def gather_rows(self, row):
        """one row of floats per rank -> list of rows on every rank"""
        if self.world == 1:
            return [list(map(float, row))]
        t = self.torch.tensor(list(map(float, row)), dtype=self.torch.float64, device=self.dev)
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [o.cpu().tolist() for o in out]


def make_solver(ctx, case, cells_z, margin, nsteps, want_mode):
    """The rank's solver with its ghost-layer exchange attached.  want_mode: 2 one launch per stage (library RCCL
    only), 1 split launches, 0 blocking exchanges.  -> (S, halo helper, info dict)"""
    a, torch, dist, rank, world = ctx.a, ctx.torch, ctx.dist, ctx.rank, ctx.world
    S = ctx.nlps.Solver(3, case["grid_n"], case["origin"], case["h"], case["cloud"], case["materials"],
                        nsteps=nsteps, stream=ctx.stream)
    info = {"halo_impl": "none", "halo_overlap_mode": None, "rccl_nranks": None}
    if world == 1:
        return S, None, info
    gn = case["grid_n"]
    lo, hi = ctx.halo_mod.SlabHalo.layer_ranges(world, cells_z, margin, gn[2])
    halo = ctx.halo_mod.SlabHalo(torch, dist, rank, world, gn[0] * gn[1], gn[2], lo, hi, mode=a.halo)  # migration helper
    nnodes = gn[0] * gn[1] * gn[2]
    band_lo, band_hi = halo.ghost_bands(rank)
    impl = a.halo_impl if ctx.backend == "nccl" else "torch"  # the gloo rehearsal has no RCCL
    if impl == "c":
        # the library creates its own communicator: rank 0's ncclUniqueId reaches the others through the process group
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            uid = torch.tensor(list(ctx.nlps.Solver.rccl_unique_id()), dtype=torch.uint8, device="cuda")
        dist.broadcast(uid, 0)
        ok = 1
        try:
            S.rccl_attach(bytes(uid.cpu().tolist()), rank, world, lo, hi, mode=1 if a.halo == "allreduce" else 0)
            n_r, r_r, mode_now = S.rccl_info()
            if n_r != world or r_r != rank:
                raise RuntimeError("RCCL reports rank %d of %d, the launcher %d of %d" % (r_r, n_r, rank, world))
            info["rccl_nranks"] = n_r
            if want_mode < mode_now:
                S.set_ghost_bands(band_lo, band_hi, want_mode)
        except Exception as e:  # e.g. librccl.so.1 not loadable on some rank
            print("rank %d: nlps_gpu_rccl_attach failed (%r)" % (rank, e), file=sys.stderr)
            ok = 0
        if ctx.reduce(ok, "min") == 0:  # all ranks fall back together to the callback through torch.distributed
            S.rccl_detach()
            impl = "torch"
            info["rccl_nranks"] = None
    if impl == "torch":
        def exchange(dptr, nfield, elem, kind, phase):
            return halo.exchange_ptr(dptr, nnodes * nfield, nfield, elem, kind, phase)

        S.set_halo_exchange(exchange)
        S._bench_exchange = exchange  # keeps the closure alive with the solver
        # the three exchanges of a step run behind the tiles that do not touch a ghost band (split launches at most:
        # the single-launch form needs the library's own exchange stream)
        S.set_ghost_bands(band_lo, band_hi, min(want_mode, 1))
        # per-step nodal work only on the layers this rank can touch (the grid grows with the rank count)
        S.set_node_window(lo[rank], hi[rank])
    info["halo_impl"] = impl
    info["halo_overlap_mode"] = S.rccl_info()[2] if impl == "c" else min(want_mode, 1)
    return S, halo, info


def floor_bcs(ctx, case, nsteps):
    nodes = ctx.synth.plane_nodes(case["grid_n"], 2, 0)
    return ctx.nlps.BccSet([{"nodes": nodes, "dim": 3, "dir": np.ones((3, nsteps), dtype=np.int32),
                             "value": np.zeros((3, nsteps))}])


def warm_solver(ctx, case, cells_z, margin, nsteps, dt, bcs, warmup):
    """Solver + exchange + warm-up steps, in the fastest overlap form that survives them: after the warm-up every rank
    reads its status word; a halo flag (16: an exchange that never arrived, a particle outside the window) or an error out
    of a step on ANY rank makes ALL ranks rebuild one form down, 2 -> 1 -> 0, in this process (nothing that has
    touched the GPU is ever re-executed)."""
    a, world = ctx.a, ctx.world
    want = ctx.overlap_cap if a.overlap else 0  # (the partition check may have lowered the cap)
    tried = []
    while True:
        S, halo, info = make_solver(ctx, case, cells_z, margin, nsteps, want)
        ok, flags = 1, 0
        try:
            S.initialise_shapefun()
            for t in range(warmup):
                S.explicit_step(bcs, t, dt)
            ctx.torch.cuda.synchronize()
            flags = S.status_flags()
        except Exception as e:
            print("rank %d: warm-up failed in overlap form %s (%r)" % (ctx.rank, info["halo_overlap_mode"], e), file=sys.stderr)
            ok = 0
        bad = ctx.reduce(0 if (ok and not (flags & ST_HALO)) else 1, "max") != 0
        tried.append({"halo_overlap_mode": info["halo_overlap_mode"], "ok": not bad})
        if not bad:
            info["overlap_forms_tried"] = tried
            return S, halo, info, flags
        mode_now = info["halo_overlap_mode"] or 0
        try:
            S.close()
        except Exception:
            pass
        if world == 1 or mode_now == 0:
            raise SystemExit("bench: the warm-up failed in every overlap form (%r)" % tried)
        want = mode_now - 1


def partition_check(ctx):
    """N > 1, before anything is timed: a small cloud split into N slabs runs four steps over the REAL communicator and
    exchange form of the run, and again as one cloud in one solver on rank 0; the partitioned fields must match
    (index maps bit for bit).  -> dict for the JSON line"""
    a, torch, dist, rank, world = ctx.a, ctx.torch, ctx.dist, ctx.rank, ctx.world
    cells, margin, nsteps = 8, 5, 4
    case = build_case(rank, world, cells, margin, cells)
    case["cloud"]["vel"][:] = [1.0, 0.5, -10.0]
    bc_nodes = ctx.synth.plane_nodes(case["grid_n"], 2, margin + 1)
    bcs = ctx.nlps.BccSet([{"nodes": bc_nodes, "dim": 3, "dir": np.ones((3, nsteps), dtype=np.int32),
                            "value": np.zeros((3, nsteps))}])
    dt = 0.4 / 100.0
    ref = None
    want, tried = (ctx.overlap_cap if a.overlap else 0), []
    while True:  # an overlapped exchange form whose partitioned run disagrees with the whole cloud is stepped down, 2 -> 1 -> 0
        res, info = _partition_run(ctx, case, cells, margin, nsteps, dt, bcs, want, ref)
        ref = res.pop("ref")
        good = res["max_rel_err"] <= PARTITION_TOL and res["index_maps_equal"] and not (res["status_flags"] & ST_HALO)
        tried.append({"halo_overlap_mode": info["halo_overlap_mode"], "ok": bool(good)})
        mode_now = info["halo_overlap_mode"] or 0
        if good or mode_now == 0:
            break
        want = mode_now - 1
    if len(tried) > 1:
        ctx.overlap_cap = min(ctx.overlap_cap, info["halo_overlap_mode"] or 0)
    res.update({"steps": nsteps, "particles": world * cells ** 3 * 8, "halo_impl": info["halo_impl"],
                "halo_overlap_mode": info["halo_overlap_mode"], "rccl_nranks": info["rccl_nranks"], "forms_tried": tried,
                "fields": "x, vel, F_n, Stress against one solver holding the whole cloud; I0 bit for bit"})
    return res


def _partition_run(ctx, case, cells, margin, nsteps, dt, bcs, want, ref):
    """one partitioned run of partition_check in the exchange form `want`; rank 0 also runs the whole cloud once (`ref`)"""
    torch, dist, rank, world = ctx.torch, ctx.dist, ctx.rank, ctx.world
    S, halo, info = make_solver(ctx, case, cells, margin, nsteps, want)
    S.set_resort_interval(2)
    S.initialise_shapefun()
    for t in range(nsteps):
        S.explicit_step(bcs, t, dt)
    flags = S.status_flags()
    st = S.download_state(fields=["x", "vel", "F_n", "Stress", "I0"])
    S.close()
    # (tensor collectives only: the rows of every rank have the same shape, and object collectives over NCCL are the one
    # thing the gloo rehearsal of this path cannot exercise)
    keys, widths = ("x", "vel", "F_n", "Stress", "I0"), (3, 3, 9, 9, 1)
    n_loc = st["x"].shape[0]
    flat = np.concatenate([np.asarray(st[k], dtype=np.float64).reshape(n_loc, w) for k, w in zip(keys, widths)], axis=1)
    rows = ctx.gather_rows(flat.reshape(-1))
    parts = []
    for r in rows:
        m = np.asarray(r, dtype=np.float64).reshape(n_loc, sum(widths))
        cols = np.cumsum((0,) + widths)
        part = {k: m[:, cols[i]:cols[i + 1]] for i, k in enumerate(keys)}
        part["I0"] = part["I0"].reshape(-1).astype(np.int64)
        parts.append(part)
    res = [0.0, 0.0, float(flags)]
    if rank == 0 and ref is None:
        clouds = [build_case(r, world, cells, margin, cells)["cloud"] for r in range(world)]
        whole = {}
        for k, v in clouds[0].items():
            whole[k] = np.concatenate([c[k] for c in clouds]) if isinstance(v, np.ndarray) else v
        whole["vel"][:] = [1.0, 0.5, -10.0]
        G = ctx.nlps.Solver(3, case["grid_n"], case["origin"], case["h"], whole, case["materials"], nsteps=nsteps,
                            stream=ctx.stream)
        G.set_resort_interval(2)
        G.initialise_shapefun()
        for t in range(nsteps):
            G.explicit_step(bcs, t, dt)
        ref = G.download_state(fields=["x", "vel", "F_n", "Stress", "I0"])
        G.close()
    if rank == 0:
        err = 0.0
        for k in ("x", "vel", "F_n", "Stress"):
            got = np.concatenate([p[k] for p in parts])
            err = max(err, float(np.max(np.abs(got - ref[k])) / max(np.max(np.abs(ref[k])), 1e-300)))
        res[0] = err
        res[1] = 0.0 if np.array_equal(np.concatenate([p["I0"] for p in parts]), ref["I0"]) else 1.0
    res = [ctx.reduce(v, "max") for v in res]
    return {"max_rel_err": res[0], "index_maps_equal": res[1] == 0.0, "status_flags": int(res[2]), "ref": ref}, info


def run_config(ctx, scaling, with_kernels=True):
    """One timed configuration: `weak` (--cells^3 x 8 particles per rank, stacked along z) or `strong` (ONE cube of
    --particles-total particles in N z-slabs).  Barrier + synchronise on both sides of exactly K steps, max over ranks.
    -> (record for the JSON line, the solver's case, per-kernel ms of rank 0's further steps)"""
    a, torch, rank, world = ctx.a, ctx.torch, ctx.rank, ctx.world
    margin = 5
    cells, cells_z = a.cells, a.cells
    if scaling == "strong":  # one cube of --particles-total particles, z-slabs of equal thickness
        cells = int(round((a.particles_total / 8.0) ** (1.0 / 3.0)))
        if cells < 8 * world:
            raise SystemExit("--scaling strong: %d cell layers are too few for %d ranks" % (cells, world))
        cells_z = [cells // world + (1 if r < cells % world else 0) for r in range(world)]  # 100 layers / 8 = 13,13,13,13,12,...
    case = build_case(rank, world, cells, margin, cells_z)
    total_steps = a.steps + a.warmup + 1
    bcs = floor_bcs(ctx, case, total_steps)
    dt = 0.1 * case["h"] / 100.0  # CFL 0.1, celerity sqrt(E/rho) = 100
    S, halo, info, _ = warm_solver(ctx, case, cells_z, margin, total_steps, dt, bcs, a.warmup)

    def step(t):
        if world > 1 and a.migrate_every > 0 and t % a.migrate_every == 0:
            cz = list(cells_z) if hasattr(cells_z, "__len__") else [cells_z] * world
            keep = (margin + sum(cz[:rank]) - 1 if rank > 0 else 0,
                    margin + sum(cz[:rank + 1]) + 1 if rank + 1 < world else case["grid_n"][2] - 1)
            if info["halo_impl"] == "c":
                S.rccl_migrate(*keep)   # select -> ncclSend / ncclRecv of counts and rows -> commit, inside the library
            else:
                halo.migrate(S, *keep)
        S.explicit_step(bcs, t, dt)

    t = a.warmup
    torch.cuda.synchronize()
    ctx.barrier()
    torch.cuda.synchronize()
    # housekeeping of the hot path inside the timed region: the library's own periodic physical re-sort, with its
    # interval set so that it fires exactly once in the K timed steps (library default: one per 50 steps)
    S.set_resort_interval(a.steps // 2 + 1)
    S.set_adaptive_resort(0.0)  # exactly one re-sort in the timed region (the adaptive policy would not fire here anyway)
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(t)
        t += 1
    torch.cuda.synchronize()
    ctx.barrier()
    torch.cuda.synchronize()
    elapsed = ctx.reduce(time.perf_counter() - t0, "max")
    flags = int(ctx.reduce(S.status_flags(), "max"))
    if flags:
        raise SystemExit("bench: particle failure flags 0x%x" % flags)
    npart = case["cloud"]["x"].shape[0]
    npart_total = int(ctx.reduce(npart, "sum"))
    rec = {"scaling": scaling, "n_gpus": world, "particles_total": npart_total, "particles_rank0": npart,
           "ms_per_step": 1e3 * elapsed / a.steps, "value": npart_total * a.steps / elapsed, "unit": "particle-steps/s",
           "steps": a.steps, "warmup": a.warmup, "grid_nodes": int(np.prod(case["grid_n"])),
           "cells": [cells, cells, int(sum(cells_z)) if hasattr(cells_z, "__len__") else int(cells_z) * world],
           "halo": a.halo if world > 1 else "none", "halo_impl": info["halo_impl"],
           "halo_overlap_mode": info["halo_overlap_mode"], "rccl_nranks": info["rccl_nranks"],
           "overlap_forms_tried": info.get("overlap_forms_tried"), "resorts_in_timed_region": 1}
    kms = None
    if with_kernels:
        # per-kernel times of further steps (HIP events on the launch stream), untimed.  Twenty of them: kernel time
        # swings by +-6 % with the position of the falling cube in the lattice (period ~17 steps: whole layers of
        # particles change tile together and break the memory-consecutive runs), so three steps read a phase, not the mean
        S.set_timing(True)
        kms = np.zeros(8)
        reps = 20
        for _ in range(reps):
            S.explicit_step(bcs, min(t, total_steps - 1), dt)
            kms += np.array(S.get_timing())
        kms /= reps
        S.set_timing(False)
        # a bracket of two event records measures record + dispatch latency besides the kernel: take the calibration
        # bracket (kms[5]: a do-nothing kernel of the same grid, measured in the same steps) off the single-kernel
        # brackets so that they read like rocprofv3's kernel durations
        ev_overhead = float(kms[5])
        kms[:4] = np.maximum(kms[:4] - ev_overhead, 0.0)
        names = ["search+activate", "lists+newton+p2g_mass_mom", "g2p_grad+stress+p2g_force", "g2p_update", "nodal",
                 "event_overhead", "exchange_wait"]
        rows = ctx.gather_rows(kms[:7])
        rec["per_rank_kernel_ms"] = [{n: float(v) for n, v in zip(names, r)} for r in rows]
    S.close()
    return rec, case, kms


def secondary_2d(ctx):
    """SECONDARY figure (not the headline): the same fused step in 2-D, 1 M particles (500^2 cells x 4, LME gamma = 3,
    Neo-Hookean), where the path sits near the roofline ridge: 810 algorithmic bytes per particle-step."""
    a, torch = ctx.a, ctx.torch
    cells, margin = 500, 5
    gc = [cells + 2 * margin] * 2
    cloud = ctx.synth.make_cloud(2, gc, [margin] * 2, [cells] * 2, h=1.0, jitter=0.05, seed=12345, velocity=[0.0, -10.0])
    nsteps = a.steps + a.warmup + 1
    S = ctx.nlps.Solver(2, ctx.synth.grid_nodes(gc), [0.0, 0.0], 1.0, cloud, [{"type": 0, "E": 1.0e7, "nu": 0.3}],
                        nsteps=nsteps, stream=ctx.stream)
    nodes = ctx.synth.plane_nodes(ctx.synth.grid_nodes(gc), 1, 0)
    bcs = ctx.nlps.BccSet([{"nodes": nodes, "dim": 2, "dir": np.ones((2, nsteps), dtype=np.int32),
                            "value": np.zeros((2, nsteps))}])
    S.initialise_shapefun()
    dt = 0.1 / 100.0
    for t in range(a.warmup):
        S.explicit_step(bcs, t, dt)
    S.set_resort_interval(a.steps // 2 + 1)
    S.set_adaptive_resort(0.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(a.warmup, a.warmup + a.steps):
        S.explicit_step(bcs, t, dt)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / a.steps
    flags = S.status_flags()
    npart = cloud["x"].shape[0]
    # per kernel (the one regime where north_star's "40 % of HBM on the P2G scatter" is physically in reach): HIP-event
    # brackets of further steps minus the calibration bracket, like the 3-D record; counter bytes from the rocprofv3 PMC
    # passes of `bench.py --workload step2d` (tools/profile.sh twod -> profiles/hbm_traffic_2d.json)
    S.set_timing(True)
    kms = np.zeros(8)
    reps = 20
    for _ in range(reps):
        S.explicit_step(bcs, nsteps - 1, dt)
        kms += np.array(S.get_timing())
    kms /= reps
    S.set_timing(False)
    kms[:4] = np.maximum(kms[:4] - float(kms[5]), 0.0)
    S.close()
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic_2d.json")))
    except Exception:
        pmc = {}
    names = ["search+activate", "lists+newton+p2g_mass_mom", "g2p_grad+stress+p2g_force", "g2p_update"]
    alg = [0, BYTES_2D["S1"] + BYTES_2D["S2"], BYTES_2D["S3"] + BYTES_2D["S4"], BYTES_2D["S5"]]
    per_kernel = {}
    for i in range(1, 4):
        t_s = kms[i] * 1e-3
        if t_s <= 0:
            continue
        e = {"kernel_ms": float(kms[i]), "algorithmic_bytes_per_particle": alg[i],
             "hbm_frac": npart * alg[i] / t_s / 1e9 / HBM_PEAK_GBS}
        if names[i] in pmc:
            e["traffic_bytes"] = pmc[names[i]] * npart / 1.0e6
            e["traffic_frac"] = e["traffic_bytes"] / t_s / 1e9 / HBM_PEAK_GBS
        per_kernel[names[i]] = e
    gbs = npart * 810.0 / (ms * 1e-3) / 1e9
    return {"workload": "2-D, %d particles (%d^2 cells x 4), LME gamma=3, Neo-Hookean, explicit step" % (npart, cells),
            "ms_per_step": ms, "value": npart / (ms * 1e-3), "unit": "particle-steps/s", "steps": a.steps,
            "algorithmic_bytes_per_particle_step": 810, "achieved_GBs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS,
            "status_flags": flags, "resorts_in_timed_region": 1, "per_kernel": per_kernel,
            "kernel_ms_all": {"search+activate": float(kms[0]), "lists+newton+p2g_mass_mom": float(kms[1]),
                              "g2p_grad+stress+p2g_force": float(kms[2]), "g2p_update": float(kms[3]), "nodal": float(kms[4])}}


def main():
    a = parse()
    if a.workload == "tangent":
        return bench_tangent(a)
    if a.workload == "residual":
        return bench_residual(a, laws=tuple(a.residual_laws.split(",")))
    ctx = Ctx(a)
    rank, world = ctx.rank, ctx.world
    if a.workload == "step2d":
        print(json.dumps(secondary_2d(ctx)), flush=True)
        return
    check = partition_check(ctx) if world > 1 and not a.no_partition_check else None
    if check is not None and (check["max_rel_err"] > PARTITION_TOL or not check["index_maps_equal"] or check["status_flags"]):
        raise SystemExit("bench: the partitioned run does not reproduce the single-solver run: %r" % check)
    rec, case, kms = run_config(ctx, a.scaling)
    other = "strong" if a.scaling == "weak" else "weak"
    rec_other, other_skipped = None, None
    if not a.no_second_scaling:
        if other == "strong" and int(round((a.particles_total / 8.0) ** (1.0 / 3.0))) < 8 * world:
            # (too few cell layers for this many ranks: the primary record stands, the second is left out and says why)
            other_skipped = "--particles-total %d gives fewer than 8 cell layers per rank at %d ranks" % (a.particles_total, world)
        else:
            rec_other, _, _ = run_config(ctx, other, with_kernels=world > 1)
    stirred = None
    second = None
    if world == 1 and not a.no_stirred:
        stirred = stirred_figure(ctx.nlps, ctx.synth, a, ctx.stream)
    if world == 1 and not a.no_secondary:
        second = secondary_2d(ctx)
    implicit = None
    if world == 1 and not a.no_implicit:  # the maintained driver's hot call at the same size (Neo-Hookean), short form
        r = bench_residual(a, stream=ctx.stream, laws=("nh",), cpu=False, emit=False)[0]
        implicit = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "fused_ms", "separate_stages_ms",
                                      "fused_over_separate", "fused_vs_separate_max_rel_diff", "roofline", "status_flags",
                                      "tangent")}
        implicit["workload"] = r["config"]["workload"]
    if rank == 0:
        npart = rec["particles_rank0"]
        names = ["search+activate", "lists+newton+p2g_mass_mom", "g2p_grad+stress+p2g_force", "g2p_update", "nodal"]
        kern = ["k_search", "k2_tile", "k3_tile", "k5_tile", None]
        alg = [0, BYTES_3D["S1"] + BYTES_3D["S2"], BYTES_3D["S3"] + BYTES_3D["S4"], BYTES_3D["S5"], 0]
        # PMC figures of the same command (rocprofv3 passes of tools/profile.sh, committed under profiles/): HBM bytes,
        # VALU instructions and LDS-array cycles per launch at 1 M particles; scaled by the particle count
        pmc = {}
        for fn in ("hbm_traffic.json", "sq_counters.json"):
            try:
                pmc.update(json.load(open(os.path.join(ROOT, "profiles", fn))))
            except Exception:
                pass
        per_kernel = {}
        scale = npart / 1.0e6
        for i in range(4):
            t_s = kms[i] * 1e-3
            if t_s <= 0:
                continue
            e = {"kernel_ms": float(kms[i]), "algorithmic_bytes_per_particle": alg[i],
                 "hbm_frac": npart * alg[i] / t_s / 1e9 / HBM_PEAK_GBS}
            tr = pmc.get(names[i])
            if tr is not None:  # counter bytes (FETCH_SIZE + WRITE_SIZE passes) / live kernel time / peak
                e["traffic_bytes"] = tr * scale
                e["traffic_frac"] = tr * scale / t_s / 1e9 / HBM_PEAK_GBS
            sq = pmc.get(kern[i] or "", {}) if isinstance(pmc.get(kern[i] or ""), dict) else {}
            if "valu_insts" in sq:  # SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x 2.4 GHz x kernel time)
                e["fp64_issue_frac"] = sq["valu_insts"] * scale * 4.0 / (1024 * 2.4e9 * t_s)
            if "lds_idx_active" in sq:  # SQ_LDS_IDX_ACTIVE (LDS-array cycles, summed over CUs) / (256 CUs x 2.4 GHz x time)
                e["lds_busy_frac"] = sq["lds_idx_active"] * scale / (256 * 2.4e9 * t_s)
            fr = {k: e[k] for k in ("hbm_frac", "fp64_issue_frac", "lds_busy_frac") if k in e}
            e["bound"] = max(fr, key=fr.get).replace("_frac", "").replace("_busy", "")
            per_kernel[names[i]] = e
        dom = int(np.argmax(kms[:4]))
        achieved = npart * alg[dom] / (kms[dom] * 1e-3) / 1e9 if kms[dom] > 0 else 0.0
        domk = per_kernel.get(names[dom], {})
        cz = rec["cells"]
        out = {
            "metric": "particle-steps/sec (P2G+stress+G2P)", "value": rec["value"],
            "unit": "particle-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": a.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "3-D elastic cube impact, %d particles/GPU (%dx%dx%d cells x 8 over %d rank(s)), LME gamma=3, "
                                   "Neo-Hookean E=1e7 nu=0.3, explicit predictor-corrector step, 1xMI355X per rank; "
                                   "z-slabs for N>1 (%s scaling)" % (npart, cz[0], cz[1], cz[2], world, a.scaling),
                       "particles_total": rec["particles_total"], "grid_nodes": rec["grid_nodes"],
                       "halo": rec["halo"], "halo_impl": rec["halo_impl"],
                       "halo_overlap": bool(rec["halo_overlap_mode"]) if world > 1 else None,
                       "halo_overlap_mode": rec["halo_overlap_mode"], "rccl_nranks": rec["rccl_nranks"],
                       "overlap_forms_tried": rec["overlap_forms_tried"],
                       "resorts_in_timed_region": 1, "library_default_resort_interval": 50},
            "roofline": {"bound": "hbm", "limiter": domk.get("bound", "hbm"), "kernel": names[dom], "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": domk.get("traffic_bytes"), "traffic_frac": domk.get("traffic_frac"),
                         "algorithmic_bytes_per_particle": alg[dom], "kernel_ms": float(kms[dom]),
                         "event_overhead_ms": float(kms[5]),
                         "hbm_frac": domk.get("hbm_frac"), "fp64_issue_frac": domk.get("fp64_issue_frac"),
                         "lds_busy_frac": domk.get("lds_busy_frac"),
                         "note": "achieved/peak/frac are the HBM roofline of the dominant kernel (algorithmic bytes / "
                                 "live kernel time); traffic_frac is the same with the HBM bytes the counters saw "
                                 "(profiles/hbm_traffic.json); bound = hbm says which roofline achieved / peak are on; "
                                 "limiter names the LARGEST of the three fractions: hbm, "
                                 "fp64_issue (SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x 2.4 GHz x time)) and lds "
                                 "(SQ_LDS_IDX_ACTIVE / (256 CUs x 2.4 GHz x time)); the SQ counts come from "
                                 "profiles/sq_counters.json (rocprofv3 PMC passes of this command)"},
            "kernel_ms_all": {names[i]: float(kms[i]) for i in range(5)},
            "exchange_wait_ms": float(kms[6]),
            "per_kernel": per_kernel,
            a.scaling: {k: v for k, v in rec.items()},
        }
        if rec_other is not None:
            out[other] = rec_other
        elif other_skipped:
            out[other] = {"skipped": other_skipped}
        if check is not None:
            out["partition_check"] = check
        if implicit is not None:
            out["implicit"] = implicit
        if stirred is not None:
            out["stirred_ms_per_step"] = stirred["ms_per_step"]
            out["stirred"] = stirred
        if second is not None:
            out["secondary"] = second
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_cells)
        print(json.dumps(out))
    if world > 1:
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
