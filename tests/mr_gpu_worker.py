"""Worker of tests/test_gpu_parity.py::test_multirank_on_one_gpu (launched by torch.distributed.run).

Every rank drives the HIP library on the SAME card for its slab of a z-stacked cloud, with the ghost-node
exchange of nl-partsol_amd/halo.py behind the C-ABI halo callback (gloo + host staging, because RCCL needs
one GPU per rank) and the node window bench.py sets for N > 1.  Rank 0 also runs the whole cloud in one
solver; the partitioned result has to match it: index maps bit for bit, fields to 1e-11."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ND = int(os.environ.get("NLPS_NDIM", "3"))
CELLS, MARGIN, NSTEPS = (8 if ND == 3 else 24), 5, 4


def rank_cloud(synth, rank, world):
    gc = [CELLS + 2 * MARGIN] * (ND - 1) + [CELLS * world + 2 * MARGIN]
    lo = [MARGIN] * (ND - 1) + [MARGIN + CELLS * rank]
    return gc, synth.make_cloud(ND, gc, lo, [CELLS] * ND, h=1.0, jitter=0.05, seed=777 + rank,
                                velocity=[1.0, 0.5, -10.0][3 - ND:])


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    import torch.distributed as dist
    import util
    # NLPS_MR_BACKEND=nccl (boxes with one GPU per rank): RCCL driven by the library itself (nlps_gpu_rccl_attach)
    rccl = os.environ.get("NLPS_MR_BACKEND", "gloo") == "nccl"
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) if rccl else 0)
    if rccl:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", torch.cuda.current_device()))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    nlps = util.nlps()
    synth = util.synth
    halo_mod = importlib.import_module("nl-partsol_amd.halo")
    gc, cloud = rank_cloud(synth, rank, world)
    gn = synth.grid_nodes(gc)
    mats = [util.NH]
    bc = {"nodes": synth.plane_nodes(gn, ND - 1, MARGIN + 1), "dim": ND, "dir": np.ones((ND, NSTEPS), dtype=np.int32),
          "value": np.zeros((ND, NSTEPS))}
    dt = 0.4 / 100.0
    work_stream = torch.cuda.Stream()  # as bench.py: library kernels and callback ops share one real stream
    torch.cuda.set_stream(work_stream)
    stream = work_stream.cuda_stream
    S = nlps.Solver(ND, gn, [0.0] * ND, 1.0, cloud, mats, nsteps=NSTEPS, stream=stream)
    nnodes = int(np.prod(gn))
    lo, hi = halo_mod.SlabHalo.layer_ranges(world, CELLS, MARGIN, gn[ND - 1], reach=3)
    halo = halo_mod.SlabHalo(torch, dist, rank, world, nnodes // gn[ND - 1], gn[ND - 1], lo, hi)
    overlap = os.environ.get("NLPS_OVERLAP", "1") == "1"
    band_lo, band_hi = halo.ghost_bands(rank)
    if rccl:
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            uid = torch.tensor(list(nlps.Solver.rccl_unique_id()), dtype=torch.uint8, device="cuda")
        dist.broadcast(uid, 0)
        S.rccl_attach(bytes(uid.cpu().tolist()), rank, world, lo, hi, mode=0)  # sets the node window and the bands
        if not overlap:
            S.set_ghost_bands(band_lo, band_hi, False)
    else:
        S.set_halo_exchange(lambda dptr, nfield, elem, kind, phase: halo.exchange_ptr(dptr, nnodes * nfield, nfield, elem,
                                                                                     kind, phase))
        S.set_node_window(lo[rank], hi[rank])
        S.set_ghost_bands(band_lo, band_hi, overlap)  # exchanges behind the interior tiles (two-phase callback)
    S.set_resort_interval(2)
    S.initialise_shapefun()
    gb = nlps.BccSet([bc])
    for t in range(NSTEPS):
        S.explicit_step(gb, t, dt)
    assert S.status_flags() == 0
    st = S.download_state()
    nn, lst = S.download_lists()
    mine = {k: st[k] for k in ("x", "vel", "acc", "Stress", "F_n", "I0", "lambda")}
    mine["nn"] = nn
    mine["active"] = S.download_active()

    # The implicit driver's residual on the partitioned cloud (nlps_gpu_lagrangian_evaluation, SURVEY 8e: "each residual
    # evaluation needs the S4 reduction"): every rank evaluates it for ITS particles, the force scatter ends in the
    # exchange of the shared layers, so a rank holds the complete residual of every node it has active -- compared node
    # by node (through the two Nodes2Mask) with the whole cloud's below.  dU is a function of the node alone.
    def residual_of(solver):
        solver.local_search()
        n2m, d2m = solver.active_masks(gb, NSTEPS - 1)
        na = solver.nactive
        ids = np.flatnonzero(n2m >= 0)
        ijk = np.stack([(ids // int(np.prod(gn[:a]))) % gn[a] for a in range(ND)], axis=1).astype(np.float64)
        dU = np.zeros((na, ND))
        dU[n2m[ids]] = 1e-3 * np.sin(0.7 * ijk + np.arange(ND)[None, :])
        Mv = solver.compute_nodal_lumped_mass()
        V, A = solver.get_nodal_field_n(Mv)
        a1, a2, a3 = 4.0e4, 4.0e2, 1.0
        R = solver.lagrangian_evaluation(dU.ravel(), V, A, Mv, [a1, a2, a3, 0.0, 0.0, 0.0], [0.0] * (ND - 1) + [-9.81])
        full = np.full((nnodes, ND), np.nan)
        full[ids] = R.reshape(-1, ND)[n2m[ids]]
        return full, solver.download_state(["F_n1", "Stress", "J_n1"])

    mine["residual"], rst = residual_of(S)
    mine["res_state"] = {k: rst[k] for k in ("F_n1", "Stress", "J_n1")}
    parts = [None] * world
    dist.gather_object(mine, parts if rank == 0 else None, dst=0)
    ok = True
    if rank == 0:
        clouds = [rank_cloud(synth, r, world)[1] for r in range(world)]
        whole = {}
        for k, v in clouds[0].items():
            whole[k] = np.concatenate([c[k] for c in clouds]) if isinstance(v, np.ndarray) else v
        G = nlps.Solver(ND, gn, [0.0] * ND, 1.0, whole, mats, nsteps=NSTEPS)
        G.set_resort_interval(2)
        G.initialise_shapefun()
        for t in range(NSTEPS):
            G.explicit_step(gb, t, dt)
        assert G.status_flags() == 0
        ref = G.download_state()
        rnn, _ = G.download_lists()
        for k in ("I0",):
            got = np.concatenate([p[k] for p in parts])
            assert np.array_equal(got, ref[k]), k
        assert np.array_equal(np.concatenate([p["nn"] for p in parts]), rnn), "NumberNodes"
        act = np.zeros_like(parts[0]["active"])
        for p in parts:
            act |= p["active"]
        assert np.array_equal(act, G.download_active()), "ActiveNode (union over ranks)"
        for k in ("x", "vel", "acc", "Stress", "F_n", "lambda"):
            got = np.concatenate([p[k] for p in parts])
            util.assert_close(got, ref[k], 1e-11 if k != "lambda" else 1e-9, "%s partitioned vs whole" % k)
        assert np.abs(ref["Stress"]).max() > 1.0, "the case must deform"
        Rw, wst = residual_of(G)
        scale = np.nanmax(np.abs(Rw))
        seen = np.zeros(nnodes, dtype=bool)
        for r_, p_ in enumerate(parts):
            have = ~np.isnan(p_["residual"][:, 0])
            assert not np.isnan(Rw[have]).any(), "rank %d has a node active that the whole cloud has not" % r_
            err = np.abs(p_["residual"][have] - Rw[have]).max() / scale
            assert err <= 1e-10, "residual of rank %d vs whole cloud: %.3e" % (r_, err)
            seen |= have
        assert np.array_equal(seen, ~np.isnan(Rw[:, 0])), "every active node of the whole cloud is active on some rank"
        for k in ("F_n1", "Stress", "J_n1"):
            got = np.concatenate([p_["res_state"][k] for p_ in parts])
            util.assert_close(got, wst[k], 1e-10, "%s after the residual call, partitioned vs whole" % k)
        print("MULTIRANK_GPU_OK ndim=%d world=%d particles=%d overlap=%s" % (ND, world, ref["x"].shape[0], overlap))
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
