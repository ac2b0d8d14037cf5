"""Worker of tests/test_gpu_parity.py::test_migration_between_ranks_on_one_gpu (launched by torch.distributed.run).

A soft block thrown along the slab axis crosses from one rank's slab into the next ones.  Every rank drives the HIP
library on the SAME card for the particles it currently owns, with the ghost-node exchange and, every 4 steps, the
particle migration of nl-partsol_amd/halo.py (gloo + host staging stands in for RCCL, which needs one GPU per
rank).  Rank 0 also runs the whole cloud in one solver; matched by global id, the partitioned result has to agree:
closest nodes and neighbour counts bit for bit, fields to 1e-10."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CELLS, MARGIN, NSTEPS, EVERY = 10, 5, 60, 4


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    import torch.distributed as dist
    import util
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nlps = util.nlps()
    synth = util.synth
    halo_mod = importlib.import_module("nl-partsol_amd.halo")
    soft = {"type": 0, "E": 1.0e5, "nu": 0.3}  # celerity 10
    gc = [CELLS + 2 * MARGIN, CELLS + 2 * MARGIN, CELLS * world + 2 * MARGIN]
    gn = synth.grid_nodes(gc)
    # the whole cloud: a block straddling the first interface, flying up at Mach 1 with some shear
    z0 = MARGIN + CELLS - 5
    whole = synth.make_cloud(3, gc, [MARGIN + 1, MARGIN + 1, z0], [5, 5, 6], h=1.0, jitter=0.05, seed=31,
                             velocity=[0.5, -0.3, 10.0])
    x = whole["x"]
    whole["vel"][:, 2] *= 1.0 + 0.15 * (x[:, 0] - x[:, 0].mean()) / 2.5
    npart = x.shape[0]
    ids = np.arange(npart, dtype=np.int32)
    bounds = [MARGIN + r * CELLS for r in range(world)] + [gn[2]]   # rank r starts owning cells [bounds[r], bounds[r+1])
    cell = np.floor(x[:, 2]).astype(int)
    mine = (cell >= bounds[rank]) & (cell < bounds[rank + 1])
    cloud = {k: (v[mine].copy() if isinstance(v, np.ndarray) and v.shape[:1] == (npart,) else v) for k, v in whole.items()}
    bc = {"nodes": synth.plane_nodes(gn, 2, 2), "dim": 3, "dir": np.ones((3, NSTEPS), dtype=np.int32),
          "value": np.zeros((3, NSTEPS))}
    dt = 0.25 / 20.0
    work_stream = torch.cuda.Stream()
    torch.cuda.set_stream(work_stream)
    S = nlps.Solver(3, gn, [0.0] * 3, 1.0, cloud, [soft], nsteps=NSTEPS, stream=work_stream.cuda_stream)
    S.set_particle_ids(ids[mine])
    lo, hi = halo_mod.SlabHalo.layer_ranges(world, CELLS, MARGIN, gn[2], reach=4)
    halo = halo_mod.SlabHalo(torch, dist, rank, world, gn[0] * gn[1], gn[2], lo, hi)
    nnodes = gn[0] * gn[1] * gn[2]
    S.set_halo_exchange(lambda dptr, nfield, elem, kind, phase: halo.exchange_ptr(dptr, nnodes * nfield, nfield, elem,
                                                                                 kind, phase))
    S.set_node_window(lo[rank], hi[rank])
    band_lo, band_hi = halo.ghost_bands(rank)
    S.set_ghost_bands(band_lo, band_hi, True)
    S.set_resort_interval(5)
    keep_lo = bounds[rank] - 1 if rank > 0 else 0
    keep_hi = bounds[rank + 1] + 1 if rank + 1 < world else gn[2] - 1
    S.initialise_shapefun()
    gb = nlps.BccSet([bc])
    moved = 0
    for t in range(NSTEPS):
        if t % EVERY == 0:
            d, u, r = halo.migrate(S, keep_lo, keep_hi)
            moved += d + u
            if os.environ.get("NLPS_DEBUG"):
                print("rank %d step %d: down %d up %d received %d now %d (keep %d..%d)" % (rank, t, d, u, r, S.num_particles(), keep_lo, keep_hi), flush=True)
        S.explicit_step(gb, t, dt)
    assert S.status_flags() == 0, "rank %d: flags %x" % (rank, S.status_flags())
    st = S.download_state()
    nn, _ = S.download_lists()
    mine_out = {k: st[k] for k in ("x", "vel", "Stress", "F_n", "I0", "lambda")}
    mine_out["nn"] = nn
    mine_out["ids"] = S.download_ids()
    mine_out["moved"] = moved
    parts = [None] * world
    dist.gather_object(mine_out, parts if rank == 0 else None, dst=0)
    if rank == 0:
        G = nlps.Solver(3, gn, [0.0] * 3, 1.0, whole, [soft], nsteps=NSTEPS)
        G.set_resort_interval(5)
        G.initialise_shapefun()
        for t in range(NSTEPS):
            G.explicit_step(gb, t, dt)
        assert G.status_flags() == 0
        ref = G.download_state()
        rnn, _ = G.download_lists()
        allid = np.concatenate([p["ids"] for p in parts])
        assert np.array_equal(np.sort(allid), ids), "every particle is owned by exactly one rank"
        for p in parts:
            assert np.all(np.diff(p["ids"]) > 0), "downloads come back in ascending global id"
        counts = [len(p["ids"]) for p in parts]
        assert sum(p["moved"] for p in parts) >= npart // 2, "the block has to change owner"
        assert counts[0] < npart // 4, "most particles must have left rank 0 (%s)" % counts
        for k in ("I0",):
            assert np.array_equal(np.concatenate([p[k] for p in parts]), ref[k][allid]), k
        assert np.array_equal(np.concatenate([p["nn"] for p in parts]), rnn[allid]), "NumberNodes"
        for k in ("x", "vel", "Stress", "F_n", "lambda"):
            got = np.concatenate([p[k] for p in parts])
            util.assert_close(got, ref[k][allid], 1e-10 if k != "lambda" else 1e-9, "%s partitioned+migrated vs whole" % k)
        print("MIGRATION_GPU_OK world=%d particles=%d owners=%s" % (world, npart, counts))
    dist.barrier()
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
