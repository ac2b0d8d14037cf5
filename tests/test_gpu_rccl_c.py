"""The ghost-layer exchange over RCCL as the LIBRARY does it (nlps_gpu_rccl_attach: ncclCommInitRank, ncclSend /
ncclRecv on a library-owned stream, add / max kernels, events), without Python in the data path.  A one-GPU box holds one
rank only, so the wire is exercised with the rank as its own neighbour (nlps_gpu_rccl_selftest_exchange); the world-2 / 3
partition logic is covered by tests/test_multirank_gloo.py (CPU) and test_multirank_on_one_gpu (callback double)."""
import numpy as np
import pytest

from util import assert_close, dirichlet_plane, gpu_setup, make_case, nlps

pytestmark = pytest.mark.gpu


def test_rccl_owned_by_the_library_world_1():
    import torch
    n = nlps()
    case = make_case(3, [11, 10, 12], [3, 3, 3], [5, 4, 6], velocity=[0.0, 0.0, -10.0])
    gn = case["grid_n"]
    nl, plane = gn[2], gn[0] * gn[1]
    S = gpu_setup(case, init=False, nsteps=4)
    uid = n.Solver.rccl_unique_id()
    assert len(uid) == 128 and any(uid)
    S.rccl_attach(uid, 0, 1, [0], [nl - 1], mode=0)
    with pytest.raises(n.NlpsError):
        S.rccl_attach(uid, 0, 1, [0], [nl - 1])  # one communicator per handle
    # the wire: lowest three layers <-> highest three layers through ncclSend / ncclRecv to itself
    rng = np.random.default_rng(0)
    for overlap in (False, True):
        for nfield, elem, kind in ((4, 8, 0), (3, 8, 0), (1, 1, 1)):
            if elem == 8:
                a = torch.tensor(rng.normal(size=nl * plane * nfield), dtype=torch.float64, device="cuda")
            else:
                a = torch.tensor(rng.integers(0, 2, size=nl * plane), dtype=torch.uint8, device="cuda")
            ref = a.clone()
            torch.cuda.synchronize()
            S.rccl_selftest_exchange(a.data_ptr(), nfield, elem, kind, overlap)
            S.synchronize()
            torch.cuda.synchronize()
            m = 3 * plane * nfield
            lo_old, hi_old = ref[:m].clone(), ref[-m:].clone()
            if kind == 0:
                ref[:m] += hi_old
                ref[-m:] += lo_old
            else:
                ref[:m] = torch.maximum(lo_old, hi_old)
                ref[-m:] = torch.maximum(lo_old, hi_old)
            assert torch.equal(a, ref), (overlap, nfield, elem, kind)
    # explicit steps with the communicator attached (no neighbours: nothing to exchange) equal a plain run
    S.initialise_shapefun()
    S2 = gpu_setup(case, nsteps=4)
    gb = n.BccSet([dirichlet_plane(case, 2, 2, 4)])
    for t in range(3):
        S.explicit_step(gb, t, 1e-3)
        S2.explicit_step(gb, t, 1e-3)
    a, b = S.download_state(), S2.download_state()
    assert np.array_equal(a["I0"], b["I0"])
    for k in ("x", "vel", "Stress", "F_n"):
        assert_close(a[k], b[k], 1e-12, f"{k}: RCCL attached vs plain")
    # reduce of a masked vector (implicit driver): world 1 = identity
    v = torch.arange(10, dtype=torch.float64, device="cuda")
    S.rccl_reduce(v.data_ptr(), 10, root=0)
    S.rccl_reduce(v.data_ptr(), 10, root=-1)
    S.synchronize()
    assert torch.equal(v.cpu(), torch.arange(10, dtype=torch.float64))
    S.rccl_detach()
    S.rccl_detach()  # idempotent
    assert S.status_flags() == 0


def test_rccl_all_reduce_mode_world_1():
    n = nlps()
    case = make_case(3, [11, 10, 9], [3, 3, 2], [5, 4, 4], velocity=[0.0, 0.0, -10.0])
    S = gpu_setup(case, init=False, nsteps=2)
    S.rccl_attach(n.Solver.rccl_unique_id(), 0, 1, [0], [case["grid_n"][2] - 1], mode=1)
    S.initialise_shapefun()
    gb = n.BccSet([dirichlet_plane(case, 2, 2, 2)])
    S.explicit_step(gb, 0, 1e-3)
    nod = S.explicit_nodal()
    assert abs(nod["mass"].reshape(-1, 3)[:, 0].sum() / case["cloud"]["mass"].sum() - 1) < 1e-12
    S.close()


@pytest.mark.parametrize("world,overlap", [(2, 1), (2, 0)])
def test_partitioned_vs_whole_over_rccl(world, overlap):
    """Boxes with one GPU per rank only (skipped on the one-GPU test box): tests/mr_gpu_worker.py with backend nccl --
    every rank on its own card, the ghost layers exchanged by the library's own RCCL path, partitioned result against
    one solver holding the whole cloud (index maps bit for bit, fields to 1e-11)."""
    import os
    import subprocess
    import sys
    import torch
    from util import free_port
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs (one per rank)" % world)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(root, "tests", "mr_gpu_worker.py")]
    env = dict(os.environ, NLPS_OVERLAP=str(overlap), NLPS_NDIM="3", NLPS_MR_BACKEND="nccl")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0 and "MULTIRANK_GPU_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.parametrize("mode", [1, 2])
def test_overlap_choreographies_world_1(mode):
    """The two overlapped forms of the explicit step with the library's exchange attached (no neighbour, so nothing moves,
    but every launch split / signal / stream wait runs): 1 = boundary and interior tiles as separate launches, 2 = ONE
    launch per stage whose boundary tiles release the exchange stream through signal memory (hipStreamWaitValue32).
    Ghost bands inside the cloud make both tile classes non-empty; results must equal a plain run."""
    n = nlps()
    case = make_case(3, [14, 13, 24], [3, 3, 3], [8, 7, 18], velocity=[0.0, 0.0, -10.0])
    nl = case["grid_n"][2]
    S = gpu_setup(case, init=False, nsteps=5)
    S.rccl_attach(n.Solver.rccl_unique_id(), 0, 1, [0], [nl - 1], mode=0)
    lo, hi = S.touched_layers()
    S.set_ghost_bands(lo + 5, hi - 5, mode)
    S.set_resort_interval(2)
    S.initialise_shapefun()
    P = gpu_setup(case, nsteps=5)
    P.set_resort_interval(2)
    gb = n.BccSet([dirichlet_plane(case, 2, 2, 5)])
    for t in range(5):
        S.explicit_step(gb, t, 1e-3)
        P.explicit_step(gb, t, 1e-3)
    a, b = S.download_state(), P.download_state()
    assert S.status_flags() == 0
    assert np.array_equal(a["I0"], b["I0"])
    for k in ("x", "vel", "Stress", "F_n"):
        assert_close(a[k], b[k], 1e-12, f"{k}: overlap mode {mode} vs plain")
    na, nb = S.explicit_nodal(), P.explicit_nodal()
    for k in ("mass", "force", "accel"):
        assert_close(na[k], nb[k], 1e-11, f"nodal {k}: overlap mode {mode} vs plain")


def test_rccl_migrate_world_1_self_loop():
    """nlps_gpu_rccl_migrate = select -> counts and rows over ncclSend / ncclRecv on the library's communicator ->
    commit, with no Python in between.  One-GPU box: the rank is its own two neighbours (nlps_gpu_rccl_selftest_migrate),
    so the particles that leave the keep range below AND above come straight back over the wire; stepping on must match a
    solver that never migrated, particle by particle (matched by global id).  The plain entry on world 1 has no neighbour
    and moves nothing."""
    n = nlps()
    case = make_case(3, [11, 10, 22], [3, 3, 3], [5, 4, 16], velocity=[1.0, 0.5, -10.0])
    nsteps = 6
    nl = case["grid_n"][2]
    A = gpu_setup(case, init=False, nsteps=nsteps)
    A.rccl_attach(n.Solver.rccl_unique_id(), 0, 1, [0], [nl - 1], mode=0)
    assert A.rccl_info() == (1, 0, 0)  # ncclCommCount, ncclCommUserRank, overlap mode of a world without neighbours
    A.initialise_shapefun()
    B = gpu_setup(case, nsteps=nsteps)
    npart = case["cloud"]["x"].shape[0]
    ids = (np.arange(npart) * 3 + 1).astype(np.int32)
    A.set_particle_ids(ids)
    B.set_particle_ids(ids)
    gb = n.BccSet([dirichlet_plane(case, 2, 2, nsteps)])
    z = case["cloud"]["x"][:, 2]
    keep_lo, keep_hi = int(z.min()) + 2, int(z.max()) - 2  # (the immigrants must fit the capacity reserved at create)
    moved = 0
    for t in range(nsteps):
        if t in (2, 4):
            d, u, g = A.rccl_migrate(keep_lo, keep_hi, selftest=True)
            assert d > 0 and u > 0 and g == d + u and A.num_particles() == npart
            moved += g
            assert A.rccl_migrate(keep_lo, keep_hi) == (0, 0, 0)  # world 1, no self-loop: the range is clamped, nobody leaves
        A.explicit_step(gb, t, 2e-3)
        B.explicit_step(gb, t, 2e-3)
    assert moved > 0 and A.status_flags() == 0
    a, b = A.download_state(), B.download_state()
    ia, ib = A.download_ids(), B.download_ids()
    assert np.array_equal(ia, np.sort(ids))
    order_b = np.argsort(ib)
    assert np.array_equal(a["I0"], b["I0"][order_b])
    for k in ("x", "vel", "Stress", "F_n", "lambda"):
        assert_close(a[k], b[k][order_b], 1e-12 if k != "lambda" else 1e-9, f"{k} after migration over the wire")
    A.close()
    B.close()
