"""Full-size cases of BASELINE.json that the oracle cannot run whole: size-independent properties plus a SAMPLED oracle
re-computation — the per-particle stress update (Constitutive.c:18-258) is independent given (DF, F_n+1, b_e,n, kappa_n,
eps_n), so the oracle re-does it for a few thousand random particles of the full-size run and must agree with what
the fused step left behind."""
import numpy as np
import pytest

from util import DP, assert_close, dirichlet_plane, gpu_setup, make_case, nlps, orc
from window_oracle import window_residual_check, window_step_check

pytestmark = pytest.mark.gpu

STATE_N = ["b_e_n", "Kappa_n", "EPS_n"]
STATE_N1 = ["F_n", "DF", "J_n", "Stress", "W", "b_e_n", "Kappa_n", "EPS_n", "rho"]


def oracle_stress_of_sample(case, idx, before, after):
    """Oracle constitutive update of the sampled particles from the downloaded kinematics of the last step."""
    o = orc()
    cloud = {k: (v[idx].copy() if isinstance(v, np.ndarray) and v.shape[:1] == case["cloud"]["x"].shape[:1] else v)
             for k, v in case["cloud"].items()}
    P = o.OracleParticles(cloud)
    P["F_n1"][:] = after["F_n"][idx]      # the corrector rolled F_n <- F_n+1
    P["DF"][:] = after["DF"][idx]
    P["J_n1"][:] = after["J_n"][idx]
    P["b_e_n"][:] = before["b_e_n"][idx]
    P["kappa_n"][:] = before["Kappa_n"][idx]
    P["eps_n"][:] = before["EPS_n"][idx]
    prm = o.default_params()
    assert o.constitutive(P, o.make_materials(case["materials"]), prm) == 0
    return P


def check_sample(case, idx, before, after, tol=1e-10):
    P = oracle_stress_of_sample(case, idx, before, after)
    E = max(m["E"] for m in case["materials"])
    assert_close(after["Stress"][idx], P["stress"], tol, "sampled Kirchhoff stress", scale=E * 1e-6)
    assert_close(after["W"][idx], P["W"], tol, "sampled energy", scale=E * 1e-4)
    plastic = np.array([case["materials"][m]["type"] == 2 for m in case["cloud"]["matidx"][idx]])
    if plastic.any():
        assert_close(after["b_e_n"][idx][plastic], P["b_e_n1"][plastic], tol, "sampled b_e")
        assert_close(after["Kappa_n"][idx][plastic], P["kappa_n1"][plastic], tol, "sampled kappa")
        assert_close(after["EPS_n"][idx][plastic], P["eps_n1"][plastic], tol, "sampled eps", scale=1e-6)
    return P


def nodal_properties(S, mass0, ndim=3):
    nod = S.explicit_nodal()
    assert abs(nod["mass"].reshape(-1, ndim)[:, 0].sum() / mass0 - 1.0) < 1e-12  # partition of unity
    f = nod["force"].reshape(-1, ndim).sum(0)  # Newton's third law: internal forces sum to zero
    assert np.all(np.abs(f) <= 1e-9 * np.abs(nod["force"]).max() * np.sqrt(nod["force"].size) + 1e-9)
    assert np.isfinite(nod["accel"]).all()
    return nod


def test_config5_drucker_prager_8m_full_size():
    """BASELINE configs[4] at its full size on ONE GPU: 8 M Drucker-Prager particles (100^3 cells x 8) under gravity,
    kernels compiled for the plastic law (k3_tile<3,2,1>)."""
    n = nlps()
    case = make_case(3, [110, 110, 110], [5, 5, 5], [100, 100, 100], material=DP)
    S = gpu_setup(case, nsteps=8)
    assert S.np == 8_000_000
    gb = n.BccSet([dirichlet_plane(case, 2, 5, 8)])
    dt = 0.1 / np.sqrt(DP["E"] / 1000.0)
    grav = [0.0, 0.0, -9.81]
    for t in range(5):
        S.explicit_step(gb, t, dt, 0.5, grav)
    before = S.download_state(STATE_N)
    # step 5 with VALUES under it: two blocks of 26^3 cells rebuilt in the oracle from the downloaded pre-step state --
    # one in the interior, one on the floor (Dirichlet plane, where the column yields first) -- particle fields, index
    # maps and nodal sums of everything the cut cannot have reached (tests/window_oracle.py)
    deep = window_step_check(S, case, gb, [dirichlet_plane(case, 2, 5, 8)], 5, dt, 0.5, grav,
                             [([40, 40, 40], [26, 26, 26]), ([42, 38, 1], [26, 26, 26])], [5, 5, 5], [105, 105, 105], 8,
                             with_lists=False, label="8 M Drucker-Prager")
    assert min(deep) >= 1000
    nodal_properties(S, case["cloud"]["mass"].sum())
    after = S.download_state(STATE_N1)
    assert np.all(after["J_n"] > 0) and np.isfinite(after["Stress"]).all()
    assert np.all(after["Kappa_n"] >= DP["kappa_0"] - 1e-9) and np.all(after["EPS_n"] >= 0)
    be = after["b_e_n"]
    assert np.allclose(be[:, [1, 2, 5]], be[:, [3, 6, 7]], atol=1e-9)
    yielding = after["EPS_n"] > before["EPS_n"]
    assert yielding.any(), "the column must yield somewhere (plastic return mapping exercised)"
    rng = np.random.default_rng(5)
    idx = np.unique(np.concatenate([rng.integers(0, S.np, 3000), np.flatnonzero(yielding)[:1500]]))
    P = check_sample(case, idx, before, after)
    assert np.count_nonzero(P["eps_n1"] > P["eps_n"]) > 0


def test_mixed_laws_1m_full_size():
    """1 M particles with three laws interleaved (Neo-Hookean / Hencky / Drucker-Prager by MatIdx): the explicit step
    of a mixed cloud at BASELINE configs[1] size."""
    n = nlps()
    soft_nh = {"type": 0, "E": 2.0e4, "nu": 0.3}
    soft_hencky = {"type": 1, "E": 1.0e4, "nu": 0.25}
    case = make_case(3, [60, 60, 60], [5, 5, 5], [50, 50, 50], material=DP, velocity=[0.0, 0.0, -1.0])
    case["materials"] = [soft_nh, soft_hencky, DP]
    npart = case["cloud"]["x"].shape[0]
    assert npart == 1_000_000
    case["cloud"]["matidx"] = (np.arange(npart) % 3).astype(np.int32)
    S = gpu_setup(case, nsteps=8)
    gb = n.BccSet([dirichlet_plane(case, 2, 5, 8)])
    dt = 0.1 / np.sqrt(2.0e4 / 1000.0)
    grav = [0.0, 0.0, -9.81]
    for t in range(5):
        S.explicit_step(gb, t, dt, 0.5, grav)
    before = S.download_state(STATE_N)
    S.explicit_step(gb, 5, dt, 0.5, grav)
    assert S.status_flags() == 0
    nodal_properties(S, case["cloud"]["mass"].sum())
    after = S.download_state(STATE_N1)
    assert np.all(after["J_n"] > 0) and np.isfinite(after["Stress"]).all()
    rng = np.random.default_rng(6)
    idx = np.unique(rng.integers(0, npart, 6000))
    P = check_sample(case, idx, before, after)
    for m in range(3):
        assert np.abs(P["stress"][case["cloud"]["matidx"][idx] == m]).max() > 0.1


def test_window_oracle_1m_neo_hookean():
    """BASELINE configs[1] (1 M particles, the bench workload) twelve steps into its fall: the thirteenth step against
    the oracle on three windows -- interior, free top surface, floor with its Dirichlet plane.  Everything that only
    exists at this size sits under these values: 2 197 tiles over several rounds of workgroups, the XCD-compacted work
    lists, canonical lists from the per-node counters, the search riding on K5, the folded step."""
    n = nlps()
    case = make_case(3, [60, 60, 60], [5, 5, 5], [50, 50, 50], velocity=[0.0, 0.0, -10.0])
    nsteps = 14
    S = gpu_setup(case, nsteps=nsteps)
    assert S.np == 1_000_000
    bc = dirichlet_plane(case, 2, 5, nsteps)
    gb = n.BccSet([bc])
    dt = 1e-3
    for t in range(12):
        S.explicit_step(gb, t, dt, 0.5, None)
    blocks = [([17, 17, 17], [26, 26, 26]), ([15, 19, 33], [26, 26, 26]), ([19, 15, 1], [26, 26, 26])]
    deep = window_step_check(S, case, gb, [bc], 12, dt, 0.5, None, blocks, [5, 5, 5], [55, 55, 55], nsteps, label="1 M")
    assert min(deep) >= 1000, deep
    # ... and the implicit driver's residual call (MODE 3 of the same kernel) on the state these 13 steps left: the search,
    # the internal forces at the nodes and the particle state it leaves, on the interior and the floor window
    nodes = window_residual_check(S, case, gb, [bc], 13, nsteps, [blocks[0], blocks[2]], [5, 5, 5], [55, 55, 55], label="1 M")
    assert min(nodes) >= 500, nodes
    S.close()


def test_window_oracle_8m_neo_hookean():
    """BASELINE configs[3]'s total size on one GPU (8 M particles: the unfolded step above 2 M particles, 17 576 tiles),
    third step, two windows: interior and the corner where three free faces meet."""
    n = nlps()
    case = make_case(3, [110, 110, 110], [5, 5, 5], [100, 100, 100], velocity=[0.0, 0.0, -10.0])
    nsteps = 4
    S = gpu_setup(case, nsteps=nsteps)
    assert S.np == 8_000_000
    bc = dirichlet_plane(case, 2, 5, nsteps)
    gb = n.BccSet([bc])
    for t in range(2):
        S.explicit_step(gb, t, 1e-3, 0.5, None)
    blocks = [([42, 42, 42], [26, 26, 26]), ([83, 83, 83], [26, 26, 26])]
    deep = window_step_check(S, case, gb, [bc], 2, 1e-3, 0.5, None, blocks, [5, 5, 5], [105, 105, 105], nsteps,
                             with_lists=False, label="8 M")
    assert min(deep) >= 1000, deep
    S.close()


def test_window_oracle_on_a_grid_of_more_than_65536_tiles():
    """The tile scan has a second form for grids of 65 536 tiles and more (commit f578b41): 1 M particles inside a grid
    of 260^3 nodes = 274 625 tiles, fourth step, one interior window."""
    n = nlps()
    case = make_case(3, [259, 259, 259], [100, 100, 100], [50, 50, 50], velocity=[3.0, -2.0, -10.0])
    x = case["cloud"]["x"]
    c = x.mean(axis=0)  # (a uniform translation leaves F = 1 and a stress of pure rounding: shear and squeeze the block)
    case["cloud"]["vel"] = case["cloud"]["vel"] + np.stack([4.0 * (x[:, 2] - c[2]), -3.0 * (x[:, 0] - c[0]),
                                                            -2.0 * (x[:, 2] - c[2])], axis=1) / 25.0
    nsteps = 5
    S = gpu_setup(case, nsteps=nsteps)
    gb = n.BccSet([])
    for t in range(3):
        S.explicit_step(gb, t, 1e-3, 0.5, None)
    deep = window_step_check(S, case, gb, [], 3, 1e-3, 0.5, None, [([112, 112, 112], [26, 26, 26])], [100, 100, 100],
                             [150, 150, 150], nsteps, with_lists=False, label="274 625 tiles")
    assert min(deep) >= 1000, deep
    S.close()
