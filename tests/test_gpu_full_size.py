"""Full-size cases of BASELINE.json that the oracle cannot run whole: size-independent properties plus a SAMPLED oracle
re-computation — the per-particle stress update (Constitutive.c:18-258) is independent given (DF, F_n+1, b_e,n, kappa_n,
eps_n), so the oracle re-does it for a few thousand random particles of the full-size run and must agree with what
the fused step left behind."""
import numpy as np
import pytest

from util import DP, assert_close, dirichlet_plane, gpu_setup, make_case, nlps, orc

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
    S.explicit_step(gb, 5, dt, 0.5, grav)
    assert S.status_flags() == 0
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
