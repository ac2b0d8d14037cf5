"""Eigensoftening: the damage hook of the maintained implicit driver's force stage with
Driver_EigenSoftening (Constitutive/Fracture/EigenSoftening.c:27-163 behind compute_damage__Constitutive__,
Constitutive.c:412-432; U-Newmark-beta.c:213-215, 1218-1224, 1313-1331, 1950-1956) against the oracle's restatement,
which runs one particle after the other like the reference at one thread -- the stress of a neighbour that came earlier in
that loop is read already scaled by (1 - damage).  Lists: compute_Beps with Initialize_Beps = false over lists that start
empty (the driver only initialises them for eigenerosion): a particle that has not moved has no neighbour."""
import numpy as np
import pytest

from test_gpu_eigenerosion import stretch_field
from test_gpu_parity import masks
from util import assert_close, gpu_setup, make_case, nlps, oracle_setup, orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ndim,law", [(2, 0), (3, 0), (3, 1)])
def test_eigensoftening_level_b_stages(ndim, law):
    o = orc()
    n = nlps()
    rng = np.random.default_rng(33)
    mat = {"type": law, "E": 1.0e6, "nu": 0.25, "Ceps": 1.5, "ft": 0.0, "heps": 2.0, "wcrit": 0.05}
    if ndim == 2:
        case = make_case(2, [14, 12], [3, 3], [7, 6], material=mat)
    else:
        case = make_case(3, [11, 10, 9], [3, 3, 2], [5, 4, 4], material=mat)
    npart = case["cloud"]["x"].shape[0]
    # half of the cloud has moved before (its neighbourhood is rebuilt), the other half never (empty list)
    dis = np.zeros_like(case["cloud"]["x"])
    dis[rng.uniform(size=npart) < 0.5] = 1e-3
    case["cloud"]["dis"] = dis
    # a few particles start failed, a few partly damaged with a fracture strain on record
    damage0 = np.zeros(npart)
    strain_f0 = np.zeros(npart)
    pick = rng.permutation(npart)
    damage0[pick[:npart // 10]] = 1.0
    damage0[pick[npart // 10: npart // 5]] = 0.3
    strain_f0[pick[npart // 10: npart // 4]] = 1e-3
    case["cloud"]["damage_n"] = damage0.copy()
    case["cloud"]["strain_f_n"] = strain_f0.copy()

    # oracle pass with ft = 0 to learn the equivalent stresses, then an ft that lets about half of the candidates through
    M, P, prm, mats = oracle_setup(case)
    n2m, na = o.active_nodes(M)
    dU = stretch_field(M, n2m, na, ndim, 0.02, rng)
    beps = (np.zeros(P.np, dtype=np.int32), np.full((P.np, o.BEPS_STRIDE), -1, dtype=np.int32))
    o.compute_beps(P, M, mats, beps=beps, initialize=False)
    assert beps[0].min() == 0 and beps[0].max() > 1
    assert o.compatibility(dU, None, P, M, n2m) == 0 and o.constitutive_eroded(P, mats, prm, damage0) == 0
    T = 5 if ndim == 2 else 9
    tau = P["stress"].reshape(npart, T)[:, : ndim * ndim].reshape(npart, ndim, ndim)
    T0 = np.linalg.eigvalsh(0.5 * (tau + np.transpose(tau, (0, 2, 1))))[:, 0]
    cand = (damage0 == 0.0) & (T0 > 0.0)
    assert cand.sum() > npart // 4
    mat["ft"] = float(np.median(T0[cand]))
    case["materials"] = [mat]

    # ---- both sides from scratch with the driver on
    M, P, prm, mats = oracle_setup(case)
    params = n.default_params()
    params.driver_eigensoftening = 1
    S = gpu_setup(case, nsteps=2, params=params)
    n2m, d2m, na = masks(S, M, [], 0, 2)
    beps = (np.zeros(P.np, dtype=np.int32), np.full((P.np, o.BEPS_STRIDE), -1, dtype=np.int32))
    damage_n, damage_n1 = damage0.copy(), damage0.copy()
    strain_f = strain_f0.copy()
    for rnd in range(2):
        o.compute_beps(P, M, mats, beps=beps, initialize=False)  # U-Newmark-beta.c:213-215
        assert o.compatibility(dU, None, P, M, n2m) == 0
        assert o.constitutive_eroded(P, mats, prm, damage_n) == 0
        sf_before = strain_f.copy()
        assert o.eigensoftening_hook(damage_n1, damage_n, strain_f, P, mats, beps) == 0
        R_o, st = o.internal_forces(P, M, n2m, d2m, na)
        assert st == 0
        S.local_compatibility_conditions(dU)
        S.constitutive_update()
        R_g = S.nodal_internal_forces(np.zeros(na * ndim))
        d = S.download_state()
        assert np.array_equal(d["Strain_f_n1"] > 0, strain_f > 0), f"round {rnd}: which particles start to fracture"
        assert_close(d["Strain_f_n1"], strain_f, 1e-10, f"round {rnd}: fracture strain")
        assert_close(d["Damage_n1"], damage_n1, 1e-10, f"round {rnd}: damage field")
        assert_close(d["Stress"], P["stress"], 1e-10, f"round {rnd}: scaled Kirchhoff stress")
        assert_close(R_g, R_o, 1e-10, f"round {rnd}: internal forces")
        started = np.count_nonzero((strain_f > 0) & (sf_before == 0))
        grew = np.count_nonzero(damage_n1 > damage_n)
        assert started > 0 and (rnd == 0 or grew > 0), (started, grew)
        # roll (U-Newmark-beta.c:1950-1956) and go again from the rolled state with a larger stretch
        o.roll_state(P)
        damage_n[:] = damage_n1
        S.update_particles_internal_variables()
        d = S.download_state(["Damage_n", "Strain_f_n"])
        assert_close(d["Damage_n"], damage_n, 1e-10, "rolled damage")
        assert_close(d["Strain_f_n"], strain_f, 1e-10, "rolled fracture strain")
        dU = 1.5 * dU
    with pytest.raises(n.NlpsError):
        S.explicit_step(n.BccSet([]), 0, 1e-4)  # the hooks live in the level-B stages only
