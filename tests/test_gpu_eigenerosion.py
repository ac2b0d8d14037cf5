"""Eigenerosion: the damage hooks of the maintained implicit driver's stages against the oracle's
restatement of Constitutive/Fracture/{Beps.c, EigenErosion.c} and U-Newmark-beta.c:1218-1224, 1313-1331, 1757-1764,
1950-1953.  A block is stretched so that every principal Kirchhoff stress is positive; the critical energy release rate
Gf is set between the quantiles of the particles' G so that part of the cloud fails."""
import numpy as np
import pytest

from test_gpu_parity import masks
from util import assert_close, gpu_setup, make_case, nlps, oracle_setup, orc

pytestmark = pytest.mark.gpu


def stretch_field(M, n2m, na, ndim, amount, rng):
    X = M.coords().reshape(-1, ndim)
    act = np.where(n2m >= 0)[0]
    dU = np.zeros((na, ndim))
    dU[n2m[act]] = amount * (X[act] - X.mean(0)) * (1.0 + 0.5 * rng.uniform(size=(act.size, 1)))
    return dU.ravel()


@pytest.mark.parametrize("ndim,law", [(2, 0), (3, 0), (3, 1)])
def test_eigenerosion_level_b_stages(ndim, law):
    o = orc()
    n = nlps()
    rng = np.random.default_rng(21)
    mat = {"type": law, "E": 1.0e6, "nu": 0.25, "Ceps": 1.5, "Gf": 0.0}
    if ndim == 2:
        case = make_case(2, [14, 12], [3, 3], [7, 6], material=mat)
    else:
        case = make_case(3, [11, 10, 9], [3, 3, 2], [5, 4, 4], material=mat)
    M, P, prm, mats = oracle_setup(case)
    n2m, na = o.active_nodes(M)
    dU = stretch_field(M, n2m, na, ndim, 0.02, rng)
    # oracle pass 1 with Gf = 0 to learn the energy release rates, then a Gf that fails about half of the cloud
    damage0 = np.zeros(P.np)
    beps = o.compute_beps(P, M, mats, initialize=True)
    assert beps[0].min() >= 1 and beps[0].max() < o.BEPS_STRIDE  # every particle is in its own neighbourhood
    assert o.compatibility(dU, None, P, M, n2m) == 0 and o.constitutive_eroded(P, mats, prm, damage0) == 0
    W = P["W"].copy()
    V = P["vol0"] * P["J_n1"]
    G = np.zeros(P.np)
    for p in range(P.np):
        q = beps[1][p, : beps[0][p]]
        G[p] = mat["Ceps"] * case["h"] / (V[p] + V[q].sum()) * (V[p] * W[p] + (V[q] * W[q]).sum())
    Gf = float(np.median(G))
    assert np.count_nonzero(np.abs(G - Gf) < 1e-9 * Gf) == 0, "no particle may sit on the threshold"
    mat["Gf"] = Gf
    case["materials"] = [mat]

    # ---- both sides from scratch with the driver on
    M, P, prm, mats = oracle_setup(case)
    params = n.default_params()
    params.driver_eigenerosion = 1
    S = gpu_setup(case, nsteps=2, params=params)
    n2m, d2m, na = masks(S, M, [], 0, 2)
    beps = o.compute_beps(P, M, mats, initialize=True)
    damage_n, damage_n1 = np.zeros(P.np), np.zeros(P.np)
    for rnd in range(2):  # second round: the failed particles are skipped by the constitutive stage
        assert o.compatibility(dU, None, P, M, n2m) == 0
        assert o.constitutive_eroded(P, mats, prm, damage_n) == 0
        assert o.eigenerosion_hook(damage_n1, damage_n, P, mats, beps, case["h"]) == 0
        R_o, st = o.internal_forces(P, M, n2m, d2m, na)
        assert st == 0
        S.local_compatibility_conditions(dU)
        S.constitutive_update()
        R_g = S.nodal_internal_forces(np.zeros(na * ndim))
        d = S.download_state()
        assert np.array_equal(d["Damage_n1"], damage_n1), f"round {rnd}: damage field"
        assert_close(d["Stress"], P["stress"], 1e-10, f"round {rnd}: scaled Kirchhoff stress")
        assert_close(d["W"], P["W"], 1e-10, f"round {rnd}: W", scale=mat["E"] * 1e-4)
        assert_close(R_g, R_o, 1e-10, f"round {rnd}: internal forces")
        failed = int(damage_n1.sum())
        assert 0 < failed < P.np
        # the tangent scales with (1 - damage) (Neo-Hookean: the law of the device assembly's first path)
        if law == 0:
            o.set_tangent_damage(damage_n1)
            K_o, pat, stt = o.tangent_matrix(P, M, mats, n2m, None, na, with_pattern=False)
            o.set_tangent_damage(None)
            rows, cols, vals = S.jacobian_evaluation(0.0, None, False)
            K_g = np.zeros_like(K_o)
            np.add.at(K_g, (rows, cols), vals)
            assert_close(K_g, K_o, 1e-9, f"round {rnd}: tangent with damage")
        # roll (U-Newmark-beta.c:1950-1953) and go again from the rolled state
        o.roll_state(P)
        damage_n[:] = damage_n1
        S.update_particles_internal_variables()
        d = S.download_state(["Damage_n"])
        assert np.array_equal(d["Damage_n"], damage_n)
        dU = 0.5 * dU
    with pytest.raises(n.NlpsError):
        S.explicit_step(n.BccSet([]), 0, 1e-4)  # the hooks live in the level-B stages only


def test_eigenerosion_frozen_neighbourhoods_of_resting_particles():
    """Beps.c:30-36: compute_Beps only rebuilds the list of a particle whose TOTAL displacement exceeds 1e-6; a particle at
    rest keeps the list of the initialisation (U-Newmark-beta.c:182-183) while its neighbours move in and out of reach.
    Half of the block is moved by a third of a cell (an update of the kinetics with nodal increments that vanish on the
    other half), then the search, the lists (Initialize_Beps = false) and the damage hook run on both sides."""
    o = orc()
    n = nlps()
    rng = np.random.default_rng(5)
    ndim = 3
    mat = {"type": 0, "E": 1.0e6, "nu": 0.25, "Ceps": 1.2, "Gf": 0.0}
    case = make_case(3, [14, 10, 9], [3, 3, 2], [8, 4, 4], material=mat)
    M, P, prm, mats = oracle_setup(case)
    params = n.default_params()
    params.driver_eigenerosion = 1
    S = gpu_setup(case, nsteps=3, params=params)
    n2m, d2m, na = masks(S, M, [], 0, 3)
    beps = o.compute_beps(P, M, mats, initialize=True)  # U-Newmark-beta.c:182-183
    # nodal increments: zero on the low-x half, a third of a cell along x on the other
    X = M.coords().reshape(-1, ndim)
    act = np.where(n2m >= 0)[0]
    dU = np.zeros((na, ndim))
    xmid = 0.5 * (case["cloud"]["x"][:, 0].min() + case["cloud"]["x"][:, 0].max())
    moving = X[act, 0] > xmid
    dU[n2m[act[moving]], 0] = 0.34 * case["h"] * (1.0 + 0.2 * rng.uniform(size=moving.sum()))
    dU = dU.ravel()
    zero = np.zeros(na * ndim)
    assert o.local_search(P, M, prm) == 0  # the driver's first step starts with the search too (U-Newmark-beta.c:198-215)
    S.local_search()
    S.active_masks(n.BccSet([]), 0)
    assert o.compatibility(zero, None, P, M, n2m) == 0 and o.constitutive(P, mats, prm) == 0
    S.local_compatibility_conditions(zero)
    S.constitutive_update()
    o.roll_state(P)
    S.update_particles_internal_variables()
    o.update_kinetics(1.0, dU, zero, zero, zero, P, M, n2m)
    S.update_particles_kinetics_FLIP_PIC(1.0, dU, zero, zero, zero)
    resting = np.sqrt((P["dis"] ** 2).sum(axis=1)) <= 1e-6
    assert 0.2 * P.np < resting.sum() < 0.8 * P.np
    # next step: search, lists, stretch, stress, hook
    assert o.local_search(P, M, prm) == 0
    S.local_search()
    n2m, d2m, na = masks(S, M, [], 1, 3)
    frozen_lists = beps[1].copy(), beps[0].copy()
    o.compute_beps(P, M, mats, beps=beps, initialize=False)
    fresh = o.compute_beps(P, M, mats, initialize=True)
    differ = [p for p in np.where(resting)[0] if set(beps[1][p, :beps[0][p]]) != set(fresh[1][p, :fresh[0][p]])]
    assert len(differ) > 0, "the case must hold resting particles whose frozen list differs from a recomputed one"
    assert all(np.array_equal(beps[1][p], frozen_lists[0][p]) for p in np.where(resting)[0])
    dU2 = stretch_field(M, n2m, na, ndim, 0.02, rng)
    damage_n, damage_n1 = np.zeros(P.np), np.zeros(P.np)
    assert o.compatibility(dU2, None, P, M, n2m) == 0 and o.constitutive_eroded(P, mats, prm, damage_n) == 0
    V = P["vol0"] * P["J_n1"]
    G = np.array([mat["Ceps"] * case["h"] / (V[p] + V[beps[1][p, :beps[0][p]]].sum()) *
                  (V[p] * P["W"][p] + (V[beps[1][p, :beps[0][p]]] * P["W"][beps[1][p, :beps[0][p]]]).sum()) for p in range(P.np)])
    # a threshold inside the spread of the resting particles whose list matters
    gs = np.sort(G[differ])
    Gf = float(0.5 * (gs[len(gs) // 2 - 1] + gs[len(gs) // 2]))  # between two particles, never on one
    assert np.count_nonzero(np.abs(G - Gf) < 1e-9 * Gf) == 0, "no particle may sit on the threshold"
    S.close()
    mat["Gf"] = Gf
    case["materials"] = [mat]
    # ---- again from scratch with that Gf, the device beside the oracle
    M, P, prm, mats = oracle_setup(case)
    S = gpu_setup(case, nsteps=3, params=params)
    n2m, d2m, na = masks(S, M, [], 0, 3)
    beps = o.compute_beps(P, M, mats, initialize=True)
    assert o.local_search(P, M, prm) == 0
    S.local_search()  # (the driver's first step: the library takes its snapshot of the initial configuration here)
    S.active_masks(n.BccSet([]), 0)
    assert o.compatibility(zero, None, P, M, n2m) == 0 and o.constitutive(P, mats, prm) == 0
    S.local_compatibility_conditions(zero)
    S.constitutive_update()
    o.roll_state(P)
    S.update_particles_internal_variables()
    o.update_kinetics(1.0, dU, zero, zero, zero, P, M, n2m)
    S.update_particles_kinetics_FLIP_PIC(1.0, dU, zero, zero, zero)
    assert o.local_search(P, M, prm) == 0
    S.local_search()
    n2m, d2m, na = masks(S, M, [], 1, 3)
    o.compute_beps(P, M, mats, beps=beps, initialize=False)
    assert o.compatibility(dU2, None, P, M, n2m) == 0 and o.constitutive_eroded(P, mats, prm, damage_n) == 0
    assert o.eigenerosion_hook(damage_n1, damage_n, P, mats, beps, case["h"]) == 0
    R_o, st = o.internal_forces(P, M, n2m, d2m, na)
    S.local_compatibility_conditions(dU2)
    S.constitutive_update()
    R_g = S.nodal_internal_forces(np.zeros(na * ndim))
    d = S.download_state()
    assert np.array_equal(d["Damage_n1"], damage_n1), "damage field with frozen neighbourhoods"
    assert 0 < damage_n1[differ].sum() < len(differ)
    assert_close(d["Stress"], P["stress"], 1e-10, "scaled Kirchhoff stress")
    assert_close(R_g, R_o, 1e-10, "internal forces")
    S.close()
