"""GPU-vs-oracle parity away from the benign defaults (h = 1, origin = 0, gamma = 3, TOL_zero = 1e-6).

The kernels evaluate LME in index space on the lattice (`l = a - h u`, the axis recurrence E(o) = E(0) G^o Q^(o^2),
squared cut-off thresholds): with h = 1 and origin = 0 all of that arithmetic is exact.  These cases use non-dyadic
spacings, a shifted origin, narrow and wide LME kernels (gamma_LME) and a tighter zero tolerance; the index maps
(I0, NumberNodes, ListNodes in chain order, ActiveNode, Nodes2Mask, dof masks: LME.c:1052-1082,
Nodes-Tools.c:46-156,476-538) must stay bit-identical, the fields keep the tolerances of test_gpu_parity.py."""
import numpy as np
import pytest

from test_gpu_parity import TOL, compare_search, lists_equal, masks
from util import DP, HENCKY, NH, assert_close, dirichlet_plane, gpu_setup, make_case, nlps, oracle_setup, orc

pytestmark = pytest.mark.gpu

ORIGIN = (-1.3, 2.7, 0.9)
# (h, shifted origin?, gamma_LME, TOL_zero_LME)
SWEEP = [
    (0.1, True, 3.0, 1e-6),
    (0.37, False, 1.8, 1e-6),
    (2.5, True, 4.0, 1e-8),
    (1.0, False, 6.0, 1e-6),
    (0.37, True, 1.8, 1e-8),
    (0.1, False, 6.0, 1e-8),
    (2.5, False, 3.0, 1e-6),
]
IDS = ["h%g-o%d-g%g-t%g" % (h, int(o), g, t) for h, o, g, t in SWEEP]


def sweep_case(ndim, cfg, material=NH, velocity=None, **kw):
    h, shifted, gamma, tol = cfg
    origin = list(ORIGIN[:ndim]) if shifted else None
    if ndim == 2:
        return make_case(2, [14, 12], [3, 3], [7, 6], material=material, velocity=velocity, h=h, origin=origin,
                         gamma=gamma, tol_zero=tol, **kw)
    return make_case(3, [11, 10, 9], [3, 3, 2], [5, 4, 4], material=material, velocity=velocity, h=h, origin=origin,
                     gamma=gamma, tol_zero=tol, **kw)


@pytest.mark.parametrize("ndim", [2, 3])
@pytest.mark.parametrize("cfg", SWEEP, ids=IDS)
def test_sweep_initialize_lme(ndim, cfg):
    case = sweep_case(ndim, cfg)
    M, P, prm, mats = oracle_setup(case)
    S = gpu_setup(case)
    compare_search(S, P, M, "initialize__LME__ %s" % (cfg,))
    # whole-grid cloud: truncated stencils and boundary classes with the same lattice parameters
    h, shifted, gamma, tol = cfg
    origin = list(ORIGIN[:ndim]) if shifted else None
    cells = [8, 7] if ndim == 2 else [6, 5, 5]
    case = make_case(ndim, cells, [0] * ndim, cells, h=h, origin=origin, gamma=gamma, tol_zero=tol)
    M, P, prm, mats = oracle_setup(case)
    S = gpu_setup(case)
    compare_search(S, P, M, "boundary %s" % (cfg,))


@pytest.mark.parametrize("ndim,material", [(2, NH), (3, NH), (2, DP), (3, DP)])
@pytest.mark.parametrize("cfg", SWEEP, ids=IDS)
def test_sweep_stage_functions(ndim, material, cfg):
    """local_search after a random motion of 0.37 h, masks, then one pass through the level-B stages."""
    o = orc()
    h = cfg[0]
    vel = [1.0 * h, -2.0 * h] if ndim == 2 else [1.0 * h, -2.0 * h, 0.5 * h]
    case = sweep_case(ndim, cfg, material=material, velocity=vel)
    rng = np.random.default_rng(11)
    case["cloud"]["acc"][:] = h * rng.normal(size=case["cloud"]["acc"].shape)
    nsteps = 3
    bcs_list = [dirichlet_plane(case, ndim - 1, 3, nsteps)]
    M, P, prm, mats = oracle_setup(case)
    dx = 0.37 * h * rng.uniform(-1, 1, size=P["x"].shape)
    P["x"][:] += dx
    P["dis"][:] += dx
    cloud2 = dict(case["cloud"])
    cloud2.update(x=P["x"].copy(), dis=P["dis"].copy(), I0=P["I0"].copy(), **{"lambda": P["lambda"].copy()},
                  beta=P["beta"].copy())
    case2 = dict(case)
    case2["cloud"] = cloud2
    S = gpu_setup(case2, init=False, nsteps=nsteps)
    assert o.local_search(P, M, prm) == 0
    S.local_search()
    compare_search(S, P, M, "local_search__LME__ %s" % (cfg,))
    n2m, d2m, na = masks(S, M, bcs_list, 1, nsteps)

    Mv_o = o.lumped_mass(P, M, n2m, na)
    Mv_g = S.compute_nodal_lumped_mass()
    assert_close(Mv_g, Mv_o, TOL, "lumped mass")
    V_o, A_o = o.nodal_field_n(Mv_o, P, M, n2m, d2m, na)
    V_g, A_g = S.get_nodal_field_n(Mv_g)
    assert_close(V_g, V_o, TOL, "nodal velocity")
    assert_close(A_g, A_o, TOL, "nodal acceleration")

    dU = (2e-2 if material["type"] == 2 else 1e-3) * h * rng.normal(size=na * ndim)
    assert o.compatibility(dU, None, P, M, n2m) == 0
    S.local_compatibility_conditions(dU)
    assert o.constitutive(P, mats, prm) == 0
    S.constitutive_update()
    st = S.download_state()
    keys = [("DF", "DF"), ("F_n1", "F_n1"), ("J_n1", "J_n1"), ("Stress", "stress"), ("W", "W")]
    if material["type"] == 2:
        keys += [("b_e_n1", "b_e_n1"), ("Kappa_n1", "kappa_n1"), ("EPS_n1", "eps_n1"), ("C_ep", "C_ep")]
        assert np.count_nonzero(P["eps_n1"] > P["eps_n"]) > 0, "the case must exercise the return mapping"
    for k, ok in keys:
        # W = E * O(strain^2) is a difference of O(1) terms (J^2 - 1, tr b - d): its rounding noise is E * O(1e-16)
        assert_close(st[k], P[ok], TOL, f"{k} after compatibility+constitutive",
                     scale=(material["E"] * 1e-4 if k == "W" else None))

    R_o, s = o.internal_forces(P, M, n2m, d2m, na)
    assert s == 0
    R_g = S.nodal_internal_forces(np.zeros(na * ndim))
    assert_close(R_g, R_o, TOL, "internal forces")
    assert np.all(R_g[d2m == -1] == 0.0)
    # the driver's ONE residual call at the same iterate (nlps_gpu_lagrangian_evaluation = __lagrangian_evaluation,
    # U-Newmark-beta.c:970-1058): the three stages above + the inertial term, same state left behind
    a1, a2, a3 = 4.0e4, 4.0e2, 1.0
    grav = [0.0] * (ndim - 1) + [-9.81]
    R_l = R_o.copy()
    free = d2m != -1
    R_l[free] += (Mv_o * (a1 * dU - a2 * V_o - a3 * A_o - np.tile(np.asarray(grav), na)))[free]
    R_f = S.lagrangian_evaluation(dU, V_g, A_g, Mv_g, [a1, a2, a3, 0.0, 0.0, 0.0], grav)
    assert_close(R_f, R_l, TOL, "residual by the one call")
    st = S.download_state()
    for k, ok in keys:
        assert_close(st[k], P[ok], TOL, f"{k} after the one residual call",
                     scale=(material["E"] * 1e-4 if k == "W" else None))

    o.roll_state(P)
    S.update_particles_internal_variables()
    dV = 1e-2 * h * rng.normal(size=na * ndim)
    dA = 1e-1 * h * rng.normal(size=na * ndim)
    o.update_kinetics(1.0, dU, V_o, dV, dA, P, M, n2m)
    S.update_particles_kinetics_FLIP_PIC(1.0, dU, V_g, dV, dA)
    st = S.download_state()
    for k, ok in (("x", "x"), ("dis", "dis"), ("vel", "vel"), ("acc", "acc"), ("F_n", "F_n"), ("J_n", "J_n"),
                  ("rho", "rho")):
        assert_close(st[k], P[ok], TOL, f"{k} after roll+kinetics")


@pytest.mark.parametrize("ndim,material", [(2, NH), (3, NH), (2, HENCKY), (3, DP)])
@pytest.mark.parametrize("cfg", SWEEP, ids=IDS)
def test_sweep_explicit_steps(ndim, material, cfg):
    """Six fused explicit steps (search, lists, Newton, P2G, stress, G2P) against the oracle's composition."""
    o = orc()
    n = nlps()
    h = cfg[0]
    cel = np.sqrt(material["E"] / 1000.0)
    v0 = -0.1 * cel * (0.02 if material["type"] == 2 else 1.0)
    vel = [0.0] * (ndim - 1) + [v0]
    case = sweep_case(ndim, cfg, material=material, velocity=vel)
    nsteps = 6
    bcs_list = [dirichlet_plane(case, ndim - 1, 2, nsteps)]
    grav = [0.0] * (ndim - 1) + [-9.81]
    dt = 0.1 * h / cel
    M, P, prm, mats = oracle_setup(case)
    S = gpu_setup(case, nsteps=nsteps)
    stepper = o.ExplicitStepper(P, M, mats, prm, o.BccSet(bcs_list), nsteps, gravity=grav)
    gb = n.BccSet(bcs_list)
    for t in range(nsteps):
        assert stepper.step(t, dt) == 0
        S.explicit_step(gb, t, dt, 0.5, grav)
        nod = S.explicit_nodal()
        assert S.nactive == stepper.out.nactive
        for k in ("mass", "dU", "force", "accel", "reaction"):
            assert_close(nod[k], stepper.nodal(k), 1e-9, f"step {t} nodal {k}")
        st = S.download_state()
        assert np.array_equal(st["I0"], P["I0"]), f"step {t}: I0"
        nn, lst = S.download_lists()
        assert np.array_equal(nn, P["nn"]) and lists_equal(nn, lst, P["list"]), f"step {t}: lists"
        assert np.array_equal(S.download_active(), M.active()), f"step {t}: ActiveNode"
        for k, ok in (("x", "x"), ("dis", "dis"), ("vel", "vel"), ("acc", "acc"), ("F_n", "F_n"), ("DF", "DF"),
                      ("Stress", "stress"), ("J_n", "J_n"), ("rho", "rho"), ("W", "W"), ("lambda", "lambda"),
                      ("b_e_n", "b_e_n"), ("Kappa_n", "kappa_n"), ("EPS_n", "eps_n")):
            assert_close(st[k], P[ok], 1e-9, f"step {t} {k}", scale={"Stress": material["E"] * 1e-9, "W": material["E"] * 1e-5}.get(k))
    assert S.status_flags() == 0


@pytest.mark.parametrize("ndim", [2, 3])
@pytest.mark.parametrize("cfg", [SWEEP[0], SWEEP[4], SWEEP[5]], ids=[IDS[0], IDS[4], IDS[5]])
def test_sweep_long_flight_index_maps(ndim, cfg):
    """The flight test of test_gpu_parity.py (particles cross cell mid-planes, I0 moves several cells, periodic
    device re-sort) on a non-dyadic, shifted lattice: closest node, lists and activation stay bit-identical."""
    o = orc()
    n = nlps()
    h, shifted, gamma, tol = cfg
    origin = list(ORIGIN[:ndim]) if shifted else None
    soft = {"type": 0, "E": 1.0e5, "nu": 0.3}  # celerity 10
    if ndim == 2:
        case = make_case(2, [12, 24], [3, 15], [5, 5], material=soft, velocity=[2.0, -10.0], h=h, origin=origin,
                         gamma=gamma, tol_zero=tol)
    else:
        case = make_case(3, [10, 10, 20], [3, 3, 12], [4, 4, 4], material=soft, velocity=[2.0, -1.0, -10.0], h=h,
                         origin=origin, gamma=gamma, tol_zero=tol)
    x = case["cloud"]["x"]
    xc = x.mean(axis=0)
    case["cloud"]["vel"][:, ndim - 1] *= 1.0 + 0.2 * (x[:, 0] - xc[0]) / (2.5 * h)
    nsteps = 40
    bcs_list = [dirichlet_plane(case, ndim - 1, 2, nsteps)]
    dt = 0.25 * h / 20.0
    M, P, prm, mats = oracle_setup(case)
    I0_start = P["I0"].copy()
    S = gpu_setup(case, nsteps=nsteps)
    S.set_resort_interval(7)
    stepper = o.ExplicitStepper(P, M, mats, prm, o.BccSet(bcs_list), nsteps)
    gb = n.BccSet(bcs_list)
    for t in range(nsteps):
        assert stepper.step(t, dt) == 0, f"oracle failed at step {t}"
        S.explicit_step(gb, t, dt)
        if t % 8 == 7 or t == nsteps - 1:
            st = S.download_state()
            assert np.array_equal(st["I0"], P["I0"]), f"step {t}: I0"
            nn, lst = S.download_lists()
            assert np.array_equal(nn, P["nn"]) and lists_equal(nn, lst, P["list"]), f"step {t}: lists"
            assert np.array_equal(S.download_active(), M.active()), f"step {t}: ActiveNode"
            for k, ok in (("x", "x"), ("vel", "vel"), ("F_n", "F_n"), ("lambda", "lambda")):
                assert_close(st[k], P[ok], 1e-8, f"step {t} {k}")
    assert S.status_flags() == 0
    X = M.coords().reshape(-1, ndim)
    assert np.abs(X[P["I0"]] - X[I0_start]).max(axis=0)[ndim - 1] >= 4.0 * h, "the block has to travel several cells"
