"""nlps_gpu_lagrangian_evaluation: the residual callback of the maintained implicit driver (__lagrangian_evaluation,
U-Newmark-beta.c:970-1058) as ONE device call, against the oracle's composition of the stage functions it is made of
(:1018-1036: velocity increments, compatibility, constitutive update, internal, traction and inertial forces)."""
import numpy as np
import pytest

from newmark import newmark_parameters, newmark_step
from util import DP, HENCKY, NH, VM, assert_close, dirichlet_plane, gpu_setup, make_case, nlps, oracle_setup, orc

pytestmark = pytest.mark.gpu

TOL = 1e-10


def _moved_case(ndim, material, nsteps, rng):
    """The cloud of test_stage_functions: moving, with accelerations, displaced by a third of a cell after the initial
    search so that the level-B search has work to do."""
    vel = [1.0, -2.0] if ndim == 2 else [1.0, -2.0, 0.5]
    if ndim == 2:
        case = make_case(2, [14, 12], [3, 3], [7, 6], material=material, velocity=vel)
    else:
        case = make_case(3, [11, 10, 9], [3, 3, 2], [5, 4, 4], material=material, velocity=vel)
    case["cloud"]["acc"][:] = rng.normal(size=case["cloud"]["acc"].shape)
    M, P, prm, mats = oracle_setup(case)
    dx = 0.37 * case["h"] * rng.uniform(-1, 1, size=P["x"].shape)
    P["x"][:] += dx
    P["dis"][:] += dx
    cloud2 = dict(case["cloud"])
    cloud2.update(x=P["x"].copy(), dis=P["dis"].copy(), I0=P["I0"].copy(), **{"lambda": P["lambda"].copy()},
                  beta=P["beta"].copy())
    case2 = dict(case)
    case2["cloud"] = cloud2
    return case2, M, P, prm, mats


def _oracle_residual(o, P, M, mats, prm, n2m, d2m, na, ndim, dU, Un_dt, Un_dt2, Mv, a, gravity, loads, step, nsteps,
                     thickness, area0, rates=True):
    """__lagrangian_evaluation composed from the oracle's stage functions, in the order of :1018-1036"""
    dU_dt = a["a4"] * dU + (a["a5"] - 1) * Un_dt + a["a6"] * Un_dt2      # __compute_nodal_velocity_increments :1836-1856
    assert o.compatibility(dU, dU_dt if rates else None, P, M, n2m) == 0
    assert o.constitutive(P, mats, prm) == 0
    R, st = o.internal_forces(P, M, n2m, d2m, na)
    assert st == 0
    if loads:
        assert o.nodal_traction_forces(R, P, M, n2m, d2m, loads, step, nsteps, thickness, area0) == 0
    free = d2m != -1
    bvec = np.tile(np.asarray(gravity, dtype=np.float64), na)
    R[free] += (Mv * (a["a1"] * dU - a["a2"] * Un_dt - a["a3"] * Un_dt2 - bvec))[free]   # :1519-1557
    return R


STATE = (("DF", "DF"), ("F_n1", "F_n1"), ("J_n1", "J_n1"), ("Stress", "stress"), ("W", "W"), ("b_e_n1", "b_e_n1"),
         ("Kappa_n1", "kappa_n1"), ("EPS_n1", "eps_n1"), ("C_ep", "C_ep"))
N_STATE = (("F_n", "F_n"), ("J_n", "J_n"), ("b_e_n", "b_e_n"), ("Kappa_n", "kappa_n"), ("EPS_n", "eps_n"))


def _compare_state(S, P, material, what, amp=1e-3, tol=TOL):
    st = S.download_state()
    plastic_law = material["type"] in (2, 3)
    for k, ok in STATE + N_STATE:
        if not plastic_law and k in ("b_e_n1", "Kappa_n1", "EPS_n1", "C_ep", "b_e_n", "Kappa_n", "EPS_n"):
            continue
        # the stored energy of a small strain is a difference of O(E) terms: W ~ E amp^2 carries E x 1e-16 of rounding on
        # either side, i.e. 1e-16 / amp^2 of its own magnitude
        assert_close(st[k], P[ok], max(tol, 2e-16 / amp ** 2) if k == "W" else tol, f"{what}: {k}")
    return st


@pytest.mark.parametrize("ndim,material", [(2, NH), (3, NH), (2, HENCKY), (3, HENCKY), (2, DP), (3, DP), (3, VM)])
def test_lagrangian_evaluation_matches_the_oracle_composition(ndim, material):
    o, n = orc(), nlps()
    rng = np.random.default_rng(11)
    nsteps, step = 3, 1
    case, M, P, prm, mats = _moved_case(ndim, material, nsteps, rng)
    bcs_list = [dirichlet_plane(case, ndim - 1, 3, nsteps)]
    S = gpu_setup(case, init=False, nsteps=nsteps)
    assert o.local_search(P, M, prm) == 0
    S.local_search()
    n2m_o, na = o.active_nodes(M)
    d2m_o, nfree = o.active_dofs(n2m_o, na, ndim, o.BccSet(bcs_list), step, nsteps)
    n2m, d2m = S.active_masks(n.BccSet(bcs_list), step)
    assert np.array_equal(n2m, n2m_o) and np.array_equal(d2m, d2m_o)
    Mv = o.lumped_mass(P, M, n2m, na)
    Un_dt, Un_dt2 = o.nodal_field_n(Mv, P, M, n2m, d2m, na)
    a = newmark_parameters(0.25, 0.5, 2.0e-3)
    alpha = [a["a1"], a["a2"], a["a3"], a["a4"], a["a5"], a["a6"]]
    grav = [0.0] * (ndim - 1) + [-9.81]
    npart = case["cloud"]["x"].shape[0]
    pick = rng.choice(npart, size=20, replace=False).astype(np.int32)
    d1, d2 = np.ones((ndim, nsteps), dtype=np.int32), np.ones((ndim, nsteps), dtype=np.int32)
    d2[0, step] = 0
    loads = [{"nodes": pick[:12], "dim": ndim, "dir": d1, "value": rng.normal(size=(ndim, nsteps)) * 1e5},
             {"nodes": pick[12:], "dim": ndim, "dir": d2, "value": rng.normal(size=(ndim, nsteps)) * 1e5}]
    area0 = rng.uniform(0.2, 0.3, size=npart) if ndim == 3 else None
    gl = n.BccSet(loads)
    plastic_law = material["type"] in (2, 3)
    # The SNES solver evaluates the residual at several iterates from ONE n state: different dU in a row, each against a
    # fresh oracle composition (Von-Mises moves its back stress in place at every evaluation, like upstream,
    # Constitutive.c:116: two evaluations at one dU differ there, on both sides alike).  Then the same call as the
    # composition of the separate stage entries, with the rate tensors, and with device-resident vectors.
    import torch
    big, small = (2e-2, 1e-2) if plastic_law else (1e-3, 2e-3)
    variants = [("fused", big, 0, False), ("fused again", small, 0, False), ("fused, third", big, 0, False),
                ("separate stages", big, S.LAGR_SEPARATE, False), ("rate tensors", small, S.LAGR_RATES, False),
                ("same step", big, S.LAGR_SAME_STEP, False), ("device vectors", big, 0, True)]
    for what, amp, flags, on_device in variants:
        dU = amp * rng.normal(size=na * ndim)
        R_o = _oracle_residual(o, P, M, mats, prm, n2m, d2m, na, ndim, dU, Un_dt, Un_dt2, Mv, a, grav, loads, step, nsteps,
                               0.5, area0)
        if on_device:  # a PETSc Vec of a GPU type: nothing crosses PCIe; R is overwritten, not accumulated into
            dev = [torch.from_numpy(np.ascontiguousarray(v)).cuda() for v in (dU, Un_dt, Un_dt2, Mv)]
            R_d = torch.full((na * ndim,), 7.0, dtype=torch.float64, device="cuda")
            S.lagrangian_evaluation(dev[0], dev[1], dev[2], dev[3], alpha, grav, gl, step, 0.5, area0, out=R_d)
            R_g = R_d.cpu().numpy()
        elif flags == S.LAGR_SAME_STEP:  # Un_dt, Un_dt2, M as staged by the evaluation before: the host arrays are not read
            z = np.full(na * ndim, np.nan)
            R_g = S.lagrangian_evaluation(dU, z, z, z, alpha, grav, gl, step, 0.5, area0, flags=flags)
        else:
            R_g = S.lagrangian_evaluation(dU, Un_dt, Un_dt2, Mv, alpha, grav, gl, step, 0.5, area0, flags=flags)
        assert_close(R_g, R_o, TOL, f"{what}: residual")
        assert np.all(R_g[d2m == -1] == 0.0), "Dirichlet dofs carry no residual"
        st = _compare_state(S, P, material, what, amp)
        if flags == S.LAGR_RATES:
            assert_close(st["dt_DF"], P["dt_DF"], TOL, "dt_DF")
            assert_close(st["dt_F_n1"], P["dt_F_n1"], TOL, "dt_F_n1")
        if plastic_law and what == "fused":
            assert np.count_nonzero(P["eps_n1"] > P["eps_n"]) > 0, "the case must yield"
    # what follows the residual in the driver follows it here: the roll and the particle update
    dV = a["a4"] * dU + (a["a5"] - 1) * Un_dt + a["a6"] * Un_dt2
    dA = a["a1"] * dU - a["a2"] * Un_dt - (a["a3"] + 1) * Un_dt2
    o.roll_state(P)
    S.update_particles_internal_variables()
    o.update_kinetics(1.0, dU, Un_dt, dV, dA, P, M, n2m)
    S.update_particles_kinetics_FLIP_PIC(1.0, dU, Un_dt, dV, dA)
    st = S.download_state()
    for k, ok in (("x", "x"), ("vel", "vel"), ("acc", "acc"), ("F_n", "F_n"), ("J_n", "J_n"), ("rho", "rho")):
        assert_close(st[k], P[ok], TOL, f"{k} after roll + kinetics")
    S.close()


@pytest.mark.parametrize("ndim,law", [(2, "neo-hookean"), (3, "drucker-prager")])
def test_tangent_after_the_fused_residual(ndim, law):
    """SNES asks for the Jacobian at the iterate it last evaluated the residual at: the assembly reads DF, F_n1, tau,
    b_e_n1 and C_ep as the residual call left them -- the same matrix as after the three separate stages."""
    n = nlps()
    rng = np.random.default_rng(3)
    material = NH if law == "neo-hookean" else DP
    nsteps, step = 2, 0
    if ndim == 2:
        case = make_case(2, [12, 11], [3, 3], [5, 4], material=material, velocity=[0.5, -1.0])
    else:
        case = make_case(3, [8, 8, 7], [3, 3, 2], [2, 2, 2], material=material, velocity=[0.5, 0.2, -1.0])
    gb = n.BccSet([dirichlet_plane(case, ndim - 1, 3 if ndim == 2 else 2, nsteps)])
    A, B = gpu_setup(case, nsteps=nsteps), gpu_setup(case, nsteps=nsteps)
    out = []
    for S, fused in ((A, True), (B, False)):
        S.local_search()
        S.active_masks(gb, step)
        Mv = S.compute_nodal_lumped_mass()
        V, Acc = S.get_nodal_field_n(Mv)
        if fused:
            dU = (2e-2 if material["type"] == 2 else 1e-3) * rng.normal(size=S.nactive * ndim)
        a = newmark_parameters(0.25, 0.5, 1.0e-2)
        alpha = [a["a1"], a["a2"], a["a3"], a["a4"], a["a5"], a["a6"]]
        if fused:
            R = S.lagrangian_evaluation(dU, V, Acc, Mv, alpha, None)
        else:
            S.local_compatibility_conditions(dU)
            S.constitutive_update()
            R = S.nodal_internal_forces(np.zeros(S.nactive * ndim))
            R = S.nodal_inertial_forces(R, Mv, dU, V, Acc, alpha, None)
        rows, cols, vals = S.jacobian_evaluation(a["a1"], Mv, True)
        out.append((R, rows, cols, vals))
    assert_close(out[0][0], out[1][0], 1e-12, "residual: fused vs stages")
    ntot = A.nactive * ndim
    K = [np.zeros((ntot, ntot)) for _ in range(2)]
    for q in range(2):
        np.add.at(K[q], (out[q][1], out[q][2]), out[q][3])
    assert np.abs(K[1]).max() > 0
    assert_close(K[0], K[1], 1e-11, "tangent after the fused residual vs after the separate stages")


@pytest.mark.parametrize("ndim", [2, 3])
def test_implicit_newmark_steps_through_the_fused_residual(ndim):
    """test_implicit_newmark_steps_with_device_stages with the device side calling nlps_gpu_lagrangian_evaluation where
    the driver calls __lagrangian_evaluation: three implicit steps (gravity, fixed floor, dt = 2.8 x the explicit limit),
    same iteration counts, quadratic tail, same particles as the oracle's stage-by-stage run."""
    from test_gpu_parity import _DeviceStages, _OracleStages

    class _FusedResidual(_DeviceStages):
        def lagrangian(self, dU, Un_dt, Un_dt2, M, alpha, gravity):
            return self.S.lagrangian_evaluation(dU, Un_dt, Un_dt2, M, alpha, gravity)

    soft = {"type": 0, "E": 2.0e5, "nu": 0.3}
    if ndim == 2:
        case = make_case(2, [12, 11], [3, 3], [5, 4], material=soft, velocity=[0.5, -1.0])
    else:
        case = make_case(3, [8, 8, 7], [3, 3, 2], [2, 2, 2], material=soft, velocity=[0.5, 0.2, -1.0])
    nsteps = 3
    bcs_list = [dirichlet_plane(case, ndim - 1, 3 if ndim == 2 else 2, nsteps)]
    grav = [0.0] * (ndim - 1) + [-9.81]
    dt = 2.0e-2
    A, B = _OracleStages(case, nsteps), _FusedResidual(case, nsteps)
    for step in range(nsteps):
        dU_o, hist_o = newmark_step(A, ndim, bcs_list, step, nsteps, dt, grav)
        dU_g, hist_g = newmark_step(B, ndim, bcs_list, step, nsteps, dt, grav)
        assert len(hist_o) == len(hist_g) and 2 <= len(hist_g) <= 8, (hist_o, hist_g)
        assert hist_g[-1] <= 1e-10 * max(1.0, hist_g[0]), hist_g
        if len(hist_g) >= 4:
            assert hist_g[-1] <= 1e-3 * hist_g[-2], hist_g
        assert_close(dU_g, dU_o, 1e-8, f"step {step}: converged dU")
    st = B.S.download_state()
    for k, ok in (("x", "x"), ("vel", "vel"), ("acc", "acc"), ("F_n", "F_n"), ("Stress", "stress"), ("J_n", "J_n")):
        assert_close(st[k], A.P[ok], 1e-8, f"after implicit steps: {k}")


def test_lagrangian_evaluation_after_explicit_steps_and_with_the_damage_hooks():
    """(a) called right after fused explicit steps (renamed n / n+1 slots, lazily made nodal arrays) it sees the reference's
    copy semantics; (b) with Driver_EigenErosion on the call runs the separate stages (every stress before any force) and
    says nothing else: same residual as calling them one by one."""
    n, o = nlps(), orc()
    rng = np.random.default_rng(5)
    nsteps = 4
    case = make_case(3, [11, 10, 9], [3, 3, 2], [5, 4, 4], material=NH, velocity=[0.0, 0.0, -10.0])
    bcs_list = [dirichlet_plane(case, 2, 2, nsteps)]
    M, P, prm, mats = oracle_setup(case)
    S = gpu_setup(case, nsteps=nsteps)
    stepper = o.ExplicitStepper(P, M, mats, prm, o.BccSet(bcs_list), nsteps)
    gb = n.BccSet(bcs_list)
    for t in range(2):
        assert stepper.step(t, 1e-4) == 0
        S.explicit_step(gb, t, 1e-4)
    assert o.local_search(P, M, prm) == 0
    S.local_search()
    n2m, na = o.active_nodes(M)
    d2m, _ = o.active_dofs(n2m, na, 3, o.BccSet(bcs_list), 2, nsteps)
    S.active_masks(gb, 2)
    Mv = o.lumped_mass(P, M, n2m, na)
    V, Acc = o.nodal_field_n(Mv, P, M, n2m, d2m, na)
    a = newmark_parameters(0.25, 0.5, 1.0e-3)
    alpha = [a["a1"], a["a2"], a["a3"], a["a4"], a["a5"], a["a6"]]
    dU = 1e-3 * rng.normal(size=na * 3)
    R_o = _oracle_residual(o, P, M, mats, prm, n2m, d2m, na, 3, dU, V, Acc, Mv, a, [0, 0, -9.81], None, 2, nsteps, 1.0, None)
    R_g = S.lagrangian_evaluation(dU, V, Acc, Mv, alpha, [0, 0, -9.81])
    assert_close(R_g, R_o, TOL, "residual after explicit steps")
    _compare_state(S, P, NH, "after explicit steps", tol=1e-9)
    S.close()
    # (b)
    prm_g = n.default_params()
    prm_g.driver_eigenerosion = 1
    mat = dict(NH, Ceps=1.5, Gf=1.0e2)
    case = make_case(3, [9, 9, 8], [3, 3, 2], [3, 3, 3], material=mat, velocity=[0.0, 0.0, -1.0])
    case["cloud"]["damage_n"] = np.zeros(case["cloud"]["x"].shape[0])
    outs = []
    for fused_entry in (True, False):
        S = gpu_setup(case, nsteps=nsteps, params=prm_g)
        S.local_search()
        S.active_masks(n.BccSet([dirichlet_plane(case, 2, 2, nsteps)]), 0)
        Mv = S.compute_nodal_lumped_mass()
        V, Acc = S.get_nodal_field_n(Mv)
        if fused_entry:
            dU = 5e-2 * rng.normal(size=S.nactive * 3)
            R = S.lagrangian_evaluation(dU, V, Acc, Mv, alpha, None)
        else:
            S.local_compatibility_conditions(dU)
            S.constitutive_update()
            R = S.nodal_internal_forces(np.zeros(S.nactive * 3))
            R = S.nodal_inertial_forces(R, Mv, dU, V, Acc, alpha, None)
        outs.append((R, S.download_state()))
        S.close()
    assert_close(outs[0][0], outs[1][0], 1e-12, "damage hooks: the entry vs the stages one by one")
    assert np.array_equal(outs[0][1]["Damage_n1"], outs[1][1]["Damage_n1"])


@pytest.mark.parametrize("ndim", [2, 3])
def test_quasi_static_residual(ndim):
    """U-Static.c's __lagrangian_evaluation (:492-551): compatibility, constitutive update, internal forces, traction
    forces and - M b on the free dofs (:1005-1033) -- the dynamic call with alpha = 0, dU standing in for the rate vectors."""
    o, n = orc(), nlps()
    rng = np.random.default_rng(17)
    nsteps, step = 2, 0
    case, M, P, prm, mats = _moved_case(ndim, HENCKY, nsteps, rng)
    bcs_list = [dirichlet_plane(case, ndim - 1, 3, nsteps)]
    S = gpu_setup(case, init=False, nsteps=nsteps)
    assert o.local_search(P, M, prm) == 0
    S.local_search()
    n2m, na = o.active_nodes(M)
    d2m, _ = o.active_dofs(n2m, na, ndim, o.BccSet(bcs_list), step, nsteps)
    S.active_masks(n.BccSet(bcs_list), step)
    Mv = o.lumped_mass(P, M, n2m, na)
    dU = 1e-3 * rng.normal(size=na * ndim)
    b = [0.0] * (ndim - 1) + [-9.81]
    assert o.compatibility(dU, None, P, M, n2m) == 0 and o.constitutive(P, mats, prm) == 0
    R_o, st = o.internal_forces(P, M, n2m, d2m, na)
    assert st == 0
    free = d2m != -1
    R_o[free] += (-Mv * np.tile(np.asarray(b), na))[free]
    R_g = S.lagrangian_evaluation(dU, dU, dU, Mv, [0.0] * 6, b)
    assert_close(R_g, R_o, TOL, "quasi-static residual")
    assert np.all(R_g[~free] == 0.0)
    _compare_state(S, P, HENCKY, "quasi-static residual")


@pytest.mark.parametrize("ndim,layout", [(2, "interleaved"), (3, "interleaved"), (3, "layers")])
def test_lagrangian_evaluation_of_a_cloud_of_three_laws(ndim, layout):
    """Constitutive.c:28-258 dispatches per particle: a cloud that mixes Neo-Hookean, Hencky and Drucker-Prager particles
    runs one fused launch per law (each workgroup compacts its tile's particles of that law); residual and particle state
    against the oracle's composition, two evaluations in a row."""
    o, n = orc(), nlps()
    rng = np.random.default_rng(23)
    nsteps, step = 2, 0
    case, M, P0, prm, _ = _moved_case(ndim, DP, nsteps, rng)
    soft_nh, soft_hencky = {"type": 0, "E": 2.0e4, "nu": 0.3}, {"type": 1, "E": 1.0e4, "nu": 0.25}
    case["materials"] = [soft_nh, soft_hencky, DP]
    npart = case["cloud"]["x"].shape[0]
    if layout == "interleaved":
        case["cloud"]["matidx"] = (np.arange(npart) % 3).astype(np.int32)
    else:
        z = case["cloud"]["x"][:, ndim - 1]
        case["cloud"]["matidx"] = np.minimum(2, ((z - z.min()) / (z.max() - z.min() + 1e-9) * 3).astype(np.int32))
    P = o.OracleParticles(case["cloud"])
    mats = o.make_materials(case["materials"])
    bcs_list = [dirichlet_plane(case, ndim - 1, 3, nsteps)]
    S = gpu_setup(case, init=False, nsteps=nsteps)
    assert o.local_search(P, M, prm) == 0
    S.local_search()
    n2m, na = o.active_nodes(M)
    d2m, _ = o.active_dofs(n2m, na, ndim, o.BccSet(bcs_list), step, nsteps)
    S.active_masks(n.BccSet(bcs_list), step)
    Mv = o.lumped_mass(P, M, n2m, na)
    V, A = o.nodal_field_n(Mv, P, M, n2m, d2m, na)
    a = newmark_parameters(0.25, 0.5, 2.0e-3)
    alpha = [a["a1"], a["a2"], a["a3"], a["a4"], a["a5"], a["a6"]]
    grav = [0.0] * (ndim - 1) + [-9.81]
    for it, amp in enumerate((2e-2, 1e-2)):
        dU = amp * rng.normal(size=na * ndim)
        R_o = _oracle_residual(o, P, M, mats, prm, n2m, d2m, na, ndim, dU, V, A, Mv, a, grav, None, step, nsteps, 1.0, None)
        R_g = S.lagrangian_evaluation(dU, V, A, Mv, alpha, grav)
        assert_close(R_g, R_o, TOL, f"three laws, evaluation {it}: residual")
        st = S.download_state()
        for k, ok in STATE + N_STATE:
            if k == "W":
                continue
            assert_close(st[k], P[ok], TOL, f"three laws, evaluation {it}: {k}", scale=1e-6 if k in ("EPS_n1", "EPS_n") else None)
    dp = case["cloud"]["matidx"] == 2
    assert np.count_nonzero(P["eps_n1"][dp] > P["eps_n"][dp]) > 0, "the plastic third must yield"
    assert np.all(P["eps_n1"][~dp] == 0.0)


@pytest.mark.parametrize("on_device", [False, True])
def test_a_failed_stress_update_in_the_fused_residual_is_reported(on_device):
    """Stress_integration__Constitutive__ failing inside the one call (here: a NaN displacement increment makes the
    eigenvalues of b NaN, Hencky.c) comes back as an error with status flag 8 -- the last kernel of the call leaves the
    status word where the host reads it after its one synchronise."""
    import torch
    n = nlps()
    rng = np.random.default_rng(5)
    nsteps, step, ndim = 2, 1, 3
    case, M, P, prm, mats = _moved_case(ndim, HENCKY, nsteps, rng)
    S = gpu_setup(case, init=False, nsteps=nsteps)
    S.local_search()
    n2m, d2m = S.active_masks(n.BccSet([dirichlet_plane(case, ndim - 1, 3, nsteps)]), step)
    na = int(n2m.max()) + 1
    a = newmark_parameters(0.25, 0.5, 2.0e-3)
    alpha = [a["a1"], a["a2"], a["a3"], a["a4"], a["a5"], a["a6"]]
    z = np.zeros(na * ndim)
    Mv = np.ones(na * ndim)
    dU = 1e-3 * rng.normal(size=na * ndim)
    bad = dU.copy()
    bad[(na // 2) * ndim] = np.nan

    def evaluate(v):
        if not on_device:
            return S.lagrangian_evaluation(v, z, z, Mv, alpha, [0.0, 0.0, 0.0], None, step, 1.0, None)
        dev = [torch.from_numpy(np.ascontiguousarray(q)).cuda() for q in (v, z, z, Mv)]
        out = torch.zeros(na * ndim, dtype=torch.float64, device="cuda")
        S.lagrangian_evaluation(dev[0], dev[1], dev[2], dev[3], alpha, [0.0, 0.0, 0.0], None, step, 1.0, None, out=out)
        return out.cpu().numpy()

    R = evaluate(dU)  # a finite increment first: no flag
    assert np.all(np.isfinite(R)) and S.status_flags() == 0
    with pytest.raises(n.NlpsError):
        evaluate(bad)
    assert S.status_flags() & 8  # (the flags stay: the reference exits at this point, Constitutive.c:18-258)
