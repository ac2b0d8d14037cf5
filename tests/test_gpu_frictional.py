"""Matsuoka-Nakai and Lade-Duncan (Matsuoka-Nakai.c:300-700, Lade-Duncan.c:290-692) behind the same
`Stress_integration` switch, HIP path against the oracle through the C-ABI.

The particles start from compressive stress states inside the yield surface, most of them close to it
(`synth.frictional_states`), and take small strain increments, so that part of the cloud goes through the elastic
branch and part through the monolithic Newton iteration with its line search.  From far outside the surface the
iteration as written upstream does not converge (tests/test_oracle.py::test_frictional_restatement); both
implementations then still walk the same path, but that is not what these tests lean on."""
import numpy as np
import pytest

from util import assert_close, dirichlet_plane, gpu_setup, nlps, oracle_setup, orc, synth
from test_gpu_parity import masks, small_case

pytestmark = pytest.mark.gpu

TOL_RR, ITS_RR = 1e-10, 20  # what upstream's Matsuoka-Nakai reader sets (InOutFun/Material/Plasticity/Matsuoka-Nakai.c:82-83)


def frictional_case(ndim, mat, velocity=None, seed=5):
    case = small_case(ndim, material=mat, velocity=velocity)
    cl = case["cloud"]
    npart = cl["x"].shape[0]
    cl["b_e_n"] = synth.frictional_states(ndim, mat, npart, seed=seed)
    cl["kappa_n"][:] = mat["kappa_0"]  # Generate-One-Phase-Analysis.c:620-626
    cl["eps_n"][:] = mat["eps_0"]
    return case


def setups(case, nsteps):
    n = nlps()
    M, P, prm, mats = oracle_setup(case)
    prm.tol_radial_returning, prm.max_iter_radial_returning = TOL_RR, ITS_RR
    gp = n.default_params()
    gp.tol_radial_returning, gp.max_iter_radial_returning = TOL_RR, ITS_RR
    S = gpu_setup(case, nsteps=nsteps, params=gp)
    return M, P, prm, mats, S


@pytest.mark.parametrize("ndim", [2, 3])
@pytest.mark.parametrize("lade_duncan", [False, True])
def test_frictional_law_level_b(ndim, lade_duncan):
    """Stress, energy, b_e, kappa, plastic strain, C_ep, internal forces and the tangent after one level-B pass."""
    o = orc()
    mat = synth.matsuoka_nakai_material(lade_duncan)
    case = frictional_case(ndim, mat)
    npart = case["cloud"]["x"].shape[0]
    nsteps = 4
    bcs_list = [dirichlet_plane(case, ndim - 1, 2, nsteps)]
    M, P, prm, mats, S = setups(case, nsteps)
    n2m, d2m, na = masks(S, M, bcs_list, 0, nsteps)
    rng = np.random.default_rng(11)
    dU = 1.5e-4 * rng.normal(size=na * ndim)
    assert o.compatibility(dU, None, P, M, n2m) == 0 and o.constitutive(P, mats, prm) == 0
    S.local_compatibility_conditions(dU)
    S.constitutive_update()
    plastic = (P["eps_n1"] != mat["eps_0"]) | (P["kappa_n1"] != mat["kappa_0"])
    assert npart // 10 < plastic.sum() < npart - npart // 10, f"both branches must be taken ({plastic.sum()} of {npart})"
    st = S.download_state()
    for k, ok in (("Stress", "stress"), ("W", "W"), ("b_e_n1", "b_e_n1"), ("Kappa_n1", "kappa_n1"), ("EPS_n1", "eps_n1"),
                  ("C_ep", "C_ep")):
        assert_close(st[k], P[ok], 1e-8, f"{'Lade-Duncan' if lade_duncan else 'Matsuoka-Nakai'} {k}")
    R_o, s = o.internal_forces(P, M, n2m, d2m, na)
    assert_close(S.nodal_internal_forces(np.zeros(na * ndim)), R_o, 1e-8, "internal forces")
    if ndim == 2:
        # The spectral tangent (Elastoplastic-Tangent-Matrix.c:42-163) works in the eigenbasis of b_e,n+1.  After an
        # elastic step upstream leaves b_e = sum_A 1 * n_A (x) n_A: in 2-D the unit matrix exactly (both eigen-solvers
        # return the coordinate axes), in 3-D the unit matrix plus rounding noise, whose eigenvectors are that noise's
        # and differ between any two solvers, so there is nothing to compare there
        K_o, pat_o, stt = o.tangent_matrix(P, M, mats, n2m, d2m, na)
        assert stt == 0
        rows, cols, vals = S.jacobian_evaluation(0.0, None, True)
        K_g = np.zeros_like(K_o)
        np.add.at(K_g, (rows, cols), vals)
        assert_close(K_g, K_o, 1e-7, "tangent matrix")
    # the roll and a second pass from the rolled state (b_e = 1 where the first step was elastic, as upstream)
    o.roll_state(P)
    S.update_particles_internal_variables()
    dU2 = 1.0e-4 * rng.normal(size=na * ndim)
    assert o.compatibility(dU2, None, P, M, n2m) == 0 and o.constitutive(P, mats, prm) == 0
    S.local_compatibility_conditions(dU2)
    S.constitutive_update()
    st = S.download_state()
    for k, ok in (("Stress", "stress"), ("b_e_n1", "b_e_n1"), ("Kappa_n1", "kappa_n1"), ("EPS_n1", "eps_n1")):
        assert_close(st[k], P[ok], 1e-8, f"second pass {k}")


@pytest.mark.parametrize("ndim", [2, 3])
@pytest.mark.parametrize("lade_duncan", [False, True])
def test_frictional_law_explicit_steps(ndim, lade_duncan):
    """Fused explicit steps (K3 compiled for the one law) from the pre-stressed cloud."""
    o = orc()
    n = nlps()
    mat = synth.matsuoka_nakai_material(lade_duncan)
    vel = [0.0, -0.02] if ndim == 2 else [0.0, 0.0, -0.02]
    case = frictional_case(ndim, mat, velocity=vel)
    nsteps = 6
    bcs_list = [dirichlet_plane(case, ndim - 1, 2, nsteps)]
    grav = [0.0] * (ndim - 1) + [-9.81]
    M, P, prm, mats, S = setups(case, nsteps)
    stepper = o.ExplicitStepper(P, M, mats, prm, o.BccSet(bcs_list), nsteps, gravity=grav)
    gb = n.BccSet(bcs_list)
    dt = 0.002  # small strain increments per step (see the module docstring)
    for t in range(nsteps):
        assert stepper.step(t, dt) == 0
        S.explicit_step(gb, t, dt, 0.5, grav)
    st = S.download_state()
    assert np.array_equal(st["I0"], P["I0"])
    assert (P["eps_n"] != mat["eps_0"]).sum() > 0, "some particles must have yielded"
    for k, ok in (("x", "x"), ("vel", "vel"), ("F_n", "F_n"), ("Stress", "stress"), ("W", "W"), ("b_e_n", "b_e_n"),
                  ("Kappa_n", "kappa_n"), ("EPS_n", "eps_n"), ("rho", "rho")):
        assert_close(st[k], P[ok], 1e-8, f"explicit steps: {k}")


@pytest.mark.parametrize("ndim", [2, 3])
@pytest.mark.parametrize("deterministic", [False, True])
def test_frictional_laws_in_a_mixed_cloud(ndim, deterministic):
    """Neo-Hookean, Matsuoka-Nakai and Lade-Duncan particles interleaved: the two frictional laws share one kernel law
    (the surface is a material constant), the dispatch kernel does not hold them, so the cloud runs one launch per
    law; also in the deterministic mode (one slab per tile and law)."""
    o = orc()
    n = nlps()
    mn, ld = synth.matsuoka_nakai_material(False), synth.matsuoka_nakai_material(True)
    soft_nh = {"type": 0, "E": 1.0e4, "nu": 0.3}
    vel = [0.0, -0.02] if ndim == 2 else [0.0, 0.0, -0.02]
    case = frictional_case(ndim, mn, velocity=vel)
    cl = case["cloud"]
    npart = cl["x"].shape[0]
    case["materials"] = [soft_nh, mn, ld]
    cl["matidx"] = (np.arange(npart) % 3).astype(np.int32)
    cl["b_e_n"][cl["matidx"] == 2] = synth.frictional_states(ndim, ld, int((cl["matidx"] == 2).sum()), seed=9)
    nsteps = 6
    bcs_list = [dirichlet_plane(case, ndim - 1, 2, nsteps)]
    grav = [0.0] * (ndim - 1) + [-9.81]
    M, P, prm, mats, S = setups(case, nsteps)
    with pytest.raises(n.NlpsError):
        S.set_law_launch_mode(2)
    S.set_deterministic(deterministic)
    stepper = o.ExplicitStepper(P, M, mats, prm, o.BccSet(bcs_list), nsteps, gravity=grav)
    gb = n.BccSet(bcs_list)
    dt = 0.002
    for t in range(nsteps):
        assert stepper.step(t, dt) == 0
        S.explicit_step(gb, t, dt, 0.5, grav)
    st = S.download_state()
    assert np.array_equal(st["I0"], P["I0"])
    for k, ok in (("x", "x"), ("vel", "vel"), ("F_n", "F_n"), ("Stress", "stress"), ("b_e_n", "b_e_n"),
                  ("Kappa_n", "kappa_n"), ("EPS_n", "eps_n")):
        assert_close(st[k], P[ok], 1e-8, f"mixed cloud: {k}")
    for m in (1, 2):
        assert (P["eps_n"][cl["matidx"] == m] != mn["eps_0"]).sum() > 0, "both frictional laws must have yielded"
