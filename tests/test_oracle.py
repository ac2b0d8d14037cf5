"""CPU tests of the oracle itself: analytic invariants, known answers, and the third-party (LAPACK)
arithmetic it restates — cross-checked against scipy's LAPACK, the same routines the reference calls
(dsyev: nl-partsol/src/Matlib/TensorLib.c:208, Constitutive/Plasticity/Drucker-Prager.c:635;
dgetrf/dgetri: Matlib/MatrixOp.c:341,359).  The reference itself cannot be built here."""
import os

import numpy as np
import pytest
from scipy.linalg import lapack

from util import DP, HENCKY, NH, make_case, oracle_setup, orc, synth


@pytest.mark.parametrize("ndim", [2, 3])
def test_mesh_rings_and_h_avg(ndim):
    o = orc()
    n = [9, 8] if ndim == 2 else [8, 7, 7]
    M = o.OracleMesh(ndim, n, [0.5] * ndim, 0.25)
    n3 = n + [1] * (3 - ndim)
    I = 3 + n3[0] * (3 + n3[1] * (3 if ndim == 3 else 0))
    r1, r2 = M.ring(1, I), M.ring(2, I)
    assert len(r1) == 3 ** ndim and len(set(r1)) == len(r1) and I in r1
    assert len(r2) == 5 ** ndim and len(set(r2)) == len(r2) and set(r1) <= set(r2)
    # mean distance to the 3^d-1 one-ring neighbours (Read_GramsBox.c:460-507): 1.2071 h / 1.4164 h
    expect = (4 + 4 * np.sqrt(2)) / 8 if ndim == 2 else (6 + 12 * np.sqrt(2) + 8 * np.sqrt(3)) / 26
    assert abs(M.h_avg()[I] / 0.25 - expect) < 1e-12
    corner = M.ring(1, 0)
    assert len(corner) == 2 ** ndim


@pytest.mark.parametrize("ndim", [2, 3])
def test_host_tables_match_oracle_rings(ndim):
    """The product's host-side stencil tables (csrc/nlps_tables.hpp) reproduce the chain order of the
    oracle's list-built NodalLocality_0 / NodalLocality for every boundary class."""
    from util import nlps
    o = orc()
    rank1, order2, count2, h1 = nlps().host_stencil_tables(ndim)
    n = [7, 6] if ndim == 2 else [6, 7, 6]
    n3 = n + [1] * (3 - ndim)
    M = o.OracleMesh(ndim, n, [0.0] * ndim, 1.0)

    def c3(i, m):
        return 0 if i == 0 else (2 if i == m - 1 else 1)

    def c5(i, m):
        return i if i < 2 else (4 - (m - 1 - i) if i > m - 3 else 2)

    for I in range(M.nnodes):
        ijk = [I % n3[0], (I // n3[0]) % n3[1], I // (n3[0] * n3[1])]
        cls3 = sum((c3(ijk[a], n3[a]) if a < ndim else 1) * 3 ** a for a in range(3))
        cls5 = sum((c5(ijk[a], n3[a]) if a < ndim else 2) * 5 ** a for a in range(3))
        ring1 = M.ring(1, I)
        for q, J in enumerate(ring1):
            d = [J % n3[0] - ijk[0], (J // n3[0]) % n3[1] - ijk[1], J // (n3[0] * n3[1]) - ijk[2]]
            assert rank1[cls3][(d[0] + 1) + 3 * (d[1] + 1) + 9 * (d[2] + 1)] == q
        assert np.sum(rank1[cls3] != 255) == len(ring1)
        ring2 = M.ring(2, I)
        assert count2[cls5] == len(ring2)
        for q, J in enumerate(ring2):
            d = [J % n3[0] - ijk[0], (J // n3[0]) % n3[1] - ijk[1], J // (n3[0] * n3[1]) - ijk[2]]
            assert order2[cls5][q] == (d[0] + 2) + 5 * (d[1] + 2) + 25 * ((d[2] + 2) if ndim == 3 else 0)
        assert abs(h1[cls3] - M.h_avg()[I]) < 1e-14


@pytest.mark.parametrize("ndim", [2, 3])
def test_lme_invariants(ndim):
    """sum p = 1, sum p l = 0 (the Newton residual, LME.c:296-302), sum grad p = 0, sum l (x) grad p = -I."""
    o = orc()
    case = make_case(ndim, [12, 12] if ndim == 2 else [9, 9, 9], [3] * ndim, [6] * ndim if ndim == 2 else [3] * ndim)
    M, P, prm, mats = oracle_setup(case)
    X = M.coords()
    nn = P["nn"]
    if ndim == 2:
        assert nn.min() >= 16 and nn.max() <= 25
    else:
        assert nn.min() >= 60 and nn.max() <= 125
    for p in range(0, P.np, max(1, P.np // 40)):
        lst = P.lists(p)
        N = o.compute_N(P, M, p)
        dN = o.compute_dN(P, M, p)
        l = P["x"][p][None, :] - X[lst]
        assert abs(N.sum() - 1.0) < 1e-14
        assert np.linalg.norm(N @ l) <= 1e-10
        assert np.abs(dN.sum(0)).max() < 1e-9
        assert np.abs(l.T @ dN + np.eye(ndim)).max() < 1e-8
        assert np.all(N > 0)


@pytest.mark.parametrize("ndim", [2, 3])
def test_search_after_motion_keeps_invariants(ndim):
    o = orc()
    case = make_case(ndim, [12, 12] if ndim == 2 else [9, 9, 9], [3] * ndim, [6] * ndim if ndim == 2 else [3] * ndim)
    M, P, prm, mats = oracle_setup(case)
    rng = np.random.default_rng(3)
    dx = 0.45 * rng.uniform(-1, 1, size=P["x"].shape)
    P["x"][:] += dx
    P["dis"][:] += dx
    assert o.local_search(P, M, prm) == 0
    X = M.coords()
    for p in range(0, P.np, max(1, P.np // 25)):
        # I0 is the closest node of the 1-ring of the previous I0 (moved < 1 cell => global closest)
        d = np.linalg.norm(X - P["x"][p], axis=1)
        assert d[P["I0"][p]] == d.min()
        lst = P.lists(p)
        assert len(set(lst)) == len(lst) and np.all(M.active()[lst] == 1)
        N = o.compute_N(P, M, p)
        assert abs(N.sum() - 1) < 1e-14 and np.linalg.norm(N @ (P["x"][p] - X[lst])) <= 1e-10


def test_masks_and_conservation():
    o = orc()
    case = make_case(2, [12, 10], [3, 3], [6, 4], velocity=[2.0, -1.0])
    M, P, prm, mats = oracle_setup(case)
    n2m, na = o.active_nodes(M)
    act = M.active().astype(bool)
    assert na == act.sum() and np.array_equal(n2m[act], np.arange(na)) and np.all(n2m[~act] == -1)
    nsteps = 2
    nodes = synth.plane_nodes(case["grid_n"], 1, 3)
    bcs = o.BccSet([{"nodes": nodes, "dim": 2, "dir": np.array([[1, 1], [0, 1]], dtype=np.int32),
                     "value": np.zeros((2, nsteps))}])
    d2m0, nf0 = o.active_dofs(n2m, na, 2, bcs, 0, nsteps)
    d2m1, nf1 = o.active_dofs(n2m, na, 2, bcs, 1, nsteps)
    nfix = int((n2m[nodes] >= 0).sum())
    assert nf0 == 2 * na - nfix and nf1 == 2 * na - 2 * nfix
    assert np.array_equal(d2m0[d2m0 >= 0], np.arange(nf0))
    Mv = o.lumped_mass(P, M, n2m, na)
    assert abs(Mv.reshape(-1, 2)[:, 0].sum() / P["mass"].sum() - 1) < 1e-13
    free = o.BccSet([])
    d2m, _ = o.active_dofs(n2m, na, 2, free, 0, 1)
    V, A = o.nodal_field_n(Mv, P, M, n2m, d2m, na)
    mom = (Mv * V).reshape(-1, 2).sum(0)
    assert np.allclose(mom, P["mass"].sum() * np.array([2.0, -1.0]), rtol=1e-12)


@pytest.mark.parametrize("n", [2, 3])
def test_sym_eigen_matches_lapack_dsyev(n):
    o = orc()
    rng = np.random.default_rng(11)
    for trial in range(200):
        A = rng.normal(size=(n, n))
        A = A @ A.T + (0.0 if trial % 5 else 1.0) * np.eye(n)
        if trial % 7 == 0:
            A = np.diag(np.diag(A))          # repeated / already diagonal cases
        if trial % 11 == 0:
            A = np.eye(n) * 1.37             # fully degenerate (b = F F^T at rest)
        st, w, v = o.sym_eigen(A)
        w_ref, v_ref, info = lapack.dsyev(A, lower=0)
        assert st == 0 and info == 0
        assert np.allclose(w, w_ref, rtol=1e-13, atol=1e-13 * abs(w_ref).max())
        assert np.allclose(v @ np.diag(w) @ v.T, A, atol=1e-12 * abs(A).max())
        assert np.allclose(v.T @ v, np.eye(n), atol=1e-13)


def test_lapack_2x2_eigenvector_matrix_is_symmetric():
    """SURVEY.md §7 hard part 3: the reference's D-P plastic branches index eigenvectors row-wise
    (Drucker-Prager.c:957,1059) and the others column-wise (:699,770); in the only buildable (2-D)
    reference both agree because dsyev's 2x2 eigenvector matrix is symmetric for the SPD matrices the
    law feeds it (b_e^tr = DF b_e DF^T; it is NOT for indefinite input).  Pinned on LAPACK."""
    rng = np.random.default_rng(5)
    for _ in range(500):
        F = np.eye(2) + 0.3 * rng.normal(size=(2, 2))
        A = F @ F.T
        w, v, info = lapack.dsyev(A, lower=0)
        assert info == 0 and abs(v[0, 1] - v[1, 0]) < 1e-15


@pytest.mark.parametrize("n", [2, 3])
def test_inverse_matches_lapack(n):
    o = orc()
    rng = np.random.default_rng(2)
    for _ in range(100):
        A = rng.normal(size=(n, n)) + 2 * np.eye(n)
        st, inv = o.inverse(A)
        lu, piv, info = lapack.dgetrf(A)
        ref, info2 = lapack.dgetri(lu, piv)
        assert st == 0 and info == 0 and info2 == 0
        assert np.allclose(inv, ref, rtol=1e-12, atol=1e-13)


def test_rcond_as_the_reference_evaluates_it():
    """rcond__TensorLib__ hands the unfactored matrix to dgecon (TensorLib.c:966-990); SURVEY.md §7
    measured 0.25 (true 1-norm rcond 0.3125... is 5/16) for [[2,1],[1,3]] on the real reference objects."""
    o = orc()
    assert abs(o.rcond_ref(np.array([[2.0, 1.0], [1.0, 3.0]])) - 0.25) < 1e-15
    A = np.array([[2.0, 1.0], [1.0, 3.0]])
    lu = np.array([[2.0, 1.0], [2.0, 4.0]])  # L_A U_A
    anorm = np.abs(A).sum(0).max()
    assert abs(o.rcond_ref(A) - 1.0 / (anorm * np.abs(np.linalg.inv(lu)).sum(0).max())) < 1e-15
    # same quantity through LAPACK's own dgecon fed with the unfactored matrix
    rc, info = lapack.dgecon(A, anorm, norm="1")
    assert info == 0 and abs(rc - o.rcond_ref(A)) < 1e-14


@pytest.mark.parametrize("ndim", [2, 3])
def test_neo_hookean_known_answers(ndim):
    o = orc()
    mat = o.make_materials([NH])[0]
    prm = o.default_params()
    T = 5 if ndim == 2 else 9
    I = synth.identity_rows(1, ndim)[0]
    st, tau, W, *_ = o.stress_one(ndim, mat, prm, I, I, 1.0, I, 0, 0)
    assert st == 0 and np.all(tau == 0.0) and W == 0.0
    a = 1.1
    F = I.copy()
    F[0] = a
    J = a
    G = NH["E"] / (2 * (1 + NH["nu"]))
    lam = NH["nu"] * NH["E"] / ((1 - 2 * NH["nu"]) * (1 + NH["nu"]))
    st, tau, W, *_ = o.stress_one(ndim, mat, prm, F, F, J, I, 0, 0)
    c0 = 0.5 * lam * (J * J - 1)
    assert abs(tau[0] - (c0 + G * (a * a - 1))) < 1e-9 * abs(tau[0])
    assert abs(tau[ndim + 1] - c0) < 1e-9 * abs(c0)
    assert abs(tau[T - 1] - c0) < 1e-9 * abs(c0)
    Wexp = 0.25 * lam * (J * J - 1) - 0.5 * lam * np.log(J) - G * np.log(J) + 0.5 * G * (a * a - 1)
    assert abs(W - Wexp) < 1e-9 * abs(Wexp)


@pytest.mark.parametrize("ndim", [2, 3])
def test_hencky_known_answers(ndim):
    o = orc()
    mat = o.make_materials([HENCKY])[0]
    prm = o.default_params()
    T = 5 if ndim == 2 else 9
    th = 0.3
    R = np.eye(ndim)
    R[:2, :2] = [[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]
    U = np.diag([1.2, 0.9, 1.05][:ndim])
    Fm = R @ U
    F = synth.identity_rows(1, ndim)[0]
    F[: ndim * ndim] = Fm.ravel()
    st, tau, W, *_ = o.stress_one(ndim, mat, prm, F, F, np.linalg.det(Fm), F, 0, 0)
    assert st == 0
    E, nu = HENCKY["E"], HENCKY["nu"]
    lam = E * nu / ((1 + nu) * (1 - 2 * nu))
    G = E / (2 * (1 + nu))
    eps = np.log(np.diag(U))                      # principal Hencky strains of b = F F^T
    eps3 = np.concatenate([eps, [0.0] * (3 - ndim)])
    tp = 2 * G * eps3 + lam * eps3.sum()
    expect = R @ np.diag(tp[:ndim]) @ R.T
    assert np.allclose(tau[: ndim * ndim].reshape(ndim, ndim), expect, rtol=1e-10, atol=1e-6)
    if ndim == 2:
        assert abs(tau[4] - tp[2]) < 1e-6
    assert abs(W - 0.5 * (tp * eps3).sum()) < 1e-6 * abs(W)


@pytest.mark.parametrize("ndim", [2, 3])
def test_drucker_prager_oedometric_path(ndim):
    """The strain-driven path of the reference's own constitutive driver
    (nl-partsol/tests/Constitutive/Drucker-Prager-Backward-Euler.c:377-388, 390-544: 50 oedometric steps
    dF_yy = 0.999, E=1e4, nu=0.2, kappa0=40, phi=39, psi=6, p_ref=-20).  That driver holds no expected
    values; checked here: elastic start, plastic loading later, yield consistency after
    every return, monotone hardening, symmetric stress."""
    o = orc()
    mat = o.make_materials([DP])[0]
    prm = o.default_params()
    T = 5 if ndim == 2 else 9
    be = synth.identity_rows(1, ndim)[0]
    F = be.copy()
    kappa, eps = DP["kappa_0"], 0.0
    DF = be.copy()
    DF[ndim + 1] = 0.999
    K = DP["E"] / (3 * (1 - 2 * DP["nu"]))
    plastic_steps = 0
    eps_hist = []
    for step in range(50):
        Fm = (DF[: ndim * ndim].reshape(ndim, ndim) @ F[: ndim * ndim].reshape(ndim, ndim))
        F[: ndim * ndim] = Fm.ravel()
        st, tau, W, be1, k1, e1 = o.stress_one(ndim, mat, prm, F, DF, np.linalg.det(Fm), be, kappa, eps)
        assert st == 0
        tm = tau[: ndim * ndim].reshape(ndim, ndim)
        assert np.allclose(tm, tm.T, atol=1e-9 * max(1.0, abs(tm).max()))
        assert e1 >= eps - 1e-15 and k1 >= kappa - 1e-12
        if e1 > eps:
            plastic_steps += 1
        be, kappa, eps = be1, k1, e1
        eps_hist.append(eps)
    assert eps_hist[0] == 0.0 and plastic_steps > 10
    assert np.all(np.isfinite(tau)) and kappa > DP["kappa_0"]


def test_ugimp_shape_functions_config1_plumbing():
    """BASELINE configs[0] names uGIMP; it is unusable upstream, so only S/dS/N are
    restated (Nodes/GIMP.c:235-295) and checked analytically: partition of unity, gradient sum zero,
    gradient = finite difference, linear completeness."""
    o = orc()
    L, lp = 1.0, np.array([0.25, 0.25])
    rng = np.random.default_rng(8)
    gx, gy = np.meshgrid(np.arange(-3, 4), np.arange(-3, 4), indexing="ij")
    nodes = np.stack([gx.ravel(), gy.ravel()], axis=1).astype(float)
    for _ in range(50):
        xp = rng.uniform(-0.5, 0.5, size=2)
        D = xp[None, :] - nodes
        S = o.N_gimp(D, lp, L)
        dS = o.dN_gimp(D, lp, L)
        assert abs(S.sum() - 1) < 1e-14 and np.all(S >= 0)
        assert np.abs(dS.sum(0)).max() < 1e-13
        assert np.abs(S @ nodes - xp).max() < 1e-13                      # linear completeness
        assert np.abs(nodes.T @ dS - np.eye(2)).max() < 1e-12
        e = 1e-6
        for j in range(2):
            dp = np.zeros(2)
            dp[j] = e
            fd = (o.N_gimp(D + dp, lp, L) - o.N_gimp(D - dp, lp, L)) / (2 * e)
            assert np.abs(fd - dS[:, j]).max() < 1e-8


def test_config1_2d_10k_particles_cpu_path():
    """BASELINE configs[0] shape: 2-D bar, 10 000 particles (50x50 cells x 4), explicit step, CPU path.
    uGIMP/linear-elastic have no runnable reference equivalent: LME + Hencky stand in.
    Left edge fixed, right-moving initial velocity; checks the plumbing end to end on the oracle."""
    o = orc()
    case = make_case(2, [60, 60], [5, 5], [50, 50], material={"type": 1, "E": 1.0e7, "nu": 0.3},
                     velocity=[1.0, 0.0])
    assert case["cloud"]["x"].shape[0] == 10000
    M, P, prm, mats = oracle_setup(case)
    nsteps = 3
    nodes = synth.plane_nodes(case["grid_n"], 0, 5)
    bcs = o.BccSet([{"nodes": nodes, "dim": 2, "dir": np.ones((2, nsteps), dtype=np.int32),
                     "value": np.zeros((2, nsteps))}])
    st = o.ExplicitStepper(P, M, mats, prm, bcs, nsteps)
    dt = 0.1 * 1.0 / 100.0
    m0 = P["mass"].sum()
    for t in range(nsteps):
        assert st.step(t, dt) == 0
        assert abs(st.nodal("mass").reshape(-1, 2)[:, 0].sum() / m0 - 1) < 1e-12
    assert np.all(P["J_n"] > 0) and np.isfinite(P["stress"]).all()
    assert P["x"][:, 0].mean() > case["cloud"]["x"][:, 0].mean()


@pytest.mark.parametrize("ndim", [2, 3])
def test_tangent_matrix_is_the_derivative_of_the_internal_forces(ndim):
    """SURVEY §8f n1.  The restated __jacobian_evaluation (Neo-Hookean stiffness density, Neo-Hookean.c:89-141) has
    no reference fixture; what pins it is calculus: column k of the matrix equals the central finite difference of
    the internal-force vector (compatibility -> constitutive -> __nodal_internal_forces) with respect to dU_k.
    Also: symmetry, the sparsity pattern against the lists, the Dirichlet identity rows."""
    o = orc()
    if ndim == 2:
        case = make_case(2, [9, 8], [3, 3], [3, 2])
    else:
        case = make_case(3, [7, 7, 7], [3, 3, 3], [1, 1, 1])
    M, P, prm, mats = oracle_setup(case)
    n2m, na = o.active_nodes(M)
    ntot = na * ndim
    free = np.arange(ntot, dtype=np.int32)
    rng = np.random.default_rng(1)
    dU = 2e-2 * rng.normal(size=ntot)

    def forces(u):
        assert o.compatibility(u, None, P, M, n2m) == 0 and o.constitutive(P, mats, prm) == 0
        R, st = o.internal_forces(P, M, n2m, free, na)
        assert st == 0
        return R

    forces(dU)
    K, pat, st = o.tangent_matrix(P, M, mats, n2m, None, na)
    assert st == 0
    assert np.abs(K - K.T).max() <= 1e-13 * np.abs(K).max()
    eps = 1e-6
    for k in rng.choice(ntot, size=8, replace=False):
        e = np.zeros(ntot)
        e[k] = eps
        fd = (forces(dU + e) - forces(dU - e)) / (2 * eps)
        assert np.abs(fd - K[:, k]).max() <= 1e-8 * np.abs(K).max(), f"column {k}"
    # pattern: a dof row sees ndim columns per node that shares a particle with its node
    share = np.zeros((na, na), dtype=bool)
    for p in range(P.np):
        ids = n2m[P["list"][p, :P["nn"][p]]]
        share[np.ix_(ids, ids)] = True
    assert np.array_equal(pat, np.repeat(share.sum(axis=1) * ndim, ndim))
    # Dirichlet dofs: identity rows and columns, the rest untouched
    forces(dU)
    d2m = free.copy()
    d2m[[0, ntot - 1]] = -1
    Kd, _, _ = o.tangent_matrix(P, M, mats, n2m, d2m, na, 3.0, np.full(ntot, 2.0))
    for d in (0, ntot - 1):
        row = np.zeros(ntot)
        row[d] = 1.0
        assert np.array_equal(Kd[d], row) and np.array_equal(Kd[:, d], row)
    inner = np.ix_(np.arange(1, ntot - 1), np.arange(1, ntot - 1))
    assert np.abs(Kd[inner] - (K + 6.0 * np.eye(ntot))[inner]).max() <= 1e-13 * np.abs(K).max(), "mass diagonal"


def test_trial_b_e_against_the_reference_test_vector():
    """The one fixture the reference's tests hold on this path: tests/Constitutive/test.py checks the index
    convention of the trial elastic left Cauchy-Green tensor b_tr = d_phi b_e d_phi^T (Drucker-Prager.c:617-633) on
    d_phi = [10, 20, 60, 40], b_e = [157, 671, 671, 1561] (integers: the product is exact)."""
    o = orc()
    d_phi = np.array([10.0, 20.0, 60.0, 40.0])
    b_e = np.array([157.0, 671.0, 671.0, 1561.0])
    expected = np.einsum("ik,kl,lj", d_phi.reshape(2, 2), b_e.reshape(2, 2), d_phi.reshape(2, 2).T)
    assert np.array_equal(o.trial_b_e(d_phi, b_e, 2), expected)
    # and in 3-D against the same einsum
    rng = np.random.default_rng(2)
    F, B = rng.normal(size=(3, 3)), rng.normal(size=(3, 3))
    assert np.abs(o.trial_b_e(F.ravel(), B.ravel(), 3) - F @ B @ F.T).max() < 1e-13


def test_spectral_tangent_against_the_reference_python_check():
    """tests/golden/ref_etm2d.npz holds the inputs and the A_ep of the reference's own numpy check of its
    elastoplastic tangent (tests/Constitutive/Elastoplastic-Tangent-Matrix.py, run unmodified by
    tests/golden/make_ref_fixtures.py; the C driver beside it, :85-175, carries the same numbers).  That check
    covers the material part: moduli a_ep in the eigenbasis of b_e plus the (tau_B - tau_A)/(lambda_B - lambda_A)
    terms, with v = D_phi^-T dN_alpha (the pushed-forward gradient) and u = dN_beta.  The maintained routine
    (src/Constitutive/Plasticity/Elastoplastic-Tangent-Matrix.c:42-163) is the same sum written with the scalar
    projections u_A, v_B, followed by the geometric term -tau (dN_beta x dN_alpha): take that term off and the
    oracle must return the reference's matrix."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_etm2d.npz"))
    o = orc()
    Kd = o.stiffness_density_spectral(g["v"], g["u"], g["b_e"], g["a_ep"], g["tau"], 2)
    geometric = -g["tau"] @ np.outer(g["u"], g["v"])
    got = Kd - geometric
    assert np.abs(got - g["A_ep"]).max() <= 1e-12 * np.abs(g["A_ep"]).max()
    # the inputs of the C driver of the same test (Elastoplastic-Tangent-Matrix.c:89-98)
    assert np.allclose(g["v"], np.linalg.inv(g["D_phi"]).T @ g["dN_alpha"], rtol=0, atol=1e-15)
    assert np.array_equal(g["u"], g["dN_beta"])


@pytest.mark.parametrize("ndim", [2, 3])
@pytest.mark.parametrize("law", ["hencky", "drucker-prager"])
def test_spectral_tangent_against_a_second_transcription(ndim, law):
    """The spectral stiffness density of Hencky (Hencky.c:98-229) and of the elastoplastic laws
    (Elastoplastic-Tangent-Matrix.c:42-163) has no reference fixture, and calculus cannot pin it either: as written
    upstream its shear-coupling terms carry a factor 1/2 on the quotient (tau_B - tau_A)/(b_B - b_A), so it is not the
    derivative of the internal force (measured: 2-8 % off at 2 % strain, and it does not linearise to isotropic
    elasticity).  A drop-in has to reproduce it as it is.  This test guards the C restatement against indexing
    slips with a second, independent transcription in numpy (numpy.linalg.eigh = the same LAPACK dsyev) assembled
    from the oracle's own grad N."""
    o = orc()
    mat = HENCKY if law == "hencky" else DP
    case = (make_case(2, [9, 8], [3, 3], [3, 2], material=mat) if ndim == 2 else
            make_case(3, [7, 7, 7], [3, 3, 3], [1, 1, 1], material=mat))
    M, P, prm, mats = oracle_setup(case)
    n2m, na = o.active_nodes(M)
    ntot = na * ndim
    rng = np.random.default_rng(3)
    dU = (2e-2 if law == "hencky" else 8e-3) * rng.normal(size=ntot)
    assert o.compatibility(dU, None, P, M, n2m) == 0 and o.constitutive(P, mats, prm) == 0
    K, _, st = o.tangent_matrix(P, M, mats, n2m, None, na, with_pattern=False)
    assert st == 0
    T = P.T
    K2 = np.zeros((ntot, ntot))
    E, nu = mat["E"], mat["nu"]
    lame, G = E * nu / ((1 + nu) * (1 - 2 * nu)), E / (2 * (1 + nu))
    for p in range(P.np):
        nn = P["nn"][p]
        conn = n2m[P["list"][p, :nn]]
        dN = o.compute_dN(P, M, p)[:nn]
        blk = lambda a: a[p, :ndim * ndim].reshape(ndim, ndim)  # noqa: E731
        DF, tau = blk(P["DF"]), blk(P["stress"])
        g1 = dN @ np.linalg.inv(DF)                      # rows: DF^-T grad N_A
        if law == "hencky":
            F1 = blk(P["F_n1"])
            b = F1 @ F1.T
            Cm = lame * np.ones((ndim, ndim)) + 2 * G * np.eye(ndim)
        else:
            b = blk(P["b_e_n1"])
            Cm = P["C_ep"][p].reshape(ndim, ndim)
        w, V = np.linalg.eigh(b)
        tw = np.linalg.eigvalsh(tau)
        for A in range(nn):
            for B in range(nn):
                u, v = g1[A] @ V, g1[B] @ V              # projections of dN_alpha, dN_beta on the eigenvectors
                Kd = np.zeros((ndim, ndim))
                for a in range(ndim):
                    for c in range(ndim):
                        Kd += Cm[a, c] * u[a] * v[c] * np.outer(V[:, a], V[:, c])
                        if a != c and abs(w[c] - w[a]) > 1e-14:
                            q = 0.5 * (tw[c] - tw[a]) / (w[c] - w[a])
                            Kd += q * (w[c] * u[c] * v[c] * np.outer(V[:, a], V[:, a]) +
                                       w[a] * v[a] * u[c] * np.outer(V[:, a], V[:, c]))
                Kd -= tau @ np.outer(g1[B], g1[A])
                ia, ib = conn[A] * ndim, conn[B] * ndim
                K2[ia:ia + ndim, ib:ib + ndim] += Kd * P["vol0"][p]
    assert np.abs(K - K2).max() <= 1e-9 * np.abs(K2).max()


@pytest.mark.parametrize("ndim", [2, 3])
def test_von_mises_restatement(ndim):
    """Von-Mises.c:212-392 has a driver without expected values upstream (tests/Constitutive/Von-Mises.c); the
    restatement is pinned by what the algorithm promises: the elastic branch is the closed form in principal Hencky
    strains (with upstream's volumetric term K*tr(eps)/3, :495-519), a plastic step lands on the yield surface
    |dev tau - back| = sqrt(2/3) (K_iso + dK_kin), perfect plasticity keeps |dev tau| = sqrt(2/3) sigma_y, the
    back stress moves along the flow direction, b_e,n+1 carries exactly the elastic strain."""
    o = orc()
    rng = np.random.default_rng(0)
    T = 5 if ndim == 2 else 9
    base = synth.von_mises_material()
    prm = o.default_params()
    K, G = base["E"] / (3 * (1 - 2 * base["nu"])), base["E"] / (2 * (1 + base["nu"]))

    def tensors(scale):
        A = scale * rng.normal(size=(ndim, ndim))
        DF = np.eye(ndim) + A
        be = np.eye(ndim) + 0.2 * scale * (A + A.T)
        d, b = np.zeros(T), np.zeros(T)
        d[:ndim * ndim], b[:ndim * ndim] = DF.ravel(), be.ravel()
        if ndim == 2:  # out-of-plane slots
            d[4] = 1.0
            b[4] = 1.01
        return DF, be, d, b

    def principal(DF, be, bzz):
        w, V = np.linalg.eigh(DF @ be @ DF.T)
        eps = 0.5 * np.log(np.append(w, bzz) if ndim == 2 else w)
        return eps, V

    # elastic
    mats = o.make_materials([dict(base, kappa_0=1e9)])
    DF, be, d, b = tensors(0.01)
    back = np.zeros(3)
    st, tau, W, b1, k1, e1 = o.stress_one(ndim, mats[0], prm, d, d, 1.0, b, 0.0, 0.0, back)
    eps, V = principal(DF, be, b[T - 1])
    tp = K * eps.sum() / 3 + 2 * G * (eps - eps.sum() / 3)
    ref = (V * tp[:ndim]) @ V.T
    assert st == 0 and e1 == 0.0 and np.all(back == 0.0)
    assert np.abs(tau[:ndim * ndim].reshape(ndim, ndim) - ref).max() <= 1e-12 * np.abs(ref).max()
    assert abs(W - 0.5 * (tp * eps).sum()) <= 1e-12 * abs(W)
    # plastic, combined hardening
    mats = o.make_materials([base])
    for trial in range(20):
        DF, be, d, b = tensors(0.05)
        back = 2.0 * rng.normal(size=3)
        back -= back.mean()
        back0, eps_n = back.copy(), 0.01 * trial
        st, tau, W, b1, k1, e1 = o.stress_one(ndim, mats[0], prm, d, d, 1.0, b, 0.0, eps_n, back)
        assert st == 0
        eps, V = principal(DF, be, b[T - 1])
        sdev = 2 * G * (eps - eps.sum() / 3) - back0
        J2 = np.linalg.norm(sdev)

        def kap(e):
            return (base["kappa_0"] + base["theta_voce"] * base["hardening_modulus"] * e +
                    (base["Kinf_voce"] - base["K0_voce"]) * (1 - np.exp(-base["delta_voce"] * e)),
                    (1 - base["theta_voce"]) * base["hardening_modulus"] * e)
        kn = kap(eps_n)
        if J2 - np.sqrt(2 / 3) * kn[0] <= 0:
            assert e1 == eps_n and np.array_equal(back, back0)
            continue
        assert e1 > eps_n
        kk = kap(e1)
        dg = (e1 - eps_n) / np.sqrt(2 / 3)
        assert abs(J2 - np.sqrt(2 / 3) * (kk[0] + kk[1] - kn[1]) - 2 * G * dg) <= 1e-9 * J2  # on the yield surface
        nflow = sdev / J2
        assert np.abs(back - back0 - np.sqrt(2 / 3) * (kk[1] - kn[1]) * nflow).max() <= 1e-12 * max(1.0, np.abs(back).max())
        # b_e,n+1 = exp(2 (eps_tr - dgamma n)) in the trial eigenvectors
        e_el = eps - dg * nflow
        bref = (V * np.exp(2 * e_el[:ndim])) @ V.T
        assert np.abs(b1[:ndim * ndim].reshape(ndim, ndim) - bref).max() <= 1e-12 * np.abs(bref).max()
    # perfect plasticity: the deviator stays on the cylinder of radius sqrt(2/3) sigma_y
    mats = o.make_materials([dict(base, hardening_modulus=0.0, Kinf_voce=base["K0_voce"])])
    DF, be, d, b = tensors(0.08)
    back = np.zeros(3)
    st, tau, W, b1, k1, e1 = o.stress_one(ndim, mats[0], prm, d, d, 1.0, b, 0.0, 0.0, back)
    assert st == 0 and e1 > 0
    tm = tau[:ndim * ndim].reshape(ndim, ndim)
    pr = np.append(np.linalg.eigvalsh(tm), tau[4]) if ndim == 2 else np.linalg.eigvalsh(tm)
    assert abs(np.linalg.norm(pr - pr.mean()) - np.sqrt(2 / 3) * base["kappa_0"]) <= 1e-9 * base["kappa_0"]


@pytest.mark.parametrize("ndim", [2, 3])
@pytest.mark.parametrize("law", ["matsuoka-nakai", "lade-duncan"])
def test_frictional_restatement(ndim, law):
    """Matsuoka-Nakai.c:300-700 / Lade-Duncan.c:290-692 come with a driver without expected values
    (tests/Constitutive/Matsuoka_Nakai.c), so the restatement is pinned by what a converged step of the algorithm
    promises: an elastic step is the Hencky closed form and (as written upstream, :410-424 and :694) leaves b_e = 1;
    after a plastic step the principal stresses sit on the yield surface of the new kappa, kappa equals its
    hardening law a1 L exp(a2 I1) exp(-a3 L), and b_e carries exactly the elastic strain C^-1 (tau + c).  States start
    inside the surface and take small strain increments: from far outside, the residual upstream adds to the diagonal
    of its tangent (:505-510) keeps the iteration from converging (measured; without that term every case converges),
    and nothing can be said about such a step but that both implementations walk the same path."""
    o = orc()
    rng = np.random.default_rng(7)
    ld = law == "lade-duncan"
    mat = synth.matsuoka_nakai_material(ld)
    mats = o.make_materials([mat])
    prm = o.default_params()
    prm.tol_radial_returning, prm.max_iter_radial_returning = 1e-10, 20
    T = 5 if ndim == 2 else 9
    E, nu, a = mat["E"], mat["nu"], mat["a_borja"]
    lame, G = E * nu / ((1 + nu) * (1 - 2 * nu)), E / (2 * (1 + nu))
    AA = np.full((3, 3), lame) + 2 * G * np.eye(3)

    def surface(t, kap):
        I1, I2, I3 = t.sum(), t[0] * t[1] + t[1] * t[2] + t[0] * t[2], t.prod()
        return np.cbrt((27 + kap) * I3) - I1 if ld else np.cbrt((9 + kap) * I3) - np.cbrt(I1 * I2)

    n = 300
    states = synth.frictional_states(ndim, mat, n, seed=3)
    nel = npl = nconv = 0
    for p in range(n):
        A = 1.5e-4 * rng.normal(size=(ndim, ndim))
        DF = np.zeros(T)
        DF[:ndim * ndim] = (np.eye(ndim) + A).ravel()
        if ndim == 2:
            DF[4] = 1.0
        be = states[p]
        st, tau, W, b1, k1, e1 = o.stress_one(ndim, mats[0], prm, DF, DF, 1.0, be, mat["kappa_0"], mat["eps_0"])
        assert st == 0 and np.all(np.isfinite(tau)) and np.all(np.isfinite(b1))
        btr = (np.eye(ndim) + A) @ be[:ndim * ndim].reshape(ndim, ndim) @ (np.eye(ndim) + A).T
        w, V = np.linalg.eigh(btr)
        etr = 0.5 * np.log(np.append(w, be[4]) if ndim == 2 else w)
        ttr = AA @ etr
        F0 = surface(ttr, mat["kappa_0"])
        tm = tau[:ndim * ndim].reshape(ndim, ndim)
        tp = np.append(np.linalg.eigvalsh(tm), tau[4]) if ndim == 2 else np.linalg.eigvalsh(tm)
        if F0 <= 1e-5:  # TOL_NR
            nel += 1
            ref = (V * ttr[:ndim]) @ V.T
            assert np.abs(tm - ref).max() <= 1e-11 * np.abs(ref).max()
            assert k1 == mat["kappa_0"] and e1 == mat["eps_0"]
            ident = np.zeros(T)
            ident[[0, 3, 4] if ndim == 2 else [0, 4, 8]] = 1.0
            assert np.abs(b1 - ident).max() <= 1e-14
            assert abs(W - 0.5 * (ttr * etr).sum()) <= 1e-11 * abs(W)
        else:
            npl += 1
            if abs(surface(tp, k1) / F0) > 1e-8:
                continue  # not converged (see above)
            nconv += 1
            assert abs(k1 - a[0] * e1 * np.exp(a[1] * tp.sum()) * np.exp(-a[2] * e1)) <= 1e-8 * max(1.0, abs(k1))
            bm = b1[:ndim * ndim].reshape(ndim, ndim)
            ee = 0.5 * np.log(np.append(np.linalg.eigvalsh(bm), b1[4]) if ndim == 2 else np.linalg.eigvalsh(bm))
            CC = np.linalg.inv(AA)
            assert np.abs(np.sort(ee) - np.sort(CC @ tp)).max() <= 1e-12
            assert e1 >= mat["eps_0"]
    assert nel > n // 10 and npl > n // 10, (nel, npl)
    assert nconv > 0.8 * npl, (nconv, npl)
