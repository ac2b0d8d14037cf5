"""Synthetic structured background grids and particle clouds.

Input generation only (numpy): the same arrays are handed to the HIP path and, in tests, to the
oracle.  Seeding follows the reference's particle generators: 2-D Q4 4/cell at xi = +-1/sqrt(3)
(nl-partsol/src/Nodes/Q4.c:359-368), 3-D H8 8/cell at xi = +-0.5 (Nodes/H8.c:406-431), natural
coordinates in [-1,1], plus a uniform jitter.  Fields are initialised like
Formulations/Displacements/U-Analisys.c:33-43,63-73 (F, b_e = identity incl. the 2-D zz slot).
"""
import numpy as np

MAT_NEO_HOOKEAN, MAT_HENCKY, MAT_DRUCKER_PRAGER = 0, 1, 2


def tensor_width(ndim):
    return 5 if ndim == 2 else 9


def identity_rows(n, ndim):
    T = tensor_width(ndim)
    a = np.zeros((n, T))
    a[:, 0] = 1.0
    a[:, ndim + 1] = 1.0
    a[:, T - 1] = 1.0
    return a


def make_cloud(ndim, grid_cells, block_lo, block_cells, h=1.0, origin=None, jitter=0.05, seed=12345,
               rho=1000.0, ppc=None, velocity=None, matidx=0):
    """Particles in the cell block [block_lo, block_lo+block_cells) of a grid with grid_cells cells.

    Returns a dict of numpy arrays in the reference's AoS layout ([np][ndim], [np][T]).
    """
    ndim = int(ndim)
    origin = np.zeros(ndim) if origin is None else np.asarray(origin, dtype=np.float64)
    block_lo = np.asarray(block_lo, dtype=np.int64)
    block_cells = np.asarray(block_cells, dtype=np.int64)
    grid_cells = np.asarray(grid_cells, dtype=np.int64)
    assert np.all(block_lo >= 0) and np.all(block_lo + block_cells <= grid_cells)
    if ppc is None:
        ppc = 4 if ndim == 2 else 8
    if ndim == 2:
        g = 1.0 / np.sqrt(3.0)
        xi = {1: [[0.0, 0.0]], 4: [[g, g], [g, -g], [-g, g], [-g, -g]]}[ppc]
    else:
        xi = {1: [[0.0, 0.0, 0.0]],
              8: [[-.5, -.5, .5], [.5, -.5, .5], [.5, .5, .5], [-.5, .5, .5],
                  [-.5, -.5, -.5], [.5, -.5, -.5], [.5, .5, -.5], [-.5, .5, -.5]]}[ppc]
    xi = np.asarray(xi)
    axes = [np.arange(block_lo[a], block_lo[a] + block_cells[a]) for a in range(ndim)]
    # cells enumerated x-fastest
    mesh = np.meshgrid(*axes[::-1], indexing="ij")
    cells = np.stack([m.ravel() for m in mesh[::-1]], axis=1).astype(np.float64)  # [ncell, ndim]
    centre = origin + h * (cells + 0.5)
    x = (centre[:, None, :] + 0.5 * h * xi[None, :, :]).reshape(-1, ndim)
    rng = np.random.default_rng(seed)
    if jitter > 0:
        x = x + rng.uniform(-jitter * h, jitter * h, size=x.shape)
    n = x.shape[0]
    vol0 = np.full(n, h ** ndim / ppc)
    vel = np.zeros((n, ndim))
    if velocity is not None:
        vel[:] = np.asarray(velocity, dtype=np.float64)
    return {
        "ndim": ndim, "x": np.ascontiguousarray(x), "dis": np.zeros((n, ndim)), "vel": vel,
        "acc": np.zeros((n, ndim)), "F_n": identity_rows(n, ndim), "b_e_n": identity_rows(n, ndim),
        "J_n": np.ones(n), "rho": np.full(n, rho), "mass": rho * vol0, "vol0": vol0,
        "kappa_n": np.zeros(n), "eps_n": np.zeros(n), "matidx": np.full(n, matidx, dtype=np.int32),
    }


def grid_nodes(grid_cells):
    return [int(c) + 1 for c in grid_cells]


def plane_nodes(n, axis, index):
    """Node ids (x-fastest numbering) of the lattice plane `axis == index`."""
    n3 = list(n) + [1] * (3 - len(n))
    I = np.arange(n3[0] * n3[1] * n3[2])
    ijk = [I % n3[0], (I // n3[0]) % n3[1], I // (n3[0] * n3[1])]
    return I[ijk[axis] == index].astype(np.int32)


def drucker_prager_material(E=1e4, nu=0.2, kappa_0=40.0, phi_deg=39.0, psi_deg=6.0, m=1.0, H=0.1, p_ref=-20.0):
    """Values of nl-partsol/tests/Constitutive/Drucker-Prager-Backward-Euler.c:377-388; eps_0 default
    kappa_0/(m*H) as in nl-partsol/src/InOutFun/Material/Plasticity/Drucker-Prager.c:200-209."""
    return {"type": MAT_DRUCKER_PRAGER, "E": E, "nu": nu, "phi_deg": phi_deg, "psi_deg": psi_deg,
            "kappa_0": kappa_0, "exponent_ortiz": m, "eps_0": (kappa_0 / (m * H)) * 1.0 ** (1.0 / m - 1.0),
            "p_ref": p_ref}


def von_mises_material():
    """J2 plasticity with combined linear isotropic / kinematic hardening plus a Voce term (Von-Mises.c:246-253);
    the numbers are of the order of the reference's tests/Constitutive/Von-Mises.c driver (unit-free)."""
    return {"type": 3, "E": 1.0e4, "nu": 0.3, "kappa_0": 30.0, "hardening_modulus": 400.0, "theta_voce": 0.6,
            "K0_voce": 30.0, "Kinf_voce": 45.0, "delta_voce": 12.0}


def matsuoka_nakai_material(lade_duncan=False):
    """Three-invariant frictional plasticity (Matsuoka-Nakai.c:300-700 / Lade-Duncan.c:290-692); the numbers are the
    active set of the reference's tests/Constitutive/Matsuoka_Nakai.c driver (:38-52: E, nu, a1..a3, alpha, the
    initial kappa and plastic strain), with the friction angle its reader derives from kappa_0 when none is given
    (InOutFun/Material/Plasticity/Matsuoka-Nakai.c:207-211).  Compressive pre-stress comes with the particles
    (`frictional_prestress`): both surfaces need negative principal stresses."""
    kappa_0 = 4.543
    return {"type": 5 if lade_duncan else 4, "E": 1.0e4, "nu": 0.2, "kappa_0": kappa_0, "eps_0": 1.065199,
            "phi_deg": float(np.degrees(np.arcsin(np.sqrt(kappa_0 / (kappa_0 + 8.0))))), "cohesion": 0.0,
            "alpha_borja": 0.162, "a_borja": (10.0, 0.0, 0.8)}


def frictional_states(ndim, mat, n, seed=0, pressure=(10.0, 30.0)):
    """Rows of b_e for n particles under compressive Kirchhoff stresses -p (1, q, r) in random principal axes (the
    driver above confines at -20, :50), with the ratios bounded so that every state lies INSIDE the yield surface of
    the material (Matsuoka-Nakai: I1 I2 / I3 < 9 + kappa_0; Lade-Duncan: I1^3 / I3 < 27 + kappa_0) but many of them
    close to it: a small strain increment then sends part of the cloud through the plastic branch."""
    rng = np.random.default_rng(seed)
    E, nu, k0 = mat["E"], mat["nu"], mat["kappa_0"]
    ld = mat["type"] == 5
    CC = np.full((3, 3), -nu / E)
    np.fill_diagonal(CC, 1.0 / E)
    T = 5 if ndim == 2 else 9
    out = np.zeros((n, T))
    for p in range(n):
        while True:
            t = -rng.uniform(*pressure) * np.array([1.0, rng.uniform(1.0, 1.3), rng.uniform(1.0, 4.0)])
            I1, I2, I3 = t.sum(), t[0] * t[1] + t[1] * t[2] + t[0] * t[2], t.prod()
            ratio = I1 ** 3 / I3 if ld else I1 * I2 / I3
            lim = (27.0 if ld else 9.0) + k0
            if ratio < 0.995 * lim and (p % 3 == 0 or ratio > 0.97 * lim):  # two of three close to the surface
                break
        e = CC @ (t + 0.0)  # no cohesion shift in the materials above
        if ndim == 3:
            Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
            out[p] = (Q @ np.diag(np.exp(2 * e)) @ Q.T).ravel()
        else:
            e = e[rng.permutation(3)]
            th = rng.uniform(0, np.pi)
            Q = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
            out[p, :4] = (Q @ np.diag(np.exp(2 * e[:2])) @ Q.T).ravel()
            out[p, 4] = np.exp(2 * e[2])
    return out
