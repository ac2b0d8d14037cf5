"""ctypes binding of the CPU oracle (oracle/nlps_oracle.c).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package (nl-partsol_amd/).  PARITY UNPINNED: see
oracle/nlps_oracle.h and DESIGN.md.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
MAXNB = 128

MAT_NEO_HOOKEAN, MAT_HENCKY, MAT_DRUCKER_PRAGER = 0, 1, 2

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class Mesh(C.Structure):
    _fields_ = [("ndim", C.c_int), ("n", C.c_int * 3), ("origin", C.c_double * 3), ("h", C.c_double),
                ("nnodes", C.c_int), ("coords", _dp), ("r1_ptr", _ip), ("r1", _ip), ("r2_ptr", _ip),
                ("r2", _ip), ("h_avg", _dp), ("active", C.POINTER(C.c_ubyte))]


_PFIELDS_D = ["x", "dis", "vel", "acc", "d_dis", "F_n", "F_n1", "DF", "stress", "b_e_n", "b_e_n1",
              "dt_F_n", "dt_F_n1", "dt_DF", "J_n", "J_n1", "rho", "mass", "vol0", "W", "kappa_n",
              "kappa_n1", "eps_n", "eps_n1"]


class CParticles(C.Structure):
    _fields_ = ([("np", C.c_int), ("ndim", C.c_int), ("T", C.c_int)] + [(k, _dp) for k in _PFIELDS_D] +
                [("matidx", _ip), ("I0", _ip), ("lambda_", _dp), ("beta", _dp), ("nn", _ip), ("list", _ip),
                 ("status", _ip), ("C_ep", _dp), ("back_stress", _dp)])


class Material(C.Structure):
    _fields_ = [("type", C.c_int), ("E", C.c_double), ("nu", C.c_double), ("phi_deg", C.c_double),
                ("psi_deg", C.c_double), ("kappa_0", C.c_double), ("exponent_ortiz", C.c_double),
                ("eps_0", C.c_double), ("p_ref", C.c_double), ("hardening_modulus", C.c_double),
                ("theta_voce", C.c_double), ("K0_voce", C.c_double), ("Kinf_voce", C.c_double),
                ("delta_voce", C.c_double), ("Ceps", C.c_double), ("Gf", C.c_double),
                ("cohesion", C.c_double), ("alpha_borja", C.c_double), ("a_borja", C.c_double * 3),
                ("ft", C.c_double), ("heps", C.c_double), ("wcrit", C.c_double)]


class Params(C.Structure):
    _fields_ = [("gamma_lme", C.c_double), ("tol_zero_lme", C.c_double), ("tol_wrapper_lme", C.c_double),
                ("max_iter_lme", C.c_int), ("tol_radial_returning", C.c_double),
                ("max_iter_radial_returning", C.c_int)]


class Bcc(C.Structure):
    _fields_ = [("nnodes", C.c_int), ("nodes", _ip), ("dim", C.c_int), ("dir", _ip), ("value", _dp)]


class StepOut(C.Structure):
    _fields_ = [("nactive", C.c_int), ("nodes2mask", _ip), ("dofs2mask", _ip), ("mass", _dp), ("dU", _dp),
                ("force", _dp), ("accel", _dp), ("reaction", _dp)]


def default_params():
    """Defaults of InOutFun/Read_GramsShapeFun.c:100-104 and Globals (TOL_Radial_Returning 1e-14, 10 its)."""
    return Params(3.0, 1e-6, 1e-10, 10, 1e-14, 10)


_FAST = False


def use_fast_build(on=True):
    """Selects libnlps_oracle_fast.so (the reference's -Ofast -fopenmp flags) for this process: timing only
    (bench.py's cpu_baseline leg); must be called before the first lib()."""
    global _FAST
    assert _LIB is None, "use_fast_build() must come before the library is loaded"
    _FAST = bool(on)


def build(force=False):
    src = os.path.join(_HERE, "nlps_oracle.c")
    out = None
    for name in ("libnlps_oracle.so", "libnlps_oracle_fast.so"):
        so = os.path.join(_HERE, name)
        if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
            subprocess.check_call(["make", "-C", _HERE, "-B", name], stdout=subprocess.DEVNULL)
        if name == ("libnlps_oracle_fast.so" if _FAST else "libnlps_oracle.so"):
            out = so
    return out


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_mesh_build.restype = C.POINTER(Mesh)
        L.orc_mesh_build.argtypes = [C.c_int, C.c_int * 3, C.c_double * 3, C.c_double]
        L.orc_mesh_free.argtypes = [C.POINTER(Mesh)]
        L.orc_rcond_ref.restype = C.c_double
        L.orc_rcond_ref.argtypes = [_dp, C.c_int]
        _LIB = L
    return _LIB


def _d(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _i(a):
    return a.ctypes.data_as(_ip) if a is not None else None


class OracleMesh:
    def __init__(self, ndim, n, origin, h):
        self.ndim, self.h = ndim, float(h)
        n3 = list(n) + [1] * (3 - len(n))
        o3 = list(origin) + [0.0] * (3 - len(origin))
        self.n = n3
        self.origin = o3
        self.ptr = lib().orc_mesh_build(ndim, (C.c_int * 3)(*n3), (C.c_double * 3)(*o3), float(h))
        self.m = self.ptr.contents
        self.nnodes = self.m.nnodes

    def coords(self):
        return np.ctypeslib.as_array(self.m.coords, shape=(self.nnodes, self.ndim))

    def h_avg(self):
        return np.ctypeslib.as_array(self.m.h_avg, shape=(self.nnodes,))

    def active(self):
        return np.ctypeslib.as_array(self.m.active, shape=(self.nnodes,))

    def ring(self, which, I):
        ptr = self.m.r1_ptr if which == 1 else self.m.r2_ptr
        arr = self.m.r1 if which == 1 else self.m.r2
        return [arr[q] for q in range(ptr[I], ptr[I + 1])]

    def __del__(self):
        try:
            lib().orc_mesh_free(self.ptr)
        except Exception:
            pass


class OracleParticles:
    """Owns numpy arrays in the reference's AoS layout and the matching C struct."""

    def __init__(self, cloud):
        """cloud: dict from nl-partsol_amd.synth.make_cloud (arrays are COPIED)."""
        self.np = int(cloud["x"].shape[0])
        self.ndim = int(cloud["x"].shape[1])
        self.T = 5 if self.ndim == 2 else 9
        n, d, T = self.np, self.ndim, self.T
        a = {}
        for k in ["x", "dis", "vel", "acc"]:
            a[k] = np.ascontiguousarray(cloud[k], dtype=np.float64).copy()
        a["d_dis"] = np.zeros((n, d))
        for k in ["F_n", "b_e_n"]:
            a[k] = np.ascontiguousarray(cloud[k], dtype=np.float64).copy()
        a["F_n1"] = a["F_n"].copy()
        a["b_e_n1"] = a["b_e_n"].copy()
        a["DF"] = np.zeros((n, T))
        a["DF"][:, 0] = 1.0
        a["DF"][:, d + 1] = 1.0
        a["DF"][:, T - 1] = 1.0
        a["stress"] = np.zeros((n, T))
        for k in ["dt_F_n", "dt_F_n1", "dt_DF"]:
            a[k] = np.zeros((n, T))
        if cloud.get("dt_F_n") is not None:
            a["dt_F_n"][:] = cloud["dt_F_n"]
        for k in ["J_n", "rho", "mass", "vol0", "kappa_n", "eps_n"]:
            a[k] = np.ascontiguousarray(cloud[k], dtype=np.float64).copy()
        a["J_n1"] = a["J_n"].copy()
        a["kappa_n1"] = a["kappa_n"].copy()
        a["eps_n1"] = a["eps_n"].copy()
        a["W"] = np.zeros(n)
        a["matidx"] = np.ascontiguousarray(cloud["matidx"], dtype=np.int32).copy()
        a["I0"] = np.zeros(n, dtype=np.int32)
        a["lambda_"] = np.zeros((n, d))
        a["beta"] = np.zeros(n)
        a["nn"] = np.zeros(n, dtype=np.int32)
        a["list"] = np.full((n, MAXNB), -1, dtype=np.int32)
        a["status"] = np.zeros(n, dtype=np.int32)
        a["C_ep"] = np.zeros((n, d * d))
        a["back_stress"] = np.zeros((n, 3))
        if cloud.get("back_stress") is not None:
            a["back_stress"][:] = cloud["back_stress"]
        for k in ("I0", "lambda_", "beta"):
            src = "lambda" if k == "lambda_" else k
            if src in cloud and cloud[src] is not None:
                a[k][...] = cloud[src]
        self.a = a
        c = CParticles()
        c.np, c.ndim, c.T = n, d, T
        for k in _PFIELDS_D + ["lambda_", "beta", "C_ep", "back_stress"]:
            setattr(c, k, _d(a[k]))
        for k in ["matidx", "I0", "nn", "list", "status"]:
            setattr(c, k, _i(a[k]))
        self.c = c

    def __getitem__(self, k):
        return self.a["lambda_" if k == "lambda" else k]

    def lists(self, p):
        return self.a["list"][p, : self.a["nn"][p]].copy()


def make_materials(mats):
    arr = (Material * len(mats))()
    for i, m in enumerate(mats):
        arr[i] = Material(int(m["type"]), float(m["E"]), float(m["nu"]), float(m.get("phi_deg", 0.0)),
                          float(m.get("psi_deg", 0.0)), float(m.get("kappa_0", 0.0)),
                          float(m.get("exponent_ortiz", 1.0)), float(m.get("eps_0", 1.0)),
                          float(m.get("p_ref", 0.0)), float(m.get("hardening_modulus", 0.0)),
                          float(m.get("theta_voce", 1.0)), float(m.get("K0_voce", 0.0)),
                          float(m.get("Kinf_voce", 0.0)), float(m.get("delta_voce", 0.0)),
                          float(m.get("Ceps", 0.0)), float(m.get("Gf", 0.0)), float(m.get("cohesion", 0.0)),
                               float(m.get("alpha_borja", 0.0)),
                               (C.c_double * 3)(*[float(v) for v in m.get("a_borja", (0.0, 0.0, 0.0))]),
                          float(m.get("ft", 0.0)), float(m.get("heps", 0.0)), float(m.get("wcrit", 1.0)))
    return arr


BEPS_STRIDE = 1024


def compute_beps(P, M, mats, beps=None, initialize=True):
    """compute_Beps__Constitutive__: (beps_n[np], beps[np][BEPS_STRIDE]) in chain order"""
    if beps is None:
        beps = (np.zeros(P.np, dtype=np.int32), np.full((P.np, BEPS_STRIDE), -1, dtype=np.int32))
    st = lib().orc_compute_beps(_i(beps[0]), _i(beps[1]), BEPS_STRIDE, C.byref(P.c), M.ptr, mats, 1 if initialize else 0)
    assert st == 0
    return beps


def constitutive_eroded(P, mats, prm, damage_n):
    return lib().orc_constitutive_eroded(C.byref(P.c), mats, C.byref(prm), _d(damage_n))


def eigenerosion_hook(damage_n1, damage_n, P, mats, beps, DeltaX):
    f = lib().orc_eigenerosion_hook
    f.argtypes = [_dp, _dp, C.POINTER(CParticles), C.POINTER(Material), _ip, _ip, C.c_int, C.c_double]
    return f(_d(damage_n1), _d(damage_n), C.byref(P.c), mats, _i(beps[0]), _i(beps[1]), BEPS_STRIDE, float(DeltaX))


def eigensoftening_hook(damage_n1, damage_n, strain_f_n1, P, mats, beps):
    f = lib().orc_eigensoftening_hook
    f.restype = C.c_int
    return f(_d(damage_n1), _d(damage_n), _d(strain_f_n1), C.byref(P.c), mats, _i(beps[0]), _i(beps[1]), BEPS_STRIDE)


def set_tangent_damage(damage_n1):
    f = lib().orc_set_tangent_damage
    f.argtypes = [_dp]
    f(_d(damage_n1) if damage_n1 is not None else None)


class BccSet:
    """Dirichlet boundaries: list of dicts {nodes:int[], dim:int, dir:int[dim,nsteps], value:float[dim,nsteps]}."""

    def __init__(self, bcs):
        self.keep = []
        self.n = len(bcs)
        self.arr = (Bcc * max(1, self.n))()
        for i, b in enumerate(bcs):
            nodes = np.ascontiguousarray(b["nodes"], dtype=np.int32)
            d = np.ascontiguousarray(b["dir"], dtype=np.int32)
            v = np.ascontiguousarray(b["value"], dtype=np.float64)
            self.keep += [nodes, d, v]
            self.arr[i] = Bcc(len(nodes), _i(nodes), int(b["dim"]), _i(d), _d(v))


def initialize_lme(P, M, prm):
    return lib().orc_initialize_lme(C.byref(P.c), M.ptr, C.byref(prm))


def local_search(P, M, prm):
    return lib().orc_local_search(C.byref(P.c), M.ptr, C.byref(prm))


def search_phase1(P, M):
    return lib().orc_search_phase1(C.byref(P.c), M.ptr)


def search_phase2(P, M, prm):
    return lib().orc_search_phase2(C.byref(P.c), M.ptr, C.byref(prm))


def active_nodes(M):
    n2m = np.zeros(M.nnodes, dtype=np.int32)
    na = lib().orc_active_nodes(_i(n2m), M.ptr)
    return n2m, na


def active_dofs(n2m, nactive, ndof, bcs, step, nsteps):
    d2m = np.zeros(max(1, nactive * ndof), dtype=np.int32)
    nfree = lib().orc_active_dofs(_i(d2m), _i(n2m), nactive, ndof, bcs.arr, bcs.n, step, nsteps)
    return d2m[: nactive * ndof], nfree


def compute_N(P, M, p):
    N = np.zeros(MAXNB)
    nn = lib().orc_compute_N(_d(N), C.byref(P.c), M.ptr, p)
    return N[:nn]


def compute_dN(P, M, p):
    dN = np.zeros(MAXNB * 3)
    nn = lib().orc_compute_dN(_d(dN), C.byref(P.c), M.ptr, p)
    return dN[: nn * P.ndim].reshape(nn, P.ndim)


def lumped_mass(P, M, n2m, nactive):
    Mv = np.zeros(nactive * P.ndim)
    lib().orc_lumped_mass(_d(Mv), C.byref(P.c), M.ptr, _i(n2m))
    return Mv


def nodal_field_n(Mv, P, M, n2m, d2m, nactive):
    V = np.zeros(nactive * P.ndim)
    A = np.zeros(nactive * P.ndim)
    lib().orc_nodal_field_n(_d(V), _d(A), _d(Mv), C.byref(P.c), M.ptr, _i(n2m), _i(d2m), nactive)
    return V, A


def compatibility(dU, dU_dt, P, M, n2m):
    return lib().orc_compatibility(_d(dU), _d(dU_dt), C.byref(P.c), M.ptr, _i(n2m))


def constitutive(P, mats, prm):
    return lib().orc_constitutive(C.byref(P.c), mats, C.byref(prm))


def internal_forces(P, M, n2m, d2m, nactive):
    R = np.zeros(nactive * P.ndim)
    st = lib().orc_internal_forces(_d(R), C.byref(P.c), M.ptr, _i(n2m), _i(d2m))
    return R, st


def nodal_traction_forces(R, P, M, n2m, d2m, loads, step, nsteps, thickness=1.0, area0=None):
    """__nodal_traction_forces: loads = list of dicts like the Dirichlet ones, nodes = particle indices; R updated in place"""
    nd = P.ndim
    load_n = np.array([len(l["nodes"]) for l in loads], dtype=np.int32)
    ids = np.concatenate([np.asarray(l["nodes"], dtype=np.int32) for l in loads]) if loads else np.zeros(0, np.int32)
    dirs = np.array([[np.asarray(l["dir"]).reshape(nd, nsteps)[i, step] for i in range(nd)] for l in loads], dtype=np.int32)
    vals = np.array([[np.asarray(l["value"]).reshape(nd, nsteps)[i, step] for i in range(nd)] for l in loads], dtype=np.float64)
    f = lib().orc_nodal_traction_forces
    f.argtypes = [_dp, C.POINTER(CParticles), C.POINTER(Mesh), _ip, _ip, C.c_int, _ip, _ip, _ip, _dp, C.c_double, _dp]
    a0 = np.ascontiguousarray(area0, dtype=np.float64) if area0 is not None else None
    return f(_d(R), C.byref(P.c), M.ptr, _i(n2m), _i(d2m), len(loads), _i(load_n), _i(np.ascontiguousarray(ids)),
             _i(np.ascontiguousarray(dirs)), _d(np.ascontiguousarray(vals)), float(thickness), _d(a0) if a0 is not None else None)


def tangent_matrix(P, M, mats, n2m, d2m, nactive, alpha_1=0.0, lumped_mass=None, with_pattern=True):
    """dense Jacobian [ntot, ntot] in masked numbering and the per-row sparsity pattern"""
    ntot = nactive * P.ndim
    K = np.zeros((ntot, ntot))
    pat = np.zeros(ntot, dtype=np.int32) if with_pattern else None
    f = lib().orc_tangent_matrix
    f.argtypes = [_dp, _ip, C.c_double, _dp, C.POINTER(CParticles), C.POINTER(Mesh), C.c_void_p, _ip, _ip, C.c_int]
    st = f(_d(K), _i(pat) if pat is not None else None, float(alpha_1),
           _d(lumped_mass) if lumped_mass is not None else None, C.byref(P.c), M.ptr, mats, _i(n2m),
           _i(d2m) if d2m is not None else None, int(nactive))
    return K, pat, st


def roll_state(P):
    lib().orc_roll_state(C.byref(P.c))


def update_kinetics(alpha_blend, dU, Un_dt, dU_dt, dU_dt2, P, M, n2m):
    f = lib().orc_update_kinetics
    f.argtypes = [C.c_double, _dp, _dp, _dp, _dp, C.POINTER(CParticles), C.POINTER(Mesh), _ip]
    return f(alpha_blend, _d(dU), _d(Un_dt), _d(dU_dt), _d(dU_dt2), C.byref(P.c), M.ptr, _i(n2m))


class ExplicitStepper:
    def __init__(self, P, M, mats, prm, bcs, nsteps, gravity=None):
        self.P, self.M, self.mats, self.prm, self.bcs, self.nsteps = P, M, mats, prm, bcs, nsteps
        nd = M.nnodes * P.ndim
        self.n2m = np.zeros(M.nnodes, dtype=np.int32)
        self.d2m = np.zeros(nd, dtype=np.int32)
        self.bufs = {k: np.zeros(nd) for k in ("mass", "dU", "force", "accel", "reaction")}
        self.gravity = None if gravity is None else np.ascontiguousarray(gravity, dtype=np.float64)
        self.out = StepOut(0, _i(self.n2m), _i(self.d2m), *[_d(self.bufs[k]) for k in
                                                            ("mass", "dU", "force", "accel", "reaction")])
        f = lib().orc_explicit_step
        f.argtypes = [C.POINTER(CParticles), C.POINTER(Mesh), C.POINTER(Material), C.POINTER(Params),
                      C.POINTER(Bcc), C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, _dp,
                      C.POINTER(StepOut)]
        self.f = f

    def step(self, t, dt, gamma=0.5):
        st = self.f(C.byref(self.P.c), self.M.ptr, self.mats, C.byref(self.prm), self.bcs.arr, self.bcs.n, t,
                    self.nsteps, dt, gamma, _d(self.gravity), C.byref(self.out))
        return st

    def nodal(self, k):
        na = self.out.nactive
        return self.bufs[k][: na * self.P.ndim]


def stress_one(ndim, mat, prm, F_n1, DF, J, b_e_n, kappa_n, eps_n, back_stress=None):
    """back_stress: optional array of 3 (Von-Mises), updated in place"""
    T = 5 if ndim == 2 else 9
    stress = np.zeros(T)
    b1 = np.zeros(T)
    W = C.c_double(0)
    k1 = C.c_double(0)
    e1 = C.c_double(0)
    f = lib().orc_stress_one
    f.argtypes = [C.c_int, C.POINTER(Material), C.POINTER(Params), _dp, _dp, C.c_double, _dp, C.c_double,
                  C.c_double, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
    F_n1 = np.ascontiguousarray(F_n1, dtype=np.float64)
    DF = np.ascontiguousarray(DF, dtype=np.float64)
    b_e_n = np.ascontiguousarray(b_e_n, dtype=np.float64)
    st = f(ndim, C.byref(mat), C.byref(prm), _d(F_n1), _d(DF), float(J), _d(b_e_n), float(kappa_n),
           float(eps_n), _d(stress), C.byref(W), _d(b1), C.byref(k1), C.byref(e1), None,
           _d(back_stress) if back_stress is not None else None)
    return st, stress, W.value, b1, k1.value, e1.value


def trial_b_e(d_phi, b_e_n, ndim):
    out = np.zeros(ndim * ndim)
    f = lib().orc_trial_b_e
    f.argtypes = [_dp, _dp, _dp, C.c_int]
    f.restype = None
    f(_d(out), _d(np.ascontiguousarray(d_phi, dtype=np.float64)), _d(np.ascontiguousarray(b_e_n, dtype=np.float64)), ndim)
    return out.reshape(ndim, ndim)


def stiffness_density_spectral(dN_alpha_n1, dN_beta_n1, b, Cmod, stress, ndim):
    """Elastoplastic-Tangent-Matrix.c:42-163 for one pair of nodes (row-major d x d)."""
    out = np.zeros(ndim * ndim)
    f = lib().orc_stiffness_density_spectral
    f.argtypes = [_dp, C.c_int, _dp, _dp, _dp, _dp, _dp]
    f.restype = C.c_int
    a = [np.ascontiguousarray(x, dtype=np.float64).ravel() for x in (dN_alpha_n1, dN_beta_n1, b, Cmod, stress)]
    st = f(_d(out), ndim, *[_d(x) for x in a])
    assert st == 0
    return out.reshape(ndim, ndim)


def sym_eigen(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    n = A.shape[0]
    w = np.zeros(3)
    v = np.zeros(n * n)
    st = lib().orc_sym_eigen(_d(w), _d(v), _d(A), n)
    return st, w[:n], v.reshape(n, n)


def inverse(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    n = A.shape[0]
    out = np.zeros(n * n)
    st = lib().orc_inverse(_d(out), _d(A), n)
    return st, out.reshape(n, n)


def rcond_ref(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    return lib().orc_rcond_ref(_d(A), A.shape[0])


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def N_gimp(Delta_Xp, lp, L):
    D = np.ascontiguousarray(Delta_Xp, dtype=np.float64)
    lp = np.ascontiguousarray(lp, dtype=np.float64)
    S = np.zeros(D.shape[0])
    lib().orc_N_gimp.argtypes = [_dp, _dp, C.c_int, C.c_int, _dp, C.c_double]
    lib().orc_N_gimp(_d(S), _d(D), D.shape[0], D.shape[1], _d(lp), float(L))
    return S


def dN_gimp(Delta_Xp, lp, L):
    D = np.ascontiguousarray(Delta_Xp, dtype=np.float64)
    lp = np.ascontiguousarray(lp, dtype=np.float64)
    dS = np.zeros(D.shape)
    lib().orc_dN_gimp.argtypes = [_dp, _dp, C.c_int, C.c_int, _dp, C.c_double]
    lib().orc_dN_gimp(_d(dS), _d(D), D.shape[0], D.shape[1], _d(lp), float(L))
    return dS
