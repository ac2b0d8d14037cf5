"""ctypes binding of libnlps_gpu.so (include/nlps_gpu.h) and a thin host-side mirror of the
reference's stage interface.

Method names follow the reference's functions so that parity tests read like the reference driver
(nl-partsol/src/Formulations/Displacements/U-Newmark-beta.c:192-409):
    local_search__MeshTools__  -> Solver.local_search()
    get_active_nodes/dofs      -> Solver.active_masks()
    __compute_nodal_lumped_mass-> Solver.compute_nodal_lumped_mass() ...
The library is REQUIRED: a missing .so or a missing GPU raises, there is no CPU fallback here.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

MAXNB = 128
MAT_NEO_HOOKEAN, MAT_HENCKY, MAT_DRUCKER_PRAGER = 0, 1, 2

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class Grid(C.Structure):
    _fields_ = [("ndim", C.c_int), ("n", C.c_int * 3), ("origin", C.c_double * 3), ("h", C.c_double),
                ("h_avg", _dp)]


class Params(C.Structure):
    _fields_ = [("gamma_lme", C.c_double), ("tol_zero_lme", C.c_double), ("tol_wrapper_lme", C.c_double),
                ("max_iter_lme", C.c_int), ("tol_radial_returning", C.c_double),
                ("max_iter_radial_returning", C.c_int), ("driver_eigenerosion", C.c_int),
                ("driver_eigensoftening", C.c_int)]


class Material(C.Structure):
    _fields_ = [("type", C.c_int), ("E", C.c_double), ("nu", C.c_double), ("phi_deg", C.c_double),
                ("psi_deg", C.c_double), ("kappa_0", C.c_double), ("exponent_ortiz", C.c_double),
                ("eps_0", C.c_double), ("p_ref", C.c_double), ("hardening_modulus", C.c_double),
                ("theta_voce", C.c_double), ("K0_voce", C.c_double), ("Kinf_voce", C.c_double),
                ("delta_voce", C.c_double), ("Ceps", C.c_double), ("Gf", C.c_double),
                ("cohesion", C.c_double), ("alpha_borja", C.c_double), ("a_borja", C.c_double * 3),
                ("ft", C.c_double), ("heps", C.c_double), ("wcrit", C.c_double)]


_PD = ["x_GC", "dis", "vel", "acc", "F_n", "F_n1", "DF", "Stress", "b_e_n", "b_e_n1", "J_n", "J_n1", "rho",
       "mass", "Vol_0", "W", "Kappa_n", "Kappa_n1", "EPS_n", "EPS_n1"]


class Particles(C.Structure):
    _fields_ = ([("np", C.c_int)] + [(k, _dp) for k in _PD] +
                [("MatIdx", _ip), ("I0", _ip), ("lambda_", _dp), ("Beta", _dp), ("dt_F_n", _dp), ("dt_F_n1", _dp),
                 ("dt_DF", _dp), ("C_ep", _dp), ("Back_stress", _dp), ("Damage_n", _dp), ("Damage_n1", _dp),
                 ("Strain_f_n", _dp), ("Strain_f_n1", _dp)])


class Bcc(C.Structure):
    _fields_ = [("nnodes", C.c_int), ("nodes", _ip), ("dim", C.c_int), ("dir", _ip), ("value", _dp)]


HALO_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int)

_LIB = None

# every symbol include/nlps_gpu.h declares
SYMBOLS = ["nlps_gpu_lagrangian_evaluation", "nlps_gpu_create", "nlps_gpu_destroy", "nlps_gpu_last_error", "nlps_gpu_synchronize",
           "nlps_gpu_download_state", "nlps_gpu_download_lists", "nlps_gpu_shape_functions", "nlps_gpu_download_active",
           "nlps_gpu_status_flags", "nlps_gpu_initialize_lme", "nlps_gpu_local_search", "nlps_gpu_active_masks",
           "nlps_gpu_set_node_numbering",
           "nlps_gpu_lumped_mass", "nlps_gpu_nodal_field_n", "nlps_gpu_compatibility", "nlps_gpu_constitutive",
           "nlps_gpu_internal_forces", "nlps_gpu_nodal_traction_forces", "nlps_gpu_roll_state", "nlps_gpu_update_kinetics",
           "nlps_gpu_explicit_step", "nlps_gpu_num_active", "nlps_gpu_explicit_nodal", "nlps_gpu_set_halo_exchange",
           "nlps_gpu_resort", "nlps_gpu_set_resort_interval", "nlps_gpu_set_adaptive_resort", "nlps_gpu_set_law_launch_mode", "nlps_gpu_set_deterministic",
           "nlps_gpu_rccl_unique_id", "nlps_gpu_rccl_attach", "nlps_gpu_rccl_attach_comm", "nlps_gpu_rccl_detach",
           "nlps_gpu_rccl_reduce", "nlps_gpu_rccl_info", "nlps_gpu_rccl_migrate", "nlps_gpu_rccl_selftest_migrate", "nlps_gpu_rccl_selftest_exchange", "nlps_gpu_touched_layers", "nlps_gpu_set_node_window", "nlps_gpu_set_ghost_bands",
           "nlps_gpu_form_initial_guess", "nlps_gpu_nodal_kinetic_increments", "nlps_gpu_nodal_inertial_forces",
           "nlps_gpu_tangent_assemble", "nlps_gpu_tangent_set_grouped", "nlps_gpu_tangent_coo",
           "nlps_gpu_sparsity_pattern",
           "nlps_gpu_migration_select", "nlps_gpu_migration_commit", "nlps_gpu_num_particles",
           "nlps_gpu_set_particle_ids", "nlps_gpu_download_ids",
           "nlps_gpu_set_timing", "nlps_gpu_get_timing", "nlps_host_stencil_tables",
           "nlps_host_io_last_error", "nlps_host_gid_mesh_info", "nlps_host_gid_mesh_read",
           "nlps_host_lattice_from_nodes", "nlps_host_particles_from_mesh", "nlps_host_write_particles_vtk", "nlps_host_write_nodes_vtk", "nlps_host_read_deck", "nlps_host_read_materials", "nlps_host_read_boundaries", "nlps_host_read_initials", "nlps_host_read_gravity", "nlps_host_read_outputs", "nlps_host_read_neumann",
           "nlps_host_read_material_assignment"]


def lib():
    """Loads the HIP library; raises if it is missing (no fallback)."""
    global _LIB
    if _LIB is None:
        path = _build.LIB
        try:
            # torch bundles its own libamdhip64.so.7; loading it FIRST makes this library bind to that
            # same runtime (one HIP runtime per process), otherwise torch.cuda sees no device afterwards.
            import torch  # noqa: F401
        except ImportError:
            pass
        if not os.path.exists(path):
            raise RuntimeError("libnlps_gpu.so is not built: run __graft_entry__.build() "
                               "(nl-partsol_amd has no CPU fallback)")
        L = C.CDLL(path)
        L.nlps_gpu_last_error.restype = C.c_char_p
        L.nlps_gpu_last_error.argtypes = [C.c_void_p]
        L.nlps_gpu_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(Grid), C.POINTER(Params),
                                      C.POINTER(Material), C.c_int, C.POINTER(Particles), C.c_int, C.c_void_p]
        L.nlps_gpu_set_resort_interval.argtypes = [C.c_void_p, C.c_int]
        L.nlps_gpu_set_law_launch_mode.argtypes = [C.c_void_p, C.c_int]
        L.nlps_gpu_set_deterministic.argtypes = [C.c_void_p, C.c_int]
        L.nlps_gpu_rccl_unique_id.argtypes = [C.c_void_p]
        L.nlps_gpu_rccl_attach.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _ip, _ip, C.c_int]
        L.nlps_gpu_rccl_attach_comm.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _ip, _ip, C.c_int]
        L.nlps_gpu_rccl_detach.argtypes = [C.c_void_p]
        L.nlps_gpu_rccl_reduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.nlps_gpu_rccl_info.argtypes = [C.c_void_p, _ip, _ip, _ip]
        L.nlps_gpu_rccl_migrate.argtypes = [C.c_void_p, C.c_int, C.c_int, _ip, _ip, _ip]
        L.nlps_gpu_rccl_selftest_migrate.argtypes = [C.c_void_p, C.c_int, C.c_int, _ip, _ip, _ip]
        L.nlps_gpu_rccl_selftest_exchange.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.nlps_gpu_set_node_window.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.nlps_gpu_set_node_numbering.argtypes = [C.c_void_p, _ip]
        L.nlps_gpu_set_ghost_bands.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.nlps_gpu_form_initial_guess.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int,
                                                  C.POINTER(Bcc), C.c_int, C.c_int]
        L.nlps_gpu_nodal_kinetic_increments.argtypes = [C.c_void_p] + [C.c_void_p] * 5 + [_dp]
        L.nlps_gpu_nodal_inertial_forces.argtypes = [C.c_void_p] + [C.c_void_p] * 5 + [_dp, _dp]
        L.nlps_gpu_tangent_assemble.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
        L.nlps_gpu_tangent_set_grouped.argtypes = [C.c_void_p, C.c_int]
        L.nlps_gpu_tangent_coo.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_int, _ip, _ip, _dp]
        L.nlps_gpu_sparsity_pattern.argtypes = [C.c_void_p, _ip]
        L.nlps_gpu_migration_select.argtypes = [C.c_void_p, C.c_int, C.c_int, _ip, _ip, _ip, C.POINTER(C.c_void_p),
                                                C.POINTER(C.c_void_p)]
        L.nlps_gpu_migration_commit.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.nlps_gpu_num_particles.argtypes = [C.c_void_p, _ip]
        L.nlps_gpu_set_particle_ids.argtypes = [C.c_void_p, _ip]
        L.nlps_gpu_download_ids.argtypes = [C.c_void_p, _ip]
        for name in ["nlps_gpu_destroy", "nlps_gpu_synchronize", "nlps_gpu_initialize_lme", "nlps_gpu_resort",
                     "nlps_gpu_local_search", "nlps_gpu_constitutive", "nlps_gpu_roll_state"]:
            getattr(L, name).argtypes = [C.c_void_p]
        L.nlps_gpu_download_state.argtypes = [C.c_void_p, C.POINTER(Particles)]
        L.nlps_gpu_download_lists.argtypes = [C.c_void_p, _ip, _ip]
        L.nlps_gpu_shape_functions.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.nlps_gpu_download_active.argtypes = [C.c_void_p, C.c_void_p]
        L.nlps_gpu_status_flags.argtypes = [C.c_void_p, _ip]
        L.nlps_gpu_active_masks.argtypes = [C.c_void_p, C.POINTER(Bcc), C.c_int, C.c_int, _ip, _ip, _ip, _ip]
        L.nlps_gpu_lumped_mass.argtypes = [C.c_void_p, C.c_void_p]
        L.nlps_gpu_nodal_field_n.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.nlps_gpu_compatibility.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.nlps_gpu_internal_forces.argtypes = [C.c_void_p, C.c_void_p]
        L.nlps_gpu_update_kinetics.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p]
        L.nlps_gpu_explicit_step.argtypes = [C.c_void_p, C.POINTER(Bcc), C.c_int, C.c_int, C.c_double,
                                             C.c_double, _dp]
        L.nlps_gpu_explicit_nodal.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        L.nlps_gpu_num_active.argtypes = [C.c_void_p, _ip]
        L.nlps_gpu_set_halo_exchange.argtypes = [C.c_void_p, HALO_FN, C.c_void_p]
        L.nlps_gpu_touched_layers.argtypes = [C.c_void_p, _ip, _ip]
        L.nlps_gpu_set_timing.argtypes = [C.c_void_p, C.c_int]
        L.nlps_gpu_get_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        _LIB = L
    return _LIB


def default_params():
    return Params(3.0, 1e-6, 1e-10, 10, 1e-14, 10, 0, 0)


def host_stencil_tables(ndim):
    """Host-only: the stencil order tables of csrc/nlps_tables.hpp (no GPU needed)."""
    rank1 = np.zeros((27, 27), dtype=np.uint8)
    order2 = np.zeros((125, 125), dtype=np.uint8)
    count2 = np.zeros(125, dtype=np.uint8)
    h1 = np.zeros(27)
    f = lib().nlps_host_stencil_tables
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    st = f(int(ndim), rank1.ctypes.data_as(C.c_void_p), order2.ctypes.data_as(C.c_void_p),
           count2.ctypes.data_as(C.c_void_p), h1.ctypes.data_as(C.c_void_p))
    if st:
        raise NlpsError("nlps_host_stencil_tables failed")
    return rank1, order2, count2, h1


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _i(a):
    return None if a is None else a.ctypes.data_as(_ip)


def _vp(a):
    """numpy array (host) or torch tensor / int (device pointer) -> void*"""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    return C.c_void_p(int(a))


class BccSet:
    """FEM_Mesh.Bounds: list of dicts {nodes, dim, dir[dim][nsteps], value[dim][nsteps]}."""

    def __init__(self, bcs):
        self.keep = []
        self.n = len(bcs)
        self.arr = (Bcc * max(1, self.n))()
        for k, b in enumerate(bcs):
            nodes = np.ascontiguousarray(b["nodes"], dtype=np.int32)
            d = np.ascontiguousarray(b["dir"], dtype=np.int32)
            v = np.ascontiguousarray(b["value"], dtype=np.float64)
            self.keep += [nodes, d, v]
            self.arr[k] = Bcc(len(nodes), _i(nodes), int(b["dim"]), _i(d), _d(v))


class NlpsError(RuntimeError):
    pass


class Solver:
    """Device-resident particle set + background grid; one method per reference stage function."""

    def __init__(self, ndim, grid_n, origin, h, cloud, materials, params=None, nsteps=1, stream=None,
                 h_avg=None):
        self.L = lib()
        self.ndim = int(ndim)
        self.T = 5 if ndim == 2 else 9
        self.np = int(cloud["x"].shape[0])
        n3 = list(grid_n) + [1] * (3 - len(grid_n))
        o3 = list(origin) + [0.0] * (3 - len(origin))
        self.grid_n = n3
        self.nnodes = n3[0] * n3[1] * n3[2]
        self._h_avg = None if h_avg is None else np.ascontiguousarray(h_avg, dtype=np.float64)
        g = Grid(self.ndim, (C.c_int * 3)(*n3), (C.c_double * 3)(*o3), float(h), _d(self._h_avg))
        self.params = params or default_params()
        mats = (Material * len(materials))()
        for k, m in enumerate(materials):
            mats[k] = Material(int(m["type"]), float(m["E"]), float(m["nu"]), float(m.get("phi_deg", 0.0)),
                               float(m.get("psi_deg", 0.0)), float(m.get("kappa_0", 0.0)),
                               float(m.get("exponent_ortiz", 1.0)), float(m.get("eps_0", 1.0)),
                               float(m.get("p_ref", 0.0)), float(m.get("hardening_modulus", 0.0)),
                               float(m.get("theta_voce", 1.0)), float(m.get("K0_voce", 0.0)),
                               float(m.get("Kinf_voce", 0.0)), float(m.get("delta_voce", 0.0)),
                               float(m.get("Ceps", 0.0)), float(m.get("Gf", 0.0)), float(m.get("cohesion", 0.0)),
                               float(m.get("alpha_borja", 0.0)),
                               (C.c_double * 3)(*[float(v) for v in m.get("a_borja", (0.0, 0.0, 0.0))]),
                               float(m.get("ft", 0.0)), float(m.get("heps", 0.0)), float(m.get("wcrit", 1.0)))
        self._host = {}
        hp = Particles()
        hp.np = self.np
        keymap = {"x_GC": "x", "dis": "dis", "vel": "vel", "acc": "acc", "F_n": "F_n", "b_e_n": "b_e_n",
                  "J_n": "J_n", "rho": "rho", "mass": "mass", "Vol_0": "vol0", "Kappa_n": "kappa_n",
                  "EPS_n": "eps_n", "lambda_": "lambda", "Beta": "beta", "dt_F_n": "dt_F_n",
                  "Back_stress": "back_stress", "Damage_n": "damage_n", "Strain_f_n": "strain_f_n"}
        for ck, k in keymap.items():
            if k in cloud and cloud[k] is not None:
                a = np.ascontiguousarray(cloud[k], dtype=np.float64)
                self._host[ck] = a
                setattr(hp, ck, _d(a))
        mi = np.ascontiguousarray(cloud["matidx"], dtype=np.int32)
        self._host["MatIdx"] = mi
        hp.MatIdx = _i(mi)
        if cloud.get("I0") is not None:
            i0 = np.ascontiguousarray(cloud["I0"], dtype=np.int32)
            self._host["I0"] = i0
            hp.I0 = _i(i0)
        self.nsteps = int(nsteps)
        self.h = C.c_void_p()
        st = self.L.nlps_gpu_create(C.byref(self.h), C.byref(g), C.byref(self.params), mats, len(materials),
                                    C.byref(hp), self.nsteps, C.c_void_p(stream) if stream else None)
        if st:
            raise NlpsError(self.L.nlps_gpu_last_error(self.h).decode())
        self.nactive = 0
        self.nfree = 0
        self._halo_cb = None

    # ------------------------------------------------------------------ helpers
    def _chk(self, st):
        if st:
            raise NlpsError(self.L.nlps_gpu_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.L.nlps_gpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self._chk(self.L.nlps_gpu_synchronize(self.h))

    # ------------------------------------------------------------------ reference stages
    def initialise_shapefun(self):          # initialise_shapefun__MeshTools__ -> initialize__LME__
        self._chk(self.L.nlps_gpu_initialize_lme(self.h))

    def local_search(self):                 # local_search__MeshTools__
        self._chk(self.L.nlps_gpu_local_search(self.h))

    def active_masks(self, bcs, step, download=True):   # get_active_nodes/_dofs__MeshTools__
        na, nf = C.c_int(0), C.c_int(0)
        n2m = np.zeros(self.nnodes, dtype=np.int32) if download else None
        d2m = np.zeros(self.nnodes * self.ndim, dtype=np.int32) if download else None
        self._chk(self.L.nlps_gpu_active_masks(self.h, bcs.arr, bcs.n, step, C.byref(na), C.byref(nf), _i(n2m),
                                               _i(d2m)))
        self.nactive, self.nfree = na.value, nf.value
        if download:
            return n2m, d2m[: self.nactive * self.ndim]
        return None, None

    def compute_nodal_lumped_mass(self, out=None):      # __compute_nodal_lumped_mass
        M = np.zeros(self.nactive * self.ndim) if out is None else out
        self._chk(self.L.nlps_gpu_lumped_mass(self.h, _vp(M)))
        return M

    def get_nodal_field_n(self, M, V=None, A=None):     # __get_nodal_field_n
        V = np.zeros(self.nactive * self.ndim) if V is None else V
        A = np.zeros(self.nactive * self.ndim) if A is None else A
        self._chk(self.L.nlps_gpu_nodal_field_n(self.h, _vp(V), _vp(A), _vp(M)))
        return V, A

    def local_compatibility_conditions(self, dU, dU_dt=None):   # __local_compatibility_conditions
        self._chk(self.L.nlps_gpu_compatibility(self.h, _vp(dU), _vp(dU_dt)))

    def constitutive_update(self):                      # __constitutive_update
        self._chk(self.L.nlps_gpu_constitutive(self.h))

    def nodal_internal_forces(self, R):                 # __nodal_internal_forces (accumulates into R)
        self._chk(self.L.nlps_gpu_internal_forces(self.h, _vp(R)))
        return R

    def nodal_traction_forces(self, R, loads, step, thickness=1.0, area0=None):   # __nodal_traction_forces
        self.L.nlps_gpu_nodal_traction_forces.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Bcc), C.c_int, C.c_int,
                                                          C.c_double, C.c_void_p]
        a0 = None if area0 is None else np.ascontiguousarray(area0, dtype=np.float64)
        self._chk(self.L.nlps_gpu_nodal_traction_forces(self.h, _vp(R), loads.arr, loads.n, int(step), float(thickness),
                                                        None if a0 is None else a0.ctypes.data))
        return R

    def update_particles_internal_variables(self):      # __update_particles_internal_variables
        self._chk(self.L.nlps_gpu_roll_state(self.h))

    def update_particles_kinetics_FLIP_PIC(self, alpha_blend, dU, Un_dt, dU_dt, dU_dt2):
        self._chk(self.L.nlps_gpu_update_kinetics(self.h, float(alpha_blend), _vp(dU), _vp(Un_dt), _vp(dU_dt),
                                                  _vp(dU_dt2)))

    def explicit_step(self, bcs, step, dt, gamma=0.5, gravity=None):
        g = None if gravity is None else np.ascontiguousarray(gravity, dtype=np.float64)
        self._chk(self.L.nlps_gpu_explicit_step(self.h, bcs.arr, bcs.n, int(step), float(dt), float(gamma), _d(g)))

    def explicit_nodal(self):
        """Nodal results of the last explicit step in masked numbering (host arrays)."""
        na = C.c_int(0)
        self._chk(self.L.nlps_gpu_num_active(self.h, C.byref(na)))
        self.nactive = na.value
        out = {k: np.zeros(self.nactive * self.ndim) for k in ("mass", "dU", "force", "accel", "reaction")}
        self._chk(self.L.nlps_gpu_explicit_nodal(self.h, *[_vp(out[k]) for k in
                                                           ("mass", "dU", "force", "accel", "reaction")]))
        return out

    # ------------------------------------------------------------------ downloads
    def download_state(self, fields=None):
        """All particle fields in the caller's order; `fields` (names as below) limits the download (large clouds)."""
        n, d, T = self.np, self.ndim, self.T
        o = {"x_GC": np.zeros((n, d)), "dis": np.zeros((n, d)), "vel": np.zeros((n, d)), "acc": np.zeros((n, d)),
             "F_n": np.zeros((n, T)), "F_n1": np.zeros((n, T)), "DF": np.zeros((n, T)), "Stress": np.zeros((n, T)),
             "b_e_n": np.zeros((n, T)), "b_e_n1": np.zeros((n, T)), "J_n": np.zeros(n), "J_n1": np.zeros(n),
             "rho": np.zeros(n), "mass": np.zeros(n), "Vol_0": np.zeros(n), "W": np.zeros(n),
             "Kappa_n": np.zeros(n), "Kappa_n1": np.zeros(n), "EPS_n": np.zeros(n), "EPS_n1": np.zeros(n),
             "lambda_": np.zeros((n, d)), "Beta": np.zeros(n), "dt_F_n": np.zeros((n, T)),
             "dt_F_n1": np.zeros((n, T)), "dt_DF": np.zeros((n, T)), "C_ep": np.zeros((n, d * d)),
             "Back_stress": np.zeros((n, 3)), "Damage_n": np.zeros(n), "Damage_n1": np.zeros(n),
             "Strain_f_n": np.zeros(n), "Strain_f_n1": np.zeros(n)}
        if fields is not None:
            alias = {"x": "x_GC", "lambda": "lambda_", "beta": "Beta"}
            want = {alias.get(k, k) for k in fields}
            o = {k: a for k, a in o.items() if k in want}
        i0 = np.zeros(n, dtype=np.int32) if fields is None or "I0" in fields else None
        hp = Particles()
        hp.np = n
        for k, a in o.items():
            setattr(hp, k, _d(a))
        hp.I0 = _i(i0)
        self._chk(self.L.nlps_gpu_download_state(self.h, C.byref(hp)))
        if i0 is not None:
            o["I0"] = i0
        for short, full in (("x", "x_GC"), ("lambda", "lambda_"), ("beta", "Beta")):
            if full in o:
                o[short] = o[full]
        return o

    def download_lists(self):
        nn = np.zeros(self.np, dtype=np.int32)
        lst = np.zeros((self.np, MAXNB), dtype=np.int32)
        self._chk(self.L.nlps_gpu_download_lists(self.h, _i(nn), _i(lst)))
        return nn, lst

    def shape_functions(self, first=0, count=None):
        """N[count][MAXNB] and dN[count][MAXNB][ndim] of particles first .. first + count - 1 in the order of their lists."""
        count = self.np - first if count is None else count
        N = np.zeros((count, MAXNB))
        dN = np.zeros((count, MAXNB, self.ndim))
        self._chk(self.L.nlps_gpu_shape_functions(self.h, int(first), int(count), _vp(N), _vp(dN)))
        return N, dN

    def download_active(self):
        a = np.zeros(self.nnodes, dtype=np.uint8)
        self._chk(self.L.nlps_gpu_download_active(self.h, a.ctypes.data_as(C.c_void_p)))
        return a

    def status_flags(self):
        f = C.c_int(0)
        self._chk(self.L.nlps_gpu_status_flags(self.h, C.byref(f)))
        return f.value

    # ------------------------------------------------------------------ multi-GPU / measurement
    def set_halo_exchange(self, pyfunc):
        """pyfunc(dptr:int, nfield:int, elem_bytes:int, kind:int[, phase:int]) -> int
        (phase 0 = exchange now, 1 = start, 2 = wait; a 4-argument function only ever sees blocking exchanges)"""
        if pyfunc is None:
            self._halo_cb = None
            self._chk(self.L.nlps_gpu_set_halo_exchange(self.h, C.cast(None, HALO_FN), None))
            return
        import inspect
        with_phase = len(inspect.signature(pyfunc).parameters) >= 5

        def _cb(ctx, dptr, nfield, elem, kind, phase):
            try:
                if with_phase:
                    return int(pyfunc(dptr, nfield, elem, kind, phase) or 0)
                if phase == 2:
                    return 0  # the matching start call already did the whole exchange in stream order
                return int(pyfunc(dptr, nfield, elem, kind) or 0)
            except Exception as e:  # never let an exception cross the C boundary
                print("halo exchange failed:", repr(e))
                return 1

        self._halo_cb = HALO_FN(_cb)
        self._chk(self.L.nlps_gpu_set_halo_exchange(self.h, self._halo_cb, None))

    def resort(self):
        self._chk(self.L.nlps_gpu_resort(self.h))

    def set_resort_interval(self, n):
        self._chk(self.L.nlps_gpu_set_resort_interval(self.h, int(n)))

    def set_adaptive_resort(self, budget, min_steps=4):
        self.L.nlps_gpu_set_adaptive_resort.argtypes = [C.c_void_p, C.c_double, C.c_int]
        self._chk(self.L.nlps_gpu_set_adaptive_resort(self.h, float(budget), int(min_steps)))

    def debug_option(self, name, value):
        """developer / test switch between equivalent launch forms (nlps_gpu_debug_option; the library reads no environment)"""
        self.L.nlps_gpu_debug_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        self._chk(self.L.nlps_gpu_debug_option(self.h, name.encode(), float(value)))

    def debug_displaced(self):
        v, d = C.c_int(0), C.c_double(0)
        self.L.nlps_gpu_debug_displaced.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        self._chk(self.L.nlps_gpu_debug_displaced(self.h, C.byref(v), C.byref(d)))
        return v.value, d.value

    def set_law_launch_mode(self, mode):
        self._chk(self.L.nlps_gpu_set_law_launch_mode(self.h, int(mode)))

    # ------------------------------------------------------------------ RCCL owned by the library
    @staticmethod
    def rccl_unique_id():
        """128-byte ncclUniqueId (made by rank 0, handed to the other ranks by the host driver)"""
        buf = (C.c_ubyte * 128)()
        if lib().nlps_gpu_rccl_unique_id(buf):
            raise NlpsError("nlps_gpu_rccl_unique_id failed (librccl.so.1 not loadable?)")
        return bytes(buf)

    def rccl_attach(self, uid, rank, world, layer_lo, layer_hi, mode=0):
        buf = (C.c_ubyte * 128).from_buffer_copy(uid)
        lo = np.ascontiguousarray(layer_lo, dtype=np.int32)
        hi = np.ascontiguousarray(layer_hi, dtype=np.int32)
        self._chk(self.L.nlps_gpu_rccl_attach(self.h, buf, int(rank), int(world), _i(lo), _i(hi), int(mode)))

    def rccl_detach(self):
        self._chk(self.L.nlps_gpu_rccl_detach(self.h))

    def rccl_reduce(self, dptr, n, root=-1):
        self._chk(self.L.nlps_gpu_rccl_reduce(self.h, _vp(dptr), int(n), int(root)))

    def rccl_info(self):
        """-> (nranks, rank) as RCCL reports them, overlap mode in force (0 blocking, 1 split launches, 2 one launch per stage)"""
        n, r, m = C.c_int(0), C.c_int(0), C.c_int(0)
        self._chk(self.L.nlps_gpu_rccl_info(self.h, C.byref(n), C.byref(r), C.byref(m)))
        return n.value, r.value, m.value

    def rccl_migrate(self, keep_lo, keep_hi, selftest=False):
        """Migration with the transport inside the library (ncclSend / ncclRecv): -> (sent_down, sent_up, received)"""
        d, u, g = C.c_int(0), C.c_int(0), C.c_int(0)
        fn = self.L.nlps_gpu_rccl_selftest_migrate if selftest else self.L.nlps_gpu_rccl_migrate
        self._chk(fn(self.h, int(keep_lo), int(keep_hi), C.byref(d), C.byref(u), C.byref(g)))
        self.num_particles()
        return d.value, u.value, g.value

    def rccl_selftest_exchange(self, dptr, nfield, elem_bytes, kind, overlap):
        self._chk(self.L.nlps_gpu_rccl_selftest_exchange(self.h, _vp(dptr), int(nfield), int(elem_bytes), int(kind),
                                                         1 if overlap else 0))

    def set_deterministic(self, on=True):
        self._chk(self.L.nlps_gpu_set_deterministic(self.h, 1 if on else 0))

    def touched_layers(self):
        lo, hi = C.c_int(0), C.c_int(0)
        self._chk(self.L.nlps_gpu_touched_layers(self.h, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def form_initial_guess(self, Un_dt, Un_dt2, dt, bcs, step, use_explicit_trial=True):  # __form_initial_guess
        dU = np.zeros(self.nactive * self.ndim)
        self._chk(self.L.nlps_gpu_form_initial_guess(self.h, _vp(dU), _vp(Un_dt), _vp(Un_dt2), float(dt),
                                                     1 if use_explicit_trial else 0, bcs.arr, bcs.n, int(step)))
        return dU

    def compute_nodal_kinetic_increments(self, dU, Un_dt, Un_dt2, alpha):  # __compute_nodal_kinetic_increments
        n = self.nactive * self.ndim
        dV, dA = np.zeros(n), np.zeros(n)
        al = np.ascontiguousarray(alpha, dtype=np.float64)
        self._chk(self.L.nlps_gpu_nodal_kinetic_increments(self.h, _vp(dV), _vp(dA), _vp(dU), _vp(Un_dt), _vp(Un_dt2),
                                                           _d(al)))
        return dV, dA

    def nodal_inertial_forces(self, R, M, dU, Un_dt, Un_dt2, alpha, gravity=None):  # __nodal_inertial_forces
        al = np.ascontiguousarray(alpha, dtype=np.float64)
        gv = None if gravity is None else np.ascontiguousarray(gravity, dtype=np.float64)
        self._chk(self.L.nlps_gpu_nodal_inertial_forces(self.h, _vp(R), _vp(M), _vp(dU), _vp(Un_dt), _vp(Un_dt2), _d(al),
                                                        _d(gv)))
        return R

    LAGR_RATES, LAGR_SEPARATE, LAGR_SAME_STEP = 1, 2, 4

    def lagrangian_evaluation(self, dU, Un_dt, Un_dt2, M, alpha, gravity=None, loads=None, step=0, thickness=1.0,
                              area0=None, flags=0, out=None):  # __lagrangian_evaluation
        """The residual of the implicit driver's SNES solve as one device call (nlps_gpu_lagrangian_evaluation).
        Vectors: numpy arrays (host) or torch tensors (device); out = the residual vector to overwrite (default: a new
        numpy array)."""
        self.L.nlps_gpu_lagrangian_evaluation.argtypes = [C.c_void_p] + [C.c_void_p] * 5 + [_dp, _dp, C.POINTER(Bcc), C.c_int,
                                                          C.c_int, C.c_double, C.c_void_p, C.c_int]
        R = np.zeros(self.nactive * self.ndim) if out is None else out
        al = np.ascontiguousarray(alpha, dtype=np.float64)
        gv = None if gravity is None else np.ascontiguousarray(gravity, dtype=np.float64)
        a0 = None if area0 is None else np.ascontiguousarray(area0, dtype=np.float64)
        self._chk(self.L.nlps_gpu_lagrangian_evaluation(
            self.h, _vp(R), _vp(dU), _vp(Un_dt), _vp(Un_dt2), _vp(M), _d(al), _d(gv), None if loads is None else loads.arr,
            0 if loads is None else loads.n, int(step), float(thickness), None if a0 is None else a0.ctypes.data, int(flags)))
        return R

    def jacobian_evaluation(self, alpha_1=0.0, lumped_mass=None, apply_dirichlet=False, on_device=False):  # __jacobian_evaluation
        """COO triplets (rows, cols, vals) of the tangent matrix in masked dof numbering.  on_device: torch tensors on
        the GPU, written by the emit kernel itself (what MatSetValuesCOO of a GPU matrix type takes; nothing crosses PCIe)."""
        nnz = C.c_longlong(0)
        self._chk(self.L.nlps_gpu_tangent_assemble(self.h, C.byref(nnz)))
        n = int(nnz.value)
        self.L.nlps_gpu_tangent_coo.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        if on_device:
            import torch
            rows = torch.empty(n, dtype=torch.int32, device="cuda")
            cols = torch.empty(n, dtype=torch.int32, device="cuda")
            vals = torch.empty(n, dtype=torch.float64, device="cuda")
        else:
            rows, cols, vals = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32), np.zeros(n)
        self._chk(self.L.nlps_gpu_tangent_coo(self.h, float(alpha_1), _vp(lumped_mass), 1 if apply_dirichlet else 0,
                                              _vp(rows), _vp(cols), _vp(vals)))
        return rows, cols, vals

    def create_sparsity_pattern(self):                  # __create_sparsity_pattern (after jacobian_evaluation)
        pat = np.zeros(self.nactive * self.ndim, dtype=np.int32)
        self._chk(self.L.nlps_gpu_sparsity_pattern(self.h, _i(pat)))
        return pat

    # ------------------------------------------------------------------ migration
    def num_particles(self):
        n = C.c_int(0)
        self._chk(self.L.nlps_gpu_num_particles(self.h, C.byref(n)))
        self.np = n.value
        return n.value

    def set_particle_ids(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        self._chk(self.L.nlps_gpu_set_particle_ids(self.h, _i(ids)))

    def download_ids(self):
        ids = np.zeros(self.num_particles(), dtype=np.int32)
        self._chk(self.L.nlps_gpu_download_ids(self.h, _i(ids)))
        return ids

    def migration_select(self, keep_lo, keep_hi):
        """-> (n_down, n_up, row_words, down_dptr, up_dptr): packed rows of the particles that leave"""
        nd, nu, rw = C.c_int(0), C.c_int(0), C.c_int(0)
        dp, up = C.c_void_p(), C.c_void_p()
        self._chk(self.L.nlps_gpu_migration_select(self.h, int(keep_lo), int(keep_hi), C.byref(nd), C.byref(nu),
                                                   C.byref(rw), C.byref(dp), C.byref(up)))
        return nd.value, nu.value, rw.value, dp.value, up.value

    def migration_commit(self, rows_a, n_a, rows_b, n_b):
        """rows_*: raw pointers (int) of packed rows, host or device"""
        self._chk(self.L.nlps_gpu_migration_commit(self.h, C.c_void_p(rows_a) if rows_a else None, int(n_a),
                                                   C.c_void_p(rows_b) if rows_b else None, int(n_b)))
        self.num_particles()

    def set_ghost_bands(self, band_lo, band_hi, overlap=True):
        """overlap: False / 0 blocking exchanges, True / 1 split launches, 2 one launch per stage (library RCCL only)"""
        self._chk(self.L.nlps_gpu_set_ghost_bands(self.h, int(band_lo), int(band_hi), int(overlap)))

    def set_node_numbering(self, lattice_of_file):
        """Masked numbering in the node order of a mesh file (None = lattice order): include/nlps_gpu.h"""
        a = None if lattice_of_file is None else np.ascontiguousarray(lattice_of_file, dtype=np.int32)
        self._chk(self.L.nlps_gpu_set_node_numbering(self.h, _i(a)))

    def set_node_window(self, layer_lo, layer_hi):
        self._chk(self.L.nlps_gpu_set_node_window(self.h, int(layer_lo), int(layer_hi)))

    def set_timing(self, on=True):
        self._chk(self.L.nlps_gpu_set_timing(self.h, 1 if on else 0))

    def get_timing(self):
        ms = (C.c_float * 8)()
        self._chk(self.L.nlps_gpu_get_timing(self.h, ms))
        return list(ms)
