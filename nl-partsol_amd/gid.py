"""Input side of the path: GiD meshes -> lattice + particles, through the C-ABI host helpers of
csrc/nlps_io.cpp (no GPU needed).

Mirrors what the reference does before its first step: `GramsBox (Type=GID,File=...)` reads the background mesh
(Nodes/Read-GID-Mesh.c), `GramsSolid2D/3D (File=..., GPxElement=n)` turns every element of a body mesh into n
particles (InOutFun/Analysis/Generate-One-Phase-Analysis.c:142-250, 569-625)."""
import ctypes as C

import numpy as np

from . import nlps as _nlps


class GidInfo(C.Structure):
    _fields_ = [("ndim", C.c_int), ("nnodes", C.c_int), ("nelem", C.c_int), ("nodes_per_elem", C.c_int),
                ("elem_type", C.c_char * 32)]


def _check(st, what):
    if st:
        f = _nlps.lib().nlps_host_io_last_error
        f.restype = C.c_char_p
        raise _nlps.NlpsError("%s: %s" % (what, f().decode()))


def read_gid_mesh(path):
    """-> dict(ndim, elem_type, coords[nnodes][ndim], conn[nelem][npe]); conn is 0-based and in the reference's chain
    order (the reverse of the file order, see include/nlps_gpu.h)."""
    L = _nlps.lib()
    info = GidInfo()
    L.nlps_host_gid_mesh_info.argtypes = [C.c_char_p, C.POINTER(GidInfo)]
    _check(L.nlps_host_gid_mesh_info(str(path).encode(), C.byref(info)), "nlps_host_gid_mesh_info")
    coords = np.zeros((info.nnodes, info.ndim))
    conn = np.zeros((info.nelem, info.nodes_per_elem), dtype=np.int32)
    L.nlps_host_gid_mesh_read.argtypes = [C.c_char_p, C.POINTER(GidInfo), C.c_void_p, C.c_void_p]
    _check(L.nlps_host_gid_mesh_read(str(path).encode(), C.byref(info), coords.ctypes.data_as(C.c_void_p),
                                     conn.ctypes.data_as(C.c_void_p)), "nlps_host_gid_mesh_read")
    return {"ndim": info.ndim, "elem_type": info.elem_type.decode(), "coords": coords, "conn": conn, "_info": info}


def lattice_from_nodes(coords):
    """-> (h, n[ndim], origin[ndim], canon[nnodes]): the GramsBox lattice behind the nodes; canon maps a file node
    to the lattice numbering (x fastest) the library's node-indexed arrays use."""
    coords = np.ascontiguousarray(coords, dtype=np.float64)
    nn, nd = coords.shape
    h = C.c_double()
    n = (C.c_int * 3)()
    o = (C.c_double * 3)()
    canon = np.zeros(nn, dtype=np.int32)
    f = _nlps.lib().nlps_host_lattice_from_nodes
    f.argtypes = [C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int * 3, C.c_double * 3, C.c_void_p]
    _check(f(nd, nn, coords.ctypes.data_as(C.c_void_p), C.byref(h), n, o, canon.ctypes.data_as(C.c_void_p)),
           "nlps_host_lattice_from_nodes")
    return h.value, list(n)[:nd], list(o)[:nd], canon


def particles_from_mesh(mesh, gp_per_elem, thickness=1.0):
    """-> (x[nelem * gp][ndim], vol0[nelem * gp]) of a Quadrilateral / Hexahedra body mesh."""
    info = mesh["_info"]
    npart = info.nelem * int(gp_per_elem)
    x = np.zeros((npart, info.ndim))
    vol0 = np.zeros(npart)
    coords = np.ascontiguousarray(mesh["coords"], dtype=np.float64)
    conn = np.ascontiguousarray(mesh["conn"], dtype=np.int32)
    f = _nlps.lib().nlps_host_particles_from_mesh
    f.argtypes = [C.POINTER(GidInfo), C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
    _check(f(C.byref(info), coords.ctypes.data_as(C.c_void_p), conn.ctypes.data_as(C.c_void_p), int(gp_per_elem),
             float(thickness), x.ctypes.data_as(C.c_void_p), vol0.ctypes.data_as(C.c_void_p)),
           "nlps_host_particles_from_mesh")
    return x, vol0


def cloud_from_mesh(mesh, gp_per_elem, rho, thickness=1.0, velocity=None, matidx=0, kappa_0=0.0):
    """The particle arrays nlps.Solver takes (the dict of synth.make_cloud) from a body mesh: positions and volumes
    as above, mass = V rho and kappa_n = kappa_0 (Generate-One-Phase-Analysis.c:607-622), the rest at the values of
    U-Analisys.c:33-105 (F = b_e = I, J = 1, zero fields)."""
    from . import synth
    x, vol0 = particles_from_mesh(mesh, gp_per_elem, thickness)
    n, nd = x.shape
    vel = np.zeros((n, nd))
    if velocity is not None:
        vel[:] = np.asarray(velocity, dtype=np.float64)
    return {"ndim": nd, "x": x, "dis": np.zeros((n, nd)), "vel": vel, "acc": np.zeros((n, nd)),
            "F_n": synth.identity_rows(n, nd), "b_e_n": synth.identity_rows(n, nd), "J_n": np.ones(n),
            "rho": np.full(n, float(rho)), "mass": vol0 * float(rho), "vol0": vol0,
            "kappa_n": np.full(n, float(kappa_0)), "eps_n": np.zeros(n), "matidx": np.full(n, int(matidx), dtype=np.int32)}


class VtkFields(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("x", "mass", "rho", "I0", "matidx", "vel", "acc", "dis", "stress", "F_n", "W",
                                          "eps")]


VTK_X_GC, VTK_P, VTK_ENERGY = 1, 2, 4


def write_particles_vtk(path, results_time_step, state, flags=0):
    """The particle file of particle_results_vtk__InOutFun__ (InOutFun/Outputs/WriteVtk.c:95-266) from a dict in the
    layout of Solver.download_state(): keys x (required), mass, rho, I0, matidx, vel, acc, dis, Stress, F_n, W, EPS_n;
    missing keys leave their block out."""
    x = np.ascontiguousarray(state["x"], dtype=np.float64)
    npart, nd = x.shape
    keep = [x]
    f = VtkFields()
    f.x = x.ctypes.data
    for field, key, dt in (("mass", "mass", np.float64), ("rho", "rho", np.float64), ("I0", "I0", np.int32),
                           ("matidx", "matidx", np.int32), ("vel", "vel", np.float64), ("acc", "acc", np.float64),
                           ("dis", "dis", np.float64), ("stress", "Stress", np.float64), ("F_n", "F_n", np.float64),
                           ("W", "W", np.float64), ("eps", "EPS_n", np.float64)):
        if state.get(key) is not None:
            a = np.ascontiguousarray(state[key], dtype=dt)
            assert a.shape[0] == npart, key
            keep.append(a)
            setattr(f, field, a.ctypes.data)
    fn = _nlps.lib().nlps_host_write_particles_vtk
    fn.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(VtkFields), C.c_int]
    _check(fn(str(path).encode(), int(results_time_step), nd, npart, C.byref(f), int(flags)),
           "nlps_host_write_particles_vtk")


def write_nodes_vtk(path, mesh, canon, active, reactions):
    """The nodal file of nodal_results_vtk__InOutFun__ (WriteVtk.c:269-405): mesh (read_gid_mesh) in file numbering,
    active[nnodes] / reactions[nnodes][ndim] in the library's lattice numbering, canon from lattice_from_nodes (or None)."""
    info = mesh["_info"]
    coords = np.ascontiguousarray(mesh["coords"], dtype=np.float64)
    conn = np.ascontiguousarray(mesh["conn"], dtype=np.int32)
    act = np.ascontiguousarray(active, dtype=np.uint8)
    rea = np.ascontiguousarray(reactions, dtype=np.float64)
    assert act.shape[0] == info.nnodes and rea.shape == (info.nnodes, info.ndim)
    cn = None if canon is None else np.ascontiguousarray(canon, dtype=np.int32)
    f = _nlps.lib().nlps_host_write_nodes_vtk
    f.argtypes = [C.c_char_p, C.POINTER(GidInfo)] + [C.c_void_p] * 5
    _check(f(str(path).encode(), C.byref(info), coords.ctypes.data, conn.ctypes.data, None if cn is None else cn.ctypes.data,
             act.ctypes.data, rea.ctypes.data), "nlps_host_write_nodes_vtk")


class Deck(C.Structure):
    _fields_ = [("box_mesh", C.c_char * 512), ("body_mesh", C.c_char * 512), ("gp_per_elem", C.c_int),
                ("scheme", C.c_char * 64), ("CFL", C.c_double), ("Cel", C.c_double), ("i0", C.c_int), ("N", C.c_int),
                ("epsilon_mass_matrix", C.c_double), ("beta_newmark", C.c_double), ("gamma_newmark", C.c_double),
                ("tol_newmark", C.c_double), ("rb_generalized_alpha", C.c_double), ("tol_generalized_alpha", C.c_double),
                ("max_iter", C.c_int), ("explicit_trial", C.c_int), ("shape_fun", C.c_char * 16),
                ("gamma_lme", C.c_double), ("tol_zero_lme", C.c_double), ("tol_wrapper_lme", C.c_double),
                ("max_iter_lme", C.c_int), ("wrapper_lme", C.c_char * 32)]


def read_deck(path):
    """The subset of a .nlp command file the path needs before its first step (see include/nlps_gpu.h) as a dict."""
    d = Deck()
    f = _nlps.lib().nlps_host_read_deck
    f.argtypes = [C.c_char_p, C.POINTER(Deck)]
    _check(f(str(path).encode(), C.byref(d)), "nlps_host_read_deck")
    out = {}
    for name, _ in Deck._fields_:
        v = getattr(d, name)
        out[name] = v.decode() if isinstance(v, bytes) else v
    return out


def read_materials(path, max_materials=16):
    """The Define-Material blocks of a command file -> list of (idx, rho, material dict as nlps.Solver takes it)."""
    mats = (_nlps.Material * max_materials)()
    rho = (C.c_double * max_materials)()
    idx = (C.c_int * max_materials)()
    n = C.c_int(0)
    f = _nlps.lib().nlps_host_read_materials
    f.argtypes = [C.c_char_p, C.c_int, C.POINTER(_nlps.Material), C.POINTER(C.c_double), C.POINTER(C.c_int),
                  C.POINTER(C.c_int)]
    _check(f(str(path).encode(), max_materials, mats, rho, idx, C.byref(n)), "nlps_host_read_materials")
    out = []
    for i in range(n.value):
        m = {k: getattr(mats[i], k) for k, _ in _nlps.Material._fields_}
        m["a_borja"] = tuple(m["a_borja"])
        out.append((idx[i], rho[i], m))
    return out


def read_boundaries(path, ndim, nsteps, gp_per_elem=0):
    """The GramsBoundary blocks of a command file -> list of dicts in the layout nlps.BccSet takes (nodes in FILE
    numbering and in the reference's reversed file order: map them through canon of lattice_from_nodes).  With
    gp_per_elem > 0: the Define-Neumann-Boundary contours instead, nodes = particle indices."""
    L = _nlps.lib()
    L.nlps_host_read_boundaries.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)] + [C.c_void_p] * 4
    L.nlps_host_read_neumann.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)] + [C.c_void_p] * 4
    if gp_per_elem:
        f = lambda p_, nd_, ns_, *rest: L.nlps_host_read_neumann(p_, nd_, ns_, int(gp_per_elem), *rest)  # noqa: E731
    else:
        f = L.nlps_host_read_boundaries
    nb = C.c_int(0)
    _check(f(str(path).encode(), ndim, nsteps, 0, 0, C.byref(nb), None, None, None, None), "nlps_host_read_boundaries")
    if nb.value == 0:
        return []
    cnt = np.zeros(nb.value, dtype=np.int32)
    _check(f(str(path).encode(), ndim, nsteps, nb.value, 0, C.byref(nb), cnt.ctypes.data, None, None, None),
           "nlps_host_read_boundaries")
    total = int(cnt.sum())
    nodes = np.zeros(max(total, 1), dtype=np.int32)
    dirs = np.zeros((nb.value, ndim, nsteps), dtype=np.int32)
    vals = np.zeros((nb.value, ndim, nsteps))
    _check(f(str(path).encode(), ndim, nsteps, nb.value, total, C.byref(nb), cnt.ctypes.data, nodes.ctypes.data,
             dirs.ctypes.data, vals.ctypes.data), "nlps_host_read_boundaries")
    out, at = [], 0
    for b in range(nb.value):
        out.append({"nodes": nodes[at:at + cnt[b]].copy(), "dim": ndim, "dir": dirs[b].copy(), "value": vals[b].copy()})
        at += cnt[b]
    return out


def read_initials(path, gp_per_elem, vel):
    """Applies the GramsInitials blocks of a command file to vel[nparticles][ndim] (in place) and returns it."""
    vel = np.ascontiguousarray(vel, dtype=np.float64)
    f = _nlps.lib().nlps_host_read_initials
    f.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    _check(f(str(path).encode(), vel.shape[1], int(gp_per_elem), vel.shape[0], vel.ctypes.data), "nlps_host_read_initials")
    return vel


def read_gravity(path, ndim, nsteps):
    """-> g[nsteps][ndim] of the gravity-field block of a command file, or None if it has none."""
    g = np.zeros((nsteps, ndim))
    found = C.c_int(0)
    f = _nlps.lib().nlps_host_read_gravity
    f.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
    _check(f(str(path).encode(), ndim, nsteps, g.ctypes.data, C.byref(found)), "nlps_host_read_gravity")
    return g if found.value else None


class Outputs(C.Structure):
    _fields_ = [("found", C.c_int), ("results_time_step", C.c_int), ("dir", C.c_char * 512),
                ("particles_file", C.c_char * 128), ("nodes_file", C.c_char * 128)] + \
               [(k, C.c_int) for k in ("global_coordinates", "mass", "density", "nodal_idx", "material_idx", "velocity",
                                       "acceleration", "displacement", "stress", "volumetric_stress",
                                       "deformation_gradient", "energy", "eps", "unsupported")]


def read_outputs(path):
    """The GramsOutputs block of a command file as a dict, or None if it has none."""
    o = Outputs()
    f = _nlps.lib().nlps_host_read_outputs
    f.argtypes = [C.c_char_p, C.POINTER(Outputs)]
    _check(f(str(path).encode(), C.byref(o)), "nlps_host_read_outputs")
    if not o.found:
        return None
    return {k: (getattr(o, k).decode() if isinstance(getattr(o, k), bytes) else getattr(o, k)) for k, _ in Outputs._fields_}


def write_selected_particles_vtk(outputs, time_step, state):
    """particle_results_vtk__InOutFun__ with the switches of a GramsOutputs block: <DIR>/<Particles-file>_<step>.vtk."""
    keep = {"x": state["x"]}
    for flag, key in (("mass", "mass"), ("density", "rho"), ("nodal_idx", "I0"), ("material_idx", "matidx"),
                      ("velocity", "vel"), ("acceleration", "acc"), ("displacement", "dis"), ("stress", "Stress"),
                      ("deformation_gradient", "F_n"), ("eps", "EPS_n")):
        if outputs[flag] and key in state:
            keep[key] = state[key]
    flags = VTK_X_GC * bool(outputs["global_coordinates"]) | VTK_P * bool(outputs["volumetric_stress"])
    if outputs["volumetric_stress"] and "Stress" not in keep:
        keep["Stress"] = state["Stress"]  # (the reference writes P without STRESS; here P rides on the stress block)
    if outputs["energy"]:
        flags |= VTK_ENERGY
        keep.update({k: state[k] for k in ("W", "vel", "mass")})
    name = "%s/%s_%d.vtk" % (outputs["dir"], outputs["particles_file"], time_step)
    write_particles_vtk(name, outputs["results_time_step"], keep, flags)
    return name


def read_neumann(path, ndim, nsteps, gp_per_elem):
    """The Define-Neumann-Boundary contours of a command file (nodes = particle indices), for Solver.nodal_traction_forces."""
    return read_boundaries(path, ndim, nsteps, gp_per_elem)


def read_material_assignment(path, gp_per_elem, nmaterials, matidx):
    """Applies the Assign-material-to-particles lines of a command file to matidx[nparticles] (in place)."""
    m = np.ascontiguousarray(matidx, dtype=np.int32)
    f = _nlps.lib().nlps_host_read_material_assignment
    f.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    _check(f(str(path).encode(), int(gp_per_elem), int(nmaterials), m.shape[0], m.ctypes.data),
           "nlps_host_read_material_assignment")
    return m
