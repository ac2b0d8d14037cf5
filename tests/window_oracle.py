"""Value-level oracle check of a WINDOW of a full-size run.  The oracle cannot step a million
particles, but one explicit step of a block of the cloud is a local computation: a particle's new state depends on
particles at most four stencil reaches (4 x 2.5 cells) away (P2G of mass / momentum -> G2P of grad dU -> P2G of the force -> G2P of the
acceleration).  So: download the pre-step state of the particles of a block of cells, rebuild exactly that block in the
oracle (a sub-lattice with its own node numbering, the downloaded I0 / lambda / beta as the warm start the reference's
search expects), step both once, and compare values -- particle fields for the particles deep enough inside the block
that nothing outside it can have reached them, nodal sums for the nodes whose whole support lies inside.
Reference: U-Verlet.c:229-253, 301-367, 530-676, 919-1084 (the step), LME.c:895-1015 (the search), U-Newmark-beta.c:528-597,
1257-1374 (lumped mass, internal forces)."""
import numpy as np

from util import assert_close, orc

R_CUT = 2.5           # reach of a particle along ONE axis, in cells: its list lives in the 5^d stencil of its closest node
                      # (|x - x_I0| <= h/2 per axis, two nodes beyond), and with the GramsBox h_avg the LME cut-off radius
                      # (3.04 h for gamma = 3, TOL_zero = 1e-6) does not shorten that along an axis
# depth (cells inside every face of the block that cuts the cloud) from which on a quantity is untouched by the cut:
DEPTH_NODAL_MASS = R_CUT + 0.1         # nodal mass / dU: every particle that lists the node is in the block
DEPTH_PARTICLE_F = 2 * R_CUT + 0.1     # x, dis, F, J, stress, lambda: gather of dU from such nodes
DEPTH_NODAL_FORCE = 3 * R_CUT + 0.1    # nodal force: scatter of such stresses
DEPTH_PARTICLE_ACC = 4 * R_CUT + 0.1   # vel, acc: gather of such forces
MESH_MARGIN = 4                        # nodes of the sub-lattice beyond the block (2-ring of every I0 + 1: interior h_avg)

PRE_FIELDS = ["x", "dis", "vel", "acc", "F_n", "J_n", "rho", "lambda", "beta", "I0", "b_e_n", "Kappa_n", "EPS_n"]
POST_FIELDS = ["x", "dis", "vel", "acc", "F_n", "J_n", "rho", "lambda", "beta", "I0", "Stress", "b_e_n", "Kappa_n", "EPS_n"]


def _depth(cells, blo, bhi, clo, chi):
    """distance (cells) to the nearest face of the block [blo, bhi) that cuts the cloud [clo, chi); inf if none does"""
    d = np.full(cells.shape[0], np.inf)
    for a in range(cells.shape[1]):
        if blo[a] > clo[a]:
            d = np.minimum(d, cells[:, a] - blo[a])
        if bhi[a] < chi[a]:
            d = np.minimum(d, bhi[a] - cells[:, a])
    return d


def window_step_check(S, case, gb, bcs_list, t, dt, gamma, grav, blocks, cloud_lo, cloud_hi, nsteps, tol=1e-10,
                      with_lists=True, label=""):
    """One explicit step of solver S (its step `t`), checked against the oracle on every block of `blocks`
    ([(lo cell, size in cells), ...] per axis lists).  cloud_lo / cloud_hi: extent of the cloud in cells at this step
    (a block face beyond it cuts nothing).  Returns the number of particles compared on all fields, per block."""
    o = orc()
    ndim, h = case["ndim"], case["h"]
    origin = np.asarray(case["origin"], dtype=np.float64)
    gn = np.asarray(case["grid_n"][:ndim])
    plastic = any(m["type"] in (2, 3) for m in case["materials"])
    pre = S.download_state(PRE_FIELDS)
    S.explicit_step(gb, t, dt, gamma, grav)
    assert S.status_flags() == 0
    post = S.download_state(POST_FIELDS)
    nod = S.explicit_nodal()
    n2m_g, _ = S.active_masks(gb, t)
    if with_lists:
        nn_g, _ = S.download_lists()
    cells_all = (pre["x"] - origin) / h
    counts = []
    for bi, (blo, bsz) in enumerate(blocks):
        blo = np.asarray(blo)
        bhi = blo + np.asarray(bsz)
        sel = np.flatnonzero(np.all((cells_all >= blo) & (cells_all < bhi), axis=1))
        assert sel.size > 0, "block %d holds no particle" % bi
        # ---- the block as an oracle problem of its own
        slo = np.maximum(blo - MESH_MARGIN, 0)
        shi = np.minimum(bhi + MESH_MARGIN, gn - 1)       # last node of the sub-lattice
        sn = (shi - slo + 1).astype(int)
        M = o.OracleMesh(ndim, [int(v) for v in sn], list(origin + slo * h), h)
        I0g = pre["I0"][sel]
        ijk = np.stack([(I0g // int(np.prod(gn[:a]))) % gn[a] for a in range(ndim)], axis=1)
        loc = ijk - slo
        assert np.all((loc >= 3) | (slo == 0)[None, :]), "closest nodes must keep their 2-ring + 1 inside the sub-lattice"
        I0s = sum(loc[:, a] * int(np.prod(sn[:a])) for a in range(ndim)).astype(np.int32)
        cloud = {"x": pre["x"][sel], "dis": pre["dis"][sel], "vel": pre["vel"][sel], "acc": pre["acc"][sel],
                 "F_n": pre["F_n"][sel], "b_e_n": pre["b_e_n"][sel], "J_n": pre["J_n"][sel], "rho": pre["rho"][sel],
                 "mass": case["cloud"]["mass"][sel], "vol0": case["cloud"]["vol0"][sel],
                 "kappa_n": pre["Kappa_n"][sel], "eps_n": pre["EPS_n"][sel], "matidx": case["cloud"]["matidx"][sel],
                 "I0": I0s, "lambda": pre["lambda"][sel], "beta": pre["beta"][sel]}
        P = o.OracleParticles(cloud)
        prm = o.default_params()
        mats = o.make_materials(case["materials"])
        sub_bcs = []
        for b in bcs_list:  # Dirichlet nodes that fall inside the sub-lattice, in its numbering
            g = np.asarray(b["nodes"], dtype=np.int64)
            gi = np.stack([(g // int(np.prod(gn[:a]))) % gn[a] for a in range(ndim)], axis=1)
            inside = np.all((gi >= slo) & (gi <= shi), axis=1)
            li = gi[inside] - slo
            nodes = sum(li[:, a] * int(np.prod(sn[:a])) for a in range(ndim)).astype(np.int32)
            sub_bcs.append(dict(b, nodes=np.sort(nodes)))
        stepper = o.ExplicitStepper(P, M, mats, prm, o.BccSet(sub_bcs), nsteps, grav)
        assert stepper.step(t, dt, gamma) == 0, "oracle step of block %d failed" % bi
        # ---- particles
        depth = _depth(cells_all[sel], blo, bhi, np.asarray(cloud_lo), np.asarray(cloud_hi))
        deep_f, deep_a = depth >= DEPTH_PARTICLE_F, depth >= DEPTH_PARTICLE_ACC
        assert deep_a.sum() >= 64, "block %d: only %d particles are deep enough for every field" % (bi, deep_a.sum())
        what = "%s block %d" % (label, bi)
        # index maps bit for bit: closest node (back in global numbering) and list length
        l2 = np.stack([(P["I0"] // int(np.prod(sn[:a]))) % sn[a] for a in range(ndim)], axis=1) + slo
        I0_back = sum(l2[:, a] * int(np.prod(gn[:a])) for a in range(ndim))
        assert np.array_equal(post["I0"][sel][deep_f], I0_back[deep_f]), what + ": I0 differs"
        if with_lists:
            assert np.array_equal(nn_g[sel][deep_f], P["nn"][deep_f]), what + ": NumberNodes differs"
        assert np.array_equal(post["beta"][sel][deep_f], P["beta"][deep_f]), what + ": beta differs"
        assert_close(post["lambda"][sel][deep_f], P["lambda"][deep_f], 1e-9, what + ": lambda")
        E = max(m["E"] for m in case["materials"])
        for k, ok, deep, scale in (("x", "x", deep_f, None), ("dis", "dis", deep_f, None), ("F_n", "F_n", deep_f, None),
                                   ("J_n", "J_n", deep_f, None), ("Stress", "stress", deep_f, E * 1e-6),
                                   ("rho", "rho", deep_f, None), ("vel", "vel", deep_a, None), ("acc", "acc", deep_a, None)):
            glob_scale = float(np.abs(post[k]).max())  # the field's magnitude over the WHOLE cloud
            assert_close(post[k][sel][deep], P[ok][deep], 1e-9 if k == "rho" else tol, "%s: %s" % (what, k),
                         scale=max(glob_scale, scale or 0.0))
        if plastic:
            for k, ok in (("b_e_n", "b_e_n"), ("Kappa_n", "kappa_n"), ("EPS_n", "eps_n")):
                assert_close(post[k][sel][deep_f], P[ok][deep_f], tol, "%s: %s" % (what, k), scale=1e-6 if k == "EPS_n" else None)
        # ---- nodes (masked numbering on both sides -> node by node through the two Nodes2Mask)
        nsub = int(np.prod(sn))
        sidx = np.arange(nsub)
        sijk = np.stack([(sidx // int(np.prod(sn[:a]))) % sn[a] for a in range(ndim)], axis=1) + slo
        gidx = sum(sijk[:, a] * int(np.prod(gn[:a])) for a in range(ndim))
        ndepth = _depth(sijk.astype(np.float64), blo, bhi, np.asarray(cloud_lo), np.asarray(cloud_hi))
        mo, mg = stepper.n2m[:nsub], n2m_g[gidx]
        for k, dmin in (("mass", DEPTH_NODAL_MASS), ("dU", DEPTH_NODAL_MASS), ("force", DEPTH_NODAL_FORCE),
                        ("accel", DEPTH_NODAL_FORCE), ("reaction", DEPTH_NODAL_FORCE)):
            pick = (ndepth >= dmin) & (mo >= 0)
            assert np.all(mg[pick] >= 0), what + ": a node active in the oracle block is not active on the device"
            a_o = stepper.nodal(k).reshape(-1, ndim)[mo[pick]]
            a_g = nod[k].reshape(-1, ndim)[mg[pick]]
            assert_close(a_g, a_o, tol, "%s: nodal %s" % (what, k), scale=float(np.abs(nod[k]).max()) or 1.0)
        # (active in the oracle block <=> active on the device, for nodes deep enough that every seed is in the block)
        inner = ndepth >= DEPTH_NODAL_MASS
        assert np.array_equal(mo[inner] >= 0, mg[inner] >= 0), what + ": ActiveNode differs"
        counts.append(int(deep_a.sum()))
    return counts


def window_residual_check(S, case, gb, bcs_list, step, nsteps, blocks, cloud_lo, cloud_hi, amp=1e-3, tol=1e-10, label=""):
    """nlps_gpu_lagrangian_evaluation of solver S (after its local search) against the oracle's composition of
    compatibility + constitutive update + internal forces on every block: the SAME nodal dU on both sides (so a node's
    force is right as soon as every particle that lists it is in the block with its true list: depth R_CUT + 1.5), the
    particle state the call leaves (DF, F_n1, J_n1, Stress) from depth 1.5 on.  alpha = 0 and no gravity: the residual is
    the internal force alone."""
    o = orc()
    ndim, h = case["ndim"], case["h"]
    origin = np.asarray(case["origin"], dtype=np.float64)
    gn = np.asarray(case["grid_n"][:ndim])
    pre = S.download_state(PRE_FIELDS)  # (before the search: the oracle block searches from the same state)
    S.local_search()
    n2m_g, d2m_g = S.active_masks(gb, step)
    na = S.nactive
    rng = np.random.default_rng(9)
    dU = amp * rng.normal(size=na * ndim)
    zero = np.zeros(na * ndim)
    one = np.ones(na * ndim)
    R_g = S.lagrangian_evaluation(dU, zero, zero, one, [0.0] * 6, None)
    post = S.download_state(["DF", "F_n1", "J_n1", "Stress"])
    cells_all = (pre["x"] - origin) / h
    counts = []
    for bi, (blo, bsz) in enumerate(blocks):
        blo = np.asarray(blo)
        bhi = blo + np.asarray(bsz)
        sel = np.flatnonzero(np.all((cells_all >= blo) & (cells_all < bhi), axis=1))
        slo = np.maximum(blo - MESH_MARGIN, 0)
        shi = np.minimum(bhi + MESH_MARGIN, gn - 1)
        sn = (shi - slo + 1).astype(int)
        M = o.OracleMesh(ndim, [int(v) for v in sn], list(origin + slo * h), h)
        I0g = pre["I0"][sel]
        ijk = np.stack([(I0g // int(np.prod(gn[:a]))) % gn[a] for a in range(ndim)], axis=1)
        loc = ijk - slo
        I0s = sum(loc[:, a] * int(np.prod(sn[:a])) for a in range(ndim)).astype(np.int32)
        cloud = {"x": pre["x"][sel], "dis": pre["dis"][sel], "vel": pre["vel"][sel], "acc": pre["acc"][sel],
                 "F_n": pre["F_n"][sel], "b_e_n": pre["b_e_n"][sel], "J_n": pre["J_n"][sel], "rho": pre["rho"][sel],
                 "mass": case["cloud"]["mass"][sel], "vol0": case["cloud"]["vol0"][sel],
                 "kappa_n": pre["Kappa_n"][sel], "eps_n": pre["EPS_n"][sel], "matidx": case["cloud"]["matidx"][sel],
                 "I0": I0s, "lambda": pre["lambda"][sel], "beta": pre["beta"][sel]}
        P = o.OracleParticles(cloud)
        prm = o.default_params()
        mats = o.make_materials(case["materials"])
        assert o.local_search(P, M, prm) == 0  # local_search__LME__ on the block (LME.c:895-1015)
        n2m_o, na_o = o.active_nodes(M)
        nsub = int(np.prod(sn))
        sidx = np.arange(nsub)
        sijk = np.stack([(sidx // int(np.prod(sn[:a]))) % sn[a] for a in range(ndim)], axis=1) + slo
        gidx = sum(sijk[:, a] * int(np.prod(gn[:a])) for a in range(ndim))
        act_o = n2m_o[:nsub] >= 0
        dU_o = np.zeros(na_o * ndim).reshape(-1, ndim)
        both = act_o & (n2m_g[gidx] >= 0)
        dU_o[n2m_o[:nsub][both]] = dU.reshape(-1, ndim)[n2m_g[gidx][both]]
        assert o.compatibility(dU_o.ravel(), None, P, M, n2m_o) == 0
        assert o.constitutive(P, mats, prm) == 0
        free = np.zeros(na_o * ndim, dtype=np.int32)  # (all dofs free: the Dirichlet dofs are compared as zeros below)
        R_o, st = o.internal_forces(P, M, n2m_o, free, na_o)
        assert st == 0
        what = "%s residual block %d" % (label, bi)
        ndepth = _depth(sijk.astype(np.float64), blo, bhi, np.asarray(cloud_lo), np.asarray(cloud_hi))
        pick = (ndepth >= R_CUT + 1.5) & act_o
        assert np.all(n2m_g[gidx][pick] >= 0) and pick.sum() > 500, what
        r_o = R_o.reshape(-1, ndim)[n2m_o[:nsub][pick]]
        r_g = R_g.reshape(-1, ndim)[n2m_g[gidx][pick]]
        fixed = d2m_g.reshape(-1, ndim)[n2m_g[gidx][pick]] == -1
        assert np.all(r_g[fixed] == 0.0), what + ": Dirichlet dofs"
        assert_close(np.where(fixed, 0.0, r_g), np.where(fixed, 0.0, r_o), tol, what + ": nodal residual",
                     scale=float(np.abs(R_g).max()))
        depth = _depth(cells_all[sel], blo, bhi, np.asarray(cloud_lo), np.asarray(cloud_hi))
        deep = depth >= 1.5
        E = max(m["E"] for m in case["materials"])
        for k, ok, scale in (("DF", "DF", None), ("F_n1", "F_n1", None), ("J_n1", "J_n1", None), ("Stress", "stress", E * amp)):
            assert_close(post[k][sel][deep], P[ok][deep], tol, "%s: %s" % (what, k), scale=scale)
        counts.append(int(pick.sum()))
    return counts
