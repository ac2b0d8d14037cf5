"""Shared helpers for the tests: package loading, scenarios, comparisons."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

synth = importlib.import_module("nl-partsol_amd.synth")


def nlps():
    return importlib.import_module("nl-partsol_amd.nlps")


def orc():
    from oracle import orc as _orc
    return _orc


NH = {"type": 0, "E": 1.0e7, "nu": 0.3}
HENCKY = {"type": 1, "E": 1.0e7, "nu": 0.3}
DP = synth.drucker_prager_material()
VM = synth.von_mises_material()


def make_case(ndim, cells, lo, blk, material=NH, velocity=None, jitter=0.05, ppc=None, seed=12345, h=1.0,
              origin=None, rho=1000.0, gamma=None, tol_zero=None):
    cells = list(cells)
    cloud = synth.make_cloud(ndim, cells, lo, blk, h=h, origin=origin, jitter=jitter, seed=seed, ppc=ppc,
                             velocity=velocity, rho=rho)
    if material["type"] == 2:  # Kappa_n = kappa_0 at start (InOutFun/Analysis/Generate-One-Phase-Analysis.c:621)
        cloud["kappa_n"][:] = material["kappa_0"]
    case = {"ndim": ndim, "cells": cells, "grid_n": synth.grid_nodes(cells),
            "origin": [0.0] * ndim if origin is None else list(origin), "h": h, "cloud": cloud,
            "materials": [material]}
    if gamma is not None or tol_zero is not None:  # LME globals other than the defaults of Read_GramsShapeFun.c:100-104
        case["lme"] = (3.0 if gamma is None else float(gamma), 1e-6 if tol_zero is None else float(tol_zero))
    return case


def oracle_setup(case, init=True):
    o = orc()
    M = o.OracleMesh(case["ndim"], case["grid_n"], case["origin"], case["h"])
    P = o.OracleParticles(case["cloud"])
    prm = o.default_params()
    if "lme" in case:
        prm.gamma_lme, prm.tol_zero_lme = case["lme"]
    mats = o.make_materials(case["materials"])
    if init:
        assert o.initialize_lme(P, M, prm) == 0
    return M, P, prm, mats


def gpu_setup(case, init=True, nsteps=1, **kw):
    n = nlps()
    if "lme" in case and "params" not in kw:
        kw["params"] = n.default_params()
        kw["params"].gamma_lme, kw["params"].tol_zero_lme = case["lme"]
    S = n.Solver(case["ndim"], case["grid_n"], case["origin"], case["h"], case["cloud"], case["materials"],
                 nsteps=nsteps, **kw)
    if init:
        S.initialise_shapefun()
    return S


def relerr(a, b, scale=None):
    """max |a-b| relative to the field's magnitude (or to `scale`, the natural size of the quantity, when
    the field itself is rounding noise, e.g. the stress E*O(eps) of a rigidly moving particle)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    s = max(np.max(np.abs(b)) if b.size else 0.0, 1e-300)
    if scale is not None:
        s = max(s, scale)
    return float(np.max(np.abs(a - b)) / s) if b.size else 0.0


def assert_close(a, b, tol, name, scale=None):
    e = relerr(a, b, scale)
    assert e <= tol, f"{name}: relative error {e:.3e} > {tol:.1e}"


def dirichlet_plane(case, axis, index, nsteps, dims=None, value=0.0):
    nodes = synth.plane_nodes(case["grid_n"], axis, index)
    d = case["ndim"]
    dirs = np.ones((d, nsteps), dtype=np.int32) if dims is None else np.asarray(dims, dtype=np.int32)
    vals = np.full((d, nsteps), value, dtype=np.float64)
    return {"nodes": nodes, "dim": d, "dir": dirs, "value": vals}


def free_port():
    """A TCP port nobody listens on right now (for torch.distributed rendezvous on 127.0.0.1)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]
