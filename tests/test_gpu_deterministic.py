"""nlps_gpu_set_deterministic: run-to-run bit-reproducibility of the fused explicit step (SURVEY §5 "race detection",
§7 hard part 2).  Every nodal sum is accumulated in a fixed order (sorted tile lists, one wave per tile, window slabs
combined in index order), so two runs of the same steps must agree BIT FOR BIT; the default path (LDS / global f64
atomics in arrival order) agrees with it to rounding only."""
import numpy as np
import pytest

from util import DP, NH, assert_close, dirichlet_plane, gpu_setup, make_case, nlps, oracle_setup, orc

pytestmark = pytest.mark.gpu

KEYS = ("x", "dis", "vel", "acc", "F_n", "DF", "Stress", "J_n", "rho", "W", "lambda", "b_e_n", "Kappa_n", "EPS_n")


def run(case, nsteps, deterministic, shuffle_seed=None, resort=0, mode=None):
    n = nlps()
    S = gpu_setup(case, nsteps=nsteps)
    S.set_deterministic(deterministic)
    S.set_resort_interval(resort)
    if mode is not None:
        S.set_law_launch_mode(mode)
    gb = n.BccSet([dirichlet_plane(case, case["ndim"] - 1, 2, nsteps)])
    grav = [0.0] * (case["ndim"] - 1) + [-9.81]
    E = max(m["E"] for m in case["materials"])
    dt = 0.1 * case["h"] / np.sqrt(E / 1000.0)
    nod = None
    for t in range(nsteps):
        S.explicit_step(gb, t, dt, 0.5, grav)
        nod = S.explicit_nodal()
    st = S.download_state()
    assert S.status_flags() == 0
    S.close()
    return st, nod


@pytest.mark.parametrize("ndim,material", [(3, NH), (2, NH), (3, DP)])
def test_two_runs_are_bit_identical(ndim, material):
    vel = [0.0] * (ndim - 1) + [-10.0 if material["type"] == 0 else -0.2]
    if ndim == 3:
        case = make_case(3, [14, 13, 12], [3, 3, 2], [8, 7, 7], material=material, velocity=vel)
    else:
        case = make_case(2, [40, 30], [3, 3], [34, 22], material=material, velocity=vel)
    a, na = run(case, 5, True, resort=3)
    b, nb = run(case, 5, True, resort=3)
    for k in KEYS:
        assert np.array_equal(a[k], b[k]), f"{k}: two deterministic runs differ"
    for k in ("mass", "dU", "force", "accel", "reaction"):
        assert np.array_equal(na[k], nb[k]), f"nodal {k}: two deterministic runs differ"
    # the default path gives the same answer to rounding
    c, nc = run(case, 5, False, resort=3)
    for k in ("x", "vel", "F_n", "Stress"):
        assert_close(c[k], a[k], 1e-11, f"{k}: atomic vs deterministic accumulation",
                     scale=(material["E"] * 1e-9 if k == "Stress" else None))
    assert np.array_equal(c["I0"], a["I0"])


def test_deterministic_mixed_laws_and_oracle_order():
    """Three laws in one cloud (per-law launches, ordered compaction) stay reproducible; the deterministic nodal sums
    agree with the oracle's single-thread summation order to a few ulp of the sum."""
    o = orc()
    case = make_case(3, [11, 10, 9], [3, 3, 2], [5, 4, 4], material=DP, velocity=[0.0, 0.0, -2.0])
    case["materials"] = [{"type": 0, "E": 2.0e4, "nu": 0.3}, {"type": 1, "E": 1.0e4, "nu": 0.25}, DP]
    npart = case["cloud"]["x"].shape[0]
    case["cloud"]["matidx"] = (np.arange(npart) % 3).astype(np.int32)
    a, na = run(case, 4, True)
    b, nb = run(case, 4, True)
    for k in KEYS:
        assert np.array_equal(a[k], b[k]), f"{k}: two deterministic runs differ"
    # oracle, one thread, particles in the caller's order
    M, P, prm, mats = oracle_setup(case)
    nsteps = 4
    bcs = [dirichlet_plane(case, 2, 2, nsteps)]
    stepper = o.ExplicitStepper(P, M, mats, prm, o.BccSet(bcs), nsteps, gravity=[0.0, 0.0, -9.81])
    dt = 0.1 * case["h"] / np.sqrt(2.0e4 / 1000.0)
    for t in range(nsteps):
        assert stepper.step(t, dt) == 0
    m_o, m_g = stepper.nodal("mass"), na["mass"]
    ulp = np.abs(m_g - m_o) / np.maximum(np.spacing(np.abs(m_o)), 1e-300)
    assert ulp.max() <= 64, f"lumped mass: {ulp.max():.0f} ulp from the oracle's summation order"
    assert_close(na["force"], stepper.nodal("force"), 1e-9, "nodal force")


def dense_case(copies):
    """`copies` jittered clouds over the same cells: 8 x copies particles per cell, i.e. as many per closest node."""
    case = make_case(3, [10, 10, 9], [3, 3, 2], [4, 4, 4], material=NH, velocity=[0.0, 0.0, -10.0])
    parts = [make_case(3, [10, 10, 9], [3, 3, 2], [4, 4, 4], material=NH, velocity=[0.0, 0.0, -10.0],
                       seed=100 + q, jitter=0.2)["cloud"] for q in range(copies)]
    cloud = {}
    for k, v in parts[0].items():
        if isinstance(v, np.ndarray):
            cloud[k] = np.ascontiguousarray(np.concatenate([c[k] for c in parts], axis=0))
        else:
            cloud[k] = v
    for k in ("mass", "vol0"):
        cloud[k] = cloud[k] / copies
    case["cloud"] = cloud
    return case


def test_deterministic_mode_with_more_particles_per_node_than_the_layer_table():
    """Clouds beyond the caps of the per-tile ordering (more than 32 particles on one closest node, more than 4096 in
    one tile) used to keep the arrival order of the binning atomics: deterministic mode now orders such tiles by slot
    index, so two runs still agree bit for bit."""
    for copies in (5, 10):  # 40 per node (layer table overflows); 80 per node and > 4096 in the full tiles
        case = dense_case(copies)
        a, na = run(case, 3, True, resort=2)
        b, nb = run(case, 3, True, resort=2)
        for k in KEYS:
            assert np.array_equal(a[k], b[k]), f"{k}: two deterministic runs differ ({copies} copies)"
        for k in ("mass", "dU", "force", "accel"):
            assert np.array_equal(na[k], nb[k]), f"nodal {k}: two deterministic runs differ ({copies} copies)"
        c, nc = run(case, 3, False, resort=2)
        for k in ("x", "vel", "F_n"):
            assert_close(c[k], a[k], 1e-11, f"{k}: atomic vs deterministic accumulation, dense cloud")
