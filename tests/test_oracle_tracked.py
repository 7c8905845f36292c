"""The oracle's tracked-gradient evaluation (oracle/pdmp_oracle.c: spdmp_zigzag_tracked, `tracked=True`) against the oracle's moving
evaluation -- the line-by-line restatement of src/sfact.jl:73-145.  Both realise the same process with the same draws; the tracked one is
the BITWISE checker of the device's tracked kernels (tests/test_gpu_track_parity.py, tests/test_gpu_track_horizon.py), so what is pinned
here (CPU only) is that it is the same sampler: identical event indices, outcomes, counters and adapted bounds, floats within 1e-9
(observed ~1e-13), on every instantiation the kernels have -- plain, a start time, a looser bounding Γ with a target mean and adapt,
slices, a bound violation."""
import numpy as np
import scipy.sparse as sp

import oracle_lib as O

TOL = 1e-9


def close(a, b, tol=TOL):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and bool(np.all(np.abs(a - b) <= tol * np.maximum(1.0, np.maximum(np.abs(a), np.abs(b)))))


def same_chain(a, b):
    assert a["status"] == b["status"] and a["num"] == b["num"] and a["nacc"] == b["nacc"] and a["ndraw_main"] == b["ndraw_main"]
    assert len(a["events"]) == len(b["events"])
    assert np.array_equal(a["events"]["i"], b["events"]["i"]) and np.array_equal(a["events"]["theta"], b["events"]["theta"])
    assert close(a["events"]["t"], b["events"]["t"]) and close(a["events"]["x"], b["events"]["x"])
    assert np.array_equal(a["acc"], b["acc"]) and np.array_equal(a["theta"], b["theta"]) and np.array_equal(a["c"], b["c"])
    assert close(a["t"], b["t"]) and close(a["x"], b["x"])  # the rebuilt lazy clocks are the reference's (src/sfact.jl:211)


def test_tracked_equals_moving_on_lattices(pkg):
    for n, T, t0 in ((8, 40.0, 0.0), (16, 20.0, 0.0), (24, 8.0, 3.0)):
        G = pkg.problems.gmrf_precision(n)
        d = n * n
        rng = np.random.default_rng(n)
        x0, th0 = rng.standard_normal(d), rng.choice([-1.0, 1.0], d)
        c = pkg.problems.column_norms(G)
        a = O.spdmp_zigzag(G, None, G, x0, th0, c, T, seed=90 + n, t0=t0)
        b = O.spdmp_zigzag(G, None, G, x0, th0, c, T, seed=90 + n, t0=t0, tracked=True)
        assert a["status"] == 0 and len(a["events"]) > 1000
        same_chain(a, b)
        assert float(np.max(np.abs(a["events"]["t"] - b["events"]["t"]))) < 1e-11


def test_tracked_equals_moving_with_looser_bound_mean_and_adapt(pkg):
    """Bounding Γ = 0.9 Γ (test/maintest.jl:23: two pairs of sums), a target mean, bounds that start too small and adapt."""
    n = 16
    G = pkg.problems.gmrf_precision(n)
    d = n * n
    rng = np.random.default_rng(5)
    mu = 0.3 * rng.standard_normal(d)
    x0, th0 = rng.standard_normal(d), rng.choice([-1.0, 1.0], d)
    c = 0.2 * pkg.problems.column_norms(G)
    kw = dict(target_mu=mu, adapt=True, factor=1.8, seed=31)
    a = O.spdmp_zigzag(0.9 * G, mu, G, x0, th0, c, 12.0, **kw)
    b = O.spdmp_zigzag(0.9 * G, mu, G, x0, th0, c, 12.0, tracked=True, **kw)
    assert a["status"] == 0 and a["c"].max() > c.max()
    same_chain(a, b)


def test_tracked_slices_and_violation(pkg):
    n = 12
    G = pkg.problems.gmrf_precision(n)
    d = n * n
    rng = np.random.default_rng(1)
    x0, th0 = rng.standard_normal(d), rng.choice([-1.0, 1.0], d)
    c = pkg.problems.column_norms(G)
    a = O.spdmp_zigzag(G, None, G, x0, th0, c, 5.0, seed=3, stop_before_T=True)
    b = O.spdmp_zigzag(G, None, G, x0, th0, c, 5.0, seed=3, stop_before_T=True, tracked=True)
    assert np.all(a["events"]["t"] < 5.0)
    same_chain(a, b)
    # a bounding Γ below the target's with a tiny c, adapt off: both stop with BOUND_VIOLATED at the same proposal
    Gb = sp.csc_matrix(0.3 * G)
    cs = np.full(d, 1e-6)
    a = O.spdmp_zigzag(Gb, None, G, 3 * x0, th0, cs, 5.0, seed=3)
    b = O.spdmp_zigzag(Gb, None, G, 3 * x0, th0, cs, 5.0, seed=3, tracked=True)
    assert a["status"] == O.ORC_BOUND_VIOLATED == b["status"]
    assert a["num"] == b["num"] and a["nacc"] == b["nacc"] and np.array_equal(a["events"]["i"], b["events"]["i"])


def test_tracked_refuses_what_the_kernels_refuse(pkg):
    n = 8
    G = pkg.problems.gmrf_precision(n)
    d = n * n
    x0, th0 = np.ones(d), np.ones(d)
    c = np.ones(d)
    r = O.spdmp_zigzag(G, None, G, x0, th0, c, 1.0, seed=1, lambda_ref=0.1, tracked=True)
    assert r["status"] == 4  # ORC_BAD_INPUT: a refresh clock
    Gt = G.tolil()
    Gt[0, 5] = Gt[5, 0] = 0.1
    r = O.spdmp_zigzag(G, None, sp.csc_matrix(Gt), x0, th0, c, 1.0, seed=1, tracked=True)
    assert r["status"] == 4  # the target's pattern differs from the bound's


def test_tracked_bounds_under_the_logistic_target(pkg):
    """p->tracked with the subsampled logistic target (config C4): bounds from carried sums, the gradient still the moving evaluation
    (oracle/pdmp_oracle.c: spdmp_zigzag_tracked_lg).  Against the moving evaluation: the same 9 000 events in the same order (82 000
    proposals, adaptation on), times and positions to 1e-9; slicing changes nothing; what the kernels refuse is refused."""
    P = pkg.problems.logistic_problem(m=20)
    lg = dict(A=P["A"], At=P["At"], y=P["y"], ny=P["ny"], mu=P["mu"], gamma0=P["gamma0"], k=10)
    rng = np.random.default_rng(1)
    th0 = P["sigma"] * rng.choice([-1.0, 1.0], P["p"])
    kw = dict(seed=5, adapt=True, factor=5.0, logistic=lg, sigma=P["sigma"])
    a = O.spdmp_zigzag(P["Gdrop"], P["mu"], P["Gdrop"], P["x0"], th0, P["c"], 30.0, **kw)
    b = O.spdmp_zigzag(P["Gdrop"], P["mu"], P["Gdrop"], P["x0"], th0, P["c"], 30.0, tracked=True, **kw)
    assert a["status"] == b["status"] == 0 and a["num"] == b["num"] > 50000 and len(a["events"]) == len(b["events"]) > 5000
    assert np.array_equal(a["events"]["i"], b["events"]["i"]) and np.array_equal(a["events"]["theta"], b["events"]["theta"])
    assert np.allclose(a["events"]["t"], b["events"]["t"], rtol=1e-9, atol=0) and np.allclose(a["events"]["x"], b["events"]["x"], rtol=1e-9, atol=1e-9)
    assert np.array_equal(a["acc"], b["acc"]) and np.array_equal(a["c"], b["c"]) and a["ndraw_global"] == b["ndraw_global"]
    assert not np.array_equal(a["events"]["t"], b["events"]["t"])  # (another arithmetic: not bit-identical)
    # the tracked clocks are the coordinates' own: never later than the moving evaluation's, which also advance with the neighbours
    assert np.all(b["t"] <= a["t"] + 1e-12) and np.any(b["t"] < a["t"])
    r = O.spdmp_zigzag(P["Gdrop"], P["mu"], P["Gdrop"], P["x0"], th0, P["c"], 1.0, tracked=True, lambda_ref=0.5, **kw)
    assert r["status"] == 4  # ORC_BAD_INPUT: a refresh clock
