"""Sticky ZigZag (sspdmp, src/ss_fact.jl) on gfx950 vs the CPU oracle, through the C ABI (-m gpu)."""
import math

import numpy as np
import pytest
import scipy.sparse as sp

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["spec", "seq"])
def kernel_mode(request, monkeypatch):
    """Every case runs twice: on the speculative 4-events-per-iteration kernel (default where the neighbourhoods allow it) and
    with PDMP_KERNEL=seq on the one-event-per-iteration kernel; both must equal the oracle bit for bit."""
    if request.param == "seq":
        monkeypatch.setenv("PDMP_KERNEL", "seq")
    else:
        monkeypatch.delenv("PDMP_KERNEL", raising=False)
    return request.param


def check(pkg, Gb, G, mu_t, x0, th0, c, kappa, T, seed, adapt=False, reversible=False, strong=False, mu_b=None):
    d = G.shape[0]
    Z = pkg.ZigZag(Gb, np.zeros(d) if mu_b is None else mu_b)
    tr, (t, x, th), (acc, num), cout = pkg.sspdmp(pkg.GaussianTarget(G, mu_t), 0.0, x0, th0, T, c, Z, kappa, seed=seed,
                                                  adapt=adapt, reversible=reversible, strong_upperbounds=strong)
    for k in range(x0.shape[0]):
        r = O.sspdmp_zigzag(Gb, mu_b, G, x0[k], th0[k], c, kappa, T, target_mu=mu_t, seed=seed + k, adapt=adapt,
                            reversible=reversible, strong_upperbounds=strong)
        assert r["status"] == 0
        ev, oe = tr[k].events, r["events"]
        assert len(ev) == len(oe), (k, len(ev), len(oe))
        for f in ("i", "t", "x", "theta"):
            assert np.array_equal(ev[f], oe[f]), (k, f)
        assert (int(acc[k]), int(num[k])) == (r["nacc"], r["num"])
        assert np.array_equal(t[k], r["t"]) and np.array_equal(x[k], r["x"]) and np.array_equal(th[k], r["theta"])
        assert np.array_equal(cout[k], r["c"])
    return tr


def test_sticky_1d_reference_parameters_and_statistics(gpu_pkg):
    """test/sticky.jl:7-36: σ² = 0.5, μ = 0.9, flow Γ = [1], c = 20, κ = 1.5, T = 2000; closed-form weight w (:30)."""
    pkg = gpu_pkg
    Gf = sp.csc_matrix(np.array([[1.0]]))
    Gt = sp.csc_matrix(np.array([[2.0]]))
    x0, th0 = np.array([[1.0], [1.0]]), np.array([[0.8], [0.8]])
    T = 2000.0
    tr = check(pkg, Gf, Gt, np.array([0.9]), x0, th0, np.array([20.0]), np.array([1.5]), T, seed=1)
    ts, xs = pkg.trace.discretize(tr[0], 0.2)
    x = xs[:, 0]
    w = math.sqrt(2 * math.pi * 0.5) / (math.sqrt(2 * math.pi * 0.5) + math.exp(-0.5 * 0.81 / 0.5) / 1.5)
    assert abs(np.mean(x != 0) - w) < 2.5 / math.sqrt(T)
    assert abs(np.mean(x) - w * 0.9) < 5.0 / math.sqrt(T)


def test_sticky_d8_and_grid(gpu_pkg):
    pkg = gpu_pkg
    G = pkg.problems.maintest_precision(8)
    rng = np.random.default_rng(1)
    x0 = rng.random((4, 8))
    th0 = rng.choice([-1.0, -0.5, 0.5, 1.0], (4, 8))
    c = 0.7 * pkg.problems.column_norms(G)
    check(pkg, 0.9 * G, G, None, x0, th0, c, np.full(8, 1000.0), 200.0, seed=12, adapt=True)  # test/sticky.jl:39-65
    check(pkg, 0.9 * G, G, None, x0, th0, c, np.full(8, 0.5), 200.0, seed=13, adapt=True)     # real sticking
    check(pkg, 0.9 * G, G, None, x0, th0, c, np.full(8, 0.5), 100.0, seed=14, adapt=True, reversible=True)
    check(pkg, 0.9 * G, G, None, x0, th0, c, np.full(8, 0.5), 100.0, seed=15, adapt=True, strong=True)
    G2 = pkg.problems.gmrf_precision(12)  # d = 144: three key blocks
    d = 144
    x0 = rng.standard_normal((3, d))
    th0 = rng.choice([-1.0, 1.0], (3, d))
    check(pkg, G2, G2, None, x0, th0, pkg.problems.column_norms(G2), 0.3 + rng.random(d), 25.0, seed=16)
    mu = 0.3 * rng.standard_normal(d)
    check(pkg, G2, G2, mu, x0, th0, pkg.problems.column_norms(G2), np.full(d, 1.0), 15.0, seed=17, mu_b=mu)


def test_golden_sticky1d(gpu_pkg, golden):
    pkg = gpu_pkg
    Gf = sp.csc_matrix(np.array([[1.0]]))
    Gt = sp.csc_matrix(np.array([[2.0]]))
    tr, _, (acc, num), _ = pkg.sspdmp(pkg.GaussianTarget(Gt, np.array([0.9])), 0.0, np.array([1.0]), np.array([0.8]), 200.0,
                                       np.array([20.0]), pkg.ZigZag(Gf, np.zeros(1)), np.array([1.5]), seed=5)
    for f in ("t", "i", "x", "theta"):
        assert np.array_equal(tr.events[f], golden["sticky1d_events"][f])
    assert [int(num), int(acc)] == golden["sticky1d_counts"].tolist()


def test_sticky_p10000_variable_selection_scale(gpu_pkg):
    """The scale of config C5 (p = 10 000, spike-and-slab variable selection): sticky ZigZag on a sparse Gaussian slab
    (100 x 100 grid-Laplace precision), thaw rates κ = (γ0/√2π)/(1/w − 1) with w = 1/2 (scripts/sticky/
    sticky_logistic_sparse.jl:194-197).  Chains match the oracle; a sizeable fraction of coordinates is stuck at 0."""
    pkg = gpu_pkg
    G = pkg.problems.gmrf_precision(100, eps=0.5)
    d = G.shape[0]
    rng = np.random.default_rng(5)
    x0 = rng.standard_normal((2, d))
    th0 = rng.choice([-1.0, 1.0], (2, d))
    gamma0, w = 0.5, 0.5
    kappa = np.full(d, (gamma0 / math.sqrt(2 * math.pi)) / (1 / w - 1))
    tr = check(pkg, G, G, None, x0, th0, pkg.problems.column_norms(G), kappa, 3.0, seed=50)
    frozen = np.mean([np.mean(np.abs(q.events["theta"][-2000:]) == 0) for q in tr])
    assert 0.05 < frozen < 0.95


def test_sticky_trace_refill_and_time_slices(gpu_pkg):
    """A 50-event trace buffer forces many TRACE_FULL stops and resumes inside one sspdmp call; STOP_BEFORE slices followed by
    the reference tail give the same chain (both kernels, via the autouse fixture)."""
    pkg = gpu_pkg
    G = pkg.problems.gmrf_precision(12, eps=0.5)
    d = G.shape[0]
    rng = np.random.default_rng(77)
    x0 = rng.standard_normal((3, d))
    th0 = rng.choice([-1.0, 1.0], (3, d))
    c = pkg.problems.column_norms(G)
    kappa = np.full(d, 0.3)
    T = 6.0
    Z = pkg.ZigZag(G, np.zeros(d))
    tr, (t, x, th), (acc, num), _ = pkg.sspdmp(pkg.GaussianTarget(G), 0.0, x0, th0, T, c, Z, kappa, seed=300, trace_capacity=50)
    refs = [O.sspdmp_zigzag(G, None, G, x0[k], th0[k], c, kappa, T, seed=300 + k) for k in range(3)]
    for k, r in enumerate(refs):
        assert r["status"] == 0 and len(r["events"]) > 200
        for f in ("i", "t", "x", "theta"):
            assert np.array_equal(tr[k].events[f], r["events"][f]), (k, f)
        assert (int(acc[k]), int(num[k])) == (r["nacc"], r["num"]) and np.array_equal(x[k], r["x"]) and np.array_equal(t[k], r["t"])
    with pkg.Ensemble(3, d, sampler=pkg._lib.SAMPLER_STICKY_ZIGZAG, trace_capacity=4096) as ens:
        ens.set_flow(Z)
        ens.set_target(pkg.GaussianTarget(G))
        ens.set_sticky(kappa)
        ens.set_state(0.0, x0, th0, c, np.arange(3, dtype=np.uint64) + 300)
        for Tk in (1.5, 1.5, 4.0):  # a repeated boundary is a no-op
            ens.run(Tk, pkg._lib.RUN_STOP_BEFORE)
        ens.run(T, pkg._lib.RUN_REFERENCE_TAIL)
        cnt = ens.counters()
        for k, r in enumerate(refs):
            ev = ens.trace(k, counters=cnt)
            assert len(ev) == len(r["events"]) and np.array_equal(ev["t"], r["events"]["t"]) and np.array_equal(ev["i"], r["events"]["i"])


def test_sticky_on_large_neighbourhoods(gpu_pkg, kernel_mode):
    """sspdmp on graphs whose two-hop zones exceed one wavefront (general kernel): random sparse precision with a dense hub column
    (the shape of a regression's intercept), adapt / reversible / strong_upperbounds variants."""
    if kernel_mode == "seq":
        pytest.skip("one kernel serves this case")
    pkg = gpu_pkg
    rng = np.random.default_rng(31)
    d = 120
    R = sp.random(d, d, density=0.04, random_state=rng, data_rvs=rng.standard_normal, format="lil")
    R[0, :] = 0.3 * rng.standard_normal(d)  # hub: coordinate 0 neighbours everything
    A = sp.csc_matrix(R)
    G = sp.csc_matrix(A.T @ A + 2.0 * sp.identity(d))
    G.sort_indices()
    assert np.diff(G.indptr).max() > 64
    x0 = rng.standard_normal((2, d))
    th0 = rng.choice([-1.0, 1.0], (2, d))
    c = 2.0 * pkg.problems.column_norms(G)
    kappa = rng.uniform(0.2, 1.5, d)
    check(pkg, G, G, None, x0, th0, c, kappa, 6.0, seed=400, adapt=True)
    check(pkg, 0.9 * G, G, 0.1 * rng.standard_normal(d), x0, th0, c, kappa, 4.0, seed=401, adapt=True, reversible=True)
    tr = check(pkg, G, G, None, x0, th0, c, kappa, 4.0, seed=402, adapt=True, strong=True)
    frozen = np.mean([np.mean(q.events["theta"] == 0) for q in tr])
    assert 0.05 < frozen < 0.6  # freeze events are a sizeable share of the trace


def test_sticky_logistic_spike_and_slab(gpu_pkg, kernel_mode):
    """sspdmp(∇ϕmoving, t0, x0, θ0, T, c, Zdrop, κ, SelfMoving(), A, At, μ, y, ny, k; adapt=true): the sticky sampler on the
    subsampled logistic target of scripts/logistic.jl (the model of scripts/sticky/sticky_logistic_sparse.jl with the stock ZigZag
    bound instead of its hand-written MyBoundLog), κ = (γ0/√2π)/(1/w − 1), w = 1/2 (:194-197)."""
    if kernel_mode == "seq":
        pytest.skip("one kernel serves this case")
    pkg = gpu_pkg
    L = pkg.problems.logistic_problem(m=20)
    p = L["p"]
    rng = np.random.default_rng(12)
    nch, T = 2, 3.0
    X0 = np.tile(L["x0"], (nch, 1))
    TH0 = L["sigma"] * rng.choice([-1.0, 1.0], (nch, p))
    kappa = np.full(p, (L["gamma0"] / math.sqrt(2 * math.pi)) / (1 / 0.5 - 1))
    Z = pkg.ZigZag(L["Gdrop"], L["mu"], L["sigma"])
    target = pkg.LogisticTarget(L["A"], L["y"], L["ny"], L["mu"], L["gamma0"], 10)
    tr, (t, x, th), (acc, num), cout = pkg.sspdmp(target, 0.0, X0, TH0, T, L["c"], Z, kappa, seed=500, adapt=True, factor=5.0)
    lg = dict(A=L["A"], At=L["At"], y=L["y"], ny=L["ny"], mu=L["mu"], gamma0=L["gamma0"], k=10)
    for k in range(nch):
        r = O.sspdmp_zigzag(L["Gdrop"], L["mu"], L["Gdrop"], X0[k], TH0[k], L["c"], kappa, T, seed=500 + k, adapt=True, factor=5.0,
                            logistic=lg)
        assert r["status"] == 0 and len(r["events"]) > 200
        ev, oe = tr[k].events, r["events"]
        assert len(ev) == len(oe), (k, len(ev), len(oe))
        for f in ("i", "t", "x", "theta"):
            assert np.array_equal(ev[f], oe[f]), (k, f)
        assert (int(acc[k]), int(num[k])) == (r["nacc"], r["num"]) and np.array_equal(cout[k], r["c"])
        assert np.array_equal(x[k], r["x"]) and np.array_equal(th[k], r["theta"]) and np.array_equal(t[k], r["t"])
        assert 0.2 < np.mean(r["theta"] == 0) < 0.95  # a good share of the coefficients sits in the spike


@pytest.mark.parametrize("n,T", [(46, 6.0), (100, 1.5)])
def test_sticky_on_larger_lattices(gpu_pkg, n, T):
    """Lattices of several key blocks (d = 2116 is not a multiple of 64; d = 10 000 is config C5's size)."""
    pkg = gpu_pkg
    G = pkg.problems.gmrf_precision(n)
    d = n * n
    rng = np.random.default_rng(n)
    x0 = rng.standard_normal((3, d))
    th0 = rng.choice([-1.0, 1.0], (3, d))
    check(pkg, G, G, None, x0, th0, pkg.problems.column_norms(G), 0.3 + rng.random(d), T, seed=500 + n)
