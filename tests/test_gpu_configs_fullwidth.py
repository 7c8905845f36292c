"""The secondary configurations of BASELINE.json at their FULL ensemble widths (-m gpu), in the style of test_gpu_fullsize.py (C3):
   C2  4096 bouncy-particle chains on the isotropic d = 1024 Gaussian,
   C4  8192 chains (one GPU's share of 65 536) of the subsampled sparse logistic regression n = 8840, p = 442,
   C5  sticky ZigZag on the logistic spike-and-slab with p = 10 000 coefficients (scripts/exampledesign.jl design scaled to 10⁴
       columns, κ of scripts/sticky/sticky_logistic_sparse.jl:197), 2 chains bit-exact + a 4096-chain run.
First and last chain of every ensemble are compared with the oracle bit for bit; the rest through properties that do not depend on
the width (all chains healthy, traces reconstruct the final state, counters add up, seeds are disjoint)."""
import math

import numpy as np
import pytest
import scipy.sparse as sp

import oracle_lib as O

pytestmark = pytest.mark.gpu

SEED0 = 0x5EED0000


@pytest.fixture(scope="module")
def c5_problem(gpu_pkg):
    return gpu_pkg.problems.spike_slab_logistic_problem(p=10_000, num_rows=2000)


def _c5_oracle(P, x0, th0, T, seed):
    lg = dict(A=P["A"], At=P["At"], y=P["y"], ny=P["ny"], mu=P["mu"], gamma0=P["gamma0"], k=12)
    return O.sspdmp_zigzag(P["G"], P["mu"], P["G"], x0, th0, P["c"], P["kappa"], T, seed=seed, adapt=True, factor=1.5, logistic=lg)


def _same_fact(ev, oe):
    assert len(ev) == len(oe), (len(ev), len(oe))
    for f in ("i", "t", "x", "theta"):
        assert np.array_equal(ev[f], oe[f]), f


def test_c5_sticky_logistic_spike_and_slab_p10000_two_chains_bitwise(gpu_pkg, c5_problem):
    """Config C5 as SURVEY 8d1 writes it: sspdmp(∇ϕmoving, t0, x0, θ0, T, c, Z, κ, SelfMoving(), A, At, μ, y, ny, k; adapt = true) with
    A = example_design_matrix scaled to p = 10 000 columns (every sampled observation reads ~5 500 coefficients), Gaussian slab,
    κ = (γ0/√2π)/(1/w − 1), Z = ZigZag(I, μ, σ) and c = ones(p) as scripts/spikeandslab.jl:96-129.  Two chains through the
    reference-shaped host call, every event, counter, bound and final state equal to the oracle's."""
    pkg, P = gpu_pkg, c5_problem
    p = P["p"]
    assert p == 10_000 and P["A"].shape == (2000, p) and 0.5 * p < P["At"].getnnz(axis=0).mean() < 0.6 * p
    rng = np.random.default_rng(1)
    nch, T = 2, 0.3
    X0 = rng.standard_normal((nch, p))
    TH0 = rng.choice([-1.0, 1.0], (nch, p))
    Z = pkg.ZigZag(P["G"], P["mu"], P["sigma"])
    target = pkg.LogisticTarget(P["A"], P["y"], P["ny"], P["mu"], P["gamma0"], 12)
    tr, (t, x, th), (acc, num), cout = pkg.sspdmp(target, 0.0, X0, TH0, T, P["c"], Z, P["kappa"], seed=900, adapt=True, factor=1.5)
    for k in range(nch):
        r = _c5_oracle(P, X0[k], TH0[k], T, 900 + k)
        assert r["status"] == 0 and len(r["events"]) > 1500
        _same_fact(tr[k].events, r["events"])
        assert (int(acc[k]), int(num[k])) == (r["nacc"], r["num"]) and np.array_equal(cout[k], r["c"])
        assert np.array_equal(x[k], r["x"]) and np.array_equal(th[k], r["theta"]) and np.array_equal(t[k], r["t"])
        frozen = r["theta"] == 0
        assert 0.05 < frozen.mean() < 0.5 and np.all(r["x"][frozen] == 0)  # variable selection at work: exact zeros
        assert r["c"].max() > 1.0  # the unit bounds were adapted


def test_c5_full_width_4096_chains(gpu_pkg, c5_problem):
    """4096 chains of config C5 on one GPU (2.6 GB of chain state, the design shared): every chain healthy, chain 0 and chain 4095
    equal to the oracle, freezes happen everywhere, every trace reconstructs its chain's frozen set."""
    pkg, P = gpu_pkg, c5_problem
    p, nch, T = P["p"], 4096, 0.05
    cap = 1200
    with pkg.Ensemble(nch, p, sampler=pkg._lib.SAMPLER_STICKY_ZIGZAG, adapt=True, factor=1.5, trace_capacity=cap) as ens:
        ens.set_flow(pkg.ZigZag(P["G"], P["mu"], P["sigma"]))
        ens.set_target(pkg.LogisticTarget(P["A"], P["y"], P["ny"], P["mu"], P["gamma0"], 12))
        ens.set_sticky(P["kappa"])
        ens.set_state_synthetic(0.0, P["c"], SEED0)
        ens.run(T)
        cnt = ens.counters()
        assert np.all(cnt["status"] == pkg._lib.CHAIN_OK), np.unique(cnt["status"], return_counts=True)
        assert cnt["ntrace"].min() > 300 and cnt["ntrace"].max() < cap
        fs_first = ens.final_state(0, 1)
        fs_last = ens.final_state(nch - 1, 1)
        for k, fs in ((0, fs_first), (nch - 1, fs_last)):
            x0, th0 = O.synthetic_state(SEED0 + k, p)
            r = _c5_oracle(P, x0, th0, T, SEED0 + k)
            _same_fact(ens.trace(k, counters=cnt), r["events"])
            assert np.array_equal(fs["x"][0], r["x"]) and np.array_equal(fs["theta"][0], r["theta"]) and np.array_equal(fs["c"][0], r["c"])
            assert (int(cnt["nacc"][k]), int(cnt["num"][k])) == (r["nacc"], r["num"])
        # width-independent properties on a spread of chains: the last event of every coordinate in the trace is its final state
        for k in (1, 777, 2048, 4000):
            ev = ens.trace(k, counters=cnt)
            fs = ens.final_state(k, 1)
            last = {}
            for e in ev:
                last[int(e["i"])] = e
            idx = np.array(sorted(last))
            thl = np.array([last[i]["theta"] for i in idx])
            assert np.array_equal(fs["theta"][0][idx], thl)
            frozen = fs["theta"][0] == 0
            assert 0.005 < frozen.mean() < 0.2 and np.all(fs["x"][0][frozen] == 0)
            assert np.all(np.diff(ev["t"]) >= 0) and ev["t"][-1] >= T
        # distinct seeds -> distinct chains
        assert len({int(cnt["ntrace"][k]) * 1_000_003 + int(cnt["ndraw_main"][k]) for k in range(0, nch, 64)}) > 55


def test_c2_full_width_4096_bps_chains(gpu_pkg):
    """Config C2 at its full width: 4096 bouncy-particle chains, d = 1024, Γ = I, λref = 1, c = 1e-3 (scripts/not_fact.jl:23-28).
    Chains 0 and 4095 against the oracle (every event time, the last event's x and θ, the final state, the counters); all chains
    healthy; refreshment and reflection counts add up to the events; per-chain x0/θ0 differ."""
    pkg = gpu_pkg
    d, nch, T = 1024, 4096, 5.0
    rng = np.random.default_rng(2)
    X0 = rng.standard_normal((nch, d))
    TH0 = rng.standard_normal((nch, d))
    cap = 160
    I = sp.identity(d, format="csc")
    with pkg.Ensemble(nch, d, sampler=pkg._lib.SAMPLER_BPS, factor=2.0, trace_capacity=cap) as ens:
        ens.set_flow_bps(pkg.BouncyParticle(I, np.zeros(d), 1.0))
        ens.set_state_bps(0.0, X0, TH0, 1e-3, np.arange(nch, dtype=np.uint64) + SEED0)
        ens.run(T)
        cnt = ens.counters()
        assert np.all(cnt["status"] == pkg._lib.CHAIN_OK)
        assert np.array_equal(cnt["nevents"], cnt["nacc"] + cnt["nrefresh"]) and cnt["ntrace"].max() < cap
        assert cnt["nrefresh"].mean() > 0.7 * T and cnt["nevents"].min() > 5
        fs = ens.bps_final_state()
        assert np.all(fs["t"] >= T) and np.all(np.isfinite(fs["x"])) and np.all(np.isfinite(fs["theta"]))
        for k in (0, nch - 1):
            r = O.pdmp_bps(I, None, X0[k], TH0[k], 1e-3, T, lambda_ref=1.0, seed=SEED0 + k, ev_cap=cap)
            t, x, th = ens.bps_trace(k, counters=cnt)
            assert np.array_equal(t, r["t_ev"]) and np.array_equal(x, r["x_ev"]) and np.array_equal(th, r["theta_ev"])
            assert (int(cnt["nacc"][k]), int(cnt["num"][k]), int(cnt["nrefresh"][k])) == (r["nacc"], r["num"], r["nrefresh"])
            assert np.array_equal(fs["x"][k], r["x"]) and np.array_equal(fs["theta"][k], r["theta"]) and fs["t"][k] == r["t"]
        # every chain's last trace record IS its final state (width-independent property)
        for k in range(0, nch, 257):
            t, x, th = ens.bps_trace(k, counters=cnt)
            assert np.array_equal(x[-1], fs["x"][k]) and np.array_equal(th[-1], fs["theta"][k]) and t[-1] == fs["t"][k]


@pytest.mark.parametrize("tracked", [False, True])
def test_c4_full_width_8192_logistic_chains(gpu_pkg, tracked):
    """Config C4 at one GPU's share of its 65 536-chain ensemble: 8192 chains of the subsampled sparse logistic regression
    (n = 8840, p = 442, k = 10, SelfMoving, adapt, factor 5; scripts/logistic.jl:167).  Chains 0 and 8191 against the oracle; all
    chains healthy; the trace of a spread of chains reconstructs its final velocities; the bounds only grow.  tracked: the same with tracked
    BOUNDS (pdmp_ensemble_set_gradient_tracking on the logistic target) against the oracle's tracked evaluation, bit for bit as well."""
    pkg = gpu_pkg
    L = pkg.problems.logistic_problem(m=20)
    p, nch, T = L["p"], 8192, 4.0
    assert (L["n"], p) == (8840, 442)
    rng = np.random.default_rng(7)
    X0 = np.tile(L["x0"], (nch, 1))
    TH0 = L["sigma"] * rng.choice([-1.0, 1.0], (nch, p))
    cap = 2048
    with pkg.Ensemble(nch, p, adapt=True, factor=5.0, trace_capacity=cap) as ens:
        ens.set_flow(pkg.ZigZag(L["Gdrop"], L["mu"], L["sigma"]))
        ens.set_target(pkg.LogisticTarget(L["A"], L["y"], L["ny"], L["mu"], L["gamma0"], 10))
        ens.set_gradient_tracking(tracked)
        ens.set_state(0.0, X0, TH0, L["c"], np.arange(nch, dtype=np.uint64) + SEED0)
        ens.run(T)
        cnt = ens.counters()
        assert np.all(cnt["status"] == pkg._lib.CHAIN_OK) and cnt["ntrace"].max() < cap and cnt["ntrace"].min() > 100
        lg = dict(A=L["A"], At=L["At"], y=L["y"], ny=L["ny"], mu=L["mu"], gamma0=L["gamma0"], k=10)
        for k in (0, nch - 1):
            r = O.spdmp_zigzag(L["Gdrop"], L["mu"], L["Gdrop"], X0[k], TH0[k], L["c"], T, seed=SEED0 + k, adapt=True, factor=5.0,
                               logistic=lg, tracked=tracked)
            assert r["status"] == 0
            _same_fact(ens.trace(k, counters=cnt), r["events"])
            fs = ens.final_state(k, 1)
            assert np.array_equal(fs["x"][0], r["x"]) and np.array_equal(fs["theta"][0], r["theta"]) and np.array_equal(fs["t"][0], r["t"])
            assert np.array_equal(fs["acc"][0], r["acc"]) and np.array_equal(fs["c"][0], r["c"]) and int(cnt["num"][k]) == r["num"]
        for k in range(5, nch, 1171):
            ev = ens.trace(k, counters=cnt)
            fs = ens.final_state(k, 1)
            flips = np.bincount(ev["i"], minlength=p)
            assert np.array_equal(fs["acc"][0], flips)  # acc[i] counts the accepted reflections of i = its trace events
            assert np.array_equal(np.sign(fs["theta"][0]), np.sign(TH0[k]) * (1 - 2 * (flips % 2)))
            assert np.all(fs["c"][0] >= L["c"]) and int(cnt["nacc"][k]) == len(ev)
