"""Randomised stress of the speculative kernels (-m gpu): random lattices / random sparse precisions, random slice boundaries,
tiny trace buffers (TRACE_FULL in the middle of a multi-event commit), adapt on/off -- every chain must equal the oracle bit
for bit.  Seeds are fixed: the cases are reproducible."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle_lib as O

pytestmark = pytest.mark.gpu


def _random_problem(pkg, rng):
    kind = rng.integers(0, 3)
    if kind == 0:
        n = int(rng.integers(3, 14))
        G = pkg.problems.gmrf_precision(n, eps=float(rng.uniform(0.01, 1.0)))
    elif kind == 1:  # banded: k up to 7, two-hop zone up to 13
        d = int(rng.integers(5, 200))
        w = int(rng.integers(1, 4))
        diags = [np.full(d, 2.0 * w + 1.0 + rng.random())] + [np.full(d - o, -rng.uniform(0.2, 1.0)) for o in range(1, w + 1)]
        G = sp.diags(diags + diags[1:], [0] + list(range(1, w + 1)) + [-o for o in range(1, w + 1)], format="csc")
    else:  # random sparse symmetric, diagonally dominant, small degree
        d = int(rng.integers(8, 120))
        R = sp.random(d, d, density=min(1.5 / d, 0.5), random_state=rng, data_rvs=rng.standard_normal, format="csc")
        A = R + R.T
        G = sp.csc_matrix(A + sp.diags(np.asarray(abs(A).sum(axis=0)).ravel() + 1.0))
    G = sp.csc_matrix(G)
    G.sort_indices()
    return G


@pytest.mark.parametrize("case", range(12))
def test_random_slices_and_tiny_traces_zigzag(gpu_pkg, case):
    pkg = gpu_pkg
    rng = np.random.default_rng(1000 + case)
    G = _random_problem(pkg, rng)
    d = G.shape[0]
    nch = 3
    x0 = rng.standard_normal((nch, d))
    th0 = rng.choice([-1.0, -0.5, 0.5, 1.0], (nch, d))
    adapt = bool(rng.integers(0, 2))
    c = pkg.problems.column_norms(G) * (1.2 if not adapt else float(rng.uniform(0.3, 1.0)))
    T = float(rng.uniform(2.0, 12.0)) * min(1.0, 60.0 / d)
    cap = int(rng.integers(8, 64))
    seed = 5000 + case
    cuts = np.sort(rng.uniform(0, T, size=int(rng.integers(1, 6))))
    refs = [O.spdmp_zigzag(G, None, G, x0[k], th0[k], c, T, seed=seed + k, adapt=adapt) for k in range(nch)]
    if any(r["status"] != 0 for r in refs):
        pytest.skip("bound too small for this draw without adapt")
    events = [[] for _ in range(nch)]
    with pkg.Ensemble(nch, d, adapt=adapt, trace_capacity=cap) as ens:
        ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
        ens.set_target(pkg.GaussianTarget(G))
        ens.set_state(0.0, x0, th0, c, np.arange(nch, dtype=np.uint64) + seed)
        for Tk, flag in [(float(v), pkg._lib.RUN_STOP_BEFORE) for v in cuts] + [(T, pkg._lib.RUN_REFERENCE_TAIL)]:
            while True:
                ens.run(Tk, flag)
                cnt = ens.counters()
                assert not np.any(cnt["status"] == pkg._lib.CHAIN_BOUND_VIOLATED)
                for k in range(nch):
                    if cnt["ntrace"][k]:
                        events[k].append(ens.trace(k, counters=cnt))
                ens.trace_reset()
                if not pkg._lib.needs_rerun(cnt["status"]):
                    break
        fs = ens.final_state()
        cnt = ens.counters()
    for k, r in enumerate(refs):
        ev = np.concatenate(events[k]) if events[k] else np.empty(0, dtype=pkg._lib.EVENT_DTYPE)
        assert len(ev) == len(r["events"]), (case, k, len(ev), len(r["events"]))
        for f in ("i", "t", "x", "theta"):
            assert np.array_equal(ev[f], r["events"][f]), (case, k, f)
        assert int(cnt["num"][k]) == r["num"] and np.array_equal(fs["acc"][k], r["acc"])
        assert np.array_equal(fs["x"][k], r["x"]) and np.array_equal(fs["theta"][k], r["theta"]) and np.array_equal(fs["t"][k], r["t"])
        if adapt:
            assert np.array_equal(fs["c"][k], r["c"])


@pytest.mark.parametrize("kern", ["auto", "seq"])
@pytest.mark.parametrize("case", range(14))
def test_random_means_bounds_and_refresh_zigzag(gpu_pkg, monkeypatch, case, kern):
    """Round 6: the options the first stress test leaves at their defaults -- a flow mean and / or a target mean (equal or not), a bounding Γ of its own
    (0.9 Γ, test/maintest.jl:23), a refresh clock with speeds σ_i, a start time -- on the speculative kernels and on the one-event kernel."""
    pkg = gpu_pkg
    if kern == "seq":
        monkeypatch.setenv("PDMP_KERNEL", "seq")
    else:
        monkeypatch.delenv("PDMP_KERNEL", raising=False)
    rng = np.random.default_rng(3000 + case)
    G = _random_problem(pkg, rng)
    d = G.shape[0]
    nch = 2
    Gb = sp.csc_matrix(0.9 * G) if rng.integers(0, 2) else G
    mu_b = 0.4 * rng.standard_normal(d) if rng.integers(0, 2) else None
    mu_t = (mu_b if (mu_b is not None and rng.integers(0, 2)) else 0.4 * rng.standard_normal(d)) if rng.integers(0, 2) else None
    sig = 0.5 + rng.random(d)
    lam = float(rng.uniform(0.2, 1.5)) if rng.integers(0, 2) else 0.0
    t0 = float(rng.uniform(0.0, 2.0)) if rng.integers(0, 2) else 0.0
    x0 = rng.standard_normal((nch, d))
    th0 = sig * rng.choice([-1.0, 1.0], (nch, d))
    adapt = bool(rng.integers(0, 2))
    c = pkg.problems.column_norms(G) * (float(rng.uniform(2.5, 4.0)) if not adapt else float(rng.uniform(0.3, 1.5)))
    T = t0 + float(rng.uniform(2.0, 12.0)) * min(1.0, 60.0 / d)
    cap = int(rng.integers(16, 128))
    seed = 3500 + 10 * case
    kw = dict(t0=t0, target_mu=mu_t, adapt=adapt, factor=1.8, sigma=sig)
    if lam > 0:
        kw["lambda_ref"] = lam
    refs = [O.spdmp_zigzag(Gb, mu_b, G, x0[k], th0[k], c, T, seed=seed + k, **kw) for k in range(nch)]
    if any(r["status"] != 0 for r in refs):
        pytest.skip("bound too small for this draw without adapt")
    events = [[] for _ in range(nch)]
    with pkg.Ensemble(nch, d, adapt=adapt, factor=1.8, trace_capacity=cap) as ens:
        ens.set_flow(pkg.ZigZag(Gb, np.zeros(d) if mu_b is None else mu_b, sig, λref=lam))
        ens.set_target(pkg.GaussianTarget(G) if mu_t is None else pkg.GaussianTarget(G, mu_t))
        ens.set_state(t0, x0, th0, c, np.arange(nch, dtype=np.uint64) + seed)
        while True:
            ens.run(T, pkg._lib.RUN_REFERENCE_TAIL)
            cnt = ens.counters()
            assert not np.any(cnt["status"] == pkg._lib.CHAIN_BOUND_VIOLATED)
            for k in range(nch):
                if cnt["ntrace"][k]:
                    events[k].append(ens.trace(k, counters=cnt))
            ens.trace_reset()
            if not pkg._lib.needs_rerun(cnt["status"]):
                break
        fs = ens.final_state()
        cnt = ens.counters()
    what = dict(case=case, kern=kern, d=d, own_bound=Gb is not G, mu_b=mu_b is not None, mu_t=mu_t is not None, lam=lam, t0=t0, adapt=adapt)
    for k, r in enumerate(refs):
        ev = np.concatenate(events[k]) if events[k] else np.empty(0, dtype=pkg._lib.EVENT_DTYPE)
        assert len(ev) == len(r["events"]), (what, k, len(ev), len(r["events"]))
        for f in ("i", "t", "x", "theta"):
            assert np.array_equal(ev[f], r["events"][f]), (what, k, f)
        assert int(cnt["num"][k]) == r["num"] and np.array_equal(fs["acc"][k], r["acc"]), what
        assert np.array_equal(fs["x"][k], r["x"]) and np.array_equal(fs["theta"][k], r["theta"]) and np.array_equal(fs["t"][k], r["t"]), what
        if adapt:
            assert np.array_equal(fs["c"][k], r["c"]), what


@pytest.mark.parametrize("case", range(8))
def test_random_slices_and_tiny_traces_sticky(gpu_pkg, case):
    pkg = gpu_pkg
    rng = np.random.default_rng(2000 + case)
    G = _random_problem(pkg, rng)
    d = G.shape[0]
    nch = 2
    x0 = rng.standard_normal((nch, d))
    th0 = rng.choice([-1.0, 1.0], (nch, d))
    c = 1.5 * pkg.problems.column_norms(G)
    kappa = rng.uniform(0.1, 2.0, d)
    reversible, strong = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    T = float(rng.uniform(2.0, 10.0)) * min(1.0, 60.0 / d)
    cap = int(rng.integers(8, 64))
    seed = 7000 + case
    cuts = np.sort(rng.uniform(0, T, size=int(rng.integers(1, 5))))
    refs = [O.sspdmp_zigzag(G, None, G, x0[k], th0[k], c, kappa, T, seed=seed + k, adapt=True, reversible=reversible,
                            strong_upperbounds=strong) for k in range(nch)]
    assert all(r["status"] == 0 for r in refs)
    events = [[] for _ in range(nch)]
    with pkg.Ensemble(nch, d, sampler=pkg._lib.SAMPLER_STICKY_ZIGZAG, adapt=True, factor=1.5, trace_capacity=cap) as ens:
        ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
        ens.set_target(pkg.GaussianTarget(G))
        ens.set_sticky(kappa, reversible, strong)
        ens.set_state(0.0, x0, th0, c, np.arange(nch, dtype=np.uint64) + seed)
        for Tk, flag in [(float(v), pkg._lib.RUN_STOP_BEFORE) for v in cuts] + [(T, pkg._lib.RUN_REFERENCE_TAIL)]:
            while True:
                ens.run(Tk, flag)
                cnt = ens.counters()
                for k in range(nch):
                    if cnt["ntrace"][k]:
                        events[k].append(ens.trace(k, counters=cnt))
                ens.trace_reset()
                if not pkg._lib.needs_rerun(cnt["status"]):
                    break
        fs = ens.final_state()
        cnt = ens.counters()
    for k, r in enumerate(refs):
        ev = np.concatenate(events[k]) if events[k] else np.empty(0, dtype=pkg._lib.EVENT_DTYPE)
        assert len(ev) == len(r["events"]), (case, k, len(ev), len(r["events"]))
        for f in ("i", "t", "x", "theta"):
            assert np.array_equal(ev[f], r["events"][f]), (case, k, f)
        assert (int(cnt["nacc"][k]), int(cnt["num"][k])) == (r["nacc"], r["num"])
        assert np.array_equal(fs["x"][k], r["x"]) and np.array_equal(fs["theta"][k], r["theta"]) and np.array_equal(fs["t"][k], r["t"])
