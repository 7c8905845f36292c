"""Randomised stress of the general kernel (-m gpu): FactBoomerang (mandatory refresh, rotation, ρ, means, speeds) on random graphs, and the ZigZag
on graphs whose neighbourhoods exceed one wavefront, with random slice boundaries and tiny trace buffers -- bit for bit the oracle.  Fixed seeds."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle_lib as O

pytestmark = pytest.mark.gpu


def _graph(pkg, rng, dense):
    if dense:  # two-hop sets beyond 64 members: R R' of a sparse R
        d = int(rng.integers(90, 180))
        R = sp.random(d, d, density=float(rng.uniform(0.04, 0.09)), random_state=rng, data_rvs=rng.standard_normal, format="csc")
        G = sp.csc_matrix(R @ R.T + 2.0 * sp.identity(d))
    else:
        kind = rng.integers(0, 3)
        if kind == 0:
            G = pkg.problems.gmrf_precision(int(rng.integers(3, 12)), eps=float(rng.uniform(0.05, 1.0)))
        elif kind == 1:
            G = pkg.problems.maintest_precision(int(rng.integers(4, 40)))
        else:
            d = int(rng.integers(8, 100))
            R = sp.random(d, d, density=min(1.5 / d, 0.5), random_state=rng, data_rvs=rng.standard_normal, format="csc")
            A = R + R.T
            G = A + sp.diags(np.asarray(abs(A).sum(axis=0)).ravel() + 1.0)
    G = sp.csc_matrix(G)
    G.sort_indices()
    return G


def _run_sliced(pkg, ens, nch, T, cuts, ok_violation=False):
    events = [[] for _ in range(nch)]
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
    return [np.concatenate(e) if e else np.empty(0, dtype=pkg._lib.EVENT_DTYPE) for e in events], ens.final_state(), ens.counters()


def _compare(what, evs, fs, cnt, refs, adapt):
    for k, r in enumerate(refs):
        ev = evs[k]
        assert len(ev) == len(r["events"]), (what, k, len(ev), len(r["events"]))
        for f in ("i", "t", "x", "theta"):
            assert np.array_equal(ev[f], r["events"][f]), (what, k, f)
        assert int(cnt["num"][k]) == r["num"] and np.array_equal(fs["acc"][k], r["acc"]), what
        assert np.array_equal(fs["x"][k], r["x"]) and np.array_equal(fs["theta"][k], r["theta"]) and np.array_equal(fs["t"][k], r["t"]), what
        if adapt:
            assert np.array_equal(fs["c"][k], r["c"]), what


@pytest.mark.parametrize("case", range(14))
def test_random_factboomerang(gpu_pkg, case):
    """src/fact_samplers.jl:37-39,58-65 (λ, ab), src/sfact.jl:29-36 (rotation), :103 (refresh draw) on random graphs, means, speeds, ρ and λref."""
    pkg = gpu_pkg
    rng = np.random.default_rng(8000 + case)
    G = _graph(pkg, rng, dense=False)
    d = G.shape[0]
    nch = 2
    mu = 0.3 * rng.standard_normal(d) if rng.integers(0, 2) else np.zeros(d)
    sig = 0.5 + rng.random(d) if rng.integers(0, 2) else np.ones(d)
    lam = float(rng.uniform(0.1, 1.0))
    rho = float(rng.uniform(0.0, 0.9)) if rng.integers(0, 2) else 0.0
    F = pkg.FactBoomerang(G, mu, lam, σ=sig, ρ=rho)
    x0 = rng.standard_normal((nch, d))
    th0 = sig * rng.standard_normal((nch, d))
    adapt = bool(rng.integers(0, 2))
    c = pkg.problems.column_norms(G) * (float(rng.uniform(2.0, 4.0)) if not adapt else float(rng.uniform(0.2, 1.0)))
    T = float(rng.uniform(4.0, 20.0)) * min(1.0, 40.0 / d)
    cap = int(rng.integers(16, 96))
    seed = 8100 + 10 * case
    cuts = np.sort(rng.uniform(0, T, size=int(rng.integers(0, 4))))
    refs = [O.spdmp_zigzag(G, mu, G, x0[k], th0[k], c, T, seed=seed + k, lambda_ref=lam, rho=rho, sigma=sig, adapt=adapt, factor=1.7,
                           factboomerang=True) for k in range(nch)]
    if any(r["status"] != 0 for r in refs):
        pytest.skip("bound too small for this draw without adapt")
    with pkg.Ensemble(nch, d, adapt=adapt, factor=1.7, trace_capacity=cap) as ens:
        ens.set_flow(F)
        ens.set_target(pkg.GaussianTarget(G))
        ens.set_state(0.0, x0, th0, c, np.arange(nch, dtype=np.uint64) + seed)
        evs, fs, cnt = _run_sliced(pkg, ens, nch, T, cuts)
    what = dict(case=case, d=d, mu=bool(np.any(mu)), rho=rho, lam=lam, adapt=adapt, cap=cap, cuts=len(cuts))
    assert sum(len(r["events"]) for r in refs) > 20, what
    _compare(what, evs, fs, cnt, refs, adapt)


@pytest.mark.parametrize("case", range(8))
def test_random_wide_neighbourhoods_zigzag(gpu_pkg, case):
    """The ZigZag where |S[i]| exceeds a wavefront (the general kernel's chunked re-bound), with its options drawn at random."""
    pkg = gpu_pkg
    rng = np.random.default_rng(8500 + case)
    G = _graph(pkg, rng, dense=True)
    d = G.shape[0]
    nch = 2
    Gb = sp.csc_matrix(0.85 * G) if rng.integers(0, 2) else G
    mu_b = 0.3 * rng.standard_normal(d) if rng.integers(0, 2) else None
    sig = 0.5 + rng.random(d)
    lam = float(rng.uniform(0.2, 1.0)) if rng.integers(0, 2) else 0.0
    x0 = rng.standard_normal((nch, d))
    th0 = sig * rng.choice([-1.0, 1.0], (nch, d))
    adapt = bool(rng.integers(0, 2))
    c = pkg.problems.column_norms(G) * (float(rng.uniform(2.5, 4.0)) if not adapt else float(rng.uniform(0.3, 1.5)))
    T = float(rng.uniform(1.0, 4.0))
    cap = int(rng.integers(16, 128))
    seed = 8600 + 10 * case
    cuts = np.sort(rng.uniform(0, T, size=int(rng.integers(0, 4))))
    kw = dict(adapt=adapt, factor=1.6, sigma=sig)
    if lam > 0:
        kw["lambda_ref"] = lam
    refs = [O.spdmp_zigzag(Gb, mu_b, G, x0[k], th0[k], c, T, seed=seed + k, **kw) for k in range(nch)]
    if any(r["status"] != 0 for r in refs):
        pytest.skip("bound too small for this draw without adapt")
    with pkg.Ensemble(nch, d, adapt=adapt, factor=1.6, trace_capacity=cap) as ens:
        ens.set_flow(pkg.ZigZag(Gb, np.zeros(d) if mu_b is None else mu_b, sig, λref=lam))
        ens.set_target(pkg.GaussianTarget(G))
        ens.set_state(0.0, x0, th0, c, np.arange(nch, dtype=np.uint64) + seed)
        evs, fs, cnt = _run_sliced(pkg, ens, nch, T, cuts)
        kname = ens.kernel_name()
    what = dict(case=case, d=d, own_bound=Gb is not G, mu_b=mu_b is not None, lam=lam, adapt=adapt, cap=cap, kernel=kname)
    _compare(what, evs, fs, cnt, refs, adapt)
