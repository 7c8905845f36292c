"""Config C3 at full size (d = 16384, 4096 chains) on one MI355X: size-independent properties (-m gpu)."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c3_run(gpu_pkg):
    pkg = gpu_pkg
    G = pkg.problems.gmrf_precision(128)
    d = G.shape[0]
    c = pkg.problems.column_norms(G)
    nch, T, cap = 4096, 0.25, 6000
    ens = pkg.Ensemble(nch, d, trace_capacity=cap)
    ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
    ens.set_target(pkg.GaussianTarget(G))
    ens.set_state_synthetic(0.0, c, 0x5EED0000)
    ens.run(T, pkg._lib.RUN_STOP_BEFORE)
    cnt = ens.counters()
    yield pkg, G, c, ens, cnt, T
    ens.close()


def test_all_chains_healthy_and_counters_consistent(c3_run):
    pkg, G, c, ens, cnt, T = c3_run
    assert np.all(cnt["status"] == pkg._lib.CHAIN_OK)
    assert np.all(cnt["t_last"] < T) and np.all(cnt["t_last"] > 0.9 * T)
    assert np.all(cnt["nacc"] == cnt["nevents"]) and np.all(cnt["ntrace"] == cnt["nevents"])
    alpha = cnt["nacc"].sum() / cnt["num"].sum()
    assert 0.1 < alpha < 0.45
    # proposals per coordinate per unit time on this target are O(4)
    rate = cnt["num"].mean() / G.shape[0] / T
    assert 2.0 < rate < 20.0


def test_sampled_chains_traces_are_sorted_and_reconstruct_the_state(c3_run):
    pkg, G, c, ens, cnt, T = c3_run
    d = G.shape[0]
    kcol = np.diff(G.indptr)
    for k in (0, 1, 2047, 4095):
        ev = ens.trace(k, counters=cnt)
        assert np.all(np.diff(ev["t"]) >= 0) and ev["t"][-1] < T
        assert np.all((ev["i"] >= 0) & (ev["i"] < d)) and np.all(np.abs(ev["theta"]) == 1.0)
        fs = ens.final_state(k, 1)
        assert np.array_equal(np.bincount(ev["i"], minlength=d), fs["acc"][0])
        # RNG bookkeeping: d initial draws + 2 per rejected proposal + (1 + k_i) per accepted one
        want = d + 2 * (int(cnt["num"][k]) - len(ev)) + int((1 + kcol[ev["i"]]).sum())
        assert int(cnt["ndraw_main"][k]) == want
        # replay the trace: position of coordinate i at its last event + drift to its clock == final state
        x0, th0 = O.synthetic_state(0x5EED0000 + k, d)
        last = {}
        for e in ev:
            last[int(e["i"])] = e
        idx = np.array(sorted(last))
        xe = np.array([last[i]["x"] for i in idx])
        te = np.array([last[i]["t"] for i in idx])
        the = np.array([last[i]["theta"] for i in idx])
        assert np.array_equal(fs["theta"][0][idx], the)
        assert np.allclose(fs["x"][0][idx], xe + the * (fs["t"][0][idx] - te), rtol=0, atol=1e-12)
        untouched = np.setdiff1d(np.arange(d), idx)
        assert np.array_equal(fs["theta"][0][untouched], th0[untouched])
        assert np.allclose(fs["x"][0][untouched], x0[untouched] + th0[untouched] * fs["t"][0][untouched], atol=1e-12)


def test_first_chain_matches_oracle_bitwise_at_full_size(c3_run):
    pkg, G, c, ens, cnt, T = c3_run
    d = G.shape[0]
    x0, th0 = O.synthetic_state(0x5EED0000, d)
    r = O.spdmp_zigzag(G, None, G, x0, th0, c, T, seed=0x5EED0000, stop_before_T=True)
    ev = ens.trace(0, counters=cnt)
    assert len(ev) == len(r["events"]) and int(cnt["num"][0]) == r["num"]
    for f in ("t", "i", "x", "theta"):
        assert np.array_equal(ev[f], r["events"][f])


def test_batch_means_equal_host_integrals(gpu_pkg):
    pkg = gpu_pkg
    G = pkg.problems.gmrf_precision(8)
    d = 64
    rng = np.random.default_rng(2)
    nch = 6
    x0, th0 = rng.standard_normal((nch, d)), rng.choice([-1.0, 1.0], (nch, d))
    c = pkg.problems.column_norms(G)
    with pkg.Ensemble(nch, d, trace_capacity=100000) as ens:
        ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
        ens.set_target(pkg.GaussianTarget(G))
        ens.set_state(0.0, x0, th0, c, np.arange(nch, dtype=np.uint64) + 7)
        ens.run(10.0, pkg._lib.RUN_STOP_BEFORE)
        s1a, s2a = ens.batch_means(0.0, 10.0)
        ens.run(25.0, pkg._lib.RUN_STOP_BEFORE)
        s1b, s2b = ens.batch_means(10.0, 25.0)
        cnt = ens.counters()
        trs = [pkg.FactTrace(None, 0.0, x0[k], th0[k], ens.trace(k, counters=cnt)) for k in range(nch)]
    ya = np.array([pkg.trace.moments(tr, 10.0)[0] for tr in trs])
    yab = np.array([pkg.trace.moments(tr, 25.0)[0] for tr in trs])
    yb = (yab * 25.0 - ya * 10.0) / 15.0
    assert np.allclose(s1a, ya.sum(axis=0), atol=1e-10) and np.allclose(s2a, (ya ** 2).sum(axis=0), atol=1e-10)
    assert np.allclose(s1b, yb.sum(axis=0), atol=1e-10) and np.allclose(s2b, (yb ** 2).sum(axis=0), atol=1e-10)
