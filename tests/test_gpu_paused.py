"""PDMP_CHAIN_PAUSED (-m gpu): the event loops keep a launch's draw / proposal counters as 32-bit differences from the header's 64-bit ones, and a
chain that has used 3 * 2^30 of them inside one pdmp_ensemble_run pauses -- resumable like TRACE_FULL, nothing to drain -- instead of wrapping
them (the advisor's finding on zz_logistic_lds_kernel, which now also folds its two stream positions into their 64-bit bases at every refill).
The limit is lowered through include/pdmp_debug.h so that a short run pauses hundreds of times; the result must not notice."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tracked,helper", [(False, 0), (True, 0), (True, 1)])
def test_local_zigzag_pauses_and_resumes_exactly(gpu_pkg, monkeypatch, tracked, helper):
    """spdmp(...) loops over paused launches (samplers.py): events, counters and final state equal the oracle's, bit for bit, on the 8-event kernel
    of the moving evaluation and on both forms of the tracked kernel."""
    pkg = gpu_pkg
    monkeypatch.setenv("PDMP_LAUNCH_COUNT_LIMIT", "3000")
    monkeypatch.setenv("PDMP_HELPER_WAVE", str(helper))
    n, T, nch = 48, 3.0, 2
    G = pkg.problems.gmrf_precision(n)
    d = n * n
    rng = np.random.default_rng(8)
    x0 = rng.standard_normal((nch, d))
    th0 = rng.choice([-1.0, 1.0], (nch, d))
    c = pkg.problems.column_norms(G)
    tr, (t, x, th), (acc, num), _ = pkg.spdmp(pkg.GaussianTarget(G), 0.0, x0, th0, T, c, pkg.ZigZag(G, np.zeros(d)), seed=77, tracked=tracked)
    for k in range(nch):
        r = O.spdmp_zigzag(G, None, G, x0[k], th0[k], c, T, seed=77 + k, tracked=tracked)
        assert r["ndraw_main"] > 20 * 3000  # (the run paused many times)
        ev = tr[k].events
        assert len(ev) == len(r["events"])
        for f in ("i", "t", "x", "theta"):
            assert np.array_equal(ev[f], r["events"][f]), f
        assert int(num[k]) == r["num"] and np.array_equal(acc[k], r["acc"])
        assert np.array_equal(x[k], r["x"]) and np.array_equal(th[k], r["theta"]) and np.array_equal(t[k], r["t"])


def test_pause_is_a_status_not_a_stop(gpu_pkg):
    """Through the C ABI: one call of run() leaves the chains PAUSED short of T, the next calls continue; the counters equal an unlimited run's."""
    pkg = gpu_pkg
    L = pkg._lib
    G = pkg.problems.gmrf_precision(48)
    d = G.shape[0]
    c = pkg.problems.column_norms(G)
    out = []
    for limit in (0, 5000):
        with pkg.Ensemble(3, d) as e:
            pkg._lib.check(e._L.pdmp_debug_set_launch_count_limit(e._h, limit))
            e.set_flow(pkg.ZigZag(G, np.zeros(d)))
            e.set_target(pkg.GaussianTarget(G))
            e.set_gradient_tracking(True)
            e.set_state_synthetic(0.0, c, 4242)
            calls = 0
            while True:
                e.run(2.0, L.RUN_STOP_BEFORE)
                calls += 1
                cn = e.counters()
                if limit and calls == 1:
                    assert np.all(cn["status"] == L.CHAIN_PAUSED) and np.all(cn["t_last"] < 2.0)
                if not np.any(cn["status"] == L.CHAIN_PAUSED):
                    break
            assert (calls > 5) == bool(limit), calls
            out.append((cn.copy(), e.final_state()))
    (c0, f0), (c1, f1) = out
    for f in ("num", "nacc", "nevents", "ndraw_main", "t_last", "status"):
        assert np.array_equal(c0[f], c1[f]), f
    for f in ("t", "x", "theta", "acc"):
        assert np.array_equal(f0[f], f1[f]), f


def test_logistic_kernel_pauses_on_its_proposal_count_and_folds_its_stream_positions(gpu_pkg):
    """zz_logistic_lds_kernel (config C4's kernel, small here): a proposal limit of 400 per launch against an unlimited run -- every counter, both
    stream positions, the adapted bounds and the final state equal."""
    pkg = gpu_pkg
    L = pkg._lib
    P = pkg.problems.logistic_problem(m=6)
    d = P["p"]
    out = []
    for limit in (0, 400):
        with pkg.Ensemble(4, d, adapt=True, factor=5.0, trace_capacity=0) as e:
            pkg._lib.check(e._L.pdmp_debug_set_launch_count_limit(e._h, limit))
            e.set_flow(pkg.ZigZag(P["Gdrop"], P["mu"], P["sigma"]))
            e.set_target(pkg.LogisticTarget(P["A"], P["y"], P["ny"], P["mu"], P["gamma0"], 10))
            rng = np.random.default_rng(3)
            e.set_state(0.0, np.tile(P["x0"], (4, 1)), P["sigma"] * rng.choice([-1.0, 1.0], (4, d)), P["c"], np.arange(4, dtype=np.uint64) + np.uint64(99))
            calls = 0
            while True:
                e.run(6.0, L.RUN_STOP_BEFORE)
                calls += 1
                cn = e.counters()
                if not np.any(cn["status"] == L.CHAIN_PAUSED):
                    break
            assert e.kernel_name() == "zz_logistic_lds_kernel"
            assert (calls > 5) == bool(limit), calls
            out.append((cn.copy(), e.final_state()))
    (c0, f0), (c1, f1) = out
    assert c0["num"].min() > 5 * 400
    for f in ("num", "nacc", "nevents", "ndraw_main", "ndraw_global", "t_last", "status"):
        assert np.array_equal(c0[f], c1[f]), f
    for f in ("t", "x", "theta", "acc", "c"):
        assert np.array_equal(f0[f], f1[f]), f
