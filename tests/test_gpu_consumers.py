"""Device-side trace consumers (pdmp_ensemble_consume_*, pdmp_consume.hip) against zigzagboomerang.jl_amd/trace.py on the SAME device traces
(-m gpu): collect(discretize(Ξ, dt)) (src/trace.jl:94-125) bit for bit, mean(Ξ) (:182-200) to rounding (the same sums, scaled once instead of
term by term) -- with a trace buffer far smaller than the trace, recycled after every slice, so the consumers see the events in pieces."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def _run(pkg, ens, G, c, T_slices, dt, K, seed, tracked, sticky=None):
    d = G.shape[0]
    nch = ens.nchains
    L = pkg._lib
    ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
    ens.set_target(pkg.GaussianTarget(G))
    if sticky is not None:
        ens.set_sticky(sticky)
    if tracked:
        ens.set_gradient_tracking(True)
    rng = np.random.default_rng(seed)
    x0, th0 = rng.standard_normal((nch, d)), rng.choice([-1.0, 1.0], (nch, d))
    ens.set_state(0.0, x0, th0, c, np.arange(nch, dtype=np.uint64) + seed)
    ens.consume_begin(dt, K)
    evs = [[] for _ in range(nch)]
    refills = 0
    for Tk in T_slices:
        while True:
            ens.run(Tk, L.RUN_STOP_BEFORE)
            cnt = ens.counters()
            ens.consume()
            for k in range(nch):
                evs[k].append(ens.trace(k, counters=cnt))
            ens.trace_reset()
            if not L.needs_rerun(cnt["status"]):
                break
            refills += 1
    traces = [pkg.FactTrace(None, 0.0, x0[k], th0[k], np.concatenate(evs[k])) for k in range(nch)]
    return traces, refills


@pytest.mark.parametrize("tracked", [False, True])
def test_discretize_and_mean_on_the_device(gpu_pkg, tracked):
    pkg = gpu_pkg
    n = 48
    G = pkg.problems.gmrf_precision(n)
    d = n * n
    c = pkg.problems.column_norms(G)
    nch, dt, K = 3, 0.37, 64
    with pkg.Ensemble(nch, d, trace_capacity=1500) as ens:  # ~2000 events per unit time: the buffer fills several times per slice
        traces, refills = _run(pkg, ens, G, c, (2.0, 5.5, 9.0), dt, K, 50, tracked)
        assert refills > 5
        m, T = ens.consume_mean()
        for k in range(nch):
            tr = traces[k]
            assert len(tr.events) > 10000 and T[k] == tr.events["t"][-1]
            grid, X = pkg.trace.discretize(tr, dt)
            gt, got = ens.consume_discretized(k)
            assert got.shape == X.shape and len(grid) > 20
            assert np.array_equal(got, X) and np.array_equal(gt, grid)   # bit for bit (times and positions)
            ref = pkg.trace.mean(tr)
            assert np.allclose(m[k], ref, rtol=1e-12, atol=1e-15)
            # ... and against the oracle's restatement of src/trace.jl (oracle/trace_oracle.c): the device consumers one hop from the reference
            ts, xs = O.trace_discretize(0.0, tr.x0, tr.θ0, tr.events, dt)
            assert len(ts) == len(gt) and np.allclose(gt, ts, rtol=0, atol=1e-10) and np.allclose(got, xs, rtol=0, atol=1e-9)
            assert np.allclose(m[k], O.trace_mean(0.0, tr.x0, tr.events), rtol=1e-12, atol=1e-15)
        # a later slice extends both (the cursors persist; points flushed earlier are re-emitted with the same values)
        ens.run(11.0, pkg._lib.RUN_STOP_BEFORE)
    # dense duplicates inside a chunk: d = 8 coordinates, every chunk of 256 events holds each of them ~32 times
    G4 = pkg.problems.maintest_precision(8)
    with pkg.Ensemble(2, 8, trace_capacity=300) as ens:
        traces, refills = _run(pkg, ens, G4, 2.0 * pkg.problems.column_norms(G4), (40.0, 130.0), 0.9, 160, 7, False)
        assert refills >= 1 and min(len(t.events) for t in traces) > 256
        m, T = ens.consume_mean()
        for k in range(2):
            grid, X = pkg.trace.discretize(traces[k], 0.9)
            assert np.array_equal(ens.consume_discretized(k)[1], X) and len(grid) > 100
            assert np.allclose(m[k], pkg.trace.mean(traces[k]), rtol=1e-12, atol=1e-15)


def test_consumers_on_sticky_traces_and_refusals(gpu_pkg):
    pkg = gpu_pkg
    L = pkg._lib
    n = 16
    G = pkg.problems.gmrf_precision(n, 0.5)
    d = n * n
    c = 1.5 * pkg.problems.column_norms(G)
    with pkg.Ensemble(2, d, sampler=L.SAMPLER_STICKY_ZIGZAG, factor=1.5, trace_capacity=900) as ens:
        traces, refills = _run(pkg, ens, G, c, (6.0, 14.0), 0.5, 40, 3, False, sticky=np.full(d, 0.8))
        m, T = ens.consume_mean()
        for k in range(2):
            assert np.sum(traces[k].events["theta"] == 0.0) > 20  # freezes in the trace
            grid, X = pkg.trace.discretize(traces[k], 0.5)
            assert np.array_equal(ens.consume_discretized(k)[1], X)
            assert np.allclose(m[k], pkg.trace.mean(traces[k]), rtol=1e-12, atol=1e-15)
        p, Tp = ens.consume_inclusion()
        for k in range(2):  # inclusion_prob(Ξ), src/trace.jl:161-178: what a sticky run is for
            ref = pkg.trace.inclusion_prob(traces[k])
            assert np.allclose(p[k], ref, rtol=1e-12, atol=1e-15) and Tp[k] == traces[k].events["t"][-1]
            assert np.allclose(p[k], O.trace_inclusion_prob(0.0, traces[k].x0, traces[k].events), rtol=1e-12, atol=1e-15)  # (oracle/trace_oracle.c)
            assert 0.05 < ref.mean() < 0.95 and ref.min() < 0.9  # coordinates do spend time at 0
    with pkg.Ensemble(1, d, trace_capacity=100) as ens:
        ens.set_flow(pkg.ZigZag(G, np.zeros(d), λref=0.2))
        ens.set_target(pkg.GaussianTarget(G))
        ens.set_state_synthetic(0.0, c, 1)
        with pytest.raises(L.PdmpError) as ei:  # a refresh clock: the trace is not time-ordered (src/sfact.jl:84-85)
            ens.consume_begin(0.5, 10)
        assert ei.value.code == L.PDMP_ERR_UNSUPPORTED
    with pkg.Ensemble(1, d, trace_capacity=100) as ens:
        ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
        ens.set_target(pkg.GaussianTarget(G))
        ens.set_state_synthetic(0.0, c, 1)
        with pytest.raises(L.PdmpError):
            ens.consume()  # consume_begin first
        ens.run(0.5, L.RUN_STOP_BEFORE)
        with pytest.raises(L.PdmpError):
            ens.consume_begin(0.5, 10)  # ... and before the first run


def test_discretized_first_row_without_events_and_a_grid_that_is_too_short(gpu_pkg):
    """collect(discretize(Ξ, dt)) starts with t0 => x0 (src/trace.jl:106-110) also when the chain has produced no event yet; a run that goes past the
    grid given to consume_begin is reported (the unclamped point count), not silently truncated."""
    pkg = gpu_pkg
    G = pkg.problems.gmrf_precision(16)
    d = 256
    rng = np.random.default_rng(0)
    x0 = rng.standard_normal((2, d))
    th0 = rng.choice([-1.0, 1.0], (2, d))
    with pkg.Ensemble(2, d, trace_capacity=4000) as ens:
        ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
        ens.set_target(pkg.GaussianTarget(G))
        ens.set_state(0.0, x0, th0, pkg.problems.column_norms(G), np.array([1, 2], dtype=np.uint64))
        ens.consume_begin(0.25, 8)
        gt, X = ens.consume_discretized(1)          # nothing has run
        assert np.array_equal(gt, [0.0]) and np.array_equal(X, x0[1:2])
        ens.run(1.0, pkg._lib.RUN_STOP_BEFORE)
        ens.consume()
        gt, X = ens.consume_discretized(0)
        assert np.array_equal(gt, [0.0, 0.25, 0.5, 0.75]), gt
        assert np.array_equal(X[0], x0[0])
        ens.trace_reset()
        ens.run(6.0, pkg._lib.RUN_STOP_BEFORE)       # 7 * 0.25 = 1.75 < 6
        ens.consume()
        with pytest.raises(ValueError, match="consume_begin was given 8 points"):
            ens.consume_discretized(0)


@pytest.mark.parametrize("tracked", [False, True])
def test_asynchronous_consumer_equals_the_synchronous_one(gpu_pkg, tracked):
    """pdmp_ensemble_consume_async: slices run back to back without trace_reset -- the consumer of slice k runs on the ensemble's second stream
    while slice k + 1 is sampled into the other trace buffer -- and mean / discretize come out bit for bit as with consume() + trace_reset()
    after every slice (same seeds, same slices); the trace handed back is empty after every call."""
    pkg = gpu_pkg
    L = pkg._lib
    n = 48
    G = pkg.problems.gmrf_precision(n)
    d = n * n
    c = pkg.problems.column_norms(G)
    nch, dt, K = 5, 0.5, 32
    slices = [1.0 + 0.9 * k for k in range(12)]
    res = []
    for mode in ("sync", "async"):
        with pkg.Ensemble(nch, d, trace_capacity=4000) as ens:
            ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
            ens.set_target(pkg.GaussianTarget(G))
            if tracked:
                ens.set_gradient_tracking(True)
            ens.set_state_synthetic(0.0, c, 321)
            ens.consume_begin(dt, K)
            for Tk in slices:
                ens.run(Tk, L.RUN_STOP_BEFORE, sync=False)
                if mode == "sync":
                    ens.sync()
                    assert not np.any(ens.counters()["status"] == L.CHAIN_TRACE_FULL)
                    ens.consume()
                    ens.trace_reset()
                else:
                    ens.consume_async()
            if mode == "async":
                assert ens.last_consume_ms() > 0.0
                cn = ens.counters()
                assert np.all(cn["ntrace"] == 0) and np.all(cn["status"] == L.CHAIN_OK) and cn["nevents"].min() > 10000
            m, T = ens.consume_mean()
            grids = [ens.consume_discretized(k) for k in range(nch)]
            res.append((m, T, grids, ens.counters()["nevents"].copy()))
    (m0, T0, g0, n0), (m1, T1, g1, n1) = res
    assert np.array_equal(n0, n1) and np.array_equal(T0, T1) and np.array_equal(m0, m1)
    for k in range(nch):
        assert np.array_equal(g0[k][0], g1[k][0]) and np.array_equal(g0[k][1], g1[k][1]) and g0[k][1].shape[0] > 20


def test_subtrace_and_cummean_on_the_device(gpu_pkg):
    """subtrace(Ξ, J) and cummean(Ξ) on the device (round 6: src/trace.jl:203-226,275-290) against oracle/trace_oracle.c: the device's subtrace of every
    segment concatenated equals the oracle's subtrace of the whole trace, event for event; the running pairs the consumer leaves beside the events are,
    BIT FOR BIT, the oracle's (t, y / (2 t)) list -- over a trace that reaches the consumer in pieces."""
    pkg = gpu_pkg
    L = pkg._lib
    n = 48
    G = pkg.problems.gmrf_precision(n)
    d = n * n
    c = pkg.problems.column_norms(G)
    nch = 2
    J = np.array([0, 5, 47, 48, 1000, 1001, 2303])
    with pkg.Ensemble(nch, d, trace_capacity=1200) as ens:
        ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
        ens.set_target(pkg.GaussianTarget(G))
        rng = np.random.default_rng(9)
        x0, th0 = rng.standard_normal((nch, d)), rng.choice([-1.0, 1.0], (nch, d))
        ens.set_state(0.0, x0, th0, c, np.arange(nch, dtype=np.uint64) + 90)
        ens.consume_begin(0.5, 8)
        ens.consume_cummean(True)
        evs, subs, cmt, cmy = [[] for _ in range(nch)], [[] for _ in range(nch)], [[] for _ in range(nch)], [[] for _ in range(nch)]
        refills = 0
        for Tk in (1.5, 4.0):
            while True:
                ens.run(Tk, L.RUN_STOP_BEFORE)
                cnt = ens.counters()
                ens.consume()
                for k in range(nch):
                    e = ens.trace(k, counters=cnt)
                    evs[k].append(e)
                    subs[k].append(ens.subtrace(k, J))
                    t, y = ens.consume_cummean_pairs(k, len(e))
                    cmt[k].append(t)
                    cmy[k].append(y)
                ens.trace_reset()
                if not L.needs_rerun(cnt["status"]):
                    break
                refills += 1
        assert refills >= 3
    for k in range(nch):
        ev = np.concatenate(evs[k])
        assert len(ev) > 4000
        ok, oi = O.trace_subtrace(J, ev)
        sub = np.concatenate(subs[k])
        assert len(sub) == len(ok) > 5
        assert np.array_equal(sub["i"], oi)
        for f in ("t", "x", "theta"):
            assert np.array_equal(sub[f], ev[f][ok]), f
        ot, oy = O.trace_cummean(0.0, x0[k], ev)
        assert np.array_equal(np.concatenate(cmt[k]), ot) and np.array_equal(np.concatenate(cmy[k]), oy)
