"""Tracked-gradient evaluation of the local ZigZag (pdmp_ensemble_set_gradient_tracking, zz_local_trackp_kernel and its relatives) against
the oracle (-m gpu), twice over:
  * against the oracle's MOVING evaluation (the restatement of src/sfact.jl:73-145): the event INDEX sequence, accept / reject outcomes,
    counters and adapted bounds are exact; event times, positions and the final state agree to 1e-9 relative (measured: ~1e-13; north
    star: 1e-6) -- sums that are advanced are not rounded like sums recomputed;
  * against the oracle's TRACKED evaluation (oracle/pdmp_oracle.c: spdmp_zigzag_tracked, itself held to the moving one by
    tests/test_oracle_tracked.py): BIT FOR BIT -- every event time, position, final clock and state -- so that a commit-rule bug of the
    speculative kernels cannot hide behind the tolerance."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle_lib as O

pytestmark = pytest.mark.gpu
TOL = 1e-9


def close(a, b, tol=TOL):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and bool(np.all(np.abs(a - b) <= tol * np.maximum(1.0, np.maximum(np.abs(a), np.abs(b)))))


def check_chain(ev, fs_t, fs_x, fs_th, acc, num, cout, r):
    oe = r["events"]
    assert len(ev) == len(oe), (len(ev), len(oe))
    assert np.array_equal(ev["i"], oe["i"])                     # index / reflection bookkeeping: exact
    assert np.array_equal(ev["theta"], oe["theta"])             # velocities are ±θ0: exact
    assert close(ev["t"], oe["t"]) and close(ev["x"], oe["x"])
    assert int(num) == r["num"] and np.array_equal(acc, r["acc"])
    assert np.array_equal(fs_th, r["theta"]) and close(fs_t, r["t"]) and close(fs_x, r["x"])
    if cout is not None:
        assert np.array_equal(cout, r["c"])


def check_chain_bitwise(ev, fs_t, fs_x, fs_th, acc, num, cout, r):
    """device-tracked == oracle-tracked, tolerance 0"""
    oe = r["events"]
    assert len(ev) == len(oe), (len(ev), len(oe))
    for f in ("i", "t", "x", "theta"):
        assert np.array_equal(ev[f], oe[f]), f
    assert int(num) == r["num"] and np.array_equal(acc, r["acc"])
    assert np.array_equal(fs_th, r["theta"]) and np.array_equal(fs_t, r["t"]) and np.array_equal(fs_x, r["x"])
    if cout is not None:
        assert np.array_equal(cout, r["c"])


@pytest.mark.parametrize("n,T", [(48, 12.0), (64, 6.0), (47, 5.0)])
def test_tracked_matches_oracle_on_lattices(gpu_pkg, n, T, trackp_form):
    """The north-star workload's relatives (n x n grid-Laplace GMRFs; 47 is odd: border templates everywhere), bound Γ == target Γ."""
    pkg = gpu_pkg
    G = pkg.problems.gmrf_precision(n)
    d = n * n
    rng = np.random.default_rng(n)
    nch = 3
    x0 = rng.standard_normal((nch, d))
    th0 = rng.choice([-1.0, 1.0], (nch, d))
    c = pkg.problems.column_norms(G)
    tr, (t, x, th), (acc, num), _ = pkg.spdmp(pkg.GaussianTarget(G), 0.0, x0, th0, T, c, pkg.ZigZag(G, np.zeros(d)), seed=700 + n, tracked=True)
    dev_t = 0.0
    for k in range(nch):
        r = O.spdmp_zigzag(G, None, G, x0[k], th0[k], c, T, seed=700 + n + k)
        assert r["status"] == 0 and len(r["events"]) > 1000
        check_chain(tr[k].events, t[k], x[k], th[k], acc[k], num[k], None, r)
        check_chain_bitwise(tr[k].events, t[k], x[k], th[k], acc[k], num[k], None,
                            O.spdmp_zigzag(G, None, G, x0[k], th0[k], c, T, seed=700 + n + k, tracked=True))
        dev_t = max(dev_t, float(np.max(np.abs(tr[k].events["t"] - r["events"]["t"]))))
    assert dev_t < 1e-11  # what is actually observed: a few 1e-14


@pytest.mark.parametrize("which", [0, 1])
def test_every_tracked_kernel_commits_the_same_sequence(gpu_pkg, monkeypatch, which):
    """The two tracked kernels -- one proposal per lane over (key, t_old) pairs (0, the default where it applies) and 8-lane groups (1) --
    against the oracle on a lattice both support (include/pdmp_debug.h: pdmp_debug_set_track_groups)."""
    pkg = gpu_pkg
    monkeypatch.setenv("PDMP_TRACK_GROUPS", str(which))
    n, T, nch = 50, 8.0, 2
    G = pkg.problems.gmrf_precision(n)
    d = n * n
    rng = np.random.default_rng(which)
    x0 = rng.standard_normal((nch, d))
    th0 = rng.choice([-1.0, 1.0], (nch, d))
    c = pkg.problems.column_norms(G)
    tr, (t, x, th), (acc, num), _ = pkg.spdmp(pkg.GaussianTarget(G), 0.0, x0, th0, T, c, pkg.ZigZag(G, np.zeros(d)), seed=4100, tracked=True)
    for k in range(nch):
        r = O.spdmp_zigzag(G, None, G, x0[k], th0[k], c, T, seed=4100 + k)
        assert r["status"] == 0 and len(r["events"]) > 1000
        check_chain(tr[k].events, t[k], x[k], th[k], acc[k], num[k], None, r)
        check_chain_bitwise(tr[k].events, t[k], x[k], th[k], acc[k], num[k], None,
                            O.spdmp_zigzag(G, None, G, x0[k], th0[k], c, T, seed=4100 + k, tracked=True))


def test_tracked_with_looser_bound_mean_and_adapt(gpu_pkg):
    """The FULL instantiation: bounding Γ = 0.9 Γ (test/maintest.jl:23: two pairs of tracked sums), a target mean, and adapt with bounds
    that start too small (c is multiplied by `factor` on violations, src/fact_samplers.jl:67-70): adapted bounds equal the oracle's."""
    pkg = gpu_pkg
    n = 50
    G = pkg.problems.gmrf_precision(n)
    d = n * n
    rng = np.random.default_rng(5)
    mu = 0.3 * rng.standard_normal(d)
    nch, T = 2, 6.0
    x0 = rng.standard_normal((nch, d))
    th0 = rng.choice([-1.0, 1.0], (nch, d))
    c = 0.2 * pkg.problems.column_norms(G)
    Z = pkg.ZigZag(sp.csc_matrix(0.9 * G), mu)
    tr, (t, x, th), (acc, num), cout = pkg.spdmp(pkg.GaussianTarget(G, mu), 0.0, x0, th0, T, c, Z, seed=31, adapt=True, factor=1.8, tracked=True)
    for k in range(nch):
        r = O.spdmp_zigzag(0.9 * G, mu, G, x0[k], th0[k], c, T, seed=31 + k, target_mu=mu, adapt=True, factor=1.8)
        assert r["status"] == 0 and r["c"].max() > c.max()
        check_chain(tr[k].events, t[k], x[k], th[k], acc[k], num[k], cout[k], r)
        check_chain_bitwise(tr[k].events, t[k], x[k], th[k], acc[k], num[k], cout[k],
                            O.spdmp_zigzag(0.9 * G, mu, G, x0[k], th0[k], c, T, seed=31 + k, target_mu=mu, adapt=True, factor=1.8, tracked=True))


@pytest.mark.parametrize("n", [48, 64])
def test_tracked_adapt_on_the_one_proposal_per_lane_kernel(gpu_pkg, n, trackp_form):
    """adapt = true on the one-proposal-per-lane tracked kernel (round 6; before, adaptation dropped to the 8-lane-group kernel): bounds that start far too
    small are multiplied by `factor` where a proposal violates them (src/sfact.jl:123-128, src/fact_samplers.jl:67-70) and the run goes on -- events,
    counters, final state AND the adapted bounds bit for bit the tracked oracle's; events and counters those of the moving oracle."""
    pkg = gpu_pkg
    G = pkg.problems.gmrf_precision(n)
    d = n * n
    rng = np.random.default_rng(n + 1)
    nch, T = 3, 4.0
    x0 = rng.standard_normal((nch, d))
    th0 = rng.choice([-1.0, 1.0], (nch, d))
    # (with the bounding Γ equal to the target's the bound exceeds the rate by exactly c_i (1 + Δt / 100): only where c_i is at rounding level can a
    # proposal violate it -- every eleventh coordinate starts there and adapts upwards until rounding no longer reaches it)
    c = pkg.problems.column_norms(G)
    c[::11] = 1e-300
    tr, (t, x, th), (acc, num), cout = pkg.spdmp(pkg.GaussianTarget(G), 0.0, x0, th0, T, c, pkg.ZigZag(G, np.zeros(d)), seed=911, adapt=True, factor=1.8,
                                                 tracked=True)
    with pkg.Ensemble(1, d, adapt=True, factor=1.8) as ens:  # (the line layout does not adapt: that form keeps the pair layout's one-wave kernel here)
        ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
        ens.set_target(pkg.GaussianTarget(G))
        ens.set_gradient_tracking(True)
        ens.set_state_synthetic(0.0, c, 1)
        ens.run(0.05)
        assert ens.kernel_name() == ("zz_local_trackp2_kernel" if trackp_form == "two_waves" else "zz_local_trackp_kernel"), ens.kernel_name()
    for k in range(nch):
        rt = O.spdmp_zigzag(G, None, G, x0[k], th0[k], c, T, seed=911 + k, adapt=True, factor=1.8, tracked=True)
        assert rt["status"] == 0 and np.count_nonzero(rt["c"] > c) > 5 and len(rt["events"]) > 1000  # (bounds did adapt)
        check_chain_bitwise(tr[k].events, t[k], x[k], th[k], acc[k], num[k], cout[k], rt)
        # (which proposals violate a vanishing bound is decided by the last bits of the rate: the MOVING evaluation adapts other coordinates at other
        # times -- its events still agree, its bounds need not: the tracked oracle is the bar here)
        r = O.spdmp_zigzag(G, None, G, x0[k], th0[k], c, T, seed=911 + k, adapt=True, factor=1.8)
        check_chain(tr[k].events, t[k], x[k], th[k], acc[k], num[k], None, r)


@pytest.mark.parametrize("case", ["flow_mean_only", "same_mean", "different_means"])
def test_tracked_with_a_flow_mean(gpu_pkg, case, trackp_form):
    """Z = ZigZag(Γ, μ) with μ ≠ 0 under gradient tracking (round 6).  The flow's Γ[:,i]·μ enters every bound (src/fact_samplers.jl:51); until round 6 the
    one-proposal-per-lane kernel ignored it (found by this test's first case: a flow mean WITHOUT a target mean diverged from the oracle after five
    events).  Now the constant rides in the record line: served with no target mean and with a target whose Γμ equals the flow's (the rate subtracts it
    too); a target mean of its own keeps the 8-lane-group kernel.  Bit for bit the tracked oracle in all three."""
    pkg = gpu_pkg
    n = 48
    G = pkg.problems.gmrf_precision(n)
    d = n * n
    rng = np.random.default_rng(3)
    mu = 0.3 * rng.standard_normal(d)
    tmu = {"flow_mean_only": None, "same_mean": mu, "different_means": 0.5 * mu}[case]
    nch, T = 2, 3.0
    x0 = rng.standard_normal((nch, d))
    th0 = rng.choice([-1.0, 1.0], (nch, d))
    c = 3.0 * pkg.problems.column_norms(G)
    tgt = pkg.GaussianTarget(G) if tmu is None else pkg.GaussianTarget(G, tmu)
    tr, (t, x, th), (acc, num), _ = pkg.spdmp(tgt, 0.0, x0, th0, T, c, pkg.ZigZag(G, mu), seed=77, tracked=True)
    with pkg.Ensemble(1, d) as ens:
        ens.set_flow(pkg.ZigZag(G, mu))
        ens.set_target(tgt)
        ens.set_gradient_tracking(True)
        ens.set_state_synthetic(0.0, c, 1)
        ens.run(0.05)
        want = "zz_local_track_kernel" if case == "different_means" else ("zz_local_trackp2_kernel" if trackp_form == "two_waves" else "zz_local_trackp_kernel")
        assert ens.kernel_name() == want, (case, ens.kernel_name())
    for k in range(nch):
        r = O.spdmp_zigzag(G, mu, G, x0[k], th0[k], c, T, seed=77 + k, target_mu=tmu, tracked=True)
        assert r["status"] == 0 and len(r["events"]) > 1000
        check_chain_bitwise(tr[k].events, t[k], x[k], th[k], acc[k], num[k], None, r)
        check_chain(tr[k].events, t[k], x[k], th[k], acc[k], num[k], None, O.spdmp_zigzag(G, mu, G, x0[k], th0[k], c, T, seed=77 + k, target_mu=tmu))


def test_tracked_with_speeds_that_are_not_one(gpu_pkg, trackp_form):
    """θ0 = ±σ_i with σ_i in [0.5, 1.5] (scripts/logistic.jl:158 starts its chains that way): the reflections keep |θ_i|, the tracked sums move by
    −2 θ_i Γ[:,i] -- nothing in the one-proposal-per-lane kernel may assume unit speeds."""
    pkg = gpu_pkg
    n = 50
    G = pkg.problems.gmrf_precision(n)
    d = n * n
    rng = np.random.default_rng(12)
    sig = 0.5 + rng.random(d)
    nch, T = 2, 3.0
    x0 = rng.standard_normal((nch, d))
    th0 = sig * rng.choice([-1.0, 1.0], (nch, d))
    c = 3.0 * pkg.problems.column_norms(G)
    tr, (t, x, th), (acc, num), _ = pkg.spdmp(pkg.GaussianTarget(G), 0.0, x0, th0, T, c, pkg.ZigZag(G, np.zeros(d), sig), seed=313, tracked=True)
    for k in range(nch):
        r = O.spdmp_zigzag(G, None, G, x0[k], th0[k], c, T, seed=313 + k, sigma=sig, tracked=True)
        assert r["status"] == 0 and len(r["events"]) > 1000
        check_chain_bitwise(tr[k].events, t[k], x[k], th[k], acc[k], num[k], None, r)


def test_tracked_slices_trace_refills_and_violation(gpu_pkg, trackp_form):
    """Slices with PDMP_RUN_STOP_BEFORE, a trace buffer that fills up several times, the reference tail (last event at t′ >= T), path
    integrals (batch means) against the host integral of the trace, and a bound violation without adapt (status, not a crash)."""
    pkg = gpu_pkg
    L = pkg._lib
    n = 48
    G = pkg.problems.gmrf_precision(n)
    d = n * n
    c = pkg.problems.column_norms(G)
    nch = 2
    with pkg.Ensemble(nch, d, trace_capacity=3000) as ens:
        ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
        ens.set_target(pkg.GaussianTarget(G))
        ens.set_gradient_tracking(True)
        ens.set_state_synthetic(0.0, c, 4242)
        evs = [[] for _ in range(nch)]
        for Tk, flag in ((1.7, L.RUN_STOP_BEFORE), (4.0, L.RUN_STOP_BEFORE), (6.5, L.RUN_REFERENCE_TAIL)):
            while True:
                ens.run(Tk, flag)
                cnt = ens.counters()
                for k in range(nch):
                    evs[k].append(ens.trace(k, counters=cnt))
                ens.trace_reset()
                if not L.needs_rerun(cnt["status"]):
                    break
            if Tk == 4.0:
                s1, s2 = ens.batch_means(0.0, 4.0)
        fs = ens.final_state()
        cnt = ens.counters()
    ys = []
    for k in range(nch):
        x0, th0 = O.synthetic_state(4242 + k, d)
        r = O.spdmp_zigzag(G, None, G, x0, th0, c, 6.5, seed=4242 + k)
        ev = np.concatenate(evs[k])
        check_chain(ev, fs["t"][k], fs["x"][k], fs["theta"][k], fs["acc"][k], cnt["num"][k], None, r)
        check_chain_bitwise(ev, fs["t"][k], fs["x"][k], fs["theta"][k], fs["acc"][k], cnt["num"][k], None,
                            O.spdmp_zigzag(G, None, G, x0, th0, c, 6.5, seed=4242 + k, tracked=True))
        assert ev["t"][-1] >= 6.5
        ys.append(pkg.trace.moments(pkg.FactTrace(None, 0.0, x0, th0, r["events"]), 4.0)[0])
    assert np.allclose(s1, np.sum(ys, axis=0), rtol=1e-9, atol=1e-11) and np.allclose(s2, np.sum(np.square(ys), axis=0), rtol=1e-9, atol=1e-11)
    # a bounding Γ below the target's with a tiny c, adapt off: the reference throws (src/sfact.jl:124); here the chains stop with
    # BOUND_VIOLATED exactly where the oracle stops
    rng = np.random.default_rng(1)
    x0 = 3 * rng.standard_normal((2, d))
    th0 = rng.choice([-1.0, 1.0], (2, d))
    Gb = sp.csc_matrix(0.3 * G)
    cs = np.full(d, 1e-6)
    with pytest.raises(RuntimeError, match="Tuning parameter `c` too small"):
        pkg.spdmp(pkg.GaussianTarget(G), 0.0, x0, th0, 5.0, cs, pkg.ZigZag(Gb, np.zeros(d)), seed=3, tracked=True)
    with pkg.Ensemble(2, d, trace_capacity=4096) as ens:
        ens.set_flow(pkg.ZigZag(Gb, np.zeros(d)))
        ens.set_target(pkg.GaussianTarget(G))
        ens.set_gradient_tracking(True)
        ens.set_state(0.0, x0, th0, cs, np.array([3, 4], dtype=np.uint64))
        ens.run(5.0)
        cnt = ens.counters()
        for k in range(2):
            r = O.spdmp_zigzag(Gb, None, G, x0[k], th0[k], cs, 5.0, seed=3 + k)
            assert r["status"] == O.ORC_BOUND_VIOLATED and cnt["status"][k] == L.CHAIN_BOUND_VIOLATED
            assert int(cnt["num"][k]) == r["num"] and int(cnt["nacc"][k]) == r["nacc"]
            ev = ens.trace(k, counters=cnt)
            assert np.array_equal(ev["i"], r["events"]["i"]) and close(ev["t"], r["events"]["t"])


def test_tracking_is_refused_where_it_does_not_apply(gpu_pkg):
    """Opt-in means no silent fall-back: graphs outside the lattice geometry, a refresh clock or the sticky sampler return
    PDMP_ERR_UNSUPPORTED at set_state."""
    pkg = gpu_pkg
    L = pkg._lib
    G = pkg.problems.maintest_precision(8)
    with pkg.Ensemble(1, 8) as ens:
        ens.set_flow(pkg.ZigZag(G, np.zeros(8)))
        ens.set_target(pkg.GaussianTarget(G))
        ens.set_gradient_tracking(True)
        with pytest.raises(L.PdmpError) as ei:
            ens.set_state_synthetic(0.0, np.ones(8), 1)
        assert ei.value.code == L.PDMP_ERR_UNSUPPORTED
        ens.set_gradient_tracking(False)
        ens.set_state_synthetic(0.0, 2 * pkg.problems.column_norms(G), 1)
    Gl = pkg.problems.gmrf_precision(48)
    with pkg.Ensemble(1, 48 * 48) as ens:
        ens.set_flow(pkg.ZigZag(Gl, np.zeros(48 * 48), λref=0.1))
        ens.set_target(pkg.GaussianTarget(Gl))
        ens.set_gradient_tracking(True)
        with pytest.raises(L.PdmpError) as ei:
            ens.set_state_synthetic(0.0, pkg.problems.column_norms(Gl), 1)
        assert ei.value.code == L.PDMP_ERR_UNSUPPORTED


@pytest.fixture(scope="module")
def c3_tracked(gpu_pkg):
    """Config C3 at full size (d = 16384, 4096 chains) on the tracked-gradient kernel -- what bench.py times."""
    pkg = gpu_pkg
    G = pkg.problems.gmrf_precision(128)
    d = G.shape[0]
    c = pkg.problems.column_norms(G)
    nch, T, cap = 4096, 1.0, 20000
    runs = {}
    for tracked in (True, False):
        ens = pkg.Ensemble(nch, d, trace_capacity=cap)
        ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
        ens.set_target(pkg.GaussianTarget(G))
        ens.set_gradient_tracking(tracked)
        ens.set_state_synthetic(0.0, c, 0x5EED0000)
        ens.run(T, pkg._lib.RUN_STOP_BEFORE)
        runs[tracked] = (ens, ens.counters())
    yield pkg, G, c, runs, T
    for ens, _ in runs.values():
        ens.close()


def test_full_size_counters_equal_the_exact_kernel_for_every_chain(c3_tracked):
    """4096 chains x d = 16384 to T = 1: the tracked kernel and the bit-identical moving kernel agree on EVERY chain's proposal count,
    accept count, event count and number of draws consumed (integer bookkeeping of ~3·10⁸ proposals), all chains healthy."""
    pkg, G, c, runs, T = c3_tracked
    ct, ce = runs[True][1], runs[False][1]
    assert np.all(ct["status"] == pkg._lib.CHAIN_OK) and np.all(ce["status"] == pkg._lib.CHAIN_OK)
    for f in ("num", "nacc", "nevents", "ntrace", "ndraw_main"):
        assert np.array_equal(ct[f], ce[f]), f
    assert close(ct["t_last"], ce["t_last"])
    assert ct["num"].sum() > 2.0e8


def test_full_size_traces_and_states_against_exact_kernel_and_oracle(c3_tracked):
    pkg, G, c, runs, T = c3_tracked
    d = G.shape[0]
    (et, ct), (ee, ce) = runs[True], runs[False]
    for k in (0, 1, 1234, 4095):
        a, b = et.trace(k, counters=ct), ee.trace(k, counters=ce)
        assert np.array_equal(a["i"], b["i"]) and np.array_equal(a["theta"], b["theta"]) and close(a["t"], b["t"]) and close(a["x"], b["x"])
        fa, fb = et.final_state(k, 1), ee.final_state(k, 1)
        assert np.array_equal(fa["acc"], fb["acc"]) and np.array_equal(fa["theta"], fb["theta"])
        assert close(fa["t"], fb["t"]) and close(fa["x"], fb["x"])  # the rebuilt lazy clocks are the reference's
    x0, th0 = O.synthetic_state(0x5EED0000, d)
    r = O.spdmp_zigzag(G, None, G, x0, th0, c, T, seed=0x5EED0000, stop_before_T=True)
    ev = et.trace(0, counters=ct)
    assert len(ev) == len(r["events"]) and int(ct["num"][0]) == r["num"]
    assert np.array_equal(ev["i"], r["events"]["i"]) and close(ev["t"], r["events"]["t"]) and close(ev["x"], r["events"]["x"])
    assert float(np.max(np.abs(ev["t"] - r["events"]["t"]))) < 1e-11
    for k in (0, 4095):  # ... and bit for bit against the oracle's tracked evaluation
        x0, th0 = O.synthetic_state(0x5EED0000 + k, d)
        rt = O.spdmp_zigzag(G, None, G, x0, th0, c, T, seed=0x5EED0000 + k, stop_before_T=True, tracked=True)
        fa = et.final_state(k, 1)
        check_chain_bitwise(et.trace(k, counters=ct), fa["t"][0], fa["x"][0], fa["theta"][0], fa["acc"][0], ct["num"][k], None, rt)


@pytest.mark.parametrize("which", [0, 1])
def test_tracked_with_a_start_time(gpu_pkg, monkeypatch, which, trackp_form):
    """t0 != 0: the reference's initial queue carries no t0 (src/sfact.jl:186), so the first proposals lie BEFORE the clocks' start; the
    pair-layout kernel's level-1 base must sit below them, and its t_old is stored (an order of times would not give it)."""
    pkg = gpu_pkg
    monkeypatch.setenv("PDMP_TRACK_GROUPS", str(which))
    n, t0, T, nch = 48, 3.0, 7.0, 2
    G = pkg.problems.gmrf_precision(n)
    d = n * n
    rng = np.random.default_rng(60 + which)
    x0 = rng.standard_normal((nch, d))
    th0 = rng.choice([-1.0, 1.0], (nch, d))
    c = pkg.problems.column_norms(G)
    tr, (t, x, th), (acc, num), _ = pkg.spdmp(pkg.GaussianTarget(G), t0, x0, th0, T, c, pkg.ZigZag(G, np.zeros(d)), seed=6100, tracked=True)
    for k in range(nch):
        r = O.spdmp_zigzag(G, None, G, x0[k], th0[k], c, T, seed=6100 + k, t0=t0)
        assert r["status"] == 0 and len(r["events"]) > 1000
        check_chain(tr[k].events, t[k], x[k], th[k], acc[k], num[k], None, r)
        check_chain_bitwise(tr[k].events, t[k], x[k], th[k], acc[k], num[k], None,
                            O.spdmp_zigzag(G, None, G, x0[k], th0[k], c, T, seed=6100 + k, t0=t0, tracked=True))
