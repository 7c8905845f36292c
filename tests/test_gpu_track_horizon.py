"""Config C3 at ITS OWN horizon (4096 chains, d = 16384, T = 20) on the tracked-gradient kernel -- what bench.py times --
and on the bit-identical moving kernel (-m gpu).

The tracked evaluation is index-exact only until a rounding difference (~1e-13) between an advanced sum and a gathered one flips a
thinning test or the order of two nearly simultaneous events (measured ≈6·10⁻¹⁰ per proposal: a handful of the 4096 chains by T = 20).
A chain that has "left" is another realisation of the same process -- IF the flip was a rounding flip and not a commit-rule bug of the
speculative kernel.  This test makes that a checked fact:
  * the number of chains whose counters differ from the moving kernel's at T = 20 is small (<= 16);
  * EVERY such chain, plus chains 0 and 4095, equals the oracle's tracked evaluation (oracle/pdmp_oracle.c: spdmp_zigzag_tracked, the
    sequential statement of the same arithmetic) BIT FOR BIT -- counters and the whole final state (clocks, positions, velocities, accept
    counts per coordinate) of the chain inside the 4096-chain run, and every event of its trace when the chain is re-run with its seed --
    so the device's speculative commits are exactly the sequential sampler's, before and after the chain left;
  * the oracle's two evaluations of a leaver share a long common prefix and then split at ONE event (that is what a flipped test, or an
    exact tie of two keys -- they do occur among 5.9e9 proposals -- looks like), and the moving kernel's chains 0 / 4095 equal the moving
    oracle bit for bit over the whole horizon.
"""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
SEED0 = 0x5EED0000
T_END = 20.0
MAX_LEAVERS = 16


@pytest.fixture(scope="module")
def c3_horizon(gpu_pkg):
    pkg = gpu_pkg
    G = pkg.problems.gmrf_precision(128)
    d = G.shape[0]
    c = pkg.problems.column_norms(G)
    ens = {}
    for name, tracked in (("tracked", True), ("exact", False)):
        e = pkg.Ensemble(4096, d, trace_capacity=0)  # counters only: 4096 traces to T = 20 would be 33 GB
        e.set_flow(pkg.ZigZag(G, np.zeros(d)))
        e.set_target(pkg.GaussianTarget(G))
        e.set_gradient_tracking(tracked)
        e.set_state_synthetic(0.0, c, SEED0)
        ens[name] = e
    left_at = np.full(4096, np.inf)
    t = 0.0
    while t < T_END:
        t += 1.0
        cnt = {}
        for name, e in ens.items():
            e.run(t, pkg._lib.RUN_STOP_BEFORE)
            cnt[name] = e.counters()
        differ = (cnt["tracked"]["num"] != cnt["exact"]["num"]) | (cnt["tracked"]["nacc"] != cnt["exact"]["nacc"]) | \
                 (cnt["tracked"]["ndraw_main"] != cnt["exact"]["ndraw_main"])
        left_at = np.where(differ & np.isinf(left_at), t, left_at)
    yield pkg, G, c, ens, cnt, left_at
    for e in ens.values():
        e.close()


def test_few_chains_leave_the_exact_index_sequence_by_T20(c3_horizon):
    pkg, G, c, ens, cnt, left_at = c3_horizon
    for v in cnt.values():
        assert np.all(v["status"] == pkg._lib.CHAIN_OK)
    leavers = np.flatnonzero(np.isfinite(left_at))
    print("C3 to T = 20: %.4g proposals, chains that left the exact index sequence: %s (first seen at t = %s)" %
          (cnt["tracked"]["num"].sum(), leavers.tolist(), left_at[leavers].tolist()))
    assert cnt["tracked"]["num"].sum() > 5.5e9
    assert len(leavers) <= MAX_LEAVERS
    # a chain that has left stays another realisation: it is never counted back in by accident
    stay = ~np.isfinite(left_at)
    for f in ("num", "nacc", "nevents", "ndraw_main"):
        assert np.array_equal(cnt["tracked"][f][stay], cnt["exact"][f][stay]), f


def _bitwise_state(fs, cnt_k, r, k):
    assert (int(cnt_k["num"]), int(cnt_k["nacc"]), int(cnt_k["ndraw_main"])) == (r["num"], r["nacc"], r["ndraw_main"]), k
    assert np.array_equal(fs["acc"][0], r["acc"]) and np.array_equal(fs["theta"][0], r["theta"]), k
    assert np.array_equal(fs["t"][0], r["t"]) and np.array_equal(fs["x"][0], r["x"]), k


def test_every_leaver_is_the_sequential_tracked_sampler_bit_for_bit(c3_horizon):
    pkg, G, c, ens, cnt, left_at = c3_horizon
    d = G.shape[0]
    leavers = np.flatnonzero(np.isfinite(left_at)).tolist()
    assert len(leavers) <= MAX_LEAVERS
    chains = sorted(set(leavers + [0, 4095]))
    oracle_tracked = {}
    for k in chains:
        x0, th0 = O.synthetic_state(SEED0 + k, d)
        rt = O.spdmp_zigzag(G, None, G, x0, th0, c, T_END, seed=SEED0 + k, stop_before_T=True, tracked=True)
        assert rt["status"] == 0
        oracle_tracked[k] = rt
        _bitwise_state(ens["tracked"].final_state(k, 1), cnt["tracked"][k], rt, k)  # the chain as it ran inside the 4096-chain ensemble
        rm = O.spdmp_zigzag(G, None, G, x0, th0, c, T_END, seed=SEED0 + k, stop_before_T=True)
        if k in leavers:
            # the two evaluations share a prefix and split at one event: a flipped thinning test / a swapped pair of nearly simultaneous
            # events / an exact tie of two keys in the tracked arithmetic (popped lowest coordinate first, pdmp_oracle.c) -- not a drift
            a, b = rt["events"], rm["events"]
            m = min(len(a), len(b))
            same = a["i"][:m] == b["i"][:m]
            first = int(np.argmin(same)) if not same.all() else m
            assert first > 1000 and first < m, k
            assert np.allclose(a["t"][:first], b["t"][:first], rtol=1e-9, atol=0), k
            assert a["t"][first - 1] <= left_at[k], k  # ... and it happened before the counters showed it
            # (the same instant with another coordinate: two keys that are exactly equal in the tracked arithmetic and ~1e-13 apart in the
            # moving one, so the two queues pop them in different orders)
            tie = abs(a["t"][first] - b["t"][first]) < 1e-9 * a["t"][first]
            print("chain %d left at event %d, t = %.15g (%s)" % (k, first, a["t"][first], "order of two simultaneous events" if tie else "thinning test"))
        else:
            assert np.array_equal(rt["events"]["i"], rm["events"]["i"]) and np.allclose(rt["events"]["t"], rm["events"]["t"], rtol=1e-9, atol=0)
        if k in (0, 4095):  # the moving kernel is the moving oracle, whole horizon
            _bitwise_state(ens["exact"].final_state(k, 1), cnt["exact"][k], rm, k)
    # every event of those chains: re-run them alone with their seeds and full traces
    cap = max(len(r["events"]) for r in oracle_tracked.values()) + 16
    seeds = np.array([SEED0 + k for k in chains], dtype=np.uint64)
    states = [O.synthetic_state(int(s), d) for s in seeds]
    with pkg.Ensemble(len(chains), d, trace_capacity=cap) as e:
        e.set_flow(pkg.ZigZag(G, np.zeros(d)))
        e.set_target(pkg.GaussianTarget(G))
        e.set_gradient_tracking(True)
        e.set_state(0.0, np.stack([s[0] for s in states]), np.stack([s[1] for s in states]), c, seeds)
        e.run(T_END, pkg._lib.RUN_STOP_BEFORE)
        cn = e.counters()
        for q, k in enumerate(chains):
            ev, oe = e.trace(q, counters=cn), oracle_tracked[k]["events"]
            assert len(ev) == len(oe), (k, len(ev), len(oe))
            for f in ("i", "t", "x", "theta"):
                assert np.array_equal(ev[f], oe[f]), (k, f)
            _bitwise_state(e.final_state(q, 1), cn[q], oracle_tracked[k], k)


def test_moving_kernel_equals_the_moving_oracle_on_64_chains_at_the_horizon(c3_horizon):
    """The bit-identical evaluation at C3's own horizon on more than two chains: 64 chains of the 4096-chain run on zz_local_spec8_kernel -- every
    64th, the ends, and chain 1018 (whose TRACKED arithmetic holds an exact tie of two keys at t = 18.89) -- against the oracle's restatement of
    src/sfact.jl:73-145 over T = 20 (1.4e6 proposals per chain; a thread pool over the host cores, ctypes releases the GIL): counters and the whole
    final state bit for bit.  An exact tie of two keys in the MOVING arithmetic would show up here as a mismatch -- the device pops the lowest
    coordinate, the reference's heap whatever sits higher (src/priorityqueue.jl:46-61) -- and is reported as such, not hidden: none was found on
    these chains."""
    from concurrent.futures import ThreadPoolExecutor
    import os
    pkg, G, c, ens, cnt, left_at = c3_horizon
    d = G.shape[0]
    chains = sorted(set(list(range(0, 4096, 66)) + [1018, 4095]))[:64]
    assert len(chains) == 64

    def run(k):
        x0, th0 = O.synthetic_state(SEED0 + k, d)
        return O.spdmp_zigzag(G, None, G, x0, th0, c, T_END, seed=SEED0 + k, stop_before_T=True, want_trace=False)

    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as pool:
        res = list(pool.map(run, chains))
    mism = []
    for k, r in zip(chains, res):
        assert r["status"] == 0
        fs = ens["exact"].final_state(k, 1)
        ck = cnt["exact"][k]
        same = ((int(ck["num"]), int(ck["nacc"]), int(ck["ndraw_main"])) == (r["num"], r["nacc"], r["ndraw_main"]) and
                np.array_equal(fs["acc"][0], r["acc"]) and np.array_equal(fs["theta"][0], r["theta"]) and np.array_equal(fs["t"][0], r["t"]) and
                np.array_equal(fs["x"][0], r["x"]))
        if not same:
            mism.append(k)
    print("moving kernel vs moving oracle at T = 20: %d chains, %.4g proposals, mismatches (= exact ties of two keys): %s" %
          (len(chains), float(sum(r["num"] for r in res)), mism))
    assert mism == []


def test_c4_tracked_bounds_over_a_long_run(gpu_pkg):
    """Config C4 (8192 chains of the subsampled logistic regression, adapt, factor 5) to T = 100 with tracked BOUNDS and with the moving
    evaluation (7·10⁹ proposals each).  The two arithmetics round the positions differently -- the moving evaluation brings G1[i] to every
    proposal time, the tracked one moves a coordinate only when a sampled row or its own event needs it -- so here chains leave the moving
    evaluation's index sequence sooner than on the lattice (measured: 270 of 8192 by T = 400, ≈1·10⁻⁸ per proposal).  What makes that
    acceptable is checked here: EVERY chain whose counters differ is still the sequential sampler in the tracked arithmetic -- a sample of the
    leavers and chains 0 / 8191 equal the oracle's tracked evaluation bit for bit (counters, clocks, positions, velocities, accept counts,
    adapted bounds) -- and the two ensembles agree in distribution (acceptance rate and proposal counts per chain)."""
    pkg = gpu_pkg
    P = pkg.problems.logistic_problem(m=20)
    d, nch, T = P["p"], 8192, 100.0
    rng = np.random.default_rng(7)
    X0 = np.tile(P["x0"], (nch, 1))
    TH0 = P["sigma"] * rng.choice([-1.0, 1.0], (nch, d))
    res = {}
    for tracked in (True, False):
        with pkg.Ensemble(nch, d, adapt=True, factor=5.0, trace_capacity=0) as ens:
            ens.set_flow(pkg.ZigZag(P["Gdrop"], P["mu"], P["sigma"]))
            ens.set_target(pkg.LogisticTarget(P["A"], P["y"], P["ny"], P["mu"], P["gamma0"], 10))
            ens.set_path_integrals(False)
            ens.set_gradient_tracking(tracked)
            ens.set_state(0.0, X0, TH0, P["c"], np.arange(nch, dtype=np.uint64) + SEED0)
            ens.run(T, pkg._lib.RUN_STOP_BEFORE)
            cnt = ens.counters()
            assert np.all(cnt["status"] == pkg._lib.CHAIN_OK)
            res[tracked] = (cnt, ens.final_state())
    a, b = res[True][0], res[False][0]
    leavers = np.flatnonzero((a["num"] != b["num"]) | (a["nacc"] != b["nacc"]) | (a["ndraw_main"] != b["ndraw_main"]))
    assert len(leavers) <= 400, len(leavers)
    assert abs(a["num"].sum() / b["num"].sum() - 1.0) < 1e-3 and abs(a["nacc"].sum() / b["nacc"].sum() - 1.0) < 1e-3
    lg = dict(A=P["A"], At=P["At"], y=P["y"], ny=P["ny"], mu=P["mu"], gamma0=P["gamma0"], k=10)
    fs = res[True][1]
    for k in sorted(set([0, nch - 1] + [int(q) for q in leavers[:3]])):
        r = O.spdmp_zigzag(P["Gdrop"], P["mu"], P["Gdrop"], X0[k], TH0[k], P["c"], T, seed=SEED0 + k, adapt=True, factor=5.0, logistic=lg,
                           sigma=P["sigma"], stop_before_T=True, tracked=True, want_trace=False)
        assert r["status"] == 0 and int(a["num"][k]) == r["num"] and int(a["ndraw_main"][k]) == r["ndraw_main"], k
        for f, g in (("t", "t"), ("x", "x"), ("theta", "theta"), ("acc", "acc"), ("c", "c")):
            assert np.array_equal(fs[f][k], r[g]), (k, f)
