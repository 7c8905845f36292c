"""Parity of the gfx950 local-ZigZag engine with the CPU oracle, through the C ABI (-m gpu).

Bar: bit-exact index/accept bookkeeping AND bit-exact event times/positions (tolerance 0: the north star
allows 1e-6 relative on floating point; the shared deterministic log/RNG make 0 ulp achievable).
"""
import hashlib

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


def assert_chain_equal(tr_ev, fs, k, num, r, label=""):
    oe = r["events"]
    assert len(tr_ev) == len(oe), (label, len(tr_ev), len(oe))
    for f in ("i", "t", "x", "theta"):
        assert np.array_equal(tr_ev[f], oe[f]), (label, f)
    assert int(num) == r["num"], label
    assert np.array_equal(fs[0][k], r["t"]) and np.array_equal(fs[1][k], r["x"]) and np.array_equal(fs[2][k], r["theta"]), label


def run_case(pkg, G, Gb, x0, th0, c, T, seed, adapt=False, target_mu=None, factor=1.8, **kw):
    d = G.shape[0]
    Z = pkg.ZigZag(Gb, np.zeros(d))
    tr, fs, (acc, num), cout = pkg.spdmp(pkg.GaussianTarget(G, target_mu), 0.0, x0, th0, T, c, Z, seed=seed, adapt=adapt,
                                         factor=factor, **kw)
    for k in range(x0.shape[0]):
        r = O.spdmp_zigzag(Gb, None, G, x0[k], th0[k], c, T, seed=seed + k, adapt=adapt, factor=factor, target_mu=target_mu)
        assert r["status"] == 0
        assert_chain_equal(tr[k].events, fs, k, num[k], r, f"chain {k}")
        assert np.array_equal(acc[k], r["acc"])
        assert np.array_equal(cout[k], r["c"])
    return tr, fs, acc, num


@pytest.mark.parametrize("n,nch,T", [(2, 3, 30.0), (4, 4, 20.0), (8, 8, 20.0), (16, 4, 8.0), (32, 2, 3.0)])
def test_grid_laplace_chains_match_oracle(gpu_pkg, n, nch, T):
    G = gpu_pkg.problems.gmrf_precision(n)
    d = n * n
    rng = np.random.default_rng(n)
    run_case(gpu_pkg, G, G, rng.standard_normal((nch, d)), rng.choice([-1.0, 1.0], (nch, d)),
             gpu_pkg.problems.column_norms(G), T, seed=100 + n)


def test_maintest_d8_bound_differs_from_target(gpu_pkg):
    """Z = ZigZag(0.9Γ, 0) against target Γ (test/maintest.jl:23,44), general θ0 magnitudes, adapt on."""
    G = gpu_pkg.problems.maintest_precision(8)
    rng = np.random.default_rng(5)
    x0 = rng.random((6, 8))
    th0 = rng.choice([-1.0, -0.5, 0.5, 1.0], (6, 8))
    run_case(gpu_pkg, G, 0.9 * G, x0, th0, 0.7 * gpu_pkg.problems.column_norms(G), 200.0, seed=77, adapt=True)


def test_adapt_updates_bounds_like_the_reference(gpu_pkg):
    G = gpu_pkg.problems.gmrf_precision(4)
    rng = np.random.default_rng(0)
    x0 = rng.standard_normal((4, 16)) * 5
    th0 = rng.choice([-1.0, 1.0], (4, 16))
    tr, fs, acc, num = run_case(gpu_pkg, G, 0.3 * G, x0, th0, np.full(16, 1e-6), 40.0, seed=1, adapt=True, factor=1.5)


def test_bound_violation_raises_like_the_reference(gpu_pkg):
    G = gpu_pkg.problems.gmrf_precision(4)
    rng = np.random.default_rng(0)
    x0 = rng.standard_normal((2, 16)) * 5
    th0 = rng.choice([-1.0, 1.0], (2, 16))
    with pytest.raises(RuntimeError, match="Tuning parameter `c` too small"):
        gpu_pkg.spdmp(gpu_pkg.GaussianTarget(G), 0.0, x0, th0, 40.0, np.full(16, 1e-6), gpu_pkg.ZigZag(0.3 * G, np.zeros(16)))


def test_target_with_mean_and_flow_mean(gpu_pkg):
    G = gpu_pkg.problems.gmrf_precision(6)
    d = 36
    rng = np.random.default_rng(9)
    mu = rng.standard_normal(d)
    x0, th0 = rng.standard_normal((3, d)), rng.choice([-1.0, 1.0], (3, d))
    c = gpu_pkg.problems.column_norms(G)
    Z = gpu_pkg.ZigZag(G, mu)
    tr, fs, (acc, num), _ = gpu_pkg.spdmp(gpu_pkg.GaussianTarget(G, mu), 0.0, x0, th0, 15.0, c, Z, seed=31)
    for k in range(3):
        r = O.spdmp_zigzag(G, mu, G, x0[k], th0[k], c, 15.0, seed=31 + k, target_mu=mu)
        assert_chain_equal(tr[k].events, fs, k, num[k], r)


@pytest.mark.parametrize("d", [1, 2, 63, 64, 65, 130])
def test_ragged_sizes_tridiagonal(gpu_pkg, d):
    """Block boundaries of the 64-ary queue: d below / at / above one and two key blocks; k = 1..3 per column."""
    main = 2.0 + 0.1 * np.arange(d)
    G = sp.diags([main] + ([[-1.0] * (d - 1)] * 2 if d > 1 else []), [0] + ([-1, 1] if d > 1 else []), format="csc")
    rng = np.random.default_rng(d)
    run_case(gpu_pkg, G, G, rng.standard_normal((2, d)), rng.choice([-1.0, 1.0], (2, d)),
             gpu_pkg.problems.column_norms(G) + 0.1, 12.0, seed=500 + d)


def test_golden_fixtures_d8_grid8(gpu_pkg, golden):
    for name, scale in (("d8", 0.9), ("grid8", 1.0)):
        G = gpu_pkg.problems.maintest_precision(8) if name == "d8" else gpu_pkg.problems.gmrf_precision(8)
        x0, th0, c = golden[f"{name}_x0"], golden[f"{name}_th0"], golden[f"{name}_c"]
        tr, (t, x, th), (acc, num), _ = gpu_pkg.spdmp(gpu_pkg.GaussianTarget(G), 0.0, x0, th0, 50.0, c,
                                                      gpu_pkg.ZigZag(scale * G, np.zeros(G.shape[0])), seed=1234)
        want = golden[f"{name}_events"]
        for f in ("t", "i", "x", "theta"):
            assert np.array_equal(tr.events[f], want[f]), (name, f)
        assert np.array_equal(acc, golden[f"{name}_acc"]) and num == golden[f"{name}_num"][0]
        assert np.array_equal(np.stack([t, x, th]), golden[f"{name}_final"])


def test_golden_c3_first_10k_events_and_trace_full_status(gpu_pkg, golden):
    """Config C3 (d=16384) with the device-generated synthetic state: first 10^4 events of 2 chains."""
    pkg = gpu_pkg
    G = pkg.problems.gmrf_precision(128)
    d = G.shape[0]
    with pkg.Ensemble(2, d, trace_capacity=10000) as ens:
        ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
        ens.set_target(pkg.GaussianTarget(G))
        ens.set_state_synthetic(0.0, pkg.problems.column_norms(G), 0x5EED0000)
        ens.run(1e9)
        cnt = ens.counters()
        assert np.all(cnt["status"] == pkg._lib.CHAIN_TRACE_FULL) and np.all(cnt["ntrace"] == 10000)
        for k in range(2):
            ev = ens.trace(k, counters=cnt)
            assert np.array_equal(ev["i"].astype(np.uint16), golden[f"c3_chain{k}_idx"])
            h = hashlib.sha256()
            for f in ("t", "x", "theta"):
                h.update(np.ascontiguousarray(ev[f]).tobytes())
            assert h.hexdigest() == str(golden[f"c3_chain{k}_hash"][0])
            assert int(cnt["num"][k]) == int(golden[f"c3_chain{k}_num"][0])


def test_time_slicing_and_trace_refill_are_exact(gpu_pkg):
    """run(T1, STOP_BEFORE) ; run(T2, STOP_BEFORE) ; run(T, REFERENCE_TAIL) with a tiny trace buffer that
    fills and is drained many times == one straight reference run."""
    pkg = gpu_pkg
    G = pkg.problems.gmrf_precision(8)
    d = 64
    rng = np.random.default_rng(4)
    x0, th0 = rng.standard_normal((5, d)), rng.choice([-1.0, 1.0], (5, d))
    c = pkg.problems.column_norms(G)
    T = 30.0
    seeds = np.arange(900, 905, dtype=np.uint64)
    with pkg.Ensemble(5, d, trace_capacity=37) as ens:
        ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
        ens.set_target(pkg.GaussianTarget(G))
        ens.set_state(0.0, x0, th0, c, seeds)
        evs = [[] for _ in range(5)]
        for Tk, flag in ((7.3, pkg._lib.RUN_STOP_BEFORE), (19.9, pkg._lib.RUN_STOP_BEFORE), (T, pkg._lib.RUN_REFERENCE_TAIL)):
            while True:
                ens.run(Tk, flag)
                cnt = ens.counters()
                for k in range(5):
                    evs[k].append(ens.trace(k, counters=cnt))
                ens.trace_reset()
                if not pkg._lib.needs_rerun(cnt["status"]):
                    break
            if flag == pkg._lib.RUN_STOP_BEFORE:
                assert np.all(cnt["t_last"] < Tk)
                assert np.all(ens.final_state()["t"] <= Tk)
        fs = ens.final_state()
        cnt = ens.counters()
    for k in range(5):
        r = O.spdmp_zigzag(G, None, G, x0[k], th0[k], c, T, seed=900 + k)
        ev = np.concatenate(evs[k])
        assert_chain_equal(ev, (fs["t"], fs["x"], fs["theta"]), k, cnt["num"][k], r)
        assert int(cnt["nevents"][k]) == len(ev) and int(cnt["ndraw_main"][k]) == r["ndraw_main"]
        # the oracle's own slice mode agrees with the device's pause point
        r1 = O.spdmp_zigzag(G, None, G, x0[k], th0[k], c, 7.3, seed=900 + k, stop_before_T=True)
        assert np.array_equal(r1["events"]["t"], ev["t"][:len(r1["events"])]) and np.all(r1["events"]["t"] < 7.3)


def test_count_only_mode_matches_traced_mode(gpu_pkg):
    pkg = gpu_pkg
    G = pkg.problems.gmrf_precision(16)
    d = 256
    c = pkg.problems.column_norms(G)
    res = []
    for cap in (0, 100000):
        with pkg.Ensemble(16, d, trace_capacity=cap) as ens:
            ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
            ens.set_target(pkg.GaussianTarget(G))
            ens.set_state_synthetic(0.0, c, 42)
            ens.run(5.0)
            res.append((ens.counters(), ens.final_state()))
    for f in ("num", "nacc", "nevents", "ndraw_main", "t_last"):
        assert np.array_equal(res[0][0][f], res[1][0][f])
    for f in ("t", "x", "theta", "acc"):
        assert np.array_equal(res[0][1][f], res[1][1][f])
    x0, th0 = O.synthetic_state(42 + 3, d)
    r = O.spdmp_zigzag(G, None, G, x0, th0, c, 5.0, seed=45)
    assert np.array_equal(res[0][1]["x"][3], r["x"]) and int(res[0][0]["num"][3]) == r["num"]


def test_refresh_clock_matches_oracle(gpu_pkg):
    """λref > 0: src/sfact.jl:78-114,188-190 (two global-rng coordinate draws, σ_i·(±1) refresh, stale-clock re-bound)."""
    pkg = gpu_pkg
    for n, lam in ((4, 0.7), (8, 0.3)):
        G = pkg.problems.gmrf_precision(n)
        d = n * n
        rng = np.random.default_rng(n)
        sig = 0.5 + rng.random(d)
        x0 = rng.standard_normal((3, d))
        th0 = sig * rng.choice([-1.0, 1.0], (3, d))
        c = 2.0 * pkg.problems.column_norms(G)
        Z = pkg.ZigZag(G, np.zeros(d), sig, λref=lam)
        tr, fs, (acc, num), _ = pkg.spdmp(pkg.GaussianTarget(G), 0.0, x0, th0, 30.0, c, Z, seed=61)
        for k in range(3):
            r = O.spdmp_zigzag(G, None, G, x0[k], th0[k], c, 30.0, seed=61 + k, lambda_ref=lam, sigma=sig)
            assert r["nrefresh"] > 3
            assert_chain_equal(tr[k].events, fs, k, num[k], r, f"refresh n={n} chain {k}")
            assert np.array_equal(acc[k], r["acc"])


@pytest.mark.parametrize("n,T,lam", [(48, 3.0, 2.0), (50, 3.0, 2.0), (128, 1.0, 8.0)])  # (50: d = 2500 is no multiple of 32 -- the clock's slot shares a key block with coordinates)
def test_refresh_clock_on_the_speculative_kernel(gpu_pkg, monkeypatch, n, T, lam):
    """λref > 0 at d = 2304 and d = 16384 (round 6): the 8-event lattice kernel (the default there) and the 4-event kernel take the run -- the
    clock's events are processed by themselves between their speculative iterations (src/sfact.jl:78-114) -- and commit, bit for bit, what the
    one-event kernel and the oracle do: events (reflections and refreshes, in order), counters, both random streams' positions, final state;
    with slices and a trace that refills."""
    pkg = gpu_pkg
    L = pkg._lib
    G = pkg.problems.gmrf_precision(n)
    d = n * n
    rng = np.random.default_rng(n)
    sig = 0.5 + rng.random(d)
    nch = 3
    x0 = rng.standard_normal((nch, d))
    th0 = sig * rng.choice([-1.0, 1.0], (nch, d))
    c = 4.0 * pkg.problems.column_norms(G)  # (|θ_i| = σ_i up to 1.5: bounds with room, no violation over the horizon)
    seeds = [977 + k for k in range(nch)]
    res = {}
    names = {"auto": "zz_local_spec8_kernel", "spec4": "zz_local_spec_kernel", "seq": "zz_local_run_kernel"}
    for kern in ("auto", "spec4", "seq"):
        with pkg.Ensemble(nch, d, trace_capacity=4000) as ens:
            ens.debug_set_kernel(kern)
            ens.set_flow(pkg.ZigZag(G, np.zeros(d), sig, λref=lam))
            ens.set_target(pkg.GaussianTarget(G))
            ens.set_state(0.0, x0, th0, c, seeds)
            evs = [[] for _ in range(nch)]
            for Tk, flag in ((0.4 * T, L.RUN_STOP_BEFORE), (T, L.RUN_REFERENCE_TAIL)):
                while True:
                    ens.run(Tk, flag)
                    cnt = ens.counters()
                    for k in range(nch):
                        evs[k].append(ens.trace(k, counters=cnt))
                    ens.trace_reset()
                    if not L.needs_rerun(cnt["status"]):
                        break
            assert ens.kernel_name() == names[kern], (kern, ens.kernel_name())
            res[kern] = ([np.concatenate(e) for e in evs], cnt, ens.final_state())
    for kern in ("auto", "spec4"):
        for f in ("num", "nacc", "nevents", "nrefresh", "ndraw_main", "ndraw_global", "status"):
            assert np.array_equal(res[kern][1][f], res["seq"][1][f]), (kern, f)
        for k in range(nch):
            for f in ("i", "t", "x", "theta"):
                assert np.array_equal(res[kern][0][k][f], res["seq"][0][k][f]), (kern, k, f)
            for f in ("t", "x", "theta", "acc"):
                assert np.array_equal(res[kern][2][f][k], res["seq"][2][f][k]), (kern, k, f)
    # the oracle: the reference-tail run to T in one piece (slices only cut the launches)
    for k in (0, nch - 1):
        r = O.spdmp_zigzag(G, None, G, x0[k], th0[k], c, T, seed=seeds[k], lambda_ref=lam, sigma=sig)
        assert r["status"] == 0 and r["nrefresh"] >= 2, (r["status"], r["nrefresh"])
        ev = res["auto"][0][k]
        assert len(ev) == len(r["events"]), (len(ev), len(r["events"]))
        for f in ("i", "t", "x", "theta"):
            assert np.array_equal(ev[f], r["events"][f]), (k, f)
        assert int(res["auto"][1]["num"][k]) == r["num"] and int(res["auto"][1]["nrefresh"][k]) == r["nrefresh"]
        assert np.array_equal(res["auto"][2]["x"][k], r["x"]) and np.array_equal(res["auto"][2]["theta"][k], r["theta"])


@pytest.mark.parametrize("which", ["random6", "lattice3d"])
def test_refresh_clock_off_the_lattice(gpu_pkg, which):
    """λref > 0 on graphs that are not the 2-d lattice (round 6): zz_local_spec8g_kernel (the default there) and the 4-event kernel against the one-event
    kernel -- events, counters, both streams, final state bit for bit -- and the oracle."""
    pkg = gpu_pkg
    L = pkg._lib
    G = pkg.problems.random_sparse_precision(2500, 6, seed=5) if which == "random6" else pkg.problems.lattice3d_precision(14)
    d = G.shape[0]
    rng = np.random.default_rng(d)
    sig = 0.5 + rng.random(d)
    nch, T, lam = 2, 2.0, 3.0
    x0 = rng.standard_normal((nch, d))
    th0 = sig * rng.choice([-1.0, 1.0], (nch, d))
    c = 4.0 * pkg.problems.column_norms(G)
    seeds = [1177 + k for k in range(nch)]
    res = {}
    for kern in ("auto", "spec4", "seq"):
        with pkg.Ensemble(nch, d, trace_capacity=3000) as ens:
            ens.debug_set_kernel(kern)
            ens.set_flow(pkg.ZigZag(G, np.zeros(d), sig, λref=lam))
            ens.set_target(pkg.GaussianTarget(G))
            ens.set_state(0.0, x0, th0, c, seeds)
            evs = [[] for _ in range(nch)]
            while True:
                ens.run(T, L.RUN_REFERENCE_TAIL)
                cnt = ens.counters()
                for k in range(nch):
                    evs[k].append(ens.trace(k, counters=cnt))
                ens.trace_reset()
                if not L.needs_rerun(cnt["status"]):
                    break
            res[kern] = ([np.concatenate(e) for e in evs], cnt, ens.final_state(), ens.kernel_name())
    assert res["auto"][3].startswith("zz_local_spec8g_kernel") and res["seq"][3] == "zz_local_run_kernel", (res["auto"][3], res["spec4"][3])
    assert res["spec4"][3].startswith("zz_local_spec_kernel"), res["spec4"][3]
    for kern in ("auto", "spec4"):
        for f in ("num", "nacc", "nevents", "nrefresh", "ndraw_main", "ndraw_global", "status"):
            assert np.array_equal(res[kern][1][f], res["seq"][1][f]), (kern, f)
        for k in range(nch):
            for f in ("i", "t", "x", "theta"):
                assert np.array_equal(res[kern][0][k][f], res["seq"][0][k][f]), (kern, k, f)
            for f in ("t", "x", "theta", "acc"):
                assert np.array_equal(res[kern][2][f][k], res["seq"][2][f][k]), (kern, k, f)
    r = O.spdmp_zigzag(G, None, G, x0[0], th0[0], c, T, seed=seeds[0], lambda_ref=lam, sigma=sig)
    assert r["status"] == 0 and r["nrefresh"] >= 2, (r["status"], r["nrefresh"])
    ev = res["auto"][0][0]
    assert len(ev) == len(r["events"])
    for f in ("i", "t", "x", "theta"):
        assert np.array_equal(ev[f], r["events"][f]), f


def test_pdmp_all_matches_oracle(gpu_pkg):
    """pdmp = spdmp with G = All() (src/sfact.jl:236): all coordinates move at every proposal."""
    pkg = gpu_pkg
    G = pkg.problems.maintest_precision(8)
    rng = np.random.default_rng(8)
    x0 = rng.random((4, 8))
    th0 = rng.choice([-1.0, 1.0], (4, 8))
    c = 0.7 * pkg.problems.column_norms(G)
    Z = pkg.ZigZag(0.9 * G, np.zeros(8))
    tr, fs, (acc, num), cout = pkg.pdmp(pkg.GaussianTarget(G), 0.0, x0, th0, 100.0, c, Z, seed=5, adapt=True)
    for k in range(4):
        r = O.spdmp_zigzag(0.9 * G, None, G, x0[k], th0[k], c, 100.0, seed=5 + k, adapt=True, move_all=True)
        assert_chain_equal(tr[k].events, fs, k, num[k], r, f"pdmp chain {k}")
        assert np.array_equal(cout[k], r["c"])
        assert np.all(fs[0][k] == fs[0][k][0])  # every clock sits at the last proposal time
    G = pkg.problems.gmrf_precision(8)
    x0 = rng.standard_normal((2, 64))
    th0 = rng.choice([-1.0, 1.0], (2, 64))
    c = pkg.problems.column_norms(G)
    tr, fs, (acc, num), _ = pkg.pdmp(pkg.GaussianTarget(G), 0.0, x0, th0, 10.0, c, pkg.ZigZag(G, np.zeros(64), λref=0.5), seed=15)
    for k in range(2):
        r = O.spdmp_zigzag(G, None, G, x0[k], th0[k], c, 10.0, seed=15 + k, move_all=True, lambda_ref=0.5,
                           sigma=np.asarray(G.diagonal()) ** -0.5)
        assert_chain_equal(tr[k].events, fs, k, num[k], r, f"pdmp+refresh chain {k}")
