"""The 8-events-per-iteration kernel of the north-star workload (d = 16384 lattice, zz_local_spec8_kernel) against the
4-event kernel, the one-event kernel, the one-proposal-per-lane kernel and the oracle: identical event sequences, counters and final states (-m gpu).

The lattice's border coordinates (6 % of them) use other blob templates than the common one, so these runs also cover
the kernel's spare template slots and the early end of a candidate list when an iteration holds more than two of them."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

MODES = (None, "spec4", "seq", "exactp")  # None: the default dispatch = the 8-event kernel for this workload
OTHERS = MODES[1:]  # "exactp": the one-proposal-per-lane kernel (pdmp_exactp.hip), opt-in; serves Γ_bound == Γ_target without adaptation
# (since round 4 zz_local_exactp_kernel is not in the default library: its runs go through the parity build, conftest.py: gpu_pkg_parity --
# a second instance of the package bound to lib/libpdmp_mi355.parity.so; every other mode runs on the default library)
BASIC = MODES[:3]


def _run_sliced(pkg, monkeypatch, mode, nch, cap, slices, seed0, n=128):
    if mode is None:
        monkeypatch.delenv("PDMP_KERNEL", raising=False)
    else:
        monkeypatch.setenv("PDMP_KERNEL", mode)
    G = pkg.problems.gmrf_precision(n)
    d = G.shape[0]
    c = pkg.problems.column_norms(G)
    evs = [[] for _ in range(nch)]
    with pkg.Ensemble(nch, d, trace_capacity=cap) as ens:
        ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
        ens.set_target(pkg.GaussianTarget(G))
        ens.set_state_synthetic(0.0, c, seed0)
        for Tk, flag in slices:
            while True:
                ens.run(Tk, flag)
                cnt = ens.counters()
                if cap > 0:
                    for k in range(nch):
                        evs[k].append(ens.trace(k, counters=cnt))
                    ens.trace_reset()
                if not pkg._lib.needs_rerun(cnt["status"]):
                    break
        cnt = ens.counters()
        fs = ens.final_state()
    ev = [np.concatenate(e) if e else None for e in evs]
    return G, c, ev, cnt, fs


def test_sliced_runs_with_trace_refills_agree_across_kernels_and_with_the_oracle(gpu_pkg, gpu_pkg_parity, monkeypatch):
    pkg = gpu_pkg
    T = 0.3
    slices = ((0.11, pkg._lib.RUN_STOP_BEFORE), (0.2, pkg._lib.RUN_STOP_BEFORE), (T, pkg._lib.RUN_REFERENCE_TAIL))
    nch, seed0 = 6, 0xABC000
    runs = {m: _run_sliced(gpu_pkg_parity if m == "exactp" else pkg, monkeypatch, m, nch, 1500, slices, seed0) for m in MODES}
    G, c, ev8, cnt8, fs8 = runs[None]
    assert np.all(cnt8["status"] == pkg._lib.CHAIN_OK)
    for m in OTHERS:
        _, _, ev, cnt, fs = runs[m]
        for f in ("num", "nacc", "nevents", "ndraw_main", "t_last", "status"):
            assert np.array_equal(cnt8[f], cnt[f]), (m, f)
        for f in ("t", "x", "theta", "acc"):
            assert np.array_equal(fs8[f], fs[f]), (m, f)
        for k in range(nch):
            assert np.array_equal(ev8[k], ev[k]), (m, k)
    d = G.shape[0]
    for k in (0, nch - 1):
        x0, th0 = O.synthetic_state(seed0 + k, d)
        r = O.spdmp_zigzag(G, None, G, x0, th0, c, T, seed=seed0 + k)
        assert len(ev8[k]) == len(r["events"]) and int(cnt8["num"][k]) == r["num"]
        for f in ("t", "i", "x", "theta"):
            assert np.array_equal(ev8[k][f], r["events"][f]), (k, f)
        assert np.array_equal(fs8["x"][k], r["x"]) and np.array_equal(fs8["theta"][k], r["theta"])
        assert np.array_equal(fs8["t"][k], r["t"]) and int(cnt8["ndraw_main"][k]) == r["ndraw_main"]
        # border coordinates did fire (other templates than the common one were in play)
        rows, cols = ev8[k]["i"] // 128, ev8[k]["i"] % 128
        assert np.any((rows < 2) | (rows > 125) | (cols < 2) | (cols > 125))


def test_count_only_mode_equals_traced_mode(gpu_pkg, monkeypatch):
    pkg = gpu_pkg
    slices = ((0.15, pkg._lib.RUN_STOP_BEFORE),)
    _, _, _, cnt0, fs0 = _run_sliced(pkg, monkeypatch, None, 64, 0, slices, 77)
    _, _, ev, cnt1, fs1 = _run_sliced(pkg, monkeypatch, None, 64, 4000, slices, 77)
    for f in ("num", "nacc", "nevents", "ndraw_main", "t_last"):
        assert np.array_equal(cnt0[f], cnt1[f]), f
    for f in ("t", "x", "theta", "acc"):
        assert np.array_equal(fs0[f], fs1[f]), f
    assert all(len(ev[k]) == int(cnt1["nevents"][k]) for k in range(64))


def test_bound_violation_stops_the_chain_at_the_same_event_in_all_kernels(gpu_pkg, monkeypatch):
    """c too small: the first proposal with l >= lbound that is accepted ends the chain (reference: error(...),
    src/sfact.jl:124) -- same status, proposal count and trace prefix on the 8-event, 4-event and one-event kernels."""
    pkg = gpu_pkg
    G = pkg.problems.gmrf_precision(128)
    d = G.shape[0]
    c = 1e-3 * pkg.problems.column_norms(G)  # with the bound's precision 0.5 G < G the affine bound is too low
    res = {}
    for m in BASIC:  # (a bounding matrix other than the target's: not the one-proposal-per-lane kernel's case)
        if m is None:
            monkeypatch.delenv("PDMP_KERNEL", raising=False)
        else:
            monkeypatch.setenv("PDMP_KERNEL", m)
        with pkg.Ensemble(8, d, trace_capacity=20000) as ens:
            ens.set_flow(pkg.ZigZag(0.5 * G, np.zeros(d)))
            ens.set_target(pkg.GaussianTarget(G))
            ens.set_state_synthetic(0.0, c, 4242)
            ens.run(0.5, pkg._lib.RUN_STOP_BEFORE)
            cnt = ens.counters()
            res[m] = (cnt, [ens.trace(k, counters=cnt) for k in range(8)], ens.final_state())
    cnt8, ev8, fs8 = res[None]
    assert np.any(cnt8["status"] == pkg._lib.CHAIN_BOUND_VIOLATED)
    for m in ("spec4", "seq"):
        cnt, ev, fs = res[m]
        for f in ("status", "num", "nacc", "nevents", "ndraw_main", "t_last"):
            assert np.array_equal(cnt8[f], cnt[f]), (m, f)
        for k in range(8):
            assert np.array_equal(ev8[k], ev[k]), (m, k)
        for f in ("t", "x", "theta"):
            assert np.array_equal(fs8[f], fs[f]), (m, f)
    # ... and it is the oracle's state at the violation (proposal counted, G[i] moved, nothing reflected or re-bounded)
    k = int(np.flatnonzero(cnt8["status"] == pkg._lib.CHAIN_BOUND_VIOLATED)[0])
    x0, th0 = O.synthetic_state(4242 + k, d)
    r = O.spdmp_zigzag(0.5 * G, None, G, x0, th0, c, 0.5, seed=4242 + k, stop_before_T=True)
    assert r["status"] != 0 and int(cnt8["num"][k]) == r["num"] and len(ev8[k]) == len(r["events"])
    assert np.array_equal(fs8["x"][k], r["x"]) and np.array_equal(fs8["t"][k], r["t"]) and np.array_equal(fs8["theta"][k], r["theta"])


def test_many_chains_longer_run_three_kernels_agree(gpu_pkg, gpu_pkg_parity, monkeypatch):
    """256 chains to T = 1 (1.4e7 proposals per kernel): counters, final states and trace digests of the 8-event, 4-event and
    one-event kernels are identical -- three independent implementations of the event loop, rare paths included."""
    import hashlib
    pkg = gpu_pkg
    slices = ((0.37, pkg._lib.RUN_STOP_BEFORE), (1.0, pkg._lib.RUN_STOP_BEFORE))
    dig = {}
    for m in MODES:
        _, _, ev, cnt, fs = _run_sliced(gpu_pkg_parity if m == "exactp" else pkg, monkeypatch, m, 256, 16000, slices, 0x51DE)
        h = hashlib.sha256()
        for k in range(256):
            h.update(np.ascontiguousarray(ev[k]).tobytes())
        for f in ("num", "nacc", "nevents", "ndraw_main", "t_last", "status"):
            h.update(np.ascontiguousarray(cnt[f]).tobytes())
        for f in ("t", "x", "theta", "acc"):
            h.update(np.ascontiguousarray(fs[f]).tobytes())
        dig[m] = (h.hexdigest(), int(cnt["num"].sum()), int(np.sum(cnt["status"] != 0)))
    assert dig[None][2] == 0 and dig[None][1] > 1.2e7
    assert dig[None] == dig["spec4"] == dig["seq"] == dig["exactp"], dig


@pytest.mark.parametrize("n,T", [(46, 1.5), (64, 1.0), (100, 0.5), (127, 0.3)])
def test_other_lattice_sizes_use_the_same_kernel(gpu_pkg, gpu_pkg_parity, monkeypatch, n, T):
    """The 8-event kernel serves every n x n lattice with 2048 <= d <= 16384 (its first level covers ceil(d / 32) blocks, the
    rest stay +Inf): d = 2116 (not a multiple of 32), 4096, 10 000, 16 129 against the other kernels and the oracle."""
    pkg = gpu_pkg
    slices = ((0.4 * T, pkg._lib.RUN_STOP_BEFORE), (T, pkg._lib.RUN_REFERENCE_TAIL))
    runs = {m: _run_sliced(gpu_pkg_parity if m == "exactp" else pkg, monkeypatch, m, 5, 3000, slices, 9000 + n, n=n) for m in MODES}
    G, c, ev8, cnt8, fs8 = runs[None]
    d = G.shape[0]
    assert np.all(cnt8["status"] == pkg._lib.CHAIN_OK)
    for m in OTHERS:
        _, _, ev, cnt, fs = runs[m]
        for f in ("num", "nacc", "nevents", "ndraw_main", "t_last"):
            assert np.array_equal(cnt8[f], cnt[f]), (m, f)
        for f in ("t", "x", "theta", "acc"):
            assert np.array_equal(fs8[f], fs[f]), (m, f)
        for k in range(5):
            assert np.array_equal(ev8[k], ev[k]), (m, k)
    x0, th0 = O.synthetic_state(9000 + n, d)
    r = O.spdmp_zigzag(G, None, G, x0, th0, c, T, seed=9000 + n)
    assert len(ev8[0]) == len(r["events"]) and int(cnt8["num"][0]) == r["num"]
    for f in ("t", "i", "x", "theta"):
        assert np.array_equal(ev8[0][f], r["events"][f]), f


@pytest.mark.parametrize("mode", BASIC)
def test_adapt_and_target_mean_on_a_lattice_of_spec8_size(gpu_pkg, monkeypatch, mode):
    """`adapt = true` with bounds that are too small at first (c is multiplied by `factor` on violations, returned per chain) and a
    target with a mean: the 8-event kernel's second instantiation, the 4-event and the one-event kernel against the oracle."""
    pkg = gpu_pkg
    if mode is None:
        monkeypatch.delenv("PDMP_KERNEL", raising=False)
    else:
        monkeypatch.setenv("PDMP_KERNEL", mode)
    n = 48
    G = pkg.problems.gmrf_precision(n)
    d = n * n
    rng = np.random.default_rng(48)
    x0 = rng.standard_normal((3, d))
    th0 = rng.choice([-1.0, 1.0], (3, d))
    mu_t = 0.2 * rng.standard_normal(d)
    c = 0.05 * pkg.problems.column_norms(G)
    T, seed = 1.2, 4800
    Z = pkg.ZigZag(0.8 * G, np.zeros(d))
    tr, fs, (acc, num), cout = pkg.spdmp(pkg.GaussianTarget(G, mu_t), 0.0, x0, th0, T, c, Z, seed=seed, adapt=True, factor=1.7)
    grew = 0
    for k in range(3):
        r = O.spdmp_zigzag(0.8 * G, None, G, x0[k], th0[k], c, T, seed=seed + k, adapt=True, factor=1.7, target_mu=mu_t)
        assert r["status"] == 0
        ev, oe = tr[k].events, r["events"]
        assert len(ev) == len(oe) and int(num[k]) == r["num"]
        for f in ("i", "t", "x", "theta"):
            assert np.array_equal(ev[f], oe[f]), (k, f)
        assert np.array_equal(fs[0][k], r["t"]) and np.array_equal(fs[1][k], r["x"]) and np.array_equal(fs[2][k], r["theta"])
        assert np.array_equal(acc[k], r["acc"]) and np.array_equal(cout[k], r["c"])
        grew += int(np.count_nonzero(cout[k] != c))
    assert grew > 0  # the adaptation did happen


def test_rounding_level_violation_with_the_targets_own_matrix(gpu_pkg, gpu_pkg_parity, monkeypatch):
    """Γ_bound == Γ_target and c at rounding level: the affine bound equals the rate up to the last bits, so an accepted proposal
    with l >= l̄ -- error(...) in the reference, src/sfact.jl:124 -- does occur.  This is the only way into the violation path of
    the one-proposal-per-lane kernel (it serves equal matrices only): same stop, counters, trace and state as the other kernels."""
    pkg = gpu_pkg
    G = pkg.problems.gmrf_precision(64)
    d = G.shape[0]
    c = 1e-17 * pkg.problems.column_norms(G)
    res = {}
    for m in MODES:
        if m is None:
            monkeypatch.delenv("PDMP_KERNEL", raising=False)
        else:
            monkeypatch.setenv("PDMP_KERNEL", m)
        pk = gpu_pkg_parity if m == "exactp" else pkg
        with pk.Ensemble(16, d, trace_capacity=30000) as ens:
            ens.set_flow(pk.ZigZag(G, np.zeros(d)))
            ens.set_target(pk.GaussianTarget(G))
            ens.set_state_synthetic(0.0, c, 0xC0DE)
            ens.run(2.0, pkg._lib.RUN_STOP_BEFORE)
            cnt = ens.counters()
            res[m] = (cnt, [ens.trace(k, counters=cnt) for k in range(16)], ens.final_state())
    cnt8, ev8, fs8 = res[None]
    assert np.any(cnt8["status"] == pkg._lib.CHAIN_BOUND_VIOLATED)
    for m in OTHERS:
        cnt, ev, fs = res[m]
        for f in ("status", "num", "nacc", "nevents", "ndraw_main", "t_last"):
            assert np.array_equal(cnt8[f], cnt[f]), (m, f)
        for k in range(16):
            assert np.array_equal(ev8[k], ev[k]), (m, k)
        for f in ("t", "x", "theta", "acc"):
            assert np.array_equal(fs8[f], fs[f]), (m, f)
