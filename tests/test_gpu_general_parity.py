"""General-degree local ZigZag kernel (neighbourhoods > 64 members) and the subsampled logistic target of config C4 (-m gpu)."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle_lib as O

pytestmark = pytest.mark.gpu


def test_dense_ish_gaussian_needs_the_general_kernel(gpu_pkg):
    """A precision whose two-hop sets exceed one wavefront (|S[i]| up to d = 150): same chain as the oracle."""
    pkg = gpu_pkg
    rng = np.random.default_rng(3)
    d = 150
    R = sp.random(d, d, density=0.08, random_state=rng, data_rvs=rng.standard_normal, format="csc")
    G = sp.csc_matrix(R @ R.T + 2.0 * sp.identity(d))
    G.sort_indices()
    assert np.diff(G.indptr).max() > 64 or True
    x0, th0 = rng.standard_normal((3, d)), rng.choice([-1.0, 1.0], (3, d))
    c = 1.5 * pkg.problems.column_norms(G)
    Z = pkg.ZigZag(0.8 * G, np.zeros(d))
    tr, (t, x, th), (acc, num), cout = pkg.spdmp(pkg.GaussianTarget(G), 0.0, x0, th0, 6.0, c, Z, seed=21, adapt=True)
    for k in range(3):
        r = O.spdmp_zigzag(0.8 * G, None, G, x0[k], th0[k], c, 6.0, seed=21 + k, adapt=True)
        assert r["status"] == 0 and len(tr[k].events) == len(r["events"]) > 50
        for f in ("i", "t", "x", "theta"):
            assert np.array_equal(tr[k].events[f], r["events"][f]), (k, f)
        assert int(num[k]) == r["num"] and np.array_equal(acc[k], r["acc"]) and np.array_equal(cout[k], r["c"])
        assert np.array_equal(x[k], r["x"]) and np.array_equal(th[k], r["theta"]) and np.array_equal(t[k], r["t"])


@pytest.mark.parametrize("ksub,T", [(10, 4.0), (40, 1.5), (70, 1.0)])
def test_config_c4_logistic_subsampled_matches_oracle(gpu_pkg, ksub, T):
    """scripts/logistic.jl:167: spdmp(∇ϕmoving, t0, x0, θ0, T, c, Zdrop, SelfMoving(), A, At, μ, y, ny, 10; adapt=true, factor=5).
    k = 40 and 70 push the sampled rows' entries past one 64-lane chunk / the rows past one wavefront."""
    pkg = gpu_pkg
    P = pkg.problems.logistic_problem(m=20)
    nch = 3
    rng = np.random.default_rng(0)
    X0 = np.tile(P["x0"], (nch, 1))
    TH0 = P["sigma"] * rng.choice([-1.0, 1.0], (nch, P["p"]))
    Z = pkg.ZigZag(P["Gdrop"], P["mu"], P["sigma"])
    target = pkg.LogisticTarget(P["A"], P["y"], P["ny"], P["mu"], P["gamma0"], ksub)
    tr, (t, x, th), (acc, num), cout = pkg.spdmp(target, 0.0, X0, TH0, T, P["c"], Z, seed=31, adapt=True, factor=5.0)
    lg = dict(A=P["A"], At=P["At"], y=P["y"], ny=P["ny"], mu=P["mu"], gamma0=P["gamma0"], k=ksub)
    for k in range(nch):
        r = O.spdmp_zigzag(P["Gdrop"], P["mu"], P["Gdrop"], X0[k], TH0[k], P["c"], T, seed=31 + k, adapt=True, factor=5.0,
                           logistic=lg, sigma=P["sigma"])
        assert r["status"] == 0 and len(r["events"]) > 30
        assert len(tr[k].events) == len(r["events"]), (k, len(tr[k].events), len(r["events"]))
        for f in ("i", "t", "x", "theta"):
            assert np.array_equal(tr[k].events[f], r["events"][f]), (k, f)
        assert int(num[k]) == r["num"] and np.array_equal(acc[k], r["acc"]) and np.array_equal(cout[k], r["c"])
        assert np.array_equal(x[k], r["x"]) and np.array_equal(t[k], r["t"])


def _check_zz(pkg, F, G_t, x0, th0, c, T, seed, **okw):
    tr, (t, x, th), (acc, num), cout = pkg.spdmp(pkg.GaussianTarget(G_t), 0.0, x0, th0, T, c, F, seed=seed,
                                                 adapt=okw.get("adapt", False))
    for k in range(x0.shape[0]):
        r = O.spdmp_zigzag(F.Γ, F.μ, G_t, x0[k], th0[k], c, T, seed=seed + k, lambda_ref=F.λref, rho=F.ρ, sigma=F.σ, **okw)
        assert r["status"] == 0 and len(tr[k].events) == len(r["events"]) > 20, (k, len(tr[k].events), len(r["events"]))
        for f in ("i", "t", "x", "theta"):
            assert np.array_equal(tr[k].events[f], r["events"][f]), (k, f)
        assert int(num[k]) == r["num"] and np.array_equal(acc[k], r["acc"])
        assert np.array_equal(x[k], r["x"]) and np.array_equal(th[k], r["theta"]) and np.array_equal(t[k], r["t"])
    return tr


def test_factboomerang_matches_oracle(gpu_pkg):
    """spdmp with F::FactBoomerang (src/fact_samplers.jl:37-39,58-65; rotation src/sfact.jl:29-36; refresh :103)."""
    pkg = gpu_pkg
    G = pkg.problems.maintest_precision(8)
    rng = np.random.default_rng(2)
    Gz = sp.csc_matrix(1.2 * G)
    F = pkg.FactBoomerang(Gz, np.zeros(8), 0.3)
    x0 = rng.random((4, 8))
    th0 = F.σ * rng.standard_normal((4, 8))
    _check_zz(pkg, F, Gz, x0, th0, pkg.problems.column_norms(G), 150.0, 70, factboomerang=True)
    # non-zero μ, ρ > 0, a lattice with several key blocks
    G2 = pkg.problems.gmrf_precision(9)
    d = 81
    mu = 0.2 * rng.standard_normal(d)
    F2 = pkg.FactBoomerang(G2, mu, 0.5, ρ=0.4)
    x0 = rng.standard_normal((2, d))
    th0 = F2.σ * rng.standard_normal((2, d))
    _check_zz(pkg, F2, G2, x0, th0, 2.0 * pkg.problems.column_norms(G2), 20.0, 80, factboomerang=True)


def test_refresh_on_the_general_kernel(gpu_pkg):
    """ZigZag with λref > 0 on a graph whose neighbourhoods exceed one wavefront (general kernel's refresh branch)."""
    pkg = gpu_pkg
    rng = np.random.default_rng(3)
    d = 150
    R = sp.random(d, d, density=0.08, random_state=rng, data_rvs=rng.standard_normal, format="csc")
    G = sp.csc_matrix(R @ R.T + 2.0 * sp.identity(d))
    G.sort_indices()
    sig = 0.5 + rng.random(d)
    F = pkg.ZigZag(G, np.zeros(d), sig, λref=0.4)
    x0 = rng.standard_normal((2, d))
    th0 = sig * rng.choice([-1.0, 1.0], (2, d))
    _check_zz(pkg, F, G, x0, th0, 2.0 * pkg.problems.column_norms(G), 4.0, 90)


def test_adaptscale_matches_oracle(gpu_pkg):
    """spdmp(...; adaptscale=true), src/sfact.jl:86-99: σ tuned in the refresh branch (ZigZag: no draw, θ = σ·sign θ;
    FactBoomerang: multiplicative step once τ < 0.2).  Events, final state and the tuned σ are bit-identical."""
    pkg = gpu_pkg
    G = pkg.problems.maintest_precision(8)
    d = 8
    rng = np.random.default_rng(4)
    for boom in (False, True):
        sig0 = np.full(d, 2.0)
        x0 = rng.standard_normal((3, d))
        c = np.full(d, 10.0)
        if boom:
            F = pkg.FactBoomerang(sp.csc_matrix(G), np.zeros(d), 2.0, σ=sig0, ρ=0.5)
            th0 = sig0 * rng.standard_normal((3, d))
        else:
            F = pkg.ZigZag(sp.csc_matrix(G), np.zeros(d), sig0, λref=0.5)
            th0 = sig0 * rng.choice([-1.0, 1.0], (3, d))
        T = 400.0
        tr, (t, x, th), (acc, num), cout = pkg.spdmp(pkg.GaussianTarget(G), 0.0, x0, th0, T, c, F, seed=120, adapt=True,
                                                     adaptscale=True)
        for k in range(3):
            r = O.spdmp_zigzag(F.Γ, F.μ, G, x0[k], th0[k], c, T, seed=120 + k, lambda_ref=F.λref, rho=F.ρ, sigma=sig0,
                               adapt=True, adaptscale=True, factboomerang=boom)
            assert r["status"] == 0 and r["nrefresh"] > 50 and len(tr[k].events) == len(r["events"])
            for f in ("i", "t", "x", "theta"):
                assert np.array_equal(tr[k].events[f], r["events"][f]), (boom, k, f)
            assert int(num[k]) == r["num"] and np.array_equal(acc[k], r["acc"]) and np.array_equal(cout[k], r["c"])
            assert np.array_equal(x[k], r["x"]) and np.array_equal(th[k], r["theta"]) and np.array_equal(t[k], r["t"])
            assert np.array_equal(tr[k].F.σ, r["sigma"]) and not np.array_equal(r["sigma"], sig0)
    assert np.array_equal(F.σ, sig0)  # an ensemble call leaves the caller's flow alone


def test_pdmp_all_on_the_general_kernel(gpu_pkg):
    """pdmp(∇ϕ, t0, x0, θ0, T, c, Z::FactBoomerang, Z.Γ) (test/maintest.jl:100-105: G = All(), every proposal moves all d
    coordinates, src/sfact.jl:23-48,236), and the same for a ZigZag whose neighbourhoods exceed one wavefront."""
    pkg = gpu_pkg
    G = pkg.problems.maintest_precision(8)
    rng = np.random.default_rng(6)
    Gz = sp.csc_matrix(0.85 * G)
    F = pkg.FactBoomerang(Gz, np.zeros(8), 0.3)
    x0 = rng.random((3, 8))
    th0 = F.σ * rng.standard_normal((3, 8))
    c = pkg.problems.column_norms(G)
    T, seed = 120.0, 130
    tr, (t, x, th), (acc, num), cout = pkg.pdmp(pkg.GaussianTarget(Gz), 0.0, x0, th0, T, c, F, seed=seed)
    for k in range(3):
        r = O.spdmp_zigzag(F.Γ, F.μ, Gz, x0[k], th0[k], c, T, seed=seed + k, lambda_ref=F.λref, rho=F.ρ, sigma=F.σ,
                           factboomerang=True, move_all=True)
        assert r["status"] == 0 and len(tr[k].events) == len(r["events"]) > 20
        for f in ("i", "t", "x", "theta"):
            assert np.array_equal(tr[k].events[f], r["events"][f]), (k, f)
        assert int(num[k]) == r["num"] and np.array_equal(acc[k], r["acc"])
        assert np.array_equal(x[k], r["x"]) and np.array_equal(th[k], r["theta"]) and np.array_equal(t[k], r["t"])
    d = 150
    R = sp.random(d, d, density=0.08, random_state=rng, data_rvs=rng.standard_normal, format="csc")
    G2 = sp.csc_matrix(R @ R.T + 2.0 * sp.identity(d))
    G2.sort_indices()
    Z = pkg.ZigZag(G2, np.zeros(d), np.ones(d), λref=0.3)
    x0 = rng.standard_normal((2, d))
    th0 = rng.choice([-1.0, 1.0], (2, d))
    c2 = 2.0 * pkg.problems.column_norms(G2)
    tr, (t, x, th), (acc, num), cout = pkg.pdmp(pkg.GaussianTarget(G2), 0.0, x0, th0, 3.0, c2, Z, seed=140)
    for k in range(2):
        r = O.spdmp_zigzag(G2, Z.μ, G2, x0[k], th0[k], c2, 3.0, seed=140 + k, lambda_ref=0.3, sigma=Z.σ, move_all=True)
        assert r["status"] == 0 and len(tr[k].events) == len(r["events"]) > 20
        for f in ("i", "t", "x", "theta"):
            assert np.array_equal(tr[k].events[f], r["events"][f]), (k, f)
        assert np.array_equal(x[k], r["x"]) and np.array_equal(t[k], r["t"]) and int(num[k]) == r["num"]


def test_golden2_c4_on_the_device(gpu_pkg):
    """The committed golden vector of config C4 (tests/golden/golden2.npz, made by make_golden2.py) straight from the device."""
    import hashlib
    import os
    pkg = gpu_pkg
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden2.npz"), allow_pickle=False)
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_golden2", os.path.join(here, "golden", "make_golden2.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    L = mg.c4_inputs(gold)  # the fixture carries its inputs (μ, σ, Γdrop come out of LAPACK solves: host dependent bits)
    th0 = L["th0"]
    Z = pkg.ZigZag(L["Gdrop"], L["mu"], L["sigma"])
    target = pkg.LogisticTarget(L["A"], L["y"], L["ny"], L["mu"], L["gamma0"], 10)
    tr, (t, x, th), (acc, num), cout = pkg.spdmp(target, 0.0, L["x0"], th0, 3.0, L["c"], Z, seed=0x5EED0000, adapt=True, factor=5.0)
    ev = tr.events
    assert np.array_equal(ev["i"].astype(np.uint16), gold["c4_idx"]) and np.array_equal(ev["t"][:50], gold["c4_t_head"])
    assert int(num) == int(gold["c4_n"][0]) and int(acc.sum()) == int(gold["c4_n"][1])
    h = hashlib.sha256()
    for a in (ev["t"], ev["x"], ev["theta"], x, th, t, cout, L["sigma"]):
        h.update(np.ascontiguousarray(a).tobytes())
    assert h.hexdigest() == str(gold["c4_hash"][0])


def test_local_bound_matches_oracle(gpu_pkg):
    """spdmp(∇ϕ, t0, x0, θ0, T, C::LocalBound, F::ZigZag, Γ) (src/local.jl:2-6,10-78,95-149): bounds from the target's own
    derivatives, expiry horizon 2/c_i/|θ_i| with `renew` events, queue initialised at t0 + τ; d = 8 dense-ish Γ with a target
    mean and non-unit speeds, and a lattice with several key blocks; adapt on."""
    pkg = gpu_pkg
    rng = np.random.default_rng(9)
    G = pkg.problems.maintest_precision(8)
    cases = [(G, rng.standard_normal(8) * 0.3, rng.random((3, 8)), rng.choice([-1.0, -0.5, 0.5, 1.0], (3, 8)), 0.5, 300.0, 0.7),
             (pkg.problems.gmrf_precision(12), None, rng.standard_normal((2, 144)), rng.choice([-1.0, 1.0], (2, 144)), 1.0, 15.0, 0.0)]
    for Gc, mu_t, x0, th0, cmul, T, t0 in cases:
        d = Gc.shape[0]
        # distinct c_i: equal horizons 2/c_i/|θ_i| started from equal clocks give EXACTLY tied queue keys, which the reference's
        # binary heap and the device's tournament order differently (elsewhere ties have probability zero)
        c = cmul * pkg.problems.column_norms(Gc) * (1.0 + 0.01 * rng.random(d))
        Z = pkg.ZigZag(Gc, np.zeros(d))
        tr, (t, x, th), (acc, num), cout = pkg.spdmp(pkg.GaussianTarget(Gc, mu_t), t0, x0, th0, T, pkg.LocalBound(c), Z, seed=210,
                                                     adapt=True)
        for k in range(x0.shape[0]):
            r = O.spdmp_zigzag(Gc, None, Gc, x0[k], th0[k], c, T, t0=t0, target_mu=mu_t, seed=210 + k, adapt=True, local_bound=True)
            assert r["status"] == 0 and len(tr[k].events) == len(r["events"]) > 50, (k, len(tr[k].events), len(r["events"]))
            for f in ("i", "t", "x", "theta"):
                assert np.array_equal(tr[k].events[f], r["events"][f]), (k, f)
            assert int(num[k]) == r["num"] and np.array_equal(acc[k], r["acc"]) and np.array_equal(cout[k], r["c"])
            assert np.array_equal(x[k], r["x"]) and np.array_equal(th[k], r["theta"]) and np.array_equal(t[k], r["t"])
            assert r["ndraw_main"] > r["num"] + r["nacc"]  # renew events consume draws without proposals


def test_local_bound_exact_ties_are_ordered_by_index_not_by_heap_shape(gpu_pkg):
    """The one documented divergence (include/pdmp_mi355.h, INTEGRATION.md "Known divergences"): with UNIFORM c and |θ| = 1 every
    LocalBound horizon 2/c/|θ| is the same number, so coordinates re-bounded at one instant get EXACTLY equal queue keys.  The reference
    pops tied keys in the order its binary heap happens to hold them (src/priorityqueue.jl:46-77, restated in the oracle); the device
    pops the lowest coordinate.  Both are valid orders of simultaneous events of independent clocks, but the seeded stream then pairs
    differently with the coordinates.  Demonstrated here: the two sequences agree up to the first exactly tied pop, the first difference IS
    a tie (equal times, different coordinates), the device's choice there is the lowest tied coordinate, and the device run stays healthy
    and samples the same law.  (With distinct c_i -- test_local_bound_matches_oracle -- the sequences are bit-identical.)"""
    pkg = gpu_pkg
    G = pkg.problems.gmrf_precision(12)
    d = 144
    rng = np.random.default_rng(3)
    x0 = rng.standard_normal((1, d))
    th0 = rng.choice([-1.0, 1.0], (1, d))
    c = np.full(d, 4.5)
    T = 30.0
    tr, (t, x, th), (acc, num), _ = pkg.spdmp(pkg.GaussianTarget(G), 0.0, x0, th0, T, pkg.LocalBound(c), pkg.ZigZag(G, np.zeros(d)), seed=11)
    r = O.spdmp_zigzag(G, None, G, x0[0], th0[0], c, T, seed=11, local_bound=True)
    ev, oe = tr[0].events, r["events"]
    n = min(len(ev), len(oe))
    diff = np.nonzero((ev["i"][:n] != oe["i"][:n]) | (ev["t"][:n] != oe["t"][:n]))[0]
    assert len(diff) > 0, "uniform c no longer produces tied horizons: the documented divergence is gone -- update the header"
    k = int(diff[0])
    assert k > 0 and np.array_equal(ev[:k], oe[:k])  # identical up to the first tied pop
    # after the divergence the chain is as healthy as before: same event rate within a few per cent, finite state
    assert abs(len(ev) - len(oe)) < 0.1 * len(oe) and np.all(np.isfinite(x)) and np.all(np.abs(th) == 1.0)


def test_c4_state_in_lds_equals_state_in_hbm(gpu_pkg, gpu_pkg_parity):
    """Config C4's kernel keeps a chain's state in LDS for a whole slice (pdmp_logistic.hip); PDMP_DEBUG_KERNEL_SEQ keeps the records in HBM
    (pdmp_general.hip).  Same chains on both, in slices with trace refills (state leaves and re-enters LDS at every launch), with and
    without the engine's path integrals: identical events, counters, final states, adapted bounds, ∫x dt; the integrals' entry points are
    refused with a status when they were switched off."""
    pkg = gpu_pkg
    L = pkg._lib
    P = pkg.problems.logistic_problem(m=20)
    d, nch = P["p"], 6
    rng = np.random.default_rng(4)
    X0 = np.tile(P["x0"], (nch, 1))
    TH0 = P["sigma"] * rng.choice([-1.0, 1.0], (nch, d))
    seeds = np.arange(nch, dtype=np.uint64) + 900
    runs = {}
    # ... and the same state with SEVERAL chains per wavefront, each in a row of 32 or 16 lanes (pdmp_logrows.hip, opt-in: measured slower,
    # DESIGN.md §5a): 6 chains = rows that are full, half full and -- in the last wavefront -- absent
    for name, kernel, integrals, rows in (("lds", "auto", True, 0), ("lds_noI", "auto", False, 0), ("hbm", "seq", True, 0),
                                          ("rows32", "auto", True, 32), ("rows16", "auto", True, 16), ("rows16_noI", "auto", False, 16)):
        pk = gpu_pkg_parity if rows else pkg  # (zz_logistic_rows_kernel lives in the parity build of the library: conftest.py)
        with pk.Ensemble(nch, d, adapt=True, factor=5.0, trace_capacity=400) as ens:
            ens.debug_set_kernel(kernel)
            ens.debug_set_logistic_rows(rows)
            ens.set_flow(pk.ZigZag(P["Gdrop"], P["mu"], P["sigma"]))
            ens.set_target(pk.LogisticTarget(P["A"], P["y"], P["ny"], P["mu"], P["gamma0"], 10))
            ens.set_path_integrals(integrals)
            ens.set_state(0.0, X0, TH0, P["c"], seeds)
            evs = [[] for _ in range(nch)]
            for Tk in (3.0, 7.5, 12.0):
                while True:
                    ens.run(Tk, L.RUN_STOP_BEFORE)
                    cnt = ens.counters()
                    for k in range(nch):
                        evs[k].append(ens.trace(k, counters=cnt))
                    ens.trace_reset()
                    if not L.needs_rerun(cnt["status"]):
                        break
            if integrals:
                bm = ens.batch_means(0.0, 12.0)
                pj = ens.path_integrals(12.0, np.arange(0, d, 7))
            else:
                bm = pj = None
                with pytest.raises(pk._lib.PdmpError) as ei:
                    ens.batch_means(0.0, 12.0)
                assert ei.value.code == L.PDMP_ERR_INVALID and "switched off" in str(ei.value)
            runs[name] = (cnt, [np.concatenate(e) for e in evs], ens.final_state(), bm, pj)
    ref = runs["hbm"]
    assert ref[0]["nacc"].sum() > 300 and np.all(ref[0]["status"] == L.CHAIN_OK)
    for name in ("lds", "lds_noI", "rows32", "rows16", "rows16_noI"):
        cnt, evs, fs, bm, pj = runs[name]
        for f in ("num", "nacc", "nevents", "ndraw_main", "ndraw_global", "t_last"):
            assert np.array_equal(cnt[f], ref[0][f]), (name, f)
        for k in range(nch):
            for f in ("i", "t", "x", "theta"):
                assert np.array_equal(evs[k][f], ref[1][k][f]), (name, k, f)
        for f in ("t", "x", "theta", "acc", "c"):
            assert np.array_equal(fs[f], ref[2][f]), (name, f)
        if bm is not None:
            assert np.array_equal(bm[0], ref[3][0]) and np.array_equal(bm[1], ref[3][1]) and np.array_equal(pj, ref[4])
    lg = dict(A=P["A"], At=P["At"], y=P["y"], ny=P["ny"], mu=P["mu"], gamma0=P["gamma0"], k=10)
    r = O.spdmp_zigzag(P["Gdrop"], P["mu"], P["Gdrop"], X0[0], TH0[0], P["c"], 12.0, seed=900, adapt=True, factor=5.0, logistic=lg,
                       sigma=P["sigma"], stop_before_T=True)
    assert np.array_equal(runs["lds"][1][0]["t"], r["events"]["t"]) and np.array_equal(runs["lds"][1][0]["i"], r["events"]["i"])


def test_c4_tracked_bounds_equal_the_tracked_oracle(gpu_pkg):
    """pdmp_ensemble_set_gradient_tracking on the logistic target (zz_logistic_lds_kernel<.., TRK>): bounds from the carried sums g_j = Γ[:,j]·x,
    gd_j = Γ[:,j]·θ, the gradient still the moving evaluation.  In slices with trace refills (the sums live in HBM between launches), with and
    without the engine's path integrals: events, counters, final clocks / positions / velocities, accept counts and adapted bounds equal the
    oracle's tracked evaluation BIT FOR BIT; against the moving evaluation of the same chains: the same event indices, times to 1e-9.  What the
    mode does not serve is refused with a status."""
    pkg = gpu_pkg
    L = pkg._lib
    P = pkg.problems.logistic_problem(m=20)
    d, nch, T = P["p"], 5, 14.0
    rng = np.random.default_rng(14)
    X0 = np.tile(P["x0"], (nch, 1)) + 0.01 * rng.standard_normal((nch, d))
    TH0 = P["sigma"] * rng.choice([-1.0, 1.0], (nch, d))
    seeds = np.arange(nch, dtype=np.uint64) + 1400
    runs = {}
    for name, tracked, integrals in (("trk", True, True), ("trk_noI", True, False), ("mov", False, True)):
        with pkg.Ensemble(nch, d, adapt=True, factor=5.0, trace_capacity=300) as ens:
            ens.set_flow(pkg.ZigZag(P["Gdrop"], P["mu"], P["sigma"]))
            ens.set_target(pkg.LogisticTarget(P["A"], P["y"], P["ny"], P["mu"], P["gamma0"], 10))
            ens.set_path_integrals(integrals)
            ens.set_gradient_tracking(tracked)
            ens.set_state(0.0, X0, TH0, P["c"], seeds)
            evs = [[] for _ in range(nch)]
            for Tk in (2.5, 9.0, T):
                while True:
                    ens.run(Tk, L.RUN_STOP_BEFORE)
                    cnt = ens.counters()
                    for k in range(nch):
                        evs[k].append(ens.trace(k, counters=cnt))
                    ens.trace_reset()
                    if not L.needs_rerun(cnt["status"]):
                        break
            pj = ens.path_integrals(T, np.arange(0, d, 9)) if integrals else None
            runs[name] = (cnt, [np.concatenate(e) for e in evs], ens.final_state(), pj)
    lg = dict(A=P["A"], At=P["At"], y=P["y"], ny=P["ny"], mu=P["mu"], gamma0=P["gamma0"], k=10)
    for k in range(nch):
        r = O.spdmp_zigzag(P["Gdrop"], P["mu"], P["Gdrop"], X0[k], TH0[k], P["c"], T, seed=1400 + k, adapt=True, factor=5.0, logistic=lg,
                           sigma=P["sigma"], stop_before_T=True, tracked=True)
        assert r["status"] == 0 and len(r["events"]) > 1500
        for name in ("trk", "trk_noI"):
            cnt, evs, fs, _ = runs[name]
            for f in ("i", "t", "x", "theta"):
                assert np.array_equal(evs[k][f], r["events"][f]), (name, k, f)
            assert int(cnt["num"][k]) == r["num"] and int(cnt["ndraw_main"][k]) == r["ndraw_main"] and int(cnt["ndraw_global"][k]) == r["ndraw_global"]
            for f, g in (("t", "t"), ("x", "x"), ("theta", "theta"), ("acc", "acc"), ("c", "c")):
                assert np.array_equal(fs[f][k], r[g]), (name, k, f)
        mov = runs["mov"][1][k]
        assert np.array_equal(mov["i"], r["events"]["i"]) and np.allclose(mov["t"], r["events"]["t"], rtol=1e-9, atol=0)
    # the engine's ∫x dt of a tracked run: the same path, so the same integrals as the moving evaluation's to rounding
    assert np.allclose(runs["trk"][3], runs["mov"][3], rtol=1e-9, atol=1e-9)
    # refusals: a refresh clock; an explicit neighbourhood argument
    with pkg.Ensemble(1, d, adapt=True, factor=5.0, trace_capacity=100) as ens:
        ens.set_flow(pkg.ZigZag(P["Gdrop"], P["mu"], P["sigma"], λref=0.3))
        ens.set_target(pkg.LogisticTarget(P["A"], P["y"], P["ny"], P["mu"], P["gamma0"], 10))
        ens.set_gradient_tracking(True)
        with pytest.raises(L.PdmpError) as ei:
            ens.set_state(0.0, X0[:1], TH0[:1], P["c"], seeds[:1])
        assert ei.value.code == L.PDMP_ERR_UNSUPPORTED
