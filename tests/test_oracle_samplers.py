"""The oracle's samplers against the reference's STATISTICAL acceptance tests (the only pins that exist:
the reference holds no golden event sequences) and against the committed golden fixtures."""
import hashlib
import math

import numpy as np
import scipy.sparse as sp

import oracle_lib as O


def test_zigzag1d_statistics():
    """test/test1d.jl:13-27: ZigZag1d on N(π/3, 1.3), T=8000, c=10."""
    mu, s2, T = math.pi / 3, 1.3, 8000.0
    ev, acc, num = O.pdmp_zigzag1d(mu, s2, 1.01, -1.5, T, 10.0, seed=3)
    N = len(ev)
    assert T / 10 < N < T * 10
    t, x = ev["t"], ev["x"]
    est = np.sum((x[:-1] + x[1:]) / 2 * np.diff(t)) / T
    assert abs(est - mu) < 2 / math.sqrt(N)
    xa, xb, dt = x[:-1], x[1:], np.diff(t)
    m2 = np.sum(dt * (xa * xa + xa * xb + xb * xb) / 3) / t[-1]
    assert abs((m2 - est ** 2) - s2) < 2.5 / math.sqrt(N) + 0.05
    assert 0 < acc / num < 1


def test_golden_c1_chain(golden):
    ev, acc, num = O.pdmp_zigzag1d(0.0, 1.0, 1.01, -1.5, 1000.0, 10.0, seed=0x5EED0000)
    got = np.stack([ev["t"], ev["x"], ev["theta"]], axis=1)
    assert np.array_equal(got, golden["c1_events"])
    assert [acc, num] == golden["c1_acc_num"].tolist()
    assert ev["t"][-2] < 1000.0  # `while t < T` counts proposals, so the last recorded flip may precede T


def _zz_case(pkg, golden, name, scale):
    G = pkg.problems.maintest_precision(8) if name == "d8" else pkg.problems.gmrf_precision(8)
    return G, scale * G, golden[f"{name}_x0"], golden[f"{name}_th0"], golden[f"{name}_c"]


def test_golden_local_zigzag_chains(pkg, golden):
    for name, scale in (("d8", 0.9), ("grid8", 1.0)):
        G, Gb, x0, th0, c = _zz_case(pkg, golden, name, scale)
        r = O.spdmp_zigzag(Gb, None, G, x0, th0, c, 50.0, seed=1234)
        ev, want = r["events"], golden[f"{name}_events"]
        assert len(ev) == len(want)
        for f in ("t", "i", "x", "theta"):
            assert np.array_equal(ev[f], want[f]), (name, f)
        assert np.array_equal(r["acc"], golden[f"{name}_acc"]) and r["num"] == golden[f"{name}_num"][0]
        assert np.array_equal(np.stack([r["t"], r["x"], r["theta"]]), golden[f"{name}_final"])
        # structure of a FactTrace: times non-decreasing, event = state after the flip, last event beyond T
        assert np.all(np.diff(ev["t"]) >= 0) and ev["t"][-1] >= 50.0 > ev["t"][-2]
        assert r["acc"].sum() == len(ev)


def test_golden_c3_first_events(pkg, golden):
    G = pkg.problems.gmrf_precision(128)
    c = pkg.problems.column_norms(G)
    for k in range(2):
        seed = 0x5EED0000 + k
        x0, th0 = O.synthetic_state(seed, G.shape[0])
        r = O.spdmp_zigzag(G, None, G, x0, th0, c, 1e9, seed=seed, max_events=10000)
        ev = r["events"][:10000]
        assert np.array_equal(ev["i"].astype(np.uint16), golden[f"c3_chain{k}_idx"])
        h = hashlib.sha256()
        for f in ("t", "x", "theta"):
            h.update(np.ascontiguousarray(ev[f]).tobytes())
        assert h.hexdigest() == str(golden[f"c3_chain{k}_hash"][0])


def test_local_zigzag_statistics_d8(pkg):
    """test/maintest.jl:37-61 (SZigZag): Γ = S S', Z = ZigZag(0.9Γ, 0), T = 1000, discretize dt = 0.5."""
    G = pkg.problems.maintest_precision(8)
    d, T = 8, 1000.0
    rng = np.random.default_rng(2)
    x0 = rng.random(d)
    th0 = rng.choice([-1.0, -0.5, 0.5, 1.0], d)
    c = 0.7 * pkg.problems.column_norms(G)
    r = O.spdmp_zigzag(0.9 * G, None, G, x0, th0, c, T, seed=21, adapt=True, factor=1.8)
    assert r["status"] == 0
    tr = pkg.FactTrace(None, 0.0, x0, th0, r["events"])
    ts, xs = pkg.trace.discretize(tr, 0.5)
    assert np.allclose(np.diff(ts), 0.5)
    S = np.linalg.inv(G.toarray())
    assert np.mean(np.abs(xs.mean(axis=0))) < 2 / math.sqrt(T) * np.sqrt(np.diag(S)).max() * 2
    assert np.mean(np.abs(np.cov(xs.T) - S)) < 2.5 / math.sqrt(T)
    # pdmp (G = All()) gives the same chain up to rounding of the lazy clocks (src/sfact.jl:236)
    r2 = O.spdmp_zigzag(0.9 * G, None, G, x0, th0, c, 20.0, seed=21, adapt=True, factor=1.8, move_all=True)
    r1 = O.spdmp_zigzag(0.9 * G, None, G, x0, th0, c, 20.0, seed=21, adapt=True, factor=1.8)
    assert np.array_equal(r1["events"]["i"], r2["events"]["i"])
    assert np.allclose(r1["events"]["t"], r2["events"]["t"], rtol=1e-9)
    # subtrace consistency, test/maintest.jl:53-58
    J = np.arange(0, d, 2)
    ts2, xs2 = pkg.trace.discretize(pkg.trace.subtrace(tr, J), 0.5)
    assert np.allclose(ts2, ts[:len(ts2)]) and np.allclose(xs2, xs[:len(ts2)][:, J])


def test_bound_violation_is_reported(pkg):
    """adapt=false and c too small: the reference throws (src/sfact.jl:124); the oracle reports the status."""
    G = pkg.problems.gmrf_precision(4)
    d = 16
    rng = np.random.default_rng(0)
    # the bound uses 0.3Γ while the target is Γ: with a tiny c the affine bound is below the true rate
    x0, th0 = rng.standard_normal(d) * 5, rng.choice([-1.0, 1.0], d)
    r = O.spdmp_zigzag(0.3 * G, None, G, x0, th0, np.full(d, 1e-6), 50.0, seed=1)
    assert r["status"] == O.ORC_BOUND_VIOLATED
    r = O.spdmp_zigzag(0.3 * G, None, G, x0, th0, np.full(d, 1e-6), 50.0, seed=1, adapt=True)
    assert r["status"] == 0 and r["c"].max() > 1e-6


def test_refresh_branch_runs(pkg):
    """λref > 0 (src/sfact.jl:78-114,188-190): refresh events are recorded, keep |θ_i| = σ_i."""
    G = pkg.problems.gmrf_precision(4)
    d = 16
    rng = np.random.default_rng(1)
    sig = np.full(d, 0.5)
    r = O.spdmp_zigzag(G, None, G, rng.standard_normal(d), sig * rng.choice([-1.0, 1.0], d), 2 * pkg.problems.column_norms(G),
                       100.0, seed=4, lambda_ref=0.3, sigma=sig)
    assert r["status"] == 0 and r["nrefresh"] > 10
    assert len(r["events"]) == r["nacc"] + r["nrefresh"]
    assert np.all(np.abs(r["theta"]) == 0.5) and r["ndraw_global"] == 3 * r["nrefresh"]


def _bps_discretize(t0, x0, th0, t_ev, x_ev, dt):
    ts = np.arange(t0, t_ev[-1], dt)
    tt = np.concatenate([[t0], t_ev])
    xx = np.vstack([x0, x_ev])
    out = np.empty((len(ts), x0.size))
    for j in range(x0.size):
        out[:, j] = np.interp(ts, tt, xx[:, j])
    return ts, out


def test_bps_statistics_d8(pkg):
    """test/maintest.jl:156-172: BouncyParticle(Γ, 0, 0.5), c = 1.1, T = 300 (mass L = I here)."""
    G = pkg.problems.maintest_precision(8)
    d, T = 8, 300.0
    rng = np.random.default_rng(3)
    x0, th0 = rng.standard_normal(d), rng.standard_normal(d)
    r = O.pdmp_bps(G, None, x0, th0, 1.1, T, lambda_ref=0.5, seed=8, ev_cap=100000)
    assert r["status"] == 0 and r["nevents"] == len(r["t_ev"]) and r["nrefresh"] > 50
    assert np.all(np.diff(r["t_ev"]) > 0) and r["t_ev"][-1] >= T
    ts, xs = _bps_discretize(0.0, x0, th0, r["t_ev"], r["x_ev"], 0.1)
    S = np.linalg.inv(G.toarray())
    assert np.mean(np.abs(xs.mean(axis=0))) < 2 / math.sqrt(T) * 1.5
    assert np.mean(np.abs(np.cov(xs.T) - S)) < 2 / math.sqrt(T) * 1.5


def test_bps_mass_matrix_statistics_d8(pkg):
    """test/maintest.jl:156-172 as the reference runs it: BouncyParticle(Γ, 0, 0.5) carries L = cholesky(Symmetric(Γ)).L
    (src/types.jl:43), used by reflect! and refresh! (src/dynamics.jl:90-97,112-126); c = 1.1, T = 300, dt = 0.1 and the
    reference's own thresholds 2/sqrt(T)."""
    G = pkg.problems.maintest_precision(8)
    d, T = 8, 300.0
    Lc = np.linalg.cholesky(G.toarray())
    rng = np.random.default_rng(3)
    x0, th0 = rng.standard_normal(d), rng.standard_normal(d)
    ok = 0
    for seed in (8, 9, 10):
        r = O.pdmp_bps(G, None, x0, th0, 1.1, T, lambda_ref=0.5, seed=seed, ev_cap=100000, mass_L=sp.csc_matrix(np.tril(Lc)))
        assert r["status"] == 0 and r["nevents"] == len(r["t_ev"]) and r["nrefresh"] > 50
        ts, xs = _bps_discretize(0.0, x0, th0, r["t_ev"], r["x_ev"], 0.1)
        S = np.linalg.inv(G.toarray())
        ok += (np.mean(np.abs(xs.mean(axis=0))) < 2 / math.sqrt(T)) and (np.mean(np.abs(np.cov(xs.T) - S)) < 2 / math.sqrt(T))
    assert ok >= 2  # the reference's envelope at its own T (one seed in three may graze it, as in the reference's CI)
    # L = I passed explicitly is the identity-mass process bit for bit
    r0 = O.pdmp_bps(G, None, x0, th0, 1.1, 20.0, lambda_ref=0.5, seed=8, ev_cap=10000)
    r1 = O.pdmp_bps(G, None, x0, th0, 1.1, 20.0, lambda_ref=0.5, seed=8, ev_cap=10000, mass_L=sp.identity(d, format="csc"))
    assert np.array_equal(r0["t_ev"], r1["t_ev"]) and np.array_equal(r0["theta_ev"], r1["theta_ev"])
    # and a genuine factor changes the process
    r2 = O.pdmp_bps(G, None, x0, th0, 1.1, 20.0, lambda_ref=0.5, seed=8, ev_cap=10000, mass_L=sp.csc_matrix(np.tril(Lc)))
    assert not np.array_equal(r0["t_ev"][:50], r2["t_ev"][:50])


def test_bps_reflection_preserves_mass_norm(pkg):
    """reflect! (src/dynamics.jl:90-93) is the reflection in the metric M = L L': θ'Mθ... the quantity it conserves is
    ‖L'θ‖² only up to the gradient direction; what holds exactly is ⟨∇ϕ, θ⟩ -> −⟨∇ϕ, θ⟩.  Checked on the events of a run with
    no refresh in between (consecutive reflection events)."""
    G = pkg.problems.maintest_precision(8)
    d = 8
    Lc = np.tril(np.linalg.cholesky(G.toarray()))
    rng = np.random.default_rng(5)
    x0, th0 = rng.standard_normal(d), rng.standard_normal(d)
    r = O.pdmp_bps(G, None, x0, th0, 1.1, 50.0, lambda_ref=1e-9, seed=2, ev_cap=10000, mass_L=sp.csc_matrix(Lc))
    assert r["status"] == 0 and r["nrefresh"] == 0 and r["nevents"] > 20
    Gd = G.toarray()
    Minv = np.linalg.inv(Lc @ Lc.T)
    th_prev = th0
    for k in range(r["nevents"]):
        g = Gd @ r["x_ev"][k]
        th_new = r["theta_ev"][k]
        z = Minv @ g
        expect = th_prev - 2 * (g @ th_prev) / (g @ z) * z
        assert np.allclose(th_new, expect, rtol=1e-10, atol=1e-12)
        assert np.isclose(g @ th_new, -(g @ th_prev), rtol=1e-9)
        th_prev = th_new


def test_bps_local_bound_and_subsample(pkg):
    """c::LocalBound for the non-factorised sampler (src/not_fact_samplers.jl:29-31, renew branch :65-71) and the `subsample`
    keyword (:53,90: an accepted reflection does not end pdmp_inner!, only refreshes are recorded)."""
    G = pkg.problems.maintest_precision(8)
    d, T = 8, 300.0
    rng = np.random.default_rng(4)
    x0, th0 = rng.standard_normal(d), rng.standard_normal(d)
    r = O.pdmp_bps(G, None, x0, th0, 1.1, T, lambda_ref=0.5, seed=3, ev_cap=100000, local_bound=True)
    assert r["status"] == 0 and r["nrefresh"] > 50
    # a renew consumes one draw and no proposal: draws = 2 (setup) + per refresh (64⌈d/128⌉ + 2) + 2 per proposal + renewals
    renewals = r["ndraw_main"] - 2 - r["nrefresh"] * (64 + 2) - 2 * r["num"]
    assert renewals >= 0
    # test/maintest.jl:182 uses LocalBound(c = 20): the horizon 2√d/c/‖θ‖ is then short and bounds do expire
    r20 = O.pdmp_bps(G, None, x0, th0, 20.0, 30.0, lambda_ref=0.5, seed=3, ev_cap=100000, local_bound=True)
    assert r20["status"] == 0 and r20["ndraw_main"] - 2 - r20["nrefresh"] * (64 + 2) - 2 * r20["num"] > 10
    ts, xs = _bps_discretize(0.0, x0, th0, r["t_ev"], r["x_ev"], 0.1)
    S = np.linalg.inv(G.toarray())
    assert np.mean(np.abs(xs.mean(axis=0))) < 2 / math.sqrt(T) * 1.5
    assert np.mean(np.abs(np.cov(xs.T) - S)) < 2 / math.sqrt(T) * 1.5
    rs = O.pdmp_bps(G, None, x0, th0, 1.1, 100.0, lambda_ref=0.5, seed=3, ev_cap=100000, subsample=True)
    assert rs["status"] == 0 and rs["nevents"] == rs["nrefresh"] and rs["nacc"] > rs["nrefresh"]


def test_golden_bps16(golden):
    d = 16
    r = O.pdmp_bps(sp.identity(d, format="csc"), None, golden["bps16_x0"], golden["bps16_th0"], 1e-3, 1e9, lambda_ref=1.0,
                   seed=99, max_events=200, ev_cap=200)
    assert np.array_equal(r["t_ev"], golden["bps16_t"])
    assert np.array_equal(r["x_ev"][-1], golden["bps16_x_last"]) and np.array_equal(r["theta_ev"][-1], golden["bps16_th_last"])
    assert [r["num"], r["nacc"], r["nrefresh"]] == golden["bps16_counts"].tolist()


def test_sticky_1d_statistics(pkg):
    """test/sticky.jl:7-36: P(X≠0) = w, E X = wμ, E X² = w(σ²+μ²) with the closed-form w (:30)."""
    sig2, mu, kappa, T = 0.5, 0.9, 1.5, 2000.0
    Gf = sp.csc_matrix(np.array([[1.0]]))
    Gt = sp.csc_matrix(np.array([[1 / sig2]]))
    r = O.sspdmp_zigzag(Gf, np.array([0.0]), Gt, np.array([1.0]), np.array([0.8]), np.array([20.0]), np.array([kappa]), T,
                        target_mu=np.array([mu]), seed=1)
    assert r["status"] == 0
    tr = pkg.FactTrace(None, 0.0, np.array([1.0]), np.array([0.8]), r["events"])
    ts, xs = pkg.trace.discretize(tr, 0.2)
    x = xs[:, 0]
    sig = math.sqrt(sig2)
    w = math.sqrt(2 * math.pi) * sig / (math.sqrt(2 * math.pi) * sig + math.exp(-0.5 * mu ** 2 / sig2) / kappa)
    assert abs(np.mean(x != 0) - w) < 2.5 / math.sqrt(T)
    assert abs(np.mean(x) - w * mu) < 5.0 / math.sqrt(T)
    assert abs(np.mean(x ** 2) - w * (sig2 + mu ** 2)) < 5.0 / math.sqrt(T)


def test_golden_sticky1d(golden):
    Gf = sp.csc_matrix(np.array([[1.0]]))
    Gt = sp.csc_matrix(np.array([[2.0]]))
    r = O.sspdmp_zigzag(Gf, np.array([0.0]), Gt, np.array([1.0]), np.array([0.8]), np.array([20.0]), np.array([1.5]), 200.0,
                        target_mu=np.array([0.9]), seed=5)
    for f in ("t", "i", "x", "theta"):
        assert np.array_equal(r["events"][f], golden["sticky1d_events"][f])
    assert [r["num"], r["nacc"]] == golden["sticky1d_counts"].tolist()


def test_sticky_d8_no_sticking(pkg):
    """test/sticky.jl:39-65: κ = 1000 ("dont stop, actually") reproduces the Gaussian moments."""
    G = pkg.problems.maintest_precision(8)
    d, T = 8, 1000.0
    rng = np.random.default_rng(1)
    x0 = rng.random(d)
    th0 = rng.choice([-1.0, -0.5, 0.5, 1.0], d)
    c = 0.7 * pkg.problems.column_norms(G)
    r = O.sspdmp_zigzag(0.9 * G, None, G, x0, th0, c, np.full(d, 1000.0), T, seed=12, adapt=True)
    assert r["status"] == 0
    tr = pkg.FactTrace(None, 0.0, x0, th0, r["events"])
    ts, xs = pkg.trace.discretize(tr, 0.5)
    S = np.linalg.inv(G.toarray())
    assert np.mean(np.abs(xs.mean(axis=0))) < 2 / math.sqrt(T) * 1.5
    assert np.mean(np.abs(np.cov(xs.T) - S)) < 2.5 / math.sqrt(T)


def test_logistic_subsampled_target_statistics(pkg):
    """Config C4 (scripts/logistic.jl): spdmp with ∇ϕmoving (SelfMoving, k = 10 subsample rows, control variate at the
    mode μ), Zdrop = ZigZag(Γdrop, μ, σ), c = 0.01, adapt = true, factor = 5.  The posterior is close to its Laplace
    approximation N(μ, Γ⁻¹): time averages land within a fraction of a posterior sd of μ."""
    P = pkg.problems.logistic_problem(m=20)
    assert (P["n"], P["p"]) == (8840, 442)  # README.md:45
    lg = dict(A=P["A"], At=P["At"], y=P["y"], ny=P["ny"], mu=P["mu"], gamma0=P["gamma0"], k=10)
    r = O.spdmp_zigzag(P["Gdrop"], P["mu"], P["Gdrop"], P["x0"], P["theta0"], P["c"], 150.0, seed=3, adapt=True, factor=5.0,
                       logistic=lg, sigma=P["sigma"])
    assert r["status"] == 0 and r["ndraw_global"] == 10 * r["num"] and r["c"].max() > P["c"].max()
    tr = pkg.FactTrace(None, 0.0, P["x0"], P["theta0"], r["events"])
    m, v = pkg.trace.moments(tr, 150.0)
    sd = np.sqrt(np.diag(np.linalg.inv(P["G"].toarray())))
    z = (m - P["mu"]) / sd
    assert np.abs(z).mean() < 0.5 and np.abs(z).max() < 4.0
    assert 0.5 < np.median(v / sd ** 2) < 1.5


def test_factboomerang_statistics_d8(pkg):
    """test/maintest.jl:112-137 (SFactBoomerang): Z = FactBoomerang(1.2Γ, 0, 0.3), target ∇ϕ(x,i) = idot(Z.Γ, i, x), T = 3000,
    discretize dt = 0.5 with the Boomerang rotation; the stationary law is N(0, inv(1.2Γ))."""
    G = pkg.problems.maintest_precision(8)
    d, T = 8, 3000.0
    Gz = sp.csc_matrix(1.2 * G)
    rng = np.random.default_rng(4)
    x0 = rng.random(d)
    sig = np.asarray(Gz.diagonal()) ** -0.5
    th0 = sig * rng.standard_normal(d)
    c = pkg.problems.column_norms(G)
    r = O.spdmp_zigzag(Gz, np.zeros(d), Gz, x0, th0, c, T, seed=9, lambda_ref=0.3, sigma=sig, factboomerang=True)
    assert r["status"] == 0 and r["nrefresh"] > 500 and len(r["events"]) == r["nacc"] + r["nrefresh"]
    Z = pkg.FactBoomerang(Gz, np.zeros(d), 0.3)
    tr = pkg.FactTrace(Z, 0.0, x0, th0, r["events"])
    ts, xs = pkg.trace.discretize(tr, 0.5)
    S = np.linalg.inv(Gz.toarray())
    assert np.mean(np.abs(xs.mean(axis=0))) < 2 / math.sqrt(T) * 1.5
    assert np.mean(np.abs(np.cov(xs.T) - S)) < 4 / math.sqrt(T)


def test_adaptscale_tunes_sigma_towards_the_target_reflection_rate(pkg):
    """src/sfact.jl:86-91: under adaptscale the ZigZag refresh branch steers every coordinate towards 0.3 accepted
    reflections per unit time by rescaling σ[i] (speed); no event-level pin exists in the reference's tests."""
    G = pkg.problems.maintest_precision(8)
    d = 8
    rng = np.random.default_rng(1)
    sg = np.full(d, 2.0)
    x0 = rng.standard_normal(d)
    th0 = sg * rng.choice([-1.0, 1.0], d)
    T = 4000.0
    r = O.spdmp_zigzag(G, np.zeros(d), G, x0, th0, np.full(d, 10.0), T, seed=3, lambda_ref=0.5, sigma=sg,
                       adaptscale=True, adapt=True)
    assert r["status"] == 0 and r["nrefresh"] > 1000
    rate = r["acc"] / T
    assert np.all(rate > 0.2) and np.all(rate < 0.6), rate
    assert np.all(r["sigma"] < 2.0) and np.all(r["sigma"] > 0.2)
    # speeds follow σ: |θ_i| of the final state equals the tuned σ_i or the σ_i of an earlier refresh (never 2.0 again)
    assert np.all(np.abs(r["theta"]) < 2.0)
    # without adaptscale σ is left alone
    r0 = O.spdmp_zigzag(G, np.zeros(d), G, x0, th0, np.full(d, 10.0), 50.0, seed=3, lambda_ref=0.5, sigma=sg, adapt=True)
    assert np.array_equal(r0["sigma"], sg)


def test_boomerang_statistics_d8(pkg):
    """test/maintest.jl:139-154 ("Boomerang": Γ = S S' target, λref = 0.5, c = 16, T = 3000, dt = 0.1): the reference pins
    mean(abs.(mean(xs))) < 2/sqrt(T); its covariance check is @test_broken with L = cholesky(Γ).L.  With the identity mass
    this build implements, the covariance matches inv(Γ) within the 2.5/sqrt(T) the reference hoped for."""
    G = pkg.problems.maintest_precision(8)
    d = 8
    rng = np.random.default_rng(1)
    T = 3000.0
    r = O.pdmp_bps(G, np.zeros(d), rng.standard_normal(d), rng.standard_normal(d), 16.0, T, lambda_ref=0.5, seed=5,
                   ev_cap=400000, boomerang_mu=np.zeros(d))
    assert r["status"] == 0 and r["nevents"] > 1000
    B = pkg.Boomerang(__import__("scipy.sparse").sparse.identity(d, format="csc"), np.zeros(d), 0.5)
    tr = pkg.PDMPTrace(B, 0.0, np.zeros(d), np.zeros(d), r["t_ev"], r["x_ev"], r["theta_ev"])
    tr.x0, tr.θ0 = r["x_ev"][0], r["theta_ev"][0]  # start the grid at the first event (initial state not kept by the helper)
    tr.t0 = r["t_ev"][0]
    tr.t, tr.x, tr.θ = r["t_ev"][1:], r["x_ev"][1:], r["theta_ev"][1:]
    ts, xs = pkg.trace.discretize(tr, 0.1)
    assert len(ts) > 0.9 * T / 0.1
    assert np.mean(np.abs(xs.mean(0))) < 2 / np.sqrt(T)
    assert np.mean(np.abs(np.cov(xs.T) - np.linalg.inv(G.toarray()))) < 2.5 / np.sqrt(T)


def test_golden2_newer_paths():
    """tests/golden/golden2.npz: logistic target (C4), FactBoomerang (spdmp / pdmp All), adaptscale, Boomerang, parallel_spdmp -- the oracle
    still reproduces the committed index sequences, counters and payload hashes."""
    import importlib.util
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_golden2", os.path.join(here, "golden", "make_golden2.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    gold = np.load(os.path.join(here, "golden", "golden2.npz"), allow_pickle=False)
    for name, fn in mg.cases(gold).items():
        r = fn()
        assert r["status"] == 0
        for key, val in mg.summarize(name, r).items():
            assert np.array_equal(val, gold[key]), key


def test_threaded_parallel_spdmp_statistics():
    """test/testparallel.jl:22-73 ("Parallel ZigZag": d = 20 tridiagonal Γ, K = 2 chunks, bound Γ2 = Γ without the cross-chunk
    entries, c = 5‖Γ[:, i]‖, Δ = 0.05, T = 1000): 0.1/√T < mean|mean(tr)| < 4/√T and mean|cov − Γ⁻¹| < 4/√T, plus the Partition
    index map round trip (:4-20).  The workers of a round touch disjoint data, so the run is deterministic whatever the thread timing
    (second half of the test) -- which is what lets the device's partitioned mode be checked bit for bit against it."""
    import scipy.sparse as sp
    from __graft_entry__ import load_package
    pkg = load_package()
    d, K, T, delta = 20, 2, 1000.0, 0.05
    G = sp.diags([np.ones(d), -0.4 * np.ones(d - 1), -0.4 * np.ones(d - 1)], [0, 1, -1], format="csc")
    k = d // K
    coo = G.tocoo()
    keep = (coo.row // k) == (coo.col // k)
    G2 = sp.csc_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=G.shape)
    G2.sort_indices()
    for i in range(d):  # partition(i) = (chunk, offset); partition(chunk, offset) = i
        q1, q2 = divmod(i, k)
        assert q1 * k + q2 == i and 0 <= q1 < K
    rng = np.random.default_rng(1)
    x0 = 0.1 * rng.standard_normal(d)
    th0 = rng.choice([-1.0, 1.0], d)
    c = 5 * pkg.problems.column_norms(G)
    r = O.parallel_spdmp(G2, None, G, x0, th0, c, T, K, delta, seed=7)
    assert r["status"] == 0 and r["nacc"] == len(r["events"]) > 2000 and r["rounds"] > 10
    assert np.all(np.diff(r["events"]["t"]) >= 0)
    tr = pkg.FactTrace(pkg.ZigZag(G2, np.zeros(d)), 0.0, x0, th0, r["events"])
    assert 0.1 / np.sqrt(T) < np.mean(np.abs(pkg.trace.mean(tr))) < 4 / np.sqrt(T)
    ts, xs = pkg.trace.discretize(tr, 0.5)
    assert np.mean(np.abs(np.cov(xs.T) - np.linalg.inv(G.toarray()))) < 4 / np.sqrt(T)
    for _ in range(3):  # deterministic: identical events, counters, rounds and final state on every run
        r2 = O.parallel_spdmp(G2, None, G, x0, th0, c, 100.0, K, delta, seed=7)
        r3 = O.parallel_spdmp(G2, None, G, x0, th0, c, 100.0, K, delta, seed=7)
        assert np.array_equal(r2["events"], r3["events"]) and r2["num"] == r3["num"] and r2["rounds"] == r3["rounds"]
        assert np.array_equal(r2["x"], r3["x"]) and np.array_equal(r2["t"], r3["t"])
    # a bound that couples the chunks is refused ("Upper bounds may not depend across chunks.", src/parallel.jl:124-127)
    assert O.parallel_spdmp(G, None, G, x0, th0, c, 1.0, K, delta, seed=7)["status"] != 0


def test_local_bound_statistics_d8(pkg):
    """spdmp with c::LocalBound (src/local.jl): no reference test pins it (performance/smartbound.jl only runs it), so the pin is the
    ZigZag envelope of test/maintest.jl:32-33 on the same target, plus the structure of the scheme: horizons expire (`renew` events
    draw without proposing) and, the Gaussian local bound being exact, no bound is ever adapted."""
    G = pkg.problems.maintest_precision(8)
    d = 8
    rng = np.random.default_rng(1)
    x0 = rng.random(d)
    th0 = rng.choice([-1.0, 1.0], d)
    T = 2000.0
    c = 0.5 * pkg.problems.column_norms(G)
    r = O.spdmp_zigzag(G, None, G, x0, th0, c, T, seed=3, local_bound=True, adapt=True)
    assert r["status"] == 0 and np.array_equal(r["c"], c)
    assert r["ndraw_main"] > d + r["num"] + r["nacc"] + (r["num"] - r["nacc"])  # more draws than proposals explain: renew events
    tr = pkg.FactTrace(pkg.ZigZag(G, np.zeros(d)), 0.0, x0, th0, r["events"])
    ts, xs = pkg.trace.discretize(tr, 0.5)
    assert np.mean(np.abs(xs.mean(0))) < 2 / np.sqrt(T)
    assert np.mean(np.abs(np.cov(xs.T) - np.linalg.inv(G.toarray()))) < 2.5 / np.sqrt(T)


def test_neighbourhood_argument_of_the_oracle(pkg):
    """G of spdmp(∇ϕ, t0, x0, θ0, T, c, G, F, ...) (src/sfact.jl:162,171-179): G = G1 given explicitly is Matched(); G ⊉ G1 trips the
    reference's @assert (:177); a larger G changes only which clocks a proposal moves -- the law of the chain is the target's either way."""
    import scipy.sparse as sp
    G = pkg.problems.gmrf_precision(6, 0.5)
    d = 36
    rng = np.random.default_rng(2)
    x0, th0 = rng.standard_normal(d), rng.choice([-1.0, 1.0], d)
    c = 1.5 * pkg.problems.column_norms(G)
    a = O.spdmp_zigzag(G, None, G, x0, th0, c, 30.0, seed=4)
    b = O.spdmp_zigzag(G, None, G, x0, th0, c, 30.0, seed=4, G=G)
    assert np.array_equal(a["events"], b["events"]) and np.array_equal(a["t"], b["t"])
    assert O.spdmp_zigzag(G, None, G, x0, th0, c, 1.0, seed=4, G=sp.identity(d, format="csc"))["status"] == 4
    full = sp.csc_matrix(np.ones((d, d)))
    m = O.spdmp_zigzag(G, None, G, x0, th0, c, 30.0, seed=4, G=full)
    e = O.spdmp_zigzag(G, None, G, x0, th0, c, 30.0, seed=4, move_all=True)  # G = All() moves the same coordinates ...
    assert np.array_equal(m["events"], e["events"]) and np.array_equal(m["x"], e["x"])  # ... so the two coincide (G2 is empty either way)
