"""Bouncy particle sampler on gfx950 vs the CPU oracle, through the C ABI (-m gpu): bit-exact events and state."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle_lib as O

pytestmark = pytest.mark.gpu


def check(pkg, G, mu, x0, th0, c, T, lam, rho=0.0, adapt=False, seed=5, factor=2.0):
    B = pkg.BouncyParticle(G, np.zeros(G.shape[0]) if mu is None else mu, lam, rho)
    tr, (t, x, th), (acc, num), cout = pkg.pdmp(None, 0.0, x0, th0, T, c, B, adapt=adapt, seed=seed, factor=factor)
    for k in range(x0.shape[0]):
        r = O.pdmp_bps(G, mu, x0[k], th0[k], c, T, lambda_ref=lam, rho=rho, adapt=adapt, factor=factor, seed=seed + k,
                       ev_cap=200000)
        assert r["status"] == 0
        assert len(tr[k].t) == r["nevents"], (k, len(tr[k].t), r["nevents"])
        assert np.array_equal(tr[k].t, r["t_ev"]) and np.array_equal(tr[k].x, r["x_ev"]) and np.array_equal(tr[k].θ, r["theta_ev"])
        assert (int(acc[k]), int(num[k])) == (r["nacc"], r["num"])
        assert t[k] == r["t"] and np.array_equal(x[k], r["x"]) and np.array_equal(th[k], r["theta"]) and cout[k] == r["c"]
    return tr


@pytest.mark.parametrize("d", [1, 7, 64, 100, 1024])
def test_isotropic_gaussian_config_c2_shape(gpu_pkg, d):
    """Γ = I, λref = 1, c = 1e-3 (values of scripts/not_fact.jl:23-28); d = 1024 is config C2's dimension."""
    rng = np.random.default_rng(d)
    nch = 3
    tr = check(gpu_pkg, sp.identity(d, format="csc"), None, rng.standard_normal((nch, d)), rng.standard_normal((nch, d)),
               1e-3, 25.0 if d < 1024 else 8.0, 1.0, seed=40 + d)
    assert all(np.all(np.diff(q.t) > 0) for q in tr)


def test_general_sparse_precision_and_mean(gpu_pkg):
    """test/maintest.jl:156-172: Γ = S S', λref = 0.5, c = 1.1 (mass L = I), plus a non-zero μ and ρ > 0."""
    G = gpu_pkg.problems.maintest_precision(8)
    rng = np.random.default_rng(3)
    x0, th0 = rng.standard_normal((4, 8)), rng.standard_normal((4, 8))
    check(gpu_pkg, G, None, x0, th0, 1.1, 60.0, 0.5, seed=8)
    check(gpu_pkg, G, rng.standard_normal(8), x0, th0, 1.1, 40.0, 0.7, rho=0.3, seed=9)
    G2 = gpu_pkg.problems.gmrf_precision(10)  # d = 100: two slots per lane, pentadiagonal gather through LDS
    check(gpu_pkg, G2, None, rng.standard_normal((2, 100)), rng.standard_normal((2, 100)), 0.5, 6.0, 1.0, seed=10)


def test_adapt_and_violation(gpu_pkg):
    G = sp.identity(4, format="csc") * 3.0
    rng = np.random.default_rng(1)
    x0, th0 = 4 * rng.standard_normal((2, 4)), rng.standard_normal((2, 4))
    # the bound of a Gaussian target with the same Γ is exact: force violations through the refresh-free tail c < 0
    with pytest.raises(RuntimeError, match="Tuning parameter `c` too small"):
        gpu_pkg.pdmp(None, 0.0, x0, th0, 30.0, -0.5, gpu_pkg.BouncyParticle(G, np.zeros(4), 1.0))
    check(gpu_pkg, G, None, x0, th0, -0.5, 30.0, 1.0, adapt=True, seed=5, factor=-2.0)


def test_golden_bps16(gpu_pkg, golden):
    d = 16
    B = gpu_pkg.BouncyParticle(sp.identity(d, format="csc"), np.zeros(d), 1.0)
    with gpu_pkg.Ensemble(1, d, sampler=gpu_pkg._lib.SAMPLER_BPS, trace_capacity=200) as ens:
        ens.set_flow_bps(B)
        ens.set_state_bps(0.0, golden["bps16_x0"][None], golden["bps16_th0"][None], 1e-3, np.array([99], dtype=np.uint64))
        ens.run(1e9)
        cnt = ens.counters()
        assert cnt["status"][0] == gpu_pkg._lib.CHAIN_TRACE_FULL and cnt["ntrace"][0] == 200
        t, x, th = ens.bps_trace(0, counters=cnt)
    assert np.array_equal(t, golden["bps16_t"])
    assert np.array_equal(x[-1], golden["bps16_x_last"]) and np.array_equal(th[-1], golden["bps16_th_last"])
    assert [int(cnt["num"][0]), int(cnt["nacc"][0]), int(cnt["nrefresh"][0])] == golden["bps16_counts"].tolist()


def test_slicing_is_exact(gpu_pkg):
    pkg = gpu_pkg
    d = 32
    G = sp.identity(d, format="csc")
    rng = np.random.default_rng(6)
    x0, th0 = rng.standard_normal((3, d)), rng.standard_normal((3, d))
    with pkg.Ensemble(3, d, sampler=pkg._lib.SAMPLER_BPS, trace_capacity=7) as ens:
        ens.set_flow_bps(pkg.BouncyParticle(G, np.zeros(d), 1.0))
        ens.set_state_bps(0.0, x0, th0, 1e-3, np.arange(3, dtype=np.uint64) + 70)
        ts = [[] for _ in range(3)]
        for Tk, flag in ((3.3, pkg._lib.RUN_STOP_BEFORE), (11.0, pkg._lib.RUN_STOP_BEFORE), (20.0, pkg._lib.RUN_REFERENCE_TAIL)):
            while True:
                ens.run(Tk, flag)
                cnt = ens.counters()
                for k in range(3):
                    ts[k].append(ens.bps_trace(k, counters=cnt)[0])
                ens.trace_reset()
                if not np.any(cnt["status"] == pkg._lib.CHAIN_TRACE_FULL):
                    break
        fs = ens.bps_final_state()
    for k in range(3):
        r = O.pdmp_bps(G, None, x0[k], th0[k], 1e-3, 20.0, lambda_ref=1.0, seed=70 + k, ev_cap=10000)
        assert np.array_equal(np.concatenate(ts[k]), r["t_ev"]) and np.array_equal(fs["x"][k], r["x"])


def test_boomerang_matches_oracle(gpu_pkg):
    """pdmp(∇ϕ!, t0, x0, θ0, T, c, B::Boomerang) (test/maintest.jl:139-154: Γ = S S' target, λref = 0.5, c = 16) with the
    identity mass the device implements: rotation, grad_correct!, the constant bound; bit-exact events and state.  Also a
    diagonal target with non-zero means (register-only path), d = 100 (two slots per lane) and adapt."""
    pkg = gpu_pkg
    rng = np.random.default_rng(21)

    def chk(Gt, mut, muf, x0, th0, c, T, lam, rho=0.0, adapt=False, seed=5):
        d = Gt.shape[0]
        B = pkg.Boomerang(sp.identity(d, format="csc"), muf, lam, rho)
        tr, (t, x, th), (acc, num), cout = pkg.pdmp(pkg.GaussianTarget(Gt, mut), 0.0, x0, th0, T, c, B, adapt=adapt, seed=seed)
        for k in range(x0.shape[0]):
            r = O.pdmp_bps(Gt, mut, x0[k], th0[k], c, T, lambda_ref=lam, rho=rho, adapt=adapt, seed=seed + k, ev_cap=200000,
                           boomerang_mu=muf)
            assert r["status"] == 0 and r["nevents"] > 10
            assert len(tr[k].t) == r["nevents"], (k, len(tr[k].t), r["nevents"])
            assert np.array_equal(tr[k].t, r["t_ev"]) and np.array_equal(tr[k].x, r["x_ev"]) and np.array_equal(tr[k].θ, r["theta_ev"])
            assert (int(acc[k]), int(num[k])) == (r["nacc"], r["num"])
            assert t[k] == r["t"] and np.array_equal(x[k], r["x"]) and np.array_equal(th[k], r["theta"]) and cout[k] == r["c"]
        return tr

    G = pkg.problems.maintest_precision(8)
    tr = chk(G, None, np.zeros(8), rng.standard_normal((3, 8)), rng.standard_normal((3, 8)), 16.0, 300.0, 0.5, seed=60)
    # the discretised path of a Boomerang trace rotates between events: sample mean near 0 (loose, short run)
    ts, xs = pkg.trace.discretize(tr[0], 0.1)
    assert len(ts) > 2000 and np.mean(np.abs(xs.mean(0))) < 0.5
    Gd = sp.diags(0.5 + rng.random(20), format="csc")
    chk(Gd, rng.standard_normal(20), 0.3 * rng.standard_normal(20), rng.standard_normal((2, 20)), rng.standard_normal((2, 20)),
        4.0, 100.0, 1.0, rho=0.4, seed=61)
    G2 = pkg.problems.gmrf_precision(10)
    chk(G2, None, np.zeros(100), rng.standard_normal((2, 100)), rng.standard_normal((2, 100)), 2.0, 30.0, 0.8, adapt=True, seed=62)
