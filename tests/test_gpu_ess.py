"""ESS estimator (pdmp_ensemble_ess_*, zigzagboomerang.jl_amd/ess.py) on the device (-m gpu): a target with a KNOWN asymptotic
variance, and the stationary moments of the lattice GMRF at probe coordinates against exact diag(inv(Γ))."""
import math

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

pytestmark = pytest.mark.gpu


def _run_ess(pkg, G, c, nch, T0, B, b, seed0=0x5EED0000, trace_capacity=0):
    d = G.shape[0]
    with pkg.Ensemble(nch, d, trace_capacity=trace_capacity) as ens:
        ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
        ens.set_target(pkg.GaussianTarget(G))
        ens.set_state_synthetic(0.0, c, seed0)
        ens.run(T0, pkg._lib.RUN_STOP_BEFORE)
        ens.ess_begin(T0)
        for k in range(B):
            ens.run(T0 + (k + 1) * b, pkg._lib.RUN_STOP_BEFORE)
            ens.ess_batch(T0 + (k + 1) * b)
        sy, sy2, sm, sm2, nb, t0, t1 = ens.ess_end()
        cnt = ens.counters()
    assert nb == B and t0 == T0 and t1 == T0 + B * b and np.all(cnt["status"] == pkg._lib.CHAIN_OK)
    return sy, sy2, sm, sm2


def test_ess_of_independent_gaussians_matches_the_closed_form(gpu_pkg):
    """d independent N(0,1) coordinates, unit speeds, no refreshment: every coordinate is a 1-d ZigZag whose time average has
    asymptotic variance σ² = E|X|³ = 2·sqrt(2/π) (ess.zigzag1d_gaussian_sigma2_asym: Poisson equation of the generator), i.e.
    ESS per unit time = 1/σ² = 0.6267 -- whatever the thinning bound c is (thinning is exact)."""
    pkg = gpu_pkg
    d, nch, T0, B, b = 64, 512, 20.0, 32, 25.0
    G = sp.identity(d, format="csc")
    sig2 = pkg.ess.zigzag1d_gaussian_sigma2_asym(1.0)
    assert abs(sig2 - 1.5957691216) < 1e-9
    for c in (1.5, 4.0):
        sy, sy2, sm, sm2 = _run_ess(pkg, G, np.full(d, c), nch, T0, B, b)
        r = pkg.ess.batch_means_ess(sy, sy2, sm, sm2, nch, B, b, np.ones(d))
        # within-chain batch means: bias O(IACT / b) downwards, ~1 % noise per coordinate with 512 x 31 degrees of freedom
        assert np.all(np.abs(r["sigma2_within"] / sig2 - 1) < 0.12), (r["sigma2_within"].min(), r["sigma2_within"].max())
        assert abs(r["sigma2_within"].mean() / sig2 - 1) < 0.06
        # between-chain: unbiased at stationarity, 6 % noise per coordinate (N = 512), 1 % in the mean over 64 coordinates
        assert abs(r["sigma2_between"].mean() / sig2 - 1) < 0.05
        assert abs(np.median(r["ess_per_time"]) - 1 / sig2) < 0.06 / sig2 * 1.5
        assert np.all(np.abs(r["mean"]) < 5 * math.sqrt(sig2 / (nch * B * b)))
        assert np.allclose(r["ess"], nch * B * b * r["ess_per_time"])


def test_batch_length_bias_is_visible_and_vanishes(gpu_pkg):
    """Why round 1's number was not an ESS: batches of one time unit or less are shorter than the autocorrelation time, every
    batch mean is then ~ one draw from π and ANY batch-means estimate degenerates to "number of batches" (ESS per time ~ 1/b).
    On the closed-form target the bias is seen directly and disappears as b grows past the autocorrelation time (σ²/Var_π = 1.6)."""
    pkg = gpu_pkg
    d, nch, T0 = 64, 256, 20.0
    G = sp.identity(d, format="csc")
    truth = 1 / pkg.ess.zigzag1d_gaussian_sigma2_asym(1.0)
    got = {}
    for b, B in ((0.25, 32), (2.0, 32), (25.0, 32)):
        sy, sy2, sm, sm2 = _run_ess(pkg, G, np.full(d, 1.5), nch, T0, B, b)
        r = pkg.ess.batch_means_ess(sy, sy2, sm, sm2, nch, B, b, np.ones(d))
        got[b] = float(np.median(r["ess_per_time"]))
        pooled = (sy2 - sy * sy / (nch * B)) / (nch * B - 1)  # variance over all (chain, batch) pairs, the round-1 estimator
        got[("pooled", b)] = float(np.median(1.0 / (b * pooled)))
    assert got[0.25] > 2.5 * truth and got[("pooled", 0.25)] > 2.5 * truth  # ~ 1/b = 4 per unit time: counts batches, not mixing
    assert 1.0 < got[2.0] / truth < 1.15                                   # a few % optimistic at b ~ the autocorrelation time
    assert abs(got[25.0] / truth - 1) < 0.08 and abs(got[("pooled", 25.0)] / truth - 1) < 0.08


def test_device_paths_reproduce_the_stationary_moments(gpu_pkg):
    """Means (device path integrals) and variances (exact second moments of the traces, trace.moments) at 32 probe coordinates of
    a 48 x 48 lattice GMRF against 0 and exact diag(inv(Γ)) (sparse LU), within Monte-Carlo error: the law the chains sample is
    N(0, Γ⁻¹) -- fixture (7) of SURVEY 8c4."""
    pkg = gpu_pkg
    n = 48
    G = pkg.problems.gmrf_precision(n)
    d = n * n
    c = pkg.problems.column_norms(G)
    nch, T0, T1 = 96, 15.0, 75.0
    rng = np.random.default_rng(12)
    lu = spla.splu(G.tocsc())
    probes = np.linspace(0, d - 1, 32).astype(int)
    var_pi = np.array([lu.solve(np.eye(1, d, p).ravel())[p] for p in probes])
    # start IN the stationary law, x0 ~ N(0, Γ⁻¹): from x0 ~ N(0, I) (scripts/gaussianrandomfield.jl:29) the lattice's constant mode
    # (eigenvalue 0.01: stationary variance 1/(0.01 d) = 0.043 per coordinate, 10 % of diag(Γ⁻¹) here) is still 10 % short after 75
    # time units -- measured with this very test -- which is a statement about the target, not about the sampler
    Lc = np.linalg.cholesky(G.toarray())
    x0 = np.linalg.solve(Lc.T, rng.standard_normal((d, nch))).T.copy()
    th0 = rng.choice([-1.0, 1.0], (nch, d))
    trs, (t, x, th), (acc, num), _ = pkg.spdmp(pkg.GaussianTarget(G), 0.0, x0, th0, T1, c, pkg.ZigZag(G, np.zeros(d)), seed=300)
    m_all, v_all = [], []
    for tr in trs:
        sub = pkg.trace.subtrace(tr, probes)
        # second moments over [T0, T1] = (T1 * M2(T1) - T0 * M2(T0)) / (T1 - T0) with M2 the raw second moment over [0, T]
        ma, va = pkg.trace.moments(sub, T0)
        mb, vb = pkg.trace.moments(sub, T1)
        m1 = (T1 * mb - T0 * ma) / (T1 - T0)
        m2 = (T1 * (vb + mb * mb) - T0 * (va + ma * ma)) / (T1 - T0)
        m_all.append(m1)
        v_all.append(m2)
    m_all, v_all = np.array(m_all), np.array(v_all)
    mean = m_all.mean(0)
    var = v_all.mean(0) - mean ** 2
    se_mean = m_all.std(0, ddof=1) / math.sqrt(nch)
    assert np.all(np.abs(mean) < 5 * se_mean + 1e-3)
    se_var = v_all.std(0, ddof=1) / math.sqrt(nch)
    assert np.all(np.abs(var - var_pi) < 5 * se_var + 0.02 * var_pi), (var / var_pi)
    assert abs(np.mean(var / var_pi) - 1) < 0.04
