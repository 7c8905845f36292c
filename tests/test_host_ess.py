"""Host-side statistics of the ESS leg, on the CPU (no engine): the multiscale batch-means estimator against a process whose
asymptotic variance is known in closed form, the exact stationary sample and marginal variances of the lattice GMRF against a
sparse solve."""
import numpy as np
import scipy.sparse.linalg as spla


def _ou_path_integrals(rng, nchains, nbatch, b, tau, nsub=8):
    """J(k b) = ∫_0^{kb} x dt for stationary Ornstein-Uhlenbeck paths, Var = 1, correlation time tau: exact transitions on a fine grid,
    trapezoid integral (the grid is much finer than tau, and the same error would bias every batch length alike)."""
    h = b / nsub
    rho = np.exp(-h / tau)
    x = rng.standard_normal(nchains)
    J = np.zeros((nbatch + 1, nchains, 1))
    acc = np.zeros(nchains)
    for k in range(nbatch):
        for _ in range(nsub):
            xn = rho * x + np.sqrt(1.0 - rho * rho) * rng.standard_normal(nchains)
            acc += 0.5 * h * (x + xn)
            x = xn
        J[k + 1, :, 0] = acc
    return J


def test_multiscale_estimator_recovers_the_asymptotic_variance_of_an_ou_process(pkg):
    """σ²_asym = 2 τ Var for the OU process; batch means at length s are biased by the factor 1 − (τ/s)(1 − e^{−s/τ}), which the dyadic table
    must show (rising with s) and the Richardson value must largely remove."""
    rng = np.random.default_rng(7)
    tau, b, B, N = 4.0, 2.0, 16, 4000
    J = _ou_path_integrals(rng, N, B, b, tau)
    r = pkg.ess.multiscale_ess(J, b, np.ones(1))
    assert np.allclose(r["scales"], b * 2.0 ** np.arange(5))
    s2 = r["sigma2"][:, 0]
    assert np.all(np.diff(s2) > 0)  # positive autocorrelation: longer batches see more of it
    exact = 2.0 * tau
    theory = exact * (1.0 - (tau / r["scales"]) * (1.0 - np.exp(-r["scales"] / tau)))
    assert np.allclose(s2, theory, rtol=0.06), (s2, theory)
    assert abs(r["sigma2_extrapolated"][0] / exact - 1.0) < 0.08 < 1.0 - s2[-1] / exact  # the extrapolation beats the largest batch length
    assert np.allclose(r["ess"][:, 0], N * B * b / s2)
    assert abs(r["last_doubling"][0] - (s2[-1] / s2[-2] - 1.0)) < 1e-12


def test_multiscale_estimator_uses_the_known_mean(pkg):
    rng = np.random.default_rng(8)
    J = _ou_path_integrals(rng, 500, 8, 1.0, 0.5)
    shifted = J + 3.0 * np.arange(9)[:, None, None]  # the same paths around mean 3
    a = pkg.ess.multiscale_ess(J, 1.0, np.ones(1))
    c = pkg.ess.multiscale_ess(shifted, 1.0, np.ones(1), mean=3.0)
    assert np.allclose(a["sigma2"], c["sigma2"], rtol=1e-9)


def test_batch_means_ess_within_and_between_agree_on_iid_batches(pkg):
    rng = np.random.default_rng(9)
    N, B, b, d = 3000, 8, 2.0, 3
    Y = rng.standard_normal((N, B, d)) / np.sqrt(b)  # independent batch means of variance 1/b: σ²_asym = 1
    M = Y.mean(axis=1)
    r = pkg.ess.batch_means_ess(Y.sum(axis=(0, 1)), (Y * Y).sum(axis=(0, 1)), M.sum(axis=0), (M * M).sum(axis=0), N, B, b, np.ones(d))
    assert np.allclose(r["sigma2_within"], 1.0, atol=0.05) and np.allclose(r["sigma2_between"], 1.0, atol=0.1)
    assert np.allclose(r["ess"], N * B * b / r["sigma2_within"])


def test_gmrf_marginal_variances_equal_a_sparse_solve(pkg):
    n = 12
    G = pkg.problems.gmrf_precision(n)
    var = pkg.problems.gmrf_marginal_variances(n)
    lu = spla.splu(G.tocsc())
    for i in (0, 1, n - 1, n, 5 * n + 7, n * n - 1):
        e = np.zeros(n * n)
        e[i] = 1.0
        assert abs(lu.solve(e)[i] - var[i]) < 1e-10 * var[i]


def test_gmrf_stationary_sample_has_the_targets_covariance(pkg):
    """x0 ~ N(0, Γ⁻¹): Γ x has covariance Γ, so the sample second moments of y = Γ x match Γ's entries (diagonal 4.01 / 3.01 / 2.01,
    neighbours −1, everything else 0), and the marginal variances match the closed form."""
    n, N = 10, 40000
    rng = np.random.default_rng(10)
    x = pkg.problems.gmrf_stationary_sample(n, N, rng)
    G = pkg.problems.gmrf_precision(n)
    assert np.allclose(x.var(axis=0), pkg.problems.gmrf_marginal_variances(n), rtol=0.06)
    y = (G @ x.T).T
    C = (y.T @ y) / N
    assert np.max(np.abs(C - G.toarray())) < 0.12
