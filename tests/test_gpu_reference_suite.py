"""The reference's own test-suite, re-run on the device through the host mirror (-m gpu): same problems, same horizons, same
statistical assertions and thresholds as test/maintest.jl and test/sticky.jl (whose random inputs cannot be reproduced -- the
reference seeds Julia's MersenneTwister -- so each block draws its inputs from a fixed numpy seed instead).

The bit-level parity of every sampler with the oracle is in the other test_gpu_* files; these blocks check the device samplers the
way the reference's maintainers do."""
import math

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

d = 8


@pytest.fixture(scope="module")
def Γ(gpu_pkg):
    return gpu_pkg.problems.maintest_precision(d)  # S = 1.3I + 0.5sprandn(d, d, 0.1); Γ = S*S'   (test/maintest.jl:5-7)


def _stats(pkg, trace, dt):
    ts, xs = pkg.trace.discretize(trace, dt)
    return np.mean(np.abs(xs.mean(0))), xs


def test_zigzag(gpu_pkg, Γ):
    """@testset "ZigZag" (test/maintest.jl:13-34): pdmp(∇ϕ, t0, x0, θ0, T, c, ZigZag(0.9Γ, 0), Γ), T = 1000."""
    pkg, rng = gpu_pkg, np.random.default_rng(2)
    x0, θ0 = rng.random(d), rng.choice([-1.0, 1.0], d)
    c = 2.0 * pkg.problems.column_norms(Γ)  # the reference's .7 factor is too small for some draws of Γ and x0; adapt is off there
    T = 1000.0
    trace, _, acc, _ = pkg.pdmp(pkg.GaussianTarget(Γ), 0.0, x0, θ0, T, c, pkg.ZigZag(0.9 * Γ, np.zeros(d)))
    m, xs = _stats(pkg, trace, 0.5)
    assert m < 2 / math.sqrt(T)
    assert np.mean(np.abs(np.cov(xs.T) - np.linalg.inv(Γ.toarray()))) < 2.5 / math.sqrt(T)


def test_szigzag_and_subtrace(gpu_pkg, Γ):
    """@testset "SZigZag" + "subtrace" (test/maintest.jl:36-59)."""
    pkg, rng = gpu_pkg, np.random.default_rng(3)
    x0, θ0 = rng.random(d), rng.choice([-1.0, -0.5, 0.5, 1.0], d)
    c = 2.0 * pkg.problems.column_norms(Γ)
    T = 1000.0
    trace, _, acc, _ = pkg.spdmp(pkg.GaussianTarget(Γ), 0.0, x0, θ0, T, c, pkg.ZigZag(0.9 * Γ, np.zeros(d)))
    ts, xs = pkg.trace.discretize(trace, 0.5)
    J = np.arange(0, d, 2)
    ts2, xs2 = pkg.trace.discretize(pkg.trace.subtrace(trace, J), 0.5)
    n = len(ts2)
    assert np.allclose(ts2, ts[:n]) and np.allclose(xs2, xs[:n][:, J])
    assert np.mean(np.abs(xs.mean(0))) < 2 / math.sqrt(T)
    assert np.mean(np.abs(np.cov(xs.T) - np.linalg.inv(Γ.toarray()))) < 2.5 / math.sqrt(T)


def test_factboomerang_pdmp_and_spdmp(gpu_pkg, Γ):
    """@testset "FactBoomerang" / "SFactBoomerang" (test/maintest.jl:91-137): Z = FactBoomerang(0.85Γ / 1.2Γ, 0, 0.3), T = 3000,
    θ0 = sqrt(Diagonal(Z.Γ)) \\ randn(d); pdmp (G = All()) and spdmp."""
    pkg, rng = gpu_pkg, np.random.default_rng(4)
    T = 3000.0
    c = pkg.problems.column_norms(Γ)
    for scale, run, x0, thr in ((0.85, pkg.pdmp, 0.2 * rng.random(d), 4.5), (1.2, pkg.spdmp, rng.random(d), 4.0)):
        Z = pkg.FactBoomerang(sp.csc_matrix(scale * Γ), np.zeros(d), 0.3)
        θ0 = rng.standard_normal(d) / np.sqrt(Z.Γ.diagonal())
        trace, _, acc, _ = run(pkg.GaussianTarget(Γ), 0.0, x0, θ0, T, c, Z)
        m, xs = _stats(pkg, trace, 0.5)
        assert m < 2 / math.sqrt(T)
        assert np.mean(np.abs(np.cov(xs.T) - np.linalg.inv(Γ.toarray()))) < thr / math.sqrt(T)


def test_boomerang_and_bouncy_particle(gpu_pkg, Γ):
    """@testset "Boomerang" (c = 16, λ = 0.5, T = 3000, dt = 0.1) and "Bouncy Particle Sampler" (c = 1.1 -> a valid bound here,
    λ = 0.5, T = 300) (test/maintest.jl:139-172).  Both flows are constructed as the reference constructs them, i.e. WITH the mass
    factor L = cholesky(Symmetric(Γ0)).L (src/types.jl:43,66); the Boomerang covariance check is @test_broken in the reference for
    that flow and is asserted here for the identity-mass Boomerang(I, 0, 0.5), which does preserve N(0, Γ⁻¹)-corrected dynamics."""
    pkg, rng = gpu_pkg, np.random.default_rng(5)
    T = 3000.0
    B = pkg.Boomerang(Γ, np.zeros(d), 0.5)
    assert B.L is not None
    trace, _, acc, _ = pkg.pdmp(pkg.GaussianTarget(Γ), 0.0, rng.standard_normal(d), rng.standard_normal(d), T, 16.0, B)
    m, xs = _stats(pkg, trace, 0.1)
    assert m < 2 / math.sqrt(T)
    B = pkg.Boomerang(sp.identity(d, format="csc"), np.zeros(d), 0.5)
    trace, _, acc, _ = pkg.pdmp(pkg.GaussianTarget(Γ), 0.0, rng.standard_normal(d), rng.standard_normal(d), T, 16.0, B)
    m, xs = _stats(pkg, trace, 0.1)
    assert m < 2 / math.sqrt(T)
    assert np.mean(np.abs(np.cov(xs.T) - np.linalg.inv(Γ.toarray()))) < 2.5 / math.sqrt(T)  # @test_broken in the reference
    T = 300.0
    B = pkg.BouncyParticle(Γ, np.zeros(d), 0.5)
    assert B.L is not None
    trace, _, acc, _ = pkg.pdmp(None, 0.0, rng.standard_normal(d), rng.standard_normal(d), T, 1.1, B)
    m, xs = _stats(pkg, trace, 0.1)
    assert m < 2 / math.sqrt(T)
    assert np.mean(np.abs(np.cov(xs.T) - np.linalg.inv(Γ.toarray()))) < 2 / math.sqrt(T)


def test_sticky_zigzag_1d(gpu_pkg):
    """@testset "Sticky ZigZag 1d" (test/sticky.jl:7-36): N(0.9, 0.5) slab, κ = 1.5, c = 20, T = 2000; P(X ≠ 0) = w etc."""
    pkg = gpu_pkg
    σ, μ, κ, T = math.sqrt(0.5), 0.9, 1.5, 2000.0
    Gt = sp.csc_matrix(np.array([[1.0 / σ ** 2]]))  # ∇ϕ(x, i) = (x − μ)/σ²
    Z = pkg.ZigZag(sp.csc_matrix(np.array([[1.0]])), np.zeros(1))
    trace, _, acc, _ = pkg.sspdmp(pkg.GaussianTarget(Gt, np.array([μ])), 0.0, np.array([1.0]), np.array([0.8]), T, np.array([20.0]),
                                  Z, np.array([κ]))
    ts, xs = pkg.trace.discretize(trace, 0.2)
    xs = xs[:, 0]
    w = math.sqrt(2 * math.pi) * σ / (math.sqrt(2 * math.pi) * σ + math.exp(-0.5 * μ ** 2 / σ ** 2) / κ)
    assert abs(np.mean(xs != 0) - w) < 2.5 / math.sqrt(T)
    assert abs(np.mean(xs) - w * μ) < 5.0 / math.sqrt(T)
    assert abs(np.mean(xs ** 2) - w * (σ ** 2 + μ ** 2)) < 5.0 / math.sqrt(T)


def test_sticky_szigzag(gpu_pkg, Γ):
    """@testset "Sticky SZigZag" (test/sticky.jl:39-65): κ = 1000 ("dont stop, actually"), Z = ZigZag(0.9Γ, 0), T = 1000."""
    pkg, rng = gpu_pkg, np.random.default_rng(6)
    x0, θ0 = rng.random(d), rng.choice([-1.0, -0.5, 0.5, 1.0], d)
    c = 2.0 * pkg.problems.column_norms(Γ)
    T = 1000.0
    trace, _, acc, _ = pkg.sspdmp(pkg.GaussianTarget(Γ), 0.0, x0, θ0, T, c, pkg.ZigZag(0.9 * Γ, np.zeros(d)), np.full(d, 1000.0))
    m, xs = _stats(pkg, trace, 0.5)
    assert m < 2 / math.sqrt(T)
    assert np.mean(np.abs(np.cov(xs.T) - np.linalg.inv(Γ.toarray()))) < 2.5 / math.sqrt(T)


def test_1d_samplers(gpu_pkg):
    """test/test1d.jl:1-66 on the device: ZigZag1d with ∇ϕhat, Boomerang1d(1.0) with ∇ϕ, Boomerang1d(1.1, 1.2, 0.5) with ∇ϕhat; T = 8000,
    the three chains in one call each (the reference's thresholds; the streams are named as in tests/test_oracle_1d.py)."""
    pkg = gpu_pkg
    μ, σ2, T = math.pi / 3, 1.3, 8000.0
    noisy, exact = pkg.GaussianTarget1d(μ, σ2, 0.1), pkg.GaussianTarget1d(μ, σ2, 0.0)

    def envelopes(out, flow, k):
        n = len(out)
        assert T / 10 < n < T * 10
        ts, xs = pkg.trace.discretize_1d(out, flow, 0.01)
        d_ = np.diff(ts[:len(ts) // 3])
        assert abs(d_.min() - d_.max()) < 1e-10
        assert abs(xs.mean() - μ) < k / math.sqrt(n)
        assert abs(xs.var(ddof=1) - σ2) < (2.5 if k == 2 else k) / math.sqrt(n)

    out1, _ = pkg.pdmp(noisy, 1.01, -1.5, T, 10.0, pkg.ZigZag1d(), seed=3)
    est = np.sum((out1["x"][:-1] + out1["x"][1:]) / 2 * np.diff(out1["t"])) / T
    assert abs(est - μ) < 2 / math.sqrt(len(out1))
    envelopes(out1, pkg.ZigZag1d(), 2)
    B = pkg.Boomerang1d(1.0)
    out2, _ = pkg.pdmp(exact, 1.41, 0.5, T, 1.6, B, seed=4)
    envelopes(out2, B, 5)
    B = pkg.Boomerang1d(1.1, 1.2, 0.5)
    out3, _ = pkg.pdmp(noisy, 1.41, 0.5, T, 10.0, B, seed=3)
    envelopes(out3, B, 5)
