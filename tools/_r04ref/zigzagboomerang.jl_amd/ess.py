"""Effective sample size of an ensemble of PDMP chains from batch means of the exact path integrals.

The reference has no ESS estimator of its own (only MCMCChains in turing/lr.jl:147-148, third party); the integrand is the one
of `mean(trace)` (src/trace.jl:182-200): Y = (1/b) ∫ x_i(t) dt over a batch of length b, exact for the piecewise-linear path.

For ONE chain observed over B consecutive batches after burn-in, b·Var(Y) -> σ²_asym,i as b grows beyond the integrated
autocorrelation time, where σ²_asym is the asymptotic variance of the time average: Var((1/T)∫x_i) ≈ σ²_asym/T.  The effective
sample size of a path of length T is T·Var_π,i/σ²_asym,i.  With N independent chains the within-chain sums of squares are pooled
(N(B−1) degrees of freedom); at stationarity the spread of the N whole-run chain means gives a second, independent estimate of the
same σ²_asym with no batch-length bias.  Pooling (chain, batch) pairs ACROSS chains -- what round 1 did -- measures Var_π instead.
"""
import math

import numpy as np


def batch_means_ess(sum_y, sum_y2, sum_m, sum_m2, nchains, nbatches, batch_len, var_pi):
    """Inputs: the four [d] device sums of Ensemble.ess_end (ΣY, ΣY² over chains x batches; ΣM, ΣM² over chains), N, B, b and the
    stationary variances Var_π,i.  Returns a dict of [d] arrays:
        sigma2_within   b·(ΣY² − B·ΣM²)/(N(B−1))          pooled within-chain batch-means estimate of σ²_asym
        sigma2_between  B·b·(ΣM² − (ΣM)²/N)/(N−1)         from the spread of the chain means (needs stationarity at T0)
        ess             N·B·b·Var_π/σ²_within             effective samples in the WHOLE ensemble run
        ess_per_time    Var_π/σ²_within                   per chain and unit of process time
        mean            ΣM/N
    """
    N, B, b = int(nchains), int(nbatches), float(batch_len)
    if B < 2 or N < 1:
        raise ValueError("need at least 2 batches")
    sum_y, sum_y2, sum_m, sum_m2 = (np.asarray(a, dtype=np.float64) for a in (sum_y, sum_y2, sum_m, sum_m2))
    var_pi = np.asarray(var_pi, dtype=np.float64)
    s_within = np.maximum(sum_y2 - B * sum_m2, 0.0)
    sig_w = b * s_within / (N * (B - 1))
    sig_b = (B * b) * (sum_m2 - sum_m * sum_m / N) / (N - 1) if N > 1 else np.full_like(sig_w, np.nan)
    tiny = np.finfo(np.float64).tiny
    return dict(sigma2_within=sig_w, sigma2_between=sig_b, ess=N * B * b * var_pi / np.maximum(sig_w, tiny),
                ess_per_time=var_pi / np.maximum(sig_w, tiny), mean=sum_m / N)


def multiscale_ess(J, batch_len, var_pi, mean=0.0):
    """Batch-means estimates of σ²_asym at EVERY dyadic batch length the run allows, from per-chain path integrals.

    J: [B+1, N, P] -- J_i(T0 + k·b) of N chains at P probe coordinates (Ensemble.path_integrals after every batch), b = batch_len,
    var_pi: [P] stationary variances, mean: the KNOWN stationary mean (0 for the centred GMRF; chains started in stationarity).
    For s = b·2^j (j = 0 .. log2 B) the batches are merged 2^j at a time and
        σ²(s) = s · mean over chains and merged batches of (Y_s − mean)²          (no centring on estimated means: unbiased at any s)
    which grows with s towards σ²_asym as s passes the integrated autocorrelation times present in x_i; at s = B·b it is the
    between-chain estimate.  ESS_i(s) = N·B·b·Var_π,i/σ²_i(s) is therefore an UPPER bound that tightens with s: the figure to quote
    is the one at the largest s, together with how much the last doubling still moved it (`last_doubling`: ≤ a few % = plateau).
    The bias of batch means is −Γ/s to first order (Γ = 2Σ_k k·γ_k), so σ²_x = 2σ²(2s) − σ²(s) (Richardson) at the two largest
    lengths removes it: `sigma2_extrapolated`, `ess_extrapolated` -- the conservative figure to headline, validated on the closed-form
    1-d target with batches of ONE autocorrelation time (tests/test_gpu_ess.py).  Standard error of each σ² is ≈ sqrt(2/(N·B/2^j))
    relative.  Returns dict(scales [S], sigma2 [S x P], ess [S x P], last_doubling [P], sigma2_extrapolated [P], ess_extrapolated [P])."""
    J = np.asarray(J, dtype=np.float64)
    Bp1, N, P = J.shape
    B = Bp1 - 1
    if B < 1:
        raise ValueError("need at least one batch")
    var_pi = np.asarray(var_pi, dtype=np.float64)
    scales, sig = [], []
    m = 1
    while B % m == 0 and m <= B:
        Y = (J[m::m] - J[:-m:m]) / (m * batch_len) - mean  # [B/m, N, P]
        scales.append(m * batch_len)
        sig.append(m * batch_len * np.mean(Y * Y, axis=(0, 1)))
        m *= 2
    sig = np.array(sig)
    tiny = np.finfo(np.float64).tiny
    ess = N * B * batch_len * var_pi[None, :] / np.maximum(sig, tiny)
    last = sig[-1] / np.maximum(sig[-2], tiny) - 1.0 if len(sig) > 1 else np.full(P, np.nan)
    sig_x = np.maximum(2.0 * sig[-1] - sig[-2], sig[-1]) if len(sig) > 1 else sig[-1]
    return dict(scales=np.array(scales), sigma2=sig, ess=ess, last_doubling=last, sigma2_extrapolated=sig_x,
                ess_extrapolated=N * B * batch_len * var_pi / np.maximum(sig_x, tiny))


# 1-d ZigZag with unit speed on N(0, s²), canonical rate (θx/s²)⁺, no refreshment: solving the Poisson equation −Lφ = x of the
# generator L g = θ g' + (θx/s²)⁺(g(x,−θ) − g(x,θ)) gives φ(x,+) − φ(x,−) = 2s², (φ(x,+) + φ(x,−))' = 2|x|, hence
# σ²_asym = 2⟨φ, x⟩ = E|X|³ = 2·sqrt(2/π)·s³  (Bierkens & Duncan 2017, Example: Gaussian target): the known-answer test of the
# estimator (tests/test_gpu_ess.py).
def zigzag1d_gaussian_sigma2_asym(s=1.0):
    return 2.0 * math.sqrt(2.0 / math.pi) * s ** 3
