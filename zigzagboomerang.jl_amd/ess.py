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


# 1-d ZigZag with unit speed on N(0, s²), canonical rate (θx/s²)⁺, no refreshment: solving the Poisson equation −Lφ = x of the
# generator L g = θ g' + (θx/s²)⁺(g(x,−θ) − g(x,θ)) gives φ(x,+) − φ(x,−) = 2s², (φ(x,+) + φ(x,−))' = 2|x|, hence
# σ²_asym = 2⟨φ, x⟩ = E|X|³ = 2·sqrt(2/π)·s³  (Bierkens & Duncan 2017, Example: Gaussian target): the known-answer test of the
# estimator (tests/test_gpu_ess.py).
def zigzag1d_gaussian_sigma2_asym(s=1.0):
    return 2.0 * math.sqrt(2.0 / math.pi) * s ** 3
