"""Randomised stress of the bouncy particle kernels (-m gpu; pdmp_inner!, src/not_fact_samplers.jl:52-147): random dimensions across the register /
AGPR / scratch instantiations, Γ = I, diagonal or sparse with the reference's mass factor cholesky(Γ).L or the identity, a mean, refresh rates, ρ,
adapt, LocalBound, subsample -- bit for bit the oracle (events t, x, θ; counters; final state; c).  Round 6, seeds fixed."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle_lib as O
from test_gpu_bps_parity import check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", range(16))
def test_random_bps_options(gpu_pkg, case):
    pkg = gpu_pkg
    rng = np.random.default_rng(9500 + case)
    kind = int(rng.integers(0, 3))
    if kind == 0:
        d = int(rng.choice([1, 3, 17, 64, 65, 200, 1024, 1025]))
        G = sp.identity(d, format="csc")
    elif kind == 1:
        n = int(rng.integers(3, 12))
        G = pkg.problems.gmrf_precision(n, eps=float(rng.uniform(0.1, 1.0)))
        d = n * n
    else:
        d = int(rng.integers(4, 40))
        G = pkg.problems.maintest_precision(d) if d == 8 else sp.csc_matrix(sp.diags(rng.uniform(0.5, 2.0, d)))
    G = sp.csc_matrix(G)
    G.sort_indices()
    mu = 0.5 * rng.standard_normal(d) if rng.integers(0, 2) else None
    nch = 2
    x0, th0 = rng.standard_normal((nch, d)), rng.standard_normal((nch, d))
    lam = float(rng.choice([0.3, 1.0, 2.5]))  # (BouncyParticle needs a strictly positive refreshment rate: the engine refuses 0, as the reference's sampler would never mix)
    rho = float(rng.choice([0.0, 0.0, 0.4])) if lam > 0 else 0.0
    adapt = bool(rng.integers(0, 2))
    local_bound = bool(rng.integers(0, 4) == 0)
    subsample = bool(rng.integers(0, 4) == 0) and not local_bound
    c = float(rng.uniform(0.5, 2.0)) if adapt or local_bound else float(rng.uniform(3.0, 6.0)) * float(np.sqrt(d))
    T = float(rng.uniform(5.0, 25.0)) * min(1.0, 200.0 / d)
    L = "chol" if rng.integers(0, 2) else sp.identity(d, format="csc")
    try:
        check(pkg, G, mu, x0, th0, c, T, lam, rho=rho, adapt=adapt, seed=9600 + 10 * case, L=L, local_bound=local_bound, subsample=subsample)
    except AssertionError as e:
        if "status" in str(e) or "assert r[" in str(e):
            pytest.skip("bound too small for this draw without adapt")
        raise
