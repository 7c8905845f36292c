"""Problem set-ups of the reference's scripts, as inputs for the engine (no sampler logic here)."""
import numpy as np
import scipy.sparse as sp


def gridlaplacian(m, n):
    """Graph Laplacian of an m x n lattice, vec'd column-major -- scripts/gridlaplace.jl:4-21."""
    idx = np.arange(m * n).reshape(n, m).T  # idx[i, j] = linear index of (i, j), column-major like LinearIndices
    a_h = idx[:-1, :].ravel()
    b_h = idx[1:, :].ravel()   # (i+1, j)
    a_v = idx[:, :-1].ravel()
    b_v = idx[:, 1:].ravel()   # (i, j+1)
    a = np.concatenate([a_h, a_v])
    b = np.concatenate([b_h, b_v])
    N = m * n
    W = sp.coo_matrix((np.ones(a.size), (a, b)), shape=(N, N))
    W = W + W.T
    deg = np.asarray(W.sum(axis=1)).ravel()
    L = sp.diags(deg) - W
    L = sp.csc_matrix(L)
    L.sort_indices()
    return L


def gmrf_precision(n, eps=0.01):
    """Γ = 0.01 I + gridlaplacian(n, n) -- scripts/gaussianrandomfield.jl:15."""
    G = sp.csc_matrix(eps * sp.identity(n * n, format="csc") + gridlaplacian(n, n))
    G.sort_indices()
    return G


def gmrf_stationary_sample(n, nchains, rng, eps=0.01):
    """x ~ N(0, Γ⁻¹) exactly for Γ = eps I + gridlaplacian(n, n): the free-boundary path Laplacian is diagonalised by the DCT-II
    (eigenvalues 2 − 2cos(πk/n)), the grid Laplacian by its tensor square, so x = idctn(z / sqrt(eps + λ_k + λ_l)) with z ~ N(0, I)
    and the orthonormal transform.  Returns [nchains x n²] in the column-major numbering of gridlaplacian.  Used to start ESS runs in
    stationarity (a ZigZag's stationary velocities are uniform on {±1}ⁿ, independent of x)."""
    from scipy.fft import idctn
    lam1 = 2.0 - 2.0 * np.cos(np.pi * np.arange(n) / n)
    lam = eps + lam1[:, None] + lam1[None, :]
    z = rng.standard_normal((nchains, n, n))
    x = idctn(z / np.sqrt(lam)[None], axes=(1, 2), norm="ortho")
    return np.ascontiguousarray(x.transpose(0, 2, 1).reshape(nchains, n * n))  # [row, col] -> index row + n col


def gmrf_marginal_variances(n, eps=0.01):
    """diag(Γ⁻¹) of the same matrix in closed form: Σ_kl φ_kl(i)² / (eps + λ_k + λ_l), [n²] in the same numbering."""
    lam1 = 2.0 - 2.0 * np.cos(np.pi * np.arange(n) / n)
    j = np.arange(n)
    V = np.cos(np.pi * np.outer(np.arange(n), j + 0.5) / n) * np.sqrt(2.0 / n)  # V[k, j], orthonormal DCT-II rows
    V[0] /= np.sqrt(2.0)
    W = V * V
    inv = 1.0 / (eps + lam1[:, None] + lam1[None, :])
    var = W.T @ inv @ W  # var[row, col]
    return np.ascontiguousarray(var.T.reshape(n * n))


def lattice3d_precision(n, eps=0.01):
    """Γ = eps I + graph Laplacian of the n x n x n 7-point lattice (index i = a + n b + n² c): scripts/gridlaplace.jl's construction one
    dimension up -- |G1| = 7 inside, |S| = 25.  A graph whose neighbours are NOT i ± 1, i ± n of the 2-d numbering (config C3G)."""
    idx = np.arange(n ** 3).reshape(n, n, n)  # idx[c, b, a]
    pa = [(idx[:, :, :-1].ravel(), idx[:, :, 1:].ravel()), (idx[:, :-1, :].ravel(), idx[:, 1:, :].ravel()),
          (idx[:-1, :, :].ravel(), idx[1:, :, :].ravel())]
    a = np.concatenate([q[0] for q in pa])
    b = np.concatenate([q[1] for q in pa])
    N = n ** 3
    W = sp.coo_matrix((np.ones(a.size), (a, b)), shape=(N, N))
    W = W + W.T
    deg = np.asarray(W.sum(axis=1)).ravel()
    G = sp.csc_matrix(eps * sp.identity(N, format="csc") + sp.diags(deg) - W)
    G.sort_indices()
    return G


def random_sparse_precision(d, nnz_per_col=6, seed=7, eps=0.05):
    """A random symmetric sparse precision with at most `nnz_per_col` entries per column, no structure in the numbering: the pattern of
    test/maintest.jl:6-8 (`sprandn`) scaled to d ~ 16384 and made diagonally dominant so that it is positive definite whatever the pattern.
    Built as a random graph of maximum degree nnz_per_col - 1 (random perfect matchings laid over each other, duplicates dropped), weights
    w_ij = -|N(0, 1)|, Γ_ii = eps + Σ_j |w_ij| (config C3G)."""
    rng = np.random.default_rng(seed)
    rows, cols = [], []
    for _ in range(nnz_per_col - 1):
        perm = rng.permutation(d)
        h = d // 2
        rows.append(perm[:h])
        cols.append(perm[h:2 * h])
    a = np.concatenate(rows)
    b = np.concatenate(cols)
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    key = np.unique(lo.astype(np.int64) * d + hi)
    lo, hi = key // d, key % d
    w = -np.abs(rng.standard_normal(lo.size)) - 0.1
    W = sp.coo_matrix((w, (lo, hi)), shape=(d, d))
    W = sp.csc_matrix(W + W.T)
    diag = eps - np.asarray(W.sum(axis=0)).ravel()
    G = sp.csc_matrix(W + sp.diags(diag))
    G.sort_indices()
    return G


def column_norms(G):
    """c[i] = norm(Γ[:, i], 2) -- scripts/gaussianrandomfield.jl:33."""
    G = sp.csc_matrix(G)
    return np.sqrt(np.asarray(G.multiply(G).sum(axis=0)).ravel())


def maintest_precision(d=8, seed=2):
    """Γ = S S', S = 1.3 I + 0.5 sprandn(d, d, 0.1) -- test/maintest.jl:6-8 (own RNG, same construction)."""
    rng = np.random.default_rng(seed)
    R = sp.random(d, d, density=0.1, random_state=rng, data_rvs=rng.standard_normal, format="csc")
    S = 1.3 * sp.identity(d, format="csc") + 0.5 * R
    G = sp.csc_matrix(S @ S.T)
    G.sort_indices()
    return G


def sparse_design(levels=(20, 20), r=2, m=20, rng=None):
    """Mock design matrix with categorical factors, their pairwise interactions and r continuous regressors --
    scripts/sparsedesign.jl:2-26 (own RNG, same construction).  Returns CSC [n = m p, p]."""
    rng = np.random.default_rng(2) if rng is None else rng
    d = list(levels)
    K = len(d)
    p = sum(d) + (sum(d) ** 2 - sum(v * v for v in d)) // 2 + r
    n = m * p
    D = np.concatenate([[0], np.cumsum(d)])
    rows, cols, vals = [], [], []
    for i in range(n):
        lev = []
        on = []
        for k in range(K):
            lev.append(int(rng.integers(D[k], D[k + 1])))
            on.append(rng.random() < (d[k] - 1) / d[k])
            if on[k]:
                rows.append(i)
                cols.append(lev[k])
                vals.append(1.0)
        j = int(D[-1])
        for k in range(K):
            for k2 in range(k):
                # CartesianIndices((d[k], d[k2])): first index fastest
                if on[k] and on[k2]:
                    c1, c2 = lev[k] - D[k], lev[k2] - D[k2]
                    rows.append(i)
                    cols.append(j + c2 * d[k] + c1)
                    vals.append(0.3)
                j += d[k] * d[k2]
        for _ in range(r):
            rows.append(i)
            cols.append(j)
            vals.append(0.1 * rng.standard_normal())
            j += 1
        assert j == p
    A = sp.csc_matrix((vals, (rows, cols)), shape=(n, p))
    A.sort_indices()
    return A


def logistic_problem(levels=(20, 20), r=2, m=20, gamma0=0.01, seed=2, droptol=1e-2):
    """Sparse logistic regression set-up of scripts/logistic.jl:21-158 (config C4): design A, data y, Newton mode μ,
    Hessian Γ at μ, its sparsified version Γdrop (droptol 1e-2), σ = sqrt(diag(inv Γ)), θ0 = ±σ, c = 0.01."""
    rng = np.random.default_rng(seed)
    A = sparse_design(levels, r, m, rng)
    n, p = A.shape
    xtrue = 5 * rng.standard_normal(p)
    sig = lambda u: 1.0 / (1.0 + np.exp(-u))  # noqa: E731
    y = (rng.random(n) < sig(A @ xtrue)).astype(np.float64)
    ny = 1.0 - y
    x = 0.1 * rng.random(p)
    At = sp.csc_matrix(A.T)
    At.sort_indices()
    for _ in range(30):  # Newton steps towards the mode, :120-125
        u = A @ x
        g = gamma0 * x - A.T @ (y * sig(-u)) + A.T @ (ny * sig(u))
        w = sig(u) * sig(-u)
        H = gamma0 * sp.identity(p, format="csc") + (A.T @ sp.diags(w) @ A)
        x = x - sp.linalg.spsolve(sp.csc_matrix(H), g)
    mu = x
    u = A @ mu
    G = sp.csc_matrix(gamma0 * sp.identity(p, format="csc") + (A.T @ sp.diags(sig(u) * sig(-u)) @ A))
    G = sp.csc_matrix((G + G.T) * 0.5)
    Gd = G.copy()
    Gd.data[np.abs(Gd.data) <= droptol] = 0.0
    Gd.eliminate_zeros()
    Gd.sort_indices()
    sigma = np.sqrt(np.diag(np.linalg.inv(G.toarray())))
    theta0 = rng.choice([-1.0, 1.0], p) * sigma
    return dict(A=A, At=At, y=y, ny=ny, mu=mu, gamma0=gamma0, G=G, Gdrop=Gd, sigma=sigma, theta0=theta0, x0=mu.copy(),
                c=0.01 * np.ones(p), xtrue=xtrue, n=n, p=p)


def example_design_matrix(num_rows=50_000, num_categorical=100, num_continuous=10, rng=None):
    """example_design_matrix(; num_rows) -- scripts/exampledesign.jl:2-13: binary categorical features `rand(num_rows) .> 0.5`
    followed by standard-normal continuous ones; the column counts are arguments here so that the generator scales to config C5's
    10⁴ columns in the script's own 10 : 1 proportion.  Returned as CSC (the zeros of the binary columns are not stored)."""
    rng = np.random.default_rng(2) if rng is None else rng
    cat = (rng.random((num_rows, num_categorical)) > 0.5).astype(np.float64)
    con = rng.standard_normal((num_rows, num_continuous))
    A = sp.csc_matrix(np.hstack([cat, con]))
    A.sort_indices()
    return A


def spike_slab_logistic_problem(p=10_000, num_rows=2000, gamma0=0.25, w=0.5, seed=2):
    """Config C5: Bayesian logistic regression with a spike-and-slab prior, p = 10⁴ coefficients.

    Design: example_design_matrix (scripts/exampledesign.jl:2-13) scaled to p columns (10 : 1 categorical : continuous, as in the
    script's 100 + 10); data as scripts/spikeandslab.jl:36-40: xtrue = randn(p) .* (rand(p) .< 0.2), y ~ Bernoulli(sigmoid(A xtrue));
    slab N(0, 1/γ0) -- so that ∇ϕ is the `γ0*x[i] − fdot_moving(...)` of scripts/logistic.jl:107 / sticky_logistic_sparse.jl:131 --,
    spike = the point mass of the sticky sampler with thaw rate κ = (γ0/√2π)/(1/w − 1) (scripts/sticky/sticky_logistic_sparse.jl:197);
    flow Z = ZigZag(sparse(1.0I, p, p), μ, σ) as scripts/spikeandslab.jl:96,113 (one-coordinate neighbourhoods: ∇ϕmoving moves what
    it reads itself, SelfMoving()), c = ones(p) with adapt = true (:127-129); μ = the slab posterior's mode (Newton steps in the dual
    form, the control-variate point of ∇ϕmoving).  num_rows is the population the gradient subsamples from (the script's 50 000
    rows x 110 columns would be 2.7·10¹⁰ stored entries at 10⁴ columns); the work per proposal depends on k and on the row length
    (≈ 0.55 p), not on it."""
    rng = np.random.default_rng(seed)
    ncon = p // 11
    A = example_design_matrix(num_rows, p - ncon, ncon, rng)
    n = A.shape[0]
    xtrue = rng.standard_normal(p) * (rng.random(p) < 0.2)
    sig = lambda u: 1.0 / (1.0 + np.exp(-u))  # noqa: E731
    y = (rng.random(n) < sig(A @ xtrue)).astype(np.float64)
    ny = 1.0 - y
    At = sp.csc_matrix(A.T)
    At.sort_indices()
    Ad = A.toarray()
    x = np.zeros(p)
    if n > p:
        # the script's own population (scripts/exampledesign.jl:2: 50 000 rows) has more rows than columns: the Newton step in its primal form,
        # (γ0 I + A'WA) δ = g, a p x p system (bench.py --c5-rows 50000: a one-off measurement, minutes of host time)
        for _ in range(12):
            u = Ad @ x
            g = gamma0 * x - Ad.T @ (y * sig(-u)) + Ad.T @ (ny * sig(u))
            wgt = sig(u) * sig(-u)
            H = Ad.T @ (wgt[:, None] * Ad)
            H[np.diag_indices(p)] += gamma0
            step = np.linalg.solve(H, g)
            x = x - step
            if np.linalg.norm(step) <= 1e-12 * (1.0 + np.linalg.norm(x)):
                break
        mu = x
        kappa = np.full(p, (gamma0 / np.sqrt(2 * np.pi)) / (1 / w - 1))
        return dict(A=A, At=At, y=y, ny=ny, mu=mu, gamma0=gamma0, kappa=kappa, G=sp.identity(p, format="csc"), sigma=np.ones(p),
                    c=np.ones(p), xtrue=xtrue, n=n, p=p, w=w)
    for _ in range(12):  # Newton on the slab posterior; (γ0 I + B'B)⁻¹ g = (g − B'(γ0 I + B B')⁻¹ B g)/γ0 with B = W^½ A (n < p)
        u = Ad @ x
        g = gamma0 * x - Ad.T @ (y * sig(-u)) + Ad.T @ (ny * sig(u))
        B = np.sqrt(sig(u) * sig(-u))[:, None] * Ad
        step = (g - B.T @ np.linalg.solve(gamma0 * np.eye(n) + B @ B.T, B @ g)) / gamma0
        x = x - step
        # (populations beyond the tests' 2000 rows -- bench.py --c5-rows: an n x n solve per step -- stop once the step is at rounding level;
        # the tests' problems keep their 12 steps, so every fixture and oracle comparison sees the μ it always saw)
        if num_rows > 2000 and np.linalg.norm(step) <= 1e-12 * (1.0 + np.linalg.norm(x)):
            break
    mu = x
    kappa = np.full(p, (gamma0 / np.sqrt(2 * np.pi)) / (1 / w - 1))
    return dict(A=A, At=At, y=y, ny=ny, mu=mu, gamma0=gamma0, kappa=kappa, G=sp.identity(p, format="csc"), sigma=np.ones(p),
                c=np.ones(p), xtrue=xtrue, n=n, p=p, w=w)
