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
