"""The optional neighbourhood argument G of spdmp(∇ϕ, t0, x0, θ0, T, c, G, F, ...) and sspdmp(∇ϕ, t0, x0, θ0, T, c, G, F, κ, ...)
(src/sfact.jl:162,171-179; src/ss_fact.jl:159,167-172) on the device (-m gpu, pdmp_ensemble_set_neighbourhood): G[i] ⊋ G1[i] is moved before
every gradient, G1[i] alone is re-bounded (one draw per member of G1), G2[i] = two-hop(G1) \\ G[i] -- bit for bit against the oracle."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle_lib as O

pytestmark = pytest.mark.gpu


def block_diagonal_part(G, k):
    coo = sp.coo_matrix(G)
    keep = (coo.row // k) == (coo.col // k)
    B = sp.csc_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=G.shape)
    B.sort_indices()
    return B


def same(tr, fs, num, acc, cout, r):
    t, x, th = fs
    assert len(tr.events) == len(r["events"]) > 50
    for f in ("i", "t", "x", "theta"):
        assert np.array_equal(tr.events[f], r["events"][f]), f
    assert int(num) == r["num"] and np.array_equal(np.asarray(acc), r["acc"] if np.ndim(acc) else r["nacc"])
    assert np.array_equal(x, r["x"]) and np.array_equal(th, r["theta"]) and np.array_equal(t, r["t"])
    if cout is not None:
        assert np.array_equal(cout, r["c"])


@pytest.mark.parametrize("n,K", [(12, 4), (20, 5)])
def test_spdmp_with_a_larger_neighbourhood(gpu_pkg, n, K):
    """test/testparallel.jl's situation without the threads: the bound's Γ is the target's without the couplings between chunks (G1 sparser),
    G = the target's pattern, c starts too small and adapts."""
    pkg = gpu_pkg
    G = pkg.problems.gmrf_precision(n)
    d = n * n
    Gb = block_diagonal_part(G, d // K)
    assert Gb.nnz < G.nnz
    rng = np.random.default_rng(n)
    x0, th0 = rng.standard_normal(d), rng.choice([-1.0, 1.0], d)
    c = 0.5 * pkg.problems.column_norms(G)
    Z = pkg.ZigZag(Gb, np.zeros(d))
    tr, fs, (acc, num), cout = pkg.spdmp(pkg.GaussianTarget(G), 0.0, x0, th0, 15.0, c, G, Z, seed=71, adapt=True, factor=1.7)
    r = O.spdmp_zigzag(Gb, None, G, x0, th0, c, 15.0, seed=71, adapt=True, factor=1.7, G=G)
    assert r["status"] == 0 and r["c"].max() > c.max()
    same(tr, fs, num, acc, cout, r)
    # it IS a different computation from Matched(): the clocks of G[i] \\ G1[i] move at every proposal
    r0 = O.spdmp_zigzag(Gb, None, Gb, x0, th0, c, 15.0, seed=71, adapt=True, factor=1.7)
    assert not np.array_equal(r0["t"], r["t"])
    # the same through the keyword, with a target mean and an ensemble of two chains
    mu = 0.2 * rng.standard_normal(d)
    X0, TH0 = rng.standard_normal((2, d)), rng.choice([-1.0, 1.0], (2, d))
    trs, (t, x, th), (acc, num), cout = pkg.spdmp(pkg.GaussianTarget(G, mu), 0.0, X0, TH0, 6.0, 2.0 * c, pkg.ZigZag(Gb, mu), seed=5, G=G, adapt=True)
    for k in range(2):
        r = O.spdmp_zigzag(Gb, mu, G, X0[k], TH0[k], 2.0 * c, 6.0, seed=5 + k, adapt=True, target_mu=mu, G=G)
        same(trs[k], (t[k], x[k], th[k]), num[k], acc[k], cout[k], r)


def test_matched_neighbourhood_given_explicitly_and_the_assert(gpu_pkg):
    pkg = gpu_pkg
    G = pkg.problems.maintest_precision(8)
    d = 8
    rng = np.random.default_rng(3)
    x0, th0 = rng.standard_normal(d), rng.choice([-1.0, 1.0], d)
    c = 2.0 * pkg.problems.column_norms(G)
    Z = pkg.ZigZag(G, np.zeros(d))
    a = pkg.spdmp(pkg.GaussianTarget(G), 0.0, x0, th0, 40.0, c, Z, seed=9)
    b = pkg.spdmp(pkg.GaussianTarget(G), 0.0, x0, th0, 40.0, c, [G[:, i].indices for i in range(d)], Z, seed=9)  # G = G1 as index lists
    assert np.array_equal(a[0].events, b[0].events) and np.array_equal(a[1][1], b[1][1])
    full = sp.csc_matrix(np.ones((d, d)))
    r = O.spdmp_zigzag(G, None, G, x0, th0, c, 40.0, seed=9, G=full)
    tr, fs, (acc, num), _ = pkg.spdmp(pkg.GaussianTarget(G), 0.0, x0, th0, 40.0, c, full, Z, seed=9)  # every coordinate moved at every proposal
    same(tr, fs, num, acc, None, r)
    # G ⊉ G1: the reference's @assert (src/sfact.jl:177)
    small = sp.identity(d, format="csc")
    with pytest.raises(AssertionError, match="G1"):
        pkg.spdmp(pkg.GaussianTarget(G), 0.0, x0, th0, 1.0, c, small, Z, seed=9)
    assert O.spdmp_zigzag(G, None, G, x0, th0, c, 1.0, seed=9, G=small)["status"] == 4
    # a target that reaches outside G is still refused (src/sfact.jl:116: ∇ϕ may read x[j], j ∈ G[i], only)
    with pytest.raises(pkg._lib.PdmpError) as ei:
        pkg.spdmp(pkg.GaussianTarget(full), 0.0, x0, th0, 1.0, c, G, Z, seed=9)
    assert ei.value.code == pkg._lib.PDMP_ERR_UNSUPPORTED


def test_sspdmp_with_a_larger_neighbourhood(gpu_pkg):
    pkg = gpu_pkg
    n = 10
    G = pkg.problems.gmrf_precision(n, 0.5)
    d = n * n
    Gb = block_diagonal_part(G, d // 2)
    rng = np.random.default_rng(11)
    x0, th0 = rng.standard_normal(d), rng.choice([-1.0, 1.0], d)
    c = 1.5 * pkg.problems.column_norms(G)
    kappa = np.full(d, 0.7)
    tr, fs, (acc, num), cout = pkg.sspdmp(pkg.GaussianTarget(G), 0.0, x0, th0, 12.0, c, G, pkg.ZigZag(Gb, np.zeros(d)), kappa, seed=21, adapt=True)
    r = O.sspdmp_zigzag(Gb, None, G, x0, th0, c, kappa, 12.0, seed=21, adapt=True, G=G)
    assert r["status"] == 0
    assert len(tr.events) == len(r["events"]) > 100
    for f in ("i", "t", "x", "theta"):
        assert np.array_equal(tr.events[f], r["events"][f]), f
    assert int(num) == r["num"] and int(acc) == r["nacc"] and np.array_equal(cout, r["c"])
    assert np.array_equal(fs[1], r["x"]) and np.array_equal(fs[0], r["t"])
    assert np.sum(tr.events["theta"] == 0.0) > 10  # freezes happened
