"""One chain over K wavefronts (pdmp_ensemble_run_partitioned, zz_partitioned_run_kernel) against the oracle's threaded restatement of the
reference's parallel_spdmp (src/parallel.jl:104-253, oracle/pdmp_oracle.c orc_parallel_spdmp) -- bit for bit: the scheme is deterministic
(the workers of a round are data-independent), both sides use Philox stream 16 + chunk for the workers and 15 for the coordinator (-m gpu)."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle_lib as O

pytestmark = pytest.mark.gpu


def chunk_diagonal(G, K):
    """The bounding Γ of test/testparallel.jl:40-47: the target's Γ without the entries that couple two chunks."""
    d = G.shape[0]
    k = d // K
    coo = sp.coo_matrix(G)
    keep = (coo.row // k) == (coo.col // k)
    G2 = sp.csc_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=G.shape)
    G2.sort_indices()
    return G2


def run_both(pkg, G, K, T, delta, c, seed, x0, th0, adapt=False, factor=1.8):
    G2 = chunk_diagonal(G, K)
    d = G.shape[0]
    r = O.parallel_spdmp(G2, None, G, x0, th0, c, T, K, delta, seed=seed, adapt=adapt, factor=factor)
    assert r["status"] == 0
    cc = np.array(c, dtype=np.float64)
    tr, (t, x, th), (acc, num) = pkg.parallel_spdmp(pkg.Partition(K, d), pkg.GaussianTarget(G), 0.0, x0, th0, T, cc, G,
                                                    pkg.ZigZag(G2, np.zeros(d)), Δ=delta, seed=seed, adapt=adapt, factor=factor)
    return r, tr, t, x, th, acc, num, cc


def assert_identical(r, tr, t, x, th, acc, num):
    ev, oe = tr.events, r["events"]
    assert len(ev) == len(oe) and len(oe) > 0
    assert np.array_equal(ev["i"], oe["i"]) and np.array_equal(ev["t"], oe["t"])
    assert np.array_equal(ev["x"], oe["x"]) and np.array_equal(ev["theta"], oe["theta"])
    assert np.array_equal(t, r["t"]) and np.array_equal(x, r["x"]) and np.array_equal(th, r["theta"])
    assert int(acc) == r["nacc"] and int(num) == r["num"]


def test_reference_test_problem_two_chunks(gpu_pkg):
    """test/testparallel.jl:22-73: d = 20 tridiagonal Γ, 2 chunks, c = 5‖Γ[:, i]‖, Δ = 0.05."""
    pkg = gpu_pkg
    d, K, T, delta = 20, 2, 200.0, 0.05
    G = sp.diags([np.ones(d), -0.4 * np.ones(d - 1), -0.4 * np.ones(d - 1)], [0, 1, -1], format="csc")
    rng = np.random.default_rng(1)
    x0 = 0.1 * rng.standard_normal(d)
    th0 = rng.choice([-1.0, 1.0], d)
    c = 5 * pkg.problems.column_norms(G)
    r, tr, t, x, th, acc, num, _ = run_both(pkg, G, K, T, delta, c, 7, x0, th0)
    assert len(r["events"]) > 1000 and r["rounds"] > 10
    assert_identical(r, tr, t, x, th, acc, num)
    # the reference's envelope on the same run (test/testparallel.jl:66-72, scaled to this T)
    assert np.mean(np.abs(pkg.trace.mean(tr))) < 4 / np.sqrt(T)


@pytest.mark.parametrize("n,K,delta", [(16, 4, 0.1), (32, 8, 0.05), (32, 16, 0.2)])
def test_lattice_chunks_of_columns(gpu_pkg, n, K, delta):
    """n x n grid-Laplace GMRF, chunks = groups of lattice columns: the coordinates next to a chunk border are the coordinator's."""
    pkg = gpu_pkg
    G = pkg.problems.gmrf_precision(n)
    d = n * n
    rng = np.random.default_rng(n + K)
    x0 = rng.standard_normal(d)
    th0 = rng.choice([-1.0, 1.0], d)
    c = 2.0 * pkg.problems.column_norms(G)
    r, tr, t, x, th, acc, num, _ = run_both(pkg, G, K, 6.0, delta, c, 900 + n, x0, th0)
    assert len(r["events"]) > 1000 and r["rounds"] > 20
    assert_identical(r, tr, t, x, th, acc, num)


def test_adapt_raises_the_bounds_like_the_oracle(gpu_pkg):
    pkg = gpu_pkg
    n, K = 16, 4
    G = pkg.problems.gmrf_precision(n)
    d = n * n
    rng = np.random.default_rng(5)
    x0 = 3.0 * rng.standard_normal(d)
    th0 = rng.choice([-1.0, 1.0], d)
    c = 0.05 * pkg.problems.column_norms(G)  # far too small: adapt! must act
    r, tr, t, x, th, acc, num, cc = run_both(pkg, G, K, 3.0, 0.1, c, 31, x0, th0, adapt=True, factor=1.8)
    assert np.any(r["c"] > c)
    assert_identical(r, tr, t, x, th, acc, num)
    assert np.array_equal(cc, r["c"])  # adapt!(c, i, factor) acted on the caller's vector
    with pytest.raises(RuntimeError, match="too small"):
        run_dev_only(pkg, G, K, c, x0, th0)


def run_dev_only(pkg, G, K, c, x0, th0):
    d = G.shape[0]
    return pkg.parallel_spdmp(pkg.Partition(K, d), pkg.GaussianTarget(G), 0.0, x0, th0, 3.0, c, G, pkg.ZigZag(chunk_diagonal(G, K), np.zeros(d)),
                              Δ=0.1, seed=31)


def test_refusals(gpu_pkg):
    pkg = gpu_pkg
    n, K = 16, 4
    G = pkg.problems.gmrf_precision(n)
    d = n * n
    x0, th0 = np.zeros(d), np.ones(d)
    c = pkg.problems.column_norms(G)
    # "Upper bounds may not depend across chunks." (src/parallel.jl:124-127): the full Γ as the bound
    with pytest.raises(RuntimeError, match="across chunks"):
        pkg.parallel_spdmp(pkg.Partition(K, d), pkg.GaussianTarget(G), 0.0, x0, th0, 1.0, c, G, pkg.ZigZag(G, np.zeros(d)))
    # chunks of unequal size
    with pytest.raises(RuntimeError, match="chunks of equal size"):
        pkg.parallel_spdmp(pkg.Partition(7, d), pkg.GaussianTarget(G), 0.0, x0, th0, 1.0, c, G, pkg.ZigZag(chunk_diagonal(G, 4), np.zeros(d)))
    # a second partitioned run on a used state, and a partitioned run where it does not apply
    ens = pkg.Ensemble(1, d, trace_capacity=4096)
    ens.set_flow(pkg.ZigZag(chunk_diagonal(G, K), np.zeros(d)))
    ens.set_target(pkg.GaussianTarget(chunk_diagonal(G, K)))
    ens.set_state(0.0, x0[None], th0[None], c, np.array([1], dtype=np.uint64))
    with pytest.raises(pkg._lib.PdmpError, match="g1_mask has") as ei:  # a mask made for another pattern: a status, not a stray read
        ens.run_partitioned(0.5, K, 0.1, np.ones(7, dtype=np.uint8))
    assert ei.value.code == pkg._lib.PDMP_ERR_INVALID
    ens.run_partitioned(0.5, K, 0.1)
    with pytest.raises(RuntimeError, match="fresh state"):
        ens.run_partitioned(1.0, K, 0.1)
    ens.close()


def test_c3_geometry_one_chain_on_sixteen_waves(gpu_pkg):
    """The north-star geometry (128 x 128 lattice, d = 16384) as ONE chain on 16 wavefronts, 8 lattice columns per chunk."""
    pkg = gpu_pkg
    G = pkg.problems.gmrf_precision(128)
    d = G.shape[0]
    x0, th0 = O.synthetic_state(0x5EED0000, d)
    c = 2.0 * pkg.problems.column_norms(G)
    r, tr, t, x, th, acc, num, _ = run_both(pkg, G, 16, 0.25, 0.05, c, 0x5EED0000, x0, th0)
    assert len(r["events"]) > 2000
    assert_identical(r, tr, t, x, th, acc, num)


def test_ensemble_of_partitioned_chains(gpu_pkg):
    """Six chains, each a workgroup of 8 wavefronts: chain k equals the oracle run with seed + k."""
    pkg = gpu_pkg
    n, K, nch, T, delta = 32, 8, 6, 3.0, 0.1
    G = pkg.problems.gmrf_precision(n)
    d = n * n
    G2 = chunk_diagonal(G, K)
    rng = np.random.default_rng(77)
    x0 = rng.standard_normal((nch, d))
    th0 = rng.choice([-1.0, 1.0], (nch, d))
    c = 2.0 * pkg.problems.column_norms(G)
    trs, (t, x, th), (acc, num) = pkg.parallel_spdmp(pkg.Partition(K, d), pkg.GaussianTarget(G), 0.0, x0, th0, T, c, G,
                                                     pkg.ZigZag(G2, np.zeros(d)), Δ=delta, seed=5000)
    for k in range(nch):
        r = O.parallel_spdmp(G2, None, G, x0[k], th0[k], c, T, K, delta, seed=5000 + k)
        assert r["status"] == 0 and len(r["events"]) > 1000
        assert_identical(r, trs[k], t[k], x[k], th[k], acc[k], num[k])


def test_committed_golden_vector_on_the_device(gpu_pkg):
    """tests/golden/golden2.npz `parallel_*` (made by make_golden2.py from the oracle): index sequence, counters, first event times and
    the SHA-256 of the time-sorted events and the final state, straight from the device."""
    import hashlib
    import importlib.util
    import os
    pkg = gpu_pkg
    here = os.path.dirname(os.path.abspath(__file__))
    gold = np.load(os.path.join(here, "golden", "golden2.npz"), allow_pickle=False)
    G = pkg.problems.gmrf_precision(16)
    d, K = 256, 4
    G2 = chunk_diagonal(G, K)
    rl = np.random.default_rng(12)
    x0, th0 = rl.standard_normal(d), rl.choice([-1.0, 1.0], d)
    c = 2.0 * pkg.problems.column_norms(G)
    tr, (t, x, th), (acc, num) = pkg.parallel_spdmp(pkg.Partition(K, d), pkg.GaussianTarget(G), 0.0, x0, th0, 6.0, c, G,
                                                    pkg.ZigZag(G2, np.zeros(d)), Δ=0.1, seed=81)
    ev = tr.events
    assert np.array_equal(ev["i"].astype(np.uint16), gold["parallel_idx"]) and np.array_equal(ev["t"][:50], gold["parallel_t_head"])
    assert int(num) == int(gold["parallel_n"][0]) and int(acc) == int(gold["parallel_n"][1])
    h = hashlib.sha256()
    for a in (ev["t"], ev["x"], ev["theta"], x, th, t, c):
        h.update(np.ascontiguousarray(a).tobytes())
    assert h.hexdigest() == str(gold["parallel_hash"][0])
