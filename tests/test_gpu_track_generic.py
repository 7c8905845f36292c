"""The one-proposal-per-lane tracked kernel OFF the benchmark stencil (-m gpu): zz_local_trackp_kernel<.., LAT = false> on graphs whose
neighbours are not i ± 1, i ± n -- the 7-point 3-d lattice and random symmetric patterns with up to 8 entries per column (`spdmp` takes any
sparse Γ: src/sfact.jl:170-179 builds G1 / G2 from the CSC pattern, test/maintest.jl:6-8 uses sprandn).  Same two bars as
tests/test_gpu_track_parity.py: BIT FOR BIT the oracle's tracked evaluation, and the moving evaluation's index sequence / counters with
floats to 1e-9."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle_lib as O
from test_gpu_track_parity import check_chain, check_chain_bitwise

pytestmark = pytest.mark.gpu


def graphs(pkg, which):
    if which == "lattice3d":
        return pkg.problems.lattice3d_precision(13)              # d = 2197, |G1| <= 7, |S| <= 25
    if which == "random6":
        return pkg.problems.random_sparse_precision(2500, 6, seed=3)   # |G1| <= 6, |S| up to 26
    if which == "random8":
        return pkg.problems.random_sparse_precision(3000, 8, seed=4)   # |G1| <= 8, |S| up to 50: beyond every blob kernel
    if which == "ragged":
        # degrees from 1 (a coordinate coupled to nothing: G1 = {i}) to 8, incl. neighbours inside one key block of 8 and across blocks
        G = sp.lil_matrix(pkg.problems.random_sparse_precision(2304, 8, seed=5))
        for i in range(0, 2304, 97):  # isolate some coordinates
            for j in list(G.rows[i]):
                if j != i:
                    G[i, j] = 0.0
                    G[j, i] = 0.0
        for a in range(0, 2296, 24):  # couple the first two coordinates of some key blocks where both have room
            b = a + 1
            if G[a, b] == 0 and np.count_nonzero(G[a].toarray()) < 8 and np.count_nonzero(G[b].toarray()) < 8:
                G[a, b] = G[b, a] = -0.37
                G[a, a] += 0.37
                G[b, b] += 0.37
        G = sp.csc_matrix(G)
        G.eliminate_zeros()
        G.sort_indices()
        return G
    raise KeyError(which)


@pytest.mark.parametrize("which,T", [("lattice3d", 10.0), ("random6", 6.0), ("random8", 5.0), ("ragged", 5.0)])
def test_generic_graph_tracked_matches_oracle(gpu_pkg, which, T, trackp_form):
    pkg = gpu_pkg
    G = graphs(pkg, which)
    d = G.shape[0]
    k = np.diff(G.indptr)
    assert k.max() <= 8 and abs(G - G.T).max() == 0
    rng = np.random.default_rng(len(which))
    nch = 3
    x0 = rng.standard_normal((nch, d))
    th0 = rng.choice([-1.0, 1.0], (nch, d))
    c = pkg.problems.column_norms(G)
    tr, (t, x, th), (acc, num), _ = pkg.spdmp(pkg.GaussianTarget(G), 0.0, x0, th0, T, c, pkg.ZigZag(G, np.zeros(d)), seed=900, tracked=True)
    for q in range(nch):
        rt = O.spdmp_zigzag(G, None, G, x0[q], th0[q], c, T, seed=900 + q, tracked=True)
        assert rt["status"] == 0 and len(rt["events"]) > 1000
        check_chain_bitwise(tr[q].events, t[q], x[q], th[q], acc[q], num[q], None, rt)
        check_chain(tr[q].events, t[q], x[q], th[q], acc[q], num[q], None, O.spdmp_zigzag(G, None, G, x0[q], th0[q], c, T, seed=900 + q))


def test_generic_graph_slices_refills_start_time_and_violation(gpu_pkg, trackp_form):
    """Slices with PDMP_RUN_STOP_BEFORE, a trace buffer that fills up several times, t0 != 0, and a bound violation (status, where the oracle stops)."""
    pkg = gpu_pkg
    L = pkg._lib
    G = graphs(pkg, "random6")
    d = G.shape[0]
    c = pkg.problems.column_norms(G)
    nch, t0 = 2, 1.5
    with pkg.Ensemble(nch, d, trace_capacity=2500) as ens:
        ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
        ens.set_target(pkg.GaussianTarget(G))
        ens.set_gradient_tracking(True)
        ens.set_state_synthetic(t0, c, 777)
        evs = [[] for _ in range(nch)]
        for Tk, flag in ((2.6, L.RUN_STOP_BEFORE), (4.0, L.RUN_STOP_BEFORE), (5.5, L.RUN_REFERENCE_TAIL)):
            while True:
                ens.run(Tk, flag)
                cnt = ens.counters()
                for q in range(nch):
                    evs[q].append(ens.trace(q, counters=cnt))
                ens.trace_reset()
                if not L.needs_rerun(cnt["status"]):
                    break
        fs = ens.final_state()
        cnt = ens.counters()
    for q in range(nch):
        x0, th0 = O.synthetic_state(777 + q, d)
        ev = np.concatenate(evs[q])
        rt = O.spdmp_zigzag(G, None, G, x0, th0, c, 5.5, seed=777 + q, t0=t0, tracked=True)
        check_chain_bitwise(ev, fs["t"][q], fs["x"][q], fs["theta"][q], fs["acc"][q], cnt["num"][q], None, rt)
        assert ev["t"][-1] >= 5.5 and len(ev) > 5000
    rng = np.random.default_rng(2)
    x0 = 3 * rng.standard_normal((2, d))
    th0 = rng.choice([-1.0, 1.0], (2, d))
    # (bound Γ == target Γ: the affine bound is exact up to c, so only a rounding-level c is violated -- after 145 and 276 proposals here)
    cs = np.full(d, 5e-15)
    with pkg.Ensemble(2, d, trace_capacity=4096) as ens:
        ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
        ens.set_target(pkg.GaussianTarget(G))
        ens.set_gradient_tracking(True)
        ens.set_state(0.0, x0, th0, cs, np.array([3, 4], dtype=np.uint64))
        ens.run(5.0)
        cnt = ens.counters()
        for q in range(2):
            r = O.spdmp_zigzag(G, None, G, x0[q], th0[q], cs, 5.0, seed=3 + q, tracked=True)
            assert r["status"] == O.ORC_BOUND_VIOLATED and cnt["status"][q] == L.CHAIN_BOUND_VIOLATED
            assert int(cnt["num"][q]) == r["num"] and int(cnt["nacc"][q]) == r["nacc"] and r["num"] > 100
            ev = ens.trace(q, counters=cnt)
            assert np.array_equal(ev["i"], r["events"]["i"]) and np.array_equal(ev["t"], r["events"]["t"])


def test_generic_kernel_equals_the_lattice_kernel_on_a_relabelled_lattice(gpu_pkg, trackp_form):
    """The same process under a permutation of the coordinates: the n x n lattice relabelled at random is 'a random graph' to the engine
    (LAT = false); every chain's proposal / accept counts differ from the plain lattice's only through the tie rule -- so compare through the
    oracle: both runs equal their own tracked oracle bit for bit, and the generic one is what a user with an arbitrary numbering gets."""
    pkg = gpu_pkg
    n = 48
    G0 = pkg.problems.gmrf_precision(n)
    d = n * n
    perm = np.random.default_rng(11).permutation(d)
    P = sp.csc_matrix((np.ones(d), (perm, np.arange(d))), shape=(d, d))
    G = sp.csc_matrix(P @ G0 @ P.T)
    G.sort_indices()
    c = pkg.problems.column_norms(G)
    rng = np.random.default_rng(12)
    x0 = rng.standard_normal((2, d))
    th0 = rng.choice([-1.0, 1.0], (2, d))
    tr, (t, x, th), (acc, num), _ = pkg.spdmp(pkg.GaussianTarget(G), 0.0, x0, th0, 6.0, c, pkg.ZigZag(G, np.zeros(d)), seed=51, tracked=True)
    for q in range(2):
        rt = O.spdmp_zigzag(G, None, G, x0[q], th0[q], c, 6.0, seed=51 + q, tracked=True)
        check_chain_bitwise(tr[q].events, t[q], x[q], th[q], acc[q], num[q], None, rt)


def test_generic_graph_full_width(gpu_pkg):
    """Config C3G at its width: d = 15625 (25^3 lattice) and d = 16384 (random, <= 6 per column), 4096 chains to T = 0.5: all chains healthy, first and
    last chain bit for bit the tracked oracle."""
    pkg = gpu_pkg
    for G in (pkg.problems.lattice3d_precision(25), pkg.problems.random_sparse_precision(16384, 6)):
        d = G.shape[0]
        c = pkg.problems.column_norms(G)
        nch, T = 4096, 0.5
        with pkg.Ensemble(nch, d, trace_capacity=int(1.5 * d * T) + 1024) as ens:
            ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
            ens.set_target(pkg.GaussianTarget(G))
            ens.set_gradient_tracking(True)
            ens.set_state_synthetic(0.0, c, 0x5EED0000)
            ens.run(T, pkg._lib.RUN_STOP_BEFORE)
            cnt = ens.counters()
            assert np.all(cnt["status"] == pkg._lib.CHAIN_OK)
            for q in (0, nch - 1):
                x0, th0 = O.synthetic_state(0x5EED0000 + q, d)
                rt = O.spdmp_zigzag(G, None, G, x0, th0, c, T, seed=0x5EED0000 + q, stop_before_T=True, tracked=True)
                fa = ens.final_state(q, 1)
                check_chain_bitwise(ens.trace(q, counters=cnt), fa["t"][0], fa["x"][0], fa["theta"][0], fa["acc"][0], cnt["num"][q], None, rt)


def adversarial(pkg, which):
    """Graphs that stress the zone logic of the speculative kernels: many events of one iteration are neighbours (distance 1) or share a neighbour
    (distance 2)."""
    if which == "cliques8":      # d / 8 complete graphs K8: |G1| = |S| = 8, every pair inside a clique conflicts
        n = 320
        B = sp.kron(sp.identity(n, format="csc"), sp.csc_matrix(-0.3 * (np.ones((8, 8)) - np.eye(8))), format="csc")
    elif which == "path":        # tridiagonal: |G1| <= 3, |S| <= 5 (not the 2-d lattice's blob geometry)
        d = 2400
        B = sp.diags([-np.ones(d - 1), -np.ones(d - 1)], [-1, 1], format="csc")
    elif which == "triangular":  # 6 neighbours: the 50 x 50 triangular lattice, |G1| = 7, |S| = 19
        n = 50
        idx = np.arange(n * n).reshape(n, n)
        pairs = [(idx[:, :-1], idx[:, 1:]), (idx[:-1, :], idx[1:, :]), (idx[:-1, :-1], idx[1:, 1:])]
        a = np.concatenate([p[0].ravel() for p in pairs])
        b = np.concatenate([p[1].ravel() for p in pairs])
        W = sp.coo_matrix((-np.ones(a.size), (a, b)), shape=(n * n, n * n))
        B = sp.csc_matrix(W + W.T)
    else:
        raise KeyError(which)
    G = sp.csc_matrix(B + sp.diags(0.05 - np.asarray(B.sum(axis=0)).ravel()))
    G.sort_indices()
    return G


@pytest.mark.parametrize("which,T", [("cliques8", 6.0), ("path", 8.0), ("triangular", 5.0)])
@pytest.mark.parametrize("tracked", [True, False])
def test_adversarial_graphs_both_evaluations(gpu_pkg, which, T, tracked, trackp_form):
    """Cliques of 8, a path and the triangular lattice on the one-proposal-per-lane tracked kernel and on the 8-event kernel of the moving evaluation
    (`zz_local_spec8g_kernel`): bit for bit their oracles."""
    pkg = gpu_pkg
    G = adversarial(pkg, which)
    d = G.shape[0]
    assert np.diff(G.indptr).max() <= 8 and abs(G - G.T).max() == 0
    rng = np.random.default_rng(17)
    nch = 3
    x0 = rng.standard_normal((nch, d))
    th0 = rng.choice([-1.0, 1.0], (nch, d))
    c = pkg.problems.column_norms(G)
    tr, (t, x, th), (acc, num), _ = pkg.spdmp(pkg.GaussianTarget(G), 0.0, x0, th0, T, c, pkg.ZigZag(G, np.zeros(d)), seed=1700, tracked=tracked)
    with pkg.Ensemble(1, d) as ens:
        ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
        ens.set_target(pkg.GaussianTarget(G))
        ens.set_gradient_tracking(tracked)
        ens.set_state_synthetic(0.0, c, 1)
        ens.run(0.05)
        assert ens.kernel_name() == (("zz_local_trackp2_kernel<LAT=false>" if trackp_form == "two_waves" else "zz_local_trackp_kernel<LAT=false>")
                                     if tracked else "zz_local_spec8g_kernel")
    for q in range(nch):
        r = O.spdmp_zigzag(G, None, G, x0[q], th0[q], c, T, seed=1700 + q, tracked=tracked)
        assert r["status"] == 0 and len(r["events"]) > 1000
        check_chain_bitwise(tr[q].events, t[q], x[q], th[q], acc[q], num[q], None, r)


def test_c3g_random_graph_at_width_over_a_longer_horizon(gpu_pkg):
    """Config C3G (random pattern, <= 6 entries per column, d = 16384, 4096 chains) over T = 4 -- 3.3e5 proposals per chain, 1.4e9 in all -- on BOTH
    evaluations off the stencil (zz_local_trackp_kernel<LAT=false>, zz_local_spec8g_kernel): 32 chains of each run (every 132nd and the last) and every chain on which the two evaluations disagree, bit
    for bit against their oracles -- counters and the whole final state --, and the two evaluations agreeing on the counters of (nearly) every
    chain: a chain may leave the moving evaluation's index sequence at a rounding flip (tests/test_gpu_track_horizon.py), a handful at most here."""
    from concurrent.futures import ThreadPoolExecutor
    import os
    pkg = gpu_pkg
    G = pkg.problems.random_sparse_precision(16384, 6)
    d = G.shape[0]
    c = pkg.problems.column_norms(G)
    nch, T, seed0 = 4096, 4.0, 0x5EED0000
    chains = sorted(set(list(range(0, nch, 132)) + [nch - 1]))[:32]
    cnts, finals, enss = {}, {}, {}
    try:
        for tracked in (True, False):
            ens = enss[tracked] = pkg.Ensemble(nch, d, trace_capacity=0)
            ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
            ens.set_target(pkg.GaussianTarget(G))
            ens.set_gradient_tracking(tracked)
            ens.set_state_synthetic(0.0, c, seed0)
            for t in (1.0, 2.0, 3.0, T):
                ens.run(t, pkg._lib.RUN_STOP_BEFORE)
            cn = ens.counters()
            assert np.all(cn["status"] == pkg._lib.CHAIN_OK)
            assert ens.kernel_name().startswith("zz_local_trackp_kernel" if tracked else "zz_local_spec8g_kernel")
            cnts[tracked] = cn
        differ = (cnts[True]["num"] != cnts[False]["num"]) | (cnts[True]["nacc"] != cnts[False]["nacc"]) | (cnts[True]["ndraw_main"] != cnts[False]["ndraw_main"])
        # every chain that left the moving evaluation's index sequence is checked too: it must still be the sequential tracked sampler bit for bit
        # (a rounding flip, not a commit of the speculative kernel that the sequential sampler would not make)
        chains = sorted(set(chains) | set(int(k) for k in np.flatnonzero(differ)[:8]))
        for tracked in (True, False):
            finals[tracked] = {k: enss[tracked].final_state(k, 1) for k in chains}
    finally:
        for e in enss.values():
            e.close()

    def run(args):
        k, tracked = args
        x0, th0 = O.synthetic_state(seed0 + k, d)
        return O.spdmp_zigzag(G, None, G, x0, th0, c, T, seed=seed0 + k, stop_before_T=True, want_trace=False, tracked=tracked)

    jobs = [(k, tr) for tr in (True, False) for k in chains]
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as pool:
        res = dict(zip(jobs, pool.map(run, jobs)))
    for (k, tracked), r in res.items():
        assert r["status"] == 0
        ck, fs = cnts[tracked][k], finals[tracked][k]
        assert (int(ck["num"]), int(ck["nacc"]), int(ck["ndraw_main"])) == (r["num"], r["nacc"], r["ndraw_main"]), (k, tracked)
        assert np.array_equal(fs["acc"][0], r["acc"]) and np.array_equal(fs["theta"][0], r["theta"]), (k, tracked)
        assert np.array_equal(fs["t"][0], r["t"]) and np.array_equal(fs["x"][0], r["x"]), (k, tracked)
    print("C3G random6 to T = %g: %.4g proposals, chains whose tracked counters differ from the moving evaluation's: %s" %
          (T, cnts[True]["num"].sum(), np.flatnonzero(differ).tolist()))
    assert cnts[True]["num"].sum() > 1.0e9 and np.count_nonzero(differ) <= 8
