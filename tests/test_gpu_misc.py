"""Error paths of the C ABI and the RCCL plumbing on one GPU (-m gpu)."""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp

import oracle_lib as O

pytestmark = pytest.mark.gpu


def test_unsupported_and_invalid_inputs_fail_loudly(gpu_pkg):
    pkg = gpu_pkg
    E = pkg._lib.PdmpError
    d = 6
    G = pkg.problems.gmrf_precision(3)[:6, :6].tocsc()
    G.sort_indices()
    with pkg.Ensemble(2, d) as ens:
        with pytest.raises(E) as ei:  # state before flow/target
            ens.set_state(0.0, np.zeros((2, d)), np.ones((2, d)), np.ones(d), np.arange(2))
        assert ei.value.code == 1
        nodiag = sp.csc_matrix(np.array([[0.0, 1.0], [1.0, 2.0]]))
        with pkg.Ensemble(1, 2) as e2:
            with pytest.raises(E) as ei:  # Γ[0,0] structurally zero: i must belong to G1[i]
                e2.set_flow(pkg.ZigZag(nodiag, np.zeros(2), np.ones(2)))
            assert ei.value.code == 4
        ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
        dense = sp.csc_matrix(np.ones((d, d)))
        with pytest.raises(E) as ei:  # target entries outside the flow's pattern (src/sfact.jl:116)
            ens.set_target(pkg.GaussianTarget(dense))
        assert ei.value.code == 4
        with pytest.raises(E):  # run before state
            ens.run(1.0)
    with pytest.raises(E) as ei:  # BPS keeps the state in registers (and scratch beyond 1024 coordinates): d <= 4096
        pkg.Ensemble(1, 4097, sampler=pkg._lib.SAMPLER_BPS)
    assert ei.value.code == 4
    with pkg.Ensemble(1, d, sampler=pkg._lib.SAMPLER_STICKY_ZIGZAG) as es:
        es.set_flow(pkg.ZigZag(G, np.zeros(d)))
        es.set_target(pkg.GaussianTarget(G))
        with pytest.raises(E):  # κ missing
            es.set_state(0.0, np.zeros((1, d)), np.ones((1, d)), np.ones(d), np.arange(1))


def test_stalled_chain_is_reported_not_hung(gpu_pkg):
    """Every bound rate identically zero (a = 0, b < 0 -> poisson_time = Inf, src/poissontime.jl:22-23): the queue is all
    +Inf, the chain is reported PDMP_CHAIN_STALLED and the kernel returns instead of spinning."""
    pkg = gpu_pkg
    d = 4
    G = sp.identity(d, format="csc")
    with pkg.Ensemble(2, d, trace_capacity=16) as ens:
        ens.set_flow(pkg.ZigZag(-1.0 * G, np.zeros(d), np.ones(d)))
        ens.set_target(pkg.GaussianTarget(G))
        ens.set_state(0.0, np.zeros((2, d)), np.ones((2, d)), np.zeros(d), np.arange(2))
        ens.run(10.0)
        cnt = ens.counters()
        assert np.all(cnt["status"] == pkg._lib.CHAIN_STALLED) and np.all(cnt["nevents"] == 0) and np.all(cnt["num"] == 0)
        ens.run(20.0)  # sticky status: a stalled chain stays put
        assert np.all(ens.counters()["status"] == pkg._lib.CHAIN_STALLED)


def test_rccl_world1_gather_of_device_resident_traces():
    """The post-run collectives on the real backend (nccl == RCCL), world_size 1, on a zero-copy view of the engine's trace.
    Runs in a fresh interpreter in the order bench.py uses for N > 1: torch (and its HIP runtime) first, then the engine."""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_world1_script.py")
    r = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_degenerate_sizes_match_oracle(gpu_pkg):
    """Edge cases: d = 1, a diagonal Γ (every neighbourhood is {i}), one chain, T <= t0 (`while t′ < T` never runs: empty trace,
    state and clocks untouched, no draw beyond the d initial ones)."""
    import scipy.sparse as sp
    pkg = gpu_pkg
    rng = np.random.default_rng(8)
    for d in (1, 3, 65):
        G = sp.diags(0.5 + rng.random(d), format="csc")
        x0 = rng.standard_normal((2, d))
        th0 = rng.choice([-1.0, 1.0], (2, d))
        c = pkg.problems.column_norms(G) + 0.1
        for T in (0.0, 25.0):
            tr, (t, x, th), (acc, num), _ = pkg.spdmp(pkg.GaussianTarget(G), 0.0, x0, th0, T, c, pkg.ZigZag(G, np.zeros(d)), seed=900)
            for k in range(2):
                r = O.spdmp_zigzag(G, None, G, x0[k], th0[k], c, T, seed=900 + k)
                assert r["status"] == 0 and len(tr[k].events) == len(r["events"])
                for f in ("i", "t", "x", "theta"):
                    assert np.array_equal(tr[k].events[f], r["events"][f]), (d, T, k, f)
                assert int(num[k]) == r["num"] and np.array_equal(x[k], r["x"]) and np.array_equal(t[k], r["t"])
                if T == 0.0:
                    assert len(tr[k].events) == 0 and np.array_equal(x[k], x0[k]) and np.all(t[k] == 0.0)
    # the same degenerate horizon for the other samplers
    B = pkg.BouncyParticle(sp.identity(4, format="csc"), np.zeros(4), 1.0)
    trb, (tb, xb, thb), (accb, numb), _ = pkg.pdmp(None, 0.0, x0[:1, :4].copy(), th0[:1, :4].copy(), 0.0, 1e-3, B, seed=5)
    assert len(trb[0].t) == 0 and int(numb[0]) == 0 and np.array_equal(xb[0], x0[0, :4])
    Gs = sp.diags(np.ones(3), format="csc")
    trs, (ts, xs, ths), (accs, nums), _ = pkg.sspdmp(pkg.GaussianTarget(Gs), 0.0, x0[:1, :3].copy(), th0[:1, :3].copy(), 0.0,
                                                     np.ones(3), pkg.ZigZag(Gs, np.zeros(3)), np.ones(3), seed=6)
    assert len(trs[0].events) == 0 and np.array_equal(xs[0], x0[0, :3])


def test_entry_points_of_the_wrong_family_return_a_status(gpu_pkg):
    """include/pdmp_mi355.h promises return codes, never crashes: factorised-sampler calls on a PDMP_SAMPLER_BPS ensemble (and the
    reverse), a malformed target pattern, and batch means on a rotating flow are all refused with a status."""
    pkg = gpu_pkg
    L = pkg._lib
    d = 4
    I4 = sp.identity(d, format="csc")
    with pkg.Ensemble(2, d, sampler=L.SAMPLER_BPS, trace_capacity=8) as ens:
        ens.set_flow_bps(pkg.BouncyParticle(I4, np.zeros(d), 1.0))
        ens.set_state_bps(0.0, np.zeros((2, d)), np.ones((2, d)), 1e-3, np.arange(2, dtype=np.uint64))
        for call in (lambda: ens.set_flow(pkg.ZigZag(I4, np.zeros(d))),  # (set_target IS accepted here: a BouncyParticle with a target of its own)
                     lambda: ens.set_state(0.0, np.zeros((2, d)), np.ones((2, d)), np.ones(d), np.arange(2)),
                     lambda: ens.set_state_synthetic(0.0, np.ones(d), 1), lambda: ens.final_state(), lambda: ens.batch_means(0.0, 1.0),
                     lambda: ens.trace(0, 0, 1), lambda: ens.ess_begin(0.0), lambda: ens.set_local_bound(True)):
            with pytest.raises(L.PdmpError) as ei:
                call()
            assert ei.value.code == L.PDMP_ERR_INVALID, str(ei.value)
        ens.run(2.0)  # the ensemble is still usable
        assert ens.counters()["num"].min() > 0
    with pkg.Ensemble(2, d, trace_capacity=8) as ens:
        for call in (lambda: ens.set_flow_bps(pkg.BouncyParticle(I4, np.zeros(d), 1.0)),
                     lambda: ens.set_state_bps(0.0, np.zeros((2, d)), np.ones((2, d)), 1e-3, np.arange(2, dtype=np.uint64)),
                     lambda: ens.bps_final_state()):
            with pytest.raises(L.PdmpError) as ei:
                call()
            assert ei.value.code == L.PDMP_ERR_INVALID
        ens.set_flow(pkg.ZigZag(I4, np.zeros(d)))
        # a target whose colptr runs backwards / whose rows leave the matrix
        import ctypes as C
        cp = np.array([0, 1, 0, 1, 2], dtype=np.int64)
        rv = np.array([0, 3], dtype=np.int64)
        nz = np.ones(2)
        st = L.load().pdmp_ensemble_set_target_gaussian_csc(ens._h, cp.ctypes.data, rv.ctypes.data, nz.ctypes.data, None)
        assert st == L.PDMP_ERR_INVALID
        cp = np.array([0, 1, 2, 3, 4], dtype=np.int64)
        rv = np.array([0, 1, 2, 7], dtype=np.int64)
        st = L.load().pdmp_ensemble_set_target_gaussian_csc(ens._h, cp.ctypes.data, rv.ctypes.data, np.ones(4).ctypes.data, None)
        assert st == L.PDMP_ERR_INVALID
    # a FactBoomerang path rotates between events: the linear path integrals are refused
    G = pkg.problems.maintest_precision(8)
    with pkg.Ensemble(1, 8, trace_capacity=64) as ens:
        ens.set_flow(pkg.FactBoomerang(G, np.zeros(8), 0.3))
        ens.set_target(pkg.GaussianTarget(G))
        ens.set_state(0.0, np.zeros((1, 8)), np.ones((1, 8)), 2 * pkg.problems.column_norms(G), np.array([3], dtype=np.uint64))
        ens.run(1.0, L.RUN_STOP_BEFORE)
        for call in (lambda: ens.batch_means(0.0, 1.0), lambda: ens.ess_begin(1.0)):
            with pytest.raises(L.PdmpError) as ei:
                call()
            assert ei.value.code == L.PDMP_ERR_UNSUPPORTED


def test_engine_rccl_entry_points_world1(gpu_pkg):
    """pdmp_comm_* / pdmp_ensemble_gather_traces / pdmp_ensemble_reduce_moments (RCCL linked by the library itself, no torch in the process):
    a one-rank communicator runs the very code path of N ranks -- ncclCommInitRank, ncclAllGather of the counts, device-side compaction of the
    trace segments, the grouped send / recv (empty at world 1: the root's own segment is a device copy), ncclReduce, ncclAllReduce."""
    pkg = gpu_pkg
    G = pkg.problems.gmrf_precision(8)
    d, nch, cap = 64, 5, 4096
    c = pkg.problems.column_norms(G)
    with pkg.parallel.Comm(0, 1, 0) as comm, pkg.Ensemble(nch, d, trace_capacity=cap) as ens:
        ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
        ens.set_target(pkg.GaussianTarget(G))
        ens.set_state_synthetic(0.0, c, 77)
        ens.run(10.0, pkg._lib.RUN_STOP_BEFORE)
        cnt = ens.counters()
        widths, counts, ev = comm.gather_traces(ens)
        host = np.concatenate([ens.trace(k, counters=cnt) for k in range(nch)])
        assert widths.tolist() == [nch] and counts.tolist() == cnt["ntrace"].tolist() and len(host) > 1000
        assert np.array_equal(ev, host)  # chain-major, event order inside a chain
        w2, c2, devbuf = comm.gather_traces(ens, to_host=False)
        assert devbuf[1] == len(host) and devbuf[0]
        s1, s2 = ens.batch_means(0.0, 10.0)
        with pkg.Ensemble(nch, d, trace_capacity=cap) as e2:  # (batch_means keeps a running J: a fresh ensemble for the collective form)
            e2.set_flow(pkg.ZigZag(G, np.zeros(d)))
            e2.set_target(pkg.GaussianTarget(G))
            e2.set_state_synthetic(0.0, c, 77)
            e2.run(10.0, pkg._lib.RUN_STOP_BEFORE)
            r1, r2 = comm.reduce_moments(e2, 0.0, 10.0)
        assert np.array_equal(r1, s1) and np.array_equal(r2, s2)
        assert comm.allreduce([1.5, -2.0], "max").tolist() == [1.5, -2.0] and comm.allreduce([3.0], "sum").tolist() == [3.0]
        comm.barrier()
        with pytest.raises(pkg._lib.PdmpError):  # an ensemble without a trace buffer has nothing to gather
            with pkg.Ensemble(1, d) as e0:
                e0.set_flow(pkg.ZigZag(G, np.zeros(d)))
                e0.set_target(pkg.GaussianTarget(G))
                e0.set_state_synthetic(0.0, c, 1)
                comm.gather_traces(e0)


def test_default_library_carries_one_kernel_per_path(gpu_pkg):
    """The measured-slower cross-implementations (zz_local_exactp_kernel, zz_logistic_rows_kernel) are not in libpdmp_mi355.so: asking for them by
    name is refused with a status that says where they live (the parity build, lib/libpdmp_mi355.parity.so), never served by another kernel."""
    pkg = gpu_pkg
    L = pkg._lib
    G = pkg.problems.gmrf_precision(48)
    d = G.shape[0]
    with pkg.Ensemble(1, d) as ens:
        with pytest.raises(L.PdmpError) as ei:
            ens.debug_set_logistic_rows(32)
        assert ei.value.code == L.PDMP_ERR_UNSUPPORTED and "parity" in str(ei.value)
    with pkg.Ensemble(1, d) as ens:
        ens.debug_set_kernel("exactp")
        ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
        ens.set_target(pkg.GaussianTarget(G))
        ens.set_state_synthetic(0.0, pkg.problems.column_norms(G), 1)
        with pytest.raises(L.PdmpError) as ei:
            ens.run(0.1)
        assert ei.value.code == L.PDMP_ERR_UNSUPPORTED and "parity" in str(ei.value)
