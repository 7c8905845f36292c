"""Host-side mirror of the reference's sampler entry points for the device-resident hot path.

    spdmp(∇ϕ, t0, x0, θ0, T, c, F::ZigZag, args...; factor=1.8, adapt=false, seed) = Ξ, (t, x, θ), (acc, num), c
                                                                        (src/sfact.jl:162-163,211,214)

Differences forced by the C ABI: `∇ϕ, args...` is replaced by an enumerated `target` (GaussianTarget);
coordinates are 0-based; x0/θ0 may be [nchains, d] to run an ensemble (outputs then carry a leading chain
axis).  Everything else -- argument order, keyword names, the 4-tuple returned, the error raised when the
bound `c` is too small with adapt=false -- follows the reference.
"""
import copy

import numpy as np
import scipy.sparse as sp

from . import _lib
from .engine import Ensemble
from .flows import (Boomerang, Boomerang1d, BouncyParticle, FactBoomerang, LocalBound, FactTrace, GaussianTarget, GaussianTarget1d, LogisticTarget,
                    PDMPTrace, ZigZag, ZigZag1d)

DEFAULT_SEED = 0x5EED0000


def _drain(ens, events):
    cnt = ens.counters()
    for k in range(ens.nchains):
        n = int(cnt["ntrace"][k])
        if n:
            events[k].append(ens.trace(k, 0, n, counters=cnt))
    ens.trace_reset()


def _split_G(GF, G):
    """spdmp(∇ϕ, t0, x0, θ0, T, c, F) or spdmp(∇ϕ, t0, x0, θ0, T, c, G, F) as in the reference (src/sfact.jl:162,214); G may also be a keyword."""
    if len(GF) == 1:
        return G, GF[0]
    if len(GF) == 2 and G is None:
        return GF[0], GF[1]
    raise TypeError("expected spdmp(target, t0, x0, θ0, T, c, [G,] F, ...)")


def spdmp(target, t0, x0, θ0, T, c, *GF, factor=1.8, adapt=False, adaptscale=False, seed=DEFAULT_SEED, device=0,
          trace_capacity=None, trace=True, tracked=False, G=None):
    """Local ZigZag: spdmp(∇ϕ, t0, x0, θ0, T, c, [G,] F::ZigZag, args...) (src/sfact.jl:162,214); without G: G = Matched().
    Returns Ξ, (t, x, θ), (acc, num), c like the reference (:211).

    G: the neighbourhoods a proposal moves before its gradient is taken (:82,171-179) -- a sparse matrix whose column patterns are the
    G[i] (e.g. the target's Γ when the bounding F.Γ is sparser) or a sequence of index arrays; G[i] ⊇ G1[i] = pattern of F.Γ[:, i] is
    asserted as in the reference (:177).

    adaptscale=True (src/sfact.jl:86-99) tunes σ in the refresh branch.  Like the reference, a single-chain call mutates
    F.σ in place; for an ensemble every chain's tuned σ is the `σ` of the flow attached to its trace (Ξ[k].F.σ).

    tracked=True (engine-only keyword): the tracked-gradient evaluation of the same process (pdmp_ensemble_set_gradient_tracking) -- the
    engine's fast path (what bench.py's headline times: 2.5 x the rate of the default on C3).  It is NOT the reference's arithmetic: the sums
    Γ[:,i]·x and Γ[:,i]·θ are carried along instead of gathered, so event times and positions agree with the default to ~1e-13 (tested to 1e-9),
    and indices, outcomes and counters are identical until such a difference flips a thinning test or the order of two almost simultaneous
    events: measured 3 of 4096 chains by T = 20 on the 128 x 128 lattice (6e-10 per proposal; tests/test_gpu_track_horizon.py).  A chain that
    has left is still a realisation of the same process, and the tracked arithmetic itself is pinned bit for bit by the oracle's
    spdmp_zigzag_tracked.  The default (False) follows the reference's evaluation order bit for bit; PdmpError(UNSUPPORTED) where no tracked
    kernel serves the graph / options."""
    G, F = _split_G(GF, G)
    return _zigzag(_lib.SAMPLER_ZIGZAG_LOCAL, target, t0, x0, θ0, T, c, F, factor, adapt, seed, device, trace_capacity, trace,
                   adaptscale=adaptscale, tracked=tracked, G=G)


def pdmp(target, *args, factor=1.8, adapt=False, subsample=False, seed=DEFAULT_SEED, device=0, trace_capacity=None, trace=True):
    """pdmp(∇ϕ, t0, x0, θ0, T, c, F, ...) -- the d-dimensional drivers, see _pdmp_nd -- or, with a 1-d flow as the sixth argument,
    pdmp(∇ϕ, x, θ, T, c, Flow::Union{ZigZag1d, Boomerang1d}; adapt=false, factor=2.0) -> Ξ, acc/num  (src/zigzagboom1d.jl:34-67)."""
    if len(args) == 5 and isinstance(args[4], (ZigZag1d, Boomerang1d)):
        if subsample:
            raise TypeError("subsample is a keyword of the non-factorised pdmp (BouncyParticle / Boomerang)")
        x0, θ0, T, c, Flow = args
        return _pdmp_1d(target, x0, θ0, T, c, Flow, 2.0 if factor == 1.8 else factor, adapt, seed, device, trace_capacity)
    if len(args) != 6:
        raise TypeError("expected pdmp(target, t0, x0, θ0, T, c, F, ...) or pdmp(target, x, θ, T, c, Flow1d, ...)")
    return _pdmp_nd(target, *args, factor=factor, adapt=adapt, subsample=subsample, seed=seed, device=device, trace_capacity=trace_capacity,
                    trace=trace)


def _pdmp_1d(target, x0, θ0, T, c, Flow, factor, adapt, seed, device, trace_capacity):
    """The 1-d samplers as an ensemble (one chain per lane, pdmp_1d_run): scalars run one chain and return (Ξ, acc/num) like the reference
    (:66), arrays of starting points run len(x0) chains (seeds seed + k) and return lists.  Ξ: structured array of (t, x, theta), the first
    entry being (0, x0, θ0) (:36).  c may be a scalar or one value per chain."""
    import ctypes as C
    if not isinstance(target, GaussianTarget1d):
        raise TypeError("the 1-d samplers take a GaussianTarget1d (∇ϕ(x) = (x − μ)/σ² + noise (rand() − 0.5), test/test1d.jl:9-10)")
    scalar = np.ndim(x0) == 0
    x0 = np.atleast_1d(np.asarray(x0, dtype=np.float64))
    n = len(x0)
    θ0 = np.broadcast_to(np.asarray(θ0, dtype=np.float64), (n,))
    cc = np.broadcast_to(np.asarray(c, dtype=np.float64), (n,))
    cap = int(trace_capacity) if trace_capacity else int(max(1024, min(1 << 20, 4 * max(float(T), 1.0))))
    boom = isinstance(Flow, Boomerang1d)
    cfg = _lib.Config1d(C.sizeof(_lib.Config1d), int(device), 1 if boom else 0, int(bool(adapt)), float(factor), n, cap, float(target.μ),
                        float(target.σ2), float(target.noise), Flow.Σ if boom else 1.0, Flow.μ if boom else 0.0, Flow.λref if boom else 1.0)
    st = np.zeros(n, dtype=_lib.STATE1D_DTYPE)
    st["x"], st["theta"], st["c"] = x0, θ0, cc
    seeds = np.uint64(seed) + np.arange(n, dtype=np.uint64)
    ev = np.empty((n, cap), dtype=_lib.EVENT1D_DTYPE)
    nev = np.zeros(n, dtype=np.int64)
    L = _lib.load()
    parts = [[] for _ in range(n)]
    while True:
        _lib.check(L.pdmp_1d_run(C.byref(cfg), st.ctypes.data, seeds.ctypes.data, float(T), ev.ctypes.data, nev.ctypes.data))
        for k in range(n):
            if nev[k]:
                parts[k].append(ev[k, :nev[k]].copy())
        if np.any(st["status"] == _lib.CHAIN_BOUND_VIOLATED):
            raise RuntimeError("Tuning parameter `c` too small.")  # :55
        if not _lib.needs_rerun(st["status"]):
            break
    Ξ = [np.concatenate(p) for p in parts]
    ratio = st["acc"] / np.maximum(st["num"], 1)
    return (Ξ[0], float(ratio[0])) if scalar else (Ξ, ratio)


def _pdmp_nd(target, t0, x0, θ0, T, c, F, *, factor=1.8, adapt=False, subsample=False, seed=DEFAULT_SEED, device=0,
             trace_capacity=None, trace=True):
    """pdmp(∇ϕ, t0, x0, θ0, T, c, F::ZigZag, args...) = spdmp(..., All(), ...) (src/sfact.jl:236): every proposal moves
    ALL coordinates (no sparsity assumption on ∇ϕ); same return value as spdmp.

    pdmp(∇ϕ!, t0, x0, θ0, T, c, B::BouncyParticle; adapt, factor=2.0) (src/not_fact_samplers.jl:117,395-396) when F is a
    BouncyParticle: the target is ∇ϕ!(y, x) = B.Γ(x − B.μ) (pass target=None) or a GaussianTarget of its own -- ab(…GlobalBound…) then
    keeps the flow's B.Γ, B.μ while gradient, rate and reflection use the target's (src/not_fact_samplers.jl:26-28,122) --, c is the scalar of GlobalBound(c) or a
    LocalBound(c) (src/not_fact_samplers.jl:29-31; the second derivative v = θ'Γθ is the Gaussian target's own);
    `subsample` as in the reference (:53,90); returns Ξ::PDMPTrace, (t, x, θ), (acc, num), c."""
    if isinstance(F, BouncyParticle):
        if target is not None and not isinstance(target, GaussianTarget):
            raise TypeError("BouncyParticle: target is None (∇ϕ!(y, x) = B.Γ(x − B.μ)) or a GaussianTarget of its own")
        return _bps(t0, x0, θ0, T, c, F, 2.0 if factor == 1.8 else factor, adapt, seed, device, trace_capacity, trace,
                    target=target, subsample=subsample)
    if isinstance(F, Boomerang):  # pdmp(∇ϕ!, t0, x0, θ0, T, c, B::Boomerang) (test/maintest.jl:139-154); target = GaussianTarget
        if not isinstance(target, GaussianTarget):
            raise TypeError("Boomerang: target must be a GaussianTarget (∇ϕ!(y, x) = Γ(x − μ))")
        return _bps(t0, x0, θ0, T, c, F, 2.0 if factor == 1.8 else factor, adapt, seed, device, trace_capacity, trace,
                    target=target, subsample=subsample)
    if subsample:
        raise TypeError("subsample is a keyword of the non-factorised pdmp (BouncyParticle / Boomerang)")
    return _zigzag(_lib.SAMPLER_ZIGZAG_ALL, target, t0, x0, θ0, T, c, F, factor, adapt, seed, device, trace_capacity, trace)


def sspdmp(target, t0, x0, θ0, T, c, *GFκ, reversible=False, strong_upperbounds=False, factor=1.5, adapt=False,
           seed=DEFAULT_SEED, device=0, trace_capacity=None, trace=True, G=None):
    """Sticky ZigZag: sspdmp(∇ϕ, t0, x0, θ0, T, c, [G,] F::ZigZag, κ, args...; reversible, strong_upperbounds, factor=1.5,
    adapt) (src/ss_fact.jl:159-160,217) -> Ξ, (t, x, θ), (acc, num), c with scalar acc, num (:175,214).  G as in spdmp (:167-172)."""
    if len(GFκ) not in (2, 3):
        raise TypeError("expected sspdmp(target, t0, x0, θ0, T, c, [G,] F, κ, ...)")
    G, F = _split_G(GFκ[:-1], G)
    κ = GFκ[-1]
    return _zigzag(_lib.SAMPLER_STICKY_ZIGZAG, target, t0, x0, θ0, T, c, F, factor, adapt, seed, device, trace_capacity, trace,
                   sticky=(np.asarray(κ, dtype=np.float64), reversible, strong_upperbounds), G=G)


class Partition:
    """Partition(nt, n) (src/parallel.jl:4-31): n coordinates in nt chunks of k = n ÷ nt; pt(i) -> (chunk, offset), pt(chunk, offset) -> i
    (0-based here)."""

    def __init__(self, nt, n):
        self.nt, self.n, self.k = int(nt), int(n), int(n) // int(nt)

    def __len__(self):
        return self.nt

    def __call__(self, *a):
        if len(a) == 1:
            return divmod(int(a[0]), self.k)
        return int(a[0]) * self.k + int(a[1])


def parallel_spdmp(partition, target, t0, x0, θ0, T, c, G, F, *, factor=1.8, adapt=False, Δ=0.1, seed=DEFAULT_SEED, device=0,
                   trace_capacity=None, trace=True):
    """parallel_spdmp(partition, ∇ϕ, t0, x0, θ0, T, c, G, F::ZigZag; factor=1.8, adapt=false, Δ=0.1) (src/parallel.jl:104-175):
    the local ZigZag with the coordinates cut into len(partition) chunks, one worker per chunk and a coordinator -- on the device one
    WAVEFRONT per chunk (pdmp_ensemble_run_partitioned).  F.Γ is the bounding precision: its pattern G1 must not leave the chunks
    ("Upper bounds may not depend across chunks.", :124-127).  G: None = the pattern of F.Γ (:113-115), or a sparse matrix whose pattern
    ⊇ F.Γ's and ⊇ the target's gives the neighbourhoods that are moved before a gradient (test/testparallel.jl:49 passes the target's).
    Returns Ξ (sorted by time, :167), (t, x, θ), (acc, num); with adapt=True a numpy `c` is updated in place like the reference's."""
    if not isinstance(F, ZigZag) or isinstance(F, FactBoomerang):
        raise TypeError("the device path of parallel_spdmp supports F::ZigZag")
    if not isinstance(target, GaussianTarget):
        raise TypeError("target must be a GaussianTarget")
    K = len(partition) if not np.isscalar(partition) else int(partition)
    x0 = np.asarray(x0, dtype=np.float64)
    θ0 = np.asarray(θ0, dtype=np.float64)
    single = x0.ndim == 1
    X0, TH0 = np.atleast_2d(x0), np.atleast_2d(θ0)
    nch, d = X0.shape
    # the flow tables carry G: the bounding Γ on G's pattern with explicit zeros, and the mask of its own structural entries
    def pattern(A):  # structural pattern (explicit zeros count, as in Julia's rowvals / nzrange)
        A = sp.csc_matrix(A)
        B = sp.csc_matrix((np.ones(A.nnz), A.indices.copy(), A.indptr.copy()), shape=A.shape)
        B.sort_indices()
        B.sum_duplicates()
        B.data[:] = 1.0
        return B

    Gb = F.Γ
    own = pattern(Gb)
    pg = own if G is None else pattern(G)
    U = (pg + own + pattern(target.Γ)).tocsc()
    U.sort_indices()
    if U.nnz != pg.nnz:
        raise ValueError("G must contain the patterns of F.Γ (G ⊇ G1, src/parallel.jl:119) and of the target")
    cols = np.repeat(np.arange(d), np.diff(U.indptr))
    rows = U.indices
    mask = np.asarray(own[rows, cols]).reshape(-1) != 0
    vals = np.asarray(sp.csc_matrix(Gb)[rows, cols]).reshape(-1)
    Γu = sp.csc_matrix((vals, rows.copy(), U.indptr.copy()), shape=Gb.shape)
    Fu = copy.copy(F)
    Fu.Γ = Γu  # (not through __post_init__: explicit zeros must stay)
    cc = np.asarray(c, dtype=np.float64)
    seeds = (np.uint64(seed) + np.arange(nch, dtype=np.uint64)) if np.isscalar(seed) else np.asarray(seed, np.uint64)
    if trace_capacity is None:
        trace_capacity = int(min(max(4096, 4.0 * d * max(T - t0, 1.0)), 1 << 24))
    cap = trace_capacity if trace else 0
    ens = Ensemble(nch, d, sampler=_lib.SAMPLER_ZIGZAG_LOCAL, adapt=adapt, factor=factor, device=device, trace_capacity=cap)
    try:
        ens.set_flow(Fu)
        ens.set_target(target)
        ens.set_state(t0, X0, TH0, cc, seeds)
        ens.run_partitioned(T, K, Δ, mask.astype(np.uint8))
        cnt = ens.counters()
        if np.any(cnt["status"] == _lib.CHAIN_BOUND_VIOLATED):
            raise RuntimeError("Tuning parameter `c` too small.")  # src/parallel.jl:42
        if np.any(cnt["status"] == _lib.CHAIN_TRACE_FULL):
            raise RuntimeError("trace_capacity too small for a partitioned run (it is not resumable): %d events" % int(cnt["nevents"].max()))
        events = [[] for _ in range(nch)]
        if trace:
            _drain(ens, events)
        fs = ens.final_state()
    finally:
        ens.close()
    traces = []
    for k_ in range(nch):
        ev = np.concatenate(events[k_]) if events[k_] else np.empty(0, dtype=_lib.EVENT_DTYPE)
        ev = ev[np.argsort(ev["t"], kind="stable")]  # sort!(Ξ.events, by=ev->ev[1]), :167
        traces.append(FactTrace(F, t0, X0[k_].copy(), TH0[k_].copy(), ev))
    acc, num = cnt["nacc"].astype(np.int64), cnt["num"].astype(np.int64)
    if adapt and single and isinstance(c, np.ndarray) and c.dtype == np.float64:
        c[:] = fs["c"][0]  # adapt!(c, i, factor) acts on the caller's vector
    if single:
        return traces[0], (fs["t"][0], fs["x"][0], fs["theta"][0]), (int(acc[0]), int(num[0]))
    return traces, (fs["t"], fs["x"], fs["theta"]), (acc, num)


def _pattern_of(G, d):
    """G as a CSC pattern: a sparse matrix (explicit zeros count, like rowvals / nzrange) or a sequence of index arrays / (i, indices) pairs."""
    if sp.issparse(G):
        G = sp.csc_matrix(G)
        return sp.csc_matrix((np.ones(G.nnz), G.indices.copy(), G.indptr.copy()), shape=G.shape)
    cols = [np.asarray(g[1] if isinstance(g, tuple) else g, dtype=np.int64) for g in G]
    if len(cols) != d:
        raise ValueError("G needs one neighbourhood per coordinate")
    indptr = np.concatenate([[0], np.cumsum([len(g) for g in cols])])
    P = sp.csc_matrix((np.ones(int(indptr[-1])), np.concatenate(cols) if cols else np.empty(0, np.int64), indptr), shape=(d, d))
    P.sort_indices()
    return P


def _zigzag(sampler, target, t0, x0, θ0, T, c, F, factor, adapt, seed, device, trace_capacity, trace, sticky=None,
            adaptscale=False, tracked=False, G=None):
    if not isinstance(F, (ZigZag, FactBoomerang)):
        raise TypeError("the device path supports F::ZigZag and F::FactBoomerang")
    if not isinstance(target, (GaussianTarget, LogisticTarget)):
        raise TypeError("target must be one of the device-resident families (GaussianTarget, LogisticTarget)")
    x0 = np.asarray(x0, dtype=np.float64)
    θ0 = np.asarray(θ0, dtype=np.float64)
    single = x0.ndim == 1
    X0 = np.atleast_2d(x0)
    TH0 = np.atleast_2d(θ0)
    nch, d = X0.shape
    local_bound = isinstance(c, LocalBound)  # spdmp(∇ϕ, t0, x0, θ0, T, C::LocalBound, F, args...), src/local.jl:95-149
    c = np.asarray(c.c if local_bound else c, dtype=np.float64)
    seeds = (np.uint64(seed) + np.arange(nch, dtype=np.uint64)) if np.isscalar(seed) else np.asarray(seed, np.uint64)
    if trace_capacity is None:
        # ~0.8 reflections per coordinate per unit time on the GMRF; generous first guess, refilled on demand
        trace_capacity = int(min(max(1024, 2.0 * d * max(T - t0, 1.0)), 1 << 22))
    cap = trace_capacity if trace else 0
    ens = Ensemble(nch, d, sampler=sampler, adapt=adapt, factor=factor, device=device,
                   trace_capacity=cap)
    try:
        ens.set_flow(F)
        if G is not None:
            try:
                ens.set_neighbourhood(_pattern_of(G, d))
            except _lib.PdmpError as exc:
                if "does not contain G1" in str(exc) or "must contain G1" in str(exc):
                    raise AssertionError("all(a.second ⊇ b.second for (a,b) in zip(G, G1)) -- src/sfact.jl:177") from exc
                raise
        ens.set_target(target)
        if sticky is not None:
            ens.set_sticky(*sticky)
        if adaptscale:
            ens.set_adaptscale(True)
        if local_bound:
            ens.set_local_bound(True)
        if tracked:
            ens.set_gradient_tracking(True)
        ens.set_state(t0, X0, TH0, c, seeds)
        events = [[] for _ in range(nch)]
        while True:
            ens.run(T, _lib.RUN_REFERENCE_TAIL)
            cnt = ens.counters()
            if np.any(cnt["status"] == _lib.CHAIN_BOUND_VIOLATED):
                raise RuntimeError("Tuning parameter `c` too small.")  # src/sfact.jl:124
            if trace:
                _drain(ens, events)
            if not _lib.needs_rerun(cnt["status"]):  # (both resume with the next run)
                break
        fs = ens.final_state()
        cnt = ens.counters()
        sig = ens.final_sigma() if adaptscale else None
    finally:
        ens.close()
    traces = []
    for k in range(nch):
        ev = np.concatenate(events[k]) if events[k] else np.empty(0, dtype=_lib.EVENT_DTYPE)
        Fk = F
        if sig is not None:
            if single:
                F.σ[:] = sig[0]  # the reference mutates F.σ (src/sfact.jl:90,97)
            else:
                Fk = copy.copy(F)
                Fk.σ = sig[k].copy()
        traces.append(FactTrace(Fk, t0, X0[k].copy(), TH0[k].copy(), ev))
    num = cnt["num"].astype(np.int64)
    c_out = fs["c"] if adapt else np.broadcast_to(c, (nch, d)).copy()
    acc = cnt["nacc"].astype(np.int64) if sticky is not None else fs["acc"]  # sticky: scalar acc (src/ss_fact.jl:175)
    if single:
        return traces[0], (fs["t"][0], fs["x"][0], fs["theta"][0]), (acc[0], int(num[0])), c_out[0]
    return traces, (fs["t"], fs["x"], fs["theta"]), (acc, num), c_out


def _bps(t0, x0, θ0, T, c, B, factor, adapt, seed, device, trace_capacity, trace, target=None, subsample=False):
    local_bound = isinstance(c, LocalBound)
    if local_bound:
        c = float(np.asarray(c.c, dtype=np.float64).reshape(-1)[0])
    x0 = np.asarray(x0, dtype=np.float64)
    θ0 = np.asarray(θ0, dtype=np.float64)
    single = x0.ndim == 1
    X0, TH0 = np.atleast_2d(x0), np.atleast_2d(θ0)
    nch, d = X0.shape
    seeds = (np.uint64(seed) + np.arange(nch, dtype=np.uint64)) if np.isscalar(seed) else np.asarray(seed, np.uint64)
    if trace_capacity is None:
        trace_capacity = int(min(max(256, 64 * max(T - t0, 1.0)), (1 << 28) // max(2 * d, 1)))
    cap = trace_capacity if trace else 0
    ens = Ensemble(nch, d, sampler=_lib.SAMPLER_BPS, adapt=adapt, factor=factor, device=device, trace_capacity=cap)
    try:
        if isinstance(B, Boomerang):
            ens.set_flow_boomerang(target, B)
        else:
            ens.set_flow_bps(B)
            if target is not None:  # ∇ϕ! of its own; ab(…GlobalBound…) keeps B.Γ, B.μ (src/not_fact_samplers.jl:26-28,122)
                ens.set_target(target)
        if local_bound or subsample:
            ens.set_bps_options(local_bound, subsample)
        ens.set_state_bps(t0, X0, TH0, float(c), seeds)
        ts = [[] for _ in range(nch)]
        xs = [[] for _ in range(nch)]
        ths = [[] for _ in range(nch)]
        while True:
            ens.run(T, _lib.RUN_REFERENCE_TAIL)
            cnt = ens.counters()
            if np.any(cnt["status"] == _lib.CHAIN_BOUND_VIOLATED):
                raise RuntimeError("Tuning parameter `c` too small.")  # src/not_fact_samplers.jl:82
            if trace:
                for k in range(nch):
                    if cnt["ntrace"][k]:
                        a, b_, c_ = ens.bps_trace(k, counters=cnt)
                        ts[k].append(a)
                        xs[k].append(b_)
                        ths[k].append(c_)
                ens.trace_reset()
            if not _lib.needs_rerun(cnt["status"]):  # (both resume with the next run)
                break
        fs = ens.bps_final_state()
        cnt = ens.counters()
    finally:
        ens.close()
    traces = []
    for k in range(nch):
        if ts[k]:
            traces.append(PDMPTrace(B, t0, X0[k].copy(), TH0[k].copy(), np.concatenate(ts[k]), np.concatenate(xs[k]),
                                    np.concatenate(ths[k])))
        else:
            traces.append(PDMPTrace(B, t0, X0[k].copy(), TH0[k].copy(), np.empty(0), np.empty((0, d)), np.empty((0, d))))
    acc, num = cnt["nacc"].astype(np.int64), cnt["num"].astype(np.int64)
    if single:
        return traces[0], (fs["t"][0], fs["x"][0], fs["theta"][0]), (int(acc[0]), int(num[0])), fs["c"][0]
    return traces, (fs["t"], fs["x"], fs["theta"]), (acc, num), fs["c"]
