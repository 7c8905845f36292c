"""ctypes binding of the CPU oracle (oracle/pdmp_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product
package never does.  The oracle restates the reference's algorithm (citations in pdmp_oracle.c).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB_PATH = os.path.join(_ORACLE_DIR, "_build", "liboracle.so")

ORC_OK, ORC_BOUND_VIOLATED, ORC_STALLED, ORC_TRACE_LIMIT = 0, 1, 2, 3

EVENT_DTYPE = np.dtype([("t", "<f8"), ("i", "<i8"), ("x", "<f8"), ("theta", "<f8")])
EVENT1D_DTYPE = np.dtype([("t", "<f8"), ("x", "<f8"), ("theta", "<f8")])


def build_oracle(force=False):
    src = os.path.join(_ORACLE_DIR, "pdmp_oracle.c")
    hdrs = [os.path.join(_ORACLE_DIR, "pdmp_oracle.h"), os.path.join(_ORACLE_DIR, "trace_oracle.c"),
            os.path.join(os.path.dirname(_ORACLE_DIR), "include", "pdmp_detmath.h")]
    if not force and os.path.exists(_LIB_PATH):
        newest = max(os.path.getmtime(p) for p in [src] + hdrs)
        if os.path.getmtime(_LIB_PATH) >= newest:
            return _LIB_PATH
    subprocess.check_call(["make", "-C", _ORACLE_DIR, "-B", "CC=gcc"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _Csc(C.Structure):
    _fields_ = [("n", C.c_int64), ("colptr", C.c_void_p), ("rowval", C.c_void_p), ("nzval", C.c_void_p)]


class _Trace(C.Structure):
    _fields_ = [("ev", C.c_void_p), ("n", C.c_int64), ("cap", C.c_int64)]


class _ZZParams(C.Structure):
    _fields_ = [("bound_gamma", C.POINTER(_Csc)), ("bound_mu", C.c_void_p), ("sigma", C.c_void_p),
                ("lambda_ref", C.c_double), ("rho", C.c_double),
                ("target_gamma", C.POINTER(_Csc)), ("target_mu", C.c_void_p),
                ("move_all", C.c_int), ("adapt", C.c_int), ("factor", C.c_double),
                ("seed", C.c_uint64), ("max_events", C.c_int64), ("stop_before_T", C.c_int),
                ("target_kind", C.c_int), ("lg_A", C.POINTER(_Csc)), ("lg_At", C.POINTER(_Csc)), ("lg_y", C.c_void_p),
                ("lg_ny", C.c_void_p), ("lg_mu", C.c_void_p), ("lg_gamma0", C.c_double), ("lg_k", C.c_int64),
                ("flow_kind", C.c_int), ("adaptscale", C.c_int), ("sigma_out", C.c_void_p), ("local_bound", C.c_int),
                ("tracked", C.c_int), ("nbr_G", C.POINTER(_Csc))]


class _ZZResult(C.Structure):
    _fields_ = [("num", C.c_int64), ("nacc", C.c_int64), ("nrefresh", C.c_int64),
                ("ndraw_main", C.c_uint64), ("ndraw_global", C.c_uint64),
                ("t_last", C.c_double), ("status", C.c_int)]


class _BpsParams(C.Structure):
    _fields_ = [("gamma", C.POINTER(_Csc)), ("mu", C.c_void_p), ("lambda_ref", C.c_double),
                ("rho", C.c_double), ("c", C.c_double), ("adapt", C.c_int), ("factor", C.c_double),
                ("seed", C.c_uint64), ("max_events", C.c_int64), ("flow_kind", C.c_int), ("flow_mu", C.c_void_p),
                ("mass_L", C.POINTER(_Csc)), ("local_bound", C.c_int), ("subsample", C.c_int),
                ("target_gamma", C.POINTER(_Csc)), ("target_mu", C.c_void_p)]


class _BpsResult(C.Structure):
    _fields_ = [("num", C.c_int64), ("nacc", C.c_int64), ("nrefresh", C.c_int64), ("nevents", C.c_int64),
                ("ndraw_main", C.c_uint64), ("t_last", C.c_double), ("c_out", C.c_double),
                ("status", C.c_int)]


class _StickyParams(C.Structure):
    _fields_ = [("bound_gamma", C.POINTER(_Csc)), ("bound_mu", C.c_void_p),
                ("target_gamma", C.POINTER(_Csc)), ("target_mu", C.c_void_p), ("kappa", C.c_void_p),
                ("adapt", C.c_int), ("factor", C.c_double), ("reversible", C.c_int),
                ("strong_upperbounds", C.c_int), ("seed", C.c_uint64), ("max_events", C.c_int64),
                ("logistic", C.POINTER(_ZZParams)), ("nbr_G", C.POINTER(_Csc))]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build_oracle())
        L.orc_poisson_time.restype = C.c_double
        L.orc_poisson_time.argtypes = [C.c_double] * 3
        L.orc_poisson_time3.restype = C.c_double
        L.orc_poisson_time3.argtypes = [C.c_double] * 4
        L.orc_idot.restype = C.c_double
        L.orc_idot.argtypes = [C.POINTER(_Csc), C.c_int64, C.c_void_p]
        L.orc_pq_new.restype = C.c_void_p
        L.orc_pq_new.argtypes = [C.c_int64]
        L.orc_pq_free.argtypes = [C.c_void_p]
        L.orc_pq_enqueue.argtypes = [C.c_void_p, C.c_int64, C.c_double]
        L.orc_pq_set.argtypes = [C.c_void_p, C.c_int64, C.c_double]
        L.orc_pq_get.restype = C.c_double
        L.orc_pq_get.argtypes = [C.c_void_p, C.c_int64]
        L.orc_pq_peek.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_double)]
        L.orc_pq_len.restype = C.c_int64
        L.orc_pq_len.argtypes = [C.c_void_p]
        L.orc_pq_check.restype = C.c_int
        L.orc_pq_check.argtypes = [C.c_void_p]
        L.orc_trace_init.argtypes = [C.POINTER(_Trace)]
        L.orc_trace_free.argtypes = [C.POINTER(_Trace)]
        L.orc_spdmp_zigzag.restype = C.c_int
        L.orc_spdmp_zigzag.argtypes = [C.c_int64, C.POINTER(_ZZParams), C.c_double, C.c_double, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(_Trace),
                                       C.POINTER(_ZZResult)]
        L.orc_pdmp_zigzag1d.restype = C.c_int64
        L.orc_pdmp_zigzag1d.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                        C.c_int, C.c_double, C.c_uint64, C.c_void_p, C.c_int64,
                                        C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.orc_pdmp_bps.restype = C.c_int
        L.orc_pdmp_bps.argtypes = [C.c_int64, C.POINTER(_BpsParams), C.c_double, C.c_double, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                   C.POINTER(_BpsResult)]
        L.orc_sspdmp_zigzag.restype = C.c_int
        L.orc_sspdmp_zigzag.argtypes = [C.c_int64, C.POINTER(_StickyParams), C.c_double, C.c_double,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(_Trace),
                                        C.POINTER(_ZZResult)]
        L.orc_spdmp_zigzag_ensemble.restype = C.c_double
        L.orc_spdmp_zigzag_ensemble.argtypes = [C.c_int64, C.POINTER(_ZZParams), C.c_double, C.c_double,
                                                C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                                C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.orc_math_probe.argtypes = [C.c_uint64, C.c_int64, C.c_void_p]
        L.orc_log.restype = C.c_double
        L.orc_log.argtypes = [C.c_double]
        L.orc_u01.restype = C.c_double
        L.orc_u01.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64]
        L.orc_randn.restype = C.c_double
        L.orc_randn.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64]
        L.orc_philox.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_synthetic_state.argtypes = [C.c_uint64, C.c_int64, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def math_probe(seed, n):
    out = np.empty((8, n))
    lib().orc_math_probe(int(seed), int(n), out.ctypes.data)
    return out


def philox(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    o = np.zeros(4, dtype=np.uint32)
    lib().orc_philox(c.ctypes.data, k.ctypes.data, o.ctypes.data)
    return o


def synthetic_state(seed, d):
    x = np.empty(d)
    th = np.empty(d)
    lib().orc_synthetic_state(int(seed), int(d), x.ctypes.data, th.ctypes.data)
    return x, th


class CscHolder:
    """Keeps numpy buffers of a scipy CSC matrix alive next to the C struct that points at them."""

    def __init__(self, A):
        import scipy.sparse as sp
        A = sp.csc_matrix(A)
        A.sort_indices()
        self.n = A.shape[1]
        self.colptr = np.ascontiguousarray(A.indptr, dtype=np.int64)
        self.rowval = np.ascontiguousarray(A.indices, dtype=np.int64)
        self.nzval = np.ascontiguousarray(A.data, dtype=np.float64)
        self.c = _Csc(self.n, self.colptr.ctypes.data, self.rowval.ctypes.data, self.nzval.ctypes.data)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def poisson_time(a, b, u):
    return lib().orc_poisson_time(float(a), float(b), float(u))


def poisson_time3(a, b, c, u):
    return lib().orc_poisson_time3(float(a), float(b), float(c), float(u))


def idot(A, j, x):
    h = A if isinstance(A, CscHolder) else CscHolder(A)
    x = _f64(x)
    return lib().orc_idot(C.byref(h.c), int(j), x.ctypes.data)


def spdmp_zigzag(bound_gamma, bound_mu, target_gamma, x0, theta0, c, T, *, t0=0.0, target_mu=None,
                 sigma=None, lambda_ref=0.0, rho=0.0, move_all=False, adapt=False, factor=1.8, seed=1,
                 max_events=0, stop_before_T=False, want_trace=True, logistic=None, factboomerang=False,
                 adaptscale=False, local_bound=False, tracked=False, G=None):
    """Local ZigZag (reference spdmp / pdmp for ZigZag).  Returns dict(events, t, x, theta, acc, num, c, ...).
    tracked=True: the tracked-gradient evaluation of the same process (the bitwise checker of the device's tracked kernels)."""
    L = lib()
    gb = bound_gamma if isinstance(bound_gamma, CscHolder) else CscHolder(bound_gamma)
    gt = target_gamma if isinstance(target_gamma, CscHolder) else CscHolder(target_gamma)
    d = gb.n
    mu = _f64(bound_mu if bound_mu is not None else np.zeros(d))
    sg = _f64(sigma if sigma is not None else np.ones(d))
    tmu = _f64(target_mu) if target_mu is not None else None
    p = _ZZParams(C.pointer(gb.c), mu.ctypes.data, sg.ctypes.data, lambda_ref, rho, C.pointer(gt.c),
                  tmu.ctypes.data if tmu is not None else None, int(move_all), int(adapt), factor, seed,
                  max_events, int(stop_before_T))
    p.flow_kind = 1 if factboomerang else 0
    sg_out = np.array(sg)
    p.adaptscale = int(adaptscale)
    p.local_bound = int(local_bound)
    p.tracked = int(tracked)
    if G is not None:  # the optional neighbourhood argument: a sparse matrix whose column patterns are the G[i]
        gh = G if isinstance(G, CscHolder) else CscHolder(G)
        p.nbr_G = C.pointer(gh.c)
    p.sigma_out = sg_out.ctypes.data
    if logistic is not None:  # dict(A, At, y, ny, mu, gamma0, k): target_kind 1
        lA = logistic["A"] if isinstance(logistic["A"], CscHolder) else CscHolder(logistic["A"])
        lAt = logistic["At"] if isinstance(logistic["At"], CscHolder) else CscHolder(logistic["At"])
        ly, lny, lmu = _f64(logistic["y"]), _f64(logistic["ny"]), _f64(logistic["mu"])
        p.target_kind = 1
        p.lg_A, p.lg_At = C.pointer(lA.c), C.pointer(lAt.c)
        p.lg_y, p.lg_ny, p.lg_mu = ly.ctypes.data, lny.ctypes.data, lmu.ctypes.data
        p.lg_gamma0, p.lg_k = float(logistic["gamma0"]), int(logistic["k"])
    x = _f64(x0).copy()
    th = _f64(theta0).copy()
    cc = _f64(c).copy()
    t = np.empty(d)
    acc = np.zeros(d, dtype=np.int64)
    tr = _Trace()
    L.orc_trace_init(C.byref(tr))
    res = _ZZResult()
    st = L.orc_spdmp_zigzag(d, C.byref(p), t0, T, x.ctypes.data, th.ctypes.data, cc.ctypes.data, t.ctypes.data,
                            acc.ctypes.data, C.byref(tr) if want_trace else None, C.byref(res))
    ev = np.empty(0, dtype=EVENT_DTYPE)
    if want_trace and tr.n:
        buf = (C.c_char * (tr.n * EVENT_DTYPE.itemsize)).from_address(tr.ev)
        ev = np.frombuffer(buf, dtype=EVENT_DTYPE).copy()
    L.orc_trace_free(C.byref(tr))
    return dict(events=ev, t=t, x=x, theta=th, acc=acc, num=res.num, nacc=res.nacc, nrefresh=res.nrefresh,
                c=cc, status=st, ndraw_main=res.ndraw_main, ndraw_global=res.ndraw_global, t_last=res.t_last,
                sigma=sg_out)


class _ParResult(C.Structure):
    _fields_ = [("num", C.c_int64), ("nacc", C.c_int64), ("rounds", C.c_int64), ("spawns", C.c_int64),
                ("seconds", C.c_double), ("status", C.c_int)]


def parallel_spdmp(bound_gamma, bound_mu, target_gamma, x0, theta0, c, T, K, delta, *, t0=0.0, adapt=False, factor=1.8,
                   seed=1, want_trace=True):
    """src/parallel.jl parallel_spdmp restated with pthreads (CPU baseline and the checker of the device's partitioned mode).  The threads of a
    round share no data, so the result does not depend on their timing (tests/test_oracle_samplers.py checks it)."""
    L = lib()
    gb = bound_gamma if isinstance(bound_gamma, CscHolder) else CscHolder(bound_gamma)
    gt = target_gamma if isinstance(target_gamma, CscHolder) else CscHolder(target_gamma)
    d = gb.n
    mu = _f64(bound_mu if bound_mu is not None else np.zeros(d))
    sg = np.ones(d)
    p = _ZZParams(C.pointer(gb.c), mu.ctypes.data, sg.ctypes.data, 0.0, 0.0, C.pointer(gt.c), None, 0, int(adapt), factor, seed,
                  0, 0)
    x = _f64(x0).copy()
    th = _f64(theta0).copy()
    cc = _f64(c).copy()
    t = np.empty(d)
    tr = _Trace()
    L.orc_trace_init(C.byref(tr))
    res = _ParResult()
    L.orc_parallel_spdmp.restype = C.c_int
    st = L.orc_parallel_spdmp(C.c_int64(d), C.byref(p), C.c_int(K), C.c_double(delta), C.c_double(t0), C.c_double(T),
                              x.ctypes.data_as(C.c_void_p), th.ctypes.data_as(C.c_void_p), cc.ctypes.data_as(C.c_void_p),
                              t.ctypes.data_as(C.c_void_p), C.byref(tr) if want_trace else None, C.byref(res))
    ev = np.empty(0, dtype=EVENT_DTYPE)
    if want_trace and tr.n:
        buf = (C.c_char * (tr.n * EVENT_DTYPE.itemsize)).from_address(tr.ev)
        ev = np.frombuffer(buf, dtype=EVENT_DTYPE).copy()
        ev = ev[np.argsort(ev["t"], kind="stable")]  # sort!(Ξ.events, by=ev->ev[1]), src/parallel.jl:167
    L.orc_trace_free(C.byref(tr))
    return dict(events=ev, t=t, x=x, theta=th, c=cc, num=res.num, nacc=res.nacc, rounds=res.rounds, spawns=res.spawns,
                seconds=res.seconds, status=st)


def pdmp_zigzag1d(mu, sigma2, x0, theta0, T, c, *, adapt=False, factor=2.0, seed=1, cap=1 << 20):
    L = lib()
    out = np.empty(cap, dtype=EVENT1D_DTYPE)
    acc = C.c_int64()
    num = C.c_int64()
    n = L.orc_pdmp_zigzag1d(mu, sigma2, x0, theta0, T, c, int(adapt), factor, seed, out.ctypes.data, cap,
                            C.byref(acc), C.byref(num))
    if n < 0:
        raise RuntimeError("Tuning parameter `c` too small.")
    return out[:min(n, cap)].copy(), acc.value, num.value


class Orc1dParams(C.Structure):
    _fields_ = [("flow", C.c_int32), ("adapt", C.c_int32), ("factor", C.c_double), ("mu", C.c_double), ("sigma2", C.c_double),
                ("noise", C.c_double), ("b_sigma", C.c_double), ("b_mu", C.c_double), ("b_lambda", C.c_double), ("seed", C.c_uint64)]


class Orc1dState(C.Structure):
    _fields_ = [("t", C.c_double), ("x", C.c_double), ("theta", C.c_double), ("c", C.c_double), ("a", C.c_double), ("b", C.c_double),
                ("t_next", C.c_double), ("t_ref", C.c_double), ("ndraw", C.c_uint64), ("num", C.c_int64), ("acc", C.c_int64),
                ("started", C.c_int32), ("status", C.c_int32)]


def pdmp_1d(mu, sigma2, x0, theta0, T, c, *, flow="zigzag", boomerang=(1.0, 0.0, 1.0), noise=0.0, adapt=False, factor=2.0, seed=1, cap=1 << 16):
    """src/zigzagboom1d.jl:34-67 for ZigZag1d() / Boomerang1d(Σ, μ, λref) on ∇ϕ(x) = (x − mu)/sigma2 + noise (rand() − 0.5); the event
    buffer of `cap` entries is refilled until the run ends.  Returns dict(events, acc, num, c, ndraw, status)."""
    L = lib()
    L.orc_pdmp_1d.restype = C.c_int64
    L.orc_pdmp_1d.argtypes = [C.POINTER(Orc1dParams), C.POINTER(Orc1dState), C.c_double, C.c_void_p, C.c_int64]
    p = Orc1dParams(1 if flow == "boomerang" else 0, int(adapt), factor, mu, sigma2, noise, boomerang[0], boomerang[1], boomerang[2], seed)
    st = Orc1dState()
    st.x, st.theta, st.c, st.started = x0, theta0, c, 0
    parts = []
    while True:
        out = np.empty(cap, dtype=EVENT1D_DTYPE)
        n = L.orc_pdmp_1d(C.byref(p), C.byref(st), T, out.ctypes.data, cap)
        parts.append(out[:n].copy())
        if st.status != 3:
            break
    return dict(events=np.concatenate(parts), acc=st.acc, num=st.num, c=st.c, ndraw=st.ndraw, status=st.status, t=st.t, x=st.x, theta=st.theta)


def pdmp_bps(gamma, mu, x0, theta0, c, T, *, t0=0.0, lambda_ref=1.0, rho=0.0, adapt=False, factor=2.0,
             seed=1, max_events=0, ev_cap=0, want_events=True, boomerang_mu=None, mass_L=None, local_bound=False,
             subsample=False, target=None):
    """BPS (gamma, mu = flow AND target) or, with boomerang_mu, Boomerang(·, boomerang_mu, λref; ρ) on the Gaussian
    target (gamma, mu).  mass_L: the lower-triangular factor F.L (scipy sparse / dense), None = identity."""
    L = lib()
    g = gamma if isinstance(gamma, CscHolder) else CscHolder(gamma)
    d = g.n
    muv = _f64(mu if mu is not None else np.zeros(d))
    p = _BpsParams(C.pointer(g.c), muv.ctypes.data, lambda_ref, rho, c, int(adapt), factor, seed, max_events)
    if boomerang_mu is not None:
        fmu = _f64(boomerang_mu)
        p.flow_kind, p.flow_mu = 1, fmu.ctypes.data
    if mass_L is not None:
        mh = mass_L if isinstance(mass_L, CscHolder) else CscHolder(mass_L)
        p.mass_L = C.pointer(mh.c)
    p.local_bound, p.subsample = int(bool(local_bound)), int(bool(subsample))
    if target is not None:  # (Γt, μt): BouncyParticle whose target differs from B.Γ(x − B.μ)
        th_ = target[0] if isinstance(target[0], CscHolder) else CscHolder(target[0])
        tmu_ = _f64(target[1] if target[1] is not None else np.zeros(d))
        p.target_gamma, p.target_mu = C.pointer(th_.c), tmu_.ctypes.data
    x = _f64(x0).copy()
    th = _f64(theta0).copy()
    if want_events:
        t_ev = np.empty(ev_cap)
        x_ev = np.empty((ev_cap, d))
        th_ev = np.empty((ev_cap, d))
        args = (t_ev.ctypes.data, x_ev.ctypes.data, th_ev.ctypes.data, ev_cap)
    else:
        t_ev = x_ev = th_ev = None
        args = (None, None, None, 0)
    res = _BpsResult()
    st = L.orc_pdmp_bps(d, C.byref(p), t0, T, x.ctypes.data, th.ctypes.data, *args, C.byref(res))
    n = min(res.nevents, ev_cap)
    return dict(t_ev=None if t_ev is None else t_ev[:n], x_ev=None if x_ev is None else x_ev[:n],
                theta_ev=None if th_ev is None else th_ev[:n], x=x, theta=th, t=res.t_last, num=res.num,
                nacc=res.nacc, nrefresh=res.nrefresh, nevents=res.nevents, c=res.c_out, status=st,
                ndraw_main=res.ndraw_main)


def sspdmp_zigzag(bound_gamma, bound_mu, target_gamma, x0, theta0, c, kappa, T, *, t0=0.0, target_mu=None,
                  adapt=False, factor=1.5, reversible=False, strong_upperbounds=False, seed=1, max_events=0, logistic=None, G=None):
    L = lib()
    gb = bound_gamma if isinstance(bound_gamma, CscHolder) else CscHolder(bound_gamma)
    gt = target_gamma if isinstance(target_gamma, CscHolder) else CscHolder(target_gamma)
    d = gb.n
    mu = _f64(bound_mu if bound_mu is not None else np.zeros(d))
    tmu = _f64(target_mu) if target_mu is not None else None
    kap = _f64(kappa)
    p = _StickyParams(C.pointer(gb.c), mu.ctypes.data, C.pointer(gt.c),
                      tmu.ctypes.data if tmu is not None else None, kap.ctypes.data, int(adapt), factor,
                      int(reversible), int(strong_upperbounds), seed, max_events)
    if logistic is not None:  # dict(A, At, y, ny, mu, gamma0, k): ∇ϕmoving with SelfMoving()
        lA = logistic["A"] if isinstance(logistic["A"], CscHolder) else CscHolder(logistic["A"])
        lAt = logistic["At"] if isinstance(logistic["At"], CscHolder) else CscHolder(logistic["At"])
        ly, lny, lmu = _f64(logistic["y"]), _f64(logistic["ny"]), _f64(logistic["mu"])
        zp = _ZZParams()
        zp.target_kind = 1
        zp.lg_A, zp.lg_At = C.pointer(lA.c), C.pointer(lAt.c)
        zp.lg_y, zp.lg_ny, zp.lg_mu = ly.ctypes.data, lny.ctypes.data, lmu.ctypes.data
        zp.lg_gamma0, zp.lg_k = float(logistic["gamma0"]), int(logistic["k"])
        p.logistic = C.pointer(zp)
    if G is not None:
        gh = G if isinstance(G, CscHolder) else CscHolder(G)
        p.nbr_G = C.pointer(gh.c)
    x = _f64(x0).copy()
    th = _f64(theta0).copy()
    cc = _f64(c).copy()
    t = np.empty(d)
    tr = _Trace()
    L.orc_trace_init(C.byref(tr))
    res = _ZZResult()
    st = L.orc_sspdmp_zigzag(d, C.byref(p), t0, T, x.ctypes.data, th.ctypes.data, cc.ctypes.data, t.ctypes.data,
                             C.byref(tr), C.byref(res))
    ev = np.empty(0, dtype=EVENT_DTYPE)
    if tr.n:
        buf = (C.c_char * (tr.n * EVENT_DTYPE.itemsize)).from_address(tr.ev)
        ev = np.frombuffer(buf, dtype=EVENT_DTYPE).copy()
    L.orc_trace_free(C.byref(tr))
    return dict(events=ev, t=t, x=x, theta=th, num=res.num, nacc=res.nacc, c=cc, status=st,
                ndraw_main=res.ndraw_main, t_last=res.t_last)


def spdmp_zigzag_ensemble(bound_gamma, bound_mu, target_gamma, x0, theta0, c, T, *, t0=0.0, seed0=1,
                          nthreads=1, adapt=False, factor=1.8):
    """CPU baseline: nchains chains on nthreads threads, events counted only. Returns (seconds, num, acc)."""
    L = lib()
    gb = bound_gamma if isinstance(bound_gamma, CscHolder) else CscHolder(bound_gamma)
    gt = target_gamma if isinstance(target_gamma, CscHolder) else CscHolder(target_gamma)
    d = gb.n
    mu = _f64(bound_mu if bound_mu is not None else np.zeros(d))
    sg = np.ones(d)
    p = _ZZParams(C.pointer(gb.c), mu.ctypes.data, sg.ctypes.data, 0.0, 0.0, C.pointer(gt.c), None, 0,
                  int(adapt), factor, 0, 0, 0)
    x0 = _f64(x0)
    theta0 = _f64(theta0)
    nch = x0.shape[0]
    cc = _f64(c)
    num = C.c_int64()
    acc = C.c_int64()
    secs = L.orc_spdmp_zigzag_ensemble(d, C.byref(p), t0, T, nch, x0.ctypes.data, theta0.ctypes.data,
                                       cc.ctypes.data, seed0, nthreads, C.byref(num), C.byref(acc))
    return secs, num.value, acc.value


class PQ:
    """Thin handle on the oracle's indexed heap (src/priorityqueue.jl)."""

    def __init__(self, cap):
        self.h = lib().orc_pq_new(cap)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_pq_free(self.h)
            self.h = None

    def enqueue(self, k, v):
        lib().orc_pq_enqueue(self.h, k, v)

    def __setitem__(self, k, v):
        lib().orc_pq_set(self.h, k, v)

    def __getitem__(self, k):
        return lib().orc_pq_get(self.h, k)

    def peek(self):
        k = C.c_int64()
        v = C.c_double()
        lib().orc_pq_peek(self.h, C.byref(k), C.byref(v))
        return k.value, v.value

    def __len__(self):
        return lib().orc_pq_len(self.h)

    def check(self):
        return bool(lib().orc_pq_check(self.h))


# ---- what callers do next with a FactTrace: oracle/trace_oracle.c (src/trace.jl restated event by event)
def _ev_arrays(events):
    return (np.ascontiguousarray(events["t"], dtype=np.float64), np.ascontiguousarray(events["i"], dtype=np.int64),
            np.ascontiguousarray(events["x"], dtype=np.float64), np.ascontiguousarray(events["theta"], dtype=np.float64))


def trace_mean(t0, x0, events):
    """Statistics.mean(trace), src/trace.jl:182-200"""
    et, ei, ex, _ = _ev_arrays(events)
    x0 = _f64(x0)
    y = np.empty(x0.size)
    L = lib()
    L.orc_trace_mean.restype = None
    L.orc_trace_mean.argtypes = [C.c_int64, C.c_double, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_trace_mean(x0.size, float(t0), x0.ctypes.data, et.size, et.ctypes.data, ei.ctypes.data, ex.ctypes.data, y.ctypes.data)
    return y


def trace_inclusion_prob(t0, x0, events):
    """inclusion_prob(trace), src/trace.jl:161-178"""
    et, ei, ex, _ = _ev_arrays(events)
    x0 = _f64(x0)
    y = np.empty(x0.size)
    L = lib()
    L.orc_trace_inclusion_prob.restype = None
    L.orc_trace_inclusion_prob.argtypes = [C.c_int64, C.c_double, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_trace_inclusion_prob(x0.size, float(t0), x0.ctypes.data, et.size, et.ctypes.data, ei.ctypes.data, ex.ctypes.data, y.ctypes.data)
    return y


def trace_cummean(t0, x0, events):
    """cummean(trace::FactTrace), src/trace.jl:203-226: (t, y) after every event, in event order (entry k belongs to coordinate events["i"][k])"""
    et, ei, ex, _ = _ev_arrays(events)
    x0 = _f64(x0)
    ot, oy = np.empty(et.size), np.empty(et.size)
    L = lib()
    L.orc_trace_cummean.restype = None
    L.orc_trace_cummean.argtypes = [C.c_int64, C.c_double, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_trace_cummean(x0.size, float(t0), x0.ctypes.data, et.size, et.ctypes.data, ei.ctypes.data, ex.ctypes.data, ot.ctypes.data, oy.ctypes.data)
    return ot, oy


def trace_discretize(t0, x0, th0, events, dt):
    """collect(discretize(trace, dt)) for a ZigZag FactTrace, src/trace.jl:100-125"""
    et, ei, ex, eth = _ev_arrays(events)
    x0, th0 = _f64(x0), _f64(th0)
    span = (et[-1] - t0) if et.size else 0.0
    cap = int(span / dt) + 8
    ts, xs = np.empty(cap), np.empty((cap, x0.size))
    L = lib()
    L.orc_trace_discretize.restype = C.c_int64
    L.orc_trace_discretize.argtypes = [C.c_int64, C.c_double, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_double, C.c_int64, C.c_void_p, C.c_void_p]
    q = L.orc_trace_discretize(x0.size, float(t0), x0.ctypes.data, th0.ctypes.data, et.size, et.ctypes.data, ei.ctypes.data, ex.ctypes.data,
                               eth.ctypes.data, float(dt), cap, ts.ctypes.data, xs.ctypes.data)
    assert q <= cap
    return ts[:q], xs[:q]


def trace_subtrace(J, events):
    """subtrace(tr, J), src/trace.jl:275-290: (indices of the kept events, their new 0-based coordinates)"""
    _, ei, _, _ = _ev_arrays(events)
    J = np.ascontiguousarray(J, dtype=np.int64)
    ok, oi = np.empty(ei.size, dtype=np.int64), np.empty(ei.size, dtype=np.int64)
    L = lib()
    L.orc_trace_subtrace.restype = C.c_int64
    L.orc_trace_subtrace.argtypes = [C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    m = L.orc_trace_subtrace(J.size, J.ctypes.data, ei.size, ei.ctypes.data, ok.ctypes.data, oi.ctypes.data)
    return ok[:m], oi[:m]
