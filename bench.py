#!/usr/bin/env python
"""bench.py -- reflection events/s (+ ESS/s) of the local-ZigZag hot path on config C3 (default), and with --config C2|C4|C5 the same
JSON line for BASELINE.json's other GPU configurations (their own metric, byte model and roofline object).

Workload (BASELINE.json configs[2], the one `metric` is quoted on):
    Γ = 0.01 I + gridlaplacian(128,128)  (scripts/gridlaplace.jl:4-21, scripts/gaussianrandomfield.jl:15), d = 16384,
    ∇ϕ(x,i) = Γ[:,i]·x, Z = ZigZag(Γ, 0), c[i] = ‖Γ[:,i]‖₂, x0 ~ N(0,I), θ0 ∈ {±1}, t0 = 0, adapt = false,
    ONE ensemble of 4096 chains: at N GPUs rank r runs chains [r 4096/N, (r+1) 4096/N) (strong scaling, the north star's form; chains are
    independent, no data-path collective; `--scaling weak` keeps 4096 chains PER GPU), Philox seeds 0x5EED0000 + chain.  At N = 1 the line also
    carries `strong_proxy` -- this GPU running the 2048 / 1024 / 512 chains that a rank of a 2 / 4 / 8-GPU job runs: the per-GPU term of the curve,
    measured -- and `pipeline`: the same steps with their trace consumed on the device (streaming mean + discretize) beside the sampler.
A "step" advances every chain by ΔT = 1.0 time units through pdmp_ensemble_run (one persistent kernel launch);
the initial state is generated on the device, so inputs are resident in HBM when the timed region starts.
Traces ARE written (32 B/event into a per-chain HBM segment, part of the algorithmic bytes) and recycled every step.

    python bench.py --gpus 1 --steps 20 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and, at N=1, `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
GRID = 128
CHAINS_PER_GPU = 4096
DT_STEP = 1.0
SEED0 = 0x5EED0000


def algorithmic_bytes(num, nacc):
    """224 B per proposal + 48 B per rejection + 616 B per accepted reflection."""
    return 224.0 * num + 48.0 * (num - nacc) + 616.0 * nacc


class _banner_to_stderr:
    """parallel.c_stdout_to_stderr without importing the package first (torch must be imported before the engine library is loaded): RCCL prints its
    version banner through C stdio on stdout; this line's stdout is ONE JSON line."""

    def __enter__(self):
        import ctypes
        self._libc = ctypes.CDLL(None)
        sys.stdout.flush()
        self._libc.fflush(None)
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *a):
        self._libc.fflush(None)
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def source_hash():
    """SHA-256 over the kernel sources and the shared headers: ties a PMC summary under profiles/ to the build it was measured on."""
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "zigzagboomerang.jl_amd", "csrc")
    for f in sorted(os.listdir(base)):
        h.update(open(os.path.join(base, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "pdmp_detmath.h"), "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline_config(pkg, config, budget_s=12.0):
    """The CPU oracle (kind="port") on the secondary configurations: an ENSEMBLE of independent chains over a pool of host threads, one chain
    per thread (the chains the GPU ensemble starts with: same seeds), after one chain alone on one thread (the reference's own way of running:
    src/sfact.jl:199-208 is a sequential loop).  ctypes releases the GIL for the duration of a call, so a Python thread pool IS one C thread per chain.
    Bounded sample: one chain alone for 1-2 seconds, then the pool with one chain of the same length per thread."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from concurrent.futures import ThreadPoolExecutor
    import oracle_lib as O
    import scipy.sparse as sp
    lim = host_cpu_limits()
    usable = lim["affinity"] or lim["cpu_count"]
    if lim["cgroup_quota_cpus"]:
        usable = max(1, min(usable, int(lim["cgroup_quota_cpus"] + 0.5)))
    nthr = int(min(usable, 64))
    if config == "C2":
        d = 1024
        Id = sp.identity(d, format="csc")

        def one(k, T):
            rng = np.random.default_rng(1000 + k)
            r = O.pdmp_bps(Id, None, rng.standard_normal(d), rng.standard_normal(d), 1e-3, T, lambda_ref=1.0, seed=SEED0 + k, want_events=False)
            return r["nevents"]
        T1, what = 40.0, "BPS chains d=1024"
    elif config == "C4":
        P = pkg.problems.logistic_problem(m=20)
        lg = dict(A=P["A"], At=P["At"], y=P["y"], ny=P["ny"], mu=P["mu"], gamma0=P["gamma0"], k=10)

        def one(k, T):
            rng = np.random.default_rng(2000 + k)
            r = O.spdmp_zigzag(P["Gdrop"], P["mu"], P["Gdrop"], P["x0"], P["sigma"] * rng.choice([-1.0, 1.0], P["p"]), P["c"], T, seed=SEED0 + k,
                               adapt=True, factor=5.0, logistic=lg, want_trace=False)
            return r["nacc"]
        T1, what = 60.0, "chains of the subsampled logistic regression (n=8840, p=442)"
    else:
        P = pkg.problems.spike_slab_logistic_problem(p=10_000, num_rows=args.c5_rows)
        lg = dict(A=P["A"], At=P["At"], y=P["y"], ny=P["ny"], mu=P["mu"], gamma0=P["gamma0"], k=12)

        def one(k, T):
            x0, th0 = O.synthetic_state(SEED0 + k, P["p"])
            r = O.sspdmp_zigzag(P["G"], P["mu"], P["G"], x0, th0, P["c"], P["kappa"], T, seed=SEED0 + k, adapt=True, factor=1.5, logistic=lg)
            return len(r["events"])
        T1, what = 1.0, "sticky chains of the p=10000 logistic spike-and-slab"
    # one chain alone (a third of the length): the single-thread rate, and the calibration of the pool's chain length
    Ts = T1 / 3
    for _ in range(4):  # (long enough to be timed: about two seconds)
        t0 = time.perf_counter()
        ev1 = one(0, Ts)
        s1 = time.perf_counter() - t0
        if s1 >= 1.0:
            break
        Ts *= min(30.0, 2.0 / max(s1, 1e-3))
    # the pool's chains run to the SAME length as the chain that was timed alone (a chain's cost per unit of process time is not constant --
    # C5's sticky chains get several times more expensive after their first time unit -- and 16 threads that stream a 131 MB design share
    # the host's memory bandwidth: extrapolating the length from the single run once made this leg take 20 minutes)
    T = float(Ts)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=nthr) as pool:
        evs = list(pool.map(lambda k: one(k, T), range(nthr)))
    secs = time.perf_counter() - t0
    return {"value": sum(evs) / secs, "unit": "events/s", "cores": nthr, "kind": "port",
            "sample": f"{nthr} {what} to T={T:.3g} on {nthr} host threads, one chain per thread ({secs:.1f} s of wall time; the GPU ensemble's first "
                      f"{nthr} seeds); before it one chain alone to T={Ts:.3g} ({s1:.1f} s): the single-thread figure",
            "single_thread_events_per_s": ev1 / s1, "parallel_efficiency": (sum(evs) / secs) / (nthr * ev1 / s1), "host": lim}


def host_cpu_limits():
    """What the process may actually use: logical CPUs, scheduler affinity, and the cgroup CPU quota (v2 cpu.max / v1 cfs_*)."""
    info = {"cpu_count": os.cpu_count() or 1, "affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
            "cgroup_quota_cpus": None}
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            info["cgroup_quota_cpus"] = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                info["cgroup_quota_cpus"] = q / per
        except Exception:
            pass
    return info


def cpu_baseline(pkg, G, c, budget_s=20.0):
    """The CPU oracle (a C port of the reference algorithm, kind="port") on the host cores, bounded sample: one chain per thread,
    thread counts swept once over {1, 8, 32, 64, all usable}; the best rate is the reported value."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    d = G.shape[0]
    lim = host_cpu_limits()
    usable = lim["affinity"] or lim["cpu_count"]
    if lim["cgroup_quota_cpus"]:
        usable = max(1, min(usable, int(lim["cgroup_quota_cpus"] + 0.5)))
    sweep_n = sorted({n for n in (1, 8, 32, 64, usable) if n <= usable})
    # calibrate on one short chain, then give every setting of the sweep the same wall-time share of the budget
    x0, th0 = O.synthetic_state(SEED0, d)
    s1, n1, a1 = O.spdmp_zigzag_ensemble(G, None, G, x0[None], th0[None], c, 1.0, seed0=SEED0, nthreads=1)
    per_chain_per_T = max(s1, 1e-3)
    T = float(np.clip((budget_s / len(sweep_n)) / per_chain_per_T, 0.5, 10.0))
    nmax = sweep_n[-1]
    X0 = np.stack([O.synthetic_state(SEED0 + k, d)[0] for k in range(nmax)])
    TH0 = np.stack([O.synthetic_state(SEED0 + k, d)[1] for k in range(nmax)])
    sweep = {}
    best = None
    for n in sweep_n:
        secs, num, acc = O.spdmp_zigzag_ensemble(G, None, G, X0[:n], TH0[:n], c, T, seed0=SEED0, nthreads=n)
        sweep[n] = {"events_per_s": acc / secs, "proposals_per_s": num / secs, "seconds": secs}
        if best is None or acc / secs > best[1]:
            best = (n, acc / secs, num / secs, acc / max(num, 1))
    single = sweep[1]["events_per_s"]
    eff = best[1] / (single * best[0])
    # the reported figure: an ENSEMBLE over the pool (SURVEY 8 d5 ii) -- many more chains than threads, taken from a queue, about ten
    # seconds of wall time at the best thread count; the chains are the GPU ensemble's first ones (same seeds, same event sequences)
    nb = best[0]
    m = int(np.clip(round(10.0 / max(sweep[nb]["seconds"], 1e-3)), 2, 64))
    nens = nb * m
    st_e = [O.synthetic_state(SEED0 + k, d) for k in range(nens)]
    Xe, THe = np.stack([q[0] for q in st_e]), np.stack([q[1] for q in st_e])
    secs_e, num_e, acc_e = O.spdmp_zigzag_ensemble(G, None, G, Xe, THe, c, T, seed0=SEED0, nthreads=nb)
    ens = {"chains": nens, "threads": nb, "T": T, "seconds": secs_e, "events_per_s": acc_e / secs_e, "proposals_per_s": num_e / secs_e}
    best = (nb, acc_e / secs_e, num_e / secs_e, acc_e / max(num_e, 1))
    why = ""
    if eff < 0.3:
        why = (" -- scaling is %.2f of linear at %d threads: " % (eff, best[0]) +
               ("the cgroup grants %.1f CPUs; " % lim["cgroup_quota_cpus"] if lim["cgroup_quota_cpus"] else "") +
               "every chain walks 1.4 MB of records + heap at random (d = 16384), so beyond the cores' private caches the threads "
               "share the host's memory latency, not its arithmetic")
    # the reference's own multithreaded path, src/parallel.jl (ONE chain split over K chunk threads + a coordinator): its bound
    # must be block diagonal over the chunks, so the cross-chunk couplings are dropped from the bounding Γ and adapt is on
    par = None
    try:
        import scipy.sparse as sp
        K = 8
        k = d // K
        coo = sp.coo_matrix(G)
        keep = (coo.row // k) == (coo.col // k)
        Gb = sp.csc_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=G.shape)
        Gb.sort_indices()
        rp = O.parallel_spdmp(Gb, None, G, x0, th0, 2.0 * c, 2.0, K, 0.1, seed=SEED0, adapt=True, want_trace=False)
        if rp["status"] == 0:
            par = {"threads": K, "delta": 0.1, "events_per_s": rp["nacc"] / rp["seconds"], "sync_rounds": int(rp["rounds"]),
                   "sample": "one chain of config C3 to T=2, restatement of parallel_spdmp (src/parallel.jl:104-253)"}
    except Exception as exc:  # the baseline is reported-only: never fail the bench on it
        par = {"error": str(exc)}
    return {"value": best[1], "unit": "reflection events/s", "cores": best[0], "kind": "port", "parallel_jl": par,
            "sample": f"{nens} chains of config C3 (d=16384; the GPU ensemble's first {nens} seeds) to T={T:.2f} on a pool of {nb} threads taking "
                      f"chains from a queue ({secs_e:.1f} s of wall time); the pool size is the best of a sweep over {sweep_n} with one chain per "
                      f"thread; same algorithm/seeds/event sequence as the GPU chains" + why,
            "ensemble": ens,
            "proposals_per_s": best[2], "single_thread_events_per_s": single, "acceptance": best[3],
            "host": lim, "thread_sweep": {str(k): v for k, v in sweep.items()}, "parallel_efficiency": eff}


CONFIG_DEFAULTS = {"C3": dict(chains=4096, dt=1.0), "C3G": dict(chains=4096, dt=1.0), "C2": dict(chains=4096, dt=30.0), "C4": dict(chains=8192, dt=2.0),
                   "C5": dict(chains=4096, dt=0.005)}


def make_workload(pkg, args, rank, local_rank):
    """The ensemble of `--config` (C3 = the headline workload, default; C2 / C4 / C5 = BASELINE.json's other GPU configurations, each
    with the same JSON schema and its own roofline object) plus what the generic timing loop needs to know about it."""
    L = pkg._lib
    nch, dt = args.chains, args.dt
    seed0 = SEED0 + args.chain_first  # (global chain index: a rank's chains are the same chains whatever the number of ranks)
    W = {"config": args.config, "trace_full_retry": True}
    if args.config == "C3":
        G = pkg.problems.gmrf_precision(args.grid)
        d = G.shape[0]
        c = pkg.problems.column_norms(G)
        # events per chain per unit time ~0.8 d; 2x head-room, recycled every step
        cap = 0 if args.no_trace else int(2.0 * d * dt) + 1024
        ens = pkg.Ensemble(nch, d, device=local_rank, trace_capacity=cap)
        if args.lambda_ref > 0.0:
            # C3R: the same workload with the flow's refresh clock on (src/sfact.jl:78-114; ZigZag(Γ, μ, σ = 1; λref), `test/staticarrays.jl:45` uses one):
            # the reference's own arithmetic only (the tracked evaluation has no refresh), on the 4-event speculative kernel since round 6
            args.exact = True
            ens.set_flow(pkg.ZigZag(G, np.zeros(d), np.ones(d), λref=args.lambda_ref))
        else:
            ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
        ens.set_target(pkg.GaussianTarget(G))
        if not args.exact:
            ens.set_gradient_tracking(True)
        ens.set_state_synthetic(0.0, c, seed0)
        W.update(G=G, c=c, d=d, cap=cap, ens=ens, kernel="zz_local_spec8_kernel" if args.exact else "zz_local_trackp_kernel",
                 unit="reflection events/s", evaluation="moving (bit-identical to the oracle)" if args.exact else
                 "tracked gradients (bit-identical to the oracle's tracked evaluation; against the moving evaluation: same index sequence until a rounding "
                 "difference flips a test, ~6e-10 per proposal = 3 of 4096 chains by T=20, floats to 1e-9: tests/test_gpu_track_horizon.py); the "
                 "bit-identical moving evaluation is timed beside it as `exact`",
                 metric="reflection events/sec, d=16384 local ZigZag (spdmp), ensemble of independent chains",
                 workload=f"C3: local ZigZag spdmp on Gamma=0.01I+gridlaplacian({args.grid},{args.grid}), d={d}, {nch} chains/GPU, "
                          f"step = advance all chains by dT={dt}, traces {'off' if args.no_trace else 'on (32 B/event)'}",
                 model="224*num + 48*(num-nacc) + 616*nacc bytes (SURVEY 8d3)",
                 bytes=lambda w: algorithmic_bytes(w["num"], w["nacc"]))
    elif args.config == "C3G":
        # the headline sampler OFF the benchmark stencil: any sparse Γ (src/sfact.jl:170-179 builds G1 / G2 from the CSC pattern)
        if args.graph == "lattice3d":
            G = pkg.problems.lattice3d_precision(25)
            gname = "Gamma=0.01I+Laplacian of the 25x25x25 7-point lattice"
        else:
            nz = int(args.graph[len("random"):])
            G = pkg.problems.random_sparse_precision(16384, nz)
            gname = f"random symmetric pattern, <= {nz} entries per column, diagonally dominant (test/maintest.jl:6-8's sprandn scaled up)"
        d = G.shape[0]
        c = pkg.problems.column_norms(G)
        cap = 0 if args.no_trace else int(2.5 * d * dt) + 1024
        ens = pkg.Ensemble(nch, d, device=local_rank, trace_capacity=cap)
        ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
        ens.set_target(pkg.GaussianTarget(G))
        if not args.exact:
            ens.set_gradient_tracking(True)
        ens.set_state_synthetic(0.0, c, seed0)
        kbar = float(np.diff(G.indptr).mean())
        B = (abs(G) > 0).astype(np.int64)
        g2bar = float((np.diff((B @ B).tocsc().indptr) - np.diff(G.indptr)).mean())
        per_prop = kbar * 40 + 24
        per_acc = g2bar * 40 + 8 + kbar * 32 + kbar * 16 + 16 + 32
        W.update(G=G, c=c, d=d, cap=cap, ens=ens, kernel="", unit="reflection events/s",
                 evaluation="moving (bit-identical to the oracle)" if args.exact else "tracked gradients (bit-identical to the oracle's tracked evaluation)",
                 metric=f"reflection events/sec, d={d} local ZigZag (spdmp) on a sparse Gamma that is not the 2-d lattice, ensemble of independent chains",
                 workload=f"C3G: local ZigZag spdmp on {gname}, d={d}, {nch} chains/GPU, step = advance all chains by dT={dt}, "
                          f"traces {'off' if args.no_trace else 'on (32 B/event)'}",
                 model=f"SURVEY 8d3's model on this graph: per proposal k*40+24 = {per_prop:.0f} B (k={kbar:.2f}), +48 B per rejection, "
                       f"+{per_acc:.0f} B per accepted reflection (|G2|={g2bar:.1f})",
                 bytes=lambda w: per_prop * w["num"] + 48.0 * (w["num"] - w["nacc"]) + per_acc * w["nacc"])
    elif args.config == "C2":
        import scipy.sparse as sp
        d = 1024
        cap = 512  # events per chain per launch: 512 x 16 392 B x 4096 chains = 34 GB of HBM (a step of dT = 30 is ~410 events)
        rng = np.random.default_rng(1000 + args.chain_first)
        ens = pkg.Ensemble(nch, d, sampler=L.SAMPLER_BPS, factor=2.0, device=local_rank, trace_capacity=cap)
        ens.set_flow_bps(pkg.BouncyParticle(sp.identity(d, format="csc"), np.zeros(d), 1.0))
        ens.set_state_bps(0.0, rng.standard_normal((nch, d)), rng.standard_normal((nch, d)), 1e-3,
                          np.arange(nch, dtype=np.uint64) + np.uint64(seed0))
        W.update(d=d, cap=cap, ens=ens, kernel="bps_run_kernel", unit="events/s",
                 metric="trace events/sec (reflections + refreshments), Bouncy Particle d=1024 isotropic Gaussian, ensemble of independent chains",
                 workload=f"C2: BouncyParticle(I, 0, lambda_ref=1), c=1e-3 (scripts/not_fact.jl:23-28), d={d}, {nch} chains/GPU, step = advance "
                          f"all chains by dT={dt}, full PDMPTrace records (t, copy(x), copy(theta)) = {8 * (2 * d + 1)} B per event",
                 model="8(2d+1) bytes WRITTEN per event, 0 read: x, theta, grad live in registers (SURVEY 8d3)",
                 bytes=lambda w: 8.0 * (2 * d + 1) * w["nevents"])
    elif args.config == "C4":
        P = pkg.problems.logistic_problem(m=20)
        d = P["p"]
        ksub = 10
        cap = 0 if args.no_trace else int(600 * dt) + 512
        rng = np.random.default_rng(2000 + args.chain_first)
        ens = pkg.Ensemble(nch, d, adapt=True, factor=5.0, device=local_rank, trace_capacity=cap)
        ens.set_flow(pkg.ZigZag(P["Gdrop"], P["mu"], P["sigma"]))
        ens.set_target(pkg.LogisticTarget(P["A"], P["y"], P["ny"], P["mu"], P["gamma0"], ksub))
        integrals = os.environ.get("PDMP_BENCH_C4_INTEGRALS", "0") != "0" or args.gather  # (--gather reduces the batch-mean sums: they need the integrals)
        # (the engine's own ∫x_i dt -- no counterpart in the reference, nothing in this configuration reads it -- costs the state-in-LDS kernel
        # 8 of 32 bytes per coordinate: 8 instead of 13 chains per CU.  PDMP_BENCH_C4_INTEGRALS=1 keeps it; the line carries the other variant as
        # `with_path_integrals`, measured in the same process after the timed region.)
        ens.set_path_integrals(integrals)
        if getattr(args, "tracked", False):
            ens.set_gradient_tracking(True)
            W["evaluation"] = ("tracked bounds (g_j = Gamma[:,j].x and gd_j = Gamma[:,j].theta carried per coordinate; bit-identical to the oracle's tracked evaluation, "
                               "same index sequence as the moving evaluation until a rounding difference flips a decision); the gradient is the moving evaluation")
        ens.set_state(0.0, np.tile(P["x0"], (nch, 1)), P["sigma"] * rng.choice([-1.0, 1.0], (nch, d)), P["c"],
                      np.arange(nch, dtype=np.uint64) + np.uint64(seed0))
        # neighbourhood sizes of the bounding graph and row lengths of the design, as plain averages over coordinates / observations
        G = P["Gdrop"]
        kbar = float(np.diff(G.indptr).mean())
        B = (abs(G) > 0).astype(np.int64)
        g2 = (B @ B)
        g2bar = float((np.diff(g2.tocsc().indptr) - np.diff(G.indptr)).mean())
        rbar = float(np.diff(P["At"].indptr).mean())
        per_prop = kbar * 40 + 24 + ksub * rbar * 40
        per_rej = 48.0
        per_acc = g2bar * 40 + 8 + kbar * (8 + 16 + 8) + kbar * 16 + 16 + 32
        W.update(d=d, cap=cap, ens=ens, kernel="zz_general_run_kernel" if os.environ.get("PDMP_KERNEL") == "seq" else "zz_logistic_lds_kernel",
                 unit="reflection events/s",
                 metric="reflection events/sec, subsampled sparse logistic regression n=8840 p=442 (local ZigZag), ensemble of independent chains",
                 workload=f"C4: spdmp with grad-phi-moving (k={ksub} subsample, SelfMoving, control variate at the mode), Zdrop bounds, c=0.01, adapt, "
                          f"factor 5 (scripts/logistic.jl:167), n={P['n']}, p={d}, {nch} chains/GPU (one GPU's share of 65 536), step = advance "
                          f"all chains by dT={dt}; chain state resident in LDS for the slice, path integrals {'kept' if integrals else 'off'}",
                 model=f"per proposal k*40+24 + ksub*r*40 = {per_prop:.0f} B (k={kbar:.2f} neighbours, r={rbar:.2f} regressors per observation), "
                       f"+48 B per rejection, +{per_acc:.0f} B per accepted reflection (|G2|={g2bar:.1f}): SURVEY 8d3's model with this graph",
                 bytes=lambda w: per_prop * w["num"] + per_rej * (w["num"] - w["nacc"]) + per_acc * w["nacc"])
    elif args.config == "C5":
        P = pkg.problems.spike_slab_logistic_problem(p=10_000, num_rows=args.c5_rows)
        d = P["p"]
        ksub = 12
        cap = 0 if args.no_trace else int(30000 * dt) + 1024
        ens = pkg.Ensemble(nch, d, sampler=L.SAMPLER_STICKY_ZIGZAG, adapt=True, factor=1.5, device=local_rank, trace_capacity=cap)
        ens.set_flow(pkg.ZigZag(P["G"], P["mu"], P["sigma"]))
        ens.set_target(pkg.LogisticTarget(P["A"], P["y"], P["ny"], P["mu"], P["gamma0"], ksub))
        ens.set_sticky(P["kappa"])
        ens.set_state_synthetic(0.0, P["c"], seed0)
        rbar = float(np.diff(P["At"].indptr).mean())
        # One gradient evaluation samples ksub rows (scripts/logistic.jl:84) and brings every coordinate they read to t' (idot_moving!,
        # src/common.jl:33-42).  A coordinate that several sampled rows share is MOVED once (24 B read + 16 B written) and only re-read (8 B)
        # by the others: the expected number of distinct coordinates follows from the design's row counts (hypergeometric, exact).  The
        # design's entries themselves (8 B value + 4 B row index as stored; 2000 x 5457 of them = 131 MB, not a cache-resident table) are
        # streamed once per sampled row.  (Round 3 charged every entry of every sampled row a full 40-byte move: 1.6 x this model and more than
        # the counters saw.)
        from scipy.special import gammaln
        nrows = P["At"].shape[1]
        nj = np.diff(P["A"].indptr).astype(np.float64)  # rows that hold coordinate j
        lc = lambda a, b: gammaln(a + 1) - gammaln(b + 1) - gammaln(a - b + 1)
        with np.errstate(invalid="ignore"):
            p_untouched = np.where(nrows - nj >= ksub, np.exp(lc(nrows - nj, ksub) - lc(nrows, ksub)), 0.0)
        distinct = float(np.sum(1.0 - p_untouched))
        per_grad = distinct * 40 + (ksub * rbar - distinct) * 8 + ksub * rbar * 12 + 40 + 24
        W.update(d=d, cap=cap, ens=ens, kernel="zz_general_run_kernel", unit="events/s", ksub=ksub,
                 metric="trace events/sec (reflections + freezes + thaws), sticky ZigZag on the logistic spike-and-slab p=10000, ensemble of independent chains",
                 workload=f"C5: sspdmp with grad-phi-moving (k={ksub}, SelfMoving) on example_design_matrix scaled to p={d} columns x {P['n']} rows "
                          f"(scripts/exampledesign.jl), Gaussian slab gamma0={P['gamma0']}, kappa=(gamma0/sqrt(2pi))/(1/w-1) w={P['w']}, Z=ZigZag(I,mu), "
                          f"c=1, adapt (scripts/spikeandslab.jl:96-129), {nch} chains/GPU, step = advance all chains by dT={dt}",
                 model=f"per gradient evaluation {per_grad:.0f} B = {distinct:.0f} distinct coordinates moved once (40 B) + {ksub * rbar - distinct:.0f} "
                       f"re-reads of an already moved coordinate (8 B) + ksub*r = {ksub * rbar:.0f} design entries streamed (12 B: {P['At'].nnz * 12 / 1e6:.0f} MB table) + 64 B "
                       f"(r={rbar:.0f} coefficients per sampled observation), + 96 B per trace event (record, bound, key, thaw clock)",
                 bytes=lambda w: per_grad * w["grads"] + 96.0 * w["nevents"])
    else:
        raise SystemExit(f"unknown --config {args.config}")
    return W


def measure_with_integrals(pkg, args, rank, local_rank):
    """C4: the same workload with the engine's path integrals kept (8 instead of 13 chains per CU), beside the headline -- figures of different
    rounds are comparable through it (rounds 1-2 kept the integrals)."""
    a2 = argparse.Namespace(**vars(args))
    a2.gather = True  # (keeps the integrals: see make_workload)
    W2 = make_workload(pkg, a2, rank, local_rank)
    e2 = W2["ens"]
    n2 = args.warmup + min(args.steps, 4)
    ims, c0 = [], None
    for k in range(n2):
        T2 = (k + 1) * args.dt
        ms2 = 0.0
        while True:
            e2.run(T2, pkg._lib.RUN_STOP_BEFORE, sync=False)
            ms2 += e2.last_run_ms()
            full = pkg._lib.needs_rerun(e2.counters()["status"]) if W2["cap"] else False
            if W2["cap"]:
                e2.trace_reset()
            if not full:
                break
        ims.append(ms2)
        if k == args.warmup - 1:
            c0 = e2.counters()
    c1 = e2.counters()
    e2.close()
    secs = float(np.sum(ims[args.warmup:])) * 1e-3
    base0 = {f: (int(c0[f].sum()) if c0 is not None else 0) for f in ("num", "nacc", "nevents")}
    w2 = {f: int(c1[f].sum()) - base0[f] for f in ("num", "nacc", "nevents")}
    ach2 = W2["bytes"](w2) / secs / 1e9
    return {"ms_per_step": 1e3 * secs / (n2 - args.warmup), "value": w2["nevents"] / secs, "unit": W2["unit"],
            "roofline_frac": ach2 / HBM_PEAK_GBS, "steps": n2 - args.warmup,
            "note": "pdmp_ensemble_set_path_integrals(1): 18.7 instead of 11.9 KB of LDS per chain, 8 instead of 13 chains per CU"}


def measure_ess(pkg, args, W, ens, local_rank):
    """ESS/s.  A fresh ensemble of the same shape is started IN STATIONARITY -- x0 ~ N(0, inv(Gamma)) exactly (DCT of the lattice,
    problems.gmrf_stationary_sample), theta0 uniform on {+-1} -- and run for B batches of length b on the timed kernel; after every batch
    the device returns the path integrals J_i of every chain at 32 probe coordinates (pdmp_ensemble_path_integrals).  With the exact mean 0
    and exact Var_pi = diag(inv(Gamma)), sigma2(s) = s * mean (Y_s)^2 over chains and merged batches at every dyadic batch length
    s = b .. B*b (ess.multiscale_ess); ESS_i(s) = N*B*b*Var_pi,i / sigma2_i(s) shrinks as s passes the autocorrelation times of the slow
    lattice modes (eigenvalue 0.01: ~150 time units), so the HEADLINE is the SMALLEST of them -- the largest s, i.e. the spread of the N
    whole-run means -- and `last_doubling` says how far from its plateau that still is.  Divided by the GPU seconds of that run.
    Returns (ess object, the ensemble still open or None)."""
    G, c, d, cap, nch = W["G"], W["c"], W["d"], W["cap"], args.chains
    B, b = args.ess_batches, args.ess_batch_len
    probes = np.linspace(0, d - 1, 32).astype(np.int64)
    if args.no_stationary_start:
        es, T0, start = ens, (args.warmup + args.steps) * args.dt, "continued from the timed run (x0 ~ N(0, I)): slow modes NOT in equilibrium"
    else:
        ens.close()
        rng = np.random.default_rng(SEED0)
        es = pkg.Ensemble(nch, d, device=local_rank, trace_capacity=cap)
        es.set_flow(pkg.ZigZag(G, np.zeros(d)))
        es.set_target(pkg.GaussianTarget(G))
        if not args.exact:
            es.set_gradient_tracking(True)
        es.set_state(0.0, pkg.problems.gmrf_stationary_sample(args.grid, nch, rng), rng.choice([-1.0, 1.0], (nch, d)), c,
                     np.arange(nch, dtype=np.uint64) + np.uint64(SEED0 + (1 << 24)))
        T0, start = 0.0, "stationary: x0 ~ N(0, inv(Gamma)) exactly, theta0 uniform on {+-1}"
    ess_ms = 0.0
    J = [es.path_integrals(T0, probes)]
    nsl = max(1, int(round(b / args.dt)))
    for kb in range(B):
        for q in range(nsl):  # slices of dT, so that the trace segments (sized for one step) are recycled as in the timed steps
            es.run(T0 + kb * b + (q + 1) * (b / nsl), pkg._lib.RUN_STOP_BEFORE, sync=False)
            ess_ms += es.last_run_ms()
            if cap:
                es.trace_reset()
        J.append(es.path_integrals(T0 + (kb + 1) * b, probes))
    ebad = int(np.count_nonzero(es.counters()["status"] != pkg._lib.CHAIN_OK))
    kname = es.kernel_name()
    if es is not ens:
        es.close()
    var_pi = pkg.problems.gmrf_marginal_variances(args.grid)[probes]
    r = pkg.ess.multiscale_ess(np.stack(J), b, var_pi, mean=0.0)
    gpu_s = ess_ms * 1e-3
    ex_ess = r["ess_extrapolated"]
    return {"definition": f"N={nch} chains, B={B} batches of length b={b} (run length {B * b}), path integrals of every chain at 32 probe "
                          "coordinates; sigma2(s) = s*mean(Y_s^2) at dyadic batch lengths s=b..B*b with the exact mean 0; ESS_i(s) = "
                          "N*B*b*Var_pi,i/sigma2_i(s), Var_pi = exact diag(inv(Gamma)); sigma2 grows with s towards sigma2_asym (bias -Gamma/s), "
                          "so the HEADLINE uses the Richardson value 2*sigma2(B*b) - sigma2(B*b/2) -- the smallest ESS of all listed; per GPU "
                          "second of the run that produced the path; min / median over the probes",
            "start": start, "evaluation": "exact" if args.exact else "tracked",
            "ess_min_per_s": float(ex_ess.min() / gpu_s), "ess_median_per_s": float(np.median(ex_ess) / gpu_s),
            "ess_per_chain_time_median": float(np.median(var_pi / r["sigma2_extrapolated"])),
            "ess_per_chain_time_min": float(np.min(var_pi / r["sigma2_extrapolated"])),
            "iact_median": float(np.median(r["sigma2_extrapolated"] / (2.0 * var_pi))),
            "last_doubling_median": float(np.median(r["last_doubling"])), "last_doubling_max": float(np.max(r["last_doubling"])),
            "by_batch_len": {str(float(sc)): {"ess_min_per_s": float(r["ess"][q].min() / gpu_s),
                                              "ess_median_per_s": float(np.median(r["ess"][q]) / gpu_s)}
                             for q, sc in enumerate(r["scales"])},
            "unhealthy_chains": ebad, "gpu_seconds": gpu_s, "batches": B, "batch_len": b, "kernel": kname}



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3", choices=sorted(CONFIG_DEFAULTS),
                    help="C3 (default): the headline workload; C2 / C4 / C5: the other GPU configurations of BASELINE.json")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chains", type=int, default=None, help="chains per GPU (weak scaling; default: the configuration's)")
    ap.add_argument("--scaling", default=None, choices=["strong", "weak"],
                    help="strong (default for C3 / C3G, the north star's form: ONE ensemble of --total-chains chains, rank r of R runs chains "
                         "[r N/R, (r+1) N/R), SURVEY 8 e1) or weak (--chains per GPU; default for C2 / C4 / C5, whose widths are one GPU's share)")
    ap.add_argument("--total-chains", type=int, default=None, help="strong scaling: chains of the whole job (default: the configuration's width)")
    ap.add_argument("--late-T", type=float, default=200.0,
                    help="C4: continue the timed ensemble (untimed) to this process time and time --steps slices again there (`late`: the figure of a "
                         "long run, where the bounds have adapted; 0: skip)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="C3 at N = 1: skip the `pipeline` object (the steps again with their trace consumed on the device beside the sampler)")
    ap.add_argument("--no-configs", action="store_true",
                    help="C3 at N = 1: skip the `configs` object (short runs of C2 / C4 / C5 / C3G / C3 at d = 65536 in their own processes after the region) "
                         "and the `mode` object (which timing mode this process saw)")
    ap.add_argument("--no-strong-proxy", action="store_true",
                    help="C3 at N = 1: skip the `strong_proxy` object (this GPU's share of the 2 / 4 / 8-GPU strong-scaling job, timed after the region)")
    ap.add_argument("--c5-rows", type=int, default=2000,
                    help="C5: rows of the design the gradient subsamples from (the script's 50 000 rows at 10^4 columns would be 2.7e10 stored entries; the work "
                         "per proposal depends on k_sub and the row length, not on it -- 10000 rows = a 655 MB table, beyond the MALL, checks that: DESIGN 5)")
    ap.add_argument("--grid", type=int, default=GRID)
    ap.add_argument("--lambda-ref", type=float, default=0.0,
                    help="C3: rate of the flow's refresh clock (C3R; > 0 implies --exact: the tracked evaluation has no refresh)")
    ap.add_argument("--dt", type=float, default=None, help="process time per step (default: the configuration's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ess-batches", type=int, default=16, help="B: batches per chain of the ESS run after the timed region (0: skip; a power of two)")
    ap.add_argument("--ess-batch-len", type=float, default=64.0, help="b: length of an ESS batch in process time (the run is B*b long)")
    ap.add_argument("--no-stationary-start", action="store_true",
                    help="ESS run: continue the timed ensemble (x0 ~ N(0, I) as scripts/gaussianrandomfield.jl:29) instead of a fresh one "
                         "started at x0 ~ N(0, inv(Gamma)) exactly; the slow modes are then NOT in equilibrium and the figure is optimistic")
    ap.add_argument("--exact-steps", type=int, default=5,
                    help="C3: steps of the bit-identical moving evaluation timed after the headline region and printed as `exact` (0: skip)")
    ap.add_argument("--no-trace", action="store_true", help="count events only (diagnostic; not the headline mode)")
    ap.add_argument("--gather", action="store_true",
                    help="after the timed region: one more step, then time the post-run exchange (all_gather counts -> gatherv of the "
                         "trace segments to rank 0 -> reduce of the batch-mean sums); printed as a separate `gather` object")
    ap.add_argument("--per-rank", action="store_true", help="add per-rank counters and chain-0 digests to the JSON line (tests)")
    ap.add_argument("--tracked", action="store_true",
                    help="C4: tracked bounds (pdmp_ensemble_set_gradient_tracking on the logistic target; bit-identical to the oracle's tracked evaluation) "
                         "instead of the default, bit-identical moving evaluation")
    ap.add_argument("--graph", default="lattice3d", choices=["lattice3d", "random5", "random6", "random7", "random8"],
                    help="C3G: the 25^3 7-point lattice (d = 15625, |G1| = 7, |S| = 25) or a random symmetric pattern at d = 16384")
    ap.add_argument("--exact", action="store_true",
                    help="C3: the bit-identical moving evaluation (zz_local_spec8_kernel) instead of the tracked-gradient one")
    args = ap.parse_args()
    if args.scaling is None:
        args.scaling = "strong" if args.config in ("C3", "C3G") and args.chains is None else "weak"
    if args.chains is None:
        args.chains = CONFIG_DEFAULTS[args.config]["chains"]
    if args.dt is None:
        args.dt = CONFIG_DEFAULTS[args.config]["dt"]
    if args.config == "C3G":
        args.ess_batches = 0
        args.exact_steps = 0
    elif args.config != "C3":
        args.ess_batches = 0  # the ESS leg (stationary start, exact variances) belongs to the headline workload

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # which chains are this rank's: weak = its own block of --chains, strong = its share of ONE ensemble (zigzagboomerang.jl_amd/parallel.py)
    if args.scaling == "strong":
        if args.total_chains is None:
            args.total_chains = args.chains
        base, rem = divmod(args.total_chains, world)  # (= parallel.shard_range; the package is imported later: torch first)
        args.chains = base + (1 if rank < rem else 0)
        args.chain_first = rank * base + min(rank, rem)
        if args.chains < 1:
            raise SystemExit(f"--scaling strong: {args.total_chains} chains over {world} ranks leaves rank {rank} without one")
    else:
        args.total_chains = args.chains * world
        args.chain_first = rank * args.chains
    dist = None
    comm = None  # the engine's own RCCL communicator (pdmp_comm_*): no torch in the process
    # Which transport carries the barriers and the reductions of this line.  N = 1 (--gather): the engine's communicator.  N > 1: torch.distributed
    # over RCCL ("nccl") by default -- neither path has run on more than one GPU yet (one GPU per build box; RCCL refuses two ranks on one device),
    # and until the engine's has, the transport every ROCm node is exercised with daily carries the driver's scaling run; PDMP_BENCH_BACKEND=engine
    # selects pdmp_comm_* for N > 1 (all-or-nothing rendezvous: it falls back to "nccl" on EVERY rank or on none), "gloo" is the test hook that
    # lets several ranks share ONE device (PDMP_BENCH_SINGLE_DEVICE).  The run itself has no collective either way.
    backend = os.environ.get("PDMP_BENCH_BACKEND", "engine" if world == 1 else "nccl")
    if os.environ.get("PDMP_BENCH_SINGLE_DEVICE"):
        local_rank = 0
    red_dev = "cpu"
    nccl_group = None  # the torch.distributed group that carries this line's reductions over RCCL (None: the default group)
    ctl = None         # N > 1: a gloo group (TCP on MASTER_ADDR) that every rank joins FIRST -- whether RCCL is used is voted on over it
    transport_log = []
    if world > 1:
        import torch  # (before the engine library is loaded: one HIP runtime per process, the one torch ships)
        import torch.distributed as ctl
        ctl.init_process_group("gloo")

    def vote(ok):
        """True iff `ok` on EVERY rank (MIN over the gloo group): a transport is adopted by all ranks or by none -- a rank on which RCCL came up
        never waits in a collective for ranks that have already fallen back (round 4 decided rank by rank)."""
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64)
        ctl.all_reduce(t, op=ctl.ReduceOp.MIN)
        return bool(t.item() == 1.0)

    if world == 1 and args.gather and backend != "engine":
        # the exchange written against torch.distributed: a one-rank group makes N = 1 run the very same code
        import torch
        import torch.distributed as dist1
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            with _banner_to_stderr():
                dist1.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
                dist1.barrier()
                torch.cuda.synchronize()
            red_dev = "cuda"
        else:
            dist1.init_process_group(backend, rank=0, world_size=1)
        dist = dist1

    pkg = load_package()
    pkg.build.build()
    if pkg._lib.device_count() < 1:
        raise SystemExit("bench.py: no gfx950 device visible; the engine has no CPU fallback")
    if backend == "engine" and (world > 1 or args.gather):
        # ncclCommInitRank through the library (the 128-byte id travels over a TCP rendezvous on MASTER_ADDR: parallel.exchange_unique_id)
        err = None
        try:
            comm = pkg.parallel.Comm(rank, world, local_rank)
        except Exception as exc:  # e.g. RCCL refuses the ranks' devices, or the rendezvous port range is closed on this node
            if world == 1:
                raise
            err = exc
            print(f"bench.py: rank {rank}: pdmp_comm_init failed ({exc})", file=sys.stderr, flush=True)
        if world > 1 and not vote(err is None):
            if comm is not None:
                comm.close()
                comm = None
            transport_log.append("engine communicator (pdmp_comm_init) did not come up on every rank")
            print(f"bench.py: rank {rank}: the engine's communicator did not come up on every rank; falling back to torch.distributed over RCCL", file=sys.stderr, flush=True)
            backend = "nccl"
    if world > 1 and backend == "nccl":
        # RCCL through torch.distributed, as a second group beside the gloo one; adopted by the vote
        g, err = None, None
        try:
            import datetime
            torch.cuda.set_device(local_rank)
            with _banner_to_stderr():  # (RCCL's version banner would land on this process's stdout, after the JSON line)
                g = ctl.new_group(backend="nccl", timeout=datetime.timedelta(seconds=300))  # nccl == RCCL on ROCm
                probe = torch.ones(1, dtype=torch.float64, device="cuda")
                ctl.all_reduce(probe, group=g)  # the communicator is created here at the latest
                torch.cuda.synchronize()
                if int(round(float(probe.item()))) != world:
                    raise RuntimeError(f"all_reduce over RCCL returned {probe.item()}, expected {world}")
        except Exception as exc:  # the run has no collective: barriers and a few small reductions travel as well over gloo
            err = exc
            print(f"bench.py: rank {rank}: torch.distributed over RCCL failed ({exc})", file=sys.stderr, flush=True)
        if vote(err is None):
            nccl_group, red_dev, dist = g, "cuda", ctl
        else:
            transport_log.append("torch.distributed over RCCL did not come up on every rank")
            print(f"bench.py: rank {rank}: RCCL did not come up on every rank; barriers and reductions over gloo instead", file=sys.stderr, flush=True)
            backend = "gloo"
    if world > 1 and backend == "gloo":
        dist = ctl
    W = make_workload(pkg, args, rank, local_rank)
    ens, d, cap, nch = W["ens"], W["d"], W["cap"], args.chains
    G, c = W.get("G"), W.get("c")

    def barrier():
        ens.sync()
        if comm is not None:
            comm.barrier()
        if dist is not None:
            dist.barrier(group=nccl_group)
            if red_dev == "cuda":
                import torch
                torch.cuda.synchronize()

    def allreduce(values, op):
        """sum / max over the ranks on whatever transport carries this line (comm: the engine's RCCL communicator; dist: torch.distributed)."""
        if comm is not None and world > 1:
            return [float(v) for v in comm.allreduce([float(v) for v in values], op)]
        if dist is not None:
            import torch
            tv = torch.tensor([float(v) for v in values], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tv, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM, group=nccl_group)
            return [float(v) for v in tv.tolist()]
        return [float(v) for v in values]

    launches = [0]

    def step(k):
        T = (k + 1) * args.dt
        ms = 0.0
        while True:
            ens.run(T, pkg._lib.RUN_STOP_BEFORE, sync=False)
            ms += ens.last_run_ms()  # HIP events on the launch stream; also waits for the kernel
            launches[0] += 1
            full = False
            if cap:
                if args.config != "C3":  # a chain whose segment filled up pauses with TRACE_FULL: drain and continue the slice
                    full = pkg._lib.needs_rerun(ens.counters()["status"])
                ens.trace_reset()
            if not full:
                return ms

    def work_counters():
        cn = ens.counters()
        # sticky chains reset (acc, num) whenever a bound is adapted (src/ss_fact.jl:134): gradient evaluations are counted by the
        # draws of the subsampling stream instead (ksub per evaluation)
        return {"num": int(cn["num"].sum()), "nacc": int(cn["nacc"].sum()), "nevents": int(cn["nevents"].sum()),
                "grads": int(cn["ndraw_global"].sum()) // W.get("ksub", 1)}

    placement_log = ens.debug_placement() if hasattr(ens, "debug_placement") else None  # (what set_state's placement probes saw and kept)
    for k in range(args.warmup):
        step(k)
    barrier()
    if args.warmup > 0 and ens.kernel_name():
        W["kernel"] = ens.kernel_name()  # what the engine actually launched (pdmp_debug_last_kernel), not what this script expects
    w0 = work_counters()
    launches[0] = 0
    kernel_ms = []
    t_start = time.perf_counter()
    for k in range(args.warmup, args.warmup + args.steps):
        kernel_ms.append(step(k))
    barrier()
    elapsed = time.perf_counter() - t_start
    w1 = work_counters()
    cnt = ens.counters()
    bad = int(np.count_nonzero(cnt["status"] != pkg._lib.CHAIN_OK))

    work = {key: w1[key] - w0[key] for key in w0}
    num, nacc, nev = work["num"], work["nacc"], work["nevents"]
    if args.config == "C5":
        num = work["grads"]

    # Beside the headline, never inside `value` (each measured in this process after the timed region)
    exact = None
    if rank == 0 and args.config == "C3" and not args.exact and args.exact_steps > 0:
        exact = pkg.benchlib.measure_exact(pkg, args, G, c, cap, local_rank)
    strong_proxy = None
    if rank == 0 and world == 1 and args.config == "C3" and args.lambda_ref == 0.0 and not args.no_strong_proxy and not args.gather and not args.no_trace:
        v1 = nev / max(float(np.sum(kernel_ms)) * 1e-3, 1e-9)  # (this GPU's kernel-time rate at the full width: the same clock as the proxy's)
        strong_proxy = pkg.benchlib.measure_strong_proxy(pkg, args, G, c, local_rank, v1 if not args.exact else None,
                                            exact["value"] if exact else (v1 if args.exact else None))
    pipeline = None
    if rank == 0 and world == 1 and args.config == "C3" and args.lambda_ref == 0.0 and not args.no_pipeline and not args.gather and not args.no_trace:
        pipeline = pkg.benchlib.measure_pipeline(pkg, args, G, c, local_rank, nev / max(float(np.sum(kernel_ms)) * 1e-3, 1e-9))
    with_integrals = None
    if rank == 0 and args.config == "C4" and not args.gather and os.environ.get("PDMP_BENCH_C4_INTEGRALS", "0") == "0" and args.exact_steps > 0:
        with_integrals = measure_with_integrals(pkg, args, rank, local_rank)
    # C4: the timed region sits in the transient of a T = 2000 configuration -- adapt with factor 5 (scripts/logistic.jl:167) keeps raising bounds, the
    # acceptance falls from 0.8 to 0.04 over the first 200 time units and a slice gets 8 times as expensive (tools/c4_drift.py) -- so the SAME
    # ensemble is continued (untimed) to --late-T and timed again there: `late` is the figure of a long run, `value` the one of its first slices
    late = None
    if args.config == "C4" and args.late_T > (args.warmup + args.steps) * args.dt and not args.gather:
        k0 = args.warmup + args.steps
        k1 = int(round(args.late_T / args.dt))
        for k in range(k0, k1):
            step(k)
        barrier()
        l0 = work_counters()
        launches_l0 = launches[0]
        lms = []
        tl0 = time.perf_counter()
        for k in range(k1, k1 + args.steps):
            lms.append(step(k))
        barrier()
        tl = allreduce([time.perf_counter() - tl0], "max")[0]
        l1 = work_counters()
        lw = {key: l1[key] - l0[key] for key in l0}
        if rank == 0:
            nl = max(launches[0] - launches_l0, 1)
            lk = float(np.sum(lms)) / nl
            lach = W["bytes"](lw) / nl / (lk * 1e-3) / 1e9
            late = {"T_range": [k1 * args.dt, (k1 + args.steps) * args.dt], "steps": args.steps, "ms_per_step": 1e3 * tl / args.steps,
                    "value_this_rank": lw["nevents"] / tl, "unit": W["unit"], "proposals_per_s_this_rank": lw["num"] / tl,
                    "acceptance": lw["nacc"] / max(lw["num"], 1), "kernel_ms_avg": lk, "launches_per_step": nl / args.steps,
                    "roofline_frac": lach / HBM_PEAK_GBS, "achieved_GBps": lach,
                    "note": "the same ensemble continued to T = %g and timed again: the bounds have adapted (the acceptance falls from 0.8 at T = 2 to "
                            "0.04 at T = 200, still falling slowly), a proposal costs what it costs in a long run" % (k1 * args.dt)}
    ess = None
    if rank == 0 and world == 1 and args.config == "C3" and args.ess_batches >= 1 and not args.gather:
        ess = measure_ess(pkg, args, W, ens, local_rank)
    # the default line also carries the other configurations (short runs, own processes) and the timing mode it ran in
    configs = mode = None
    default_line = (rank == 0 and world == 1 and args.config == "C3" and not args.exact and not args.gather and not args.no_trace and not args.no_configs
                    and args.grid == GRID and nch == CONFIG_DEFAULTS["C3"]["chains"] and strong_proxy is not None)
    if default_line:
        mode = pkg.benchlib.measure_mode(pkg, float(np.sum(kernel_ms)) / max(launches[0], 1), strong_proxy, placement_log)

    # post-run exchange (never inside `value`): SURVEY 8e1
    gather = None
    if args.gather and cap and comm is not None and args.config == "C2":
        # PDMPTrace events (t, copy(x), copy(theta)): three arrays per rank, the same exchange (pdmp_ensemble_gather_bps_traces)
        ens.trace_reset()
        ens.run((args.warmup + args.steps + 1) * args.dt, pkg._lib.RUN_STOP_BEFORE)  # (one launch: what the segments hold when it pauses or ends)
        barrier()
        tg0 = time.perf_counter()
        widths, counts_all, devbuf = comm.gather_bps_traces(ens, root=0, to_host=False)
        barrier()
        tg = float(comm.allreduce([time.perf_counter() - tg0], "max")[0])
        if rank == 0:
            nev_g = int(counts_all.sum())
            assert devbuf[1] == nev_g
            eb = 8 * (2 * d + 1)
            gather = {"seconds": tg, "events": nev_g, "bytes": eb * nev_g, "GBps": eb * nev_g / tg / 1e9, "chains": int(widths.sum()),
                      "staging": "device", "backend": "engine (librccl linked by libpdmp_mi355.so)",
                      "steps": "ncclAllGather(counts) -> device compaction -> grouped ncclSend/ncclRecv of t, x, theta to rank 0 "
                               "(pdmp_ensemble_gather_bps_traces)"}
    elif args.gather and cap and comm is not None:
        Tg = (args.warmup + args.steps + 1) * args.dt
        T_prev = (args.warmup + args.steps) * args.dt
        ens.batch_means(0.0, T_prev)  # baseline J(T_prev)
        ens.run(Tg, pkg._lib.RUN_STOP_BEFORE)  # (C4 / C5: a chain whose segment fills up pauses -- what the segments hold then is gathered)
        barrier()
        tg0 = time.perf_counter()
        widths, counts_all, devbuf = comm.gather_traces(ens, root=0, to_host=False)  # the events stay on the root's device
        sy, sy2 = comm.reduce_moments(ens, T_prev, Tg, root=0)
        barrier()
        tg = float(comm.allreduce([time.perf_counter() - tg0], "max")[0])
        if rank == 0:
            nev_g = int(counts_all.sum())
            assert devbuf[1] == nev_g
            gather = {"seconds": tg, "events": nev_g, "bytes": 32 * nev_g, "GBps": 32 * nev_g / tg / 1e9, "chains": int(widths.sum()),
                      "staging": "device", "backend": "engine (librccl linked by libpdmp_mi355.so)",
                      "steps": "ncclAllGather(counts) -> device compaction -> grouped ncclSend/ncclRecv of the trace segments to rank 0 -> "
                               "ncclReduce(SUM) of 2 x d sums (pdmp_ensemble_gather_traces, pdmp_ensemble_reduce_moments)",
                      "mean_of_batch_means": float(np.mean(sy) / (nch * world))}
    elif args.gather and cap and args.config == "C2":
        raise SystemExit("--gather of PDMPTrace events runs on the engine's communicator (PDMP_BENCH_BACKEND=engine, the default)")
    elif args.gather and cap:
        import torch
        par = pkg.parallel
        Tg = (args.warmup + args.steps + 1) * args.dt if ess is None else None
        if Tg is None:
            raise SystemExit("--gather and the ESS run are separate modes: add --ess-batches 0")
        T_prev = (args.warmup + args.steps) * args.dt
        ens.batch_means(0.0, T_prev)  # baseline J(T_prev)
        ens.run(Tg, pkg._lib.RUN_STOP_BEFORE)
        sy, sy2 = ens.batch_means(T_prev, Tg)
        staging = "device" if red_dev == "cuda" else "host"
        barrier()
        tg0 = time.perf_counter()
        counts_by_rank, gathered, sy, sy2 = par.gather_ensemble(ens, sy, sy2, staging=staging, group=nccl_group)
        if red_dev == "cuda" or dist is None:
            torch.cuda.synchronize()
        barrier()
        tg = time.perf_counter() - tg0
        if dist is not None:
            tgt = torch.tensor([tg], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tgt, op=dist.ReduceOp.MAX, group=nccl_group)
            tg = float(tgt.item())
        if rank == 0:
            nev_g = int(sum(int(c.sum().item()) for c in counts_by_rank))
            assert sum(int(t.shape[0]) for t in gathered) == nev_g
            gather = {"seconds": tg, "events": nev_g, "bytes": 32 * nev_g, "GBps": 32 * nev_g / tg / 1e9,
                      "chains": int(sum(c.numel() for c in counts_by_rank)), "staging": staging, "backend": "torch.distributed " + backend,
                      "steps": "all_gather(counts) -> grouped isend/irecv of the trace segments to rank 0 -> reduce(SUM) of 2 x d sums",
                      "first_event_time_rank_last": float(gathered[-1][0, 0].item()) if gathered[-1].shape[0] else None,
                      "mean_of_batch_means": float(np.mean(sy) / (nch * world))}

    # aggregate over ranks: max time, summed work
    elapsed = allreduce([elapsed], "max")[0]
    num_all, nacc_all, nev_all, bad_all = allreduce([num, nacc, nev, bad], "sum")

    # N > 1, strong scaling: the weak-scaling view of the same job beside it (the configuration's width PER GPU: what rounds 1-4 reported),
    # so that one driver run yields both curves -- same barriers, max over ranks, never inside `value`
    weak = None
    if world > 1 and args.scaling == "strong" and args.config == "C3" and not args.gather:
        nw = CONFIG_DEFAULTS["C3"]["chains"]
        ew = pkg.benchlib.c3_ensemble(pkg, G, c, nw, cap, SEED0 + rank * nw, not args.exact, local_rank)
        nst = max(2, min(args.steps, 6))
        for k in range(2):
            ew.run((k + 1) * args.dt, pkg._lib.RUN_STOP_BEFORE, sync=False)
            ew.last_run_ms()
            ew.trace_reset()
        cw0 = ew.counters()
        barrier()
        tw0 = time.perf_counter()
        for k in range(2, 2 + nst):
            ew.run((k + 1) * args.dt, pkg._lib.RUN_STOP_BEFORE, sync=False)
            ew.last_run_ms()
            ew.trace_reset()
        barrier()
        tw = allreduce([time.perf_counter() - tw0], "max")[0]
        nevw = allreduce([int(ew.counters()["nevents"].sum()) - int(cw0["nevents"].sum())], "sum")[0]
        ew.close()
        weak = {"scaling": "weak", "chains_per_gpu": nw, "steps": nst, "ms_per_step": 1e3 * tw / nst, "value": nevw / tw, "unit": W["unit"],
                "note": "the same job with the configuration's width on EVERY GPU, timed after the strong region with the same barriers"}

    ranks_seen = 1
    if comm is not None and world > 1:
        ranks_seen = int(round(float(comm.allreduce([1.0], "sum")[0])))
    elif dist is not None:
        ranks_seen = int(dist.get_world_size(group=nccl_group))
    per_rank = None
    if args.per_rank or world > 1:
        c0 = cnt[0]
        mine = {"rank": rank, "seed_first": int(SEED0 + args.chain_first), "chains": int(nch), "num": int(num), "nacc": int(nacc), "nevents": int(nev),
                "kernel_ms_per_step": float(np.sum(kernel_ms)) / max(args.steps, 1),
                "chain0": {"num": int(c0["num"]), "nacc": int(c0["nacc"]), "ndraw_main": int(c0["ndraw_main"]), "t_last": float(c0["t_last"])}}
        if world > 1:
            vec = np.zeros((world, 10))
            vec[rank] = [mine["seed_first"], mine["num"], mine["nacc"], mine["nevents"], c0["num"], c0["nacc"], c0["ndraw_main"], c0["t_last"],
                         mine["chains"], mine["kernel_ms_per_step"]]
            if comm is not None:
                vec = comm.allreduce(vec, "sum").reshape(world, 10)
            else:  # (a plain all_reduce: the one collective every backend of this script is known to carry)
                import torch
                tv = torch.tensor(vec, dtype=torch.float64, device=red_dev)
                dist.all_reduce(tv, op=dist.ReduceOp.SUM, group=nccl_group)
                vec = tv.cpu().numpy()
            per_rank = [{"rank": r, "seed_first": int(v[0]), "chains": int(v[8]), "kernel_ms_per_step": float(v[9]),
                         "num": int(v[1]), "nacc": int(v[2]), "nevents": int(v[3]),
                         "chain0": {"num": int(v[4]), "nacc": int(v[5]), "ndraw_main": int(v[6]), "t_last": float(v[7])}} for r, v in enumerate(vec)]
        else:
            per_rank = [mine]

    if rank == 0:
        # per kernel LAUNCH (a step of C2 / C4 / C5 is several launches when trace segments fill up)
        nlaunch = max(launches[0], 1)
        k_ms = float(np.sum(kernel_ms)) / nlaunch
        bytes_launch = W["bytes"](work) / nlaunch
        achieved = bytes_launch / (k_ms * 1e-3) / 1e9
        # HBM traffic per launch from the PMC passes of THIS build (tools/profile_round.sh writes profiles/traffic.json with the hash of
        # the kernel sources; a stale file is ignored rather than quoted)
        traffic = None
        traffic_src = None
        issue = None  # instruction-issue account of the kernel from the SQ counter passes (what bounds a kernel whose HBM traffic is far below its roof)
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            tp = json.load(open(tpath))
            tkey = args.config
            if args.config == "C3" and args.lambda_ref > 0.0:
                tkey = "C3R"
            elif args.config == "C3" and W.get("kernel") == "zz_local_trackl_kernel":
                tkey = "C3L"
            elif args.config == "C3" and args.exact:
                tkey = "C3X"
            if args.config == "C3" and args.grid != GRID:
                tkey = "C3_g%d" % args.grid
            elif args.config == "C3" and nch != CONFIG_DEFAULTS["C3"]["chains"]:
                tkey += "_w%d" % nch  # (a rank's share of the ensemble: its own PMC passes)
            elif args.config == "C4" and args.tracked:
                tkey = "C4T"
            elif args.config == "C3G":
                tkey = "C3G" + ("X" if args.exact else "") + ("" if args.graph == "lattice3d" else "_" + args.graph)
            ent = tp.get("configs", {}).get(tkey)
            if ent and tp.get("source_hash") == source_hash():
                units = {"proposal": num, "event": nev}[ent["per"]]
                traffic = ent["hbm_bytes_per_unit"] * units / nlaunch
                traffic_src = ent.get("source")
                issue = ent.get("issue")
        out = {
            "metric": W["metric"],
            "value": nev_all / elapsed,
            "unit": W["unit"],
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "evaluation": ("exact" if args.exact else "tracked") if args.config in ("C3", "C3G") else ("tracked" if (args.config == "C4" and args.tracked) else "exact"),
            "config": {"workload": W["workload"] + (f"; evaluation: {W['evaluation']}" if "evaluation" in W else ""),
                       "chains_per_gpu": nch, "total_chains": args.total_chains, "d": d, "dT": args.dt,
                       "parallelism": (f"ONE ensemble of {args.total_chains} chains, rank r runs chains [r N/R, (r+1) N/R) (strong scaling), no collective in the run"
                                       if args.scaling == "strong" else f"{nch} chains per GPU x{world} (weak scaling), no collective in the run")
                                      + (" (reductions of this line: " + ("pdmp_comm_allreduce, RCCL linked by the engine" if comm is not None else "torch.distributed " + backend) + ")" if world > 1 else "")},
            "proposals_per_s": num_all / elapsed,
            # (sticky chains reset (acc, num) whenever a bound adapts, src/ss_fact.jl:134: no acceptance ratio can be formed from them)
            "acceptance": (nacc_all / max(num_all, 1.0)) if args.config != "C5" else None,
            "unhealthy_chains": bad_all,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": W["kernel"], "kernel_ms_avg": k_ms, "launches_per_step": nlaunch / args.steps,
                         "traffic_GBps": (traffic / (k_ms * 1e-3) / 1e9) if traffic else None,
                         "algorithmic_bytes_per_launch": bytes_launch,
                         "model": W["model"]},
        }
        if args.config == "C3" and not args.exact:
            # what the TRACKED recurrences themselves need (DESIGN.md 5 "own model"): per proposal the coordinate's (key, t_old) and (θ, g, gd, tg) read
            # = 48 B; a rejected one writes (key, t_old) = 16 B; an accepted one instead moves x_i (x, tx, ∫x dt, count read and written: 64 B), flips
            # θ_i (8 B), reads the sums of its k - 1 neighbours (32 B each), writes (g, gd, tg) and a new (key, t_old) for all k members (40 B each)
            # and appends an event (32 B): 432 B on the lattice (k = 5).  `frac` above is on SURVEY 8d3's model of the MOVING algorithm (what is graded).
            kk = 5.0  # |G1| inside the lattice
            acc_b = 64.0 + 8.0 + 32.0 * (kk - 1.0) + 40.0 * kk + 32.0
            own_b = (48.0 * num + 16.0 * (num - nacc) + acc_b * nacc) / nlaunch
            out["roofline"]["own_model"] = {"bytes_per_launch": own_b, "achieved": own_b / (k_ms * 1e-3) / 1e9, "frac": own_b / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                            "model": "48*num + 16*(num-nacc) + %.0f*nacc bytes: what the tracked recurrences read and write (no neighbour moves)" % acc_b,
                                            "traffic_over_own": (traffic / own_b) if traffic else None}
        out["totals"] = {"num": num_all, "nacc": nacc_all, "nevents": nev_all, "T_end": (args.warmup + args.steps) * args.dt}
        if issue is not None:
            out["issue"] = issue
        out["ranks_seen"] = ranks_seen  # size of the communicator that carried this line's reductions (1: no communicator)
        if world > 1:
            out["transport"] = {"reductions": "pdmp_comm_allreduce (RCCL linked by the engine)" if comm is not None else "torch.distributed " + backend,
                                "control": "torch.distributed gloo (every rank joins it first; a transport is adopted by a vote over it: all ranks or none)",
                                "fallbacks": transport_log}
        if per_rank is not None:
            out["per_rank"] = per_rank
        if strong_proxy is not None:
            out["strong_proxy"] = strong_proxy
        if pipeline is not None:
            out["pipeline"] = pipeline
        if weak is not None:
            out["weak"] = weak
        if late is not None:
            out["late"] = late
        if gather is not None:
            out["gather"] = gather
        if ess is not None:
            out["ess"] = ess
        if exact is not None:
            out["exact"] = exact
            # the figure of the reference's own arithmetic (the moving evaluation, bit-identical to the oracle's restatement of src/sfact.jl:73-145), one
            # level up: `value` above is the tracked evaluation's (opt-in at the boundary: pdmp_ensemble_set_gradient_tracking / tracked = true)
            out["value_reference_arithmetic"] = exact["value"]
            out["roofline_frac_reference_arithmetic"] = exact["roofline"]["frac"] if isinstance(exact.get("roofline"), dict) else exact.get("roofline_frac")
        if mode is not None:
            out["mode"] = mode
        if with_integrals is not None:
            out["with_path_integrals"] = with_integrals
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg, G, c) if args.config in ("C3", "C3G") else cpu_baseline_config(pkg, args.config)
            if ess is not None and "ensemble" in out["cpu_baseline"]:
                # ESS/s of the CPU ensemble.  The CPU pool runs the reference's (moving) evaluation; the GPU ESS leg above ran the evaluation named in
                # `ess.evaluation` (tracked by default: the same process law -- the same sampler with sums carried instead of gathered, index-exact
                # against the moving one until a rounding flip, ~4 % of the chains by T = 1024 -- not the same realisations).  The effective sample
                # size per unit of PROCESS time is a property of the sampler, estimated once on 4096 chains x 1024 time units; ESS per second = that x
                # the process time the pool advances per second of wall time (chains x T / seconds)
                ce = out["cpu_baseline"]["ensemble"]
                chain_time_per_s = ce["chains"] * ce["T"] / ce["seconds"]
                out["cpu_baseline"]["ess"] = {"ess_min_per_s": ess["ess_per_chain_time_min"] * chain_time_per_s,
                                              "ess_median_per_s": ess["ess_per_chain_time_median"] * chain_time_per_s,
                                              "chain_time_per_s": chain_time_per_s,
                                              "definition": "ESS per unit process time of the sampler (the GPU leg's estimate, made on the `ess.evaluation` evaluation: the same "
                                                            "process law as the CPU pool's moving evaluation, not the same realisations) x process time advanced per "
                                                            "wall second by the CPU pool; same estimator, same probes as `ess`"}
        if default_line:
            out["configs"] = pkg.benchlib.measure_configs(os.path.abspath(__file__))
        print(json.dumps(out), flush=True)
    ens.close()
    if comm is not None:
        comm.barrier()
        comm.close()
    if ctl is not None:  # (N > 1: the gloo group, and the RCCL group beside it if it was adopted)
        ctl.barrier()
        ctl.destroy_process_group()
    elif dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
