"""Measurements bench.py makes AFTER its timed region (none of them inside `value`): the other configurations of BASELINE.json as short runs,
and what tells a slow-mode box from a regression.  Kept here so that bench.py stays the contract and its timing loop."""
import json
import os
import subprocess
import sys
import time

import numpy as np

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (bench.py's figure)
SEED0 = 0x5EED0000     # bench.py's first seed: chain k of an ensemble runs seed SEED0 + k


# (name, bench.py arguments): short runs -- a few steps each, no CPU baseline, no side measurements
SECONDARY = [
    ("C2", ["--config", "C2", "--steps", "3", "--warmup", "1"]),
    ("C4", ["--config", "C4", "--steps", "3", "--warmup", "1"]),  # carries `late` (the same ensemble at T = 200)
    ("C3G_random6_tracked", ["--config", "C3G", "--graph", "random6", "--steps", "3", "--warmup", "1"]),
    ("C3G_random6_exact", ["--config", "C3G", "--graph", "random6", "--exact", "--steps", "2", "--warmup", "1"]),
    ("C3R_lambda1_exact", ["--config", "C3", "--lambda-ref", "1.0", "--steps", "2", "--warmup", "1"]),
    ("C3_grid256_w1024", ["--config", "C3", "--grid", "256", "--chains", "1024", "--steps", "2", "--warmup", "1"]),
    # (last, and with 5000 design rows: a 327 MB table, beyond the 256 MB memory-side cache like the 655 MB one of --c5-rows 10000, whose
    # generation -- an n x n solve per Newton step on the host -- takes 90 s, more than this object's budget)
    ("C5_rows5000", ["--config", "C5", "--c5-rows", "5000", "--steps", "3", "--warmup", "1"]),
]
QUIET = ["--no-cpu-baseline", "--ess-batches", "0", "--exact-steps", "0", "--no-strong-proxy", "--no-pipeline"]


def measure_configs(bench_path, budget_s=115.0, per_run_s=60.0):
    """One bench.py process per secondary configuration (BASELINE.json `configs`; the same JSON contract, a few steps): value, ms per step,
    roofline fraction and the kernel the engine launched.  Runs until the budget is spent; what did not fit is named."""
    out, t0 = {}, time.perf_counter()
    for name, argv in SECONDARY:
        left = budget_s - (time.perf_counter() - t0)
        if left < 8.0:
            out[name] = {"skipped": "budget of %.0f s spent" % budget_s}
            continue
        t1 = time.perf_counter()
        try:
            r = subprocess.run([sys.executable, bench_path] + argv + QUIET, capture_output=True, text=True, timeout=min(per_run_s, left))
            lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
            if r.returncode != 0 or not lines:
                out[name] = {"error": (r.stderr or r.stdout)[-300:], "wall_s": time.perf_counter() - t1}
                continue
            j = json.loads(lines[-1])
            e = {"value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "frac": j["roofline"]["frac"],
                 "kernel": j["roofline"]["kernel"], "steps": j["steps"], "chains_per_gpu": j["config"]["chains_per_gpu"], "d": j["config"]["d"],
                 "evaluation": j.get("evaluation"), "unhealthy_chains": j.get("unhealthy_chains"), "wall_s": time.perf_counter() - t1}
            if j.get("late"):
                e["late"] = {k: j["late"][k] for k in ("T_range", "ms_per_step", "value_this_rank", "roofline_frac", "acceptance")}
            out[name] = e
        except subprocess.TimeoutExpired:
            out[name] = {"error": "timed out", "wall_s": time.perf_counter() - t1}
    out["wall_s"] = time.perf_counter() - t0
    out["what"] = ("short runs of the other configurations in their own processes after the timed region (3 or 2 steps each: figures to orient by, "
                   "`python bench.py --config ..` is the measurement)")
    return out


def measure_mode(pkg, full_width_kernel_ms, strong_proxy, placement=None):
    """Which timing mode did this process run in (the same binary runs the full-width C3 slice in ~38.5 or in ~45.5 ms -- by the
    ALLOCATION that backs the records and the pairs, round 6 -- while 2048 chains and the memory probes do not move)?  The full-width kernel time,
    this GPU's 2048-chain slice from the strong proxy, the random-line probes (128-byte lines read, and read + written back, by 4096 x 64 lanes),
    and what set_state's placement probes saw and kept (`placement`: pdmp_debug_placement), all from this process."""
    r = {}
    for write, name in ((0, "read"), (1, "read_write")):
        ms = min(pkg._lib.sector_probe(4096, 16384, 400, write) for _ in range(2))
        n = 4096 * 64 * 4 * 400 * (1 if write == 0 else 2)
        r[name + "_TBps_of_128B_lines"] = n * 128 / (ms * 1e-3) / 1e12
    w2048 = None
    if strong_proxy:
        for e in strong_proxy.get("by_gpus", []):
            if e.get("chains_per_gpu") == 2048 and e.get("tracked"):
                w2048 = e["tracked"]["ms_per_step"]
    label = "fast" if full_width_kernel_ms < 41.0 else ("slow" if full_width_kernel_ms > 43.5 else "between")
    return {"full_width_kernel_ms": full_width_kernel_ms, "label": label, "chains_2048_ms": w2048, "random_lines": r, "placement": placement,
            "rule": "zz_local_trackp_kernel at 4096 chains x d = 16384: < 41 ms fast, > 43.5 ms slow (observed: 38.2-39.0 and 44.7-46.5); the mode belongs to the "
                    "allocations behind the records and the pairs: set_state times a short launch, re-allocates and keeps the fastest (placement: the probes in "
                    "ms and what was kept; PDMP_PLACE_TUNE=0 turns it off and the mode is the driver's coin again); the 2048-chain slice (21.3-22.8 ms) and the "
                    "line probes (7.7 / 5.5 TB/s) do not see it"}

# ---- the C3 side measurements of bench.py (moved here in round 6: bench.py keeps the contract, the workloads and the timed loop)
def algorithmic_bytes(num, nacc):
    """224 B per proposal + 48 B per rejection + 616 B per accepted reflection."""
    return 224.0 * num + 48.0 * (num - nacc) + 616.0 * nacc


def c3_ensemble(pkg, G, c, nch, cap, seed0, tracked, device=0):
    """A C3 / C3G ensemble of `nch` chains with the seeds seed0 + chain on either evaluation."""
    d = G.shape[0]
    e = pkg.Ensemble(nch, d, device=device, trace_capacity=cap)
    e.set_flow(pkg.ZigZag(G, np.zeros(d)))
    e.set_target(pkg.GaussianTarget(G))
    if tracked:
        e.set_gradient_tracking(True)
    e.set_state_synthetic(0.0, c, seed0)
    return e


def timed_slices(pkg, e, dt, nwarm, nsteps, cap):
    """nwarm + nsteps slices of dT on ensemble e: (seconds of the kernels of the last nsteps [HIP events], counter differences over them)."""
    ms, c0 = [], None
    for k in range(nwarm + nsteps):
        if k == nwarm:
            c0 = e.counters()
        e.run((k + 1) * dt, pkg._lib.RUN_STOP_BEFORE, sync=False)
        ms.append(e.last_run_ms())
        if cap:
            e.trace_reset()
    c1 = e.counters()
    secs = float(np.sum(ms[nwarm:])) * 1e-3
    work = {f: int(c1[f].sum()) - int(c0[f].sum()) for f in ("num", "nacc", "nevents")}
    return secs, work, int(np.count_nonzero(c1["status"] != pkg._lib.CHAIN_OK))


def measure_exact(pkg, args, G, c, cap, local_rank):
    """The bit-identical (moving) evaluation beside the headline (never inside `value`): same workload, seeds and step on zz_local_spec8_kernel."""
    ex = c3_ensemble(pkg, G, c, args.chains, cap, SEED0 + args.chain_first, False, local_rank)
    xs, w, bad = timed_slices(pkg, ex, args.dt, 2, args.exact_steps, cap)
    kname = ex.kernel_name()
    ex.close()
    xach = algorithmic_bytes(w["num"], w["nacc"]) / xs / 1e9
    return {"kernel": kname, "evaluation": "moving: bit-identical to the oracle (indices, outcomes, times, positions)",
            "steps": args.exact_steps, "ms_per_step": 1e3 * xs / args.exact_steps, "value": w["nacc"] / xs, "unit": "reflection events/s",
            "proposals_per_s": w["num"] / xs, "roofline": {"bound": "hbm", "achieved": xach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                           "frac": xach / HBM_PEAK_GBS},
            "unhealthy_chains": bad,
            "note": "kernel time from HIP events of this process, after the timed region; same seeds, step and trace handling"}


def measure_strong_proxy(pkg, args, G, c, local_rank, value_1gpu, exact_1gpu):
    """The per-GPU term of the north star's strong-scaling curve, measured on THIS GPU: rank 0's share of the --total-chains ensemble on a job of
    R = 2, 4, 8 GPUs is chains [0, N/R) -- the run has no collective and the tables are replicated, so one GPU running N/R chains IS what every
    rank of that job does (src/sfact.jl:199-208 is the loop each chain runs).  Projected job rate = R x this GPU's rate at N/R chains;
    efficiency = that / (R x the 1-GPU rate)."""
    rows = []
    d = G.shape[0]
    for R in (2, 4, 8):
        n = args.total_chains // R
        if n < 1:
            continue
        row = {"gpus": R, "chains_per_gpu": n, "waves_per_simd": n / 1024.0}
        for name, tracked, ref in (("tracked", True, value_1gpu), ("exact", False, exact_1gpu)):
            if ref is None:
                continue
            cap = int(2.0 * d * args.dt) + 1024
            e = c3_ensemble(pkg, G, c, n, cap, SEED0, tracked, local_rank)
            secs, w, bad = timed_slices(pkg, e, args.dt, 2, max(2, min(args.steps, 6)), cap)
            kname = e.kernel_name()
            e.close()
            ach = algorithmic_bytes(w["num"], w["nacc"]) / secs / 1e9
            row[name] = {"kernel": kname, "ms_per_step": 1e3 * secs / max(2, min(args.steps, 6)), "events_per_s": w["nacc"] / secs,
                         "proposals_per_s": w["num"] / secs, "events_per_s_per_chain": w["nacc"] / secs / n,
                         "roofline_frac": ach / HBM_PEAK_GBS, "projected_job_events_per_s": R * w["nacc"] / secs,
                         "projected_efficiency": (R * w["nacc"] / secs) / (R * ref), "unhealthy_chains": bad}
        rows.append(row)
    return {"what": f"this GPU running the share N/R of the {args.total_chains}-chain ensemble that a rank of an R-GPU job runs (no collective in the run: "
                    "the per-GPU term of the strong-scaling curve, measured, not modelled); efficiency = rate at N/R chains / rate at N chains",
            "by_gpus": rows}


def measure_pipeline(pkg, args, G, c, local_rank, ref_rate):
    """The measured steps again with their trace CONSUMED (src/sfact.jl:211 returns Ξ; discretize and mean are what callers run over it next:
    src/trace.jl:106-125,182-200, test/maintest.jl:28-29): after every slice pdmp_ensemble_consume_async applies the slice's events to the
    per-(chain, coordinate) cursors -- streaming mean(Ξ) and collect(discretize(Ξ, 0.5)) -- on the ensemble's second stream, two trace buffers, no
    trace_reset.  Both orders are timed: the consumer BETWEEN the slices (the library's choice for an ensemble that fills the device) and BESIDE
    the next slice.  End-to-end = events / wall time of the loop; beside it the consumer alone and what the host could drain over PCIe instead."""
    d = G.shape[0]
    nch, dt = args.chains, args.dt
    nw, ns = 2, max(2, min(args.steps, 6))
    grid_dt = 0.5
    K = int(round((nw + ns + 2) * dt / grid_dt)) + 2
    cap = int(2.0 * d * dt) + 1024
    modes = {}
    alone_ms, alone_ev, drain, chk = [], [], None, None
    for name, mode in (("between_slices", 0), ("beside_next_slice", 1)):
        e = c3_ensemble(pkg, G, c, nch, cap, SEED0 + args.chain_first, not args.exact, local_rank)
        e.debug_set_consumer_overlap(mode)
        e.consume_begin(grid_dt, K)
        for k in range(nw):
            e.run((k + 1) * dt, pkg._lib.RUN_STOP_BEFORE, sync=False)
            e.consume_async()
        e.sync()
        c0 = e.counters()
        run_ms = []
        t0 = time.perf_counter()
        for k in range(nw, nw + ns):
            e.run((k + 1) * dt, pkg._lib.RUN_STOP_BEFORE, sync=False)
            e.consume_async()                 # (returns at once)
            run_ms.append(e.last_run_ms())    # (waits for slice k only)
        e.sync()
        wall = time.perf_counter() - t0
        c1 = e.counters()
        nev = int(c1["nevents"].sum()) - int(c0["nevents"].sum())
        modes[name] = {"ms_per_step": 1e3 * wall / ns, "sampler_kernel_ms_per_step": float(np.mean(run_ms)), "end_to_end_events_per_s": nev / wall,
                       "fraction_of_timed_value": (nev / wall) / ref_rate if ref_rate else None,
                       "unhealthy_chains": int(np.count_nonzero(c1["status"] != pkg._lib.CHAIN_OK))}
        if mode == 0:
            # the consumer alone: two more slices, each consumed with nothing beside it
            for k in range(nw + ns, nw + ns + 2):
                n0 = int(e.counters()["nevents"].sum())
                e.run((k + 1) * dt, pkg._lib.RUN_STOP_BEFORE)
                e.sync()
                e.consume_async()
                alone_ms.append(e.last_consume_ms())
                alone_ev.append(int(e.counters()["nevents"].sum()) - n0)
            drain = e.debug_host_drain_gbps(1 << 30)
            m, Tl = e.consume_mean(0, 1)
            chk = {"chain0_mean_abs_max": float(np.max(np.abs(m[0]))), "chain0_T_last": float(Tl[0])}
        e.close()
    cons_bytes = np.mean(alone_ev) * (32 + 2 * 32) + nch * d * (dt / grid_dt) * (32 + 8.0)  # events + cursor read / write; per grid row: cursor read + point
    cons_s = float(np.mean(alone_ms)) * 1e-3
    # The sampler ALONE, measured in the same minutes: its kernel time in the between-slices loop, where no kernel runs beside it (the timed value of
    # the line may be minutes old, and the full-width launch has two timing modes that come and go: DESIGN.md 5)
    alone_adjacent_ms = modes["between_slices"]["sampler_kernel_ms_per_step"]
    for m_ in modes.values():
        m_["fraction_of_sampler_alone"] = alone_adjacent_ms / m_["ms_per_step"]
    best = max(modes, key=lambda k: modes[k]["end_to_end_events_per_s"])
    out = {"what": "the timed steps with their trace consumed on the device (streaming mean + discretize at dt = 0.5 over every chain and coordinate) by "
                   "pdmp_ensemble_consume_async: a second stream, two trace buffers, no trace_reset",
           "steps": ns, "order": best, "library_default_order": "between_slices" if nch > 2048 else "beside_next_slice",
           "end_to_end_events_per_s": modes[best]["end_to_end_events_per_s"], "fraction_of_sampler_alone": modes[best]["fraction_of_sampler_alone"],
           "fraction_of_timed_value": modes[best]["fraction_of_timed_value"], "sampler_alone_ms_per_step": alone_adjacent_ms,
           "fraction_note": "fraction_of_sampler_alone = the sampler's kernel time per step in the between-slices loop (nothing runs beside it there) / this loop's "
                            "wall time per step; fraction_of_timed_value compares with the line's `value`, measured minutes earlier and possibly in the other timing mode",
           "ms_per_step": modes[best]["ms_per_step"], "by_order": modes,
           "consumer_alone": {"ms_per_step": float(np.mean(alone_ms)), "events_per_s": float(np.mean(alone_ev)) / cons_s,
                              "algorithmic_bytes_per_step": cons_bytes, "GBps": cons_bytes / cons_s / 1e9,
                              "roofline_frac": cons_bytes / cons_s / 1e9 / HBM_PEAK_GBS,
                              "model": "32 B per event + 32 B cursor read + 32 B cursor written; per grid row (d x chains x dT / 0.5 per step) a cursor read + 8 B"},
           "trace_GBps_at_this_rate": 32.0 * modes[best]["end_to_end_events_per_s"] / 1e9, "host_drain_GBps": drain,
           "host_drain_note": "device -> pinned host copy of 1 GiB of the trace buffer: what draining Ξ over PCIe instead would be limited to"}
    out.update(chk or {})
    return out
