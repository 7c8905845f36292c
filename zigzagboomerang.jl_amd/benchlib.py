"""Measurements bench.py makes AFTER its timed region (none of them inside `value`): the other configurations of BASELINE.json as short runs,
and what tells a slow-mode box from a regression.  Kept here so that bench.py stays the contract and its timing loop."""
import json
import os
import subprocess
import sys
import time

HBM_PEAK_GBS = 8000.0

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


def measure_mode(pkg, full_width_kernel_ms, strong_proxy):
    """Which timing mode did this process run in (the same binary runs the full-width C3 slice in ~38.5 or in ~45.5 ms, by box and by
    the minute, while 2048 chains and the memory probes do not move)?  The full-width kernel time, this GPU's 2048-chain slice from the strong
    proxy and the random-line probes (128-byte lines read, and read + written back, by 4096 x 64 lanes), all from this process."""
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
    return {"full_width_kernel_ms": full_width_kernel_ms, "label": label, "chains_2048_ms": w2048, "random_lines": r,
            "rule": "zz_local_trackp_kernel at 4096 chains x d = 16384: < 41 ms fast, > 43.5 ms slow (observed: 38.2-39.0 and 44.7-46.5); the 2048-chain slice "
                    "(21.3-22.8 ms) and the line probes (7.7 / 5.5 TB/s) are the same in both modes -- if THEY move, it is not the mode"}
