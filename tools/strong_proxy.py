#!/usr/bin/env python
"""The per-GPU term of the north star's STRONG-scaling curve, measured on one GPU: config C3 with 4096 / N chains for N = 1, 2, 4, 8 (the run
has no collective, so a rank of an N-GPU job does exactly this), both evaluations, optionally with the phase profile of chain 0.

    python tools/strong_proxy.py [--phase] [--widths 4096,2048,1024,512] [--steps 8]
Prints one JSON line per (evaluation, width)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--widths", default="4096,2048,1024,512")
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--phase", action="store_true")
ap.add_argument("--evals", default="tracked,exact")
ap.add_argument("--grid", type=int, default=128)
args = ap.parse_args()

pkg = load_package()
G = pkg.problems.gmrf_precision(args.grid)
d = G.shape[0]
c = pkg.problems.column_norms(G)
for ev in args.evals.split(","):
    for nch in [int(w) for w in args.widths.split(",")]:
        ens = pkg.Ensemble(nch, d, trace_capacity=2 * d + 1024)
        ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
        ens.set_target(pkg.GaussianTarget(G))
        if ev == "tracked":
            ens.set_gradient_tracking(True)
        ens.set_state_synthetic(0.0, c, 0x5EED0000)
        ms = []
        for k in range(args.warmup + args.steps):
            if k == args.warmup:
                c0 = ens.counters()
            ens.run(float(k + 1), pkg._lib.RUN_STOP_BEFORE, sync=False)
            ms.append(ens.last_run_ms())
            ens.trace_reset()
        c1 = ens.counters()
        secs = float(np.sum(ms[args.warmup:])) * 1e-3
        nacc = int(c1["nacc"].sum()) - int(c0["nacc"].sum())
        num = int(c1["num"].sum()) - int(c0["num"].sum())
        out = {"evaluation": ev, "chains": nch, "d": d, "kernel": ens.kernel_name(), "ms_per_step": 1e3 * secs / args.steps,
               "events_per_s": nacc / secs, "proposals_per_s": num / secs, "events_per_s_per_chain": nacc / secs / nch,
               "waves_per_simd": nch / 1024.0, "bad": int(np.count_nonzero(c1["status"] != pkg._lib.CHAIN_OK))}
        if args.phase:  # one more step on the profiling instantiation (chain 0's cycle counters)
            ens.debug_phase_profile(True)
            ens.run(float(args.warmup + args.steps + 1), pkg._lib.RUN_STOP_BEFORE, sync=False)
            out["phase_ms"] = ens.last_run_ms()
            c0, c1 = c1, ens.counters()
            kind, ph = ens.debug_phase_cycles()
            it = max(ph[10], 1.0)
            out["phase"] = {"kind": kind, "iters": ph[10], "cycles_per_iter": [round(ph[q] / it) for q in range(10)],
                            "proposals_per_iter": (int(c1["num"][0]) - int(c0["num"][0])) / it,
                            "rounds": ph[14] / it, "guess_cycles": ph[15] / it, "cand_selected": ph[11] / it, "cand_after_zone": ph[12] / it, "cand_eval": ph[13] / it}
        print(json.dumps(out), flush=True)
        ens.close()
