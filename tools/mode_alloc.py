#!/usr/bin/env python
"""Is the full-width slice's timing mode a property of the PROCESS or of the ALLOCATION?  Inside one process: create the C3 ensemble, time a few
slices, close it -- several times; then the same keeping every earlier ensemble alive (so each new one gets other physical memory).
    python tools/mode_alloc.py [--rounds 5] [--steps 6]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--chains", type=int, default=4096)
ap.add_argument("--exact", action="store_true", help="the moving (bit-identical) evaluation instead of the tracked one")
ap.add_argument("--caps", default="33792", help="trace capacities (events per chain) to try in turn, in the keep-all phase")
args = ap.parse_args()
pkg = load_package()
G = pkg.problems.gmrf_precision(128)
d = G.shape[0]
c = pkg.problems.column_norms(G)


def one(keep, cap=2 * d + 1024):
    ens = pkg.Ensemble(args.chains, d, trace_capacity=cap)
    ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
    ens.set_target(pkg.GaussianTarget(G))
    if not args.exact:
        ens.set_gradient_tracking(True)
    ens.set_state_synthetic(0.0, c, 0x5EED0000)
    ms = []
    global last_addr, last_ev
    last_ev = []
    for k in range(2 + args.steps):
        n0 = int(ens.counters()["nacc"].sum())
        ens.run(float(k + 1), pkg._lib.RUN_STOP_BEFORE, sync=False)
        ms.append(ens.last_run_ms())
        cn = ens.counters()
        last_ev.append((round((int(cn["nacc"].sum()) - n0) / 1e6, 2), int(np.count_nonzero(cn["status"] != pkg._lib.CHAIN_OK)), round(float(cn["t_last"].min()), 3)))
        ens.trace_reset()
    last_addr = {k: hex(v) for k, v in ens.debug_buffer_addresses().items()}
    last_addr["placement"] = ens.debug_placement()
    if keep is None:
        ens.close()
    else:
        keep.append(ens)
    return float(np.mean(ms[2:])), [round(v, 1) for v in ms[2:]]


for phase, keep, cap in [("close_each", None, 2 * d + 1024)] + [("keep_all", [], int(v)) for v in args.caps.split(",")]:
    for r in range(args.rounds):
        m, all_ms = one(keep, cap)
        print(json.dumps({"phase": phase, "cap": cap, "round": r, "ms": round(m, 2), "ev": last_addr["ev"], "kp": last_addr["kp"], "placement": last_addr["placement"], "slices": all_ms, "Mevents": last_ev[2:]}), flush=True)
    if keep:
        # the kept ensembles again, in order: is the mode a property of the allocation?
        for j, ens in enumerate(keep if args.steps > 100 else []):
            ms = []
            for k in range(4):
                ens.run(float(2 + args.steps + k + 1), pkg._lib.RUN_STOP_BEFORE, sync=False)
                ms.append(ens.last_run_ms())
                ens.trace_reset()
            print(json.dumps({"phase": "kept_again", "ensemble": j, "slices": [round(v, 1) for v in ms]}), flush=True)
        for ens in keep:
            ens.close()
