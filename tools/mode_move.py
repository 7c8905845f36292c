#!/usr/bin/env python
"""Which ARRAY's placement carries the full-width slice's timing mode?  One C3 ensemble; each array in turn is copied into newly allocated memory several
times (pdmp_debug_move_buffer), a few slices timed after every move.      python tools/mode_move.py [--moves 6] [--arrays 0,1,2,3,4]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--moves", type=int, default=6)
ap.add_argument("--arrays", default="0,1,2,3,4")
ap.add_argument("--chains", type=int, default=4096)
args = ap.parse_args()
NAMES = ["records", "pairs", "trace", "headers", "constants", "keys"]
pkg = load_package()
G = pkg.problems.gmrf_precision(128)
d = G.shape[0]
ens = pkg.Ensemble(args.chains, d, trace_capacity=2 * d + 1024)
ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
ens.set_target(pkg.GaussianTarget(G))
ens.set_gradient_tracking(True)
ens.set_state_synthetic(0.0, pkg.problems.column_norms(G), 0x5EED0000)
T = [0.0]


def slices(n=3):
    ms = []
    c0 = ens.counters()
    for _ in range(n):
        T[0] += 1.0
        ens.run(T[0], pkg._lib.RUN_STOP_BEFORE, sync=False)
        ms.append(ens.last_run_ms())
        ens.trace_reset()
    c1 = ens.counters()
    return (round(float(np.mean(ms[1:])), 2), round((int(c1["nacc"].sum()) - int(c0["nacc"].sum())) / n / 1e6, 2), int(np.count_nonzero(c1["status"] != pkg._lib.CHAIN_OK)))


slices(2)
print(json.dumps({"start": slices()}), flush=True)
for a in [int(v) for v in args.arrays.split(",")]:
    seq = []
    for _ in range(args.moves):
        ens.debug_move_buffer(a)
        seq.append(slices())
    print(json.dumps({"moved": NAMES[a], "ms_Mevents_bad_after_each_move": seq}), flush=True)
ens.close()
