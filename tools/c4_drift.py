#!/usr/bin/env python
"""Config C4: kernel time per slice (dT = 2) and acceptance as the bounds adapt, T = 0 .. Tmax: where does the transient end?
    python tools/c4_drift.py [Tmax] [chains]"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
Tmax = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
P = pkg.problems.logistic_problem(m=20)
d = P["p"]
dt = 2.0
ens = pkg.Ensemble(nch, d, adapt=True, factor=5.0, trace_capacity=int(600 * dt) + 512)
ens.set_flow(pkg.ZigZag(P["Gdrop"], P["mu"], P["sigma"]))
ens.set_target(pkg.LogisticTarget(P["A"], P["y"], P["ny"], P["mu"], P["gamma0"], 10))
ens.set_path_integrals(False)
rng = np.random.default_rng(2000)
ens.set_state(0.0, np.tile(P["x0"], (nch, 1)), P["sigma"] * rng.choice([-1.0, 1.0], (nch, d)), P["c"], np.arange(nch, dtype=np.uint64) + np.uint64(0x5EED0000))
k = 0
prev = ens.counters()
while (k + 1) * dt <= Tmax:
    ms = 0.0
    while True:
        ens.run((k + 1) * dt, pkg._lib.RUN_STOP_BEFORE, sync=False)
        ms += ens.last_run_ms()
        cn = ens.counters()
        full = bool(np.any(cn["status"] == pkg._lib.CHAIN_TRACE_FULL))
        ens.trace_reset()
        if not full:
            break
    num = int(cn["num"].sum()) - int(prev["num"].sum()); acc = int(cn["nacc"].sum()) - int(prev["nacc"].sum())
    prev = cn
    if k % 5 == 4 or k < 4:
        print(json.dumps({"T": (k + 1) * dt, "ms": round(ms, 2), "proposals": num, "acceptance": round(acc / max(num, 1), 4),
                          "ns_per_proposal_per_chain": round(1e6 * ms / max(num, 1) * nch, 1)}), flush=True)
    k += 1
