#!/usr/bin/env python
"""Secondary measurement: config C4 of BASELINE.json -- sparse logistic regression n = 8840, p = 442 (scripts/logistic.jl),
local ZigZag with the subsampled gradient ∇ϕmoving (k = 10, control variate at the mode, SelfMoving), Zdrop bounds,
c = 0.01, adapt = true, factor = 5; one GPU's share (8192 chains) of the 65 536-chain ensemble.  Prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
nch = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
P = pkg.problems.logistic_problem(m=20)
p = P["p"]
rng = np.random.default_rng(1)
X0 = np.tile(P["x0"], (nch, 1))
TH0 = P["sigma"] * rng.choice([-1.0, 1.0], (nch, p))
ens = pkg.Ensemble(nch, p, adapt=True, factor=5.0, trace_capacity=0)
ens.set_flow(pkg.ZigZag(P["Gdrop"], P["mu"], P["sigma"]))
ens.set_target(pkg.LogisticTarget(P["A"], P["y"], P["ny"], P["mu"], P["gamma0"], 10))
ens.set_state(0.0, X0, TH0, P["c"], np.arange(nch, dtype=np.uint64) + 0x5EED0000)
ens.run(2.0, pkg._lib.RUN_STOP_BEFORE)  # warm-up slice (bounds adapt)
t0 = ens.totals()
ens.run(2.0 + T, pkg._lib.RUN_STOP_BEFORE, sync=False)
ms = ens.last_run_ms()
t1 = ens.totals()
cnt = ens.counters()
print(json.dumps({"config": f"C4: logistic n=8840 p=442, subsampled grad k=10 (SelfMoving), {nch} chains, dT={T}",
                  "kernel_ms": ms, "proposals_per_s": (t1["num"] - t0["num"]) / (ms * 1e-3),
                  "events_per_s": (t1["nevents"] - t0["nevents"]) / (ms * 1e-3),
                  "acceptance": (t1["nacc"] - t0["nacc"]) / max(t1["num"] - t0["num"], 1),
                  "unhealthy_chains": int(np.count_nonzero(cnt["status"] != 0))}))
ens.close()
