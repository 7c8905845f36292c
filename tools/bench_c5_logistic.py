#!/usr/bin/env python
"""Secondary measurement: the logistic version of config C5 -- sticky ZigZag (sspdmp) on the spike-and-slab sparse logistic regression
of scripts/sticky/sticky_logistic_sparse.jl (n = 8840, p = 442, subsampled ∇ϕmoving with k = 10, κ = (γ0/√2π)/(1/w − 1), w = 1/2,
stock ZigZag bound on the dropped Hessian, adapt = true, factor 5); prints one JSON line."""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
nch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
P = pkg.problems.logistic_problem(m=20)
p = P["p"]
rng = np.random.default_rng(1)
X0 = np.tile(P["x0"], (nch, 1))
TH0 = P["sigma"] * rng.choice([-1.0, 1.0], (nch, p))
kappa = np.full(p, (P["gamma0"] / math.sqrt(2 * math.pi)) / (1 / 0.5 - 1))
ens = pkg.Ensemble(nch, p, sampler=pkg._lib.SAMPLER_STICKY_ZIGZAG, adapt=True, factor=5.0, trace_capacity=0)
ens.set_flow(pkg.ZigZag(P["Gdrop"], P["mu"], P["sigma"]))
ens.set_target(pkg.LogisticTarget(P["A"], P["y"], P["ny"], P["mu"], P["gamma0"], 10))
ens.set_sticky(kappa)
ens.set_state(0.0, X0, TH0, P["c"], np.arange(nch, dtype=np.uint64) + 0x5EED0000)
ens.run(2.0, pkg._lib.RUN_STOP_BEFORE)
e0 = ens.totals()
ens.run(2.0 + T, pkg._lib.RUN_STOP_BEFORE, sync=False)
ms = ens.last_run_ms()
e1 = ens.totals()
cnt = ens.counters()
fs = ens.final_state(0, min(nch, 256))
print(json.dumps({"config": f"C5-logistic: sticky ZigZag, spike-and-slab logistic n=8840 p=442 (k=10), {nch} chains, dT={T}",
                  "kernel_ms": ms, "events_per_s": (e1["nevents"] - e0["nevents"]) / (ms * 1e-3),
                  "frozen_fraction_at_end": float(np.mean(fs["theta"] == 0)),
                  "unhealthy_chains": int(np.count_nonzero(cnt["status"] != 0))}))
ens.close()
