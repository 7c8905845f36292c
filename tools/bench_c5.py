#!/usr/bin/env python
"""Secondary measurement: config C5 of BASELINE.json -- sticky ZigZag (sspdmp, src/ss_fact.jl:78-215) for spike-and-slab
variable selection at p = 10 000: sparse Gaussian slab (100 x 100 grid-Laplace precision), thaw rates
κ = (γ0/√2π)/(1/w − 1), w = 1/2 (scripts/sticky/sticky_logistic_sparse.jl:194-197); one GPU's share of the ensemble.
Prints one JSON line.  Events = reflections + freezes + thaws (everything sspdmp pushes to the trace)."""
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
T = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
G = pkg.problems.gmrf_precision(100, eps=0.5)
d = G.shape[0]
gamma0, w = 0.5, 0.5
kappa = np.full(d, (gamma0 / math.sqrt(2 * math.pi)) / (1 / w - 1))
ens = pkg.Ensemble(nch, d, sampler=pkg._lib.SAMPLER_STICKY_ZIGZAG, trace_capacity=0)
ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
ens.set_target(pkg.GaussianTarget(G))
ens.set_sticky(kappa)
ens.set_state_synthetic(0.0, pkg.problems.column_norms(G), 0x5EED0000)
ens.run(1.0, pkg._lib.RUN_STOP_BEFORE)  # warm-up slice: the frozen fraction settles
t0 = ens.totals()
ens.run(1.0 + T, pkg._lib.RUN_STOP_BEFORE, sync=False)
ms = ens.last_run_ms()
t1 = ens.totals()
cnt = ens.counters()
print(json.dumps({"config": f"C5: sticky ZigZag, p={d} Gaussian spike-and-slab, {nch} chains, dT={T}",
                  "kernel_ms": ms, "proposals_per_s": (t1["num"] - t0["num"]) / (ms * 1e-3),
                  "events_per_s": (t1["nevents"] - t0["nevents"]) / (ms * 1e-3),
                  "unhealthy_chains": int(np.count_nonzero(cnt["status"] != 0))}))
ens.close()
