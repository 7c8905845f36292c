#!/usr/bin/env python
"""Does the per-process speed mode of the headline workload follow the device allocations?  Re-create the ensemble several
times inside ONE process and print the steady kernel time of each incarnation."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
G = pkg.problems.gmrf_precision(128)
d = G.shape[0]
c = pkg.problems.column_norms(G)
keep = []
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    ens = pkg.Ensemble(4096, d, trace_capacity=40000)
    ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
    ens.set_target(pkg.GaussianTarget(G))
    ens.set_state_synthetic(0.0, c, 0x5EED0000)
    ms = []
    for s in range(6):
        ens.run(float(s + 1), pkg._lib.RUN_STOP_BEFORE)
        ens.trace_reset()
        ms.append(ens.last_run_ms())
    print(rep, " ".join(f"{m:.0f}" for m in ms), flush=True)
    if len(sys.argv) > 2:
        keep.append(ens)  # keep the allocation alive: the next incarnation lands elsewhere
    else:
        ens.close()
