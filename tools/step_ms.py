#!/usr/bin/env python
"""Per-launch kernel time of the headline workload over a long run (is the 116 / 130 ms split per process or in time?)."""
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
ens = pkg.Ensemble(4096, d, trace_capacity=40000)
ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
ens.set_target(pkg.GaussianTarget(G))
ens.set_state_synthetic(0.0, c, 0x5EED0000)
ms = []
for s in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    ens.run(float(s + 1), pkg._lib.RUN_STOP_BEFORE)
    ens.trace_reset()
    ms.append(ens.last_run_ms())
print(" ".join(f"{m:.0f}" for m in ms))
ens.close()
