#!/usr/bin/env python
"""Does the ms per step of the headline workload depend on where the state landed in memory?  One process, the ensemble created and
destroyed several times (optionally with a dummy allocation of a different size in front each time):   python tools/placement_probe.py [rounds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
import torch  # noqa: E402

G = pkg.problems.gmrf_precision(128)
d = G.shape[0]
c = pkg.problems.column_norms(G)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for r in range(rounds):
    pad = torch.empty((r * 37 + 1) * (1 << 20), dtype=torch.uint8, device="cuda") if os.environ.get("PAD") else None
    ens = pkg.Ensemble(4096, d, trace_capacity=int(2.0 * d) + 1024)
    ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
    ens.set_target(pkg.GaussianTarget(G))
    ens.set_gradient_tracking(True)
    ens.set_state_synthetic(0.0, c, 0x5EED0000)
    ms = []
    for k in range(5):
        ens.trace_reset()
        ens.run(float(k + 1), pkg._lib.RUN_STOP_BEFORE)
        ms.append(ens.last_run_ms())
    print("round", r, " ".join("%.2f" % m for m in ms), flush=True)
    ens.close()
    del pad
