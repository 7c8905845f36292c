#!/usr/bin/env python
"""set_state's placement probes allocate and free tens of GB: (1) free device memory before / inside / after sixteen full-width ensembles (both evaluations) --
nothing may be left behind; (2) the same with most of the device taken by somebody else: the probes must stop early or be skipped, never fail the call.
    python tools/probes/leak_check.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
G = pkg.problems.gmrf_precision(128)
d = G.shape[0]
c = pkg.problems.column_norms(G)


def free():
    f, _ = torch.cuda.mem_get_info(0)
    return round(f / 2**30, 2)


def one(tracked):
    ens = pkg.Ensemble(4096, d, trace_capacity=2 * d + 1024)
    ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
    ens.set_target(pkg.GaussianTarget(G))
    if tracked:
        ens.set_gradient_tracking(True)
    ens.set_state_synthetic(0.0, c, 1234)
    ens.run(1.0, pkg._lib.RUN_STOP_BEFORE)
    out = (free(), round(ens.last_run_ms(), 2), ens.debug_placement()[-90:])
    ens.close()
    return out


print("free GB at start", free())
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    a, b = one(True), one(False)
    print(r, "inside", a[0], b[0], "after", free(), "ms", a[1], b[1])
for hog_gb in (200, 240, 262):
    hog = torch.empty(hog_gb << 30, dtype=torch.uint8, device="cuda:0")
    a = one(True)
    print("with", hog_gb, "GB taken: free inside", a[0], "ms", a[1], "|", a[2])
    del hog
    torch.cuda.empty_cache()
print("free GB at the end", free())
