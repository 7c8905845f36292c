#!/usr/bin/env python
"""Experiment: does the memory order of the coordinates matter for the C3 kernel?  Runs the north-star workload with the grid
relabelled in th x tw tiles (records of a tile contiguous in HBM) and prints events/s per labelling."""
import json
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
n = 128
G0 = sp.csc_matrix(pkg.problems.gmrf_precision(n))
d = n * n
nch = 4096


def tiled_perm(th, tw):
    r, c = np.divmod(np.arange(d), n)
    tile = (r // th) * (n // tw) + (c // tw)
    within = (r % th) * tw + (c % tw)
    new = tile * (th * tw) + within  # new label of old coordinate
    return new


for name, (th, tw) in [("rowmajor", (1, n)), ("tile8x8", (8, 8)), ("tile4x16", (4, 16)), ("tile4x4", (4, 4)), ("tile2x32", (2, 32)),
                       ("tile16x16", (16, 16))]:
    new = tiled_perm(th, tw)
    old_of_new = np.argsort(new)
    G = sp.csc_matrix(G0[old_of_new][:, old_of_new])
    G.sort_indices()
    c = pkg.problems.column_norms(G)
    ens = pkg.Ensemble(nch, d, trace_capacity=int(1.2 * d))
    ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
    ens.set_target(pkg.GaussianTarget(G))
    ens.set_state_synthetic(0.0, c, 0x5EED0000)
    ens.run(1.0, pkg._lib.RUN_STOP_BEFORE)
    ens.trace_reset()
    t0 = ens.totals()
    ms = 0.0
    for k in range(3):
        ens.run(2.0 + k, pkg._lib.RUN_STOP_BEFORE, sync=False)
        ms += ens.last_run_ms()
        ens.trace_reset()
    t1 = ens.totals()
    print(json.dumps({"labelling": name, "events_per_s": (t1["nevents"] - t0["nevents"]) / (ms * 1e-3),
                      "proposals_per_s": (t1["num"] - t0["num"]) / (ms * 1e-3), "ms": ms}), flush=True)
    ens.close()
