#!/usr/bin/env python
"""Proposals committed per iteration of the speculative kernel (chain 0): PDMP_PHASE=1 python tools/commit_rate.py"""
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
ens.run(0.5, pkg._lib.RUN_STOP_BEFORE)
ens.trace_reset()
n0 = ens.counters()["num"][0]
ens.run(1.5, pkg._lib.RUN_STOP_BEFORE)
n1 = ens.counters()["num"][0]
print("chain 0 proposals in the slice:", int(n1 - n0), "(divide by the `iters` printed above)")
ens.close()
