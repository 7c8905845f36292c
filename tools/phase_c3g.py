#!/usr/bin/env python
"""Per-phase cycle profile (chain 0) of the moving evaluation's kernel on config C3G's graphs: python tools/phase_c3g.py [lattice3d|random6] [kernel]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
graph = sys.argv[1] if len(sys.argv) > 1 else "lattice3d"
if len(sys.argv) > 2:
    os.environ["PDMP_KERNEL"] = sys.argv[2]
G = pkg.problems.lattice3d_precision(25) if graph == "lattice3d" else pkg.problems.random_sparse_precision(16384, int(graph[6:]))
d = G.shape[0]
c = pkg.problems.column_norms(G)
for nch in (256, 4096):
    ens = pkg.Ensemble(nch, d, trace_capacity=40000)
    ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
    ens.set_target(pkg.GaussianTarget(G))
    ens.set_state_synthetic(0.0, c, 0x5EED0000)
    ens.run(0.5, pkg._lib.RUN_STOP_BEFORE)
    ens.trace_reset()
    ens.debug_phase_profile(True)
    n0 = ens.counters()["num"][0]
    ens.run(1.0, pkg._lib.RUN_STOP_BEFORE)
    kind, ph = ens.debug_phase_cycles()
    n1 = ens.counters()["num"][0]
    it = max(ph[10], 1.0)
    print(graph, ens.kernel_name(), "chains", nch, "iters=%.0f cycles/iter:" % ph[10], " ".join("p%d=%.0f" % (q, ph[q] / it) for q in range(10)),
          "| committed per iteration: %.2f" % ((n1 - n0) / it), "| ms", round(ens.last_run_ms(), 2), flush=True)
    ens.close()
