"""4096 chains on the line-layout kernel with the invariant-checking build (PDMP_MI355_LIB=...tlcheck.so): prints TLCHECK lines"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from __graft_entry__ import load_package
pkg = load_package()
n = 128
T = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
G = pkg.problems.gmrf_precision(n)
d = n * n
c = pkg.problems.column_norms(G)
with pkg.Ensemble(nch, d, trace_capacity=0) as e:
    e.set_flow(pkg.ZigZag(G, np.zeros(d)))
    e.set_target(pkg.GaussianTarget(G))
    e.set_gradient_tracking(True)
    e.set_state_synthetic(0.0, c, 0x5EED0000)
    t = 0.0
    while t < T:
        t = min(T, t + 1.0)
        e.run(t, pkg._lib.RUN_STOP_BEFORE)
        cnt = e.counters()
        print("t", t, "kernel", e.kernel_name(), "num", int(cnt["num"].sum()), "status", np.unique(cnt["status"]), flush=True)
