import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from __graft_entry__ import load_package
import oracle_lib as O
pkg = load_package()
n = 256
G = pkg.problems.gmrf_precision(n); d = n*n; c = pkg.problems.column_norms(G)
T = 0.5
for tracked in (True, False):
    e = pkg.Ensemble(2, d, trace_capacity=2*d)
    e.set_flow(pkg.ZigZag(G, np.zeros(d))); e.set_target(pkg.GaussianTarget(G))
    if tracked: e.set_gradient_tracking(True)
    e.set_state_synthetic(0.0, c, 777)
    e.run(T, pkg._lib.RUN_STOP_BEFORE)
    cn = e.counters(); print("tracked" if tracked else "moving", e.kernel_name(), cn["num"], cn["nacc"], cn["status"], e.last_run_ms())
    for k in range(2):
        x0, th0 = O.synthetic_state(777 + k, d)
        r = O.spdmp_zigzag(G, None, G, x0, th0, c, T, seed=777 + k, stop_before_T=True, tracked=tracked)
        ev = e.trace(k, counters=cn)
        ok = len(ev) == len(r["events"]) and all(np.array_equal(ev[f], r["events"][f]) for f in ("i","t","x","theta")) and int(cn["num"][k]) == r["num"]
        fs = e.final_state(k, 1)
        ok = ok and np.array_equal(fs["x"][0], r["x"]) and np.array_equal(fs["t"][0], r["t"]) and np.array_equal(fs["theta"][0], r["theta"])
        print("  chain", k, "events", len(ev), "bitwise", ok)
    e.close()
