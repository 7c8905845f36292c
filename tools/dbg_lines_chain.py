"""one chain of the 4096-chain C3 ensemble (seed SEED0 + k) on the line-layout kernel against the tracked oracle: first divergence"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["PDMP_TRACK_LINES"] = os.environ.get("PDMP_TRACK_LINES", "1")
os.environ["PDMP_HELPER_WAVE"] = "0"
from __graft_entry__ import load_package
import oracle_lib as O
pkg = load_package()
SEED0 = 0x5EED0000
T = float(sys.argv[1])
chains = [int(a) for a in sys.argv[2:]]
n = 128
G = pkg.problems.gmrf_precision(n)
d = n * n
c = pkg.problems.column_norms(G)
for k in chains:
    with pkg.Ensemble(1, d, trace_capacity=400000) as e:
        e.set_flow(pkg.ZigZag(G, np.zeros(d)))
        e.set_target(pkg.GaussianTarget(G))
        e.set_gradient_tracking(True)
        e.set_state_synthetic(0.0, c, SEED0 + k)
        evs = []
        t = 0.0
        while t < T:
            t = min(T, t + 1.0)
            while True:
                e.run(t, pkg._lib.RUN_STOP_BEFORE)
                cnt = e.counters()
                evs.append(e.trace(0, counters=cnt))
                e.trace_reset()
                if cnt["status"][0] != pkg._lib.CHAIN_TRACE_FULL:
                    break
        ev = np.concatenate(evs)
        x0, th0 = O.synthetic_state(SEED0 + k, d)
        r = O.spdmp_zigzag(G, None, G, x0, th0, c, T, seed=SEED0 + k, stop_before_T=True, tracked=True)
        oe = r["events"]
        m = min(len(ev), len(oe))
        bad = np.nonzero((ev["i"][:m] != oe["i"][:m]) | (ev["t"][:m] != oe["t"][:m]) | (ev["x"][:m] != oe["x"][:m]))[0]
        print("chain", k, e.kernel_name(), "events", len(ev), len(oe), "num", cnt["num"][0], r["num"], "first bad", bad[:3], flush=True)
        if len(bad):
            b = bad[0]
            for q in range(max(0, b - 2), min(m, b + 3)):
                print("  ", q, tuple(ev[q]), tuple(oe[q]))
