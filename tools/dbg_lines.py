"""first divergence of the line-layout tracked kernel from the tracked oracle (debug aid)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["PDMP_TRACK_LINES"] = os.environ.get("PDMP_TRACK_LINES", "1")
os.environ["PDMP_HELPER_WAVE"] = "0"
from __graft_entry__ import load_package
import oracle_lib as O
pkg = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
nch = 3
G = pkg.problems.gmrf_precision(n)
d = n * n
rng = np.random.default_rng(n)
x0 = rng.standard_normal((nch, d))
th0 = rng.choice([-1.0, 1.0], (nch, d))
c = pkg.problems.column_norms(G)
with pkg.Ensemble(nch, d, trace_capacity=200000) as e:
    e.set_flow(pkg.ZigZag(G, np.zeros(d)))
    e.set_target(pkg.GaussianTarget(G))
    e.set_gradient_tracking(True)
    e.set_state(0.0, x0, th0, c, [700 + n + k for k in range(nch)])
    e.run(T, pkg._lib.RUN_REFERENCE_TAIL)
    print("kernel", e.kernel_name())
    cnt = e.counters()
    print({k: cnt[k] for k in ("num", "nacc", "status", "ndraw_main")})
    for k in range(nch):
        ev = e.trace(k, counters=cnt)
        r = O.spdmp_zigzag(G, None, G, x0[k], th0[k], c, T, seed=700 + n + k, tracked=True)
        oe = r["events"]
        m = min(len(ev), len(oe))
        bad = np.nonzero((ev["i"][:m] != oe["i"][:m]) | (ev["t"][:m] != oe["t"][:m]) | (ev["x"][:m] != oe["x"][:m]))[0]
        print("chain", k, "events", len(ev), len(oe), "num", cnt["num"][k], r["num"], "first bad", bad[:3])
        if len(bad):
            b = bad[0]
            for q in range(max(0, b - 2), min(m, b + 3)):
                print("  ", q, tuple(ev[q]), tuple(oe[q]))
