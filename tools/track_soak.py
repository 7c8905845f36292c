#!/usr/bin/env python
"""Long-horizon comparison on the headline workload: 4096 chains of config C3 advanced to T with the default tracked kernel (pairs), the
tracked kernel over key blocks of 16 (the same floats by construction) and the bit-identical moving evaluation; per-chain counters
(proposals, accepted reflections) compared after every unit of time.  The two tracked kernels must never differ; a tracked chain may
leave the exact one when a float difference of ~1e-13 flips an accept test or the order of two nearly simultaneous events.
   tools/track_soak.py [T=10]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
T = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
G = pkg.problems.gmrf_precision(128)
d = G.shape[0]
c = pkg.problems.column_norms(G)
ens = {}
for name, tracked, which in (("pairs", True, 0), ("blocks16", True, 1), ("exact", False, 0)):  # ("blocks16": now the 8-lane-group kernel)
    e = pkg.Ensemble(4096, d, trace_capacity=0)
    pkg._lib.check(e._L.pdmp_debug_set_track_groups(e._h, which))
    e.set_flow(pkg.ZigZag(G, np.zeros(d)))
    e.set_target(pkg.GaussianTarget(G))
    e.set_gradient_tracking(tracked)
    e.set_state_synthetic(0.0, c, 0x5EED0000)
    ens[name] = e
t = 0.0
bad = 0


def ndiff(a, b):
    return int(np.count_nonzero((a["num"] != b["num"]) | (a["nacc"] != b["nacc"])))


while t < T:
    t += 1.0
    cnt = {}
    for name, e in ens.items():
        e.run(t, pkg._lib.RUN_STOP_BEFORE)
        cnt[name] = e.counters()
    healthy = all(np.all(v["status"] == 0) for v in cnt.values())
    d_pb, d_pe, d_be = ndiff(cnt["pairs"], cnt["blocks16"]), ndiff(cnt["pairs"], cnt["exact"]), ndiff(cnt["blocks16"], cnt["exact"])
    bad += (d_pb != 0) or (not healthy)
    print("t = %4.0f  proposals %.4g  chains whose counters differ: pairs vs blocks16 %d, pairs vs exact %d, blocks16 vs exact %d   (%.1f / %.1f / %.1f ms)" %
          (t, cnt["pairs"]["num"].sum(), d_pb, d_pe, d_be, ens["pairs"].last_run_ms(), ens["blocks16"].last_run_ms(), ens["exact"].last_run_ms()), flush=True)
sys.exit(1 if bad else 0)
