#!/usr/bin/env python
"""Stress of the default tracked kernel against the oracle: lattices of several sizes, several chains, longer runs, a looser bound
(more proposals per reflection) -- event indices, counters and final velocities exact, times and positions to 1e-9.   tools/track_stress.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402
import oracle_lib as O  # noqa: E402

pkg = load_package()
bad = 0
for n, T, cmul, nch in [(46, 20.0, 1.0, 3), (61, 10.0, 1.0, 3), (90, 6.0, 1.0, 2), (128, 3.0, 1.0, 2), (64, 8.0, 3.0, 2), (100, 4.0, 0.9, 2)]:
    G = pkg.problems.gmrf_precision(n)
    d = n * n
    rng = np.random.default_rng(n)
    x0 = rng.standard_normal((nch, d))
    th0 = rng.choice([-1.0, 1.0], (nch, d))
    c = cmul * pkg.problems.column_norms(G)
    try:
        tr, (t, x, th), (acc, num), _ = pkg.spdmp(pkg.GaussianTarget(G), 0.0, x0, th0, T, c, pkg.ZigZag(G, np.zeros(d)), seed=9000 + n, tracked=True,
                                                  trace_capacity=int(3 * d * T) + 4096)
    except RuntimeError as exc:
        print(n, "device:", exc)
        r = O.spdmp_zigzag(G, None, G, x0[0], th0[0], c, T, seed=9000 + n)
        print("   oracle status", r["status"])
        continue
    for k in range(nch):
        r = O.spdmp_zigzag(G, None, G, x0[k], th0[k], c, T, seed=9000 + n + k)
        ev, oe = tr[k].events, r["events"]
        ok = (len(ev) == len(oe) and np.array_equal(ev["i"], oe["i"]) and int(num[k]) == r["num"] and np.array_equal(acc[k], r["acc"])
              and np.array_equal(th[k], r["theta"]) and np.allclose(ev["t"], oe["t"], rtol=1e-9, atol=0) and np.allclose(x[k], r["x"], rtol=1e-9, atol=1e-9)
              and np.allclose(t[k], r["t"], rtol=1e-9, atol=0))
        bad += not ok
        print(n, T, cmul, k, "events", len(oe), "proposals", r["num"], "OK" if ok else "MISMATCH", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
