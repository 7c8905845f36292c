#!/usr/bin/env python
"""Randomised bitwise soak of the tracked kernels' three forms against the oracle's tracked evaluation (tolerance 0: events, final clocks, positions,
velocities, counters): random lattice sizes (d = 2116 .. 25600: both sides of d = 16384), horizons, bound multipliers, start times and seeds, with the
one-wave and the two-wave form forced in turn (PDMP_HELPER_WAVE) and the launcher's own choice.   python tools/track_soak_bitwise.py [rounds] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402
import oracle_lib as O  # noqa: E402

pkg = load_package()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20260929)
bad = 0
for r_ in range(rounds):
    n = int(rng.integers(46, 161))
    d = n * n
    T = float(rng.uniform(0.6, 3.0) * (64.0 / n) ** 2 * 4.0)
    cmul = float(rng.choice([1.0, 1.0, 1.5, 3.0]))
    t0 = float(rng.choice([0.0, 0.0, 2.5]))
    nch = int(rng.integers(1, 4))
    seed = int(rng.integers(1, 1 << 30))
    G = pkg.problems.gmrf_precision(n)
    x0 = rng.standard_normal((nch, d))
    th0 = rng.choice([-1.0, 1.0], (nch, d))
    c = cmul * pkg.problems.column_norms(G)
    forms = ("", "0", "1") if d <= 16384 else ("",)
    ref = [O.spdmp_zigzag(G, None, G, x0[k], th0[k], c, t0 + T, t0=t0, seed=seed + k, tracked=True) for k in range(nch)]
    for form in forms:
        if form:
            os.environ["PDMP_HELPER_WAVE"] = form
        else:
            os.environ.pop("PDMP_HELPER_WAVE", None)
        tr, (t, x, th), (acc, num), _ = pkg.spdmp(pkg.GaussianTarget(G), t0, x0, th0, t0 + T, c, pkg.ZigZag(G, np.zeros(d)), seed=seed, tracked=True,
                                                  trace_capacity=int(8 * d * T * max(cmul, 1.0)) + 8192)
        for k in range(nch):
            r = ref[k]
            ev, oe = tr[k].events, r["events"]
            ok = (len(ev) == len(oe) and all(np.array_equal(ev[f], oe[f]) for f in ("i", "t", "x", "theta")) and int(num[k]) == r["num"]
                  and np.array_equal(acc[k], r["acc"]) and np.array_equal(th[k], r["theta"]) and np.array_equal(t[k], r["t"]) and np.array_equal(x[k], r["x"]))
            bad += not ok
            print("n=%d d=%d T=%.3f c x %.1f t0=%.1f form=%s chain %d: events %d proposals %d %s" % (n, d, T, cmul, t0, form or "auto", k, len(oe), r["num"],
                                                                                                  "OK" if ok else "MISMATCH"), flush=True)
os.environ.pop("PDMP_HELPER_WAVE", None)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
