"""First-light script for a GPU box (not a pytest file): runs the parity ladder with verbose diagnostics.

    python tests/gpu_first_light.py [--big]

Writes nothing; prints what diverges first so that a failing `pytest -m gpu` can be understood quickly.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import oracle_lib as O  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()


def first_diff(a, b):
    n = min(len(a), len(b))
    for k in range(n):
        if a[k] != b[k]:
            return k
    return n if len(a) != len(b) else -1


def probe():
    n = 1 << 20
    dev = pkg._lib.math_probe(12345, n)
    host = O.math_probe(12345, n)
    names = ["u01", "log", "div", "sqrt", "poisson_time", "randn", "exp", "sincos"]
    ok = True
    for r, nm in enumerate(names):
        same = (dev[r] == host[r]) | (np.isnan(dev[r]) & np.isnan(host[r]))
        bad = np.flatnonzero(~same)
        print(f"probe {nm:13s}: {len(bad)} / {n} differ")
        if len(bad):
            ok = False
            k = bad[0]
            print("   first:", k, dev[r, k].hex(), host[r, k].hex())
    return ok


def parity(G, nch, T, seed, label, x0=None, th0=None, c=None, bound_scale=1.0, verbose=True):
    d = G.shape[0]
    rng = np.random.default_rng(seed)
    if x0 is None:
        x0 = rng.standard_normal((nch, d))
        th0 = rng.choice([-1.0, 1.0], (nch, d))
    if c is None:
        c = pkg.problems.column_norms(G)
    Z = pkg.ZigZag(bound_scale * G, np.zeros(d))
    t0 = time.time()
    tr, (t, x, th), (acc, num), _ = pkg.spdmp(pkg.GaussianTarget(G), 0.0, x0, th0, T, c, Z, seed=seed)
    tg = time.time() - t0
    allok = True
    for k in range(nch):
        r = O.spdmp_zigzag(bound_scale * G, None, G, x0[k], th0[k], c, T, seed=seed + k)
        ev, oe = tr[k].events, r["events"]
        fd_i = first_diff(ev["i"], oe["i"])
        ok = (len(ev) == len(oe) and fd_i == -1 and np.array_equal(ev["t"], oe["t"]) and
              np.array_equal(ev["x"], oe["x"]) and np.array_equal(ev["theta"], oe["theta"]) and
              int(num[k]) == r["num"] and np.array_equal(acc[k], r["acc"]) and np.array_equal(x[k], r["x"]) and
              np.array_equal(th[k], r["theta"]) and np.array_equal(t[k], r["t"]))
        allok &= ok
        if verbose and (not ok or k == 0):
            print(f"  [{label}] chain {k}: gpu events {len(ev)} oracle {len(oe)} num {int(num[k])}/{r['num']} "
                  f"first idx diff {fd_i} ok={ok}")
            if not ok:
                j = fd_i if fd_i >= 0 else first_diff(ev["t"], oe["t"])
                lo = max(0, j - 2)
                print("   gpu   :", ev[lo:j + 3])
                print("   oracle:", oe[lo:j + 3])
    print(f"[{label}] d={d} chains={nch} T={T}: parity {'OK' if allok else 'FAILED'}  ({tg:.2f}s incl. setup)")
    return allok


def main():
    print("devices:", pkg._lib.device_count())
    ok = probe()
    ok &= parity(pkg.problems.gmrf_precision(4), 2, 5.0, 3, "grid4")
    ok &= parity(pkg.problems.gmrf_precision(8), 4, 20.0, 5, "grid8")
    ok &= parity(pkg.problems.maintest_precision(8), 3, 50.0, 7, "maintest-d8", bound_scale=1.0)
    ok &= parity(pkg.problems.gmrf_precision(16), 8, 10.0, 9, "grid16")
    ok &= parity(pkg.problems.gmrf_precision(32), 4, 4.0, 11, "grid32")
    if "--big" in sys.argv:
        ok &= parity(pkg.problems.gmrf_precision(128), 2, 0.5, 13, "grid128")
    print("ALL OK" if ok else "SOME FAILED")
    # quick throughput look
    if "--bench" in sys.argv:
        G = pkg.problems.gmrf_precision(128)
        d = G.shape[0]
        c = pkg.problems.column_norms(G)
        for nch in (1024, 4096):
            ens = pkg.Ensemble(nch, d, trace_capacity=0)
            ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
            ens.set_target(pkg.GaussianTarget(G))
            ens.set_state_synthetic(0.0, c, 0x5EED0000)
            for T in (0.25, 0.5, 1.0):
                ens.run(T, pkg._lib.RUN_STOP_BEFORE)
                ms = ens.last_run_ms()
                tot = ens.totals()
                print(f"bench nch={nch} T={T}: kernel {ms:.1f} ms, totals {tot}")
            ens.close()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
