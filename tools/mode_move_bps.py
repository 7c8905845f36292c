#!/usr/bin/env python
"""Config C2 (Bouncy Particle, d = 1024, 4096 chains, full PDMPTrace records) runs its steps at 4.8 or at 5.6 ms by the process: which ARRAY's placement is it?
One ensemble; each array in turn is copied into newly allocated memory several times (pdmp_debug_move_buffer), a few steps timed after every move.
    python tools/mode_move_bps.py [--moves 5] [--arrays 6,7,8,9,10] [--cap 512]"""
import argparse
import json
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--moves", type=int, default=5)
ap.add_argument("--arrays", default="6,7,8,9,10")
ap.add_argument("--cap", type=int, default=512)
ap.add_argument("--dt", type=float, default=30.0)
args = ap.parse_args()
NAMES = {6: "event x", 7: "event theta", 8: "x", 9: "theta", 10: "event t"}
pkg = load_package()
L = pkg._lib
nch, d = 4096, 1024
rng = np.random.default_rng(1000)
ens = pkg.Ensemble(nch, d, sampler=L.SAMPLER_BPS, factor=2.0, trace_capacity=args.cap)
ens.set_flow_bps(pkg.BouncyParticle(sp.identity(d, format="csc"), np.zeros(d), 1.0))
ens.set_state_bps(0.0, rng.standard_normal((nch, d)), rng.standard_normal((nch, d)), 1e-3, np.arange(nch, dtype=np.uint64) + np.uint64(0x5EED0000))
T = [0.0]


def steps(n=3):
    ms = []
    c0 = ens.counters()
    for _ in range(n):
        T[0] += args.dt
        tot = 0.0
        while True:
            ens.run(T[0], L.RUN_STOP_BEFORE)
            tot += ens.last_run_ms()
            full = L.needs_rerun(ens.counters()["status"])
            ens.trace_reset()
            if not full:
                break
        ms.append(tot)
    c1 = ens.counters()
    return round(float(np.mean(ms[1:])), 2), round((int(c1["nevents"].sum()) - int(c0["nevents"].sum())) / n / 1e6, 3)


steps(2)
print(json.dumps({"start": steps()}), flush=True)
for a in [int(v) for v in args.arrays.split(",")]:
    seq = []
    for _ in range(args.moves):
        ens.debug_move_buffer(a)
        seq.append(steps())
    print(json.dumps({"moved": NAMES[a], "ms_Mevents_after_each_move": seq}), flush=True)
ens.close()
