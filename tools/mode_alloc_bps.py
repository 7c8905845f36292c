#!/usr/bin/env python
"""Config C2's ensemble created several times in one process: step time and what set_state_bps's placement probes saw (tools/mode_alloc.py for the
Bouncy Particle).    python tools/mode_alloc_bps.py [--rounds 8]   (PDMP_PLACE_TUNE=0: without the probes)"""
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
ap.add_argument("--rounds", type=int, default=8)
args = ap.parse_args()
pkg = load_package()
L = pkg._lib
nch, d, cap, dt = 4096, 1024, 512, 30.0
rng = np.random.default_rng(1000)
x0, th0 = rng.standard_normal((nch, d)), rng.standard_normal((nch, d))
for r in range(args.rounds):
    ens = pkg.Ensemble(nch, d, sampler=L.SAMPLER_BPS, factor=2.0, trace_capacity=cap)
    ens.set_flow_bps(pkg.BouncyParticle(sp.identity(d, format="csc"), np.zeros(d), 1.0))
    ens.set_state_bps(0.0, x0, th0, 1e-3, np.arange(nch, dtype=np.uint64) + np.uint64(0x5EED0000))
    ms = []
    for k in range(5):
        tot = 0.0
        while True:
            ens.run((k + 1) * dt, L.RUN_STOP_BEFORE)
            tot += ens.last_run_ms()
            full = L.needs_rerun(ens.counters()["status"])
            ens.trace_reset()
            if not full:
                break
        ms.append(round(tot, 2))
    print(json.dumps({"round": r, "ms": ms[2:], "placement": ens.debug_placement()}), flush=True)
    ens.close()
