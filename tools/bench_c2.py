#!/usr/bin/env python
"""Secondary measurement (not the headline bench): config C2 of BASELINE.json -- Bouncy Particle Sampler on the isotropic
d = 1024 Gaussian (Γ = I, μ = 0, λref = 1, ρ = 0, c = 1e-3: the values of scripts/not_fact.jl:23-28), 4096 chains on one
MI355X, run to T = 100 in slices with full PDMPTrace records (t, copy(x), copy(θ)) = 16 392 B per event written to HBM.
Roofline of this kernel = HBM WRITE bandwidth; prints one JSON line."""
import json
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
d, nch, T, dT = 1024, 4096, 100.0, 10.0
rng = np.random.default_rng(0)
x0 = rng.standard_normal((nch, d))
th0 = rng.standard_normal((nch, d))
cap = int(sys.argv[1]) if len(sys.argv) > 1 else 512  # events per chain per launch: cap x 16 392 B x 4096 chains of HBM (512 -> 34 GB)
dT = float(sys.argv[2]) if len(sys.argv) > 2 else dT
ens = pkg.Ensemble(nch, d, sampler=pkg._lib.SAMPLER_BPS, trace_capacity=cap)
ens.set_flow_bps(pkg.BouncyParticle(sp.identity(d, format="csc"), np.zeros(d), 1.0))
ens.set_state_bps(0.0, x0, th0, 1e-3, np.arange(nch, dtype=np.uint64) + 0x5EED0000)
ms = 0.0
launches = 0
Tk = dT
while True:
    ens.run(Tk, pkg._lib.RUN_STOP_BEFORE, sync=False)
    ms += ens.last_run_ms()
    launches += 1
    cnt = ens.counters()
    ens.trace_reset()
    if np.any(cnt["status"] == pkg._lib.CHAIN_TRACE_FULL):
        continue
    if Tk >= T:
        break
    Tk += dT
cnt = ens.counters()
ev = int(cnt["nevents"].sum())
num = int(cnt["num"].sum())
byts = ev * 8.0 * (2 * d + 1)
print(json.dumps({"config": "C2: BPS d=1024 isotropic Gaussian, 4096 chains, T=100, lambda_ref=1, c=1e-3, full traces",
                  "events": ev, "proposals": num, "refresh": int(cnt["nrefresh"].sum()), "kernel_ms_total": ms,
                  "launches": launches, "events_per_s": ev / (ms * 1e-3), "proposals_per_s": num / (ms * 1e-3),
                  "roofline": {"bound": "hbm", "achieved": byts / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                               "frac": byts / (ms * 1e-3) / 1e9 / 8000.0,
                               "model": "8(2d+1) bytes written per event, 0 read (state in registers)"},
                  "unhealthy_chains": int(np.count_nonzero(cnt["status"] != 0))}))
ens.close()
