#!/usr/bin/env python
"""Per-phase cycle profile of the speculative event-loop kernel (chain 0), recorded through include/pdmp_debug.h
(pdmp_debug_set_phase_profile / pdmp_debug_phase_profile):   python tools/phase_profile.py
Phases: p0 candidate selection · p1 level-1 loads issued + RNG window + blob landed in LDS · p2 header/S read ·
p3 neighbour records requested, zone-conflict check (ends when the records are needed) · p4 move, gradient, accept chain ·
p5 G2 move + re-bound · p6 patched block minimum + validation · p7 commit stores · p8 level-1 updates of other blocks."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
G = pkg.problems.gmrf_precision(128)
d = G.shape[0]
c = pkg.problems.column_norms(G)
for nch in (256, 1024, 4096):
    ens = pkg.Ensemble(nch, d, trace_capacity=40000)
    ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
    ens.set_target(pkg.GaussianTarget(G))
    if os.environ.get("TRACK"):
        ens.set_gradient_tracking(True)  # TRACK=1: the tracked-gradient kernel (same phases)
    ens.set_state_synthetic(0.0, c, 0x5EED0000)
    ens.run(0.5, pkg._lib.RUN_STOP_BEFORE)
    ens.trace_reset()
    print("chains", nch, flush=True)
    ens.debug_phase_profile(True)
    n0 = ens.counters()["num"][0]
    ens.run(1.5, pkg._lib.RUN_STOP_BEFORE)
    kind, ph = ens.debug_phase_cycles()
    n1 = ens.counters()["num"][0]
    it = max(ph[10], 1.0)
    print("iters=%.0f cycles/iter:" % ph[10], " ".join("p%d=%.0f" % (q, ph[q] / it) for q in range(10)),
          "| proposals committed per iteration: %.2f" % ((n1 - n0) / it),
          ("| candidates per iteration: selected %.2f, after the zone cut %.2f, after window / accept limits %.2f" % (ph[11] / it, ph[12] / it, ph[13] / it))
          if ph[11] > 0 else "", flush=True)
    print("kernel ms (profiling instantiation)", ens.last_run_ms(), flush=True)
    ens.close()
