#!/usr/bin/env python
"""Per-phase cycle profile of the general-neighbourhood kernel (chain 0) on a secondary configuration of bench.py:
   tools/phase_general.py C4|C5      phases: select, move G1, gradient, coin + G2, re-bound, re-queue, tail"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
pkg = load_package()
ap = argparse.Namespace(config=cfg, chains=bench.CONFIG_DEFAULTS[cfg]["chains"], dt=bench.CONFIG_DEFAULTS[cfg]["dt"], grid=128, no_trace=False, exact=False, gather=False, tracked=False, graph="lattice3d")
W = bench.make_workload(pkg, ap, 0, 0)
ens = W["ens"]
L = pkg._lib
T = 0.0
for k in range(3):
    T += ap.dt
    ens.trace_reset()
    ens.run(T, L.RUN_STOP_BEFORE)
ens.debug_phase_profile(True)
ens.trace_reset()
T += ap.dt
ens.run(T, L.RUN_STOP_BEFORE)
kind, ph = ens.debug_phase_cycles()
n = max(ph[10], 1.0)
names = ["select", "move G1", "gradient", "coin + G2", "re-bound", "re-queue", "tail / (LDS kernel: own re-bound sums)"]
print(cfg, "kind", kind, "proposals of chain 0:", int(ph[10]), "kernel ms", round(ens.last_run_ms(), 2))
tot = sum(ph[:7])
for q in range(7):
    print("  %-10s %8.0f cycles/proposal  %4.1f %%" % (names[q], ph[q] / n, 100.0 * ph[q] / tot))
