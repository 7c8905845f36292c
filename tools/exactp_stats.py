"""Iteration statistics of the one-proposal-per-lane kernel of the moving evaluation (zz_local_exactp_kernel, opt-in) on the C3
workload: iterations, candidates, events, what the zone test and the exposure prefix leave, and -- with a library built by
`python zigzagboomerang.jl_amd/build.py --variant xph -D X_PHASES=1` and selected with PDMP_MI355_LIB -- cycles per phase of
chain 0 (ring, select, loads + key lines, events, sums, rank space, groups, validate + commit, tail).  GPU box only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
G = pkg.problems.gmrf_precision(128)
d = G.shape[0]
c = pkg.problems.column_norms(G)
os.environ["PDMP_KERNEL"] = "exactp"
with pkg.Ensemble(4096, d, trace_capacity=40000) as e:
    e.set_flow(pkg.ZigZag(G, np.zeros(d)))
    e.set_target(pkg.GaussianTarget(G))
    e.set_state_synthetic(0.0, c, 0x5EED0000)
    e.run(2.0, pkg._lib.RUN_STOP_BEFORE)
    e.trace_reset()
    e.debug_phase_profile(True)
    e.run(3.0, pkg._lib.RUN_STOP_BEFORE)
    kind, ph = e.debug_phase_cycles()
    it = max(ph[10], 1)
    print("exactp ms", round(e.last_run_ms(), 2), "iters", int(ph[10]), "cand/it %.1f" % (ph[11] / it), "events/it %.1f" % (ph[14] / it),
          "zone/it %.1f" % (ph[12] / it), "commit/it %.1f" % (ph[13] / it))
    tot = sum(ph[:9])
    if tot > 0:
        print("phases cycles/iter:", [int(v / it) for v in ph[:9]], "total/iter", int(tot / it))
