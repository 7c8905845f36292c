#!/bin/bash
# A/B of the headline kernel against the round-4 build inside one GPU session (interleaved).  tools/_r04ref/ (git-ignored) holds the round-4
# package and its library: `git worktree add /tmp/r04tree <round-4 commit>`, build there, copy __graft_entry__.py + zigzagboomerang.jl_amd/.
cat > /tmp/ab_one.py <<'PY'
import os, sys, numpy as np
root = sys.argv[1]
sys.path.insert(0, root)
from __graft_entry__ import load_package
pkg = load_package()
G = pkg.problems.gmrf_precision(128); d = G.shape[0]; c = pkg.problems.column_norms(G)
for nch in [int(w) for w in sys.argv[2].split(",")]:
    ens = pkg.Ensemble(nch, d, trace_capacity=2 * d + 1024)
    ens.set_flow(pkg.ZigZag(G, np.zeros(d))); ens.set_target(pkg.GaussianTarget(G))
    if len(sys.argv) < 4: ens.set_gradient_tracking(True)
    ens.set_state_synthetic(0.0, c, 0x5EED0000)
    ms = []
    for k in range(10):
        ens.run(float(k + 1), pkg._lib.RUN_STOP_BEFORE, sync=False); ms.append(ens.last_run_ms()); ens.trace_reset()
    cn = ens.counters()
    print(os.path.basename(root) or "new", nch, ens.kernel_name(), "ms/step %.2f" % np.mean(ms[2:]), "num", int(cn["num"].sum()), "nacc", int(cn["nacc"].sum()),
          "ndraw", int(cn["ndraw_main"].sum()), flush=True)
    ens.close()
PY
W=${1:-4096,2048}
for rep in 1 2; do
  python /tmp/ab_one.py $PWD/tools/_r04ref $W $2
  python /tmp/ab_one.py $PWD $W $2
done
