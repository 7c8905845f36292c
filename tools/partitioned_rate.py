#!/usr/bin/env python
"""Rate of the partitioned mode (one chain = a workgroup of K wavefronts, pdmp_ensemble_run_partitioned) on the north-star geometry:
one chain alone (what the reference's parallel_spdmp is for) and an ensemble that fills the GPU, next to the oracle's threaded restatement
on this host's cores.   usage: tools/partitioned_rate.py [K] [T]"""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
G = pkg.problems.gmrf_precision(128)
d = G.shape[0]
k = d // K
coo = sp.coo_matrix(G)
keep = (coo.row // k) == (coo.col // k)
Gb = sp.csc_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=G.shape)
Gb.sort_indices()
c = 2.0 * pkg.problems.column_norms(G)
own = sp.csc_matrix((np.ones(Gb.nnz), Gb.indices, Gb.indptr), shape=G.shape)
cols = np.repeat(np.arange(d), np.diff(G.indptr))
mask = (np.asarray(own[G.indices, cols]).reshape(-1) != 0).astype(np.uint8)
vals = np.asarray(Gb[G.indices, cols]).reshape(-1)
Gu = sp.csc_matrix((vals, G.indices.copy(), G.indptr.copy()), shape=G.shape)
F = pkg.ZigZag(Gu, np.zeros(d))
F.Γ = Gu
for nch in (1, 256 * 16 // K):
    ens = pkg.Ensemble(nch, d, adapt=True, factor=1.8, trace_capacity=int(3.0 * d * T) + 4096)
    ens.set_flow(F)
    ens.set_target(pkg.GaussianTarget(G))
    ens.set_state_synthetic(0.0, c, 0x5EED0000)
    t0 = time.perf_counter()
    ens.run_partitioned(T, K, 0.1, mask)
    wall = time.perf_counter() - t0
    ms = ens.last_run_ms()
    cnt = ens.counters()
    assert np.all(cnt["status"] == 0), cnt["status"]
    print(json.dumps({"chains": nch, "K": K, "T": T, "kernel_ms": round(ms, 2), "events": int(cnt["nacc"].sum()),
                      "events_per_s": float(cnt["nacc"].sum() / (ms * 1e-3)), "rounds_per_chain": float(cnt["nrefresh"].mean()),
                      "host_wall_s": round(wall, 3)}), flush=True)
    ens.close()
try:
    import oracle_lib as O
    x0, th0 = O.synthetic_state(0x5EED0000, d)
    Kc = min(K, 8)
    kc = d // Kc
    keepc = (coo.row // kc) == (coo.col // kc)
    Gc = sp.csc_matrix((coo.data[keepc], (coo.row[keepc], coo.col[keepc])), shape=G.shape)
    Gc.sort_indices()
    r = O.parallel_spdmp(Gc, None, G, x0, th0, c, T, Kc, 0.1, seed=0x5EED0000, adapt=True, want_trace=False)
    print(json.dumps({"oracle_threads": Kc, "events_per_s": r["nacc"] / r["seconds"], "rounds": int(r["rounds"])}), flush=True)
except Exception as exc:  # (the oracle is test infrastructure: absent -> only the device figures)
    print(json.dumps({"oracle": str(exc)}))
