"""Writes the plain-text fixtures tools/julia_crosscheck.jl reads (no Julia package can read .npz without extra dependencies):
the inputs and the oracle's event sequences of the d8 / grid8 local-ZigZag chains of golden.npz, a BPS chain with a genuine
mass factor, and the 1-d sticky chain.  Every float is written as its IEEE-754 bit pattern (16 hex digits).

    python tests/golden/export_crosscheck.py
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as O  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
P = pkg.problems


def hx(a):
    return " ".join("%016x" % v for v in np.ascontiguousarray(a, dtype=np.float64).view(np.uint64).ravel())


def write_matrix(f, name, A):
    A = sp.coo_matrix(A)
    f.write("%s %d %d\n" % (name, A.shape[0], A.nnz))
    for i, j, v in zip(A.row, A.col, A.data):
        f.write("%d %d %s\n" % (i + 1, j + 1, hx([v])))  # 1-based for Julia


def main():
    g = np.load(os.path.join(HERE, "golden.npz"))
    for name, G, scale in (("d8", P.maintest_precision(8), 0.9), ("grid8", P.gmrf_precision(8), 1.0)):
        ev = g[f"{name}_events"]
        with open(os.path.join(HERE, f"crosscheck_spdmp_{name}.txt"), "w") as f:
            f.write("sampler spdmp\nseed 1234\nT %s\nscale %s\n" % (hx([50.0]), hx([scale])))
            write_matrix(f, "Gamma", G)
            f.write("x0 %s\ntheta0 %s\nc %s\n" % (hx(g[f"{name}_x0"]), hx(g[f"{name}_th0"]), hx(g[f"{name}_c"])))
            f.write("num %d\nacc %s\nevents %d\n" % (int(g[f"{name}_num"][0]), " ".join(str(int(a)) for a in g[f"{name}_acc"]), len(ev)))
            for e in ev:
                f.write("%s %d %s %s\n" % (hx([e["t"]]), int(e["i"]) + 1, hx([e["x"]]), hx([e["theta"]])))
    # BPS with the mass factor L = cholesky(Γ).L, test/maintest.jl:156-172 shape (λref = 0.5, c = 1.1), T = 20
    G = P.maintest_precision(8)
    Lc = np.tril(np.linalg.cholesky(G.toarray()))
    rng = np.random.default_rng(3)
    x0, th0 = rng.standard_normal(8), rng.standard_normal(8)
    r = O.pdmp_bps(G, None, x0, th0, 1.1, 20.0, lambda_ref=0.5, seed=77, ev_cap=5000, mass_L=sp.csc_matrix(Lc))
    assert r["status"] == 0
    with open(os.path.join(HERE, "crosscheck_bps_d8.txt"), "w") as f:
        f.write("sampler bps\nseed 77\nT %s\nlambda_ref %s\nrho %s\nc %s\n" % (hx([20.0]), hx([0.5]), hx([0.0]), hx([1.1])))
        write_matrix(f, "Gamma", G)
        write_matrix(f, "L", Lc)
        f.write("x0 %s\ntheta0 %s\n" % (hx(x0), hx(th0)))
        f.write("num %d\nacc %d\nevents %d\n" % (r["num"], r["nacc"], r["nevents"]))
        for k in range(r["nevents"]):
            f.write("%s %s %s\n" % (hx([r["t_ev"][k]]), hx(r["x_ev"][k]), hx(r["theta_ev"][k])))
    # sticky 1-d, test/sticky.jl:7-36 parameters (golden.npz: sticky1d_events)
    ev = g["sticky1d_events"]
    with open(os.path.join(HERE, "crosscheck_sspdmp_1d.txt"), "w") as f:
        f.write("sampler sspdmp\nseed 5\nT %s\n" % hx([200.0]))
        f.write("sigma2 %s\nmu %s\nkappa %s\nc %s\nx0 %s\ntheta0 %s\n" % (hx([0.5]), hx([0.9]), hx([1.5]), hx([20.0]), hx([1.0]), hx([0.8])))
        f.write("num %d\nacc %d\nevents %d\n" % (int(g["sticky1d_counts"][0]), int(g["sticky1d_counts"][1]), len(ev)))
        for e in ev:
            f.write("%s %d %s %s\n" % (hx([e["t"]]), int(e["i"]) + 1, hx([e["x"]]), hx([e["theta"]])))
    print("wrote crosscheck_*.txt")


if __name__ == "__main__":
    main()
