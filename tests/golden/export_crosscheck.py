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
OUT = HERE  # where the fixtures are written (tests/test_oracle_crosscheck_fixtures.py points it at a scratch directory)


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
        with open(os.path.join(OUT, f"crosscheck_spdmp_{name}.txt"), "w") as f:
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
    with open(os.path.join(OUT, "crosscheck_bps_d8.txt"), "w") as f:
        f.write("sampler bps\nseed 77\nT %s\nlambda_ref %s\nrho %s\nc %s\n" % (hx([20.0]), hx([0.5]), hx([0.0]), hx([1.1])))
        write_matrix(f, "Gamma", G)
        write_matrix(f, "L", Lc)
        f.write("x0 %s\ntheta0 %s\n" % (hx(x0), hx(th0)))
        f.write("num %d\nacc %d\nevents %d\n" % (r["num"], r["nacc"], r["nevents"]))
        for k in range(r["nevents"]):
            f.write("%s %s %s\n" % (hx([r["t_ev"][k]]), hx(r["x_ev"][k]), hx(r["theta_ev"][k])))
    # sticky 1-d, test/sticky.jl:7-36 parameters (golden.npz: sticky1d_events)
    ev = g["sticky1d_events"]
    with open(os.path.join(OUT, "crosscheck_sspdmp_1d.txt"), "w") as f:
        f.write("sampler sspdmp\nseed 5\nT %s\n" % hx([200.0]))
        f.write("sigma2 %s\nmu %s\nkappa %s\nc %s\nx0 %s\ntheta0 %s\n" % (hx([0.5]), hx([0.9]), hx([1.5]), hx([20.0]), hx([1.0]), hx([0.8])))
        f.write("num %d\nacc %d\nevents %d\n" % (int(g["sticky1d_counts"][0]), int(g["sticky1d_counts"][1]), len(ev)))
        for e in ev:
            f.write("%s %d %s %s\n" % (hx([e["t"]]), int(e["i"]) + 1, hx([e["x"]]), hx([e["theta"]])))
    write_more()
    write_1d()
    print("wrote crosscheck_*.txt")


def write_1d():
    """The 1-d samplers of src/zigzagboom1d.jl:34-67 on the closures of test/test1d.jl:9-10 (tools/julia_crosscheck.jl: check_1d)."""
    mu, s2 = np.pi / 3, 1.3
    for name, kw, x0, th0, c in (("zigzag1d", dict(flow="zigzag", noise=0.1), 1.01, -1.5, 10.0),
                                 ("boomerang1d", dict(flow="boomerang", boomerang=(1.1, 1.2, 0.5), noise=0.1), 1.41, 0.5, 10.0)):
        r = O.pdmp_1d(mu, s2, x0, th0, 200.0, c, seed=3, **kw)
        assert r["status"] == 0 and len(r["events"]) > 50
        with open(os.path.join(OUT, f"crosscheck_{name}.txt"), "w") as f:
            f.write("sampler %s\nseed 3\nT %s\nmu %s\nsigma2 %s\nnoise %s\n" % (name, hx([200.0]), hx([mu]), hx([s2]), hx([kw["noise"]])))
            if name == "boomerang1d":
                b = kw["boomerang"]
                f.write("b_sigma %s\nb_mu %s\nb_lambda %s\n" % (hx([b[0]]), hx([b[1]]), hx([b[2]])))
            f.write("x0 %s\ntheta0 %s\nc %s\n" % (hx([x0]), hx([th0]), hx([c])))
            f.write("num %d\nacc %d\nndraw %d\nevents %d\n" % (r["num"], r["acc"], r["ndraw"], len(r["events"])))
            for e in r["events"]:
                f.write("%s %s %s\n" % (hx([e["t"]]), hx([e["x"]]), hx([e["theta"]])))


def write_fact(f, r, d):
    f.write("num %d\nacc %s\nnrefresh %d\nevents %d\n" % (r["num"], " ".join(str(int(a)) for a in r["acc"]), r["nrefresh"], len(r["events"])))
    for e in r["events"]:
        f.write("%s %d %s %s\n" % (hx([e["t"]]), int(e["i"]) + 1, hx([e["x"]]), hx([e["theta"]])))


def write_more():
    """Round 3: every other family the engine builds -- FactBoomerang spdmp, ZigZag with a refresh clock, the factorised LocalBound,
    Boomerang, the subsampled logistic gradient -- so that ONE Julia run pins them all (tools/julia_crosscheck.jl: check_*)."""
    rng = np.random.default_rng(11)
    G = P.maintest_precision(8)
    d = 8
    c = P.column_norms(G)
    # (a) FactBoomerang, test/maintest.jl:114-137: Z = FactBoomerang(1.2Γ, 0, 0.3), ∇ϕ(x, i, Γ) = idot(Γ, i, x)
    x0 = rng.random(d)
    Gf = sp.csc_matrix(1.2 * G)
    th0 = rng.standard_normal(d) / np.sqrt(Gf.diagonal())
    sigma = 1.0 / np.sqrt(Gf.diagonal())  # FactBoomerang(Γ, μ, λ) sets σ = (Vector(diag(Γ))).^(-0.5), src/types.jl:79
    r = O.spdmp_zigzag(Gf, np.zeros(d), G, x0, th0, c, 30.0, seed=21, lambda_ref=0.3, sigma=sigma, factboomerang=True)
    assert r["status"] == 0 and r["nrefresh"] > 3
    with open(os.path.join(OUT, "crosscheck_factboomerang_d8.txt"), "w") as f:
        f.write("sampler factboomerang\nseed 21\nT %s\nscale %s\nlambda_ref %s\n" % (hx([30.0]), hx([1.2]), hx([0.3])))
        write_matrix(f, "Gamma", G)
        f.write("x0 %s\ntheta0 %s\nc %s\n" % (hx(x0), hx(th0), hx(c)))
        write_fact(f, r, d)
    # (b) ZigZag with a refresh clock λref = 0.4 (src/sfact.jl:78-114; test/staticarrays.jl:45 uses one): σ = 1
    x0, th0 = rng.standard_normal(d), rng.choice([-1.0, 1.0], d)
    r = O.spdmp_zigzag(G, np.zeros(d), G, x0, th0, 1.5 * c, 40.0, seed=22, lambda_ref=0.4)
    assert r["status"] == 0 and r["nrefresh"] > 5
    with open(os.path.join(OUT, "crosscheck_zigzag_refresh_d8.txt"), "w") as f:
        f.write("sampler zigzag_refresh\nseed 22\nT %s\nscale %s\nlambda_ref %s\n" % (hx([40.0]), hx([1.0]), hx([0.4])))
        write_matrix(f, "Gamma", G)
        f.write("x0 %s\ntheta0 %s\nc %s\n" % (hx(x0), hx(th0), hx(1.5 * c)))
        write_fact(f, r, d)
    # (c) factorised LocalBound, src/local.jl:95-149 with the (∇ϕi, vi) callback of performance/smartbound.jl:45-59; distinct c_i/|θ_i|
    # (tied horizons are the one documented divergence: INTEGRATION.md)
    x0 = rng.standard_normal(d)
    th0 = rng.choice([-1.0, 1.0], d) * (1.0 + 0.1 * np.arange(d))
    cl = 0.6 * c
    r = O.spdmp_zigzag(G, np.zeros(d), G, x0, th0, cl, 25.0, seed=23, local_bound=True)
    assert r["status"] == 0 and len(r["events"]) > 50
    with open(os.path.join(OUT, "crosscheck_localbound_d8.txt"), "w") as f:
        f.write("sampler localbound\nseed 23\nT %s\nscale %s\n" % (hx([25.0]), hx([1.0])))
        write_matrix(f, "Gamma", G)
        f.write("x0 %s\ntheta0 %s\nc %s\n" % (hx(x0), hx(th0), hx(cl)))
        write_fact(f, r, d)
    # (d) Boomerang, test/maintest.jl:139-154: B = Boomerang(Γ, 0, 0.5), c = 16, ∇ϕ!(y, x) = Γx, L = cholesky(Γ).L
    x0, th0 = rng.standard_normal(d), rng.standard_normal(d)
    Lc = np.tril(np.linalg.cholesky(G.toarray()))
    r = O.pdmp_bps(G, None, x0, th0, 16.0, 15.0, lambda_ref=0.5, seed=24, ev_cap=5000, boomerang_mu=np.zeros(d), mass_L=sp.csc_matrix(Lc))
    assert r["status"] == 0 and r["nevents"] > 5
    with open(os.path.join(OUT, "crosscheck_boomerang_d8.txt"), "w") as f:
        f.write("sampler boomerang\nseed 24\nT %s\nlambda_ref %s\nrho %s\nc %s\n" % (hx([15.0]), hx([0.5]), hx([0.0]), hx([16.0])))
        write_matrix(f, "Gamma", G)
        write_matrix(f, "L", Lc)
        f.write("x0 %s\ntheta0 %s\n" % (hx(x0), hx(th0)))
        f.write("num %d\nacc %d\nevents %d\n" % (r["num"], r["nacc"], r["nevents"]))
        for k in range(r["nevents"]):
            f.write("%s %s %s\n" % (hx([r["t_ev"][k]]), hx(r["x_ev"][k]), hx(r["theta_ev"][k])))
    # (e) subsampled logistic gradient ∇ϕmoving (scripts/logistic.jl:78-107,167) on sparse_design([2, 2], 2, 6): n = 60, p = 10
    Pl = P.logistic_problem(levels=(2, 2), r=2, m=6, seed=5)
    p_ = Pl["p"]
    lg = dict(A=Pl["A"], At=Pl["At"], y=Pl["y"], ny=Pl["ny"], mu=Pl["mu"], gamma0=Pl["gamma0"], k=3)
    th0 = Pl["sigma"] * rng.choice([-1.0, 1.0], p_)
    r = O.spdmp_zigzag(Pl["Gdrop"], Pl["mu"], Pl["Gdrop"], Pl["x0"], th0, Pl["c"], 30.0, seed=25, adapt=True, factor=5.0, logistic=lg, sigma=Pl["sigma"])
    assert r["status"] == 0 and len(r["events"]) > 30
    with open(os.path.join(OUT, "crosscheck_logistic_p%d.txt" % p_), "w") as f:
        f.write("sampler logistic\nseed 25\nT %s\ngamma0 %s\nksub 3\nfactor %s\n" % (hx([30.0]), hx([Pl["gamma0"]]), hx([5.0])))
        A = sp.coo_matrix(Pl["A"])
        f.write("A %d %d %d\n" % (A.shape[0], A.shape[1], A.nnz))
        for i, j, v in zip(A.row, A.col, A.data):
            f.write("%d %d %s\n" % (i + 1, j + 1, hx([v])))
        write_matrix(f, "Gamma", Pl["Gdrop"])
        f.write("y %s\nny %s\nmu %s\nsigma %s\n" % (hx(Pl["y"]), hx(Pl["ny"]), hx(Pl["mu"]), hx(Pl["sigma"])))
        f.write("x0 %s\ntheta0 %s\nc %s\ncout %s\n" % (hx(Pl["x0"]), hx(th0), hx(Pl["c"]), hx(r["c"])))
        f.write("ndraw_global %d\n" % r["ndraw_global"])
        write_fact(f, r, p_)


if __name__ == "__main__":
    main()
