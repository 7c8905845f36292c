"""Generates the committed golden vectors from the CPU oracle (Julia is absent in the build image: the
reference itself cannot be run).  Re-run after any deliberate change of the
numerical contract:   python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as O  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
P = pkg.problems


def payload_hash(ev):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(ev["t"]).tobytes())
    h.update(np.ascontiguousarray(ev["x"]).tobytes())
    h.update(np.ascontiguousarray(ev["theta"]).tobytes())
    return h.hexdigest()


def main():
    out = {}
    # (1) poisson_time table over every branch incl. the b<0 boundary
    rows = []
    for a in (-2.0, -0.5, 0.0, 0.3, 1.1, 4.0):
        for b in (-2.0, -0.5, 0.0, 0.3, 1.0, 7.0):
            for u in (2.0 ** -53, 0.01, 0.37, 0.5, 0.93, 1 - 2.0 ** -53):
                rows.append((a, b, u, O.poisson_time(a, b, u)))
    for a in (0.7, 1.3):  # boundary -log u == a^2/(-2b)
        for b in (-0.4, -1.9):
            ustar = float(np.exp(a * a / (2 * b)))
            for u in (np.nextafter(ustar, 0), ustar, np.nextafter(ustar, 1)):
                rows.append((a, b, float(u), O.poisson_time(a, b, float(u))))
    out["poisson_table"] = np.array(rows)
    rows3 = []
    for a in (-1.0, 0.0, 0.8):
        for b in (-1.5, 0.0, 0.6):
            for c in (0.01, 0.5):
                for u in (0.05, 0.5, 0.95):
                    rows3.append((a, b, c, u, O.poisson_time3(a, b, c, u)))
    out["poisson3_table"] = np.array(rows3)

    # (2) config C1: 1-d ZigZag on N(0,1), x0=1.01, θ0=-1.5, c=10, T=1000 (shape of test/test1d.jl:13-15)
    ev, acc, num = O.pdmp_zigzag1d(0.0, 1.0, 1.01, -1.5, 1000.0, 10.0, seed=0x5EED0000)
    out["c1_events"] = np.stack([ev["t"], ev["x"], ev["theta"]], axis=1)
    out["c1_acc_num"] = np.array([acc, num])

    # (3) d=8 Γ = S S' (test/maintest.jl:6-8) and 8x8 grid-Laplace chains, T=50
    for name, G, scale in (("d8", P.maintest_precision(8), 0.9), ("grid8", P.gmrf_precision(8), 1.0)):
        d = G.shape[0]
        rng = np.random.default_rng(42)
        x0 = rng.random(d) if name == "d8" else rng.standard_normal(d)
        th0 = rng.choice([-1.0, 1.0], d)
        c = (0.7 if name == "d8" else 1.0) * P.column_norms(G)
        if name == "d8":
            c = 2.0 * c  # the reference's 0.7 factor violates the bound for this Γ draw; keep a valid bound
        r = O.spdmp_zigzag(scale * G, None, G, x0, th0, c, 50.0, seed=1234)
        assert r["status"] == 0
        out[f"{name}_x0"] = x0
        out[f"{name}_th0"] = th0
        out[f"{name}_c"] = c
        out[f"{name}_events"] = r["events"]
        out[f"{name}_acc"] = r["acc"]
        out[f"{name}_num"] = np.array([r["num"]])
        out[f"{name}_final"] = np.stack([r["t"], r["x"], r["theta"]])

    # (4) config C3 d=16384: 2 chains, synthetic initial state, first 10^4 events: index sequence + payload hash
    G = P.gmrf_precision(128)
    c = P.column_norms(G)
    for k in range(2):
        seed = 0x5EED0000 + k
        x0, th0 = O.synthetic_state(seed, G.shape[0])
        r = O.spdmp_zigzag(G, None, G, x0, th0, c, 1e9, seed=seed, max_events=10000)
        ev = r["events"][:10000]
        out[f"c3_chain{k}_idx"] = ev["i"].astype(np.uint16)
        out[f"c3_chain{k}_hash"] = np.array([payload_hash(ev)])
        out[f"c3_chain{k}_num"] = np.array([r["num"]])
        out[f"c3_chain{k}_tlast"] = np.array([ev["t"][-1]])

    # (5) BPS d=16 isotropic: first 200 events (time + hash of x, θ rows)
    d = 16
    import scipy.sparse as sp
    rng = np.random.default_rng(7)
    x0, th0 = rng.standard_normal(d), rng.standard_normal(d)
    r = O.pdmp_bps(sp.identity(d, format="csc"), None, x0, th0, 1e-3, 1e9, lambda_ref=1.0, seed=99, max_events=200,
                   ev_cap=200)
    out["bps16_x0"] = x0
    out["bps16_th0"] = th0
    out["bps16_t"] = r["t_ev"]
    out["bps16_x_last"] = r["x_ev"][-1]
    out["bps16_th_last"] = r["theta_ev"][-1]
    out["bps16_counts"] = np.array([r["num"], r["nacc"], r["nrefresh"]])

    # (6) sticky 1-d with the parameters of test/sticky.jl:7-36 (σ²=0.5, μ=0.9, flow Γ=[1], c=20, κ=1.5), T=200
    Gf = sp.csc_matrix(np.array([[1.0]]))
    Gt = sp.csc_matrix(np.array([[2.0]]))
    r = O.sspdmp_zigzag(Gf, np.array([0.0]), Gt, np.array([1.0]), np.array([0.8]), np.array([20.0]), np.array([1.5]),
                        200.0, target_mu=np.array([0.9]), seed=5)
    out["sticky1d_events"] = r["events"]
    out["sticky1d_counts"] = np.array([r["num"], r["nacc"]])

    # (7) exact diag(Γ⁻¹) of the 128x128 GMRF at 32 probe coordinates
    import scipy.sparse.linalg as spla
    lu = spla.splu(sp.csc_matrix(G))
    probes = np.linspace(0, G.shape[0] - 1, 32).astype(np.int64)
    dg = []
    for p in probes:
        e = np.zeros(G.shape[0])
        e[p] = 1.0
        dg.append(lu.solve(e)[p])
    out["c3_probe_idx"] = probes
    out["c3_probe_var"] = np.array(dg)

    np.savez_compressed(os.path.join(HERE, "golden.npz"), **out)
    print("wrote", os.path.join(HERE, "golden.npz"), os.path.getsize(os.path.join(HERE, "golden.npz")), "bytes")


if __name__ == "__main__":
    main()
