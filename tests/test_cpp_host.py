"""The C++ host mirror (include/pdmp_mi355.hpp) and its example program: compile check here, parity on the GPU box."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _exe(pkg):
    exes = pkg.build.build_examples()
    exe = [e for e in exes if e.endswith("gmrf_spdmp")]
    assert exe and os.access(exe[0], os.X_OK)
    return exe[0]


def _fnv1a(h, data):
    for b in data:
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_cpp_example_builds_and_fails_loudly_without_a_device(pkg):
    """g++ -std=c++17 against the header; without a gfx950 device the program must exit non-zero with the ABI's error text
    (no CPU fallback).  On a GPU box this test only checks the build."""
    exe = _exe(pkg)
    if pkg._lib.device_count() > 0:
        return
    p = subprocess.run([exe, "4", "1"], capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "no CPU fallback" in p.stderr


@pytest.mark.gpu
def test_cpp_spdmp_matches_oracle(gpu_pkg):
    """examples/gmrf_spdmp.cpp (scripts/gaussianrandomfield.jl through pdmp::spdmp) on the device vs the CPU oracle: event count,
    counters and an FNV-1a of every event + the final (x, θ, t)."""
    pkg = gpu_pkg
    exe = _exe(pkg)
    n, T, seed = 16, 20.0, 0x1234
    p = subprocess.run([exe, str(n), repr(T), hex(seed)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    d_s, nev_s, num_s, acc_s, h_s, tl_s = p.stdout.split()
    G = pkg.problems.gmrf_precision(n, eps=0.01)
    d = n * n
    i = np.arange(d)
    x0 = ((i * 37) % 101) / 50.0 - 1.0
    th0 = np.where(i % 3 == 0, -1.0, 1.0)
    Gc = G.tocsc()
    c = np.array([np.sqrt(sum(v * v for v in Gc.data[Gc.indptr[k]:Gc.indptr[k + 1]])) for k in range(d)])
    r = O.spdmp_zigzag(G, None, G, x0, th0, c, T, seed=seed)
    assert r["status"] == 0
    ev = r["events"]
    assert int(d_s) == d and int(nev_s) == len(ev) and int(num_s) == r["num"] and int(acc_s) == int(r["acc"].sum())
    h = 14695981039346656037
    h = _fnv1a(h, ev.tobytes())  # records are (t, i, x, theta), 32 B, the order the program hashes
    h = _fnv1a(h, r["x"].tobytes() + r["theta"].tobytes() + r["t"].tobytes())
    assert int(h_s, 16) == h
    assert float(tl_s) == ev["t"][-1]


@pytest.mark.gpu
def test_cpp_spdmp_tracked_option(gpu_pkg):
    """Options::tracked through the C++ mirror on a 48 x 48 lattice: the same number of events, proposals and accepted reflections as the
    oracle and the same last event time to 1e-9 (the tracked evaluation is index-exact, not bit-exact: the payload hash differs from
    the moving evaluation's, which the second run reproduces)."""
    pkg = gpu_pkg
    exe = _exe(pkg)
    n, T, seed = 48, 4.0, 0x77
    out = {}
    for mode in ("tracked", "exact"):
        args = [exe, str(n), repr(T), hex(seed)] + (["tracked"] if mode == "tracked" else [])
        p = subprocess.run(args, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr
        out[mode] = p.stdout.split()
    G = pkg.problems.gmrf_precision(n, eps=0.01)
    d = n * n
    i = np.arange(d)
    x0 = ((i * 37) % 101) / 50.0 - 1.0
    th0 = np.where(i % 3 == 0, -1.0, 1.0)
    Gc = G.tocsc()
    c = np.array([np.sqrt(sum(v * v for v in Gc.data[Gc.indptr[k]:Gc.indptr[k + 1]])) for k in range(d)])
    r = O.spdmp_zigzag(G, None, G, x0, th0, c, T, seed=seed)
    for mode in ("tracked", "exact"):
        d_s, nev_s, num_s, acc_s, h_s, tl_s = out[mode]
        assert int(d_s) == d and int(nev_s) == len(r["events"]) and int(num_s) == r["num"] and int(acc_s) == int(r["acc"].sum())
        assert abs(float(tl_s) - r["events"]["t"][-1]) <= 1e-9 * r["events"]["t"][-1]
    assert float(out["exact"][5]) == r["events"]["t"][-1] and out["exact"][4] != out["tracked"][4]


@pytest.mark.gpu
def test_cpp_parallel_spdmp_matches_oracle(gpu_pkg):
    """pdmp::parallel_spdmp (src/parallel.jl through the C++ mirror, 4 chunks = 4 wavefronts) against the oracle's threaded restatement:
    event count, (acc, num) and the FNV-1a of the time-sorted events + final (x, θ, t)."""
    import scipy.sparse as sp
    pkg = gpu_pkg
    exe = _exe(pkg)
    n, T, seed, K = 16, 5.0, 0x4321, 4
    p = subprocess.run([exe, str(n), repr(T), hex(seed), "parallel:%d" % K], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    d_s, nev_s, num_s, acc_s, h_s, tl_s = p.stdout.split()
    G = pkg.problems.gmrf_precision(n, eps=0.01)
    d = n * n
    i = np.arange(d)
    x0 = ((i * 37) % 101) / 50.0 - 1.0
    th0 = np.where(i % 3 == 0, -1.0, 1.0)
    Gc = G.tocsc()
    c = 2.0 * np.array([np.sqrt(sum(v * v for v in Gc.data[Gc.indptr[k]:Gc.indptr[k + 1]])) for k in range(d)])
    coo = sp.coo_matrix(G)
    keep = (coo.row // (d // K)) == (coo.col // (d // K))
    G2 = sp.csc_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=G.shape)
    G2.sort_indices()
    r = O.parallel_spdmp(G2, None, G, x0, th0, c, T, K, 0.1, seed=seed)
    assert r["status"] == 0 and len(r["events"]) > 500
    ev = r["events"]
    assert int(d_s) == d and int(nev_s) == len(ev) and int(num_s) == r["num"] and int(acc_s) == r["nacc"]
    h = _fnv1a(14695981039346656037, ev.tobytes())
    h = _fnv1a(h, r["x"].tobytes() + r["theta"].tobytes() + r["t"].tobytes())
    assert int(h_s, 16) == h and float(tl_s) == ev["t"][-1]


@pytest.mark.gpu
def test_cpp_1d_sampler_matches_oracle(gpu_pkg):
    """pdmp::pdmp(GaussianTarget1d, x0, θ0, T, c, Boomerang1d) of the C++ mirror (examples/gmrf_spdmp.cpp, mode `1d`): every event of every
    chain, hashed chain by chain, equals the oracle's; the event buffers hold 50 events, so the runs are resumed dozens of times."""
    pkg = gpu_pkg
    exe = _exe(pkg)
    n, T, seed = 70, 400.0, 0x77
    p = subprocess.run([exe, str(n), repr(T), hex(seed), "1d"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    n_s, tot_s, h_s, acc_s = p.stdout.split()
    h, total, acc0 = 14695981039346656037, 0, None
    for k in range(n):
        r = O.pdmp_1d(3.14159265358979323846 / 3, 1.3, 1.41 + 0.01 * k, 0.5, T, 10.0, flow="boomerang", boomerang=(1.1, 1.2, 0.5), noise=0.1,
                      seed=seed + k)
        assert r["status"] == 0
        total += len(r["events"])
        h = _fnv1a(h, r["events"].tobytes())
        if k == 0:
            acc0 = r["acc"] / r["num"]
    assert int(n_s) == n and int(tot_s) == total and int(h_s, 16) == h and float(acc_s) == acc0
