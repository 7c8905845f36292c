"""The 1-d samplers on the device (pdmp_1d_run: one chain per lane) against the oracle, bit for bit (-m gpu): both
flows, the noisy gradient of test/test1d.jl:10, adaptation, the bound-violation stop, event buffers that fill up and are refilled."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def _same(ev, r):
    assert len(ev) == len(r["events"])
    for f in ("t", "x", "theta"):
        assert np.array_equal(ev[f], r["events"][f]), f


@pytest.mark.parametrize("flow,kw", [("zigzag", dict(noise=0.0)), ("zigzag", dict(noise=0.1)),
                                     ("boomerang", dict(boomerang=(1.0, 0.0, 1.0), noise=0.0)),
                                     ("boomerang", dict(boomerang=(1.1, 1.2, 0.5), noise=0.1))])
def test_ensemble_of_1d_chains_equals_the_oracle(gpu_pkg, flow, kw):
    pkg = gpu_pkg
    mu, s2, T = np.pi / 3, 1.3, 300.0
    rng = np.random.default_rng(17)
    n = 200  # four wavefronts, the last one partly filled
    x0 = rng.standard_normal(n) + 1.0
    th0 = np.where(rng.random(n) < 0.5, -1.5, 0.5)
    c = 10.0 if kw.get("noise") or flow == "zigzag" else 1.6
    F = pkg.ZigZag1d() if flow == "zigzag" else pkg.Boomerang1d(*kw["boomerang"])
    # a buffer of 64 events per chain: every chain is resumed several times
    Xi, ratio = pkg.pdmp(pkg.GaussianTarget1d(mu, s2, kw["noise"]), x0, th0, T, c, F, seed=900, trace_capacity=64)
    for k in (0, 1, 63, 64, 127, 199):
        r = O.pdmp_1d(mu, s2, float(x0[k]), float(th0[k]), T, c, flow=flow, seed=900 + k, **kw)
        _same(Xi[k], r)
        assert ratio[k] == r["acc"] / r["num"]
        assert Xi[k][0]["t"] == 0.0 and Xi[k][0]["x"] == x0[k] and Xi[k][-1]["t"] >= 0.0
    assert all(len(e) > 20 for e in Xi)


def test_scalar_call_has_the_references_shape(gpu_pkg):
    pkg = gpu_pkg
    Xi, ratio = pkg.pdmp(pkg.GaussianTarget1d(np.pi / 3, 1.3, 0.1), 1.01, -1.5, 1000.0, 10.0, pkg.ZigZag1d(), seed=3)  # test/test1d.jl:13-15
    r = O.pdmp_1d(np.pi / 3, 1.3, 1.01, -1.5, 1000.0, 10.0, flow="zigzag", noise=0.1, seed=3)
    _same(Xi, r)
    assert isinstance(ratio, float) and ratio == r["acc"] / r["num"]
    est = np.sum((Xi["x"][:-1] + Xi["x"][1:]) / 2 * np.diff(Xi["t"])) / 1000.0
    assert abs(est - np.pi / 3) < 0.2


def test_adaptation_and_the_violation_stop(gpu_pkg):
    pkg = gpu_pkg
    tgt = pkg.GaussianTarget1d(2.0, 1.0, 0.0)
    B = pkg.Boomerang1d(1.0, 0.0, 0.5)
    with pytest.raises(RuntimeError, match="Tuning parameter `c` too small."):
        pkg.pdmp(tgt, 3.0, 1.0, 5000.0, 1e-3, B, seed=11)
    Xi, ratio = pkg.pdmp(tgt, np.array([3.0, -1.0, 0.5]), 1.0, 5000.0, 1e-3, B, seed=11, adapt=True)
    for k, x0 in enumerate((3.0, -1.0, 0.5)):
        r = O.pdmp_1d(2.0, 1.0, x0, 1.0, 5000.0, 1e-3, flow="boomerang", boomerang=(1.0, 0.0, 0.5), seed=11 + k, adapt=True)
        _same(Xi[k], r)
        assert r["c"] > 1e-3


def test_c_abi_state_round_trip_and_argument_checks(gpu_pkg):
    """The entry point itself: state in / state out (counters, draws consumed, the adapted c), per-chain c, and refusals."""
    pkg = gpu_pkg
    L = pkg._lib.load()
    n, cap = 3, 4096
    cfg = pkg._lib.Config1d(C.sizeof(pkg._lib.Config1d), 0, 0, 1, 2.0, n, cap, 0.5, 1.3, 0.0, 1.0, 0.0, 1.0)
    st = np.zeros(n, dtype=pkg._lib.STATE1D_DTYPE)
    st["x"], st["theta"], st["c"] = [1.0, -2.0, 0.3], [1.0, -1.0, 1.0], [0.01, 5.0, 1.0]
    seeds = np.array([5, 6, 7], dtype=np.uint64)
    ev = np.empty((n, cap), dtype=pkg._lib.EVENT1D_DTYPE)
    nev = np.zeros(n, dtype=np.int64)
    assert L.pdmp_1d_run(C.byref(cfg), st.ctypes.data, seeds.ctypes.data, 100.0, ev.ctypes.data, nev.ctypes.data) == 0
    for k in range(n):
        r = O.pdmp_1d(0.5, 1.3, [1.0, -2.0, 0.3][k], [1.0, -1.0, 1.0][k], 100.0, [0.01, 5.0, 1.0][k], seed=5 + k, adapt=True)
        _same(ev[k, :nev[k]], r)
        assert st["num"][k] == r["num"] and st["acc"][k] == r["acc"] and st["ndraw"][k] == r["ndraw"] and st["c"][k] == r["c"]
        assert st["t"][k] == r["t"] and st["x"][k] == r["x"] and st["theta"][k] == r["theta"] and st["status"][k] == 0
    bad = pkg._lib.Config1d(C.sizeof(pkg._lib.Config1d), 0, 7, 0, 2.0, n, cap, 0.5, 1.3, 0.0, 1.0, 0.0, 1.0)
    assert L.pdmp_1d_run(C.byref(bad), st.ctypes.data, seeds.ctypes.data, 1.0, ev.ctypes.data, nev.ctypes.data) == pkg._lib.PDMP_ERR_INVALID
    bad = pkg._lib.Config1d(C.sizeof(pkg._lib.Config1d), 0, 1, 0, 2.0, n, cap, 0.5, 1.3, 0.0, 1.0, 0.0, 0.0)  # Boomerang1d with λref = 0
    assert L.pdmp_1d_run(C.byref(bad), st.ctypes.data, seeds.ctypes.data, 1.0, ev.ctypes.data, nev.ctypes.data) == pkg._lib.PDMP_ERR_INVALID
    bad = pkg._lib.Config1d(4, 0, 0, 0, 2.0, n, cap, 0.5, 1.3, 0.0, 1.0, 0.0, 1.0)
    assert L.pdmp_1d_run(C.byref(bad), st.ctypes.data, seeds.ctypes.data, 1.0, ev.ctypes.data, nev.ctypes.data) == pkg._lib.PDMP_ERR_INVALID
