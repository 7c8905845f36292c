"""State left behind by a bound violation (`adapt = false`, reference: error(...)): the speculative kernels, the
one-event kernels and the oracle agree on counters, traces and positions (-m gpu)."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def _run(pkg, monkeypatch, mode, G, Gb, x0, th0, c, T, seeds, kappa=None):
    if mode is None:
        monkeypatch.delenv("PDMP_KERNEL", raising=False)
    else:
        monkeypatch.setenv("PDMP_KERNEL", mode)
    nch, d = x0.shape
    sampler = pkg._lib.SAMPLER_STICKY_ZIGZAG if kappa is not None else pkg._lib.SAMPLER_ZIGZAG_LOCAL
    with pkg.Ensemble(nch, d, trace_capacity=50000, sampler=sampler) as ens:
        ens.set_flow(pkg.ZigZag(Gb, np.zeros(d)))
        ens.set_target(pkg.GaussianTarget(G))
        if kappa is not None:
            ens.set_sticky(kappa, False, False)
        ens.set_state(0.0, x0, th0, c, seeds)
        ens.run(T, pkg._lib.RUN_REFERENCE_TAIL)
        cnt = ens.counters()
        return cnt, [ens.trace(k, counters=cnt) for k in range(nch)], ens.final_state()


@pytest.mark.parametrize("sticky", [False, True])
def test_violation_state_small_lattice(gpu_pkg, monkeypatch, sticky):
    pkg = gpu_pkg
    G = pkg.problems.gmrf_precision(8)
    d = 64
    rng = np.random.default_rng(5)
    x0 = 3.0 * rng.standard_normal((6, d))
    th0 = rng.choice([-1.0, 1.0], (6, d))
    c = np.full(d, 1e-4)
    seeds = np.arange(300, 306, dtype=np.uint64)
    kappa = np.full(d, 0.7) if sticky else None
    res = {m: _run(pkg, monkeypatch, m, G, 0.4 * G, x0, th0, c, 50.0, seeds, kappa) for m in (None, "seq")}
    cnt_s, ev_s, fs_s = res[None]
    cnt_q, ev_q, fs_q = res["seq"]
    assert np.any(cnt_s["status"] == pkg._lib.CHAIN_BOUND_VIOLATED)
    for f in ("status", "num", "nacc", "nevents", "ndraw_main", "t_last"):
        assert np.array_equal(cnt_s[f], cnt_q[f]), f
    for k in range(6):
        assert np.array_equal(ev_s[k], ev_q[k]), k
    for f in ("t", "x", "theta"):
        assert np.array_equal(fs_s[f], fs_q[f]), f
    for k in np.flatnonzero(cnt_s["status"] == pkg._lib.CHAIN_BOUND_VIOLATED)[:2]:
        if sticky:
            r = O.sspdmp_zigzag(0.4 * G, None, G, x0[k], th0[k], c, kappa, 50.0, seed=300 + int(k))
        else:
            r = O.spdmp_zigzag(0.4 * G, None, G, x0[k], th0[k], c, 50.0, seed=300 + int(k))
        assert r["status"] != 0 and int(cnt_s["num"][k]) == r["num"] and len(ev_s[k]) == len(r["events"])
        assert np.array_equal(fs_s["x"][k], r["x"]) and np.array_equal(fs_s["t"][k], r["t"])
