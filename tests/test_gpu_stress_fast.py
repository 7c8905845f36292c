"""Randomised stress of the FAST kernels (-m gpu; d >= 2048, where the 8-event kernels and the one-proposal-per-lane tracked kernel take over): random
lattice sides, a flow mean or none, a target mean (the same or none), speeds that are not one, a start time, adapt, a refresh clock (moving evaluation
only), random slice boundaries, trace buffers that fill inside a launch, every form of the tracked kernel -- each chain bit for bit the oracle of its
evaluation.  Round 6 added it after a combination no test had (a flow mean without a target mean under gradient tracking) turned out mis-bounded.
Seeds are fixed: the cases are reproducible."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", range(30))
def test_random_options_on_the_fast_kernels(gpu_pkg, monkeypatch, case):
    pkg = gpu_pkg
    L = pkg._lib
    rng = np.random.default_rng(7000 + case)
    n = int(rng.integers(46, 62))
    graph = ["lattice", "lattice", "random6", "lattice3d"][int(rng.integers(0, 4))]
    if graph == "lattice":
        G = pkg.problems.gmrf_precision(n, eps=float(rng.uniform(0.01, 0.3)))
    elif graph == "random6":
        G = pkg.problems.random_sparse_precision(int(rng.integers(2100, 3000)), 6, seed=int(rng.integers(1, 100)))
    else:
        G = pkg.problems.lattice3d_precision(int(rng.integers(13, 15)))
    d = G.shape[0]
    tracked = bool(rng.integers(0, 2))
    form = ["one_wave", "two_waves", "lines"][int(rng.integers(0, 3))]
    monkeypatch.setenv("PDMP_HELPER_WAVE", "1" if form == "two_waves" else "0")
    monkeypatch.setenv("PDMP_TRACK_LINES", "1" if form == "lines" else "0")
    mu = 0.4 * rng.standard_normal(d) if rng.integers(0, 2) else None
    tmu = mu if (mu is not None and rng.integers(0, 2)) else None
    sig = (0.5 + rng.random(d)) if rng.integers(0, 2) else np.ones(d)
    lam = float(rng.uniform(1.0, 4.0)) if (not tracked and rng.integers(0, 2)) else 0.0
    adapt = bool(rng.integers(0, 2))
    t0 = float(rng.uniform(0.0, 3.0)) if rng.integers(0, 2) else 0.0
    nch = 2
    x0 = rng.standard_normal((nch, d))
    th0 = sig * rng.choice([-1.0, 1.0], (nch, d))
    c = float(rng.uniform(3.0, 5.0)) * pkg.problems.column_norms(G)
    T = t0 + float(rng.uniform(0.8, 2.0))
    cap = int(rng.integers(300, 2500))
    seeds = [8100 + 10 * case + k for k in range(nch)]
    cuts = np.sort(rng.uniform(t0, T, size=int(rng.integers(0, 4))))
    kw = dict(t0=t0, target_mu=tmu, adapt=adapt, factor=1.8, sigma=sig)
    if lam > 0.0:
        kw["lambda_ref"] = lam
    bmu = mu if mu is not None else None
    refs = [O.spdmp_zigzag(G, bmu, G, x0[k], th0[k], c, T, seed=seeds[k], tracked=tracked, **kw) for k in range(nch)]
    assert all(r["status"] == 0 for r in refs)
    evs = [[] for _ in range(nch)]
    with pkg.Ensemble(nch, d, adapt=adapt, factor=1.8, trace_capacity=cap) as ens:
        ens.set_flow(pkg.ZigZag(G, np.zeros(d) if mu is None else mu, sig, λref=lam))
        ens.set_target(pkg.GaussianTarget(G) if tmu is None else pkg.GaussianTarget(G, tmu))
        if tracked:
            ens.set_gradient_tracking(True)
        ens.set_state(t0, x0, th0, c, seeds)
        for Tk, flag in [(float(v), L.RUN_STOP_BEFORE) for v in cuts] + [(T, L.RUN_REFERENCE_TAIL)]:
            while True:
                ens.run(Tk, flag)
                cnt = ens.counters()
                assert not np.any(cnt["status"] == L.CHAIN_BOUND_VIOLATED)
                for k in range(nch):
                    evs[k].append(ens.trace(k, counters=cnt))
                ens.trace_reset()
                if not L.needs_rerun(cnt["status"]):
                    break
        kname = ens.kernel_name()
        fs = ens.final_state()
    what = dict(case=case, graph=graph, d=d, tracked=tracked, form=form, flow_mean=mu is not None, target_mean=tmu is not None, lam=lam, adapt=adapt, t0=t0, cap=cap,
                cuts=len(cuts), kernel=kname)
    assert kname.startswith("zz_local_track") if tracked else kname.startswith("zz_local_spec"), what
    for k in range(nch):
        r = refs[k]
        ev = np.concatenate(evs[k])
        assert len(ev) == len(r["events"]) > 500, (what, len(ev), len(r["events"]))
        for f in ("i", "t", "x", "theta"):
            assert np.array_equal(ev[f], r["events"][f]), (what, k, f)
        assert int(cnt["num"][k]) == r["num"], what
        for f, g in (("t", "t"), ("x", "x"), ("theta", "theta"), ("acc", "acc")):
            assert np.array_equal(fs[f][k], r[g]), (what, k, f)
        if adapt:
            assert np.array_equal(fs["c"][k], r["c"]), (what, k)
