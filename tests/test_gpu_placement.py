"""set_state's placement probes (-m gpu): a device-filling ZigZag ensemble is timed on a short launch, its pairs / keys and records are re-allocated
and the fastest combination kept (csrc/pdmp_capi.hip: init_state_tuned; DESIGN.md 5 "The timing modes are a property of the allocation").  What the
caller gets must be bit for bit the ensemble set_state alone makes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(pkg, tracked, tune, nch, n, T, explicit_state):
    G = pkg.problems.gmrf_precision(n)
    d = G.shape[0]
    c = pkg.problems.column_norms(G)
    with pkg.Ensemble(nch, d, trace_capacity=64) as ens:
        ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
        ens.set_target(pkg.GaussianTarget(G))
        if tracked:
            ens.set_gradient_tracking(True)
        ens.debug_set_placement(tune=tune)
        if explicit_state:
            rng = np.random.default_rng(5)
            x0 = rng.standard_normal((nch, d))
            th0 = rng.choice([-1.0, 1.0], (nch, d))
            ens.set_state(0.0, x0, th0, c, np.arange(nch, dtype=np.uint64) + 77)
        else:
            ens.set_state_synthetic(0.0, c, 0xABCD)
        log = ens.debug_placement()
        assert ens.kernel_name() == ""  # (the probes' launches are not the caller's)
        cnt0 = ens.counters()
        assert int(cnt0["num"].sum()) == 0 and int(cnt0["ntrace"].sum()) == 0 and np.all(cnt0["status"] == pkg._lib.CHAIN_OK)
        while True:
            ens.run(T, pkg._lib.RUN_STOP_BEFORE)
            cnt = ens.counters()
            ens.trace_reset()
            if not pkg._lib.needs_rerun(cnt["status"]):
                break
        fs = ens.final_state()
        return log, cnt, fs, ens.kernel_name()


@pytest.mark.parametrize("tracked,explicit_state", [(True, False), (True, True), (False, False)])
def test_probed_placement_leaves_the_ensemble_as_set_state_makes_it(gpu_pkg, tracked, explicit_state):
    pkg = gpu_pkg
    nch, n, T = 4096, 96, 0.05  # records: 4096 x 9216 x 128 B (tracked) / 64 B: above the 2 GB from which set_state probes
    log1, cnt1, fs1, k1 = _run(pkg, tracked, 1, nch, n, T, explicit_state)
    log0, cnt0, fs0, k0 = _run(pkg, tracked, 0, nch, n, T, explicit_state)
    assert "placement probes" in log1 and "kept r" in log1, log1
    assert "placement probes" not in log0, log0
    assert k0 == k1 and ("trackp" in k1) == tracked
    for f in ("num", "nacc", "nevents", "ndraw_main", "status", "t_last"):
        assert np.array_equal(cnt1[f], cnt0[f]), f
    assert int(cnt1["nacc"].sum()) > 10 * nch
    for f in ("x", "theta", "t", "acc"):
        assert np.array_equal(fs1[f], fs0[f]), f


def test_small_ensembles_are_not_probed(gpu_pkg):
    pkg = gpu_pkg
    log, cnt, fs, k = _run(pkg, True, 1, 256, 48, 0.5, False)
    assert "placement probes" not in log


def test_placement_arguments_are_checked(gpu_pkg):
    pkg = gpu_pkg
    with pkg.Ensemble(2, 16) as ens:
        with pytest.raises(pkg._lib.PdmpError):
            ens.debug_set_placement(tune=2)
        with pytest.raises(pkg._lib.PdmpError):
            ens.debug_set_placement(place=1, rec=b"013")
        ens.debug_set_placement(tune=0, place=0)


def _run_bps(pkg, tune):
    import scipy.sparse as sp
    nch, d, cap = 1024, 1024, 256  # 2 GB per event array: from there set_state_bps probes
    rng = np.random.default_rng(9)
    with pkg.Ensemble(nch, d, sampler=pkg._lib.SAMPLER_BPS, factor=2.0, trace_capacity=cap) as ens:
        ens.set_flow_bps(pkg.BouncyParticle(sp.identity(d, format="csc"), np.zeros(d), 1.0))
        ens.debug_set_placement(tune=tune)
        ens.set_state_bps(0.0, rng.standard_normal((nch, d)), rng.standard_normal((nch, d)), 1e-3, np.arange(nch, dtype=np.uint64) + 5)
        log = ens.debug_placement()
        cnt0 = ens.counters()
        assert int(cnt0["nevents"].sum()) <= nch and np.all(cnt0["status"] == pkg._lib.CHAIN_OK)
        ens.run(3.0, pkg._lib.RUN_STOP_BEFORE)
        cnt = ens.counters()
        t, x, th = ens.bps_trace(7, counters=cnt) if hasattr(ens, "bps_trace") else (None, None, None)
        return log, cnt, (t, x, th)


def test_bps_placement_probes_leave_the_ensemble_as_set_state_makes_it(gpu_pkg):
    pkg = gpu_pkg
    log1, cnt1, tr1 = _run_bps(pkg, 1)
    log0, cnt0, tr0 = _run_bps(pkg, 0)
    assert "placement probes" in log1 and "kept x" in log1, log1
    assert "placement probes" not in log0
    for f in ("num", "nacc", "nevents", "ntrace", "status", "t_last"):
        assert np.array_equal(cnt1[f], cnt0[f]), f
    assert int(cnt1["nevents"].sum()) > 5 * 1024
    for a, b in zip(tr1, tr0):
        if a is not None:
            assert np.array_equal(a, b)
