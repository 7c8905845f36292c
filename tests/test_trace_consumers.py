"""Host-side trace consumers (zigzagboomerang.jl_amd/trace.py, mirrors of src/trace.jl) on oracle-made traces (CPU only)."""
import numpy as np
import scipy.sparse as sp

import oracle_lib as O


def _fact_trace(pkg, T=200.0):
    G = pkg.problems.maintest_precision(8)
    rng = np.random.default_rng(3)
    x0, th0 = rng.random(8), rng.choice([-1.0, 1.0], 8)
    r = O.spdmp_zigzag(0.9 * G, None, G, x0, th0, 2.0 * pkg.problems.column_norms(G), T, seed=21)
    return pkg.FactTrace(pkg.ZigZag(0.9 * G, np.zeros(8)), 0.0, x0, th0, r["events"])


def test_fact_trace_mean_cummean_collect_discretize_subtrace(pkg):
    tr = _fact_trace(pkg)
    m = pkg.trace.mean(tr)  # src/trace.jl:182-200
    cm = pkg.trace.cummean(tr)  # :203-225: the last running value of coordinate i is ∫x_i up to ITS last event / (2 t)
    assert len(cm) == 8 and all(len(t) == len(y) for t, y in cm)
    T = tr.events["t"][-1]
    for i, (t, y) in enumerate(cm):
        assert t[0] == 0.0 and np.all(np.diff(t) > 0)
        assert abs(y[-1] * t[-1] / T - m[i]) < 1e-12  # same integral, normalised by T instead of the coordinate's last event time
    ts, xs = pkg.trace.collect(tr)  # :44-63, last event not applied
    assert len(ts) == len(tr.events) and np.all(np.diff(ts) >= 0)
    td, xd = pkg.trace.discretize(tr, 0.5)  # :94-125
    assert np.allclose(np.diff(td), 0.5) and xd.shape == (len(td), 8)
    J = np.array([1, 4, 6])
    t2, x2 = pkg.trace.discretize(pkg.trace.subtrace(tr, J), 0.5)  # :275-290, test/maintest.jl:52-57
    n = len(t2)
    assert np.allclose(t2, td[:n]) and np.allclose(x2, xd[:n][:, J])
    p = pkg.trace.inclusion_prob(tr)  # :161-178: a continuous ZigZag path is never exactly 0
    assert np.all(p > 0.9) and np.all(p <= 1.0)


def test_pdmp_trace_mean_cummean_discretize(pkg):
    d = 6
    rng = np.random.default_rng(4)
    G = sp.identity(d, format="csc")
    x0, th0 = rng.standard_normal(d), rng.standard_normal(d)
    r = O.pdmp_bps(G, None, x0, th0, 1e-3, 300.0, lambda_ref=1.0, seed=8, ev_cap=100000)
    B = pkg.BouncyParticle(G, np.zeros(d), 1.0)
    tr = pkg.PDMPTrace(B, 0.0, x0, th0, r["t_ev"], r["x_ev"], r["theta_ev"])
    cm = pkg.trace.cummean(tr)  # src/trace.jl:248-266
    m = pkg.trace.mean(tr)      # :229-246 (no ½ in the reference)
    assert cm.shape == (len(tr.t), d) and np.allclose(2.0 * cm[-1], m)
    assert np.all(np.abs(cm[-1]) < 0.5)  # N(0, I) target: the running mean is near 0 after T = 300
    td, xd = pkg.trace.discretize(tr, 0.25)
    assert np.allclose(np.diff(td), 0.25) and td[-1] < tr.t[-1] and xd.shape == (len(td), d)
    # between two events the bouncy particle moves on a straight line
    k = np.searchsorted(tr.t, td[10], side="right") - 1
    xk = (tr.x[k] if k >= 0 else x0) + (tr.θ[k] if k >= 0 else th0) * (td[10] - (tr.t[k] if k >= 0 else 0.0))
    assert np.allclose(xd[10], xk)
