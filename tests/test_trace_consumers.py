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
    assert len(ts) == 1 + len(tr.events) == len(tr) and np.all(np.diff(ts) >= 0)  # Base.length(FT), src/trace.jl:42
    assert ts[-1] == ts[-2] and np.array_equal(xs[-1], xs[-2])  # the last event is never applied (:56)
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


def _sequential_collect(tr, boom):
    """Base.iterate(FT::FactTrace) written as the reference writes it (src/trace.jl:44-63): move ALL coordinates to the event time
    (move_forward!: linear, src/dynamics.jl:11-15, or the rotation about μ, :29-36), then overwrite x[i], θ[i]; the last event is
    not applied and the final state is yielded twice."""
    t, x, th = tr.t0, tr.x0.copy(), tr.θ0.copy()
    ts, xs = [t], [x.copy()]
    ev = tr.events
    for k in range(len(ev) - 1):
        t2, i, xi, thi = ev[k]
        if boom:
            mu, s, c = tr.F.μ, np.sin(t2 - t), np.cos(t2 - t)
            x, th = (x - mu) * c + th * s + mu, -(x - mu) * s + th * c
        else:
            x = x + th * (t2 - t)
        t = t2
        x[i], th[i] = xi, thi
        ts.append(t)
        xs.append(x.copy())
    ts.append(t)
    xs.append(x.copy())
    return np.array(ts), np.array(xs)


def _sequential_discretize(tr, dt, boom):
    """Base.iterate(D::Discretize{<:FactTrace}) as written (src/trace.jl:106-125)."""
    ev = tr.events
    t, x, th = tr.t0, tr.x0.copy(), tr.θ0.copy()
    ts, xs = [t], [x.copy()]
    k, n = 0, len(ev)

    def flow(x, th, tau):
        if boom:
            mu, s, c = tr.F.μ, np.sin(tau), np.cos(tau)
            return (x - mu) * c + th * s + mu, -(x - mu) * s + th * c
        return x + th * tau, th

    while True:
        step, done = dt, False
        while True:
            if k >= n:
                done = True
                break
            ti = ev["t"][k]
            if t + step < ti:
                x, th = flow(x, th, step)
                t += step
                break
            x, th = flow(x, th, ti - t)
            step -= ti - t
            t = ti
            x[ev["i"][k]], th[ev["i"][k]] = ev["x"][k], ev["theta"][k]
            k += 1
        if done:
            break
        ts.append(t)
        xs.append(x.copy())
    return np.array(ts), np.array(xs)


def test_vectorised_consumers_equal_the_reference_loops(pkg):
    """The closed-form, per-coordinate consumers against event-by-event restatements of src/trace.jl, for a ZigZag trace and for a
    FactBoomerang trace (whose path ROTATES between events: collect must not move it linearly)."""
    tr = _fact_trace(pkg, T=60.0)
    ts, xs = pkg.trace.collect(tr)
    ts0, xs0 = _sequential_collect(tr, False)
    assert np.array_equal(ts, ts0) and np.allclose(xs, xs0, rtol=0, atol=1e-11)
    td, xd = pkg.trace.discretize(tr, 0.37)
    td0, xd0 = _sequential_discretize(tr, 0.37, False)
    assert len(td) == len(td0) and np.allclose(td, td0, rtol=0, atol=1e-10) and np.allclose(xd, xd0, rtol=0, atol=1e-9)
    # mean / inclusion_prob / moments against the per-event loops of src/trace.jl:161-200
    ev = tr.events
    x, t, y, p = tr.x0.copy(), np.full(8, tr.t0), np.zeros(8), np.zeros(8)
    T = ev["t"][-1]
    for t2, i, xi, _ in ev:
        y[i] += (x[i] + xi) * (t2 - t[i]) * (1 / (2 * T))
        p[i] += ((x[i] != 0) | (xi != 0)) * (t2 - t[i]) / T
        t[i], x[i] = t2, xi
    assert np.array_equal(pkg.trace.mean(tr), y) and np.array_equal(pkg.trace.inclusion_prob(tr), p)
    m, v = pkg.trace.moments(tr, 50.0)
    grid, xg = pkg.trace.discretize(tr, 0.001)
    sel = grid < 50.0
    assert np.allclose(m, xg[sel].mean(0), atol=2e-3) and np.allclose(v, xg[sel].var(0), atol=5e-3)
    # FactBoomerang: events from the oracle, rotation between events
    G = pkg.problems.maintest_precision(8)
    rng = np.random.default_rng(5)
    x0, th0 = rng.standard_normal(8), rng.standard_normal(8)
    F = pkg.FactBoomerang(sp.csc_matrix(1.2 * G), 0.1 * rng.standard_normal(8), 0.3)
    r = O.spdmp_zigzag(F.Γ, F.μ, G, x0, th0, 2.0 * pkg.problems.column_norms(G), 40.0, seed=4, lambda_ref=0.3, sigma=F.σ,
                       factboomerang=True, adapt=True)
    trb = pkg.FactTrace(F, 0.0, x0, th0, r["events"])
    assert len(trb.events) > 50
    ts, xs = pkg.trace.collect(trb)
    ts0, xs0 = _sequential_collect(trb, True)
    assert np.array_equal(ts, ts0) and np.allclose(xs, xs0, rtol=0, atol=1e-10)
    assert not np.allclose(xs, _sequential_collect(trb, False)[1], atol=1e-3)  # a linear move is a different path
    td, xd = pkg.trace.discretize(trb, 0.25)
    td0, xd0 = _sequential_discretize(trb, 0.25, True)
    assert len(td) == len(td0) and np.allclose(xd, xd0, rtol=0, atol=1e-9)


def test_host_consumers_equal_the_oracle_restatement_of_trace_jl(pkg):
    """zigzagboomerang.jl_amd/trace.py against oracle/trace_oracle.c -- the event-by-event loops of src/trace.jl:100-125,161-226,275-290 in C, one
    hop from the reference -- on an oracle-made ZigZag trace: mean, inclusion_prob, cummean, collect(discretize), subtrace."""
    tr = _fact_trace(pkg, T=600.0)
    ev = tr.events
    assert len(ev) > 2000
    assert np.allclose(pkg.trace.mean(tr), O.trace_mean(tr.t0, tr.x0, ev), rtol=1e-13, atol=0)
    assert np.allclose(pkg.trace.inclusion_prob(tr), O.trace_inclusion_prob(tr.t0, tr.x0, ev), rtol=1e-13, atol=0)
    ot, oy = O.trace_cummean(tr.t0, tr.x0, ev)
    cm = pkg.trace.cummean(tr)
    for j in range(tr.x0.size):
        own = np.nonzero(ev["i"] == j)[0]
        t, y = cm[j]
        assert t[0] == tr.t0 and y[0] == tr.x0[j]
        assert np.array_equal(t[1:], ot[own]) and np.allclose(y[1:], oy[own], rtol=1e-12, atol=1e-15)
    for dt in (0.37, 1.0):
        td, xd = pkg.trace.discretize(tr, dt)
        ts, xs = O.trace_discretize(tr.t0, tr.x0, tr.θ0, ev, dt)
        assert len(td) == len(ts) and np.allclose(td, ts, rtol=0, atol=1e-10) and np.allclose(xd, xs, rtol=0, atol=1e-9)
    J = np.array([0, 3, 4, 7])
    sub = pkg.trace.subtrace(tr, J)
    ok, oi = O.trace_subtrace(J, ev)
    assert np.array_equal(sub.events["i"], oi) and np.array_equal(sub.events["t"], ev["t"][ok]) and np.array_equal(sub.events["x"], ev["x"][ok])
    assert np.array_equal(sub.x0, tr.x0[J]) and np.array_equal(sub.θ0, tr.θ0[J])
