"""Trace consumers, mirroring src/trace.jl for FactTrace (what every caller of spdmp does next with Ξ).

Events are structured arrays (t, i, x, theta) with 0-based i.  Vectorised numpy, host side only: these
are the callers' post-processing, not part of the device hot path.
"""
import numpy as np

from .flows import Boomerang, FactBoomerang, FactTrace, PDMPTrace


def _flow(tr, x, th, dt):
    """move_forward!(dt, ...) in place: linear for ZigZag (src/dynamics.jl:11-15), rotation about μ for FactBoomerang (:29-36)."""
    if isinstance(tr.F, FactBoomerang):
        mu = tr.F.μ
        s, c = np.sin(dt), np.cos(dt)
        xn = (x - mu) * c + th * s + mu
        th[:] = -(x - mu) * s + th * c
        x[:] = xn
    else:
        x += th * dt


def collect(tr: FactTrace):
    """collect(Ξ): list of (t, x) at event times -- Base.iterate(FT::FactTrace), src/trace.jl:44-63.

    Like the reference, the LAST event is not applied (:56)."""
    t, x, th = tr.t0, tr.x0.copy(), tr.θ0.copy()
    ts, xs = [t], [x.copy()]
    ev = tr.events
    for k in range(len(ev) - 1):
        t2, i, xi, thi = ev[k]
        x += th * (t2 - t)  # move_forward!, src/dynamics.jl:11-15
        t = t2
        x[i] = xi
        th[i] = thi
        ts.append(t)
        xs.append(x.copy())
    return np.array(ts), np.array(xs)


def _discretize_pdmp(tr: PDMPTrace, dt):
    """collect(discretize(Ξ::PDMPTrace, dt)) -- src/trace.jl:102-106,129-150: the grid t0, t0+dt, ... up to (excluding) the
    last event time; between events the state flows from the latest event (linear, or the Boomerang rotation about μ).
    Closed form per grid point (the reference accumulates dt steps)."""
    d = len(tr.x0)
    if len(tr.t) == 0:
        return np.array([tr.t0]), tr.x0[None].copy()
    n = int(np.ceil((tr.t[-1] - tr.t0) / dt))
    grid = tr.t0 + dt * np.arange(n)
    grid = grid[grid < tr.t[-1]]
    te = np.concatenate([[tr.t0], tr.t])
    X = np.vstack([tr.x0[None], tr.x.reshape(-1, d)])
    TH = np.vstack([tr.θ0[None], tr.θ.reshape(-1, d)])
    idx = np.searchsorted(te, grid, side="right") - 1
    tau = (grid - te[idx])[:, None]
    if isinstance(tr.F, Boomerang):
        mu = tr.F.μ
        xs = (X[idx] - mu) * np.cos(tau) + TH[idx] * np.sin(tau) + mu
    else:
        xs = X[idx] + TH[idx] * tau
    return grid, xs


def discretize(tr, dt):
    """collect(discretize(Ξ, dt)): positions on the grid t0, t0+dt, ... -- src/trace.jl:94-125 (FactTrace), :129-150 (PDMPTrace)."""
    if isinstance(tr, PDMPTrace):
        return _discretize_pdmp(tr, dt)
    ev = tr.events
    t, x, th = tr.t0, tr.x0.copy(), tr.θ0.copy()
    ts, xs = [t], [x.copy()]
    k = 0
    n = len(ev)
    while True:
        step = dt
        done = False
        while True:
            if k >= n:
                done = True
                break
            ti = ev["t"][k]
            if t + step < ti:
                _flow(tr, x, th, step)
                t += step
                break
            d_t = ti - t
            step -= d_t
            _flow(tr, x, th, d_t)
            t = ti
            i = ev["i"][k]
            x[i] = ev["x"][k]
            th[i] = ev["theta"][k]
            k += 1
        if done:
            break
        ts.append(t)
        xs.append(x.copy())
    return np.array(ts), np.array(xs)


def _mean_pdmp(tr: PDMPTrace):
    """Statistics.mean(Ξ::PDMPTrace) -- src/trace.jl:229-246, restated as written: Σ (x + x₂)(t₂ − t) over consecutive events divided
    by the LAST event time T -- the reference omits the ½ of the trapezoid rule here (its cummean, :248-266, has it), so this is
    twice the time average of the interpolated path."""
    d = len(tr.x0)
    X = np.vstack([tr.x0[None], np.asarray(tr.x).reshape(-1, d)])
    te = np.concatenate([[tr.t0], tr.t])
    y = ((X[:-1] + X[1:]) * np.diff(te)[:, None]).sum(0)
    return y / tr.t[-1]


def cummean(tr):
    """cummean(Ξ) -- src/trace.jl:203-225 (FactTrace: per coordinate the running (t, ∫x/(2t))) and :248-266 (PDMPTrace: the
    running vector y/(2t) after every event).  Returns a list of (t, y) array pairs per coordinate, resp. an [n x d] array."""
    if isinstance(tr, PDMPTrace):
        d = len(tr.x0)
        X = np.vstack([tr.x0[None], np.asarray(tr.x).reshape(-1, d)])
        te = np.concatenate([[tr.t0], tr.t])
        y = np.cumsum((X[:-1] + X[1:]) * np.diff(te)[:, None], axis=0)
        return y / (2.0 * te[1:, None])
    ev = tr.events
    x = tr.x0.copy()
    y = np.zeros_like(x)
    t = np.full(x.shape, tr.t0)
    ts = [[tr.t0] for _ in x]
    ys = [[xi] for xi in x]
    for t2, i, xi, _ in ev:
        y[i] += (x[i] + xi) * (t2 - t[i])
        t[i] = t2
        x[i] = xi
        ts[i].append(t2)
        ys[i].append(y[i] / (2 * t2))
    return [(np.array(a), np.array(b)) for a, b in zip(ts, ys)]


def mean(tr):
    """mean(Ξ): time average of the piecewise-linear path per coordinate -- src/trace.jl:182-200 (FactTrace), :229-246 (PDMPTrace)."""
    if isinstance(tr, PDMPTrace):
        return _mean_pdmp(tr)
    ev = tr.events
    x = tr.x0.copy()
    y = np.zeros_like(x)
    T = ev["t"][-1]
    t = np.full(x.shape, tr.t0)
    scale = 1 / (2 * T)
    for t2, i, xi, _ in ev:
        y[i] += (x[i] + xi) * (t2 - t[i]) * scale
        t[i] = t2
        x[i] = xi
    return y


def moments(tr: FactTrace, T_end=None):
    """Exact time averages of x_i and x_i² over [t0, T_end] (segments of coordinate i between ITS events);
    the tail after a coordinate's last event is extrapolated with its last velocity.  Used by the tests."""
    ev = tr.events
    d = tr.x0.size
    if T_end is None:
        T_end = ev["t"][-1]
    s1 = np.zeros(d)
    s2 = np.zeros(d)
    x = tr.x0.copy()
    th = tr.θ0.copy()
    t = np.full(d, tr.t0)
    for t2, i, xi, thi in ev:
        if t2 > T_end:
            continue
        dt = t2 - t[i]
        xa, xb = x[i], x[i] + th[i] * dt
        s1[i] += dt * (xa + xb) / 2
        s2[i] += dt * (xa * xa + xa * xb + xb * xb) / 3
        t[i], x[i], th[i] = t2, xi, thi
    dt = T_end - t
    xb = x + th * dt
    s1 += dt * (x + xb) / 2
    s2 += dt * (x * x + x * xb + xb * xb) / 3
    L = T_end - tr.t0
    m = s1 / L
    return m, s2 / L - m * m


def subtrace(tr: FactTrace, J):
    """subtrace(Ξ, J): trace of the subvector x[J] -- src/trace.jl:275-290."""
    J = np.asarray(J)
    assert np.all(np.diff(J) > 0)
    ev = tr.events
    loc = np.searchsorted(J, ev["i"])
    loc_c = np.minimum(loc, len(J) - 1)
    keep = J[loc_c] == ev["i"]
    sub = ev[keep].copy()
    sub["i"] = loc_c[keep]
    return FactTrace(tr.F, tr.t0, tr.x0[J].copy(), tr.θ0[J].copy(), sub)


def inclusion_prob(tr: FactTrace):
    """inclusion_prob(Ξ): fraction of time each coordinate is non-zero -- src/trace.jl:161-178."""
    ev = tr.events
    x = tr.x0.copy()
    y = np.zeros_like(x)
    T = ev["t"][-1]
    t = np.full(x.shape, tr.t0)
    for t2, i, xi, _ in ev:
        y[i] += ((x[i] != 0) | (xi != 0)) * (t2 - t[i]) / T
        t[i] = t2
        x[i] = xi
    return y
