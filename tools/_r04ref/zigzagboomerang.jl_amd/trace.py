"""Trace consumers, mirroring src/trace.jl for FactTrace (what every caller of spdmp does next with Ξ).

Events are structured arrays (t, i, x, theta) with 0-based i.  Host side only: these are the callers' post-processing
(SURVEY.md 8f1), not part of the device hot path.  Everything is vectorised over EVENTS (no interpreter loop per event): a
coordinate's path depends on its own events only -- between two of them it flows freely (linearly, or by the FactBoomerang's
rotation about μ_i) -- so the events are grouped per coordinate once (stable sort) and every consumer is a handful of array
operations; the only Python loops left run over coordinates (collect / discretize), never over events.  The reference moves
all coordinates step by step (x += θ·Δt per event); the closed forms used here agree with it to rounding.
"""
import numpy as np

from .flows import Boomerang, FactBoomerang, FactTrace, PDMPTrace


def _is_boom(tr):
    return isinstance(tr.F, FactBoomerang)


def _free_flow(tr, j, x, th, tau):
    """State of coordinate(s) j a time tau after (x, θ): linear (src/dynamics.jl:11-15) or the rotation about μ_j (:29-36)."""
    if _is_boom(tr):
        mu = tr.F.μ[j]
        s, c = np.sin(tau), np.cos(tau)
        return (x - mu) * c + th * s + mu, -(x - mu) * s + th * c
    return x + th * tau, th + 0.0 * tau


def _prev_same_coordinate(tr):
    """For every event k (in time order): time, position and velocity the coordinate had after ITS previous event (the initial
    state for its first one)."""
    ev = tr.events
    n = len(ev)
    i = ev["i"].astype(np.int64)
    order = np.argsort(i, kind="stable")  # by coordinate, time order kept inside a coordinate
    si = i[order]
    prev_sorted = np.empty(n, dtype=np.int64)
    if n:
        prev_sorted[0] = -1
        prev_sorted[1:] = np.where(si[1:] == si[:-1], order[:-1], -1)
    prev = np.empty(n, dtype=np.int64)
    prev[order] = prev_sorted
    has = prev >= 0
    pc = np.maximum(prev, 0)
    tp = np.where(has, ev["t"][pc], tr.t0)
    xp = np.where(has, ev["x"][pc], tr.x0[i])
    thp = np.where(has, ev["theta"][pc], tr.θ0[i])
    return i, tp, xp, thp


def _groups(tr):
    """Events grouped per coordinate: (order, ptr) with order[ptr[j]:ptr[j+1]] the event indices of coordinate j in time order."""
    i = tr.events["i"].astype(np.int64)
    order = np.argsort(i, kind="stable")
    ptr = np.zeros(tr.x0.size + 1, dtype=np.int64)
    np.cumsum(np.bincount(i, minlength=tr.x0.size), out=ptr[1:])
    return order, ptr


def _states_at(tr, times, upto=None):
    """x_j(times[r]) for every coordinate j: [len(times) x d].  upto[r] = number of leading events (trace order) applied at row r;
    None: every event with t <= times[r]."""
    ev = tr.events
    d = tr.x0.size
    order, ptr = _groups(tr)
    out = np.empty((len(times), d))
    for j in range(d):
        own = order[ptr[j]:ptr[j + 1]]
        if upto is None:
            pos = np.searchsorted(ev["t"][own], times, side="right") - 1
        else:
            pos = np.searchsorted(own, upto - 1, side="right") - 1
        has = pos >= 0
        k = own[np.maximum(pos, 0)] if len(own) else np.zeros(len(times), dtype=np.int64)
        if len(own):
            tl = np.where(has, ev["t"][k], tr.t0)
            xl = np.where(has, ev["x"][k], tr.x0[j])
            thl = np.where(has, ev["theta"][k], tr.θ0[j])
        else:
            tl, xl, thl = tr.t0, tr.x0[j], tr.θ0[j]
        out[:, j] = _free_flow(tr, j, xl, thl, times - tl)[0]
    return out


def collect(tr: FactTrace):
    """collect(Ξ): (t, x) pairs of Base.iterate(FT::FactTrace), src/trace.jl:44-63 -- the initial state, the state after each
    event but the last (which is never applied, :56), and that state once more: 1 + length(events) entries (:42)."""
    ev = tr.events
    n = len(ev)
    if n == 0:
        return np.array([tr.t0]), tr.x0[None].copy()
    ts = np.concatenate([[tr.t0], ev["t"][:n - 1], [ev["t"][n - 2] if n > 1 else tr.t0]])
    upto = np.concatenate([[0], np.arange(1, n), [n - 1]])
    return ts, _states_at(tr, ts, upto=upto)


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


def discretize_1d(events, flow, dt):
    """discretize(x::Vector, Flow::Union{ZigZag1d, Boomerang1d}, dt) -- src/discretise.jl:10-42, the skeleton of the 1-d samplers to a
    trajectory: the clock advances by dt inside a segment (the step that would cross the next event is shortened and the remainder carried
    into the following segment, :28-29), the state flows by move_forward from the previous grid point (src/dynamics.jl:66-68,79-82), the last
    segment is not emitted (:18) and the final clock / position is appended (:40).  events: structured (t, x, theta).  Returns (t, x)."""
    from .flows import Boomerang1d
    boom = isinstance(flow, Boomerang1d)
    mu = flow.μ if boom else 0.0
    n = len(events)
    ts, xs = [0.0], [float(events["x"][0])]
    clock, dt_cur = 0.0, float(dt)
    xi, th = float(events["x"][0]), float(events["theta"][0])

    def move(tau, clock, xi, th):
        if boom:
            s, c = np.sin(tau), np.cos(tau)
            return clock + tau, (xi - mu) * c + th * s + mu, -(xi - mu) * s + th * c
        return tau + clock, xi + th * tau, th

    k = 0
    while k < n - 2:
        tau_next = float(events["t"][k + 1])
        while clock + dt_cur <= tau_next:
            if th == 0.0:
                clock += dt_cur
            else:
                clock, xi, th = move(dt_cur, clock, xi, th)
            ts.append(clock)
            xs.append(xi)
            dt_cur = float(dt)
        dt_cur = dt_cur - (tau_next - clock)
        if th == 0.0:
            clock = tau_next
        else:
            clock, xi, th = move(tau_next - clock, clock, xi, th)
        k += 1
        xi, th = float(events["x"][k]), float(events["theta"][k])
    ts.append(clock)
    xs.append(xi)
    return np.array(ts), np.array(xs)


def discretize(tr, dt):
    """collect(discretize(Ξ, dt)): positions on the grid t0, t0+dt, ... -- src/trace.jl:94-125 (FactTrace: a grid point is
    emitted while it lies before the last event; events at or before a grid time are applied), :129-150 (PDMPTrace)."""
    if isinstance(tr, PDMPTrace):
        return _discretize_pdmp(tr, dt)
    ev = tr.events
    if len(ev) == 0:
        return np.array([tr.t0]), tr.x0[None].copy()
    # The reference consumes the events IN TRACE ORDER and stops at the first one later than the grid time (:111-113).  A refresh
    # of a coordinate whose clock lags (src/sfact.jl:84-85: the refreshed i is not moved to t′) is recorded with its stale time, so a
    # trace with λref > 0 is not sorted; the number of events applied at grid time g is the first index whose time exceeds g,
    # i.e. a search in the running maximum of the event times.
    tmax = np.maximum.accumulate(ev["t"])
    n = int(np.ceil((tmax[-1] - tr.t0) / dt)) + 1
    grid = tr.t0 + dt * np.arange(n)
    grid = grid[grid < tmax[-1]]
    if len(grid) == 0:
        grid = np.array([tr.t0])
    return grid, _states_at(tr, grid, upto=np.searchsorted(tmax, grid, side="right"))


def _mean_pdmp(tr: PDMPTrace):
    """Statistics.mean(Ξ::PDMPTrace) -- src/trace.jl:229-246, restated as written: Σ (x + x₂)(t₂ − t) over consecutive events divided
    by the LAST event time T -- the reference omits the ½ of the trapezoid rule here (its cummean, :248-266, has it), so this is
    twice the time average of the interpolated path."""
    d = len(tr.x0)
    X = np.vstack([tr.x0[None], np.asarray(tr.x).reshape(-1, d)])
    te = np.concatenate([[tr.t0], tr.t])
    y = ((X[:-1] + X[1:]) * np.diff(te)[:, None]).sum(0)
    return y / tr.t[-1]


def _segment_integrals(tr):
    """Per event k of coordinate i: Δt since i's previous event and the reference's trapezoid term (x_prev + x_k)·Δt, where x_k is
    the RECORDED position (src/trace.jl:191-195; for a ZigZag the path between two events of i is the chord, so this is exact)."""
    i, tp, xp, _ = _prev_same_coordinate(tr)
    ev = tr.events
    dt = ev["t"] - tp
    return i, dt, (xp + ev["x"]) * dt, xp


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
    _, _, term, _ = _segment_integrals(tr)
    order, ptr = _groups(tr)
    out = []
    for j in range(tr.x0.size):
        own = order[ptr[j]:ptr[j + 1]]
        t = ev["t"][own]
        y = np.cumsum(term[own]) / (2 * t) if len(own) else np.empty(0)
        out.append((np.concatenate([[tr.t0], t]), np.concatenate([[tr.x0[j]], y])))
    return out


def mean(tr):
    """mean(Ξ): time average of the piecewise-linear path per coordinate -- src/trace.jl:182-200 (FactTrace), :229-246 (PDMPTrace)."""
    if isinstance(tr, PDMPTrace):
        return _mean_pdmp(tr)
    ev = tr.events
    i, _, term, _ = _segment_integrals(tr)
    T = ev["t"][-1]
    return np.bincount(i, weights=term * (1 / (2 * T)), minlength=tr.x0.size)  # (summed per coordinate in event order, like :191-196)


def moments(tr: FactTrace, T_end=None):
    """Exact time averages of x_i and x_i² over [t0, T_end] (segments of coordinate i between ITS events, linear flow);
    the tail after a coordinate's last event is extrapolated with its last velocity.  Used by the tests and the ESS validation."""
    ev = tr.events
    d = tr.x0.size
    if T_end is None:
        T_end = ev["t"][-1]
    keep = ev["t"] <= T_end
    sub = FactTrace(tr.F, tr.t0, tr.x0, tr.θ0, ev[keep])
    i, tp, xa, tha = _prev_same_coordinate(sub)
    e = sub.events
    dt = e["t"] - tp
    xb = xa + tha * dt
    s1 = np.bincount(i, weights=dt * (xa + xb) / 2, minlength=d)
    s2 = np.bincount(i, weights=dt * (xa * xa + xa * xb + xb * xb) / 3, minlength=d)
    # the open segment after each coordinate's last event
    tl, xl, thl = np.full(d, tr.t0), tr.x0.astype(np.float64).copy(), tr.θ0.astype(np.float64).copy()
    if len(e):
        order, ptr = _groups(sub)
        last = order[np.maximum(ptr[1:] - 1, 0)]
        has = ptr[1:] > ptr[:-1]
        tl = np.where(has, e["t"][last], tl)
        xl = np.where(has, e["x"][last], xl)
        thl = np.where(has, e["theta"][last], thl)
    dt = T_end - tl
    xb = xl + thl * dt
    s1 += dt * (xl + xb) / 2
    s2 += dt * (xl * xl + xl * xb + xb * xb) / 3
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
    i, dt, _, xp = _segment_integrals(tr)
    T = ev["t"][-1]
    return np.bincount(i, weights=((xp != 0) | (ev["x"] != 0)) * dt / T, minlength=tr.x0.size)
