/*
 * trace_oracle.c -- CPU ORACLE of what callers do next with a FactTrace: test infrastructure only, like pdmp_oracle.c.
 *
 * The reference's consumers are event-by-event loops over Ξ.events = [(t, i, x_i, θ_i), ...] (src/trace.jl); they are restated here operation by
 * operation in the order Julia evaluates them (compiled with -ffp-contract=off), for the ZigZag's piecewise-linear flow (move_forward!(τ, t, x, θ,
 * ::ZigZag): x .+= θ .* τ, src/dynamics.jl:11-15).  The host-side vectorised consumers (zigzagboomerang.jl_amd/trace.py) and the device consumers
 * (csrc/pdmp_consume.hip) are both held to THESE loops by tests/test_trace_consumers.py and tests/test_gpu_consumers.py -- one hop from src/trace.jl.
 *
 * Events arrive as four parallel arrays (t, i (0-based), x, θ) of length n; x0, θ0 have length d.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Statistics.mean(trace), src/trace.jl:182-200: y[i] += (x[i] + xi) * (t2 - t[i]) * scale with scale = 1 / (2 T), T = the LAST event's time
 * (coordinates are integrated up to their own last event only: the reference's behaviour) */
void orc_trace_mean(int64_t d, double t0, const double* x0, int64_t n, const double* et, const int64_t* ei, const double* ex, double* y) {
    double* x = (double*)malloc((size_t)d * sizeof(double));
    double* t = (double*)malloc((size_t)d * sizeof(double));
    memcpy(x, x0, (size_t)d * sizeof(double));
    for (int64_t j = 0; j < d; ++j) {
        t[j] = t0;
        y[j] = 0.0;
    }
    if (n > 0) {
        const double T = et[n - 1];
        const double scale = 1.0 / (2.0 * T);
        for (int64_t k = 0; k < n; ++k) {
            const double t2 = et[k];
            const int64_t i = ei[k];
            const double xi = ex[k];
            y[i] += (x[i] + xi) * (t2 - t[i]) * scale;
            t[i] = t2;
            x[i] = xi;
        }
    }
    free(x);
    free(t);
}

/* inclusion_prob(trace), src/trace.jl:161-178: y[i] += (x[i] ≠ 0 | xi ≠ 0) * (t2 - t[i]) / T -- read as the script means it (the segment counts
 * unless the coordinate sits at zero at both of its ends; Julia's own precedence would apply `|` to the floats first and throw) */
void orc_trace_inclusion_prob(int64_t d, double t0, const double* x0, int64_t n, const double* et, const int64_t* ei, const double* ex, double* y) {
    double* x = (double*)malloc((size_t)d * sizeof(double));
    double* t = (double*)malloc((size_t)d * sizeof(double));
    memcpy(x, x0, (size_t)d * sizeof(double));
    for (int64_t j = 0; j < d; ++j) {
        t[j] = t0;
        y[j] = 0.0;
    }
    if (n > 0) {
        const double T = et[n - 1];
        for (int64_t k = 0; k < n; ++k) {
            const double t2 = et[k];
            const int64_t i = ei[k];
            const double xi = ex[k];
            const double ind = (x[i] != 0.0 || xi != 0.0) ? 1.0 : 0.0;
            y[i] += ind * (t2 - t[i]) / T;
            t[i] = t2;
            x[i] = xi;
        }
    }
    free(x);
    free(t);
}

/* cummean(trace::FactTrace), src/trace.jl:203-226: per coordinate the list (t, y / (2 t)) after each of ITS events, preceded by (t0, x0[i]); written as
 * one entry per event in event order (out_t[k], out_y[k] belong to coordinate ei[k]): y[i] += (x[i] + xi) * (t2 - t[i]); push y[i] / (2 t[i]) */
void orc_trace_cummean(int64_t d, double t0, const double* x0, int64_t n, const double* et, const int64_t* ei, const double* ex, double* out_t,
                       double* out_y) {
    double* x = (double*)malloc((size_t)d * sizeof(double));
    double* t = (double*)malloc((size_t)d * sizeof(double));
    double* y = (double*)calloc((size_t)d, sizeof(double));
    memcpy(x, x0, (size_t)d * sizeof(double));
    for (int64_t j = 0; j < d; ++j) t[j] = t0;
    for (int64_t k = 0; k < n; ++k) {
        const double t2 = et[k];
        const int64_t i = ei[k];
        const double xi = ex[k];
        y[i] += (x[i] + xi) * (t2 - t[i]);
        t[i] = t2;
        x[i] = xi;
        out_t[k] = t[i];
        out_y[k] = y[i] / (2.0 * t[i]);
    }
    free(x);
    free(t);
    free(y);
}

/* collect(discretize(trace, dt)) for FactTrace{ZigZag}, src/trace.jl:106-125 with the first element of :100-104 (t0 => x0): returns the number of grid
 * points written (at most cap); ts[q], xs[q * d ..] -- the iteration ends when the events run out (the point after the last event is never produced) */
int64_t orc_trace_discretize(int64_t d, double t0, const double* x0, const double* th0, int64_t n, const double* et, const int64_t* ei, const double* ex,
                             const double* eth, double dt_grid, int64_t cap, double* ts, double* xs) {
    double* x = (double*)malloc((size_t)d * sizeof(double));
    double* th = (double*)malloc((size_t)d * sizeof(double));
    memcpy(x, x0, (size_t)d * sizeof(double));
    memcpy(th, th0, (size_t)d * sizeof(double));
    double t = t0;
    int64_t k = 0, q = 0;
    if (q < cap) {
        ts[q] = t;
        memcpy(xs + (size_t)q * (size_t)d, x, (size_t)d * sizeof(double));
    }
    q += 1;
    for (;;) {
        double dt = dt_grid;
        int produced = 0;
        for (;;) {
            if (k >= n) break; /* k > length(FT.events) && return nothing */
            const double ti = et[k];
            if (t + dt < ti) {
                for (int64_t j = 0; j < d; ++j) x[j] = x[j] + th[j] * dt; /* move_forward!(dt, t, x, θ, F) */
                t = t + dt;
                produced = 1;
                break;
            } else { /* move not more than to ti to change direction */
                const double del = ti - t;
                dt = dt - del;
                for (int64_t j = 0; j < d; ++j) x[j] = x[j] + th[j] * del;
                t = ti;
                x[ei[k]] = ex[k];
                th[ei[k]] = eth[k];
                k = k + 1;
            }
        }
        if (!produced) break;
        if (q < cap) {
            ts[q] = t;
            memcpy(xs + (size_t)q * (size_t)d, x, (size_t)d * sizeof(double));
        }
        q += 1;
    }
    free(x);
    free(th);
    return q;
}

/* subtrace(tr, J), src/trace.jl:275-290: the events whose coordinate lies in the sorted index set J (0-based here), renumbered by position in J;
 * returns their number, out_k[m] = index of the m-th kept event in the original list, out_i[m] = its new coordinate */
int64_t orc_trace_subtrace(int64_t nJ, const int64_t* J, int64_t n, const int64_t* ei, int64_t* out_k, int64_t* out_i) {
    int64_t m = 0;
    for (int64_t k = 0; k < n; ++k) {
        /* r = searchsorted(J, ev[2]); isempty(r) && continue */
        int64_t lo = 0, hi = nJ;
        while (lo < hi) {
            const int64_t mid = (lo + hi) / 2;
            if (J[mid] < ei[k]) lo = mid + 1;
            else hi = mid;
        }
        if (lo == nJ || J[lo] != ei[k]) continue;
        out_k[m] = k;
        out_i[m] = lo;
        m += 1;
    }
    return m;
}
