// pdmp_consume.hip -- what callers do next with the chains, on the device: path integrals at probe coordinates (ESS estimators on the
// host see N x B x 32 numbers instead of N x d records) and, below, the streaming trace consumers (discretize / mean of src/trace.jl).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pdmp_engine.hpp"

namespace pdmp {

// J_i(T) = ∫_{t0}^{T} x_i(s) ds of every chain at `nprobe` coordinates (the integrand of mean(trace), src/trace.jl:182-200): the record
// carries the integral up to the coordinate's own clock and the linear piece from there (first sector of ZzRec and TrRec alike).
__global__ __launch_bounds__(256) void zz_path_integrals_kernel(const ZzRec* rec0, int64_t rec_stride, int64_t d, int64_t nchains,
                                                                const int64_t* __restrict__ probes, int64_t nprobe, double T,
                                                                double* __restrict__ out) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= nchains * nprobe) return;
    const int64_t ch = k / nprobe, p = k - ch * nprobe;
    const int64_t i = probes[p];
    const ZzRec* r = reinterpret_cast<const ZzRec*>(reinterpret_cast<const char*>(rec0) + (ch * d + i) * rec_stride);
    const double dt = T - r->t;
    out[k] = r->I + dt * (r->x + r->th * (dt * 0.5));
}

int launch_zz_path_integrals(const ZzRec* rec, int64_t rec_stride, int64_t d, int64_t nchains, const int64_t* probes, int64_t nprobe,
                             double T, double* out, void* stream) {
    const int64_t n = nchains * nprobe;
    hipLaunchKernelGGL(zz_path_integrals_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rec, rec_stride, d,
                       nchains, probes, nprobe, T, out);
    return (int)hipGetLastError();
}


// ------------------------------------------------------------------------------------------ streaming trace consumers
//
// mean(Ξ) (src/trace.jl:182-200) and collect(discretize(Ξ, dt)) (:94-125) of FactTrace traces, computed ON THE DEVICE from the engine's
// trace buffer, slice by slice: a C3-sized trace set (8 MB per chain x 4096) never crosses PCIe and need not even exist at once -- the
// buffer is recycled after every slice, the consumer keeps a CURSOR per (chain, coordinate): the time, position and velocity after the
// coordinate's last consumed event, and the running Σ (x_prev + x_k)(t_k − t_prev) of the reference's trapezoid rule (:191-195).
// A coordinate's path depends on its own events only, so the events of a slice are applied 256 at a time by a workgroup per chain; two
// events of one coordinate inside a chunk (rare) keep their order: an event waits for the latest earlier event of its coordinate.
// Grid positions are the closed form x_c + θ_c (g − t_c) from the cursor, g = t0 + k dt -- the arithmetic of trace.py (bitwise equal to
// it; the reference itself steps all coordinates through every event and agrees to rounding).  A grid ROW is written by the chain's
// workgroup for all d coordinates at once (coalesced: cursors read and positions written thread by thread), after the events with t <= g and
// as soon as a later event shows that the run has passed g (:111: a point is emitted while it lies before the last event) -- round 5; before,
// every event scattered the points of its own coordinate, 2.5 eight-byte stores per event at C3's rates, which cost more than the events.
// Sorted traces only (ZigZag without refresh clock; the sticky sampler's traces included).
struct ConsumeCursor {
    double t, x, th, y;  // clock, position, velocity after the coordinate's last consumed event; Σ (x_prev + x_k)(t_k − t_prev)
};
static_assert(sizeof(ConsumeCursor) == 32, "a cursor never straddles a 128-byte line");
// (sticky ensembles keep one more sum per cursor, in an array of its own behind the cursors: z = Σ (x_prev ≠ 0 | x_k ≠ 0)(t_k − t_prev), the time the
// coordinate was not stuck at 0 -- inclusion_prob, src/trace.jl:161-178; without freezes that is the time to the coordinate's last event)
struct ConsumeMeta {
    uint64_t consumed;  // events of this chain consumed so far (global event index)
    double t_last;      // time of the last of them (t0 before the first)
    uint64_t row_next;  // the first grid row not written yet by the streaming consumer
    uint64_t pad;
};

__global__ __launch_bounds__(256) void consume_init_kernel(const ZzRec* rec0, int64_t rec_stride, int64_t d, int64_t nchains, double t0,
                                                           ConsumeCursor* cur, double* zs, ConsumeMeta* meta, double* grid, int64_t K) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < nchains) {
        ConsumeMeta m;
        m.consumed = 0;
        m.t_last = t0;
        m.row_next = 0;
        m.pad = 0;
        meta[k] = m;
    }
    if (k >= nchains * d) return;
    const ZzRec* r = reinterpret_cast<const ZzRec*>(reinterpret_cast<const char*>(rec0) + k * rec_stride);
    ConsumeCursor c;
    c.t = t0;
    c.x = r->x;  // (before any run: the records hold x0, θ0 at t0)
    c.th = r->th;
    c.y = 0.0;
    cur[k] = c;
    if (zs) zs[k] = 0.0;
    // the first element of collect(discretize(Ξ, dt)) is t0 => x0 whatever follows (src/trace.jl:106-110): also for a chain without an event
    if (grid && K > 0) grid[(k / d) * K * d + (k % d)] = r->x;
}

// grid point k of coordinate i: the closed form from the cursor (events with t <= g applied: src/trace.jl:111-121)
__device__ __forceinline__ void consume_emit(double* grid_chain, int64_t d, int64_t K, double t0, double dt, uint32_t i, const ConsumeCursor& c,
                                             double t_until, bool closed_end) {
    if (!grid_chain || K <= 0) return;
    // candidates: the grid times inside [c.t, t_until) (or [c.t, t_until] at the very end of the run)
    double kf = floor((c.t - t0) / dt) - 1.0;
    int64_t k = (kf > 0.0) ? (int64_t)kf : 0;
    while (k < K && t0 + dt * (double)k < c.t) ++k;
    for (; k < K; ++k) {
        const double g = t0 + dt * (double)k;
        if (closed_end ? !(g <= t_until) : !(g < t_until)) break;
        grid_chain[k * d + i] = c.x + c.th * (g - c.t);
    }
}

// (snap: the (ntrace, nevents) pairs a finished launch left, taken by consume_snapshot_kernel -- the asynchronous consumer reads a buffer the
// event loop no longer writes while the headers already count the next slice; nullptr: the headers themselves)
constexpr int CONSUME_HASH = 4096;  // slots of the per-chunk table that finds two events of one coordinate
__global__ __launch_bounds__(256) void consume_events_kernel(const pdmp_event* ev0, int64_t cap, const DevChain* hdr, const uint64_t* __restrict__ snap,
                                                             int64_t d, ConsumeCursor* cur0, double* zs0, ConsumeMeta* meta, double* grid0, int64_t K,
                                                             double t0, double dt, double2* cm0) {
    const int64_t chain = blockIdx.x;
    const int tid = threadIdx.x;
    __shared__ uint32_t s_i[256];
    __shared__ int s_done[256];
    __shared__ uint8_t s_own[CONSUME_HASH];
    __shared__ uint64_t s_split;
    const uint64_t ntrace = snap ? snap[2 * chain] : hdr[chain].c.ntrace, nevents = snap ? snap[2 * chain + 1] : hdr[chain].c.nevents;
    ConsumeMeta m = meta[chain];
    const uint64_t first_global = nevents - ntrace;  // global index of buffer slot 0
    uint64_t pos = (m.consumed > first_global) ? (m.consumed - first_global) : 0;  // first unconsumed slot
    if (pos >= ntrace) return;
    const pdmp_event* ev = ev0 + chain * cap;
    ConsumeCursor* cur = cur0 + chain * d;
    double* zs = zs0 ? zs0 + chain * d : nullptr;
    double* grid = grid0 ? grid0 + chain * K * d : nullptr;
    uint64_t row = m.row_next;
    for (;;) {
        // the events up to the next grid time g (t <= g: an event AT g belongs before the row), then the row -- if the slice goes on beyond g
        const bool want_row = grid && (int64_t)row < K;
        const double g = t0 + dt * (double)row;
        uint64_t split = ntrace;
        if (want_row) {
            if (tid == 0) {  // first slot in [pos, ntrace) whose time exceeds g (the trace is sorted by time)
                uint64_t lo = pos, hi = ntrace;
                while (lo < hi) {
                    const uint64_t mid = lo + ((hi - lo) >> 1);
                    if (ev[mid].t <= g) lo = mid + 1;
                    else hi = mid;
                }
                s_split = lo;
            }
            __syncthreads();
            split = s_split;
        }
        for (uint64_t base = pos; base < split; base += 256) {
            const uint64_t e = base + (uint64_t)tid;
            const bool valid = e < split;
            pdmp_event evt;
            evt.t = 0.0;
            evt.i = 0;
            evt.x = evt.theta = 0.0;
            if (valid) evt = ev[e];
            // two events of one coordinate in this chunk?  every event writes its thread into a slot of a hash table; an event that does not find
            // itself there after everybody wrote shares the slot with another one (the same coordinate, or -- ~8 times per chunk -- another that
            // hashes alike): those few sort themselves out by the ordered rounds below, everybody else is alone on its coordinate
            const uint32_t hs = ((uint32_t)evt.i * 0x9E3779B1u) >> 20;  // (12 bits)
            s_i[tid] = valid ? (uint32_t)evt.i : 0xffffffffu;
            s_done[tid] = valid ? 0 : 1;
            if (valid) s_own[hs] = (uint8_t)tid;
            __syncthreads();
            const bool clash = valid && s_own[hs] != (uint8_t)tid;
            __syncthreads();
            if (clash) s_own[hs] = 0xff;  // (marks the slot for the event that owned it: it is part of the clash too)
            __syncthreads();
            const bool contested = valid && (clash || s_own[hs] == 0xff);
            int dep = -1;  // the latest earlier event of the same coordinate inside this chunk
            if (contested)
                for (int q = tid - 1; q >= 0; --q)
                    if (s_i[q] == (uint32_t)evt.i) {
                        dep = q;
                        break;
                    }
            bool mine_done = !valid;
            for (int round = 0; round < 256; ++round) {
                const bool ready = !mine_done && (dep < 0 || s_done[dep] != 0);
                __syncthreads();  // (everybody has read the flags of this round)
                if (ready) {
                    const uint32_t i = (uint32_t)evt.i;
                    ConsumeCursor c = cur[i];
                    c.y += (c.x + evt.x) * (evt.t - c.t);  // src/trace.jl:193 without the common factor 1/(2T)
                    if (zs && (c.x != 0.0 || evt.x != 0.0)) zs[i] += evt.t - c.t;  // :172 without the common factor 1/T (−0.0 of a freeze counts as 0)
                    c.t = evt.t;
                    c.x = evt.x;
                    c.th = evt.theta;
                    cur[i] = c;
                    // cummean(Ξ), src/trace.jl:203-226: after each event of coordinate i the pair (t[i], y[i] / (2 t[i])) -- the same sum, the same
                    // order of operations per coordinate; written beside the event's slot (round 6)
                    if (cm0) cm0[chain * cap + (int64_t)e] = make_double2(c.t, c.y / (2.0 * c.t));
                    __threadfence_block();
                    s_done[tid] = 1;
                    mine_done = true;
                }
                const int left = __syncthreads_count(mine_done ? 0 : 1);
                if (left == 0) break;
            }
            __syncthreads();
        }
        pos = split;
        if (!want_row || split >= ntrace) break;  // (no later event in this slice: whether the run passes g is not known yet)
        __threadfence_block();
        __syncthreads();
        double* const grow = grid + row * d;
        for (int64_t i = tid; i < d; i += 256) {
            const ConsumeCursor c = cur[i];
            grow[i] = c.x + c.th * (g - c.t);
        }
        row += 1;
    }
    if (tid == 0) {
        m.consumed = nevents;
        m.t_last = ev[ntrace - 1].t;
        m.row_next = row;
        meta[chain] = m;
    }
}

// the grid points after a coordinate's last event, up to the chain's last event time (a point is emitted while it lies before it, :111)
__global__ __launch_bounds__(256) void consume_flush_kernel(int64_t d, const ConsumeCursor* cur0, const ConsumeMeta* meta, double* grid0, int64_t K,
                                                            double t0, double dt) {
    const int64_t chain = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= d || !grid0) return;
    consume_emit(grid0 + chain * K * d, d, K, t0, dt, (uint32_t)i, cur0[chain * d + i], meta[chain].t_last, false);
}

__global__ __launch_bounds__(256) void consume_mean_kernel(int64_t d, int64_t chain_first, int64_t n, const ConsumeCursor* cur0, const ConsumeMeta* meta,
                                                           double* mean_out, double* T_out) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n * d) return;
    const int64_t q = k / d, i = k - q * d;
    const int64_t chain = chain_first + q;
    const double T = meta[chain].t_last;
    mean_out[k] = cur0[chain * d + i].y * (1 / (2 * T));  // y[i] summed over i's events, scaled once (src/trace.jl:190 scales every term)
    if (i == 0 && T_out) T_out[q] = T;
}

__global__ __launch_bounds__(256) void consume_inclusion_kernel(int64_t d, int64_t chain_first, int64_t n, const ConsumeCursor* cur0, const double* zs0,
                                                                const ConsumeMeta* meta, double t0, double* out, double* T_out) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n * d) return;
    const int64_t q = k / d, i = k - q * d;
    const int64_t chain = chain_first + q;
    const double T = meta[chain].t_last;
    // (src/trace.jl:172 divides every term; without freezes every term counts: the sum telescopes to the time of the coordinate's last event)
    out[k] = (zs0 ? zs0[chain * d + i] : cur0[chain * d + i].t - t0) / T;
    if (i == 0 && T_out) T_out[q] = T;
}

// the cursors, and behind them the z sums of a sticky ensemble
size_t consume_cursor_bytes(bool with_z) { return sizeof(ConsumeCursor) + (with_z ? sizeof(double) : 0); }
size_t consume_meta_bytes() { return sizeof(ConsumeMeta); }
static double* z_of(void* cur, int64_t d, int64_t nchains, bool with_z) {
    return with_z ? reinterpret_cast<double*>(static_cast<ConsumeCursor*>(cur) + nchains * d) : nullptr;
}

int launch_consume_inclusion(int64_t d, int64_t nchains, bool with_z, double t0, int64_t chain_first, int64_t n, void* cur, const void* meta, double* out,
                             double* T_out, void* stream) {
    const int64_t tot = n * d;
    hipLaunchKernelGGL(consume_inclusion_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d, chain_first, n,
                       static_cast<const ConsumeCursor*>(cur), z_of(cur, d, nchains, with_z), static_cast<const ConsumeMeta*>(meta), t0, out, T_out);
    return (int)hipGetLastError();
}

int launch_consume_init(const ZzRec* rec, int64_t rec_stride, int64_t d, int64_t nchains, double t0, void* cur, bool with_z, void* meta, double* grid,
                        int64_t K, void* stream) {
    const int64_t n = nchains * d;
    hipLaunchKernelGGL(consume_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rec, rec_stride, d, nchains, t0,
                       static_cast<ConsumeCursor*>(cur), z_of(cur, d, nchains, with_z), static_cast<ConsumeMeta*>(meta), grid, K);
    return (int)hipGetLastError();
}
int launch_consume_events(const pdmp_event* ev, int64_t cap, const DevChain* hdr, const uint64_t* snap, int64_t d, int64_t nchains, void* cur,
                          bool with_z, void* meta, double* grid, int64_t K, double t0, double dt, void* stream, double* cummean_pairs) {
    hipLaunchKernelGGL(consume_events_kernel, dim3((unsigned)nchains), dim3(256), 0, (hipStream_t)stream, ev, cap, hdr, snap, d,
                       static_cast<ConsumeCursor*>(cur), z_of(cur, d, nchains, with_z), static_cast<ConsumeMeta*>(meta), grid, K, t0, dt,
                       reinterpret_cast<double2*>(cummean_pairs));
    return (int)hipGetLastError();
}

// subtrace(tr, J), src/trace.jl:275-290, on a chain's trace segment: the events whose coordinate lies in J, renumbered by their position in J (loc[i] =
// position of i in J, or -1), in their order; one workgroup, chunks of 256 events compacted by a block-wide prefix count
__global__ __launch_bounds__(256) void trace_subtrace_kernel(const pdmp_event* __restrict__ ev, int64_t n, const int32_t* __restrict__ loc, pdmp_event* __restrict__ out,
                                                            int64_t out_cap, unsigned long long* n_out) {
    __shared__ uint32_t s_cnt[256];
    __shared__ unsigned long long s_base;
    const int tid = threadIdx.x;
    if (tid == 0) s_base = 0ull;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 256) {
        const int64_t k = base + tid;
        pdmp_event e;
        e.t = 0.0;
        e.i = 0;
        e.x = e.theta = 0.0;
        int32_t l = -1;
        if (k < n) {
            e = ev[k];
            l = loc[e.i];
        }
        s_cnt[tid] = (l >= 0) ? 1u : 0u;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {  // inclusive prefix sum
            const uint32_t v = (tid >= off) ? s_cnt[tid - off] : 0u;
            __syncthreads();
            s_cnt[tid] += v;
            __syncthreads();
        }
        const unsigned long long at = s_base + (unsigned long long)s_cnt[tid] - 1ull;
        if (l >= 0 && (int64_t)at < out_cap) {
            e.i = (int64_t)l;
            out[at] = e;
        }
        __syncthreads();
        if (tid == 255) s_base += (unsigned long long)s_cnt[255];
        __syncthreads();
    }
    if (tid == 0) *n_out = s_base;
}
int launch_trace_subtrace(const pdmp_event* ev, int64_t n, const int32_t* loc, pdmp_event* out, int64_t out_cap, unsigned long long* n_out, void* stream) {
    hipLaunchKernelGGL(trace_subtrace_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ev, n, loc, out, out_cap, n_out);
    return (int)hipGetLastError();
}
// what a finished launch left in its trace segments, per chain, and the segments handed back empty (the event loop's next launch writes the OTHER
// buffer from slot 0): one thread per chain, stream-ordered behind the launch
__global__ __launch_bounds__(256) void consume_snapshot_kernel(DevChain* hdr, int64_t nchains, uint64_t* __restrict__ snap) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= nchains) return;
    snap[2 * k] = hdr[k].c.ntrace;
    snap[2 * k + 1] = hdr[k].c.nevents;
    hdr[k].c.ntrace = 0;
}
int launch_consume_snapshot(DevChain* hdr, int64_t nchains, uint64_t* snap, void* stream) {
    hipLaunchKernelGGL(consume_snapshot_kernel, dim3((unsigned)((nchains + 255) / 256)), dim3(256), 0, (hipStream_t)stream, hdr, nchains, snap);
    return (int)hipGetLastError();
}
int launch_consume_flush(int64_t d, int64_t nchains, const void* cur, const void* meta, double* grid, int64_t K, double t0, double dt, void* stream) {
    hipLaunchKernelGGL(consume_flush_kernel, dim3((unsigned)((d + 255) / 256), (unsigned)nchains), dim3(256), 0, (hipStream_t)stream, d,
                       static_cast<const ConsumeCursor*>(cur), static_cast<const ConsumeMeta*>(meta), grid, K, t0, dt);
    return (int)hipGetLastError();
}
int launch_consume_mean(int64_t d, int64_t chain_first, int64_t n, const void* cur, const void* meta, double* mean_out, double* T_out, void* stream) {
    const int64_t tot = n * d;
    hipLaunchKernelGGL(consume_mean_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d, chain_first, n,
                       static_cast<const ConsumeCursor*>(cur), static_cast<const ConsumeMeta*>(meta), mean_out, T_out);
    return (int)hipGetLastError();
}

}  // namespace pdmp
