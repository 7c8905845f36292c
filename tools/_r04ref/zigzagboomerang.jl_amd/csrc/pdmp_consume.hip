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
// it; the reference itself steps all coordinates through every event and agrees to rounding).  Sorted traces only (ZigZag without
// refresh clock; the sticky sampler's traces included).
struct ConsumeCursor {
    double t, x, th, y;  // clock, position, velocity after the coordinate's last consumed event; Σ (x_prev + x_k)(t_k − t_prev)
    double z;            // Σ (x_prev ≠ 0 | x_k ≠ 0)(t_k − t_prev): the time the coordinate was not stuck at 0 (inclusion_prob, src/trace.jl:161-178)
};
struct ConsumeMeta {
    uint64_t consumed;  // events of this chain consumed so far (global event index)
    double t_last;      // time of the last of them (t0 before the first)
    uint64_t pad[2];
};

__global__ __launch_bounds__(256) void consume_init_kernel(const ZzRec* rec0, int64_t rec_stride, int64_t d, int64_t nchains, double t0,
                                                           ConsumeCursor* cur, ConsumeMeta* meta, double* grid, int64_t K) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < nchains) {
        ConsumeMeta m;
        m.consumed = 0;
        m.t_last = t0;
        m.pad[0] = m.pad[1] = 0;
        meta[k] = m;
    }
    if (k >= nchains * d) return;
    const ZzRec* r = reinterpret_cast<const ZzRec*>(reinterpret_cast<const char*>(rec0) + k * rec_stride);
    ConsumeCursor c;
    c.t = t0;
    c.x = r->x;  // (before any run: the records hold x0, θ0 at t0)
    c.th = r->th;
    c.y = 0.0;
    c.z = 0.0;
    cur[k] = c;
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

__global__ __launch_bounds__(256) void consume_events_kernel(const pdmp_event* ev0, int64_t cap, const DevChain* hdr, int64_t d, ConsumeCursor* cur0,
                                                             ConsumeMeta* meta, double* grid0, int64_t K, double t0, double dt) {
    const int64_t chain = blockIdx.x;
    const int tid = threadIdx.x;
    __shared__ uint32_t s_i[256];
    __shared__ int s_done[256];
    const uint64_t ntrace = hdr[chain].c.ntrace, nevents = hdr[chain].c.nevents;
    ConsumeMeta m = meta[chain];
    const uint64_t first_global = nevents - ntrace;  // global index of buffer slot 0
    uint64_t begin = (m.consumed > first_global) ? (m.consumed - first_global) : 0;  // first unconsumed slot
    if (begin >= ntrace) return;
    const pdmp_event* ev = ev0 + chain * cap;
    ConsumeCursor* cur = cur0 + chain * d;
    double* grid = grid0 ? grid0 + chain * K * d : nullptr;
    for (uint64_t base = begin; base < ntrace; base += 256) {
        const uint64_t e = base + (uint64_t)tid;
        const bool valid = e < ntrace;
        pdmp_event evt;
        evt.t = 0.0;
        evt.i = 0;
        evt.x = evt.theta = 0.0;
        if (valid) evt = ev[e];
        s_i[tid] = valid ? (uint32_t)evt.i : 0xffffffffu;
        s_done[tid] = valid ? 0 : 1;
        __syncthreads();
        int dep = -1;  // the latest earlier event of the same coordinate inside this chunk
        if (valid)
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
                consume_emit(grid, d, K, t0, dt, i, c, evt.t, false);
                c.y += (c.x + evt.x) * (evt.t - c.t);  // src/trace.jl:193 without the common factor 1/(2T)
                if (c.x != 0.0 || evt.x != 0.0) c.z += evt.t - c.t;  // :172 without the common factor 1/T (−0.0 of a freeze counts as 0)
                c.t = evt.t;
                c.x = evt.x;
                c.th = evt.theta;
                cur[i] = c;
                __threadfence_block();
                s_done[tid] = 1;
                mine_done = true;
            }
            const int left = __syncthreads_count(mine_done ? 0 : 1);
            if (left == 0) break;
        }
        __syncthreads();
    }
    if (tid == 0) {
        m.consumed = nevents;
        m.t_last = ev[ntrace - 1].t;
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

__global__ __launch_bounds__(256) void consume_inclusion_kernel(int64_t d, int64_t chain_first, int64_t n, const ConsumeCursor* cur0,
                                                                const ConsumeMeta* meta, double* out, double* T_out) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n * d) return;
    const int64_t q = k / d, i = k - q * d;
    const int64_t chain = chain_first + q;
    const double T = meta[chain].t_last;
    out[k] = cur0[chain * d + i].z / T;  // (src/trace.jl:172 divides every term)
    if (i == 0 && T_out) T_out[q] = T;
}

int launch_consume_inclusion(int64_t d, int64_t chain_first, int64_t n, const void* cur, const void* meta, double* out, double* T_out, void* stream) {
    const int64_t tot = n * d;
    hipLaunchKernelGGL(consume_inclusion_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d, chain_first, n,
                       static_cast<const ConsumeCursor*>(cur), static_cast<const ConsumeMeta*>(meta), out, T_out);
    return (int)hipGetLastError();
}

size_t consume_cursor_bytes() { return sizeof(ConsumeCursor); }
size_t consume_meta_bytes() { return sizeof(ConsumeMeta); }

int launch_consume_init(const ZzRec* rec, int64_t rec_stride, int64_t d, int64_t nchains, double t0, void* cur, void* meta, double* grid, int64_t K,
                        void* stream) {
    const int64_t n = nchains * d;
    hipLaunchKernelGGL(consume_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rec, rec_stride, d, nchains, t0,
                       static_cast<ConsumeCursor*>(cur), static_cast<ConsumeMeta*>(meta), grid, K);
    return (int)hipGetLastError();
}
int launch_consume_events(const pdmp_event* ev, int64_t cap, const DevChain* hdr, int64_t d, int64_t nchains, void* cur, void* meta, double* grid,
                          int64_t K, double t0, double dt, void* stream) {
    hipLaunchKernelGGL(consume_events_kernel, dim3((unsigned)nchains), dim3(256), 0, (hipStream_t)stream, ev, cap, hdr, d,
                       static_cast<ConsumeCursor*>(cur), static_cast<ConsumeMeta*>(meta), grid, K, t0, dt);
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
