// pdmp_kernels.hip -- gfx950 (CDNA4) kernels of the PDMP ensemble engine.
//
// Mapping: ONE CHAIN PER WAVEFRONT (64 lanes), one wavefront per workgroup, persistent over a time slice.
// Chains are independent, so there is no inter-workgroup communication at all; the read-only
// "neighbourhood program" tables are shared through L2 / Infinity Cache.
//
// The event-time queue (reference: binary heap SPriorityQueue, src/priorityqueue.jl) is a two-level
// 64-ary tournament matched to the wave width:
//   level 0: d keys in HBM (one coalesced 512-byte block = 64 keys = one load instruction),
//   level 1: (min, argmin) of each block in LDS; peek = 64-lane strided scan + DPP min-reduction.
// change-key of the popped coordinate = patch its (prefetched) block in registers and re-reduce; a
// neighbour's new key either lowers its block minimum (LDS write only) or, rarely, forces a block rescan.
//
// Arithmetic is written operation-by-operation in the reference's order (no FMA contraction: the file is
// compiled with -ffp-contract=off), so that event sequences are bit-identical to oracle/pdmp_oracle.c.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "../../include/pdmp_detmath.h"
#include "pdmp_engine.hpp"

namespace pdmp {

#define PDMP_INF __builtin_inf()

// ------------------------------------------------------------------------------------------ lane helpers

__device__ __forceinline__ double readlane_f64(double v, int srclane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ uint32_t readlane_u32(uint32_t v, int srclane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, srclane);
}
__device__ __forceinline__ uint32_t uniform_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ double uniform_f64(double v) {
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// One v_min_f64.  Written as the instruction itself: fmin() of a value that came out of a load or a DPP move is preceded by a
// canonicalising v_max_f64 x, x per operand (IEEE-mode minnum lowering), i.e. three DP instructions per minimum in the queue
// reductions.  Keys are never NaN unless a chain has diverged; v_min_f64 then returns the other operand (NaN loses, as +Inf).
__device__ __forceinline__ double min_f64(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double max_f64(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Minimum over the 64 lanes, returned wave-uniform.  4 DPP steps inside each row of 16 lanes (quad xor-1, quad xor-2, half-row
// mirror, row mirror), then row_bcast:15 (row r takes lane 15 of row r-1) and row_bcast:31 (rows 2, 3 take lane 31): lane 63 ends
// with the minimum of the four rows.  Lanes that have no source read 0 and hold garbage afterwards; only lane 63 is read.
__device__ __forceinline__ double wave_min_f64(double v) {
    v = min_f64(v, dpp_f64<0xB1>(v));   // quad_perm [1,0,3,2]
    v = min_f64(v, dpp_f64<0x4E>(v));   // quad_perm [2,3,0,1]
    v = min_f64(v, dpp_f64<0x141>(v));  // row_half_mirror
    v = min_f64(v, dpp_f64<0x140>(v));  // row_mirror
    v = min_f64(v, dpp_f64<0x142>(v));  // row_bcast:15
    v = min_f64(v, dpp_f64<0x143>(v));  // row_bcast:31
    return readlane_f64(v, 63);
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    for (int off = 32; off >= 1; off >>= 1) {
        uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
        v = (o < v) ? o : v;
    }
    return uniform_u32(v);
}

// pos(x) = max(zero(x), x), src/common.jl:8
__device__ __forceinline__ double pos_part(double x) {
    return (x > 0.0) ? x : ((x != x) ? x : 0.0);
}

// poisson_time(a, b, u), src/poissontime.jl:8-30 (device restatement; the oracle has its own)
__device__ __forceinline__ double dev_poisson_time(double a, double b, double u) {
    const double L = pdmp_log(u);
    if (b > 0) {
        const double r = a / b;
        if (a < 0) {
            return sqrt(-L * 2.0 / b) - r;
        } else {
            return sqrt(r * r - L * 2.0 / b) - r;
        }
    } else if (b == 0) {
        if (a > 0) {
            return -L / a;
        } else {
            return PDMP_INF;
        }
    } else {
        if (a <= 0) {
            return PDMP_INF;
        } else if (-L <= -(a * a) / b + (a * a) / (2 * b)) {
            const double r = a / b;
            return -sqrt(r * r - L * 2.0 / b) - r;
        } else {
            return PDMP_INF;
        }
    }
}

// ------------------------------------------------------------------------------------------ init kernel
//
// src/sfact.jl:164-190: t = fill(t0), t_old = copy(t), b[i] = ab(G1,i,x,θ,c,F), Q[i] = poisson_time(b[i], rand(rng))
// with the d uniforms drawn in order i = 0..d-1 (draw index = i), then the refresh clock (draw index d).
// One thread per (chain, coordinate).
__global__ __launch_bounds__(256) void zz_init_kernel(ZzInitParams P) {
    const int64_t chain = blockIdx.x;
    const int64_t i = (int64_t)blockIdx.y * 256 + threadIdx.x;
    const int64_t d = P.d;
    const uint64_t seed = P.seeds ? P.seeds[chain] : (P.seed0 + (uint64_t)chain);
    ZzRec* rec = P.rec + chain * d;
    double* keys = P.keys + chain * P.dk;

    if (i < d) {
        auto x_of = [&](int64_t r) -> double {
            return P.x0 ? P.x0[chain * d + r] : pdmp_randn(seed, PDMP_STREAM_INIT, (uint64_t)r);
        };
        auto th_of = [&](int64_t r) -> double {
            return P.th0 ? P.th0[chain * d + r]
                         : ((pdmp_u01(seed, PDMP_STREAM_INIT, (uint64_t)(d + r)) < 0.5) ? -1.0 : 1.0);
        };
        const double xi = x_of(i), thi = th_of(i);
        double gx = 0.0, gt = 0.0;  // idot(Γ,i,x), idot(Γ,i,θ): src/common.jl:16-24
        for (uint32_t p = P.tb.colptr[i]; p < P.tb.colptr[i + 1]; ++p) {
            const uint32_t r = P.tb.rowval[p];
            const double v = P.tb.bval[p];
            gx += v * x_of(r);
            gt += v * th_of(r);
        }
        const double ci = P.tb.c_shared[i];
        if (P.c_chain) P.c_chain[chain * d + i] = ci;
        double a = ci + (gx - P.tb.gmu_b[i]) * thi;  // src/fact_samplers.jl:51
        double b = ci / 100 + thi * gt;             // :52
        if (P.flow_kind == 1) {  // ab(G, i, x, θ, c, Z::FactBoomerang), src/fact_samplers.jl:58-65
            double zz = 0.0;
            for (uint32_t p = P.tb.colptr[i]; p < P.tb.colptr[i + 1]; ++p) {
                const uint32_t r = P.tb.rowval[p];
                const double dx = x_of(r) - P.mu[r];
                const double tr = th_of(r);
                zz += dx * dx + tr * tr;
            }
            const double z = sqrt(zz);
            const double z2 = xi * xi + thi * thi;
            a = ci * sqrt(z2) * z + z2 * P.diag[i];
            b = 0.0;
        }
        double key;
        if (P.local_bound) {
            // src/local.jl:119-124: b[i] = ab(G, i, x, θ, C::LocalBound, ∇ϕi, vi, Z) (:2-6) from the TARGET's derivatives,
            // τ, renew[i] = next_time(t[i], b[i], rand(rng)) (src/not_fact_samplers.jl:43-50): τ includes t0
            double hx = 0.0, ht = 0.0;
            for (uint32_t p = P.tb.colptr[i]; p < P.tb.colptr[i + 1]; ++p) {
                const uint32_t r = P.tb.rowval[p];
                const double v = P.tb.tval[p];
                hx += v * x_of(r);
                ht += v * th_of(r);
            }
            const double gi = P.tb.gmu_t ? (hx - P.tb.gmu_t[i]) : hx;
            a = ci + gi * thi;
            b = ci / 100 + thi * ht;
            const double hz = 2.0 / ci / fabs(thi);
            const double dt = dev_poisson_time(a, b, pdmp_u01(seed, PDMP_STREAM_MAIN, (uint64_t)i));
            const bool rn = dt > hz;
            key = P.t0 + (rn ? hz : dt);
            P.thf[chain * d + i] = rn ? 1.0 : 0.0;
        } else {
            key = dev_poisson_time(a, b, pdmp_u01(seed, PDMP_STREAM_MAIN, (uint64_t)i));  // :186
        }
        uint64_t fflag = 0;
        if (P.sticky) {
            // src/ss_fact.jl:178-188: the first event of i is the earlier of its reflection proposal and its hitting time of 0
            const double tfreez = (thi * xi >= 0) ? PDMP_INF : (-xi / thi);  // freezing_time, :10-16
            if (key > tfreez) {
                fflag = 1;
                key = P.t0 + tfreez;
            } else {
                key = P.t0 + key;
            }
            if (P.thf) P.thf[chain * d + i] = 0.0;
        }
        if (P.track) {
            // tracked-gradient records: the target's sums next to the bound's (gx, gt above ARE the bound's sums at t0)
            double hx = 0.0, ht = 0.0;
            for (uint32_t p = P.tb.colptr[i]; p < P.tb.colptr[i + 1]; ++p) {
                const uint32_t r = P.tb.rowval[p];
                const double v = P.tb.tval[p];
                hx += v * x_of(r);
                ht += v * th_of(r);
            }
            TrRec r;
            r.x = xi;
            r.th = thi;
            r.tx = P.t0;
            r.I = 0.0;
            r.g = hx;
            r.gd = ht;
            r.tg = P.t0;
            r.acc = 0;
            r.a = a;
            r.b = b;
            r.t_old = P.t0;
            r.tprop = P.t0;
            r.gb = gx;
            r.gdb = gt;
            r.tacc = P.t0;
            r.pad = 0.0;
            (reinterpret_cast<TrRec*>(P.rec) + chain * d)[i] = r;
            keys[i] = key;
        } else {
        ZzRec r;
        r.x = xi;
        r.th = thi;
        r.t = P.t0;
        r.I = 0.0;
        r.t_old = P.t0;
        r.a = a;
        r.b = b;
        r.acc = fflag;  // sticky: f[i] ("the next event of i is a freeze"), src/ss_fact.jl:165; otherwise acc[i] = 0
        rec[i] = r;
        keys[i] = key;
        }
    } else if (i < P.dk) {
        double key = PDMP_INF;
        if (i == d && P.has_refresh) {
            // src/sfact.jl:189: enqueue!(Q, n+1 => waiting_time_ref(rng, F)) = randexp(rng)/λref
            key = -pdmp_log(pdmp_u01(seed, PDMP_STREAM_MAIN, (uint64_t)d)) / P.lambda_ref;
        }
        keys[i] = key;
    }
    if (i == 0) {
        DevChain h;
        h.c.t_last = P.t0;
        h.c.num = 0;
        h.c.nacc = 0;
        h.c.nrefresh = 0;
        h.c.ntrace = 0;
        h.c.nevents = 0;
        h.c.ndraw_main = (uint64_t)d + (P.has_refresh ? 1u : 0u);
        h.c.ndraw_global = 0;
        h.c.status = PDMP_CHAIN_OK;
        h.c.reserved = 0;
        h.seed = seed;
        h.t0 = P.t0;
        h.t_event = P.t0;
        h.tl_scale = 0.0;
        for (int k = 0; k < 3; ++k) h.pad[k] = 0;
        P.hdr[chain] = h;
    }
}

// ------------------------------------------------------------------------------------------ event loop
//
// spdmp_inner! (src/sfact.jl:73-145) under the driver loop `while t′ < T` (:199-208), G = Matched().
//
// Dependent memory levels per proposal: LDS peek -> {neighbourhood blob (L2/MALL), rec[i], popped key block}
// -> neighbour records (HBM).  Everything a proposal needs that is a function of i alone lives in ONE
// contiguous blob (built on the host, pdmp_capi.hip: build_blob) fetched with a single coalesced wave load.
// A workgroup is one wavefront, so no s_barrier / vmcnt(0) fence is ever needed: DS operations of a wave
// execute in order, and stores are fire-and-forget.

size_t zz_local_lds_bytes(uint32_t nblk_pad, uint32_t blob_w_pad) {
    return (size_t)nblk_pad * 8 + 3 * 64 * 8 + (size_t)blob_w_pad * 8 + (size_t)nblk_pad * 4;
}

// poisson_time(a, b, u) with L = log(u) supplied (src/poissontime.jl:8-30)
__device__ __forceinline__ double dev_poisson_time_L(double a, double b, double L) {
    // The three b != 0 formulas share a / b, L * 2 / b and the square root (sqrt(-L * 2.0 / b) is sqrt(-(L * 2.0 / b)) bit for
    // bit), so a wavefront whose lanes disagree on the signs of a and b runs ONE division pair and ONE square root instead of
    // one set per branch; only the admissibility test of the b < 0 branch keeps its own two divisions.
    if (b == 0) return (a > 0) ? -L / a : PDMP_INF;
    const double r = a / b;
    const double q = L * 2.0 / b;
    const double sq = sqrt((b > 0 && a < 0) ? -q : r * r - q);
    if (b > 0) return sq - r;
    if (a <= 0) return PDMP_INF;
    if (-L <= -(a * a) / b + (a * a) / (2 * b)) return -sq - r;
    return PDMP_INF;
}

// Cross-lane hand-off through LDS inside ONE wavefront: DS operations execute in issue order, so no s_barrier
// and no s_waitcnt vmcnt(0) is needed -- but the COMPILER must be told that memory changed behind the thread's
// back (otherwise it may forward a lane's own earlier store to its later load of the same slot).
#define LDS_ORDER()                      \
    do {                                 \
        __builtin_amdgcn_wave_barrier(); \
        asm volatile("" ::: "memory");   \
    } while (0)

// Level-1 entry (block minimum, argmin) of the block of coordinate j after keys[j] became kj: smaller than the entry -> replace;
// j WAS the entry and grew -> rescan the 64 keys of the block (lowest index on ties); otherwise nothing to do.
__device__ __forceinline__ void level1_update(double* bk, uint32_t* bi, const double* keys, int lane, uint32_t j, double kj) {
    const uint32_t bj = j >> 6;
    LDS_ORDER();
    const double cur = bk[bj];
    const uint32_t ci = bi[bj];
    if (kj < cur || (kj == cur && j < ci)) {
        if (lane == 0) {
            bk[bj] = kj;
            bi[bj] = j;
        }
    } else if (ci == j) {
        const double kv = __hip_atomic_load(keys + (size_t)bj * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double mn = wave_min_f64(kv);
        const uint64_t bl = __ballot(kv == mn);
        const int arg = bl ? (__ffsll((unsigned long long)bl) - 1) : 0;
        if (lane == 0) {
            bk[bj] = mn;
            bi[bj] = bj * 64 + (uint32_t)arg;
        }
    }
}


__global__ __launch_bounds__(64) void zz_local_run_kernel(ZzRunParams P) {
    const int lane = threadIdx.x;
    const int64_t chain = blockIdx.x;
    const int64_t d = P.d;
    const uint32_t nblk = P.nblk;
    const uint32_t W = P.blob_w, SW = P.blob_sw, PW = P.blob_pw, KMAX = P.blob_kmax;
    const uint32_t R = 4 + PW + KMAX;

    extern __shared__ __align__(16) unsigned char smem[];
    double* bk = reinterpret_cast<double*>(smem);  // [nblk_pad] block minima
    double* sx = bk + P.nblk_pad;                  // [64] x of S[i] after the move
    double* sth = sx + 64;                         // [64] θ of S[i]
    double* pk = sth + 64;                         // [64] patched copy of the popped key block
    uint64_t* lb = reinterpret_cast<uint64_t*>(pk + 64);               // [blob_w_pad] neighbourhood blob of i
    uint32_t* bi = reinterpret_cast<uint32_t*>(lb + P.blob_w_pad);     // [nblk_pad] block argmin (coordinate id)

    ZzRec* rec = P.rec + chain * d;
    double* keys = P.keys + chain * P.dk;
    DevChain* hdr = P.hdr + chain;
    pdmp_event* ev = P.ev ? P.ev + chain * P.trace_cap : nullptr;
    double* cmut = P.c_chain ? (P.c_chain + chain * d) : nullptr;

    uint32_t status = hdr->c.status;
    if (status == PDMP_CHAIN_BOUND_VIOLATED || status == PDMP_CHAIN_STALLED) return;
    const uint64_t seed = hdr->seed;
    uint64_t nm = hdr->c.ndraw_main, ng = hdr->c.ndraw_global;
    uint64_t num = hdr->c.num, nacc = hdr->c.nacc, ntrace = hdr->c.ntrace, nevents = hdr->c.nevents;
    uint64_t nrefresh = hdr->c.nrefresh;
    double t_last = hdr->c.t_last;
    double t_event = hdr->t_event;
    status = PDMP_CHAIN_OK;

    const double T = P.T;
    const bool stop_before = (P.flags & PDMP_RUN_STOP_BEFORE) != 0;
    const bool adapt = P.adapt != 0;
    const bool has_refresh = P.has_refresh != 0;
    const bool move_all = P.move_all != 0;

    // smove_forward!(::All, ...) = move every coordinate (src/sfact.jl:19,23-28): the `pdmp` driver, G = All()
    auto sweep_all = [&](double tnew) {
        for (int64_t q = lane; q < d; q += 64) {
            ZzRec* r = rec + q;
            const double x0 = r->x, th0 = r->th, t0 = r->t, I0 = r->I;
            const double dt = tnew - t0;
            const double xn = x0 + th0 * dt;
            r->x = xn;
            r->t = tnew;
            r->I = I0 + dt * ((x0 + xn) * 0.5);
        }
    };
    // generic level-1 update for one changed key (j, kj) whose new value is already stored in keys[]
    auto queue_update = [&](uint32_t j, double kj) {
        level1_update(bk, bi, keys, lane, j, kj);
        LDS_ORDER();
    };

    // ---- rebuild level 1 of the queue from the keys in HBM (each lane scans whole blocks)
    for (uint32_t b = lane; b < nblk; b += 64) {
        const double* kp = keys + (size_t)b * 64;
        double mk = kp[0];
        uint32_t mi = 0;
#pragma unroll 8
        for (int q = 1; q < 64; ++q) {
            const double v = kp[q];
            if (v < mk) {
                mk = v;
                mi = q;
            }
        }
        bk[b] = mk;
        bi[b] = b * 64 + mi;
    }
    LDS_ORDER();

    bool running = stop_before || (t_event < T);  // `while t′ < T`, src/sfact.jl:199
    PrioTurn prio;
    while (running) {
        prio.step();
        if (P.trace_cap > 0 && ntrace >= (uint64_t)P.trace_cap) {
            status = PDMP_CHAIN_TRACE_FULL;
            break;
        }
        // ---------------- peek(Q), src/sfact.jl:77
        double mk = PDMP_INF;
        uint32_t mb = 0xffffffffu;
        for (uint32_t b = lane; b < nblk; b += 64) {
            const double v = bk[b];
            if (v < mk) {
                mk = v;
                mb = b;
            }
        }
        const double tp = wave_min_f64(mk);
        if (!(tp < PDMP_INF)) {  // +Inf (or NaN): nothing can happen any more
            status = PDMP_CHAIN_STALLED;
            break;
        }
        if (stop_before && !(tp < T)) break;
        const uint64_t ball = __ballot(mk == tp);
        uint32_t blk;
        if (__popcll(ball) == 1) {
            blk = readlane_u32(mb, __ffsll((unsigned long long)ball) - 1);
        } else {  // exact tie between blocks: lowest coordinate wins
            blk = wave_min_u32((mk == tp) ? mb : 0xffffffffu);
        }
        const uint32_t i = uniform_u32(bi[blk]);
        t_last = tp;

        if (has_refresh && i == (uint32_t)d) {
            // ---------------- refresh clock popped: src/sfact.jl:78-114 (restated with its quirks: the coordinate whose
            // neighbourhood is moved (:80) and the coordinate that is refreshed (:84) are two independent draws from the
            // "global rng" stream, and G1[i] is re-bounded at the coordinates' own, possibly stale, clocks)
            const uint32_t i1 = pdmp_randint(seed, PDMP_STREAM_GLOBAL, ng, (uint32_t)d);
            ng += 1;
            if (move_all) {
                sweep_all(tp);
            } else {
                const uint64_t* bsrc = P.blob + (size_t)P.tix[i1] * P.blob_w_pad;
                for (uint32_t w = lane; w < W; w += 64) lb[w] = bsrc[w];
                LDS_ORDER();
                const int k1 = (int)uniform_u32((uint32_t)(lb[0] & 0xff));
                if (lane < k1) {
                    const uint64_t sw = lb[1 + (lane >> 1)];
                    const uint32_t s1 = i1 + ((lane & 1) ? (uint32_t)(sw >> 32) : (uint32_t)sw);  // ids are stored relative to i
                    ZzRec* r1 = rec + s1;
                    const double x0 = r1->x, th0 = r1->th, t0 = r1->t, I0 = r1->I;
                    const double dt = tp - t0;
                    const double xn = x0 + th0 * dt;
                    r1->x = xn;
                    r1->t = tp;
                    r1->I = I0 + dt * ((x0 + xn) * 0.5);
                }
                LDS_ORDER();
            }
            const uint32_t i2 = pdmp_randint(seed, PDMP_STREAM_GLOBAL, ng, (uint32_t)d);
            ng += 1;
            {
                const uint64_t* bsrc = P.blob + (size_t)P.tix[i2] * P.blob_w_pad;
                for (uint32_t w = lane; w < W; w += 64) lb[w] = bsrc[w];
            }
            LDS_ORDER();
            const uint64_t hw = lb[0];
            const int k = (int)uniform_u32((uint32_t)(hw & 0xff));
            const int m = (int)uniform_u32((uint32_t)((hw >> 8) & 0xff));
            const int self = (int)uniform_u32((uint32_t)((hw >> 16) & 0xff));
            const int kjmax = (int)uniform_u32((uint32_t)((hw >> 24) & 0xff));
            uint32_t s = i2;
            if (lane < m) {
                const uint64_t sw = lb[1 + (lane >> 1)];
                s = i2 + ((lane & 1) ? (uint32_t)(sw >> 32) : (uint32_t)sw);
            }
            ZzRec* rs = rec + s;
            double x = 0.0, th = 0.0, t = 0.0, I = 0.0;
            if (lane < m) {
                x = rs->x;
                th = rs->th;
                t = rs->t;
                I = rs->I;
            }
            if (!move_all && lane >= k && lane < m) {  // smove_forward!(G2, i, ...), :85
                const double dt = tp - t;
                const double xn = x + th * dt;
                I = I + dt * ((x + xn) * 0.5);
                x = xn;
                t = tp;
            }
            const double usign = pdmp_u01(seed, PDMP_STREAM_MAIN, nm);  // θ[i] = σ[i]*rand(rng, (-1,1)), :100-101
            nm += 1;
            if (lane == self) th = P.tb.sigma[i2] * ((usign < 0.5) ? -1.0 : 1.0);
            // Q[n+1] = t′ + waiting_time_ref(F) = t′ + randexp()/λref from the global rng, :108
            const double newref = tp + (-pdmp_log(pdmp_u01(seed, PDMP_STREAM_GLOBAL, ng))) / P.lambda_ref;
            ng += 1;
            if (lane < m) {
                sx[lane] = x;
                sth[lane] = th;
            }
            LDS_ORDER();
            const uint32_t sub = 1 + SW + (uint32_t)lane * R;
            double key = PDMP_INF;
            if (lane < k) {  // :110-114
                const double gmu = __longlong_as_double((long long)lb[sub + 1]);
                const double cj = cmut ? cmut[s] : __longlong_as_double((long long)lb[sub + 2]);
                const int kj = (int)(lb[sub + 3] & 0xff);
                double gx = 0.0, gt = 0.0;
                for (int base = 0; base < kjmax; base += 8) {
                    const uint64_t pw = lb[sub + 4 + (base >> 3)];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int pp = base + q;
                        if (pp < kj) {
                            const double v = __longlong_as_double((long long)lb[sub + 4 + PW + pp]);
                            const int ps = (int)((pw >> (8 * q)) & 0xff);
                            gx += v * sx[ps];
                            gt += v * sth[ps];
                        }
                    }
                }
                const double a = cj + (gx - gmu) * th;
                const double b = cj / 100 + th * gt;
                const double L = pdmp_log(pdmp_u01(seed, PDMP_STREAM_MAIN, nm + (uint64_t)lane));
                key = t + dev_poisson_time_L(a, b, L);  // Q[j] = t[j] + poisson_time(...): t[j] is j's OWN clock here
                rs->t_old = t;
                rs->a = a;
                rs->b = b;
                keys[s] = key;
            }
            nm += (uint64_t)k;
            if (lane < m) {
                rs->x = x;
                rs->th = th;
                rs->t = t;
                rs->I = I;
            }
            if (lane == 0) keys[d] = newref;
            for (int jj = 0; jj <= k; ++jj) {
                const uint32_t j = (jj < k) ? readlane_u32(s, jj) : (uint32_t)d;
                const double kj = (jj < k) ? readlane_f64(key, jj < k ? jj : 0) : newref;
                queue_update(j, kj);
            }
            const double t_i = readlane_f64(t, self), x_i = readlane_f64(x, self), th_i2 = readlane_f64(th, self);
            if (ev && lane == 0) {  // event(i, t, x, θ, F) = (t[i], i, x[i], θ[i]), :143
                pdmp_event e;
                e.t = t_i;
                e.i = (int64_t)i2;
                e.x = x_i;
                e.theta = th_i2;
                ev[ntrace] = e;
            }
            nrefresh += 1;
            ntrace += 1;
            nevents += 1;
            t_event = tp;
            if (!stop_before && !(tp < T)) running = false;
            continue;
        }
        if (move_all) sweep_all(tp);

        // ---------------- level-1 loads: everything that is a function of i alone
        {
            const uint64_t* bsrc = P.blob + (size_t)P.tix[i] * P.blob_w_pad;
            for (uint32_t w = lane; w < W; w += 64) lb[w] = bsrc[w];
        }
        const ZzRec* ri = rec + i;
        const double told_i = ri->t_old, a_i = ri->a, b_i = ri->b;
        const uint64_t acc_i = ri->acc;
        const double kb0 = keys[(size_t)blk * 64 + lane];

        // ---------------- random numbers of this proposal, computed under the memory latency.
        // draw nm is the thinning coin (:121); draw nm+1+jj re-bounds the jj-th member of G1[i] on accept (:134),
        // draw nm+1 re-bounds i on reject (:139).  Lane 63 evaluates the coin, lane jj its own re-bound draw.
        double ucoin, Llane;
        if (KMAX < 64) {
            const uint64_t idx = (lane == 63) ? nm : (nm + 1 + (uint64_t)lane);
            const double u = pdmp_u01(seed, PDMP_STREAM_MAIN, idx);
            ucoin = readlane_f64(u, 63);
            Llane = pdmp_log(u);
        } else {
            ucoin = pdmp_u01(seed, PDMP_STREAM_MAIN, nm);
            Llane = pdmp_log(pdmp_u01(seed, PDMP_STREAM_MAIN, nm + 1 + (uint64_t)lane));
        }
        // read lane 0's draw HERE, in wave-uniform control flow: inside the divergent re-bound block below lane 0
        // may be inactive, and the compiler is free to sink the computation of Llane into that block.
        const double L_reject = readlane_f64(Llane, 0);

        // ---------------- neighbourhood header and member list from the blob (now in LDS)
        LDS_ORDER();
        const uint64_t hw = lb[0];
        const int k = (int)uniform_u32((uint32_t)(hw & 0xff));
        const int m = (int)uniform_u32((uint32_t)((hw >> 8) & 0xff));
        const int self = (int)uniform_u32((uint32_t)((hw >> 16) & 0xff));
        const int kjmax = (int)uniform_u32((uint32_t)((hw >> 24) & 0xff));
        uint32_t s = i;
        if (lane < m) {
            const uint64_t sw = lb[1 + (lane >> 1)];
            s = i + ((lane & 1) ? (uint32_t)(sw >> 32) : (uint32_t)sw);
        }
        ZzRec* rs = rec + s;
        // ---------------- level-2 loads: positions of G[i] and (speculatively) of G2[i]
        double x = 0.0, th = 0.0, t = 0.0, I = 0.0;
        if (lane < m) {
            x = rs->x;
            th = rs->th;
            t = rs->t;
            I = rs->I;
        }
        const uint32_t sub = 1 + SW + (uint32_t)lane * R;  // this lane's sub-record (valid for lane < k)
        double tv = 0.0, cj = 0.0;
        if (lane < k) {
            tv = __longlong_as_double((long long)lb[sub + 0]);
            cj = cmut ? cmut[s] : __longlong_as_double((long long)lb[sub + 2]);
        }

        // ---------------- smove_forward!(G, i, t, x, θ, t′, F), src/sfact.jl:6-12,82
        if (lane < k) {
            const double dt = tp - t;
            const double xn = x + th * dt;
            I = I + dt * ((x + xn) * 0.5);
            x = xn;
            t = tp;
        }
        // ---------------- ∇ϕ(x, i) = idot(Γt, i, x) sequentially in ascending row order, src/common.jl:16-24
        double g = 0.0;
        for (int p = 0; p < k; ++p) g += readlane_f64(tv, p) * readlane_f64(x, p);
        if (P.tb.gmu_t) g = g - P.tb.gmu_t[i];
        const double th_i = readlane_f64(th, self);
        const double l = pos_part(g * th_i);                    // sλ, src/sfact.jl:69,119
        const double lbound = pos_part(a_i + b_i * (tp - told_i));  // sλ̄, :70,119 (t[i] == t′ after the move)
        num += 1;                                               // :120
        const bool accept = (ucoin * lbound < l);               // :121
        bool violated = false;
        int nmoved = k;
        if (accept) {
            nacc += 1;               // acc[i] += 1, :122
            violated = (l >= lbound);  // :123
            if (violated && !adapt) {
                // reference: error("Tuning parameter `c` too small."), :124 -> per-chain status word
                status = PDMP_CHAIN_BOUND_VIOLATED;
                if (lane < k) {
                    rs->x = x;
                    rs->t = t;
                    rs->I = I;
                }
                nm += 1;
                break;
            }
            // smove_forward!(G2, i, ...), :129
            if (lane >= k && lane < m) {
                const double dt = tp - t;
                const double xn = x + th * dt;
                I = I + dt * ((x + xn) * 0.5);
                x = xn;
                t = tp;
            }
            nmoved = m;
            if (lane == self) th = -th;  // reflect!, src/dynamics.jl:46-49, :130
        }
        // ---------------- stage (x, θ) of the moved neighbourhood for the re-bounding lanes
        if (lane < nmoved) {
            sx[lane] = x;
            sth[lane] = th;
        }
        pk[lane] = kb0;
        LDS_ORDER();

        // ---------------- ab + new event time for j in G1[i] (accept, :131-135) or for i alone (reject, :137-139)
        const bool active = accept ? (lane < k) : (lane == self);
        double key = PDMP_INF;
        if (active) {
            const double gmu = __longlong_as_double((long long)lb[sub + 1]);
            const int kj = (int)(lb[sub + 3] & 0xff);
            double gx = 0.0, gt = 0.0;
            for (int base = 0; base < kjmax; base += 8) {
                const uint64_t pw = lb[sub + 4 + (base >> 3)];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int pp = base + q;
                    if (pp < kj) {
                        const double v = __longlong_as_double((long long)lb[sub + 4 + PW + pp]);
                        const int ps = (int)((pw >> (8 * q)) & 0xff);
                        gx += v * sx[ps];
                        gt += v * sth[ps];
                    }
                }
            }
            if (violated && lane == self) {  // adapt!(c, i, factor), src/fact_samplers.jl:67-70, :127
                cj *= P.factor;
                cmut[s] = cj;
            }
            const double a = cj + (gx - gmu) * th;  // src/fact_samplers.jl:51
            const double b = cj / 100 + th * gt;   // :52
            const double L = accept ? Llane : L_reject;
            key = t + dev_poisson_time_L(a, b, L);  // Q[j] = t[j] + poisson_time(b[j], rand(rng))
            rs->t_old = t;                          // t_old[j] = t[j]
            rs->a = a;
            rs->b = b;
            keys[s] = key;
            if ((s >> 6) == blk) pk[s & 63] = key;
        }
        if (P.dbg && chain == 0 && (int64_t)(num - 1) < P.dbg_cap) {
            double* D = P.dbg + (num - 1) * 16;
            const double ks = readlane_f64(key, self), ts = readlane_f64(t, self), Ls = readlane_f64(Llane, self);
            const double xs0 = readlane_f64(x, 0), cjs = readlane_f64(cj, self);
            if (lane == 0) {
                D[0] = tp; D[1] = (double)i; D[2] = accept ? 1.0 : 0.0; D[3] = (double)k; D[4] = (double)m;
                D[5] = (double)self; D[6] = l; D[7] = lbound; D[8] = ucoin; D[9] = ks; D[10] = ts; D[11] = Ls;
                D[12] = g; D[13] = a_i; D[14] = xs0; D[15] = cjs;
            }
        }
        nm += accept ? (uint64_t)(1 + k) : 2u;

        // ---------------- write back the moved coordinates
        if (lane < nmoved) {
            rs->x = x;
            rs->th = th;
            rs->t = t;
            rs->I = I;
        }
        if (accept && lane == self) rs->acc = acc_i + 1;

        // ---------------- queue: re-reduce the popped block from the patched copy
        LDS_ORDER();
        {
            const double kb = pk[lane];
            const double mn = wave_min_f64(kb);
            const uint64_t bl = __ballot(kb == mn);
            const int arg = bl ? (__ffsll((unsigned long long)bl) - 1) : 0;
            if (lane == 0) {
                bk[blk] = mn;
                bi[blk] = blk * 64 + (uint32_t)arg;
            }
        }
        // ---------------- queue: neighbours that live in other blocks
        if (accept) {
            for (int jj = 0; jj < k; ++jj) {
                const uint32_t j = readlane_u32(s, jj);
                const uint32_t bj = j >> 6;
                if (bj == blk) continue;
                const double kj = readlane_f64(key, jj);
                LDS_ORDER();
                const double cur = bk[bj];
                const uint32_t ci = bi[bj];
                if (kj < cur || (kj == cur && j < ci)) {
                    if (lane == 0) {
                        bk[bj] = kj;
                        bi[bj] = j;
                    }
                } else if (ci == j) {
                    // j was its block's minimum and moved later: rescan that block (keys[] already updated;
                    // same-wave store -> load to one address is ordered by the memory pipeline)
                    const double kv = __hip_atomic_load(keys + (size_t)bj * 64 + lane, __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT);
                    const double mn = wave_min_f64(kv);
                    const uint64_t bl = __ballot(kv == mn);
                    const int arg = bl ? (__ffsll((unsigned long long)bl) - 1) : 0;
                    if (lane == 0) {
                        bk[bj] = mn;
                        bi[bj] = bj * 64 + (uint32_t)arg;
                    }
                }
            }
            // ---------------- event(i, t, x, θ, F) = (t[i], i, x[i], θ[i]), src/sfact.jl:50-52,143
            const double x_i = readlane_f64(x, self);
            const double thn_i = readlane_f64(th, self);
            if (ev && lane == 0) {
                pdmp_event e;
                e.t = tp;
                e.i = (int64_t)i;
                e.x = x_i;
                e.theta = thn_i;
                ev[ntrace] = e;
            }
            ntrace += 1;
            nevents += 1;
            t_event = tp;
            if (!stop_before && !(tp < T)) running = false;  // `while t′ < T`
        }
        LDS_ORDER();
    }

    if (lane == 0) {
        hdr->c.t_last = t_last;
        hdr->t_event = t_event;
        hdr->c.num = num;
        hdr->c.nacc = nacc;
        hdr->c.ntrace = ntrace;
        hdr->c.nevents = nevents;
        hdr->c.ndraw_main = nm;
        hdr->c.ndraw_global = ng;
        hdr->c.nrefresh = nrefresh;
        hdr->c.status = status;
    }
}

// ------------------------------------------------------------------------------------------ sticky ZigZag
//
// sspdmp_inner! (src/ss_fact.jl:78-157) under the driver loop `while t′ < T` (:202-211): three kinds of popped key --
// freeze (f[i]: x_i hits 0, the coordinate sticks, its thaw clock −log(rand())/κ_i is queued), thaw (x_i == 0 && θ_i == 0:
// the saved speed θf[i] is restored) and reflection proposal (as spdmp_inner!, but frozen coordinates neither move nor get
// re-bounded).  One event per iteration; the record's `acc` word holds f[i], θf lives in its own per-chain array.
// Draws (the reference uses the global rng for all of them): draw nm is the thaw time / reversible sign / thinning coin,
// then one draw per re-bounded (non-frozen) coordinate in ascending order.

size_t zz_sticky_lds_bytes(uint32_t nblk_pad, uint32_t blob_w_pad) {
    return (size_t)nblk_pad * 8 + 4 * 64 * 8 + (size_t)blob_w_pad * 8 + (size_t)nblk_pad * 4;
}

__global__ __launch_bounds__(64) void zz_sticky_run_kernel(ZzRunParams P) {
    const int lane = threadIdx.x;
    const int64_t chain = blockIdx.x;
    const int64_t d = P.d;
    const uint32_t nblk = P.nblk;
    const uint32_t W = P.blob_w, SW = P.blob_sw, PW = P.blob_pw, KMAX = P.blob_kmax;
    const uint32_t R = 4 + PW + KMAX;

    extern __shared__ __align__(16) unsigned char smem[];
    double* bk = reinterpret_cast<double*>(smem);
    double* sx = bk + P.nblk_pad;
    double* sth = sx + 64;
    double* LU = sth + 64;   // logs of the 64 candidate draws nm + lane
    double* UU = LU + 64;    // the draws themselves
    uint64_t* lb = reinterpret_cast<uint64_t*>(UU + 64);
    uint32_t* bi = reinterpret_cast<uint32_t*>(lb + P.blob_w_pad);

    ZzRec* rec = P.rec + chain * d;
    double* keys = P.keys + chain * P.dk;
    double* thf = P.thf + chain * d;
    DevChain* hdr = P.hdr + chain;
    pdmp_event* ev = P.ev ? P.ev + chain * P.trace_cap : nullptr;
    double* cmut = P.c_chain ? (P.c_chain + chain * d) : nullptr;

    uint32_t status = hdr->c.status;
    if (status == PDMP_CHAIN_BOUND_VIOLATED || status == PDMP_CHAIN_STALLED) return;
    const uint64_t seed = hdr->seed;
    uint64_t nm = hdr->c.ndraw_main;
    uint64_t num = hdr->c.num, nacc = hdr->c.nacc, ntrace = hdr->c.ntrace, nevents = hdr->c.nevents;
    double t_last = hdr->c.t_last;
    double t_event = hdr->t_event;
    status = PDMP_CHAIN_OK;
    const double T = P.T;
    const bool stop_before = (P.flags & PDMP_RUN_STOP_BEFORE) != 0;
    const bool adapt = P.adapt != 0;

    for (uint32_t b = lane; b < nblk; b += 64) {
        const double* kp = keys + (size_t)b * 64;
        double mk = kp[0];
        uint32_t mi = 0;
#pragma unroll 8
        for (int q = 1; q < 64; ++q) {
            const double v = kp[q];
            if (v < mk) {
                mk = v;
                mi = q;
            }
        }
        bk[b] = mk;
        bi[b] = b * 64 + mi;
    }
    LDS_ORDER();

    auto queue_update = [&](uint32_t j, double kj) {
        level1_update(bk, bi, keys, lane, j, kj);
        LDS_ORDER();
    };

    bool running = stop_before || (t_event < T);
    PrioTurn prio;
    while (running) {
        prio.step();
        if (P.trace_cap > 0 && ntrace >= (uint64_t)P.trace_cap) {
            status = PDMP_CHAIN_TRACE_FULL;
            break;
        }
        // ---------------- peek(Q), src/ss_fact.jl:83
        double mk = PDMP_INF;
        uint32_t mb = 0xffffffffu;
        for (uint32_t b = lane; b < nblk; b += 64) {
            const double v = bk[b];
            if (v < mk) {
                mk = v;
                mb = b;
            }
        }
        const double tp = wave_min_f64(mk);
        if (!(tp < PDMP_INF)) {
            status = PDMP_CHAIN_STALLED;
            break;
        }
        if (stop_before && !(tp < T)) break;
        const uint64_t ball = __ballot(mk == tp);
        uint32_t blk;
        if (__popcll(ball) == 1) {
            blk = readlane_u32(mb, __ffsll((unsigned long long)ball) - 1);
        } else {
            blk = wave_min_u32((mk == tp) ? mb : 0xffffffffu);
        }
        const uint32_t i = uniform_u32(bi[blk]);
        t_last = tp;

        {
            const uint64_t* bsrc = P.blob + (size_t)P.tix[i] * P.blob_w_pad;
            for (uint32_t w = lane; w < W; w += 64) lb[w] = bsrc[w];
        }
        const ZzRec* ri = rec + i;
        const double told_i = ri->t_old, a_i = ri->a, b_i = ri->b;
        const bool f_i = ri->acc != 0;
        const double x_i0 = ri->x, th_i0 = ri->th;
        {
            const double u = pdmp_u01(seed, PDMP_STREAM_MAIN, nm + (uint64_t)lane);
            UU[lane] = u;
            LU[lane] = pdmp_log(u);
        }
        LDS_ORDER();
        const uint64_t hw = lb[0];
        const int k = (int)uniform_u32((uint32_t)(hw & 0xff));
        const int m = (int)uniform_u32((uint32_t)((hw >> 8) & 0xff));
        const int self = (int)uniform_u32((uint32_t)((hw >> 16) & 0xff));
        const int kjmax = (int)uniform_u32((uint32_t)((hw >> 24) & 0xff));
        uint32_t s = i;
        if (lane < m) {
            const uint64_t sw = lb[1 + (lane >> 1)];
            s = i + ((lane & 1) ? (uint32_t)(sw >> 32) : (uint32_t)sw);
        }
        ZzRec* rs = rec + s;
        double x = 0.0, th = 0.0, t = 0.0, I = 0.0;
        if (lane < m) {
            x = rs->x;
            th = rs->th;
            t = rs->t;
            I = rs->I;
        }
        const uint32_t sub = 1 + SW + (uint32_t)lane * R;
        double tv = 0.0, cj = 0.0;
        if (lane < k) {
            tv = __longlong_as_double((long long)lb[sub + 0]);
            cj = cmut ? cmut[s] : __longlong_as_double((long long)lb[sub + 2]);
        }
        auto move_lane = [&]() {  // t[i], x[i] = t′, x[i] + θ[i]*(t′ - t[i])
            const double dt = tp - t;
            const double xn = x + th * dt;
            I = I + dt * ((x + xn) * 0.5);
            x = xn;
            t = tp;
        };

        const bool is_freeze = uniform_u32(f_i ? 1u : 0u) != 0;
        const bool is_thaw = !is_freeze && uniform_u32((x_i0 == 0 && th_i0 == 0) ? 1u : 0u) != 0;
        uint32_t ndraw0 = 0;     // draws consumed before the per-coordinate re-bound draws
        bool rebound_set = false;  // lane takes part in the re-bound
        bool emit = true;
        bool moved_hi = false;     // lanes k..m were (ss)moved
        double key_self_extra = PDMP_INF;  // freeze: the thaw clock of i (i itself is not re-bounded)
        bool violated = false;
        double told_self = 0.0;
        bool write_self_bound = false;

        if (is_freeze) {  // ---- case 1, :87-107
            if (lane == self) move_lane();  // smove_forward!(i, ...), :88
            const double xs = readlane_f64(x, self);
            if (fabs(xs) > 1e-8) {  // :89-91 error("x[i] = ... !≈ 0")
                status = PDMP_CHAIN_BOUND_VIOLATED;
                break;
            }
            if (lane == self) {
                thf[s] = th;      // θf[i], θ[i] = θ[i], 0.0, :93
                x = 0.0 * th;     // x[i] = -0*θ[i], :92 (Int -0 == 0: the sign is θ's)
                th = 0.0;
            }
            key_self_extra = tp - LU[0] / P.kappa[i];  // Q[i] = t[i] - log(rand())/κ[i], :96
            ndraw0 = 1;
            told_self = tp;
            write_self_bound = true;
            if (!P.strong_upperbounds) {  // :97-107
                if (lane < m && th != 0.0) move_lane();
                moved_hi = true;
                rebound_set = (lane < k) && (th != 0.0);
            }
        } else if (is_thaw) {  // ---- case 2, :108-123
            if (lane == self) {
                t = tp;          // :109
                th = thf[s];     // θ[i], θf[i] = θf[i], 0.0, :110
                thf[s] = 0.0;
                if (P.reversible) th *= (UU[0] < 0.5) ? -1.0 : 1.0;  // :111-113
            }
            ndraw0 = P.reversible ? 1u : 0u;
            if (lane < m && th != 0.0) move_lane();  // :115-116 (i itself: x + θ*0)
            moved_hi = true;
            rebound_set = (lane < k) && (th != 0.0);  // :117-123
        } else {  // ---- reflection proposal, :124-152
            if (lane < k && th != 0.0) move_lane();  // :125
            double g = 0.0;
            for (int p = 0; p < k; ++p) g += readlane_f64(tv, p) * readlane_f64(x, p);
            if (P.tb.gmu_t) g = g - P.tb.gmu_t[i];
            const double th_s = readlane_f64(th, self);
            const double l = pos_part(g * th_s);
            const double lbound = pos_part(a_i + b_i * (tp - told_i));  // :128
            num += 1;
            ndraw0 = 1;
            if (UU[0] * lbound < l) {  // :130
                nacc += 1;
                if (l > lbound) {  // :132
                    if (!adapt) {
                        status = PDMP_CHAIN_BOUND_VIOLATED;
                        if (lane < k) {
                            rs->x = x;
                            rs->t = t;
                            rs->I = I;
                        }
                        nm += 1;
                        break;
                    }
                    nacc = 0;  // acc = num = 0, :134
                    num = 0;
                    violated = true;
                }
                if (lane >= k && lane < m && th != 0.0) move_lane();  // :138
                moved_hi = true;
                if (lane == self) th = -th;  // :139
                rebound_set = (lane < k) && (th != 0.0);
            } else {  // :147-151
                rebound_set = (lane == self);
                emit = false;
            }
        }
        const int nst = moved_hi ? m : k;
        if (lane < nst) {
            sx[lane] = x;
            sth[lane] = th;
        }
        LDS_ORDER();
        // ---------------- ab + queue_time! for the re-bound set, :54-65
        const uint64_t rball = __ballot(rebound_set);
        const uint32_t rank = (uint32_t)__popcll(rball & ((1ull << lane) - 1ull));
        double key = PDMP_INF;
        if (rebound_set) {
            const double gmu = __longlong_as_double((long long)lb[sub + 1]);
            const int kj = (int)(lb[sub + 3] & 0xff);
            double gx = 0.0, gt = 0.0;
            for (int base = 0; base < kjmax; base += 8) {
                const uint64_t pw = lb[sub + 4 + (base >> 3)];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int pp = base + q;
                    if (pp < kj) {
                        const double v = __longlong_as_double((long long)lb[sub + 4 + PW + pp]);
                        const int ps = (int)((pw >> (8 * q)) & 0xff);
                        gx += v * sx[ps];
                        gt += v * sth[ps];
                    }
                }
            }
            if (violated && lane == self) {
                cj *= P.factor;
                cmut[s] = cj;
            }
            const double a = cj + (gx - gmu) * th;
            const double b = cj / 100 + th * gt;
            const double L = LU[ndraw0 + rank];
            const double trefl = dev_poisson_time_L(a, b, L);
            const double tfreeze = (th * x >= 0) ? PDMP_INF : (-x / th);  // freezing_time, :10-16
            const bool fz = tfreeze <= trefl;                              // :57
            key = t + (fz ? tfreeze : trefl);
            rs->t_old = t;
            rs->a = a;
            rs->b = b;
            rs->acc = fz ? 1u : 0u;
            keys[s] = key;
        }
        nm += (uint64_t)ndraw0 + (uint64_t)__popcll(rball);
        if (lane < nst || lane == self) {
            rs->x = x;
            rs->th = th;
            rs->t = t;
            rs->I = I;
        }
        if (write_self_bound && lane == self) {
            rs->t_old = told_self;  // t_old[i] = t[i], :94
            rs->acc = 0;            // f[i] = false, :95
            keys[s] = key_self_extra;
        }
        if (is_thaw && lane == self && !rebound_set) rs->t_old = tp;  // :114 (only reachable if θf was 0)
        // ---------------- level 1 of the queue
        if (write_self_bound) queue_update(i, key_self_extra);
        for (int jj = 0; jj < k; ++jj) {
            if (!((rball >> jj) & 1ull)) continue;
            queue_update(readlane_u32(s, jj), readlane_f64(key, jj));
        }
        if (emit) {  // push!(Ξ, event(i, t, x, θ, F)), :154
            const double t_s = readlane_f64(t, self), x_s = readlane_f64(x, self), th_s2 = readlane_f64(th, self);
            if (ev && lane == 0) {
                pdmp_event e;
                e.t = t_s;
                e.i = (int64_t)i;
                e.x = x_s;
                e.theta = th_s2;
                ev[ntrace] = e;
            }
            ntrace += 1;
            nevents += 1;
            t_event = tp;
            if (!stop_before && !(tp < T)) running = false;
        }
        LDS_ORDER();
    }

    if (lane == 0) {
        hdr->c.t_last = t_last;
        hdr->t_event = t_event;
        hdr->c.num = num;
        hdr->c.nacc = nacc;
        hdr->c.ntrace = ntrace;
        hdr->c.nevents = nevents;
        hdr->c.ndraw_main = nm;
        hdr->c.status = status;
    }
}

// ------------------------------------------------------------------------------------------ speculative event loop
//
// Same chain semantics as zz_local_run_kernel, but up to E = 4 events of the chain are processed per iteration, one per
// 16-lane group (a DPP row), and committed only as far as they are PROVABLY what sequential processing would do:
//
//   select   the E smallest block minima of the queue (distinct blocks), in time order t_0 <= t_1 <= ...
//   execute  every event on the state as it is (loads, move, gradient, re-bound) WITHOUT storing anything
//   resolve  the accept chain in time order: event r's thinning coin is draw nm + off_r of the chain's stream, where
//            off_r counts the draws events 0..r-1 consume (2 on reject, 1 + k on accept) -- one wave-wide Philox call
//            produces all 64 candidate draws of the iteration
//   validate event r >= 1 is committed iff all earlier ones are, its two-hop zone S[i_r] is disjoint from theirs (so
//            nothing it read was written by them) and every key they produce or expose -- the re-reduced minimum of
//            their popped block and all their new keys -- is > t_r (so it really is the next event of the chain)
//   commit   the valid prefix: stores, level-1 updates, event records; the rest is discarded and re-selected.
//
// Zone-disjoint events with that key condition commute exactly, so the committed sequence (indices, accept/reject,
// times, positions, RNG draws) is bit-identical to the sequential kernel and to the oracle.  The gain: 4x the
// memory-level parallelism per wavefront and one instruction stream for 4 events.
// Requirements: |S[i]| <= 16, max column nnz <= 15, d + 1 <= 64 * 8 * 64; otherwise zz_local_run_kernel is used.

// LDS layout of the speculative kernel: every fixed-size array sits at a compile-time offset (folds into the DS
// instruction's immediate, no SGPR per array); the three size-dependent arrays come last.
constexpr uint32_t SP_U = 0;        // [64] f64 draws rng_base + lane
constexpr uint32_t SP_LU = 512;     // [64] f64 their logs
constexpr uint32_t SP_SX = 1024;    // [4][16] f64
constexpr uint32_t SP_STH = 1536;   // [4][16] f64
constexpr uint32_t SP_PK = 2048;    // [4][64] f64 patched key blocks
constexpr uint32_t SP_SLT = 4096;   // [4] f64 candidate keys
constexpr uint32_t SP_SLH = 4128;   // [4] f64 second-best entry of the offering lanes
constexpr uint32_t SP_LR = 4160;    // [4] f64 true rates
constexpr uint32_t SP_LBR = 4192;   // [4] f64 bounds
constexpr uint32_t SP_MR = 4224;    // [4] f64 what each event exposes (validation)
constexpr uint32_t SP_Z = 4256;     // [64] u32 zone ids
constexpr uint32_t SP_KR = 4512;    // [4] u32 k per event
constexpr uint32_t SP_SLB = 4528;   // [4] u32 candidate blocks
constexpr uint32_t SP_OFR = 4544;   // [8] u32 draw offsets after 0..4 events
constexpr uint32_t SP_LB = 4576;    // [4][Wpad] u64 blobs, then bk[nblk_pad] f64, bi[nblk_pad] u32

// WIDE (17 <= |S[i]| <= 32: two zone members per lane of a 16-lane row; the second one is always G2-only because |G1| <= 15): the same
// arrays with 32 slots per group where a slot is a zone member
// arrays with 32 slots per group where a slot is a zone member.  The patched key blocks take the place of sx / sth once the re-bound has read
// them (as in the 8-event kernel), so that a chain stays inside 10 KB: 16 chains per CU, all 4096 chains of the ensemble resident at once.
constexpr uint32_t SPW_SX = 1024;    // [4][32] f64
constexpr uint32_t SPW_STH = 2048;   // [4][32] f64
constexpr uint32_t SPW_PK = 1024;    // [4][64] f64 (over sx / sth)
constexpr uint32_t SPW_SLT = 3072, SPW_SLH = 3104, SPW_LR = 3136, SPW_LBR = 3168, SPW_MR = 3200;
constexpr uint32_t SPW_Z = 3232;     // [4][32] u32 zone ids
constexpr uint32_t SPW_KR = 3744, SPW_SLB = 3760, SPW_OFR = 3776;
constexpr uint32_t SPW_LB = 3808;

size_t zz_spec_lds_bytes(uint32_t nblk_pad, uint32_t blob_w_pad) {
    return (size_t)SP_LB + (size_t)4 * blob_w_pad * 8 + (size_t)nblk_pad * 8 + (size_t)nblk_pad * 4;
}
size_t zz_spec_wide_lds_bytes(uint32_t nblk_pad, uint32_t blob_w_pad) {
    return (size_t)SPW_LB + (size_t)4 * blob_w_pad * 8 + (size_t)nblk_pad * 8 + (size_t)nblk_pad * 4;
}

// minimum over the 16 lanes of a DPP row, returned in every lane of the row
__device__ __forceinline__ double row_min_f64(double v) {
    v = min_f64(v, dpp_f64<0xB1>(v));
    v = min_f64(v, dpp_f64<0x4E>(v));
    v = min_f64(v, dpp_f64<0x141>(v));
    v = min_f64(v, dpp_f64<0x140>(v));
    return v;
}


// ---- pieces shared by the two speculative kernels (always inlined: same code as written in place) ----

// Select up to E candidate events: lane l owns the level-1 entries l, l+64, ...; it offers its best entry and remembers its
// second best.  A lane offers only ONE entry per iteration, so the candidates are the E smallest entries only if no lane holds
// two of them -- the lane's second best therefore enters the validation bound of every later event, which keeps the commit rule
// exact.  Winners publish (key, second best, block) straight into the LDS slots.  Returns the number selected.
// FULLQ: the first level has exactly 4 * 64 entries (the launcher's promise for the 128 x 128 lattice): no bounds predicate, and
// (best, second best, argmin) of the lane's four entries come out of a 5-comparator network instead of a compare-select chain.
template <int NE, int E, bool FULLQ = false>
__device__ __forceinline__ int spec_select(const double* bk, uint32_t nblk, int lane, bool stop_before, double T, double* SLT,
                                           double* SLH, uint32_t* SLB, bool& first_inf) {
    double best = PDMP_INF, second = PDMP_INF;
    uint32_t bestb = 0;
    if constexpr (FULLQ && NE == 4) {
        const double k0 = bk[lane], k1 = bk[lane + 64], k2 = bk[lane + 128], k3 = bk[lane + 192];
        const double m01 = min_f64(k0, k1), x01 = max_f64(k0, k1), m23 = min_f64(k2, k3), x23 = max_f64(k2, k3);
        best = min_f64(m01, m23);
        second = min_f64(max_f64(m01, m23), min_f64(x01, x23));
        bestb = (uint32_t)lane + ((k0 == best) ? 0u : (k1 == best) ? 64u : (k2 == best) ? 128u : 192u);  // lowest block on ties
    } else {
#pragma unroll
        for (int q = 0; q < NE; ++q) {
            const uint32_t b = (uint32_t)lane + 64u * q;
            const double v = (b < nblk) ? bk[b] : PDMP_INF;
            const bool lt = v < best;
            second = min_f64(second, lt ? best : v);
            bestb = lt ? b : bestb;
            best = lt ? v : best;
        }
    }
    int Esel = 0;
    first_inf = false;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        if (Esel == r) {
            const double tpr = wave_min_f64(best);
            if (!(tpr < PDMP_INF)) {
                if (r == 0) first_inf = true;
            } else if (!(stop_before && !(tpr < T))) {
                const uint64_t ball = __ballot(best == tpr);
                const int wl = __ffsll((unsigned long long)ball) - 1;
                if (lane == wl) {
                    SLT[r] = best;
                    SLH[r] = second;
                    SLB[r] = bestb;
                    best = PDMP_INF;
                }
                Esel = r + 1;
            }
        }
    }
    return Esel;
}

// Zone conflicts with earlier groups (exact: compare member ids through LDS): does this lane's member id occur in the zone of
// a group q < g?  A cheap necessary condition runs first -- the id must lie inside the [min, max] id span of an earlier group's
// zone (two u32 row reductions, six scalar reads) -- and only if some lane of the wave passes it (about one iteration in three
// on the 128 x 128 lattice) are the 48 id comparisons made.
template <int CTRL>
__device__ __forceinline__ uint32_t zone_dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}
template <int E>
__device__ __forceinline__ bool spec_zone_conflict(const uint32_t* Z, uint32_t s, int g, bool member) {
    uint32_t lo = member ? s : 0xffffffffu, hi = member ? s : 0u;
    {
        uint32_t o;
        o = zone_dpp_u32<0xB1>(lo);   lo = (o < lo) ? o : lo;
        o = zone_dpp_u32<0x4E>(lo);   lo = (o < lo) ? o : lo;
        o = zone_dpp_u32<0x141>(lo);  lo = (o < lo) ? o : lo;
        o = zone_dpp_u32<0x140>(lo);  lo = (o < lo) ? o : lo;
        o = zone_dpp_u32<0xB1>(hi);   hi = (o > hi) ? o : hi;
        o = zone_dpp_u32<0x4E>(hi);   hi = (o > hi) ? o : hi;
        o = zone_dpp_u32<0x141>(hi);  hi = (o > hi) ? o : hi;
        o = zone_dpp_u32<0x140>(hi);  hi = (o > hi) ? o : hi;
    }
    bool maybe = false;
#pragma unroll
    for (int q = 0; q < E - 1; ++q) {
        const uint32_t lq = readlane_u32(lo, 16 * q), hq = readlane_u32(hi, 16 * q);
        maybe = maybe || ((q < g) && s >= lq && s <= hq);
    }
    maybe = maybe && member;
    if (__ballot(maybe) == 0) return false;
    const uint2* Z2 = reinterpret_cast<const uint2*>(Z);
    bool myconf = false;
#pragma unroll
    for (int q = 0; q < E - 1; ++q) {
        bool hit = false;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint2 zz = Z2[q * 8 + j];
            hit = hit || (zz.x == s) || (zz.y == s);
        }
        myconf = myconf || (hit && (q < g));
    }
    return myconf && member;
}

// WIDE: two ids per lane (s, s2), 32 per group: Z[q][32]
template <int E>
__device__ __forceinline__ bool spec_zone_conflict_wide(const uint32_t* Z, uint32_t s, uint32_t s2, int g, bool member, bool member2) {
    uint32_t lo = member ? s : 0xffffffffu, hi = member ? s : 0u;
    lo = (member2 && s2 < lo) ? s2 : lo;
    hi = (member2 && s2 > hi) ? s2 : hi;
    {
        uint32_t o;
        o = zone_dpp_u32<0xB1>(lo);   lo = (o < lo) ? o : lo;
        o = zone_dpp_u32<0x4E>(lo);   lo = (o < lo) ? o : lo;
        o = zone_dpp_u32<0x141>(lo);  lo = (o < lo) ? o : lo;
        o = zone_dpp_u32<0x140>(lo);  lo = (o < lo) ? o : lo;
        o = zone_dpp_u32<0xB1>(hi);   hi = (o > hi) ? o : hi;
        o = zone_dpp_u32<0x4E>(hi);   hi = (o > hi) ? o : hi;
        o = zone_dpp_u32<0x141>(hi);  hi = (o > hi) ? o : hi;
        o = zone_dpp_u32<0x140>(hi);  hi = (o > hi) ? o : hi;
    }
    bool maybe = false;
#pragma unroll
    for (int q = 0; q < E - 1; ++q) {
        const uint32_t lq = readlane_u32(lo, 16 * q), hq = readlane_u32(hi, 16 * q);
        maybe = maybe || ((q < g) && ((member && s >= lq && s <= hq) || (member2 && s2 >= lq && s2 <= hq)));
    }
    if (__ballot(maybe) == 0) return false;
    const uint4* Z4 = reinterpret_cast<const uint4*>(Z);
    bool myconf = false;
#pragma unroll
    for (int q = 0; q < E - 1; ++q) {
        bool hit = false;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint4 zz = Z4[q * 8 + j];
            hit = hit || (member && (zz.x == s || zz.y == s || zz.z == s || zz.w == s)) ||
                  (member2 && (zz.x == s2 || zz.y == s2 || zz.z == s2 || zz.w == s2));
        }
        myconf = myconf || (hit && (q < g));
    }
    return myconf;
}

// Minimum of the group's patched copy of the popped key block (4 keys per lane): row minimum, this lane's candidate.
__device__ __forceinline__ void spec_patched_min(const double* pk, int gl, uint32_t blk, double& rowmin, double& candmin,
                                                 uint32_t& cand) {
    const double2* pk2 = reinterpret_cast<const double2*>(pk + gl * 4);
    const double2 p01 = pk2[0], p23 = pk2[1];
    double lm = p01.x;
    uint32_t li = 0;
    if (p01.y < lm) {
        lm = p01.y;
        li = 1;
    }
    if (p23.x < lm) {
        lm = p23.x;
        li = 2;
    }
    if (p23.y < lm) {
        lm = p23.y;
        li = 3;
    }
    candmin = lm;
    cand = blk * 64u + (uint32_t)gl * 4u + li;
    rowmin = row_min_f64(lm);
}

// PLAIN: the configuration of the north-star workload -- adapt = false, target without a mean shift, G2 fetched on accept --
// as compile-time facts (the per-chain bound array, the Γμ look-up and the eager-G2 path drop out of the instantiation).
template <int NE, bool PROF, bool PLAIN, bool WIDE>
__device__ __forceinline__ void zz_local_spec_body(const ZzRunParams& P_in) {
    static_assert(!(PLAIN && WIDE), "PLAIN is the lattice's geometry");
    ZzRunParams P = P_in;
    if constexpr (PLAIN) {
        P.adapt = 0;
        P.c_chain = nullptr;
        P.tb.gmu_t = nullptr;
        P.flags |= 0x100;
        P.blob_w_pad = 58;  // 1 + SW + KMAX * (4 + PW + KMAX) words of the lattice's blob: LDS offsets become immediates
    }
    constexpr int E = 4;
    const int lane = threadIdx.x;
    const int g = lane >> 4;   // group = DPP row = event slot
    const int gl = lane & 15;  // lane inside the group
    const int64_t chain = blockIdx.x;
    const int64_t d = P.d;
    const uint32_t nblk = P.nblk;
    // PLAIN also fixes the blob geometry of the 4-neighbour lattice (|G1| <= 5, |S| <= 13): loop bounds and record strides become
    // immediates
    const uint32_t W2 = P.blob_w_pad >> 1, SW = PLAIN ? 7u : P.blob_sw, PW = PLAIN ? 1u : P.blob_pw, KMAX = PLAIN ? 5u : P.blob_kmax;
    const uint32_t R_ = 4 + PW + KMAX;

    extern __shared__ __align__(16) unsigned char smem[];
    constexpr uint32_t O_SX = WIDE ? SPW_SX : SP_SX, O_STH = WIDE ? SPW_STH : SP_STH, O_PK = WIDE ? SPW_PK : SP_PK, O_SLT = WIDE ? SPW_SLT : SP_SLT,
                       O_SLH = WIDE ? SPW_SLH : SP_SLH, O_LR = WIDE ? SPW_LR : SP_LR, O_LBR = WIDE ? SPW_LBR : SP_LBR, O_MR = WIDE ? SPW_MR : SP_MR,
                       O_Z = WIDE ? SPW_Z : SP_Z, O_KR = WIDE ? SPW_KR : SP_KR, O_SLB = WIDE ? SPW_SLB : SP_SLB, O_OFR = WIDE ? SPW_OFR : SP_OFR,
                       O_LB = WIDE ? SPW_LB : SP_LB, GSLOTS = WIDE ? 32u : 16u;
    double* const U = reinterpret_cast<double*>(smem + SP_U);
    double* const LU = reinterpret_cast<double*>(smem + SP_LU);
    double* const SLT = reinterpret_cast<double*>(smem + O_SLT);
    double* const SLH = reinterpret_cast<double*>(smem + O_SLH);
    double* const Lr = reinterpret_cast<double*>(smem + O_LR);
    double* const LBr = reinterpret_cast<double*>(smem + O_LBR);
    double* const Mr = reinterpret_cast<double*>(smem + O_MR);
    uint32_t* const Z = reinterpret_cast<uint32_t*>(smem + O_Z);
    uint32_t* const Kr = reinterpret_cast<uint32_t*>(smem + O_KR);
    uint32_t* const SLB = reinterpret_cast<uint32_t*>(smem + O_SLB);
    uint32_t* const OFR = reinterpret_cast<uint32_t*>(smem + O_OFR);
    double* const bk = reinterpret_cast<double*>(smem + O_LB + (size_t)4 * P.blob_w_pad * 8);
    uint32_t* const bi = reinterpret_cast<uint32_t*>(bk + P.nblk_pad);
    // per-group views
    double* const sx = reinterpret_cast<double*>(smem + O_SX) + g * GSLOTS;
    double* const sth = reinterpret_cast<double*>(smem + O_STH) + g * GSLOTS;
    double* const pk = reinterpret_cast<double*>(smem + O_PK) + g * 64;
    uint64_t* const lb = reinterpret_cast<uint64_t*>(smem + O_LB) + (size_t)g * P.blob_w_pad;

    ZzRec* rec = P.rec + chain * d;
    double* keys = P.keys + chain * P.dk;
    DevChain* hdr = P.hdr + chain;
    pdmp_event* ev = P.ev ? P.ev + chain * P.trace_cap : nullptr;
    double* cmut = P.c_chain ? (P.c_chain + chain * d) : nullptr;

    uint32_t status = hdr->c.status;
    if (status == PDMP_CHAIN_BOUND_VIOLATED || status == PDMP_CHAIN_STALLED) return;
    const uint64_t seed = hdr->seed;
    const uint64_t nm0 = hdr->c.ndraw_main, ntrace0 = hdr->c.ntrace;
    uint32_t dnm = 0, dnum = 0, dnacc = 0;  // 32-bit deltas of this launch (a launch advances a chain by far < 2^32 draws)
    uint32_t dnref = 0;                     // refresh events of this launch (recorded in the trace like reflections, src/sfact.jl:143)
    uint64_t ng = hdr->c.ndraw_global;      // the "global rng" stream of the refresh clock (:80,:84,:108)
    uint32_t vnacc = 0;                     // 1 if the launch ends on a bound violation (acc is bumped before the check)
    double t_last = hdr->c.t_last;
    double t_event = hdr->t_event;
    status = PDMP_CHAIN_OK;

    const double T = P.T;
    const bool stop_before = (P.flags & PDMP_RUN_STOP_BEFORE) != 0;
    const bool adapt = P.adapt != 0;
    const uint32_t trace_room = (P.trace_cap > 0)
                                    ? (uint32_t)(((uint64_t)P.trace_cap > ntrace0) ? ((uint64_t)P.trace_cap - ntrace0) : 0)
                                    : 0xffffffffu;  // events this launch may still record

    for (uint32_t b = lane; b < nblk; b += 64) {
        const double* kp = keys + (size_t)b * 64;
        double mk = kp[0];
        uint32_t mi = 0;
#pragma unroll 8
        for (int q = 1; q < 64; ++q) {
            const double v = kp[q];
            if (v < mk) {
                mk = v;
                mi = q;
            }
        }
        bk[b] = mk;
        bi[b] = b * 64 + mi;
    }
    LDS_ORDER();

    uint32_t rng_base = 0xffffffffu;  // first draw (as a delta to nm0) held in U/LU; none yet
    uint64_t ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t ph_t0 = PROF ? (uint64_t)__builtin_readcyclecounter() : 0;
    uint64_t ph_iters = 0;
#define PHASE(k)                                                          \
    do {                                                                  \
        if (PROF) {                                                       \
            const uint64_t now_ = (uint64_t)__builtin_readcyclecounter(); \
            ph[k] += now_ - ph_t0;                                        \
            ph_t0 = now_;                                                 \
        }                                                                 \
    } while (0)

    bool running = stop_before || (t_event < T);
    PrioTurn prio;
    while (running) {
        prio.step();
        if (dnacc + dnref >= trace_room) {
            status = PDMP_CHAIN_TRACE_FULL;
            break;
        }
        if (dnm >= P.count_limit) {  // (32-bit counters of the launch: pause, the host runs again)
            status = PDMP_CHAIN_PAUSED;
            break;
        }
        // ---------------- select up to E candidate events (spec_select)
        bool first_inf;
        int Esel = spec_select<NE, E, PLAIN && NE == 4>(bk, nblk, lane, stop_before, T, SLT, SLH, SLB, first_inf);
        if (Esel == 0) {
            if (first_inf) status = PDMP_CHAIN_STALLED;
            break;
        }
        LDS_ORDER();
        if (P.has_refresh) {
            // ---------------- the refresh clock (key d, src/sfact.jl:78-114) among the candidates: the events before it go through the speculative
            // iteration as usual; once it is the chain's NEXT event it is processed by itself, the whole wave on one event, exactly as
            // zz_local_run_kernel does (a refresh moves the neighbourhood of one random coordinate and re-bounds that of another: no zone
            // test covers it, and at rate λref against thousands of proposals per unit time it need not be fast)
            const uint64_t rfb = __ballot(g < Esel && gl == 0 && bi[SLB[g]] == (uint32_t)d);
            if (rfb) {
                const int rg = (__ffsll((unsigned long long)rfb) - 1) >> 4;
                if (rg > 0) {
                    Esel = rg;
                } else {
                    const double tp = uniform_f64(SLT[0]);
                    uint64_t* const lb0 = reinterpret_cast<uint64_t*>(smem + O_LB);
                    double* const sx0 = reinterpret_cast<double*>(smem + O_SX);
                    double* const sth0 = reinterpret_cast<double*>(smem + O_STH);
                    const uint64_t nm = nm0 + (uint64_t)dnm;
                    t_last = tp;
                    const uint32_t i1 = pdmp_randint(seed, PDMP_STREAM_GLOBAL, ng, (uint32_t)d);  // :80
                    ng += 1;
                    {
                        const uint64_t* bsrc = P.blob + (size_t)P.tix[i1] * P.blob_w_pad;
                        for (uint32_t w = lane; w < P.blob_w; w += 64) lb0[w] = bsrc[w];
                        LDS_ORDER();
                        const int k1 = (int)uniform_u32((uint32_t)(lb0[0] & 0xff));
                        if (lane < k1) {  // smove_forward!(G, i1, ...), :82
                            const uint64_t sw = lb0[1 + (lane >> 1)];
                            const uint32_t s1 = i1 + ((lane & 1) ? (uint32_t)(sw >> 32) : (uint32_t)sw);
                            ZzRec* r1 = rec + s1;
                            const double x0 = r1->x, th0 = r1->th, t0 = r1->t, I0 = r1->I;
                            const double dt = tp - t0;
                            const double xn = x0 + th0 * dt;
                            r1->x = xn;
                            r1->t = tp;
                            r1->I = I0 + dt * ((x0 + xn) * 0.5);
                        }
                        LDS_ORDER();
                    }
                    const uint32_t i2 = pdmp_randint(seed, PDMP_STREAM_GLOBAL, ng, (uint32_t)d);  // :84
                    ng += 1;
                    {
                        const uint64_t* bsrc = P.blob + (size_t)P.tix[i2] * P.blob_w_pad;
                        for (uint32_t w = lane; w < P.blob_w; w += 64) lb0[w] = bsrc[w];
                    }
                    LDS_ORDER();
                    const uint64_t hw = lb0[0];
                    const int k = (int)uniform_u32((uint32_t)(hw & 0xff));
                    const int m = (int)uniform_u32((uint32_t)((hw >> 8) & 0xff));
                    const int self = (int)uniform_u32((uint32_t)((hw >> 16) & 0xff));
                    const int kjmax = (int)uniform_u32((uint32_t)((hw >> 24) & 0xff));
                    uint32_t s = i2;
                    if (lane < m) {
                        const uint64_t sw = lb0[1 + (lane >> 1)];
                        s = i2 + ((lane & 1) ? (uint32_t)(sw >> 32) : (uint32_t)sw);
                    }
                    ZzRec* rs = rec + s;
                    double x = 0.0, th = 0.0, t = 0.0, I = 0.0;
                    if (lane < m) {
                        x = rs->x;
                        th = rs->th;
                        t = rs->t;
                        I = rs->I;
                    }
                    if (lane >= k && lane < m) {  // smove_forward!(G2, i, ...), :85
                        const double dt = tp - t;
                        const double xn = x + th * dt;
                        I = I + dt * ((x + xn) * 0.5);
                        x = xn;
                        t = tp;
                    }
                    const double usign = pdmp_u01(seed, PDMP_STREAM_MAIN, nm);  // θ[i] = σ[i]*rand(rng, (-1,1)), :100-101
                    if (lane == self) th = P.tb.sigma[i2] * ((usign < 0.5) ? -1.0 : 1.0);
                    // Q[n+1] = t′ + waiting_time_ref(F) = t′ + randexp()/λref from the global rng, :108
                    const double newref = tp + (-pdmp_log(pdmp_u01(seed, PDMP_STREAM_GLOBAL, ng))) / P.lambda_ref;
                    ng += 1;
                    if (lane < m) {
                        sx0[lane] = x;
                        sth0[lane] = th;
                    }
                    LDS_ORDER();
                    const uint32_t sub0 = 1 + SW + (uint32_t)lane * R_;
                    double key = PDMP_INF;
                    if (lane < k) {  // :110-114 (at the coordinates' OWN, possibly stale, clocks: the reference's behaviour)
                        const double gmu = __longlong_as_double((long long)lb0[sub0 + 1]);
                        const double cj = cmut ? cmut[s] : __longlong_as_double((long long)lb0[sub0 + 2]);
                        const int kj = (int)(lb0[sub0 + 3] & 0xff);
                        double gx = 0.0, gt = 0.0;
                        for (int base = 0; base < kjmax; base += 8) {
                            const uint64_t pw = lb0[sub0 + 4 + (base >> 3)];
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                const int pp = base + q;
                                if (pp < kj) {
                                    const double v = __longlong_as_double((long long)lb0[sub0 + 4 + PW + pp]);
                                    const int ps = (int)((pw >> (8 * q)) & 0xff);
                                    gx += v * sx0[ps];
                                    gt += v * sth0[ps];
                                }
                            }
                        }
                        const double a = cj + (gx - gmu) * th;
                        const double b = cj / 100 + th * gt;
                        const double L = pdmp_log(pdmp_u01(seed, PDMP_STREAM_MAIN, nm + 1 + (uint64_t)lane));
                        key = t + dev_poisson_time_L(a, b, L);
                        rs->t_old = t;
                        rs->a = a;
                        rs->b = b;
                        keys[s] = key;
                    }
                    dnm += 1u + (uint32_t)k;
                    if (lane < m) {
                        rs->x = x;
                        rs->th = th;
                        rs->t = t;
                        rs->I = I;
                    }
                    if (lane == 0) keys[d] = newref;
                    for (int jj = 0; jj <= k; ++jj) {
                        const uint32_t j = (jj < k) ? readlane_u32(s, jj) : (uint32_t)d;
                        const double kjv = (jj < k) ? readlane_f64(key, jj < k ? jj : 0) : newref;
                        level1_update(bk, bi, keys, lane, j, kjv);
                        LDS_ORDER();
                    }
                    const double t_i = readlane_f64(t, self), x_i = readlane_f64(x, self), th_i2 = readlane_f64(th, self);
                    if (ev && lane == 0) {  // event(i, t, x, θ, F) = (t[i], i, x[i], θ[i]), :143
                        pdmp_event e;
                        e.t = t_i;
                        e.i = (int64_t)i2;
                        e.x = x_i;
                        e.theta = th_i2;
                        ev[ntrace0 + dnacc + dnref] = e;
                    }
                    dnref += 1;
                    t_event = tp;
                    if (!stop_before && !(tp < T)) running = false;
                    LDS_ORDER();
                    continue;
                }
            }
        }
        PHASE(0);
        if (PROF) ph_iters += 1;
        const bool gvalid = g < Esel;
        const double tp = gvalid ? SLT[g] : PDMP_INF;
        const uint32_t blk = gvalid ? SLB[g] : 0u;
        const double hidg = gvalid ? SLH[g] : PDMP_INF;
        const uint32_t i = gvalid ? bi[blk] : 0u;

        // ---------------- level-1 loads (functions of i alone), per group
        {
            const uint32_t tixi = gvalid ? P.tix[i] : 0u;
            const ulonglong2* bsrc = reinterpret_cast<const ulonglong2*>(P.blob + (size_t)tixi * P.blob_w_pad);
            ulonglong2* bdst = reinterpret_cast<ulonglong2*>(lb);
            if (gvalid) {
                for (uint32_t w = gl; w < W2; w += 16) bdst[w] = bsrc[w];
            }
        }
        // ---------------- candidate draws: LDS holds draws rng_base .. rng_base+63 of the chain's stream and their logs.
        // An iteration consumes at most E*(1+KMAX) <= 64 of them (about 10 on C3), so one wave-wide Philox + log call
        // serves several iterations; the window is refilled only when the worst case would run past its end.
        if (dnm < rng_base || dnm + E * (1u + KMAX) > rng_base + 64u) {
            rng_base = dnm;
            const double u = pdmp_u01(seed, PDMP_STREAM_MAIN, nm0 + (uint64_t)dnm + (uint64_t)lane);
            U[lane] = u;
            LU[lane] = pdmp_log(u);
        }
        const uint32_t rng_off = dnm - rng_base;
        LDS_ORDER();
        PHASE(1);
        // ---------------- neighbourhood header and member list
        int k = 0, m = 0, self = 0, kjmax = 0;
        uint32_t s = 0xffffff00u + (uint32_t)lane;
        uint32_t s2 = 0xffffff40u + (uint32_t)lane;  // (WIDE: zone position gl + 16)
        if (gvalid) {
            const uint64_t hw = lb[0];
            k = (int)(hw & 0xff);
            m = (int)((hw >> 8) & 0xff);
            self = (int)((hw >> 16) & 0xff);
            kjmax = (int)((hw >> 24) & 0xff);
            if (gl < m) {
                const uint64_t sw = lb[1 + (gl >> 1)];
                s = i + ((gl & 1) ? (uint32_t)(sw >> 32) : (uint32_t)sw);
            }
            if (WIDE && gl + 16 < m) {
                const uint64_t sw = lb[1 + 8 + (gl >> 1)];
                s2 = i + ((gl & 1) ? (uint32_t)(sw >> 32) : (uint32_t)sw);
            }
        }
        const bool member = gvalid && gl < m;
        const bool member2 = WIDE && gvalid && gl + 16 < m;
        PHASE(2);
        ZzRec* rs = rec + (member ? s : i);
        ZzRec* rs2 = rec + (member2 ? s2 : i);
        double x = 0.0, th = 0.0, t = 0.0, I = 0.0;
        double x2 = 0.0, th2 = 0.0, t2 = 0.0, I2 = 0.0;
        const bool lazy_g2 = (P.flags & 0x100) != 0;  // fetch G2[i] only once the event is accepted (default)
        if (member && (!lazy_g2 || gl < k)) {
            x = rs->x;
            th = rs->th;
            t = rs->t;
            I = rs->I;
        }
        if (WIDE && member2 && !lazy_g2) {
            x2 = rs2->x;
            th2 = rs2->th;
            t2 = rs2->t;
            I2 = rs2->I;
        }
        // All HBM-latency loads of the iteration are issued HERE, in one batch, behind the compiler barrier of the blob
        // hand-off: vmcnt retires in order, so an HBM load issued before the (L2-hot) blob load would make the blob wait a
        // full HBM latency -- two serialized HBM round trips per iteration instead of one.
        const ZzRec* ri = rec + i;
        double told_i = 0.0, a_i = 0.0, b_i = 0.0;
        uint64_t acc_i = 0;
        double kq[4] = {PDMP_INF, PDMP_INF, PDMP_INF, PDMP_INF};
        if (gvalid) {
            told_i = ri->t_old;
            a_i = ri->a;
            b_i = ri->b;
            acc_i = ri->acc;
            const double2* kp = reinterpret_cast<const double2*>(keys + (size_t)blk * 64 + gl * 4);
            const double2 k01 = kp[0], k23 = kp[1];
            kq[0] = k01.x;
            kq[1] = k01.y;
            kq[2] = k23.x;
            kq[3] = k23.y;
        }
        if (WIDE) {
            Z[g * 32 + gl] = s;
            Z[g * 32 + 16 + gl] = s2;
        } else {
            Z[lane] = s;
        }
        const uint32_t sub = 1 + SW + (uint32_t)gl * R_;
        double cj = 0.0;
        if (gvalid && gl < k) cj = cmut ? cmut[s] : __longlong_as_double((long long)lb[sub + 2]);

        // ---------------- zone conflicts with earlier groups (exact: compare member ids)
        LDS_ORDER();
        const uint64_t confball = WIDE ? __ballot(spec_zone_conflict_wide<E>(Z, s, s2, g, member, member2))
                                       : __ballot(spec_zone_conflict<E>(Z, s, g, member));
        PHASE(3);

        // ---------------- smove_forward!(G, i, ...), gradient, rates
        if (gvalid && gl < k) {
            const double dt = tp - t;
            const double xn = x + th * dt;
            I = I + dt * ((x + xn) * 0.5);
            x = xn;
            t = tp;
        }
        if (member) {
            sx[gl] = x;
            sth[gl] = th;
        }
        LDS_ORDER();
        {
            double gr = 0.0;
            for (uint32_t p = 0; p < KMAX; ++p) {
                if ((int)p < k) gr += __longlong_as_double((long long)lb[1 + SW + p * R_]) * sx[p];
            }
            if (gvalid) {
                if (P.tb.gmu_t) gr = gr - P.tb.gmu_t[i];
                const double th_i = sth[self];
                const double l = pos_part(gr * th_i);
                const double lbound = pos_part(a_i + b_i * (tp - told_i));
                if (gl == 0) {
                    Lr[g] = l;
                    LBr[g] = lbound;
                    Kr[g] = (uint32_t)k;
                }
            }
        }
        LDS_ORDER();
        // ---------------- accept chain in time order: every lane walks it (per-lane arithmetic on LDS broadcasts, no
        // scalar registers), keeping only its own group's outcome
        uint32_t accept_u = 0, violated_u = 0, myoff = 0;
        {
            uint32_t off = 0;
#pragma unroll
            for (int r = 0; r < E; ++r) {
                if (g == r) myoff = off;
                if (lane == 0) OFR[r] = off;
                if (r < Esel) {
                    const double coin = U[rng_off + off];
                    const double l = Lr[r], lbound = LBr[r];
                    const uint32_t a_r = (coin * lbound < l) ? 1u : 0u;        // :121
                    const uint32_t v_r = (a_r && (l >= lbound)) ? 1u : 0u;     // :123
                    off += a_r ? (1u + Kr[r]) : 2u;
                    if (g == r) {
                        accept_u = a_r;
                        violated_u = v_r;
                    }
                }
            }
            if (lane == 0) OFR[E] = off;
        }
        const bool accept = accept_u != 0;
        const bool violated = violated_u != 0;
        PHASE(4);

        int nmoved = k;
        if (gvalid && accept) {
            if (gl >= k && gl < m) {  // smove_forward!(G2, i, ...), :129
                if (lazy_g2) {
                    x = rs->x;
                    th = rs->th;
                    t = rs->t;
                    I = rs->I;
                }
                const double dt = tp - t;
                const double xn = x + th * dt;
                I = I + dt * ((x + xn) * 0.5);
                x = xn;
                t = tp;
            }
            if (WIDE && member2) {
                if (lazy_g2) {
                    x2 = rs2->x;
                    th2 = rs2->th;
                    t2 = rs2->t;
                    I2 = rs2->I;
                }
                const double dt = tp - t2;
                const double xn = x2 + th2 * dt;
                I2 = I2 + dt * ((x2 + xn) * 0.5);
                x2 = xn;
                t2 = tp;
                sx[16 + gl] = x2;
                sth[16 + gl] = th2;
            }
            nmoved = m;
            if (gl == self) th = -th;  // reflect!, :130
        }
        if (gvalid && gl < nmoved) {
            sx[gl] = x;
            sth[gl] = th;
        }
        if (!WIDE) {
            double2* pk2 = reinterpret_cast<double2*>(pk + gl * 4);
            pk2[0] = make_double2(kq[0], kq[1]);
            pk2[1] = make_double2(kq[2], kq[3]);
        }
        LDS_ORDER();
        // ---------------- re-bound (ab + poisson_time) -- results stay in registers until the commit
        const bool active = gvalid && (accept ? (gl < k) : (gl == self));
        double key = PDMP_INF, a = 0.0, b = 0.0;
        if (active) {
            const double gmu = __longlong_as_double((long long)lb[sub + 1]);
            const int kj = (int)(lb[sub + 3] & 0xff);
            double gx = 0.0, gt = 0.0;
            for (int base = 0; base < (PLAIN ? 1 : kjmax); base += 8) {
                const uint64_t pw = lb[sub + 4 + (base >> 3)];
#pragma unroll
                for (int q = 0; q < (PLAIN ? 5 : 8); ++q) {
                    const int pp = base + q;
                    if (pp < kj) {
                        const double v = __longlong_as_double((long long)lb[sub + 4 + PW + pp]);
                        const int ps = (int)((pw >> (8 * q)) & 0xff);
                        gx += v * sx[ps];
                        gt += v * sth[ps];
                    }
                }
            }
            if (violated && gl == self) cj *= P.factor;  // adapt!(c, i, factor), :127 (stored at commit)
            a = cj + (gx - gmu) * th;
            b = cj / 100 + th * gt;
            const double L = LU[rng_off + myoff + 1 + (accept ? (uint32_t)gl : 0u)];
            key = t + dev_poisson_time_L(a, b, L);
            if (!WIDE && (s >> 6) == blk) pk[s & 63] = key;
        }
        LDS_ORDER();
        if (WIDE) {  // the patched key blocks go where sx / sth were: all their readers are done
            double2* pk2 = reinterpret_cast<double2*>(pk + gl * 4);
            pk2[0] = make_double2(kq[0], kq[1]);
            pk2[1] = make_double2(kq[2], kq[3]);
            LDS_ORDER();
            if (active && (s >> 6) == blk) pk[s & 63] = key;
            LDS_ORDER();
        }
        PHASE(5);
        // ---------------- patched minimum of the popped block, and everything this event could expose
        double rowmin, candmin;
        uint32_t cand;
        spec_patched_min(pk, gl, blk, rowmin, candmin, cand);
        const uint64_t winball = __ballot(gvalid && candmin == rowmin);
        const int wl2 = __ffs((unsigned)((winball >> (16 * g)) & 0xffffu)) - 1;
        const double keymin = row_min_f64(key);
        const double expose = min_f64(min_f64(rowmin, keymin), hidg);
        if (gl == 0) Mr[g] = expose;
        LDS_ORDER();
        // ---------------- validate: event g commits iff all earlier ones do, its zone is disjoint from theirs, and nothing
        // they produce or expose comes before it.  Per-lane evaluation + ballots; the prefix is resolved on the scalar unit.
        uint32_t Rc;
        uint32_t nacc_c;
        int vsel = -1;  // the event that violates its bound, if it is the chain's next one
        {
            const double m0 = Mr[0], m1 = Mr[1], m2 = Mr[2];
            const double pref = (g == 0) ? PDMP_INF : (g == 1) ? m0 : (g == 2) ? min_f64(m0, m1) : min_f64(min_f64(m0, m1), m2);
            const bool confg = ((confball >> (16 * g)) & 0xffffull) != 0;
            const bool okg = gvalid && ((g == 0) || (!confg && pref > tp));
            const bool vstop = violated && !adapt;  // reference: error(...), :124 -> the event is not committed
            const uint64_t okball = __ballot(okg && !vstop && gl == 0);
            const uint64_t vball = __ballot(okg && vstop && gl == 0);
            const uint64_t accball = __ballot(gvalid && accept && gl == 0);
            // compact one bit per group
            auto bits4 = [](uint64_t m_) -> uint32_t {
                return (uint32_t)((m_ & 1ull) | ((m_ >> 15) & 2ull) | ((m_ >> 30) & 4ull) | ((m_ >> 45) & 8ull));
            };
            const uint32_t okb = bits4(okball), vb = bits4(vball), accb = bits4(accball);
            const uint32_t gap = ~okb & 0xfu;  // the run of committable slots from slot 0 ends at the first zero bit
            const uint32_t r_ok = gap ? (uint32_t)(__ffs((int)gap) - 1) : (uint32_t)E;
            // stop AFTER an accepted event that fills the trace or passes T (`while t′ < T`)
            Rc = 0;
            nacc_c = 0;
            bool stopped = false;
            // the usual case needs no walk: the slice mode stops on time alone, and the trace has room for every accepted slot
            const uint32_t acc_run = accb & ((1u << r_ok) - 1u);
            const bool plainrun = stop_before && !(P.trace_cap > 0 && dnacc + dnref + (uint32_t)__popc(acc_run) >= trace_room);
            if (plainrun) {
                Rc = r_ok;
                nacc_c = (uint32_t)__popc(acc_run);
            }
            for (uint32_t r = 0; !plainrun && r < r_ok && !stopped; ++r) {
                Rc = r + 1;
                if ((accb >> r) & 1u) {
                    nacc_c += 1;
                    if (dnacc + dnref + nacc_c >= trace_room && P.trace_cap > 0) {
                        status = PDMP_CHAIN_TRACE_FULL;
                        stopped = true;
                    }
                    if (!stop_before && !(uniform_f64(SLT[r]) < T)) {
                        running = false;
                        stopped = true;
                    }
                }
            }
            // the first event that does not commit violates its bound (and nothing stopped the chain before it)
            if (!stopped && r_ok < (uint32_t)E && ((vb >> r_ok) & 1u)) {
                status = PDMP_CHAIN_BOUND_VIOLATED;
                vsel = (int)r_ok;
            }
        }
        PHASE(6);

        // ---------------- commit the valid prefix
        const bool commit = gvalid && (uint32_t)g < Rc;
        const uint64_t accball2 = __ballot(commit && accept && gl == 0);
        if (commit) {
            if (gl < nmoved) {
                rs->x = x;
                rs->th = th;
                rs->t = t;
                rs->I = I;
            }
            if (WIDE && accept && member2) {
                rs2->x = x2;
                rs2->th = th2;
                rs2->t = t2;
                rs2->I = I2;
            }
            if (active) {
                rs->t_old = t;
                rs->a = a;
                rs->b = b;
                keys[s] = key;
                if (violated && gl == self) cmut[s] = cj;
            }
            if (accept && gl == self) rs->acc = acc_i + 1;
            if (gl == wl2) {
                bk[blk] = rowmin;
                bi[blk] = cand;
            }
            if (accept && gl == self && ev) {
                const uint32_t rank = (uint32_t)__popcll(accball2 & ((1ull << (16 * g)) - 1ull));
                pdmp_event e;
                e.t = tp;
                e.i = (int64_t)i;
                e.x = x;
                e.theta = th;
                ev[ntrace0 + dnacc + dnref + rank] = e;
            }
        }
        LDS_ORDER();
        PHASE(7);
        // ---------------- level-1 updates for re-bounded neighbours living in other blocks.  A block's final entry is the
        // smallest (key, coordinate) among its old entry and the new keys, whatever the order: when no two of these lanes aim at
        // one block (claims through the now idle zone-id array) and none has to rescan, every lane updates its block by itself in
        // one LDS round trip; otherwise one by one in event order.
        const bool upd = commit && accept && gl < k && (s >> 6) != blk;
        if (__ballot(upd) != 0) {
            LDS_ORDER();
            const uint32_t bjv = upd ? (s >> 6) : 0u;
            const double curv = bk[bjv];
            const uint32_t civ = bi[bjv];
            const bool lower = upd && (key < curv || (key == curv && s < civ));
            const bool resc = upd && !lower && civ == s;
            if (lower) Z[bjv & 63u] = (uint32_t)lane;
            LDS_ORDER();
            const bool lost = lower && Z[bjv & 63u] != (uint32_t)lane;
            if (__ballot(lost || resc) == 0) {
                if (lower) {
                    bk[bjv] = key;
                    bi[bjv] = s;
                }
            } else {
            for (uint32_t r = 0; r < Rc; ++r) {
                if (!((accball2 >> (16 * r)) & 1ull)) continue;
                const uint32_t own = uniform_u32(SLB[r]);
                const int kr = (int)uniform_u32(Kr[r]);
                for (int jj = 0; jj < kr; ++jj) {
                    const uint32_t j = readlane_u32(s, 16 * (int)r + jj);
                    if ((j >> 6) == own) continue;
                    level1_update(bk, bi, keys, lane, j, readlane_f64(key, 16 * (int)r + jj));
                }
            }
            }
        }
        PHASE(8);
        // ---------------- the violating proposal itself (reference: counted, G[i] moved, acc bumped -- then error(...), :120-124):
        // what zz_local_run_kernel and the oracle leave behind
        if (vsel >= 0) {
            if (g == vsel && gl < k) {
                rs->x = x;
                rs->t = t;
                rs->I = I;
            }
            dnum += 1;
            vnacc = 1;
            dnm += uniform_u32(OFR[vsel]) + 1u - (Rc > 0 ? uniform_u32(OFR[Rc]) : 0u);
        }
        // ---------------- counters
        if (Rc > 0) {
            dnum += Rc;
            dnacc += nacc_c;
            dnm += uniform_u32(OFR[Rc]);
            t_last = uniform_f64(SLT[Rc - 1]);
            const uint32_t accc = (uint32_t)(((accball2 & 1ull)) | ((accball2 >> 15) & 2ull) | ((accball2 >> 30) & 4ull) |
                                             ((accball2 >> 45) & 8ull));
            if (accc) t_event = uniform_f64(SLT[31 - __builtin_clz(accc)]);
        }
        if (vsel >= 0) t_last = uniform_f64(SLT[vsel]);  // the violating event's time is the chain's current time
        if (status != PDMP_CHAIN_OK) break;
        LDS_ORDER();
    }

    if (PROF && P.dbg && chain == 0 && lane == 0) {
        for (int q = 0; q < 10; ++q) P.dbg[q] = (double)ph[q];
        P.dbg[10] = (double)ph_iters;
    }
#undef PHASE
    if (lane == 0) {
        hdr->c.t_last = t_last;
        hdr->t_event = t_event;
        hdr->c.num += dnum;
        hdr->c.nacc += dnacc + vnacc;
        hdr->c.ntrace = ntrace0 + dnacc + dnref;
        hdr->c.nevents += dnacc + dnref;
        hdr->c.nrefresh += dnref;
        hdr->c.ndraw_global = ng;
        hdr->c.ndraw_main = nm0 + dnm;
        hdr->c.status = status;
    }
}

template <int NE, bool PROF, bool PLAIN = false>
__global__ __launch_bounds__(64) void zz_local_spec_kernel(ZzRunParams P_in) {
    zz_local_spec_body<NE, PROF, PLAIN, false>(P_in);
}
// 17 <= |S[i]| <= 32 (two zone members per lane): held to 128 registers, i.e. 4 waves per SIMD with the 16 chains per CU the LDS allows
template <int NE, bool PROF>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void zz_local_spec_wide_kernel(ZzRunParams P_in) {
    zz_local_spec_body<NE, PROF, false, true>(P_in);
}

// ------------------------------------------------------------------------------------------ 8 events per iteration
//
// zz_local_spec_kernel's scheme with EIGHT event slots per iteration, one per 8-lane group, for the north-star workload (the
// PLAIN configuration on the 128 x 128 lattice: |G1| <= 5 <= 8 lanes, |S| <= 13 <= 16 = two zone members per lane).  The
// instruction stream of an iteration -- selection, loads, accept chain, validation, commit -- is issued once for eight events
// instead of four.  The kernel sits at the memory system's random-sector rate and at the instruction issue rate of 4 waves per
// SIMD at the same time, so both the sectors and the instructions per event count.  What changes:
//   queue     the first level has 512 entries over key blocks of 32: a popped block is four 64-byte sectors, not eight
//   select    one wave minimum m, then every first-level entry <= m + sel_dt is a candidate (compares + population counts);
//             the candidates (<= 16, else sel_dt is halved) are compacted into LDS by ballot prefix counts and rank
//             themselves against each other; ranks 0..7 become the slots.  The slots hold exactly the smallest entries in
//             time order whatever sel_dt is, so there is no hidden second-best to carry into the validation bound
//   templates slot 0 of the LDS blob area holds the lattice's common template for the whole launch; the (border) events of an
//             iteration that need another one share two spare slots, a third such event ends the iteration's candidate list
//   members   lane gl of a group owns zone positions gl and gl + 8; positions >= 8 are always G2-only (k <= 5), so their
//             records are touched on accept only; all HBM loads of an iteration are one straight-line batch
//   accept    every lane evaluates the thinning test of every event for the draw offset equal to its lane number; the ballots
//             are walked on the scalar unit (offset of event r+1 = offset of r + 2 or 1 + k_r), no dependent LDS round trips
//   LDS       10 136 bytes per chain = 16 chains per CU: the selection scratch, then zone ids + sx/sth, then the patched key
//             blocks take turns in one 2816-byte area; pitches and piece order are chosen against bank conflicts
// Validation and commit rules are unchanged, so the committed sequence is bit-identical to the other kernels and the oracle.
constexpr uint32_t S8_LU = 0;       // [64] f64 logs of the draw window (the draws themselves stay in a register per lane)
constexpr uint32_t S8_R = 512;      // 2816 bytes used in turn by: TK/TB (selection), zone ids + sx/sth, the patched key blocks
constexpr uint32_t S8_SXP = 18;     // doubles per group in sx / sth (16 + 2: eight groups on eight different banks)
constexpr uint32_t S8_SX = S8_R;            // [8][18] f64
constexpr uint32_t S8_STH = S8_R + 1152;    // [8][18] f64
constexpr uint32_t S8_Z = S8_R + 2304;      // [8][16] u32 zone ids
constexpr uint32_t S8_PK = S8_R;            // [8][32] f64 patched key blocks (sx / sth / zone ids are dead by then)
constexpr uint32_t SEL_CAP = 16;            // candidates ranked per iteration (power of two)
constexpr uint32_t S8_TK = S8_R;            // [64] f64 candidate keys (selection only; the first SEL_CAP are ranked)
constexpr uint32_t S8_TB = S8_R + 64 * 8;       // [64] u32 their blocks (the first SEL_CAP are ranked)
constexpr uint32_t S8_SLT = 3328;   // [8] f64 candidate keys
constexpr uint32_t S8_LR = 3392;    // [8] f64 true rates
constexpr uint32_t S8_LBR = 3456;   // [8] f64 bounds
constexpr uint32_t S8_MR = 3520;    // [8] f64 what each event exposes
constexpr uint32_t S8_SLB = 3584;   // [8] u32 candidate blocks
constexpr uint32_t S8_LB = 3616;    // [3][58] u64 blob slots
constexpr uint32_t S8_NBLK = 512;   // first-level entries: key blocks of 32 (four 64-byte sectors per popped block)
constexpr uint32_t S8_BK = S8_LB + 3 * 58 * 8;      // [512] f64 block minima
constexpr uint32_t S8_BI = S8_BK + S8_NBLK * 8;     // [512] u16 their coordinates (d = 16384)
constexpr uint32_t S8_SELDT = S8_BI + S8_NBLK * 2;  // f64 selection threshold above the minimum
constexpr uint32_t S8_CL = S8_SELDT + 8;            // [64] u8 claims of the parallel first-level update
constexpr uint32_t S8_BYTES = S8_CL + 64;           // 10200
constexpr uint32_t S8_PR = S8_SLT;              // [16][4] u32 partial ranks (aliases SLT .. MR, which are written after the ranking)

static_assert(S8_BYTES <= 10240, "the 8-event kernel needs 16 workgroups per CU: 160 KB / 16");
static_assert(S8_R + 2816 <= S8_SLT && S8_Z + 8 * 16 * 4 <= S8_SLT && S8_PK + 8 * 32 * 8 <= S8_SLT, "shared scratch area overflows");
static_assert(S8_TB + 64 * 4 <= S8_Z, "selection scratch must not reach the zone ids");
static_assert(S8_SLB + 8 * 4 <= S8_LB && (S8_LB % 16) == 0 && (S8_BK % 16) == 0 && (S8_STH % 16) == 0 && (S8_Z % 16) == 0,
              "LDS sub-arrays must stay 16-byte aligned");
size_t zz_spec8_lds_bytes() { return S8_BYTES; }

// the largest double below a finite x (x > 0, or x < 0, or x == 0 all handled by the integer image)
__device__ __forceinline__ double pdmp_below(double x) {
    long long b = __double_as_longlong(x);
    if (x > 0) b -= 1;
    else if (x < 0) b += 1;
    else b = (long long)0x8000000000000001ull;  // -denorm_min
    return __longlong_as_double(b);
}
// value of lane `src` (any lane, per-lane choice): two ds_bpermute_b32
__device__ __forceinline__ double bperm_f64(double v, uint32_t src) {
    const int lo = __builtin_amdgcn_ds_bpermute((int)(src << 2), __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute((int)(src << 2), __double2hiint(v));
    return __hiloint2double(hi, lo);
}
// minimum over the 8 lanes of a group, returned in every lane of the group
__device__ __forceinline__ double grp8_min_f64(double v) {
    v = min_f64(v, dpp_f64<0xB1>(v));
    v = min_f64(v, dpp_f64<0x4E>(v));
    v = min_f64(v, dpp_f64<0x141>(v));  // row_half_mirror: reverses each half row
    return v;
}
__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t ab = (a < b) ? a : b;
    return (ab < c) ? ab : c;  // v_min3_u32
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}

// FULL: also `adapt` (per-chain bounds c, multiplied by `factor` when a proposal violates its bound, src/fact_samplers.jl:67-70)
// and a target with a mean (Γμ subtracted from the gradient); the north-star instantiation has neither.
template <bool PROF, bool FULL = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void zz_local_spec8_kernel(ZzRunParams P) {
    constexpr int E = 8;
    constexpr uint32_t SW = 7, PW = 1, KMAX = 5, R_ = 4 + PW + KMAX, WPAD = 58, W2 = WPAD / 2;
    const uint32_t nblk = P.nblk;  // 32-key blocks that hold coordinates (<= S8_NBLK); first-level entries beyond stay +Inf
    const int lane = threadIdx.x;
    const int g = lane >> 3;  // group = event slot
    const int gl = lane & 7;  // lane inside the group
    const int64_t chain = blockIdx.x;
    const int64_t d = P.d;

    extern __shared__ __align__(16) unsigned char smem[];
    double* const LU = reinterpret_cast<double*>(smem + S8_LU);
    double* const SLT = reinterpret_cast<double*>(smem + S8_SLT);
    double* const Lr = reinterpret_cast<double*>(smem + S8_LR);
    double* const LBr = reinterpret_cast<double*>(smem + S8_LBR);
    double* const Mr = reinterpret_cast<double*>(smem + S8_MR);
    uint32_t* const SLB = reinterpret_cast<uint32_t*>(smem + S8_SLB);
    uint32_t* const Z = reinterpret_cast<uint32_t*>(smem + S8_Z);
    double* const bk = reinterpret_cast<double*>(smem + S8_BK);
    uint16_t* const bi = reinterpret_cast<uint16_t*>(smem + S8_BI);
    uint64_t* const LB = reinterpret_cast<uint64_t*>(smem + S8_LB);
    double* const TK = reinterpret_cast<double*>(smem + S8_TK);
    uint32_t* const TB = reinterpret_cast<uint32_t*>(smem + S8_TB);
    double* const SELDT = reinterpret_cast<double*>(smem + S8_SELDT);
    uint32_t* const PR = reinterpret_cast<uint32_t*>(smem + S8_PR);
    double* const sx = reinterpret_cast<double*>(smem + S8_SX) + g * S8_SXP;
    double* const sth = reinterpret_cast<double*>(smem + S8_STH) + g * S8_SXP;
    double* const pk = reinterpret_cast<double*>(smem + S8_PK) + g * 32;
    uint32_t* const zg = Z + g * 16;
    const uint32_t pk_t = (uint32_t)g & 1u;  // odd groups store the two 16-byte pieces of a lane's chunk swapped: no bank conflicts

    ZzRec* rec = P.rec + chain * d;
    double* keys = P.keys + chain * P.dk;
    DevChain* hdr = P.hdr + chain;
    pdmp_event* ev = P.ev ? P.ev + chain * P.trace_cap : nullptr;
    double* cmut = (FULL && P.c_chain) ? (P.c_chain + chain * d) : nullptr;
    const bool adapt = FULL && P.adapt != 0;

    uint32_t status = hdr->c.status;
    if (status == PDMP_CHAIN_BOUND_VIOLATED || status == PDMP_CHAIN_STALLED) return;
    const uint64_t seed = hdr->seed;
    const uint64_t nm0 = hdr->c.ndraw_main, ntrace0 = hdr->c.ntrace;
    uint32_t dnm = 0, dnum = 0, dnacc = 0;
    uint32_t vnacc = 0;  // 1 if the launch ends on a bound violation (acc is bumped before the check)
    // the flow's refresh clock (src/sfact.jl:78-114, round 6): its time is a wave-uniform scalar here, not a queue entry (key d lies behind the
    // first level's 512 blocks); its events are processed by themselves between the speculative iterations, as zz_local_run_kernel does
    const bool has_refresh = P.has_refresh != 0;
    double t_ref = has_refresh ? keys[d] : PDMP_INF;
    uint32_t dnref = 0;                 // refresh events of this launch (recorded in the trace like reflections, :143)
    uint64_t ng = hdr->c.ndraw_global;  // the "global rng" stream of the clock (:80,:84,:108)
    double t_last = hdr->c.t_last;
    double t_event = hdr->t_event;
    status = PDMP_CHAIN_OK;

    const double T = P.T;
    const bool stop_before = (P.flags & PDMP_RUN_STOP_BEFORE) != 0;
    const uint32_t trace_room = (P.trace_cap > 0)
                                    ? (uint32_t)(((uint64_t)P.trace_cap > ntrace0) ? ((uint64_t)P.trace_cap - ntrace0) : 0)
                                    : 0xffffffffu;
    const uint32_t common = P.common_tix;

    if (lane == 0) SELDT[0] = 1e-3;  // any positive start: the steering rule finds the scale within a few iterations
    // slot 0 <- the common template, for the whole launch
    if (lane < (int)W2) {
        reinterpret_cast<ulonglong2*>(LB)[lane] = reinterpret_cast<const ulonglong2*>(P.blob + (size_t)common * WPAD)[lane];
    }
    for (uint32_t b = lane; b < nblk; b += 64) {
        const double* kp = keys + (size_t)b * 32;
        // (the refresh clock's slot, key d, sits inside the last block when d is no multiple of 32: it is no coordinate -- its time lives in t_ref,
        // and the slot holds +Inf in memory for as long as this launch runs, so that the rescans of its block do not see it either)
        double mk = (has_refresh && b * 32 == (uint32_t)d) ? PDMP_INF : kp[0];
        uint32_t mi = 0;
#pragma unroll 8
        for (int q = 1; q < 32; ++q) {
            const double v = (has_refresh && b * 32 + (uint32_t)q == (uint32_t)d) ? PDMP_INF : kp[q];
            if (v < mk) {
                mk = v;
                mi = q;
            }
        }
        bk[b] = mk;
        bi[b] = (uint16_t)(b * 32 + mi);
    }
    for (uint32_t b = nblk + lane; b < S8_NBLK; b += 64) {
        bk[b] = PDMP_INF;
        bi[b] = 0;
    }
    LDS_ORDER();
    if (has_refresh && lane == 0) __hip_atomic_store(keys + d, PDMP_INF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    uint32_t rng_base = 0xffffffffu;
    double ureg = 0.0;  // draw rng_base + lane of the chain's stream
    uint64_t ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t ph_t0 = PROF ? (uint64_t)__builtin_readcyclecounter() : 0;
    uint64_t ph_iters = 0;
#define PHASE(k)                                                          \
    do {                                                                  \
        if (PROF) {                                                       \
            const uint64_t now_ = (uint64_t)__builtin_readcyclecounter(); \
            ph[k] += now_ - ph_t0;                                        \
            ph_t0 = now_;                                                 \
        }                                                                 \
    } while (0)

    bool running = stop_before || (t_event < T);
    PrioTurn prio;
    while (running) {
        prio.step();
        if (dnacc + dnref >= trace_room) {
            status = PDMP_CHAIN_TRACE_FULL;
            break;
        }
        if (dnm >= P.count_limit) {  // (32-bit counters of the launch: pause, the host runs again)
            status = PDMP_CHAIN_PAUSED;
            break;
        }
        // ---------------- select the (up to) E smallest block minima, in time order, WITHOUT a tournament per candidate: one
        // wave minimum m, then every first-level entry below the threshold m + sel_dt is a candidate -- four compares and four
        // population counts tell how many there are.  The candidates (at most SEL_CAP, else the threshold is halved) are compacted
        // into LDS by ballot prefix counts, each ranks itself against the others with broadcast reads, and ranks 0..E-1 become
        // the event slots.  Whatever sel_dt is, the slots hold exactly the smallest entries of the queue, so the committed
        // sequence does not depend on it; it is steered towards ~12 candidates per iteration.
        int Esel = 0;
        bool first_inf = false, do_ref = false;
        {
            double kk[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) kk[j] = bk[lane + 64 * j];
            const double mloc = min_f64(min_f64(min_f64(kk[0], kk[1]), min_f64(kk[2], kk[3])),
                                        min_f64(min_f64(kk[4], kk[5]), min_f64(kk[6], kk[7])));
            const double mq = wave_min_f64(mloc);
            do_ref = has_refresh && t_ref < mq && !(stop_before && !(t_ref < T));  // (a coordinate's event at the clock's very time goes first)
            if (do_ref) {
            } else if (!(mq < PDMP_INF)) {
                first_inf = true;
            } else if (!(stop_before && !(mq < T))) {
                if (lane < (int)SEL_CAP) TK[lane] = PDMP_INF;
                double dt_sel = uniform_f64(SELDT[0]);
                // (the candidate masks are recomputed where they are needed instead of being kept: eight 64-bit masks would
                // crowd the scalar registers)
                // Compaction: entry (lane, j) gets index (candidates of slots < j) + (candidates of slot j in lower lanes).  There is
                // no separate counting pass: the scratch arrays take up to 64 candidates, and a pass that ends with more than
                // SEL_CAP is repeated with half the threshold.
                auto below = [](uint64_t m_) -> uint32_t {
                    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_, 0u));
                };
                double tau;
                uint32_t C;
                for (int tries = 0;; ++tries) {
                    tau = mq + dt_sel;  // (>= mq: the minimum itself always qualifies)
                    if (stop_before && !(tau < T)) tau = pdmp_below(T);
                    if (!(tau < t_ref)) tau = (t_ref > mq) ? pdmp_below(t_ref) : mq;  // nothing at or beyond the refresh clock's time (but the minimum itself)
                    const bool pile = tries > 64;  // more than SEL_CAP entries EQUAL to the minimum: one (lowest block) per iteration
                    if (tries >= 64) tau = mq;     // a pile of exactly equal keys: the entries equal to the minimum only
                    uint32_t base = 0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const bool cj_ = kk[j] <= tau;
                        uint64_t Mj = __ballot(cj_);
                        if (pile) Mj = (base == 0 && Mj) ? (Mj & (~Mj + 1)) : 0ull;
                        if (cj_ && ((Mj >> lane) & 1ull)) {
                            const uint32_t ix = base + below(Mj);
                            if (ix < 64u) {
                                TK[ix] = kk[j];
                                TB[ix] = (uint32_t)lane + 64u * j;
                            }
                        }
                        base += (uint32_t)__popcll(Mj);
                    }
                    C = base;
                    if (C <= SEL_CAP) break;
                    dt_sel *= 0.5;
                    LDS_ORDER();
                    if (lane < (int)SEL_CAP) TK[lane] = PDMP_INF;  // (entries past the new count must read +Inf in the ranking)
                }
                LDS_ORDER();
                // rank of candidate n among all (ties by index), on a 16 x 4 grid: lane = 16 * part + n counts the candidates
                // 4 * part .. 4 * part + 3 that precede n; the four partial counts meet in LDS.  Unused entries hold +Inf.
                {
                    const uint32_t n = (uint32_t)lane & 15u, part = (uint32_t)lane >> 4;
                    const double own = TK[n];
                    const double2* T2 = reinterpret_cast<const double2*>(TK + 4 * part);
                    const double2 o01 = T2[0], o23 = T2[1];
                    const uint32_t q = 4 * part;
                    uint32_t pr = 0;
                    pr += (o01.x < own || (o01.x == own && q + 0 < n)) ? 1u : 0u;
                    pr += (o01.y < own || (o01.y == own && q + 1 < n)) ? 1u : 0u;
                    pr += (o23.x < own || (o23.x == own && q + 2 < n)) ? 1u : 0u;
                    pr += (o23.y < own || (o23.y == own && q + 3 < n)) ? 1u : 0u;
                    PR[n * 4 + part] = pr;
                    LDS_ORDER();
                    if ((uint32_t)lane < C) {
                        const uint4 p4 = reinterpret_cast<const uint4*>(PR)[lane];
                        const uint32_t rank = p4.x + p4.y + p4.z + p4.w;
                        if (rank < (uint32_t)E) {
                            SLT[rank] = own;
                            SLB[rank] = TB[lane];
                        }
                    }
                }
                Esel = (C < (uint32_t)E) ? (int)C : E;
                // steer the threshold: ~10 candidates next time
                const double f = (C > 14u) ? 0.8 : (C < 11u) ? ((C < 6u) ? 2.0 : 1.2) : 1.0;
                if (lane == 0) SELDT[0] = dt_sel * f;
            }
        }
        if (do_ref) {
            // ---------------- the refresh clock is the chain's next event: src/sfact.jl:78-114, restated with its quirks exactly as
            // zz_local_run_kernel does (two independent coordinate draws from the global stream -- the neighbourhood that is moved, :80, and the
            // coordinate that is refreshed, :84 --, G1[i] re-bounded at the coordinates' own, possibly stale, clocks); the whole wave on one event
            const double tp = t_ref;
            uint64_t* const lb0 = LB + WPAD;  // (blob slot 1: free between iterations)
            double* const sx0 = reinterpret_cast<double*>(smem + S8_SX);
            double* const sth0 = reinterpret_cast<double*>(smem + S8_STH);
            const uint64_t nm = nm0 + (uint64_t)dnm;
            t_last = tp;
            const uint32_t i1 = pdmp_randint(seed, PDMP_STREAM_GLOBAL, ng, (uint32_t)d);
            ng += 1;
            {
                const uint64_t* bsrc = P.blob + (size_t)P.tix[i1] * WPAD;
                if (lane < (int)WPAD) lb0[lane] = bsrc[lane];
                LDS_ORDER();
                const int k1 = (int)uniform_u32((uint32_t)(lb0[0] & 0xff));
                if (lane < k1) {  // smove_forward!(G, i1, ...), :82
                    const uint64_t sw = lb0[1 + (lane >> 1)];
                    const uint32_t s1 = i1 + ((lane & 1) ? (uint32_t)(sw >> 32) : (uint32_t)sw);
                    ZzRec* r1 = rec + s1;
                    const double x0 = r1->x, th0 = r1->th, t0 = r1->t, I0 = r1->I;
                    const double dt = tp - t0;
                    const double xn = x0 + th0 * dt;
                    r1->x = xn;
                    r1->t = tp;
                    r1->I = I0 + dt * ((x0 + xn) * 0.5);
                }
                LDS_ORDER();
            }
            const uint32_t i2 = pdmp_randint(seed, PDMP_STREAM_GLOBAL, ng, (uint32_t)d);
            ng += 1;
            {
                const uint64_t* bsrc = P.blob + (size_t)P.tix[i2] * WPAD;
                if (lane < (int)WPAD) lb0[lane] = bsrc[lane];
            }
            LDS_ORDER();
            const uint64_t hw = lb0[0];
            const int k = (int)uniform_u32((uint32_t)(hw & 0xff));
            const int m = (int)uniform_u32((uint32_t)((hw >> 8) & 0xff));
            const int self = (int)uniform_u32((uint32_t)((hw >> 16) & 0xff));
            uint32_t s = i2;
            if (lane < m) {
                const uint64_t sw = lb0[1 + (lane >> 1)];
                s = i2 + ((lane & 1) ? (uint32_t)(sw >> 32) : (uint32_t)sw);
            }
            ZzRec* rs = rec + s;
            double x = 0.0, th = 0.0, t = 0.0, I = 0.0;
            if (lane < m) {
                x = rs->x;
                th = rs->th;
                t = rs->t;
                I = rs->I;
            }
            if (lane >= k && lane < m) {  // smove_forward!(G2, i, ...), :85
                const double dt = tp - t;
                const double xn = x + th * dt;
                I = I + dt * ((x + xn) * 0.5);
                x = xn;
                t = tp;
            }
            const double usign = pdmp_u01(seed, PDMP_STREAM_MAIN, nm);  // θ[i] = σ[i]*rand(rng, (-1,1)), :100-101
            if (lane == self) th = P.tb.sigma[i2] * ((usign < 0.5) ? -1.0 : 1.0);
            const double newref = tp + (-pdmp_log(pdmp_u01(seed, PDMP_STREAM_GLOBAL, ng))) / P.lambda_ref;  // :108
            ng += 1;
            if (lane < m) {
                sx0[lane] = x;
                sth0[lane] = th;
            }
            LDS_ORDER();
            const uint32_t sub0 = 1 + SW + (uint32_t)lane * R_;
            double key = PDMP_INF;
            if (lane < k) {  // :110-114
                const double gmu = __longlong_as_double((long long)lb0[sub0 + 1]);
                const double cj = cmut ? cmut[s] : __longlong_as_double((long long)lb0[sub0 + 2]);
                const int kj = (int)(lb0[sub0 + 3] & 0xff);
                const uint64_t pw = lb0[sub0 + 4];
                double gx = 0.0, gt = 0.0;
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    if (q < kj) {
                        const double v = __longlong_as_double((long long)lb0[sub0 + 4 + PW + q]);
                        const int ps = (int)((pw >> (8 * q)) & 0xff);
                        gx += v * sx0[ps];
                        gt += v * sth0[ps];
                    }
                }
                const double a = cj + (gx - gmu) * th;
                const double b = cj / 100 + th * gt;
                const double L = pdmp_log(pdmp_u01(seed, PDMP_STREAM_MAIN, nm + 1 + (uint64_t)lane));
                key = t + dev_poisson_time_L(a, b, L);
                rs->t_old = t;
                rs->a = a;
                rs->b = b;
                keys[s] = key;
            }
            dnm += 1u + (uint32_t)k;
            if (lane < m) {
                rs->x = x;
                rs->th = th;
                rs->t = t;
                rs->I = I;
            }
            t_ref = newref;  // (stored into keys[d] when the launch ends)
            for (int jj = 0; jj < k; ++jj) {  // first level: blocks of 32 keys
                const uint32_t j = readlane_u32(s, jj);
                const double kjv = readlane_f64(key, jj);
                const uint32_t bj = j >> 5;
                LDS_ORDER();
                const double cur = bk[bj];
                const uint32_t ci = bi[bj];
                if (kjv < cur || (kjv == cur && j < ci)) {
                    if (lane == 0) {
                        bk[bj] = kjv;
                        bi[bj] = (uint16_t)j;
                    }
                } else if (ci == j) {
                    const double kv = (lane < 32) ? __hip_atomic_load(keys + (size_t)bj * 32 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : PDMP_INF;
                    const double mn = wave_min_f64(kv);
                    const uint64_t bl = __ballot(kv == mn);
                    const int arg = bl ? (__ffsll((unsigned long long)bl) - 1) : 0;
                    if (lane == 0) {
                        bk[bj] = mn;
                        bi[bj] = (uint16_t)(bj * 32 + (uint32_t)arg);
                    }
                }
                LDS_ORDER();
            }
            const double t_i = readlane_f64(t, self), x_i = readlane_f64(x, self), th_i2 = readlane_f64(th, self);
            if (ev && lane == 0) {  // event(i, t, x, θ, F) = (t[i], i, x[i], θ[i]), :143
                pdmp_event e;
                e.t = t_i;
                e.i = (int64_t)i2;
                e.x = x_i;
                e.theta = th_i2;
                ev[ntrace0 + dnacc + dnref] = e;
            }
            dnref += 1;
            t_event = tp;
            if (!stop_before && !(tp < T)) running = false;
            LDS_ORDER();
            continue;
        }
        if (Esel == 0) {
            if (first_inf && !(t_ref < PDMP_INF)) status = PDMP_CHAIN_STALLED;
            if (first_inf && t_ref < PDMP_INF && !(stop_before && !(t_ref < T))) {  // (no coordinate has a finite key, the clock does: its event is next)
                continue;
            }
            break;
        }
        LDS_ORDER();
        PHASE(0);
        if (PROF) ph_iters += 1;
        bool gvalid = g < Esel;
        const double tp = gvalid ? SLT[g] : PDMP_INF;
        const uint32_t blk = gvalid ? SLB[g] : 0u;
        const uint32_t i = gvalid ? (uint32_t)bi[blk] : 0u;
        const uint32_t tixi = gvalid ? P.tix[i] : common;

        // ---------------- candidate draws (window of 64 draws and their logs in LDS, as in zz_local_spec_kernel)
        if (dnm < rng_base || dnm + E * (1u + KMAX) > rng_base + 64u) {
            rng_base = dnm;
            ureg = pdmp_u01(seed, PDMP_STREAM_MAIN, nm0 + (uint64_t)dnm + (uint64_t)lane);
            LU[lane] = pdmp_log(ureg);
        }
        const uint32_t rng_off = dnm - rng_base;
        // ---------------- blob slots: 0 for the common template, 1 and 2 for the first two events that need another one
        uint32_t slot = 0;
        {
            const bool nc = gvalid && tixi != common;
            const uint64_t ncball = __ballot(nc && gl == 0);
            if (ncball != 0) {
                const uint32_t rank = (uint32_t)__popcll(ncball & ((1ull << (8 * g)) - 1ull));
                if (__popcll(ncball) > 2) {  // the third such event and everything after it wait for the next iteration
                    uint64_t m_ = ncball;
                    m_ &= m_ - 1;
                    m_ &= m_ - 1;
                    const int cut = (__ffsll((unsigned long long)m_) - 1) >> 3;
                    Esel = cut;
                    gvalid = g < Esel;
                }
                if (nc && gvalid) {
                    slot = 1 + rank;
                    const ulonglong2* bsrc = reinterpret_cast<const ulonglong2*>(P.blob + (size_t)tixi * WPAD);
                    ulonglong2* bdst = reinterpret_cast<ulonglong2*>(LB + slot * WPAD);
                    for (uint32_t w = gl; w < W2; w += 8) bdst[w] = bsrc[w];
                }
            }
        }
        const uint64_t* lb = LB + slot * WPAD;
        LDS_ORDER();
        PHASE(1);
        // ---------------- neighbourhood header and member list: positions gl and gl + 8 of S[i]
        // (straight-line: every lane reads its slot's words, the selects below sort out who is a member)
        const uint32_t hw = gvalid ? (uint32_t)lb[0] : 0u;
        const int k = (int)(hw & 0xff), m = (int)((hw >> 8) & 0xff), self = (int)((hw >> 16) & 0xff);
        const bool memberA = gl < m, memberB = gl + 8 < m;  // (m = 0 in an empty slot)
        const uint64_t swa = lb[1 + (gl >> 1)], swb = lb[5 + (gl >> 1)];
        const uint32_t sA = memberA ? i + ((gl & 1) ? (uint32_t)(swa >> 32) : (uint32_t)swa) : 0xffffff00u + (uint32_t)lane;
        const uint32_t sB = memberB ? i + ((gl & 1) ? (uint32_t)(swb >> 32) : (uint32_t)swb) : 0xffffff40u + (uint32_t)lane;
        PHASE(2);
        // All HBM loads of the iteration in ONE straight-line batch (no exec-masked regions: lanes without a record of their own
        // read i's, which coalesces with the group's other readers of it; empty slots read coordinate 0): the wait counters
        // stay exact and nothing here is serialised behind an earlier round trip.
        ZzRec* rsA = rec + ((memberA && gl < k) ? sA : i);
        ZzRec* rsB = rec + (memberB ? sB : i);
        const ZzRec* ri = rec + i;
        double x = rsA->x, th = rsA->th, t = rsA->t, I = rsA->I;
        const double told_i = ri->t_old, a_i = ri->a, b_i = ri->b;
        const uint64_t acc_i = ri->acc;
        double kq[4];
        {
            const double2* kp = reinterpret_cast<const double2*>(keys + (size_t)blk * 32 + gl * 4);
            const double2 k01 = kp[0], k23 = kp[1];
            kq[0] = k01.x;
            kq[1] = k01.y;
            kq[2] = k23.x;
            kq[3] = k23.y;
        }
        rsA = rec + (memberA ? sA : i);
        zg[gl] = sA;
        zg[8 + gl] = sB;
        const uint32_t sub = 1 + SW + (uint32_t)gl * R_;
        double cj = __longlong_as_double((long long)lb[sub + 2]);  // (used by lanes gl < k only)
        if (FULL && cmut) cj = cmut[(gl < k) ? sA : i];

        // ---------------- zone conflicts with earlier groups: id spans first, the exact id comparison only for pairs of groups
        // whose spans overlap
        LDS_ORDER();
        uint64_t confball;
        {
            uint32_t lo = memberA ? sA : 0xffffffffu, hi = memberA ? sA : 0u;
            lo = (memberB && sB < lo) ? sB : lo;
            hi = (memberB && sB > hi) ? sB : hi;
            uint32_t o;
            o = dpp_u32<0xB1>(lo);   lo = (o < lo) ? o : lo;
            o = dpp_u32<0x4E>(lo);   lo = (o < lo) ? o : lo;
            o = dpp_u32<0x141>(lo);  lo = (o < lo) ? o : lo;
            o = dpp_u32<0xB1>(hi);   hi = (o > hi) ? o : hi;
            o = dpp_u32<0x4E>(hi);   hi = (o > hi) ? o : hi;
            o = dpp_u32<0x141>(hi);  hi = (o > hi) ? o : hi;
            // A pair of groups (q < g) whose spans overlap is compared exactly by the WHOLE wave: lane L holds id L & 15 of g
            // against ids 4 (L >> 4) .. + 3 of q -- the 256 id pairs in four xor / two min instructions per lane.  Empty
            // positions hold sentinels that equal nothing.
            uint32_t confmask = 0;
            const uint4* Z4 = reinterpret_cast<const uint4*>(Z);
            // lane gl of group g looks at the pair (g, q = gl): one ballot finds all pairs of groups whose spans overlap
            {
                const uint32_t lq = (uint32_t)__builtin_amdgcn_ds_bpermute(32 * gl, (int)lo);  // span of group gl (its lane 0)
                const uint32_t hq = (uint32_t)__builtin_amdgcn_ds_bpermute(32 * gl, (int)hi);
                uint64_t ovb = __ballot(gvalid && gl < g && lo <= hq && lq <= hi);
                while (ovb != 0) {
                    const int bit = __ffsll((unsigned long long)ovb) - 1;
                    const int gsel = bit >> 3, q = bit & 7;
                    ovb &= ovb - 1;
                    const uint32_t idg = Z[gsel * 16 + (lane & 15)];
                    const uint4 zq = Z4[q * 4 + (lane >> 4)];
                    const uint32_t mn = umin3(idg ^ zq.x, idg ^ zq.y, umin3(idg ^ zq.z, idg ^ zq.w, 0xffffffffu));
                    if (__ballot(mn == 0u) != 0) confmask |= 1u << gsel;
                }
            }
            confball = confmask;
        }
        PHASE(3);

        // ---------------- smove_forward!(G, i, ...), gradient, rates
        // (every lane runs the move: lanes that hold no G1 member carry i's record and are re-loaded before they matter)
        {
            const double dt = tp - t;
            const double xn = x + th * dt;
            I = I + dt * ((x + xn) * 0.5);
            x = xn;
            t = tp;
            sx[gl] = x;
            sth[gl] = th;
        }
        LDS_ORDER();
        double l, lbound;
        {
            // Γ[:, i] . x in ascending row order; the template's entries past k are 0.0 and sx[0..7] are all finite numbers of
            // this iteration, so the unconditional tail adds exact zeros
            double gr = 0.0;
#pragma unroll
            for (uint32_t p = 0; p < KMAX; ++p) gr += __longlong_as_double((long long)lb[1 + SW + p * R_]) * sx[p];
            if (FULL && P.tb.gmu_t) gr = gr - P.tb.gmu_t[i];
            const double th_i = sth[self];
            l = pos_part(gr * th_i);
            lbound = pos_part(a_i + b_i * (tp - told_i));
            if (gl == 0) {
                Lr[g] = l;
                LBr[g] = lbound;
            }
        }
        LDS_ORDER();
        // ---------------- accept chain in time order.  Lane o evaluates every event's test for the draw at offset o; the ballots
        // are then walked on the scalar unit: event r reads its bit at the offset the earlier outcomes imply.
        // The offsets (each <= 48) travel packed six bits apiece in one 64-bit scalar: offset after r events = bits 6r .. 6r+5.
        uint32_t accbits = 0;
        uint64_t offpack = 0;
        {
            const double coin = bperm_f64(ureg, (rng_off + (uint32_t)lane) & 63u);
            uint32_t off = 0;
#pragma unroll
            for (int r = 0; r < E; ++r) {  // slots >= Esel hold stale rates: their bits are masked off below, their offsets unused
                const uint64_t am_r = __ballot(coin * LBr[r] < Lr[r]);  // :121
                const uint32_t a_r = (uint32_t)(am_r >> off) & 1u;
                const uint32_t k_r = readlane_u32((uint32_t)k, 8 * r);
                off += a_r ? (1u + k_r) : 2u;
                off = (off < 63u) ? off : 63u;  // (only stale slots can run past the window; keeps the shifts defined)
                accbits |= a_r << r;
                offpack |= (uint64_t)off << (6 * (r + 1));
            }
            accbits &= (1u << Esel) - 1u;
        }
        const uint32_t myoff = (uint32_t)(offpack >> (6 * g)) & 63u;
        const bool accept = gvalid && ((accbits >> g) & 1u) != 0;
        const bool violated = accept && (l >= lbound);  // :123
        PHASE(4);

        double x2 = 0.0, th2 = 0.0, t2 = 0.0, I2 = 0.0;
        if (accept) {
            if (gl >= k && gl < m) {  // smove_forward!(G2, i, ...), :129
                x = rsA->x;
                th = rsA->th;
                t = rsA->t;
                I = rsA->I;
                const double dt = tp - t;
                const double xn = x + th * dt;
                I = I + dt * ((x + xn) * 0.5);
                x = xn;
                t = tp;
            }
            if (memberB) {
                x2 = rsB->x;
                th2 = rsB->th;
                t2 = rsB->t;
                I2 = rsB->I;
                const double dt = tp - t2;
                const double xn = x2 + th2 * dt;
                I2 = I2 + dt * ((x2 + xn) * 0.5);
                x2 = xn;
                t2 = tp;
                sx[8 + gl] = x2;
                sth[8 + gl] = th2;
            }
            if (gl == self) th = -th;  // reflect!, :130
            if (gl < m) {
                sx[gl] = x;
                sth[gl] = th;
            }
        }
        LDS_ORDER();
        // ---------------- re-bound (ab + poisson_time) -- results stay in registers until the commit
        const bool active = gvalid && (accept ? (gl < k) : (gl == self));
        double key = PDMP_INF, a = 0.0, b = 0.0;
        if (active) {
            const double gmu = __longlong_as_double((long long)lb[sub + 1]);
            const int kj = (int)(lb[sub + 3] & 0xff);
            const uint64_t pw = lb[sub + 4];
            double gx = 0.0, gt = 0.0;
#pragma unroll
            for (int q = 0; q < (int)KMAX; ++q) {
                if (q < kj) {
                    const double v = __longlong_as_double((long long)lb[sub + 4 + PW + q]);
                    const int ps = (int)((pw >> (8 * q)) & 0xff);
                    gx += v * sx[ps];
                    gt += v * sth[ps];
                }
            }
            if (FULL && violated && gl == self) cj *= P.factor;  // adapt!(c, i, factor), :127 (stored at commit)
            a = cj + (gx - gmu) * th;
            b = cj / 100 + th * gt;
            const double L = LU[(rng_off + myoff + 1u + (accept ? (uint32_t)gl : 0u)) & 63u];
            key = t + dev_poisson_time_L(a, b, L);
        }
        LDS_ORDER();
        // the patched copy of the popped key block goes where sx / sth / the zone ids were: all their readers are done
        // (the four 16-byte pieces of a lane's 64-byte chunk are stored in the order piece ^ pk_t, pk_t = 0..3 over the four lanes
        // of a quarter wave that would otherwise share their banks: b128 accesses without bank conflicts)
        {
            double2* pk2 = reinterpret_cast<double2*>(pk + gl * 4);
            pk2[0 ^ pk_t] = make_double2(kq[0], kq[1]);
            pk2[1 ^ pk_t] = make_double2(kq[2], kq[3]);
        }
        LDS_ORDER();
        if (active && (sA >> 5) == blk) {
            const uint32_t e_ = sA & 31u;
            pk[(e_ & ~3u) + ((((e_ & 3u) >> 1) ^ pk_t) << 1) + (e_ & 1u)] = key;
        }
        LDS_ORDER();
        PHASE(5);
        // ---------------- patched minimum of the popped block, and everything this event could expose
        double rowmin;
        uint32_t cand;
        int wl2;
        {
            const double2* pk2 = reinterpret_cast<const double2*>(pk + gl * 4);
            const double2 p01 = pk2[0 ^ pk_t], p23 = pk2[1 ^ pk_t];
            double lm = p01.x;
            uint32_t li = 0;
#define PMIN(v, idx)    \
    do {                \
        if ((v) < lm) { \
            lm = (v);   \
            li = (idx); \
        }               \
    } while (0)
            PMIN(p01.y, 1);
            PMIN(p23.x, 2);
            PMIN(p23.y, 3);
#undef PMIN
            cand = blk * 32u + (uint32_t)gl * 4u + li;
            rowmin = grp8_min_f64(lm);
            const uint64_t winball = __ballot(gvalid && lm == rowmin);
            wl2 = __ffs((unsigned)((winball >> (8 * g)) & 0xffu)) - 1;
        }
        const double keymin = grp8_min_f64(key);
        const double expose = min_f64(rowmin, keymin);
        if (gl == 0) Mr[g] = expose;
        LDS_ORDER();
        // ---------------- validate: event g commits iff all earlier ones do, its zone is disjoint from theirs, and nothing they
        // produce or expose comes before it
        uint32_t Rc;
        uint32_t nacc_c;
        int vsel = -1;  // the event that violates its bound, if it is the chain's next one
        {
            double pref = PDMP_INF;
#pragma unroll
            for (int q = 0; q < E - 1; ++q) {
                const double mq = Mr[q];
                pref = (q < g) ? min_f64(pref, mq) : pref;
            }
            const bool confg = ((confball >> g) & 1ull) != 0;
            const bool okg = gvalid && ((g == 0) || (!confg && pref > tp));
            const bool vstop = violated && !adapt;  // reference: error(...), :124 -> the event is not committed
            const uint64_t okball = __ballot(okg && !vstop && gl == 0);
            const uint64_t vball = __ballot(okg && vstop && gl == 0);
            const uint64_t accball = __ballot(accept && gl == 0);
            // length of the run of committable slots from slot 0 (one bit per slot at bit 8 r): first zero among those bits
            const uint64_t gap = ~okball & 0x0101010101010101ull;
            const uint32_t r_ok = gap ? (uint32_t)((__ffsll((unsigned long long)gap) - 1) >> 3) : (uint32_t)E;
            Rc = 0;
            nacc_c = 0;
            bool stopped = false;
            // the usual case needs no walk: the slice mode stops on time alone, and the trace has room for every accepted slot
            const uint32_t nacc_all = (uint32_t)__popcll(accball & ((r_ok < 8u) ? ((1ull << (8 * r_ok)) - 1ull) : ~0ull));
            const bool plainrun = stop_before && !(P.trace_cap > 0 && dnacc + dnref + nacc_all >= trace_room);
            if (plainrun) {
                Rc = r_ok;
                nacc_c = nacc_all;
            }
            for (uint32_t r = 0; !plainrun && r < r_ok && !stopped; ++r) {
                Rc = r + 1;
                if ((accball >> (8 * r)) & 1ull) {
                    nacc_c += 1;
                    if (dnacc + dnref + nacc_c >= trace_room && P.trace_cap > 0) {
                        status = PDMP_CHAIN_TRACE_FULL;
                        stopped = true;
                    }
                    if (!stop_before && !(uniform_f64(SLT[r]) < T)) {
                        running = false;
                        stopped = true;
                    }
                }
            }
            if (!stopped && r_ok < (uint32_t)E && ((vball >> (8 * r_ok)) & 1ull)) {
                status = PDMP_CHAIN_BOUND_VIOLATED;
                vsel = (int)r_ok;
            }
        }
        PHASE(6);

        // ---------------- commit the valid prefix
        const bool commit = gvalid && (uint32_t)g < Rc;
        const uint64_t accball2 = __ballot(commit && accept && gl == 0);
        if (commit) {
            if (gl < (accept ? m : k)) {
                rsA->x = x;
                rsA->th = th;
                rsA->t = t;
                rsA->I = I;
            }
            if (accept && memberB) {
                rsB->x = x2;
                rsB->th = th2;
                rsB->t = t2;
                rsB->I = I2;
            }
            if (active) {
                rsA->t_old = t;
                rsA->a = a;
                rsA->b = b;
                keys[sA] = key;
                if (FULL && violated && gl == self) cmut[sA] = cj;
            }
            if (accept && gl == self) rsA->acc = acc_i + 1;
            if (gl == wl2) {
                bk[blk] = rowmin;
                bi[blk] = (uint16_t)cand;
            }
            if (accept && gl == self && ev) {
                const uint32_t rank = (uint32_t)__popcll(accball2 & ((1ull << (8 * g)) - 1ull));
                pdmp_event e;
                e.t = tp;
                e.i = (int64_t)i;
                e.x = x;
                e.theta = th;
                ev[ntrace0 + dnacc + dnref + rank] = e;
            }
        }
        LDS_ORDER();
        PHASE(7);
        // ---------------- level-1 updates for re-bounded neighbours living in other blocks.  The final entry of a block is the
        // smallest (key, coordinate) among its old entry and the new keys, whatever the order -- so when no two of these lanes aim
        // at one block (checked through a small claim table) and none has to rescan, every lane updates its block by itself, in
        // one LDS round trip for all of them; otherwise the updates are made one by one in event order.
        const bool upd = commit && accept && gl < k && (sA >> 5) != blk;
        if (__ballot(upd) != 0) {
            uint8_t* const CL = reinterpret_cast<uint8_t*>(smem + S8_CL);
            LDS_ORDER();
            const uint32_t bjv = upd ? (sA >> 5) : 0u;
            const double curv = bk[bjv];
            const uint32_t civ = bi[bjv];
            const bool lower = upd && (key < curv || (key == curv && sA < civ));
            const bool resc = upd && !lower && civ == sA;
            if (lower) CL[bjv & 63u] = (uint8_t)lane;
            LDS_ORDER();
            const bool lost = lower && CL[bjv & 63u] != (uint8_t)lane;
            if (__ballot(lost || resc) == 0) {
                if (lower) {
                    bk[bjv] = key;
                    bi[bjv] = (uint16_t)sA;
                }
            } else {
            for (uint32_t r = 0; r < Rc; ++r) {
                if (!((accball2 >> (8 * r)) & 1ull)) continue;
                const uint32_t own = uniform_u32(SLB[r]);
                const int kr = (int)readlane_u32((uint32_t)k, 8 * (int)r);
                for (int jj = 0; jj < kr; ++jj) {
                    const uint32_t j = readlane_u32(sA, 8 * (int)r + jj);
                    if ((j >> 5) == own) continue;
                    // first-level entry of j's 32-key block: a lower key replaces it; if j WAS the entry and grew, the block is rescanned
                    const double kj = readlane_f64(key, 8 * (int)r + jj);
                    const uint32_t bj = j >> 5;
                    LDS_ORDER();
                    const double cur = bk[bj];
                    const uint32_t ci = bi[bj];
                    if (kj < cur || (kj == cur && j < ci)) {
                        if (lane == 0) {
                            bk[bj] = kj;
                            bi[bj] = (uint16_t)j;
                        }
                    } else if (ci == j) {
                        const double kv = __hip_atomic_load(keys + (size_t)bj * 32 + (lane & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const double mn = wave_min_f64(kv);
                        const uint64_t bl = __ballot(kv == mn);
                        const int arg = bl ? (__ffsll((unsigned long long)bl) - 1) : 0;
                        if (lane == 0) {
                            bk[bj] = mn;
                            bi[bj] = (uint16_t)(bj * 32 + (uint32_t)arg);
                        }
                    }
                }
            }
            }
        }
        PHASE(8);
        // ---------------- the violating proposal itself (reference: counted, G[i] moved, acc bumped -- then error(...), :120-124):
        // what zz_local_run_kernel and the oracle leave behind
        if (vsel >= 0) {
            if (g == vsel && gl < k) {
                rsA->x = x;
                rsA->t = t;
                rsA->I = I;
            }
            dnum += 1;
            vnacc = 1;
            dnm += ((uint32_t)(offpack >> (6 * vsel)) & 63u) + 1u - ((uint32_t)(offpack >> (6 * Rc)) & 63u);
        }
        // ---------------- counters
        if (Rc > 0) {
            dnum += Rc;
            dnacc += nacc_c;
            dnm += (uint32_t)(offpack >> (6 * Rc)) & 63u;
            t_last = uniform_f64(SLT[Rc - 1]);
            if (accball2) t_event = uniform_f64(SLT[(63 - __builtin_clzll(accball2)) >> 3]);
        }
        if (vsel >= 0) t_last = uniform_f64(SLT[vsel]);  // the violating event's time is the chain's current time
        if (status != PDMP_CHAIN_OK) break;
        LDS_ORDER();
    }

    if (PROF && P.dbg && chain == 0 && lane == 0) {
        for (int q = 0; q < 10; ++q) P.dbg[q] = (double)ph[q];
        P.dbg[10] = (double)ph_iters;
    }
#undef PHASE
    if (lane == 0) {
        hdr->c.t_last = t_last;
        hdr->t_event = t_event;
        hdr->c.num += dnum;
        hdr->c.nacc += dnacc + vnacc;
        hdr->c.ntrace = ntrace0 + dnacc + dnref;
        hdr->c.nevents += dnacc + dnref;
        hdr->c.nrefresh += dnref;
        hdr->c.ndraw_global = ng;
        if (has_refresh) keys[d] = t_ref;
        hdr->c.ndraw_main = nm0 + dnm;
        hdr->c.status = status;
    }
}

#include "pdmp_spec8g.inc"

// ------------------------------------------------------------------------------------------ tracked-gradient loop
//
// zz_local_spec8_kernel's scheme (eight event slots per iteration, threshold selection, scalar accept walk, exact validation, commit of
// the valid prefix) on a different evaluation of the SAME process: instead of moving the neighbourhood G[i] to t′ at every proposal and
// gathering Γ[:,i]·x from it (src/sfact.jl:82,116 -- five scattered read-modify-writes per proposal, each a whole 128-byte line from HBM,
// profiles/r02_*_pmc_calibration.txt), every coordinate carries g_i = Γ[:,i]·x and its rate of change gd_i = Γ[:,i]·θ.  Both are exact
// between reflections (the flow is linear), so a proposal reads ONE line -- its own record -- and a rejected one writes one sector of it;
// only an accepted reflection (18 % of the proposals on the lattice) visits the neighbours j ∈ G1[i], to bring their sums to t′, add
// Γ[i,j] δθ_i to gd_j and re-bound them.  The draws, the thinning test, the bounds and the queue are the reference's; the floating-point
// values of g differ from a fresh gather in the last bits (sums are advanced, not recomputed), so this kernel reproduces the reference's
// event INDEX sequence, accept/reject outcomes and counters exactly and its times / positions to ~1e-13 (tests: 1e-9; north star: 1e-6),
// where zz_local_spec8_kernel is bit-identical.  The process is not chaotic -- a relative perturbation of 1e-10 of x0 stays 1e-10 after
// 5·10⁴ events, with an identical index sequence (measured on the oracle) -- so the agreement does not decay with the run length.
// Opt-in (pdmp_ensemble_set_gradient_tracking); needs the lattice blob geometry of the 8-event kernel and symmetric Γ.
// The reference's lazy clocks t[j] and positions x[j] at those clocks (src/sfact.jl:211) are rebuilt on demand by zz_track_unpack_kernel
// from the times of the last proposal / accept around j.
// FULL: `adapt` (per-chain bounds c), a target with a mean, and a bounding Γ whose values differ from the target's (two pairs of tracked
// sums); the north-star instantiation has none of these.
template <bool PROF, bool FULL = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void zz_local_track_kernel(ZzRunParams P) {
    constexpr int E = 8;
    constexpr uint32_t SW = 7, PW = 1, KMAX = 5, R_ = 4 + PW + KMAX, WPAD = 58, W2 = WPAD / 2;
    const uint32_t nblk = P.nblk;  // 32-key blocks that hold coordinates (<= S8_NBLK); first-level entries beyond stay +Inf
    const int lane = threadIdx.x;
    const int g = lane >> 3;  // group = event slot
    const int gl = lane & 7;  // lane inside the group
    const int64_t chain = blockIdx.x;
    const int64_t d = P.d;

    extern __shared__ __align__(16) unsigned char smem[];
    double* const LU = reinterpret_cast<double*>(smem + S8_LU);
    double* const SLT = reinterpret_cast<double*>(smem + S8_SLT);
    double* const Lr = reinterpret_cast<double*>(smem + S8_LR);
    double* const LBr = reinterpret_cast<double*>(smem + S8_LBR);
    double* const Mr = reinterpret_cast<double*>(smem + S8_MR);
    uint32_t* const SLB = reinterpret_cast<uint32_t*>(smem + S8_SLB);
    uint32_t* const Z = reinterpret_cast<uint32_t*>(smem + S8_Z);
    double* const bk = reinterpret_cast<double*>(smem + S8_BK);
    uint16_t* const bi = reinterpret_cast<uint16_t*>(smem + S8_BI);
    uint64_t* const LB = reinterpret_cast<uint64_t*>(smem + S8_LB);
    double* const TK = reinterpret_cast<double*>(smem + S8_TK);
    uint32_t* const TB = reinterpret_cast<uint32_t*>(smem + S8_TB);
    double* const SELDT = reinterpret_cast<double*>(smem + S8_SELDT);
    uint32_t* const PR = reinterpret_cast<uint32_t*>(smem + S8_PR);
    double* const pk = reinterpret_cast<double*>(smem + S8_PK) + g * 32;
    uint32_t* const zg = Z + g * 16;
    const uint32_t pk_t = (uint32_t)g & 1u;  // odd groups store the two 16-byte pieces of a lane's chunk swapped: no bank conflicts

    TrRec* rec = reinterpret_cast<TrRec*>(P.rec) + chain * d;
    const bool two_sums = FULL && P.track_two_sums != 0;  // the bounding Γ differs from the target's: (gb, gdb) next to (g, gd)
    double* keys = P.keys + chain * P.dk;
    DevChain* hdr = P.hdr + chain;
    pdmp_event* ev = P.ev ? P.ev + chain * P.trace_cap : nullptr;
    double* cmut = (FULL && P.c_chain) ? (P.c_chain + chain * d) : nullptr;
    const bool adapt = FULL && P.adapt != 0;

    uint32_t status = hdr->c.status;
    if (status == PDMP_CHAIN_BOUND_VIOLATED || status == PDMP_CHAIN_STALLED) return;
    const uint64_t seed = hdr->seed;
    const uint64_t nm0 = hdr->c.ndraw_main, ntrace0 = hdr->c.ntrace;
    uint32_t dnm = 0, dnum = 0, dnacc = 0;
    uint32_t vnacc = 0;  // 1 if the launch ends on a bound violation (acc is bumped before the check)
    double t_last = hdr->c.t_last;
    double t_event = hdr->t_event;
    status = PDMP_CHAIN_OK;

    const double T = P.T;
    const bool stop_before = (P.flags & PDMP_RUN_STOP_BEFORE) != 0;
    const uint32_t trace_room = (P.trace_cap > 0)
                                    ? (uint32_t)(((uint64_t)P.trace_cap > ntrace0) ? ((uint64_t)P.trace_cap - ntrace0) : 0)
                                    : 0xffffffffu;
    const uint32_t common = P.common_tix;

    if (lane == 0) SELDT[0] = 1e-3;  // any positive start: the steering rule finds the scale within a few iterations
    // slot 0 <- the common template, for the whole launch
    if (lane < (int)W2) {
        reinterpret_cast<ulonglong2*>(LB)[lane] = reinterpret_cast<const ulonglong2*>(P.blob + (size_t)common * WPAD)[lane];
    }
    for (uint32_t b = lane; b < nblk; b += 64) {
        const double* kp = keys + (size_t)b * 32;
        double mk = kp[0];
        uint32_t mi = 0;
#pragma unroll 8
        for (int q = 1; q < 32; ++q) {
            const double v = kp[q];
            if (v < mk) {
                mk = v;
                mi = q;
            }
        }
        bk[b] = mk;
        bi[b] = (uint16_t)(b * 32 + mi);
    }
    for (uint32_t b = nblk + lane; b < S8_NBLK; b += 64) {
        bk[b] = PDMP_INF;
        bi[b] = 0;
    }
    LDS_ORDER();

    uint32_t rng_base = 0xffffffffu;
    double ureg = 0.0;  // draw rng_base + lane of the chain's stream
    uint64_t ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t ph_t0 = PROF ? (uint64_t)__builtin_readcyclecounter() : 0;
    uint64_t ph_iters = 0;
#define PHASE(k)                                                          \
    do {                                                                  \
        if (PROF) {                                                       \
            const uint64_t now_ = (uint64_t)__builtin_readcyclecounter(); \
            ph[k] += now_ - ph_t0;                                        \
            ph_t0 = now_;                                                 \
        }                                                                 \
    } while (0)

    bool running = stop_before || (t_event < T);
    PrioTurn prio;
    while (running) {
        prio.step();
        if (dnacc >= trace_room) {
            status = PDMP_CHAIN_TRACE_FULL;
            break;
        }
        if (dnm >= P.count_limit) {  // (32-bit counters of the launch: pause, the host runs again)
            status = PDMP_CHAIN_PAUSED;
            break;
        }
        // ---------------- select the (up to) E smallest block minima, in time order, WITHOUT a tournament per candidate: one
        // wave minimum m, then every first-level entry below the threshold m + sel_dt is a candidate -- four compares and four
        // population counts tell how many there are.  The candidates (at most SEL_CAP, else the threshold is halved) are compacted
        // into LDS by ballot prefix counts, each ranks itself against the others with broadcast reads, and ranks 0..E-1 become
        // the event slots.  Whatever sel_dt is, the slots hold exactly the smallest entries of the queue, so the committed
        // sequence does not depend on it; it is steered towards ~12 candidates per iteration.
        int Esel = 0;
        bool first_inf = false;
        {
            double kk[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) kk[j] = bk[lane + 64 * j];
            const double mloc = min_f64(min_f64(min_f64(kk[0], kk[1]), min_f64(kk[2], kk[3])),
                                        min_f64(min_f64(kk[4], kk[5]), min_f64(kk[6], kk[7])));
            const double mq = wave_min_f64(mloc);
            if (!(mq < PDMP_INF)) {
                first_inf = true;
            } else if (!(stop_before && !(mq < T))) {
                if (lane < (int)SEL_CAP) TK[lane] = PDMP_INF;
                double dt_sel = uniform_f64(SELDT[0]);
                // (the candidate masks are recomputed where they are needed instead of being kept: eight 64-bit masks would
                // crowd the scalar registers)
                // Compaction: entry (lane, j) gets index (candidates of slots < j) + (candidates of slot j in lower lanes).  There is
                // no separate counting pass: the scratch arrays take up to 64 candidates, and a pass that ends with more than
                // SEL_CAP is repeated with half the threshold.
                auto below = [](uint64_t m_) -> uint32_t {
                    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_, 0u));
                };
                double tau;
                uint32_t C;
                for (int tries = 0;; ++tries) {
                    tau = mq + dt_sel;  // (>= mq: the minimum itself always qualifies)
                    if (stop_before && !(tau < T)) tau = pdmp_below(T);
                    const bool pile = tries > 64;  // more than SEL_CAP entries EQUAL to the minimum: one (lowest block) per iteration
                    if (tries >= 64) tau = mq;     // a pile of exactly equal keys: the entries equal to the minimum only
                    uint32_t base = 0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const bool cj_ = kk[j] <= tau;
                        uint64_t Mj = __ballot(cj_);
                        if (pile) Mj = (base == 0 && Mj) ? (Mj & (~Mj + 1)) : 0ull;
                        if (cj_ && ((Mj >> lane) & 1ull)) {
                            const uint32_t ix = base + below(Mj);
                            if (ix < 64u) {
                                TK[ix] = kk[j];
                                TB[ix] = (uint32_t)lane + 64u * j;
                            }
                        }
                        base += (uint32_t)__popcll(Mj);
                    }
                    C = base;
                    if (C <= SEL_CAP) break;
                    dt_sel *= 0.5;
                    LDS_ORDER();
                    if (lane < (int)SEL_CAP) TK[lane] = PDMP_INF;  // (entries past the new count must read +Inf in the ranking)
                }
                LDS_ORDER();
                // rank of candidate n among all (ties by index), on a 16 x 4 grid: lane = 16 * part + n counts the candidates
                // 4 * part .. 4 * part + 3 that precede n; the four partial counts meet in LDS.  Unused entries hold +Inf.
                {
                    const uint32_t n = (uint32_t)lane & 15u, part = (uint32_t)lane >> 4;
                    const double own = TK[n];
                    const double2* T2 = reinterpret_cast<const double2*>(TK + 4 * part);
                    const double2 o01 = T2[0], o23 = T2[1];
                    const uint32_t q = 4 * part;
                    uint32_t pr = 0;
                    pr += (o01.x < own || (o01.x == own && q + 0 < n)) ? 1u : 0u;
                    pr += (o01.y < own || (o01.y == own && q + 1 < n)) ? 1u : 0u;
                    pr += (o23.x < own || (o23.x == own && q + 2 < n)) ? 1u : 0u;
                    pr += (o23.y < own || (o23.y == own && q + 3 < n)) ? 1u : 0u;
                    PR[n * 4 + part] = pr;
                    LDS_ORDER();
                    if ((uint32_t)lane < C) {
                        const uint4 p4 = reinterpret_cast<const uint4*>(PR)[lane];
                        const uint32_t rank = p4.x + p4.y + p4.z + p4.w;
                        if (rank < (uint32_t)E) {
                            SLT[rank] = own;
                            SLB[rank] = TB[lane];
                        }
                    }
                }
                Esel = (C < (uint32_t)E) ? (int)C : E;
                // steer the threshold: ~10 candidates next time
                const double f = (C > 14u) ? 0.8 : (C < 11u) ? ((C < 6u) ? 2.0 : 1.2) : 1.0;
                if (lane == 0) SELDT[0] = dt_sel * f;
            }
        }
        if (Esel == 0) {
            if (first_inf) status = PDMP_CHAIN_STALLED;
            break;
        }
        LDS_ORDER();
        PHASE(0);
        if (PROF) ph_iters += 1;
        bool gvalid = g < Esel;
        const double tp = gvalid ? SLT[g] : PDMP_INF;
        const uint32_t blk = gvalid ? SLB[g] : 0u;
        const uint32_t i = gvalid ? (uint32_t)bi[blk] : 0u;
        const uint32_t tixi = gvalid ? P.tix[i] : common;

        // ---------------- candidate draws (window of 64 draws and their logs in LDS, as in zz_local_spec_kernel)
        if (dnm < rng_base || dnm + E * (1u + KMAX) > rng_base + 64u) {
            rng_base = dnm;
            ureg = pdmp_u01(seed, PDMP_STREAM_MAIN, nm0 + (uint64_t)dnm + (uint64_t)lane);
            LU[lane] = pdmp_log(ureg);
        }
        const uint32_t rng_off = dnm - rng_base;
        // ---------------- blob slots: 0 for the common template, 1 and 2 for the first two events that need another one
        uint32_t slot = 0;
        {
            const bool nc = gvalid && tixi != common;
            const uint64_t ncball = __ballot(nc && gl == 0);
            if (ncball != 0) {
                const uint32_t rank = (uint32_t)__popcll(ncball & ((1ull << (8 * g)) - 1ull));
                if (__popcll(ncball) > 2) {  // the third such event and everything after it wait for the next iteration
                    uint64_t m_ = ncball;
                    m_ &= m_ - 1;
                    m_ &= m_ - 1;
                    const int cut = (__ffsll((unsigned long long)m_) - 1) >> 3;
                    Esel = cut;
                    gvalid = g < Esel;
                }
                if (nc && gvalid) {
                    slot = 1 + rank;
                    const ulonglong2* bsrc = reinterpret_cast<const ulonglong2*>(P.blob + (size_t)tixi * WPAD);
                    ulonglong2* bdst = reinterpret_cast<ulonglong2*>(LB + slot * WPAD);
                    for (uint32_t w = gl; w < W2; w += 8) bdst[w] = bsrc[w];
                }
            }
        }
        const uint64_t* lb = LB + slot * WPAD;
        LDS_ORDER();
        PHASE(1);
        // ---------------- neighbourhood header and member list: positions gl and gl + 8 of S[i]
        // (straight-line: every lane reads its slot's words, the selects below sort out who is a member)
        const uint32_t hw = gvalid ? (uint32_t)lb[0] : 0u;
        const int k = (int)(hw & 0xff), m = (int)((hw >> 8) & 0xff), self = (int)((hw >> 16) & 0xff);
        (void)m;
        const bool memberA = gl < k;  // the zone of an event is G1[i]: nothing else is read or written (k = 0 in an empty slot)
        const uint64_t swa = lb[1 + (gl >> 1)];
        const uint32_t sA = memberA ? i + ((gl & 1) ? (uint32_t)(swa >> 32) : (uint32_t)swa) : 0xffffff00u + (uint32_t)lane;
        const uint32_t sB = 0xffffff40u + (uint32_t)lane;
        constexpr bool memberB = false;
        PHASE(2);
        // All HBM loads of the iteration in ONE straight-line batch (no exec-masked regions: lanes without a record of their own
        // read i's, which coalesces with the group's other readers of it; empty slots read coordinate 0): the wait counters
        // stay exact and nothing here is serialised behind an earlier round trip.
        // own record of i (one 128-byte line; every lane of the group reads the same addresses) and the popped key block
        TrRec* const ri = rec + i;
        double x = ri->x, th = ri->th, tx = ri->tx, I = ri->I;
        const double g_i = ri->g, gd_i = ri->gd, tg_i = ri->tg;
        const uint64_t acc_i = ri->acc;
        const double told_i = ri->t_old, a_i = ri->a, b_i = ri->b;
        double gb_i = 0.0, gdb_i = 0.0;
        if (two_sums) {
            gb_i = ri->gb;
            gdb_i = ri->gdb;
        }
        double kq[4];
        {
            const double2* kp = reinterpret_cast<const double2*>(keys + (size_t)blk * 32 + gl * 4);
            const double2 k01 = kp[0], k23 = kp[1];
            kq[0] = k01.x;
            kq[1] = k01.y;
            kq[2] = k23.x;
            kq[3] = k23.y;
        }
        TrRec* const rsA = rec + (memberA ? sA : i);
        zg[gl] = sA;
        zg[8 + gl] = sB;
        const uint32_t sub = 1 + SW + (uint32_t)gl * R_;
        double cj = __longlong_as_double((long long)lb[sub + 2]);  // (used by lanes gl < k only)
        if (FULL && cmut) cj = cmut[(gl < k) ? sA : i];

        // ---------------- zone conflicts with earlier groups: id spans first, the exact id comparison only for pairs of groups
        // whose spans overlap
        LDS_ORDER();
        uint64_t confball;
        {
            uint32_t lo = memberA ? sA : 0xffffffffu, hi = memberA ? sA : 0u;
            lo = (memberB && sB < lo) ? sB : lo;
            hi = (memberB && sB > hi) ? sB : hi;
            uint32_t o;
            o = dpp_u32<0xB1>(lo);   lo = (o < lo) ? o : lo;
            o = dpp_u32<0x4E>(lo);   lo = (o < lo) ? o : lo;
            o = dpp_u32<0x141>(lo);  lo = (o < lo) ? o : lo;
            o = dpp_u32<0xB1>(hi);   hi = (o > hi) ? o : hi;
            o = dpp_u32<0x4E>(hi);   hi = (o > hi) ? o : hi;
            o = dpp_u32<0x141>(hi);  hi = (o > hi) ? o : hi;
            // A pair of groups (q < g) whose spans overlap is compared exactly by the WHOLE wave: lane L holds id L & 15 of g
            // against ids 4 (L >> 4) .. + 3 of q -- the 256 id pairs in four xor / two min instructions per lane.  Empty
            // positions hold sentinels that equal nothing.
            uint32_t confmask = 0;
            const uint4* Z4 = reinterpret_cast<const uint4*>(Z);
            // lane gl of group g looks at the pair (g, q = gl): one ballot finds all pairs of groups whose spans overlap
            {
                const uint32_t lq = (uint32_t)__builtin_amdgcn_ds_bpermute(32 * gl, (int)lo);  // span of group gl (its lane 0)
                const uint32_t hq = (uint32_t)__builtin_amdgcn_ds_bpermute(32 * gl, (int)hi);
                uint64_t ovb = __ballot(gvalid && gl < g && lo <= hq && lq <= hi);
                while (ovb != 0) {
                    const int bit = __ffsll((unsigned long long)ovb) - 1;
                    const int gsel = bit >> 3, q = bit & 7;
                    ovb &= ovb - 1;
                    const uint32_t idg = Z[gsel * 16 + (lane & 15)];
                    const uint4 zq = Z4[q * 4 + (lane >> 4)];
                    const uint32_t mn = umin3(idg ^ zq.x, idg ^ zq.y, umin3(idg ^ zq.z, idg ^ zq.w, 0xffffffffu));
                    if (__ballot(mn == 0u) != 0) confmask |= 1u << gsel;
                }
            }
            confball = confmask;
        }
        PHASE(3);

        // ---------------- rates: the tracked sums stand in for smove_forward!(G, i, ...) + idot (src/sfact.jl:82,116-119): g_i(t′) = g_i + gd_i (t′ − tg_i)
        double l, lbound;
        const double g_now = g_i + gd_i * (tp - tg_i);
        const double gb_now = two_sums ? (gb_i + gdb_i * (tp - tg_i)) : g_now;
        {
            double gr = g_now;
            if (FULL && P.tb.gmu_t) gr = gr - P.tb.gmu_t[i];
            l = pos_part(gr * th);
            lbound = pos_part(a_i + b_i * (tp - told_i));
            if (gl == 0) {
                Lr[g] = l;
                LBr[g] = lbound;
            }
        }
        LDS_ORDER();
        // ---------------- accept chain in time order.  Lane o evaluates every event's test for the draw at offset o; the ballots
        // are then walked on the scalar unit: event r reads its bit at the offset the earlier outcomes imply.
        // The offsets (each <= 48) travel packed six bits apiece in one 64-bit scalar: offset after r events = bits 6r .. 6r+5.
        uint32_t accbits = 0;
        uint64_t offpack = 0;
        {
            const double coin = bperm_f64(ureg, (rng_off + (uint32_t)lane) & 63u);
            uint32_t off = 0;
#pragma unroll
            for (int r = 0; r < E; ++r) {  // slots >= Esel hold stale rates: their bits are masked off below, their offsets unused
                const uint64_t am_r = __ballot(coin * LBr[r] < Lr[r]);  // :121
                const uint32_t a_r = (uint32_t)(am_r >> off) & 1u;
                const uint32_t k_r = readlane_u32((uint32_t)k, 8 * r);
                off += a_r ? (1u + k_r) : 2u;
                off = (off < 63u) ? off : 63u;  // (only stale slots can run past the window; keeps the shifts defined)
                accbits |= a_r << r;
                offpack |= (uint64_t)off << (6 * (r + 1));
            }
            accbits &= (1u << Esel) - 1u;
        }
        const uint32_t myoff = (uint32_t)(offpack >> (6 * g)) & 63u;
        const bool accept = gvalid && ((accbits >> g) & 1u) != 0;
        const bool violated = accept && (l >= lbound);  // :123
        PHASE(4);

        // ---------------- accept: reflect!(i) (:130) changes θ_i by δ; every j in G1[i] brings its sums to t′, takes Γ[i,j] δ into its
        // velocity sum and is re-bounded (:131-135); reject: i is re-bounded from its own sums (:137-140)
        const bool active = gvalid && (accept ? (gl < k) : (gl == self));
        double key = PDMP_INF, a = 0.0, b = 0.0;
        double gj = g_now, gdj = gd_i, gbj = gb_now, gdbj = two_sums ? gdb_i : gd_i, thj = th;
        if (accept && gl == self) {  // event(i, t, x, θ, F) needs x_i at t′ (src/sfact.jl:50-52): the position is brought up on accepts only
            const double dtx = tp - tx;
            const double xn = x + th * dtx;
            I = I + dtx * ((x + xn) * 0.5);
            x = xn;
            tx = tp;
        }
        if (active) {
            const double th_new_i = accept ? -th : th;
            const double delta = accept ? (th_new_i - th) : 0.0;
            if (accept && gl != self) {
                thj = rsA->th;
                const double gj0 = rsA->g, gdj0 = rsA->gd, tgj = rsA->tg;
                gj = gj0 + gdj0 * (tp - tgj);
                gdj = gdj0;
                if (two_sums) {
                    const double gbj0 = rsA->gb, gdbj0 = rsA->gdb;
                    gbj = gbj0 + gdbj0 * (tp - tgj);
                    gdbj = gdbj0;
                } else {
                    gbj = gj;
                    gdbj = gdj;
                }
            } else {
                thj = th_new_i;
            }
            const double gmu = __longlong_as_double((long long)lb[sub + 1]);
            if (accept) {
                // Γt[i, j] = Γt[j, i] (the stored column of i; the precision matrix is symmetric, checked on the host) and Γ[i, j] of
                // the bounding matrix: the entry of column j that sits at i's position
                const double ct = __longlong_as_double((long long)lb[sub + 0]);
                gdj += ct * delta;
                if (two_sums) {
                    const int kj = (int)(lb[sub + 3] & 0xff);
                    const uint64_t pw = lb[sub + 4];
                    double cbv = 0.0;
#pragma unroll
                    for (int q = 0; q < (int)KMAX; ++q) {
                        const int ps = (int)((pw >> (8 * q)) & 0xff);
                        if (q < kj && ps == self) cbv = __longlong_as_double((long long)lb[sub + 4 + PW + q]);
                    }
                    gdbj += cbv * delta;
                } else {
                    gdbj = gdj;
                }
            }
            if (FULL && violated && gl == self) cj *= P.factor;  // adapt!(c, i, factor), :127 (stored at commit)
            a = cj + (gbj - gmu) * thj;
            b = cj / 100 + thj * gdbj;
            const double L = LU[(rng_off + myoff + 1u + (accept ? (uint32_t)gl : 0u)) & 63u];
            key = tp + dev_poisson_time_L(a, b, L);
        }
        LDS_ORDER();
        // the patched copy of the popped key block goes where the zone ids were: all their readers are done
        // (the four 16-byte pieces of a lane's 64-byte chunk are stored in the order piece ^ pk_t, pk_t = 0..3 over the four lanes
        // of a quarter wave that would otherwise share their banks: b128 accesses without bank conflicts)
        {
            double2* pk2 = reinterpret_cast<double2*>(pk + gl * 4);
            pk2[0 ^ pk_t] = make_double2(kq[0], kq[1]);
            pk2[1 ^ pk_t] = make_double2(kq[2], kq[3]);
        }
        LDS_ORDER();
        if (active && (sA >> 5) == blk) {
            const uint32_t e_ = sA & 31u;
            pk[(e_ & ~3u) + ((((e_ & 3u) >> 1) ^ pk_t) << 1) + (e_ & 1u)] = key;
        }
        LDS_ORDER();
        PHASE(5);
        // ---------------- patched minimum of the popped block, and everything this event could expose
        double rowmin;
        uint32_t cand;
        int wl2;
        {
            const double2* pk2 = reinterpret_cast<const double2*>(pk + gl * 4);
            const double2 p01 = pk2[0 ^ pk_t], p23 = pk2[1 ^ pk_t];
            double lm = p01.x;
            uint32_t li = 0;
#define PMIN(v, idx)    \
    do {                \
        if ((v) < lm) { \
            lm = (v);   \
            li = (idx); \
        }               \
    } while (0)
            PMIN(p01.y, 1);
            PMIN(p23.x, 2);
            PMIN(p23.y, 3);
#undef PMIN
            cand = blk * 32u + (uint32_t)gl * 4u + li;
            rowmin = grp8_min_f64(lm);
            const uint64_t winball = __ballot(gvalid && lm == rowmin);
            wl2 = __ffs((unsigned)((winball >> (8 * g)) & 0xffu)) - 1;
        }
        const double keymin = grp8_min_f64(key);
        const double expose = min_f64(rowmin, keymin);
        if (gl == 0) Mr[g] = expose;
        LDS_ORDER();
        // ---------------- validate: event g commits iff all earlier ones do, its zone is disjoint from theirs, and nothing they
        // produce or expose comes before it
        uint32_t Rc;
        uint32_t nacc_c;
        int vsel = -1;  // the event that violates its bound, if it is the chain's next one
        {
            double pref = PDMP_INF;
#pragma unroll
            for (int q = 0; q < E - 1; ++q) {
                const double mq = Mr[q];
                pref = (q < g) ? min_f64(pref, mq) : pref;
            }
            const bool confg = ((confball >> g) & 1ull) != 0;
            const bool okg = gvalid && ((g == 0) || (!confg && pref > tp));
            const bool vstop = violated && !adapt;  // reference: error(...), :124 -> the event is not committed
            const uint64_t okball = __ballot(okg && !vstop && gl == 0);
            const uint64_t vball = __ballot(okg && vstop && gl == 0);
            const uint64_t accball = __ballot(accept && gl == 0);
            // length of the run of committable slots from slot 0 (one bit per slot at bit 8 r): first zero among those bits
            const uint64_t gap = ~okball & 0x0101010101010101ull;
            const uint32_t r_ok = gap ? (uint32_t)((__ffsll((unsigned long long)gap) - 1) >> 3) : (uint32_t)E;
            Rc = 0;
            nacc_c = 0;
            bool stopped = false;
            // the usual case needs no walk: the slice mode stops on time alone, and the trace has room for every accepted slot
            const uint32_t nacc_all = (uint32_t)__popcll(accball & ((r_ok < 8u) ? ((1ull << (8 * r_ok)) - 1ull) : ~0ull));
            const bool plainrun = stop_before && !(P.trace_cap > 0 && dnacc + nacc_all >= trace_room);
            if (plainrun) {
                Rc = r_ok;
                nacc_c = nacc_all;
            }
            for (uint32_t r = 0; !plainrun && r < r_ok && !stopped; ++r) {
                Rc = r + 1;
                if ((accball >> (8 * r)) & 1ull) {
                    nacc_c += 1;
                    if (dnacc + nacc_c >= trace_room && P.trace_cap > 0) {
                        status = PDMP_CHAIN_TRACE_FULL;
                        stopped = true;
                    }
                    if (!stop_before && !(uniform_f64(SLT[r]) < T)) {
                        running = false;
                        stopped = true;
                    }
                }
            }
            if (!stopped && r_ok < (uint32_t)E && ((vball >> (8 * r_ok)) & 1ull)) {
                status = PDMP_CHAIN_BOUND_VIOLATED;
                vsel = (int)r_ok;
            }
        }
        PHASE(6);

        // ---------------- commit the valid prefix
        const bool commit = gvalid && (uint32_t)g < Rc;
        const uint64_t accball2 = __ballot(commit && accept && gl == 0);
        if (commit) {
            if (active) {
                if (accept) {
                    rsA->g = gj;
                    rsA->gd = gdj;
                    rsA->tg = tp;
                    if (two_sums) {
                        rsA->gb = gbj;
                        rsA->gdb = gdbj;
                    }
                }
                rsA->a = a;
                rsA->b = b;
                rsA->t_old = tp;
                keys[sA] = key;
                if (FULL && violated && gl == self) cmut[sA] = cj;
                if (gl == self) {
                    ri->tprop = tp;  // every proposal of i moves G[i] to t′ in the reference: kept to rebuild its clocks (zz_track_unpack_kernel)
                    if (accept) {
                        ri->x = x;
                        ri->th = -th;
                        ri->tx = tx;
                        ri->I = I;
                        ri->acc = acc_i + 1;
                        ri->tacc = tp;   // an accepted event also moves G2[i]
                    }
                }
            }
            if (gl == wl2) {
                bk[blk] = rowmin;
                bi[blk] = (uint16_t)cand;
            }
            if (accept && gl == self && ev) {
                const uint32_t rank = (uint32_t)__popcll(accball2 & ((1ull << (8 * g)) - 1ull));
                pdmp_event e;
                e.t = tp;
                e.i = (int64_t)i;
                e.x = x;
                e.theta = -th;
                ev[ntrace0 + dnacc + rank] = e;
            }
        }
        LDS_ORDER();
        PHASE(7);
        // ---------------- level-1 updates for re-bounded neighbours living in other blocks.  The final entry of a block is the
        // smallest (key, coordinate) among its old entry and the new keys, whatever the order -- so when no two of these lanes aim
        // at one block (checked through a small claim table) and none has to rescan, every lane updates its block by itself, in
        // one LDS round trip for all of them; otherwise the updates are made one by one in event order.
        const bool upd = commit && accept && gl < k && (sA >> 5) != blk;
        if (__ballot(upd) != 0) {
            uint8_t* const CL = reinterpret_cast<uint8_t*>(smem + S8_CL);
            LDS_ORDER();
            const uint32_t bjv = upd ? (sA >> 5) : 0u;
            const double curv = bk[bjv];
            const uint32_t civ = bi[bjv];
            const bool lower = upd && (key < curv || (key == curv && sA < civ));
            const bool resc = upd && !lower && civ == sA;
            if (lower) CL[bjv & 63u] = (uint8_t)lane;
            LDS_ORDER();
            const bool lost = lower && CL[bjv & 63u] != (uint8_t)lane;
            if (__ballot(lost || resc) == 0) {
                if (lower) {
                    bk[bjv] = key;
                    bi[bjv] = (uint16_t)sA;
                }
            } else {
            for (uint32_t r = 0; r < Rc; ++r) {
                if (!((accball2 >> (8 * r)) & 1ull)) continue;
                const uint32_t own = uniform_u32(SLB[r]);
                const int kr = (int)readlane_u32((uint32_t)k, 8 * (int)r);
                for (int jj = 0; jj < kr; ++jj) {
                    const uint32_t j = readlane_u32(sA, 8 * (int)r + jj);
                    if ((j >> 5) == own) continue;
                    // first-level entry of j's 32-key block: a lower key replaces it; if j WAS the entry and grew, the block is rescanned
                    const double kj = readlane_f64(key, 8 * (int)r + jj);
                    const uint32_t bj = j >> 5;
                    LDS_ORDER();
                    const double cur = bk[bj];
                    const uint32_t ci = bi[bj];
                    if (kj < cur || (kj == cur && j < ci)) {
                        if (lane == 0) {
                            bk[bj] = kj;
                            bi[bj] = (uint16_t)j;
                        }
                    } else if (ci == j) {
                        const double kv = __hip_atomic_load(keys + (size_t)bj * 32 + (lane & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const double mn = wave_min_f64(kv);
                        const uint64_t bl = __ballot(kv == mn);
                        const int arg = bl ? (__ffsll((unsigned long long)bl) - 1) : 0;
                        if (lane == 0) {
                            bk[bj] = mn;
                            bi[bj] = (uint16_t)(bj * 32 + (uint32_t)arg);
                        }
                    }
                }
            }
            }
        }
        PHASE(8);
        // ---------------- the violating proposal itself (reference: counted, G[i] moved, acc bumped -- then error(...), :120-124):
        // what zz_local_run_kernel and the oracle leave behind
        if (vsel >= 0) {
            if (g == vsel && gl == self) ri->tprop = uniform_f64(SLT[vsel]);
            dnum += 1;
            vnacc = 1;
            dnm += ((uint32_t)(offpack >> (6 * vsel)) & 63u) + 1u - ((uint32_t)(offpack >> (6 * Rc)) & 63u);
        }
        // ---------------- counters
        if (Rc > 0) {
            dnum += Rc;
            dnacc += nacc_c;
            dnm += (uint32_t)(offpack >> (6 * Rc)) & 63u;
            t_last = uniform_f64(SLT[Rc - 1]);
            if (accball2) t_event = uniform_f64(SLT[(63 - __builtin_clzll(accball2)) >> 3]);
        }
        if (vsel >= 0) t_last = uniform_f64(SLT[vsel]);  // the violating event's time is the chain's current time
        if (status != PDMP_CHAIN_OK) break;
        LDS_ORDER();
    }

    if (PROF && P.dbg && chain == 0 && lane == 0) {
        for (int q = 0; q < 10; ++q) P.dbg[q] = (double)ph[q];
        P.dbg[10] = (double)ph_iters;
    }
#undef PHASE
    if (lane == 0) {
        hdr->c.t_last = t_last;
        hdr->t_event = t_event;
        hdr->c.num += dnum;
        hdr->c.nacc += dnacc + vnacc;
        hdr->c.ntrace = ntrace0 + dnacc;
        hdr->c.nevents += dnacc;
        hdr->c.ndraw_main = nm0 + dnm;
        hdr->c.status = status;
    }
}


// ------------------------------------------------------------------------------------------ speculative sticky loop
//
// zz_local_spec_kernel's scheme (up to 4 events of one chain per iteration, one per 16-lane row, exact validation, commit of
// the valid prefix) for sspdmp_inner! (src/ss_fact.jl:78-157).  What changes per event:
//   type      freeze (f[i], :87-107), thaw (x[i] == 0 && θ[i] == 0, :108-123) or reflection proposal (:124-152); known from i's own
//             record, so every record of S[i] is fetched up front (no lazy G2)
//   moves     only coordinates with θ != 0 (ssmove_forward!, :25-45)
//   draws     freeze: the thaw clock (1) + one per re-bounded neighbour unless strong_upperbounds; thaw: the reversible coin
//             (0/1) + one per non-frozen member of G1[i]; proposal: coin + one per non-frozen member on accept, coin + 1 on reject.
//             The offsets of later events follow from the types and accept outcomes of the earlier ones, resolved in the same
//             per-lane chain walk; a re-bounded lane uses draw (offset + head + its rank among the re-bounded lanes)
//   keys      queue_time! (:54-66): min(reflection proposal, hitting time of 0), the winner recorded in the record's flag word
//   events    freeze, thaw and accepted reflection are all trace events (:154); (acc, num) are scalars and are reset to 0 by
//             an adapted bound violation (:134)
// Validation and commit are those of the ZigZag kernel (zone disjointness + exposure bound), so the committed sequence is
// bit-identical to zz_sticky_run_kernel and to the oracle.
constexpr uint32_t SPS_TY = 4576;   // [4] u32 event type: 0 proposal, 1 freeze, 2 thaw
constexpr uint32_t SPS_NR = 4592;   // [4] u32 re-bounded lanes if the event "happens" (accept / freeze / thaw)
constexpr uint32_t SPS_ND = 4608;   // [4] u32 head draws (coin / thaw clock / reversible coin)
constexpr uint32_t SPS_LB = 4640;   // [4][Wpad] u64 blobs, then bk[nblk_pad] f64, bi[nblk_pad] u32

size_t zz_sticky_spec_lds_bytes(uint32_t nblk_pad, uint32_t blob_w_pad) {
    return (size_t)SPS_LB + (size_t)4 * blob_w_pad * 8 + (size_t)nblk_pad * 8 + (size_t)nblk_pad * 4;
}

// PLAIN: adapt = false, reversible = false, strong_upperbounds = false, target without a mean shift, and the blob geometry of the
// 4-neighbour lattice (config C5) as compile-time facts
template <int NE, bool PLAIN = false>
__global__ __launch_bounds__(64) void zz_sticky_spec_kernel(ZzRunParams P_in) {
    ZzRunParams P = P_in;
    if constexpr (PLAIN) {
        P.adapt = 0;
        P.c_chain = nullptr;
        P.tb.gmu_t = nullptr;
        P.reversible = 0;
        P.strong_upperbounds = 0;
        P.blob_w_pad = 58;
    }
    constexpr int E = 4;
    const int lane = threadIdx.x;
    const int g = lane >> 4;
    const int gl = lane & 15;
    const int64_t chain = blockIdx.x;
    const int64_t d = P.d;
    const uint32_t nblk = P.nblk;
    const uint32_t W2 = P.blob_w_pad >> 1, SW = PLAIN ? 7u : P.blob_sw, PW = PLAIN ? 1u : P.blob_pw, KMAX = PLAIN ? 5u : P.blob_kmax;
    const uint32_t R_ = 4 + PW + KMAX;

    extern __shared__ __align__(16) unsigned char smem[];
    double* const U = reinterpret_cast<double*>(smem + SP_U);
    double* const LU = reinterpret_cast<double*>(smem + SP_LU);
    double* const SLT = reinterpret_cast<double*>(smem + SP_SLT);
    double* const SLH = reinterpret_cast<double*>(smem + SP_SLH);
    double* const Lr = reinterpret_cast<double*>(smem + SP_LR);
    double* const LBr = reinterpret_cast<double*>(smem + SP_LBR);
    double* const Mr = reinterpret_cast<double*>(smem + SP_MR);
    uint32_t* const Z = reinterpret_cast<uint32_t*>(smem + SP_Z);
    uint32_t* const SLB = reinterpret_cast<uint32_t*>(smem + SP_SLB);
    uint32_t* const OFR = reinterpret_cast<uint32_t*>(smem + SP_OFR);
    uint32_t* const TY = reinterpret_cast<uint32_t*>(smem + SPS_TY);
    uint32_t* const NR = reinterpret_cast<uint32_t*>(smem + SPS_NR);
    uint32_t* const ND = reinterpret_cast<uint32_t*>(smem + SPS_ND);
    double* const bk = reinterpret_cast<double*>(smem + SPS_LB + (size_t)4 * P.blob_w_pad * 8);
    uint32_t* const bi = reinterpret_cast<uint32_t*>(bk + P.nblk_pad);
    double* const sx = reinterpret_cast<double*>(smem + SP_SX) + g * 16;
    double* const sth = reinterpret_cast<double*>(smem + SP_STH) + g * 16;
    double* const pk = reinterpret_cast<double*>(smem + SP_PK) + g * 64;
    uint64_t* const lb = reinterpret_cast<uint64_t*>(smem + SPS_LB) + (size_t)g * P.blob_w_pad;

    ZzRec* rec = P.rec + chain * d;
    double* keys = P.keys + chain * P.dk;
    double* thf = P.thf + chain * d;
    DevChain* hdr = P.hdr + chain;
    pdmp_event* ev = P.ev ? P.ev + chain * P.trace_cap : nullptr;
    double* cmut = P.c_chain ? (P.c_chain + chain * d) : nullptr;

    uint32_t status = hdr->c.status;
    if (status == PDMP_CHAIN_BOUND_VIOLATED || status == PDMP_CHAIN_STALLED) return;
    const uint64_t seed = hdr->seed;
    const uint64_t nm0 = hdr->c.ndraw_main, ntrace0 = hdr->c.ntrace;
    uint64_t num = hdr->c.num, nacc = hdr->c.nacc;  // scalars with the reset quirk (:134): kept absolute
    uint32_t dnm = 0, dnev = 0;                     // draws / trace events of this launch
    double t_last = hdr->c.t_last;
    double t_event = hdr->t_event;
    status = PDMP_CHAIN_OK;

    const double T = P.T;
    const bool stop_before = (P.flags & PDMP_RUN_STOP_BEFORE) != 0;
    const bool adapt = P.adapt != 0;
    const bool strong = P.strong_upperbounds != 0;
    const bool reversible = P.reversible != 0;
    const uint32_t trace_room = (P.trace_cap > 0)
                                    ? (uint32_t)(((uint64_t)P.trace_cap > ntrace0) ? ((uint64_t)P.trace_cap - ntrace0) : 0)
                                    : 0xffffffffu;

    for (uint32_t b = lane; b < nblk; b += 64) {
        const double* kp = keys + (size_t)b * 64;
        double mk = kp[0];
        uint32_t mi = 0;
#pragma unroll 8
        for (int q = 1; q < 64; ++q) {
            const double v = kp[q];
            if (v < mk) {
                mk = v;
                mi = q;
            }
        }
        bk[b] = mk;
        bi[b] = b * 64 + mi;
    }
    LDS_ORDER();

    uint32_t rng_base = 0xffffffffu;
    bool running = stop_before || (t_event < T);
    PrioTurn prio;
    while (running) {
        prio.step();
        if (dnev >= trace_room) {
            status = PDMP_CHAIN_TRACE_FULL;
            break;
        }
        if (dnm >= P.count_limit) {  // (32-bit counters of the launch: pause, the host runs again)
            status = PDMP_CHAIN_PAUSED;
            break;
        }
        // ---------------- select up to E candidate events (spec_select)
        bool first_inf;
        const int Esel = spec_select<NE, E>(bk, nblk, lane, stop_before, T, SLT, SLH, SLB, first_inf);
        if (Esel == 0) {
            if (first_inf) status = PDMP_CHAIN_STALLED;
            break;
        }
        LDS_ORDER();
        const bool gvalid = g < Esel;
        const double tp = gvalid ? SLT[g] : PDMP_INF;
        const uint32_t blk = gvalid ? SLB[g] : 0u;
        const double hidg = gvalid ? SLH[g] : PDMP_INF;
        const uint32_t i = gvalid ? bi[blk] : 0u;

        {
            const uint32_t tixi = gvalid ? P.tix[i] : 0u;
            const ulonglong2* bsrc = reinterpret_cast<const ulonglong2*>(P.blob + (size_t)tixi * P.blob_w_pad);
            ulonglong2* bdst = reinterpret_cast<ulonglong2*>(lb);
            if (gvalid) {
                for (uint32_t w = gl; w < W2; w += 16) bdst[w] = bsrc[w];
            }
        }
        if (dnm < rng_base || dnm + E * (1u + KMAX) > rng_base + 64u) {
            rng_base = dnm;
            const double u = pdmp_u01(seed, PDMP_STREAM_MAIN, nm0 + (uint64_t)dnm + (uint64_t)lane);
            U[lane] = u;
            LU[lane] = pdmp_log(u);
        }
        const uint32_t rng_off = dnm - rng_base;
        LDS_ORDER();
        int k = 0, m = 0, self = 0, kjmax = 0;
        uint32_t s = 0xffffff00u + (uint32_t)lane;
        if (gvalid) {
            const uint64_t hw = lb[0];
            k = (int)(hw & 0xff);
            m = (int)((hw >> 8) & 0xff);
            self = (int)((hw >> 16) & 0xff);
            kjmax = (int)((hw >> 24) & 0xff);
            if (gl < m) {
                const uint64_t sw = lb[1 + (gl >> 1)];
                s = i + ((gl & 1) ? (uint32_t)(sw >> 32) : (uint32_t)sw);
            }
        }
        const bool member = gvalid && gl < m;
        ZzRec* rs = rec + (member ? s : i);
        double x = 0.0, th = 0.0, t = 0.0, I = 0.0;
        if (member) {
            x = rs->x;
            th = rs->th;
            t = rs->t;
            I = rs->I;
        }
        const ZzRec* ri = rec + i;
        double told_i = 0.0, a_i = 0.0, b_i = 0.0, thf_i = 0.0, kappa_i = 1.0;
        uint64_t flag_i = 0;
        double kq[4] = {PDMP_INF, PDMP_INF, PDMP_INF, PDMP_INF};
        if (gvalid) {
            told_i = ri->t_old;
            a_i = ri->a;
            b_i = ri->b;
            flag_i = ri->acc;
            thf_i = thf[i];
            kappa_i = P.kappa[i];
            const double2* kp = reinterpret_cast<const double2*>(keys + (size_t)blk * 64 + gl * 4);
            const double2 k01 = kp[0], k23 = kp[1];
            kq[0] = k01.x;
            kq[1] = k01.y;
            kq[2] = k23.x;
            kq[3] = k23.y;
        }
        Z[lane] = s;
        const uint32_t sub = 1 + SW + (uint32_t)gl * R_;
        double cj = 0.0;
        if (gvalid && gl < k) cj = cmut ? cmut[s] : __longlong_as_double((long long)lb[sub + 2]);
        // raw (x, θ) of the members: the event type needs i's own pair
        if (member) {
            sx[gl] = x;
            sth[gl] = th;
        }

        // ---------------- zone conflicts with earlier groups
        LDS_ORDER();
        const uint64_t confball = __ballot(spec_zone_conflict<E>(Z, s, g, member));

        // ---------------- event type, moves that do not depend on a draw, gradient
        const double x_i0 = gvalid ? sx[self] : 1.0, th_i0 = gvalid ? sth[self] : 1.0;
        const bool is_freeze = gvalid && (flag_i != 0);
        const bool is_thaw = gvalid && !is_freeze && (x_i0 == 0 && th_i0 == 0);
        const bool is_prop = gvalid && !is_freeze && !is_thaw;
        auto move_lane = [&]() {  // t[i], x[i] = t′, x[i] + θ[i]*(t′ - t[i])
            const double dt = tp - t;
            const double xn = x + th * dt;
            I = I + dt * ((x + xn) * 0.5);
            x = xn;
            t = tp;
        };
        bool xerr = false;        // freeze with |x[i]| > 1e-8: the reference errors (:89-91)
        double thf_new = 0.0;     // value thf[i] takes at commit (freeze: saved speed; thaw: 0)
        if (is_freeze) {
            if (gl == self) {
                move_lane();  // smove_forward!(i, ...), :88
                xerr = fabs(x) > 1e-8;
                thf_new = th;  // θf[i], θ[i] = θ[i], 0.0, :93
                x = 0.0 * th;  // x[i] = -0*θ[i], :92
                th = 0.0;
            }
            if (!strong && member && th != 0.0) move_lane();  // ssmove_forward!(G, i) and (G2, i), :98-99
        } else if (is_thaw) {
            if (gl == self) {
                t = tp;        // :109
                th = thf_i;    // θ[i], θf[i] = θf[i], 0.0, :110 (sign under `reversible` resolved with the draw below)
            }
            if (member && th != 0.0) move_lane();  // :115-116 (i itself: x + θ*0)
        } else if (is_prop) {
            if (gl < k && th != 0.0) move_lane();  // :125
        }
        const uint64_t xerrball = __ballot(xerr);
        LDS_ORDER();
        if (member) {
            sx[gl] = x;
            sth[gl] = th;
        }
        LDS_ORDER();
        // re-bounded lanes if the event happens: non-frozen members of G1[i] (freeze: i itself now has θ = 0; nothing if strong)
        const bool reb_if = gvalid && gl < k && th != 0.0 && !(is_freeze && strong);
        const uint64_t rebball = __ballot(reb_if);
        const uint32_t rowmask = (uint32_t)((rebball >> (16 * g)) & 0xffffull);
        {
            double gr = 0.0;
            for (uint32_t p = 0; p < KMAX; ++p) {
                if ((int)p < k) gr += __longlong_as_double((long long)lb[1 + SW + p * R_]) * sx[p];
            }
            if (gvalid) {
                if (P.tb.gmu_t) gr = gr - P.tb.gmu_t[i];
                const double th_i = sth[self];
                const double l = pos_part(gr * th_i);
                const double lbound = pos_part(a_i + b_i * (tp - told_i));  // :128
                if (gl == 0) {
                    Lr[g] = l;
                    LBr[g] = lbound;
                    TY[g] = is_freeze ? 1u : (is_thaw ? 2u : 0u);
                    NR[g] = (uint32_t)__popc(rowmask);
                    ND[g] = is_freeze ? 1u : (is_thaw ? (reversible ? 1u : 0u) : 1u);
                }
            }
        }
        LDS_ORDER();
        // ---------------- chain walk in time order: draw offsets, accept outcomes
        uint32_t accept_u = 0, violated_u = 0, myoff = 0;
        {
            uint32_t off = 0;
#pragma unroll
            for (int r = 0; r < E; ++r) {
                if (g == r) myoff = off;
                if (lane == 0) OFR[r] = off;
                if (r < Esel) {
                    const uint32_t ty = TY[r];
                    const double coin = U[rng_off + off];
                    const double l = Lr[r], lbound = LBr[r];
                    const uint32_t a_r = (ty != 0u) ? 1u : ((coin * lbound < l) ? 1u : 0u);  // :130
                    const uint32_t v_r = (ty == 0u && a_r && (l > lbound)) ? 1u : 0u;      // :132
                    off += a_r ? (ND[r] + NR[r]) : 2u;
                    if (g == r) {
                        accept_u = a_r;
                        violated_u = v_r;
                    }
                }
            }
            if (lane == 0) OFR[E] = off;
        }
        const bool happens = gvalid && accept_u != 0;  // freeze, thaw or accepted reflection: a trace event
        const bool violated = violated_u != 0;

        if (happens && is_prop) {
            if (gl >= k && gl < m && th != 0.0) move_lane();  // ssmove_forward!(G2, i), :138
            if (gl == self) th = -th;                         // reflect!, :139
        }
        if (happens && is_thaw && reversible && gl == self) th *= (U[rng_off + myoff] < 0.5) ? -1.0 : 1.0;  // :111-113
        if (member) {
            sx[gl] = x;
            sth[gl] = th;
        }
        {
            double2* pk2 = reinterpret_cast<double2*>(pk + gl * 4);
            pk2[0] = make_double2(kq[0], kq[1]);
            pk2[1] = make_double2(kq[2], kq[3]);
        }
        LDS_ORDER();
        // ---------------- ab + queue_time! (:54-66) for the re-bound set
        const bool active = gvalid && (happens ? reb_if : (gl == self));
        double key = PDMP_INF, a = 0.0, b = 0.0;
        uint32_t fzflag = 0;
        if (active) {
            const double gmu = __longlong_as_double((long long)lb[sub + 1]);
            const int kj = (int)(lb[sub + 3] & 0xff);
            double gx = 0.0, gt = 0.0;
            for (int base = 0; base < (PLAIN ? 1 : kjmax); base += 8) {
                const uint64_t pw = lb[sub + 4 + (base >> 3)];
#pragma unroll
                for (int q = 0; q < (PLAIN ? 5 : 8); ++q) {
                    const int pp = base + q;
                    if (pp < kj) {
                        const double v = __longlong_as_double((long long)lb[sub + 4 + PW + pp]);
                        const int ps = (int)((pw >> (8 * q)) & 0xff);
                        gx += v * sx[ps];
                        gt += v * sth[ps];
                    }
                }
            }
            if (violated && gl == self) cj *= P.factor;  // adapt!(c, i, factor), :135
            a = cj + (gx - gmu) * th;
            b = cj / 100 + th * gt;
            const uint32_t rank = happens ? (uint32_t)__popc(rowmask & ((1u << gl) - 1u)) : 0u;
            const uint32_t head = happens ? ((is_prop || is_freeze) ? 1u : (reversible ? 1u : 0u)) : 1u;
            const double L = LU[rng_off + myoff + head + rank];
            const double trefl = dev_poisson_time_L(a, b, L);
            const double tfreeze = (th * x >= 0) ? PDMP_INF : (-x / th);  // freezing_time, :10-16
            const bool fz = tfreeze <= trefl;                              // :57
            fzflag = fz ? 1u : 0u;
            key = t + (fz ? tfreeze : trefl);
        }
        if (is_freeze && gl == self) key = tp - LU[rng_off + myoff] / kappa_i;  // Q[i] = t[i] - log(rand())/κ[i], :96
        const bool newkey = active || (is_freeze && gl == self);
        if (newkey && (s >> 6) == blk) pk[s & 63] = key;
        LDS_ORDER();
        // ---------------- patched minimum of the popped block, exposure
        double rowmin, candmin;
        uint32_t cand;
        spec_patched_min(pk, gl, blk, rowmin, candmin, cand);
        const uint64_t winball = __ballot(gvalid && candmin == rowmin);
        const int wl2 = __ffs((unsigned)((winball >> (16 * g)) & 0xffffu)) - 1;
        const double keymin = row_min_f64(newkey ? key : PDMP_INF);
        const double expose = min_f64(min_f64(rowmin, keymin), hidg);
        if (gl == 0) Mr[g] = expose;
        LDS_ORDER();
        // ---------------- validate
        uint32_t Rc;
        uint32_t happb_c;  // bit r: committed event r is a trace event
        int vsel = -1;       // the event that stops the chain with an error, if it is the chain's next one
        bool vprop = false;  // ... and it is a proposal that violates its bound (not the x[i] != 0 check of a freeze)
        {
            const double m0 = Mr[0], m1 = Mr[1], m2 = Mr[2];
            const double pref = (g == 0) ? PDMP_INF : (g == 1) ? m0 : (g == 2) ? min_f64(m0, m1) : min_f64(min_f64(m0, m1), m2);
            const bool confg = ((confball >> (16 * g)) & 0xffffull) != 0;
            const bool okg = gvalid && ((g == 0) || (!confg && pref > tp));
            const bool xerrg = ((xerrball >> (16 * g)) & 0xffffull) != 0;
            const bool vstop = (violated && !adapt) || xerrg;  // reference: error(...), :90, :133
            const uint64_t okball = __ballot(okg && !vstop && gl == 0);
            const uint64_t vball = __ballot(okg && vstop && gl == 0);
            const uint64_t happball = __ballot(happens && gl == 0);
            auto bits4 = [](uint64_t m_) -> uint32_t {
                return (uint32_t)((m_ & 1ull) | ((m_ >> 15) & 2ull) | ((m_ >> 30) & 4ull) | ((m_ >> 45) & 8ull));
            };
            const uint32_t okb = bits4(okball), vb = bits4(vball), happb = bits4(happball);
            const uint32_t gap = ~okb & 0xfu;  // the run of committable slots from slot 0 ends at the first zero bit
            const uint32_t r_ok = gap ? (uint32_t)(__ffs((int)gap) - 1) : (uint32_t)E;
            Rc = 0;
            happb_c = 0;
            uint32_t nev_c = 0;
            bool stopped = false;
            // the usual case needs no walk: the slice mode stops on time alone, and the trace has room for every event of the run
            const uint32_t happ_run = happb & ((1u << r_ok) - 1u);
            const bool plainrun = stop_before && !(P.trace_cap > 0 && dnev + (uint32_t)__popc(happ_run) >= trace_room);
            if (plainrun) {
                Rc = r_ok;
                happb_c = happ_run;
            }
            for (uint32_t r = 0; !plainrun && r < r_ok && !stopped; ++r) {
                Rc = r + 1;
                if ((happb >> r) & 1u) {
                    happb_c |= 1u << r;
                    nev_c += 1;
                    if (dnev + nev_c >= trace_room && P.trace_cap > 0) {
                        status = PDMP_CHAIN_TRACE_FULL;
                        stopped = true;
                    }
                    if (!stop_before && !(uniform_f64(SLT[r]) < T)) {
                        running = false;
                        stopped = true;
                    }
                }
            }
            if (!stopped && r_ok < (uint32_t)E && ((vb >> r_ok) & 1u)) {
                status = PDMP_CHAIN_BOUND_VIOLATED;
                vsel = (int)r_ok;
                vprop = ((bits4(__ballot(okg && violated && !adapt && !xerrg && gl == 0)) >> r_ok) & 1u) != 0;
            }
        }

        // ---------------- commit the valid prefix
        const bool commit = gvalid && (uint32_t)g < Rc;
        const uint64_t happball2 = __ballot(commit && happens && gl == 0);
        const uint64_t nkball = __ballot(commit && newkey);
        if (commit) {
            if (member && (happens || gl < k)) {  // what was (possibly) moved; frozen members write back their own values
                rs->x = x;
                rs->th = th;
                rs->t = t;
                rs->I = I;
            }
            if (active) {
                rs->t_old = t;
                rs->a = a;
                rs->b = b;
                rs->acc = fzflag;
                keys[s] = key;
                if (violated && gl == self) cmut[s] = cj;
            }
            if (gl == self) {
                if (is_freeze) {
                    rs->t_old = tp;  // :94
                    rs->acc = 0;     // f[i] = false, :95
                    keys[s] = key;
                    thf[s] = thf_new;
                } else if (is_thaw) {
                    thf[s] = 0.0;
                    if (!active) rs->t_old = tp;  // :114 (i is not re-bounded when its saved speed was 0)
                }
            }
            if (gl == wl2) {
                bk[blk] = rowmin;
                bi[blk] = cand;
            }
            if (happens && gl == self && ev) {
                const uint32_t rank = (uint32_t)__popcll(happball2 & ((1ull << (16 * g)) - 1ull));
                pdmp_event e;
                e.t = t;  // event(i, t, x, θ, F) = (t[i], i, x[i], θ[i]), :154
                e.i = (int64_t)i;
                e.x = x;
                e.theta = th;
                ev[ntrace0 + dnev + rank] = e;
            }
        }
        LDS_ORDER();
        // ---------------- level-1 updates for new keys living in other blocks.  A block's final entry is the smallest
        // (key, coordinate) among its old entry and the new keys, whatever the order: when no two of these lanes aim at one block
        // (claims through the now idle zone-id array) and none has to rescan, every lane updates its block by itself in one LDS
        // round trip; otherwise one by one in event order.
        const bool upd = commit && newkey && (s >> 6) != blk;
        if (__ballot(upd) != 0) {
            LDS_ORDER();
            const uint32_t bjv = upd ? (s >> 6) : 0u;
            const double curv = bk[bjv];
            const uint32_t civ = bi[bjv];
            const bool lower = upd && (key < curv || (key == curv && s < civ));
            const bool resc = upd && !lower && civ == s;
            if (lower) Z[bjv & 63u] = (uint32_t)lane;
            LDS_ORDER();
            const bool lost = lower && Z[bjv & 63u] != (uint32_t)lane;
            if (__ballot(lost || resc) == 0) {
                if (lower) {
                    bk[bjv] = key;
                    bi[bjv] = s;
                }
            } else {
            for (uint32_t r = 0; r < Rc; ++r) {
                uint32_t lanes = (uint32_t)((nkball >> (16 * r)) & 0xffffull);
                const uint32_t own = uniform_u32(SLB[r]);
                while (lanes) {
                    const int jj = __ffs((int)lanes) - 1;
                    lanes &= lanes - 1u;
                    const uint32_t j = readlane_u32(s, 16 * (int)r + jj);
                    if ((j >> 6) == own) continue;
                    level1_update(bk, bi, keys, lane, j, readlane_f64(key, 16 * (int)r + jj));
                }
            }
            }
        }
        // ---------------- counters (scalar acc, num with the reset of an adapted violation, :131-136)
        if (Rc > 0) {
            const uint64_t violball = __ballot(commit && violated && gl == 0);
            for (uint32_t r = 0; r < Rc; ++r) {
                if (uniform_u32(TY[r]) != 0u) continue;
                num += 1;
                if ((happb_c >> r) & 1u) nacc += 1;
                if ((violball >> (16 * r)) & 1ull) {
                    num = 0;
                    nacc = 0;
                }
            }
            dnev += (uint32_t)__popc(happb_c);
            dnm += uniform_u32(OFR[Rc]);
            t_last = uniform_f64(SLT[Rc - 1]);
            if (happb_c) t_event = uniform_f64(SLT[31 - __builtin_clz(happb_c)]);
        }
        // the event the chain stops on (reference: error(...), :90 / :133): the proposal was counted, its coin drawn, acc bumped and
        // G[i] moved before the check -- what zz_sticky_run_kernel and the oracle leave behind
        if (vsel >= 0) {
            if (vprop) {
                if (g == vsel && gl < k) {
                    rs->x = x;
                    rs->t = t;
                    rs->I = I;
                }
                num += 1;
                nacc += 1;
                dnm += uniform_u32(OFR[vsel]) + 1u - (Rc > 0 ? uniform_u32(OFR[Rc]) : 0u);
            }
            t_last = uniform_f64(SLT[vsel]);
        }
        if (status != PDMP_CHAIN_OK) break;
        LDS_ORDER();
    }

    if (lane == 0) {
        hdr->c.t_last = t_last;
        hdr->t_event = t_event;
        hdr->c.num = num;
        hdr->c.nacc = nacc;
        hdr->c.ntrace = ntrace0 + dnev;
        hdr->c.nevents += dnev;
        hdr->c.ndraw_main = nm0 + dnm;
        hdr->c.status = status;
    }
}

// ------------------------------------------------------------------------------------------ unpack / moments

// final state (t, x, θ), acc, c of chains [chain_first, chain_first + n): src/sfact.jl:211
__global__ __launch_bounds__(256) void zz_unpack_kernel(const ZzRec* rec, const double* c_src, int64_t c_stride,
                                                        int64_t d, int64_t chain_first, double* t, double* x,
                                                        double* th, int64_t* acc, double* c) {
    const int64_t n = blockIdx.x;
    const int64_t i = (int64_t)blockIdx.y * 256 + threadIdx.x;
    if (i >= d) return;
    const ZzRec r = rec[(chain_first + n) * d + i];
    const int64_t o = n * d + i;
    if (t) t[o] = r.t;
    if (x) x[o] = r.x;
    if (th) th[o] = r.th;
    if (acc) acc[o] = (int64_t)r.acc;
    if (c) c[o] = c_src[(chain_first + n) * c_stride + i];
}

// Final state of tracked-gradient chains in the reference's terms (src/sfact.jl:211: per-coordinate lazy clocks t, positions AT those
// clocks, velocities): the reference's t[j] is the time of the last proposal inside G[j] or of the last accepted event inside S[j] (an
// accept moves G2 as well, :129), whichever is later -- both are kept per coordinate (tprop, tacc); x[j] is the tracked position moved
// linearly from its own clock to that time.
__global__ __launch_bounds__(256) void zz_track_unpack_kernel(const TrRec* rec0, ZzTables tb, const double* c_src, int64_t c_stride, int64_t d,
                                                              int64_t chain_first, double t0, double* t, double* x, double* th,
                                                              int64_t* acc, double* c, const double2* kp0, int64_t dk) {
    const int64_t n = blockIdx.x;
    const int64_t i = (int64_t)blockIdx.y * 256 + threadIdx.x;
    if (i >= d) return;
    const TrRec* rec = rec0 + (chain_first + n) * d;
    const TrRec r = rec[i];
    const double2* kp = kp0 ? kp0 + (chain_first + n) * dk : nullptr;
    double tr = t0;
    const uint32_t k = tb.colptr[i + 1] - tb.colptr[i];
    const uint32_t s0 = tb.sptr[i], s1 = tb.sptr[i + 1];
    for (uint32_t p = s0; p < s1; ++p) {  // S[i] = G1[i] followed by G2[i]; the patterns are symmetric: j ∈ S[i] <=> i ∈ S[j]
        const uint32_t j = tb.sidx[p];
        const double tpj = kp ? kp[j].y : rec[j].tprop, taj = kp ? rec[j].tx : rec[j].tacc;  // (pair layout: the last accept's time is the position's clock)
        if (p - s0 < k && tpj > tr) tr = tpj;
        if (taj > tr) tr = taj;
    }
    const int64_t o = n * d + i;
    if (t) t[o] = tr;
    if (x) x[o] = r.x + r.th * (tr - r.tx);
    if (th) th[o] = r.th;
    if (acc) acc[o] = (int64_t)r.acc;
    if (c) c[o] = c_src[(chain_first + n) * c_stride + i];
}

// Batch means of the exact path integral: J = I + ∫_t^T (x + θ(s-t)) ds, Y = (J - Jprev)/ΔT per chain,
// ΣY and ΣY² over chains.  Threads own a coordinate and walk a group of chains (records are 64 B, so a
// warp of consecutive coordinates reads consecutive sectors).
__global__ __launch_bounds__(256) void zz_batch_means_kernel(const ZzRec* rec0, int64_t rec_stride, double* jprev, int64_t d,
                                                             int64_t nchains, int64_t chains_per_group,
                                                             double T_prev, double T, double* sum_y, double* sum_y2) {
    // (rec_stride: 64 for ZzRec, 128 for TrRec, whose first sector has the same fields)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= d) return;
    const int64_t c0 = (int64_t)blockIdx.y * chains_per_group;
    const int64_t c1 = (c0 + chains_per_group < nchains) ? (c0 + chains_per_group) : nchains;
    const double inv = 1.0 / (T - T_prev);
    double s1 = 0.0, s2 = 0.0;
    for (int64_t ch = c0; ch < c1; ++ch) {
        const ZzRec* r = reinterpret_cast<const ZzRec*>(reinterpret_cast<const char*>(rec0) + (ch * d + i) * rec_stride);
        const double dt = T - r->t;
        const double J = r->I + dt * (r->x + r->th * (dt * 0.5));
        const double y = (J - jprev[ch * d + i]) * inv;
        jprev[ch * d + i] = J;
        s1 += y;
        s2 += y * y;
    }
    atomicAdd(sum_y + i, s1);
    atomicAdd(sum_y2 + i, s2);
}

// ESS accumulators (pdmp_ensemble_ess_*): mode 0 snapshots J(T) of every (chain, coordinate) into jprev AND jstart; mode 1 is a
// batch -- Y = (J − jprev)/ΔT, jprev = J, acc[0] += Y, acc[1] += Y² --; mode 2 closes the run -- the chain's own mean over the
// whole run M = (J − jstart)/(T − T0), acc[2] += M, acc[3] += M² (jprev / jstart untouched).  acc is [4 x d].
__global__ __launch_bounds__(256) void zz_ess_kernel(const ZzRec* rec0, int64_t rec_stride, double* jprev, double* jstart, int64_t d,
                                                     int64_t nchains, int64_t chains_per_group, int mode, double T_prev, double T,
                                                     double* acc) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= d) return;
    const int64_t c0 = (int64_t)blockIdx.y * chains_per_group;
    const int64_t c1 = (c0 + chains_per_group < nchains) ? (c0 + chains_per_group) : nchains;
    const double inv = (mode == 0) ? 0.0 : 1.0 / (T - T_prev);
    double s1 = 0.0, s2 = 0.0;
    for (int64_t ch = c0; ch < c1; ++ch) {
        const ZzRec* r = reinterpret_cast<const ZzRec*>(reinterpret_cast<const char*>(rec0) + (ch * d + i) * rec_stride);
        const double dt = T - r->t;
        const double J = r->I + dt * (r->x + r->th * (dt * 0.5));
        if (mode == 0) {
            jprev[ch * d + i] = J;
            jstart[ch * d + i] = J;
        } else {
            const double y = (J - ((mode == 1) ? jprev : jstart)[ch * d + i]) * inv;
            if (mode == 1) jprev[ch * d + i] = J;
            s1 += y;
            s2 += y * y;
        }
    }
    if (mode == 1) {
        atomicAdd(acc + i, s1);
        atomicAdd(acc + d + i, s2);
    } else if (mode == 2) {
        atomicAdd(acc + 2 * d + i, s1);
        atomicAdd(acc + 3 * d + i, s2);
    }
}

// ------------------------------------------------------------------------------------------ math probe
//
// Evaluates the shared numerical contract on the device so that a test can compare it bit-for-bit with
// the host: row 0 u01, 1 pdmp_log(u), 2 a/b, 3 sqrt, 4 poisson_time(a,b,w), 5 pdmp_randn.
__global__ __launch_bounds__(256) void math_probe_kernel(uint64_t seed, int64_t n, double* out) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const double u = pdmp_u01(seed, 0u, (uint64_t)k);
    const double v = pdmp_u01(seed, 1u, (uint64_t)k);
    const double w = pdmp_u01(seed, 2u, (uint64_t)k);
    const double a = (u - 0.5) * 8.0;
    const double b = ((k % 7) == 0) ? 0.0 : (v - 0.5) * 4.0;
    out[0 * n + k] = u;
    out[1 * n + k] = pdmp_log(u);
    out[2 * n + k] = a / ((v - 0.5) * 4.0);
    out[3 * n + k] = sqrt(u * 1000.0 + v);
    out[4 * n + k] = dev_poisson_time(a, b, w);
    out[5 * n + k] = pdmp_randn(seed, 3u, (uint64_t)k);
    out[6 * n + k] = pdmp_exp((u - 0.5) * 60.0 + v);
    {
        double sn_, cs_;
        pdmp_sincos((w - 0.5) * 400.0, &sn_, &cs_);
        out[7 * n + k] = sn_ + 2.0 * cs_;
    }
}

int launch_math_probe(uint64_t seed, int64_t n, double* out, void* stream) {
    hipLaunchKernelGGL(math_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seed,
                       n, out);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------ launchers

int launch_zz_init(const ZzInitParams& p, void* stream) {
    dim3 grid((unsigned)p.nchains, (unsigned)((p.dk + 255) / 256));
    hipLaunchKernelGGL(zz_init_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

bool zz_spec_supported(uint32_t nblk, uint32_t mmax, uint32_t kmax) {
    // (17 <= mmax <= 32: the WIDE instantiation, two zone members per lane)
    return mmax <= 32 && kmax <= 15 && nblk <= 64 * 8;
}

int launch_zz_local_spec(const ZzRunParams& p, int64_t nchains, void* stream, const char** kname) {
    const size_t lds = zz_spec_lds_bytes(p.nblk_pad, p.blob_w_pad);
    const int ne = (int)((p.nblk + 63) / 64);
    dim3 grid((unsigned)nchains), block(64);
    const bool plain_cfg = !p.adapt && p.c_chain == nullptr && p.tb.gmu_t == nullptr && (p.flags & 0x100);
    const bool plain = plain_cfg && p.blob_sw == 7 && p.blob_pw == 1 && p.blob_kmax == 5 && p.blob_w_pad == 58;
    // without a refresh clock the last key block holds only the (infinite) refresh slot: when d fills 256 blocks exactly the
    // queue's first level is scanned as 4 entries per lane instead of 5
    const bool plain4 = plain && !p.has_refresh && p.d == 256 * 64 && p.nblk == 257;
    // the 8-event kernel: the lattice blob geometry, no refresh clock, 2048 <= d <= 16384 (its first level has 512 entries over
    // 32-key blocks; below 64 blocks there are too few candidates for eight slots)
    const bool geom = (p.flags & 0x100) && p.blob_sw == 7 && p.blob_pw == 1 && p.blob_kmax == 5 && p.blob_w_pad == 58;
    const bool spec8 = geom && p.d >= 2048 && p.d <= (int64_t)S8_NBLK * 32 &&
                       !p.force_spec4;  // (pdmp_debug_set_kernel: A/B runs and parity tests of the 4-event kernel)
    const bool wide = p.blob_sw > 8;  // |S[i]| up to 32: two zone members per lane
    // eight events per iteration on any graph with |G1| <= 8, |S| <= 32 (tables built by the host when the geometry fits), plain configuration
    const bool spec8g = !spec8 && p.g8_line != nullptr && (p.flags & 0x100) && p.d >= 2048 && p.d <= (int64_t)S8_NBLK * 32 && !p.force_spec4;
    if (kname) *kname = spec8 ? "zz_local_spec8_kernel" : spec8g ? (p.g8_gw == 16 ? "zz_local_spec8g_kernel<GW=16>" : "zz_local_spec8g_kernel") : wide ? "zz_local_spec_kernel<WIDE>" : "zz_local_spec_kernel";
    if (spec8g) {
        ZzRunParams q = p;
        q.nblk = (uint32_t)((p.d + 31) / 32);
        const bool same = p.g8_gamt == nullptr;  // the target's Γ values are the bounding ones: the gradient's coefficients come from the member lines
        const size_t l8 = zz_spec8g_lds_bytes();
        const hipStream_t st_ = (hipStream_t)stream;
        if (p.g8_gw == 16) {  // four events per iteration, |S| up to 64
            if (!plain_cfg && same) hipLaunchKernelGGL((zz_local_spec8g_kernel<false, true, true, 16>), grid, block, l8, st_, q);
            else if (!plain_cfg) hipLaunchKernelGGL((zz_local_spec8g_kernel<false, false, true, 16>), grid, block, l8, st_, q);
            else if (same) hipLaunchKernelGGL((zz_local_spec8g_kernel<false, true, false, 16>), grid, block, l8, st_, q);
            else hipLaunchKernelGGL((zz_local_spec8g_kernel<false, false, false, 16>), grid, block, l8, st_, q);
        } else if (!plain_cfg) {  // adaptation and / or a target mean
            if (same) hipLaunchKernelGGL((zz_local_spec8g_kernel<false, true, true>), grid, block, l8, st_, q);
            else hipLaunchKernelGGL((zz_local_spec8g_kernel<false, false, true>), grid, block, l8, st_, q);
        } else if (p.dbg && same) hipLaunchKernelGGL((zz_local_spec8g_kernel<true, true>), grid, block, l8, st_, q);
        else if (p.dbg) hipLaunchKernelGGL((zz_local_spec8g_kernel<true, false>), grid, block, l8, st_, q);
        else if (same) hipLaunchKernelGGL((zz_local_spec8g_kernel<false, true>), grid, block, l8, st_, q);
        else hipLaunchKernelGGL((zz_local_spec8g_kernel<false, false>), grid, block, l8, st_, q);
        return (int)hipGetLastError();
    }
    if (wide) {
        const size_t ldsw = zz_spec_wide_lds_bytes(p.nblk_pad, p.blob_w_pad);
        if (p.dbg) hipLaunchKernelGGL((zz_local_spec_wide_kernel<8, true>), grid, block, ldsw, (hipStream_t)stream, p);
        else if (ne <= 1) hipLaunchKernelGGL((zz_local_spec_wide_kernel<1, false>), grid, block, ldsw, (hipStream_t)stream, p);
        else if (ne <= 2) hipLaunchKernelGGL((zz_local_spec_wide_kernel<2, false>), grid, block, ldsw, (hipStream_t)stream, p);
        else if (ne <= 5) hipLaunchKernelGGL((zz_local_spec_wide_kernel<5, false>), grid, block, ldsw, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((zz_local_spec_wide_kernel<8, false>), grid, block, ldsw, (hipStream_t)stream, p);
        return (int)hipGetLastError();
    }
    if (spec8) {
        ZzRunParams q = p;
        q.nblk = (uint32_t)((p.d + 31) / 32);
        if (p.dbg && plain) hipLaunchKernelGGL((zz_local_spec8_kernel<true>), grid, block, zz_spec8_lds_bytes(), (hipStream_t)stream, q);
        else if (p.dbg) hipLaunchKernelGGL((zz_local_spec8_kernel<true, true>), grid, block, zz_spec8_lds_bytes(), (hipStream_t)stream, q);
        else if (plain) hipLaunchKernelGGL((zz_local_spec8_kernel<false>), grid, block, zz_spec8_lds_bytes(), (hipStream_t)stream, q);
        else hipLaunchKernelGGL((zz_local_spec8_kernel<false, true>), grid, block, zz_spec8_lds_bytes(), (hipStream_t)stream, q);
    } else if (p.dbg) {  // per-phase cycle profile (pdmp_debug_set_phase_profile)
        if (plain4) {
            ZzRunParams q = p;
            q.nblk = 256;
            hipLaunchKernelGGL((zz_local_spec_kernel<4, true, true>), grid, block, lds, (hipStream_t)stream, q);
        } else {
            hipLaunchKernelGGL((zz_local_spec_kernel<8, true>), grid, block, lds, (hipStream_t)stream, p);
        }
    } else if (ne <= 1) {
        hipLaunchKernelGGL((zz_local_spec_kernel<1, false>), grid, block, lds, (hipStream_t)stream, p);
    } else if (ne <= 2) {
        hipLaunchKernelGGL((zz_local_spec_kernel<2, false>), grid, block, lds, (hipStream_t)stream, p);
    } else if (ne <= 5) {
        if (plain4) {
            ZzRunParams q = p;
            q.nblk = 256;
            hipLaunchKernelGGL((zz_local_spec_kernel<4, false, true>), grid, block, lds, (hipStream_t)stream, q);
        } else if (plain) {
            hipLaunchKernelGGL((zz_local_spec_kernel<5, false, true>), grid, block, lds, (hipStream_t)stream, p);
        } else {
            hipLaunchKernelGGL((zz_local_spec_kernel<5, false>), grid, block, lds, (hipStream_t)stream, p);
        }
    } else {
        hipLaunchKernelGGL((zz_local_spec_kernel<8, false>), grid, block, lds, (hipStream_t)stream, p);
    }
    return (int)hipGetLastError();
}

int launch_zz_sticky_spec(const ZzRunParams& p, int64_t nchains, void* stream) {
    const size_t lds = zz_sticky_spec_lds_bytes(p.nblk_pad, p.blob_w_pad);
    const int ne = (int)((p.nblk + 63) / 64);
    dim3 grid((unsigned)nchains), block(64);
    if (ne <= 1) {
        hipLaunchKernelGGL((zz_sticky_spec_kernel<1>), grid, block, lds, (hipStream_t)stream, p);
    } else if (ne <= 2) {
        hipLaunchKernelGGL((zz_sticky_spec_kernel<2>), grid, block, lds, (hipStream_t)stream, p);
    } else if (ne <= 5) {
        const bool plain = !p.adapt && p.c_chain == nullptr && p.tb.gmu_t == nullptr && !p.reversible && !p.strong_upperbounds &&
                           p.blob_sw == 7 && p.blob_pw == 1 && p.blob_kmax == 5 && p.blob_w_pad == 58;
        if (plain) hipLaunchKernelGGL((zz_sticky_spec_kernel<5, true>), grid, block, lds, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((zz_sticky_spec_kernel<5>), grid, block, lds, (hipStream_t)stream, p);
    } else {
        hipLaunchKernelGGL((zz_sticky_spec_kernel<8>), grid, block, lds, (hipStream_t)stream, p);
    }
    return (int)hipGetLastError();
}

int launch_zz_sticky_run(const ZzRunParams& p, int64_t nchains, void* stream) {
    const size_t lds = zz_sticky_lds_bytes(p.nblk_pad, p.blob_w_pad);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(zz_sticky_run_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(zz_sticky_run_kernel, dim3((unsigned)nchains), dim3(64), lds, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

int launch_zz_local_run(const ZzRunParams& p, int64_t nchains, void* stream) {
    const size_t lds = zz_local_lds_bytes(p.nblk_pad, p.blob_w_pad);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(zz_local_run_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(zz_local_run_kernel, dim3((unsigned)nchains), dim3(64), lds, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

int launch_zz_unpack(const ZzRec* rec, const double* c_src, int64_t c_stride, int64_t d, int64_t chain_first,
                     int64_t n, double* t, double* x, double* th, int64_t* acc, double* c, void* stream) {
    dim3 grid((unsigned)n, (unsigned)((d + 255) / 256));
    hipLaunchKernelGGL(zz_unpack_kernel, grid, dim3(256), 0, (hipStream_t)stream, rec, c_src, c_stride, d,
                       chain_first, t, x, th, acc, c);
    return (int)hipGetLastError();
}

int launch_zz_track_unpack(const TrRec* rec, const ZzTables& tb, const double* c_src, int64_t c_stride, int64_t d, int64_t chain_first,
                           int64_t n, double t0, double* t, double* x, double* th, int64_t* acc, double* c, const double* kp, int64_t dk,
                           void* stream) {
    dim3 grid((unsigned)n, (unsigned)((d + 255) / 256));
    hipLaunchKernelGGL(zz_track_unpack_kernel, grid, dim3(256), 0, (hipStream_t)stream, rec, tb, c_src, c_stride, d, chain_first, t0, t, x,
                       th, acc, c, reinterpret_cast<const double2*>(kp), dk);
    return (int)hipGetLastError();
}

bool zz_spec8_geometry(const ZzRunParams& p) {
    return p.blob_sw == 7 && p.blob_pw == 1 && p.blob_kmax == 5 && p.blob_w_pad == 58 && !p.has_refresh && p.d >= 2048 &&
           p.d <= (int64_t)S8_NBLK * 32;
}

int launch_zz_local_track(const ZzRunParams& p, int64_t nchains, void* stream) {
    if (!zz_spec8_geometry(p)) return -1;
    dim3 grid((unsigned)nchains), block(64);
    ZzRunParams q = p;
    q.nblk = (uint32_t)((p.d + 31) / 32);
    const bool plain = !p.adapt && p.c_chain == nullptr && p.tb.gmu_t == nullptr && !p.track_two_sums;
    if (p.dbg && plain) hipLaunchKernelGGL((zz_local_track_kernel<true>), grid, block, zz_spec8_lds_bytes(), (hipStream_t)stream, q);
    else if (p.dbg) hipLaunchKernelGGL((zz_local_track_kernel<true, true>), grid, block, zz_spec8_lds_bytes(), (hipStream_t)stream, q);
    else if (plain) hipLaunchKernelGGL((zz_local_track_kernel<false>), grid, block, zz_spec8_lds_bytes(), (hipStream_t)stream, q);
    else hipLaunchKernelGGL((zz_local_track_kernel<false, true>), grid, block, zz_spec8_lds_bytes(), (hipStream_t)stream, q);
    return (int)hipGetLastError();
}

int launch_zz_batch_means(const ZzRec* rec, int64_t rec_stride, double* jprev, int64_t d, int64_t nchains, double T_prev, double T,
                          double* sum_y, double* sum_y2, void* stream) {
    const int64_t groups = (nchains < 64) ? 1 : 64;
    const int64_t per = (nchains + groups - 1) / groups;
    dim3 grid((unsigned)((d + 255) / 256), (unsigned)groups);
    hipLaunchKernelGGL(zz_batch_means_kernel, grid, dim3(256), 0, (hipStream_t)stream, rec, rec_stride, jprev, d, nchains, per,
                       T_prev, T, sum_y, sum_y2);
    return (int)hipGetLastError();
}

int launch_zz_ess(const ZzRec* rec, int64_t rec_stride, double* jprev, double* jstart, int64_t d, int64_t nchains, int mode, double T_prev,
                  double T, double* acc, void* stream) {
    const int64_t groups = (nchains < 64) ? 1 : 64;
    const int64_t per = (nchains + groups - 1) / groups;
    dim3 grid((unsigned)((d + 255) / 256), (unsigned)groups);
    hipLaunchKernelGGL(zz_ess_kernel, grid, dim3(256), 0, (hipStream_t)stream, rec, rec_stride, jprev, jstart, d, nchains, per, mode,
                       T_prev, T, acc);
    return (int)hipGetLastError();
}

}  // namespace pdmp
