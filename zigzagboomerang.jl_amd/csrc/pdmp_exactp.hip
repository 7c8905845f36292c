// pdmp_exactp.hip -- zz_local_exactp_kernel: the BIT-IDENTICAL (moving) evaluation of the local ZigZag on the plain lattice, one proposal per
// LANE -- the scheme of pdmp_trackp.hip (float lower bounds as the queue's first level, threshold selection, accept chain as a fix-point,
// exposure prefix, commit of the valid prefix) carried over to the reference's own arithmetic: every proposal moves G1[i] with the lazy
// clocks (src/sfact.jl:6-12,82), gathers Γ[:,i]·x in idot's order (src/common.jl:16-24), thins (:119-121) and re-bounds (:131-140) exactly
// as spdmp_inner! does, so indices, outcomes, times, positions and the final state equal the oracle's bit for bit.
//
// zz_local_spec8_kernel gives an event an 8-lane group: 7.3 committed proposals per iteration, 7.1 lines read per proposal (key blocks of 32
// = two lines, records of the neighbours one line each).  Here
//   * a candidate's lane holds its event's data: its own record and its four neighbours' (i ± 1, i ± n on the lattice: no tables) are requested
//     together with the key line, using the position bits of the first level; the move, the gradient, both rates and the rejected
//     proposal's new bound are one lane's straight-line arithmetic.  What depends on the ORDER of the events (accept chain, zones,
//     exposure prefix) runs in rank space on a handful of scalars; the moved records are formed again at commit from a second read;
//   * key blocks of 16 (one line): 1024 first-level entries as 4-byte lower bounds (4 KB of LDS);
//   * zones: a rejected proposal WRITES G1[i] (radius 1 on the lattice), an accepted one S[i] (radius 2): events r > m conflict iff their
//     lattice distance is <= radius(m) + radius(r); the list ends at the first conflict (everything before it is untouched);
//   * the <= 8 accepted events of an iteration are finished by 8-lane groups (the last groups of the wave): the 13 members of S[i] moved, θ_i
//     flipped, the five members of G1[i] re-bounded from a 5 x 5 window of (x, θ) staged in LDS.
// Records (ZzRec) and keys keep the layout every other entry point reads: nothing is converted, slices and trace refills work as before.
// MEASURED (MI355X, C3, 4096 chains, ΔT = 1): 19.9 committed proposals per iteration (8-event kernel: 7.3), 3707 iterations per chain -- and
// 136 ms per step against the 8-event kernel's 98 ms: an iteration is ~3500 static instructions long and costs 85k cycles at 4 waves/SIMD
// (commit + store drain 45 %, accepted groups 17 %, loads + key lines 15 %, rank space 10 %); without ANY neighbour store it would still
// take 108 ms.  So this kernel is OPT-IN (PDMP_DEBUG_KERNEL_EXACTP, PDMP_KERNEL=exactp in the Python host) and a fourth independent
// implementation in the parity suite; the 8-event kernel stays the default of the moving evaluation.  DESIGN.md has the account.
// Requirements: the n x n lattice in column-major numbering with the bounding Γ equal to the target's, no adaptation, no target mean, no
// refresh clock, 2048 <= d <= 16384 (elsewhere: zz_local_spec8_kernel and its relatives).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/pdmp_detmath.h"
#include "pdmp_engine.hpp"

namespace pdmp {

#define X_INF __builtin_inf()
#define X_ORDER()                        \
    do {                                 \
        __builtin_amdgcn_wave_barrier(); \
        asm volatile("" ::: "memory");   \
    } while (0)

namespace {

__device__ __forceinline__ double x_readlane(double v, int srclane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double x_uniform(double v) {
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double x_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double x_min(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double x_wave_min(double v) {
    v = x_min(v, x_dpp<0xB1>(v));
    v = x_min(v, x_dpp<0x4E>(v));
    v = x_min(v, x_dpp<0x141>(v));
    v = x_min(v, x_dpp<0x140>(v));
    v = x_min(v, x_dpp<0x142>(v));
    v = x_min(v, x_dpp<0x143>(v));
    return x_readlane(v, 63);
}
__device__ __forceinline__ double x_grp8_min(double v) {  // minimum over the 8 lanes of a group, in every lane of the group
    v = x_min(v, x_dpp<0xB1>(v));
    v = x_min(v, x_dpp<0x4E>(v));
    v = x_min(v, x_dpp<0x141>(v));
    return v;
}
__device__ __forceinline__ double x_pos(double x) {
    return (x > 0.0) ? x : ((x != x) ? x : 0.0);
}
__device__ __forceinline__ double x_poisson_time_L(double a, double b, double L) {  // src/poissontime.jl:8-30 with L = log(u)
    if (b == 0) return (a > 0) ? -L / a : X_INF;
    const double r = a / b;
    const double q = L * 2.0 / b;
    const double sq = sqrt((b > 0 && a < 0) ? -q : r * r - q);
    if (b > 0) return sq - r;
    if (a <= 0) return X_INF;
    if (-L <= -(a * a) / b + (a * a) / (2 * b)) return -sq - r;
    return X_INF;
}
__device__ __forceinline__ double x_below(double x) {  // the largest double below a finite x
    long long b = __double_as_longlong(x);
    if (x > 0) b -= 1;
    else if (x < 0) b += 1;
    else b = (long long)0x8000000000000001ull;
    return __longlong_as_double(b);
}
template <int CTRL, int ROWM, int BANKM>
__device__ __forceinline__ uint32_t x_dpp_id_u32(uint32_t identity, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)src, CTRL, ROWM, BANKM, false);
}
__device__ __forceinline__ uint32_t x_scan_add_u32(uint32_t v) {  // inclusive
    uint32_t x = v;
    x += x_dpp_id_u32<0x111, 0xf, 0xf>(0u, v);
    x += x_dpp_id_u32<0x112, 0xf, 0xf>(0u, v);
    x += x_dpp_id_u32<0x113, 0xf, 0xf>(0u, v);
    x += x_dpp_id_u32<0x114, 0xf, 0xe>(0u, x);
    x += x_dpp_id_u32<0x118, 0xf, 0xc>(0u, x);
    x += x_dpp_id_u32<0x142, 0xa, 0xf>(0u, x);
    x += x_dpp_id_u32<0x143, 0xc, 0xf>(0u, x);
    return x;
}
template <int CTRL, int ROWM, int BANKM>
__device__ __forceinline__ double x_dpp_inf(double src) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(src), CTRL, ROWM, BANKM, false);
    const int hi = __builtin_amdgcn_update_dpp(0x7FF00000, __double2hiint(src), CTRL, ROWM, BANKM, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double x_scan_min_f64(double v) {  // inclusive
    double x = v;
    x = x_min(x, x_dpp_inf<0x111, 0xf, 0xf>(v));
    x = x_min(x, x_dpp_inf<0x112, 0xf, 0xf>(v));
    x = x_min(x, x_dpp_inf<0x113, 0xf, 0xf>(v));
    x = x_min(x, x_dpp_inf<0x114, 0xf, 0xe>(x));
    x = x_min(x, x_dpp_inf<0x118, 0xf, 0xc>(x));
    x = x_min(x, x_dpp_inf<0x142, 0xa, 0xf>(x));
    x = x_min(x, x_dpp_inf<0x143, 0xc, 0xf>(x));
    return x;
}
__device__ __forceinline__ double x_shfl(double v, uint32_t src) {
    const int lo = __builtin_amdgcn_ds_bpermute((int)(src << 2), __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute((int)(src << 2), __double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ uint32_t x_shfl_u32(uint32_t v, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)v);
}

// first-level entry: a lower bound of (key - tb) as the bit pattern of a non-negative float, 16 ulp below the nearest float of the difference,
// with the argument's position inside its block of 16 in the four low bits (bit patterns of non-negative floats order like the floats)
__device__ __forceinline__ uint32_t q_enc(double key, double tb, uint32_t pos) {
    double dlt = key - tb;
    dlt = (dlt > 0.0) ? dlt : 0.0;
    const uint32_t b = __float_as_uint((float)dlt);
    return (((b >= 16u) ? b - 16u : 0u) & ~15u) | pos;
}
__device__ __forceinline__ uint32_t q_thr(double tau, double tb) {  // the largest pattern a block with exact minimum <= tau can carry
    double dlt = tau - tb;
    dlt = (dlt > 0.0) ? dlt : 0.0;
    return __float_as_uint((float)dlt) + 1u;
}
__device__ __forceinline__ double q_dec(uint32_t bits, double tb) {
    return x_below(tb + (double)__uint_as_float(bits & ~15u));
}
constexpr uint32_t Q_INFBITS = 0x7f7ffff0u;  // patterns from here on: the block is empty (+Inf)

}  // namespace

// LDS layout (bytes)
constexpr uint32_t X_LB = 0;         // [1024] u32 lower bounds of the block minima (+ argument position), key blocks of 16
constexpr uint32_t X_EX = 4096;      // [64] f64 what event e exposes; before that the exact block minima of the candidates
constexpr uint32_t X_RS = 4608;      // [56] f64 candidates: block minimum without the argument
constexpr uint32_t X_PB = 5056;      // [64] u8 candidates: position of the argument | position of the runner-up << 4
constexpr uint32_t X_TB = 5248;      // [64] u16 candidate blocks, compaction order
constexpr uint32_t X_ACL = 5376;     // [8] u16 the accepted events
constexpr uint32_t X_RO = 5392;      // [64] u8 candidate of each rank
constexpr uint32_t X_SELDT = 5456;   // f64 selection threshold above the minimum
constexpr uint32_t X_ZS = 5472;      // [8 groups][25] (x, θ) of the 5 x 5 window around an accepted event, at t′
constexpr uint32_t X_BYTES = X_ZS + 8 * 25 * 16;
constexpr uint32_t X_NBLK = 1024;
constexpr uint32_t X_WIN = 128;  // draws held in registers (two per lane)
constexpr int X_CMAX = 56;       // candidates per iteration (7 block-scan passes of 8)
constexpr int X_AMAX = 8;        // accepted events per iteration (one group each)
#ifndef X_GROW
#define X_GROW 1.02
#define X_SHRINK 0.98
#define X_SLACK 5u
#endif
static_assert(X_BYTES <= 10240, "16 chains per CU: 160 KB / 16");

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void zz_local_exactp_kernel(ZzRunParams P) {
    const int lane = threadIdx.x;
    const int g = lane >> 3;
    const int64_t chain = blockIdx.x;
    const int64_t d = P.d;
    const uint32_t nblk = P.nblk;
    const uint32_t nlat = (uint32_t)P.lattice_n, nmagic = P.lattice_magic;

    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t* const lbf = reinterpret_cast<uint32_t*>(smem + X_LB);
    double* const EX = reinterpret_cast<double*>(smem + X_EX);
    double* const KM = EX;
    double* const RS = reinterpret_cast<double*>(smem + X_RS);
    uint8_t* const PB = reinterpret_cast<uint8_t*>(smem + X_PB);
    uint16_t* const TB = reinterpret_cast<uint16_t*>(smem + X_TB);
    uint16_t* const ACL = reinterpret_cast<uint16_t*>(smem + X_ACL);
    uint8_t* const RO = reinterpret_cast<uint8_t*>(smem + X_RO);
    double* const SELDT = reinterpret_cast<double*>(smem + X_SELDT);
    double2* const ZS = reinterpret_cast<double2*>(smem + X_ZS) + g * 25;

    ZzRec* const rec = P.rec + chain * d;
    double* const keys = P.keys + chain * P.dk;
    DevChain* const hdr = P.hdr + chain;
    pdmp_event* const evout = P.ev ? P.ev + chain * P.trace_cap : nullptr;
    const CoordConst* const cc = P.tb.cc_shared;

    uint32_t status = hdr->c.status;
    if (status == PDMP_CHAIN_BOUND_VIOLATED || status == PDMP_CHAIN_STALLED) return;
    const uint64_t seed = hdr->seed;
    const uint64_t nm0 = hdr->c.ndraw_main, ntrace0 = hdr->c.ntrace;
    uint32_t dnm = 0, dnum = 0, dnacc = 0, vnacc = 0;
    double ureg[2] = {0.0, 0.0};
    uint32_t uidx[2] = {0xffffffffu, 0xffffffffu};
    double t_last = hdr->c.t_last;
    double t_event = hdr->t_event;
    status = PDMP_CHAIN_OK;
    const double T = P.T;
    const bool stop_before = (P.flags & PDMP_RUN_STOP_BEFORE) != 0;
    const uint32_t trace_room = (P.trace_cap > 0)
                                    ? (uint32_t)(((uint64_t)P.trace_cap > ntrace0) ? ((uint64_t)P.trace_cap - ntrace0) : 0)
                                    : 0xffffffffu;

    if (lane == 0) SELDT[0] = 1e-3;
    double tb;  // base of the first-level bounds: below every key (wave-uniform; moves up with the front)
    {
        double mloc = t_last;
        for (uint32_t b = lane; b < nblk; b += 64) {
            const double* p = keys + (size_t)b * 16;
#pragma unroll
            for (int q = 0; q < 16; ++q) mloc = x_min(mloc, p[q]);
        }
        tb = x_wave_min(mloc);
    }
    for (uint32_t b = lane; b < X_NBLK; b += 64) {
        uint32_t e = Q_INFBITS;
        if (b < nblk) {
            const double* p = keys + (size_t)b * 16;
            double mk = p[0];
            uint32_t mi = 0;
#pragma unroll
            for (int q = 1; q < 16; ++q) {
                const double v = p[q];
                if (v < mk) {
                    mk = v;
                    mi = q;
                }
            }
            e = (mk < X_INF) ? q_enc(mk, tb, mi) : Q_INFBITS;
        }
        lbf[b] = e;
    }
    X_ORDER();

#ifdef X_PHASES
    uint64_t ph[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t ph_t = __builtin_amdgcn_s_memtime();
#define X_PH(k)                                              \
    do {                                                     \
        __builtin_amdgcn_s_waitcnt(0);                       \
        const uint64_t n_ = __builtin_amdgcn_s_memtime();    \
        ph[k] += n_ - ph_t;                                  \
        ph_t = n_;                                           \
    } while (0)
#else
#define X_PH(k) do { } while (0)
#endif
    PrioTurn prio;
    uint32_t st_iters = 0, st_raw = 0, st_events = 0, st_zone = 0, st_commit = 0;  // diagnostics (pdmp_debug_set_phase_profile)
    bool need_rebase = false;
    uint32_t idle = 0;
    bool running = stop_before || (t_event < T);
    const int lane_outer = lane;
    while (running) {
        // (the lane index is made opaque once per iteration: what depends on it alone -- group offsets, window offsets, LDS addresses -- is
        // two or three instructions to form again, and hoisted out of the loop it would sit in registers the loop body needs, or in scratch)
        int lane_it = lane_outer;
        asm volatile("" : "+v"(lane_it));
        const int lane = lane_it;
        const int g = lane >> 3, gl = lane & 7;
        prio.step();
        if (dnacc >= trace_room) {
            status = PDMP_CHAIN_TRACE_FULL;
            break;
        }
        if (dnm >= P.count_limit) {  // (32-bit counters of the launch: pause, the host runs again)
            status = PDMP_CHAIN_PAUSED;
            break;
        }
        X_PH(8);
        // ---------------- ring of uniforms: draws dnm .. dnm + 127, two per lane
        {
            const uint32_t n0 = dnm + (((uint32_t)lane - dnm) & 63u);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t n = n0 + 64u * (uint32_t)h;
                const int q = (int)((n >> 6) & 1u);
                const bool need0 = (q == 0) && uidx[0] != n, need1 = (q == 1) && uidx[1] != n;
                if (__ballot(need0 || need1) != 0) {
                    const double u = pdmp_u01(seed, PDMP_STREAM_MAIN, nm0 + (uint64_t)n);
                    if (need0) {
                        ureg[0] = u;
                        uidx[0] = n;
                    }
                    if (need1) {
                        ureg[1] = u;
                        uidx[1] = n;
                    }
                }
            }
        }
        auto draw = [&](uint32_t n) -> double {  // draw nm0 + n for dnm <= n < dnm + 128 (every lane calls it)
            const double v0 = x_shfl(ureg[0], n & 63u), v1 = x_shfl(ureg[1], n & 63u);
            return ((n >> 6) & 1u) ? v1 : v0;
        };
        X_PH(0);
        // ---------------- select: every block whose lower bound is within the threshold, at most X_CMAX of them
        int C = 0;
        bool stalled = false, finished = false;
        double dt_used = 0.0;
        uint32_t Cc = 0;
        double tau = 0.0;
        bool tau_clipped = false;
        {
            uint32_t kk[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // lane's 16 entries: blocks 4 (lane + 64 j) + 0..3
                const uint4 v = reinterpret_cast<const uint4*>(lbf)[lane + 64 * j];
                kk[4 * j + 0] = v.x;
                kk[4 * j + 1] = v.y;
                kk[4 * j + 2] = v.z;
                kk[4 * j + 3] = v.w;
            }
            uint32_t mloc = kk[0];
#pragma unroll
            for (int j = 1; j < 16; ++j) mloc = (kk[j] < mloc) ? kk[j] : mloc;
            for (int off = 32; off >= 1; off >>= 1) {
                const uint32_t o = (uint32_t)__shfl_xor((int)mloc, off, 64);
                mloc = (o < mloc) ? o : mloc;
            }
            uint32_t mqb = mloc;
            double mql = q_dec(mqb, tb);
            if (mqb < Q_INFBITS && (need_rebase || mql - tb > 0.25)) {
#pragma unroll
                for (int j = 0; j < 16; ++j) kk[j] = (kk[j] >= Q_INFBITS) ? Q_INFBITS : q_enc(q_dec(kk[j], tb), mql, kk[j] & 15u);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    reinterpret_cast<uint4*>(lbf)[lane + 64 * j] = make_uint4(kk[4 * j + 0], kk[4 * j + 1], kk[4 * j + 2], kk[4 * j + 3]);
                tb = mql;
                need_rebase = false;
                mloc = kk[0];
#pragma unroll
                for (int j = 1; j < 16; ++j) mloc = (kk[j] < mloc) ? kk[j] : mloc;
                for (int off = 32; off >= 1; off >>= 1) {
                    const uint32_t o = (uint32_t)__shfl_xor((int)mloc, off, 64);
                    mloc = (o < mloc) ? o : mloc;
                }
                mqb = mloc;
                mql = q_dec(mqb, tb);
                X_ORDER();
            }
            if (mqb >= Q_INFBITS) {
                stalled = true;
            } else if (stop_before && !(mql < T)) {
                finished = true;
            } else {
                double dt_sel = x_uniform(SELDT[0]);
                uint32_t cm = 0, ncl = 0, incl = 0;
                for (int tries = 0;; ++tries) {
                    tau = mql + dt_sel;
                    tau_clipped = false;
                    if (stop_before && !(tau < T)) {
                        tau = x_below(T);
                        tau_clipped = true;
                    }
                    if (tries >= 64) tau = mql;
                    const uint32_t thr = q_thr(tau, tb);
                    cm = 0;
#pragma unroll
                    for (int j = 15; j >= 0; --j) cm = cm + cm + ((kk[j] <= thr) ? 1u : 0u);
                    ncl = (uint32_t)__builtin_popcount(cm);
                    incl = x_scan_add_u32(ncl);
                    Cc = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    if (Cc <= (uint32_t)X_CMAX || tries >= 64) break;
                    dt_sel *= 0.5;
                }
                {
                    uint32_t ix = incl - ncl, m_ = cm;
                    while (__ballot(m_ != 0u) != 0) {
                        if (m_ != 0u) {
                            const uint32_t j = (uint32_t)(__ffs((int)m_) - 1);
                            if (ix < 64u) TB[ix] = (uint16_t)(4u * ((uint32_t)lane + 64u * (j >> 2)) + (j & 3u));
                            ix += 1;
                            m_ &= m_ - 1u;
                        }
                    }
                }
                dt_used = dt_sel;
            }
        }
        if (stalled) {
            status = PDMP_CHAIN_STALLED;
            break;
        }
        if (finished) break;
        X_ORDER();
        const bool crowded = Cc > (uint32_t)X_CMAX;
        if (crowded) Cc = (uint32_t)X_CMAX;
        X_PH(1);
        // ---------------- candidate lane c: its block, and -- requested now, with the block's line -- the records the position bits point at:
        // the coordinate's own and its lattice neighbours' {i − n, i − 1, i + 1, i + n} (absent ones: the own record again, unused)
        const bool isc = (uint32_t)lane < Cc;
        const uint32_t cblk = isc ? (uint32_t)TB[lane] : 0u;
        const uint32_t cpos = isc ? (lbf[cblk] & 15u) : 0u;
        const uint32_t ci = cblk * 16u + cpos;
        const uint32_t ccol = __umulhi(ci, nmagic), crow = ci - ccol * nlat;
        const ZzRec* const rS = rec + ci;
        const ZzRec* const rL = rec + ((ccol > 0u) ? ci - nlat : ci);
        const ZzRec* const rU = rec + ((crow > 0u) ? ci - 1u : ci);
        const ZzRec* const rD = rec + ((crow + 1u < nlat) ? ci + 1u : ci);
        const ZzRec* const rR = rec + ((ccol + 1u < nlat) ? ci + nlat : ci);
        // (x, θ) and (t, I) of the five; the own record's bound (t_old, a), (b, acc)
        // ... and c_i, c_i / 100, Γ[G1[i], i] in G1's order (the L2-resident table, one half line per coordinate)
        const double2 z2 = make_double2(0.0, 0.0);
        double2 cS0 = z2, cS1 = z2, cS2 = z2, cL0 = z2, cL1 = z2, cU0 = z2, cU1 = z2, cD0 = z2, cD1 = z2, cR0 = z2, cR1 = z2;
        double2 kA = z2, kB = z2, kC = z2, kD = z2;
        if (isc) {  // (the other lanes ask for nothing)
            cS0 = *reinterpret_cast<const double2*>(&rS->x), cS1 = *reinterpret_cast<const double2*>(&rS->t);
            cS2 = *reinterpret_cast<const double2*>(&rS->t_old);
            cL0 = *reinterpret_cast<const double2*>(&rL->x), cL1 = *reinterpret_cast<const double2*>(&rL->t);
            cU0 = *reinterpret_cast<const double2*>(&rU->x), cU1 = *reinterpret_cast<const double2*>(&rU->t);
            cD0 = *reinterpret_cast<const double2*>(&rD->x), cD1 = *reinterpret_cast<const double2*>(&rD->t);
            cR0 = *reinterpret_cast<const double2*>(&rR->x), cR1 = *reinterpret_cast<const double2*>(&rR->t);
            const double2* const kp = reinterpret_cast<const double2*>(cc + ci);
            kA = kp[0];  // (c, c / 100)
            kB = kp[1];  // ((cp, k), Γ[0])
            kC = kp[2];  // (Γ[1], Γ[2])
            kD = kp[3];  // (Γ[3], Γ[4])
        }
        const double b_own = isc ? rS->b : 0.0;
        // ---------------- the candidates' key lines, 8 per pass (one per 8-lane group, two keys per lane): exact minimum, its position, the
        // minimum of the rest and its position -- staged in LDS per candidate
        {
            const int npass = ((int)Cc + 7) >> 3;
            for (int p0 = 0; p0 < npass; p0 += 4) {
                double2 k2[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = 8 * (p0 + q) + g;
                    k2[q] = make_double2(X_INF, X_INF);
                    if (e < (int)Cc) k2[q] = *reinterpret_cast<const double2*>(keys + (size_t)TB[e] * 16 + 2 * gl);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (8 * (p0 + q) >= (int)Cc) continue;  // (uniform)
                    const int e = 8 * (p0 + q) + g;
                    const double ka = k2[q].x, kb = k2[q].y;  // positions 2 gl and 2 gl + 1
                    const bool bfirst = kb < ka;              // (ties: the lower position)
                    const double lm = bfirst ? kb : ka;
                    const double gm = x_grp8_min(lm);
                    const uint64_t winball = __ballot(lm == gm);
                    const int wl = __ffs((unsigned)((winball >> (8 * g)) & 0xffu)) - 1;  // (>= 0)
                    const uint32_t wpos = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((uint32_t)(lane & ~7) + (uint32_t)wl) << 2),
                                                                                 (int)(2u * (uint32_t)gl + (bfirst ? 1u : 0u)));
                    // the rest: the winner's lane offers its other key
                    const double lr = (gl == wl) ? (bfirst ? ka : kb) : lm;
                    const uint32_t lrpos = (gl == wl) ? (2u * (uint32_t)gl + (bfirst ? 0u : 1u)) : (2u * (uint32_t)gl + (bfirst ? 1u : 0u));
                    const double gr = x_grp8_min(lr);
                    const uint64_t rball = __ballot(lr == gr);
                    const int rl = __ffs((unsigned)((rball >> (8 * g)) & 0xffu)) - 1;
                    const uint32_t rpos = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((uint32_t)(lane & ~7) + (uint32_t)(rl < 0 ? 0 : rl)) << 2), (int)lrpos);
                    if (gl == wl && e < (int)Cc) {
                        KM[e] = gm;
                        RS[e] = gr;
                        PB[e] = (uint8_t)(wpos | (rpos << 4));
                    }
                }
            }
        }
        X_ORDER();
        X_PH(2);
        // ---------------- events = candidates whose exact minimum is within the threshold; everybody refreshes its bound
        const double c_km = isc ? KM[lane] : X_INF;
        const uint32_t c_pb = isc ? (uint32_t)PB[lane] : 0u;
        if (isc) lbf[cblk] = (c_km < X_INF) ? q_enc(c_km, tb, c_pb & 15u) : Q_INFBITS;
        bool isev = isc && c_km <= tau;
        if (crowded) {
            isev = false;
            need_rebase = true;
        }
        const double own = isev ? c_km : X_INF;
        uint32_t rank = 0;
        for (uint32_t m0 = 0; m0 < Cc; m0 += 4) {
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) {
                const double km = x_readlane(own, (int)(m0 + q));
                rank += (km < own) ? 1u : 0u;
            }
        }
        const uint64_t evb = __ballot(isev);
        int nev = __popcll(evb);
        if (isev) RO[rank] = (uint8_t)lane;
        X_ORDER();
        const bool dup = isev && RO[rank] != (uint8_t)lane;
        if (__ballot(dup) != 0) {
            // exactly equal keys among the events: one event this iteration, the tied minimum of the lowest block (= lowest coordinate)
            const double mn = x_wave_min(own);
            uint32_t bsel = (isev && own == mn) ? cblk : 0xffffffffu;
            for (int off = 32; off >= 1; off >>= 1) {
                const uint32_t o = (uint32_t)__shfl_xor((int)bsel, off, 64);
                bsel = (o < bsel) ? o : bsel;
            }
            X_ORDER();
            if (isev && cblk == bsel) RO[0] = (uint8_t)lane;
            nev = 1;
            X_ORDER();
        }
        {
            const bool wrongpos = isev && (c_pb & 15u) != cpos;  // the records were requested at another position: the list ends there
            uint32_t wr = wrongpos ? rank : 0xffffffffu;
            for (int off = 32; off >= 1; off >>= 1) {
                const uint32_t o = (uint32_t)__shfl_xor((int)wr, off, 64);
                wr = (o < wr) ? o : wr;
            }
            if (wr < (uint32_t)nev) nev = (int)wr;
        }
        C = nev;
        const int Craw = (int)Cc;
        st_iters += 1;
        st_raw += Cc;
        st_events += (uint32_t)nev;
        if (C == 0) {
            if (tau_clipped && !crowded && __ballot(isc && c_km <= tau) == 0) break;  // stop_before: every key is at or beyond T
            if (++idle > 4096u) {
                status = PDMP_CHAIN_STALLED;
                break;
            }
            if (lane == 0) SELDT[0] = dt_used * 2.0;
            X_ORDER();
            continue;
        }
        X_PH(3);
        // ---------------- Two lane spaces.  The CANDIDATE's lane keeps the heavy data -- the five records it requested -- and does the
        // arithmetic of its event (move, gradient, rates, the rejected proposal's bound, the commit's stores).  Everything that depends on the
        // events' ORDER -- the accept chain, the zone test, the exposure prefix -- runs in RANK space: lane q there is the event of rank q, and
        // only a handful of scalars (l, l̄, |G1|, the lattice coordinates, t′) travel there, the outcomes travel back.
        const bool evc0 = isev && rank < (uint32_t)C;  // (rank < C: the records were requested at the right position)
        const uint32_t i = ci;
        const double tp = c_km;
        const double rest = isc ? RS[lane] : X_INF;
        const uint32_t rarg = cblk * 16u + (c_pb >> 4);
        const double th = cS0.y;
        const double told_i = cS2.x, a_i = cS2.y, b_i = b_own;
        const uint32_t col_i = ccol, row_i = crow;
        const uint32_t rc_c = evc0 ? (row_i | (col_i << 8)) : 0xffffu;  // lattice coordinates packed for the zone test
        const bool hasL = col_i > 0u, hasU = row_i > 0u, hasD = row_i + 1u < nlat, hasR = col_i + 1u < nlat;
        const uint32_t k_c = 1u + (hasL ? 1u : 0u) + (hasU ? 1u : 0u) + (hasD ? 1u : 0u) + (hasR ? 1u : 0u);
        // ---------------- smove_forward!(G, i, ...) (:82), ∇ϕ = idot(Γ, i, x) in ascending row order (:116), the rates (:119).  Only the two
        // sums, the rates and the would-be bound of a rejection are kept: the moved records themselves are formed again at commit time from a
        // second (cache-hot) read -- twenty doubles less to hold across the accept chain and the groups' work.
        double gsum = 0.0, s2 = 0.0, a2 = 0.0, b2 = 0.0;
        double l_c, lb_c;
        {
            uint32_t q = 0;
            auto gam_at = [&](uint32_t q_) -> double {  // Γ[G1[i][q], i]: a select among registers
                const double lo = (q_ == 0u) ? kB.y : ((q_ == 1u) ? kC.x : kC.y);
                const double hi = (q_ == 3u) ? kD.x : kD.y;
                return (q_ < 3u) ? lo : hi;
            };
#define X_TERM(HAS, V0, V1)                                      \
    do {                                                         \
        if (HAS) {                                               \
            const double nx_ = (V0).x + (V0).y * (tp - (V1).x);  \
            const double w_ = gam_at(q);                         \
            gsum += w_ * nx_;                                    \
            s2 += w_ * (V0).y;                                   \
            q += 1u;                                             \
        }                                                        \
    } while (0)
            X_TERM(hasL, cL0, cL1);
            X_TERM(hasU, cU0, cU1);
            X_TERM(true, cS0, cS1);
            X_TERM(hasD, cD0, cD1);
            X_TERM(hasR, cR0, cR1);
#undef X_TERM
            l_c = x_pos(gsum * th);
            lb_c = x_pos(a_i + b_i * (tp - told_i));
            a2 = kA.x + gsum * th;  // ab of a rejected proposal (src/fact_samplers.jl:50-54)
            b2 = kA.y + th * s2;
        }
        X_PH(4);
        // ---------------- rank space
        const bool evq0 = lane < C;
        const uint32_t srcq = evq0 ? (uint32_t)RO[lane] : 0u;
        const double l = x_shfl(l_c, srcq), lbound = x_shfl(lb_c, srcq);
        const uint32_t k_i = x_shfl_u32(k_c, srcq);
        const uint32_t rc_b = x_shfl_u32(rc_c, srcq);
        const uint32_t rc_i = evq0 ? rc_b : 0xffffu;
        const double tpq = evq0 ? KM[srcq] : X_INF;
        bool ev = evq0;
        // accept chain: offsets and outcomes as a fix-point
        uint32_t cost = ev ? 2u : 0u;
        uint32_t off = 0;
        bool acc = false;
        for (int round = 0; round < 66; ++round) {
            const uint32_t incl = x_scan_add_u32(cost);
            off = incl - cost;
            const bool inwin = ev && (off + 1u + k_i <= X_WIN);
            const double u = draw(dnm + ((off < 127u) ? off : 127u));
            acc = inwin && (u * lbound < l);  // :121
            const uint32_t nc = ev ? (acc ? (1u + k_i) : 2u) : 0u;
            const bool changed = nc != cost;
            cost = nc;
            if (__ballot(changed) == 0) break;
        }
        {
            const uint64_t outb = __ballot(ev && !(off + 1u + k_i <= X_WIN));
            if (outb) {
                const int cut = __ffsll((unsigned long long)outb) - 1;
                C = (cut < C) ? cut : C;
            }
        }
        {
            uint64_t ab = __ballot(acc) & ((C < 64) ? ((1ull << C) - 1ull) : ~0ull);
            if (__popcll(ab) > X_AMAX) {
                uint64_t m_ = ab;
                for (int q = 0; q < X_AMAX; ++q) m_ &= m_ - 1;
                C = __ffsll((unsigned long long)m_) - 1;
            }
        }
        // zones.  A rejected proposal writes G1[i] (radius 1), an accepted one S[i] (radius 2): a later event r is disturbed by an earlier m
        // iff their lattice distance is <= radius(m) + radius(r).  The list ends at the first disturbed event.
        {
            const uint64_t accb = __ballot(acc);
            const uint32_t myrad = acc ? 2u : 1u;
            uint64_t confb = 0;
            for (int m = 0; m < C; ++m) {
                const uint32_t rcm = (uint32_t)__builtin_amdgcn_readlane((int)rc_i, m);
                const uint32_t radm = ((accb >> m) & 1ull) ? 2u : 1u;
                const uint32_t sad = __builtin_amdgcn_sad_u8(rc_i, rcm, 0u);
                confb |= __ballot(lane > m && sad <= radm + myrad);
            }
            const uint64_t cb = confb & ((C < 64) ? ((1ull << C) - 1ull) : ~0ull);
            if (cb) {
                const int c0 = __ffsll((unsigned long long)cb) - 1;
                C = (c0 < C) ? c0 : C;
            }
        }
        // a proposal that violates its bound ends the run (adapt = false: error(...), :124): nothing after it is looked at
        const bool violated0 = acc && (l >= lbound);
        int vsel = -1;
        {
            const uint64_t vb = __ballot(violated0) & ((C < 64) ? ((1ull << C) - 1ull) : ~0ull);
            if (vb) {
                vsel = __ffsll((unsigned long long)vb) - 1;
                C = vsel;
            }
        }
        ev = lane < C;
        acc = acc && ev;
        st_zone += (uint32_t)C;
        const uint64_t accball = __ballot(acc);  // (bit q: the event of RANK q is accepted)
        const int nacc_it = __popcll(accball);
        if (acc) ACL[__popcll(accball & ((1ull << lane) - 1ull))] = (uint16_t)lane;
        X_ORDER();
        // ---------------- back in the candidates' lanes: the outcome of the own event, its draw offset
        const uint32_t off_c = x_shfl_u32(off, (rank < 64u) ? rank : 0u);
        const bool evc = isev && rank < (uint32_t)C;
        const bool acc_c = evc && ((accball >> rank) & 1ull) != 0;
        // a rejected proposal's new bound (:137-140): ab from the moved neighbourhood (src/fact_samplers.jl:50-54)
        double key2 = X_INF;
        {
            const double ur = draw(dnm + ((off_c + 1u < 127u) ? off_c + 1u : 127u));  // (every lane takes part in the ring's ds_bpermute)
            if (evc && !acc_c) key2 = tp + x_poisson_time_L(a2, b2, pdmp_log(ur));
        }
        X_PH(5);
        // ---------------- accepted events, one 8-lane group each (the LAST nacc_it groups of the wave, in event order)
        const int g0 = 8 - nacc_it;
        const bool gact = g >= g0;
        const uint32_t ea = gact ? (uint32_t)ACL[g - g0] : 0u;  // rank of the group's event
        const uint32_t la = (uint32_t)RO[ea];                    // ... and the lane of its candidate
        const uint32_t ia_b = x_shfl_u32(i, la);  // (cross-lane reads sit in wave-uniform control flow: a disabled source lane reads as 0)
        const uint32_t ia = gact ? ia_b : 0u;
        const uint32_t offa = x_shfl_u32(off, ea);
        const uint32_t blka_b = x_shfl_u32(cblk, la);
        const uint32_t blka = gact ? blka_b : 0u;
        const double tpa = x_shfl(tp, la);
        const double resta_b = x_shfl(rest, la);
        const uint32_t rarga_b = x_shfl_u32(rarg, la);
        const uint32_t cola = __umulhi(ia, nmagic), rowa = ia - cola * nlat;
        // the 13 members of S[ia], two per lane: window offsets (dc, dr) with |dc| + |dr| <= 2, inside the grid.  Everything is COMPUTED here
        // (records read, moved values in registers, (x, θ) of the window staged in LDS, the members' new bounds and keys); nothing is stored
        // before the commit prefix is known.
        double keyj = X_INF, aj = 0.0, bj = 0.0;
        uint32_t jmem = 0;
        bool memb = false;
        double zx[2] = {0.0, 0.0}, zI[2] = {0.0, 0.0}, zth = 0.0;
        uint32_t zj[2] = {0u, 0u};
        bool zin[2] = {false, false};
        uint64_t zacc = 0;
        {
            // offsets in the order L, U, self, D, R (G1, ascending coordinate), then G2 -- (dc + 2, dr + 2), a nibble per member
            const uint64_t DCP = 0x4332211032221ull, DRP = 0x2314031223212ull;
            // members of G1[ia] (lanes 0..4 of the group: L, U, self, D, R where present): their constants are requested first
            const int glc = (gl < 5) ? gl : 2;
            const int dcm = (int)((DCP >> (4 * glc)) & 15u) - 2, drm = (int)((DRP >> (4 * glc)) & 15u) - 2;
            const int cj = (int)cola + dcm, rj = (int)rowa + drm;
            memb = gact && gl < 5 && cj >= 0 && cj < (int)nlat && rj >= 0 && rj < (int)nlat;
            double2 jA = make_double2(0.0, 0.0), jB = jA, jC = jA, jD = jA;
            if (memb) {
                jmem = (uint32_t)(rj + cj * (int)nlat);
                const double2* const kp = reinterpret_cast<const double2*>(cc + jmem);
                jA = kp[0];
                jB = kp[1];
                jC = kp[2];
                jD = kp[3];
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int p = gl + 8 * h;
                const int pc = (p < 13) ? p : 2;
                const int dc = (int)((DCP >> (4 * pc)) & 15u) - 2, dr = (int)((DRP >> (4 * pc)) & 15u) - 2;
                const int cz = (int)cola + dc, rz = (int)rowa + dr;
                zin[h] = gact && p < 13 && cz >= 0 && cz < (int)nlat && rz >= 0 && rz < (int)nlat;
                if (zin[h]) {
                    zj[h] = (uint32_t)(rz + cz * (int)nlat);
                    const ZzRec* const rz_ = rec + zj[h];
                    const double2 v0 = *reinterpret_cast<const double2*>(&rz_->x), v1 = *reinterpret_cast<const double2*>(&rz_->t);
                    const double dt_ = tpa - v1.x;
                    const double xn = v0.x + v0.y * dt_;  // smove_forward!(G, ...) / (G2, ...), :82,129
                    zx[h] = xn;
                    zI[h] = v1.y + dt_ * ((v0.x + xn) * 0.5);
                    const bool self = (p == 2);
                    if (self) {
                        zth = -v0.y;  // reflect!, :130
                        zacc = rz_->acc;
                    }
                    ZS[(dc + 2) * 5 + (dr + 2)] = make_double2(xn, self ? -v0.y : v0.y);
                }
            }
            X_ORDER();
            // Γ[:,j]·x, Γ[:,j]·θ over G1[j] in ascending order, from the staged window
            // rank of this member among the present ones = its draw (:131-135: one uniform per member, ascending)
            const uint32_t rk = (uint32_t)((gl > 0 && cola > 0u) ? 1 : 0) + (uint32_t)((gl > 1 && rowa > 0u) ? 1 : 0) + (uint32_t)((gl > 2) ? 1 : 0) +
                                (uint32_t)((gl > 3 && rowa + 1u < nlat) ? 1 : 0);
            const uint32_t dix = offa + 1u + rk;
            const double uj = draw(dnm + ((dix < 127u) ? dix : 127u));
            if (memb) {
                double s1j = 0.0, s2j = 0.0;
                uint32_t q = 0;
                const int NC[5] = {-1, 0, 0, 0, 1}, NR[5] = {0, -1, 0, 1, 0};
#pragma unroll
                for (int e = 0; e < 5; ++e) {
                    const int c2 = cj + NC[e], r2 = rj + NR[e];
                    if (c2 >= 0 && c2 < (int)nlat && r2 >= 0 && r2 < (int)nlat) {
                        const double2 v = ZS[(dcm + NC[e] + 2) * 5 + (drm + NR[e] + 2)];
                        const double lo = (q == 0u) ? jB.y : ((q == 1u) ? jC.x : jC.y);
                        const double hi = (q == 3u) ? jD.x : jD.y;
                        const double w = (q < 3u) ? lo : hi;
                        s1j += w * v.x;
                        s2j += w * v.y;
                        q += 1u;
                    }
                }
                const double thj = ZS[(dcm + 2) * 5 + (drm + 2)].y;
                aj = jA.x + s1j * thj;  // src/fact_samplers.jl:51
                bj = jA.y + thj * s2j;  // :52
                keyj = tpa + x_poisson_time_L(aj, bj, pdmp_log(uj));
            }
        }
        // new minimum of the popped block of a rejected event, and what the event exposes
        double rowmin = X_INF;
        uint32_t cand = i;
        if (evc && !acc_c) {
            const bool mine = key2 < rest || (key2 == rest && i < rarg);
            rowmin = mine ? key2 : rest;
            cand = mine ? i : rarg;
        }
        // the accepted event's block: a LOWER BOUND of its new minimum is enough (level 1 holds bounds): the smaller of the block without the
        // event -- which may still count a member's OLD key: then the bound is stale low and costs a look later -- and the members' new keys in it
        double rowmin_a = X_INF;
        uint32_t cand_a = 0;
        {
            const double kin = (memb && (jmem >> 4) == blka) ? keyj : X_INF;
            const double kinmin = x_grp8_min(kin);
            const uint64_t winball = __ballot(gact && kin == kinmin);
            const int wl = __ffs((unsigned)((winball >> (8 * g)) & 0xffu)) - 1;
            const uint32_t jwin = x_shfl_u32(jmem, (uint32_t)(lane & ~7) + (uint32_t)(wl < 0 ? 0 : wl));
            const bool restwins = resta_b <= kinmin;
            rowmin_a = restwins ? resta_b : kinmin;
            cand_a = restwins ? (rarga_b & 15u) : (jwin & 15u);
            const double keymin = x_grp8_min(memb ? keyj : X_INF);
            if (gact && gl == 0) EX[ea] = x_min(rowmin_a, keymin);
        }
        if (evc && !acc_c) EX[rank] = rowmin;  // (by rank; rowmin <= key2: the new key is one of its candidates)
        X_ORDER();
        X_PH(6);
        // ---------------- validate: nothing produced or exposed by the earlier events comes before t′ (zone conflicts ended the list already)
        uint32_t Rc;
        {
            const double expo = ev ? EX[lane] : X_INF;
            const double prev = x_shfl(expo, (uint32_t)((lane > 0) ? lane - 1 : 0));
            const double pref = x_scan_min_f64((lane > 0) ? prev : X_INF);  // exclusive prefix minimum
            const bool okr = ev && (lane == 0 || pref > tpq);
            const uint64_t bad = ~__ballot(okr);
            const uint32_t r_ok = bad ? (uint32_t)(__ffsll((unsigned long long)bad) - 1) : 64u;
            Rc = (r_ok < (uint32_t)C) ? r_ok : (uint32_t)C;
            if (vsel != (int)Rc) vsel = -1;  // the violating proposal counts only once everything before it is committed
            // the trace's room and the end of the run (`while t′ < T` looks at accepted events only, :199)
            const uint64_t accc = accball & ((Rc < 64u) ? ((1ull << Rc) - 1ull) : ~0ull);
            uint64_t walk = accc;
            uint32_t na = 0;
            bool stopped = false;
            while (walk && !stopped) {
                const int r = __ffsll((unsigned long long)walk) - 1;
                walk &= walk - 1;
                na += 1;
                if (P.trace_cap > 0 && dnacc + na >= trace_room) {
                    status = PDMP_CHAIN_TRACE_FULL;
                    stopped = true;
                }
                if (!stop_before && !(x_readlane(tpq, r) < T)) {
                    running = false;
                    stopped = true;
                }
                if (stopped) Rc = (uint32_t)r + 1u;
            }
            if (stopped) vsel = -1;
        }
        if (lane == 0) SELDT[0] = dt_used * (((int)Rc >= Craw) ? X_GROW : (((int)Rc + (int)X_SLACK < Craw) ? X_SHRINK : 1.0));
        // ---------------- commit the valid prefix
        const bool commit = evc && rank < Rc;
        // the moved neighbourhood of a proposal (:82): x_j += θ_j (t′ − t_j), t_j = t′ (and the engine's ∫x dt), from a second read of the records
        auto store_moved = [&](uint32_t j) {
            ZzRec* const w = rec + j;
            const double2 v0 = *reinterpret_cast<const double2*>(&w->x), v1 = *reinterpret_cast<const double2*>(&w->t);
            const double dt_ = tp - v1.x;
            const double xn = v0.x + v0.y * dt_;
            w->x = xn;
            *reinterpret_cast<double2*>(&w->t) = make_double2(tp, v1.y + dt_ * ((v0.x + xn) * 0.5));
        };
        if (commit && !acc_c) {  // a rejected proposal: the moved neighbourhood (:82), its new bound and key (:137-140)
            if (hasL) store_moved(i - nlat);
            if (hasU) store_moved(i - 1u);
            store_moved(i);
            if (hasD) store_moved(i + 1u);
            if (hasR) store_moved(i + nlat);
            ZzRec* const wS = rec + i;
            *reinterpret_cast<double2*>(&wS->t_old) = make_double2(tp, a2);
            wS->b = b2;
            keys[i] = key2;
            lbf[cblk] = (rowmin < X_INF) ? q_enc(rowmin, tb, cand & 15u) : Q_INFBITS;
        }
        const bool gcommit = gact && ea < Rc;
        const uint64_t acc_cm = accball & ((Rc < 64u) ? ((1ull << Rc) - 1ull) : ~0ull);
        if (gcommit) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (zin[h]) {  // S[ia] at t′
                    ZzRec* const w = rec + zj[h];
                    w->x = zx[h];
                    *reinterpret_cast<double2*>(&w->t) = make_double2(tpa, zI[h]);
                }
            }
            if (gl == 2) {  // (the lane that holds the event's own coordinate: window position 2)
                ZzRec* const w = rec + ia;
                w->th = zth;
                w->acc = zacc + 1;
                if (evout) {
                    const uint32_t rnk = (uint32_t)__popcll(acc_cm & ((1ull << ea) - 1ull));
                    pdmp_event e;  // event(i, t, x, θ, F) = (t[i], i, x[i], θ[i]), src/sfact.jl:50-52
                    e.t = tpa;
                    e.i = (int64_t)ia;
                    e.x = zx[0];
                    e.theta = zth;
                    evout[ntrace0 + dnacc + rnk] = e;
                }
            }
            if (memb) {  // b[j], t_old[j], Q[j] of the members (:131-135)
                ZzRec* const w = rec + jmem;
                *reinterpret_cast<double2*>(&w->t_old) = make_double2(tpa, aj);
                w->b = bj;
                keys[jmem] = keyj;
            }
            if (gl == 0) lbf[blka] = (rowmin_a < X_INF) ? q_enc(rowmin_a, tb, cand_a) : Q_INFBITS;
        }
        X_ORDER();
        // ---------------- bounds of the blocks of re-bounded neighbours: lowered where the new key is below them (an LDS atomic minimum); a key
        // that ROSE leaves its block's bound stale low, which costs a look at the block later and nothing else
        {
            const bool upd = gcommit && memb && (jmem >> 4) != blka;
            if (upd && keyj < X_INF) atomicMin(&lbf[jmem >> 4], q_enc(keyj, tb, jmem & 15u));
        }
        X_ORDER();
        X_PH(7);
        // ---------------- counters; the violating proposal itself (counted, acc bumped, then error(...), :120-124)
        st_commit += Rc;
        if (Rc > 0u) {
            const uint32_t costL = (uint32_t)__builtin_amdgcn_readlane((int)cost, (int)(Rc - 1u));
            const uint32_t offL = (uint32_t)__builtin_amdgcn_readlane((int)off, (int)(Rc - 1u));
            dnum += Rc;
            idle = 0;
            dnacc += (uint32_t)__popcll(acc_cm);
            dnm += offL + costL;
            t_last = x_readlane(tpq, (int)(Rc - 1u));
            if (acc_cm) t_event = x_readlane(tpq, 63 - __builtin_clzll(acc_cm));
        }
        if (vsel >= 0) {  // (vsel == Rc: every earlier event is committed)
            // the reference moved G[i] (:82) and counted the proposal before it threw (:120-124); acc[i] -- bumped there too, and gone with
            // the exception -- stays as it is, as in the other kernels (the chain's nacc counts it)
            if (isev && rank == (uint32_t)vsel) {
                if (hasL) store_moved(i - nlat);
                if (hasU) store_moved(i - 1u);
                store_moved(i);
                if (hasD) store_moved(i + 1u);
                if (hasR) store_moved(i + nlat);
            }
            dnum += 1;
            vnacc = 1;
            dnm += 1;  // its coin
            t_last = x_readlane(tpq, vsel);
            status = PDMP_CHAIN_BOUND_VIOLATED;
        }
        if (status != PDMP_CHAIN_OK) break;
        X_ORDER();
    }

    if (P.dbg && chain == 0 && lane == 0) {
        P.dbg[10] = (double)st_iters;
        P.dbg[11] = (double)st_raw;
        P.dbg[12] = (double)st_zone;
        P.dbg[13] = (double)st_commit;
        P.dbg[14] = (double)st_events;
#ifdef X_PHASES
        for (int k = 0; k < 9; ++k) P.dbg[k] = (double)ph[k];
#endif
    }
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

bool zz_exactp_supported(const ZzRunParams& p) {
    return p.lattice_n >= 16 && p.lattice_n <= 128 && !p.adapt && p.c_chain == nullptr && p.tb.gmu_t == nullptr && !p.track_two_sums &&
           !p.has_refresh && !p.move_all && p.d >= 2048 && p.d <= (int64_t)X_NBLK * 16 && p.tb.cc_shared != nullptr;
}

int launch_zz_local_exactp(const ZzRunParams& p, int64_t nchains, void* stream) {
    dim3 grid((unsigned)nchains), block(64);
    ZzRunParams q = p;
    q.nblk = (uint32_t)((p.d + 15) / 16);  // (dk is a multiple of 64, the padding keys are +Inf)
    hipLaunchKernelGGL(zz_local_exactp_kernel, grid, block, X_BYTES, (hipStream_t)stream, q);
    return (int)hipGetLastError();
}

}  // namespace pdmp
