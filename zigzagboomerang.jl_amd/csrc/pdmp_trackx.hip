// pdmp_trackx.hip -- zz_local_trackx_kernel: pdmp_trackw.hip's one-proposal-per-lane tracked-gradient kernel over key blocks of SIXTEEN.
//
// What limits pdmp_trackw.hip is how many candidates can commit per iteration: only block MINIMA are candidates, so the second key of a
// popped block that falls inside the candidate window ends the committable prefix -- a birthday bound near sqrt(#blocks) ~ 20 events with 512
// blocks of 32 keys -- and every candidate reads two 128-byte lines of keys.  Here the queue's first level has 1024 entries over blocks of 16
// keys (ONE line per block scan, twice the blocks).  1024 doubles are 8 of the 10 KB of LDS a chain may use, so everything else left LDS:
//   * the ring of uniforms lives in two registers per lane (128 draws; a draw is fetched from its lane with ds_bpermute)
//   * candidates are ranked by reading each other's keys with v_readlane, the event times are read back from the first level itself
//   * block-scan results travel from the 8-lane groups to the event lanes through ds_bpermute, not through LDS arrays
//   * the first level stores the argument of a block minimum as a 4-bit position (one byte)
// Everything else -- evaluation, fix-point accept chain, accepted events in 8-lane groups, validation, commit -- is pdmp_trackw.hip's, and the
// committed sequence is the same (index-exact against the oracle, floats to ~1e-13).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/pdmp_detmath.h"
#include "pdmp_engine.hpp"

namespace pdmp {

#define W_INF __builtin_inf()
#define W_ORDER()                        \
    do {                                 \
        __builtin_amdgcn_wave_barrier(); \
        asm volatile("" ::: "memory");   \
    } while (0)

namespace {

__device__ __forceinline__ double w_readlane(double v, int srclane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double w_uniform(double v) {
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double w_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double w_min(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double w_wave_min(double v) {
    v = w_min(v, w_dpp<0xB1>(v));
    v = w_min(v, w_dpp<0x4E>(v));
    v = w_min(v, w_dpp<0x141>(v));
    v = w_min(v, w_dpp<0x140>(v));
    v = w_min(v, w_dpp<0x142>(v));
    v = w_min(v, w_dpp<0x143>(v));
    return w_readlane(v, 63);
}
// minimum over the 8 lanes of a group, in every lane of the group
__device__ __forceinline__ double w_grp8_min(double v) {
    v = w_min(v, w_dpp<0xB1>(v));
    v = w_min(v, w_dpp<0x4E>(v));
    v = w_min(v, w_dpp<0x141>(v));
    return v;
}
__device__ __forceinline__ double w_pos(double x) {
    return (x > 0.0) ? x : ((x != x) ? x : 0.0);
}
__device__ __forceinline__ double w_poisson_time_L(double a, double b, double L) {  // src/poissontime.jl:8-30 with L = log(u)
    if (b == 0) return (a > 0) ? -L / a : W_INF;
    const double r = a / b;
    const double q = L * 2.0 / b;
    const double sq = sqrt((b > 0 && a < 0) ? -q : r * r - q);
    if (b > 0) return sq - r;
    if (a <= 0) return W_INF;
    if (-L <= -(a * a) / b + (a * a) / (2 * b)) return -sq - r;
    return W_INF;
}
__device__ __forceinline__ double w_below(double x) {  // the largest double below a finite x
    long long b = __double_as_longlong(x);
    if (x > 0) b -= 1;
    else if (x < 0) b += 1;
    else b = (long long)0x8000000000000001ull;
    return __longlong_as_double(b);
}

// DPP prefix operations over the 64 lanes (row_shr 1, 2, 3 of the input, then row_shr 4 / 8 of the partial result inside the enabled banks,
// then row_bcast 15 / 31 across the rows): lanes without a source keep the identity.
template <int CTRL, int ROWM, int BANKM>
__device__ __forceinline__ uint32_t w_dpp_id_u32(uint32_t identity, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)src, CTRL, ROWM, BANKM, false);
}
__device__ __forceinline__ uint32_t w_scan_add_u32(uint32_t v) {  // inclusive
    uint32_t x = v;
    x += w_dpp_id_u32<0x111, 0xf, 0xf>(0u, v);
    x += w_dpp_id_u32<0x112, 0xf, 0xf>(0u, v);
    x += w_dpp_id_u32<0x113, 0xf, 0xf>(0u, v);
    x += w_dpp_id_u32<0x114, 0xf, 0xe>(0u, x);
    x += w_dpp_id_u32<0x118, 0xf, 0xc>(0u, x);
    x += w_dpp_id_u32<0x142, 0xa, 0xf>(0u, x);
    x += w_dpp_id_u32<0x143, 0xc, 0xf>(0u, x);
    return x;
}
template <int CTRL, int ROWM, int BANKM>
__device__ __forceinline__ double w_dpp_inf(double src) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(src), CTRL, ROWM, BANKM, false);
    const int hi = __builtin_amdgcn_update_dpp(0x7FF00000, __double2hiint(src), CTRL, ROWM, BANKM, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double w_scan_min_f64(double v) {  // inclusive
    double x = v;
    x = w_min(x, w_dpp_inf<0x111, 0xf, 0xf>(v));
    x = w_min(x, w_dpp_inf<0x112, 0xf, 0xf>(v));
    x = w_min(x, w_dpp_inf<0x113, 0xf, 0xf>(v));
    x = w_min(x, w_dpp_inf<0x114, 0xf, 0xe>(x));
    x = w_min(x, w_dpp_inf<0x118, 0xf, 0xc>(x));
    x = w_min(x, w_dpp_inf<0x142, 0xa, 0xf>(x));
    x = w_min(x, w_dpp_inf<0x143, 0xc, 0xf>(x));
    return x;
}
__device__ __forceinline__ double w_shfl(double v, uint32_t src) {
    const int lo = __builtin_amdgcn_ds_bpermute((int)(src << 2), __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute((int)(src << 2), __double2hiint(v));
    return __hiloint2double(hi, lo);
}

}  // namespace

// LDS layout (bytes)
constexpr uint32_t W_BK = 0;         // [1024] f64 block minima (first level of the queue, key blocks of 16)
constexpr uint32_t W_BI = 8192;      // [1024] u8 position of the minimum inside its block
constexpr uint32_t W_EX = 9216;      // [64] f64 what event e exposes
constexpr uint32_t W_SLB = 9728;     // [64] u16 event blocks, rank order
constexpr uint32_t W_TB = 9856;      // [64] u16 candidate blocks, compaction order
constexpr uint32_t W_ACL = 9984;     // [8] u16 the accepted events
constexpr uint32_t W_RO = 10000;     // [64] u8 owner of each rank (duplicate detection); later the claims of the parallel first-level update
constexpr uint32_t W_CL = W_RO;
constexpr uint32_t W_SELDT = 10064;  // f64 selection threshold above the minimum
constexpr uint32_t W_BYTES = 10072;
constexpr uint32_t W_NBLK = 1024;
constexpr uint32_t W_WIN = 128;      // draws held in registers (two per lane)
constexpr int W_CMAX = 56;           // candidates per iteration (7 block-scan passes of 8)
constexpr int W_AMAX = 8;            // accepted events per iteration (one group each)
// steering of the selection threshold (measured: 1.3 / 0.85 / 6 is 3 % slower, 1.1 / 0.7 / 2 as well)
#ifndef W_GROW
#define W_GROW 1.15
#define W_SHRINK 0.8
#define W_SLACK 3u
#endif
static_assert(W_BYTES <= 10240, "16 chains per CU: 160 KB / 16");

template <bool PROF>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void zz_local_trackx_kernel(ZzRunParams P) {
    const int lane = threadIdx.x;
    const int g = lane >> 3, gl = lane & 7;
    const int64_t chain = blockIdx.x;
    const int64_t d = P.d;
    const uint32_t nblk = P.nblk;
    const uint32_t nlat = (uint32_t)P.lattice_n, nmagic = P.lattice_magic;

    extern __shared__ __align__(16) unsigned char smem[];
    double* const bk = reinterpret_cast<double*>(smem + W_BK);
    uint8_t* const bi = reinterpret_cast<uint8_t*>(smem + W_BI);
    double* const EX = reinterpret_cast<double*>(smem + W_EX);
    uint16_t* const SLB = reinterpret_cast<uint16_t*>(smem + W_SLB);
    uint16_t* const TB = reinterpret_cast<uint16_t*>(smem + W_TB);
    uint16_t* const ACL = reinterpret_cast<uint16_t*>(smem + W_ACL);
    uint8_t* const RO = reinterpret_cast<uint8_t*>(smem + W_RO);
    uint8_t* const CL = reinterpret_cast<uint8_t*>(smem + W_CL);
    double* const SELDT = reinterpret_cast<double*>(smem + W_SELDT);

    TrRec* const rec = reinterpret_cast<TrRec*>(P.rec) + chain * d;
    double* const keys = P.keys + chain * P.dk;
    DevChain* const hdr = P.hdr + chain;
    pdmp_event* const evout = P.ev ? P.ev + chain * P.trace_cap : nullptr;

    uint32_t status = hdr->c.status;
    if (status == PDMP_CHAIN_BOUND_VIOLATED || status == PDMP_CHAIN_STALLED) return;
    const uint64_t seed = hdr->seed;
    const uint64_t nm0 = hdr->c.ndraw_main, ntrace0 = hdr->c.ntrace;
    uint32_t dnm = 0, dnum = 0, dnacc = 0, vnacc = 0;
    // ring of uniforms in registers: ureg[q] holds draw nm0 + uidx[q], the unique index n in [dnm, dnm + 128) with n % 64 == lane and (n / 64) % 2 == q
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
    for (uint32_t b = lane; b < nblk; b += 64) {
        const double* kp = keys + (size_t)b * 16;
        double mk = kp[0];
        uint32_t mi = 0;
#pragma unroll 8
        for (int q = 1; q < 16; ++q) {
            const double v = kp[q];
            if (v < mk) {
                mk = v;
                mi = q;
            }
        }
        bk[b] = mk;
        bi[b] = (uint8_t)mi;
    }
    for (uint32_t b = nblk + lane; b < W_NBLK; b += 64) {
        bk[b] = W_INF;
        bi[b] = 0;
    }
    W_ORDER();

    uint64_t ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t ph_t0 = PROF ? (uint64_t)__builtin_readcyclecounter() : 0;
    uint64_t ph_iters = 0, ph_raw = 0, ph_zone = 0, ph_eval = 0;  // candidates: selected, after the zone cut, after the window and accept limits
#define WPHASE(k)                                                         \
    do {                                                                  \
        if (PROF) {                                                       \
            const uint64_t now_ = (uint64_t)__builtin_readcyclecounter(); \
            ph[k] += now_ - ph_t0;                                        \
            ph_t0 = now_;                                                 \
        }                                                                 \
    } while (0)

    PrioTurn prio;
    bool running = stop_before || (t_event < T);
    while (running) {
        prio.step();
        if (dnacc >= trace_room) {
            status = PDMP_CHAIN_TRACE_FULL;
            break;
        }
        // ---------------- ring of uniforms: draws dnm .. dnm + 127, two per lane
        {
            const uint32_t n0 = dnm + (((uint32_t)lane - dnm) & 63u);  // the smallest n >= dnm with n % 64 == lane
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
        WPHASE(7);
        auto draw = [&](uint32_t n) -> double {  // draw nm0 + n for dnm <= n < dnm + 128 (every lane calls it: ds_bpermute)
            const double v0 = w_shfl(ureg[0], n & 63u), v1 = w_shfl(ureg[1], n & 63u);
            return ((n >> 6) & 1u) ? v1 : v0;
        };
        // ---------------- select: every first-level entry <= m + sel_dt, at most W_CMAX of them
        int C = 0;
        bool first_inf = false;
        double dt_used = 0.0;
        {
            double kk[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) kk[j] = bk[lane + 64 * j];
            double mloc = kk[0];
#pragma unroll
            for (int j = 1; j < 16; ++j) mloc = w_min(mloc, kk[j]);
            const double mq = w_wave_min(mloc);
            if (!(mq < W_INF)) {
                first_inf = true;
            } else if (!(stop_before && !(mq < T))) {
                double dt_sel = w_uniform(SELDT[0]);
                // per lane: bit j of cm = first-level entry lane + 64 j is a candidate; the threshold halves until at most W_CMAX are
                uint32_t cm = 0, ncl = 0, incl = 0, Cc = 0;
                bool tied = false;
                for (int tries = 0;; ++tries) {
                    double tau = mq + dt_sel;
                    if (stop_before && !(tau < T)) tau = w_below(T);
                    if (tries >= 64) tau = mq;
                    cm = 0;
#pragma unroll
                    for (int j = 15; j >= 0; --j) cm = cm + cm + ((kk[j] <= tau) ? 1u : 0u);
                    ncl = (uint32_t)__builtin_popcount(cm);
                    incl = w_scan_add_u32(ncl);
                    Cc = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    if (Cc <= (uint32_t)W_CMAX) break;
                    if (tries >= 64) {  // more than W_CMAX keys EQUAL the minimum
                        tied = true;
                        break;
                    }
                    dt_sel *= 0.5;
                }
                if (tied) {
                    cm = 0;
                    ncl = 0;
                    incl = 0;
                    Cc = 0;
                }
                {
                    uint32_t ix = incl - ncl, m_ = cm;
                    while (__ballot(m_ != 0u) != 0) {
                        if (m_ != 0u) {
                            TB[ix] = (uint16_t)((uint32_t)lane + 64u * (uint32_t)(__ffs((int)m_) - 1));
                            ix += 1;
                            m_ &= m_ - 1u;
                        }
                    }
                }
                W_ORDER();
                WPHASE(8);
                // rank of candidate `lane` among all: the number of strictly smaller keys, read lane by lane (no LDS)
                const bool isc = (uint32_t)lane < Cc;
                const uint32_t myb = isc ? (uint32_t)TB[lane] : 0u;
                const double own = isc ? bk[myb] : W_INF;
                uint32_t rank = 0;
                for (uint32_t m0 = 0; m0 < Cc; m0 += 4) {  // (lanes past the candidates hold +Inf: reading them changes nothing)
#pragma unroll
                    for (uint32_t q = 0; q < 4; ++q) {
                        const double km = w_readlane(own, (int)(m0 + q));
                        rank += (km < own) ? 1u : 0u;
                    }
                }
                if (isc) RO[rank] = (uint8_t)lane;
                W_ORDER();
                const bool dup = isc && RO[rank] != (uint8_t)lane;
                if (tied || __ballot(dup) != 0) {
                    // exactly equal keys among the candidates (probability zero unless keys are tied by construction): one event this
                    // iteration, the tied minimum of the lowest block
                    uint32_t bsel = 0xffffffffu;
#pragma unroll
                    for (int j = 15; j >= 0; --j) bsel = (kk[j] == mq) ? ((uint32_t)lane + 64u * (uint32_t)j) : bsel;
                    for (int off = 32; off >= 1; off >>= 1) {
                        const uint32_t o = (uint32_t)__shfl_xor((int)bsel, off, 64);
                        bsel = (o < bsel) ? o : bsel;
                    }
                    W_ORDER();
                    if (lane == 0) SLB[0] = (uint16_t)bsel;
                    Cc = 1;
                } else if (isc) {
                    SLB[rank] = (uint16_t)myb;
                }
                WPHASE(9);
                C = (int)Cc;
                dt_used = dt_sel;
            }
        }
        if (C == 0) {
            if (first_inf) status = PDMP_CHAIN_STALLED;
            break;
        }
        W_ORDER();
        WPHASE(0);
        if (PROF) ph_iters += 1;
        if (PROF) ph_raw += (uint64_t)C;
        const int Craw = C;

        // ---------------- lane r = event r: own record, neighbourhood size, c_i
        bool ev = lane < C;
        const uint32_t blk = ev ? (uint32_t)SLB[lane] : 0u;
        const double tp = ev ? bk[blk] : W_INF;  // the event time IS the block minimum
        const uint32_t i = ev ? (blk * 16u + (uint32_t)bi[blk]) : 0u;
        const TrRec* const ri = rec + i;
        const double th = ri->th;
        const double g_i = ri->g, gd_i = ri->gd, tg_i = ri->tg;
        const double told_i = ri->t_old, a_i = ri->a, b_i = ri->b;
        const double2 c_i2 = *reinterpret_cast<const double2*>(&P.tb.cc_shared[i].c);
        const double c_i = c_i2.x;
        const uint32_t k_i = P.tb.cc_shared[i].k;
        // lattice coordinates packed for the zone test: byte 0 = row, byte 1 = column
        const uint32_t col_i = __umulhi(i, nmagic);
        const uint32_t rc_i = ev ? ((i - col_i * nlat) | (col_i << 8)) : 0xffffu;
        // ---------------- (the own-record loads are in flight) zones: event r cannot commit with an earlier event whose G1 meets its own
        // (Manhattan distance of the lattice coordinates <= 2): the candidate list ends at the first such event, BEFORE its key block is read
        {
            uint64_t confb = 0;
            for (int m0 = 0; m0 < C - 1; m0 += 4) {  // (lanes past the events hold a far-away cell: reading them changes nothing)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = m0 + q;
                    const uint32_t rcm = (uint32_t)__builtin_amdgcn_readlane((int)rc_i, m);
                    const uint32_t sad = __builtin_amdgcn_sad_u8(rc_i, rcm, 0u);
                    const uint64_t near = __ballot(sad <= 2u);
                    confb |= near & (~0ull << (m + 1));
                }
            }
            const uint64_t cb = confb & ((C < 64) ? ((1ull << C) - 1ull) : ~0ull);
            if (cb) {
                const int c0 = __ffsll((unsigned long long)cb) - 1;
                C = (c0 < C) ? c0 : C;
            }
            ev = lane < C;
        }
        // ---------------- popped blocks without their popped coordinate: 8 events per pass, one per 8-lane group (2 keys per lane = one line
        // per block).  The results stay in registers (every lane of a group holds its group's) and are fetched by the event lanes with ds_bpermute.
        double rest = W_INF;     // lane e: minimum of event e's block without coordinate i_e ...
        uint32_t rarg = 0;       // ... and its coordinate
        {
            const int npass = (C + 7) >> 3;
            for (int p0 = 0; p0 < npass; p0 += 4) {
                double2 k2[4];
                uint32_t be4[4], ie4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = 8 * (p0 + q) + g;
                    const bool eg = e < C;
                    be4[q] = eg ? (uint32_t)SLB[eg ? e : 0] : 0u;
                    const uint32_t ib = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((uint32_t)(eg ? e : 0) << 2), (int)i);
                    ie4[q] = eg ? ib : 0xffffffffu;
                    k2[q] = make_double2(W_INF, W_INF);
                    if (eg) k2[q] = reinterpret_cast<const double2*>(keys + (size_t)be4[q] * 16)[gl];
                }
                double gmq[4];
                uint32_t gcq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    gmq[q] = W_INF;
                    gcq[q] = 0;
                    if (8 * (p0 + q) >= C) continue;  // (uniform)
                    double kq0 = k2[q].x, kq1 = k2[q].y;
                    const uint32_t c0 = be4[q] * 16u + (uint32_t)gl * 2u;
                    if (c0 + 0u == ie4[q]) kq0 = W_INF;
                    if (c0 + 1u == ie4[q]) kq1 = W_INF;
                    const bool hi = kq1 < kq0;
                    const double lm = hi ? kq1 : kq0;
                    const uint32_t lc = c0 + (hi ? 1u : 0u);
                    const double gm = w_grp8_min(lm);
                    const uint64_t winball = __ballot(lm == gm);
                    const int wl = __ffs((unsigned)((winball >> (8 * g)) & 0xffu)) - 1;  // (>= 0: some lane of the group holds the minimum)
                    gmq[q] = gm;
                    gcq[q] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((uint32_t)(lane & ~7) + (uint32_t)wl) << 2), (int)lc);
                }
                // event lane e = 8 (p0 + q) + g' takes pass q's value from group g' = e & 7 (any lane of it: its first)
                const uint32_t srcl = (uint32_t)(lane & 7) * 8u;
                const int myq = (lane >> 3) - p0;  // which pass of this batch holds lane's event (0..3), if any
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (8 * (p0 + q) >= C) continue;  // (uniform)
                    const double v = w_shfl(gmq[q], srcl);
                    const uint32_t c = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(srcl << 2), (int)gcq[q]);
                    if (myq == q) {
                        rest = v;
                        rarg = c;
                    }
                }
            }
        }
        if (PROF) ph_zone += (uint64_t)C;
        W_ORDER();
        WPHASE(1);
        // ---------------- rates from the tracked sums (src/sfact.jl:116-119 with g_i(t′) = g_i + gd_i (t′ − tg_i))
        const double g_now = g_i + gd_i * (tp - tg_i);
        const double l = w_pos(g_now * th);
        const double lbound = w_pos(a_i + b_i * (tp - told_i));
        // ---------------- accept chain: offsets and outcomes as a fix-point (every round settles the events up to the next change)
        uint32_t cost = ev ? 2u : 0u;
        uint32_t off = 0;
        bool acc = false;
        for (int round = 0; round < 66; ++round) {
            const uint32_t incl = w_scan_add_u32(cost);
            off = incl - cost;
            const bool inwin = ev && (off + 1u + k_i <= W_WIN);
            const double u = draw(dnm + ((off < 127u) ? off : 127u));
            acc = inwin && (u * lbound < l);  // :121
            const uint32_t nc = ev ? (acc ? (1u + k_i) : 2u) : 0u;
            const bool changed = nc != cost;
            cost = nc;
            if (__ballot(changed) == 0) break;
        }
        // events whose draws would leave the ring wait for the next iteration
        {
            const uint64_t outb = __ballot(ev && !(off + 1u + k_i <= W_WIN));
            if (outb) {
                const int cut = __ffsll((unsigned long long)outb) - 1;
                C = (cut < C) ? cut : C;
            }
        }
        // at most W_AMAX accepted events per iteration: the candidate list ends before the next one
        {
            uint64_t ab = __ballot(acc) & ((C < 64) ? ((1ull << C) - 1ull) : ~0ull);
            if (__popcll(ab) > W_AMAX) {
                uint64_t m_ = ab;
                for (int q = 0; q < W_AMAX; ++q) m_ &= m_ - 1;
                C = __ffsll((unsigned long long)m_) - 1;
            }
        }
        // a proposal that violates its bound ends the run (adapt = false: error(...), :124): nothing after it is looked at
        const bool violated0 = acc && (l >= lbound);
        int vsel = -1;
        {
            const uint64_t vb = __ballot(violated0) & ((C < 64) ? ((1ull << C) - 1ull) : ~0ull);
            if (vb) {
                vsel = __ffsll((unsigned long long)vb) - 1;
                C = vsel;  // the violating event itself is not committed
            }
        }
        ev = lane < C;
        acc = acc && ev;
        if (PROF) ph_eval += (uint64_t)C;
        const uint64_t accball = __ballot(acc);
        const int nacc_it = __popcll(accball);
        if (acc) ACL[__popcll(accball & ((1ull << lane) - 1ull))] = (uint16_t)lane;
        W_ORDER();
        WPHASE(2);
        // ---------------- accepted events, one 8-lane group each: members of G1[i] (ascending, :131-135)
        // (the groups of the accepted events are the LAST nacc_it groups of the wave, in event order: the low lanes -- lane r = event r -- are
        // then free to re-bound their rejected proposals in the same evaluation, see below)
        const int g0 = 8 - nacc_it;
        const bool gact = g >= g0;
        const uint32_t ea = gact ? (uint32_t)ACL[g - g0] : 0u;
        const uint32_t ia_b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ea << 2), (int)i);
        const uint32_t off_b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ea << 2), (int)off);
        const uint32_t ia = gact ? ia_b : 0u;
        const uint32_t blka = gact ? (uint32_t)SLB[ea] : 0u;
        const double tpa = gact ? bk[blka] : 0.0;
        const uint32_t offa = gact ? off_b : 0u;
        // G1[ia] on the lattice, ascending: {ia − n, ia − 1, ia, ia + 1, ia + n} inside the grid -- computed, so that the members' records are
        // requested at once; the CSC tables are read for the VALUES only (Γ[j, i] = Γ[i, j]: symmetric, checked on the host)
        const uint32_t cola = __umulhi(ia, nmagic), rowa = ia - cola * nlat;
        const bool hasL = cola > 0u, hasU = rowa > 0u, hasD = rowa + 1u < nlat, hasR = cola + 1u < nlat;
        const uint32_t ka = gact ? (1u + (hasL ? 1u : 0u) + (hasU ? 1u : 0u) + (hasD ? 1u : 0u) + (hasR ? 1u : 0u)) : 0u;
        const bool mem = gact && (uint32_t)gl < ka;
        uint32_t jm = ia;
        {
            // position gl among the present members in the order L, U, self, D, R
            uint32_t pos = (uint32_t)gl;
            const uint32_t cand5[5] = {ia - nlat, ia - 1u, ia, ia + 1u, ia + nlat};
            const bool has5[5] = {hasL, hasU, true, hasD, hasR};
            uint32_t seen = 0;
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                if (has5[q]) {
                    if (seen == pos && mem) jm = cand5[q];
                    seen += 1;
                }
            }
        }
        const double gam = mem ? P.tb.cc_shared[ia].gam[gl] : 0.0;  // (from tval[cp + gl] instead: 8 % more L2 misses, no faster)
        TrRec* const rj = rec + jm;
        TrRec* const ria = rec + ia;
        // (the reflecting coordinate's own fields are read again by its group: the lines are in L2)
        const double th_ia = ria->th;
        double xa = ria->x, txa = ria->tx, Ia = ria->I;
        const uint64_t acc_ia = ria->acc;
        const double thj0 = rj->th, gj0 = rj->g, gdj0 = rj->gd, tgj = rj->tg;
        const double2 cjm2 = *reinterpret_cast<const double2*>(&P.tb.cc_shared[jm].c);
        const double2 ka01 = reinterpret_cast<const double2*>(keys + (size_t)blka * 16)[gl];  // the popped block of the accepted event (patched below)
        // ---------------- ONE evaluation of the new bound and key per lane (logarithm, two divisions, square root): the re-bound of a rejected
        // proposal (:137-140) in its event lane, the re-bound of a member of G1 (:131-135) in its group lane.  A lane that is both (more than
        // 64 − 8 nacc candidates) evaluates its rejected proposal again below.
        const bool selfl = mem && jm == ia;
        const double thj = selfl ? -th_ia : thj0;
        const double gj = gj0 + gdj0 * (tpa - tgj);
        const double gdj = gdj0 + gam * (-2.0 * th_ia);  // θ_i -> −θ_i
        double a2, b2, key2;
        {
            const uint32_t dix = gact ? (offa + 1u + (uint32_t)gl) : (off + 1u);
            const double L = pdmp_log(draw(dnm + ((dix < 127u) ? dix : 127u)));
            const double cc = gact ? cjm2.x : c_i, cc100 = gact ? cjm2.y : c_i2.y;
            const double gg = gact ? gj : g_now, tt = gact ? thj : th, gdd = gact ? gdj : gd_i;
            a2 = cc + gg * tt;
            b2 = cc100 + tt * gdd;
            key2 = (gact ? tpa : tp) + w_poisson_time_L(a2, b2, L);
        }
        const double aj = a2, bj = b2;
        const double keyj = mem ? key2 : W_INF;
        if (__ballot(ev && !acc && gact) != 0) {
            const double L = pdmp_log(draw(dnm + ((off + 1u < 127u) ? off + 1u : 127u)));
            const double a2e = c_i + g_now * th;
            const double b2e = c_i2.y + th * gd_i;
            const double k2e = tp + w_poisson_time_L(a2e, b2e, L);
            if (gact) {
                a2 = a2e;
                b2 = b2e;
                key2 = k2e;
            }
        }
        // new minimum of the popped block of a rejected event, and what the event exposes
        double rowmin = W_INF;
        uint32_t cand = i;
        if (ev && !acc) {
            const bool mine = key2 < rest || (key2 == rest && i < rarg);
            rowmin = mine ? key2 : rest;
            cand = mine ? i : rarg;
        }
        if (selfl) {  // event(i, t, x, θ, F) (src/sfact.jl:50-52): x_i at t′
            const double dtx = tpa - txa;
            const double xn = xa + th_ia * dtx;
            Ia = Ia + dtx * ((xa + xn) * 0.5);
            xa = xn;
            txa = tpa;
        }
        // the popped block of the accepted event with the members' new keys patched in
        double rowmin_a = W_INF;
        uint32_t cand_a = 0;
        int wl_a = -1;
        {
            double kq0 = ka01.x, kq1 = ka01.y;
#pragma unroll
            for (int m = 0; m < 5; ++m) {
                const uint32_t src = (uint32_t)(lane & ~7) + (uint32_t)m;
                const uint32_t jq = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)jm);
                const double kv = w_shfl(keyj, src);
                const bool inblk = ((uint32_t)m < ka) && ((jq >> 4) == blka) && (((jq & 15u) >> 1) == (uint32_t)gl);
                if (inblk) {
                    kq0 = (jq & 1u) ? kq0 : kv;
                    kq1 = (jq & 1u) ? kv : kq1;
                }
            }
            const bool hi = kq1 < kq0;
            const double lm = hi ? kq1 : kq0;
            rowmin_a = w_grp8_min(lm);
            const uint64_t winball = __ballot(gact && lm == rowmin_a);
            wl_a = __ffs((unsigned)((winball >> (8 * g)) & 0xffu)) - 1;
            cand_a = (uint32_t)gl * 2u + (hi ? 1u : 0u);
            const double keymin = w_grp8_min(keyj);
            if (gact && gl == 0) EX[ea] = w_min(rowmin_a, keymin);
        }
        if (ev && !acc) EX[lane] = rowmin;  // (rowmin <= key2: the new key is one of its candidates)
        W_ORDER();
        WPHASE(3);
        // ---------------- validate: all earlier events commit, zones disjoint, nothing produced or exposed earlier than t′
        uint32_t Rc;
        {
            const double expo = ev ? EX[lane] : W_INF;
            const double prev = w_shfl(expo, (uint32_t)((lane > 0) ? lane - 1 : 0));
            const double pref = w_scan_min_f64((lane > 0) ? prev : W_INF);  // exclusive prefix minimum
            const bool okr = ev && (lane == 0 || pref > tp);  // (zone conflicts ended the candidate list already)
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
                if (!stop_before && !(w_readlane(tp, r) < T)) {
                    running = false;
                    stopped = true;
                }
                if (stopped) Rc = (uint32_t)r + 1u;
            }
            if (stopped) vsel = -1;
        }
        // steer the threshold so that the raw candidate list is just longer than what can commit
        if (lane == 0) SELDT[0] = dt_used * (((int)Rc >= Craw) ? W_GROW : (((int)Rc + (int)W_SLACK < Craw) ? W_SHRINK : 1.0));
        WPHASE(4);
        // ---------------- commit the valid prefix
        const bool commit = ev && (uint32_t)lane < Rc;
        if (commit && !acc) {
            TrRec* const rw = rec + i;
            rw->a = a2;
            rw->b = b2;
            rw->t_old = tp;
            rw->tprop = tp;
            keys[i] = key2;
            bk[blk] = rowmin;
            bi[blk] = (uint8_t)(cand & 15u);
        }
        const bool gcommit = gact && ea < Rc;
        const uint64_t acc_c = accball & ((Rc < 64u) ? ((1ull << Rc) - 1ull) : ~0ull);
        if (gcommit) {
            if (mem) {
                rj->g = gj;
                rj->gd = gdj;
                rj->tg = tpa;
                rj->a = aj;
                rj->b = bj;
                rj->t_old = tpa;
                keys[jm] = keyj;
            }
            if (selfl) {
                ria->x = xa;
                ria->th = -th_ia;
                ria->tx = txa;
                ria->I = Ia;
                ria->acc = acc_ia + 1;
                ria->tprop = tpa;
                ria->tacc = tpa;
                if (evout) {
                    const uint32_t rnk = (uint32_t)__popcll(acc_c & ((1ull << ea) - 1ull));
                    pdmp_event e;
                    e.t = tpa;
                    e.i = (int64_t)ia;
                    e.x = xa;
                    e.theta = -th_ia;
                    evout[ntrace0 + dnacc + rnk] = e;
                }
            }
            if (gl == wl_a) {
                bk[blka] = rowmin_a;
                bi[blka] = (uint8_t)cand_a;
            }
        }
        W_ORDER();
        WPHASE(5);
        // ---------------- first-level entries of re-bounded neighbours living in other blocks (as in the 8-event kernels)
        const bool upd = gcommit && mem && (jm >> 4) != blka;
        if (__ballot(upd) != 0) {
            W_ORDER();
            const uint32_t bjv = upd ? (jm >> 4) : 0u;
            const double curv = bk[bjv];
            const uint32_t civ = bjv * 16u + (uint32_t)bi[bjv];
            const bool lower = upd && (keyj < curv || (keyj == curv && jm < civ));
            const bool resc = upd && !lower && civ == jm;
            if (lower) CL[bjv & 63u] = (uint8_t)lane;
            W_ORDER();
            const bool lost = lower && CL[bjv & 63u] != (uint8_t)lane;
            if (__ballot(lost || resc) == 0) {
                if (lower) {
                    bk[bjv] = keyj;
                    bi[bjv] = (uint8_t)(jm & 15u);
                }
            } else {
                uint64_t todo = __ballot(upd);
                while (todo) {  // one by one, in lane order (= event order, members ascending)
                    const int src = __ffsll((unsigned long long)todo) - 1;
                    todo &= todo - 1;
                    const uint32_t j = (uint32_t)__builtin_amdgcn_readlane((int)jm, src);
                    const double kj = w_readlane(keyj, src);
                    const uint32_t bj_ = j >> 4;
                    W_ORDER();
                    const double cur = bk[bj_];
                    const uint32_t ci = bj_ * 16u + (uint32_t)bi[bj_];
                    if (kj < cur || (kj == cur && j < ci)) {
                        if (lane == 0) {
                            bk[bj_] = kj;
                            bi[bj_] = (uint8_t)(j & 15u);
                        }
                    } else if (ci == j) {
                        const double kv = __hip_atomic_load(keys + (size_t)bj_ * 16 + (lane & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const double mn = w_wave_min(kv);
                        const uint64_t bl = __ballot(kv == mn);
                        const int arg = bl ? (__ffsll((unsigned long long)bl) - 1) : 0;
                        if (lane == 0) {
                            bk[bj_] = mn;
                            bi[bj_] = (uint8_t)(arg & 15);
                        }
                    }
                }
            }
        }
        WPHASE(6);
        // ---------------- counters; the violating proposal itself (counted, acc bumped, then error(...), :120-124)
        if (Rc > 0u) {
            const uint32_t costL = (uint32_t)__builtin_amdgcn_readlane((int)cost, (int)(Rc - 1u));
            const uint32_t offL = (uint32_t)__builtin_amdgcn_readlane((int)off, (int)(Rc - 1u));
            dnum += Rc;
            dnacc += (uint32_t)__popcll(acc_c);
            dnm += offL + costL;
            t_last = w_readlane(tp, (int)(Rc - 1u));
            if (acc_c) t_event = w_readlane(tp, 63 - __builtin_clzll(acc_c));
        }
        if (vsel >= 0) {  // (vsel == Rc: every earlier event is committed)
            const double tpv = w_readlane(tp, vsel);
            const uint32_t iv = (uint32_t)__builtin_amdgcn_readlane((int)i, vsel);
            if (lane == 0) rec[iv].tprop = tpv;
            dnum += 1;
            vnacc = 1;
            dnm += 1;  // its coin
            t_last = tpv;
            status = PDMP_CHAIN_BOUND_VIOLATED;
        }
        if (status != PDMP_CHAIN_OK) break;
        W_ORDER();
    }

    if (PROF && P.dbg && chain == 0 && lane == 0) {
        for (int q = 0; q < 10; ++q) P.dbg[q] = (double)ph[q];
        P.dbg[10] = (double)ph_iters;
        P.dbg[11] = (double)ph_raw;
        P.dbg[12] = (double)ph_zone;
        P.dbg[13] = (double)ph_eval;
    }
#undef WPHASE
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

bool zz_trackx_supported(const ZzRunParams& p) {
    return p.lattice_n >= 16 && p.lattice_n <= 128 && !p.adapt && p.c_chain == nullptr && p.tb.gmu_t == nullptr && !p.track_two_sums &&
           !p.has_refresh && p.d >= 2048 && p.d <= (int64_t)W_NBLK * 16;
}

int launch_zz_local_trackx(const ZzRunParams& p, int64_t nchains, void* stream) {
    dim3 grid((unsigned)nchains), block(64);
    ZzRunParams q = p;
    q.nblk = (uint32_t)((p.d + 15) / 16);  // (dk is a multiple of 64, the padding keys are +Inf)
    if (p.dbg) hipLaunchKernelGGL((zz_local_trackx_kernel<true>), grid, block, W_BYTES, (hipStream_t)stream, q);
    else hipLaunchKernelGGL((zz_local_trackx_kernel<false>), grid, block, W_BYTES, (hipStream_t)stream, q);
    return (int)hipGetLastError();
}

}  // namespace pdmp
