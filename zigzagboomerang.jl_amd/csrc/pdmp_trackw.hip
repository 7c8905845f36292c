// pdmp_trackw.hip -- the tracked-gradient local ZigZag, ONE PROPOSAL PER LANE (zz_local_trackw_kernel).
//
// With tracked gradients (pdmp_kernels.hip: zz_local_track_kernel) a proposal is a scalar piece of work -- one record, one thinning test, one
// re-bound -- yet the 8-event kernel still spends an 8-lane group and ~150 wavefront instructions on it (selection, templates, validation are
// paid per iteration of eight).  Here an iteration takes up to 56 candidate events, the smallest block minima of the queue in time order, and
// gives each to a lane:
//   select    threshold + compaction as in the 8-event kernel, then every candidate ranks itself against all others (64 compares per lane)
//   evaluate  lane r: own record, rates, the re-bound it would make if rejected
//   accept    the draw offset of event r is the number of draws the earlier events consume (2 per reject, 1 + k per accept): a fix-point of
//             {prefix sum over the lanes, thinning test at that offset}; each round fixes every event up to the next newly accepted one
//   accepted  (18 %: at most 8 per iteration, the rest waits) one 8-lane group each: the members of G1[i] are brought to t′, take Γ[i,j] δθ_i into
//             their velocity sums and are re-bounded; the group also rescans the popped key block with the members' new keys patched in
//   blocks    rejected events: the minimum of the popped block WITHOUT its popped coordinate is scanned by 8-lane groups (8 events per pass) from the
//             block as it is in HBM; new block minimum = the smaller of that and the new key
//   validate  event r commits iff all earlier ones do, its zone G1[i_r] meets none of theirs (Manhattan distance of the lattice coordinates <= 2:
//             one v_sad_u8 per pair) and nothing they produce or expose precedes it (prefix minimum over the lanes)
//   commit    rejected events by their lanes (one sector + one key), accepted ones by their groups; first-level updates as in the 8-event kernels
// The committed sequence is the one of zz_local_track_kernel (same draws, same tests, same arithmetic per event): index-exact against the oracle,
// floats to ~1e-13.  Requirements beyond those of gradient tracking: the n x n lattice in column-major numbering (ids i = row + n col, G1[i] the
// 5-point stencil), the plain configuration (no adapt, no means, bounding Γ == target Γ), d <= 16384.
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

// (second − first) key of a block as a 16-bit hint: units of 2^-20, rounded down, 65535 = "at least 0.0625 away (or unknown large)"
__device__ __forceinline__ uint32_t w_qdelta(double delta) {
    return (delta < 0.0624) ? (uint32_t)(delta * 1048576.0) : 65535u;  // (NaN from Inf − Inf compares false: 65535)
}
__device__ __forceinline__ double w_udelta(uint32_t q) {
    return (q >= 65535u) ? W_INF : (double)q * (1.0 / 1048576.0);
}

}  // namespace

// LDS layout (bytes)
constexpr uint32_t W_BK = 0;         // [512] f64 block minima (first level of the queue, key blocks of 32)
constexpr uint32_t W_BI = 4096;      // [512] u16 their coordinates
constexpr uint32_t W_U = 5120;       // [256] f64 ring of uniforms: slot n & 255 holds draw nm0 + n for dnm <= n < wend
constexpr uint32_t W_SLT = 7168;     // [64] f64 event times, rank order
constexpr uint32_t W_TK = 7680;      // [64] f64 candidate keys, compaction order; later EX[e]: what event e exposes
constexpr uint32_t W_RM = 8192;      // [64] f64 minimum of the popped block without its popped coordinate / patched minimum (accepted events)
constexpr uint32_t W_SLB = 8704;     // [64] u16 event blocks, rank order
constexpr uint32_t W_TB = 8832;      // [64] u16 candidate blocks, compaction order; later RC[e]: coordinate of RM[e]
constexpr uint32_t W_D2R = 8960;     // [64] u16 (second − first) minimum of a popped block's other keys, quantised (hint)
constexpr uint32_t W_ACL = 9088;     // [8] u16 the accepted events
constexpr uint32_t W_RO = 9104;      // [64] u8 owner of each rank (duplicate detection); later the claims of the parallel first-level update
constexpr uint32_t W_CL = W_RO;
constexpr uint32_t W_SELDT = 9168;   // f64 selection threshold above the minimum
constexpr uint32_t W_D2 = 9176;      // [512] u16 per block: (second smallest key − smallest key) in units of 2^-20, rounded DOWN, saturating -- a HINT
                                     // that lets the selection stop where a popped block's second key would end the committable prefix anyway
constexpr uint32_t W_BYTES = 10200;
constexpr uint32_t W_NBLK = 512;
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
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void zz_local_trackw_kernel(ZzRunParams P) {
    const int lane = threadIdx.x;
    const int g = lane >> 3, gl = lane & 7;
    const int64_t chain = blockIdx.x;
    const int64_t d = P.d;
    const uint32_t nblk = P.nblk;
    const uint32_t nlat = (uint32_t)P.lattice_n, nmagic = P.lattice_magic;

    extern __shared__ __align__(16) unsigned char smem[];
    double* const bk = reinterpret_cast<double*>(smem + W_BK);
    uint16_t* const bi = reinterpret_cast<uint16_t*>(smem + W_BI);
    double* const U = reinterpret_cast<double*>(smem + W_U);
    double* const SLT = reinterpret_cast<double*>(smem + W_SLT);
    double* const TK = reinterpret_cast<double*>(smem + W_TK);
    double* const EX = TK;
    double* const RM = reinterpret_cast<double*>(smem + W_RM);
    uint16_t* const SLB = reinterpret_cast<uint16_t*>(smem + W_SLB);
    uint16_t* const TB = reinterpret_cast<uint16_t*>(smem + W_TB);
    uint16_t* const RC = TB;
    uint16_t* const D2 = reinterpret_cast<uint16_t*>(smem + W_D2);
    uint16_t* const D2R = reinterpret_cast<uint16_t*>(smem + W_D2R);
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
    uint32_t wend = 0;  // draws nm0 + [dnm, wend) are in the ring
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
        const double* kp = keys + (size_t)b * 32;
        double mk = kp[0], m2 = W_INF;
        uint32_t mi = 0;
#pragma unroll 8
        for (int q = 1; q < 32; ++q) {
            const double v = kp[q];
            if (v < mk) {
                m2 = mk;
                mk = v;
                mi = q;
            } else if (v < m2) {
                m2 = v;
            }
        }
        bk[b] = mk;
        bi[b] = (uint16_t)(b * 32 + mi);
        D2[b] = (uint16_t)w_qdelta(m2 - mk);
    }
    for (uint32_t b = nblk + lane; b < W_NBLK; b += 64) {
        bk[b] = W_INF;
        bi[b] = 0;
        D2[b] = 65535;
    }
    W_ORDER();

    uint64_t ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t ph_t0 = PROF ? (uint64_t)__builtin_readcyclecounter() : 0;
    uint64_t ph_iters = 0;
#define WPHASE(k)                                                         \
    do {                                                                  \
        if (PROF) {                                                       \
            const uint64_t now_ = (uint64_t)__builtin_readcyclecounter(); \
            ph[k] += now_ - ph_t0;                                        \
            ph_t0 = now_;                                                 \
        }                                                                 \
    } while (0)

    bool running = stop_before || (t_event < T);
    while (running) {
        if (dnacc >= trace_room) {
            status = PDMP_CHAIN_TRACE_FULL;
            break;
        }
        // ---------------- ring of uniforms: make draws dnm .. dnm + 255 available
        while (wend < dnm + 256u) {
            const uint32_t n = wend + (uint32_t)lane;
            if (n < dnm + 256u) U[n & 255u] = pdmp_u01(seed, PDMP_STREAM_MAIN, nm0 + (uint64_t)n);
            wend += 64u;
        }
        wend = (wend < dnm + 256u) ? wend : (dnm + 256u);
        // ---------------- select: every first-level entry <= m + sel_dt, at most W_CMAX of them
        int C = 0;
        bool first_inf = false;
        double dt_used = 0.0;
        {
            double kk[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) kk[j] = bk[lane + 64 * j];
            const double mloc = w_min(w_min(w_min(kk[0], kk[1]), w_min(kk[2], kk[3])), w_min(w_min(kk[4], kk[5]), w_min(kk[6], kk[7])));
            const double mq = w_wave_min(mloc);
            if (!(mq < W_INF)) {
                first_inf = true;
            } else if (!(stop_before && !(mq < T))) {
                TK[lane] = W_INF;
                double dt_sel = w_uniform(SELDT[0]);
                auto below = [](uint64_t m_) -> uint32_t {
                    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_, 0u));
                };
                uint32_t Cc;
                for (int tries = 0;; ++tries) {
                    double tau = mq + dt_sel;
                    if (stop_before && !(tau < T)) tau = w_below(T);
                    const bool pile = tries > 64;
                    if (tries >= 64) tau = mq;
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
                                TB[ix] = (uint16_t)((uint32_t)lane + 64u * j);
                            }
                        }
                        base += (uint32_t)__popcll(Mj);
                    }
                    Cc = base;
                    if (Cc <= (uint32_t)W_CMAX) break;
                    dt_sel *= 0.5;
                    W_ORDER();
                    TK[lane] = W_INF;
                }
                W_ORDER();
                // rank of candidate `lane` among all: the number of strictly smaller keys (entries past the count hold +Inf)
                const double own = TK[lane];
                uint32_t rank = 0;
                {
                    const double2* T2 = reinterpret_cast<const double2*>(TK);
                    const int mh = (int)((Cc + 1u) >> 1);  // (entries past the count hold +Inf: never smaller)
#pragma unroll 4
                    for (int m = 0; m < mh; ++m) {
                        const double2 o = T2[m];
                        rank += (o.x < own) ? 1u : 0u;
                        rank += (o.y < own) ? 1u : 0u;
                    }
                }
                const bool isc = (uint32_t)lane < Cc;
                if (isc) RO[rank] = (uint8_t)lane;
                W_ORDER();
                const bool dup = isc && RO[rank] != (uint8_t)lane;
                if (__ballot(dup) != 0) {
                    // exactly equal keys among the candidates (probability zero unless keys are tied by construction): one event this
                    // iteration, the tied minimum of the lowest block
                    const uint64_t mm = __ballot(isc && own == mq);
                    uint32_t bsel = isc && own == mq ? (uint32_t)TB[lane] : 0xffffffffu;
                    for (int off = 32; off >= 1; off >>= 1) {
                        const uint32_t o = (uint32_t)__shfl_xor((int)bsel, off, 64);
                        bsel = (o < bsel) ? o : bsel;
                    }
                    (void)mm;
                    W_ORDER();
                    if (lane == 0) {
                        SLT[0] = mq;
                        SLB[0] = (uint16_t)bsel;
                    }
                    Cc = 1;
                } else if (isc) {
                    SLT[rank] = own;
                    SLB[rank] = TB[lane];
                }
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
        const int Craw = C;

        // ---------------- lane r = event r: own record, neighbourhood size, c_i
        bool ev = lane < C;
        const double tp = ev ? SLT[lane] : W_INF;
        const uint32_t blk = ev ? (uint32_t)SLB[lane] : 0u;
        const uint32_t i = ev ? (uint32_t)bi[blk] : 0u;
        const TrRec* const ri = rec + i;
        const double th = ri->th;
        const double g_i = ri->g, gd_i = ri->gd, tg_i = ri->tg;
        const double told_i = ri->t_old, a_i = ri->a, b_i = ri->b;
        const double c_i = P.tb.c_shared[i];
        const uint32_t cp_i = P.tb.colptr[i];
        const uint32_t k_i = P.tb.colptr[i + 1] - cp_i;
        // lattice coordinates packed for the zone test: byte 0 = row, byte 1 = column
        const uint32_t col_i = __umulhi(i, nmagic);
        const uint32_t rc_i = (i - col_i * nlat) | (col_i << 8);
        // ---------------- (the own-record loads are in flight) where would the committable prefix end anyway?  (a) zones: event r cannot
        // commit with an earlier event whose G1 meets its own (Manhattan distance <= 2); (b) the second key of an earlier event's block, which
        // becomes that block's minimum once its first is popped -- known approximately from the D2 hints (a lower bound: cuts early at worst).
        // Candidates from the first such position on are dropped BEFORE their key blocks are read; the exact tests follow in `validate`.
        {
            uint64_t confb = 0;
            for (int m = 0; m < C - 1; ++m) {
                const uint32_t rcm = (uint32_t)__builtin_amdgcn_readlane((int)rc_i, m);
                const uint32_t sad = __builtin_amdgcn_sad_u8(rc_i, rcm, 0u);
                const uint64_t near = __ballot(sad <= 2u);
                confb |= near & (~0ull << (m + 1));
            }
            const double lb2 = ev ? (tp + w_udelta((uint32_t)D2[blk])) : W_INF;
            const double prevl = w_shfl(lb2, (uint32_t)((lane > 0) ? lane - 1 : 0));
            const double pmin = w_scan_min_f64((lane > 0) ? prevl : W_INF);
            const bool cut = ev && lane > 0 && (((confb >> lane) & 1ull) || !(pmin > tp));
            const uint64_t cb = __ballot(cut);
            if (cb) {
                const int c0 = __ffsll((unsigned long long)cb) - 1;
                C = (c0 < C) ? c0 : C;
            }
            ev = lane < C;
            // steer the threshold so that the raw candidate list is just longer than what can commit
            if (lane == 0) SELDT[0] = dt_used * ((C >= Craw) ? W_GROW : ((C + (int)W_SLACK < Craw) ? W_SHRINK : 1.0));
        }
        // ---------------- popped blocks without their popped coordinate: 8 events per pass, one per 8-lane group (4 keys per lane)
        W_ORDER();
        {
            // all loads of a batch of four passes first (one HBM round trip per batch), then the reductions
            const int npass = (C + 7) >> 3;
            for (int p0 = 0; p0 < npass; p0 += 4) {
                double2 k01[4], k23[4];
                uint32_t be4[4], ie4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = 8 * (p0 + q) + g;
                    const bool eg = e < C;
                    be4[q] = eg ? (uint32_t)SLB[eg ? e : 0] : 0u;
                    const uint32_t ib = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((uint32_t)(eg ? e : 0) << 2), (int)i);
                    ie4[q] = eg ? ib : 0xffffffffu;
                    const double2* kp = reinterpret_cast<const double2*>(keys + (size_t)be4[q] * 32 + gl * 4);
                    k01[q] = kp[0];
                    k23[q] = kp[1];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = 8 * (p0 + q) + g;
                    const bool eg = e < C;
                    double kq0 = k01[q].x, kq1 = k01[q].y, kq2 = k23[q].x, kq3 = k23[q].y;
                    const uint32_t c0 = be4[q] * 32u + (uint32_t)gl * 4u;
                    if (c0 + 0u == ie4[q]) kq0 = W_INF;
                    if (c0 + 1u == ie4[q]) kq1 = W_INF;
                    if (c0 + 2u == ie4[q]) kq2 = W_INF;
                    if (c0 + 3u == ie4[q]) kq3 = W_INF;
                    double lm = kq0, lm2 = W_INF;
                    uint32_t li = 0;
                    if (kq1 < lm) {
                        lm2 = lm;
                        lm = kq1;
                        li = 1;
                    } else {
                        lm2 = kq1;
                    }
                    if (kq2 < lm) {
                        lm2 = lm;
                        lm = kq2;
                        li = 2;
                    } else if (kq2 < lm2) {
                        lm2 = kq2;
                    }
                    if (kq3 < lm) {
                        lm2 = lm;
                        lm = kq3;
                        li = 3;
                    } else if (kq3 < lm2) {
                        lm2 = kq3;
                    }
                    const double gm = w_grp8_min(lm);
                    const uint64_t winball = __ballot(eg && lm == gm);
                    const int wl = __ffs((unsigned)((winball >> (8 * g)) & 0xffu)) - 1;
                    const double gm2 = w_grp8_min((gl == wl) ? lm2 : lm);  // the second smallest of the 31 other keys
                    if (eg && gl == wl) {
                        RM[e] = gm;
                        RC[e] = (uint16_t)(c0 + li);
                        D2R[e] = (uint16_t)w_qdelta(gm2 - gm);
                    }
                }
            }
        }
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
            const bool inwin = ev && (off + 1u + k_i <= 256u);
            const double u = U[(dnm + off) & 255u];
            acc = inwin && (u * lbound < l);  // :121
            const uint32_t nc = ev ? (acc ? (1u + k_i) : 2u) : 0u;
            const bool changed = nc != cost;
            cost = nc;
            if (__ballot(changed) == 0) break;
        }
        // events whose draws would leave the ring wait for the next iteration
        {
            const uint64_t outb = __ballot(ev && !(off + 1u + k_i <= 256u));
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
        const uint64_t accball = __ballot(acc);
        const int nacc_it = __popcll(accball);
        if (acc) ACL[__popcll(accball & ((1ull << lane) - 1ull))] = (uint16_t)lane;
        W_ORDER();
        WPHASE(2);
        // ---------------- accepted events, one 8-lane group each: members of G1[i] (ascending, :131-135)
        const bool gact = g < nacc_it;
        const uint32_t ea = gact ? (uint32_t)ACL[g] : 0u;
        const uint32_t ia_b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ea << 2), (int)i);
        const uint32_t off_b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ea << 2), (int)off);
        const uint32_t cp_b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ea << 2), (int)cp_i);
        const uint32_t ia = gact ? ia_b : 0u;
        const double tpa = gact ? SLT[ea] : 0.0;
        const uint32_t offa = gact ? off_b : 0u;
        const uint32_t blka = gact ? (uint32_t)SLB[ea] : 0u;
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
        const uint32_t cpa = gact ? cp_b : 0u;
        const double gam = mem ? P.tb.tval[cpa + (uint32_t)gl] : 0.0;
        TrRec* const rj = rec + jm;
        TrRec* const ria = rec + ia;
        // (the reflecting coordinate's own fields are read again by its group: the lines are in L2)
        const double th_ia = ria->th;
        double xa = ria->x, txa = ria->tx, Ia = ria->I;
        const uint64_t acc_ia = ria->acc;
        const double thj0 = rj->th, gj0 = rj->g, gdj0 = rj->gd, tgj = rj->tg;
        const double cjm = P.tb.c_shared[jm];
        const double2* const kpa = reinterpret_cast<const double2*>(keys + (size_t)blka * 32 + gl * 4);
        const double2 ka01 = kpa[0], ka23 = kpa[1];  // the popped block of the accepted event (patched below)
        // (the loads of the accepted events' groups are in flight: the rejected proposals' re-bounds -- logarithm, divisions, square root, no memory -- run under them)
        // ---------------- the re-bound of a rejected proposal (:137-140), by its own lane
        double a2, b2, key2;
        {
            const double L = pdmp_log(U[(dnm + off + 1u) & 255u]);
            a2 = c_i + g_now * th;
            b2 = c_i / 100 + th * gd_i;
            key2 = tp + w_poisson_time_L(a2, b2, L);
        }
        // new minimum of the popped block of a rejected event, and what the event exposes
        double rowmin = W_INF;
        uint32_t cand = i, d2new = 65535u;
        if (ev && !acc) {
            const double rest = RM[lane];
            const uint32_t rarg = RC[lane];
            const bool mine = key2 < rest || (key2 == rest && i < rarg);
            rowmin = mine ? key2 : rest;
            cand = mine ? i : rarg;
            // hint for the next pop of this block: its second key is `rest` if the new key leads, else the smaller of the new key and the
            // (lower bound of the) second of the rest
            const double dr = w_udelta((uint32_t)D2R[lane]);
            const double dk = key2 - rest;
            d2new = mine ? w_qdelta(rest - key2) : w_qdelta((dk < dr) ? dk : dr);
        }
        double keyj = W_INF, aj = 0.0, bj = 0.0, gj = 0.0, gdj = 0.0;
        const bool selfl = mem && jm == ia;
        {
            const double delta = -2.0 * th_ia;  // θ_i -> −θ_i
            const double thj = selfl ? -th_ia : thj0;
            gj = gj0 + gdj0 * (tpa - tgj);
            gdj = gdj0 + gam * delta;
            aj = cjm + gj * thj;
            bj = cjm / 100 + thj * gdj;
            const double L = pdmp_log(U[(dnm + offa + 1u + (uint32_t)gl) & 255u]);
            if (mem) keyj = tpa + w_poisson_time_L(aj, bj, L);
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
        uint32_t cand_a = 0, d2_a = 65535u;
        int wl_a = -1;
        {
            double kq[4] = {ka01.x, ka01.y, ka23.x, ka23.y};
#pragma unroll
            for (int m = 0; m < 5; ++m) {
                const uint32_t src = (uint32_t)(lane & ~7) + (uint32_t)m;
                const uint32_t jq = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)jm);
                const double kv = w_shfl(keyj, src);
                const bool inblk = ((uint32_t)m < ka) && ((jq >> 5) == blka) && (((jq & 31u) >> 2) == (uint32_t)gl);
                if (inblk) {
                    const uint32_t w = jq & 3u;
                    kq[0] = (w == 0u) ? kv : kq[0];
                    kq[1] = (w == 1u) ? kv : kq[1];
                    kq[2] = (w == 2u) ? kv : kq[2];
                    kq[3] = (w == 3u) ? kv : kq[3];
                }
            }
            double lm = kq[0], lm2 = W_INF;
            uint32_t li = 0;
            if (kq[1] < lm) {
                lm2 = lm;
                lm = kq[1];
                li = 1;
            } else {
                lm2 = kq[1];
            }
            if (kq[2] < lm) {
                lm2 = lm;
                lm = kq[2];
                li = 2;
            } else if (kq[2] < lm2) {
                lm2 = kq[2];
            }
            if (kq[3] < lm) {
                lm2 = lm;
                lm = kq[3];
                li = 3;
            } else if (kq[3] < lm2) {
                lm2 = kq[3];
            }
            rowmin_a = w_grp8_min(lm);
            const uint64_t winball = __ballot(gact && lm == rowmin_a);
            wl_a = __ffs((unsigned)((winball >> (8 * g)) & 0xffu)) - 1;
            d2_a = w_qdelta(w_grp8_min((gl == wl_a) ? lm2 : lm) - rowmin_a);
            cand_a = blka * 32u + (uint32_t)gl * 4u + li;
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
            bi[blk] = (uint16_t)cand;
            D2[blk] = (uint16_t)d2new;
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
                bi[blka] = (uint16_t)cand_a;
                D2[blka] = (uint16_t)d2_a;
            }
        }
        W_ORDER();
        WPHASE(5);
        // ---------------- first-level entries of re-bounded neighbours living in other blocks (as in the 8-event kernels)
        const bool upd = gcommit && mem && (jm >> 5) != blka;
        if (__ballot(upd) != 0) {
            W_ORDER();
            const uint32_t bjv = upd ? (jm >> 5) : 0u;
            const double curv = bk[bjv];
            const uint32_t civ = bi[bjv];
            const bool lower = upd && (keyj < curv || (keyj == curv && jm < civ));
            const bool resc = upd && !lower && civ == jm;
            if (lower) CL[bjv & 63u] = (uint8_t)lane;
            W_ORDER();
            const bool lost = lower && CL[bjv & 63u] != (uint8_t)lane;
            if (__ballot(lost || resc) == 0) {
                if (lower) {  // the old minimum becomes the second key
                    bk[bjv] = keyj;
                    bi[bjv] = (uint16_t)jm;
                    D2[bjv] = (uint16_t)w_qdelta(curv - keyj);
                } else if (upd) {  // (a hint: a lost race between two lanes only makes it optimistic, the exact tests do not use it)
                    const uint32_t qn = w_qdelta(keyj - curv), qo = (uint32_t)D2[bjv];
                    if (qn < qo) D2[bjv] = (uint16_t)qn;
                }
            } else {
                uint64_t todo = __ballot(upd);
                while (todo) {  // one by one, in lane order (= event order, members ascending)
                    const int src = __ffsll((unsigned long long)todo) - 1;
                    todo &= todo - 1;
                    const uint32_t j = (uint32_t)__builtin_amdgcn_readlane((int)jm, src);
                    const double kj = w_readlane(keyj, src);
                    const uint32_t bj_ = j >> 5;
                    W_ORDER();
                    const double cur = bk[bj_];
                    const uint32_t ci = bi[bj_];
                    if (kj < cur || (kj == cur && j < ci)) {
                        if (lane == 0) {
                            bk[bj_] = kj;
                            bi[bj_] = (uint16_t)j;
                            D2[bj_] = (uint16_t)w_qdelta(cur - kj);
                        }
                    } else if (ci == j) {
                        const double kv = __hip_atomic_load(keys + (size_t)bj_ * 32 + (lane & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const double mn = w_wave_min(kv);
                        const uint64_t bl = __ballot(kv == mn);
                        const int arg = bl ? (__ffsll((unsigned long long)bl) - 1) : 0;
                        const double m2 = w_wave_min(((lane & 31) == (arg & 31)) ? W_INF : kv);
                        if (lane == 0) {
                            bk[bj_] = mn;
                            bi[bj_] = (uint16_t)(bj_ * 32 + (uint32_t)(arg & 31));
                            D2[bj_] = (uint16_t)w_qdelta(m2 - mn);
                        }
                    } else if (lane == 0) {
                        const uint32_t qn = w_qdelta(kj - cur), qo = (uint32_t)D2[bj_];
                        if (qn < qo) D2[bj_] = (uint16_t)qn;
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

bool zz_trackw_supported(const ZzRunParams& p) {
    return p.lattice_n >= 16 && p.lattice_n <= 128 && !p.adapt && p.c_chain == nullptr && p.tb.gmu_t == nullptr && !p.track_two_sums &&
           !p.has_refresh && p.d >= 2048 && p.d <= (int64_t)W_NBLK * 32;
}

int launch_zz_local_trackw(const ZzRunParams& p, int64_t nchains, void* stream) {
    dim3 grid((unsigned)nchains), block(64);
    ZzRunParams q = p;
    q.nblk = (uint32_t)((p.d + 31) / 32);
    if (p.dbg) hipLaunchKernelGGL((zz_local_trackw_kernel<true>), grid, block, W_BYTES, (hipStream_t)stream, q);
    else hipLaunchKernelGGL((zz_local_trackw_kernel<false>), grid, block, W_BYTES, (hipStream_t)stream, q);
    return (int)hipGetLastError();
}

}  // namespace pdmp
