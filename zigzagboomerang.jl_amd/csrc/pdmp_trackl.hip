// pdmp_trackl.hip -- zz_local_trackl_kernel: pdmp_trackp.hip's one-proposal-per-lane tracked-gradient event loop on the LINE layout (round 6).
//
// What bounds zz_local_trackp_kernel at full width is the number of 128-byte LINES a proposal touches: its block's line of
// (key, t_old) pairs and its record's line, 2 per proposal + 2 per re-bounded neighbour -- 3.2 lines read per proposal by the counters.  Here a
// proposal touches ONE line each way:
//   * level 0 of the queue IS the record: a line holds everything the proposals of a coordinate PAIR (2 b, 2 b + 1) read and write -- the two
//     (key, t_old) pairs, (θ, g, gd, tg) of either coordinate and the constants (c_i, c_i / 100): TrLine, pdmp_engine.hpp.  A rejected proposal
//     reads its line and dirties 16 bytes of it; a re-bounded neighbour costs one line instead of two; the pair mate of a reflecting coordinate
//     is one of its lattice neighbours and costs none.  What only i's own accepted events touch (x_i at its clock, ∫ x dt, the count) is a
//     32-byte cold record per coordinate;
//   * level 1 then has d / 2 = 8192 entries, and 16 chains per CU leave 10 KB of LDS per chain: an entry is NINE BITS -- the block's minimum on
//     a WHEEL of 512 quanta, floor((key − tref) s) mod 512, a byte array and a bit plane.  s is chosen so that a quantum holds about as many block
//     minima as an iteration commits; the selection takes whole quanta [F, F + m) from the front F (a SWAR range test over the lane's 128 bytes),
//     reads the candidates' lines -- which yields the exact keys -- and ranks and thins as pdmp_trackp.hip does.  A block whose minimum lies a
//     whole revolution (or more) ahead is looked at once per revolution in vain (exp(−rate x revolution): about one look in six); an entry is
//     exact or stale LOW (a re-bounded neighbour only ever lowers it; the owner of the line refreshes it), never high, so no event is missed.
// The committed sequence and every float are those of the oracle's tracked evaluation (oracle/pdmp_oracle.c: spdmp_zigzag_tracked), bit for bit:
// the arithmetic of an event is pdmp_trackp.hip's, lane for lane; only where the operands are fetched from differs.  Neither s nor the window
// enters a result (tests/test_gpu_track_parity.py and the other users of the `trackp_form` fixture run this form too).
// Reference: the loop each chain runs is src/sfact.jl:73-145 under :199-208; the queue it replaces src/priorityqueue.jl:44-117.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "../../include/pdmp_detmath.h"
#include "pdmp_engine.hpp"

namespace pdmp {

#define L_INF __builtin_inf()
#define L_ORDER()                        \
    do {                                 \
        __builtin_amdgcn_wave_barrier(); \
        asm volatile("" ::: "memory");   \
    } while (0)

namespace {

__device__ __forceinline__ double l_readlane(double v, int srclane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double l_uniform(double v) {
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double l_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double l_min(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double l_wave_min(double v) {
    v = l_min(v, l_dpp<0xB1>(v));
    v = l_min(v, l_dpp<0x4E>(v));
    v = l_min(v, l_dpp<0x141>(v));
    v = l_min(v, l_dpp<0x140>(v));
    v = l_min(v, l_dpp<0x142>(v));
    v = l_min(v, l_dpp<0x143>(v));
    return l_readlane(v, 63);
}
__device__ __forceinline__ double l_grp8_min(double v) {  // minimum over the 8 lanes of a group, in every lane of the group
    v = l_min(v, l_dpp<0xB1>(v));
    v = l_min(v, l_dpp<0x4E>(v));
    v = l_min(v, l_dpp<0x141>(v));
    return v;
}
__device__ __forceinline__ double l_pos(double x) {
    return (x > 0.0) ? x : ((x != x) ? x : 0.0);
}
__device__ __forceinline__ double l_poisson_time_L(double a, double b, double L) {  // src/poissontime.jl:8-30 with L = log(u)
    if (b == 0) return (a > 0) ? -L / a : L_INF;
    const double r = a / b;
    const double q = L * 2.0 / b;
    const double sq = sqrt((b > 0 && a < 0) ? -q : r * r - q);
    if (b > 0) return sq - r;
    if (a <= 0) return L_INF;
    if (-L <= -(a * a) / b + (a * a) / (2 * b)) return -sq - r;
    return L_INF;
}
__device__ __forceinline__ uint32_t l_wave_min_u32(uint32_t v) {
    auto step = [](uint32_t x, auto ctrl) -> uint32_t {
        const uint32_t o = (uint32_t)__builtin_amdgcn_mov_dpp((int)x, decltype(ctrl)::value, 0xf, 0xf, true);
        return (o < x) ? o : x;
    };
    v = step(v, std::integral_constant<int, 0xB1>{});
    v = step(v, std::integral_constant<int, 0x4E>{});
    v = step(v, std::integral_constant<int, 0x141>{});
    v = step(v, std::integral_constant<int, 0x140>{});
    v = step(v, std::integral_constant<int, 0x142>{});
    v = step(v, std::integral_constant<int, 0x143>{});
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
template <int CTRL, int ROWM, int BANKM>
__device__ __forceinline__ uint32_t l_dpp_id_u32(uint32_t identity, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)src, CTRL, ROWM, BANKM, false);
}
__device__ __forceinline__ uint32_t l_scan_add_u32(uint32_t v) {  // inclusive
    uint32_t x = v;
    x += l_dpp_id_u32<0x111, 0xf, 0xf>(0u, v);
    x += l_dpp_id_u32<0x112, 0xf, 0xf>(0u, v);
    x += l_dpp_id_u32<0x113, 0xf, 0xf>(0u, v);
    x += l_dpp_id_u32<0x114, 0xf, 0xe>(0u, x);
    x += l_dpp_id_u32<0x118, 0xf, 0xc>(0u, x);
    x += l_dpp_id_u32<0x142, 0xa, 0xf>(0u, x);
    x += l_dpp_id_u32<0x143, 0xc, 0xf>(0u, x);
    return x;
}
template <int CTRL, int ROWM, int BANKM>
__device__ __forceinline__ double l_dpp_inf(double src) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(src), CTRL, ROWM, BANKM, false);
    const int hi = __builtin_amdgcn_update_dpp(0x7FF00000, __double2hiint(src), CTRL, ROWM, BANKM, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double l_scan_min_f64(double v) {  // inclusive
    double x = v;
    x = l_min(x, l_dpp_inf<0x111, 0xf, 0xf>(v));
    x = l_min(x, l_dpp_inf<0x112, 0xf, 0xf>(v));
    x = l_min(x, l_dpp_inf<0x113, 0xf, 0xf>(v));
    x = l_min(x, l_dpp_inf<0x114, 0xf, 0xe>(x));
    x = l_min(x, l_dpp_inf<0x118, 0xf, 0xc>(x));
    x = l_min(x, l_dpp_inf<0x142, 0xa, 0xf>(x));
    x = l_min(x, l_dpp_inf<0x143, 0xc, 0xf>(x));
    return x;
}
__device__ __forceinline__ double l_shfl(double v, uint32_t src) {
    const int lo = __builtin_amdgcn_ds_bpermute((int)(src << 2), __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute((int)(src << 2), __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// LDS layout (bytes)
struct LL {
    static constexpr uint32_t NB2 = 8192;          // coordinate pairs (lines) a chain may have: d <= 16384
    static constexpr uint32_t IMG = 0;             // [NB2] u8: the low 8 bits of the block minimum's wheel position
    static constexpr uint32_t HB = 8192;           // [NB2] bits: the ninth, laid out so that a lane's 128 blocks are its own 16 bytes (hb_word / hb_bit)
    static constexpr uint32_t EX = 9216;           // [64] f64 what event e exposes; before that the candidates' keys where the ranks need them exactly
    static constexpr uint32_t SLB = 9728;          // [64] u16 event blocks, rank order
    static constexpr uint32_t TB = 9856;           // [64] u16 candidate blocks, compaction order; then the ranks' 15-bit images
    static constexpr uint32_t ACL = 9984;          // [8] u16 the accepted events
    static constexpr uint32_t BYTES = 10000;
};
static_assert(LL::BYTES <= 10240, "16 chains per CU: 160 KB / 16");
constexpr int L_CMAX = 64;      // candidates per iteration
constexpr uint32_t L_WIN = 128; // draws an iteration may consume (two per lane in registers)
constexpr int L_AMAX = 8;       // accepted events per iteration (one 8-lane group each)
constexpr int L_NHYP = 4;       // hypotheses of the accept chain's first guess
constexpr int32_t L_MMAX = 128; // quanta a window may span (the SWAR test takes at most 128)
// block minima per quantum the scale aims at, and the events an iteration's window aims at (pdmp_debug_set_helper_steering overrides both: A/B)
#ifndef L_TARGETQ
#define L_TARGETQ 30.0
#define L_TARGET 56.0
#endif

// block b's ninth bit: word and mask in the plane.  Lane o = (b >> 4) & 63 scans the 16-byte pieces o + 64 j (j = b >> 10) of the byte array; its
// 128 blocks' bits are words 4 o .. 4 o + 3, ordered like the flags its range test packs: bit 8 y + w + 4 (j & 1) of word j >> 1 for byte y of
// word w of piece j.
__device__ __forceinline__ uint32_t hb_word(uint32_t b) {
    return (((b >> 4) & 63u) << 2) + (b >> 11);
}
__device__ __forceinline__ uint32_t hb_bit(uint32_t b) {
    return 1u << (((b & 3u) << 3) + ((b >> 2) & 3u) + (((b >> 10) & 1u) << 2));
}
__device__ __forceinline__ void img_set(unsigned char* smem, uint32_t b, uint32_t w9) {
    smem[LL::IMG + b] = (unsigned char)(w9 & 255u);
    uint32_t* const hw = reinterpret_cast<uint32_t*>(smem + LL::HB) + hb_word(b);
    const uint32_t bit = hb_bit(b);
    if (w9 & 256u) atomicOr(hw, bit);
    else atomicAnd(hw, ~bit);
}
__device__ __forceinline__ uint32_t img_get(const unsigned char* smem, uint32_t b) {
    const uint32_t lo = (uint32_t)smem[LL::IMG + b];
    const uint32_t hw = reinterpret_cast<const uint32_t*>(smem + LL::HB)[hb_word(b)];
    return lo | ((hw & hb_bit(b)) ? 256u : 0u);
}
// the quantum of a time: floor((t − tref) s), saturated to 32 bits (+Inf and keys beyond the launch's range sit at the top: they are "a revolution
// or more ahead" for every front the launch can reach -- it pauses before 2^30).  Monotone in t: correctly rounded −, x and floor are.
__device__ __forceinline__ int32_t tl_q(double t, double tref, double s) {
    const double v = floor((t - tref) * s);
    return (v >= 2147483520.0) ? (int32_t)0x7fffffff : ((v <= -2147483648.0) ? (int32_t)0x80000000 : (int32_t)v);
}

}  // namespace

template <bool PROF>
__device__ __forceinline__ void trackl_body(const ZzRunParams& P) {
    const int lane0 = threadIdx.x & 63;
    int lane = lane0;
    const int64_t chain = blockIdx.x;
    const int64_t d = P.d;
    (void)d;  // (the invariant check of -DPDMP_TL_CHECK reads it)
    const uint32_t nb2p = (uint32_t)(P.dk / 2);  // lines of a chain (dk is a multiple of 64: whole 16-byte pieces of the byte array)
    const uint32_t nlat = (uint32_t)P.lattice_n, nmagic = P.lattice_magic;

    extern __shared__ __align__(16) unsigned char smem[];
    double* const EX = reinterpret_cast<double*>(smem + LL::EX);
    double* const KM = EX;  // (exact block minima of the candidates, until the events are set up)
    uint16_t* const SLB = reinterpret_cast<uint16_t*>(smem + LL::SLB);
    uint16_t* const TB = reinterpret_cast<uint16_t*>(smem + LL::TB);
    uint16_t* const ACL = reinterpret_cast<uint16_t*>(smem + LL::ACL);
    const uint4* const IMG4 = reinterpret_cast<const uint4*>(smem + LL::IMG);
    const uint4* const HB4 = reinterpret_cast<const uint4*>(smem + LL::HB);

    char* const lines = reinterpret_cast<char*>(reinterpret_cast<TrLine*>(P.tl_lines) + chain * (int64_t)nb2p);
    TrCold* const cold = reinterpret_cast<TrCold*>(P.tl_cold) + chain * P.dk;
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
    const double targetq = (P.hw_gain > 0.0) ? P.hw_gain : L_TARGETQ;
    const double raw_target = (P.hw_target != 0u) ? (double)P.hw_target : L_TARGET;

    // ---------------- the wheel: scale s (quanta per unit time), reference time, images of every block minimum
    const double tref = t_last;  // (wave-uniform, fixed for the launch)
    double s = hdr->tl_scale;
    int32_t F = 0;               // front: every key's quantum is >= F
    // one pass over the chain's lines: images (s > 0) and the exact minimum key; with counting = true the blocks below tmin0 + 2^-e, e = 2 .. 17
    auto pass = [&](const bool write, const bool counting, const double tmin0, uint32_t* cnt) -> double {
        double mloc = L_INF;
        for (int j = 0; j < 8; ++j) {
            const uint32_t u = (uint32_t)lane + 64u * (uint32_t)j;
            if (16u * u >= nb2p) break;
            uint32_t wv[4] = {0u, 0u, 0u, 0u};
            uint32_t hbits = 0u;
#pragma unroll
            for (uint32_t r = 0; r < 16u; ++r) {
                const uint32_t b = 16u * u + r;
                const double4 s0 = *reinterpret_cast<const double4*>(lines + (size_t)b * 128);
                const double mk = l_min(s0.x, s0.z);
                mloc = l_min(mloc, mk);
                if (write) {
                    const uint32_t w9 = (uint32_t)tl_q(mk, tref, s) & 511u;
                    wv[r >> 2] |= (w9 & 255u) << ((r & 3u) << 3);
                    if (w9 & 256u) hbits |= 1u << r;
                }
                if (counting) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) cnt[e] += (mk < tmin0 + __builtin_ldexp(1.0, -(e + 2))) ? 1u : 0u;
                }
            }
            if (write) {
                reinterpret_cast<uint4*>(smem + LL::IMG)[u] = make_uint4(wv[0], wv[1], wv[2], wv[3]);
                // the piece's 16 ninth bits into the lane's plane words: bit r = 4 w + y of the piece -> bit 8 y + w + 4 (j & 1) of word j >> 1
                uint32_t hp = 0u;
#pragma unroll
                for (uint32_t r = 0; r < 16u; ++r) hp |= ((hbits >> r) & 1u) << (((r & 3u) << 3) + (r >> 2) + (((uint32_t)j & 1u) << 2));
                uint32_t* const hw = reinterpret_cast<uint32_t*>(smem + LL::HB) + ((uint32_t)lane << 2) + ((uint32_t)j >> 1);
                const uint32_t keep = ((uint32_t)j & 1u) ? 0x0f0f0f0fu : 0xf0f0f0f0u;
                *hw = (*hw & keep) | hp;  // (the lane's own words: nobody else writes them here)
            }
        }
        return l_wave_min(mloc);
    };
    bool stalled0 = false;
    auto rebuild = [&]() {
        uint32_t dummy[16];
        const double tmin = pass(true, false, 0.0, dummy);
        L_ORDER();
        if (!(tmin < L_INF)) stalled0 = true;
        F = tl_q(tmin, tref, s);
    };
    {
        // (plane words start from zero: a lane's pass writes its own)
        reinterpret_cast<uint4*>(smem + LL::HB)[lane] = make_uint4(0u, 0u, 0u, 0u);
        for (uint32_t u = lane; u < LL::NB2 / 16u; u += 64u) reinterpret_cast<uint4*>(smem + LL::IMG)[u] = make_uint4(0u, 0u, 0u, 0u);
        L_ORDER();
        if (!(s > 0.0)) {
            // first launch of the chain: the density of block minima at the front -- the smallest Δ = 2^-e below which at least 96 blocks lie
            uint32_t cnt[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) cnt[e] = 0u;
            uint32_t dummy[16];
            const double tmin = pass(false, false, 0.0, dummy);
            if (tmin < L_INF) {
                (void)pass(false, true, tmin, cnt);
                double rho = 4.0;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)l_scan_add_u32(cnt[e]), 63);
                    if (tot >= 96u || e == 0) rho = (double)(tot > 0u ? tot : 1u) * __builtin_ldexp(1.0, e + 2);
                }
                s = l_uniform(rho / targetq);
            } else {
                s = 1.0;
            }
        }
        rebuild();
    }
    if (stalled0) {
        if (lane == 0) hdr->c.status = PDMP_CHAIN_STALLED;
        return;
    }
    int32_t m = 1;              // quanta of the next window
    double evq = targetq, rqf = 1.5 * targetq;  // events / candidates a NEW quantum brings, running means
    int32_t prev_base = 0;      // quanta of the current window that the window before it had looked at already
    double prev_left = 0.0;     // ... and the events in them that it did not commit
    double inv_s = 1.0 / s;
    uint32_t since_rescale = 0;

    uint64_t ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t ph_t0 = PROF ? (uint64_t)__builtin_readcyclecounter() : 0;
    uint64_t ph_rounds = 0, ph_guess = 0;
    uint64_t ph_iters = 0, ph_raw = 0, ph_zone = 0, ph_eval = 0, ph_nev = 0, ph_ties = 0, ph_cut1 = 0, ph_cut2 = 0, ph_alias = 0, ph_near = 0, ph_meff = 0;
#ifdef PDMP_PHASE_MARKS
#define LMARK(k) asm volatile("; LPHASE " #k)
#else
#define LMARK(k)
#endif
#define LPHASE(k)                                                         \
    do {                                                                  \
        LMARK(k);                                                         \
        if (PROF) {                                                       \
            const uint64_t now_ = (uint64_t)__builtin_readcyclecounter(); \
            ph[k] += now_ - ph_t0;                                        \
            ph_t0 = now_;                                                 \
        }                                                                 \
    } while (0)

#ifdef PDMP_TL_CHECK
    int tlc_bad = 0;
#endif
    PrioTurn prio;
    uint32_t idle = 0;  // consecutive iterations without an event
    bool running = stop_before || (t_event < T);
    while (running) {
        // (the lane index is made opaque once per iteration: LLVM otherwise hoists lane-only arithmetic -- masks, LDS addresses, the validity words
        // of the scan -- out of this loop, runs out of registers and SPILLS it; the reload inside the loop waits with vmcnt(0) for every store in flight)
        lane = lane0;
        asm volatile("" : "+v"(lane));
        const int g = lane >> 3, gl = lane & 7;
        prio.step();
        if (dnacc >= trace_room) {
            status = PDMP_CHAIN_TRACE_FULL;
            break;
        }
        if (dnm >= P.count_limit || F > (int32_t)0x3fffffff) {  // (32-bit counters and quanta of the launch: pause, the host runs again)
            status = PDMP_CHAIN_PAUSED;
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
        LPHASE(7);
        auto draw = [&](uint32_t n) -> double {  // draw nm0 + n for dnm <= n < dnm + L_WIN (every lane calls it: ds_bpermute)
            const double v0 = l_shfl(ureg[0], n & 63u), v1 = l_shfl(ureg[1], n & 63u);
            return ((n >> 6) & 1u) ? v1 : v0;
        };
        auto drawlog = [&](uint32_t n) -> double { return pdmp_log(draw(n)); };
#ifdef PDMP_TL_CHECK  // (debug build: the wheel's invariant -- an image is never beyond its line's minimum -- checked against the lines, every iteration)
        {
            const uint32_t fwc = (uint32_t)F & 511u;
            for (uint32_t b = lane; b < nb2p; b += 64) {
                const double4 s0 = *reinterpret_cast<const double4*>(lines + (size_t)b * 128);
                const double mk = l_min(s0.x, s0.z);
                const int64_t R = (int64_t)tl_q(mk, tref, s) - (int64_t)F;
                const uint32_t r = (img_get(smem, b) - fwc) & 511u;
                if (R < (int64_t)r && tlc_bad < 3) { tlc_bad += 1; printf("TLCHECK chain %d iter %u block %u r %u R %lld F %d key %.17g tlast %.17g dnum %u\n", (int)chain, (unsigned)prio.it, b, r, (long long)R, F, mk, t_last, dnum); }
            }
        }
#endif
        // ---------------- select: every block whose image lies in the window [F, F + meff) of the wheel
        const int32_t QT = stop_before ? tl_q(T, tref, s) : (int32_t)0x7fffffff;
        if (stop_before && QT < F) break;  // every key is at or beyond T
        uint32_t Cc = 0;
        int32_t meff = m;
        bool crowded = false;
        {
            uint32_t wv[32];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint4 v = IMG4[lane + 64 * j];
                wv[4 * j + 0] = v.x;
                wv[4 * j + 1] = v.y;
                wv[4 * j + 2] = v.z;
                wv[4 * j + 3] = v.w;
            }
            const uint4 hb4 = HB4[lane];
            const uint32_t hbw[4] = {hb4.x, hb4.y, hb4.z, hb4.w};
            const uint32_t fw = (uint32_t)F & 511u, lo = fw & 255u;
            const uint32_t FHX = (fw & 256u) ? 0u : ~0u;  // (plane word ^ FHX) has a one where the ninth bit equals the front's
            // the window does not cross the half-revolution (the ninth bit of every block in it is the front's) nor QT
            if (meff > (int32_t)(256u - lo)) meff = (int32_t)(256u - lo);
            if (meff > L_MMAX) meff = L_MMAX;
            if (stop_before && (int64_t)QT - (int64_t)F + 1 < (int64_t)meff) meff = (int32_t)((int64_t)QT - (int64_t)F + 1);
            // validity of the lane's blocks as the packed flags order them (word q covers the 16-byte pieces lane + 64 (2 q) and lane + 64 (2 q + 1))
            uint32_t VM[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t u0 = (uint32_t)lane + 64u * (uint32_t)(2 * q), u1 = u0 + 64u;
                VM[q] = ((16u * u0 < nb2p) ? 0x0f0f0f0fu : 0u) | ((16u * u1 < nb2p) ? 0xf0f0f0f0u : 0u);
            }
            uint32_t PK[4];
            for (int tries = 0;; ++tries) {
                // bytes y = (image − lo) mod 256 < meff, four per word: per-byte subtraction without borrows between the bytes, then y's bit 7 clear
                // and (y & 127) + (128 − meff) without a carry into bit 7
                const uint32_t LO4 = lo * 0x01010101u, L7 = LO4 & 0x7f7f7f7fu, NLH = ~LO4 & 0x80808080u;
                const uint32_t K4 = (uint32_t)(128 - meff) * 0x01010101u;
                uint32_t accn[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int wi = 0; wi < 32; ++wi) {
                    const uint32_t x = wv[wi];
                    const uint32_t z = ((x | 0x80808080u) - L7) ^ (x & 0x80808080u) ^ NLH;
                    const uint32_t un = ((z & 0x7f7f7f7fu) + K4) | z;  // bit 7 of a byte: it is NOT in the window
                    const int j = wi >> 2, w = wi & 3;
                    const int pos = w + 4 * (j & 1);
                    accn[j >> 1] |= (un >> (7 - pos)) & (0x01010101u << pos);
                }
                uint32_t ncl = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    PK[q] = ~accn[q] & VM[q] & (hbw[q] ^ FHX);
                    ncl += (uint32_t)__builtin_popcount(PK[q]);
                }
                const uint32_t incl = l_scan_add_u32(ncl);
                Cc = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                if (Cc <= (uint32_t)L_CMAX) {
                    uint32_t ix = incl - ncl;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        uint32_t m_ = PK[q];
                        while (__ballot(m_ != 0u) != 0) {
                            if (m_ != 0u) {
                                const uint32_t bit = (uint32_t)(__ffs((int)m_) - 1);
                                const uint32_t jj = 2u * (uint32_t)q + ((bit >> 2) & 1u);
                                TB[ix] = (uint16_t)(16u * ((uint32_t)lane + 64u * jj) + 4u * (bit & 3u) + (bit >> 3));
                                ix += 1;
                                m_ &= m_ - 1u;
                            }
                        }
                    }
                    break;
                }
                if (meff == 1 || tries >= 8) {
                    crowded = true;  // one quantum holds more blocks than the wave has lanes: the scale is too coarse here
                    break;
                }
                meff = (Cc > 96u && meff > 2) ? meff / 2 : meff - 1;  // (windows are a few quanta: one less is the list just shorter)
            }
        }
        if (crowded) {
            if (!(s < 1e15)) {
                status = PDMP_CHAIN_STALLED;  // (more than 64 block minima that no scale separates: keys tied by construction)
                break;
            }
            s = l_uniform(s * 2.0);
            inv_s = 1.0 / s;
            evq = evq * 0.5;
            rqf = rqf * 0.5;
            since_rescale = 0;
            rebuild();
            m = 1;
            prev_base = 0;
            prev_left = 0.0;
            if (++idle > 4096u) {
                status = PDMP_CHAIN_STALLED;
                break;
            }
            continue;
        }
        L_ORDER();
        LPHASE(8);
        // ---------------- candidate lane c: ITS line -- both coordinates' pairs, sums and constants: eight 16-byte loads, the lines are distinct
        const bool isc = (uint32_t)lane < Cc;
        const uint32_t cblk = isc ? (uint32_t)TB[lane] : 0u;
        double c_km = L_INF, c_rs = L_INF, c_tp = 0.0, c_th = 0.0, c_g = 0.0, c_gd = 0.0, c_tg = 0.0, c_c = 0.0, c_c100 = 0.0;
        uint32_t c_pb = 0u;
        {
            const char* const ln = lines + (size_t)cblk * 128;
            const double2 k0 = *reinterpret_cast<const double2*>(ln + 0), k1 = *reinterpret_cast<const double2*>(ln + 16);
            const double2 a0 = *reinterpret_cast<const double2*>(ln + 32), b0 = *reinterpret_cast<const double2*>(ln + 48);
            const double2 a1 = *reinterpret_cast<const double2*>(ln + 64), b1 = *reinterpret_cast<const double2*>(ln + 80);
            const double2 cc0 = *reinterpret_cast<const double2*>(ln + 96), cc1 = *reinterpret_cast<const double2*>(ln + 112);
            const bool one = k1.x < k0.x;  // (the lower coordinate on ties)
            if (isc) {
                c_km = one ? k1.x : k0.x;
                c_rs = one ? k0.x : k1.x;
                c_tp = one ? k1.y : k0.y;
                c_th = one ? a1.x : a0.x;
                c_g = one ? a1.y : a0.y;
                c_gd = one ? b1.x : b0.x;
                c_tg = one ? b1.y : b0.y;
                c_c = one ? cc1.x : cc0.x;
                c_c100 = one ? cc1.y : cc0.y;
                c_pb = one ? 1u : 0u;
            }
        }
        // ---------------- events = candidates whose exact minimum lies in the window; everybody refreshes its image
        const int32_t c_q = tl_q(c_km, tref, s);
#ifdef PDMP_TL_CHECK
        if (isc && chain == (int64_t)P.dbg_cap && cblk == (uint32_t)P.hw_ahead) printf("W refresh it %u F %d q %d km %.17g\n", (unsigned)prio.it, F, c_q, c_km);
#endif
        if (isc) img_set(smem, cblk, (uint32_t)c_q & 511u);
        const bool isev = isc && ((int64_t)c_q - (int64_t)F < (int64_t)meff) && (!stop_before || c_km < T);
        const double own = isev ? c_km : L_INF;
        // rank = the number of events with a smaller key: on 15-bit images of the keys first (pdmp_trackp.hip), exactly where two images meet
        const uint64_t evb = __ballot(isev);
        int nev = __popcll(evb);
        const int nev0 = nev;
        uint32_t rank = 0, rsum = 0xffffffffu;
        {
            uint16_t* const QK = TB;  // (the candidates' blocks are in registers by now)
            const double tlo = tref + (double)F * inv_s;
            const double scale = 32766.0 * s * __builtin_amdgcn_rcp((double)meff);
            const double img = (own - tlo) * scale;
            const uint32_t qi = isev ? (uint32_t)l_pos(img) : 32767u;
            const uint32_t qk = isev ? ((qi < 32766u) ? qi : 32766u) : 32767u;
            L_ORDER();
            QK[lane] = (uint16_t)qk;
            L_ORDER();
            typedef short pk16 __attribute__((ext_vector_type(2)));
            typedef unsigned short upk16 __attribute__((ext_vector_type(2)));
            const pk16 own2 = {(short)qk, (short)qk};
            upk16 cnt = {0, 0};
            const uint4* const QK4 = reinterpret_cast<const uint4*>(QK);
#pragma unroll
            for (uint32_t h = 0; h < 2u; ++h) {
                if (h == 0u || Cc > 32u) {
                    const uint4 w0 = QK4[4u * h + 0u], w1 = QK4[4u * h + 1u], w2 = QK4[4u * h + 2u], w3 = QK4[4u * h + 3u];
                    const uint32_t ww[16] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z, w3.w};
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const pk16 k2 = __builtin_bit_cast(pk16, ww[q]);
                        const pk16 df = k2 - own2;
                        cnt += __builtin_bit_cast(upk16, df) >> (unsigned short)15;
                    }
                }
            }
            rank = (uint32_t)cnt.x + (uint32_t)cnt.y;
            L_ORDER();
            rsum = (uint32_t)__builtin_amdgcn_readlane((int)l_scan_add_u32(isev ? rank : 0u), 63);
        }
        if (rsum != (uint32_t)(nev * (nev - 1) / 2)) {
            KM[lane] = own;
            L_ORDER();
            rank = 0;
            for (uint32_t m0 = 0; m0 < Cc; m0 += 8) {
                double km[8];
#pragma unroll
                for (uint32_t q = 0; q < 8; ++q) km[q] = KM[m0 + q];
#pragma unroll
                for (uint32_t q = 0; q < 8; ++q) rank += (km[q] < own) ? 1u : 0u;
            }
            L_ORDER();
            rsum = (uint32_t)__builtin_amdgcn_readlane((int)l_scan_add_u32(isev ? rank : 0u), 63);
        }
        // exactly equal keys among the events: one event this iteration, the tied minimum of the lowest block
        bool slot = isev;
        if (rsum != (uint32_t)(nev * (nev - 1) / 2)) {
            const double mn = l_wave_min(own);
            const uint32_t bsel = l_wave_min_u32((isev && own == mn) ? cblk : 0xffffffffu);
            slot = isev && cblk == bsel;
            rank = 0;
            nev = 1;
            if (PROF) ph_ties += 1;
        }
        // ---------------- lane r = event r: everything moves over from its candidate's lane, pushed to lane `rank` (ds_permute)
        const uint32_t pdst = ((slot && rank < (uint32_t)nev) ? rank : 63u) << 2;
        auto push32 = [&](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_ds_permute((int)pdst, (int)v); };
        auto push64 = [&](double v) -> double {
            const int lo = __builtin_amdgcn_ds_permute((int)pdst, __double2loint(v)), hi = __builtin_amdgcn_ds_permute((int)pdst, __double2hiint(v));
            return __hiloint2double(hi, lo);
        };
        const double e_km = push64(c_km), e_rs = push64(c_rs), e_tp = push64(c_tp);
        const double e_th = push64(c_th), e_g = push64(c_g), e_gd = push64(c_gd), e_tg = push64(c_tg), e_c = push64(c_c), e_c100 = push64(c_c100);
        const uint32_t e_bu = push32(cblk | (c_pb << 16));
        L_ORDER();
        LPHASE(9);
        int C = nev;
        if (PROF) ph_iters += 1;
        if (PROF) ph_raw += (uint64_t)Cc;
        if (PROF) ph_nev += (uint64_t)nev0;
        if (PROF) ph_alias += (uint64_t)__popcll(__ballot(isc && ((int64_t)c_q - (int64_t)F >= 512)));
        if (PROF) ph_near += (uint64_t)__popcll(__ballot(isc && !isev && ((int64_t)c_q - (int64_t)F < 512)));
        if (PROF) ph_meff += (uint64_t)meff;
        asm volatile("" ::"v"(c_th), "v"(c_g), "v"(c_gd), "v"(c_tg), "v"(c_c), "v"(c_c100));
        if (C == 0) {
            // no key in the window (stale or revolved images refreshed): the front moves to its end
            if (stop_before && (int64_t)QT - (int64_t)F < (int64_t)meff) break;  // the window reached T: every key is at or beyond T
            if (++idle > 4096u) {
                status = PDMP_CHAIN_STALLED;
                break;
            }
            if ((idle & 63u) == 0u) {  // a long empty stretch: the exact front from the lines (and the chain without a finite key)
                rebuild();
                if (stalled0) {
                    status = PDMP_CHAIN_STALLED;
                    break;
                }
                m = 1;
                prev_base = 0;
                prev_left = 0.0;
                continue;
            }
            F += meff;
            m = (2 * meff < L_MMAX) ? 2 * meff : L_MMAX;
            prev_base = 0;
            prev_left = 0.0;
            continue;
        }
        bool ev = lane < C;
        const double tp = ev ? e_km : L_INF;  // the event time: the exact block minimum
        const double rest = ev ? e_rs : L_INF;
        const double tprop_i = ev ? e_tp : 0.0;
        const uint32_t blk = e_bu & 0xffffu, pbe = e_bu >> 16;
        const double th = e_th, g_i = e_g, gd_i = e_gd, tg_i = e_tg;
        const double c_i = e_c, c100_i = e_c100;
        const uint32_t i = ev ? (blk * 2u + (pbe & 1u)) : 0u;
        L_ORDER();
        if (ev) SLB[lane] = (uint16_t)blk;
        L_ORDER();
        LPHASE(0);
        uint32_t rc_i = 0xffffu, k_i;
        {
            // lattice coordinates packed for the zone test: byte 0 = row, byte 1 = column
            const uint32_t col_i = __umulhi(i, nmagic);
            const uint32_t row_i = i - col_i * nlat;
            rc_i = ev ? (row_i | (col_i << 8)) : 0xffffu;
            k_i = 1u + (col_i > 0u ? 1u : 0u) + (row_i > 0u ? 1u : 0u) + (row_i + 1u < nlat ? 1u : 0u) + (col_i + 1u < nlat ? 1u : 0u);
        }
        L_ORDER();
        // ---------------- rates from the tracked sums (src/sfact.jl:116-119 with g_i(t′) = g_i + gd_i (t′ − tg_i))
        const double g_now = g_i + gd_i * (tp - tg_i);
        const double l = l_pos(g_now * th);
        // the bound in force (src/fact_samplers.jl:50-54), re-derived from the stored operands (pdmp_trackp.hip)
        const double told_i = tprop_i;
        const double a_i = c_i + (g_i + gd_i * (told_i - tg_i)) * th;
        const double b_i = c100_i + th * gd_i;
        const double lbound = l_pos(a_i + b_i * (tp - told_i));
        // ---------------- accept chain: offsets and outcomes as a fix-point
        const uint32_t ex_i = k_i - 1u;
        uint32_t off = 2u * (uint32_t)lane;
        bool acc = false;
        {
            const uint32_t etyp = P.typ_extra;
            const uint64_t pl1 = __ballot(ev && (ex_i & 1u)), pl2 = __ballot(ev && (ex_i & 2u)), pl4 = __ballot(ev && (ex_i & 4u));
            auto offsets = [&](uint64_t ab) -> uint32_t {
                const uint64_t a1 = ab & pl1, a2 = ab & pl2, a4 = ab & pl4;
                const uint32_t n1 = __builtin_amdgcn_mbcnt_hi((uint32_t)(a1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)a1, 0u));
                const uint32_t n2 = __builtin_amdgcn_mbcnt_hi((uint32_t)(a2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)a2, 0u));
                const uint32_t n4 = __builtin_amdgcn_mbcnt_hi((uint32_t)(a4 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)a4, 0u));
                return 2u * (uint32_t)lane + n1 + 2u * n2 + 4u * n4;
            };
            uint64_t accb = 0;
            const uint64_t tg0_ = PROF ? (uint64_t)__builtin_readcyclecounter() : 0;
            {
                uint64_t hb[L_NHYP];
                double uj[L_NHYP];
#pragma unroll
                for (int j = 0; j < L_NHYP; ++j) {
                    const uint32_t oj = off + etyp * (uint32_t)j;
                    uj[j] = draw(dnm + ((oj < L_WIN - 1u) ? oj : L_WIN - 1u));
                }
#pragma unroll
                for (int j = 0; j < L_NHYP; ++j) {
                    const uint32_t oj = off + etyp * (uint32_t)j;
                    hb[j] = __ballot((int)ev & (int)(oj + 1u + k_i <= L_WIN) & (int)(uj[j] * lbound < l));
                }
                const uint64_t odd = __ballot(ev && ex_i != etyp);
                uint64_t live = ~0ull;
#pragma unroll
                for (int j = 0; j < L_NHYP; ++j) {
                    const uint64_t rem = hb[j] & live;
                    const uint64_t low = rem & (0ull - rem);
                    accb |= low;
                    live = (low & ~odd) ? ~(low | (low - 1ull)) : 0ull;
                }
                if (accb != 0ull) off = offsets(accb);
            }
            if (PROF) ph_guess += (uint64_t)__builtin_readcyclecounter() - tg0_;
            for (int round = 0; round < 72; ++round) {
                if (PROF) ph_rounds += 1;
                const bool inwin = ev && (off + 1u + k_i <= L_WIN);
                const double u = draw(dnm + ((off < L_WIN - 1u) ? off : L_WIN - 1u));
                acc = inwin && (u * lbound < l);  // :121
                const uint64_t nb_ = __ballot(acc);
                if (nb_ == accb) break;
                accb = nb_;
                off = offsets(accb);
            }
        }
        const uint32_t cost = ev ? (acc ? (1u + k_i) : 2u) : 0u;
        // events whose draws would leave the ring wait for the next iteration
        {
            const uint64_t outb = __ballot(ev && !(off + 1u + k_i <= L_WIN));
            if (outb) {
                const int cut = __ffsll((unsigned long long)outb) - 1;
                C = (cut < C) ? cut : C;
            }
        }
        if (PROF) ph_cut1 += (uint64_t)C;
        LPHASE(1);
        // at most L_AMAX accepted events per iteration: the candidate list ends before the next one
        {
            uint64_t ab = __ballot(acc) & ((C < 64) ? ((1ull << C) - 1ull) : ~0ull);
            if (__popcll(ab) > L_AMAX) {
                uint64_t m_ = ab;
                for (int q = 0; q < L_AMAX; ++q) m_ &= m_ - 1;
                C = __ffsll((unsigned long long)m_) - 1;
            }
        }
        if (PROF) ph_cut2 += (uint64_t)C;
        uint32_t rekey_by = 0xffffffffu;  // the first accepted later event that re-bounds this lane's coordinate (its key is then not this lane's to store)
        // ---------------- zones: an ACCEPTED event m disturbs a later event r within lattice distance 1 (r's sums change), at distance 2 if r is
        // accepted too (they share a neighbour).  The list ends at the first disturbed event.
        {
            uint64_t confb = 0;
            uint64_t ab = __ballot(acc) & ((C < 64) ? ((1ull << C) - 1ull) : ~0ull);
            while (ab) {
                const int mm = __ffsll((unsigned long long)ab) - 1;
                ab &= ab - 1;
                const uint32_t rcm = (uint32_t)__builtin_amdgcn_readlane((int)rc_i, mm);
                const uint32_t sad = __builtin_amdgcn_sad_u8(rc_i, rcm, 0u);
                const uint64_t hit = __ballot(sad <= 1u || (sad <= 2u && acc));
                confb |= hit & (~0ull << (mm + 1));
                // an EARLIER rejected event next to accepted event m: m's group writes that coordinate's new key after it
                if (sad <= 1u && lane < mm && (uint32_t)mm < rekey_by) rekey_by = (uint32_t)mm;
            }
            const uint64_t cb = confb & ((C < 64) ? ((1ull << C) - 1ull) : ~0ull);
            if (cb) {
                const int c0 = __ffsll((unsigned long long)cb) - 1;
                C = (c0 < C) ? c0 : C;
            }
        }
        if (PROF) ph_zone += (uint64_t)C;
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
        L_ORDER();
        LPHASE(2);
        // ---------------- accepted events, one 8-lane group each: members of G1[i] (ascending, :131-135); the groups of the accepted events are the
        // LAST groups of the wave, in event order: the low lanes -- lane r = event r -- re-bound their rejected proposals in the same evaluation
        double key2 = L_INF;  // the new key of this lane's rejected proposal (event lanes)
        const int g0 = 8 - nacc_it;
        const bool gact = g >= g0;
        const uint32_t ea = gact ? (uint32_t)ACL[g - g0] : 0u;  // the event of this lane's group
        const uint32_t ia_b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ea << 2), (int)i);
        const uint32_t off_b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ea << 2), (int)off);
        const uint32_t ia = gact ? ia_b : 0u;
        const uint32_t blka = gact ? (uint32_t)SLB[ea] : 0u;
        const double tpa_b = l_shfl(tp, ea);
        const double tpa = gact ? tpa_b : 0.0;
        const uint32_t offa = gact ? off_b : 0u;
        uint32_t jm = ia;
        bool mem = false;
        {
            // G1[ia] on the lattice, ascending: {ia − n, ia − 1, ia, ia + 1, ia + n} inside the grid
            const uint32_t cola = __umulhi(ia, nmagic), rowa = ia - cola * nlat;
            const bool hasL = cola > 0u, hasU = rowa > 0u, hasD = rowa + 1u < nlat, hasR = cola + 1u < nlat;
            const uint32_t ka = gact ? (1u + (hasL ? 1u : 0u) + (hasU ? 1u : 0u) + (hasD ? 1u : 0u) + (hasR ? 1u : 0u)) : 0u;
            mem = gact && (uint32_t)gl < ka;
            const uint32_t cand5[5] = {ia - nlat, ia - 1u, ia, ia + 1u, ia + nlat};
            const bool has5[5] = {hasL, hasU, true, hasD, hasR};
            uint32_t seen = 0;
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                if (has5[q]) {
                    if (seen == (uint32_t)gl && mem) jm = cand5[q];
                    seen += 1;
                }
            }
        }
        char* const lj = lines + (size_t)(jm >> 1) * 128;   // the member's line, its half
        char* const lia = lines + (size_t)(ia >> 1) * 128;  // the reflecting coordinate's
        const uint32_t pj = jm & 1u, pa = ia & 1u;
        TrCold* const cia = cold + ia;
        double gam = 0.0;  // Γ[jm, ia]: member gl of G1[ia] (shared table, L2: requested next to the members' lines)
        if (mem) gam = P.tb.cc_shared[ia].gam[gl < 5 ? gl : 0];
        const double th_ia = *reinterpret_cast<const double*>(lia + 32 + 32 * pa);
        const double2 cx = *reinterpret_cast<const double2*>(&cia->x);   // (x, tx)
        const double2 cI = *reinterpret_cast<const double2*>(&cia->I);   // (I, acc)
        double xa = cx.x, txa = cx.y, Ia = cI.x;
        const uint64_t acc_ia = (uint64_t)__double_as_longlong(cI.y);
        const double2 sj0 = *reinterpret_cast<const double2*>(lj + 32 + 32 * pj);  // (θ, g)
        double2 sj1 = *reinterpret_cast<const double2*>(lj + 48 + 32 * pj);        // (gd, tg)
        double2 cjm2 = *reinterpret_cast<const double2*>(lj + 96 + 16 * pj);       // (c, c / 100)
        double kmate = *reinterpret_cast<const double*>(lj + 16 * (pj ^ 1u));      // the member's pair mate's key (same line: the line's new image, exactly)
        asm volatile("" : "+v"(cjm2.x), "+v"(cjm2.y), "+v"(sj1.x), "+v"(sj1.y), "+v"(kmate));
        const double thj0 = sj0.x, gj0 = sj0.y, gdj0 = sj1.x, tgj = sj1.y;
        const double resta_b = l_shfl(rest, ea);  // the accepted event's pair mate's key
        // ---------------- ONE evaluation of the new bound and key per lane: the re-bound of a rejected proposal (:137-140) in its event lane, the
        // re-bound of a member of G1 (:131-135) in its group lane.  A lane that is both evaluates its rejected proposal again below.
        const bool selfl = mem && jm == ia;
        const double thj = selfl ? -th_ia : thj0;
        const double gj = gj0 + gdj0 * (tpa - tgj);
        const double gdj = gdj0 + gam * (-2.0 * th_ia);  // θ_i -> −θ_i
        double key2l;
        {
            const bool mine = gact;
            const uint32_t dix = mine ? (offa + 1u + (uint32_t)gl) : (off + 1u);
            const double Lg = drawlog(dnm + ((dix < L_WIN - 1u) ? dix : L_WIN - 1u));
            const double cc = mine ? cjm2.x : c_i, cc100 = mine ? cjm2.y : c100_i;
            const double gg = mine ? gj : g_now, tt = mine ? thj : th, gdd = mine ? gdj : gd_i;
            const double a2l = cc + gg * tt;
            const double b2l = cc100 + tt * gdd;
            key2l = (mine ? tpa : tp) + l_poisson_time_L(a2l, b2l, Lg);
        }
        const double keyj = mem ? key2l : L_INF;
        if (__ballot(ev && !acc && gact) != 0) {
            const double Le = drawlog(dnm + ((off + 1u < L_WIN - 1u) ? off + 1u : L_WIN - 1u));
            const double a2e = c_i + g_now * th;
            const double b2e = c100_i + th * gd_i;
            const double k2e = tp + l_poisson_time_L(a2e, b2e, Le);
            if (gact) key2l = k2e;
        }
        key2 = key2l;
        if (selfl) {  // event(i, t, x, θ, F) (src/sfact.jl:50-52): x_i at t′
            const double dtx = tpa - txa;
            const double xn = xa + th_ia * dtx;
            Ia = Ia + dtx * ((xa + xn) * 0.5);
            xa = xn;
            txa = tpa;
        }
        // the accepted event's line: its mate's key, or -- where the mate is a member of G1[ia] -- the mate's new key
        const double kin = (mem && (jm >> 1) == blka) ? keyj : L_INF;  // (ia itself and, on the lattice, its mate)
        const double kinmin = l_grp8_min(kin);
        const bool mate_in = l_grp8_min((mem && jm == (ia ^ 1u)) ? 0.0 : 1.0) == 0.0;  // the mate is re-bounded by this event: its old key is void
        const double rowmin_a = mate_in ? kinmin : l_min(resta_b, kinmin);
        const double keymin = l_grp8_min(keyj);
        if (gact && gl == 0) EX[ea] = l_min(rowmin_a, keymin);
        asm volatile("" ::"v"(xa), "v"(txa), "v"(Ia), "v"(acc_ia));
        // new minimum of the line of a rejected event, and what the event exposes
        double rowmin = L_INF;
        if (ev && !acc) rowmin = l_min(key2, rest);
        if (ev && !acc) EX[lane] = rowmin;
        L_ORDER();
        LPHASE(3);
        // ---------------- validate: all earlier events commit, zones disjoint, nothing produced or exposed earlier than t′
        uint32_t Rc;
        {
            const double prev = (lane > 0 && lane <= C) ? EX[lane - 1] : L_INF;  // what event lane − 1 exposes
            const double pref = l_scan_min_f64(prev);                            // exclusive prefix minimum
            const bool okr = ev && (lane == 0 || pref > tp);
            const uint64_t bad = ~__ballot(okr);
            const uint32_t r_ok = bad ? (uint32_t)(__ffsll((unsigned long long)bad) - 1) : 64u;
            Rc = (r_ok < (uint32_t)C) ? r_ok : (uint32_t)C;
            if (vsel != (int)Rc) vsel = -1;  // the violating proposal counts only once everything before it is committed
            const uint64_t accc = accball & ((Rc < 64u) ? ((1ull << Rc) - 1ull) : ~0ull);
            const bool isacc_c = ((accc >> lane) & 1ull) != 0ull;
            const uint32_t na_l = __builtin_amdgcn_mbcnt_hi((uint32_t)(accc >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)accc, 0u)) + 1u;
            const uint64_t fullb = __ballot(isacc_c && P.trace_cap > 0 && dnacc + na_l >= trace_room);
            const uint64_t endb = __ballot(isacc_c && !stop_before && !(tp < T));
            const uint64_t stopb = fullb | endb;
            if (stopb) {
                const int r = __ffsll((unsigned long long)stopb) - 1;
                if ((fullb >> r) & 1ull) status = PDMP_CHAIN_TRACE_FULL;
                if ((endb >> r) & 1ull) running = false;
                Rc = (uint32_t)r + 1u;
                vsel = -1;
            }
        }
        LPHASE(4);
        // the front after this iteration: the quantum of the last committed time (every key that remains is at or beyond it).  An image written
        // from here on must not lie behind it -- the front passes the window's positions without looking at them again
        int32_t Fn = F;
        if (Rc > 0u) {
            const int32_t fq = tl_q(l_readlane(tp, (int)(Rc - 1u)), tref, s);
            Fn = (fq > F) ? fq : F;
        }
        auto img_pos = [&](double key) -> uint32_t {
            const int32_t q = tl_q(key, tref, s);
            return (uint32_t)((q > Fn) ? q : Fn) & 511u;
        };
        // ---------------- commit the valid prefix
        const bool commit = ev && (uint32_t)lane < Rc;
        if (commit && !acc) {  // a rejected proposal: ONE 16-byte store into the line it read, and the line's new image
            if (!(rekey_by < Rc)) *reinterpret_cast<double2*>(lines + (size_t)blk * 128 + 16 * (pbe & 1u)) = make_double2(key2, tp);
#ifdef PDMP_TL_CHECK
            if (chain == (int64_t)P.dbg_cap && blk == (uint32_t)P.hw_ahead) printf("W reject it %u F %d q %d rowmin %.17g key2 %.17g rest %.17g lane %d\n", (unsigned)prio.it, F, tl_q(rowmin, tref, s), rowmin, key2, rest, lane);
#endif
            img_set(smem, blk, img_pos(rowmin));
        }
        const uint64_t acc_c = accball & ((Rc < 64u) ? ((1ull << Rc) - 1ull) : ~0ull);
        const bool gcommit = gact && ea < Rc;
        if (gcommit) {
            if (mem) {
                *reinterpret_cast<double*>(lj + 40 + 32 * pj) = gj;
                *reinterpret_cast<double2*>(lj + 48 + 32 * pj) = make_double2(gdj, tpa);
                *reinterpret_cast<double2*>(lj + 16 * pj) = make_double2(keyj, tpa);  // (the bound of every member is computed now)
            }
            if (selfl) {
                *reinterpret_cast<double*>(lia + 32 + 32 * pa) = -th_ia;
                *reinterpret_cast<double2*>(&cia->x) = make_double2(xa, txa);
                *reinterpret_cast<double2*>(&cia->I) = make_double2(Ia, __longlong_as_double((long long)(acc_ia + 1)));
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
#ifdef PDMP_TL_CHECK
            if (gl == 0 && chain == (int64_t)P.dbg_cap && blka == (uint32_t)P.hw_ahead) printf("W group it %u F %d q %d rowmin_a %.17g ea %u\n", (unsigned)prio.it, F, tl_q(rowmin_a, tref, s), rowmin_a, ea);
#endif
            if (gl == 0) img_set(smem, blka, img_pos(rowmin_a));
        }
        L_ORDER();
        LPHASE(5);
        // ---------------- images of the lines of re-bounded neighbours: LOWERED where the new key's quantum is below them (a key that rose leaves
        // its line's image stale low: a look in vain later, nothing else).  An image INSIDE this iteration's window cannot be compared -- it is
        // the refreshed image of a key that this re-bound has just replaced, or of one a revolution ahead, and the front is about to pass it --:
        // it becomes the new front's position, i.e. "look at the line next time".  Two lanes may aim at one line (two accepted events whose
        // neighbours are pair mates): every lane checks afterwards that the image is not above its key, and those that find it so repeat one at a time.
        {
            const bool want0 = gcommit && mem && (jm >> 1) != blka;
            if (__ballot(want0) != 0) {
                const int64_t rn64 = (int64_t)tl_q(keyj, tref, s) - (int64_t)F;
                const uint32_t rn = (rn64 < 512) ? (uint32_t)rn64 : 512u;  // (512: a revolution or more ahead -- below no image)
                const uint32_t bj = want0 ? (jm >> 1) : 0u;
                const uint32_t fw = (uint32_t)F & 511u;
                const uint32_t dn = (uint32_t)(Fn - F);
                const uint32_t cur = img_get(smem, bj);
                const uint32_t rcur = (cur - fw) & 511u;
                const bool inw = rcur < (uint32_t)meff;
                // the pair mate's key as the line held it: beyond this window it is no event of this iteration, hence still the mate's key --
                // the line's minimum is known exactly and its image is SET (a key that rose leaves nothing stale behind)
                const bool trust = want0 && ((int64_t)tl_q(kmate, tref, s) - (int64_t)F >= (int64_t)meff);
                const uint32_t wx = img_pos(l_min(keyj, kmate));
#ifdef PDMP_TL_CHECK
                if (want0 && chain == (int64_t)P.dbg_cap && bj == (uint32_t)P.hw_ahead) printf("W lower it %u F %d rn %u cur %u rcur %u keyj %.17g jm %u kmate %.17g trust %d\n", (unsigned)prio.it, F, rn, cur, rcur, keyj, jm, kmate, (int)trust);
#endif
                // two lanes aiming at one line (the two coordinates of a pair, re-bounded by two accepted events) would mix their bytes and ninth
                // bits: every acting lane claims its line in a 64-entry table first (EX is free by now); where a claim is lost -- or two lines
                // share an entry -- the lanes act one after the other, each on what the one before left (lowering only: the mate's key is void)
                const bool act = want0 && (trust || inw || rn < rcur);
                unsigned char* const CL = smem + LL::EX;
                if (act) CL[bj & 63u] = (unsigned char)lane;
                L_ORDER();
                const bool lost = act && CL[bj & 63u] != (unsigned char)lane;
                if (__ballot(lost) == 0) {
                    if (act) img_set(smem, bj, trust ? wx : (((inw ? dn : rn) + fw) & 511u));
                } else {
                    uint64_t todo = __ballot(want0);
                    while (todo) {
                        const int L0 = __ffsll((unsigned long long)todo) - 1;
                        todo &= todo - 1;
                        if (lane == L0) {
                            const uint32_t rc2 = (img_get(smem, bj) - fw) & 511u;
                            const bool inw2 = rc2 < (uint32_t)meff;
                            if (inw2 || rn < rc2) img_set(smem, bj, ((inw2 ? dn : rn) + fw) & 511u);
                        }
                        L_ORDER();
                    }
                }
            }
        }
        L_ORDER();
        LPHASE(6);
        // ---------------- counters; the violating proposal itself (counted, acc bumped, then error(...), :120-124)
        if (Rc > 0u) {
            const uint32_t costL = (uint32_t)__builtin_amdgcn_readlane((int)cost, (int)(Rc - 1u));
            const uint32_t offL = (uint32_t)__builtin_amdgcn_readlane((int)off, (int)(Rc - 1u));
            dnum += Rc;
            idle = 0;
            dnacc += (uint32_t)__popcll(acc_c);
            dnm += offL + costL;
            t_last = l_readlane(tp, (int)(Rc - 1u));
            if (acc_c) t_event = l_readlane(tp, 63 - __builtin_clzll(acc_c));
        }
        if (vsel >= 0) {  // (vsel == Rc: every earlier event is committed)
            const double tpv = l_readlane(tp, vsel);
            const uint32_t iv = (uint32_t)__builtin_amdgcn_readlane((int)i, vsel);
            if (lane == 0) *reinterpret_cast<double*>(lines + (size_t)(iv >> 1) * 128 + 16 * (iv & 1u) + 8) = tpv;
            dnum += 1;
            vnacc = 1;
            dnm += 1;  // its coin
            t_last = tpv;
            status = PDMP_CHAIN_BOUND_VIOLATED;
        }
        if (status != PDMP_CHAIN_OK) break;
        // ---------------- the next window: what is left of this one -- from the quantum of the last committed time to its end: `left` events read
        // and not committed -- and as many new quanta as the candidate lanes have room for, by the running means of what a new quantum brings
        {
            const int32_t Bend = F + meff;
            const int32_t newq = meff - prev_base;  // new quanta this window took in (fewer than asked for where the list was cut)
            if (newq >= 1) {
                const double inv = __builtin_amdgcn_rcp((double)newq);
                const double rawn = (double)Cc - 1.3 * prev_left, evn = (double)nev0 - prev_left;
                rqf = l_uniform(rqf + (((rawn > 0.0) ? rawn : 0.0) * inv - rqf) * (1.0 / 16.0));
                evq = l_uniform(evq + (((evn > 0.0) ? evn : 0.0) * inv - evq) * (1.0 / 16.0));
            }
            F = Fn;
            const double left = (double)(nev0 - (int)Rc);
            const double room = raw_target - 1.3 * left;
            double extra = (room > 0.0) ? floor(room * __builtin_amdgcn_rcp((rqf > 1.0) ? rqf : 1.0)) : 0.0;
            if (extra < 1.0 && left < 12.0) extra = 1.0;
            extra = (extra < 32.0) ? extra : 32.0;
            prev_base = Bend - F;
            prev_left = l_uniform(left);
            const int32_t mn = prev_base + (int32_t)extra;
            m = (mn < 1) ? 1 : ((mn > L_MMAX) ? L_MMAX : mn);
            // the scale follows the density of events: a quantum should hold about targetq of them
            since_rescale += 1;
            if (since_rescale >= 512u && (evq > 1.5 * targetq || evq < 0.6 * targetq)) {
                s = l_uniform(s * evq / targetq);
                inv_s = l_uniform(1.0 / s);
                evq = targetq;
                rqf = 1.5 * targetq;
                since_rescale = 0;
                rebuild();
                m = 1;
                prev_base = 0;
                prev_left = 0.0;
            }
        }
        L_ORDER();
    }

    if (PROF && P.dbg && chain == 0 && lane == 0) {
        for (int q = 0; q < 10; ++q) P.dbg[q] = (double)ph[q];
        P.dbg[10] = (double)ph_iters;
        P.dbg[11] = (double)ph_raw;
        P.dbg[12] = (double)ph_zone;
        P.dbg[13] = (double)ph_eval;
        P.dbg[14] = (double)ph_rounds;
        P.dbg[15] = (double)ph_guess;
    }
#undef LPHASE
    if (lane == 0) {
        hdr->c.t_last = t_last;
        hdr->t_event = t_event;
        hdr->c.num += dnum;
        hdr->c.nacc += dnacc + vnacc;
        hdr->c.ntrace = ntrace0 + dnacc;
        hdr->c.nevents += dnacc;
        hdr->c.ndraw_main = nm0 + dnm;
        hdr->c.status = status;
        hdr->tl_scale = s;
    }
}

template <bool PROF>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void zz_local_trackl_kernel(ZzRunParams P) {
    trackl_body<PROF>(P);
}

bool zz_trackl_supported(const ZzRunParams& p) {
    // the plain lattice with an even side (a coordinate's pair mate i ^ 1 is then its lattice neighbour in the same column), every pair in LDS
    return p.lattice_n >= 16 && p.lattice_n <= 128 && (p.lattice_n % 2) == 0 && p.d <= (int64_t)LL::NB2 * 2 && p.d >= 2048 && !p.adapt && p.c_chain == nullptr &&
           p.tb.gmu_t == nullptr && !p.track_two_sums && !p.has_refresh;
}

int launch_zz_local_trackl(const ZzRunParams& p, int64_t nchains, void* stream) {
    dim3 grid((unsigned)nchains), block(64);
    if (p.dbg) hipLaunchKernelGGL((zz_local_trackl_kernel<true>), grid, block, LL::BYTES, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((zz_local_trackl_kernel<false>), grid, block, LL::BYTES, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

// records + pairs (pdmp_trackp.hip's layout: what set_state builds and every reader of the state takes) -> lines + cold records, and back
__global__ __launch_bounds__(256) void zz_trackl_pack_kernel(const TrRecP* __restrict__ rec, const double2* __restrict__ kp, TrLine* __restrict__ lines,
                                                            TrCold* __restrict__ cold, int64_t d, int64_t dk, int64_t nchains) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;  // pair
    if (b >= dk / 2) return;
    for (int64_t ch = blockIdx.y; ch < nchains; ch += gridDim.y) {
        TrLine L;
        double* const Lw = reinterpret_cast<double*>(&L);
        for (int p = 0; p < 2; ++p) {
            const int64_t i = 2 * b + p;
            double key = L_INF, told = 0.0, th = 0.0, gq = 0.0, gd = 0.0, tg = 0.0, c = 0.0, c100 = 0.0;
            if (i < d) {
                const TrRecP* r = rec + ch * d + i;
                const double2 k2 = kp[ch * dk + i];
                key = k2.x;
                told = k2.y;
                th = r->th;
                gq = r->g;
                gd = r->gd;
                tg = r->tg;
                c = r->c;
                c100 = r->c100;
                TrCold cc;
                cc.x = r->x;
                cc.tx = r->tx;
                cc.I = r->I;
                cc.acc = r->acc;
                cold[ch * dk + i] = cc;
            }
            Lw[2 * p + 0] = key;
            Lw[2 * p + 1] = told;
            Lw[4 + 4 * p + 0] = th;
            Lw[4 + 4 * p + 1] = gq;
            Lw[4 + 4 * p + 2] = gd;
            Lw[4 + 4 * p + 3] = tg;
            Lw[12 + 2 * p + 0] = c;
            Lw[12 + 2 * p + 1] = c100;
        }
        lines[ch * (dk / 2) + b] = L;
    }
}
__global__ __launch_bounds__(256) void zz_trackl_unpack_kernel(const TrLine* __restrict__ lines, const TrCold* __restrict__ cold, TrRecP* __restrict__ rec,
                                                              double2* __restrict__ kp, int64_t d, int64_t dk, int64_t nchains) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= d) return;
    for (int64_t ch = blockIdx.y; ch < nchains; ch += gridDim.y) {
        const double* const Lw = reinterpret_cast<const double*>(lines + ch * (dk / 2) + (i >> 1));
        const int p = (int)(i & 1);
        const TrCold cc = cold[ch * dk + i];
        TrRecP* r = rec + ch * d + i;
        r->x = cc.x;
        r->th = Lw[4 + 4 * p + 0];
        r->tx = cc.tx;
        r->I = cc.I;
        r->g = Lw[4 + 4 * p + 1];
        r->gd = Lw[4 + 4 * p + 2];
        r->tg = Lw[4 + 4 * p + 3];
        r->acc = cc.acc;
        kp[ch * dk + i] = make_double2(Lw[2 * p + 0], Lw[2 * p + 1]);
    }
}
int launch_zz_trackl_pack(const void* rec, const void* kp, void* lines, void* cold, int64_t d, int64_t dk, int64_t nchains, void* stream) {
    const unsigned gy = (unsigned)((nchains < 1024) ? nchains : 1024);
    hipLaunchKernelGGL(zz_trackl_pack_kernel, dim3((unsigned)((dk / 2 + 255) / 256), gy), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const TrRecP*>(rec), reinterpret_cast<const double2*>(kp), reinterpret_cast<TrLine*>(lines),
                       reinterpret_cast<TrCold*>(cold), d, dk, nchains);
    return (int)hipGetLastError();
}
int launch_zz_trackl_unpack(const void* lines, const void* cold, void* rec, void* kp, int64_t d, int64_t dk, int64_t nchains, void* stream) {
    const unsigned gy = (unsigned)((nchains < 1024) ? nchains : 1024);
    hipLaunchKernelGGL(zz_trackl_unpack_kernel, dim3((unsigned)((d + 255) / 256), gy), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const TrLine*>(lines), reinterpret_cast<const TrCold*>(cold), reinterpret_cast<TrRecP*>(rec),
                       reinterpret_cast<double2*>(kp), d, dk, nchains);
    return (int)hipGetLastError();
}

}  // namespace pdmp
