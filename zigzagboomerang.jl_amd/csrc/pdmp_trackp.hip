// pdmp_trackp.hip -- zz_local_trackp_kernel: pdmp_trackx.hip's one-proposal-per-lane tracked-gradient kernel with ONE dirty line per rejected
// proposal.
//
// What bounds pdmp_trackx.hip is the memory system, and mostly its write-backs: a rejected proposal (82 %) reads its record line
// and its key block's line and dirties BOTH (bound + proposal time in the record, the new key in the key block); a dirty line costs about 1.8
// line reads.  Here a rejected proposal dirties one line:
//   * the queue's level 0 holds PAIRS (key, t_old = the time the coordinate's bound was last computed: its own last proposal or the last
//     re-basing of its sums, whichever came later), 8 coordinates per 128-byte line; the record is only READ by a proposal -- its bound is
//     re-derived, bit for bit, from what the record and the pair hold: a = c + (g + gd (t_old − tg)) θ, b = c/100 + θ gd (every re-bound happens
//     either at the coordinate's own proposal or when g, gd are re-based, so these are the stored values);
//   * 2048 blocks of 8 do not fit the LDS as doubles, so level 1 is a LOWER BOUND of each block's minimum: (min − base) rounded down to a float,
//     minus 8 ulp, with the argument's position in the three low bits, compared as integers.  Every block whose bound is within the threshold is
//     a candidate; its line is read anyway (the exposure test needs the block's second key), which yields the exact minimum -- candidates are
//     ranked by exact keys, those beyond the exact threshold are no events and just refresh their bound.  Bounds may go stale LOW (a block
//     minimum whose key rose: its neighbour re-bounds it), never high: lowering is an LDS atomic minimum, and no rescans exist any more.  The
//     record of a candidate is requested together with its line using the position bits; a position that turns out wrong ends the list there.
// The committed sequence and every float are those of the oracle's tracked evaluation (oracle/pdmp_oracle.c: spdmp_zigzag_tracked), bit for bit;
// against the moving evaluation: index-exact, floats to ~1e-13.  Ensembles started at t0 > 0 -- whose first proposals lie BEFORE t0, the
// reference's initial keys carry no t0 (src/sfact.jl:186) -- run here too: t_old is stored, not inferred from an order of times.  The final
// clocks (zz_track_unpack_kernel) take t_old where the record layout has the time of the last own proposal: the same maximum, because a
// re-basing of j's sums is an accepted event of some n ∈ G1[j], which moved all of S[n] ⊇ G1[j] and is counted there.
//
// Two instantiations.  LAT = true is the plain n x n lattice (config C3): G1[i] = {i − n, i − 1, i, i + 1, i + n} inside the grid is computed from
// the index, the zone test is a distance in (row, column).  LAT = false is ANY symmetric sparse Γ with |G1[i]| <= 8 (src/sfact.jl:170-179 takes
// any CSC pattern; test/maintest.jl:6-8 uses sprandn): G1[i] travels with the record line as eight 16-bit ids, Γ[G1[i], i] comes from a shared
// table, and the zone tests compare ids -- an accepted event m disturbs a later event r iff i_r ∈ G1[i_m] (r's sums change), or r is accepted
// too and G1[i_r] ∩ G1[i_m] ≠ ∅ (both write a common member); on the lattice these are the distances 1 and 2.  Everything else is shared.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

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
__device__ __forceinline__ uint32_t w_grp8_min_u32(uint32_t v) {
    uint32_t o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);
    v = (o < v) ? o : v;
    o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true);
    v = (o < v) ? o : v;
    o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, true);
    return (o < v) ? o : v;
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

__device__ __forceinline__ uint32_t w_wave_min_u32(uint32_t v) {  // (DPP: six steps on the vector unit instead of six ds_bpermute round trips)
    auto step = [](uint32_t x, auto ctrl) -> uint32_t {
        const uint32_t o = (uint32_t)__builtin_amdgcn_mov_dpp((int)x, decltype(ctrl)::value, 0xf, 0xf, true);
        return (o < x) ? o : x;
    };
    v = step(v, std::integral_constant<int, 0xB1>{});
    v = step(v, std::integral_constant<int, 0x4E>{});
    v = step(v, std::integral_constant<int, 0x141>{});
    v = step(v, std::integral_constant<int, 0x140>{});
    // (row_bcast: lanes that receive nothing read 0 and are not used: the result is lane 63's)
    v = step(v, std::integral_constant<int, 0x142>{});
    v = step(v, std::integral_constant<int, 0x143>{});
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
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

// level-1 entry: a lower bound of (key - tb) as the bit pattern of a non-negative float, 8 ulp below the (nearest-rounded: off by at most half
// an ulp) difference, with the argument's position in the 3 low bits (bit patterns of non-negative floats order like the floats)
__device__ __forceinline__ uint32_t p_enc(double key, double tb, uint32_t pos) {
    double dlt = key - tb;
    dlt = (dlt > 0.0) ? dlt : 0.0;
    const uint32_t b = __float_as_uint((float)dlt);
    return (((b >= 8u) ? b - 8u : 0u) & ~7u) | pos;
}
// the largest pattern a block with exact minimum <= tau can carry
__device__ __forceinline__ uint32_t p_thr(double tau, double tb) {
    double dlt = tau - tb;
    dlt = (dlt > 0.0) ? dlt : 0.0;
    return __float_as_uint((float)dlt) + 1u;  // (one ulp above the nearest float; +Inf stays above every finite pattern)
}
__device__ __forceinline__ double p_dec(uint32_t bits, double tb) {
    return w_below(tb + (double)__uint_as_float(bits & ~7u));  // (one ulp below the rounded sum: never above the exact one)
}
constexpr uint32_t P_INFBITS = 0x7f7ffff8u;  // patterns from here on: the block is empty (+Inf)
__device__ __forceinline__ uint32_t lbf_pos(const unsigned char* smem, uint32_t b) {  // position bits of block b's level-1 entry
    return reinterpret_cast<const uint32_t*>(smem)[b] & 7u;
}

}  // namespace

// LDS layout (bytes).  HW = the two-wave form (below): room for 64 candidates, the control words shared with the helper wave and the ring of draws.
template <bool HW, bool BIG = false>
struct WL {
    static_assert(!(HW && BIG), "the two-wave form keeps 2048 block bounds");
    static constexpr uint32_t NBLK = BIG ? 8192 : 2048;    // key blocks of 8 a chain may have: d <= 16384, or (BIG, round 5) d <= 65536
    static constexpr uint32_t XO = (NBLK - 2048) * 4;      // (everything behind the bounds moves up by what they take more)
    static constexpr uint32_t LB = 0;                      // [NBLK] u32 lower bounds of the block minima (+ argument position), key blocks of 8
    static constexpr uint32_t EX = XO + 8192;              // [64] f64 what event e exposes; before that the candidates' keys where the ranks need them exactly
    // (XO + 8704 .. SLB: the candidate / event slots of the forms before the record was pushed from lane to lane: free)
    static constexpr uint32_t SLB = XO + (HW ? 9792 : 9664);  // [64] u16 event blocks, rank order
    static constexpr uint32_t TB = XO + (HW ? 9920 : 9792);   // [64] u16 candidate blocks, compaction order
    static constexpr uint32_t ACL = XO + (HW ? 10048 : 9920); // [8] u16 the accepted events
    static constexpr uint32_t NB = XO + (HW ? 10144 : 10016); // (LAT = false) [8][8] u16 G1 of the accepted events, in event order
    static constexpr uint32_t CTL = 10272;                 // (HW) control words shared by the two waves (struct WCtl)
    static constexpr uint32_t HPF = 10336;                 // (HW) [64] u16 the helper wave's list of coordinates whose lines it requests
    static constexpr uint32_t RING = 10464;                // (HW) [W_NR] (u, log u): draw n of the launch at slot n % W_NR
    static constexpr int CMAX = HW ? 64 : 56;              // candidates per iteration (block-scan passes of 8)
    static constexpr uint32_t WIN = HW ? 256u : 128u;      // draws an iteration may consume (single wave: two per lane in registers)
    static constexpr uint32_t BYTES = HW ? RING + 512 * 16 : NB + 128;  // dynamic LDS of a chain
};
constexpr uint32_t W_NR = 512;  // (HW) ring slots
constexpr uint32_t W_BYTES_1W = WL<false>::BYTES, W_BYTES_BIG = WL<false, true>::BYTES;
constexpr uint32_t W_BYTES_HW = WL<true>::BYTES;  // 18 656 bytes: 8 chains per CU
constexpr int W_AMAX = 8;            // accepted events per iteration (one group each)
// steering of the selection threshold (measured: 1.15 / 0.8 / 3 -- pdmp_trackx.hip's -- is 3.5 % slower here, where every candidate's line is read)
#ifndef W_GROW
#define W_GROW 1.02
#define W_SHRINK 0.98
#define W_SLACK 5u
#endif
// ... of the two-wave form: the raw candidate count is steered towards a target, by a gain (A/B at 512 / 1024 chains: 48 at 0.3 is 3 % faster than
// the step rule above with 1.1 / 0.95 / 20, and than a target of 56)
#define W_TARGET_HW 48u
#define W_GAIN_HW 0.3
#define W_TARGET_1W 44u                // ... of the one-wave forms where they use a target (56 candidate lanes, a window of 128 draws)
#define W_TARGET_1W_MAX_PER_CU 12      // ... which they do up to three chains per SIMD (3072 chains on 256 CUs)
#define W_NHYP 8     // hypotheses of the accept chain's first guess, two-wave form (a draw is one LDS read)
#define W_NHYP_1W 4  // ... single-wave form (a draw is two ds_bpermute pairs; A/B at 2048 chains: 4 is 0.6 % faster than none, 8 is 2 % slower)
#define W_PF_AHEAD 1.0  // the helper wave requests the lines of every block within this many window lengths beyond the window (1, 2, 4 measured: 1)
static_assert(WL<false>::BYTES <= 10240, "16 chains per CU: 160 KB / 16");
static_assert(WL<false, true>::BYTES <= 40960, "d <= 65536: 4 chains per CU (one per SIMD), 160 KB / 4");
static_assert(W_BYTES_HW == WL<true>::RING + W_NR * 16 && W_BYTES_HW <= 20480, "8 chains per CU: 160 KB / 8");
struct WCtl {  // (HW) written by one wave, polled by the other: DS operations of a wave execute in order, so data written before a word is visible with it
    uint32_t filled;    // helper: draws [0, filled) of the launch are in the ring
    uint32_t consumed;  // main: draws [0, consumed) are used up (their slots may be overwritten)
    uint32_t exitf;     // main: the launch is over
    uint32_t pad;
    double pf_tau, pf_dt, pf_tb;  // main: end and length of the current window, base of the level-1 bounds (what the helper prefetches against)
};
// The control words are reached through an LDS-qualified pointer: a volatile access through a generic pointer compiles to FLAT loads and stores
// with system-scope bits, each followed by s_waitcnt vmcnt(0) -- the poll at the head of every iteration then waits for every global store of
// the commit before it (read off the ISA in round 5).  With the address space stated they are ds_read / ds_write under lgkmcnt alone.
typedef volatile __attribute__((address_space(3))) WCtl* WCtlPtr;
__device__ __forceinline__ WCtlPtr w_ctl(unsigned char* smem, uint32_t offset) {
    return (WCtlPtr)(__attribute__((address_space(3))) unsigned char*)(smem + offset);
}

namespace {
// is the 16-bit value v (given twice: v | v << 16) one of the eight halves of nb?
__device__ __forceinline__ bool nb_has(const uint4 nb, uint32_t v2) {
    const uint32_t x0 = nb.x ^ v2, x1 = nb.y ^ v2, x2 = nb.z ^ v2, x3 = nb.w ^ v2;
    const uint32_t z = ((x0 - 0x00010001u) & ~x0) | ((x1 - 0x00010001u) & ~x1) | ((x2 - 0x00010001u) & ~x2) | ((x3 - 0x00010001u) & ~x3);
    return (z & 0x80008000u) != 0u;  // (a half is zero: the test is exact for "any half", the ids are below 2^15)
}
// number of ids in nb: the 0xFFFF halves stand at the end, and a valid id (< 2^15) does not start with a one bit
__device__ __forceinline__ uint32_t nb_count(const uint4 nb) {
    const uint64_t hi = ((uint64_t)nb.w << 32) | nb.z, lo = ((uint64_t)nb.y << 32) | nb.x;
    const uint32_t ph = (~hi == 0ull) ? 4u : ((uint32_t)__builtin_clzll(~hi) >> 4);
    const uint32_t pl = (~lo == 0ull) ? 4u : ((uint32_t)__builtin_clzll(~lo) >> 4);
    return 8u - ((ph == 4u) ? 4u + pl : ph);
}
}  // namespace

// The helper wave of the two-wave form (HW): wave 1 of the chain's workgroup.  It computes nothing the result depends on beyond the chain's
// stream of uniforms, which it produces AHEAD of the main wave -- draw n of the launch and its logarithm into slot n % W_NR of a ring in LDS, 64
// per pass, as far as the main wave's published consumption leaves room -- and it requests the lines the main wave's NEXT windows will read
// (every block whose level-1 bound lies within W_PF_AHEAD window lengths beyond the current window: its key line and the record its position bits
// point at), so that they are in the L2 when the main wave asks.  A request is a load whose value is thrown away; what the helper reads of
// level 1 may be mid-update -- a wrong guess costs a line, never a result.  (Until late in round 5 it also requested the four lattice neighbours'
// records of every such coordinate -- used by the 18 % that are accepted: 3.5 x the algorithmic bytes, 5.75 TB/s at 1024 chains, where leaving them
// out is worth 8 % and lets the two-wave form win up to seven chains per CU (1792: pdmp_capi.hip).)
template <bool LAT>
__device__ __forceinline__ void trackp_helper(const ZzRunParams& P, unsigned char* smem, const int lane, const int64_t chain, const uint64_t seed,
                                           const uint64_t nm0) {
    using L = WL<true, false>;
    const WCtlPtr ctl = w_ctl(smem, L::CTL);
    double2* const ring = reinterpret_cast<double2*>(smem + L::RING);
    const uint4* const lb4 = reinterpret_cast<const uint4*>(smem + L::LB);
    const int64_t d = P.d;
    const char* const recb = reinterpret_cast<const char*>(reinterpret_cast<const TrRecP*>(P.rec) + chain * d);
    const char* const kpb = reinterpret_cast<const char*>(reinterpret_cast<const double2*>(P.keys) + chain * P.dk);
    uint16_t* const HPF = reinterpret_cast<uint16_t*>(smem + L::HPF);
    uint32_t filled = 0, sink = 0;
    double pf_done = -W_INF;  // blocks with bounds up to here have been requested
    for (;;) {
        if (ctl->exitf != 0u) break;
        bool did = false;
        const uint32_t cons = ctl->consumed;
        while (filled + 64u <= cons + W_NR) {  // the ring comes first: as far as the main wave's consumption leaves room
            const uint32_t n = filled + (uint32_t)lane;
            const double u = pdmp_u01(seed, PDMP_STREAM_MAIN, nm0 + (uint64_t)n);
            ring[n & (W_NR - 1u)] = make_double2(u, pdmp_log(u));
            W_ORDER();
            filled += 64u;
            if (lane == 0) ctl->filled = filled;
            did = true;
        }
        const double tau = ctl->pf_tau, dts = ctl->pf_dt, tbs = ctl->pf_tb;
        const double target = tau + P.hw_ahead * dts;
        if (target > pf_done) {
            // every block whose bound lies in (pf_done, target]: compacted into a list, one coordinate per lane, all its lines requested at once
            // (hi stays below the patterns of empty blocks: a target at +Inf -- a window doubled through many idle iterations, a large hw_ahead --
            // must not select the padding blocks behind the chain's last coordinate, whose lines do not exist)
            const uint32_t lo = (pf_done > tbs) ? p_thr(pf_done, tbs) : 0u;
            const uint32_t hi0 = p_thr(target, tbs), hi = (hi0 < P_INFBITS) ? hi0 : P_INFBITS - 1u;
            uint32_t cm = 0;
#pragma unroll
            for (int j = 7; j >= 0; --j) {
                const uint4 v = lb4[lane + 64 * j];
                cm = (cm << 4) | ((v.x > lo && v.x <= hi) ? 1u : 0u) | ((v.y > lo && v.y <= hi) ? 2u : 0u) | ((v.z > lo && v.z <= hi) ? 4u : 0u) |
                     ((v.w > lo && v.w <= hi) ? 8u : 0u);
            }
            const uint32_t ncl = (uint32_t)__builtin_popcount(cm);
            const uint32_t incl = w_scan_add_u32(ncl);
            const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            uint32_t ix = incl - ncl, m_ = cm;
            while (m_ != 0u) {
                const uint32_t j = (uint32_t)(__ffs((int)m_) - 1);
                const uint32_t b = 4u * ((uint32_t)lane + 64u * (j >> 2)) + (j & 3u);
                if (ix < 64u) HPF[ix] = (uint16_t)(b * 8u + (lbf_pos(smem, b)));
                ix += 1;
                m_ &= m_ - 1u;
            }
            W_ORDER();
            if ((uint32_t)lane < tot && (uint32_t)HPF[lane] < (uint32_t)d) {
                const uint32_t i = (uint32_t)HPF[lane];
                const char* const kl = kpb + (size_t)(i >> 3) * 128;
                const char* const rl = recb + (size_t)i * 128;
                uint32_t a0 = *reinterpret_cast<const uint32_t*>(kl), a1 = *reinterpret_cast<const uint32_t*>(kl + 64);
                uint32_t a2 = *reinterpret_cast<const uint32_t*>(rl), a3 = *reinterpret_cast<const uint32_t*>(rl + 64);
                sink ^= a0 ^ a1 ^ a2 ^ a3;
            }
            asm volatile("" ::"v"(sink));
            pf_done = target;
            did = true;
        }
        if (!did) __builtin_amdgcn_s_sleep(4);
    }
    asm volatile("" ::"v"(sink));
}

template <bool PROF, bool LAT, bool HW, bool BIG>
__device__ __forceinline__ void trackp_body(const ZzRunParams& P) {
    using L = WL<HW, BIG>;
    constexpr uint32_t W_NBLK = L::NBLK;
    constexpr int NCH = (int)(W_NBLK / 2048u);  // chunks of 2048 block bounds: 32 per lane each
    constexpr int AMAXT = W_AMAX;  // accepted events per iteration: one pass of 8 groups (a second pass for events 8 .. 15 was built and measured in
    // round 5 on the two-wave form: the iteration's cost grows with the window as fast as what it commits -- 17.8 against 17.5 ms at 512 chains)
    constexpr int W_CMAX = L::CMAX;
    constexpr uint32_t W_WIN = L::WIN;
    const int lane = threadIdx.x & 63;
    const int g = lane >> 3, gl = lane & 7;
    const int64_t chain = blockIdx.x;
    const int64_t d = P.d;
    const uint32_t nblk = P.nblk;
    const uint32_t nlat = (uint32_t)P.lattice_n, nmagic = P.lattice_magic;

    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t* const lbf = reinterpret_cast<uint32_t*>(smem + L::LB);
    double* const EX = reinterpret_cast<double*>(smem + L::EX);
    double* const KM = EX;  // (exact block minima of the candidates, until the events are set up)
    uint16_t* const SLB = reinterpret_cast<uint16_t*>(smem + L::SLB);
    uint16_t* const TB = reinterpret_cast<uint16_t*>(smem + L::TB);
    uint16_t* const ACL = reinterpret_cast<uint16_t*>(smem + L::ACL);
    uint4* const NB4 = reinterpret_cast<uint4*>(smem + L::NB);        // (LAT = false only)
    uint16_t* const NB16 = reinterpret_cast<uint16_t*>(smem + L::NB);
    const WCtlPtr ctl = w_ctl(smem, L::CTL);                                                   // (HW only)
    const double2* const ring = reinterpret_cast<const double2*>(smem + L::RING);              // (HW only)

    TrRecP* const rec = reinterpret_cast<TrRecP*>(P.rec) + chain * d;
    double2* const kp = reinterpret_cast<double2*>(P.keys) + chain * P.dk;  // (key, t_old) per coordinate
    DevChain* const hdr = P.hdr + chain;
    pdmp_event* const evout = P.ev ? P.ev + chain * P.trace_cap : nullptr;

    uint32_t status = hdr->c.status;
    if (status == PDMP_CHAIN_BOUND_VIOLATED || status == PDMP_CHAIN_STALLED) return;  // (both waves)
    const uint64_t seed = hdr->seed;
    const uint64_t nm0 = hdr->c.ndraw_main, ntrace0 = hdr->c.ntrace;
    if (HW) {
        // the two waves part here: the control words are set before either polls them (the one barrier of the kernel; the header is read by
        // both waves above and written by the main wave at the very end)
        if (threadIdx.x == 0) {
            ctl->filled = 0u;
            ctl->consumed = 0u;
            ctl->exitf = 0u;
            ctl->pf_tau = -W_INF;
            ctl->pf_dt = 0.0;
            ctl->pf_tb = 0.0;
        }
        __syncthreads();
        if (threadIdx.x >= 64) {
            trackp_helper<LAT>(P, smem, lane, chain, seed, nm0);
            return;
        }
    }
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

    double seldt = 1e-3;  // (wave-uniform) the selection threshold above the queue's minimum
    // base of the level-1 bounds: below every key (the initial keys of the reference carry no t0, src/sfact.jl:186)
    double tb;  // (wave-uniform; moves up with the front)
    {
        double mloc = t_last;
        for (uint32_t b = lane; b < nblk; b += 64) {
            const double2* p = kp + (size_t)b * 8;
#pragma unroll
            for (int q = 0; q < 8; ++q) mloc = w_min(mloc, p[q].x);
        }
        tb = w_wave_min(mloc);
    }
    for (uint32_t b = lane; b < W_NBLK; b += 64) {
        uint32_t e = P_INFBITS;
        if (b < nblk) {
            const double2* p = kp + (size_t)b * 8;
            double mk = p[0].x;
            uint32_t mi = 0;
#pragma unroll
            for (int q = 1; q < 8; ++q) {
                const double v = p[q].x;
                if (v < mk) {
                    mk = v;
                    mi = q;
                }
            }
            e = (mk < W_INF) ? p_enc(mk, tb, mi) : P_INFBITS;
        }
        lbf[b] = e;
    }
    W_ORDER();

    uint64_t ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t ph_t0 = PROF ? (uint64_t)__builtin_readcyclecounter() : 0;
    uint64_t ph_rounds = 0, ph_guess = 0;
    uint64_t ph_iters = 0, ph_raw = 0, ph_zone = 0, ph_eval = 0;  // candidates: selected, after the zone cut, after the window and accept limits
#ifdef PDMP_PHASE_MARKS  // (ISA reading: a comment line per phase boundary in the assembly of the non-profiling instantiations)
#define WMARK(k) asm volatile("; WPHASE " #k)
#else
#define WMARK(k)
#endif
#define WPHASE(k)                                                         \
    do {                                                                  \
        WMARK(k);                                                         \
        if (PROF) {                                                       \
            const uint64_t now_ = (uint64_t)__builtin_readcyclecounter(); \
            ph[k] += now_ - ph_t0;                                        \
            ph_t0 = now_;                                                 \
        }                                                                 \
    } while (0)

    PrioTurn prio;
    bool need_rebase = false;
    uint32_t idle = 0;  // consecutive iterations without an event
    bool running = stop_before || (t_event < T);
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
        // ---------------- ring of uniforms: draws dnm .. dnm + 127, two per lane (HW: dnm .. dnm + 255 from the helper wave's ring in LDS)
        if (HW) {
            if (lane == 0) ctl->consumed = dnm;
            while (ctl->filled < dnm + W_WIN) __builtin_amdgcn_s_sleep(1);
            W_ORDER();
        } else {
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
        auto draw = [&](uint32_t n) -> double {  // draw nm0 + n for dnm <= n < dnm + W_WIN (every lane calls it: ds_bpermute)
            if (HW) return ring[n & (W_NR - 1u)].x;
            const double v0 = w_shfl(ureg[0], n & 63u), v1 = w_shfl(ureg[1], n & 63u);
            return ((n >> 6) & 1u) ? v1 : v0;
        };
        auto drawlog = [&](uint32_t n) -> double {  // its logarithm (HW: computed once, by the helper wave, with the same pdmp_log)
            if (HW) return ring[n & (W_NR - 1u)].y;
            return pdmp_log(draw(n));
        };
        // ---------------- select: every block whose lower bound is within the threshold, at most W_CMAX of them
        int C = 0;
        bool stalled = false, finished = false;
        double dt_used = 0.0;
        uint32_t Cc = 0;
        double tau = 0.0, mql_sel = 0.0;  // the window's end and the lower bound of the next event time it starts from
        bool tau_clipped = false;
        {
            // lane's entries: in chunk c (2048 block bounds), blocks 2048 c + 4 (lane + 64 j) + 0..3, j < 8 (one 16-byte read each).  One chunk
            // (d <= 16384): the 32 entries stay in registers through the three passes below; four (d <= 65536): every pass reads them again
            uint32_t kk[32];
            auto loadc = [&](int c) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint4 v = reinterpret_cast<const uint4*>(lbf)[512 * c + lane + 64 * j];
                    kk[4 * j + 0] = v.x;
                    kk[4 * j + 1] = v.y;
                    kk[4 * j + 2] = v.z;
                    kk[4 * j + 3] = v.w;
                }
            };
            auto lanemin = [&]() -> uint32_t {
                uint32_t m_ = kk[0];
#pragma unroll
                for (int j = 1; j < 32; ++j) m_ = (kk[j] < m_) ? kk[j] : m_;
                return m_;
            };
            uint32_t mloc = 0xffffffffu;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                loadc(c);
                const uint32_t m_ = lanemin();
                mloc = (m_ < mloc) ? m_ : mloc;
            }
            uint32_t mqb = w_wave_min_u32(mloc);
            double mql = p_dec(mqb, tb);  // a lower bound of the next event time
            if (mqb < P_INFBITS && (need_rebase || mql - tb > 0.25)) {
                // move the base of the bounds up to the front (a float resolves 2^-24 of its distance from the base)
                mloc = 0xffffffffu;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (NCH > 1) loadc(c);
#pragma unroll
                    for (int j = 0; j < 32; ++j)
                        kk[j] = (kk[j] >= P_INFBITS) ? P_INFBITS : p_enc(p_dec(kk[j], tb), mql, kk[j] & 7u);
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        reinterpret_cast<uint4*>(lbf)[512 * c + lane + 64 * j] = make_uint4(kk[4 * j + 0], kk[4 * j + 1], kk[4 * j + 2], kk[4 * j + 3]);
                    const uint32_t m_ = lanemin();
                    mloc = (m_ < mloc) ? m_ : mloc;
                }
                tb = mql;
                need_rebase = false;
                mqb = w_wave_min_u32(mloc);
                mql = p_dec(mqb, tb);
                W_ORDER();
            }
            if (mqb >= P_INFBITS) {
                stalled = true;
            } else if (stop_before && !(mql < T)) {
                finished = true;
            } else {
                double dt_sel = seldt;
                uint32_t cm[NCH], ncl = 0, incl = 0;
                for (int tries = 0;; ++tries) {
                    tau = mql + dt_sel;
                    tau_clipped = false;
                    if (stop_before && !(tau < T)) {
                        tau = w_below(T);
                        tau_clipped = true;
                    }
                    if (tries >= 64) tau = mql;
                    // (below the patterns of empty blocks whatever the window has grown to through idle iterations: the blocks behind the chain's
                    // last coordinate hold P_INFBITS and have no lines)
                    const uint32_t thr0 = p_thr(tau, tb), thr = (thr0 < P_INFBITS) ? thr0 : P_INFBITS - 1u;
                    ncl = 0;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        if (NCH > 1) loadc(c);
                        uint32_t m_ = 0;
#pragma unroll
                        for (int j = 31; j >= 0; --j) m_ = m_ + m_ + ((kk[j] <= thr) ? 1u : 0u);
                        cm[c] = m_;
                        ncl += (uint32_t)__builtin_popcount(m_);
                    }
                    incl = w_scan_add_u32(ncl);
                    Cc = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    if (Cc <= (uint32_t)W_CMAX || tries >= 64) break;
                    // (HW: shrink to what the count asks for, not by half -- the 64 candidate lanes are the resource the threshold is steered by)
                    dt_sel *= HW ? (double)(W_CMAX - 6) / (double)Cc : 0.5;
                }
                {
                    // (more than W_CMAX blocks inside the narrowest threshold: the first W_CMAX are looked at, see below)
                    uint32_t ix = incl - ncl;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        uint32_t m_ = cm[c];
                        while (__ballot(m_ != 0u) != 0) {
                            if (m_ != 0u) {
                                const uint32_t j = (uint32_t)(__ffs((int)m_) - 1);
                                if (ix < 64u) TB[ix] = (uint16_t)(2048u * (uint32_t)c + 4u * ((uint32_t)lane + 64u * (j >> 2)) + (j & 3u));
                                ix += 1;
                                m_ &= m_ - 1u;
                            }
                        }
                    }
                }
                dt_used = dt_sel;
                mql_sel = mql;
            }
        }
        if (stalled) {
            status = PDMP_CHAIN_STALLED;
            break;
        }
        if (finished) break;
        if (HW && lane == 0) {  // what the helper wave prefetches against
            ctl->pf_tau = tau;
            ctl->pf_dt = dt_used;
            ctl->pf_tb = tb;
        }
        W_ORDER();
        WPHASE(8);
        const bool crowded = Cc > (uint32_t)W_CMAX;  // exact ties beyond W_CMAX blocks (keys tied by construction): handled one event at a time
        if (crowded) Cc = (uint32_t)W_CMAX;
        // ---------------- candidate lane c: its block's line -- the 8 (key, time) pairs, eight 16-byte loads of ITS lane (the lines are distinct, the
        // request count is that of a group of 8 lanes reading one pair each) -- and the record the position bits point at, requested together
        const bool isc = (uint32_t)lane < Cc;
        const uint32_t cblk = isc ? (uint32_t)TB[lane] : 0u;
        double2 q8[8];
        {
            const double2* const bl = kp + (size_t)cblk * 8;
#pragma unroll
            for (int q = 0; q < 8; ++q) q8[q] = isc ? bl[q] : make_double2(W_INF, 0.0);
        }
        const uint32_t cpos = isc ? (lbf[cblk] & 7u) : 0u;
        const uint32_t ci = cblk * 8u + cpos;
        const TrRecP* const rci = rec + ci;
        const double c_th = rci->th, c_g = rci->g, c_gd = rci->gd, c_tg = rci->tg;
        const double2 c_c2 = *reinterpret_cast<const double2*>(&rci->c);  // (same line: no table in the event loop)
        // Γ[:,i]·μ of the flow (src/fact_samplers.jl:51: a = c + (Γ[:,i]·x − Γ[:,i]·μ) θ_i), kept in the record line's spare word where a mean is set
        // (round 6: until then a flow mean without a target mean was silently ignored here); track_mean = 2: the target has the same mean and the
        // rate subtracts it too (∇ϕ_i = Γ[:,i]·x − (Γμ)_i)
        const int tmean = P.track_mean;  // (wave-uniform)
        double c_gmu = 0.0;
        if (tmean) c_gmu = rci->tacc;
        uint4 c_nb = make_uint4(0u, 0u, 0u, 0u);
        if (!LAT) c_nb = *reinterpret_cast<const uint4*>(&rci->gam0);  // G1[ci]: eight 16-bit ids
        // the exact minimum of the block, its position (the lowest on ties) and time, the minimum of the rest and its position: in the lane's own
        // registers, one pass for all candidates instead of one per 8 of them
        double c_km = W_INF, c_rs = W_INF, c_tp = 0.0;
        uint32_t c_pb = 0u;
        {
            double m1 = q8[0].x, t1 = q8[0].y;
            uint32_t p1 = 0u;
#pragma unroll
            for (int q = 1; q < 8; ++q) {
                const bool lt = q8[q].x < m1;
                m1 = lt ? q8[q].x : m1;
                t1 = lt ? q8[q].y : t1;
                p1 = lt ? (uint32_t)q : p1;
            }
            double m2 = W_INF;
#pragma unroll
            for (int q = 0; q < 8; ++q) m2 = w_min(m2, ((uint32_t)q == p1) ? W_INF : q8[q].x);
            uint32_t p2 = 0u;
#pragma unroll
            for (int q = 7; q >= 0; --q) p2 = ((((uint32_t)q == p1) ? W_INF : q8[q].x) == m2) ? (uint32_t)q : p2;  // (the lowest position that holds it)
            if (isc) {
                c_km = m1;
                c_rs = m2;
                c_tp = t1;
                c_pb = p1 | (p2 << 4);
            }
        }
        // ---------------- events = candidates whose exact minimum is within the threshold; everybody refreshes its bound
        if (isc) lbf[cblk] = (c_km < W_INF) ? p_enc(c_km, tb, c_pb & 7u) : P_INFBITS;
        bool isev = isc && c_km <= tau;
        if (crowded) {
            // the narrowest threshold still holds more than W_CMAX blocks (their bounds fall into one float bucket): no event this iteration --
            // the looked-at blocks get exact bounds against a base moved up to the front, which separates them; blocks not looked at come next
            isev = false;
            need_rebase = true;
        }
        const double own = isev ? c_km : W_INF;
        // rank = the number of events with a smaller key.  First on 15-bit images of the keys -- (key − front) scaled so that the window maps onto
        // 0 .. 32766, a monotone map: distinct images order like their keys -- two candidates per packed instruction (a 16-bit difference, its sign
        // bit, a 16-bit add) from one 128-byte table that every lane reads whole; candidates that are no events hold 32767.  Two events with one
        // image give equal ranks, the ranks' sum falls short of 0 + 1 + .. + (nev − 1), and the exact comparison below takes over (about one
        // iteration in fifty at 48 candidates); it is also what tells exactly tied keys.
        const uint64_t evb = __ballot(isev);
        int nev = __popcll(evb);
        uint32_t rank = 0, rsum = 0xffffffffu;
        if (tau > mql_sel) {
            uint16_t* const QK = TB;  // (the candidates' blocks are in registers by now)
            const double scale = 32766.0 * __builtin_amdgcn_rcp(tau - mql_sel);
            const double img = (own - mql_sel) * scale;
            const uint32_t qi = isev ? (uint32_t)w_pos(img) : 32767u;
            const uint32_t qk = isev ? ((qi < 32766u) ? qi : 32766u) : 32767u;
            QK[lane] = (uint16_t)qk;
            W_ORDER();
            typedef short pk16 __attribute__((ext_vector_type(2)));
            typedef unsigned short upk16 __attribute__((ext_vector_type(2)));
            const pk16 own2 = {(short)qk, (short)qk};
            upk16 cnt = {0, 0};
            const uint4* const QK4 = reinterpret_cast<const uint4*>(QK);
#pragma unroll
            for (uint32_t h = 0; h < 2u; ++h) {
                if (h == 0u || Cc > 32u) {  // (wave-uniform; 32 candidates per batch of four 16-byte reads issued together: slots past the candidates hold 32767)
                    const uint4 w0 = QK4[4u * h + 0u], w1 = QK4[4u * h + 1u], w2 = QK4[4u * h + 2u], w3 = QK4[4u * h + 3u];
                    const uint32_t ww[16] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z, w3.w};
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const pk16 k2 = __builtin_bit_cast(pk16, ww[q]);
                        const pk16 df = k2 - own2;  // (< 0 where the candidate's image is smaller: both are below 2^15)
                        cnt += __builtin_bit_cast(upk16, df) >> (unsigned short)15;
                    }
                }
            }
            rank = (uint32_t)cnt.x + (uint32_t)cnt.y;
            W_ORDER();
            rsum = (uint32_t)__builtin_amdgcn_readlane((int)w_scan_add_u32(isev ? rank : 0u), 63);
        }
        if (rsum != (uint32_t)(nev * (nev - 1) / 2)) {
            // every lane compares its key with all of them, read from LDS one after the other (one address for the wave: a broadcast)
            KM[lane] = own;  // (KM is free here: the event slots are written into it only after the ranks are known)
            W_ORDER();
            rank = 0;
            for (uint32_t m0 = 0; m0 < Cc; m0 += 8) {  // (lanes past the candidates hold +Inf: reading them changes nothing)
                double km[8];
#pragma unroll
                for (uint32_t q = 0; q < 8; ++q) km[q] = KM[m0 + q];
#pragma unroll
                for (uint32_t q = 0; q < 8; ++q) rank += (km[q] < own) ? 1u : 0u;
            }
            W_ORDER();
            rsum = (uint32_t)__builtin_amdgcn_readlane((int)w_scan_add_u32(isev ? rank : 0u), 63);
        }
        // exactly equal keys among the events (probability zero unless keys are tied by construction) give equal ranks -- ranks count the strictly
        // smaller keys, so their sum then falls short of 0 + 1 + .. + (nev − 1): one event this iteration, the tied minimum of the lowest block
        bool slot = isev;  // this lane's candidate takes event slot `rank`
        if (rsum != (uint32_t)(nev * (nev - 1) / 2)) {
            const double mn = w_wave_min(own);
            const uint32_t bsel = w_wave_min_u32((isev && own == mn) ? cblk : 0xffffffffu);
            slot = isev && cblk == bsel;  // (rank 0: nothing is smaller than the minimum)
            nev = 1;
        }
        // a candidate whose record was requested at the wrong position ends the list at its rank
        {
            const bool wrongpos = isev && (c_pb & 7u) != cpos;
            const uint32_t wr = w_wave_min_u32(wrongpos ? rank : 0xffffffffu);
            if (wr < (uint32_t)nev) nev = (int)wr;
        }
        // ---------------- lane r = event r: everything moves over from its candidate's lane, PUSHED to lane `rank` (ds_permute: one trip through the
        // LDS crossbar, no LDS memory; until late in round 5 the record went through slots in LDS -- a write, a wait and a read, two trips -- or,
        // in the one-wave form, was pulled by 14 ds_bpermute behind a table of source lanes).  A candidate that is no event of this iteration
        // parks its words on lane 63, which is an event's lane only when all 64 candidates are events -- and then nobody parks.
        const uint32_t pdst = ((slot && rank < (uint32_t)nev) ? rank : 63u) << 2;
        auto push32 = [&](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_ds_permute((int)pdst, (int)v); };
        auto push64 = [&](double v) -> double {
            const int lo = __builtin_amdgcn_ds_permute((int)pdst, __double2loint(v)), hi = __builtin_amdgcn_ds_permute((int)pdst, __double2hiint(v));
            return __hiloint2double(hi, lo);
        };
        const double e_km = push64(c_km), e_rs = push64(c_rs), e_tp = push64(c_tp);
        const double e_th = push64(c_th), e_g = push64(c_g), e_gd = push64(c_gd), e_tg = push64(c_tg), e_c = push64(c_c2.x), e_c100 = push64(c_c2.y);
        double e_gmu = 0.0;
        if (tmean) e_gmu = push64(c_gmu);
        const uint32_t e_bu = push32(cblk | (c_pb << 16));
        uint4 e_nb = make_uint4(~0u, ~0u, ~0u, ~0u);
        if (!LAT) e_nb = make_uint4(push32(c_nb.x), push32(c_nb.y), push32(c_nb.z), push32(c_nb.w));
        W_ORDER();
        WPHASE(9);
        C = nev;
        if (PROF) ph_iters += 1;
        if (PROF) ph_raw += (uint64_t)Cc;
        const int Craw = (int)Cc;
        // (the candidates' record loads are named here, before the first branch that can leave the iteration: hipcc's structured control flow has
        // edges from the idle path below to the loop's latch that no wave ever takes, and a load still in flight along one of them put
        // s_waitcnt vmcnt(0) at the loop's head and latch -- where it waits for the COMMIT'S STORES of every iteration: read off the ISA in round 5)
        asm volatile("" ::"v"(c_th), "v"(c_g), "v"(c_gd), "v"(c_tg), "v"(c_c2.x), "v"(c_c2.y), "v"(c_nb.x), "v"(c_gmu));
        if (C == 0) {
            // nothing to do in this window (stale bounds refreshed, a wrong position fixed, or no key before T)
            if (tau_clipped && !crowded && __ballot(isc && c_km <= tau) == 0) break;  // stop_before: every key is at or beyond T
            if (++idle > 4096u) {
                status = PDMP_CHAIN_STALLED;  // (more than W_CMAX exactly tied block minima: keys tied by construction)
                break;
            }
            seldt = w_uniform(dt_used * 2.0);
            continue;
        }
        bool ev = lane < C;
        const double tp = ev ? e_km : W_INF;  // the event time: the exact block minimum
        const double rest = ev ? e_rs : W_INF;
        const double tprop_i = ev ? e_tp : 0.0;
        const uint32_t blk = e_bu & 0xffffu, pbe = e_bu >> 16;
        const double th = e_th, g_i = e_g, gd_i = e_gd, tg_i = e_tg;
        const double2 c_i2 = make_double2(e_c, e_c100);
        const uint4 nb_i = LAT ? make_uint4(~0u, ~0u, ~0u, ~0u) : e_nb;
        const uint32_t i = ev ? (blk * 8u + (pbe & 7u)) : 0u;
        const uint32_t rarg = blk * 8u + (pbe >> 4);
        const double c_i = c_i2.x;
        W_ORDER();
        if (ev) SLB[lane] = (uint16_t)blk;
        W_ORDER();
        WPHASE(0);
        uint32_t rc_i = 0xffffu, k_i;
        if (LAT) {
            // lattice coordinates packed for the zone test: byte 0 = row, byte 1 = column
            const uint32_t col_i = __umulhi(i, nmagic);
            const uint32_t row_i = i - col_i * nlat;
            rc_i = ev ? (row_i | (col_i << 8)) : 0xffffu;
            // |G1[i]| on the lattice: the cell and its neighbours inside the grid
            k_i = 1u + (col_i > 0u ? 1u : 0u) + (row_i > 0u ? 1u : 0u) + (row_i + 1u < nlat ? 1u : 0u) + (col_i + 1u < nlat ? 1u : 0u);
        } else {
            // G1[i] came with the candidate's line
            k_i = nb_count(nb_i);
        }
        W_ORDER();
        // ---------------- rates from the tracked sums (src/sfact.jl:116-119 with g_i(t′) = g_i + gd_i (t′ − tg_i))
        const double g_now = g_i + gd_i * (tp - tg_i);
        const double gmu_i = e_gmu;
        const double l = w_pos(((tmean == 2) ? g_now - gmu_i : g_now) * th);
        // the bound in force (src/fact_samplers.jl:50-54), re-derived: it was computed at t_old (the coordinate's last proposal or the last
        // re-basing of its sums, whichever came later: stored with the key) from exactly these operands
        const double told_i = tprop_i;
        const double g_told = g_i + gd_i * (told_i - tg_i);
        const double a_i = c_i + (tmean ? g_told - gmu_i : g_told) * th;
        const double b_i = c_i2.y + th * gd_i;
        const double lbound = w_pos(a_i + b_i * (tp - told_i));
        // ---------------- accept chain: offsets and outcomes as a fix-point (every round settles the events up to the next change).  Event r's coin is
        // draw 2 r + the draws the accepted events before it took beyond two (k − 1 each, k <= 8): three ballots of the accept mask by the bits of
        // k − 1, counted below the lane (v_mbcnt) -- no prefix scan
        const uint32_t ex_i = k_i - 1u;
        uint32_t off = 2u * (uint32_t)lane;
        bool acc = false;
        {
            // A first guess from W_NHYP hypotheses at once: "j accepted events of the usual size before me" puts my coin at 2 r + j e (e = the graph's
            // most frequent k − 1: 4 inside the lattice).  The accept masks are walked on the scalar unit -- the first accepted event under
            // hypothesis 0, the next one after it under hypothesis 1, ... -- until the masks run out or an accepted event of another size ends the
            // walk.  The fix-point below starts from that guess and is what decides: from ANY start, round t leaves the first t events exact, and
            // it ends only when a round reproduces its own input, which only the sequential outcome does.
            const uint32_t etyp = P.typ_extra;
            const uint64_t pl1 = __ballot(ev && (ex_i & 1u)), pl2 = __ballot(ev && (ex_i & 2u)), pl4 = __ballot(ev && (ex_i & 4u));
            auto offsets = [&](uint64_t ab) -> uint32_t {  // 2 r + the extra draws of the accepted events before lane r: their k − 1, bit plane by bit plane
                const uint64_t a1 = ab & pl1, a2 = ab & pl2, a4 = ab & pl4;
                const uint32_t n1 = __builtin_amdgcn_mbcnt_hi((uint32_t)(a1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)a1, 0u));
                const uint32_t n2 = __builtin_amdgcn_mbcnt_hi((uint32_t)(a2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)a2, 0u));
                const uint32_t n4 = __builtin_amdgcn_mbcnt_hi((uint32_t)(a4 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)a4, 0u));
                return 2u * (uint32_t)lane + n1 + 2u * n2 + 4u * n4;
            };
            uint64_t accb = 0;
            const uint64_t tg0_ = PROF ? (uint64_t)__builtin_readcyclecounter() : 0;
            constexpr int NHYP = HW ? W_NHYP : W_NHYP_1W;
            if (NHYP > 0) {
                uint64_t hb[NHYP > 0 ? NHYP : 1];
                double uj[NHYP > 0 ? NHYP : 1];
#pragma unroll
                for (int j = 0; j < NHYP; ++j) {  // (the reads first, by every lane: one round trip)
                    const uint32_t oj = off + etyp * (uint32_t)j;
                    uj[j] = draw(dnm + ((oj < W_WIN - 1u) ? oj : W_WIN - 1u));
                }
#pragma unroll
                for (int j = 0; j < NHYP; ++j) {
                    const uint32_t oj = off + etyp * (uint32_t)j;
                    hb[j] = __ballot((int)ev & (int)(oj + 1u + k_i <= W_WIN) & (int)(uj[j] * lbound < l));
                }
                const uint64_t odd = __ballot(ev && ex_i != etyp);
                uint64_t live = ~0ull;  // the lanes after the last accepted event found; 0 once the walk has ended
#pragma unroll
                for (int j = 0; j < NHYP; ++j) {
                    const uint64_t rem = hb[j] & live;
                    const uint64_t low = rem & (0ull - rem);  // its lowest set bit (0 if none)
                    accb |= low;
                    live = (low & ~odd) ? ~(low | (low - 1ull)) : 0ull;
                }
                if (accb != 0ull) off = offsets(accb);
            }
            if (PROF) ph_guess += (uint64_t)__builtin_readcyclecounter() - tg0_;
            for (int round = 0; round < 72; ++round) {
                if (PROF) ph_rounds += 1;
                const bool inwin = ev && (off + 1u + k_i <= W_WIN);
                const double u = draw(dnm + ((off < W_WIN - 1u) ? off : W_WIN - 1u));
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
            const uint64_t outb = __ballot(ev && !(off + 1u + k_i <= W_WIN));
            if (outb) {
                const int cut = __ffsll((unsigned long long)outb) - 1;
                C = (cut < C) ? cut : C;
            }
        }
        WPHASE(1);  // (the accept chain)
        // at most AMAXT accepted events per iteration: the candidate list ends before the next one
        {
            uint64_t ab = __ballot(acc) & ((C < 64) ? ((1ull << C) - 1ull) : ~0ull);
            if (__popcll(ab) > AMAXT) {
                uint64_t m_ = ab;
                for (int q = 0; q < AMAXT; ++q) m_ &= m_ - 1;
                C = __ffsll((unsigned long long)m_) - 1;
            }
        }
        uint32_t rekey_by = 0xffffffffu;  // the first accepted later event that re-bounds this lane's coordinate (its key is then not this lane's to store)
        // ---------------- zones.  Only an ACCEPTED event m disturbs a later event r: within lattice distance 1 it changes r's sums (r's outcome
        // above is then garbage), at distance 2 the two share a neighbour, which matters only if r is accepted too.  A rejected event writes its
        // own (key, time) pair and nothing else.  The list ends at the first disturbed event (everything before it is unaffected).
        if (LAT && HW) {
            // every event lane looks at all accepted events at once: they (<= W_AMAX of them inside the list) put (row, column, event) into LDS,
            // eight words that every lane reads back -- no walk over the accepted events on the scalar unit (a readlane, a ballot and the
            // scalar moves between them per event: ~300 cycles each for a wave alone on its SIMD)
            uint32_t* const ZA = reinterpret_cast<uint32_t*>(smem + L::NB);  // (the ids' room of the LAT = false instantiation)
            const uint64_t ab = __ballot(acc) & ((C < 64) ? ((1ull << C) - 1ull) : ~0ull);
            if (lane < W_AMAX) ZA[lane] = 0x00fffefeu;  // (an empty slot: event 255 -- behind every lane, so it disturbs none, and a key it claims is stored by its lane all the same)
            W_ORDER();
            if (acc && lane < C)
                ZA[__builtin_amdgcn_mbcnt_hi((uint32_t)(ab >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ab, 0u))] = rc_i | ((uint32_t)lane << 16);
            W_ORDER();
            const uint4 z0 = reinterpret_cast<const uint4*>(ZA)[0], z1 = reinterpret_cast<const uint4*>(ZA)[1];
            const uint32_t zz[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
            bool conf = false;
#pragma unroll
            for (int q = 0; q < W_AMAX; ++q) {
                const uint32_t m = zz[q] >> 16;
                const uint32_t sad = __builtin_amdgcn_sad_u8(rc_i, zz[q] & 0xffffu, 0u);
                conf = conf || (((uint32_t)lane > m) && (sad <= 1u || (sad <= 2u && acc)));
                // an EARLIER rejected event next to accepted event m: m's group writes that coordinate's new key after it
                if (sad <= 1u && (uint32_t)lane < m && m < rekey_by) rekey_by = m;
            }
            const uint64_t cb = __ballot(conf) & ((C < 64) ? ((1ull << C) - 1ull) : ~0ull);
            if (cb) {
                const int c0 = __ffsll((unsigned long long)cb) - 1;
                C = (c0 < C) ? c0 : C;
            }
        } else if (LAT) {
            // (with several waves on the SIMD the walk over the accepted events costs less than eight tests per lane: A/B at 2048 chains, round 5)
            uint64_t confb = 0;
            uint64_t ab = __ballot(acc) & ((C < 64) ? ((1ull << C) - 1ull) : ~0ull);
            while (ab) {
                const int m = __ffsll((unsigned long long)ab) - 1;
                ab &= ab - 1;
                const uint32_t rcm = (uint32_t)__builtin_amdgcn_readlane((int)rc_i, m);
                const uint32_t sad = __builtin_amdgcn_sad_u8(rc_i, rcm, 0u);
                const uint64_t hit = __ballot(sad <= 1u || (sad <= 2u && acc));
                confb |= hit & (~0ull << (m + 1));
                // an EARLIER rejected event next to accepted event m: m's group writes that coordinate's new key after it
                if (sad <= 1u && lane < m && (uint32_t)m < rekey_by) rekey_by = (uint32_t)m;
            }
            const uint64_t cb = confb & ((C < 64) ? ((1ull << C) - 1ull) : ~0ull);
            if (cb) {
                const int c0 = __ffsll((unsigned long long)cb) - 1;
                C = (c0 < C) ? c0 : C;
            }
        } else {
            // the same two tests on ids.  The accepted events' G1 go to LDS in event order (slot s = the s-th accepted event; the accepted events
            // that survive the cuts below are a prefix of them, so the slots stay valid); lane (g, gl) of the wave holds member gl of slot g.
            const uint64_t ab0 = __ballot(acc) & ((C < 64) ? ((1ull << C) - 1ull) : ~0ull);
            const int nacc0 = __popcll(ab0);  // (<= W_AMAX)
            if (acc && lane < C) {
                const int sl = __popcll(ab0 & ((1ull << lane) - 1ull));
                NB4[sl] = nb_i;
                ACL[sl] = (uint16_t)lane;
            }
            W_ORDER();
            const uint32_t jt = (g < nacc0) ? (uint32_t)NB16[lane] : 0xffffu;  // (slot g, member gl: NB16[8 g + gl])
            const uint32_t jt2 = jt | (jt << 16), i2 = i | (i << 16);
            uint64_t confb = 0, ghit = 0;
            uint64_t ab = ab0;
            for (int sl = 0; sl < nacc0; ++sl) {
                const int m = __ffsll((unsigned long long)ab) - 1;
                ab &= ab - 1;
                const uint4 nbm = NB4[sl];  // (one address for the wave: a broadcast read)
                const bool d1 = ev && nb_has(nbm, i2);  // i_r ∈ G1[i_m]  (m itself included: masked below)
                confb |= __ballot(d1) & (~0ull << (m + 1));
                if (d1 && lane < m && (uint32_t)m < rekey_by) rekey_by = (uint32_t)m;
                // a LATER accepted event that shares a member with m
                ghit |= __ballot(g > sl && jt != 0xffffu && nb_has(nbm, jt2));
            }
            if (ghit) {
                const int gs = (__ffsll((unsigned long long)ghit) - 1) >> 3;
                const int r2 = (int)ACL[gs];
                confb |= 1ull << r2;
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
        uint64_t vb_adapt = 0;  // adapt = true (round 6): the accepted events that multiply their bound by `factor` and go on (adapt!(c, i, factor), :127, src/fact_samplers.jl:67-70)
        {
            const uint64_t vb = __ballot(violated0) & ((C < 64) ? ((1ull << C) - 1ull) : ~0ull);
            if (vb) {
                if (P.adapt) {
                    vb_adapt = vb;
                } else {
                    vsel = __ffsll((unsigned long long)vb) - 1;
                    C = vsel;  // the violating event itself is not committed
                }
            }
        }
        ev = lane < C;
        acc = acc && ev;
        if (PROF) ph_eval += (uint64_t)C;
        const uint64_t accball = __ballot(acc);
        const int nacc_it = __popcll(accball);
        if (LAT && !HW && acc) ACL[__popcll(accball & ((1ull << lane) - 1ull))] = (uint16_t)lane;  // (LAT = false: written with the zones, same slots; HW on the lattice: not read)
        W_ORDER();
        WPHASE(2);
        // ---------------- accepted events, one 8-lane group each: members of G1[i] (ascending, :131-135)
        // (the groups of the accepted events are the LAST groups of the wave, in event order: the low lanes -- lane r = event r -- are then free
        // to re-bound their rejected proposals in the same evaluation, see below)
        struct GOut {
            bool gact, mem, selfl, adapted;
            double c_new, c100_new;
            uint32_t ea, ia, blka, jm, cand_a;
            double tpa, gj, gdj, keyj, xa, txa, Ia, th_ia, rowmin_a;
            uint64_t acc_ia;
        };
        double key2 = W_INF;  // the new key of this lane's rejected proposal (event lanes)
        auto group_stage = [&](const int base, const int ng, const bool first) -> GOut {
            GOut o;
            const int g0 = 8 - ng;
            const bool gact = g >= g0;
            uint32_t ea = 0u;  // the event of this lane's group: the (base + g − g0)-th accepted one
            if (HW) {
                // (the positions of the accept mask's set bits, walked on the scalar unit: no round trip through LDS)
                uint64_t m_ = accball;
                const int want = base + g - g0;
#pragma unroll
                for (int n = 0; n < AMAXT; ++n) {
                    const uint32_t pos = m_ ? (uint32_t)(__ffsll((unsigned long long)m_) - 1) : 0u;
                    m_ &= m_ - 1ull;
                    ea = (want == n) ? pos : ea;
                }
            } else {
                ea = gact ? (uint32_t)ACL[g - g0] : 0u;
            }
            const uint32_t ia_b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ea << 2), (int)i);
            const uint32_t off_b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ea << 2), (int)off);
            const uint32_t ia = gact ? ia_b : 0u;
            const uint32_t blka = gact ? (uint32_t)SLB[ea] : 0u;
            const double tpa_b = w_shfl(tp, ea);
            const double tpa = gact ? tpa_b : 0.0;
            const uint32_t offa = gact ? off_b : 0u;
            uint32_t ka = 0, jm = ia;
            bool mem = false;
            if (LAT) {
                // G1[ia] on the lattice, ascending: {ia − n, ia − 1, ia, ia + 1, ia + n} inside the grid -- computed, so that the members' records
                // are requested at once; the CSC tables are read for the VALUES only (Γ[j, i] = Γ[i, j]: symmetric, checked on the host)
                const uint32_t cola = __umulhi(ia, nmagic), rowa = ia - cola * nlat;
                const bool hasL = cola > 0u, hasU = rowa > 0u, hasD = rowa + 1u < nlat, hasR = cola + 1u < nlat;
                ka = gact ? (1u + (hasL ? 1u : 0u) + (hasU ? 1u : 0u) + (hasD ? 1u : 0u) + (hasR ? 1u : 0u)) : 0u;
                mem = gact && (uint32_t)gl < ka;
                // position gl among the present members in the order L, U, self, D, R
                const uint32_t pos = (uint32_t)gl;
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
            } else {
                // member gl of G1[ia]: the ids the event's candidate lane read with its record (LDS slot g − g0), ascending (:131-135)
                const uint32_t jraw = gact ? (uint32_t)NB16[8 * (g - g0) + gl] : 0xffffu;
                mem = jraw != 0xffffu;
                jm = mem ? jraw : ia;
            }
            TrRecP* const rj = rec + jm;
            TrRecP* const ria = rec + ia;
            double gam = 0.0;  // Γ[jm, ia]: member gl of G1[ia]
            if (LAT) {
                if (mem) gam = (gl == 0) ? ria->gam0 : (gl == 1) ? ria->gam1 : (gl == 2) ? ria->gam2 : (gl == 3) ? ria->gam3 : ria->gam4;
            } else {
                if (mem) gam = P.tb.gam8[(size_t)ia * 8 + (size_t)gl];  // (shared table, L2: requested next to the members' records)
            }
            // (the reflecting coordinate's own fields are read again by its group: the lines are in L2)
            const double th_ia = ria->th;
            double xa = ria->x, txa = ria->tx, Ia = ria->I;
            const uint64_t acc_ia = ria->acc;
            const double thj0 = rj->th, gj0 = rj->g, gdj0 = rj->gd, tgj = rj->tg;
            double gmu_j = 0.0;
            if (tmean) gmu_j = rj->tacc;
            double2 cjm2 = *reinterpret_cast<const double2*>(&rj->c);
            asm volatile("" : "+v"(cjm2.x), "+v"(cjm2.y));  // (one 16-byte load with the others: hipcc sank the second half under the select below, a dependent round trip)
            const double resta_b = w_shfl(rest, ea);  // the accepted event's block without it, and where that minimum sits
            const uint32_t rarga_b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ea << 2), (int)rarg);
            // ---------------- ONE evaluation of the new bound and key per lane (logarithm, two divisions, square root): the re-bound of a
            // rejected proposal (:137-140) in its event lane (first pass), the re-bound of a member of G1 (:131-135) in its group lane.  A lane
            // that is both (more than 64 − 8 ng candidates) evaluates its rejected proposal again below.
            const bool selfl = mem && jm == ia;
            // a violated bound under adapt: c_i <- factor c_i before G1[i] is re-bounded (only i's own bound reads c_i)
            const bool adapted = selfl && ((vb_adapt >> ea) & 1ull) != 0ull;
            if (adapted) {
                cjm2.x = cjm2.x * P.factor;
                cjm2.y = cjm2.x / 100;
            }
            const double thj = selfl ? -th_ia : thj0;
            const double gj = gj0 + gdj0 * (tpa - tgj);
            const double gdj = gdj0 + gam * (-2.0 * th_ia);  // θ_i -> −θ_i
            double a2l, b2l, key2l;
            {
                const bool mine = gact || !first;  // (a later pass evaluates for its groups only)
                const uint32_t dix = mine ? (offa + 1u + (uint32_t)gl) : (off + 1u);
                const double Lg = drawlog(dnm + ((dix < W_WIN - 1u) ? dix : W_WIN - 1u));
                const double cc = mine ? cjm2.x : c_i, cc100 = mine ? cjm2.y : c_i2.y;
                const double gg0 = mine ? gj : g_now, tt = mine ? thj : th, gdd = mine ? gdj : gd_i;
                const double gg = tmean ? gg0 - (mine ? gmu_j : gmu_i) : gg0;
                a2l = cc + gg * tt;
                b2l = cc100 + tt * gdd;
                key2l = (mine ? tpa : tp) + w_poisson_time_L(a2l, b2l, Lg);
            }
            const double keyj = mem ? key2l : W_INF;
            if (first) {
                if (__ballot(ev && !acc && gact) != 0) {
                    const double Le = drawlog(dnm + ((off + 1u < W_WIN - 1u) ? off + 1u : W_WIN - 1u));
                    const double a2e = c_i + (tmean ? g_now - gmu_i : g_now) * th;
                    const double b2e = c_i2.y + th * gd_i;
                    const double k2e = tp + w_poisson_time_L(a2e, b2e, Le);
                    if (gact) key2l = k2e;
                }
                key2 = key2l;
            }
            if (selfl) {  // event(i, t, x, θ, F) (src/sfact.jl:50-52): x_i at t′
                const double dtx = tpa - txa;
                const double xn = xa + th_ia * dtx;
                Ia = Ia + dtx * ((xa + xn) * 0.5);
                xa = xn;
                txa = tpa;
            }
            // the accepted event's block: a LOWER BOUND of its new minimum is enough (level 1 holds bounds): the smaller of the block without the
            // event -- which may still count a member's OLD key: then the bound is stale low and costs a look later -- and the members' new keys in it
            const double kin = (mem && (jm >> 3) == blka) ? keyj : W_INF;
            const double kinmin = w_grp8_min(kin);
            // position bits of the member that holds it (the lowest lane of the group on ties): a DPP minimum of (lane in group, position)
            const uint32_t wkey = (gact && kin == kinmin) ? (((uint32_t)gl << 3) | (jm & 7u)) : 0xffu;
            const uint32_t jwin = w_grp8_min_u32(wkey);  // (no lane of an inactive group is read below)
            const bool restwins = resta_b <= kinmin;
            const double rowmin_a = restwins ? resta_b : kinmin;
            const uint32_t cand_a = restwins ? (rarga_b & 7u) : (jwin & 7u);
            const double keymin = w_grp8_min(keyj);
            if (gact && gl == 0) EX[ea] = w_min(rowmin_a, keymin);
            // (the reflecting coordinate's fields are used under lane masks only: named here, or the path around the commit carries their loads to
            // the loop's head, where the wait for them is a wait for the commit's stores)
            asm volatile("" ::"v"(xa), "v"(txa), "v"(Ia), "v"(acc_ia));
            o.gact = gact;
            o.mem = mem;
            o.selfl = selfl;
            o.adapted = adapted;
            o.c_new = cjm2.x;
            o.c100_new = cjm2.y;
            o.ea = ea;
            o.ia = ia;
            o.blka = blka;
            o.jm = jm;
            o.cand_a = cand_a;
            o.tpa = tpa;
            o.gj = gj;
            o.gdj = gdj;
            o.keyj = keyj;
            o.xa = xa;
            o.txa = txa;
            o.Ia = Ia;
            o.th_ia = th_ia;
            o.rowmin_a = rowmin_a;
            o.acc_ia = acc_ia;
            return o;
        };
        const GOut G0 = group_stage(0, nacc_it, true);
        // new minimum of the popped block of a rejected event, and what the event exposes
        double rowmin = W_INF;
        uint32_t cand = i;
        if (ev && !acc) {
            const bool mine = key2 < rest || (key2 == rest && i < rarg);
            rowmin = mine ? key2 : rest;
            cand = mine ? i : rarg;
        }
        if (ev && !acc) EX[lane] = rowmin;  // (rowmin <= key2: the new key is one of its candidates)
        W_ORDER();
        WPHASE(3);
        // ---------------- validate: all earlier events commit, zones disjoint, nothing produced or exposed earlier than t′
        uint32_t Rc;
        {
            const double prev = (lane > 0 && lane <= C) ? EX[lane - 1] : W_INF;  // what event lane − 1 exposes
            const double pref = w_scan_min_f64(prev);  // exclusive prefix minimum
            const bool okr = ev && (lane == 0 || pref > tp);  // (zone conflicts ended the candidate list already)
            const uint64_t bad = ~__ballot(okr);
            const uint32_t r_ok = bad ? (uint32_t)(__ffsll((unsigned long long)bad) - 1) : 64u;
            Rc = (r_ok < (uint32_t)C) ? r_ok : (uint32_t)C;
            if (vsel != (int)Rc) vsel = -1;  // the violating proposal counts only once everything before it is committed
            // the trace's room and the end of the run (`while t′ < T` looks at accepted events only, :199): the first accepted event that fills
            // the trace or reaches T ends the list behind it -- every event lane tests itself, two ballots (a walk over the accepted events on
            // the scalar unit cost a lone wave ~100 cycles per event)
            const uint64_t accc = accball & ((Rc < 64u) ? ((1ull << Rc) - 1ull) : ~0ull);
            const bool isacc_c = ((accc >> lane) & 1ull) != 0ull;
            const uint32_t na_l = __builtin_amdgcn_mbcnt_hi((uint32_t)(accc >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)accc, 0u)) + 1u;  // accepted events up to this one
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
        // steer the threshold so that the raw candidate list is just longer than what can commit
        if (HW || P.hw_target != 0u) {
            // the two-wave form (and the one-wave forms of an under-occupied launch: hw_target set by the launcher) aims the raw candidate count at a target -- a candidate read in vain costs an under-occupied device nothing, idle
            // candidate lanes do -- moving a fraction of the way per iteration
            const double want = (double)P.hw_target / (double)(Craw > 0 ? Craw : 1);
            seldt = w_uniform(dt_used * (1.0 + P.hw_gain * (((want < 2.0) ? want : 2.0) - 1.0)));
        } else {
            seldt = w_uniform(dt_used * (((int)Rc >= Craw) ? W_GROW : (((int)Rc + (int)W_SLACK < Craw) ? W_SHRINK : 1.0)));
        }
        WPHASE(4);
        // ---------------- commit the valid prefix
        const bool commit = ev && (uint32_t)lane < Rc;
        if (commit && !acc) {  // a rejected proposal: ONE 16-byte store (the record stays clean), and the block's new bound
            if (!(rekey_by < Rc)) kp[i] = make_double2(key2, tp);  // (else a later accepted neighbour of this iteration re-bounds i: its pair)
            lbf[blk] = (rowmin < W_INF) ? p_enc(rowmin, tb, cand & 7u) : P_INFBITS;
        }
        const uint64_t acc_c = accball & ((Rc < 64u) ? ((1ull << Rc) - 1ull) : ~0ull);
        auto commit_groups = [&](const GOut& o) {
            const bool gcommit = o.gact && o.ea < Rc;
            if (gcommit) {
                TrRecP* const rj = rec + o.jm;
                TrRecP* const ria = rec + o.ia;
                if (o.mem) {
                    rj->g = o.gj;
                    rj->gd = o.gdj;
                    rj->tg = o.tpa;
                    kp[o.jm] = make_double2(o.keyj, o.tpa);  // (the bound of every member is computed now)
                }
                if (o.selfl) {
                    ria->x = o.xa;
                    ria->th = -o.th_ia;
                    ria->tx = o.txa;
                    ria->I = o.Ia;
                    ria->acc = o.acc_ia + 1;
                    if (o.adapted) *reinterpret_cast<double2*>(&ria->c) = make_double2(o.c_new, o.c100_new);
                    // (the time of i's last accept IS its position's clock tx in this layout -- x_i is brought up on i's accepts only --: a store into
                    // the line's fourth sector, dirty for nothing else, went with it; zz_track_unpack_kernel reads tx)
                    if (evout) {
                        const uint32_t rnk = (uint32_t)__popcll(acc_c & ((1ull << o.ea) - 1ull));
                        pdmp_event e;
                        e.t = o.tpa;
                        e.i = (int64_t)o.ia;
                        e.x = o.xa;
                        e.theta = -o.th_ia;
                        evout[ntrace0 + dnacc + rnk] = e;
                    }
                }
                if (gl == 0) lbf[o.blka] = (o.rowmin_a < W_INF) ? p_enc(o.rowmin_a, tb, o.cand_a) : P_INFBITS;  // (lane 0 of the group stores the bound)
            }
            return gcommit;
        };
        const bool gc0 = commit_groups(G0);
        W_ORDER();
        WPHASE(5);
        // ---------------- bounds of the blocks of re-bounded neighbours: lowered where the new key is below them (an LDS atomic minimum); a key
        // that ROSE leaves its block's bound stale low, which costs a look at the block later and nothing else
        {
            if (gc0 && G0.mem && (G0.jm >> 3) != G0.blka && G0.keyj < W_INF) atomicMin(&lbf[G0.jm >> 3], p_enc(G0.keyj, tb, G0.jm & 7u));
        }
        W_ORDER();
        WPHASE(6);
        // ---------------- counters; the violating proposal itself (counted, acc bumped, then error(...), :120-124)
        if (Rc > 0u) {
            const uint32_t costL = (uint32_t)__builtin_amdgcn_readlane((int)cost, (int)(Rc - 1u));
            const uint32_t offL = (uint32_t)__builtin_amdgcn_readlane((int)off, (int)(Rc - 1u));
            dnum += Rc;
            idle = 0;
            dnacc += (uint32_t)__popcll(acc_c);
            dnm += offL + costL;
            t_last = w_readlane(tp, (int)(Rc - 1u));
            if (acc_c) t_event = w_readlane(tp, 63 - __builtin_clzll(acc_c));
        }
        if (vsel >= 0) {  // (vsel == Rc: every earlier event is committed)
            const double tpv = w_readlane(tp, vsel);
            const uint32_t iv = (uint32_t)__builtin_amdgcn_readlane((int)i, vsel);
            if (lane == 0) kp[iv].y = tpv;
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
        P.dbg[14] = (double)ph_rounds;
        P.dbg[15] = (double)ph_guess;
    }
#undef WPHASE
    if (HW && lane == 0) ctl->exitf = 1u;
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

template <bool PROF, bool LAT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void zz_local_trackp_kernel(ZzRunParams P) {
    trackp_body<PROF, LAT, false, false>(P);
}
// 16384 < d <= 65536 (the lattice up to 256 x 256; round 5): 8192 block bounds take 32 KB of LDS, so four chains share a CU -- one wave per SIMD,
// the register file to itself -- and the selection scans four chunks of 2048 bounds.  Same loop, same results.
template <bool PROF>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) void zz_local_trackp_big_kernel(ZzRunParams P) {
    trackp_body<PROF, true, false, true>(P);
}
// The two-wave form: one chain per WORKGROUP of two wavefronts -- the main wave runs the event loop above, the helper wave (trackp_helper)
// keeps the ring of draws filled and the next windows' lines on their way.  For ensembles that leave SIMDs idle (a rank's share of a
// strong-scaled job: 2048 / 1024 / 512 chains on 1024 SIMDs), where a chain's rate is set by ONE wave's dependent chain of instructions and
// round trips and nothing else runs beside it.  Same committed sequence, same floats (same draws, same pdmp_log, same arithmetic in the same
// lanes): only who computes a uniform and when a line is requested differ.
template <bool PROF, bool LAT>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 4))) void zz_local_trackp2_kernel(ZzRunParams P) {
    trackp_body<PROF, LAT, true, false>(P);
}

bool zz_trackp_supported(const ZzRunParams& p) {
    // the plain lattice (neighbours from the index), or any graph whose ids and values are tabulated (|G1| <= TRACKP_KMAX, checked by the host)
    // (the lattice up to 256 x 256: d <= 65536 on 8192 block bounds, four chains per CU; a graph's ids are 16 bits with 0xFFFF for "none" and
    // compared as halves below 2^15: d <= 16384)
    const bool lattice = p.lattice_n >= 16 && p.lattice_n <= 256 && p.d <= (int64_t)WL<false, true>::NBLK * 8;
    const bool graph = p.lattice_n == 0 && p.tb.nb16 != nullptr && p.tb.gam8 != nullptr && p.d <= (int64_t)WL<false>::NBLK * 8;
    // (adapt: the per-chain bounds are the c / c100 words of the record lines -- round 6)
    // (means: track_mean is set by the host -- 0 none, 1 the flow's only, 2 flow and target with the same Γμ; a target mean that differs from the flow's
    // keeps the 8-lane-group kernel: the host passes gmu_t then)
    return (lattice || graph) && p.tb.gmu_t == nullptr && !p.track_two_sums && !p.has_refresh && p.d >= 2048;
}

int launch_zz_local_trackp(const ZzRunParams& p, int64_t nchains, void* stream) {
    dim3 grid((unsigned)nchains), block(64);
    ZzRunParams q = p;
    q.nblk = (uint32_t)((p.d + 7) / 8);  // (dk is a multiple of 64, the padding keys are +Inf)
    const bool lat = p.lattice_n != 0;
    if (!p.helper_wave && !(q.hw_gain > 0.0)) {
        // one-wave forms: the step rule (W_GROW / W_SHRINK / W_SLACK: few candidates read in vain) where the launch fills the device and the memory
        // system is the other limit; a target count, as in the two-wave form, where it does not (at most three chains per SIMD, and always at
        // d > 16384, where LDS admits one chain per SIMD): a chain's rate is then set by how much one iteration commits.  A/B in one session,
        // round 5: 2048 chains 25.8 -> 22.0 ms, d = 65536 at 1024 chains 90.4 -> 80.4 ms; 4096 chains 46.6 -> 48.1 ms (so not there)
        if (nchains <= (int64_t)W_TARGET_1W_MAX_PER_CU * (p.n_cu > 0 ? p.n_cu : 256) || p.d > (int64_t)WL<false>::NBLK * 8) {
            q.hw_gain = W_GAIN_HW;
            q.hw_target = W_TARGET_1W;
        } else {
            q.hw_target = 0u;
        }
    }
    if (p.d > (int64_t)WL<false>::NBLK * 8) {  // 8192 block bounds in LDS (the lattice only), one wave per chain
        if (p.dbg) hipLaunchKernelGGL((zz_local_trackp_big_kernel<true>), grid, block, W_BYTES_BIG, (hipStream_t)stream, q);
        else hipLaunchKernelGGL((zz_local_trackp_big_kernel<false>), grid, block, W_BYTES_BIG, (hipStream_t)stream, q);
        return (int)hipGetLastError();
    }
    if (p.helper_wave) {
        dim3 block2(128);
        if (!(q.hw_gain > 0.0)) {  // (not set by pdmp_debug_set_helper_steering: the defaults)
            q.hw_gain = W_GAIN_HW;
            q.hw_target = W_TARGET_HW;
            q.hw_ahead = W_PF_AHEAD;
        }
        if (lat) {
            if (p.dbg) hipLaunchKernelGGL((zz_local_trackp2_kernel<true, true>), grid, block2, W_BYTES_HW, (hipStream_t)stream, q);
            else hipLaunchKernelGGL((zz_local_trackp2_kernel<false, true>), grid, block2, W_BYTES_HW, (hipStream_t)stream, q);
        } else {
            if (p.dbg) hipLaunchKernelGGL((zz_local_trackp2_kernel<true, false>), grid, block2, W_BYTES_HW, (hipStream_t)stream, q);
            else hipLaunchKernelGGL((zz_local_trackp2_kernel<false, false>), grid, block2, W_BYTES_HW, (hipStream_t)stream, q);
        }
        return (int)hipGetLastError();
    }
    if (lat) {
        if (p.dbg) hipLaunchKernelGGL((zz_local_trackp_kernel<true, true>), grid, block, W_BYTES_1W, (hipStream_t)stream, q);
        else hipLaunchKernelGGL((zz_local_trackp_kernel<false, true>), grid, block, W_BYTES_1W, (hipStream_t)stream, q);
    } else {
        if (p.dbg) hipLaunchKernelGGL((zz_local_trackp_kernel<true, false>), grid, block, W_BYTES_1W, (hipStream_t)stream, q);
        else hipLaunchKernelGGL((zz_local_trackp_kernel<false, false>), grid, block, W_BYTES_1W, (hipStream_t)stream, q);
    }
    return (int)hipGetLastError();
}

// the per-coordinate constants into the two free sectors of every record (after the init kernel)
__global__ __launch_bounds__(256) void zz_trackp_consts_kernel(TrRecP* __restrict__ rec, const CoordConst* __restrict__ cc,
                                                              const uint16_t* __restrict__ nb16, const double* __restrict__ gmu, int64_t d, int64_t nchains) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= d) return;
    const CoordConst e = cc[i];
    double n01 = 0.0, n23 = 0.0;  // (any graph: G1[i] as eight 16-bit ids where the lattice keeps gam0, gam1)
    if (nb16) {
        const double2 v = reinterpret_cast<const double2*>(nb16)[i];
        n01 = v.x;
        n23 = v.y;
    }
    for (int64_t ch = blockIdx.y; ch < nchains; ch += gridDim.y) {
        TrRecP* r = rec + ch * d + i;
        r->c = e.c;
        r->c100 = e.c100;
        r->gam0 = nb16 ? n01 : e.gam[0];
        r->gam1 = nb16 ? n23 : e.gam[1];
        r->gam2 = e.gam[2];
        r->gam3 = e.gam[3];
        r->gam4 = e.gam[4];
        r->tacc = gmu ? gmu[i] : 0.0;  // (the spare word: Γ[:,i]·μ of the flow where a mean is set)
    }
}
int launch_zz_trackp_consts(void* rec, const CoordConst* cc, const uint16_t* nb16, const double* gmu, int64_t d, int64_t nchains, void* stream) {
    const unsigned gy = (unsigned)((nchains < 1024) ? nchains : 1024);
    hipLaunchKernelGGL(zz_trackp_consts_kernel, dim3((unsigned)((d + 255) / 256), gy), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<TrRecP*>(rec), cc, nb16, gmu, d, nchains);
    return (int)hipGetLastError();
}

// adapt: the per-chain bounds live in the record lines; final_state reads them from the engine's c_chain array -- copied there on demand
__global__ __launch_bounds__(256) void zz_trackp_c_out_kernel(const TrRecP* __restrict__ rec, double* __restrict__ c_chain, int64_t n) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < n) c_chain[k] = rec[k].c;
}
int launch_zz_trackp_c_out(const void* rec, double* c_chain, int64_t n, void* stream) {
    hipLaunchKernelGGL(zz_trackp_c_out_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const TrRecP*>(rec), c_chain, n);
    return (int)hipGetLastError();
}

// (key) -> (key, t0) pairs, after the init kernel: every coordinate's last own proposal is the start of the run
__global__ __launch_bounds__(256) void zz_keys_to_pairs_kernel(const double* __restrict__ keys, double2* __restrict__ kp, int64_t n, double t0) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < n) kp[k] = make_double2(keys[k], t0);
}
int launch_zz_keys_to_pairs(const double* keys, void* kp, int64_t n, double t0, void* stream) {
    hipLaunchKernelGGL(zz_keys_to_pairs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, keys,
                       reinterpret_cast<double2*>(kp), n, t0);
    return (int)hipGetLastError();
}

}  // namespace pdmp
