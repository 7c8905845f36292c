// pdmp_logistic.hip -- zz_logistic_lds_kernel: the local ZigZag on the subsampled logistic target (config C4: scripts/logistic.jl:78-107,167,
// spdmp_inner! src/sfact.jl:73-145) for SMALL d, with the chain's state RESIDENT IN LDS for the whole time slice.
//
// pdmp_general.hip runs this configuration with every record in HBM: a proposal is a chain of ~8 dependent L2 / HBM round trips
// (queue block -> tables -> neighbour records -> sampled rows -> their records -> re-bound -> queue rescan), ~34 000 cycles with one proposal
// owning the wavefront, and 2.75 x the algorithmic traffic.  A chain of d = 442 coordinates is 14 KB of {x, θ, t, ∫x} and 3.5 KB of event
// times: here both live in LDS from the first event of a launch to the last (loaded once, written back once), so
//   * the queue is the key array itself: peek = 7 strided LDS reads per lane + a DPP minimum -- no first level, no rescans;
//   * every move (G1[i], the k_sub sampled rows' regressors, G2[i] on accept) and every dot product of a re-bound is an LDS access;
//   * what is left in global memory is read-only and L2-resident -- one packed 32-byte header per coordinate, one packed 128-byte record per
//     observation (its <= 6 regressors with their coefficients, y, m−y and the two control-variate constants: scripts/logistic.jl:86-93),
//     the design's column lists, the flow's (member, entry) tables -- plus the bound (t_old, a, b) of the popped coordinate (one record read
//     per proposal, requested before the tables are walked) and the fire-and-forget stores of new bounds, accept counts and events.
// One chain per wavefront, one wavefront per workgroup (no barriers); 8 chains per CU by LDS at d = 442 (18.7 KB each; 13 -- 11.9 KB each, and 128 registers -- when the engine's own path
// integrals ∫x_i dt are switched off, pdmp_ensemble_set_path_integrals).  With so few waves per SIMD nothing hides a wave's own latency, so
// the iteration is laid out as a short dependent chain:
//   * all of the iteration's random numbers in ONE lane-parallel Philox evaluation and ONE logarithm, taken while the header load is in
//     flight: lanes 0 .. k_sub−1 the sampled observations (global-rng stream), lane k_sub the thinning coin, the lanes above it log(u) of
//     the draws an accepted event's re-bounds will use (the first of them is the rejected proposal's);
//   * loads issued level by level -- a wave's loads return in order, so a slow one must never sit in front of a fast one needed sooner:
//     [1] the header; [2] G1[i] with its Γ values, the sampled entries of the design's column; [3] the sampled observations' records, then
//     the only reads that may come from HBM (bound in force, accept count, c_i), needed last; what an ACCEPTED event needs on top (the members'
//     table entries, their c_j and Γ[:,j]·μ, G2[i]) is requested by the accepted event -- one proposal in seven -- and not with every proposal;
//   * the rejected proposal's new bound (81 % of the proposals) is complete before the outcome is known: Γ[:,i]·x and Γ[:,i]·θ are formed
//     by the lanes that move G1[i] (x of G1[i] is final then; θ_i flips only on accept, which re-bounds all of G1[i] afresh).
// The arithmetic, the draw order and the summation orders are those of zz_general_run_kernel<.., LGFAST, ..> (and of the oracle): results are
// bit-identical (tests/test_gpu_general_parity.py, tests/test_gpu_configs_fullwidth.py run unchanged).
#include <hip/hip_runtime.h>

#include <cstdint>

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
__device__ __forceinline__ double l_pos(double x) {
    return (x > 0.0) ? x : ((x != x) ? x : 0.0);
}
__device__ __forceinline__ double l_poisson_time_L(double a, double b, double L) {  // src/poissontime.jl:8-30 with L = log(u)
    if (b == 0) return (a > 0) ? (-L / a) : L_INF;
    const double r = a / b;
    const double q = L * 2.0 / b;
    const double sq = sqrt((b > 0 && a < 0) ? -q : r * r - q);
    if (b > 0) return sq - r;
    if (a <= 0) return L_INF;
    if (-L <= -(a * a) / b + (a * a) / (2 * b)) return -sq - r;
    return L_INF;
}
__device__ __forceinline__ double l_sigmoid(double x) {  // sigmoid(x) = inv(one(x) + exp(-x)), scripts/logistic.jl:33
    return 1.0 / (1.0 + pdmp_exp(-x));
}

constexpr uint32_t LG_PCH = 64;  // (member, entry) products staged per chunk by the re-bound of an accepted event
constexpr int LG_KREG = 8;       // event times per lane: coordinate j lives in lane j % 64, slot j / 64 (d < 512)

// minimum of (key, index) pairs over the wave, lowest index on exactly equal keys; result in every lane
__device__ __forceinline__ void l_wave_argmin(double& key, uint32_t& idx) {
#define L_STEP(CTRL)                                                                                   \
    do {                                                                                               \
        const double k2 = l_dpp<CTRL>(key);                                                            \
        const uint32_t i2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)idx, CTRL, 0xf, 0xf, true);        \
        const bool take = (k2 < key) || (k2 == key && i2 < idx);                                       \
        key = take ? k2 : key;                                                                         \
        idx = take ? i2 : idx;                                                                         \
    } while (0)
    L_STEP(0xB1);
    L_STEP(0x4E);
    L_STEP(0x141);
    L_STEP(0x140);
    // (row_bcast fills lanes without a source with zeros: only the lanes that feed lane 63 matter, and those have one)
    L_STEP(0x142);
    L_STEP(0x143);
#undef L_STEP
    key = l_readlane(key, 63);
    idx = (uint32_t)__builtin_amdgcn_readlane((int)idx, 63);
}
__device__ __forceinline__ double l_shfl(double v, uint32_t src) {
    const int lo = __builtin_amdgcn_ds_bpermute((int)(src << 2), __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute((int)(src << 2), __double2hiint(v));
    return __hiloint2double(hi, lo);
}

}  // namespace

size_t zz_logistic_lds_bytes(int64_t d, int64_t dk, bool with_I) {
    (void)dk;  // (the event times live in registers: LG_KREG per lane)
    return (size_t)d * (with_I ? 32 : 24) + (size_t)2 * LG_PCH * 8 + (size_t)LG_PCH * 4 + 8;  // (+8: the chunk buffers start on 16 bytes for odd d too)
}

// TRK: tracked BOUNDS (pdmp_ensemble_set_gradient_tracking on this configuration; oracle: spdmp_zigzag_tracked_lg): every coordinate carries
// g_j = Γ[:,j]·x and gd_j = Γ[:,j]·θ at time tg_j in LT.trk, so that a proposal moves nothing but coordinate i and what the sampled rows read, a
// rejection re-derives its bound from (g_i, gd_i, tg_i), and an accepted event updates the k members of G1[i] instead of moving its two-hop
// set and summing every member's column afresh.  The gradient is the moving evaluation unchanged.
template <bool PROF, bool WITH_I, bool TRK = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void zz_logistic_lds_kernel(ZzRunParams P, ZzGeneralParams Q, ZzLogisticTables LT) {
    const int lane = threadIdx.x;
    const int64_t chain = blockIdx.x;
    const uint32_t d = (uint32_t)P.d;
    const uint32_t dk = (uint32_t)P.dk;

    extern __shared__ __align__(16) unsigned char smem[];
    double2* const xt = reinterpret_cast<double2*>(smem);                            // [d] (x_j, θ_j)
    double* const tt = reinterpret_cast<double*>(smem + (size_t)d * 16);             // [d] t_j
    double* const px = tt + d + (d & 1u);                                            // [LG_PCH] products Γ[r, j] x_r of one chunk; new keys (16-byte aligned)
    double* const pt = px + LG_PCH;                                                  // [LG_PCH] products Γ[r, j] θ_r
    uint32_t* const pj = reinterpret_cast<uint32_t*>(pt + LG_PCH);                   // [LG_PCH] coordinates of the new keys
    double* const II = reinterpret_cast<double*>(pj + LG_PCH);                       // [d] ∫ x_j up to t_j (WITH_I)

    ZzRec* const rec = P.rec + chain * (int64_t)d;
    double* const keys = P.keys + chain * P.dk;
    DevChain* const hdr = P.hdr + chain;
    pdmp_event* const ev = P.ev ? P.ev + chain * P.trace_cap : nullptr;
    double* const cmut = P.c_chain ? (P.c_chain + chain * (int64_t)d) : nullptr;
    const double* const cvec = cmut ? cmut : P.tb.c_shared;
    double4* const trkc = TRK ? reinterpret_cast<double4*>(LT.trk) + chain * (int64_t)d : nullptr;  // (g, gd, tg, -) per coordinate

    uint32_t status = hdr->c.status;
    if (status == PDMP_CHAIN_BOUND_VIOLATED || status == PDMP_CHAIN_STALLED) return;
    const uint64_t seed = hdr->seed;
    // counters of the launch as 32-bit differences (the wavefront's scalar registers are short: 64-bit running counters were spilled to vector
    // lanes and reloaded inside the loop); a launch makes far fewer than 2^32 draws
    uint64_t nm0 = hdr->c.ndraw_main, ng0 = hdr->c.ndraw_global;  // (bases of the 32-bit stream positions dnm, dng: advanced at every refill)
    const uint64_t ntrace0 = hdr->c.ntrace;
    uint32_t dnm = 0, dng = 0, dnum = 0, dnacc = 0, dnev = 0;
    const uint32_t trace_room = (P.trace_cap > 0) ? (uint32_t)(((uint64_t)P.trace_cap > ntrace0) ? ((uint64_t)P.trace_cap - ntrace0) : 0) : 0xffffffffu;
    double t_last = hdr->c.t_last;
    double t_event = hdr->t_event;
    status = PDMP_CHAIN_OK;
    const double T = P.T;
    const bool stop_before = (P.flags & PDMP_RUN_STOP_BEFORE) != 0;
    const bool adapt = P.adapt != 0;
    const int nq = (int)Q.ksub;  // sampled observations per gradient, one per lane (<= 32)

    uint64_t ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t ph_t0 = PROF ? (uint64_t)__builtin_readcyclecounter() : 0;
#ifdef PDMP_PHASE_MARKS  // (ISA reading: a comment line per phase boundary in the assembly)
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

    // ---------------- the chain's state comes on chip
    for (uint32_t j = lane; j < d; j += 64) {
        const ZzRec* r = rec + j;
        xt[j] = make_double2(r->x, r->th);
        tt[j] = r->t;
        if (WITH_I) II[j] = r->I;
    }
    // the queue: the key array itself, in registers -- coordinate j in lane j % 64, slot j / 64 (padding and the refresh slot: +Inf)
    double kreg[LG_KREG];
#pragma unroll
    for (int q = 0; q < LG_KREG; ++q) {
        const uint32_t j = (uint32_t)lane + 64u * (uint32_t)q;
        kreg[q] = (j < dk) ? keys[j] : L_INF;
    }
    auto set_key = [&](uint32_t j, double key) {  // (j wave-uniform or not: the owner lane takes it)
        const bool mine = (j & 63u) == (uint32_t)lane;
        const uint32_t slot = j >> 6;
#pragma unroll
        for (int q = 0; q < LG_KREG; ++q) kreg[q] = (mine && slot == (uint32_t)q) ? key : kreg[q];
    };
    L_ORDER();

    // smove_forward!(i::Int, ...) (src/sfact.jl:13-16) of one coordinate in LDS; returns (x at t′, θ).  A second move to the same t′ is
    // the identity (dt = 0), so lanes that meet on a coordinate store the same values.
    auto move1 = [&](uint32_t j, double tp) -> double2 {
        const double2 a = xt[j];
        const double t0 = tt[j];
        const double dt = tp - t0;
        const double xn = a.x + a.y * dt;
        xt[j].x = xn;
        tt[j] = tp;
        if (WITH_I) II[j] = II[j] + dt * ((a.x + xn) * 0.5);
        return make_double2(xn, a.y);
    };
    auto move_members = [&](uint32_t sp0, uint32_t p0, uint32_t p1, double tp) {
        for (uint32_t base = p0; base < p1; base += 64) {
            const uint32_t pp = base + (uint32_t)lane;
            if (pp < p1) (void)move1(P.tb.sidx[sp0 + pp], tp);
        }
        L_ORDER();
    };

    // s1 += px[z0 .. z1), s2 += pt[z0 .. z1) in order: the LDS reads of 8 terms are issued together, the adds stay sequential
    auto run_sums = [&](uint32_t z0, uint32_t z1, double& s1, double& s2) {
        uint32_t z = z0;
#pragma unroll 1
        for (; z + 8 <= z1; z += 8) {
            double u[8], w[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                u[q] = px[z + q];
                w[q] = pt[z + q];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                s1 += u[q];
                s2 += w[q];
            }
        }
        if (z < z1) {
            double u[8], w[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                u[q] = px[(z + q) & (LG_PCH - 1)];
                w[q] = pt[(z + q) & (LG_PCH - 1)];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (z + q < z1) {
                    s1 += u[q];
                    s2 += w[q];
                }
            }
        }
    };

    // the same sums over px[0 .. n), pt[0 .. n) for a WAVE-UNIFORM n (the proposal's own re-bound: |G1[i]|, 3 at the median): the remainder past the
    // last full group of 8 is added under scalar branches -- the predicated form above costs 16 adds and 32 selects whatever the remainder is
    auto run_sums_uniform = [&](uint32_t n, double& s1, double& s2) {
        const uint32_t n8 = n & ~7u;
        if (n8) run_sums(0u, n8, s1, s2);
        const uint32_t rem = (uint32_t)__builtin_amdgcn_readfirstlane((int)(n - n8));
#define LG_ADD(q)   \
    do {            \
        s1 += u[q]; \
        s2 += w[q]; \
    } while (0)
        if (rem) {  // (four at a time: the kernel sits at its register limit)
            double u[4], w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                u[q] = px[(n8 + q) & (LG_PCH - 1)];
                w[q] = pt[(n8 + q) & (LG_PCH - 1)];
            }
            switch (rem) {
                case 1: LG_ADD(0); break;
                case 2: LG_ADD(0); LG_ADD(1); break;
                case 3: LG_ADD(0); LG_ADD(1); LG_ADD(2); break;
                default: LG_ADD(0); LG_ADD(1); LG_ADD(2); LG_ADD(3); break;
            }
        }
        if (rem > 4u) {
            double u[3], w[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                u[q] = px[(n8 + 4 + q) & (LG_PCH - 1)];
                w[q] = pt[(n8 + 4 + q) & (LG_PCH - 1)];
            }
            switch (rem) {
                case 5: LG_ADD(0); break;
                case 6: LG_ADD(0); LG_ADD(1); break;
                default: LG_ADD(0); LG_ADD(1); LG_ADD(2); break;
            }
        }
#undef LG_ADD
    };

    // Random numbers, two blocks of 64 kept in registers: lane r holds draw gbase + r of the global-rng stream (its 64 bits: the sampled
    // observations are pdmp_randint of them) and draw mbase + r of the main stream (the uniform and its logarithm).  A proposal uses k_sub of
    // the first and 2 (rejected) or 1 + k (accepted) of the second, so one Philox pass serves 6 proposals' observations, one Philox + log pass
    // ~13 proposals' coins and bounds -- instead of one of each per proposal.  The lanes that WORK on the sampled observations are the ones
    // that hold their draws: [goff, goff + k_sub), goff = ng − gbase.
    constexpr uint32_t LG_MMARGIN = 26;  // members of an accepted event whose draws the block is guaranteed to hold (more: formed on demand)
    uint32_t gbase = 0u - 64u, mbase = 0u - 64u;  // (as differences too; empty blocks: the first iteration fills them)
    uint64_t gbits = 0;
    double mu = 0.0, mL = 0.0;
    bool running = stop_before || (t_event < T);
    PrioTurn prio;
    // (every load of the set-up is complete before the loop is entered: with loads pending on the loop's entry edge the compiler puts
    // s_waitcnt vmcnt(0) at the loop's head, where every iteration then also waits for the stores of the one before it)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    while (running) {
        prio.step();
        if (dnev >= trace_room) {
            status = PDMP_CHAIN_TRACE_FULL;
            break;
        }
        if (dnum >= P.count_limit) {  // (32-bit proposal count of the launch: pause, the host runs again)
            status = PDMP_CHAIN_PAUSED;
            break;
        }
        // ---------------- peek(Q), src/sfact.jl:77: the minimum of the key array, lowest coordinate on exact ties
        double tp = kreg[0];
        uint32_t i = (uint32_t)lane;
#pragma unroll
        for (int q = 1; q < LG_KREG; ++q) {
            const bool lt = kreg[q] < tp;  // (strict: the lower coordinate keeps an exact tie)
            tp = lt ? kreg[q] : tp;
            i = lt ? ((uint32_t)lane + 64u * (uint32_t)q) : i;
        }
        {
            // the minimum first (a DPP reduction of the keys alone), then who holds it: one lane in all but exactly tied cases, where the
            // (key, coordinate) reduction decides as before -- a third fewer vector instructions than reducing pairs every time
            const double gmin = l_wave_min(tp);
            const uint64_t holders = __ballot(tp == gmin);
            if (__popcll(holders) == 1) {
                i = (uint32_t)__builtin_amdgcn_readlane((int)i, __ffsll((unsigned long long)holders) - 1);
                tp = gmin;
            } else {
                l_wave_argmin(tp, i);
            }
        }
        if (!(tp < L_INF)) {
            status = PDMP_CHAIN_STALLED;
            break;
        }
        if (stop_before && !(tp < T)) break;
        t_last = tp;
        const LgCoord H = LT.coord[i];  // [1]
        LPHASE(0);
        // ---------------- every random number of the iteration, one Philox evaluation: lane q < k_sub -> draw ng + q of the global-rng stream
        // (rand(sampler), scripts/logistic.jl:84); lane k_sub -> the thinning coin, draw nm (:121); lane k_sub + 1 + r -> draw nm + 1 + r, the
        // uniform of the r-th re-bound of this proposal (r = 0: the rejected proposal's own, :139; r < k: the members of an accepted one, :134)
        if ((dng - gbase) + (uint32_t)nq > 64u) {  // (uniform)
            // the stream positions are 32-bit differences from a 64-bit base: folded into the base at every refill (two scalar additions),
            // so that no run length wraps them -- k_sub draws per proposal pass 2^32 after ~1.3e8 proposals of one chain
            ng0 += (uint64_t)dng;
            dng = 0u;
            gbase = 0u;
            gbits = pdmp_bits64(seed, PDMP_STREAM_GLOBAL, ng0 + (uint64_t)lane);
        }
        if ((dnm - mbase) + 2u + LG_MMARGIN > 64u) {
            nm0 += (uint64_t)dnm;
            dnm = 0u;
            mbase = 0u;
            mu = pdmp_bits_to_u01(pdmp_bits64(seed, PDMP_STREAM_MAIN, nm0 + (uint64_t)lane));
            mL = pdmp_log(mu);
        }
        const uint32_t goff = dng - gbase, moff = dnm - mbase;
        const bool qa = (uint32_t)lane - goff < (uint32_t)nq;
        const uint64_t bits = gbits;
        const double ucoin = l_readlane(mu, (int)moff);
        const uint32_t cp0 = H.cp0, k = H.k, sp0 = H.sp0, m = H.m;
        // [2]: the sampled entries of column i of the design FIRST (the observation records hang on them: the longest chain of look-ups), then
        // G1[i] and its Γ values -- every lane loads (lanes past k re-read the last entry): a load under a lane mask makes the number of loads
        // in flight unknown to the compiler, which then waits for ALL of them before the next level is requested (one more round trip)
        const uint32_t rdraw = (uint32_t)(((bits >> 32) * (uint64_t)H.l) >> 32);  // pdmp_randint
        const uint32_t ii = H.r0 + (qa ? rdraw : 0u);
        const uint32_t row = LT.a_row[ii];
        const double v = LT.a_val[ii];
        const bool gm = (uint32_t)lane < k;
        // (with tracked bounds only an accepted event, one proposal in seven, looks at G1[i]: its tables are read there, not here)
        const uint32_t lm = gm ? (uint32_t)lane : (k - 1u);  // (k >= 1: the diagonal)
        const uint32_t jm = TRK ? 0u : P.tb.sidx[sp0 + lm];
        const double wm = TRK ? 0.0 : P.tb.bval[cp0 + lm];
        uint4 mrec0 = make_uint4(0u, 0u, 0u, 0u);
        uint32_t qs0 = 0, qe0 = 0, g2a = 0xffffffffu;
        // [3]: the sampled observations
        const LgObs* const ob = LT.obs + row;
        const double4 c0 = *reinterpret_cast<const double4*>(&ob->y);      // y, ny, sn0, ns0
        const double4 w0 = *reinterpret_cast<const double4*>(&ob->val[0]);
        const double2 w1 = *reinterpret_cast<const double2*>(&ob->val[4]);
        const uint4 ix = *reinterpret_cast<const uint4*>(&ob->idx[0]);     // idx[0..5], ne, pad
        // ... and what may come from HBM: needed at the thinning test only, and requested AFTER the observation records (a wave's loads return in
        // order; the compiler would otherwise hoist these requests, which depend on i alone, in front of them)
        asm volatile("" ::: "memory");
        double cj0 = 0.0, gmu0 = 0.0;
        double4 trk_i = make_double4(0.0, 0.0, 0.0, 0.0);  // (g, gd, tg, -) of i
        if constexpr (TRK) {
            trk_i = trkc[i];
        }
        const double gmu_i = P.tb.gmu_b[i];
        const ZzRec* const ri = rec + i;
        const double told_i = ri->t_old, a_i = ri->a, b_i = ri->b;
        const uint64_t acc_i = ri->acc;
        const double c_i = cvec[i];
        // ---------------- smove_forward!(G, i, ...), :82, and with it the sums of i's own re-bound: Γ[:,i]·x, Γ[:,i]·θ in idot's order
        double s1r = 0.0, s2r = 0.0;
        if constexpr (TRK) {
            L_ORDER();
            if (lane == 0) (void)move1(i, tp);  // x_i at t′: the prior term and the event record read it
            L_ORDER();
        } else {
            L_ORDER();
            if (gm) {
                const double2 nx = move1(jm, tp);
                px[lane] = wm * nx.x;
                pt[lane] = wm * nx.y;
            }
            L_ORDER();
            run_sums_uniform((k < 64u) ? k : 64u, s1r, s2r);  // (every lane the same sums: LDS broadcasts)
            for (uint32_t base = 64u; base < k; base += 64u) {  // (columns beyond 64 entries: the intercept's)
                L_ORDER();
                const uint32_t pp = base + (uint32_t)lane;
                if (pp < k) {
                    const double2 nx = move1(P.tb.sidx[sp0 + pp], tp);
                    const double w = P.tb.bval[cp0 + pp];
                    px[lane] = w * nx.x;
                    pt[lane] = w * nx.y;
                }
                L_ORDER();
                run_sums_uniform((k - base < 64u) ? (k - base) : 64u, s1r, s2r);
            }
            L_ORDER();
        }
        LPHASE(1);
        // ---------------- ∇ϕmoving = γ0 x[i] − fdot_moving(A, At, i, t, x, θ, t′, F, μ, y, ny, k), scripts/logistic.jl:78-95,107
        double g;
        {
            const double prior = Q.gamma0 * xt[i].x;
            const int ne = qa ? (int)(ix.w & 0xffffu) : 0;
            const uint32_t id[6] = {ix.x & 0xffffu, ix.x >> 16, ix.y & 0xffffu, ix.y >> 16, ix.z & 0xffffu, ix.z >> 16};
            const double wv[6] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y};
            // idot_moving!(At, row, t, x, θ, t′, F), src/common.jl:33-42: every lane its own row, entries in ascending order
            double u = 0.0;
#pragma unroll
            for (int e = 0; e < 6; ++e) {
                if (e < ne) u += wv[e] * move1(id[e], tp).x;
                L_ORDER();
            }
            const double w = H.lk * v;  // l / k * vals[i]
            // the two sigmoids of an observation are evaluated SIDE BY SIDE: its own lane takes sigmoid(-u), the lane 32 further on (idle: at most
            // 32 observations are sampled) takes sigmoid(u) -- one exponential and one division per lane instead of two
            const bool qb = (((uint32_t)lane - 32u - goff) & 63u) < (uint32_t)nq;  // partner of an observation lane
            const double u_p = l_shfl(u, (uint32_t)(lane ^ 32));
            const double sg = l_sigmoid(qb ? u_p : -u);
            const double sg_p = l_shfl(sg, (uint32_t)(lane ^ 32));
            const double t1 = w * c0.x * sg;        // sigmoidn(u) = sigmoid(-u)
            const double t2 = w * c0.y * (-sg_p);   // nsigmoid(u) = -sigmoid(u)
            const double t3 = w * c0.x * c0.z;              // sigmoidn(u0), u0 = idot(At, row, μ): tabulated per observation
            const double t4 = w * c0.y * c0.w;              // nsigmoid(u0)
            // the four terms of every sampled observation go through LDS (over the chunk buffers, idle during a gradient): 2 reads of 16 bytes
            // and 4 adds per observation instead of 8 v_readlane and 4 adds -- the order of the sum is that of the draws (scripts/logistic.jl:84-92)
            double s = 0.0;
            {
                double2* const q2 = reinterpret_cast<double2*>(px);  // [k_sub][2] (k_sub <= 32: px and pt together)
                L_ORDER();
                if (qa) {
                    const uint32_t qrel = (uint32_t)lane - goff;
                    q2[2 * qrel] = make_double2(t1, t2);
                    q2[2 * qrel + 1] = make_double2(t3, t4);
                }
                L_ORDER();
                for (int z = 0; z < nq; z += 2) {
                    const double2 a0 = q2[2 * z], b0 = q2[2 * z + 1];
                    const bool two = z + 1 < nq;
                    const double2 a1 = q2[two ? 2 * z + 2 : 2 * z], b1 = q2[two ? 2 * z + 3 : 2 * z + 1];
                    s += a0.x;
                    s += a0.y;
                    s -= b0.x;
                    s -= b0.y;
                    if (two) {
                        s += a1.x;
                        s += a1.y;
                        s -= b1.x;
                        s -= b1.y;
                    }
                }
                L_ORDER();
            }
            dng += (uint32_t)Q.ksub;
            g = prior - s;
        }
        LPHASE(2);
        // (every load of the iteration is named here, before the first branch that can reach the loop's latch: hipcc's structured control flow has
        // edges no wave takes, and a load still in flight along one of them puts s_waitcnt vmcnt(0) at the loop's head -- a wait for this iteration's
        // STORES: round 5, read off the ISA of the tracked lattice kernel first)
        asm volatile("" ::"v"(c_i), "v"(gmu_i), "v"(told_i), "v"(a_i), "v"(b_i), "v"(acc_i), "v"(jm), "v"(wm), "v"(trk_i.x), "v"(trk_i.y), "v"(trk_i.z));
        const double th_i = xt[i].y;
        const double l_rate = l_pos(g * th_i);                   // :119
        const double lbound = l_pos(a_i + b_i * (tp - told_i));  // :119
        dnum += 1;
        dnm += 1;  // the coin is draw nm, :121
        const bool accept = (ucoin * lbound < l_rate);
        if (!accept) {
            // ---------------- rejected (:137-139): the bound from the sums taken above
            if constexpr (TRK) {
                s1r = trk_i.x + trk_i.y * (tp - trk_i.z);  // g_i advanced to t′
                s2r = trk_i.y;
            }
            const double a = c_i + (s1r - gmu_i) * th_i;  // src/fact_samplers.jl:51
            const double b = c_i / 100 + th_i * s2r;      // :52
            // (the new bound goes out before the event time is worked out: 70 instructions of the stores' way to the L2)
            if (lane == 0) {
                ZzRec* r = rec + i;
                r->t_old = tp;
                r->a = a;
                r->b = b;
            }
            asm volatile("" ::: "memory");
            const double key = tp + l_poisson_time_L(a, b, l_readlane(mL, (int)moff + 1));
            set_key(i, key);
            dnm += 1;
            L_ORDER();
            LPHASE(4);
            continue;
        }
        // ---------------- accepted
        dnacc += 1;
        double ci_new = c_i;
        if (l_rate >= lbound) {  // :123
            if (!adapt) {
                status = PDMP_CHAIN_BOUND_VIOLATED;
                break;
            }
            ci_new = c_i * P.factor;  // adapt!(c, i, factor), :127
            if (lane == 0) cmut[i] = ci_new;
        }
        // smove_forward!(G2, i, ...), :129 (the first 64 members were requested with the header's second level)
        if constexpr (!TRK) {
            // (what only an accepted event -- one proposal in seven -- needs is requested here, not with every proposal)
            mrec0 = Q.member[cp0 + (gm ? (uint32_t)lane : 0u)];  // (first 64 members)
            qs0 = P.tb.qptr[cp0];
            qe0 = P.tb.qptr[cp0 + ((k < 64u) ? k : 64u)];
            g2a = (k + (uint32_t)lane < m) ? P.tb.sidx[sp0 + k + (uint32_t)lane] : 0xffffffffu;  // G2[i], first 64
            cj0 = cvec[gm ? mrec0.x : i];
            gmu0 = P.tb.gmu_b[gm ? mrec0.x : i];
            if (g2a != 0xffffffffu) (void)move1(g2a, tp);
            if (k + 64u < m) move_members(sp0, k + 64u, m, tp);
        }
        if (lane == 0) {
            xt[i].y = -th_i;  // reflect!, :130
            rec[i].acc = acc_i + 1;
        }
        L_ORDER();
        LPHASE(3);
        if constexpr (TRK) {
            // ---------------- the members of G1[i], one per lane: sums advanced to t′, gd_j += Γ[j,i]·(−2θ_i), bound and event time (:131-135)
            const double delta = -th_i - th_i;
            for (uint32_t base = 0; base < k; base += 64) {
                const uint32_t jj = base + (uint32_t)lane;
                const bool valid = jj < k;
                const uint32_t j = valid ? P.tb.sidx[sp0 + jj] : i;
                const double w = valid ? P.tb.bval[cp0 + jj] : 0.0;
                const double4 tr = trkc[j];
                const double cj_tab = cvec[j];
                const double cj = (j == i) ? ci_new : cj_tab;
                const double gmu = P.tb.gmu_b[j];
                const uint32_t src = moff + 1u + jj;  // draw nm + jj (nm already counts the coin)
                double Ldraw = l_shfl(mL, (src < 64u) ? src : 63u);
                if (__ballot(valid && src >= 64u) != 0) {
                    const double Lx = pdmp_log(pdmp_u01(seed, PDMP_STREAM_MAIN, nm0 + (uint64_t)dnm + (uint64_t)jj));
                    Ldraw = (src >= 64u) ? Lx : Ldraw;
                }
                if (valid) {
                    const double gj = tr.x + tr.y * (tp - tr.z);
                    const double gdj = tr.y + w * delta;  // Γ[j, i] = Γ[i, j]
                    const double thj = xt[j].y;
                    const double a = cj + (gj - gmu) * thj;  // src/fact_samplers.jl:51
                    const double b = cj / 100 + thj * gdj;   // :52
                    const double keyj = tp + l_poisson_time_L(a, b, Ldraw);
                    trkc[j] = make_double4(gj, gdj, tp, 0.0);
                    ZzRec* r = rec + j;
                    r->t_old = tp;
                    r->a = a;
                    r->b = b;
                    px[lane] = keyj;
                    pj[lane] = j;
                }
                L_ORDER();
                const uint32_t last = (base + 64u < k) ? (base + 64u) : k;
                for (uint32_t z = 0; z < last - base; z += 4) {
                    uint32_t jn[4];
                    double kn[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        jn[q] = pj[(z + q) & (LG_PCH - 1)];
                        kn[q] = px[(z + q) & (LG_PCH - 1)];
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (z + q < last - base) set_key(jn[q], kn[q]);
                }
                L_ORDER();
            }
        } else
        // ---------------- ab + new event time of every member of G1[i] (:131-135; src/fact_samplers.jl:50-54).  The dot products keep idot's
        // order (ascending row); their products are formed LG_PCH at a time by all lanes, then every lane adds up the run of its member.
        for (uint32_t base = 0; base < k; base += 64) {
            const uint32_t jj = base + (uint32_t)lane;
            const bool valid = jj < k;
            const uint4 mrec = (base == 0) ? mrec0 : Q.member[cp0 + (valid ? jj : (k - 1u))];
            const uint32_t j = mrec.x;
            const uint32_t kj = valid ? mrec.y : 0u;
            const uint32_t q0 = mrec.z;
            const uint32_t last = (base + 64u < k) ? (base + 64u) : k;
            const uint32_t qs = (base == 0) ? qs0 : P.tb.qptr[cp0 + base], qe = (base == 0) ? qe0 : P.tb.qptr[cp0 + last];
            const double cj_tab = (base == 0) ? cj0 : cvec[valid ? j : i];
            const double cj = (j == i) ? ci_new : cj_tab;  // (c_i as adapted by THIS proposal travels in a register)
            const double gmu = (base == 0) ? gmu0 : P.tb.gmu_b[valid ? j : i];
            // draw nm + jj (nm already counts the coin): from the block where it reaches, else formed now
            const uint32_t src = moff + 1u + jj;
            double Ldraw = l_shfl(mL, (src < 64u) ? src : 63u);
            if (__ballot(valid && src >= 64u) != 0) {
                const double Lx = pdmp_log(pdmp_u01(seed, PDMP_STREAM_MAIN, nm0 + (uint64_t)dnm + (uint64_t)jj));
                Ldraw = (src >= 64u) ? Lx : Ldraw;
            }
            double s1 = 0.0, s2 = 0.0;
            // (the table entries of the next chunk are requested before this chunk's products are summed)
            uint32_t rn = 0;
            double wn = 0.0;
            if (qs + (uint32_t)lane < qe) {
                rn = LT.qrow16[qs + (uint32_t)lane];
                wn = Q.qbval[qs + (uint32_t)lane];
            }
            for (uint32_t cb = qs; cb < qe; cb += LG_PCH) {
                const uint32_t ce = (cb + LG_PCH < qe) ? (cb + LG_PCH) : qe;
                const uint32_t rc = rn;
                const double wc = wn;
                if (cb + LG_PCH + (uint32_t)lane < qe) {
                    rn = LT.qrow16[cb + LG_PCH + (uint32_t)lane];
                    wn = Q.qbval[cb + LG_PCH + (uint32_t)lane];
                }
                L_ORDER();
                if (cb + (uint32_t)lane < ce) {
                    const double2 a = xt[rc];
                    px[lane] = wc * a.x;
                    pt[lane] = wc * a.y;
                }
                L_ORDER();
                const uint32_t z0 = (q0 > cb) ? q0 : cb, z1 = (q0 + kj < ce) ? (q0 + kj) : ce;
                if (z0 < z1) run_sums(z0 - cb, z1 - cb, s1, s2);
            }
            L_ORDER();
            double keyj = L_INF;
            if (valid) {
                const double thj = xt[j].y;
                const double a = cj + (s1 - gmu) * thj;  // src/fact_samplers.jl:51
                const double b = cj / 100 + thj * s2;    // :52
                keyj = tp + l_poisson_time_L(a, b, Ldraw);
                ZzRec* r = rec + j;
                r->t_old = tp;
                r->a = a;
                r->b = b;
                px[lane] = keyj;
                pj[lane] = j;
            }
            L_ORDER();
            // the new keys go to their owner lanes
            for (uint32_t z = 0; z < last - base; z += 4) {
                uint32_t jn[4];
                double kn[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    jn[q] = pj[(z + q) & (LG_PCH - 1)];
                    kn[q] = px[(z + q) & (LG_PCH - 1)];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (z + q < last - base) set_key(jn[q], kn[q]);
            }
            L_ORDER();
        }
        dnm += k;
        L_ORDER();
        LPHASE(4);
        if (ev && lane == 0) {
            pdmp_event e;
            e.t = tp;
            e.i = (int64_t)i;
            e.x = xt[i].x;
            e.theta = -th_i;
            ev[ntrace0 + dnev] = e;
        }
        dnev += 1;
        t_event = tp;
        if (!stop_before && !(tp < T)) running = false;
        L_ORDER();
        LPHASE(5);
    }
    // ---------------- the state goes back (every other entry point reads the records)
    L_ORDER();
    for (uint32_t j = lane; j < d; j += 64) {
        const double2 a = xt[j];
        ZzRec* r = rec + j;
        r->x = a.x;
        r->th = a.y;
        r->t = tt[j];
        if (WITH_I) r->I = II[j];
    }
#pragma unroll
    for (int q = 0; q < LG_KREG; ++q) {
        const uint32_t j = (uint32_t)lane + 64u * (uint32_t)q;
        if (j < dk) keys[j] = kreg[q];
    }
    if (PROF && chain == 0 && lane == 0 && P.dbg) {
        for (int q = 0; q < 8; ++q) P.dbg[q] = (double)ph[q];
        P.dbg[10] = (double)dnum;
    }
#undef LPHASE
    if (lane == 0) {
        hdr->c.t_last = t_last;
        hdr->t_event = t_event;
        hdr->c.num += dnum;
        hdr->c.nacc += dnacc;
        hdr->c.ntrace = ntrace0 + dnev;
        hdr->c.nevents += dnev;
        hdr->c.ndraw_main = nm0 + dnm;
        hdr->c.ndraw_global = ng0 + dng;
        hdr->c.status = status;
    }
}

// g_j = Γ[:,j]·x, gd_j = Γ[:,j]·θ in idot's order (the sums of the initial bounds, src/sfact.jl:184-186) at t0: the tracked state's start
__global__ __launch_bounds__(256) void zz_logistic_track_init_kernel(const ZzRec* __restrict__ rec, ZzTables tb, int64_t d, int64_t nchains, double t0,
                                                                     double* __restrict__ trk) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= nchains * d) return;
    const int64_t chain = k / d, j = k - chain * d;
    const ZzRec* r = rec + chain * d;
    double g = 0.0, gd = 0.0;
    for (uint32_t p = tb.colptr[j]; p < tb.colptr[j + 1]; ++p) {
        const uint32_t row = tb.rowval[p];
        g += tb.bval[p] * r[row].x;
        gd += tb.bval[p] * r[row].th;
    }
    reinterpret_cast<double4*>(trk)[k] = make_double4(g, gd, t0, 0.0);
}
int launch_zz_logistic_track_init(const ZzRec* rec, const ZzTables& tb, int64_t d, int64_t nchains, double t0, double* trk, void* stream) {
    const int64_t n = nchains * d;
    hipLaunchKernelGGL(zz_logistic_track_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rec, tb, d, nchains, t0, trk);
    return (int)hipGetLastError();
}

bool zz_logistic_lds_supported(const ZzRunParams& p, const ZzGeneralParams& q, const ZzLogisticTables& lt) {
    return lt.coord != nullptr && !q.masked && q.target_kind == 1 && q.ksub >= 1 && q.ksub <= 32 && q.lg_ne_max <= 6 && !p.move_all && !p.has_refresh &&
           !q.local_bound && !q.sticky && q.flow_kind == 0 && !q.adaptscale && p.dk <= 64 * LG_KREG &&
           zz_logistic_lds_bytes(p.d, p.dk, true) <= 64 * 1024;
}

int launch_zz_logistic_lds(const ZzRunParams& p, const ZzGeneralParams& q, const ZzLogisticTables& lt, bool with_I, int64_t nchains,
                           void* stream) {
    const size_t lds = zz_logistic_lds_bytes(p.d, p.dk, with_I);
    const dim3 grid((unsigned)nchains), block(64);
    if (lt.trk != nullptr) {  // tracked bounds (no profiling instantiation)
        if (with_I) hipLaunchKernelGGL((zz_logistic_lds_kernel<false, true, true>), grid, block, lds, (hipStream_t)stream, p, q, lt);
        else hipLaunchKernelGGL((zz_logistic_lds_kernel<false, false, true>), grid, block, lds, (hipStream_t)stream, p, q, lt);
        return (int)hipGetLastError();
    }
    if (p.dbg) hipLaunchKernelGGL((zz_logistic_lds_kernel<true, true>), grid, block, zz_logistic_lds_bytes(p.d, p.dk, true), (hipStream_t)stream, p, q, lt);
    else if (with_I) hipLaunchKernelGGL((zz_logistic_lds_kernel<false, true>), grid, block, lds, (hipStream_t)stream, p, q, lt);
    else hipLaunchKernelGGL((zz_logistic_lds_kernel<false, false>), grid, block, lds, (hipStream_t)stream, p, q, lt);
    return (int)hipGetLastError();
}

}  // namespace pdmp
