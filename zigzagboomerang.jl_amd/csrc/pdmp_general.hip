// pdmp_general.hip -- local ZigZag event loop for neighbourhoods of ANY size (|G1[i]|, |S[i]| up to 4096) and for the
// subsampled logistic target of config C4 (scripts/logistic.jl:78-95,107: ∇ϕmoving with SelfMoving()).
//
// Same chain semantics as zz_local_run_kernel (spdmp_inner!, src/sfact.jl:73-145; one chain per wavefront, two-level
// 64-ary queue), but the neighbourhood is walked in chunks of 64 lanes and the read-only tables are the flow's CSC arrays
// (no per-coordinate blob: its size grows with the square of the column count).  Moved coordinates are written back at
// once and the (x, θ) of S[i] are staged in LDS by position.  This is the correctness path for dense-ish graphs such as
// the droptol-Hessian of the logistic regression (one column has 167 entries, two-hop sets reach 293); the grid-Laplace
// north star never comes here.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "../../include/pdmp_detmath.h"
#include "pdmp_engine.hpp"

namespace pdmp {

#define G_INF __builtin_inf()
#define G_ORDER()                        \
    do {                                 \
        __builtin_amdgcn_wave_barrier(); \
        asm volatile("" ::: "memory");   \
    } while (0)

__device__ __forceinline__ double g_readlane(double v, int srclane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ uint32_t g_uniform(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
template <int CTRL>
__device__ __forceinline__ double g_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);  // every lane has a valid source: no tied `old` operand, no copies
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// one v_min_f64 (fmin() would canonicalise each loaded / DPP-moved operand with a v_max_f64 x, x first); NaN loses
__device__ __forceinline__ double g_min(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// row steps, then row_bcast:15 / row_bcast:31: lane 63 holds the minimum of the wave (the other lanes garbage)
__device__ __forceinline__ double g_wave_min(double v) {
    v = g_min(v, g_dpp<0xB1>(v));
    v = g_min(v, g_dpp<0x4E>(v));
    v = g_min(v, g_dpp<0x141>(v));
    v = g_min(v, g_dpp<0x140>(v));
    v = g_min(v, g_dpp<0x142>(v));
    v = g_min(v, g_dpp<0x143>(v));
    return g_readlane(v, 63);
}
__device__ __forceinline__ double g_pos(double x) {
    return (x > 0.0) ? x : ((x != x) ? x : 0.0);
}
// poisson_time(a, b, u), src/poissontime.jl:8-30
__device__ __forceinline__ double g_poisson_time(double a, double b, double u) {
    const double L = pdmp_log(u);
    if (b > 0) {
        const double r = a / b;
        if (a < 0) return sqrt(-L * 2.0 / b) - r;
        return sqrt(r * r - L * 2.0 / b) - r;
    } else if (b == 0) {
        return (a > 0) ? (-L / a) : G_INF;
    } else {
        if (a <= 0) return G_INF;
        if (-L <= -(a * a) / b + (a * a) / (2 * b)) {
            const double r = a / b;
            return -sqrt(r * r - L * 2.0 / b) - r;
        }
        return G_INF;
    }
}
// the same with L = log(u) already taken (the draw's index is known before the bound is: Philox and the logarithm run while
// the dot products' operands are still on their way)
__device__ __forceinline__ double g_poisson_time_L(double a, double b, double L) {
    // the b != 0 formulas share a / b, L * 2 / b and the square root (sqrt(-L * 2.0 / b) == sqrt(-(L * 2.0 / b)) bit for bit):
    // lanes that disagree on the signs of a and b run one division pair and one square root, not one set per branch
    if (b == 0) return (a > 0) ? (-L / a) : G_INF;
    const double r = a / b;
    const double q = L * 2.0 / b;
    const double sq = sqrt((b > 0 && a < 0) ? -q : r * r - q);
    if (b > 0) return sq - r;
    if (a <= 0) return G_INF;
    if (-L <= -(a * a) / b + (a * a) / (2 * b)) return -sq - r;
    return G_INF;
}
// sigmoid(x) = inv(one(x) + exp(-x)), scripts/logistic.jl:33
__device__ __forceinline__ double g_sigmoid(double x) {
    return 1.0 / (1.0 + pdmp_exp(-x));
}

#define G_PCH 128u  // (member, entry) products staged per chunk by the re-bound step
#define G_PROW 65u  // doubles between the product lines of two sampled rows (ranged sweep)
#define G_NR 3     // sampled rows whose records are in flight together in the ranged sweep

size_t zz_general_lds_bytes(uint32_t nblk_pad, uint32_t mmax_pad, bool boom) {
    return (size_t)nblk_pad * 8 + (size_t)(boom ? 3 : 2) * mmax_pad * 8 + (size_t)nblk_pad * 4 + (size_t)2 * G_PCH * 8;
}
// ... plus, for the ranged sweep of long logistic rows, one line of products per sampled row (65 doubles apart: the lanes that add them up
// read different banks)
size_t zz_general_ranged_lds_bytes(int ksub) {  // (what the product lines need beyond the chunk buffers they overlay)
    const size_t need = (size_t)ksub * G_PROW * 8, have = (size_t)2 * G_PCH * 8;
    return need > have ? need - have : 0;
}

// LGFAST: instantiation for the plain spdmp + subsampled-logistic configuration (config C4): ZigZag flow, no refresh clock, no
// G = All(), no LocalBound, no adaptscale, not sticky -- the other modes' branches, scalars and table pointers drop out.
// RANGED: instantiation with the coordinate-range sweep of long logistic rows (two rows in flight: 27 more registers, 4 waves per SIMD)
template <bool PROF, bool LGFAST, bool RANGED>
__global__ __launch_bounds__(64) void zz_general_run_kernel(ZzRunParams P_in, ZzGeneralParams Q_in) {
    ZzRunParams P = P_in;
    ZzGeneralParams Q = Q_in;
    if constexpr (LGFAST) {
        Q.masked = 0;
        Q.ksub = 10;  // k = 10 sampled observations per gradient (scripts/logistic.jl:167): one batch, fixed trip counts
        P.move_all = 0;
        P.has_refresh = 0;
        Q.local_bound = 0;
        Q.sticky = 0;
        Q.flow_kind = 0;
        Q.adaptscale = 0;
        Q.target_kind = 1;
    }
    const int lane = threadIdx.x;
    const int64_t chain = blockIdx.x;
    const int64_t d = P.d;
    const uint32_t nblk = P.nblk;

    extern __shared__ __align__(16) unsigned char smem[];
    double* bk = reinterpret_cast<double*>(smem);
    double* sx = bk + P.nblk_pad;       // [mmax_pad] x of S[i] by position
    double* sth = sx + Q.mmax_pad;      // [mmax_pad] θ of S[i]
    double* smu = sth + Q.mmax_pad;     // [mmax_pad] μ of S[i] (FactBoomerang only: not allocated for ZigZag)
    uint32_t* bi = reinterpret_cast<uint32_t*>(smu + ((Q.flow_kind == 1) ? Q.mmax_pad : 0u));
    double* px = reinterpret_cast<double*>(bi + P.nblk_pad);  // [G_PCH] products of the bound's dot products, one chunk
    double* pt = px + G_PCH;                                   // [G_PCH]
    double* sprod = px;  // [64] products A'[e, row] * x[e] of one chunk (logistic gradient; not live at the same time)

    ZzRec* rec = P.rec + chain * d;
    // The moving half of a record -- (x, θ, t, ∫x dt) -- is reached through H(j): the first 32 bytes of rec[j], or, where the engine has split the
    // ensemble's state for this launch (the sweeps of long logistic rows, config C5: half the bytes per swept coordinate and twice the
    // coordinates per cache line), entry j of a packed array.  The bound, the accept flag and the keys stay where they are.
    struct ZzHot {
        double x, th, t, I;
    };
    char* const hot_base = Q.hot ? reinterpret_cast<char*>(Q.hot + (size_t)chain * (size_t)d * 4) : reinterpret_cast<char*>(rec);
    const int64_t hot_stride = Q.hot ? 32 : 64;
    auto H = [&](int64_t j) -> ZzHot* { return reinterpret_cast<ZzHot*>(hot_base + j * hot_stride); };
    double* keys = P.keys + chain * P.dk;
    DevChain* hdr = P.hdr + chain;
    pdmp_event* ev = P.ev ? P.ev + chain * P.trace_cap : nullptr;
    double* cmut = P.c_chain ? (P.c_chain + chain * d) : nullptr;
    const double* cvec = cmut ? cmut : P.tb.c_shared;
    double* sigc = Q.sig_chain ? (Q.sig_chain + chain * d) : nullptr;
    const bool local = Q.local_bound != 0;
    const bool sticky = Q.sticky != 0;
    uint32_t reb_count = 0;  // sticky: members re-bounded so far by the current rebound() call
    double* thf = sticky ? (P.thf + chain * d) : nullptr;
    double* rnw = local ? (Q.renew_chain + chain * d) : nullptr;

    uint32_t status = hdr->c.status;
    if (status == PDMP_CHAIN_BOUND_VIOLATED || status == PDMP_CHAIN_STALLED) return;
    const uint64_t seed = hdr->seed;
    uint64_t nm = hdr->c.ndraw_main, ng = hdr->c.ndraw_global;
    uint64_t num = hdr->c.num, nacc = hdr->c.nacc, ntrace = hdr->c.ntrace, nevents = hdr->c.nevents;
    uint64_t nrefresh = hdr->c.nrefresh;
    const bool boom = Q.flow_kind == 1;
    const bool has_refresh = P.has_refresh != 0;
    const double rhobar = sqrt(1 - Q.rho * Q.rho);
    double t_last = hdr->c.t_last;
    double t_event = hdr->t_event;
    status = PDMP_CHAIN_OK;
    const double T = P.T;
    const bool stop_before = (P.flags & PDMP_RUN_STOP_BEFORE) != 0;
    const bool adapt = P.adapt != 0;
    // pdmp_debug_set_phase_profile: cycles per phase, chain 0 (diagnostic build of the same loop)
    uint64_t ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t ph_t0 = PROF ? (uint64_t)__builtin_readcyclecounter() : 0;
#define GPHASE(k)                                                         \
    do {                                                                  \
        if (PROF) {                                                       \
            const uint64_t now_ = (uint64_t)__builtin_readcyclecounter(); \
            ph[k] += now_ - ph_t0;                                        \
            ph_t0 = now_;                                                 \
        }                                                                 \
    } while (0)

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
    G_ORDER();

    // level 1 of the queue after keys[] of G1[i][jj0 .. jj1) (and optionally of one extra coordinate) changed: every 64-key
    // block that holds a changed key is rescanned once (min, lowest index on ties -- the same pair the per-key update keeps)
    auto requeue = [&](uint32_t cp0, uint32_t jj0, uint32_t jj1, bool has_extra, uint32_t extra_j) {
        G_ORDER();
        const uint32_t cnt = jj1 - jj0 + (has_extra ? 1u : 0u);
        for (uint32_t base = 0; base < cnt; base += 64) {
            const uint32_t q = base + (uint32_t)lane;
            const bool valid = q < cnt;
            uint32_t bj = 0xffffffffu;
            if (valid) bj = ((jj0 + q < jj1) ? P.tb.rowval[cp0 + jj0 + q] : extra_j) >> 6;
            uint64_t todo = __ballot(valid);
            while (todo) {
                const int lead = __ffsll((unsigned long long)todo) - 1;
                const uint32_t bsel = (uint32_t)__builtin_amdgcn_readlane((int)bj, lead);
                todo &= ~__ballot(bj == bsel);
                const double kv = __hip_atomic_load(keys + (size_t)bsel * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const double mn = g_wave_min(kv);
                const uint64_t bl = __ballot(kv == mn);
                const int arg = bl ? (__ffsll((unsigned long long)bl) - 1) : 0;
                if (lane == 0) {
                    bk[bsel] = mn;
                    bi[bsel] = bsel * 64 + (uint32_t)arg;
                }
            }
        }
        G_ORDER();
    };
    // move the members S[i][p0 .. p1) to time tp, write them back, stage (x, θ) by position
    auto move_members = [&](uint32_t sp0, uint32_t p0, uint32_t p1, double tp) {
        for (uint32_t base = p0; base < p1; base += 64) {
            const uint32_t pp = base + (uint32_t)lane;
            if (pp < p1) {
                const uint32_t j = P.tb.sidx[sp0 + pp];
                ZzHot* r = H(j);
                const double x0 = r->x, th0 = r->th, t0 = r->t, I0 = r->I;
                const double dt = tp - t0;
                if (sticky && th0 == 0.0) {  // ssmove_forward!, src/ss_fact.jl:36-45: frozen coordinates keep their clock
                    sx[pp] = x0;
                    sth[pp] = th0;
                } else if (!boom) {
                    const double xn = x0 + th0 * dt;  // smove_forward!, src/sfact.jl:6-12
                    r->x = xn;
                    r->t = tp;
                    r->I = I0 + dt * ((x0 + xn) * 0.5);
                    sx[pp] = xn;
                    sth[pp] = th0;
                } else {
                    // smove_forward!(G, i, t, x, θ, t′, B::FactBoomerang), src/sfact.jl:29-36: rotation about μ
                    const double mj = Q.mu[j];
                    double sn, cs;
                    pdmp_sincos(dt, &sn, &cs);
                    const double xn = (x0 - mj) * cs + th0 * sn + mj;
                    const double thn = -(x0 - mj) * sn + th0 * cs;
                    r->x = xn;
                    r->th = thn;
                    r->t = tp;
                    r->I = I0 + (mj * dt + (x0 - mj) * sn + th0 * (1.0 - cs));
                    sx[pp] = xn;
                    sth[pp] = thn;
                    smu[pp] = mj;
                }
            }
        }
        G_ORDER();
    };
    // G = All() (pdmp, src/sfact.jl:236): smove_forward!(t, x, θ, t′, F) over all d coordinates (:23-28, :37-48)
    auto move_everything = [&](double tp) {
        for (uint32_t base = 0; base < (uint32_t)d; base += 64) {
            const uint32_t j = base + (uint32_t)lane;
            if (j < (uint32_t)d) {
                ZzHot* r = H(j);
                const double x0 = r->x, th0 = r->th, t0 = r->t, I0 = r->I;
                const double dt = tp - t0;
                if (!boom) {
                    const double xn = x0 + th0 * dt;
                    r->x = xn;
                    r->t = tp;
                    r->I = I0 + dt * ((x0 + xn) * 0.5);
                } else {
                    const double mj = Q.mu[j];
                    double sn, cs;
                    pdmp_sincos(dt, &sn, &cs);
                    r->x = (x0 - mj) * cs + th0 * sn + mj;
                    r->th = -(x0 - mj) * sn + th0 * cs;
                    r->t = tp;
                    r->I = I0 + (mj * dt + (x0 - mj) * sn + th0 * (1.0 - cs));
                }
            }
        }
        G_ORDER();
    };
    // stage members WITHOUT moving them (refresh branch: G1[i] is re-bounded at the coordinates' own clocks)
    auto stage_members = [&](uint32_t sp0, uint32_t p0, uint32_t p1) {
        for (uint32_t base = p0; base < p1; base += 64) {
            const uint32_t pp = base + (uint32_t)lane;
            if (pp < p1) {
                const uint32_t j = P.tb.sidx[sp0 + pp];
                sx[pp] = H(j)->x;
                sth[pp] = H(j)->th;
                if (boom) smu[pp] = Q.mu[j];
            }
        }
        G_ORDER();
    };
    // ab + new event time for the members jj0 .. jj1 of G1[i]; own_clock: Q[j] = t[j] + ... at j's own (stale) clock.
    // The dot products Γ[:,j]·x, Γ[:,j]·θ keep idot's order (ascending row, src/common.jl:16-24) but their PRODUCTS are formed
    // 64 at a time: the (member, entry) pairs of one pass are contiguous in pos16 / qbidx, the lanes stream them through LDS
    // in chunks of G_PCH, and every lane then adds up the run that belongs to its member.  (One lane per member walking its
    // column alone pays an L2 round trip per entry: 167 in a row for the intercept of config C4.)
    auto rebound = [&](uint32_t cp0, uint32_t jj0, uint32_t jj1, double tp, uint64_t draw0, bool per_member_draw,
                       bool own_clock) {
        for (uint32_t base = jj0; base < jj1; base += 64) {
            const uint32_t jj = base + (uint32_t)lane;
            const bool valid = jj < jj1;
            const uint32_t jjc = valid ? jj : (jj1 - 1u);
            const uint4 mrec = Q.member[cp0 + jjc];
            const uint32_t j = mrec.x;
            const uint32_t kj = valid ? mrec.y : 0u;
            const uint32_t q0 = mrec.z;
            const uint32_t last = (base + 64u < jj1) ? (base + 64u) : jj1;
            const uint32_t qs = P.tb.qptr[cp0 + base], qe = P.tb.qptr[cp0 + last];
            // sticky (src/ss_fact.jl:101-106,118-122,141-146): frozen members (θ[j] == 0) are neither re-bounded nor given a draw;
            // a re-bounded member takes draw draw0 + (its rank among the re-bounded ones)
            bool live = valid;
            uint32_t rank = jjc - jj0;
            if (sticky || Q.masked) {
                // (G ⊋ G1: the members of G[i] \ G1[i] are moved with the others but neither re-bounded nor given a draw, src/sfact.jl:131)
                live = valid && (!sticky || sth[jjc] != 0.0) && (!Q.masked || mrec.w != 0u);
                const uint64_t lball = __ballot(live);
                rank = reb_count + (uint32_t)__popcll(lball & ((1ull << lane) - 1ull));
                reb_count += (uint32_t)__popcll(lball);
            }
            const uint64_t di = per_member_draw ? (draw0 + (uint64_t)rank) : draw0;
            const double Ldraw = pdmp_log(pdmp_u01(seed, PDMP_STREAM_MAIN, di));
            double s1 = 0.0, s2 = 0.0;  // ZigZag: Γ[:,j]·x, Γ[:,j]·θ; FactBoomerang: Σ (x−μ)² + θ²
            for (uint32_t cb = qs; cb < qe; cb += G_PCH) {
                const uint32_t ce = (cb + G_PCH < qe) ? (cb + G_PCH) : qe;
                G_ORDER();
                for (uint32_t f = cb + (uint32_t)lane; f < ce; f += 64) {
                    const uint32_t ps = Q.pos16[f];
                    if (!boom) {
                        const double v = local ? Q.qtval[f] : Q.qbval[f];
                        px[f - cb] = v * sx[ps];
                        pt[f - cb] = v * sth[ps];
                    } else {
                        const double dx = sx[ps] - smu[ps];
                        px[f - cb] = dx * dx + sth[ps] * sth[ps];
                    }
                }
                G_ORDER();
                const uint32_t z0 = (q0 > cb) ? q0 : cb, z1 = (q0 + kj < ce) ? (q0 + kj) : ce;
                // sequential sums; the LDS reads of 8 terms are issued together, the adds stay in order
                uint32_t z = z0;
                if (!boom) {
                    for (; z + 8 <= z1; z += 8) {
                        double u[8], w[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            u[q] = px[z - cb + q];
                            w[q] = pt[z - cb + q];
                        }
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            s1 += u[q];
                            s2 += w[q];
                        }
                    }
                    for (; z < z1; ++z) {
                        s1 += px[z - cb];
                        s2 += pt[z - cb];
                    }
                } else {
                    for (; z + 8 <= z1; z += 8) {
                        double u[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) u[q] = px[z - cb + q];
#pragma unroll
                        for (int q = 0; q < 8; ++q) s1 += u[q];
                    }
                    for (; z < z1; ++z) s1 += px[z - cb];
                }
            }
            G_ORDER();
            if (live) {
                const double cj = cvec[j];
                const double xj = sx[jj], thj = sth[jj];
                double a, b;
                double hz = G_INF;
                if (local) {  // ab(G, j, x, θ, C::LocalBound, ∇ϕj, vj, Z), src/local.jl:2-6
                    const double gj = P.tb.gmu_t ? (s1 - P.tb.gmu_t[j]) : s1;
                    a = cj + gj * thj;
                    b = cj / 100 + thj * s2;
                    hz = 2.0 / cj / fabs(thj);
                } else if (!boom) {
                    a = cj + (s1 - P.tb.gmu_b[j]) * thj;  // src/fact_samplers.jl:51
                    b = cj / 100 + thj * s2;             // :52
                } else {
                    const double z = sqrt(s1);  // ab(G, i, x, θ, c, Z::FactBoomerang), src/fact_samplers.jl:58-65
                    const double z2 = xj * xj + thj * thj;
                    a = cj * sqrt(z2) * z + z2 * Q.diag[j];
                    b = 0.0;
                }
                ZzRec* r = rec + j;
                const double tj = own_clock ? H(j)->t : tp;
                double dtn = g_poisson_time_L(a, b, Ldraw);
                if (local) {  // next_time, src/not_fact_samplers.jl:43-50: the bound expires after its horizon
                    const bool rn = dtn > hz;
                    dtn = rn ? hz : dtn;
                    rnw[j] = rn ? 1.0 : 0.0;
                }
                if (sticky) {  // queue_time!, src/ss_fact.jl:54-66: the earlier of the reflection proposal and the hitting time of 0
                    const double tfreeze = (thj * xj >= 0) ? G_INF : (-xj / thj);  // freezing_time, :10-16
                    const bool fz = tfreeze <= dtn;
                    dtn = fz ? tfreeze : dtn;
                    r->acc = fz ? 1u : 0u;  // f[j]
                }
                const double key = tj + dtn;
                r->t_old = tj;
                r->a = a;
                r->b = b;
                keys[j] = key;
            }
        }
    };

    bool running = stop_before || (t_event < T);
    PrioTurn prio;
    while (running) {
        prio.step();
        if (P.trace_cap > 0 && ntrace >= (uint64_t)P.trace_cap) {
            status = PDMP_CHAIN_TRACE_FULL;
            break;
        }
        // ---------------- peek(Q), src/sfact.jl:77
        double mk = G_INF;
        uint32_t mb = 0xffffffffu;
        for (uint32_t b = lane; b < nblk; b += 64) {
            const double v = bk[b];
            if (v < mk) {
                mk = v;
                mb = b;
            }
        }
        const double tp = g_wave_min(mk);
        if (!(tp < G_INF)) {
            status = PDMP_CHAIN_STALLED;
            break;
        }
        if (stop_before && !(tp < T)) break;
        uint32_t blk;
        {
            const uint64_t ball = __ballot(mk == tp);
            uint32_t cand = (mk == tp) ? mb : 0xffffffffu;  // exact ties: lowest block
            for (int off = 32; off >= 1; off >>= 1) {
                const uint32_t o = (uint32_t)__shfl_xor((int)cand, off, 64);
                cand = (o < cand) ? o : cand;
            }
            (void)ball;
            blk = g_uniform(cand);
        }
        const uint32_t i = g_uniform(bi[blk]);
        t_last = tp;
        GPHASE(0);

        const uint32_t cp0 = P.tb.colptr[i];
        const uint32_t k = P.tb.colptr[i + 1] - cp0;
        const uint32_t sp0 = P.tb.sptr[i];
        const uint32_t m = P.tb.sptr[i + 1] - sp0;
        const uint32_t self = Q.selfpos16[i];
        const ZzRec* ri = rec + i;
        const double told_i = ri->t_old, a_i = ri->a, b_i = ri->b;
        const uint64_t acc_i = ri->acc;

        if (has_refresh && i == (uint32_t)d) {
            // ---------------- refresh clock, src/sfact.jl:78-114 (quirks restated: two independent global-rng coordinate
            // draws :80,:84; G1[i] re-bounded at the coordinates' own clocks :110-114)
            const uint32_t i1 = pdmp_randint(seed, PDMP_STREAM_GLOBAL, ng, (uint32_t)d);
            ng += 1;
            if (P.move_all) move_everything(tp);
            else move_members(P.tb.sptr[i1], 0, P.tb.colptr[i1 + 1] - P.tb.colptr[i1], tp);  // :82
            const uint32_t i2 = pdmp_randint(seed, PDMP_STREAM_GLOBAL, ng, (uint32_t)d);
            ng += 1;
            const uint32_t cp2 = P.tb.colptr[i2];
            const uint32_t k2 = P.tb.colptr[i2 + 1] - cp2;
            const uint32_t sp2 = P.tb.sptr[i2];
            const uint32_t m2 = P.tb.sptr[i2 + 1] - sp2;
            const uint32_t self2 = Q.selfpos16[i2];
            if (P.move_all) stage_members(sp2, k2, m2);  // G2 = nothing (:172): nothing to move, values still needed below
            else move_members(sp2, k2, m2, tp);           // smove_forward!(G2, i, ...), :85
            stage_members(sp2, 0, k2);
            double thn;
            double sg2 = sigc ? sigc[i2] : P.tb.sigma[i2];
            if (Q.adaptscale && !boom) {  // :86-91, no random draw
                const double adapt_g = 0.01, adapt_t0 = 15., adapt_k = 0.75;
                const double acc2 = (double)(1 + (int64_t)rec[i2].acc);
                const double pre = pdmp_log(2.0) - sqrt(1.0 + tp) / (adapt_g * (1.0 + tp + adapt_t0)) *
                                                       pdmp_log(acc2 / (1.0 + 0.3 * tp));
                const double eta = pdmp_exp(-adapt_k * pdmp_log(1 + tp));  // (1 + t′)^(-adapt_κ)
                sg2 = pdmp_exp(eta * pre + (1 - eta) * pdmp_log(sg2));
                const double tho = sth[self2];
                thn = sg2 * ((tho > 0) ? 1.0 : ((tho < 0) ? -1.0 : tho));  // σ[i]*sign(θ[i])
            } else {
                if (Q.adaptscale) {  // :93-98
                    const double ti2 = H(i2)->t;
                    const double effi = (1 + 2 * Q.rho / (1 - Q.rho));
                    const double tau = effi / (ti2 * P.lambda_ref);
                    if (tau < 0.2) {
                        const double r = 0.3 * ti2 / (double)(int64_t)rec[i2].acc;
                        const double dir = (double)((r > 1.66) - (r < 0.6));
                        const double sq = sqrt(tau / P.lambda_ref);
                        sg2 = sg2 * pdmp_exp(dir * 0.03 * ((1.0 < sq) ? 1.0 : sq));
                    }
                }
                if (boom) {  // :103  θ[i] = ρ θ[i] + ρ̄ σ[i] randn(rng)
                    thn = Q.rho * sth[self2] + rhobar * sg2 * pdmp_randn(seed, PDMP_STREAM_MAIN, nm);
                } else {     // :100-101  θ[i] = σ[i] rand(rng, (-1,1))
                    thn = sg2 * ((pdmp_u01(seed, PDMP_STREAM_MAIN, nm) < 0.5) ? -1.0 : 1.0);
                }
                nm += 1;
            }
            if (sigc && lane == 0) sigc[i2] = sg2;
            G_ORDER();
            if (lane == 0) {
                sth[self2] = thn;
                H(i2)->th = thn;
            }
            G_ORDER();
            const double newref = tp + (-pdmp_log(pdmp_u01(seed, PDMP_STREAM_GLOBAL, ng))) / P.lambda_ref;  // :108
            ng += 1;
            reb_count = 0;
            rebound(cp2, 0, k2, tp, nm, true, true);  // :110-114
            nm += Q.masked ? (uint64_t)reb_count : (uint64_t)k2;
            if (lane == 0) keys[d] = newref;
            requeue(cp2, 0, k2, true, (uint32_t)d);
            if (ev && lane == 0) {  // event(i, t, x, θ, F) = (t[i], i, x[i], θ[i]) at i's own clock, :143
                pdmp_event e;
                e.t = __hip_atomic_load(&H(i2)->t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                e.i = (int64_t)i2;
                e.x = sx[self2];
                e.theta = thn;
                ev[ntrace] = e;
            }
            nrefresh += 1;
            ntrace += 1;
            nevents += 1;
            t_event = tp;
            if (!stop_before && !(tp < T)) running = false;
            G_ORDER();
            continue;
        }
        // ∇ϕmoving of the subsampled logistic target (SelfMoving: it moves what it reads); needs sx[self] = x[i] at t′
        auto logistic_gradient = [&]() -> double {
            double urow = 0.0;
            // ∇ϕmoving = γ0*x[i] - fdot_moving(A, At, i, t, x, θ, t′, F, μ, y, ny, k), scripts/logistic.jl:78-95,107
            const double prior = Q.gamma0 * sx[self];
            double s = 0.0;
            const int64_t r0 = Q.A_colptr[i];
            const int64_t l = Q.A_colptr[i + 1] - r0;
            // The k_sub sampled observations are handled 64 at a time, one per lane: the draws, the row look-ups and the
            // sigmoids run side by side; only the sums keep the reference's order (u over a row's entries, s over q).
            // Moving a coordinate twice to the same t′ is the identity (dt = 0), so rows that share coordinates may move
            // them concurrently: every lane writes the same values.
            for (int64_t qb = 0; qb < Q.ksub; qb += 64) {
                const int nq = (int)((Q.ksub - qb < 64) ? (Q.ksub - qb) : 64);
                const bool qa = lane < nq;
                // rand(sampler): draw ng + q of the global-rng stream
                const uint32_t rdraw = pdmp_randint(seed, PDMP_STREAM_GLOBAL, ng + (uint64_t)qb + (uint64_t)lane, (uint32_t)l);
                const int64_t ii = r0 + (int64_t)(qa ? rdraw : 0u);
                const int64_t row = Q.A_rowval[ii];
                const double v = Q.A_nzval[ii];
                const int64_t e0 = Q.At_colptr[row];
                const int ne = qa ? (int)(Q.At_colptr[row + 1] - e0) : 0;
                const double yr = Q.y[row], nyr = Q.ny[row], sn0 = Q.sn0[row], ns0 = Q.ns0[row];
                int incl = ne;  // inclusive scan of the row lengths over the lanes (LGFAST: only lanes 0..9 carry rows)
                for (int off = 1; off < (LGFAST ? 16 : 64); off <<= 1) {
                    const int o = __shfl_up(incl, off, 64);
                    if (lane >= off) incl += o;
                }
                const int etot = __builtin_amdgcn_readlane(incl, LGFAST ? 15 : 63);
                const int excl = incl - ne;
                // Long rows of a dense design (thousands of coefficients, the sampled rows share most of their coordinates): swept in coordinate
                // RANGES, every sampled row's entries inside a range before the next range, so that a record comes from HBM once per evaluation
                // and from L2 for the other rows (row by row it is re-fetched k_sub times: 10^4 records x 4096 chains do not stay cached).  A row's
                // running sum still takes its entries in ascending order, and a move is the identity after the first, so nothing else changes.
                const bool ranged = RANGED && !LGFAST && Q.lg_range > 0 && etot >= 2048;
                if constexpr (RANGED) if (ranged) {
                    // Lane z keeps row z's cursor and running sum.  A super-step takes ONE chunk of every row that still has entries inside the
                    // range -- G_NR rows at a time, so that all their records are requested before any is used --, moves the coordinates and
                    // leaves the products in LDS, one line of 64 per row; then every row's products are added up by ITS lane, all rows side by
                    // side, in entry order.  (Summing a row's chunk with wave-uniform readlane + add costs 3 instructions per matrix entry for
                    // the whole wavefront -- 2·10⁵ per gradient of config C5, which made this kernel VALU-bound; now it is one LDS read and
                    // one add per entry of the LONGEST chunk of the super-step.)
                    int64_t e_cur = e0;
                    const int64_t e_end = e0 + (int64_t)ne;
                    urow = 0.0;
                    double* const prodm = px;  // [k_sub][G_PROW]: over the bound's chunk buffers (not live during a gradient) and beyond
                    const uint32_t NOIX = 0x7fffffffu;
                    auto rd64 = [&](int64_t v, int z) -> int64_t {
                        return (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), z) << 32) |
                                         (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)v, z));
                    };
                    const uint64_t allrows = (nq >= 64) ? ~0ull : ((1ull << nq) - 1ull);
                    for (int64_t rb = 0; rb < d; rb += Q.lg_range) {
                        const int64_t rend = rb + Q.lg_range;
                        uint64_t active = allrows;  // rows that may still have entries below rend (wave-uniform)
                        while (active) {
                            int mycnt = 0;
                            uint64_t m_ = active;
                            while (m_) {
                                // G_NR rows at a time: all their records are requested before any is used
                                int zr[G_NR];
                                bool has[G_NR];
                                int64_t ec[G_NR];
                                uint32_t cc[G_NR];
                                double we[G_NR];
                                bool in[G_NR];
                                int cnt[G_NR];
#pragma unroll
                                for (int r = 0; r < G_NR; ++r) {
                                    has[r] = m_ != 0;
                                    zr[r] = has[r] ? (__ffsll((unsigned long long)m_) - 1) : zr[0];
                                    if (has[r]) m_ &= m_ - 1;
                                    ec[r] = rd64(e_cur, zr[r]);
                                    const int64_t ee = rd64(e_end, zr[r]);
                                    const int64_t f = ec[r] + lane;
                                    cc[r] = NOIX;
                                    we[r] = 0.0;
                                    if (has[r] && f < ee) {
                                        cc[r] = Q.At_row32[f];
                                        we[r] = Q.At_nzval[f];
                                    }
                                }
                                double x_[G_NR], th_[G_NR], t_[G_NR], I_[G_NR];
#pragma unroll
                                for (int r = 0; r < G_NR; ++r) {
                                    in[r] = cc[r] != NOIX && (int64_t)cc[r] < rend;  // (entries ascend: the lanes inside the range are a prefix)
                                    cnt[r] = __popcll(__ballot(in[r]));
                                    x_[r] = th_[r] = t_[r] = I_[r] = 0.0;
                                    if (in[r]) {
                                        const ZzHot* const h = H(cc[r]);
                                        x_[r] = h->x;
                                        th_[r] = h->th;
                                        t_[r] = h->t;
                                        I_[r] = h->I;
                                    }
                                }
                                // every row's move is worked out first -- that uses every load of the super-step --, then the stores go out together.
                                // Written row by row (move, store, next row) the compiler has to assume a load may still be pending when it reaches
                                // the next row's use (the loads stand under lane masks) and waits with s_waitcnt vmcnt(0) -- behind the stores it
                                // has just issued: every row, and the loop's head, then waited for a write to reach the L2.
                                double xe_[G_NR], In_[G_NR];
                                bool st_[G_NR];
#pragma unroll
                                for (int r = 0; r < G_NR; ++r) {
                                    const double dt = tp - t_[r];
                                    xe_[r] = x_[r] + th_[r] * dt;
                                    In_[r] = I_[r] + dt * ((x_[r] + xe_[r]) * 0.5);
                                    st_[r] = in[r] && dt != 0.0;  // (a coordinate an earlier row -- or the proposal's own move -- brought to t′ already)
                                }
                                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): nothing is in flight any more, and the compiler knows
#pragma unroll
                                for (int r = 0; r < G_NR; ++r) {
                                    if (in[r]) {  // (a record two rows move gets the same values stored twice)
                                        ZzHot* const h = H(cc[r]);
                                        if (st_[r]) {
                                            h->x = xe_[r];
                                            h->t = tp;
                                            h->I = In_[r];
                                        }
                                        prodm[zr[r] * G_PROW + lane] = we[r] * xe_[r];
                                    }
                                    if (has[r]) {
                                        if (lane == zr[r]) {
                                            e_cur = ec[r] + cnt[r];
                                            mycnt = cnt[r];
                                        }
                                        if (cnt[r] < 64) active &= ~(1ull << zr[r]);
                                    }
                                }
                            }
                            G_ORDER();
                            for (int k2 = 0; k2 < 64; ++k2) {
                                const bool more = k2 < mycnt;
                                if (__ballot(more) == 0) break;
                                if (more) urow += prodm[lane * G_PROW + k2];
                            }
                            G_ORDER();
                        }
                    }
                }
                // idot_moving!(At, row, t, x, θ, t′, F), src/common.jl:33-42: move the rows' coordinates; products to LDS
                // (LGFAST: every observation has at most 6 regressors, so the 10 rows always fit one 64-entry chunk)
                for (int fb = 0; fb < (ranged ? 0 : (LGFAST ? ((etot > 0) ? 1 : 0) : etot)); fb += 64) {
                    const int f = fb + lane;
                    int q = 0;
                    for (int z = 0; z < nq; ++z) q += (__builtin_amdgcn_readlane(incl, z) <= f) ? 1 : 0;
                    q = (q < nq) ? q : (nq - 1);
                    const int exq = __shfl(excl, q, 64);
                    const int e0lo = __shfl((int)(uint32_t)(uint64_t)e0, q, 64);
                    const int e0hi = __shfl((int)(uint32_t)((uint64_t)e0 >> 32), q, 64);
                    if (f < etot) {
                        const int64_t eq = (int64_t)(((uint64_t)(uint32_t)e0hi << 32) | (uint64_t)(uint32_t)e0lo) + (int64_t)(f - exq);
                        const int64_t cc = (int64_t)Q.At_row32[eq];
                        const double we = Q.At_nzval[eq];
                        ZzHot* r = H(cc);
                        const double x0 = r->x, th0 = r->th, t0 = r->t, I0 = r->I;
                        const double dt = tp - t0;
                        const double xe = x0 + th0 * dt;
                        if (dt != 0.0) {  // (a coordinate an earlier row of this evaluation -- or the proposal's own move -- brought to t′ already:
                                          // the move is the identity, and not storing it keeps a re-fetched line clean: the k_sub rows of a
                                          // dense design share most of their coordinates)
                            r->x = xe;
                            r->t = tp;
                            r->I = I0 + dt * ((x0 + xe) * 0.5);
                        }
                        sprod[lane] = we * xe;
                    }
                    G_ORDER();
                    // every lane continues the running sum of its row over the entries of this chunk, in entry order
                    {
                        const int z0 = (excl > fb) ? excl : fb, z1 = (incl < fb + 64) ? incl : (fb + 64);
                        if (z0 < z1) {
                            double u = (excl >= fb) ? 0.0 : urow;
                            for (int z = z0; z < z1; ++z) u += sprod[z - fb];
                            urow = u;
                        }
                    }
                    G_ORDER();
                }
                const double u = (ne > 0) ? urow : 0.0;
                const double w = (double)l / (double)Q.ksub * v;
                const double t1 = w * yr * g_sigmoid(-u);     // sigmoidn(u) = sigmoid(-u)
                const double t2 = w * nyr * (-g_sigmoid(u));  // nsigmoid(u) = -sigmoid(u)
                const double t3 = w * yr * sn0;               // sigmoidn(u0), u0 = idot(At, row, μ): tabulated per observation
                const double t4 = w * nyr * ns0;              // nsigmoid(u0)
                for (int z = 0; z < nq; ++z) {
                    s += g_readlane(t1, z);
                    s += g_readlane(t2, z);
                    s -= g_readlane(t3, z);
                    s -= g_readlane(t4, z);
                }
            }
            ng += (uint64_t)Q.ksub;
            return prior - s;
        };
        if (sticky) {
            // ---------------- sspdmp_inner!, src/ss_fact.jl:78-157, for neighbourhoods of any size
            const double x_i0 = H(i)->x, th_i0 = H(i)->th;
            const bool is_freeze = g_uniform(acc_i != 0 ? 1u : 0u) != 0;  // f[i]: rec.acc holds the flag for sticky chains
            const bool is_thaw = !is_freeze && g_uniform((x_i0 == 0 && th_i0 == 0) ? 1u : 0u) != 0;
            bool emit = true;
            if (is_freeze) {  // case 1, :87-107
                const double dt = tp - H(i)->t;
                const double xs = x_i0 + th_i0 * dt;  // smove_forward!(i, ...), :88
                if (fabs(xs) > 1e-8) {                // :89-91
                    status = PDMP_CHAIN_BOUND_VIOLATED;
                    break;
                }
                const double knew = tp - pdmp_log(pdmp_u01(seed, PDMP_STREAM_MAIN, nm)) / P.kappa[i];  // :96
                nm += 1;
                if (lane == 0) {
                    ZzRec* w = rec + i;
                    ZzHot* wh = H(i);
                    wh->I = wh->I + dt * ((x_i0 + xs) * 0.5);
                    wh->x = 0.0 * th_i0;  // x[i] = -0*θ[i], :92
                    wh->th = 0.0;         // :93
                    wh->t = tp;
                    w->t_old = tp;       // :94
                    w->acc = 0;          // f[i] = false, :95
                    thf[i] = th_i0;
                    keys[i] = knew;
                }
                G_ORDER();
                if (!P.strong_upperbounds) {  // :97-107
                    move_members(sp0, 0, m, tp);  // G and G2, non-frozen only (i is frozen now)
                    reb_count = 0;
                    rebound(cp0, 0, k, tp, nm, true, false);
                    nm += reb_count;
                    requeue(cp0, 0, k, false, 0u);  // includes i's own block
                } else {
                    requeue(cp0, self, self + 1u, false, 0u);
                }
            } else if (is_thaw) {  // case 2, :108-123
                double thn = thf[i];  // θ[i], θf[i] = θf[i], 0.0, :110
                uint32_t head = 0;
                if (P.reversible) {  // :111-113
                    thn *= (pdmp_u01(seed, PDMP_STREAM_MAIN, nm) < 0.5) ? -1.0 : 1.0;
                    head = 1;
                }
                if (lane == 0) {
                    ZzRec* w = rec + i;
                    H(i)->t = tp;      // :109
                    H(i)->th = thn;
                    w->t_old = tp;  // :114
                    thf[i] = 0.0;
                }
                G_ORDER();
                move_members(sp0, 0, m, tp);  // :115-116 (i itself: dt = 0)
                reb_count = 0;
                rebound(cp0, 0, k, tp, nm + head, true, false);  // :117-123, non-frozen members including i
                nm += head + reb_count;
                requeue(cp0, 0, k, false, 0u);
            } else {  // reflection proposal, :124-152
                move_members(sp0, 0, k, tp);  // ssmove_forward!(G, i, ...), :125
                double g = 0.0;
                if (Q.target_kind == 1) {  // ∇ϕ_(∇ϕ, t, x, θ, i, t′, F, S::SelfMoving, args...), src/sfact.jl:68
                    g = logistic_gradient();
                } else {
                    for (uint32_t p = 0; p < k; ++p) g += P.tb.tval[cp0 + p] * sx[p];
                    if (P.tb.gmu_t) g = g - P.tb.gmu_t[i];
                }
                const double th_i = sth[self];
                const double l_rate = g_pos(g * th_i);
                const double lbound = g_pos(a_i + b_i * (tp - told_i));  // :128
                num += 1;
                const double coin = pdmp_u01(seed, PDMP_STREAM_MAIN, nm);
                nm += 1;
                if (coin * lbound < l_rate) {  // :130
                    nacc += 1;
                    if (l_rate > lbound) {  // :132
                        if (!adapt) {
                            status = PDMP_CHAIN_BOUND_VIOLATED;
                            break;
                        }
                        nacc = 0;  // acc = num = 0, :134
                        num = 0;
                        if (lane == 0) cmut[i] = cvec[i] * P.factor;  // :135
                    }
                    move_members(sp0, k, m, tp);  // :138
                    if (lane == 0) {
                        sth[self] = -th_i;  // :139
                        H(i)->th = -th_i;
                    }
                    G_ORDER();
                    reb_count = 0;
                    rebound(cp0, 0, k, tp, nm, true, false);  // :140-146
                    nm += reb_count;
                    requeue(cp0, 0, k, false, 0u);
                } else {  // :147-151
                    reb_count = 0;
                    rebound(cp0, self, self + 1u, tp, nm, true, false);
                    nm += reb_count;
                    requeue(cp0, self, self + 1u, false, 0u);
                    emit = false;
                }
            }
            if (emit) {  // push!(Ξ, event(i, t, x, θ, F)), :154
                G_ORDER();
                if (ev && lane == 0) {
                    const ZzHot* w = H(i);
                    pdmp_event e;
                    e.t = __hip_atomic_load(&w->t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    e.i = (int64_t)i;
                    e.x = __hip_atomic_load(&w->x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    e.theta = __hip_atomic_load(&w->th, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ev[ntrace] = e;
                }
                ntrace += 1;
                nevents += 1;
                t_event = tp;
                if (!stop_before && !(tp < T)) running = false;
            }
            G_ORDER();
            continue;
        }
        if (P.move_all) {
            move_everything(tp);
            stage_members(sp0, 0, k);
        } else {
            move_members(sp0, 0, k, tp);  // smove_forward!(G, i, ...), :82
        }
        const double ucoin = pdmp_u01(seed, PDMP_STREAM_MAIN, nm);  // thinning coin: its index is known before the gradient is
        if (local && g_uniform((__hip_atomic_load(rnw + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0.0) ? 1u : 0u)) {
            // src/local.jl:36-43: the bound of i expired -- renew it from the moved state (one draw), no proposal
            rebound(cp0, self, self + 1u, tp, nm, false, false);
            nm += 1;
            requeue(cp0, self, self + 1u, false, 0u);
            G_ORDER();
            continue;
        }
        GPHASE(1);
        // ---------------- gradient
        double g;
        if (Q.target_kind == 0) {  // ∇ϕ(x, i) = idot(Γt, i, x) [- idot(Γt, i, μt)]
            g = 0.0;
            for (uint32_t p = 0; p < k; ++p) g += P.tb.tval[cp0 + p] * sx[p];
            if (P.tb.gmu_t) g = g - P.tb.gmu_t[i];
        } else {
            g = logistic_gradient();
        }
        GPHASE(2);
        const double th_i = sth[self];
        const double l_rate = boom ? g_pos((g - (sx[self] - Q.mu[i]) * Q.diag[i]) * th_i)  // src/fact_samplers.jl:37-39
                                   : g_pos(g * th_i);                                        // :119
        const double lbound = g_pos(a_i + b_i * (tp - told_i));     // :119
        num += 1;
        nm += 1;  // the coin is draw nm (taken above, before the gradient), :121
        const bool accept = (ucoin * lbound < l_rate);
        bool violated = false;
        if (accept) {
            nacc += 1;
            violated = (l_rate >= lbound);  // :123
            if (violated && !adapt) {
                status = PDMP_CHAIN_BOUND_VIOLATED;
                break;
            }
            if (violated && lane == 0) cmut[i] = cvec[i] * P.factor;  // adapt!(c, i, factor), :127
            if (P.move_all) stage_members(sp0, k, m);
            else move_members(sp0, k, m, tp);                         // smove_forward!(G2, i, ...), :129
            if (lane == 0) {
                sth[self] = -th_i;  // reflect!, :130
                H(i)->th = -th_i;
                rec[i].acc = acc_i + 1;
            }
            G_ORDER();
        }
        GPHASE(3);
        // ---------------- re-bound: all of G1[i] on accept (:131-135), i alone on reject (:137-139)
        const uint32_t jj0 = accept ? 0u : self;
        const uint32_t jj1 = accept ? k : self + 1u;
        reb_count = 0;
        rebound(cp0, jj0, jj1, tp, nm, accept, false);
        nm += Q.masked ? (uint64_t)reb_count : (accept ? (uint64_t)k : 1u);
        GPHASE(4);
        // ---------------- level 1 of the queue (keys[] already hold the new values)
        requeue(cp0, jj0, jj1, false, 0u);
        GPHASE(5);
        if (accept) {
            if (ev && lane == 0) {
                pdmp_event e;
                e.t = tp;
                e.i = (int64_t)i;
                e.x = sx[self];
                e.theta = -th_i;
                ev[ntrace] = e;
            }
            ntrace += 1;
            nevents += 1;
            t_event = tp;
            if (!stop_before && !(tp < T)) running = false;
        }
        G_ORDER();
        GPHASE(6);
    }
    if (PROF && chain == 0 && lane == 0 && P.dbg) {
        for (int q = 0; q < 8; ++q) P.dbg[q] = (double)ph[q];
        P.dbg[10] = (double)(num - hdr->c.num);
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

// the moving halves of the records to a packed array and back (one pass over the state each way: ~1 % of a C5 slice)
__global__ __launch_bounds__(256) void zz_hot_split_kernel(const ZzRec* __restrict__ rec, double* __restrict__ hot, int64_t n) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const double2 a = *reinterpret_cast<const double2*>(&rec[k].x), b = *reinterpret_cast<const double2*>(&rec[k].t);
    reinterpret_cast<double2*>(hot)[2 * k] = a;
    reinterpret_cast<double2*>(hot)[2 * k + 1] = b;
}
__global__ __launch_bounds__(256) void zz_hot_merge_kernel(ZzRec* __restrict__ rec, const double* __restrict__ hot, int64_t n) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    *reinterpret_cast<double2*>(&rec[k].x) = reinterpret_cast<const double2*>(hot)[2 * k];
    *reinterpret_cast<double2*>(&rec[k].t) = reinterpret_cast<const double2*>(hot)[2 * k + 1];
}

int launch_zz_general_run(const ZzRunParams& p, const ZzGeneralParams& q_in, int64_t nchains, void* stream) {
    ZzGeneralParams q = q_in;
    size_t lds = zz_general_lds_bytes(p.nblk_pad, q.mmax_pad, q.flow_kind == 1);
    const bool prof = p.dbg != nullptr;
    const bool lgfast = !prof && !q.masked && q.target_kind == 1 && q.ksub == 10 && q.lg_ne_max <= 6 && !p.move_all && !p.has_refresh && !q.local_bound && !q.sticky &&
                        q.flow_kind == 0 && !q.adaptscale;
    const bool rng = !prof && !lgfast && q.target_kind == 1 && q.lg_range > 0;
    if (rng) lds += zz_general_ranged_lds_bytes(q.ksub);
    const void* fn = prof ? reinterpret_cast<const void*>(zz_general_run_kernel<true, false, false>)
                   : lgfast ? reinterpret_cast<const void*>(zz_general_run_kernel<false, true, false>)
                   : rng ? reinterpret_cast<const void*>(zz_general_run_kernel<false, false, true>)
                         : reinterpret_cast<const void*>(zz_general_run_kernel<false, false, false>);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    const dim3 grid((unsigned)nchains), block(64);
    // the ranged sweeps run on the split state (see H() in the kernel); every other instantiation keeps the records as they are
    const bool split = rng && q.hot != nullptr;
    if (!split) q.hot = nullptr;
    const int64_t nrec = nchains * p.d;
    if (split) {
        hipLaunchKernelGGL(zz_hot_split_kernel, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p.rec, q.hot, nrec);
        hipLaunchKernelGGL((zz_general_run_kernel<false, false, true>), grid, block, lds, (hipStream_t)stream, p, q);
        hipLaunchKernelGGL(zz_hot_merge_kernel, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p.rec, q.hot, nrec);
        return (int)hipGetLastError();
    }
    if (prof) hipLaunchKernelGGL((zz_general_run_kernel<true, false, false>), grid, block, lds, (hipStream_t)stream, p, q);
    else if (lgfast) hipLaunchKernelGGL((zz_general_run_kernel<false, true, false>), grid, block, lds, (hipStream_t)stream, p, q);
    else if (rng) hipLaunchKernelGGL((zz_general_run_kernel<false, false, true>), grid, block, lds, (hipStream_t)stream, p, q);
    else hipLaunchKernelGGL((zz_general_run_kernel<false, false, false>), grid, block, lds, (hipStream_t)stream, p, q);
    return (int)hipGetLastError();
}

}  // namespace pdmp
