// pdmp_logrows.hip -- zz_logistic_rows_kernel<W>: the local ZigZag on the subsampled logistic target (config C4) with 64 / W CHAINS PER
// WAVEFRONT, each in a row of W = 16 or 32 lanes, every chain's state resident in LDS as in zz_logistic_lds_kernel (pdmp_logistic.hip).
//
// Why: a proposal of this sampler is ~630 vector instructions in which at most a dozen lanes do something useful (k_sub = 10 sampled
// observations, k ≈ 6 neighbours) -- one chain per wavefront spends the SIMDs on idle lanes (74–85 % VALU-busy at 0.13 of the roofline).  The
// independent units that can fill those lanes are the CHAINS: the LDS holds 12 of them per CU however they are spread over wavefronts, so
// 3 wavefronts of 4 chains (or 6 of 2) execute each instruction once for 4 (2) proposals.  Every quantity that is wave-uniform in the
// one-chain kernel (the popped coordinate, its time, the table header, the counters) is uniform within a ROW here and lives in vector
// registers; rows diverge where their chains do (rejected / accepted, run finished).  Cross-lane traffic stays inside rows: DPP butterflies
// for the queue's minimum, ds_bpermute for broadcasts, LDS scratch per row for the ordered sums -- and every such operation sits where its
// whole row is active.  The arithmetic, draw order and summation orders are those of zz_logistic_lds_kernel, zz_general_run_kernel<LGFAST>
// and the oracle: bit-identical (tests/test_gpu_general_parity.py, tests/test_gpu_configs_fullwidth.py).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/pdmp_detmath.h"
#include "pdmp_engine.hpp"

namespace pdmp {

#define R_INF __builtin_inf()
#define R_ORDER()                        \
    do {                                 \
        __builtin_amdgcn_wave_barrier(); \
        asm volatile("" ::: "memory");   \
    } while (0)

namespace {

template <int CTRL>
__device__ __forceinline__ double r_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double r_perm(double v, uint32_t srclane) {  // value of lane srclane (which must be active)
    const int lo = __builtin_amdgcn_ds_bpermute((int)(srclane << 2), __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute((int)(srclane << 2), __double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ uint32_t r_perm_u32(uint32_t v, uint32_t srclane) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(srclane << 2), (int)v);
}
__device__ __forceinline__ double r_pos(double x) {
    return (x > 0.0) ? x : ((x != x) ? x : 0.0);
}
__device__ __forceinline__ double r_poisson_time_L(double a, double b, double L) {  // src/poissontime.jl:8-30 with L = log(u)
    if (b == 0) return (a > 0) ? (-L / a) : R_INF;
    const double r = a / b;
    const double q = L * 2.0 / b;
    const double sq = sqrt((b > 0 && a < 0) ? -q : r * r - q);
    if (b > 0) return sq - r;
    if (a <= 0) return R_INF;
    if (-L <= -(a * a) / b + (a * a) / (2 * b)) return -sq - r;
    return R_INF;
}
__device__ __forceinline__ double r_sigmoid(double x) {  // sigmoid(x) = inv(one(x) + exp(-x)), scripts/logistic.jl:33
    return 1.0 / (1.0 + pdmp_exp(-x));
}

// minimum of (key, index) pairs over a row of W lanes, lowest index on exactly equal keys; result in every lane of the row
template <int W>
__device__ __forceinline__ void r_row_argmin(double& key, uint32_t& idx, int lane) {
#define R_STEP(CTRL)                                                                            \
    do {                                                                                        \
        const double k2 = r_dpp<CTRL>(key);                                                     \
        const uint32_t i2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)idx, CTRL, 0xf, 0xf, true); \
        const bool take = (k2 < key) || (k2 == key && i2 < idx);                                \
        key = take ? k2 : key;                                                                  \
        idx = take ? i2 : idx;                                                                  \
    } while (0)
    R_STEP(0xB1);   // quad_perm [1,0,3,2]
    R_STEP(0x4E);   // quad_perm [2,3,0,1]
    R_STEP(0x141);  // row_half_mirror
    R_STEP(0x140);  // row_mirror: every lane of a 16-lane row now holds the row's minimum
#undef R_STEP
    if constexpr (W == 32) {
        const double k2 = r_perm(key, (uint32_t)lane ^ 16u);
        const uint32_t i2 = r_perm_u32(idx, (uint32_t)lane ^ 16u);
        const bool take = (k2 < key) || (k2 == key && i2 < idx);
        key = take ? k2 : key;
        idx = take ? i2 : idx;
    }
}

}  // namespace

template <int W>
constexpr int rows_kreg() {
    return 448 / W;  // event times per lane: coordinate j lives in lane j % W of its row, slot j / W (d <= 448)
}

size_t zz_logistic_rows_lds_bytes(int64_t d, int W, bool with_I) {  // per WAVEFRONT: 64 / W chains
    const size_t per_chain = (size_t)d * (with_I ? 32 : 24) + (size_t)2 * W * 8 + (size_t)W * 4 + 16;
    return (size_t)(64 / W) * ((per_chain + 15) & ~(size_t)15);
}

template <int W, bool WITH_I>
__global__ __launch_bounds__(64) void zz_logistic_rows_kernel(ZzRunParams P, ZzGeneralParams Q, ZzLogisticTables LT, int64_t nchains) {
    constexpr int G = 64 / W;
    constexpr int KREG = rows_kreg<W>();
    const int lane = threadIdx.x;
    const uint32_t rl = (uint32_t)lane & (uint32_t)(W - 1);  // lane inside the row
    const uint32_t rbase = (uint32_t)lane & ~(uint32_t)(W - 1);
    const int row = lane / W;
    const int64_t chain = (int64_t)blockIdx.x * G + row;
    if (chain >= nchains) return;  // (whole rows leave)
    const uint32_t d = (uint32_t)P.d;
    const uint32_t dk = (uint32_t)P.dk;

    extern __shared__ __align__(16) unsigned char smem[];
    const size_t per_chain = (((size_t)d * (WITH_I ? 32 : 24) + (size_t)2 * W * 8 + (size_t)W * 4 + 16) + 15) & ~(size_t)15;
    unsigned char* const base = smem + (size_t)row * per_chain;
    double2* const xt = reinterpret_cast<double2*>(base);                  // [d] (x_j, θ_j)
    double* const tt = reinterpret_cast<double*>(base + (size_t)d * 16);   // [d] t_j
    double* const px = tt + d;                                             // [W] products Γ[r, j] x_r of one chunk; new keys
    double* const pt = px + W;                                             // [W] products Γ[r, j] θ_r
    uint32_t* const pj = reinterpret_cast<uint32_t*>(pt + W);              // [W] coordinates of the new keys
    double* const II = reinterpret_cast<double*>(pj + W + 2);              // [d] ∫ x_j up to t_j (WITH_I); (+2: keeps 8-byte alignment for any W)

    ZzRec* const rec = P.rec + chain * (int64_t)d;
    double* const keys = P.keys + chain * P.dk;
    DevChain* const hdr = P.hdr + chain;
    pdmp_event* const ev = P.ev ? P.ev + chain * P.trace_cap : nullptr;
    double* const cmut = P.c_chain ? (P.c_chain + chain * (int64_t)d) : nullptr;
    const double* const cvec = cmut ? cmut : P.tb.c_shared;

    uint32_t status = hdr->c.status;
    if (status == PDMP_CHAIN_BOUND_VIOLATED || status == PDMP_CHAIN_STALLED) return;
    const uint64_t seed = hdr->seed;
    uint64_t nm = hdr->c.ndraw_main, ng = hdr->c.ndraw_global;
    uint64_t num = hdr->c.num, nacc = hdr->c.nacc, ntrace = hdr->c.ntrace, nevents = hdr->c.nevents;
    double t_last = hdr->c.t_last;
    double t_event = hdr->t_event;
    status = PDMP_CHAIN_OK;
    const double T = P.T;
    const bool stop_before = (P.flags & PDMP_RUN_STOP_BEFORE) != 0;
    const bool adapt = P.adapt != 0;
    const uint32_t nq = (uint32_t)Q.ksub;  // sampled observations per gradient, one per lane (nq + 2 <= W)

    // ---------------- the chain's state comes on chip
    for (uint32_t j = rl; j < d; j += W) {
        const ZzRec* r = rec + j;
        xt[j] = make_double2(r->x, r->th);
        tt[j] = r->t;
        if (WITH_I) II[j] = r->I;
    }
    // the queue: the key array itself, in registers -- coordinate j in row lane j % W, slot j / W (padding and the refresh slot: +Inf)
    double kreg[KREG];
#pragma unroll
    for (int q = 0; q < KREG; ++q) {
        const uint32_t j = rl + (uint32_t)W * (uint32_t)q;
        kreg[q] = (j < dk) ? keys[j] : R_INF;
    }
    auto set_key = [&](uint32_t j, double key) {  // (j the same in the whole row or not: the owner lane takes it)
        const bool mine = (j & (uint32_t)(W - 1)) == rl;
        const uint32_t slot = j / (uint32_t)W;
#pragma unroll
        for (int q = 0; q < KREG; ++q) kreg[q] = (mine && slot == (uint32_t)q) ? key : kreg[q];
    };
    R_ORDER();

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
    // s1 += px[z0 .. z1), s2 += pt[z0 .. z1) in order
    auto run_sums = [&](uint32_t z0, uint32_t z1, double& s1, double& s2) {
        for (uint32_t z = z0; z < z1; ++z) {
            s1 += px[z];
            s2 += pt[z];
        }
    };

    bool running = stop_before || (t_event < T);
    while (running) {
        if (P.trace_cap > 0 && ntrace >= (uint64_t)P.trace_cap) {
            status = PDMP_CHAIN_TRACE_FULL;
            break;
        }
        // ---------------- peek(Q), src/sfact.jl:77: the minimum of the key array, lowest coordinate on exact ties
        double tp = kreg[0];
        uint32_t i = rl;
#pragma unroll
        for (int q = 1; q < KREG; ++q) {
            const bool lt = kreg[q] < tp;  // (strict: the lower coordinate keeps an exact tie)
            tp = lt ? kreg[q] : tp;
            i = lt ? (rl + (uint32_t)W * (uint32_t)q) : i;
        }
        r_row_argmin<W>(tp, i, lane);
        if (!(tp < R_INF)) {
            status = PDMP_CHAIN_STALLED;
            break;
        }
        if (stop_before && !(tp < T)) break;
        t_last = tp;
        const LgCoord H = LT.coord[i];
        // ---------------- every random number of the iteration, one Philox evaluation: row lane q < k_sub -> draw ng + q of the global-rng stream
        // (rand(sampler), scripts/logistic.jl:84); lane k_sub -> the thinning coin, draw nm (:121); lane k_sub + 1 + r -> draw nm + 1 + r, the
        // uniform of the r-th re-bound of this proposal (r = 0: the rejected proposal's own, :139; r < k: the members of an accepted one, :134)
        const bool qa = rl < nq;
        const uint64_t bits = pdmp_bits64(seed, qa ? PDMP_STREAM_GLOBAL : PDMP_STREAM_MAIN, qa ? (ng + (uint64_t)rl) : (nm + (uint64_t)(rl - nq)));
        const double udraw = pdmp_bits_to_u01(bits);
        const double Lmem = pdmp_log(udraw);
        const double ucoin = r_perm(udraw, rbase + nq);
        const double Lrej = r_perm(Lmem, rbase + nq + 1u);
        const uint32_t cp0 = H.cp0, k = H.k, sp0 = H.sp0, m = H.m;
        const uint32_t rdraw = (uint32_t)(((bits >> 32) * (uint64_t)H.l) >> 32);  // pdmp_randint
        const uint32_t ii = H.r0 + (qa ? rdraw : 0u);
        const uint32_t orow = LT.a_row[ii];
        const double v = LT.a_val[ii];
        const LgObs* const ob = LT.obs + orow;
        const double4 c0 = *reinterpret_cast<const double4*>(&ob->y);  // y, ny, sn0, ns0
        const double4 w0 = *reinterpret_cast<const double4*>(&ob->val[0]);
        const double2 w1 = *reinterpret_cast<const double2*>(&ob->val[4]);
        const uint4 ix = *reinterpret_cast<const uint4*>(&ob->idx[0]);  // idx[0..5], ne, pad
        const double gmu_i = P.tb.gmu_b[i];
        const ZzRec* const ri = rec + i;
        const double told_i = ri->t_old, a_i = ri->a, b_i = ri->b;
        const uint64_t acc_i = ri->acc;
        const double c_i = cvec[i];
        // ---------------- smove_forward!(G, i, ...), :82, and with it the sums of i's own re-bound: Γ[:,i]·x, Γ[:,i]·θ in idot's order
        double s1r = 0.0, s2r = 0.0;
        for (uint32_t mb = 0; mb < k; mb += W) {
            R_ORDER();
            const uint32_t pp = mb + rl;
            if (pp < k) {
                const double2 nx = move1(P.tb.sidx[sp0 + pp], tp);
                const double w = P.tb.bval[cp0 + pp];
                px[rl] = w * nx.x;
                pt[rl] = w * nx.y;
            }
            R_ORDER();
            run_sums(0u, (k - mb < (uint32_t)W) ? (k - mb) : (uint32_t)W, s1r, s2r);  // (every lane of the row the same sums: LDS broadcasts)
        }
        R_ORDER();
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
                R_ORDER();
            }
            const double w = H.lk * v;  // l / k * vals[i]
            const double t1 = w * c0.x * r_sigmoid(-u);    // sigmoidn(u) = sigmoid(-u)
            const double t2 = w * c0.y * (-r_sigmoid(u));  // nsigmoid(u) = -sigmoid(u)
            const double t3 = w * c0.x * c0.z;             // sigmoidn(u0), u0 = idot(At, row, μ): tabulated per observation
            const double t4 = w * c0.y * c0.w;             // nsigmoid(u0)
            double s = 0.0;
            for (uint32_t z = 0; z < nq; ++z) {  // (in the order of the draws, scripts/logistic.jl:84-92)
                s += r_perm(t1, rbase + z);
                s += r_perm(t2, rbase + z);
                s -= r_perm(t3, rbase + z);
                s -= r_perm(t4, rbase + z);
            }
            ng += (uint64_t)Q.ksub;
            g = prior - s;
        }
        const double th_i = xt[i].y;
        const double l_rate = r_pos(g * th_i);                   // :119
        const double lbound = r_pos(a_i + b_i * (tp - told_i));  // :119
        num += 1;
        nm += 1;  // the coin is draw nm, :121
        const bool accept = (ucoin * lbound < l_rate);
        if (!accept) {
            // ---------------- rejected (:137-139): the bound from the sums taken above
            const double a = c_i + (s1r - gmu_i) * th_i;  // src/fact_samplers.jl:51
            const double b = c_i / 100 + th_i * s2r;      // :52
            const double key = tp + r_poisson_time_L(a, b, Lrej);
            if (rl == 0) {
                ZzRec* r = rec + i;
                r->t_old = tp;
                r->a = a;
                r->b = b;
            }
            set_key(i, key);
            nm += 1;
            R_ORDER();
            continue;
        }
        // ---------------- accepted
        nacc += 1;
        double ci_new = c_i;
        const bool violated = l_rate >= lbound;  // :123
        if (violated && !adapt) {
            status = PDMP_CHAIN_BOUND_VIOLATED;
            break;
        }
        if (violated) {
            ci_new = c_i * P.factor;  // adapt!(c, i, factor), :127
            if (rl == 0) cmut[i] = ci_new;
        }
        // smove_forward!(G2, i, ...), :129
        for (uint32_t mb = k; mb < m; mb += W) {
            const uint32_t pp = mb + rl;
            if (pp < m) (void)move1(P.tb.sidx[sp0 + pp], tp);
        }
        R_ORDER();
        if (rl == 0) {
            xt[i].y = -th_i;  // reflect!, :130
            rec[i].acc = acc_i + 1;
        }
        R_ORDER();
        // ---------------- ab + new event time of every member of G1[i] (:131-135; src/fact_samplers.jl:50-54).  The dot products keep idot's
        // order (ascending row); their products are formed W at a time by the row's lanes, then every lane adds up the run of its member.
        for (uint32_t mb = 0; mb < k; mb += W) {
            const uint32_t jj = mb + rl;
            const bool valid = jj < k;
            const uint4 mrec = Q.member[cp0 + (valid ? jj : (k - 1u))];
            const uint32_t j = mrec.x;
            const uint32_t kj = valid ? mrec.y : 0u;
            const uint32_t q0 = mrec.z;
            const uint32_t last = (mb + (uint32_t)W < k) ? (mb + (uint32_t)W) : k;
            const uint32_t qs = P.tb.qptr[cp0 + mb], qe = P.tb.qptr[cp0 + last];
            const double cj_tab = cvec[valid ? j : i];
            const double cj = (j == i) ? ci_new : cj_tab;  // (c_i as adapted by THIS proposal travels in a register)
            const double gmu = P.tb.gmu_b[valid ? j : i];
            // draw nm + jj (nm already counts the coin): taken at the top of the iteration where the row's lanes reach, else now
            const uint32_t src = nq + 1u + jj;
            double Ldraw = r_perm(Lmem, rbase + ((src < (uint32_t)W) ? src : (uint32_t)(W - 1)));
            if (valid && src >= (uint32_t)W) Ldraw = pdmp_log(pdmp_u01(seed, PDMP_STREAM_MAIN, nm + (uint64_t)jj));
            double s1 = 0.0, s2 = 0.0;
            for (uint32_t cb = qs; cb < qe; cb += W) {
                const uint32_t ce = (cb + (uint32_t)W < qe) ? (cb + (uint32_t)W) : qe;
                R_ORDER();
                if (cb + rl < ce) {
                    const uint32_t rc = LT.qrow16[cb + rl];
                    const double wc = Q.qbval[cb + rl];
                    const double2 a = xt[rc];
                    px[rl] = wc * a.x;
                    pt[rl] = wc * a.y;
                }
                R_ORDER();
                const uint32_t z0 = (q0 > cb) ? q0 : cb, z1 = (q0 + kj < ce) ? (q0 + kj) : ce;
                if (z0 < z1) run_sums(z0 - cb, z1 - cb, s1, s2);
            }
            R_ORDER();
            if (valid) {
                const double thj = xt[j].y;
                const double a = cj + (s1 - gmu) * thj;  // src/fact_samplers.jl:51
                const double b = cj / 100 + thj * s2;    // :52
                const double keyj = tp + r_poisson_time_L(a, b, Ldraw);
                ZzRec* r = rec + j;
                r->t_old = tp;
                r->a = a;
                r->b = b;
                px[rl] = keyj;
                pj[rl] = j;
            }
            R_ORDER();
            // the new keys go to their owner lanes
            for (uint32_t z = 0; z < last - mb; ++z) set_key(pj[z], px[z]);
            R_ORDER();
        }
        nm += (uint64_t)k;
        if (ev && rl == 0) {
            pdmp_event e;
            e.t = tp;
            e.i = (int64_t)i;
            e.x = xt[i].x;
            e.theta = -th_i;
            ev[ntrace] = e;
        }
        ntrace += 1;
        nevents += 1;
        t_event = tp;
        if (!stop_before && !(tp < T)) running = false;
        R_ORDER();
    }
    // ---------------- the state goes back (every other entry point reads the records)
    R_ORDER();
    for (uint32_t j = rl; j < d; j += W) {
        const double2 a = xt[j];
        ZzRec* r = rec + j;
        r->x = a.x;
        r->th = a.y;
        r->t = tt[j];
        if (WITH_I) r->I = II[j];
    }
#pragma unroll
    for (int q = 0; q < KREG; ++q) {
        const uint32_t j = rl + (uint32_t)W * (uint32_t)q;
        if (j < dk) keys[j] = kreg[q];
    }
    if (rl == 0) {
        hdr->c.t_last = t_last;
        hdr->t_event = t_event;
        hdr->c.num = num;
        hdr->c.nacc = nacc;
        hdr->c.ntrace = ntrace;
        hdr->c.nevents = nevents;
        hdr->c.ndraw_main = nm;
        hdr->c.ndraw_global = ng;
        hdr->c.status = status;
    }
}

bool zz_logistic_rows_supported(const ZzRunParams& p, const ZzGeneralParams& q, const ZzLogisticTables& lt, int W) {
    return (W == 16 || W == 32) && zz_logistic_lds_supported(p, q, lt) && p.dk <= 448 && (int)q.ksub + 2 <= W && p.dbg == nullptr &&
           zz_logistic_rows_lds_bytes(p.d, W, true) <= 64 * 1024;
}

int launch_zz_logistic_rows(const ZzRunParams& p, const ZzGeneralParams& q, const ZzLogisticTables& lt, bool with_I, int W, int64_t nchains,
                            void* stream) {
    const size_t lds = zz_logistic_rows_lds_bytes(p.d, W, with_I);
    const int G = 64 / W;
    const dim3 grid((unsigned)((nchains + G - 1) / G)), block(64);
    if (W == 16) {
        if (with_I) hipLaunchKernelGGL((zz_logistic_rows_kernel<16, true>), grid, block, lds, (hipStream_t)stream, p, q, lt, nchains);
        else hipLaunchKernelGGL((zz_logistic_rows_kernel<16, false>), grid, block, lds, (hipStream_t)stream, p, q, lt, nchains);
    } else {
        if (with_I) hipLaunchKernelGGL((zz_logistic_rows_kernel<32, true>), grid, block, lds, (hipStream_t)stream, p, q, lt, nchains);
        else hipLaunchKernelGGL((zz_logistic_rows_kernel<32, false>), grid, block, lds, (hipStream_t)stream, p, q, lt, nchains);
    }
    return (int)hipGetLastError();
}

}  // namespace pdmp
