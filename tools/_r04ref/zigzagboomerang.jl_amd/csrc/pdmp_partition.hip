// pdmp_partition.hip -- zz_partitioned_run_kernel: ONE chain advanced by K wavefronts.
//
// The GPU analogue of the reference's multithreaded local ZigZag, parallel_spdmp (src/parallel.jl:104-253): the coordinates are cut into K
// chunks of d / K (Partition, :26), every chunk has its own queue and its own worker, and a coordinator handles the coordinates whose
// neighbourhood G[i] leaves their chunk.  Here a chain is a WORKGROUP of K wavefronts:
//   * worker phase -- wave ti runs parallel_spdmp_inner! (:63-102) on chunk ti: peek the chunk's queue; an inner coordinate whose time is
//     within the horizon tnext is proposed right away (parallel_innermost!, :34-61, Philox stream 16 + ti); anything else parks the worker with
//     (i, t′, acc, num) and tnext = t′ + Δ.  Workers touch inner coordinates only, whose G, G1 and G2 lie inside the chunk: no two waves share data.
//   * coordinator phase (after a workgroup barrier) -- wave 0 runs one round of parallel_spdmp_outer! (:176-253): the parked chunks in the order
//     of their times (insertion sort of the permutation, kept between rounds), each head proposed by the coordinator (stream 15) unless a
//     neighbouring chunk is still behind it (waitfor); tmin = min t′; the chunks that are not waiting are woken.
// The scheme is deterministic (the workers of one round are data-independent), so the result is checked BIT FOR BIT against the oracle's
// restatement with threads (oracle/pdmp_oracle.c orc_parallel_spdmp): tests/test_gpu_partitioned.py.  The arithmetic is the moving evaluation
// of spdmp_inner! (smove_forward!, idot in ascending row order, ab, poisson_time) written out with per-lane serial sums.
//
// The queue of a chunk: its keys in HBM (the ensemble's key array, chunk ti = keys[ti k .. ti k + k)), the minima of its 64-key blocks in LDS;
// a changed key re-scans its block.  Exactly tied keys pop lowest index first (the reference: heap order) -- probability zero, as everywhere.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/pdmp_detmath.h"
#include "pdmp_engine.hpp"

namespace pdmp {

#define Q_INF __builtin_inf()
#define Q_ORDER()                        \
    do {                                 \
        __builtin_amdgcn_wave_barrier(); \
        asm volatile("" ::: "memory");   \
    } while (0)

namespace {

__device__ __forceinline__ double q_readlane(double v, int srclane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double q_wave_min(double v) {
    for (int off = 32; off >= 1; off >>= 1) {
        const double o = __shfl_xor(v, off, 64);
        v = (o < v) ? o : v;
    }
    return v;
}
__device__ __forceinline__ double q_pos(double x) {  // src/common.jl:8
    return (x > 0.0) ? x : ((x != x) ? x : 0.0);
}
__device__ __forceinline__ double q_poisson_time(double a, double b, double u) {  // src/poissontime.jl:8-30
    const double L = pdmp_log(u);
    if (b > 0) {
        const double r = a / b;
        if (a < 0) return sqrt(-L * 2.0 / b) - r;
        return sqrt(r * r - L * 2.0 / b) - r;
    } else if (b == 0) {
        return (a > 0) ? -L / a : Q_INF;
    } else {
        if (a <= 0) return Q_INF;
        if (-L <= -(a * a) / b + (a * a) / (2 * b)) {
            const double r = a / b;
            return -sqrt(r * r - L * 2.0 / b) - r;
        }
        return Q_INF;
    }
}
// loads / stores of data that another lane or wave of the WORKGROUP wrote (all of a workgroup's waves share one vector L1: workgroup scope
// keeps them cached; nothing outside the workgroup touches a chain during a launch)
__device__ __forceinline__ double q_ld(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void q_st(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

}  // namespace

// LDS: [K nbc] f64 block minima | [K] f64 ret_t, tpr, evtime | [K] u64 ret_acc, ret_num | [K] i32 ret_i, waitfor, perm, wake | scalars
__global__ __launch_bounds__(1024) void zz_partitioned_run_kernel(ZzPartParams P) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int K = P.K;
    const int64_t chain = blockIdx.x;
    const int64_t d = P.d, k = P.k;
    const uint32_t nbc = (uint32_t)P.nbc;

    extern __shared__ __align__(16) unsigned char smem[];
    double* const bk = reinterpret_cast<double*>(smem);
    double* const ret_t = bk + (size_t)K * nbc;
    double* const tpr = ret_t + K;
    double* const evtime = tpr + K;
    uint64_t* const ret_acc = reinterpret_cast<uint64_t*>(evtime + K);
    uint64_t* const ret_num = ret_acc + K;
    int32_t* const ret_i = reinterpret_cast<int32_t*>(ret_num + K);
    int32_t* const waitfor = ret_i + K;
    int32_t* const perm = waitfor + K;
    int32_t* const wake = perm + K;
    uint32_t* const sh = reinterpret_cast<uint32_t*>(wake + K);  // [0] done, [1] error, [2] trace count, [3] overflow
    uint16_t* const bi = reinterpret_cast<uint16_t*>(sh + 4);

    ZzRec* const rec = P.rec + chain * d;
    double* const keys = P.keys + chain * P.dk;
    DevChain* const hdr = P.hdr + chain;
    pdmp_event* const evout = P.ev ? P.ev + chain * P.trace_cap : nullptr;
    double* const cmut = P.c_chain ? P.c_chain + chain * d : nullptr;
    const double* const cvec = cmut ? cmut : P.tb.c_shared;
    const uint64_t seed = hdr->seed;
    const double t0 = hdr->t0;
    const double T = P.T;

    // ---- the queue of chunk ti
    auto rescan = [&](int ti, uint32_t b) {
        const uint32_t idx = b * 64u + (uint32_t)lane;
        const double kv = (idx < (uint32_t)k) ? q_ld(keys + (size_t)ti * k + idx) : Q_INF;
        const double m = q_wave_min(kv);
        const uint64_t bl = __ballot(kv == m);
        const int arg = bl ? (__ffsll((unsigned long long)bl) - 1) : 0;
        if (lane == 0) {
            bk[(size_t)ti * nbc + b] = m;
            bi[(size_t)ti * nbc + b] = (uint16_t)arg;
        }
        Q_ORDER();
    };
    auto peek = [&](int ti, uint32_t& ii, double& tp) {
        double m = Q_INF;
        uint32_t mb = 0xffffffffu;
        for (uint32_t b = (uint32_t)lane; b < nbc; b += 64u) {
            const double v = bk[(size_t)ti * nbc + b];
            if (v < m || mb == 0xffffffffu) {
                m = v;
                mb = b;
            }
        }
        const double mm = q_wave_min(m);
        uint32_t cand = (m == mm && mb != 0xffffffffu) ? mb : 0xffffffffu;
        for (int off = 32; off >= 1; off >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)cand, off, 64);
            cand = (o < cand) ? o : cand;
        }
        tp = mm;
        ii = cand * 64u + (uint32_t)bi[(size_t)ti * nbc + cand];
    };
    auto push_event = [&](double t, int64_t i, double x, double th) {
        if (lane == 0 && evout) {
            const uint32_t slot = atomicAdd(&sh[2], 1u);
            if ((int64_t)slot < P.trace_cap) {
                pdmp_event e;
                e.t = t;
                e.i = i;
                e.x = x;
                e.theta = th;
                evout[slot] = e;
            } else {
                sh[3] = 1u;
            }
        }
    };

    // ---- parallel_innermost! (src/parallel.jl:34-61) on the 64 lanes of one wave; returns 1 (accepted), 0, or -1 (bound violated, adapt off)
    auto innermost = [&](int64_t i, double tp, uint32_t stream, uint64_t& nd) -> int {
        const int ti = (int)(i / k);
        const uint32_t cp = P.tb.colptr[i], kG = P.tb.colptr[i + 1] - cp;
        const bool mG = (uint32_t)lane < kG;
        const uint32_t j = mG ? P.tb.rowval[cp + lane] : (uint32_t)i;
        ZzRec* const rj = rec + j;
        double xn = 0.0, thj = 0.0;
        if (mG) {  // smove_forward!(G, i, t, x, θ, t′, F), src/sfact.jl:6-16
            const double xj = q_ld(&rj->x), tj = q_ld(&rj->t);
            thj = q_ld(&rj->th);
            const double dt = tp - tj;
            xn = xj + thj * dt;
            q_st(&rj->I, q_ld(&rj->I) + dt * ((xj + xn) * 0.5));
            q_st(&rj->x, xn);
            q_st(&rj->t, tp);
        }
        // ∇ϕ(x, i) = idot(Γt, i, x) (− (Γt μt)_i): products side by side, summed in ascending row order
        const double pr = mG ? P.tb.tval[cp + lane] * xn : 0.0;
        double s = 0.0;
        for (uint32_t q = 0; q < kG; ++q) s += q_readlane(pr, (int)q);
        double gi = s;
        if (P.tb.gmu_t) gi = gi - P.tb.gmu_t[i];
        const uint64_t selfb = __ballot(mG && j == (uint32_t)i);
        const int sp = selfb ? (__ffsll((unsigned long long)selfb) - 1) : 0;
        double th_i = q_readlane(thj, sp);
        const double x_i = q_readlane(xn, sp);
        const ZzRec* const ri = rec + i;
        const double a_i = q_ld(&ri->a), b_i = q_ld(&ri->b), told_i = q_ld(&ri->t_old);
        const double l = q_pos(gi * th_i);
        const double lb = q_pos(a_i + b_i * (tp - told_i));
        const double u = pdmp_u01(seed, stream, nd);
        nd += 1;
        const bool accept = u * lb < l;
        bool m1;  // the lanes whose member is re-bounded
        if (accept) {
            if (l >= lb) {
                if (!P.adapt) return -1;  // error("Tuning parameter `c` too small."), :42
                if (lane == 0) q_st(cmut + i, q_ld(cmut + i) * P.factor);  // adapt!(c, i, factor)
            }
            const uint32_t g2a = P.g2ptr[i], g2b = P.g2ptr[i + 1];
            for (uint32_t base = g2a; base < g2b; base += 64u) {  // smove_forward!(G2, ...)
                const uint32_t idx = base + (uint32_t)lane;
                if (idx < g2b) {
                    ZzRec* const r2 = rec + P.g2idx[idx];
                    const double x2 = q_ld(&r2->x), t2 = q_ld(&r2->t), th2 = q_ld(&r2->th);
                    const double dt = tp - t2;
                    const double xn2 = x2 + th2 * dt;
                    q_st(&r2->I, q_ld(&r2->I) + dt * ((x2 + xn2) * 0.5));
                    q_st(&r2->x, xn2);
                    q_st(&r2->t, tp);
                }
            }
            th_i = -th_i;  // reflect!
            if (lane == sp) {
                q_st(&rec[i].th, th_i);
                thj = th_i;
                rec[i].acc += 1;
            }
            m1 = mG && P.g1mask[cp + lane] != 0;
        } else {
            m1 = mG && j == (uint32_t)i;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        Q_ORDER();
        const uint64_t m1b = __ballot(m1);
        const uint32_t rk = (uint32_t)__popcll(m1b & ((1ull << lane) - 1ull));
        if (m1) {  // ab(G1, j, x, θ, c, F), src/fact_samplers.jl:50-54: two idots over column j of the bounding Γ, each lane its own serial sum
            double sx = 0.0, sth = 0.0;
            const uint32_t p0 = P.tb.colptr[j], p1 = P.tb.colptr[j + 1];
            for (uint32_t p = p0; p < p1; ++p) {
                if (!P.g1mask[p]) continue;
                const ZzRec* const rr = rec + P.tb.rowval[p];
                const double bv = P.tb.bval[p];
                sx += bv * q_ld(&rr->x);
                sth += bv * q_ld(&rr->th);
            }
            const double cj = cmut ? q_ld(cmut + j) : cvec[j];
            const double aj = cj + (sx - P.tb.gmu_b[j]) * thj;
            const double bj = cj / 100 + thj * sth;
            const double uj = pdmp_u01(seed, stream, nd + rk);
            q_st(&rj->a, aj);
            q_st(&rj->b, bj);
            q_st(&rj->t_old, tp);
            q_st(keys + j, tp + q_poisson_time(aj, bj, uj));
        }
        nd += (uint64_t)__popcll(m1b);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        Q_ORDER();
        uint32_t lastb = 0xffffffffu;
        for (uint64_t todo = m1b; todo; todo &= todo - 1) {
            const int q = __ffsll((unsigned long long)todo) - 1;
            const uint32_t jq = (uint32_t)__builtin_amdgcn_readlane((int)j, q);
            const uint32_t b = (uint32_t)((jq - (uint32_t)((int64_t)ti * k)) >> 6);
            if (b != lastb) rescan(ti, b);
            lastb = b;
        }
        if (accept) push_event(tp, i, x_i, th_i);
        return accept ? 1 : 0;
    };

    // ---- set-up: first level of every chunk, round state
    if (wave < K) {
        for (uint32_t b = 0; b < nbc; ++b) rescan(wave, b);
        if (lane == 0) {
            tpr[wave] = t0;
            evtime[wave] = 0.0;
            waitfor[wave] = 0;
            perm[wave] = wave;
            wake[wave] = 1;
            ret_t[wave] = t0;
            ret_i[wave] = 0;
            ret_acc[wave] = 0;
            ret_num[wave] = 0;
        }
    }
    if (threadIdx.x == 0) {
        sh[0] = 0;
        sh[1] = 0;
        sh[2] = 0;
        sh[3] = 0;
    }
    __syncthreads();

    double tnext = t0 + P.delta;  // :67
    uint64_t nd_w = 0, nd_outer = 0;
    uint64_t acc_tot = 0, num_tot = 0, rounds = 0;
    double tmin = t0;
    bool parked_for_good = false;
    for (;;) {
        // ---------------- worker phase (parallel_spdmp_inner!)
        if (wave < K && wake[wave] && !parked_for_good) {
            uint64_t acc = 0, num = 0;  // :96 (and :65)
            for (;;) {
                num += 1;
                uint32_t ii;
                double tp;
                peek(wave, ii, tp);
                const int64_t i = (int64_t)wave * k + ii;
                if (!P.inner[i] || tp > tnext) {  // :73
                    tnext = tp + P.delta;
                    if (lane == 0) {
                        ret_i[wave] = (int32_t)i;
                        ret_t[wave] = tp;
                        ret_acc[wave] = acc;
                        ret_num[wave] = num;
                    }
                    break;
                }
                const int ok = innermost(i, tp, 16u + (uint32_t)wave, nd_w);
                if (ok < 0) {  // the reference's task dies with error(...); the chunk parks for good and the run ends
                    if (lane == 0) {
                        sh[1] = 1u;
                        ret_i[wave] = (int32_t)i;
                        ret_t[wave] = Q_INF;
                        ret_acc[wave] = acc;
                        ret_num[wave] = num;
                    }
                    parked_for_good = true;
                    break;
                }
                if (ok) acc += 1;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __syncthreads();
        // ---------------- coordinator phase (one round of parallel_spdmp_outer!)
        if (wave == 0) {
            for (int ti = 0; ti < K; ++ti)
                if (waitfor[ti] == 0 && lane == 0) evtime[ti] = ret_t[ti];  // (the workers' events are in the trace already)
            Q_ORDER();
            if (lane == 0) {  // sortperm!(perm, evtime, alg=InsertionSort), :204
                for (int a = 1; a < K; ++a) {
                    const int v = perm[a];
                    int b = a - 1;
                    while (b >= 0 && evtime[perm[b]] > evtime[v]) {
                        perm[b + 1] = perm[b];
                        --b;
                    }
                    perm[b + 1] = v;
                }
            }
            Q_ORDER();
            for (int a = 0; a < K; ++a) {  // :206-233
                const int ti = perm[a];
                const int64_t i = ret_i[ti];
                const double tpi = ret_t[ti];
                if (waitfor[ti] == 0) {
                    num_tot += ret_num[ti];
                    acc_tot += ret_acc[ti];
                    Q_ORDER();
                    if (lane == 0) tpr[ti] = tpi;
                }
                Q_ORDER();
                const uint32_t cp = P.tb.colptr[i], kG = P.tb.colptr[i + 1] - cp;
                bool behind = false;
                if ((uint32_t)lane < kG) {
                    const int64_t j = P.tb.rowval[cp + lane];
                    behind = j != i && tpr[j / k] < tpi;
                }
                const bool wf = __ballot(behind) != 0;
                Q_ORDER();
                if (lane == 0) waitfor[ti] = wf ? (int32_t)(i + 1) : 0;
                Q_ORDER();
                if (wf) continue;
                if (!(tpi < Q_INF)) continue;  // a chunk parked for good
                const int ok = innermost(i, tpi, 15u, nd_outer);
                if (ok < 0) {
                    if (lane == 0) sh[1] = 1u;
                } else if (ok) {
                    acc_tot += 1;
                }
                Q_ORDER();
            }
            Q_ORDER();
            tmin = tpr[0];
            for (int ti = 1; ti < K; ++ti) tmin = (tpr[ti] < tmin) ? tpr[ti] : tmin;  // :235
            if (sh[1]) tmin = T;
            rounds += 1;
            const bool done = tmin >= T;
            Q_ORDER();
            if (lane == 0) {
                for (int ti = 0; ti < K; ++ti) wake[ti] = (waitfor[ti] == 0 || done) ? 1 : 0;  // :241-249
                sh[0] = done ? 1u : 0u;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __syncthreads();
        if (sh[0]) break;
    }
    if (threadIdx.x == 0) {
        const uint32_t ntr = sh[2];
        const uint64_t kept = evout ? (((int64_t)ntr < P.trace_cap) ? (uint64_t)ntr : (uint64_t)P.trace_cap) : 0;
        hdr->c.t_last = tmin;
        hdr->t_event = tmin;
        hdr->c.num = num_tot;
        hdr->c.nacc = acc_tot;
        hdr->c.nrefresh = rounds;  // (no refresh clock in this scheme: the field carries the number of coordinator rounds)
        hdr->c.ntrace = kept;
        hdr->c.nevents = evout ? (uint64_t)ntr : acc_tot;
        hdr->c.status = sh[1] ? PDMP_CHAIN_BOUND_VIOLATED : (sh[3] ? PDMP_CHAIN_TRACE_FULL : PDMP_CHAIN_OK);
    }
}

size_t zz_partitioned_lds_bytes(int K, int nbc) {
    return (size_t)K * nbc * 8 + (size_t)K * (3 * 8 + 2 * 8 + 4 * 4) + 16 + (size_t)K * nbc * 2 + 16;
}

int launch_zz_partitioned(const ZzPartParams& p, int64_t nchains, void* stream) {
    dim3 grid((unsigned)nchains), block((unsigned)(64 * p.K));
    const size_t lds = zz_partitioned_lds_bytes(p.K, p.nbc);
    hipLaunchKernelGGL(zz_partitioned_run_kernel, grid, block, lds, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

}  // namespace pdmp
