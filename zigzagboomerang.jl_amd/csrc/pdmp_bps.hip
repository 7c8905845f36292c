// pdmp_bps.hip -- Bouncy Particle Sampler ensemble on gfx950: pdmp_inner! (src/not_fact_samplers.jl:52-97) under the
// driver loop `while t < T` (:136-144), GlobalBound(c), Gaussian target ∇ϕ!(y,x) = Γ(x-μ), mass L = I.
// BOOM = true: the same loop for Flow = Boomerang(I, μ_flow, λref; ρ) (rotation, grad_correct!, constant bound).
//
// One chain per wavefront.  The d-vectors x, θ, ∇ϕ live in REGISTERS (element e = slot*64 + lane, NS slots per lane), so a
// proposal touches HBM only to emit an event: (t, copy(x), copy(θ)) = 8(2d+1) bytes, written fully coalesced
// (src/not_fact_samplers.jl:39-41).  That write stream is the roofline of this kernel.
// Dot products use ONE fixed summation order, the one oracle/pdmp_oracle.c restates (dot_wave64): per-lane partial sums
// over the slots in order, then the xor-butterfly 1,2,4,8 inside each DPP row and the four row sums ((r0+r1)+(r2+r3)).
// A general CSC Γ is applied by staging the operand vector in LDS and gathering (idot order, src/common.jl:16-24); a diagonal
// Γ (config C2: Γ = I) takes a register-only path.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "../../include/pdmp_detmath.h"
#include "pdmp_engine.hpp"

namespace pdmp {

#define BPS_INF __builtin_inf()

__device__ __forceinline__ double b_readlane(double v, int srclane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double b_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);  // every lane has a valid source: no tied `old` operand, no copies
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// all-lanes sum in the oracle's order: xor 1, 2 (quads), 4, 8 (row of 16), then (r0+r1)+(r2+r3)
__device__ __forceinline__ double wave_sum_f64(double v) {
    v = v + b_dpp<0xB1>(v);   // lane ^ 1
    v = v + b_dpp<0x4E>(v);   // lane ^ 2
    v = v + b_dpp<0x141>(v);  // other quad of the half row (same value in every lane of a quad: == lane ^ 4)
    v = v + b_dpp<0x140>(v);  // other half row (== lane ^ 8)
    const double r0 = b_readlane(v, 0), r1 = b_readlane(v, 16), r2 = b_readlane(v, 32), r3 = b_readlane(v, 48);
    return (r0 + r1) + (r2 + r3);  // lane ^ 16, then lane ^ 32
}

__device__ __forceinline__ double bps_pos(double x) {
    return (x > 0.0) ? x : ((x != x) ? x : 0.0);
}

// poisson_time(a, b, u), src/poissontime.jl:8-30
__device__ __forceinline__ double bps_poisson_time(double a, double b, double u) {
    const double L = pdmp_log(u);
    if (b > 0) {
        const double r = a / b;
        if (a < 0) return sqrt(-L * 2.0 / b) - r;
        return sqrt(r * r - L * 2.0 / b) - r;
    } else if (b == 0) {
        return (a > 0) ? (-L / a) : BPS_INF;
    } else {
        if (a <= 0) return BPS_INF;
        if (-L <= -(a * a) / b + (a * a) / (2 * b)) {
            const double r = a / b;
            return -sqrt(r * r - L * 2.0 / b) - r;
        }
        return BPS_INF;
    }
}

// IDENT: Γ = I and μ = 0 exactly (config C2, the isotropic target): ∇ϕ!(y, x) = 0 + 1·(x − 0) is x itself, so the gradient array,
// μ and the diagonal leave the register file (256 -> about 100 VGPRs at NS = 16: 1 -> 4 waves per SIMD) and Γθ = θ.
// poisson_time(a, b, u) with L = log(u) already taken (the draw's index is known before the rates are: the logarithm is
// evaluated off the critical path)
__device__ __forceinline__ double bps_poisson_time_L(double a, double b, double L) {
    if (b > 0) {
        const double r = a / b;
        if (a < 0) return sqrt(-L * 2.0 / b) - r;
        return sqrt(r * r - L * 2.0 / b) - r;
    } else if (b == 0) {
        return (a > 0) ? (-L / a) : BPS_INF;
    } else {
        if (a <= 0) return BPS_INF;
        if (-L <= -(a * a) / b + (a * a) / (2 * b)) {
            const double r = a / b;
            return -sqrt(r * r - L * 2.0 / b) - r;
        }
        return BPS_INF;
    }
}

// IDENT: Γ = I and μ = 0 exactly (config C2, the isotropic target): ∇ϕ!(y, x) = 0 + 1·(x − 0) is x itself, so the gradient array,
// μ and the diagonal leave the register file (256 -> about 100 VGPRs at NS = 16: 1 -> 4 waves per SIMD) and Γθ = θ.
// t′ − t = poisson_time(a, b, rand(rng)) behind a call as well (log polynomial, two divisions, sqrt: constants and temporaries)
__device__ __attribute__((noinline)) double bps_next_dt(uint64_t seed, uint64_t n, double a, double b) {
    return bps_poisson_time(a, b, pdmp_u01(seed, PDMP_STREAM_MAIN, n));
}

// FULL: d == 64 NS exactly, so the `element < d` guards (and their exec-mask bookkeeping) are compile-time true.
// EXT: the extended instantiation (general Γ only) adds what the fast ones leave out -- a caller-supplied mass factor L
// (reflect!, refresh!, Boomerang's grad_correct!: column-oriented substitution through LDS, the order oracle/pdmp_oracle.c
// fixes), c::LocalBound with its horizon and the renew branch (src/not_fact_samplers.jl:29-31,65-71), and `subsample` (:90).
template <int NS, bool DIAG, bool BOOM, bool IDENT, bool FULL = false, bool EXT = false>
__global__ __launch_bounds__(64) void bps_run_kernel(BpsRunParams P) {
    const int lane = threadIdx.x;
    const int64_t chain = blockIdx.x;
    const int64_t d = P.d;
    extern __shared__ __align__(16) unsigned char smem[];
    double* tmp = reinterpret_cast<double*>(smem);  // [d] operand of the CSC gather (general Γ); the normals of a refresh

    double* gx = P.x + chain * d;
    double* gth = P.th + chain * d;
    double* sc = P.scal + chain * 8;  // {t, a, b, tp, tau_ref, c}
    DevChain* hdr = P.hdr + chain;

    uint32_t status = hdr->c.status;
    if (status == PDMP_CHAIN_BOUND_VIOLATED || status == PDMP_CHAIN_STALLED) return;
    status = PDMP_CHAIN_OK;
    const uint64_t seed = hdr->seed;
    uint64_t nm = hdr->c.ndraw_main;
    uint64_t num = hdr->c.num, nacc = hdr->c.nacc, nrefresh = hdr->c.nrefresh, ntrace = hdr->c.ntrace,
             nevents = hdr->c.nevents;
    double t = sc[0], a = sc[1], b = sc[2], tp = sc[3], tau_ref = sc[4], c = sc[5];
    bool renew = EXT && sc[6] != 0.0;  // next_time's flag (src/not_fact_samplers.jl:43-50): t′ is the bound's expiry, not a proposal
    double hz = EXT ? sc[7] : BPS_INF; // abc[3]
    const bool has_mass = EXT && P.Lcp != nullptr;

    constexpr int NG = IDENT ? 1 : NS;
    double x[NS], th[NS], g[NG], mu[NG], dg[NG];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int64_t e = (int64_t)s * 64 + lane;
        const bool in = FULL || e < d;
        x[s] = in ? gx[e] : 0.0;
        th[s] = in ? gth[e] : 0.0;
        if constexpr (!IDENT) {
            mu[s] = in ? P.mu[e] : 0.0;
            dg[s] = (DIAG && in) ? P.nzval[e] : 0.0;
            g[s] = 0.0;
        }
    }
    if constexpr (IDENT) g[0] = mu[0] = dg[0] = 0.0;
    const double rho = P.rho, rhobar = sqrt(1 - rho * rho);  // src/dynamics.jl:113
    const double T = P.T;
    const bool stop_before = (P.flags & PDMP_RUN_STOP_BEFORE) != 0;

    // y = Γ v  with v = in[] (-mu if sub): idot per output element, ascending row order
    // (target = true: the TARGET's Γt, μt where the ensemble has one of its own -- extended instantiation -- else the flow's)
    auto apply_gamma = [&](const double (&in)[NS], bool sub_mu, double (&out)[NG], bool target = false) {
        if constexpr (IDENT) {
            (void)in;
            (void)sub_mu;
            (void)out;
        } else if (EXT && target && P.t_colptr) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int64_t e = (int64_t)s * 64 + lane;
                if (FULL || e < d) tmp[e] = sub_mu ? (in[s] - P.t_mu[e]) : in[s];
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int64_t e = (int64_t)s * 64 + lane;
                double y = 0.0;
                if (FULL || e < d) {
                    for (int64_t p = P.t_colptr[e]; p < P.t_colptr[e + 1]; ++p) y += P.t_nzval[p] * tmp[P.t_rowval[p]];
                }
                out[s] = y;
            }
        } else if (DIAG) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const double v = sub_mu ? (in[s] - mu[s]) : in[s];
                out[s] = 0.0 + dg[s] * v;
            }
        } else {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int64_t e = (int64_t)s * 64 + lane;
                if (FULL || e < d) tmp[e] = sub_mu ? (in[s] - mu[s]) : in[s];
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int64_t e = (int64_t)s * 64 + lane;
                double y = 0.0;
                if (FULL || e < d) {
                    for (int64_t p = P.colptr[e]; p < P.colptr[e + 1]; ++p) y += P.nzval[p] * tmp[P.rowval[p]];
                }
                out[s] = y;
            }
        }
    };
    auto dot = [&](const double (&u)[NS], const double (&v)[NS]) -> double {
        double part = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int64_t e = (int64_t)s * 64 + lane;
            if (FULL || e < d) part += u[s] * v[s];
        }
        return wave_sum_f64(part);
    };
    // ab(x, θ, C::GlobalBound, ...) = (c + θ'(Γ(x-μ)), θ'(Γθ), Inf), src/not_fact_samplers.jl:26-28, and next_time :43-50
    // Boomerang: ab(x, θ, C::GlobalBound, ...) = (sqrt(normsq(θ) + normsq(x − μ))·c, 0, Inf), src/not_fact_samplers.jl:34-36
    auto boom_a = [&]() -> double {
        double dx[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int64_t e = (int64_t)s * 64 + lane;
            dx[s] = (FULL || e < d) ? (x[s] - P.mu_flow[e]) : 0.0;
        }
        return sqrt(dot(th, th) + dot(dx, dx)) * c;
    };
    // tmp <- L \ tmp and tmp <- L' \ tmp: column-oriented substitution (one column per step, its off-diagonal entries one per
    // lane), every element updated in the order of the columns -- exactly tri_solve_lower / tri_solve_upper of the oracle.
    // One wavefront: DS operations retire in order, so a step sees the previous step's updates without a barrier.
    auto solve_lower = [&]() {
        for (int64_t j = 0; j < d; ++j) {
            asm volatile("" ::: "memory");
            const int32_t p0 = P.Lcp[j], p1 = P.Lcp[j + 1];
            const double yj = tmp[j] / P.Lnz[p0];
            asm volatile("" ::: "memory");
            if (lane == 0) tmp[j] = yj;
            for (int32_t p = p0 + 1 + lane; p < p1; p += 64) {
                const int32_t r = P.Lrv[p];
                tmp[r] = tmp[r] - P.Lnz[p] * yj;
            }
        }
        asm volatile("" ::: "memory");
    };
    auto solve_upper = [&]() {
        for (int64_t j = d - 1; j >= 0; --j) {
            asm volatile("" ::: "memory");
            const int32_t p0 = P.Ucp[j], p1 = P.Ucp[j + 1] - 1;
            const double zj = tmp[j] / P.Unz[p1];
            asm volatile("" ::: "memory");
            if (lane == 0) tmp[j] = zj;
            for (int32_t p = p0 + lane; p < p1; p += 64) {
                const int32_t r = P.Urv[p];
                tmp[r] = tmp[r] - P.Unz[p] * zj;
            }
        }
        asm volatile("" ::: "memory");
    };
    auto to_lds = [&](const double (&v)[NS]) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int64_t e = (int64_t)s * 64 + lane;
            if (FULL || e < d) tmp[e] = v[s];
        }
        asm volatile("" ::: "memory");
    };
    auto from_lds = [&](double (&v)[NS]) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int64_t e = (int64_t)s * 64 + lane;
            v[s] = (FULL || e < d) ? tmp[e] : 0.0;
        }
        asm volatile("" ::: "memory");
    };
    auto rebound = [&](double Lnext) {
        if constexpr (IDENT) {
            a = c + dot(th, x);  // θ'(Γ(x−μ)) with Γ(x−μ) = x
            b = dot(th, th);     // θ'(Γθ) with Γθ = θ
        } else if (BOOM) {
            a = boom_a();
            b = 0.0;
        } else {
            // GlobalBound: (c + θ'(B.Γ(x − B.μ)), θ'(B.Γθ)) with the FLOW's Γ, μ (:26-28) -- θ'∇ϕx only when the target is B.Γ(x − B.μ);
            // LocalBound: (c + dot(θ, ∇ϕx), v) with the TARGET's gradient and v = θ'Γtθ (:29-31)
            const bool own_target = EXT && P.t_colptr != nullptr;
            if (own_target && !P.local_bound) {
                double gb[NG];
                apply_gamma(x, true, gb);
                a = c + dot(th, gb);
            } else {
                a = c + dot(th, g);
            }
            double gt[NS];
            apply_gamma(th, false, gt, own_target && P.local_bound);
            b = dot(th, gt);
        }
        if constexpr (EXT) {
            // ab(x, θ, C::LocalBound, ∇ϕx, v, B) = (c + dot(θ, ∇ϕx), v, 2√d/c/‖θ‖₂), :29-31; next_time, :43-50
            hz = (P.local_bound && !BOOM) ? 2 * sqrt((double)d) / c / sqrt(dot(th, th)) : BPS_INF;
            const double dt = bps_poisson_time_L(a, b, Lnext);
            renew = dt > hz;
            tp = renew ? t + hz : t + dt;
        } else {
            tp = t + bps_poisson_time_L(a, b, Lnext);
        }
        nm += 1;
    };
    // move_forward!(τ, t, x, θ, Flow): linear (src/dynamics.jl:11-15) or the rotation about μ (:29-36)
    auto move = [&](double tau) {
        if (BOOM) {
            double sn, cs;
            pdmp_sincos(tau, &sn, &cs);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int64_t e = (int64_t)s * 64 + lane;
                const double m = (FULL || e < d) ? P.mu_flow[e] : 0.0;
                const double xn = (x[s] - m) * cs + th[s] * sn + m;
                const double tn = -(x[s] - m) * sn + th[s] * cs;
                x[s] = xn;
                th[s] = tn;
            }
        } else {
#pragma unroll
            for (int s = 0; s < NS; ++s) x[s] += th[s] * tau;
        }
    };
    // ∇ϕx = ∇ϕ!(∇ϕx, x); grad_correct!: Boomerang subtracts L'\(L\(x − μ)) = x − μ for L = I (src/not_fact_samplers.jl:9-12)
    auto gradient = [&]() {
        if constexpr (!IDENT) {
            apply_gamma(x, true, g, true);
            if (BOOM) {
                if (has_mass) {  // grad_correct!: y .-= L'\(L\(x − μ)), src/not_fact_samplers.jl:9-12
                    double dx[NS];
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const int64_t e = (int64_t)s * 64 + lane;
                        dx[s] = (FULL || e < d) ? (x[s] - P.mu_flow[e]) : 0.0;
                    }
                    to_lds(dx);
                    solve_lower();
                    solve_upper();
                    from_lds(dx);
#pragma unroll
                    for (int s = 0; s < NS; ++s) g[s] -= dx[s];
                } else {
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const int64_t e = (int64_t)s * 64 + lane;
                        if (FULL || e < d) g[s] -= x[s] - P.mu_flow[e];
                    }
                }
            }
        }
    };

    bool running = stop_before || (t < T);  // `while t < T`, :136
    while (running) {  // (no priority turns here: the trace writes bound this kernel, and turns cost 15 % on BASELINE.json's C2)
        if constexpr (NS > 16) {  // (the counters are wave-uniform: said so, they live in scalar registers instead of competing with 5 x NS doubles)
            auto uni = [](uint64_t v) -> uint64_t {
                return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
                       (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
            };
            num = uni(num);
            nacc = uni(nacc);
            nrefresh = uni(nrefresh);
            ntrace = uni(ntrace);
            nevents = uni(nevents);
            nm = uni(nm);
        }
        if (P.trace_cap > 0 && ntrace >= (uint64_t)P.trace_cap) {
            status = PDMP_CHAIN_TRACE_FULL;
            break;
        }
        const bool is_ref = tau_ref < tp;  // :55
        const double tnext = is_ref ? tau_ref : tp;
        if (!(tnext < BPS_INF)) {
            status = PDMP_CHAIN_STALLED;
            break;
        }
        if (stop_before && !(tnext < T)) break;
        const double tau = tnext - t;  // :56 / :73
        t += tau;
        move(tau);
        bool emit = false;
        if (is_ref) {
            // refresh!, src/dynamics.jl:112-118 with L = I: θ .*= ρ; θ .+= ρ̄ randn(rng, d)  (draw nm + e for element e)
            // The d normals go through LDS: a ROLLED loop holds one Box-Muller body (its constants and temporaries are live only
            // here, next to x and θ), the unrolled update then reads them back -- 16 inlined bodies, or a call, cost ~40 VGPRs
            // across the whole event loop (161 -> 123 at NS = 16: 3 -> 4 waves per SIMD).
            // (element 128a + 64b + lane is Box-Muller branch b of block nm + 64a + lane: one evaluation serves two slots)
            asm volatile("" ::: "memory");
#pragma unroll 1
            for (int a2 = 0; a2 < (NS + 1) / 2; ++a2) {
                const int64_t e0 = (int64_t)a2 * 128 + lane, e1 = e0 + 64;
                double z0, z1;
                pdmp_randn2(seed, PDMP_STREAM_MAIN, nm + (uint64_t)(a2 * 64 + lane), &z0, &z1);
                if (FULL || e0 < d) tmp[e0] = z0;
                if (FULL || e1 < d) tmp[e1] = z1;
            }
            asm volatile("" ::: "memory");
            if (has_mass) solve_upper();  // u = ρ̄*(L'\randn(rng, d)), src/dynamics.jl:115
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int64_t e = (int64_t)s * 64 + lane;
                th[s] *= rho;
                if (FULL || e < d) th[s] += rhobar * tmp[e];
            }
            nm += (uint64_t)(((d + 127) >> 7) << 6);
            gradient();                                                                                    // :58-59
            tau_ref = t + (-pdmp_log(pdmp_u01(seed, PDMP_STREAM_MAIN, nm)) / P.lambda_ref);                  // :61
            nm += 1;
            rebound(pdmp_log(pdmp_u01(seed, PDMP_STREAM_MAIN, nm)));  // :62-63
            nrefresh += 1;
            emit = true;  // :64
        } else if (EXT && renew) {
            // :65-71: the bound expired -- move (done above), gradient, new bound, new proposal; no thinning step, no event
            gradient();
            rebound(pdmp_log(pdmp_u01(seed, PDMP_STREAM_MAIN, nm)));
        } else {
            // both draws of a proposal have known indices (coin: nm, next_time: nm + 1 on accept and on reject alike): Philox and
            // the logarithm run beside the gradient and the reductions instead of after them
            const double coin = pdmp_u01(seed, PDMP_STREAM_MAIN, nm);
            const double Lnext = pdmp_log(pdmp_u01(seed, PDMP_STREAM_MAIN, nm + 1));
            gradient();  // :75-76
            double gt;
            if constexpr (IDENT) gt = dot(x, th);
            else gt = dot(g, th);
            const double l = bps_pos(gt);            // λ, :14
            const double lb = bps_pos(a + b * tau);  // :77
            num += 1;
            nm += 1;
            if (coin * lb <= l) {  // :79
                nacc += 1;
                if (l > lb) {  // :81
                    if (!P.adapt) {
                        status = PDMP_CHAIN_BOUND_VIOLATED;  // reference: error(...), :82
                        break;
                    }
                    c *= P.factor;  // :83
                }
                // reflect!, src/dynamics.jl:90-93 with L = I: θ .-= (2 dot(∇ϕx,θ)/normsq(∇ϕx)) ∇ϕx
                if (has_mass) {
                    // θ .-= (2 dot(∇ϕx,θ)/normsq(L\∇ϕx)) (L'\(L\∇ϕx)), src/dynamics.jl:90-93
                    if constexpr (!IDENT) {
                        double w[NS];
                        to_lds(g);
                        solve_lower();
                        from_lds(w);
                        const double nrm = dot(w, w);
                        solve_upper();
                        from_lds(w);
                        const double coef = 2 * gt / nrm;
#pragma unroll
                        for (int s = 0; s < NS; ++s) th[s] -= coef * w[s];
                    }
                } else {
                    double nrm;
                    if constexpr (IDENT) nrm = dot(x, x);
                    else nrm = dot(g, g);
                    const double coef = 2 * gt / nrm;
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        if constexpr (IDENT) th[s] -= coef * x[s];
                        else th[s] -= coef * g[s];
                    }
                }
                rebound(Lnext);  // :86-89
                emit = !(EXT && P.subsample);  // :90 `!subsample && return`
            } else {
                if (BOOM) {
                    a = boom_a();  // :92 recomputed after the rotation (b stays 0)
                } else if (EXT && P.t_colptr != nullptr && !P.local_bound) {
                    if constexpr (!IDENT) {  // :92 with a target of its own: θ'(B.Γ(x − B.μ)) is not θ'∇ϕx
                        double gb[NG];
                        apply_gamma(x, true, gb);
                        a = c + dot(th, gb);
                    }
                } else {
                    a = c + gt;  // :92 (θ'g == g'θ bit for bit; b = θ'Γθ is unchanged because θ is)
                }
                const double dt = bps_poisson_time_L(a, b, Lnext);  // :93 (the horizon is unchanged: θ and c are)
                if constexpr (EXT) {
                    renew = dt > hz;
                    tp = renew ? t + hz : t + dt;
                } else {
                    tp = t + dt;
                }
                nm += 1;
            }
        }
        if (emit) {
            // push!(Ξ, (t, copy(x), copy(θ), nothing)), :138, :39-41 -- coalesced 8(2d+1)-byte record
            if (P.trace_cap > 0) {
                const int64_t slot = chain * P.trace_cap + (int64_t)ntrace;
                if (lane == 0) P.ev_t[slot] = t;
                double* ex = P.ev_x + slot * d;
                double* eth = P.ev_th + slot * d;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int64_t e = (int64_t)s * 64 + lane;
                    if (FULL || e < d) {
                        ex[e] = x[s];
                        eth[e] = th[s];
                    }
                }
            }
            ntrace += 1;
            nevents += 1;
            if (!stop_before && !(t < T)) running = false;
        }
    }

#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int64_t e = (int64_t)s * 64 + lane;
        if (FULL || e < d) {
            gx[e] = x[s];
            gth[e] = th[s];
        }
    }
    if (lane == 0) {
        sc[0] = t;
        sc[1] = a;
        sc[2] = b;
        sc[3] = tp;
        sc[4] = tau_ref;
        sc[5] = c;
        if constexpr (EXT) {
            sc[6] = renew ? 1.0 : 0.0;
            sc[7] = hz;
        }
        hdr->c.t_last = t;
        hdr->t_event = t;
        hdr->c.num = num;
        hdr->c.nacc = nacc;
        hdr->c.nrefresh = nrefresh;
        hdr->c.ntrace = ntrace;
        hdr->c.nevents = nevents;
        hdr->c.ndraw_main = nm;
        hdr->c.status = status;
    }
}

// Initial state, src/not_fact_samplers.jl:117-135: τref = randexp(rng)/λref (draw 0), ∇ϕx, abc = ab(...), t′ = next_time (draw 1).
template <int NS, bool BOOM>
__global__ __launch_bounds__(64) void bps_init_kernel(BpsRunParams P, const uint64_t* seeds, double t0, double c0) {
    const int lane = threadIdx.x;
    const int64_t chain = blockIdx.x;
    const int64_t d = P.d;
    extern __shared__ __align__(16) unsigned char smem[];
    double* tmp = reinterpret_cast<double*>(smem);
    const double* gx = P.x + chain * d;
    const double* gth = P.th + chain * d;
    const uint64_t seed = seeds[chain];
    double x[NS], th[NS], g[NS], gt[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int64_t e = (int64_t)s * 64 + lane;
        x[s] = (e < d) ? gx[e] : 0.0;
        th[s] = (e < d) ? gth[e] : 0.0;
    }
    auto apply_gamma = [&](const double (&in)[NS], bool sub_mu, double (&out)[NS]) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int64_t e = (int64_t)s * 64 + lane;
            if (e < d) tmp[e] = sub_mu ? (in[s] - P.mu[e]) : in[s];
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int64_t e = (int64_t)s * 64 + lane;
            double y = 0.0;
            if (e < d) {
                for (int64_t p = P.colptr[e]; p < P.colptr[e + 1]; ++p) y += P.nzval[p] * tmp[P.rowval[p]];
            }
            out[s] = y;
        }
    };
    auto dot = [&](const double (&u)[NS], const double (&v)[NS]) -> double {
        double part = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int64_t e = (int64_t)s * 64 + lane;
            if (e < d) part += u[s] * v[s];
        }
        return wave_sum_f64(part);
    };
    auto apply_target = [&](const double (&in)[NS], bool sub_mu, double (&out)[NS]) {  // the ensemble's own target Γt, μt
        asm volatile("" ::: "memory");
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int64_t e = (int64_t)s * 64 + lane;
            if (e < d) tmp[e] = sub_mu ? (in[s] - P.t_mu[e]) : in[s];
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int64_t e = (int64_t)s * 64 + lane;
            double y = 0.0;
            if (e < d) {
                for (int64_t p = P.t_colptr[e]; p < P.t_colptr[e + 1]; ++p) y += P.t_nzval[p] * tmp[P.t_rowval[p]];
            }
            out[s] = y;
        }
    };
    const bool own_target = !BOOM && P.t_colptr != nullptr;
    const double tau_ref = -pdmp_log(pdmp_u01(seed, PDMP_STREAM_MAIN, 0)) / P.lambda_ref;  // :121
    if (own_target) apply_target(x, true, g);
    else apply_gamma(x, true, g);                                                          // :122-123
    double a, b;
    if (BOOM) {  // grad_correct! only shifts g (unused by the Boomerang bound); ab, src/not_fact_samplers.jl:34-36
        double dx[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int64_t e = (int64_t)s * 64 + lane;
            dx[s] = (e < d) ? (x[s] - P.mu_flow[e]) : 0.0;
        }
        a = sqrt(dot(th, th) + dot(dx, dx)) * c0;
        b = 0.0;
    } else {
        if (own_target && !P.local_bound) {  // ab(…GlobalBound…) with the flow's Γ, μ (:26-28)
            double gb[NS];
            apply_gamma(x, true, gb);
            a = c0 + dot(th, gb);
        } else {
            a = c0 + dot(th, g);                                                           // :126
        }
        if (own_target && P.local_bound) apply_target(th, false, gt);  // v = θ'Γtθ (:29-31)
        else apply_gamma(th, false, gt);
        b = dot(th, gt);
    }
    double tp = t0 + bps_poisson_time(a, b, pdmp_u01(seed, PDMP_STREAM_MAIN, 1));          // :135
    double hz = BPS_INF;
    bool renew = false;
    if (!BOOM && P.local_bound) {  // next_time with the LocalBound horizon, :29-31,43-50
        hz = 2 * sqrt((double)d) / c0 / sqrt(dot(th, th));
        const double dt = bps_poisson_time(a, b, pdmp_u01(seed, PDMP_STREAM_MAIN, 1));
        renew = dt > hz;
        tp = renew ? t0 + hz : t0 + dt;
    }
    if (lane == 0) {
        double* sc = P.scal + chain * 8;
        sc[0] = t0;
        sc[1] = a;
        sc[2] = b;
        sc[3] = tp;
        sc[4] = tau_ref;
        sc[5] = c0;
        sc[6] = renew ? 1.0 : 0.0;
        sc[7] = hz;
        DevChain h;
        h.c.t_last = t0;
        h.c.num = 0;
        h.c.nacc = 0;
        h.c.nrefresh = 0;
        h.c.ntrace = 0;
        h.c.nevents = 0;
        h.c.ndraw_main = 2;
        h.c.ndraw_global = 0;
        h.c.status = PDMP_CHAIN_OK;
        h.c.reserved = 0;
        h.seed = seed;
        h.t0 = t0;
        h.t_event = t0;
        h.tl_scale = 0.0;
        for (int k = 0; k < 3; ++k) h.pad[k] = 0;
        P.hdr[chain] = h;
    }
}

template <int NS>
static int launch_ns(const BpsRunParams& p, int64_t nchains, bool diag, bool init, const uint64_t* seeds, double t0,
                     double c0, void* stream) {
    const size_t lds = (size_t)p.d * 8;
    dim3 grid((unsigned)nchains), block(64);
    const bool boom = p.flow_kind == 1;
    if (init) {
        if (boom) hipLaunchKernelGGL((bps_init_kernel<NS, true>), grid, block, lds, (hipStream_t)stream, p, seeds, t0, c0);
        else hipLaunchKernelGGL((bps_init_kernel<NS, false>), grid, block, lds, (hipStream_t)stream, p, seeds, t0, c0);
    } else if (p.ext) {
        if (boom) hipLaunchKernelGGL((bps_run_kernel<NS, false, true, false, false, true>), grid, block, lds, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((bps_run_kernel<NS, false, false, false, false, true>), grid, block, lds, (hipStream_t)stream, p);
    } else if (boom) {
        if (diag) hipLaunchKernelGGL((bps_run_kernel<NS, true, true, false>), grid, block, lds, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((bps_run_kernel<NS, false, true, false>), grid, block, lds, (hipStream_t)stream, p);
    } else if (diag && p.ident && p.d == (int64_t)NS * 64) {
        hipLaunchKernelGGL((bps_run_kernel<NS, true, false, true, true>), grid, block, lds, (hipStream_t)stream, p);
    } else if (diag && p.ident) {
        hipLaunchKernelGGL((bps_run_kernel<NS, true, false, true>), grid, block, lds, (hipStream_t)stream, p);
    } else if (diag) {
        hipLaunchKernelGGL((bps_run_kernel<NS, true, false, false>), grid, block, lds, (hipStream_t)stream, p);
    } else {
        hipLaunchKernelGGL((bps_run_kernel<NS, false, false, false>), grid, block, lds, (hipStream_t)stream, p);
    }
    return (int)hipGetLastError();
}

template <int NS>
static int launch_big(const BpsRunParams& p, int64_t nchains, bool init, const uint64_t* seeds, double t0, double c0, void* stream) {
    const size_t lds = (size_t)p.d * 8;
    dim3 grid((unsigned)nchains), block(64);
    const bool boom = p.flow_kind == 1;
    if (init) {
        if (boom) hipLaunchKernelGGL((bps_init_kernel<NS, true>), grid, block, lds, (hipStream_t)stream, p, seeds, t0, c0);
        else hipLaunchKernelGGL((bps_init_kernel<NS, false>), grid, block, lds, (hipStream_t)stream, p, seeds, t0, c0);
    } else if (boom) {
        hipLaunchKernelGGL((bps_run_kernel<NS, false, true, false, false, true>), grid, block, lds, (hipStream_t)stream, p);
    } else {
        hipLaunchKernelGGL((bps_run_kernel<NS, false, false, false, false, true>), grid, block, lds, (hipStream_t)stream, p);
    }
    return (int)hipGetLastError();
}

static int dispatch(const BpsRunParams& p, int64_t nchains, bool diag, bool init, const uint64_t* seeds, double t0,
                    double c0, void* stream) {
    const int64_t ns = (p.d + 63) / 64;
    if (ns <= 1) return launch_ns<1>(p, nchains, diag, init, seeds, t0, c0, stream);
    if (ns <= 2) return launch_ns<2>(p, nchains, diag, init, seeds, t0, c0, stream);
    if (ns <= 4) return launch_ns<4>(p, nchains, diag, init, seeds, t0, c0, stream);
    if (ns <= 8) return launch_ns<8>(p, nchains, diag, init, seeds, t0, c0, stream);
    if (ns <= 16) return launch_ns<16>(p, nchains, diag, init, seeds, t0, c0, stream);
    // beyond 1024 coordinates the vectors no longer fit the register file as they are used here: the general instantiation (every option) is
    // compiled for 32 and 64 slots per lane, the compiler keeping what does not fit in AGPRs and scratch -- a capability, not a fast path
    if (ns <= 32) return launch_big<32>(p, nchains, init, seeds, t0, c0, stream);
    if (ns <= 64) return launch_big<64>(p, nchains, init, seeds, t0, c0, stream);
    return -1;
}

// Write-only probe with the event-record store pattern of bps_run_kernel (one wave per chain, `nrec` records of x(d) and θ(d)
// written as NS x 512-byte coalesced stores each): the ceiling the C2 roofline fraction is read against.
__global__ __launch_bounds__(64) void bps_write_probe_kernel(double* ev_x, double* ev_th, int64_t d, int64_t cap, int64_t nrec) {
    const int lane = threadIdx.x;
    const int64_t chain = blockIdx.x;
    double v = (double)chain + 1e-3 * lane;
    for (int64_t r = 0; r < nrec; ++r) {
        const int64_t slot = chain * cap + r;
        double* ex = ev_x + slot * d;
        double* eth = ev_th + slot * d;
        for (int64_t e = lane; e < d; e += 64) {
            ex[e] = v;
            eth[e] = -v;
        }
        v += 1.0;
    }
}
int launch_bps_write_probe(double* ev_x, double* ev_th, int64_t d, int64_t cap, int64_t nrec, int64_t nchains, void* stream) {
    hipLaunchKernelGGL(bps_write_probe_kernel, dim3((unsigned)nchains), dim3(64), 0, (hipStream_t)stream, ev_x, ev_th, d, cap, nrec);
    return (int)hipGetLastError();
}

// Random 32-byte-sector traffic of the local ZigZag kernels' record accesses, and nothing else: one wavefront per chain, every lane
// reads the first half of four pseudo-random 64-byte records of its chain per round (and, with `write`, stores it back changed).
// The rate this reaches is the practical ceiling for the scattered part of the event loop's memory traffic.
__global__ __launch_bounds__(64) void sector_probe_kernel(double* rec, int64_t d, int rounds, int write, double* sink) {
    const int lane = threadIdx.x;
    const int64_t chain = blockIdx.x;
    double* base = rec + chain * d * 8;
    uint32_t h = (uint32_t)chain * 2654435761u + (uint32_t)lane * 40503u + 12345u;
    double acc = 0.0;
    if (write >= 13) {
        // What ONE REJECTED PROPOSAL of the tracked kernels costs the memory system, in two layouts (64 proposals per round and wavefront):
        //   13: as built -- the coordinate's record line read by its lane (two sectors) and a 32-byte sector of it written back; the key
        //       block's line read by an 8-lane group (16 B per lane) and one 8-byte key written (TWO dirty lines per proposal)
        //   14: keys and proposal times interleaved -- the record line only READ; a 256-byte aligned pair of lines read by the group
        //       (32 B per lane) and 16 bytes of it written (ONE dirty line per proposal)
        //   15 / 16: the reads of 13 / 14 alone
        //   17: 13 with non-temporal stores
        //   18: (key, proposal time) pairs in blocks of EIGHT -- the record line only read, ONE pair line read (16 B per lane) and 16 bytes of it written
        const bool pairs = (write == 14 || write == 16), wr = (write <= 14 || write >= 17), nt = (write == 17), p8 = (write == 18);
        const int g = lane >> 3, gl = lane & 7;
        uint32_t hg = (uint32_t)chain * 2654435761u + (uint32_t)g * 40503u + 777u;
        for (int r = 0; r < rounds; ++r) {
            h = h * 1664525u + 1013904223u;
            double* recl = base + (size_t)((h >> 8) % (uint32_t)(d / 2 - 4)) * 16;
            const double2 r0 = reinterpret_cast<double2*>(recl)[0], r1 = reinterpret_cast<double2*>(recl)[2];
            double2 k[8], k2[8];
            double* bl[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                hg = hg * 1664525u + 1013904223u;
                const uint32_t rnd = hg >> 8;
                if (pairs) {
                    bl[q] = base + (size_t)(rnd % (uint32_t)(d / 4 - 4)) * 32;
                    k[q] = reinterpret_cast<double2*>(bl[q])[2 * gl];
                    k2[q] = reinterpret_cast<double2*>(bl[q])[2 * gl + 1];
                } else {
                    bl[q] = base + (size_t)(rnd % (uint32_t)(d / 2 - 4)) * 16;
                    k[q] = reinterpret_cast<double2*>(bl[q])[gl];
                    k2[q] = k[q];
                }
            }
            acc += r0.x + r1.y;
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += k[q].x + k2[q].y;
            if (wr) {
                if (p8) {
                } else if (nt) {
                    __builtin_nontemporal_store(r0.x + 1.0, recl + 8);
                    __builtin_nontemporal_store(r1.y, recl + 9);
                    __builtin_nontemporal_store(r1.x, recl + 10);
                    __builtin_nontemporal_store(r1.y, recl + 11);
                } else if (!pairs) {
                    reinterpret_cast<double2*>(recl)[4] = make_double2(r0.x + 1.0, r1.y), reinterpret_cast<double2*>(recl)[5] = r1;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (gl == (q & 7)) {
                        if (p8) reinterpret_cast<double2*>(bl[q])[gl] = make_double2(k[q].x + 1.0, k[q].y);
                        else if (pairs) reinterpret_cast<double2*>(bl[q])[2 * gl] = make_double2(k[q].x + 1.0, k[q].y);
                        else if (nt) __builtin_nontemporal_store(k[q].x + 1.0, bl[q] + 2 * gl);
                        else bl[q][2 * gl] = k[q].x + 1.0;
                    }
                }
            }
        }
        if (acc == 123.456) sink[0] = acc;
        return;
    }
    if (write >= 7) {
        // WIDE requests: a group of lanes reads one contiguous, aligned run as a unit (the TA merges the lanes of one instruction
        // that fall into one 128-byte line into a single request).  7/8: groups of 4 lanes, a random 128-byte line, each lane two
        // 16-byte pieces (read / read and write back); 9/10: groups of 2 lanes, a random 64-byte record; 11/12: groups of 8 lanes of
        // which 6 read a 96-byte run at a random 32-byte boundary (a lattice row's three hot records packed back to back).
        const int gsz = (write <= 8) ? 4 : (write <= 10) ? 2 : 8;
        const int gl = lane & (gsz - 1);
        const bool wr = (write == 8 || write == 10 || write == 12);
        uint32_t hg = (uint32_t)chain * 2654435761u + (uint32_t)(lane / gsz) * 40503u + 12345u;
        for (int r = 0; r < rounds; ++r) {
            double2* p[4];
            double2* p2[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                hg = hg * 1664525u + 1013904223u;
                const uint32_t rnd = hg >> 8;
                if (write <= 8) {
                    double* line = base + (size_t)(rnd % (uint32_t)(d / 2 - 2)) * 16;  // 128-byte lines
                    p[q] = reinterpret_cast<double2*>(line) + gl;
                    p2[q] = p[q] + 4;
                } else if (write <= 10) {
                    double* recp = base + (size_t)(rnd % (uint32_t)(d - 4)) * 8;  // 64-byte records
                    p[q] = reinterpret_cast<double2*>(recp) + gl;
                    p2[q] = p[q] + 2;
                } else {
                    double* run = base + (size_t)(rnd % (uint32_t)(2 * d - 16)) * 4;  // 32-byte boundary, 96 bytes
                    p[q] = reinterpret_cast<double2*>(run) + (gl < 6 ? gl : 0);
                    p2[q] = p[q];
                }
            }
            double2 a[4], b[4];
            const bool act = (write <= 10) || gl < 6;
            if (act) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a[q] = p[q][0];
                    if (write <= 10) b[q] = p2[q][0];
                    else b[q] = a[q];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc += a[q].x + b[q].y;
                    if (wr) {
                        p[q][0] = make_double2(a[q].x + 1.0, a[q].y);
                        if (write <= 10) p2[q][0] = make_double2(b[q].x, b[q].y + 1.0);
                    }
                }
            }
        }
        if (acc == 123.456) sink[0] = acc;
        return;
    }
    for (int r = 0; r < rounds; ++r) {
        double2* p[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            h = h * 1664525u + 1013904223u;
            p[q] = reinterpret_cast<double2*>(base + (size_t)((h >> 8) % (uint32_t)(d - 4)) * 8);
        }
        double2 a[4], b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a[q] = p[q][0];
            b[q] = p[q][1];
        }
        if (write == 5 || write == 6) {  // a lattice row's three neighbours: hot halves at 64-byte pitch (5) or packed at 32 (6)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const size_t pitch = (write == 5) ? 4 : 2;  // in double2 units
                const double2 l0 = p[q][pitch], l1 = p[q][pitch + 1], r0 = p[q][2 * pitch], r1 = p[q][2 * pitch + 1];
                acc += l0.x + l1.y + r0.x + r1.y;
            }
        }
        if (write == 2 || write == 3) {  // whole 64-byte records (is a second 32-byte sector of the same record free?)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double2 c2 = p[q][2], d2 = p[q][3];
                acc += c2.x + d2.y;
                if (write == 3) {  // ... and written back whole
                    p[q][2] = make_double2(c2.x + 1.0, c2.y);
                    p[q][3] = make_double2(d2.x, d2.y + 1.0);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc += a[q].x + b[q].y;
            if (write == 1 || write == 3) {
                p[q][0] = make_double2(a[q].x + 1.0, a[q].y);
                p[q][1] = make_double2(b[q].x, b[q].y + 1.0);
            }
        }
    }
    if (acc == 123.456) sink[0] = acc;
}
int launch_sector_probe(double* rec, int64_t d, int64_t nchains, int rounds, int write, double* sink, void* stream) {
    hipLaunchKernelGGL(sector_probe_kernel, dim3((unsigned)nchains), dim3(64), 0, (hipStream_t)stream, rec, d, rounds, write, sink);
    return (int)hipGetLastError();
}

int launch_bps_init(const BpsRunParams& p, int64_t nchains, const uint64_t* seeds, double t0, double c0, void* stream) {
    return dispatch(p, nchains, false, true, seeds, t0, c0, stream);
}
int launch_bps_run(const BpsRunParams& p, int64_t nchains, bool diag, void* stream) {
    return dispatch(p, nchains, diag, false, nullptr, 0.0, 0.0, stream);
}

}  // namespace pdmp
