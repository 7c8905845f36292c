// pdmp_mi355.hpp -- C++17 host-side mirror of the reference's sampler entry points on top of the C ABI (pdmp_mi355.h).
//
// The reference is Julia; its boundary is the method signatures
//     spdmp (∇ϕ, t0, x0, θ0, T, c, F::Union{ZigZag,FactBoomerang}, args...; factor=1.8, adapt=false, adaptscale=false, seed)
//                                                   -> Ξ::FactTrace, (t, x, θ), (acc, num), c      (src/sfact.jl:162-163,211,214)
//     pdmp  (∇ϕ, ...same...)  = spdmp(..., All(), ...)                                              (src/sfact.jl:236)
//     pdmp  (∇ϕ!, t0, x0, θ0, T, c, Flow::Union{BouncyParticle,Boomerang}; adapt, factor=2.0)
//                                                   -> Ξ::PDMPTrace, (t, x, θ), (acc, num), c      (src/not_fact_samplers.jl:117,146)
//     sspdmp(∇ϕ, t0, x0, θ0, T, c, F::ZigZag, κ, args...; reversible, strong_upperbounds, factor=1.5, adapt)
//                                                                                                  (src/ss_fact.jl:159-160,217)
// Same names, argument order, keyword meaning (struct Options), return shape (struct Result) and error behaviour (the
// reference's `error("Tuning parameter `c` too small.")`, src/sfact.jl:124, becomes std::runtime_error with that text).
// Differences forced by the C ABI: `∇ϕ, args...` is an enumerated target (GaussianTarget, LogisticTarget); coordinates are
// 0-based; one call runs ONE chain (pass nchains > 1 through pdmp::Ensemble directly for ensembles).
// Header-only; link with -lpdmp_mi355.  There is no CPU fallback: without a gfx950 device every call throws.
#ifndef PDMP_MI355_HPP
#define PDMP_MI355_HPP

#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "pdmp_mi355.h"

namespace pdmp {

struct Error : std::runtime_error {
    pdmp_status code;
    Error(pdmp_status c, const std::string& what) : std::runtime_error(what), code(c) {}
};
inline void check(pdmp_status st) {
    if (st != PDMP_OK) throw Error(st, pdmp_last_error());
}

// SparseMatrixCSC{Float64,Int64}, 0-based (SparseArrays layout, src/common.jl:17-20)
struct SparseCSC {
    int64_t n = 0;
    std::vector<int64_t> colptr, rowval;
    std::vector<double> nzval;
};

// ZigZag(Γ, μ, σ=ones; ρ=0, λref=0)  src/types.jl:19-27
struct ZigZag {
    SparseCSC Gamma;
    std::vector<double> mu, sigma;
    double lambda_ref = 0.0, rho = 0.0;
};
// FactBoomerang(Γ, μ, λ, σ=diag(Γ)^-1/2; ρ=0)  src/types.jl:71-79
struct FactBoomerang {
    SparseCSC Gamma;
    std::vector<double> mu, sigma;
    double lambda_ref = 0.1, rho = 0.0;
};
// BouncyParticle(Γ, μ, λ, ρ, U, L)  src/types.jl:35-45.  L is the mass factor the reference's short constructor computes as
// cholesky(Symmetric(Γ)).L (:43): LOWER triangular CSC with a stored diagonal; leave it empty (n == 0) only for Γ = I -- the engine refuses a
// general Γ without its factor (PDMP_ERR_UNSUPPORTED) instead of running with a silent L = I.
struct BouncyParticle {
    SparseCSC Gamma;
    std::vector<double> mu;
    double lambda_ref = 1.0, rho = 0.0;
    SparseCSC L;
};
// Boomerang(Γ, μ, λ; ρ=0)  src/types.jl:59-66: Γ enters through its factor L only (empty: identity)
struct Boomerang {
    std::vector<double> mu;
    double lambda_ref = 1.0, rho = 0.0;
    SparseCSC L;
};
// ∇ϕ(x, i, Γ) = idot(Γ, i, x) [- idot(Γ, i, μ)]  (scripts/gaussianrandomfield.jl:25)
struct GaussianTarget {
    SparseCSC Gamma;
    std::vector<double> mu;  // empty: no shift
};
// ∇ϕmoving(t, x, θ, i, t′, F, A, At, μ, y, ny, k), SelfMoving()  (scripts/logistic.jl:78-95,107,167)
struct LogisticTarget {
    int64_t n = 0;  // observations
    SparseCSC A, At;
    std::vector<double> y, ny, mu;
    double gamma0 = 0.01;
    int64_t k_sub = 10;
};

struct Options {  // the reference's keyword arguments
    double factor = 1.8;
    bool adapt = false;
    bool adaptscale = false;
    bool local_bound = false;  // c is LocalBound(c): spdmp(∇ϕ, t0, x0, θ0, T, C::LocalBound, F, args...), src/local.jl:95-149; for the
                               // non-factorised pdmp: src/not_fact_samplers.jl:29-31,65-71
    bool subsample = false;    // pdmp(∇ϕ!, ...; subsample) of the non-factorised samplers, src/not_fact_samplers.jl:53,90
    bool tracked = false;      // engine-only: tracked-gradient evaluation of spdmp (pdmp_ensemble_set_gradient_tracking), the engine's fast
                               // path (2.5 x the default's rate on C3).  NOT the reference's arithmetic: floats to ~1e-13 instead of bit for
                               // bit; indices / counters / bounds identical until such a difference flips a test -- measured 3 of 4096
                               // chains by T = 20 on the 128 x 128 lattice (6e-10 per proposal).  false: the reference's evaluation order
    uint64_t seed = 0x5EED0000ull;
    int device = 0;
    int64_t trace_capacity = 0;  // 0: sized from d and T, refilled on demand
    // sspdmp only
    bool reversible = false, strong_upperbounds = false;
    // the optional argument G of spdmp(∇ϕ, t0, x0, θ0, T, c, G, F, ...) / sspdmp(..., c, G, F, κ, ...) (src/sfact.jl:162, src/ss_fact.jl:159):
    // column patterns = the G[i] (values unused); empty = Matched().  G ⊇ G1 or the call throws (the reference's @assert, src/sfact.jl:177)
    SparseCSC G;
};

using Event = pdmp_event;  // (t, i, x, θ), src/trace.jl:38; i 0-based

struct FactTrace {  // FactTrace(F, t0, x0, θ0, events), src/trace.jl:7-13
    double t0 = 0.0;
    std::vector<double> x0, theta0;
    std::vector<Event> events;
};
struct PDMPTrace {  // PDMPTrace(F, t0, x0, θ0, events), src/trace.jl:20-27; events (t, copy(x), copy(θ))
    double t0 = 0.0;
    int64_t d = 0;
    std::vector<double> x0, theta0;
    std::vector<double> t, x, theta;  // [n], [n x d], [n x d]
};

template <class Trace>
struct Result {  // Ξ, (t, x, θ), (acc, num), c
    Trace trace;
    std::vector<double> t, x, theta;  // factorised samplers: t is the vector of per-coordinate clocks (src/sfact.jl:210-211)
    std::vector<int64_t> acc;         // per coordinate (spdmp) or one entry (BPS, sticky: scalar acc)
    int64_t num = 0;
    std::vector<double> c;
    std::vector<double> sigma;  // tuned F.σ when adaptscale (the reference mutates F.σ in place)
};

// RAII handle on pdmp_ensemble
class Ensemble {
  public:
    Ensemble(int64_t nchains, int64_t d, int sampler, const Options& o, int64_t trace_capacity) : nchains_(nchains), d_(d) {
        pdmp_config cfg{};
        cfg.struct_size = sizeof cfg;
        cfg.device = o.device;
        cfg.sampler = sampler;
        cfg.adapt = o.adapt ? 1 : 0;
        cfg.factor = o.factor;
        cfg.nchains = nchains;
        cfg.d = d;
        cfg.trace_capacity = trace_capacity;
        check(pdmp_ensemble_create(&cfg, &h_));
    }
    ~Ensemble() {
        if (h_) pdmp_ensemble_destroy(h_);
    }
    Ensemble(const Ensemble&) = delete;
    Ensemble& operator=(const Ensemble&) = delete;
    pdmp_ensemble* get() const { return h_; }
    int64_t nchains() const { return nchains_; }
    int64_t d() const { return d_; }

    std::vector<pdmp_chain_counters> counters() const {
        std::vector<pdmp_chain_counters> c((size_t)nchains_);
        check(pdmp_ensemble_counters(h_, c.data()));
        return c;
    }
    // run to T; drains the trace of chain 0..nchains-1 into `sink(chain, events)` whenever the buffer fills
    template <class Sink>
    std::vector<pdmp_chain_counters> run_and_drain(double T, Sink&& sink) {
        for (;;) {
            check(pdmp_ensemble_run(h_, T, PDMP_RUN_REFERENCE_TAIL, nullptr));
            check(pdmp_ensemble_sync(h_));
            auto cnt = counters();
            for (const auto& k : cnt)
                if (k.status == PDMP_CHAIN_BOUND_VIOLATED) throw std::runtime_error("Tuning parameter `c` too small.");
            sink(cnt);
            bool full = false;
            for (const auto& k : cnt) full = full || k.status == PDMP_CHAIN_TRACE_FULL || k.status == PDMP_CHAIN_PAUSED;  // (both resume with the next run)
            if (!full) return cnt;
        }
    }

  private:
    pdmp_ensemble* h_ = nullptr;
    int64_t nchains_, d_;
};

namespace detail {
inline const double* opt(const std::vector<double>& v) { return v.empty() ? nullptr : v.data(); }
inline int64_t default_capacity(int64_t d, double t0, double T) {
    const double span = std::max(T - t0, 1.0);
    return (int64_t)std::min(std::max(1024.0, 2.0 * (double)d * span), (double)(1 << 22));
}
inline void set_target(Ensemble& e, const GaussianTarget& g) {
    check(pdmp_ensemble_set_target_gaussian_csc(e.get(), g.Gamma.colptr.data(), g.Gamma.rowval.data(), g.Gamma.nzval.data(),
                                                opt(g.mu)));
}
inline void set_target(Ensemble& e, const LogisticTarget& g) {
    check(pdmp_ensemble_set_target_logistic(e.get(), g.n, g.A.colptr.data(), g.A.rowval.data(), g.A.nzval.data(),
                                            g.At.colptr.data(), g.At.rowval.data(), g.At.nzval.data(), g.y.data(), g.ny.data(),
                                            g.mu.data(), g.gamma0, g.k_sub));
}
inline void set_flow(Ensemble& e, const ZigZag& F) {
    check(pdmp_ensemble_set_flow_zigzag(e.get(), F.Gamma.colptr.data(), F.Gamma.rowval.data(), F.Gamma.nzval.data(), opt(F.mu),
                                        opt(F.sigma), F.lambda_ref, F.rho));
}
inline void set_flow(Ensemble& e, const FactBoomerang& F) {
    check(pdmp_ensemble_set_flow_factboomerang(e.get(), F.Gamma.colptr.data(), F.Gamma.rowval.data(), F.Gamma.nzval.data(),
                                               opt(F.mu), opt(F.sigma), F.lambda_ref, F.rho));
}

template <class Target, class Flow>
Result<FactTrace> factorised(int sampler, const Target& target, double t0, const std::vector<double>& x0,
                             const std::vector<double>& theta0, double T, const std::vector<double>& c, const Flow& F,
                             const std::vector<double>* kappa, const Options& o) {
    const int64_t d = (int64_t)x0.size();
    const int64_t cap = o.trace_capacity > 0 ? o.trace_capacity : default_capacity(d, t0, T);
    Ensemble e(1, d, sampler, o, cap);
    set_flow(e, F);
    if (o.G.n > 0) check(pdmp_ensemble_set_neighbourhood(e.get(), o.G.colptr.data(), o.G.rowval.data()));
    set_target(e, target);
    if (kappa) check(pdmp_ensemble_set_sticky(e.get(), kappa->data(), o.reversible ? 1 : 0, o.strong_upperbounds ? 1 : 0));
    if (o.adaptscale) check(pdmp_ensemble_set_adaptscale(e.get(), 1));
    if (o.local_bound) check(pdmp_ensemble_set_local_bound(e.get(), 1));
    if (o.tracked) check(pdmp_ensemble_set_gradient_tracking(e.get(), 1));
    const uint64_t seed = o.seed;
    check(pdmp_ensemble_set_state(e.get(), t0, x0.data(), theta0.data(), c.data(), &seed));
    Result<FactTrace> R;
    R.trace.t0 = t0;
    R.trace.x0 = x0;
    R.trace.theta0 = theta0;
    auto cnt = e.run_and_drain(T, [&](const std::vector<pdmp_chain_counters>& k) {
        const int64_t m = (int64_t)k[0].ntrace;
        if (m > 0) {
            const size_t old = R.trace.events.size();
            R.trace.events.resize(old + (size_t)m);
            check(pdmp_ensemble_trace_copy(e.get(), 0, 0, m, R.trace.events.data() + old));
        }
        check(pdmp_ensemble_trace_reset(e.get()));
    });
    R.t.resize((size_t)d);
    R.x.resize((size_t)d);
    R.theta.resize((size_t)d);
    R.acc.resize((size_t)d);
    R.c.resize((size_t)d);
    check(pdmp_ensemble_final_state(e.get(), 0, 1, R.t.data(), R.x.data(), R.theta.data(), R.acc.data(), R.c.data()));
    if (!o.adapt) R.c = c;
    if (kappa) R.acc.assign(1, (int64_t)cnt[0].nacc);  // sticky: scalar acc, src/ss_fact.jl:175
    R.num = (int64_t)cnt[0].num;
    if (o.adaptscale) {
        R.sigma.resize((size_t)d);
        check(pdmp_ensemble_final_sigma(e.get(), 0, 1, R.sigma.data()));
    }
    return R;
}

inline Result<PDMPTrace> not_factorised(Ensemble& e, double t0, const std::vector<double>& x0, const std::vector<double>& theta0,
                                        double T, double c, const Options& o) {
    const int64_t d = (int64_t)x0.size();
    const uint64_t seed = o.seed;
    check(pdmp_ensemble_set_state_bps(e.get(), t0, x0.data(), theta0.data(), c, &seed));
    Result<PDMPTrace> R;
    R.trace.t0 = t0;
    R.trace.d = d;
    R.trace.x0 = x0;
    R.trace.theta0 = theta0;
    auto cnt = e.run_and_drain(T, [&](const std::vector<pdmp_chain_counters>& k) {
        const int64_t m = (int64_t)k[0].ntrace;
        if (m > 0) {
            const size_t old = R.trace.t.size();
            R.trace.t.resize(old + (size_t)m);
            R.trace.x.resize((old + (size_t)m) * (size_t)d);
            R.trace.theta.resize((old + (size_t)m) * (size_t)d);
            check(pdmp_ensemble_bps_trace_copy(e.get(), 0, 0, m, R.trace.t.data() + old, R.trace.x.data() + old * (size_t)d,
                                               R.trace.theta.data() + old * (size_t)d));
        }
        check(pdmp_ensemble_trace_reset(e.get()));
    });
    R.t.resize(1);
    R.x.resize((size_t)d);
    R.theta.resize((size_t)d);
    R.c.resize(1);
    check(pdmp_ensemble_bps_final_state(e.get(), 0, 1, R.t.data(), R.x.data(), R.theta.data(), R.c.data()));
    R.acc.assign(1, (int64_t)cnt[0].nacc);
    R.num = (int64_t)cnt[0].num;
    return R;
}
}  // namespace detail

// spdmp(∇ϕ, t0, x0, θ0, T, c, F, args...; factor, adapt, adaptscale, seed)  -- src/sfact.jl:162-212,214
template <class Target, class Flow>
Result<FactTrace> spdmp(const Target& target, double t0, const std::vector<double>& x0, const std::vector<double>& theta0,
                        double T, const std::vector<double>& c, const Flow& F, const Options& o = {}) {
    return detail::factorised(PDMP_SAMPLER_ZIGZAG_LOCAL, target, t0, x0, theta0, T, c, F, nullptr, o);
}
// pdmp(∇ϕ, t0, x0, θ0, T, c, F, args...) = spdmp(..., All(), ...)  -- src/sfact.jl:236
template <class Target, class Flow>
Result<FactTrace> pdmp(const Target& target, double t0, const std::vector<double>& x0, const std::vector<double>& theta0,
                       double T, const std::vector<double>& c, const Flow& F, const Options& o = {}) {
    return detail::factorised(PDMP_SAMPLER_ZIGZAG_ALL, target, t0, x0, theta0, T, c, F, nullptr, o);
}
// parallel_spdmp(partition, ∇ϕ, t0, x0, θ0, T, c, G, F::ZigZag; factor, adapt, Δ)  -- src/parallel.jl:104-175: the chain on `nt` wavefronts,
// one per chunk of d / nt coordinates.  F.Gamma is the bounding Γ written on G's pattern (explicit zeros where it has no entry) and g1_mask
// marks its own structural entries (empty: all of them); the result's events are sorted by time (:167), acc holds the scalar of the reference.
template <class Target>
Result<FactTrace> parallel_spdmp(int nt, const Target& target, double t0, const std::vector<double>& x0, const std::vector<double>& theta0,
                                 double T, const std::vector<double>& c, const ZigZag& F, const std::vector<uint8_t>& g1_mask,
                                 double Delta = 0.1, const Options& o = {}) {
    const int64_t d = (int64_t)x0.size();
    const int64_t cap = o.trace_capacity > 0 ? o.trace_capacity : 2 * detail::default_capacity(d, t0, T);
    Ensemble e(1, d, PDMP_SAMPLER_ZIGZAG_LOCAL, o, cap);
    detail::set_flow(e, F);
    detail::set_target(e, target);
    const uint64_t seed = o.seed;
    check(pdmp_ensemble_set_state(e.get(), t0, x0.data(), theta0.data(), c.data(), &seed));
    check(pdmp_ensemble_run_partitioned(e.get(), T, nt, Delta, g1_mask.empty() ? nullptr : g1_mask.data(), (int64_t)g1_mask.size(), nullptr));
    const auto cnt = e.counters();
    if (cnt[0].status == PDMP_CHAIN_BOUND_VIOLATED) throw std::runtime_error("Tuning parameter `c` too small.");
    if (cnt[0].status == PDMP_CHAIN_TRACE_FULL) throw std::runtime_error("trace_capacity too small for a partitioned run (it is not resumable)");
    Result<FactTrace> R;
    R.trace.t0 = t0;
    R.trace.x0 = x0;
    R.trace.theta0 = theta0;
    R.trace.events.resize((size_t)cnt[0].ntrace);
    if (cnt[0].ntrace > 0) check(pdmp_ensemble_trace_copy(e.get(), 0, 0, (int64_t)cnt[0].ntrace, R.trace.events.data()));
    std::stable_sort(R.trace.events.begin(), R.trace.events.end(), [](const pdmp_event& a, const pdmp_event& b) { return a.t < b.t; });
    R.t.resize((size_t)d);
    R.x.resize((size_t)d);
    R.theta.resize((size_t)d);
    R.c.resize((size_t)d);
    std::vector<int64_t> acc_vec((size_t)d);
    check(pdmp_ensemble_final_state(e.get(), 0, 1, R.t.data(), R.x.data(), R.theta.data(), acc_vec.data(), R.c.data()));
    if (!o.adapt) R.c = c;
    R.acc.assign(1, (int64_t)cnt[0].nacc);
    R.num = (int64_t)cnt[0].num;
    return R;
}
// sspdmp(∇ϕ, t0, x0, θ0, T, c, F::ZigZag, κ, args...; reversible, strong_upperbounds, factor=1.5, adapt)  -- src/ss_fact.jl:159-217
template <class Target>
Result<FactTrace> sspdmp(const Target& target, double t0, const std::vector<double>& x0, const std::vector<double>& theta0,
                         double T, const std::vector<double>& c, const ZigZag& F, const std::vector<double>& kappa,
                         Options o = {}) {
    if (o.factor == 1.8) o.factor = 1.5;
    return detail::factorised(PDMP_SAMPLER_STICKY_ZIGZAG, target, t0, x0, theta0, T, c, F, &kappa, o);
}
// pdmp(∇ϕ!, t0, x0, θ0, T, c, B::BouncyParticle; adapt, factor=2.0) with ∇ϕ!(y, x) = B.Γ(x − B.μ)  -- src/not_fact_samplers.jl:117-147
inline Result<PDMPTrace> pdmp(double t0, const std::vector<double>& x0, const std::vector<double>& theta0, double T, double c,
                              const BouncyParticle& B, Options o = {}) {
    if (o.factor == 1.8) o.factor = 2.0;
    const int64_t d = (int64_t)x0.size();
    const int64_t cap = o.trace_capacity > 0 ? o.trace_capacity : std::max<int64_t>(256, (int64_t)(64 * std::max(T - t0, 1.0)));
    Ensemble e(1, d, PDMP_SAMPLER_BPS, o, cap);
    check(pdmp_ensemble_set_flow_bps(e.get(), B.Gamma.colptr.data(), B.Gamma.rowval.data(), B.Gamma.nzval.data(),
                                     detail::opt(B.mu), B.lambda_ref, B.rho));
    if (B.L.n > 0) check(pdmp_ensemble_set_mass_cholesky(e.get(), B.L.colptr.data(), B.L.rowval.data(), B.L.nzval.data()));
    if (o.local_bound || o.subsample) check(pdmp_ensemble_set_bps_options(e.get(), o.local_bound ? 1 : 0, o.subsample ? 1 : 0));
    return detail::not_factorised(e, t0, x0, theta0, T, c, o);
}
// pdmp(∇ϕ!, t0, x0, θ0, T, c, B::BouncyParticle; ...) with a target of its own, ∇ϕ!(y, x) = Γt(x − μt): ab(…GlobalBound…) keeps B.Γ, B.μ
// (src/not_fact_samplers.jl:26-28), gradient / rate / reflection use the target's (:122)
inline Result<PDMPTrace> pdmp(const GaussianTarget& target, double t0, const std::vector<double>& x0, const std::vector<double>& theta0,
                              double T, double c, const BouncyParticle& B, Options o = {}) {
    if (o.factor == 1.8) o.factor = 2.0;
    const int64_t d = (int64_t)x0.size();
    const int64_t cap = o.trace_capacity > 0 ? o.trace_capacity : std::max<int64_t>(256, (int64_t)(64 * std::max(T - t0, 1.0)));
    Ensemble e(1, d, PDMP_SAMPLER_BPS, o, cap);
    check(pdmp_ensemble_set_flow_bps(e.get(), B.Gamma.colptr.data(), B.Gamma.rowval.data(), B.Gamma.nzval.data(),
                                     detail::opt(B.mu), B.lambda_ref, B.rho));
    check(pdmp_ensemble_set_target_gaussian_csc(e.get(), target.Gamma.colptr.data(), target.Gamma.rowval.data(),
                                                target.Gamma.nzval.data(), detail::opt(target.mu)));
    if (B.L.n > 0) check(pdmp_ensemble_set_mass_cholesky(e.get(), B.L.colptr.data(), B.L.rowval.data(), B.L.nzval.data()));
    if (o.local_bound || o.subsample) check(pdmp_ensemble_set_bps_options(e.get(), o.local_bound ? 1 : 0, o.subsample ? 1 : 0));
    return detail::not_factorised(e, t0, x0, theta0, T, c, o);
}
// pdmp(∇ϕ!, t0, x0, θ0, T, c, B::Boomerang; ...) with ∇ϕ!(y, x) = Γt(x − μt)  -- test/maintest.jl:139-154
inline Result<PDMPTrace> pdmp(const GaussianTarget& target, double t0, const std::vector<double>& x0,
                              const std::vector<double>& theta0, double T, double c, const Boomerang& B, Options o = {}) {
    if (o.factor == 1.8) o.factor = 2.0;
    const int64_t d = (int64_t)x0.size();
    const int64_t cap = o.trace_capacity > 0 ? o.trace_capacity : std::max<int64_t>(256, (int64_t)(64 * std::max(T - t0, 1.0)));
    Ensemble e(1, d, PDMP_SAMPLER_BPS, o, cap);
    check(pdmp_ensemble_set_flow_boomerang(e.get(), target.Gamma.colptr.data(), target.Gamma.rowval.data(),
                                           target.Gamma.nzval.data(), detail::opt(target.mu), detail::opt(B.mu), B.lambda_ref,
                                           B.rho));
    if (B.L.n > 0) {
        check(pdmp_ensemble_set_mass_cholesky(e.get(), B.L.colptr.data(), B.L.rowval.data(), B.L.nzval.data()));
    } else {  // (an empty L is this struct's "identity": the ABI wants it spelled out for a Boomerang)
        std::vector<int64_t> cp((size_t)d + 1), rv((size_t)d);
        std::vector<double> nz((size_t)d, 1.0);
        for (int64_t k = 0; k < d; ++k) cp[(size_t)k] = rv[(size_t)k] = k;
        cp[(size_t)d] = d;
        check(pdmp_ensemble_set_mass_cholesky(e.get(), cp.data(), rv.data(), nz.data()));
    }
    if (o.subsample) check(pdmp_ensemble_set_bps_options(e.get(), 0, 1));
    return detail::not_factorised(e, t0, x0, theta0, T, c, o);
}

// ---- the one-dimensional samplers: pdmp(∇ϕ, x, θ, T, c, Flow::Union{ZigZag1d, Boomerang1d}; adapt, factor = 2.0) -> Ξ, acc/num
// (src/zigzagboom1d.jl:34-67).  ∇ϕ(x) = (x − μ)/σ² + noise (rand() − 0.5) (test/test1d.jl:9-10).
struct ZigZag1d {};
struct Boomerang1d {
    double Sigma = 1.0, mu = 0.0, lambda_ref = 1.0;  // Boomerang1d(Σ, μ, λ), src/types.jl:89-100
};
struct GaussianTarget1d {
    double mu = 0.0, sigma2 = 1.0, noise = 0.0;
};
struct Result1d {
    std::vector<pdmp_event1d> trace;  // Ξ: (t, x, θ), the first entry (0, x0, θ0)
    double acceptance;                // acc/num
    double c;                         // the tuning parameter after adaptation
};
namespace detail {
inline std::vector<Result1d> run_1d(pdmp_1d_config cfg, const std::vector<double>& x0, const std::vector<double>& theta0, double T, double c,
                                    const Options& o) {
    const size_t n = x0.size();
    cfg.struct_size = (uint32_t)sizeof(pdmp_1d_config);
    cfg.device = o.device;
    cfg.adapt = o.adapt ? 1 : 0;
    cfg.factor = (o.factor == 1.8) ? 2.0 : o.factor;  // (the 1-d driver's default, :34)
    cfg.nchains = (int64_t)n;
    cfg.trace_capacity = o.trace_capacity > 0 ? o.trace_capacity : 4096;
    std::vector<pdmp_1d_state> st(n);
    std::vector<uint64_t> seeds(n);
    for (size_t k = 0; k < n; ++k) {
        st[k] = pdmp_1d_state{};
        st[k].x = x0[k];
        st[k].theta = theta0[k];
        st[k].c = c;
        seeds[k] = o.seed + k;
    }
    std::vector<pdmp_event1d> ev(n * (size_t)cfg.trace_capacity);
    std::vector<int64_t> nev(n);
    std::vector<Result1d> out(n);
    for (;;) {
        check(pdmp_1d_run(&cfg, st.data(), seeds.data(), T, ev.data(), nev.data()));
        bool again = false;
        for (size_t k = 0; k < n; ++k) {
            out[k].trace.insert(out[k].trace.end(), ev.begin() + (ptrdiff_t)(k * (size_t)cfg.trace_capacity),
                                ev.begin() + (ptrdiff_t)(k * (size_t)cfg.trace_capacity + (size_t)nev[k]));
            if (st[k].status == PDMP_CHAIN_BOUND_VIOLATED) throw std::runtime_error("Tuning parameter `c` too small.");  // :55
            again = again || st[k].status == PDMP_CHAIN_TRACE_FULL || st[k].status == PDMP_CHAIN_PAUSED;
        }
        if (!again) break;
    }
    for (size_t k = 0; k < n; ++k) {
        out[k].acceptance = st[k].num ? (double)st[k].acc / (double)st[k].num : 0.0;
        out[k].c = st[k].c;
    }
    return out;
}
}  // namespace detail
inline std::vector<Result1d> pdmp(const GaussianTarget1d& g, const std::vector<double>& x0, const std::vector<double>& theta0, double T, double c,
                                  const ZigZag1d&, const Options& o = {}) {
    pdmp_1d_config cfg{};
    cfg.flow = PDMP_1D_ZIGZAG;
    cfg.mu = g.mu, cfg.sigma2 = g.sigma2, cfg.noise = g.noise;
    cfg.b_sigma = 1.0, cfg.b_mu = 0.0, cfg.b_lambda = 1.0;
    return detail::run_1d(cfg, x0, theta0, T, c, o);
}
inline std::vector<Result1d> pdmp(const GaussianTarget1d& g, const std::vector<double>& x0, const std::vector<double>& theta0, double T, double c,
                                  const Boomerang1d& B, const Options& o = {}) {
    pdmp_1d_config cfg{};
    cfg.flow = PDMP_1D_BOOMERANG;
    cfg.mu = g.mu, cfg.sigma2 = g.sigma2, cfg.noise = g.noise;
    cfg.b_sigma = B.Sigma, cfg.b_mu = B.mu, cfg.b_lambda = B.lambda_ref;
    return detail::run_1d(cfg, x0, theta0, T, c, o);
}

}  // namespace pdmp

#endif  // PDMP_MI355_HPP
