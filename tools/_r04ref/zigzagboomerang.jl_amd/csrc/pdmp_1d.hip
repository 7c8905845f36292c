// pdmp_1d.hip -- the reference's one-dimensional samplers (src/zigzagboom1d.jl:34-67: pdmp(∇ϕ, x, θ, T, c, ::ZigZag1d / ::Boomerang1d)) as an
// ensemble: ONE CHAIN PER LANE.  A 1-d chain is a handful of scalars, so there is nothing to spread over a wavefront; 64 independent
// chains share an instruction stream and diverge where their branches do (refresh vs proposal, accept vs reject).  The gradient is the one of
// the reference's own test (test/test1d.jl:9-10): ∇ϕ(x) = (x − μ)/σ² + noise (rand() − 0.5); every random number is a draw of the chain's MAIN
// stream in the program order of the reference's global generator (see oracle/pdmp_oracle.c: orc_pdmp_1d, which this equals bit for bit).
// State is handed in and out, so a run continues after its event buffer filled up.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#include "../../include/pdmp_detmath.h"
#include "../../include/pdmp_mi355.h"

extern "C" void pdmp_set_last_error_(const char* msg);  // pdmp_capi.hip

namespace {

#define D1_INF __builtin_inf()

__device__ __forceinline__ double d1_pos(double x) {  // pos(x), src/common.jl:8
    return (x > 0.0) ? x : ((x != x) ? x : 0.0);
}
// poisson_time(a, b, u), src/poissontime.jl:8-30
__device__ __forceinline__ double d1_poisson_time(double a, double b, double u) {
    const double L = pdmp_log(u);
    if (b > 0) {
        const double r = a / b;
        if (a < 0) return sqrt(-L * 2.0 / b) - r;
        return sqrt(r * r - L * 2.0 / b) - r;
    } else if (b == 0) {
        return (a > 0) ? -L / a : D1_INF;
    } else {
        if (a <= 0) return D1_INF;
        if (-L <= -(a * a) / b + (a * a) / (2 * b)) {
            const double r = a / b;
            return -sqrt(r * r - L * 2.0 / b) - r;
        }
        return D1_INF;
    }
}

struct D1Params {
    pdmp_1d_config cfg;
    double T;
    pdmp_1d_state* st;
    const uint64_t* seeds;
    pdmp_event1d* ev;
    int64_t* nev;
};

__global__ __launch_bounds__(64) void pdmp1d_run_kernel(D1Params P) {
    const int64_t chain = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (chain >= P.cfg.nchains) return;
    const bool boom = P.cfg.flow == PDMP_1D_BOOMERANG;
    const uint64_t seed = P.seeds[chain];
    const double bmu = P.cfg.b_mu, bsig = P.cfg.b_sigma, blam = P.cfg.b_lambda;
    pdmp_1d_state S = P.st[chain];
    pdmp_event1d* const out = P.ev + chain * P.cfg.trace_capacity;
    const int64_t cap = P.cfg.trace_capacity;
    double t = S.t, x = S.x, th = S.theta, c = S.c, a = S.a, b = S.b, tp = S.t_next, t_ref = S.t_ref;
    uint64_t nm = S.ndraw;
    int64_t num = S.num, acc = S.acc, n = 0;
    auto push = [&](double t_, double x_, double th_) {
        out[n].t = t_;
        out[n].x = x_;
        out[n].theta = th_;
        n += 1;
    };
    auto bound = [&]() {  // ab(x, θ, c, Flow), :15-16
        if (boom) {
            a = sqrt(th * th + (x - bmu) * (x - bmu)) * c;
            b = 0.0;
        } else {
            a = c + th * x;
            b = th * th;
        }
    };
    int32_t status = PDMP_CHAIN_OK;
    if (!S.started) {
        t = 0.0;  // :35
        if (cap > 0) push(t, x, th);  // :36
        t_ref = boom ? t + pdmp_randexp_from_u(pdmp_u01(seed, PDMP_STREAM_MAIN, nm++)) / blam : D1_INF;  // :19-20,37
        bound();
        tp = t + d1_poisson_time(a, b, pdmp_u01(seed, PDMP_STREAM_MAIN, nm++));  // :40
        S.started = 1;
    }
    while (t < P.T) {  // :41
        if (n >= cap) {
            status = PDMP_CHAIN_TRACE_FULL;
            break;
        }
        if (t_ref < tp) {  // :42: refresh (Boomerang1d only: a ZigZag1d's t_ref is +Inf)
            const double tau = t_ref - t;
            double sn, cs;
            pdmp_sincos(tau, &sn, &cs);  // move_forward, src/dynamics.jl:79-82
            const double xn = (x - bmu) * cs + th * sn + bmu;
            t = t + tau;
            x = xn;
            th = sqrt(bsig) * pdmp_randn(seed, PDMP_STREAM_MAIN, nm++);                      // :44
            t_ref = t + pdmp_randexp_from_u(pdmp_u01(seed, PDMP_STREAM_MAIN, nm++)) / blam;  // :45
            push(t, x, th);                                                                  // :46
        } else {
            const double tau = tp - t;  // :48
            if (boom) {
                double sn, cs;
                pdmp_sincos(tau, &sn, &cs);
                const double xn = (x - bmu) * cs + th * sn + bmu, tn = -(x - bmu) * sn + th * cs;
                x = xn;
                th = tn;
                t = t + tau;
            } else {
                t = tau + t;  // src/dynamics.jl:66-68
                x = x + th * tau;
            }
            double gx = (x - P.cfg.mu) / P.cfg.sigma2;                                                        // test/test1d.jl:9
            if (P.cfg.noise != 0.0) gx = gx + P.cfg.noise * (pdmp_u01(seed, PDMP_STREAM_MAIN, nm++) - 0.5);  // :10
            const double l = boom ? d1_pos(th * (gx - (x - bmu) / bsig)) : d1_pos(th * gx);  // λ, :5-6
            const double lb = d1_pos(a + b * tau);                                           // λ_bar, :9,50
            num += 1;
            if (pdmp_u01(seed, PDMP_STREAM_MAIN, nm++) * lb < l) {  // :52
                acc += 1;
                const bool violated = l >= lb;  // :54
                if (violated && !P.cfg.adapt) {
                    status = PDMP_CHAIN_BOUND_VIOLATED;  // error("Tuning parameter `c` too small."), :55
                    break;
                }
                c = violated ? c * P.cfg.factor : c;  // :56
                th = -th;                             // :58
                push(t, x, th);                       // :60
            }
        }
        bound();                                                                  // :63
        tp = t + d1_poisson_time(a, b, pdmp_u01(seed, PDMP_STREAM_MAIN, nm++));  // :64
    }
    S.t = t;
    S.x = x;
    S.theta = th;
    S.c = c;
    S.a = a;
    S.b = b;
    S.t_next = tp;
    S.t_ref = t_ref;
    S.ndraw = nm;
    S.num = num;
    S.acc = acc;
    S.status = status;
    P.st[chain] = S;
    P.nev[chain] = n;
}

pdmp_status d1_fail(pdmp_status st, const char* what, const char* detail) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s%s%s", what, detail ? ": " : "", detail ? detail : "");
    pdmp_set_last_error_(buf);
    return st;
}

template <class T>
struct D1Buf {
    T* p = nullptr;
    hipError_t alloc(size_t n) { return hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)); }
    ~D1Buf() {
        if (p) (void)hipFree(p);
    }
};

}  // namespace

extern "C" pdmp_status pdmp_1d_run(const pdmp_1d_config* cfg, pdmp_1d_state* state, const uint64_t* seeds, double T, pdmp_event1d* events,
                                   int64_t* nevents) {
    if (!cfg || !state || !seeds || !nevents) return d1_fail(PDMP_ERR_INVALID, "pdmp_1d_run: null argument", nullptr);
    if (cfg->struct_size != sizeof(pdmp_1d_config)) return d1_fail(PDMP_ERR_INVALID, "pdmp_1d_run: pdmp_1d_config.struct_size mismatch", nullptr);
    if (cfg->nchains < 1 || cfg->trace_capacity < 1 || !events)
        return d1_fail(PDMP_ERR_INVALID, "pdmp_1d_run: nchains >= 1 and an event buffer of trace_capacity >= 1 per chain are needed", nullptr);
    if (cfg->flow != PDMP_1D_ZIGZAG && cfg->flow != PDMP_1D_BOOMERANG) return d1_fail(PDMP_ERR_INVALID, "pdmp_1d_run: unknown flow", nullptr);
    if (!(cfg->sigma2 > 0) || (cfg->flow == PDMP_1D_BOOMERANG && (!(cfg->b_sigma > 0) || !(cfg->b_lambda > 0))))
        return d1_fail(PDMP_ERR_INVALID, "pdmp_1d_run: sigma2 > 0 (and Boomerang1d's Σ > 0, λref > 0) required", nullptr);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || cfg->device < 0 || cfg->device >= ndev)
        return d1_fail(PDMP_ERR_NO_DEVICE, "pdmp_1d_run: no such gfx950 device (the engine has no CPU fallback)", nullptr);
#define D1_TRY(expr)                                                                  \
    do {                                                                              \
        const hipError_t e_ = (expr);                                                 \
        if (e_ != hipSuccess) return d1_fail(PDMP_ERR_HIP, #expr, hipGetErrorString(e_)); \
    } while (0)
    D1_TRY(hipSetDevice(cfg->device));
    const size_t n = (size_t)cfg->nchains, cap = (size_t)cfg->trace_capacity;
    D1Buf<pdmp_1d_state> d_st;
    D1Buf<uint64_t> d_seed;
    D1Buf<pdmp_event1d> d_ev;
    D1Buf<int64_t> d_n;
    D1_TRY(d_st.alloc(n));
    D1_TRY(d_seed.alloc(n));
    D1_TRY(d_ev.alloc(n * cap));
    D1_TRY(d_n.alloc(n));
    D1_TRY(hipMemcpy(d_st.p, state, n * sizeof(pdmp_1d_state), hipMemcpyHostToDevice));
    D1_TRY(hipMemcpy(d_seed.p, seeds, n * sizeof(uint64_t), hipMemcpyHostToDevice));
    D1Params P;
    P.cfg = *cfg;
    P.T = T;
    P.st = d_st.p;
    P.seeds = d_seed.p;
    P.ev = d_ev.p;
    P.nev = d_n.p;
    hipLaunchKernelGGL(pdmp1d_run_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, P);
    D1_TRY(hipGetLastError());
    D1_TRY(hipDeviceSynchronize());
    D1_TRY(hipMemcpy(state, d_st.p, n * sizeof(pdmp_1d_state), hipMemcpyDeviceToHost));
    D1_TRY(hipMemcpy(nevents, d_n.p, n * sizeof(int64_t), hipMemcpyDeviceToHost));
    D1_TRY(hipMemcpy(events, d_ev.p, n * cap * sizeof(pdmp_event1d), hipMemcpyDeviceToHost));
#undef D1_TRY
    return PDMP_OK;
}
