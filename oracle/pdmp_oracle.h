/*
 * pdmp_oracle.h -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded, line-by-line restatement of the event-loop hot path of
 * ZigZagBoomerang.jl (reference @ v0.13.2, /root/reference).  It exists to check the gfx950 kernels:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * library (libpdmp_mi355.so) never links, loads or calls anything in this directory.
 *
 * PARITY STATUS: "parity unpinned" at the bit level.  The reference is Julia; no Julia toolchain is
 * present in the build image, the reference tests hold no golden event sequences and do not pin the
 * RNG stream.  What IS pinned against the reference's own tests:
 *   - orc_poisson_time / orc_poisson_time3 against the analytic identity of test/poisson.jl:9-49;
 *   - the indexed heap against deterministic push/pop/change-key properties (test/priority.jl style);
 *   - the samplers against the statistical envelopes of test/maintest.jl:32-33,59-60,170-171,
 *     test/test1d.jl:18-27 and test/sticky.jl:30-34.
 *
 * Conventions that differ from the reference on purpose (documented draw order):
 *   - coordinates are 0-based;
 *   - every `rand(rng)` of the reference is draw #n (n = 0,1,2,...) of the chain's Philox stream
 *     PDMP_STREAM_MAIN (include/pdmp_detmath.h); draws the reference takes from Julia's GLOBAL rng
 *     (src/sfact.jl:80,84,108) are draw #m of PDMP_STREAM_GLOBAL;
 *   - log is pdmp_log, randexp is -pdmp_log(u), randn is Box-Muller (pdmp_randn).
 */
#ifndef PDMP_ORACLE_H
#define PDMP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* CSC sparse matrix, 0-based, rows ascending inside a column (SparseArrays layout, src/common.jl:17-20). */
typedef struct {
    int64_t n;
    const int64_t* colptr; /* n+1 */
    const int64_t* rowval; /* nnz */
    const double* nzval;   /* nnz */
} orc_csc;

/* One FactTrace event: (t[i], i, x[i], theta[i]) AFTER the flip (src/sfact.jl:50-52, src/trace.jl:38). */
typedef struct {
    double t;
    int64_t i;
    double x;
    double theta;
} orc_event;

typedef struct {
    orc_event* ev;
    int64_t n;
    int64_t cap;
} orc_trace;

void orc_trace_init(orc_trace* tr);
void orc_trace_free(orc_trace* tr);

/* src/poissontime.jl:8-30 and :39-65 */
double orc_poisson_time(double a, double b, double u);
double orc_poisson_time3(double a, double b, double c, double u);
/* src/common.jl:16-24 */
double orc_idot(const orc_csc* A, int64_t j, const double* x);

/* Indexed binary min-heap, src/priorityqueue.jl:9-117.  Keys 0..n-1 must be enqueued in order. */
typedef struct orc_pq orc_pq;
orc_pq* orc_pq_new(int64_t capacity);
void orc_pq_free(orc_pq* q);
void orc_pq_enqueue(orc_pq* q, int64_t key, double val);
void orc_pq_set(orc_pq* q, int64_t key, double val);
double orc_pq_get(const orc_pq* q, int64_t key);
void orc_pq_peek(const orc_pq* q, int64_t* key, double* val);
int64_t orc_pq_len(const orc_pq* q);
int orc_pq_check(const orc_pq* q); /* 1 if heap order + index map are consistent */

/* status codes shared by the samplers */
#define ORC_OK 0
#define ORC_BOUND_VIOLATED 1 /* l >= lb with adapt=false: the reference throws (src/sfact.jl:124) */
#define ORC_STALLED 2        /* queue minimum is +Inf: no further event can occur               */
#define ORC_TRACE_LIMIT 3    /* max_events reached                                                */
#define ORC_BAD_INPUT 4      /* malformed argument (e.g. a mass factor that is not lower triangular) */

typedef struct {
    /* flow Z = ZigZag(Gamma, mu, sigma; lambda_ref, rho)  (src/types.jl:19-27) */
    const orc_csc* bound_gamma;
    const double* bound_mu;   /* d */
    const double* sigma;      /* d, used by the refresh branch only */
    double lambda_ref;
    double rho;
    /* target: grad phi(x,i) = idot(target_gamma,i,x) [- idot(target_gamma,i,target_mu)] */
    const orc_csc* target_gamma;
    const double* target_mu; /* NULL = no shift (scripts/gaussianrandomfield.jl:25) */
    int move_all;            /* 0: G = Matched() (spdmp, src/sfact.jl:214); 1: G = All() (pdmp, :236) */
    int adapt;
    double factor;
    uint64_t seed;
    int64_t max_events; /* <=0: unlimited */
    int stop_before_T;  /* 0: reference loop `while t' < T` (last event has t' >= T);
                           1: pause BEFORE popping a key >= T (slice boundary used by the device engine) */
    /* target_kind 1: subsampled logistic regression with control variate, evaluated with SelfMoving()/ExtendedForm:
     * ∇ϕmoving(t,x,θ,i,t′,F,A,At,μ,y,ny,k) = γ0*x[i] - fdot_moving(A,At,i,t,x,θ,t′,F,μ,y,ny,k)
     * (scripts/logistic.jl:78-95,107,167; ∇ϕ_ dispatch src/sfact.jl:67-68).  The k subsample indices come from the
     * global rng (PDMP_STREAM_GLOBAL), one draw each. */
    int target_kind;       /* 0: Gaussian (fields above), 1: logistic */
    const orc_csc* lg_A;   /* n x p design, CSC (column = coordinate) */
    const orc_csc* lg_At;  /* p x n = A', CSC (column = observation) */
    const double* lg_y;    /* [n] successes */
    const double* lg_ny;   /* [n] m - y */
    const double* lg_mu;   /* [p] control-variate point μ */
    double lg_gamma0;      /* prior precision γ0 */
    int64_t lg_k;          /* subsample size */
    /* flow_kind 1: F = FactBoomerang(Γ, μ, λref, σ; ρ) (src/types.jl:71-79) instead of ZigZag: Hamiltonian rotation between
     * events (src/sfact.jl:29-36), rate (∇ϕi − (x_i−μ_i)Γ_ii)θ_i (src/fact_samplers.jl:37-39), constant bound a = c_i√z2·z + z2·Γ_ii,
     * b = 0 (:58-65), mandatory refresh θ_i = ρθ_i + ρ̄σ_i·randn (src/sfact.jl:103; hasrefresh, src/fact_samplers.jl:18). */
    int flow_kind;
    /* adaptscale = true (src/sfact.jl:86-99): the refresh branch retunes σ[i] before redrawing θ[i].  ZigZag: Robbins-Monro
     * style update of log σ[i] towards 0.3 accepted reflections per unit time, then θ[i] = σ[i]·sign(θ[i]) WITHOUT a random
     * draw (:87-91); FactBoomerang: σ[i] *= exp(±0.03·min(1, √(τ/λref))) when τ = (1+2ρ/(1−ρ))/(t[i]·λref) < 0.2 (:93-98).
     * σ is then per-chain state: sigma_out (d, may be NULL) receives the final values.  `^` is exp(y·log x) here. */
    int adaptscale;
    double* sigma_out;
    /* c::LocalBound (src/local.jl:2-6,10-78,95-149): bounds from the target's own first and second directional derivatives with
     * an expiry horizon 2/c_i/|θ_i|; ZigZag flow, Gaussian target, no refresh clock (the reference's refresh branch is JointFlow only). */
    int local_bound;
    /* tracked = 1: the tracked-gradient evaluation of the same process (pdmp_oracle.c: spdmp_zigzag_tracked) -- the BITWISE checker of
     * the device's tracked kernels; agrees with the moving evaluation (tracked = 0, the reference's) in every index and to ~1e-13 in the
     * floats until a rounding difference flips a thinning test (measured: ~6e-10 per proposal on config C3). */
    int tracked;
    /* the optional argument G of spdmp / sspdmp (src/sfact.jl:162,171-179; src/ss_fact.jl:159,167-172): column patterns = the G[i] (values
     * unused); NULL = Matched().  G[i] ⊇ G1[i] or the call returns ORC_BAD_INPUT (the reference's @assert). */
    const orc_csc* nbr_G;
} orc_zz_params;

typedef struct {
    int64_t num;       /* proposals */
    int64_t nacc;      /* accepted reflections (sum of acc) */
    int64_t nrefresh;  /* refresh events */
    uint64_t ndraw_main;
    uint64_t ndraw_global;
    double t_last;     /* t' of the last returned event */
    int status;
} orc_zz_result;

/*
 * Local ZigZag, src/sfact.jl:73-145 (spdmp_inner!) under the driver :162-212.
 * x, theta: in = x0, theta0; out = state at the per-coordinate clocks t_out (NOT advanced to T, :210).
 * c: in/out (mutated when adapt, src/fact_samplers.jl:67-70).  acc: per-coordinate accept counts (d).
 * tr may be NULL (count only).
 */
int orc_spdmp_zigzag(int64_t d, const orc_zz_params* p, double t0, double T, double* x, double* theta,
                     double* c, double* t_out, int64_t* acc, orc_trace* tr, orc_zz_result* res);

/* 1-d ZigZag, src/zigzagboom1d.jl:34-67 with grad phi(x) = (x - mu)/sigma2.  Events (t,x,theta). */
typedef struct {
    double t, x, theta;
} orc_event1d;
/* Both 1-d flows of src/zigzagboom1d.jl:34-67 (flow 0: ZigZag1d, 1: Boomerang1d(Σ, μ, λref)) on the target of test/test1d.jl:9-10,
 * ∇ϕ(x) = (x − mu)/sigma2 + noise (rand() − 0.5); resumable through the state (started = 0 on the first call: x, theta, c are the start). */
typedef struct {
    int32_t flow, adapt;
    double factor, mu, sigma2, noise, b_sigma, b_mu, b_lambda;
    uint64_t seed;
} orc_1d_params;
typedef struct {
    double t, x, theta, c, a, b, t_next, t_ref;
    uint64_t ndraw;
    int64_t num, acc;
    int32_t started, status; /* status: 0 ok, 1 bound violated (adapt = 0), 3 event buffer full */
} orc_1d_state;
int64_t orc_pdmp_1d(const orc_1d_params* p, orc_1d_state* st, double T, orc_event1d* out, int64_t cap);
int64_t orc_pdmp_zigzag1d(double mu, double sigma2, double x, double theta, double T, double c, int adapt,
                          double factor, uint64_t seed, orc_event1d* out, int64_t cap, int64_t* acc,
                          int64_t* num);

/*
 * Bouncy particle sampler, src/not_fact_samplers.jl:52-97 under the driver :117-147, GlobalBound(c),
 * Gaussian target grad phi!(y,x) = Gamma*(x - mu_t) written as CSC mat-vec, mass factor L supplied by the
 * caller (mass_L below; NULL = identity, exact for Gamma = I, config C2).
 * Events: t_ev[k], and x_ev/theta_ev rows of length d (copy(x), copy(theta), src/not_fact_samplers.jl:39-41).
 */
typedef struct {
    const orc_csc* gamma; /* B.Gamma used by ab (src/not_fact_samplers.jl:26-28) and by the target */
    const double* mu;     /* B.mu */
    double lambda_ref;
    double rho;
    double c;
    int adapt;
    double factor;
    uint64_t seed;
    int64_t max_events;
    /* flow_kind 1: Flow = Boomerang(I, flow_mu, λref; ρ) (src/types.jl:59-66, L = I): Hamiltonian rotation about flow_mu
     * (src/dynamics.jl:29-36), grad_correct! ∇ϕx −= x − flow_mu (src/not_fact_samplers.jl:9-12), constant bound
     * a = √(‖θ‖² + ‖x − flow_mu‖²)·c, b = 0 (:34-36).  gamma/mu above are then the TARGET's precision and mean. */
    int flow_kind;
    const double* flow_mu;
    /* Mass factor F.L = cholesky(Symmetric(Γ)).L (src/types.jl:43,66) as a lower-triangular CSC matrix (NULL: identity), used by
     * reflect! (src/dynamics.jl:90-97), refresh! (:112-126) and Boomerang's grad_correct! (src/not_fact_samplers.jl:9-12); the
     * substitution order is fixed in pdmp_oracle.c (tri_solve_lower / tri_solve_upper). */
    const orc_csc* mass_L;
    int local_bound; /* c::LocalBound (src/not_fact_samplers.jl:29-31): horizon 2√d/c/‖θ‖ and the renew branch :65-71 (BPS only) */
    int subsample;   /* kwarg subsample (:53,90): an accepted reflection does not end pdmp_inner! */
    /* BouncyParticle with a target of its own: ∇ϕ!(y, x) = target_gamma (x − target_mu) (the caller's, src/not_fact_samplers.jl:122) while
     * ab(…GlobalBound…) keeps the flow's B.Γ, B.μ (:26-28).  NULL: the target is B.Γ(x − B.μ) itself. */
    const orc_csc* target_gamma;
    const double* target_mu;
} orc_bps_params;
typedef struct {
    int64_t num, nacc, nrefresh, nevents;
    uint64_t ndraw_main;
    double t_last;
    double c_out;
    int status;
} orc_bps_result;
int orc_pdmp_bps(int64_t d, const orc_bps_params* p, double t0, double T, double* x, double* theta,
                 double* t_ev, double* x_ev, double* theta_ev, int64_t ev_cap, orc_bps_result* res);

/*
 * Sticky ZigZag, src/ss_fact.jl:78-157 under the driver :159-215 (Gaussian CSC target as above).
 * kappa: d thaw rates.  Events as FactTrace.
 */
typedef struct {
    const orc_csc* bound_gamma;
    const double* bound_mu;
    const orc_csc* target_gamma;
    const double* target_mu;
    const double* kappa;
    int adapt;
    double factor;
    int reversible;
    int strong_upperbounds;
    uint64_t seed;
    int64_t max_events;
    /* logistic != NULL: ∇ϕ = ∇ϕmoving(t, x, θ, i, t′, F, A, At, μ, y, ny, k) with SelfMoving() (scripts/logistic.jl:78-95,107; the
     * sticky script scripts/sticky/sticky_logistic_sparse.jl:83-100,131 has the same helper): only the lg_* fields and `seed` of
     * *logistic are read; target_gamma / target_mu are then ignored.  idot_moving! moves what it reads with smove_forward!, frozen
     * coordinates included (their clock advances, x + 0·dt). */
    const orc_zz_params* logistic;
    const orc_csc* nbr_G; /* the optional argument G (src/ss_fact.jl:159,167-172), as in orc_zz_params; NULL = G1 */
} orc_sticky_params;
int orc_sspdmp_zigzag(int64_t d, const orc_sticky_params* p, double t0, double T, double* x, double* theta,
                      double* c, double* t_out, orc_trace* tr, orc_zz_result* res);

/*
 * CPU baseline driver: run `nchains` independent local-ZigZag chains (same target, per-chain x0/theta0
 * rows, seed = seed0 + chain) on `nthreads` POSIX threads, counting events only.  Returns wall seconds.
 */
double orc_spdmp_zigzag_ensemble(int64_t d, const orc_zz_params* p, double t0, double T, int64_t nchains,
                                 const double* x0, const double* theta0, const double* c, uint64_t seed0,
                                 int nthreads, int64_t* num_total, int64_t* acc_total);

/* Threaded local ZigZag of src/parallel.jl (parallel_spdmp, :104-150): ONE chain on K <= 64 worker threads + a coordinator;
 * timing baseline only (the event order depends on thread interleaving, as in the reference).  The bound's Γ (p->bound_gamma)
 * must be block diagonal over the K chunks of d/K coordinates (:124-127); p->target_gamma supplies G.  tr receives the events
 * (unsorted by time across chunks, like Ξ before the final sort! of :167). */
typedef struct {
    int64_t num, nacc, rounds, spawns;
    double seconds; /* wall time of the threaded section */
    int status;
} orc_par_result;
int orc_parallel_spdmp(int64_t d, const orc_zz_params* p, int K, double delta, double t0, double T, double* x, double* theta,
                       double* c, double* t_out, orc_trace* tr, orc_par_result* res);

/* helpers for the tests: the shared numerical contract evaluated on the host */
void orc_math_probe(uint64_t seed, int64_t n, double* out);
double orc_log(double x);
double orc_u01(uint64_t seed, uint32_t stream, uint64_t n);
double orc_randn(uint64_t seed, uint32_t stream, uint64_t n);
void orc_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void orc_synthetic_state(uint64_t seed, int64_t d, double* x0, double* theta0);

#ifdef __cplusplus
}
#endif
#endif
