/*
 * pdmp_mi355.h -- C ABI of libpdmp_mi355.so, the MI355X (gfx950) ensemble engine for the PDMP event loop.
 *
 * The reference (ZigZagBoomerang.jl @ v0.13.2) has no FFI: its boundary is the Julia method signatures
 *     spdmp(∇ϕ, t0, x0, θ0, T, c, [G,] F::ZigZag, args...; factor, adapt, seed) -> Ξ,(t,x,θ),(acc,num),c
 *                                                               (src/sfact.jl:162-163,211,214)
 *     pdmp (∇ϕ, ...same...)                = spdmp(..., All(), ...)   (src/sfact.jl:236)
 *     pdmp (∇ϕ!, t0,x0,θ0,T,c, B::BouncyParticle; ...)               (src/not_fact_samplers.jl:117,395)
 *     sspdmp(∇ϕ, t0,x0,θ0,T,c, [G,] F::ZigZag, κ, args...; ...)       (src/ss_fact.jl:159-160,217)
 * Each entry point below names the reference lines it replaces.  A Julia `ccall` shim that keeps those
 * signatures is shown in INTEGRATION.md; the Python mirror lives in zigzagboomerang.jl_amd/samplers.py.
 *
 * Conventions
 *   - plain C, caller-owned HOST pointers unless a name ends in `_dev`; the library never frees caller
 *     memory and never keeps a caller pointer after the call returns;
 *   - coordinates are 0-based (the Julia shim adds 1); matrices are CSC with int64 colptr/rowval,
 *     rows ascending inside a column (SparseArrays layout, src/common.jl:17-20);
 *   - return codes, never exceptions; per-chain status words replace the reference's `error(...)`
 *     (src/sfact.jl:124): one diverged chain never aborts the ensemble;
 *   - one ensemble is bound to one HIP device; calls on one ensemble must be serialised by the caller,
 *     different ensembles are independent (no globals except the thread-local error string; the library reads no
 *     environment variables -- diagnostics are per-ensemble calls declared in pdmp_debug.h);
 *   - a user gradient closure cannot cross a C ABI: targets are an enumerated set (Gaussian CSC now).
 *   - randomness: chain k draws from Philox4x32-10 keyed by seeds[k]; draw order is the reference's
 *     `rand(rng)` call order (include/pdmp_detmath.h, oracle/pdmp_oracle.h).
 */
#ifndef PDMP_MI355_H
#define PDMP_MI355_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: pdmp_ensemble_run_partitioned takes the mask's length; pdmp_ensemble_path_integrals, the trace consumers, the G argument
 *    (pdmp_ensemble_set_neighbourhood), pdmp_ensemble_set_target_bps and the pdmp_comm_* (RCCL) entry points; Boomerang ensembles
 *    need an explicit mass factor like BouncyParticle ones.
 * 3: pdmp_ensemble_consume_discretized no longer clamps *npoints (it reports the grid points the trace reaches; row 0 is always x0),
 *    pdmp_ensemble_gather_bps_traces / pdmp_comm_gathered_bps_copy / pdmp_ensemble_bps_trace_dev exist, and ensembles of at most seven chains per compute unit (1792 on an MI355X)
 *    run the tracked local ZigZag with a helper wavefront per chain (same results; include/pdmp_debug.h: pdmp_debug_set_helper_wave).
 *    A host binding must check pdmp_abi_version() at load time. */
#define PDMP_ABI_VERSION 3

typedef enum {
    PDMP_OK = 0,
    PDMP_ERR_INVALID = 1,     /* bad argument / call order */
    PDMP_ERR_NO_DEVICE = 2,   /* no gfx950 HIP device: the engine has NO CPU fallback */
    PDMP_ERR_HIP = 3,         /* a HIP runtime call failed; see pdmp_last_error() */
    PDMP_ERR_UNSUPPORTED = 4, /* configuration outside what the kernels implement */
    PDMP_ERR_NOMEM = 5
} pdmp_status;

/* sampler = which reference driver the ensemble reproduces */
#define PDMP_SAMPLER_ZIGZAG_LOCAL 0 /* spdmp, G = Matched()      src/sfact.jl:73-145,214 */
#define PDMP_SAMPLER_ZIGZAG_ALL 1   /* pdmp,  G = All()          src/sfact.jl:236        */
#define PDMP_SAMPLER_BPS 2          /* pdmp,  BouncyParticle     src/not_fact_samplers.jl:52-147 */
#define PDMP_SAMPLER_STICKY_ZIGZAG 3 /* sspdmp                   src/ss_fact.jl:78-215   */

/* per-chain status words */
#define PDMP_CHAIN_OK 0
#define PDMP_CHAIN_BOUND_VIOLATED 1 /* l >= lb with adapt = false (reference: error, src/sfact.jl:124) */
#define PDMP_CHAIN_STALLED 2        /* queue minimum is +Inf */
#define PDMP_CHAIN_TRACE_FULL 3     /* trace buffer full: drain with pdmp_ensemble_trace_* and run again */
#define PDMP_CHAIN_PAUSED 4         /* a launch counts its draws and proposals in 32 bits: a chain that has used 3 * 2^30 of them inside ONE call of
                                     * pdmp_ensemble_run (~10^9 proposals: only without a trace buffer) stops there, nothing to drain -- run again,
                                     * the run continues exactly (ABI 3) */

/* run flags */
#define PDMP_RUN_REFERENCE_TAIL 0 /* `while t′ < T` (src/sfact.jl:199): the last event has t′ >= T        */
#define PDMP_RUN_STOP_BEFORE 1    /* pause BEFORE popping a key >= T (slice boundary; resumable, exact)  */

typedef struct pdmp_ensemble pdmp_ensemble;

typedef struct {
    uint32_t struct_size; /* sizeof(pdmp_config), for ABI growth */
    int32_t device;       /* HIP device ordinal */
    int32_t sampler;      /* PDMP_SAMPLER_* */
    int32_t adapt;        /* kwarg adapt  (src/sfact.jl:163) */
    double factor;        /* kwarg factor (src/sfact.jl:163: 1.8; BPS 2.0; sticky 1.5) */
    int64_t nchains;      /* ensemble width on THIS device */
    int64_t d;            /* dimension */
    int64_t trace_capacity; /* events kept per chain (0: count only, no trace) */
} pdmp_config;

/* One FactTrace event (t[i], i, x[i], θ[i]) after the flip: src/sfact.jl:50-52, src/trace.jl:38. 32 bytes. */
typedef struct {
    double t;
    int64_t i; /* 0-based */
    double x;
    double theta;
} pdmp_event;

/* Per-chain counters (the reference returns (acc, num); src/sfact.jl:211). */
typedef struct {
    double t_last;      /* t′ of the last processed event/proposal */
    uint64_t num;       /* proposals                (num, src/sfact.jl:120) */
    uint64_t nacc;      /* accepted reflections     (sum(acc), :122)        */
    uint64_t nrefresh;  /* refresh events           (:78-114)               */
    uint64_t ntrace;    /* events currently in the trace buffer             */
    uint64_t nevents;   /* events emitted since set_state (never reset)     */
    uint64_t ndraw_main;
    uint64_t ndraw_global;
    uint32_t status;    /* PDMP_CHAIN_* */
    uint32_t reserved;
} pdmp_chain_counters;

const char* pdmp_last_error(void);
int pdmp_abi_version(void);
/* number of usable gfx950 devices (0 if none / HIP not initialisable) */
int pdmp_device_count(void);

pdmp_status pdmp_ensemble_create(const pdmp_config* cfg, pdmp_ensemble** out);
void pdmp_ensemble_destroy(pdmp_ensemble* ens);

/*
 * Flow Z = ZigZag(Γ, μ, σ; λref, ρ) (src/types.jl:19-27).  Γ is the BOUNDING precision used by ab()
 * (src/fact_samplers.jl:50-54); its column pattern defines G1 (src/sfact.jl:170) and G2 (:178).
 * sigma may be NULL (ones).  lambda_ref > 0 enables the refresh clock (src/sfact.jl:78-114,188-190).
 */
pdmp_status pdmp_ensemble_set_flow_zigzag(pdmp_ensemble* ens, const int64_t* colptr, const int64_t* rowval,
                                          const double* nzval, const double* mu, const double* sigma,
                                          double lambda_ref, double rho);

/*
 * Flow F = FactBoomerang(Γ, μ, λref, σ; ρ) (src/types.jl:71-79) for spdmp: Hamiltonian rotation about μ between events
 * (src/sfact.jl:29-36), rate ((∇ϕ_i − (x_i−μ_i)Γ_ii)θ_i)⁺ (src/fact_samplers.jl:37-39), bound a = c_i√(x_i²+θ_i²)·z + (x_i²+θ_i²)Γ_ii,
 * b = 0 (:58-65), mandatory refresh θ_i = ρθ_i + ρ̄σ_i·randn at rate λref > 0 (src/sfact.jl:103, src/fact_samplers.jl:18).
 * Same arguments as set_flow_zigzag; sampler must be PDMP_SAMPLER_ZIGZAG_LOCAL (the factorised driver spdmp).
 */
pdmp_status pdmp_ensemble_set_flow_factboomerang(pdmp_ensemble* ens, const int64_t* colptr, const int64_t* rowval,
                                                 const double* nzval, const double* mu, const double* sigma,
                                                 double lambda_ref, double rho);

/*
 * The neighbourhood argument of spdmp(∇ϕ, t0, x0, θ0, T, c, G, F, ...) / sspdmp(∇ϕ, t0, x0, θ0, T, c, G, F, κ, ...) (src/sfact.jl:162,171-179;
 * src/ss_fact.jl:159,167-172): G[i] = g_rowval[g_colptr[i] .. g_colptr[i+1]) (ascending) is what a proposal of i moves before the gradient is
 * taken (:82) and what the gradient may read (:116); G1[i] = the pattern of column i of the flow's Γ stays what is re-bounded (:131-135), and
 * G2[i] = ∪_{j ∈ G1[i]} G1[j] \ G[i] what an accepted event moves on top (:178).  G[i] ⊇ G1[i] is required (the reference's @assert, :177):
 * PDMP_ERR_INVALID otherwise.  Without this call G = Matched() = G1.  Call after set_flow_* and BEFORE set_target_* (the target's pattern
 * may then use all of G); such ensembles run on the general-neighbourhood kernel.  Not for PDMP_SAMPLER_ZIGZAG_ALL (G = All()).
 */
pdmp_status pdmp_ensemble_set_neighbourhood(pdmp_ensemble* ens, const int64_t* g_colptr, const int64_t* g_rowval);

/*
 * Target ∇ϕ(x, i) = Γt[:,i]·x  [ − Γt[:,i]·μt ]  (idot, src/common.jl:16-24; closure of
 * scripts/gaussianrandomfield.jl:25, test/maintest.jl:9).  The pattern of Γt must be contained in the
 * flow's Γ pattern (the reference reads only x[j], j in G[i]: src/sfact.jl:116).  mu may be NULL.
 * Must be called after set_flow_zigzag.
 */
pdmp_status pdmp_ensemble_set_target_gaussian_csc(pdmp_ensemble* ens, const int64_t* colptr,
                                                  const int64_t* rowval, const double* nzval,
                                                  const double* mu);

/*
 * Target of config C4: subsampled logistic regression with a control variate at μ, evaluated with SelfMoving():
 *   ∇ϕmoving(t,x,θ,i,t′,F,A,At,μ,y,ny,k) = γ0*x[i] − fdot_moving(A,At,i,t,x,θ,t′,F,μ,y,ny,k)   (scripts/logistic.jl:78-95,107,167)
 * A is the n x p design in CSC (column = coordinate), At = A' in CSC (column = observation), y / ny the per-observation
 * success / failure counts, k_sub the subsample size; the k_sub row indices of every proposal are draws of the chain's
 * PDMP_STREAM_GLOBAL stream (the reference takes them from Julia's global rng).  Must follow set_flow_zigzag.
 */
pdmp_status pdmp_ensemble_set_target_logistic(pdmp_ensemble* ens, int64_t n, const int64_t* A_colptr,
                                              const int64_t* A_rowval, const double* A_nzval, const int64_t* At_colptr,
                                              const int64_t* At_rowval, const double* At_nzval, const double* y,
                                              const double* ny, const double* mu, double gamma0, int64_t k_sub);

/*
 * Initial state (src/sfact.jl:164-190): x0, theta0 are [nchains x d] row-major; c is [d] (copied; with
 * adapt each chain gets its own copy, returned by final_state -- no hidden mutation of caller memory,
 * unlike src/fact_samplers.jl:68); seeds is [nchains].  Computes b[i] = ab(...) and the initial queue
 * Q[i] = poisson_time(b[i], rand(rng)) ON THE DEVICE, drawing the first d uniforms of each chain in
 * order i = 0..d-1 (:184-187; like the reference, t0 is not added to the initial keys).
 * For a ZigZag ensemble that fills the device (>= 1024 chains, >= 2 GB of records) the call also chooses WHERE the records and the queue's level 0
 * lie: it times a short launch, re-allocates the two arrays a few times and keeps the fastest placement (0.03-0.1 s; the state handed back is
 * the one described above, bit for bit: DESIGN.md 5 "The timing modes are a property of the allocation"; pdmp_debug_set_placement turns it off).
 */
pdmp_status pdmp_ensemble_set_state(pdmp_ensemble* ens, double t0, const double* x0, const double* theta0,
                                    const double* c, const uint64_t* seeds);

/*
 * Synthetic initial state generated on the device (benchmarks): seeds[k] = seed0 + k,
 * x0[k][i] = randn (Philox stream PDMP_STREAM_INIT, draw i), theta0[k][i] = ±1 (draw d+i)
 * -- the shape of scripts/gaussianrandomfield.jl:29-30.
 */
pdmp_status pdmp_ensemble_set_state_synthetic(pdmp_ensemble* ens, double t0, const double* c, uint64_t seed0);

/*
 * Advance every chain to time T (the `while t′ < T` driver loop, src/sfact.jl:199-208, around
 * spdmp_inner!, :73-145).  Asynchronous on `stream` (a hipStream_t, NULL = the ensemble's own stream);
 * re-entrant: calling run again with a larger T continues the same event sequence.
 */
pdmp_status pdmp_ensemble_run(pdmp_ensemble* ens, double T, int flags, void* stream);
pdmp_status pdmp_ensemble_sync(pdmp_ensemble* ens);
/*
 * parallel_spdmp (src/parallel.jl:104-253): every chain advanced by K wavefronts -- the coordinates cut into K chunks of d / K
 * (Partition(nt, n), :26), one worker per chunk (parallel_spdmp_inner!, :63-102, horizon Δ = `delta`) and a coordinator for the
 * coordinates whose neighbourhood leaves their chunk (parallel_spdmp_outer!, :176-253).  PDMP_SAMPLER_ZIGZAG_LOCAL on a Gaussian target,
 * lambda_ref = 0; `adapt` as configured.  G = the pattern of the flow tables; G1 = the slots with g1_mask[p] != 0 (NULL: all of them;
 * mask_len must equal the number of stored entries of the Γ given to set_flow_zigzag, else PDMP_ERR_INVALID), the
 * structural pattern of the bounding Γ -- pass the bounding Γ on the union pattern with zeros and mask them out when, as in
 * test/testparallel.jl:40-49, it is the target's Γ without the cross-chunk entries.  "Upper bounds may not depend across chunks." (:124-127)
 * returns PDMP_ERR_INVALID.  Starts from a fresh state only (set_state, then ONE call); blocks until done.  The trace holds the events in
 * the order the waves emitted them: sort by time as the reference does (:167); a buffer that is too small sets PDMP_CHAIN_TRACE_FULL and the
 * trace is incomplete.  Counters: num / nacc as the reference returns them, nrefresh = coordinator rounds.  Bit-identical to the oracle's
 * threaded restatement (tests/test_gpu_partitioned.py); 1 <= K <= 16, K | d, columns of at most 64 entries.
 */
pdmp_status pdmp_ensemble_run_partitioned(pdmp_ensemble* ens, double T, int K, double delta, const uint8_t* g1_mask, int64_t mask_len,
                                          void* stream);
/* duration of the most recent run's event-loop kernel, from HIP events recorded on its stream (syncs) */
pdmp_status pdmp_ensemble_last_run_ms(pdmp_ensemble* ens, float* ms);

pdmp_status pdmp_ensemble_counters(pdmp_ensemble* ens, pdmp_chain_counters* out /* [nchains] */);
/* sum over chains of num / nacc / nevents, reduced on the host from the counters */
pdmp_status pdmp_ensemble_totals(pdmp_ensemble* ens, uint64_t* num, uint64_t* nacc, uint64_t* nevents);

/* trace access (FactTrace.events, src/trace.jl:7-13): copy events [first, first+count) of one chain */
pdmp_status pdmp_ensemble_trace_copy(pdmp_ensemble* ens, int64_t chain, int64_t first, int64_t count,
                                     pdmp_event* out);
/* forget buffered events (after draining them); clears PDMP_CHAIN_TRACE_FULL */
pdmp_status pdmp_ensemble_trace_reset(pdmp_ensemble* ens);

/*
 * Final state of chains [chain_first, chain_first+n): per-coordinate clocks t, positions x, velocities θ
 * (NOT advanced to T, src/sfact.jl:210-211), accept counts acc (int64, :182) and bounds c.  Any output
 * pointer may be NULL.  All arrays are [n x d] row-major.
 */
pdmp_status pdmp_ensemble_final_state(pdmp_ensemble* ens, int64_t chain_first, int64_t n, double* t, double* x,
                                      double* theta, int64_t* acc, double* c);

/*
 * Path integrals for ESS / moments (no counterpart in the reference; the integrand is the one of
 * mean(trace), src/trace.jl:182-200).  Requires all chains paused with PDMP_RUN_STOP_BEFORE at time T.
 * Adds, for this batch [T_prev, T], Y = (1/ΔT) ∫ x_i dt per chain and accumulates over chains:
 *   sum_y[i] += Y, sum_y2[i] += Y*Y   (device reduction; outputs are host [d] arrays, may be NULL)
 */
pdmp_status pdmp_ensemble_batch_means(pdmp_ensemble* ens, double T_prev, double T, double* sum_y, double* sum_y2);

/*
 * Effective sample size by batch means INSIDE each chain (no counterpart in the reference; integrand as above).  All chains
 * paused with PDMP_RUN_STOP_BEFORE at the times named:
 *   ess_begin(T0)   after burn-in: snapshots the path integrals J_i(T0) of every chain;
 *   ess_batch(T)    closes the batch (T_last, T]: Y = (J(T) − J(T_last))/(T − T_last) per chain and coordinate, device sums
 *                   ΣY and ΣY² over chains and batches (call it B times with equally long batches);
 *   ess_end(...)    the chain means over the whole run, M = (J(T_B) − J(T0))/(T_B − T0): ΣM and ΣM² over chains; returns all
 *                   four [d] sums, the number of batches and the run's end points.
 * With N chains, B batches of length b:  within-chain  S_w = ΣY² − B·ΣM²,  σ²_asym = b·S_w/(N(B−1))  (pooled over chains),
 * between-chain  σ²_asym ≈ B·b·(ΣM² − (ΣM)²/N)/(N−1)  (valid at stationarity),  ESS_i = N·B·b·Var_π,i/σ²_asym,i
 * (zigzagboomerang.jl_amd/ess.py).  ZigZag flows only (a FactBoomerang path rotates between events: PDMP_ERR_UNSUPPORTED, as for
 * pdmp_ensemble_batch_means).
 */
pdmp_status pdmp_ensemble_ess_begin(pdmp_ensemble* ens, double T0);
pdmp_status pdmp_ensemble_ess_batch(pdmp_ensemble* ens, double T);
pdmp_status pdmp_ensemble_ess_end(pdmp_ensemble* ens, double* sum_y, double* sum_y2, double* sum_m, double* sum_m2,
                                  int64_t* nbatches, double* T0, double* T1);

/*
 * The engine keeps ∫ x_i dt per chain and coordinate next to the state (what batch_means / ess_* / path_integrals read; the reference has no
 * such thing).  enable = 0 drops it: those calls then return PDMP_ERR_INVALID, and the kernels that hold a chain's state on chip (small d:
 * the subsampled logistic target, pdmp_logistic.hip) fit 13 chains per CU instead of 8 (config C4, d = 442: 11.9 against 18.7 KB of LDS per chain; measured 105 against 133 ms per step).  Default: kept.  Call before set_state.
 */
pdmp_status pdmp_ensemble_set_path_integrals(pdmp_ensemble* ens, int enable);

/*
 * J_i(T) = ∫_{t0}^{T} x_i(s) ds of EVERY chain at `nprobe` probe coordinates (0-based): out is [nchains x nprobe] row-major.  All chains
 * paused with PDMP_RUN_STOP_BEFORE at T.  Differences of successive calls are per-chain batch integrals, from which the host forms any
 * ESS estimator (batch means at several batch lengths, between-chain variances: zigzagboomerang.jl_amd/ess.py: multiscale_ess) while the
 * N x d records stay on the device.  ZigZag flows only, like pdmp_ensemble_batch_means.
 */
pdmp_status pdmp_ensemble_path_integrals(pdmp_ensemble* ens, double T, int64_t nprobe, const int64_t* probes, double* out);

/* ------------------------------------------------------------------ sticky ZigZag (PDMP_SAMPLER_STICKY_ZIGZAG)
 *
 * sspdmp(∇ϕ, t0, x0, θ0, T, c, F::ZigZag, κ, args...; reversible=false, strong_upperbounds=false, factor=1.5, adapt)
 * (src/ss_fact.jl:159-160,217).  Call after set_flow_zigzag / set_target_gaussian_csc and before set_state: kappa is the
 * [d] vector of thaw rates.  Events are FactTrace events (freeze, thaw and reflection alike, :154); the counters' nacc/num
 * are the reference's scalar (acc, num) (:175,214).  The refresh clock is not implemented by the reference (:86).
 */
pdmp_status pdmp_ensemble_set_sticky(pdmp_ensemble* ens, const double* kappa, int reversible, int strong_upperbounds);

/* ------------------------------------------------------------------ adaptscale (σ tuning in the refresh branch)
 *
 * spdmp(...; adaptscale=true) (src/sfact.jl:74,86-99,163): each refresh of coordinate i first retunes F.σ[i] -- ZigZag:
 * log σ[i] is pulled towards 0.3 accepted reflections per unit time and θ[i] = σ[i]·sign(θ[i]) is set WITHOUT a random draw
 * (:87-91); FactBoomerang: σ[i] *= exp(±0.03·min(1, √(τ/λref))) once τ = (1+2ρ/(1−ρ))/(t[i]·λref) < 0.2 (:93-98).  The
 * reference mutates F.σ in place; here σ becomes per-chain device state, initialised from the flow's sigma at set_state
 * and read back with pdmp_ensemble_final_sigma ([n x d]).  Needs a refresh clock (lambda_ref > 0), PDMP_SAMPLER_ZIGZAG_LOCAL
 * and the Gaussian target.  Call after set_flow_* and before set_state.  x^y is evaluated as pdmp_exp(y·pdmp_log x).
 */
pdmp_status pdmp_ensemble_set_adaptscale(pdmp_ensemble* ens, int enable);
pdmp_status pdmp_ensemble_final_sigma(pdmp_ensemble* ens, int64_t chain_first, int64_t n, double* sigma);

/* ------------------------------------------------------------------ tracked gradients (an evaluation strategy, not a sampler)
 *
 * enable = 1: the local ZigZag loop keeps, per coordinate, g_i = Γ[:,i]·x and gd_i = Γ[:,i]·θ instead of moving G[i] and gathering them at
 * every proposal (src/sfact.jl:82,116): a proposal then touches its own record only and an accepted reflection its G1 neighbours
 * -- about half the HBM traffic of the moving evaluation.  Same process, same draws, same thinning decisions: the event INDEX sequence,
 * accept / reject outcomes, (acc, num) and adapted bounds equal the reference's exactly; event times, positions and the final
 * (t, x, θ) agree to ~1e-13 relative instead of bit for bit, because sums that are advanced are not rounded like sums that are
 * recomputed (tests/test_gpu_track_parity.py: 1e-9; the default, enable = 0, stays bit-identical).  "Exactly" holds until a float
 * difference of that size flips an accept test or the order of two nearly simultaneous events -- measured on the north-star workload: one chain
 * in 4096 after 1.7e9 proposals, about 6e-10 per proposal (tools/track_soak.py); from there on that chain is another realisation of the same
 * process (the draws land on different events), not a wrong one.  pdmp_ensemble_final_state rebuilds
 * the reference's lazy clocks t[j] (src/sfact.jl:211) from the times of the last proposal / accept around j.
 * Requirements (else set_state returns PDMP_ERR_UNSUPPORTED -- never a silent fall-back): PDMP_SAMPLER_ZIGZAG_LOCAL, ZigZag flow
 * without refresh, Gaussian target, symmetric Γ, lattice-like neighbourhoods (|G1| <= 5, |S| <= 13) and 2048 <= d <= 16384.
 * With the LOGISTIC target (config C4; the same requirements on the flow, G = Matched(), a state that fits the LDS-resident kernel) the
 * BOUNDS are tracked and the subsampled gradient stays the moving evaluation: a proposal moves coordinate i and what its sampled rows read,
 * a rejection re-derives its bound from (g_i, gd_i, tg_i), an accepted event updates the k members of G1[i] instead of moving its two-hop
 * set and summing every member's column afresh.  The oracle's spdmp_zigzag_tracked_lg states it sequentially; the device equals it bit for
 * bit; the final clocks t[j] are the tracked process's own (a coordinate is as old as its last own event or visit by a sampled row).
 * Call before set_state.
 */
pdmp_status pdmp_ensemble_set_gradient_tracking(pdmp_ensemble* ens, int enable);

/* ------------------------------------------------------------------ c::LocalBound (src/local.jl)
 *
 * spdmp(∇ϕ, t0, x0, θ0, T, C::LocalBound, F::ZigZag, args...) (src/local.jl:95-149): the bound of coordinate j is built from the
 * target's own directional derivatives, b[j] = (c_j + ∇ϕj·θ_j, c_j/100 + vj, 2/c_j/|θ_j|) (:2-6) with (∇ϕj, vj) =
 * (Γt[:,j]·x − Γt[:,j]·μt, θ_j·Γt[:,j]·θ) for the Gaussian target (the `(∇ϕi, vi)` callback of performance/smartbound.jl:45-59),
 * and expires after its horizon: next_time (src/not_fact_samplers.jl:43-50) queues min(proposal, horizon) and a `renew` flag;
 * a renew event re-bounds j without a thinning step (:36-43).  The queue is initialised with t0 + τ (:122).  ZigZag flow without
 * refresh clock, Gaussian target whose pattern equals the flow's, PDMP_SAMPLER_ZIGZAG_LOCAL.  Call after set_flow / set_target,
 * before set_state; `c` of set_state is then LocalBound(c).c.
 * Ties: coordinates with equal c_i/|θ_i| re-bounded at the same instant get EXACTLY equal horizon keys; the reference pops tied
 * keys in heap order (src/priorityqueue.jl:46-77), this engine by lowest index.  Both are valid orders of simultaneous events of
 * independent clocks, but the random streams then pair differently: bit parity with the reference needs distinct c_i/|θ_i|.
 */
pdmp_status pdmp_ensemble_set_local_bound(pdmp_ensemble* ens, int enable);

/* ------------------------------------------------------------------ Bouncy particle sampler (PDMP_SAMPLER_BPS)
 *
 * pdmp(∇ϕ!, t0, x0, θ0, T, c, B::BouncyParticle; adapt, factor=2.0) -> Ξ::PDMPTrace, (t, x, θ), (acc, num), c
 * (src/not_fact_samplers.jl:117-147,395-396) with GlobalBound(c) and the Gaussian target ∇ϕ!(y, x) = Γ(x − μ)
 * (test/maintest.jl:163).  Γ is B.Γ: it enters ab() (:26-28) and -- unless pdmp_ensemble_set_target_gaussian_csc is called AFTER this
 * (accepted on this family: ∇ϕ!(y, x) = Γt(x − μt) of its own; ab(…GlobalBound…) keeps B.Γ, B.μ, LocalBound takes θ'∇ϕx and θ'Γtθ) -- the target.  The mass factor B.L =
 * cholesky(Symmetric(Γ)).L (src/types.jl:43) is identity for Γ = I (config C2); for any other Γ the caller must hand it over
 * with pdmp_ensemble_set_mass_cholesky before set_state_bps, which otherwise returns PDMP_ERR_UNSUPPORTED.
 * Events are (t, copy(x), copy(θ)) (:39-41); dot products use the fixed summation order stated in oracle/pdmp_oracle.c.
 */
pdmp_status pdmp_ensemble_set_flow_bps(pdmp_ensemble* ens, const int64_t* colptr, const int64_t* rowval,
                                       const double* nzval, const double* mu, double lambda_ref, double rho);

/* Flow = Boomerang(Γ, μ_flow, λref; ρ) (src/types.jl:59-66): Hamiltonian rotation about μ_flow between events
 * (src/dynamics.jl:29-36), grad_correct! ∇ϕx −= L'\(L\(x − μ_flow)) (src/not_fact_samplers.jl:9-12), constant bound
 * (√(‖θ‖² + ‖x − μ_flow‖²)·c, 0, Inf) (:34-36); same pdmp_inner! loop, events, counters and entry points as the bouncy
 * particle (create the ensemble with PDMP_SAMPLER_BPS).  The CSC matrix and mu_target describe the TARGET
 * ∇ϕ!(y, x) = Γt(x − μt) (test/maintest.jl:146).  The flow's own Γ enters only through its factor L = cholesky(Symmetric(Γ)).L
 * (src/types.jl:66), which the caller MUST hand over with pdmp_ensemble_set_mass_cholesky before set_state_bps (an identity factor for
 * Γ = I): without one set_state_bps returns PDMP_ERR_UNSUPPORTED -- never a silent L = I. */
pdmp_status pdmp_ensemble_set_flow_boomerang(pdmp_ensemble* ens, const int64_t* colptr, const int64_t* rowval,
                                             const double* nzval, const double* mu_target, const double* mu_flow,
                                             double lambda_ref, double rho);
/*
 * Mass factor F.L of BouncyParticle / Boomerang (src/types.jl:43,66): a LOWER-triangular d x d matrix in CSC form (rows
 * ascending, hence the diagonal entry first in its column; non-zero diagonal).  Used by reflect!
 * θ .-= (2⟨∇ϕx,θ⟩/‖L\∇ϕx‖²)·L'\(L\∇ϕx) (src/dynamics.jl:90-97), refresh! θ = ρθ + ρ̄·L'\randn(d) (:112-126) and Boomerang's
 * grad_correct! (src/not_fact_samplers.jl:9-12).  The reference factorises inside the constructor (dense LAPACK, or CHOLMOD with
 * its fill-reducing permutation for a sparse Γ, whose `.L` is the factor of the PERMUTED matrix); the caller -- the Julia shim
 * passes `sparse(B.L)` -- owns that choice, the engine only solves with what it is given, by column-oriented substitution in
 * the order oracle/pdmp_oracle.c states.  Call after set_flow_bps / set_flow_boomerang and before set_state_bps.
 */
pdmp_status pdmp_ensemble_set_mass_cholesky(pdmp_ensemble* ens, const int64_t* colptr, const int64_t* rowval,
                                            const double* nzval);
/*
 * local_bound: `c` of set_state_bps is LocalBound(c).c -- ab = (c + ⟨θ,∇ϕx⟩, v, 2√d/c/‖θ‖₂) with v = θ'Γθ (the second
 * directional derivative a `(∇ϕx, v)` gradient callback returns for the Gaussian target), next_time's horizon and the renew
 * branch of pdmp_inner! (src/not_fact_samplers.jl:29-31,43-50,65-71); BouncyParticle only.
 * subsample: kwarg `subsample` (:53,90): an accepted reflection does not end pdmp_inner!, only refreshments are recorded.
 * Call after set_flow_* and before set_state_bps.
 */
pdmp_status pdmp_ensemble_set_bps_options(pdmp_ensemble* ens, int local_bound, int subsample);
/* x0, theta0: [nchains x d]; c: the scalar bound constant (GlobalBound(c) or LocalBound(c)); seeds: [nchains] */
pdmp_status pdmp_ensemble_set_state_bps(pdmp_ensemble* ens, double t0, const double* x0, const double* theta0, double c,
                                        const uint64_t* seeds);
/* events [first, first+count) of one chain: t [count], x and theta [count x d] (any may be NULL) */
pdmp_status pdmp_ensemble_bps_trace_copy(pdmp_ensemble* ens, int64_t chain, int64_t first, int64_t count, double* t,
                                         double* x, double* theta);
/* final (t, x, θ) and adapted c of chains [chain_first, chain_first+n): t, c are [n]; x, theta are [n x d] */
pdmp_status pdmp_ensemble_bps_final_state(pdmp_ensemble* ens, int64_t chain_first, int64_t n, double* t, double* x,
                                          double* theta, double* c);

/* ------------------------------------------------------------------ trace consumers on the device (what callers do next with Ξ)
 *
 * mean(Ξ) (src/trace.jl:182-200) and collect(discretize(Ξ, dt)) (:94-125) of every chain's FactTrace, computed from the engine's trace
 * buffer slice by slice, so that a trace set far larger than the buffer (config C3: 8 MB per chain x 4096) is consumed without ever leaving
 * the device:
 *   consume_begin(grid_dt, grid_points)  after set_state, before the first run: snapshots (t0, x0, θ0) as every coordinate's cursor;
 *                                        grid_points > 0 reserves [nchains x grid_points x d] doubles for the grid t0 + k·grid_dt;
 *   consume()                            after EVERY run slice and before trace_reset: applies the buffered events (each coordinate's in
 *                                        order) -- the trapezoid sums of mean, the grid points a finished segment covers;
 *   consume_mean(chain_first, n, ...)    mean [n x d] and the last event time T of each chain (the reference's scale 1/(2T));
 *   consume_inclusion(chain_first, n, ...)  inclusion_prob [n x d] (:161-178: time with x_i ≠ 0 before or after an event of i, over T);
 *   consume_discretized(chain, k_first, k_count, out, npoints, grid_dev)
 *                                        rows k_first .. of chain's grid positions [k_count x d] (row k belongs to time t0 + k grid_dt; row 0 is
 *                                        x0); *npoints = number of grid times the reference would emit so far (those before the chain's last
 *                                        event; at least t0) -- NOT clamped: a value above grid_points means the grid of consume_begin was too
 *                                        short and the later rows were dropped; *grid_dev = the whole device array, for consumers that stay on the
 *                                        device.  Any of the three may be NULL.
 * Time-ordered traces of piecewise-linear paths only: ZigZag flow without refresh clock (spdmp, pdmp, sspdmp).  Values are those of
 * zigzagboomerang.jl_amd/trace.py (discretize: bitwise; mean: the same sums scaled once instead of term by term).
 */
pdmp_status pdmp_ensemble_consume_begin(pdmp_ensemble* ens, double grid_dt, int64_t grid_points);
pdmp_status pdmp_ensemble_consume(pdmp_ensemble* ens);
/* The same beside the sampler (ABI 3): what the last pdmp_ensemble_run (on `stream`, 0 = the ensemble's own) wrote is consumed on a second stream of
 * the ensemble and the trace segments are handed back EMPTY at once -- no pdmp_ensemble_trace_reset; the next run writes the other of two trace
 * buffers while this slice is consumed, and waits only for the consumer of the slice before it.  The reference returns Ξ (src/sfact.jl:211) and
 * callers then run discretize / mean over it (src/trace.jl:106-125,182-200): at the sampler's rate (32 B per event) a C3 trace exceeds what PCIe
 * drains, so this is how a full-rate run is USED.  Returns without waiting; consume_mean / _inclusion / _discretized and every entry point that
 * reads state wait for the consumers.  pdmp_ensemble_last_consume_ms: kernel time of the last asynchronous consumer (waits for it). */
pdmp_status pdmp_ensemble_consume_async(pdmp_ensemble* ens, void* stream);
pdmp_status pdmp_ensemble_last_consume_ms(pdmp_ensemble* ens, float* ms);
pdmp_status pdmp_ensemble_consume_mean(pdmp_ensemble* ens, int64_t chain_first, int64_t n, double* mean, double* T_last);
/* inclusion_prob(Ξ) (src/trace.jl:161-178) per chain: the fraction of [t0, T] each coordinate spent away from 0 -- what a sticky run is made for */
pdmp_status pdmp_ensemble_consume_inclusion(pdmp_ensemble* ens, int64_t chain_first, int64_t n, double* prob, double* T_last);
pdmp_status pdmp_ensemble_consume_discretized(pdmp_ensemble* ens, int64_t chain, int64_t k_first, int64_t k_count, double* out,
                                              int64_t* npoints, void** grid_dev);
/* cummean(Ξ) (src/trace.jl:203-226) on the device (round 6): once enabled (after consume_begin; the synchronous consumer), pdmp_ensemble_consume also leaves,
 * for every event of the segment it consumes, the running pair (t_i, Σ (x_prev + x_k)(t_k − t_prev) / (2 t_i)) of the event's coordinate -- the sums are
 * carried from segment to segment, so the pairs are those of the whole run's trace (bit for bit the reference's order of operations per coordinate).
 * _copy: the pairs of slots [first, first + count) of a chain's segment, the slots pdmp_ensemble_trace_copy returns the events of. */
pdmp_status pdmp_ensemble_consume_cummean(pdmp_ensemble* ens, int enable);
pdmp_status pdmp_ensemble_consume_cummean_copy(pdmp_ensemble* ens, int64_t chain, int64_t first, int64_t count, double* t, double* y);
/* subtrace(Ξ, J) (src/trace.jl:275-290) on the device (round 6): the events of a chain's current trace segment whose coordinate lies in the ascending
 * 0-based index set J, renumbered by their position in J, compacted on the device; n_out = how many there are (at most out_cap are written to `out`). */
pdmp_status pdmp_ensemble_subtrace_copy(pdmp_ensemble* ens, int64_t chain, const int64_t* J, int64_t nJ, pdmp_event* out, int64_t out_cap,
                                        int64_t* n_out);

/* what the ensemble was created with (any pointer may be NULL) */
pdmp_status pdmp_ensemble_info(pdmp_ensemble* ens, int64_t* nchains, int64_t* d, int64_t* trace_capacity, int* device);

/* ------------------------------------------------------------------ the post-run exchange of a sharded ensemble (RCCL over xGMI)
 *
 * Chains are independent: rank r of R runs its own block of them with NO collective (one process, one ensemble, one communicator per GPU).
 * What follows a run (nothing in the reference -- it is single-process):
 *   pdmp_comm_unique_id     on ONE rank: an opaque PDMP_COMM_ID_BYTES token (ncclGetUniqueId) the host hands to the other ranks by its own means
 *                           (MPI.jl, a file, a socket: zigzagboomerang.jl_amd/parallel.py uses a TCP rendezvous on MASTER_ADDR);
 *   pdmp_comm_init          on every rank, collectively (ncclCommInitRank on `device`);
 *   pdmp_comm_barrier / pdmp_comm_allreduce (host doubles, PDMP_COMM_SUM | PDMP_COMM_MAX): timing and counter reductions of a benchmark;
 *   pdmp_ensemble_gather_traces   collective.  Every rank learns nchains_by_rank [world] and counts (events in the trace buffer of every
 *                           chain of the whole ensemble, rank-major; counts_cap entries available).  The trace segments travel to `root`:
 *                           ncclAllGather of the counts -> each rank compacts its segments on the device -> ONE grouped ncclSend / ncclRecv
 *                           (a gatherv: every peer streams over its own direct xGMI link).  On root the events of all chains lie back to back in
 *                           rank-major, chain-major order in a device buffer the communicator owns (*events_dev, valid until the next gather on
 *                           it or pdmp_comm_destroy) and, if events_host != NULL, are copied there (events_cap events available);
 *   pdmp_ensemble_reduce_moments  collective: pdmp_ensemble_batch_means on every rank, ncclReduce(sum) of the 2 d sums onto root's sum_y, sum_y2.
 * Calls on one communicator are serialised by the caller.  world = 1 is valid (the same code path on one GPU).
 */
#define PDMP_COMM_ID_BYTES 128
#define PDMP_COMM_SUM 0
#define PDMP_COMM_MAX 1
typedef struct pdmp_comm pdmp_comm;
pdmp_status pdmp_comm_unique_id(void* id, int64_t id_bytes);
pdmp_status pdmp_comm_init(const void* id, int rank, int world, int device, pdmp_comm** out);
void pdmp_comm_destroy(pdmp_comm* comm);
pdmp_status pdmp_comm_info(const pdmp_comm* comm, int* rank, int* world);
pdmp_status pdmp_comm_barrier(pdmp_comm* comm);
pdmp_status pdmp_comm_allreduce(pdmp_comm* comm, double* inout, int64_t n, int op);
pdmp_status pdmp_ensemble_gather_traces(pdmp_ensemble* ens, pdmp_comm* comm, int root, int64_t* nchains_by_rank, uint64_t* counts,
                                        int64_t counts_cap, pdmp_event* events_host, int64_t events_cap, void** events_dev,
                                        int64_t* nevents_total);
/* events [first, first + count) of what the last pdmp_ensemble_gather_traces left on this rank (root), device -> host */
pdmp_status pdmp_comm_gathered_copy(pdmp_comm* comm, pdmp_event* out, int64_t first, int64_t count);
pdmp_status pdmp_ensemble_reduce_moments(pdmp_ensemble* ens, pdmp_comm* comm, int root, double T_prev, double T, double* sum_y,
                                         double* sum_y2);
/* The same exchange for the PDMPTrace of the non-factorised samplers (pdmp on BouncyParticle / Boomerang: events (t, copy(x), copy(θ)),
 * src/not_fact_samplers.jl:39-41, 8 (2 d + 1) bytes each).  On root the events of all chains lie rank-major, chain-major in three device arrays the
 * communicator owns -- t [total], x [total x d], θ [total x d] -- valid until the next gather on it; pdmp_comm_gathered_bps_copy fetches a range
 * (any of t / x / theta may be NULL).  Collective; argument errors of ONE rank (a counts buffer that is too small, a failed allocation) are agreed
 * on before anything is sent, so every rank returns the error and the communicator stays usable -- the same holds for
 * pdmp_ensemble_gather_traces. */
pdmp_status pdmp_ensemble_gather_bps_traces(pdmp_ensemble* ens, pdmp_comm* comm, int root, int64_t* nchains_by_rank, uint64_t* counts,
                                            int64_t counts_cap, void** t_dev, void** x_dev, void** theta_dev, int64_t* nevents_total);
pdmp_status pdmp_comm_gathered_bps_copy(pdmp_comm* comm, double* t, double* x, double* theta, int64_t first, int64_t count);

/* ------------------------------------------------------------------ the one-dimensional samplers
 *
 * pdmp(∇ϕ, x, θ, T, c, Flow::Union{ZigZag1d, Boomerang1d}; adapt = false, factor = 2.0) -> Ξ::Vector{(t, x, θ)}, acc/num
 * (src/zigzagboom1d.jl:34-67; ab :15-16, λ :5-6, move_forward src/dynamics.jl:66-68,79-82) for an ensemble of independent chains, one
 * chain per LANE.  The gradient is the device-resident form of the reference's own test closure (test/test1d.jl:9-10):
 * ∇ϕ(x) = (x − mu)/sigma2 + noise·(rand() − 0.5).  Draw order: the reference's calls on its global generator, in program order, are the
 * draws of the chain's main stream.  One call runs every chain until t >= T, a bound violation (adapt = 0: PDMP_CHAIN_BOUND_VIOLATED, the
 * reference's error(...)) or a full event buffer (PDMP_CHAIN_TRACE_FULL: call again with the returned state -- the run continues exactly
 * where it stopped).  state[k] on the first call: {x, theta, c} = the start, started = 0, everything else 0.  events: host buffer
 * [nchains x trace_capacity], nevents[k] of chain k's row are valid (the first one of a fresh run is (0, x0, θ0), :36).  Host pointers only.
 */
#define PDMP_1D_ZIGZAG 0
#define PDMP_1D_BOOMERANG 1
typedef struct {
    uint32_t struct_size; /* sizeof(pdmp_1d_config) */
    int32_t device;
    int32_t flow;  /* PDMP_1D_ZIGZAG: ZigZag1d(); PDMP_1D_BOOMERANG: Boomerang1d(b_sigma, b_mu, b_lambda) (src/types.jl:82-100) */
    int32_t adapt; /* c *= factor on a violated bound instead of stopping (:54-56) */
    double factor;
    int64_t nchains;
    int64_t trace_capacity;
    double mu, sigma2, noise;        /* the target's gradient (test/test1d.jl:5-10) */
    double b_sigma, b_mu, b_lambda; /* Boomerang1d's Σ, μ, λref */
} pdmp_1d_config;
typedef struct {
    double t, x, theta;
} pdmp_event1d;
typedef struct {
    double t, x, theta, c; /* time, position, velocity, tuning parameter (grows under adapt) */
    double a, b, t_next, t_ref; /* the bound in force, the next proposal and refresh times */
    uint64_t ndraw;             /* draws of the main stream consumed */
    int64_t num, acc;           /* proposals, accepted proposals (:38,51,53) */
    int32_t started, status;    /* status: PDMP_CHAIN_* */
} pdmp_1d_state;
pdmp_status pdmp_1d_run(const pdmp_1d_config* cfg, pdmp_1d_state* state /* [nchains], in/out */, const uint64_t* seeds /* [nchains] */,
                        double T, pdmp_event1d* events /* [nchains x trace_capacity] */, int64_t* nevents /* [nchains] */);

/* raw device pointers for zero-copy consumers (e.g. an RCCL gather of trace segments) */
pdmp_status pdmp_ensemble_trace_dev(pdmp_ensemble* ens, void** events_dev, int64_t* capacity);
pdmp_status pdmp_ensemble_counters_dev(pdmp_ensemble* ens, void** counters_dev);
/* ... of a BouncyParticle / Boomerang ensemble: event times [nchains x capacity], positions and velocities [nchains x capacity x d] */
pdmp_status pdmp_ensemble_bps_trace_dev(pdmp_ensemble* ens, void** t_dev, void** x_dev, void** theta_dev);

#ifdef __cplusplus
}
#endif
#endif /* PDMP_MI355_H */
