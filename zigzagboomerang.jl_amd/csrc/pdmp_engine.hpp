// pdmp_engine.hpp -- internal data layout of libpdmp_mi355.so (not part of the C ABI).
//
// HBM layout (all per-ensemble, one ensemble per device):
//
//   shared, read-only, L2/MALL resident (built once from the flow + target, "neighbourhood program"):
//     colptr[d+1] u32, rowval[nnz] u32      G1[i] = rows of column i of the bounding Γ   (src/sfact.jl:170)
//     bval[nnz] f64, tval[nnz] f64          bounding Γ values; target Γ values aligned to the same slots
//     gmu_b[d], gmu_t[d] f64                idot(Γ,i,μ) constants of ab() / of the target
//     sptr[d+1], sidx[...] u32              S[i] = G1[i] (ascending) followed by G2[i] (ascending, :178)
//     qptr[nnz+1] u32, pos[...] u8          for slot (i,jj): positions inside S[i] of the members of G1[j]
//     selfpos[d] u8                         position of i inside G1[i]
//     c[d], sigma[d] f64
//
//   per chain (chain-major, nothing shared between chains):
//     rec[d]  64-byte records {x, θ, t, I, t_old, a, b, acc}: ONE 64 B sector per touched coordinate
//     keys[dk] f64, dk = 64*nblk            event-time queue level 0 (padded with +Inf; key d = refresh)
//     hdr      128 B                        counters, RNG draw indices, status
//     ev[cap]  32-byte events               FactTrace segment
//
//   LDS (per wavefront = per chain): level-1 of the queue: (min key, argmin) of every 64-key block,
//   plus a 64-slot (x,θ) scratch for the neighbourhood being re-bounded.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/pdmp_mi355.h"

namespace pdmp {

struct alignas(64) ZzRec {
    double x;      // position at the coordinate's own clock t          (src/sfact.jl:168,180)
    double th;     // velocity θ
    double t;      // per-coordinate clock
    double I;      // ∫ x dt accumulated up to t (engine-only, for batch means)
    double t_old;  // time the bound (a,b) was computed                  (src/sfact.jl:169)
    double a, b;   // affine bound a + b (t - t_old)                     (src/fact_samplers.jl:50-54)
    uint64_t acc;  // accepted reflections of this coordinate            (src/sfact.jl:182)
};
static_assert(sizeof(ZzRec) == 64, "record must be one 64-byte sector");

// Record of the tracked-gradient kernel (zz_local_track_kernel): ONE 128-byte line per coordinate, four 32-byte sectors ordered by
// how often they are written -- a rejected proposal of i dirties sector 2 only, an accepted one sectors 0-3 of i and 1-3 of its
// neighbours.  Sector 0 has the field order of ZzRec's first half, so the path-integral kernels read both layouts.
struct alignas(128) TrRec {
    double x, th, tx, I;        // position at its own clock tx (brought up on i's accepts only), velocity, ∫ x dt up to tx
    double g, gd, tg;           // g = Γt[:,i]·x and gd = Γt[:,i]·θ at time tg: ∇ϕ_i(t′) = g + gd (t′ − tg) − (Γt μt)_i
    uint64_t acc;               // accepted reflections of i
    double a, b, t_old, tprop;  // bound (src/fact_samplers.jl:50-54) and the time of i's last proposal
    double gb, gdb, tacc, pad;  // the same sums with the bounding Γ (when it differs from the target's); time of i's last accept
};
static_assert(sizeof(TrRec) == 128, "tracked record must be one 128-byte line");
// The same line as pdmp_trackp.hip uses it: the bound and the proposal time are not stored there (they live with the key), and the two free
// sectors carry what a proposal needs that depends on the coordinate alone -- so that no shared table is read in the event loop.
struct alignas(128) TrRecP {
    double x, th, tx, I;
    double g, gd, tg;
    uint64_t acc;
    double c, c100, gam0, gam1;  // the constant bound c_i, c_i / 100 (src/fact_samplers.jl:50-54), Γ[G1[i], i] in G1's order (up to five on the lattice)
    double gam2, gam3, tacc, gam4;
};
static_assert(sizeof(TrRecP) == 128 && offsetof(TrRecP, tacc) == offsetof(TrRec, tacc), "same line, same tacc slot");
// On a graph that is not the plain lattice (LAT = false) the members of G1[i] cannot be computed from i: the 16 bytes of gam0, gam1 hold
// G1[i] as eight 16-bit ids (ascending, 0xFFFF past the end) -- they arrive with the line a proposal reads anyway, so an accepted event
// can request its members' records at once -- and the values Γ[G1[i], i] come from the shared table ZzTables::gam8 (L2-resident, requested
// side by side with the members' records).

// The LINE layout of the one-proposal-per-lane tracked kernel at full width (pdmp_trackl.hip, round 6): everything a proposal of coordinate i
// reads or writes is in ONE 128-byte line that it shares with its pair mate i ^ 1 -- the queue's level 0 IS the record.  Sector 0: the two
// (key, t_old) pairs (a rejected proposal dirties 16 bytes of it and nothing else); sectors 1, 2: (θ, g, gd, tg) of either coordinate (written
// on accepts in G1[i]); sector 3: what depends on the coordinate alone (c_i, c_i / 100: no table in the event loop).  What only an ACCEPTED
// event of i itself touches -- the position at its own clock, ∫ x dt, the reflection count -- is a 32-byte cold record per coordinate.
struct alignas(128) TrLine {
    double key0, told0, key1, told1;
    double th0, g0, gd0, tg0;
    double th1, g1, gd1, tg1;
    double c0, c100_0, c1, c100_1;
};
static_assert(sizeof(TrLine) == 128, "one line per coordinate pair");
struct alignas(32) TrCold {
    double x, tx, I;
    uint64_t acc;
};
static_assert(sizeof(TrCold) == 32, "one sector per coordinate");

struct alignas(128) DevChain {
    pdmp_chain_counters c;  // 72 bytes, copied out verbatim by pdmp_ensemble_counters
    uint64_t seed;
    double t0;
    double t_event;  // t′ of the last RETURNED event: the loop variable of `while t′ < T` (src/sfact.jl:199)
    double tl_scale;  // pdmp_trackl.hip: quanta per unit time of the chain's event-time wheel (0: not chosen yet), kept from launch to launch
    uint64_t pad[3];
};
static_assert(sizeof(DevChain) == 128, "chain header is 128 bytes");

#if defined(__HIPCC__)
// Wave priority in turns.  A SIMD's arbiter serves its OLDEST wave first: of the chains that share a SIMD for a whole launch (one chain per
// wave, 4 or more per SIMD) the oldest finishes well before the youngest, which then runs out the launch alone.  Every event loop calls
// prio_turn(iteration) at its top: the waves of a SIMD take turns at the top priority (a turn = PDMP_PRIO_TURN iterations; measured on the
// headline workload: 6 % faster than without, turns of 1 .. 1024 iterations within 1 % of each other).
constexpr uint32_t PDMP_PRIO_TURN = 256u;
struct PrioTurn {
    uint32_t slot, it;
    __device__ __forceinline__ PrioTurn() : slot((uint32_t)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)) & 3u), it(0) {}  // HW_ID.wave_id
    __device__ __forceinline__ void step() {
        if ((it & (PDMP_PRIO_TURN - 1u)) == 0u) {
            const uint32_t pr = ((it / PDMP_PRIO_TURN) + slot) & 3u;
            switch (pr) {
                case 0: __builtin_amdgcn_s_setprio(0); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                default: __builtin_amdgcn_s_setprio(3); break;
            }
        }
        it += 1;
    }
};
#endif

struct CoordConst {
    double c, c100;
    uint32_t cp, k;
    double gam[5];  // Γt[G1[i], i] (tval[cp .. cp + k)) where k <= 5, else unused
};
static_assert(sizeof(CoordConst) == 64, "half a line");

// Read-only tables of the local ZigZag kernels (device pointers).
struct ZzTables {
    const uint32_t* __restrict__ colptr;
    const uint32_t* __restrict__ rowval;
    const double* __restrict__ bval;
    const double* __restrict__ tval;
    const double* __restrict__ gmu_b;
    const double* __restrict__ gmu_t;  // nullptr: target has no mean shift
    const uint32_t* __restrict__ sptr;
    const uint32_t* __restrict__ sidx;
    const uint32_t* __restrict__ qptr;
    const uint8_t* __restrict__ pos;
    const uint8_t* __restrict__ selfpos;
    const double* __restrict__ c_shared;
    const double2* __restrict__ c2_shared;  // {c_i, c_i / 100} (the constant bound and its slope, src/sfact.jl:36-39), tracked kernels
    const CoordConst* __restrict__ cc_shared;  // the same with colptr[i] and |G1[i]|: everything a proposal needs that depends on i alone, 64 bytes
    const double* __restrict__ sigma;
    // any symmetric graph with |G1| <= 8 on the one-proposal-per-lane tracked kernel (pdmp_trackp.hip, LAT = false):
    const double* __restrict__ gam8;    // [d][8] Γt[G1[i], i] in G1's order, 0.0 past |G1[i]|
    const uint16_t* __restrict__ nb16;  // [d][8] G1[i] ascending, 0xFFFF past |G1[i]| (d <= 16384)
};

// A launch keeps its draw / proposal counters as 32-bit differences from the header's 64-bit ones: a chain that reaches this many pauses
// (PDMP_CHAIN_PAUSED, resumable like TRACE_FULL with nothing to drain) instead of wrapping them
constexpr uint32_t PDMP_LAUNCH_COUNT_LIMIT = 0xC0000000u;

struct ZzRunParams {
    ZzTables tb;
    ZzRec* rec;
    double* keys;
    DevChain* hdr;
    pdmp_event* ev;
    double* c_chain;  // per-chain bounds when adapt, else nullptr
    double* dbg;  // optional [dbg_cap x 16] per-proposal diagnostics of chain 0 (pdmp_debug_set_proposal_dump), else nullptr
    int64_t dbg_cap;
    const uint64_t* __restrict__ blob;  // [ntemplates x blob_w_pad] neighbourhood programs (layout: pdmp_capi.hip build_blob)
    const uint32_t* __restrict__ tix;   // [d] template of coordinate i (coordinates with the same relative program share one)
    uint32_t blob_w, blob_w_pad, blob_sw, blob_pw, blob_kmax;
    uint32_t common_tix;  // the template most coordinates share (kept in LDS for a whole launch by the 8-event kernel)
    int64_t d;
    int64_t dk;        // padded key count per chain (multiple of 64)
    int64_t trace_cap;
    uint32_t nblk;
    uint32_t nblk_pad;  // nblk rounded up to even (keeps LDS sub-arrays 16-byte aligned)
    double T;
    double factor;
    double lambda_ref;
    int32_t flags;
    int32_t adapt;
    int32_t has_refresh;
    int32_t move_all;  // G = All(): the `pdmp` driver for ZigZag (src/sfact.jl:236)
    int32_t force_spec4;  // diagnostics (pdmp_debug_set_kernel): keep the 4-event kernel where the 8-event one would run
    int32_t track_two_sums;  // tracked-gradient kernel: the bounding Γ differs from the target's (two pairs of sums per coordinate)
    int32_t track_mean;      // zz_local_trackp: 0 no mean, 1 the flow's Γμ in the bounds only, 2 also in the rate (the target has the same Γμ); host-set
    uint32_t count_limit;    // a chain whose launch has used this many draws (proposals: the logistic kernel) pauses: PDMP_LAUNCH_COUNT_LIMIT, or a test's
    uint32_t typ_extra;      // zz_local_trackp: the most frequent |G1[i]| − 1 (the accept chain's first guess; any value is correct)
    double hw_gain, hw_ahead;  // ... its steering of the selection threshold (gain towards a target count) and how far ahead the helper requests lines
    uint32_t hw_target;
    int32_t n_cu;            // (host side) compute units of the device: the launchers' width thresholds are per CU (0: 256)
    int32_t helper_wave;     // zz_local_trackp: the two-wave form (a helper wave per chain: ring of draws + prefetch), for under-occupied launches
    // the line layout (pdmp_trackl.hip): [nchains x dk / 2] TrLine, [nchains x dk] TrCold; null elsewhere
    void* tl_lines;
    void* tl_cold;
    int32_t lattice_n;       // n if the graph is the n x n 5-point lattice in column-major numbering (i = row + n col), else 0
    uint32_t lattice_magic;  // ceil(2^32 / n): column of i = umulhi(i, magic) for i < 2^16
    // per-coordinate tables of zz_local_spec8g_kernel (pdmp_spec8g.inc: |G1| <= 8, |S| <= 32), one 128-byte line per coordinate each, or null
    const uint64_t* __restrict__ g8_line;    // [d][16]: S[i] transposed (4 x u16 per lane) | positions inside S[i] of the members of G1[j], j = G1[i][gl]
    const double* __restrict__ g8_member;    // [d][16]: Γ[G1[j], j] (8) | c_j | Γ[:,j]·μ | -
    const double* __restrict__ g8_gamt;      // [d][8] Γt[G1[i], i] when the target's values differ from the bounding ones, else null
    int32_t g8_gw;                           // lanes per event: 8 (line = 16 words) or 16 (|S| up to 64: line = 24 words)
    // sticky ZigZag (src/ss_fact.jl)
    const double* __restrict__ kappa;  // [d] thaw rates
    double* thf;                       // [nchains x d] saved speeds θf
    int32_t reversible, strong_upperbounds;
};

struct ZzInitParams {
    ZzTables tb;
    ZzRec* rec;
    double* keys;
    DevChain* hdr;
    double* c_chain;
    const double* x0;  // [nchains x d] staging (nullptr: synthetic)
    const double* th0;
    const uint64_t* seeds;  // [nchains] (nullptr: seed0 + chain)
    uint64_t seed0;
    int64_t d;
    int64_t dk;
    int64_t nchains;
    double t0;
    double lambda_ref;
    int32_t has_refresh;
    int32_t sticky;  // src/ss_fact.jl:178-188: initial key = min(reflection proposal, hitting time of 0), flag f[i]
    double* thf;
    int32_t flow_kind;               // 1: FactBoomerang bound ab (src/fact_samplers.jl:58-65)
    const double* __restrict__ mu;   // [d]
    const double* __restrict__ diag; // [d]
    int32_t local_bound;             // c::LocalBound (src/local.jl): bounds from the target's derivatives + expiry horizon; thf = renew flags
    int32_t track;                   // records are TrRec (tracked-gradient kernel): also g = Γt[:,i]·x0, gd = Γt[:,i]·θ0 and the bound's sums
};

// One chain over K wavefronts (pdmp_partition.hip): the reference's parallel_spdmp (src/parallel.jl)
struct ZzPartParams {
    ZzTables tb;
    ZzRec* rec;
    double* keys;
    DevChain* hdr;
    pdmp_event* ev;
    double* c_chain;                       // per-chain bounds when adapt, else nullptr
    const uint8_t* __restrict__ inner;     // [d] G[i] lies inside i's chunk (:114)
    const uint8_t* __restrict__ g1mask;    // [nnz] the slot is a structural entry of the bounding Γ (G1, :117)
    const uint32_t* __restrict__ g2ptr;    // [d + 1], [..]: G2[i] = two-hop(G1) \ G[i], ascending (:121)
    const uint32_t* __restrict__ g2idx;
    int64_t d, dk, trace_cap, k;
    int32_t K, nbc, adapt, pad;
    double T, delta, factor;
};
size_t zz_partitioned_lds_bytes(int K, int nbc);
int launch_zz_partitioned(const ZzPartParams& p, int64_t nchains, void* stream);

// General-degree local ZigZag (pdmp_general.hip): CSC tables instead of the blob, optional logistic target
struct ZzGeneralParams {
    const uint16_t* __restrict__ pos16;      // like ZzTables::pos, 16-bit positions inside S[i]
    const uint16_t* __restrict__ selfpos16;  // position of i inside G1[i]
    const double* __restrict__ qbval;        // like pos16: the Γ value of the same (member, entry) pair
    const uint4* __restrict__ member;        // per entry p of column i: {j, k_j, qptr[p], -}
    uint32_t mmax_pad;                       // LDS scratch slots (max |S[i]|, padded)
    int32_t target_kind;                     // 0 Gaussian CSC, 1 subsampled logistic (scripts/logistic.jl:107)
    const int64_t* __restrict__ A_colptr;    // design A (n x p), CSC by coordinate
    const int64_t* __restrict__ A_rowval;
    const double* __restrict__ A_nzval;
    const int64_t* __restrict__ At_colptr;   // A' (p x n), CSC by observation
    const int64_t* __restrict__ At_rowval;
    const uint32_t* __restrict__ At_row32;  // the same indices in 4 bytes (p < 2^31): what the sweeps of long rows stream
    const double* __restrict__ At_nzval;
    const double* __restrict__ y;
    const double* __restrict__ ny;
    const double* __restrict__ sn0;          // sigmoidn(idot(At, row, μ)) per observation (control variate, tabulated on the host)
    const double* __restrict__ ns0;          // nsigmoid(idot(At, row, μ))
    double gamma0;
    int64_t ksub;
    int32_t lg_ne_max;  // most regressors of any observation (max column length of A')
    double* hot;        // [nchains x d x 4] scratch for the split state of the ranged sweeps (x, θ, t, ∫x dt per coordinate), or null
    int32_t lg_range;   // > 0: long rows (dense designs) are swept in coordinate ranges of this width, all sampled rows per range (see pdmp_general.hip)
    int32_t sticky;  // sspdmp (src/ss_fact.jl) on this kernel: rec.acc is the freeze flag f[i], P.thf / P.kappa are in use
    // flow_kind 1: FactBoomerang (src/types.jl:71-79)
    int32_t flow_kind;
    const double* __restrict__ mu;    // [d] flow mean
    const double* __restrict__ diag;  // [d] Γ[i,i]
    double rho;
    // c::LocalBound (src/local.jl:2-6,10-78): qtval = the TARGET's Γ values in the (member, entry) layout of qbval; renew flags in
    // renew_chain [nchains x d] (0.0 / 1.0)
    int32_t local_bound;
    int32_t masked;  // the tables' pattern is G ⊋ G1 (pdmp_ensemble_set_neighbourhood): member[..].w flags the entries of G1 -- only they are re-bounded
    const double* __restrict__ qtval;
    double* renew_chain;  // written and re-read by the same wave: no __restrict__ (a scalar-cache load would see stale flags)
    // adaptscale (src/sfact.jl:86-99): per-chain σ [nchains x d], nullptr when off
    double* __restrict__ sig_chain;
    int32_t adaptscale;
};
int launch_zz_general_run(const ZzRunParams& p, const ZzGeneralParams& q, int64_t nchains, void* stream);

// Small-d subsampled logistic ZigZag with the chain state resident in LDS (pdmp_logistic.hip): packed read-only tables
struct LgCoord {  // everything a proposal needs that depends on the coordinate alone
    uint32_t cp0, k;    // column of the bounding Γ: G1[i] = rowval[cp0 .. cp0 + k)
    uint32_t sp0, m;    // S[i] = sidx[sp0 .. sp0 + m) = G1[i] followed by G2[i]
    uint32_t l, r0;     // observations with a non-zero entry in column i of the design A: a_row / a_val [r0 .. r0 + l)
    double lk;          // l / k_sub, the weight of a sampled observation (scripts/logistic.jl:85: the same IEEE division, done once on the host)
};
static_assert(sizeof(LgCoord) == 32, "one sector");
struct alignas(128) LgObs {  // one observation (a column of A'): scripts/logistic.jl:86-93
    double y, ny, sn0, ns0;  // successes, failures, sigmoidn(A'[:,row]·μ), nsigmoid(A'[:,row]·μ)
    double val[6];           // A'[idx[e], row]
    uint16_t idx[6];         // its regressors, ascending
    uint16_t ne, pad16;
    uint32_t pad[8];
};
static_assert(sizeof(LgObs) == 128, "one line");
struct ZzLogisticTables {
    const LgCoord* __restrict__ coord;     // [d]
    const LgObs* __restrict__ obs;         // [n]
    const uint32_t* __restrict__ a_row;    // [nnz(A)] observation of every entry of A, column by column
    const double* __restrict__ a_val;      // [nnz(A)]
    const uint16_t* __restrict__ qrow16;   // like ZzGeneralParams::qbval: the ROW (coordinate) of the same (member, entry) pair
    double* trk;                           // [nchains x d x 4] (g, gd, tg, -): tracked sums of the bounds (pdmp_ensemble_set_gradient_tracking), or null
};
int launch_zz_logistic_track_init(const ZzRec* rec, const ZzTables& tb, int64_t d, int64_t nchains, double t0, double* trk, void* stream);
size_t zz_logistic_lds_bytes(int64_t d, int64_t dk, bool with_I);
bool zz_logistic_lds_supported(const ZzRunParams& p, const ZzGeneralParams& q, const ZzLogisticTables& lt);
int launch_zz_logistic_lds(const ZzRunParams& p, const ZzGeneralParams& q, const ZzLogisticTables& lt, bool with_I, int64_t nchains,
                           void* stream);
// 64 / W chains per wavefront, each in a row of W = 16 or 32 lanes (pdmp_logrows.hip)
size_t zz_logistic_rows_lds_bytes(int64_t d, int W, bool with_I);
bool zz_logistic_rows_supported(const ZzRunParams& p, const ZzGeneralParams& q, const ZzLogisticTables& lt, int W);
int launch_zz_logistic_rows(const ZzRunParams& p, const ZzGeneralParams& q, const ZzLogisticTables& lt, bool with_I, int W, int64_t nchains,
                            void* stream);
size_t zz_general_lds_bytes(uint32_t nblk_pad, uint32_t mmax_pad, bool boom);

// Bouncy particle sampler (pdmp_bps.hip): per chain x[d], θ[d] (SoA), 8 scalars {t, a, b, t′, τref, c, -, -}
struct BpsRunParams {
    const int64_t* __restrict__ colptr;
    const int64_t* __restrict__ rowval;
    const double* __restrict__ nzval;
    const double* __restrict__ mu;
    double* x;
    double* th;
    double* scal;
    DevChain* hdr;
    double* ev_t;   // [nchains x cap]
    double* ev_x;   // [nchains x cap x d]
    double* ev_th;  // [nchains x cap x d]
    int64_t d;
    int64_t trace_cap;
    double T, factor, lambda_ref, rho;
    int32_t flags, adapt;
    int32_t flow_kind;                  // 0 BouncyParticle, 1 Boomerang (L = I): mu_flow is the centre of rotation
    int32_t ident;                      // Γ == I and μ == 0 exactly (isotropic target): gradient-free register layout
    const double* __restrict__ mu_flow;  // [d]
    // extended instantiation (ext != 0): mass factor L (lower CSC, diagonal first) and L' (upper CSC, diagonal last), nullptr =
    // identity; c::LocalBound; subsample
    int32_t ext, local_bound, subsample, pad_;
    // BouncyParticle with a target of its own (nullptr: the target is B.Γ(x − B.μ)): ∇ϕ!(y, x) = Γt(x − μt); extended instantiation only
    const int64_t* __restrict__ t_colptr;
    const int64_t* __restrict__ t_rowval;
    const double* __restrict__ t_nzval;
    const double* __restrict__ t_mu;
    const int32_t* __restrict__ Lcp;
    const int32_t* __restrict__ Lrv;
    const double* __restrict__ Lnz;
    const int32_t* __restrict__ Ucp;
    const int32_t* __restrict__ Urv;
    const double* __restrict__ Unz;
};
int launch_bps_write_probe(double* ev_x, double* ev_th, int64_t d, int64_t cap, int64_t nrec, int64_t nchains, void* stream);
int launch_sector_probe(double* rec, int64_t d, int64_t nchains, int rounds, int write, double* sink, void* stream);
int launch_bps_init(const BpsRunParams& p, int64_t nchains, const uint64_t* seeds, double t0, double c0, void* stream);
int launch_bps_run(const BpsRunParams& p, int64_t nchains, bool diag, void* stream);

// launch wrappers implemented in pdmp_kernels.hip (hipStream_t passed as void*)
int launch_zz_init(const ZzInitParams& p, void* stream);
int launch_zz_local_run(const ZzRunParams& p, int64_t nchains, void* stream);
int launch_zz_local_spec(const ZzRunParams& p, int64_t nchains, void* stream, const char** kname = nullptr);
int launch_zz_sticky_run(const ZzRunParams& p, int64_t nchains, void* stream);
int launch_zz_sticky_spec(const ZzRunParams& p, int64_t nchains, void* stream);
bool zz_spec_supported(uint32_t nblk, uint32_t mmax, uint32_t kmax);
size_t zz_spec_lds_bytes(uint32_t nblk_pad, uint32_t blob_w_pad);
size_t zz_spec_wide_lds_bytes(uint32_t nblk_pad, uint32_t blob_w_pad);  // 17 <= |S[i]| <= 32
int launch_zz_unpack(const ZzRec* rec, const double* c_src, int64_t c_stride, int64_t d, int64_t chain_first,
                     int64_t n, double* t, double* x, double* th, int64_t* acc, double* c, void* stream);
int launch_zz_batch_means(const ZzRec* rec, int64_t rec_stride, double* jprev, int64_t d, int64_t nchains, double T_prev, double T,
                          double* sum_y, double* sum_y2, void* stream);
int launch_zz_local_track(const ZzRunParams& p, int64_t nchains, void* stream);
bool zz_exactp_supported(const ZzRunParams& p);  // pdmp_exactp.hip: the moving evaluation, one proposal per lane
int launch_zz_local_exactp(const ZzRunParams& p, int64_t nchains, void* stream);
bool zz_trackp_supported(const ZzRunParams& p);
int launch_zz_local_trackp(const ZzRunParams& p, int64_t nchains, void* stream);
int launch_zz_keys_to_pairs(const double* keys, void* kp, int64_t n, double t0, void* stream);
int launch_zz_trackp_c_out(const void* rec, double* c_chain, int64_t n, void* stream);
int launch_zz_trackp_consts(void* rec, const CoordConst* cc, const uint16_t* nb16, const double* gmu, int64_t d, int64_t nchains, void* stream);
// pdmp_trackl.hip: the same kernel on the line layout (full-width launches of the plain lattice, d <= 16384)
bool zz_trackl_supported(const ZzRunParams& p);
int launch_zz_local_trackl(const ZzRunParams& p, int64_t nchains, void* stream);
int launch_zz_trackl_pack(const void* rec, const void* kp, void* lines, void* cold, int64_t d, int64_t dk, int64_t nchains, void* stream);
int launch_zz_trackl_unpack(const void* lines, const void* cold, void* rec, void* kp, int64_t d, int64_t dk, int64_t nchains, void* stream);
constexpr int TRACKP_KMAX = 8;  // |G1[i]| the generic instantiation takes (one member per lane of an 8-lane group)
bool zz_spec8_geometry(const ZzRunParams& p);  // the 8-event kernels' requirements on the neighbourhood blob and on d
// kp != nullptr: the (key, time of the last own proposal) pairs of pdmp_trackp.hip hold tprop instead of the records (chain stride dk pairs)
int launch_zz_track_unpack(const TrRec* rec, const ZzTables& tb, const double* c_src, int64_t c_stride, int64_t d, int64_t chain_first,
                           int64_t n, double t0, double* t, double* x, double* th, int64_t* acc, double* c, const double* kp, int64_t dk,
                           void* stream);
int launch_zz_ess(const ZzRec* rec, int64_t rec_stride, double* jprev, double* jstart, int64_t d, int64_t nchains, int mode, double T_prev,
                  double T, double* acc, void* stream);
size_t zz_local_lds_bytes(uint32_t nblk_pad, uint32_t blob_w_pad);
// pdmp_consume.hip
size_t consume_cursor_bytes(bool with_z);
size_t consume_meta_bytes();
int launch_consume_init(const ZzRec* rec, int64_t rec_stride, int64_t d, int64_t nchains, double t0, void* cur, bool with_z, void* meta, double* grid,
                        int64_t K, void* stream);
int launch_consume_snapshot(DevChain* hdr, int64_t nchains, uint64_t* snap, void* stream);
int launch_consume_events(const pdmp_event* ev, int64_t cap, const DevChain* hdr, const uint64_t* snap, int64_t d, int64_t nchains, void* cur,
                          bool with_z, void* meta, double* grid, int64_t K, double t0, double dt, void* stream, double* cummean_pairs = nullptr);
// pdmp_place.hip: an array of several GB built from chunks of the device's three memory classes in turn (one contiguous address range)
struct Placement {
    void* va = nullptr;
    size_t va_bytes = 0, chunk = 0, walked = 0;  // walked: chunks created while looking for the classes
    double seconds = 0.0;
    std::vector<void*> handles;
    std::string classes;  // one digit per chunk in address order, e.g. "012012012"
};
struct PlaceConfig {
    int enabled = 0;  // experimental (pdmp_place.hip's header says why): off unless pdmp_debug_set_placement turns it on
    size_t chunk_mb = 1024, min_mb = 3072, max_walk = 192;
    std::string rec, kp, ev;  // class patterns of the records / the pairs / the trace ("" = "012" for arrays of at least min_mb, nothing below)
};
bool placed_alloc(size_t bytes, Placement& out, const PlaceConfig* cfg, const std::string* forced_pattern);  // false: nothing is held
void placed_free(Placement& p);
int launch_trace_subtrace(const pdmp_event* ev, int64_t n, const int32_t* loc, pdmp_event* out, int64_t out_cap, unsigned long long* n_out, void* stream);
int launch_consume_flush(int64_t d, int64_t nchains, const void* cur, const void* meta, double* grid, int64_t K, double t0, double dt, void* stream);
int launch_consume_mean(int64_t d, int64_t chain_first, int64_t n, const void* cur, const void* meta, double* mean_out, double* T_out, void* stream);
int launch_consume_inclusion(int64_t d, int64_t nchains, bool with_z, double t0, int64_t chain_first, int64_t n, void* cur, const void* meta, double* out,
                             double* T_out, void* stream);
int launch_zz_path_integrals(const ZzRec* rec, int64_t rec_stride, int64_t d, int64_t nchains, const int64_t* probes, int64_t nprobe,
                             double T, double* out, void* stream);
int launch_math_probe(uint64_t seed, int64_t n, double* out, void* stream);

}  // namespace pdmp
