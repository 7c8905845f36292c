/*
 * pdmp_oracle.c -- CPU ORACLE (test infrastructure only; see pdmp_oracle.h for the parity status).
 *
 * Every function cites the reference lines (relative to /root/reference) it restates.  The arithmetic
 * is written operation by operation in the order Julia evaluates it; the file is compiled with
 * -ffp-contract=off so that no a*b+c is fused.
 */
#include "pdmp_oracle.h"
#include "../include/pdmp_detmath.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ small helpers */

/* pos(x) = max(zero(x), x), src/common.jl:8 */
static inline double pos(double x) {
    return (x > 0.0) ? x : ((x != x) ? x : 0.0);
}

void orc_trace_init(orc_trace* tr) {
    tr->ev = NULL;
    tr->n = 0;
    tr->cap = 0;
}
void orc_trace_free(orc_trace* tr) {
    free(tr->ev);
    tr->ev = NULL;
    tr->n = tr->cap = 0;
}
static void trace_push(orc_trace* tr, double t, int64_t i, double x, double th) {
    if (!tr) return;
    if (tr->n == tr->cap) {
        tr->cap = tr->cap ? 2 * tr->cap : 1024;
        tr->ev = (orc_event*)realloc(tr->ev, (size_t)tr->cap * sizeof(orc_event));
    }
    tr->ev[tr->n].t = t;
    tr->ev[tr->n].i = i;
    tr->ev[tr->n].x = x;
    tr->ev[tr->n].theta = th;
    tr->n++;
}

/* ------------------------------------------------------------------ src/poissontime.jl */

/* poisson_time(a, b, u), src/poissontime.jl:8-30 */
double orc_poisson_time(double a, double b, double u) {
    if (b > 0) {
        if (a < 0) {
            return sqrt(-pdmp_log(u) * 2.0 / b) - a / b; /* :11 */
        } else {
            return sqrt((a / b) * (a / b) - pdmp_log(u) * 2.0 / b) - a / b; /* :13 */
        }
    } else if (b == 0) {
        if (a > 0) {
            return -pdmp_log(u) / a; /* :17 */
        } else {
            return INFINITY; /* :19 */
        }
    } else {
        if (a <= 0) {
            return INFINITY; /* :23 */
        } else if (-pdmp_log(u) <= -(a * a) / b + (a * a) / (2 * b)) { /* :24 */
            return -sqrt((a / b) * (a / b) - pdmp_log(u) * 2.0 / b) - a / b; /* :25 */
        } else {
            return INFINITY; /* :27 */
        }
    }
}

/* poisson_time((a,b,c), u), src/poissontime.jl:39-65 : rate c + (a + b t)^+ */
double orc_poisson_time3(double a, double b, double c, double u) {
    double lu = pdmp_log(u);
    if (b > 0) {
        if (a < 0) {
            if (-c * a / b + lu < 0.0) {
                return sqrt(-2 * b * lu + c * c + 2 * a * c) / b - (a + c) / b; /* :43 */
            } else {
                return -lu / c; /* :45 */
            }
        } else {
            return sqrt(-lu * 2.0 * b + (a + c) * (a + c)) / b - (a + c) / b; /* :48 */
        }
    } else if (b == 0) {
        if (a > 0) {
            return -lu / (a + c); /* :52 */
        } else {
            return -lu / c; /* :54 */
        }
    } else {
        if (a <= 0.0) {
            return -lu / c; /* :58 */
        } else if (-c * a / b - (a * a) / (2 * b) + lu > 0.0) {
            return +sqrt((a + c) * (a + c) - 2.0 * lu * b) / b - (a + c) / b; /* :60 */
        } else {
            return (-lu + (a * a) / (2 * b)) / c; /* :62 */
        }
    }
}

/* idot(A::SparseMatrixCSC, j, x), src/common.jl:16-24 : sequential sum in CSC (ascending row) order */
double orc_idot(const orc_csc* A, int64_t j, const double* x) {
    double s = 0.0;
    for (int64_t p = A->colptr[j]; p < A->colptr[j + 1]; ++p) {
        s += A->nzval[p] * x[A->rowval[p]];
    }
    return s;
}

/* ------------------------------------------------------------------ src/priorityqueue.jl */

struct orc_pq {
    int64_t len, cap;
    int64_t* xs_key; /* 1-based heap positions, xs[k] = key => val */
    double* xs_val;
    int64_t* index;  /* key -> heap position */
    int lex;         /* 0: the reference's order (isless on the values; tied values pop in heap-shape order, :50);
                        1: ties broken by the lower key -- the device queues' rule (see spdmp_zigzag_tracked) */
};

/* isless on Float64 (Base): NaN sorts last, -0.0 < 0.0 */
static inline int f_isless(double a, double b) {
    if (a != a) return 0;
    if (b != b) return 1;
    if (a < b) return 1;
    if (a == b) return signbit(a) && !signbit(b);
    return 0;
}

/* "entry (va, ka) comes before entry (vb, kb)" */
static inline int pq_lt(const orc_pq* q, double va, int64_t ka, double vb, int64_t kb) {
    if (f_isless(va, vb)) return 1;
    return q->lex && va == vb && ka < kb;
}

orc_pq* orc_pq_new(int64_t capacity) {
    orc_pq* q = (orc_pq*)calloc(1, sizeof(orc_pq));
    q->cap = capacity;
    q->xs_key = (int64_t*)malloc((size_t)(capacity + 1) * sizeof(int64_t));
    q->xs_val = (double*)malloc((size_t)(capacity + 1) * sizeof(double));
    q->index = (int64_t*)malloc((size_t)(capacity + 1) * sizeof(int64_t));
    return q;
}
void orc_pq_free(orc_pq* q) {
    if (!q) return;
    free(q->xs_key);
    free(q->xs_val);
    free(q->index);
    free(q);
}
int64_t orc_pq_len(const orc_pq* q) {
    return q->len;
}

/* percolate_down!, src/priorityqueue.jl:46-61 (on equal children the RIGHT child is taken, :50) */
static void pq_down(orc_pq* q, int64_t i) {
    int64_t xk = q->xs_key[i];
    double xv = q->xs_val[i];
    int64_t l;
    while ((l = 2 * i) <= q->len) {
        int64_t r = 2 * i + 1;
        int64_t j = (r > q->len || pq_lt(q, q->xs_val[l], q->xs_key[l], q->xs_val[r], q->xs_key[r])) ? l : r;
        if (pq_lt(q, q->xs_val[j], q->xs_key[j], xv, xk)) {
            q->index[q->xs_key[j]] = i;
            q->xs_key[i] = q->xs_key[j];
            q->xs_val[i] = q->xs_val[j];
            i = j;
        } else {
            break;
        }
    }
    q->index[xk] = i;
    q->xs_key[i] = xk;
    q->xs_val[i] = xv;
}
/* percolate_up!, src/priorityqueue.jl:63-77 */
static void pq_up(orc_pq* q, int64_t i) {
    int64_t xk = q->xs_key[i];
    double xv = q->xs_val[i];
    while (i > 1) {
        int64_t j = i / 2;
        if (pq_lt(q, xv, xk, q->xs_val[j], q->xs_key[j])) {
            q->index[q->xs_key[j]] = i;
            q->xs_key[i] = q->xs_key[j];
            q->xs_val[i] = q->xs_val[j];
            i = j;
        } else {
            break;
        }
    }
    q->index[xk] = i;
    q->xs_key[i] = xk;
    q->xs_val[i] = xv;
}
/* enqueue!, src/priorityqueue.jl:107-117 (keys must arrive in order) */
void orc_pq_enqueue(orc_pq* q, int64_t key, double val) {
    if (q->len != key || q->len >= q->cap) abort(); /* "Elements must be enqueue! in order" */
    q->len++;
    q->xs_key[q->len] = key;
    q->xs_val[q->len] = val;
    q->index[key] = q->len;
    pq_up(q, q->len);
}
/* setindex!, src/priorityqueue.jl:95-105 */
void orc_pq_set(orc_pq* q, int64_t key, double val) {
    int64_t i = q->index[key];
    double old = q->xs_val[i];
    q->xs_val[i] = val;
    if (f_isless(old, val)) {
        pq_down(q, i);
    } else {
        pq_up(q, i);
    }
}
double orc_pq_get(const orc_pq* q, int64_t key) {
    return q->xs_val[q->index[key]];
}
/* peek, src/priorityqueue.jl:44 */
void orc_pq_peek(const orc_pq* q, int64_t* key, double* val) {
    *key = q->xs_key[1];
    *val = q->xs_val[1];
}
int orc_pq_check(const orc_pq* q) {
    for (int64_t i = 1; i <= q->len; ++i) {
        if (q->index[q->xs_key[i]] != i) return 0;
        if (i > 1 && f_isless(q->xs_val[i], q->xs_val[i / 2])) return 0;
    }
    return 1;
}

/* ------------------------------------------------------------------ neighbourhood graphs */

typedef struct {
    int64_t* ptr; /* d+1 */
    int64_t* idx;
} nbr_graph;

static void graph_free(nbr_graph* g) {
    free(g->ptr);
    free(g->idx);
    g->ptr = g->idx = NULL;
}

/* G1[i] = rowvals(F.Γ)[nzrange(F.Γ, i)], src/sfact.jl:170 */
static nbr_graph graph_g1(const orc_csc* A) {
    nbr_graph g;
    int64_t d = A->n;
    g.ptr = (int64_t*)malloc((size_t)(d + 1) * sizeof(int64_t));
    g.idx = (int64_t*)malloc((size_t)A->colptr[d] * sizeof(int64_t) + 8);
    memcpy(g.ptr, A->colptr, (size_t)(d + 1) * sizeof(int64_t));
    memcpy(g.idx, A->rowval, (size_t)A->colptr[d] * sizeof(int64_t));
    return g;
}

static int cmp_i64(const void* a, const void* b) {
    int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
    return (x > y) - (x < y);
}

/* G2[i] = setdiff(union(G1[j] for j in G1[i]), G[i]), src/sfact.jl:178 (stored ascending; the order of
 * a neighbourhood only fixes the order of independent per-coordinate moves) */
static nbr_graph graph_g2(const nbr_graph* g1, int64_t d) {
    nbr_graph g;
    g.ptr = (int64_t*)malloc((size_t)(d + 1) * sizeof(int64_t));
    int64_t cap = 16 * d + 16, n = 0;
    g.idx = (int64_t*)malloc((size_t)cap * sizeof(int64_t));
    int64_t* tmp = NULL;
    int64_t tmpcap = 0;
    for (int64_t i = 0; i < d; ++i) {
        g.ptr[i] = n;
        int64_t m = 0;
        for (int64_t p = g1->ptr[i]; p < g1->ptr[i + 1]; ++p) {
            int64_t j = g1->idx[p];
            m += g1->ptr[j + 1] - g1->ptr[j];
        }
        if (m > tmpcap) {
            tmpcap = 2 * m;
            tmp = (int64_t*)realloc(tmp, (size_t)tmpcap * sizeof(int64_t));
        }
        m = 0;
        for (int64_t p = g1->ptr[i]; p < g1->ptr[i + 1]; ++p) {
            int64_t j = g1->idx[p];
            for (int64_t q = g1->ptr[j]; q < g1->ptr[j + 1]; ++q) tmp[m++] = g1->idx[q];
        }
        qsort(tmp, (size_t)m, sizeof(int64_t), cmp_i64);
        int64_t last = -1;
        for (int64_t k = 0; k < m; ++k) {
            int64_t v = tmp[k];
            if (v == last) continue;
            last = v;
            /* setdiff with G[i] = G1[i] (Matched) */
            int in_g = 0;
            for (int64_t p = g1->ptr[i]; p < g1->ptr[i + 1]; ++p) {
                if (g1->idx[p] == v) {
                    in_g = 1;
                    break;
                }
            }
            if (in_g) continue;
            if (n == cap) {
                cap *= 2;
                g.idx = (int64_t*)realloc(g.idx, (size_t)cap * sizeof(int64_t));
            }
            g.idx[n++] = v;
        }
    }
    g.ptr[d] = n;
    free(tmp);
    return g;
}

/* ------------------------------------------------------------------ local ZigZag: src/sfact.jl */

typedef struct {
    int64_t d;
    const orc_zz_params* p;
    nbr_graph g1, g2;
    double* gmu_bound;  /* idot(Z.Γ, i, Z.μ), src/fact_samplers.jl:51 */
    double* gmu_target; /* idot(Γt, i, μt) or NULL */
} zz_ctx;

/* smove_forward!(i::Int, ...), src/sfact.jl:13-16 */
static inline void move1(int64_t j, double* t, double* x, const double* th, double tp) {
    x[j] = x[j] + th[j] * (tp - t[j]);
    t[j] = tp;
}
/* smove_forward!(G, i, ...), src/sfact.jl:6-12 */
static inline void move_nbrs(const nbr_graph* g, int64_t i, double* t, double* x, const double* th, double tp) {
    for (int64_t p = g->ptr[i]; p < g->ptr[i + 1]; ++p) move1(g->idx[p], t, x, th, tp);
}
/* smove_forward!(t, x, θ, t′, Z), src/sfact.jl:23-28 */
static inline void move_all(int64_t d, double* t, double* x, const double* th, double tp) {
    for (int64_t j = 0; j < d; ++j) move1(j, t, x, th, tp);
}

/* smove_forward!(G, i, t, x, θ, t′, B::FactBoomerang), src/sfact.jl:29-36: rotation about μ by the elapsed time */
static inline void boom_move1(int64_t j, double* t, double* x, double* th, double tp, const double* mu) {
    const double tau = tp - t[j];
    double s, c;
    pdmp_sincos(tau, &s, &c);
    const double xo = x[j], tho = th[j];
    x[j] = (xo - mu[j]) * c + tho * s + mu[j];
    th[j] = -(xo - mu[j]) * s + tho * c;
    t[j] = tp;
}
static inline void flow_move_nbrs(int kind, const double* mu, const nbr_graph* g, int64_t i, double* t, double* x, double* th,
                                  double tp) {
    if (!kind) {
        move_nbrs(g, i, t, x, th, tp);
        return;
    }
    for (int64_t p = g->ptr[i]; p < g->ptr[i + 1]; ++p) boom_move1(g->idx[p], t, x, th, tp, mu);
}
static inline void flow_move_all(int kind, const double* mu, int64_t d, double* t, double* x, double* th, double tp) {
    if (!kind) {
        move_all(d, t, x, th, tp);
        return;
    }
    for (int64_t j = 0; j < d; ++j) boom_move1(j, t, x, th, tp, mu);
}
/* ab(G, i, x, θ, c, Z::FactBoomerang), src/fact_samplers.jl:58-65 */
static inline void boom_ab(const orc_csc* G, const double* mu, const double* diag, int64_t i, const double* x,
                           const double* th, const double* c, double* a, double* b) {
    double zz = 0.0;
    for (int64_t p = G->colptr[i]; p < G->colptr[i + 1]; ++p) {
        const int64_t j = G->rowval[p];
        zz += (x[j] - mu[j]) * (x[j] - mu[j]) + th[j] * th[j];
    }
    const double z = sqrt(zz);
    const double z2 = x[i] * x[i] + th[i] * th[i];
    *a = c[i] * sqrt(z2) * z + z2 * diag[i];
    *b = 0.0;
}

/* ab(G, i, x, θ, c, Z::ZigZag), src/fact_samplers.jl:50-54 with loosen(c,x) = c + x (:41) */
static inline void zz_ab(const orc_csc* G, const double* gmu, int64_t i, const double* x, const double* th,
                         const double* c, double* a, double* b) {
    *a = c[i] + (orc_idot(G, i, x) - gmu[i]) * th[i];
    *b = c[i] / 100 + th[i] * orc_idot(G, i, th);
}

/* ∇ϕ(x, i, Γ) = idot(Γ, i, x), scripts/gaussianrandomfield.jl:25, test/maintest.jl:9 */
static inline double zz_grad(const zz_ctx* cx, int64_t i, const double* x) {
    double g = orc_idot(cx->p->target_gamma, i, x);
    if (cx->gmu_target) g = g - cx->gmu_target[i];
    return g;
}

/* sigmoid(x) = inv(one(x) + exp(-x)), scripts/logistic.jl:33; sigmoidn, nsigmoid :56-57 */
static inline double lg_sigmoid(double x) {
    return 1.0 / (1.0 + pdmp_exp(-x));
}
static inline double lg_sigmoidn(double x) {
    return lg_sigmoid(-x);
}
static inline double lg_nsigmoid(double x) {
    return -lg_sigmoid(x);
}

/* ∇ϕmoving = γ0*x[i] - fdot_moving(...), scripts/logistic.jl:78-95,107 (idot_moving!: src/common.jl:33-42) */
static double logistic_grad_moving(const orc_zz_params* p, int64_t j, double* t, double* x, const double* th, double tp,
                                   uint64_t seed, uint64_t* ng) {
    const orc_csc* A = p->lg_A;
    const orc_csc* At = p->lg_At;
    const double prior = p->lg_gamma0 * x[j];
    double s = 0.0;
    const int64_t r0 = A->colptr[j];
    const int64_t l = A->colptr[j + 1] - r0;
    const int64_t k = p->lg_k;
    for (int64_t q = 0; q < k; ++q) {
        const int64_t ii = r0 + (int64_t)pdmp_randint(seed, PDMP_STREAM_GLOBAL, (*ng)++, (uint32_t)l); /* rand(sampler) */
        const int64_t row = A->rowval[ii];
        const double v = A->nzval[ii];
        double u = 0.0; /* idot_moving!(At, rows[i], t, x, θ, t′, F) */
        for (int64_t e = At->colptr[row]; e < At->colptr[row + 1]; ++e) {
            const int64_t cc = At->rowval[e];
            move1(cc, t, x, th, tp);
            u += At->nzval[e] * x[cc];
        }
        const double w = (double)l / (double)k * v;
        s += w * p->lg_y[row] * lg_sigmoidn(u);
        s += w * p->lg_ny[row] * lg_nsigmoid(u);
        const double u0 = orc_idot(At, row, p->lg_mu);
        s -= w * p->lg_y[row] * lg_sigmoidn(u0);
        s -= w * p->lg_ny[row] * lg_nsigmoid(u0);
    }
    return prior - s;
}

/* poisson_time(a, b, u) with L = log(u) already taken (src/poissontime.jl:8-30): the same operations on the same operands as
 * orc_poisson_time -- (-L)*2/b and -(L*2/b) are the same double -- in the form the tracked kernels evaluate it. */
static double poisson_time_L(double a, double b, double L) {
    if (b == 0) return (a > 0) ? -L / a : INFINITY;
    const double r = a / b;
    const double q = L * 2.0 / b;
    if (b > 0) return sqrt((a < 0) ? -q : r * r - q) - r;
    if (a <= 0) return INFINITY;
    if (-L <= -(a * a) / b + (a * a) / (2 * b)) return -sqrt(r * r - q) - r;
    return INFINITY;
}

/* entry (row, col) of a CSC matrix (0.0 if it is not stored) */
static double csc_entry(const orc_csc* A, int64_t row, int64_t col) {
    for (int64_t p = A->colptr[col]; p < A->colptr[col + 1]; ++p)
        if (A->rowval[p] == row) return A->nzval[p];
    return 0.0;
}

/*
 * TRACKED-GRADIENT evaluation of spdmp_inner! (p->tracked): the bitwise statement of what the device's tracked kernels compute
 * (zz_local_trackp_kernel, pdmp_trackp.hip; zz_local_track_kernel, pdmp_kernels.hip).  The PROCESS is the reference's -- same queue,
 * same draws in the same order, same thinning test (src/sfact.jl:116-140), same bounds (src/fact_samplers.jl:50-54) -- but the
 * gradient is not gathered from a moved neighbourhood (src/sfact.jl:82,116): every coordinate carries
 *     g_i = Γt[:,i]·x  and  gd_i = Γt[:,i]·θ  at time tg_i      (and gb_i, gdb_i with the bounding Γ when that differs),
 * which the linear flow advances exactly in real arithmetic, g_i(t′) = g_i + gd_i (t′ − tg_i).  In floating point the advanced sum
 * and the gathered sum differ in their last bits, so this function is NOT bit-identical to the moving evaluation above (they agree
 * to ~1e-13; tests/test_oracle_tracked.py holds them to 1e-9 with identical indices), while the DEVICE kernels are held to this
 * function bit for bit.  Every operation below is written in the kernels' order:
 *   proposal of i at t′:   g_now = g + gd (t′ − tg);  l = (g_now − (Γt μt)_i) θ_i ⁺;  lb = (a + b (t′ − t_old))⁺      (:119)
 *   reject:                a = c_i + (gb_now − (Γ μ)_i) θ_i,  b = c_i/100 + θ_i gdb,  t_old = t′,  key = t′ + poisson_time   (:137-140)
 *   accept, j ∈ G1[i] ascending (:131-135): g_j = g_j + gd_j (t′ − tg_j),  gd_j = gd_j + Γt[j,i]·(−2θ_i),  tg_j = t′, same bound
 *                          formulas with θ_j (−θ_i for j = i);  x_i is brought to t′ on its own accepts only (event(), :50-52).
 * The pair-layout kernel does not store (a, b, t_old): it re-derives them from t_old = max(tprop, tg) and the same operands, which
 * yields the same doubles (a bound is computed either at the coordinate's own proposal or when its sums are re-based).
 * TIES.  Two finite keys are exactly equal with probability ~2.5e-10 per event on config C3 (event gaps ~1.4e-5, one ulp at t ~ 16 is
 * 3.6e-15): 4096 chains to T = 20 are 5.9e9 proposals, so a tie or two DOES occur at the north star's scale.  The reference pops tied
 * keys in the order its heap happens to hold them (src/priorityqueue.jl:46-61, right child on equal children); the device queues pop
 * the LOWEST COORDINATE first.  Both orders are valid for simultaneous events of independent clocks, but they hand the stream's draws
 * to different events, so the chains part.  This function is the kernels' checker and uses THEIR rule (the queue below compares
 * (key, coordinate) pairs); the moving evaluation above keeps the reference's heap.  (Observed: chain 1018 of the C3 ensemble,
 * coordinates 11340 and 15903 both due at t = 18.887967430385714 in the tracked arithmetic.)
 * Final state as zz_track_unpack_kernel rebuilds it (src/sfact.jl:211): t[j] = the later of the last proposal inside G1[j] and the
 * last accepted event inside S[j] = G1[j] ∪ G2[j]; x[j] moved linearly from its own clock to t[j].
 * Requirements (those of the kernels): ZigZag flow, Gaussian target, no refresh clock, Matched(); the target's Γ is symmetric
 * and has the pattern of the bounding Γ.
 */
static int spdmp_zigzag_tracked(int64_t d, const orc_zz_params* p, double t0, double T, double* x, double* th, double* c, double* t,
                                int64_t* acc, orc_trace* tr, orc_zz_result* res) {
    const orc_csc* Gb = p->bound_gamma;
    const orc_csc* Gt = p->target_gamma;
    if (p->flow_kind || p->lambda_ref > 0 || p->move_all || p->target_kind || p->local_bound || p->adaptscale) return ORC_BAD_INPUT;
    if (Gt->colptr[d] != Gb->colptr[d] || memcmp(Gt->colptr, Gb->colptr, (size_t)(d + 1) * sizeof(int64_t)) ||
        memcmp(Gt->rowval, Gb->rowval, (size_t)Gb->colptr[d] * sizeof(int64_t)))
        return ORC_BAD_INPUT;
    const int two_sums = memcmp(Gt->nzval, Gb->nzval, (size_t)Gb->colptr[d] * sizeof(double)) != 0;
    nbr_graph g1 = graph_g1(Gb);
    nbr_graph g2 = graph_g2(&g1, d);
    double* gmu_b = (double*)malloc((size_t)d * sizeof(double));
    double* gmu_t = p->target_mu ? (double*)malloc((size_t)d * sizeof(double)) : NULL;
    for (int64_t i = 0; i < d; ++i) gmu_b[i] = orc_idot(Gb, i, p->bound_mu);
    if (gmu_t)
        for (int64_t i = 0; i < d; ++i) gmu_t[i] = orc_idot(Gt, i, p->target_mu);
    double* tx = (double*)malloc((size_t)d * sizeof(double));    /* clock of x_i */
    double* g = (double*)malloc((size_t)d * sizeof(double));     /* Γt[:,i]·x at tg */
    double* gd = (double*)malloc((size_t)d * sizeof(double));    /* Γt[:,i]·θ */
    double* gb = (double*)malloc((size_t)d * sizeof(double));    /* the same with the bounding Γ */
    double* gdb = (double*)malloc((size_t)d * sizeof(double));
    double* tg = (double*)malloc((size_t)d * sizeof(double));
    double* ba = (double*)malloc((size_t)d * sizeof(double));
    double* bb = (double*)malloc((size_t)d * sizeof(double));
    double* t_old = (double*)malloc((size_t)d * sizeof(double));
    double* tprop = (double*)malloc((size_t)d * sizeof(double)); /* last own proposal */
    double* tacc = (double*)malloc((size_t)d * sizeof(double));  /* last own accept */
    const uint64_t seed = p->seed;
    uint64_t nm = 0;
    orc_pq* Q = orc_pq_new(d + 1);
    Q->lex = 1; /* exactly tied keys pop lowest coordinate first, as on the device */
    for (int64_t i = 0; i < d; ++i) {
        g[i] = orc_idot(Gt, i, x);
        gd[i] = orc_idot(Gt, i, th);
        gb[i] = orc_idot(Gb, i, x);
        gdb[i] = orc_idot(Gb, i, th);
        tx[i] = tg[i] = t_old[i] = tprop[i] = tacc[i] = t0;
        acc[i] = 0;
        ba[i] = c[i] + (gb[i] - gmu_b[i]) * th[i]; /* src/fact_samplers.jl:51 */
        bb[i] = c[i] / 100 + th[i] * gdb[i];       /* :52 */
    }
    for (int64_t i = 0; i < d; ++i) /* src/sfact.jl:186 (t0 is not added) */
        orc_pq_enqueue(Q, i, orc_poisson_time(ba[i], bb[i], pdmp_u01(seed, PDMP_STREAM_MAIN, nm++)));

    int64_t num = 0, nacc = 0;
    int status = ORC_OK;
    double tp = t0;
    int done = 0;
    while (!done && tp < T) { /* src/sfact.jl:199 */
        for (;;) {
            int64_t i;
            double tq;
            orc_pq_peek(Q, &i, &tq); /* :77 */
            if (p->stop_before_T && !(tq < T)) {
                done = 1;
                break;
            }
            if (tq == INFINITY) {
                status = ORC_STALLED;
                done = 1;
                break;
            }
            tp = tq;
            const double g_now = g[i] + gd[i] * (tp - tg[i]);
            const double gb_now = two_sums ? (gb[i] + gdb[i] * (tp - tg[i])) : g_now;
            double gr = g_now;
            if (gmu_t) gr = gr - gmu_t[i];
            const double l = pos(gr * th[i]);                        /* :119 */
            const double lb = pos(ba[i] + bb[i] * (tp - t_old[i]));   /* :119, src/sfact.jl:70 */
            num += 1;
            tprop[i] = tp; /* the reference moved G[i] to t′ (:82) */
            if (pdmp_u01(seed, PDMP_STREAM_MAIN, nm++) * lb < l) { /* :121 */
                acc[i] += 1;
                nacc += 1;
                if (l >= lb) { /* :123 */
                    if (!p->adapt) {
                        status = ORC_BOUND_VIOLATED;
                        done = 1;
                        break;
                    }
                    c[i] *= p->factor; /* :127 */
                }
                const double th_i = th[i];
                const double delta = -th_i - th_i; /* θ_i -> −θ_i (:130); the kernels form it as (−θ) − θ = −2θ exactly */
                { /* event(i, t, x, θ, F): x_i at t′ */
                    const double dtx = tp - tx[i];
                    x[i] = x[i] + th_i * dtx;
                    tx[i] = tp;
                }
                th[i] = -th_i;
                tacc[i] = tp;
                for (int64_t q = g1.ptr[i]; q < g1.ptr[i + 1]; ++q) { /* :131-135 */
                    const int64_t j = g1.idx[q];
                    const double ct = Gt->nzval[q]; /* Γt[j, i] (= Γt[i, j]) */
                    double gj, gdj, gbj, gdbj;
                    if (j == i) {
                        gj = g_now;
                        gdj = gd[i];
                        gbj = gb_now;
                        gdbj = two_sums ? gdb[i] : gd[i];
                    } else {
                        gj = g[j] + gd[j] * (tp - tg[j]);
                        gdj = gd[j];
                        if (two_sums) {
                            gbj = gb[j] + gdb[j] * (tp - tg[j]);
                            gdbj = gdb[j];
                        } else {
                            gbj = gj;
                            gdbj = gdj;
                        }
                    }
                    gdj += ct * delta;
                    if (two_sums) gdbj += csc_entry(Gb, i, j) * delta; /* Γ[i, j]: the entry of column j at row i */
                    else gdbj = gdj;
                    g[j] = gj;
                    gd[j] = gdj;
                    gb[j] = gbj;
                    gdb[j] = gdbj;
                    tg[j] = tp;
                    ba[j] = c[j] + (gbj - gmu_b[j]) * th[j];
                    bb[j] = c[j] / 100 + th[j] * gdbj;
                    t_old[j] = tp;
                    const double L = pdmp_log(pdmp_u01(seed, PDMP_STREAM_MAIN, nm++));
                    orc_pq_set(Q, j, tp + poisson_time_L(ba[j], bb[j], L));
                }
                trace_push(tr, tp, i, x[i], th[i]); /* :143 */
                break;
            } else { /* :136-140 */
                ba[i] = c[i] + (gb_now - gmu_b[i]) * th[i];
                bb[i] = c[i] / 100 + th[i] * (two_sums ? gdb[i] : gd[i]);
                t_old[i] = tp;
                const double L = pdmp_log(pdmp_u01(seed, PDMP_STREAM_MAIN, nm++));
                orc_pq_set(Q, i, tp + poisson_time_L(ba[i], bb[i], L));
                continue;
            }
        }
        if (p->max_events > 0 && nacc >= p->max_events && !done) {
            status = ORC_TRACE_LIMIT;
            done = 1;
        }
    }
    /* the reference's lazy clocks and the positions at those clocks (zz_track_unpack_kernel) */
    for (int64_t j = 0; j < d; ++j) {
        double tr_ = t0;
        for (int64_t q = g1.ptr[j]; q < g1.ptr[j + 1]; ++q) {
            const int64_t m = g1.idx[q];
            if (tprop[m] > tr_) tr_ = tprop[m];
            if (tacc[m] > tr_) tr_ = tacc[m];
        }
        for (int64_t q = g2.ptr[j]; q < g2.ptr[j + 1]; ++q)
            if (tacc[g2.idx[q]] > tr_) tr_ = tacc[g2.idx[q]];
        t[j] = tr_;
    }
    for (int64_t j = 0; j < d; ++j) x[j] = x[j] + th[j] * (t[j] - tx[j]);
    if (res) {
        res->num = num;
        res->nacc = nacc;
        res->nrefresh = 0;
        res->ndraw_main = nm;
        res->ndraw_global = 0;
        res->t_last = tp;
        res->status = status;
    }
    orc_pq_free(Q);
    free(tx); free(g); free(gd); free(gb); free(gdb); free(tg); free(ba); free(bb); free(t_old); free(tprop); free(tacc);
    free(gmu_b);
    free(gmu_t);
    graph_free(&g1);
    graph_free(&g2);
    return status;
}

/*
 * Tracked BOUNDS under the subsampled logistic target (p->tracked with target_kind = 1): the bitwise statement of zz_logistic_lds_kernel's
 * tracked instantiation (pdmp_logistic.hip).  The gradient stays the reference's moving evaluation -- ∇ϕmoving samples k observations and moves
 * the coordinates their rows read (scripts/logistic.jl:78-95,107) -- but the affine bounds (src/fact_samplers.jl:50-54) no longer gather
 * Γ[:,j]·x and Γ[:,j]·θ from a moved neighbourhood: every coordinate carries g_j = Γ[:,j]·x and gd_j = Γ[:,j]·θ at time tg_j, advanced
 * exactly as in spdmp_zigzag_tracked above.  So a proposal moves nothing but coordinate i itself (its position enters the prior term and the
 * event record) and what the sampled rows read; smove_forward!(G, i, ...) (:82) and smove_forward!(G2, i, ...) (:129) have no reader left.
 *   proposal of i at t′:  x_i to t′;  l = (∇ϕmoving θ_i)⁺;  lb = (a_i + b_i (t′ − t_old_i))⁺;  coin                          (:116-121)
 *   reject:               a = c_i + (g_i + gd_i (t′ − tg_i) − (Γμ)_i) θ_i,  b = c_i/100 + θ_i gd_i,  t_old = t′,  new key     (:137-140)
 *   accept:               θ_i = −θ_i;  j ∈ G1[i] ascending: g_j += gd_j (t′ − tg_j), gd_j += Γ[j,i]·(−2θ_i), tg_j = t′, bound, key (:130-135)
 * Same queue rule as the other tracked function (lowest coordinate on exactly tied keys), same draws in the same order as the moving
 * evaluation.  The clocks returned in t are the tracked process's own (a coordinate is as old as its last own event or its last visit by a
 * sampled row), the positions are the positions at those clocks.  Γ must be symmetric.
 */
static int spdmp_zigzag_tracked_lg(int64_t d, const orc_zz_params* p, double t0, double T, double* x, double* th, double* c, double* t,
                                   int64_t* acc, orc_trace* tr, orc_zz_result* res) {
    const orc_csc* Gb = p->bound_gamma;
    if (p->flow_kind || p->lambda_ref > 0 || p->move_all || p->local_bound || p->adaptscale || p->nbr_G) return ORC_BAD_INPUT;
    double* gmu_b = (double*)malloc((size_t)d * sizeof(double));
    double* g = (double*)malloc((size_t)d * sizeof(double));
    double* gd = (double*)malloc((size_t)d * sizeof(double));
    double* tg = (double*)malloc((size_t)d * sizeof(double));
    double* ba = (double*)malloc((size_t)d * sizeof(double));
    double* bb = (double*)malloc((size_t)d * sizeof(double));
    double* t_old = (double*)malloc((size_t)d * sizeof(double));
    const uint64_t seed = p->seed;
    uint64_t nm = 0, ng = 0;
    orc_pq* Q = orc_pq_new(d + 1);
    Q->lex = 1;
    for (int64_t i = 0; i < d; ++i) {
        gmu_b[i] = orc_idot(Gb, i, p->bound_mu);
        g[i] = orc_idot(Gb, i, x);
        gd[i] = orc_idot(Gb, i, th);
        t[i] = tg[i] = t_old[i] = t0;
        acc[i] = 0;
        ba[i] = c[i] + (g[i] - gmu_b[i]) * th[i]; /* src/fact_samplers.jl:51 */
        bb[i] = c[i] / 100 + th[i] * gd[i];       /* :52 */
    }
    for (int64_t i = 0; i < d; ++i) /* src/sfact.jl:186 */
        orc_pq_enqueue(Q, i, orc_poisson_time(ba[i], bb[i], pdmp_u01(seed, PDMP_STREAM_MAIN, nm++)));
    int64_t num = 0, nacc = 0;
    int status = ORC_OK;
    double tp = t0;
    int done = 0;
    while (!done && tp < T) { /* :199 */
        for (;;) {
            int64_t i;
            double tq;
            orc_pq_peek(Q, &i, &tq); /* :77 */
            if (p->stop_before_T && !(tq < T)) {
                done = 1;
                break;
            }
            if (tq == INFINITY) {
                status = ORC_STALLED;
                done = 1;
                break;
            }
            tp = tq;
            move1(i, t, x, th, tp);
            const double gi = logistic_grad_moving(p, i, t, x, th, tp, seed, &ng);
            const double l = pos(gi * th[i]);                      /* :119 */
            const double lb = pos(ba[i] + bb[i] * (tp - t_old[i])); /* :119 */
            num += 1;
            if (pdmp_u01(seed, PDMP_STREAM_MAIN, nm++) * lb < l) { /* :121 */
                acc[i] += 1;
                nacc += 1;
                if (l >= lb) { /* :123 */
                    if (!p->adapt) {
                        status = ORC_BOUND_VIOLATED;
                        done = 1;
                        break;
                    }
                    c[i] *= p->factor; /* :127 */
                }
                const double th_i = th[i];
                const double delta = -th_i - th_i;
                th[i] = -th_i; /* :130 */
                for (int64_t q = Gb->colptr[i]; q < Gb->colptr[i + 1]; ++q) { /* :131-135 */
                    const int64_t j = Gb->rowval[q];
                    const double gj = g[j] + gd[j] * (tp - tg[j]);
                    const double gdj = gd[j] + Gb->nzval[q] * delta; /* Γ[j, i] = Γ[i, j] */
                    g[j] = gj;
                    gd[j] = gdj;
                    tg[j] = tp;
                    ba[j] = c[j] + (gj - gmu_b[j]) * th[j];
                    bb[j] = c[j] / 100 + th[j] * gdj;
                    t_old[j] = tp;
                    const double L = pdmp_log(pdmp_u01(seed, PDMP_STREAM_MAIN, nm++));
                    orc_pq_set(Q, j, tp + poisson_time_L(ba[j], bb[j], L));
                }
                trace_push(tr, tp, i, x[i], th[i]); /* :143 */
                break;
            } else { /* :136-140 */
                const double g_now = g[i] + gd[i] * (tp - tg[i]);
                ba[i] = c[i] + (g_now - gmu_b[i]) * th[i];
                bb[i] = c[i] / 100 + th[i] * gd[i];
                t_old[i] = tp;
                const double L = pdmp_log(pdmp_u01(seed, PDMP_STREAM_MAIN, nm++));
                orc_pq_set(Q, i, tp + poisson_time_L(ba[i], bb[i], L));
                continue;
            }
        }
        if (p->max_events > 0 && nacc >= p->max_events && !done) {
            status = ORC_TRACE_LIMIT;
            done = 1;
        }
    }
    if (res) {
        res->num = num;
        res->nacc = nacc;
        res->nrefresh = 0;
        res->ndraw_main = nm;
        res->ndraw_global = ng;
        res->t_last = tp;
        res->status = status;
    }
    orc_pq_free(Q);
    free(gmu_b); free(g); free(gd); free(tg); free(ba); free(bb); free(t_old);
    return status;
}

static nbr_graph graph_g2x(const nbr_graph* g1, const nbr_graph* gsub, int64_t d);

/* @assert all(a.second ⊇ b.second for (a,b) in zip(G, G1)), src/sfact.jl:177 (both ascending) */
static int graph_contains(const nbr_graph* g, const nbr_graph* g1, int64_t d) {
    for (int64_t i = 0; i < d; ++i) {
        int64_t q = g->ptr[i];
        for (int64_t p = g1->ptr[i]; p < g1->ptr[i + 1]; ++p) {
            while (q < g->ptr[i + 1] && g->idx[q] < g1->idx[p]) ++q;
            if (q == g->ptr[i + 1] || g->idx[q] != g1->idx[p]) return 0;
        }
    }
    return 1;
}

int orc_spdmp_zigzag(int64_t d, const orc_zz_params* p, double t0, double T, double* x, double* th,
                     double* c, double* t, int64_t* acc, orc_trace* tr, orc_zz_result* res) {
    if (p->tracked && p->target_kind == 1) return spdmp_zigzag_tracked_lg(d, p, t0, T, x, th, c, t, acc, tr, res);
    if (p->tracked) return spdmp_zigzag_tracked(d, p, t0, T, x, th, c, t, acc, tr, res);
    zz_ctx cx;
    cx.d = d;
    cx.p = p;
    cx.g1 = graph_g1(p->bound_gamma);
    /* the optional argument G (src/sfact.jl:162,171-179): what a proposal moves (:82); Matched() = G1 */
    nbr_graph gG = p->nbr_G ? graph_g1(p->nbr_G) : cx.g1;
    if (p->nbr_G && !graph_contains(&gG, &cx.g1, d)) {
        graph_free(&gG);
        graph_free(&cx.g1);
        return ORC_BAD_INPUT;
    }
    if (!p->move_all) {
        cx.g2 = p->nbr_G ? graph_g2x(&cx.g1, &gG, d) : graph_g2(&cx.g1, d); /* src/sfact.jl:178 */
    } else {
        cx.g2.ptr = cx.g2.idx = NULL; /* G2 = nothing, src/sfact.jl:175 */
    }
    cx.gmu_bound = (double*)malloc((size_t)d * sizeof(double));
    for (int64_t i = 0; i < d; ++i) cx.gmu_bound[i] = orc_idot(p->bound_gamma, i, p->bound_mu);
    cx.gmu_target = NULL;
    if (p->target_mu) {
        cx.gmu_target = (double*)malloc((size_t)d * sizeof(double));
        for (int64_t i = 0; i < d; ++i) cx.gmu_target[i] = orc_idot(p->target_gamma, i, p->target_mu);
    }
    const int kind = p->flow_kind;
    const double* fmu = p->bound_mu;
    double* fdiag = (double*)malloc((size_t)d * sizeof(double)); /* Γ[i,i] */
    for (int64_t i = 0; i < d; ++i) {
        fdiag[i] = 0.0;
        for (int64_t q = p->bound_gamma->colptr[i]; q < p->bound_gamma->colptr[i + 1]; ++q)
            if (p->bound_gamma->rowval[q] == i) fdiag[i] = p->bound_gamma->nzval[q];
    }
    const double rhobar = sqrt(1 - p->rho * p->rho);
    double* sig = NULL; /* F.σ: mutable under adaptscale */
    if (p->sigma) {
        sig = (double*)malloc((size_t)d * sizeof(double));
        memcpy(sig, p->sigma, (size_t)d * sizeof(double));
    }
#define FLOW_AB(j, aa, bb)                                                             \
    do {                                                                               \
        if (kind)                                                                      \
            boom_ab(p->bound_gamma, fmu, fdiag, (j), x, th, c, (aa), (bb));              \
        else                                                                           \
            zz_ab(p->bound_gamma, cx.gmu_bound, (j), x, th, c, (aa), (bb));             \
    } while (0)
    const int hasrefresh = kind ? 1 : (p->lambda_ref > 0); /* src/fact_samplers.jl:18-19 */
    /* C::LocalBound (src/local.jl): b[j] = ab(G, j, x, θ, C, ∇ϕj, vj, Z) = (c_j + ∇ϕj θ_j, c_j/100 + vj, 2/c_j/|θ_j|) (:2-6) with the
     * target's own derivatives (∇ϕj, vj) = (idot(Γt, j, x) − ..., θ_j·idot(Γt, j, θ)) (performance/smartbound.jl:45-59), and
     * τ, renew[j] = next_time(t[j], b[j], rand(rng)) (src/not_fact_samplers.jl:43-50): the bound expires after its horizon. */
    unsigned char* renew = p->local_bound ? (unsigned char*)calloc((size_t)d, 1) : NULL;
#define REQUEUE(j, tbase)                                                                                   \
    do {                                                                                                    \
        if (p->local_bound) {                                                                               \
            const double gj_ = zz_grad(&cx, (j), x);                                                        \
            const double vj_ = th[(j)] * orc_idot(p->target_gamma, (j), th);                                \
            ba[(j)] = c[(j)] + gj_ * th[(j)];                                                               \
            bb[(j)] = c[(j)] / 100 + vj_;                                                                   \
            const double hz_ = 2.0 / c[(j)] / fabs(th[(j)]);                                                \
            const double dt_ = orc_poisson_time(ba[(j)], bb[(j)], pdmp_u01(seed, PDMP_STREAM_MAIN, nm++));  \
            renew[(j)] = dt_ > hz_;                                                                         \
            orc_pq_set(Q, (j), (tbase) + (renew[(j)] ? hz_ : dt_));                                         \
        } else {                                                                                            \
            FLOW_AB((j), &ba[(j)], &bb[(j)]);                                                               \
            orc_pq_set(Q, (j), (tbase) + orc_poisson_time(ba[(j)], bb[(j)], pdmp_u01(seed, PDMP_STREAM_MAIN, nm++))); \
        }                                                                                                   \
    } while (0)
    const uint64_t seed = p->seed;
    uint64_t nm = 0, ng = 0; /* draw counters: main stream, "global rng" stream */

    double* t_old = (double*)malloc((size_t)d * sizeof(double));
    double* ba = (double*)malloc((size_t)d * sizeof(double));
    double* bb = (double*)malloc((size_t)d * sizeof(double));
    for (int64_t i = 0; i < d; ++i) {
        t[i] = t0; /* src/sfact.jl:168 */
        t_old[i] = t0;
        acc[i] = 0;
    }
    orc_pq* Q = orc_pq_new(d + 1);
    if (p->local_bound) {
        /* src/local.jl:119-124: b[i] = ab(...), τ, renew[i] = next_time(t[i], b[i], rand(rng)), enqueue!(Q, i => τ) (τ includes t0) */
        for (int64_t i = 0; i < d; ++i) {
            const double gi_ = zz_grad(&cx, i, x);
            const double vi_ = th[i] * orc_idot(p->target_gamma, i, th);
            ba[i] = c[i] + gi_ * th[i];
            bb[i] = c[i] / 100 + vi_;
            const double hz = 2.0 / c[i] / fabs(th[i]);
            const double dt = orc_poisson_time(ba[i], bb[i], pdmp_u01(seed, PDMP_STREAM_MAIN, nm++));
            renew[i] = dt > hz;
            orc_pq_enqueue(Q, i, t0 + (renew[i] ? hz : dt));
        }
    } else {
        for (int64_t i = 0; i < d; ++i) FLOW_AB(i, &ba[i], &bb[i]); /* :184 */
        for (int64_t i = 0; i < d; ++i) {
            /* :186  enqueue!(Q, i => poisson_time(b[i], rand(rng)))   (t0 is NOT added in the reference) */
            orc_pq_enqueue(Q, i, orc_poisson_time(ba[i], bb[i], pdmp_u01(seed, PDMP_STREAM_MAIN, nm++)));
        }
    }
    if (hasrefresh) {
        /* :189  waiting_time_ref(rng, F) = randexp(rng)/λref, src/dynamics.jl:100 */
        orc_pq_enqueue(Q, d, -pdmp_log(pdmp_u01(seed, PDMP_STREAM_MAIN, nm++)) / p->lambda_ref);
    }

    int64_t num = 0, nacc = 0, nrefresh = 0;
    int status = ORC_OK;
    double tp = t0; /* t′ */
    int done = 0;
    /* driver loop `while t′ < T`, src/sfact.jl:199-208, around spdmp_inner!, :73-145 */
    while (!done && tp < T) {
        for (;;) { /* spdmp_inner!: while true */
            int64_t i;
            double tq;
            orc_pq_peek(Q, &i, &tq); /* :77 */
            if (p->stop_before_T && !(tq < T)) {
                done = 1;
                break;
            }
            if (tq == INFINITY) {
                status = ORC_STALLED;
                done = 1;
                break;
            }
            tp = tq;
            int refresh = i >= d; /* :78 */
            if (refresh) i = (int64_t)pdmp_randint(seed, PDMP_STREAM_GLOBAL, ng++, (uint32_t)d); /* :80 */
            if (p->move_all) {
                flow_move_all(kind, fmu, d, t, x, th, tp); /* :19 */
            } else {
                flow_move_nbrs(kind, fmu, &gG, i, t, x, th, tp); /* :82 (G; Matched: G1) */
            }
            if (refresh) {
                i = (int64_t)pdmp_randint(seed, PDMP_STREAM_GLOBAL, ng++, (uint32_t)d); /* :84 */
                if (!p->move_all) flow_move_nbrs(kind, fmu, &cx.g2, i, t, x, th, tp);    /* :85 */
                if (p->adaptscale && !kind) { /* :86-91 */
                    const double adapt_g = 0.01, adapt_t0 = 15., adapt_k = 0.75;
                    const double pre = pdmp_log(2.0) - sqrt(1.0 + tp) / (adapt_g * (1.0 + tp + adapt_t0)) *
                                                           pdmp_log((double)(1 + acc[i]) / (1.0 + 0.3 * tp));
                    const double eta = pdmp_exp(-adapt_k * pdmp_log(1 + tp)); /* (1 + t′)^(-adapt_κ) */
                    sig[i] = pdmp_exp(eta * pre + (1 - eta) * pdmp_log(sig[i]));
                    th[i] = sig[i] * ((th[i] > 0) ? 1.0 : ((th[i] < 0) ? -1.0 : th[i])); /* σ[i]*sign(θ[i]) */
                } else {
                    if (p->adaptscale) { /* :93-98 */
                        const double effi = (1 + 2 * p->rho / (1 - p->rho));
                        const double tau = effi / (t[i] * p->lambda_ref);
                        if (tau < 0.2) {
                            const double r = 0.3 * t[i] / (double)acc[i];
                            const double dir = (double)((r > 1.66) - (r < 0.6));
                            const double sq = sqrt(tau / p->lambda_ref);
                            sig[i] = sig[i] * pdmp_exp(dir * 0.03 * ((1.0 < sq) ? 1.0 : sq));
                        }
                    }
                    if (kind) {
                        /* :103  θ[i] = F.ρ*θ[i] + F.ρ̄*F.σ[i]*randn(rng, eltype(θ)) */
                        th[i] = p->rho * th[i] + rhobar * sig[i] * pdmp_randn(seed, PDMP_STREAM_MAIN, nm++);
                    } else {
                        /* :100-101  θ[i] = F.σ[i]*rand(rng, (-1,1)) */
                        double u = pdmp_u01(seed, PDMP_STREAM_MAIN, nm++);
                        th[i] = sig[i] * ((u < 0.5) ? -1.0 : 1.0);
                    }
                }
                /* :108  Q[n+1] = t′ + waiting_time_ref(F)  (global rng) */
                orc_pq_set(Q, d, tp + (-pdmp_log(pdmp_u01(seed, PDMP_STREAM_GLOBAL, ng++)) / p->lambda_ref));
                for (int64_t q = cx.g1.ptr[i]; q < cx.g1.ptr[i + 1]; ++q) { /* :110-114 */
                    int64_t j = cx.g1.idx[q];
                    FLOW_AB(j, &ba[j], &bb[j]);
                    t_old[j] = t[j];
                    orc_pq_set(Q, j, t[j] + orc_poisson_time(ba[j], bb[j], pdmp_u01(seed, PDMP_STREAM_MAIN, nm++)));
                }
                nrefresh++;
                trace_push(tr, t[i], i, x[i], th[i]); /* :143 */
                break;
            }
            if (p->local_bound && renew[i]) { /* src/local.jl:36-43: the bound of i expired -- renew it, no proposal */
                t_old[i] = t[i];
                REQUEUE(i, t[i]);
                continue;
            }
            double gi = (p->target_kind == 1) ? logistic_grad_moving(p, i, t, x, th, tp, seed, &ng)
                                              : zz_grad(&cx, i, x); /* :118 */
            double l = kind ? pos((gi - (x[i] - fmu[i]) * fdiag[i]) * th[i])  /* src/fact_samplers.jl:37-39 */
                            : pos(gi * th[i]);                               /* :119, src/fact_samplers.jl:28-30 */
            double lb = pos(ba[i] + bb[i] * (t[i] - t_old[i])); /* :119, src/sfact.jl:70 */
            num += 1;                                           /* :120 */
            if (pdmp_u01(seed, PDMP_STREAM_MAIN, nm++) * lb < l) { /* :121 */
                acc[i] += 1;
                nacc += 1;
                if (l >= lb) { /* :123 */
                    if (!p->adapt) {
                        status = ORC_BOUND_VIOLATED; /* :124 error(...) */
                        done = 1;
                        break;
                    }
                    c[i] *= p->factor; /* :127, src/fact_samplers.jl:67-70 */
                }
                if (!p->move_all) flow_move_nbrs(kind, fmu, &cx.g2, i, t, x, th, tp); /* :129 */
                th[i] = -th[i];                                        /* :130, src/dynamics.jl:46-49 */
                for (int64_t q = cx.g1.ptr[i]; q < cx.g1.ptr[i + 1]; ++q) { /* :131-135; src/local.jl:61-67 */
                    int64_t j = cx.g1.idx[q];
                    t_old[j] = t[j];
                    REQUEUE(j, t[j]);
                }
                trace_push(tr, t[i], i, x[i], th[i]); /* :143 with event() :50-52 */
                break;
            } else { /* :136-140; src/local.jl:68-73 */
                t_old[i] = t[i];
                REQUEUE(i, t[i]);
                continue;
            }
        }
        if (p->max_events > 0 && (nacc + nrefresh) >= p->max_events && !done) {
            status = ORC_TRACE_LIMIT;
            done = 1;
        }
    }
    if (res) {
        res->num = num;
        res->nacc = nacc;
        res->nrefresh = nrefresh;
        res->ndraw_main = nm;
        res->ndraw_global = ng;
        res->t_last = tp;
        res->status = status;
    }
    if (p->sigma_out && sig) memcpy(p->sigma_out, sig, (size_t)d * sizeof(double));
    free(sig);
    orc_pq_free(Q);
    free(renew);
    free(t_old);
    free(ba);
    free(bb);
    free(fdiag);
#undef REQUEUE
#undef FLOW_AB
    free(cx.gmu_bound);
    free(cx.gmu_target);
    if (p->nbr_G) graph_free(&gG);
    graph_free(&cx.g1);
    if (cx.g2.ptr) graph_free(&cx.g2);
    return status;
}

/* ------------------------------------------------------------------ 1-d ZigZag: src/zigzagboom1d.jl */

int64_t orc_pdmp_zigzag1d(double mu, double sigma2, double x, double th, double T, double c, int adapt,
                          double factor, uint64_t seed, orc_event1d* out, int64_t cap, int64_t* acc_out,
                          int64_t* num_out) {
    uint64_t nm = 0;
    double t = 0.0; /* :35 */
    int64_t n = 0;
    if (n < cap) {
        out[n].t = t;
        out[n].x = x;
        out[n].theta = th;
    }
    n++; /* :36 */
    /* t_ref = t + waiting_time_ref(ZigZag1d) = Inf, :19,37 -> the refresh branch :42-46 never fires */
    int64_t num = 0, acc = 0;
    double a = c + th * x, b = th * th; /* ab, :15 */
    double tp = t + orc_poisson_time(a, b, pdmp_u01(seed, PDMP_STREAM_MAIN, nm++)); /* :40 */
    while (t < T) {                                                                    /* :41 */
        double tau = tp - t;                                                           /* :48 */
        t = tau + t;                                                                   /* move_forward, src/dynamics.jl:66-68 */
        x = x + th * tau;
        double gx = (x - mu) / sigma2;  /* ∇ϕ(x), test/test1d.jl:9 */
        double l = pos(th * gx);        /* λ, :5 */
        double lb = pos(a + b * tau);   /* λ_bar, :9,50 */
        num += 1;
        if (pdmp_u01(seed, PDMP_STREAM_MAIN, nm++) * lb < l) { /* :52 */
            acc += 1;
            if (l >= lb) { /* :54 */
                if (!adapt) return -1;
                c *= factor;
            }
            th = -th; /* :58 */
            if (n < cap) {
                out[n].t = t;
                out[n].x = x;
                out[n].theta = th;
            }
            n++; /* :60 */
        }
        a = c + th * x; /* :63 */
        b = th * th;
        tp = t + orc_poisson_time(a, b, pdmp_u01(seed, PDMP_STREAM_MAIN, nm++)); /* :64 */
    }
    *acc_out = acc;
    *num_out = num;
    return n;
}

/* The same loop for both 1-d flows (src/zigzagboom1d.jl:34-67) with the gradient of the reference's own test, test/test1d.jl:9-10:
 * ∇ϕ(x) = (x − μ)/σ² + noise·(rand() − 0.5).  Every random number of this function is a call on the global generator in the
 * reference; here they are the draws of the chain's MAIN stream in program order: [randexp of the first refresh time (Boomerang1d, :37)],
 * the first event time (:40), then per iteration either {randn (:44), randexp (:45)} or {[the gradient's noise (:50)], the coin (:52)},
 * followed by the next event time (:64).  state (in/out) makes a run resumable when the event buffer fills. */
int64_t orc_pdmp_1d(const orc_1d_params* p, orc_1d_state* st, double T, orc_event1d* out, int64_t cap) {
    const int boom = p->flow == 1;
    double t = st->t, x = st->x, th = st->theta, c = st->c, a = st->a, b = st->b, tp = st->t_next, t_ref = st->t_ref;
    uint64_t nm = st->ndraw;
    int64_t num = st->num, acc = st->acc, n = 0;
    if (!st->started) {
        t = 0.0; /* :35 */
        if (n < cap) {
            out[n].t = t;
            out[n].x = x;
            out[n].theta = th;
        }
        n++; /* :36 */
        t_ref = boom ? t + pdmp_randexp_from_u(pdmp_u01(p->seed, PDMP_STREAM_MAIN, nm++)) / p->b_lambda : INFINITY; /* :19-20,37 */
        if (boom) {
            a = sqrt(th * th + (x - p->b_mu) * (x - p->b_mu)) * c; /* ab, :16 */
            b = 0.0;
        } else {
            a = c + th * x; /* ab, :15 */
            b = th * th;
        }
        tp = t + orc_poisson_time(a, b, pdmp_u01(p->seed, PDMP_STREAM_MAIN, nm++)); /* :40 */
        st->started = 1;
    }
    int status = 0;
    while (t < T) { /* :41 */
        if (n >= cap) {
            status = 3; /* event buffer full: state saved, call again */
            break;
        }
        if (t_ref < tp) { /* :42 */
            const double tau = t_ref - t;
            double sn, cs;
            pdmp_sincos(tau, &sn, &cs); /* move_forward, src/dynamics.jl:79-82 (only Boomerang1d has a finite t_ref) */
            const double xn = (x - p->b_mu) * cs + th * sn + p->b_mu;
            t = t + tau;
            x = xn;
            th = sqrt(p->b_sigma) * pdmp_randn(p->seed, PDMP_STREAM_MAIN, nm++);                      /* :44 */
            t_ref = t + pdmp_randexp_from_u(pdmp_u01(p->seed, PDMP_STREAM_MAIN, nm++)) / p->b_lambda; /* :45 */
            out[n].t = t;
            out[n].x = x;
            out[n].theta = th;
            n++; /* :46 */
        } else {
            const double tau = tp - t; /* :48 */
            if (boom) {
                double sn, cs;
                pdmp_sincos(tau, &sn, &cs);
                const double xn = (x - p->b_mu) * cs + th * sn + p->b_mu, tn = -(x - p->b_mu) * sn + th * cs;
                x = xn;
                th = tn;
                t = t + tau;
            } else {
                t = tau + t; /* src/dynamics.jl:66-68 */
                x = x + th * tau;
            }
            double gx = (x - p->mu) / p->sigma2;                                                        /* test/test1d.jl:9 */
            if (p->noise != 0.0) gx = gx + p->noise * (pdmp_u01(p->seed, PDMP_STREAM_MAIN, nm++) - 0.5); /* :10 */
            const double l = boom ? pos(th * (gx - (x - p->b_mu) / p->b_sigma)) : pos(th * gx); /* λ, :5-6 */
            const double lb = pos(a + b * tau);                                                 /* λ_bar, :9,50 */
            num += 1;
            if (pdmp_u01(p->seed, PDMP_STREAM_MAIN, nm++) * lb < l) { /* :52 */
                acc += 1;
                if (l >= lb) { /* :54 */
                    if (!p->adapt) {
                        status = 1; /* error("Tuning parameter `c` too small."), :55 */
                        break;
                    }
                    c *= p->factor; /* :56 */
                }
                th = -th; /* :58 */
                out[n].t = t;
                out[n].x = x;
                out[n].theta = th;
                n++; /* :60 */
            }
        }
        if (boom) {
            a = sqrt(th * th + (x - p->b_mu) * (x - p->b_mu)) * c; /* :63 */
            b = 0.0;
        } else {
            a = c + th * x;
            b = th * th;
        }
        tp = t + orc_poisson_time(a, b, pdmp_u01(p->seed, PDMP_STREAM_MAIN, nm++)); /* :64 */
    }
    st->t = t;
    st->x = x;
    st->theta = th;
    st->c = c;
    st->a = a;
    st->b = b;
    st->t_next = tp;
    st->t_ref = t_ref;
    st->ndraw = nm;
    st->num = num;
    st->acc = acc;
    st->status = status;
    return n;
}

/* ------------------------------------------------------------------ BPS: src/not_fact_samplers.jl */

/*
 * dot(a, b) over d elements.  The reference calls BLAS ddot (summation order unspecified); the order is
 * FIXED here to the one a 64-lane wavefront uses: lane l sums elements l, l+64, l+128, ... in order,
 * then lanes are combined by the xor-butterfly 1,2,4,8,16,32 (quads, rows of 16, then the four rows: the order the
 * gfx950 DPP reduction takes).
 */
static double dot_wave64(const double* a, const double* b, int64_t d) {
    double part[64];
    for (int l = 0; l < 64; ++l) {
        double s = 0.0;
        for (int64_t k = l; k < d; k += 64) s += a[k] * b[k];
        part[l] = s;
    }
    for (int off = 1; off <= 32; off <<= 1) {
        double nxt[64];
        for (int l = 0; l < 64; ++l) nxt[l] = part[l] + part[l ^ off];
        memcpy(part, nxt, sizeof part);
    }
    return part[0];
}

/* y = Γ (x - μ): Wrapper(∇ϕ!) with ∇ϕ!(y,x) = mul!(y, Γ, x) (test/maintest.jl:163), Γ symmetric so the
 * CSC column gather equals the row product; per output the sum runs in ascending index order. */
static void bps_grad(const orc_csc* G, const double* mu, const double* x, double* tmp, double* y, int64_t d) {
    for (int64_t k = 0; k < d; ++k) tmp[k] = x[k] - mu[k];
    for (int64_t r = 0; r < d; ++r) y[r] = orc_idot(G, r, tmp);
}

/* move_forward!(τ, t, x, θ, Flow): BouncyParticle src/dynamics.jl:11-15 (linear), Boomerang :29-36 (rotation about μ) */
static void nf_move(const orc_bps_params* p, int64_t d, double tau, double* x, double* th) {
    if (p->flow_kind == 0) {
        for (int64_t k = 0; k < d; ++k) x[k] += th[k] * tau;
    } else {
        double sn, cs;
        pdmp_sincos(tau, &sn, &cs);
        for (int64_t k = 0; k < d; ++k) {
            const double m = p->flow_mu[k];
            const double xn = (x[k] - m) * cs + th[k] * sn + m;
            const double tn = -(x[k] - m) * sn + th[k] * cs;
            x[k] = xn;
            th[k] = tn;
        }
    }
}
/*
 * Mass matrix: F.L = cholesky(Symmetric(Γ)).L (src/types.jl:43,66), a lower-triangular factor handed over in CSC form (rows
 * ascending, so the diagonal entry is the FIRST of its column).  The reference solves with LAPACK/CHOLMOD (order of operations
 * unspecified); the order is FIXED here to column-oriented substitution, which a wavefront reproduces exactly:
 *   y = L \ b :  for j = 0..d-1:    y_j = b_j / L_jj;  b_r -= L_rj * y_j  for the rows r > j of column j (ascending)
 *   z = L' \ y:  for j = d-1..0:    z_j = y_j / L_jj;  y_r -= L_jr * z_j  for the columns r < j of ROW j of L (ascending r)
 * (row j of L = column j of L', built once as a CSC transpose).  Every b_r receives its updates in the order of j, and no
 * update is a sum of more than one product, so no summation order is left open.
 */
typedef struct {
    int64_t d;
    int64_t* cp; /* L, diag first */
    int64_t* rv;
    double* nz;
    int64_t* tcp; /* L' (upper), diag last */
    int64_t* trv;
    double* tnz;
} tri_factor;

static int tri_build(tri_factor* F, const orc_csc* L) {
    const int64_t d = L->n, nnz = L->colptr[d];
    F->d = d;
    F->cp = (int64_t*)malloc((size_t)(d + 1) * sizeof(int64_t));
    F->rv = (int64_t*)malloc((size_t)(nnz ? nnz : 1) * sizeof(int64_t));
    F->nz = (double*)malloc((size_t)(nnz ? nnz : 1) * sizeof(double));
    F->tcp = (int64_t*)calloc((size_t)(d + 2), sizeof(int64_t));
    F->trv = (int64_t*)malloc((size_t)(nnz ? nnz : 1) * sizeof(int64_t));
    F->tnz = (double*)malloc((size_t)(nnz ? nnz : 1) * sizeof(double));
    memcpy(F->cp, L->colptr, (size_t)(d + 1) * sizeof(int64_t));
    memcpy(F->rv, L->rowval, (size_t)nnz * sizeof(int64_t));
    memcpy(F->nz, L->nzval, (size_t)nnz * sizeof(double));
    for (int64_t j = 0; j < d; ++j) {
        if (F->cp[j + 1] <= F->cp[j] || F->rv[F->cp[j]] != j) return -1; /* not lower triangular with a stored diagonal */
        for (int64_t p = F->cp[j]; p < F->cp[j + 1]; ++p) F->tcp[F->rv[p] + 2]++;
    }
    for (int64_t j = 0; j < d; ++j) F->tcp[j + 2] += F->tcp[j + 1];
    for (int64_t j = 0; j < d; ++j)
        for (int64_t p = F->cp[j]; p < F->cp[j + 1]; ++p) {
            int64_t q = F->tcp[F->rv[p] + 1]++;
            F->trv[q] = j;
            F->tnz[q] = F->nz[p];
        }
    return 0;
}
static void tri_free(tri_factor* F) {
    free(F->cp);
    free(F->rv);
    free(F->nz);
    free(F->tcp);
    free(F->trv);
    free(F->tnz);
}
/* b <- L \ b */
static void tri_solve_lower(const tri_factor* F, double* b) {
    for (int64_t j = 0; j < F->d; ++j) {
        const double yj = b[j] / F->nz[F->cp[j]];
        b[j] = yj;
        for (int64_t p = F->cp[j] + 1; p < F->cp[j + 1]; ++p) b[F->rv[p]] = b[F->rv[p]] - F->nz[p] * yj;
    }
}
/* y <- L' \ y */
static void tri_solve_upper(const tri_factor* F, double* y) {
    for (int64_t j = F->d - 1; j >= 0; --j) {
        const double zj = y[j] / F->tnz[F->tcp[j + 1] - 1];
        y[j] = zj;
        for (int64_t p = F->tcp[j]; p < F->tcp[j + 1] - 1; ++p) y[F->trv[p]] = y[F->trv[p]] - F->tnz[p] * zj;
    }
}

/* ∇ϕx = ∇ϕ!(∇ϕx, x) then grad_correct!, src/not_fact_samplers.jl:5-12: Boomerang subtracts L'\(L\(x − μ)) (= x − μ for L = I) */
static void nf_grad(const orc_bps_params* p, const tri_factor* M, int64_t d, const double* x, double* tmp, double* g) {
    if (p->flow_kind == 0 && p->target_gamma) bps_grad(p->target_gamma, p->target_mu, x, tmp, g, d); /* ∇ϕ! is the caller's, :122 */
    else bps_grad(p->gamma, p->mu, x, tmp, g, d);
    if (p->flow_kind == 1) {
        if (M) {
            for (int64_t k = 0; k < d; ++k) tmp[k] = x[k] - p->flow_mu[k];
            tri_solve_lower(M, tmp);
            tri_solve_upper(M, tmp);
            for (int64_t k = 0; k < d; ++k) g[k] -= tmp[k];
        } else {
            for (int64_t k = 0; k < d; ++k) g[k] -= x[k] - p->flow_mu[k];
        }
    }
}
/* ab(x, θ, C::GlobalBound, ∇ϕx, v, Flow), src/not_fact_samplers.jl:26-28 (BouncyParticle), :34-36 (Boomerang);
 * ab(x, θ, C::LocalBound, ∇ϕx, v, B::BouncyParticle) = (c + dot(θ, ∇ϕx), v, 2√d/c/‖θ‖₂), :29-31, with v = θ'Γθ, the second
 * directional derivative a `(∇ϕx, v)`-returning gradient callback supplies for the Gaussian target */
static void nf_ab(const orc_bps_params* p, int64_t d, double c, const double* x, const double* th, const double* g,
                  double* tmp, double* gth, double* a, double* b, double* horizon) {
    *horizon = INFINITY;
    if (p->flow_kind == 0) {
        /* GlobalBound: (c + θ'(B.Γ(x − B.μ)), θ'(B.Γθ), Inf) with the FLOW's Γ, μ (:26-28) -- which is θ'∇ϕx only when the target is
         * B.Γ(x − B.μ) itself; LocalBound: (c + dot(θ, ∇ϕx), v, ...) with the TARGET's gradient and second derivative v = θ'Γtθ (:29-31) */
        const int own_target = p->target_gamma != NULL;
        if (own_target && !p->local_bound) {
            bps_grad(p->gamma, p->mu, x, tmp, gth, d);
            *a = c + dot_wave64(th, gth, d);
        } else {
            *a = c + dot_wave64(th, g, d);
        }
        const orc_csc* Gv = (own_target && p->local_bound) ? p->target_gamma : p->gamma;
        for (int64_t r = 0; r < d; ++r) gth[r] = orc_idot(Gv, r, th);
        *b = dot_wave64(th, gth, d);
        if (p->local_bound) *horizon = 2 * sqrt((double)d) / c / sqrt(dot_wave64(th, th, d));
    } else {
        for (int64_t k = 0; k < d; ++k) tmp[k] = x[k] - p->flow_mu[k];
        *a = sqrt(dot_wave64(th, th, d) + dot_wave64(tmp, tmp, d)) * c; /* sqrt(normsq(θ) + normsq(x − μ))*C.c */
        *b = 0.0;
    }
}
/* next_time(t, abc, z), src/not_fact_samplers.jl:43-50 */
static double nf_next_time(double t, double a, double b, double horizon, double u, int* renew) {
    const double dt = orc_poisson_time(a, b, u);
    if (dt > horizon) {
        *renew = 1;
        return t + horizon;
    }
    *renew = 0;
    return t + dt;
}

int orc_pdmp_bps(int64_t d, const orc_bps_params* p, double t0, double T, double* x, double* th, double* t_ev,
                 double* x_ev, double* th_ev, int64_t ev_cap, orc_bps_result* res) {
    const uint64_t seed = p->seed;
    uint64_t nm = 0;
    double* g = (double*)malloc((size_t)d * sizeof(double));
    double* tmp = (double*)malloc((size_t)d * sizeof(double));
    double* gth = (double*)malloc((size_t)d * sizeof(double));
    double* w = (double*)malloc((size_t)d * sizeof(double));
    tri_factor Mf;
    const tri_factor* M = NULL;
    if (p->mass_L) {
        if (p->mass_L->n != d || tri_build(&Mf, p->mass_L) != 0) {
            free(g);
            free(tmp);
            free(gth);
            free(w);
            return ORC_BAD_INPUT;
        }
        M = &Mf;
    }
    double t = t0;
    double c = p->c;
    int64_t num = 0, acc = 0, nrefresh = 0, nev = 0;
    int status = ORC_OK;
    const double rho = p->rho, rhobar = sqrt(1 - rho * rho); /* src/dynamics.jl:113 */
    double a, b, hz;
    int renew = 0;

    double tau_ref = -pdmp_log(pdmp_u01(seed, PDMP_STREAM_MAIN, nm++)) / p->lambda_ref; /* :121 */
    nf_grad(p, M, d, x, tmp, g);                                                         /* :122-123 */
    nf_ab(p, d, c, x, th, g, tmp, gth, &a, &b, &hz);                                     /* :126 */
    double tp = nf_next_time(t, a, b, hz, pdmp_u01(seed, PDMP_STREAM_MAIN, nm++), &renew); /* :43-50, :135 */

    while (t < T) { /* :136 */
        for (;;) {  /* pdmp_inner!, :52-97 */
            if (tau_ref < tp) { /* :55 refresh */
                double tau = tau_ref - t;
                t += tau; /* move_forward!, :56 */
                nf_move(p, d, tau, x, th);
                /* refresh!, src/dynamics.jl:112-126:  θ .*= ρ; u = ρ̄*(L'\randn(rng, d)); θ .+= u */
                /* randn(rng, d): the reference draws d normals from its stream; here the d-vector comes from ⌈d/128⌉·64 Philox
                 * blocks, both Box-Muller branches of a block in use: element k = 128a + 64b + l (l < 64, b ∈ {0,1}) is branch b
                 * (cos, sin) of block nm + 64a + l -- the layout in which a 64-lane wavefront holds two elements per lane pair of
                 * slots, so that one Philox / log / sqrt / sincos evaluation yields two of its normals */
                for (int64_t k = 0; k < d; ++k) th[k] *= rho;
                for (int64_t k = 0; k < d; ++k) {
                    double z0, z1;
                    pdmp_randn2(seed, PDMP_STREAM_MAIN, nm + (uint64_t)(((k >> 7) << 6) + (k & 63)), &z0, &z1);
                    w[k] = ((k >> 6) & 1) ? z1 : z0;
                }
                if (M) tri_solve_upper(M, w);
                for (int64_t k = 0; k < d; ++k) th[k] += rhobar * w[k];
                nm += (uint64_t)(((d + 127) >> 7) << 6);
                nf_grad(p, M, d, x, tmp, g);                                                  /* :58-59 */
                tau_ref = t + (-pdmp_log(pdmp_u01(seed, PDMP_STREAM_MAIN, nm++)) / p->lambda_ref); /* :61 */
                nf_ab(p, d, c, x, th, g, tmp, gth, &a, &b, &hz);                              /* :62 */
                tp = nf_next_time(t, a, b, hz, pdmp_u01(seed, PDMP_STREAM_MAIN, nm++), &renew); /* :63 */
                nrefresh++;
                break; /* :64 return */
            }
            if (renew) { /* :65-71: the bound's horizon expired (LocalBound only): move, re-bound, no thinning step */
                double tau = tp - t;
                t += tau;
                nf_move(p, d, tau, x, th);
                nf_grad(p, M, d, x, tmp, g);
                nf_ab(p, d, c, x, th, g, tmp, gth, &a, &b, &hz);
                tp = nf_next_time(t, a, b, hz, pdmp_u01(seed, PDMP_STREAM_MAIN, nm++), &renew);
                continue;
            }
            double tau = tp - t; /* :73 */
            t += tau;
            nf_move(p, d, tau, x, th);   /* :74 */
            nf_grad(p, M, d, x, tmp, g); /* :75-76 */
            double gt = dot_wave64(g, th, d);
            double l = pos(gt);            /* λ, :14 */
            double lb = pos(a + b * tau);  /* :77 */
            num += 1;
            if (pdmp_u01(seed, PDMP_STREAM_MAIN, nm++) * lb <= l) { /* :79 */
                acc += 1;
                if (l > lb) { /* :81 */
                    if (!p->adapt) {
                        status = ORC_BOUND_VIOLATED;
                        goto finish;
                    }
                    c *= p->factor; /* :83 */
                }
                /* reflect!, src/dynamics.jl:90-97: θ .-= (2 dot(∇ϕx,θ)/normsq(L\∇ϕx)) (L'\(L\∇ϕx)) */
                if (M) {
                    memcpy(w, g, (size_t)d * sizeof(double));
                    tri_solve_lower(M, w);
                    double nrm = dot_wave64(w, w, d);
                    tri_solve_upper(M, w);
                    double coef = 2 * gt / nrm;
                    for (int64_t k = 0; k < d; ++k) th[k] -= coef * w[k];
                } else {
                    double nrm = dot_wave64(g, g, d);
                    double coef = 2 * gt / nrm;
                    for (int64_t k = 0; k < d; ++k) th[k] -= coef * g[k];
                }
                /* :86-87 gradient again (x unchanged: same values) */
                nf_ab(p, d, c, x, th, g, tmp, gth, &a, &b, &hz);                              /* :88 */
                tp = nf_next_time(t, a, b, hz, pdmp_u01(seed, PDMP_STREAM_MAIN, nm++), &renew); /* :89 */
                if (!p->subsample) break;                                                     /* :90 */
            } else {
                nf_ab(p, d, c, x, th, g, tmp, gth, &a, &b, &hz);                              /* :92 (recomputed: bit parity) */
                tp = nf_next_time(t, a, b, hz, pdmp_u01(seed, PDMP_STREAM_MAIN, nm++), &renew); /* :93 */
            }
        }
        /* push!(Ξ, event(t, x, θ, Flow)) = (t, copy(x), copy(θ), nothing), :138, :39-41 */
        if (nev < ev_cap) {
            if (t_ev) t_ev[nev] = t;
            if (x_ev) memcpy(x_ev + nev * d, x, (size_t)d * sizeof(double));
            if (th_ev) memcpy(th_ev + nev * d, th, (size_t)d * sizeof(double));
        }
        nev++;
        if (p->max_events > 0 && nev >= p->max_events) {
            status = ORC_TRACE_LIMIT;
            break;
        }
    }
finish:
    if (res) {
        res->num = num;
        res->nacc = acc;
        res->nrefresh = nrefresh;
        res->nevents = nev;
        res->ndraw_main = nm;
        res->t_last = t;
        res->c_out = c;
        res->status = status;
    }
    free(g);
    free(tmp);
    free(gth);
    free(w);
    if (M) tri_free(&Mf);
    return status;
}

/* ------------------------------------------------------------------ sticky ZigZag: src/ss_fact.jl */

/* freezing_time, src/ss_fact.jl:10-16 */
static inline double freezing_time(double x, double th) {
    if (th * x >= 0) return INFINITY;
    return -x / th;
}
/* ssmove_forward!(G, i, ...), src/ss_fact.jl:38-45 */
static inline void ssmove_nbrs(const nbr_graph* g, int64_t i, double* t, double* x, const double* th, double tp) {
    for (int64_t p = g->ptr[i]; p < g->ptr[i + 1]; ++p) {
        int64_t j = g->idx[p];
        if (th[j] != 0.0) move1(j, t, x, th, tp);
    }
}
/* queue_time!, src/ss_fact.jl:54-65 */
static inline void queue_time(orc_pq* Q, const double* t, const double* x, const double* th, int64_t i,
                              const double* ba, const double* bb, unsigned char* f, uint64_t seed, uint64_t* nm) {
    double trefl = orc_poisson_time(ba[i], bb[i], pdmp_u01(seed, PDMP_STREAM_MAIN, (*nm)++));
    double tfreeze = freezing_time(x[i], th[i]);
    if (tfreeze <= trefl) {
        f[i] = 1;
        orc_pq_set(Q, i, t[i] + tfreeze);
    } else {
        f[i] = 0;
        orc_pq_set(Q, i, t[i] + trefl);
    }
}

int orc_sspdmp_zigzag(int64_t d, const orc_sticky_params* p, double t0, double T, double* x, double* th,
                      double* c, double* t, orc_trace* tr, orc_zz_result* res) {
    nbr_graph g1 = graph_g1(p->bound_gamma);
    /* the optional argument G (src/ss_fact.jl:159,167-172): what ssmove_forward!(G, i, ...) moves; nothing = G1 */
    nbr_graph gG = p->nbr_G ? graph_g1(p->nbr_G) : g1;
    if (p->nbr_G && !graph_contains(&gG, &g1, d)) {
        graph_free(&gG);
        graph_free(&g1);
        return ORC_BAD_INPUT;
    }
    nbr_graph g2 = p->nbr_G ? graph_g2x(&g1, &gG, d) : graph_g2(&g1, d); /* :172 */
    double* gmu = (double*)malloc((size_t)d * sizeof(double));
    double* gmt = NULL;
    for (int64_t i = 0; i < d; ++i) gmu[i] = orc_idot(p->bound_gamma, i, p->bound_mu);
    if (p->target_mu && !p->logistic) {
        gmt = (double*)malloc((size_t)d * sizeof(double));
        for (int64_t i = 0; i < d; ++i) gmt[i] = orc_idot(p->target_gamma, i, p->target_mu);
    }
    const uint64_t seed = p->seed;
    uint64_t nm = 0, ng = 0;
    double* t_old = (double*)malloc((size_t)d * sizeof(double));
    double* ba = (double*)malloc((size_t)d * sizeof(double));
    double* bb = (double*)malloc((size_t)d * sizeof(double));
    double* thf = (double*)calloc((size_t)d, sizeof(double));        /* θf, :174 */
    unsigned char* f = (unsigned char*)calloc((size_t)d, 1);         /* :165 */
    for (int64_t i = 0; i < d; ++i) t[i] = t_old[i] = t0;            /* :163-164 */
    orc_pq* Q = orc_pq_new(d + 1);
    for (int64_t i = 0; i < d; ++i) zz_ab(p->bound_gamma, gmu, i, x, th, c, &ba[i], &bb[i]); /* :177 */
    for (int64_t i = 0; i < d; ++i) {                                                          /* :178-188 */
        double trefl = orc_poisson_time(ba[i], bb[i], pdmp_u01(seed, PDMP_STREAM_MAIN, nm++));
        double tfreez = freezing_time(x[i], th[i]);
        if (trefl > tfreez) {
            f[i] = 1;
            orc_pq_enqueue(Q, i, t0 + tfreez);
        } else {
            f[i] = 0;
            orc_pq_enqueue(Q, i, t0 + trefl);
        }
    }
    int64_t num = 0, acc = 0, nev = 0;
    int status = ORC_OK;
    double tp = t0;
    while (tp < T) { /* :202 */
        for (;;) {   /* sspdmp_inner!, :78-157 */
            int64_t i;
            double tq;
            orc_pq_peek(Q, &i, &tq); /* :83 */
            if (tq == INFINITY) {
                status = ORC_STALLED;
                goto finish;
            }
            tp = tq;
            if (f[i]) { /* :87 case 1: to be frozen */
                move1(i, t, x, th, tp); /* :88 */
                if (fabs(x[i]) > 1e-8) {  /* :89-91 */
                    status = ORC_BOUND_VIOLATED;
                    goto finish;
                }
                x[i] = 0.0 * th[i]; /* :92  x[i] = -0*θ[i]  (Int -0 == 0, so the sign is θ's) */
                thf[i] = th[i];
                th[i] = 0.0; /* :93 */
                t_old[i] = t[i];
                f[i] = 0;
                orc_pq_set(Q, i, t[i] - pdmp_log(pdmp_u01(seed, PDMP_STREAM_MAIN, nm++)) / p->kappa[i]); /* :96 */
                if (!p->strong_upperbounds) { /* :97-107 */
                    ssmove_nbrs(&gG, i, t, x, th, tp);
                    ssmove_nbrs(&g2, i, t, x, th, tp);
                    for (int64_t q = g1.ptr[i]; q < g1.ptr[i + 1]; ++q) {
                        int64_t j = g1.idx[q];
                        if (th[j] != 0) {
                            zz_ab(p->bound_gamma, gmu, j, x, th, c, &ba[j], &bb[j]);
                            t_old[j] = t[j];
                            queue_time(Q, t, x, th, j, ba, bb, f, seed, &nm);
                        }
                    }
                }
            } else if (x[i] == 0 && th[i] == 0) { /* :108 case 2: was frozen */
                t[i] = tp;                         /* :109 */
                th[i] = thf[i];
                thf[i] = 0.0; /* :110 */
                if (p->reversible) { /* :111-113 */
                    double u = pdmp_u01(seed, PDMP_STREAM_MAIN, nm++);
                    th[i] *= (u < 0.5) ? -1.0 : 1.0;
                }
                t_old[i] = t[i];
                ssmove_nbrs(&gG, i, t, x, th, tp); /* :115 */
                ssmove_nbrs(&g2, i, t, x, th, tp); /* :116 */
                for (int64_t q = g1.ptr[i]; q < g1.ptr[i + 1]; ++q) { /* :117-123 */
                    int64_t j = g1.idx[q];
                    if (th[j] != 0) {
                        zz_ab(p->bound_gamma, gmu, j, x, th, c, &ba[j], &bb[j]);
                        t_old[j] = t[j];
                        queue_time(Q, t, x, th, j, ba, bb, f, seed, &nm);
                    }
                }
            } else { /* :124 reflection proposal */
                ssmove_nbrs(&gG, i, t, x, th, tp); /* :125 */
                double gi;
                if (p->logistic) { /* ∇ϕ_(∇ϕ, t, x, θ, i, t′, F, S::SelfMoving, args...), src/sfact.jl:68 */
                    gi = logistic_grad_moving(p->logistic, i, t, x, th, tp, seed, &ng);
                } else {
                    gi = orc_idot(p->target_gamma, i, x);
                    if (gmt) gi = gi - gmt[i];
                }
                double l = pos(gi * th[i]);
                double lb = pos(ba[i] + bb[i] * (t[i] - t_old[i])); /* :128 */
                num += 1;
                if (pdmp_u01(seed, PDMP_STREAM_MAIN, nm++) * lb < l) { /* :130 */
                    acc += 1;
                    if (l > lb) { /* :132 */
                        if (!p->adapt) {
                            status = ORC_BOUND_VIOLATED;
                            goto finish;
                        }
                        acc = num = 0; /* :134 */
                        c[i] *= p->factor;
                    }
                    ssmove_nbrs(&g2, i, t, x, th, tp); /* :138 */
                    th[i] = -th[i];                    /* :139 */
                    for (int64_t q = g1.ptr[i]; q < g1.ptr[i + 1]; ++q) { /* :140-146 */
                        int64_t j = g1.idx[q];
                        if (th[j] != 0) {
                            zz_ab(p->bound_gamma, gmu, j, x, th, c, &ba[j], &bb[j]);
                            t_old[j] = t[j];
                            queue_time(Q, t, x, th, j, ba, bb, f, seed, &nm);
                        }
                    }
                } else { /* :147-151 */
                    zz_ab(p->bound_gamma, gmu, i, x, th, c, &ba[i], &bb[i]);
                    t_old[i] = t[i];
                    queue_time(Q, t, x, th, i, ba, bb, f, seed, &nm);
                    continue;
                }
            }
            trace_push(tr, t[i], i, x[i], th[i]); /* :154 */
            nev++;
            break; /* :155 */
        }
        if (p->max_events > 0 && nev >= p->max_events) {
            status = ORC_TRACE_LIMIT;
            break;
        }
    }
finish:
    if (res) {
        res->num = num;
        res->nacc = acc;
        res->nrefresh = 0;
        res->ndraw_main = nm;
        res->ndraw_global = ng;
        res->t_last = tp;
        res->status = status;
    }
    orc_pq_free(Q);
    free(t_old);
    free(ba);
    free(bb);
    free(thf);
    free(f);
    free(gmu);
    free(gmt);
    if (p->nbr_G) graph_free(&gG);
    graph_free(&g1);
    graph_free(&g2);
    return status;
}

/* ------------------------------------------------------------------ CPU baseline ensemble driver */

typedef struct {
    int64_t d;
    const orc_zz_params* p;
    double t0, T;
    int64_t nchains;
    const double *x0, *th0, *c;
    uint64_t seed0;
    int64_t next; /* atomic work counter */
    int64_t num, acc;
    pthread_mutex_t mu;
} ens_job;

static void* ens_worker(void* arg) {
    ens_job* job = (ens_job*)arg;
    int64_t d = job->d;
    double* x = (double*)malloc((size_t)d * sizeof(double));
    double* th = (double*)malloc((size_t)d * sizeof(double));
    double* c = (double*)malloc((size_t)d * sizeof(double));
    double* t = (double*)malloc((size_t)d * sizeof(double));
    int64_t* acc = (int64_t*)malloc((size_t)d * sizeof(int64_t));
    int64_t mynum = 0, myacc = 0;
    for (;;) {
        int64_t k = __atomic_fetch_add(&job->next, 1, __ATOMIC_RELAXED);
        if (k >= job->nchains) break;
        memcpy(x, job->x0 + k * d, (size_t)d * sizeof(double));
        memcpy(th, job->th0 + k * d, (size_t)d * sizeof(double));
        memcpy(c, job->c, (size_t)d * sizeof(double));
        orc_zz_params p = *job->p;
        p.seed = job->seed0 + (uint64_t)k;
        orc_zz_result r;
        orc_spdmp_zigzag(d, &p, job->t0, job->T, x, th, c, t, acc, NULL, &r);
        mynum += r.num;
        myacc += r.nacc;
    }
    pthread_mutex_lock(&job->mu);
    job->num += mynum;
    job->acc += myacc;
    pthread_mutex_unlock(&job->mu);
    free(x);
    free(th);
    free(c);
    free(t);
    free(acc);
    return NULL;
}

double orc_spdmp_zigzag_ensemble(int64_t d, const orc_zz_params* p, double t0, double T, int64_t nchains,
                                 const double* x0, const double* th0, const double* c, uint64_t seed0,
                                 int nthreads, int64_t* num_total, int64_t* acc_total) {
    ens_job job;
    memset(&job, 0, sizeof job);
    job.d = d;
    job.p = p;
    job.t0 = t0;
    job.T = T;
    job.nchains = nchains;
    job.x0 = x0;
    job.th0 = th0;
    job.c = c;
    job.seed0 = seed0;
    pthread_mutex_init(&job.mu, NULL);
    if (nthreads < 1) nthreads = 1;
    pthread_t* th = (pthread_t*)malloc((size_t)nthreads * sizeof(pthread_t));
    struct timespec ts0, ts1;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    for (int k = 0; k < nthreads; ++k) pthread_create(&th[k], NULL, ens_worker, &job);
    for (int k = 0; k < nthreads; ++k) pthread_join(th[k], NULL);
    clock_gettime(CLOCK_MONOTONIC, &ts1);
    free(th);
    pthread_mutex_destroy(&job.mu);
    *num_total = job.num;
    *acc_total = job.acc;
    return (double)(ts1.tv_sec - ts0.tv_sec) + 1e-9 * (double)(ts1.tv_nsec - ts0.tv_nsec);
}

/* ------------------------------------------------------------------ host side of the device math probe */
/* Same expressions as math_probe_kernel (zigzagboomerang.jl_amd/csrc/pdmp_kernels.hip), evaluated with the
 * oracle's own poisson_time and libm sqrt: out is [8 x n] row-major. */
/* ------------------------------------------------------------------ threaded local ZigZag: src/parallel.jl
 *
 * CPU BASELINE ONLY (north star: "the reference single-threaded and src/parallel.jl multithreaded CPU paths timed on the
 * box's own host cores").  One chain, K worker threads, each owning a contiguous chunk of k = d/K coordinates and its own
 * queue (Partition, :5-30).  A worker processes the "inner" coordinates of its chunk (all of G[i] inside the chunk, :114)
 * until the top of its queue is a boundary coordinate or lies more than Δ past its last hand-off (:76-77); it then clears
 * its latch bit and sleeps (:78-95).  When every worker sleeps, the coordinator (parallel_spdmp_outer!, :176-253) merges the
 * event lists, walks the reported coordinates in time order, processes those whose neighbouring chunks have all reached
 * that time with the same step (parallel_innermost!, :34-61) and wakes the chunks it served.  The bounding Γ must be block
 * diagonal over the chunks (:124-127), the target's neighbourhoods G may cross them.
 * Thread interleaving makes the event sequence run-dependent (as in the reference, whose workers seed Rng() themselves,
 * :68,183): pinned statistically only (test/testparallel.jl:59-73).  Workers draw from Philox stream 16 + thread id.
 */
#include <pthread.h>

typedef struct par_ctx par_ctx;
typedef struct {
    par_ctx* cx;
    int ti;
    pthread_t th;
    pthread_mutex_t m;   /* wakeup[ti] */
    pthread_cond_t cv;
    int wake, done;      /* set by the coordinator under m */
    /* ret[] = i, t′, acc, num (:78) */
    int64_t ret_i, ret_acc, ret_num;
    double ret_t;
    orc_trace events;
    uint64_t ndraw;
} par_worker;

struct par_ctx {
    int64_t d, k; /* chunk size */
    int K;
    const orc_zz_params* p;
    zz_ctx zc;
    nbr_graph g, g1, g2; /* G (target pattern), G1 (bound pattern), G2 = two-hop(G1) \ G */
    double *t, *x, *th, *c, *t_old, *ba, *bb;
    unsigned char* inner;
    orc_pq** Q;
    double t0, delta;
    int adapt;
    double factor;
    /* latch (:143): bit ti set while worker ti runs */
    pthread_mutex_t lm;
    pthread_cond_t lcv;
    uint64_t active;
    int error;
    par_worker* w;
};

static nbr_graph graph_g2x(const nbr_graph* g1, const nbr_graph* gsub, int64_t d) {
    nbr_graph g;
    g.ptr = (int64_t*)malloc((size_t)(d + 1) * sizeof(int64_t));
    int64_t cap = 16 * d + 16, n = 0;
    g.idx = (int64_t*)malloc((size_t)cap * sizeof(int64_t));
    int64_t* tmp = NULL;
    int64_t tmpcap = 0;
    for (int64_t i = 0; i < d; ++i) {
        g.ptr[i] = n;
        int64_t m = 0;
        for (int64_t q = g1->ptr[i]; q < g1->ptr[i + 1]; ++q) m += g1->ptr[g1->idx[q] + 1] - g1->ptr[g1->idx[q]];
        if (m > tmpcap) {
            tmpcap = 2 * m;
            tmp = (int64_t*)realloc(tmp, (size_t)tmpcap * sizeof(int64_t));
        }
        m = 0;
        for (int64_t q = g1->ptr[i]; q < g1->ptr[i + 1]; ++q) {
            int64_t j = g1->idx[q];
            for (int64_t r = g1->ptr[j]; r < g1->ptr[j + 1]; ++r) tmp[m++] = g1->idx[r];
        }
        qsort(tmp, (size_t)m, sizeof(int64_t), cmp_i64);
        int64_t last = -1;
        for (int64_t e = 0; e < m; ++e) {
            int64_t v = tmp[e];
            if (v == last) continue;
            last = v;
            int in_g = 0;
            for (int64_t q = gsub->ptr[i]; q < gsub->ptr[i + 1]; ++q)
                if (gsub->idx[q] == v) in_g = 1;
            if (in_g) continue;
            if (n == cap) {
                cap *= 2;
                g.idx = (int64_t*)realloc(g.idx, (size_t)cap * sizeof(int64_t));
            }
            g.idx[n++] = v;
        }
    }
    g.ptr[d] = n;
    free(tmp);
    return g;
}

/* parallel_innermost!, src/parallel.jl:34-61; returns 1 on an accepted reflection, -1 on a bound violation without adapt */
static int par_innermost(par_ctx* cx, uint64_t seed, uint32_t stream, uint64_t* nd, int64_t i, double tp) {
    double *t = cx->t, *x = cx->x, *th = cx->th;
    move_nbrs(&cx->g, i, t, x, th, tp);              /* :36 */
    const double gi = zz_grad(&cx->zc, i, x);         /* :37 */
    const double l = pos(gi * th[i]);                 /* :38 */
    const double lb = pos(cx->ba[i] + cx->bb[i] * (t[i] - cx->t_old[i]));
    if (pdmp_u01(seed, stream, (*nd)++) * lb < l) {   /* :40 */
        if (l >= lb) {                                /* :41 */
            if (!cx->adapt) return -1;
            cx->c[i] *= cx->factor;                   /* :43 */
        }
        move_nbrs(&cx->g2, i, t, x, th, tp);          /* :45 */
        th[i] = -th[i];                               /* :46 */
        for (int64_t q = cx->g1.ptr[i]; q < cx->g1.ptr[i + 1]; ++q) { /* :47-52 */
            const int64_t j = cx->g1.idx[q];
            zz_ab(cx->p->bound_gamma, cx->zc.gmu_bound, j, x, th, cx->c, &cx->ba[j], &cx->bb[j]);
            cx->t_old[j] = t[j];
            orc_pq_set(cx->Q[j / cx->k], j % cx->k, t[j] + orc_poisson_time(cx->ba[j], cx->bb[j], pdmp_u01(seed, stream, (*nd)++)));
        }
        return 1;
    }
    zz_ab(cx->p->bound_gamma, cx->zc.gmu_bound, i, x, th, cx->c, &cx->ba[i], &cx->bb[i]); /* :55-58 */
    cx->t_old[i] = t[i];
    orc_pq_set(cx->Q[i / cx->k], i % cx->k, t[i] + orc_poisson_time(cx->ba[i], cx->bb[i], pdmp_u01(seed, stream, (*nd)++)));
    return 0;
}

/* parallel_spdmp_inner!, src/parallel.jl:63-102 */
static void* par_worker_main(void* arg) {
    par_worker* w = (par_worker*)arg;
    par_ctx* cx = w->cx;
    const int ti = w->ti;
    const uint64_t seed = cx->p->seed;
    int64_t acc = 0, num = 0;
    double tnext = cx->t0 + cx->delta; /* :67 */
    for (;;) {
        num += 1; /* :70 */
        int64_t ii;
        double tp;
        orc_pq_peek(cx->Q[ti], &ii, &tp); /* :71 */
        const int64_t i = (int64_t)ti * cx->k + ii;
        if (!cx->inner[i] || tp > tnext) { /* :73 */
            tnext = tp + cx->delta;
            w->ret_i = i;
            w->ret_t = tp;
            w->ret_acc = acc;
            w->ret_num = num;
            pthread_mutex_lock(&w->m); /* lock(wakeup), :81 */
            pthread_mutex_lock(&cx->lm);
            cx->active &= ~(1ull << ti); /* :79 */
            if (cx->active == 0) pthread_cond_broadcast(&cx->lcv); /* last one turns the light off, :82-87 */
            pthread_mutex_unlock(&cx->lm);
            while (!w->wake) pthread_cond_wait(&w->cv, &w->m); /* :90 */
            w->wake = 0;
            const int done = w->done;
            pthread_mutex_unlock(&w->m);
            if (done) return NULL; /* :92-95 */
            acc = num = 0;         /* :96 */
        } else {
            const int ok = par_innermost(cx, seed, 16u + (uint32_t)ti, &w->ndraw, i, tp); /* :98 */
            if (ok < 0) {
                cx->error = 1; /* reference: error(...) inside the task; here the chunk simply parks for good */
                w->ret_i = i;
                w->ret_t = INFINITY;
                w->ret_acc = acc;
                w->ret_num = num;
                pthread_mutex_lock(&w->m);
                pthread_mutex_lock(&cx->lm);
                cx->active &= ~(1ull << ti);
                if (cx->active == 0) pthread_cond_broadcast(&cx->lcv);
                pthread_mutex_unlock(&cx->lm);
                while (!w->wake) pthread_cond_wait(&w->cv, &w->m);
                pthread_mutex_unlock(&w->m);
                return NULL;
            }
            if (!ok) continue; /* :99 */
            acc += 1;
            trace_push(&w->events, cx->t[i], i, cx->x[i], cx->th[i]); /* :101 */
        }
    }
}

int orc_parallel_spdmp(int64_t d, const orc_zz_params* p, int K, double delta, double t0, double T, double* x, double* th,
                       double* c, double* t_out, orc_trace* tr, orc_par_result* res) {
    if (K < 1 || K > 64 || d % K != 0) return ORC_BOUND_VIOLATED;
    par_ctx cx;
    memset(&cx, 0, sizeof cx);
    cx.d = d;
    cx.K = K;
    cx.k = d / K; /* Partition(nt, n) = Partition{div(n, nt)}, :26 */
    cx.p = p;
    cx.t0 = t0;
    cx.delta = delta;
    cx.adapt = p->adapt;
    cx.factor = p->factor;
    cx.x = x;
    cx.th = th;
    cx.c = c;
    cx.t = t_out;
    cx.g = graph_g1(p->target_gamma);      /* G = pattern of the target's Γ (test/testparallel.jl:49) */
    cx.g1 = graph_g1(p->bound_gamma);      /* :117 */
    cx.g2 = graph_g2x(&cx.g1, &cx.g, d);   /* :121 */
    cx.zc.d = d;
    cx.zc.p = p;
    cx.zc.gmu_bound = (double*)malloc((size_t)d * sizeof(double));
    for (int64_t i = 0; i < d; ++i) cx.zc.gmu_bound[i] = orc_idot(p->bound_gamma, i, p->bound_mu);
    cx.zc.gmu_target = NULL;
    if (p->target_mu) {
        cx.zc.gmu_target = (double*)malloc((size_t)d * sizeof(double));
        for (int64_t i = 0; i < d; ++i) cx.zc.gmu_target[i] = orc_idot(p->target_gamma, i, p->target_mu);
    }
    int status = ORC_OK;
    /* :124-127 "Upper bounds may not depend across chunks." (G1 ⊆ G is asserted at :119) */
    for (int64_t i = 0; i < d && status == ORC_OK; ++i)
        for (int64_t q = cx.g2.ptr[i]; q < cx.g2.ptr[i + 1]; ++q)
            if (cx.g2.idx[q] / cx.k != i / cx.k) status = ORC_BOUND_VIOLATED;
    for (int64_t i = 0; i < d && status == ORC_OK; ++i)
        for (int64_t q = cx.g1.ptr[i]; q < cx.g1.ptr[i + 1]; ++q)
            if (cx.g1.idx[q] / cx.k != i / cx.k) status = ORC_BOUND_VIOLATED;
    cx.inner = (unsigned char*)malloc((size_t)d);
    for (int64_t i = 0; i < d; ++i) { /* :114 */
        cx.inner[i] = 1;
        for (int64_t q = cx.g.ptr[i]; q < cx.g.ptr[i + 1]; ++q)
            if (cx.g.idx[q] / cx.k != i / cx.k) cx.inner[i] = 0;
    }
    cx.t_old = (double*)malloc((size_t)d * sizeof(double));
    cx.ba = (double*)malloc((size_t)d * sizeof(double));
    cx.bb = (double*)malloc((size_t)d * sizeof(double));
    cx.Q = (orc_pq**)malloc((size_t)K * sizeof(orc_pq*));
    uint64_t nd0 = 0;
    for (int64_t i = 0; i < d; ++i) cx.t[i] = cx.t_old[i] = t0;
    for (int ti = 0; ti < K; ++ti) cx.Q[ti] = orc_pq_new(cx.k);
    for (int64_t i = 0; i < d; ++i) zz_ab(p->bound_gamma, cx.zc.gmu_bound, i, x, th, c, &cx.ba[i], &cx.bb[i]); /* :132 */
    for (int64_t i = 0; i < d; ++i) /* :133-136 (global rng -> main stream here) */
        orc_pq_enqueue(cx.Q[i / cx.k], i % cx.k, orc_poisson_time(cx.ba[i], cx.bb[i], pdmp_u01(p->seed, PDMP_STREAM_MAIN, nd0++)));
    pthread_mutex_init(&cx.lm, NULL);
    pthread_cond_init(&cx.lcv, NULL);
    cx.w = (par_worker*)calloc((size_t)K, sizeof(par_worker));
    double* tpr = (double*)malloc((size_t)K * sizeof(double));   /* t′ per chunk, :109 */
    double* evtime = (double*)calloc((size_t)K, sizeof(double));
    int* perm = (int*)malloc((size_t)K * sizeof(int));
    int64_t* waitfor = (int64_t*)calloc((size_t)K, sizeof(int64_t)); /* 0 = none; i + 1 otherwise */
    int64_t acc = 0, num = 0, rounds = 0, spawns = 0;
    uint64_t nd_outer = 0;
    struct timespec ts0, ts1;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    if (status == ORC_OK) {
        for (int ti = 0; ti < K; ++ti) {
            tpr[ti] = t0;
            perm[ti] = ti;
            par_worker* w = &cx.w[ti];
            w->cx = &cx;
            w->ti = ti;
            pthread_mutex_init(&w->m, NULL);
            pthread_cond_init(&w->cv, NULL);
            orc_trace_init(&w->events);
            cx.active |= 1ull << ti; /* :156 */
        }
        for (int ti = 0; ti < K; ++ti) pthread_create(&cx.w[ti].th, NULL, par_worker_main, &cx.w[ti]); /* :157 */
        /* parallel_spdmp_outer!, :176-253 */
        double tmin = t0;
        while (tmin < T) {
            pthread_mutex_lock(&cx.lm); /* :186-193 */
            while (cx.active != 0) pthread_cond_wait(&cx.lcv, &cx.lm);
            pthread_mutex_unlock(&cx.lm);
            for (int ti = 0; ti < K; ++ti) { /* :194-203 */
                if (waitfor[ti] == 0) {
                    par_worker* w = &cx.w[ti];
                    spawns += 1;
                    for (int64_t e = 0; e < w->events.n; ++e)
                        trace_push(tr, w->events.ev[e].t, w->events.ev[e].i, w->events.ev[e].x, w->events.ev[e].theta);
                    w->events.n = 0;
                    evtime[ti] = w->ret_t;
                }
            }
            for (int a = 1; a < K; ++a) { /* sortperm!(perm, evtime, alg=InsertionSort), :204 (perm is reused) */
                const int v = perm[a];
                int b = a - 1;
                while (b >= 0 && evtime[perm[b]] > evtime[v]) {
                    perm[b + 1] = perm[b];
                    --b;
                }
                perm[b + 1] = v;
            }
            for (int a = 0; a < K; ++a) { /* :206-233 */
                const int ti = perm[a];
                par_worker* w = &cx.w[ti];
                const int64_t i = w->ret_i;
                const double tpi = w->ret_t;
                if (waitfor[ti] == 0) {
                    num += w->ret_num;
                    acc += w->ret_acc;
                    tpr[ti] = tpi;
                }
                waitfor[ti] = 0;
                for (int64_t q = cx.g.ptr[i]; q < cx.g.ptr[i + 1]; ++q) { /* :216-222 */
                    const int64_t j = cx.g.idx[q];
                    if (j == i) continue;
                    if (tpr[j / cx.k] < tpi) waitfor[ti] = i + 1;
                }
                if (waitfor[ti] != 0) continue;
                if (!(tpi < INFINITY)) continue; /* a parked chunk (error) */
                const int ok = par_innermost(&cx, p->seed, 15u, &nd_outer, i, tpi); /* :226 */
                if (ok < 0) {
                    cx.error = 1;
                } else if (ok) {
                    acc += 1;
                    trace_push(tr, cx.t[i], i, cx.x[i], cx.th[i]); /* :230 */
                }
            }
            tmin = tpr[0];
            for (int ti = 1; ti < K; ++ti) tmin = (tpr[ti] < tmin) ? tpr[ti] : tmin; /* :235 */
            if (cx.error) tmin = T; /* reference: the error propagates; stop everything */
            rounds += 1;
            for (int ti = 0; ti < K; ++ti) { /* :241-249 */
                if (waitfor[ti] == 0 || tmin >= T) {
                    par_worker* w = &cx.w[ti];
                    pthread_mutex_lock(&cx.lm);
                    cx.active |= 1ull << ti;
                    pthread_mutex_unlock(&cx.lm);
                    pthread_mutex_lock(&w->m);
                    w->wake = 1;
                    w->done = tmin >= T;
                    pthread_cond_signal(&w->cv);
                    pthread_mutex_unlock(&w->m);
                }
            }
        }
        for (int ti = 0; ti < K; ++ti) pthread_join(cx.w[ti].th, NULL); /* :163-165 */
        if (cx.error) status = ORC_BOUND_VIOLATED;
    }
    clock_gettime(CLOCK_MONOTONIC, &ts1);
    if (res) {
        res->num = num;
        res->nacc = acc;
        res->rounds = rounds;
        res->spawns = spawns;
        res->seconds = (double)(ts1.tv_sec - ts0.tv_sec) + 1e-9 * (double)(ts1.tv_nsec - ts0.tv_nsec);
        res->status = status;
    }
    for (int ti = 0; ti < K; ++ti) {
        if (cx.w[ti].cx) {
            orc_trace_free(&cx.w[ti].events);
            pthread_mutex_destroy(&cx.w[ti].m);
            pthread_cond_destroy(&cx.w[ti].cv);
        }
        orc_pq_free(cx.Q[ti]);
    }
    pthread_mutex_destroy(&cx.lm);
    pthread_cond_destroy(&cx.lcv);
    free(cx.w);
    free(tpr);
    free(evtime);
    free(perm);
    free(waitfor);
    free(cx.Q);
    free(cx.t_old);
    free(cx.ba);
    free(cx.bb);
    free(cx.inner);
    free(cx.zc.gmu_bound);
    free(cx.zc.gmu_target);
    graph_free(&cx.g);
    graph_free(&cx.g1);
    graph_free(&cx.g2);
    return status;
}

void orc_math_probe(uint64_t seed, int64_t n, double* out) {
    for (int64_t k = 0; k < n; ++k) {
        const double u = pdmp_u01(seed, 0u, (uint64_t)k);
        const double v = pdmp_u01(seed, 1u, (uint64_t)k);
        const double w = pdmp_u01(seed, 2u, (uint64_t)k);
        const double a = (u - 0.5) * 8.0;
        const double b = ((k % 7) == 0) ? 0.0 : (v - 0.5) * 4.0;
        out[0 * n + k] = u;
        out[1 * n + k] = pdmp_log(u);
        out[2 * n + k] = a / ((v - 0.5) * 4.0);
        out[3 * n + k] = sqrt(u * 1000.0 + v);
        out[4 * n + k] = orc_poisson_time(a, b, w);
        out[5 * n + k] = pdmp_randn(seed, 3u, (uint64_t)k);
        out[6 * n + k] = pdmp_exp((u - 0.5) * 60.0 + v);
        {
            double sn_, cs_;
            pdmp_sincos((w - 0.5) * 400.0, &sn_, &cs_);
            out[7 * n + k] = sn_ + 2.0 * cs_;
        }
    }
}
double orc_log(double x) { return pdmp_log(x); }
double orc_u01(uint64_t seed, uint32_t stream, uint64_t n) { return pdmp_u01(seed, stream, n); }
double orc_randn(uint64_t seed, uint32_t stream, uint64_t n) { return pdmp_randn(seed, stream, n); }
void orc_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    pdmp_u32x4 r = pdmp_philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
    for (int k = 0; k < 4; ++k) out[k] = r.v[k];
}
/* synthetic initial state of pdmp_ensemble_set_state_synthetic (include/pdmp_mi355.h) */
void orc_synthetic_state(uint64_t seed, int64_t d, double* x0, double* theta0) {
    for (int64_t i = 0; i < d; ++i) {
        x0[i] = pdmp_randn(seed, PDMP_STREAM_INIT, (uint64_t)i);
        theta0[i] = (pdmp_u01(seed, PDMP_STREAM_INIT, (uint64_t)(d + i)) < 0.5) ? -1.0 : 1.0;
    }
}
