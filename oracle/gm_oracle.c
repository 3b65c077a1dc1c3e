/*
 * gm_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See gm_oracle.h for scope, parity status and the rules on who may call this.
 * All file:line citations are relative to /root/reference.
 */
#include "gm_oracle.h"

#include <limits.h>
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <arpa/inet.h>
#include <omp.h>

int gmo_max_threads(void) { return omp_get_max_threads(); }

/* ------------------------------------------------------------------ */
/* drand48: the 48-bit LCG of SVID / glibc (stdlib/drand48-iter.c):    */
/*   X' = (0x5DEECE66D * X + 0xB) mod 2^48, result X'/2^48.           */
/* srand48(seed): X = (seed << 16) | 0x330E.                           */
/* ------------------------------------------------------------------ */
#define GMO_LCG_A 0x5DEECE66DULL
#define GMO_LCG_C 0xBULL
#define GMO_LCG_MASK ((1ULL << 48) - 1)

void gmo_srand48(gmo_rand48_t* s, long seed) {
    s->x = ((((uint64_t) seed) << 16) | 0x330EULL) & GMO_LCG_MASK;
}

double gmo_drand48(gmo_rand48_t* s) {
    s->x = (GMO_LCG_A * s->x + GMO_LCG_C) & GMO_LCG_MASK;
    return (double) s->x * (1.0 / 281474976710656.0); /* exact: 48 bits fit a double */
}

/* ------------------------------------------------------------------ */
/* create_RMAT_graph, edge generation part (graph_gen.cc:159-259)      */
/* ------------------------------------------------------------------ */
int gmo_rmat_edge_list(gmo_node_t N, gmo_edge_t M, long seed,
                       double a, double b, double c, int permute,
                       gmo_node_t* src, gmo_node_t* dest, int64_t* attempts_out) {
    double d;
    if (!(a + b + c < 1)) return -1;            /* graph_gen.cc:161 assert */
    d = 1 - (a + b + c);

    gmo_rand48_t R;
    gmo_srand48(&R, seed);                      /* graph_gen.cc:165 */

    gmo_node_t SCALE = (gmo_node_t) log2((double) N);   /* :179 */
    int64_t attempts = 0;

    for (gmo_edge_t i = 0; i < M; i++) {        /* :182 */
        gmo_node_t u = 1;
        gmo_node_t v = 1;
        gmo_node_t step = N / 2;
        double av = a, bv = b, cv = c, dv = d;

        attempts++;
        double p = gmo_drand48(&R);
        if (p < av) {
        } else if (p < (av + bv)) {
            v += step;
        } else if (p < (av + bv + cv)) {
            u += step;
        } else {
            v += step;
            u += step;
        }
        for (gmo_node_t j = 1; j < SCALE; j++) {
            step = step / 2;
            double var = 0.1;
            av *= 0.95 + var * gmo_drand48(&R);   /* :204-207, draw order a,b,c,d */
            bv *= 0.95 + var * gmo_drand48(&R);
            cv *= 0.95 + var * gmo_drand48(&R);
            dv *= 0.95 + var * gmo_drand48(&R);

            double S = av + bv + cv + dv;
            av = av / S;
            bv = bv / S;
            cv = cv / S;
            dv = dv / S;

            p = gmo_drand48(&R);
            if (p < av) {
            } else if (p < (av + bv)) {
                v += step;
            } else if (p < (av + bv + cv)) {
                u += step;
            } else {
                v += step;
                u += step;
            }
        }
        src[i] = u - 1;
        dest[i] = v - 1;
        if (src[i] == dest[i]) {                /* :231-235 reject self loops */
            i = i - 1;
            continue;
        }
    }

    if (permute) {                              /* :240-259 */
        gmo_node_t* P = (gmo_node_t*) malloc(sizeof(gmo_node_t) * (size_t) N);
        if (!P) return -2;
        for (gmo_node_t i = 0; i < N; i++) P[i] = i;
        for (gmo_node_t i = 0; i < N; i++) {
            gmo_node_t j = (gmo_node_t) (N * gmo_drand48(&R));
            gmo_node_t t = P[j];
            P[j] = P[i];
            P[i] = t;
        }
        for (gmo_edge_t i = 0; i < M; i++) {
            src[i] = P[src[i]];
            dest[i] = P[dest[i]];
        }
        free(P);
    }
    if (attempts_out) *attempts_out = attempts;
    return 0;
}

/* graph_gen.cc:262-280 */
void gmo_csr_from_edges(gmo_node_t N, gmo_edge_t M,
                        const gmo_node_t* src, const gmo_node_t* dest,
                        gmo_edge_t* begin, gmo_node_t* node_idx) {
    gmo_edge_t* degree = (gmo_edge_t*) calloc((size_t) N, sizeof(gmo_edge_t));
    for (gmo_edge_t i = 0; i < M; i++) degree[src[i]]++;
    begin[0] = 0;
    for (gmo_node_t i = 1; i <= N; i++) begin[i] = begin[i - 1] + degree[i - 1];
    for (gmo_edge_t i = 0; i < M; i++) {
        gmo_node_t u = src[i];
        gmo_node_t v = dest[i];
        gmo_edge_t pos = degree[u]--;
        node_idx[begin[u] + pos - 1] = v;
    }
    free(degree);
}

int gmo_create_rmat_graph(gmo_node_t N, gmo_edge_t M, long seed,
                          double a, double b, double c, int permute,
                          gmo_edge_t* begin, gmo_node_t* node_idx,
                          int64_t* attempts_out) {
    gmo_node_t* src = (gmo_node_t*) malloc(sizeof(gmo_node_t) * (size_t) (M > 0 ? M : 1));
    gmo_node_t* dst = (gmo_node_t*) malloc(sizeof(gmo_node_t) * (size_t) (M > 0 ? M : 1));
    if (!src || !dst) { free(src); free(dst); return -2; }
    int rc = gmo_rmat_edge_list(N, M, seed, a, b, c, permute, src, dst, attempts_out);
    if (rc == 0) gmo_csr_from_edges(N, M, src, dst, begin, node_idx);
    free(src);
    free(dst);
    return rc;
}

/* ------------------------------------------------------------------ */
/* do_semi_sort (gm_graph.cc:380-423,468-503): per-row ascending sort  */
/* of node_idx (the aux maps e_idx2idx/e_rev2idx are not on the path). */
/* ------------------------------------------------------------------ */
static int cmp_node(const void* x, const void* y) {
    gmo_node_t a = *(const gmo_node_t*) x, b = *(const gmo_node_t*) y;
    return (a > b) - (a < b);
}

void gmo_semi_sort(gmo_node_t N, const gmo_edge_t* begin, gmo_node_t* node_idx) {
#pragma omp parallel for schedule(dynamic, 4096)
    for (gmo_node_t i = 0; i < N; i++) {
        gmo_edge_t sz = begin[i + 1] - begin[i];
        if (sz > 1) qsort(node_idx + begin[i], (size_t) sz, sizeof(gmo_node_t), cmp_node);
    }
}

/* ------------------------------------------------------------------ */
/* make_reverse_edges (gm_graph.cc:205-304) followed by                */
/* do_semi_sort_reverse (:461-466).  The reference scatters with an    */
/* atomic fetch-add (row order nondeterministic) and then sorts rows;  */
/* a sequential counting sort over ascending sources yields the same   */
/* sorted rows directly.                                               */
/* ------------------------------------------------------------------ */
void gmo_make_reverse_edges(gmo_node_t N, gmo_edge_t M,
                            const gmo_edge_t* begin, const gmo_node_t* node_idx,
                            gmo_edge_t* r_begin, gmo_node_t* r_node_idx) {
    gmo_edge_t* cnt = (gmo_edge_t*) calloc((size_t) N + 1, sizeof(gmo_edge_t));
    for (gmo_edge_t e = 0; e < M; e++) cnt[node_idx[e]]++;
    gmo_edge_t sum = 0;
    for (gmo_node_t i = 0; i < N; i++) {
        r_begin[i] = sum;
        sum += cnt[i];
        cnt[i] = r_begin[i];
    }
    r_begin[N] = sum;
    for (gmo_node_t i = 0; i < N; i++)
        for (gmo_edge_t e = begin[i]; e < begin[i + 1]; e++)
            r_node_idx[cnt[node_idx[e]]++] = i;
    free(cnt);
}

/* gm_graph.cc:589-633 == shl_graph.cc:18-62 */
gmo_edge_t gmo_get_edge_idx_for_src_dest(const gmo_edge_t* begin,
                                         const gmo_node_t* node_idx,
                                         gmo_node_t src, gmo_node_t to) {
    gmo_edge_t begin_edge = begin[src];
    gmo_edge_t end_edge = begin[src + 1] - 1;
    if (begin_edge > end_edge) return -1;
    gmo_node_t left_node = node_idx[begin_edge];
    gmo_node_t right_node = node_idx[end_edge];
    if (to == left_node) return begin_edge;
    if (to == right_node) return end_edge;
    while (begin_edge < end_edge) {
        left_node = node_idx[begin_edge];
        right_node = node_idx[end_edge];
        if (to < left_node) return -1;
        if (to > right_node) return -1;
        gmo_edge_t mid_edge = (begin_edge + end_edge) / 2;
        gmo_node_t mid_node = node_idx[mid_edge];
        if (to == mid_node) return mid_edge;
        if (to < mid_node) {
            if (end_edge == mid_edge) return -1;
            end_edge = mid_edge;
        } else if (to > mid_node) {
            if (begin_edge == mid_edge) return -1;
            begin_edge = mid_edge;
        }
    }
    return -1;
}

/* ------------------------------------------------------------------ */
/* pagerank -- restated emission:                    */
/*   source apps/src/pagerank.gm:1-20                                  */
/*   in-neighbour loop over r_begin/r_node_idx                         */
/*       (src/backend_cpp/gm_cpp_gen_foreach.cc:149-190,268-312)       */
/*   OutDegree() -> (G.begin[w+1]-G.begin[w]) cast to double           */
/*       (src/backend_cpp/gm_cpplib_gen.cc:456-472,                    */
/*        src/frontend/gm_coercion.cc:6-37)                            */
/*   diff -> thread-private partial + ATOMIC_ADD<double>               */
/*       (src/backend_cpp/gm_cpp_opt_reduce_scalar.cc:141-257,         */
/*        apps/output_cpp/gm_graph/inc/gm_atomic_operations.h:4-14)    */
/*   deferred write -> rank_nxt + copy-back loop                       */
/*       (src/backend_cpp/gm_cpp_opt_defer.cc:127-224)                 */
/*   schedule(dynamic,128) (src/backend_cpp/gm_cpp_gen.cc:1874-1909)   */
/* ------------------------------------------------------------------ */
void gmo_pagerank(gmo_node_t numNodes,
                  const gmo_edge_t* begin,
                  const gmo_edge_t* r_begin, const gmo_node_t* r_node_idx,
                  double e, double d, int32_t max, double* G_pg_rank,
                  int nthreads, int32_t* iters_out, double* diff_out) {
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    double diff = 0.0;
    int32_t cnt = 0;
    double N = (double) numNodes;
    double* G_pg_rank_nxt = (double*) malloc(sizeof(double) * (size_t) (numNodes > 0 ? numNodes : 1));

#pragma omp parallel for num_threads(nthreads)
    for (gmo_node_t t0 = 0; t0 < numNodes; t0++)
        G_pg_rank[t0] = 1 / N;

    do {
        diff = ((float) (0.000000));
#pragma omp parallel num_threads(nthreads)
        {
            double diff_prv = ((float) (0.000000));
#pragma omp for nowait schedule(dynamic, 128)
            for (gmo_node_t t = 0; t < numNodes; t++) {
                double val;
                double __S1 = ((float) (0.000000));
                for (gmo_edge_t w_idx = r_begin[t]; w_idx < r_begin[t + 1]; w_idx++) {
                    gmo_node_t w = r_node_idx[w_idx];
                    __S1 = __S1 + G_pg_rank[w] / ((double) ((begin[w + 1] - begin[w])));
                }
                val = (1 - d) / N + d * __S1;
                diff_prv = diff_prv + fabs(val - G_pg_rank[t]);
                G_pg_rank_nxt[t] = val;
            }
            /* ATOMIC_ADD<double>(&diff, diff_prv) */
            if (diff_prv != 0) {
#pragma omp atomic
                diff += diff_prv;
            }
        }
#pragma omp parallel for num_threads(nthreads)
        for (gmo_node_t i3 = 0; i3 < numNodes; i3++)
            G_pg_rank[i3] = G_pg_rank_nxt[i3];
        cnt = cnt + 1;
    } while ((diff > e) && (cnt < max));

    free(G_pg_rank_nxt);
    if (iters_out) *iters_out = cnt;
    if (diff_out) *diff_out = diff;
}

/* ------------------------------------------------------------------ */
/* avg_teen_cnt -- restated emission:            */
/*   source apps/src/avg_teen_cnt.gm:1-13                              */
/*   Count(t: n.InNbrs)(filter) -> Sum of 1 under the filter, sequential*/
/*   inner loop over r_begin/r_node_idx (gm_cpp_gen_foreach.cc:161,286)*/
/*   Avg(n)(filter){Int} -> int32 sum __S, int64 count _cnt, then      */
/*   _avg = (0 == _cnt) ? 0 : __S / (double) _cnt                      */
/*       (src/opt/gm_syntax_sugar2.cc:264-296,421-436), (Float) cast.  */
/*   Scalar reductions: thread-private partials + ATOMIC_ADD           */
/*       (gm_cpp_opt_reduce_scalar.cc:141-257) -- integer, so exact.   */
/* ------------------------------------------------------------------ */
float gmo_avg_teen_cnt(gmo_node_t numNodes, const gmo_edge_t* r_begin, const gmo_node_t* r_node_idx,
                       const int32_t* G_age, int32_t* G_teen_cnt, int32_t K, int nthreads) {
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    float avg = 0;
    double _avg4 = 0;
    int64_t _cnt3 = 0;
    int32_t __S2 = 0;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 128)
    for (gmo_node_t n = 0; n < numNodes; n++) {
        int32_t __S1 = 0;
        for (gmo_edge_t t_idx = r_begin[n]; t_idx < r_begin[n + 1]; t_idx++) {
            gmo_node_t t = r_node_idx[t_idx];
            if ((G_age[t] >= 10) && (G_age[t] < 20)) __S1 = __S1 + 1;
        }
        G_teen_cnt[n] = __S1;
    }
#pragma omp parallel num_threads(nthreads)
    {
        int32_t __S2_prv = 0;
        int64_t _cnt3_prv = 0;
#pragma omp for nowait
        for (gmo_node_t n0 = 0; n0 < numNodes; n0++) {
            if (G_age[n0] > K) {
                __S2_prv = __S2_prv + G_teen_cnt[n0];
                _cnt3_prv = _cnt3_prv + 1;
            }
        }
        __atomic_fetch_add(&_cnt3, _cnt3_prv, __ATOMIC_SEQ_CST);
        __atomic_fetch_add(&__S2, __S2_prv, __ATOMIC_SEQ_CST);
    }
    _avg4 = (0 == _cnt3) ? ((float) (0.000000)) : (__S2 / ((double) _cnt3));
    avg = (float) _avg4;
    return avg;
}

/* ------------------------------------------------------------------ */
/* conduct -- restated emission:                 */
/*   source apps/src/conduct.gm:1-14.  Three Int (int32) sums:         */
/*   degrees of the members, of the non-members (u.Degree() ->         */
/*   begin[u+1]-begin[u], gm_cpplib_gen.cc:456-472) and the member ->  */
/*   non-member edges; Float m = min(Din, Dout) (coercion to float,    */
/*   src/frontend/gm_coercion.cc:6-37); m == 0 -> 0 or +INF = FLT_MAX  */
/*   (gm_cpp_gen.cc:1773-1799); else (float) Cross / m.                */
/* ------------------------------------------------------------------ */
float gmo_conduct(gmo_node_t numNodes, const gmo_edge_t* begin, const gmo_node_t* node_idx,
                  const int32_t* G_member, int32_t num, int nthreads) {
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    int32_t Din = 0, Dout = 0, Cross = 0;
#pragma omp parallel num_threads(nthreads)
    {
        int32_t din_prv = 0, dout_prv = 0, cross_prv = 0;
#pragma omp for nowait schedule(dynamic, 128)
        for (gmo_node_t u = 0; u < numNodes; u++) {
            if (G_member[u] == num) {
                din_prv = din_prv + (begin[u + 1] - begin[u]);
                int32_t __S3 = 0;
                for (gmo_edge_t j_idx = begin[u]; j_idx < begin[u + 1]; j_idx++) {
                    gmo_node_t j = node_idx[j_idx];
                    if (G_member[j] != num) __S3 = __S3 + 1;
                }
                cross_prv = cross_prv + __S3;
            } else dout_prv = dout_prv + (begin[u + 1] - begin[u]);
        }
        __atomic_fetch_add(&Din, din_prv, __ATOMIC_SEQ_CST);
        __atomic_fetch_add(&Dout, dout_prv, __ATOMIC_SEQ_CST);
        __atomic_fetch_add(&Cross, cross_prv, __ATOMIC_SEQ_CST);
    }
    float m = (float) ((Din < Dout) ? Din : Dout);
    if (m == 0) return (Cross == 0) ? ((float) (0.000000)) : FLT_MAX;
    return Cross / m;
}

/* ------------------------------------------------------------------ */
/* sssp -- restated emission:                    */
/*   source apps/src/sssp.gm:1-30 -- hop_dist's loop with an edge      */
/*   property: `Edge e = s.ToEdge()` is the iterator of the out-       */
/*   neighbour loop (src/backend_cpp/gm_cpp_gen_foreach.cc:161,286),   */
/*   so e.len is G_len[s_idx]; everything else as gmo_hop_dist below.  */
/*   Result: dist[v] = length of a shortest path root -> v over out-   */
/*   edges, INT_MAX if unreachable (unique, schedule independent).     */
/* ------------------------------------------------------------------ */
void gmo_sssp(gmo_node_t numNodes,
              const gmo_edge_t* begin, const gmo_node_t* node_idx, const int32_t* G_len,
              gmo_node_t root, int32_t* G_dist, int nthreads, int32_t* rounds_out) {
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    size_t n = (size_t) (numNodes > 0 ? numNodes : 1);
    uint8_t* G_updated = (uint8_t*) malloc(n);
    uint8_t* G_updated_nxt = (uint8_t*) malloc(n);
    int32_t* G_dist_nxt = (int32_t*) malloc(sizeof(int32_t) * n);
    int fin = 0;
    int32_t rounds = 0;

#pragma omp parallel for num_threads(nthreads)
    for (gmo_node_t t0 = 0; t0 < numNodes; t0++) {
        G_dist[t0] = (t0 == root) ? 0 : INT_MAX;
        G_updated[t0] = (t0 == root) ? 1 : 0;
        G_dist_nxt[t0] = G_dist[t0];
        G_updated_nxt[t0] = G_updated[t0];
    }

    while (!fin) {
        int __E8 = 0;
        fin = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 128)
        for (gmo_node_t nn = 0; nn < numNodes; nn++) {
            if (G_updated[nn]) {
                for (gmo_edge_t s_idx = begin[nn]; s_idx < begin[nn + 1]; s_idx++) {
                    gmo_node_t s = node_idx[s_idx];
                    gmo_edge_t e = s_idx;
                    int32_t nv = G_dist[nn] + G_len[e];
                    int32_t cur = __atomic_load_n(&G_dist_nxt[s], __ATOMIC_RELAXED);
                    while (cur > nv) {
                        if (__atomic_compare_exchange_n(&G_dist_nxt[s], &cur, nv, 0,
                                                        __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
                            G_updated_nxt[s] = 1;
                            break;
                        }
                    }
                }
            }
        }
#pragma omp parallel num_threads(nthreads)
        {
            int __E8_prv = 0;
#pragma omp for nowait
            for (gmo_node_t t4 = 0; t4 < numNodes; t4++) {
                G_dist[t4] = G_dist_nxt[t4];
                G_updated[t4] = G_updated_nxt[t4];
                G_updated_nxt[t4] = 0;
                __E8_prv = __E8_prv || G_updated[t4];
            }
            if (__E8_prv) {
#pragma omp atomic write
                __E8 = 1;
            }
        }
        fin = !__E8;
        rounds++;
    }
    if (rounds_out) *rounds_out = rounds;
    free(G_updated);
    free(G_updated_nxt);
    free(G_dist_nxt);
}

/* ------------------------------------------------------------------ */
/* hop_dist -- restated emission:                    */
/*   source apps/src/hop_dist.gm:3-31                                  */
/*   merged init loop (src/opt/gm_merge_loops.cc:228)                  */
/*   out-neighbour loop over begin/node_idx                            */
/*       (src/backend_cpp/gm_cpp_gen_foreach.cc:161,286)               */
/*   <dist_nxt;updated_nxt> min= <dist+1;True> -> test, lock, test     */
/*       (src/backend_cpp/gm_cpp_gen.cc:1563-1741;                     */
/*        apps/output_cpp/gm_graph/src/gm_lock.cc:48-58)               */
/*   +INF -> INT_MAX (gm_cpp_gen.cc:1774-1800)                         */
/*   Exist -> OR reduction (src/opt/gm_syntax_sugar2.cc:256-258)       */
/* The per-node spinlock is restated as an atomic min (same final      */
/* dist_nxt, and updated_nxt is set exactly when the min lowered it).  */
/* ------------------------------------------------------------------ */
void gmo_hop_dist(gmo_node_t numNodes,
                  const gmo_edge_t* begin, const gmo_node_t* node_idx,
                  gmo_node_t root, int32_t* G_dist, int nthreads,
                  int32_t* levels_out) {
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    size_t n = (size_t) (numNodes > 0 ? numNodes : 1);
    uint8_t* G_updated = (uint8_t*) malloc(n);
    uint8_t* G_updated_nxt = (uint8_t*) malloc(n);
    int32_t* G_dist_nxt = (int32_t*) malloc(sizeof(int32_t) * n);
    int fin = 0;
    int32_t levels = 0;

#pragma omp parallel for num_threads(nthreads)
    for (gmo_node_t t0 = 0; t0 < numNodes; t0++) {
        G_dist[t0] = (t0 == root) ? 0 : INT_MAX;
        G_updated[t0] = (t0 == root) ? 1 : 0;
        G_dist_nxt[t0] = G_dist[t0];
        G_updated_nxt[t0] = G_updated[t0];
    }

    while (!fin) {
        int __E8 = 0;
        fin = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 128)
        for (gmo_node_t nn = 0; nn < numNodes; nn++) {
            if (G_updated[nn]) {
                for (gmo_edge_t s_idx = begin[nn]; s_idx < begin[nn + 1]; s_idx++) {
                    gmo_node_t s = node_idx[s_idx];
                    int32_t nv = G_dist[nn] + 1;
                    int32_t cur = __atomic_load_n(&G_dist_nxt[s], __ATOMIC_RELAXED);
                    while (cur > nv) {
                        if (__atomic_compare_exchange_n(&G_dist_nxt[s], &cur, nv, 0,
                                                        __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
                            G_updated_nxt[s] = 1;
                            break;
                        }
                    }
                }
            }
        }
#pragma omp parallel num_threads(nthreads)
        {
            int __E8_prv = 0;
#pragma omp for nowait
            for (gmo_node_t t4 = 0; t4 < numNodes; t4++) {
                G_dist[t4] = G_dist_nxt[t4];
                G_updated[t4] = G_updated_nxt[t4];
                G_updated_nxt[t4] = 0;
                __E8_prv = __E8_prv || G_updated[t4];
            }
            if (__E8_prv) {
#pragma omp atomic write
                __E8 = 1;
            }
        }
        fin = !__E8;
        levels++;
    }
    free(G_updated);
    free(G_updated_nxt);
    free(G_dist_nxt);
    if (levels_out) *levels_out = levels;
}

/* ------------------------------------------------------------------ */
/* gm_common_neighbor_iter (gm_common_neighbor_iter.cc:3-44): the       */
/* iterator behind `Foreach(u: s.CommonNbrs(d))`                        */
/* (src/backend_cpp/gm_cpp_opt_common_nbr.cc:11-26 rewrites             */
/* `Foreach(t: x.Nbrs){ If (t.IsNbrFrom(y)) ..}` into it).  reset():    */
/* both cursors at the row starts, finished if either row is empty;     */
/* get_next(): take the next slot of s's row, advance d's cursor while  */
/* it points below the value, report the value if d's cursor stops ON   */
/* it; once either cursor runs off its row the iterator is finished     */
/* (after the current value has been judged).  Rows are semi-sorted.    */
/* Returns how many items it yields; the first `cap` go to out.         */
/* ------------------------------------------------------------------ */
int64_t gmo_common_nbrs(const gmo_edge_t* begin, const gmo_node_t* node_idx,
                        gmo_node_t src, gmo_node_t dest, gmo_node_t* out, int64_t cap) {
    gmo_edge_t src_idx = begin[src], src_end = begin[src + 1];
    gmo_edge_t dest_idx = begin[dest], dest_end = begin[dest + 1];
    int finished = (src_idx == src_end) || (dest_idx == dest_end);
    int64_t n = 0;
    while (!finished) {                                  /* get_next() */
        gmo_node_t t = node_idx[src_idx];
        src_idx++;
        if (src_idx == src_end) finished = 1;
        int common = 0;                                  /* check_common(t) */
        for (;;) {
            gmo_node_t r = node_idx[dest_idx];
            if (r == t) { common = 1; break; }
            if (r > t) break;
            dest_idx++;
            if (dest_idx == dest_end) { finished = 1; break; }
        }
        if (common) {
            if (n < cap) out[n] = t;
            n++;
        }
    }
    return n;
}

/* Triangle counting written with the iterator:                         */
/*   Foreach(v: G.Nodes) Foreach(u: v.Nbrs)(u > v)                      */
/*     Foreach(w: v.CommonNbrs(u))(w > u) T += 1;                       */
/* emitted as  gm_common_neighbor_iter w_I(G, v, u);                    */
/*   for (node_t w = w_I.get_next(); w != gm_graph::NIL_NODE; w = ...)  */
int64_t gmo_triangle_counting_cn(gmo_node_t N, const gmo_edge_t* begin, const gmo_node_t* node_idx, int nthreads) {
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    int64_t T = 0;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 128) reduction(+ : T)
    for (gmo_node_t v = 0; v < N; v++) {
        gmo_edge_t cap = begin[v + 1] - begin[v];
        gmo_node_t* buf = (gmo_node_t*) malloc(sizeof(gmo_node_t) * (size_t) (cap > 0 ? cap : 1));
        for (gmo_edge_t u_idx = begin[v]; u_idx < begin[v + 1]; u_idx++) {
            gmo_node_t u = node_idx[u_idx];
            if (u > v) {
                int64_t n = gmo_common_nbrs(begin, node_idx, v, u, buf, cap);
                for (int64_t i = 0; i < n; i++)
                    if (buf[i] > u) T = T + 1;
            }
        }
        free(buf);
    }
    return T;
}

/* ------------------------------------------------------------------ */
/* comp_BC (betweenness centrality estimate over a seed sequence)      */
/*   source apps/src/bc.gm:4-31.  `InBFS(v: G.Nodes From s)` becomes a   */
/*   subclass of gm_bfs_template<short, omp, false, false, true>        */
/*   (src/backend_cpp/gm_cpp_gen_bfs.cc:88-221: level_t = short,        */
/*   save_child because DownNbrs is used) whose visit_fw / visit_rv     */
/*   hold the loop bodies, driven by prepare(s) / do_bfs_forward() /    */
/*   do_bfs_reverse() (:256-265; gm_bfs_template.h:44-312).             */
/*   `v.UpNbrs`  -> loop over the REVERSE row of v with                 */
/*        if (get_level(w) != (get_curr_level() - 1)) continue;         */
/*   `v.DownNbrs`-> loop over the forward row of v with                 */
/*        if (!is_down_edge(w_idx)) continue;                           */
/*   (gm_cpp_gen_foreach.cc:161-176); an edge v->u is a down edge iff   */
/*   level(u) == level(v) + 1 (gm_bfs_template.h:581-622).  Both loops  */
/*   walk edge SLOTS, so repeated edges count repeatedly.  Sum over     */
/*   Float -> a float accumulator, terms added in row order             */
/*   (src/opt/gm_syntax_sugar2.cc:182-330); `v.BC += v.delta @ s`       */
/*   is applied to v itself.  visit_fw / visit_rv run for EVERY vertex  */
/*   of a level, the root included: this fork's bc.gm has no            */
/*   `(v != s)` filter (upstream Green-Marl's has), so the forward      */
/*   visit of s overwrites s.sigma = 1 with the empty sum 0, every      */
/*   sigma stays 0 and every reached vertex with a BFS child gets       */
/*   0/0 = NaN.  skip_root = 1 gives the upstream form (both filters).  */
/*   Temporaries sigma / delta come from gm_rt_allocate_float           */
/*   (uninitialised; delta is only read where it was written).          */
/* ------------------------------------------------------------------ */
void gmo_bc(gmo_node_t N, const gmo_edge_t* begin, const gmo_node_t* node_idx,
            const gmo_edge_t* r_begin, const gmo_node_t* r_node_idx,
            const gmo_node_t* seeds, int32_t nseeds, int skip_root, float* G_BC) {
    size_t n = (size_t) (N > 0 ? N : 1);
    float* G_sigma = (float*) malloc(sizeof(float) * n);
    float* G_delta = (float*) malloc(sizeof(float) * n);
    int32_t* level = (int32_t*) malloc(sizeof(int32_t) * n);
    gmo_node_t* order = (gmo_node_t*) malloc(sizeof(gmo_node_t) * n);   /* BFS order: levels are contiguous */
    for (gmo_node_t t0 = 0; t0 < N; t0++) G_BC[t0] = 0;
    for (int32_t si = 0; si < nseeds; si++) {
        const gmo_node_t s = seeds[si];
        for (gmo_node_t t1 = 0; t1 < N; t1++) G_sigma[t1] = 0;
        G_sigma[s] = 1;
        /* levels: the template's traversal order inside a level does not matter for a per-vertex visit */
        for (gmo_node_t i = 0; i < N; i++) level[i] = -2;                /* __INVALID_LEVEL, gm_bfs_template.h:725 */
        size_t head = 0, tail = 0;
        level[s] = 0;
        order[tail++] = s;
        while (head < tail) {
            gmo_node_t v = order[head++];
            for (gmo_edge_t e = begin[v]; e < begin[v + 1]; e++) {
                gmo_node_t u = node_idx[e];
                if (level[u] == -2) { level[u] = level[v] + 1; order[tail++] = u; }
            }
        }
        /* forward: visit_fw(v) for every v, level by level */
        for (size_t i = 0; i < tail; i++) {
            gmo_node_t v = order[i];
            if (skip_root && v == s) continue;
            float __S1 = 0;
            for (gmo_edge_t w_idx = r_begin[v]; w_idx < r_begin[v + 1]; w_idx++) {
                gmo_node_t w = r_node_idx[w_idx];
                if (level[w] != (level[v] - 1)) continue;
                __S1 = __S1 + G_sigma[w];
            }
            G_sigma[v] = __S1;
        }
        /* reverse: visit_rv(v), deepest level first */
        for (size_t i = tail; i-- > 0;) {
            gmo_node_t v = order[i];
            if (skip_root && v == s) continue;
            float __S2 = 0;
            for (gmo_edge_t w_idx = begin[v]; w_idx < begin[v + 1]; w_idx++) {
                gmo_node_t w = node_idx[w_idx];
                if (level[w] != level[v] + 1) continue;                /* !is_down_edge(w_idx) */
                __S2 = __S2 + G_sigma[v] / G_sigma[w] * (1 + G_delta[w]);
            }
            G_delta[v] = __S2;
            G_BC[v] = G_BC[v] + G_delta[v];
        }
    }
    free(G_sigma); free(G_delta); free(level); free(order);
}

void gmo_bfs_queue(gmo_node_t N,
                   const gmo_edge_t* begin, const gmo_node_t* node_idx,
                   gmo_node_t root, int32_t* dist) {
    gmo_node_t* q = (gmo_node_t*) malloc(sizeof(gmo_node_t) * (size_t) (N > 0 ? N : 1));
    for (gmo_node_t i = 0; i < N; i++) dist[i] = INT_MAX;
    if (root < 0 || root >= N) { free(q); return; }
    size_t head = 0, tail = 0;
    dist[root] = 0;
    q[tail++] = root;
    while (head < tail) {
        gmo_node_t v = q[head++];
        for (gmo_edge_t e = begin[v]; e < begin[v + 1]; e++) {
            gmo_node_t s = node_idx[e];
            if (dist[s] == INT_MAX) {
                dist[s] = dist[v] + 1;
                q[tail++] = s;
            }
        }
    }
    free(q);
}

/* ------------------------------------------------------------------ */
/* triangle_counting -- restated emission:           */
/*   source apps/src/triangle_counting.gm:1-13                         */
/*   HasEdgeTo -> is_neighbor(w,u) binary search on the forward row of */
/*       w (src/backend_cpp/gm_cpplib_gen.cc:487-499;                  */
/*        apps/output_cpp/gm_graph/src/shl_graph.cc:14-62)             */
/*   T += 1 -> T_prv + ATOMIC_ADD<int64_t>                             */
/* ------------------------------------------------------------------ */
int64_t gmo_triangle_counting(gmo_node_t numNodes,
                              const gmo_edge_t* begin, const gmo_node_t* node_idx,
                              int nthreads) {
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    int64_t T = 0;
#pragma omp parallel num_threads(nthreads)
    {
        int64_t T_prv = 0;
#pragma omp for nowait schedule(dynamic, 128)
        for (gmo_node_t v = 0; v < numNodes; v++) {
            for (gmo_edge_t u_idx = begin[v]; u_idx < begin[v + 1]; u_idx++) {
                gmo_node_t u = node_idx[u_idx];
                if (u > v) {
                    for (gmo_edge_t w_idx = begin[v]; w_idx < begin[v + 1]; w_idx++) {
                        gmo_node_t w = node_idx[w_idx];
                        if (w > u) {
                            if (gmo_get_edge_idx_for_src_dest(begin, node_idx, w, u) != -1)
                                T_prv = T_prv + 1;
                        }
                    }
                }
            }
        }
#pragma omp atomic
        T += T_prv;
    }
    return T;
}

/* lower bound of x in sorted a[lo,hi) */
static gmo_edge_t lb(const gmo_node_t* a, gmo_edge_t lo, gmo_edge_t hi, gmo_node_t x) {
    while (lo < hi) {
        gmo_edge_t mid = lo + (hi - lo) / 2;
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

int64_t gmo_triangle_counting_merge(gmo_node_t N,
                                    const gmo_edge_t* begin, const gmo_node_t* node_idx,
                                    const gmo_edge_t* r_begin, const gmo_node_t* r_node_idx,
                                    int nthreads) {
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    int64_t T = 0;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 64) reduction(+:T)
    for (gmo_node_t v = 0; v < N; v++) {
        gmo_edge_t b = begin[v], e = begin[v + 1];
        for (gmo_edge_t i = b; i < e; i++) {
            gmo_node_t u = node_idx[i];
            if (u <= v) continue;
            /* tail: slots of row v with value > u (multiset) */
            gmo_edge_t tb = lb(node_idx, i, e, u + 1), te = e;
            gmo_edge_t rb = r_begin[u], re = r_begin[u + 1];
            /* in-row of u restricted to values > u */
            rb = lb(r_node_idx, rb, re, u + 1);
            gmo_edge_t tl = te - tb, rl = re - rb;
            if (tl == 0 || rl == 0) continue;
            if (tl * 8 < rl) {
                for (gmo_edge_t j = tb; j < te; j++) {
                    gmo_node_t w = node_idx[j];
                    gmo_edge_t p = lb(r_node_idx, rb, re, w);
                    if (p < re && r_node_idx[p] == w) T++;
                }
            } else {
                gmo_edge_t j = tb, k = rb;
                while (j < te && k < re) {
                    gmo_node_t w = node_idx[j], x = r_node_idx[k];
                    if (w < x) j++;
                    else if (w > x) k++;
                    else { T++; j++; }   /* tail counts with multiplicity, in-row is a set */
                }
            }
        }
    }
    return T;
}

gmo_edge_t gmo_symmetrize(gmo_node_t N, gmo_edge_t M,
                          const gmo_edge_t* begin, const gmo_node_t* node_idx,
                          gmo_edge_t* out_begin, gmo_node_t* out_node_idx) {
    int64_t* off = (int64_t*) calloc((size_t) N + 1, sizeof(int64_t));
    for (gmo_node_t u = 0; u < N; u++)
        for (gmo_edge_t e = begin[u]; e < begin[u + 1]; e++) {
            gmo_node_t v = node_idx[e];
            if (v == u) continue;
            off[u + 1]++;
            off[v + 1]++;
        }
    for (gmo_node_t i = 0; i < N; i++) off[i + 1] += off[i];
    int64_t tot = off[N];
    gmo_node_t* tmp = (gmo_node_t*) malloc(sizeof(gmo_node_t) * (size_t) (tot > 0 ? tot : 1));
    int64_t* pos = (int64_t*) malloc(sizeof(int64_t) * ((size_t) N + 1));
    memcpy(pos, off, sizeof(int64_t) * ((size_t) N + 1));
    for (gmo_node_t u = 0; u < N; u++)
        for (gmo_edge_t e = begin[u]; e < begin[u + 1]; e++) {
            gmo_node_t v = node_idx[e];
            if (v == u) continue;
            tmp[pos[u]++] = v;
            tmp[pos[v]++] = u;
        }
    gmo_edge_t outM = 0;
    for (gmo_node_t u = 0; u < N; u++) {
        int64_t b = off[u], e = off[u + 1];
        if (e - b > 1) qsort(tmp + b, (size_t) (e - b), sizeof(gmo_node_t), cmp_node);
        out_begin[u] = outM;
        for (int64_t k = b; k < e; k++)
            if (k == b || tmp[k] != tmp[k - 1]) out_node_idx[outM++] = tmp[k];
    }
    out_begin[N] = outM;
    (void) M;
    free(tmp);
    free(pos);
    free(off);
    return outM;
}

/* gm_graph_binary_loader.cc:207-252 (store), :42-205 (load); all big-endian */
int gmo_store_binary(const char* path, gmo_node_t N, gmo_edge_t M,
                     const gmo_edge_t* begin, const gmo_node_t* node_idx) {
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    uint32_t hdr[5] = { htonl(0x03939999u), htonl(4u), htonl(4u), htonl((uint32_t) N), htonl((uint32_t) M) };
    fwrite(hdr, 4, 5, f);
    for (gmo_node_t i = 0; i < N + 1; i++) { uint32_t x = htonl((uint32_t) begin[i]); fwrite(&x, 4, 1, f); }
    for (gmo_edge_t i = 0; i < M; i++) { uint32_t x = htonl((uint32_t) node_idx[i]); fwrite(&x, 4, 1, f); }
    fclose(f);
    return 0;
}

int gmo_load_binary(const char* path, gmo_node_t* N, gmo_edge_t* M,
                    gmo_edge_t* begin, gmo_node_t* node_idx) {
    FILE* f = fopen(path, "rb");
    if (!f) return -1;
    uint32_t hdr[5];
    if (fread(hdr, 4, 5, f) != 5) { fclose(f); return -2; }
    if (ntohl(hdr[0]) != 0x03939999u || ntohl(hdr[1]) != 4 || ntohl(hdr[2]) != 4) { fclose(f); return -3; }
    *N = (gmo_node_t) ntohl(hdr[3]);
    *M = (gmo_edge_t) ntohl(hdr[4]);
    if (begin) {
        for (gmo_node_t i = 0; i < *N + 1; i++) { uint32_t x; if (fread(&x, 4, 1, f) != 1) { fclose(f); return -4; } begin[i] = (gmo_edge_t) ntohl(x); }
        for (gmo_edge_t i = 0; i < *M; i++) { uint32_t x; if (fread(&x, 4, 1, f) != 1) { fclose(f); return -4; } node_idx[i] = (gmo_node_t) ntohl(x); }
    }
    fclose(f);
    return 0;
}
