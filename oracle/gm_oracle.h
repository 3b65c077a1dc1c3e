/*
 * gm_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * A plain-C restatement of the Green-Marl hot path: the three emitted kernels
 * (pagerank / hop_dist / triangle_counting) and the gm_graph CSR preparation
 * steps they depend on (RMAT generation, freeze, semi-sort, reverse edges).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (libgmx.so) never links or calls it.
 *
 * Parity status: PINNED.  Every function here is checked against the compiled
 * reference runtime (oracle/_ref, built from /root/reference sources in place
 * by oracle/Makefile) through oracle/make_golden.py, and against the committed
 * fixtures in tests/golden/ that script produced.  The emitted kernels
 * themselves (apps/output_cpp/generated/) cannot be produced in this image
 * (gm_comp needs flex), so the kernel restatements follow
 * the code-generator rules cited per function and are cross-checked through
 * independent reference code paths (gm_bfs_template levels, gm_graph::is_neighbor).
 *
 * All file:line citations are relative to /root/reference.
 */
#ifndef GM_ORACLE_H_
#define GM_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* node_t / edge_t = int32 (apps/output_cpp/gm_graph/inc/gm_graph_typedef.h:17-18) */
typedef int32_t gmo_node_t;
typedef int32_t gmo_edge_t;

/* ---- drand48 LCG (glibc: X' = 0x5DEECE66D*X + 0xB mod 2^48; srand48 sets
 *      X = seed<<16 | 0x330E).  Restated so the stream is libc independent. ---- */
typedef struct { uint64_t x; } gmo_rand48_t;
void   gmo_srand48(gmo_rand48_t* s, long seed);
double gmo_drand48(gmo_rand48_t* s);

/* ---- graph generation: create_RMAT_graph
 *      (apps/output_cpp/gm_graph/src/graph_gen.cc:159-287).
 * Writes the CSR exactly as the reference leaves it (rows filled back to front,
 * NOT yet semi-sorted).  begin has N+1 entries, node_idx has M entries.
 * attempts_out (optional) receives the number of edge attempts drawn
 * (M + rejected self loops). */
int gmo_create_rmat_graph(gmo_node_t N, gmo_edge_t M, long seed,
                          double a, double b, double c, int permute,
                          gmo_edge_t* begin, gmo_node_t* node_idx,
                          int64_t* attempts_out);

/* The raw (src,dst) list in generation order, after optional permutation
 * (graph_gen.cc:181-259). */
int gmo_rmat_edge_list(gmo_node_t N, gmo_edge_t M, long seed,
                       double a, double b, double c, int permute,
                       gmo_node_t* src, gmo_node_t* dst, int64_t* attempts_out);

/* CSR from an edge list with the reference's placement rule
 * (graph_gen.cc:262-280): row u is filled from its last slot backwards. */
void gmo_csr_from_edges(gmo_node_t N, gmo_edge_t M,
                        const gmo_node_t* src, const gmo_node_t* dst,
                        gmo_edge_t* begin, gmo_node_t* node_idx);

/* do_semi_sort (gm_graph.cc:380-423,468-503): sort every row ascending. */
void gmo_semi_sort(gmo_node_t N, const gmo_edge_t* begin, gmo_node_t* node_idx);

/* make_reverse_edges + do_semi_sort_reverse (gm_graph.cc:205-304,461-466).
 * r_begin has N+1 entries, r_node_idx has M; rows come out ascending. */
void gmo_make_reverse_edges(gmo_node_t N, gmo_edge_t M,
                            const gmo_edge_t* begin, const gmo_node_t* node_idx,
                            gmo_edge_t* r_begin, gmo_node_t* r_node_idx);

/* get_edge_idx_for_src_dest / is_neighbor (gm_graph.cc:589-633, 60-66;
 * shl_graph.cc:14-62).  Row of src must be sorted.  Returns edge idx or -1. */
gmo_edge_t gmo_get_edge_idx_for_src_dest(const gmo_edge_t* begin,
                                         const gmo_node_t* node_idx,
                                         gmo_node_t src, gmo_node_t to);

/* ---- emitted kernel: pagerank (apps/src/pagerank.gm:1-20; emission rules in
 *      SURVEY.md section 8 a-1).  fp64 end to end, Jacobi, no dangling mass.
 * nthreads<=0 -> omp default.  iters_out/diff_out optional. */
void gmo_pagerank(gmo_node_t N,
                  const gmo_edge_t* begin,
                  const gmo_edge_t* r_begin, const gmo_node_t* r_node_idx,
                  double e, double d, int32_t max_iter, double* rank,
                  int nthreads, int32_t* iters_out, double* diff_out);

/* ---- emitted kernels: avg_teen_cnt (apps/src/avg_teen_cnt.gm:1-13) and conduct (apps/src/conduct.gm:1-14),
 * count-reductions over in-/out-neighbours with an integer node property. */
float gmo_avg_teen_cnt(gmo_node_t N, const gmo_edge_t* r_begin, const gmo_node_t* r_node_idx,
                       const int32_t* age, int32_t* teen_cnt, int32_t K, int nthreads);
float gmo_conduct(gmo_node_t N, const gmo_edge_t* begin, const gmo_node_t* node_idx,
                  const int32_t* member, int32_t num, int nthreads);

/* ---- emitted kernel: sssp (apps/src/sssp.gm:1-30): hop_dist with the edge property
 * len[E] (indexed by forward edge slot) in place of 1.  dist[v] = shortest path length, INT_MAX unreachable. */
void gmo_sssp(gmo_node_t N,
              const gmo_edge_t* begin, const gmo_node_t* node_idx, const int32_t* len,
              gmo_node_t root, int32_t* dist, int nthreads, int32_t* rounds_out);

/* ---- emitted kernel: hop_dist (apps/src/hop_dist.gm:3-31).
 * Level-synchronous push over OUT edges, INT_MAX = unreached. */
void gmo_hop_dist(gmo_node_t N,
                  const gmo_edge_t* begin, const gmo_node_t* node_idx,
                  gmo_node_t root, int32_t* dist, int nthreads,
                  int32_t* levels_out);

/* Independent second statement of the same result: plain queue BFS.
 * Used only to cross-check gmo_hop_dist. */
/* gm_common_neighbor_iter(G, src, dest): number of items, the first cap in out; triangle counting written with it */
int64_t gmo_common_nbrs(const gmo_edge_t* begin, const gmo_node_t* node_idx,
                        gmo_node_t src, gmo_node_t dest, gmo_node_t* out, int64_t cap);
int64_t gmo_triangle_counting_cn(gmo_node_t N, const gmo_edge_t* begin, const gmo_node_t* node_idx, int nthreads);

/* comp_BC of apps/src/bc.gm (skip_root: upstream's `(v != s)` filters) */
void gmo_bc(gmo_node_t N, const gmo_edge_t* begin, const gmo_node_t* node_idx,
            const gmo_edge_t* r_begin, const gmo_node_t* r_node_idx,
            const gmo_node_t* seeds, int32_t nseeds, int skip_root, float* G_BC);

void gmo_bfs_queue(gmo_node_t N,
                   const gmo_edge_t* begin, const gmo_node_t* node_idx,
                   gmo_node_t root, int32_t* dist);

/* ---- emitted kernel: triangle_counting (apps/src/triangle_counting.gm:1-13;
 *      SURVEY.md 8 a-3).  Rows must be semi-sorted. */
int64_t gmo_triangle_counting(gmo_node_t N,
                              const gmo_edge_t* begin, const gmo_node_t* node_idx,
                              int nthreads);

/* Same count through a different route (per (v,u): merge of the N(v) tail with
 * the in-row of u).  Needs the reverse CSR.  Used to validate big inputs where
 * the emitted O(sum d^2 log d) form is too slow. */
int64_t gmo_triangle_counting_merge(gmo_node_t N,
                                    const gmo_edge_t* begin, const gmo_node_t* node_idx,
                                    const gmo_edge_t* r_begin, const gmo_node_t* r_node_idx,
                                    int nthreads);

/* Symmetrise + de-duplicate + drop self loops (measurement prep for the TC
 * config).  Returns new edge count; out arrays sized by caller
 * (out_node_idx capacity 2*M). */
gmo_edge_t gmo_symmetrize(gmo_node_t N, gmo_edge_t M,
                          const gmo_edge_t* begin, const gmo_node_t* node_idx,
                          gmo_edge_t* out_begin, gmo_node_t* out_node_idx);

/* binary .bin format, big-endian (gm_graph_binary_loader.cc:19-40,42-252). */
int gmo_store_binary(const char* path, gmo_node_t N, gmo_edge_t M,
                     const gmo_edge_t* begin, const gmo_node_t* node_idx);
/* two-call protocol: begin==NULL -> only N/M are returned. */
int gmo_load_binary(const char* path, gmo_node_t* N, gmo_edge_t* M,
                    gmo_edge_t* begin, gmo_node_t* node_idx);

int gmo_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
