/*
 * ref_harness.cc -- TEST INFRASTRUCTURE (never shipped, never on the GPU box).
 *
 * A thin extern "C" shim over the REAL reference runtime (libgmgraph objects
 * compiled from /root/reference/apps/output_cpp/gm_graph/src/*.cc in place by
 * oracle/Makefile into oracle/_ref/).  oracle/make_golden.py loads the
 * resulting oracle/_ref/libgmref.so to (1) pin oracle/gm_oracle.c and
 * (2) generate the committed fixtures under tests/golden/.
 *
 * The three emitted kernels are not in the reference tree (generated code is
 * git-ignored and gm_comp cannot be built here: no flex), so the kernel bodies
 * below are the plain emission restated against the reference's own gm_graph /
 * gm_rt_* / ATOMIC_* / spinlock primitives, i.e. they compile and run against
 * the genuine runtime the generated code would link with.  gm_bfs_template and
 * gm_graph::is_neighbor are pure reference code and serve as independent
 * checks of hop_dist and triangle_counting.
 */
#include <limits.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include "gm.h"          // reference umbrella header (apps/output_cpp/gm_graph/inc/gm.h)
#include "gm_common_neighbor_iter.h"
#include "gm_rand.h"
#include "graph_gen.h"   // create_RMAT_graph

extern "C" {

/* glibc drand48 stream, to pin gmo_drand48 */
void ref_drand48_stream(long seed, int n, double* out) {
    srand48(seed);
    for (int i = 0; i < n; i++) out[i] = drand48();
}

/* raw create_RMAT_graph CSR (before any sort) */
int ref_rmat_csr(int32_t N, int32_t M, long seed, double a, double b, double c, int permute,
                 int32_t* begin, int32_t* node_idx) {
    gm_graph* g = create_RMAT_graph(N, M, seed, a, b, c, permute != 0);
    memcpy(begin, g->begin, sizeof(int32_t) * ((size_t) N + 1));
    memcpy(node_idx, g->node_idx, sizeof(int32_t) * (size_t) M);
    delete g;
    return 0;
}

static gm_graph* make_graph(int32_t N, int32_t M, const int32_t* begin, const int32_t* node_idx) {
    gm_graph* g = new gm_graph();
    g->prepare_external_creation(N, M);
    memcpy(g->begin, begin, sizeof(int32_t) * ((size_t) N + 1));
    memcpy(g->node_idx, node_idx, sizeof(int32_t) * (size_t) M);
    return g;
}

/* what load_binary does after reading: do_semi_sort + make_reverse_edges
 * (gm_graph_binary_loader.cc:191-197) */
int ref_prepare(int32_t N, int32_t M, const int32_t* begin, const int32_t* node_idx,
                int32_t* s_node_idx, int32_t* r_begin, int32_t* r_node_idx) {
    gm_graph* g = make_graph(N, M, begin, node_idx);
    g->do_semi_sort();
    g->make_reverse_edges();
    memcpy(s_node_idx, g->node_idx, sizeof(int32_t) * (size_t) M);
    memcpy(r_begin, g->r_begin, sizeof(int32_t) * ((size_t) N + 1));
    memcpy(r_node_idx, g->r_node_idx, sizeof(int32_t) * (size_t) M);
    delete g;
    return 0;
}

/* store_binary / load_binary round trip through the reference code */
int ref_store_binary(const char* path, int32_t N, int32_t M, const int32_t* begin, const int32_t* node_idx) {
    gm_graph* g = make_graph(N, M, begin, node_idx);
    bool ok = g->store_binary((char*) path);
    delete g;
    return ok ? 0 : -1;
}

int ref_load_binary(const char* path, int32_t* N, int32_t* M,
                    int32_t* begin, int32_t* node_idx, int32_t* r_begin, int32_t* r_node_idx) {
    gm_graph g;
    if (!g.load_binary((char*) path)) return -1;
    *N = g.num_nodes();
    *M = g.num_edges();
    if (begin) {
        memcpy(begin, g.begin, sizeof(int32_t) * ((size_t) *N + 1));
        memcpy(node_idx, g.node_idx, sizeof(int32_t) * (size_t) *M);
        memcpy(r_begin, g.r_begin, sizeof(int32_t) * ((size_t) *N + 1));
        memcpy(r_node_idx, g.r_node_idx, sizeof(int32_t) * (size_t) *M);
    }
    return 0;
}

/* The reference's adjacency-list loader (gm_graph_adj_loader.cc:112-219: sparse vertex keys are renumbered in order
 * of appearance, destination-only vertices appended, rows semi-sorted, reverse edges built), one double vertex
 * value and one double edge value per entry -- the layout of its own sample files
 * (apps/output_cpp/gm_graph/test/very_small_sample.adj, test/cpp_be/test.adj). */
int ref_load_adj(const char* path, int32_t* N, int32_t* M, int32_t* begin, int32_t* node_idx, int32_t cap_n, int32_t cap_m) {
    gm_graph g;
    std::vector<VALUE_TYPE> vs, es;
    vs.push_back(GMTYPE_DOUBLE);
    es.push_back(GMTYPE_DOUBLE);
    std::vector<void*> vp, ep;
    if (!g.load_adjacency_list(path, vs, es, vp, ep, " \t", false)) return -1;
    *N = g.num_nodes();
    *M = g.num_edges();
    if (*N > cap_n || *M > cap_m) return -2;
    memcpy(begin, g.begin, sizeof(int32_t) * ((size_t) *N + 1));
    memcpy(node_idx, g.node_idx, sizeof(int32_t) * (size_t) *M);
    return 0;
}

/* ---- pagerank: plain emission against the reference runtime ---- */
int ref_pagerank(int32_t N_, int32_t M, const int32_t* begin, const int32_t* node_idx,
                 double e, double d, int32_t max, double* G_pg_rank, int nthreads, int32_t* iters) {
    gm_graph* gp = make_graph(N_, M, begin, node_idx);
    gm_graph& G = *gp;
    G.do_semi_sort();
    gm_rt_set_num_threads(nthreads);

    //Initializations
    gm_rt_initialize();
    G.freeze();
    G.make_reverse_edges();

    double diff = 0.0;
    int32_t cnt = 0;
    double N = 0.0;
    double* G_pg_rank_nxt = gm_rt_allocate_double(G.num_nodes(), gm_rt_thread_id());

    cnt = 0;
    N = (double) (G.num_nodes());

    #pragma omp parallel for
    for (node_t t0 = 0; t0 < G.num_nodes(); t0++)
        G_pg_rank[t0] = 1 / N;

    do {
        diff = ((float) (0.000000));
        #pragma omp parallel
        {
            double diff_prv = 0.0;
            diff_prv = ((float) (0.000000));

            #pragma omp for nowait schedule(dynamic,128)
            for (node_t t = 0; t < G.num_nodes(); t++) {
                double val = 0.0;
                double __S1 = 0.0;
                __S1 = ((float) (0.000000));
                for (edge_t w_idx = G.r_begin[t]; w_idx < G.r_begin[t + 1]; w_idx++) {
                    node_t w = G.r_node_idx[w_idx];
                    __S1 = __S1 + G_pg_rank[w] / ((double) ((G.begin[w + 1] - G.begin[w])));
                }
                val = (1 - d) / N + d * __S1;
                diff_prv = diff_prv + std::abs((val - G_pg_rank[t]));
                G_pg_rank_nxt[t] = val;
            }
            ATOMIC_ADD<double>(&diff, diff_prv);
        }
        #pragma omp parallel for
        for (node_t i3 = 0; i3 < G.num_nodes(); i3++)
            G_pg_rank[i3] = G_pg_rank_nxt[i3];
        cnt = cnt + 1;
    } while ((diff > e) && (cnt < max));

    gm_rt_cleanup();
    if (iters) *iters = cnt;
    delete gp;
    return 0;
}

/* ---- hop_dist: plain emission against the reference runtime (spinlock form) ---- */
int ref_hop_dist(int32_t N, int32_t M, const int32_t* begin, const int32_t* node_idx,
                 int32_t root_, int32_t* G_dist, int nthreads) {
    gm_graph* gp = make_graph(N, M, begin, node_idx);
    gm_graph& G = *gp;
    node_t root = root_;
    gm_rt_set_num_threads(nthreads);

    gm_rt_initialize();
    G.freeze();

    bool fin = false;
    bool* G_updated = gm_rt_allocate_bool(G.num_nodes(), gm_rt_thread_id());
    bool* G_updated_nxt = gm_rt_allocate_bool(G.num_nodes(), gm_rt_thread_id());
    int32_t* G_dist_nxt = gm_rt_allocate_int(G.num_nodes(), gm_rt_thread_id());

    fin = false;
    #pragma omp parallel for
    for (node_t t0 = 0; t0 < G.num_nodes(); t0++) {
        G_dist[t0] = (t0 == root) ? 0 : INT_MAX;
        G_updated[t0] = (t0 == root) ? true : false;
        G_dist_nxt[t0] = G_dist[t0];
        G_updated_nxt[t0] = G_updated[t0];
    }
    while (!fin) {
        bool __E8 = false;
        fin = true;
        __E8 = false;
        #pragma omp parallel for schedule(dynamic,128)
        for (node_t n = 0; n < G.num_nodes(); n++) {
            if (G_updated[n]) {
                for (edge_t s_idx = G.begin[n]; s_idx < G.begin[n + 1]; s_idx++) {
                    node_t s = G.node_idx[s_idx];
                    { // argmin(argmax) - test and test-and-set
                        int32_t dist_nxt_new = G_dist[n] + 1;
                        if (G_dist_nxt[s] > dist_nxt_new) {
                            bool updated_nxt_arg = true;
                            gm_spinlock_acquire_for_node(s);
                            if (G_dist_nxt[s] > dist_nxt_new) {
                                G_dist_nxt[s] = dist_nxt_new;
                                G_updated_nxt[s] = updated_nxt_arg;
                            }
                            gm_spinlock_release_for_node(s);
                        }
                    }
                }
            }
        }
        #pragma omp parallel
        {
            bool __E8_prv = false;
            #pragma omp for nowait
            for (node_t t4 = 0; t4 < G.num_nodes(); t4++) {
                G_dist[t4] = G_dist_nxt[t4];
                G_updated[t4] = G_updated_nxt[t4];
                G_updated_nxt[t4] = false;
                __E8_prv = __E8_prv || G_updated[t4];
            }
            ATOMIC_OR(&__E8, __E8_prv);
        }
        fin = !__E8;
    }
    gm_rt_cleanup();
    delete gp;
    return 0;
}

/* ---- sssp: plain emission (apps/src/sssp.gm:1-30) against the reference runtime: hop_dist's loop with the
 * edge property G_len[s_idx] (`Edge e = s.ToEdge()` is the neighbour iterator) ---- */
int ref_sssp(int32_t N, int32_t M, const int32_t* begin, const int32_t* node_idx, const int32_t* G_len,
             int32_t root_, int32_t* G_dist, int nthreads) {
    gm_graph* gp = make_graph(N, M, begin, node_idx);
    gm_graph& G = *gp;
    node_t root = root_;
    gm_rt_set_num_threads(nthreads);

    gm_rt_initialize();
    G.freeze();

    bool fin = false;
    bool* G_updated = gm_rt_allocate_bool(G.num_nodes(), gm_rt_thread_id());
    bool* G_updated_nxt = gm_rt_allocate_bool(G.num_nodes(), gm_rt_thread_id());
    int32_t* G_dist_nxt = gm_rt_allocate_int(G.num_nodes(), gm_rt_thread_id());

    fin = false;
    #pragma omp parallel for
    for (node_t t0 = 0; t0 < G.num_nodes(); t0++) {
        G_dist[t0] = (t0 == root) ? 0 : INT_MAX;
        G_updated[t0] = (t0 == root) ? true : false;
        G_dist_nxt[t0] = G_dist[t0];
        G_updated_nxt[t0] = G_updated[t0];
    }
    while (!fin) {
        bool __E8 = false;
        fin = true;
        __E8 = false;
        #pragma omp parallel for schedule(dynamic,128)
        for (node_t n = 0; n < G.num_nodes(); n++) {
            if (G_updated[n]) {
                for (edge_t s_idx = G.begin[n]; s_idx < G.begin[n + 1]; s_idx++) {
                    node_t s = G.node_idx[s_idx];
                    edge_t e;
                    e = s_idx;
                    { // argmin(argmax) - test and test-and-set
                        int32_t dist_nxt_new = G_dist[n] + G_len[e];
                        if (G_dist_nxt[s] > dist_nxt_new) {
                            bool updated_nxt_arg = true;
                            gm_spinlock_acquire_for_node(s);
                            if (G_dist_nxt[s] > dist_nxt_new) {
                                G_dist_nxt[s] = dist_nxt_new;
                                G_updated_nxt[s] = updated_nxt_arg;
                            }
                            gm_spinlock_release_for_node(s);
                        }
                    }
                }
            }
        }
        #pragma omp parallel
        {
            bool __E8_prv = false;
            #pragma omp for nowait
            for (node_t t4 = 0; t4 < G.num_nodes(); t4++) {
                G_dist[t4] = G_dist_nxt[t4];
                G_updated[t4] = G_updated_nxt[t4];
                G_updated_nxt[t4] = false;
                __E8_prv = __E8_prv || G_updated[t4];
            }
            ATOMIC_OR(&__E8, __E8_prv);
        }
        fin = !__E8;
    }
    gm_rt_cleanup();
    delete gp;
    return 0;
}

/* ---- avg_teen_cnt / conduct: plain emissions (apps/src/avg_teen_cnt.gm, conduct.gm) against the reference
 * runtime (gm_graph reverse edges, ATOMIC_ADD); see oracle/gm_oracle.c for the generator rules followed ---- */
float ref_avg_teen_cnt(int32_t N, int32_t M, const int32_t* begin, const int32_t* node_idx,
                       const int32_t* G_age, int32_t* G_teen_cnt, int32_t K, int nthreads) {
    gm_graph* gp = make_graph(N, M, begin, node_idx);
    gm_graph& G = *gp;
    gm_rt_set_num_threads(nthreads);
    gm_rt_initialize();
    G.freeze();
    G.make_reverse_edges();

    float avg = 0;
    double _avg4 = 0;
    int64_t _cnt3 = 0;
    int32_t __S2 = 0;
    #pragma omp parallel for schedule(dynamic,128)
    for (node_t n = 0; n < G.num_nodes(); n++) {
        int32_t __S1 = 0;
        for (edge_t t_idx = G.r_begin[n]; t_idx < G.r_begin[n + 1]; t_idx++) {
            node_t t = G.r_node_idx[t_idx];
            if ((G_age[t] >= 10) && (G_age[t] < 20)) __S1 = __S1 + 1;
        }
        G_teen_cnt[n] = __S1;
    }
    #pragma omp parallel
    {
        int32_t __S2_prv = 0;
        int64_t _cnt3_prv = 0;
        #pragma omp for nowait
        for (node_t n0 = 0; n0 < G.num_nodes(); n0++) {
            if (G_age[n0] > K) {
                __S2_prv = __S2_prv + G_teen_cnt[n0];
                _cnt3_prv = _cnt3_prv + 1;
            }
        }
        ATOMIC_ADD<int64_t>(&_cnt3, _cnt3_prv);
        ATOMIC_ADD<int32_t>(&__S2, __S2_prv);
    }
    _avg4 = (0 == _cnt3) ? ((float)(0.000000)) : (__S2 / ((double)_cnt3));
    avg = (float)_avg4;
    gm_rt_cleanup();
    delete gp;
    return avg;
}

float ref_conduct(int32_t N, int32_t M, const int32_t* begin, const int32_t* node_idx,
                  const int32_t* G_member, int32_t num, int nthreads) {
    gm_graph* gp = make_graph(N, M, begin, node_idx);
    gm_graph& G = *gp;
    gm_rt_set_num_threads(nthreads);
    gm_rt_initialize();
    G.freeze();

    int32_t Din = 0, Dout = 0, Cross = 0;
    #pragma omp parallel
    {
        int32_t Din_prv = 0;
        #pragma omp for nowait
        for (node_t u = 0; u < G.num_nodes(); u++)
            if (G_member[u] == num) Din_prv = Din_prv + (G.begin[u + 1] - G.begin[u]);
        ATOMIC_ADD<int32_t>(&Din, Din_prv);
    }
    #pragma omp parallel
    {
        int32_t Dout_prv = 0;
        #pragma omp for nowait
        for (node_t u0 = 0; u0 < G.num_nodes(); u0++)
            if (G_member[u0] != num) Dout_prv = Dout_prv + (G.begin[u0 + 1] - G.begin[u0]);
        ATOMIC_ADD<int32_t>(&Dout, Dout_prv);
    }
    #pragma omp parallel
    {
        int32_t Cross_prv = 0;
        #pragma omp for nowait schedule(dynamic,128)
        for (node_t u1 = 0; u1 < G.num_nodes(); u1++) {
            if (G_member[u1] == num) {
                int32_t __S3 = 0;
                for (edge_t j_idx = G.begin[u1]; j_idx < G.begin[u1 + 1]; j_idx++) {
                    node_t j = G.node_idx[j_idx];
                    if (G_member[j] != num) __S3 = __S3 + 1;
                }
                Cross_prv = Cross_prv + __S3;
            }
        }
        ATOMIC_ADD<int32_t>(&Cross, Cross_prv);
    }
    float m = (float)((Din < Dout) ? Din : Dout);
    float ret;
    if (m == 0) ret = (Cross == 0) ? ((float)(0.000000)) : FLT_MAX;
    else ret = Cross / m;
    gm_rt_cleanup();
    delete gp;
    return ret;
}

/* (The reference also checks in one piece of generator output, apps/output_cpp/gm_graph/test/sssp_dijkstra.cc.
 * It cannot serve as a second, pure-reference opinion on the distances: its priority map frees a node and
 * then writes to it -- gm_mutatable_priority_map.h:1007-1013, heap-use-after-free under AddressSanitizer on
 * the first removeMinKey_seq() -- so the cross-check of the sssp distances is scipy's Dijkstra instead,
 * oracle/make_golden.py.) */

/* ---- pure reference BFS: gm_bfs_template (gm_bfs_template.h:14-754) ---- */
class ref_bfs_t : public gm_bfs_template<short, true, false, false, false>
{
  public:
    ref_bfs_t(gm_graph& g, int32_t* lv) : gm_bfs_template<short, true, false, false, false>(g), level(lv) {}
  protected:
    virtual void visit_fw(node_t t) { level[t] = get_curr_level(); }
    virtual void visit_rv(node_t t) {}
    virtual bool check_navigator(node_t t, edge_t nx) { return true; }
  private:
    int32_t* level;
};

int ref_bfs_levels(int32_t N, int32_t M, const int32_t* begin, const int32_t* node_idx,
                   int32_t root, int32_t* level, int nthreads, int with_reverse) {
    gm_graph* gp = make_graph(N, M, begin, node_idx);
    gp->do_semi_sort();
    if (with_reverse) gp->make_reverse_edges();
    gm_rt_set_num_threads(nthreads);
    gm_rt_initialize();
    for (int32_t i = 0; i < N; i++) level[i] = INT_MAX;
    {
        ref_bfs_t bfs(*gp, level);
        bfs.prepare(root, gm_rt_get_num_threads());
        bfs.do_bfs_forward();
    }
    delete gp;
    return 0;
}

/* ---- the reference's uniform generator (graph_gen.cc:12-55), as is: rows as it leaves them (unsorted) ----
 * (the xorshift mode is only safe while gm_rand32 stays non-negative: `r % N` of a negative draw indexes
 * degree[] out of bounds, graph_gen.cc:27-35; the caller checks that with ref_rand32_min first) */
int ref_uniform_graph(int32_t N, int32_t M, long seed, int use_xorshift, int32_t* begin, int32_t* node_idx) {
    gm_graph G;
    create_uniform_random_graph_new(G, N, M, seed, use_xorshift != 0);
    memcpy(begin, G.begin, sizeof(int32_t) * ((size_t) N + 1));
    memcpy(node_idx, G.node_idx, sizeof(int32_t) * (size_t) M);
    return 0;
}
int32_t ref_rand32_min(long seed, int32_t n) {
    gm_rand r(seed);
    int32_t mn = 0x7fffffff;
    for (int32_t i = 0; i < n; i++) {
        int32_t v = (int32_t) r.rand();
        if (v < mn) mn = v;
    }
    return mn;
}

/* ---- the reference's gm_common_neighbor_iter (gm_common_neighbor_iter.cc), as is ---- */
int64_t ref_common_nbrs(int32_t N, int32_t M, const int32_t* begin, const int32_t* node_idx, int32_t n_pairs,
                        const int32_t* src, const int32_t* dst, int64_t* counts, int32_t* items, int64_t cap) {
    gm_graph* gp = make_graph(N, M, begin, node_idx);
    gp->do_semi_sort();
    int64_t total = 0;
    for (int32_t i = 0; i < n_pairs; i++) {
        gm_common_neighbor_iter it(*gp, src[i], dst[i]);
        int64_t c = 0;
        for (node_t w = it.get_next(); w != gm_graph::NIL_NODE; w = it.get_next()) {
            if (total < cap) items[total] = w;
            total++;
            c++;
        }
        counts[i] = c;
    }
    delete gp;
    return total;
}

/* triangle counting through that iterator (emitted form of Foreach(w: v.CommonNbrs(u))(w > u)) */
int64_t ref_triangle_counting_cn(int32_t N, int32_t M, const int32_t* begin, const int32_t* node_idx) {
    gm_graph* gp = make_graph(N, M, begin, node_idx);
    gm_graph& G = *gp;
    G.freeze();
    G.do_semi_sort();
    int64_t T = 0;
    for (node_t v = 0; v < G.num_nodes(); v++)
        for (edge_t u_idx = G.begin[v]; u_idx < G.begin[v + 1]; u_idx++) {
            node_t u = G.node_idx[u_idx];
            if (u > v) {
                gm_common_neighbor_iter w_I(G, v, u);
                for (node_t w = w_I.get_next(); w != gm_graph::NIL_NODE; w = w_I.get_next())
                    if (w > u) T = T + 1;
            }
        }
    delete gp;
    return T;
}

/* ---- comp_BC (apps/src/bc.gm): the traversal, the level bookkeeping, is_down_edge and the forward / reverse
 *      sweeps are the reference's gm_bfs_template<short, true, false, false, true> (save_child: DownNbrs is used,
 *      gm_cpp_gen_bfs.cc:88-98); visit_fw / visit_rv are the emitted loop bodies (gm_cpp_gen_foreach.cc:161-176). ---- */
class ref_bc_bfs_t : public gm_bfs_template<short, true, false, false, true>
{
  public:
    // _T: the graph object the template traverses; _G: the graph whose rows the visit bodies walk (the same graph;
    // two objects only so that the traversal can be run without reverse edges, see ref_bc)
    ref_bc_bfs_t(gm_graph& _T, gm_graph& _G, float*& _G_sigma, float*& _G_delta, float*& _G_BC, node_t& _s, int _skip_root)
        : gm_bfs_template<short, true, false, false, true>(_T), G(_G), G_sigma(_G_sigma), G_delta(_G_delta), G_BC(_G_BC),
          s(_s), skip_root(_skip_root) {}
  private:
    gm_graph& G;
    float*& G_sigma;
    float*& G_delta;
    float*& G_BC;
    node_t& s;
    int skip_root;
  protected:
    virtual void visit_fw(node_t v) {
        if (skip_root && v == s) return;   /* upstream's InBFS(...)(v != s) */
        float __S1 = ((float) (0.000000));
        for (edge_t w_idx = G.r_begin[v]; w_idx < G.r_begin[v + 1]; w_idx++) {
            node_t w = G.r_node_idx[w_idx];
            if (get_level(w) != (get_curr_level() - 1)) continue;
            __S1 = __S1 + G_sigma[w];
        }
        G_sigma[v] = __S1;
    }
    virtual void visit_rv(node_t v) {
        if (skip_root && v == s) return;
        float __S2 = ((float) (0.000000));
        for (edge_t w_idx = G.begin[v]; w_idx < G.begin[v + 1]; w_idx++) {
            node_t w = G.node_idx[w_idx];
            if (!is_down_edge(w_idx)) continue;
            __S2 = __S2 + G_sigma[v] / G_sigma[w] * (1 + G_delta[w]);
        }
        G_delta[v] = __S2;
        G_BC[v] = G_BC[v] + G_delta[v];
    }
    virtual bool check_navigator(node_t v, edge_t v_idx) { return true; }
};

/* One graph object with reverse edges, exactly what the emission does.  Known defects of the reference on this path,
 * all in gm_bfs_template.h: (1) the template may switch to its bottom-up state (ST_RD -> ST_RRD, :400-403 -- that
 * transition lacks the `!save_child` guard the other two have, :374,389), where a vertex stops at its first parent and
 * NO down edges are recorded (check_parent_rrd :689-708): DownNbrs then silently misses edges and BC comes out too
 * small (the same transition also lacks the has_reverse_edge() guard, so a graph without reverse edges crashes
 * there); (2) with several threads a down edge is lost when a second parent sees the child's visited bit before the
 * first has stored the child's level (:596-620); (3) the destructor, see below. */
int ref_bc(int32_t N, int32_t M, const int32_t* begin, const int32_t* node_idx, const int32_t* seeds, int32_t nseeds,
           int skip_root, float* G_BC, int nthreads) {
    gm_graph* gp = make_graph(N, M, begin, node_idx);
    gm_graph& G = *gp;
    gm_rt_set_num_threads(nthreads);
    gm_rt_initialize();
    G.freeze();
    G.do_semi_sort();
    G.make_reverse_edges();
    gm_graph* tp = gp;
    float* G_sigma = gm_rt_allocate_float(G.num_nodes(), gm_rt_thread_id());
    float* G_delta = gm_rt_allocate_float(G.num_nodes(), gm_rt_thread_id());
    #pragma omp parallel for
    for (node_t t0 = 0; t0 < G.num_nodes(); t0++) G_BC[t0] = 0;
    for (int32_t i = 0; i < nseeds; i++) {
        node_t s = seeds[i];
        #pragma omp parallel for
        for (node_t t1 = 0; t1 < G.num_nodes(); t1++) G_sigma[t1] = 0;
        G_sigma[s] = 1;
        /* The emission declares the BFS object on the stack.  With save_child its destructor does
         * `delete [] down_edge_set` on a pointer that came from `new std::set<edge_t>()` (gm_bfs_template.h:30-31,40):
         * glibc aborts ("free(): invalid pointer").  The harness therefore never destroys the object (a leak per
         * seed in a test tool); everything it computes before that point is the reference's own code. */
        ref_bc_bfs_t* _BFS = new ref_bc_bfs_t(*tp, G, G_sigma, G_delta, G_BC, s, skip_root);
        _BFS->prepare(s, gm_rt_get_num_threads());
        _BFS->do_bfs_forward();
        _BFS->do_bfs_reverse();
    }
    gm_rt_cleanup();
    if (tp != gp) delete tp;
    delete gp;
    return 0;
}

/* ---- triangle_counting: plain emission, HasEdgeTo through the reference's
 *      gm_graph::is_neighbor (gm_graph.cc:60-66,589-633) ---- */
int64_t ref_triangle_counting(int32_t N, int32_t M, const int32_t* begin, const int32_t* node_idx, int nthreads) {
    gm_graph* gp = make_graph(N, M, begin, node_idx);
    gm_graph& G = *gp;
    gm_rt_set_num_threads(nthreads);
    gm_rt_initialize();
    G.freeze();
    G.do_semi_sort();

    int64_t T = 0;
    #pragma omp parallel
    {
        int64_t T_prv = 0;
        #pragma omp for nowait schedule(dynamic,128)
        for (node_t v = 0; v < G.num_nodes(); v++) {
            for (edge_t u_idx = G.begin[v]; u_idx < G.begin[v + 1]; u_idx++) {
                node_t u = G.node_idx[u_idx];
                if (u > v) {
                    for (edge_t w_idx = G.begin[v]; w_idx < G.begin[v + 1]; w_idx++) {
                        node_t w = G.node_idx[w_idx];
                        if (w > u) {
                            if (G.is_neighbor(w, u)) T_prv = T_prv + 1;
                        }
                    }
                }
            }
        }
        ATOMIC_ADD<int64_t>(&T, T_prv);
    }
    gm_rt_cleanup();
    delete gp;
    return T;
}

/* is_neighbor probe for direct pinning of gmo_get_edge_idx_for_src_dest */
int ref_is_neighbor_many(int32_t N, int32_t M, const int32_t* begin, const int32_t* node_idx,
                         int n, const int32_t* src, const int32_t* to, int32_t* out_edge_idx) {
    gm_graph* gp = make_graph(N, M, begin, node_idx);
    gp->do_semi_sort();
    for (int i = 0; i < n; i++) out_edge_idx[i] = gp->get_edge_idx_for_src_dest(src[i], to[i]);
    delete gp;
    return 0;
}

}  // extern "C"
