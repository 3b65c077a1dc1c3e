// gmx_internal.h -- shared internals of libgmx (HIP, gfx950 only).
#ifndef GMX_INTERNAL_H_
#define GMX_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include <time.h>
#include <stdlib.h>
#include <string>
#include <vector>

#include "gmx.h"

void gmx_set_error(const char* fmt, ...);

#define GMX_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t _e = (call);                                                            \
        if (_e != hipSuccess) {                                                            \
            gmx_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(_e)); \
            return GMX_ERR_HIP;                                                            \
        }                                                                                  \
    } while (0)

#define GMX_CHECK(call)                 \
    do {                                \
        int _s = (call);                \
        if (_s != GMX_OK) return _s;    \
    } while (0)

#define GMX_REQUIRE(cond, ...)          \
    do {                                \
        if (!(cond)) {                  \
            gmx_set_error(__VA_ARGS__); \
            return GMX_ERR_ARG;         \
        }                               \
    } while (0)

// Simple owning device buffer.
template <typename T>
struct dbuf {
    T* p = nullptr;
    size_t n = 0;
    dbuf() {}
    dbuf(const dbuf&) = delete;
    dbuf& operator=(const dbuf&) = delete;
    ~dbuf() { release(); }
    int alloc(size_t count) {
        release();
        n = count;
        size_t bytes = (count ? count : 1) * sizeof(T);
        hipError_t e = hipMalloc((void**) &p, bytes);
        if (e != hipSuccess) {
            p = nullptr;
            n = 0;
            gmx_set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
            return GMX_ERR_NOMEM;
        }
        return GMX_OK;
    }
    void release() {
        if (p) (void) hipFree(p);
        p = nullptr;
        n = 0;
    }
    T* take() { T* q = p; p = nullptr; n = 0; return q; }
};

// ---- workspace: where the big temporaries of graph construction and plan builds live ----
// Device memory that has just been freed is not free: the driver wipes it before it hands it out again, and a
// hipMalloc that needs such memory waits for the wipe -- measured here as stalls of 1-2.4 s inside an RMAT-26 plan build
// (an 8 GB allocation behind the ~100 GB of temporaries the graph construction had freed), against 0.2 s without them.
// So the multi-gigabyte temporaries are carved from chunks that stay allocated for the life of the process (stack
// discipline: gmx_ws_scope rewinds on exit; a buffer's release() is a no-op) and nothing big is ever freed between a
// graph's construction and its plans.  gmx_workspace_release() (gmx.h) gives the chunks back.
struct gmx_ws_mark { size_t chunk, off; };
void* gmx_ws_alloc(size_t bytes);
gmx_ws_mark gmx_ws_top();
void gmx_ws_rewind(gmx_ws_mark m);
struct gmx_ws_scope {
    gmx_ws_mark m;
    gmx_ws_scope() : m(gmx_ws_top()) {}
    // (nothing may still be running on memory that the next build will hand out again: also on the error paths)
    ~gmx_ws_scope() { (void) hipDeviceSynchronize(); gmx_ws_rewind(m); }
    gmx_ws_scope(const gmx_ws_scope&) = delete;
    gmx_ws_scope& operator=(const gmx_ws_scope&) = delete;
};
// dbuf's interface on workspace memory (valid until the enclosing gmx_ws_scope ends)
template <typename T>
struct wbuf {
    T* p = nullptr;
    size_t n = 0;
    wbuf() {}
    wbuf(const wbuf&) = delete;
    wbuf& operator=(const wbuf&) = delete;
    int alloc(size_t count) {
        n = count;
        p = (T*) gmx_ws_alloc((count ? count : 1) * sizeof(T));
        if (!p) { n = 0; return GMX_ERR_NOMEM; }
        return GMX_OK;
    }
    void release() { p = nullptr; n = 0; }
};

struct gmx_bfs;

struct gmx_graph {
    int64_t V = 0, E = 0;
    bool has_reverse = false;
    // original numbering, rows ascending (semi-sorted)
    dbuf<int32_t> begin, node_idx, r_begin, r_node_idx;
    // gm_graph's e_idx2idx (gm_graph.h:141, do_semi_sort gm_graph.cc:468-503): slot of the uploaded (unsorted) forward
    // CSR that every slot of the sorted rows came from.  Empty: the upload was already in order (identity).
    dbuf<int32_t> e_idx2idx;
    int device = 0;
    // PageRank plans built by the whole-kernel entries (fp32, fp64), kept for the next call on the same
    // graph: the plan is graph preprocessing, like the reverse CSR.  Freed with the graph.
    gmx_pr* pr_cache[4] = {nullptr, nullptr, nullptr, nullptr};   // [2], [3]: the pull-sweep plans used for d outside (0, 1]
    // the same for the multi-GPU form of those entries (gmx_pr_multi.hip): N rank states driven by one host thread
    struct gmx_pr_multi* pr_multi_cache[2] = {nullptr, nullptr};
    bool pr_multi_refused = false;   // the multi-GPU exchange failed its first-contact check on this box: single GPU from then on
    // triangle counting: -1 not examined, 0 general graph, 1 symmetric and simple -> `tc_oriented` holds the
    // forward CSR of the same graph renumbered by ascending degree (its reverse CSR is the same arrays)
    int tc_sym_state = -1;
    gmx_graph* tc_oriented = nullptr;
    // on the oriented copy: the adjacency among its tc_hubs highest vertices (the hubs: ids >= V - tc_hubs) as a bit matrix,
    // row u - (V - tc_hubs), bit w - (V - tc_hubs) (gmx_tc.hip)
    dbuf<uint32_t> tc_hub_bits;
    int64_t tc_hubs = 0;
    // hop_dist: the single-rank traversal state (queues, bitmaps, dist[]) of the whole-kernel entry, kept for the
    // next call on the same graph instead of nine allocations per call
    gmx_bfs* bfs_cache = nullptr;
    // bottom-up BFS: per vertex the in-neighbour to try first (the one with most out-edges among the first of its
    // in-row; -1: no in-edges).  Graph preprocessing like the reverse CSR, built with the first traversal object.
    dbuf<int32_t> bfs_hint;
    // ... and the BFS_HUBS vertices with most out-edges ("hubs"; all vertices if there are fewer): a hint that is a hub is
    // stored as its slot in this list, and a bottom-up level probes the hubs' frontier bits in LDS (gmx_bfs.hip)
    dbuf<int32_t> bfs_hub_id;     // [bfs_hubs]; empty when every vertex is a hub (slot = id)
    int64_t bfs_hubs = 0;
    bool bfs_hint_plain = false;  // V > 2^30: the hints are plain vertex ids (no room for the flag bits)
};

// ---- graph construction helpers (gmx_graph.hip) ----
// keys are (row << 32 | col).  Sorts keys in place (double buffer), then writes
// begin[V+1] and idx[E].
int gmx_csr_from_keys(uint64_t* keys, uint64_t* keys_alt, int64_t V, int64_t E,
                      int32_t* begin, int32_t* idx, hipStream_t stream, int32_t* slots = nullptr);   // slots: [E] key order -> input position
// keys[e] = (row(e) << 32 | col(e)) from a CSR; if transpose, (col << 32 | row).
// perm (optional): map both endpoints through perm[] first.
int gmx_keys_from_csr(const int32_t* begin, const int32_t* idx, int64_t V, int64_t E,
                      bool transpose, const int32_t* perm, uint64_t* keys, hipStream_t stream);
int gmx_keys_from_edges(const int32_t* src, const int32_t* dst, int64_t E, bool transpose,
                        const int32_t* perm, uint64_t* keys, hipStream_t stream);

// ---- cold-source part of the PageRank sweep (gmx_pr_cold.hip) ----
// Sources whose id inside their rank range [r * slice, (r + 1) * slice) is >= T are "cold": their edges are not
// gathered by the pull sweep but pushed through plan-time-ordered bins (see gmx_pr_cold.hip).
struct pr_cold;
struct pr_cold_params {
    int elem;        // 4 (float) or 8 (double)
    int nranks;
    int64_t slice;   // ids per rank range of the contribution replica
    int64_t T;       // hot ids per rank range
    int64_t row_lo;  // first owned row (internal numbering)
    int64_t nactive; // owned rows with in-edges
    const int32_t* index_of_row;   // [rows] local row -> position in the active-row list (device)
    const int32_t* deg_by_id;      // [nranks * slice] out-degree by internal id (device; padding ids < 0), or NULL
};
// keys[Ec]: (row << 32 | source) of the cold in-edges of the owned rows, any order (device).
int pr_cold_create(const uint64_t* keys, int64_t Ec, const pr_cold_params& prm, hipStream_t s, pr_cold** out);
void pr_cold_free(pr_cold* c);
// what the fused finish needs to apply the PageRank update of active row i itself (see pr_combine_kernel)
struct pr_cold_fuse {
    const int32_t* active;     // [nactive] local row of active row i
    const int32_t* outdeg_c;   // [nactive]
    void* rk_c;                // [nactive] x elem: ranks, updated in place
    void* next_owned;          // owned range of the replica being produced, indexed by local row
    double base, d;
};
// enqueue phases 1-3: reads the contribution replica; fuse == NULL leaves the row sums in pr_cold_partial(),
// otherwise the rows are finished in place and pr_cold_diff_partials() holds the |val - rank| partials
int pr_cold_launch(pr_cold* c, const void* contrib, const pr_cold_fuse* fuse, hipStream_t s);
// the same in pieces (see pr_cold_set_parts in gmx_pr_cold.hip): phase 1 per tile class, phases 2-3 per row part
int pr_cold_set_parts(pr_cold* c, int nparts, const int64_t* act_bound, int64_t hub, int64_t live);
int pr_cold_parts(const pr_cold* c);
int64_t pr_cold_class_items(const pr_cold* c, int cls);   // phase-1 work items of a tile class
int pr_cold_gather(pr_cold* c, const void* contrib, int cls, hipStream_t s);
int pr_cold_accumulate(pr_cold* c, const pr_cold_fuse* fuse, int part, hipStream_t s);
const double* pr_cold_diff_partials(const pr_cold* c, int64_t* n);
bool pr_cold_covers_all_rows(const pr_cold* c);
// fp32 plans keep ONE 2^-62 fixed-point limb: true if that provably holds the 1e-6 bar for damping d on a graph of N
// vertices (see pr_cold_limb_guard in gmx_pr_cold.hip); fp64 plans (two limbs): always true
bool pr_cold_limb_guard(const pr_cold* c, double d, double N);
const void* pr_cold_partial(const pr_cold* c);   // [nactive] x elem, indexed like the per-slice partial sums
int64_t pr_cold_edges(const pr_cold* c);
int64_t pr_cold_items(const pr_cold* c);

// ---- whole-kernel PageRank over several GPUs from one host thread (gmx_pr_multi.hip) ----
struct gmx_pr_multi;
int gmx_pr_multi_ranks(const gmx_graph* g);    // ranks the entry uses for this graph (GMX_DEVICES, GMX_PR_RANKS); 1 = single GPU
int gmx_pr_multi_create(gmx_graph* g, int elem, int nranks, gmx_pr_multi** out);
int gmx_pr_multi_run(gmx_pr_multi* m, double e, double d, int32_t max_iter, void* rank_host, gmx_stats_t* stats);
void gmx_pr_multi_free(gmx_pr_multi* m);
bool gmx_pr_multi_verified(const gmx_pr_multi* m);   // false: the first exchange has not passed its check

// GMX_PLAN_TIMING=1: where a plan build spends its time (synchronises at every mark)
struct gmx_tick {
    bool on;
    double t0;
    const char* who;
    static double now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
    explicit gmx_tick(const char* w) : on(getenv("GMX_PLAN_TIMING") != nullptr), t0(0), who(w) { if (on) { (void) hipDeviceSynchronize(); t0 = now(); } }
    void mark(const char* what) {
        if (!on) return;
        (void) hipDeviceSynchronize();
        const double t = now();
        fprintf(stderr, "gmx timing %s: %-28s %8.2f ms\n", who, what, (t - t0) * 1e3);
        t0 = t;
    }
};

void gmx_touch_pagerank();
void gmx_touch_pr_cold();
void gmx_touch_bfs();
void gmx_warm_modules();   // once per process: load every translation unit's code object (see gmx_touch_*)

static inline int gmx_bits_for(int64_t v) {  // bits needed to represent values in [0, v)
    int b = 1;
    while ((1LL << b) < v && b < 32) b++;
    return b;
}

#endif
