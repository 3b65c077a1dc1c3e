/*
 * gmx.h -- C ABI of the MI355X-native Green-Marl graph-kernel hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI: its hot path is the
 * body of three generated C++ functions
 *     void    pagerank(gm_graph& G, double e, double d, int32_t max, double* G_pg_rank);
 *     void    hop_dist(gm_graph& G, int32_t* G_dist, node_t& root);
 *     int64_t triangle_counting(gm_graph& G);
 * (signature rules: /root/reference/src/backend_cpp/gm_cpp_gen.cc:520-608,938-1017;
 *  call sites: apps/output_cpp/src/pagerank_main.cc:28, hop_dist_main.cc:28,
 *  triangle_counting_main.cc:14) that read gm_graph's public CSR arrays
 * (apps/output_cpp/gm_graph/inc/gm_graph.h:133-142).  The C++ entries with
 * exactly those signatures live in green-marl_amd/generated/ and are a few
 * lines each: they hand gm_graph's raw arrays to the functions below.
 * Everything here is extern "C", plain pointers and sizes, int status returns
 * (0 = ok; the reference has no error channel, so the C++ entries turn a
 * non-zero status into fprintf(stderr)+abort()).
 *
 * All file:line citations are relative to /root/reference.
 */
#ifndef GMX_H_
#define GMX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GMX_OK            0
#define GMX_ERR_ARG      -1   /* bad argument                       */
#define GMX_ERR_HIP      -2   /* a HIP runtime call failed          */
#define GMX_ERR_NODEVICE -3   /* no gfx950 device visible           */
#define GMX_ERR_NOMEM    -4
#define GMX_ERR_STATE    -5   /* object not in the required state   */

/* node_t / edge_t are int32 (gm_graph_typedef.h:17-18, the reference default). */
typedef int32_t gmx_node_t;
typedef int32_t gmx_edge_t;

typedef struct gmx_graph gmx_graph_t;     /* device-resident CSR (+ reverse CSR) */
typedef struct gmx_pr    gmx_pr_t;        /* device-resident PageRank state      */

/* Message for the last non-zero status returned on this thread. */
const char* gmx_last_error(void);

/* ---- device ---- */
int gmx_device_count(int* count);
int gmx_set_device(int device);
typedef struct {
    char     name[128];
    char     arch[32];          /* "gfx950..." */
    int32_t  compute_units;
    int32_t  clock_mhz;
    int64_t  hbm_bytes;
    int32_t  l2_bytes;
    int32_t  lds_bytes_per_cu;
} gmx_device_info_t;
int gmx_device_info(gmx_device_info_t* info);
/* Measured device-to-device copy rate (GB/s, bytes read + bytes written) of a `bytes`-sized buffer, `iters` copies:
 * the achievable HBM ceiling bench.py reports next to the data-sheet peak. */
int gmx_copy_bandwidth(int64_t bytes, int iters, double* gbs);

/* The multi-gigabyte temporaries of graph construction and plan builds come from a workspace that stays allocated for
 * the life of the process (freed device memory is wiped by the driver before reuse, and allocations that need it stall
 * for seconds at RMAT-26 sizes).  gmx_workspace_bytes: what it holds now; gmx_workspace_release: give it back (e.g. once
 * every graph and plan is built; anything built later allocates it again). */
int64_t gmx_workspace_bytes(void);
int gmx_workspace_release(void);

/* ---- graph: replaces the emitted prologue `G.freeze(); G.make_reverse_edges();
 *      [G.do_semi_sort();]` + Shoal copy-in (gm_cpp_gen.cc:1307-1368, 670-778) ---- */

#define GMX_GRAPH_SORT_ROWS    0x1u  /* rows of node_idx are not sorted yet: do_semi_sort on device   */
#define GMX_GRAPH_NO_REVERSE   0x2u  /* do not keep a reverse CSR (BFS top-down only / TC binary search) */

/* Upload a host CSR.  begin[V+1], node_idx[E] = gm_graph::begin / node_idx.
 * r_begin / r_node_idx = gm_graph::r_begin / r_node_idx, or NULL: the reverse
 * CSR is then built on the device (gm_graph::make_reverse_edges, gm_graph.cc:205-304,
 * followed by do_semi_sort_reverse, :461-466). */
int gmx_graph_upload(const gmx_edge_t* begin, const gmx_node_t* node_idx,
                     const gmx_edge_t* r_begin, const gmx_node_t* r_node_idx,
                     int64_t V, int64_t E, uint32_t flags, gmx_graph_t** out);

/* Build from an unordered host edge list (src[i] -> dst[i]); rows come out
 * semi-sorted, multi-edges and self loops are kept. */
int gmx_graph_from_edges(const gmx_node_t* src, const gmx_node_t* dst,
                         int64_t V, int64_t E, uint32_t flags, gmx_graph_t** out);

/* create_RMAT_graph(N, M, rseed, a, b, c, permute) on the device, bit-compatible
 * with the reference's srand48/drand48 stream (graph_gen.cc:159-287), followed by
 * do_semi_sort + make_reverse_edges as load_binary does
 * (gm_graph_binary_loader.cc:191-197). */
int gmx_graph_create_rmat(int64_t N, int64_t M, long seed, double a, double b, double c,
                          int permute, uint32_t flags, gmx_graph_t** out);

/* Undirected simple version of g (both orientations, no duplicates, no self loops); the result is
 * its own transpose.  Measurement preparation of the triangle-counting config. */
int gmx_graph_symmetrize(const gmx_graph_t* g, gmx_graph_t** out);
/* The same calls for a host side built with GM_EDGE64 (edge_t = int64_t, gm_graph_typedef.h:8-20): edge offsets and edge
 * maps as int64 arrays.  The device keeps 32-bit edge offsets: E must be below 2^31 - 2^27 (GMX_ERR_ARG otherwise). */
int gmx_graph_upload_e64(const int64_t* begin, const gmx_node_t* node_idx, const int64_t* r_begin, const gmx_node_t* r_node_idx,
                         int64_t V, int64_t E, uint32_t flags, gmx_graph_t** out);
int gmx_graph_download_e64(const gmx_graph_t* g, int64_t* begin, gmx_node_t* node_idx, int64_t* r_begin, gmx_node_t* r_node_idx);
int gmx_graph_edge_order_e64(const gmx_graph_t* g, int64_t* e_idx2idx, int* is_identity);
int gmx_graph_reverse_edge_map_e64(const gmx_graph_t* g, int64_t* e_rev2idx);
int gmx_graph_free(gmx_graph_t* g);
int64_t gmx_graph_num_nodes(const gmx_graph_t* g);
int64_t gmx_graph_num_edges(const gmx_graph_t* g);
/* gm_graph's e_idx2idx[E] (gm_graph.h:141, do_semi_sort gm_graph.cc:468-503) after an upload with
 * GMX_GRAPH_SORT_ROWS: for every slot of the sorted rows, the slot of the uploaded CSR it came from (equal
 * destinations keep their order).  *is_identity = 1 when the upload was already in order; e_idx2idx is then not
 * written.  Edge properties passed later (gmx_sssp's len) are indexed by the UPLOADED slots. */
int gmx_graph_edge_order(const gmx_graph_t* g, gmx_edge_t* e_idx2idx, int* is_identity);
/* gm_graph's e_rev2idx[E] (gm_graph.h:141-142): forward slot mirrored by each slot of the reverse CSR. */
int gmx_graph_reverse_edge_map(const gmx_graph_t* g, gmx_edge_t* e_rev2idx);
/* Copy the device CSR back (any pointer may be NULL).  Lets a host gm_graph be
 * filled from a device-generated graph. */
int gmx_graph_download(const gmx_graph_t* g, gmx_edge_t* begin, gmx_node_t* node_idx,
                       gmx_edge_t* r_begin, gmx_node_t* r_node_idx);

/* ---- run statistics (all optional outputs) ---- */
typedef struct {
    int32_t iterations;       /* pagerank: cnt; hop_dist: levels; tc: 1          */
    int32_t reserved;
    double  last_diff;        /* pagerank: diff of the last iteration            */
    double  kernel_ms;        /* device time of the timed hot loop (hipEvents)   */
    double  h2d_ms;           /* property copy-in  (Shoal copy-in analogue)      */
    double  d2h_ms;           /* property copy-out (Shoal copy-back analogue)    */
    int64_t edges_examined;   /* hop_dist: edges actually inspected              */
    int64_t vertices_reached; /* hop_dist                                         */
    int64_t edges_reached;    /* hop_dist: out-edges of the reached vertices (Graph500 TEPS numerator) */
} gmx_stats_t;

/* ---- whole-kernel entries (what the three generated C++ functions call) ---- */

/* pagerank(G, e, d, max, G_pg_rank): fp64 storage + fp64 arithmetic.
 * rank_host[V] is caller-owned (pagerank_main.cc:18-25) and written on return. */
int gmx_pagerank_f64(gmx_graph_t* g, double e, double d, int32_t max_iter,
                     double* rank_host, gmx_stats_t* stats);
/* Node_Prop<Float> variant (BASELINE config 2): fp32 STORAGE; partial row sums are formed in fp64, rounded to fp32 once
 * per (source tile, row) pair, added exactly in 64-bit fixed point and rounded once more per vertex.  Within 1e-6
 * relative of the fp64 result; not bit- or iteration-comparable with what gm_comp would emit
 * for a Float property (fp32 sums in thread order), whose iteration count near the threshold e may differ. */
int gmx_pagerank_f32(gmx_graph_t* g, float e, float d, int32_t max_iter,
                     float* rank_host, gmx_stats_t* stats);

/* hop_dist(G, G_dist, root): BFS depth along out-edges, INT_MAX = unreached. */
int gmx_hop_dist(gmx_graph_t* g, gmx_node_t root, int32_t* dist_host, gmx_stats_t* stats);

/* hop_dist over several GPUs: replicated CSR, every rank runs gmx_bfs_step_begin /
 * [exchange] / gmx_bfs_step_end per level until the frontier is empty.  step_begin runs a top-down level
 * entirely (needs_exchange = 0: every rank expands the whole, small frontier) or the rank's share of a
 * bottom-up level, writing its slice [slice_offset, slice_offset + slice_words) of the found bitmap
 * (64 vertices per word; needs_exchange = 1 when nranks > 1: all-gather the slices in place).  step_end
 * applies the bitmap to the rank's dist[] replica and returns the next frontier's size -- the same number on
 * every rank, so the ranks agree on direction and termination without further communication. */
typedef struct gmx_bfs gmx_bfs_t;
int gmx_bfs_create(gmx_graph_t* g, int rank, int nranks, gmx_bfs_t** out);
int gmx_bfs_free(gmx_bfs_t* b);
int gmx_bfs_start(gmx_bfs_t* b, gmx_node_t root);
int gmx_bfs_step_begin(gmx_bfs_t* b, int* needs_exchange);
int gmx_bfs_found_bitmap(gmx_bfs_t* b, void** words, int64_t* total_words, int64_t* slice_offset, int64_t* slice_words);
int gmx_bfs_step_end(gmx_bfs_t* b, int64_t* next_count);
int gmx_bfs_download(gmx_bfs_t* b, int32_t* dist_host, gmx_stats_t* stats);

/* The BFS object of `InBFS(v: G.Nodes From s) {..} InReverse {..}` (gm_bfs_template.h:14-312 as instantiated by
 * gm_cpp_gen_bfs.cc:88-275: level_t = short, save_child when DownNbrs is used).
 * gmx_bfs_levels = prepare(root) + do_bfs_forward(): level_host[V] as the template's visited_level (unvisited = -2,
 * gm_bfs_template.h:725), *nlevels = deepest level + 1.
 * gmx_bc = comp_BC(G, BC, Seeds) of apps/src/bc.gm (driver apps/output_cpp/src/bc_main.cc:43): for every seed, the
 * traversal, visit_fw (sigma = Sum over UpNbrs) level by level, visit_rv (delta = Sum over DownNbrs, BC += delta)
 * deepest level first; Float properties, every Sum added in row-slot order.  skip_root = 0 is this fork's bc.gm (the
 * root is visited like any vertex, so its sigma = 1 is overwritten by an empty sum: all sigma 0, NaN wherever a
 * reached vertex has a BFS child); skip_root = 1 is upstream Green-Marl's `(v != s)` form.  bc_host[V] is written. */
int gmx_bfs_levels(gmx_graph_t* g, gmx_node_t root, int16_t* level_host, int32_t* nlevels);
int gmx_bc(gmx_graph_t* g, const gmx_node_t* seeds, int32_t nseeds, int skip_root, float* bc_host, gmx_stats_t* stats);

/* sssp(G, dist, len, root) (apps/src/sssp.gm; driver apps/output_cpp/src/sssp_main.cc:42): shortest path lengths
 * over out-edges with the caller's edge property len[E] (indexed by forward edge slot), INT_MAX = unreachable.
 * stats: iterations = relaxation rounds, h2d_ms = upload of len, vertices_reached = queue entries over all rounds. */
int gmx_sssp(gmx_graph_t* g, gmx_node_t root, const int32_t* len_host, int32_t* dist_host, gmx_stats_t* stats);

/* avg_teen_cnt(G, age, teen_cnt, K) (apps/src/avg_teen_cnt.gm; driver avg_teen_cnt_main.cc:24) and
 * conduct(G, member, num) (apps/src/conduct.gm; driver conduct_main.cc:45): count-reductions over neighbours
 * with the caller's int32 node property; integers exact, the returned float formed by the emitted expression. */
int gmx_avg_teen_cnt(gmx_graph_t* g, const int32_t* age_host, int32_t K, int32_t* teen_cnt_host, float* avg, gmx_stats_t* stats);
int gmx_conduct(gmx_graph_t* g, const int32_t* member_host, int32_t num, float* result, gmx_stats_t* stats);

/* triangle_counting(G) with the emitted multiplicity rule. */
int gmx_triangle_counting(gmx_graph_t* g, int64_t* count, gmx_stats_t* stats);
/* Multi-GPU form (replicated CSR, final all-reduce of int64): the count contributed by part
 * `part` of `nparts` of the edge slots (dealt in blocks, round-robin); the parts add up to the full count. */
int gmx_triangle_counting_part(gmx_graph_t* g, int part, int nparts, int64_t* count, gmx_stats_t* stats);

/* The common-neighbour iterator, gm_common_neighbor_iter(G, s, d) (gm_common_neighbor_iter.cc:21-44): the slots of s's
 * row, in order and with their multiplicity, whose value occurs in d's row (Foreach(u: s.CommonNbrs(d)) <=>
 * Foreach(u: s.Nbrs)(d.isNbr(u)), gm_common_neighbor_iter.h:11-17).  gmx_common_nbrs lists them for one pair (*n = how
 * many there are; at most cap are written), gmx_common_nbr_counts counts them for many pairs, and
 * gmx_triangle_counting_cn is triangle counting written with the iterator
 *   Foreach(v: G.Nodes) Foreach(u: v.Nbrs)(u > v) Foreach(w: v.CommonNbrs(u))(w > u) T += 1;
 * (equal to gmx_triangle_counting on a symmetric graph; on a directed one it asks for u -> w where the emitted
 * triangle_counting.gm asks for w -> u). */
int gmx_common_nbrs(gmx_graph_t* g, gmx_node_t s, gmx_node_t d, gmx_node_t* out, int64_t cap, int64_t* n);
int gmx_common_nbr_counts(gmx_graph_t* g, const gmx_node_t* src, const gmx_node_t* dst, int64_t npairs, int64_t* counts);
int gmx_triangle_counting_cn(gmx_graph_t* g, int64_t* count, gmx_stats_t* stats);

/* ---- device-resident PageRank stepping (bench.py / multi-GPU driver) ----
 * A gmx_pr_t owns the rows [row_lo,row_hi) of the (internally relabelled) graph
 * and a full replica of the contribution vector.  One step = one PageRank
 * iteration over the owned rows.  With nranks > 1 the caller exchanges the
 * owned slice of the new contribution vector between steps (all-gather over
 * RCCL); slices are equal sized and contiguous by construction. */
#define GMX_PR_F32 4
#define GMX_PR_F64 8
/* options bit flags */
#define GMX_PR_RELABEL   0x1u   /* degree-sorted internal numbering (default on in whole-kernel entries) */
#define GMX_PR_HOT_LDS   0x2u   /* keep the hottest contributions in LDS */
#define GMX_PR_SLICED    0x4u   /* split in-edges by source slice, one slice per XCD L2 (needs GMX_PR_RELABEL) */
#define GMX_PR_COLD_PB   0x8u   /* with GMX_PR_SLICED: edges from the cold sources (the tail of the degree order, past what the
                                   L2s hold; GMX_PR_COLD=<hot ids per rank range> overrides the size rule) leave the pull
                                   sweep and go through plan-time-ordered destination bins: no 128-byte line per gather */
/* The option set the whole-kernel entries use for a graph of V vertices on nranks ranks. */
uint32_t gmx_pr_default_options(int64_t V, int nranks);
int gmx_pr_create(gmx_graph_t* g, int elem_bytes, int rank, int nranks, uint32_t options, gmx_pr_t** out);
int gmx_pr_free(gmx_pr_t* p);
int gmx_pr_reset(gmx_pr_t* p, double d);                 /* rank = 1/N, contrib = rank/outdeg, cnt = 0 */
/* Enqueue one iteration on `stream` (a hipStream_t, NULL = default stream). Asynchronous. */
int gmx_pr_step(gmx_pr_t* p, void* stream);
/* Row chunks: with C > 1 a step is enqueued as C pieces (gmx_pr_step_chunk 0..C-1, in that order), each
 * finishing the new contributions of one sub-range [offset, offset+count) of the rank's exchanged prefix
 * (gmx_pr_chunk_range; identical on every rank, together they tile [0, gmx_pr_exchange_count)), so that
 * the exchange of chunk c runs while chunk c+1 is computed.  The sub-ranges are walked from the back: the
 * first pieces hold most of the rows but few edges, the last one the hubs.  gmx_pr_step() enqueues all
 * chunks.  Only the sliced variant splits; otherwise the chunk count stays 1.
 * gmx_pr_contrib_next_full is the replica the running step writes. */
int gmx_pr_set_chunks(gmx_pr_t* p, int chunks);
int gmx_pr_num_chunks(gmx_pr_t* p, int* chunks);
int gmx_pr_chunk_range(gmx_pr_t* p, int chunk, int64_t* offset, int64_t* count);
int gmx_pr_step_chunk(gmx_pr_t* p, int chunk, void* stream);
int gmx_pr_contrib_next_full(gmx_pr_t* p, void** dev_ptr, int64_t* count);
/* Peer push: the exchange of the N > 1 step without a collective kernel.  The host side passes the ranks'
 * replica handles around (gmx_pr_contrib_buffers -> gmx_ipc_export -> its own transport -> gmx_ipc_open) and
 * hands the mapped pointers to gmx_pr_set_peers (arrays of nranks entries, own entry ignored).  After
 * gmx_pr_step_chunk(c, stream), gmx_pr_push_chunk(c, stream) copies the chunk's piece into every peer's
 * replica on per-peer copy streams (SDMA over xGMI), ordered after the chunk's kernels;
 * gmx_pr_push_current pushes the whole exchanged prefix of the current replica (after a reset);
 * gmx_pr_push_join makes `stream` wait for all copies issued so far.  The caller then runs its per-step
 * barrier (e.g. the all-reduce of diff): a rank may start the next step only after every rank's copies
 * have completed. */
#define GMX_IPC_HANDLE_BYTES 64
int gmx_ipc_export(void* dev_ptr, void* handle /* GMX_IPC_HANDLE_BYTES */);
int gmx_ipc_open(const void* handle, void** dev_ptr);
int gmx_ipc_close(void* dev_ptr);
int gmx_pr_contrib_buffers(gmx_pr_t* p, void** buf0, void** buf1, int64_t* bytes);
int gmx_pr_set_peers(gmx_pr_t* p, void* const* peer_buf0, void* const* peer_buf1);
int gmx_pr_push_chunk(gmx_pr_t* p, int chunk, void* stream);
int gmx_pr_push_current(gmx_pr_t* p, void* stream);
int gmx_pr_push_join(gmx_pr_t* p, void* stream);
/* Packed form of the push ("send only what is read").  Rank q reads source w iff w has an out-edge into a row q owns;
 * on an 8-rank partition of RMAT-26 that is 46 % of the (source, reader) incidences of the full prefixes.  The plan of a
 * rank (nranks > 1, degree order) holds, per peer, the sorted positions of ITS range that the peer reads and the
 * positions of the peer's range that IT reads -- the same list on both sides by construction (gmx_pr_packed_info
 * returns GMX_ERR_STATE when the plan has none).  gmx_pr_push_packed(c) gathers the chunk's entries of those lists into a
 * staging buffer and copies each peer's piece into the peer's landing zone (double buffered by replica parity; wired
 * up like the replicas: gmx_pr_recv_buffers -> gmx_ipc_export -> transport -> gmx_ipc_open -> gmx_pr_set_peers_packed
 * with the offset of the caller's segment in each peer's zone = the peer's recv_offsets[caller]); chunk -1 pushes the
 * current replica's whole prefix (after a reset).  After the per-step barrier gmx_pr_unpack(c) scatters what has landed
 * into the replica the next step reads (c = -1: all chunks).  Positions nobody reads are never touched: afterwards the
 * replicas agree with an all-gather only on the positions gmx_pr_recv_list names. */
int gmx_pr_packed_info(gmx_pr_t* p, int64_t* send_counts, int64_t* recv_counts, int64_t* recv_offsets);   /* [nranks] each, elements */
int gmx_pr_recv_buffers(gmx_pr_t* p, void** buf0, void** buf1, int64_t* bytes);
int gmx_pr_set_peers_packed(gmx_pr_t* p, void* const* peer_recv0, void* const* peer_recv1, const int64_t* my_offset, const int64_t* my_count);
int gmx_pr_push_packed(gmx_pr_t* p, int chunk, void* stream);
int gmx_pr_unpack(gmx_pr_t* p, int chunk, void* stream);
int gmx_pr_recv_list(gmx_pr_t* p, int r, void** dev_ptr, int64_t* count);
/* Pipelined form of the pushed step (plans with every in-edge binned: gmx_pr_gather_classes returns 2, else 0).
 * Phase 1 of the binned sweep is the only part of a step that reads the peers' contributions, and most of its work
 * sits in the tiles of the hub sources, whose contributions are the small LAST chunk of every rank's exchange.  So:
 *   gmx_pr_step_gather(p, 0, s)   phase 1 over the tiles that hold hub sources only   (needs the peers' last chunk)
 *   gmx_pr_step_gather(p, 1, s)   phase 1 over the other tiles                        (needs all chunks)
 *   gmx_pr_step_chunk(p, c, s) / gmx_pr_push_chunk(p, c, s) for c = 0 .. chunks-1 as before (the chunks' copies
 *   alternate between two sets of copy streams, so the hub chunk does not queue behind the tail chunk)
 * and the caller may start class 0 of the next step as soon as every rank's LAST chunk has landed
 * (gmx_pr_push_join_chunk(p, chunks-1, s) + its barrier), while the earlier chunks are still travelling; class 1
 * waits for those.  Without the gather calls gmx_pr_step_chunk(p, 0, s) enqueues phase 1 itself. */
/* All launches of one gmx_pr_t (gmx_pr_step, gmx_pr_step_gather, gmx_pr_step_chunk) must be enqueued in call order on
 * ONE stream: the persistent kernels of the binned sweep draw their work from device counters that advance from
 * launch to launch, and the host passes each launch the value it expects to find. */
int gmx_pr_gather_classes(gmx_pr_t* p, int* classes);
int gmx_pr_gather_items(gmx_pr_t* p, int tile_class, int64_t* items);   /* phase-1 work items of a class (set by gmx_pr_set_chunks) */
int gmx_pr_step_gather(gmx_pr_t* p, int tile_class, void* stream);
int gmx_pr_push_join_chunk(gmx_pr_t* p, int chunk, void* stream);
/* Device pointer + element count of the slice of the *current* contribution
 * vector this rank produced in the last step (for the exchange), and of the
 * whole replica. */
int gmx_pr_contrib_slice(gmx_pr_t* p, void** dev_ptr, int64_t* count);
int gmx_pr_contrib_full(gmx_pr_t* p, void** dev_ptr, int64_t* count);
/* Leading entries of every rank's range that have to be exchanged: vertices without out-edges are never
 * gathered and sit at the end of each range in the degree order (same value on every rank; <= slice count). */
int gmx_pr_exchange_count(gmx_pr_t* p, int64_t* count);
/* Device pointer to the fp64 `diff` partial of the last step (1 element). */
int gmx_pr_diff_ptr(gmx_pr_t* p, void** dev_ptr);
/* Blocking: returns diff of the last step (local rows only). */
int gmx_pr_diff(gmx_pr_t* p, void* stream, double* diff);
/* Blocking: ranks of the owned rows scattered into rank_host[V] at original
 * vertex ids (other entries untouched). */
int gmx_pr_download(gmx_pr_t* p, void* rank_host);
/* Device timing of the kernels of gmx_pr_step (the row-reduction kernel plus its small fix-up /
 * combine / diff-reduce kernels: together they move the algorithmic bytes of one iteration), with
 * hipEvents recorded on the stream they are launched on.  enable != 0 starts a fresh
 * measurement; gmx_pr_kernel_time blocks until the recorded steps finished and returns their
 * count and mean duration (bench.py's roofline.achieved uses exactly this). */
int gmx_pr_timing(gmx_pr_t* p, int enable);
int gmx_pr_kernel_time(gmx_pr_t* p, int32_t* launches, double* mean_ms);
/* '+'-joined names of those kernels as rocprofv3 prints them (prefix match). */
const char* gmx_pr_kernel_name(gmx_pr_t* p);
/* Algorithmic bytes / edges one step of this rank processes. */
int gmx_pr_work(gmx_pr_t* p, int64_t* edges, int64_t* rows, int64_t* algorithmic_bytes);
/* The binned part of the plan (GMX_PR_COLD_PB): hot ids per rank range (-1 = no binned part, 0 = every edge is
 * binned), edges taken out of the pull sweep, and the items ((tile, row) pair sums + cell padding) phase 2 streams. */
int gmx_pr_cold_info(gmx_pr_t* p, int64_t* hot_ids, int64_t* cold_edges, int64_t* padded_items);

#ifdef __cplusplus
}
#endif
#endif
