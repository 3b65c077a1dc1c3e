// gmx_pagerank.hip -- the PageRank neighbour-reduction hot loop for gfx950.
//
// Replaces the body of the emitted `pagerank` (source /root/reference/apps/src/pagerank.gm:1-20,
// emission rules SURVEY.md section 8 a-1):
//     for t in nodes:  S = sum_{w in InNbrs(t)} rank[w] / outdeg(w)
//                      val = (1-d)/N + d*S;  diff += |val - rank[t]|;  rank_nxt[t] = val
// Device formulation (same arithmetic per term, fp64 row sums):
//   * contrib[w] = rank[w] / (double) outdeg(w) is produced once per vertex by the row that
//     owns w (the reference recomputes the same quotient once per edge);
//   * vertices are internally renumbered by descending out-degree (the gather frequency of
//     contrib[w] IS outdeg(w)), which packs the hot part of the vector into few cache lines;
//   * work is cut by a MERGE-PATH decomposition of (row ends, edges) into blocks of PRW_ITEMS
//     path items, so hubs and empty rows cost the same;
//   * every WAVE is an independent worker over such blocks (no workgroup barrier in the loop): it
//     streams its slice of r_node_idx with coalesced non-temporal loads, gathers contrib[] (from an
//     LDS tile for the hottest ids, else from L2/HBM), stages the values in its private LDS slice,
//     and every lane walks PRW_PER consecutive path items; rows spanning lanes are closed by a
//     segmented wave scan, rows spanning blocks by fp64 partials that a small fix-up kernel adds in
//     edge order -- no float atomics anywhere, results are run-to-run deterministic;
//   * for graphs whose contribution vector outgrows the caches (GMX_PR_SLICED) the in-edges are split
//     by SOURCE slice, one slice per group of XCDs, so each private L2 (and each CU's LDS tile)
//     only ever serves 1/ns of the vector; per-slice row sums are added by pr_combine_kernel.
// Roofline: HBM-bound; algorithmic bytes per iteration E*(4+s) + V*(8+3s).
#pragma clang fp contract(off)

#include "gmx_internal.h"

#include <math.h>
#include <rocprim/rocprim.hpp>

struct pr_blk { int32_t r, e; };  // merge-path start of a workgroup: local row index, local edge index

#define PR_MAX_SLICES 8
#define PR_COMBINE_GRID 4096
#define PR_MAX_CHUNKS 8
#define PR_QUEUE_STRIDE 64       // work-queue counters live 256 bytes apart (one atomic unit each)
#define PRW_QUEUE_CHUNK 16       // merge-path blocks (512 items) claimed per dequeue by a wave (at most)
#define PR_RUN_SHIFT 11         // sliced numbering: a slice owns runs of 2^11 consecutive ids (all L2 channels)

// Per-slice device arrays of the XCD-sliced variant: slice s holds the in-edges whose SOURCE
// lies in slice s of the contribution vector, slice(id) = (id >> PR_RUN_SHIFT) % ns (runs of 2048 ids, so a
// slice spreads over every L2 channel and no cache line is shared between slices).  Workgroups running on XCD x pull work for slice x % ns, so each
// XCD's private L2 only ever caches 1/ns of the vector.
struct pr_slice_desc {
    const pr_blk* blk;
    int64_t nblk;
    const int32_t* rb;
    const int32_t* ridx;
    double* part_first;
    double* part_last;
    void* partial;   // [active rows of the rank] row sums restricted to this slice (element type S), indexed by
                     // the row's position in active[]; rows the slice has no edge for keep the 0 they were
                     // initialised with (the graph is static)
    const int32_t* rowid;   // compact row -> position in active[]: only rows with an edge in the slice are stored
    int64_t crows;          // number of compact rows
    // per launch (row chunk): blocks [k_lo, k_hi) are reduced, openers [f_lo, f_hi) are fixed up
    int64_t k_lo, k_hi, f_lo, f_hi;
    int qchunk;   // blocks a wave claims per dequeue: <= PRW_QUEUE_CHUNK, smaller when the window is short
    // static part of the schedule: wave w of the slice's st_waves waves takes the claims w, w + st_waves, ...
    // for st_rounds rounds without touching the queue; the queue hands out the rest, from block k_lo + st_total
    int st_rounds, st_waves;
    int64_t st_total;
};
struct pr_sliced_args {
    pr_slice_desc s[PR_MAX_SLICES];
    int ns;
    unsigned int* queue;   // [ns] next unclaimed merge-path block of each slice
};

struct gmx_pr {
    gmx_graph* g = nullptr;
    int elem = 4;
    int rank = 0, nranks = 1;
    uint32_t options = 0;
    int64_t V = 0;        // vertices of the whole graph
    int64_t slice = 0;    // rows per rank (Vpad / nranks)
    int64_t Vpad = 0;     // slice * nranks, size of the contribution replica
    int64_t rows = 0;     // local row ids of this rank (sliced: includes padding rows, outdeg -1)
    int64_t rows_real = 0;  // vertices owned by this rank
    int64_t row_lo = 0;   // first owned row in the internal numbering
    int64_t El = 0;       // edges of the owned rows
    int64_t exchange_count = 0;   // leading entries of a rank's range that other ranks can ever read
    dbuf<int32_t> inv;    // internal id -> original id for owned rows [rows]
    dbuf<int32_t> rb_own, ridx_own;
    const int32_t* rb = nullptr;    // local r_begin' [rows+1]
    const int32_t* ridx = nullptr;  // r_node_idx' (internal source ids) [El]
    dbuf<int32_t> outdeg; // [rows]
    dbuf<char> rk;        // rank of owned rows [rows] x elem
    dbuf<char> contrib[2];  // replicas [Vpad] x elem
    int cur = 0;          // contrib[cur] is read by the next step
    int64_t nblk = 0;
    int items = 0;
    bool hot = false;
    int persistent_grid = 0;
    dbuf<pr_blk> blk;     // [nblk+1]
    dbuf<double> part_first, part_last;   // [nblk] partial row sums leaving a workgroup
    dbuf<double> diff_part;               // [2*nblk] per-workgroup |val-rank| partials (main, fix-up)
    dbuf<double> diff;    // [1]
    double d = 0.85;
    int32_t cnt = 0;
    // XCD-sliced variant
    int ns = 0;
    pr_sliced_args sl;
    dbuf<int32_t> sl_rb, sl_ridx, sl_rowid, sl_active;   // sl_active: local rows with in-edges
    dbuf<uint8_t> sl_is_active;
    int64_t sl_nactive = 0;
    dbuf<int32_t> sl_outdeg_c;   // [nactive] out-degree of active row i (dense copy for the combine pass)
    dbuf<char> sl_rk_c;          // [nactive] x elem: rank of active row i; rk[] holds the rows without in-edges

    // row chunks of a step (1 = whole step at once); tables from pr_chunk_table_kernel
    int nchunks = 1;
    int64_t ch_row[PR_MAX_CHUNKS + 1];                       // boundary rows
    int64_t ch_blk[PR_MAX_CHUNKS + 1][PR_MAX_SLICES];        // first block per slice
    int64_t ch_act[PR_MAX_CHUNKS + 1];                       // first active-row entry
    dbuf<pr_blk> sl_blk;
    dbuf<double> sl_part_first, sl_part_last;
    dbuf<char> sl_partial;
    dbuf<unsigned int> sl_queue;
    int64_t sl_nblk_total = 0;
    // cold sources (local id >= cold_T in their rank range): their edges bypass the pull sweep (gmx_pr_cold.hip)
    pr_cold* cold = nullptr;
    int64_t cold_T = -1;      // hot ids per rank range; -1: no binned part, 0: every edge is binned
    int64_t Eh = 0;           // edges the pull sweep reduces (El minus the cold ones)
    // peer push (exchange over xGMI by the copy engines): the other ranks' replicas mapped into this
    // process, one copy stream per peer
    int step_next = 1;                       // replica the running step writes
    std::vector<char*> peer_buf[2];          // [nranks] base of rank r's contrib[b]; own entry unused
    // two sets of copy streams: the chunks of a step alternate between them, so the small hub piece (sent last, needed
    // first by the peers' next step) does not queue behind the long tail piece
    std::vector<hipStream_t> push_stream[2]; // [nranks]
    std::vector<hipEvent_t> push_done[2];    // [nranks]
    // packed exchange ("send only what is read", nranks > 1 with the degree order): rank q reads source w iff w has an
    // out-edge into a row q owns.  send_list[q]: positions in MY range that rank q reads (ascending); recv_list[r]: positions
    // in rank r's range that I read -- the same list rank r holds as its send_list[me], by construction of both from the
    // same edges.  A pushed chunk is packed per peer into sbuf, copied into the peer's landing zone rbuf[parity] at the
    // offset of my segment there, and scattered into the peer's replica by gmx_pr_unpack.
    bool packed = false;
    std::vector<int64_t> send_cnt, send_off, recv_cnt, recv_off;   // [nranks] elements; offsets into slist / sbuf and rlist / rbuf
    dbuf<int32_t> slist, rlist;              // concatenated send / recv position lists
    dbuf<char> sbuf, rbuf[2];                // packed staging (send) and landing zones (recv, by replica parity)
    std::vector<char*> peer_rbuf[2];         // [nranks] base of rank r's rbuf[b]
    std::vector<int64_t> peer_roff;          // [nranks] where MY segment starts in rank r's landing zone (elements)
    std::vector<int64_t> sbound, rbound;     // [nranks * (nchunks + 1)] chunk boundaries inside the per-peer lists (list indices)
    int bound_chunks = 0;                    // the chunk count sbound / rbound were made for
    int gather_mask = 0;                     // tile classes whose phase 1 has been enqueued for the running step (binned, fused form)
    hipEvent_t push_ready = nullptr;         // "this chunk is computed", recorded on the step's stream
    double* h_diff = nullptr;                // pinned landing place of gmx_pr_diff
    // dominant-kernel timing (hipEvents on the launch stream)
    bool timing = false;
    std::vector<hipEvent_t> ev;   // pairs
    int ev_used = 0;
    ~gmx_pr() {
        for (hipEvent_t e : ev) (void) hipEventDestroy(e);
        for (int q = 0; q < 2; q++) {
            for (hipStream_t st : push_stream[q])
                if (st) {
                    (void) hipStreamSynchronize(st);
                    (void) hipStreamDestroy(st);
                }
            for (hipEvent_t e : push_done[q])
                if (e) (void) hipEventDestroy(e);
        }
        if (push_ready) (void) hipEventDestroy(push_ready);
        if (h_diff) (void) hipHostFree(h_diff);
        pr_cold_free(cold);
    }
};

#define PR_MAX_TIMED 256
static inline void pr_ev_begin(gmx_pr* p, hipStream_t s) {
    if (!p->timing || p->ev_used >= PR_MAX_TIMED) return;
    if ((int) p->ev.size() < 2 * (p->ev_used + 1)) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        p->ev.push_back(a);
        p->ev.push_back(b);
    }
    (void) hipEventRecord(p->ev[2 * p->ev_used], s);
}
static inline void pr_ev_end(gmx_pr* p, hipStream_t s) {
    if (!p->timing || p->ev_used >= PR_MAX_TIMED || (int) p->ev.size() < 2 * (p->ev_used + 1)) return;
    (void) hipEventRecord(p->ev[2 * p->ev_used + 1], s);
    p->ev_used++;
}

static int grid_for(int64_t n, int block = 256, int max_blocks = 256 * 16) {
    int64_t b = (n + block - 1) / block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (int) b;
}

// ------------------------------------------------------------------ plan kernels
__global__ void pr_degkey_kernel(const int32_t* __restrict__ begin, int64_t V,
                                 uint32_t* __restrict__ key, int32_t* __restrict__ id) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < V; i += stride) {
        key[i] = 0x7fffffffu - (uint32_t) (begin[i + 1] - begin[i]);  // descending out-degree
        id[i] = (int32_t) i;
    }
}

// order[j] = original id of the j-th hottest vertex (NULL: identity); deal positions to ranks.
// ns > 0 (XCD-sliced): inside a rank, hotness position q is dealt to slice q % ns and the slices
// are laid out as interleaved runs of 2^PR_RUN_SHIFT ids, so every slice gets the same share
// of hot vertices and its ids stay dense.
__global__ void pr_perm_kernel(const int32_t* __restrict__ order, int64_t V, int64_t slice, int nranks, int ns,
                               int32_t* __restrict__ perm) {
    int64_t j = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; j < V; j += stride) {
        if (!order) { perm[j] = (int32_t) j; continue; }
        int64_t l = j / nranks;
        if (ns > 0) {
            const int64_t sl = l % ns, q = l / ns, run = (int64_t) 1 << PR_RUN_SHIFT;
            l = ((q >> PR_RUN_SHIFT) * ns + sl) * run + (q & (run - 1));
        }
        perm[order[j]] = (int32_t) ((j % nranks) * slice + l);
    }
}

__global__ void pr_owned_kernel(const int32_t* __restrict__ perm, const int32_t* __restrict__ begin, int64_t V,
                                int64_t row_lo, int64_t rows, int32_t* __restrict__ inv, int32_t* __restrict__ outdeg) {
    int64_t v = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; v < V; v += stride) {
        int64_t l = (int64_t) perm[v] - row_lo;
        if (l >= 0 && l < rows) {
            inv[l] = (int32_t) v;
            outdeg[l] = begin[v + 1] - begin[v];
        }
    }
}

// out-degree by internal id (all ranks' ranges; padding ids keep the -1 they were initialised with)
__global__ void pr_deg_by_id_kernel(const int32_t* __restrict__ perm, const int32_t* __restrict__ begin, int64_t V, int32_t* __restrict__ deg) {
    int64_t v = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; v < V; v += stride) deg[perm[v]] = begin[v + 1] - begin[v];
}

// local CSR of the owned rows out of the globally sorted keys (row' << 32 | src')
__global__ void pr_local_csr_kernel(const uint64_t* __restrict__ keys, int64_t E, int64_t row_lo, int64_t rows,
                                    int64_t k_lo, int64_t El, int32_t* __restrict__ rb, int32_t* __restrict__ ridx) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (int64_t e = i; e < El; e += stride) ridx[e] = (int32_t) (uint32_t) (keys[k_lo + e] & 0xffffffffu);
    for (int64_t r = i; r <= rows; r += stride) {
        uint64_t target = (uint64_t) (row_lo + r) << 32;
        int64_t lo = 0, hi = E;
        while (lo < hi) {
            int64_t mid = (lo + hi) >> 1;
            if (keys[mid] < target) lo = mid + 1; else hi = mid;
        }
        rb[r] = (int32_t) (lo - k_lo);
    }
}

// first position of a sorted uint32 array holding a value >= t
__global__ void pr_key32_bound_kernel(const uint32_t* __restrict__ keys, int64_t n, uint32_t t, int64_t* __restrict__ out) {
    if (threadIdx.x || blockIdx.x) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < t) lo = mid + 1; else hi = mid;
    }
    out[0] = lo;
}

__global__ void pr_key_bound_kernel(const uint64_t* __restrict__ keys, int64_t E, uint64_t t0, uint64_t t1,
                                    int64_t* __restrict__ out) {
    if (threadIdx.x > 1 || blockIdx.x) return;
    uint64_t target = threadIdx.x ? t1 : t0;
    int64_t lo = 0, hi = E;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < target) lo = mid + 1; else hi = mid;
    }
    out[threadIdx.x] = lo;
}

// ---- XCD-sliced plan: slice(src') = (src' >> 5) % ns ----
__global__ void pr_slice_key_kernel(const uint64_t* __restrict__ keys, int64_t n, int ns, uint8_t* __restrict__ sl) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < n; i += stride) sl[i] = (uint8_t) ((((uint32_t) keys[i]) >> PR_RUN_SHIFT) % (uint32_t) ns);
}

__global__ void pr_slice_offsets_kernel(const uint8_t* __restrict__ sl_sorted, int64_t n, int ns, int64_t* __restrict__ off) {
    int t = threadIdx.x;
    if (blockIdx.x || t > ns) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((int) sl_sorted[mid] < t) lo = mid + 1; else hi = mid;
    }
    off[t] = lo;
}

// vals = keys grouped by slice (stable).  flag[i] = 1 where position i starts a new (slice,row) pair.
__global__ void pr_slice_flag_kernel(const uint64_t* __restrict__ vals, const uint8_t* __restrict__ sl, int64_t n,
                                     int32_t* __restrict__ flag, int32_t* __restrict__ ridx_all) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const uint64_t kx = vals[i];
        ridx_all[i] = (int32_t) (uint32_t) (kx & 0xffffffffu);
        flag[i] = (i == 0 || sl[i] != sl[i - 1] || (vals[i - 1] >> 32) != (kx >> 32)) ? 1 : 0;
    }
}

// pos = exclusive scan of flag.  Pair p of slice s (global pair index pos[i]) gets
//   rowid[p] = row - row_lo,  rb[p + s] = i - off[s]   (each slice's rb has one extra, terminating entry).
__global__ void pr_slice_pairs_kernel(const uint64_t* __restrict__ vals, const uint8_t* __restrict__ sl,
                                      const int32_t* __restrict__ flag, const int32_t* __restrict__ pos,
                                      const int64_t* __restrict__ off, int64_t n, int64_t row_lo,
                                      int32_t* __restrict__ rowid_all, int32_t* __restrict__ rb_all) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        if (!flag[i]) continue;
        const int s = sl[i];
        const int64_t p = pos[i];
        rowid_all[p] = (int32_t) ((int64_t) (vals[i] >> 32) - row_lo);
        rb_all[p + s] = (int32_t) (i - off[s]);
    }
}

// merge-path split (Merrill & Garland): diagonal k*items over (row ends, edge indices)
__global__ void pr_blocks_kernel(const int32_t* __restrict__ rb, int64_t rows, int64_t El, int items,
                                 int64_t nblk, pr_blk* __restrict__ blk) {
    int64_t k = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (k > nblk) return;
    int64_t total = rows + El;
    int64_t dk = k * (int64_t) items;
    if (dk > total) dk = total;
    int64_t lo = dk > El ? dk - El : 0;
    int64_t hi = dk < rows ? dk : rows;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((int64_t) rb[mid + 1] <= dk - mid - 1) lo = mid + 1; else hi = mid;
    }
    blk[k].r = (int32_t) lo;
    blk[k].e = (int32_t) (dk - lo);
}

// ---- packed exchange lists ----
// rmask[l] bit q: owned source l (position in this rank's range) has an out-edge into a row of rank q.  One thread per
// KFC-sized run of forward edges; a bit already set is not written again (hubs would otherwise serialise on one word).
__global__ void pr_reader_mask_kernel(const int32_t* __restrict__ begin, const int32_t* __restrict__ idx, int64_t V, int64_t E,
                                      const int32_t* __restrict__ perm, int64_t slice, int64_t row_lo, int64_t rows, unsigned int* __restrict__ rmask) {
    constexpr int CH = 16;
    int64_t c = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x, nchunks = (E + CH - 1) / CH;
    for (; c < nchunks; c += stride) {
        const int64_t e0 = c * CH, e1 = e0 + CH < E ? e0 + CH : E;
        int64_t lo = 0, hi = V;
        while (hi - lo > 1) {
            int64_t mid = (lo + hi) >> 1;
            if ((int64_t) begin[mid] <= e0) lo = mid; else hi = mid;
        }
        int64_t row = lo, next = begin[row + 1];
        int64_t l = (int64_t) perm[row] - row_lo;
        unsigned int have = 0, known = 0;
        for (int64_t e = e0; e < e1; e++) {
            while (e >= next) {
                row++;
                next = begin[row + 1];
                l = (int64_t) perm[row] - row_lo;
                known = 0;
            }
            if (l < 0 || l >= rows) continue;
            if (!known) { have = rmask[l]; known = 1; }
            const unsigned int bit = 1u << (unsigned) (perm[idx[e]] / slice);
            if (!(have & bit)) { atomicOr(&rmask[l], bit); have |= bit; }
        }
    }
}
// mark[src] = 1 for every source among the keys (row << 32 | source)
__global__ void pr_mark_sources_kernel(const uint64_t* __restrict__ keys, int64_t n, uint8_t* __restrict__ mark) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < n; i += stride) mark[(uint32_t) keys[i]] = 1;
}
struct pr_mask_has_bit {
    const unsigned int* rmask;
    unsigned int bit;
    __device__ bool operator()(const int32_t& l) const { return (rmask[l] & bit) != 0; }
};
struct pr_is_marked {
    const uint8_t* mark;   // already offset to the range
    __device__ bool operator()(const int32_t& l) const { return mark[l] != 0; }
};
// first index of a sorted int32 list holding a value >= t, for a few (list, t) pairs: q[3 i] = list offset, q[3 i + 1] = length, q[3 i + 2] = t
__global__ void pr_list_bounds_kernel(const int32_t* __restrict__ list, const int64_t* __restrict__ q, int nq, int64_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const int32_t* a = list + q[3 * i];
    int64_t lo = 0, hi = q[3 * i + 1];
    const int64_t t = q[3 * i + 2];
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t) a[mid] < t) lo = mid + 1; else hi = mid;
    }
    out[i] = lo;
}
// sbuf[off[q] + i] = own[list[off[q] + i]] for i in [lo[q], hi[q]) of every peer q (blockIdx.y)
struct pr_pack_args { int64_t off[16], lo[16], hi[16]; };
template <typename S>
__global__ void pr_pack_kernel(pr_pack_args a, const int32_t* __restrict__ list, const S* __restrict__ own, S* __restrict__ out) {
    const int q = blockIdx.y;
    const int64_t n = a.hi[q] - a.lo[q], base = a.off[q] + a.lo[q];
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[base + i] = own[list[base + i]];
}
// replica[r * slice + list[off[r] + i]] = in[off[r] + i] for i in [lo[r], hi[r]) of every sender r (blockIdx.y)
template <typename S>
__global__ void pr_unpack_kernel(pr_pack_args a, const int32_t* __restrict__ list, const S* __restrict__ in, S* __restrict__ replica, int64_t slice) {
    const int r = blockIdx.y;
    const int64_t n = a.hi[r] - a.lo[r], base = a.off[r] + a.lo[r];
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < n; i += stride) replica[(int64_t) r * slice + list[base + i]] = in[base + i];
}

// selects the in-edges of the owned rows out of (row << 32 | source) keys
struct pr_row_in_range {
    uint64_t lo, hi;
    __device__ bool operator()(const uint64_t& k) const { return k >= lo && k < hi; }
};

// selects the in-edges the pull sweep keeps: source inside the hot prefix of its rank range
struct pr_is_hot_key {
    uint32_t slice, T;
    __device__ bool operator()(const uint64_t& k) const { return ((uint32_t) k) % slice < T; }
};

// ------------------------------------------------------------------ hot loop
template <typename S>
__device__ __forceinline__ void pr_finalize(int64_t r, double sum, double base, double d,
                                            S* __restrict__ rk, const int32_t* __restrict__ outdeg,
                                            S* __restrict__ contrib_next_owned, double& diff_acc) {
    // streamed once per iteration: non-temporal, so the caches keep the gathered contrib lines
    const int32_t od = __builtin_nontemporal_load(outdeg + r);
    if (od < 0) return;   // padding row of the sliced numbering: no vertex lives here
    double val = base + d * sum;
    double old = (double) __builtin_nontemporal_load(rk + r);
    S vs = (S) val;
    diff_acc += fabs((double) vs - old);
    __builtin_nontemporal_store(vs, rk + r);
    __builtin_nontemporal_store(od > 0 ? (S) ((double) vs / (double) od) : (S) 0, contrib_next_owned + r);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Output policies of the row reduction: either apply the PageRank update directly, or leave
// the per-slice row sum for the combine pass.
template <typename S>
struct out_final {
    S* rk;
    const int32_t* outdeg;
    S* next_owned;
    double base, d;
    double* part_first;
    double* part_last;
    __device__ __forceinline__ void row(int64_t r, double sum, double& diff_acc) const {
        pr_finalize<S>(r, sum, base, d, rk, outdeg, next_owned, diff_acc);
    }
};
template <typename S>
struct out_partial {
    S* partial;
    const int32_t* rowid;
    double* part_first;
    double* part_last;
    __device__ __forceinline__ void row(int64_t r, double sum, double&) const { partial[rowid[r]] = (S) sum; }
};

template <int THREADS>
__device__ __forceinline__ void pr_block_diff(double diff_acc, double* s_red, double* __restrict__ dst) {
    constexpr int NW = THREADS / 64;
    const int tid = threadIdx.x;
    diff_acc = wave_sum(diff_acc);
    if ((tid & 63) == 0) s_red[tid >> 6] = diff_acc;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < NW; w++) t += s_red[w];
        *dst = t;
    }
}

__device__ __forceinline__ int pr_xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return (int) (v & 0xf);
}

// ======================================================================================
// Wave-worker formulation: every WAVE is an independent worker that walks
// merge-path blocks of PRW_ITEMS items; there is no workgroup barrier in the loop, the row
// starts / outdeg / old rank of a block live in registers, the next block's index slice is
// prefetched while the current block is reduced, and the per-wave |val-rank| partial stays in a
// register until the end.  A workgroup only shares the optional LDS tile of hot contributions.
// ======================================================================================
#define PRW_ITEMS 512
#define PRW_PER (PRW_ITEMS / 64)
#define PRW_ROWU ((PRW_ITEMS + 1 + 63) / 64)

#define PRW_PAD(j) ((j) + ((j) >> 5))   // one pad word per 32: stride-8 lane accesses hit 32 different banks
// Debug build (make debug: -DGMX_PR_BOUNDS): every index the wave workers derive from the block table is checked
// against what the block may touch, a violation traps -- the launch dies at the faulty access instead of reading a
// neighbour's page (tests/test_gpu_parity.py::test_pagerank_small_shapes_with_bounds_checks).
#ifdef GMX_PR_BOUNDS
#define PRW_CHECK(cond) do { if (!(cond)) __builtin_trap(); } while (0)
#else
#define PRW_CHECK(cond) do { } while (0)
#endif
template <typename S>
struct prw_lds {
    S val[PRW_ITEMS + PRW_ITEMS / 32 + 2];
    int32_t rb[PRW_ITEMS + 2];
};

__device__ __forceinline__ double wave_allsum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <typename S>
struct prw_final {
    S* rk;
    S* next_owned;
    double base, d;
    double* part_first;
    double* part_last;
    static constexpr bool needs_vertex_data = true;
    __device__ __forceinline__ int32_t preload(int64_t) const { return 0; }
    __device__ __forceinline__ void row(int64_t r, double sum, int32_t od, S old, double& diff_acc) const {
        if (od < 0) return;   // padding row
        double val = base + d * sum;
        S vs = (S) val;
        diff_acc += fabs((double) vs - (double) old);
        __builtin_nontemporal_store(vs, rk + r);
        __builtin_nontemporal_store(od > 0 ? (S) ((double) vs / (double) od) : (S) 0, next_owned + r);
    }
};
template <typename S>
struct prw_partial {
    S* partial;
    const int32_t* rowid;
    double* part_first;
    double* part_last;
    static constexpr bool needs_vertex_data = false;
    // `rid` = rowid[r], preloaded with the row starts (no dependent load inside the reduction loop)
    __device__ __forceinline__ void row(int64_t, double sum, int32_t rid, S, double&) const { partial[rid] = (S) sum; }
    __device__ __forceinline__ int32_t preload(int64_t r) const { return __builtin_nontemporal_load(rowid + r); }
};

// A wave keeps THREE blocks in flight (software pipeline, no barrier anywhere):
//   stage A  prw_load_idx : the index slice of block k+2 is loaded into registers;
//   stage B  prw_issue    : for block k+1 (indices already there) the row starts, the per-row data of the
//                           output policy and the contribution gathers are issued into registers;
//   stage C  prw_consume  : block k (everything already in registers) is staged in the wave's LDS slice,
//                           reduced and written out.
// So the gathers of the next block and the index stream of the one after are always outstanding while a
// block is being reduced.
// How ids map to slots of the LDS tile.  Unsliced: slot = id.  Sliced: the tile holds, for every rank's
// range (1 << rank_shift ids; rank_shift == 0: a single range), the per_rank hottest ids of the home slice;
// an id's slice bits sit above the run offset and are squeezed out.
struct prw_tile {
    int lim;         // usable slots (0: no tile for this block)
    int sliced;      // 0: slot = id
    int sl_shift;    // log2(number of slices)
    int rank_shift;  // log2(ids per rank range), 0 = one range
    int per_rank;    // tile slots per rank range
    __device__ __forceinline__ int slot(int32_t id) const {
        if (!sliced) return id < lim ? id : -1;
        const int local = rank_shift ? (id & ((1 << rank_shift) - 1)) : id;
        const int q = ((local >> (PR_RUN_SHIFT + sl_shift)) << PR_RUN_SHIFT) | (local & ((1 << PR_RUN_SHIFT) - 1));
        if (q >= per_rank) return -1;
        return (rank_shift ? (id >> rank_shift) * per_rank : 0) + q;
    }
};

template <typename S>
struct prw_regs {
    int32_t rbv[PRW_ROWU];   // row starts
    int32_t od[PRW_ROWU];    // out-degree (final) / compact row id (partial) of the rows that finish in the block
    S old[PRW_ROWU];         // old rank (final only)
    S vv[PRW_PER];           // gathered contributions
};

template <bool NT>
__device__ __forceinline__ void prw_load_idx(const int32_t* __restrict__ ridx, int e0, int ne, int32_t (&ix)[PRW_PER]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int u = 0; u < PRW_PER; u++) {
        const int j = lane + 64 * u;
        ix[u] = -1;
        if (j < ne) ix[u] = NT ? __builtin_nontemporal_load(ridx + e0 + j) : ridx[e0 + j];
    }
}

template <typename S, typename OUT>
__device__ __forceinline__ void prw_issue(const S* __restrict__ s_hot, const prw_tile tile,
                                          pr_blk b0, pr_blk b1,
                                          const int32_t* __restrict__ rb, const int32_t* __restrict__ outdeg,
                                          const S* __restrict__ rk_old, const S* __restrict__ contrib,
                                          const int32_t (&ix)[PRW_PER], const OUT& out, prw_regs<S>& g) {
    const int lane = threadIdx.x & 63;
    const int r0 = b0.r, nr = b1.r - b0.r + 1;
    // a block is PRW_ITEMS consecutive items of the merge path: row ends [b0.r, b1.r) and edges [b0.e, b1.e)
    PRW_CHECK(b0.r >= 0 && b1.r >= b0.r && b1.e >= b0.e && (b1.r - b0.r) + (b1.e - b0.e) <= PRW_ITEMS);
#pragma unroll
    for (int u = 0; u < PRW_ROWU; u++) {
        const int i = lane + 64 * u;
        g.rbv[u] = 0;
        g.od[u] = -1;
        g.old[u] = (S) 0;
        if (i < nr) {
            g.rbv[u] = __builtin_nontemporal_load(rb + r0 + i);
            if (i < nr - 1) {
                if (OUT::needs_vertex_data) {
                    g.od[u] = __builtin_nontemporal_load(outdeg + r0 + i);
                    g.old[u] = __builtin_nontemporal_load(rk_old + r0 + i);
                } else g.od[u] = out.preload((int64_t) r0 + i);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < PRW_PER; u++) {
        g.vv[u] = (S) 0;
        const int32_t id = ix[u];
        if (id >= 0) {
            const int slot = tile.lim > 0 ? tile.slot(id) : -1;
            if (slot >= 0) g.vv[u] = s_hot[slot];
            else g.vv[u] = contrib[id];
        }
    }
}

// Row reduction: every lane walks exactly PRW_PER consecutive items of the block's merge path (edges
// and row ends), so there is no divergence and no per-row loop; rows that span lanes are closed by a
// wave-wide segmented scan of the lanes' open sums.  A finished row's sum is parked in the LDS slot of
// the row's last edge (already consumed, hence free) and written out one lane per row, coalesced.
template <typename S, typename OUT>
__device__ __forceinline__ void prw_consume(prw_lds<S>* __restrict__ w, int64_t k, pr_blk b0, pr_blk b1, int64_t rows,
                                            const prw_regs<S>& g, const OUT& out, double& diff_acc) {
    const int lane = threadIdx.x & 63;
    const int r0 = b0.r, e0 = b0.e, r1 = b1.r, e1 = b1.e;
    const int ne = e1 - e0;
    const int nr = r1 - r0 + 1;
    const int nends = nr - 1;          // row-end items of this block
    const int total = nends + ne;      // path items of this block
    PRW_CHECK(k >= 0 && ne >= 0 && ne <= PRW_ITEMS && nr >= 1 && nr <= PRW_ITEMS + 1 && total <= PRW_ITEMS);
    PRW_CHECK(r1 <= rows);             // rb[r1] is the last row start a block reads: rb has rows + 1 entries
#pragma unroll
    for (int u = 0; u < PRW_ROWU; u++) {
        const int i = lane + 64 * u;
        if (i < nr) w->rb[i] = g.rbv[u];
    }
#pragma unroll
    for (int u = 0; u < PRW_PER; u++) {
        const int j = lane + 64 * u;
        if (j < ne) {
            PRW_CHECK(PRW_PAD(j) < (int) (sizeof(w->val) / sizeof(w->val[0])));
            w->val[PRW_PAD(j)] = g.vv[u];
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes are visible to all its lanes
    __builtin_amdgcn_wave_barrier();

    const bool first_started_here = (w->rb[0] >= e0);
    // ---- this lane's start on the merge path: i row ends and j edges consumed before item d0 ----
    const int d0 = lane * PRW_PER;
    const bool active = d0 < total;
    int i = 0, j = 0;
    if (active) {
        int lo = d0 > ne ? d0 - ne : 0, hi = d0 < nends ? d0 : nends;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            PRW_CHECK(mid + 1 < nr);
            if (w->rb[mid + 1] - e0 <= d0 - mid - 1) lo = mid + 1; else hi = mid;
        }
        i = lo;
        j = d0 - lo;
        PRW_CHECK(i >= 0 && i <= nends && j >= 0 && j <= ne);
    }
    double acc = 0.0, head = 0.0;
    int first_end = -1;
    int row_start = j;                                  // first in-lane edge of the row being summed
    int cur_end = (active && i < nends) ? w->rb[i + 1] - e0 : 0x7fffffff;
#pragma unroll
    for (int sidx = 0; sidx < PRW_PER; sidx++) {
        if (active && d0 + sidx < total) {
            if (j < cur_end) {
                PRW_CHECK(j < ne);
                acc += (double) w->val[PRW_PAD(j)];
                j++;
            } else {                                    // row i ends here
                if (first_end < 0) {
                    head = acc;
                    first_end = i;
                } else if (j > row_start) w->val[PRW_PAD(j - 1)] = (S) acc;   // row lies inside this lane
                acc = 0.0;
                i++;
                row_start = j;
                cur_end = (i < nends) ? w->rb[i + 1] - e0 : 0x7fffffff;
            }
        }
    }
    // ---- segmented inclusive scan over lanes of the open sums (a lane with a row end restarts the segment) ----
    double v = acc;
    int f = first_end >= 0 ? 1 : 0;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double vo = __shfl_up(v, off, 64);
        const int fo = __shfl_up(f, off, 64);
        if (lane >= off) {
            if (!f) v += vo;
            f |= fo;
        }
    }
    double carry_in = __shfl_up(v, 1, 64);
    if (lane == 0) carry_in = 0.0;
    const double block_tail = __shfl(v, 63, 64);       // open sum of row r1 at the end of the block
    if (first_end >= 0) {
        const int fe_end = w->rb[first_end + 1] - e0;
        int fs = w->rb[first_end] - e0;
        if (fs < 0) fs = 0;
        if (fe_end > fs) w->val[PRW_PAD(fe_end - 1)] = (S) (head + carry_in);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();

    // ---- coalesced output: lane per row ----
#pragma unroll
    for (int u = 0; u < PRW_ROWU; u++) {
        if (64 * u >= nr) break;
        const int ri = lane + 64 * u;
        if (ri < nr) {
            int lo = g.rbv[u] - e0;
            if (lo < 0) lo = 0;
            const bool started = (ri > 0) || first_started_here;
            if (ri < nr - 1) {
                const int hi = w->rb[ri + 1] - e0;
                const double sum = hi > lo ? (double) w->val[PRW_PAD(hi - 1)] : 0.0;
                if (!started) out.part_first[k] = sum;
                else out.row((int64_t) r0 + ri, sum, g.od[u], g.old[u], diff_acc);
            } else {
                if (!started) out.part_first[k] = block_tail;
                else if (r0 + ri < rows && ne > lo) out.part_last[k] = block_tail;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// Unsliced: wave w of the grid handles blocks w, w + W, w + 2W, ... (static, deterministic).
template <typename S, int WAVES, int HOT, bool NT>
__global__ void __launch_bounds__(WAVES * 64)
pr_wave_kernel(const pr_blk* __restrict__ blk, int64_t nblk, int64_t rows,
               const int32_t* __restrict__ rb, const int32_t* __restrict__ ridx,
               const int32_t* __restrict__ outdeg, S* __restrict__ rk,
               const S* __restrict__ contrib, S* __restrict__ contrib_next_owned,
               double base, double d,
               double* __restrict__ part_first, double* __restrict__ part_last, double* __restrict__ diff_part,
               int64_t ncontrib) {
    __shared__ prw_lds<S> lds[WAVES];
    __shared__ S s_hot[HOT > 0 ? HOT : 1];
    if (HOT > 0) {
        for (int i = threadIdx.x; i < HOT; i += WAVES * 64) s_hot[i] = i < ncontrib ? contrib[i] : (S) 0;
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t W = (int64_t) gridDim.x * WAVES;
    prw_lds<S>* w = &lds[wv];
    prw_final<S> out{rk, contrib_next_owned, base, d, part_first, part_last};
    double diff_acc = 0.0;

    // pipeline registers: C = block being reduced, B = block whose loads are in flight, A = index slice ahead
    int64_t kC = (int64_t) blockIdx.x * WAVES + wv, kB = kC + W;
    pr_blk c0 = {0, 0}, c1 = {0, 0}, b0 = {0, 0}, b1 = {0, 0};
    prw_regs<S> gC, gB;
    int32_t ixB[PRW_PER], ixA[PRW_PER];
    if (kC < nblk) {
        c0 = blk[kC];
        c1 = blk[kC + 1];
        prw_load_idx<NT>(ridx, c0.e, c1.e - c0.e, ixB);
        prw_issue<S>(s_hot, prw_tile{HOT, 0, 0, 0, HOT}, c0, c1, rb, outdeg, rk, contrib, ixB, out, gC);
    }
    if (kB < nblk) {
        b0 = blk[kB];
        b1 = blk[kB + 1];
        prw_load_idx<NT>(ridx, b0.e, b1.e - b0.e, ixB);
    }
    while (kC < nblk) {
        const int64_t kA = kB + W;
        pr_blk a0 = {0, 0}, a1 = {0, 0};
        if (kA < nblk) {
            a0 = blk[kA];
            a1 = blk[kA + 1];
            prw_load_idx<NT>(ridx, a0.e, a1.e - a0.e, ixA);
        }
        if (kB < nblk) prw_issue<S>(s_hot, prw_tile{HOT, 0, 0, 0, HOT}, b0, b1, rb, outdeg, rk, contrib, ixB, out, gB);
        prw_consume<S>(w, kC, c0, c1, rows, gC, out, diff_acc);
        kC = kB;
        c0 = b0;
        c1 = b1;
        gC = gB;
        kB = kA;
        b0 = a0;
        b1 = a1;
#pragma unroll
        for (int u = 0; u < PRW_PER; u++) ixB[u] = ixA[u];
    }
    diff_acc = wave_sum(diff_acc);
    if (lane == 0) diff_part[(int64_t) blockIdx.x * WAVES + wv] = diff_acc;
}

// Sliced: waves on XCD x claim chunks of slice x % ns (stealing from the others when drained).
template <typename S, int WAVES, int HOT, bool NT>
__global__ void __launch_bounds__(WAVES * 64)
pr_wave_sliced_kernel(pr_sliced_args a, int64_t ncontrib, const S* __restrict__ contrib, int ns_shift, int rank_shift, int nranks) {
    __shared__ prw_lds<S> lds[WAVES];
    __shared__ S s_hot[HOT > 0 ? HOT : 1];
    const int home = pr_xcc_id() % a.ns;
    const int per_rank = HOT / (nranks > 0 ? nranks : 1);
    __shared__ int s_drained[PR_MAX_SLICES];   // slice queues this workgroup has seen empty
    if (threadIdx.x < PR_MAX_SLICES) s_drained[threadIdx.x] = 0;
    if (HOT > 0) {
        // slot (r, q) <-> id (r << rank_shift) + (((q >> RUN) * ns + home) << RUN | (q & (RUN-1))):
        // the per_rank hottest ids of the home slice inside every rank's range
        for (int r = 0; r < nranks; r++) {
            const int64_t id0 = (int64_t) r << rank_shift;
            S* dst = s_hot + r * per_rank;
#pragma unroll 4
            for (int q = threadIdx.x; q < per_rank; q += WAVES * 64) {
                const int64_t id = id0 + (((((int64_t) q >> PR_RUN_SHIFT) * a.ns + home) << PR_RUN_SHIFT) | (q & ((1 << PR_RUN_SHIFT) - 1)));
                dst[q] = id < ncontrib ? contrib[id] : (S) 0;
            }
        }
    }
    __syncthreads();
    const prw_tile tile_home{HOT > 0 ? per_rank * nranks : 0, 1, ns_shift, nranks > 1 ? rank_shift : 0, per_rank};
    const prw_tile tile_none{0, 1, ns_shift, 0, 0};
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    prw_lds<S>* w = &lds[wv];
    double unused = 0.0;

    // Work distribution.  Merge-path blocks are equal work by construction, so most of a slice's window is
    // dealt out statically (interleaved over the waves the launch gives that slice); only the tail goes
    // through the per-slice queue, which also lets drained waves steal from other slices.  The queue
    // counters are the scarce resource: atomics on one address retire at ~90 per microsecond, device wide.
    // (wave-uniform state; lane 0 performs the atomic, the result is broadcast)
    long long k_next = 0, k_end = 0;
    int cur = home, first_try = 0;
    // the static deal is keyed by blockIdx alone (unique by construction); workgroups go round-robin over
    // the 8 XCDs, so this is the home slice
    const int st_slice = (int) ((blockIdx.x & 7) % a.ns);
    const int st_qc = a.s[st_slice].qchunk;
    // this wave's static claims: st_next, st_next + st_stride, ... (st_left of them)
    long long st_next = a.s[st_slice].k_lo +
                        (long long) ((((blockIdx.x >> 3) * (8 / a.ns) + (blockIdx.x & 7) / a.ns) * WAVES + wv)) * st_qc;
    const long long st_stride = (long long) a.s[st_slice].st_waves * st_qc;
    int st_left = a.s[st_slice].st_rounds;
    auto claim = [&](long long& k_out, int& sl_out) {
        long long k = -1;
        if (k_next < k_end) k = k_next++;
        else if (st_left > 0) {
            k = st_next;
            st_next += st_stride;
            st_left--;
            k_next = k + 1;
            k_end = k + st_qc;   // static claims are whole and inside the window
            cur = st_slice;
        } else {
            for (int t = first_try; t < a.ns; t++) {
                cur = (home + t) % a.ns;
                first_try = t + 1;
                // a queue some wave of this workgroup has seen empty costs an LDS read, not another atomic
                if (__builtin_amdgcn_readfirstlane(*(volatile int*) &s_drained[cur])) continue;   // (scalar: keeps the loop uniform)
                const long long nb = a.s[cur].k_hi;
                const int qc = a.s[cur].qchunk;
                unsigned int kk0 = 0;
                if (lane == 0) kk0 = atomicAdd(&a.queue[cur * PR_QUEUE_STRIDE], (unsigned int) qc);
                const long long kk = a.s[cur].k_lo + a.s[cur].st_total + (long long) (unsigned int) __builtin_amdgcn_readfirstlane((int) kk0);
                if (kk < nb) {
                    k = kk;
                    k_next = kk + 1;
                    k_end = kk + qc < nb ? kk + qc : nb;
                    first_try = t;
                    break;
                }
                if (lane == 0) *(volatile int*) &s_drained[cur] = 1;
            }
        }
        k_out = k;
        sl_out = cur;
    };
    auto policy = [&](int sl) {
        const pr_slice_desc& sd = a.s[sl];
        return prw_partial<S>{(S*) sd.partial, sd.rowid, sd.part_first, sd.part_last};
    };

    // pipeline registers: C = block being reduced, B = block whose loads are in flight, A = index slice ahead
    long long kC, kB, kA;
    int slC, slB, slA;
    pr_blk c0 = {0, 0}, c1 = {0, 0}, b0 = {0, 0}, b1 = {0, 0};
    prw_regs<S> gC, gB;
    int32_t ixB[PRW_PER], ixA[PRW_PER];
    claim(kC, slC);
    if (kC >= 0) {
        c0 = a.s[slC].blk[kC];
        c1 = a.s[slC].blk[kC + 1];
        prw_load_idx<NT>(a.s[slC].ridx, c0.e, c1.e - c0.e, ixB);
        prw_issue<S>(s_hot, slC == home ? tile_home : tile_none, c0, c1, a.s[slC].rb,
                     (const int32_t*) nullptr, (const S*) nullptr, contrib, ixB, policy(slC), gC);
        claim(kB, slB);
    } else {
        kB = -1;
        slB = home;
    }
    if (kB >= 0) {
        b0 = a.s[slB].blk[kB];
        b1 = a.s[slB].blk[kB + 1];
        prw_load_idx<NT>(a.s[slB].ridx, b0.e, b1.e - b0.e, ixB);
    }
    while (kC >= 0) {
        pr_blk a0 = {0, 0}, a1 = {0, 0};
        kA = -1;
        slA = home;
        if (kB >= 0) {
            claim(kA, slA);
            if (kA >= 0) {
                a0 = a.s[slA].blk[kA];
                a1 = a.s[slA].blk[kA + 1];
                prw_load_idx<NT>(a.s[slA].ridx, a0.e, a1.e - a0.e, ixA);
            }
            prw_issue<S>(s_hot, slB == home ? tile_home : tile_none, b0, b1, a.s[slB].rb,
                         (const int32_t*) nullptr, (const S*) nullptr, contrib, ixB, policy(slB), gB);
        }
        prw_consume<S>(w, (int64_t) kC, c0, c1, a.s[slC].crows, gC, policy(slC), unused);
        kC = kB;
        slC = slB;
        c0 = b0;
        c1 = b1;
        gC = gB;
        kB = kA;
        slB = slA;
        b0 = a0;
        b1 = a1;
#pragma unroll
        for (int u = 0; u < PRW_PER; u++) ixB[u] = ixA[u];
    }
}

// Rows that span workgroups: the workgroup that OPENED row r (r == blk[k+1].r, rb[r] in
// [e0,e1)) left part_last[k]; every later workgroup touching r (blk[k'].r == r) left part_first[k'].
// One thread per opener adds the first few pieces itself; a row that goes on (a hub: thousands of blocks)
// is summed by the whole wave, lanes striding over part_first in a fixed order.
#define PR_FIX_SERIAL 16
template <typename S, typename OUT>
__device__ __forceinline__ void pr_fixup_one(const pr_blk* __restrict__ blk, int64_t nblk, int64_t rows,
                                             const int32_t* __restrict__ rb, int64_t k, bool valid, const OUT& out, double& diff_acc) {
    const int lane = threadIdx.x & 63;
    bool opener = false, pending = false;
    long long r = 0, kk = 0;
    double total = 0.0;
    if (valid) {
        const pr_blk b0 = blk[k], b1 = blk[k + 1];
        r = b1.r;
        if (r < rows) {
            const int32_t rs = rb[r];
            opener = rs >= b0.e && rs < b1.e;
        }
        if (opener) {
            total = out.part_last[k];
            pending = true;
            for (kk = k + 1; kk < nblk && kk <= k + PR_FIX_SERIAL; kk++) {
                total += out.part_first[kk];
                if (blk[kk + 1].r > r) {
                    pending = false;
                    break;
                }
            }
            if (kk >= nblk) pending = false;
        }
    }
    unsigned long long m = __ballot(pending);
    while (m) {   // whole wave: blocks kk, kk+1, ... of lane src's row, 64 per round, until one starts in a later row
        const int src = __ffsll((long long) m) - 1;
        m &= m - 1;
        const long long r0 = __shfl(r, src, 64);
        long long q = __shfl(kk, src, 64) + lane;
        double t = 0.0;
        for (;; q += 64) {
            const bool in = q < nblk && blk[q].r == r0;
            if (in) t += out.part_first[q];
            if (__ballot(in) != ~0ull) break;
        }
        t = __shfl(wave_sum(t), 0, 64);
        if (lane == src) total += t;
    }
    if (opener) out.row(r, total, diff_acc);
}

template <typename S>
__global__ void __launch_bounds__(256)
pr_fixup_kernel(const pr_blk* __restrict__ blk, int64_t nblk, int64_t rows,
                const int32_t* __restrict__ rb, const int32_t* __restrict__ outdeg,
                S* __restrict__ rk, S* __restrict__ contrib_next_owned, double base, double d,
                double* __restrict__ part_first, double* __restrict__ part_last,
                const double* __restrict__ diff_main, int64_t n_main, double* __restrict__ diff_out) {
    __shared__ double s_red[256 / 64];
    int64_t k = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    double diff_acc = 0.0;
    out_final<S> out{rk, outdeg, contrib_next_owned, base, d, part_first, part_last};
    pr_fixup_one<S>(blk, nblk, rows, rb, k, k < nblk, out, diff_acc);
    if (k < n_main) diff_acc += diff_main[k];   // fold in the main kernel's partials (fixed order)
    pr_block_diff<256>(diff_acc, s_red, diff_out + blockIdx.x);
}

template <typename S>
__global__ void pr_sliced_fixup_kernel(pr_sliced_args a, int64_t rows) {
    const int sl = blockIdx.y;
    const pr_slice_desc& sd = a.s[sl];
    int64_t k = sd.f_lo + (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    double unused = 0.0;
    out_partial<S> out{(S*) sd.partial, sd.rowid, sd.part_first, sd.part_last};
    pr_fixup_one<S>(sd.blk, sd.nblk, sd.crows, sd.rb, k, k < sd.f_hi, out, unused);
}

// Sum the slices in fixed order and apply the PageRank update.  Only rows that have in-edges are
// visited (active[]): a row without in-edges gets (1-d)/N in the first sweep and never changes again
// (SURVEY.md appendix A), so pr_inactive_first_kernel settles those once, right after a reset.
// Everything the pass streams is indexed by the position i in active[] (partial sums, out-degree, rank), so
// it reads whole lines of useful data although only ~40 % of the rows are active; the one row-indexed access
// is the store of the new contribution.
template <typename S>
__global__ void __launch_bounds__(256)
pr_combine_kernel(pr_sliced_args a, const int32_t* __restrict__ active, int64_t i_lo, int64_t i_hi,
                  const int32_t* __restrict__ outdeg_c, S* __restrict__ rk_c, const S* __restrict__ cold,
                  S* __restrict__ contrib_next_owned, double base, double d, double* __restrict__ diff_part) {
    __shared__ double s_red[256 / 64];
    double diff_acc = 0.0;
    int64_t i = i_lo + (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < i_hi; i += stride) {
        const int64_t r = __builtin_nontemporal_load(active + i);
        double sum = 0.0;
        for (int sl = 0; sl < a.ns; sl++) sum += (double) __builtin_nontemporal_load((const S*) a.s[sl].partial + i);
        if (cold) sum += (double) __builtin_nontemporal_load(cold + i);   // the cold sources' part, after the slices
        const int32_t od = __builtin_nontemporal_load(outdeg_c + i);
        const double val = base + d * sum;
        const double old = (double) __builtin_nontemporal_load(rk_c + i);
        const S vs = (S) val;
        diff_acc += fabs((double) vs - old);
        __builtin_nontemporal_store(vs, rk_c + i);
        __builtin_nontemporal_store(od > 0 ? (S) ((double) vs / (double) od) : (S) 0, contrib_next_owned + r);
    }
    pr_block_diff<256>(diff_acc, s_red, diff_part + blockIdx.x);
    // the sweep of this chunk is over: leave the work queues at zero for the next launch (saves a memset)
    if (blockIdx.x == 0 && threadIdx.x < PR_MAX_SLICES) a.queue[threadIdx.x * PR_QUEUE_STRIDE] = 0;
}

// Plan: position of every active row in active[]; the slices' compact rows are renamed to it.
__global__ void pr_active_index_kernel(const int32_t* __restrict__ active, int64_t nactive, int32_t* __restrict__ index_of_row) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < nactive; i += stride) index_of_row[active[i]] = (int32_t) i;
}
__global__ void pr_remap_kernel(int32_t* __restrict__ ids, int64_t n, const int32_t* __restrict__ map) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < n; i += stride) ids[i] = map[ids[i]];
}
__global__ void pr_gather_i32_kernel(const int32_t* __restrict__ active, int64_t nactive, const int32_t* __restrict__ src, int32_t* __restrict__ dst) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < nactive; i += stride) dst[i] = src[active[i]];
}
// rank of the active rows: dense <-> row order (reset / download)
template <typename S>
__global__ void pr_rk_dense_kernel(const int32_t* __restrict__ active, int64_t nactive, S* __restrict__ rk, S* __restrict__ rk_c, int to_rows) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < nactive; i += stride) {
        if (to_rows) rk[active[i]] = rk_c[i];
        else rk_c[i] = rk[active[i]];
    }
}

// First sweep after a reset: rows with no in-edges get their (from now on constant) rank and contribution.
// Later sweeps only rewrite the active rows of the replica they produce, so the value has to end up in BOTH
// replicas: each row chunk writes the replica being produced (its piece may travel right away), and
// pr_inactive_copy_kernel patches the replica being read once the whole sweep has finished reading it.
template <typename S>
__global__ void __launch_bounds__(256)
pr_inactive_first_kernel(const uint8_t* __restrict__ is_active, int64_t r_lo, int64_t r_hi, const int32_t* __restrict__ outdeg,
                         S* __restrict__ rk, S* __restrict__ contrib_next_owned,
                         double base, double d, double* __restrict__ diff_part) {
    __shared__ double s_red[256 / 64];
    double diff_acc = 0.0;
    int64_t r = r_lo + (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; r < r_hi; r += stride) {
        if (is_active[r]) continue;
        pr_finalize<S>(r, 0.0, base, d, rk, outdeg, contrib_next_owned, diff_acc);
    }
    pr_block_diff<256>(diff_acc, s_red, diff_part + blockIdx.x);
}

template <typename S>
__global__ void __launch_bounds__(256)
pr_inactive_copy_kernel(const uint8_t* __restrict__ is_active, int64_t rows, S* __restrict__ contrib_cur_owned,
                        const S* __restrict__ contrib_next_owned) {
    int64_t r = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; r < rows; r += stride)
        if (!is_active[r]) contrib_cur_owned[r] = contrib_next_owned[r];
}

// Row chunks: for boundary row B, per slice the merge-path block that holds the first path item of the first
// compact row >= B (rows >= B lie in that block and later ones, rows < B in that block and earlier ones),
// and the first entry of active[] that is >= B.  out[(ns + 1) * c + sl], last column = active.
__global__ void pr_chunk_table_kernel(pr_sliced_args a, const int32_t* __restrict__ active, int64_t nactive,
                                      const int64_t* __restrict__ bounds, int nb, int items, int64_t* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nb * (a.ns + 1)) return;
    const int c = t / (a.ns + 1), sl = t % (a.ns + 1);
    const int64_t B = bounds[c];
    if (sl == a.ns) {
        int64_t lo = 0, hi = nactive;
        while (lo < hi) {
            int64_t mid = (lo + hi) >> 1;
            if (active[mid] < B) lo = mid + 1; else hi = mid;
        }
        out[t] = lo;
        return;
    }
    const pr_slice_desc& sd = a.s[sl];
    int64_t alo = 0, ahi = nactive;
    while (alo < ahi) {   // position in active[] of the first row >= B
        int64_t mid = (alo + ahi) >> 1;
        if (active[mid] < B) alo = mid + 1; else ahi = mid;
    }
    int64_t lo = 0, hi = sd.crows;
    while (lo < hi) {   // first compact row at or after it (compact rows are named by their position in active[])
        int64_t mid = (lo + hi) >> 1;
        if (sd.rowid[mid] < alo) lo = mid + 1; else hi = mid;
    }
    // path position of a row's first item = edges before it + row ends before it
    out[t] = lo < sd.crows ? ((int64_t) sd.rb[lo] + lo) / items : sd.nblk;
}

__global__ void pr_mark_active_kernel(const uint64_t* __restrict__ keys, int64_t n, int64_t row_lo, uint8_t* __restrict__ is_active) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < n; i += stride)
        if (i == 0 || (keys[i - 1] >> 32) != (keys[i] >> 32)) is_active[(int64_t) (keys[i] >> 32) - row_lo] = 1;
}

// sums `take` entries of each of `groups` groups laid out `stride` apart (fixed order: deterministic)
__global__ void pr_diff_reduce_kernel(const double* __restrict__ part, int64_t groups, int64_t stride, int64_t take, double* __restrict__ out) {
    __shared__ double s[1024 / 64];
    double t = 0.0;
    for (int64_t i = threadIdx.x; i < groups * take; i += blockDim.x) t += part[(i / take) * stride + i % take];
    t = wave_sum(t);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0;
        for (int w = 0; w < (int) (blockDim.x >> 6); w++) r += s[w];
        *out = r;
    }
}

// the same for two arrays one after the other (fixed order)
__global__ void pr_diff_reduce2_kernel(const double* __restrict__ a, int64_t na, const double* __restrict__ b, int64_t nb, double* __restrict__ out) {
    __shared__ double s[1024 / 64];
    double t = 0.0;
    for (int64_t i = threadIdx.x; i < na + nb; i += blockDim.x) t += i < na ? a[i] : b[i - na];
    t = wave_sum(t);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0;
        for (int w = 0; w < (int) (blockDim.x >> 6); w++) r += s[w];
        *out = r;
    }
}

template <typename S>
__global__ void pr_reset_kernel(int64_t rows, double N, const int32_t* __restrict__ outdeg,
                                S* __restrict__ rk, S* __restrict__ contrib_owned) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < rows; i += stride) {
        int32_t od = outdeg[i];
        S r0 = od < 0 ? (S) 0 : (S) (1 / N);      // G.pg_rank = 1 / N   (od < 0: padding row)
        rk[i] = r0;
        contrib_owned[i] = od > 0 ? (S) ((double) r0 / (double) od) : (S) 0;
    }
}

template <typename S>
__global__ void pr_unpermute_kernel(int64_t rows, const int32_t* __restrict__ inv, const S* __restrict__ rk,
                                    S* __restrict__ out_orig) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < rows; i += stride)
        if (inv[i] >= 0) out_orig[inv[i]] = rk[i];
}

// ------------------------------------------------------------------ plan
static int pr_build_chunks(gmx_pr* p, int C);
// every in-edge of the owned rows is binned and every row is reached by the bins' fused finish
static inline bool pr_fused(const gmx_pr* p) { return p->cold && p->Eh == 0 && pr_cold_covers_all_rows(p->cold); }

extern "C" int gmx_pr_create(gmx_graph_t* g, int elem_bytes, int rank, int nranks, uint32_t options, gmx_pr_t** out) {
    GMX_REQUIRE(out, "out is NULL");
    *out = nullptr;
    GMX_REQUIRE(g, "graph is NULL");
    GMX_REQUIRE(g->has_reverse, "pagerank needs the reverse CSR (graph built with GMX_GRAPH_NO_REVERSE)");
    GMX_REQUIRE(elem_bytes == 4 || elem_bytes == 8, "elem_bytes must be 4 or 8");
    GMX_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank %d / nranks %d", rank, nranks);
    gmx_pr* p = new gmx_pr();
    p->g = g;
    p->elem = elem_bytes;
    p->rank = rank;
    p->nranks = nranks;
    p->options = options;
    p->V = g->V;
    const bool relabel = (options & GMX_PR_RELABEL) != 0;
    if ((options & GMX_PR_SLICED) && relabel) {
        // measured (profiles/): one slice per XCD L2 from 2^24 vertices on (RMAT-26: 4.80 vs 5.08 ms), below that
        // two XCDs share a slice (fewer (slice,row) pairs; RMAT-22: 0.265 vs 0.280 ms)
        p->ns = g->V >= (1LL << 24) ? 8 : 4;
        const char* ev = getenv("GMX_PR_SLICES");
        if (ev && atoi(ev) >= 1 && atoi(ev) <= PR_MAX_SLICES) p->ns = atoi(ev);
    }
    p->slice = (g->V + nranks - 1) / nranks;
    if (p->slice < 1) p->slice = 1;
    if (p->ns > 0) {   // whole runs for every slice
        const int64_t unit = (int64_t) p->ns << PR_RUN_SHIFT;
        p->slice = (p->slice + unit - 1) / unit * unit;
    }
    p->Vpad = p->slice * nranks;
    if (p->Vpad >= (1LL << 31)) {
        gmx_set_error("padded vertex count %lld exceeds int32", (long long) p->Vpad);
        delete p;
        return GMX_ERR_ARG;
    }
    p->row_lo = (int64_t) rank * p->slice;
    if (relabel) p->rows_real = g->V > rank ? (g->V - rank + nranks - 1) / nranks : 0;
    else {
        int64_t hi = (int64_t) (rank + 1) * p->slice;
        if (hi > g->V) hi = g->V;
        p->rows_real = hi > p->row_lo ? hi - p->row_lo : 0;
    }
    p->rows = p->ns > 0 ? p->slice : p->rows_real;
    p->exchange_count = p->slice;
    p->hot = (options & GMX_PR_HOT_LDS) != 0 && relabel && (nranks == 1 || p->ns > 0);
    p->items = PRW_ITEMS;

    const int64_t V = g->V, E = g->E;
    hipStream_t s = 0;
    int st = GMX_OK;
    gmx_tick tick("pr plan");
    gmx_ws_scope ws;   // the temporaries of the build are workspace memory (wbuf)
    do {
        if ((st = p->inv.alloc((size_t) p->rows)) || (st = p->outdeg.alloc((size_t) p->rows))) break;
        if (p->rows && (hipMemset(p->inv.p, 0xff, sizeof(int32_t) * (size_t) p->rows) != hipSuccess ||
                        hipMemset(p->outdeg.p, 0xff, sizeof(int32_t) * (size_t) p->rows) != hipSuccess)) { gmx_set_error("pr plan: memset failed"); st = GMX_ERR_HIP; break; }
        if (!relabel && nranks == 1) {
            // identity numbering, whole graph: use the graph's reverse CSR as is
            p->rb = g->r_begin.p;
            p->ridx = g->r_node_idx.p;
            p->El = E;
            wbuf<int32_t> perm;
            if ((st = perm.alloc((size_t) V))) break;
            hipLaunchKernelGGL(pr_perm_kernel, dim3(grid_for(V)), dim3(256), 0, s, (const int32_t*) nullptr, V, p->slice, nranks, 0, perm.p);
            hipLaunchKernelGGL(pr_owned_kernel, dim3(grid_for(V)), dim3(256), 0, s, perm.p, g->begin.p, V, p->row_lo, p->rows, p->inv.p, p->outdeg.p);
            if (hipStreamSynchronize(s) != hipSuccess) { gmx_set_error("pr plan (identity) failed"); st = GMX_ERR_HIP; break; }
        } else {
            wbuf<int32_t> perm;
            if ((st = perm.alloc((size_t) V))) break;
            if (relabel) {
                wbuf<uint32_t> key, key2;
                wbuf<int32_t> id, order;
                if ((st = key.alloc((size_t) V)) || (st = key2.alloc((size_t) V)) || (st = id.alloc((size_t) V)) ||
                    (st = order.alloc((size_t) V))) break;
                tick.mark("allocations");
                hipLaunchKernelGGL(pr_degkey_kernel, dim3(grid_for(V)), dim3(256), 0, s, g->begin.p, V, key.p, id.p);
                size_t tb = 0;
                hipError_t he = rocprim::radix_sort_pairs(nullptr, tb, key.p, key2.p, id.p, order.p, (size_t) V, 0u, 32u, s);
                wbuf<char> tmp;
                if (he == hipSuccess && (st = tmp.alloc(tb))) break;
                if (he == hipSuccess) he = rocprim::radix_sort_pairs((void*) tmp.p, tb, key.p, key2.p, id.p, order.p, (size_t) V, 0u, 32u, s);
                if (he == hipSuccess) {
                    hipLaunchKernelGGL(pr_perm_kernel, dim3(grid_for(V)), dim3(256), 0, s, (const int32_t*) order.p, V, p->slice, nranks, p->ns, perm.p);
                    he = hipStreamSynchronize(s);
                }
                if (he != hipSuccess) { gmx_set_error("pr plan: degree sort failed: %s", hipGetErrorString(he)); st = GMX_ERR_HIP; break; }
                // vertices with out-degree 0 are never gathered; in the degree order they form the tail of every
                // rank's range, so only a prefix of each range has to travel between ranks
                wbuf<int64_t> nzd;
                if ((st = nzd.alloc(2))) break;
                hipLaunchKernelGGL(pr_key32_bound_kernel, dim3(1), dim3(64), 0, s, (const uint32_t*) key2.p, V, 0x7fffffffu, nzd.p);
                int64_t nz = V;
                if (hipMemcpy(&nz, nzd.p, sizeof(int64_t), hipMemcpyDeviceToHost) != hipSuccess) { gmx_set_error("pr plan: copy failed"); st = GMX_ERR_HIP; break; }
                int64_t need = (nz + nranks - 1) / nranks;
                const int64_t unit = p->ns > 0 ? ((int64_t) p->ns << PR_RUN_SHIFT) : 1;
                need = (need + unit - 1) / unit * unit;
                p->exchange_count = need < p->slice ? need : p->slice;
            } else {
                hipLaunchKernelGGL(pr_perm_kernel, dim3(grid_for(V)), dim3(256), 0, s, (const int32_t*) nullptr, V, p->slice, nranks, 0, perm.p);
            }
            hipLaunchKernelGGL(pr_owned_kernel, dim3(grid_for(V)), dim3(256), 0, s, perm.p, g->begin.p, V, p->row_lo, p->rows, p->inv.p, p->outdeg.p);
            tick.mark("degree order");
            // keys (perm[dst] << 32 | perm[src]) from the reverse CSR, sorted; then cut the owned rows
            wbuf<uint64_t> keys, alt;
            if ((st = keys.alloc((size_t) E)) || (st = alt.alloc((size_t) E))) break;
            if ((st = gmx_keys_from_csr(g->r_begin.p, g->r_node_idx.p, V, E, false, perm.p, keys.p, s))) break;
            const uint64_t* sorted = keys.p;
            tick.mark("keys");
            // Every in-edge binned (the default from 2^20 vertices): the binned plan orders its edges itself and asks for
            // them in ANY order, so the (row, source) sort of all E keys -- the most expensive step of a plan build -- is
            // left out: with one rank the keys are used as they come, with several the owned rows are selected (stable).
            int64_t cold_T_req = 0;
            {
                const char* ev = getenv("GMX_PR_COLD");
                if (ev && *ev) cold_T_req = atoll(ev);
                if (p->ns > 0) cold_T_req = cold_T_req / ((int64_t) p->ns << PR_RUN_SHIFT) * ((int64_t) p->ns << PR_RUN_SHIFT);
            }
            const bool unsorted_ok = p->ns > 0 && (options & GMX_PR_COLD_PB) && cold_T_req == 0 && ((int64_t) p->ns << PR_RUN_SHIFT) <= p->slice;
            int64_t hb[2] = {0, 0};
            if (unsorted_ok) {
                if (nranks == 1) hb[1] = E;
                else {
                    wbuf<int64_t> nsel;
                    if ((st = nsel.alloc(1))) break;
                    pr_row_in_range pred{(uint64_t) p->row_lo << 32, (uint64_t) (p->row_lo + p->rows) << 32};
                    size_t tb = 0;
                    hipError_t he = rocprim::select(nullptr, tb, (const uint64_t*) keys.p, alt.p, nsel.p, (size_t) E, pred, s);
                    wbuf<char> tmp;
                    if (he == hipSuccess && (st = tmp.alloc(tb))) break;
                    if (he == hipSuccess) he = rocprim::select((void*) tmp.p, tb, (const uint64_t*) keys.p, alt.p, nsel.p, (size_t) E, pred, s);
                    if (he == hipSuccess) he = hipMemcpy(&hb[1], nsel.p, sizeof(int64_t), hipMemcpyDeviceToHost);
                    if (he != hipSuccess) { gmx_set_error("pr plan: owned-row selection failed: %s", hipGetErrorString(he)); st = GMX_ERR_HIP; break; }
                    sorted = alt.p;
                }
            }
            if (!unsorted_ok && E > 1) {
                rocprim::double_buffer<uint64_t> db(keys.p, alt.p);
                size_t tb = 0;
                unsigned end_bit = 32 + (unsigned) gmx_bits_for(p->Vpad);
                hipError_t he = rocprim::radix_sort_keys(nullptr, tb, db, (size_t) E, 0u, end_bit, s);
                wbuf<char> tmp;
                if (he == hipSuccess && (st = tmp.alloc(tb))) break;
                if (he == hipSuccess) he = rocprim::radix_sort_keys((void*) tmp.p, tb, db, (size_t) E, 0u, end_bit, s);
                if (he == hipSuccess) he = hipStreamSynchronize(s);
                if (he != hipSuccess) { gmx_set_error("pr plan: key sort failed: %s", hipGetErrorString(he)); st = GMX_ERR_HIP; break; }
                sorted = db.current();
            }
            if (!unsorted_ok) {
                wbuf<int64_t> bounds;
                if ((st = bounds.alloc(2))) break;
                hipLaunchKernelGGL(pr_key_bound_kernel, dim3(1), dim3(64), 0, s, sorted, E,
                                   (uint64_t) p->row_lo << 32, (uint64_t) (p->row_lo + p->rows) << 32, bounds.p);
                if (hipMemcpy(hb, bounds.p, sizeof(hb), hipMemcpyDeviceToHost) != hipSuccess) { gmx_set_error("pr plan: bounds copy failed"); st = GMX_ERR_HIP; break; }
            }
            p->El = hb[1] - hb[0];
            p->Eh = p->El;
            if (p->ns > 0) {
                const int ns = p->ns;
                const int64_t rows = p->rows;
                // ---- cold sources: ids >= cold_T inside their rank range leave the pull sweep ----
                const uint64_t* own = sorted + hb[0];   // all in-edges of the owned rows, (row, source) ascending
                const uint64_t* hk = own;               // the ones the pull sweep keeps
                const uint64_t* cold_keys = nullptr;
                int64_t Ec = 0;
                p->cold_T = -1;
                if (options & GMX_PR_COLD_PB) {
                    // default 0: every edge is binned.  Measured on RMAT-26 fp32 (profiles/round2_*): 2.25 ms per
                    // iteration, against 3.57 with the 180 K hottest sources (the ones the LDS tiles of the pull sweep
                    // hold) left to the pull sweep, 4.08 with the first 1 Mi (what the L2s hold), 4.76 without bins
                    const int64_t unit = (int64_t) ns << PR_RUN_SHIFT;
                    int64_t T = 0;
                    const char* ev = getenv("GMX_PR_COLD");
                    if (ev && *ev) T = atoll(ev);
                    T = T / unit * unit;
                    // worth it only with at least a few LDS tiles of cold sources
                    const int64_t margin = unit;
                    if (T >= 0 && T + margin <= p->slice && p->El > 0) p->cold_T = T;
                }
                if (p->cold_T >= 0 && unsorted_ok) {   // nothing stays with the pull sweep
                    p->Eh = 0;
                    cold_keys = own;
                    Ec = p->El;
                } else if (p->cold_T >= 0) {
                    uint64_t* other = (sorted == keys.p) ? alt.p : keys.p;
                    wbuf<int64_t> nsel;
                    if ((st = nsel.alloc(1))) break;
                    pr_is_hot_key pred{(uint32_t) p->slice, (uint32_t) p->cold_T};
                    size_t tb = 0;
                    hipError_t he = rocprim::partition(nullptr, tb, own, other, nsel.p, (size_t) p->El, pred, s);
                    wbuf<char> tmp;
                    if (he == hipSuccess && (st = tmp.alloc(tb))) break;
                    if (he == hipSuccess) he = rocprim::partition((void*) tmp.p, tb, own, other, nsel.p, (size_t) p->El, pred, s);
                    int64_t nh = 0;
                    if (he == hipSuccess) he = hipMemcpy(&nh, nsel.p, sizeof(int64_t), hipMemcpyDeviceToHost);
                    if (he != hipSuccess) { gmx_set_error("pr plan: hot/cold partition failed: %s", hipGetErrorString(he)); st = GMX_ERR_HIP; break; }
                    hk = other;                 // selected keys keep their order
                    p->Eh = nh;
                    cold_keys = other + nh;     // rejected keys (reversed): any order will do
                    Ec = p->El - nh;
                }
                // group the kept keys by source slice (stable 1-pass radix sort on the slice number)
                const int64_t El = p->Eh;
                wbuf<uint8_t> sk, sk2;
                wbuf<uint64_t> vals;
                wbuf<int64_t> off;
                if ((st = sk.alloc((size_t) El)) || (st = sk2.alloc((size_t) El)) || (st = vals.alloc((size_t) El)) ||
                    (st = off.alloc(PR_MAX_SLICES + 1))) break;
                hipLaunchKernelGGL(pr_slice_key_kernel, dim3(grid_for(El)), dim3(256), 0, s, hk, El, ns, sk.p);
                hipError_t he = hipSuccess;
                if (El > 0) {
                    size_t tb = 0;
                    he = rocprim::radix_sort_pairs(nullptr, tb, sk.p, sk2.p, hk, vals.p, (size_t) El, 0u, 3u, s);
                    wbuf<char> tmp;
                    if (he == hipSuccess && (st = tmp.alloc(tb))) break;
                    if (he == hipSuccess) he = rocprim::radix_sort_pairs((void*) tmp.p, tb, sk.p, sk2.p, hk, vals.p, (size_t) El, 0u, 3u, s);
                    if (he == hipSuccess) he = hipStreamSynchronize(s);
                }
                if (he != hipSuccess) { gmx_set_error("pr plan: slice sort failed: %s", hipGetErrorString(he)); st = GMX_ERR_HIP; break; }
                hipLaunchKernelGGL(pr_slice_offsets_kernel, dim3(1), dim3(64), 0, s, (const uint8_t*) sk2.p, El, ns, off.p);
                int64_t hoff[PR_MAX_SLICES + 1];
                if (hipMemcpy(hoff, off.p, sizeof(int64_t) * (ns + 1), hipMemcpyDeviceToHost) != hipSuccess) { gmx_set_error("pr plan: slice offsets copy failed"); st = GMX_ERR_HIP; break; }
                // rows with in-edges (the only ones the combine pass has to visit)
                {
                    if ((st = p->sl_is_active.alloc((size_t) (rows ? rows : 1))) || (st = p->sl_active.alloc((size_t) (rows ? rows : 1)))) break;
                    if (hipMemsetAsync(p->sl_is_active.p, 0, (size_t) (rows ? rows : 1), s) != hipSuccess) { gmx_set_error("pr plan: memset failed"); st = GMX_ERR_HIP; break; }
                    if (p->El > 0) hipLaunchKernelGGL(pr_mark_active_kernel, dim3(grid_for(p->El)), dim3(256), 0, s, own, p->El, p->row_lo, p->sl_is_active.p);
                    wbuf<int64_t> cnt;
                    if ((st = cnt.alloc(1))) break;
                    size_t tb = 0;
                    rocprim::counting_iterator<int32_t> ids(0);
                    he = rocprim::select(nullptr, tb, ids, (const uint8_t*) p->sl_is_active.p, p->sl_active.p, cnt.p, (size_t) rows, s);
                    wbuf<char> tmp3;
                    if (he == hipSuccess && (st = tmp3.alloc(tb))) break;
                    if (he == hipSuccess) he = rocprim::select((void*) tmp3.p, tb, ids, (const uint8_t*) p->sl_is_active.p, p->sl_active.p, cnt.p, (size_t) rows, s);
                    if (he == hipSuccess) he = hipMemcpy(&p->sl_nactive, cnt.p, sizeof(int64_t), hipMemcpyDeviceToHost);
                    if (he != hipSuccess) { gmx_set_error("pr plan: active-row list failed: %s", hipGetErrorString(he)); st = GMX_ERR_HIP; break; }
                }
                // compact rows: one (slice,row) pair per row that has at least one edge in the slice
                wbuf<int32_t> flag, pos;
                if ((st = flag.alloc((size_t) El + 1)) || (st = pos.alloc((size_t) El + 1)) || (st = p->sl_ridx.alloc((size_t) El))) break;
                hipLaunchKernelGGL(pr_slice_flag_kernel, dim3(grid_for(El)), dim3(256), 0, s, (const uint64_t*) vals.p,
                                   (const uint8_t*) sk2.p, El, flag.p, p->sl_ridx.p);
                int64_t npairs = 0;
                int64_t pair_off[PR_MAX_SLICES + 1];
                if (El > 0) {
                    size_t tb = 0;
                    he = rocprim::exclusive_scan(nullptr, tb, flag.p, pos.p, 0, (size_t) El, rocprim::plus<int32_t>(), s);
                    wbuf<char> tmp2;
                    if (he == hipSuccess && (st = tmp2.alloc(tb))) break;
                    if (he == hipSuccess) he = rocprim::exclusive_scan((void*) tmp2.p, tb, flag.p, pos.p, 0, (size_t) El, rocprim::plus<int32_t>(), s);
                    if (he == hipSuccess) he = hipStreamSynchronize(s);
                    if (he != hipSuccess) { gmx_set_error("pr plan: pair scan failed: %s", hipGetErrorString(he)); st = GMX_ERR_HIP; break; }
                    int32_t lastp = 0, lastf = 0;
                    if (hipMemcpy(&lastp, pos.p + (El - 1), 4, hipMemcpyDeviceToHost) != hipSuccess ||
                        hipMemcpy(&lastf, flag.p + (El - 1), 4, hipMemcpyDeviceToHost) != hipSuccess) { gmx_set_error("pr plan: copy failed"); st = GMX_ERR_HIP; break; }
                    npairs = (int64_t) lastp + lastf;
                }
                for (int q = 0; q <= ns; q++) {   // pairs before the first position of slice q
                    pair_off[q] = npairs;
                    if (hoff[q] < El) {
                        int32_t v = 0;
                        if (hipMemcpy(&v, pos.p + hoff[q], 4, hipMemcpyDeviceToHost) != hipSuccess) { gmx_set_error("pr plan: copy failed"); st = GMX_ERR_HIP; break; }
                        pair_off[q] = v;
                    }
                }
                if (st) break;
                if ((st = p->sl_rb.alloc((size_t) npairs + ns + 1)) || (st = p->sl_rowid.alloc((size_t) (npairs ? npairs : 1)))) break;
                hipLaunchKernelGGL(pr_slice_pairs_kernel, dim3(grid_for(El)), dim3(256), 0, s, (const uint64_t*) vals.p,
                                   (const uint8_t*) sk2.p, (const int32_t*) flag.p, (const int32_t*) pos.p, (const int64_t*) off.p,
                                   El, p->row_lo, p->sl_rowid.p, p->sl_rb.p);
                wbuf<int32_t> index_of_row;
                {   // name the compact rows by their position in active[] and make the dense per-active-row copies
                    const size_t na = (size_t) (p->sl_nactive ? p->sl_nactive : 1);
                    if ((st = index_of_row.alloc((size_t) (rows ? rows : 1))) || (st = p->sl_outdeg_c.alloc(na)) ||
                        (st = p->sl_rk_c.alloc(na * elem_bytes))) break;
                    if (p->sl_nactive > 0) {
                        hipLaunchKernelGGL(pr_active_index_kernel, dim3(grid_for(p->sl_nactive)), dim3(256), 0, s,
                                           (const int32_t*) p->sl_active.p, p->sl_nactive, index_of_row.p);
                        if (npairs > 0)
                            hipLaunchKernelGGL(pr_remap_kernel, dim3(grid_for(npairs)), dim3(256), 0, s, p->sl_rowid.p, npairs, (const int32_t*) index_of_row.p);
                        hipLaunchKernelGGL(pr_gather_i32_kernel, dim3(grid_for(p->sl_nactive)), dim3(256), 0, s,
                                           (const int32_t*) p->sl_active.p, p->sl_nactive, (const int32_t*) p->outdeg.p, p->sl_outdeg_c.p);
                    }
                    if (hipStreamSynchronize(s) != hipSuccess) { gmx_set_error("pr plan: active-row renaming failed"); st = GMX_ERR_HIP; break; }
                }
                if (nranks > 1 && relabel && nranks <= 16 && !getenv("GMX_PR_NO_PACKED")) {
                    // ---- packed exchange lists (see the struct) ----
                    wbuf<unsigned int> rmask;
                    wbuf<uint8_t> mark;
                    wbuf<int64_t> nsel;
                    wbuf<char> tmp;
                    if ((st = rmask.alloc((size_t) rows)) || (st = mark.alloc((size_t) p->Vpad)) || (st = nsel.alloc(1))) break;
                    hipError_t he = hipMemsetAsync(rmask.p, 0, sizeof(unsigned int) * (size_t) rows, s);
                    if (he == hipSuccess) he = hipMemsetAsync(mark.p, 0, (size_t) p->Vpad, s);
                    if (he != hipSuccess) { gmx_set_error("pr plan: memset failed"); st = GMX_ERR_HIP; break; }
                    hipLaunchKernelGGL(pr_reader_mask_kernel, dim3(grid_for((E + 15) / 16)), dim3(256), 0, s, g->begin.p, g->node_idx.p, V, E,
                                       (const int32_t*) perm.p, p->slice, p->row_lo, rows, rmask.p);
                    if (p->El > 0) hipLaunchKernelGGL(pr_mark_sources_kernel, dim3(grid_for(p->El)), dim3(256), 0, s, own, p->El, mark.p);
                    wbuf<int32_t> sl_tmp, rl_tmp;
                    if ((st = sl_tmp.alloc((size_t) p->exchange_count * nranks + 1)) || (st = rl_tmp.alloc((size_t) p->exchange_count * nranks + 1))) break;
                    p->send_cnt.assign((size_t) nranks, 0); p->send_off.assign((size_t) nranks + 1, 0);
                    p->recv_cnt.assign((size_t) nranks, 0); p->recv_off.assign((size_t) nranks + 1, 0);
                    rocprim::counting_iterator<int32_t> ids(0);
                    bool ok = true;
                    for (int q = 0; q < nranks && ok; q++) {
                        for (int dir = 0; dir < 2 && ok; dir++) {   // 0: what rank q reads of mine, 1: what I read of rank q's
                            int64_t n = 0;
                            if (q != rank) {
                                int32_t* out = (dir ? rl_tmp.p + p->recv_off[(size_t) q] : sl_tmp.p + p->send_off[(size_t) q]);
                                size_t tb = 0;
                                if (dir == 0) {
                                    pr_mask_has_bit pred{rmask.p, 1u << q};
                                    he = rocprim::select(nullptr, tb, ids, out, nsel.p, (size_t) p->exchange_count, pred, s);
                                    if (he == hipSuccess && tmp.n < tb) { tmp.release(); if (tmp.alloc(tb)) he = hipErrorOutOfMemory; }
                                    if (he == hipSuccess) he = rocprim::select((void*) tmp.p, tb, ids, out, nsel.p, (size_t) p->exchange_count, pred, s);
                                } else {
                                    pr_is_marked pred{mark.p + (size_t) q * (size_t) p->slice};
                                    he = rocprim::select(nullptr, tb, ids, out, nsel.p, (size_t) p->exchange_count, pred, s);
                                    if (he == hipSuccess && tmp.n < tb) { tmp.release(); if (tmp.alloc(tb)) he = hipErrorOutOfMemory; }
                                    if (he == hipSuccess) he = rocprim::select((void*) tmp.p, tb, ids, out, nsel.p, (size_t) p->exchange_count, pred, s);
                                }
                                if (he == hipSuccess) he = hipMemcpy(&n, nsel.p, sizeof(int64_t), hipMemcpyDeviceToHost);
                                if (he != hipSuccess) ok = false;
                            }
                            if (dir) { p->recv_cnt[(size_t) q] = n; p->recv_off[(size_t) q + 1] = p->recv_off[(size_t) q] + n; }
                            else { p->send_cnt[(size_t) q] = n; p->send_off[(size_t) q + 1] = p->send_off[(size_t) q] + n; }
                        }
                    }
                    if (!ok) { gmx_set_error("pr plan: exchange lists failed: %s", hipGetErrorString(he)); st = GMX_ERR_HIP; break; }
                    const size_t ns_ = (size_t) p->send_off[(size_t) nranks], nr_ = (size_t) p->recv_off[(size_t) nranks];
                    if ((st = p->slist.alloc(ns_ ? ns_ : 1)) || (st = p->rlist.alloc(nr_ ? nr_ : 1)) || (st = p->sbuf.alloc((ns_ ? ns_ : 1) * elem_bytes)) ||
                        (st = p->rbuf[0].alloc((nr_ ? nr_ : 1) * elem_bytes)) || (st = p->rbuf[1].alloc((nr_ ? nr_ : 1) * elem_bytes))) break;
                    if ((ns_ && hipMemcpyAsync(p->slist.p, sl_tmp.p, sizeof(int32_t) * ns_, hipMemcpyDeviceToDevice, s) != hipSuccess) ||
                        (nr_ && hipMemcpyAsync(p->rlist.p, rl_tmp.p, sizeof(int32_t) * nr_, hipMemcpyDeviceToDevice, s) != hipSuccess) ||
                        hipStreamSynchronize(s) != hipSuccess) { gmx_set_error("pr plan: exchange list copy failed"); st = GMX_ERR_HIP; break; }
                    p->packed = true;
                }
                tick.mark("owned rows, active rows");
                if (p->cold_T >= 0) {   // tile- and bin-major streams of the binned edges
                    wbuf<int32_t> deg_by_id;
                    if ((st = deg_by_id.alloc((size_t) p->Vpad))) break;
                    if (hipMemsetAsync(deg_by_id.p, 0xff, sizeof(int32_t) * (size_t) p->Vpad, s) != hipSuccess) { gmx_set_error("pr plan: memset failed"); st = GMX_ERR_HIP; break; }
                    hipLaunchKernelGGL(pr_deg_by_id_kernel, dim3(grid_for(V)), dim3(256), 0, s, (const int32_t*) perm.p, g->begin.p, V, deg_by_id.p);
                    pr_cold_params cp{elem_bytes, nranks, p->slice, p->cold_T, p->row_lo, p->sl_nactive, index_of_row.p, deg_by_id.p};
                    if ((st = pr_cold_create(cold_keys, Ec, cp, s, &p->cold))) break;
                }
                int64_t nblk_s[PR_MAX_SLICES], blk_off[PR_MAX_SLICES + 1];
                blk_off[0] = 0;
                for (int q = 0; q < ns; q++) {
                    const int64_t crows = pair_off[q + 1] - pair_off[q], es = hoff[q + 1] - hoff[q];
                    const int32_t term = (int32_t) es;   // terminating rb entry of the slice
                    if (hipMemcpy(p->sl_rb.p + pair_off[q + 1] + q, &term, 4, hipMemcpyHostToDevice) != hipSuccess) { gmx_set_error("pr plan: copy failed"); st = GMX_ERR_HIP; break; }
                    int64_t tot = crows + es;
                    nblk_s[q] = (tot + p->items - 1) / p->items;
                    blk_off[q + 1] = blk_off[q] + nblk_s[q] + 1;
                }
                if (st) break;
                p->sl_nblk_total = blk_off[ns];
                size_t npart = (size_t) (p->sl_nblk_total ? p->sl_nblk_total : 1);
                const size_t nact_alloc = (size_t) (p->sl_nactive ? p->sl_nactive : 1);
                const size_t partial_bytes = (size_t) ns * nact_alloc * elem_bytes;
                if ((st = p->sl_blk.alloc(npart)) || (st = p->sl_part_first.alloc(npart)) || (st = p->sl_part_last.alloc(npart)) ||
                    (st = p->sl_partial.alloc(partial_bytes)) || (st = p->sl_queue.alloc(PR_MAX_SLICES * PR_QUEUE_STRIDE))) break;
                if (hipMemset(p->sl_partial.p, 0, partial_bytes) != hipSuccess ||
                    hipMemset(p->sl_queue.p, 0, sizeof(unsigned int) * PR_MAX_SLICES * PR_QUEUE_STRIDE) != hipSuccess) { gmx_set_error("pr plan: memset failed"); st = GMX_ERR_HIP; break; }
                memset(&p->sl, 0, sizeof(p->sl));
                p->sl.ns = ns;
                p->sl.queue = p->sl_queue.p;
                for (int q = 0; q < ns; q++) {
                    pr_slice_desc& sd = p->sl.s[q];
                    sd.blk = p->sl_blk.p + blk_off[q];
                    sd.nblk = nblk_s[q];
                    sd.rb = p->sl_rb.p + pair_off[q] + q;
                    sd.ridx = p->sl_ridx.p + hoff[q];
                    sd.part_first = p->sl_part_first.p + blk_off[q];
                    sd.part_last = p->sl_part_last.p + blk_off[q];
                    sd.partial = p->sl_partial.p + (size_t) q * nact_alloc * elem_bytes;
                    sd.rowid = p->sl_rowid.p + pair_off[q];
                    sd.crows = pair_off[q + 1] - pair_off[q];
                    hipLaunchKernelGGL(pr_blocks_kernel, dim3(grid_for(nblk_s[q] + 1, 256, 1 << 30)), dim3(256), 0, s,
                                       sd.rb, sd.crows, hoff[q + 1] - hoff[q], p->items, nblk_s[q], p->sl_blk.p + blk_off[q]);
                }
                if (hipStreamSynchronize(s) != hipSuccess) { gmx_set_error("pr plan: slice csr failed"); st = GMX_ERR_HIP; break; }
            } else {
                if ((st = p->rb_own.alloc((size_t) p->rows + 1)) || (st = p->ridx_own.alloc((size_t) p->El))) break;
                hipLaunchKernelGGL(pr_local_csr_kernel, dim3(grid_for(p->El > p->rows ? p->El : p->rows + 1)), dim3(256), 0, s,
                                   sorted, E, p->row_lo, p->rows, hb[0], p->El, p->rb_own.p, p->ridx_own.p);
                if (hipStreamSynchronize(s) != hipSuccess) { gmx_set_error("pr plan: local csr failed"); st = GMX_ERR_HIP; break; }
                p->rb = p->rb_own.p;
                p->ridx = p->ridx_own.p;
            }
        }
        // merge-path blocks
        int64_t total = p->rows + p->El;
        p->nblk = p->ns > 0 ? 2 * PR_MAX_CHUNKS * PR_COMBINE_GRID : (total + p->items - 1) / p->items;   // sliced: diff partials of the combine (+ first-sweep) grids of every chunk
        if ((st = p->blk.alloc((size_t) p->nblk + 1))) break;
        if (p->ns == 0)
            hipLaunchKernelGGL(pr_blocks_kernel, dim3(grid_for(p->nblk + 1, 256, 1 << 30)), dim3(256), 0, s,
                               p->rb, p->rows, p->El, p->items, p->nblk, p->blk.p);
        size_t nb = (size_t) (p->nblk ? p->nblk : 1);
        if ((st = p->part_first.alloc(nb)) || (st = p->part_last.alloc(nb)) || (st = p->diff_part.alloc(2 * nb + 4096)) ||
            (st = p->diff.alloc(1)) || (st = p->rk.alloc((size_t) (p->rows ? p->rows : 1) * elem_bytes)) ||
            (st = p->contrib[0].alloc((size_t) p->Vpad * elem_bytes)) || (st = p->contrib[1].alloc((size_t) p->Vpad * elem_bytes))) break;
        if (hipMemset(p->contrib[0].p, 0, (size_t) p->Vpad * elem_bytes) != hipSuccess ||
            hipMemset(p->contrib[1].p, 0, (size_t) p->Vpad * elem_bytes) != hipSuccess ||
            hipMemset(p->diff_part.p, 0, (2 * nb + 4096) * sizeof(double)) != hipSuccess ||
            hipMemset(p->diff.p, 0, sizeof(double)) != hipSuccess ||
            hipDeviceSynchronize() != hipSuccess) { gmx_set_error("pr plan: memset failed"); st = GMX_ERR_HIP; break; }
        hipDeviceProp_t prop;
        int dev = 0;
        (void) hipGetDevice(&dev);
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { gmx_set_error("hipGetDeviceProperties failed"); st = GMX_ERR_HIP; break; }
        p->persistent_grid = prop.multiProcessorCount;
        if (p->ns > 0 && (st = pr_build_chunks(p, 1))) break;
        tick.mark("binned plan + rest");
    } while (0);
    if (st != GMX_OK) { delete p; return st; }
    *out = p;
    return GMX_OK;
}

extern "C" int gmx_pr_free(gmx_pr_t* p) {
    delete p;
    return GMX_OK;
}

extern "C" int gmx_pr_reset(gmx_pr_t* p, double d) {
    GMX_REQUIRE(p, "pr is NULL");
    if (p->cold) {
        // the binned sweep adds in 2^-62 fixed point: contributions must stay in [0, 1], which the iteration keeps for
        // 0 < d <= 1 (ranks >= 0, their sum <= 1) and nothing keeps outside (the reference accepts any d > 0,
        // pagerank_main.cc:59-62: the whole-kernel entries run those on the pull sweep, which sums in floating point)
        GMX_REQUIRE(d > 0.0 && d <= 1.0, "damping factor %g outside (0, 1]: not representable by the binned sweep (plan without GMX_PR_COLD_PB)", d);
        if (!pr_cold_limb_guard(p->cold, d, (double) p->V)) {
            gmx_set_error("fp32 binned sweep: one 2^-62 limb cannot guarantee 1e-6 for d = %g on this graph (use the fp64 plan)", d);
            return GMX_ERR_STATE;
        }
    }
    p->d = d;
    p->cnt = 0;
    p->cur = 0;
    p->gather_mask = 0;
    if (p->rows > 0) {
        if (p->elem == 4)
            hipLaunchKernelGGL(pr_reset_kernel<float>, dim3(grid_for(p->rows)), dim3(256), 0, 0, p->rows, (double) p->V,
                               p->outdeg.p, (float*) p->rk.p, (float*) p->contrib[0].p + p->row_lo);
        else
            hipLaunchKernelGGL(pr_reset_kernel<double>, dim3(grid_for(p->rows)), dim3(256), 0, 0, p->rows, (double) p->V,
                               p->outdeg.p, (double*) p->rk.p, (double*) p->contrib[0].p + p->row_lo);
        if (p->ns > 0 && p->sl_nactive > 0) {
            if (p->elem == 4)
                hipLaunchKernelGGL(pr_rk_dense_kernel<float>, dim3(grid_for(p->sl_nactive)), dim3(256), 0, 0, (const int32_t*) p->sl_active.p,
                                   p->sl_nactive, (float*) p->rk.p, (float*) p->sl_rk_c.p, 0);
            else
                hipLaunchKernelGGL(pr_rk_dense_kernel<double>, dim3(grid_for(p->sl_nactive)), dim3(256), 0, 0, (const int32_t*) p->sl_active.p,
                                   p->sl_nactive, (double*) p->rk.p, (double*) p->sl_rk_c.p, 0);
        }
    }
    GMX_HIP(hipGetLastError());
    GMX_HIP(hipDeviceSynchronize());
    return GMX_OK;
}

// Unsliced step: wave kernel + fix-up of rows spanning blocks (which also folds the diff partials).
template <typename S, int WAVES, int HOT>
static void launch_wave(gmx_pr* p, hipStream_t s) {
    const double N = (double) p->V;
    const double base = (1 - p->d) / N;
    S* next_owned = (S*) p->contrib[1 - p->cur].p + p->row_lo;
    // workgroups per CU: one with the LDS tile (it fills the LDS), else what the ~90 VGPRs admit (5 waves/SIMD)
    const int per_cu = HOT > 0 ? 1 : 5;
    int64_t grid = (int64_t) p->persistent_grid * per_cu;
    const int64_t need = (p->nblk + WAVES - 1) / WAVES;
    if (grid > need) grid = need;
    const int64_t n_main = grid * WAVES;   // per-wave |val-rank| partials
    hipLaunchKernelGGL((pr_wave_kernel<S, WAVES, HOT, true>), dim3((unsigned) grid), dim3(WAVES * 64), 0, s,
                       p->blk.p, p->nblk, p->rows, p->rb, p->ridx, p->outdeg.p, (S*) p->rk.p,
                       (const S*) p->contrib[p->cur].p, next_owned, base, p->d, p->part_first.p, p->part_last.p,
                       p->diff_part.p, p->Vpad);
    const int64_t n = p->nblk > n_main ? p->nblk : n_main;
    const int64_t fix_blocks = (n + 255) / 256;
    double* diff_fix = p->diff_part.p + p->nblk + 4096;
    hipLaunchKernelGGL(pr_fixup_kernel<S>, dim3((unsigned) fix_blocks), dim3(256), 0, s,
                       p->blk.p, p->nblk, p->rows, p->rb, p->outdeg.p, (S*) p->rk.p, next_owned, base, p->d,
                       p->part_first.p, p->part_last.p, (const double*) p->diff_part.p, n_main, diff_fix);
    hipLaunchKernelGGL(pr_diff_reduce_kernel, dim3(1), dim3(1024), 0, s, (const double*) diff_fix, (int64_t) 1, (int64_t) fix_blocks, (int64_t) fix_blocks, p->diff.p);
}

// Row-chunk boundaries over the exchanged prefix [0, exchange_count) of a rank's range, the same on every rank.
// Rows are in hotness order, so almost all edges belong to the first rows: chunk j covers the fractions
// [16^-(C-j), 16^-(C-j-1)) of the prefix (chunk 0 starts at 0), rounded to whole runs.  A step walks the chunks
// from the LAST (many rows, few edges -- most of the bytes to exchange are ready almost at once) to the first
// (the hubs: most of the sweep time, which then hides the exchange of everything else).  With C = 2 on an
// 8-rank partition of RMAT-26 the hub sixteenth of the prefix still holds ~70 % of the edges: its sweep covers
// the exchange of the other 94 % of the bytes, and only 6 % of the exchange is left exposed.
static int64_t pr_chunk_bound(const gmx_pr* p, int j) {
    const int C = p->nchunks;
    if (j <= 0) return 0;
    if (j >= C) return p->exchange_count;
    const int64_t unit = (int64_t) 1 << PR_RUN_SHIFT;
    int64_t b = p->exchange_count >> (4 * (C - j));
    b = (b + unit - 1) / unit * unit;
    return b < p->exchange_count ? b : p->exchange_count;
}

static int pr_build_chunks(gmx_pr* p, int C) {
    if (C < 1) C = 1;
    if (C > PR_MAX_CHUNKS) C = PR_MAX_CHUNKS;
    p->nchunks = C;
    for (int j = 0; j <= C; j++) {
        const int64_t b = pr_chunk_bound(p, j);
        p->ch_row[j] = (j == C || b > p->rows) ? p->rows : b;   // rows past the prefix ride with the last chunk
    }
    dbuf<int64_t> bounds, table;
    const int n = (C + 1) * (p->ns + 1);
    GMX_CHECK(bounds.alloc((size_t) C + 1));
    GMX_CHECK(table.alloc((size_t) n));
    GMX_HIP(hipMemcpy(bounds.p, p->ch_row, sizeof(int64_t) * (C + 1), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pr_chunk_table_kernel, dim3(1), dim3(128), 0, 0, p->sl, (const int32_t*) p->sl_active.p, p->sl_nactive,
                       (const int64_t*) bounds.p, C + 1, p->items, table.p);
    std::vector<int64_t> h((size_t) n);
    GMX_HIP(hipMemcpy(h.data(), table.p, sizeof(int64_t) * n, hipMemcpyDeviceToHost));
    for (int j = 0; j <= C; j++) {
        for (int q = 0; q < p->ns; q++) p->ch_blk[j][q] = (j == C) ? p->sl.s[q].nblk : (j == 0 ? 0 : h[(size_t) j * (p->ns + 1) + q]);
        p->ch_act[j] = (j == C) ? p->sl_nactive : (j == 0 ? 0 : h[(size_t) j * (p->ns + 1) + p->ns]);
    }
    if (pr_fused(p))   // the binned phases in the same pieces: bins by row chunk, tiles by "hub sources only"
        GMX_CHECK(pr_cold_set_parts(p->cold, C, p->ch_act, C > 1 ? pr_chunk_bound(p, 1) : p->exchange_count, p->exchange_count));
    if (getenv("GMX_PR_DEBUG"))
        for (int j = 0; j <= C; j++) {
            fprintf(stderr, "gmx pr chunk %d: row %lld active %lld blocks", j, (long long) p->ch_row[j], (long long) p->ch_act[j]);
            for (int q = 0; q < p->ns; q++) fprintf(stderr, " %lld", (long long) p->ch_blk[j][q]);
            fprintf(stderr, "\n");
        }
    return GMX_OK;
}

// Sliced step, one row chunk (c counts in processing order, i.e. from the last row chunk to the first):
// per-slice row sums of the chunk's blocks, fix-up of the rows that begin in them, then the combine pass over
// the chunk's rows.  The block holding a chunk boundary belongs to the LATER row chunk, which runs first, so
// everything a row needs has been computed when its chunk is combined.  The last call closes the step.
template <typename S>
static int launch_sliced_chunk(gmx_pr* p, int c, hipStream_t s) {
    const double N = (double) p->V;
    const double base = (1 - p->d) / N;
    const int C = p->nchunks;
    const int j = C - 1 - c;
    S* next_owned = (S*) p->contrib[1 - p->cur].p + p->row_lo;
    pr_sliced_args a = p->sl;
    if (pr_fused(p)) {
        // every edge is binned: the binned phases finish the rows themselves (no partial-sum array, no combine pass).
        // Phase 1 (unless gmx_pr_step_gather has enqueued it already), then phases 2-3 of this chunk's bins.
        const pr_cold_fuse fz{(const int32_t*) p->sl_active.p, (const int32_t*) p->sl_outdeg_c.p, (void*) p->sl_rk_c.p, (void*) next_owned, base, p->d};
        if (c == 0) {
            for (int k = 0; k < 2; k++)
                if (!(p->gather_mask & (1 << k))) GMX_CHECK(pr_cold_gather(p->cold, p->contrib[p->cur].p, k, s));
        }
        double* dfirst = p->diff_part.p + PR_COMBINE_GRID;
        if (c == 0 && p->cnt == 0) {   // first sweep after a reset: settle the rows without in-edges once (all chunks' rows)
            hipLaunchKernelGGL(pr_inactive_first_kernel<S>, dim3(PR_COMBINE_GRID), dim3(256), 0, s, (const uint8_t*) p->sl_is_active.p,
                               (int64_t) 0, p->rows, p->outdeg.p, (S*) p->rk.p, next_owned, base, p->d, dfirst);
            hipLaunchKernelGGL(pr_inactive_copy_kernel<S>, dim3(PR_COMBINE_GRID), dim3(256), 0, s, (const uint8_t*) p->sl_is_active.p,
                               p->rows, (S*) p->contrib[p->cur].p + p->row_lo, (const S*) next_owned);
        }
        if (pr_cold_parts(p->cold) == C) GMX_CHECK(pr_cold_accumulate(p->cold, &fz, c, s));
        else if (c == 0) GMX_CHECK(pr_cold_accumulate(p->cold, &fz, -1, s));
        if (c == C - 1) {
            int64_t nd = 0;
            const double* dp = pr_cold_diff_partials(p->cold, &nd);
            hipLaunchKernelGGL(pr_diff_reduce2_kernel, dim3(1), dim3(1024), 0, s, dp, nd, (const double*) dfirst,
                               (int64_t) (p->cnt == 0 ? PR_COMBINE_GRID : 0), p->diff.p);
        }
        return GMX_OK;
    }
    // the binned sources' row sums of ALL rows, once per step, before the first chunk is combined
    if (c == 0 && p->cold) GMX_CHECK(pr_cold_launch(p->cold, p->contrib[p->cur].p, nullptr, s));
    int64_t maxfix = 0, total_blk = 0;
    for (int q = 0; q < p->ns; q++) {
        pr_slice_desc& sd = a.s[q];
        sd.k_lo = sd.f_lo = p->ch_blk[j][q];
        sd.k_hi = sd.f_hi = p->ch_blk[j + 1][q];
        if (sd.f_hi - sd.f_lo > maxfix) maxfix = sd.f_hi - sd.f_lo;
        total_blk += sd.k_hi - sd.k_lo;
    }
    if (total_blk > 0) {
        int ns_shift = 0;
        while ((1 << ns_shift) < p->ns) ns_shift++;
        // the tile's slot arithmetic is shifts only: power-of-two slice count, and with several ranks
        // power-of-two range length and rank count (true for the RMAT benchmarks)
        int rank_shift = 0;
        while ((1LL << rank_shift) < p->slice) rank_shift++;
        const bool pow2 = (1 << ns_shift) == p->ns &&
                          (p->nranks == 1 || ((1LL << rank_shift) == p->slice && (p->nranks & (p->nranks - 1)) == 0));
        // claim size: about 8 dequeues per wave over the window, so the tail of a short window (many ranks,
        // row chunks) stays a small fraction of it
        auto set_qchunk = [&](int64_t grid, int waves_per_wg) {
            const int64_t waves = grid * waves_per_wg;
            const int64_t per_slice = waves / p->ns > 0 ? waves / p->ns : 1;
            // static deal: needs every slice to own the same number of whole workgroups per XCD
            const bool can_static = p->ns <= 8 && 8 % p->ns == 0 && grid % 8 == 0 && !getenv("GMX_PR_NO_STATIC");
            for (int q = 0; q < p->ns; q++) {
                pr_slice_desc& sd = a.s[q];
                const int64_t win = sd.k_hi - sd.k_lo;
                int64_t qc = win / (per_slice * 8);
                sd.qchunk = (int) (qc < 1 ? 1 : qc > PRW_QUEUE_CHUNK ? PRW_QUEUE_CHUNK : qc);
                sd.st_waves = (int) per_slice;
                sd.st_rounds = 0;
                if (can_static) sd.st_rounds = (int) (win * 7 / 8 / (per_slice * sd.qchunk));   // 7/8 static, the tail dynamic
                sd.st_total = (int64_t) sd.st_rounds * per_slice * sd.qchunk;
            }
        };
        if (p->hot && pow2) {
            constexpr int HOTQ = sizeof(S) == 4 ? 22528 : 7168;
            int64_t grid = p->persistent_grid;
            if (grid * 16 > total_blk) grid = (total_blk + 15) / 16;
            if (grid >= 8) grid -= grid % 8;
            set_qchunk(grid, 16);
            hipLaunchKernelGGL((pr_wave_sliced_kernel<S, 16, HOTQ, true>), dim3((unsigned) grid), dim3(1024), 0, s, a, p->Vpad,
                               (const S*) p->contrib[p->cur].p, ns_shift, rank_shift, p->nranks);
        } else {
            int64_t grid = (int64_t) p->persistent_grid * 5;
            if (grid * 4 > total_blk) grid = (total_blk + 3) / 4;
            if (grid >= 8) grid -= grid % 8;
            set_qchunk(grid, 4);
            hipLaunchKernelGGL((pr_wave_sliced_kernel<S, 4, 0, true>), dim3((unsigned) grid), dim3(256), 0, s, a, p->Vpad,
                               (const S*) p->contrib[p->cur].p, ns_shift, rank_shift, p->nranks);
        }
    }
    if (maxfix > 0)
        hipLaunchKernelGGL(pr_sliced_fixup_kernel<S>, dim3((unsigned) ((maxfix + 255) / 256), p->ns), dim3(256), 0, s, a, p->rows);
    double* dpart = p->diff_part.p + (int64_t) c * 2 * PR_COMBINE_GRID;
    if (p->Eh == 0) a.ns = 0;   // every edge is binned: nothing to add but the binned sums
    hipLaunchKernelGGL(pr_combine_kernel<S>, dim3(PR_COMBINE_GRID), dim3(256), 0, s, a, (const int32_t*) p->sl_active.p,
                       p->ch_act[j], p->ch_act[j + 1], (const int32_t*) p->sl_outdeg_c.p, (S*) p->sl_rk_c.p,
                       (const S*) pr_cold_partial(p->cold), next_owned, base, p->d, dpart);
    if (p->cnt == 0) {   // first sweep after a reset: settle the rows without in-edges once
        hipLaunchKernelGGL(pr_inactive_first_kernel<S>, dim3(PR_COMBINE_GRID), dim3(256), 0, s, (const uint8_t*) p->sl_is_active.p,
                           p->ch_row[j], p->ch_row[j + 1], p->outdeg.p, (S*) p->rk.p, next_owned, base, p->d,
                           dpart + PR_COMBINE_GRID);
        if (c == C - 1)
            hipLaunchKernelGGL(pr_inactive_copy_kernel<S>, dim3(PR_COMBINE_GRID), dim3(256), 0, s, (const uint8_t*) p->sl_is_active.p,
                               p->rows, (S*) p->contrib[p->cur].p + p->row_lo, (const S*) next_owned);
    }
    if (c == C - 1)   // per chunk: the combine partials, and in the first sweep those of the rows without in-edges
        hipLaunchKernelGGL(pr_diff_reduce_kernel, dim3(1), dim3(1024), 0, s, (const double*) p->diff_part.p, (int64_t) C,
                           (int64_t) 2 * PR_COMBINE_GRID, (int64_t) (p->cnt == 0 ? 2 : 1) * PR_COMBINE_GRID, p->diff.p);
    return GMX_OK;
}

extern "C" int gmx_pr_set_chunks(gmx_pr_t* p, int chunks) {
    GMX_REQUIRE(p, "pr is NULL");
    GMX_REQUIRE(chunks >= 1 && chunks <= PR_MAX_CHUNKS, "chunks must be in 1..%d", PR_MAX_CHUNKS);
    if (p->ns == 0) {   // the unsliced step is one piece
        p->nchunks = 1;
        return GMX_OK;
    }
    return pr_build_chunks(p, chunks);
}

extern "C" int gmx_pr_num_chunks(gmx_pr_t* p, int* chunks) {
    GMX_REQUIRE(p && chunks, "NULL argument");
    *chunks = p->nchunks;
    return GMX_OK;
}

extern "C" int gmx_pr_chunk_range(gmx_pr_t* p, int chunk, int64_t* offset, int64_t* count) {
    GMX_REQUIRE(p && offset && count, "NULL argument");
    GMX_REQUIRE(chunk >= 0 && chunk < p->nchunks, "chunk %d out of range", chunk);
    if (p->ns == 0) {
        *offset = 0;
        *count = p->exchange_count;
        return GMX_OK;
    }
    // the same on every rank (boundaries depend on exchange_count and the chunk count only): entries past a
    // short last rank's rows are padding that stays 0.  Chunks are numbered in processing order.
    const int j = p->nchunks - 1 - chunk;
    const int64_t lo = pr_chunk_bound(p, j), hi = pr_chunk_bound(p, j + 1);
    *offset = lo;
    *count = hi - lo;
    return GMX_OK;
}

extern "C" int gmx_pr_gather_classes(gmx_pr_t* p, int* classes) {
    GMX_REQUIRE(p && classes, "NULL argument");
    *classes = pr_fused(p) ? 2 : 0;
    return GMX_OK;
}

extern "C" int gmx_pr_gather_items(gmx_pr_t* p, int tile_class, int64_t* items) {
    GMX_REQUIRE(p && items, "NULL argument");
    *items = pr_fused(p) ? pr_cold_class_items(p->cold, tile_class) : 0;
    return GMX_OK;
}

// Binned, fused form only (gmx_pr_gather_classes > 0): enqueue phase 1 -- the only part of a step that reads the
// other ranks' contributions -- for one tile class ahead of gmx_pr_step_chunk(0).  Class 0: tiles all of whose live
// sources lie in the hub piece of a rank range (chunk_range of the LAST chunk), i.e. what the peers send last and
// what arrives first; class 1: the rest.  Class 0 opens the step.
extern "C" int gmx_pr_step_gather(gmx_pr_t* p, int cls, void* stream) {
    GMX_REQUIRE(p, "pr is NULL");
    GMX_REQUIRE(pr_fused(p), "gmx_pr_step_gather needs a plan with every edge binned (gmx_pr_gather_classes)");
    GMX_REQUIRE(cls == 0 || cls == 1, "tile class %d out of range", cls);
    GMX_REQUIRE(!(p->gather_mask & (1 << cls)) && (cls == 0 || (p->gather_mask & 1)), "tile classes are enqueued once per step, 0 before 1");
    hipStream_t s = (hipStream_t) stream;
    if (cls == 0) {
        pr_ev_begin(p, s);
        p->step_next = 1 - p->cur;
    }
    if (p->rows > 0) GMX_CHECK(pr_cold_gather(p->cold, p->contrib[p->cur].p, cls, s));
    p->gather_mask |= 1 << cls;
    GMX_HIP(hipGetLastError());
    return GMX_OK;
}

extern "C" int gmx_pr_step_chunk(gmx_pr_t* p, int chunk, void* stream) {
    GMX_REQUIRE(p, "pr is NULL");
    GMX_REQUIRE(chunk >= 0 && chunk < p->nchunks, "chunk %d out of range", chunk);
    hipStream_t s = (hipStream_t) stream;
    if (chunk == 0 && !(p->gather_mask & 1)) {
        pr_ev_begin(p, s);   // one "launch" of the roofline = all kernels of one step
        p->step_next = 1 - p->cur;
    }
    if (p->ns > 0) {
        if (p->rows > 0) {
            if (p->elem == 4) GMX_CHECK(launch_sliced_chunk<float>(p, chunk, s));
            else GMX_CHECK(launch_sliced_chunk<double>(p, chunk, s));
        }
    } else if (p->nblk > 0) {
        if (p->hot) {
            if (p->elem == 4) launch_wave<float, 16, 22528>(p, s);
            else launch_wave<double, 16, 7168>(p, s);
        } else {
            if (p->elem == 4) launch_wave<float, 4, 0>(p, s);
            else launch_wave<double, 4, 0>(p, s);
        }
    }
    GMX_HIP(hipGetLastError());
    if (chunk == 0) p->gather_mask = 0;
    if (chunk == p->nchunks - 1) {
        pr_ev_end(p, s);
        p->cur = 1 - p->cur;
        p->cnt++;
    }
    return GMX_OK;
}

extern "C" int gmx_pr_step(gmx_pr_t* p, void* stream) {
    GMX_REQUIRE(p, "pr is NULL");
    for (int c = 0; c < p->nchunks; c++) GMX_CHECK(gmx_pr_step_chunk(p, c, stream));
    return GMX_OK;
}

// Device pointer of the replica the CURRENT step writes (the exchange of a finished chunk targets it).
extern "C" int gmx_pr_contrib_next_full(gmx_pr_t* p, void** dev_ptr, int64_t* count) {
    GMX_REQUIRE(p && dev_ptr && count, "NULL argument");
    *dev_ptr = p->contrib[1 - p->cur].p;
    *count = p->Vpad;
    return GMX_OK;
}

// ------------------------------------------------------------------ peer push
// The exchange without a collective kernel: every rank maps the other ranks' replicas (hipIpc handles, passed
// around by the host side) and, as soon as a chunk of its own range is computed, copies that piece straight
// into each peer's replica -- one hipMemcpyAsync per peer on its own stream, i.e. the SDMA engines over the
// point-to-point xGMI links, all 7 at once, while the CUs go on with the next chunk.  Ordering between
// ranks (nobody reads a replica before every piece has landed, nobody overwrites one still being read) is
// the caller's per-step barrier: the all-reduce of `diff` the algorithm needs anyway.
extern "C" int gmx_ipc_export(void* dev_ptr, void* handle) {
    GMX_REQUIRE(dev_ptr && handle, "NULL argument");
    static_assert(sizeof(hipIpcMemHandle_t) == GMX_IPC_HANDLE_BYTES, "handle size");
    GMX_HIP(hipIpcGetMemHandle((hipIpcMemHandle_t*) handle, dev_ptr));
    return GMX_OK;
}

extern "C" int gmx_ipc_open(const void* handle, void** dev_ptr) {
    GMX_REQUIRE(dev_ptr && handle, "NULL argument");
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    GMX_HIP(hipIpcOpenMemHandle(dev_ptr, h, hipIpcMemLazyEnablePeerAccess));
    return GMX_OK;
}

extern "C" int gmx_ipc_close(void* dev_ptr) {
    GMX_REQUIRE(dev_ptr, "NULL argument");
    GMX_HIP(hipIpcCloseMemHandle(dev_ptr));
    return GMX_OK;
}

extern "C" int gmx_pr_contrib_buffers(gmx_pr_t* p, void** buf0, void** buf1, int64_t* bytes) {
    GMX_REQUIRE(p && buf0 && buf1 && bytes, "NULL argument");
    *buf0 = p->contrib[0].p;
    *buf1 = p->contrib[1].p;
    *bytes = p->Vpad * p->elem;
    return GMX_OK;
}

static int pr_ensure_push_streams(gmx_pr* p);
extern "C" int gmx_pr_set_peers(gmx_pr_t* p, void* const* peer_buf0, void* const* peer_buf1) {
    GMX_REQUIRE(p && peer_buf0 && peer_buf1, "NULL argument");
    for (int r = 0; r < p->nranks; r++)
        if (r != p->rank) GMX_REQUIRE(peer_buf0[r] && peer_buf1[r], "peer %d: NULL replica pointer", r);
    GMX_CHECK(pr_ensure_push_streams(p));
    p->peer_buf[0].assign((char* const*) peer_buf0, (char* const*) peer_buf0 + p->nranks);
    p->peer_buf[1].assign((char* const*) peer_buf1, (char* const*) peer_buf1 + p->nranks);
    return GMX_OK;
}

// entries [offset, offset+count) of this rank's range, from replica b to the same place in every peer's replica b
static int pr_push_range(gmx_pr* p, int b, int64_t offset, int64_t count, hipStream_t s, int set = 0) {
    if (p->nranks == 1) return GMX_OK;
    GMX_REQUIRE(!p->peer_buf[0].empty(), "gmx_pr_set_peers has not been called");
    GMX_REQUIRE(offset >= 0 && count >= 0 && offset + count <= p->slice, "push range outside the rank's range");
    if (count == 0) return GMX_OK;
    GMX_HIP(hipEventRecord(p->push_ready, s));
    const size_t at = (size_t) (p->row_lo + offset) * p->elem, bytes = (size_t) count * p->elem;
    for (int i = 1; i < p->nranks; i++) {
        const int r = (p->rank + i) % p->nranks;   // every rank starts with a different peer
        GMX_HIP(hipStreamWaitEvent(p->push_stream[set][r], p->push_ready, 0));
        GMX_HIP(hipMemcpyAsync(p->peer_buf[b][r] + at, p->contrib[b].p + at, bytes, hipMemcpyDeviceToDevice, p->push_stream[set][r]));
    }
    return GMX_OK;
}

extern "C" int gmx_pr_push_chunk(gmx_pr_t* p, int chunk, void* stream) {
    GMX_REQUIRE(p, "pr is NULL");
    int64_t off = 0, cnt = 0;
    GMX_CHECK(gmx_pr_chunk_range(p, chunk, &off, &cnt));
    return pr_push_range(p, p->step_next, off, cnt, (hipStream_t) stream, chunk & 1);
}

extern "C" int gmx_pr_push_current(gmx_pr_t* p, void* stream) {
    GMX_REQUIRE(p, "pr is NULL");
    return pr_push_range(p, p->cur, 0, p->exchange_count, (hipStream_t) stream);
}

static int pr_push_join_set(gmx_pr* p, int set, hipStream_t stream) {
    for (int r = 0; r < (int) p->push_stream[set].size(); r++) {
        if (!p->push_stream[set][r]) continue;
        GMX_HIP(hipEventRecord(p->push_done[set][r], p->push_stream[set][r]));
        GMX_HIP(hipStreamWaitEvent(stream, p->push_done[set][r], 0));
    }
    return GMX_OK;
}

extern "C" int gmx_pr_push_join(gmx_pr_t* p, void* stream) {
    GMX_REQUIRE(p, "pr is NULL");
    GMX_CHECK(pr_push_join_set(p, 0, (hipStream_t) stream));
    return pr_push_join_set(p, 1, (hipStream_t) stream);
}

// `stream` waits for the copies of ONE chunk (and whatever was queued before them on the same copy streams)
extern "C" int gmx_pr_push_join_chunk(gmx_pr_t* p, int chunk, void* stream) {
    GMX_REQUIRE(p, "pr is NULL");
    GMX_REQUIRE(chunk >= 0 && chunk < p->nchunks, "chunk %d out of range", chunk);
    return pr_push_join_set(p, chunk & 1, (hipStream_t) stream);
}

// ------------------------------------------------------------------ packed exchange
// "Send only what is read": see the members of gmx_pr.  The host side wires the ranks up like the plain push
// (gmx_pr_recv_buffers -> gmx_ipc_export -> transport -> gmx_ipc_open -> gmx_pr_set_peers_packed, plus the offsets each
// rank reports through gmx_pr_packed_info), then per chunk gmx_pr_push_packed after gmx_pr_step_chunk, the per-step
// barrier, and gmx_pr_unpack before the next step's phase 1 reads the replica.
extern "C" int gmx_pr_packed_info(gmx_pr_t* p, int64_t* send_counts, int64_t* recv_counts, int64_t* recv_offsets) {
    GMX_REQUIRE(p, "pr is NULL");
    if (!p->packed) return GMX_ERR_STATE;   // (not an error to report: the plan has no lists -- one rank, or not the degree order)
    for (int r = 0; r < p->nranks; r++) {
        if (send_counts) send_counts[r] = p->send_cnt[(size_t) r];
        if (recv_counts) recv_counts[r] = p->recv_cnt[(size_t) r];
        if (recv_offsets) recv_offsets[r] = p->recv_off[(size_t) r];
    }
    return GMX_OK;
}

extern "C" int gmx_pr_recv_buffers(gmx_pr_t* p, void** buf0, void** buf1, int64_t* bytes) {
    GMX_REQUIRE(p && buf0 && buf1 && bytes, "NULL argument");
    GMX_REQUIRE(p->packed, "the plan has no packed exchange lists");
    *buf0 = p->rbuf[0].p;
    *buf1 = p->rbuf[1].p;
    *bytes = (int64_t) p->rbuf[0].n;
    return GMX_OK;
}

static int pr_ensure_push_streams(gmx_pr* p) {
    if (!p->push_stream[0].empty()) return GMX_OK;
    for (int q = 0; q < 2; q++) {
        p->push_stream[q].assign((size_t) p->nranks, nullptr);
        p->push_done[q].assign((size_t) p->nranks, nullptr);
        for (int r = 0; r < p->nranks; r++) {
            if (r == p->rank) continue;
            GMX_HIP(hipStreamCreateWithFlags(&p->push_stream[q][r], hipStreamNonBlocking));
            GMX_HIP(hipEventCreateWithFlags(&p->push_done[q][r], hipEventDisableTiming));
        }
    }
    GMX_HIP(hipEventCreateWithFlags(&p->push_ready, hipEventDisableTiming));
    return GMX_OK;
}

// peer_recv0/1[r]: rank r's landing zones as mapped here; my_offset[r]: where MY segment starts in them (rank r's
// recv_offsets[me], elements); my_count[r] (optional): rank r's recv_counts[me], checked against what I would send
extern "C" int gmx_pr_set_peers_packed(gmx_pr_t* p, void* const* peer_recv0, void* const* peer_recv1, const int64_t* my_offset, const int64_t* my_count) {
    GMX_REQUIRE(p && peer_recv0 && peer_recv1 && my_offset, "NULL argument");
    GMX_REQUIRE(p->packed, "the plan has no packed exchange lists");
    for (int r = 0; r < p->nranks; r++)
        if (r != p->rank) {
            GMX_REQUIRE(peer_recv0[r] && peer_recv1[r], "peer %d: NULL landing zone", r);
            GMX_REQUIRE(!my_count || my_count[r] == p->send_cnt[(size_t) r],
                        "peer %d expects %lld entries from rank %d, which would send %lld", r, (long long) my_count[r], p->rank, (long long) p->send_cnt[(size_t) r]);
        }
    GMX_CHECK(pr_ensure_push_streams(p));
    p->peer_rbuf[0].assign((char* const*) peer_recv0, (char* const*) peer_recv0 + p->nranks);
    p->peer_rbuf[1].assign((char* const*) peer_recv1, (char* const*) peer_recv1 + p->nranks);
    p->peer_roff.assign(my_offset, my_offset + p->nranks);
    return GMX_OK;
}

// chunk boundaries inside the per-peer lists: sbound / rbound[r * (C + 1) + j] = first list index whose position is >= the
// j-th boundary of the exchanged prefix (row order; chunk c in processing order covers j = C - 1 - c)
static int64_t pr_chunk_bound(const gmx_pr* p, int j);
static int pr_packed_bounds(gmx_pr* p) {
    const int C = p->nchunks, N = p->nranks;
    if (p->bound_chunks == C && !p->sbound.empty()) return GMX_OK;
    std::vector<int64_t> q;
    for (int dir = 0; dir < 2; dir++)
        for (int r = 0; r < N; r++)
            for (int j = 0; j <= C; j++) {
                q.push_back(dir ? p->recv_off[(size_t) r] : p->send_off[(size_t) r]);
                q.push_back(dir ? p->recv_cnt[(size_t) r] : p->send_cnt[(size_t) r]);
                q.push_back(p->ns > 0 ? pr_chunk_bound(p, j) : (j == 0 ? 0 : p->exchange_count));
            }
    const int nq = (int) (q.size() / 3), half = nq / 2;
    dbuf<int64_t> dq, dout;
    GMX_CHECK(dq.alloc(q.size()));
    GMX_CHECK(dout.alloc((size_t) nq));
    GMX_HIP(hipMemcpy(dq.p, q.data(), sizeof(int64_t) * q.size(), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pr_list_bounds_kernel, dim3((half + 63) / 64), dim3(64), 0, 0, (const int32_t*) p->slist.p, (const int64_t*) dq.p, half, dout.p);
    hipLaunchKernelGGL(pr_list_bounds_kernel, dim3((half + 63) / 64), dim3(64), 0, 0, (const int32_t*) p->rlist.p, (const int64_t*) dq.p + 3 * half, half, dout.p + half);
    std::vector<int64_t> h((size_t) nq);
    GMX_HIP(hipMemcpy(h.data(), dout.p, sizeof(int64_t) * (size_t) nq, hipMemcpyDeviceToHost));
    p->sbound.assign(h.begin(), h.begin() + half);
    p->rbound.assign(h.begin() + half, h.end());
    p->bound_chunks = C;
    return GMX_OK;
}

// list index range of chunk `chunk` (-1: the whole list) for peer r
static void pr_packed_range(const gmx_pr* p, bool recv, int r, int chunk, int64_t* lo, int64_t* hi) {
    const int C = p->nchunks;
    const std::vector<int64_t>& b = recv ? p->rbound : p->sbound;
    const int j = chunk < 0 ? 0 : C - 1 - chunk, j1 = chunk < 0 ? C : j + 1;
    *lo = b[(size_t) r * (C + 1) + j];
    *hi = b[(size_t) r * (C + 1) + j1];
}

// pack the chunk's live entries of replica b for every peer and copy them into the peers' landing zones of parity b
static int pr_push_packed_range(gmx_pr* p, int b, int chunk, hipStream_t s, int set) {
    if (p->nranks == 1) return GMX_OK;
    GMX_REQUIRE(p->packed && !p->peer_rbuf[0].empty(), "gmx_pr_set_peers_packed has not been called");
    GMX_CHECK(pr_packed_bounds(p));
    pr_pack_args a;
    int64_t most = 0;
    for (int q = 0; q < p->nranks; q++) {
        a.off[q] = p->send_off[(size_t) q];
        pr_packed_range(p, false, q, chunk, &a.lo[q], &a.hi[q]);
        if (q == p->rank) a.hi[q] = a.lo[q] = 0;
        most = std::max(most, a.hi[q] - a.lo[q]);
    }
    if (most > 0) {
        const char* own = p->contrib[b].p + (size_t) p->row_lo * p->elem;
        const dim3 grid((unsigned) grid_for(most, 256, 1024), (unsigned) p->nranks);
        if (p->elem == 4) hipLaunchKernelGGL(pr_pack_kernel<float>, grid, dim3(256), 0, s, a, (const int32_t*) p->slist.p, (const float*) own, (float*) p->sbuf.p);
        else hipLaunchKernelGGL(pr_pack_kernel<double>, grid, dim3(256), 0, s, a, (const int32_t*) p->slist.p, (const double*) own, (double*) p->sbuf.p);
        GMX_HIP(hipGetLastError());
    }
    GMX_HIP(hipEventRecord(p->push_ready, s));
    for (int i = 1; i < p->nranks; i++) {
        const int q = (p->rank + i) % p->nranks;   // every rank starts with a different peer
        GMX_HIP(hipStreamWaitEvent(p->push_stream[set][q], p->push_ready, 0));
        const int64_t n = a.hi[q] - a.lo[q];
        if (n > 0)
            GMX_HIP(hipMemcpyAsync(p->peer_rbuf[b][q] + (size_t) (p->peer_roff[(size_t) q] + a.lo[q]) * p->elem,
                                   p->sbuf.p + (size_t) (a.off[q] + a.lo[q]) * p->elem, (size_t) n * p->elem, hipMemcpyDeviceToDevice, p->push_stream[set][q]));
    }
    return GMX_OK;
}

extern "C" int gmx_pr_push_packed(gmx_pr_t* p, int chunk, void* stream) {
    GMX_REQUIRE(p, "pr is NULL");
    GMX_REQUIRE(chunk >= -1 && chunk < p->nchunks, "chunk %d out of range", chunk);
    // chunk >= 0: the piece the running step has just finished (replica step_next); -1: the whole prefix of the current replica
    return pr_push_packed_range(p, chunk < 0 ? p->cur : p->step_next, chunk, (hipStream_t) stream, chunk < 0 ? 0 : (chunk & 1));
}

// scatter what the peers packed for chunk `chunk` (-1: everything) from the landing zone of the current replica's parity
// into the current replica (the one the next step reads).  Call after the barrier that follows the pushes.
extern "C" int gmx_pr_unpack(gmx_pr_t* p, int chunk, void* stream) {
    GMX_REQUIRE(p, "pr is NULL");
    if (p->nranks == 1) return GMX_OK;
    GMX_REQUIRE(p->packed, "the plan has no packed exchange lists");
    GMX_REQUIRE(chunk >= -1 && chunk < p->nchunks, "chunk %d out of range", chunk);
    GMX_CHECK(pr_packed_bounds(p));
    pr_pack_args a;
    int64_t most = 0;
    for (int r = 0; r < p->nranks; r++) {
        a.off[r] = p->recv_off[(size_t) r];
        pr_packed_range(p, true, r, chunk, &a.lo[r], &a.hi[r]);
        if (r == p->rank) a.hi[r] = a.lo[r] = 0;
        most = std::max(most, a.hi[r] - a.lo[r]);
    }
    if (most == 0) return GMX_OK;
    const int b = p->cur;
    const dim3 grid((unsigned) grid_for(most, 256, 1024), (unsigned) p->nranks);
    if (p->elem == 4) hipLaunchKernelGGL(pr_unpack_kernel<float>, grid, dim3(256), 0, (hipStream_t) stream, a, (const int32_t*) p->rlist.p, (const float*) p->rbuf[b].p, (float*) p->contrib[b].p, p->slice);
    else hipLaunchKernelGGL(pr_unpack_kernel<double>, grid, dim3(256), 0, (hipStream_t) stream, a, (const int32_t*) p->rlist.p, (const double*) p->rbuf[b].p, (double*) p->contrib[b].p, p->slice);
    GMX_HIP(hipGetLastError());
    return GMX_OK;
}

// the positions (inside rank r's range) this rank reads: device pointer + count (for checks)
extern "C" int gmx_pr_recv_list(gmx_pr_t* p, int r, void** dev_ptr, int64_t* count) {
    GMX_REQUIRE(p && dev_ptr && count && r >= 0 && r < p->nranks, "bad argument");
    GMX_REQUIRE(p->packed, "the plan has no packed exchange lists");
    *dev_ptr = p->rlist.p + p->recv_off[(size_t) r];
    *count = p->recv_cnt[(size_t) r];
    return GMX_OK;
}

extern "C" int gmx_pr_contrib_slice(gmx_pr_t* p, void** dev_ptr, int64_t* count) {
    GMX_REQUIRE(p && dev_ptr && count, "NULL argument");
    *dev_ptr = p->contrib[p->cur].p + (size_t) p->row_lo * p->elem;
    *count = p->slice;
    return GMX_OK;
}

extern "C" int gmx_pr_exchange_count(gmx_pr_t* p, int64_t* count) {
    GMX_REQUIRE(p && count, "NULL argument");
    *count = p->exchange_count;
    return GMX_OK;
}

extern "C" int gmx_pr_contrib_full(gmx_pr_t* p, void** dev_ptr, int64_t* count) {
    GMX_REQUIRE(p && dev_ptr && count, "NULL argument");
    *dev_ptr = p->contrib[p->cur].p;
    *count = p->Vpad;
    return GMX_OK;
}

extern "C" int gmx_pr_diff_ptr(gmx_pr_t* p, void** dev_ptr) {
    GMX_REQUIRE(p && dev_ptr, "NULL argument");
    *dev_ptr = p->diff.p;
    return GMX_OK;
}

extern "C" int gmx_pr_diff(gmx_pr_t* p, void* stream, double* diff) {
    GMX_REQUIRE(p && diff, "NULL argument");
    // the emitted loop reads diff every iteration: through pinned memory, so the read-back is one small DMA and
    // not a staged pageable copy
    if (!p->h_diff) GMX_HIP(hipHostMalloc((void**) &p->h_diff, sizeof(double), hipHostMallocDefault));
    GMX_HIP(hipMemcpyAsync(p->h_diff, p->diff.p, sizeof(double), hipMemcpyDeviceToHost, (hipStream_t) stream));
    GMX_HIP(hipStreamSynchronize((hipStream_t) stream));
    *diff = *p->h_diff;
    return GMX_OK;
}

extern "C" int gmx_pr_download(gmx_pr_t* p, void* rank_host) {
    GMX_REQUIRE(p && rank_host, "NULL argument");
    GMX_HIP(hipDeviceSynchronize());
    if (p->rows == 0) return GMX_OK;
    // un-permute on the device into original order, then copy the touched entries
    dbuf<char> tmp;
    GMX_CHECK(tmp.alloc((size_t) p->V * p->elem));
    if (p->nranks > 1) GMX_HIP(hipMemcpy(tmp.p, rank_host, (size_t) p->V * p->elem, hipMemcpyHostToDevice));
    if (p->ns > 0 && p->sl_nactive > 0) {   // the sliced step keeps the ranks of the rows with in-edges densely
        if (p->elem == 4)
            hipLaunchKernelGGL(pr_rk_dense_kernel<float>, dim3(grid_for(p->sl_nactive)), dim3(256), 0, 0, (const int32_t*) p->sl_active.p,
                               p->sl_nactive, (float*) p->rk.p, (float*) p->sl_rk_c.p, 1);
        else
            hipLaunchKernelGGL(pr_rk_dense_kernel<double>, dim3(grid_for(p->sl_nactive)), dim3(256), 0, 0, (const int32_t*) p->sl_active.p,
                               p->sl_nactive, (double*) p->rk.p, (double*) p->sl_rk_c.p, 1);
    }
    if (p->elem == 4)
        hipLaunchKernelGGL(pr_unpermute_kernel<float>, dim3(grid_for(p->rows)), dim3(256), 0, 0, p->rows, p->inv.p, (const float*) p->rk.p, (float*) tmp.p);
    else
        hipLaunchKernelGGL(pr_unpermute_kernel<double>, dim3(grid_for(p->rows)), dim3(256), 0, 0, p->rows, p->inv.p, (const double*) p->rk.p, (double*) tmp.p);
    GMX_HIP(hipGetLastError());
    GMX_HIP(hipMemcpy(rank_host, tmp.p, (size_t) p->V * p->elem, hipMemcpyDeviceToHost));
    return GMX_OK;
}

// Variant choice by size (measured on RMAT, edge factor 16, fp32; ms per iteration):
//          plain   LDS tile  sliced pull  binned (tile-gather / bin-accumulate, gmx_pr_cold.hip)
// 2^16     0.018   0.028     0.056
// 2^18     0.033   0.041     0.084
// 2^19     0.060   0.057     0.090        0.068
// 2^20     0.098   0.086     0.115        0.083
// 2^21     0.197   0.156     0.152        0.121
// 2^22                       0.264        0.177
// 2^24                       1.04         0.49
// 2^26                       4.80         1.70
// Up to 2^18 vertices the whole contribution vector sits in every L2 and filling an LDS tile per launch costs
// more than it saves; up to 2^20 the unsliced kernel with the tile wins; beyond, every edge goes through the
// bins (GMX_PR_SLICED only provides the row bookkeeping they share with the sliced pull sweep, which remains
// reachable by leaving GMX_PR_COLD_PB out).
extern "C" uint32_t gmx_pr_default_options(int64_t V, int nranks) {
    uint32_t o = GMX_PR_RELABEL;
    if (V > (1LL << 18)) o |= GMX_PR_HOT_LDS;   // with several ranks the tile is only used by the sliced kernel
    if (V > (1LL << 20)) o |= GMX_PR_SLICED | GMX_PR_COLD_PB;
    return o;
}

extern "C" int gmx_pr_timing(gmx_pr_t* p, int enable) {
    GMX_REQUIRE(p, "pr is NULL");
    p->timing = enable != 0;
    p->ev_used = 0;
    return GMX_OK;
}

extern "C" int gmx_pr_kernel_time(gmx_pr_t* p, int32_t* launches, double* mean_ms) {
    GMX_REQUIRE(p && launches && mean_ms, "NULL argument");
    GMX_HIP(hipDeviceSynchronize());
    double tot = 0.0;
    for (int i = 0; i < p->ev_used; i++) {
        float ms = 0;
        GMX_HIP(hipEventElapsedTime(&ms, p->ev[2 * i], p->ev[2 * i + 1]));
        tot += ms;
    }
    *launches = p->ev_used;
    *mean_ms = p->ev_used ? tot / p->ev_used : 0.0;
    return GMX_OK;
}

extern "C" const char* gmx_pr_kernel_name(gmx_pr_t* p) {
    if (!p) return "";
    if (p->ns > 0 && p->cold && p->Eh == 0)
        return "pr_cold_tile_kernel+pr_cold_accum_kernel+pr_cold_reduce_few_kernel+pr_cold_reduce_kernel+pr_diff_reduce2_kernel";
    if (p->ns > 0 && p->cold)
        return "pr_cold_tile_kernel+pr_cold_accum_kernel+pr_cold_reduce_few_kernel+pr_cold_reduce_kernel+pr_wave_sliced_kernel+pr_sliced_fixup_kernel+pr_combine_kernel+pr_diff_reduce_kernel";
    if (p->ns > 0) return "pr_wave_sliced_kernel+pr_sliced_fixup_kernel+pr_combine_kernel+pr_diff_reduce_kernel";
    return "pr_wave_kernel+pr_fixup_kernel+pr_diff_reduce_kernel";
}

extern "C" int gmx_pr_work(gmx_pr_t* p, int64_t* edges, int64_t* rows, int64_t* algorithmic_bytes) {
    GMX_REQUIRE(p, "pr is NULL");
    if (edges) *edges = p->El;
    if (rows) *rows = p->rows_real;
    // E*(4+s) + V*(8+3s)
    if (algorithmic_bytes) *algorithmic_bytes = p->El * (4 + p->elem) + p->rows_real * (8 + 3 * (int64_t) p->elem);
    return GMX_OK;
}

extern "C" int gmx_pr_cold_info(gmx_pr_t* p, int64_t* hot_ids, int64_t* cold_edges, int64_t* padded_items) {
    GMX_REQUIRE(p, "pr is NULL");
    if (hot_ids) *hot_ids = p->cold ? p->cold_T : -1;
    if (cold_edges) *cold_edges = pr_cold_edges(p->cold);
    if (padded_items) *padded_items = pr_cold_items(p->cold);
    return GMX_OK;
}

// ------------------------------------------------------------------ whole-kernel entries
template <typename S>
static int pagerank_entry(gmx_graph_t* g, double e, double d, int32_t max_iter, S* rank_host, gmx_stats_t* stats) {
    GMX_REQUIRE(g && rank_host, "NULL argument");
    if (stats) memset(stats, 0, sizeof(*stats));
    if (g->V == 0) return GMX_OK;
    // several GPUs (or GMX_PR_RANKS > 1): one host thread drives N rank states (gmx_pr_multi.hip); the plan is cached
    // on the graph like the single-GPU one
    // d outside (0, 1] (accepted by the reference driver, pagerank_main.cc:59-62): contributions leave [0, 1], so the
    // fixed-point bins are out; the pull sweep sums in floating point
    const bool plain_d = d > 0.0 && d <= 1.0;
    const int nranks = (g->pr_multi_refused || !plain_d) ? 1 : gmx_pr_multi_ranks(g);
    const char* ex = getenv("GMX_EXCHANGE");
    if (!g->pr_multi_refused && plain_d && (nranks > 1 || (ex && *ex && strcmp(ex, "peer") != 0))) {   // (an RCCL exchange can be asked for with one rank too)
        gmx_pr_multi*& mc = g->pr_multi_cache[sizeof(S) == 4 ? 0 : 1];
        if (mc == nullptr) GMX_CHECK(gmx_pr_multi_create(g, (int) sizeof(S), nranks, &mc));
        const int mst = gmx_pr_multi_run(mc, e, d, max_iter, (void*) rank_host, stats);
        if (mst == GMX_OK || gmx_pr_multi_verified(mc)) return mst;
        // first contact failed (the replicas are not what the owners hold): say so and compute on one device
        fprintf(stderr, "gmx: %s -- falling back to the single-GPU pagerank\n", gmx_last_error());
        gmx_pr_multi_free(mc);
        mc = nullptr;
        g->pr_multi_refused = true;
        GMX_HIP(hipSetDevice(g->device));
    }
    gmx_pr_t*& cached = g->pr_cache[(sizeof(S) == 4 ? 0 : 1) + (plain_d ? 0 : 2)];
    if (cached == nullptr)
        GMX_CHECK(gmx_pr_create(g, (int) sizeof(S), 0, 1, gmx_pr_default_options(g->V, 1) & ~(plain_d ? 0u : (uint32_t) GMX_PR_COLD_PB), &cached));
    gmx_pr_t* p = cached;
    int st = gmx_pr_reset(p, d);
    if (st == GMX_ERR_STATE && sizeof(S) == 4) {
        // the fp32 plan's single fixed-point limb cannot hold the bar here (pr_cold_limb_guard): compute with the
        // two-limb fp64 plan and round once per vertex
        std::vector<double> tmp((size_t) g->V);
        GMX_CHECK(pagerank_entry<double>(g, e, d, max_iter, tmp.data(), stats));
        for (int64_t i = 0; i < g->V; i++) rank_host[i] = (S) tmp[(size_t) i];
        return GMX_OK;
    }
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double diff = 0.0;
    int32_t cnt = 0;
    if (st == GMX_OK) {
        (void) hipEventCreate(&ev0);
        (void) hipEventCreate(&ev1);
        (void) hipEventRecord(ev0, 0);
        // do { ... cnt++; } while ((diff > e) && (cnt < max));   pagerank.gm:9-19
        do {
            if ((st = gmx_pr_step(p, nullptr)) != GMX_OK) break;
            if ((st = gmx_pr_diff(p, nullptr, &diff)) != GMX_OK) break;
            cnt++;
        } while ((diff > e) && (cnt < max_iter));
        (void) hipEventRecord(ev1, 0);
        (void) hipEventSynchronize(ev1);
    }
    if (st == GMX_OK) {
        float ms = 0;
        (void) hipEventElapsedTime(&ms, ev0, ev1);
        hipEvent_t c0, c1;
        (void) hipEventCreate(&c0);
        (void) hipEventCreate(&c1);
        (void) hipEventRecord(c0, 0);
        st = gmx_pr_download(p, rank_host);
        (void) hipEventRecord(c1, 0);
        (void) hipEventSynchronize(c1);
        float cms = 0;
        (void) hipEventElapsedTime(&cms, c0, c1);
        (void) hipEventDestroy(c0);
        (void) hipEventDestroy(c1);
        if (stats) {
            stats->iterations = cnt;
            stats->last_diff = diff;
            stats->kernel_ms = ms;
            stats->d2h_ms = cms;
        }
    }
    if (ev0) (void) hipEventDestroy(ev0);
    if (ev1) (void) hipEventDestroy(ev1);
    return st;
}

extern "C" int gmx_pagerank_f64(gmx_graph_t* g, double e, double d, int32_t max_iter, double* rank_host, gmx_stats_t* stats) {
    return pagerank_entry<double>(g, e, d, max_iter, rank_host, stats);
}

extern "C" int gmx_pagerank_f32(gmx_graph_t* g, float e, float d, int32_t max_iter, float* rank_host, gmx_stats_t* stats) {
    return pagerank_entry<float>(g, (double) e, (double) d, max_iter, rank_host, stats);
}

// Loads this translation unit's code object (the HIP runtime does that lazily, at the first launch of one of its kernels:
// tens of milliseconds that would otherwise fall into the first timed call) -- called once from the graph constructors.
void gmx_touch_pagerank() {
    hipFuncAttributes attr;
    (void) hipFuncGetAttributes(&attr, (const void*) pr_degkey_kernel);
}
