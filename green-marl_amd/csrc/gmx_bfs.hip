// gmx_bfs.hip -- hop_dist (BFS depth along out-edges) for gfx950.
//
// Replaces the body of the emitted `hop_dist` (source /root/reference/apps/src/hop_dist.gm:3-31;
// restated emission SURVEY.md section 8 a-2): a level-synchronous push where every updated
// vertex relaxes dist_nxt of its out-neighbours under a per-node lock, INT_MAX = unreached.
// The result (BFS depth from root) is unique, so the device version is free to choose the
// traversal: direction-optimising as the reference's own runtime BFS does
// (apps/output_cpp/gm_graph/inc/gm_bfs_template.h:352-421):
//   * top-down:  frontier queue; the out-degrees of the frontier are prefix-summed and the
//                (vertex, edge) sequence is cut by merge-path into equal pieces, so a hub row is
//                spread over the whole chip and consecutive lanes read consecutive node_idx
//                entries; first-writer-wins via atomicMin on dist (the emitted `min=`), next
//                queue built with __ballot-aggregated appends;
//   * bottom-up: when the frontier exceeds 5 % of V (RRD_THRESHOLD, gm_bfs_template.h:359),
//                every unvisited vertex scans its in-row for a parent whose bit is set in the
//                frontier bitmap (V/8 bytes: fits L2) and stops at the first hit; the vertices found
//                are reported as a bitmap (one ballot per wave), applied to dist[] and reused as the
//                next level's frontier -- and, with several GPUs, exchanged slice by slice.
// Integer only: bit-exact against the CPU result by construction.
#include "gmx_internal.h"

#include <float.h>
#include <limits.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <rocprim/rocprim.hpp>

#define BFS_THREADS 256

// Statistics that thousands of waves add to (edges inspected, vertices found by a bottom-up level) are spread over
// BFS_SHARDS cache lines and summed by the host after the read-back: atomics on ONE line retire at ~90 per
// microsecond device-wide, and two such adds per wave were 0.37 ms of a 0.39 ms bottom-up level at RMAT-20 (and
// ~0.7 ms per bottom-up level at RMAT-26).  Both run on: a level's count is the difference to the previous total.
#define BFS_SHARDS 64
struct bfs_counters {
    unsigned long long next_count;     // top-down: tail of the next queue = vertices discovered in this level
    unsigned long long next_edges;     // their out-edges: the next level's merge-path length and the input of the
                                       // direction decision, known without a pass over the new queue
    unsigned long long pad0[14];
    struct {
        unsigned long long edges;      // edges inspected so far
        unsigned long long found;      // vertices found by bottom-up levels so far
        unsigned long long pad[14];
    } shard[BFS_SHARDS];
};
// A level's totals for the host: summed on the device and written straight into pinned host memory, the level's tag
// last.  The host spins on the tag -- a stream synchronisation takes ~20 us to wake up, which at small scales was
// most of a level.
struct bfs_level_totals {
    unsigned long long next_count, next_edges, edges, found;
    volatile unsigned long long tag;
};
__device__ __forceinline__ void bfs_count(bfs_counters* __restrict__ ctr, unsigned long long edges, unsigned long long found) {
    const int sh = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (BFS_SHARDS - 1);
    if (edges) atomicAdd(&ctr->shard[sh].edges, edges);
    if (found) atomicAdd(&ctr->shard[sh].found, found);
}
__global__ void bfs_totals_kernel(const bfs_counters* __restrict__ ctr, bfs_level_totals* __restrict__ out, unsigned long long tag) {
    const int lane = threadIdx.x;   // 64 threads, one shard each
    unsigned long long e = ctr->shard[lane].edges, f = ctr->shard[lane].found;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        e += __shfl_down(e, o, 64);
        f += __shfl_down(f, o, 64);
    }
    if (lane == 0) {
        out->next_count = ctr->next_count;
        out->next_edges = ctr->next_edges;
        out->edges = e;
        out->found = f;
        __threadfence_system();
        out->tag = tag;
    }
}
static void bfs_totals(const bfs_counters& h, unsigned long long* edges, unsigned long long* found) {
    unsigned long long e = 0, f = 0;
    for (int i = 0; i < BFS_SHARDS; i++) {
        e += h.shard[i].edges;
        f += h.shard[i].found;
    }
    *edges = e;
    *found = f;
}

// everything a traversal starts from, in one launch: dist[], both bitmaps, the counters, the root in the first queue
// and its out-degree on its way to the host (next_edges of "level -1", tag 0 of the run)
__global__ void bfs_init_kernel(int32_t* __restrict__ dist, int64_t V, int32_t root, unsigned long long* __restrict__ bm0,
                                unsigned long long* __restrict__ bm1, int64_t words, bfs_counters* __restrict__ ctr,
                                int32_t* __restrict__ q0, const int32_t* __restrict__ begin, bfs_level_totals* __restrict__ out,
                                unsigned long long tag) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    if (i < (int64_t) (sizeof(bfs_counters) / sizeof(unsigned long long))) ((unsigned long long*) ctr)[i] = 0ull;
    if (i == 0) {
        if (root >= 0) q0[0] = root;
        out->next_count = root >= 0 ? 1 : 0;
        out->next_edges = root >= 0 ? (unsigned long long) (begin[root + 1] - begin[root]) : 0ull;
        out->edges = out->found = 0;
        __threadfence_system();
        out->tag = tag;
    }
    for (int64_t w = i; w < words; w += stride) bm0[w] = bm1[w] = 0ull;
    for (; i < V; i += stride) dist[i] = (i == root) ? 0 : INT_MAX;
}

#define BFS_ITEMS 2048   // merge-path items (frontier vertices + their out-edges) per workgroup

__global__ void bfs_degree_kernel(const int32_t* __restrict__ begin, const int32_t* __restrict__ q, int64_t n,
                                  int32_t* __restrict__ deg) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        int32_t v = q[i];
        deg[i] = begin[v + 1] - begin[v];
    }
}

// A small frontier (n <= BFS_SMALL_SCAN): degrees and their exclusive prefix sums off[0..n] in ONE launch of one
// workgroup -- the library scan is two launches behind a degree kernel, and with the two read-backs per level that
// made a level of a small graph cost ~150 us of launches and round trips.  Also clears the counters of the level
// that is about to run (saves a memset).
#define BFS_SMALL_CHUNK 16384   // items per pass of the workgroup
#define BFS_SMALL_PER (BFS_SMALL_CHUNK / 1024)
// (one pass: a full pass is ~30 us -- 16 K dependent gathers of begin[] from ONE compute unit -- so a second pass already
// loses against the library scan's three launches; measured at 36.5 K items: 90 us against 49 us)
#define BFS_SMALL_SCAN BFS_SMALL_CHUNK
__global__ void __launch_bounds__(1024)
bfs_degree_scan_small_kernel(const int32_t* __restrict__ begin, const int32_t* __restrict__ q, int n,
                             int64_t* __restrict__ off, bfs_counters* __restrict__ ctr) {
    __shared__ long long s_wave[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    long long carry = 0;   // degrees of the passes before this one
    for (int c0 = 0; c0 < n || c0 == 0; c0 += BFS_SMALL_CHUNK) {
        long long d[BFS_SMALL_PER], sum = 0;
#pragma unroll
        for (int j = 0; j < BFS_SMALL_PER; j++) {
            const int i = c0 + tid * BFS_SMALL_PER + j;
            d[j] = 0;
            if (i < n) {
                const int32_t v = q[i];
                d[j] = begin[v + 1] - begin[v];
            }
            sum += d[j];
        }
        long long incl = sum;   // inclusive scan of the threads' sums: inside the wave, then over the 16 waves
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const long long t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (c0) __syncthreads();   // (the previous pass has read s_wave)
        if (lane == 63) s_wave[wv] = incl;
        __syncthreads();
        long long wbase = carry, total = 0;
        for (int w = 0; w < 16; w++) {
            if (w < wv) wbase += s_wave[w];
            total += s_wave[w];
        }
        long long run = wbase + incl - sum;
#pragma unroll
        for (int j = 0; j < BFS_SMALL_PER; j++) {
            const int i = c0 + tid * BFS_SMALL_PER + j;
            if (i < n) off[i] = run;
            run += d[j];
            if (i == n - 1) off[n] = run;
        }
        carry += total;
    }
    if (tid == 0) {
        if (n == 0) off[0] = 0;
        ctr->next_count = 0;
        ctr->next_edges = 0;
    }
}

// rows consumed at the merge-path diagonals k * BFS_ITEMS, k = 0 .. nb, of (n frontier rows, m edges): one thread per
// diagonal.  Searched inside the level kernels -- two threads, log2(n) dependent loads over off[], the other 254 at the barrier
// -- it was most of a workgroup's time once n is in the millions (an sssp round at RMAT-24: ~20 of ~25 us per workgroup).
__global__ void bfs_merge_split_kernel(const int64_t* __restrict__ off, int64_t n, int64_t m, int64_t nb, int64_t* __restrict__ split) {
    const int64_t k = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (k > nb) return;
    int64_t dk = k * BFS_ITEMS;
    if (dk > n + m) dk = n + m;
    int64_t lo = dk > m ? dk - m : 0, hi = dk < n ? dk : n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (off[mid + 1] <= dk - mid - 1) lo = mid + 1; else hi = mid;
    }
    split[k] = lo;
}

// off[0..n] = exclusive prefix sums of the frontier degrees (off[n] = frontier edges).
// Block k handles path items [k*BFS_ITEMS, (k+1)*BFS_ITEMS) of the merged (vertex ends, edges) sequence.
__global__ void __launch_bounds__(BFS_THREADS)
bfs_topdown_kernel(const int32_t* __restrict__ begin, const int32_t* __restrict__ node_idx,
                   const int32_t* __restrict__ cur_q, int64_t n, const int64_t* __restrict__ off, int64_t m,
                   int32_t level, int32_t* __restrict__ dist, int32_t* __restrict__ next_q,
                   bfs_counters* __restrict__ ctr) {
    __shared__ int64_t s_off[BFS_ITEMS + 2];
    __shared__ int32_t s_row[BFS_ITEMS + 2];
    __shared__ int64_t s_split[2][2];
    // the vertices this workgroup discovers wait here: the queue tail is claimed ONCE per workgroup (a claim per wave
    // and round -- two atomics on one cache line -- was most of a top-down level out of a hub: 42 K waves, ~90 per us)
    __shared__ int32_t s_win[BFS_ITEMS];
    __shared__ unsigned int s_nwin;
    __shared__ unsigned long long s_deg, s_base;
    const int tid = threadIdx.x;
    if (tid == 0) {
        s_nwin = 0;
        s_deg = 0;
    }
    if (tid < 2) {   // merge-path split of diagonals k*ITEMS and (k+1)*ITEMS
        int64_t dk = ((int64_t) blockIdx.x + tid) * BFS_ITEMS;
        if (dk > n + m) dk = n + m;
        int64_t lo = dk > m ? dk - m : 0, hi = dk < n ? dk : n;
        while (lo < hi) {
            int64_t mid = (lo + hi) >> 1;
            if (off[mid + 1] <= dk - mid - 1) lo = mid + 1; else hi = mid;
        }
        s_split[tid][0] = lo;
        s_split[tid][1] = dk - lo;
    }
    __syncthreads();
    const int64_t v0 = s_split[0][0], e0 = s_split[0][1], v1 = s_split[1][0], e1 = s_split[1][1];
    const int nv = (int) (v1 - v0) + 1;   // frontier slots touched (the last one may be partial / == n)
    for (int i = tid; i < nv; i += BFS_THREADS) {
        int64_t vi = v0 + i;
        s_off[i] = vi <= n ? off[vi < n ? vi : n] : m;
        s_row[i] = vi < n ? begin[cur_q[vi]] : 0;
    }
    if (tid == 0) s_off[nv] = m + 1;   // sentinel
    __syncthreads();
    unsigned long long inspected = 0, deg = 0;
    // A thread's edges (at most BFS_ITEMS / BFS_THREADS) go through the four dependent accesses of an edge -- its slot,
    // dist[] of the neighbour, the atomicMin, the winner's out-degree -- TOGETHER, one access kind at a time: edge by
    // edge the level out of a hub (RMAT-26 from vertex 0: 0.98 M edges) was 8 x 4 round trips per thread, 75 us.
    constexpr int K = BFS_ITEMS / BFS_THREADS;
    if (e1 > e0) {   // (workgroup-uniform)
        int32_t sv[K], dv[K], old[K];
        bool valid[K], won[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int64_t x = e0 + tid + (int64_t) k * BFS_THREADS;
            valid[k] = x < e1;
            const int64_t xc = valid[k] ? x : e0;   // (loads stay unconditional: a slot that exists)
            // frontier slot of edge x: last i with s_off[i] <= x
            int lo = 0, hi = nv - 1;
            while (lo < hi) {
                int mid = (lo + hi + 1) >> 1;
                if (s_off[mid] <= xc) lo = mid; else hi = mid - 1;
            }
            sv[k] = node_idx[(int64_t) s_row[lo] + (xc - s_off[lo])];
            inspected += valid[k];
        }
#pragma unroll
        for (int k = 0; k < K; k++) dv[k] = dist[sv[k]];
        // <s.dist_nxt; s.updated_nxt> min= <n.dist + 1; True>   (hop_dist.gm:21)
#pragma unroll
        for (int k = 0; k < K; k++) {
            old[k] = 0;
            if (valid[k] && dv[k] == INT_MAX) old[k] = atomicMin(&dist[sv[k]], level + 1);
        }
#pragma unroll
        for (int k = 0; k < K; k++) won[k] = valid[k] && dv[k] == INT_MAX && old[k] == INT_MAX;
        int32_t dg[K];
#pragma unroll
        for (int k = 0; k < K; k++) {   // (unconditional, the losers read row 0: no branch, so the loads overlap)
            const int32_t r = won[k] ? sv[k] : 0;
            dg[k] = begin[r + 1] - begin[r];
        }
#pragma unroll
        for (int k = 0; k < K; k++) {
            const unsigned long long mw = __ballot(won[k]);
            if (mw) {
                const int lane = tid & 63;
                const int leader = __ffsll((long long) mw) - 1;
                unsigned int at = 0;
                if (lane == leader) at = atomicAdd(&s_nwin, (unsigned int) __popcll(mw));
                at = __shfl(at, leader, 64);
                if (won[k]) {
                    s_win[at + __popcll(mw & ((1ULL << lane) - 1))] = sv[k];
                    deg += (unsigned long long) dg[k];
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        inspected += __shfl_down(inspected, o, 64);
        deg += __shfl_down(deg, o, 64);
    }
    if ((tid & 63) == 0) {
        bfs_count(ctr, inspected, 0);
        if (deg) atomicAdd(&s_deg, deg);
    }
    __syncthreads();
    const unsigned int nwin = s_nwin;
    if (nwin == 0) return;   // (workgroup-uniform)
    if (tid == 0) {
        s_base = atomicAdd(&ctr->next_count, (unsigned long long) nwin);
        if (s_deg) atomicAdd(&ctr->next_edges, s_deg);
    }
    __syncthreads();
    for (unsigned int i = tid; i < nwin; i += BFS_THREADS) next_q[s_base + i] = s_win[i];
}

// A SPARSE frontier (about one out-edge per vertex: the tail levels of a traversal): one vertex per lane, no degree pass,
// no scan, no merge-path.  RMAT-26's fifth level (36.5 K vertices, 36.0 K edges) cost 4 + ~45 (library scan) + 22 us of
// kernels and launches for what is one round of gathers.  A lane walks up to BFS_SPARSE_OWN edges itself; what is left of
// a longer row is taken by the whole wave afterwards, 64 edges per step, so no lane is ever alone with a hub.
#define BFS_SPARSE_OWN 16
__device__ __forceinline__ void bfs_sparse_visit(bool on, int32_t s, int32_t level, int32_t* __restrict__ dist, const int32_t* __restrict__ begin,
                                                 int32_t* __restrict__ next_q, bfs_counters* __restrict__ ctr, int lane,
                                                 unsigned long long& deg) {
    bool won = false;
    if (on && dist[s] == INT_MAX) won = atomicMin(&dist[s], level + 1) == INT_MAX;
    const unsigned long long mw = __ballot(won);
    if (mw) {   // (the lanes that are here together claim together)
        const int leader = __ffsll((long long) mw) - 1;
        unsigned long long at = 0;
        if (lane == leader) at = atomicAdd(&ctr->next_count, (unsigned long long) __popcll(mw));
        at = __shfl(at, leader, 64);
        if (won) {
            next_q[at + __popcll(mw & ((1ULL << lane) - 1))] = s;
            deg += (unsigned long long) (begin[s + 1] - begin[s]);
        }
    }
}
__global__ void __launch_bounds__(BFS_THREADS)
bfs_topdown_sparse_kernel(const int32_t* __restrict__ begin, const int32_t* __restrict__ node_idx, const int32_t* __restrict__ cur_q,
                          int64_t n, int32_t level, int32_t* __restrict__ dist, int32_t* __restrict__ next_q, bfs_counters* __restrict__ ctr) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    unsigned long long inspected = 0, deg = 0;
    int32_t b = 0, e = 0;
    if (i < n) {
        const int32_t v = cur_q[i];
        b = begin[v];
        e = begin[v + 1];
    }
    const int32_t own_e = e - b > BFS_SPARSE_OWN ? b + BFS_SPARSE_OWN : e;
    // (the loop runs while ANY lane of the wave has an edge left, so that the winners' ballots see whole waves)
    for (int32_t k = 0; __ballot(b + k < own_e) != 0ull; k++) {
        const bool on = b + k < own_e;
        const int32_t s = on ? node_idx[b + k] : 0;
        inspected += on;
        bfs_sparse_visit(on, s, level, dist, begin, next_q, ctr, lane, deg);
    }
    unsigned long long pending = __ballot(own_e < e);
    while (pending) {
        const int src = __ffsll((long long) pending) - 1;
        pending &= pending - 1;
        const int32_t rb = __shfl(own_e, src, 64), re = __shfl(e, src, 64);
        for (int32_t x0 = rb; x0 < re; x0 += 64) {
            const bool on = x0 + lane < re;
            const int32_t s = on ? node_idx[x0 + lane] : 0;
            inspected += on;
            bfs_sparse_visit(on, s, level, dist, begin, next_q, ctr, lane, deg);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        inspected += __shfl_down(inspected, o, 64);
        deg += __shfl_down(deg, o, 64);
    }
    if (lane == 0) {
        bfs_count(ctr, inspected, 0);
        if (deg) atomicAdd(&ctr->next_edges, deg);
    }
}

// The level out of the ROOT (every traversal's first): one row, nobody else visited.  Its distinct entries (the rows are
// sorted: a repeated edge sits next to its copy) other than the root itself all get level + 1 -- no look at dist[], no
// atomicMin: the general kernel spent 75-84 us on the 0.98 M out-edges of RMAT-26's vertex 0, four dependent random accesses
// per edge.  With bm32 the frontier bitmap of the next level is written here too (a large row is followed by a bottom-up
// level, which otherwise starts with a pass over the new queue: 35 us).
__global__ void __launch_bounds__(BFS_THREADS)
bfs_first_level_kernel(const int32_t* __restrict__ begin, const int32_t* __restrict__ node_idx, int32_t root, int32_t level,
                       int32_t* __restrict__ dist, int32_t* __restrict__ next_q, bfs_counters* __restrict__ ctr,
                       unsigned int* __restrict__ bm32 /* or NULL */) {
    __shared__ int32_t s_win[BFS_ITEMS];
    __shared__ unsigned int s_nwin;
    __shared__ unsigned long long s_deg, s_base;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) {
        s_nwin = 0;
        s_deg = 0;
    }
    __syncthreads();
    const int32_t b = begin[root], e = begin[root + 1];
    constexpr int K = BFS_ITEMS / BFS_THREADS;
    const int32_t x0 = b + (int32_t) blockIdx.x * BFS_ITEMS;   // (the grid covers e - b)
    int32_t sv[K], pv[K], dg[K];
    bool won[K];
    unsigned long long inspected = 0, deg = 0;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int32_t x = x0 + tid + k * BFS_THREADS;
        const int32_t xc = x < e ? x : e - 1;
        sv[k] = node_idx[xc];
        pv[k] = xc > b ? node_idx[xc - 1] : -1;
        won[k] = x < e && sv[k] != root && sv[k] != pv[k];
        inspected += x < e;
    }
#pragma unroll
    for (int k = 0; k < K; k++) {   // (unconditional, the others read row 0: no branch, so the loads overlap)
        const int32_t r = won[k] ? sv[k] : 0;
        dg[k] = begin[r + 1] - begin[r];
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
        if (won[k]) {
            dist[sv[k]] = level + 1;
            if (bm32) atomicOr(&bm32[sv[k] >> 5], 1u << (sv[k] & 31));
        }
        const unsigned long long mw = __ballot(won[k]);
        if (mw) {
            const int leader = __ffsll((long long) mw) - 1;
            unsigned int at = 0;
            if (lane == leader) at = atomicAdd(&s_nwin, (unsigned int) __popcll(mw));
            at = __shfl(at, leader, 64);
            if (won[k]) {
                s_win[at + __popcll(mw & ((1ULL << lane) - 1))] = sv[k];
                deg += (unsigned long long) dg[k];
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        inspected += __shfl_down(inspected, o, 64);
        deg += __shfl_down(deg, o, 64);
    }
    if (lane == 0) {
        bfs_count(ctr, inspected, 0);
        if (deg) atomicAdd(&s_deg, deg);
    }
    __syncthreads();
    const unsigned int nwin = s_nwin;
    if (nwin == 0) return;   // (workgroup-uniform)
    if (tid == 0) {
        s_base = atomicAdd(&ctr->next_count, (unsigned long long) nwin);
        if (s_deg) atomicAdd(&ctr->next_edges, s_deg);
    }
    __syncthreads();
    for (unsigned int i = tid; i < nwin; i += BFS_THREADS) next_q[s_base + i] = s_win[i];
}

// frontier bitmap of a level straight from dist[] (coalesced reads, one __ballot per 64 vertices, no atomics)
__global__ void bfs_level_bitmap_kernel(const int32_t* __restrict__ dist, int64_t V, int32_t level,
                                        unsigned long long* __restrict__ bm64) {
    int64_t v = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    const int64_t vend = (V + 63) / 64 * 64;
    for (; v < vend; v += stride) {
        const unsigned long long m = __ballot(v < V && dist[v] == level);
        if ((threadIdx.x & 63) == 0) bm64[v >> 6] = m;
    }
}

// frontier bitmap from the frontier QUEUE (what a top-down level leaves): one atomicOr per vertex into a cleared,
// L2-resident bitmap instead of a pass over dist[] (256 MB at RMAT-26: 105 us)
__global__ void bfs_queue_bitmap_kernel(const int32_t* __restrict__ q, int64_t n, unsigned int* __restrict__ bm32) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const int32_t v = q[i];
        atomicOr(&bm32[v >> 5], 1u << (v & 31));
    }
}

// queue of a level straight from dist[] (ballot-aggregated append)
__global__ void bfs_level_queue_kernel(const int32_t* __restrict__ dist, int64_t V, int32_t level,
                                       int32_t* __restrict__ q, unsigned long long* __restrict__ qcount) {
    int64_t v = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    const int64_t vend = (V + 63) / 64 * 64;
    for (; v < vend; v += stride) {
        const bool in = v < V && dist[v] == level;
        const unsigned long long m = __ballot(in);
        if (m) {
            const int lane = threadIdx.x & 63;
            const int leader = __ffsll((long long) m) - 1;
            unsigned long long base = 0;
            if (lane == leader) base = atomicAdd(qcount, (unsigned long long) __popcll(m));
            base = __shfl(base, leader, 64);
            if (in) q[base + __popcll(m & ((1ULL << lane) - 1))] = (int32_t) v;
        }
    }
}

// queue of the vertices whose bit is set (the frontier a bottom-up level left behind): 8 bytes per 64 vertices to
// read instead of their dist[] entries.  A block takes a contiguous run of words, counts its bits, claims its piece of
// the queue with ONE atomic (atomics on one address retire at ~90 per microsecond: a claim per word cost 170 us at
// RMAT-26, one per wave as much) and walks its words again to write.
__global__ void __launch_bounds__(BFS_THREADS)
bfs_bitmap_queue_kernel(const unsigned long long* __restrict__ bm64, int64_t words,
                        int32_t* __restrict__ q, unsigned long long* __restrict__ qcount,
                        const int32_t* __restrict__ begin, bfs_counters* __restrict__ ctr /* next_edges += the queue's out-edges */) {
    __shared__ unsigned int s_wave[BFS_THREADS / 64];
    __shared__ unsigned long long s_deg[BFS_THREADS / 64];
    __shared__ unsigned long long s_base;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t per = (words + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t) blockIdx.x * per, hi = lo + per < words ? lo + per : words;
    unsigned int c = 0;
    for (int64_t w = lo + threadIdx.x; w < hi; w += BFS_THREADS) c += (unsigned int) __popcll(bm64[w]);
    // thread t owns the words lo + t, lo + t + T, ...: its queue entries follow those of the threads before it
    unsigned int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    unsigned int before = 0, total = 0;
    for (int i = 0; i < BFS_THREADS / 64; i++) {
        if (i < wv) before += s_wave[i];
        total += s_wave[i];
    }
    if (threadIdx.x == 0) s_base = total ? atomicAdd(qcount, (unsigned long long) total) : 0ull;
    __syncthreads();
    unsigned long long at = s_base + before + (incl - c), deg = 0;
    for (int64_t w = lo + threadIdx.x; w < hi; w += BFS_THREADS) {
        unsigned long long m = bm64[w];
        while (m) {
            const int b = __ffsll((long long) m) - 1;
            m &= m - 1;
            const int64_t v = w * 64 + b;
            q[at++] = (int32_t) v;
            deg += (unsigned long long) (begin[v + 1] - begin[v]);
        }
    }
    // the queue's out-edges: what the top-down level that follows cuts by merge-path, known to the host with the
    // level's totals instead of after a scan + copy + stream synchronisation
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) deg += __shfl_down(deg, o, 64);
    if (lane == 0) s_deg[wv] = deg;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int i = 0; i < BFS_THREADS / 64; i++) t += s_deg[i];
        if (t) atomicAdd(&ctr->next_edges, t);
    }
}

// out-edges of the reached vertices (what Graph500 divides by the traversal time)
__global__ void bfs_edges_reached_kernel(const int32_t* __restrict__ dist, const int32_t* __restrict__ begin, int64_t V,
                                         unsigned long long* __restrict__ out) {
    int64_t v = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    unsigned long long acc = 0;
    for (; v < V; v += stride)
        if (dist[v] != INT_MAX) acc += (unsigned long long) (begin[v + 1] - begin[v]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    __shared__ unsigned long long s_acc[BFS_THREADS / 64];
    if ((threadIdx.x & 63) == 0) s_acc[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {   // one atomic per block
        unsigned long long t = 0;
        for (int i = 0; i < (int) (blockDim.x >> 6); i++) t += s_acc[i];
        if (t) atomicAdd(out, t);
    }
}

static int grid_for(int64_t n, int block = BFS_THREADS, int max_blocks = 256 * 8) {
    int64_t b = (n + block - 1) / block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (int) b;
}

// ------------------------------------------------------------------ the traversal as a stepping object (1..N ranks)
// replicated CSR, 1-D vertex ranges.  The levels that matter at scale are the
// bottom-up ones (the few levels that reach most of the graph), and those partition by DESTINATION: rank k
// looks for parents only for the unvisited vertices of its range and reports them as its slice of a
// "found" bitmap -- V/8/N bytes to exchange per level (1 MiB at RMAT-26 on 8 GPUs), after which every rank
// applies the whole bitmap to its dist[] replica and uses it as the next frontier.  Top-down levels (small
// frontiers, little work) are simply run by every rank on the whole frontier: no exchange at all.  Every
// rank sees the same frontier sizes, so all take the same direction decisions without talking.
struct gmx_bfs {
    gmx_graph* g = nullptr;
    int rank = 0, nranks = 1;
    int64_t V = 0, slice_words = 0, words = 0;   // bitmap words (64 vertices each): per rank, total (padded)
    dbuf<int32_t> dist, q0, q1, deg;
    dbuf<int64_t> off;
    dbuf<char> scan_tmp;
    size_t scan_bytes = 0;
    dbuf<unsigned long long> cand;    // unvisited vertices with in-edges, kept from one bottom-up level to the next
    bool cand_valid = false;
    dbuf<unsigned long long> bm[2];   // frontier / found, swapped after every bottom-up level
    dbuf<unsigned long long> hub_bits;   // [BFS_HUBS / 64] the frontier bits of the graph's hubs, rebuilt before every bottom-up level
    int fr = 0;                       // bm[fr] = frontier, bm[1 - fr] = found
    int32_t root = -1;
    int first_bm_level = -1;          // >= 0: bm[fr] already holds the frontier of this level (written by bfs_first_level_kernel)
    bool bm_clean[2] = {false, false};   // all zero (since bfs_init_kernel): the first switch to bottom-up needs no memset
    dbuf<bfs_counters> ctr;
    dbuf<unsigned long long> qcount;
    int32_t level = 0;
    int64_t cur_count = 0, reached = 0, explored = 0;
    unsigned long long found_total = 0;   // bottom-up finds so far (the device counter runs on)
    int64_t cur_edges = -1;           // out-edges of the current queue when the level that built it counted them, else -1
    unsigned long long edges = 0;
    bool frontier_is_bitmap = false, frontier_bm_valid = false, pending_bottom_up = false;
    int32_t* cur_q = nullptr;
    int32_t* next_q = nullptr;
    // per-level read-backs (frontier size, frontier edges) go through pinned host memory: a level costs two
    // host round trips, and with pageable memory each is a staged copy -- at RMAT-24 that was most of the time
    bfs_counters* h_ctr = nullptr;
    bfs_level_totals* h_tot = nullptr;   // pinned; written by bfs_totals_kernel
    unsigned long long tot_tag = 0;
    int64_t* h_mf = nullptr;
    ~gmx_bfs() {
        if (h_ctr) (void) hipHostFree(h_ctr);
        if (h_tot) (void) hipHostFree(h_tot);
        if (h_mf) (void) hipHostFree(h_mf);
    }
};

// The in-neighbour a bottom-up level tries first: of the first BFS_HINT_SCAN entries of the in-row the one with most
// out-edges -- the vertex most likely to be in a frontier early (offline on RMAT-22: 98.6 % of the vertices a bottom-up
// level finds are found through it; a second hint would add 1.2 %).  A hit settles the vertex from 8 bytes (dist, hint)
// without touching its row; at RMAT-26 the first bottom-up level otherwise streams the whole reverse CSR (4.3 GB)
// because every unvisited vertex reads at least the first line of its row.
#define BFS_HINT_SCAN 32
// Encoding of hint[v] (unless the graph has more than 2^30 vertices: then plain ids, -1 = none):
//   -1                         no in-edges at all
//   bit 31 (BFS_H_ONLY)        the hint is the ONLY in-neighbour: when it misses there is no row to walk.  RMAT-26 from
//                              vertex 0: 76 % of the vertices the first bottom-up level leaves without a parent have
//                              in-degree 1 (they are the ones found late, or never), 99.5 % in the later levels -- and a
//                              failed walk has no early exit.
//   bit 30 (BFS_H_HUB)         the low bits are a SLOT of the graph's hub list (the BFS_HUBS vertices with most out-edges),
//                              not a vertex id: its frontier bit is probed in the workgroup's LDS copy of the hubs' bits.
//                              The hint probes are what a bottom-up level spends most of its time on once dist[] and
//                              hint[] stream at the HBM rate (RMAT-26, first level: 27.7 M probes, one 64-byte request for
//                              four useful bytes each, 165 of 450 us); offline on RMAT-22, 72 % of the hints of the
//                              vertices looking in that level are among the V/512 vertices with most out-edges.
#define BFS_H_ONLY 0x80000000u
#define BFS_H_HUB 0x40000000u
#define BFS_H_MASK 0x3FFFFFFFu
#define BFS_HUBS 131072   // 16 KiB of LDS per workgroup
#define BFS_HUB_MIN_V (1LL << 25)
__global__ void bfs_hint_kernel(const int32_t* __restrict__ begin, const int32_t* __restrict__ r_begin,
                                const int32_t* __restrict__ r_node_idx, int64_t V, const int32_t* __restrict__ slot_of /* NULL: slot = id */,
                                int plain, int32_t* __restrict__ hint) {
    int64_t v = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; v < V; v += stride) {
        const int32_t b = r_begin[v], e = r_begin[v + 1];
        const int32_t stop = e - b > BFS_HINT_SCAN ? b + BFS_HINT_SCAN : e;
        int32_t best = -1, best_deg = -1;
        for (int32_t i = b; i < stop; i++) {
            const int32_t w = r_node_idx[i];
            const int32_t d = begin[w + 1] - begin[w];
            if (d > best_deg) {
                best_deg = d;
                best = w;
            }
        }
        if (best < 0 || plain) {
            hint[v] = best;
            continue;
        }
        const int32_t slot = slot_of ? slot_of[best] : -1;   // (no hub list: ids only)
        uint32_t code = slot >= 0 ? (BFS_H_HUB | (uint32_t) slot) : (uint32_t) best;
        if (e - b == 1) code |= BFS_H_ONLY;
        hint[v] = (int32_t) code;
    }
}

// hub list: keys for the sort by out-degree, the slot table, and a level's hub bits
__global__ void bfs_degkey_kernel(const int32_t* __restrict__ begin, int64_t V, uint32_t* __restrict__ key, int32_t* __restrict__ id) {
    int64_t v = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; v < V; v += stride) {
        key[v] = (uint32_t) (begin[v + 1] - begin[v]);
        id[v] = (int32_t) v;
    }
}
__global__ void bfs_slot_of_kernel(const int32_t* __restrict__ hub_id, int n, int32_t* __restrict__ slot_of) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) slot_of[hub_id[s]] = s;
}
// hub_bits[s / 32] bit s % 32 = frontier bit of hub s (n a multiple of 64: the list is padded with vertex 0)
__global__ void bfs_hub_bits_kernel(const uint32_t* __restrict__ frontier_bm, const int32_t* __restrict__ hub_id, int n,
                                    unsigned long long* __restrict__ hub_bits) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    bool bit = false;
    if (s < n) {
        const int32_t v = hub_id[s];
        bit = (frontier_bm[v >> 5] >> (v & 31)) & 1u;
    }
    const unsigned long long m = __ballot(bit);
    if ((threadIdx.x & 63) == 0 && s < n) hub_bits[s >> 6] = m;
}

#define BFS_BU_OWN 32   // in-row entries a vertex checks alone before its wave helps
// bit i of the low 16 bits -> bit 4 i
__device__ __forceinline__ unsigned long long bfs_spread4(unsigned long long x) {
    x = (x | (x << 24)) & 0x000000FF000000FFull;
    x = (x | (x << 12)) & 0x000F000F000F000Full;
    x = (x | (x << 6)) & 0x0303030303030303ull;
    x = (x | (x << 3)) & 0x1111111111111111ull;
    return x;
}
// lane l of a wave holds a bit for each of the vertices 4 l .. 4 l + 3 of 256; bal[j] = __ballot(bit j): the 64-bit word
// of the vertices 64 k .. 64 k + 63
__device__ __forceinline__ unsigned long long bfs_word_of(const unsigned long long bal[4], int k) {
    unsigned long long w = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) w |= bfs_spread4((bal[j] >> (16 * k)) & 0xFFFFull) << j;
    return w;
}

// owned vertices [v_lo, v_hi), both multiples of 64: whole found words, no atomics
//
// Who still looks for a parent: in the first bottom-up level of a run every vertex with dist == INT_MAX; the level
// leaves the ones that found none AND have in-edges as a bitmap (cand), and the following levels read that -- 8 bytes
// per 64 vertices instead of their dist[] entries.
//
// A wave step takes 256 consecutive vertices, FOUR PER LANE: dist[] and hint[] arrive as one 16-byte load per lane, and
// a lane's four hint probes are in flight together.  With one vertex per lane (rounds 2-3) the level ran at the depth of
// the L1 miss queues (47 of ~64 line requests in flight per CU all the time, `profiles/round3_bfs_*_mempipe_pmc.txt`),
// and a 4-byte-per-lane load is a 64-byte request where a 16-byte-per-lane load is a 128-byte one: streaming dist[] and
// hint[] alone (536 MB at RMAT-26) took 238 us.  The vertices whose hint missed and that have more than one in-edge
// then walk their rows -- compacted first (their offsets go through LDS), so that the walk runs with full waves
// instead of once per vertex slot with a quarter of the lanes.
__global__ void __launch_bounds__(BFS_THREADS)
bfs_bottomup_part_kernel(const int32_t* __restrict__ r_begin, const int32_t* __restrict__ r_node_idx,
                         int64_t v_lo, int64_t v_hi, int64_t V, const uint32_t* __restrict__ frontier_bm,
                         const int32_t* dist, unsigned long long* __restrict__ found_bm,
                         int32_t* dist_w /* NULL or == dist */, int32_t next_level, bfs_counters* __restrict__ ctr,
                         const unsigned long long* cand_in /* NULL: take the unvisited from dist[] */,
                         unsigned long long* cand_out, const int32_t* __restrict__ hint, int plain,
                         const uint32_t* __restrict__ hub_bits, int hub_words /* > 0: probe the hubs in an LDS copy */) {
    __shared__ uint32_t s_hub[BFS_HUBS / 32];             // the frontier bits of the hubs, by slot (see BFS_H_HUB)
    __shared__ uint8_t s_walk[BFS_THREADS / 64][256];     // offsets (0..255) of the wave step's vertices that walk their row
    __shared__ uint32_t s_wfound[BFS_THREADS / 64][8];    // ... and which of them found a parent
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t nwaves = (int64_t) gridDim.x * (BFS_THREADS / 64);
    const int64_t word_hi = v_hi >> 6;
    unsigned long long inspected = 0, found_cnt = 0;
    for (int i = threadIdx.x; i < hub_words; i += BFS_THREADS) s_hub[i] = hub_bits[i];
    __syncthreads();
    for (int64_t w0 = (v_lo >> 6) + 4 * ((int64_t) blockIdx.x * (BFS_THREADS / 64) + wv); w0 < word_hi; w0 += 4 * nwaves) {
        const int nw = word_hi - w0 < 4 ? (int) (word_hi - w0) : 4;   // words of this step (wave-uniform)
        const int64_t v = w0 * 64 + 4 * lane;                         // this lane's vertices: v .. v + 3
        const bool lane_in = (lane >> 4) < nw;
        bool active[4];
        int32_t d4[4] = {0, 0, 0, 0};
        if (cand_in) {
            const unsigned long long cw = lane_in ? cand_in[w0 + (lane >> 4)] : 0ull;
            const unsigned nib = (unsigned) (cw >> (4 * (lane & 15))) & 15u;
            if (__ballot(nib != 0u) == 0ull) {   // nobody here looks for a parent
                if (lane < nw) found_bm[w0 + lane] = 0ull;
                continue;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) active[j] = (nib >> j) & 1u;
        } else {
            if (lane_in && v + 3 < V) {
                const int4 q = *reinterpret_cast<const int4*>(dist + v);
                d4[0] = q.x; d4[1] = q.y; d4[2] = q.z; d4[3] = q.w;
            } else if (lane_in) {
#pragma unroll
                for (int j = 0; j < 4; j++) d4[j] = v + j < V ? dist[v + j] : 0;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) active[j] = d4[j] == INT_MAX;
        }
        int32_t h[4] = {-1, -1, -1, -1};   // -1: not looking, or no in-edges at all
        if (active[0] || active[1] || active[2] || active[3]) {
            if (v + 3 < V) {
                const int4 q = *reinterpret_cast<const int4*>(hint + v);
                h[0] = q.x; h[1] = q.y; h[2] = q.z; h[3] = q.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) h[j] = v + j < V ? hint[v + j] : -1;
            }
        }
        bool only[4], found[4], need[4], hub[4];
        uint32_t probe[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (!active[j]) h[j] = -1;
            const uint32_t raw = (uint32_t) h[j];
            const bool coded = !plain && h[j] != -1;
            only[j] = coded && (raw & BFS_H_ONLY);
            hub[j] = coded && (raw & BFS_H_HUB);
            if (coded) h[j] = (int32_t) (raw & BFS_H_MASK);   // a vertex id, or a hub slot
        }
        // (the probes are unconditional, from a harmless address where they do not apply: four loads in flight, no branches
        // between them)
        if (hub_words > 0) {   // (kernel argument: uniform)  hubs in the LDS copy, the others in the frontier bitmap
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t g = frontier_bm[(h[j] >= 0 && !hub[j] ? h[j] : 0) >> 5];
                const uint32_t l = s_hub[(hub[j] ? h[j] : 0) >> 5];
                probe[j] = hub[j] ? l : g;
            }
        } else {               // one load either way: the hubs' bits are a 16 KiB array in memory
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t* base = hub[j] ? hub_bits : frontier_bm;
                probe[j] = base[(h[j] >= 0 ? h[j] : 0) >> 5];
            }
        }
        bool any_need = false;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bool has_in = h[j] >= 0;
            found[j] = has_in && ((probe[j] >> (h[j] & 31)) & 1u);
            inspected += has_in;
            need[j] = has_in && !found[j] && !only[j];
            any_need |= need[j];
        }
        if (__ballot(any_need) != 0ull) {
            // the walkers of this step, compacted: one per lane, 64 at a time
            if (lane < 8) s_wfound[wv][lane] = 0u;
            int nwalk = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned long long m = __ballot(need[j]);
                if (need[j]) s_walk[wv][nwalk + __popcll(m & ((1ull << lane) - 1ull))] = (uint8_t) (4 * lane + j);
                nwalk += __popcll(m);
            }
            for (int base = 0; base < nwalk; base += 64) {
                const bool mine = base + lane < nwalk;
                const int off = mine ? s_walk[wv][base + lane] : 0;
                const int64_t t = w0 * 64 + off;
                bool f = false;
                int32_t rest_b = 0, rest_e = 0;
                if (mine) {
                    // A vertex looks for a parent among the first BFS_BU_OWN entries of its in-row by itself (most rows are
                    // shorter, and most long ones find a parent at once); what is left of the long rows is then searched by
                    // the whole wave, one row at a time, 64 entries per step with an early exit -- a lone lane walking a
                    // 10^4-entry row of a vertex that has no parent in this level was the tail of the whole level.
                    const int32_t b = r_begin[t], e = r_begin[t + 1];
                    const int32_t own_e = e - b > BFS_BU_OWN ? b + BFS_BU_OWN : e;
                    // four entries and their four bitmap probes per step, all in flight together; the first hit still
                    // ends the walk and only the entries up to it count as inspected
                    for (int32_t i = b; i < own_e && !f; i += 4) {
                        const int32_t last = own_e - 1;
                        const int32_t w0e = r_node_idx[i], w1 = r_node_idx[i + 1 < last ? i + 1 : last], w2 = r_node_idx[i + 2 < last ? i + 2 : last],
                                      w3 = r_node_idx[i + 3 < last ? i + 3 : last];
                        const uint32_t p0 = frontier_bm[w0e >> 5], p1 = frontier_bm[w1 >> 5], p2 = frontier_bm[w2 >> 5], p3 = frontier_bm[w3 >> 5];
                        const int n = own_e - i < 4 ? own_e - i : 4;
                        const bool h0 = (p0 >> (w0e & 31)) & 1u, h1 = n > 1 && ((p1 >> (w1 & 31)) & 1u), h2 = n > 2 && ((p2 >> (w2 & 31)) & 1u),
                                   h3 = n > 3 && ((p3 >> (w3 & 31)) & 1u);
                        f = h0 || h1 || h2 || h3;
                        inspected += h0 ? 1 : h1 ? 2 : h2 ? 3 : h3 ? 4 : n;
                    }
                    if (!f && own_e < e) {
                        rest_b = own_e;
                        rest_e = e;
                    }
                }
                unsigned long long pending = __ballot(rest_e > rest_b);
                while (pending) {
                    const int src = __ffsll((long long) pending) - 1;
                    pending &= pending - 1;
                    const int32_t rb = __shfl(rest_b, src, 64), re = __shfl(rest_e, src, 64);
                    bool hit = false;
                    for (int32_t i0 = rb; i0 < re && !hit; i0 += 64) {
                        const int32_t i = i0 + lane;
                        bool here = false;
                        if (i < re) {
                            const int32_t w = r_node_idx[i];
                            inspected++;
                            here = (frontier_bm[w >> 5] & (1u << (w & 31))) != 0;
                        }
                        hit = __ballot(here) != 0ull;
                    }
                    if (lane == src) f = hit;
                }
                if (f) atomicOr(&s_wfound[wv][off >> 5], 1u << (off & 31));
            }
            const uint32_t fw = s_wfound[wv][lane >> 3];   // the bits of the vertices 4 lane .. 4 lane + 3
#pragma unroll
            for (int j = 0; j < 4; j++) found[j] = found[j] || ((fw >> ((4 * lane + j) & 31)) & 1u);
        }
        unsigned long long fb[4], sb[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            fb[j] = __ballot(found[j]);
            sb[j] = __ballot(h[j] >= 0 && !found[j]);   // still looking, and worth asking again (has in-edges)
        }
        if (lane < nw) {
            const unsigned long long m = bfs_word_of(fb, lane);
            found_bm[w0 + lane] = m;
            cand_out[w0 + lane] = bfs_word_of(sb, lane);
            if (dist_w) found_cnt += (unsigned long long) __popcll(m);
        }
        // single rank: the whole bitmap is this rank's, so dist[] can be settled right here (dist_w aliases dist;
        // a vertex only ever reads its own entry) and the separate apply pass is not needed
        if (dist_w && (found[0] || found[1] || found[2] || found[3])) {
            if (!cand_in && v + 3 < V) {   // the lane holds all four entries: one 16-byte store
                int4 q;
                q.x = found[0] ? next_level : d4[0]; q.y = found[1] ? next_level : d4[1];
                q.z = found[2] ? next_level : d4[2]; q.w = found[3] ? next_level : d4[3];
                *reinterpret_cast<int4*>(dist_w + v) = q;
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (found[j]) dist_w[v + j] = next_level;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        inspected += __shfl_down(inspected, off, 64);
        found_cnt += __shfl_down(found_cnt, off, 64);
    }
    if (lane == 0) bfs_count(ctr, inspected, found_cnt);
}

// every rank, whole bitmap: dist[v] = next_level where the bit is set; counts the new frontier
__global__ void bfs_apply_found_kernel(const unsigned long long* __restrict__ found_bm, int64_t V, int32_t next_level,
                                       int32_t* __restrict__ dist, bfs_counters* __restrict__ ctr) {
    int64_t v = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    const int64_t vend = (V + 63) / 64 * 64;
    unsigned long long cnt = 0;
    for (; v < vend; v += stride) {
        const unsigned long long m = found_bm[v >> 6];
        if (v < V && ((m >> (v & 63)) & 1ULL)) dist[v] = next_level;
        if ((threadIdx.x & 63) == 0) cnt += (unsigned long long) __popcll(m);
    }
    if ((threadIdx.x & 63) == 0) bfs_count(ctr, 0, cnt);
}

static int64_t bfs_hub_min_v() {
    const char* e = getenv("GMX_BFS_HUB_MIN_V");
    const long long v = e ? atoll(e) : 0;
    return v > 0 ? (int64_t) v : BFS_HUB_MIN_V;
}

// The per-graph preprocessing of the bottom-up levels: the hub list (BFS_HUBS vertices with most out-edges) and hint[].
static int bfs_build_hints(gmx_graph* g) {
    const int64_t V = g->V;
    GMX_CHECK(g->bfs_hint.alloc((size_t) (V ? V : 1)));
    // development / test options: GMX_BFS_PLAIN_HINTS=1 forces the plain encoding, GMX_BFS_HUB_MIN_V=<n> moves the size from
    // which graphs get a hub list (and levels the LDS copy) -- so that every form can be run on small graphs
    g->bfs_hint_plain = V > (int64_t) BFS_H_MASK || getenv("GMX_BFS_PLAIN_HINTS") != nullptr;
    const int64_t hub_min_v = bfs_hub_min_v();
    // hubs only where a bottom-up level is long enough to gain from them: below 2^25 vertices the extra launch per level
    // (the hubs' bits) cost more than the probes it saved (RMAT-20/22/24: 5-10 % slower with hubs, RMAT-26: 10 % faster)
    g->bfs_hubs = !g->bfs_hint_plain && V >= hub_min_v && V > BFS_HUBS ? BFS_HUBS : 0;
    if (V == 0) return GMX_OK;
    gmx_ws_scope scope;   // (temporaries from the workspace: see gmx_internal.h)
    wbuf<int32_t> slot_of;
    if (g->bfs_hubs > 0) {
        wbuf<uint32_t> key, key2;
        wbuf<int32_t> id, id2;
        wbuf<char> tmp;
        GMX_CHECK(key.alloc((size_t) V));
        GMX_CHECK(key2.alloc((size_t) V));
        GMX_CHECK(id.alloc((size_t) V));
        GMX_CHECK(id2.alloc((size_t) V));
        GMX_CHECK(slot_of.alloc((size_t) V));
        GMX_CHECK(g->bfs_hub_id.alloc((size_t) BFS_HUBS));
        hipLaunchKernelGGL(bfs_degkey_kernel, dim3(grid_for(V)), dim3(BFS_THREADS), 0, 0, (const int32_t*) g->begin.p, V, key.p, id.p);
        size_t tb = 0;
        GMX_HIP(rocprim::radix_sort_pairs_desc(nullptr, tb, key.p, key2.p, id.p, id2.p, (size_t) V, 0u, 32u, 0));
        GMX_CHECK(tmp.alloc(tb));
        GMX_HIP(rocprim::radix_sort_pairs_desc((void*) tmp.p, tb, key.p, key2.p, id.p, id2.p, (size_t) V, 0u, 32u, 0));
        GMX_HIP(hipMemcpyAsync(g->bfs_hub_id.p, id2.p, sizeof(int32_t) * BFS_HUBS, hipMemcpyDeviceToDevice, 0));
        GMX_HIP(hipMemsetAsync(slot_of.p, 0xFF, sizeof(int32_t) * (size_t) V, 0));
        hipLaunchKernelGGL(bfs_slot_of_kernel, dim3(BFS_HUBS / BFS_THREADS), dim3(BFS_THREADS), 0, 0, (const int32_t*) g->bfs_hub_id.p, BFS_HUBS, slot_of.p);
    }
    hipLaunchKernelGGL(bfs_hint_kernel, dim3(grid_for(V)), dim3(BFS_THREADS), 0, 0, (const int32_t*) g->begin.p,
                       (const int32_t*) g->r_begin.p, (const int32_t*) g->r_node_idx.p, V, (const int32_t*) slot_of.p,
                       g->bfs_hint_plain ? 1 : 0, g->bfs_hint.p);
    GMX_HIP(hipGetLastError());
    return GMX_OK;   // (the scope's destructor waits for the kernels before the temporaries are handed out again)
}

extern "C" int gmx_bfs_create(gmx_graph_t* g, int rank, int nranks, gmx_bfs_t** out) {
    GMX_REQUIRE(out, "out is NULL");
    *out = nullptr;
    GMX_REQUIRE(g, "graph is NULL");
    GMX_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank %d / nranks %d", rank, nranks);
    GMX_REQUIRE(nranks == 1 || g->has_reverse, "the partitioned traversal needs the reverse CSR");
    if (g->has_reverse && !g->bfs_hint.p) {   // graph preprocessing for the bottom-up levels
        GMX_CHECK(bfs_build_hints(g));
    }
    gmx_bfs* b = new gmx_bfs();
    b->g = g;
    b->rank = rank;
    b->nranks = nranks;
    b->V = g->V;
    const int64_t w = (g->V + 63) / 64;
    b->slice_words = (w + nranks - 1) / nranks;
    if (b->slice_words < 1) b->slice_words = 1;
    b->words = b->slice_words * nranks;
    const size_t V1 = (size_t) (g->V ? g->V : 1);
    int st = GMX_OK;
    if ((st = b->dist.alloc(V1)) || (st = b->q0.alloc(V1)) || (st = b->q1.alloc(V1)) || (st = b->deg.alloc(V1)) ||
        (st = b->off.alloc(V1 + 2)) || (st = b->ctr.alloc(1)) || (st = b->qcount.alloc(1)) ||
        (st = b->bm[0].alloc((size_t) b->words)) || (st = b->bm[1].alloc((size_t) b->words)) || (st = b->cand.alloc((size_t) b->words)) ||
        (st = b->hub_bits.alloc(BFS_HUBS / 64))) {
        delete b;
        return st;
    }
    if (hipHostMalloc((void**) &b->h_tot, sizeof(bfs_level_totals), hipHostMallocDefault) != hipSuccess) {
        gmx_set_error("hipHostMalloc failed");
        delete b;
        return GMX_ERR_NOMEM;
    }
    memset(b->h_tot, 0, sizeof(bfs_level_totals));
    if (hipHostMalloc((void**) &b->h_ctr, sizeof(bfs_counters), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**) &b->h_mf, sizeof(int64_t), hipHostMallocDefault) != hipSuccess) {
        delete b;
        gmx_set_error("bfs: pinned host allocation failed");
        return GMX_ERR_HIP;
    }
    if (rocprim::inclusive_scan(nullptr, b->scan_bytes, b->deg.p, b->off.p + 1, V1, rocprim::plus<int64_t>(), 0) != hipSuccess ||
        (st = b->scan_tmp.alloc(b->scan_bytes))) {
        delete b;
        gmx_set_error("bfs: scan setup failed");
        return st ? st : GMX_ERR_HIP;
    }
    *out = b;
    return GMX_OK;
}

extern "C" int gmx_bfs_free(gmx_bfs_t* b) {
    delete b;
    return GMX_OK;
}

// spin on the tag of the pinned totals; after ~2 s of nothing fall back to a synchronisation (which also reports a
// failed kernel)
static int bfs_wait_totals(gmx_bfs* b, unsigned long long tag) {
    const auto t0 = std::chrono::steady_clock::now();
    unsigned long long spins = 0;
    while (b->h_tot->tag != tag) {
        if ((++spins & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
            GMX_HIP(hipStreamSynchronize(0));
            GMX_REQUIRE(b->h_tot->tag == tag, "bfs: the level's totals did not arrive");
            break;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return GMX_OK;
}

extern "C" int gmx_bfs_start(gmx_bfs_t* b, gmx_node_t root) {
    GMX_REQUIRE(b, "bfs is NULL");
    const int64_t V = b->V;
    const bool root_ok = root >= 0 && root < V;
    // one launch; the root's out-degree comes back through pinned memory (see bfs_level_totals)
    const unsigned long long tag = ++b->tot_tag;
    hipLaunchKernelGGL(bfs_init_kernel, dim3(grid_for(V > (int64_t) (sizeof(bfs_counters) / 8) ? V : (int64_t) (sizeof(bfs_counters) / 8))), dim3(BFS_THREADS), 0, 0,
                       b->dist.p, V, root_ok ? (int32_t) root : -1, b->bm[0].p, b->bm[1].p, b->words, b->ctr.p, b->q0.p,
                       (const int32_t*) b->g->begin.p, b->h_tot, tag);
    GMX_HIP(hipGetLastError());
    b->level = 0;
    b->cur_count = b->reached = root_ok ? 1 : 0;
    b->explored = 0;
    b->edges = 0;
    b->frontier_is_bitmap = b->frontier_bm_valid = b->pending_bottom_up = false;
    b->cand_valid = false;
    b->fr = 0;
    b->bm_clean[0] = b->bm_clean[1] = true;   // bfs_init_kernel clears both
    b->root = root_ok ? (int32_t) root : -1;
    b->first_bm_level = -1;
    b->cur_q = b->q0.p;
    b->next_q = b->q1.p;
    b->cur_edges = -1;
    b->found_total = 0;
    GMX_CHECK(bfs_wait_totals(b, tag));
    if (root_ok) b->cur_edges = (int64_t) b->h_tot->next_edges;
    return GMX_OK;
}

// frontier out-degrees -> off[], returns their sum (the same helper steps as in gmx_hop_dist)
static int bfs_frontier_edges(gmx_bfs* b, int64_t* m_f) {
    hipLaunchKernelGGL(bfs_degree_kernel, dim3(grid_for(b->cur_count)), dim3(BFS_THREADS), 0, 0, b->g->begin.p, b->cur_q, b->cur_count, b->deg.p);
    size_t tb = b->scan_bytes;
    GMX_HIP(rocprim::inclusive_scan(b->scan_tmp.p, tb, b->deg.p, b->off.p + 1, (size_t) b->cur_count, rocprim::plus<int64_t>(), 0));
    GMX_HIP(hipMemsetAsync(b->off.p, 0, sizeof(int64_t), 0));
    GMX_HIP(hipMemcpyAsync(b->h_mf, b->off.p + b->cur_count, sizeof(int64_t), hipMemcpyDeviceToHost, 0));
    GMX_HIP(hipStreamSynchronize(0));
    *m_f = *b->h_mf;
    return GMX_OK;
}

extern "C" int gmx_bfs_step_begin(gmx_bfs_t* b, int* needs_exchange) {
    GMX_REQUIRE(b && needs_exchange, "NULL argument");
    *needs_exchange = 0;
    b->pending_bottom_up = false;
    if (b->cur_count <= 0) return GMX_OK;
    gmx_graph* g = b->g;
    const int64_t V = b->V;
    int64_t m_f = 0;
    bool bottom_up, have_off = false, cleared = false;
    if (b->frontier_is_bitmap) bottom_up = b->cur_count > V / 24;
    else {
        if (b->cur_edges >= 0) m_f = b->cur_edges;   // counted by the level that built the queue: no pass, no read-back
        else {
            GMX_CHECK(bfs_frontier_edges(b, &m_f));
            have_off = true;
        }
        bottom_up = g->has_reverse && (b->cur_count > V / 20 || m_f > (g->E - b->explored) / 14);
        b->explored += m_f;
    }
    if (bottom_up) {
        // (next_count / next_edges are top-down counters: a bottom-up level neither reads nor writes them, and the switch back
        // to top-down clears them where it needs them -- a memset here was a launch slot of ~7 us per level)
        if (!b->frontier_bm_valid) {   // first bottom-up level after queue levels: frontier = {v : dist[v] == level}
            if (!b->frontier_is_bitmap && b->first_bm_level == b->level) {
                // ... which the level out of the root has written already
            } else if (!b->frontier_is_bitmap) {   // ... which is the queue the last top-down level wrote
                if (!b->bm_clean[b->fr]) GMX_HIP(hipMemsetAsync(b->bm[b->fr].p, 0, sizeof(unsigned long long) * (size_t) b->words, 0));
                hipLaunchKernelGGL(bfs_queue_bitmap_kernel, dim3(grid_for(b->cur_count, BFS_THREADS, 256 * 16)), dim3(BFS_THREADS), 0, 0,
                                   (const int32_t*) b->cur_q, b->cur_count, (unsigned int*) b->bm[b->fr].p);
            } else
                hipLaunchKernelGGL(bfs_level_bitmap_kernel, dim3(grid_for(V, BFS_THREADS, 256 * 16)), dim3(BFS_THREADS), 0, 0,
                                   (const int32_t*) b->dist.p, V, b->level, b->bm[b->fr].p);
        }
        b->bm_clean[0] = b->bm_clean[1] = false;   // the frontier was just written, the level writes the other one
        const int64_t v_lo = (int64_t) b->rank * b->slice_words * 64;
        const int64_t v_hi = v_lo + b->slice_words * 64;
        // the hubs' frontier bits, by slot
        const uint32_t* hub_bits = (const uint32_t*) b->bm[b->fr].p;   // (no hub list: never read through a slot)
        int hub_words = 0;
        if (g->bfs_hub_id.p) {
            hipLaunchKernelGGL(bfs_hub_bits_kernel, dim3(BFS_HUBS / BFS_THREADS), dim3(BFS_THREADS), 0, 0, (const uint32_t*) b->bm[b->fr].p,
                               (const int32_t*) g->bfs_hub_id.p, BFS_HUBS, b->hub_bits.p);
            hub_bits = (const uint32_t*) b->hub_bits.p;
            hub_words = BFS_HUBS / 32;
        }
        // The LDS copy (16 KiB per workgroup, filled once by each of one resident set of workgroups) pays where the level is
        // long: the first bottom-up level of a run over at least 2^25 vertices (RMAT-26: 450 -> 323 us; the later levels, with
        // a fifth of the vertices still looking, were slower with it).  Otherwise the same 16 KiB are probed in memory.
        const bool use_lds = hub_words > 0 && !b->cand_valid && v_hi - v_lo >= bfs_hub_min_v();
        hipLaunchKernelGGL(bfs_bottomup_part_kernel, dim3(grid_for((v_hi - v_lo + 3) / 4, BFS_THREADS, use_lds ? 256 * 7 : 256 * 32)), dim3(BFS_THREADS), 0, 0,
                           g->r_begin.p, g->r_node_idx.p, v_lo, v_hi, V, (const uint32_t*) b->bm[b->fr].p,
                           (const int32_t*) b->dist.p, b->bm[1 - b->fr].p, b->nranks == 1 ? b->dist.p : nullptr, b->level + 1, b->ctr.p,
                           b->cand_valid ? (const unsigned long long*) b->cand.p : nullptr, b->cand.p, (const int32_t*) g->bfs_hint.p,
                           g->bfs_hint_plain ? 1 : 0, hub_bits, use_lds ? hub_words : 0);
        b->cand_valid = true;
        b->pending_bottom_up = true;
        *needs_exchange = b->nranks > 1 ? 1 : 0;
    } else {
        b->cand_valid = false;         // a top-down level visits vertices the candidate bitmap would still hold
        if (b->frontier_is_bitmap) {   // back from bottom-up: rebuild the queue and its edge offsets
            GMX_HIP(hipMemsetAsync(b->qcount.p, 0, sizeof(unsigned long long), 0));
            if (b->frontier_bm_valid) {   // the frontier is the bitmap the last bottom-up level found
                GMX_HIP(hipMemsetAsync(&b->ctr.p->next_count, 0, 2 * sizeof(unsigned long long), 0));   // next_count, next_edges
                hipLaunchKernelGGL(bfs_bitmap_queue_kernel, dim3(grid_for((V + 63) / 64, BFS_THREADS, 256 * 4)), dim3(BFS_THREADS), 0, 0,
                                   (const unsigned long long*) b->bm[b->fr].p, (V + 63) / 64, b->cur_q, b->qcount.p,
                                   (const int32_t*) g->begin.p, b->ctr.p);
                const unsigned long long tag = ++b->tot_tag;
                hipLaunchKernelGGL(bfs_totals_kernel, dim3(1), dim3(64), 0, 0, (const bfs_counters*) b->ctr.p, b->h_tot, tag);
                GMX_HIP(hipGetLastError());
                GMX_CHECK(bfs_wait_totals(b, tag));
                m_f = (int64_t) b->h_tot->next_edges;   // the offsets follow below, without a read-back
            } else {
                hipLaunchKernelGGL(bfs_level_queue_kernel, dim3(grid_for(V, BFS_THREADS, 256 * 16)), dim3(BFS_THREADS), 0, 0,
                                   (const int32_t*) b->dist.p, V, b->level, b->cur_q, b->qcount.p);
                GMX_CHECK(bfs_frontier_edges(b, &m_f));
                have_off = true;
            }
            b->frontier_is_bitmap = false;
            b->explored += m_f;
        }
        b->frontier_bm_valid = false;
        if (b->level == 0 && b->cur_count == 1 && b->root >= 0 && !have_off) {   // the level out of the root
            // (the counters are clear: bfs_init_kernel)  A row this large is followed by a bottom-up level: write its bitmap too
            // (liberal: a wrong guess costs a memset of the bitmap later, a missed one a pass over the queue now)
            const bool with_bm = g->has_reverse && b->bm_clean[b->fr] && m_f >= 64 && m_f >= V / 1024;
            const int64_t nb1 = (m_f + BFS_ITEMS - 1) / BFS_ITEMS;
            if (nb1 > 0)
                hipLaunchKernelGGL(bfs_first_level_kernel, dim3((unsigned) nb1), dim3(BFS_THREADS), 0, 0, (const int32_t*) g->begin.p,
                                   (const int32_t*) g->node_idx.p, b->root, b->level, b->dist.p, b->next_q, b->ctr.p,
                                   with_bm ? (unsigned int*) b->bm[b->fr].p : nullptr);
            if (with_bm && nb1 > 0) {
                b->bm_clean[b->fr] = false;
                b->first_bm_level = 1;
            }
            int32_t* t = b->cur_q;
            b->cur_q = b->next_q;
            b->next_q = t;
            b->cur_edges = -2;
            GMX_HIP(hipGetLastError());
            return GMX_OK;
        }
        if (!have_off && b->cur_count <= (1 << 20) && m_f <= 2 * b->cur_count + 1024) {   // a sparse frontier: one vertex per lane
            GMX_HIP(hipMemsetAsync(&b->ctr.p->next_count, 0, 2 * sizeof(unsigned long long), 0));
            hipLaunchKernelGGL(bfs_topdown_sparse_kernel, dim3(grid_for(b->cur_count, BFS_THREADS, 1 << 20)), dim3(BFS_THREADS), 0, 0,
                               (const int32_t*) g->begin.p, (const int32_t*) g->node_idx.p, (const int32_t*) b->cur_q, b->cur_count, b->level,
                               b->dist.p, b->next_q, b->ctr.p);
            int32_t* t = b->cur_q;
            b->cur_q = b->next_q;
            b->next_q = t;
            b->cur_edges = -2;
            GMX_HIP(hipGetLastError());
            return GMX_OK;
        }
        if (!have_off) {   // merge-path offsets of the queue
            if (b->cur_count <= BFS_SMALL_SCAN) {
                hipLaunchKernelGGL(bfs_degree_scan_small_kernel, dim3(1), dim3(1024), 0, 0, g->begin.p, (const int32_t*) b->cur_q,
                                   (int) b->cur_count, b->off.p, b->ctr.p);
                cleared = true;
            } else {
                hipLaunchKernelGGL(bfs_degree_kernel, dim3(grid_for(b->cur_count)), dim3(BFS_THREADS), 0, 0, g->begin.p, b->cur_q, b->cur_count, b->deg.p);
                size_t tb = b->scan_bytes;
                GMX_HIP(rocprim::inclusive_scan(b->scan_tmp.p, tb, b->deg.p, b->off.p + 1, (size_t) b->cur_count, rocprim::plus<int64_t>(), 0));
                GMX_HIP(hipMemsetAsync(b->off.p, 0, sizeof(int64_t), 0));
            }
        }
        if (!cleared) GMX_HIP(hipMemsetAsync(&b->ctr.p->next_count, 0, 2 * sizeof(unsigned long long), 0));
        const int64_t nb = (b->cur_count + m_f + BFS_ITEMS - 1) / BFS_ITEMS;
        if (nb > 0)
            hipLaunchKernelGGL(bfs_topdown_kernel, dim3((unsigned) nb), dim3(BFS_THREADS), 0, 0,
                               g->begin.p, g->node_idx.p, b->cur_q, b->cur_count, b->off.p, m_f, b->level, b->dist.p, b->next_q, b->ctr.p);
        int32_t* t = b->cur_q;
        b->cur_q = b->next_q;
        b->next_q = t;
    }
    b->cur_edges = bottom_up ? -1 : -2;   // -2: a top-down level is running, step_end learns the next queue's edges
    GMX_HIP(hipGetLastError());
    return GMX_OK;
}

extern "C" int gmx_bfs_found_bitmap(gmx_bfs_t* b, void** words, int64_t* total_words, int64_t* slice_offset, int64_t* slice_words) {
    GMX_REQUIRE(b && words && total_words && slice_offset && slice_words, "NULL argument");
    *words = b->bm[1 - b->fr].p;
    *total_words = b->words;
    *slice_offset = (int64_t) b->rank * b->slice_words;
    *slice_words = b->slice_words;
    return GMX_OK;
}

extern "C" int gmx_bfs_step_end(gmx_bfs_t* b, int64_t* next_count) {
    GMX_REQUIRE(b && next_count, "NULL argument");
    *next_count = 0;
    if (b->cur_count <= 0) return GMX_OK;
    if (b->pending_bottom_up) {
        if (b->nranks > 1) {   // (a single rank has settled dist[] and the count inside the bottom-up kernel)
            hipLaunchKernelGGL(bfs_apply_found_kernel, dim3(grid_for(b->V, BFS_THREADS, 256 * 16)), dim3(BFS_THREADS), 0, 0,
                               (const unsigned long long*) b->bm[1 - b->fr].p, b->V, b->level + 1, b->dist.p, b->ctr.p);
            GMX_HIP(hipGetLastError());
        }
        b->fr = 1 - b->fr;   // what was found is the next frontier
        b->frontier_is_bitmap = b->frontier_bm_valid = true;
        b->pending_bottom_up = false;
    }
    const unsigned long long tag = ++b->tot_tag;
    hipLaunchKernelGGL(bfs_totals_kernel, dim3(1), dim3(64), 0, 0, (const bfs_counters*) b->ctr.p, b->h_tot, tag);
    GMX_HIP(hipGetLastError());
    GMX_CHECK(bfs_wait_totals(b, tag));
    struct { unsigned long long next_count, next_edges; } h = {b->h_tot->next_count, b->h_tot->next_edges};
    const unsigned long long edges = b->h_tot->edges, found = b->h_tot->found;
    // a top-down level leaves its queue tail in next_count, a bottom-up level its finds in the running total
    if (getenv("GMX_BFS_DEBUG"))
        fprintf(stderr, "gmx bfs: level %d %s: frontier %lld -> inspected %llu, next frontier %lld\n", b->level, b->cur_edges == -2 ? "top-down" : "bottom-up",
                (long long) b->cur_count, edges - b->edges, (long long) (b->cur_edges == -2 ? h.next_count : found - b->found_total));
    b->cur_count = b->cur_edges == -2 ? (int64_t) h.next_count : (int64_t) (found - b->found_total);
    b->found_total = found;
    b->cur_edges = b->cur_edges == -2 ? (int64_t) h.next_edges : -1;
    b->edges = edges;
    b->reached += b->cur_count;
    b->level++;
    *next_count = b->cur_count;
    return GMX_OK;
}

extern "C" int gmx_bfs_download(gmx_bfs_t* b, int32_t* dist_host, gmx_stats_t* stats) {
    GMX_REQUIRE(b && dist_host, "NULL argument");
    if (stats) memset(stats, 0, sizeof(*stats));
    if (b->V > 0) GMX_HIP(hipMemcpy(dist_host, b->dist.p, sizeof(int32_t) * (size_t) b->V, hipMemcpyDeviceToHost));
    if (stats) {
        stats->iterations = b->level;
        stats->edges_examined = (int64_t) b->edges;   // this rank's share in the partitioned levels
        stats->vertices_reached = b->reached;
    }
    return GMX_OK;
}

// The whole-kernel entry (what the generated hop_dist() calls): the stepping object with a single rank.
extern "C" int gmx_hop_dist(gmx_graph_t* g, gmx_node_t root, int32_t* dist_host, gmx_stats_t* stats) {
    GMX_REQUIRE(g && dist_host, "NULL argument");
    if (stats) memset(stats, 0, sizeof(*stats));
    if (g->V == 0) return GMX_OK;
    if (!g->bfs_cache) GMX_CHECK(gmx_bfs_create(g, 0, 1, &g->bfs_cache));   // graph preprocessing, like the reverse CSR
    gmx_bfs_t* b = g->bfs_cache;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    for (hipEvent_t& e : ev) (void) hipEventCreate(&e);
    int st = GMX_OK;
    (void) hipEventRecord(ev[0], 0);
    if ((st = gmx_bfs_start(b, root)) == GMX_OK) {
        int64_t next = 0;
        int need = 0;
        do {
            if ((st = gmx_bfs_step_begin(b, &need)) != GMX_OK) break;
            if ((st = gmx_bfs_step_end(b, &next)) != GMX_OK) break;
        } while (next > 0);
    }
    (void) hipEventRecord(ev[1], 0);
    (void) hipEventSynchronize(ev[1]);
    unsigned long long edges_reached = 0;
    if (st == GMX_OK && stats) {   // statistics only, outside the timed traversal
        if (hipMemset(b->qcount.p, 0, sizeof(unsigned long long)) == hipSuccess) {
            hipLaunchKernelGGL(bfs_edges_reached_kernel, dim3(grid_for(g->V)), dim3(BFS_THREADS), 0, 0, (const int32_t*) b->dist.p, g->begin.p, g->V, b->qcount.p);
            (void) hipMemcpy(&edges_reached, b->qcount.p, sizeof(edges_reached), hipMemcpyDeviceToHost);
        }
    }
    if (st == GMX_OK) {
        (void) hipEventRecord(ev[2], 0);
        st = gmx_bfs_download(b, dist_host, stats);
        (void) hipEventRecord(ev[3], 0);
        (void) hipEventSynchronize(ev[3]);
        if (st == GMX_OK && stats) {
            float ms = 0, cms = 0;
            (void) hipEventElapsedTime(&ms, ev[0], ev[1]);
            (void) hipEventElapsedTime(&cms, ev[2], ev[3]);
            stats->kernel_ms = ms;
            stats->d2h_ms = cms;
            stats->edges_reached = (int64_t) edges_reached;
        }
    }
    for (hipEvent_t e : ev)
        if (e) (void) hipEventDestroy(e);
    return st;
}

// ------------------------------------------------------------------ sssp
// The emitted `sssp` (/root/reference/apps/src/sssp.gm:1-30) is hop_dist's loop with an edge property:
//     <s.dist_nxt; s.updated_nxt> min= <n.dist + e.len; True>     e = the out-edge slot being walked
// until nothing changes.  dist[v] is the length of a shortest path over out-edges (INT_MAX: unreachable) --
// unique, so the device is free to relax asynchronously: the updated vertices form a queue, their out-edges
// are cut by merge-path exactly as in the top-down BFS level, every edge does atomicMin(dist[s], dist[n] +
// len[e]) in place, and a vertex whose distance dropped enters the next queue once per round (round stamp).
// Integer only: bit-exact against the CPU result.
__global__ void __launch_bounds__(BFS_THREADS)
sssp_relax_kernel(const int32_t* __restrict__ begin, const int32_t* __restrict__ node_idx, const int32_t* __restrict__ len,
                  const int32_t* __restrict__ cur_q, int64_t n, const int64_t* __restrict__ off, int64_t m,
                  int32_t round, int32_t* __restrict__ dist, int32_t* __restrict__ stamp, int32_t* __restrict__ next_q,
                  bfs_counters* __restrict__ ctr, const int64_t* __restrict__ split) {
    __shared__ int64_t s_off[BFS_ITEMS + 2];
    __shared__ int32_t s_row[BFS_ITEMS + 2];
    __shared__ int32_t s_dist[BFS_ITEMS + 2];
    __shared__ int32_t s_win[BFS_ITEMS];   // (one queue-tail claim per workgroup, as in bfs_topdown_kernel)
    __shared__ unsigned int s_nwin;
    __shared__ unsigned long long s_base;
    const int tid = threadIdx.x;
    if (tid == 0) s_nwin = 0;
    // merge-path split of the diagonals k * ITEMS and (k + 1) * ITEMS (bfs_merge_split_kernel): (rows consumed, edges consumed)
    int64_t d0 = (int64_t) blockIdx.x * BFS_ITEMS, d1 = d0 + BFS_ITEMS;
    if (d1 > n + m) d1 = n + m;
    const int64_t v0 = split[blockIdx.x], v1 = split[blockIdx.x + 1], e0 = d0 - v0, e1 = d1 - v1;
    const int nv = (int) (v1 - v0) + 1;
    for (int i = tid; i < nv; i += BFS_THREADS) {
        const int64_t vi = v0 + i;
        s_off[i] = vi <= n ? off[vi < n ? vi : n] : m;
        const int32_t v = vi < n ? cur_q[vi] : 0;
        s_row[i] = vi < n ? begin[v] : 0;
        s_dist[i] = vi < n ? dist[v] : 0;      // may already be lower than when v was queued: even better
    }
    if (tid == 0) s_off[nv] = m + 1;
    __syncthreads();
    unsigned long long inspected = 0;
    for (int64_t x = e0 + tid; x < e1; x += BFS_THREADS) {
        int lo = 0, hi = nv - 1;
        while (lo < hi) {
            int mid = (lo + hi + 1) >> 1;
            if (s_off[mid] <= x) lo = mid; else hi = mid - 1;
        }
        const int64_t e = (int64_t) s_row[lo] + (x - s_off[lo]);
        const int32_t s = node_idx[e];
        const int32_t nd = s_dist[lo] + len[e];
        inspected++;
        bool won = false;
        if (nd < dist[s] && nd < atomicMin(&dist[s], nd)) won = atomicExch(&stamp[s], round) != round;
        const unsigned long long mk = __ballot(won);
        if (mk) {
            const int lane = threadIdx.x & 63;
            const int leader = __ffsll((long long) mk) - 1;
            unsigned int at = 0;
            if (lane == leader) at = atomicAdd(&s_nwin, (unsigned int) __popcll(mk));
            at = __shfl(at, leader, 64);
            if (won) s_win[at + __popcll(mk & ((1ULL << lane) - 1))] = s;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) inspected += __shfl_down(inspected, o, 64);
    if ((tid & 63) == 0) bfs_count(ctr, inspected, 0);
    __syncthreads();
    const unsigned int nwin = s_nwin;
    if (nwin == 0) return;   // (workgroup-uniform)
    if (tid == 0) s_base = atomicAdd(&ctr->next_count, (unsigned long long) nwin);
    __syncthreads();
    for (unsigned int i = tid; i < nwin; i += BFS_THREADS) next_q[s_base + i] = s_win[i];
}

__global__ void sssp_init_kernel(int32_t* __restrict__ dist, int32_t* __restrict__ stamp, int64_t V, int32_t root) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < V; i += stride) {
        dist[i] = (i == root) ? 0 : INT_MAX;
        stamp[i] = -1;
    }
}

// hipEvents / pinned words that are released on every return path
struct ev_guard {
    hipEvent_t e = nullptr;
    ~ev_guard() { if (e) (void) hipEventDestroy(e); }
    int create() { GMX_HIP(hipEventCreate(&e)); return GMX_OK; }
};
template <typename T>
struct pinned_guard {
    T* p = nullptr;
    ~pinned_guard() { if (p) (void) hipHostFree(p); }
    int alloc() { GMX_HIP(hipHostMalloc((void**) &p, sizeof(T), hipHostMallocDefault)); return GMX_OK; }
};

// dst[j] = src[order[j]]: an edge property given by uploaded slot, brought into the order of the sorted rows
__global__ void gather_by_order_kernel(const int32_t* __restrict__ src, const int32_t* __restrict__ order, int64_t n, int32_t* __restrict__ dst) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = src[order[i]];
}

extern "C" int gmx_sssp(gmx_graph_t* g, gmx_node_t root, const int32_t* len_host, int32_t* dist_host, gmx_stats_t* stats) {
    GMX_REQUIRE(g && dist_host, "NULL argument");
    GMX_REQUIRE(len_host || g->E == 0, "len is NULL");
    if (stats) memset(stats, 0, sizeof(*stats));
    const int64_t V = g->V;
    if (V == 0) return GMX_OK;
    const bool root_ok = root >= 0 && root < V;
    dbuf<int32_t> dist, stamp, q0, q1, deg, len;
    dbuf<int64_t> off, split;
    dbuf<bfs_counters> ctr;
    dbuf<char> scan_tmp;
    size_t scan_bytes = 0;
    GMX_CHECK(split.alloc((size_t) ((V + g->E) / BFS_ITEMS + 3)));   // one entry per merge-path diagonal of a round
    GMX_CHECK(dist.alloc((size_t) V));
    GMX_CHECK(stamp.alloc((size_t) V));
    GMX_CHECK(q0.alloc((size_t) V));
    GMX_CHECK(q1.alloc((size_t) V));
    GMX_CHECK(deg.alloc((size_t) V));
    GMX_CHECK(off.alloc((size_t) V + 2));
    GMX_CHECK(ctr.alloc(1));
    GMX_CHECK(len.alloc((size_t) (g->E ? g->E : 1)));
    GMX_HIP(rocprim::inclusive_scan(nullptr, scan_bytes, deg.p, off.p + 1, (size_t) V, rocprim::plus<int64_t>(), 0));
    GMX_CHECK(scan_tmp.alloc(scan_bytes));
    ev_guard evg[4];
    hipEvent_t ev[4];
    for (int i = 0; i < 4; i++) {
        GMX_CHECK(evg[i].create());
        ev[i] = evg[i].e;
    }
    GMX_HIP(hipEventRecord(ev[2], 0));
    if (g->E) GMX_HIP(hipMemcpy(len.p, len_host, sizeof(int32_t) * (size_t) g->E, hipMemcpyHostToDevice));   // the property is the caller's
    dbuf<int32_t> len_sorted;
    if (g->E && g->e_idx2idx.p) {
        // the rows were sorted on upload: len[] is indexed by the caller's (unsorted) slots, the kernel walks the sorted ones
        GMX_CHECK(len_sorted.alloc((size_t) g->E));
        hipLaunchKernelGGL(gather_by_order_kernel, dim3(grid_for(g->E)), dim3(BFS_THREADS), 0, 0, (const int32_t*) len.p,
                           (const int32_t*) g->e_idx2idx.p, g->E, len_sorted.p);
    }
    const int32_t* len_dev = len_sorted.p ? len_sorted.p : len.p;
    GMX_HIP(hipEventRecord(ev[3], 0));
    GMX_HIP(hipEventRecord(ev[0], 0));
    hipLaunchKernelGGL(sssp_init_kernel, dim3(grid_for(V)), dim3(BFS_THREADS), 0, 0, dist.p, stamp.p, V, root_ok ? root : -1);
    int64_t cur_count = 0, requeued = 0;
    unsigned long long edges = 0;
    int32_t round = 0;
    int32_t* cur_q = q0.p;
    int32_t* next_q = q1.p;
    if (root_ok) {
        GMX_HIP(hipMemcpy(q0.p, &root, sizeof(int32_t), hipMemcpyHostToDevice));
        cur_count = 1;
    }
    // per-round read-backs through pinned memory (two host round trips per round, dozens of rounds)
    pinned_guard<bfs_counters> pg_ctr;
    pinned_guard<int64_t> pg_mf;
    GMX_CHECK(pg_ctr.alloc());
    GMX_CHECK(pg_mf.alloc());
    bfs_counters* h_ctr = pg_ctr.p;
    int64_t* h_mf = pg_mf.p;
    GMX_HIP(hipMemsetAsync(ctr.p, 0, sizeof(bfs_counters), 0));
    while (cur_count > 0) {
        GMX_HIP(hipMemsetAsync(&ctr.p->next_count, 0, sizeof(unsigned long long), 0));   // `edges` keeps accumulating
        hipLaunchKernelGGL(bfs_degree_kernel, dim3(grid_for(cur_count)), dim3(BFS_THREADS), 0, 0, g->begin.p, cur_q, cur_count, deg.p);
        size_t tb = scan_bytes;
        GMX_HIP(rocprim::inclusive_scan(scan_tmp.p, tb, deg.p, off.p + 1, (size_t) cur_count, rocprim::plus<int64_t>(), 0));
        GMX_HIP(hipMemsetAsync(off.p, 0, sizeof(int64_t), 0));
        GMX_HIP(hipMemcpyAsync(h_mf, off.p + cur_count, sizeof(int64_t), hipMemcpyDeviceToHost, 0));
        GMX_HIP(hipStreamSynchronize(0));
        const int64_t m_f = *h_mf;
        const int64_t nb = (cur_count + m_f + BFS_ITEMS - 1) / BFS_ITEMS;
        if (nb > 0) {
            hipLaunchKernelGGL(bfs_merge_split_kernel, dim3((unsigned) ((nb + 1 + BFS_THREADS - 1) / BFS_THREADS)), dim3(BFS_THREADS), 0, 0,
                               (const int64_t*) off.p, cur_count, m_f, nb, split.p);
            hipLaunchKernelGGL(sssp_relax_kernel, dim3((unsigned) nb), dim3(BFS_THREADS), 0, 0, g->begin.p, g->node_idx.p,
                               len_dev, cur_q, cur_count, off.p, m_f, round, dist.p, stamp.p, next_q, ctr.p, (const int64_t*) split.p);
        }
        GMX_HIP(hipGetLastError());
        GMX_HIP(hipMemcpyAsync(h_ctr, ctr.p, sizeof(bfs_counters), hipMemcpyDeviceToHost, 0));
        GMX_HIP(hipStreamSynchronize(0));
        const bfs_counters& h = *h_ctr;
        unsigned long long found_unused = 0;
        cur_count = (int64_t) h.next_count;
        bfs_totals(h, &edges, &found_unused);
        requeued += cur_count;
        int32_t* t = cur_q;
        cur_q = next_q;
        next_q = t;
        round++;
    }
    GMX_HIP(hipEventRecord(ev[1], 0));
    GMX_HIP(hipEventSynchronize(ev[1]));
    GMX_HIP(hipMemcpy(dist_host, dist.p, sizeof(int32_t) * (size_t) V, hipMemcpyDeviceToHost));
    if (stats) {
        float ms = 0, hms = 0;
        (void) hipEventElapsedTime(&ms, ev[0], ev[1]);
        (void) hipEventElapsedTime(&hms, ev[2], ev[3]);
        stats->iterations = round;
        stats->kernel_ms = ms;
        stats->h2d_ms = hms;
        stats->edges_examined = (int64_t) edges;
        stats->vertices_reached = requeued + (root_ok ? 1 : 0);   // queue entries over all rounds (a vertex may re-enter)
    }
    return GMX_OK;
}

// ------------------------------------------------------------------ BFS object: InBFS / InReverse
// The device counterpart of gm_bfs_template<level_t = short, ..., save_child> (gm_bfs_template.h:14-312) as the
// emitted `InBFS(v: G.Nodes From s) {..} InReverse {..}` uses it (gm_cpp_gen_bfs.cc:88-275):
//   prepare(root) + do_bfs_forward()   the direction-optimising traversal above records every vertex's level; the
//                                      vertices are then listed level by level (the template's level queues);
//   visit_fw(v) for every v, level 0 first; do_bfs_reverse(): visit_rv(v), deepest level first (:272-312);
//   v.UpNbrs   = slots of v's REVERSE row whose source lies one level up   (gm_cpp_gen_foreach.cc:173: the emitted
//                `if (get_level(w) != (get_curr_level() - 1)) continue;`)
//   v.DownNbrs = slots of v's forward row that are down edges; an edge v -> u is one iff level(u) == level(v) + 1
//                (iterate_neighbor_* :581-682).  The device derives that from the levels instead of keeping the
//                template's one byte per edge; where the template drops down edges (bottom-up levels record none,
//                :400-403,689-708; several threads race on the child's level, :596-620) this is the definition
//                those flags were meant to implement (tests/golden/manifest.json lists the fixtures where the compiled
//                reference deviates from it).
// A visit is a device callback: one thread per vertex of the level walks the row in slot order, so a float Sum adds
// its terms in exactly the emitted order and the results are bit-identical to the single-threaded reference.
struct bfs_order {
    dbuf<uint32_t> key, key2;     // levels as sort keys (unreached: INT_MAX, last)
    dbuf<int32_t> id, order;      // order[]: vertices by level
    dbuf<int64_t> off;            // [levels + 1] first entry of every level in order[]
    dbuf<char> tmp;
    size_t tmp_bytes = 0;
    std::vector<int64_t> h_off;
    int32_t levels = 0;           // number of levels (deepest + 1)
};

__global__ void bfs_order_key_kernel(const int32_t* __restrict__ dist, int64_t V, uint32_t* __restrict__ key, int32_t* __restrict__ id) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < V; i += stride) {
        key[i] = (uint32_t) dist[i];
        id[i] = (int32_t) i;
    }
}
__global__ void bfs_order_off_kernel(const uint32_t* __restrict__ key, int64_t V, int32_t levels, int64_t* __restrict__ off) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l > levels) return;
    int64_t lo = 0, hi = V;   // first entry with level >= l
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (key[mid] < (uint32_t) l) lo = mid + 1; else hi = mid;
    }
    off[l] = lo;
}

// b has finished a traversal: list its vertices by level
static int bfs_make_order(gmx_bfs* b, bfs_order* o) {
    const int64_t V = b->V;
    if (!o->key.p) {
        GMX_CHECK(o->key.alloc((size_t) V));
        GMX_CHECK(o->key2.alloc((size_t) V));
        GMX_CHECK(o->id.alloc((size_t) V));
        GMX_CHECK(o->order.alloc((size_t) V));
        GMX_HIP(rocprim::radix_sort_pairs(nullptr, o->tmp_bytes, o->key.p, o->key2.p, o->id.p, o->order.p, (size_t) V, 0u, 32u, 0));
        GMX_CHECK(o->tmp.alloc(o->tmp_bytes));
    }
    o->levels = b->level;   // levels completed by the traversal = deepest level + 1
    hipLaunchKernelGGL(bfs_order_key_kernel, dim3(grid_for(V)), dim3(BFS_THREADS), 0, 0, (const int32_t*) b->dist.p, V, o->key.p, o->id.p);
    size_t tb = o->tmp_bytes;
    GMX_HIP(rocprim::radix_sort_pairs((void*) o->tmp.p, tb, o->key.p, o->key2.p, o->id.p, o->order.p, (size_t) V, 0u, 32u, 0));
    if (o->off.n < (size_t) o->levels + 2) GMX_CHECK(o->off.alloc((size_t) o->levels + 2));
    hipLaunchKernelGGL(bfs_order_off_kernel, dim3((o->levels + 1 + 255) / 256), dim3(256), 0, 0, (const uint32_t*) o->key2.p, V, o->levels, o->off.p);
    o->h_off.resize((size_t) o->levels + 1);
    GMX_HIP(hipMemcpy(o->h_off.data(), o->off.p, sizeof(int64_t) * ((size_t) o->levels + 1), hipMemcpyDeviceToHost));
    return GMX_OK;
}

// root's traversal -> b->dist; returns when it is complete
static int bfs_run(gmx_bfs* b, gmx_node_t root) {
    GMX_CHECK(gmx_bfs_start(b, root));
    int64_t next = 0;
    int need = 0;
    do {
        GMX_CHECK(gmx_bfs_step_begin(b, &need));
        GMX_CHECK(gmx_bfs_step_end(b, &next));
    } while (next > 0);
    return GMX_OK;
}

// ---- the visits of an InBFS traversal (gm_bfs_template.h:69-312: visit_fw / visit_rv, as gm_cpp_gen_bfs.cc:88-275 emits
// them) as device functors.  A VISIT is: for every vertex v of a level, S = Sum over its UpNbrs (in-neighbours one level
// closer, DIR = -1, through the reverse CSR) or DownNbrs (out-neighbours one level deeper, DIR = +1: the template's
// is_down_edge) of term(v, w), then finish(v, S).  A new InBFS app is two structs with
//     static constexpr int DIR;  float prep(v) const;  float term(float prep_of_v, w) const;  void finish(v, S) const;
// handed to bfs_sweep(): comp_BC below is the first.
// The Float sums must round exactly like the sequential emission -- S = S + term, one term after the other in row-slot
// order -- so the ADDS of a row are a serial chain whatever is done.  Everything else is parallel: a wave takes 64 rows of
// the level; rows shorter than BFS_VISIT_SMALL are walked one per lane; the longer rows go onto the level's list and are
// taken, one wave per row, 64 slots at a time: coalesced loads of the slots, the neighbours' levels and values gathered
// by 64 lanes at once, the filter as one ballot, each lane's term computed with the emission's own float expression,
// and the passing terms added in slot order through v_readlane (hub rows of 10^5 slots were one lane's dependent-load
// chain in round 2: comp_BC on RMAT-24, five seeds, 1.69 s -> 0.04 s).
#define BFS_VISIT_SMALL 32
#define BFS_DENSE_PASS 16  // passing lanes of a chunk from which all 64 lanes are added in order (the others as +0.0f)
#define BFS_BIG_CHUNKS 8   // 64-slot chunks of a long row in flight per step

// S = S + term for every set bit of `pass`, ascending lane = ascending slot (wave-uniform result)
// the same sum when most lanes pass: all 64 lanes in order, the others contributing +0.0f -- no loop over the mask's bits, the
// lane indices are constants.  x + (+0.0f) == x bit for bit unless x is -0.0f, so this is for visits whose terms are never
// negative (Visit::NONNEG): S then starts at +0.0f and stays non-negative.
__device__ __forceinline__ float bfs_ordered_add_dense(float S, float term, bool pass) {
    const int t = __builtin_bit_cast(int, pass ? term : 0.0f);
#pragma unroll
    for (int j = 0; j < 64; j++) S = S + __builtin_bit_cast(float, __builtin_amdgcn_readlane(t, j));
    return S;
}
__device__ __forceinline__ float bfs_ordered_add(float S, float term, unsigned long long pass) {
    while (pass) {
        const int j = __builtin_ctzll(pass);
        S = S + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, term), j));
        pass &= pass - 1;
    }
    return S;
}

// the wave's long rows go onto the level's list (one atomic per wave), to be taken by bfs_visit_big_kernel
__device__ __forceinline__ void bfs_append_big(bool is_big, int32_t v, int lane, int32_t* __restrict__ big_list, unsigned int* __restrict__ big_count) {
    const unsigned long long m = __ballot(is_big);
    if (!m) return;
    unsigned int at = 0;
    if (lane == 0) at = atomicAdd(big_count, (unsigned int) __builtin_popcountll(m));
    at = __builtin_amdgcn_readfirstlane(at);
    if (is_big) big_list[at + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = v;
}

// level `level` = order[lo, hi); begin / idx: the reverse CSR for DIR = -1, the forward CSR for DIR = +1
template <class Visit>
__global__ void __launch_bounds__(BFS_THREADS)
bfs_visit_kernel(const int32_t* __restrict__ order, int64_t lo, int64_t hi, const uint32_t* __restrict__ nbr_level_bm,
                 const int32_t* __restrict__ begin, const int32_t* __restrict__ idx, int32_t skip, Visit vis,
                 int32_t* __restrict__ big_list, unsigned int* __restrict__ big_count) {
    const int lane = threadIdx.x & 63;
    int64_t base = lo + ((int64_t) blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64;
    const int64_t stride = (int64_t) gridDim.x * (blockDim.x >> 6) * 64;
    for (; base < hi; base += stride) {   // (wave-uniform)
        const int64_t i = base + lane;
        int32_t v = -1, rb = 0, deg = 0;
        if (i < hi) {
            v = order[i];
            if (v == skip) v = -1;
            else { rb = begin[v]; deg = begin[v + 1] - rb; }
        }
        if (v >= 0 && deg < BFS_VISIT_SMALL) {
            const float pv = vis.prep(v);
            float S = 0.0f;
            for (int32_t e = rb; e < rb + deg; e++) {
                const int32_t w = idx[e];
                if (!((nbr_level_bm[w >> 5] >> (w & 31)) & 1u)) continue;   // (w is not one level up / down)
                S = S + vis.term(pv, w);
            }
            vis.finish(v, S);
        }
        bfs_append_big(v >= 0 && deg >= BFS_VISIT_SMALL, v, lane, big_list, big_count);
    }
}

// the long rows of the level, one wave per row (any wave takes any row: a level's hubs run side by side)
template <class Visit>
__global__ void __launch_bounds__(BFS_THREADS)
bfs_visit_big_kernel(const int32_t* __restrict__ big_list, const unsigned int* __restrict__ big_count, const uint32_t* __restrict__ nbr_level_bm,
                     const int32_t* __restrict__ begin, const int32_t* __restrict__ idx, Visit vis) {
    const int lane = threadIdx.x & 63;
    const unsigned int n = *big_count, nwaves = gridDim.x * (blockDim.x >> 6);
    for (unsigned int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); k < n; k += nwaves) {   // (wave-uniform)
        const int32_t vb = big_list[k], rbb = begin[vb], degb = begin[vb + 1] - rbb;
        const float pv = vis.prep(vb);
        float S = 0.0f;
        // BFS_BIG_CHUNKS x 64 slots per step: the slots, then their level bits, then the passing slots' terms are each
        // requested for all chunks of the step before the first is used -- the adds stay one chain in slot order, but a hub's
        // row (10^5 slots, one wave) is no longer 1600 round trips of three dependent loads each (3.2 ms per level, the
        // level's whole time)
        for (int32_t c = 0; c < degb; c += 64 * BFS_BIG_CHUNKS) {
            int32_t w[BFS_BIG_CHUNKS];
            uint32_t bits[BFS_BIG_CHUNKS];
            bool pass[BFS_BIG_CHUNKS];
            float term[BFS_BIG_CHUNKS];
#pragma unroll
            for (int k = 0; k < BFS_BIG_CHUNKS; k++) {
                const int32_t e = c + 64 * k + lane;
                w[k] = idx[rbb + (e < degb ? e : degb - 1)];
            }
#pragma unroll
            for (int k = 0; k < BFS_BIG_CHUNKS; k++) bits[k] = nbr_level_bm[w[k] >> 5];
#pragma unroll
            for (int k = 0; k < BFS_BIG_CHUNKS; k++) {
                pass[k] = c + 64 * k + lane < degb && ((bits[k] >> (w[k] & 31)) & 1u);
                term[k] = pass[k] ? vis.term(pv, w[k]) : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < BFS_BIG_CHUNKS; k++) {
                const unsigned long long m = __ballot(pass[k]);
                if (Visit::NONNEG && __builtin_popcountll(m) > BFS_DENSE_PASS) S = bfs_ordered_add_dense(S, term[k], pass[k]);
                else S = bfs_ordered_add(S, term[k], m);
            }
        }
        if (lane == 0) vis.finish(vb, S);
    }
}

// one level of a visit: the short rows and the list of the long ones, then the long ones
template <class Visit>
static int bfs_visit_level(gmx_graph* g, gmx_bfs* b, const int32_t* order, int64_t lo, int64_t hi, int32_t level, int32_t skip, const Visit& vis,
                           int32_t* big_list, unsigned int* big_count) {
    if (hi <= lo) return GMX_OK;
    const int32_t* begin = Visit::DIR < 0 ? g->r_begin.p : g->begin.p;
    const int32_t* idx = Visit::DIR < 0 ? g->r_node_idx.p : g->node_idx.p;
    GMX_HIP(hipMemsetAsync(big_count, 0, sizeof(unsigned int), 0));
    // "w is one level up / down" as a bitmap of that level (V / 8 bytes: it stays in the L2s) instead of a gather from level[]
    // per slot -- a 64-byte line for four bytes, for every slot of every row, where only the slots that pass go on to gather
    // sigma / delta (RMAT-24, the hub and four more seeds: 120 -> 104 ms).  The traversal's bitmaps are idle during the sweeps.
    const uint32_t* nbr_bm = (const uint32_t*) b->bm[0].p;
    hipLaunchKernelGGL(bfs_level_bitmap_kernel, dim3(grid_for(b->V, BFS_THREADS, 256 * 16)), dim3(BFS_THREADS), 0, 0, (const int32_t*) b->dist.p, b->V,
                       level + Visit::DIR, b->bm[0].p);
    hipLaunchKernelGGL((bfs_visit_kernel<Visit>), dim3(grid_for(hi - lo)), dim3(BFS_THREADS), 0, 0, order, lo, hi, nbr_bm,
                       begin, idx, skip, vis, big_list, big_count);
    hipLaunchKernelGGL((bfs_visit_big_kernel<Visit>), dim3(grid_for(hi - lo, 4, 1024)), dim3(BFS_THREADS), 0, 0, (const int32_t*) big_list,
                       (const unsigned int*) big_count, nbr_bm, begin, idx, vis);
    return GMX_OK;
}

// InBFS(v: G.Nodes From s) { visit_fw } InReverse { visit_rv }: the forward visit level by level from the root, the
// reverse visit from the deepest level back (gm_bfs_template.h:69-312); `ord` holds the traversal's level order
template <class VisitFw, class VisitRv>
static int bfs_sweep(gmx_graph* g, gmx_bfs* b, const bfs_order& ord, int32_t skip, const VisitFw& fw, const VisitRv& rv, int32_t* big_list, unsigned int* big_count) {
    for (int32_t l = 0; l < ord.levels; l++)
        GMX_CHECK(bfs_visit_level(g, b, (const int32_t*) ord.order.p, ord.h_off[(size_t) l], ord.h_off[(size_t) l + 1], l, skip, fw, big_list, big_count));
    for (int32_t l = ord.levels - 1; l >= 0; l--)
        GMX_CHECK(bfs_visit_level(g, b, (const int32_t*) ord.order.p, ord.h_off[(size_t) l], ord.h_off[(size_t) l + 1], l, skip, rv, big_list, big_count));
    GMX_HIP(hipGetLastError());
    return GMX_OK;
}

// comp_BC's two visits (apps/src/bc.gm:16-29)
// sigma and delta of a vertex side by side (x = sigma, y = delta): the reverse visit needs both of every down-neighbour that
// passes, and one 8-byte gather is one line request where two 4-byte gathers from two arrays are two
struct bc_visit_fw {   // v.sigma = Sum(w: v.UpNbrs){ w.sigma }
    static constexpr int DIR = -1;
    static constexpr bool NONNEG = true;   // path counts
    float2* sd;
    __device__ float prep(int32_t) const { return 0.0f; }
    __device__ float term(float, int32_t w) const { return sd[w].x; }
    __device__ void finish(int32_t v, float S) const { sd[v].x = S; }
};
struct bc_visit_rv {   // v.delta = Sum(w: v.DownNbrs){ v.sigma / w.sigma * (1 + w.delta) };  v.BC += v.delta
    static constexpr int DIR = +1;
    static constexpr bool NONNEG = true;   // sigma > 0 on reached vertices, delta >= 0
    float2* sd;
    float* bc;
    __device__ float prep(int32_t v) const { return sd[v].x; }
    __device__ float term(float sv, int32_t w) const {
        const float2 q = sd[w];
        return sv / q.x * (1 + q.y);
    }
    __device__ void finish(int32_t v, float S) const {
        sd[v].y = S;
        bc[v] = bc[v] + S;
    }
};

__global__ void fill_sigma_kernel(float2* __restrict__ sd, int64_t n, int64_t one_at) {   // G.sigma = 0; s.sigma = 1 (delta stays)
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < n; i += stride) sd[i].x = i == one_at ? 1.0f : 0.0f;
}

__global__ void fill_f32_kernel(float* __restrict__ p, int64_t n, float v, int64_t one_at, float one_v) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = i == one_at ? one_v : v;
}

extern "C" int gmx_bc(gmx_graph_t* g, const gmx_node_t* seeds, int32_t nseeds, int skip_root, float* bc_host, gmx_stats_t* stats) {
    GMX_REQUIRE(g && bc_host && (seeds || nseeds == 0) && nseeds >= 0, "bad argument");
    GMX_REQUIRE(g->has_reverse, "comp_BC needs the reverse CSR (UpNbrs)");
    if (stats) memset(stats, 0, sizeof(*stats));
    const int64_t V = g->V;
    if (V == 0) return GMX_OK;
    for (int32_t i = 0; i < nseeds; i++) GMX_REQUIRE(seeds[i] >= 0 && seeds[i] < V, "seed %d out of range", seeds[i]);
    if (!g->bfs_cache) GMX_CHECK(gmx_bfs_create(g, 0, 1, &g->bfs_cache));
    gmx_bfs* b = g->bfs_cache;
    dbuf<float2> sd;
    dbuf<float> bc;
    dbuf<int32_t> big_list;
    dbuf<unsigned int> big_count;
    GMX_CHECK(big_list.alloc((size_t) V));
    GMX_CHECK(big_count.alloc(1));
    GMX_CHECK(sd.alloc((size_t) V));
    GMX_HIP(hipMemsetAsync(sd.p, 0, sizeof(float2) * (size_t) V, 0));   // (delta of a vertex no visit has written is never read; zero all the same)
    GMX_CHECK(bc.alloc((size_t) V));
    bfs_order ord;
    ev_guard e0, e1;
    GMX_CHECK(e0.create());
    GMX_CHECK(e1.create());
    GMX_HIP(hipEventRecord(e0.e, 0));
    hipLaunchKernelGGL(fill_f32_kernel, dim3(grid_for(V)), dim3(BFS_THREADS), 0, 0, bc.p, V, 0.0f, (int64_t) -1, 0.0f);   // G.BC = 0
    int64_t reached = 0;
    for (int32_t si = 0; si < nseeds; si++) {   // For (s: Seeds.Items): sequential, as emitted
        const gmx_node_t s = seeds[si];
        hipLaunchKernelGGL(fill_sigma_kernel, dim3(grid_for(V)), dim3(BFS_THREADS), 0, 0, sd.p, V, (int64_t) s);   // G.sigma = 0; s.sigma = 1
        GMX_CHECK(bfs_run(b, s));
        GMX_CHECK(bfs_make_order(b, &ord));
        reached += ord.h_off[(size_t) ord.levels];
        const int32_t skip = skip_root ? s : -1;
        GMX_CHECK(bfs_sweep(g, b, ord, skip, bc_visit_fw{sd.p}, bc_visit_rv{sd.p, bc.p}, big_list.p, big_count.p));
    }
    GMX_HIP(hipEventRecord(e1.e, 0));
    GMX_HIP(hipEventSynchronize(e1.e));
    GMX_HIP(hipMemcpy(bc_host, bc.p, sizeof(float) * (size_t) V, hipMemcpyDeviceToHost));
    if (stats) {
        float ms = 0;
        (void) hipEventElapsedTime(&ms, e0.e, e1.e);
        stats->iterations = nseeds;
        stats->kernel_ms = ms;
        stats->vertices_reached = reached;
    }
    return GMX_OK;
}

// prepare(root) + do_bfs_forward() alone: every vertex's level as the template keeps it (level_t = short, unvisited
// vertices hold __INVALID_LEVEL = -2, gm_bfs_template.h:725) and the number of levels
__global__ void bfs_level_short_kernel(const int32_t* __restrict__ dist, int64_t V, int16_t* __restrict__ level) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < V; i += stride) level[i] = dist[i] == INT_MAX ? (int16_t) -2 : (int16_t) dist[i];
}

extern "C" int gmx_bfs_levels(gmx_graph_t* g, gmx_node_t root, int16_t* level_host, int32_t* nlevels) {
    GMX_REQUIRE(g && level_host, "NULL argument");
    if (nlevels) *nlevels = 0;
    const int64_t V = g->V;
    if (V == 0) return GMX_OK;
    GMX_REQUIRE(root >= 0 && root < V, "root %d out of range", root);
    if (!g->bfs_cache) GMX_CHECK(gmx_bfs_create(g, 0, 1, &g->bfs_cache));
    gmx_bfs* b = g->bfs_cache;
    GMX_CHECK(bfs_run(b, root));
    GMX_REQUIRE(b->level < 32767, "%d levels do not fit level_t = short", b->level);
    dbuf<int16_t> lv;
    GMX_CHECK(lv.alloc((size_t) V));
    hipLaunchKernelGGL(bfs_level_short_kernel, dim3(grid_for(V)), dim3(BFS_THREADS), 0, 0, (const int32_t*) b->dist.p, V, lv.p);
    GMX_HIP(hipMemcpy(level_host, lv.p, sizeof(int16_t) * (size_t) V, hipMemcpyDeviceToHost));
    if (nlevels) *nlevels = b->level;
    return GMX_OK;
}

// ------------------------------------------------------------------ avg_teen_cnt, conduct
// Count-reductions over neighbours with an integer node property (/root/reference/apps/src/avg_teen_cnt.gm,
// conduct.gm).  Both are "expand the out-edges of the vertices that pass a filter and do something per edge":
// the selected vertices are queued (ballot-aggregated append), their out-degrees prefix-summed and the edges
// cut by merge-path as in a top-down BFS level.  Integer arithmetic only; the final float is formed on the
// host from the exact integers with the emitted expression, so results are bit-identical.
//   MODE 0: cnt[dst] += 1                     (teen_cnt[n] = #in-neighbours passing the filter)
//   MODE 1: total += (prop[dst] != num)       (conduct's Cross)
template <int MODE>
__global__ void __launch_bounds__(BFS_THREADS)
edge_count_kernel(const int32_t* __restrict__ begin, const int32_t* __restrict__ node_idx,
                  const int32_t* __restrict__ cur_q, int64_t n, const int64_t* __restrict__ off, int64_t m,
                  const int32_t* __restrict__ prop, int32_t num, int32_t* __restrict__ cnt,
                  unsigned long long* __restrict__ total) {
    __shared__ int64_t s_off[BFS_ITEMS + 2];
    __shared__ int32_t s_row[BFS_ITEMS + 2];
    __shared__ int64_t s_split[2][2];
    const int tid = threadIdx.x;
    if (tid < 2) {
        int64_t dk = ((int64_t) blockIdx.x + tid) * BFS_ITEMS;
        if (dk > n + m) dk = n + m;
        int64_t lo = dk > m ? dk - m : 0, hi = dk < n ? dk : n;
        while (lo < hi) {
            int64_t mid = (lo + hi) >> 1;
            if (off[mid + 1] <= dk - mid - 1) lo = mid + 1; else hi = mid;
        }
        s_split[tid][0] = lo;
        s_split[tid][1] = dk - lo;
    }
    __syncthreads();
    const int64_t v0 = s_split[0][0], e0 = s_split[0][1], v1 = s_split[1][0], e1 = s_split[1][1];
    const int nv = (int) (v1 - v0) + 1;
    for (int i = tid; i < nv; i += BFS_THREADS) {
        const int64_t vi = v0 + i;
        s_off[i] = vi <= n ? off[vi < n ? vi : n] : m;
        s_row[i] = vi < n ? begin[cur_q[vi]] : 0;
    }
    if (tid == 0) s_off[nv] = m + 1;
    __syncthreads();
    unsigned long long acc = 0;
    for (int64_t x = e0 + tid; x < e1; x += BFS_THREADS) {
        int lo = 0, hi = nv - 1;
        while (lo < hi) {
            int mid = (lo + hi + 1) >> 1;
            if (s_off[mid] <= x) lo = mid; else hi = mid - 1;
        }
        const int32_t s = node_idx[(int64_t) s_row[lo] + (x - s_off[lo])];
        if (MODE == 0) atomicAdd(&cnt[s], 1);
        else acc += prop[s] != num ? 1 : 0;
    }
    if (MODE == 1) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        if ((tid & 63) == 0 && acc) atomicAdd(total, acc);
    }
}

// filter: 0: 10 <= prop < 20 (teen), 1: prop == num
__global__ void select_queue_kernel(const int32_t* __restrict__ prop, int64_t V, int filter, int32_t num,
                                    int32_t* __restrict__ q, unsigned long long* __restrict__ qcount) {
    int64_t v = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    const int64_t vend = (V + 63) / 64 * 64;
    for (; v < vend; v += stride) {
        bool in = false;
        if (v < V) {
            const int32_t p = prop[v];
            in = filter == 0 ? (p >= 10 && p < 20) : (p == num);
        }
        const unsigned long long mk = __ballot(in);
        if (mk) {
            const int lane = threadIdx.x & 63;
            const int leader = __ffsll((long long) mk) - 1;
            unsigned long long base = 0;
            if (lane == leader) base = atomicAdd(qcount, (unsigned long long) __popcll(mk));
            base = __shfl(base, leader, 64);
            if (in) q[base + __popcll(mk & ((1ULL << lane) - 1))] = (int32_t) v;
        }
    }
}

// out[0] += sum of val[v] (or of the out-degree) over the vertices passing the test, out[1] += their number
//   test 0: prop[v] > num   test 1: prop[v] == num   test 2: prop[v] != num
__global__ void filtered_sum_kernel(const int32_t* __restrict__ prop, const int32_t* __restrict__ val, const int32_t* __restrict__ begin,
                                    int64_t V, int test, int32_t num, unsigned long long* __restrict__ out) {
    int64_t v = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    unsigned long long s = 0, c = 0;
    for (; v < V; v += stride) {
        const int32_t p = prop[v];
        const bool ok = test == 0 ? p > num : test == 1 ? p == num : p != num;
        if (ok) {
            s += (unsigned long long) (long long) (val ? val[v] : begin[v + 1] - begin[v]);
            c++;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_down(s, o, 64);
        c += __shfl_down(c, o, 64);
    }
    // one pair of adds per workgroup (a pair per wave, all on one cache line, was 225 us of adds for a 20 us pass)
    __shared__ unsigned long long s_s[BFS_THREADS / 64], s_c[BFS_THREADS / 64];
    if ((threadIdx.x & 63) == 0) {
        s_s[threadIdx.x >> 6] = s;
        s_c[threadIdx.x >> 6] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long ts = 0, tc = 0;
        for (int i = 0; i < (int) (blockDim.x >> 6); i++) {
            ts += s_s[i];
            tc += s_c[i];
        }
        if (ts) atomicAdd(&out[0], ts);
        if (tc) atomicAdd(&out[1], tc);
    }
}

// queue the vertices passing `filter`, then run edge_count_kernel<MODE> over their out-edges
template <int MODE>
static int expand_selected(gmx_graph* g, const int32_t* prop, int filter, int32_t num, int32_t* cnt, unsigned long long* total) {
    const int64_t V = g->V;
    dbuf<int32_t> q, deg;
    dbuf<int64_t> off;
    dbuf<unsigned long long> qcount;
    dbuf<char> scan_tmp;
    size_t scan_bytes = 0;
    GMX_CHECK(q.alloc((size_t) V));
    GMX_CHECK(deg.alloc((size_t) V));
    GMX_CHECK(off.alloc((size_t) V + 2));
    GMX_CHECK(qcount.alloc(1));
    GMX_HIP(hipMemset(qcount.p, 0, sizeof(unsigned long long)));
    hipLaunchKernelGGL(select_queue_kernel, dim3(grid_for(V, BFS_THREADS, 256 * 16)), dim3(BFS_THREADS), 0, 0, prop, V, filter, num, q.p, qcount.p);
    unsigned long long nq = 0;
    GMX_HIP(hipMemcpy(&nq, qcount.p, sizeof(nq), hipMemcpyDeviceToHost));
    if (nq == 0) return GMX_OK;
    hipLaunchKernelGGL(bfs_degree_kernel, dim3(grid_for((int64_t) nq)), dim3(BFS_THREADS), 0, 0, g->begin.p, q.p, (int64_t) nq, deg.p);
    GMX_HIP(rocprim::inclusive_scan(nullptr, scan_bytes, deg.p, off.p + 1, (size_t) nq, rocprim::plus<int64_t>(), 0));
    GMX_CHECK(scan_tmp.alloc(scan_bytes));
    GMX_HIP(rocprim::inclusive_scan(scan_tmp.p, scan_bytes, deg.p, off.p + 1, (size_t) nq, rocprim::plus<int64_t>(), 0));
    GMX_HIP(hipMemsetAsync(off.p, 0, sizeof(int64_t), 0));
    int64_t m_f = 0;
    GMX_HIP(hipMemcpy(&m_f, off.p + nq, sizeof(int64_t), hipMemcpyDeviceToHost));
    const int64_t nb = ((int64_t) nq + m_f + BFS_ITEMS - 1) / BFS_ITEMS;
    if (nb > 0)
        hipLaunchKernelGGL(edge_count_kernel<MODE>, dim3((unsigned) nb), dim3(BFS_THREADS), 0, 0, g->begin.p, g->node_idx.p,
                           (const int32_t*) q.p, (int64_t) nq, (const int64_t*) off.p, m_f, prop, num, cnt, total);
    GMX_HIP(hipGetLastError());
    return GMX_OK;
}

// Pull formulation with the predicate as a bitmap (V/8 bytes: resident in every L2): a row walks its neighbour
// list and probes the bitmap -- no atomics, no gathers from a V-sized array.  Rows longer than ROWCNT_LONG are
// left to whole waves.  mode bits: 1 = count the neighbours whose bit is CLEAR (else set); rows are taken only
// if their own bit in `row_bm` is set (row_bm == NULL: all rows).  Per-row counts go to cnt (if given), their
// sum to total (if given).
#define ROWCNT_LONG 256
__global__ void pred_bitmap_kernel(const int32_t* __restrict__ prop, int64_t V, int filter, int32_t num, unsigned long long* __restrict__ bm64) {
    int64_t v = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    const int64_t vend = (V + 63) / 64 * 64;
    for (; v < vend; v += stride) {
        bool in = false;
        if (v < V) {
            const int32_t p = prop[v];
            in = filter == 0 ? (p >= 10 && p < 20) : (p == num);
        }
        const unsigned long long m = __ballot(in);
        if ((threadIdx.x & 63) == 0) bm64[v >> 6] = m;
    }
}

__global__ void __launch_bounds__(BFS_THREADS)
row_count_kernel(const int32_t* __restrict__ begin, const int32_t* __restrict__ idx, int64_t V,
                 const uint32_t* __restrict__ row_bm, const uint32_t* __restrict__ probe_bm, int invert,
                 int32_t* __restrict__ cnt, int32_t* __restrict__ long_rows, unsigned long long* __restrict__ nlong,
                 unsigned long long* __restrict__ total) {
    int64_t v = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    unsigned long long acc = 0;
    for (; v < V; v += stride) {
        if (row_bm && !((row_bm[v >> 5] >> (v & 31)) & 1u)) {
            if (cnt) cnt[v] = 0;
            continue;
        }
        const int32_t b = begin[v], e = begin[v + 1];
        if (e - b > ROWCNT_LONG) {
            long_rows[atomicAdd(nlong, 1ULL)] = (int32_t) v;
            continue;
        }
        int32_t c = 0;
        for (int32_t i = b; i < e; i++) {
            const int32_t w = idx[i];
            const unsigned bit = (probe_bm[w >> 5] >> (w & 31)) & 1u;
            c += invert ? (int32_t) (bit ^ 1u) : (int32_t) bit;
        }
        if (cnt) cnt[v] = c;
        acc += (unsigned long long) c;
    }
    if (total) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        if ((threadIdx.x & 63) == 0 && acc) atomicAdd(total, acc);
    }
}

__global__ void __launch_bounds__(BFS_THREADS)
row_count_long_kernel(const int32_t* __restrict__ begin, const int32_t* __restrict__ idx, const int32_t* __restrict__ long_rows,
                      unsigned long long nlong, const uint32_t* __restrict__ probe_bm, int invert,
                      int32_t* __restrict__ cnt, unsigned long long* __restrict__ total) {
    const int lane = threadIdx.x & 63;
    unsigned long long wave = ((unsigned long long) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned long long nwaves = ((unsigned long long) gridDim.x * blockDim.x) >> 6;
    unsigned long long acc = 0;
    for (; wave < nlong; wave += nwaves) {
        const int32_t v = long_rows[wave];
        const int32_t b = begin[v], e = begin[v + 1];
        unsigned long long c = 0;
        for (int32_t i = b + lane; i < e; i += 64) {
            const int32_t w = idx[i];
            const unsigned bit = (probe_bm[w >> 5] >> (w & 31)) & 1u;
            c += invert ? (bit ^ 1u) : bit;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
        if (lane == 0) {
            if (cnt) cnt[v] = (int32_t) c;
            acc += c;
        }
    }
    if (total && lane == 0 && acc) atomicAdd(total, acc);
}

// The same count over the FLAT slot array (round 3): a workgroup takes ROWCNT_ITEMS consecutive items of the merged sequence
// (row ends, slots) -- merge-path over begin[], as in the top-down BFS level -- so that the slots are read coalesced, eight per
// thread and all in flight, whatever the rows look like; a slot finds its row by bisection in the LDS copy of the range's
// begin[] and adds its bit to the row's LDS counter.  With one row per lane (above) every lane walked its own list: 4-byte
// loads, one 64-byte request each -- RMAT-24: avg_teen_cnt 6.1 ms, conduct 5.0 ms for 1 GB of slots.
#define ROWCNT_ITEMS 2048
// rows consumed at the diagonals k * ROWCNT_ITEMS, k = 0 .. nb: one thread per diagonal.  (Searched by the workgroups
// themselves -- two threads, 24 dependent loads over begin[], everybody else at the barrier -- this was 19 of the ~27 us a
// workgroup took.)
__global__ void row_count_split_kernel(const int32_t* __restrict__ begin, int64_t V, int64_t E, int64_t nb, int64_t* __restrict__ split) {
    const int64_t k = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (k > nb) return;
    int64_t dk = k * ROWCNT_ITEMS;
    if (dk > V + E) dk = V + E;
    int64_t lo = dk > E ? dk - E : 0, hi = dk < V ? dk : V;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t) begin[mid + 1] <= dk - mid - 1) lo = mid + 1; else hi = mid;
    }
    split[k] = lo;
}

__global__ void __launch_bounds__(BFS_THREADS)
row_count_flat_kernel(const int32_t* __restrict__ begin, const int32_t* __restrict__ idx, int64_t V, int64_t E, const int64_t* __restrict__ split,
                      const uint32_t* __restrict__ row_bm, const uint32_t* __restrict__ probe_bm, int invert,
                      int32_t* __restrict__ cnt /* zeroed */, unsigned long long* __restrict__ total) {
    __shared__ int32_t s_off[ROWCNT_ITEMS + 2];
    __shared__ int32_t s_cnt[ROWCNT_ITEMS + 2];
    const int tid = threadIdx.x;
    // merge-path split of the diagonals k * ITEMS and (k + 1) * ITEMS: (rows consumed, slots consumed)
    int64_t d0 = (int64_t) blockIdx.x * ROWCNT_ITEMS, d1 = d0 + ROWCNT_ITEMS;
    if (d1 > V + E) d1 = V + E;
    const int64_t v0 = split[blockIdx.x], v1 = split[blockIdx.x + 1], e0 = d0 - v0, e1 = d1 - v1;
    const int nv = (int) (v1 - v0) + 1;   // rows touched: v0 .. v1 (the last one may be partial, or == V)
    for (int i = tid; i < nv; i += BFS_THREADS) {
        const int64_t vi = v0 + i;
        s_off[i] = vi <= V ? begin[vi < V ? vi : V] : (int32_t) E;
        // (the row's own bit, asked once per row, rides in the counter's sign: -1 = this row counts nothing)
        s_cnt[i] = row_bm && vi < V && !((row_bm[vi >> 5] >> (vi & 31)) & 1u) ? -1 : 0;
    }
    if (tid == 0) s_off[nv] = INT_MAX;   // sentinel
    __syncthreads();
    if (e1 > e0) {   // (workgroup-uniform)
        constexpr int K = ROWCNT_ITEMS / BFS_THREADS;
        const int lane = tid & 63;
        int32_t w[K], row[K];
        bool on[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int64_t x = e0 + tid + (int64_t) k * BFS_THREADS;
            on[k] = x < e1;
            const int64_t xc = on[k] ? x : e0;
            int lo = 0, hi = nv - 1;     // row of slot x: last i with s_off[i] <= x
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if ((int64_t) s_off[mid] <= xc) lo = mid; else hi = mid - 1;
            }
            row[k] = lo;
            on[k] = on[k] && s_cnt[lo] >= 0;   // (nobody adds to a row that counts nothing, so the sign stays)
        }
#pragma unroll
        for (int k = 0; k < K; k++) w[k] = on[k] ? idx[e0 + tid + (int64_t) k * BFS_THREADS] : 0;   // (unasked rows' slots are not read)
        uint32_t pw[K];
#pragma unroll
        for (int k = 0; k < K; k++) pw[k] = probe_bm[w[k] >> 5];
#pragma unroll
        for (int k = 0; k < K; k++) {
            // consecutive lanes hold consecutive slots: the lanes of one row are a run, and the run's first lane adds the
            // run's count -- one LDS add per (row, wave, pass) instead of one per slot on the same word
            const bool bit = on[k] && ((((pw[k] >> (w[k] & 31)) & 1u) != 0) != (invert != 0));
            const unsigned long long m = __ballot(bit);
            const int prev = __shfl_up(row[k], 1, 64);
            const unsigned long long heads = __ballot(lane == 0 || prev != row[k]);
            if (m && ((heads >> lane) & 1ull)) {
                const unsigned long long after = lane == 63 ? 0ull : heads >> (lane + 1);
                const int len = after ? __ffsll((long long) after) : 64 - lane;   // lanes of this run
                const unsigned long long mask = (len == 64 ? ~0ull : ((1ull << len) - 1ull)) << lane;
                const int c = __popcll(m & mask);
                if (c) atomicAdd(&s_cnt[row[k]], c);
            }
        }
    }
    __syncthreads();
    unsigned long long acc = 0;
    for (int i = tid; i < nv; i += BFS_THREADS) {
        const int32_t c = s_cnt[i];
        if (c <= 0) continue;
        acc += (unsigned long long) c;
        if (cnt) {
            const int64_t r = v0 + i;
            // a row whose slots all lie in this workgroup's range is written; the (at most two) rows cut by the range add
            const bool whole = (int64_t) s_off[i] >= e0 && i + 1 < nv && (int64_t) s_off[i + 1] <= e1;
            if (whole) cnt[r] = c; else atomicAdd(&cnt[r], c);
        }
    }
    if (total) {   // one add per workgroup, on one of 64 words (131 K workgroups adding to ONE word: ~90 adds per us, 3.7 ms)
        __shared__ unsigned long long s_red[BFS_THREADS / 64];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        if ((tid & 63) == 0) s_red[tid >> 6] = acc;
        __syncthreads();
        if (tid == 0) {
            unsigned long long t = 0;
            for (int i = 0; i < BFS_THREADS / 64; i++) t += s_red[i];
            if (t) atomicAdd(&total[blockIdx.x & 63], t);
        }
    }
}
__global__ void sum_shards_kernel(const unsigned long long* __restrict__ shard, unsigned long long* __restrict__ total) {
    unsigned long long t = shard[threadIdx.x];   // 64 threads
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
    if (threadIdx.x == 0 && t) atomicAdd(total, t);
}

// rows (all, or those whose bit is set in row_bm) count their neighbours by the probe bitmap; cnt (if given) is zeroed by the caller
static int count_by_bitmap(const int32_t* begin, const int32_t* idx, int64_t V, int64_t E, const unsigned long long* row_bm,
                           const unsigned long long* probe_bm, int invert, int32_t* cnt, unsigned long long* total) {
    if (getenv("GMX_ROWCNT_PER_ROW")) {   // development option: the one-row-per-lane form
        dbuf<int32_t> long_rows;
        dbuf<unsigned long long> nlong;
        GMX_CHECK(long_rows.alloc((size_t) V));
        GMX_CHECK(nlong.alloc(1));
        GMX_HIP(hipMemsetAsync(nlong.p, 0, sizeof(unsigned long long), 0));
        hipLaunchKernelGGL(row_count_kernel, dim3(grid_for(V, BFS_THREADS, 256 * 32)), dim3(BFS_THREADS), 0, 0, begin, idx, V,
                           (const uint32_t*) row_bm, (const uint32_t*) probe_bm, invert, cnt, long_rows.p, nlong.p, total);
        unsigned long long h = 0;
        GMX_HIP(hipMemcpy(&h, nlong.p, sizeof(h), hipMemcpyDeviceToHost));
        if (h) {
            int64_t wb = (int64_t) ((h * 64 + BFS_THREADS - 1) / BFS_THREADS);
            if (wb > 256 * 32) wb = 256 * 32;
            hipLaunchKernelGGL(row_count_long_kernel, dim3((unsigned) wb), dim3(BFS_THREADS), 0, 0, begin, idx, (const int32_t*) long_rows.p, h,
                               (const uint32_t*) probe_bm, invert, cnt, total);
        }
        GMX_HIP(hipGetLastError());
        return GMX_OK;
    }
    const int64_t nb = (V + E + ROWCNT_ITEMS - 1) / ROWCNT_ITEMS;
    if (nb > 0) {
        dbuf<int64_t> split;
        dbuf<unsigned long long> shard;
        GMX_CHECK(split.alloc((size_t) nb + 1));
        GMX_CHECK(shard.alloc(64));
        GMX_HIP(hipMemsetAsync(shard.p, 0, 64 * sizeof(unsigned long long), 0));
        hipLaunchKernelGGL(row_count_split_kernel, dim3((unsigned) ((nb + 1 + BFS_THREADS - 1) / BFS_THREADS)), dim3(BFS_THREADS), 0, 0, begin, V, E, nb, split.p);
        hipLaunchKernelGGL(row_count_flat_kernel, dim3((unsigned) nb), dim3(BFS_THREADS), 0, 0, begin, idx, V, E, (const int64_t*) split.p,
                           (const uint32_t*) row_bm, (const uint32_t*) probe_bm, invert, cnt, total ? shard.p : nullptr);
        if (total) hipLaunchKernelGGL(sum_shards_kernel, dim3(1), dim3(64), 0, 0, (const unsigned long long*) shard.p, total);
        GMX_HIP(hipDeviceSynchronize());   // (split and shard are released here)
    }
    GMX_HIP(hipGetLastError());
    return GMX_OK;
}

extern "C" int gmx_avg_teen_cnt(gmx_graph_t* g, const int32_t* age_host, int32_t K, int32_t* teen_cnt_host, float* avg,
                                gmx_stats_t* stats) {
    GMX_REQUIRE(g && avg && (teen_cnt_host || g->V == 0) && (age_host || g->V == 0), "NULL argument");
    if (stats) memset(stats, 0, sizeof(*stats));
    *avg = 0;
    const int64_t V = g->V;
    if (V == 0) return GMX_OK;
    dbuf<int32_t> age, cnt;
    dbuf<unsigned long long> acc;
    GMX_CHECK(age.alloc((size_t) V));
    GMX_CHECK(cnt.alloc((size_t) V));
    GMX_CHECK(acc.alloc(2));
    ev_guard evg[2];   // released on every return path
    hipEvent_t ev[2];
    for (int i = 0; i < 2; i++) {
        GMX_CHECK(evg[i].create());
        ev[i] = evg[i].e;
    }
    GMX_HIP(hipMemcpy(age.p, age_host, sizeof(int32_t) * (size_t) V, hipMemcpyHostToDevice));
    GMX_HIP(hipEventRecord(ev[0], 0));
    GMX_HIP(hipMemsetAsync(cnt.p, 0, sizeof(int32_t) * (size_t) V, 0));
    GMX_HIP(hipMemsetAsync(acc.p, 0, 2 * sizeof(unsigned long long), 0));
    // n.teen_cnt = Count(t: n.InNbrs)(t.age >= 10 && t.age < 20)
    if (g->has_reverse) {   // as written: every row walks its in-neighbours, the filter being a bitmap in the L2
        dbuf<unsigned long long> teen;
        GMX_CHECK(teen.alloc((size_t) ((V + 63) / 64)));
        hipLaunchKernelGGL(pred_bitmap_kernel, dim3(grid_for(V, BFS_THREADS, 256 * 16)), dim3(BFS_THREADS), 0, 0, (const int32_t*) age.p, V, 0, 0, teen.p);
        GMX_CHECK(count_by_bitmap(g->r_begin.p, g->r_node_idx.p, V, g->E, nullptr, teen.p, 0, cnt.p, nullptr));
        GMX_HIP(hipDeviceSynchronize());   // (teen is released at the end of this block)
    } else {                // forward CSR only: one increment per out-edge of a teen (integer atomics: same counts)
        GMX_CHECK(expand_selected<0>(g, age.p, 0, 0, cnt.p, nullptr));
    }
    // Avg(n: G.Nodes)(n.age > K){n.teen_cnt}: int32 sum, int64 count (gm_syntax_sugar2.cc:264-296)
    hipLaunchKernelGGL(filtered_sum_kernel, dim3(grid_for(V)), dim3(BFS_THREADS), 0, 0, (const int32_t*) age.p, (const int32_t*) cnt.p,
                       (const int32_t*) nullptr, V, 0, K, acc.p);
    GMX_HIP(hipEventRecord(ev[1], 0));
    unsigned long long h[2];
    GMX_HIP(hipMemcpy(h, acc.p, sizeof(h), hipMemcpyDeviceToHost));
    GMX_HIP(hipMemcpy(teen_cnt_host, cnt.p, sizeof(int32_t) * (size_t) V, hipMemcpyDeviceToHost));
    const int32_t S = (int32_t) (uint32_t) h[0];          // the emitted sum is an int32 and wraps like one
    const int64_t n = (int64_t) h[1];
    const double a = (0 == n) ? ((float) (0.000000)) : (S / ((double) n));
    *avg = (float) a;
    if (stats) {
        float ms = 0;
        (void) hipEventElapsedTime(&ms, ev[0], ev[1]);
        stats->iterations = 1;
        stats->kernel_ms = ms;
    }
    return GMX_OK;
}

extern "C" int gmx_conduct(gmx_graph_t* g, const int32_t* member_host, int32_t num, float* result, gmx_stats_t* stats) {
    GMX_REQUIRE(g && result && (member_host || g->V == 0), "NULL argument");
    if (stats) memset(stats, 0, sizeof(*stats));
    *result = 0;
    const int64_t V = g->V;
    unsigned long long h[6] = {0, 0, 0, 0, 0, 0};
    ev_guard evg[2];   // released on every return path
    hipEvent_t ev[2];
    for (int i = 0; i < 2; i++) {
        GMX_CHECK(evg[i].create());
        ev[i] = evg[i].e;
    }
    if (V > 0) {
        dbuf<int32_t> member;
        dbuf<unsigned long long> acc;   // [0,1] Din + count, [2,3] Dout + count, [4] Cross
        GMX_CHECK(member.alloc((size_t) V));
        GMX_CHECK(acc.alloc(6));
        GMX_HIP(hipMemcpy(member.p, member_host, sizeof(int32_t) * (size_t) V, hipMemcpyHostToDevice));
        GMX_HIP(hipEventRecord(ev[0], 0));
        GMX_HIP(hipMemsetAsync(acc.p, 0, 6 * sizeof(unsigned long long), 0));
        hipLaunchKernelGGL(filtered_sum_kernel, dim3(grid_for(V)), dim3(BFS_THREADS), 0, 0, (const int32_t*) member.p, (const int32_t*) nullptr,
                           (const int32_t*) g->begin.p, V, 1, num, acc.p);
        hipLaunchKernelGGL(filtered_sum_kernel, dim3(grid_for(V)), dim3(BFS_THREADS), 0, 0, (const int32_t*) member.p, (const int32_t*) nullptr,
                           (const int32_t*) g->begin.p, V, 2, num, acc.p + 2);
        {   // Cross: members count their out-neighbours that are not members (membership as a bitmap in the L2)
            dbuf<unsigned long long> mem_bm;
            GMX_CHECK(mem_bm.alloc((size_t) ((V + 63) / 64)));
            hipLaunchKernelGGL(pred_bitmap_kernel, dim3(grid_for(V, BFS_THREADS, 256 * 16)), dim3(BFS_THREADS), 0, 0, (const int32_t*) member.p, V, 1, num, mem_bm.p);
            GMX_CHECK(count_by_bitmap(g->begin.p, g->node_idx.p, V, g->E, mem_bm.p, mem_bm.p, 1, nullptr, acc.p + 4));
            GMX_HIP(hipDeviceSynchronize());
        }
        GMX_HIP(hipEventRecord(ev[1], 0));
        GMX_HIP(hipMemcpy(h, acc.p, sizeof(h), hipMemcpyDeviceToHost));
    }
    const int32_t Din = (int32_t) (uint32_t) h[0], Dout = (int32_t) (uint32_t) h[2], Cross = (int32_t) (uint32_t) h[4];
    const float m = (float) ((Din < Dout) ? Din : Dout);
    if (m == 0) *result = (Cross == 0) ? ((float) (0.000000)) : FLT_MAX;
    else *result = Cross / m;
    if (stats && V > 0) {
        float ms = 0;
        (void) hipEventElapsedTime(&ms, ev[0], ev[1]);
        stats->iterations = 1;
        stats->kernel_ms = ms;
    }
    return GMX_OK;
}

// Loads this translation unit's code object (the HIP runtime does that lazily, at the first launch of one of its kernels:
// tens of milliseconds that would otherwise fall into the first timed call) -- called once from the graph constructors.
void gmx_touch_bfs() {
    hipFuncAttributes attr;
    (void) hipFuncGetAttributes(&attr, (const void*) bfs_totals_kernel);
}
