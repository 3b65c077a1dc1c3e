// gmx_pagerank.hip -- the PageRank neighbour-reduction hot loop for gfx950.
//
// Replaces the body of the emitted `pagerank` (source /root/reference/apps/src/pagerank.gm:1-20,
// emission rules SURVEY.md section 8 a-1):
//     for t in nodes:  S = sum_{w in InNbrs(t)} rank[w] / outdeg(w)
//                      val = (1-d)/N + d*S;  diff += |val - rank[t]|;  rank_nxt[t] = val
// Device formulation (same arithmetic per term, fp64 row sums):
//   * contrib[w] = rank[w] / (double) outdeg(w) is produced once per vertex by the row that
//     owns w (the reference recomputes the same quotient once per edge);
//   * rows are processed by a MERGE-PATH decomposition of (row ends, edges): every
//     workgroup gets exactly ITEMS path items, so hubs and empty rows cost the same;
//   * a workgroup streams its slice of r_node_idx with coalesced non-temporal loads, gathers
//     contrib[] into LDS, reduces each row from LDS (thread per short row, wave per long
//     row, __shfl_down), and applies the rank update in place;
//   * rows that span workgroups leave fp64 partials that a small fix-up kernel adds in edge
//     order, so the result is run-to-run deterministic (no float atomics anywhere);
//   * vertices are internally renumbered by descending out-degree (the gather frequency of
//     contrib[w] IS outdeg(w)), which packs the hot part of the vector into few cache lines;
//     the hottest entries can additionally be served from LDS (GMX_PR_HOT_LDS).
// Roofline: HBM-bound; algorithmic bytes per iteration E*(4+s) + V*(8+3s).
#pragma clang fp contract(off)

#include "gmx_internal.h"

#include <math.h>
#include <rocprim/rocprim.hpp>

#define PR_LONG 96   // rows with more in-block edges than this are reduced by a whole wave

struct pr_blk { int32_t r, e; };  // merge-path start of a workgroup: local row index, local edge index

struct gmx_pr {
    gmx_graph* g = nullptr;
    int elem = 4;
    int rank = 0, nranks = 1;
    uint32_t options = 0;
    int64_t V = 0;        // vertices of the whole graph
    int64_t slice = 0;    // rows per rank (Vpad / nranks)
    int64_t Vpad = 0;     // slice * nranks, size of the contribution replica
    int64_t rows = 0;     // real rows owned by this rank (<= slice)
    int64_t row_lo = 0;   // first owned row in the internal numbering
    int64_t El = 0;       // edges of the owned rows
    dbuf<int32_t> inv;    // internal id -> original id for owned rows [rows]
    dbuf<int32_t> rb_own, ridx_own;
    const int32_t* rb = nullptr;    // local r_begin' [rows+1]
    const int32_t* ridx = nullptr;  // r_node_idx' (internal source ids) [El]
    dbuf<int32_t> outdeg; // [rows]
    dbuf<char> rk;        // rank of owned rows [rows] x elem
    dbuf<char> contrib[2];  // replicas [Vpad] x elem
    int cur = 0;          // contrib[cur] is read by the next step
    int64_t nblk = 0;
    int items = 0, threads = 0;
    bool hot = false;
    int persistent_grid = 0;
    dbuf<pr_blk> blk;     // [nblk+1]
    dbuf<double> part_first, part_last;   // [nblk] partial row sums leaving a workgroup
    dbuf<double> diff_part;               // [2*nblk] per-workgroup |val-rank| partials (main, fix-up)
    dbuf<double> diff;    // [1]
    double d = 0.85;
    int32_t cnt = 0;
};

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
__global__ void pr_perm_kernel(const int32_t* __restrict__ order, int64_t V, int64_t slice, int nranks,
                               int32_t* __restrict__ perm) {
    int64_t j = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; j < V; j += stride) {
        if (order) perm[order[j]] = (int32_t) ((j % nranks) * slice + j / nranks);
        else perm[j] = (int32_t) j;
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

// ------------------------------------------------------------------ hot loop
template <typename S>
__device__ __forceinline__ void pr_finalize(int64_t r, double sum, double base, double d,
                                            S* __restrict__ rk, const int32_t* __restrict__ outdeg,
                                            S* __restrict__ contrib_next_owned, double& diff_acc) {
    double val = base + d * sum;
    double old = (double) rk[r];
    S vs = (S) val;
    diff_acc += fabs((double) vs - old);
    rk[r] = vs;
    int32_t od = outdeg[r];
    contrib_next_owned[r] = od > 0 ? (S) ((double) vs / (double) od) : (S) 0;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// THREADS threads, ITEMS merge-path items per workgroup pass.  HOT > 0: contributions of the
// HOT hottest vertices (internal ids 0..HOT-1) are staged in LDS once per workgroup and the
// workgroup walks the block list persistently.  NT: non-temporal loads for the index stream.
template <typename S, int THREADS, int ITEMS, int HOT, bool NT>
__global__ void __launch_bounds__(THREADS)
pr_step_kernel(const pr_blk* __restrict__ blk, int64_t nblk, int64_t rows,
               const int32_t* __restrict__ rb, const int32_t* __restrict__ ridx,
               const int32_t* __restrict__ outdeg, S* __restrict__ rk,
               const S* __restrict__ contrib, S* __restrict__ contrib_next_owned,
               double base, double d,
               double* __restrict__ part_first, double* __restrict__ part_last, double* __restrict__ diff_part) {
    constexpr int PER = ITEMS / THREADS;
    constexpr int NW = THREADS / 64;
    __shared__ S s_val[ITEMS];
    __shared__ int32_t s_rb[ITEMS + 2];
    __shared__ int32_t s_long[ITEMS / PR_LONG + 2];
    __shared__ int32_t s_nlong;
    __shared__ double s_red[NW];
    __shared__ S s_hot[HOT > 0 ? HOT : 1];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    if (HOT > 0) {
        for (int i = tid; i < HOT; i += THREADS) s_hot[i] = contrib[i];
    }

    for (int64_t k = blockIdx.x; k < nblk; k += gridDim.x) {
        const pr_blk b0 = blk[k], b1 = blk[k + 1];
        const int r0 = b0.r, e0 = b0.e, r1 = b1.r, e1 = b1.e;
        const int ne = e1 - e0;
        const int nr = r1 - r0 + 1;  // rows touched; the last one (r1) does not finish here
        double diff_acc = 0.0;

        __syncthreads();  // previous pass done with LDS (and s_hot visible)
        if (tid == 0) s_nlong = 0;
        for (int i = tid; i < nr; i += THREADS) s_rb[i] = rb[r0 + i];   // r1 <= rows, rb has rows+1 entries

        // ---- gather: coalesced index stream, random contribution reads, staged in LDS ----
        {
            int32_t ix[PER];
            S vv[PER];
#pragma unroll
            for (int u = 0; u < PER; u++) {
                int j = tid + u * THREADS;
                ix[u] = -1;
                if (j < ne) ix[u] = NT ? __builtin_nontemporal_load(ridx + e0 + j) : ridx[e0 + j];
            }
#pragma unroll
            for (int u = 0; u < PER; u++) {
                vv[u] = (S) 0;
                if (ix[u] >= 0) {
                    if (HOT > 0 && ix[u] < HOT) vv[u] = s_hot[ix[u]];
                    else vv[u] = contrib[ix[u]];
                }
            }
#pragma unroll
            for (int u = 0; u < PER; u++) {
                int j = tid + u * THREADS;
                if (j < ne) s_val[j] = vv[u];
            }
        }
        __syncthreads();

        const bool first_started_here = (s_rb[0] >= e0);   // rb[r0] == e0
        // ---- short rows: one thread per row, sequential fp64 sum in edge order ----
        for (int i = tid; i < nr; i += THREADS) {
            int lo = s_rb[i] - e0;
            if (lo < 0) lo = 0;
            int hi = (i < nr - 1) ? s_rb[i + 1] - e0 : ne;
            if (hi - lo > PR_LONG) {
                int q = atomicAdd(&s_nlong, 1);
                s_long[q] = i;
                continue;
            }
            double sum = 0.0;
            for (int j = lo; j < hi; j++) sum += (double) s_val[j];
            const bool started = (i > 0) || first_started_here;
            const bool finished = (i < nr - 1);
            if (!started) part_first[k] = sum;
            else if (finished) pr_finalize<S>((int64_t) r0 + i, sum, base, d, rk, outdeg, contrib_next_owned, diff_acc);
            else if (r0 + i < rows && hi > lo) part_last[k] = sum;
        }
        __syncthreads();
        // ---- long rows: one wave per row, strided fp64 partials + shuffle reduction ----
        const int nlong = s_nlong;
        for (int q = wave; q < nlong; q += NW) {
            // the list was filled in arbitrary order; each entry is handled independently
            const int i = s_long[q];
            int lo = s_rb[i] - e0;
            if (lo < 0) lo = 0;
            const int hi = (i < nr - 1) ? s_rb[i + 1] - e0 : ne;
            double sum = 0.0;
            for (int j = lo + lane; j < hi; j += 64) sum += (double) s_val[j];
            sum = wave_sum(sum);
            if (lane == 0) {
                const bool started = (i > 0) || first_started_here;
                const bool finished = (i < nr - 1);
                if (!started) part_first[k] = sum;
                else if (finished) pr_finalize<S>((int64_t) r0 + i, sum, base, d, rk, outdeg, contrib_next_owned, diff_acc);
                else if (r0 + i < rows) part_last[k] = sum;
            }
        }
        // ---- |val - rank| partial of this workgroup pass ----
        diff_acc = wave_sum(diff_acc);
        if (lane == 0) s_red[wave] = diff_acc;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < NW; w++) t += s_red[w];
            diff_part[k] = t;
        }
    }
}

// Rows that span workgroups: the workgroup that OPENED row r (r == blk[k+1].r, rb[r] in
// [e0,e1)) left part_last[k]; every later workgroup touching r left part_first[k'].
template <typename S>
__global__ void pr_fixup_kernel(const pr_blk* __restrict__ blk, int64_t nblk, int64_t rows,
                                const int32_t* __restrict__ rb, const int32_t* __restrict__ outdeg,
                                S* __restrict__ rk, S* __restrict__ contrib_next_owned, double base, double d,
                                const double* __restrict__ part_first, const double* __restrict__ part_last,
                                double* __restrict__ diff_fix) {
    int64_t k = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nblk) return;
    double diff_acc = 0.0;
    const pr_blk b0 = blk[k], b1 = blk[k + 1];
    const int64_t r = b1.r;
    if (r < rows) {
        const int32_t rs = rb[r];
        if (rs >= b0.e && rs < b1.e) {
            double total = part_last[k];
            for (int64_t kk = k + 1; kk < nblk; kk++) {
                total += part_first[kk];
                if (blk[kk + 1].r > r) break;
            }
            pr_finalize<S>(r, total, base, d, rk, outdeg, contrib_next_owned, diff_acc);
        }
    }
    diff_fix[k] = diff_acc;
}

__global__ void pr_diff_reduce_kernel(const double* __restrict__ part, int64_t n, double* __restrict__ out) {
    __shared__ double s[1024 / 64];
    double t = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) t += part[i];
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
        S r0 = (S) (1 / N);                       // G.pg_rank = 1 / N
        rk[i] = r0;
        int32_t od = outdeg[i];
        contrib_owned[i] = od > 0 ? (S) ((double) r0 / (double) od) : (S) 0;
    }
}

template <typename S>
__global__ void pr_unpermute_kernel(int64_t rows, const int32_t* __restrict__ inv, const S* __restrict__ rk,
                                    S* __restrict__ out_orig) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < rows; i += stride) out_orig[inv[i]] = rk[i];
}

// ------------------------------------------------------------------ plan
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
    p->slice = (g->V + nranks - 1) / nranks;
    if (p->slice < 1) p->slice = 1;
    p->Vpad = p->slice * nranks;
    p->row_lo = (int64_t) rank * p->slice;
    const bool relabel = (options & GMX_PR_RELABEL) != 0;
    if (relabel) p->rows = g->V > rank ? (g->V - rank + nranks - 1) / nranks : 0;
    else {
        int64_t hi = (int64_t) (rank + 1) * p->slice;
        if (hi > g->V) hi = g->V;
        p->rows = hi > p->row_lo ? hi - p->row_lo : 0;
    }
    p->hot = (options & GMX_PR_HOT_LDS) != 0 && relabel && nranks == 1;
    p->threads = p->hot ? 1024 : 256;
    p->items = p->hot ? 4096 : 2048;

    const int64_t V = g->V, E = g->E;
    hipStream_t s = 0;
    int st = GMX_OK;
    do {
        if ((st = p->inv.alloc((size_t) p->rows)) || (st = p->outdeg.alloc((size_t) p->rows))) break;
        if (!relabel && nranks == 1) {
            // identity numbering, whole graph: use the graph's reverse CSR as is
            p->rb = g->r_begin.p;
            p->ridx = g->r_node_idx.p;
            p->El = E;
            dbuf<int32_t> perm;
            if ((st = perm.alloc((size_t) V))) break;
            hipLaunchKernelGGL(pr_perm_kernel, dim3(grid_for(V)), dim3(256), 0, s, (const int32_t*) nullptr, V, p->slice, nranks, perm.p);
            hipLaunchKernelGGL(pr_owned_kernel, dim3(grid_for(V)), dim3(256), 0, s, perm.p, g->begin.p, V, p->row_lo, p->rows, p->inv.p, p->outdeg.p);
            if (hipStreamSynchronize(s) != hipSuccess) { gmx_set_error("pr plan (identity) failed"); st = GMX_ERR_HIP; break; }
        } else {
            dbuf<int32_t> perm;
            if ((st = perm.alloc((size_t) V))) break;
            if (relabel) {
                dbuf<uint32_t> key, key2;
                dbuf<int32_t> id, order;
                if ((st = key.alloc((size_t) V)) || (st = key2.alloc((size_t) V)) || (st = id.alloc((size_t) V)) ||
                    (st = order.alloc((size_t) V))) break;
                hipLaunchKernelGGL(pr_degkey_kernel, dim3(grid_for(V)), dim3(256), 0, s, g->begin.p, V, key.p, id.p);
                size_t tb = 0;
                hipError_t he = rocprim::radix_sort_pairs(nullptr, tb, key.p, key2.p, id.p, order.p, (size_t) V, 0u, 32u, s);
                dbuf<char> tmp;
                if (he == hipSuccess && (st = tmp.alloc(tb))) break;
                if (he == hipSuccess) he = rocprim::radix_sort_pairs((void*) tmp.p, tb, key.p, key2.p, id.p, order.p, (size_t) V, 0u, 32u, s);
                if (he == hipSuccess) {
                    hipLaunchKernelGGL(pr_perm_kernel, dim3(grid_for(V)), dim3(256), 0, s, (const int32_t*) order.p, V, p->slice, nranks, perm.p);
                    he = hipStreamSynchronize(s);
                }
                if (he != hipSuccess) { gmx_set_error("pr plan: degree sort failed: %s", hipGetErrorString(he)); st = GMX_ERR_HIP; break; }
            } else {
                hipLaunchKernelGGL(pr_perm_kernel, dim3(grid_for(V)), dim3(256), 0, s, (const int32_t*) nullptr, V, p->slice, nranks, perm.p);
            }
            hipLaunchKernelGGL(pr_owned_kernel, dim3(grid_for(V)), dim3(256), 0, s, perm.p, g->begin.p, V, p->row_lo, p->rows, p->inv.p, p->outdeg.p);
            // keys (perm[dst] << 32 | perm[src]) from the reverse CSR, sorted; then cut the owned rows
            dbuf<uint64_t> keys, alt;
            if ((st = keys.alloc((size_t) E)) || (st = alt.alloc((size_t) E))) break;
            if ((st = gmx_keys_from_csr(g->r_begin.p, g->r_node_idx.p, V, E, false, perm.p, keys.p, s))) break;
            const uint64_t* sorted = keys.p;
            if (E > 1) {
                rocprim::double_buffer<uint64_t> db(keys.p, alt.p);
                size_t tb = 0;
                unsigned end_bit = 32 + (unsigned) gmx_bits_for(p->Vpad);
                hipError_t he = rocprim::radix_sort_keys(nullptr, tb, db, (size_t) E, 0u, end_bit, s);
                dbuf<char> tmp;
                if (he == hipSuccess && (st = tmp.alloc(tb))) break;
                if (he == hipSuccess) he = rocprim::radix_sort_keys((void*) tmp.p, tb, db, (size_t) E, 0u, end_bit, s);
                if (he == hipSuccess) he = hipStreamSynchronize(s);
                if (he != hipSuccess) { gmx_set_error("pr plan: key sort failed: %s", hipGetErrorString(he)); st = GMX_ERR_HIP; break; }
                sorted = db.current();
            }
            dbuf<int64_t> bounds;
            if ((st = bounds.alloc(2))) break;
            hipLaunchKernelGGL(pr_key_bound_kernel, dim3(1), dim3(64), 0, s, sorted, E,
                               (uint64_t) p->row_lo << 32, (uint64_t) (p->row_lo + p->rows) << 32, bounds.p);
            int64_t hb[2] = {0, 0};
            if (hipMemcpy(hb, bounds.p, sizeof(hb), hipMemcpyDeviceToHost) != hipSuccess) { gmx_set_error("pr plan: bounds copy failed"); st = GMX_ERR_HIP; break; }
            p->El = hb[1] - hb[0];
            if ((st = p->rb_own.alloc((size_t) p->rows + 1)) || (st = p->ridx_own.alloc((size_t) p->El))) break;
            hipLaunchKernelGGL(pr_local_csr_kernel, dim3(grid_for(p->El > p->rows ? p->El : p->rows + 1)), dim3(256), 0, s,
                               sorted, E, p->row_lo, p->rows, hb[0], p->El, p->rb_own.p, p->ridx_own.p);
            if (hipStreamSynchronize(s) != hipSuccess) { gmx_set_error("pr plan: local csr failed"); st = GMX_ERR_HIP; break; }
            p->rb = p->rb_own.p;
            p->ridx = p->ridx_own.p;
        }
        // merge-path blocks
        int64_t total = p->rows + p->El;
        p->nblk = (total + p->items - 1) / p->items;
        if ((st = p->blk.alloc((size_t) p->nblk + 1))) break;
        hipLaunchKernelGGL(pr_blocks_kernel, dim3(grid_for(p->nblk + 1, 256, 1 << 30)), dim3(256), 0, s,
                           p->rb, p->rows, p->El, p->items, p->nblk, p->blk.p);
        size_t nb = (size_t) (p->nblk ? p->nblk : 1);
        if ((st = p->part_first.alloc(nb)) || (st = p->part_last.alloc(nb)) || (st = p->diff_part.alloc(2 * nb)) ||
            (st = p->diff.alloc(1)) || (st = p->rk.alloc((size_t) (p->rows ? p->rows : 1) * elem_bytes)) ||
            (st = p->contrib[0].alloc((size_t) p->Vpad * elem_bytes)) || (st = p->contrib[1].alloc((size_t) p->Vpad * elem_bytes))) break;
        if (hipMemset(p->contrib[0].p, 0, (size_t) p->Vpad * elem_bytes) != hipSuccess ||
            hipMemset(p->contrib[1].p, 0, (size_t) p->Vpad * elem_bytes) != hipSuccess ||
            hipMemset(p->diff_part.p, 0, 2 * nb * sizeof(double)) != hipSuccess ||
            hipMemset(p->diff.p, 0, sizeof(double)) != hipSuccess ||
            hipDeviceSynchronize() != hipSuccess) { gmx_set_error("pr plan: memset failed"); st = GMX_ERR_HIP; break; }
        hipDeviceProp_t prop;
        int dev = 0;
        (void) hipGetDevice(&dev);
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { gmx_set_error("hipGetDeviceProperties failed"); st = GMX_ERR_HIP; break; }
        p->persistent_grid = prop.multiProcessorCount;
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
    p->d = d;
    p->cnt = 0;
    p->cur = 0;
    if (p->rows > 0) {
        if (p->elem == 4)
            hipLaunchKernelGGL(pr_reset_kernel<float>, dim3(grid_for(p->rows)), dim3(256), 0, 0, p->rows, (double) p->V,
                               p->outdeg.p, (float*) p->rk.p, (float*) p->contrib[0].p + p->row_lo);
        else
            hipLaunchKernelGGL(pr_reset_kernel<double>, dim3(grid_for(p->rows)), dim3(256), 0, 0, p->rows, (double) p->V,
                               p->outdeg.p, (double*) p->rk.p, (double*) p->contrib[0].p + p->row_lo);
    }
    GMX_HIP(hipGetLastError());
    GMX_HIP(hipDeviceSynchronize());
    return GMX_OK;
}

template <typename S, int THREADS, int ITEMS, int HOT>
static void launch_step(gmx_pr* p, hipStream_t s, int grid) {
    const double N = (double) p->V;
    const double base = (1 - p->d) / N;
    S* next_owned = (S*) p->contrib[1 - p->cur].p + p->row_lo;
    hipLaunchKernelGGL((pr_step_kernel<S, THREADS, ITEMS, HOT, true>), dim3(grid), dim3(THREADS), 0, s,
                       p->blk.p, p->nblk, p->rows, p->rb, p->ridx, p->outdeg.p, (S*) p->rk.p,
                       (const S*) p->contrib[p->cur].p, next_owned, base, p->d,
                       p->part_first.p, p->part_last.p, p->diff_part.p);
    hipLaunchKernelGGL(pr_fixup_kernel<S>, dim3((unsigned) ((p->nblk + 255) / 256)), dim3(256), 0, s,
                       p->blk.p, p->nblk, p->rows, p->rb, p->outdeg.p, (S*) p->rk.p, next_owned, base, p->d,
                       p->part_first.p, p->part_last.p, p->diff_part.p + p->nblk);
}

extern "C" int gmx_pr_step(gmx_pr_t* p, void* stream) {
    GMX_REQUIRE(p, "pr is NULL");
    hipStream_t s = (hipStream_t) stream;
    if (p->nblk > 0) {
        if (p->hot) {
            int grid = p->persistent_grid < p->nblk ? p->persistent_grid : (int) p->nblk;
            if (p->elem == 4) launch_step<float, 1024, 4096, 28672>(p, s, grid);
            else launch_step<double, 1024, 4096, 12288>(p, s, grid);
        } else {
            GMX_REQUIRE(p->nblk < (1LL << 31), "too many workgroups");
            if (p->elem == 4) launch_step<float, 256, 2048, 0>(p, s, (int) p->nblk);
            else launch_step<double, 256, 2048, 0>(p, s, (int) p->nblk);
        }
        hipLaunchKernelGGL(pr_diff_reduce_kernel, dim3(1), dim3(1024), 0, s, (const double*) p->diff_part.p, 2 * p->nblk, p->diff.p);
    }
    GMX_HIP(hipGetLastError());
    p->cur = 1 - p->cur;
    p->cnt++;
    return GMX_OK;
}

extern "C" int gmx_pr_contrib_slice(gmx_pr_t* p, void** dev_ptr, int64_t* count) {
    GMX_REQUIRE(p && dev_ptr && count, "NULL argument");
    *dev_ptr = p->contrib[p->cur].p + (size_t) p->row_lo * p->elem;
    *count = p->slice;
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
    GMX_HIP(hipMemcpyAsync(diff, p->diff.p, sizeof(double), hipMemcpyDeviceToHost, (hipStream_t) stream));
    GMX_HIP(hipStreamSynchronize((hipStream_t) stream));
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
    if (p->elem == 4)
        hipLaunchKernelGGL(pr_unpermute_kernel<float>, dim3(grid_for(p->rows)), dim3(256), 0, 0, p->rows, p->inv.p, (const float*) p->rk.p, (float*) tmp.p);
    else
        hipLaunchKernelGGL(pr_unpermute_kernel<double>, dim3(grid_for(p->rows)), dim3(256), 0, 0, p->rows, p->inv.p, (const double*) p->rk.p, (double*) tmp.p);
    GMX_HIP(hipGetLastError());
    GMX_HIP(hipMemcpy(rank_host, tmp.p, (size_t) p->V * p->elem, hipMemcpyDeviceToHost));
    return GMX_OK;
}

extern "C" int gmx_pr_work(gmx_pr_t* p, int64_t* edges, int64_t* rows, int64_t* algorithmic_bytes) {
    GMX_REQUIRE(p, "pr is NULL");
    if (edges) *edges = p->El;
    if (rows) *rows = p->rows;
    // E*(4+s) + V*(8+3s)
    if (algorithmic_bytes) *algorithmic_bytes = p->El * (4 + p->elem) + p->rows * (8 + 3 * (int64_t) p->elem);
    return GMX_OK;
}

// ------------------------------------------------------------------ whole-kernel entries
template <typename S>
static int pagerank_entry(gmx_graph_t* g, double e, double d, int32_t max_iter, S* rank_host, gmx_stats_t* stats) {
    GMX_REQUIRE(g && rank_host, "NULL argument");
    if (stats) memset(stats, 0, sizeof(*stats));
    if (g->V == 0) return GMX_OK;
    gmx_pr_t* p = nullptr;
    GMX_CHECK(gmx_pr_create(g, (int) sizeof(S), 0, 1, GMX_PR_RELABEL, &p));
    int st = gmx_pr_reset(p, d);
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
    gmx_pr_free(p);
    return st;
}

extern "C" int gmx_pagerank_f64(gmx_graph_t* g, double e, double d, int32_t max_iter, double* rank_host, gmx_stats_t* stats) {
    return pagerank_entry<double>(g, e, d, max_iter, rank_host, stats);
}

extern "C" int gmx_pagerank_f32(gmx_graph_t* g, float e, float d, int32_t max_iter, float* rank_host, gmx_stats_t* stats) {
    return pagerank_entry<float>(g, (double) e, (double) d, max_iter, rank_host, stats);
}
