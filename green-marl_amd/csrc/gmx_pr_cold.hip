// gmx_pr_cold.hip -- the binned part of the PageRank sweep: tile-gather / bin-accumulate.
//
// The pull sweep of gmx_pagerank.hip gathers contrib[w] per in-edge (emitted loop:
// /root/reference/apps/src/pagerank.gm:10-17, `Sum(w: t.InNbrs){w.pg_rank / w.OutDegree()}`) through a 32-bit
// index; a gather that misses the XCD's L2 costs a whole 128-byte line for 4 useful bytes, and one that hits is
// still an L2 request of its own.  The graph is static, so the in-edges whose source is past the first T ids of
// its rank range (T = 0: all of them) are laid out at plan time so that NO gather leaves the CU:
//
//   tile-major   the sources are cut into TILES whose contributions fit the LDS (128 KiB); a tile's edges are
//                ordered by destination BIN (BINROWS consecutive rows with in-edges -- "active" rows), then by
//                row, and stored as 16 bits each: 15 bits of tile-local source number + 1 bit "last edge of its
//                (tile, row) PAIR".  A (tile, bin) CELL is padded to whole groups of PRC_G edges.
//   bin-major    one item per pair: 16-bit bin-local row number + the value slot that phase 1 fills; a cell's items
//                are contiguous in both orders, a bin's cells follow each other.
//
//   phase 1  pr_cold_tile_kernel, persistent, one 1024-thread workgroup per CU, work items (tile, run of groups) from
//            one queue; an item's tile of contributions is copied into LDS (coalesced), then
//            * pair form  (tiles whose pairs average >= 1.25 edges -- the hot sources): every wave takes blocks of 512
//              entries, a lane 8 consecutive ones (one 16-byte load, prefetched by hand-counted inline-asm loads),
//              gathers from LDS, sums in fp64 along the lane, closes pairs that span lanes with a segmented wave scan
//              (DPP), stages the pair sums in LDS and stores them as runs of consecutive value slots.  Pairs are cut
//              at block ends at plan time (a longer pair simply becomes several items of the same row), so a block
//              needs nothing from its neighbours: no fix-up pass.
//            * edge form  (same kernel, same queue; tiles where almost every pair is a single edge -- the cold tail):
//              every edge is an item; 8 lanes copy a group with one 8-byte load, four LDS reads and one 16-byte store
//              each, so every store instruction writes whole aligned 128-byte lines.
//            The only metadata is one int per group of 32 edges (slot of the first pair that ends in the group).
//   phase 2  pr_cold_accum_kernel: one workgroup per bin (hub bins are split into chunks): streams values + 16-bit
//            row numbers (coalesced) and adds into 64-bit FIXED-POINT accumulators in LDS.  Integer adds commute,
//            so the result does not depend on the order the lanes arrive in: bit-reproducible without ordering
//            anything.  fp32 values: one limb, 2^-62 resolution (fp32 keeps 24 bits; the smallest contribution of
//            an RMAT-26 run is ~2^-34).  fp64 values: two limbs (2^-62 and 2^-(62+lo_bits)), > 100 bits below 1.0.
//            With every in-edge binned the workgroup then applies the PageRank update to its rows itself (FUSE).
//   phase 3  pr_cold_reduce_kernel / pr_cold_reduce_few_kernel add the chunk accumulators of the split bins.
//
// Output: the finished rows (fused form), or cold[i], the sum over the binned in-neighbours of active row i, element
// type S, which pr_combine_kernel adds after the per-slice sums of the pull sweep (fixed order).
// The step can be enqueued in parts (pr_cold_set_parts: phase 1 by tile class, phases 2-3 by bin range) for the
// pipelined multi-GPU step.  What bounds the kernels, and what was tried: DESIGN.md section 4.1.
// HBM bytes (fp32): 2.125 per edge + 10 per pair (4 written, 4 + 2 read), instead of 4 + a gather per edge.
#pragma clang fp contract(off)

#include "gmx_internal.h"

#include <algorithm>
#include <rocprim/rocprim.hpp>

#define PRC_G 32                 // edges / items per group (a cell is padded to whole groups)
#define PRC_BLK_GROUPS 16        // groups per wave block of the pair kernel (512 edges); tiles start on block boundaries
#define PRC_LDS_BYTES 131072     // accumulator array of phase 2
#define PRC_STAGE_BYTES 2048     // per wave: pair sums of a block on their way to coalesced stores
#define PRC_LDS_LIMIT 163840     // LDS of a CU
// elements of the contribution tile of phase 1: what the 16 stages (+ a dummy slot per lane) and a few words leave
static constexpr int prc_tile_elems(int elem) { return (PRC_LDS_LIMIT - 16 * (PRC_STAGE_BYTES + 64 * elem) - 64) / elem; }
// ... of a plan stored in length classes throughout: the only staging left is class H's 64 results per wave, so the tile
// takes the rest of the LDS, up to what a 15-bit source number reaches (fp32: 32768, fp64: 19448 -- a quarter fewer
// tiles, cells and pairs than with the generic form's tile)
static constexpr int prc_tile_elems_classed(int elem) {
    return (PRC_LDS_LIMIT - 16 * 64 * elem - 64) / elem < 32768 ? (PRC_LDS_LIMIT - 16 * 64 * elem - 64) / elem / 8 * 8 : 32768;
}
#define PRC_THREADS 1024
#define PRC_WAVES (PRC_THREADS / 64)
#define PRC_UNROLL 4             // pieces (8 groups = 256 items) a wave keeps in flight
#define PRC_PAD 0xffffu          // row number of a padding item
#define PRC_END 0x8000u          // "last edge of its pair" bit of a tile-major entry
#define PRC_Q1P 0                // work counters, 256 bytes apart
#define PRC_Q1E 64
#define PRC_Q2 128

// debug build (make debug): indices derived from the plan's tables are checked, a violation traps
#ifdef GMX_PR_BOUNDS
#define PRC_CHECK(cond) do { if (!(cond)) __builtin_trap(); } while (0)
#else
#define PRC_CHECK(cond) do { } while (0)
#endif

// Length classes (round 3).  The generic pair form spends ~230 VALU instructions on every 512-entry block whatever the
// block holds, and phase 1 was VALU-bound on it.  In the tiles that hold most of the edges the pairs
// are therefore sorted, at plan time, into CLASSES with a regular shape, each class a stream of its own ("virtual
// tile" = tile * 8 + class in the sort key), so that the kernel needs neither per-entry flags nor scans nor staging:
//   E1 / E2 / E4 / E8   pairs of 1 / 2 / 3-4 / 5-8 edges, padded to 1 / 2 / 4 / 8 entries: a lane (8 entries, one
//                       16-byte load) owns 8 / 4 / 2 / 1 whole pairs, sums them in fp64 in entry order and stores the
//                       results as one vector into consecutive item slots -- no cross-lane step at all;
//   H                   pairs of 9 or more edges, padded to whole lanes (multiples of 8 entries): every lane belongs to
//                       ONE pair, the only flag is "this lane closes its pair" (on the lane's last entry, which may be
//                       a padding entry), one segmented wave scan of the lane sums, at most 64 results per block,
//                       staged and stored with ONE store instruction per block;
//   class 0             the generic pair form / the edge form of round 2, kept for the tiles whose (tile, bin) cells are
//                       too small to be cut five ways (every (class, bin) sub-cell is padded to a group of 32 entries).
#define PRC_CLS_BITS 3
enum { PRC_CL_GEN = 0, PRC_CL_E1 = 1, PRC_CL_E2 = 2, PRC_CL_E4 = 3, PRC_CL_E8 = 4, PRC_CL_H = 5 };
enum { PRC_FORM_EDGE = 0, PRC_FORM_PAIR = 1, PRC_FORM_E1 = 2, PRC_FORM_E2 = 3, PRC_FORM_E4 = 4, PRC_FORM_E8 = 5, PRC_FORM_H = 6,
       PRC_FORM_TILE = 7 };   // TILE: a run of groups of a classed tile; the class streams it crosses come from the stream table
enum { PRC_TM_EDGE = 0, PRC_TM_PAIR = 1, PRC_TM_CLASSED = 2, PRC_TM_SINGLES = 3 };   // how a tile is stored (SINGLES: every edge an item, as class E1)
__host__ __device__ static inline int prc_class_of_len(int len) { return len <= 1 ? PRC_CL_E1 : len == 2 ? PRC_CL_E2 : len <= 4 ? PRC_CL_E4 : len <= 8 ? PRC_CL_E8 : PRC_CL_H; }
__host__ __device__ static inline int prc_padded_len(int cls, int len) {
    return cls == PRC_CL_GEN ? len : cls == PRC_CL_H ? (len + 7) & ~7 : 1 << (cls - 1);
}

// How many 16-byte stores a lane of class E_L issues for element size `elem`, and the interleave that goes with it.
// With ONE store per lane (fp32 E2 / E4 / E8, fp64 E4 / E8) consecutive lanes hold consecutive results and the store
// instruction writes whole lines.  With K > 1 stores (fp32 E1, fp64 E1 / E2) that layout makes every instruction write a
// 16-byte piece out of each lane's 32 or 64 bytes -- K partial writes per line, and the L2 takes a transaction for each
// (measured: the fp64 class kernel ran at half the HBM rate).  Those classes are therefore laid out in OCTETS at plan
// time (prc_interleaved_pos): 8 lanes own the results of two groups, and lane m's q-th store holds results
// [R m, R m + R) of the q-th 128-byte chunk of those two groups' results -- every instruction writes whole lines.
__host__ __device__ static inline int prc_e_stores(int L, int elem) { return ((8 / L) * elem + 15) / 16; }
// position of entry j of result i (of 32 / L) of absolute group G in the stream, for an interleaved class
__host__ __device__ static inline int64_t prc_interleaved_pos(int L, int elem, int64_t G, int i, int j) {
    const int K = prc_e_stores(L, elem), R = 16 / elem, cpg = K / 2;   // chunks (8 R results) per group
    const int ch = i / (8 * R), within = i % (8 * R), m = within / R, r = within % R, q = (int) (G & 1) * cpg + ch;
    return (G & ~(int64_t) 1) * PRC_G + 8 * m + (q * R + r) * L + j;
}

struct prc_item1 { int32_t tile, g0, g1, form; };         // phase 1: groups [g0, g1) of the tile-major stream of a (physical) tile; form PRC_FORM_*
struct prc_item2 { int32_t bin, g0, g1, slot; };          // phase 2: groups [g0, g1) of the bin-major stream; slot < 0: sole chunk
struct prc_item3 { int32_t bin, slot0, nslots, pad; };    // phase 3: a split bin

struct pr_cold {
    pr_cold_params prm;
    int64_t Ec = 0;          // edges handled here
    int64_t P1 = 0;          // tile-major entries (edges + padding)
    int64_t P2 = 0;          // bin-major items (pairs + padding)
    int64_t npairs = 0;
    int64_t ncold = 0;       // source positions
    int tile = 0, binrows = 0, limbs = 1, lo_bits = 42;
    int64_t ntiles = 0, nbins = 0, ncells = 0;
    dbuf<uint16_t> srcl;     // [P1] tile-major: source number | PRC_END
    dbuf<int32_t> ob;        // [P1 / G] item slot of the first pair that ends in the group
    dbuf<uint16_t> rowl;     // [P2] bin-major row numbers
    dbuf<char> val;          // [P2] x elem, bin-major: written by phase 1, read by phase 2
    dbuf<char> cold;         // [nactive] x elem
    dbuf<prc_item1> it1p;    // pair items, then edge items
    dbuf<int32_t> torg;      // [2 * ntiles] rank range and offset behind the hot/cold border where a tile starts
    dbuf<int32_t> vstart;    // [8 * ntiles + 1] first group of every (tile, class) stream
    dbuf<prc_item2> it2;
    dbuf<prc_item3> it3;
    int64_t n1p = 0, n1e = 0, n2 = 0, n3 = 0, nslots = 0;
    // The step can be cut into parts (pr_cold_set_parts): phase 1 by tile CLASS (0: every live source of the tile lies
    // in the hub piece of its rank range, 1: the rest), phases 2-3 by row part (bins), in processing order.
    std::vector<prc_item1> h1p, h1e;    // host copies of the work lists, in the order they were made (h1e: edge-form items)
    std::vector<prc_item2> h2;
    std::vector<prc_item3> h3;
    int nparts = 1;
    int64_t o1[3] = {0, 0, 0};          // it1p: items [o1[k], o1[k + 1]) belong to tile class k ...
    int64_t o1g[2] = {0, 0};            // ... the class-form items first, the generic pair / edge items from o1g[k]
    std::vector<int64_t> o2, o3, o3few; // it2 / it3: items [o[c], o[c + 1]) belong to part c; the first o3few[c] of a part's it3 items have few slots
    uint32_t q1 = 0, q1c = 0, q2 = 0;   // what the work counters of phase 1 (generic, class forms) / 2 hold before the next launch
    dbuf<unsigned long long> scratch;   // [nslots][limbs][binrows]
    dbuf<unsigned int> queue;           // [3 * 64]
    dbuf<double> diffp;                 // fused finish: [n2] partials of phase 2, then [n3 * binrows / 64] of phase 3
    int grid = 256;
    bool all_bins = false;   // every bin has items (the fused finish reaches every active row)
    std::vector<int32_t> tile_maxdeg;   // [ntiles] largest out-degree among a tile's sources (host; empty: unknown)
};

static int prc_grid_for(int64_t n, int block = 256) {
    int64_t b = (n + block - 1) / block;
    if (b < 1) b = 1;
    if (b > 256 * 16) b = 256 * 16;
    return (int) b;
}

// ------------------------------------------------------------------ plan kernels
// key' = tile | class | bin | bin-local row | tile-local source   (a tile holds tile_src = TILE - 1 sources: the last LDS
// slot stays zero and is what padding entries point at).  The class field is 0 until prc_class_kernel fills it.
__global__ void prc_keys_kernel(const uint64_t* __restrict__ keys, int64_t n, pr_cold_params prm, int tile_src, int binrows,
                                int binbits, uint64_t* __restrict__ out) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    const int64_t span = prm.slice - prm.T;
    for (; i < n; i += stride) {
        const uint64_t k = keys[i];
        const int64_t row = (int64_t) (k >> 32) - prm.row_lo;
        const int64_t src = (int64_t) (uint32_t) k;
        const int64_t a = prm.index_of_row[row];
        const int64_t cp = (src / prm.slice) * span + (src % prm.slice - prm.T);
        const uint64_t t = (uint64_t) (cp / tile_src), sl = (uint64_t) (cp % tile_src);
        const uint64_t b = (uint64_t) (a / binrows), rl = (uint64_t) (a % binrows);
        out[i] = (t << (32 + binbits + PRC_CLS_BITS)) | (b << 32) | (rl << 16) | sl;
    }
}

// cflag: first edge of a cell (same virtual tile and bin).  ps: first edge of a (virtual tile, row) pair.  n entries each.
// `single` (optional): the tile modes; in PRC_TM_SINGLES tiles every edge is a pair of its own (edge tiles stored as class E1).
__global__ void prc_runs_kernel(const uint64_t* __restrict__ k, int64_t n, const uint8_t* __restrict__ single, int tshift,
                                int32_t* __restrict__ cflag, int32_t* __restrict__ ps) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        cflag[i] = (i == 0 || (k[i] >> 32) != (k[i - 1] >> 32)) ? 1 : 0;
        ps[i] = (i == 0 || (k[i] >> 16) != (k[i - 1] >> 16) || (single && single[k[i] >> tshift] == PRC_TM_SINGLES)) ? 1 : 0;
    }
}

// pstart[p] = first edge of pair p; pstart[np] = n
__global__ void prc_pair_start_kernel(const int32_t* __restrict__ ps, const int32_t* __restrict__ pincl, int64_t n, int64_t np,
                                      int32_t* __restrict__ pstart) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i <= n; i += stride) {
        if (i == n) { pstart[np] = (int32_t) n; continue; }
        if (ps[i]) pstart[pincl[i] - 1] = (int32_t) i;
    }
}

// per tile t (0..ntiles): its first edge, and the pairs / cells before it (keys sorted, class field still 0)
__global__ void prc_tile_stats_kernel(const uint64_t* __restrict__ k, int64_t n, int shift, int64_t ntiles, const int32_t* __restrict__ pincl,
                                      const int32_t* __restrict__ cincl, int64_t np, int64_t nc, int32_t* __restrict__ out) {
    int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; t <= ntiles; t += stride) {
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t) (k[mid] >> shift) < t) lo = mid + 1; else hi = mid;
        }
        out[3 * t + 0] = (int32_t) lo;
        out[3 * t + 1] = lo < n ? pincl[lo] - 1 : (int32_t) np;
        out[3 * t + 2] = lo < n ? cincl[lo] - 1 : (int32_t) nc;
    }
}

// writes the class of every edge's pair into its key (tiles stored in classes only)
__global__ void prc_class_kernel(uint64_t* __restrict__ k, int64_t n, const int32_t* __restrict__ pincl, const int32_t* __restrict__ pstart,
                                 const uint8_t* __restrict__ mode, int binbits) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const uint64_t key = k[i];
        const int m = mode[key >> (32 + binbits + PRC_CLS_BITS)];
        if (m == PRC_TM_SINGLES) { k[i] = key | ((uint64_t) PRC_CL_E1 << (32 + binbits)); continue; }
        if (m != PRC_TM_CLASSED) continue;
        const int32_t p = pincl[i] - 1;
        k[i] = key | ((uint64_t) prc_class_of_len(pstart[p + 1] - pstart[p]) << (32 + binbits));
    }
}

// plen[p] = entries pair p occupies in the tile-major stream (plen[np] = 0)
__global__ void prc_pair_len_kernel(const uint64_t* __restrict__ k, const int32_t* __restrict__ pstart, int64_t np, int binbits,
                                    int32_t* __restrict__ plen) {
    int64_t p = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; p <= np; p += stride) {
        if (p == np) { plen[p] = 0; continue; }
        const int cls = (int) ((k[pstart[p]] >> (32 + binbits)) & ((1u << PRC_CLS_BITS) - 1));
        plen[p] = prc_padded_len(cls, pstart[p + 1] - pstart[p]);
    }
}

// first[c] = position of the first edge of cell c (first[ncells] = n); ckey[c] = tile | bin
__global__ void prc_cell_first_kernel(const uint64_t* __restrict__ k, const int32_t* __restrict__ cflag,
                                      const int32_t* __restrict__ incl, const int32_t* __restrict__ pincl, int64_t n, int64_t ncells, int64_t np,
                                      int32_t* __restrict__ first, int32_t* __restrict__ pfirst, uint32_t* __restrict__ ckey) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i <= n; i += stride) {
        if (i == n) { first[ncells] = (int32_t) n; pfirst[ncells] = (int32_t) np; continue; }
        if (cflag[i]) {
            first[incl[i] - 1] = (int32_t) i;
            pfirst[incl[i] - 1] = pincl[i] - 1;     // a cell starts with a pair
            ckey[incl[i] - 1] = (uint32_t) (k[i] >> 32);
        }
    }
}

// tile-major groups of a cell: its pairs' padded lengths, rounded up to whole groups (groups[ncells] = 0)
__global__ void prc_cell_entry_groups_kernel(const int32_t* __restrict__ pfirst, const int32_t* __restrict__ ppos, int64_t ncells,
                                             int32_t* __restrict__ groups) {
    int64_t c = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; c <= ncells; c += stride) groups[c] = c < ncells ? (ppos[pfirst[c + 1]] - ppos[pfirst[c]] + PRC_G - 1) / PRC_G : 0;
}

// groups[c] = ceil(count / G) where count = pre[first[c + 1]] - pre[first[c]] (pre == NULL: the edges themselves);
// groups[ncells] = 0.  Optionally the raw counts too.
__global__ void prc_cell_groups_kernel(const int32_t* __restrict__ first, const int32_t* __restrict__ pre, int64_t ncells,
                                       int32_t* __restrict__ groups, int32_t* __restrict__ counts) {
    int64_t c = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; c <= ncells; c += stride) {
        int32_t cnt = 0;
        if (c < ncells) cnt = pre ? pre[first[c + 1]] - pre[first[c]] : first[c + 1] - first[c];
        groups[c] = (cnt + PRC_G - 1) / PRC_G;
        if (counts) counts[c] = cnt;
    }
}

// per virtual tile v (0..nvt): index of its first cell and the raw group prefix there
__global__ void prc_vt_table_kernel(const uint32_t* __restrict__ ckey, int64_t ncells, int binbits, int64_t nvt,
                                    const int32_t* __restrict__ c1raw, int32_t* __restrict__ out) {
    int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; t <= nvt; t += stride) {
        int64_t lo = 0, hi = ncells;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t) (ckey[mid] >> binbits) < t) lo = mid + 1; else hi = mid;
        }
        out[2 * t + 0] = (int32_t) lo;
        out[2 * t + 1] = c1raw[lo];            // arrays have ncells + 1 entries
    }
}

// c1[c] = c1raw[c] + delta[tile(c)]: tiles start on block boundaries
__global__ void prc_cell_start_kernel(const int32_t* __restrict__ c1raw, const uint32_t* __restrict__ ckey, int binbits,
                                      const int32_t* __restrict__ delta, int64_t ncells, int32_t* __restrict__ c1) {
    int64_t c = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; c < ncells; c += stride) c1[c] = c1raw[c] + delta[ckey[c] >> binbits];
}

// Position of every edge in the padded tile-major stream, and whether it closes an ITEM: edge tiles make every edge an
// item; the generic form and class H cut their pairs at the ends of the 512-entry blocks of the stream (a block then
// needs nothing from its neighbours); the E classes have one item per pair.
__global__ void prc_pos_end_kernel(const uint64_t* __restrict__ k, const int32_t* __restrict__ cincl, const int32_t* __restrict__ pincl,
                                   const int32_t* __restrict__ pstart, const int32_t* __restrict__ ppos, const int32_t* __restrict__ pfirst,
                                   const int32_t* __restrict__ c1, const uint8_t* __restrict__ mode, int binbits, int elem, int64_t n,
                                   int32_t* __restrict__ pos, int32_t* __restrict__ endf) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i <= n; i += stride) {
        if (i == n) { endf[i] = 0; continue; }
        const int32_t c = cincl[i] - 1, p = pincl[i] - 1;
        const uint32_t vt = (uint32_t) (k[i] >> (32 + binbits));
        const int cls = (int) (vt & ((1u << PRC_CLS_BITS) - 1));
        int64_t at = (int64_t) c1[c] * PRC_G + (ppos[p] - ppos[pfirst[c]]) + (i - pstart[p]);
        if (cls >= PRC_CL_E1 && cls <= PRC_CL_E8 && prc_e_stores(1 << (cls - 1), elem) > 1) {   // octet layout (see prc_interleaved_pos)
            const int L = 1 << (cls - 1), rel = (int) (at & (PRC_G - 1));
            at = prc_interleaved_pos(L, elem, at / PRC_G, rel / L, rel % L);
        }
        pos[i] = (int32_t) at;
        const bool last = i + 1 == pstart[p + 1];
        const bool cut = (at & (PRC_BLK_GROUPS * PRC_G - 1)) == PRC_BLK_GROUPS * PRC_G - 1;
        int e;
        if (mode[vt >> PRC_CLS_BITS] == PRC_TM_EDGE) e = 1;
        else if (cls == PRC_CL_GEN || cls == PRC_CL_H) e = last || cut;
        else e = last;
        endf[i] = e;
    }
}

// key of the bin-major order of the cells
__global__ void prc_cell_key2_kernel(const uint32_t* __restrict__ ckey, int64_t ncells, int binbits, int tilebits,
                                     uint32_t* __restrict__ key2, int32_t* __restrict__ id) {
    int64_t c = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; c < ncells; c += stride) {
        const uint32_t t = ckey[c] >> binbits, b = ckey[c] & ((1u << binbits) - 1);
        key2[c] = (b << tilebits) | t;
        id[c] = (int32_t) c;
    }
}

__global__ void prc_gather_i32_kernel(const int32_t* __restrict__ src, const int32_t* __restrict__ idx, int64_t n, int32_t* __restrict__ dst) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = src[idx[i]];
}
__global__ void prc_scatter_i32_kernel(const int32_t* __restrict__ src, const int32_t* __restrict__ idx, int64_t n, int32_t* __restrict__ dst) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[idx[i]] = src[i];
}
__global__ void prc_fill_u16_kernel(uint16_t* __restrict__ p, int64_t n, uint16_t v) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

// edge i -> its entry in the tile-major stream; if it closes an item, the item's row number in the bin-major stream.
// Class H keeps its only flag on the LAST entry of a lane: when the item does not end there, the flag goes onto the
// padding entry that does (padding entries point at the tile's zero slot).
__global__ void prc_fill_items_kernel(const uint64_t* __restrict__ k, const int32_t* __restrict__ cincl,
                                      const int32_t* __restrict__ first, const int32_t* __restrict__ pos,
                                      const int32_t* __restrict__ c2, const int32_t* __restrict__ endf,
                                      const int32_t* __restrict__ endpre, int binbits, uint16_t zero_slot, int64_t n,
                                      uint16_t* __restrict__ srcl, uint16_t* __restrict__ rowl) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const int32_t c = cincl[i] - 1;
        const uint32_t lo = (uint32_t) k[i];
        const int cls = (int) ((k[i] >> (32 + binbits)) & ((1u << PRC_CLS_BITS) - 1));
        const int64_t at = pos[i];
        const bool e = endf[i] != 0;
        if (cls == PRC_CL_H) {
            if (e && (at & 7) != 7) {
                srcl[at] = (uint16_t) (lo & 0x7fffu);
                srcl[at | 7] = (uint16_t) (zero_slot | PRC_END);
            } else srcl[at] = (uint16_t) ((lo & 0x7fffu) | (e ? PRC_END : 0u));
        } else srcl[at] = (uint16_t) ((lo & 0x7fffu) | (e ? PRC_END : 0u));
        if (e) rowl[(int64_t) c2[c] * PRC_G + (endpre[i] - endpre[first[c]])] = (uint16_t) (lo >> 16);
    }
}

// tile-major group g -> item slot its first item goes to.  Generic / edge form: the first pair that ends in the group.
// E classes: the group's first pair (a group holds 32 / L whole pairs).  Class H: the first piece that ends at or
// behind the group's start.  Groups of a class stream that lie in the padding behind a cell or a stream get `pad_slot`
// (the E forms store every lane's results: padding must land outside the items); elsewhere 0, unused.
__global__ void prc_fill_ob_kernel(const int32_t* __restrict__ c1, const int32_t* __restrict__ c2, const int32_t* __restrict__ first,
                                   const int32_t* __restrict__ pfirst, const int32_t* __restrict__ pstart, const int32_t* __restrict__ ppos,
                                   const int32_t* __restrict__ endpre, const uint32_t* __restrict__ ckey, int binbits, int64_t ncells,
                                   int64_t ngroups, int32_t pad_slot, int32_t* __restrict__ ob) {
    int64_t g = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; g < ngroups; g += stride) {
        int64_t lo = 0, hi = ncells;   // last cell with c1 <= g
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t) c1[mid] <= g) lo = mid + 1; else hi = mid;
        }
        const int64_t c = lo - 1;
        int32_t v = 0;
        if (c >= 0) {
            const int cls = (int) ((ckey[c] >> binbits) & ((1u << PRC_CLS_BITS) - 1));
            const int32_t p0 = pfirst[c];
            const int64_t x = (g - c1[c]) * PRC_G, entries = ppos[pfirst[c + 1]] - ppos[p0];
            const int32_t slot0 = c2[c] * PRC_G;
            if (x >= entries) v = cls != PRC_CL_GEN ? pad_slot : 0;
            else if (cls == PRC_CL_GEN) v = slot0 + (endpre[first[c] + x] - endpre[first[c]]);   // no padding inside the cell: entry offset = edge offset
            else if (cls != PRC_CL_H) v = slot0 + (int32_t) ((g - c1[c]) * (PRC_G >> (cls - 1)));
            else {
                int64_t a = p0, b = pfirst[c + 1];   // the pair whose (padded) entries hold offset x: last one starting at or before x
                while (b - a > 1) {
                    const int64_t mid = (a + b) >> 1;
                    if ((int64_t) ppos[mid] - ppos[p0] <= x) a = mid; else b = mid;
                }
                const int64_t abs_pair = (int64_t) c1[c] * PRC_G + (ppos[a] - ppos[p0]), abs_g = (int64_t) c1[c] * PRC_G + x;
                const int64_t cuts = abs_g / (PRC_BLK_GROUPS * PRC_G) - abs_pair / (PRC_BLK_GROUPS * PRC_G);   // block ends inside the pair before the group
                v = slot0 + (endpre[pstart[a]] - endpre[first[c]]) + (int32_t) cuts;
            }
        }
        ob[g] = v;
    }
}

// table[b] = first item group of the first cell (bin-major order) whose bin is >= b
__global__ void prc_bin_table_kernel(const uint32_t* __restrict__ key2s, const int32_t* __restrict__ c2s, int64_t ncells,
                                     int tilebits, int64_t ngroups, int64_t nbins, int32_t* __restrict__ table) {
    int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; t <= nbins; t += stride) {
        int64_t lo = 0, hi = ncells;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t) (key2s[mid] >> tilebits) < t) lo = mid + 1; else hi = mid;
        }
        table[t] = lo < ncells ? c2s[lo] : (int32_t) ngroups;
    }
}

// largest out-degree among the sources of every tile (the limb guard's input)
__global__ void prc_tile_maxdeg_kernel(const int32_t* __restrict__ deg, int64_t nids, int64_t slice, int64_t T, int tile_src,
                                       int32_t* __restrict__ tmax) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    const int64_t span = slice - T;
    for (int64_t base = i - (threadIdx.x & 63); base < nids; base += stride) {   // (whole waves: the shuffles below need them)
        const int64_t id = base + (threadIdx.x & 63);
        int32_t dg = 0;
        int64_t t = -1;
        if (id < nids) {
            const int64_t l = id % slice;
            dg = deg[id];
            if (l >= T && dg > 0) t = ((id / slice) * span + (l - T)) / tile_src;
        }
        // 64 consecutive ids lie in one tile or two: the lanes that share the first lane's tile reduce among themselves
        const int64_t t0 = __shfl(t, 0, 64);
        int32_t m = t == t0 ? dg : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_down(m, off, 64));
        if ((threadIdx.x & 63) == 0 && t0 >= 0 && m > 0) atomicMax(&tmax[t0], m);
        if (t >= 0 && t != t0) atomicMax(&tmax[t], dg);
    }
}

// ------------------------------------------------------------------ hot loop
typedef unsigned int prc_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int prc_u32x4 __attribute__((ext_vector_type(4)));
typedef float prc_f32x4 __attribute__((ext_vector_type(4)));
typedef double prc_f64x4 __attribute__((ext_vector_type(4)));
template <typename S> struct prc_vec4;
template <> struct prc_vec4<float> { typedef prc_f32x4 type; };
template <> struct prc_vec4<double> { typedef prc_f64x4 type; };

// contribution tile `tile` (TILE - 1 sources) -> LDS; the last slot is the zero that padding entries read.  The
// tile's sources are consecutive ids inside a rank range and continue at the hot/cold border of the next range; the
// plan stores where each tile starts (org[2 * tile] = rank range, org[2 * tile + 1] = offset behind the border), so
// the copy is a few contiguous pieces with all loads of a piece in flight together.
template <typename S, int TILE>
__device__ __forceinline__ void prc_load_tile(S* __restrict__ s_tile, int tile, const int32_t* __restrict__ org,
                                              const S* __restrict__ contrib, int nranks, int64_t span, int64_t slice, int64_t T) {
    int r = org[2 * tile];
    int64_t l = org[2 * tile + 1];
    int i0 = 0;
    while (i0 < TILE - 1 && r < nranks) {   // (workgroup-uniform)
        const int64_t left = span - l;
        const int n = left < TILE - 1 - i0 ? (int) left : TILE - 1 - i0;
        const S* __restrict__ src = contrib + ((int64_t) r * slice + T + l);
#pragma unroll 16
        for (int j = threadIdx.x; j < n; j += PRC_THREADS) s_tile[i0 + j] = __builtin_nontemporal_load(src + j);
        i0 += n;
        r++;
        l = 0;
    }
    for (int j = i0 + threadIdx.x; j < TILE; j += PRC_THREADS) s_tile[j] = (S) 0;
}

// wave-level data movement on the VALU (DPP): no LDS traffic next to the tile gathers
#define PRC_DPP_ROW_SHR(n) (0x110 + (n))
#define PRC_DPP_WAVE_SHR1 0x138
#define PRC_DPP_BCAST15 0x142
#define PRC_DPP_BCAST31 0x143
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int prc_dpp_i(int v) {   // lanes without a source (or outside ROW_MASK) read 0
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, ROW_MASK == 0xf);   // all rows written: no need to preset the result
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double prc_dpp_d(double v) {
    const int lo = prc_dpp_i<CTRL, ROW_MASK>(__double2loint(v)), hi = prc_dpp_i<CTRL, ROW_MASK>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
// one step of the segmented inclusive scan: (v, f) <- (f ? v : v + v', f | f') with the (v', f') the DPP pattern delivers
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void prc_seg_step(double& v, int& f) {
    const double vo = prc_dpp_d<CTRL, ROW_MASK>(v);
    const int fo = prc_dpp_i<CTRL, ROW_MASK>(f);
    if (!f) v += vo;
    f |= fo;
}

// clears a double where mask is all ones (mask is 0 or -1): two bit operations, no compare
__device__ __forceinline__ double prc_clear_if(double x, int mask) {
    return __hiloint2double(__double2hiint(x) & ~mask, __double2loint(x) & ~mask);
}

// One block of the pair form: 8 entries of this lane in `cur`, `o` = slot of the first pair that ends in the lane's
// group.  Written for instruction count (see DESIGN.md 4.1 for what bounds the kernel): pair ends are 0 / -1 masks taken from the
// entries by bit-field extracts, nothing branches on them.
//   pass 1  the lane's open tail (sum behind its last pair end) -> segmented wave scan -> the open sum that enters
//           the lane (carry);
//   pass 2  the running sum starts at the carry; at every entry it is written to the wave's LDS stage -- to the
//           pair's ordinal if the entry ends a pair, to a dummy slot otherwise -- and cleared behind a pair end;
//   then the staged sums leave as runs of consecutive slots (a lane's own stores would each be an L2 request).
template <typename S>
__device__ __forceinline__ void prc_pair_block(const S* __restrict__ s_tile, S* __restrict__ stage, const prc_u32x4 cur,
                                               const int32_t o, S* __restrict__ val, const unsigned sink, const int lane) {
    constexpr int CAP = PRC_STAGE_BYTES / (int) sizeof(S);   // staged pair sums per pass (fp32: a whole block's)
    const unsigned w[4] = {cur.x, cur.y, cur.z, cur.w};
    S fv[8];
    int m[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
        fv[u] = s_tile[__builtin_amdgcn_ubfe(w[u >> 1], 16 * (u & 1), 15)];
        m[u] = __builtin_amdgcn_sbfe((int) w[u >> 1], 16 * (u & 1) + 15, 1);   // -1: last entry of its pair
    }
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < 8; u++) acc = prc_clear_if(acc + (double) fv[u], m[u]);
    const int many = (m[0] | m[1]) | (m[2] | m[3]) | ((m[4] | m[5]) | (m[6] | m[7]));
    const int cnt = -(((m[0] + m[1]) + (m[2] + m[3])) + ((m[4] + m[5]) + (m[6] + m[7])));
    // open sums across lanes: segmented inclusive scan (a lane holding a pair end restarts the segment); four steps
    // inside the rows of 16 lanes, then the row totals travel to the later rows
    double v = acc;
    int f = many & 1;
    prc_seg_step<PRC_DPP_ROW_SHR(1), 0xf>(v, f);
    prc_seg_step<PRC_DPP_ROW_SHR(2), 0xf>(v, f);
    prc_seg_step<PRC_DPP_ROW_SHR(4), 0xf>(v, f);
    prc_seg_step<PRC_DPP_ROW_SHR(8), 0xf>(v, f);
    prc_seg_step<PRC_DPP_BCAST15, 0xa>(v, f);
    prc_seg_step<PRC_DPP_BCAST31, 0xc>(v, f);
    const double carry = prc_dpp_d<PRC_DPP_WAVE_SHR1, 0xf>(v);   // lane 0 reads 0: pairs never cross a block start
    // ordinal of this lane's first pair inside the block (exclusive wave scan of the counts) ...
    int inc = cnt;
    inc += prc_dpp_i<PRC_DPP_ROW_SHR(1), 0xf>(inc);
    inc += prc_dpp_i<PRC_DPP_ROW_SHR(2), 0xf>(inc);
    inc += prc_dpp_i<PRC_DPP_ROW_SHR(4), 0xf>(inc);
    inc += prc_dpp_i<PRC_DPP_ROW_SHR(8), 0xf>(inc);
    inc += prc_dpp_i<PRC_DPP_BCAST15, 0xa>(inc);
    inc += prc_dpp_i<PRC_DPP_BCAST31, 0xc>(inc);
    const int total = __builtin_amdgcn_readlane(inc, 63);
    const int base = inc - cnt;
    // ... and of the group's first pair: slot = ordinal + shift, with one shift per group (equal for the groups of a cell)
    const int c1 = prc_dpp_i<PRC_DPP_ROW_SHR(1), 0xf>(cnt), c2 = prc_dpp_i<PRC_DPP_ROW_SHR(2), 0xf>(cnt),
              c3 = prc_dpp_i<PRC_DPP_ROW_SHR(3), 0xf>(cnt);
    const int q = lane & 3;
    const int gbase = base - ((q >= 1 ? c1 : 0) + (q >= 2 ? c2 : 0) + (q >= 3 ? c3 : 0));
    const int shift = o - gbase;
    int gcnt = cnt + __builtin_amdgcn_update_dpp(0, cnt, 0xb1, 0xf, 0xf, false);     // quad_perm [1,0,3,2]
    gcnt += __builtin_amdgcn_update_dpp(0, gcnt, 0x4e, 0xf, 0xf, false);              // quad_perm [2,3,0,1]: pairs ending in the group
    const unsigned long long live = __ballot(gcnt > 0);   // groups without a pair end (padding) have no say
    const int shift0 = __builtin_amdgcn_readlane(shift, live ? __builtin_ctzll(live) : 0);
    const bool one_run = __ballot(gcnt > 0 && shift != shift0) == 0ull;   // one cell, or cells that follow each other
    // Every block issues exactly 8 store instructions (lanes with nothing to store write to their sink slot), on
    // either path: the compiler can then count the stores between a prefetch and its use exactly, instead of
    // draining every outstanding load and store (s_waitcnt vmcnt(0)) in front of each block.
    constexpr int NPASS = PRC_BLK_GROUPS * PRC_G / CAP;   // fp32: 1, fp64: 2
    constexpr int ROUNDS = CAP / 64;                       // store rounds per pass: NPASS * ROUNDS == 8
#pragma unroll
    for (int p = 0; p < NPASS; p++) {
        const int p0 = p * CAP;
        int ord = base - p0;
        const int dummy = CAP + lane;
        S xs[8];
        int os[8];
        acc = carry;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            acc += (double) fv[u];
            xs[u] = (S) acc;
            os[u] = ord;
            int at = (ord & m[u]) | (dummy & ~m[u]);
            if (NPASS > 1) at = (unsigned) at < (unsigned) CAP ? at : dummy;
            PRC_CHECK(at >= 0 && at < CAP + 64);
            stage[at] = xs[u];
            ord -= m[u];
            acc = prc_clear_if(acc, m[u]);
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes have landed
        __builtin_amdgcn_wave_barrier();
        int p1 = total - p0;                    // staged ordinals [0, p1) = block ordinals [p0, p0 + p1)
        p1 = p1 < 0 ? 0 : (p1 > CAP ? CAP : p1);
        if (one_run) {
#pragma unroll
            for (int r = 0; r < ROUNDS; r++) {
                const int i = lane + 64 * r;
                const unsigned at = i < p1 ? (unsigned) (shift0 + p0 + i) : sink;
                PRC_CHECK(at <= sink);   // item slots lie below the sink slots
                val[at] = stage[i];
            }
        } else {
            // the block spans cells that are not adjacent in the bin-major order: every lane stores its own pair ends
#pragma unroll
            for (int u = p * ROUNDS; u < (p + 1) * ROUNDS; u++) {
                const unsigned at = m[u] ? (unsigned) (shift + p0 + os[u]) : sink;
                PRC_CHECK(at <= sink);
                val[at] = xs[u];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Phase 1, pair form.  Work item = (tile, groups [g0, g1)) with a whole number of super-steps: a wave walks its blocks
// PRC_PAIR_DEPTH at a time (block (k * DEPTH + j) * 16 + wave) and keeps the entries of the next PRC_PAIR_DEPTH in
// flight (64 KiB per CU) -- the stream comes from HBM, ~2 us away.
//
// The prefetch loads are inline asm with hand-counted waits.  hipcc's own bookkeeping turned every variant of this
// loop into "s_waitcnt vmcnt(0)" (or a register rotation) in front of each block, i.e. into draining the loads it had
// just issued together with every store of the previous blocks: the kernel then runs at the latency of one block
// per round trip.  VMEM operations retire in issue order and every block issues exactly 8 stores (prc_pair_block),
// so the count is exact: behind the loads of a set of blocks come the 8 * DEPTH stores of the set processed meanwhile
// and the 2 * DEPTH loads of the next set -- waiting for vmcnt <= 10 * DEPTH leaves exactly those in flight.  The two
// sets live in two named register groups (no rotation, so no copy of a register whose load is still in flight), and
// the wait statement names every destination "+v", so nothing reads them before it.
#define PRC_PAIR_DEPTH 4
#define PRC_SUPER_GROUPS (PRC_WAVES * PRC_PAIR_DEPTH * PRC_BLK_GROUPS)   // groups per super-step of a workgroup (1024)
#define PRC_STR2(x) #x
#define PRC_STR(x) PRC_STR2(x)
struct prc_set {   // the entries of PRC_PAIR_DEPTH blocks of one lane
    prc_u32x4 e0, e1, e2, e3;
    int32_t o0, o1, o2, o3;
};
__device__ __forceinline__ void prc_set_load(prc_set& t, const prc_u32x4* in, const int32_t* ob) {
    static_assert(PRC_PAIR_DEPTH == 4, "prc_set holds four blocks");
    asm volatile("global_load_dwordx4 %0, %8, off nt\n\t"
                 "global_load_dwordx4 %1, %9, off nt\n\t"
                 "global_load_dwordx4 %2, %10, off nt\n\t"
                 "global_load_dwordx4 %3, %11, off nt\n\t"
                 "global_load_dword %4, %12, off nt\n\t"
                 "global_load_dword %5, %13, off nt\n\t"
                 "global_load_dword %6, %14, off nt\n\t"
                 "global_load_dword %7, %15, off nt"
                 : "=&v"(t.e0), "=&v"(t.e1), "=&v"(t.e2), "=&v"(t.e3), "=&v"(t.o0), "=&v"(t.o1), "=&v"(t.o2), "=&v"(t.o3)
                 : "v"(in), "v"(in + PRC_WAVES * 64), "v"(in + 2 * PRC_WAVES * 64), "v"(in + 3 * PRC_WAVES * 64),
                   "v"(ob), "v"(ob + PRC_WAVES * PRC_BLK_GROUPS), "v"(ob + 2 * PRC_WAVES * PRC_BLK_GROUPS), "v"(ob + 3 * PRC_WAVES * PRC_BLK_GROUPS)
                 : "memory");
}
// N = VMEM operations issued after the set's loads that may still be in flight
#define PRC_SET_WAIT(t, N)                                                                                         \
    asm volatile("s_waitcnt vmcnt(" PRC_STR(N) ")"                                                                  \
                 : "+v"((t).e0), "+v"((t).e1), "+v"((t).e2), "+v"((t).e3), "+v"((t).o0), "+v"((t).o1), "+v"((t).o2), "+v"((t).o3) \
                 :: "memory")

// the pair form of one work item (see above); the tile is in LDS
template <typename S>
__device__ __forceinline__ void prc_pair_item(const prc_item1 d, const S* __restrict__ s_tile, S* __restrict__ stage,
                                              const uint16_t* __restrict__ srcl, const int32_t* __restrict__ ob,
                                              S* __restrict__ val, const unsigned sink, const int wv, const int lane) {
    const int nsuper = (d.g1 - d.g0) / PRC_SUPER_GROUPS;
    // 8 entries per lane; block b of the item starts 64 vectors (16 groups) after block b - 1
    const prc_u32x4* in = (const prc_u32x4*) srcl + ((int64_t) d.g0 * (PRC_G / 8) + wv * 64 + lane);
    const int32_t* obp = ob + (d.g0 + wv * PRC_BLK_GROUPS + (lane >> 2));
    constexpr int64_t IN_STEP = PRC_PAIR_DEPTH * PRC_WAVES * 64;
    constexpr int64_t OB_STEP = PRC_PAIR_DEPTH * PRC_WAVES * PRC_BLK_GROUPS;
    prc_set A, B;
#define PRC_PROCESS(t)                                                    \
    do {                                                                  \
        prc_pair_block<S>(s_tile, stage, (t).e0, (t).o0, val, sink, lane); \
        prc_pair_block<S>(s_tile, stage, (t).e1, (t).o1, val, sink, lane); \
        prc_pair_block<S>(s_tile, stage, (t).e2, (t).o2, val, sink, lane); \
        prc_pair_block<S>(s_tile, stage, (t).e3, (t).o3, val, sink, lane); \
    } while (0)
    // super-step 0 (set A): only the loads of super-step 1 are younger
    prc_set_load(A, in, obp);
    {
        const int kn = 1 < nsuper ? 1 : 0;
        prc_set_load(B, in + kn * IN_STEP, obp + kn * OB_STEP);
    }
    PRC_SET_WAIT(A, 8);
    PRC_PROCESS(A);
    int k = 1;
#pragma unroll 1
    for (; k + 1 < nsuper; k += 2) {
        prc_set_load(A, in + (int64_t) (k + 1) * IN_STEP, obp + (int64_t) (k + 1) * OB_STEP);
        PRC_SET_WAIT(B, 40);   // 32 stores of the previous super-step + the 8 loads just issued
        PRC_PROCESS(B);
        const int kn = k + 2 < nsuper ? k + 2 : k + 1;
        prc_set_load(B, in + (int64_t) kn * IN_STEP, obp + (int64_t) kn * OB_STEP);
        PRC_SET_WAIT(A, 40);
        PRC_PROCESS(A);
    }
    // B holds the last super-step (k < nsuper) or a redundant copy of it; either way exactly the 32 stores of the set
    // processed last are younger than its loads.  ONE wait in front of the branch: with a wait per arm the register
    // allocator copied B's registers in front of the arm's wait, i.e. read registers whose load was in flight (harmless
    // there -- the arm never used them -- but exactly what tools/isa_check.py exists to refuse).
    PRC_SET_WAIT(B, 32);
    if (k < nsuper) PRC_PROCESS(B);
#undef PRC_PROCESS
}

// ---- the class forms (see the top of the file) ----
typedef float prc_f32x2 __attribute__((ext_vector_type(2)));
typedef double prc_f64x2 __attribute__((ext_vector_type(2)));

// store instructions one block of a form issues (exactly: the prefetch waits count them)
template <typename S, int FORM> struct prc_form_stores {
    static constexpr int NP = FORM == PRC_FORM_H ? 1 : 8 >> (FORM - PRC_FORM_E1);     // results per lane
    static constexpr int value = (NP * (int) sizeof(S) + 15) / 16;
};

// NP consecutive results of a lane -> val[slot ..], in 16-byte pieces (slot is a multiple of NP by construction)
template <typename S, int NP> __device__ __forceinline__ void prc_store_results(S* __restrict__ dst, const S (&out)[NP]);
template <> __device__ __forceinline__ void prc_store_results<float, 1>(float* __restrict__ dst, const float (&out)[1]) { __builtin_nontemporal_store(out[0], dst); }
template <> __device__ __forceinline__ void prc_store_results<float, 2>(float* __restrict__ dst, const float (&out)[2]) {
    prc_f32x2 v; v.x = out[0]; v.y = out[1];
    __builtin_nontemporal_store(v, (prc_f32x2*) dst);
}
template <> __device__ __forceinline__ void prc_store_results<float, 4>(float* __restrict__ dst, const float (&out)[4]) {
    prc_f32x4 v; v.x = out[0]; v.y = out[1]; v.z = out[2]; v.w = out[3];
    __builtin_nontemporal_store(v, (prc_f32x4*) dst);
}
template <> __device__ __forceinline__ void prc_store_results<float, 8>(float* __restrict__ dst, const float (&out)[8]) {
    prc_f32x4 a, b; a.x = out[0]; a.y = out[1]; a.z = out[2]; a.w = out[3]; b.x = out[4]; b.y = out[5]; b.z = out[6]; b.w = out[7];
    __builtin_nontemporal_store(a, (prc_f32x4*) dst);
    __builtin_nontemporal_store(b, (prc_f32x4*) dst + 1);
}
template <> __device__ __forceinline__ void prc_store_results<double, 1>(double* __restrict__ dst, const double (&out)[1]) { __builtin_nontemporal_store(out[0], dst); }
template <> __device__ __forceinline__ void prc_store_results<double, 2>(double* __restrict__ dst, const double (&out)[2]) {
    prc_f64x2 v; v.x = out[0]; v.y = out[1];
    __builtin_nontemporal_store(v, (prc_f64x2*) dst);
}
template <> __device__ __forceinline__ void prc_store_results<double, 4>(double* __restrict__ dst, const double (&out)[4]) {
#pragma unroll
    for (int q = 0; q < 2; q++) { prc_f64x2 v; v.x = out[2 * q]; v.y = out[2 * q + 1]; __builtin_nontemporal_store(v, (prc_f64x2*) dst + q); }
}
template <> __device__ __forceinline__ void prc_store_results<double, 8>(double* __restrict__ dst, const double (&out)[8]) {
#pragma unroll
    for (int q = 0; q < 4; q++) { prc_f64x2 v; v.x = out[2 * q]; v.y = out[2 * q + 1]; __builtin_nontemporal_store(v, (prc_f64x2*) dst + q); }
}

// One block of an E class: the lane's 8 entries are 8 / L whole pairs of L entries (padding entries read the tile's zero
// slot); `o` = item slot of the first pair of the lane's group.  fp64 sums in entry order, one rounding to S per pair.
template <typename S, int L>
__device__ __forceinline__ void prc_block_E(const S* __restrict__ s_tile, const prc_u32x4 cur, const int32_t o, S* __restrict__ val, const int lane) {
    constexpr int NP = 8 / L;
    constexpr int K = (NP * (int) sizeof(S) + 15) / 16;
    const unsigned w[4] = {cur.x, cur.y, cur.z, cur.w};
    S fv[8];
#pragma unroll
    for (int u = 0; u < 8; u++) fv[u] = s_tile[__builtin_amdgcn_ubfe(w[u >> 1], 16 * (u & 1), 15)];
    S out[NP];
#pragma unroll
    for (int q = 0; q < NP; q++) {
        if (L == 1) out[q] = fv[q];
        else {
            double acc = (double) fv[q * L];
#pragma unroll
            for (int j = 1; j < L; j++) acc += (double) fv[q * L + j];
            out[q] = (S) acc;
        }
    }
    if constexpr (K == 1) {
        PRC_CHECK(o >= 0 && (o & (NP - 1)) == 0);
        prc_store_results<S, NP>(val + (o + (lane & 3) * NP), out);
    } else {
        // octet layout: stores 0 .. K/2-1 go to the even group of the lane's octet, the others to the odd one
        constexpr int R = 16 / (int) sizeof(S), CPG = K / 2;
        const int o_lo = __builtin_amdgcn_update_dpp(0, o, 0x114, 0xf, 0xf, true);    // row_shr:4  (from lane - 4)
        const int o_hi = __builtin_amdgcn_update_dpp(0, o, 0x104, 0xf, 0xf, true);    // row_shl:4  (from lane + 4)
        const bool upper = (lane & 4) != 0;
        const int o_even = upper ? o_lo : o, o_odd = upper ? o : o_hi;
        const int m = lane & 7;
#pragma unroll
        for (int q = 0; q < K; q++) {
            S piece[R];
#pragma unroll
            for (int r = 0; r < R; r++) piece[r] = out[q * R + r];
            const int at = (q / CPG ? o_odd : o_even) + (q % CPG) * 8 * R + R * m;
            PRC_CHECK(at >= 0 && (at & (R - 1)) == 0);
            prc_store_results<S, R>(val + at, piece);
        }
    }
}

// One block of class H: every lane belongs to one pair; bit 15 of the lane's LAST entry = "the pair (or its piece in
// this block) ends with this lane".  Lane sums in fp64, segmented inclusive scan over the lanes (a lane behind a
// closing lane starts a new segment), the closing lanes stage their result at the ordinal of their piece, one store.
template <typename S>
__device__ __forceinline__ void prc_block_H(const S* __restrict__ s_tile, S* __restrict__ stage, const prc_u32x4 cur, const int32_t o,
                                            S* __restrict__ val, const unsigned sink, const int lane) {
    const unsigned w[4] = {cur.x, cur.y, cur.z, cur.w};
    S fv[8];
#pragma unroll
    for (int u = 0; u < 8; u++) fv[u] = s_tile[__builtin_amdgcn_ubfe(w[u >> 1], 16 * (u & 1), 15)];
    double v = (double) fv[0];
#pragma unroll
    for (int u = 1; u < 8; u++) v += (double) fv[u];
    const int endm = __builtin_amdgcn_sbfe((int) w[3], 31, 1);                 // -1: closes its piece
    int f = prc_dpp_i<PRC_DPP_WAVE_SHR1, 0xf>(endm) & 1;                        // the lane before closed one: a segment starts here
    prc_seg_step<PRC_DPP_ROW_SHR(1), 0xf>(v, f);
    prc_seg_step<PRC_DPP_ROW_SHR(2), 0xf>(v, f);
    prc_seg_step<PRC_DPP_ROW_SHR(4), 0xf>(v, f);
    prc_seg_step<PRC_DPP_ROW_SHR(8), 0xf>(v, f);
    prc_seg_step<PRC_DPP_BCAST15, 0xa>(v, f);
    prc_seg_step<PRC_DPP_BCAST31, 0xc>(v, f);
    const unsigned long long ends = __ballot(endm != 0);
    const int total = __builtin_popcountll(ends);
    const int ord = (int) __builtin_amdgcn_mbcnt_hi((unsigned) (ends >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) ends, 0u));   // pieces closed by earlier lanes
    // slot = ordinal + shift with one shift per group: o is the slot of the first piece that closes in the lane's group
    const unsigned g4 = (unsigned) (ends >> (lane & ~3)) & 0xfu;                // the group's closing lanes
    const int gbase = ord - __builtin_popcount(g4 & ((1u << (lane & 3)) - 1u));
    const int shift = o - gbase;
    const unsigned long long live = __ballot(g4 != 0u);
    const int shift0 = __builtin_amdgcn_readlane(shift, live ? __builtin_ctzll(live) : 0);
    const bool one_run = __ballot(g4 != 0u && shift != shift0) == 0ull;
    const S x = (S) v;
    unsigned at;
    S y;
    if (one_run) {
        if (endm) stage[ord] = x;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes have landed
        __builtin_amdgcn_wave_barrier();
        y = stage[lane];
        at = lane < total ? (unsigned) (shift0 + lane) : sink;
        __builtin_amdgcn_wave_barrier();
    } else {   // the block spans cells that are not adjacent in the bin-major order: every closing lane stores its own
        y = x;
        at = endm ? (unsigned) (shift + ord) : sink;
    }
    PRC_CHECK(at <= sink);
    val[at] = y;   // exactly one store instruction per block on either path
}

template <typename S, int FORM>
__device__ __forceinline__ void prc_class_block(const S* __restrict__ s_tile, S* __restrict__ stage, const prc_u32x4 cur, const int32_t o,
                                                S* __restrict__ val, const unsigned sink, const int lane) {
    if constexpr (FORM == PRC_FORM_H) prc_block_H<S>(s_tile, stage, cur, o, val, sink, lane);
    else prc_block_E<S, 1 << (FORM - PRC_FORM_E1)>(s_tile, cur, o, val, lane);
}

#define PRC_SET_WAIT_N(t, N)                                                                                        \
    asm volatile("s_waitcnt vmcnt(%8)"                                                                              \
                 : "+v"((t).e0), "+v"((t).e1), "+v"((t).e2), "+v"((t).e3), "+v"((t).o0), "+v"((t).o1), "+v"((t).o2), "+v"((t).o3) \
                 : "n"(N) : "memory")

// One work item of a class form: blocks [0, nblocks) of 16 groups.  Whole super-steps (64 blocks: 4 per wave) run through
// the same two-set prefetch as the generic form, with a block's exact store count in the waits; the blocks left over
// (fewer than 64, the end of a class stream) are loaded plainly.
template <typename S, int FORM>
__device__ __forceinline__ void prc_class_item(const prc_item1 d, const S* __restrict__ s_tile, S* __restrict__ stage,
                                               const uint16_t* __restrict__ srcl, const int32_t* __restrict__ ob,
                                               S* __restrict__ val, const unsigned sink, const int wv, const int lane) {
    constexpr int NST = prc_form_stores<S, FORM>::value;
    const int nblocks = (d.g1 - d.g0) / PRC_BLK_GROUPS;
    const int nsuper = nblocks / (PRC_WAVES * PRC_PAIR_DEPTH);
    const prc_u32x4* in = (const prc_u32x4*) srcl + ((int64_t) d.g0 * (PRC_G / 8) + wv * 64 + lane);
    const int32_t* obp = ob + (d.g0 + wv * PRC_BLK_GROUPS + (lane >> 2));
    constexpr int64_t IN_STEP = PRC_PAIR_DEPTH * PRC_WAVES * 64;
    constexpr int64_t OB_STEP = PRC_PAIR_DEPTH * PRC_WAVES * PRC_BLK_GROUPS;
    if (nsuper > 0) {
        prc_set A, B;
#define PRC_PROCESS(t)                                                                  \
    do {                                                                                \
        prc_class_block<S, FORM>(s_tile, stage, (t).e0, (t).o0, val, sink, lane);       \
        prc_class_block<S, FORM>(s_tile, stage, (t).e1, (t).o1, val, sink, lane);       \
        prc_class_block<S, FORM>(s_tile, stage, (t).e2, (t).o2, val, sink, lane);       \
        prc_class_block<S, FORM>(s_tile, stage, (t).e3, (t).o3, val, sink, lane);       \
    } while (0)
        prc_set_load(A, in, obp);
        {
            const int kn = 1 < nsuper ? 1 : 0;
            prc_set_load(B, in + kn * IN_STEP, obp + kn * OB_STEP);
        }
        PRC_SET_WAIT_N(A, 8);
        PRC_PROCESS(A);
        int k = 1;
#pragma unroll 1
        for (; k + 1 < nsuper; k += 2) {
            prc_set_load(A, in + (int64_t) (k + 1) * IN_STEP, obp + (int64_t) (k + 1) * OB_STEP);
            PRC_SET_WAIT_N(B, 4 * NST + 8);   // the stores of the previous super-step + the 8 loads just issued
            PRC_PROCESS(B);
            const int kn = k + 2 < nsuper ? k + 2 : k + 1;
            prc_set_load(B, in + (int64_t) kn * IN_STEP, obp + (int64_t) kn * OB_STEP);
            PRC_SET_WAIT_N(A, 4 * NST + 8);
            PRC_PROCESS(A);
        }
        PRC_SET_WAIT_N(B, 4 * NST);            // B: the last super-step, or a redundant copy of it (see prc_pair_item)
        if (k < nsuper) PRC_PROCESS(B);
#undef PRC_PROCESS
    }
    {   // the blocks left over (fewer than a super-step): the wave's up to four loads go out together
        const int b0 = nsuper * (PRC_WAVES * PRC_PAIR_DEPTH);
        prc_u32x4 cur[PRC_PAIR_DEPTH];
        int32_t o[PRC_PAIR_DEPTH];
#pragma unroll
        for (int j = 0; j < PRC_PAIR_DEPTH; j++)
            if (b0 + j * PRC_WAVES + wv < nblocks) {   // (wave-uniform)
                const int64_t off = (int64_t) (b0 + j * PRC_WAVES);
                cur[j] = __builtin_nontemporal_load(in + off * 64);
                o[j] = __builtin_nontemporal_load(obp + off * PRC_BLK_GROUPS);
            }
#pragma unroll
        for (int j = 0; j < PRC_PAIR_DEPTH; j++)
            if (b0 + j * PRC_WAVES + wv < nblocks) prc_class_block<S, FORM>(s_tile, stage, cur[j], o[j], val, sink, lane);
    }
}

// The edge form of one work item: every entry is an item.  A lane handles 4 consecutive entries of one group: one
// 8-byte load, four LDS reads, one 16/32-byte store; 8 lanes cover a group, a wave 8 groups ("piece") per instruction.
template <typename S>
__device__ __forceinline__ void prc_edge_item(const prc_item1 d, const S* __restrict__ s_tile,
                                              const uint16_t* __restrict__ srcl, const int32_t* __restrict__ ob,
                                              S* __restrict__ val, const int wv, const int lane) {
    typedef typename prc_vec4<S>::type V4;
    const int npieces = (d.g1 - d.g0 + 7) >> 3;
    for (int base = 0; base < npieces; base += PRC_WAVES * PRC_UNROLL) {
        prc_u32x2 ids[PRC_UNROLL];
        int32_t o[PRC_UNROLL];
#pragma unroll
        for (int u = 0; u < PRC_UNROLL; u++) {
            const int pc = base + u * PRC_WAVES + wv;
            const int g = d.g0 + pc * 8 + (lane >> 3);
            o[u] = -1;
            if (pc < npieces && g < d.g1) {
                ids[u] = __builtin_nontemporal_load((const prc_u32x2*) srcl + (int64_t) g * (PRC_G / 4) + (lane & 7));
                o[u] = __builtin_nontemporal_load(ob + g);
            }
        }
#pragma unroll
        for (int u = 0; u < PRC_UNROLL; u++) {
            if (o[u] < 0) continue;
            V4 v;
            v.x = s_tile[ids[u].x & 0x7fffu];
            v.y = s_tile[(ids[u].x >> 16) & 0x7fffu];
            v.z = s_tile[ids[u].y & 0x7fffu];
            v.w = s_tile[(ids[u].y >> 16) & 0x7fffu];
            __builtin_nontemporal_store(v, (V4*) (val + o[u]) + (lane & 7));   // o is a multiple of G: aligned lines
        }
    }
}

// Phase 1.  Two instantiations per element type, launched one after the other: CLASSED = true runs the class forms
// (E1 .. H), CLASSED = false the generic pair form and the edge form (one kernel holding all of them needs more than the
// 128 VGPRs a 1024-thread workgroup may have: the fp32 build spilled).  Each has its own work counter; the big items
// come first, the small ones fill the gaps at the end.
template <typename S, int TILE, bool CLASSED>
__global__ void __launch_bounds__(PRC_THREADS)
pr_cold_tile_kernel(const prc_item1* __restrict__ items, int n_items, unsigned int* __restrict__ queue,
                    const S* __restrict__ contrib, const int32_t* __restrict__ org, int nranks, int64_t span, int64_t slice, int64_t T,
                    const uint16_t* __restrict__ srcl, const int32_t* __restrict__ ob, S* __restrict__ val, unsigned sink_base,
                    unsigned qbase, const int32_t* __restrict__ vst) {
    __shared__ S s_tile[TILE];
    __shared__ S s_stage[PRC_WAVES][CLASSED ? 64 : PRC_STAGE_BYTES / sizeof(S) + 64];   // generic form: + a dummy slot per lane
    __shared__ int s_item;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned sink = sink_base + (blockIdx.x * PRC_WAVES + wv) * 64 + lane;   // where a lane's idle stores go
    int loaded = -1;
    for (;;) {
        __syncthreads();   // everybody is done with s_item and the tile of the previous item
        if (tid == 0) s_item = (int) (atomicAdd(&queue[CLASSED ? PRC_Q1E : PRC_Q1P], 1u) - qbase);
        __syncthreads();
        const int it = s_item;
        if ((unsigned) it >= (unsigned) n_items) break;   // (unsigned: a counter that ran away from the host's copy ends the kernel)
        const prc_item1 d = items[it];
        if (d.tile != loaded) {   // (workgroup-uniform) chunks of one tile often follow each other
            prc_load_tile<S, TILE>(s_tile, d.tile, org, contrib, nranks, span, slice, T);
            loaded = d.tile;
            __syncthreads();
        }
        if constexpr (CLASSED) {
            // the item is a run of groups of one tile: walk the class streams it crosses (stream c of tile t covers the
            // groups [vst[8 t + c], vst[8 t + c + 1]); all bounds are multiples of a block)
            const int32_t* vs = vst + ((int64_t) d.tile << PRC_CLS_BITS);
            for (int cls = PRC_CL_E1; cls <= PRC_CL_H; cls++) {
                const int32_t lo = max(d.g0, vs[cls]), hi = min(d.g1, vs[cls + 1]);
                if (lo >= hi) continue;   // (workgroup-uniform)
                const prc_item1 seg{d.tile, lo, hi, cls + 1};
                switch (cls) {
                case PRC_CL_E1: prc_class_item<S, PRC_FORM_E1>(seg, s_tile, s_stage[wv], srcl, ob, val, sink, wv, lane); break;
                case PRC_CL_E2: prc_class_item<S, PRC_FORM_E2>(seg, s_tile, s_stage[wv], srcl, ob, val, sink, wv, lane); break;
                case PRC_CL_E4: prc_class_item<S, PRC_FORM_E4>(seg, s_tile, s_stage[wv], srcl, ob, val, sink, wv, lane); break;
                case PRC_CL_E8: prc_class_item<S, PRC_FORM_E8>(seg, s_tile, s_stage[wv], srcl, ob, val, sink, wv, lane); break;
                default: prc_class_item<S, PRC_FORM_H>(seg, s_tile, s_stage[wv], srcl, ob, val, sink, wv, lane); break;
                }
            }
        } else {
            if (d.form == PRC_FORM_PAIR) prc_pair_item<S>(d, s_tile, s_stage[wv], srcl, ob, val, sink, wv, lane);
            else prc_edge_item<S>(d, s_tile, srcl, ob, val, wv, lane);
        }
    }
}

// value -> fixed point.  Contributions are in [0, 1] (rank <= 1, out-degree >= 1) and a row's cold terms add up
// to at most the sum of all ranks (<= 1), so the 2^-62 limb cannot overflow; the second limb of the fp64 form
// holds what the first one truncates, with lo_bits chosen at plan time from the largest number of terms.
__device__ __forceinline__ unsigned long long prc_fix_hi(double v) { return (unsigned long long) (long long) (v * 0x1p62); }

// The PageRank update of active row i from its finished neighbour sum (pagerank.gm:13-16; what pr_combine_kernel
// does when the pull sweep has a share): val = (1-d)/N + d * sum, diff += |val - rank|, new rank and contribution.
template <typename S>
__device__ __forceinline__ void prc_finish_row(const pr_cold_fuse& fz, int64_t i, double sum, double& diff_acc) {
    const int32_t od = __builtin_nontemporal_load(fz.outdeg_c + i);
    const int64_t r = __builtin_nontemporal_load(fz.active + i);
    const double old = (double) __builtin_nontemporal_load((const S*) fz.rk_c + i);
    const S vs = (S) (fz.base + fz.d * sum);
    diff_acc += fabs((double) vs - old);
    __builtin_nontemporal_store(vs, (S*) fz.rk_c + i);
    __builtin_nontemporal_store(od > 0 ? (S) ((double) vs / (double) od) : (S) 0, (S*) fz.next_owned + r);
}

// sum of the block's diff_acc in a fixed order (lanes by shuffles, then the waves in order) -> *dst
template <int THREADS>
__device__ __forceinline__ void prc_block_diff(double diff_acc, double* s_red, double* __restrict__ dst) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) diff_acc += __shfl_down(diff_acc, off, 64);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = diff_acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < THREADS / 64; w++) t += s_red[w];
        *dst = t;
    }
}

template <typename S, int BINROWS, int LIMBS, bool FUSE>
__global__ void __launch_bounds__(PRC_THREADS)
pr_cold_accum_kernel(const prc_item2* __restrict__ items, int n_items, unsigned int* __restrict__ queue,
                     const uint16_t* __restrict__ rowl, const S* __restrict__ val, int64_t nactive, double lo_scale,
                     S* __restrict__ cold, unsigned long long* __restrict__ scratch, pr_cold_fuse fz,
                     double* __restrict__ diff_part, unsigned qbase) {
    typedef typename prc_vec4<S>::type V4;
    __shared__ unsigned long long s_acc[LIMBS * BINROWS];
    __shared__ double s_red[PRC_WAVES];
    __shared__ int s_item;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (;;) {
        __syncthreads();   // the previous item has been flushed
        if (tid == 0) s_item = (int) (atomicAdd(&queue[PRC_Q2], 1u) - qbase);
#pragma unroll 4
        for (int i = tid; i < LIMBS * BINROWS; i += PRC_THREADS) s_acc[i] = 0ull;
        __syncthreads();
        const int it = s_item;
        if ((unsigned) it >= (unsigned) n_items) break;   // (unsigned: a counter that ran away from the host's copy ends the kernel)
        const prc_item2 d = items[it];
        const int npieces = (d.g1 - d.g0 + 7) >> 3;
        for (int base = 0; base < npieces; base += PRC_WAVES * PRC_UNROLL) {
            prc_u32x2 ids[PRC_UNROLL];
            V4 v[PRC_UNROLL];
            bool ok[PRC_UNROLL];
#pragma unroll
            for (int u = 0; u < PRC_UNROLL; u++) {
                const int pc = base + u * PRC_WAVES + wv;
                const int g = d.g0 + pc * 8 + (lane >> 3);
                ok[u] = pc < npieces && g < d.g1;
                if (ok[u]) {
                    const int64_t at = (int64_t) g * (PRC_G / 4) + (lane & 7);
                    ids[u] = __builtin_nontemporal_load((const prc_u32x2*) rowl + at);
                    v[u] = __builtin_nontemporal_load((const V4*) val + at);
                }
            }
#pragma unroll
            for (int u = 0; u < PRC_UNROLL; u++) {
                if (!ok[u]) continue;
                const unsigned r[4] = {ids[u].x & 0xffffu, ids[u].x >> 16, ids[u].y & 0xffffu, ids[u].y >> 16};
                const double x[4] = {(double) v[u].x, (double) v[u].y, (double) v[u].z, (double) v[u].w};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (r[j] == PRC_PAD) continue;
                    PRC_CHECK(r[j] < (unsigned) BINROWS && x[j] >= 0.0 && x[j] <= 1.0);
                    const unsigned long long hi = prc_fix_hi(x[j]);
                    __hip_atomic_fetch_add(&s_acc[r[j]], hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (LIMBS > 1) {
                        const double rem = x[j] - (double) (long long) hi * 0x1p-62;   // exact: the bits hi dropped
                        const unsigned long long lo = (unsigned long long) (long long) (rem * 0x1p62 * lo_scale);   // rem < 2^-62
                        __hip_atomic_fetch_add(&s_acc[BINROWS + r[j]], lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        }
        __syncthreads();
        double diff_acc = 0.0;
        if (d.slot < 0) {
            const int64_t i0 = (int64_t) d.bin * BINROWS;
#pragma unroll 4
            for (int r = tid; r < BINROWS; r += PRC_THREADS) {
                if (i0 + r >= nactive) break;
                double sum = (double) (long long) s_acc[r] * 0x1p-62;
                if (LIMBS > 1) sum += (double) (long long) s_acc[BINROWS + r] * (0x1p-62 / lo_scale);
                if (FUSE) prc_finish_row<S>(fz, i0 + r, sum, diff_acc);
                else __builtin_nontemporal_store((S) sum, cold + i0 + r);
            }
        } else {
            unsigned long long* dst = scratch + (int64_t) d.slot * (LIMBS * BINROWS);
#pragma unroll 4
            for (int i = tid; i < LIMBS * BINROWS; i += PRC_THREADS) dst[i] = s_acc[i];
        }
        if (FUSE) prc_block_diff<PRC_THREADS>(diff_acc, s_red, diff_part + it);   // one partial per work item (0 for chunks)
    }
}

// Phase 3.  A split bin's chunk accumulators (nslots x BINROWS, 128 KiB apart) are added per row: a block takes 64
// consecutive rows (one 512-byte line set per slot and wave) and deals the slots to its 16 waves, which then meet
// in LDS.  Integer adds: any order gives the same bits.
template <typename S, int BINROWS, int LIMBS, bool FUSE>
__global__ void __launch_bounds__(1024)
pr_cold_reduce_kernel(const prc_item3* __restrict__ items, int64_t nactive, double lo_scale,
                      const unsigned long long* __restrict__ scratch, S* __restrict__ cold, pr_cold_fuse fz,
                      double* __restrict__ diff_part) {
    __shared__ unsigned long long s_part[LIMBS][16][64];
    const prc_item3 d = items[blockIdx.y];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * 64 + lane;
    const int64_t i0 = (int64_t) d.bin * BINROWS;
    unsigned long long hi = 0, lo = 0;
    const unsigned long long* src = scratch + (int64_t) d.slot0 * (LIMBS * BINROWS) + r;
#pragma unroll 4
    for (int s = wv; s < d.nslots; s += 16) {
        hi += __builtin_nontemporal_load(src + (int64_t) s * (LIMBS * BINROWS));
        if (LIMBS > 1) lo += __builtin_nontemporal_load(src + (int64_t) s * (LIMBS * BINROWS) + BINROWS);
    }
    s_part[0][wv][lane] = hi;
    if (LIMBS > 1) s_part[LIMBS - 1][wv][lane] = lo;
    __syncthreads();
    if (wv == 0 && i0 + r < nactive) {
        hi = 0;
        lo = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            hi += s_part[0][w][lane];
            if (LIMBS > 1) lo += s_part[LIMBS - 1][w][lane];
        }
        double v = (double) (long long) hi * 0x1p-62;
        if (LIMBS > 1) v += (double) (long long) lo * (0x1p-62 / lo_scale);
        if (FUSE) {
            double diff_acc = 0.0;
            prc_finish_row<S>(fz, i0 + r, v, diff_acc);
            s_part[0][0][lane] = (unsigned long long) __double_as_longlong(diff_acc);   // (wave 0 only: its own slots)
        } else cold[i0 + r] = (S) v;
    } else if (FUSE && wv == 0) s_part[0][0][lane] = 0ull;
    if (FUSE && wv == 0) {   // the 64 rows' |val - rank| in lane order: one partial per block
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            double t = 0.0;
            for (int l = 0; l < 64; l++) t += __longlong_as_double((long long) s_part[0][0][l]);
            diff_part[(int64_t) blockIdx.y * gridDim.x + blockIdx.x] = t;
        }
    }
}

// The same for bins split into a few chunks only (small partitions: most bins are cut in two or three): a wave takes
// 64 rows and walks the slots itself -- no idle waves, no LDS round.
#define PRC_FEW_SLOTS 8
template <typename S, int BINROWS, int LIMBS, bool FUSE>
__global__ void __launch_bounds__(256)
pr_cold_reduce_few_kernel(const prc_item3* __restrict__ items, int64_t nactive, double lo_scale,
                          const unsigned long long* __restrict__ scratch, S* __restrict__ cold, pr_cold_fuse fz,
                          double* __restrict__ diff_part) {
    const prc_item3 d = items[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int rg = blockIdx.x * 4 + (threadIdx.x >> 6);   // group of 64 rows
    const int r = rg * 64 + lane;
    const int64_t i0 = (int64_t) d.bin * BINROWS;
    unsigned long long hi = 0, lo = 0;
    const unsigned long long* src = scratch + (int64_t) d.slot0 * (LIMBS * BINROWS) + r;
#pragma unroll 4
    for (int s = 0; s < d.nslots; s++) {
        hi += __builtin_nontemporal_load(src + (int64_t) s * (LIMBS * BINROWS));
        if (LIMBS > 1) lo += __builtin_nontemporal_load(src + (int64_t) s * (LIMBS * BINROWS) + BINROWS);
    }
    double diff_acc = 0.0;
    if (i0 + r < nactive) {
        double v = (double) (long long) hi * 0x1p-62;
        if (LIMBS > 1) v += (double) (long long) lo * (0x1p-62 / lo_scale);
        if (FUSE) prc_finish_row<S>(fz, i0 + r, v, diff_acc);
        else cold[i0 + r] = (S) v;
    }
    if (FUSE) {   // the 64 rows' |val - rank| in lane order, like the many-slot kernel
        double t = 0.0;
        for (int l = 0; l < 64; l++) t += __shfl(diff_acc, l, 64);
        if (lane == 0) diff_part[(int64_t) blockIdx.y * (BINROWS / 64) + rg] = t;
    }
}

// ------------------------------------------------------------------ work lists
// (Re)build the device work lists from the host copies: phase 1 grouped by tile class, phases 2-3 by part; inside a
// group the big items come first (the queue then balances the tail with small ones), pair items before edge items,
// few-slot reductions before many-slot ones.
static int prc_upload_lists(pr_cold* c, const std::vector<uint8_t>* tile_class, const std::vector<int32_t>* bin_part, int nparts) {
    auto by_size1 = [](const prc_item1& a, const prc_item1& b) { return a.g1 - a.g0 > b.g1 - b.g0; };
    auto by_size2 = [](const prc_item2& a, const prc_item2& b) { return a.g1 - a.g0 > b.g1 - b.g0; };
    std::vector<prc_item1> l1;
    for (int k = 0; k < 2; k++) {
        c->o1[k] = (int64_t) l1.size();
        for (int part = 0; part < 3; part++) {   // class forms | generic pair form | edge form
            if (part == 1) c->o1g[k] = (int64_t) l1.size();
            std::vector<prc_item1> a;
            for (const prc_item1& it : (part < 2 ? c->h1p : c->h1e))
                if ((tile_class ? (int) (*tile_class)[(size_t) it.tile] : 0) == k && (part == 2 || (part == 0) == (it.form >= PRC_FORM_E1))) a.push_back(it);
            std::stable_sort(a.begin(), a.end(), by_size1);
            l1.insert(l1.end(), a.begin(), a.end());
        }
    }
    c->o1[2] = (int64_t) l1.size();
    std::vector<prc_item2> l2;
    std::vector<prc_item3> l3;
    c->o2.assign((size_t) nparts + 1, 0);
    c->o3.assign((size_t) nparts + 1, 0);
    c->o3few.assign((size_t) nparts, 0);
    for (int q = 0; q < nparts; q++) {
        c->o2[(size_t) q] = (int64_t) l2.size();
        c->o3[(size_t) q] = (int64_t) l3.size();
        std::vector<prc_item2> a;
        for (const prc_item2& it : c->h2)
            if ((bin_part ? (*bin_part)[(size_t) it.bin] : 0) == q) a.push_back(it);
        std::stable_sort(a.begin(), a.end(), by_size2);
        l2.insert(l2.end(), a.begin(), a.end());
        for (int few = 1; few >= 0; few--)
            for (const prc_item3& it : c->h3)
                if ((bin_part ? (*bin_part)[(size_t) it.bin] : 0) == q && (it.nslots <= PRC_FEW_SLOTS ? 1 : 0) == few) {
                    l3.push_back(it);
                    if (few) c->o3few[(size_t) q]++;
                }
    }
    c->o2[(size_t) nparts] = (int64_t) l2.size();
    c->o3[(size_t) nparts] = (int64_t) l3.size();
    c->nparts = nparts;
    if ((int64_t) l1.size() != c->n1p + c->n1e || (int64_t) l2.size() != c->n2 || (int64_t) l3.size() != c->n3) {
        gmx_set_error("pr cold: work lists lost items while being regrouped");
        return GMX_ERR_ARG;
    }
    if (!l1.empty()) GMX_HIP(hipMemcpy(c->it1p.p, l1.data(), sizeof(prc_item1) * l1.size(), hipMemcpyHostToDevice));
    if (!l2.empty()) GMX_HIP(hipMemcpy(c->it2.p, l2.data(), sizeof(prc_item2) * l2.size(), hipMemcpyHostToDevice));
    if (!l3.empty()) GMX_HIP(hipMemcpy(c->it3.p, l3.data(), sizeof(prc_item3) * l3.size(), hipMemcpyHostToDevice));
    return GMX_OK;
}

// Cut the step into parts.  act_bound[j] (j = 0 .. nparts, row order): position in the active-row list where row part
// j starts; parts are numbered in PROCESSING order, which is from the last row part to the first, and the bin that
// holds a boundary goes with the later row part (processed earlier), so every row of a part's range is finished when
// the part is.  Tile classes: the entries [0, hub) of a rank range are its hub piece, [hub, live) the rest of what is
// ever read; a tile is class 0 if every live source it holds lies in a hub piece.
int pr_cold_set_parts(pr_cold* c, int nparts, const int64_t* act_bound, int64_t hub, int64_t live) {
    if (!c || c->Ec == 0) return GMX_OK;
    if (nparts <= 1) return prc_upload_lists(c, nullptr, nullptr, 1);
    std::vector<int32_t> bin_part((size_t) c->nbins, 0);
    {
        std::vector<int64_t> bb((size_t) nparts + 1, 0);
        for (int j = 1; j < nparts; j++) bb[(size_t) j] = std::min<int64_t>(c->nbins, act_bound[j] / c->binrows);
        bb[(size_t) nparts] = c->nbins;
        for (int j = 0; j < nparts; j++)
            for (int64_t b = bb[(size_t) j]; b < bb[(size_t) j + 1]; b++) bin_part[(size_t) b] = nparts - 1 - j;
    }
    std::vector<uint8_t> tile_class((size_t) c->ntiles, 0);
    const int64_t span = c->prm.slice - c->prm.T, tile_src = c->tile - 1;
    for (int64_t t = 0; t < c->ntiles; t++) {
        const int64_t cp0 = t * tile_src, cp1 = std::min<int64_t>(cp0 + tile_src, span * c->prm.nranks);
        bool hub_only = true;
        for (int64_t r = cp0 / span; r * span < cp1 && hub_only; r++) {
            const int64_t o0 = c->prm.T + std::max<int64_t>(cp0, r * span) - r * span;          // range offsets [o0, o1)
            const int64_t o1 = c->prm.T + std::min<int64_t>(cp1, (r + 1) * span) - r * span;
            if (o0 < live && std::min(o1, live) > hub) hub_only = false;
        }
        tile_class[(size_t) t] = hub_only ? 0 : 1;
    }
    return prc_upload_lists(c, &tile_class, &bin_part, nparts);
}

int pr_cold_parts(const pr_cold* c) { return c ? c->nparts : 1; }
int64_t pr_cold_class_items(const pr_cold* c, int cls) { return c && cls >= 0 && cls < 2 ? c->o1[cls + 1] - c->o1[cls] : 0; }

// ------------------------------------------------------------------ plan
#define PRC_TRY(expr, what)                                                                  \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            gmx_set_error("pr cold plan: %s failed: %s", what, hipGetErrorString(_e));       \
            st = GMX_ERR_HIP;                                                                \
            goto done;                                                                       \
        }                                                                                    \
    } while (0)
#define PRC_ALLOC(buf, count)                        \
    do {                                             \
        if ((st = (buf).alloc((size_t) (count)))) goto done; \
    } while (0)

static int prc_env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

// exclusive prefix sums of n int32 values (in != out)
static hipError_t prc_exscan(const int32_t* in, int32_t* out, int64_t n, wbuf<char>& tmp, hipStream_t s) {
    size_t tb = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, tb, in, out, 0, (size_t) n, rocprim::plus<int32_t>(), s);
    if (e != hipSuccess) return e;
    if (tmp.n < tb) {
        tmp.release();
        if (tmp.alloc(tb)) return hipErrorOutOfMemory;
    }
    return rocprim::exclusive_scan((void*) tmp.p, tb, in, out, 0, (size_t) n, rocprim::plus<int32_t>(), s);
}

// inclusive prefix sums of n int32 values (in != out)
static hipError_t prc_inscan(const int32_t* in, int32_t* out, int64_t n, wbuf<char>& tmp, hipStream_t s) {
    size_t tb = 0;
    hipError_t e = rocprim::inclusive_scan(nullptr, tb, in, out, (size_t) n, rocprim::plus<int32_t>(), s);
    if (e != hipSuccess) return e;
    if (tmp.n < tb) {
        tmp.release();
        if (tmp.alloc(tb)) return hipErrorOutOfMemory;
    }
    return rocprim::inclusive_scan((void*) tmp.p, tb, in, out, (size_t) n, rocprim::plus<int32_t>(), s);
}

int pr_cold_create(const uint64_t* keys, int64_t Ec, const pr_cold_params& prm, hipStream_t s, pr_cold** outp) {
    *outp = nullptr;
    GMX_REQUIRE(prm.elem == 4 || prm.elem == 8, "pr cold: bad element size");
    GMX_REQUIRE(prm.T >= 0 && prm.T < prm.slice, "pr cold: threshold outside the rank range");
    GMX_REQUIRE(Ec < (1LL << 31) - (1LL << 27), "pr cold: %lld edges exceed the int32 stream positions", (long long) Ec);
    pr_cold* c = new pr_cold();
    c->prm = prm;
    c->Ec = Ec;
    // GMX_PR_COLD_CLASS_DENSITY: 1 (default) = every tile in length classes (pair tiles by pair length, edge tiles as
    // class E1) with the larger tile that leaves; 0 = round 2's generic pair / edge forms; n > 1 = classes only for the
    // pair tiles whose (tile, bin) cells average >= n edges, the generic forms for the rest (development)
    const int64_t class_density = prc_env_int("GMX_PR_COLD_CLASS_DENSITY", 1);
    const bool all_classed = class_density == 1;
    c->tile = all_classed ? prc_tile_elems_classed(prm.elem) : prc_tile_elems(prm.elem);
    const int tile_src = c->tile - 1;
    c->limbs = prm.elem == 4 ? 1 : 2;
    c->binrows = PRC_LDS_BYTES / 8 / c->limbs;
    c->ncold = (int64_t) prm.nranks * (prm.slice - prm.T);
    c->ntiles = (c->ncold + tile_src - 1) / tile_src;
    c->nbins = (prm.nactive + c->binrows - 1) / c->binrows;
    if (c->nbins < 1) c->nbins = 1;
    const int tilebits = gmx_bits_for(c->ntiles), binbits = gmx_bits_for(c->nbins), vtbits = tilebits + PRC_CLS_BITS;
    const int64_t nvt = c->ntiles << PRC_CLS_BITS;
    // an edge tile has pairs averaging fewer than this many edges (GMX_PR_COLD_PAIR_X100 overrides, in percent)
    const double pair_min = prc_env_int("GMX_PR_COLD_PAIR_X100", 125) / 100.0;
    // (measured on RMAT-26 fp32 with the generic form's tile, ms per step: density 0 -> 1.75, 1024 (80 tiles classed) -> 1.49,
    // 256 (189 tiles) -> 1.46, every pair tile -> 1.43: the (class, bin) sub-cells stay fuller than feared, 1.51 M -> 1.61 M cells)
    int st = GMX_OK;
    gmx_tick tick("cold plan");
    gmx_ws_scope ws;   // every temporary below is workspace memory
    wbuf<uint64_t> k1, k2;
    wbuf<int32_t> cflag, cincl, ps, pincl, pstart, plen, ppos, pos, endf, endpre, first, pfirst, groups, c1raw, c1, vtab, delta, counts, id, order2, groups2,
        c2s, c2, tab2, tstat;
    wbuf<uint32_t> ckey, key2, key2s;
    wbuf<uint8_t> mode;
    wbuf<char> tmp;
    uint64_t* sk = nullptr;
    std::vector<int32_t> hts, hvt, hdelta, vstart, h2;
    std::vector<uint8_t> hmode;
    std::vector<prc_item1> v1p, v1e;
    std::vector<prc_item2> v2;
    std::vector<prc_item3> v3;
    int64_t ngroups1 = 0, ngroups2 = 0, nc = 0, np = 0, pair_edges = 0, pair_tiles = 0, class_tiles = 0, class_edges = 0;
    hipDeviceProp_t prop;
    int dev = 0;
    if (vtbits + binbits > 32) {
        gmx_set_error("pr cold: %lld tiles x %lld bins do not fit the sort key", (long long) c->ntiles, (long long) c->nbins);
        st = GMX_ERR_ARG;
        goto done;
    }
    PRC_ALLOC(c->cold, (size_t) (prm.nactive ? prm.nactive : 1) * prm.elem);
    PRC_TRY(hipMemsetAsync(c->cold.p, 0, (size_t) (prm.nactive ? prm.nactive : 1) * prm.elem, s), "memset");
    PRC_ALLOC(c->queue, 3 * 64);
    PRC_TRY(hipMemsetAsync(c->queue.p, 0, 3 * 64 * sizeof(unsigned int), s), "memset");
    (void) hipGetDevice(&dev);
    PRC_TRY(hipGetDeviceProperties(&prop, dev), "hipGetDeviceProperties");
    c->grid = prop.multiProcessorCount;
    if (Ec == 0) {
        PRC_TRY(hipStreamSynchronize(s), "sync");
        *outp = c;
        return GMX_OK;
    }
    {   // ---- (tile, [class,] bin, row, source) keys, sorted ----
        PRC_ALLOC(k1, Ec);
        PRC_ALLOC(k2, Ec);
        hipLaunchKernelGGL(prc_keys_kernel, dim3(prc_grid_for(Ec)), dim3(256), 0, s, keys, Ec, prm, tile_src, c->binrows, binbits, k1.p);
        rocprim::double_buffer<uint64_t> db(k1.p, k2.p);
        size_t tb = 0;
        const unsigned end_bit = 32u + (unsigned) binbits + (unsigned) vtbits;
        // (the 16 source bits stay out of the sort: the sort is stable, so a pair's sources keep the order the caller's
        // keys came in -- a fixed order is all the sums need -- and two of eight radix passes are saved)
        PRC_TRY(rocprim::radix_sort_keys(nullptr, tb, db, (size_t) Ec, 16u, end_bit, s), "sort size");
        PRC_ALLOC(tmp, tb);
        PRC_TRY(rocprim::radix_sort_keys((void*) tmp.p, tb, db, (size_t) Ec, 16u, end_bit, s), "sort");
        PRC_TRY(hipStreamSynchronize(s), "sort sync");
        sk = db.current();
    }
    tick.mark("keys + main sort");
    PRC_ALLOC(cflag, Ec + 1);
    PRC_ALLOC(cincl, Ec + 1);
    PRC_ALLOC(ps, Ec + 1);
    PRC_ALLOC(pincl, Ec + 1);
    PRC_ALLOC(mode, c->ntiles + 1);
    PRC_ALLOC(tstat, 3 * (c->ntiles + 1));
    hmode.assign((size_t) c->ntiles + 1, PRC_TM_EDGE);
    for (int round = 0; round < 2; round++) {
        // ---- cells ((virtual) tile, bin) and pairs ((virtual) tile, row): runs of the sorted keys ----
        hipLaunchKernelGGL(prc_runs_kernel, dim3(prc_grid_for(Ec)), dim3(256), 0, s, (const uint64_t*) sk, Ec,
                           round ? (const uint8_t*) mode.p : (const uint8_t*) nullptr, 32 + binbits + PRC_CLS_BITS, cflag.p, ps.p);
        PRC_TRY(prc_inscan(cflag.p, cincl.p, Ec, tmp, s), "scan");
        PRC_TRY(prc_inscan(ps.p, pincl.p, Ec, tmp, s), "scan");
        {
            int32_t lastc = 0, lastp = 0;
            PRC_TRY(hipMemcpyAsync(&lastc, cincl.p + (Ec - 1), 4, hipMemcpyDeviceToHost, s), "copy");
            PRC_TRY(hipMemcpyAsync(&lastp, pincl.p + (Ec - 1), 4, hipMemcpyDeviceToHost, s), "copy");
            PRC_TRY(hipStreamSynchronize(s), "sync");
            c->ncells = nc = lastc;
            np = lastp;
        }
        if (pstart.n < (size_t) np + 1) { pstart.release(); PRC_ALLOC(pstart, np + 1); }
        hipLaunchKernelGGL(prc_pair_start_kernel, dim3(prc_grid_for(Ec + 1)), dim3(256), 0, s, (const int32_t*) ps.p, (const int32_t*) pincl.p, Ec, np, pstart.p);
        if (round == 1) break;
        // ---- per tile: edge form, generic pair form, or length classes ----
        hipLaunchKernelGGL(prc_tile_stats_kernel, dim3(prc_grid_for(c->ntiles + 1)), dim3(256), 0, s, (const uint64_t*) sk, Ec,
                           32 + binbits + PRC_CLS_BITS, c->ntiles, (const int32_t*) pincl.p, (const int32_t*) cincl.p, np, nc, tstat.p);
        hts.resize((size_t) 3 * (c->ntiles + 1));
        PRC_TRY(hipMemcpyAsync(hts.data(), tstat.p, sizeof(int32_t) * hts.size(), hipMemcpyDeviceToHost, s), "copy");
        PRC_TRY(hipStreamSynchronize(s), "sync");
        for (int64_t t = 0; t < c->ntiles; t++) {
            const int64_t ne = (int64_t) hts[3 * (t + 1)] - hts[3 * t], npt = (int64_t) hts[3 * (t + 1) + 1] - hts[3 * t + 1],
                          nct = (int64_t) hts[3 * (t + 1) + 2] - hts[3 * t + 2];
            hmode[t] = (npt > 0 && (double) ne >= pair_min * (double) npt) ? PRC_TM_PAIR : PRC_TM_EDGE;
            if (hmode[t] == PRC_TM_PAIR) { pair_edges += ne; pair_tiles++; }
            if (hmode[t] == PRC_TM_PAIR && class_density > 0 && ne >= class_density * nct) { hmode[t] = PRC_TM_CLASSED; class_tiles++; class_edges += ne; }
            else if (hmode[t] == PRC_TM_EDGE && all_classed && ne > 0) { hmode[t] = PRC_TM_SINGLES; class_tiles++; class_edges += ne; }
        }
        PRC_TRY(hipMemcpyAsync(mode.p, hmode.data(), hmode.size(), hipMemcpyHostToDevice, s), "copy");
        if (class_tiles == 0) continue;   // (round 1 only re-derives what it already has: cheap next to the sort)
        // ---- the pairs of the classed tiles move into their class's stream: stable sort on the (tile, class) field ----
        hipLaunchKernelGGL(prc_class_kernel, dim3(prc_grid_for(Ec)), dim3(256), 0, s, sk, Ec, (const int32_t*) pincl.p, (const int32_t*) pstart.p,
                           (const uint8_t*) mode.p, binbits);
        {
            uint64_t* other = sk == k1.p ? k2.p : k1.p;
            rocprim::double_buffer<uint64_t> db(sk, other);
            size_t tb = 0;
            const unsigned b0 = 32u + (unsigned) binbits, b1 = b0 + (unsigned) vtbits;
            PRC_TRY(rocprim::radix_sort_keys(nullptr, tb, db, (size_t) Ec, b0, b1, s), "sort size");
            if (tmp.n < tb) { tmp.release(); PRC_ALLOC(tmp, tb); }
            PRC_TRY(rocprim::radix_sort_keys((void*) tmp.p, tb, db, (size_t) Ec, b0, b1, s), "sort");
            PRC_TRY(hipStreamSynchronize(s), "sort sync");
            sk = db.current();
        }
    }
    if (sk == k1.p) k2.release(); else k1.release();
    tick.mark("runs, classes, class sort");
    ps.release();
    // ---- padded pair lengths -> stream offsets; cells; groups per cell ----
    PRC_ALLOC(plen, np + 1);
    PRC_ALLOC(ppos, np + 1);
    hipLaunchKernelGGL(prc_pair_len_kernel, dim3(prc_grid_for(np + 1)), dim3(256), 0, s, (const uint64_t*) sk, (const int32_t*) pstart.p, np, binbits, plen.p);
    PRC_TRY(prc_exscan(plen.p, ppos.p, np + 1, tmp, s), "scan");
    plen.release();
    PRC_ALLOC(first, nc + 1);
    PRC_ALLOC(pfirst, nc + 1);
    PRC_ALLOC(ckey, nc + 1);
    PRC_ALLOC(groups, nc + 1);
    PRC_ALLOC(c1raw, nc + 1);
    PRC_ALLOC(c1, nc + 1);
    PRC_ALLOC(vtab, 2 * (nvt + 1));
    PRC_ALLOC(delta, nvt + 1);
    hipLaunchKernelGGL(prc_cell_first_kernel, dim3(prc_grid_for(Ec + 1)), dim3(256), 0, s, (const uint64_t*) sk, (const int32_t*) cflag.p,
                       (const int32_t*) cincl.p, (const int32_t*) pincl.p, Ec, nc, np, first.p, pfirst.p, ckey.p);
    hipLaunchKernelGGL(prc_cell_entry_groups_kernel, dim3(prc_grid_for(nc + 1)), dim3(256), 0, s, (const int32_t*) pfirst.p, (const int32_t*) ppos.p, nc, groups.p);
    PRC_TRY(prc_exscan(groups.p, c1raw.p, nc + 1, tmp, s), "scan");
    hipLaunchKernelGGL(prc_vt_table_kernel, dim3(prc_grid_for(nvt + 1)), dim3(256), 0, s, (const uint32_t*) ckey.p, nc, binbits, nvt, (const int32_t*) c1raw.p, vtab.p);
    hvt.resize((size_t) 2 * (nvt + 1));
    PRC_TRY(hipMemcpyAsync(hvt.data(), vtab.p, sizeof(int32_t) * hvt.size(), hipMemcpyDeviceToHost, s), "copy");
    PRC_TRY(hipStreamSynchronize(s), "sync");
    cflag.release();
    tick.mark("pair lengths, cells, groups");
    // ---- where the streams of the virtual tiles start: the generic pair form runs whole super-steps, every other
    //      stream starts on a block boundary ----
    hdelta.assign((size_t) nvt + 1, 0);
    vstart.assign((size_t) nvt + 1, 0);
    for (int64_t v = 0; v < nvt; v++) {
        const int64_t g = (int64_t) hvt[2 * (v + 1) + 1] - hvt[2 * v + 1];
        const bool generic_pair = (v & ((1 << PRC_CLS_BITS) - 1)) == PRC_CL_GEN && hmode[v >> PRC_CLS_BITS] == PRC_TM_PAIR;
        const int64_t unit = generic_pair ? PRC_SUPER_GROUPS : PRC_BLK_GROUPS;
        hdelta[v] = vstart[v] - hvt[2 * v + 1];
        const int64_t nxt = (int64_t) vstart[v] + (g + unit - 1) / unit * unit;
        if (nxt * PRC_G >= (1LL << 31)) {
            gmx_set_error("pr cold: %lld padded entries exceed int32", (long long) (nxt * PRC_G));
            st = GMX_ERR_ARG;
            goto done;
        }
        vstart[v + 1] = (int32_t) nxt;
    }
    ngroups1 = vstart[nvt];
    c->P1 = ngroups1 * PRC_G;
    PRC_TRY(hipMemcpyAsync(delta.p, hdelta.data(), sizeof(int32_t) * hdelta.size(), hipMemcpyHostToDevice, s), "copy");
    hipLaunchKernelGGL(prc_cell_start_kernel, dim3(prc_grid_for(nc)), dim3(256), 0, s, (const int32_t*) c1raw.p,
                       (const uint32_t*) ckey.p, binbits, (const int32_t*) delta.p, nc, c1.p);
    // ---- entry positions, item ends, items per cell, bin-major order ----
    PRC_ALLOC(pos, Ec);
    PRC_ALLOC(endf, Ec + 1);
    PRC_ALLOC(endpre, Ec + 1);
    hipLaunchKernelGGL(prc_pos_end_kernel, dim3(prc_grid_for(Ec + 1)), dim3(256), 0, s, (const uint64_t*) sk, (const int32_t*) cincl.p,
                       (const int32_t*) pincl.p, (const int32_t*) pstart.p, (const int32_t*) ppos.p, (const int32_t*) pfirst.p,
                       (const int32_t*) c1.p, (const uint8_t*) mode.p, binbits, prm.elem, Ec, pos.p, endf.p);
    PRC_TRY(prc_exscan(endf.p, endpre.p, Ec + 1, tmp, s), "scan");   // endpre: item ends before an edge
    tick.mark("positions, item ends");
    pincl.release();
    PRC_ALLOC(counts, nc + 1);
    PRC_ALLOC(key2, nc);
    PRC_ALLOC(key2s, nc);
    PRC_ALLOC(id, nc);
    PRC_ALLOC(order2, nc);
    PRC_ALLOC(groups2, nc + 1);
    PRC_ALLOC(c2s, nc + 1);
    PRC_ALLOC(c2, nc);
    hipLaunchKernelGGL(prc_cell_groups_kernel, dim3(prc_grid_for(nc + 1)), dim3(256), 0, s, (const int32_t*) first.p,
                       (const int32_t*) endpre.p, nc, groups.p, counts.p);
    hipLaunchKernelGGL(prc_cell_key2_kernel, dim3(prc_grid_for(nc)), dim3(256), 0, s, (const uint32_t*) ckey.p, nc, binbits,
                       vtbits, key2.p, id.p);
    {
        size_t tb = 0;
        PRC_TRY(rocprim::radix_sort_pairs(nullptr, tb, key2.p, key2s.p, id.p, order2.p, (size_t) nc, 0u, (unsigned) (binbits + vtbits), s), "sort size");
        if (tmp.n < tb) { tmp.release(); PRC_ALLOC(tmp, tb); }
        PRC_TRY(rocprim::radix_sort_pairs((void*) tmp.p, tb, key2.p, key2s.p, id.p, order2.p, (size_t) nc, 0u, (unsigned) (binbits + vtbits), s), "sort");
    }
    PRC_TRY(hipMemsetAsync(groups2.p + nc, 0, sizeof(int32_t), s), "memset");
    hipLaunchKernelGGL(prc_gather_i32_kernel, dim3(prc_grid_for(nc)), dim3(256), 0, s, (const int32_t*) groups.p,
                       (const int32_t*) order2.p, nc, groups2.p);
    PRC_TRY(prc_exscan(groups2.p, c2s.p, nc + 1, tmp, s), "scan");
    hipLaunchKernelGGL(prc_scatter_i32_kernel, dim3(prc_grid_for(nc)), dim3(256), 0, s, (const int32_t*) c2s.p,
                       (const int32_t*) order2.p, nc, c2.p);
    {
        int32_t tot = 0, ni = 0;
        PRC_TRY(hipMemcpyAsync(&tot, c2s.p + nc, 4, hipMemcpyDeviceToHost, s), "copy");
        PRC_TRY(hipMemcpyAsync(&ni, endpre.p + Ec, 4, hipMemcpyDeviceToHost, s), "copy");
        PRC_TRY(hipStreamSynchronize(s), "sync");
        ngroups2 = tot;
        c->npairs = ni;
    }
    if (ngroups2 * PRC_G >= (1LL << 31) - (int64_t) c->grid * PRC_THREADS) {
        gmx_set_error("pr cold: %lld padded items exceed int32", (long long) (ngroups2 * PRC_G));
        st = GMX_ERR_ARG;
        goto done;
    }
    c->P2 = ngroups2 * PRC_G;
    tick.mark("bin-major order");
    // ---- the two streams ----
    PRC_ALLOC(c->srcl, c->P1);
    PRC_ALLOC(c->ob, ngroups1);
    PRC_ALLOC(c->rowl, c->P2);
    PRC_ALLOC(c->val, ((size_t) c->P2 + (size_t) c->grid * PRC_THREADS) * prm.elem);   // + the sink slots behind the items
    hipLaunchKernelGGL(prc_fill_u16_kernel, dim3(prc_grid_for(c->P1)), dim3(256), 0, s, c->srcl.p, c->P1, (uint16_t) tile_src);
    PRC_TRY(hipMemsetAsync(c->rowl.p, 0xff, (size_t) c->P2 * 2, s), "memset");
    PRC_TRY(hipMemsetAsync(c->val.p, 0, ((size_t) c->P2 + (size_t) c->grid * PRC_THREADS) * prm.elem, s), "memset");
    hipLaunchKernelGGL(prc_fill_items_kernel, dim3(prc_grid_for(Ec)), dim3(256), 0, s, (const uint64_t*) sk, (const int32_t*) cincl.p,
                       (const int32_t*) first.p, (const int32_t*) pos.p, (const int32_t*) c2.p, (const int32_t*) endf.p,
                       (const int32_t*) endpre.p, binbits, (uint16_t) tile_src, Ec, c->srcl.p, c->rowl.p);
    hipLaunchKernelGGL(prc_fill_ob_kernel, dim3(prc_grid_for(ngroups1)), dim3(256), 0, s, (const int32_t*) c1.p, (const int32_t*) c2.p,
                       (const int32_t*) first.p, (const int32_t*) pfirst.p, (const int32_t*) pstart.p, (const int32_t*) ppos.p,
                       (const int32_t*) endpre.p, (const uint32_t*) ckey.p, binbits, nc, ngroups1, (int32_t) c->P2, c->ob.p);
    PRC_ALLOC(tab2, c->nbins + 1);
    hipLaunchKernelGGL(prc_bin_table_kernel, dim3(prc_grid_for(c->nbins + 1)), dim3(256), 0, s, (const uint32_t*) key2s.p,
                       (const int32_t*) c2s.p, nc, vtbits, ngroups2, c->nbins, tab2.p);
    h2.resize((size_t) c->nbins + 1);
    PRC_TRY(hipMemcpyAsync(h2.data(), tab2.p, sizeof(int32_t) * h2.size(), hipMemcpyDeviceToHost, s), "copy");
    PRC_TRY(hipStreamSynchronize(s), "sync");
    tick.mark("streams filled");
    {   // an accumulator never receives more terms than its bin has items: that bound sizes the second limb
        int64_t biggest = 1;
        for (int64_t b = 0; b < c->nbins; b++) biggest = std::max<int64_t>(biggest, ((int64_t) h2[b + 1] - h2[b]) * PRC_G);
        // every term of the second limb is < 2^lo_bits; `biggest` of them must stay below 2^63
        c->lo_bits = 62 - gmx_bits_for(biggest + 1);
        if (c->lo_bits > 62) c->lo_bits = 62;
        if (c->lo_bits < 20) c->lo_bits = 20;
    }
    {   // where the tiles start in the contribution replica
        const int64_t span = prm.slice - prm.T;
        std::vector<int32_t> horg((size_t) 2 * c->ntiles + 2, 0);
        for (int64_t t = 0; t < c->ntiles; t++) {
            const int64_t cp = t * (int64_t) tile_src;
            horg[2 * t] = (int32_t) (cp / span);
            horg[2 * t + 1] = (int32_t) (cp % span);
        }
        PRC_ALLOC(c->torg, horg.size());
        PRC_TRY(hipMemcpy(c->torg.p, horg.data(), sizeof(int32_t) * horg.size(), hipMemcpyHostToDevice), "copy");
    }
    if (prm.deg_by_id) {   // per tile: the largest out-degree of its sources
        wbuf<int32_t> tmax;
        PRC_ALLOC(tmax, c->ntiles);
        PRC_TRY(hipMemsetAsync(tmax.p, 0, sizeof(int32_t) * (size_t) c->ntiles, s), "memset");
        const int64_t nids = (int64_t) prm.nranks * prm.slice;
        hipLaunchKernelGGL(prc_tile_maxdeg_kernel, dim3(prc_grid_for(nids)), dim3(256), 0, s, prm.deg_by_id, nids, prm.slice, prm.T, tile_src, tmax.p);
        c->tile_maxdeg.assign((size_t) c->ntiles, 0);
        PRC_TRY(hipMemcpyAsync(c->tile_maxdeg.data(), tmax.p, sizeof(int32_t) * (size_t) c->ntiles, hipMemcpyDeviceToHost, s), "copy");
        PRC_TRY(hipStreamSynchronize(s), "sync");
    }
    // ---- work lists ----
    {
        // Work item sizes: about four items per CU (an item costs a tile copy -- 124 KiB, as much as 2 Ki groups of
        // entries -- an accumulator flush and a few barriers: RMAT-26 fp32 runs 1.74 ms per iteration with 32 Ki-group
        // items, 1.84 with 8 Ki, 2.06 with 2 Ki), at most 1 Mi entries, and only two per CU when four would make them
        // smaller than 8 Ki groups (rank 0 of an 8-rank partition of RMAT-26: 0.312 ms with 8 Ki, 0.335 with 4 Ki,
        // 0.455 with 1 Ki); phase 2 then only splits the hub bins (every split costs 128 KiB of accumulators written
        // and read again: a bin is cut when it exceeds three chunks).
        const int64_t want1 = std::max<int64_t>(ngroups1 / ((int64_t) c->grid * 4), std::min<int64_t>(8192, ngroups1 / ((int64_t) c->grid * 2)));
        const int64_t want2 = ngroups2 / ((int64_t) c->grid * 2);
        const int ch1 = (int) std::min<int64_t>(262144, std::max<int64_t>(PRC_SUPER_GROUPS, prc_env_int("GMX_PR_COLD_CHUNK1", (int) std::min<int64_t>(want1, 32768)))) / PRC_SUPER_GROUPS * PRC_SUPER_GROUPS;
        const int ch2 = (int) std::min<int64_t>(32768, std::max<int64_t>(8, prc_env_int("GMX_PR_COLD_CHUNK", (int) std::min<int64_t>(std::max<int64_t>(want2, 2048), 32768)))) / 8 * 8;
        // a bin is split when it exceeds thr2 (a split costs its accumulators written and read again)
        const int64_t thr2 = (int64_t) ch2 * prc_env_int("GMX_PR_COLD_SPLIT_X100", 300) / 100;
        for (int64_t t = 0; t < c->ntiles; t++) {
            const int64_t v0 = t << PRC_CLS_BITS;
            if (hmode[t] == PRC_TM_CLASSED || hmode[t] == PRC_TM_SINGLES) {   // one run of groups over all class streams of the tile (whole blocks; the padding ends no item)
                const int32_t g0 = vstart[v0 + PRC_CL_E1], g1 = vstart[v0 + PRC_CL_H + 1];
                for (int32_t g = g0; g < g1; g += ch1) v1p.push_back({(int32_t) t, g, std::min(g1, g + ch1), PRC_FORM_TILE});
                continue;
            }
            const int32_t graw = hvt[2 * (v0 + 1) + 1] - hvt[2 * v0 + 1];
            if (graw == 0) continue;
            const int32_t g0 = vstart[v0];
            const int form = hmode[t] == PRC_TM_PAIR ? PRC_FORM_PAIR : PRC_FORM_EDGE;
            // pair tiles run whole super-steps; edge tiles stop at the last real group (an entry there would be stored)
            const int32_t g1 = form == PRC_FORM_PAIR ? vstart[v0 + 1] : g0 + graw;
            for (int32_t g = g0; g < g1; g += ch1) (form == PRC_FORM_EDGE ? v1e : v1p).push_back({(int32_t) t, g, std::min(g1, g + ch1), form});
        }
        PRC_ALLOC(c->vstart, vstart.size());
        PRC_TRY(hipMemcpy(c->vstart.p, vstart.data(), sizeof(int32_t) * vstart.size(), hipMemcpyHostToDevice), "copy");
        int32_t slot = 0;
        c->all_bins = true;
        for (int64_t b = 0; b < c->nbins; b++) {
            const int32_t g0 = h2[b], g1 = h2[b + 1];
            if (g1 <= g0) { c->all_bins = false; continue; }
            if (g1 - g0 <= thr2) { v2.push_back({(int32_t) b, g0, g1, -1}); continue; }
            const int32_t s0 = slot;
            for (int32_t g = g0; g < g1; g += ch2) v2.push_back({(int32_t) b, g, std::min(g1, g + ch2), slot++});
            v3.push_back({(int32_t) b, s0, slot - s0, 0});
        }
        c->n1p = (int64_t) v1p.size();
        c->n1e = (int64_t) v1e.size();
        c->n2 = (int64_t) v2.size();
        c->n3 = (int64_t) v3.size();
        c->nslots = slot;
        PRC_ALLOC(c->it1p, std::max<size_t>(1, v1p.size() + v1e.size()));
        PRC_ALLOC(c->it2, std::max<size_t>(1, v2.size()));
        PRC_ALLOC(c->it3, std::max<size_t>(1, v3.size()));
        PRC_ALLOC(c->scratch, std::max<size_t>(1, (size_t) slot * c->limbs * c->binrows));
        PRC_ALLOC(c->diffp, std::max<size_t>(1, v2.size() + v3.size() * (size_t) (c->binrows / 64)));
        PRC_TRY(hipMemset(c->diffp.p, 0, sizeof(double) * std::max<size_t>(1, v2.size() + v3.size() * (size_t) (c->binrows / 64))), "memset");
        c->h1p.swap(v1p);
        c->h1e.swap(v1e);
        c->h2.swap(v2);
        c->h3.swap(v3);
        if ((st = prc_upload_lists(c, nullptr, nullptr, 1))) goto done;
    }
    PRC_TRY(hipStreamSynchronize(s), "sync");
    tick.mark("tables, work lists");
    if (getenv("GMX_PR_DEBUG")) {
        std::vector<int64_t> bs;
        for (int64_t b = 0; b < c->nbins; b++) bs.push_back(((int64_t) h2[b + 1] - h2[b]) * PRC_G);
        std::sort(bs.begin(), bs.end());
        int64_t over[4] = {0, 0, 0, 0}, lim[4] = {131072, 262144, 524288, 1048576}, sum_over[4] = {0, 0, 0, 0};
        for (int64_t v : bs)
            for (int q = 0; q < 4; q++)
                if (v > lim[q]) { over[q]++; sum_over[q] += v; }
        fprintf(stderr, "gmx pr cold: bin items min %lld median %lld p90 %lld max %lld; bins over 128K/256K/512K/1M items: %lld/%lld/%lld/%lld holding %lld/%lld/%lld/%lld items\n",
                (long long) bs.front(), (long long) bs[bs.size() / 2], (long long) bs[bs.size() * 9 / 10], (long long) bs.back(),
                (long long) over[0], (long long) over[1], (long long) over[2], (long long) over[3], (long long) sum_over[0],
                (long long) sum_over[1], (long long) sum_over[2], (long long) sum_over[3]);
        int64_t cg[1 << PRC_CLS_BITS] = {0};
        for (int64_t v = 0; v < nvt; v++) cg[v & ((1 << PRC_CLS_BITS) - 1)] += (int64_t) vstart[v + 1] - vstart[v];
        fprintf(stderr, "gmx pr cold: T %lld, %lld edges -> %lld entries in %lld cells (%lld tiles x %lld bins); %lld pair tiles with %lld edges, of them %lld "
                "in length classes with %lld edges; groups by class generic/E1/E2/E4/E8/H %lld/%lld/%lld/%lld/%lld/%lld; %lld natural pairs -> %lld item ends -> "
                "%lld items; work items %lld + %lld / %lld / %lld, %lld slots, lo_bits %d\n", (long long) prm.T, (long long) Ec,
                (long long) c->P1, (long long) nc, (long long) c->ntiles, (long long) c->nbins, (long long) pair_tiles, (long long) pair_edges,
                (long long) class_tiles, (long long) class_edges, (long long) cg[0], (long long) cg[1], (long long) cg[2], (long long) cg[3],
                (long long) cg[4], (long long) cg[5], (long long) np, (long long) c->npairs, (long long) c->P2, (long long) c->n1p,
                (long long) c->n1e, (long long) c->n2, (long long) c->n3, (long long) c->nslots, c->lo_bits);
    }
done:
    if (st != GMX_OK) {
        delete c;
        return st;
    }
    *outp = c;
    return GMX_OK;
}

void pr_cold_free(pr_cold* c) { delete c; }

const void* pr_cold_partial(const pr_cold* c) { return c ? (const void*) c->cold.p : nullptr; }
int64_t pr_cold_edges(const pr_cold* c) { return c ? c->Ec : 0; }
int64_t pr_cold_items(const pr_cold* c) { return c ? c->P2 : 0; }

// phase 1 over the tile classes [k0, k1): per class the class-form items, then the generic pair / edge items
template <typename S, int TILE>
static void prc_gather_tile(pr_cold* c, const void* contrib, int k0, int k1, hipStream_t s) {
    const int64_t span = c->prm.slice - c->prm.T;
    for (int k = k0; k < k1; k++) {
        const int64_t nc = c->o1g[k] - c->o1[k], ng = c->o1[k + 1] - c->o1g[k];
        if (nc > 0) {
            const unsigned grid = (unsigned) std::min<int64_t>(c->grid, nc);
            hipLaunchKernelGGL((pr_cold_tile_kernel<S, TILE, true>), dim3(grid), dim3(PRC_THREADS), 0, s,
                               (const prc_item1*) c->it1p.p + c->o1[k], (int) nc, c->queue.p, (const S*) contrib, (const int32_t*) c->torg.p, c->prm.nranks, span,
                               c->prm.slice, c->prm.T, (const uint16_t*) c->srcl.p, (const int32_t*) c->ob.p, (S*) c->val.p, (unsigned) c->P2, c->q1c, (const int32_t*) c->vstart.p);
            c->q1c += (uint32_t) nc + grid;   // every workgroup claims until its first miss
        }
        if constexpr (TILE == prc_tile_elems((int) sizeof(S))) {   // (the generic forms exist with their own tile size only)
            if (ng > 0) {
                const unsigned grid = (unsigned) std::min<int64_t>(c->grid, ng);
                hipLaunchKernelGGL((pr_cold_tile_kernel<S, TILE, false>), dim3(grid), dim3(PRC_THREADS), 0, s,
                                   (const prc_item1*) c->it1p.p + c->o1g[k], (int) ng, c->queue.p, (const S*) contrib, (const int32_t*) c->torg.p, c->prm.nranks, span,
                                   c->prm.slice, c->prm.T, (const uint16_t*) c->srcl.p, (const int32_t*) c->ob.p, (S*) c->val.p, (unsigned) c->P2, c->q1, (const int32_t*) c->vstart.p);
                c->q1 += (uint32_t) ng + grid;
            }
        }
    }
}
template <typename S>
static void prc_gather(pr_cold* c, const void* contrib, int k0, int k1, hipStream_t s) {
    if (c->tile == prc_tile_elems_classed((int) sizeof(S))) prc_gather_tile<S, prc_tile_elems_classed((int) sizeof(S))>(c, contrib, k0, k1, s);
    else prc_gather_tile<S, prc_tile_elems((int) sizeof(S))>(c, contrib, k0, k1, s);
}

// phases 2 and 3 of part q
template <typename S, bool FUSE>
static void prc_accumulate(pr_cold* c, const pr_cold_fuse& fz, int q, hipStream_t s) {
    constexpr int LIMBS = sizeof(S) == 4 ? 1 : 2;
    constexpr int BINROWS = PRC_LDS_BYTES / 8 / LIMBS;
    const double lo_scale = ldexp(1.0, c->lo_bits);
    const int64_t a2 = c->o2[(size_t) q], n2 = c->o2[(size_t) q + 1] - a2;
    if (n2 > 0) {
        const unsigned grid = (unsigned) std::min<int64_t>(c->grid, n2);
        hipLaunchKernelGGL((pr_cold_accum_kernel<S, BINROWS, LIMBS, FUSE>), dim3(grid), dim3(PRC_THREADS), 0, s,
                           (const prc_item2*) c->it2.p + a2, (int) n2, c->queue.p, (const uint16_t*) c->rowl.p, (const S*) c->val.p,
                           c->prm.nactive, lo_scale, (S*) c->cold.p, c->scratch.p, fz, c->diffp.p + a2, c->q2);
        c->q2 += (uint32_t) n2 + grid;
    }
    const int64_t a3 = c->o3[(size_t) q], n3 = c->o3[(size_t) q + 1] - a3, few = c->o3few[(size_t) q];
    if (few > 0)
        hipLaunchKernelGGL((pr_cold_reduce_few_kernel<S, BINROWS, LIMBS, FUSE>), dim3(BINROWS / 256, (unsigned) few), dim3(256), 0, s,
                           (const prc_item3*) c->it3.p + a3, c->prm.nactive, lo_scale, (const unsigned long long*) c->scratch.p, (S*) c->cold.p,
                           fz, c->diffp.p + c->n2 + a3 * (BINROWS / 64));
    if (n3 > few)
        hipLaunchKernelGGL((pr_cold_reduce_kernel<S, BINROWS, LIMBS, FUSE>), dim3(BINROWS / 64, (unsigned) (n3 - few)), dim3(1024), 0, s,
                           (const prc_item3*) c->it3.p + a3 + few, c->prm.nactive, lo_scale, (const unsigned long long*) c->scratch.p, (S*) c->cold.p,
                           fz, c->diffp.p + c->n2 + (a3 + few) * (BINROWS / 64));
}

// The host mirrors the self-advancing work counters (q1, q1c, q2 += items + grid per launch).  A launch that was refused
// never advanced its counter, so the mirror would be ahead of the device from then on (the kernels compare unsigned and
// just end, but every later step would do nothing): bring both back to zero, in stream order, and report the error.
static int prc_after_launches(pr_cold* c, hipStream_t s) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return GMX_OK;
    (void) hipMemsetAsync(c->queue.p, 0, 3 * 64 * sizeof(unsigned int), s);
    c->q1 = c->q1c = c->q2 = 0;
    gmx_set_error("pr cold: kernel launch failed: %s (work counters reset)", hipGetErrorString(e));
    return GMX_ERR_HIP;
}

// Phase 1 of tile class cls (-1: both): reads the contribution replica, fills the value slots of the class's pairs.
int pr_cold_gather(pr_cold* c, const void* contrib, int cls, hipStream_t s) {
    if (!c || c->Ec == 0 || c->n2 == 0) return GMX_OK;
    const int k0 = cls < 0 ? 0 : cls, k1 = cls < 0 ? 2 : cls + 1;
    if (c->prm.elem == 4) prc_gather<float>(c, contrib, k0, k1, s);
    else prc_gather<double>(c, contrib, k0, k1, s);
    return prc_after_launches(c, s);
}

// Phases 2-3 of part `part` (-1: all, in processing order).  fuse == NULL: leave the row sums in pr_cold_partial().
// Otherwise apply the PageRank update right where a row's sum is finished (needs every edge binned and every bin with
// rows to own an item) and leave |val - rank| partials in pr_cold_diff_partials().
int pr_cold_accumulate(pr_cold* c, const pr_cold_fuse* fuse, int part, hipStream_t s) {
    if (!c || c->Ec == 0 || c->n2 == 0) return GMX_OK;
    const pr_cold_fuse none{};
    const int q0 = part < 0 ? 0 : part, q1 = part < 0 ? c->nparts : part + 1;
    for (int q = q0; q < q1; q++) {
        if (c->prm.elem == 4) {
            if (fuse) prc_accumulate<float, true>(c, *fuse, q, s);
            else prc_accumulate<float, false>(c, none, q, s);
        } else {
            if (fuse) prc_accumulate<double, true>(c, *fuse, q, s);
            else prc_accumulate<double, false>(c, none, q, s);
        }
    }
    return prc_after_launches(c, s);
}

int pr_cold_launch(pr_cold* c, const void* contrib, const pr_cold_fuse* fuse, hipStream_t s) {
    GMX_CHECK(pr_cold_gather(c, contrib, -1, s));
    return pr_cold_accumulate(c, fuse, -1, s);
}

const double* pr_cold_diff_partials(const pr_cold* c, int64_t* n) {
    *n = c ? c->n2 + c->n3 * (c->binrows / 64) : 0;
    return c ? c->diffp.p : nullptr;
}

// The fp32 form adds the pair sums (fp32 values) into ONE limb of 2^-62: a value >= 2^-39 converts exactly, a smaller
// one loses less than 2^-62.  A pair sum is at least the smallest contribution of its tile, and a contribution is at
// least (1-d)/N / outdeg (every rank is >= the teleport term), so only tiles holding a source of out-degree
// > (1-d)/N * 2^39 can produce inexact terms: R such tiles give a row at most R * (1 + TILE/512) truncated terms
// (pairs are cut at the ends of 512-entry blocks), i.e. an absolute error below R * 64 * 2^-62 in its sum and
// d * that in its rank, which is >= (1-d)/N.  The guard asks for a quarter of the 1e-6 bar (2^-22 relative).
// False (d = 1 included: no teleport term, nothing bounds a rank from below) sends the caller to the fp64 plan.
bool pr_cold_limb_guard(const pr_cold* c, double d, double N) {
    if (!c || c->Ec == 0 || c->limbs > 1) return true;
    if (!(d > 0.0 && d < 1.0)) return false;
    if (c->tile_maxdeg.empty()) return false;
    const double base = (1.0 - d) / N, dstar = base * 0x1p39;
    int64_t risky = 0;
    for (int32_t m : c->tile_maxdeg) risky += (double) m > dstar;
    if (getenv("GMX_PR_DEBUG"))
        fprintf(stderr, "gmx pr cold: fp32 limb guard: %lld tiles hold a source of out-degree > %.0f, at most %.1f allowed (d = %g)\n",
                (long long) risky, dstar, 0x1p-22 * base / (64.0 * 0x1p-62 * d), d);
    return (double) risky * 64.0 * 0x1p-62 * d <= 0x1p-22 * base;
}

// every active row lies in a bin that has at least one item: the fused finish then visits all of them
bool pr_cold_covers_all_rows(const pr_cold* c) { return c && c->all_bins; }

// Loads this translation unit's code object (the HIP runtime does that lazily, at the first launch of one of its kernels:
// tens of milliseconds that would otherwise fall into the first timed call) -- called once from the graph constructors.
void gmx_touch_pr_cold() {
    hipFuncAttributes attr;
    (void) hipFuncGetAttributes(&attr, (const void*) prc_runs_kernel);
}
