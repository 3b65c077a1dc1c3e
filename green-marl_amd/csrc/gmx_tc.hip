// gmx_tc.hip -- triangle_counting for gfx950.
//
// Replaces the body of the emitted `triangle_counting` (source
// /root/reference/apps/src/triangle_counting.gm:1-13; restated emission SURVEY.md section 8 a-3):
//     T = #{ (i, j) edge slots of row v : u = node_idx[i] > v, w = node_idx[j] > u, (w -> u) in E }
// i.e. duplicate slots of row v count multiply, the `w.HasEdgeTo(u)` test is boolean
// (binary search on the forward row of w, apps/output_cpp/gm_graph/src/shl_graph.cc:14-62).
// Device formulation: per edge slot (v,u) with u > v, intersect the TAIL of row v (slots with
// value > u, a sorted multiset) with the IN-row of u (w -> u  <=>  w in r_row(u)).  The shorter
// side is walked, the longer side is binary searched:
//   * work <= TC_SMALL: one thread per slot;
//   * otherwise the slot goes to a list and a whole wave strides over the shorter side
//     (coalesced reads, per-lane binary search, __shfl_down reduction).
// Without a reverse CSR the search goes into the forward row of w exactly as emitted.
// A symmetric graph without self-loops and duplicate slots (the symmetrised, de-duplicated inputs triangle
// counting is benchmarked on) has T = number of triangles, which does not depend on the vertex numbering:
// it is counted on a copy renumbered by ascending degree, where "tail of v above u" and "in-row of u above u"
// hold only the higher-degree neighbours -- hubs come last and own almost nothing.  Any other graph is
// counted in its own numbering, as emitted.
// Integer only: exact.
#include "gmx_internal.h"

#include <string.h>
#include <rocprim/rocprim.hpp>

#define TC_THREADS 256
#define TC_SMALL 48

__device__ __forceinline__ int32_t tc_lower_bound(const int32_t* __restrict__ a, int32_t lo, int32_t hi, int32_t x) {
    while (lo < hi) {
        int32_t mid = lo + ((hi - lo) >> 1);
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ bool tc_contains(const int32_t* __restrict__ a, int32_t lo, int32_t hi, int32_t x) {
    int32_t p = tc_lower_bound(a, lo, hi, x);
    return p < hi && a[p] == x;
}

// Multi-GPU: the edge slots are dealt to `nparts` parts in blocks of 2^TC_DEAL_SHIFT slots, round-robin
// (slot work varies by orders of magnitude with the degrees involved; a contiguous split would leave the
// hub rows to one rank).  Local index i of part `part` <-> global slot:
#define TC_DEAL_SHIFT 12
__device__ __forceinline__ int64_t tc_slot_of(int64_t i, int part, int nparts) {
    return ((((i >> TC_DEAL_SHIFT) * nparts) + part) << TC_DEAL_SHIFT) | (i & ((1 << TC_DEAL_SHIFT) - 1));
}

struct tc_pair { int32_t tb, te, rb, re; };  // tail [tb,te) in node_idx, in-row [rb,re) in r_node_idx (values > u)

// walk `short side`, search `long side`; lanes = 1 (thread) or 64 (wave)
template <int LANES>
__device__ __forceinline__ unsigned long long tc_intersect(const int32_t* __restrict__ node_idx,
                                                           const int32_t* __restrict__ r_node_idx,
                                                           tc_pair p, int lane) {
    unsigned long long c = 0;
    const int32_t tl = p.te - p.tb, rl = p.re - p.rb;
    if (tl <= rl) {
        // every tail slot (with multiplicity) that occurs in the in-row
        for (int32_t j = p.tb + lane; j < p.te; j += LANES)
            c += tc_contains(r_node_idx, p.rb, p.re, node_idx[j]) ? 1 : 0;
    } else {
        // every DISTINCT value of the in-row, times its multiplicity in the tail
        for (int32_t k = p.rb + lane; k < p.re; k += LANES) {
            int32_t x = r_node_idx[k];
            if (k > p.rb && r_node_idx[k - 1] == x) continue;
            int32_t lo = tc_lower_bound(node_idx, p.tb, p.te, x);
            int32_t hi = tc_lower_bound(node_idx, lo, p.te, x + 1);
            c += (unsigned long long) (hi - lo);
        }
    }
    return c;
}

__global__ void __launch_bounds__(TC_THREADS)
tc_slots_kernel(const int32_t* __restrict__ begin, const int32_t* __restrict__ node_idx,
                const int32_t* __restrict__ r_begin, const int32_t* __restrict__ r_node_idx,
                int64_t V, int64_t E, int64_t nlocal, int part, int nparts,
                tc_pair* __restrict__ big, unsigned long long* __restrict__ nbig,
                unsigned long long* __restrict__ total) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t) gridDim.x * blockDim.x;
    unsigned long long c = 0;
    for (; i < nlocal; i += stride) {
        const int64_t e = tc_slot_of(i, part, nparts);
        if (e >= E) continue;
        // v = row of slot e
        int64_t lo = 0, hi = V;
        while (hi - lo > 1) {
            int64_t mid = (lo + hi) >> 1;
            if ((int64_t) begin[mid] <= e) lo = mid; else hi = mid;
        }
        const int32_t v = (int32_t) lo, u = node_idx[e];
        if (u <= v) continue;
        tc_pair p;
        p.te = begin[v + 1];
        p.tb = tc_lower_bound(node_idx, (int32_t) e, p.te, u + 1);
        if (p.tb >= p.te) continue;
        p.re = r_begin[u + 1];
        p.rb = tc_lower_bound(r_node_idx, r_begin[u], p.re, u + 1);
        if (p.rb >= p.re) continue;
        const int32_t tl = p.te - p.tb, rl = p.re - p.rb;
        if ((tl < rl ? tl : rl) <= TC_SMALL) c += tc_intersect<1>(node_idx, r_node_idx, p, 0);
        else big[atomicAdd(nbig, 1ULL)] = p;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(total, c);
}

__global__ void __launch_bounds__(TC_THREADS)
tc_big_kernel(const int32_t* __restrict__ node_idx, const int32_t* __restrict__ r_node_idx,
              const tc_pair* __restrict__ big, unsigned long long nbig, unsigned long long* __restrict__ total) {
    const int lane = threadIdx.x & 63;
    unsigned long long wave = ((unsigned long long) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned long long nwaves = ((unsigned long long) gridDim.x * blockDim.x) >> 6;
    unsigned long long c = 0;
    for (; wave < nbig; wave += nwaves) c += tc_intersect<64>(node_idx, r_node_idx, big[wave], lane);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if (lane == 0 && c) atomicAdd(total, c);
}

// emitted form (no reverse CSR): thread per slot pair is hopeless on hubs, so one wave per
// slot (v,u) strides over the tail and binary searches u in the forward row of w.
__global__ void __launch_bounds__(TC_THREADS)
tc_forward_only_kernel(const int32_t* __restrict__ begin, const int32_t* __restrict__ node_idx,
                       int64_t V, int64_t E, int64_t nlocal, int part, int nparts, unsigned long long* __restrict__ total) {
    const int lane = threadIdx.x & 63;
    int64_t i = ((int64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t) gridDim.x * blockDim.x) >> 6;
    unsigned long long c = 0;
    for (; i < nlocal; i += nwaves) {
        const int64_t e = tc_slot_of(i, part, nparts);
        if (e >= E) continue;
        int64_t lo = 0, hi = V;
        while (hi - lo > 1) {
            int64_t mid = (lo + hi) >> 1;
            if ((int64_t) begin[mid] <= e) lo = mid; else hi = mid;
        }
        const int32_t v = (int32_t) lo, u = node_idx[e];
        if (u <= v) continue;
        const int32_t te = begin[v + 1];
        const int32_t tb = tc_lower_bound(node_idx, (int32_t) e, te, u + 1);
        for (int32_t j = tb + lane; j < te; j += 64) {
            const int32_t w = node_idx[j];
            c += tc_contains(node_idx, begin[w], begin[w + 1], u) ? 1 : 0;   // w.HasEdgeTo(u)
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if (lane == 0 && c) atomicAdd(total, c);
}

// ------------------------------------------------------------------ oriented counting with the list in LDS
// On the degree-ordered copy of a symmetric simple graph only the upper lists matter: Up(x) = neighbours of x
// with a larger id (a suffix of the sorted row), and
//     T = sum over v, over u in Up(v), of | {w in Up(v), w > u}  intersect  Up(u) |.
// One wave per vertex v: Up(v) is staged in the wave's LDS slice once and reused for every u.  The u's are
// taken 64 at a time, one per lane: short intersections are walked by the lane alone, long ones by the whole
// wave (lanes stream Up(u), coalesced, and binary-search the staged tail above u's position -- or, when Up(u)
// is the much longer list, stream the staged tail and search Up(u) in memory).  Vertices are claimed from a counter in
// groups of 64 spread over the id range; the multi-GPU form deals the vertices to `nparts` parts round-robin.
#define TCO_WAVES 4
#define TCO_CLAIM 8     // work items per dequeue
#define TCO_ALONE 4     // a lane walks a side of up to this many entries by itself (measured flat from 0 to 8; 48: +25 %)
#define TCO_RATIO 4     // stream Up(u) and search the staged tail while |Up(u)| <= TCO_RATIO * |tail| (2: +15 %, 16: +7 %)
#define TCO_PIECES 4    // 256-byte pieces of Up(u) a wave keeps in flight
#define TCO_HUB_TAIL 4 // against a hub u the tail of Up(v) is the side to walk while it is at most this many times |Up(u)| (2 .. 64: the same; 1: +30 %)
#define TCO_HUB_MAX 131072   // hubs of the bit matrix (2 GiB at most; half of that up to 2^24 vertices)
#define TCO_CAP 1024    // upper-list entries staged per wave (4 KiB; 4 waves: 16 KiB of LDS per workgroup, 8 workgroups per CU)

__global__ void tc_up_begin_kernel(const int32_t* __restrict__ begin, const int32_t* __restrict__ node_idx, int64_t V,
                                   int32_t* __restrict__ up_begin) {
    int64_t v = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; v < V; v += stride) up_begin[v] = tc_lower_bound(node_idx, begin[v], begin[v + 1], (int32_t) v + 1);
}

// The hubs' adjacency as bits.  On the degree-ordered copy the H highest ids are the H vertices of highest degree, and every
// upper neighbour of a hub is a hub: Up(u) of hub u is row u of an H x H bit matrix.  "is w in Up(u)" is then ONE load
// instead of a binary search over Up(u) in memory (~10 dependent loads on as many lines), and a slot whose u is a hub is
// cheapest from the side of v's tail -- a few entries, the hubs above u -- however long Up(u) is.  (RMAT-20 symmetrised,
// offline: 83 % of the oriented edges end in one of the V/64 highest vertices, and they carry 92 % of the entries the
// kernel would otherwise stream or search.)  One wave per hub row.
__global__ void tc_hub_bits_kernel(const int32_t* __restrict__ begin, const int32_t* __restrict__ node_idx, const int32_t* __restrict__ up_begin,
                                   int64_t base, int64_t H, uint32_t* __restrict__ bits) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t) blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= H) return;
    const int64_t u = base + row;
    uint32_t* dst = bits + row * (H >> 5);
    for (int32_t p = up_begin[u] + lane; p < begin[u + 1]; p += 64) {
        const int64_t w = node_idx[p] - base;   // > row
        atomicOr(&dst[w >> 5], 1u << (w & 31));
    }
}

__device__ __forceinline__ int32_t tco_lds_lower_bound(const int32_t* a, int32_t lo, int32_t hi, int32_t x) {
    while (lo < hi) {
        const int32_t mid = lo + ((hi - lo) >> 1);
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// groups of 64 slots per vertex (a vertex with da upper neighbours has da - 1 slots that can close a triangle)
__global__ void tc_groups_kernel(const int32_t* __restrict__ begin, const int32_t* __restrict__ up_begin, int64_t V,
                                 int32_t* __restrict__ groups) {
    int64_t v = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; v < V; v += stride) {
        const int32_t da = begin[v + 1] - up_begin[v];
        groups[v] = da >= 2 ? (da - 1 + 63) / 64 : 0;
    }
}

// Work item = (vertex v, group g of 64 of its slots): a hub-like vertex (thousands of upper neighbours, each
// with thousands of its own) is spread over many waves.  grp_off[V+1] = exclusive scan of the group counts.
__global__ void __launch_bounds__(TCO_WAVES * 64)
tc_oriented_kernel(const int32_t* __restrict__ begin, const int32_t* __restrict__ node_idx, const int32_t* __restrict__ up_begin,
                   const int32_t* __restrict__ grp_off, int64_t V, int part, int nparts, int alone_max, int ratio,
                   unsigned long long* __restrict__ next_claim, unsigned long long* __restrict__ total,
                   const uint32_t* __restrict__ hub_bits, int64_t hub_base /* V: no hubs */, int hub_words /* per row */, int hub_tail) {
    __shared__ int32_t s_up[TCO_WAVES][TCO_CAP];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    int32_t* A = s_up[wv];
    const int64_t G = grp_off[V];
    const int64_t Q = G > part ? (G - part + nparts - 1) / nparts : 0;   // items part, part + nparts, ...
    unsigned long long c = 0;
    for (;;) {
        unsigned long long b = 0;
        if (lane == 0) b = atomicAdd(next_claim, (unsigned long long) TCO_CLAIM);
        b = __shfl(b, 0, 64);
        if ((int64_t) b >= Q) break;
        int64_t v = -1;
        for (int64_t q = (int64_t) b; q < (int64_t) b + TCO_CLAIM && q < Q; q++) {
            const int64_t item = q * nparts + part;
            if (v < 0 || (int64_t) grp_off[v + 1] <= item) {   // vertex of the item: last v with grp_off[v] <= item
                int64_t lo = 0, hi = V;
                while (hi - lo > 1) {
                    const int64_t mid = (lo + hi) >> 1;
                    if ((int64_t) grp_off[mid] <= item) lo = mid; else hi = mid;
                }
                v = lo;
            }
            const int32_t base = (int32_t) (item - grp_off[v]) * 64;   // first slot of the group
            const int32_t ab = up_begin[v] + base, ae = begin[v + 1];
            const int32_t da = ae - ab;                                  // Up(v) from the group's first slot on
            const bool in_lds = da <= TCO_CAP;
            if (in_lds) {
                for (int32_t k = lane; k < da; k += 64) A[k] = node_idx[ab + k];
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0xc07f);     // the wave's own LDS writes are visible to all its lanes
                __builtin_amdgcn_wave_barrier();
            }
            // one slot per lane.  A lane whose shorter side is short walks it alone (probing the staged list or
            // Up(u)); the others are taken one after the other by the whole wave, lanes striding over the side
            // that is cheaper to stream.
            const int32_t i = lane;
            const bool act = i + 1 < da;
            int32_t bb = 0, be = 0, u = 0;
            if (act) {
                u = in_lds ? A[i] : node_idx[ab + i];
                bb = up_begin[u];
                be = begin[u + 1];
            }
            const int32_t db = be - bb, ta = act ? da - (i + 1) : 0;
            // the side to walk: the tail of Up(v) above u, or Up(u).  Against a hub u a tail entry costs one bit probe,
            // so the tail is the side unless it is far the longer one
            const bool hubu = act && u >= hub_base;
            const bool tail_side = hubu ? ta <= hub_tail * db : ta < db;
            const uint32_t* hrow = hub_bits + (hubu ? (int64_t) (u - hub_base) * hub_words : 0);
            const int32_t shorter = db == 0 || ta == 0 ? 0 : (tail_side ? ta : db);
            if (act && shorter > 0 && shorter <= alone_max) {
                if (tail_side && hubu) {
                    for (int32_t p = i + 1; p < da; p++) {
                        const int64_t w = (in_lds ? A[p] : node_idx[ab + p]) - hub_base;
                        c += (hrow[w >> 5] >> (w & 31)) & 1u;
                    }
                } else if (!tail_side) {
                    for (int32_t p = bb; p < be; p++) {
                        const int32_t w = node_idx[p];
                        if (in_lds) {
                            const int32_t f = tco_lds_lower_bound(A, i + 1, da, w);
                            c += (f < da && A[f] == w) ? 1 : 0;
                        } else c += tc_contains(node_idx, ab + i + 1, ae, w) ? 1 : 0;
                    }
                } else {
                    for (int32_t p = i + 1; p < da; p++)
                        c += tc_contains(node_idx, bb, be, in_lds ? A[p] : node_idx[ab + p]) ? 1 : 0;
                }
            }
            unsigned long long m = __ballot(act && shorter > alone_max);
            while (m) {
                const int src = __ffsll((long long) m) - 1;
                m &= m - 1;
                const int32_t sbb = __shfl(bb, src, 64), sbe = __shfl(be, src, 64);
                const int32_t sdb = sbe - sbb, sta = da - (src + 1);
                const int32_t su = __shfl(u, src, 64);
                if (su >= hub_base && sta <= hub_tail * sdb) {   // a hub: the lanes stride over the tail, one bit probe each
                    const uint32_t* srow = hub_bits + (int64_t) (su - hub_base) * hub_words;
                    for (int32_t p = src + 1 + lane; p < da; p += 64) {
                        const int64_t w = (in_lds ? A[p] : node_idx[ab + p]) - hub_base;
                        c += (srow[w >> 5] >> (w & 31)) & 1u;
                    }
                } else if (in_lds && sdb <= ratio * sta) {   // stream Up(u), search the staged tail
                    // TCO_PIECES 256-byte pieces of Up(u) in flight per wave before the searches start: with one piece per
                    // step the next load waited behind ten dependent LDS probes (282 -> 264 ms on symmetrised RMAT-24).
                    // Measured and not kept (profiles/round3_tc_*: the kernel fetches 0.68 TB per call at 2.4 TB/s with the
                    // VALU 10 % and the LDS 1 % busy): 8 pieces (274 ms), the next slot's first pieces requested while the
                    // current slot is searched (266 ms), Up(v) as an LDS hash set instead of a sorted list (one or two reads
                    // per probe, but 8 KiB per wave: 20 waves per CU instead of 32 -> 593 ms), every edge handled at the end
                    // with the longer list (0.44 TB fetched, 308 ms), the kilobyte of a step as one 16-byte load per lane instead of four
                    // 4-byte ones (what halved hop_dist's bottom-up levels: 275 ms here -- the lists are short, and four entries
                    // per lane leave most lanes without a search), the neighbour's two row-header words as one 8-byte gather from a
                    // packed {first upper neighbour, row end} array instead of two 4-byte gathers (286 ms).  What limits it is the rate at which the memory system
                    // delivers these scattered 0.25-2.5 KB list reads, not the wave's own latency chain.
                    for (int32_t p = sbb + lane; p < sbe; p += TCO_PIECES * 64) {
                        int32_t w[TCO_PIECES];
#pragma unroll
                        for (int k = 0; k < TCO_PIECES; k++) w[k] = p + 64 * k < sbe ? node_idx[p + 64 * k] : -1;
#pragma unroll
                        for (int k = 0; k < TCO_PIECES; k++) {
                            if (w[k] < 0) continue;
                            const int32_t f = tco_lds_lower_bound(A, src + 1, da, w[k]);
                            c += (f < da && A[f] == w[k]) ? 1 : 0;
                        }
                    }
                } else {                               // stream the tail of Up(v), search Up(u) in memory
                    for (int32_t p = src + 1 + lane; p < da; p += 64)
                        c += tc_contains(node_idx, sbb, sbe, in_lds ? A[p] : node_idx[ab + p]) ? 1 : 0;
                }
            }
            __builtin_amdgcn_wave_barrier();            // all lanes are done with A before it is overwritten
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if (lane == 0 && c) atomicAdd(total, c);
}

// The staged-list kernel is the default on the degree-ordered copy (GMX_TC_NO_LDS=1 selects the slot kernels that
// search both lists in memory).  RMAT-24 symmetrised: 282 ms against 585 ms.  Its history is a lesson in
// occupancy: 1.8 s with one wave walking a whole vertex, 580 ms with vertices interleaved over the waves,
// 604 ms with 64-slot work items and 12 KiB of LDS per wave (12 waves per CU), 298 ms with 4 KiB per wave
// (32 waves per CU; lists over 1024 entries are searched in memory), 282 ms after tuning the two thresholds.
static bool tc_use_lds() { return !getenv("GMX_TC_NO_LDS"); }

// ------------------------------------------------------------------ degree-oriented copy
static int tc_grid(int64_t n) {
    int64_t b = (n + TC_THREADS - 1) / TC_THREADS;
    return (int) (b < 1 ? 1 : b > 256 * 16 ? 256 * 16 : b);
}

// keys = (row << 32 | col) of the forward CSR, row-major: symmetric <=> both CSRs are the same arrays;
// simple <=> no key with row == col and no two equal neighbours
__global__ void tc_sym_check_kernel(const uint64_t* __restrict__ keys, const int32_t* __restrict__ node_idx,
                                    const int32_t* __restrict__ r_node_idx, const int32_t* __restrict__ begin,
                                    const int32_t* __restrict__ r_begin, int64_t V, int64_t E, int* __restrict__ bad) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    bool b = false;
    for (; i < E || i <= V; i += stride) {
        if (i <= V && begin[i] != r_begin[i]) b = true;
        if (i < E) {
            const uint64_t k = keys[i];
            if ((uint32_t) (k >> 32) == (uint32_t) k) b = true;
            if (i > 0 && keys[i - 1] == k) b = true;
            if (node_idx[i] != r_node_idx[i]) b = true;
        }
    }
    if (__ballot(b) && (threadIdx.x & 63) == 0) atomicOr(bad, 1);
}

__global__ void tc_degkey_kernel(const int32_t* __restrict__ begin, int64_t V, uint32_t* __restrict__ key, int32_t* __restrict__ id) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < V; i += stride) {
        key[i] = (uint32_t) (begin[i + 1] - begin[i]);   // ascending degree
        id[i] = (int32_t) i;
    }
}

__global__ void tc_invert_kernel(const int32_t* __restrict__ order, int64_t V, int32_t* __restrict__ perm) {
    int64_t j = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; j < V; j += stride) perm[order[j]] = (int32_t) j;
}

// The graph to count on: g itself, or (symmetric simple graphs) its cached degree-ordered copy.
static int tc_counting_graph(gmx_graph* g, gmx_graph** out, bool* oriented) {
    *out = g;
    *oriented = false;
    if (!g->has_reverse || g->E == 0 || getenv("GMX_TC_NO_ORIENT")) return GMX_OK;
    if (g->tc_sym_state < 0) {
        hipStream_t s = 0;
        dbuf<uint64_t> keys, alt;
        dbuf<int> bad;
        GMX_CHECK(keys.alloc((size_t) g->E));
        GMX_CHECK(bad.alloc(1));
        GMX_HIP(hipMemsetAsync(bad.p, 0, sizeof(int), s));
        GMX_CHECK(gmx_keys_from_csr(g->begin.p, g->node_idx.p, g->V, g->E, false, nullptr, keys.p, s));
        hipLaunchKernelGGL(tc_sym_check_kernel, dim3(tc_grid(g->E > g->V ? g->E : g->V + 1)), dim3(TC_THREADS), 0, s,
                           (const uint64_t*) keys.p, g->node_idx.p, g->r_node_idx.p, g->begin.p, g->r_begin.p, g->V, g->E, bad.p);
        int hb = 1;
        GMX_HIP(hipMemcpy(&hb, bad.p, sizeof(int), hipMemcpyDeviceToHost));
        g->tc_sym_state = hb ? 0 : 1;
        if (g->tc_sym_state == 1) {
            dbuf<uint32_t> key, key2;
            dbuf<int32_t> id, order, perm;
            GMX_CHECK(key.alloc((size_t) g->V));
            GMX_CHECK(key2.alloc((size_t) g->V));
            GMX_CHECK(id.alloc((size_t) g->V));
            GMX_CHECK(order.alloc((size_t) g->V));
            GMX_CHECK(perm.alloc((size_t) g->V));
            hipLaunchKernelGGL(tc_degkey_kernel, dim3(tc_grid(g->V)), dim3(TC_THREADS), 0, s, g->begin.p, g->V, key.p, id.p);
            size_t tb = 0;
            GMX_HIP(rocprim::radix_sort_pairs(nullptr, tb, key.p, key2.p, id.p, order.p, (size_t) g->V, 0u, 32u, s));
            dbuf<char> tmp;
            GMX_CHECK(tmp.alloc(tb));
            GMX_HIP(rocprim::radix_sort_pairs((void*) tmp.p, tb, key.p, key2.p, id.p, order.p, (size_t) g->V, 0u, 32u, s));
            hipLaunchKernelGGL(tc_invert_kernel, dim3(tc_grid(g->V)), dim3(TC_THREADS), 0, s, (const int32_t*) order.p, g->V, perm.p);
            gmx_graph* o = new gmx_graph();
            o->V = g->V;
            o->E = g->E;
            o->device = g->device;
            int st = GMX_OK;
            if ((st = alt.alloc((size_t) g->E)) || (st = o->begin.alloc((size_t) g->V + 1)) || (st = o->node_idx.alloc((size_t) g->E)) ||
                (st = gmx_keys_from_csr(g->begin.p, g->node_idx.p, g->V, g->E, false, perm.p, keys.p, s)) ||
                (st = gmx_csr_from_keys(keys.p, alt.p, g->V, g->E, o->begin.p, o->node_idx.p, s))) {
                delete o;
                return st;
            }
            // first upper neighbour of every row (kept in the copy's otherwise unused r_begin)
            if ((st = o->r_begin.alloc((size_t) g->V + 1))) {
                delete o;
                return st;
            }
            hipLaunchKernelGGL(tc_up_begin_kernel, dim3(tc_grid(g->V)), dim3(TC_THREADS), 0, s, o->begin.p, o->node_idx.p, g->V, o->r_begin.p);
            // the hubs' adjacency as a bit matrix (GMX_TC_HUBS=<n> sets their number, 0 = none: development option)
            {
                int64_t H = g->V > (1LL << 24) ? TCO_HUB_MAX : TCO_HUB_MAX / 2;   // (RMAT-24: 65536 and 131072 hubs measured the same, 32768 +30 %)
                if (H > g->V) H = g->V;
                if (const char* e = getenv("GMX_TC_HUBS")) H = atoll(e) < g->V ? atoll(e) : g->V;
                H &= ~(int64_t) 63;
                if (H > 0) {
                    const size_t words = (size_t) H * (size_t) (H >> 5);
                    if ((st = o->tc_hub_bits.alloc(words))) {
                        delete o;
                        return st;
                    }
                    GMX_HIP(hipMemsetAsync(o->tc_hub_bits.p, 0, words * sizeof(uint32_t), s));
                    hipLaunchKernelGGL(tc_hub_bits_kernel, dim3((unsigned) ((H + 3) / 4)), dim3(256), 0, s, (const int32_t*) o->begin.p,
                                       (const int32_t*) o->node_idx.p, (const int32_t*) o->r_begin.p, g->V - H, H, o->tc_hub_bits.p);
                    o->tc_hubs = H;
                }
            }
            // work items of the staged-list kernel: exclusive scan of the 64-slot groups per vertex (kept in the
            // copy's otherwise unused r_node_idx)
            {
                dbuf<int32_t> groups;
                if ((st = groups.alloc((size_t) g->V + 1)) || (st = o->r_node_idx.alloc((size_t) g->V + 1))) {
                    delete o;
                    return st;
                }
                GMX_HIP(hipMemsetAsync(groups.p + g->V, 0, sizeof(int32_t), s));
                hipLaunchKernelGGL(tc_groups_kernel, dim3(tc_grid(g->V)), dim3(TC_THREADS), 0, s, o->begin.p, (const int32_t*) o->r_begin.p, g->V, groups.p);
                size_t tb2 = 0;
                GMX_HIP(rocprim::exclusive_scan(nullptr, tb2, groups.p, o->r_node_idx.p, 0, (size_t) g->V + 1, rocprim::plus<int32_t>(), s));
                dbuf<char> tmp2;
                GMX_CHECK(tmp2.alloc(tb2));
                GMX_HIP(rocprim::exclusive_scan((void*) tmp2.p, tb2, groups.p, o->r_node_idx.p, 0, (size_t) g->V + 1, rocprim::plus<int32_t>(), s));
                GMX_HIP(hipStreamSynchronize(s));
            }
            GMX_HIP(hipStreamSynchronize(s));
            g->tc_oriented = o;
        }
    }
    if (g->tc_sym_state == 1 && g->tc_oriented) {
        *out = g->tc_oriented;
        *oriented = true;
    }
    return GMX_OK;
}

static int tc_count_part(gmx_graph_t* g, int part, int nparts, bool common_nbr_form, int64_t* count, gmx_stats_t* stats);

extern "C" int gmx_triangle_counting(gmx_graph_t* g, int64_t* count, gmx_stats_t* stats) {
    return tc_count_part(g, 0, 1, false, count, stats);
}

extern "C" int gmx_triangle_counting_part(gmx_graph_t* g, int part, int nparts, int64_t* count, gmx_stats_t* stats) {
    return tc_count_part(g, part, nparts, false, count, stats);
}

// triangle counting written with the common-neighbour iterator (gm_common_neighbor_iter.cc:21-44):
//   Foreach(v: G.Nodes) Foreach(u: v.Nbrs)(u > v) Foreach(w: v.CommonNbrs(u))(w > u) T += 1;
// w walks the slots of v's row (with multiplicity) and passes when it occurs in the FORWARD row of u -- the emitted
// triangle_counting.gm asks for w -> u instead (HasEdgeTo), i.e. for the in-row of u.  Same kernels, other rows.
extern "C" int gmx_triangle_counting_cn(gmx_graph_t* g, int64_t* count, gmx_stats_t* stats) {
    return tc_count_part(g, 0, 1, true, count, stats);
}

static int tc_count_part(gmx_graph_t* g, int part, int nparts, bool common_nbr_form, int64_t* count, gmx_stats_t* stats) {
    GMX_REQUIRE(g && count, "NULL argument");
    GMX_REQUIRE(nparts >= 1 && part >= 0 && part < nparts, "bad part %d / nparts %d", part, nparts);
    if (stats) memset(stats, 0, sizeof(*stats));
    *count = 0;
    if (g->E == 0) return GMX_OK;
    // graph preprocessing (cached on the graph like the reverse CSR; outside the timed region)
    gmx_graph* cg = nullptr;
    bool oriented = false;
    GMX_CHECK(tc_counting_graph(g, &cg, &oriented));
    const int32_t* rbeg = oriented ? cg->begin.p : g->r_begin.p;         // symmetric: the in-rows are the out-rows
    const int32_t* ridx = oriented ? cg->node_idx.p : g->r_node_idx.p;
    if (common_nbr_form && !oriented) {   // membership in the forward row of u
        rbeg = g->begin.p;
        ridx = g->node_idx.p;
    }
    const bool have_rows = g->has_reverse || oriented || common_nbr_form;
    g = cg;
    // local slot indices of this part: whole deal blocks (slots past E are skipped in the kernels)
    const int64_t deal = (int64_t) 1 << TC_DEAL_SHIFT;
    const int64_t nblocks = (g->E + deal - 1) / deal;
    const int64_t nlocal = ((nblocks - part + nparts - 1) / nparts) * deal;
    if (nlocal <= 0) return GMX_OK;
    dbuf<unsigned long long> ctr;   // [0] total, [1] nbig
    GMX_CHECK(ctr.alloc(2));
    GMX_HIP(hipMemset(ctr.p, 0, 2 * sizeof(unsigned long long)));
    hipEvent_t ev0, ev1;
    GMX_HIP(hipEventCreate(&ev0));
    GMX_HIP(hipEventCreate(&ev1));
    GMX_HIP(hipEventRecord(ev0, 0));
    int64_t blocks = (nlocal + TC_THREADS - 1) / TC_THREADS;
    if (oriented && tc_use_lds()) {
        hipLaunchKernelGGL(tc_oriented_kernel, dim3(256 * 8), dim3(TCO_WAVES * 64), 0, 0, g->begin.p, g->node_idx.p,
                           (const int32_t*) g->r_begin.p, (const int32_t*) g->r_node_idx.p, g->V, part, nparts,
                           getenv("GMX_TC_ALONE") ? atoi(getenv("GMX_TC_ALONE")) : TCO_ALONE, getenv("GMX_TC_RATIO") ? atoi(getenv("GMX_TC_RATIO")) : TCO_RATIO, ctr.p + 1, ctr.p, (const uint32_t*) g->tc_hub_bits.p, g->V - g->tc_hubs, (int) (g->tc_hubs >> 5),
                           getenv("GMX_TC_HUB_TAIL") ? atoi(getenv("GMX_TC_HUB_TAIL")) : TCO_HUB_TAIL);
        GMX_HIP(hipGetLastError());
    } else if (have_rows) {
        dbuf<tc_pair> big;
        GMX_CHECK(big.alloc((size_t) nlocal));
        if (blocks > 256 * 64) blocks = 256 * 64;
        hipLaunchKernelGGL(tc_slots_kernel, dim3((unsigned) blocks), dim3(TC_THREADS), 0, 0,
                           g->begin.p, g->node_idx.p, rbeg, ridx, g->V, g->E, nlocal, part, nparts, big.p, ctr.p + 1, ctr.p);
        GMX_HIP(hipGetLastError());
        unsigned long long h[2];
        GMX_HIP(hipMemcpy(h, ctr.p, sizeof(h), hipMemcpyDeviceToHost));
        if (h[1]) {
            int64_t wb = (int64_t) ((h[1] * 64 + TC_THREADS - 1) / TC_THREADS);
            if (wb > 256 * 64) wb = 256 * 64;
            hipLaunchKernelGGL(tc_big_kernel, dim3((unsigned) wb), dim3(TC_THREADS), 0, 0,
                               g->node_idx.p, ridx, big.p, h[1], ctr.p);
            GMX_HIP(hipGetLastError());
        }
        GMX_HIP(hipDeviceSynchronize());
    } else {
        int64_t wb = (nlocal * 64 + TC_THREADS - 1) / TC_THREADS;
        if (wb > 256 * 64) wb = 256 * 64;
        hipLaunchKernelGGL(tc_forward_only_kernel, dim3((unsigned) wb), dim3(TC_THREADS), 0, 0,
                           g->begin.p, g->node_idx.p, g->V, g->E, nlocal, part, nparts, ctr.p);
        GMX_HIP(hipGetLastError());
    }
    GMX_HIP(hipEventRecord(ev1, 0));
    GMX_HIP(hipEventSynchronize(ev1));
    float ms = 0;
    (void) hipEventElapsedTime(&ms, ev0, ev1);
    (void) hipEventDestroy(ev0);
    (void) hipEventDestroy(ev1);
    unsigned long long h = 0;
    GMX_HIP(hipMemcpy(&h, ctr.p, sizeof(h), hipMemcpyDeviceToHost));
    *count = (int64_t) h;
    if (stats) {
        stats->iterations = 1;
        stats->kernel_ms = ms;
    }
    return GMX_OK;
}

// ------------------------------------------------------------------ common-neighbour iterator
// gm_common_neighbor_iter(G, s, d) (gm_common_neighbor_iter.cc:21-44; Foreach(u: s.CommonNbrs(d)) <=>
// Foreach(u: s.Nbrs)(d.isNbr(u)), gm_common_neighbor_iter.h:11-17): the slots of s's row, in order and with their
// multiplicity, whose value occurs in d's row.  Rows are semi-sorted.  One wave per pair: 64 slots of s's row at a
// time, each lane looks its value up in d's row, a ballot keeps the order.
__global__ void __launch_bounds__(TC_THREADS)
common_nbr_count_kernel(const int32_t* __restrict__ begin, const int32_t* __restrict__ node_idx,
                        const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int64_t npairs, int64_t* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    int64_t wave = ((int64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t) gridDim.x * blockDim.x) >> 6;
    for (; wave < npairs; wave += nwaves) {
        const int32_t s = src[wave], d = dst[wave];
        const int32_t sb = begin[s], se = begin[s + 1], db = begin[d], de = begin[d + 1];
        unsigned long long c = 0;
        for (int32_t j = sb + lane; j < se; j += 64) c += tc_contains(node_idx, db, de, node_idx[j]) ? 1 : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
        if (lane == 0) counts[wave] = (int64_t) c;
    }
}

__global__ void __launch_bounds__(64)
common_nbr_list_kernel(const int32_t* __restrict__ begin, const int32_t* __restrict__ node_idx, int32_t s, int32_t d,
                       int32_t* __restrict__ out, int64_t cap, int64_t* __restrict__ n_out) {
    const int lane = threadIdx.x & 63;
    const int32_t sb = begin[s], se = begin[s + 1], db = begin[d], de = begin[d + 1];
    int64_t n = 0;
    for (int32_t j0 = sb; j0 < se; j0 += 64) {
        const int32_t j = j0 + lane;
        const int32_t t = j < se ? node_idx[j] : 0;
        const bool hit = j < se && tc_contains(node_idx, db, de, t);
        const unsigned long long m = __ballot(hit);
        const int64_t at = n + __builtin_popcountll(m & ((1ull << lane) - 1));
        if (hit && at < cap) out[at] = t;
        n += __builtin_popcountll(m);
    }
    if (lane == 0) *n_out = n;
}

extern "C" int gmx_common_nbr_counts(gmx_graph_t* g, const gmx_node_t* src, const gmx_node_t* dst, int64_t npairs, int64_t* counts) {
    GMX_REQUIRE(g && counts && ((src && dst) || npairs == 0) && npairs >= 0, "bad argument");
    if (npairs == 0) return GMX_OK;
    for (int64_t i = 0; i < npairs; i++)
        GMX_REQUIRE(src[i] >= 0 && src[i] < g->V && dst[i] >= 0 && dst[i] < g->V, "pair %lld: vertex out of range", (long long) i);
    dbuf<int32_t> ds, dd;
    dbuf<int64_t> dc;
    GMX_CHECK(ds.alloc((size_t) npairs));
    GMX_CHECK(dd.alloc((size_t) npairs));
    GMX_CHECK(dc.alloc((size_t) npairs));
    GMX_HIP(hipMemcpy(ds.p, src, sizeof(int32_t) * (size_t) npairs, hipMemcpyHostToDevice));
    GMX_HIP(hipMemcpy(dd.p, dst, sizeof(int32_t) * (size_t) npairs, hipMemcpyHostToDevice));
    int64_t blocks = (npairs * 64 + TC_THREADS - 1) / TC_THREADS;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(common_nbr_count_kernel, dim3((unsigned) blocks), dim3(TC_THREADS), 0, 0, g->begin.p, g->node_idx.p,
                       (const int32_t*) ds.p, (const int32_t*) dd.p, npairs, dc.p);
    GMX_HIP(hipGetLastError());
    GMX_HIP(hipMemcpy(counts, dc.p, sizeof(int64_t) * (size_t) npairs, hipMemcpyDeviceToHost));
    return GMX_OK;
}

extern "C" int gmx_common_nbrs(gmx_graph_t* g, gmx_node_t s, gmx_node_t d, gmx_node_t* out, int64_t cap, int64_t* n) {
    GMX_REQUIRE(g && n && (out || cap == 0) && cap >= 0, "bad argument");
    GMX_REQUIRE(s >= 0 && s < g->V && d >= 0 && d < g->V, "vertex out of range");
    dbuf<int32_t> dout;
    dbuf<int64_t> dn;
    GMX_CHECK(dout.alloc((size_t) (cap ? cap : 1)));
    GMX_CHECK(dn.alloc(1));
    hipLaunchKernelGGL(common_nbr_list_kernel, dim3(1), dim3(64), 0, 0, g->begin.p, g->node_idx.p, s, d, dout.p, cap, dn.p);
    GMX_HIP(hipGetLastError());
    GMX_HIP(hipMemcpy(n, dn.p, sizeof(int64_t), hipMemcpyDeviceToHost));
    const int64_t m = *n < cap ? *n : cap;
    if (m > 0) GMX_HIP(hipMemcpy(out, dout.p, sizeof(int32_t) * (size_t) m, hipMemcpyDeviceToHost));
    return GMX_OK;
}
