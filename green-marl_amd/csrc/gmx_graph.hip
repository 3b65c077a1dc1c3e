// gmx_graph.hip -- device graph construction: upload, RMAT generation, CSR build.
//
// Replaces (all relative to /root/reference/apps/output_cpp/gm_graph):
//   src/graph_gen.cc:159-287     create_RMAT_graph          -> rmat_attempts_kernel (+ host permutation)
//   src/gm_graph.cc:380-503      do_semi_sort               -> 64-bit key radix sort
//   src/gm_graph.cc:205-304      make_reverse_edges         -> transposed keys + the same sort
// gfx950 only.  FP contraction is OFF in this file: the RMAT generator must
// round exactly like the reference's x86-64 build (no FMA).
#pragma clang fp contract(off)

#include "gmx_internal.h"

#include <math.h>
#include <string.h>
#include <rocprim/rocprim.hpp>

// ------------------------------------------------------------------ errors
static thread_local char g_err[1024] = "";

void gmx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* gmx_last_error(void) { return g_err; }

// ------------------------------------------------------------------ device
// ------------------------------------------------------------------ workspace (see gmx_internal.h)
namespace {
struct ws_chunk { char* base; size_t cap; int device; };
struct ws_state {
    std::vector<ws_chunk> chunks;
    size_t cur = 0, off = 0;
};
ws_state g_ws;
constexpr size_t WS_ALIGN = 512;
}   // namespace

void* gmx_ws_alloc(size_t bytes) {
    bytes = (bytes + WS_ALIGN - 1) / WS_ALIGN * WS_ALIGN;
    int dev = 0;
    (void) hipGetDevice(&dev);
    for (;;) {
        if (g_ws.cur < g_ws.chunks.size()) {
            ws_chunk& c = g_ws.chunks[g_ws.cur];
            if (c.device == dev && g_ws.off + bytes <= c.cap) {
                void* p = c.base + g_ws.off;
                g_ws.off += bytes;
                return p;
            }
            g_ws.cur++;       // does not fit (or another device's chunk): try the next one
            g_ws.off = 0;
            continue;
        }
        // a new chunk: as large as everything held so far (256 MiB .. 32 GiB), so that a build's many buffers share a few
        size_t held = 0;
        for (const ws_chunk& c : g_ws.chunks) held += c.cap;
        size_t want = held < ((size_t) 256 << 20) ? ((size_t) 256 << 20) : held > ((size_t) 32 << 30) ? ((size_t) 32 << 30) : held;
        ws_chunk c{nullptr, bytes > want ? bytes : want, dev};
        hipError_t e = hipMalloc((void**) &c.base, c.cap);
        if (e != hipSuccess) {
            gmx_set_error("workspace: hipMalloc(%zu bytes) failed: %s", c.cap, hipGetErrorString(e));
            return nullptr;
        }
        g_ws.chunks.push_back(c);
    }
}
gmx_ws_mark gmx_ws_top() { return {g_ws.cur, g_ws.off}; }
void gmx_ws_rewind(gmx_ws_mark m) {
    g_ws.cur = m.chunk;
    g_ws.off = m.off;
}
extern "C" int gmx_workspace_release(void) {
    for (ws_chunk& c : g_ws.chunks) {
        (void) hipSetDevice(c.device);
        (void) hipFree(c.base);
    }
    g_ws.chunks.clear();
    g_ws.cur = g_ws.off = 0;
    return GMX_OK;
}
extern "C" int64_t gmx_workspace_bytes(void) {
    size_t t = 0;
    for (const ws_chunk& c : g_ws.chunks) t += c.cap;
    return (int64_t) t;
}

extern "C" int gmx_device_count(int* count) {
    GMX_REQUIRE(count, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        gmx_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return GMX_ERR_NODEVICE;
    }
    *count = n;
    return GMX_OK;
}

extern "C" int gmx_set_device(int device) {
    GMX_HIP(hipSetDevice(device));
    return GMX_OK;
}

extern "C" int gmx_device_info(gmx_device_info_t* info) {
    GMX_REQUIRE(info, "info is NULL");
    int dev = 0;
    GMX_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    GMX_HIP(hipGetDeviceProperties(&p, dev));
    memset(info, 0, sizeof(*info));
    strncpy(info->name, p.name, sizeof(info->name) - 1);
    strncpy(info->arch, p.gcnArchName, sizeof(info->arch) - 1);
    info->compute_units = p.multiProcessorCount;
    info->clock_mhz = p.clockRate / 1000;
    info->hbm_bytes = (int64_t) p.totalGlobalMem;
    info->l2_bytes = p.l2CacheSize;
    info->lds_bytes_per_cu = (int32_t) p.sharedMemPerMultiprocessor;
    return GMX_OK;
}

// Device copy bandwidth: the measured ceiling next to the 8 TB/s data-sheet figure (SURVEY.md 8d asks for both).
typedef float gmx_f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) copy_f32x4_kernel(const gmx_f32x4* __restrict__ src, gmx_f32x4* __restrict__ dst, int64_t n) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < n; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

extern "C" int gmx_copy_bandwidth(int64_t bytes, int iters, double* gbs) {
    GMX_REQUIRE(gbs && bytes >= 4096 && iters >= 1, "bad argument");
    const int64_t n = bytes / 16;
    dbuf<gmx_f32x4> a, b;
    GMX_CHECK(a.alloc((size_t) n));
    GMX_CHECK(b.alloc((size_t) n));
    GMX_HIP(hipMemset(a.p, 1, (size_t) n * 16));
    hipEvent_t e0, e1;
    GMX_HIP(hipEventCreate(&e0));
    GMX_HIP(hipEventCreate(&e1));
    hipLaunchKernelGGL(copy_f32x4_kernel, dim3(256 * 8), dim3(256), 0, 0, (const gmx_f32x4*) a.p, b.p, n);   // warm-up
    (void) hipEventRecord(e0, 0);
    for (int i = 0; i < iters; i++)
        hipLaunchKernelGGL(copy_f32x4_kernel, dim3(256 * 8), dim3(256), 0, 0, (const gmx_f32x4*) a.p, b.p, n);
    (void) hipEventRecord(e1, 0);
    hipError_t e = hipEventSynchronize(e1);
    float ms = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void) hipEventDestroy(e0);
    (void) hipEventDestroy(e1);
    GMX_HIP(e);
    *gbs = 2.0 * (double) n * 16.0 * iters / (ms * 1e-3) / 1e9;   // bytes read + bytes written
    return GMX_OK;
}

// ------------------------------------------------------------------ keys <-> CSR
// key = (row << 32 | col); 2^31 vertices max (node_t is int32).
#define KFC_CHUNK 16   // consecutive edges per thread: one binary search for the row of the first, then a walk along begin[]
__global__ void keys_from_csr_kernel(const int32_t* __restrict__ begin, const int32_t* __restrict__ idx,
                                     int64_t V, int64_t E, int transpose, const int32_t* __restrict__ perm,
                                     uint64_t* __restrict__ keys) {
    int64_t c = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x, nchunks = (E + KFC_CHUNK - 1) / KFC_CHUNK;
    for (; c < nchunks; c += stride) {
        const int64_t e0 = c * KFC_CHUNK, e1 = e0 + KFC_CHUNK < E ? e0 + KFC_CHUNK : E;
        int64_t lo = 0, hi = V;  // largest r with begin[r] <= e0
        while (hi - lo > 1) {
            int64_t mid = (lo + hi) >> 1;
            if ((int64_t) begin[mid] <= e0) lo = mid; else hi = mid;
        }
        int64_t row = lo, next = begin[row + 1];
        uint32_t r = perm ? (uint32_t) perm[row] : (uint32_t) row;
        for (int64_t e = e0; e < e1; e++) {
            while (e >= next) {   // (rows without edges are skipped)
                row++;
                next = begin[row + 1];
                r = perm ? (uint32_t) perm[row] : (uint32_t) row;
            }
            uint32_t col = (uint32_t) idx[e];
            if (perm) col = (uint32_t) perm[col];
            keys[e] = transpose ? (((uint64_t) col << 32) | r) : (((uint64_t) r << 32) | col);
        }
    }
}

__global__ void keys_from_edges_kernel(const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                       int64_t E, int transpose, const int32_t* __restrict__ perm,
                                       uint64_t* __restrict__ keys) {
    int64_t e = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; e < E; e += stride) {
        uint32_t r = (uint32_t) src[e], c = (uint32_t) dst[e];
        if (perm) { r = (uint32_t) perm[r]; c = (uint32_t) perm[c]; }
        keys[e] = transpose ? (((uint64_t) c << 32) | r) : (((uint64_t) r << 32) | c);
    }
}

__global__ void csr_extract_kernel(const uint64_t* __restrict__ keys, int64_t V, int64_t E,
                                   int32_t* __restrict__ begin, int32_t* __restrict__ idx) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (int64_t e = i; e < E; e += stride) idx[e] = (int32_t) (uint32_t) (keys[e] & 0xffffffffu);
    // begin[r] = first position whose row >= r
    for (int64_t r = i; r <= V; r += stride) {
        uint64_t target = (uint64_t) r << 32;
        int64_t lo = 0, hi = E;  // first index with keys[idx] >= target
        while (lo < hi) {
            int64_t mid = (lo + hi) >> 1;
            if (keys[mid] < target) lo = mid + 1; else hi = mid;
        }
        begin[r] = (int32_t) lo;
    }
}

static int grid_for(int64_t n, int block = 256, int max_blocks = 256 * 16) {
    int64_t b = (n + block - 1) / block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (int) b;
}

int gmx_keys_from_csr(const int32_t* begin, const int32_t* idx, int64_t V, int64_t E,
                      bool transpose, const int32_t* perm, uint64_t* keys, hipStream_t stream) {
    if (E == 0) return GMX_OK;
    hipLaunchKernelGGL(keys_from_csr_kernel, dim3(grid_for((E + KFC_CHUNK - 1) / KFC_CHUNK)), dim3(256), 0, stream,
                       begin, idx, V, E, transpose ? 1 : 0, perm, keys);
    GMX_HIP(hipGetLastError());
    return GMX_OK;
}

int gmx_keys_from_edges(const int32_t* src, const int32_t* dst, int64_t E, bool transpose,
                        const int32_t* perm, uint64_t* keys, hipStream_t stream) {
    if (E == 0) return GMX_OK;
    hipLaunchKernelGGL(keys_from_edges_kernel, dim3(grid_for(E)), dim3(256), 0, stream,
                       src, dst, E, transpose ? 1 : 0, perm, keys);
    GMX_HIP(hipGetLastError());
    return GMX_OK;
}

__global__ void iota_i32_kernel(int32_t* __restrict__ p, int64_t n) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = (int32_t) i;
}

// flag[0] |= 1 if some row of the CSR is not ascending
__global__ void rows_unsorted_kernel(const uint64_t* __restrict__ keys, int64_t E, int* __restrict__ flag) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    bool bad = false;
    for (; i < E; i += stride) bad |= keys[i] < keys[i - 1];   // keys are (row << 32 | col) in slot order
    if (bad) atomicOr(flag, 1);
}

int gmx_csr_from_keys(uint64_t* keys, uint64_t* keys_alt, int64_t V, int64_t E,
                      int32_t* begin, int32_t* idx, hipStream_t stream, int32_t* slots) {
    gmx_ws_scope ws;
    const uint64_t* sorted = keys;
    wbuf<int32_t> iota;
    if (slots) {
        GMX_CHECK(iota.alloc((size_t) (E ? E : 1)));
        hipLaunchKernelGGL(iota_i32_kernel, dim3(grid_for(E)), dim3(256), 0, stream, E > 1 ? iota.p : slots, E);
    }
    if (E > 1) {
        unsigned end_bit = 32 + (unsigned) gmx_bits_for(V);
        size_t tmp_bytes = 0;
        wbuf<char> tmp;
        if (slots) {   // stable: equal (row, col) keys keep their input order
            GMX_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, keys_alt, iota.p, slots, (size_t) E, 0u, end_bit, stream));
            GMX_CHECK(tmp.alloc(tmp_bytes));
            GMX_HIP(rocprim::radix_sort_pairs((void*) tmp.p, tmp_bytes, keys, keys_alt, iota.p, slots, (size_t) E, 0u, end_bit, stream));
            GMX_HIP(hipStreamSynchronize(stream));
            sorted = keys_alt;
        } else {
            rocprim::double_buffer<uint64_t> db(keys, keys_alt);
            GMX_HIP(rocprim::radix_sort_keys(nullptr, tmp_bytes, db, (size_t) E, 0u, end_bit, stream));
            GMX_CHECK(tmp.alloc(tmp_bytes));
            GMX_HIP(rocprim::radix_sort_keys((void*) tmp.p, tmp_bytes, db, (size_t) E, 0u, end_bit, stream));
            GMX_HIP(hipStreamSynchronize(stream));
            sorted = db.current();
        }
    }
    hipLaunchKernelGGL(csr_extract_kernel, dim3(grid_for(E > V ? E : V + 1)), dim3(256), 0, stream,
                       sorted, V, E, begin, idx);
    GMX_HIP(hipGetLastError());
    GMX_HIP(hipStreamSynchronize(stream));
    return GMX_OK;
}

// Build forward (and reverse) CSR of g from device keys (row<<32|col), consuming them.
// sorted_csr (optional): the uploaded CSR the keys were made from; when its rows turn out to be in order already it
// becomes the forward CSR as it is (no sort, identity e_idx2idx)
static int build_from_forward_keys(gmx_graph* g, wbuf<uint64_t>& keys, wbuf<uint64_t>& alt, bool want_reverse, bool keep_order = false,
                                   dbuf<int32_t>* up_begin = nullptr, dbuf<int32_t>* up_idx = nullptr) {
    hipStream_t s = 0;
    int32_t* slots = nullptr;
    bool in_order = false;
    if ((keep_order || (up_begin && up_idx)) && g->E > 1) {   // rows out of order + keep_order: remember where every sorted slot came from (e_idx2idx)
        dbuf<int> flag;
        GMX_CHECK(flag.alloc(1));
        GMX_HIP(hipMemsetAsync(flag.p, 0, sizeof(int), s));
        hipLaunchKernelGGL(rows_unsorted_kernel, dim3(grid_for(g->E)), dim3(256), 0, s, (const uint64_t*) keys.p, g->E, flag.p);
        int h = 0;
        GMX_HIP(hipMemcpy(&h, flag.p, sizeof(int), hipMemcpyDeviceToHost));
        if (h && keep_order) {
            GMX_CHECK(g->e_idx2idx.alloc((size_t) g->E));
            slots = g->e_idx2idx.p;
        }
        in_order = !h && up_begin && up_idx;
    }
    if (in_order) {   // (what a file written after a semi-sort looks like)
        g->begin.p = up_begin->take();
        g->begin.n = (size_t) g->V + 1;
        g->node_idx.p = up_idx->take();
        g->node_idx.n = (size_t) g->E;
    } else {
        GMX_CHECK(g->begin.alloc((size_t) g->V + 1));
        GMX_CHECK(g->node_idx.alloc((size_t) g->E));
        GMX_CHECK(gmx_csr_from_keys(keys.p, alt.p, g->V, g->E, g->begin.p, g->node_idx.p, s, slots));
    }
    if (want_reverse) {
        GMX_CHECK(g->r_begin.alloc((size_t) g->V + 1));
        GMX_CHECK(g->r_node_idx.alloc((size_t) g->E));
        GMX_CHECK(gmx_keys_from_csr(g->begin.p, g->node_idx.p, g->V, g->E, true, nullptr, keys.p, s));
        GMX_CHECK(gmx_csr_from_keys(keys.p, alt.p, g->V, g->E, g->r_begin.p, g->r_node_idx.p, s));
        g->has_reverse = true;
    }
    return GMX_OK;
}

void gmx_warm_modules() {
    static bool done = false;
    if (done) return;
    done = true;
    gmx_touch_pagerank();
    gmx_touch_pr_cold();
    gmx_touch_bfs();
}

static int check_sizes(int64_t V, int64_t E) {
    gmx_warm_modules();   // (every graph constructor passes through here)
    GMX_REQUIRE(V >= 0 && V < (1LL << 31) - 1, "V=%lld out of int32 node_t range", (long long) V);
    GMX_REQUIRE(E >= 0 && E < (1LL << 31), "E=%lld out of int32 edge_t range", (long long) E);
    return GMX_OK;
}

// ------------------------------------------------------------------ upload
extern "C" int gmx_graph_upload(const gmx_edge_t* begin, const gmx_node_t* node_idx,
                                const gmx_edge_t* r_begin, const gmx_node_t* r_node_idx,
                                int64_t V, int64_t E, uint32_t flags, gmx_graph_t** out) {
    gmx_ws_scope ws;
    GMX_REQUIRE(out, "out is NULL");
    *out = nullptr;
    GMX_CHECK(check_sizes(V, E));
    GMX_REQUIRE(begin && (node_idx || E == 0), "begin/node_idx is NULL");
    GMX_REQUIRE(begin[0] == 0 && (int64_t) begin[V] == E, "begin[0]=%d begin[V]=%d do not match E=%lld",
                begin[0], begin[V], (long long) E);
    gmx_graph* g = new gmx_graph();
    g->V = V;
    g->E = E;
    (void) hipGetDevice(&g->device);
    int st = GMX_OK;
    do {
        bool want_rev = !(flags & GMX_GRAPH_NO_REVERSE);
        if ((flags & GMX_GRAPH_SORT_ROWS) || (want_rev && !r_begin)) {
            // go through keys: sorts rows and/or builds the reverse CSR on the device
            dbuf<int32_t> tb, ti;
            wbuf<uint64_t> keys, alt;
            if ((st = tb.alloc((size_t) V + 1)) || (st = ti.alloc((size_t) E)) ||
                (st = keys.alloc((size_t) E)) || (st = alt.alloc((size_t) E))) break;
            if (hipMemcpy(tb.p, begin, sizeof(int32_t) * ((size_t) V + 1), hipMemcpyHostToDevice) != hipSuccess ||
                (E && hipMemcpy(ti.p, node_idx, sizeof(int32_t) * (size_t) E, hipMemcpyHostToDevice) != hipSuccess)) {
                gmx_set_error("H2D copy of CSR failed");
                st = GMX_ERR_HIP;
                break;
            }
            if ((st = gmx_keys_from_csr(tb.p, ti.p, V, E, false, nullptr, keys.p, 0))) break;
            if ((st = build_from_forward_keys(g, keys, alt, want_rev, (flags & GMX_GRAPH_SORT_ROWS) != 0, &tb, &ti))) break;
        } else {
            if ((st = g->begin.alloc((size_t) V + 1)) || (st = g->node_idx.alloc((size_t) E))) break;
            if (hipMemcpy(g->begin.p, begin, sizeof(int32_t) * ((size_t) V + 1), hipMemcpyHostToDevice) != hipSuccess ||
                (E && hipMemcpy(g->node_idx.p, node_idx, sizeof(int32_t) * (size_t) E, hipMemcpyHostToDevice) != hipSuccess)) {
                gmx_set_error("H2D copy of CSR failed");
                st = GMX_ERR_HIP;
                break;
            }
            if (want_rev) {
                if (!(r_node_idx || E == 0)) { gmx_set_error("r_node_idx is NULL"); st = GMX_ERR_ARG; break; }
                if ((st = g->r_begin.alloc((size_t) V + 1)) || (st = g->r_node_idx.alloc((size_t) E))) break;
                if (hipMemcpy(g->r_begin.p, r_begin, sizeof(int32_t) * ((size_t) V + 1), hipMemcpyHostToDevice) != hipSuccess ||
                    (E && hipMemcpy(g->r_node_idx.p, r_node_idx, sizeof(int32_t) * (size_t) E, hipMemcpyHostToDevice) != hipSuccess)) {
                    gmx_set_error("H2D copy of reverse CSR failed");
                    st = GMX_ERR_HIP;
                    break;
                }
                g->has_reverse = true;
            }
        }
    } while (0);
    if (st != GMX_OK) { delete g; return st; }
    *out = g;
    return GMX_OK;
}

extern "C" int gmx_graph_from_edges(const gmx_node_t* src, const gmx_node_t* dst,
                                    int64_t V, int64_t E, uint32_t flags, gmx_graph_t** out) {
    gmx_ws_scope ws;
    GMX_REQUIRE(out, "out is NULL");
    *out = nullptr;
    GMX_CHECK(check_sizes(V, E));
    GMX_REQUIRE((src && dst) || E == 0, "src/dst is NULL");
    for (int64_t i = 0; i < E; i++)
        GMX_REQUIRE(src[i] >= 0 && src[i] < V && dst[i] >= 0 && dst[i] < V, "edge %lld endpoint out of range", (long long) i);
    gmx_graph* g = new gmx_graph();
    g->V = V;
    g->E = E;
    (void) hipGetDevice(&g->device);
    int st = GMX_OK;
    do {
        wbuf<int32_t> ds, dd;
        wbuf<uint64_t> keys, alt;
        if ((st = ds.alloc((size_t) E)) || (st = dd.alloc((size_t) E)) ||
            (st = keys.alloc((size_t) E)) || (st = alt.alloc((size_t) E))) break;
        if (E && (hipMemcpy(ds.p, src, sizeof(int32_t) * (size_t) E, hipMemcpyHostToDevice) != hipSuccess ||
                  hipMemcpy(dd.p, dst, sizeof(int32_t) * (size_t) E, hipMemcpyHostToDevice) != hipSuccess)) {
            gmx_set_error("H2D copy of edge list failed");
            st = GMX_ERR_HIP;
            break;
        }
        if ((st = gmx_keys_from_edges(ds.p, dd.p, E, false, nullptr, keys.p, 0))) break;
        ds.release();
        dd.release();
        if ((st = build_from_forward_keys(g, keys, alt, !(flags & GMX_GRAPH_NO_REVERSE)))) break;
    } while (0);
    if (st != GMX_OK) { delete g; return st; }
    *out = g;
    return GMX_OK;
}

// ------------------------------------------------------------------ RMAT on device
// drand48: X' = (0x5DEECE66D X + 0xB) mod 2^48; value X'/2^48.  A jump by n steps
// is the affine map x -> a_n x + c_n; jumps by D*2^k steps are tabulated on the
// host so every thread reaches its first attempt in <= 48 multiply-adds.
#define LCG_A 0x5DEECE66DULL
#define LCG_C 0xBULL
#define LCG_MASK ((1ULL << 48) - 1)

struct lcg_map { uint64_t a, c; };
struct lcg_table { lcg_map pw[48]; };  // pw[k] = jump by (draws per attempt) * 2^k

static inline lcg_map lcg_then(lcg_map f, lcg_map g) {  // x -> g(f(x))
    lcg_map r;
    r.a = (g.a * f.a) & LCG_MASK;
    r.c = (g.a * f.c + g.c) & LCG_MASK;
    return r;
}

static lcg_map lcg_jump(uint64_t n) {  // n single steps
    lcg_map r = {1, 0}, p = {LCG_A, LCG_C};
    while (n) {
        if (n & 1) r = lcg_then(r, p);
        p = lcg_then(p, p);
        n >>= 1;
    }
    return r;
}

__device__ __forceinline__ double lcg_next(uint64_t& x) {
    x = (LCG_A * x + LCG_C) & LCG_MASK;
    return (double) x * (1.0 / 281474976710656.0);
}

// One thread = ATT_PER_THREAD consecutive attempts.  Attempt t consumes exactly
// D = 1 + 5*(SCALE-1) draws (graph_gen.cc:191-227), so its start state is X0
// advanced t*D steps.  Self loops are rejected by the reference (:231-235):
// here the slot is appended to bad_slots and refilled by a later round.
#define ATT_PER_THREAD 8
__global__ void rmat_attempts_kernel(lcg_table tab, uint64_t x0, int64_t t_begin, int64_t count,
                                     const int64_t* __restrict__ slot_map,
                                     int32_t N, int32_t SCALE, double a, double b, double c, double d,
                                     int32_t* __restrict__ src, int32_t* __restrict__ dst,
                                     int64_t* __restrict__ bad_slots, unsigned long long bad_cap,
                                     unsigned long long* __restrict__ bad_count) {
    int64_t i0 = ((int64_t) blockIdx.x * blockDim.x + threadIdx.x) * ATT_PER_THREAD;
    if (i0 >= count) return;
    // jump to attempt t_begin + i0
    uint64_t t = (uint64_t) (t_begin + i0);
    uint64_t x = x0;
    for (int k = 0; k < 48 && (t >> k); k++)
        if ((t >> k) & 1) x = (tab.pw[k].a * x + tab.pw[k].c) & LCG_MASK;

    int64_t iend = i0 + ATT_PER_THREAD < count ? i0 + ATT_PER_THREAD : count;
    for (int64_t i = i0; i < iend; i++) {
        int32_t u = 1, v = 1;
        int32_t step = N / 2;
        double av = a, bv = b, cv = c, dv = d;
        double p = lcg_next(x);
        if (p < av) {
        } else if (p < (av + bv)) {
            v += step;
        } else if (p < (av + bv + cv)) {
            u += step;
        } else {
            v += step;
            u += step;
        }
        for (int32_t j = 1; j < SCALE; j++) {
            step = step / 2;
            double var = 0.1;
            av *= 0.95 + var * lcg_next(x);
            bv *= 0.95 + var * lcg_next(x);
            cv *= 0.95 + var * lcg_next(x);
            dv *= 0.95 + var * lcg_next(x);
            double S = av + bv + cv + dv;
            av = av / S;
            bv = bv / S;
            cv = cv / S;
            dv = dv / S;
            p = lcg_next(x);
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
        int64_t slot = slot_map ? slot_map[i] : i;
        src[slot] = u - 1;
        dst[slot] = v - 1;
        if (u == v) {
            unsigned long long k = atomicAdd(bad_count, 1ULL);
            if (k < bad_cap) bad_slots[k] = slot;   // past the list's end only the count grows: the host reports the overflow
        }
    }
}

__global__ void apply_perm_kernel(int32_t* __restrict__ a, int64_t n, const int32_t* __restrict__ P) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < n; i += stride) a[i] = P[a[i]];
}

extern "C" int gmx_graph_create_rmat(int64_t N, int64_t M, long seed, double a, double b, double c,
                                     int permute, uint32_t flags, gmx_graph_t** out) {
    gmx_ws_scope ws;
    GMX_REQUIRE(out, "out is NULL");
    *out = nullptr;
    GMX_CHECK(check_sizes(N, M));
    GMX_REQUIRE(N >= 2, "N must be >= 2");
    GMX_REQUIRE(a + b + c < 1, "a+b+c must be < 1 (graph_gen.cc:161)");
    double d = 1 - (a + b + c);
    int32_t SCALE = (int32_t) log2((double) N);   // graph_gen.cc:179
    uint64_t D = 1 + 5 * (uint64_t) (SCALE > 1 ? SCALE - 1 : 0);
    uint64_t x0 = ((((uint64_t) seed) << 16) | 0x330EULL) & LCG_MASK;  // srand48

    lcg_table tab;
    tab.pw[0] = lcg_jump(D);
    for (int k = 1; k < 48; k++) tab.pw[k] = lcg_then(tab.pw[k - 1], tab.pw[k - 1]);

    gmx_graph* g = new gmx_graph();
    g->V = N;
    g->E = M;
    (void) hipGetDevice(&g->device);
    int st = GMX_OK;
    do {
        wbuf<int32_t> src, dst;
        wbuf<int64_t> bad, slots;
        dbuf<unsigned long long> nbad;
        size_t bad_cap = (size_t) M + 16;  // worst case: every attempt of a round is a self loop
        if (bad_cap > (1u << 24)) bad_cap = (size_t) (M / 16) + (1u << 20);
        if ((st = src.alloc((size_t) M)) || (st = dst.alloc((size_t) M)) || (st = bad.alloc(bad_cap)) ||
            (st = slots.alloc(bad_cap)) || (st = nbad.alloc(1))) break;
        int64_t attempts = 0, count = M;
        bool first = true;
        hipError_t he = hipSuccess;
        while (count > 0) {
            if ((he = hipMemset(nbad.p, 0, sizeof(unsigned long long))) != hipSuccess) break;
            int64_t threads = (count + ATT_PER_THREAD - 1) / ATT_PER_THREAD;
            int64_t blocks = (threads + 255) / 256;
            hipLaunchKernelGGL(rmat_attempts_kernel, dim3((unsigned) blocks), dim3(256), 0, 0,
                               tab, x0, attempts, count, first ? (const int64_t*) nullptr : (const int64_t*) slots.p,
                               (int32_t) N, SCALE, a, b, c, d, src.p, dst.p, bad.p, (unsigned long long) bad_cap, nbad.p);
            if ((he = hipGetLastError()) != hipSuccess) break;
            unsigned long long nb = 0;
            if ((he = hipMemcpy(&nb, nbad.p, sizeof(nb), hipMemcpyDeviceToHost)) != hipSuccess) break;
            attempts += count;
            if (nb > bad_cap) { gmx_set_error("rmat: self-loop list overflow (%llu)", nb); st = GMX_ERR_STATE; break; }
            if (nb) {
                // the order in which rejected slots are refilled does not matter for the
                // edge multiset, but keep it deterministic: sort the slot list
                size_t tb = 0;
                if ((he = rocprim::radix_sort_keys(nullptr, tb, bad.p, slots.p, (size_t) nb)) != hipSuccess) break;
                wbuf<char> tmp;
                if ((st = tmp.alloc(tb))) break;
                if ((he = rocprim::radix_sort_keys((void*) tmp.p, tb, bad.p, slots.p, (size_t) nb)) != hipSuccess) break;
                if ((he = hipDeviceSynchronize()) != hipSuccess) break;
            }
            count = (int64_t) nb;
            first = false;
        }
        if (st) break;
        if (he != hipSuccess) { gmx_set_error("rmat generation: %s", hipGetErrorString(he)); st = GMX_ERR_HIP; break; }
        bad.release();
        slots.release();

        if (permute) {  // graph_gen.cc:240-259; sequential by construction, N steps on the host
            std::vector<int32_t> P((size_t) N);
            for (int64_t i = 0; i < N; i++) P[i] = (int32_t) i;
            lcg_map j = lcg_jump((uint64_t) attempts * D);
            uint64_t x = (j.a * x0 + j.c) & LCG_MASK;
            for (int64_t i = 0; i < N; i++) {
                x = (LCG_A * x + LCG_C) & LCG_MASK;
                double r = (double) x * (1.0 / 281474976710656.0);
                int32_t k = (int32_t) ((int32_t) N * r);
                int32_t tmp = P[k];
                P[k] = P[i];
                P[i] = tmp;
            }
            wbuf<int32_t> dP;
            if ((st = dP.alloc((size_t) N))) break;
            if (hipMemcpy(dP.p, P.data(), sizeof(int32_t) * (size_t) N, hipMemcpyHostToDevice) != hipSuccess) {
                gmx_set_error("H2D copy of permutation failed");
                st = GMX_ERR_HIP;
                break;
            }
            hipLaunchKernelGGL(apply_perm_kernel, dim3(grid_for(M)), dim3(256), 0, 0, src.p, M, dP.p);
            hipLaunchKernelGGL(apply_perm_kernel, dim3(grid_for(M)), dim3(256), 0, 0, dst.p, M, dP.p);
            if (hipDeviceSynchronize() != hipSuccess) { gmx_set_error("apply_perm failed"); st = GMX_ERR_HIP; break; }
        }

        wbuf<uint64_t> keys, alt;
        if ((st = keys.alloc((size_t) M)) || (st = alt.alloc((size_t) M))) break;
        if ((st = gmx_keys_from_edges(src.p, dst.p, M, false, nullptr, keys.p, 0))) break;
        if (hipDeviceSynchronize() != hipSuccess) { gmx_set_error("keys_from_edges failed"); st = GMX_ERR_HIP; break; }
        src.release();
        dst.release();
        if ((st = build_from_forward_keys(g, keys, alt, !(flags & GMX_GRAPH_NO_REVERSE)))) break;
    } while (0);
    if (st != GMX_OK) { delete g; return st; }
    *out = g;
    return GMX_OK;
}

// ------------------------------------------------------------------ symmetrise
// keys of both orientations of every non-loop edge
__global__ void sym_keys_kernel(const uint64_t* __restrict__ fwd, int64_t E, uint64_t* __restrict__ out) {
    int64_t e = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t) gridDim.x * blockDim.x;
    const uint64_t NONE = ~0ULL;   // sorts last, removed afterwards
    for (; e < E; e += stride) {
        const uint64_t k = fwd[e];
        const uint64_t u = k >> 32, v = k & 0xffffffffu;
        out[2 * e] = (u == v) ? NONE : k;
        out[2 * e + 1] = (u == v) ? NONE : ((v << 32) | u);
    }
}

struct key_is_edge {
    __device__ bool operator()(const uint64_t& k) const { return k != ~0ULL; }
};

// Undirected simple version of g: every edge in both directions, duplicates and self loops removed
// (the measurement preparation of the triangle-counting config).
extern "C" int gmx_graph_symmetrize(const gmx_graph_t* g, gmx_graph_t** out) {
    gmx_ws_scope ws;
    GMX_REQUIRE(g && out, "NULL argument");
    *out = nullptr;
    GMX_REQUIRE(2 * g->E < (1LL << 31), "symmetrised edge count exceeds int32 edge_t");
    const int64_t E = g->E, V = g->V;
    gmx_graph* h = new gmx_graph();
    h->V = V;
    (void) hipGetDevice(&h->device);
    int st = GMX_OK;
    do {
        wbuf<uint64_t> fwd, both, alt, uniq;
        dbuf<int64_t> count;
        if ((st = fwd.alloc((size_t) E)) || (st = both.alloc((size_t) 2 * E)) || (st = alt.alloc((size_t) 2 * E)) ||
            (st = uniq.alloc((size_t) 2 * E)) || (st = count.alloc(1))) break;
        if ((st = gmx_keys_from_csr(g->begin.p, g->node_idx.p, V, E, false, nullptr, fwd.p, 0))) break;
        if (E) hipLaunchKernelGGL(sym_keys_kernel, dim3(grid_for(E)), dim3(256), 0, 0, (const uint64_t*) fwd.p, E, both.p);
        fwd.release();
        int64_t n = 0;
        if (E) {
            rocprim::double_buffer<uint64_t> db(both.p, alt.p);
            size_t tb = 0;
            hipError_t he = rocprim::radix_sort_keys(nullptr, tb, db, (size_t) (2 * E), 0u, 64u, 0);
            wbuf<char> tmp;
            if (he == hipSuccess && (st = tmp.alloc(tb))) break;
            if (he == hipSuccess) he = rocprim::radix_sort_keys((void*) tmp.p, tb, db, (size_t) (2 * E), 0u, 64u, 0);
            size_t ub = 0;
            if (he == hipSuccess) he = rocprim::unique(nullptr, ub, db.current(), uniq.p, count.p, (size_t) (2 * E), rocprim::equal_to<uint64_t>(), 0);
            wbuf<char> tmp2;
            if (he == hipSuccess && (st = tmp2.alloc(ub))) break;
            if (he == hipSuccess) he = rocprim::unique((void*) tmp2.p, ub, db.current(), uniq.p, count.p, (size_t) (2 * E), rocprim::equal_to<uint64_t>(), 0);
            if (he == hipSuccess) he = hipMemcpy(&n, count.p, sizeof(int64_t), hipMemcpyDeviceToHost);
            if (he != hipSuccess) { gmx_set_error("symmetrize: %s", hipGetErrorString(he)); st = GMX_ERR_HIP; break; }
            // the sentinel (self loops), if present, is the last unique key
            uint64_t last = 0;
            if (n > 0 && hipMemcpy(&last, uniq.p + (n - 1), sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) { gmx_set_error("symmetrize: copy failed"); st = GMX_ERR_HIP; break; }
            if (n > 0 && last == ~0ULL) n--;
        }
        h->E = n;
        both.release();
        // uniq holds the sorted forward keys; the graph is its own transpose
        if ((st = h->begin.alloc((size_t) V + 1)) || (st = h->node_idx.alloc((size_t) n)) ||
            (st = h->r_begin.alloc((size_t) V + 1)) || (st = h->r_node_idx.alloc((size_t) n))) break;
        hipLaunchKernelGGL(csr_extract_kernel, dim3(grid_for(n > V ? n : V + 1)), dim3(256), 0, 0, (const uint64_t*) uniq.p, V, n, h->begin.p, h->node_idx.p);
        if (hipMemcpy(h->r_begin.p, h->begin.p, sizeof(int32_t) * ((size_t) V + 1), hipMemcpyDeviceToDevice) != hipSuccess ||
            (n && hipMemcpy(h->r_node_idx.p, h->node_idx.p, sizeof(int32_t) * (size_t) n, hipMemcpyDeviceToDevice) != hipSuccess) ||
            hipDeviceSynchronize() != hipSuccess) { gmx_set_error("symmetrize: csr build failed"); st = GMX_ERR_HIP; break; }
        h->has_reverse = true;
    } while (0);
    if (st != GMX_OK) { delete h; return st; }
    *out = h;
    return GMX_OK;
}

// ------------------------------------------------------------------ misc
extern "C" int gmx_graph_edge_order(const gmx_graph_t* g, gmx_edge_t* e_idx2idx, int* is_identity) {
    GMX_REQUIRE(g, "graph is NULL");
    const bool ident = g->e_idx2idx.p == nullptr;
    if (is_identity) *is_identity = ident ? 1 : 0;
    if (e_idx2idx && !ident && g->E > 0)
        GMX_HIP(hipMemcpy(e_idx2idx, g->e_idx2idx.p, sizeof(int32_t) * (size_t) g->E, hipMemcpyDeviceToHost));
    return GMX_OK;
}

extern "C" int gmx_graph_free(gmx_graph_t* g) {
    if (g) {
        for (gmx_pr*& p : g->pr_cache) {
            if (p) gmx_pr_free(p);
            p = nullptr;
        }
        for (gmx_pr_multi*& m : g->pr_multi_cache) {
            if (m) gmx_pr_multi_free(m);
            m = nullptr;
        }
        delete g->tc_oriented;
        g->tc_oriented = nullptr;
        if (g->bfs_cache) gmx_bfs_free(g->bfs_cache);
        g->bfs_cache = nullptr;
    }
    delete g;
    return GMX_OK;
}

extern "C" int64_t gmx_graph_num_nodes(const gmx_graph_t* g) { return g ? g->V : -1; }
extern "C" int64_t gmx_graph_num_edges(const gmx_graph_t* g) { return g ? g->E : -1; }

extern "C" int gmx_graph_download(const gmx_graph_t* g, gmx_edge_t* begin, gmx_node_t* node_idx,
                                  gmx_edge_t* r_begin, gmx_node_t* r_node_idx) {
    GMX_REQUIRE(g, "graph is NULL");
    if (begin) GMX_HIP(hipMemcpy(begin, g->begin.p, sizeof(int32_t) * ((size_t) g->V + 1), hipMemcpyDeviceToHost));
    if (node_idx && g->E) GMX_HIP(hipMemcpy(node_idx, g->node_idx.p, sizeof(int32_t) * (size_t) g->E, hipMemcpyDeviceToHost));
    if (r_begin || r_node_idx) GMX_REQUIRE(g->has_reverse, "graph has no reverse CSR");
    if (r_begin) GMX_HIP(hipMemcpy(r_begin, g->r_begin.p, sizeof(int32_t) * ((size_t) g->V + 1), hipMemcpyDeviceToHost));
    if (r_node_idx && g->E) GMX_HIP(hipMemcpy(r_node_idx, g->r_node_idx.p, sizeof(int32_t) * (size_t) g->E, hipMemcpyDeviceToHost));
    return GMX_OK;
}

// e_rev2idx of gm_graph (/root/reference/apps/output_cpp/gm_graph/inc/gm_graph.h:141-142; built by
// make_reverse_edges, gm_graph.cc:205-304): for every slot of the reverse CSR the forward slot it mirrors.
// The reverse CSR is the forward edge list sorted by (dst, src); sorting the forward slot numbers along
// (stable radix sort of (dst << 32 | src) keys with the slot as value) gives the map directly, the k-th copy
// of a repeated edge in an in-row mirroring the k-th slot holding it in the out-row.
__global__ void iota_kernel(int32_t* __restrict__ a, int64_t n) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < n; i += stride) a[i] = (int32_t) i;
}

extern "C" int gmx_graph_reverse_edge_map(const gmx_graph_t* g, gmx_edge_t* e_rev2idx) {
    gmx_ws_scope ws;
    GMX_REQUIRE(g && e_rev2idx, "NULL argument");
    if (g->E == 0) return GMX_OK;
    hipStream_t s = 0;
    wbuf<uint64_t> keys, keys2;
    wbuf<int32_t> val, val2;
    GMX_CHECK(keys.alloc((size_t) g->E));
    GMX_CHECK(keys2.alloc((size_t) g->E));
    GMX_CHECK(val.alloc((size_t) g->E));
    GMX_CHECK(val2.alloc((size_t) g->E));
    GMX_CHECK(gmx_keys_from_csr(g->begin.p, g->node_idx.p, g->V, g->E, true, nullptr, keys.p, s));
    hipLaunchKernelGGL(iota_kernel, dim3(grid_for(g->E)), dim3(256), 0, s, val.p, g->E);
    size_t tb = 0;
    const unsigned end_bit = 32 + (unsigned) gmx_bits_for(g->V);
    GMX_HIP(rocprim::radix_sort_pairs(nullptr, tb, keys.p, keys2.p, val.p, val2.p, (size_t) g->E, 0u, end_bit, s));
    wbuf<char> tmp;
    GMX_CHECK(tmp.alloc(tb));
    GMX_HIP(rocprim::radix_sort_pairs((void*) tmp.p, tb, keys.p, keys2.p, val.p, val2.p, (size_t) g->E, 0u, end_bit, s));
    GMX_HIP(hipMemcpy(e_rev2idx, val2.p, sizeof(int32_t) * (size_t) g->E, hipMemcpyDeviceToHost));
    return GMX_OK;
}

// ------------------------------------------------------------------ GM_EDGE64 host builds (include/gmx.h)
// edge_t = int64_t on the host, 32-bit edge offsets on the device: narrow on the way in, widen on the way out.
extern "C" int gmx_graph_upload_e64(const int64_t* begin, const gmx_node_t* node_idx, const int64_t* r_begin, const gmx_node_t* r_node_idx,
                                    int64_t V, int64_t E, uint32_t flags, gmx_graph_t** out) {
    GMX_REQUIRE(out, "out is NULL");
    *out = nullptr;
    GMX_REQUIRE(V >= 0 && E >= 0 && (begin || V == 0), "bad argument");
    GMX_REQUIRE(E < (1LL << 31) - (1LL << 27), "%lld edges: the device kernels keep 32-bit edge offsets (below 2^31 - 2^27 edges)", (long long) E);
    std::vector<int32_t> b32((size_t) V + 1, 0), rb32;
    for (int64_t i = 0; i <= V && begin; i++) {
        GMX_REQUIRE(begin[i] >= 0 && begin[i] <= E, "begin[%lld] = %lld outside [0, E]", (long long) i, (long long) begin[i]);
        b32[(size_t) i] = (int32_t) begin[i];
    }
    if (r_begin) {
        rb32.resize((size_t) V + 1);
        for (int64_t i = 0; i <= V; i++) {
            GMX_REQUIRE(r_begin[i] >= 0 && r_begin[i] <= E, "r_begin[%lld] = %lld outside [0, E]", (long long) i, (long long) r_begin[i]);
            rb32[(size_t) i] = (int32_t) r_begin[i];
        }
    }
    return gmx_graph_upload(b32.data(), node_idx, r_begin ? rb32.data() : nullptr, r_node_idx, V, E, flags, out);
}

extern "C" int gmx_graph_download_e64(const gmx_graph_t* g, int64_t* begin, gmx_node_t* node_idx, int64_t* r_begin, gmx_node_t* r_node_idx) {
    GMX_REQUIRE(g, "graph is NULL");
    std::vector<int32_t> b32(begin ? (size_t) g->V + 1 : 0), rb32(r_begin ? (size_t) g->V + 1 : 0);
    GMX_CHECK(gmx_graph_download(g, begin ? b32.data() : nullptr, node_idx, r_begin ? rb32.data() : nullptr, r_node_idx));
    for (size_t i = 0; i < b32.size(); i++) begin[i] = b32[i];
    for (size_t i = 0; i < rb32.size(); i++) r_begin[i] = rb32[i];
    return GMX_OK;
}

extern "C" int gmx_graph_edge_order_e64(const gmx_graph_t* g, int64_t* e_idx2idx, int* is_identity) {
    GMX_REQUIRE(g && e_idx2idx && is_identity, "NULL argument");
    std::vector<int32_t> m((size_t) g->E);
    GMX_CHECK(gmx_graph_edge_order(g, m.data(), is_identity));
    if (!*is_identity)
        for (size_t i = 0; i < m.size(); i++) e_idx2idx[i] = m[i];
    return GMX_OK;
}

extern "C" int gmx_graph_reverse_edge_map_e64(const gmx_graph_t* g, int64_t* e_rev2idx) {
    GMX_REQUIRE(g && e_rev2idx, "NULL argument");
    std::vector<int32_t> m((size_t) g->E);
    GMX_CHECK(gmx_graph_reverse_edge_map(g, m.data()));
    for (size_t i = 0; i < m.size(); i++) e_rev2idx[i] = m[i];
    return GMX_OK;
}
