// gmx_pr_multi.hip -- the whole-kernel PageRank entry over several GPUs, driven from ONE host thread.
//
// "single host thread drives all devices/streams"; 8e: 1-D vertex partition, replicated contribution
// vector, one exchange per iteration, diff = sum of the ranks' partials.  This is the C++ form of what
// green-marl_amd/dist_pagerank.py does with one process per GPU: gmx_pagerank_f64 / _f32 (and through them
// generated/pagerank.cc and bin/pagerank -- the emitted driver) use every visible device.
//
//   ranks      GMX_PR_RANKS (default: the devices in use); rank r lives on device r % devices, so several rank
//              states can share a device (how the one-GPU test box exercises the orchestration)
//   devices    GMX_DEVICES=<n> | all (the path is opt-in: without GMX_DEVICES / GMX_PR_RANKS the entry stays on one device)
//   graph      the CSR is copied once to every other device in use (replicated, as 8e prescribes) and cached
//   step       every rank's sweep is enqueued on its own stream; behind it, on the same stream, the rank's new
//              contributions go straight into every other rank's replica (hipMemcpyPeerAsync: the copy engines over
//              xGMI, no kernel, no IPC handles inside one process); an event per rank, which every other rank's
//              stream waits for, is the barrier -- a rank starts iteration k + 1 when all pieces of iteration k
//              have landed in its replica.  Replicas are double buffered, and a piece of iteration k + 1 can only
//              arrive after its sender has received everybody's iteration-k piece, i.e. after every sweep k ended.
//   GMX_EXCHANGE=allgather | allreduce   the exchange as RCCL calls from this thread (one communicator per device,
//              group calls): in-place ncclAllGather of the rank ranges, or the literal form of BASELINE.json's
//              north_star -- every replica zeroed outside its owner's range, then ncclAllReduce(sum).  librccl is
//              loaded on demand (dlopen), needs one rank per device.
//   diff       the emitted loop reads diff every iteration (pagerank.gm:18): the ranks' fp64 partials are read back
//              through pinned memory and added in rank order (what ATOMIC_ADD<double> does with threads).
#include "gmx_internal.h"

#include <dlfcn.h>
#include <stdlib.h>

#include <condition_variable>
#include <mutex>
#include <thread>

// ---- the few RCCL entry points used, resolved at run time ----
typedef void* rccl_comm_t;
struct rccl_api {
    void* lib = nullptr;
    int (*CommInitAll)(rccl_comm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(rccl_comm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, rccl_comm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool load() {
        if (lib) return true;
        for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return false;
        CommInitAll = (decltype(CommInitAll)) dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy)) dlsym(lib, "ncclCommDestroy");
        AllGather = (decltype(AllGather)) dlsym(lib, "ncclAllGather");
        AllReduce = (decltype(AllReduce)) dlsym(lib, "ncclAllReduce");
        GroupStart = (decltype(GroupStart)) dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd)) dlsym(lib, "ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString)) dlsym(lib, "ncclGetErrorString");
        return CommInitAll && CommDestroy && AllGather && AllReduce && GroupStart && GroupEnd;
    }
};
static rccl_api g_rccl;
#define RCCL_FLOAT32 7   // ncclFloat32 / ncclFloat64 / ncclSum of rccl.h
#define RCCL_FLOAT64 8
#define RCCL_SUM 0

enum { EX_PEER = 0, EX_ALLGATHER = 1, EX_ALLREDUCE = 2 };

struct gmx_pr_multi {
    int nranks = 0, ndev = 0, elem = 0, exchange = EX_PEER;
    std::vector<int> dev;                 // device of rank r
    std::vector<gmx_graph*> graph;        // per device slot; [0] is the caller's graph (not owned)
    std::vector<gmx_pr_t*> pr;
    std::vector<hipStream_t> stream;
    std::vector<hipEvent_t> done;
    std::vector<rccl_comm_t> comm;
    // pipelined peer exchange (two row chunks; see gmx_pr_multi_run): copy streams and "landed" events per
    // (sender, receiver, chunk), "computed" events per (rank, chunk)
    bool pipelined = false;
    std::vector<hipStream_t> cstream;     // [(r * nranks + q) * 2 + c]
    std::vector<hipEvent_t> landed;       // same index
    std::vector<hipEvent_t> computed;     // [r * 2 + c]
    int home = 0;                         // device the caller's graph lives on
    bool verified = false;                // the first exchange has been checked against the owners' ranges (verify_replicas)
    bool packed = false;                  // peer exchange of the packed lists (gmx_pr_push_packed / gmx_pr_unpack)
    ~gmx_pr_multi() {
        for (size_t r = 0; r < pr.size(); r++) {
            (void) hipSetDevice(dev[r]);
            if (pr[r]) gmx_pr_free(pr[r]);
            if (r < stream.size() && stream[r]) (void) hipStreamDestroy(stream[r]);
            if (r < done.size() && done[r]) (void) hipEventDestroy(done[r]);
            for (size_t q = 0; q < pr.size() && !cstream.empty(); q++)
                for (int c = 0; c < 2; c++) {
                    const size_t i = (r * pr.size() + q) * 2 + c;
                    if (cstream[i]) {
                        (void) hipStreamSynchronize(cstream[i]);
                        (void) hipStreamDestroy(cstream[i]);
                    }
                    if (landed[i]) (void) hipEventDestroy(landed[i]);
                }
            for (int c = 0; c < 2 && !computed.empty(); c++)
                if (computed[r * 2 + c]) (void) hipEventDestroy(computed[r * 2 + c]);
        }
        for (rccl_comm_t c : comm)
            if (c && g_rccl.CommDestroy) (void) g_rccl.CommDestroy(c);
        for (size_t d = 1; d < graph.size(); d++)
            if (graph[d]) {
                (void) hipSetDevice(graph[d]->device);
                delete graph[d];
            }
        (void) hipSetDevice(home);
    }
};

void gmx_pr_multi_free(gmx_pr_multi* m) { delete m; }

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

// ranks the whole-kernel entry would use for this graph (1: the single-GPU path).  Several GPUs are OPT-IN: an unchanged
// caller on a shared multi-GPU node keeps to the device its graph lives on unless GMX_DEVICES (how many devices to use,
// "all" = every visible one) or GMX_PR_RANKS (rank states; more than devices = several per device, the one-GPU test
// form) is set.
int gmx_pr_multi_ranks(const gmx_graph* g) {
    const char* ed = getenv("GMX_DEVICES");
    const char* er = getenv("GMX_PR_RANKS");
    if (!(ed && *ed) && !(er && *er)) return 1;
    int ndev = 1;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) ndev = 1;
    const int want_dev = (ed && !strcmp(ed, "all")) ? ndev : env_int("GMX_DEVICES", er && *er ? ndev : 1);
    if (want_dev >= 1 && want_dev < ndev) ndev = want_dev;
    int nranks = env_int("GMX_PR_RANKS", ndev);
    if (nranks < 1) nranks = 1;
    if ((int64_t) nranks > g->V) nranks = 1;
    return nranks;
}

// a copy of the CSR on `device` (hipMemcpyPeer from the graph's own device)
static int clone_graph(const gmx_graph* g, int device, gmx_graph** out) {
    gmx_graph* h = new gmx_graph();
    h->V = g->V;
    h->E = g->E;
    h->has_reverse = g->has_reverse;
    h->device = device;
    int st = GMX_OK;
    GMX_HIP(hipSetDevice(device));
    auto copy = [&](dbuf<int32_t>& dst, const dbuf<int32_t>& src) -> int {
        if (!src.p) return GMX_OK;
        GMX_CHECK(dst.alloc(src.n));
        GMX_HIP(hipMemcpyPeer(dst.p, device, src.p, g->device, sizeof(int32_t) * (src.n ? src.n : 1)));
        return GMX_OK;
    };
    if ((st = copy(h->begin, g->begin)) || (st = copy(h->node_idx, g->node_idx)) || (st = copy(h->r_begin, g->r_begin)) ||
        (st = copy(h->r_node_idx, g->r_node_idx))) {
        delete h;
        return st;
    }
    *out = h;
    return GMX_OK;
}

int gmx_pr_multi_create(gmx_graph* g, int elem, int nranks, gmx_pr_multi** out) {
    *out = nullptr;
    gmx_pr_multi* m = new gmx_pr_multi();
    int st = GMX_OK;
    do {
        int ndev = 1;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) ndev = 1;
        const char* ed = getenv("GMX_DEVICES");
        const int want_dev = (ed && !strcmp(ed, "all")) ? ndev : env_int("GMX_DEVICES", ndev);
        if (want_dev >= 1 && want_dev < ndev) ndev = want_dev;
        if (ndev > nranks) ndev = nranks;
        m->nranks = nranks;
        m->ndev = ndev;
        m->elem = elem;
        m->home = g->device;
        const char* ex = getenv("GMX_EXCHANGE");
        if (ex && !strcmp(ex, "allgather")) m->exchange = EX_ALLGATHER;
        else if (ex && !strcmp(ex, "allreduce")) m->exchange = EX_ALLREDUCE;
        else if (ex && *ex && strcmp(ex, "peer")) { gmx_set_error("GMX_EXCHANGE=%s: expected peer, allgather or allreduce", ex); st = GMX_ERR_ARG; break; }
        if (m->exchange != EX_PEER && nranks != ndev) {
            gmx_set_error("GMX_EXCHANGE=%s needs one rank per device (%d ranks on %d devices)", ex, nranks, ndev);
            st = GMX_ERR_ARG;
            break;
        }
        // device slot d: the graph's own device first, then the others in order
        std::vector<int> slot_dev;
        slot_dev.push_back(g->device);
        for (int d = 0; d < 64 && (int) slot_dev.size() < ndev; d++)
            if (d != g->device) slot_dev.push_back(d);
        m->graph.assign((size_t) ndev, nullptr);
        m->graph[0] = g;
        for (int d = 1; d < ndev && st == GMX_OK; d++) st = clone_graph(g, slot_dev[d], &m->graph[d]);
        if (st) break;
        for (int a = 0; a < ndev; a++)      // copies between replicas go device to device where the fabric allows
            for (int b = 0; b < ndev; b++) {
                int can = 0;
                if (a != b && hipDeviceCanAccessPeer(&can, slot_dev[a], slot_dev[b]) == hipSuccess && can) {
                    (void) hipSetDevice(slot_dev[a]);
                    (void) hipDeviceEnablePeerAccess(slot_dev[b], 0);   // "already enabled" is fine
                    (void) hipGetLastError();
                }
            }
        m->dev.resize((size_t) nranks);
        m->pr.assign((size_t) nranks, nullptr);
        m->stream.assign((size_t) nranks, nullptr);
        m->done.assign((size_t) nranks, nullptr);
        const uint32_t options = gmx_pr_default_options(g->V, nranks);
        for (int r = 0; r < nranks && st == GMX_OK; r++) {
            const int d = r % ndev;
            m->dev[r] = slot_dev[d];
            if (hipSetDevice(m->dev[r]) != hipSuccess) { gmx_set_error("hipSetDevice(%d) failed", m->dev[r]); st = GMX_ERR_HIP; break; }
            if ((st = gmx_pr_create(m->graph[d], elem, r, nranks, options, &m->pr[r]))) break;
            if (hipStreamCreateWithFlags(&m->stream[r], hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&m->done[r], hipEventDisableTiming) != hipSuccess) { gmx_set_error("stream/event creation failed"); st = GMX_ERR_HIP; break; }
        }
        if (st) break;
        // "send only what is read": when every rank's plan has the lists, wire the landing zones (one process: raw pointers)
        if (m->exchange == EX_PEER && nranks > 1 && nranks <= 16 && env_int("GMX_PUSH_PACKED", 1) != 0) {
            std::vector<std::vector<int64_t>> roff((size_t) nranks, std::vector<int64_t>((size_t) nranks, 0)), rcnt = roff;
            std::vector<void*> z0((size_t) nranks, nullptr), z1((size_t) nranks, nullptr);
            bool ok = true;
            for (int r = 0; r < nranks && ok; r++) {
                int64_t bytes = 0;
                ok = hipSetDevice(m->dev[r]) == hipSuccess && gmx_pr_packed_info(m->pr[r], nullptr, rcnt[(size_t) r].data(), roff[(size_t) r].data()) == GMX_OK &&
                     gmx_pr_recv_buffers(m->pr[r], &z0[(size_t) r], &z1[(size_t) r], &bytes) == GMX_OK;
            }
            for (int r = 0; r < nranks && ok; r++) {
                std::vector<int64_t> off((size_t) nranks, 0), cnt((size_t) nranks, 0);
                for (int q = 0; q < nranks; q++) { off[(size_t) q] = roff[(size_t) q][(size_t) r]; cnt[(size_t) q] = rcnt[(size_t) q][(size_t) r]; }
                ok = hipSetDevice(m->dev[r]) == hipSuccess && gmx_pr_set_peers_packed(m->pr[r], z0.data(), z1.data(), off.data(), cnt.data()) == GMX_OK;
            }
            m->packed = ok;
        }
        // Peer copies + every in-edge binned: with GMX_PR_MULTI_PIPELINE=1 the step is cut in two row chunks and
        // pipelined (see pipelined_iteration).  Off by default: ONE host thread then issues ~480 HIP calls per
        // iteration for 8 ranks (16 launches, 112 copies on streams of their own, their events and waits) against ~130
        // for the step in one piece, and at ~5 us a call that is longer than the 0.4 ms the ranks compute -- measured
        // with 8 rank states on one GPU (RMAT-24, 50 iterations): 567 ms pipelined, 380 ms in one piece.  The
        // one-process-per-GPU driver (dist_pagerank.py) issues ~30 calls per rank and iteration and pipelines by default.
        if (m->exchange == EX_PEER && nranks > 1 && !m->packed && env_int("GMX_PR_MULTI_PIPELINE", 0) != 0) {
            bool ok = true;
            for (int r = 0; r < nranks && ok; r++) {
                int classes = 0, chunks = 0;
                if (hipSetDevice(m->dev[r]) != hipSuccess) { ok = false; break; }
                ok = gmx_pr_gather_classes(m->pr[r], &classes) == GMX_OK && classes == 2 && gmx_pr_set_chunks(m->pr[r], 2) == GMX_OK &&
                     gmx_pr_num_chunks(m->pr[r], &chunks) == GMX_OK && chunks == 2;
            }
            if (ok) {
                m->cstream.assign((size_t) nranks * nranks * 2, nullptr);
                m->landed.assign((size_t) nranks * nranks * 2, nullptr);
                m->computed.assign((size_t) nranks * 2, nullptr);
                for (int r = 0; r < nranks && st == GMX_OK; r++) {
                    if (hipSetDevice(m->dev[r]) != hipSuccess) { gmx_set_error("hipSetDevice(%d) failed", m->dev[r]); st = GMX_ERR_HIP; break; }
                    for (int c = 0; c < 2 && st == GMX_OK; c++) {
                        if (hipEventCreateWithFlags(&m->computed[(size_t) r * 2 + c], hipEventDisableTiming) != hipSuccess) st = GMX_ERR_HIP;
                        for (int q = 0; q < nranks && st == GMX_OK; q++) {
                            if (q == r) continue;
                            const size_t i = ((size_t) r * nranks + q) * 2 + c;
                            if (hipStreamCreateWithFlags(&m->cstream[i], hipStreamNonBlocking) != hipSuccess ||
                                hipEventCreateWithFlags(&m->landed[i], hipEventDisableTiming) != hipSuccess) st = GMX_ERR_HIP;
                        }
                    }
                    if (st) gmx_set_error("stream/event creation failed");
                }
                if (st) break;
                m->pipelined = true;
            } else {
                for (int r = 0; r < nranks; r++) {   // back to one piece
                    (void) hipSetDevice(m->dev[r]);
                    (void) gmx_pr_set_chunks(m->pr[r], 1);
                }
            }
        }
        if (m->exchange != EX_PEER) {
            if (!g_rccl.load()) { gmx_set_error("GMX_EXCHANGE: librccl.so could not be loaded (%s)", dlerror()); st = GMX_ERR_STATE; break; }
            m->comm.assign((size_t) nranks, nullptr);
            const int rc = g_rccl.CommInitAll(m->comm.data(), nranks, m->dev.data());
            if (rc != 0) { gmx_set_error("ncclCommInitAll: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error"); st = GMX_ERR_STATE; break; }
        }
    } while (0);
    (void) hipSetDevice(g->device);
    if (st != GMX_OK) { delete m; return st; }
    *out = m;
    return GMX_OK;
}

// the replica of rank r that holds the newest contributions, and the geometry shared by all ranks
static char* replica(gmx_pr_multi* m, int r) {
    void* p = nullptr;
    int64_t n = 0;
    (void) gmx_pr_contrib_full(m->pr[r], &p, &n);
    return (char*) p;
}

// every rank's range [r * slice, r * slice + need) of its newest replica -> the same place in every other replica
static int exchange(gmx_pr_multi* m) {
    void* sp = nullptr;
    int64_t slice = 0, need = 0, total = 0;
    GMX_CHECK(gmx_pr_contrib_slice(m->pr[0], &sp, &slice));
    GMX_CHECK(gmx_pr_exchange_count(m->pr[0], &need));
    GMX_CHECK(gmx_pr_contrib_full(m->pr[0], &sp, &total));
    const size_t es = (size_t) m->elem;
    if (m->exchange == EX_PEER && m->packed) {   // pack -> copy engines -> (all landed) -> unpack
        for (int r = 0; r < m->nranks; r++) {
            GMX_HIP(hipSetDevice(m->dev[r]));
            GMX_CHECK(gmx_pr_push_packed(m->pr[r], -1, m->stream[r]));
            GMX_CHECK(gmx_pr_push_join(m->pr[r], m->stream[r]));
            GMX_HIP(hipEventRecord(m->done[r], m->stream[r]));
        }
        for (int q = 0; q < m->nranks; q++) {
            GMX_HIP(hipSetDevice(m->dev[q]));
            for (int r = 0; r < m->nranks; r++)
                if (r != q) GMX_HIP(hipStreamWaitEvent(m->stream[q], m->done[r], 0));
            GMX_CHECK(gmx_pr_unpack(m->pr[q], -1, m->stream[q]));
        }
        return GMX_OK;
    }
    if (m->exchange == EX_PEER) {
        for (int r = 0; r < m->nranks; r++) {
            GMX_HIP(hipSetDevice(m->dev[r]));
            const size_t at = (size_t) r * (size_t) slice * es, bytes = (size_t) need * es;
            const char* src = replica(m, r) + at;
            for (int i = 1; i < m->nranks; i++) {
                const int q = (r + i) % m->nranks;   // every rank starts with a different peer
                GMX_HIP(hipMemcpyPeerAsync(replica(m, q) + at, m->dev[q], src, m->dev[r], bytes, m->stream[r]));
            }
            GMX_HIP(hipEventRecord(m->done[r], m->stream[r]));
        }
        for (int q = 0; q < m->nranks; q++)
            for (int r = 0; r < m->nranks; r++)
                if (r != q) GMX_HIP(hipStreamWaitEvent(m->stream[q], m->done[r], 0));
        return GMX_OK;
    }
    const int dt = m->elem == 4 ? RCCL_FLOAT32 : RCCL_FLOAT64;
    if (m->exchange == EX_ALLREDUCE) {   // the literal north-star form: owned range filled, rest zero => sum = concatenation
        for (int r = 0; r < m->nranks; r++) {
            GMX_HIP(hipSetDevice(m->dev[r]));
            char* buf = replica(m, r);
            const size_t lo = (size_t) r * (size_t) slice * es, hi = lo + (size_t) slice * es, end = (size_t) total * es;
            if (lo) GMX_HIP(hipMemsetAsync(buf, 0, lo, m->stream[r]));
            if (end > hi) GMX_HIP(hipMemsetAsync(buf + hi, 0, end - hi, m->stream[r]));
        }
    }
    int rc = g_rccl.GroupStart();
    for (int r = 0; r < m->nranks && rc == 0; r++) {
        char* buf = replica(m, r);
        if (m->exchange == EX_ALLGATHER) rc = g_rccl.AllGather(buf + (size_t) r * (size_t) slice * es, buf, (size_t) slice, dt, m->comm[r], m->stream[r]);
        else rc = g_rccl.AllReduce(buf, buf, (size_t) total, dt, RCCL_SUM, m->comm[r], m->stream[r]);
    }
    if (rc == 0) rc = g_rccl.GroupEnd();
    if (rc != 0) { gmx_set_error("RCCL exchange failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error"); return GMX_ERR_STATE; }
    return GMX_OK;
}

// First contact.  The peer copies (or RCCL calls) of this path cannot be exercised on the one-GPU development boxes,
// so the first exchange of a gmx_pr_multi is checked before anything is computed from it: every rank's replica must
// hold, in every other rank's range, exactly the words the owner holds there.  Word compare on the receiving device
// (the owner's range is fetched with one more peer copy into scratch).  A mismatch makes gmx_pr_multi_run return
// GMX_ERR_STATE with `verified` still false; the entry then notes it on stderr and runs the single-GPU path.
__global__ void prm_compare_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, int64_t n, unsigned long long* __restrict__ bad) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    unsigned long long c = 0;
    for (; i < n; i += stride) c += a[i] != b[i];
    if (c) atomicAdd(bad, c);
}

__global__ void prm_compare_list_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, const int32_t* __restrict__ list, int64_t n,
                                        int words_per_elem, unsigned long long* __restrict__ bad) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    unsigned long long c = 0;
    for (; i < n; i += stride)
        for (int w = 0; w < words_per_elem; w++) c += a[(int64_t) list[i] * words_per_elem + w] != b[(int64_t) list[i] * words_per_elem + w];
    if (c) atomicAdd(bad, c);
}

struct prm_event {   // released on every return path
    hipEvent_t e = nullptr;
    ~prm_event() { if (e) (void) hipEventDestroy(e); }
};

static int verify_replicas(gmx_pr_multi* m) {
    void* sp = nullptr;
    int64_t slice = 0, need = 0;
    GMX_CHECK(gmx_pr_contrib_slice(m->pr[0], &sp, &slice));
    GMX_CHECK(gmx_pr_exchange_count(m->pr[0], &need));
    for (int r = 0; r < m->nranks; r++) {
        GMX_HIP(hipSetDevice(m->dev[r]));
        GMX_HIP(hipStreamSynchronize(m->stream[r]));
    }
    const size_t es = (size_t) m->elem, words = (size_t) need * es / 4;
    unsigned long long total_bad = 0;
    for (int q = 0; q < m->nranks && words > 0; q++) {
        GMX_HIP(hipSetDevice(m->dev[q]));
        dbuf<uint32_t> tmp;
        dbuf<unsigned long long> bad;
        GMX_CHECK(tmp.alloc(words));
        GMX_CHECK(bad.alloc(1));
        GMX_HIP(hipMemset(bad.p, 0, sizeof(unsigned long long)));
        for (int r = 0; r < m->nranks; r++) {
            if (r == q) continue;
            const size_t at = (size_t) r * (size_t) slice * es;
            GMX_HIP(hipMemcpyPeer(tmp.p, m->dev[q], replica(m, r) + at, m->dev[r], words * 4));
            if (m->packed) {   // only the positions rank q reads have travelled
                void* lp = nullptr;
                int64_t ln = 0;
                GMX_CHECK(gmx_pr_recv_list(m->pr[q], r, &lp, &ln));
                hipLaunchKernelGGL(prm_compare_list_kernel, dim3(1024), dim3(256), 0, 0, (const uint32_t*) tmp.p, (const uint32_t*) (replica(m, q) + at),
                                   (const int32_t*) lp, ln, (int) (es / 4), bad.p);
            } else
                hipLaunchKernelGGL(prm_compare_kernel, dim3(1024), dim3(256), 0, 0, (const uint32_t*) tmp.p, (const uint32_t*) (replica(m, q) + at), (int64_t) words, bad.p);
            GMX_HIP(hipDeviceSynchronize());
        }
        unsigned long long hb = 0;
        GMX_HIP(hipMemcpy(&hb, bad.p, sizeof(hb), hipMemcpyDeviceToHost));
        total_bad += hb;
    }
    if (total_bad) {
        gmx_set_error("multi-GPU pagerank: first exchange failed its check (%llu words differ between a replica and the owner's range)", total_bad);
        return GMX_ERR_STATE;
    }
    m->verified = true;
    return GMX_OK;
}

// do { sweep; exchange; diff } while (diff > e && cnt < max)   (pagerank.gm:9-19), then the ranks of every rank's rows
// One pipelined iteration of all ranks (peer copies, two row chunks; the order of DistPageRank._step_pipelined with
// events in place of barriers -- one host thread sees every stream):
//   gather(0)  phase 1 over the hub tiles     after the peers' HUB chunks of the previous iteration have landed
//   gather(1)  phase 1 over the other tiles   after their TAIL chunks have landed too
//   chunk 0    tail bins -> tail copies to every peer (copy streams of their own), chunk 1  hub bins -> hub copies
// so the long tail copies run under the hub bins of this iteration and the hub tiles of the next one.  A replica is
// written by a peer only after events that follow the owner's last read of it (two buffers, as in the Python driver).
static int pipelined_iteration(gmx_pr_multi* m, bool first) {
    const int N = m->nranks;
    const size_t es = (size_t) m->elem;
    std::vector<char*> next((size_t) N, nullptr);
    int64_t slice = 0;
    for (int r = 0; r < N; r++) {
        void* p = nullptr;
        int64_t n = 0;
        GMX_HIP(hipSetDevice(m->dev[r]));
        GMX_CHECK(gmx_pr_contrib_next_full(m->pr[r], &p, &n));
        next[(size_t) r] = (char*) p;
        GMX_CHECK(gmx_pr_contrib_slice(m->pr[r], &p, &slice));
    }
    auto idx = [&](int r, int q, int c) { return ((size_t) r * N + q) * 2 + c; };
    for (int cls = 0; cls < 2; cls++)
        for (int r = 0; r < N; r++) {
            GMX_HIP(hipSetDevice(m->dev[r]));
            if (!first)   // (the first iteration reads what exchange() delivered, already ordered on the compute streams)
                for (int q = 0; q < N; q++)
                    if (q != r) GMX_HIP(hipStreamWaitEvent(m->stream[r], m->landed[idx(q, r, cls == 0 ? 1 : 0)], 0));
            GMX_CHECK(gmx_pr_step_gather(m->pr[r], cls, m->stream[r]));
        }
    for (int c = 0; c < 2; c++)
        for (int r = 0; r < N; r++) {
            GMX_HIP(hipSetDevice(m->dev[r]));
            GMX_CHECK(gmx_pr_step_chunk(m->pr[r], c, m->stream[r]));
            GMX_HIP(hipEventRecord(m->computed[(size_t) r * 2 + c], m->stream[r]));
            int64_t off = 0, cnt = 0;
            GMX_CHECK(gmx_pr_chunk_range(m->pr[r], c, &off, &cnt));
            for (int i = 1; i < N; i++) {
                const int q = (r + i) % N;   // every rank starts with a different peer
                const size_t at = ((size_t) r * (size_t) slice + (size_t) off) * es;
                GMX_HIP(hipStreamWaitEvent(m->cstream[idx(r, q, c)], m->computed[(size_t) r * 2 + c], 0));
                if (cnt > 0)
                    GMX_HIP(hipMemcpyPeerAsync(next[(size_t) q] + at, m->dev[q], next[(size_t) r] + at, m->dev[r], (size_t) cnt * es, m->cstream[idx(r, q, c)]));
                GMX_HIP(hipEventRecord(m->landed[idx(r, q, c)], m->cstream[idx(r, q, c)]));
            }
        }
    return GMX_OK;
}

// every copy of the pipelined exchange has landed (before the ranks are read back, reset or freed)
static int pipelined_drain(gmx_pr_multi* m) {
    for (size_t i = 0; i < m->cstream.size(); i++)
        if (m->cstream[i]) GMX_HIP(hipStreamSynchronize(m->cstream[i]));
    return GMX_OK;
}

// ---- the iteration loop with one host thread per rank (packed peer exchange) ----
// One thread issuing every rank's launches, copies and event calls is host-bound long before 8 GPUs are busy: ~370 HIP
// calls per iteration for 8 ranks at 3-5 us each is 1-2 ms, against the ~0.3 ms a rank computes.  The emitted entry is
// called from one thread and opens its own parallel regions (#pragma omp parallel in the reference's emission,
// gm_cpp_gen.cc:1874-1909); here the "parallel region" is one std::thread per rank for the duration of the loop, each
// issuing only its own device's ~45 calls.  Two host barriers per iteration: A -- every rank has recorded its
// "my pieces are on their way" event before anybody waits on it; B -- every rank's diff is on the host.
namespace {
struct host_barrier {
    std::mutex mu;
    std::condition_variable cv;
    int n, waiting = 0;
    unsigned long long gen = 0;
    explicit host_barrier(int n_) : n(n_) {}
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        const unsigned long long g = gen;
        if (++waiting == n) { waiting = 0; gen++; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
};
}   // namespace

static int threaded_loop(gmx_pr_multi* m, double e, int32_t max_iter, double* diff_out, int32_t* cnt_out) {
    const int N = m->nranks;
    host_barrier bar(N);
    std::vector<int> status((size_t) N, GMX_OK);
    std::vector<std::string> message((size_t) N);
    std::vector<double> dr((size_t) N, 0.0);
    double diff = 0.0;
    int32_t cnt = 0;
    auto body = [&](int r) {
        int st = hipSetDevice(m->dev[r]) == hipSuccess ? GMX_OK : GMX_ERR_HIP;
        int32_t it = 0;   // iterations done (the same number in every thread)
        for (;;) {
            // every thread runs every barrier of an iteration, whatever its own status: nobody is left waiting
            if (st == GMX_OK) st = gmx_pr_step(m->pr[r], m->stream[r]);
            if (st == GMX_OK) st = gmx_pr_push_packed(m->pr[r], -1, m->stream[r]);
            if (st == GMX_OK) st = gmx_pr_push_join(m->pr[r], m->stream[r]);
            if (st == GMX_OK && hipEventRecord(m->done[r], m->stream[r]) != hipSuccess) st = GMX_ERR_HIP;
            bar.wait();                                                   // A
            for (int q = 0; q < N && st == GMX_OK; q++)
                if (q != r && hipStreamWaitEvent(m->stream[r], m->done[q], 0) != hipSuccess) st = GMX_ERR_HIP;
            if (st == GMX_OK) st = gmx_pr_unpack(m->pr[r], -1, m->stream[r]);
            if (st == GMX_OK) st = gmx_pr_diff(m->pr[r], m->stream[r], &dr[(size_t) r]);   // (synchronises the rank's stream)
            if (st != GMX_OK && status[(size_t) r] == GMX_OK) { status[(size_t) r] = st; message[(size_t) r] = gmx_last_error(); }
            bar.wait();                                                   // B
            bool go;
            {   // every thread forms the same sum in rank order and takes the same decision
                double t = 0.0;
                bool bad = false;
                for (int q = 0; q < N; q++) { t += dr[(size_t) q]; bad = bad || status[(size_t) q] != GMX_OK; }
                it++;
                go = !bad && (t > e) && (it < max_iter);
                if (r == 0) { diff = t; cnt = it; }
            }
            if (!go) break;   // (dr[] is next written behind barrier A of the next iteration, which needs every thread)
        }
    };
    std::vector<std::thread> workers;
    for (int r = 1; r < N; r++) workers.emplace_back(body, r);
    body(0);
    for (std::thread& t : workers) t.join();
    (void) hipSetDevice(m->dev[0]);
    *diff_out = diff;
    *cnt_out = cnt;
    for (int r = 0; r < N; r++)
        if (status[(size_t) r] != GMX_OK) {
            gmx_set_error("rank %d: %s", r, message[(size_t) r].c_str());
            return status[(size_t) r];
        }
    return GMX_OK;
}

int gmx_pr_multi_run(gmx_pr_multi* m, double e, double d, int32_t max_iter, void* rank_host, gmx_stats_t* stats) {
    double diff = 0.0;
    int32_t cnt = 0;
    for (int r = 0; r < m->nranks; r++) {
        GMX_HIP(hipSetDevice(m->dev[r]));
        GMX_CHECK(gmx_pr_reset(m->pr[r], d));
    }
    prm_event g0, g1;
    GMX_HIP(hipSetDevice(m->dev[0]));
    GMX_HIP(hipEventCreate(&g0.e));
    GMX_HIP(hipEventCreate(&g1.e));
    hipEvent_t ev0 = g0.e, ev1 = g1.e;
    int st = GMX_OK;
    do {
        if ((st = exchange(m))) break;   // the reset filled every rank's own range of the current replica only
        if (!m->verified && m->nranks > 1 && (st = verify_replicas(m))) break;
        GMX_HIP(hipSetDevice(m->dev[0]));
        (void) hipEventRecord(ev0, m->stream[0]);
        if (m->packed && !m->pipelined && m->exchange == EX_PEER && m->nranks > 1 && env_int("GMX_PR_MULTI_THREADS", 1) != 0) {
            if ((st = threaded_loop(m, e, max_iter, &diff, &cnt))) break;
        } else do {
            if (m->pipelined) {
                if ((st = pipelined_iteration(m, cnt == 0))) break;
            } else {
                for (int r = 0; r < m->nranks && st == GMX_OK; r++) {
                    if (hipSetDevice(m->dev[r]) != hipSuccess) { gmx_set_error("hipSetDevice failed"); st = GMX_ERR_HIP; break; }
                    st = gmx_pr_step(m->pr[r], m->stream[r]);
                }
                if (st || (st = exchange(m))) break;
            }
            diff = 0.0;
            for (int r = 0; r < m->nranks && st == GMX_OK; r++) {   // rank order: a fixed sum
                double dr = 0.0;
                if (hipSetDevice(m->dev[r]) != hipSuccess) { gmx_set_error("hipSetDevice failed"); st = GMX_ERR_HIP; break; }
                st = gmx_pr_diff(m->pr[r], m->stream[r], &dr);
                diff += dr;
            }
            if (st) break;
            cnt++;
        } while ((diff > e) && (cnt < max_iter));
        if (m->pipelined) {
            const int st2 = pipelined_drain(m);
            if (st == GMX_OK) st = st2;
        }
        if (st) break;
        GMX_HIP(hipSetDevice(m->dev[0]));
        (void) hipEventRecord(ev1, m->stream[0]);
        (void) hipEventSynchronize(ev1);
        float ms = 0;
        (void) hipEventElapsedTime(&ms, ev0, ev1);
        double t0 = 0;
        {
            hipEvent_t c0, c1;
            (void) hipEventCreate(&c0);
            (void) hipEventCreate(&c1);
            (void) hipEventRecord(c0, 0);
            for (int r = 0; r < m->nranks && st == GMX_OK; r++) {
                if (hipSetDevice(m->dev[r]) != hipSuccess) { gmx_set_error("hipSetDevice failed"); st = GMX_ERR_HIP; break; }
                st = gmx_pr_download(m->pr[r], rank_host);   // scatters the rows this rank owns
            }
            (void) hipSetDevice(m->dev[0]);
            (void) hipEventRecord(c1, 0);
            (void) hipEventSynchronize(c1);
            float cms = 0;
            (void) hipEventElapsedTime(&cms, c0, c1);
            t0 = cms;
            (void) hipEventDestroy(c0);
            (void) hipEventDestroy(c1);
        }
        if (stats) {
            stats->iterations = cnt;
            stats->last_diff = diff;
            stats->kernel_ms = ms;
            stats->d2h_ms = t0;
        }
    } while (0);
    (void) hipSetDevice(m->home);
    return st;
}

bool gmx_pr_multi_verified(const gmx_pr_multi* m) { return m && m->verified; }
