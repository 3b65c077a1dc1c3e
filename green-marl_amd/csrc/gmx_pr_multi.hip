// gmx_pr_multi.hip -- the whole-kernel PageRank entry over several GPUs, driven from ONE host thread.
//
// "single host thread drives all devices/streams"; 8e: 1-D vertex partition, replicated contribution
// vector, one exchange per iteration, diff = sum of the ranks' partials.  This is the C++ form of what
// green-marl_amd/dist_pagerank.py does with one process per GPU: gmx_pagerank_f64 / _f32 (and through them
// generated/pagerank.cc and bin/pagerank -- the emitted driver) use every visible device.
//
//   ranks      GMX_PR_RANKS (default: the devices in use); rank r lives on device r % devices, so several rank
//              states can share a device (how the one-GPU test box exercises the orchestration)
//   devices    GMX_DEVICES (default: all visible)
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
    int home = 0;                         // device the caller's graph lives on
    ~gmx_pr_multi() {
        for (size_t r = 0; r < pr.size(); r++) {
            (void) hipSetDevice(dev[r]);
            if (pr[r]) gmx_pr_free(pr[r]);
            if (r < stream.size() && stream[r]) (void) hipStreamDestroy(stream[r]);
            if (r < done.size() && done[r]) (void) hipEventDestroy(done[r]);
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

// ranks the whole-kernel entry would use for this graph (1: the single-GPU path)
int gmx_pr_multi_ranks(const gmx_graph* g) {
    int ndev = 1;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) ndev = 1;
    const int want_dev = env_int("GMX_DEVICES", ndev);
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
        const int want_dev = env_int("GMX_DEVICES", ndev);
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

// do { sweep; exchange; diff } while (diff > e && cnt < max)   (pagerank.gm:9-19), then the ranks of every rank's rows
int gmx_pr_multi_run(gmx_pr_multi* m, double e, double d, int32_t max_iter, void* rank_host, gmx_stats_t* stats) {
    double diff = 0.0;
    int32_t cnt = 0;
    for (int r = 0; r < m->nranks; r++) {
        GMX_HIP(hipSetDevice(m->dev[r]));
        GMX_CHECK(gmx_pr_reset(m->pr[r], d));
    }
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    GMX_HIP(hipSetDevice(m->dev[0]));
    GMX_HIP(hipEventCreate(&ev0));
    GMX_HIP(hipEventCreate(&ev1));
    int st = GMX_OK;
    do {
        if ((st = exchange(m))) break;   // the reset filled every rank's own range of the current replica only
        GMX_HIP(hipSetDevice(m->dev[0]));
        (void) hipEventRecord(ev0, m->stream[0]);
        do {
            for (int r = 0; r < m->nranks && st == GMX_OK; r++) {
                if (hipSetDevice(m->dev[r]) != hipSuccess) { gmx_set_error("hipSetDevice failed"); st = GMX_ERR_HIP; break; }
                st = gmx_pr_step(m->pr[r], m->stream[r]);
            }
            if (st || (st = exchange(m))) break;
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
    (void) hipEventDestroy(ev0);
    (void) hipEventDestroy(ev1);
    (void) hipSetDevice(m->home);
    return st;
}
