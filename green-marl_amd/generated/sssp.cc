// sssp.cc -- body of the generated `sssp` procedure, MI355X build.
// Emitted prologue: gm_rt_initialize(); G.freeze();   The edge property G_len is indexed by the forward edge
// slot (e = s.ToEdge() is the neighbour iterator) and stays the caller's: it is copied in on every call.
#include "sssp.h"
#include "gmx.h"

void sssp(gm_graph& G, int32_t* G_dist, int32_t* G_len, node_t& root) {
    gm_rt_initialize();
    G.freeze();
    gmx_graph_t* dev = G.device_mirror();
    gmx_stats_t st;
    if (dev == NULL || gmx_sssp(dev, root, G_len, G_dist, &st) != GMX_OK) {
        fprintf(stderr, "sssp: %s\n", gmx_last_error());
        abort();
    }
    gm_rt_cleanup();
}
