// hop_dist.cc -- body of the generated `hop_dist` procedure, MI355X build.
// Emitted prologue: gm_rt_initialize(); G.freeze();   (no reverse edges needed by the .gm source;
// the device BFS uses them for its bottom-up levels when the host graph already has them.)
#include "hop_dist.h"
#include "gmx.h"

void hop_dist(gm_graph& G, int32_t* G_dist, node_t& root) {
    gm_rt_initialize();
    G.freeze();
    gmx_graph_t* dev = G.device_mirror();
    gmx_stats_t st;
    if (dev == NULL || gmx_hop_dist(dev, root, G_dist, &st) != GMX_OK) {
        fprintf(stderr, "hop_dist: %s\n", gmx_last_error());
        abort();
    }
    gm_rt_cleanup();
}
