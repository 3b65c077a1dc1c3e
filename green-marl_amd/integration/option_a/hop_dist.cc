// Option A body of `hop_dist` (apps/src/hop_dist.gm; call site apps/output_cpp/src/hop_dist_main.cc:28).
#include "hop_dist.h"
#include "gmx_binding.h"

void hop_dist(gm_graph& G, int32_t* G_dist, node_t& root) {
    gm_rt_initialize();
    G.freeze();
    GMX_OR_DIE("hop_dist", gmx_hop_dist(gmx_mirror_of(G, false, "hop_dist"), root, G_dist, NULL));
    gm_rt_cleanup();
}
