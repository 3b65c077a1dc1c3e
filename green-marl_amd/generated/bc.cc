// bc.cc -- body of the generated `comp_BC` procedure, MI355X build.
// Emitted prologue: gm_rt_initialize(); G.freeze(); G.make_reverse_edges();  (UpNbrs walks the reverse rows.)
// The emission instantiates a gm_bfs_template<short, omp, false, false, true> subclass per seed and runs
// prepare / do_bfs_forward / do_bfs_reverse on it (gm_cpp_gen_bfs.cc:88-275); the device BFS object does the same
// sweeps (csrc/gmx_bfs.hip, gmx_bc).  GMX_BC_SKIP_ROOT=1 selects upstream Green-Marl's `(v != s)` filters, which
// this fork's bc.gm lacks (without them every sigma is 0 and reached inner vertices get NaN -- the reference's
// own result, see include/gmx.h).
#include "bc.h"
#include "gmx.h"
#include <vector>

void comp_BC(gm_graph& G, float* G_BC, gm_node_seq& Seeds) {
    gm_rt_initialize();
    G.freeze();
    G.make_reverse_edges();
    std::vector<node_t> seeds;
    gm_node_seq::seq_iter it = Seeds.prepare_seq_iteration();
    while (it.has_next()) seeds.push_back(it.get_next());
    const char* skip = getenv("GMX_BC_SKIP_ROOT");
    gmx_graph_t* dev = G.device_mirror();
    gmx_stats_t st;
    if (dev == NULL || gmx_bc(dev, seeds.data(), (int32_t) seeds.size(), skip && atoi(skip) != 0, G_BC, &st) != GMX_OK) {
        fprintf(stderr, "comp_BC: %s\n", gmx_last_error());
        abort();
    }
    gm_rt_cleanup();
}
