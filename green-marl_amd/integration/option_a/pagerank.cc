// Option A body of `pagerank` (apps/src/pagerank.gm; call site apps/output_cpp/src/pagerank_main.cc:28) against the
// reference's gm_graph.  Emitted prologue / epilogue kept (gm_cpp_gen.cc:1307-1368, 1446-1507).
#include "pagerank.h"
#include "gmx_binding.h"

void pagerank(gm_graph& G, double e, double d, int32_t max, double* G_pg_rank) {
    gm_rt_initialize();
    G.freeze();
    G.make_reverse_edges();
    GMX_OR_DIE("pagerank", gmx_pagerank_f64(gmx_mirror_of(G, true, "pagerank"), e, d, max, G_pg_rank, NULL));
    gm_rt_cleanup();
}
