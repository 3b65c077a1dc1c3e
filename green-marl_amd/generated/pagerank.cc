// pagerank.cc -- body of the generated `pagerank` procedure, MI355X build.
// The reference's emitted body (restated in SURVEY.md section 8 a-1) is: prologue
// `gm_rt_initialize(); G.freeze(); G.make_reverse_edges();`, the do/while neighbour-reduction
// loop, epilogue `gm_rt_cleanup()`.  Here the loop runs on the device behind the C ABI
// (include/gmx.h); the caller-owned property array is written back before returning, as the
// Shoal copy-back does in the reference (gm_cpp_gen.cc:1457-1486).
#include "pagerank.h"
#include "gmx.h"

static gmx_graph_t* mirror_or_die(gm_graph& G, const char* who) {
    gmx_graph_t* dev = G.device_mirror();
    if (dev == NULL) {   // the reference has no error channel (void return): same convention as its asserts
        fprintf(stderr, "%s: no device graph (%s)\n", who, gmx_last_error());
        abort();
    }
    return dev;
}

void pagerank(gm_graph& G, double e, double d, int32_t max, double* G_pg_rank) {
    gm_rt_initialize();
    G.freeze();
    G.make_reverse_edges();
    gmx_stats_t st;
    if (gmx_pagerank_f64(mirror_or_die(G, "pagerank"), e, d, max, G_pg_rank, &st) != GMX_OK) {
        fprintf(stderr, "pagerank: %s\n", gmx_last_error());
        abort();
    }
    gm_rt_cleanup();
}

void pagerank(gm_graph& G, float e, float d, int32_t max, float* G_pg_rank) {
    gm_rt_initialize();
    G.freeze();
    G.make_reverse_edges();
    gmx_stats_t st;
    if (gmx_pagerank_f32(mirror_or_die(G, "pagerank"), e, d, max, G_pg_rank, &st) != GMX_OK) {
        fprintf(stderr, "pagerank: %s\n", gmx_last_error());
        abort();
    }
    gm_rt_cleanup();
}
