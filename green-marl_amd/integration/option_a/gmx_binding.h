// gmx_binding.h -- Option A of INTEGRATION.md: the generated procedures bound to the REFERENCE's own gm_graph class.
//
// Everything a maintainer of libshoal/Green-Marl adds to run the hot path on an MI355X without touching gm_graph:
// this header + the three bodies next to it replace apps/output_cpp/generated/{pagerank,hop_dist,triangle_counting}.cc;
// link with -lgmx.  It reads exactly the public CSR arrays the emitted code reads
// (/root/reference/apps/output_cpp/gm_graph/inc/gm_graph.h:133-142) and the state queries of :151-173.
// tests/test_host_cpp.py::test_option_a_links_against_the_reference_gm_graph compiles these files against the
// reference's headers and links them with the reference's compiled gm_graph objects and unchanged drivers.
#ifndef GMX_BINDING_H_
#define GMX_BINDING_H_

#include <stdio.h>
#include <stdlib.h>
#include "gm.h"
#include "gmx.h"

// One device mirror per host graph, keyed on the arrays it was uploaded from: the graph is frozen while a
// procedure runs (emitted prologue), so the arrays identify its contents until the next thaw()/reload.
static inline gmx_graph_t* gmx_mirror_of(gm_graph& G, bool need_reverse, const char* who) {
    struct slot { const edge_t* begin; const node_t* idx; const edge_t* r_begin; node_t n; edge_t m; gmx_graph_t* dev; };
    static slot cache = {NULL, NULL, NULL, 0, 0, NULL};
    const edge_t* rb = (need_reverse && G.has_reverse_edge()) ? G.r_begin : NULL;
    if (cache.dev && cache.begin == G.begin && cache.idx == G.node_idx && cache.n == G.num_nodes() && cache.m == G.num_edges() &&
        (rb == NULL || cache.r_begin == rb))
        return cache.dev;
    if (cache.dev) gmx_graph_free(cache.dev);
    cache.dev = NULL;
    // rows sorted already (load_binary leaves them so, gm_graph_binary_loader.cc:191-195): the host's reverse CSR is
    // taken as is; otherwise the device sorts the rows and builds the reverse CSR itself
    const bool sorted = G.is_semi_sorted();
    const uint32_t flags = sorted ? 0u : GMX_GRAPH_SORT_ROWS;
    if (gmx_graph_upload(G.begin, G.node_idx, sorted && G.has_reverse_edge() ? G.r_begin : NULL,
                         sorted && G.has_reverse_edge() ? G.r_node_idx : NULL, G.num_nodes(), G.num_edges(), flags, &cache.dev) != GMX_OK) {
        fprintf(stderr, "%s: %s\n", who, gmx_last_error());   // the reference has no error channel on this path
        abort();
    }
    cache.begin = G.begin;
    cache.idx = G.node_idx;
    cache.r_begin = G.has_reverse_edge() ? G.r_begin : NULL;
    cache.n = G.num_nodes();
    cache.m = G.num_edges();
    return cache.dev;
}

#define GMX_OR_DIE(who, call)                                   \
    do {                                                        \
        if ((call) != GMX_OK) {                                 \
            fprintf(stderr, "%s: %s\n", who, gmx_last_error()); \
            abort();                                            \
        }                                                       \
    } while (0)

#endif
