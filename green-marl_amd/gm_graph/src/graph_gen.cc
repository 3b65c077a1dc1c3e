// graph_gen.cc -- synthetic graphs (signatures: ../inc/graph_gen.h).
// create_RMAT_graph delegates to the device generator (gmx_graph_create_rmat), which reproduces
// /root/reference/apps/output_cpp/gm_graph/src/graph_gen.cc:159-287 draw for draw; the returned graph
// is already semi-sorted with reverse edges (the state load_binary leaves a graph in) and keeps the
// device copy as its mirror.  The uniform generators follow graph_gen.cc:12-105 (glibc rand()).
#include "graph_gen.h"

#include <stdlib.h>
#include <string.h>

#include "gm_rand.h"
#include "gmx.h"

bool create_RMAT_graph_in(gm_graph& G, node_t N, edge_t M, long rseed, double a, double b, double c, bool permute) {
    gmx_graph_t* dev = NULL;
    if (gmx_graph_create_rmat(N, M, rseed, a, b, c, permute ? 1 : 0, 0, &dev) != GMX_OK) {
        fprintf(stderr, "create_RMAT_graph: %s\n", gmx_last_error());
        return false;
    }
    if (!G.adopt_device_graph(dev)) {
        gmx_graph_free(dev);
        return false;
    }
    return true;
}

gm_graph* create_RMAT_graph(node_t N, edge_t M, long rseed, double a, double b, double c, bool permute) {
    gm_graph* g = new gm_graph();
    if (!create_RMAT_graph_in(*g, N, M, rseed, a, b, c, permute)) {
        delete g;
        return NULL;
    }
    return g;
}

void create_uniform_random_graph_new(gm_graph& G, node_t N, edge_t M, long seed, bool use_xorshift_rng) {
    gm_rand xr(seed);
    if (!use_xorshift_rng) srand((unsigned) seed);
    G.prepare_external_creation(N, M);
    node_t* src = new node_t[(size_t) M];
    node_t* dst = new node_t[(size_t) M];
    edge_t* deg = new edge_t[(size_t) N]();
    for (edge_t i = 0; i < M; i++) {
        node_t r = use_xorshift_rng ? (node_t) xr.rand() : (node_t) rand();
        src[i] = (node_t) (((r % N) + N) % N);
        r = use_xorshift_rng ? (node_t) xr.rand() : (node_t) rand();
        dst[i] = (node_t) (((r % N) + N) % N);
        deg[src[i]]++;
    }
    G.begin[0] = 0;
    for (node_t v = 0; v < N; v++) G.begin[v + 1] = G.begin[v] + deg[v];
    for (edge_t i = 0; i < M; i++) {   // rows fill from their last slot backwards (graph_gen.cc:45-52)
        const node_t u = src[i];
        G.node_idx[G.begin[u] + --deg[u]] = dst[i];
    }
    delete[] src;
    delete[] dst;
    delete[] deg;
}

gm_graph* create_uniform_random_graph(node_t N, edge_t M, long seed, bool use_xorshift_rng) {
    gm_graph* g = new gm_graph();
    create_uniform_random_graph_new(*g, N, M, seed, use_xorshift_rng);
    return g;
}
