// unsorted_check.cc -- a host gm_graph whose rows are NOT semi-sorted (prepare_external_creation with caller-filled
// rows, what create_uniform_random_graph_new leaves behind, graph_gen.cc:12-105):
//   1. store_binary / load_binary: the loader semi-sorts (device radix sort with a GPU, host sort without) and must
//      leave e_idx2idx mapping every sorted slot to its slot in the file (gm_graph.cc:468-503);
//   2. (with "gpu" as argv[2]) sssp through the generated entry on the UNSORTED graph: the edge property is
//      indexed by the caller's slots, the device mirror sorts its rows -- distances must equal a host
//      Bellman-Ford over the rows as stored (the reference's sssp never sorts).
//   unsorted_check <tmp.bin> [gpu]
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "gm.h"
#include "sssp.h"

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const node_t N = 5000;
    const edge_t M = 60000;
    gm_graph G;
    G.prepare_external_creation(N, M);
    unsigned long long x = 88172645463325252ULL;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (unsigned) (x >> 11); };
    std::vector<int> deg(N, 0);
    for (edge_t e = 0; e < M; e++) deg[rnd() % N]++;
    G.begin[0] = 0;
    for (node_t v = 0; v < N; v++) G.begin[v + 1] = G.begin[v] + deg[v];
    for (edge_t e = 0; e < M; e++) G.node_idx[e] = (node_t) (rnd() % N);   // arrival order: rows unsorted, duplicates kept
    std::vector<node_t> file_idx(G.node_idx, G.node_idx + M);
    if (G.is_semi_sorted()) return 3;
    if (!G.store_binary(argv[1])) return 4;

    gm_graph H;
    if (!H.load_binary(argv[1])) return 5;
    if (!H.is_semi_sorted() || !H.has_reverse_edge() || H.e_idx2idx == NULL) return 6;
    std::vector<char> seen(M, 0);
    for (node_t v = 0; v < N; v++)
        for (edge_t e = H.begin[v]; e < H.begin[v + 1]; e++) {
            if (H.begin[v] != G.begin[v]) return 7;
            if (e > H.begin[v] && H.node_idx[e - 1] > H.node_idx[e]) return 8;           // ascending rows
            const edge_t o = H.e_idx2idx[e];
            if (o < G.begin[v] || o >= G.begin[v + 1] || seen[o]++) return 9;             // a permutation inside the row
            if (file_idx[o] != H.node_idx[e]) return 10;                                   // sorted slot -> slot of the file
        }
    for (node_t v = 0; v < N; v++)                                                         // reverse CSR of the sorted rows
        for (edge_t e = H.r_begin[v]; e < H.r_begin[v + 1]; e++)
            if (H.node_idx[H.e_rev2idx[e]] != v) return 11;
    printf("load_binary of unsorted rows ok\n");

    if (argc > 2) {
        std::vector<int32_t> len(M), dist(N), want(N, INT_MAX);
        for (edge_t e = 0; e < M; e++) len[e] = 1 + (int32_t) (rnd() % 40);
        node_t root = 17;
        sssp(G, dist.data(), len.data(), root);           // G: frozen, rows as stored
        want[root] = 0;
        for (bool changed = true; changed;) {
            changed = false;
            for (node_t v = 0; v < N; v++) {
                if (want[v] == INT_MAX) continue;
                for (edge_t e = G.begin[v]; e < G.begin[v + 1]; e++)
                    if (want[v] + len[e] < want[G.node_idx[e]]) { want[G.node_idx[e]] = want[v] + len[e]; changed = true; }
            }
        }
        for (node_t v = 0; v < N; v++)
            if (dist[v] != want[v]) { fprintf(stderr, "sssp: dist[%d] = %d, expected %d\n", v, dist[v], want[v]); return 12; }
        if (G.is_semi_sorted()) return 13;                 // sssp must not have touched the host rows
        printf("sssp on an unsorted host graph ok\n");
    }
    return 0;
}
