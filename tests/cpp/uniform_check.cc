// uniform_check.cc -- dumps what the host library's create_uniform_random_graph_new produces, for comparison with the
// arrays the compiled reference produced (graph_gen.cc:12-55; tests/golden, oracle/make_golden.py section 3b).
//   uniform_check <N> <M> <seed> <use_xorshift> <dump.txt>
#include <stdio.h>
#include <stdlib.h>
#include "gm.h"
#include "graph_gen.h"

int main(int argc, char** argv) {
    if (argc < 6) return 2;
    gm_graph G;
    create_uniform_random_graph_new(G, (node_t) atol(argv[1]), (edge_t) atol(argv[2]), atol(argv[3]), atoi(argv[4]) != 0);
    if (!G.is_frozen() || G.is_semi_sorted()) return 3;          // rows stay as generated, like the reference's
    FILE* f = fopen(argv[5], "w");
    if (!f) return 4;
    fprintf(f, "begin");
    for (node_t v = 0; v <= G.num_nodes(); v++) fprintf(f, " %d", G.begin[v]);
    fprintf(f, "\nnode_idx");
    for (edge_t e = 0; e < G.num_edges(); e++) fprintf(f, " %d", G.node_idx[e]);
    fprintf(f, "\n");
    fclose(f);
    gm_graph* g2 = create_uniform_random_graph((node_t) atol(argv[1]), (edge_t) atol(argv[2]), atol(argv[3]), atoi(argv[4]) != 0);
    for (edge_t e = 0; e < G.num_edges(); e++)
        if (g2->node_idx[e] != G.node_idx[e]) return 5;
    delete g2;
    return 0;
}
