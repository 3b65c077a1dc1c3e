// graph_gen.h -- synthetic graph constructors with the reference's signatures
// (/root/reference/apps/output_cpp/gm_graph/inc/graph_gen.h:6-10).  create_RMAT_graph runs the
// reference's generator (graph_gen.cc:159-287) on the MI355X, bit-compatible with its drand48 stream.
#ifndef GRAPH_GEN_H_
#define GRAPH_GEN_H_
#include "gm_graph.h"

gm_graph* create_RMAT_graph(node_t N, edge_t M, long rseed, double a, double b, double c, bool permute);
// Same generator, filling an existing object (MI355X addition used by the drivers' RMAT:<scale> input).
bool create_RMAT_graph_in(gm_graph& G, node_t N, edge_t M, long rseed, double a, double b, double c, bool permute);
gm_graph* create_uniform_random_graph(node_t N, edge_t M, long seed, bool use_xorshift_rng);
void create_uniform_random_graph_new(gm_graph& G, node_t N, edge_t M, long seed, bool use_xorshift_rng);

#endif
