// gm_graph_check.cc -- exercises the host gm_graph API (no GPU needed) and dumps arrays for the
// Python test to compare with the reference-generated fixtures.
//   gm_graph_check <in.bin> <out.bin> <dump.txt>
#include <stdio.h>
#include <vector>
#include <stdlib.h>
#include "gm.h"

template <typename T>   // (edge_t arrays are int64 in a GM_EDGE64 build)
static void dump(FILE* f, const char* name, const T* a, long n) {
    fprintf(f, "%s", name);
    for (long i = 0; i < n; i++) fprintf(f, " %lld", (long long) a[i]);
    fprintf(f, "\n");
}

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    gm_graph_check_node_edge_size_at_link_time();
    gm_graph G;
    if (!G.load_binary(argv[1])) return 1;
    if (!G.is_frozen() || !G.is_semi_sorted() || !G.has_reverse_edge()) return 3;
    FILE* f = fopen(argv[3], "w");
    dump(f, "begin", G.begin, G.num_nodes() + 1);
    dump(f, "node_idx", G.node_idx, G.num_edges());
    dump(f, "r_begin", G.r_begin, G.num_nodes() + 1);
    dump(f, "r_node_idx", G.r_node_idx, G.num_edges());
    // e_rev2idx maps every reverse edge to a forward edge with swapped endpoints, one to one (copies of a
    // repeated edge in order)
    std::vector<char> seen((size_t) G.num_edges(), 0);
    for (node_t v = 0; v < G.num_nodes(); v++)
        for (edge_t e = G.r_begin[v]; e < G.r_begin[v + 1]; e++) {
            edge_t fe = G.e_rev2idx[e];
            if (fe < 0 || fe >= G.num_edges() || seen[fe]++) return 8;
            if (e > G.r_begin[v] && G.r_node_idx[e - 1] == G.r_node_idx[e] && G.e_rev2idx[e - 1] >= fe) return 9;
            if (G.node_idx[fe] != v) return 4;
            node_t src = G.r_node_idx[e];
            if (!(G.begin[src] <= fe && fe < G.begin[src + 1])) return 5;
        }
    // is_neighbor agrees with a linear scan
    int bad = 0;
    for (node_t v = 0; v < G.num_nodes() && v < 64; v++)
        for (node_t w = 0; w < G.num_nodes(); w++) {
            bool lin = false;
            for (edge_t e = G.begin[v]; e < G.begin[v + 1]; e++) lin |= (G.node_idx[e] == w);
            if (lin != G.is_neighbor(v, w) || lin != G.has_edge_to(v, w)) bad++;
            edge_t ei = G.get_edge_idx_for_src_dest(v, w);
            if (lin != (ei != gm_graph::NIL_EDGE)) bad++;
            if (ei != gm_graph::NIL_EDGE && G.node_idx[ei] != w) bad++;
        }
    if (bad) return 6;
    if (!G.store_binary(argv[2])) return 7;

    // editable form: build a graph edge by edge, freeze (semi-sorts), thaw, re-freeze
    gm_graph H;
    for (int i = 0; i < 6; i++) H.add_node();
    int es[][2] = {{0, 5}, {0, 1}, {0, 3}, {2, 1}, {2, 0}, {5, 4}, {0, 1}, {3, 3}};
    for (auto& e : es) H.add_edge(e[0], e[1]);
    if (!H.has_edge(0, 3) || H.has_edge(1, 0)) return 8;
    H.freeze();
    dump(f, "h_begin", H.begin, H.num_nodes() + 1);
    dump(f, "h_node_idx", H.node_idx, H.num_edges());
    if (H.get_num_edges(0) != 4) return 9;
    // edge ids survive the sort: id 2 was the edge 0->3
    if (H.node_idx[H.get_edge_idx(2)] != 3) return 10;
    H.make_reverse_edges();
    dump(f, "h_r_begin", H.r_begin, H.num_nodes() + 1);
    dump(f, "h_r_node_idx", H.r_node_idx, H.num_edges());
    H.prepare_edge_source();
    dump(f, "h_node_idx_src", H.node_idx_src, H.num_edges());
    dump(f, "h_r_node_idx_src", H.r_node_idx_src, H.num_edges());
    edge_t id = H.add_edge(4, 2);   // thaws
    if (H.is_frozen() || id != 8) return 11;
    H.freeze();
    if (H.num_edges() != 9 || !H.is_neighbor(4, 2)) return 12;
    // external creation path
    gm_graph E;
    E.prepare_external_creation(3, 2);
    E.begin[0] = 0; E.begin[1] = 2; E.begin[2] = 2; E.begin[3] = 2;
    E.node_idx[0] = 2; E.node_idx[1] = 1;
    E.do_semi_sort();
    if (E.node_idx[0] != 1 || E.node_idx[1] != 2 || E.get_org_edge_idx(0) != 1) return 13;
    gm_rand32 r;
    int32_t r1 = r.rand(), r2 = r.rand(), r3 = r.rand();
    fprintf(f, "rand32 %d %d %d\n", r1, r2, r3);
    fclose(f);
    gm_rt_set_num_threads(2);
    if (gm_rt_get_num_threads() != 2) return 14;
    return 0;
}
