// gm_graph.h -- host-side graph container with the reference's gm_graph public API.
//
// Clean-room implementation of the interface in
// /root/reference/apps/output_cpp/gm_graph/inc/gm_graph.h:119-445 (public CSR arrays :133-142,
// NIL ids :144-145, queries :147-181, freeze/thaw :186-187, make_reverse_edges / do_semi_sort /
// prepare_edge_source :195-197, add_node/add_edge :202-203, prepare_external_creation :242-243,
// load_binary/store_binary :244-245).  The generated kernels and the benchmark drivers touch the
// object only through these members, so code written against the reference compiles unchanged.
// What is new: the object can carry a device mirror (libgmx handle) that the generated entry
// points create on first use and that every mutation drops.
#ifndef GM_GRAPH_H_
#define GM_GRAPH_H_
#include <assert.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "gm_graph_typedef.h"

typedef node_t node_id;
typedef edge_t edge_id;

struct edge_dest_t { node_id dest; edge_id edge; };   // one entry of the editable adjacency form

struct gmx_graph;   // device mirror (include/gmx.h)

class gm_graph
{
  public:
    gm_graph();
    ~gm_graph();

    // ---- frozen (CSR) form: raw arrays, read directly by generated code ----
    edge_t* begin;           // [N+1] first out-edge of every node
    node_t* node_idx;        // [M]   destination of every out-edge
    node_t* node_idx_src;    // [M]   source of every out-edge       (prepare_edge_source)
    edge_t* r_begin;         // [N+1] first in-edge of every node    (make_reverse_edges)
    node_t* r_node_idx;      // [M]   source of every in-edge
    node_t* r_node_idx_src;  // [M]   destination of every in-edge   (prepare_edge_source)
    edge_t* e_idx2idx;       // [M]   sorted edge idx  -> edge idx at freeze time (do_semi_sort)
    edge_t* e_rev2idx;       // [M]   reverse edge idx -> forward edge idx

    static const node_t NIL_NODE = (node_t) -1;
    static const edge_t NIL_EDGE = (edge_t) -1;

    // membership queries on semi-sorted rows
    bool is_neighbor(node_t src, node_t to);
    bool has_edge_to(node_t source, node_t to);
    edge_t get_edge_idx_for_src_dest(node_t src, node_t dest);

    node_t num_nodes() { return _numNodes; }
    edge_t num_edges() { return _numEdges; }
    bool has_reverse_edge() { return _reverse_edge; }
    bool is_frozen() { return _frozen; }
    bool is_directed() { return _directed; }
    bool is_semi_sorted() { return _semi_sorted; }
    bool has_separate_edge_idx() { return e_id2idx != NULL; }
    bool is_edge_source_ready() { return node_idx_src != NULL; }

    // ---- form changes ----
    void thaw();                   // CSR -> editable adjacency lists
    void freeze();                 // editable -> CSR (then semi-sorts, as the reference does)
    void make_reverse_edges();     // in-edge CSR (freezes first)
    void do_semi_sort();           // sort every row by destination (freezes first)
    void prepare_edge_source();    // node_idx_src / r_node_idx_src

    // ---- editing (thaws a frozen graph) ----
    node_id add_node();
    edge_id add_edge(node_id n, node_id m);
    bool is_node(node_id n) { return n < _numNodes; }
    bool is_edge(edge_id e) { return e < _numEdges; }
    bool has_edge(node_id from, node_id to);               // editable form only
    edge_t get_num_edges(node_id from) { return begin[from + 1] - begin[from]; }

    // ---- external creation: caller fills begin[] / node_idx[] itself ----
    void prepare_external_creation(node_t n, edge_t m);
    void prepare_external_creation(node_t n, edge_t m, bool clean_key_id_mappings);

    // ---- binary graph format (big-endian; gm_graph_binary_loader.cc:19-40) ----
#define MAGIC_WORD_BIN 0x03939999
    bool store_binary(char* filename);
    bool load_binary(char* filename);   // also semi-sorts and builds reverse edges, like the reference

    // id <-> idx (node ids are their indices; edge ids are remembered across freeze)
    edge_t get_edge_idx(edge_id e) { return e_id2idx == NULL ? e : e_id2idx[e]; }
    node_t get_node_idx(node_id n) { return n; }
    edge_id get_edge_id(edge_t e) { return e_idx2id == NULL ? e : e_idx2id[e]; }
    node_id get_node_id(node_t n) { return n; }
    edge_t get_org_edge_idx(edge_id e) { return e_idx2idx == NULL ? e : e_idx2idx[e]; }

    void clear_graph();
    void clear_graph(bool clean_key_id_mappings);

    // ---- MI355X additions ----
    // Mark the CSR as already semi-sorted / reversed (for callers that filled the arrays themselves).
    void set_semi_sorted(bool v) { _semi_sorted = v; }
    // Adopt a graph that lives on the device (e.g. gmx_graph_create_rmat): downloads its CSR and
    // reverse CSR into this object and keeps the handle as the mirror.
    bool adopt_device_graph(gmx_graph* dev);
    // Device mirror used by the generated entry points; created on demand, NULL on failure.
    gmx_graph* device_mirror();
    void drop_device_mirror();

  private:
    void release_csr();
    bool build_reverse_on_device();   // load_binary: reverse CSR via the GPU, false -> caller uses the host path
    void sort_rows(edge_t* row_begin, node_t* dest, edge_t* aux, edge_t* aux2);

    node_t _numNodes;
    edge_t _numEdges;
    bool _reverse_edge, _frozen, _directed, _semi_sorted;
    std::vector<std::vector<edge_dest_t> > _adj;   // editable form
    edge_t* e_id2idx;
    edge_t* e_idx2id;
    gmx_graph* _dev;
};

#endif
