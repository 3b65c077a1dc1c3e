// gm_graph.cc -- host graph container (clean-room; API contract in ../inc/gm_graph.h).
// Behavioural reference: /root/reference/apps/output_cpp/gm_graph/src/gm_graph.cc
//   freeze :68-113 (always semi-sorts afterwards), thaw :116-140, make_reverse_edges :205-304,
//   do_semi_sort :468-503, prepare_edge_source :426-459, get_edge_idx_for_src_dest :589-633.
#include "gm_graph.h"

#include <string.h>
#include <algorithm>
#include <numeric>

#include <omp.h>

#include "gmx.h"

// edge offsets cross the C ABI as edge_t: a GM_EDGE64 build of this library uses the 64-bit variants of the graph calls
#ifdef GM_EDGE64
#define gmx_graph_upload gmx_graph_upload_e64
#define gmx_graph_download gmx_graph_download_e64
#define gmx_graph_edge_order gmx_graph_edge_order_e64
#define gmx_graph_reverse_edge_map gmx_graph_reverse_edge_map_e64
#endif

int GM_SIZE_CHECK_VAR;

gm_graph::gm_graph()
    : begin(NULL), node_idx(NULL), node_idx_src(NULL), r_begin(NULL), r_node_idx(NULL), r_node_idx_src(NULL),
      e_idx2idx(NULL), e_rev2idx(NULL), _numNodes(0), _numEdges(0), _reverse_edge(false), _frozen(false),
      _directed(true), _semi_sorted(false), e_id2idx(NULL), e_idx2id(NULL), _dev(NULL) {}

gm_graph::~gm_graph() {
    drop_device_mirror();
    release_csr();
}

void gm_graph::release_csr() {
    delete[] begin; begin = NULL;
    delete[] node_idx; node_idx = NULL;
    delete[] node_idx_src; node_idx_src = NULL;
    delete[] r_begin; r_begin = NULL;
    delete[] r_node_idx; r_node_idx = NULL;
    delete[] r_node_idx_src; r_node_idx_src = NULL;
    delete[] e_idx2idx; e_idx2idx = NULL;
    delete[] e_rev2idx; e_rev2idx = NULL;
    delete[] e_id2idx; e_id2idx = NULL;
    delete[] e_idx2id; e_idx2id = NULL;
}

void gm_graph::drop_device_mirror() {
    if (_dev) gmx_graph_free(_dev);
    _dev = NULL;
}

void gm_graph::clear_graph(bool) {
    drop_device_mirror();
    _adj.clear();
    release_csr();
    _numNodes = 0;
    _numEdges = 0;
    _frozen = _reverse_edge = _semi_sorted = false;
}

void gm_graph::clear_graph() { clear_graph(true); }

// ---------------------------------------------------------------- editing
node_t gm_graph::add_node() {
    if (_frozen) thaw();
    _adj.resize((size_t) _numNodes + 1);
    return _numNodes++;
}

edge_t gm_graph::add_edge(node_t n, node_t m) {
    assert(is_node(n));
    assert(is_node(m));
    if (_frozen) thaw();
    edge_dest_t d;
    d.dest = m;
    d.edge = _numEdges;
    _adj[n].push_back(d);
    return _numEdges++;
}

bool gm_graph::has_edge(node_id from, node_id to) {
    assert(!_frozen);
    const std::vector<edge_dest_t>& row = _adj[from];
    for (size_t i = 0; i < row.size(); i++)
        if (row[i].dest == to) return true;
    return false;
}

void gm_graph::freeze() {
    if (_frozen) return;
    drop_device_mirror();
    const node_t N = _numNodes;
    const edge_t M = _numEdges;
    _adj.resize((size_t) N);
    begin = new edge_t[(size_t) N + 1];
    node_idx = new node_t[(size_t) M];
    e_idx2id = new edge_t[(size_t) M];
    e_id2idx = new edge_t[(size_t) M];
    edge_t pos = 0;
    for (node_t v = 0; v < N; v++) {
        begin[v] = pos;
        for (size_t k = 0; k < _adj[v].size(); k++, pos++) {
            node_idx[pos] = _adj[v][k].dest;
            e_idx2id[pos] = _adj[v][k].edge;
            e_id2idx[_adj[v][k].edge] = pos;
        }
    }
    begin[N] = pos;
    std::vector<std::vector<edge_dest_t> >().swap(_adj);
    _frozen = true;
    _semi_sorted = false;
    _reverse_edge = false;
    do_semi_sort();
}

void gm_graph::thaw() {
    if (!_frozen) return;
    drop_device_mirror();
    _adj.assign((size_t) _numNodes, std::vector<edge_dest_t>());
    for (node_t v = 0; v < _numNodes; v++)
        for (edge_t e = begin[v]; e < begin[v + 1]; e++) {
            edge_dest_t d;
            d.dest = node_idx[e];
            d.edge = get_edge_id(e);
            _adj[v].push_back(d);
        }
    release_csr();
    _frozen = _semi_sorted = _reverse_edge = false;
}

void gm_graph::prepare_external_creation(node_t n, edge_t m) { prepare_external_creation(n, m, true); }

void gm_graph::prepare_external_creation(node_t n, edge_t m, bool clean) {
    clear_graph(clean);
    begin = new edge_t[(size_t) n + 1];
    node_idx = new node_t[(size_t) m];
    _numNodes = n;
    _numEdges = m;
    _frozen = true;
}

// ---------------------------------------------------------------- derived indices
// Sort the entries of every row by destination, carrying up to two per-edge side arrays.
void gm_graph::sort_rows(edge_t* row_begin, node_t* dest, edge_t* aux, edge_t* aux2) {
    const node_t N = _numNodes;
#pragma omp parallel
    {
        std::vector<edge_t> order;
        std::vector<node_t> d;
        std::vector<edge_t> a, a2;
#pragma omp for schedule(dynamic, 1024)
        for (node_t v = 0; v < N; v++) {
            const edge_t lo = row_begin[v], n = row_begin[v + 1] - lo;
            if (n < 2) continue;
            bool sorted = true;
            for (edge_t i = 1; i < n && sorted; i++) sorted = dest[lo + i - 1] <= dest[lo + i];
            if (sorted) continue;
            order.resize(n);
            std::iota(order.begin(), order.end(), 0);
            const node_t* key = dest + lo;
            std::stable_sort(order.begin(), order.end(), [key](edge_t x, edge_t y) { return key[x] < key[y]; });
            d.assign(dest + lo, dest + lo + n);
            for (edge_t i = 0; i < n; i++) dest[lo + i] = d[order[i]];
            if (aux) {
                a.assign(aux + lo, aux + lo + n);
                for (edge_t i = 0; i < n; i++) aux[lo + i] = a[order[i]];
            }
            if (aux2) {
                a2.assign(aux2 + lo, aux2 + lo + n);
                for (edge_t i = 0; i < n; i++) aux2[lo + i] = a2[order[i]];
            }
        }
    }
}

void gm_graph::do_semi_sort() {
    if (!_frozen) freeze();
    if (_semi_sorted) return;
    drop_device_mirror();
    const edge_t M = _numEdges;
    e_idx2idx = new edge_t[(size_t) M];
#pragma omp parallel for
    for (edge_t e = 0; e < M; e++) e_idx2idx[e] = e;
    sort_rows(begin, node_idx, e_idx2idx, e_idx2id);
    if (e_id2idx != NULL) {
#pragma omp parallel for
        for (edge_t e = 0; e < M; e++) e_id2idx[e_idx2id[e]] = e;
    }
    if (_reverse_edge) sort_rows(r_begin, r_node_idx, e_rev2idx, NULL);
    _semi_sorted = true;
}

void gm_graph::make_reverse_edges() {
    if (_reverse_edge) return;
    if (!_frozen) freeze();
    const node_t N = _numNodes;
    const edge_t M = _numEdges;
    r_begin = new edge_t[(size_t) N + 1];
    r_node_idx = new node_t[(size_t) M];
    e_rev2idx = new edge_t[(size_t) M];
    // in-degree histogram, exclusive scan, then a scatter that walks sources in ascending order:
    // in-rows therefore come out sorted by source whether or not the forward rows are sorted.
    std::vector<edge_t> cursor((size_t) N + 1, 0);
    for (edge_t e = 0; e < M; e++) cursor[(size_t) node_idx[e] + 1]++;
    for (node_t v = 0; v < N; v++) cursor[(size_t) v + 1] += cursor[v];
    memcpy(r_begin, cursor.data(), sizeof(edge_t) * ((size_t) N + 1));
    for (node_t v = 0; v < N; v++)
        for (edge_t e = begin[v]; e < begin[v + 1]; e++) {
            const edge_t slot = cursor[node_idx[e]]++;
            r_node_idx[slot] = v;
            e_rev2idx[slot] = e;
        }
    _reverse_edge = true;
    if (node_idx_src != NULL && r_node_idx_src == NULL) {
        r_node_idx_src = new node_t[(size_t) M];
        for (node_t v = 0; v < N; v++)
            for (edge_t e = r_begin[v]; e < r_begin[v + 1]; e++) r_node_idx_src[e] = v;
    }
}

void gm_graph::prepare_edge_source() {
    assert(node_idx_src == NULL);
    const node_t N = _numNodes;
    node_idx_src = new node_t[(size_t) _numEdges];
#pragma omp parallel for schedule(static, 4096)
    for (node_t v = 0; v < N; v++)
        for (edge_t e = begin[v]; e < begin[v + 1]; e++) node_idx_src[e] = v;
    if (_reverse_edge) {
        r_node_idx_src = new node_t[(size_t) _numEdges];
#pragma omp parallel for schedule(static, 4096)
        for (node_t v = 0; v < N; v++)
            for (edge_t e = r_begin[v]; e < r_begin[v + 1]; e++) r_node_idx_src[e] = v;
    }
}

// ---------------------------------------------------------------- queries
edge_t gm_graph::get_edge_idx_for_src_dest(node_t src, node_t to) {
    // rows are semi-sorted: first slot holding `to`, or NIL_EDGE
    const node_t* lo = node_idx + begin[src];
    const node_t* hi = node_idx + begin[src + 1];
    const node_t* p = std::lower_bound(lo, hi, to);
    return (p != hi && *p == to) ? (edge_t) (p - node_idx) : NIL_EDGE;
}

bool gm_graph::is_neighbor(node_t src, node_t to) { return get_edge_idx_for_src_dest(src, to) != NIL_EDGE; }
bool gm_graph::has_edge_to(node_t source, node_t to) { return is_neighbor(source, to); }

// ---------------------------------------------------------------- device mirror
gmx_graph* gm_graph::device_mirror() {
    if (_dev) return _dev;
    if (!_frozen) freeze();
    uint32_t flags = _semi_sorted ? 0u : GMX_GRAPH_SORT_ROWS;
    gmx_graph_t* h = NULL;
    int st = gmx_graph_upload(begin, node_idx, (_reverse_edge && _semi_sorted) ? r_begin : NULL,
                              (_reverse_edge && _semi_sorted) ? r_node_idx : NULL, _numNodes, _numEdges, flags, &h);
    if (st != GMX_OK) {
        fprintf(stderr, "gm_graph: cannot create the device mirror: %s\n", gmx_last_error());
        return NULL;
    }
    _dev = h;
    return _dev;
}

bool gm_graph::adopt_device_graph(gmx_graph* dev) {
    clear_graph();
    const int64_t N = gmx_graph_num_nodes(dev), M = gmx_graph_num_edges(dev);
    if (N < 0 || M < 0) return false;
    begin = new edge_t[(size_t) N + 1];
    node_idx = new node_t[(size_t) M];
    r_begin = new edge_t[(size_t) N + 1];
    r_node_idx = new node_t[(size_t) M];
    if (gmx_graph_download(dev, begin, node_idx, r_begin, r_node_idx) != GMX_OK) {
        fprintf(stderr, "gm_graph: download of the device graph failed: %s\n", gmx_last_error());
        release_csr();
        return false;
    }
    _numNodes = (node_t) N;
    _numEdges = (edge_t) M;
    _frozen = _semi_sorted = _reverse_edge = true;
    _dev = dev;
    return true;
}
