// gm_graph_binary.cc -- the custom binary graph format.
// Format (all big-endian; /root/reference/apps/output_cpp/gm_graph/src/gm_graph_binary_loader.cc:19-26):
//   magic 0x03939999 | sizeof(node_t) | sizeof(edge_t) | N | M | begin[N+1] | node_idx[M]
// (N, node_idx: node_t-sized fields; M, begin: edge_t-sized fields -- 8 bytes in a GM_EDGE64 build)
// A byte-flipped legacy variant (magic readable without swapping) is accepted too (:68-87).
// Like the reference (:191-197) a loaded graph is semi-sorted and gets its reverse edges.
// The reference reads element by element; this reader pulls each array with one fread and swaps in parallel.
#include <arpa/inet.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>

#include <stdlib.h>
#include <sys/time.h>

#include "gm_graph.h"
#include "gmx.h"

// edge offsets cross the C ABI as edge_t: a GM_EDGE64 build of this library uses the 64-bit variants of the graph calls
#ifdef GM_EDGE64
#define gmx_graph_upload gmx_graph_upload_e64
#define gmx_graph_download gmx_graph_download_e64
#define gmx_graph_edge_order gmx_graph_edge_order_e64
#define gmx_graph_reverse_edge_map gmx_graph_reverse_edge_map_e64
#endif

// GMX_LOAD_TIMING=1: where load_binary spends its time (stderr)
static double load_now_ms() {
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return tv.tv_sec * 1e3 + tv.tv_usec * 1e-3;
}
static void load_mark(const char* what, double& t) {
    if (!getenv("GMX_LOAD_TIMING")) return;
    const double n = load_now_ms();
    fprintf(stderr, "load_binary: %-28s %8.1f ms\n", what, n - t);
    t = n;
}

static bool read_u32_array(FILE* f, int32_t* dst, size_t n, bool swap) {
    if (n && fread(dst, 4, n, f) != n) return false;
    if (swap) {
#pragma omp parallel for
        for (size_t i = 0; i < n; i++) dst[i] = (int32_t) ntohl((uint32_t) dst[i]);
    }
    return true;
}

static inline uint64_t swap_u64(uint64_t v) { return ((uint64_t) ntohl((uint32_t) v) << 32) | ntohl((uint32_t) (v >> 32)); }

// n edge-typed fields of `esz` bytes each (4 or 8; a library built with a wider edge_t reads the narrower files too,
// like the reference's BITS_TO_EDGE, gm_graph_binary_loader.cc:118-126)
static bool read_edge_array(FILE* f, edge_t* dst, size_t n, uint32_t esz, bool swap) {
    if (esz == 4 && sizeof(edge_t) == 4) return read_u32_array(f, (int32_t*) dst, n, swap);
    if (esz == 4) {   // widen in place, back to front
        if (n && fread(dst, 4, n, f) != n) return false;
        const int32_t* narrow = (const int32_t*) dst;
        for (size_t i = n; i-- > 0;) {
            const int32_t v = narrow[i];
            dst[i] = (edge_t) (swap ? (int32_t) ntohl((uint32_t) v) : v);
        }
        return true;
    }
    if (sizeof(edge_t) != 8) return false;
    if (n && fread(dst, 8, n, f) != n) return false;
    if (swap) {
#pragma omp parallel for
        for (size_t i = 0; i < n; i++) dst[i] = (edge_t) swap_u64((uint64_t) dst[i]);
    }
    return true;
}

bool gm_graph::load_binary(char* filename) {
    clear_graph();
    FILE* f = fopen(filename, "rb");
    if (f == NULL) {
        fprintf(stderr, "cannot open %s for reading\n", filename);
        return false;
    }
    double tmark = load_now_ms();
    uint32_t hdr[3];
    bool ok = fread(hdr, 4, 3, f) == 3;
    bool swap = true;
    if (ok) {
        if (ntohl(hdr[0]) == MAGIC_WORD_BIN) swap = true;
        else if (hdr[0] == MAGIC_WORD_BIN) swap = false;
        else {
            fprintf(stderr, "wrong file format, KEY mismatch: %08x, expected:%08x\n", ntohl(hdr[0]), MAGIC_WORD_BIN);
            ok = false;
        }
    } else fprintf(stderr, "wrong file format\n");
    uint32_t nsz = 0, esz = 0;
    if (ok) {
        nsz = swap ? ntohl(hdr[1]) : hdr[1];
        esz = swap ? ntohl(hdr[2]) : hdr[2];
        // (the reference accepts files whose fields are narrower than the library's types, :93-101)
        if (nsz != sizeof(node_t) || (esz != 4 && esz != 8) || esz > sizeof(edge_t)) {
            fprintf(stderr, "node_t/edge_t size mismatch: file has %u/%u bytes, library expects %zu/%zu; please re-generate the graph\n",
                    nsz, esz, sizeof(node_t), sizeof(edge_t));
            ok = false;
        }
    }
    int32_t n_file = 0;
    edge_t m_file = 0;
    if (ok) ok = read_u32_array(f, &n_file, 1, swap) && read_edge_array(f, &m_file, 1, esz, swap);
    if (ok && (n_file < 0 || m_file < 0)) ok = false;
    if (ok) {
        printf("N = %ld, M = %ld\n", (long) n_file, (long) m_file);
        prepare_external_creation(n_file, m_file);
        ok = read_edge_array(f, begin, (size_t) n_file + 1, esz, swap) && read_u32_array(f, node_idx, (size_t) m_file, swap);
        if (!ok) fprintf(stderr, "Error reading the CSR arrays\n");
        else if (begin[0] != 0 || begin[n_file] != m_file) {
            fprintf(stderr, "corrupt file: begin[] does not cover the edge array\n");
            ok = false;
        }
    }
    fclose(f);
    if (!ok) {
        clear_graph();
        return false;
    }
    load_mark("read + byte swap", tmark);
    // Post-load preparation (the reference: per-row std::sort + atomic scatter, both on the host,
    // gm_graph_binary_loader.cc:191-197).  With a GPU present both happen on the device (SURVEY.md section 8f
    // rank 1): do_semi_sort as a 64-bit (row, column) key radix sort that carries the slot numbers (e_idx2idx; a
    // file whose rows are already in order -- anything store_binary wrote after a semi-sort -- is recognised by one
    // pass over the keys and keeps the identity map), make_reverse_edges as the same sort over the transposed keys;
    // the device copy stays as the mirror the kernels use.  Without a device: the host versions.
    int ndev = 0;
    if (_numEdges > 0 && e_idx2id == NULL && gmx_device_count(&ndev) == GMX_OK && ndev > 0 && build_reverse_on_device()) return true;
    do_semi_sort();
    load_mark("do_semi_sort (host)", tmark);
    make_reverse_edges();
    return true;
}

bool gm_graph::build_reverse_on_device() {
    gmx_graph_t* dev = NULL;
    double tmark = load_now_ms();
    const uint32_t flags = _semi_sorted ? 0u : GMX_GRAPH_SORT_ROWS;
    if (gmx_graph_upload(begin, node_idx, NULL, NULL, _numNodes, _numEdges, flags, &dev) != GMX_OK) {
        fprintf(stderr, "load_binary: device preparation failed (%s), using the host path\n", gmx_last_error());
        return false;
    }
    load_mark("upload + device row sort + reverse CSR", tmark);
    const node_t N = _numNodes;
    const edge_t M = _numEdges;
    if (!_semi_sorted) {   // do_semi_sort (gm_graph.cc:468-503): sorted rows and the map to the slots of the file
        int ident = 1;
        edge_t* map = new edge_t[(size_t) M];
        bool ok = gmx_graph_edge_order(dev, map, &ident) == GMX_OK;
        if (ok && ident) {
#pragma omp parallel for
            for (edge_t e = 0; e < M; e++) map[e] = e;
        } else if (ok) ok = gmx_graph_download(dev, NULL, node_idx, NULL, NULL) == GMX_OK;
        if (!ok) {
            fprintf(stderr, "load_binary: device row sort failed (%s), using the host path\n", gmx_last_error());
            delete[] map;
            gmx_graph_free(dev);
            return false;
        }
        delete[] e_idx2idx;
        e_idx2idx = map;
        _semi_sorted = true;
        load_mark(ident ? "rows already in order: identity e_idx2idx" : "download sorted rows + e_idx2idx", tmark);
    }
    r_begin = new edge_t[(size_t) N + 1];
    r_node_idx = new node_t[(size_t) M];
    e_rev2idx = new edge_t[(size_t) M];
    load_mark("host allocations", tmark);
    if (gmx_graph_download(dev, NULL, NULL, r_begin, r_node_idx) != GMX_OK) {
        fprintf(stderr, "load_binary: download failed (%s), using the host path\n", gmx_last_error());
        gmx_graph_free(dev);
        delete[] r_begin; r_begin = NULL;
        delete[] r_node_idx; r_node_idx = NULL;
        delete[] e_rev2idx; e_rev2idx = NULL;
        return false;
    }
    load_mark("download reverse CSR", tmark);
    // reverse slot -> forward slot (the k-th copy of (w -> t) in the in-row of t mirrors the k-th slot holding t
    // in row w): the device sorts the forward slot numbers along with the (dst, src) keys
    if (gmx_graph_reverse_edge_map(dev, e_rev2idx) != GMX_OK) {
        fprintf(stderr, "load_binary: edge map failed (%s), using the host path\n", gmx_last_error());
        gmx_graph_free(dev);
        delete[] r_begin; r_begin = NULL;
        delete[] r_node_idx; r_node_idx = NULL;
        delete[] e_rev2idx; e_rev2idx = NULL;
        return false;
    }
    load_mark("e_rev2idx (device) + download", tmark);
    _reverse_edge = true;
    drop_device_mirror();
    _dev = dev;
    return true;
}

bool gm_graph::store_binary(char* filename) {
    if (!_frozen) freeze();
    FILE* f = fopen(filename, "wb");
    if (f == NULL) {
        fprintf(stderr, "cannot open %s for writing\n", filename);
        return false;
    }
    const size_t N = (size_t) _numNodes, M = (size_t) _numEdges;
    uint32_t hdr[4] = {htonl(MAGIC_WORD_BIN), htonl((uint32_t) sizeof(node_t)), htonl((uint32_t) sizeof(edge_t)), htonl((uint32_t) _numNodes)};
    bool ok = fwrite(hdr, 4, 4, f) == 4;
    const size_t chunk = 1 << 20;
    uint64_t* buf = new uint64_t[chunk];
    // an edge-typed field: sizeof(edge_t) bytes, big-endian (gm_graph_binary_loader.cc:238-244)
    auto put_edges = [&](const edge_t* src, size_t n) {
        for (size_t off = 0; off < n && ok; off += chunk) {
            const size_t c = n - off < chunk ? n - off : chunk;
            if (sizeof(edge_t) == 4) {
                uint32_t* b32 = (uint32_t*) buf;
                for (size_t i = 0; i < c; i++) b32[i] = htonl((uint32_t) src[off + i]);
            } else {
                for (size_t i = 0; i < c; i++) buf[i] = swap_u64((uint64_t) src[off + i]);
            }
            ok = fwrite(buf, sizeof(edge_t), c, f) == c;
        }
    };
    const edge_t m_field = _numEdges;
    put_edges(&m_field, 1);
    put_edges(begin, N + 1);
    for (size_t off = 0; off < M && ok; off += chunk) {
        const size_t c = M - off < chunk ? M - off : chunk;
        uint32_t* b32 = (uint32_t*) buf;
        for (size_t i = 0; i < c; i++) b32[i] = htonl((uint32_t) node_idx[off + i]);
        ok = fwrite(b32, 4, c, f) == c;
    }
    delete[] buf;
    fclose(f);
    return ok;
}
