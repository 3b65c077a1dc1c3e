// gm_graph_typedef.h -- node/edge id types of the host graph API.
// Interface contract: /root/reference/apps/output_cpp/gm_graph/inc/gm_graph_typedef.h:8-49
// (node_t/edge_t widths, VALUE_TYPE codes, link-time size check).  The reference default is 32-bit ids.
// GM_EDGE64 (edge_t = int64_t, node_t = int32_t; apps/output_cpp/common.mk:39-64) is supported on the HOST side -- the
// graph class, the .bin format with 8-byte edge fields, the generated entries -- so that an application built for that
// configuration links and runs (make -C green-marl_amd host64 -> libgmgraph_e64.a); the device kernels keep 32-bit edge
// offsets, so a graph handed to them must have fewer than 2^31 - 2^27 edges (the C ABI's *_e64 calls refuse more).
// GM_NODE64 is rejected at compile time.
#ifndef GM_GRAPH_TYPEDEF_H
#define GM_GRAPH_TYPEDEF_H
#include <stdint.h>
#include <vector>

#if defined(GM_NODE64)
#error "libgmx is built for 32-bit node_t (the reference default, setup.mk NODE_SIZE=32)"
#endif

// ids are indices into the CSR arrays
using node_t = int32_t;
#ifdef GM_EDGE64
using edge_t = int64_t;
#else
using edge_t = int32_t;
#endif

// property value kinds, in the numbering the reference's loaders and generated code use
enum VALUE_TYPE {
    GMTYPE_BOOL = 0,
    GMTYPE_INT = 1,
    GMTYPE_LONG = 2,
    GMTYPE_FLOAT = 3,
    GMTYPE_DOUBLE = 4,
    GMTYPE_NODE = 5,
    GMTYPE_EDGE = 6,
    GMTYPE_END = 7
};

// property column containers (names used by generated code)
template <typename T> using gm_column = std::vector<T>;
using GM_BVECT = gm_column<bool>;
using GM_IVECT = gm_column<int32_t>;
using GM_LVECT = gm_column<int64_t>;
using GM_FVECT = gm_column<float>;
using GM_DVECT = gm_column<double>;
using GM_NVECT = gm_column<node_t>;
using GM_EVECT = gm_column<edge_t>;

// An application compiled for other id widths references a differently named symbol and fails to link
// (a library defines only the symbol of the configuration it was built for).
#ifdef GM_EDGE64
#define GM_SIZE_CHECK_VAR link_error_becuase_gm_graph_lib_is_configured_as_node32_edge64_but_the_application_is_not
#else
#define GM_SIZE_CHECK_VAR link_error_becuase_gm_graph_lib_is_configured_as_node32_edge32_but_the_application_is_not
#endif
extern int GM_SIZE_CHECK_VAR;
static inline void gm_graph_check_node_edge_size_at_link_time() { GM_SIZE_CHECK_VAR = 0; }

#endif
