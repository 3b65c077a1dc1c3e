// gm_graph_typedef.h -- node/edge id types of the host graph API.
// Interface contract: /root/reference/apps/output_cpp/gm_graph/inc/gm_graph_typedef.h:8-49
// (node_t/edge_t widths, VALUE_TYPE codes, link-time size check).  The MI355X kernels are built for
// the reference default, 32-bit ids; the 64-bit configurations are rejected at compile time.
#ifndef GM_GRAPH_TYPEDEF_H
#define GM_GRAPH_TYPEDEF_H
#include <stdint.h>
#include <vector>

#if defined(GM_NODE64) || defined(GM_EDGE64)
#error "libgmx is built for 32-bit node_t/edge_t (the reference default, setup.mk NODE_SIZE=32 EDGE_SIZE=32)"
#endif

// ids are indices into the CSR arrays
using node_t = int32_t;
using edge_t = int32_t;

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
// (the library defines only the 32/32 one).
#define GM_SIZE_CHECK_VAR link_error_becuase_gm_graph_lib_is_configured_as_node32_edge32_but_the_application_is_not
extern int GM_SIZE_CHECK_VAR;
static inline void gm_graph_check_node_edge_size_at_link_time() { GM_SIZE_CHECK_VAR = 0; }

#endif
