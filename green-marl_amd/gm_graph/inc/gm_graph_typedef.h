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

typedef int32_t node_t;
typedef int32_t edge_t;
#define GM_SIZE_CHECK_VAR link_error_becuase_gm_graph_lib_is_configured_as_node32_edge32_but_the_application_is_not

enum VALUE_TYPE { GMTYPE_BOOL = 0, GMTYPE_INT, GMTYPE_LONG, GMTYPE_FLOAT, GMTYPE_DOUBLE, GMTYPE_NODE, GMTYPE_EDGE, GMTYPE_END };

typedef std::vector<double> GM_DVECT;
typedef std::vector<float> GM_FVECT;
typedef std::vector<bool> GM_BVECT;
typedef std::vector<int64_t> GM_LVECT;
typedef std::vector<int32_t> GM_IVECT;
typedef std::vector<node_t> GM_NVECT;
typedef std::vector<edge_t> GM_EVECT;

// Referencing this symbol makes an application built with other id widths fail to link.
extern int GM_SIZE_CHECK_VAR;
static inline void gm_graph_check_node_edge_size_at_link_time() { GM_SIZE_CHECK_VAR = 0; }

#endif
