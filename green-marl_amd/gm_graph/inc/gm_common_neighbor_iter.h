// gm_common_neighbor_iter.h -- host side of `Foreach(u: s.CommonNbrs(d))`
// (/root/reference/apps/output_cpp/gm_graph/inc/gm_common_neighbor_iter.h:8-49, src/gm_common_neighbor_iter.cc:3-44):
// the slots of s's row, in order and with their multiplicity, whose value occurs in d's row; rows semi-sorted.
// The device form is gmx_common_nbrs / gmx_common_nbr_counts / gmx_triangle_counting_cn (include/gmx.h).
#ifndef GM_NEIGHBOR_ITER_H
#define GM_NEIGHBOR_ITER_H
#include "gm_graph.h"

class gm_common_neighbor_iter
{
  public:
    gm_common_neighbor_iter(gm_graph& g, node_t s, node_t d)
        : G(g), s_lo(g.begin[s]), s_hi(g.begin[s + 1]), d_lo(g.begin[d]), d_hi(g.begin[d + 1]) { reset(); }
    void reset() {
        s_at = s_lo;
        d_at = d_lo;
        done = s_lo == s_hi || d_lo == d_hi;
    }
    node_t get_next() {
        while (!done) {
            const node_t t = G.node_idx[s_at++];
            if (s_at == s_hi) done = true;              // the value in hand is still judged
            while (G.node_idx[d_at] < t) {
                if (++d_at == d_hi) {                   // d's row is exhausted: nothing further can match
                    done = true;
                    return gm_graph::NIL_NODE;
                }
            }
            if (G.node_idx[d_at] == t) return t;
        }
        return gm_graph::NIL_NODE;
    }

  private:
    gm_graph& G;
    edge_t s_lo, s_hi, d_lo, d_hi, s_at, d_at;
    bool done;
};
#endif
