// triangle_counting_cn.cc -- body of `triangle_counting_cn`, MI355X build.  Emitted prologue: gm_rt_initialize();
// G.freeze(); G.do_semi_sort();  (CommonNbrs needs semi-sorted rows, like HasEdgeTo.)
#include "triangle_counting_cn.h"
#include "gmx.h"

int64_t triangle_counting_cn(gm_graph& G) {
    gm_rt_initialize();
    G.freeze();
    G.do_semi_sort();
    gmx_graph_t* dev = G.device_mirror();
    int64_t T = 0;
    gmx_stats_t st;
    if (dev == NULL || gmx_triangle_counting_cn(dev, &T, &st) != GMX_OK) {
        fprintf(stderr, "triangle_counting_cn: %s\n", gmx_last_error());
        abort();
    }
    gm_rt_cleanup();
    return T;
}
