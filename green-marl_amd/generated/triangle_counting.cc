// triangle_counting.cc -- body of the generated `triangle_counting` procedure, MI355X build
// (SURVEY.md section 8 a-3).  Emitted prologue: gm_rt_initialize(); G.freeze(); G.do_semi_sort();
// (HasEdgeTo marks the procedure NEED_SEMI_SORT, src/backend_cpp/gm_cpp_gen_misc_check.cc:40-47).
#include "triangle_counting.h"
#include "gmx.h"

int64_t triangle_counting(gm_graph& G) {
    gm_rt_initialize();
    G.freeze();
    G.do_semi_sort();
    gmx_graph_t* dev = G.device_mirror();
    gmx_stats_t st;
    int64_t T = 0;
    if (dev == NULL || gmx_triangle_counting(dev, &T, &st) != GMX_OK) {
        fprintf(stderr, "triangle_counting: %s\n", gmx_last_error());
        abort();
    }
    gm_rt_cleanup();
    return T;
}
