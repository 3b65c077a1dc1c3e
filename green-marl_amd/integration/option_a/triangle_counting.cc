// Option A body of `triangle_counting` (apps/src/triangle_counting.gm; call site
// apps/output_cpp/src/triangle_counting_main.cc:14).  HasEdgeTo marks the procedure NEED_SEMI_SORT
// (gm_cpp_gen_misc_check.cc:40-47): the emitted prologue sorts the rows.
#include "triangle_counting.h"
#include "gmx_binding.h"

int64_t triangle_counting(gm_graph& G) {
    gm_rt_initialize();
    G.freeze();
    G.do_semi_sort();
    int64_t T = 0;
    GMX_OR_DIE("triangle_counting", gmx_triangle_counting(gmx_mirror_of(G, false, "triangle_counting"), &T, NULL));
    gm_rt_cleanup();
    return T;
}
