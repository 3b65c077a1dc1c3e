// avg_teen_cnt.cc -- body of the generated `avg_teen_cnt` procedure, MI355X build.
// Emitted prologue: gm_rt_initialize(); G.freeze(); G.make_reverse_edges() (the .gm iterates InNbrs; the device
// counts the same edges from their source side, so only the forward CSR is read).
#include "avg_teen_cnt.h"
#include "gmx.h"

float avg_teen_cnt(gm_graph& G, int32_t* G_age, int32_t* G_teen_cnt, int32_t K) {
    gm_rt_initialize();
    G.freeze();
    G.make_reverse_edges();
    gmx_graph_t* dev = G.device_mirror();
    gmx_stats_t st;
    float avg = 0;
    if (dev == NULL || gmx_avg_teen_cnt(dev, G_age, K, G_teen_cnt, &avg, &st) != GMX_OK) {
        fprintf(stderr, "avg_teen_cnt: %s\n", gmx_last_error());
        abort();
    }
    gm_rt_cleanup();
    return avg;
}
