// conduct.cc -- body of the generated `conduct` procedure, MI355X build.
// Emitted prologue: gm_rt_initialize(); G.freeze();
#include "conduct.h"
#include "gmx.h"

float conduct(gm_graph& G, int32_t* G_member, int32_t num) {
    gm_rt_initialize();
    G.freeze();
    gmx_graph_t* dev = G.device_mirror();
    gmx_stats_t st;
    float c = 0;
    if (dev == NULL || gmx_conduct(dev, G_member, num, &c, &st) != GMX_OK) {
        fprintf(stderr, "conduct: %s\n", gmx_last_error());
        abort();
    }
    gm_rt_cleanup();
    return c;
}
