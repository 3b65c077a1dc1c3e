// gm_runtime.cc -- thread-count shim (contract: ../inc/gm_runtime.h).
#include "gm_runtime.h"

static int g_threads = 0;
static bool g_init = false;

void gm_rt_initialize() {
    if (g_init) return;
    g_init = true;
    if (g_threads <= 0) g_threads = omp_get_max_threads();
}
bool gm_rt_is_initialized() { return g_init; }
int gm_rt_get_num_threads() { return g_threads > 0 ? g_threads : omp_get_max_threads(); }
void gm_rt_set_num_threads(int n) {
    if (n <= 0) return;
    g_threads = n;
    omp_set_num_threads(n);
}
int gm_rt_thread_id() { return omp_get_thread_num(); }
void gm_rt_cleanup() {}
