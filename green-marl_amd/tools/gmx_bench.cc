// gmx_bench.cc -- tiny C++ check of the C ABI without Python: device info + RMAT-12 round trip.
#include <stdio.h>
#include <vector>
#include "gmx.h"

int main() {
    int n = 0;
    if (gmx_device_count(&n) != GMX_OK || n < 1) {
        fprintf(stderr, "no device: %s\n", gmx_last_error());
        return 2;
    }
    gmx_device_info_t info;
    gmx_device_info(&info);
    printf("device: %s %s, %d CUs\n", info.name, info.arch, info.compute_units);
    gmx_graph_t* g = NULL;
    if (gmx_graph_create_rmat(1 << 12, 16 << 12, 1997, 0.57, 0.19, 0.19, 0, 0, &g) != GMX_OK) {
        fprintf(stderr, "rmat: %s\n", gmx_last_error());
        return 1;
    }
    std::vector<double> rank(1 << 12);
    gmx_stats_t st;
    if (gmx_pagerank_f64(g, 0.001, 0.85, 100, rank.data(), &st) != GMX_OK) return 1;
    printf("pagerank: %d iterations, rank[0] = %0.9lf, %.3f ms\n", st.iterations, rank[0], st.kernel_ms);
    gmx_graph_free(g);
    return 0;
}
