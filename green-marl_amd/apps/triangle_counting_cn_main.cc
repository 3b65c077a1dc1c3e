// triangle_counting_cn_main.cc -- driver of the common-neighbour form of triangle counting
// (generated/triangle_counting_cn.h); same command line and output line as triangle_counting_main.cc.
#include "common_main.h"
#include "triangle_counting_cn.h"

int main(int argc, char** argv) {
    int64_t triangles = 0;
    gm_app app;
    app.kernel([&](gm_graph& G) { triangles = triangle_counting_cn(G); return true; })
        .report([&](gm_graph&) {
            printf("number of triangles: %d\n", (int) triangles);
            if ((int64_t) (int) triangles != triangles) printf("number of triangles (64-bit): %lld\n", (long long) triangles);
            return true;
        });
    return app.exec(argc, argv);
}
