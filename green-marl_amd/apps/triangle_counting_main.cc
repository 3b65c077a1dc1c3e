// triangle_counting_main.cc -- triangle counting benchmark driver; output line of
// /root/reference/apps/output_cpp/src/triangle_counting_main.cc:19 (`number of triangles: %d`: the reference
// narrows the int64 result to int; the same line is printed so its checker sees it, plus the full count when
// it does not fit).
#include "common_main.h"
#include "triangle_counting.h"

int main(int argc, char** argv) {
    int64_t triangles = 0;
    gm_app app;
    app.kernel([&](gm_graph& G) { triangles = triangle_counting(G); return true; })
        .report([&](gm_graph&) {
            printf("number of triangles: %d\n", (int) triangles);
            if ((int64_t) (int) triangles != triangles) printf("number of triangles (64-bit): %lld\n", (long long) triangles);
            return true;
        });
    return app.exec(argc, argv);
}
