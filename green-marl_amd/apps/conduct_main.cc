// conduct_main.cc -- conductance benchmark driver; inputs and output of
// /root/reference/apps/output_cpp/src/conduct_main.cc (four groups of 10/20/30/40 % drawn from gm_rand32 in
// vertex order :27-38; conduct() summed over the groups :43-46; `sum C = %lf` :54).
#include "common_main.h"
#include "conduct.h"
#include "gm_rand.h"

int main(int argc, char** argv) {
    static const int upper[4] = {10, 30, 60, 100};   // cumulative percentages of the groups
    std::vector<int32_t> group;
    double total = 0;
    gm_app app;
    app.setup([&](gm_graph& G) {
            gm_rand32 rng;
            group.resize((size_t) G.num_nodes());
            for (int32_t& m : group) {
                const int32_t r = rng.rand() % 100;
                m = 0;
                while (r >= upper[m]) m++;
            }
            return true;
        })
        .kernel([&](gm_graph& G) {
            total = 0;
            for (int num = 0; num < 4; num++) total += conduct(G, group.data(), num);
            return true;
        })
        .report([&](gm_graph&) { printf("sum C = %lf\n", total); return true; });
    return app.exec(argc, argv);
}
