// bc_main.cc -- betweenness-centrality benchmark driver; command line and output of
// /root/reference/apps/output_cpp/src/bc_main.cc (5 seeds drawn from gm_rand32 :34-41, BC[0..3] printed with
// %0.9lf :47-53).
#include "common_main.h"
#include "bc.h"

int main(int argc, char** argv) {
    std::vector<float> BC;
    gm_node_seq Seeds;
    gm_app app;
    app.usage("")
        .setup([&](gm_graph& G) { BC.assign((size_t) G.num_nodes(), 0.0f); return true; })
        .kernel([&](gm_graph& G) {
            gm_rand32 xorshift_rng;
            for (int i = 0; i < 5; i++) {   // pick 5 random starting points
                node_t t;
                do {
                    t = xorshift_rng.rand();
                } while (t >= G.num_nodes() || t < 0);
                Seeds.push_back(t);
            }
            comp_BC(G, BC.data(), Seeds);
            return true;
        })
        .report([&](gm_graph& G) {
            for (node_t v = 0; v < 4 && v < G.num_nodes(); v++) printf("BC[%d] = %0.9lf\n", (int) v, (double) BC[v]);
            return true;
        });
    return app.exec(argc, argv);
}
