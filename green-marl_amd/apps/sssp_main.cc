// sssp_main.cc -- single-source shortest paths benchmark driver; inputs and output of
// /root/reference/apps/output_cpp/src/sssp_main.cc (root 0 :20; edge lengths (rand % 100) + 1 drawn from
// gm_rand32 in slot order :33-34; dist[0..9] :50-52).  Addition: an optional app argument overrides the root.
#include "common_main.h"
#include "sssp.h"
#include "gm_rand.h"

int main(int argc, char** argv) {
    node_t root = 0;
    std::vector<int32_t> length, dist;
    gm_app app;
    app.usage("[root=0]")
        .args([&](const std::vector<std::string>& a) {
            if (!a.empty()) root = (node_t) atol(a[0].c_str());
            return true;
        })
        .setup([&](gm_graph& G) {
            gm_rand32 rng;
            dist.assign((size_t) G.num_nodes(), 0);
            length.resize((size_t) G.num_edges());
            for (int32_t& l : length) l = (rng.rand() % 100) + 1;   // 1 .. 100
            return true;
        })
        .kernel([&](gm_graph& G) { sssp(G, dist.data(), length.data(), root); return true; })
        .report([&](gm_graph& G) {
            for (node_t v = 0; v < 10 && v < G.num_nodes(); v++) printf("dist[%d] = %d\n", (int) v, dist[v]);
            return true;
        });
    return app.exec(argc, argv);
}
