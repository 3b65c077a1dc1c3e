// hop_dist_main.cc -- BFS (hop distance) benchmark driver; command line and output of
// /root/reference/apps/output_cpp/src/hop_dist_main.cc (root 0 :21, dist[0..9] :36-38).
// Addition: an optional app argument overrides the root.
#include "common_main.h"
#include "hop_dist.h"

int main(int argc, char** argv) {
    node_t root = 0;
    std::vector<int32_t> dist;
    gm_app app;
    app.usage("[root=0]")
        .args([&](const std::vector<std::string>& a) {
            if (!a.empty()) root = (node_t) atol(a[0].c_str());
            return true;
        })
        .setup([&](gm_graph& G) { dist.assign((size_t) G.num_nodes(), 0); return true; })
        .kernel([&](gm_graph& G) { hop_dist(G, dist.data(), root); return true; })
        .report([&](gm_graph& G) {
            for (node_t v = 0; v < 10 && v < G.num_nodes(); v++) printf("dist[%d] = %d\n", (int) v, dist[v]);
            return true;
        });
    return app.exec(argc, argv);
}
