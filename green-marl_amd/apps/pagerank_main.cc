// pagerank_main.cc -- PageRank benchmark driver.  Command line, defaults and output lines are those of
// /root/reference/apps/output_cpp/src/pagerank_main.cc (e=0.001 d=0.85 max=100 :11-16, positional overrides
// :51-66, rank[0..3] printed with %0.9lf :36-38), so scripts/extract_result.py reads it unchanged.
// Addition: a 4th app argument "f32" selects the Node_Prop<Float> entry (BASELINE config 2).
#include "common_main.h"
#include "pagerank.h"

int main(int argc, char** argv) {
    int max_iter = 100;
    double eps = 0.001, damping = 0.85;
    bool f32 = false;
    std::vector<double> rank64;
    std::vector<float> rank32;
    gm_app app;
    app.usage("[max_iteration=100] [eplision=0.001] [delta=0.85] [f32]")
        .args([&](const std::vector<std::string>& a) {
            if (a.size() > 0 && (max_iter = atoi(a[0].c_str())) <= 0) return false;
            if (a.size() > 1 && (eps = atof(a[1].c_str())) <= 0) return false;
            if (a.size() > 2 && (damping = atof(a[2].c_str())) <= 0) return false;
            f32 = a.size() > 3 && a[3] == "f32";
            return true;
        })
        .setup([&](gm_graph& G) {
            if (f32) rank32.assign((size_t) G.num_nodes(), 0.f);
            else rank64.assign((size_t) G.num_nodes(), 0.0);
            return true;
        })
        .kernel([&](gm_graph& G) {
            if (f32) pagerank(G, (float) eps, (float) damping, max_iter, rank32.data());
            else pagerank(G, eps, damping, max_iter, rank64.data());
            return true;
        })
        .report([&](gm_graph& G) {
            for (node_t v = 0; v < 4 && v < G.num_nodes(); v++)
                printf("rank[%d] = %0.9lf\n", (int) v, f32 ? (double) rank32[v] : rank64[v]);
            return true;
        });
    return app.exec(argc, argv);
}
