// avg_teen_cnt_main.cc -- "average number of teenage followers" benchmark driver; inputs and output of
// /root/reference/apps/output_cpp/src/avg_teen_cnt_main.cc (every age 10 :16-18; K = 5 unless given :46-50;
// `avg = %0.9lf` :34).
#include "common_main.h"
#include "avg_teen_cnt.h"

int main(int argc, char** argv) {
    int K = 5;
    float avg = 0;
    std::vector<int32_t> age, teen_cnt;
    gm_app app;
    app.usage("[K=5]")
        .args([&](const std::vector<std::string>& a) { return a.empty() || (K = atoi(a[0].c_str())) > 0; })
        .setup([&](gm_graph& G) {
            age.assign((size_t) G.num_nodes(), 10);
            teen_cnt.assign((size_t) G.num_nodes(), 0);
            return true;
        })
        .kernel([&](gm_graph& G) { avg = avg_teen_cnt(G, age.data(), teen_cnt.data(), K); return true; })
        .report([&](gm_graph&) { printf("avg = %0.9lf\n", avg); return true; });
    return app.exec(argc, argv);
}
