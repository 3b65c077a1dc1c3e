// pagerank_main.cc -- PageRank benchmark driver, same protocol and output lines as
// /root/reference/apps/output_cpp/src/pagerank_main.cc (defaults e=0.001 d=0.85 max=100 :11-16,
// positional overrides :51-66, prints rank[0..3] with %0.9lf :36-38).
// Addition: a 4th optional app argument "f32" selects the Node_Prop<Float> entry (BASELINE config 2).
#include "common_main.h"
#include "pagerank.h"

class my_main : public main_t
{
  public:
    double* rank;
    float* rank32;
    int max_iter;
    double e, d;
    bool use_f32;

    my_main() : rank(NULL), rank32(NULL), max_iter(100), e(0.001), d(0.85), use_f32(false) {}

    virtual bool prepare() {
        if (use_f32) rank32 = new float[G.num_nodes()];
        else rank = new double[G.num_nodes()];
        return true;
    }

    virtual bool run() {
        if (use_f32) pagerank(G, (float) e, (float) d, max_iter, rank32);
        else pagerank(G, e, d, max_iter, rank);
        return true;
    }

    virtual bool post_process() {
        for (int i = 0; i < 4 && i < G.num_nodes(); i++)
            printf("rank[%d] = %0.9lf\n", i, use_f32 ? (double) rank32[i] : rank[i]);
        delete[] rank;
        delete[] rank32;
        return true;
    }

    virtual void print_arg_info() { printf("[max_iteration=100] [eplision=0.001] [delta=0.85] [f32]"); }

    virtual bool check_args(int argc, char** argv) {
        if (argc > 0 && (max_iter = atoi(argv[0])) <= 0) return false;
        if (argc > 1 && (e = atof(argv[1])) <= 0) return false;
        if (argc > 2 && (d = atof(argv[2])) <= 0) return false;
        if (argc > 3) use_f32 = strcmp(argv[3], "f32") == 0;
        return true;
    }
};

int main(int argc, char** argv) {
    my_main M;
    M.main(argc, argv);
}
