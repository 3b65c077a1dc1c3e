// sssp_main.cc -- single-source shortest paths benchmark driver; protocol and output of
// /root/reference/apps/output_cpp/src/sssp_main.cc (root 0 :20, edge lengths (rand % 100) + 1 from
// gm_rand32 in slot order :33-34, prints dist[0..9] :50-52).  Addition: an optional app argument overrides the root.
#include "common_main.h"
#include "sssp.h"
#include "gm_rand.h"

class my_main : public main_t
{
  public:
    int32_t* len;   // length of each edge
    int32_t* dist;  // distance of each node
    node_t root;

    my_main() : len(NULL), dist(NULL), root(0) {}
    virtual ~my_main() {
        delete[] len;
        delete[] dist;
    }

    virtual bool prepare() {
        gm_rand32 xorshift_rng;
        dist = new int32_t[G.num_nodes()];
        len = new int32_t[G.num_edges()];
        for (edge_t i = 0; i < G.num_edges(); i++) len[i] = (xorshift_rng.rand() % 100) + 1;   // 1 .. 100
        return true;
    }

    virtual bool run() {
        sssp(G, dist, len, root);
        return true;
    }

    virtual bool post_process() {
        for (int i = 0; i < 10 && i < G.num_nodes(); i++) printf("dist[%d] = %d\n", i, dist[i]);
        return true;
    }

    virtual void print_arg_info() { printf("[root=0]"); }

    virtual bool check_args(int argc, char** argv) {
        if (argc > 0) root = (node_t) atol(argv[0]);
        return true;
    }
};

int main(int argc, char** argv) {
    my_main M;
    M.main(argc, argv);
}
