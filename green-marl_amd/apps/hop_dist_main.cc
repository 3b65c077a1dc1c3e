// hop_dist_main.cc -- BFS (hop distance) benchmark driver; protocol and output of
// /root/reference/apps/output_cpp/src/hop_dist_main.cc (root fixed to 0 :21, prints dist[0..9] :36-38).
// Addition: an optional app argument overrides the root.
#include "common_main.h"
#include "hop_dist.h"

class my_main : public main_t
{
  public:
    int32_t* dist;
    node_t root;

    my_main() : dist(NULL), root(0) {}
    virtual ~my_main() { delete[] dist; }

    virtual bool prepare() {
        dist = new int32_t[G.num_nodes()];
        return true;
    }

    virtual bool run() {
        hop_dist(G, dist, root);
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
