// common_main.h -- benchmark driver skeleton with the reference's main_t protocol
// (/root/reference/apps/output_cpp/src/common_main.h:29-251): main() parses
// `<graph_name> <num_threads> <nfspath> [app args]`, loads the graph, times run() only, prints
// `graph loading time=`, `reverse edge creation time=`, `running time=` (ms) and the terminator
// line the reference's result checker greps (scripts/extract_result.py:372-395).
// Addition: <graph_name> of the form RMAT:<scale>[:<permute>[:<edge_factor>]] builds the reference's
// RMAT graph (seed 1997, a,b,c = .57,.19,.19) on the device instead of reading a .bin file.
#ifndef COMMON_MAIN_H
#define COMMON_MAIN_H

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <omp.h>

#include "gm.h"
#include "graph_gen.h"
#include "shl.h"

class main_t
{
  protected:
    gm_graph G;
    gm_graph& get_graph() { return G; }
    double time_to_exclude;
    void add_time_to_exlude(double ms) { time_to_exclude += ms; }

    static double now_ms() {
        struct timeval t;
        gettimeofday(&t, NULL);
        return t.tv_sec * 1000.0 + t.tv_usec * 0.001;
    }

  public:
    main_t() : time_to_exclude(0) {}
    virtual ~main_t() {}

    virtual void main(int argc, char** argv) {
        gm_graph_check_node_edge_size_at_link_time();
        if (argc < 4) {
            printf("%s <graph_name> <num_threads> <nfspath>", argv[0]);
            print_arg_info();
            printf("\n");
            exit(EXIT_FAILURE);
        }
        if (!check_args(argc - 4, &argv[4])) {
            printf("error procesing argument\n");
            printf("%s <graph_name> <num_threads> ", argv[0]);
            print_arg_info();
            printf("\n");
            exit(EXIT_FAILURE);
        }
        int nthreads = shl__init(atoi(argv[2]), 0);
        printf("running with %d threads\n", nthreads);
        gm_rt_set_num_threads(nthreads);

        double t0 = now_ms();
        if (strncmp(argv[1], "RMAT:", 5) == 0) {
            int scale = 0, permute = 0, ef = 16;
            sscanf(argv[1] + 5, "%d:%d:%d", &scale, &permute, &ef);
            if (scale < 1 || scale > 26 || ef < 1 || ((long long) ef << scale) >= (1LL << 31)) {
                printf("bad RMAT spec %s\n", argv[1]);
                exit(EXIT_FAILURE);
            }
            printf("generating graph... %s\n", argv[1]);
            if (!create_RMAT_graph_in(G, (node_t) 1 << scale, (edge_t) ef << scale, 1997, 0.57, 0.19, 0.19, permute != 0)) {
                printf("error generating graph\n");
                exit(EXIT_FAILURE);
            }
            printf("N = %ld, M = %ld\n", (long) G.num_nodes(), (long) G.num_edges());
            printf("graph loading time=%lf\n", now_ms() - t0);
        } else {
            printf("loading graph... %s\n", argv[1]);
            if (!G.load_binary(argv[1])) {
                printf("error reading graph\n");
                exit(EXIT_FAILURE);
            }
            printf("graph loading time=%lf\n", now_ms() - t0);
        }
        t0 = now_ms();
        G.make_reverse_edges();
        printf("reverse edge creation time=%lf\n", now_ms() - t0);

        if (!prepare()) {
            printf("Error prepare data\n");
            exit(EXIT_FAILURE);
        }
        t0 = now_ms();
        bool b = run();
        printf("running time=%lf\n", now_ms() - t0 - time_to_exclude);
        if (!b) {
            printf("Error runing algortihm\n");
            exit(EXIT_FAILURE);
        }
        if (!post_process()) {
            printf("Error post processing\n");
            exit(EXIT_FAILURE);
        }
        b = cleanup();
        printf("XXXXXXXXXX GM DONE XXXXXXXXXXXXXX\n");
        if (!b) exit(EXIT_FAILURE);
    }

    virtual bool check_answer() { return true; }
    virtual bool run() = 0;
    virtual bool prepare() { return true; }
    virtual bool post_process() { return true; }
    virtual bool cleanup() { return true; }
    virtual bool check_args(int argc, char** argv) { return true; }
    virtual void print_arg_info() {}
};

// This tree's own drivers are written against gm_app: the same protocol, with the three phases given as
// callables instead of a subclass per benchmark.
#include <functional>
#include <string>
#include <vector>

class gm_app : public main_t
{
  public:
    typedef std::function<bool(gm_graph&)> phase_fn;
    typedef std::function<bool(const std::vector<std::string>&)> args_fn;

    gm_app& usage(const char* text) { usage_ = text; return *this; }
    gm_app& args(args_fn f) { args_ = f; return *this; }
    gm_app& setup(phase_fn f) { setup_ = f; return *this; }
    gm_app& kernel(phase_fn f) { kernel_ = f; return *this; }
    gm_app& report(phase_fn f) { report_ = f; return *this; }
    int exec(int argc, char** argv) {
        main(argc, argv);
        return 0;
    }

    virtual bool run() { return kernel_ ? kernel_(G) : false; }
    virtual bool prepare() { return setup_ ? setup_(G) : true; }
    virtual bool post_process() { return report_ ? report_(G) : true; }
    virtual void print_arg_info() { printf("%s", usage_.c_str()); }
    virtual bool check_args(int argc, char** argv) {
        std::vector<std::string> v(argv, argv + argc);
        return args_ ? args_(v) : true;
    }

  private:
    std::string usage_;
    args_fn args_;
    phase_fn setup_, kernel_, report_;
};

#endif
