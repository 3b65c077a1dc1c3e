// avg_teen_cnt_main.cc -- "average number of teenage followers" benchmark driver; protocol and output of
// /root/reference/apps/output_cpp/src/avg_teen_cnt_main.cc (every age 10 :16-18, K = 5 unless given :46-50,
// prints `avg = %0.9lf` :34).
#include "common_main.h"
#include "avg_teen_cnt.h"

class my_main : public main_t
{
  public:
    int32_t* age;
    int32_t* teen_cnt;
    int K;
    float avg;

    my_main() : age(NULL), teen_cnt(NULL), K(5), avg(0) {}
    virtual ~my_main() {
        delete[] age;
        delete[] teen_cnt;
    }

    virtual bool prepare() {
        age = new int32_t[G.num_nodes()];
        for (node_t i = 0; i < G.num_nodes(); i++) age[i] = 10;
        teen_cnt = new int32_t[G.num_nodes()];
        return true;
    }

    virtual bool run() {
        avg = avg_teen_cnt(G, age, teen_cnt, K);
        return true;
    }

    virtual bool post_process() {
        printf("avg = %0.9lf\n", avg);
        return true;
    }

    virtual void print_arg_info() { printf("[K=5]"); }

    virtual bool check_args(int argc, char** argv) {
        if (argc > 0) {
            K = atoi(argv[0]);
            if (K <= 0) return false;
        }
        return true;
    }
};

int main(int argc, char** argv) {
    my_main M;
    M.main(argc, argv);
}
