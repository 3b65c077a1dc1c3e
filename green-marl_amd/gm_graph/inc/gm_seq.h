// gm_seq.h -- Node_Sequence / Edge_Sequence as they appear in an emitted SIGNATURE
// (`Seeds: Node_Sequence` -> `gm_node_seq& Seeds`, /root/reference/src/backend_cpp/gm_cpp_gen.cc:520-608; the
// reference's container: apps/output_cpp/gm_graph/inc/gm_seq.h:33-147).  The collection library itself is outside
// the accelerated path; this is the part of its interface that a driver
// (bc_main.cc:21-41: new / push_back / delete) and an entry point (iteration over the items) touch.
#ifndef GM_SEQ_H_
#define GM_SEQ_H_
#include <list>
#include "gm_graph_typedef.h"

template <typename IT, typename T>
class Seq_Iterator
{
  public:
    Seq_Iterator(IT b, IT e) : cur(b), end(e) {}
    bool has_next() { return cur != end; }
    T get_next() { T v = *cur; ++cur; return v; }
  private:
    IT cur, end;
};

template <typename T>
class gm_seq
{
  public:
    gm_seq() {}
    explicit gm_seq(int /*max_threads*/) {}
    void push_back(T e) { Q.push_back(e); }
    void push_front(T e) { Q.push_front(e); }
    T pop_back() { T v = Q.back(); Q.pop_back(); return v; }
    T pop_front() { T v = Q.front(); Q.pop_front(); return v; }
    void clear() { Q.clear(); }
    int get_size() { return (int) Q.size(); }
    std::list<T>& get_list() { return Q; }
    typedef Seq_Iterator<typename std::list<T>::iterator, T> seq_iter;
    typedef Seq_Iterator<typename std::list<T>::reverse_iterator, T> rev_iter;
    typedef seq_iter par_iter;
    seq_iter prepare_seq_iteration() { return seq_iter(Q.begin(), Q.end()); }
    rev_iter prepare_rev_iteration() { return rev_iter(Q.rbegin(), Q.rend()); }
    par_iter prepare_par_iteration(int thread_id, int max_threads) {
        return thread_id == 0 ? par_iter(Q.begin(), Q.end()) : par_iter(Q.end(), Q.end());
    }
  private:
    std::list<T> Q;
};

typedef gm_seq<node_t> gm_node_seq;
typedef gm_seq<edge_t> gm_edge_seq;
#endif
