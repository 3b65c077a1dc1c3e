// triangle_counting_cn.h -- entry point gm_comp would emit for
//   Procedure triangle_counting_cn(G: Graph): Long {
//     Long T = 0;
//     Foreach(v: G.Nodes) Foreach(u: v.Nbrs)(u > v) Foreach(w: v.CommonNbrs(u))(w > u) { T += 1; }
//     Return T; }
// (CommonNbrs: /root/reference/src/backend_cpp/gm_cpp_opt_common_nbr.cc:11-26, emitted through
// gm_common_neighbor_iter; not one of the reference's apps -- SURVEY.md section 8f rank 4 asks for the iterator
// and one app that uses it).
#ifndef GM_GENERATED_CPP_TRIANGLE_COUNTING_CN_H
#define GM_GENERATED_CPP_TRIANGLE_COUNTING_CN_H

#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <float.h>
#include <limits.h>
#include <cmath>
#include <algorithm>
#include <omp.h>
#include "gm.h"

int64_t triangle_counting_cn(gm_graph& G);

#endif
