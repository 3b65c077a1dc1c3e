// pagerank.h -- entry point with the signature gm_comp emits for apps/src/pagerank.gm
// (/root/reference/src/backend_cpp/gm_cpp_gen.cc:520-608; call site apps/output_cpp/src/pagerank_main.cc:28).
#ifndef GM_GENERATED_CPP_PAGERANK_H
#define GM_GENERATED_CPP_PAGERANK_H

#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <float.h>
#include <limits.h>
#include <cmath>
#include <algorithm>
#include <omp.h>
#include "gm.h"

void pagerank(gm_graph& G, double e,
    double d, int32_t max,
    double*G_pg_rank);

// What gm_comp would emit had pagerank.gm declared `e,d: Float; pg_rank: Node_Prop<Float>`
// (BASELINE config 2: fp32 ranks).  Same rules: Float -> float by value, property -> float*.
void pagerank(gm_graph& G, float e,
    float d, int32_t max,
    float*G_pg_rank);

#endif
