// avg_teen_cnt.h -- entry point with the signature gm_comp emits for apps/src/avg_teen_cnt.gm
// (call site /root/reference/apps/output_cpp/src/avg_teen_cnt_main.cc:24; N_P<Int> -> int32_t*, Int -> int32_t, Float return).
#ifndef GM_GENERATED_CPP_AVG_TEEN_CNT_H
#define GM_GENERATED_CPP_AVG_TEEN_CNT_H

#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <float.h>
#include <limits.h>
#include <cmath>
#include <algorithm>
#include <omp.h>
#include "gm.h"

float avg_teen_cnt(gm_graph& G, int32_t* G_age,
    int32_t* G_teen_cnt, int32_t K);

#endif
