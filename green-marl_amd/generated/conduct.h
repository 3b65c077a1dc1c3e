// conduct.h -- entry point with the signature gm_comp emits for apps/src/conduct.gm
// (call site /root/reference/apps/output_cpp/src/conduct_main.cc:45; N_P<Int> -> int32_t*, Int -> int32_t, Float return).
#ifndef GM_GENERATED_CPP_CONDUCT_H
#define GM_GENERATED_CPP_CONDUCT_H

#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <float.h>
#include <limits.h>
#include <cmath>
#include <algorithm>
#include <omp.h>
#include "gm.h"

float conduct(gm_graph& G, int32_t* G_member,
    int32_t num);

#endif
