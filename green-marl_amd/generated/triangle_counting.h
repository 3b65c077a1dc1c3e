// triangle_counting.h -- entry point with the signature gm_comp emits for apps/src/triangle_counting.gm
// (call site /root/reference/apps/output_cpp/src/triangle_counting_main.cc:14; Long return -> int64_t).
#ifndef GM_GENERATED_CPP_TRIANGLE_COUNTING_H
#define GM_GENERATED_CPP_TRIANGLE_COUNTING_H

#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <float.h>
#include <limits.h>
#include <cmath>
#include <algorithm>
#include <omp.h>
#include "gm.h"

int64_t triangle_counting(gm_graph& G);

#endif
