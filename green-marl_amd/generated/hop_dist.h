// hop_dist.h -- entry point with the signature gm_comp emits for apps/src/hop_dist.gm
// (call site /root/reference/apps/output_cpp/src/hop_dist_main.cc:28; Node in-arg -> node_t&).
#ifndef GM_GENERATED_CPP_HOP_DIST_H
#define GM_GENERATED_CPP_HOP_DIST_H

#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <float.h>
#include <limits.h>
#include <cmath>
#include <algorithm>
#include <omp.h>
#include "gm.h"

void hop_dist(gm_graph& G, int32_t* G_dist,
    node_t& root);

#endif
