// bc.h -- entry point with the signature gm_comp emits for apps/src/bc.gm:4
//   Procedure comp_BC(G: Graph, BC: N_P<Float>, Seeds: Node_Sequence)
// (call site /root/reference/apps/output_cpp/src/bc_main.cc:43; property -> float*, collection -> gm_node_seq&,
// src/backend_cpp/gm_cpp_gen.cc:520-608).
#ifndef GM_GENERATED_CPP_BC_H
#define GM_GENERATED_CPP_BC_H

#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <float.h>
#include <limits.h>
#include <cmath>
#include <algorithm>
#include <omp.h>
#include "gm.h"

void comp_BC(gm_graph& G, float* G_BC,
    gm_node_seq& Seeds);

#endif
