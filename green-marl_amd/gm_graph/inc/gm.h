// gm.h -- umbrella header of the host API (what generated code and drivers include;
// cf. /root/reference/apps/output_cpp/gm_graph/inc/gm.h:14-38).  Only the modules on the
// accelerated path exist here: graph container, runtime shim, RNG helpers, the sequence type of bc's signature.
#ifndef GM_H_
#define GM_H_
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "gm_graph_typedef.h"
#include "gm_graph.h"
#include "gm_runtime.h"
#include "gm_rand.h"
#include "gm_seq.h"
#include "gm_common_neighbor_iter.h"
#endif
