// gm_runtime.h -- the slice of the reference's runtime API the drivers and generated entries call
// (/root/reference/apps/output_cpp/gm_graph/inc/gm_runtime.h:50-60): a thread-count holder.
// On the GPU build the count only sizes the host-side OpenMP loops of gm_graph; kernels run on the device.
#ifndef GM_RUNTIME_H_
#define GM_RUNTIME_H_
#include <omp.h>

void gm_rt_initialize();
bool gm_rt_is_initialized();
int gm_rt_get_num_threads();
void gm_rt_set_num_threads(int n);
int gm_rt_thread_id();
void gm_rt_cleanup();

#endif
