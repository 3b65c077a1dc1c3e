// shl.h -- stand-in for the un-vendored Shoal header the reference's drivers include
// (/root/reference/apps/output_cpp/src/common_main.h:11,95-99: `nthreads = shl__init(nthreads, 0|1)`).
// Shoal only places arrays in NUMA memory; on the MI355X build array placement is the device
// mirror's job, so initialisation is the identity.
#ifndef SHL_H_STUB_
#define SHL_H_STUB_
static inline int shl__init(int num_threads, int /*is_static_schedule*/) { return num_threads; }
#endif
