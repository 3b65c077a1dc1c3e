// gm_rand.cc -- xorshift steps (sequence contract: /root/reference/apps/output_cpp/gm_graph/src/gm_rand.cc:10-24;
// note the 32-bit variant ASSIGNS the middle shift, as the reference does).
#include "gm_rand.h"

int64_t gm_rand64::rand() {
    uint64_t x = (uint64_t) state;
    x ^= x << 13;
    x = (uint64_t) ((int64_t) x ^ ((int64_t) x >> 7));
    x ^= x << 17;
    state = (int64_t) x;
    return state;
}

int32_t gm_rand32::rand() {
    int32_t x = state;
    x = (int32_t) ((uint32_t) x ^ ((uint32_t) x << 13));
    x = x >> 17;
    x = (int32_t) ((uint32_t) x ^ ((uint32_t) x << 5));
    state = x;
    return state;
}
