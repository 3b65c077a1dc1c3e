// gm_rand.h -- xorshift generators with the reference's class names
// (/root/reference/apps/output_cpp/gm_graph/inc/gm_rand.h:12-68; hop_dist_main.cc:19 instantiates
// gm_rand32).  Sequences follow the reference's update rules (gm_rand.cc:10-24) so that code seeded
// the same way draws the same numbers.
#ifndef GM_RAND_H_
#define GM_RAND_H_
#include <stdint.h>

class gm_rand64
{
  public:
    gm_rand64() : state(0x0139408DCBBF7A44LL) {}
    gm_rand64(int64_t seed) : state(seed) {}
    int64_t rand();
  private:
    int64_t state;
};

class gm_rand32
{
  public:
    gm_rand32() : state((int32_t) 2463534242u) {}
    gm_rand32(int32_t seed) : state(seed) {}
    int32_t rand();
  private:
    int32_t state;
};

class gm_rand
{
  public:
    gm_rand() {}
    gm_rand(long seed) : rng((int32_t) seed) {}
    int32_t rand() { return rng.rand(); }
  private:
    gm_rand32 rng;
};

#endif
