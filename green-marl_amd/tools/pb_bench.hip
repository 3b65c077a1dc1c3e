// pb_bench.hip -- development microbenchmark for the planned propagation-blocking PageRank sweep
// (DESIGN.md section 7): are its two phases really streaming on gfx950?
//   phase 1 (scatter): walk the edges in SOURCE order (16 per source), read the source's value and the
//                      edge's static slot (4 B, sequential), store the value at that slot.  Slots are laid
//                      out bin-major (NB destination-range bins), inside a bin by the XCD that will write
//                      them, inside that in source order -- so every (bin, XCD) region is filled front to
//                      back by one XCD, and the lines being filled at any time (NB x 8 x ~2) fit its L2.
//   phase 2 (accumulate): one workgroup per bin streams the bin's values and 16-bit local row ids and adds
//                      them into row sums held in LDS (64-bit fixed point: order independent), then writes
//                      the bin's rows out.
// Destinations are uniformly random here (no hub rows: the LDS same-address serialisation of a real hub is a
// separate problem, see DESIGN).  Reports time and bytes per edge of each phase.  Build:
//   hipcc --offload-arch=gfx950 -O3 -o pb_bench pb_bench.hip
#include <hip/hip_runtime.h>
#include <string.h>
#include <rocprim/rocprim.hpp>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define EDGES_PER_BLOCK 2048   // edge block = unit dealt round-robin to workgroups (hence to XCDs)
#define DEG 16

__device__ __forceinline__ uint32_t hash32(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return (uint32_t) x;
}

// key = (bin << 3 | xcd of the edge's block), value = edge id; dst local row = hash
__global__ void make_keys(int64_t E, int nbins, int rows_per_bin, uint32_t* __restrict__ key, uint32_t* __restrict__ val,
                          uint16_t* __restrict__ dst_local_by_edge) {
    int64_t e = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; e < E; e += stride) {
        const uint32_t h = hash32((uint64_t) e * 0x9e3779b97f4a7c15ULL + 7);
        const uint32_t bin = h % (uint32_t) nbins;
        const uint32_t xcd = (uint32_t) ((e / EDGES_PER_BLOCK) & 7);
        key[e] = (bin << 3) | xcd;
        val[e] = (uint32_t) e;
        dst_local_by_edge[e] = (uint16_t) ((h >> 12) % (uint32_t) rows_per_bin);
    }
}

// sorted position i holds edge val[i]: slot_of_edge[val[i]] = i, dst_local_by_slot[i] = dst_local_by_edge[val[i]]
__global__ void invert(int64_t E, const uint32_t* __restrict__ val, const uint16_t* __restrict__ dl_e,
                       uint32_t* __restrict__ slot_of_edge, uint16_t* __restrict__ dl_s) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < E; i += stride) {
        slot_of_edge[val[i]] = (uint32_t) i;
        dl_s[i] = dl_e[val[i]];
    }
}

__global__ void bin_offsets(int64_t E, const uint32_t* __restrict__ key_sorted, int nbins, int64_t* __restrict__ off) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i <= E; i += stride) {
        const int64_t b0 = i == 0 ? -1 : (int64_t) (key_sorted[i - 1] >> 3);
        const int64_t b1 = i == E ? nbins : (int64_t) (key_sorted[i] >> 3);
        for (int64_t b = b0 + 1; b <= b1; b++) off[b] = i;
    }
}

// phase 1: workgroup w takes edge blocks w, w + G, ...
__global__ void __launch_bounds__(256) scatter_kernel(int64_t E, const float* __restrict__ contrib,
                                                      const uint32_t* __restrict__ slot_of_edge, float* __restrict__ msg) {
    const int64_t nblocks = (E + EDGES_PER_BLOCK - 1) / EDGES_PER_BLOCK;
    for (int64_t k = blockIdx.x; k < nblocks; k += gridDim.x) {
        const int64_t e0 = k * EDGES_PER_BLOCK;
#pragma unroll
        for (int u = 0; u < EDGES_PER_BLOCK / 256; u++) {
            const int64_t e = e0 + u * 256 + threadIdx.x;
            if (e < E) {
                const uint32_t s = __builtin_nontemporal_load(slot_of_edge + e);
                const float v = contrib[e / DEG];
                msg[s] = v;
            }
        }
    }
}

// phase 2: one workgroup per bin (bins taken round-robin)
template <bool FIXED>
__global__ void __launch_bounds__(1024) accumulate_kernel(int nbins, int rows_per_bin, const int64_t* __restrict__ off,
                                                          const float* __restrict__ msg, const uint16_t* __restrict__ dl,
                                                          float* __restrict__ out) {
    extern __shared__ unsigned long long sums[];   // rows_per_bin x 8 B
    for (int b = blockIdx.x; b < nbins; b += gridDim.x) {
        for (int r = threadIdx.x; r < rows_per_bin; r += 1024) sums[r] = 0;
        __syncthreads();
        const int64_t lo = off[b], hi = off[b + 1];
        for (int64_t i = lo + threadIdx.x; i < hi; i += 1024) {
            const float v = __builtin_nontemporal_load(msg + i);
            const uint16_t r = __builtin_nontemporal_load(dl + i);
            if (FIXED) atomicAdd(&sums[r], (unsigned long long) ((double) v * 4611686018427387904.0));   // 2^62
            else atomicAdd((float*) &sums[r], v);
        }
        __syncthreads();
        for (int r = threadIdx.x; r < rows_per_bin; r += 1024)
            out[(int64_t) b * rows_per_bin + r] = FIXED ? (float) ((double) sums[r] / 4611686018427387904.0) : *(float*) &sums[r];
        __syncthreads();
    }
}

int main(int argc, char** argv) {
    const int scale = argc > 1 ? atoi(argv[1]) : 29;          // log2(edges)
    const int rows_per_bin = argc > 2 ? atoi(argv[2]) : 16384;
    const int64_t E = 1LL << scale, V = E / DEG;
    const int nbins = (int) ((V + rows_per_bin - 1) / rows_per_bin);
    printf("E=%lld V=%lld bins=%d x %d rows\n", (long long) E, (long long) V, nbins, rows_per_bin);
    uint32_t *key, *key2, *val, *val2, *slot;
    uint16_t *dl_e, *dl_s;
    float *contrib, *msg, *out;
    int64_t* off;
    CK(hipMalloc(&key, E * 4)); CK(hipMalloc(&key2, E * 4)); CK(hipMalloc(&val, E * 4)); CK(hipMalloc(&val2, E * 4));
    CK(hipMalloc(&slot, E * 4)); CK(hipMalloc(&dl_e, E * 2)); CK(hipMalloc(&dl_s, E * 2));
    CK(hipMalloc(&contrib, V * 4)); CK(hipMalloc(&msg, E * 4)); CK(hipMalloc(&out, (int64_t) nbins * rows_per_bin * 4));
    CK(hipMalloc(&off, (nbins + 2) * 8));
    CK(hipMemset(contrib, 0, V * 4));
    hipLaunchKernelGGL(make_keys, dim3(4096), dim3(256), 0, 0, E, nbins, rows_per_bin, key, val, dl_e);
    size_t tb = 0;
    int bits = 3; while ((1 << (bits - 3)) < nbins) bits++;
    CK(rocprim::radix_sort_pairs(nullptr, tb, key, key2, val, val2, (size_t) E, 0u, (unsigned) bits, 0));
    void* tmp; CK(hipMalloc(&tmp, tb));
    CK(rocprim::radix_sort_pairs(tmp, tb, key, key2, val, val2, (size_t) E, 0u, (unsigned) bits, 0));   // stable: source order inside (bin, xcd)
    hipLaunchKernelGGL(invert, dim3(4096), dim3(256), 0, 0, E, (const uint32_t*) val2, (const uint16_t*) dl_e, slot, dl_s);
    hipLaunchKernelGGL(bin_offsets, dim3(4096), dim3(256), 0, 0, E, (const uint32_t*) key2, nbins, off);
    CK(hipDeviceSynchronize());
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int grid : {256, 512, 1024, 2048}) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(a, 0));
            hipLaunchKernelGGL(scatter_kernel, dim3(grid), dim3(256), 0, 0, E, (const float*) contrib, (const uint32_t*) slot, msg);
            CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
        }
        printf("phase 1 scatter  grid %4d: %.3f ms  -> %.1f G edges/s, %.2f TB/s of the 8.25 B/edge it has to move\n",
               grid, best, E / best / 1e6, E * 8.25 / best / 1e9);
    }
    for (int fixed = 0; fixed < 2; fixed++) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(a, 0));
            if (fixed) hipLaunchKernelGGL(accumulate_kernel<true>, dim3(256), dim3(1024), rows_per_bin * 8, 0, nbins, rows_per_bin, (const int64_t*) off, (const float*) msg, (const uint16_t*) dl_s, out);
            else hipLaunchKernelGGL(accumulate_kernel<false>, dim3(256), dim3(1024), rows_per_bin * 8, 0, nbins, rows_per_bin, (const int64_t*) off, (const float*) msg, (const uint16_t*) dl_s, out);
            CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
        }
        printf("phase 2 accumulate (%s LDS adds): %.3f ms -> %.1f G edges/s, %.2f TB/s of the 6 B/edge + 4 B/row it has to move\n",
               fixed ? "64-bit fixed-point" : "fp32", best, E / best / 1e6, (E * 6.0 + V * 4.0) / best / 1e9);
    }
    return 0;
}
