// gather_bench.hip -- development microbenchmark: rate of random 4-byte gathers on gfx950 by table
// size (LDS / L2 / Infinity Cache / HBM) and load flavour.  Build:
//   hipcc --offload-arch=gfx950 -O3 -o gather_bench gather_bench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float ld_plain(const float* p) { return *p; }
__device__ __forceinline__ float ld_nt(const float* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ float ld_sc1(const float* p) {
    float v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ float ld_sc0sc1(const float* p) {
    float v;
    asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// mode 0 plain, 1 nt, 2 sc1(asm, serialised wait: latency bound, only indicative), 3 sc0 sc1
template <int MODE, int PER>
__global__ void __launch_bounds__(256) gather_kernel(const uint32_t* __restrict__ idx, int64_t n,
                                                     const float* __restrict__ tab, float* __restrict__ out) {
    int64_t base = ((int64_t) blockIdx.x * 256) * PER + threadIdx.x;
    float acc = 0.f;
    uint32_t ix[PER];
#pragma unroll
    for (int u = 0; u < PER; u++) {
        int64_t j = base + (int64_t) u * 256;
        ix[u] = j < n ? __builtin_nontemporal_load(idx + j) : 0u;
    }
#pragma unroll
    for (int u = 0; u < PER; u++) {
        float v;
        if (MODE == 0) v = ld_plain(tab + ix[u]);
        else if (MODE == 1) v = ld_nt(tab + ix[u]);
        else if (MODE == 2) v = ld_sc1(tab + ix[u]);
        else v = ld_sc0sc1(tab + ix[u]);
        acc += v;
    }
    if (acc == 12345.678f) out[0] = acc;  // keep the loads alive
}

// 8-byte elements: same request count, twice the bytes
template <int PER>
__global__ void __launch_bounds__(256) gather8_kernel(const uint32_t* __restrict__ idx, int64_t n,
                                                      const double* __restrict__ tab, float* __restrict__ out) {
    int64_t base = ((int64_t) blockIdx.x * 256) * PER + threadIdx.x;
    double acc = 0;
    uint32_t ix[PER];
#pragma unroll
    for (int u = 0; u < PER; u++) {
        int64_t j = base + (int64_t) u * 256;
        ix[u] = j < n ? __builtin_nontemporal_load(idx + j) : 0u;
    }
#pragma unroll
    for (int u = 0; u < PER; u++) acc += tab[ix[u]];
    if (acc == 12345.678) out[0] = (float) acc;
}

// table in LDS (entries <= 28672), persistent blocks of 1024 threads
template <int PER>
__global__ void __launch_bounds__(1024) gather_lds_kernel(const uint32_t* __restrict__ idx, int64_t n, int entries,
                                                          const float* __restrict__ tab, float* __restrict__ out) {
    __shared__ float s[28672];
    for (int i = threadIdx.x; i < entries; i += 1024) s[i] = tab[i];
    __syncthreads();
    float acc = 0.f;
    for (int64_t blk = blockIdx.x; blk * 1024 * PER < n; blk += gridDim.x) {
        int64_t base = blk * 1024 * PER + threadIdx.x;
#pragma unroll
        for (int u = 0; u < PER; u++) {
            int64_t j = base + (int64_t) u * 1024;
            uint32_t ix = j < n ? __builtin_nontemporal_load(idx + j) : 0u;
            acc += s[ix];
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

// stream only (index read, no gather): the streaming ceiling of this access shape
template <int PER>
__global__ void __launch_bounds__(256) stream_kernel(const uint32_t* __restrict__ idx, int64_t n, float* __restrict__ out) {
    int64_t base = ((int64_t) blockIdx.x * 256) * PER + threadIdx.x;
    uint32_t acc = 0;
#pragma unroll
    for (int u = 0; u < PER; u++) {
        int64_t j = base + (int64_t) u * 256;
        acc += j < n ? __builtin_nontemporal_load(idx + j) : 0u;
    }
    if (acc == 0x12345678u) out[0] = (float) acc;
}

__global__ void fill_idx_kernel(uint32_t* idx, int64_t n, uint64_t entries, uint64_t salt) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t) gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint64_t z = (uint64_t) i + salt + 0x9E3779B97F4A7C15ULL;   // splitmix64
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z = z ^ (z >> 31);
        idx[i] = (uint32_t) (z % entries);
    }
}

template <typename F>
static double time_ms(F f, int reps = 5) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int r = 0; r < reps; r++) f();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char** argv) {
    const int64_t n = 1LL << 28;  // 256M gathers per launch
    uint32_t* d_idx;
    float* d_out;
    CK(hipMalloc(&d_idx, n * 4));
    CK(hipMalloc(&d_out, 64));
    const size_t max_entries = 1ULL << 28;  // 1 GiB of floats
    float* d_tab;
    CK(hipMalloc(&d_tab, max_entries * 8));  // also used as the double table (up to 2^27 doubles... sized x8 to be safe)
    CK(hipMemset(d_tab, 0, max_entries * 8));
    constexpr int PER = 8;
    const int blocks = (int) ((n + 256 * PER - 1) / (256 * PER));
    double ms = time_ms([&] { hipLaunchKernelGGL(stream_kernel<PER>, dim3(blocks), dim3(256), 0, 0, d_idx, n, d_out); });
    printf("index stream only: %.3f ms  %.1f GB/s\n", ms, n * 4 / ms / 1e6);
    size_t sizes[] = {1u << 20, 1u << 28};
    for (size_t entries : sizes) {
        hipLaunchKernelGGL(fill_idx_kernel, dim3(4096), dim3(256), 0, 0, d_idx, n, (uint64_t) entries, (uint64_t) entries * 7919);
        CK(hipDeviceSynchronize());
        printf("table %9zu entries (%8.2f MiB fp32):", entries, entries * 4 / 1048576.0);
        ms = time_ms([&] { hipLaunchKernelGGL((gather_kernel<0, PER>), dim3(blocks), dim3(256), 0, 0, d_idx, n, d_tab, d_out); });
        printf("  plain %7.1f G/s", n / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL((gather_kernel<1, PER>), dim3(blocks), dim3(256), 0, 0, d_idx, n, d_tab, d_out); });
        printf("  nt %7.1f G/s", n / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL((gather8_kernel<PER>), dim3(blocks), dim3(256), 0, 0, d_idx, n, (const double*) d_tab, d_out); });
        printf("  f64 %7.1f G/s", n / ms / 1e6);
        if (entries <= 28672) {
            ms = time_ms([&] { hipLaunchKernelGGL((gather_lds_kernel<PER>), dim3(256), dim3(1024), 0, 0, d_idx, n, (int) entries, d_tab, d_out); });
            printf("  LDS %7.1f G/s", n / ms / 1e6);
        }
        if (argc > 1) {
            ms = time_ms([&] { hipLaunchKernelGGL((gather_kernel<3, PER>), dim3(blocks), dim3(256), 0, 0, d_idx, n, d_tab, d_out); }, 1);
            printf("  sc0sc1(serial) %7.1f G/s", n / ms / 1e6);
        }
        printf("\n");
        fflush(stdout);
    }
    // memory-type experiment: does an uncached / fine-grained table change the cost of a miss?
    {
        const unsigned flags[] = {hipDeviceMallocUncached, hipDeviceMallocFinegrained};
        const char* names[] = {"uncached", "finegrained"};
        for (int m = 0; m < 2; m++) {
            for (size_t entries : {(size_t) 1 << 20, (size_t) 1 << 24, (size_t) 1 << 28}) {
                float* t2 = nullptr;
                hipError_t e = hipExtMallocWithFlags((void**) &t2, entries * 4, flags[m]);
                if (e != hipSuccess) { printf("%s alloc failed: %s\n", names[m], hipGetErrorString(e)); (void) hipGetLastError(); continue; }
                CK(hipMemset(t2, 0, entries * 4));
                hipLaunchKernelGGL(fill_idx_kernel, dim3(4096), dim3(256), 0, 0, d_idx, n, (uint64_t) entries, (uint64_t) entries * 7919);
                CK(hipDeviceSynchronize());
                ms = time_ms([&] { hipLaunchKernelGGL((gather_kernel<0, PER>), dim3(blocks), dim3(256), 0, 0, d_idx, n, t2, d_out); });
                printf("%s table %9zu entries: plain %7.1f G/s", names[m], entries, n / ms / 1e6);
                ms = time_ms([&] { hipLaunchKernelGGL((gather_kernel<1, PER>), dim3(blocks), dim3(256), 0, 0, d_idx, n, t2, d_out); });
                printf("  nt %7.1f G/s\n", n / ms / 1e6);
                fflush(stdout);
                CK(hipFree(t2));
            }
        }
    }
    return 0;
}
