// valu_bench.hip -- issue cost of single instructions on gfx950 (development microbenchmark).
// Every kernel runs ITER x 8 independent copies of one instruction per wave, 16 waves per CU (4 per SIMD), all CUs:
// cycles per wave-instruction and SIMD = elapsed x clock x SIMDs / wave-instructions.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_bench valu_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 4096
#define OP8(body) body(0) body(1) body(2) body(3) body(4) body(5) body(6) body(7)
#define KERNEL(name, decl, body, sink)                                                     \
    __global__ void __launch_bounds__(1024) name(float* out, int n) {                      \
        decl;                                                                              \
        for (int i = 0; i < n; i++) { OP8(body) }                                          \
        sink;                                                                              \
    }
// registers: f[8] floats, d[8] doubles, u[8] unsigned, l[8] 64-bit
#define DECL_F float f[8]; for (int k = 0; k < 8; k++) f[k] = threadIdx.x * 1.0f + k
#define DECL_D double d[8]; for (int k = 0; k < 8; k++) d[k] = threadIdx.x * 1.0 + k
#define DECL_U unsigned u[8]; for (int k = 0; k < 8; k++) u[k] = threadIdx.x + k
#define DECL_L unsigned long long l[8]; for (int k = 0; k < 8; k++) l[k] = threadIdx.x + k
#define SINK_F { float t = 0; for (int k = 0; k < 8; k++) t += f[k]; if (t == 12345.678f) out[0] = t; }
#define SINK_D { double t = 0; for (int k = 0; k < 8; k++) t += d[k]; if (t == 12345.678) out[0] = (float) t; }
#define SINK_U { unsigned t = 0; for (int k = 0; k < 8; k++) t += u[k]; if (t == 0x12345678u) out[0] = t; }
#define SINK_L { unsigned long long t = 0; for (int k = 0; k < 8; k++) t += l[k]; if (t == 0x123456789ull) out[0] = 1; }
#define B_ADD_F32(k) asm volatile("v_add_f32 %0, %0, %0" : "+v"(f[k]));
#define B_AND_B32(k) asm volatile("v_and_b32 %0, 0x7fffffff, %0" : "+v"(u[k]));
#define B_BFE(k) asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(u[k]));
#define B_BFI(k) asm volatile("v_bfi_b32 %0, %0, %0, %0" : "+v"(u[k]));
#define B_ADD_F64(k) asm volatile("v_add_f64 %0, %0, %0" : "+v"(d[k]));
#define B_FMA_F64(k) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[k]));
#define B_CVT_F64_F32(k) asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(d[k]) : "v"(f0));
#define B_CVT_F32_F64(k) asm volatile("v_cvt_f32_f64 %0, %1" : "+v"(f[k]) : "v"(d0));
#define B_LSHL_B64(k) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(l[k]));
#define B_LSHL_ADD_U64(k) asm volatile("v_lshl_add_u64 %0, %0, 2, %0" : "+v"(l[k]));
#define B_DPP(k) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[k]));
#define B_ADD_DPP(k) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[k]));
#define B_CNDMASK(k) asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(u[k]));
#define B_CMP(k) asm volatile("v_cmp_lt_u32 vcc, %0, %0" : : "v"(u[k]) : "vcc");
#define B_LDEXP_F64(k) asm volatile("v_ldexp_f64 %0, %0, 3" : "+v"(d[k]));
#define B_TRUNC_F64(k) asm volatile("v_trunc_f64 %0, %0" : "+v"(d[k]));
#define B_CVT_U32_F64(k) asm volatile("v_cvt_u32_f64 %0, %1" : "+v"(u[k]) : "v"(d0));
#define B_SALU(k) asm volatile("s_add_u32 %0, %0, 1" : "+s"(su[k]));
#define B_READLANE(k) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(su[k]) : "v"(u[k]));
KERNEL(k_add_f32, DECL_F, B_ADD_F32, SINK_F)
KERNEL(k_and_b32, DECL_U, B_AND_B32, SINK_U)
KERNEL(k_bfe_u32, DECL_U, B_BFE, SINK_U)
KERNEL(k_bfi_b32, DECL_U, B_BFI, SINK_U)
KERNEL(k_add_f64, DECL_D, B_ADD_F64, SINK_D)
KERNEL(k_fma_f64, DECL_D, B_FMA_F64, SINK_D)
KERNEL(k_cvt_f64_f32, DECL_D; float f0 = threadIdx.x, B_CVT_F64_F32, SINK_D)
KERNEL(k_cvt_f32_f64, DECL_F; double d0 = threadIdx.x, B_CVT_F32_F64, SINK_F)
KERNEL(k_lshlrev_b64, DECL_L, B_LSHL_B64, SINK_L)
KERNEL(k_lshl_add_u64, DECL_L, B_LSHL_ADD_U64, SINK_L)
KERNEL(k_mov_dpp, DECL_U, B_DPP, SINK_U)
KERNEL(k_add_u32_dpp, DECL_U, B_ADD_DPP, SINK_U)
KERNEL(k_cndmask, DECL_U, B_CNDMASK, SINK_U)
KERNEL(k_cmp, DECL_U, B_CMP, SINK_U)
KERNEL(k_ldexp_f64, DECL_D, B_LDEXP_F64, SINK_D)
KERNEL(k_trunc_f64, DECL_D, B_TRUNC_F64, SINK_D)
KERNEL(k_cvt_u32_f64, DECL_U; double d0 = threadIdx.x, B_CVT_U32_F64, SINK_U)
#define DECL_SU unsigned su[8]; for (int k = 0; k < 8; k++) su[k] = k; DECL_U
#define SINK_SU { for (int k = 0; k < 8; k++) u[0] += su[k]; } SINK_U
KERNEL(k_readlane, DECL_SU, B_READLANE, SINK_SU)
#define DECL_AB DECL_U; unsigned a = threadIdx.x * 3, b = threadIdx.x * 5; unsigned long long mask = 0x5555aaaa3333ccccull ^ (unsigned long long) n
#define B_CND_INDEP(k) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(u[k]) : "v"(a), "v"(b));
#define B_CND_E64(k) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(u[k]) : "v"(a), "v"(b), "s"(mask));
#define B_CND_E64_DEP(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[k]) : "v"(b), "s"(mask));
#define B_CMP_CND(k) asm volatile("v_cmp_lt_u32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(u[k]) : "v"(a), "v"(b) : "vcc");
#define B_CMP_E64_CND(k) asm volatile("v_cmp_lt_u32_e64 s[20:21], %1, %0\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]" : "+v"(u[k]) : "v"(a), "v"(b) : "s20", "s21");
#define B_BITOP3(k) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xca" : "+v"(u[k]) : "v"(a), "v"(b));
#define B_LSHL_ADD_U32(k) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(u[k]) : "v"(a));
#define B_ADD3(k) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(u[k]) : "v"(a), "v"(b));
#define B_ADD_U32(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k]) : "v"(a));
#define B_OR3(k) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(u[k]) : "v"(a), "v"(b));
#define B_AND_OR(k) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[k]) : "v"(a), "v"(b));
KERNEL(k_cnd_indep, DECL_AB, B_CND_INDEP, SINK_U)
KERNEL(k_cnd_e64, DECL_AB, B_CND_E64, SINK_U)
KERNEL(k_cnd_e64_dep, DECL_AB, B_CND_E64_DEP, SINK_U)
KERNEL(k_cmp_cnd, DECL_AB, B_CMP_CND, SINK_U)
KERNEL(k_cmp_e64_cnd, DECL_AB, B_CMP_E64_CND, SINK_U)
KERNEL(k_bitop3, DECL_AB, B_BITOP3, SINK_U)
KERNEL(k_lshl_add_u32, DECL_AB, B_LSHL_ADD_U32, SINK_U)
KERNEL(k_add3_u32, DECL_AB, B_ADD3, SINK_U)
KERNEL(k_add_u32, DECL_AB, B_ADD_U32, SINK_U)
KERNEL(k_or3, DECL_AB, B_OR3, SINK_U)
KERNEL(k_and_or, DECL_AB, B_AND_OR, SINK_U)
// LDS: random 4-byte gathers from a 120 KiB table, consecutive 4-byte reads, u64 atomics
__global__ void __launch_bounds__(1024) k_lds_gather(float* out, int n) {
    __shared__ float tab[30720];
    for (int i = threadIdx.x; i < 30720; i += 1024) tab[i] = i;
    __syncthreads();
    unsigned x = threadIdx.x * 2654435761u;
    float t = 0;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) { x = x * 1664525u + 1013904223u; t += tab[(x >> 8) % 30720u]; }
    }
    if (t == 12345.678f) out[0] = t;
}
__global__ void __launch_bounds__(1024) k_lds_linear(float* out, int n) {
    __shared__ float tab[30720];
    for (int i = threadIdx.x; i < 30720; i += 1024) tab[i] = i;
    __syncthreads();
    float t = 0;
    int at = threadIdx.x;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) { t += tab[at]; at = (at + 1024) % 30720; }
    }
    if (t == 12345.678f) out[0] = t;
}
template <typename K>
static void run(const char* name, K kern, float* out, double clock_ghz, int cus) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    kern<<<cus, 1024>>>(out, 16);
    hipEventRecord(a);
    kern<<<cus, 1024>>>(out, ITER);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double winst = (double) cus * 16 * ITER * 8;   // wave-instructions
    printf("%-16s %8.3f ms  %6.2f cycles per wave-instruction and SIMD\n", name, ms, ms * 1e-3 * clock_ghz * 1e9 * cus * 4 / winst);
    fflush(stdout);
}
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate * 1e-6;
    printf("%s: %d CUs, %.2f GHz\n", p.gcnArchName, p.multiProcessorCount, ghz);
    fflush(stdout);
    float* out;
    hipMalloc(&out, 64);
    const int cus = p.multiProcessorCount;
#define RUN(k) run(#k, k, out, ghz, cus)
    RUN(k_add_f32); RUN(k_and_b32); RUN(k_bfe_u32); RUN(k_bfi_b32); RUN(k_add_f64); RUN(k_fma_f64); RUN(k_cvt_f64_f32); RUN(k_cvt_f32_f64);
    RUN(k_lshlrev_b64); RUN(k_lshl_add_u64); RUN(k_mov_dpp); RUN(k_add_u32_dpp); RUN(k_cndmask); RUN(k_cmp); RUN(k_ldexp_f64); RUN(k_trunc_f64);
    RUN(k_cvt_u32_f64); RUN(k_readlane);
    RUN(k_cnd_indep); RUN(k_cnd_e64); RUN(k_cnd_e64_dep); RUN(k_cmp_cnd); RUN(k_cmp_e64_cnd); RUN(k_bitop3); RUN(k_lshl_add_u32); RUN(k_add3_u32); RUN(k_add_u32); RUN(k_or3); RUN(k_and_or); RUN(k_lds_gather); RUN(k_lds_linear);
    return 0;
}
