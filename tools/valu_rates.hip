// valu_rates.hip — measures per-SIMD issue cost (cycles per wave64 instruction) of the VALU ops the renderer
// kernels lean on.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITERS 4096
#define UNROLL 16

#define KERNEL(name, decl, body)                                                                 \
__global__ void __launch_bounds__(256) name(float *out, uint64_t *cyc) {                              \
    decl;                                                                                        \
    uint64_t t0 = __builtin_readcyclecounter();                                                  \
    for (int i = 0; i < ITERS; i++) {                                                            \
        _Pragma("unroll") for (int u = 0; u < UNROLL; u++) { body; }                             \
    }                                                                                            \
    uint64_t t1 = __builtin_readcyclecounter();                                                  \
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink;                                           \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                             \
}

// 8 independent chains so dependent-issue latency does not bound a single wave
#define DECL_F float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; float sink = 0
#define FIN_F sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7

KERNEL(k_fma, DECL_F,
    asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %4, %4, %4, %5\n v_fma_f32 %6, %6, %6, %7"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); FIN_F)
KERNEL(k_mul_lo, DECL_F,
    asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %4, %4, %5\n v_mul_lo_u32 %6, %6, %7"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); FIN_F)
KERNEL(k_mul_u24, DECL_F,
    asm volatile("v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %2, %2, %3\n v_mul_u32_u24 %4, %4, %5\n v_mul_u32_u24 %6, %6, %7"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); FIN_F)
KERNEL(k_rcp, DECL_F,
    asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %2, %2\n v_rcp_f32 %4, %4\n v_rcp_f32 %6, %6"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); FIN_F)
KERNEL(k_cndmask, DECL_F,
    asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %2, %2, %3, vcc"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc"); FIN_F)
KERNEL(k_min, DECL_F,
    asm volatile("v_min_f32 %0, %0, %1\n v_min_f32 %2, %2, %3\n v_max_f32 %4, %4, %5\n v_max_f32 %6, %6, %7"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); FIN_F)
KERNEL(k_cvt_ubyte, DECL_F,
    asm volatile("v_cvt_f32_ubyte0 %0, %1\n v_cvt_f32_ubyte1 %2, %3\n v_cvt_f32_ubyte2 %4, %5\n v_cvt_f32_ubyte3 %6, %7"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); FIN_F)
KERNEL(k_lshl_or, DECL_F,
    asm volatile("v_lshl_or_b32 %0, %0, 3, %1\n v_lshl_or_b32 %2, %2, 3, %3\n v_and_b32 %4, %4, %5\n v_xor_b32 %6, %6, %7"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); FIN_F)

#define q0 qq0

KERNEL(k_add_u32, DECL_F,
    asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3\n v_add_u32 %4, %4, %5\n v_add_u32 %6, %6, %7"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); FIN_F)
KERNEL(k_lshr_xor, DECL_F,
    asm volatile("v_lshrrev_b32 %0, %1, %0\n v_xor_b32 %2, %2, %3\n v_lshrrev_b32 %4, 22, %4\n v_xor_b32 %6, %6, %7"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); FIN_F)
KERNEL(k_mad_u64, DECL_F; uint64_t qq0 = threadIdx.x; uint64_t q1 = qq0 + 3,
    asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %3, %2, %0\n v_mad_u64_u32 %1, vcc, %3, %2, %1"
                 : "+v"(q0), "+v"(q1) : "v"(a2), "v"(a3) : "vcc"); a0 += (float)(uint32_t)q0; a1 += (float)(uint32_t)q1; FIN_F)
KERNEL(k_cvt_u32, DECL_F,
    asm volatile("v_cvt_f32_u32 %0, %1\n v_cvt_u32_f32 %2, %3\n v_cvt_f32_i32 %4, %5\n v_cvt_f32_u32 %6, %7"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); FIN_F)
KERNEL(k_fract_med3, DECL_F,
    asm volatile("v_fract_f32 %0, %1\n v_med3_f32 %2, %2, %3, %1\n v_fract_f32 %4, %5\n v_med3_f32 %6, %6, %7, %5"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); FIN_F)
KERNEL(k_log_rsq, DECL_F,
    asm volatile("v_log_f32 %0, %0\n v_rsq_f32 %2, %2\n v_log_f32 %4, %4\n v_rsq_f32 %6, %6"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); FIN_F)
KERNEL(k_add3_lshladd, DECL_F,
    asm volatile("v_add3_u32 %0, %0, %1, %2\n v_lshl_add_u32 %2, %2, 2, %3\n v_add3_u32 %4, %4, %5, %6\n v_lshl_add_u32 %6, %6, 2, %7"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); FIN_F)
KERNEL(k_sdwa_sub, DECL_F,
    asm volatile("v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n v_cvt_f32_ubyte0_sdwa %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_sub_u32_sdwa %4, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n v_cvt_f32_ubyte0_sdwa %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); FIN_F)
// ONE dependent chain per wave (what a renderer wave looks like): issue rate then depends on how many waves share the SIMD
KERNEL(k_fma_dep, DECL_F,
    asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %0, %0, %0, %1"
                 : "+v"(a0), "+v"(a1)); FIN_F)
KERNEL(k_int_dep, DECL_F,
    asm volatile("v_lshrrev_b32 %0, 3, %0\n v_xor_b32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %1"
                 : "+v"(a0), "+v"(a1)); FIN_F)
KERNEL(k_mul_lo_dep, DECL_F,
    asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %0, %0, %1"
                 : "+v"(a0), "+v"(a1)); FIN_F)
// the renderer's PCG round as the compiler emits it, one dependent chain (9 VALU instructions per round, 4 rounds per body)
__device__ __forceinline__ uint32_t pcg_round(uint32_t x) { x = x * 747796405u + 2891336453u; x = ((x >> ((x >> 28u) + 4u)) ^ x) * 277803737u; return (x >> 22u) ^ x; }
__global__ void __launch_bounds__(256) k_pcg_dep(float *out, uint64_t *cyc) {
    uint32_t x = threadIdx.x;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) { x = pcg_round(x); x = pcg_round(x); x = pcg_round(x); x = pcg_round(x); }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)x;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

typedef float float2_ __attribute__((ext_vector_type(2)));
#define DECL_P float2_ p0 = {(float)threadIdx.x, 1.f}, p1 = p0 + 1.f, p2 = p0 + 2.f, p3 = p0 + 3.f, p4 = p0 + 4.f, p5 = p0 + 5.f, p6 = p0 + 6.f, p7 = p0 + 7.f; float sink = 0
#define FIN_P { float2_ s = p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7; sink = s.x + s.y; }
KERNEL(k_pk_fma, DECL_P,
    asm volatile("v_pk_fma_f32 %0, %0, %0, %1\n v_pk_fma_f32 %2, %2, %2, %3\n v_pk_fma_f32 %4, %4, %4, %5\n v_pk_fma_f32 %6, %6, %6, %7"
                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)); FIN_P)
KERNEL(k_pk_mul, DECL_P,
    asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %2, %2, %3\n v_pk_add_f32 %4, %4, %5\n v_pk_add_f32 %6, %6, %7"
                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)); FIN_P)

// LDS reads next to VALU: does a ds_read_b32 stream steal VALU issue slots?
__global__ void __launch_bounds__(256) k_fma_lds(float *out, uint64_t *cyc) {
    __shared__ float tab[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) tab[i] = i;
    __syncthreads();
    DECL_F;
    uint32_t idx = threadIdx.x * 7;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            float v = tab[(idx + u * 37) & 4095];
            asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %4, %4, %4, %5\n v_fma_f32 %6, %6, %6, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            a1 += v;
        }
        idx += 13;
    }
    uint64_t t1 = __builtin_readcyclecounter();
    FIN_F;
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename K> static void run(const char *name, K k, int insts_per_body, int blocks_per_cu) {
    int nblocks = 256 * blocks_per_cu;
    float *out; uint64_t *cyc;
    hipMalloc(&out, (size_t)nblocks * 256 * 4); hipMalloc(&cyc, nblocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(nblocks), dim3(256), 0, 0, out, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(nblocks), dim3(256), 0, 0, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts_per_wave = (double)ITERS * UNROLL * insts_per_body;
    double waves_per_simd = blocks_per_cu;          // 4 waves per block, 4 SIMDs per CU
    // wall-clock cycles per instruction per SIMD at an assumed 2.4 GHz (upper bound on clock)
    double cyc_per_inst = ms * 1e-3 * 2.4e9 / (insts_per_wave * waves_per_simd);
    printf("%-12s blocks/CU %d  %.3f ms  -> %.2f cycles/inst/SIMD @2.4GHz (less if clock lower)\n", name, blocks_per_cu, ms, cyc_per_inst);
    hipFree(out); hipFree(cyc);
}

int main() {
    int bs[] = {1, 2, 4, 7, 8};
    for (int bi = 0; bi < 5; bi++) {
        int b = bs[bi];
        run("fma", k_fma, 4, b); run("pk_fma", k_pk_fma, 4, b); run("pk_mul/add", k_pk_mul, 4, b);
        run("mul_lo_u32", k_mul_lo, 4, b); run("mul_u32_u24", k_mul_u24, 4, b); run("rcp", k_rcp, 4, b);
        run("cmp+cndmask", k_cndmask, 4, b); run("min/max", k_min, 4, b); run("cvt_ubyte", k_cvt_ubyte, 4, b);
        run("lshl_or/and", k_lshl_or, 4, b); run("fma+ds_read", k_fma_lds, 4, b);
        run("add_u32", k_add_u32, 4, b); run("lshr/xor", k_lshr_xor, 4, b); run("mad_u64_u32", k_mad_u64, 4, b);
        run("cvt_u32", k_cvt_u32, 4, b); run("fract/med3", k_fract_med3, 4, b); run("log/rsq", k_log_rsq, 4, b);
        run("add3/lshl_add", k_add3_lshladd, 4, b); run("sdwa sub/cvt", k_sdwa_sub, 4, b);
        run("fma DEP", k_fma_dep, 4, b); run("int DEP", k_int_dep, 4, b); run("mul_lo DEP", k_mul_lo_dep, 4, b);
        run("pcg DEP(36)", k_pcg_dep, 36, b);
    }
    return 0;
}
