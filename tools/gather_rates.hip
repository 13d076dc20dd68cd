// gather_rates.hip — cycles per wave64 vector-memory instruction per CU for the access shapes of the volume sampler
// (L2/L1-resident data, many waves): coalesced vs per-lane gather, aligned vs unaligned, 1/2/4/8/16-byte loads.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>

#define ITERS 2048

template <int BYTES, int SHIFT>
__global__ void __launch_bounds__(256) k_gather(const uint8_t *base, const uint32_t *offs, uint32_t mask, uint32_t *out) {
    uint32_t o = offs[blockIdx.x * 256 + threadIdx.x];
    uint32_t acc = 0;
    for (int i = 0; i < ITERS; i++) {
        const uint8_t *p = base + (o & mask) + SHIFT;
        if (BYTES == 1) { acc += *p; }
        else if (BYTES == 2) { uint16_t v; __builtin_memcpy(&v, p, 2); acc += v; }
        else if (BYTES == 4) { uint32_t v; __builtin_memcpy(&v, p, 4); acc += v; }
        else if (BYTES == 8) { uint64_t v; __builtin_memcpy(&v, p, 8); acc += (uint32_t)v + (uint32_t)(v >> 32); }
        else if (BYTES == 12) { struct { uint32_t a, b, c; } v; __builtin_memcpy(&v, __builtin_assume_aligned(p, 4), 12); acc += v.a + v.b + v.c; }
        else { uint4 v; __builtin_memcpy(&v, p, 16); acc += v.x + v.y + v.z + v.w; }
        o = o * 1664525u + 1013904223u + (acc & 1u);      // next pseudo-random offset (depends on the load: serial per lane)
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <typename K> static void run(const char *name, K k, const uint8_t *d, const uint32_t *doffs, uint32_t mask, uint32_t *dout, int blocks_per_cu) {
    int nb = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d, doffs, mask, dout);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d, doffs, mask, dout);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_insts_per_cu = (double)ITERS * 4 * blocks_per_cu;
    printf("%-34s %d blk/CU: %.3f ms -> %.1f cycles per wave-load per CU @2.4GHz\n", name, blocks_per_cu, ms, ms * 1e-3 * 2.4e9 / wave_insts_per_cu);
}

int main() {
    size_t bytes = 64u << 20;                     // 64 MiB window: L2 + Infinity-Cache resident
    uint8_t *d; hipMalloc(&d, bytes + 64); hipMemset(d, 1, bytes + 64);
    int nthreads = 256 * 8 * 256;
    uint32_t *h = (uint32_t *)malloc(nthreads * 4), *doffs, *dout;
    hipMalloc(&doffs, nthreads * 4); hipMalloc(&dout, nthreads * 4);
    for (int mode = 1; mode < 3; mode++) {
        // mode 0: random over 64 MiB (64 distinct lines per wave); 1: random within a 4 KiB window per wave (L1-resident, many lanes
        // share lines); 2: 8x8-tile-like: lanes of a wave within 1 KiB
        uint32_t x = 12345;
        for (int t = 0; t < nthreads; t++) { x = x * 1664525u + 1013904223u; h[t] = x; }
        hipMemcpy(doffs, h, nthreads * 4, hipMemcpyHostToDevice);
        uint32_t mask = mode == 0 ? (uint32_t)(bytes - 1) & ~15u : (mode == 1 ? 4095u & ~15u : 1023u & ~15u);
        const char *mn = mode == 0 ? "random 64MiB" : (mode == 1 ? "within 4KiB" : "within 1KiB");
        char nm[64];
        for (int b = 8; b <= 8; b *= 2) {
            snprintf(nm, 64, "%s 4B aligned", mn); run(nm, k_gather<4, 0>, d, doffs, mask, dout, b);
            snprintf(nm, 64, "%s 8B aligned", mn); run(nm, k_gather<8, 0>, d, doffs, mask, dout, b);
            snprintf(nm, 64, "%s 8B +1", mn); run(nm, k_gather<8, 1>, d, doffs, mask, dout, b);
            snprintf(nm, 64, "%s 8B +4 (dword aligned)", mn); run(nm, k_gather<8, 4>, d, doffs, mask, dout, b);
            snprintf(nm, 64, "%s 12B +0", mn); run(nm, k_gather<12, 0>, d, doffs, mask, dout, b);
            snprintf(nm, 64, "%s 12B +4 (dword aligned)", mn); run(nm, k_gather<12, 4>, d, doffs, mask, dout, b);
            snprintf(nm, 64, "%s 12B +8 (crosses 16B)", mn); run(nm, k_gather<12, 8>, d, doffs, mask, dout, b);
            snprintf(nm, 64, "%s 16B aligned", mn); run(nm, k_gather<16, 0>, d, doffs, mask, dout, b);
            snprintf(nm, 64, "%s 16B +4", mn); run(nm, k_gather<16, 4>, d, doffs, mask, dout, b);
        }
    }
    return 0;
}
