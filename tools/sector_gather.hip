// sector_gather.hip — does a cache-policy bit make a cold random gather cheaper than one 128-byte line per lane?
// line_gather.hip found 55-57 G lines/s (= 7 TB/s of 128-byte lines) for one line per lane from a 128 MiB / 2 GiB table whatever the
// access shape.  Here the same gather — every lane one aligned 8-byte load at a random 64-byte sector — goes through raw buffer loads
// with each combination of the gfx950 cache-policy bits (sc0 = 1, nt = 2, sc1 = 16), and a second shape reads both 64-byte halves of
// the lane's line (if a miss moves 64 bytes, the second half costs a second request; if it moves the line, it is free).
// Build: hipcc --offload-arch=gfx950 -O3 -o sector_gather sector_gather.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITERS 256
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));

__device__ inline __amdgpu_buffer_rsrc_t make_rsrc(const void *p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)bytes, 0x00020000);
}

template <int AUX, int BOTH>
__global__ void __launch_bounds__(256) k_sect(const uint8_t *base, uint32_t bytes, uint32_t sector_mask, uint32_t *out) {
    __amdgpu_buffer_rsrc_t rs = make_rsrc(base, bytes);
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            x = x * 1664525u + 1013904223u;
            uint32_t off = ((x >> 7) & sector_mask) * 64u + 8u * (x & 7u);
            v2u v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, AUX);
            acc += v.x;
            if (BOTH) {
                v2u w = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(off ^ 64u), 0, AUX);
                acc += w.y;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <typename K> static void run(const char *name, K k, const uint8_t *d, uint32_t bytes, uint32_t *dout, const char *where) {
    int nb = 256 * 7;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    uint32_t mask = bytes / 64u - 1u;
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d, bytes, mask, dout);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d, bytes, mask, dout);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double samples = (double)nb * 256 * ITERS * 4;
    printf("%-9s %-40s %8.3f ms  %6.1f G samples/s\n", where, name, ms, samples / ms * 1e-6);
    fflush(stdout);
}

#define BOTHSHAPES(aux, label) \
    run(label " one 8-B load", k_sect<aux, 0>, d, (uint32_t)tb.sz, dout, tb.w); \
    run(label " both halves of the line", k_sect<aux, 1>, d, (uint32_t)tb.sz, dout, tb.w);

int main() {
    size_t bytes = 2ull << 30;
    uint8_t *d; if (hipMalloc(&d, bytes + 256) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(d, 1, bytes + 256);
    uint32_t *dout; (void)hipMalloc(&dout, 256 * 7 * 256 * 4);
    struct { const char *w; size_t sz; } tabs[] = { { "L2 2MiB", 2u << 20 }, { "MALL128M", 128u << 20 }, { "HBM 1GiB", 1ull << 30 } };
    for (auto &tb : tabs) {
        BOTHSHAPES(0, "plain       ")
        BOTHSHAPES(1, "sc0         ")
        BOTHSHAPES(2, "nt          ")
        BOTHSHAPES(16, "sc1         ")
        BOTHSHAPES(17, "sc0 sc1     ")
        BOTHSHAPES(3, "sc0 nt      ")
        BOTHSHAPES(18, "nt sc1      ")
        BOTHSHAPES(19, "sc0 nt sc1  ")
    }
    return 0;
}
