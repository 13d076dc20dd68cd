// line_gather.hip — throughput of the volume sampler's access shape: every lane of a wave reads a few bytes of its OWN
// 128-byte brick slot (64 distinct lines per wave instruction), for tables that live in L2 (2 MiB), in the Infinity Cache
// (128 MiB) or in HBM (2 GiB).  Answers: what does the second tap window cost, does dword alignment matter beyond L1,
// does a miss move a whole 128-byte line or a 64-byte half, do nt loads (L1 bypass) help.
// Build: hipcc --offload-arch=gfx950 -O3 -o line_gather line_gather.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITERS 512
typedef uint64_t u64_any __attribute__((aligned(1)));
struct W3 { uint32_t a, b, c; };

template <int MODE>
__global__ void __launch_bounds__(256) k_lines(const uint8_t *base, uint32_t slot_mask, uint32_t *out) {
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            x = x * 1664525u + 1013904223u;
            uint32_t slot = (x >> 8) & slot_mask;
            uint32_t o = ((x >> 3) & 3u) + 5u * ((x >> 5) & 3u) + 25u * (x & 3u);       // a brick-cell offset, 0..93
            const uint8_t *p = base + (size_t)slot * 128u;
            if (MODE == 0) { acc += (uint32_t)*(const u64_any *)(p + o); }
            else if (MODE == 1) { acc += (uint32_t)*(const u64_any *)(p + o) + (uint32_t)(*(const u64_any *)(p + o + 25) >> 32); }
            else if (MODE == 2) {
                W3 q0 = *(const W3 *)(p + (o & ~3u)), q1 = *(const W3 *)(p + ((o + 25) & ~3u));
                acc += q0.a + q0.c + q1.b + q1.c;
            }
            else if (MODE == 3) { acc += (uint32_t)*(const uint64_t *)(p + 8) + (uint32_t)*(const uint64_t *)(p + 32); }
            else if (MODE == 4) { acc += (uint32_t)*(const uint64_t *)(p + 8) + (uint32_t)*(const uint64_t *)(p + 96); }
            else if (MODE == 5) { uint4 v = *(const uint4 *)(p + 16 * (o & 7u)); acc += v.x + v.w; }
            else if (MODE == 6) {
                acc += (uint32_t)__builtin_nontemporal_load((const u64_any *)(p + o)) + (uint32_t)(__builtin_nontemporal_load((const u64_any *)(p + o + 25)) >> 32);
            }
            else if (MODE == 7) {                       // 64-byte slots (twice as many of them): windows +o, +o+16, o in 0..42
                const uint8_t *p64 = base + (size_t)(((x >> 8) & (slot_mask * 2u + 1u))) * 64u;
                uint32_t o2 = ((x >> 3) & 3u) + 4u * ((x >> 5) & 3u) + 16u * (x % 3u);
                if (o2 > 40u) o2 = 40u;
                acc += (uint32_t)*(const u64_any *)(p64 + o2) + (uint32_t)(*(const u64_any *)(p64 + o2 + 16) >> 32);
            }
            else if (MODE == 8) { acc += (uint32_t)*(const uint64_t *)(p + 8 * (o & 15u)); }           // one aligned 8-byte load
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <typename K> static void run(const char *name, K k, const uint8_t *d, uint32_t slot_mask, uint32_t *dout, const char *where) {
    int nb = 256 * 7;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d, slot_mask, dout);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d, slot_mask, dout);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double samples = (double)nb * 256 * ITERS * 4;
    printf("%-8s %-44s %.3f ms  %.1f Gsamples/s  (%.2f TB/s if a 128-B line moves per sample)\n", where, name, ms, samples / ms * 1e-6, samples * 128 / ms * 1e-9);
    fflush(stdout);
}

int main() {
    size_t bytes = 2ull << 30;
    uint8_t *d; if (hipMalloc(&d, bytes + 256) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(d, 1, bytes + 256);
    uint32_t *dout; (void)hipMalloc(&dout, 256 * 7 * 256 * 4);
    struct { const char *w; size_t sz; } tabs[] = { { "L2 2MiB", 2u << 20 }, { "MALL128M", 128u << 20 }, { "HBM 2GiB", 2ull << 30 } };
    for (auto &tb : tabs) {
        uint32_t mask = (uint32_t)(tb.sz / 128 - 1);
        run("1 window  8B unaligned", k_lines<0>, d, mask, dout, tb.w);
        run("1 window  8B aligned", k_lines<8>, d, mask, dout, tb.w);
        run("2 windows 8B unaligned (+o, +o+25)  [now]", k_lines<1>, d, mask, dout, tb.w);
        run("2 windows 12B dword-aligned", k_lines<2>, d, mask, dout, tb.w);
        run("2 x 8B aligned, same 64-B half", k_lines<3>, d, mask, dout, tb.w);
        run("2 x 8B aligned, different halves", k_lines<4>, d, mask, dout, tb.w);
        run("1 x 16B aligned", k_lines<5>, d, mask, dout, tb.w);
        run("2 windows 8B unaligned, nt loads", k_lines<6>, d, mask, dout, tb.w);
        run("64-B slots: 2 windows 8B unaligned", k_lines<7>, d, mask, dout, tb.w);
    }
    return 0;
}
