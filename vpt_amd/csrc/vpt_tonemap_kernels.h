// vpt_tonemap_kernels.h — the tone-mapping kernels over vpt_tonemap.h's per-texel functions; included by vpt_post.hip only.
#pragma once
#include "vpt_tonemap.h"

__global__ void k_tonemap_fuse_args(uint8_t *table, TmFuse f) { *(TmFuse *)(table + VPT_TM_FUSE_OFFSET) = f; }
template <int KIND>
__global__ void __launch_bounds__(256) k_tonemap_table(uint8_t *table, TonemapParams p) {
    uint32_t h = blockIdx.x * 256u + threadIdx.x;             // 256 workgroups
    uint32_t packed = tonemap_texel<KIND>(make_uint2(h | (h << 16), h | (h << 16)), p);
    table[h] = (uint8_t)(packed & 0xffu);
    if (h == 0) table[65536] = (uint8_t)(packed >> 24);       // Range never reads it: its alpha goes through table[h]
}
// Artistic with uSaturation == 1 (the default): mix(dot * gray, rgb, 1) = fma(rgb, 1, g * 0) = rgb whenever g is finite
// (g * 0 is a signed zero; adding it changes at most the sign of a zero, which pow treats alike), so the channel byte
// is again a function of one half: table[h] = unorm8(pow((h - low) / range, e)).  A non-finite g (a channel is inf /
// NaN or the dot product overflows) makes every channel NaN -> 0, which the per-pixel test below reproduces.
__global__ void __launch_bounds__(256) k_tonemap_table_artistic(uint8_t *table, TonemapParams p) {
    uint32_t h = blockIdx.x * 256u + threadIdx.x;
    float c = __half2float(__ushort_as_half((unsigned short)h));
    float range = p.high - p.low;
    float e = (-vpt_logf((p.mid - p.low) / range) / vpt_logf(2.0f)) / p.gamma;
    table[h] = (uint8_t)to_unorm8(vpt_powf((c - p.low) / range, e));
    if (h == 0) table[65536] = 255;
}
__global__ void __launch_bounds__(256) k_tonemap_apply_table_artistic(const uint2 *src, uint32_t *dst, size_t n, const uint8_t *table, TonemapParams p) {
    float range = p.high - p.low;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint2 t = src[i];
        float4 c = half4_to_float4(t);
        f3 v = { (c.x - p.low) / range, (c.y - p.low) / range, (c.z - p.low) / range };
        const float gray = 0.57735026918962576f;
        float z = (dot3(v, f3{ gray, gray, gray }) * gray) * (1.0f - p.saturation);     // g * 0: a signed zero, or NaN
        uint32_t rgb = (uint32_t)table[t.x & 0xffffu] | ((uint32_t)table[t.x >> 16] << 8) | ((uint32_t)table[t.y & 0xffffu] << 16);
        dst[i] = ((z == 0.0f) ? rgb : 0u) | 0xff000000u;
    }
}
template <bool ALPHA_FROM_TABLE>
__global__ void __launch_bounds__(256) k_tonemap_apply_table(const uint2 *src, uint32_t *dst, size_t n, const uint8_t *table) {
    uint32_t alpha = (uint32_t)table[65536] << 24;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint2 t = src[i];
        uint32_t r = table[t.x & 0xffffu], g = table[t.x >> 16], b = table[t.y & 0xffffu];
        uint32_t a = ALPHA_FROM_TABLE ? ((uint32_t)table[t.y >> 16] << 24) : alpha;
        dst[i] = r | (g << 8) | (b << 16) | a;
    }
}
// the same with the byte table copied into LDS first (64 KiB + 4 B per workgroup, two workgroups per CU): pays from about
// 6 Mpixel on, where the global-memory form is bound by the texture path's byte gathers rather than by the image stream
template <bool ALPHA_FROM_TABLE>
__global__ void __launch_bounds__(1024) k_tonemap_apply_table_lds(const uint2 *src, uint32_t *dst, size_t n, const uint8_t *table) {
    extern __shared__ uint32_t tm_lds[];
    const uint4 *t4 = (const uint4 *)table;                     // the table allocation is padded to a multiple of 16 bytes
    for (int i = threadIdx.x; i < (VPT_TM_TABLE_ENTRIES + 15) / 16; i += 1024) ((uint4 *)tm_lds)[i] = t4[i];
    __syncthreads();
    const uint8_t *lt = (const uint8_t *)tm_lds;
    uint32_t alpha = (uint32_t)lt[65536] << 24;
    const size_t stride = (size_t)gridDim.x * 1024;
    for (size_t i0 = (size_t)blockIdx.x * 1024 + threadIdx.x; i0 < n; i0 += 4 * stride) {
        uint2 t[4];                                              // four texels in flight per thread
#pragma unroll
        for (int u = 0; u < 4; u++) { size_t i = i0 + (size_t)u * stride; t[u] = i < n ? src[i] : make_uint2(0u, 0u); }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            size_t i = i0 + (size_t)u * stride;
            uint32_t r = lt[t[u].x & 0xffffu], g = lt[t[u].x >> 16], b = lt[t[u].y & 0xffffu];
            uint32_t a = ALPHA_FROM_TABLE ? ((uint32_t)lt[t[u].y >> 16] << 24) : alpha;
            if (i < n) dst[i] = r | (g << 8) | (b << 16) | a;
        }
    }
}
// Artistic at saturation 1 with the table in LDS (k_tonemap_apply_table_artistic's arithmetic)
__global__ void __launch_bounds__(1024) k_tonemap_apply_table_artistic_lds(const uint2 *src, uint32_t *dst, size_t n, const uint8_t *table, TonemapParams p) {
    extern __shared__ uint32_t tm_lds[];
    const uint4 *t4 = (const uint4 *)table;
    for (int i = threadIdx.x; i < (VPT_TM_TABLE_ENTRIES + 15) / 16; i += 1024) ((uint4 *)tm_lds)[i] = t4[i];
    __syncthreads();
    const uint8_t *lt = (const uint8_t *)tm_lds;
    const float range = p.high - p.low;
    const size_t stride = (size_t)gridDim.x * 1024;
    for (size_t i0 = (size_t)blockIdx.x * 1024 + threadIdx.x; i0 < n; i0 += 4 * stride) {
        uint2 t[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { size_t i = i0 + (size_t)u * stride; t[u] = i < n ? src[i] : make_uint2(0u, 0u); }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            size_t i = i0 + (size_t)u * stride;
            float4 c = half4_to_float4(t[u]);
            f3 v = { (c.x - p.low) / range, (c.y - p.low) / range, (c.z - p.low) / range };
            const float gray = 0.57735026918962576f;
            float z = (dot3(v, f3{ gray, gray, gray }) * gray) * (1.0f - p.saturation);     // g * 0: a signed zero, or NaN
            uint32_t rgb = (uint32_t)lt[t[u].x & 0xffffu] | ((uint32_t)lt[t[u].x >> 16] << 8) | ((uint32_t)lt[t[u].y & 0xffffu] << 16);
            if (i < n) dst[i] = ((z == 0.0f) ? rgb : 0u) | 0xff000000u;
        }
    }
}

// n texels, grid-stride; src RGBA16F, dst RGBA8 (both row-major, same pixel order)
template <int KIND>
__global__ void __launch_bounds__(256) k_tonemap(const uint2 *src, uint32_t *dst, size_t n, TonemapParams p) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = tonemap_texel<KIND>(src[i], p);
}
