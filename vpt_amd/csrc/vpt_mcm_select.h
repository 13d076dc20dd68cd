// vpt_mcm_select.h — what the MCM translation units (vpt_mcm.hip: tile classes, buckets; vpt_mcm_hit.hip: the integrate kernels;
// vpt_mcm_seq.hip: frame sequences in one launch) share on the host side: the sampler variant of a renderer's kernels and the switch from
// that run-time value to a template argument.
#pragma once
#include "vpt_internal.h"

typedef void (*PassKernel)(PassArgs);
// the sampler variant of the tile-class kernels (LINEAR one-channel byte volumes): VPT_V_WIDE | VPT_V_FAST | VPT_V_REC
static inline int class_variant(const vpt_renderer *r, const PassArgs &a) {
    return (variant_of(r) & VPT_V_WIDE) | (r->fast_math ? VPT_V_FAST : 0) | (a.vol.records ? VPT_V_REC : 0);
}
#define VARIANT_CASES(...) switch (v) { \
        case 0: { constexpr int V = 0; return __VA_ARGS__; } \
        case VPT_V_WIDE: { constexpr int V = VPT_V_WIDE; return __VA_ARGS__; } \
        case VPT_V_FAST: { constexpr int V = VPT_V_FAST; return __VA_ARGS__; } \
        case VPT_V_FAST | VPT_V_WIDE: { constexpr int V = VPT_V_FAST | VPT_V_WIDE; return __VA_ARGS__; } \
        case VPT_V_REC: { constexpr int V = VPT_V_REC; return __VA_ARGS__; } \
        case VPT_V_REC | VPT_V_WIDE: { constexpr int V = VPT_V_REC | VPT_V_WIDE; return __VA_ARGS__; } \
        case VPT_V_REC | VPT_V_FAST: { constexpr int V = VPT_V_REC | VPT_V_FAST; return __VA_ARGS__; } \
        default: { constexpr int V = VPT_V_REC | VPT_V_FAST | VPT_V_WIDE; return __VA_ARGS__; } }
// NEAREST / two-channel / float volumes: the HIT tiles through the general kernel of the volume's variant (from a tile list), the MISS
// tiles through the one-phase sampler of k_mcm_miss (miss_sample_any)
#define FORMAT_CASES(...) switch (v & (VPT_V_NEAREST | VPT_V_RG | VPT_V_F32)) { \
        case VPT_V_NEAREST: { constexpr int F = VPT_V_NEAREST; return __VA_ARGS__; } \
        case VPT_V_RG: { constexpr int F = VPT_V_RG; return __VA_ARGS__; } \
        case VPT_V_RG | VPT_V_NEAREST: { constexpr int F = VPT_V_RG | VPT_V_NEAREST; return __VA_ARGS__; } \
        case VPT_V_F32: { constexpr int F = VPT_V_F32; return __VA_ARGS__; } \
        case VPT_V_F32 | VPT_V_NEAREST: { constexpr int F = VPT_V_F32 | VPT_V_NEAREST; return __VA_ARGS__; } \
        case VPT_V_F32 | VPT_V_RG: { constexpr int F = VPT_V_F32 | VPT_V_RG; return __VA_ARGS__; } \
        default: { constexpr int F = VPT_V_F32 | VPT_V_RG | VPT_V_NEAREST; return __VA_ARGS__; } }
// vpt_mcm_hit.hip: k_mcm_integrate / k_mcm_integrate_early by variant (fuse: + _renderFrame)
PassKernel mcm_hit_kernel(bool fuse, int v, bool early);                    // v: class_variant()
PassKernel mcm_format_hit_kernel(bool fuse, int v, bool wide, bool fast);   // v: variant_of(), another volume format
int mcm_general_pass(vpt_renderer *r, const PassArgs &a, bool fuse);        // the whole image through the general kernel
