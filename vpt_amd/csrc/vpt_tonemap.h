// vpt_tonemap.h — the reference's ten tone mappers as one templated per-pixel kernel (gfx950).
//
// the step right after the renderer path.  Source = the renderer's RGBA16F render buffer
// (row-major local rows), target = RGBA8, one thread per pixel: 8 B in + 4 B out, HBM streaming.
// Reference fragment shaders: src/glsl/tonemappers/{Artistic,Range,Reinhard,Reinhard2,Uncharted2,Filmic,Unreal,Aces,
// Lottes,Uchimura}ToneMapper.glsl (main() at :37, :33, :40, :42, :70, :44, :40, :50, :66, :94).
// Arithmetic contract: fp32, operations in the shader's order, IEEE division (the build sets
// -fhip-fp32-correctly-rounded-divide-sqrt), pow(x, y) = exp(y * log(x)) on the contract's polynomial log / exp.
#pragma once
#include "vpt_device.h"

// e^x — same routine as the oracle's vpo_expf: reduction by ln 2, degree-6 polynomial, exact scaling
VPT_DEV float vpt_expf(float x) {
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500E-4f;
    p = fmaf(p, r, 1.3981999507E-3f);
    p = fmaf(p, r, 8.3334519073E-3f);
    p = fmaf(p, r, 4.1665795894E-2f);
    p = fmaf(p, r, 1.6666665459E-1f);
    p = fmaf(p, r, 5.0000001201E-1f);
    float y = fmaf(p, r * r, r) + 1.0f;
    float v = ldexpf(y, (int)n);                              // v_ldexp_f32: exact, denormal results rounded once
    v = (x > 89.0f) ? __builtin_inff() : v;
    v = (x < -104.0f) ? 0.0f : v;
    return (x != x) ? x : v;
}
VPT_DEV float vpt_powf(float x, float y) { return vpt_expf(y * vpt_logf(x)); }

struct TonemapParams {          // = struct vpt_tonemap_params (include/vpt.h)
    float low, mid, high, saturation;
    float min, max;
    float exposure;
    float gamma;
};

#define VPT_TM_ARTISTIC   0
#define VPT_TM_RANGE      1
#define VPT_TM_REINHARD   2
#define VPT_TM_REINHARD2  3
#define VPT_TM_UNCHARTED2 4
#define VPT_TM_FILMIC     5
#define VPT_TM_UNREAL     6
#define VPT_TM_ACES       7
#define VPT_TM_LOTTES     8
#define VPT_TM_UCHIMURA   9

VPT_DEV float tm_uncharted2_curve(float x) {                  // Uncharted2ToneMapper.glsl:32-41
    const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
    return ((x * (A * x + C * B) + D * E) / (x * (A * x + B) + D * F)) - E / F;
}
// the scalar curve of the eight exposure/gamma mappers
template <int KIND>
VPT_DEV float tm_curve(float x) {
    if (KIND == VPT_TM_REINHARD) return x / (1.0f + x);                                         // ReinhardToneMapper.glsl:32-34
    if (KIND == VPT_TM_REINHARD2) return (x * (1.0f + x / (4.0f * 4.0f))) / (1.0f + x);         // Reinhard2ToneMapper.glsl:32-35
    if (KIND == VPT_TM_UNCHARTED2) {                                                            // Uncharted2ToneMapper.glsl:43-49
        float white_scale = 1.0f / tm_uncharted2_curve(11.2f);
        return tm_uncharted2_curve(2.0f * x) * white_scale;
    }
    if (KIND == VPT_TM_FILMIC) {                                                                // FilmicToneMapper.glsl:32-36
        float X = vmax(0.0f, x - 0.004f);
        return vpt_powf((X * (6.2f * X + 0.5f)) / (X * (6.2f * X + 1.7f) + 0.06f), 2.2f);
    }
    if (KIND == VPT_TM_UNREAL) return x / (x + 0.155f) * 1.019f;                                // UnrealToneMapper.glsl:32-34
    if (KIND == VPT_TM_ACES) return vclamp01((x * (2.51f * x + 0.03f)) / (x * (2.43f * x + 0.59f) + 0.14f));   // AcesToneMapper.glsl:32-39
    if (KIND == VPT_TM_LOTTES) {                                                                // LottesToneMapper.glsl:32-47
        const float a = 1.6f, d = 0.977f;
        const float b = (float)1.0730397117173704, c = (float)0.16741993817791725;             // the shader's const b, c in float64, rounded once
        return vpt_powf(x, a) / (vpt_powf(x, a * d) * b + c);
    }
    // Uchimura: UchimuraToneMapper.glsl:32-61 with P = 1, a = 1, m = 0.22, l = 0.4, c = 1.33, b = 0
    const float P = 1.0f, a = 1.0f, m = 0.22f, l = 0.4f, c = 1.33f, b = 0.0f;
    float l0 = ((P - m) * l) / a;
    float S0 = m + l0, S1 = m + a * l0;
    float CP = -((a * P) / (P - S1)) / P;
    float t = vclamp01((x - 0.0f) / (m - 0.0f));
    float w0 = 1.0f - t * t * (3.0f - 2.0f * t);
    float w2 = (x < m + l0) ? 0.0f : 1.0f;
    float w1 = 1.0f - w0 - w2;
    float T = m * vpt_powf(x / m, c) + b;
    float S = P - (P - S1) * vpt_expf(CP * (x - S0));
    float L = m + a * (x - m);
    return T * w0 + L * w1 + S * w2;
}

VPT_DEV float4 half4_to_float4(uint2 h) {
    return make_float4(__half2float(__ushort_as_half((unsigned short)(h.x & 0xffffu))), __half2float(__ushort_as_half((unsigned short)(h.x >> 16))),
                       __half2float(__ushort_as_half((unsigned short)(h.y & 0xffffu))), __half2float(__ushort_as_half((unsigned short)(h.y >> 16))));
}
VPT_DEV uint32_t pack_unorm8x4(float r, float g, float b, float a) {
    return to_unorm8(r) | (to_unorm8(g) << 8) | (to_unorm8(b) << 16) | (to_unorm8(a) << 24);
}

template <int KIND>
VPT_DEV uint32_t tonemap_texel(uint2 texel, const TonemapParams &p) {
    float4 c = half4_to_float4(texel);
    if (KIND == VPT_TM_ARTISTIC) {                                                              // ArtisticToneMapper.glsl:37-46
        float range = p.high - p.low;
        f3 v = { (c.x - p.low) / range, (c.y - p.low) / range, (c.z - p.low) / range };
        const float gray = 0.57735026918962576f;
        float g = dot3(v, f3{ gray, gray, gray }) * gray;
        float e = (-vpt_logf((p.mid - p.low) / range) / vpt_logf(2.0f)) / p.gamma;
        return pack_unorm8x4(vpt_powf(mixf(g, v.x, p.saturation), e), vpt_powf(mixf(g, v.y, p.saturation), e),
                             vpt_powf(mixf(g, v.z, p.saturation), e), 1.0f);
    }
    if (KIND == VPT_TM_RANGE) {                                                                 // RangeToneMapper.glsl:33-36
        float range = p.max - p.min, e = 1.0f / p.gamma;
        return pack_unorm8x4(vpt_powf((c.x - p.min) / range, e), vpt_powf((c.y - p.min) / range, e),
                             vpt_powf((c.z - p.min) / range, e), vpt_powf((c.w - p.min) / range, e));
    }
    float e = 1.0f / p.gamma;                                  // pow(vec4(curve(src.rgb * uExposure), 1), vec4(1.0 / uGamma))
    return pack_unorm8x4(vpt_powf(tm_curve<KIND>(c.x * p.exposure), e), vpt_powf(tm_curve<KIND>(c.y * p.exposure), e),
                         vpt_powf(tm_curve<KIND>(c.z * p.exposure), e), vpt_powf(1.0f, e));
}

// ---- table form -------------------------------------------------------------------------------------------------
// For Range and the eight curve mappers every output byte is a function of ONE half-precision input (and the pass's
// uniforms): a 65 536-entry byte table, filled by evaluating exactly the code above on every half bit pattern, turns
// the pass into byte gathers — bit-identical by construction, and HBM-bound instead of VALU-bound.
// table[h] = the colour-channel byte for input half h; entry 65 536 = the alpha byte of the curve mappers.
#define VPT_TM_TABLE_ENTRIES 65537
// Behind the table (VPT_TONEMAPPER_OPTION_FUSE): what a renderer's fused frame store needs besides the table itself, so that its kernels
// carry ONE pointer (two SGPRs for the life of the wave) and fetch the rest with scalar loads where they store
#define VPT_TM_FUSE_OFFSET 65600
struct TmFuse { uint32_t *out; int mode; float low, range, one_minus_saturation; };   // mode 1 = the eight curve mappers, 2 = Range, 3 = Artistic (saturation 1)
#define VPT_TM_TABLE_BYTES (VPT_TM_FUSE_OFFSET + sizeof(TmFuse))
