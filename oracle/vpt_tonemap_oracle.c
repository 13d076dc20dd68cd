/* vpt_tonemap_oracle.c — CPU restatement of the reference's ten tone mappers.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/oracle.py).  Follows, fragment shader by fragment shader:
 *   src/glsl/tonemappers/ArtisticToneMapper.glsl:37-46     RangeToneMapper.glsl:33-36
 *   ReinhardToneMapper.glsl:32-43   Reinhard2ToneMapper.glsl:32-45   Uncharted2ToneMapper.glsl:32-73
 *   FilmicToneMapper.glsl:32-47     UnrealToneMapper.glsl:32-43      AcesToneMapper.glsl:32-53
 *   LottesToneMapper.glsl:32-69     UchimuraToneMapper.glsl:32-97
 * and the fixed-function part around them: the source is the renderer's RGBA16F colour attachment sampled at texel
 * centres (AbstractToneMapper.js:34-37, RenderingContext.js:184-187 give both the same resolution, so the LINEAR/NEAREST
 * filter returns the texel itself), the target is RGBA8 (AbstractToneMapper.js:66-79): unorm8 = RNE(clamp(x,0,1)*255).
 *
 * Parity: UNPINNED by the reference (it has no tests and its GLSL cannot run here).  Pinned by closed forms in
 * tests/test_tonemap_oracle.py (float64 evaluation of the same formulas, +-1 LSB) and by exp/pow accuracy tests.
 *
 * Arithmetic contract (tone-map addendum): fp32 RNE, no contraction, operations in the order the
 * GLSL writes them, IEEE division; pow(x, y) = exp_c(y * log_c(x)) (GLSL ES 3.00 section 8.2 defines pow through
 * exp2/log2 and leaves x < 0 undefined: here it yields NaN -> 0 on the unorm8 write); exp(x) = exp_c(x);
 * compile-time constant expressions that call pow (Lottes b, c) are evaluated in float64 and rounded once.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define VPO_API __attribute__((visibility("default")))

float vpo_logf(float x);                                   /* vpt_oracle.c */

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* e^x: Cody-Waite reduction by ln 2 (rint), degree-6 Cephes polynomial, exact scaling.  NaN -> NaN, > 89 -> inf, < -104 -> 0 */
VPO_API float vpo_expf(float x) {
    if (x != x) return x;
    if (x > 89.0f) return INFINITY;
    if (x < -104.0f) return 0.0f;
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500E-4f;
    p = fmaf(p, r, 1.3981999507E-3f);
    p = fmaf(p, r, 8.3334519073E-3f);
    p = fmaf(p, r, 4.1665795894E-2f);
    p = fmaf(p, r, 1.6666665459E-1f);
    p = fmaf(p, r, 5.0000001201E-1f);
    float r2 = r * r;
    float y = fmaf(p, r2, r) + 1.0f;
    return ldexpf(y, (int)n);
}
VPO_API float vpo_powf(float x, float y) { return vpo_expf(y * vpo_logf(x)); }

/* IEEE half -> float (exact) */
VPO_API float vpo_f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    if (e == 0x1fu) return u2f(sign | 0x7f800000u | (m << 13));
    if (e == 0) {
        float v = (float)m * 5.9604644775390625e-8f;      /* m * 2^-24, exact */
        return sign ? -v : v;
    }
    return u2f(sign | ((e + 112u) << 23) | (m << 13));
}

/* IEEE minNum / maxNum with -0 < +0, as the renderer oracle's vmin / vmax */
static inline float vmin(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return (f2u(a) & 0x80000000u) ? a : b;
    return a < b ? a : b;
}
static inline float vmax(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return (f2u(a) & 0x80000000u) ? b : a;
    return a > b ? a : b;
}
static inline float clamp01(float x) { return vmin(vmax(x, 0.0f), 1.0f); }
static inline uint8_t to_unorm8(float f) {
    float c = clamp01(f);
    if (c != c) c = 0.0f;
    return (uint8_t)rintf(c * 255.0f);
}

typedef struct {
    float low, mid, high, saturation;      /* Artistic (ArtisticToneMapper.js:14-47) */
    float min, max;                        /* Range (RangeToneMapper.js:14-34) */
    float exposure;                        /* the eight curve mappers (ReinhardToneMapper.js:14-29) */
    float gamma;                           /* all */
} vpo_tonemap_params;

enum { TM_ARTISTIC = 0, TM_RANGE, TM_REINHARD, TM_REINHARD2, TM_UNCHARTED2, TM_FILMIC, TM_UNREAL, TM_ACES, TM_LOTTES, TM_UCHIMURA };

static float reinhard(float x) { return x / (1.0f + x); }                                            /* ReinhardToneMapper.glsl:32-34 */
static float reinhard2(float x) { return (x * (1.0f + x / (4.0f * 4.0f))) / (1.0f + x); }             /* Reinhard2ToneMapper.glsl:32-35 */
static float uncharted2_curve(float x) {                                                             /* Uncharted2ToneMapper.glsl:32-41 */
    const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
    return ((x * (A * x + C * B) + D * E) / (x * (A * x + B) + D * F)) - E / F;
}
static float uncharted2(float color) {                                                               /* Uncharted2ToneMapper.glsl:43-49 */
    const float W = 11.2f, bias = 2.0f;
    float curr = uncharted2_curve(bias * color);
    float white_scale = 1.0f / uncharted2_curve(W);
    return curr * white_scale;
}
static float filmic(float x) {                                                                       /* FilmicToneMapper.glsl:32-36 */
    float X = vmax(0.0f, x - 0.004f);
    float result = (X * (6.2f * X + 0.5f)) / (X * (6.2f * X + 1.7f) + 0.06f);
    return vpo_powf(result, 2.2f);
}
static float unreal(float x) { return x / (x + 0.155f) * 1.019f; }                                   /* UnrealToneMapper.glsl:32-34 */
static float aces(float x) {                                                                         /* AcesToneMapper.glsl:32-39 */
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    return clamp01((x * (a * x + b)) / (x * (c * x + d) + e));
}
static float lottes(float x) {                                                                       /* LottesToneMapper.glsl:32-47 */
    const float a = 1.6f, d = 0.977f;
    /* const b, c of the shader: float64 evaluation of its constant expressions, rounded once */
    const float b = (float)1.0730397117173704, c = (float)0.16741993817791725;
    return vpo_powf(x, a) / (vpo_powf(x, a * d) * b + c);
}
static float smoothstep_f(float e0, float e1, float x) {
    float t = clamp01((x - e0) / (e1 - e0));
    return t * t * (3.0f - 2.0f * t);
}
static float step_f(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
static float uchimura(float x) {                                                                     /* UchimuraToneMapper.glsl:32-61 */
    const float P = 1.0f, a = 1.0f, m = 0.22f, l = 0.4f, c = 1.33f, b = 0.0f;
    float l0 = ((P - m) * l) / a;
    float S0 = m + l0;
    float S1 = m + a * l0;
    float C2 = (a * P) / (P - S1);
    float CP = -C2 / P;
    float w0 = 1.0f - smoothstep_f(0.0f, m, x);
    float w2 = step_f(m + l0, x);
    float w1 = 1.0f - w0 - w2;
    float T = m * vpo_powf(x / m, c) + b;
    float S = P - (P - S1) * vpo_expf(CP * (x - S0));
    float L = m + a * (x - m);
    return T * w0 + L * w1 + S * w2;
}

/* one pass of tone mapper `kind` over npix RGBA16F texels -> RGBA8 */
VPO_API int vpo_tonemap(int kind, const vpo_tonemap_params *p, const uint16_t *src, uint8_t *dst, size_t npix) {
    if (kind < TM_ARTISTIC || kind > TM_UCHIMURA) return -1;
    for (size_t k = 0; k < npix; k++) {
        float c[4], o[4];
        for (int q = 0; q < 4; q++) c[q] = vpo_f16_to_f32(src[4 * k + q]);
        if (kind == TM_ARTISTIC) {                              /* ArtisticToneMapper.glsl:37-46 */
            for (int q = 0; q < 4; q++) c[q] = (c[q] - p->low) / (p->high - p->low);
            const float gray = 0.57735026918962576f;            /* normalize(vec3(1)) */
            float d = fmaf(c[2], gray, fmaf(c[1], gray, c[0] * gray));
            float g = d * gray;
            float midpoint = (p->mid - p->low) / (p->high - p->low);
            float exponent = -vpo_logf(midpoint) / vpo_logf(2.0f);
            float e = exponent / p->gamma;
            for (int q = 0; q < 3; q++) o[q] = vpo_powf(fmaf(c[q], p->saturation, g * (1.0f - p->saturation)), e);
            o[3] = 1.0f;
        } else if (kind == TM_RANGE) {                          /* RangeToneMapper.glsl:33-36: all four channels */
            float e = 1.0f / p->gamma;
            for (int q = 0; q < 4; q++) o[q] = vpo_powf((c[q] - p->min) / (p->max - p->min), e);
        } else {                                                /* pow(vec4(curve(src.rgb * uExposure), 1), vec4(1.0 / uGamma)) */
            float e = 1.0f / p->gamma;
            for (int q = 0; q < 3; q++) {
                float x = c[q] * p->exposure, y;
                switch (kind) {
                    case TM_REINHARD:   y = reinhard(x); break;
                    case TM_REINHARD2:  y = reinhard2(x); break;
                    case TM_UNCHARTED2: y = uncharted2(x); break;
                    case TM_FILMIC:     y = filmic(x); break;
                    case TM_UNREAL:     y = unreal(x); break;
                    case TM_ACES:       y = aces(x); break;
                    case TM_LOTTES:     y = lottes(x); break;
                    default:            y = uchimura(x); break;
                }
                o[q] = vpo_powf(y, e);
            }
            o[3] = vpo_powf(1.0f, e);
        }
        for (int q = 0; q < 4; q++) dst[4 * k + q] = to_unorm8(o[q]);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------------------------
 * The transfer-function widget's canvas (ui/TransferFunction/TransferFunction.js:110-121: clear, then one full-target draw per bump
 * with gl.blendFunc(ONE, ONE_MINUS_SRC_ALPHA); glsl/TransferFunction.glsl:32-35: oColor = uColor * exp(-r * r),
 * r = length((uPosition - vPosition) / uSize), vPosition = the pixel centre in [0, 1]^2), handed over as texImage2D(canvas) does
 * (AbstractRenderer.js:99-104): texel row 0 = the canvas's top row; `unpremultiply`: the colour divided by alpha again (a premultiplied
 * WebGL canvas uploaded with UNPACK_PREMULTIPLY_ALPHA_WEBGL = false).  bumps: 8 floats each = position.xy, size.xy, color.rgba.
 * Parity unpinned: `precision mediump` exp and the browser's un-premultiplication are implementation-defined; this is the restatement
 * vpt_transfer_function_rasterize is compared with, bit for bit.
 * ------------------------------------------------------------------------------------------------------------------------------ */
VPO_API int vpo_tf_rasterize(const float *bumps, int count, int width, int height, int unpremultiply, uint8_t *out) {
    for (int j = 0; j < height; j++) {
        for (int i = 0; i < width; i++) {
            float u = ((float)i + 0.5f) / (float)width;
            float v = ((float)(height - 1 - j) + 0.5f) / (float)height;
            uint32_t d[4] = { 0u, 0u, 0u, 0u };
            for (int k = 0; k < count; k++) {
                const float *b = bumps + 8 * k;
                float dx = (b[0] - u) / b[2], dy = (b[1] - v) / b[3];
                float xx = dx * dx, yy = dy * dy;
                float r = sqrtf(xx + yy);
                float e = vpo_expf(-(r * r));
                float s[4];
                for (int q = 0; q < 4; q++) s[q] = clamp01(b[4 + q] * e);
                float k1 = 1.0f - s[3];
                for (int q = 0; q < 4; q++) {
                    float t = ((float)d[q] / 255.0f) * k1;
                    d[q] = to_unorm8(s[q] + t);
                }
            }
            if (unpremultiply) {
                if (d[3] == 0u) d[0] = d[1] = d[2] = 0u;
                else for (int q = 0; q < 3; q++) { uint32_t c = (d[q] * 255u + d[3] / 2u) / d[3]; d[q] = c > 255u ? 255u : c; }
            }
            for (int q = 0; q < 4; q++) out[((size_t)j * width + i) * 4 + q] = (uint8_t)d[q];
        }
    }
    return 0;
}
