/*
 * vpt_oracle.c — CPU ORACLE for the MIP / EAM / MCS / MCM renderer passes of MOj0/vpt.
 *
 * THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (vpt_amd/, include/,
 * js/) never includes, links or calls anything in oracle/.
 *
 * What it is: a scalar, strict-fp32 C restatement of the arithmetic of the reference's
 * fragment programs (GLSL ES 3.00) and of the fixed-function GL semantics they rely on
 * (texture filtering, unorm8 / half-float render-target conversion).  Each function
 * cites the reference file:line (relative to /root/reference) it follows.
 *
 * PARITY STATUS: the reference ships no tests, golden images or fixtures for this path
 * (package.json:9) and no WebGL2 driver exists in this container, so no GPU run of the
 * reference pins this file.  Since round 4 it is pinned by the reference's own shader
 * TEXT, executed: oracle/glsl_interp.py interprets src/glsl/renderers/*.glsl and
 * src/glsl/tonemappers/*.glsl as read from the reference tree (tests/golden/
 * make_glsl_fixtures.py -> tests/golden/glsl_r04.json) and tests/test_glsl_reference.py
 * holds this oracle to the results — MIP, EAM and the tone mappers byte for byte, MCS /
 * ISO / Depth within rounding with identical control flow, MCM with identical photon
 * histories.  Where WebGL leaves the rounding to the implementation the interpreter rounds
 * as the contract below does.  Also pinned:
 *   - the inverse-MVP recipe, against outputs of the reference's vendored gl-matrix 3.4.1
 *     run under node (tests/golden/mvp_inverse.json, made by tests/golden/make_mvp_fixture.js);
 *   - the PCG hash / squash / uniform chain, against known answers (tests/golden/pcg_kat.json);
 *   - closed-form analytic checks (homogeneous media, nearest-filter MIP == integer max).
 *
 * NUMERIC CONTRACT ("fixed-seed mode"): all arithmetic is IEEE binary32,
 * round-to-nearest-even, no contraction except where fmaf() is written, denormals kept.
 * log / sin / cos / atan2 / asin are the polynomial routines below (NOT libm) so that a
 * GPU implementation of the same operation sequence is bit-identical.  Compile with
 *   gcc -O2 -ffp-contract=off -fno-fast-math -mfma
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define VPO_API __attribute__((visibility("default")))

typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;
typedef struct { float x, y; } v2;

/* ------------------------------------------------------------------------------------------
 * scene / frame descriptors (ctypes mirrors live in oracle/oracle.py)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const uint8_t *volume;      /* linear R8 volume, x fastest (Volume.js:58-75 texSubImage3D) */
    int32_t nx, ny, nz;
    int32_t filter;             /* 0 = NEAREST, 1 = LINEAR (Volume.js:115-125) */
    const uint8_t *tf_rgba;     /* SRGB8_ALPHA8 transfer function, row 0 first (AbstractRenderer.js:31-44,99-104) */
    int32_t tf_w, tf_h;
    const uint8_t *env_rgba;    /* RGBA8 environment map, LINEAR/CLAMP (RenderingContext.js:90-101) */
    int32_t env_w, env_h;
    int32_t channels;           /* 1 = R8 (RAWReader.js:36-38), 2 = RG8 interleaved: texture(uVolume, p).rg has both (0 means 1) */
    int32_t dtype;              /* 0 = UNSIGNED_BYTE texels, normalised v/255; 1 = FLOAT texels (R32F / R16F widened exactly): the value itself,
                                 * LINEAR-filtered the same way (OES_texture_float_linear, RenderingContext.js:78); `volume` then points at floats */
} vpo_scene;

typedef struct {
    int32_t width, height;      /* full image plane */
    int32_t y0, y1;             /* rows processed by this call: [y0, y1) */
    float mvp_inv[16];          /* uMvpInverseMatrix, column-major (MIPRenderer.js:86-97) */
    float seed;                 /* uRandSeed   (Math.random() in the reference) */
    float offset;               /* uOffset     (MIPRenderer.js:84, EAMRenderer.js:103) */
    float step;                 /* uStepSize = fl32(1/steps) or fl32(1/slices) */
    float extinction;           /* uExtinction */
    float anisotropy;           /* uAnisotropy */
    uint32_t max_bounces;       /* uMaxBounces */
    uint32_t steps;             /* uSteps (MCM) */
    float light_dir[3];         /* uScatteringDirection (MCSRenderer.js:106-117) */
    float mix;                  /* uMix (EAM) / uInvFrameNumber (MCS) */
    float blur;                 /* uBlur */
    float inv_res[2];           /* uInverseResolution */
    int32_t nthreads;           /* OpenMP threads over rows (<=1: scalar) */
    float isovalue;             /* uIsovalue (ISORenderer.js:100) */
    float gradient_step;        /* uGradientStep (ISORenderer.js:168: 0.005) */
    float threshold;            /* uThreshold (DepthRenderer.js:112) */
} vpo_frame;

/* ------------------------------------------------------------------------------------------
 * GLSL built-ins restated
 * ---------------------------------------------------------------------------------------- */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
/* GLSL min / max (ES 3.00 §8.3: undefined for NaN operands).  Contract: IEEE-754 minNum / maxNum — a NaN operand
 * yields the other operand, and -0 orders below +0 (what v_min_f32 / v_max_f32 compute on gfx950). */
static inline float vmin(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    if (a == 0.0f && b == 0.0f) return (f2u(a) >> 31) ? a : b;
    return (b < a) ? b : a;
}
static inline float vmax(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    if (a == 0.0f && b == 0.0f) return (f2u(a) >> 31) ? b : a;
    return (a < b) ? b : a;
}
static inline float vclamp01(float x) { return vmin(vmax(x, 0.0f), 1.0f); }
/* GLSL mix(x,y,a) = x*(1-a) + y*a */
static inline float mixf(float a, float b, float t) { return fmaf(b, t, a * (1.0f - t)); }
static inline v3 mix3(v3 a, v3 b, float t) {
    v3 r = { mixf(a.x, b.x, t), mixf(a.y, b.y, t), mixf(a.z, b.z, t) };
    return r;
}
static inline float dot3(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline float length3(v3 a) { return sqrtf(dot3(a, a)); }
static inline v3 sub3(v3 a, v3 b) { v3 r = { a.x - b.x, a.y - b.y, a.z - b.z }; return r; }
/* software reciprocal: integer seed + 3 Newton-Raphson steps; <= 2 ulp for normal x in [2^-125, 2^125]
 * (GLSL ES 3.00 §4.5.1 allows 2.5 ULP for a/b).  x = 0, inf, NaN -> NaN.  Used for divisors that cannot be 0 in a
 * sane scene (homogeneous w, sample counts, segment lengths); slab tests keep IEEE 1/d. */
VPO_API float vpo_rcp_nr(float x) {
    float r = u2f(0x7EF311C7u - f2u(x));
    r = fmaf(fmaf(-x, r, 1.0f), r, r);
    r = fmaf(fmaf(-x, r, 1.0f), r, r);
    r = fmaf(fmaf(-x, r, 1.0f), r, r);
    return r;
}
/* the same with the iterate clamped to +-FLT_MAX before the last step: 1/(+-0) = +-inf, as the slab test needs;
 * identical to vpo_rcp_nr wherever that is finite (NaN iterates only arise outside the domain, |x| > 2^125 or NaN x). */
VPO_API float vpo_rcp_nrz(float x) {
    float r = u2f(0x7EF311C7u - f2u(x));
    r = fmaf(fmaf(-x, r, 1.0f), r, r);
    r = fmaf(fmaf(-x, r, 1.0f), r, r);
    /* v_med3_f32(r, -FLT_MAX, FLT_MAX): a NaN operand makes the instruction return min3 of the others (= -FLT_MAX) */
    if (r != r) r = -3.402823466e+38f;
    if (r > 3.402823466e+38f) r = 3.402823466e+38f;
    if (r < -3.402823466e+38f) r = -3.402823466e+38f;
    r = fmaf(fmaf(-x, r, 1.0f), r, r);
    return r;
}
/* software reciprocal square root: integer seed + 3 Newton-Raphson steps (normal x > 0) */
VPO_API float vpo_rsqrt_nr(float x) {
    float y = u2f(0x5F375A86u - (f2u(x) >> 1));
    float h = 0.5f * x;
    y = y * fmaf(-h * y, y, 1.5f);
    y = y * fmaf(-h * y, y, 1.5f);
    y = y * fmaf(-h * y, y, 1.5f);
    return y;
}
/* sqrt(x) = x * inversesqrt(x) (GLSL ES 3.00 §4.5.1: sqrt inherits the precision of 1/inversesqrt); sqrt(+0) = +0 */
VPO_API float vpo_sqrt_nr(float x) { return x * vpo_rsqrt_nr(x); }
/* GLSL normalize(v) = v * inversesqrt(dot(v, v)) */
static inline v3 normalize3(v3 a) {
    float inv = vpo_rsqrt_nr(dot3(a, a));
    v3 r = { a.x * inv, a.y * inv, a.z * inv };
    return r;
}
/* p + t*d */
static inline v3 madd3(v3 p, float t, v3 d) {
    v3 r = { fmaf(t, d.x, p.x), fmaf(t, d.y, p.y), fmaf(t, d.z, p.z) };
    return r;
}

/* ------------------------------------------------------------------------------------------
 * deterministic transcendental routines (the contract's log / sin / cos / atan2 / asin)
 * ---------------------------------------------------------------------------------------- */
/* natural log for x in {0} U [2^-126, +inf]; x == 0 -> -inf.  Cephes-style minimax polynomial. */
VPO_API float vpo_logf(float x) {
    if (x == 0.0f) return -INFINITY;
    if (!(x > 0.0f)) return NAN;
    if (x == INFINITY) return INFINITY;
    uint32_t b = f2u(x);
    int32_t e = (int32_t)(b >> 23) - 126;                 /* x = m * 2^e, m in [0.5,1) */
    float m = u2f((b & 0x007fffffu) | 0x3f000000u);
    if (m < 0.70710678118654752440f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
    float fe = (float)e;
    float z = m * m;
    float p = 7.0376836292E-2f;
    p = fmaf(p, m, -1.1514610310E-1f);
    p = fmaf(p, m, 1.1676998740E-1f);
    p = fmaf(p, m, -1.2420140846E-1f);
    p = fmaf(p, m, 1.4249322787E-1f);
    p = fmaf(p, m, -1.6668057665E-1f);
    p = fmaf(p, m, 2.0000714765E-1f);
    p = fmaf(p, m, -2.4999993993E-1f);
    p = fmaf(p, m, 3.3333331174E-1f);
    float y = (p * m) * z;
    y = fmaf(-2.12194440e-4f, fe, y);
    y = fmaf(-0.5f, z, y);
    float r = m + y;
    r = fmaf(0.693359375f, fe, r);
    return r;
}

/* sin and cos of a >= 0 (used on [0, 2*pi]); Cody-Waite reduction by pi/2 + Cephes kernels. */
VPO_API void vpo_sincosf(float a, float *s_out, float *c_out) {
    float q = rintf(a * 0.63661977236758134308f);
    float r = fmaf(q, -1.5703125f, a);
    r = fmaf(q, -4.837512969970703125e-4f, r);
    r = fmaf(q, -7.54978995489188216e-8f, r);
    float r2 = r * r;
    float ps = -1.9515295891E-4f;
    ps = fmaf(ps, r2, 8.3321608736E-3f);
    ps = fmaf(ps, r2, -1.6666654611E-1f);
    float sr = fmaf(ps * r2, r, r);
    float pc = 2.443315711809948E-005f;
    pc = fmaf(pc, r2, -1.388731625493765E-003f);
    pc = fmaf(pc, r2, 4.166664568298827E-002f);
    float cr = fmaf(pc * r2, r2, fmaf(-0.5f, r2, 1.0f));
    int32_t n = ((int32_t)q) & 3;
    float s, c;
    if (n == 0)      { s = sr;  c = cr;  }
    else if (n == 1) { s = cr;  c = -sr; }
    else if (n == 2) { s = -sr; c = -cr; }
    else             { s = -cr; c = sr;  }
    *s_out = s; *c_out = c;
}

/* atan for t in [0, +inf] (Cephes atanf ranges) */
static float atan_pos(float t) {
    float y0;
    if (t > 2.414213562373095f) { y0 = 1.5707963267948966f; t = -(1.0f / t); }
    else if (t > 0.4142135623730950f) { y0 = 0.7853981633974483f; t = (t - 1.0f) / (t + 1.0f); }
    else { y0 = 0.0f; }
    float z = t * t;
    float p = 8.05374449538e-2f;
    p = fmaf(p, z, -1.38776856032E-1f);
    p = fmaf(p, z, 1.99777106478E-1f);
    p = fmaf(p, z, -3.33329491539E-1f);
    float r = fmaf(p * z, t, t);
    return y0 + r;
}
/* GLSL atan(y, x) */
VPO_API float vpo_atan2f(float y, float x) {
    if (x != x || y != y) return NAN;
    float ay = fabsf(y), ax = fabsf(x);
    float r;
    if (ax == 0.0f && ay == 0.0f) r = 0.0f;
    else if (ax == INFINITY && ay == INFINITY) r = 0.7853981633974483f;
    else r = atan_pos(ay / ax);                    /* ay/0 = +inf -> pi/2 */
    if (f2u(x) >> 31) r = 3.14159265358979323846f - r;
    return (f2u(y) >> 31) ? -r : r;
}
/* GLSL asin(x); |x| > 1 -> NaN */
VPO_API float vpo_asinf(float x) {
    float a = fabsf(x);
    if (!(a <= 1.0f)) return NAN;
    float z, t;
    int big = a > 0.5f;
    if (big) { z = 0.5f * (1.0f - a); t = sqrtf(z); } else { t = a; z = a * a; }
    float p = 4.2163199048E-2f;
    p = fmaf(p, z, 2.4181311049E-2f);
    p = fmaf(p, z, 4.5470025998E-2f);
    p = fmaf(p, z, 7.4953002686E-2f);
    p = fmaf(p, z, 1.6666752422E-1f);
    float r = fmaf(p * z, t, t);
    if (big) r = 1.5707963267948966f - (r + r);
    return (f2u(x) >> 31) ? -r : r;
}

/* ------------------------------------------------------------------------------------------
 * RNG: mixins/random/hash/pcg.glsl:3-7, squashlinear.glsl:7-9, distribution/uniformdivision.glsl:3-6
 * ---------------------------------------------------------------------------------------- */
VPO_API uint32_t vpo_pcg(uint32_t x) {
    x = x * 747796405u + 2891336453u;
    x = ((x >> ((x >> 28u) + 4u)) ^ x) * 277803737u;
    return (x >> 22u) ^ x;
}
VPO_API uint32_t vpo_hash3(uint32_t x, uint32_t y, uint32_t z) {
    return vpo_pcg(19u * x + 47u * y + 101u * z + 131u);
}
/* float(state) / float(~0u): float(~0u) rounds to 2^32, so the quotient is exact; range [0,1] */
VPO_API float vpo_random_uniform(uint32_t *state) {
    *state = vpo_pcg(*state);
    return (float)(*state) * 0x1p-32f;
}
/* distribution/exponential.glsl:3-5: -log(u)/rate, evaluated as -log(u) * inv_rate with inv_rate = 1/rate
 * computed once per pass (contract, DESIGN.md §3: a division by a pass-uniform value is one IEEE reciprocal + multiply;
 * GLSL ES 3.00 §4.5.1 allows 2.5 ULP for a/b) */
static inline float random_exponential(uint32_t *state, float inv_rate) {
    return -vpo_logf(vpo_random_uniform(state)) * inv_rate;
}
/* distribution/square.glsl:3-7 */
static inline v2 random_square(uint32_t *state) {
    v2 r; r.x = vpo_random_uniform(state); r.y = vpo_random_uniform(state); return r;
}
/* distribution/disk.glsl:3-7, constants.glsl:4 (TWOPI 6.28318530718) */
static inline v2 random_disk(uint32_t *state) {
    float radius = vpo_sqrt_nr(vpo_random_uniform(state));
    float angle = 6.28318530718f * vpo_random_uniform(state);
    float s, c; vpo_sincosf(angle, &s, &c);
    v2 r = { radius * c, radius * s };
    return r;
}
/* distribution/sphere.glsl:4-10 (Marsaglia) */
static inline v3 random_sphere(uint32_t *state) {
    v2 d = random_disk(state);
    float norm = fmaf(d.y, d.y, d.x * d.x);
    float radius = 2.0f * vpo_sqrt_nr(1.0f - norm);
    float z = fmaf(-2.0f, norm, 1.0f);
    v3 r = { radius * d.x, radius * d.y, z };
    return r;
}
VPO_API void vpo_random_sphere(uint32_t *state, float *out3) {
    v3 r = random_sphere(state); out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}

/* ------------------------------------------------------------------------------------------
 * fixed-function GL semantics
 * ---------------------------------------------------------------------------------------- */
/* float -> unorm8 render-target write: round-to-nearest-even of clamp(f,0,1)*255; NaN -> 0 */
static inline uint8_t to_unorm8(float f) {
    float c = vclamp01(f);
    if (c != c) c = 0.0f;
    return (uint8_t)rintf(c * 255.0f);
}
static inline float from_unorm8(uint8_t c) { return (float)c / 255.0f; }

/* float -> IEEE half, round-to-nearest-even (RGBA16F render target, AbstractRenderer.js:142-155) */
VPO_API uint16_t vpo_f32_to_f16(float f) {
    uint32_t x = f2u(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? (0x200u | ((ax >> 13) & 0x3ffu)) : 0u));
    if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);           /* rounds to >= 65520 -> inf */
    if (ax < 0x33000001u) return (uint16_t)sign;                         /* <= 2^-25 -> 0 (tie to even) */
    int32_t e = (int32_t)(ax >> 23) - 127;
    uint32_t m = (ax & 0x7fffffu) | 0x800000u;
    uint32_t shift, base;
    if (e < -14) { shift = (uint32_t)(13 + (-14 - e)); base = 0; }       /* subnormal half */
    else { shift = 13; base = (uint32_t)(e + 15) << 10; m &= 0x7fffffu; }
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    return (uint16_t)(sign | (base + q));
}

/* SRGB8 decode (GL ES 3.0 §3.8.16); done in double, rounded once to float */
static float srgb_to_linear(uint8_t c) {
    double cs = (double)c / 255.0;
    double cl = (cs <= 0.04045) ? cs / 12.92 : pow((cs + 0.055) / 1.055, 2.4);
    return (float)cl;
}
VPO_API float vpo_srgb_to_linear(uint8_t c) { return srgb_to_linear(c); }

/* LINEAR filter coordinate: u = s*N - 0.5 clamped into [-1, N]; i0 = floor(u); f = u - i0.
 * Clamping first keeps float->int conversion defined for inf/NaN (NaN -> -1). Texel indices are
 * clamped afterwards (CLAMP_TO_EDGE), so the clamp of u never changes a finite in-range result. */
static inline void linear_coord(float s, int32_t n, int32_t *i0, int32_t *i1, float *f) {
    float u = fmaf(s, (float)n, -0.5f);
    if (!(u > -1.0f)) u = -1.0f;
    if (u > (float)n) u = (float)n;
    float fl = floorf(u);
    int32_t i = (int32_t)fl;
    *f = u - fl;
    int32_t a = i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
    int32_t b = i + 1 < 0 ? 0 : (i + 1 > n - 1 ? n - 1 : i + 1);
    *i0 = a; *i1 = b;
}
static inline int32_t nearest_coord(float s, int32_t n) {
    float u = s * (float)n;
    if (!(u > 0.0f)) u = 0.0f;
    if (u > (float)(n - 1)) u = (float)(n - 1);
    return (int32_t)floorf(u);
}
static inline float lerpf(float a, float b, float f) { return fmaf(f, b - a, a); }
/* texel normalisation: one multiply by fl32(1/255) (255 * VPO_INV255 == 1.0f exactly) */
#define VPO_INV255 0.00392156862745098f

/* texture(uVolume, p).r — R8 normalised, CLAMP_TO_EDGE (Volume.js:49-60).  Interpolates the integer
 * texel values (x, then y, then z) and normalises once by * fl32(1/255). */
static float sample_volume_channel(const vpo_scene *sc, v3 p, int channel) {
    size_t nch = sc->channels == 2 ? 2 : 1;
    size_t sx = nch, sy = nch * (size_t)sc->nx, sz = nch * (size_t)sc->nx * (size_t)sc->ny;
    const int f32 = sc->dtype == 1;
    const uint8_t *v = sc->volume + channel;
    const float *vf = (const float *)sc->volume + channel;
#define TEXEL(ix, iy, iz) (f32 ? vf[(ix) * sx + (iy) * sy + (iz) * sz] : (float)v[(ix) * sx + (iy) * sy + (iz) * sz])
    const float norm = f32 ? 1.0f : VPO_INV255;          /* float texels are not normalised (x * 1.0f is exact) */
    if (sc->filter == 0) {
        int32_t x = nearest_coord(p.x, sc->nx), y = nearest_coord(p.y, sc->ny), z = nearest_coord(p.z, sc->nz);
        return TEXEL(x, y, z) * norm;
    }
    int32_t x0, x1, y0, y1, z0, z1; float fx, fy, fz;
    linear_coord(p.x, sc->nx, &x0, &x1, &fx);
    linear_coord(p.y, sc->ny, &y0, &y1, &fy);
    linear_coord(p.z, sc->nz, &z0, &z1, &fz);
    float c000 = TEXEL(x0, y0, z0), c100 = TEXEL(x1, y0, z0);
    float c010 = TEXEL(x0, y1, z0), c110 = TEXEL(x1, y1, z0);
    float c001 = TEXEL(x0, y0, z1), c101 = TEXEL(x1, y0, z1);
    float c011 = TEXEL(x0, y1, z1), c111 = TEXEL(x1, y1, z1);
#undef TEXEL
    float c00 = lerpf(c000, c100, fx), c10 = lerpf(c010, c110, fx);
    float c01 = lerpf(c001, c101, fx), c11 = lerpf(c011, c111, fx);
    float c0 = lerpf(c00, c10, fy), c1 = lerpf(c01, c11, fy);
    return lerpf(c0, c1, fz) * norm;
}
static inline float sample_volume(const vpo_scene *sc, v3 p) { return sample_volume_channel(sc, p, 0); }

/* decoded float4 tables, built once per call */
typedef struct {
    const vpo_scene *sc;
    v4 *tf;     /* tf_w*tf_h, sRGB-decoded rgb, linear alpha */
    v4 *env;    /* env_w*env_h, c/255 */
} scene_tables;

static void tables_init(scene_tables *t, const vpo_scene *sc) {
    t->sc = sc;
    size_t n = (size_t)sc->tf_w * (size_t)sc->tf_h;
    t->tf = (v4 *)malloc(n * sizeof(v4));
    for (size_t i = 0; i < n; i++) {
        const uint8_t *c = sc->tf_rgba + 4 * i;
        t->tf[i].x = srgb_to_linear(c[0]); t->tf[i].y = srgb_to_linear(c[1]);
        t->tf[i].z = srgb_to_linear(c[2]); t->tf[i].w = from_unorm8(c[3]);
    }
    n = (size_t)sc->env_w * (size_t)sc->env_h;
    t->env = (v4 *)malloc(n * sizeof(v4));
    for (size_t i = 0; i < n; i++) {
        const uint8_t *c = sc->env_rgba + 4 * i;
        t->env[i].x = from_unorm8(c[0]); t->env[i].y = from_unorm8(c[1]);
        t->env[i].z = from_unorm8(c[2]); t->env[i].w = from_unorm8(c[3]);
    }
}
static void tables_free(scene_tables *t) { free(t->tf); free(t->env); }

static inline v4 lerp4(v4 a, v4 b, float f) {
    v4 r = { lerpf(a.x, b.x, f), lerpf(a.y, b.y, f), lerpf(a.z, b.z, f), lerpf(a.w, b.w, f) };
    return r;
}
/* 2D LINEAR / CLAMP_TO_EDGE lookup in a float4 table (x lerps first, then y) */
static v4 sample_2d(const v4 *tex, int32_t w, int32_t h, float s, float t) {
    int32_t x0, x1, y0, y1; float fx, fy;
    linear_coord(s, w, &x0, &x1, &fx);
    linear_coord(t, h, &y0, &y1, &fy);
    v4 a = lerp4(tex[(size_t)y0 * w + x0], tex[(size_t)y0 * w + x1], fx);
    v4 b = lerp4(tex[(size_t)y1 * w + x0], tex[(size_t)y1 * w + x1], fx);
    return lerp4(a, b, fy);
}

/* sampleVolumeColor: MIPRenderer.glsl:45-49 (= EAM :46-50, MCS :64-68, MCM :85-89).
 * R8 volume => .rg = (r, 0); RG8 volume => both channels filtered, the transfer function is looked up in 2-D. */
static inline v4 sample_volume_color(const scene_tables *t, v3 p, uint64_t *ns) {
    float r = sample_volume(t->sc, p);
    float g = t->sc->channels == 2 ? sample_volume_channel(t->sc, p, 1) : 0.0f;
    (*ns)++;
    return sample_2d(t->tf, t->sc->tf_w, t->sc->tf_h, r, g);
}

/* sampleEnvironmentMap: MCSRenderer.glsl:59-62, MCMRenderer.glsl:80-83 (INVPI 0.31830988618).
 * Contract: a 1x1 map is the constant texel (any coordinate, NaN included). */
static v4 sample_environment(const scene_tables *t, v3 d) {
    if (t->sc->env_w == 1 && t->sc->env_h == 1) return t->env[0];
    float a = vpo_atan2f(d.x, -d.z);
    float b = vpo_asinf(-d.y) * 2.0f;
    float s = fmaf(a * 0.31830988618f, 0.5f, 0.5f);
    float tt = fmaf(b * 0.31830988618f, 0.5f, 0.5f);
    return sample_2d(t->env, t->sc->env_w, t->sc->env_h, s, tt);
}

/* ------------------------------------------------------------------------------------------
 * ray set-up
 * ---------------------------------------------------------------------------------------- */
/* inverseMvp * (x, y, z, 1), column-major, fma chain */
static inline v4 mat4_mul_point(const float *m, float x, float y, float z) {
    v4 r;
    /* constant part first (m[8..11]*z + m[12..15], z = +-1), then the x and y terms; GLSL leaves the order open */
    r.x = fmaf(m[4], y, fmaf(m[0], x, fmaf(m[8],  z, m[12])));
    r.y = fmaf(m[5], y, fmaf(m[1], x, fmaf(m[9],  z, m[13])));
    r.z = fmaf(m[6], y, fmaf(m[2], x, fmaf(m[10], z, m[14])));
    r.w = fmaf(m[7], y, fmaf(m[3], x, fmaf(m[11], z, m[15])));
    return r;
}
/* mixins/unproject.glsl:3-10, evaluated at the pixel's own NDC (the reference interpolates the
 * three vertex results, which is the same affine map). */
static inline void unproject(float px, float py, const float *m, v3 *from, v3 *to) {
    v4 n = mat4_mul_point(m, px, py, -1.0f);
    v4 f = mat4_mul_point(m, px, py, 1.0f);
    float in = vpo_rcp_nr(n.w), jf = vpo_rcp_nr(f.w);   /* xyz / w as xyz * rcp(w) */
    from->x = n.x * in; from->y = n.y * in; from->z = n.z * in;
    to->x = f.x * jf; to->y = f.y * jf; to->z = f.z * jf;
}
/* pixel centre in NDC: 2*(i+0.5)/W - 1, written (2i+1)/W - 1 */
static inline float pixel_ndc(int32_t i, int32_t n) { return (float)(2 * i + 1) / (float)n - 1.0f; }
/* vPosition = position*0.5 + 0.5 (MIPRenderer.glsl:87, MCMRenderer.glsl:118) */
static inline float ndc_to_uv(float p) { return fmaf(p, 0.5f, 0.5f); }

/* mixins/intersectCube.glsl:3-11 */
static inline v2 intersect_cube(v3 o, v3 d) {
    v3 inv = { vpo_rcp_nrz(d.x), vpo_rcp_nrz(d.y), vpo_rcp_nrz(d.z) };     /* (a - o) / d as (a - o) * rcp(d), rcp(+-0) = +-inf */
    v3 tmin = { (0.0f - o.x) * inv.x, (0.0f - o.y) * inv.y, (0.0f - o.z) * inv.z };
    v3 tmax = { (1.0f - o.x) * inv.x, (1.0f - o.y) * inv.y, (1.0f - o.z) * inv.z };
    v3 t1 = { vmin(tmin.x, tmax.x), vmin(tmin.y, tmax.y), vmin(tmin.z, tmax.z) };
    v3 t2 = { vmax(tmin.x, tmax.x), vmax(tmin.y, tmax.y), vmax(tmin.z, tmax.z) };
    v2 r = { vmax(vmax(t1.x, t1.y), t1.z), vmin(vmin(t2.x, t2.y), t2.z) };
    return r;
}

#define FOR_ROWS(fr) \
    _Pragma("omp parallel for schedule(dynamic, 1) reduction(+:ns) num_threads(nth)") \
    for (int32_t j = (fr)->y0; j < (fr)->y1; j++)

static inline int clamp_threads(const vpo_frame *fr) { return fr->nthreads > 1 ? fr->nthreads : 1; }

/* ==========================================================================================
 * MIP  (MIPRenderer.glsl)
 * ======================================================================================== */
/* generate/fragment main(): MIPRenderer.glsl:51-72.  frame: R8, row-major, row 0 = bottom. */
VPO_API uint64_t vpo_mip_generate(const vpo_scene *sc, const vpo_frame *fr, uint8_t *frame) {
    scene_tables t; tables_init(&t, sc);
    uint64_t ns = 0; int nth = clamp_threads(fr);
    FOR_ROWS(fr) {
        for (int32_t i = 0; i < fr->width; i++) {
            v3 rf, rt;
            unproject(pixel_ndc(i, fr->width), pixel_ndc(j, fr->height), fr->mvp_inv, &rf, &rt);
            v3 dir = sub3(rt, rf);
            v2 tb = intersect_cube(rf, dir);
            tb.x = vmax(tb.x, 0.0f); tb.y = vmax(tb.y, 0.0f);
            float out;
            if (tb.x >= tb.y) {
                out = 0.0f;
            } else {
                v3 from = mix3(rf, rt, tb.x), to = mix3(rf, rt, tb.y);
                float tt = 0.0f, val = 0.0f, offset = fr->offset;
                do {
                    v3 pos = mix3(from, to, offset);
                    val = vmax(sample_volume_color(&t, pos, &ns).w, val);
                    tt += fr->step;
                    float m = offset + fr->step;            /* mod(offset + uStepSize, 1.0) */
                    offset = m - floorf(m);
                } while (tt < 1.0f);
                out = val;
            }
            frame[(size_t)j * fr->width + i] = to_unorm8(out);
        }
    }
    tables_free(&t);
    return ns;
}
/* integrate: MIPRenderer.glsl:105-109 — max of two unorm8 == integer max */
VPO_API void vpo_mip_integrate(const vpo_frame *fr, uint8_t *acc, const uint8_t *frame) {
    for (int32_t j = fr->y0; j < fr->y1; j++)
        for (int32_t i = 0; i < fr->width; i++) {
            size_t k = (size_t)j * fr->width + i;
            float a = from_unorm8(acc[k]), f = from_unorm8(frame[k]);
            acc[k] = to_unorm8(vmax(a, f));
        }
}
/* render: MIPRenderer.glsl:141-144 -> RGBA16F */
VPO_API void vpo_mip_render(const vpo_frame *fr, const uint8_t *acc, uint16_t *out) {
    for (int32_t j = fr->y0; j < fr->y1; j++)
        for (int32_t i = 0; i < fr->width; i++) {
            size_t k = (size_t)j * fr->width + i;
            uint16_t h = vpo_f32_to_f16(from_unorm8(acc[k]));
            out[4 * k + 0] = h; out[4 * k + 1] = h; out[4 * k + 2] = h; out[4 * k + 3] = 0x3c00u;
        }
}
/* reset: MIPRenderer.glsl:168-170 */
VPO_API void vpo_mip_reset(const vpo_frame *fr, uint8_t *acc) {
    for (int32_t j = fr->y0; j < fr->y1; j++) memset(acc + (size_t)j * fr->width, 0, (size_t)fr->width);
}

/* ==========================================================================================
 * EAM  (EAMRenderer.glsl)
 * ======================================================================================== */
/* generate/fragment main(): EAMRenderer.glsl:52-80.  frame: RGBA8. */
VPO_API uint64_t vpo_eam_generate(const vpo_scene *sc, const vpo_frame *fr, uint8_t *frame) {
    scene_tables t; tables_init(&t, sc);
    uint64_t ns = 0; int nth = clamp_threads(fr);
    FOR_ROWS(fr) {
        for (int32_t i = 0; i < fr->width; i++) {
            v3 rf, rt;
            unproject(pixel_ndc(i, fr->width), pixel_ndc(j, fr->height), fr->mvp_inv, &rf, &rt);
            v3 dir = sub3(rt, rf);
            v2 tb = intersect_cube(rf, dir);
            tb.x = vmax(tb.x, 0.0f); tb.y = vmax(tb.y, 0.0f);
            v4 o = { 0.0f, 0.0f, 0.0f, 1.0f };
            if (!(tb.x >= tb.y)) {
                v3 from = mix3(rf, rt, tb.x), to = mix3(rf, rt, tb.y);
                float ray_step_length = length3(sub3(from, to)) * fr->step;
                float tt = fr->step * fr->offset;
                v4 acc = { 0.0f, 0.0f, 0.0f, 0.0f };
                float k = ray_step_length * fr->extinction;
                while (tt < 1.0f && acc.w < 0.99f) {
                    v3 pos = mix3(from, to, tt);
                    v4 c = sample_volume_color(&t, pos, &ns);
                    c.w *= k;
                    c.x *= c.w; c.y *= c.w; c.z *= c.w;
                    float w = 1.0f - acc.w;
                    acc.x = fmaf(w, c.x, acc.x); acc.y = fmaf(w, c.y, acc.y);
                    acc.z = fmaf(w, c.z, acc.z); acc.w = fmaf(w, c.w, acc.w);
                    tt += fr->step;
                }
                if (acc.w > 1.0f) { float ia = vpo_rcp_nr(acc.w); acc.x *= ia; acc.y *= ia; acc.z *= ia; }
                o.x = acc.x; o.y = acc.y; o.z = acc.z;
            }
            uint8_t *px = frame + 4 * ((size_t)j * fr->width + i);
            px[0] = to_unorm8(o.x); px[1] = to_unorm8(o.y); px[2] = to_unorm8(o.z); px[3] = to_unorm8(o.w);
        }
    }
    tables_free(&t);
    return ns;
}
/* integrate: EAMRenderer.glsl:115-119 — mix(acc, frame, uMix) re-quantised to RGBA8 */
VPO_API void vpo_eam_integrate(const vpo_frame *fr, uint8_t *acc, const uint8_t *frame) {
    for (int32_t j = fr->y0; j < fr->y1; j++)
        for (int32_t i = 0; i < 4 * fr->width; i++) {
            size_t k = (size_t)j * 4 * fr->width + i;
            acc[k] = to_unorm8(mixf(from_unorm8(acc[k]), from_unorm8(frame[k]), fr->mix));
        }
}
/* render: EAMRenderer.glsl:151-153 */
VPO_API void vpo_eam_render(const vpo_frame *fr, const uint8_t *acc, uint16_t *out) {
    for (int32_t j = fr->y0; j < fr->y1; j++)
        for (int32_t i = 0; i < 4 * fr->width; i++) {
            size_t k = (size_t)j * 4 * fr->width + i;
            out[k] = vpo_f32_to_f16(from_unorm8(acc[k]));
        }
}
/* reset: EAMRenderer.glsl:177-179 */
VPO_API void vpo_eam_reset(const vpo_frame *fr, uint8_t *acc) {
    for (int32_t j = fr->y0; j < fr->y1; j++)
        for (int32_t i = 0; i < fr->width; i++) {
            uint8_t *px = acc + 4 * ((size_t)j * fr->width + i);
            px[0] = 0; px[1] = 0; px[2] = 0; px[3] = 255;
        }
}

/* ==========================================================================================
 * ISO  (ISORenderer.glsl) — SURVEY section 8f row 3.  frame / accumulation: RGBA16F "closest hit" (xyz, t)
 * (ISORenderer.js:165-197), so every stored value is a half; render shades it into the RGBA16F render buffer.
 * ======================================================================================== */
float vpo_f16_to_f32(uint16_t h);                                         /* vpt_tonemap_oracle.c */
static inline void store_half4(uint16_t *px, float x, float y, float z, float w) {
    px[0] = vpo_f32_to_f16(x); px[1] = vpo_f32_to_f16(y); px[2] = vpo_f32_to_f16(z); px[3] = vpo_f32_to_f16(w);
}
/* generate/fragment main(): ISORenderer.glsl:52-76 — back-to-front march, the LAST hit written is the closest one.
 * fr->steps = uSteps, fr->step = fl(1 / float(uSteps)) (the shader's own IEEE division, :64). */
VPO_API uint64_t vpo_iso_generate(const vpo_scene *sc, const vpo_frame *fr, uint16_t *frame) {
    scene_tables t; tables_init(&t, sc);
    uint64_t ns = 0; int nth = clamp_threads(fr);
    FOR_ROWS(fr) {
        for (int32_t i = 0; i < fr->width; i++) {
            v3 rf, rt;
            unproject(pixel_ndc(i, fr->width), pixel_ndc(j, fr->height), fr->mvp_inv, &rf, &rt);
            v3 dir = sub3(rt, rf);
            v2 tb = intersect_cube(rf, dir);
            tb.x = vmax(tb.x, 0.0f); tb.y = vmax(tb.y, 0.0f);
            v4 closest = { -1.0f, -1.0f, -1.0f, -1.0f };
            if (!(tb.x >= tb.y)) {
                v3 from = mix3(rf, rt, tb.x), to = mix3(rf, rt, tb.y);
                float tt = 1.0f - fr->offset * fr->step;
                for (uint32_t s = 0; s < fr->steps; s++) {
                    v3 pos = mix3(from, to, tt);
                    float value = sample_volume_color(&t, pos, &ns).w;
                    if (value >= fr->isovalue) { closest.x = pos.x; closest.y = pos.y; closest.z = pos.z; closest.w = tt; }
                    tt -= fr->step;
                }
            }
            store_half4(frame + 4 * ((size_t)j * fr->width + i), closest.x, closest.y, closest.z, closest.w);
        }
    }
    tables_free(&t);
    return ns;
}
/* integrate: ISORenderer.glsl:111-121 — keep the hit with the smaller positive t */
VPO_API void vpo_iso_integrate(const vpo_frame *fr, uint16_t *acc, const uint16_t *frame) {
    for (int32_t j = fr->y0; j < fr->y1; j++)
        for (int32_t i = 0; i < fr->width; i++) {
            size_t k = 4 * ((size_t)j * fr->width + i);
            float fw = vpo_f16_to_f32(frame[k + 3]), aw = vpo_f16_to_f32(acc[k + 3]);
            int take_frame = (fw > 0.0f && aw > 0.0f) ? (fw < aw) : (fw > 0.0f);
            if (take_frame) memcpy(acc + k, frame + k, 8);
        }
}
/* gradient(): ISORenderer.glsl:165-177 — central differences of the transfer function's alpha */
static v3 iso_gradient(const scene_tables *t, v3 p, float h, uint64_t *ns) {
    v3 px = { p.x + h, p.y, p.z }, py = { p.x, p.y + h, p.z }, pz = { p.x, p.y, p.z + h };
    v3 nx = { p.x - h, p.y, p.z }, ny = { p.x, p.y - h, p.z }, nz = { p.x, p.y, p.z - h };
    v3 pos = { sample_volume_color(t, px, ns).w, sample_volume_color(t, py, ns).w, sample_volume_color(t, pz, ns).w };
    v3 neg = { sample_volume_color(t, nx, ns).w, sample_volume_color(t, ny, ns).w, sample_volume_color(t, nz, ns).w };
    float d = 2.0f * h;
    v3 g = { (pos.x - neg.x) / d, (pos.y - neg.y) / d, (pos.z - neg.z) / d };
    return g;
}
/* render/fragment main(): ISORenderer.glsl:179-191.  fr->light_dir = uLight (model space, normalised by the host,
 * ISORenderer.js:152-166). */
VPO_API uint64_t vpo_iso_render(const vpo_scene *sc, const vpo_frame *fr, const uint16_t *acc, uint16_t *out) {
    scene_tables t; tables_init(&t, sc);
    uint64_t ns = 0; int nth = clamp_threads(fr);
    v3 light = { fr->light_dir[0], fr->light_dir[1], fr->light_dir[2] };
    FOR_ROWS(fr) {
        for (int32_t i = 0; i < fr->width; i++) {
            size_t k = 4 * ((size_t)j * fr->width + i);
            float w = vpo_f16_to_f32(acc[k + 3]);
            if (w > 0.0f) {
                v3 pos = { vpo_f16_to_f32(acc[k]), vpo_f16_to_f32(acc[k + 1]), vpo_f16_to_f32(acc[k + 2]) };
                v3 normal = normalize3(iso_gradient(&t, pos, fr->gradient_step, &ns));
                float lambert = vmax(dot3(normal, light), 0.0f);
                v4 material = sample_volume_color(&t, pos, &ns);
                store_half4(out + k, material.x * lambert, material.y * lambert, material.z * lambert, 1.0f);
            } else {
                store_half4(out + k, 1.0f, 1.0f, 1.0f, 1.0f);
            }
        }
    }
    tables_free(&t);
    return ns;
}
/* reset: ISORenderer.glsl:215-217 */
VPO_API void vpo_iso_reset(const vpo_frame *fr, uint16_t *acc) {
    for (int32_t j = fr->y0; j < fr->y1; j++)
        for (int32_t i = 0; i < 4 * fr->width; i++) acc[(size_t)j * 4 * fr->width + i] = 0xbc00u;      /* half(-1) */
}

/* ==========================================================================================
 * Depth  (DepthRenderer.glsl) — SURVEY section 8f row 3.  frame / accumulation: R32F (DepthRenderer.js:165-189).
 * ======================================================================================== */
/* generate/fragment main(): DepthRenderer.glsl:53-79 */
VPO_API uint64_t vpo_depth_generate(const vpo_scene *sc, const vpo_frame *fr, float *frame) {
    scene_tables t; tables_init(&t, sc);
    uint64_t ns = 0; int nth = clamp_threads(fr);
    FOR_ROWS(fr) {
        for (int32_t i = 0; i < fr->width; i++) {
            v3 rf, rt;
            unproject(pixel_ndc(i, fr->width), pixel_ndc(j, fr->height), fr->mvp_inv, &rf, &rt);
            v3 dir = sub3(rt, rf);
            v2 tb = intersect_cube(rf, dir);
            tb.x = vmax(tb.x, 0.0f); tb.y = vmax(tb.y, 0.0f);
            float depth = -1.0f;
            if (!(tb.x >= tb.y)) {
                v3 from = mix3(rf, rt, tb.x), to = mix3(rf, rt, tb.y);
                float ray_step_length = length3(sub3(from, to)) * fr->step;
                float tt = fr->step * fr->offset;
                float accumulator = 0.0f;
                while (tt < 1.0f && accumulator < fr->threshold) {
                    v3 pos = mix3(from, to, tt);
                    float a = sample_volume_color(&t, pos, &ns).w;
                    accumulator += (1.0f - accumulator) * a * ray_step_length * fr->extinction;
                    tt += fr->step;
                }
                if (!(accumulator < fr->threshold)) depth = mixf(tb.x, tb.y, tt);
            }
            frame[(size_t)j * fr->width + i] = depth;
        }
    }
    tables_free(&t);
    return ns;
}
/* integrate: DepthRenderer.glsl:114-118 — mix(accumulator, frame, uMix), the R32F target keeps .r */
VPO_API void vpo_depth_integrate(const vpo_frame *fr, float *acc, const float *frame) {
    for (int32_t j = fr->y0; j < fr->y1; j++)
        for (int32_t i = 0; i < fr->width; i++) {
            size_t k = (size_t)j * fr->width + i;
            acc[k] = mixf(acc[k], frame[k], fr->mix);
        }
}
/* render: DepthRenderer.glsl:150-153 */
VPO_API void vpo_depth_render(const vpo_frame *fr, const float *acc, uint16_t *out) {
    for (int32_t j = fr->y0; j < fr->y1; j++)
        for (int32_t i = 0; i < fr->width; i++) {
            size_t k = (size_t)j * fr->width + i;
            uint16_t h = vpo_f32_to_f16(acc[k]);
            out[4 * k + 0] = h; out[4 * k + 1] = h; out[4 * k + 2] = h; out[4 * k + 3] = 0x3c00u;
        }
}
/* reset: DepthRenderer.glsl:177-179 — vec4(0, 0, 0, 1) into an R32F target */
VPO_API void vpo_depth_reset(const vpo_frame *fr, float *acc) {
    for (int32_t j = fr->y0; j < fr->y1; j++)
        for (int32_t i = 0; i < fr->width; i++) acc[(size_t)j * fr->width + i] = 0.0f;
}

/* ==========================================================================================
 * LAO  (LAORenderer.glsl) — SURVEY section 8f row 3.  An experimental shader of the reference: its "random" numbers are
 * rand(vPosition * seed) with a constant seed, i.e. one fixed value per pixel, and its frames do not accumulate
 * (integrate copies the frame).  frame / accumulation: RGBA8 (LAORenderer.js:217-243).
 * ======================================================================================== */
float vpo_expf(float x);                                                    /* vpt_tonemap_oracle.c */
float vpo_powf(float x, float y);
typedef struct {
    int32_t local_ambient_occlusion;    /* uLocalAmbientOcclusion */
    float lao_weight;                   /* uLAOWeight */
    int32_t num_lao_samples;            /* uNumLAOSamples */
    float lao_step_size;                /* uLAOStepSize */
    int32_t soft_shadows;               /* uSoftShadows */
    float shadows_weight;               /* uShadowsWeight */
    int32_t num_shadow_samples;         /* uNumShadowSamples */
    float light_radius;                 /* uLightRadious */
    float light_coefficient;            /* uLightCoeficient */
    float light_position[3];            /* uLightPosition */
} vpo_lao_params;
/* mixins/rand.glsl:3-13 — mat2 is column-major; the two dot products are fma chains (first column first) */
static v2 lao_rand(float px, float py) {
    const float m00 = 23.14069263277926f, m01 = 2.665144142690225f, m10 = 12.98987893203892f, m11 = 78.23376739376591f;
    float dx = fmaf(m10, py, m00 * px), dy = fmaf(m11, py, m01 * px);
    float s, c, s2, c2;
    vpo_sincosf(dx, &s, &c); vpo_sincosf(dy, &s2, &c2);
    float ax = c * 1235.6789f, ay = s2 * 4378.5453f;
    v2 r = { ax - floorf(ax), ay - floorf(ay) };
    return r;
}
static inline float sample_volume_raw(const scene_tables *t, v3 p, uint64_t *ns) { (*ns)++; return sample_volume(t->sc, p); }
/* generate/fragment main(): LAORenderer.glsl:97-191; the vertex stage's vLight (:25) = (M^-1 * (uLightPosition, 1)).xyz, no divide */
VPO_API uint64_t vpo_lao_generate(const vpo_scene *sc, const vpo_frame *fr, const vpo_lao_params *lp, uint8_t *frame) {
    scene_tables t; tables_init(&t, sc);
    uint64_t ns = 0; int nth = clamp_threads(fr);
    v4 lh = mat4_mul_point(fr->mvp_inv, lp->light_position[0], lp->light_position[1], lp->light_position[2]);
    const v3 vl = { lh.x, lh.y, lh.z };
    const float vs = 1.0f / 32.0f;                                         /* voxelSize, :59 */
    const float rs = lao_rand(3.14f, 2.71f).x;                              /* rand(seed).x */
    FOR_ROWS(fr) {
        for (int32_t i = 0; i < fr->width; i++) {
            float px = pixel_ndc(i, fr->width), py = pixel_ndc(j, fr->height);
            v3 rf, rt;
            unproject(px, py, fr->mvp_inv, &rf, &rt);
            v3 dir = sub3(rt, rf);
            v2 tb = intersect_cube(rf, dir);
            tb.x = vmax(tb.x, 0.0f); tb.y = vmax(tb.y, 0.0f);
            v4 o = { 0.0f, 0.0f, 0.0f, 1.0f };
            if (!(tb.x >= tb.y)) {
                v3 from = mix3(rf, rt, tb.x), to = mix3(rf, rt, tb.y);
                const float R = lao_rand(px * 3.14f, py * 2.71f).x;         /* every rand(vPosition * seed) of the shader */
                float tt = vclamp01((R * fr->step) * 1.5f);
                v4 acc = { 0.0f, 0.0f, 0.0f, 0.0f };
                while (tt < 1.0f && acc.w < 0.99f) {
                    if (acc.w > 0.98f) break;
                    v3 pos = mix3(from, to, tt);
                    tt += fr->step;
                    v3 gx0 = { pos.x - vs, pos.y, pos.z }, gx1 = { pos.x + vs, pos.y, pos.z };
                    v3 gy0 = { pos.x, pos.y - vs, pos.z }, gy1 = { pos.x, pos.y + vs, pos.z };
                    v3 gz0 = { pos.x, pos.y, pos.z - vs }, gz1 = { pos.x, pos.y, pos.z + vs };
                    v3 grad;
                    grad.x = sample_volume_raw(&t, gx0, &ns) - sample_volume_raw(&t, gx1, &ns);
                    grad.y = sample_volume_raw(&t, gy0, &ns) - sample_volume_raw(&t, gy1, &ns);
                    grad.z = sample_volume_raw(&t, gz0, &ns) - sample_volume_raw(&t, gz1, &ns);
                    float value = sample_volume_raw(&t, pos, &ns);
                    float lao = 0.0f, soft = 0.0f;
                    if (lp->local_ambient_occlusion) {
                        float a = 0.0f;                                     /* accumuLAOContribution: not reset between samples */
                        for (int32_t samp = 0; samp < lp->num_lao_samples; samp++) {
                            for (float u = 0.001f; u < 1.0f; u += lp->lao_step_size) {
                                float rc = -1.0f + 2.0f * R;
                                v3 rd = { rc, rc, rc };
                                rd = normalize3(rd); rd.x *= R; rd.y *= R; rd.z *= R;
                                float m = mixf(0.0f, lp->light_radius, u);
                                v3 hv = { (vl.x + rd.x * m) - pos.x, (vl.y + rd.y * m) - pos.y, (vl.z + rd.z * m) - pos.z };
                                hv = normalize3(hv);
                                v3 sp = { pos.x + hv.x * u, pos.y + hv.y * u, pos.z + hv.z * u };
                                a += sample_volume_raw(&t, sp, &ns) * vpo_powf(1.0f - u, 2.0f);
                                if (!(lp->lao_step_size > 0.0f)) break;     /* a zero step would never end; one sample then */
                            }
                            a /= lp->light_coefficient;
                            a = vclamp01(a);
                            lao += a;
                        }
                        lao /= (float)lp->num_lao_samples;
                    }
                    if (lp->soft_shadows) {
                        float a = 0.0f;
                        for (int32_t samp = 0; samp < lp->num_shadow_samples; samp++) {
                            v3 rd = { -1.0f + vl.x * R, vl.y + R * vl.z, -1.0f + 2.0f * rs };
                            rd = normalize3(rd); rd.x *= R; rd.y *= R; rd.z *= R;
                            v3 sp = { pos.x + rd.x * lp->light_radius, pos.y + rd.y * lp->light_radius, pos.z + rd.z * lp->light_radius };
                            float s1 = sample_volume_raw(&t, sp, &ns) * 0.2f;
                            a += (sample_volume_raw(&t, sp, &ns) * s1) * vpo_powf(length3(rd), 1.0f);
                        }
                        a = vpo_powf(a, 1.0f);
                        a /= (float)lp->num_shadow_samples;
                        a *= 20.0f;
                        a = vclamp01(a);
                        soft = mixf(1.0f - soft, a, 1.2f);
                    }
                    soft /= 1.3f;
                    soft = vclamp01(soft);
                    v4 c = sample_2d(t.tf, sc->tf_w, sc->tf_h, value, length3(grad));     /* getColor(value, gradientMagnitude(grad)) */
                    float w1 = lao * lp->lao_weight, w2 = soft * lp->shadows_weight;
                    c.x = mixf(c.x, c.x * 0.15f, w1); c.y = mixf(c.y, c.y * 0.18f, w1); c.z = mixf(c.z, c.z * 0.32f, w1); c.w = mixf(c.w, c.w * 1.0f, w1);
                    c.x = mixf(c.x, c.x * 0.15f, w2); c.y = mixf(c.y, c.y * 0.18f, w2); c.z = mixf(c.z, c.z * 0.22f, w2); c.w = mixf(c.w, c.w * 1.0f, w2);
                    float k = 1.0f - acc.w;
                    acc.x += (k * c.x) * value; acc.y += (k * c.y) * value; acc.z += (k * c.z) * value;
                    acc.w += ((k * value) * fr->extinction) / 100.0f;
                    if (acc.w > 0.9f) break;
                }
                if (acc.w > 1.0f) { acc.x /= acc.w; acc.y /= acc.w; acc.z /= acc.w; }
                o.x = acc.x; o.y = acc.y; o.z = acc.z;
            }
            uint8_t *pxl = frame + 4 * ((size_t)j * fr->width + i);
            pxl[0] = to_unorm8(o.x); pxl[1] = to_unorm8(o.y); pxl[2] = to_unorm8(o.z); pxl[3] = to_unorm8(o.w);
        }
    }
    tables_free(&t);
    return ns;
}
/* integrate: LAORenderer.glsl:225-227 — oColor = texture(uFrame): the frame replaces the accumulation */
VPO_API void vpo_lao_integrate(const vpo_frame *fr, uint8_t *acc, const uint8_t *frame) {
    for (int32_t j = fr->y0; j < fr->y1; j++)
        memcpy(acc + 4 * (size_t)j * fr->width, frame + 4 * (size_t)j * fr->width, 4 * (size_t)fr->width);
}
/* render: LAORenderer.glsl:259-261; reset: :285-287 — as EAM's */
VPO_API void vpo_lao_render(const vpo_frame *fr, const uint8_t *acc, uint16_t *out) { vpo_eam_render(fr, acc, out); }
VPO_API void vpo_lao_reset(const vpo_frame *fr, uint8_t *acc) { vpo_eam_reset(fr, acc); }

/* ==========================================================================================
 * DOS  (DOSRenderer.glsl) — SURVEY section 8f row 3: directional occlusion shading.  The image is swept front to back
 * one view-aligned slice per full-screen pass; every pass reads the previous pass's colour (RGBA32F, NEAREST) and
 * occlusion (R32F, LINEAR, default wrap = REPEAT: DOSRenderer.js:287-305 sets none) buffers and writes the next ones
 * (DOSRenderer.js:199-259).  Buffers here: row-major, row 0 = bottom.
 * ======================================================================================== */
/* reset: DOSRenderer.glsl:137-144 */
VPO_API void vpo_dos_reset(const vpo_frame *fr, float *color, float *occlusion) {
    size_t n = (size_t)fr->width * fr->height;
    for (size_t k = 0; k < n; k++) { color[4 * k] = 0.0f; color[4 * k + 1] = 0.0f; color[4 * k + 2] = 0.0f; color[4 * k + 3] = 0.0f; occlusion[k] = 1.0f; }
}
/* LINEAR / REPEAT tap pair.  Contract: u = s*n - 0.5; a coordinate that is NaN or beyond 1e9 texels reads texel 0. */
static inline void repeat_coord(float s, int32_t n, int32_t *i0, int32_t *i1, float *f) {
    float u = fmaf(s, (float)n, -0.5f);
    if (!(fabsf(u) < 1.0e9f)) u = 0.0f;
    float fl = floorf(u);
    *f = u - fl;
    int32_t i = (int32_t)fl % n;
    if (i < 0) i += n;
    *i0 = i; *i1 = (i + 1 == n) ? 0 : i + 1;
}
static float sample_occlusion(const float *occ, int32_t w, int32_t h, float s, float t) {
    int32_t x0, x1, y0, y1; float fx, fy;
    repeat_coord(s, w, &x0, &x1, &fx);
    repeat_coord(t, h, &y0, &y1, &fy);
    float a = lerpf(occ[(size_t)y0 * w + x0], occ[(size_t)y0 * w + x1], fx);
    float b = lerpf(occ[(size_t)y1 * w + x0], occ[(size_t)y1 * w + x1], fx);
    return lerpf(a, b, fy);
}
/* one slice = one draw of integrate/fragment main(): DOSRenderer.glsl:73-89 (vertex :17-23: vPosition3D = unprojected
 * (position, uDepth), evaluated at the pixel's own NDC).  uSliceDistance = fr->step, uExtinction = fr->extinction,
 * slice[3] = (uOcclusionScale.x, uOcclusionScale.y, uDepth), samples = the RG32F texel row of DOSRenderer.js:103-140.
 * Returns the number of volume samples. */
VPO_API uint64_t vpo_dos_integrate_slice(const vpo_scene *sc, const vpo_frame *fr, const float *slice, const float *samples, int32_t nsamples,
                                         const float *color_in, const float *occ_in, float *color_out, float *occ_out) {
    scene_tables t; tables_init(&t, sc);
    uint64_t ns = 0; int nth = clamp_threads(fr);
    const int32_t W = fr->width, H = fr->height;
    const float sd = fr->step;
    FOR_ROWS(fr) {
        for (int32_t i = 0; i < W; i++) {
            size_t k = (size_t)j * W + i;
            float px = pixel_ndc(i, W), py = pixel_ndc(j, H);
            float uvx = ndc_to_uv(px), uvy = ndc_to_uv(py);
            const float *prev = color_in + 4 * k;
            float prev_occ = occ_in[k];                              /* texture() at the texel's own centre */
            v4 hp = mat4_mul_point(fr->mvp_inv, px, py, slice[2]);
            float rw = vpo_rcp_nr(hp.w);
            v3 pos = { hp.x * rw, hp.y * rw, hp.z * rw };
            float *oc = color_out + 4 * k;
            if (pos.x > 1.0f || pos.y > 1.0f || pos.z > 1.0f || pos.x < 0.0f || pos.y < 0.0f || pos.z < 0.0f) {
                oc[0] = prev[0]; oc[1] = prev[1]; oc[2] = prev[2]; oc[3] = prev[3];
                occ_out[k] = prev_occ;
            } else {
                v4 ts = sample_volume_color(&t, pos, &ns);
                float ext = ts.w * fr->extinction;
                float e = vpo_expf((-ext) * sd);
                float alpha = 1.0f - e;
                float k1 = 1.0f - prev[3];
                oc[0] = prev[0] + ((ts.x * prev_occ) * alpha) * k1;
                oc[1] = prev[1] + ((ts.y * prev_occ) * alpha) * k1;
                oc[2] = prev[2] + ((ts.z * prev_occ) * alpha) * k1;
                oc[3] = vmin(prev[3] + alpha, 1.0f);
                float o = 0.0f;                                       /* calculateOcclusion: :61-70 */
                for (int32_t q = 0; q < nsamples; q++)
                    o += sample_occlusion(occ_in, W, H, uvx + samples[2 * q] * slice[0], uvy + samples[2 * q + 1] * slice[1]);
                occ_out[k] = (o / (float)nsamples) * e;
            }
        }
    }
    tables_free(&t);
    return ns;
}
/* render: DOSRenderer.glsl:113-116 — mix(vec4(1), vec4(color.rgb, 1), color.a) */
VPO_API void vpo_dos_render(const vpo_frame *fr, const float *color, uint16_t *out) {
    for (int32_t j = fr->y0; j < fr->y1; j++)
        for (int32_t i = 0; i < fr->width; i++) {
            size_t k = (size_t)j * fr->width + i;
            const float *c = color + 4 * k;
            store_half4(out + 4 * k, mixf(1.0f, c[0], c[3]), mixf(1.0f, c[1], c[3]), mixf(1.0f, c[2], c[3]), mixf(1.0f, 1.0f, c[3]));
        }
}

/* ==========================================================================================
 * MCS  (MCSRenderer.glsl)
 * ======================================================================================== */
#define VPO_MAX_TRACK_ITERS 65536u   /* safety net shared with the GPU kernels */

/* sampleDistance: MCSRenderer.glsl:70-87.  "dist > maxDistance" is written !(dist <= max) so a NaN
 * segment terminates. */
static float mcs_sample_distance(const scene_tables *t, uint32_t *state, v3 from, v3 to, float inv_ext, uint64_t *ns) {
    float max_distance = length3(sub3(from, to));
    float inv_max = vpo_rcp_nr(max_distance);         /* dist / maxDistance as dist * rcp(maxDistance) */
    float dist = 0.0f;
    for (uint32_t it = 0; it < VPO_MAX_TRACK_ITERS; it++) {
        dist += random_exponential(state, inv_ext);
        if (!(dist <= max_distance)) break;
        v3 p = mix3(from, to, dist * inv_max);
        v4 ts = sample_volume_color(t, p, ns);
        if (vpo_random_uniform(state) < ts.w) break;
    }
    return dist;
}
/* sampleTransmittance: MCSRenderer.glsl:89-105 */
static float mcs_sample_transmittance(const scene_tables *t, uint32_t *state, v3 from, v3 to, float inv_ext, uint64_t *ns) {
    float max_distance = length3(sub3(from, to));
    float inv_max = vpo_rcp_nr(max_distance);
    float dist = 0.0f, tr = 1.0f;
    for (uint32_t it = 0; it < VPO_MAX_TRACK_ITERS; it++) {
        dist += random_exponential(state, inv_ext);
        if (!(dist <= max_distance)) break;
        v3 p = mix3(from, to, dist * inv_max);
        v4 ts = sample_volume_color(t, p, ns);
        tr *= 1.0f - ts.w;
    }
    return tr;
}
/* generate/fragment main(): MCSRenderer.glsl:107-137.  frame: RGBA32F. */
VPO_API uint64_t vpo_mcs_generate(const vpo_scene *sc, const vpo_frame *fr, float *frame) {
    scene_tables t; tables_init(&t, sc);
    uint64_t ns = 0; int nth = clamp_threads(fr);
    v3 L = { fr->light_dir[0], fr->light_dir[1], fr->light_dir[2] };
    const float inv_ext = 1.0f / fr->extinction;
    FOR_ROWS(fr) {
        for (int32_t i = 0; i < fr->width; i++) {
            float px = pixel_ndc(i, fr->width), py = pixel_ndc(j, fr->height);
            v3 rf, rt;
            unproject(px, py, fr->mvp_inv, &rf, &rt);
            v3 dir = sub3(rt, rf);
            v3 dir_unit = normalize3(dir);
            v2 tb = intersect_cube(rf, dir);
            tb.x = vmax(tb.x, 0.0f); tb.y = vmax(tb.y, 0.0f);
            v4 o;
            if (tb.x >= tb.y) {
                o = sample_environment(&t, dir_unit);
            } else {
                v3 from = mix3(rf, rt, tb.x), to = mix3(rf, rt, tb.y);
                float max_distance = length3(sub3(from, to));
                uint32_t state = vpo_hash3(f2u(ndc_to_uv(px)), f2u(ndc_to_uv(py)), f2u(fr->seed));
                float dist = mcs_sample_distance(&t, &state, from, to, inv_ext, &ns);
                if (!(dist <= max_distance)) {
                    o = sample_environment(&t, dir_unit);
                } else {
                    from = mix3(from, to, dist * vpo_rcp_nr(max_distance));
                    v2 tb2 = intersect_cube(from, L);
                    tb2.y = vmax(tb2.y, 0.0f);
                    to = madd3(from, tb2.y, L);
                    v4 diffuse = sample_volume_color(&t, from, &ns);
                    v4 light = sample_environment(&t, L);
                    float tr = mcs_sample_transmittance(&t, &state, from, to, inv_ext, &ns);
                    o.x = (diffuse.x * light.x) * tr; o.y = (diffuse.y * light.y) * tr;
                    o.z = (diffuse.z * light.z) * tr; o.w = (diffuse.w * light.w) * tr;
                }
            }
            float *q = frame + 4 * ((size_t)j * fr->width + i);
            q[0] = o.x; q[1] = o.y; q[2] = o.z; q[3] = o.w;
        }
    }
    tables_free(&t);
    return ns;
}
/* integrate: MCSRenderer.glsl:173-177 — acc + (frame - acc) * uInvFrameNumber */
VPO_API void vpo_mcs_integrate(const vpo_frame *fr, float *acc, const float *frame) {
    for (int32_t j = fr->y0; j < fr->y1; j++)
        for (int32_t i = 0; i < 4 * fr->width; i++) {
            size_t k = (size_t)j * 4 * fr->width + i;
            acc[k] = fmaf(frame[k] - acc[k], fr->mix, acc[k]);
        }
}
/* render: MCSRenderer.glsl:210-213 */
VPO_API void vpo_mcs_render(const vpo_frame *fr, const float *acc, uint16_t *out) {
    for (int32_t j = fr->y0; j < fr->y1; j++)
        for (int32_t i = 0; i < 4 * fr->width; i++) {
            size_t k = (size_t)j * 4 * fr->width + i;
            out[k] = vpo_f32_to_f16(acc[k]);
        }
}
/* reset: MCSRenderer.glsl:238-240 */
VPO_API void vpo_mcs_reset(const vpo_frame *fr, float *acc) {
    for (int32_t j = fr->y0; j < fr->y1; j++)
        for (int32_t i = 0; i < fr->width; i++) {
            float *q = acc + 4 * ((size_t)j * fr->width + i);
            q[0] = 0.0f; q[1] = 0.0f; q[2] = 0.0f; q[3] = 1.0f;
        }
}

/* ==========================================================================================
 * MCM  (MCMRenderer.glsl, mixins/Photon.glsl, mixins/unprojectRand.glsl)
 * state: 4 RGBA32F planes [pos.xyz,0] [dir.xyz,bounces] [transmittance.rgb,0] [radiance.rgb,samples]
 * (MCMRenderer.js:214-263), each row-major width*height*4 floats.
 * ======================================================================================== */
typedef struct {
    v3 position, direction, transmittance, radiance;
    uint32_t bounces, samples;
} photon;

/* mixins/unprojectRand.glsl:3-24 */
static void unproject_rand(uint32_t *state, float px, float py, const vpo_frame *fr, v3 *from, v3 *to) {
    /* random_disk(state) * blur: with blur == 0 the two uniforms are still drawn, and the product is an exact zero */
    float ox = 0.0f, oy = 0.0f;
    if (fr->blur == 0.0f) { vpo_random_uniform(state); vpo_random_uniform(state); }
    else { v2 d = random_disk(state); ox = d.x * fr->blur; oy = d.y * fr->blur; }
    v2 sq = random_square(state);
    float ax = fmaf(sq.x, 2.0f, -1.0f) * fr->inv_res[0];
    float ay = fmaf(sq.y, 2.0f, -1.0f) * fr->inv_res[1];
    v4 n = mat4_mul_point(fr->mvp_inv, px + ox, py + oy, -1.0f);
    v4 f = mat4_mul_point(fr->mvp_inv, px + ax, py + ay, 1.0f);
    float in = vpo_rcp_nr(n.w), jf = vpo_rcp_nr(f.w);
    from->x = n.x * in; from->y = n.y * in; from->z = n.z * in;
    to->x = f.x * jf; to->y = f.y * jf; to->z = f.z * jf;
}
/* resetPhoton: MCMRenderer.glsl:70-78 */
static void reset_photon(uint32_t *state, photon *ph, float px, float py, const vpo_frame *fr) {
    v3 from, to;
    unproject_rand(state, px, py, fr, &from, &to);
    ph->direction = normalize3(sub3(to, from));
    ph->bounces = 0u;
    v2 tb = intersect_cube(from, ph->direction);
    tb.x = vmax(tb.x, 0.0f);
    ph->position = madd3(from, tb.x, ph->direction);
    ph->transmittance.x = 1.0f; ph->transmittance.y = 1.0f; ph->transmittance.z = 1.0f;
}
/* sampleHenyeyGreensteinAngleCosine: MCMRenderer.glsl:91-95 */
static float hg_cos(uint32_t *state, float g) {
    float g2 = g * g;
    float c = (1.0f - g2) * vpo_rcp_nr(fmaf(2.0f * g, vpo_random_uniform(state), 1.0f - g));
    return fmaf(-c, c, 1.0f + g2) * vpo_rcp_nr(2.0f * g);
}
/* sampleHenyeyGreenstein: MCMRenderer.glsl:97-106 (EPS 1e-5) */
static v3 sample_hg(uint32_t *state, float g, v3 dir) {
    v3 u = random_sphere(state);
    if (fabsf(g) < 1e-5f) return u;
    float hgcos = hg_cos(state, g);
    float ud = dot3(u, dir);
    v3 c = { fmaf(-ud, dir.x, u.x), fmaf(-ud, dir.y, u.y), fmaf(-ud, dir.z, u.z) };
    c = normalize3(c);
    float s = vpo_sqrt_nr(fmaf(-hgcos, hgcos, 1.0f));
    v3 r = { fmaf(s, c.x, hgcos * dir.x), fmaf(s, c.y, hgcos * dir.y), fmaf(s, c.z, hgcos * dir.z) };
    return r;
}
static inline float max3f(v3 v) { return vmax(vmax(v.x, v.y), v.z); }

static inline void photon_load(photon *ph, const float *const st[4], size_t k) {
    ph->position.x = st[0][4 * k]; ph->position.y = st[0][4 * k + 1]; ph->position.z = st[0][4 * k + 2];
    ph->direction.x = st[1][4 * k]; ph->direction.y = st[1][4 * k + 1]; ph->direction.z = st[1][4 * k + 2];
    ph->bounces = (uint32_t)(st[1][4 * k + 3] + 0.5f);
    ph->transmittance.x = st[2][4 * k]; ph->transmittance.y = st[2][4 * k + 1]; ph->transmittance.z = st[2][4 * k + 2];
    ph->radiance.x = st[3][4 * k]; ph->radiance.y = st[3][4 * k + 1]; ph->radiance.z = st[3][4 * k + 2];
    ph->samples = (uint32_t)(st[3][4 * k + 3] + 0.5f);
}
static inline void photon_store(const photon *ph, float *const st[4], size_t k) {
    st[0][4 * k] = ph->position.x; st[0][4 * k + 1] = ph->position.y; st[0][4 * k + 2] = ph->position.z; st[0][4 * k + 3] = 0.0f;
    st[1][4 * k] = ph->direction.x; st[1][4 * k + 1] = ph->direction.y; st[1][4 * k + 2] = ph->direction.z; st[1][4 * k + 3] = (float)ph->bounces;
    st[2][4 * k] = ph->transmittance.x; st[2][4 * k + 1] = ph->transmittance.y; st[2][4 * k + 2] = ph->transmittance.z; st[2][4 * k + 3] = 0.0f;
    st[3][4 * k] = ph->radiance.x; st[3][4 * k + 1] = ph->radiance.y; st[3][4 * k + 2] = ph->radiance.z; st[3][4 * k + 3] = (float)ph->samples;
}

/* reset/fragment main(): MCMRenderer.glsl:259-275 — seeded from NDC vPosition */
VPO_API void vpo_mcm_reset(const vpo_frame *fr, float *s0, float *s1, float *s2, float *s3) {
    float *const st[4] = { s0, s1, s2, s3 };
    for (int32_t j = fr->y0; j < fr->y1; j++)
        for (int32_t i = 0; i < fr->width; i++) {
            float px = pixel_ndc(i, fr->width), py = pixel_ndc(j, fr->height);
            photon ph;
            uint32_t state = vpo_hash3(f2u(px), f2u(py), f2u(fr->seed));
            v3 from, to;
            unproject_rand(&state, px, py, fr, &from, &to);
            ph.direction = normalize3(sub3(to, from));
            v2 tb = intersect_cube(from, ph.direction);
            tb.x = vmax(tb.x, 0.0f);
            ph.position = madd3(from, tb.x, ph.direction);
            ph.transmittance.x = ph.transmittance.y = ph.transmittance.z = 1.0f;
            ph.radiance.x = ph.radiance.y = ph.radiance.z = 1.0f;
            ph.bounces = 0u; ph.samples = 0u;
            photon_store(&ph, st, (size_t)j * fr->width + i);
        }
}

/* integrate/fragment main(): MCMRenderer.glsl:116-172 — seeded from the [0,1]-mapped position */
VPO_API uint64_t vpo_mcm_integrate(const vpo_scene *sc, const vpo_frame *fr, float *s0, float *s1, float *s2, float *s3) {
    scene_tables t; tables_init(&t, sc);
    float *const st[4] = { s0, s1, s2, s3 };
    const float *const cst[4] = { s0, s1, s2, s3 };
    uint64_t ns = 0; int nth = clamp_threads(fr);
    const float inv_ext = 1.0f / fr->extinction;
    FOR_ROWS(fr) {
        for (int32_t i = 0; i < fr->width; i++) {
            size_t k = (size_t)j * fr->width + i;
            float px = pixel_ndc(i, fr->width), py = pixel_ndc(j, fr->height);
            photon ph;
            photon_load(&ph, cst, k);
            uint32_t state = vpo_hash3(f2u(ndc_to_uv(px)), f2u(ndc_to_uv(py)), f2u(fr->seed));
            for (uint32_t s = 0u; s < fr->steps; s++) {
                float dist = random_exponential(&state, inv_ext);
                ph.position = madd3(ph.position, dist, ph.direction);
                v4 vs = sample_volume_color(&t, ph.position, &ns);
                float p_null = 1.0f - vs.w;
                float p_scat;
                if (ph.bounces >= fr->max_bounces) p_scat = 0.0f;
                else { v3 c = { vs.x, vs.y, vs.z }; p_scat = vs.w * max3f(c); }
                float p_abs = 1.0f - p_null - p_scat;
                float wheel = vpo_random_uniform(&state);
                v3 p = ph.position;
                int oob = (p.x > 1.0f || p.y > 1.0f || p.z > 1.0f || p.x < 0.0f || p.y < 0.0f || p.z < 0.0f);
                if (oob || wheel < p_abs) {
                    /* out of bounds: radiance = transmittance * env ; absorption: radiance = 0 (MCMRenderer.glsl:143-157) */
                    v3 rad = { 0.0f, 0.0f, 0.0f };
                    if (oob) {
                        v4 env = sample_environment(&t, ph.direction);
                        rad.x = ph.transmittance.x * env.x; rad.y = ph.transmittance.y * env.y; rad.z = ph.transmittance.z * env.z;
                    }
                    ph.samples++;
                    float inv_n = vpo_rcp_nr((float)ph.samples);    /* (rad - radiance) / n as * rcp(n) */
                    ph.radiance.x += (rad.x - ph.radiance.x) * inv_n;
                    ph.radiance.y += (rad.y - ph.radiance.y) * inv_n;
                    ph.radiance.z += (rad.z - ph.radiance.z) * inv_n;
                    reset_photon(&state, &ph, px, py, fr);
                } else if (wheel < p_abs + p_scat) {
                    ph.transmittance.x *= vs.x; ph.transmittance.y *= vs.y; ph.transmittance.z *= vs.z;
                    ph.direction = sample_hg(&state, fr->anisotropy, ph.direction);
                    ph.bounces++;
                }
            }
            photon_store(&ph, st, k);
        }
    }
    tables_free(&t);
    return ns;
}
/* render: MCMRenderer.glsl:204-206 */
VPO_API void vpo_mcm_render(const vpo_frame *fr, const float *s3, uint16_t *out) {
    for (int32_t j = fr->y0; j < fr->y1; j++)
        for (int32_t i = 0; i < fr->width; i++) {
            size_t k = (size_t)j * fr->width + i;
            out[4 * k + 0] = vpo_f32_to_f16(s3[4 * k + 0]);
            out[4 * k + 1] = vpo_f32_to_f16(s3[4 * k + 1]);
            out[4 * k + 2] = vpo_f32_to_f16(s3[4 * k + 2]);
            out[4 * k + 3] = 0x3c00u;
        }
}

/* probes used by tests */
VPO_API float vpo_sample_volume(const vpo_scene *sc, float x, float y, float z) {
    v3 p = { x, y, z }; return sample_volume(sc, p);
}
VPO_API void vpo_sample_volume_color(const vpo_scene *sc, float x, float y, float z, float *out4) {
    scene_tables t; tables_init(&t, sc);
    uint64_t ns = 0; v3 p = { x, y, z };
    v4 c = sample_volume_color(&t, p, &ns);
    out4[0] = c.x; out4[1] = c.y; out4[2] = c.z; out4[3] = c.w;
    tables_free(&t);
}
VPO_API void vpo_sample_environment(const vpo_scene *sc, float x, float y, float z, float *out4) {
    scene_tables t; tables_init(&t, sc);
    v3 d = { x, y, z };
    v4 c = sample_environment(&t, d);
    out4[0] = c.x; out4[1] = c.y; out4[2] = c.z; out4[3] = c.w;
    tables_free(&t);
}
VPO_API void vpo_unproject(const float *m, float px, float py, float *from3, float *to3) {
    v3 f, t; unproject(px, py, m, &f, &t);
    from3[0] = f.x; from3[1] = f.y; from3[2] = f.z; to3[0] = t.x; to3[1] = t.y; to3[2] = t.z;
}
VPO_API void vpo_intersect_cube(const float *o3, const float *d3, float *out2) {
    v3 o = { o3[0], o3[1], o3[2] }, d = { d3[0], d3[1], d3[2] };
    v2 r = intersect_cube(o, d); out2[0] = r.x; out2[1] = r.y;
}
VPO_API float vpo_min(float a, float b) { return vmin(a, b); }
VPO_API float vpo_max(float a, float b) { return vmax(a, b); }
VPO_API int vpo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
