// vpt_device.h — device-side building blocks of the renderer kernels (gfx950).
//
// Numeric contract: IEEE binary32, round-to-nearest-even, contraction OFF
// (-ffp-contract=off) — every fused multiply-add is an explicit fmaf(); correctly rounded
// division and sqrt (-fhip-fp32-correctly-rounded-divide-sqrt); denormals kept.
// log / sin / cos / atan2 / asin are the polynomial routines below, not OCML, so results do not
// depend on a vendor math library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VPT_DEV __device__ __forceinline__

struct f3 { float x, y, z; };
struct f2 { float x, y; };

// ---- GLSL built-ins ------------------------------------------------------------------------
VPT_DEV float vmin(float a, float b) { return (b < a) ? b : a; }   // min(x,y) = y<x ? y : x
VPT_DEV float vmax(float a, float b) { return (a < b) ? b : a; }   // max(x,y) = x<y ? y : x
VPT_DEV float vclamp01(float x) { return vmin(vmax(x, 0.0f), 1.0f); }
VPT_DEV float mixf(float a, float b, float t) { return fmaf(b, t, a * (1.0f - t)); }  // x*(1-a)+y*a
VPT_DEV f3 mix3(f3 a, f3 b, float t) { return f3{ mixf(a.x, b.x, t), mixf(a.y, b.y, t), mixf(a.z, b.z, t) }; }
VPT_DEV float dot3(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
VPT_DEV float length3(f3 a) { return sqrtf(dot3(a, a)); }
VPT_DEV f3 sub3(f3 a, f3 b) { return f3{ a.x - b.x, a.y - b.y, a.z - b.z }; }
VPT_DEV f3 normalize3(f3 a) { float inv = 1.0f / length3(a); return f3{ a.x * inv, a.y * inv, a.z * inv }; }
VPT_DEV f3 madd3(f3 p, float t, f3 d) { return f3{ fmaf(t, d.x, p.x), fmaf(t, d.y, p.y), fmaf(t, d.z, p.z) }; }
VPT_DEV float lerpf(float a, float b, float f) { return fmaf(f, b - a, a); }
VPT_DEV float4 lerp4(float4 a, float4 b, float f) {
    return make_float4(lerpf(a.x, b.x, f), lerpf(a.y, b.y, f), lerpf(a.z, b.z, f), lerpf(a.w, b.w, f));
}

// ---- transcendental routines ---------------------------------------------------------------
// natural log on {0} U [2^-126, inf]
VPT_DEV float vpt_logf(float x) {
    uint32_t b = __float_as_uint(x);
    int32_t e = (int32_t)(b >> 23) - 126;
    float m = __uint_as_float((b & 0x007fffffu) | 0x3f000000u);
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
    if (x == 0.0f) r = -__builtin_inff();
    if (!(x >= 0.0f)) r = __builtin_nanf("");
    if (x == __builtin_inff()) r = __builtin_inff();
    return r;
}

// sin / cos of a >= 0 (used on [0, 2*pi])
VPT_DEV void vpt_sincosf(float a, float &s, float &c) {
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
    float s0 = (n & 1) ? cr : sr;
    float c0 = (n & 1) ? sr : cr;
    s = (n & 2) ? -s0 : s0;
    c = ((n + 1) & 2) ? -c0 : c0;
}

VPT_DEV float vpt_atan_pos(float t) {
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
VPT_DEV float vpt_atan2f(float y, float x) {
    if (x != x || y != y) return __builtin_nanf("");
    float ay = fabsf(y), ax = fabsf(x);
    float r;
    if (ax == 0.0f && ay == 0.0f) r = 0.0f;
    else if (ax == __builtin_inff() && ay == __builtin_inff()) r = 0.7853981633974483f;
    else r = vpt_atan_pos(ay / ax);
    if (__float_as_uint(x) >> 31) r = 3.14159265358979323846f - r;
    return (__float_as_uint(y) >> 31) ? -r : r;
}
VPT_DEV float vpt_asinf(float x) {
    float a = fabsf(x);
    if (!(a <= 1.0f)) return __builtin_nanf("");
    float z, t;
    bool big = a > 0.5f;
    if (big) { z = 0.5f * (1.0f - a); t = sqrtf(z); } else { t = a; z = a * a; }
    float p = 4.2163199048E-2f;
    p = fmaf(p, z, 2.4181311049E-2f);
    p = fmaf(p, z, 4.5470025998E-2f);
    p = fmaf(p, z, 7.4953002686E-2f);
    p = fmaf(p, z, 1.6666752422E-1f);
    float r = fmaf(p * z, t, t);
    if (big) r = 1.5707963267948966f - (r + r);
    return (__float_as_uint(x) >> 31) ? -r : r;
}

// ---- RNG: mixins/random/hash/pcg.glsl:3-7, squashlinear.glsl:7-9, distribution/*.glsl ---------
VPT_DEV uint32_t pcg(uint32_t x) {
    x = x * 747796405u + 2891336453u;
    x = ((x >> ((x >> 28u) + 4u)) ^ x) * 277803737u;
    return (x >> 22u) ^ x;
}
VPT_DEV uint32_t hash3(uint32_t x, uint32_t y, uint32_t z) { return pcg(19u * x + 47u * y + 101u * z + 131u); }
// float(state)/float(~0u): the divisor rounds to 2^32, the quotient is exact
VPT_DEV float random_uniform(uint32_t &state) { state = pcg(state); return (float)state * 0x1p-32f; }
VPT_DEV float random_exponential(uint32_t &state, float rate) { return -vpt_logf(random_uniform(state)) / rate; }
VPT_DEV f2 random_disk(uint32_t &state) {
    float radius = sqrtf(random_uniform(state));
    float angle = 6.28318530718f * random_uniform(state);
    float s, c; vpt_sincosf(angle, s, c);
    return f2{ radius * c, radius * s };
}
VPT_DEV f3 random_sphere(uint32_t &state) {
    f2 d = random_disk(state);
    float norm = fmaf(d.y, d.y, d.x * d.x);
    float radius = 2.0f * sqrtf(1.0f - norm);
    float z = fmaf(-2.0f, norm, 1.0f);
    return f3{ radius * d.x, radius * d.y, z };
}

// ---- render-target conversions --------------------------------------------------------------
VPT_DEV uint32_t to_unorm8(float f) {
    float c = vclamp01(f);
    if (c != c) c = 0.0f;
    return (uint32_t)rintf(c * 255.0f);
}
VPT_DEV float from_unorm8(uint32_t c) { return (float)c / 255.0f; }
VPT_DEV uint16_t to_half_bits(float f) { return __half_as_ushort(__float2half_rn(f)); }

// ---- ray set-up -----------------------------------------------------------------------------
struct Mat4 { float m[16]; };
VPT_DEV float4 mat4_mul_point(const Mat4 &M, float x, float y, float z) {
    const float *m = M.m;
    float4 r;
    r.x = fmaf(m[12], 1.0f, fmaf(m[8],  z, fmaf(m[4], y, m[0] * x)));
    r.y = fmaf(m[13], 1.0f, fmaf(m[9],  z, fmaf(m[5], y, m[1] * x)));
    r.z = fmaf(m[14], 1.0f, fmaf(m[10], z, fmaf(m[6], y, m[2] * x)));
    r.w = fmaf(m[15], 1.0f, fmaf(m[11], z, fmaf(m[7], y, m[3] * x)));
    return r;
}
// mixins/unproject.glsl:3-10
VPT_DEV void unproject(float px, float py, const Mat4 &M, f3 &from, f3 &to) {
    float4 n = mat4_mul_point(M, px, py, -1.0f);
    float4 f = mat4_mul_point(M, px, py, 1.0f);
    from = f3{ n.x / n.w, n.y / n.w, n.z / n.w };
    to = f3{ f.x / f.w, f.y / f.w, f.z / f.w };
}
VPT_DEV float pixel_ndc(int i, int n) { return (float)(2 * i + 1) / (float)n - 1.0f; }
VPT_DEV float ndc_to_uv(float p) { return fmaf(p, 0.5f, 0.5f); }

// mixins/intersectCube.glsl:3-11
VPT_DEV f2 intersect_cube(f3 o, f3 d) {
    f3 tmin = { (0.0f - o.x) / d.x, (0.0f - o.y) / d.y, (0.0f - o.z) / d.z };
    f3 tmax = { (1.0f - o.x) / d.x, (1.0f - o.y) / d.y, (1.0f - o.z) / d.z };
    f3 t1 = { vmin(tmin.x, tmax.x), vmin(tmin.y, tmax.y), vmin(tmin.z, tmax.z) };
    f3 t2 = { vmax(tmin.x, tmax.x), vmax(tmin.y, tmax.y), vmax(tmin.z, tmax.z) };
    return f2{ vmax(vmax(t1.x, t1.y), t1.z), vmin(vmin(t2.x, t2.y), t2.z) };
}

// ---- bricked Z-order volume -------------------------------------------------------------------
// Layout: 4^3-voxel bricks stored with a +1 apron = 5^3 = 125 bytes in a 128-byte
// slot (one L2 line holds every tap of a trilinear sample); brick (bx,by,bz) sits at slot
// morton3(bx,by,bz).  Inside a slot the byte of local voxel (lx,ly,lz) in [0,5)^3 is lz*25+ly*5+lx.
#define VPT_BRICK        4
#define VPT_BRICK_SHIFT  2
#define VPT_BRICK_BYTES  128

struct DevVolume {
    const uint8_t *bricks;
    int nx, ny, nz;
    float fnx, fny, fnz;
    int filter;            // VPT_FILTER_*
};

VPT_DEV uint32_t spread3(uint32_t x) {   // 10 bits -> every third bit
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8))  & 0x0300F00Fu;
    x = (x | (x << 4))  & 0x030C30C3u;
    x = (x | (x << 2))  & 0x09249249u;
    return x;
}
VPT_DEV uint32_t morton3(uint32_t x, uint32_t y, uint32_t z) {
    return spread3(x) | (spread3(y) << 1) | (spread3(z) << 2);
}
VPT_DEV const uint8_t *brick_addr(const DevVolume &v, int x, int y, int z) {
    uint32_t slot = morton3((uint32_t)x >> VPT_BRICK_SHIFT, (uint32_t)y >> VPT_BRICK_SHIFT, (uint32_t)z >> VPT_BRICK_SHIFT);
    uint32_t off = (uint32_t)(z & 3) * 25u + (uint32_t)(y & 3) * 5u + (uint32_t)(x & 3);
    return v.bricks + ((size_t)slot << 7) + off;
}
// LINEAR coordinate: u = s*N - 0.5 clamped to [-1, N]; cell index remapped so that both taps live in
// one apron brick: i = -1 -> (0, f = 0); i = N -> N-1 (both taps equal the edge voxel there).
VPT_DEV void linear_cell(float s, float fn, int n, int &i, float &f) {
    float u = fmaf(s, fn, -0.5f);
    if (!(u > -1.0f)) u = -1.0f;
    if (u > fn) u = fn;
    float fl = floorf(u);
    f = u - fl;
    i = (int)fl;
    if (i < 0) { i = 0; f = 0.0f; }
    if (i > n - 1) i = n - 1;
}
VPT_DEV int nearest_cell(float s, float fn, int n) {
    float u = s * fn;
    if (!(u > 0.0f)) u = 0.0f;
    float hi = (float)(n - 1);
    if (u > hi) u = hi;
    return (int)floorf(u);
}
// texture(uVolume, p).r for an R8 volume (Volume.js:49-60): integer texel values interpolated x, y, z,
// normalised once by /255.
VPT_DEV float sample_volume(const DevVolume &v, f3 p) {
    if (v.filter == 0) {
        int x = nearest_cell(p.x, v.fnx, v.nx), y = nearest_cell(p.y, v.fny, v.ny), z = nearest_cell(p.z, v.fnz, v.nz);
        return (float)(*brick_addr(v, x, y, z)) / 255.0f;
    }
    int x, y, z; float fx, fy, fz;
    linear_cell(p.x, v.fnx, v.nx, x, fx);
    linear_cell(p.y, v.fny, v.ny, y, fy);
    linear_cell(p.z, v.fnz, v.nz, z, fz);
    const uint8_t *a = brick_addr(v, x, y, z);
    // taps: +0,+1 (y,z) ; +5,+6 (y+1,z) ; +25,+26 (y,z+1) ; +30,+31 (y+1,z+1): two 8-byte windows
    uint64_t w0, w1;
    __builtin_memcpy(&w0, a, 8);
    __builtin_memcpy(&w1, a + 25, 8);
    uint32_t l0 = (uint32_t)w0, h0 = (uint32_t)(w0 >> 32), l1 = (uint32_t)w1, h1 = (uint32_t)(w1 >> 32);
    float c000 = (float)(l0 & 0xffu), c100 = (float)((l0 >> 8) & 0xffu);
    float c010 = (float)((h0 >> 8) & 0xffu), c110 = (float)((h0 >> 16) & 0xffu);
    float c001 = (float)(l1 & 0xffu), c101 = (float)((l1 >> 8) & 0xffu);
    float c011 = (float)((h1 >> 8) & 0xffu), c111 = (float)((h1 >> 16) & 0xffu);
    float c00 = lerpf(c000, c100, fx), c10 = lerpf(c010, c110, fx);
    float c01 = lerpf(c001, c101, fx), c11 = lerpf(c011, c111, fx);
    float c0 = lerpf(c00, c10, fy), c1 = lerpf(c01, c11, fy);
    return lerpf(c0, c1, fz) / 255.0f;
}

// transfer function: row 0 of the decoded SRGB8_ALPHA8 table, LINEAR / CLAMP_TO_EDGE, staged in LDS.
// (R8 volume => lookup at (r, 0): both bilinear rows clamp to row 0, so row 0 alone is exact.)
VPT_DEV float4 sample_tf(const float4 *tf, int tf_w, float tf_fw, float r) {
    float u = fmaf(r, tf_fw, -0.5f);
    if (!(u > -1.0f)) u = -1.0f;
    if (u > tf_fw) u = tf_fw;
    float fl = floorf(u);
    float f = u - fl;
    int i = (int)fl;
    int i0 = max(i, 0), i1 = min(i + 1, tf_w - 1);
    i0 = min(i0, tf_w - 1);
    return lerp4(tf[i0], tf[i1], f);
}

// environment map (RGBA8 decoded to float4 in HBM): MCSRenderer.glsl:59-62 / MCMRenderer.glsl:80-83
struct DevEnv {
    const float4 *texels;
    int w, h;
    float4 constant;       // the texel of a 1x1 map
};
VPT_DEV void linear_taps(float s, int n, int &i0, int &i1, float &f) {
    float fn = (float)n;
    float u = fmaf(s, fn, -0.5f);
    if (!(u > -1.0f)) u = -1.0f;
    if (u > fn) u = fn;
    float fl = floorf(u);
    f = u - fl;
    int i = (int)fl;
    i0 = min(max(i, 0), n - 1);
    i1 = min(max(i + 1, 0), n - 1);
}
VPT_DEV float4 sample_environment(const DevEnv &e, f3 d) {
    if (e.w == 1 && e.h == 1) return e.constant;
    float a = vpt_atan2f(d.x, -d.z);
    float b = vpt_asinf(-d.y) * 2.0f;
    float s = fmaf(a * 0.31830988618f, 0.5f, 0.5f);
    float t = fmaf(b * 0.31830988618f, 0.5f, 0.5f);
    int x0, x1, y0, y1; float fx, fy;
    linear_taps(s, e.w, x0, x1, fx);
    linear_taps(t, e.h, y0, y1, fy);
    float4 r0 = lerp4(e.texels[(size_t)y0 * e.w + x0], e.texels[(size_t)y0 * e.w + x1], fx);
    float4 r1 = lerp4(e.texels[(size_t)y1 * e.w + x0], e.texels[(size_t)y1 * e.w + x1], fx);
    return lerp4(r0, r1, fy);
}
