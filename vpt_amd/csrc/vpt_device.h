// vpt_device.h — device-side building blocks of the renderer kernels (gfx950).
//
// Numeric contract: IEEE binary32, round-to-nearest-even, contraction OFF
// (-ffp-contract=off) — every fused multiply-add is an explicit fmaf(); IEEE division and sqrt where
// `/` and sqrtf() are written (-fhip-fp32-correctly-rounded-divide-sqrt); denormals kept.
// log / sin / cos / atan2 / asin / rcp_nr / rsqrt_nr are the routines below (plain +,*,fma and bit
// operations), not OCML or the hardware approximations, so results do not depend on a vendor library
// and the CPU oracle reproduces them bit for bit.
//
// Cost model these routines are shaped for (tools/valu_rates.hip, measured on MI355X, >= 4 waves/SIMD):
// v_fma/v_mul/v_add_f32 ~2.6 cycles per wave64 instruction, every other VALU op (integer, compare,
// select, convert, min/max) ~4, transcendental (v_rcp, v_sqrt) ~8; an IEEE division expands to ~10
// instructions (~38 cycles).  The MCM pass is VALU-issue bound, so the rules are: keep float work in
// fma form, avoid compare+select pairs, move table lookups to LDS (its own issue port).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VPT_DEV __device__ __forceinline__

struct f3 { float x, y, z; };
struct f2 { float x, y; };

// ---- GLSL built-ins ------------------------------------------------------------------------
// min / max: IEEE minNum / maxNum (a NaN operand yields the other one; -0 < +0) == v_min_f32 / v_max_f32.
// GLSL leaves NaN operands undefined (ES 3.00 §8.3), so any consistent choice is within the language.
VPT_DEV float vmin(float a, float b) { return fminf(a, b); }
VPT_DEV float vmax(float a, float b) { return fmaxf(a, b); }
VPT_DEV float vclamp01(float x) { return vmin(vmax(x, 0.0f), 1.0f); }
VPT_DEV float mixf(float a, float b, float t) { return fmaf(b, t, a * (1.0f - t)); }  // x*(1-a)+y*a
VPT_DEV f3 mix3(f3 a, f3 b, float t) { return f3{ mixf(a.x, b.x, t), mixf(a.y, b.y, t), mixf(a.z, b.z, t) }; }
VPT_DEV float dot3(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
VPT_DEV float length3(f3 a) { return sqrtf(dot3(a, a)); }
VPT_DEV f3 sub3(f3 a, f3 b) { return f3{ a.x - b.x, a.y - b.y, a.z - b.z }; }
VPT_DEV f3 madd3(f3 p, float t, f3 d) { return f3{ fmaf(t, d.x, p.x), fmaf(t, d.y, p.y), fmaf(t, d.z, p.z) }; }
VPT_DEV float lerpf(float a, float b, float f) { return fmaf(f, b - a, a); }
VPT_DEV float4 lerp4(float4 a, float4 b, float f) {
    return make_float4(lerpf(a.x, b.x, f), lerpf(a.y, b.y, f), lerpf(a.z, b.z, f), lerpf(a.w, b.w, f));
}

// ---- software reciprocal / reciprocal square root ---------------------------------------------
// rcp_nr(x) ~ 1/x: integer seed + 3 Newton-Raphson steps (7 instructions, ~16 cycles vs ~38 for IEEE `/`);
// |relative error| < 2.5e-7 (<= 2 ulp) for normal x with 2^-125 <= |x| <= 2^125, within the 2.5 ULP GLSL ES 3.00
// §4.5.1 allows for a/b.  Used where the divisor cannot be 0/inf in a sane scene (homogeneous w, sample counts,
// segment lengths); x = 0, inf, NaN give NaN.  Slab tests keep IEEE `/` because 1/0 = +-inf is semantic there.
VPT_DEV float rcp_nr(float x) {
    float r = __uint_as_float(0x7EF311C7u - __float_as_uint(x));
    r = fmaf(fmaf(-x, r, 1.0f), r, r);
    r = fmaf(fmaf(-x, r, 1.0f), r, r);
    r = fmaf(fmaf(-x, r, 1.0f), r, r);
    return r;
}
// rcp_nrz: the same with the iterate clamped to +-FLT_MAX before the last step, so that 1/(+-0) = +-inf (and tiny
// denormals overflow to inf) as the slab test of intersectCube needs; identical to rcp_nr wherever that is finite.
VPT_DEV float rcp_nrz(float x) {
    float r = __uint_as_float(0x7EF311C7u - __float_as_uint(x));
    r = fmaf(fmaf(-x, r, 1.0f), r, r);
    r = fmaf(fmaf(-x, r, 1.0f), r, r);
    r = __builtin_amdgcn_fmed3f(r, -3.402823466e+38f, 3.402823466e+38f);
    r = fmaf(fmaf(-x, r, 1.0f), r, r);
    return r;
}
// rsqrt_nr(x) ~ 1/sqrt(x) for normal x > 0: integer seed + 3 Newton-Raphson steps; relative error < 3e-7.
// GLSL normalize(v) = v * inversesqrt(dot(v,v)) (inversesqrt: 2 ULP, ES 3.00 §4.5.1).
VPT_DEV float rsqrt_nr(float x) {
    float y = __uint_as_float(0x5F375A86u - (__float_as_uint(x) >> 1));
    float h = 0.5f * x;
    y = y * fmaf(-h * y, y, 1.5f);
    y = y * fmaf(-h * y, y, 1.5f);
    y = y * fmaf(-h * y, y, 1.5f);
    return y;
}
VPT_DEV f3 normalize3(f3 a) { float inv = rsqrt_nr(dot3(a, a)); return f3{ a.x * inv, a.y * inv, a.z * inv }; }
// sqrt_nr(x) = x * rsqrt_nr(x): sqrt(+0) = +0 (the seed stays finite); GLSL: sqrt inherits 1/inversesqrt precision
VPT_DEV float sqrt_nr(float x) { return x * rsqrt_nr(x); }

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
// the same for x = k * 2^-32 (k = 1 .. 2^32, the range of random_uniform except 0): no NaN / inf / negative cases
VPT_DEV float vpt_logf_uniform(float x) {
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
// -log(u)/rate as -log(u) * inv_rate, inv_rate = 1/rate once per pass (contract)
VPT_DEV float random_exponential(uint32_t &state, float inv_rate) { return -vpt_logf_uniform(random_uniform(state)) * inv_rate; }
VPT_DEV f2 random_disk(uint32_t &state) {
    float radius = sqrt_nr(random_uniform(state));
    float angle = 6.28318530718f * random_uniform(state);
    float s, c; vpt_sincosf(angle, s, c);
    return f2{ radius * c, radius * s };
}
VPT_DEV f3 random_sphere(uint32_t &state) {
    f2 d = random_disk(state);
    float norm = fmaf(d.y, d.y, d.x * d.x);
    float radius = 2.0f * sqrt_nr(1.0f - norm);
    float z = fmaf(-2.0f, norm, 1.0f);
    return f3{ radius * d.x, radius * d.y, z };
}

// ---- render-target conversions --------------------------------------------------------------
VPT_DEV uint32_t to_unorm8(float f) { return (uint32_t)rintf(vclamp01(f) * 255.0f); }   // NaN -> 0 (maxNum)
VPT_DEV float from_unorm8(uint32_t c) { return (float)c / 255.0f; }
// RGBA16F write: RNE conversion of the ALREADY ROUNDED fp32 value (a shader outputs fp32; the render target converts).
// Written as an opaque instruction: given `(half)fma(a, b, c)` or `(half)(a * b)` the compiler otherwise selects
// v_fma_mixlo_f16, which rounds the exact result ONCE to half — a different value in the rare double-rounding cases
// (found by tests/test_gpu_fuzz.py: 1 pixel in 5760 of a Depth frame).
VPT_DEV uint16_t to_half_bits(float f) {
    uint32_t h;
    asm("v_cvt_f16_f32 %0, %1" : "=v"(h) : "v"(f));
    return (uint16_t)h;
}

// ---- ray set-up -----------------------------------------------------------------------------
struct Mat4 { float m[16]; };
// inverseMvp * (x, y, z, 1): the constant part is summed first (c = m[8..11]*z + m[12..15], a pass constant for z = -1
// and z = +1), then the x and y terms are accumulated — 2 fma per row in the per-event code.  GLSL does not fix the
// summation order of a matrix-vector product.
VPT_DEV float4 mat4_mul_point(const Mat4 &M, float x, float y, float z) {
    const float *m = M.m;
    float4 r;
    r.x = fmaf(m[4], y, fmaf(m[0], x, fmaf(m[8],  z, m[12])));
    r.y = fmaf(m[5], y, fmaf(m[1], x, fmaf(m[9],  z, m[13])));
    r.z = fmaf(m[6], y, fmaf(m[2], x, fmaf(m[10], z, m[14])));
    r.w = fmaf(m[7], y, fmaf(m[3], x, fmaf(m[11], z, m[15])));
    return r;
}
VPT_DEV f3 dehomogenize(float4 v) { float i = rcp_nr(v.w); return f3{ v.x * i, v.y * i, v.z * i }; }   // xyz / w
// mixins/unproject.glsl:3-10
VPT_DEV void unproject(float px, float py, const Mat4 &M, f3 &from, f3 &to) {
    from = dehomogenize(mat4_mul_point(M, px, py, -1.0f));
    to = dehomogenize(mat4_mul_point(M, px, py, 1.0f));
}
VPT_DEV float pixel_ndc(int i, int n) { return (float)(2 * i + 1) / (float)n - 1.0f; }
VPT_DEV float ndc_to_uv(float p) { return fmaf(p, 0.5f, 0.5f); }

// mixins/intersectCube.glsl:3-11; (a - o) / d as (a - o) * rcp(d), with the reciprocal form whose 1/0 is +-inf
VPT_DEV f3 cube_inv_dir(f3 d) { return f3{ rcp_nrz(d.x), rcp_nrz(d.y), rcp_nrz(d.z) }; }
VPT_DEV f2 intersect_cube(f3 o, f3 d) {
    f3 inv = cube_inv_dir(d);
    f3 tmin = { (0.0f - o.x) * inv.x, (0.0f - o.y) * inv.y, (0.0f - o.z) * inv.z };
    f3 tmax = { (1.0f - o.x) * inv.x, (1.0f - o.y) * inv.y, (1.0f - o.z) * inv.z };
    f3 t1 = { vmin(tmin.x, tmax.x), vmin(tmin.y, tmax.y), vmin(tmin.z, tmax.z) };
    f3 t2 = { vmax(tmin.x, tmax.x), vmax(tmin.y, tmax.y), vmax(tmin.z, tmax.z) };
    return f2{ vmax(vmax(t1.x, t1.y), t1.z), vmin(vmin(t2.x, t2.y), t2.z) };
}
// only tnear (resetPhoton, MCMRenderer.glsl:75-76)
VPT_DEV float intersect_cube_near(f3 o, f3 d) {
    f3 inv = cube_inv_dir(d);
    float ax = vmin((0.0f - o.x) * inv.x, (1.0f - o.x) * inv.x);
    float ay = vmin((0.0f - o.y) * inv.y, (1.0f - o.y) * inv.y);
    float az = vmin((0.0f - o.z) * inv.z, (1.0f - o.z) * inv.z);
    return vmax(vmax(ax, ay), az);
}

// ---- bricked Z-order volume -------------------------------------------------------------------
// Layout in HBM: 4^3-voxel bricks stored with a +1 apron = 5^3 = 125 bytes in a 128-byte slot
// (one L2 line holds every tap of a trilinear sample; RG8 volumes: 256-byte slots, R brick at +0, G brick at +128).
// Brick (bx,by,bz) sits at slot CX[bx] | CY[by] | CZ[bz]: a Z-order code with as many bits per axis as the axis needs
// (the classic Morton code for a cube; built by the host, vpt_volume_create).
// Inside a slot the byte of local voxel (lx,ly,lz) in [0,5)^3 is lz*25 + ly*5 + lx.
//
// Addressing: the byte offset of voxel cell (x,y,z) is SEPARABLE:
//     off(x,y,z) = TX[x] + TY[y] + TZ[z],   TX[i] = (CX[i>>2] << s) + (i&3),
//                                             TY[i] = (CY[i>>2] << s) + (i&3)*5,
//                                             TZ[i] = (CZ[i>>2] << s) + (i&3)*25        (s = 7, or 8 for RG8)
// The three tables (nx+ny+nz dwords) are staged in LDS once per workgroup, so a sample's address costs three
// ds_read_b32 and one v_add3_u32 instead of ~30 VALU instructions of bit interleaving, and the 32-bit sum feeds
// global_load's SGPR-base + VGPR-offset form directly (bricked size <= 4 GiB; above that the tables hold the brick
// codes alone and the in-brick offset is computed, cell_addr<WIDE>).
#define VPT_BRICK        4
#define VPT_BRICK_SHIFT  2
#define VPT_BRICK_BYTES  128

struct DevVolume {
    const uint8_t *bricks;
    const uint32_t *tab32;   // TX | TY | TZ (nx + ny + nz entries), 32-bit offsets
    const uint32_t *tabc;    // Morton brick codes only (bricked size > 4 GiB: the byte offset no longer fits 32 bits)
    int nx, ny, nz;
    float fnx, fny, fnz;     // (float)n
    float hx, hy, hz;        // (float)(n - 1)
    int filter;              // VPT_FILTER_*
    int channels;            // 1 = R8, 2 = RG8: the G brick follows the R brick in a 256-byte slot (R at +0, G at +128)
    uint32_t slot_shift;     // log2 of the slot size: 7 (R8) or 8 (RG8)
    uint32_t elem_shift;     // log2 of the bytes per texel channel: 0 (UNSIGNED_BYTE) or 2 (FLOAT)
    const uint32_t *atlas;   // boundary atlas (one-channel volumes; null: not built or switched off) — see sample_volume_boundary
    uint32_t atlas_face, atlas_shift;   // dwords per face image, log2 of its row pitch
    // column records (one-channel byte volumes, LINEAR filter; null: not built or switched off) — see record_addr
    const uint8_t *records;
    const uint32_t *rtab32;  // RX | RY (nx + ny entries): byte offset of the column's first record
    const uint32_t *rtabc;   // the columns' Z-order codes alone (records beyond 4 GiB)
    uint32_t rec_col_bytes;  // bytes of one column: nz * 4
};
// LDS image of the per-workgroup tables: [tf pairs][TX][TY][TZ]
struct LdsTables {
    const float4 *tf;        // tf_w pairs: { t[i], t[min(i+1,w-1)] - t[i] }
    const uint32_t *tx, *ty, *tz;
};

// LINEAR filter cell: u = s*N - 0.5 clamped to [0, N-1]; i = trunc(u); f = u - i.
// Equal to the GL definition (taps clamp(i0), clamp(i0+1) of the unclamped u; oracle linear_coord) because every
// clamped case degenerates to an exact edge value: u < 0 -> (t[0], f = 0); u >= N-1 -> (t[N-1], apron = t[N-1]).
// 4 instructions: v_fma, v_med3 (clamp; a NaN operand makes it return min3 of the others = 0), v_fract (= u - floor(u),
// exact for u >= 0), v_cvt_u32 (truncation == floor for u >= 0).
VPT_DEV void linear_cell(float s, float fn, float hi, uint32_t &i, float &f) {
    float u = __builtin_amdgcn_fmed3f(fmaf(s, fn, -0.5f), 0.0f, hi);
    f = __builtin_amdgcn_fractf(u);
    i = (uint32_t)u;
}
VPT_DEV uint32_t nearest_cell(float s, float fn, float hi) {
    return (uint32_t)__builtin_amdgcn_fmed3f(s * fn, 0.0f, hi);   // u >= 0: truncation == floor
}
// texture(uVolume, p).r for an R8 volume (Volume.js:49-60): integer texel values interpolated x, y, z,
// normalised once by * fl32(1/255)  (255 * VPT_INV255 == 1.0f exactly).
#define VPT_INV255 0.00392156862745098f
template <bool WIDE>
VPT_DEV const uint8_t *cell_addr(const DevVolume &v, const LdsTables &t, uint32_t x, uint32_t y, uint32_t z) {
    if (WIDE) {
        // > 4 GiB of bricks (2048^3: 16 GiB): the tables hold the 27-bit Morton code of the brick only (still 32-bit
        // entries: 24 KB of LDS for 2048^3 where 64-bit byte offsets took 48 KB and halved the occupancy); the in-brick
        // offset costs five VALU instructions instead of riding along in the table
        uint32_t code = t.tx[x] + t.ty[y] + t.tz[z];
        uint32_t intra = ((x & 3u) + (y & 3u) * 5u + (z & 3u) * 25u) << v.elem_shift;
        return v.bricks + (((uint64_t)code << v.slot_shift) + intra);
    }
    return v.bricks + (uint32_t)(t.tx[x] + t.ty[y] + t.tz[z]);
}
// V: variant bits fixed at launch — bit 0 = 64-bit offset tables (WIDE), bit 1 = NEAREST filter.  No run-time branch
// inside the sampler: consecutive samples of a ray stay straight-line code, so their loads are issued together.
#define VPT_V_WIDE    1
#define VPT_V_NEAREST 2
#define VPT_V_ALIGNED 4   // fetch the two tap windows as dword-aligned 12-byte loads + v_alignbyte (texture-path bound kernels)
#define VPT_V_FAST    16  // MCM / MCS: hardware rcp / rsq / sqrt / log / sin / cos and shorter algebraic forms (no bit-exact CPU twin; VPT_OPTION_FAST_MATH)
#define VPT_V_F32     32  // FLOAT texels (R32F; R16F widened on upload): 5^3 floats in a 512-byte slot, no normalisation
#define VPT_V_RG      8   // two-channel (RG8) volume: texture(uVolume, p).rg has both channels, the transfer function is looked up in 2-D
#define VPT_V_REC     64  // in-cube samples from the column records instead of the bricks (one-channel byte volumes, LINEAR filter; MCM)
// the eight taps around a cell of one channel's brick and their trilinear blend: taps +0,+1 (y,z) ; +5,+6 (y+1,z) ;
// +25,+26 (y,z+1) ; +30,+31 (y+1,z+1) = two 8-byte windows of one line.
// tools/gather_rates.hip (MI355X, L1-resident gathers): a dword-aligned 8/12/16-byte wave load costs ~27-33 cycles
// of the CU's texture path, a byte-aligned 8-byte one 2x that.  Kernels bound by that path (MIP, EAM: ~60 VALU
// instructions per sample) fetch 12 aligned bytes per window and realign in registers (v_alignbyte_b32); the
// VALU-bound MCM / MCS keep the two unaligned 8-byte loads (fewer instructions).
template <int B> VPT_DEV float cvt_ubyte(uint32_t w) {
    float r;
    if (B == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(r) : "v"(w));
    else if (B == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(r) : "v"(w));
    else if (B == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(r) : "v"(w));
    else asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(r) : "v"(w));
    return r;
}
// the blend of the eight taps held in the two 8-byte windows (l0 | h0 = window +0, l1 | h1 = window +25)
VPT_DEV float trilinear_blend(uint32_t l0, uint32_t h0, uint32_t l1, uint32_t h1, float fx, float fy, float fz) {
    // byte -> float straight out of the loaded dwords (v_cvt_f32_ubyteN: the byte select is free).  Written as opaque
    // instructions: left to itself the compiler turns (float)b - (float)a into (float)(b - a) and spends 18 integer-class
    // instructions (shifts, SDWA subtracts, two conversions per pair) where 8 conversions + 4 fp32 subtracts do —
    // integer-class VALU instructions cost ~4.3 cycles per wave on this chip against ~2.6 for fp32 add/mul/fma
    // (tools/valu_rates.hip).  Same values, same results.
    float c000 = cvt_ubyte<0>(l0), c100 = cvt_ubyte<1>(l0);
    float c010 = cvt_ubyte<1>(h0), c110 = cvt_ubyte<2>(h0);
    float c001 = cvt_ubyte<0>(l1), c101 = cvt_ubyte<1>(l1);
    float c011 = cvt_ubyte<1>(h1), c111 = cvt_ubyte<2>(h1);
    float c00 = lerpf(c000, c100, fx), c10 = lerpf(c010, c110, fx);
    float c01 = lerpf(c001, c101, fx), c11 = lerpf(c011, c111, fx);
    float c0 = lerpf(c00, c10, fy), c1 = lerpf(c01, c11, fy);
    return lerpf(c0, c1, fz) * VPT_INV255;
}
// the two 12-byte windows of the ALIGNED form, by GLOBAL loads: `a` = the (byte-aligned) address of the cell's first tap.  The dword-aligned
// address is formed by pointer arithmetic — round 3 went through uintptr_t, which loses the address space: the compiler emitted flat_load,
// whose completion also counts on lgkmcnt, so every wait for an LDS table lookup of the NEXT sample waited for the bricks of THIS one (round 4)
struct TapWindow { uint32_t a, b, c; };
VPT_DEV TapWindow load_window(const uint8_t *a, uint32_t phase) {
    TapWindow w;
    __builtin_memcpy(&w, __builtin_assume_aligned(a - phase, 4), 12);
    return w;
}
// the same with the address still in its two parts — the uniform base of the brick array and the 32-bit offset of the cell (volumes up to
// 4 GiB of bricks) —: the dword alignment is taken off the OFFSET, and the load keeps global_load's SGPR-base + VGPR-offset form (no 64-bit
// address arithmetic per window).  The brick array is 256-byte aligned, so the byte phase of the address is the offset's.
VPT_DEV float aligned_taps(const uint8_t *base, uint32_t off, float fx, float fy, float fz) {
    const uint32_t o1 = off + 25u, s0 = off & 3u, s1 = o1 & 3u;
    TapWindow q0, q1;
    __builtin_memcpy(&q0, __builtin_assume_aligned(base + (off - s0), 4), 12);
    __builtin_memcpy(&q1, __builtin_assume_aligned(base + (o1 - s1), 4), 12);
    const uint32_t l0 = __builtin_amdgcn_alignbyte(q0.b, q0.a, s0), h0 = __builtin_amdgcn_alignbyte(q0.c, q0.b, s0);
    const uint32_t l1 = __builtin_amdgcn_alignbyte(q1.b, q1.a, s1), h1 = __builtin_amdgcn_alignbyte(q1.c, q1.b, s1);
    return trilinear_blend(l0, h0, l1, h1, fx, fy, fz);
}
template <int V>
VPT_DEV float trilinear_taps(const uint8_t *a, float fx, float fy, float fz) {
    uint32_t l0, h0, l1, h1;
    if (V & VPT_V_ALIGNED) {
        const uint32_t s0 = (uint32_t)(uintptr_t)a & 3u, s1 = ((uint32_t)(uintptr_t)a + 25u) & 3u;   // (v_alignbyte_b32 reads S2[4:0] on CDNA: mask)
        const TapWindow q0 = load_window(a, s0), q1 = load_window(a + 25, s1);
        l0 = __builtin_amdgcn_alignbyte(q0.b, q0.a, s0); h0 = __builtin_amdgcn_alignbyte(q0.c, q0.b, s0);
        l1 = __builtin_amdgcn_alignbyte(q1.b, q1.a, s1); h1 = __builtin_amdgcn_alignbyte(q1.c, q1.b, s1);
    } else {
        uint64_t w0, w1;
        __builtin_memcpy(&w0, a, 8);
        __builtin_memcpy(&w1, a + 25, 8);
        l0 = (uint32_t)w0; h0 = (uint32_t)(w0 >> 32); l1 = (uint32_t)w1; h1 = (uint32_t)(w1 >> 32);
    }
    return trilinear_blend(l0, h0, l1, h1, fx, fy, fz);
}
// ---- column records (round 4) -------------------------------------------------------------------------------------------
// The photons of the MCM renderer sample at independent random positions: after its first event no two lanes of a wave share a
// brick, every in-cube sample is a cold line, and what it costs is the texture path's work per gather (two byte-aligned 8-byte
// windows: ~2 x 60 cycles per wave instruction against ~30 for a dword-aligned load, tools/gather_rates.hip) and the 64-byte
// sectors it pulls (1.5 on average for the two windows of a 128-byte brick line).  So the volume is kept a THIRD time for this
// access pattern, the way the boundary atlas keeps the faces: record(x, y, z) = the four texels of the cell's x-y footprint in ONE
// dword — [t(x,y,z), t(x+1,y,z), t(x,y+1,z), t(x+1,y+1,z)], indices clamped like the apron — and the records of a voxel column
// (x, y, 0 .. nz-1) contiguous, so that the eight taps of a trilinear sample are records z and z + 1 of one column: ONE dword-aligned
// 8-byte gather from one 64-byte sector (two where the pair straddles a sector: 1 in 16).  Same taps, same x -> y -> z lerps as
// trilinear_blend: bit-identical.  4 bytes per voxel (512^3: 512 MiB, 1024^3: 4 GiB, 2048^3: 32 GiB of the 288 GB).  Columns are
// ordered by a 2-D Z-order code of (x, y) with as many bits per axis as the axis needs; the address is separable,
//     off(x, y, z) = RX[x] + RY[y] + 4 z,   RX[x] = CX[x] * 4 nz,   RY[y] = CY[y] * 4 nz
// two LDS lookups and a shift-add (beyond 4 GiB the tables hold the column code and the product is one v_mad_u64_u32).
// At z = nz - 1 the filter weight fz is exactly 0 (linear_cell), so what the load's second dword holds — the next column's first
// record, or the 8 bytes of padding behind the last column — never reaches the result.
template <bool WIDE>
VPT_DEV const uint8_t *record_addr(const DevVolume &v, const LdsTables &t, uint32_t x, uint32_t y, uint32_t z) {
    if (WIDE) return v.records + ((uint64_t)(t.tx[x] + t.ty[y]) * v.rec_col_bytes + (z << 2));
    return v.records + (uint32_t)(t.tx[x] + t.ty[y] + (z << 2));
}
// the blend of records z (lo) and z + 1 (hi): the same values and the same order of operations as trilinear_blend
VPT_DEV float record_blend(uint32_t lo, uint32_t hi, float fx, float fy, float fz) {
    float c000 = cvt_ubyte<0>(lo), c100 = cvt_ubyte<1>(lo), c010 = cvt_ubyte<2>(lo), c110 = cvt_ubyte<3>(lo);
    float c001 = cvt_ubyte<0>(hi), c101 = cvt_ubyte<1>(hi), c011 = cvt_ubyte<2>(hi), c111 = cvt_ubyte<3>(hi);
    float c00 = lerpf(c000, c100, fx), c10 = lerpf(c010, c110, fx);
    float c01 = lerpf(c001, c101, fx), c11 = lerpf(c011, c111, fx);
    float c0 = lerpf(c00, c10, fy), c1 = lerpf(c01, c11, fy);
    return lerpf(c0, c1, fz) * VPT_INV255;
}
VPT_DEV uint64_t record_load(const uint8_t *a) {
    uint64_t w;
    __builtin_memcpy(&w, __builtin_assume_aligned(a, 4), 8);
    return w;
}
// ---- boundary atlas ---------------------------------------------------------------------------------------------------
// For a position with a coordinate outside [0, 1] (MCM samples before its bounds test, MCMRenderer.glsl:132-142) the clamped
// filter cell of that axis k is u_k = 0 or N_k - 1 exactly: i_k is the first or last voxel plane and f_k = 0, so
// lerp(a, b, 0) = fma(0, b - a, a) = a EXACTLY and the trilinear blend above equals the bilinear blend of four taps on that
// plane, in the same x -> y -> z order of the remaining two axes.  The six boundary planes are kept a second time as "face
// images" with the 2 x 2 footprint of every cell in ONE dword — [t(a,b), t(a+1,b), t(a,b+1), t(a+1,b+1)], indices clamped like
// the bricks' apron — so such a sample costs one aligned 4-byte gather from a small table (6 MiB for 512^3; L2-resident)
// instead of two unaligned 8-byte gathers from the bricks.  Face f = 2 * axis + side (side 1 = plane N_k - 1) starts at dword
// f * atlas_face; cell (a, b) of a face sits at (b << atlas_shift) + a with (a, b) = (y, z), (x, z), (x, y) for axis x, y, z.
// Precondition: some coordinate of p is > 1 or < 0 (not NaN) — the caller's bounds test.
VPT_DEV uint32_t boundary_cell(const DevVolume &v, f3 p, float &out_fa, float &out_fb) {
    // (members copied into locals first: `c ? v.fny : v.fnx` on struct members is an lvalue conditional — a select of ADDRESSES
    // into the kernel argument block, which then has to live in scratch memory: measured 5x slower)
    const float fnx = v.fnx, fny = v.fny, fnz = v.fnz, hx = v.hx, hy = v.hy, hz = v.hz;
    const uint32_t face = v.atlas_face, sh = v.atlas_shift;
    const bool ox = (p.x > 1.0f) || (p.x < 0.0f);
    const bool oy = (p.y > 1.0f) || (p.y < 0.0f);
    // The face is the first out-of-range axis k; its side follows from the coordinate itself (p_k > 1: plane N_k - 1, p_k < 0:
    // plane 0), so the clamped axis needs no filter cell at all.  The photons of a wave mostly leave through the same face
    // (8 x 8 neighbouring pixels): the three wave-uniform cases are separate straight-line paths without per-lane selects,
    // and only a wave whose lanes disagree takes the generic one.
    // (ballots of the four compares themselves, OR-ed as scalars: a ballot of `ox` makes the compiler materialise the predicate as
    // 0 / 1 in a VGPR and compare it again — four VALU instructions per sample for nothing)
    const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
    const unsigned long long bx = __builtin_amdgcn_ballot_w64(p.x > 1.0f) | __builtin_amdgcn_ballot_w64(p.x < 0.0f);
    const unsigned long long by = __builtin_amdgcn_ballot_w64(p.y > 1.0f) | __builtin_amdgcn_ballot_w64(p.y < 0.0f);
    uint32_t a, b, idx; float fa, fb;
    if (bx == act) {
        linear_cell(p.y, fny, hy, a, fa); linear_cell(p.z, fnz, hz, b, fb);
        idx = (p.x > 1.0f ? face : 0u) + ((b << sh) + a);
    } else if (bx == 0ull && by == act) {
        linear_cell(p.x, fnx, hx, a, fa); linear_cell(p.z, fnz, hz, b, fb);
        idx = (p.y > 1.0f ? 3u * face : 2u * face) + ((b << sh) + a);
    } else if (bx == 0ull && by == 0ull) {
        linear_cell(p.x, fnx, hx, a, fa); linear_cell(p.y, fny, hy, b, fb);
        idx = (p.z > 1.0f ? 5u * face : 4u * face) + ((b << sh) + a);
    } else {
        const bool oxy = ox || oy;
        const float pa = ox ? p.y : p.x, pb = oxy ? p.z : p.y;
        linear_cell(pa, ox ? fny : fnx, ox ? hy : hx, a, fa);
        linear_cell(pb, oxy ? fnz : fny, oxy ? hz : hy, b, fb);
        const float pk = ox ? p.x : (oy ? p.y : p.z);
        const uint32_t f = (ox ? 0u : (oy ? 2u : 4u)) + (pk > 1.0f ? 1u : 0u);
        idx = f * face + ((b << sh) + a);
    }
    out_fa = fa; out_fb = fb;
    return idx;
}
// the same cell per lane, without the wave-uniform fast paths, for either filter (the MISS-tile kernel of NEAREST / two-channel / float
// volumes): NEAREST takes texel clamp(floor(s N)) of each in-face axis — what nearest_cell gives the brick sampler — and no weights
template <bool NEAREST>
VPT_DEV uint32_t boundary_cell_lane(const DevVolume &v, f3 p, float &out_fa, float &out_fb) {
    const float fnx = v.fnx, fny = v.fny, fnz = v.fnz, hx = v.hx, hy = v.hy, hz = v.hz;
    const bool ox = (p.x > 1.0f) || (p.x < 0.0f);
    const bool oy = (p.y > 1.0f) || (p.y < 0.0f);
    const bool oxy = ox || oy;
    const float pa = ox ? p.y : p.x, pb = oxy ? p.z : p.y;
    const float fna = ox ? fny : fnx, ha = ox ? hy : hx, fnb = oxy ? fnz : fny, hb = oxy ? hz : hy;
    uint32_t a, b; float fa = 0.0f, fb = 0.0f;
    if (NEAREST) { a = nearest_cell(pa, fna, ha); b = nearest_cell(pb, fnb, hb); }
    else { linear_cell(pa, fna, ha, a, fa); linear_cell(pb, fnb, hb, b, fb); }
    const float pk = ox ? p.x : (oy ? p.y : p.z);
    const uint32_t f = (ox ? 0u : (oy ? 2u : 4u)) + (pk > 1.0f ? 1u : 0u);
    out_fa = fa; out_fb = fb;
    return f * v.atlas_face + ((b << v.atlas_shift) + a);
}
VPT_DEV float boundary_blend(uint32_t w, float fa, float fb) {
    float c00 = cvt_ubyte<0>(w), c10 = cvt_ubyte<1>(w), c01 = cvt_ubyte<2>(w), c11 = cvt_ubyte<3>(w);
    return lerpf(lerpf(c00, c10, fa), lerpf(c01, c11, fa), fb) * VPT_INV255;
}
// texture(uVolume, clamp(p)).rg through the boundary atlas for ANY volume format (V: VPT_V_NEAREST | VPT_V_RG | VPT_V_F32): channel c's faces lie
// 6 * atlas_face cells behind channel c - 1's; float volumes keep four floats per cell.  Same texels, same order of operations as
// sample_volume_rg<V> at the clamped position (float texels: finite ones, see vpt_volume_finalize).  Precondition as boundary_cell's.
template <int V>
VPT_DEV f2 sample_boundary_rg(const DevVolume &v, f3 p) {
    constexpr bool NEAREST = (V & VPT_V_NEAREST) != 0, RG = (V & VPT_V_RG) != 0, F32 = (V & VPT_V_F32) != 0;
    float fa, fb;
    const uint32_t idx = boundary_cell_lane<NEAREST>(v, p, fa, fb);
    float val[2] = { 0.0f, 0.0f };
#pragma unroll
    for (int c = 0; c < (RG ? 2 : 1); c++) {
        const uint32_t cell = idx + (uint32_t)c * 6u * v.atlas_face;
        if (F32) {
            const float4 t = ((const float4 *)v.atlas)[cell];
            val[c] = NEAREST ? t.x : lerpf(lerpf(t.x, t.y, fa), lerpf(t.z, t.w, fa), fb);
        } else {
            const uint32_t w = v.atlas[cell];
            val[c] = NEAREST ? cvt_ubyte<0>(w) * VPT_INV255 : boundary_blend(w, fa, fb);
        }
    }
    return f2{ val[0], val[1] };
}
VPT_DEV float sample_volume_boundary(const DevVolume &v, f3 p) {
    float fa, fb;
    const uint32_t idx = boundary_cell(v, p, fa, fb);
    return boundary_blend(v.atlas[idx], fa, fb);
}
// texture(uVolume, p).rg: r always, g only for RG8 volumes (V & VPT_V_RG; an R8 volume has g = 0).  The cell and its
// brick address are computed once for both channels.
template <int V>
VPT_DEV f2 sample_volume_rg(const DevVolume &v, const LdsTables &t, f3 p) {
    constexpr bool WIDE = (V & VPT_V_WIDE) != 0;
    constexpr bool RG = (V & VPT_V_RG) != 0;
    if (V & VPT_V_F32) {
        // FLOAT texels (Volume.js:84-105 `FLOAT` / `HALF_FLOAT`; LINEAR filtering of float textures: OES_texture_float_linear,
        // RenderingContext.js:78): the brick holds 5^3 floats, a row's two taps are one dword-aligned 8-byte load; same
        // x -> y -> z lerps, the value is used as it is (no normalisation)
        if (V & VPT_V_NEAREST) {
            uint32_t x = nearest_cell(p.x, v.fnx, v.hx), y = nearest_cell(p.y, v.fny, v.hy), z = nearest_cell(p.z, v.fnz, v.hz);
            const float *b = (const float *)cell_addr<WIDE>(v, t, x, y, z);
            return f2{ b[0], RG ? b[128] : 0.0f };                          // (two channels: the G brick sits 128 floats behind the R brick)
        }
        uint32_t x, y, z; float fx, fy, fz;
        linear_cell(p.x, v.fnx, v.hx, x, fx);
        linear_cell(p.y, v.fny, v.hy, y, fy);
        linear_cell(p.z, v.fnz, v.hz, z, fz);
        const float *b0 = (const float *)cell_addr<WIDE>(v, t, x, y, z);
        f2 out = { 0.0f, 0.0f };
#pragma unroll
        for (int c = 0; c < (RG ? 2 : 1); c++) {
            const float *b = b0 + 128 * c;
            // (the taps are dword-aligned, not 8-byte aligned: loaded through memcpy — still one global_load_dwordx2 each)
            float2 r00, r10, r01, r11;
            __builtin_memcpy(&r00, b, 8); __builtin_memcpy(&r10, b + 5, 8); __builtin_memcpy(&r01, b + 25, 8); __builtin_memcpy(&r11, b + 30, 8);
            float c00 = lerpf(r00.x, r00.y, fx), c10 = lerpf(r10.x, r10.y, fx);
            float c01 = lerpf(r01.x, r01.y, fx), c11 = lerpf(r11.x, r11.y, fx);
            const float val = lerpf(lerpf(c00, c10, fy), lerpf(c01, c11, fy), fz);
            if (c == 0) out.x = val; else out.y = val;
        }
        return out;
    }
    if (V & VPT_V_NEAREST) {
        uint32_t x = nearest_cell(p.x, v.fnx, v.hx), y = nearest_cell(p.y, v.fny, v.hy), z = nearest_cell(p.z, v.fnz, v.hz);
        const uint8_t *a = cell_addr<WIDE>(v, t, x, y, z);
        return f2{ (float)a[0] * VPT_INV255, RG ? (float)a[128] * VPT_INV255 : 0.0f };
    }
    uint32_t x, y, z; float fx, fy, fz;
    linear_cell(p.x, v.fnx, v.hx, x, fx);
    linear_cell(p.y, v.fny, v.hy, y, fy);
    linear_cell(p.z, v.fnz, v.hz, z, fz);
    if (V & VPT_V_REC) {                                          // (one channel: the launch code never combines REC with RG)
        const uint64_t w = record_load(record_addr<WIDE>(v, t, x, y, z));
        return f2{ record_blend((uint32_t)w, (uint32_t)(w >> 32), fx, fy, fz), 0.0f };
    }
    if ((V & VPT_V_ALIGNED) && !WIDE) {
        const uint32_t off = t.tx[x] + t.ty[y] + t.tz[z];
        return f2{ aligned_taps(v.bricks, off, fx, fy, fz), RG ? aligned_taps(v.bricks, off + 128u, fx, fy, fz) : 0.0f };
    }
    const uint8_t *a = cell_addr<WIDE>(v, t, x, y, z);
    return f2{ trilinear_taps<V>(a, fx, fy, fz), RG ? trilinear_taps<V>(a + 128, fx, fy, fz) : 0.0f };
}
template <int V>
VPT_DEV float sample_volume(const DevVolume &v, const LdsTables &t, f3 p) { return sample_volume_rg<V & ~VPT_V_RG>(v, t, p).x; }

// transfer function: row 0 of the decoded SRGB8_ALPHA8 table, LINEAR / CLAMP_TO_EDGE, staged in LDS as
// (value, forward difference) pairs.  R8 volume => lookup at (r, 0): both bilinear rows clamp to row 0, so row 0
// alone is exact.  Same clamping argument as linear_cell.
VPT_DEV float4 sample_tf(const float4 *tf_pairs, float tf_fw, float tf_hi, float r) {
    float u = __builtin_amdgcn_fmed3f(fmaf(r, tf_fw, -0.5f), 0.0f, tf_hi);
    float f = __builtin_amdgcn_fractf(u);
    uint32_t i = (uint32_t)u;
    float4 a = tf_pairs[2 * i], d = tf_pairs[2 * i + 1];
    return make_float4(fmaf(f, d.x, a.x), fmaf(f, d.y, a.y), fmaf(f, d.z, a.z), fmaf(f, d.w, a.w));
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
// the whole transfer function (w x h decoded float4 in HBM, L2-resident), LINEAR / CLAMP_TO_EDGE in both axes: the
// lookup of RG8 volumes, texture(uTransferFunction, volumeSample.rg) (x lerps first, then y, as the oracle's sample_2d)
VPT_DEV float4 sample_tf2d(const float4 *tf, int w, int h, float r, float g) {
    int x0, x1, y0, y1; float fx, fy;
    linear_taps(r, w, x0, x1, fx);
    linear_taps(g, h, y0, y1, fy);
    float4 r0 = lerp4(tf[(size_t)y0 * w + x0], tf[(size_t)y0 * w + x1], fx);
    float4 r1 = lerp4(tf[(size_t)y1 * w + x0], tf[(size_t)y1 * w + x1], fx);
    return lerp4(r0, r1, fy);
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
