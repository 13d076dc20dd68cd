// vpt_kernels_iso_depth.h — the ISO and Depth renderers on the same launch shape,
// sampler and buffers-in-thread-order scheme as vpt_kernels.h.
//   ISO   : src/glsl/renderers/ISORenderer.glsl   — frame / accumulation RGBA16F "closest hit" (xyz, t) (ISORenderer.js:165-197)
//   Depth : src/glsl/renderers/DepthRenderer.glsl — frame / accumulation R32F (DepthRenderer.js:165-189)
#pragma once
#include "vpt_kernels.h"

VPT_DEV float half_lo(uint32_t w) { return __half2float(__ushort_as_half((unsigned short)(w & 0xffffu))); }
VPT_DEV float half_hi(uint32_t w) { return __half2float(__ushort_as_half((unsigned short)(w >> 16))); }

// =============================================================================================
// ISO
// =============================================================================================
// generate/fragment main(): ISORenderer.glsl:52-76 — back-to-front march; the last hit written is the closest one.
// a.steps = uSteps, a.step = fl(1 / float(uSteps)).  Returns the RGBA16F texel (position, t) or (-1,-1,-1,-1).
template <int V>
VPT_DEV uint2 iso_pixel(const PassArgs &a, const LdsTables &t, const Pix &p, uint32_t &ns) {
    f3 rf, rt;
    unproject(ndc_col(a.pm, p.i), ndc_row(a.pm, p.j), a.mvp_inv, rf, rt);
    f3 dir = sub3(rt, rf);
    f2 tb = intersect_cube(rf, dir);
    tb.x = vmax(tb.x, 0.0f); tb.y = vmax(tb.y, 0.0f);
    float cx = -1.0f, cy = -1.0f, cz = -1.0f, cw = -1.0f;
    if (!(tb.x >= tb.y)) {
        f3 from = mix3(rf, rt, tb.x), to = mix3(rf, rt, tb.y);
        float tt = 1.0f - a.offset * a.step;
        // sample positions do not depend on sampled values: VPT_UNROLL of them are in flight together (as in the MIP march)
        uint32_t s = 0;
        while (s < a.steps) {
            f3 pos[VPT_UNROLL]; float tv[VPT_UNROLL], al[VPT_UNROLL];
#pragma unroll
            for (int u = 0; u < VPT_UNROLL; u++) { tv[u] = tt; pos[u] = mix3(from, to, tt); tt -= a.step; }
#pragma unroll
            for (int u = 0; u < VPT_UNROLL; u++) al[u] = sample_volume_color<V>(a, t, pos[u]).w;
#pragma unroll
            for (int u = 0; u < VPT_UNROLL; u++) {
                if (s + (uint32_t)u < a.steps) {
                    ns++;
                    if (al[u] >= a.isovalue) { cx = pos[u].x; cy = pos[u].y; cz = pos[u].z; cw = tv[u]; }
                }
            }
            s += VPT_UNROLL;
        }
    }
    return pack_half4(cx, cy, cz, cw);
}
// integrate: ISORenderer.glsl:111-121 — keep the hit with the smaller positive t (texels copied bit for bit)
VPT_DEV uint2 iso_closer(uint2 acc, uint2 frame) {
    float fw = half_hi(frame.y), aw = half_hi(acc.y);
    bool take_frame = (fw > 0.0f && aw > 0.0f) ? (fw < aw) : (fw > 0.0f);
    return take_frame ? frame : acc;
}
// render/fragment main(): ISORenderer.glsl:179-191 with gradient() :165-177.  a.light = uLight, a.gradient_step = uGradientStep
template <int V>
VPT_DEV uint2 iso_shade(const PassArgs &a, const LdsTables &t, uint2 closest, uint32_t &ns) {
    if (!(half_hi(closest.y) > 0.0f)) return pack_half4(1.0f, 1.0f, 1.0f, 1.0f);
    f3 pos = { half_lo(closest.x), half_hi(closest.x), half_lo(closest.y) };
    float h = a.gradient_step;
    float px = sample_volume_color<V>(a, t, f3{ pos.x + h, pos.y, pos.z }).w;
    float py = sample_volume_color<V>(a, t, f3{ pos.x, pos.y + h, pos.z }).w;
    float pz = sample_volume_color<V>(a, t, f3{ pos.x, pos.y, pos.z + h }).w;
    float nx = sample_volume_color<V>(a, t, f3{ pos.x - h, pos.y, pos.z }).w;
    float ny = sample_volume_color<V>(a, t, f3{ pos.x, pos.y - h, pos.z }).w;
    float nz = sample_volume_color<V>(a, t, f3{ pos.x, pos.y, pos.z - h }).w;
    float d = 2.0f * h;
    f3 normal = normalize3(f3{ (px - nx) / d, (py - ny) / d, (pz - nz) / d });
    float lambert = vmax(dot3(normal, a.light), 0.0f);
    float4 material = sample_volume_color<V>(a, t, pos);
    ns += 7;
    return pack_half4(material.x * lambert, material.y * lambert, material.z * lambert, 1.0f);
}
// MODE 0: _generateFrame only.  MODE 1: the whole render(): generate, integrate, renderFrame in one pass.
#ifndef VPT_ISO_WAVES
#define VPT_ISO_WAVES 7
#endif
template <int MODE, int V>
__global__ void __launch_bounds__(VPT_BLOCK) __attribute__((amdgpu_waves_per_eu(VPT_ISO_WAVES, 8))) k_iso(PassArgs a) {
    if (a.multi_passes > 1u) multi_pass_select(a, a.frame_base, 0); else apply_frame_table(a);
    extern __shared__ float4 lds_raw[];
    LdsTables t = stage_lds<(V & VPT_V_WIDE) != 0>(lds_raw, a);
    Pix p = map_pixel(a.pm);
    uint32_t ns = 0;
    if (p.valid) {
        uint2 q = iso_pixel<V>(a, t, p, ns);
        uint2 *frame = (uint2 *)a.frame, *acc = (uint2 *)a.acc;
        if (MODE == 0) {
            frame[p.k] = q;
        } else {
            uint2 m = iso_closer(acc[p.k], q);
            // VPT_PLAY_FUSED: the remaining passes of the sequence; every pass but the last one only leaves its 7 shading
            // samples in the counter (its render buffer is overwritten by the next pass anyway)
            if (a.multi_passes > 1u) {
                uint32_t base = a.frame_base;
                for (uint32_t f = 1; f < a.multi_passes; f++) {
                    if (half_hi(m.y) > 0.0f) ns += 7;
                    multi_pass_select(a, base, f);
                    m = iso_closer(m, iso_pixel<V>(a, t, p, ns));
                }
            }
            acc[p.k] = m;
            store_frame(a, p, iso_shade<V>(a, t, m, ns));
        }
    }
    count_samples(a.samples, ns);
}
__global__ void __launch_bounds__(VPT_BLOCK) k_iso_integrate(PassArgs a) {
    Pix p = map_pixel(a.pm);
    if (!p.valid) return;
    uint2 *frame = (uint2 *)a.frame, *acc = (uint2 *)a.acc;
    acc[p.k] = iso_closer(acc[p.k], frame[p.k]);
}
template <int V>
__global__ void __launch_bounds__(VPT_BLOCK) k_iso_render(PassArgs a) {
    extern __shared__ float4 lds_raw[];
    LdsTables t = stage_lds<(V & VPT_V_WIDE) != 0>(lds_raw, a);
    Pix p = map_pixel(a.pm);
    uint32_t ns = 0;
    if (p.valid) a.render[(size_t)p.l * a.pm.W + p.i] = iso_shade<V>(a, t, ((const uint2 *)a.acc)[p.k], ns);
    count_samples(a.samples, ns);
}
__global__ void __launch_bounds__(VPT_BLOCK) k_iso_reset(PassArgs a) {    // ISORenderer.glsl:215-217: vec4(-1)
    Pix p = map_pixel(a.pm);
    if (p.tile) ((uint2 *)a.acc)[p.k] = make_uint2(0xbc00bc00u, 0xbc00bc00u);
}

// =============================================================================================
// Depth
// =============================================================================================
// generate/fragment main(): DepthRenderer.glsl:53-79
template <int V>
VPT_DEV float depth_pixel(const PassArgs &a, const LdsTables &t, const Pix &p, uint32_t &ns) {
    f3 rf, rt;
    unproject(ndc_col(a.pm, p.i), ndc_row(a.pm, p.j), a.mvp_inv, rf, rt);
    f3 dir = sub3(rt, rf);
    f2 tb = intersect_cube(rf, dir);
    tb.x = vmax(tb.x, 0.0f); tb.y = vmax(tb.y, 0.0f);
    float depth = -1.0f;
    if (!(tb.x >= tb.y)) {
        f3 from = mix3(rf, rt, tb.x), to = mix3(rf, rt, tb.y);
        float ray_step_length = length3(sub3(from, to)) * a.step;
        float tt = a.step * a.offset;
        float accumulator = 0.0f;
        while (tt < 1.0f && accumulator < a.threshold) {
            float al = sample_volume_color<V>(a, t, mix3(from, to, tt)).w;
            ns++;
            accumulator += (1.0f - accumulator) * al * ray_step_length * a.extinction;
            tt += a.step;
        }
        if (!(accumulator < a.threshold)) depth = mixf(tb.x, tb.y, tt);
    }
    return depth;
}
template <int MODE, int V>
__global__ void __launch_bounds__(VPT_BLOCK) VPT_WAVES_ATTR(VPT_DEPTH_WAVES) k_depth(PassArgs a) {
    apply_frame_table(a);
    extern __shared__ float4 lds_raw[];
    LdsTables t = stage_lds<(V & VPT_V_WIDE) != 0>(lds_raw, a);
    Pix p = map_pixel(a.pm);
    uint32_t ns = 0;
    if (p.valid) {
        float *frame = (float *)a.frame, *acc = (float *)a.acc;
        if (MODE == 0) {
            frame[p.k] = depth_pixel<V>(a, t, p, ns);
        } else {
            float m = acc[p.k];
            uint32_t base = a.frame_base;
            for (uint32_t f = 0, np = multi_pass_count(a); f < np; f++) {
                multi_pass_select(a, base, f);
                m = mixf(m, depth_pixel<V>(a, t, p, ns), a.mix);     // DepthRenderer.glsl:114-118
            }
            acc[p.k] = m;
            store_frame(a, p, pack_half4(m, m, m, 1.0f));    // :150-153
        }
    }
    count_samples(a.samples, ns);
}
__global__ void __launch_bounds__(VPT_BLOCK) k_depth_integrate(PassArgs a) {
    Pix p = map_pixel(a.pm);
    if (!p.valid) return;
    float *frame = (float *)a.frame, *acc = (float *)a.acc;
    acc[p.k] = mixf(acc[p.k], frame[p.k], a.mix);
}
__global__ void __launch_bounds__(VPT_BLOCK) k_depth_render(PassArgs a) {
    Pix p = map_pixel(a.pm);
    if (!p.valid) return;
    float m = ((const float *)a.acc)[p.k];
    a.render[(size_t)p.l * a.pm.W + p.i] = pack_half4(m, m, m, 1.0f);
}
__global__ void __launch_bounds__(VPT_BLOCK) k_depth_reset(PassArgs a) {  // DepthRenderer.glsl:177-179: vec4(0,0,0,1) into R32F
    Pix p = map_pixel(a.pm);
    if (p.tile) ((float *)a.acc)[p.k] = 0.0f;
}

// =============================================================================================
// LAO — src/glsl/renderers/LAORenderer.glsl.  An experimental shader of the reference: its
// "random" numbers are rand(vPosition * seed) with a constant seed — one fixed value per pixel — and frames do not
// accumulate (integrate copies the frame).  frame / accumulation: RGBA8 (LAORenderer.js:217-243).
// =============================================================================================
// mixins/rand.glsl:3-13 (mat2 is column-major; the dot products are fma chains, first column first)
VPT_DEV f2 lao_rand(float px, float py) {
    const float m00 = 23.14069263277926f, m01 = 2.665144142690225f, m10 = 12.98987893203892f, m11 = 78.23376739376591f;
    float dx = fmaf(m10, py, m00 * px), dy = fmaf(m11, py, m01 * px);
    float s, c, s2, c2;
    vpt_sincosf(dx, s, c); vpt_sincosf(dy, s2, c2);
    float ax = c * 1235.6789f, ay = s2 * 4378.5453f;
    return f2{ ax - floorf(ax), ay - floorf(ay) };
}
// generate/fragment main(): LAORenderer.glsl:97-191; vLight (:25) = (M^-1 * (uLightPosition, 1)).xyz, no divide
// pow(1 - u, 2) of the occlusion march depends on the tap index alone (u = 0.001, += uLAOStepSize in fp32): a workgroup
// evaluates the first LAO_POW_TABLE of them once into LDS instead of a log + exp per tap (about a third of the kernel's
// instructions); taps beyond the table (a step below 1/256) evaluate it in place.
#define LAO_POW_TABLE 256
template <int V>
VPT_DEV uint32_t lao_pixel(const PassArgs &a, const LdsTables &t, const Pix &p, uint32_t &ns, const float *pow_table) {
    float px = ndc_col(a.pm, p.i), py = ndc_row(a.pm, p.j);
    f3 rf, rt;
    unproject(px, py, a.mvp_inv, rf, rt);
    f3 dir = sub3(rt, rf);
    f2 tb = intersect_cube(rf, dir);
    tb.x = vmax(tb.x, 0.0f); tb.y = vmax(tb.y, 0.0f);
    float ox = 0.0f, oy = 0.0f, oz = 0.0f;
    if (!(tb.x >= tb.y)) {
        const LaoParams &lp = a.lao;
        float4 lh = mat4_mul_point(a.mvp_inv, lp.light_position[0], lp.light_position[1], lp.light_position[2]);
        const f3 vl = { lh.x, lh.y, lh.z };
        const float vs = 1.0f / 32.0f;                                   // voxelSize, :59
        const float rs = lao_rand(3.14f, 2.71f).x;                        // rand(seed).x
        f3 from = mix3(rf, rt, tb.x), to = mix3(rf, rt, tb.y);
        const float R = lao_rand(px * 3.14f, py * 2.71f).x;               // every rand(vPosition * seed) of the shader
        float tt = vclamp01((R * a.step) * 1.5f);
        float ax = 0.0f, ay = 0.0f, az = 0.0f, aw = 0.0f;
        while (tt < 1.0f && aw < 0.99f) {
            if (aw > 0.98f) break;
            f3 pos = mix3(from, to, tt);
            tt += a.step;
            f3 grad;
            grad.x = sample_volume<V>(a.vol, t, f3{ pos.x - vs, pos.y, pos.z }) - sample_volume<V>(a.vol, t, f3{ pos.x + vs, pos.y, pos.z });
            grad.y = sample_volume<V>(a.vol, t, f3{ pos.x, pos.y - vs, pos.z }) - sample_volume<V>(a.vol, t, f3{ pos.x, pos.y + vs, pos.z });
            grad.z = sample_volume<V>(a.vol, t, f3{ pos.x, pos.y, pos.z - vs }) - sample_volume<V>(a.vol, t, f3{ pos.x, pos.y, pos.z + vs });
            float value = sample_volume<V>(a.vol, t, pos);
            ns += 7;
            float lao = 0.0f, soft = 0.0f;
            if (lp.local_ambient_occlusion) {
                float acc = 0.0f;                                         // accumuLAOContribution: not reset between samples
                for (int samp = 0; samp < lp.num_lao_samples; samp++) {
                    int tap = 0;
                    for (float u = 0.001f; u < 1.0f; u += lp.lao_step_size, tap++) {
                        float rc = -1.0f + 2.0f * R;
                        f3 rd = normalize3(f3{ rc, rc, rc });
                        rd = f3{ rd.x * R, rd.y * R, rd.z * R };
                        float m = mixf(0.0f, lp.light_radius, u);
                        f3 hv = normalize3(f3{ (vl.x + rd.x * m) - pos.x, (vl.y + rd.y * m) - pos.y, (vl.z + rd.z * m) - pos.z });
                        f3 sp = { pos.x + hv.x * u, pos.y + hv.y * u, pos.z + hv.z * u };
                        acc += sample_volume<V>(a.vol, t, sp) * (tap < LAO_POW_TABLE ? pow_table[tap] : vpt_powf(1.0f - u, 2.0f));
                        ns++;
                        if (!(lp.lao_step_size > 0.0f)) break;             // a zero step would never end: one sample then
                    }
                    acc /= lp.light_coefficient;
                    acc = vclamp01(acc);
                    lao += acc;
                }
                lao /= (float)lp.num_lao_samples;
            }
            if (lp.soft_shadows) {
                float acc = 0.0f;
                // every shadow "sample" of the shader is the same one (its random direction is a per-pixel constant): the
                // position is fetched once (the shader samples it twice per iteration — counted) and the loop only accumulates
                f3 rd = normalize3(f3{ -1.0f + vl.x * R, vl.y + R * vl.z, -1.0f + 2.0f * rs });
                rd = f3{ rd.x * R, rd.y * R, rd.z * R };
                f3 sp = { pos.x + rd.x * lp.light_radius, pos.y + rd.y * lp.light_radius, pos.z + rd.z * lp.light_radius };
                const float v1 = sample_volume<V>(a.vol, t, sp);
                const float term = (v1 * (v1 * 0.2f)) * vpt_powf(length3(rd), 1.0f);
                for (int samp = 0; samp < lp.num_shadow_samples; samp++) { acc += term; ns += 2; }
                acc = vpt_powf(acc, 1.0f);
                acc /= (float)lp.num_shadow_samples;
                acc *= 20.0f;
                acc = vclamp01(acc);
                soft = mixf(1.0f - soft, acc, 1.2f);
            }
            soft /= 1.3f;
            soft = vclamp01(soft);
            float4 c = sample_tf2d(a.tf, a.tf_w, a.tf_h, value, length3(grad));   // getColor(value, gradientMagnitude(grad))
            float w1 = lao * lp.lao_weight, w2 = soft * lp.shadows_weight;
            c = make_float4(mixf(c.x, c.x * 0.15f, w1), mixf(c.y, c.y * 0.18f, w1), mixf(c.z, c.z * 0.32f, w1), mixf(c.w, c.w * 1.0f, w1));
            c = make_float4(mixf(c.x, c.x * 0.15f, w2), mixf(c.y, c.y * 0.18f, w2), mixf(c.z, c.z * 0.22f, w2), mixf(c.w, c.w * 1.0f, w2));
            float k = 1.0f - aw;
            ax += (k * c.x) * value; ay += (k * c.y) * value; az += (k * c.z) * value;
            aw += ((k * value) * a.extinction) / 100.0f;
            if (aw > 0.9f) break;
        }
        if (aw > 1.0f) { ax /= aw; ay /= aw; az /= aw; }
        ox = ax; oy = ay; oz = az;
    }
    return to_unorm8(ox) | (to_unorm8(oy) << 8) | (to_unorm8(oz) << 16) | (to_unorm8(1.0f) << 24);
}
// MODE 0: _generateFrame only.  MODE 1: the whole render(): generate, integrate (copy, LAORenderer.glsl:225-227), renderFrame
template <int MODE, int V>
__global__ void __launch_bounds__(VPT_BLOCK) VPT_WAVES_ATTR(VPT_LAO_WAVES) k_lao(PassArgs a) {
    extern __shared__ float4 lds_raw[];
    __shared__ float pow_table[LAO_POW_TABLE];
    if (threadIdx.x < 64) {                                   // lane l evaluates taps l, l + 64, ...: every lane walks the same u chain
        int tap = 0;
        for (float u = 0.001f; u < 1.0f && tap < LAO_POW_TABLE; u += a.lao.lao_step_size, tap++)
            if ((tap & 63) == (int)threadIdx.x) pow_table[tap] = vpt_powf(1.0f - u, 2.0f);
    }
    LdsTables t = stage_lds<(V & VPT_V_WIDE) != 0>(lds_raw, a);      // (its barrier also publishes pow_table)
    Pix p = map_pixel(a.pm);
    uint32_t ns = 0;
    if (p.valid) {
        uint32_t q = lao_pixel<V>(a, t, p, ns, pow_table);
        if (MODE == 0) {
            ((uint32_t *)a.frame)[p.k] = q;
        } else {
            ((uint32_t *)a.acc)[p.k] = q;
            store_frame(a, p, eam_to_half4(q));
        }
    }
    count_samples(a.samples, ns);
}
// reset / render: the shaders of the EAM renderer (LAORenderer.glsl:285-287: (0, 0, 0, 1) into RGBA8; :259-261)
__global__ void __launch_bounds__(VPT_BLOCK) k_lao_reset(PassArgs a) {
    Pix p = map_pixel(a.pm);
    if (p.tile) ((uint32_t *)a.acc)[p.k] = 0xff000000u;
}
__global__ void __launch_bounds__(VPT_BLOCK) k_lao_render(PassArgs a) {
    Pix p = map_pixel(a.pm);
    if (!p.valid) return;
    store_frame_texel(&a.render[(size_t)p.l * a.pm.W + p.i], eam_to_half4(((uint32_t *)a.acc)[p.k]));
}
__global__ void __launch_bounds__(VPT_BLOCK) k_lao_integrate(PassArgs a) {
    Pix p = map_pixel(a.pm);
    if (p.valid) ((uint32_t *)a.acc)[p.k] = ((const uint32_t *)a.frame)[p.k];
}

// =============================================================================================
// DOS — DOSRenderer.glsl: one launch per view-aligned slice; each reads the previous slice's colour
// (RGBA32F, own texel: st0, in place) and occlusion (R32F, LINEAR / REPEAT: neighbour reads, hence separate in / out buffers
// st2 / st3 and ROW-MAJOR layout instead of the tile order of the other renderers) and writes the next ones (DOSRenderer.js:240-259)
// =============================================================================================
// LINEAR / REPEAT tap pair at texel coordinate u = s*n - 0.5; contract: a coordinate that is NaN or beyond 1e9 texels reads
// texel 0 (oracle repeat_coord)
VPT_DEV void repeat_taps_u(float u, int n, int &i0, int &i1, float &f) {
    if (!(fabsf(u) < 1.0e9f)) u = 0.0f;
    float fl = floorf(u);
    f = u - fl;
    int i = (int)fl;
    if ((unsigned)i >= (unsigned)n) {           // outside the image: wrap (an integer division, ~40 instructions — kept off the common path)
        i %= n;
        if (i < 0) i += n;
    }
    i0 = i; i1 = (i + 1 == n) ? 0 : i + 1;
}
// integrate/fragment main(): DOSRenderer.glsl:73-89; vertex :17-23 (vPosition3D = the unprojected (position, uDepth))
template <int V>
__global__ void __launch_bounds__(VPT_BLOCK) VPT_WAVES_ATTR(VPT_DOS_WAVES) k_dos_slice(PassArgs a) {
    extern __shared__ float4 lds_raw[];
    // plain 2-D grid over the tiles of the launch rectangle (row-major buffers: no tile order, no XCD interleave to keep)
    const DosParams &d = a.dos;
    Pix p;
    p.i = (d.tile_x0 + (int)blockIdx.x) * VPT_TILE + ((int)threadIdx.x & (VPT_TILE - 1));
    p.j = (d.tile_y0 + (int)blockIdx.y) * VPT_TILE + ((int)threadIdx.x / VPT_TILE);
    p.valid = p.i < a.pm.W && p.j < a.pm.H;
    const int W = a.pm.W, H = a.pm.H;
    const size_t k = (size_t)p.j * W + p.i;
    const float *occ_in = (const float *)a.st2;
    float px = 0.0f, py = 0.0f, prev_occ = 0.0f;
    float4 prev = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    f3 pos = { -1.0f, -1.0f, -1.0f };
    bool in_volume = false;
    if (p.valid) {
        px = ndc_col(a.pm, p.i); py = ndc_row(a.pm, p.j);
        prev_occ = occ_in[k];
        prev = a.st0[k];                          // in flight together with the occlusion texel (both come from beyond the L2 after a kernel boundary)
        pos = dehomogenize(mat4_mul_point(a.mvp_inv, px, py, d.depth));
        in_volume = !(pos.x > 1.0f || pos.y > 1.0f || pos.z > 1.0f || pos.x < 0.0f || pos.y < 0.0f || pos.z < 0.0f);
    }
    // a tile the slice does not touch copies its occlusion and is done: no table staging, no sampling (at either end of the
    // sweep that is every tile of the rectangle)
    if (!__syncthreads_or(in_volume)) {
        if (p.valid) ((float *)a.st3)[k] = prev_occ;
        return;
    }
    LdsTables t = stage_lds<(V & VPT_V_WIDE) != 0>(lds_raw, a);
    uint32_t ns = 0;
    if (p.valid) {
        float oo = prev_occ;
        // a pixel outside the volume keeps its colour: the colour buffer is updated IN PLACE (a pass reads only the
        // pixel's own colour texel, so the reference's ping-pong is not needed for it), and such pixels — most of the
        // image with the default camera — move 8 bytes instead of 40
        if (in_volume) {
            float4 oc;
            float4 ts = sample_volume_color<V>(a, t, pos);
            ns = 1;
            float ext = ts.w * a.extinction;
            float e = vpt_expf((-ext) * a.step);
            float alpha = 1.0f - e;
            float k1 = 1.0f - prev.w;
            oc.x = prev.x + ((ts.x * prev_occ) * alpha) * k1;
            oc.y = prev.y + ((ts.y * prev_occ) * alpha) * k1;
            oc.z = prev.z + ((ts.z * prev_occ) * alpha) * k1;
            oc.w = vmin(prev.w + alpha, 1.0f);
            float uvx = ndc_to_uv(px), uvy = ndc_to_uv(py);
            float o = 0.0f;                                        // calculateOcclusion: :61-70
            const float fW = (float)W, fH = (float)H;
#pragma unroll 4
            for (int q = 0; q < d.nsamples; q++) {                 // (a batch of 8 taps in flight costs 126 VGPRs and half the waves: slower)
                float2 off = d.samples[q];
                int x0, x1, y0, y1; float fx, fy;
                repeat_taps_u(fmaf(uvx + off.x * d.scale_x, fW, -0.5f), W, x0, x1, fx);
                repeat_taps_u(fmaf(uvy + off.y * d.scale_y, fH, -0.5f), H, y0, y1, fy);
                o += lerpf(lerpf(occ_in[(size_t)y0 * W + x0], occ_in[(size_t)y0 * W + x1], fx),
                           lerpf(occ_in[(size_t)y1 * W + x0], occ_in[(size_t)y1 * W + x1], fx), fy);
            }
            oo = (o / (float)d.nsamples) * e;
            // an opaque pixel (prev.a == 1) gets its old colour back bit for bit: the store is skipped then
            if (__float_as_uint(oc.x) != __float_as_uint(prev.x) || __float_as_uint(oc.y) != __float_as_uint(prev.y) ||
                __float_as_uint(oc.z) != __float_as_uint(prev.z) || __float_as_uint(oc.w) != __float_as_uint(prev.w)) a.st0[k] = oc;
        }
        ((float *)a.st3)[k] = oo;
    }
    count_samples(a.samples, ns);
}
// reset: DOSRenderer.glsl:137-144.  BOTH occlusion buffers are set: a pixel outside the launch rectangle is never
// written again, and it reads 1.0 whichever of the two is current
__global__ void __launch_bounds__(VPT_BLOCK) k_dos_reset(PassArgs a) {
    Pix p = map_pixel(a.pm);
    if (!p.valid) return;
    size_t k = (size_t)p.j * a.pm.W + p.i;
    a.st0[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    ((float *)a.st2)[k] = 1.0f;
    ((float *)a.st3)[k] = 1.0f;
}
// render: DOSRenderer.glsl:113-116
__global__ void __launch_bounds__(VPT_BLOCK) k_dos_render(PassArgs a) {
    Pix p = map_pixel(a.pm);
    if (!p.valid) return;
    float4 c = a.st0[(size_t)p.j * a.pm.W + p.i];
    a.render[(size_t)p.l * a.pm.W + p.i] = pack_half4(mixf(1.0f, c.x, c.w), mixf(1.0f, c.y, c.w), mixf(1.0f, c.z, c.w), mixf(1.0f, 1.0f, c.w));
}
