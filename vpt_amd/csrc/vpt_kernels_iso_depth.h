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
template <int MODE, int V>
__global__ void __launch_bounds__(VPT_BLOCK) k_iso(PassArgs a) {
    apply_frame_table(a);
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
            acc[p.k] = m;
            a.render[(size_t)p.l * a.pm.W + p.i] = iso_shade<V>(a, t, m, ns);
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
__global__ void __launch_bounds__(VPT_BLOCK) k_depth(PassArgs a) {
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
            uint32_t base = a.multi_passes > 1u ? *a.frame_counter : 0u;
            for (uint32_t f = 0, np = multi_pass_count(a); f < np; f++) {
                multi_pass_select(a, base, f);
                m = mixf(m, depth_pixel<V>(a, t, p, ns), a.mix);     // DepthRenderer.glsl:114-118
            }
            acc[p.k] = m;
            a.render[(size_t)p.l * a.pm.W + p.i] = pack_half4(m, m, m, 1.0f);    // :150-153
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
