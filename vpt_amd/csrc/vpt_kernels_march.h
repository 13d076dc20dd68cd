// vpt_kernels_march.h — the MIP, EAM and MCS pass kernels (gfx950, wave64); included by vpt_march.hip only.
#pragma once
#include "vpt_kernels.h"

// =============================================================================================
// MIP — MIPRenderer.glsl
// =============================================================================================
// generate/fragment main(): MIPRenderer.glsl:51-72; returns the unorm8 frame value
template <int V>
VPT_DEV uint32_t mip_pixel(const PassArgs &a, const LdsTables &t, const Pix &p, uint32_t &ns) {
    f3 rf, rt;
    unproject(ndc_col(a.pm, p.i), ndc_row(a.pm, p.j), a.mvp_inv, rf, rt);
    f3 dir = sub3(rt, rf);
    f2 tb = intersect_cube(rf, dir);
    tb.x = vmax(tb.x, 0.0f); tb.y = vmax(tb.y, 0.0f);
    float out = 0.0f;
    if (!(tb.x >= tb.y)) {
        f3 from = mix3(rf, rt, tb.x), to = mix3(rf, rt, tb.y);
        float tt = 0.0f, val = 0.0f, offset = a.offset;
        // The march is latency bound (LDS table -> brick line -> LDS transfer function per sample, ~1.5 us): the sample
        // positions do not depend on sampled values, so VPT_UNROLL_MIP samples are put in flight together.  The trip count
        // is still decided by the fp32 accumulation of t (do { ... } while (t < 1)); samples past the exit are fetched
        // speculatively at a valid position and discarded; max() is exact, so the grouping does not change the result.
        bool more = true;
        do {
            f3 pos[VPT_UNROLL_MIP]; bool act[VPT_UNROLL_MIP];
#pragma unroll
            for (int u = 0; u < VPT_UNROLL_MIP; u++) {
                act[u] = more;
                pos[u] = mix3(from, to, offset);
                if (more) {
                    tt += a.step;
                    float m = offset + a.step;
                    offset = m - floorf(m);
                    more = tt < 1.0f;
                }
            }
            float al[VPT_UNROLL_MIP];
#pragma unroll
            for (int u = 0; u < VPT_UNROLL_MIP; u++) al[u] = sample_volume_color<V>(a, t, pos[u]).w;
#pragma unroll
            for (int u = 0; u < VPT_UNROLL_MIP; u++) if (act[u]) { val = vmax(al[u], val); ns++; }
        } while (more);
        out = val;
    }
    return to_unorm8(out);
}
// MODE 0: _generateFrame only (frame <- value).  MODE 1: whole render(): generate, integrate
// (MIPRenderer.glsl:105-109, max on unorm8), renderFrame (:141-144) in one pass.
template <int MODE, int V>
__global__ void __launch_bounds__(VPT_BLOCK) VPT_WAVES_ATTR(VPT_MIP_WAVES) k_mip(PassArgs a) {
    apply_frame_table(a);
    extern __shared__ float4 lds_raw[];
    LdsTables t = stage_lds<(V & VPT_V_WIDE) != 0>(lds_raw, a);
    Pix p = map_pixel(a.pm);
    uint32_t ns = 0;
    if (p.valid) {
        uint8_t *frame = (uint8_t *)a.frame, *acc = (uint8_t *)a.acc;
        if (MODE == 0) {
            frame[p.k] = (uint8_t)mip_pixel<V>(a, t, p, ns);
        } else {
            uint32_t m = acc[p.k], base = a.frame_base;
            for (uint32_t f = 0, np = multi_pass_count(a); f < np; f++) {
                multi_pass_select(a, base, f);
                m = max(m, mip_pixel<V>(a, t, p, ns));
            }
            acc[p.k] = (uint8_t)m;
            float v = from_unorm8(m);
            store_frame(a, p, pack_half4(v, v, v, 1.0f));
        }
    }
    count_samples(a.samples, ns);
}
__global__ void __launch_bounds__(VPT_BLOCK) k_mip_integrate(PassArgs a) {
    Pix p = map_pixel(a.pm);
    if (!p.valid) return;
    uint8_t *frame = (uint8_t *)a.frame, *acc = (uint8_t *)a.acc;
    // max(acc, frame) on unorm8 values == integer max (c/255 is monotone)
    acc[p.k] = (uint8_t)max((uint32_t)acc[p.k], (uint32_t)frame[p.k]);
}
__global__ void __launch_bounds__(VPT_BLOCK) k_mip_render(PassArgs a) {
    Pix p = map_pixel(a.pm);
    if (!p.valid) return;
    float v = from_unorm8(((uint8_t *)a.acc)[p.k]);
    store_frame_texel(&a.render[(size_t)p.l * a.pm.W + p.i], pack_half4(v, v, v, 1.0f));
}
__global__ void __launch_bounds__(VPT_BLOCK) k_mip_reset(PassArgs a) {   // MIPRenderer.glsl:168-170
    Pix p = map_pixel(a.pm);
    if (p.tile) ((uint8_t *)a.acc)[p.k] = 0;
}

// =============================================================================================
// EAM — EAMRenderer.glsl
// =============================================================================================
// generate/fragment main(): EAMRenderer.glsl:52-80; returns packed RGBA8
template <int V>
VPT_DEV uint32_t eam_pixel(const PassArgs &a, const LdsTables &t, const Pix &p, uint32_t &ns) {
    f3 rf, rt;
    unproject(ndc_col(a.pm, p.i), ndc_row(a.pm, p.j), a.mvp_inv, rf, rt);
    f3 dir = sub3(rt, rf);
    f2 tb = intersect_cube(rf, dir);
    tb.x = vmax(tb.x, 0.0f); tb.y = vmax(tb.y, 0.0f);
    float ox = 0.0f, oy = 0.0f, oz = 0.0f;
    if (!(tb.x >= tb.y)) {
        f3 from = mix3(rf, rt, tb.x), to = mix3(rf, rt, tb.y);
        float ray_step_length = length3(sub3(from, to)) * a.step;
        float tt = a.step * a.offset;
        float ax = 0.0f, ay = 0.0f, az = 0.0f, aw = 0.0f;
        float kk = ray_step_length * a.extinction;
        // while (t < 1 && A.a < 0.99): the fetch positions depend only on t, so VPT_UNROLL samples are fetched together
        // (speculatively past early termination) and composited in order under the reference's per-sample condition.
        bool alive = true;
        while (alive) {
            float tq[VPT_UNROLL];
            tq[0] = tt;
#pragma unroll
            for (int u = 1; u < VPT_UNROLL; u++) tq[u] = tq[u - 1] + a.step;
            float4 c[VPT_UNROLL];
#pragma unroll
            for (int u = 0; u < VPT_UNROLL; u++) c[u] = sample_volume_color<V>(a, t, mix3(from, to, tq[u]));
#pragma unroll
            for (int u = 0; u < VPT_UNROLL; u++) {
                if (alive && tq[u] < 1.0f && aw < 0.99f) {
                    ns++;
                    float cw = c[u].w * kk;
                    float cx = c[u].x * cw, cy = c[u].y * cw, cz = c[u].z * cw;
                    float w = 1.0f - aw;
                    ax = fmaf(w, cx, ax); ay = fmaf(w, cy, ay); az = fmaf(w, cz, az); aw = fmaf(w, cw, aw);
                    tt = tq[u] + a.step;
                } else {
                    alive = false;
                }
            }
        }
        if (aw > 1.0f) { float ia = rcp_nr(aw); ax *= ia; ay *= ia; az *= ia; }
        ox = ax; oy = ay; oz = az;
    }
    return to_unorm8(ox) | (to_unorm8(oy) << 8) | (to_unorm8(oz) << 16) | (255u << 24);
}
// integrate: EAMRenderer.glsl:115-119, per channel, re-quantised to unorm8
VPT_DEV uint32_t eam_mix(uint32_t acc, uint32_t frame, float m) {
    uint32_t r = 0;
    for (int c = 0; c < 4; c++) {
        float av = from_unorm8((acc >> (8 * c)) & 0xffu), fv = from_unorm8((frame >> (8 * c)) & 0xffu);
        r |= to_unorm8(mixf(av, fv, m)) << (8 * c);
    }
    return r;
}
template <int MODE, int V>
__global__ void __launch_bounds__(VPT_BLOCK) VPT_WAVES_ATTR(VPT_EAM_WAVES) k_eam(PassArgs a) {
    apply_frame_table(a);
    extern __shared__ float4 lds_raw[];
    LdsTables t = stage_lds<(V & VPT_V_WIDE) != 0>(lds_raw, a);
    Pix p = map_pixel(a.pm);
    uint32_t ns = 0;
    if (p.valid) {
        uint32_t *frame = (uint32_t *)a.frame, *acc = (uint32_t *)a.acc;
        if (MODE == 0) {
            frame[p.k] = eam_pixel<V>(a, t, p, ns);
        } else {
            uint32_t m = acc[p.k], base = a.frame_base;
            for (uint32_t f = 0, np = multi_pass_count(a); f < np; f++) {
                multi_pass_select(a, base, f);
                m = eam_mix(m, eam_pixel<V>(a, t, p, ns), a.mix);
            }
            acc[p.k] = m;
            store_frame(a, p, eam_to_half4(m));
        }
    }
    count_samples(a.samples, ns);
}
__global__ void __launch_bounds__(VPT_BLOCK) k_eam_integrate(PassArgs a) {
    Pix p = map_pixel(a.pm);
    if (!p.valid) return;
    uint32_t *frame = (uint32_t *)a.frame, *acc = (uint32_t *)a.acc;
    acc[p.k] = eam_mix(acc[p.k], frame[p.k], a.mix);
}
__global__ void __launch_bounds__(VPT_BLOCK) k_eam_render(PassArgs a) {
    Pix p = map_pixel(a.pm);
    if (!p.valid) return;
    store_frame_texel(&a.render[(size_t)p.l * a.pm.W + p.i], eam_to_half4(((uint32_t *)a.acc)[p.k]));
}
__global__ void __launch_bounds__(VPT_BLOCK) k_eam_reset(PassArgs a) {   // EAMRenderer.glsl:177-179
    Pix p = map_pixel(a.pm);
    if (p.tile) ((uint32_t *)a.acc)[p.k] = 0xff000000u;
}

// =============================================================================================
// MCS — MCSRenderer.glsl
// =============================================================================================
// sampleDistance: MCSRenderer.glsl:70-87
template <int V>
VPT_DEV float mcs_sample_distance(const PassArgs &a, const LdsTables &t, uint32_t &state, f3 from, f3 to, uint32_t &ns) {
    float max_distance = length3(sub3(from, to));
    float inv_max = rcp_nr(max_distance);
    float dist = 0.0f;
    for (uint32_t it = 0; it < VPT_MAX_TRACK_ITERS; it++) {
        dist += random_exponential(state, a.inv_extinction);
        if (!(dist <= max_distance)) break;
        f3 p = mix3(from, to, dist * inv_max);
        float4 ts = sample_volume_color<V>(a, t, p);
        ns++;
        if (random_uniform(state) < ts.w) break;
    }
    return dist;
}
// sampleTransmittance: MCSRenderer.glsl:89-105
template <int V>
VPT_DEV float mcs_sample_transmittance(const PassArgs &a, const LdsTables &t, uint32_t &state, f3 from, f3 to, uint32_t &ns) {
    float max_distance = length3(sub3(from, to));
    float inv_max = rcp_nr(max_distance);
    float dist = 0.0f, tr = 1.0f;
    for (uint32_t it = 0; it < VPT_MAX_TRACK_ITERS; it++) {
        dist += random_exponential(state, a.inv_extinction);
        if (!(dist <= max_distance)) break;
        f3 p = mix3(from, to, dist * inv_max);
        float4 ts = sample_volume_color<V>(a, t, p);
        ns++;
        tr *= 1.0f - ts.w;
    }
    return tr;
}
// generate/fragment main(): MCSRenderer.glsl:107-137
template <int V>
VPT_DEV float4 mcs_pixel(const PassArgs &a, const LdsTables &t, const Pix &p, uint32_t &ns) {
    float px = ndc_col(a.pm, p.i), py = ndc_row(a.pm, p.j);
    f3 rf, rt;
    unproject(px, py, a.mvp_inv, rf, rt);
    f3 dir = sub3(rt, rf);
    f3 dir_unit = normalize3(dir);
    f2 tb = intersect_cube(rf, dir);
    tb.x = vmax(tb.x, 0.0f); tb.y = vmax(tb.y, 0.0f);
    if (tb.x >= tb.y) return sample_environment(a.env, dir_unit);
    f3 from = mix3(rf, rt, tb.x), to = mix3(rf, rt, tb.y);
    float max_distance = length3(sub3(from, to));
    uint32_t state = hash3(__float_as_uint(ndc_to_uv(px)), __float_as_uint(ndc_to_uv(py)), __float_as_uint(a.seed));
    float dist = mcs_sample_distance<V>(a, t, state, from, to, ns);
    if (!(dist <= max_distance)) return sample_environment(a.env, dir_unit);
    from = mix3(from, to, dist * rcp_nr(max_distance));
    f2 tb2 = intersect_cube(from, a.light);
    tb2.y = vmax(tb2.y, 0.0f);
    to = madd3(from, tb2.y, a.light);
    float4 diffuse = sample_volume_color<V>(a, t, from);
    ns++;
    float4 light = sample_environment(a.env, a.light);
    float tr = mcs_sample_transmittance<V>(a, t, state, from, to, ns);
    return make_float4((diffuse.x * light.x) * tr, (diffuse.y * light.y) * tr,
                       (diffuse.z * light.z) * tr, (diffuse.w * light.w) * tr);
}
VPT_DEV float4 mcs_mix(float4 acc, float4 frame, float inv) {   // MCSRenderer.glsl:173-177
    return make_float4(fmaf(frame.x - acc.x, inv, acc.x), fmaf(frame.y - acc.y, inv, acc.y),
                       fmaf(frame.z - acc.z, inv, acc.z), fmaf(frame.w - acc.w, inv, acc.w));
}
template <int MODE, int V>
__global__ void __launch_bounds__(VPT_BLOCK) VPT_WAVES_ATTR(VPT_MCS_WAVES) k_mcs(PassArgs a) {
    apply_frame_table(a);
    extern __shared__ float4 lds_raw[];
    LdsTables t = stage_lds<(V & VPT_V_WIDE) != 0>(lds_raw, a);
    Pix p = map_pixel(a.pm);
    uint32_t ns = 0;
    if (p.valid) {
        float4 *frame = (float4 *)a.frame, *acc = (float4 *)a.acc;
        if (MODE == 0) {
            frame[p.k] = mcs_pixel<V>(a, t, p, ns);
        } else {
            float4 m = acc[p.k];
            uint32_t base = a.frame_base;
            for (uint32_t f = 0, np = multi_pass_count(a); f < np; f++) {
                multi_pass_select(a, base, f);
                m = mcs_mix(m, mcs_pixel<V>(a, t, p, ns), a.mix);
            }
            acc[p.k] = m;
            store_frame(a, p, pack_half4(m.x, m.y, m.z, m.w));
        }
    }
    count_samples(a.samples, ns);
}
// ---- persistent-wave MCS with active-ray compaction ------------------------------------------------------------
// The tracking loops of MCSRenderer.glsl:70-105 have data-dependent lengths (0 .. extinction * chord events), so in the
// one-thread-per-pixel kernel above finished lanes idle until the longest ray of their wave ends and whole workgroups
// idle behind the image's heavy region (measured at extinction 200: 78 % lanes active, 43 % wave occupancy).
// Here waves are persistent: every lane is a small state machine (idle -> distance sampling -> shadow ray) that
// executes ONE tracking event per loop trip; when >= VPT_REFILL lanes of a wave are idle they are refilled by
// __ballot / __popcll compaction with the next pixels of the wave's current 8x8 tile.  Tiles are drawn from
// VPT_WORK_SHARDS atomic counters (shard c hands out tiles c, c + SHARDS, ...; a wave starts at its own shard and
// steals from the next ones when it runs dry) — ONE counter serialises at ~88 returning atomics per us, which alone
// cost 0.37 ms for the 32 k tiles of a 1080p frame.  A pixel's result depends only on its own seed, so the output is
// bit-identical to k_mcs.
#define VPT_REFILL 16
#define VPT_WORK_SHARDS 256
#define VPT_WORK_STRIDE 32      // uint32 words between shard counters (one 128-B line each)
struct McsLane {
    int phase;                  // 0 idle, 1 sampleDistance, 2 sampleTransmittance
    int i, l, k;                // pixel column, local row, tile-order buffer index
    uint32_t state, it;
    f3 from, to, dir_unit;
    float dist, maxd, invmax, tr;
    float4 diffuse;
};
VPT_DEV int pixel_buffer_index(const PixMap &m, int i, int l) {
    int t = (l >> 4) * m.tiles_x + (i >> 4);
    int w = ((i >> 3) & 1) | (((l >> 3) & 1) << 1);
    return t * VPT_BLOCK + w * 64 + ((i & 7) | ((l & 7) << 3));
}
template <int MODE>
VPT_DEV void mcs_write(const PassArgs &a, const McsLane &s, float4 c) {
    float4 *frame = (float4 *)a.frame, *acc = (float4 *)a.acc;
    if (MODE == 0) {
        frame[s.k] = c;
    } else {
        float4 m = mcs_mix(acc[s.k], c, a.mix);
        acc[s.k] = m;
        store_frame_texel(&a.render[(size_t)s.l * a.pm.W + s.i], pack_half4(m.x, m.y, m.z, m.w));
    }
}
template <int MODE, int V>
__global__ void __launch_bounds__(VPT_BLOCK) k_mcs_persist(PassArgs a, uint32_t *counter, int ntx8, int ntiles8) {
    extern __shared__ float4 lds_raw[];
    LdsTables t = stage_lds<(V & VPT_V_WIDE) != 0>(lds_raw, a);
    const int lane = (int)threadIdx.x & 63;
    McsLane s;
    s.phase = 0; s.i = s.l = s.k = 0; s.state = 0; s.it = 0;
    s.from = s.to = s.dir_unit = f3{ 0.0f, 0.0f, 0.0f };
    s.dist = s.maxd = s.invmax = 0.0f; s.tr = 1.0f; s.diffuse = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint32_t ns = 0;
    int cur_tile = -1, cur_off = 64;        // wave-uniform: the tile being handed out and its next unassigned pixel
    bool exhausted = false;                 // wave-uniform: every shard ran out
    int shard = (int)((blockIdx.x * 4u + (threadIdx.x >> 6)) % VPT_WORK_SHARDS), shards_tried = 0;   // wave-uniform
    const float4 light = sample_environment(a.env, a.light);

    for (;;) {
        unsigned long long idle = __ballot(s.phase == 0);
        int nidle = __popcll(idle);
        if (nidle == 64 && exhausted) break;
        if (!exhausted && (nidle >= VPT_REFILL)) {
            // ---- compaction: idle lane with rank r (among idle lanes) takes pixel cur_off + r of the pending tile(s)
            int rank = __popcll(idle & ((1ull << lane) - 1ull));
            int avail = 64 - cur_off;
            int tile1 = -1;
            if (nidle > avail) {
                while (shards_tried < VPT_WORK_SHARDS) {          // bounded: a wave leaves each shard at most once
                    uint32_t n = 0;
                    if (lane == 0) n = atomicAdd(counter + shard * VPT_WORK_STRIDE, 1u);
                    n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
                    uint32_t tnew = n * VPT_WORK_SHARDS + (uint32_t)shard;
                    if (tnew < (uint32_t)ntiles8) { tile1 = (int)tnew; break; }
                    shard = (shard + 1) % VPT_WORK_SHARDS; shards_tried++;
                }
            }
            if (s.phase == 0) {
                int idx = cur_off + rank, tile = cur_tile;
                if (idx >= 64) { idx -= 64; tile = tile1; }
                if (tile >= 0) {
                    int ty = tile / ntx8, tx = tile - ty * ntx8;
                    int i = tx * 8 + (idx & 7), l = ty * 8 + (idx >> 3);
                    int j = global_row(a.pm, l);
                    if (i < a.pm.W && l < a.pm.local_h && j < a.pm.H) {
                        // generate/fragment main() up to the first tracking loop: MCSRenderer.glsl:107-122
                        s.i = i; s.l = l; s.k = pixel_buffer_index(a.pm, i, l);
                        float px = ndc_col(a.pm, i), py = ndc_row(a.pm, j);
                        f3 rf, rt;
                        unproject(px, py, a.mvp_inv, rf, rt);
                        f3 dir = sub3(rt, rf);
                        s.dir_unit = normalize3(dir);
                        f2 tb = intersect_cube(rf, dir);
                        tb.x = vmax(tb.x, 0.0f); tb.y = vmax(tb.y, 0.0f);
                        if (tb.x >= tb.y) {
                            mcs_write<MODE>(a, s, sample_environment(a.env, s.dir_unit));
                        } else {
                            s.from = mix3(rf, rt, tb.x); s.to = mix3(rf, rt, tb.y);
                            s.maxd = length3(sub3(s.from, s.to));
                            s.invmax = rcp_nr(s.maxd);
                            s.state = hash3(__float_as_uint(ndc_to_uv(px)), __float_as_uint(ndc_to_uv(py)), __float_as_uint(a.seed));
                            s.dist = 0.0f; s.it = 0; s.phase = 1;
                        }
                    }
                }
            }
            if (nidle > avail) {
                if (tile1 >= 0) { cur_tile = tile1; cur_off = nidle - avail; }
                else { cur_tile = -1; cur_off = 64; exhausted = true; }
            } else {
                cur_off += nidle;
            }
        }
        // ---- one tracking event for every active lane (both loops share the body)
        if (s.phase != 0) {
            s.dist += random_exponential(s.state, a.inv_extinction);
            if (!(s.dist <= s.maxd)) {
                float4 c;
                if (s.phase == 1) c = sample_environment(a.env, s.dir_unit);                       // MCSRenderer.glsl:124-127
                else c = make_float4((s.diffuse.x * light.x) * s.tr, (s.diffuse.y * light.y) * s.tr,
                                     (s.diffuse.z * light.z) * s.tr, (s.diffuse.w * light.w) * s.tr);   // :136
                mcs_write<MODE>(a, s, c);
                s.phase = 0;
            } else {
                f3 p = mix3(s.from, s.to, s.dist * s.invmax);
                float4 ts = sample_volume_color<V>(a, t, p);
                ns++;
                bool last = (s.it == VPT_MAX_TRACK_ITERS - 1u);
                s.it++;
                if (s.phase == 1) {
                    bool accept = random_uniform(s.state) < ts.w;
                    if (accept || last) {
                        // scatter point: MCSRenderer.glsl:129-135
                        f2 tb2 = intersect_cube(p, a.light);
                        tb2.y = vmax(tb2.y, 0.0f);
                        s.diffuse = sample_volume_color<V>(a, t, p);
                        ns++;
                        s.from = p;
                        s.to = madd3(p, tb2.y, a.light);
                        s.maxd = length3(sub3(s.from, s.to));
                        s.invmax = rcp_nr(s.maxd);
                        s.dist = 0.0f; s.tr = 1.0f; s.it = 0; s.phase = 2;
                    }
                } else {
                    s.tr *= 1.0f - ts.w;
                    if (last) {
                        mcs_write<MODE>(a, s, make_float4((s.diffuse.x * light.x) * s.tr, (s.diffuse.y * light.y) * s.tr,
                                                          (s.diffuse.z * light.z) * s.tr, (s.diffuse.w * light.w) * s.tr));
                        s.phase = 0;
                    }
                }
            }
        }
    }
    count_samples(a.samples, ns);
}
__global__ void __launch_bounds__(VPT_BLOCK) k_mcs_integrate(PassArgs a) {
    Pix p = map_pixel(a.pm);
    if (!p.valid) return;
    float4 *frame = (float4 *)a.frame, *acc = (float4 *)a.acc;
    acc[p.k] = mcs_mix(acc[p.k], frame[p.k], a.mix);
}
__global__ void __launch_bounds__(VPT_BLOCK) k_mcs_render(PassArgs a) {   // MCSRenderer.glsl:210-213
    Pix p = map_pixel(a.pm);
    if (!p.valid) return;
    float4 m = ((float4 *)a.acc)[p.k];
    store_frame_texel(&a.render[(size_t)p.l * a.pm.W + p.i], pack_half4(m.x, m.y, m.z, m.w));
}
__global__ void __launch_bounds__(VPT_BLOCK) k_mcs_reset(PassArgs a) {    // MCSRenderer.glsl:238-240
    Pix p = map_pixel(a.pm);
    if (p.tile) ((float4 *)a.acc)[p.k] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
}

