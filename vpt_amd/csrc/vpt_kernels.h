// vpt_kernels.h — the per-pixel pass kernels (gfx950, wave64).
//
// Launch shape: one thread per pixel, 256-thread workgroups covering a 16x16 pixel
// tile as 2x2 waves of 8x8 pixels (neighbouring rays of a wave walk the same bricks).  Workgroup ids
// are remapped so that the blocks that share an XCD (blockIdx % 8) own the tile diagonals (tx + ty) % 8
// (map_pixel below): equal work per XCD for any image or shard shape.  Per-pixel buffers (frame, accumulation, MCM photon
// state) are stored in THREAD order (tile-major), so every wave access is one contiguous 64*size
// segment; only the RGBA16F render buffer — the product handed to the caller — is row-major.
// LDS per workgroup: the transfer function as (value, difference) pairs and the three brick-offset
// tables of the volume (vpt_device.h), staged once at kernel entry.
#pragma once
#include "vpt_device.h"
#include "vpt_tonemap.h"

#define VPT_TILE        16
#define VPT_BLOCK       256
#define VPT_MAX_TRACK_ITERS 65536u
#define VPT_UNROLL      4          // samples in flight per ray in the EAM / ISO / Depth marches (8: EAM 4 %, ISO 8 % slower —
                                   // their early exits throw the speculative samples away)
#define VPT_UNROLL_MIP  8          // MIP has no early exit: 8 in flight is 3 % faster than 4

struct PixMap {
    int W, H;          // full image plane
    int local_h;       // rows held by this renderer (== H when unsharded)
    int tiles_x, ntiles;
    int G, g, R;       // shard: world, rank, rows per block
    int rshift;        // log2(R) when R is a power of two, else -1
    int ty0;           // first tile row of this launch (a pass split over two streams launches two row ranges; else 0)
    const float *ndc_x, *ndc_y;   // pixel-centre NDC per column / global row: fl(fl((2i+1)/W) - 1), W resp. H entries
    // launch over a LIST of tiles instead of the whole tile grid (tile classes): workgroup b of a
    // 1-D grid takes tile tile_list[b] = tx | ty << 16; null = the 2-D grid below
    const uint32_t *tile_list;
    int list_n;        // entries of tile_list this launch covers (one-wave workgroups pad their grid to whole groups of eight tiles)
};
struct Pix { int i, j, l, k; bool valid, tile; };   // tile: the workgroup maps to a tile of the buffers (valid or padding pixel)

// Workgroup -> tile.  Blocks are dealt round-robin over the 8 XCDs (b % 8 labels the blocks that share an XCD's L2).
// XCD x owns the tiles on the diagonals (tx + ty) % 8 == x: every XCD gets tiles from the whole image (cube-missing
// and cube-crossing tiles balance; a contiguous band per XCD left the centre XCDs ~1.5x the work) and the same number
// of tiles from EVERY tile row, so a shard that holds only a few tile rows still fills all 8 XCDs evenly (ownership
// by whole tile rows measured 3 % slower on the full frame and 10-14 % slower on 1/2 and 1/8 shards).
// global row of local row l of this shard
VPT_DEV int global_row(const PixMap &m, int l) {
    if (m.G == 1) return l;
    int lb = (m.rshift >= 0) ? (l >> m.rshift) : (l / m.R);
    return (lb * m.G + m.g) * m.R + (l - lb * m.R);
}
// The grid is 2-D so that no integer division is needed: gridDim.x = tiles_x rounded up to a multiple of 8,
// blockIdx.y = tile row; the linear workgroup id (y * gridDim.x + x) is what the dispatcher deals over the XCDs and
// gridDim.x is a multiple of 8, so blockIdx.x & 7 labels the XCD.  Within a group of 8 adjacent tiles of a row the
// tile of XCD x is the one with (tx + ty) % 8 == x; groups past tiles_x fail the p.i < W test.
VPT_DEV Pix map_pixel(const PixMap &m) {
    int xcd = (int)blockIdx.x & 7, ty = (int)blockIdx.y + m.ty0;
    int tx, w, lane = (int)threadIdx.x & 63;
    if (m.tile_list) {
        // listed tiles (one scalar load): consecutive workgroups = consecutive list entries, dealt round-robin over the XCDs
        int e = (int)blockIdx.x;
        w = (int)threadIdx.x >> 6;
        if (blockDim.x == 64) {
            // one-wave workgroups: blockIdx.x = (group * 4 + wave) * 8 + xcd, entry = group * 8 + xcd — the four waves of a tile
            // stay on one XCD (they walk the same bricks), as in the 2-D grid's one-wave form below
            w = ((int)blockIdx.x >> 3) & 3;
            e = ((int)blockIdx.x >> 5) * 8 + xcd;
        }
        if (e >= m.list_n) { Pix q; q.i = q.j = q.l = q.k = 0; q.valid = false; q.tile = false; return q; }
        const uint32_t t = m.tile_list[e];
        tx = (int)(t & 0xffffu); ty = (int)(t >> 16);
    } else if (blockDim.x == 64) {
        // one-wave workgroups (the ray marchers when their LDS image is small): blockIdx.x = (group * 4 + wave) * 8 + xcd.
        // Same tile -> XCD map and the same buffer order; the unit the dispatcher balances over the CUs is a quarter
        // of a tile, so the few cube-crossing tiles of a frame spread evenly (wave-uniform branch on blockDim)
        w = ((int)blockIdx.x >> 3) & 3;
        tx = ((int)blockIdx.x >> 5) * 8 + ((xcd - ty) & 7);
    } else {
        w = (int)threadIdx.x >> 6;
        tx = ((int)blockIdx.x & ~7) + ((xcd - ty) & 7);
    }
    int t = ty * m.tiles_x + tx;
    Pix p;
    p.i = tx * VPT_TILE + (w & 1) * 8 + (lane & 7);
    p.l = ty * VPT_TILE + (w >> 1) * 8 + (lane >> 3);
    p.k = t * VPT_BLOCK + w * 64 + lane;
    p.j = global_row(m, p.l);
    p.valid = (p.i < m.W) && (p.l < m.local_h) && (p.j < m.H);
    p.tile = tx < m.tiles_x;
    return p;
}
// pixel-centre NDC from the host-built tables (the IEEE divisions (2i+1)/W are done once per image size, not per pixel)
VPT_DEV float ndc_col(const PixMap &m, int i) { return m.ndc_x[i]; }
VPT_DEV float ndc_row(const PixMap &m, int j) { return m.ndc_y[j]; }

struct LaoParams {               // = struct vpt_lao_params (include/vpt.h)
    int local_ambient_occlusion; float lao_weight; int num_lao_samples; float lao_step_size;
    int soft_shadows; float shadows_weight; int num_shadow_samples; float light_radius; float light_coefficient;
    float light_position[3];
};
// DOS slice pass: uOcclusionSamples / uOcclusionSamplesCount / uOcclusionScale / uDepth (DOSRenderer.js:212-254); the colour
// buffer (updated in place) travels in st0, the occlusion ping-pong pair in st2 (in) and st3 (out)
// tile_x0 / tile_y0: first 16x16 tile of the launch rectangle (the screen bounding box of the volume; the grid covers it)
struct DosParams { const float2 *samples; int nsamples; float scale_x, scale_y, depth; int tile_x0, tile_y0; };
struct PassArgs {
    PixMap pm;
    DevVolume vol;
    DevEnv env;
    const float4 *tf; int tf_w, tf_h; float tf_fw, tf_hi;   // decoded row 0 of the transfer function; (float)w, (float)(w-1)
    Mat4 mvp_inv;
    float seed, offset, step, extinction, inv_extinction, anisotropy;
    uint32_t max_bounces, steps;
    f3 light;
    float mix, blur, inv_w, inv_h;
    float isovalue, gradient_step, threshold;   // ISO / Depth (vpt_kernels_iso_depth.h)
    union {
        LaoParams lao;           // LAO renderer (vpt_kernels_iso_depth.h)
        DosParams dos;           // DOS renderer: one slice (vpt_kernels_iso_depth.h)
    };
    uint32_t multi_passes;       // > 1: the fused (MODE 1) kernels run that many passes per pixel in one launch (VPT_PLAY_FUSED)
    uint32_t miss_load_pos;      // k_mcm_miss: 1 = the position array is up to date (first classified pass after a reset or a whole-image pass): load it
    unsigned long long *violations;   // k_mcm_miss<.., CHECK>: events of "miss" tiles that were inside the cube (must stay 0)
    void *frame;                 // tile order
    void *acc;                   // tile order (ping-pong collapsed: each pixel reads and writes only itself)
    float4 *st0, *st1, *st2, *st3;   // MCM photon state, tile order
    uint2 *render;               // RGBA16F, row-major local rows
    unsigned long long *samples; // volume-sample counter
    // frame sequences (vpt_renderer_play / hipGraph replay): the per-frame uniforms come from a device table indexed
    // by a device-side frame counter, so one captured launch sequence serves every replay
    const struct FrameVar *frame_table;     // ring of frame_mask + 1 entries
    const uint32_t *frame_counter;          // frames played so far (monotonic); entry = counter & frame_mask
    // tone mapping fused into the fused passes' frame store (VPT_TONEMAPPER_OPTION_FUSE): the armed tone mapper's byte table
    // (vpt_tonemap.h: 65 536 entries indexed by the half bits of a channel + the constant alpha) with the TmFuse block behind it — the RGBA8
    // output (row-major local rows, like `render`), the mapper's form and Artistic's uniforms; null = no tone mapper armed
    const uint8_t *tm_table;
    uint32_t frame_mask;
    uint32_t frame_base;                    // fused sequences (multi_passes > 1): index of the sequence's first frame, BY VALUE — a pass split over
                                            // streams must not read the device counter, which the context's stream advances behind its own range only
};
struct FrameVar { float seed, offset, mix, pad0; float lx, ly, lz, pad1; };   // the uniforms that change per frame
VPT_DEV void apply_frame_table(PassArgs &a) {
    if (a.frame_table) {
        FrameVar v = a.frame_table[*a.frame_counter & a.frame_mask];
        a.seed = v.seed; a.offset = v.offset; a.mix = v.mix;
        a.light = f3{ v.lx, v.ly, v.lz };
    }
}
__global__ void k_advance_frame(uint32_t *counter) { *counter = *counter + 1u; }
// VPT_PLAY_FUSED for the accumulating renderers: pass f of a fused launch takes its per-frame uniforms from the f-th
// entry of the frame table after the device frame counter, exactly as launch f of the unfused sequence would
VPT_DEV uint32_t multi_pass_count(const PassArgs &a) { return a.multi_passes > 1u ? a.multi_passes : 1u; }
VPT_DEV void multi_pass_select(PassArgs &a, uint32_t base, uint32_t f) {
    if (a.multi_passes > 1u) {
        FrameVar v = a.frame_table[(base + f) & a.frame_mask];
        a.seed = v.seed; a.offset = v.offset; a.mix = v.mix;
        a.light = f3{ v.lx, v.ly, v.lz };
    }
}

// dynamic LDS: [tf pairs: tf_w * 2 float4][TX nx][TY ny][TZ nz], 4-byte entries (byte offsets, or brick codes when WIDE)
template <bool WIDE>
VPT_DEV LdsTables stage_lds(float4 *lds, const PassArgs &a) {
    const int nthreads = (int)blockDim.x;
    for (int t = (int)threadIdx.x; t < a.tf_w; t += nthreads) {
        float4 v = a.tf[t], n = a.tf[min(t + 1, a.tf_w - 1)];
        lds[2 * t] = v;
        lds[2 * t + 1] = make_float4(n.x - v.x, n.y - v.y, n.z - v.z, n.w - v.w);
    }
    int ntab = a.vol.nx + a.vol.ny + a.vol.nz;
    uint32_t *tab = (uint32_t *)(lds + 2 * a.tf_w);
    const uint32_t *src = WIDE ? a.vol.tabc : a.vol.tab32;
    // 16 bytes per lane: the staging is on the critical path of every workgroup (a 512^3 table image is 6 KiB: two round
    // trips for 256 threads instead of six)
    const int n4 = ntab >> 2;
    for (int t = (int)threadIdx.x; t < n4; t += nthreads) ((uint4 *)tab)[t] = ((const uint4 *)src)[t];
    for (int t = (n4 << 2) + (int)threadIdx.x; t < ntab; t += nthreads) tab[t] = src[t];
    __syncthreads();
    LdsTables r;
    r.tf = lds;
    r.tx = tab; r.ty = tab + a.vol.nx; r.tz = tab + (a.vol.nx + a.vol.ny);
    return r;
}
// sampleVolumeColor: MIPRenderer.glsl:45-49 (= EAM :46-50, MCS :64-68, MCM :85-89)
template <int V>
VPT_DEV float4 sample_volume_color(const PassArgs &a, const LdsTables &t, f3 p) {
    if (V & VPT_V_RG) {
        f2 rg = sample_volume_rg<V>(a.vol, t, p);
        return sample_tf2d(a.tf, a.tf_w, a.tf_h, rg.x, rg.y);
    }
    float r = sample_volume<V>(a.vol, t, p);
    return sample_tf(t.tf, a.tf_fw, a.tf_hi, r);
}
// volume-sample counter: wave reduction (shuffles) -> workgroup reduction (one LDS word) -> ONE global atomic per
// workgroup, spread over VPT_COUNTER_SLOTS addresses on separate 128-B lines (32 k same-address atomics per frame
// serialised at ~12 ns each and dominated the cheap passes); the host sums the slots.
#define VPT_COUNTER_SLOTS 64
#define VPT_COUNTER_STRIDE 16      // unsigned long long words per slot = 128 B
VPT_DEV void count_samples(unsigned long long *ctr, uint32_t n) {
    __shared__ uint32_t block_sum;
    if (threadIdx.x == 0) block_sum = 0;
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off);
    if (((int)threadIdx.x & 63) == 0 && n) atomicAdd(&block_sum, n);
    __syncthreads();
    if (threadIdx.x == 0 && block_sum)
        atomicAdd(ctr + (size_t)((blockIdx.y * gridDim.x + blockIdx.x) % VPT_COUNTER_SLOTS) * VPT_COUNTER_STRIDE, (unsigned long long)block_sum);
}
VPT_DEV uint2 pack_half4(float x, float y, float z, float w) {
    uint2 r;
    r.x = (uint32_t)to_half_bits(x) | ((uint32_t)to_half_bits(y) << 16);
    r.y = (uint32_t)to_half_bits(z) | ((uint32_t)to_half_bits(w) << 16);
    return r;
}
// A frame's texels are written once and read by somebody else later (tone mapper, gather, read-back): stored non-temporally,
// they do not take L2 lines from the photon state, the accumulators and the brick lines (measured: -1 % per MCM frame, -0.5 .. -1 % for the ray marchers, same bits)
VPT_DEV void store_frame_texel(uint2 *dst, uint2 v) {
    __builtin_nontemporal_store(((unsigned long long)v.y << 32) | v.x, (unsigned long long *)dst);
}
// the armed tone mapper's RGBA8 texel of an RGBA16F texel: the same table lookups as k_tonemap_apply_table* (vpt_tonemap.h) — bit-identical to
// the separate pass by construction; *out (if asked for) = the tone mapper's output image
VPT_DEV uint32_t tone_map_texel(const uint8_t *table, uint2 v, uint32_t **out) {
    typedef const TmFuse __attribute__((address_space(4))) *FuseArgs;       // constant address space: scalar loads
    const FuseArgs f = (FuseArgs)(uintptr_t)(table + VPT_TM_FUSE_OFFSET);
    const int mode = f->mode;
    if (out) *out = f->out;
    uint32_t rgb = (uint32_t)table[v.x & 0xffffu] | ((uint32_t)table[v.x >> 16] << 8) | ((uint32_t)table[v.y & 0xffffu] << 16);
    if (mode == 3) {                                            // k_tonemap_apply_table_artistic: the grey term must be finite
        const float low = f->low, range = f->range;
        float4 c = half4_to_float4(v);
        f3 w = { (c.x - low) / range, (c.y - low) / range, (c.z - low) / range };
        const float gray = 0.57735026918962576f;
        float z = (dot3(w, f3{ gray, gray, gray }) * gray) * f->one_minus_saturation;
        return ((z == 0.0f) ? rgb : 0u) | 0xff000000u;
    }
    return rgb | ((uint32_t)table[mode == 2 ? (v.y >> 16) : 65536u] << 24);
}
// the frame store of the fused passes: the RGBA16F texel into the render buffer, and — when a tone mapper is armed on this renderer —
// its tone-mapped RGBA8 texel into the tone mapper's output as well, which saves that pass and its launch per displayed frame
VPT_DEV void store_frame(const PassArgs &a, const Pix &p, uint2 v) {
    const size_t idx = (size_t)p.l * a.pm.W + p.i;
    store_frame_texel(&a.render[idx], v);
    if (a.tm_table) {                                           // wave-uniform
        uint32_t *out;
        const uint32_t texel = tone_map_texel(a.tm_table, v, &out);
        out[idx] = texel;
    }
}

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
__global__ void __launch_bounds__(VPT_BLOCK) k_mip(PassArgs a) {
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
VPT_DEV uint2 eam_to_half4(uint32_t q) {   // render: EAMRenderer.glsl:151-153
    return pack_half4(from_unorm8(q & 0xffu), from_unorm8((q >> 8) & 0xffu),
                      from_unorm8((q >> 16) & 0xffu), from_unorm8(q >> 24));
}
template <int MODE, int V>
__global__ void __launch_bounds__(VPT_BLOCK) k_eam(PassArgs a) {
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
__global__ void __launch_bounds__(VPT_BLOCK) k_mcs(PassArgs a) {
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
#ifdef VPT_WITH_PERSISTENT_KERNELS   // measured-slower alternatives, not in the default build: make EXTRA=-DVPT_WITH_PERSISTENT_KERNELS
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
#endif
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

// =============================================================================================
// MCM — MCMRenderer.glsl, mixins/Photon.glsl, mixins/unprojectRand.glsl
// =============================================================================================
struct Photon {
    f3 position, direction, transmittance, radiance;
    uint32_t bounces, samples;
};
// mixins/unprojectRand.glsl:3-24.  The near-plane point depends on the pixel only when blur == 0 (the reference
// always passes 0: MCMRenderer.js:93,157): the two disk uniforms are still drawn, their product with 0 is an exact
// zero, so `from` equals the un-jittered unprojection `from0` computed once per pixel.
// NOBLUR: the caller knows blur == 0 (k_mcm_miss: the tile classes are only used with it), the general branch is not compiled
template <bool NOBLUR = false>
VPT_DEV void unproject_rand(uint32_t &state, float px, float py, const PassArgs &a, f3 from0, f3 &from, f3 &to) {
    if (NOBLUR || a.blur == 0.0f) {
        random_uniform(state); random_uniform(state);
        from = from0;
    } else {
        f2 d = random_disk(state);
        from = dehomogenize(mat4_mul_point(a.mvp_inv, px + d.x * a.blur, py + d.y * a.blur, -1.0f));
    }
    float sx = random_uniform(state), sy = random_uniform(state);
    float ax = fmaf(sx, 2.0f, -1.0f) * a.inv_w;
    float ay = fmaf(sy, 2.0f, -1.0f) * a.inv_h;
    to = dehomogenize(mat4_mul_point(a.mvp_inv, px + ax, py + ay, 1.0f));
}
VPT_DEV f3 unproject_near(float px, float py, const PassArgs &a) {
    return dehomogenize(mat4_mul_point(a.mvp_inv, px + 0.0f, py + 0.0f, -1.0f));
}
// resetPhoton: MCMRenderer.glsl:70-78
// the photon's start on its ray: from + max(tnear, 0) * direction (MCMRenderer.glsl:75-77).  A function of (from, direction) alone:
// the kernels of cube-missing tiles (k_mcm_miss) recompute it from the stored direction instead of storing it.
VPT_DEV f3 photon_start(f3 from, f3 dir) {
    float tnear = vmax(intersect_cube_near(from, dir), 0.0f);
    return madd3(from, tnear, dir);
}
template <bool NOBLUR = false>
VPT_DEV void reset_photon(uint32_t &state, Photon &ph, float px, float py, const PassArgs &a, f3 from0) {
    f3 from, to;
    unproject_rand<NOBLUR>(state, px, py, a, from0, from, to);
    ph.direction = normalize3(sub3(to, from));
    ph.bounces = 0u;
    ph.position = photon_start(from, ph.direction);
    ph.transmittance = f3{ 1.0f, 1.0f, 1.0f };
}
// sampleHenyeyGreensteinAngleCosine: MCMRenderer.glsl:91-95
VPT_DEV float hg_cos(uint32_t &state, float g) {
    float g2 = g * g;
    float c = (1.0f - g2) * rcp_nr(fmaf(2.0f * g, random_uniform(state), 1.0f - g));
    return fmaf(-c, c, 1.0f + g2) * rcp_nr(2.0f * g);
}
// sampleHenyeyGreenstein: MCMRenderer.glsl:97-106
VPT_DEV f3 sample_hg(uint32_t &state, float g, f3 dir) {
    f3 u = random_sphere(state);
    if (fabsf(g) < 1e-5f) return u;
    float hgcos = hg_cos(state, g);
    float ud = dot3(u, dir);
    f3 c = { fmaf(-ud, dir.x, u.x), fmaf(-ud, dir.y, u.y), fmaf(-ud, dir.z, u.z) };
    c = normalize3(c);
    float s = sqrt_nr(fmaf(-hgcos, hgcos, 1.0f));
    return f3{ fmaf(s, c.x, hgcos * dir.x), fmaf(s, c.y, hgcos * dir.y), fmaf(s, c.z, hgcos * dir.z) };
}
// radiance += (rad - radiance) / float(samples)   (MCMRenderer.glsl:147-150,154-157), as * (1/n)
VPT_DEV void photon_deposit(Photon &ph, f3 rad) {
    ph.samples++;
    float inv_n = rcp_nr((float)ph.samples);
    ph.radiance.x += (rad.x - ph.radiance.x) * inv_n;
    ph.radiance.y += (rad.y - ph.radiance.y) * inv_n;
    ph.radiance.z += (rad.z - ph.radiance.z) * inv_n;
}

// reset/fragment main(): MCMRenderer.glsl:259-275 (seeded from the NDC position)
__global__ void __launch_bounds__(VPT_BLOCK) k_mcm_reset(PassArgs a) {
    Pix p = map_pixel(a.pm);
    if (!p.tile) return;
    Photon ph;
    if (p.valid) {
        float px = ndc_col(a.pm, p.i), py = ndc_row(a.pm, p.j);
        uint32_t state = hash3(__float_as_uint(px), __float_as_uint(py), __float_as_uint(a.seed));
        reset_photon(state, ph, px, py, a, unproject_near(px, py, a));
    } else {
        ph.position = f3{ 0.0f, 0.0f, 0.0f };
        ph.direction = f3{ 0.0f, 0.0f, 1.0f };
    }
    ((f3 *)a.st0)[p.k] = ph.position;
    a.st1[p.k] = make_float4(ph.direction.x, ph.direction.y, ph.direction.z, 0.0f);
    ((f3 *)a.st2)[p.k] = f3{ 1.0f, 1.0f, 1.0f };
    a.st3[p.k] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
}

// the `steps` delta-tracking events of one pixel on its persistent photon: MCMRenderer.glsl:128-166
// sampleVolumeColor of an event.  The reference samples BEFORE its bounds test (MCMRenderer.glsl:132-142), so the sample of an
// event that leaves the cube is taken at the clamped position (CLAMP_TO_EDGE) and then discarded.  It is executed here too, for
// every lane (the empty asm keeps the compiler from proving it dead on the out-of-bounds path): an out-of-cube position has at
// least one coordinate clamped onto the first or last voxel plane of its axis with weight 0 there, so its eight-tap footprint
// degenerates to four taps on a face of the volume — fetched from the boundary atlas (vpt_device.h, sample_volume_boundary:
// one aligned dword instead of two unaligned 8-byte gathers; bit-identical value).  At the benchmark camera 93 % of the events
// end outside the cube (80 % of the pixels never meet it), and the pass was bound by the texture path's gather rate.
// LINEAR one-channel byte volumes: the sample in two phases.  A wave whose lanes disagree (HIT tiles: some photons inside the cube,
// some outside) would run the two samplers one after the other, each waiting for its own load: mcm_sample_issue puts the loads in
// flight — the atlas dword for the lanes outside, the two brick windows for the lanes inside —, mcm_sample_finish blends them, so
// that both kinds fly together (a wave-event of a HIT tile is a chain of dependent latencies: -1 memory latency per event), and
// the EARLY event loops below put the out-of-cube lanes' path end between the two.  The second phase tests an opaque copy of the
// predicate, or the compiler would thread the phases back into one branch.
struct SampleLoads { uint32_t aw; uint64_t w0, w1; float f0, f1, f2; uint32_t atlas; };
template <int V>
VPT_DEV SampleLoads mcm_sample_issue(const PassArgs &a, const LdsTables &t, f3 p, bool oob) {
    constexpr bool WIDE = (V & VPT_V_WIDE) != 0;
    SampleLoads s;
    s.aw = 0u; s.w0 = 0ull; s.w1 = 0ull; s.f0 = 0.0f; s.f1 = 0.0f; s.f2 = 0.0f;
    const bool at = oob && a.vol.atlas != nullptr;
    if (at) {
        s.aw = a.vol.atlas[boundary_cell(a.vol, p, s.f0, s.f1)];
    } else {
        uint32_t x, y, z;
        linear_cell(p.x, a.vol.fnx, a.vol.hx, x, s.f0);
        linear_cell(p.y, a.vol.fny, a.vol.hy, y, s.f1);
        linear_cell(p.z, a.vol.fnz, a.vol.hz, z, s.f2);
        const uint8_t *b = cell_addr<WIDE>(a.vol, t, x, y, z);
        __builtin_memcpy(&s.w0, b, 8);
        __builtin_memcpy(&s.w1, b + 25, 8);
    }
    s.atlas = at ? 1u : 0u;
    asm volatile("" : "+v"(s.atlas));
    return s;
}
VPT_DEV float4 mcm_sample_finish(const PassArgs &a, const LdsTables &t, const SampleLoads &s) {
    float r;
    if (s.atlas) r = boundary_blend(s.aw, s.f0, s.f1);
    else r = trilinear_blend((uint32_t)s.w0, (uint32_t)(s.w0 >> 32), (uint32_t)s.w1, (uint32_t)(s.w1 >> 32), s.f0, s.f1, s.f2);
    float4 vs = sample_tf(t.tf, a.tf_fw, a.tf_hi, r);
    asm volatile("" : "+v"(vs.w));
    return vs;
}
template <int V>
VPT_DEV float4 mcm_sample(const PassArgs &a, const LdsTables &t, f3 p, bool oob) {
    if (!(V & (VPT_V_NEAREST | VPT_V_RG | VPT_V_F32))) return mcm_sample_finish(a, t, mcm_sample_issue<V>(a, t, p, oob));
    float4 vs = sample_volume_color<V>(a, t, p);
    asm volatile("" : "+v"(vs.w));
    return vs;
}
template <int V>
VPT_DEV void mcm_events(const PassArgs &a, const LdsTables &t, Photon &ph, float px, float py) {
    const f3 from0 = unproject_near(px, py, a);

    uint32_t state = hash3(__float_as_uint(ndc_to_uv(px)), __float_as_uint(ndc_to_uv(py)), __float_as_uint(a.seed));
    for (uint32_t s = 0u; s < a.steps; s++) {
        float dist = random_exponential(state, a.inv_extinction);
        ph.position = madd3(ph.position, dist, ph.direction);
        f3 q = ph.position;
        // any(greaterThan(pos, 1)) || any(lessThan(pos, 0)), NaN components compare false
        bool oob = (vmax(vmax(q.x, q.y), q.z) > 1.0f) || (vmin(vmin(q.x, q.y), q.z) < 0.0f);
        float4 vs = mcm_sample<V>(a, t, q, oob);
        float p_null = 1.0f - vs.w;
        float p_scat = (ph.bounces >= a.max_bounces) ? 0.0f : vs.w * vmax(vmax(vs.x, vs.y), vs.z);
        float p_abs = 1.0f - p_null - p_scat;
        float wheel = random_uniform(state);
        if (oob || wheel < p_abs) {
            // out of bounds: radiance = transmittance * env; absorption: radiance = 0 — one shared deposit + resetPhoton
            f3 rad = { 0.0f, 0.0f, 0.0f };
            if (oob) {
                float4 env = sample_environment(a.env, ph.direction);
                rad = f3{ ph.transmittance.x * env.x, ph.transmittance.y * env.y, ph.transmittance.z * env.z };
            }
            photon_deposit(ph, rad);
            reset_photon(state, ph, px, py, a, from0);
        } else if (wheel < p_abs + p_scat) {
            ph.transmittance.x *= vs.x; ph.transmittance.y *= vs.y; ph.transmittance.z *= vs.z;
            ph.direction = sample_hg(state, a.anisotropy, ph.direction);
            ph.bounces++;
        }
    }
}
// The same events with the out-of-cube lanes' path end (deposit + resetPhoton: they need the random stream only, and whether a
// position is out of bounds is known before its sample) placed BETWEEN issuing the sample's loads and consuming them.  Same draws in
// the same order, same arithmetic: bit-identical.  It costs ~12 more live registers, so it is the form of the HIT-tile kernel where
// occupancy is not what limits it — a shard's few tiles, whose pass is one wave per SIMD walking a chain of dependent latencies
// (DESIGN.md section 8).  LINEAR one-channel byte volumes only (the tile classes' precondition).
template <int V>
VPT_DEV void mcm_events_early(const PassArgs &a, const LdsTables &t, Photon &ph, float px, float py) {
    const f3 from0 = unproject_near(px, py, a);
    uint32_t state = hash3(__float_as_uint(ndc_to_uv(px)), __float_as_uint(ndc_to_uv(py)), __float_as_uint(a.seed));
    for (uint32_t s = 0u; s < a.steps; s++) {
        float dist = random_exponential(state, a.inv_extinction);
        ph.position = madd3(ph.position, dist, ph.direction);
        f3 q = ph.position;
        bool oob = (vmax(vmax(q.x, q.y), q.z) > 1.0f) || (vmin(vmin(q.x, q.y), q.z) < 0.0f);
        const SampleLoads ld = mcm_sample_issue<V>(a, t, q, oob);
        const uint32_t bounces = ph.bounces;                       // (the scattering probability reads the count before the path end)
        float wheel = random_uniform(state);
        if (oob) {
            float4 env = sample_environment(a.env, ph.direction);
            photon_deposit(ph, f3{ ph.transmittance.x * env.x, ph.transmittance.y * env.y, ph.transmittance.z * env.z });
            reset_photon(state, ph, px, py, a, from0);
        }
        float4 vs = mcm_sample_finish(a, t, ld);
        float p_null = 1.0f - vs.w;
        float p_scat = (bounces >= a.max_bounces) ? 0.0f : vs.w * vmax(vmax(vs.x, vs.y), vs.z);
        float p_abs = 1.0f - p_null - p_scat;
        if (!oob) {
            if (wheel < p_abs) {
                photon_deposit(ph, f3{ 0.0f, 0.0f, 0.0f });
                reset_photon(state, ph, px, py, a, from0);
            } else if (wheel < p_abs + p_scat) {
                ph.transmittance.x *= vs.x; ph.transmittance.y *= vs.y; ph.transmittance.z *= vs.z;
                ph.direction = sample_hg(state, a.anisotropy, ph.direction);
                ph.bounces++;
            }
        }
    }
}
// ---- fast-arithmetic variant of the MCM events (VPT_OPTION_FAST_MATH, kernel variant bit VPT_V_FAST) ---------------------
// The same shader (MCMRenderer.glsl:128-166, resetPhoton :70-78, HG :91-106) with the arithmetic a GPU driver gives GLSL:
// v_rcp_f32 / v_rsq_f32 / v_sqrt_f32 / v_log_f32 / v_sin_f32 / v_cos_f32 (1 ulp class, GLSL ES 3.00 §4.5.1 allows 2.5 ulp for a/b and
// leaves log / sin / cos implementation-defined) instead of the contract's software routines, and algebraically equal forms
// that need fewer instructions:
//   * -log(u)/rate = log2(u) * (-ln 2 / rate);
//   * (random_square * 2 - 1) * inverseResolution = k * (2^-31 / W) - 1 / W with k the PCG state as a float;
//   * inverseMvp * (p + jitter, 1, 1) = inverseMvp * (p, 1, 1) + jitter.x * column0 + jitter.y * column1 (the first term is a
//     pixel constant);
//   * normalize(to.xyz / to.w - from) = sign(to.w) * normalize(to.xyz - to.w * from): no division by w.
// The integer PCG stream is identical, so the two variants take the same decisions except where a comparison falls
// within rounding error; there is NO bit-exact CPU twin of this variant — it is checked against the contract oracle by
// first-event agreement and converged-image statistics (tests/test_gpu_fast_math.py, tolerance in DESIGN.md §3).
VPT_DEV float hw_rcp(float x) { return __builtin_amdgcn_rcpf(x); }        // 1/(+-0) = +-inf, as the slab test needs
VPT_DEV float hw_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
VPT_DEV float hw_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
VPT_DEV float hw_log2(float x) { return __builtin_amdgcn_logf(x); }       // log2(0) = -inf
VPT_DEV float pcg_float(uint32_t &state) { state = pcg(state); return (float)state; }
// sampleHenyeyGreenstein with hardware sqrt / sin / cos; v_sin_f32 / v_cos_f32 take their argument in revolutions:
// sin(TWOPI * u) = v_sin_f32(u)
VPT_DEV f3 sample_hg_fast(uint32_t &state, float g, f3 dir) {
    float u1 = pcg_float(state) * 0x1p-32f, u2 = pcg_float(state) * 0x1p-32f;
    float r0 = hw_sqrt(u1);
    f2 d = { r0 * __builtin_amdgcn_cosf(u2), r0 * __builtin_amdgcn_sinf(u2) };
    float norm = fmaf(d.y, d.y, d.x * d.x);
    float radius = 2.0f * hw_sqrt(1.0f - norm);
    f3 u = { radius * d.x, radius * d.y, fmaf(-2.0f, norm, 1.0f) };
    if (fabsf(g) < 1e-5f) return u;
    float g2 = g * g;
    float c = (1.0f - g2) * hw_rcp(fmaf(2.0f * g, pcg_float(state) * 0x1p-32f, 1.0f - g));
    float hgcos = fmaf(-c, c, 1.0f + g2) * hw_rcp(2.0f * g);
    float ud = dot3(u, dir);
    f3 cc = { fmaf(-ud, dir.x, u.x), fmaf(-ud, dir.y, u.y), fmaf(-ud, dir.z, u.z) };
    float sq = hw_sqrt(fmaf(-hgcos, hgcos, 1.0f)) * hw_rsq(dot3(cc, cc));
    return f3{ fmaf(sq, cc.x, hgcos * dir.x), fmaf(sq, cc.y, hgcos * dir.y), fmaf(sq, cc.z, hgcos * dir.z) };
}
// the end of a path in the fast variant: deposit `rad`, then resetPhoton (MCMRenderer.glsl:146-151 / 153-158, :70-78)
struct FastPixel { f3 from0; float4 fb; float jx, jy; };
// photon_start in the fast variant's arithmetic: min((0 - f) * iv, (1 - f) * iv) = -f * iv + min(iv, 0), one min and one fma per slab
VPT_DEV f3 photon_start_fast(f3 from0, f3 dir) {
    f3 iv = { hw_rcp(dir.x), hw_rcp(dir.y), hw_rcp(dir.z) };
    float tx = fmaf(-from0.x, iv.x, vmin(iv.x, 0.0f));
    float ty = fmaf(-from0.y, iv.y, vmin(iv.y, 0.0f));
    float tz = fmaf(-from0.z, iv.z, vmin(iv.z, 0.0f));
    float tnear = vmax(vmax(vmax(tx, ty), tz), 0.0f);
    return madd3(from0, tnear, dir);
}
// the pixel constants of the fast variant's resetPhoton with blur == 0: the near-plane point and the far-plane base point (homogeneous)
VPT_DEV FastPixel fast_pixel(const PassArgs &a, float px, float py) {
    FastPixel c;
    const float4 nb = mat4_mul_point(a.mvp_inv, px, py, -1.0f);
    const float inw = hw_rcp(nb.w);
    c.from0 = f3{ nb.x * inw, nb.y * inw, nb.z * inw };
    c.fb = mat4_mul_point(a.mvp_inv, px, py, 1.0f);
    c.jx = 0x1p-31f * a.inv_w; c.jy = 0x1p-31f * a.inv_h;
    return c;
}
template <bool NOBLUR = false>
VPT_DEV void fast_path_end(const PassArgs &a, const FastPixel &c, uint32_t &state, Photon &ph, f3 rad, float px, float py) {
    const float *m = a.mvp_inv.m;
    ph.samples++;
    float inv_n = hw_rcp((float)ph.samples);
    ph.radiance.x = fmaf(rad.x - ph.radiance.x, inv_n, ph.radiance.x);
    ph.radiance.y = fmaf(rad.y - ph.radiance.y, inv_n, ph.radiance.y);
    ph.radiance.z = fmaf(rad.z - ph.radiance.z, inv_n, ph.radiance.z);
    if (NOBLUR || a.blur == 0.0f) {
        state = pcg(pcg(state));                          // the disk sample's two draws (multiplied by blur = 0)
        float ax = fmaf(pcg_float(state), c.jx, -a.inv_w);
        float ay = fmaf(pcg_float(state), c.jy, -a.inv_h);
        float4 th = { fmaf(m[4], ay, fmaf(m[0], ax, c.fb.x)), fmaf(m[5], ay, fmaf(m[1], ax, c.fb.y)),
                      fmaf(m[6], ay, fmaf(m[2], ax, c.fb.z)), fmaf(m[7], ay, fmaf(m[3], ax, c.fb.w)) };
        f3 d = { fmaf(-th.w, c.from0.x, th.x), fmaf(-th.w, c.from0.y, th.y), fmaf(-th.w, c.from0.z, th.z) };
        float inv = __builtin_copysignf(hw_rsq(dot3(d, d)), th.w);
        f3 dir = { d.x * inv, d.y * inv, d.z * inv };
        ph.direction = dir;
        ph.position = photon_start_fast(c.from0, dir);
        ph.bounces = 0u;
        ph.transmittance = f3{ 1.0f, 1.0f, 1.0f };
    } else {
        reset_photon(state, ph, px, py, a, c.from0);      // depth-of-field runs: the contract's general path
    }
}
template <int V>
VPT_DEV void mcm_events_fast(const PassArgs &a, const LdsTables &t, Photon &ph, float px, float py) {
    const FastPixel c = fast_pixel(a, px, py);
    // -ln(u * 2^-32) / extinction = (log2(u) - 32) * ld
    const float ld = -0.6931471805599453f * a.inv_extinction, ld32 = -32.0f * ld;

    uint32_t state = hash3(__float_as_uint(ndc_to_uv(px)), __float_as_uint(ndc_to_uv(py)), __float_as_uint(a.seed));
    for (uint32_t s = 0u; s < a.steps; s++) {
        float dist = fmaf(hw_log2(pcg_float(state)), ld, ld32);
        ph.position = madd3(ph.position, dist, ph.direction);
        f3 q = ph.position;
        bool oob = (vmax(vmax(q.x, q.y), q.z) > 1.0f) || (vmin(vmin(q.x, q.y), q.z) < 0.0f);
        float4 vs = mcm_sample<V>(a, t, q, oob);
        float p_null = 1.0f - vs.w;
        float p_scat = (ph.bounces >= a.max_bounces) ? 0.0f : vs.w * vmax(vmax(vs.x, vs.y), vs.z);
        float p_abs = 1.0f - p_null - p_scat;
        float wheel = pcg_float(state) * 0x1p-32f;
        if (oob || wheel < p_abs) {
            f3 rad = { 0.0f, 0.0f, 0.0f };
            if (oob) {
                float4 env = sample_environment(a.env, ph.direction);
                rad = f3{ ph.transmittance.x * env.x, ph.transmittance.y * env.y, ph.transmittance.z * env.z };
            }
            fast_path_end(a, c, state, ph, rad, px, py);
        } else if (wheel < p_abs + p_scat) {
            ph.transmittance.x *= vs.x; ph.transmittance.y *= vs.y; ph.transmittance.z *= vs.z;
            ph.direction = sample_hg_fast(state, a.anisotropy, ph.direction);
            ph.bounces++;
        }
    }
}
template <int V>
VPT_DEV void mcm_events_fast_early(const PassArgs &a, const LdsTables &t, Photon &ph, float px, float py) {     // see mcm_events_early
    const FastPixel c = fast_pixel(a, px, py);
    const float ld = -0.6931471805599453f * a.inv_extinction, ld32 = -32.0f * ld;
    uint32_t state = hash3(__float_as_uint(ndc_to_uv(px)), __float_as_uint(ndc_to_uv(py)), __float_as_uint(a.seed));
    for (uint32_t s = 0u; s < a.steps; s++) {
        float dist = fmaf(hw_log2(pcg_float(state)), ld, ld32);
        ph.position = madd3(ph.position, dist, ph.direction);
        f3 q = ph.position;
        bool oob = (vmax(vmax(q.x, q.y), q.z) > 1.0f) || (vmin(vmin(q.x, q.y), q.z) < 0.0f);
        const SampleLoads lds = mcm_sample_issue<V>(a, t, q, oob);
        const uint32_t bounces = ph.bounces;
        float wheel = pcg_float(state) * 0x1p-32f;
        if (oob) {
            float4 env = sample_environment(a.env, ph.direction);
            fast_path_end(a, c, state, ph, f3{ ph.transmittance.x * env.x, ph.transmittance.y * env.y, ph.transmittance.z * env.z }, px, py);
        }
        float4 vs = mcm_sample_finish(a, t, lds);
        float p_null = 1.0f - vs.w;
        float p_scat = (bounces >= a.max_bounces) ? 0.0f : vs.w * vmax(vmax(vs.x, vs.y), vs.z);
        float p_abs = 1.0f - p_null - p_scat;
        if (!oob) {
            if (wheel < p_abs) {
                fast_path_end(a, c, state, ph, f3{ 0.0f, 0.0f, 0.0f }, px, py);
            } else if (wheel < p_abs + p_scat) {
                ph.transmittance.x *= vs.x; ph.transmittance.y *= vs.y; ph.transmittance.z *= vs.z;
                ph.direction = sample_hg_fast(state, a.anisotropy, ph.direction);
                ph.bounces++;
            }
        }
    }
}
// Photon state in HBM (MCMRenderer.js:214-263 keeps four RGBA32F attachments: [pos, 0] [dir, bounces] [T, 0] [radiance, samples]):
// the two constant zeros are not stored — position and transmittance are 12-byte texels (one dwordx3 per lane, a wave's
// 64 texels one contiguous 768-byte segment), direction+bounces and radiance+samples 16-byte texels: 56 bytes per pixel each
// way instead of 64.  vpt_renderer_read re-expands them to RGBA32F (k_detile_mcm3).
struct PhotonState { f3 s0; float4 s1; f3 s2; float4 s3; };
VPT_DEV PhotonState photon_load(const PassArgs &a, int k) {
    PhotonState s;
    s.s0 = ((const f3 *)a.st0)[k]; s.s1 = a.st1[k]; s.s2 = ((const f3 *)a.st2)[k]; s.s3 = a.st3[k];
    return s;
}
VPT_DEV void photon_store(const PassArgs &a, int k, const struct Photon &ph);
VPT_DEV Photon photon_unpack(float4 s0, float4 s1, float4 s2, float4 s3) {   // MCMRenderer.glsl:117-126
    Photon ph;
    ph.position = f3{ s0.x, s0.y, s0.z };
    ph.direction = f3{ s1.x, s1.y, s1.z };
    ph.bounces = (uint32_t)(s1.w + 0.5f);
    ph.transmittance = f3{ s2.x, s2.y, s2.z };
    ph.radiance = f3{ s3.x, s3.y, s3.z };
    ph.samples = (uint32_t)(s3.w + 0.5f);
    return ph;
}
VPT_DEV Photon photon_unpack(const PhotonState &s) {
    return photon_unpack(make_float4(s.s0.x, s.s0.y, s.s0.z, 0.0f), s.s1, make_float4(s.s2.x, s.s2.y, s.s2.z, 0.0f), s.s3);
}
VPT_DEV void photon_store(const PassArgs &a, int k, const Photon &ph) {       // MCMRenderer.glsl:168-171
    ((f3 *)a.st0)[k] = ph.position;
    a.st1[k] = make_float4(ph.direction.x, ph.direction.y, ph.direction.z, (float)ph.bounces);
    ((f3 *)a.st2)[k] = ph.transmittance;
    a.st3[k] = make_float4(ph.radiance.x, ph.radiance.y, ph.radiance.z, (float)ph.samples);
}

#ifdef VPT_WITH_PERSISTENT_KERNELS
// Persistent form of the integrate pass: every wave walks several 8x8-pixel segments of the tile-ordered state arrays
// (segment g = lanes [64g, 64g+64)) and loads the NEXT segment's photon state (4 x dwordx4 per lane) before it starts the
// current segment's events; LDS tables are staged once per workgroup instead of once per tile.  Measured (512^3, 1080p,
// steps 8): 0.156 ms vs 0.148 ms for the one-workgroup-per-tile kernel — the prefetch registers cost two waves per SIMD
// (96 vs 72 VGPRs) and the state stream is only ~20 us of the frame (timing builds without state loads / stores:
// -4 us / -13 us), so this form is kept as an option (VPT_OPTION_MCM_PERSISTENT), not the default.
template <bool FUSE_RENDER, int V, bool PREFETCH>
__global__ void __launch_bounds__(VPT_BLOCK) __attribute__((amdgpu_waves_per_eu(PREFETCH ? 5 : 7, 8))) k_mcm_persist(PassArgs a, int nseg) {
    extern __shared__ float4 lds_raw[];
    LdsTables t = stage_lds<(V & VPT_V_WIDE) != 0>(lds_raw, a);
    const int lane = (int)threadIdx.x & 63;
    const int nwaves = (int)gridDim.x * (VPT_BLOCK / 64);
    int g = (int)blockIdx.x * (VPT_BLOCK / 64) + ((int)threadIdx.x >> 6);
    if (g >= nseg) return;
    size_t k = (size_t)g * 64 + lane;
    PhotonState nx;
    if (PREFETCH) nx = photon_load(a, (int)k);
    while (g < nseg) {
        const int gc = g;
        const size_t kc = k;
        if (!PREFETCH) nx = photon_load(a, (int)k);
        Photon ph = photon_unpack(nx);
        g += nwaves;
        k = (size_t)g * 64 + lane;
        if (PREFETCH && g < nseg) {              // wave-uniform: prefetch the next segment's photon state
            nx = photon_load(a, (int)k);
        }
        int t16 = gc >> 2, w = gc & 3;
        int ty = t16 / a.pm.tiles_x, tx = t16 - ty * a.pm.tiles_x;
        int i = tx * VPT_TILE + (w & 1) * 8 + (lane & 7);
        int l = ty * VPT_TILE + (w >> 1) * 8 + (lane >> 3);
        int j = global_row(a.pm, l);
        if (i < a.pm.W && l < a.pm.local_h && j < a.pm.H) {
            mcm_events<V>(a, t, ph, ndc_col(a.pm, i), ndc_row(a.pm, j));
            photon_store(a, (int)kc, ph);
            if (FUSE_RENDER)
                a.render[(size_t)l * a.pm.W + i] = pack_half4(ph.radiance.x, ph.radiance.y, ph.radiance.z, 1.0f);
        }
    }
}

#endif

#ifndef VPT_MCM_WAVES
#define VPT_MCM_WAVES 7          // waves per SIMD the integrate kernel is compiled for (72 VGPRs; 8 needs 64: A/B in DESIGN.md section 5)
#endif
// integrate/fragment main(): MCMRenderer.glsl:116-172.  FUSE_RENDER additionally performs
// _renderFrame (MCMRenderer.glsl:204-206) on the radiance it just produced.
template <bool FUSE_RENDER, int V>
__global__ void __launch_bounds__(VPT_BLOCK) __attribute__((amdgpu_waves_per_eu(VPT_MCM_WAVES, 8))) k_mcm_integrate(PassArgs a) {
    apply_frame_table(a);
    // the photon state (4 x dwordx4 per lane, one contiguous 1 KiB segment per wave and array) does not depend on the LDS
    // image: its loads are issued first, so they fly while the workgroup stages the tables and hashes its seed
    Pix p = map_pixel(a.pm);
    PhotonState st;
    if (p.tile) st = photon_load(a, p.k);
    extern __shared__ float4 lds_raw[];
    LdsTables t = stage_lds<(V & VPT_V_WIDE) != 0>(lds_raw, a);
    if (!p.valid) return;
    float px = ndc_col(a.pm, p.i), py = ndc_row(a.pm, p.j);
    Photon ph = photon_unpack(st);
    if (V & VPT_V_FAST) mcm_events_fast<V & ~VPT_V_FAST>(a, t, ph, px, py);
    else mcm_events<V>(a, t, ph, px, py);
    photon_store(a, p.k, ph);
    if (FUSE_RENDER) store_frame(a, p, pack_half4(ph.radiance.x, ph.radiance.y, ph.radiance.z, 1.0f));
}
// the HIT-tile kernel with the early path end (mcm_events_early), compiled for 5 waves per SIMD: selected by the library for tile lists
// short enough to be resident at once at that occupancy (shards)
template <bool FUSE_RENDER, int V>
__global__ void __launch_bounds__(VPT_BLOCK) __attribute__((amdgpu_waves_per_eu(5, 8))) k_mcm_integrate_early(PassArgs a) {
    apply_frame_table(a);
    Pix p = map_pixel(a.pm);
    PhotonState st;
    if (p.tile) st = photon_load(a, p.k);
    extern __shared__ float4 lds_raw[];
    LdsTables t = stage_lds<(V & VPT_V_WIDE) != 0>(lds_raw, a);
    if (!p.valid) return;
    float px = ndc_col(a.pm, p.i), py = ndc_row(a.pm, p.j);
    Photon ph = photon_unpack(st);
    if (V & VPT_V_FAST) mcm_events_fast_early<V & ~VPT_V_FAST>(a, t, ph, px, py);
    else mcm_events_early<V>(a, t, ph, px, py);
    photon_store(a, p.k, ph);
    if (FUSE_RENDER) store_frame(a, p, pack_half4(ph.radiance.x, ph.radiance.y, ph.radiance.z, 1.0f));
}
// =============================================================================================
// Tile classes (round 3).  The host classifies every 16x16 tile against the cube once per reset (vpt_hip.hip classify_tiles:
// a conservative test of the tile's pixel columns, jitter included, against the projected cube, enlarged): a MISS tile is one
// none of whose camera rays can meet the cube.  resetPhoton (MCMRenderer.glsl:70-78) parks such a photon at from + tnear * dir,
// outside the cube, on a ray that leaves it behind — so EVERY event of a MISS tile's pixel is "sample at the clamped position
// (MCMRenderer.glsl:132), find the position out of bounds (:135), deposit transmittance * environment (:136-140), resetPhoton
// (:141)": no in-cube sample, no absorption / scattering / null branch, and at the end of every pass transmittance = (1, 1, 1),
// bounces = 0 and position = photon_start(from0, direction).  k_mcm_miss runs exactly that straight-line event — the same
// arithmetic, draw for draw, as the oob lanes of k_mcm_integrate, so the buffers are bit-identical — with
//   * the sample still executed for every event (boundary atlas gather + transfer-function lookup, result kept alive),
//   * 32 instead of 56 bytes of photon state each way: only [direction, bounces] and [radiance, samples] are read and written; the
//     position is recomputed from the direction, and the position / transmittance arrays of MISS tiles are brought up to date by
//     k_mcm_materialize before anything else looks at them (read-back, whole-image kernels, a changed matrix),
//   * no brick tables in LDS (the atlas needs none), fewer registers: 8 waves per SIMD.
// At the benchmark camera ~78 % of the tiles are MISS tiles.  HIT tiles run k_mcm_integrate from their own tile list.
// CHECK: count the events that contradict the classification (tests / VPT_OPTION_VERIFY_TILE_CLASSES; must stay 0).
// =============================================================================================
VPT_DEV const float4 *stage_tf(float4 *lds, const PassArgs &a) {
    const int nthreads = (int)blockDim.x;
    for (int t = (int)threadIdx.x; t < a.tf_w; t += nthreads) {
        float4 v = a.tf[t], n = a.tf[min(t + 1, a.tf_w - 1)];
        lds[2 * t] = v;
        lds[2 * t + 1] = make_float4(n.x - v.x, n.y - v.y, n.z - v.z, n.w - v.w);
    }
    __syncthreads();
    return lds;
}
// the executed-and-discarded sample of an out-of-cube event: texture(uVolume, clamp(p)) through the boundary atlas, then the
// transfer function (MCMRenderer.glsl:85-89,132).  Precondition (the tile class): p has a coordinate outside [0, 1].
// In two phases, like mcm_sample: miss_sample_issue puts the atlas gather in flight, miss_sample_finish blends it and looks the
// transfer function up; the event's path end (which needs the random stream only) sits between the two, under the load's latency.
struct MissLoad { uint32_t aw; float fa, fb; };
template <bool CHECK>
VPT_DEV MissLoad miss_sample_issue(const PassArgs &a, f3 q, unsigned long long *violations) {
    MissLoad l;
    l.aw = a.vol.atlas[boundary_cell(a.vol, q, l.fa, l.fb)];
    if (CHECK) {
        const bool oob = (vmax(vmax(q.x, q.y), q.z) > 1.0f) || (vmin(vmin(q.x, q.y), q.z) < 0.0f);
        if (!oob) atomicAdd(violations, 1ull);
    }
    return l;
}
VPT_DEV void miss_sample_finish(const PassArgs &a, const float4 *tf, const MissLoad &l) {
    float4 vs = sample_tf(tf, a.tf_fw, a.tf_hi, boundary_blend(l.aw, l.fa, l.fb));
    asm volatile("" : "+v"(vs.w));
}
// LATE: the sample is consumed after the path end (under whose arithmetic its load flies) instead of right where the shader samples
template <int V, bool CHECK, bool LATE>
VPT_DEV void mcm_events_miss(const PassArgs &a, const float4 *tf, Photon &ph, float px, float py, f3 from0) {
    uint32_t state = hash3(__float_as_uint(ndc_to_uv(px)), __float_as_uint(ndc_to_uv(py)), __float_as_uint(a.seed));
    for (uint32_t s = 0u; s < a.steps; s++) {
        float dist = random_exponential(state, a.inv_extinction);
        ph.position = madd3(ph.position, dist, ph.direction);
        const MissLoad l = miss_sample_issue<CHECK>(a, ph.position, a.violations);
        if (!LATE) miss_sample_finish(a, tf, l);
        random_uniform(state);                                     // the wheel draw (its value decides nothing out of bounds)
        float4 env = sample_environment(a.env, ph.direction);      // transmittance is (1, 1, 1): radiance = 1 * env, exactly env
        photon_deposit(ph, f3{ env.x, env.y, env.z });
        reset_photon<true>(state, ph, px, py, a, from0);
        if (LATE) miss_sample_finish(a, tf, l);
    }
}
template <int V, bool CHECK, bool LATE>
VPT_DEV void mcm_events_miss_fast(const PassArgs &a, const float4 *tf, const FastPixel &c, Photon &ph, float px, float py) {
    const float ld = -0.6931471805599453f * a.inv_extinction, ld32 = -32.0f * ld;
    uint32_t state = hash3(__float_as_uint(ndc_to_uv(px)), __float_as_uint(ndc_to_uv(py)), __float_as_uint(a.seed));
    for (uint32_t s = 0u; s < a.steps; s++) {
        float dist = fmaf(hw_log2(pcg_float(state)), ld, ld32);
        ph.position = madd3(ph.position, dist, ph.direction);
        const MissLoad l = miss_sample_issue<CHECK>(a, ph.position, a.violations);
        if (!LATE) miss_sample_finish(a, tf, l);
        state = pcg(state);                                        // the wheel draw
        float4 env = sample_environment(a.env, ph.direction);
        fast_path_end<true>(a, c, state, ph, f3{ env.x, env.y, env.z }, px, py);
        if (LATE) miss_sample_finish(a, tf, l);
    }
}
template <bool FUSE_RENDER, int V, bool CHECK, bool LATE>
__global__ void __launch_bounds__(VPT_BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) k_mcm_miss(PassArgs a) {
    apply_frame_table(a);
    Pix p = map_pixel(a.pm);
    float4 s1 = make_float4(0.0f, 0.0f, 1.0f, 0.0f), s3 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    f3 s0 = { 0.0f, 0.0f, 0.0f };
    if (p.tile) {
        s1 = a.st1[p.k]; s3 = a.st3[p.k];
        if (a.miss_load_pos) s0 = ((const f3 *)a.st0)[p.k];
    }
    extern __shared__ float4 lds_raw[];
    const float4 *tf = stage_tf(lds_raw, a);
    if (!p.valid) return;
    const float px = ndc_col(a.pm, p.i), py = ndc_row(a.pm, p.j);
    Photon ph;
    ph.direction = f3{ s1.x, s1.y, s1.z };
    ph.bounces = 0u;
    ph.transmittance = f3{ 1.0f, 1.0f, 1.0f };
    ph.radiance = f3{ s3.x, s3.y, s3.z };
    ph.samples = (uint32_t)(s3.w + 0.5f);
    if (V & VPT_V_FAST) {
        const FastPixel c = fast_pixel(a, px, py);
        ph.position = a.miss_load_pos ? s0 : photon_start_fast(c.from0, ph.direction);
        mcm_events_miss_fast<V & ~VPT_V_FAST, CHECK, LATE>(a, tf, c, ph, px, py);
    } else {
        const f3 from0 = unproject_near(px, py, a);
        ph.position = a.miss_load_pos ? s0 : photon_start(from0, ph.direction);
        mcm_events_miss<V, CHECK, LATE>(a, tf, ph, px, py, from0);
    }
    a.st1[p.k] = make_float4(ph.direction.x, ph.direction.y, ph.direction.z, 0.0f);
    a.st3[p.k] = make_float4(ph.radiance.x, ph.radiance.y, ph.radiance.z, (float)ph.samples);
    if (FUSE_RENDER) store_frame(a, p, pack_half4(ph.radiance.x, ph.radiance.y, ph.radiance.z, 1.0f));
}
// brings the position / transmittance arrays of the MISS tiles up to date: position = photon_start(from0, direction) in the
// arithmetic of the variant that ran the last pass, transmittance = (1, 1, 1)
template <bool FAST>
__global__ void __launch_bounds__(VPT_BLOCK) k_mcm_materialize(PassArgs a) {
    Pix p = map_pixel(a.pm);
    if (!p.valid) return;
    const float px = ndc_col(a.pm, p.i), py = ndc_row(a.pm, p.j);
    const float4 s1 = a.st1[p.k];
    const f3 dir = { s1.x, s1.y, s1.z };
    f3 pos;
    if (FAST) pos = photon_start_fast(fast_pixel(a, px, py).from0, dir);
    else pos = photon_start(unproject_near(px, py, a), dir);
    ((f3 *)a.st0)[p.k] = pos;
    ((f3 *)a.st2)[p.k] = f3{ 1.0f, 1.0f, 1.0f };
}

// `npasses` whole render() passes of one pixel in ONE launch (vpt_renderer_play, VPT_PLAY_FUSED): the photon state
// stays in registers between passes — one 64 B read + 64 B write per pixel for the whole sequence instead of per pass —
// and the launch / staging cost is paid once.  Pass f re-seeds from the f-th entry of the frame table exactly as
// launch f of the unfused sequence would; the render buffer receives the last pass's radiance, which is all that is
// left of the unfused sequence's render buffer as well.
template <int V, bool FRAMES>
VPT_DEV void mcm_multi_body(PassArgs &a, uint32_t npasses, uint2 *ring, uint32_t slot_pixels) {
    extern __shared__ float4 lds_raw[];
    LdsTables t = stage_lds<(V & VPT_V_WIDE) != 0>(lds_raw, a);
    Pix p = map_pixel(a.pm);
    if (!p.valid) return;
    float px = ndc_col(a.pm, p.i), py = ndc_row(a.pm, p.j);
    Photon ph = photon_unpack(photon_load(a, p.k));
    uint32_t base = a.frame_base;
    for (uint32_t f = 0; f < npasses; f++) {
        a.seed = a.frame_table[(base + f) & a.frame_mask].seed;
        if (V & VPT_V_FAST) mcm_events_fast<V & ~VPT_V_FAST>(a, t, ph, px, py);
        else mcm_events<V>(a, t, ph, px, py);
        // the unfused sequence stores the counters as floats between passes and re-reads them with uint(w + 0.5):
        // identical for every count below 2^24
        // VPT_PLAY_FRAMES: every pass's frame is written (slot f of the frame ring), as `npasses` render() calls would show them
        if (FRAMES) store_frame_texel(&ring[(size_t)f * slot_pixels + (size_t)p.l * a.pm.W + p.i], pack_half4(ph.radiance.x, ph.radiance.y, ph.radiance.z, 1.0f));
    }
    photon_store(a, p.k, ph);
    store_frame_texel(&a.render[(size_t)p.l * a.pm.W + p.i], pack_half4(ph.radiance.x, ph.radiance.y, ph.radiance.z, 1.0f));
}
template <int V>
__global__ void __launch_bounds__(VPT_BLOCK) __attribute__((amdgpu_waves_per_eu(7, 8))) k_mcm_multi(PassArgs a, uint32_t npasses) {
    mcm_multi_body<V, false>(a, npasses, nullptr, 0u);
}
// the same with every pass's frame written to the ring: one more live address per lane (and, since the sample's loads are issued in
// a phase of their own, two more windows in flight): compiled for 4 waves per SIMD (128 VGPRs; at 96 it spills three registers)
template <int V>
__global__ void __launch_bounds__(VPT_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 8))) k_mcm_frames(PassArgs a, uint32_t npasses, uint2 *ring, uint32_t slot_pixels) {
    mcm_multi_body<V, true>(a, npasses, ring, slot_pixels);
}
__global__ void k_advance_frames(uint32_t *counter, uint32_t n) { *counter = *counter + n; }

// ---- a bucket of frames by ONE launch per tile class (VPT_OPTION_BUCKET_KERNEL, vpt_renderer_play_into) --------------------------
// `nframes` render() passes of the tiles of one class, each pass's frame written to its slot of the caller's bucket (slot f =
// ring + f * slot_pixels): what nframes launches of k_mcm_integrate<true> / k_mcm_miss<true> with those render targets write, with
// the photon state in registers from the first pass to the last.  A rank's share of a sharded frame is a few hundred tiles: its pass
// is ~8 us of arithmetic behind ~4-5 us of launch gap, table staging and state traffic (profiles/r03_shard8.json), and a bucket of
// frames that one collective moves anyway (vpt_amd/tiles.py FrameGather) pays those once.  The frames' seeds travel BY VALUE: the two
// classes run on two streams that nothing orders against an upload of the frame table.
#define VPT_BUCKET_FRAMES 16
struct FrameSeeds { float seed[VPT_BUCKET_FRAMES]; };
// DISPLAY: the slots hold the frames as the armed tone mapper shows them (RGBA8 through its table, PassArgs.tm_table) instead of RGBA16F:
// half the bytes for the collective that moves the bucket (vpt_renderer_play_into_display)
template <bool DISPLAY>
VPT_DEV void bucket_store(const PassArgs &a, void *ring, size_t texel, uint2 v) {
    if (DISPLAY) ((uint32_t *)ring)[texel] = tone_map_texel(a.tm_table, v, nullptr);
    else store_frame_texel((uint2 *)ring + texel, v);
}
template <int V, bool EARLY, bool DISPLAY>
__global__ void __launch_bounds__(VPT_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 8)))
k_mcm_bucket_hit(PassArgs a, FrameSeeds fs, uint32_t nframes, void *ring, uint32_t slot_pixels) {
    Pix p = map_pixel(a.pm);
    PhotonState st;
    if (p.tile) st = photon_load(a, p.k);
    extern __shared__ float4 lds_raw[];
    LdsTables t = stage_lds<(V & VPT_V_WIDE) != 0>(lds_raw, a);
    if (!p.valid) return;
    const float px = ndc_col(a.pm, p.i), py = ndc_row(a.pm, p.j);
    Photon ph = photon_unpack(st);
    size_t texel = (size_t)p.l * a.pm.W + p.i;
    for (uint32_t f = 0; f < nframes; f++) {
        a.seed = fs.seed[f];
        if (V & VPT_V_FAST) {
            if (EARLY) mcm_events_fast_early<V & ~VPT_V_FAST>(a, t, ph, px, py);
            else mcm_events_fast<V & ~VPT_V_FAST>(a, t, ph, px, py);
        } else {
            if (EARLY) mcm_events_early<V>(a, t, ph, px, py);
            else mcm_events<V>(a, t, ph, px, py);
        }
        // (between two launches the counters travel as floats and come back through uint(w + 0.5): the identity below 2^24)
        bucket_store<DISPLAY>(a, ring, texel, pack_half4(ph.radiance.x, ph.radiance.y, ph.radiance.z, 1.0f));
        texel += slot_pixels;
    }
    photon_store(a, p.k, ph);
    // (VPT_PLAY_FRAMES: the render buffer shows the last frame, as after `nframes` render() calls; null for a caller's bucket)
    if (a.render) store_frame_texel(&a.render[(size_t)p.l * a.pm.W + p.i], pack_half4(ph.radiance.x, ph.radiance.y, ph.radiance.z, 1.0f));
}
template <int V, bool LATE, bool DISPLAY>     // (the contract's arithmetic keeps six more values alive across the frame loop: 6 waves per SIMD there)
__global__ void __launch_bounds__(VPT_BLOCK) __attribute__((amdgpu_waves_per_eu((V & VPT_V_FAST) ? 8 : 6, 8)))
k_mcm_bucket_miss(PassArgs a, FrameSeeds fs, uint32_t nframes, void *ring, uint32_t slot_pixels) {
    Pix p = map_pixel(a.pm);
    float4 s1 = make_float4(0.0f, 0.0f, 1.0f, 0.0f), s3 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    f3 s0 = { 0.0f, 0.0f, 0.0f };
    if (p.tile) {
        s1 = a.st1[p.k]; s3 = a.st3[p.k];
        if (a.miss_load_pos) s0 = ((const f3 *)a.st0)[p.k];
    }
    extern __shared__ float4 lds_raw[];
    const float4 *tf = stage_tf(lds_raw, a);
    if (!p.valid) return;
    const float px = ndc_col(a.pm, p.i), py = ndc_row(a.pm, p.j);
    Photon ph;
    ph.direction = f3{ s1.x, s1.y, s1.z };
    ph.bounces = 0u;
    ph.transmittance = f3{ 1.0f, 1.0f, 1.0f };
    ph.radiance = f3{ s3.x, s3.y, s3.z };
    ph.samples = (uint32_t)(s3.w + 0.5f);
    size_t texel = (size_t)p.l * a.pm.W + p.i;
    if (V & VPT_V_FAST) {
        const FastPixel c = fast_pixel(a, px, py);
        ph.position = a.miss_load_pos ? s0 : photon_start_fast(c.from0, ph.direction);
        for (uint32_t f = 0; f < nframes; f++) {
            a.seed = fs.seed[f];
            mcm_events_miss_fast<V & ~VPT_V_FAST, false, LATE>(a, tf, c, ph, px, py);
            bucket_store<DISPLAY>(a, ring, texel, pack_half4(ph.radiance.x, ph.radiance.y, ph.radiance.z, 1.0f));
            texel += slot_pixels;
        }
    } else {
        const f3 from0 = unproject_near(px, py, a);
        ph.position = a.miss_load_pos ? s0 : photon_start(from0, ph.direction);
        for (uint32_t f = 0; f < nframes; f++) {
            a.seed = fs.seed[f];
            mcm_events_miss<V, false, LATE>(a, tf, ph, px, py, from0);
            bucket_store<DISPLAY>(a, ring, texel, pack_half4(ph.radiance.x, ph.radiance.y, ph.radiance.z, 1.0f));
            texel += slot_pixels;
        }
    }
    a.st1[p.k] = make_float4(ph.direction.x, ph.direction.y, ph.direction.z, 0.0f);
    a.st3[p.k] = make_float4(ph.radiance.x, ph.radiance.y, ph.radiance.z, (float)ph.samples);
    if (a.render) store_frame_texel(&a.render[(size_t)p.l * a.pm.W + p.i], pack_half4(ph.radiance.x, ph.radiance.y, ph.radiance.z, 1.0f));
}
__global__ void __launch_bounds__(VPT_BLOCK) k_mcm_render(PassArgs a) {   // MCMRenderer.glsl:204-206
    Pix p = map_pixel(a.pm);
    if (!p.valid) return;
    float4 r = a.st3[p.k];
    a.render[(size_t)p.l * a.pm.W + p.i] = pack_half4(r.x, r.y, r.z, 1.0f);
}

// =============================================================================================
// layout helpers
// =============================================================================================
// tile-order per-pixel buffer -> row-major local rows (read-back only); elem = bytes per pixel
__global__ void __launch_bounds__(VPT_BLOCK) k_detile(PixMap pm, const uint8_t *src, uint8_t *dst, int elem) {
    Pix p = map_pixel(pm);
    if (!(p.i < pm.W && p.l < pm.local_h)) return;
    const uint8_t *s = src + (size_t)p.k * elem;
    uint8_t *d = dst + ((size_t)p.l * pm.W + p.i) * elem;
    for (int b = 0; b < elem; b++) d[b] = s[b];
}

// the MCM position / transmittance arrays (12-byte texels, tile order) -> RGBA32F rows with w = 0, as the reference's attachments hold them
__global__ void __launch_bounds__(VPT_BLOCK) k_detile_mcm3(PixMap pm, const f3 *src, float4 *dst) {
    Pix p = map_pixel(pm);
    if (!(p.i < pm.W && p.l < pm.local_h)) return;
    f3 v = src[p.k];
    dst[(size_t)p.l * pm.W + p.i] = make_float4(v.x, v.y, v.z, 0.0f);
}

// texSubImage3D: contiguous block (bw x bh x bd, `ch` interleaved bytes per voxel) -> linear volume at (x0,y0,z0)
__global__ void k_blit_block(uint8_t *vol, int nx, int ny, const uint8_t *blk, int x0, int y0, int z0, int bw, int bh, int bd, int ch) {
    size_t n = (size_t)bw * bh * bd;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        int x = (int)(t % bw); size_t r = t / bw; int y = (int)(r % bh); int z = (int)(r / bh);
        size_t dst = (((size_t)(z0 + z) * ny + (y0 + y)) * nx + (x0 + x)) * ch;
        for (int c = 0; c < ch; c++) vol[dst + c] = blk[t * ch + c];
    }
}
// linear volume -> apron bricks in Morton order; one 128-thread workgroup per brick
// (3-D grid: a 1-D grid of 2048^3's 2^27 bricks x 128 threads exceeds HIP's 2^32 work-items per dimension)
// `ch` = 1 (R8: 128-byte slots) or 2 (RG8: 256-byte slots, the R brick at +0 and the G brick at +128)
// codes = the brick-code tables CX | CY | CZ (indexed by voxel coordinate): slot(bx,by,bz) = CX[4bx] + CY[4by] + CZ[4bz]
// a workgroup re-lays VPT_BRICKIFY_RUN consecutive bricks of a brick row (one brick per workgroup: 2^21 tiny workgroups for
// 512^3, 0.74 ms = 0.5 TB/s, bound by workgroup launches)
#define VPT_BRICKIFY_RUN 16
__global__ void __launch_bounds__(128) k_brickify(const uint8_t *lin, uint8_t *bricks, int nx, int ny, int nz, int ch, const uint32_t *codes, int first_brick) {
    const int by = (int)blockIdx.y, bz = (int)blockIdx.z;
    const int t = (int)threadIdx.x;
    const int nbx = (nx + VPT_BRICK - 1) / VPT_BRICK;
    const int lx = t % 5, ly = (t / 5) % 5, lz = t / 25;
    const int y = min(by * VPT_BRICK + ly, ny - 1), z = min(bz * VPT_BRICK + lz, nz - 1);
    const size_t row = ((size_t)z * ny + y) * nx;
    const uint32_t cyz = codes[nx + 4 * by] + codes[nx + ny + 4 * bz];
    const int bx0 = first_brick + (int)blockIdx.x * VPT_BRICKIFY_RUN;
    for (int c = 0; c < ch; c++) {
        uint8_t v[VPT_BRICKIFY_RUN];
#pragma unroll
        for (int u = 0; u < VPT_BRICKIFY_RUN; u++) {
            int bx = min(bx0 + u, nbx - 1);
            int x = min(bx * VPT_BRICK + lx, nx - 1);
            v[u] = (t < 125) ? lin[(row + x) * ch + c] : (uint8_t)0;
        }
#pragma unroll
        for (int u = 0; u < VPT_BRICKIFY_RUN; u++) {
            int bx = bx0 + u;
            if (bx < nbx) {
                size_t slot = (size_t)(codes[4 * bx] + cyz) << (ch == 2 ? 8 : 7);
                bricks[slot + (size_t)c * 128 + t] = v[u];
            }
        }
    }
}

// The same re-layout for one-channel volumes whose rows are dword-aligned (nx % 4 == 0), staged through LDS: a workgroup
// takes a block of 16 x 4 x 4 bricks (64 x 16 x 16 voxels + the apron column, row and slice): 17 x 17 source rows of 17
// dwords, loaded as dwords with all of a thread's loads in flight before its first LDS write, then writes the 256 brick
// slots as 16-byte pieces in the order of the brick codes — with Z-order codes the block is four contiguous 8 KiB runs of the
// brick array (a wave instruction = eight whole consecutive slots).  No division in either loop.
#define VPT_BRICKIFY_ROWS 4       // brick rows (y) and brick layers (z) per workgroup of k_brickify_strip
__global__ void __launch_bounds__(256) k_brickify_strip(const uint8_t *lin, uint8_t *bricks, int nx, int ny, int nz, const uint32_t *codes) {
    constexpr int NR = VPT_BRICK * VPT_BRICKIFY_ROWS + 1;                             // 17 voxel rows / slices incl. the apron
    __shared__ uint32_t rows[NR * NR][17];
    const int by0 = (int)blockIdx.y * VPT_BRICKIFY_ROWS, bz0 = (int)blockIdx.z * VPT_BRICKIFY_ROWS, t = (int)threadIdx.x;
    const int nby = (ny + VPT_BRICK - 1) / VPT_BRICK, nbz = (nz + VPT_BRICK - 1) / VPT_BRICK;
    const int x0 = (int)blockIdx.x * (VPT_BRICK * VPT_BRICKIFY_RUN);                 // multiple of 64; nx % 4 == 0 guaranteed by the launch
    {   // thread t < 255 loads dword d = t % 17 of the rows t / 17 + 15 i, i = 0 .. 19 (row r = slice r / 17, voxel row r % 17)
        const int d = t % 17, r0 = t / 17;
        if (t < 255) {
            // past the row's end (the apron of the last brick column, the unused tail of a partial strip): voxel nx-1 replicated
            const bool inside = x0 + 4 * d < nx;
            uint32_t v[20];
            int ry = r0, zi = 0;
#pragma unroll
            for (int i = 0; i < 20; i++) {
                if (zi < NR) {
                    const int y = min(by0 * VPT_BRICK + ry, ny - 1), z = min(bz0 * VPT_BRICK + zi, nz - 1);
                    const uint8_t *row = lin + ((size_t)z * ny + y) * nx;
                    v[i] = inside ? *(const uint32_t *)(row + x0 + 4 * d) : (uint32_t)row[nx - 1] * 0x01010101u;
                }
                ry += 15; if (ry >= NR) { ry -= NR; zi++; }
            }
            int r = r0;
#pragma unroll
            for (int i = 0; i < 20; i++) {
                if (r < NR * NR) rows[r][d] = v[i];
                r += 15;
            }
        }
    }
    __syncthreads();
    const uint8_t *lb = (const uint8_t *)rows;
    const int w8 = t & 7, s = t >> 3;                        // 16-byte piece w8 of brick u = s + 32 it of the block
    // byte b = 16 w8 + k of a brick = voxel (lx, ly, lz), b = lx + 5 ly + 25 lz; 125..127 are padding
    const int b0 = 16 * w8, lz0 = b0 / 25, rem0 = b0 - 25 * lz0, ly0 = rem0 / 5, lx0 = rem0 - 5 * ly0;
#pragma unroll 2
    for (int it = 0; it < 8; it++) {
        // u in Z-order over the low two bits of (ux, uy, uz), then the high bits of ux: consecutive u = consecutive brick codes
        const int u = s + 32 * it;
        const int ux = (u & 1) | ((u >> 2) & 2) | ((u >> 4) & 12), uy = ((u >> 1) & 1) | ((u >> 3) & 2), uz = ((u >> 2) & 1) | ((u >> 4) & 2);
        if (x0 + 4 * ux >= nx || by0 + uy >= nby || bz0 + uz >= nbz) continue;
        const int base = ((VPT_BRICK * uz) * NR + VPT_BRICK * uy) * 68 + 4 * ux;
        uint32_t o[4] = { 0u, 0u, 0u, 0u };
        int lx = lx0, ly = ly0, lz = lz0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (b0 + k < 125) o[k >> 2] |= (uint32_t)lb[base + (lz * NR + ly) * 68 + lx] << (8 * (k & 3));
            if (++lx == 5) { lx = 0; if (++ly == 5) { ly = 0; lz++; } }
        }
        const size_t slot = (size_t)(codes[x0 + 4 * ux] + codes[nx + 4 * (by0 + uy)] + codes[nx + ny + 4 * (bz0 + uz)]) << 7;
        *(uint4 *)(bricks + slot + 16 * w8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// FLOAT volumes: linear floats -> 5^3-float apron bricks in 512-byte slots (slot = brick code << 9; two channels: 1024-byte slots,
// the G brick 512 bytes behind the R brick); VPT_BRICKIFY_RUN bricks of a brick row per workgroup, thread t < 125 carries local
// voxel t of each
__global__ void __launch_bounds__(128) k_brickify_f32(const float *lin, float *bricks, int nx, int ny, int nz, int ch, const uint32_t *codes) {
    const int by = (int)blockIdx.y, bz = (int)blockIdx.z, t = (int)threadIdx.x;
    if (t >= 125) return;
    const int nbx = (nx + VPT_BRICK - 1) / VPT_BRICK;
    const int lx = t % 5, ly = (t / 5) % 5, lz = t / 25;
    const int y = min(by * VPT_BRICK + ly, ny - 1), z = min(bz * VPT_BRICK + lz, nz - 1);
    const size_t row = ((size_t)z * ny + y) * nx;
    const uint32_t cyz = codes[nx + 4 * by] + codes[nx + ny + 4 * bz];
    const int shift = ch == 2 ? 8 : 7;                                          // floats per slot: 128 or 256
    for (int u = 0; u < VPT_BRICKIFY_RUN; u++) {
        int bx = (int)blockIdx.x * VPT_BRICKIFY_RUN + u;
        if (bx >= nbx) break;
        int x = min(bx * VPT_BRICK + lx, nx - 1);
        const size_t slot = (size_t)(codes[4 * bx] + cyz) << shift;
        for (int c = 0; c < ch; c++) bricks[slot + (size_t)c * 128 + t] = lin[(row + x) * ch + c];
    }
}

// boundary atlas (vpt_device.h sample_volume_boundary): thread c of [0, cx + cy + cz) builds cell c of the low-side AND the
// high-side face of its axis from the linear volume.  Face x: cells (a, b) = (y, z); y: (x, z); z: (x, y); face f = 2 * axis +
// side at dword f * face, cell (a, b) at (b << shift) + a.
__global__ void __launch_bounds__(256) k_build_atlas(const uint8_t *lin, uint32_t *atlas, int nx, int ny, int nz, uint32_t face, uint32_t shift) {
    size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t cx = (size_t)ny * nz, cy = (size_t)nx * nz, cz = (size_t)nx * ny;
    int axis, na, nb;
    if (c < cx) { axis = 0; na = ny; nb = nz; }
    else if (c < cx + cy) { axis = 1; c -= cx; na = nx; nb = nz; }
    else if (c < cx + cy + cz) { axis = 2; c -= cx + cy; na = nx; nb = ny; }
    else return;
    const int a = (int)(c % (size_t)na), b = (int)(c / (size_t)na);
    const int a1 = min(a + 1, na - 1), b1 = min(b + 1, nb - 1);
    const int nk = axis == 0 ? nx : (axis == 1 ? ny : nz);
    for (int side = 0; side < 2; side++) {
        const int k = side ? nk - 1 : 0;
        auto vox = [&](int p, int q) -> uint32_t {
            int x = axis == 0 ? k : p, y = axis == 0 ? p : (axis == 1 ? k : q), z = axis == 2 ? k : q;
            return lin[((size_t)z * ny + y) * nx + x];
        };
        atlas[(size_t)(2 * axis + side) * face + ((size_t)b << shift) + a] = vox(a, b) | (vox(a1, b) << 8) | (vox(a, b1) << 16) | (vox(a1, b1) << 24);
    }
}

// streaming read: every lane pulls 16 B per iteration, grid-stride; the xor keeps the loads alive
typedef unsigned int vpt_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(VPT_BLOCK) void k_stream_read(const uint4 *src_, size_t n16, uint32_t *sink) {
    const vpt_u32x4 *src = (const vpt_u32x4 *)src_;
    size_t stride = (size_t)gridDim.x * VPT_BLOCK;
    vpt_u32x4 acc = { 0u, 0u, 0u, 0u };
    size_t i = (size_t)blockIdx.x * VPT_BLOCK + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        vpt_u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
        vpt_u32x4 c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
        acc ^= a ^ b ^ c ^ d;
    }
    for (; i < n16; i += stride) acc ^= src[i];
    uint32_t v = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (v == 0x9E3779B9u) *sink = v;                      // practically never: the buffer is zero-filled
}

// probes (tests)
__global__ void k_probe_math(int which, const float *in, float *out, size_t n) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    float r = 0.0f, s, c;
    switch (which) {
        case 0: r = vpt_logf(in[t]); break;
        case 1: vpt_sincosf(in[t], s, c); r = s; break;
        case 2: vpt_sincosf(in[t], s, c); r = c; break;
        case 3: r = vpt_asinf(in[t]); break;
        case 4: r = vpt_atan2f(in[2 * t], in[2 * t + 1]); break;
        case 5: r = __uint_as_float(pcg(__float_as_uint(in[t]))); break;
        case 6: { uint32_t st = __float_as_uint(in[t]); r = random_uniform(st); } break;
        case 7: r = __uint_as_float((uint32_t)to_half_bits(in[t])); break;
        case 8: r = rcp_nr(in[t]); break;
        case 9: r = rsqrt_nr(in[t]); break;
        case 10: r = vmin(in[2 * t], in[2 * t + 1]); break;
        case 11: r = vmax(in[2 * t], in[2 * t + 1]); break;
        case 12: r = vpt_logf_uniform(in[t]); break;
        case 13: r = rcp_nrz(in[t]); break;
        case 14: r = sqrt_nr(in[t]); break;
        case 15: r = vpt_expf(in[t]); break;
        case 16: r = vpt_powf(in[2 * t], in[2 * t + 1]); break;
    }
    out[t] = r;
}
template <int V>
__global__ void __launch_bounds__(VPT_BLOCK) k_probe_sample(PassArgs a, const float *xyz, float4 *out, size_t n) {
    extern __shared__ float4 lds_raw[];
    LdsTables t = stage_lds<(V & VPT_V_WIDE) != 0>(lds_raw, a);
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = sample_volume_color<V>(a, t, f3{ xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2] });
}
