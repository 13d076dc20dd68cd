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
#ifndef VPT_UNROLL
#define VPT_UNROLL      4          // samples in flight per ray in the EAM / ISO / Depth marches (8: EAM 4 %, ISO 8 % slower —
                                   // their early exits throw the speculative samples away)
#endif
#ifndef VPT_UNROLL_MIP
#define VPT_UNROLL_MIP  8          // MIP has no early exit: 8 in flight is 3 % faster than 4
#endif
// waves per SIMD a kernel is compiled for (its register budget): VPT_WAVES_ATTR(n) = amdgpu_waves_per_eu(n, 8), n = 0 leaves the choice
// to the compiler.  The per-kernel values are measured ones (profiles/experiments.md, "occupancy")
#define VPT_WAVES_ATTR(n)  VPT_WAVES_ATTR_(n)
#define VPT_WAVES_ATTR_(n) VPT_WAVES_ATTR_##n
#define VPT_WAVES_ATTR_0
#define VPT_WAVES_ATTR_2 __attribute__((amdgpu_waves_per_eu(2, 8)))
#define VPT_WAVES_ATTR_3 __attribute__((amdgpu_waves_per_eu(3, 8)))
#define VPT_WAVES_ATTR_4 __attribute__((amdgpu_waves_per_eu(4, 8)))
#define VPT_WAVES_ATTR_5 __attribute__((amdgpu_waves_per_eu(5, 8)))
#define VPT_WAVES_ATTR_6 __attribute__((amdgpu_waves_per_eu(6, 8)))
#define VPT_WAVES_ATTR_7 __attribute__((amdgpu_waves_per_eu(7, 8)))
#define VPT_WAVES_ATTR_8 __attribute__((amdgpu_waves_per_eu(8, 8)))
#ifndef VPT_MIP_WAVES
#define VPT_MIP_WAVES 6
#endif
#ifndef VPT_EAM_WAVES
#define VPT_EAM_WAVES 0
#endif
#ifndef VPT_MCS_WAVES
#define VPT_MCS_WAVES 0
#endif
#ifndef VPT_DEPTH_WAVES
#define VPT_DEPTH_WAVES 0
#endif
#ifndef VPT_LAO_WAVES
#define VPT_LAO_WAVES 0
#endif
#ifndef VPT_DOS_WAVES
#define VPT_DOS_WAVES 0
#endif

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
    uint32_t miss_verify;        // k_mcm_miss of the other volume formats: 1 = count the events inside the cube (VPT_OPTION_VERIFY_TILE_CLASSES; a run-time flag there)
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
#define VPT_BUCKET_FRAMES 16     // frames one launch of the bucket kernels holds (vpt_kernels_mcm.h; VPT_OPTION_BUCKET_KERNEL)
struct FrameVar { float seed, offset, mix, pad0; float lx, ly, lz, pad1; };   // the uniforms that change per frame
VPT_DEV void apply_frame_table(PassArgs &a) {
    if (a.frame_table) {
        FrameVar v = a.frame_table[*a.frame_counter & a.frame_mask];
        a.seed = v.seed; a.offset = v.offset; a.mix = v.mix;
        a.light = f3{ v.lx, v.ly, v.lz };
    }
}
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
// REC: the two column tables RX | RY of the column records (vpt_device.h record_addr) in place of the three brick tables
template <bool WIDE, bool REC = false>
VPT_DEV LdsTables stage_lds(float4 *lds, const PassArgs &a) {
    const int nthreads = (int)blockDim.x;
    for (int t = (int)threadIdx.x; t < a.tf_w; t += nthreads) {
        float4 v = a.tf[t], n = a.tf[min(t + 1, a.tf_w - 1)];
        lds[2 * t] = v;
        lds[2 * t + 1] = make_float4(n.x - v.x, n.y - v.y, n.z - v.z, n.w - v.w);
    }
    int ntab = REC ? a.vol.nx + a.vol.ny : a.vol.nx + a.vol.ny + a.vol.nz;
    uint32_t *tab = (uint32_t *)(lds + 2 * a.tf_w);
    const uint32_t *src = REC ? (WIDE ? a.vol.rtabc : a.vol.rtab32) : (WIDE ? a.vol.tabc : a.vol.tab32);
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

// an RGBA8 accumulator texel as the RGBA16F render texel (EAM / LAO _renderFrame)
VPT_DEV uint2 eam_to_half4(uint32_t q) {   // render: EAMRenderer.glsl:151-153
    return pack_half4(from_unorm8(q & 0xffu), from_unorm8((q >> 8) & 0xffu),
                      from_unorm8((q >> 16) & 0xffu), from_unorm8(q >> 24));
}
