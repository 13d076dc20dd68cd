// vpt_hip.hip — C-ABI implementation (include/vpt.h) over the HIP kernels in vpt_kernels.h.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
//        -fno-gpu-flush-denormals-to-zero -shared -fPIC (see vpt_amd/csrc/Makefile)
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_ext.h>
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "../../include/vpt.h"
#include <chrono>
#include "vpt_kernels.h"
#include "vpt_kernels_iso_depth.h"
#include "vpt_srgb_lut.h"

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, ...) {
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    return fail(VPT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define VPT_TRY(expr) do { int r_ = (expr); if (r_ != VPT_OK) return r_; } while (0)

extern "C" const char *vpt_last_error(void) { return g_err; }
extern "C" const char *vpt_version(void) { return "vpt-mi355x 0.1 (gfx950)"; }

// ---------------------------------------------------------------------------------------------
// objects
// ---------------------------------------------------------------------------------------------
struct vpt_tonemapper;
struct vpt_context {
    int device;
    hipStream_t stream;
    bool owns_stream;
    std::vector<vpt_tonemapper *> tonemappers;   // live tone mappers: a destroyed renderer is unbound from them
    std::vector<struct vpt_renderer *> renderers; // live renderers: a destroyed volume is unbound from them
};

struct vpt_volume {
    vpt_context *ctx;
    int nx, ny, nz;
    int channels;          // 1 = R8 / R32F, 2 = RG8 (interleaved)
    bool f32;              // FLOAT texels (VPT_FORMAT_R32F): 4 bytes per voxel, 512-byte brick slots
    int vox_bytes;         // bytes per voxel of the linear storage: channels * (f32 ? 4 : 1)
    int filter;
    uint8_t *linear;       // nx*ny*nz*channels, the "texture storage" blocks are uploaded into
    uint8_t *bricks;       // apron bricks, Morton order
    size_t brick_bytes;
    uint32_t *tab32;       // separable brick-offset tables TX | TY | TZ (vpt_device.h), 32-bit form
    uint32_t *tabc;        // brick Morton codes (always built; used when brick_bytes > 4 GiB, vpt_device.h cell_addr<WIDE>)
    bool wide;
    bool dirty;            // blocks uploaded since the last brickify
    bool any_upload;
    uint8_t *staging; size_t staging_bytes;
    uint32_t *atlas;       // boundary atlas: the six outer voxel planes as 2 x 2-footprint dwords (vpt_device.h sample_volume_boundary); one-channel volumes
    size_t atlas_dwords;
    uint32_t atlas_face, atlas_shift;   // dwords per face image (row pitch x rows), log2 of the row pitch
};

// Tile classes (vpt_kernels.h, "Tile classes"): per reset the host sorts the 16x16 tiles into those none of whose camera rays
// can meet the cube (MISS) and the rest (HIT).  While every pass uses the reset's matrix (and blur == 0) a MISS tile's photons
// never enter the cube, so its passes run k_mcm_miss on 32 B of state; `stale` says that the position / transmittance arrays
// of the MISS tiles are behind and k_mcm_materialize must run before anything but k_mcm_miss looks at them.
struct TileClasses {
    bool enabled, verify;          // VPT_OPTION_TILE_CLASSES (default on), VPT_OPTION_VERIFY_TILE_CLASSES
    bool valid;                    // the lists describe `mvp` for the present geometry, and every pass since that reset used it
    float mvp[16];
    uint32_t *list; int capacity;  // device: n_hit HIT tiles, then n_miss MISS tiles, each tx | ty << 16
    int n_hit, n_miss;
    bool stale, stale_fast;        // MISS tiles' position / transmittance arrays are behind; the pass that left them ran the fast variant
    unsigned long long *violations;
    // the accumulating ray marchers (MIP, EAM, ISO, MCS, Depth): see marcher_track
    uint64_t passes, fused_passes; // generate / fused passes since the reset
    bool poisoned;                 // a pass since the reset used another matrix than the first: nothing can be skipped until the next reset
    bool first_mix_one;            // the first pass since the reset was a fused pass with mix == 1 (MCS, Depth: accumulator = frame exactly)
    bool list_now;                 // the launch being enqueued covers the HIT tiles only
    bool reset_seen;               // vpt_renderer_reset has run on the present buffers (zero-filled buffers are not a reset)
};
struct vpt_renderer {
    vpt_context *ctx;
    int kind;
    int W, H;
    int G, g, R;
    int local_h;
    int tiles_x, tiles_y, ntiles;
    size_t npix_padded;     // ntiles * 256
    uint64_t valid_pixels;  // owned pixels inside the image
    vpt_volume *vol;
    float4 *tf; int tf_w, tf_h;
    float4 *env; int env_w, env_h; float4 env_const; bool env_opaque;   // env_opaque: every texel's alpha is 255
    void *frame, *acc;
    float4 *st[4];
    uint2 *render;
    uint2 *render_target;          // caller-owned redirect of the render buffer (or null)
    uint2 *frame_ring; int ring_frames;   // VPT_PLAY_FRAMES: VPT_FRAME_SLOTS frames of W x local_h RGBA16F (allocated on first use); frames of the last call
    float *ndc_x, *ndc_y;          // pixel-centre NDC tables (W and H entries)
    FrameVar *frame_table; FrameVar *frame_staging; uint32_t *frame_counter;   // device ring of per-frame uniforms + pinned staging
    uint64_t frames_played;        // host copy of the monotonic device frame counter
    bool warmed;                   // at least one eager fused render() has run (lazy allocations done)
    struct PlayGraph *play_graph;  // cached hipGraph of a frame sequence
    uint32_t *work_counter;        // tile counter of the persistent MCS kernel
    bool mcs_persistent;           // use k_mcs_persist (active-ray compaction) for the MCS generate pass
    LaoParams lao;                 // LAO renderer parameters (vpt_renderer_set_lao_params; defaults LAORenderer.js:17-108)
    float2 *dos_samples; int dos_nsamples;   // DOS: uOcclusionSamples (vpt_renderer_set_occlusion_samples)
    int dos_rect[4]; bool dos_rect_valid;   // DOS: tile rectangle [x0, y0, x1, y1) of the previous integrate call (see dos_tile_rect)
    int dos_cur;                   // DOS: which of the occlusion buffers st[2|3] holds the latest slice (colour: st[0], in place)
    // VPT_OPTION_SPLIT_STREAMS = K: the MCM pass is launched as K tile-row ranges, all but the first on private side streams.  A
    // pixel's pass depends on its own previous pass only, so the ranges never wait for each other: the launch gap, ramp and tail
    // of one overlap the body of the others.  Every other entry point joins the side streams into the context's stream first.
    bool target_is_callers;        // render_target was set by vpt_renderer_set_render_target (not by the gather pipeline)
    bool no_split;                 // set while a frame sequence is being captured into a hipGraph (one stream only)
    bool split_callers;            // VPT_OPTION_SPLIT_CALLER_TARGETS: such passes are split too, the caller joins (vpt_renderer_join)
    int last_ranges;               // how many tile-row ranges (streams) the last sampling launch used
    hipEvent_t *stop_events;       // gather pipeline: event i is attached to range i's launch (hipExtLaunchKernel stop event: the
    bool stop_used;                // dispatch packet's own completion signal, no barrier packet behind the kernel)
    int split; hipStream_t side[VPT_MAX_SPLIT - 1]; hipEvent_t ev_fork, ev_join[VPT_MAX_SPLIT - 1]; bool side_busy, main_dirty;
    int boundary_atlas;            // VPT_OPTION_BOUNDARY_ATLAS (default 1): MCM takes out-of-cube samples from the volume's boundary atlas
    int fast_math;                 // VPT_OPTION_FAST_MATH: MCM events with hardware rcp / rsq / log / sin / cos (k_mcm_integrate<.., V | VPT_V_FAST>)
    int mcm_persistent;            // 0: k_mcm_integrate; 1: k_mcm_persist; 2: k_mcm_persist with next-segment prefetch // (persistent waves, state prefetch) for the MCM integrate pass
    struct TileClasses cls;        // MCM: HIT / MISS tile lists of the last reset's matrix (see classify_tiles)
    int last_layout;               // how the last sampling launch mapped tiles to streams: 0 = tile-row ranges, 1 = tile lists
    // tone mapping fused into the fused passes' frame store: the armed tone mapper (null: none), whether its output holds the tone-mapped
    // image of what the render buffer holds now, and the store's arguments (PassArgs.tm_*)
    struct vpt_tonemapper *tm_owner; bool tm_valid; const uint8_t *tm_table; uint32_t *tm_out; int tm_mode;
    uint64_t bucket_launches;      // buckets of frames run by k_mcm_bucket_* so far (vpt_renderer_bucket_launches)
    bool bucket_kernel;            // VPT_OPTION_BUCKET_KERNEL: vpt_renderer_play_into runs a bucket's frames by one launch per tile class
    int hit_form;                  // VPT_OPTION_HIT_KERNEL_FORM: 0 = by the number of HIT tiles, 1 = k_mcm_integrate, 2 = k_mcm_integrate_early
    unsigned long long *samples;   // device counter (MIP/EAM/MCS)
    uint64_t samples_host;         // analytic part (MCM)
    void *scratch; size_t scratch_bytes;
    bool profiling;
    int profile_every; uint64_t profile_seq;   // time every n-th launch of the dominant kernel
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    std::vector<uint32_t> event_launches;   // kernel launches covered by each event pair (1, or the frames of a graph replay)
    size_t events_used;
    // the same around the first launch a pass puts on a SIDE stream (tile classes: the MISS-tile kernel), for the passes `events` samples
    std::vector<std::pair<hipEvent_t, hipEvent_t>> side_events; size_t side_events_used; bool timed_now;
};

static const size_t COUNTER_BYTES = (size_t)VPT_COUNTER_SLOTS * VPT_COUNTER_STRIDE * sizeof(unsigned long long);

struct PlayGraph;
static void play_graph_free(PlayGraph *g);

static size_t frame_elem(int kind) {
    switch (kind) {
        case VPT_RENDERER_MIP: return 1;
        case VPT_RENDERER_EAM: return 4;
        case VPT_RENDERER_MCS: return 16;
        case VPT_RENDERER_ISO: return 8;      // RGBA16F (ISORenderer.js:165-197)
        case VPT_RENDERER_DEPTH: return 4;    // R32F (DepthRenderer.js:165-189)
        case VPT_RENDERER_LAO: return 4;      // RGBA8 (LAORenderer.js:217-243)
        case VPT_RENDERER_DOS: return 0;      // colour RGBA32F (st[0]) + occlusion R32F double-buffered (st[2], st[3]), ROW-MAJOR (DOSRenderer.js:273-313)
        default: return 0;
    }
}

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
extern "C" int vpt_device_count(int *count) {
    if (!count) return fail(VPT_ERR_INVALID, "count is null");
    HIP_TRY(hipGetDeviceCount(count));
    return VPT_OK;
}
extern "C" int vpt_context_create(int device, vpt_context **out) {
    if (!out) return fail(VPT_ERR_INVALID, "out is null");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(VPT_ERR_INVALID, "device %d out of range (%d devices)", device, n);
    HIP_TRY(hipSetDevice(device));
    vpt_context *c = new vpt_context();
    c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fail(VPT_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    c->owns_stream = true;
    *out = c;
    return VPT_OK;
}
extern "C" int vpt_context_create_on_stream(int device, void *hip_stream, vpt_context **out) {
    if (!out) return fail(VPT_ERR_INVALID, "out is null");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(VPT_ERR_INVALID, "device %d out of range (%d devices)", device, n);
    HIP_TRY(hipSetDevice(device));
    vpt_context *c = new vpt_context();
    c->device = device;
    c->stream = (hipStream_t)hip_stream;
    c->owns_stream = false;
    *out = c;
    return VPT_OK;
}
extern "C" int vpt_context_destroy(vpt_context *c) {
    if (!c) return VPT_OK;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    if (c->owns_stream) hipStreamDestroy(c->stream);
    delete c;
    return VPT_OK;
}
static int join_side(struct vpt_renderer *r);
extern "C" int vpt_context_synchronize(vpt_context *c) {
    if (!c) return fail(VPT_ERR_INVALID, "context is null");
    HIP_TRY(hipSetDevice(c->device));
    for (vpt_renderer *r : c->renderers) VPT_TRY(join_side(r));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VPT_OK;
}

// ---------------------------------------------------------------------------------------------
// volume — Volume.js:31-78
// ---------------------------------------------------------------------------------------------
extern "C" int vpt_volume_create(vpt_context *c, int w, int h, int d, int format, vpt_volume **out) {
    if (!c || !out) return fail(VPT_ERR_INVALID, "null argument");
    if (format != VPT_FORMAT_R8 && format != VPT_FORMAT_RG8 && format != VPT_FORMAT_R32F && format != VPT_FORMAT_RG32F) return fail(VPT_ERR_UNSUPPORTED, "Unknown volume datatype: %d", format);  // Volume.js:103
    if (w < 1 || h < 1 || d < 1 || w > 4096 || h > 4096 || d > 4096)
        return fail(VPT_ERR_INVALID, "volume dimensions %dx%dx%d out of range [1,4096]", w, h, d);
    HIP_TRY(hipSetDevice(c->device));
    vpt_volume *v = new vpt_volume();
    memset(v, 0, sizeof(*v));
    v->ctx = c; v->nx = w; v->ny = h; v->nz = d;
    v->channels = (format == VPT_FORMAT_RG8 || format == VPT_FORMAT_RG32F) ? 2 : 1;
    v->f32 = format == VPT_FORMAT_R32F || format == VPT_FORMAT_RG32F;
    v->vox_bytes = v->channels * (v->f32 ? 4 : 1);
    // RG8: 256-byte slots (R brick at +0, G brick at +128); R32F: 512-byte slots; RG32F: 1024-byte slots (G brick at +512)
    const int slot_shift = (v->f32 ? 9 : 7) + (v->channels == 2 ? 1 : 0);
    const uint64_t eb = v->f32 ? 4 : 1;                  // bytes per texel channel
    v->filter = VPT_FILTER_LINEAR;                       // Volume.js:53-54
    int nbx = (w + 3) / 4, nby = (h + 3) / 4, nbz = (d + 3) / 4;
    // Z-order over the bricks with exactly as many bits per axis as the axis needs: the low bits of x, y, z interleave
    // (x lowest, as the classic Morton code), and once an axis runs out of bits the longer axes continue alone — a cube
    // gets the classic code, a 4096 x 2 x 3 volume 2^10 slots instead of 2^30.  code(bx,by,bz) = CX[bx] | CY[by] | CZ[bz].
    int nbits[3] = { 0, 0, 0 };
    { int nb[3] = { nbx, nby, nbz }; for (int ax = 0; ax < 3; ax++) while ((1 << nbits[ax]) < nb[ax]) nbits[ax]++; }
    int bitpos[3][16]; int total_bits = 0;
    for (int level = 0; level < 16; level++)
        for (int ax = 0; ax < 3; ax++) if (level < nbits[ax]) bitpos[ax][level] = total_bits++;
    auto axis_code = [&](int ax, uint32_t b) { uint64_t c = 0; for (int k = 0; k < nbits[ax]; k++) c |= (uint64_t)((b >> k) & 1u) << bitpos[ax][k]; return c; };
    size_t max_slot = (size_t)(axis_code(0, (uint32_t)nbx - 1) | axis_code(1, (uint32_t)nby - 1) | axis_code(2, (uint32_t)nbz - 1));
    v->brick_bytes = (max_slot + 1) << slot_shift;
    hipError_t e = hipMalloc(&v->linear, (size_t)w * h * d * v->vox_bytes);
    if (e == hipSuccess) e = hipMalloc(&v->bricks, v->brick_bytes + 64);   // +64: the 8-byte tap windows end <= byte 125+7
    if (e == hipSuccess && v->channels == 1 && !v->f32) {
        // boundary atlas: six face images (axis x: ny x nz cells, y: nx x nz, z: nx x ny; low side, high side) with one common
        // power-of-two row pitch and one common size, one dword per cell
        int pitch = 1, shift = 0;
        while (pitch < std::max(w, h)) { pitch <<= 1; shift++; }
        v->atlas_shift = (uint32_t)shift;
        v->atlas_face = (uint32_t)pitch * (uint32_t)std::max(h, d);
        v->atlas_dwords = 6 * (size_t)v->atlas_face;
        e = hipMalloc(&v->atlas, v->atlas_dwords * 4);
    }
    if (e != hipSuccess) {
        if (v->atlas) hipFree(v->atlas);
        if (v->bricks) hipFree(v->bricks);
        if (v->linear) hipFree(v->linear);
        delete v;
        return fail(VPT_ERR_HIP, "hipMalloc volume %dx%dx%d: %s", w, h, d, hipGetErrorString(e));
    }
    HIP_TRY(hipMemsetAsync(v->linear, 0, (size_t)w * h * d * v->vox_bytes, c->stream));   // texStorage3D zero-initialises
    {   // offset tables: off(x,y,z) = TX[x] + TY[y] + TZ[z]
        std::vector<uint64_t> t64((size_t)w + h + d);
        std::vector<uint32_t> t32(t64.size());
        for (int i = 0; i < w; i++) t64[i] = (axis_code(0, (uint32_t)i >> 2) << slot_shift) + (uint64_t)(i & 3) * eb;
        for (int i = 0; i < h; i++) t64[(size_t)w + i] = (axis_code(1, (uint32_t)i >> 2) << slot_shift) + (uint64_t)(i & 3) * 5 * eb;
        for (int i = 0; i < d; i++) t64[(size_t)w + h + i] = (axis_code(2, (uint32_t)i >> 2) << slot_shift) + (uint64_t)(i & 3) * 25 * eb;
        for (size_t i = 0; i < t64.size(); i++) t32[i] = (uint32_t)t64[i];
        for (size_t i = 0; i < t64.size(); i++) t64[i] >>= slot_shift;     // the brick's Morton code alone (WIDE variant)
        std::vector<uint32_t> tc(t64.size());
        for (size_t i = 0; i < t64.size(); i++) tc[i] = (uint32_t)t64[i];
        v->wide = v->brick_bytes > 0xffffffffull;
        HIP_TRY(hipMalloc(&v->tab32, t32.size() * 4));
        HIP_TRY(hipMalloc(&v->tabc, tc.size() * 4));
        HIP_TRY(hipMemcpy(v->tab32, t32.data(), t32.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(v->tabc, tc.data(), tc.size() * 4, hipMemcpyHostToDevice));
    }
    v->dirty = true;
    *out = v;
    return VPT_OK;
}
static int volume_upload(vpt_volume *v, int x, int y, int z, int w, int h, int d, const void *data, size_t nbytes, bool on_device) {
    if (!v || !data) return fail(VPT_ERR_INVALID, "null argument");
    if (w < 1 || h < 1 || d < 1 || x < 0 || y < 0 || z < 0 || x + w > v->nx || y + h > v->ny || z + d > v->nz)
        return fail(VPT_ERR_INVALID, "block (%d,%d,%d)+(%d,%d,%d) outside volume %dx%dx%d", x, y, z, w, h, d, v->nx, v->ny, v->nz);
    size_t need = (size_t)w * h * d * v->vox_bytes;
    if (nbytes < need) return fail(VPT_ERR_INVALID, "block data too short: %zu < %zu", nbytes, need);
    vpt_context *c = v->ctx;
    HIP_TRY(hipSetDevice(c->device));
    for (vpt_renderer *r : c->renderers) if (r->vol == v) VPT_TRY(join_side(r));
    bool full_xy = (x == 0 && y == 0 && w == v->nx && h == v->ny);
    if (full_xy) {   // contiguous run of z-slices (RAWReader.js:47-63 produces exactly these)
        uint8_t *dst = v->linear + (size_t)z * v->nx * v->ny * v->vox_bytes;
        HIP_TRY(hipMemcpyAsync(dst, data, need, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    } else {
        const uint8_t *src = (const uint8_t *)data;
        if (!on_device) {
            if (v->staging_bytes < need) {
                if (v->staging) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(v->staging)); v->staging = nullptr; }
                HIP_TRY(hipMalloc(&v->staging, need));
                v->staging_bytes = need;
            }
            HIP_TRY(hipMemcpyAsync(v->staging, data, need, hipMemcpyHostToDevice, c->stream));
            src = v->staging;
        }
        int grid = (int)((need / v->vox_bytes + 255) / 256); if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(k_blit_block, dim3(grid), dim3(256), 0, c->stream, v->linear, v->nx, v->ny, src, x, y, z, w, h, d, v->vox_bytes);
        HIP_TRY(hipGetLastError());
    }
    if (!on_device) HIP_TRY(hipStreamSynchronize(c->stream));   // host buffer may be released by the caller
    v->dirty = true; v->any_upload = true;
    return VPT_OK;
}
extern "C" int vpt_volume_upload_block(vpt_volume *v, int x, int y, int z, int w, int h, int d, const void *data, size_t nbytes) {
    return volume_upload(v, x, y, z, w, h, d, data, nbytes, false);
}
extern "C" int vpt_volume_upload_block_device(vpt_volume *v, int x, int y, int z, int w, int h, int d, const void *data, size_t nbytes) {
    return volume_upload(v, x, y, z, w, h, d, data, nbytes, true);
}
extern "C" int vpt_volume_finalize(vpt_volume *v) {
    if (!v) return fail(VPT_ERR_INVALID, "volume is null");
    if (!v->dirty) return VPT_OK;
    vpt_context *c = v->ctx;
    HIP_TRY(hipSetDevice(c->device));
    int nbx = (v->nx + 3) / 4, nby = (v->ny + 3) / 4, nbz = (v->nz + 3) / 4;
    if (nby > 65535 || nbz > 65535) return fail(VPT_ERR_UNSUPPORTED, "too many bricks");
    const int strips = (nbx + VPT_BRICKIFY_RUN - 1) / VPT_BRICKIFY_RUN;
    // one-channel volumes with dword-aligned rows go through the LDS-staged kernel (dword loads and stores)
    int fast = (v->channels == 1 && v->nx % 4 == 0) ? strips : 0;
    if (v->f32) {
        hipLaunchKernelGGL(k_brickify_f32, dim3((unsigned)strips, (unsigned)nby, (unsigned)nbz), dim3(128), 0, c->stream, (const float *)v->linear, (float *)v->bricks, v->nx, v->ny, v->nz, v->channels, v->tabc);
        fast = strips;                                    // nothing left for the byte kernels
    } else if (fast > 0)
        hipLaunchKernelGGL(k_brickify_strip, dim3((unsigned)fast, (unsigned)((nby + VPT_BRICKIFY_ROWS - 1) / VPT_BRICKIFY_ROWS), (unsigned)((nbz + VPT_BRICKIFY_ROWS - 1) / VPT_BRICKIFY_ROWS)), dim3(256), 0, c->stream, v->linear, v->bricks, v->nx, v->ny, v->nz, v->tabc);
    if (fast < strips)
        hipLaunchKernelGGL(k_brickify, dim3((unsigned)(strips - fast), (unsigned)nby, (unsigned)nbz), dim3(128), 0, c->stream, v->linear, v->bricks, v->nx, v->ny, v->nz, v->channels, v->tabc, fast * VPT_BRICKIFY_RUN);
    if (v->atlas) {
        size_t cells = (size_t)v->ny * v->nz + (size_t)v->nx * v->nz + (size_t)v->nx * v->ny;
        hipLaunchKernelGGL(k_build_atlas, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, c->stream, v->linear, v->atlas, v->nx, v->ny, v->nz,
                           v->atlas_face, v->atlas_shift);
    }
    HIP_TRY(hipGetLastError());
    v->dirty = false;
    return VPT_OK;
}
extern "C" int vpt_volume_set_filter(vpt_volume *v, int filter) {
    if (!v) return fail(VPT_ERR_INVALID, "volume is null");
    v->filter = (filter == VPT_FILTER_LINEAR) ? VPT_FILTER_LINEAR : VPT_FILTER_NEAREST;   // Volume.js:121
    return VPT_OK;
}
extern "C" int vpt_volume_set_wide_tables(vpt_volume *v, int wide) {
    if (!v) return fail(VPT_ERR_INVALID, "volume is null");
    if (!wide && v->brick_bytes > 0xffffffffull) return fail(VPT_ERR_INVALID, "bricked layout exceeds 4 GiB: 64-bit offset tables are required");
    v->wide = wide != 0;
    return VPT_OK;
}
extern "C" int vpt_volume_bricked_bytes(vpt_volume *v, uint64_t *n) {
    if (!v || !n) return fail(VPT_ERR_INVALID, "null argument");
    *n = v->brick_bytes;
    return VPT_OK;
}
static void renderers_unbind(vpt_context *c, vpt_volume *v);
extern "C" int vpt_volume_destroy(vpt_volume *v) {
    if (!v) return VPT_OK;
    hipSetDevice(v->ctx->device);
    for (vpt_renderer *r : v->ctx->renderers) if (r->vol == v) join_side(r);
    hipStreamSynchronize(v->ctx->stream);
    renderers_unbind(v->ctx, v);              // a renderer still bound to it reports "no ready volume" instead of reading freed memory
    if (v->linear) hipFree(v->linear);
    if (v->bricks) hipFree(v->bricks);
    if (v->atlas) hipFree(v->atlas);
    if (v->staging) hipFree(v->staging);
    if (v->tab32) hipFree(v->tab32);
    if (v->tabc) hipFree(v->tabc);
    delete v;
    return VPT_OK;
}

// ---------------------------------------------------------------------------------------------
// renderer
// ---------------------------------------------------------------------------------------------
static void renderer_free_buffers(vpt_renderer *r) {
    if (r->frame) hipFree(r->frame);
    if (r->acc) hipFree(r->acc);
    for (int i = 0; i < 4; i++) if (r->st[i]) hipFree(r->st[i]);
    if (r->render) hipFree(r->render);
    if (r->scratch) hipFree(r->scratch);
    if (r->ndc_x) hipFree(r->ndc_x);
    if (r->ndc_y) hipFree(r->ndc_y);
    r->ndc_x = r->ndc_y = nullptr;
    r->frame = r->acc = nullptr; r->render = nullptr; r->scratch = nullptr; r->scratch_bytes = 0;
    for (int i = 0; i < 4; i++) r->st[i] = nullptr;
}
// _rebuildBuffers: AbstractRenderer.js:78-92 (+ the per-renderer buffer specs)
static int renderer_alloc_buffers(vpt_renderer *r) {
    vpt_context *c = r->ctx;
    if (r->play_graph) { hipStreamSynchronize(c->stream); play_graph_free(r->play_graph); r->play_graph = nullptr; }
    r->render_target = nullptr; r->target_is_callers = false;   // an external target was sized for the old geometry
    r->tm_owner = nullptr; r->tm_valid = false; r->tm_mode = 0;  // a fused tone mapper's output was sized for it too: it re-arms itself
    if (r->frame_ring) { hipFree(r->frame_ring); r->frame_ring = nullptr; } r->ring_frames = 0;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    renderer_free_buffers(r);
    r->cls.valid = false; r->cls.stale = false;            // new geometry, zeroed state: classes come back with the next reset
    r->cls.passes = 0; r->cls.fused_passes = 0; r->cls.reset_seen = false;   // (zeroed buffers are not a reset: nothing is skipped before one)
    r->dos_cur = 0; r->dos_rect_valid = false;
    int nblocks = (r->H + r->R - 1) / r->R;                 // row blocks in the image
    int mine = (nblocks - r->g + r->G - 1) / r->G;          // blocks b with b % G == g
    int max_blocks = (nblocks + r->G - 1) / r->G;           // every rank pads to this (equal-size gather)
    if (mine < 0) mine = 0;
    r->local_h = max_blocks * r->R;
    if (r->G == 1) r->local_h = r->H;
    r->tiles_x = (r->W + VPT_TILE - 1) / VPT_TILE;
    r->tiles_y = (r->local_h + VPT_TILE - 1) / VPT_TILE;
    r->ntiles = r->tiles_x * r->tiles_y;
    r->npix_padded = (size_t)r->ntiles * VPT_BLOCK;
    uint64_t valid = 0;
    for (int l = 0; l < r->local_h; l++) {
        int lb = l / r->R; int j = (lb * r->G + r->g) * r->R + (l - lb * r->R);
        if (j < r->H) valid += (uint64_t)r->W;
    }
    r->valid_pixels = valid;
    size_t fe = frame_elem(r->kind);
    if (fe) {
        HIP_TRY(hipMalloc(&r->frame, r->npix_padded * fe));
        HIP_TRY(hipMalloc(&r->acc, r->npix_padded * fe));
        HIP_TRY(hipMemsetAsync(r->frame, 0, r->npix_padded * fe, c->stream));
        HIP_TRY(hipMemsetAsync(r->acc, 0, r->npix_padded * fe, c->stream));
    } else {
        for (int i = 0; i < 4; i++) {
            // MCM: position (0) and transmittance (2) are 12-byte texels — their fourth float is a constant 0 in the reference's
            // attachments (MCMRenderer.glsl:168,170) and is not stored; DOS keeps float4 / float arrays in the same slots
            size_t texel = (r->kind == VPT_RENDERER_MCM && (i == 0 || i == 2)) ? 3 * sizeof(float) : sizeof(float4);
            HIP_TRY(hipMalloc(&r->st[i], r->npix_padded * texel));
            HIP_TRY(hipMemsetAsync(r->st[i], 0, r->npix_padded * texel, c->stream));
        }
    }
    {   // pixel-centre NDC: fl(fl((2i+1)/W) - 1), the exact per-pixel expression of the contract
        std::vector<float> nx((size_t)r->W), ny((size_t)r->H);
        for (int i = 0; i < r->W; i++) nx[i] = (float)(2 * i + 1) / (float)r->W - 1.0f;
        for (int j = 0; j < r->H; j++) ny[j] = (float)(2 * j + 1) / (float)r->H - 1.0f;
        HIP_TRY(hipMalloc(&r->ndc_x, nx.size() * sizeof(float)));
        HIP_TRY(hipMalloc(&r->ndc_y, ny.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(r->ndc_x, nx.data(), nx.size() * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(r->ndc_y, ny.data(), ny.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    size_t rb = (size_t)r->W * r->local_h * sizeof(uint2);
    HIP_TRY(hipMalloc(&r->render, rb));
    HIP_TRY(hipMemsetAsync(r->render, 0, rb, c->stream));
    return VPT_OK;
}
static int upload_table(vpt_context *c, float4 **dst, const std::vector<float4> &host) {
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (*dst) { HIP_TRY(hipFree(*dst)); *dst = nullptr; }
    HIP_TRY(hipMalloc(dst, host.size() * sizeof(float4)));
    HIP_TRY(hipMemcpy(*dst, host.data(), host.size() * sizeof(float4), hipMemcpyHostToDevice));
    return VPT_OK;
}
extern "C" int vpt_renderer_set_transfer_function(vpt_renderer *r, const uint8_t *rgba, int w, int h) {
    if (!r || !rgba) return fail(VPT_ERR_INVALID, "null argument");
    VPT_TRY(join_side(r));
    if (w < 1 || h < 1 || w > 2048 || h > 4096) return fail(VPT_ERR_INVALID, "transfer function size %dx%d out of range", w, h);
    // SRGB8_ALPHA8: rgb decoded before filtering, alpha linear (AbstractRenderer.js:36,99-104)
    std::vector<float4> t((size_t)w * h);
    for (size_t i = 0; i < t.size(); i++)
        t[i] = make_float4(VPT_SRGB_TO_LINEAR[rgba[4 * i]], VPT_SRGB_TO_LINEAR[rgba[4 * i + 1]],
                           VPT_SRGB_TO_LINEAR[rgba[4 * i + 2]], (float)rgba[4 * i + 3] / 255.0f);
    VPT_TRY(upload_table(r->ctx, &r->tf, t));
    r->tf_w = w; r->tf_h = h;
    return VPT_OK;
}
extern "C" int vpt_renderer_set_environment(vpt_renderer *r, const uint8_t *rgba, int w, int h) {
    if (!r || !rgba) return fail(VPT_ERR_INVALID, "null argument");
    VPT_TRY(join_side(r));
    if (w < 1 || h < 1 || w > 16384 || h > 16384) return fail(VPT_ERR_INVALID, "environment size %dx%d out of range", w, h);
    std::vector<float4> t((size_t)w * h);
    for (size_t i = 0; i < t.size(); i++)
        t[i] = make_float4((float)rgba[4 * i] / 255.0f, (float)rgba[4 * i + 1] / 255.0f,
                           (float)rgba[4 * i + 2] / 255.0f, (float)rgba[4 * i + 3] / 255.0f);
    VPT_TRY(upload_table(r->ctx, &r->env, t));
    r->env_w = w; r->env_h = h; r->env_const = t[0];
    r->env_opaque = true;
    for (size_t i = 0; i < t.size(); i++) if (rgba[4 * i + 3] != 255) { r->env_opaque = false; break; }
    r->cls.poisoned = true;                                  // (MCS: the fixed points of the ray-missing pixels move with the environment)
    return VPT_OK;
}
extern "C" int vpt_renderer_create(vpt_context *c, int kind, int width, int height, vpt_renderer **out) {
    if (!c || !out) return fail(VPT_ERR_INVALID, "null argument");
    if (kind < VPT_RENDERER_MIP || kind > VPT_RENDERER_DOS) return fail(VPT_ERR_INVALID, "No suitable class");  // RendererFactory.js:21
    if (width < 1 || height < 1 || width > 32768 || height > 32768) return fail(VPT_ERR_INVALID, "resolution %dx%d out of range", width, height);
    HIP_TRY(hipSetDevice(c->device));
    vpt_renderer *r = new vpt_renderer();
    r->ctx = c; r->kind = kind; r->W = width; r->H = height;
    c->renderers.push_back(r);
    r->G = 1; r->g = 0; r->R = 8;
    r->vol = nullptr; r->tf = nullptr; r->env = nullptr;
    r->frame = r->acc = nullptr; r->render = nullptr; r->scratch = nullptr; r->scratch_bytes = 0;
    for (int i = 0; i < 4; i++) r->st[i] = nullptr;
    r->samples = nullptr; r->samples_host = 0; r->profiling = false; r->events_used = 0; r->profile_every = 1; r->profile_seq = 0;
    r->side_events_used = 0; r->timed_now = false;
    r->ndc_x = r->ndc_y = nullptr;
    r->frame_table = nullptr; r->frame_staging = nullptr; r->frame_counter = nullptr; r->frames_played = 0;
    r->warmed = false; r->play_graph = nullptr;
    r->fast_math = 0; r->boundary_atlas = 1;
    r->frame_ring = nullptr; r->ring_frames = 0; r->split = 1; r->target_is_callers = false; r->no_split = false; r->split_callers = false; r->last_ranges = 1; r->stop_events = nullptr; r->stop_used = false; r->ev_fork = nullptr; for (int i = 0; i < VPT_MAX_SPLIT - 1; i++) { r->side[i] = nullptr; r->ev_join[i] = nullptr; } r->side_busy = false; r->main_dirty = true; r->mcm_persistent = 0; r->work_counter = nullptr; r->mcs_persistent = false;   // measured slower than k_mcs at every extinction tried
    r->render_target = nullptr;
    memset(&r->cls, 0, sizeof(r->cls)); r->cls.enabled = true; r->last_layout = 0; r->hit_form = 0; r->bucket_kernel = false; r->bucket_launches = 0;
    r->tm_owner = nullptr; r->tm_valid = false; r->tm_table = nullptr; r->tm_out = nullptr; r->tm_mode = 0;
    r->lao = LaoParams{ 1, 0.69f, 1, 0.05f, 1, 0.54f, 10, 0.19f, 1.0f, { 2.0f, 12.0f, 3.0f } };
    int rc = renderer_alloc_buffers(r);
    if (rc == VPT_OK) {
        hipError_t e = hipMalloc(&r->samples, COUNTER_BYTES);
        if (e == hipSuccess) e = hipMemsetAsync(r->samples, 0, COUNTER_BYTES, c->stream);
        if (e != hipSuccess) rc = fail(VPT_ERR_HIP, "hipMalloc counter: %s", hipGetErrorString(e));
    }
    static const uint8_t default_tf[8] = { 255, 0, 0, 0, 255, 0, 0, 255 };   // AbstractRenderer.js:31-44
    static const uint8_t default_env[4] = { 255, 255, 255, 255 };            // RenderingContext.js:90-101
    if (rc == VPT_OK) rc = vpt_renderer_set_transfer_function(r, default_tf, 2, 1);
    if (rc == VPT_OK) rc = vpt_renderer_set_environment(r, default_env, 1, 1);
    if (rc != VPT_OK) { vpt_renderer_destroy(r); return rc; }
    *out = r;
    return VPT_OK;
}
static void tonemappers_unbind(vpt_context *c, vpt_renderer *r);
extern "C" int vpt_renderer_destroy(vpt_renderer *r) {
    if (!r) return VPT_OK;
    join_side(r);
    hipSetDevice(r->ctx->device);
    hipStreamSynchronize(r->ctx->stream);
    tonemappers_unbind(r->ctx, r);            // a tone mapper still bound to this renderer falls back to the white placeholder
    for (size_t i = 0; i < r->ctx->renderers.size(); i++)
        if (r->ctx->renderers[i] == r) { r->ctx->renderers.erase(r->ctx->renderers.begin() + (long)i); break; }
    renderer_free_buffers(r);                 // renderer-owned buffers only; volume is NOT owned (Volume.js:17-22)
    if (r->tf) hipFree(r->tf);
    if (r->env) hipFree(r->env);
    if (r->samples) hipFree(r->samples);
    if (r->dos_samples) hipFree(r->dos_samples);
    if (r->work_counter) hipFree(r->work_counter);
    if (r->cls.list) hipFree(r->cls.list);
    if (r->cls.violations) hipFree(r->cls.violations);
    if (r->frame_ring) hipFree(r->frame_ring);
    if (r->frame_table) hipFree(r->frame_table);
    if (r->frame_staging) hipHostFree(r->frame_staging);
    if (r->frame_counter) hipFree(r->frame_counter);
    for (int i = 0; i < VPT_MAX_SPLIT - 1; i++) if (r->side[i]) { hipStreamDestroy(r->side[i]); hipEventDestroy(r->ev_join[i]); }
    if (r->ev_fork) hipEventDestroy(r->ev_fork);
    if (r->play_graph) play_graph_free(r->play_graph);
    for (auto &ev : r->events) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); }
    for (auto &ev : r->side_events) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); }
    delete r;
    return VPT_OK;
}
extern "C" int vpt_renderer_set_shard(vpt_renderer *r, int rank, int world, int rows_per_block) {
    if (!r) return fail(VPT_ERR_INVALID, "renderer is null");
    VPT_TRY(join_side(r));
    if (world < 1 || rank < 0 || rank >= world || rows_per_block < 1) return fail(VPT_ERR_INVALID, "bad shard %d/%d rows %d", rank, world, rows_per_block);
    if (r->kind == VPT_RENDERER_DOS && world > 1)
        return fail(VPT_ERR_UNSUPPORTED, "the DOS renderer does not shard: every slice reads its neighbours' occlusion across rows");
    r->G = world; r->g = rank; r->R = rows_per_block;
    return renderer_alloc_buffers(r);
}
extern "C" int vpt_renderer_local_rows(vpt_renderer *r, int *rows) {
    if (!r || !rows) return fail(VPT_ERR_INVALID, "null argument");
    *rows = r->local_h;
    return VPT_OK;
}
extern "C" int vpt_renderer_global_row(vpt_renderer *r, int l, int *j) {
    if (!r || !j) return fail(VPT_ERR_INVALID, "null argument");
    if (l < 0 || l >= r->local_h) return fail(VPT_ERR_INVALID, "local row %d out of range", l);
    int lb = l / r->R; int g = (lb * r->G + r->g) * r->R + (l - lb * r->R);
    *j = (g < r->H) ? g : -1;
    return VPT_OK;
}
extern "C" int vpt_renderer_resize(vpt_renderer *r, int width, int height) {
    if (!r) return fail(VPT_ERR_INVALID, "renderer is null");
    VPT_TRY(join_side(r));
    if (width < 1 || height < 1 || width > 32768 || height > 32768) return fail(VPT_ERR_INVALID, "resolution %dx%d out of range", width, height);
    if (width == r->W && height == r->H) return VPT_OK;   // AbstractRenderer.js:107
    r->W = width; r->H = height;
    return renderer_alloc_buffers(r);
}
extern "C" int vpt_renderer_set_volume(vpt_renderer *r, vpt_volume *v) {
    if (!r) return fail(VPT_ERR_INVALID, "renderer is null");
    VPT_TRY(join_side(r));
    if (v && v->ctx != r->ctx) return fail(VPT_ERR_INVALID, "volume belongs to another context");
    r->vol = v;
    return VPT_OK;
}

// ---------------------------------------------------------------------------------------------
// tile classes: which 16x16 tiles can no camera ray of theirs take into the cube?  (host only, double precision)
// ---------------------------------------------------------------------------------------------
// inverse of a column-major float matrix by Gauss-Jordan with partial pivoting; out[row][col]; false: singular
static bool invert_matrix(const float *m, double out[4][4]) {
    double a[4][8];
    for (int row = 0; row < 4; row++)
        for (int col = 0; col < 4; col++) { a[row][col] = (double)m[col * 4 + row]; a[row][4 + col] = row == col ? 1.0 : 0.0; }
    for (int col = 0; col < 4; col++) {
        int piv = col;
        for (int row = col + 1; row < 4; row++) if (fabs(a[row][col]) > fabs(a[piv][col])) piv = row;
        if (!(fabs(a[piv][col]) > 1e-300)) return false;
        if (piv != col) for (int k = 0; k < 8; k++) std::swap(a[piv][k], a[col][k]);
        double inv = 1.0 / a[col][col];
        for (int k = 0; k < 8; k++) a[col][k] *= inv;
        for (int row = 0; row < 4; row++) if (row != col) { double f = a[row][col]; for (int k = 0; k < 8; k++) a[row][k] -= f * a[col][k]; }
    }
    for (int row = 0; row < 4; row++) for (int col = 0; col < 4; col++) out[row][col] = a[row][4 + col];
    return true;
}
// A camera ray of pixel (i, j) — unprojectRand with blur == 0 (mixins/unprojectRand.glsl:3-24) — joins the NDC points
// (x_i, y_j, -1) and (x_i + ax, y_j + ay, +1), |ax| <= 1/W, |ay| <= 1/H.  The unprojection is projective, so the world-space
// line is the image of that NDC line, whose point at depth z lies within (1/W, 1/H) * |z + 1| / 2 of (x_i, y_j).  The cube
// [0,1]^3 (taken as [-VPT_CLASS_EPS, 1 + VPT_CLASS_EPS]^3 against the kernels' fp32 rounding) with all eight corners in front
// of the eye plane (clip w > 0) maps onto the convex hull of its projected corners, at depths z in [zmin, zmax].  Hence: if the
// rectangle of a tile's pixel centres, widened by that drift at the cube's depths plus one pixel, does not touch the hull of the
// cube's projection, no ray of the tile — and no photon travelling along one — can be inside the cube: the tile is a MISS tile.
// Exact arithmetic is not needed, only conservatism: every doubt (a corner at or behind the eye plane, a singular matrix, wild
// depths) makes every tile a HIT tile, which is always correct.  classes[ty * tiles_x + tx] = 1 for MISS.  Rows are the LOCAL rows
// of shard (g of G, R rows per block): a tile whose 16 local rows are several runs of global rows is tested run by run.
#ifndef VPT_CLASS_EPS
#define VPT_CLASS_EPS 0.001
#endif
static void classify_tiles(int W, int H, int local_h, int G, int g, int R, const float *mvp_inverse, std::vector<uint8_t> &classes, int *ptx, int *pty) {
    const int tiles_x = (W + VPT_TILE - 1) / VPT_TILE, tiles_y = (local_h + VPT_TILE - 1) / VPT_TILE;
    *ptx = tiles_x; *pty = tiles_y;
    classes.assign((size_t)tiles_x * tiles_y, 0);
    double M[4][4];
    for (int k = 0; k < 16; k++) if (!(fabsf(mvp_inverse[k]) < 1e30f)) return;       // NaN / inf / absurd entries
    if (!invert_matrix(mvp_inverse, M)) return;
    double px[8], py[8], zmin = 1e300, zmax = -1e300;
    for (int c = 0; c < 8; c++) {
        const double e = VPT_CLASS_EPS;
        double p[4] = { (c & 1) ? 1.0 + e : -e, (c & 2) ? 1.0 + e : -e, (c & 4) ? 1.0 + e : -e, 1.0 }, q[4];
        for (int row = 0; row < 4; row++) q[row] = M[row][0] * p[0] + M[row][1] * p[1] + M[row][2] * p[2] + M[row][3] * p[3];
        if (!(q[3] > 1e-3)) return;                              // at or behind the eye plane
        px[c] = q[0] / q[3]; py[c] = q[1] / q[3];
        const double z = q[2] / q[3];
        if (!(fabs(px[c]) < 1e6) || !(fabs(py[c]) < 1e6) || !(fabs(z) < 1e3)) return;
        zmin = std::min(zmin, z); zmax = std::max(zmax, z);
    }
    // convex hull (monotone chain, counter-clockwise)
    int order[8]; for (int c = 0; c < 8; c++) order[c] = c;
    std::sort(order, order + 8, [&](int a, int b) { return px[a] < px[b] || (px[a] == px[b] && py[a] < py[b]); });
    auto cross = [&](double ax, double ay, double bx, double by, double cx, double cy) { return (bx - ax) * (cy - ay) - (by - ay) * (cx - ax); };
    double hx[18], hy[18]; int nh = 0;
    for (int pass = 0; pass < 2; pass++) {
        const int start = nh;
        for (int k = 0; k < 8; k++) {
            const int c = pass == 0 ? order[k] : order[7 - k];
            while (nh - start >= 2 && cross(hx[nh - 2], hy[nh - 2], hx[nh - 1], hy[nh - 1], px[c], py[c]) <= 0.0) nh--;
            hx[nh] = px[c]; hy[nh] = py[c]; nh++;
        }
        nh--;                                                    // the last point of a chain opens the next one
    }
    if (nh < 3) return;                                          // degenerate projection
    double bx0 = 1e300, bx1 = -1e300, by0 = 1e300, by1 = -1e300;
    for (int k = 0; k < nh; k++) { bx0 = std::min(bx0, hx[k]); bx1 = std::max(bx1, hx[k]); by0 = std::min(by0, hy[k]); by1 = std::max(by1, hy[k]); }
    const double drift = 1.01 * std::max(fabs(zmin + 1.0), fabs(zmax + 1.0)) * 0.5;
    const double mx = drift / W + 2.0 / W, my = drift / H + 2.0 / H;      // the jitter's drift at the cube's depths + one pixel
    auto rect_misses = [&](double x0, double x1, double y0, double y1) {
        if (x1 < bx0 || x0 > bx1 || y1 < by0 || y0 > by1) return true;
        for (int k = 0; k < nh; k++) {                           // a hull edge with the whole rectangle strictly on its outer side
            const int k1 = (k + 1) % nh;
            const double ex = hx[k1] - hx[k], ey = hy[k1] - hy[k];
            const double tol = -1e-12 * (fabs(ex) + fabs(ey) + 1.0);
            if (cross(hx[k], hy[k], hx[k1], hy[k1], x0, y0) < tol && cross(hx[k], hy[k], hx[k1], hy[k1], x1, y0) < tol &&
                cross(hx[k], hy[k], hx[k1], hy[k1], x0, y1) < tol && cross(hx[k], hy[k], hx[k1], hy[k1], x1, y1) < tol) return true;
        }
        return false;
    };
    for (int ty = 0; ty < tiles_y; ty++) {
        // runs of consecutive global rows among the tile row's local rows (rows past the image are padding: no pixels)
        int run0[16], run1[16], nruns = 0;
        for (int l = ty * VPT_TILE; l < std::min((ty + 1) * VPT_TILE, local_h); l++) {
            int j = l;
            if (G > 1) { int lb = l / R; j = (lb * G + g) * R + (l - lb * R); }
            if (j >= H) continue;
            if (nruns && run1[nruns - 1] + 1 == j) run1[nruns - 1] = j;
            else { run0[nruns] = run1[nruns] = j; nruns++; }
        }
        for (int tx = 0; tx < tiles_x; tx++) {
            const int i0 = tx * VPT_TILE, i1 = std::min(i0 + VPT_TILE - 1, W - 1);
            const double x0 = (2.0 * i0 + 1.0) / W - 1.0 - mx, x1 = (2.0 * i1 + 1.0) / W - 1.0 + mx;
            bool miss = true;
            for (int k = 0; k < nruns && miss; k++)
                miss = rect_misses(x0, x1, (2.0 * run0[k] + 1.0) / H - 1.0 - my, (2.0 * run1[k] + 1.0) / H - 1.0 + my);
            classes[(size_t)ty * tiles_x + tx] = miss ? 1 : 0;
        }
    }
}
// (extension, host only: no GPU is touched) the classification as tests/test_tile_classes.py checks it against brute force
extern "C" int vpt_classify_tiles(int width, int height, int rank, int world, int rows_per_block, const float *mvp_inverse,
                                  uint8_t *classes, size_t nclasses, int *tiles_x, int *tiles_y) {
    if (!mvp_inverse || !tiles_x || !tiles_y) return fail(VPT_ERR_INVALID, "null argument");
    if (width < 1 || height < 1 || width > 32768 || height > 32768 || world < 1 || rank < 0 || rank >= world || rows_per_block < 1)
        return fail(VPT_ERR_INVALID, "bad geometry");
    int local_h = height;
    if (world > 1) { int nblocks = (height + rows_per_block - 1) / rows_per_block; local_h = ((nblocks + world - 1) / world) * rows_per_block; }
    std::vector<uint8_t> cls;
    classify_tiles(width, height, local_h, world, rank, rows_per_block, mvp_inverse, cls, tiles_x, tiles_y);
    if (classes) {
        if (nclasses < cls.size()) return fail(VPT_ERR_INVALID, "classes buffer too small: %zu < %zu", nclasses, cls.size());
        memcpy(classes, cls.data(), cls.size());
    }
    return VPT_OK;
}

static int make_args(vpt_renderer *r, const vpt_uniforms *u, bool need_volume, PassArgs *a) {
    memset(a, 0, sizeof(*a));
    a->pm.W = r->W; a->pm.H = r->H; a->pm.local_h = r->local_h;
    a->pm.tiles_x = r->tiles_x; a->pm.ntiles = r->ntiles;
    a->pm.G = r->G; a->pm.g = r->g; a->pm.R = r->R;
    a->pm.rshift = -1;
    a->pm.ty0 = 0;
    for (int sft = 0; sft < 16; sft++) if ((1 << sft) == r->R) a->pm.rshift = sft;
    a->pm.ndc_x = r->ndc_x; a->pm.ndc_y = r->ndc_y;
    if (need_volume) {
        if (!r->vol || !r->vol->any_upload) return fail(VPT_ERR_NO_VOLUME, "renderer has no ready volume");
        VPT_TRY(vpt_volume_finalize(r->vol));
        vpt_volume *v = r->vol;
        a->vol.bricks = v->bricks; a->vol.nx = v->nx; a->vol.ny = v->ny; a->vol.nz = v->nz;
        a->vol.fnx = (float)v->nx; a->vol.fny = (float)v->ny; a->vol.fnz = (float)v->nz;
        a->vol.hx = (float)(v->nx - 1); a->vol.hy = (float)(v->ny - 1); a->vol.hz = (float)(v->nz - 1);
        a->vol.tab32 = v->tab32; a->vol.tabc = v->tabc;
        a->vol.filter = v->filter;
        a->vol.channels = v->channels; a->vol.slot_shift = (v->f32 ? 9u : 7u) + (v->channels == 2 ? 1u : 0u);
        a->vol.elem_shift = v->f32 ? 2u : 0u;
        a->vol.atlas = r->boundary_atlas ? v->atlas : nullptr;
        a->vol.atlas_face = v->atlas_face; a->vol.atlas_shift = v->atlas_shift;
    }
    a->env.texels = r->env; a->env.w = r->env_w; a->env.h = r->env_h; a->env.constant = r->env_const;
    a->tf = r->tf; a->tf_w = r->tf_w; a->tf_h = r->tf_h; a->tf_fw = (float)r->tf_w; a->tf_hi = (float)(r->tf_w - 1);
    if (u) {
        memcpy(a->mvp_inv.m, u->mvp_inverse, sizeof(float) * 16);
        a->seed = u->rand_seed; a->offset = u->offset; a->step = u->step_size;
        a->extinction = u->extinction; a->anisotropy = u->anisotropy;
        a->inv_extinction = 1.0f / u->extinction;      // -log(u)/rate is evaluated as -log(u) * (1/rate)
        a->max_bounces = u->max_bounces; a->steps = u->steps;
        a->light = f3{ u->light_direction[0], u->light_direction[1], u->light_direction[2] };
        a->mix = u->mix; a->blur = u->blur;
        a->isovalue = u->isovalue; a->gradient_step = u->gradient_step; a->threshold = u->threshold;
    }
    if (r->kind == VPT_RENDERER_LAO) a->lao = r->lao;
    a->inv_w = (float)(1.0 / (double)r->W);     // gl.uniform2f(uInverseResolution, 1/res, 1/res): MCMRenderer.js:91,155
    a->inv_h = (float)(1.0 / (double)r->H);
    a->frame = r->frame; a->acc = r->acc;
    a->st0 = r->st[0]; a->st1 = r->st[1]; a->st2 = r->st[2]; a->st3 = r->st[3];
    if (r->kind == VPT_RENDERER_DOS) {            // colour: st[0], in place; occlusion: in = the latest of st[2|3], out = the other
        a->st1 = nullptr; a->st2 = r->st[2 + r->dos_cur]; a->st3 = r->st[3 - r->dos_cur];
    }
    a->render = r->render_target ? r->render_target : r->render;
    if (r->tm_owner && r->tm_mode && !r->render_target) {
        a->tm_table = r->tm_table;
    }
    a->samples = r->samples;
    return VPT_OK;
}
// dynamic LDS of the sampling kernels: transfer-function pairs + the three brick-offset tables
static size_t lds_bytes(const vpt_renderer *r) {
    const vpt_volume *v = r->vol;
    return (size_t)r->tf_w * 2 * sizeof(float4) + (size_t)(v->nx + v->ny + v->nz) * 4;
}
static dim3 tile_grid(const vpt_renderer *r) { return dim3((unsigned)(r->tiles_x + 7) / 8u * 8u, (unsigned)r->tiles_y); }
// Ray-marching kernels (MIP, EAM, ISO, Depth, MCS) run as one-wave workgroups when 28 of their LDS images fit a CU: with
// the default camera only ~20 % of the tiles cross the cube, about one resident round of 4-wave workgroups, which the
// dispatcher cannot rebalance (measured: 3.3e11 samples/s against 5.7e11 when every tile crosses the cube).
static bool wave_blocks(const vpt_renderer *r) {
    return r->kind != VPT_RENDERER_MCM && r->kind != VPT_RENDERER_DOS && lds_bytes(r) * 28 <= 150 * 1024;
}
// ---- streams that really run side by side ----------------------------------------------------------------------------------
// HIP gives a stream one of a few hardware queues (four by default) and does not say which: two streams on one queue execute their
// kernels one after the other.  Measured: the three tile-row ranges of an EAM frame 51.9 us on three queues, 83 us when the process had
// created one or two other streams first; the gather pipeline's hand-off 1.5 or 6 us.  So a stream that has to
// overlap others is PICKED: candidates are created one by one and each is tried against the streams it must overlap — a 100 us spin kernel
// on either side; side by side they take the time of one, on one queue the time of two — until one passes (at most 8; the rejected ones
// are destroyed afterwards, the first candidate stands if none passes, e.g. under a profiler that serialises dispatches).
// VPT_STREAM_PROBE=0 in the environment: take the first candidate, as rounds 1-3 did.
__global__ void k_spin(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    unsigned int it = 0;
    while (wall_clock64() - t0 < ticks && ++it < 4000000u) {}       // every wave leaves: by the clock, or by the count
}
static double spin_ms(hipStream_t a, hipStream_t b, unsigned long long ticks) {
    hipStreamSynchronize(a);
    if (b) hipStreamSynchronize(b);
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, ticks);
    if (b) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, b, ticks);
    hipStreamSynchronize(a);
    if (b) hipStreamSynchronize(b);
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
static bool streams_overlap(hipStream_t a, hipStream_t b) {
    const unsigned long long ticks = 10000;                          // 100 us of the 100 MHz wall clock
    spin_ms(a, b, 100);                                              // (code object, queues: first use)
    double one = 1e30, two = 1e30;
    for (int k = 0; k < 2; k++) { one = std::min(one, spin_ms(a, nullptr, ticks)); two = std::min(two, spin_ms(a, b, ticks)); }
    return two < 1.5 * one;
}
// a new non-blocking stream that overlaps every stream of `others` (null entries skipped)
static hipError_t create_overlapping_stream(hipStream_t *out, const hipStream_t *others, int n_others) {
    static const bool probe = []() { const char *e = getenv("VPT_STREAM_PROBE"); return !(e && e[0] == '0'); }();
    hipStream_t tried[8]; int nt = 0; hipStream_t chosen = nullptr;
    while (nt < 8 && !chosen) {
        hipStream_t c;
        hipError_t e = hipStreamCreateWithFlags(&c, hipStreamNonBlocking);
        if (e != hipSuccess) { if (nt == 0) return e; (void)hipGetLastError(); break; }
        tried[nt++] = c;
        bool ok = true;
        for (int i = 0; i < n_others && ok && probe; i++) if (others[i]) ok = streams_overlap(others[i], c);
        if (ok) chosen = c;
    }
    if (!chosen) chosen = tried[0];
    for (int i = 0; i < nt; i++) if (tried[i] != chosen) hipStreamDestroy(tried[i]);
    (void)hipGetLastError();
    *out = chosen;
    return hipSuccess;
}

// the side stream's work happens-before everything enqueued on the context's stream from here on
static int join_side(vpt_renderer *r) {
    if (!r || !r->side_busy) return VPT_OK;
    for (int i = 0; i < VPT_MAX_SPLIT - 1; i++) if (r->side[i]) {
        HIP_TRY(hipEventRecord(r->ev_join[i], r->side[i]));
        HIP_TRY(hipStreamWaitEvent(r->ctx->stream, r->ev_join[i], 0));
    }
    r->side_busy = false; r->main_dirty = true;
    return VPT_OK;
}
// one sampling launch; in the gather pipeline the range's "rendered" event rides on the dispatch itself
template <typename K>
static void launch_range(K kernel, vpt_renderer *r, dim3 grid, dim3 block, size_t lds, hipStream_t stream, const PassArgs &a, int range) {
    if (r->stop_events) {
        hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, stream, nullptr, r->stop_events[range], 0, a);
        r->stop_used = true;
    } else {
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, a);
    }
}
template <typename K>
static int launch_sampling(K kernel, vpt_renderer *r, const PassArgs &a, unsigned) {
    size_t lds = lds_bytes(r);
    if (lds > 160 * 1024) return fail(VPT_ERR_UNSUPPORTED, "transfer function + volume tables need %zu B of LDS (> 160 KiB)", lds);
    if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // one-wave workgroups (the ray marchers when their LDS image is small): four times as many blocks along x, see map_pixel
    const bool wave = wave_blocks(r);
    const unsigned xmul = wave ? 4u : 1u;
    const dim3 block(wave ? 64u : (unsigned)VPT_BLOCK);
    const bool split = r->split >= 2 && !r->no_split && (!r->target_is_callers || r->split_callers);
    // (a frame rendered into caller memory — vpt_renderer_set_render_target — is consumed by work the caller enqueues on the
    // context's stream right behind it: such passes stay on that stream unless the caller has taken the join upon itself
    // (VPT_OPTION_SPLIT_CALLER_TARGETS + vpt_renderer_join).  The gather pipeline waits for every range itself.)
    if (r->cls.list_now) {
        // the HIT tiles only (marcher_track): K equal parts of the list on the K streams
        if (r->side_busy && r->last_layout != 1) VPT_TRY(join_side(r));
        r->last_layout = 1;
        const int k = split ? std::min(r->split, r->cls.n_hit) : 1;
        if (k >= 2 && r->main_dirty) {
            HIP_TRY(hipEventRecord(r->ev_fork, r->ctx->stream));
            for (int i = 0; i + 1 < r->split; i++) HIP_TRY(hipStreamWaitEvent(r->side[i], r->ev_fork, 0));
            r->main_dirty = false;
        }
        for (int i = 0; i < k; i++) {
            const int h0 = (int)((long long)r->cls.n_hit * i / k), h1 = (int)((long long)r->cls.n_hit * (i + 1) / k);
            PassArgs part = a;
            part.pm.tile_list = r->cls.list + h0; part.pm.list_n = h1 - h0;
            const unsigned blocks = wave ? (unsigned)((h1 - h0 + 7) / 8) * 32u : (unsigned)(h1 - h0);
            launch_range(kernel, r, dim3(blocks), block, lds, i == 0 ? r->ctx->stream : r->side[i - 1], part, i);
        }
        if (k >= 2) r->side_busy = true;
        r->last_ranges = k;
        return VPT_OK;
    }
    if (r->side_busy && r->last_layout != 0) VPT_TRY(join_side(r));     // the previous pass dealt tile LISTS to the streams
    r->last_layout = 0;
    if (split && r->tiles_y >= r->split) {
        dim3 g = tile_grid(r);
        const unsigned k = (unsigned)r->split;
        if (r->main_dirty) {      // whatever the context's stream did to the renderer's buffers since the last join comes first
            HIP_TRY(hipEventRecord(r->ev_fork, r->ctx->stream));
            for (unsigned i = 0; i + 1 < k; i++) HIP_TRY(hipStreamWaitEvent(r->side[i], r->ev_fork, 0));
            r->main_dirty = false;
        }
        for (unsigned i = 0; i < k; i++) {            // tile rows [g.y * i / k, g.y * (i + 1) / k)
            const unsigned y0 = g.y * i / k, y1 = g.y * (i + 1u) / k;
            PassArgs part = a;
            part.pm.ty0 = (int)y0;
            launch_range(kernel, r, dim3(g.x * xmul, y1 - y0), block, lds, i == 0 ? r->ctx->stream : r->side[i - 1], part, (int)i);
        }
        r->side_busy = true; r->last_ranges = (int)k;
    } else {
        dim3 g = tile_grid(r);
        launch_range(kernel, r, dim3(g.x * xmul, g.y), block, lds, r->ctx->stream, a, 0);
    }
    return VPT_OK;
}
// the instantiation for (addressing, filter, channels): V = VPT_V_WIDE | VPT_V_NEAREST | VPT_V_RG bits
static int variant_of(const vpt_renderer *r) {
    return (r->vol->wide ? VPT_V_WIDE : 0) | (r->vol->filter == VPT_FILTER_NEAREST ? VPT_V_NEAREST : 0) | (r->vol->channels == 2 ? VPT_V_RG : 0) |
           (r->vol->f32 ? VPT_V_F32 : 0);
}
#define LAUNCH_S(KT, r, a) do { \
    unsigned g_ = (unsigned)(r)->ntiles; \
    switch (variant_of(r)) { \
        case 0: VPT_TRY(launch_sampling(KT(0), (r), (a), g_)); break; \
        case 1: VPT_TRY(launch_sampling(KT(1), (r), (a), g_)); break; \
        case 2: VPT_TRY(launch_sampling(KT(2), (r), (a), g_)); break; \
        case 3: VPT_TRY(launch_sampling(KT(3), (r), (a), g_)); break; \
        case 8: VPT_TRY(launch_sampling(KT(8), (r), (a), g_)); break; \
        case 9: VPT_TRY(launch_sampling(KT(9), (r), (a), g_)); break; \
        case 10: VPT_TRY(launch_sampling(KT(10), (r), (a), g_)); break; \
        case 11: VPT_TRY(launch_sampling(KT(11), (r), (a), g_)); break; \
        case 32: VPT_TRY(launch_sampling(KT(32), (r), (a), g_)); break; \
        case 33: VPT_TRY(launch_sampling(KT(33), (r), (a), g_)); break; \
        case 34: VPT_TRY(launch_sampling(KT(34), (r), (a), g_)); break; \
        case 35: VPT_TRY(launch_sampling(KT(35), (r), (a), g_)); break; \
        case 40: VPT_TRY(launch_sampling(KT(40), (r), (a), g_)); break; \
        case 41: VPT_TRY(launch_sampling(KT(41), (r), (a), g_)); break; \
        case 42: VPT_TRY(launch_sampling(KT(42), (r), (a), g_)); break; \
        default: VPT_TRY(launch_sampling(KT(43), (r), (a), g_)); break; \
    } } while (0)
// (dword-aligned 12-byte taps + v_alignbyte for MIP / EAM — re-measured in round 3 on the HIT tiles only, 256^3 1080p: EAM 63.7 us
// aligned against 72.3 unaligned on one stream, 52.1 / 63.3 on three; MIP 56.3 / 72.0, 45.1 / 64.1)
#define K_MIP0(V) (k_mip<0, V | VPT_V_ALIGNED>)
#define K_MIP1(V) (k_mip<1, V | VPT_V_ALIGNED>)
#define K_EAM0(V) (k_eam<0, V | VPT_V_ALIGNED>)
#define K_EAM1(V) (k_eam<1, V | VPT_V_ALIGNED>)
#define K_MCS0(V) (k_mcs<0, V>)
#define K_MCS1(V) (k_mcs<1, V>)
#define K_ISO0(V) (k_iso<0, V>)
#define K_ISO1(V) (k_iso<1, V>)
#define K_ISOR(V) (k_iso_render<V>)
#define K_DEPTH0(V) (k_depth<0, V>)
#define K_DEPTH1(V) (k_depth<1, V>)
#define K_LAO0(V) (k_lao<0, V>)
#define K_LAO1(V) (k_lao<1, V>)
#define K_MCM0(V) (k_mcm_integrate<false, V>)
#define K_MCM1(V) (k_mcm_integrate<true, V>)
#define K_MCM0F(V) (k_mcm_integrate<false, V | VPT_V_FAST>)
#define K_MCM1F(V) (k_mcm_integrate<true, V | VPT_V_FAST>)

// ---- MCM passes over the tile classes ---------------------------------------------------------------------------------
typedef void (*PassKernel)(PassArgs);
static bool mcm_classes_usable(const vpt_renderer *r, const PassArgs &a) {
    return r->cls.enabled && r->cls.valid && a.blur == 0.0f && memcmp(r->cls.mvp, a.mvp_inv.m, sizeof(r->cls.mvp)) == 0;
}
// the kernel side of it: LINEAR one-channel volume with its boundary atlas (what k_mcm_miss samples), no persistent-wave option
static bool mcm_classes_runnable(const vpt_renderer *r, const PassArgs &a) {
    return a.vol.atlas != nullptr && (variant_of(r) & ~VPT_V_WIDE) == 0 && !r->mcm_persistent;
}
// position / transmittance of the MISS tiles, as the last pass's arithmetic would have stored them
static int mcm_materialize(vpt_renderer *r) {
    if (!r->cls.stale) return VPT_OK;
    VPT_TRY(join_side(r));
    HIP_TRY(hipSetDevice(r->ctx->device));
    PassArgs a;
    VPT_TRY(make_args(r, nullptr, false, &a));
    memcpy(a.mvp_inv.m, r->cls.mvp, sizeof(r->cls.mvp));
    a.pm.tile_list = r->cls.list + r->cls.n_hit; a.pm.list_n = r->cls.n_miss;
    if (r->cls.n_miss > 0) {
        if (r->cls.stale_fast) hipLaunchKernelGGL(k_mcm_materialize<true>, dim3((unsigned)r->cls.n_miss), dim3(VPT_BLOCK), 0, r->ctx->stream, a);
        else hipLaunchKernelGGL(k_mcm_materialize<false>, dim3((unsigned)r->cls.n_miss), dim3(VPT_BLOCK), 0, r->ctx->stream, a);
        HIP_TRY(hipGetLastError());
    }
    r->cls.stale = false;
    return VPT_OK;
}
// classifies the tiles for `mvp_inverse` (see classify_tiles) and puts the lists on the device: HIT tiles first, then MISS tiles
static int classes_build(vpt_renderer *r, const float *mvp_inverse) {
    r->cls.valid = false;
    std::vector<uint8_t> cls; int tx, ty;
    classify_tiles(r->W, r->H, r->local_h, r->G, r->g, r->R, mvp_inverse, cls, &tx, &ty);
    if (tx != r->tiles_x || ty != r->tiles_y || tx > 0xffff || ty > 0xffff) return VPT_OK;
    std::vector<uint32_t> list(cls.size());
    int nh = 0, nm = 0;
    for (int pass = 0; pass < 2; pass++)
        for (int y = 0; y < ty; y++) for (int x = 0; x < tx; x++)
            if ((int)cls[(size_t)y * tx + x] == pass) { list[(size_t)nh + nm] = (uint32_t)x | ((uint32_t)y << 16); (pass ? nm : nh)++; }
    VPT_TRY(join_side(r));                                      // passes in flight read the old lists
    if (r->cls.capacity < (int)list.size()) {
        HIP_TRY(hipStreamSynchronize(r->ctx->stream));
        if (r->cls.list) { HIP_TRY(hipFree(r->cls.list)); r->cls.list = nullptr; }
        HIP_TRY(hipMalloc(&r->cls.list, list.size() * sizeof(uint32_t)));
        r->cls.capacity = (int)list.size();
    }
    if (!r->cls.violations) {
        HIP_TRY(hipMalloc(&r->cls.violations, sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(r->cls.violations, 0, sizeof(unsigned long long), r->ctx->stream));
    }
    // (the lists travel on the context's stream, behind the passes that read the old ones)
    HIP_TRY(hipMemcpyAsync(r->cls.list, list.data(), list.size() * sizeof(uint32_t), hipMemcpyHostToDevice, r->ctx->stream));
    HIP_TRY(hipStreamSynchronize(r->ctx->stream));              // `list` is pageable host memory about to go out of scope
    r->main_dirty = true;
    r->cls.n_hit = nh; r->cls.n_miss = nm;
    memcpy(r->cls.mvp, mvp_inverse, sizeof(r->cls.mvp));
    r->cls.valid = true;
    return VPT_OK;
}
// an MCM reset with matrix u->mvp_inverse has just been enqueued
static int mcm_classify(vpt_renderer *r, const vpt_uniforms *u) {
    r->cls.valid = false; r->cls.stale = false;               // the reset rewrote every array
    if (!r->cls.enabled || u->blur != 0.0f) return VPT_OK;
    return classes_build(r, u->mvp_inverse);
}
// The accumulating ray marchers (MIP, EAM, ISO; MCS and Depth under a condition).  A pixel whose ray misses the cube contributes a
// constant frame value — MIP 0 (MIPRenderer.glsl:57-59), EAM (0,0,0,1) (EAMRenderer.glsl:58-60), ISO "no hit" (ISORenderer.glsl:58-61),
// Depth -1, MCS the environment along the ray (MCSRenderer.glsl:113-116) — and its accumulator sits at a fixed point of the
// integrate pass from the reset on (MIP: max(acc, 0) = acc; EAM: (0,0,0,1) re-quantises to itself for any mix; ISO: "no hit" never
// replaces anything) or from the first pass on, if that pass had mix == 1 (MCS: acc = env, then env + (env - env) * m = env; Depth:
// acc = -1, then -(m + fl(1 - m)) = -1 for every m in [0, 1]).  So once one whole fused pass has run since the reset, and as long as
// every pass since the reset used ONE matrix, the tiles none of whose rays meet the cube (classify_tiles — the same conservative
// MISS class as MCM's) hold final values in accumulator and render buffer, and a fused pass needs to launch the HIT tiles only.
// Called for every generate / fused pass; sets r->cls.list_now for the launch that follows.
static int marcher_track(vpt_renderer *r, const PassArgs &a, bool fused, float first_mix) {
    // (MCS: the alpha channel's first step is fl(fl(e - 1) + 1) from the reset's 1, which is e itself only for e = 1: opaque environments)
    TileClasses &c = r->cls;
    c.list_now = false;
    const int k = r->kind;
    if (k != VPT_RENDERER_MIP && k != VPT_RENDERER_EAM && k != VPT_RENDERER_ISO && k != VPT_RENDERER_MCS && k != VPT_RENDERER_DEPTH) return VPT_OK;
    if (c.passes == 0) {
        memcpy(c.mvp, a.mvp_inv.m, sizeof(c.mvp));
        c.valid = false; c.poisoned = false;
        c.first_mix_one = fused && first_mix == 1.0f;
    } else if (memcmp(c.mvp, a.mvp_inv.m, sizeof(c.mvp)) != 0) {
        c.poisoned = true;
    }
    c.passes++;
    if (fused && c.enabled && !c.poisoned && !r->no_split /* not while a graph is being captured: its grid would be frozen */) {
        const bool fixed_point = k == VPT_RENDERER_DEPTH ? c.first_mix_one : (k == VPT_RENDERER_MCS ? (c.first_mix_one && r->env_opaque) : true);
        if (c.fused_passes >= 1 && fixed_point && c.reset_seen) {
            if (!c.valid) VPT_TRY(classes_build(r, c.mvp));
            c.list_now = c.valid && c.n_hit > 0 && c.n_miss > 0;
        }
    }
    if (fused) c.fused_passes++;
    return VPT_OK;
}
// one MCM pass (integrate, or render() = integrate + renderFrame) as list launches: the HIT tiles through k_mcm_integrate on the
// context's stream, the MISS tiles through k_mcm_miss — with VPT_OPTION_SPLIT_STREAMS = K as K - 1 equal parts on the side streams,
// so that the latency-bound HIT tiles and the arithmetic-bound MISS tiles share the chip for the whole frame
template <bool FUSE>
static int launch_mcm_classes(vpt_renderer *r, const PassArgs &a) {
    const bool wide = (variant_of(r) & VPT_V_WIDE) != 0, fast = r->fast_math != 0, check = r->cls.verify;
    PassKernel kh, km;
    // the HIT tiles: few enough to be resident at once at 5 waves per SIMD (a shard's share) -> the form with the early path end,
    // whose pass is one wave per SIMD walking a chain of dependent latencies; else the 7-waves form (VPT_OPTION_HIT_KERNEL_FORM overrides)
    const bool early = r->hit_form == 2 || (r->hit_form == 0 && r->cls.n_hit <= 1280);
    if (early) {
        if (fast) kh = wide ? (PassKernel)k_mcm_integrate_early<FUSE, VPT_V_WIDE | VPT_V_FAST> : (PassKernel)k_mcm_integrate_early<FUSE, VPT_V_FAST>;
        else kh = wide ? (PassKernel)k_mcm_integrate_early<FUSE, VPT_V_WIDE> : (PassKernel)k_mcm_integrate_early<FUSE, 0>;
    } else {
        if (fast) kh = wide ? (PassKernel)k_mcm_integrate<FUSE, VPT_V_WIDE | VPT_V_FAST> : (PassKernel)k_mcm_integrate<FUSE, VPT_V_FAST>;
        else kh = wide ? (PassKernel)k_mcm_integrate<FUSE, VPT_V_WIDE> : (PassKernel)k_mcm_integrate<FUSE, 0>;
    }
    // the MISS tiles: the sample consumed after the path end (its gather flies under that arithmetic) — whole frame 80.8 -> 79.3-79.7 us
    // fast-math, 96.1 -> 92.8 bit-exact, rank 3 of 8's share 18.4 -> 17.1 bit-exact but 15.8 -> 16.9 fast-math: there the sample is
    // consumed where the shader takes it
    const bool late = !(fast && early);
    if (check) km = fast ? (late ? (PassKernel)k_mcm_miss<FUSE, VPT_V_FAST, true, true> : (PassKernel)k_mcm_miss<FUSE, VPT_V_FAST, true, false>)
                         : (PassKernel)k_mcm_miss<FUSE, 0, true, true>;
    else km = fast ? (late ? (PassKernel)k_mcm_miss<FUSE, VPT_V_FAST, false, true> : (PassKernel)k_mcm_miss<FUSE, VPT_V_FAST, false, false>)
                   : (PassKernel)k_mcm_miss<FUSE, 0, false, true>;
    const size_t lds_hit = lds_bytes(r), lds_miss = (size_t)r->tf_w * 2 * sizeof(float4);
    if (lds_hit > 160 * 1024) return fail(VPT_ERR_UNSUPPORTED, "transfer function + volume tables need %zu B of LDS (> 160 KiB)", lds_hit);
    if (lds_hit > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)kh, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_hit));
    int k = 1;
    if (r->split >= 2 && !r->no_split && (!r->target_is_callers || r->split_callers)) k = r->split;
    if (r->side_busy && r->last_layout != 1) VPT_TRY(join_side(r));      // the tile -> stream map changes: order the streams once
    r->last_layout = 1;
    struct Part { PassKernel kernel; const uint32_t *list; int n; size_t lds; };
    Part parts[VPT_MAX_SPLIT]; int np = 0;
    // (measured, 1080p headline frame, us per frame: HIT | MISS on two streams 81.0; HIT | MISS/2 | MISS/2 82.3-83.0; HIT/2 | HIT/2 | MISS
    // 82.7-84.1; four streams 93; one stream, HIT then MISS: 102.  Capping the HIT kernel's residency (dynamic LDS) to 2 / 3 / 4 / 5
    // workgroups per CU so that MISS waves always sit beside its waves: 99 / 91 / 83.4 / 82.2 against 81.6 uncapped — DESIGN.md section 5)
    const int hit_parts = r->cls.n_hit > 0 ? 1 : 0;
    const int miss_parts = std::max(1, k - hit_parts);
    for (int i = 0; i < hit_parts; i++) {
        const int h0 = (int)((long long)r->cls.n_hit * i / hit_parts), h1 = (int)((long long)r->cls.n_hit * (i + 1) / hit_parts);
        if (h1 > h0) parts[np++] = Part{ kh, r->cls.list + h0, h1 - h0, lds_hit };
    }
    for (int i = 0; i < miss_parts; i++) {
        const int m0 = (int)((long long)r->cls.n_miss * i / miss_parts), m1 = (int)((long long)r->cls.n_miss * (i + 1) / miss_parts);
        if (m1 > m0) parts[np++] = Part{ km, r->cls.list + r->cls.n_hit + m0, m1 - m0, lds_miss };
    }
    if (k == 1) {
        // one stream: the launches follow each other; the dispatch's completion event (gather pipeline) rides on the last
        for (int i = 0; i < np; i++) {
            PassArgs part = a;
            part.pm.tile_list = parts[i].list; part.pm.list_n = parts[i].n; part.miss_load_pos = r->cls.stale ? 0u : 1u; part.violations = r->cls.violations;
            if (i + 1 == np) launch_range(parts[i].kernel, r, dim3((unsigned)parts[i].n), dim3(VPT_BLOCK), parts[i].lds, r->ctx->stream, part, 0);
            else hipLaunchKernelGGL(parts[i].kernel, dim3((unsigned)parts[i].n), dim3(VPT_BLOCK), parts[i].lds, r->ctx->stream, part);
        }
        r->last_ranges = 1;
    } else {
        if (r->main_dirty) {
            HIP_TRY(hipEventRecord(r->ev_fork, r->ctx->stream));
            for (int i = 0; i + 1 < k; i++) HIP_TRY(hipStreamWaitEvent(r->side[i], r->ev_fork, 0));
            r->main_dirty = false;
        }
        // (the MISS-tile kernel is the longer of the two and goes first: its stream is the one a short sequence of frames waits for at the
        // end — blocks of 5 frames 87.5 -> 85.3 us per frame, of 20 frames 81.9 -> 81.1, long sequences the same)
        for (int i = np - 1; i >= 0; i--) {
            PassArgs part = a;
            part.pm.tile_list = parts[i].list; part.pm.list_n = parts[i].n; part.miss_load_pos = r->cls.stale ? 0u : 1u; part.violations = r->cls.violations;
            // profiling: the context's stream is bracketed by the caller (Timed); the first side launch gets a pair of its own
            hipEvent_t e1 = nullptr;
            if (i == 1 && r->timed_now) {
                if (r->side_events_used == r->side_events.size()) {
                    hipEvent_t a0, a1;
                    if (hipEventCreate(&a0) == hipSuccess && hipEventCreate(&a1) == hipSuccess) r->side_events.push_back({ a0, a1 });
                }
                if (r->side_events_used < r->side_events.size()) {
                    hipEventRecord(r->side_events[r->side_events_used].first, r->side[0]);
                    e1 = r->side_events[r->side_events_used++].second;
                }
            }
            launch_range(parts[i].kernel, r, dim3((unsigned)parts[i].n), dim3(VPT_BLOCK), parts[i].lds, i == 0 ? r->ctx->stream : r->side[i - 1], part, i);
            if (e1) hipEventRecord(e1, r->side[0]);
        }
        r->side_busy = true; r->last_ranges = std::max(np, 1);
    }
    r->cls.stale = r->cls.n_miss > 0; r->cls.stale_fast = fast;
    return VPT_OK;
}

// VPT_OPTION_BUCKET_KERNEL: frames [0, count) of a bucket (frame f -> ring + f * slot_pixels texels) by one launch per tile class —
// k_mcm_bucket_hit on the context's stream, k_mcm_bucket_miss on the first side stream.  *ready = false: the preconditions of the tile
// classes do not hold (launch_mcm_pass) and the caller plays the frames one by one.
typedef void (*BucketKernel)(PassArgs, FrameSeeds, uint32_t, void *, uint32_t);
static int mcm_before_pass(vpt_renderer *r, const PassArgs &a, bool *same_matrix);
static int mcm_bucket_ready(vpt_renderer *r, const PassArgs &a, bool *ready) {
    bool same = false;
    VPT_TRY(mcm_before_pass(r, a, &same));
    const bool two_streams = r->split >= 2 && !r->no_split && (!r->target_is_callers || r->split_callers);
    *ready = same && r->cls.enabled && mcm_classes_runnable(r, a) && two_streams && !r->cls.verify;
    return VPT_OK;
}
template <bool DISPLAY>
static void bucket_kernels(bool wide, bool fast, bool early, BucketKernel *kh, BucketKernel *km) {
    if (early) {
        if (fast) *kh = wide ? (BucketKernel)k_mcm_bucket_hit<VPT_V_WIDE | VPT_V_FAST, true, DISPLAY> : (BucketKernel)k_mcm_bucket_hit<VPT_V_FAST, true, DISPLAY>;
        else *kh = wide ? (BucketKernel)k_mcm_bucket_hit<VPT_V_WIDE, true, DISPLAY> : (BucketKernel)k_mcm_bucket_hit<0, true, DISPLAY>;
    } else {
        if (fast) *kh = wide ? (BucketKernel)k_mcm_bucket_hit<VPT_V_WIDE | VPT_V_FAST, false, DISPLAY> : (BucketKernel)k_mcm_bucket_hit<VPT_V_FAST, false, DISPLAY>;
        else *kh = wide ? (BucketKernel)k_mcm_bucket_hit<VPT_V_WIDE, false, DISPLAY> : (BucketKernel)k_mcm_bucket_hit<0, false, DISPLAY>;
    }
    const bool late = !(fast && early);
    *km = fast ? (late ? (BucketKernel)k_mcm_bucket_miss<VPT_V_FAST, true, DISPLAY> : (BucketKernel)k_mcm_bucket_miss<VPT_V_FAST, false, DISPLAY>)
               : (BucketKernel)k_mcm_bucket_miss<0, true, DISPLAY>;
}
// display_table: null = RGBA16F slots; else the armed tone mapper's table — RGBA8 slots (slot_pixels counts texels either way)
static int launch_mcm_bucket(vpt_renderer *r, const PassArgs &a, const FrameVar *v, int count, void *ring, uint32_t slot_pixels, bool last_to_render_buffer,
                             const uint8_t *display_table = nullptr) {
    if (count < 1 || count > VPT_BUCKET_FRAMES) return fail(VPT_ERR_INVALID, "a bucket launch holds 1..%d frames", VPT_BUCKET_FRAMES);
    const bool wide = (variant_of(r) & VPT_V_WIDE) != 0, fast = r->fast_math != 0;
    // HIT tiles few enough to be resident at once at the kernel's four waves per SIMD: the form with the early path end (launch_mcm_classes)
    const bool early = r->hit_form == 2 || (r->hit_form == 0 && r->cls.n_hit <= 1024);
    BucketKernel kh, km;
    if (display_table) bucket_kernels<true>(wide, fast, early, &kh, &km);
    else bucket_kernels<false>(wide, fast, early, &kh, &km);
    const size_t lds_hit = lds_bytes(r), lds_miss = (size_t)r->tf_w * 2 * sizeof(float4);
    if (lds_hit > 160 * 1024) return fail(VPT_ERR_UNSUPPORTED, "transfer function + volume tables need %zu B of LDS (> 160 KiB)", lds_hit);
    if (lds_hit > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)kh, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_hit));
    if (r->side_busy && r->last_layout != 1) VPT_TRY(join_side(r));
    r->last_layout = 1;
    if (r->main_dirty) {
        HIP_TRY(hipEventRecord(r->ev_fork, r->ctx->stream));
        for (int i = 0; i + 1 < r->split; i++) HIP_TRY(hipStreamWaitEvent(r->side[i], r->ev_fork, 0));
        r->main_dirty = false;
    }
    FrameSeeds fs;
    for (int f = 0; f < VPT_BUCKET_FRAMES; f++) fs.seed[f] = f < count ? v[f].seed : 0.0f;
    PassArgs part = a;
    part.miss_load_pos = r->cls.stale ? 0u : 1u; part.violations = r->cls.violations; part.tm_table = display_table;
    if (!last_to_render_buffer) part.render = nullptr;
    if (r->cls.n_hit > 0) {
        part.pm.tile_list = r->cls.list; part.pm.list_n = r->cls.n_hit;
        hipLaunchKernelGGL(kh, dim3((unsigned)r->cls.n_hit), dim3(VPT_BLOCK), lds_hit, r->ctx->stream, part, fs, (uint32_t)count, ring, slot_pixels);
    }
    if (r->cls.n_miss > 0) {
        part.pm.tile_list = r->cls.list + r->cls.n_hit; part.pm.list_n = r->cls.n_miss;
        hipLaunchKernelGGL(km, dim3((unsigned)r->cls.n_miss), dim3(VPT_BLOCK), lds_miss, r->side[0], part, fs, (uint32_t)count, ring, slot_pixels);
    }
    r->side_busy = true; r->last_ranges = 2;
    r->cls.stale = r->cls.n_miss > 0; r->cls.stale_fast = fast;
    r->tm_valid = false;
    r->bucket_launches++;
    return VPT_OK;
}

#ifdef VPT_WITH_PERSISTENT_KERNELS
template <typename K>
static int launch_mcs_persist(K kernel, vpt_renderer *r, const PassArgs &a) {
    size_t lds = lds_bytes(r);
    if (lds > 160 * 1024) return fail(VPT_ERR_UNSUPPORTED, "transfer function + volume tables need %zu B of LDS (> 160 KiB)", lds);
    if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const size_t counter_bytes = (size_t)VPT_WORK_SHARDS * VPT_WORK_STRIDE * sizeof(uint32_t);
    if (!r->work_counter) HIP_TRY(hipMalloc(&r->work_counter, counter_bytes));
    HIP_TRY(hipMemsetAsync(r->work_counter, 0, counter_bytes, r->ctx->stream));
    int ntx8 = (r->W + 7) / 8, nty8 = (r->local_h + 7) / 8, ntiles8 = ntx8 * nty8;
    int blocks = (ntiles8 + 3) / 4;
    if (blocks > 256 * 6) blocks = 256 * 6;           // persistent: every wave resident, tiles drawn from the counter
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(VPT_BLOCK), lds, r->ctx->stream, a, r->work_counter, ntx8, ntiles8);
    return VPT_OK;
}
// persistent MCM: as many workgroups as are resident at once (occupancy query x CUs), never more than there are segments
template <typename K>
static int launch_mcm_persist(K kernel, vpt_renderer *r, const PassArgs &a) {
    size_t lds = lds_bytes(r);
    if (lds > 160 * 1024) return fail(VPT_ERR_UNSUPPORTED, "transfer function + volume tables need %zu B of LDS (> 160 KiB)", lds);
    if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, VPT_BLOCK, lds));
    if (per_cu < 1) per_cu = 1;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, r->ctx->device));
    int nseg = r->tiles_x * ((r->local_h + VPT_TILE - 1) / VPT_TILE) * 4;
    int blocks = per_cu * prop.multiProcessorCount;
    if (blocks > (nseg + 3) / 4) blocks = (nseg + 3) / 4;
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(VPT_BLOCK), lds, r->ctx->stream, a, nseg);
    return VPT_OK;
}
#define LAUNCH_MCM_PERSIST(FUSE, r, a) do { \
    int v_ = ((r)->vol->wide ? VPT_V_WIDE : 0) | ((r)->vol->filter == VPT_FILTER_NEAREST ? VPT_V_NEAREST : 0); \
    switch (v_) { \
        case 0: VPT_TRY((r)->mcm_persistent == 2 ? launch_mcm_persist((k_mcm_persist<FUSE, 0, true>), (r), (a)) : launch_mcm_persist((k_mcm_persist<FUSE, 0, false>), (r), (a))); break; \
        case 1: VPT_TRY(launch_mcm_persist((k_mcm_persist<FUSE, 1, false>), (r), (a))); break; \
        case 2: VPT_TRY((r)->mcm_persistent == 2 ? launch_mcm_persist((k_mcm_persist<FUSE, 2, true>), (r), (a)) : launch_mcm_persist((k_mcm_persist<FUSE, 2, false>), (r), (a))); break; \
        default: VPT_TRY(launch_mcm_persist((k_mcm_persist<FUSE, 3, false>), (r), (a))); break; \
    } } while (0)
#define LAUNCH_MCS_PERSIST(MODE, r, a) do { \
    int v_ = ((r)->vol->wide ? VPT_V_WIDE : 0) | ((r)->vol->filter == VPT_FILTER_NEAREST ? VPT_V_NEAREST : 0); \
    switch (v_) { \
        case 0: VPT_TRY(launch_mcs_persist((k_mcs_persist<MODE, 0>), (r), (a))); break; \
        case 1: VPT_TRY(launch_mcs_persist((k_mcs_persist<MODE, 1>), (r), (a))); break; \
        case 2: VPT_TRY(launch_mcs_persist((k_mcs_persist<MODE, 2>), (r), (a))); break; \
        default: VPT_TRY(launch_mcs_persist((k_mcs_persist<MODE, 3>), (r), (a))); break; \
    } } while (0)

#endif

// MCM passes with a matrix (or a blur) other than the reset's: the photons of MISS tiles may now enter the cube — the classes are
// void until the next reset.  Whole-image kernels need the MISS tiles' position / transmittance arrays up to date first.
static int mcm_before_pass(vpt_renderer *r, const PassArgs &a, bool *same_matrix) {
    const bool same = r->cls.valid && a.blur == 0.0f && memcmp(r->cls.mvp, a.mvp_inv.m, sizeof(r->cls.mvp)) == 0;
    if (r->cls.valid && !same) { VPT_TRY(mcm_materialize(r)); r->cls.valid = false; }
    if (same_matrix) *same_matrix = same;
    return VPT_OK;
}
template <bool FUSE>
static int launch_mcm_pass(vpt_renderer *r, const PassArgs &a) {
    bool same = false;
    VPT_TRY(mcm_before_pass(r, a, &same));
    // The two kernels of the classes pay on two streams (1080p headline frame 81 us against 99-106 for the general kernel; rank 3 of 8's
    // share 17.7 against 19.4) and lose when they have to follow each other on ONE stream (102; the share: 30.5 against 20.7): a pass
    // that must stay on the context's stream — no VPT_OPTION_SPLIT_STREAMS, a caller-owned render target without
    // VPT_OPTION_SPLIT_CALLER_TARGETS, a sequence being captured — runs the general kernel.
    const bool two_streams = r->split >= 2 && !r->no_split && (!r->target_is_callers || r->split_callers);
    if (same && r->cls.enabled && mcm_classes_runnable(r, a) && two_streams) return launch_mcm_classes<FUSE>(r, a);
    VPT_TRY(mcm_materialize(r));
#ifdef VPT_WITH_PERSISTENT_KERNELS
    if (r->mcm_persistent && r->vol->channels == 1 && !r->vol->f32) { LAUNCH_MCM_PERSIST(FUSE, r, a); return VPT_OK; }
#endif
    if (r->fast_math) { if (FUSE) LAUNCH_S(K_MCM1F, r, a); else LAUNCH_S(K_MCM0F, r, a); }
    else { if (FUSE) LAUNCH_S(K_MCM1, r, a); else LAUNCH_S(K_MCM0, r, a); }
    return VPT_OK;
}

static int check_step(const vpt_uniforms *u) {
    // step sizes <= 0 or NaN would never advance t: the reference's spinner enforces min 1 (MIPRenderer.js:24, EAMRenderer.js:34)
    if (!(u->step_size > 0.0f)) return fail(VPT_ERR_INVALID, "step_size must be > 0");
    if (u->step_size < 1.0f / 65536.0f) return fail(VPT_ERR_INVALID, "step_size below 1/65536 (more than 65536 steps per ray)");
    return VPT_OK;
}

static int check_iso(const vpt_uniforms *u) {
    if (u->steps < 1 || u->steps > 65536) return fail(VPT_ERR_INVALID, "ISO steps %u outside [1, 65536]", u->steps);   // ISORenderer.js:20-25: min 1
    return VPT_OK;
}

struct Timed {   // HIP events around the dominant kernel (or around one graph replay of `launches` of them)
    vpt_renderer *r; bool on; size_t idx;
    Timed(vpt_renderer *r_, bool dominant, uint32_t launches = 1) : r(r_), on(r_->profiling && dominant), idx(0) {
        if (on) on = (r->profile_seq++ % (uint64_t)r->profile_every) == 0;
        if (!on) return;
        if (r->events_used == r->events.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
            r->events.push_back({ a, b });
            r->event_launches.push_back(1);
        }
        idx = r->events_used++;
        r->event_launches[idx] = launches;
        hipEventRecord(r->events[idx].first, r->ctx->stream);
        r->timed_now = true;
    }
    ~Timed() { if (on) { hipEventRecord(r->events[idx].second, r->ctx->stream); r->timed_now = false; } }
};

#define LAUNCH(kernel, r, a, lds) hipLaunchKernelGGL(kernel, tile_grid(r), dim3(VPT_BLOCK), (lds), (r)->ctx->stream, (a))

extern "C" int vpt_renderer_reset(vpt_renderer *r, const vpt_uniforms *u) {
    if (!r) return fail(VPT_ERR_INVALID, "renderer is null");
    VPT_TRY(join_side(r));
    if (r->kind == VPT_RENDERER_MCM && !u) return fail(VPT_ERR_INVALID, "MCM reset needs uniforms (uMvpInverseMatrix, uRandSeed)");
    HIP_TRY(hipSetDevice(r->ctx->device));
    PassArgs a;
    VPT_TRY(make_args(r, u, false, &a));
    r->cls.passes = 0; r->cls.fused_passes = 0; r->cls.poisoned = false; r->cls.list_now = false; r->cls.reset_seen = true;
    if (r->kind != VPT_RENDERER_MCM) r->cls.valid = false;
    switch (r->kind) {
        case VPT_RENDERER_MIP: LAUNCH(k_mip_reset, r, a, 0); break;
        case VPT_RENDERER_EAM: LAUNCH(k_eam_reset, r, a, 0); break;
        case VPT_RENDERER_MCS: LAUNCH(k_mcs_reset, r, a, 0); break;
        case VPT_RENDERER_MCM: LAUNCH(k_mcm_reset, r, a, 0); HIP_TRY(hipGetLastError()); VPT_TRY(mcm_classify(r, u)); break;
        case VPT_RENDERER_ISO: LAUNCH(k_iso_reset, r, a, 0); break;
        case VPT_RENDERER_DEPTH: LAUNCH(k_depth_reset, r, a, 0); break;
        case VPT_RENDERER_LAO: LAUNCH(k_eam_reset, r, a, 0); break;           // LAORenderer.glsl:285-287: (0, 0, 0, 1) into RGBA8
        case VPT_RENDERER_DOS: LAUNCH(k_dos_reset, r, a, 0); r->dos_rect_valid = false; break;
    }
    HIP_TRY(hipGetLastError());
    return VPT_OK;
}
extern "C" int vpt_renderer_generate(vpt_renderer *r, const vpt_uniforms *u) {
    if (!r) return fail(VPT_ERR_INVALID, "null argument");
    VPT_TRY(join_side(r));
    if (r->kind == VPT_RENDERER_DOS) return VPT_OK;                 // DOSRenderer.js has no _generateFrame (AbstractRenderer.js:122-124: empty)
    if (!u) return fail(VPT_ERR_INVALID, "null argument");
    if (r->kind == VPT_RENDERER_MCM) return VPT_OK;                 // MCMRenderer.js:118-119: empty
    HIP_TRY(hipSetDevice(r->ctx->device));
    if (r->kind != VPT_RENDERER_MCS) VPT_TRY(check_step(u));
    if (r->kind == VPT_RENDERER_ISO) VPT_TRY(check_iso(u));
    PassArgs a;
    VPT_TRY(make_args(r, u, true, &a));
    VPT_TRY(marcher_track(r, a, false, 0.0f));
    {
        Timed t(r, true);
        switch (r->kind) {
            case VPT_RENDERER_MIP: LAUNCH_S(K_MIP0, r, a); break;
            case VPT_RENDERER_EAM: LAUNCH_S(K_EAM0, r, a); break;
            case VPT_RENDERER_ISO: LAUNCH_S(K_ISO0, r, a); break;
            case VPT_RENDERER_DEPTH: LAUNCH_S(K_DEPTH0, r, a); break;
            case VPT_RENDERER_LAO: LAUNCH_S(K_LAO0, r, a); break;
            case VPT_RENDERER_MCS:
#ifdef VPT_WITH_PERSISTENT_KERNELS
                if (r->mcs_persistent && r->vol->channels == 1 && !r->vol->f32) { LAUNCH_MCS_PERSIST(0, r, a); break; }
#endif
                LAUNCH_S(K_MCS0, r, a); break;
        }
    }
    HIP_TRY(hipGetLastError());
    return VPT_OK;
}
extern "C" int vpt_renderer_integrate(vpt_renderer *r, const vpt_uniforms *u) {
    if (!r || !u) return fail(VPT_ERR_INVALID, "null argument");
    if (r->kind == VPT_RENDERER_DOS) return fail(VPT_ERR_INVALID, "the DOS integrate step is a sequence of slices: call vpt_renderer_integrate_slices");
    // the marchers' integrate reads the frame their (possibly split) generate launch wrote; MCM's integrate IS the split launch and
    // depends on its own ranges' previous passes only
    if (r->kind != VPT_RENDERER_MCM) VPT_TRY(join_side(r));
    HIP_TRY(hipSetDevice(r->ctx->device));
    PassArgs a;
    VPT_TRY(make_args(r, u, r->kind == VPT_RENDERER_MCM, &a));
    switch (r->kind) {
        case VPT_RENDERER_MIP: LAUNCH(k_mip_integrate, r, a, 0); break;
        case VPT_RENDERER_EAM: LAUNCH(k_eam_integrate, r, a, 0); break;
        case VPT_RENDERER_MCS: LAUNCH(k_mcs_integrate, r, a, 0); break;
        case VPT_RENDERER_ISO: LAUNCH(k_iso_integrate, r, a, 0); break;
        case VPT_RENDERER_DEPTH: LAUNCH(k_depth_integrate, r, a, 0); break;
        case VPT_RENDERER_LAO: LAUNCH(k_lao_integrate, r, a, 0); break;
        case VPT_RENDERER_MCM: {
            Timed t(r, true);
            VPT_TRY(launch_mcm_pass<false>(r, a));
            r->samples_host += r->valid_pixels * (uint64_t)u->steps;   // exactly W*H*steps per pass (MCMRenderer.glsl:129-133)
        } break;
    }
    HIP_TRY(hipGetLastError());
    return VPT_OK;
}
extern "C" int vpt_renderer_render_frame(vpt_renderer *r, const vpt_uniforms *u) {
    if (!r) return fail(VPT_ERR_INVALID, "renderer is null");
    VPT_TRY(join_side(r));
    r->tm_valid = false;                                       // (the hook kernels do not tone-map: the next vpt_tonemapper_render runs its own pass)
    HIP_TRY(hipSetDevice(r->ctx->device));
    if (r->kind == VPT_RENDERER_ISO && !u) return fail(VPT_ERR_INVALID, "ISO renderFrame needs uniforms (uLight, uGradientStep)");
    PassArgs a;
    VPT_TRY(make_args(r, u, r->kind == VPT_RENDERER_ISO, &a));     // the ISO render pass samples the volume
    switch (r->kind) {
        case VPT_RENDERER_ISO: LAUNCH_S(K_ISOR, r, a); break;
        case VPT_RENDERER_DEPTH: LAUNCH(k_depth_render, r, a, 0); break;
        case VPT_RENDERER_LAO: LAUNCH(k_eam_render, r, a, 0); break;           // LAORenderer.glsl:259-261
        case VPT_RENDERER_DOS: LAUNCH(k_dos_render, r, a, 0); break;
        case VPT_RENDERER_MIP: LAUNCH(k_mip_render, r, a, 0); break;
        case VPT_RENDERER_EAM: LAUNCH(k_eam_render, r, a, 0); break;
        case VPT_RENDERER_MCS: LAUNCH(k_mcs_render, r, a, 0); break;
        case VPT_RENDERER_MCM: LAUNCH(k_mcm_render, r, a, 0); break;
    }
    HIP_TRY(hipGetLastError());
    return VPT_OK;
}
// the fused render() launch of the renderer's kind (generate -> integrate -> renderFrame in one kernel)
static int launch_fused(vpt_renderer *r, const PassArgs &a) {
    // (a fused pass writes the armed tone mapper's output with every texel it writes; a pass whose arguments carry no tone map — a caller's
    // render target, the gather ring, a multi-pass or captured sequence — leaves that output behind the render buffer)
    r->tm_valid = r->tm_valid && a.tm_table != nullptr && a.multi_passes <= 1 && !r->no_split;
    VPT_TRY(marcher_track(r, a, true, a.mix));
    struct ListOff { vpt_renderer *r; ~ListOff() { r->cls.list_now = false; } } list_off{ r };
    switch (r->kind) {
        case VPT_RENDERER_MIP: LAUNCH_S(K_MIP1, r, a); break;
        case VPT_RENDERER_EAM: LAUNCH_S(K_EAM1, r, a); break;
        case VPT_RENDERER_ISO: LAUNCH_S(K_ISO1, r, a); break;
        case VPT_RENDERER_DEPTH: LAUNCH_S(K_DEPTH1, r, a); break;
        case VPT_RENDERER_LAO: LAUNCH_S(K_LAO1, r, a); break;
        case VPT_RENDERER_MCS:
#ifdef VPT_WITH_PERSISTENT_KERNELS
            if (r->mcs_persistent && r->vol->channels == 1 && !r->vol->f32) { LAUNCH_MCS_PERSIST(1, r, a); break; }   // (walks every tile)
#endif
            LAUNCH_S(K_MCS1, r, a); break;
        case VPT_RENDERER_MCM: VPT_TRY(launch_mcm_pass<true>(r, a)); break;
    }
    return VPT_OK;
}
extern "C" int vpt_renderer_render(vpt_renderer *r, const vpt_uniforms *u) {
    if (!r || !u) return fail(VPT_ERR_INVALID, "null argument");
    if (r->kind == VPT_RENDERER_DOS) return fail(VPT_ERR_UNSUPPORTED, "the DOS renderer has no single-launch render(): its slices depend on each other across pixels");
    HIP_TRY(hipSetDevice(r->ctx->device));
    if (r->kind != VPT_RENDERER_MCS && r->kind != VPT_RENDERER_MCM) VPT_TRY(check_step(u));
    if (r->kind == VPT_RENDERER_ISO) VPT_TRY(check_iso(u));
    PassArgs a;
    VPT_TRY(make_args(r, u, true, &a));
    {
        Timed t(r, true);
        VPT_TRY(launch_fused(r, a));
        if (r->kind == VPT_RENDERER_MCM) r->samples_host += r->valid_pixels * (uint64_t)u->steps;
    }
    HIP_TRY(hipGetLastError());
    r->warmed = true;
    return VPT_OK;
}

// ---------------------------------------------------------------------------------------------
// frame sequences: `count` render() passes per host call, per-frame uniforms in a device table, optional hipGraph replay
// ---------------------------------------------------------------------------------------------
struct PlayGraph {
    hipGraph_t graph; hipGraphExec_t exec;
    int count; bool with_gather; PassArgs key;
    bool ran;
};
static void play_graph_free(PlayGraph *g) {
    if (!g) return;
    if (g->exec) hipGraphExecDestroy(g->exec);
    if (g->graph) hipGraphDestroy(g->graph);
    delete g;
}
// Appends the per-frame uniforms of the next `count` frames to the device ring (through a pinned staging ring, so the
// copy is asynchronous and the host never waits) and returns the PassArgs shared by the frames.  The device frame
// counter is monotonic: a captured graph needs no per-replay patching.
#define VPT_FRAME_RING 2048
static int play_args(vpt_renderer *r, const vpt_uniforms *base, int count, PassArgs *a) {
    if (count < 1 || count > VPT_FRAME_RING / 4) return fail(VPT_ERR_INVALID, "frame count %d out of range [1, %d]", count, VPT_FRAME_RING / 4);
    if (r->kind == VPT_RENDERER_MIP || r->kind == VPT_RENDERER_EAM) VPT_TRY(check_step(base));
    return make_args(r, base, true, a);
}
// eager frames carry their uniforms in the kernel arguments
static inline PassArgs frame_args(const PassArgs &a, const FrameVar &v) {
    PassArgs f = a;
    f.seed = v.seed; f.offset = v.offset; f.mix = v.mix; f.light = f3{ v.lx, v.ly, v.lz };
    return f;
}
static int play_upload_table(vpt_renderer *r, const float *vars, int count, PassArgs *a) {
    vpt_context *c = r->ctx;
    static_assert(sizeof(FrameVar) == 8 * sizeof(float), "FrameVar is 8 floats");
    if (!r->frame_table) {
        HIP_TRY(hipMalloc(&r->frame_table, (size_t)VPT_FRAME_RING * sizeof(FrameVar)));
        HIP_TRY(hipHostMalloc((void **)&r->frame_staging, (size_t)VPT_FRAME_RING * sizeof(FrameVar), hipHostMallocDefault));
        HIP_TRY(hipMalloc(&r->frame_counter, sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(r->frame_counter, 0, sizeof(uint32_t), c->stream));
        r->frames_played = 0;
    }
    // a staging slot is reused VPT_FRAME_RING frames later: never let more than half a ring be in flight
    if ((r->frames_played % (VPT_FRAME_RING / 2)) + (uint64_t)count > VPT_FRAME_RING / 2) HIP_TRY(hipStreamSynchronize(c->stream));
    const FrameVar *src = (const FrameVar *)vars;
    int pos = (int)(r->frames_played % VPT_FRAME_RING);
    int first = count < VPT_FRAME_RING - pos ? count : VPT_FRAME_RING - pos;
    memcpy(r->frame_staging + pos, src, (size_t)first * sizeof(FrameVar));
    HIP_TRY(hipMemcpyAsync(r->frame_table + pos, r->frame_staging + pos, (size_t)first * sizeof(FrameVar), hipMemcpyHostToDevice, c->stream));
    if (first < count) {
        memcpy(r->frame_staging, src + first, (size_t)(count - first) * sizeof(FrameVar));
        HIP_TRY(hipMemcpyAsync(r->frame_table, r->frame_staging, (size_t)(count - first) * sizeof(FrameVar), hipMemcpyHostToDevice, c->stream));
    }
    a->frame_base = (uint32_t)r->frames_played;       // == the device counter when the sequence starts (both advance by `count` per sequence)
    r->frames_played += (uint64_t)count;
    a->frame_table = r->frame_table;
    a->frame_counter = r->frame_counter;
    a->frame_mask = VPT_FRAME_RING - 1;
    return VPT_OK;
}
template <typename K>
static int launch_multi(K kernel, vpt_renderer *r, const PassArgs &a, uint32_t npasses) {
    size_t lds = lds_bytes(r);
    if (lds > 160 * 1024) return fail(VPT_ERR_UNSUPPORTED, "transfer function + volume tables need %zu B of LDS (> 160 KiB)", lds);
    if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kernel, tile_grid(r), dim3(VPT_BLOCK), lds, r->ctx->stream, a, npasses);
    return VPT_OK;
}
template <typename K>
static int launch_frames(K kernel, vpt_renderer *r, const PassArgs &a, uint32_t npasses, uint2 *ring) {
    size_t lds = lds_bytes(r);
    if (lds > 160 * 1024) return fail(VPT_ERR_UNSUPPORTED, "transfer function + volume tables need %zu B of LDS (> 160 KiB)", lds);
    if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kernel, tile_grid(r), dim3(VPT_BLOCK), lds, r->ctx->stream, a, npasses, ring, (uint32_t)((size_t)r->W * r->local_h));
    return VPT_OK;
}
static int launch_mcm_multi(vpt_renderer *r, const PassArgs &a, uint32_t npasses, uint2 *ring = nullptr) {
    r->tm_valid = false;
    VPT_TRY(mcm_before_pass(r, a, nullptr));
    VPT_TRY(mcm_materialize(r));                      // a whole-image kernel: every tile's full photon state
    if (r->side_busy) VPT_TRY(join_side(r));
#define MULTI_CASES(F) switch (variant_of(r)) { \
        case 0: return ring ? launch_frames(k_mcm_frames<0 | F>, r, a, npasses, ring) : launch_multi(k_mcm_multi<0 | F>, r, a, npasses); \
        case 1: return ring ? launch_frames(k_mcm_frames<1 | F>, r, a, npasses, ring) : launch_multi(k_mcm_multi<1 | F>, r, a, npasses); \
        case 2: return ring ? launch_frames(k_mcm_frames<2 | F>, r, a, npasses, ring) : launch_multi(k_mcm_multi<2 | F>, r, a, npasses); \
        case 3: return ring ? launch_frames(k_mcm_frames<3 | F>, r, a, npasses, ring) : launch_multi(k_mcm_multi<3 | F>, r, a, npasses); \
        case 8: return ring ? launch_frames(k_mcm_frames<8 | F>, r, a, npasses, ring) : launch_multi(k_mcm_multi<8 | F>, r, a, npasses); \
        case 9: return ring ? launch_frames(k_mcm_frames<9 | F>, r, a, npasses, ring) : launch_multi(k_mcm_multi<9 | F>, r, a, npasses); \
        case 10: return ring ? launch_frames(k_mcm_frames<10 | F>, r, a, npasses, ring) : launch_multi(k_mcm_multi<10 | F>, r, a, npasses); \
        case 11: return ring ? launch_frames(k_mcm_frames<11 | F>, r, a, npasses, ring) : launch_multi(k_mcm_multi<11 | F>, r, a, npasses); \
        case 32: return ring ? launch_frames(k_mcm_frames<32 | F>, r, a, npasses, ring) : launch_multi(k_mcm_multi<32 | F>, r, a, npasses); \
        case 33: return ring ? launch_frames(k_mcm_frames<33 | F>, r, a, npasses, ring) : launch_multi(k_mcm_multi<33 | F>, r, a, npasses); \
        case 34: return ring ? launch_frames(k_mcm_frames<34 | F>, r, a, npasses, ring) : launch_multi(k_mcm_multi<34 | F>, r, a, npasses); \
        case 35: return ring ? launch_frames(k_mcm_frames<35 | F>, r, a, npasses, ring) : launch_multi(k_mcm_multi<35 | F>, r, a, npasses); \
        case 40: return ring ? launch_frames(k_mcm_frames<40 | F>, r, a, npasses, ring) : launch_multi(k_mcm_multi<40 | F>, r, a, npasses); \
        case 41: return ring ? launch_frames(k_mcm_frames<41 | F>, r, a, npasses, ring) : launch_multi(k_mcm_multi<41 | F>, r, a, npasses); \
        case 42: return ring ? launch_frames(k_mcm_frames<42 | F>, r, a, npasses, ring) : launch_multi(k_mcm_multi<42 | F>, r, a, npasses); \
        default: return ring ? launch_frames(k_mcm_frames<43 | F>, r, a, npasses, ring) : launch_multi(k_mcm_multi<43 | F>, r, a, npasses); }
    if (r->fast_math) MULTI_CASES(VPT_V_FAST)
    MULTI_CASES(0)
#undef MULTI_CASES
}
static bool play_key_equal(const PassArgs &x, const PassArgs &y) { return memcmp(&x, &y, sizeof(PassArgs)) == 0; }

extern "C" int vpt_renderer_play(vpt_renderer *r, const vpt_uniforms *base, const float *frame_vars, int count, int use_graph) {
    if (!r || !base || !frame_vars) return fail(VPT_ERR_INVALID, "null argument");
    VPT_TRY(join_side(r));
    if (r->kind == VPT_RENDERER_DOS) return fail(VPT_ERR_UNSUPPORTED, "frame sequences are not defined for the DOS renderer: drive it slice by slice");
    vpt_context *c = r->ctx;
    HIP_TRY(hipSetDevice(c->device));
    PassArgs a;
    VPT_TRY(play_args(r, base, count, &a));
    if (use_graph == VPT_PLAY_GRAPH && r->warmed) {
        VPT_TRY(play_upload_table(r, frame_vars, count, &a));
        // a captured sequence runs whole-image kernels (a graph freezes its grids; tile lists change with every reset)
        if (r->kind == VPT_RENDERER_MCM) { VPT_TRY(mcm_before_pass(r, a, nullptr)); VPT_TRY(mcm_materialize(r)); }
        const float first_mix = ((const FrameVar *)frame_vars)[0].mix;
        a.frame_base = 0;                                   // (replays index the table by the device counter; the graph's key must not move)
        PlayGraph *g = r->play_graph;
        if (!g || g->with_gather || g->count != count || !play_key_equal(g->key, a)) {
            if (g) { HIP_TRY(hipStreamSynchronize(c->stream)); play_graph_free(g); r->play_graph = nullptr; }
            g = new PlayGraph(); memset(g, 0, sizeof(*g));
            g->count = count; g->with_gather = false; g->key = a;
            HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
            int rc = VPT_OK;
            r->no_split = true;                       // a captured sequence lives on the capturing stream alone
            for (int i = 0; i < count && rc == VPT_OK; i++) {
                PassArgs f = a;
                if (i == 0) f.mix = first_mix;                // (per-frame uniforms come from the table; marcher_track wants the first pass's mix)
                rc = launch_fused(r, f);
                hipLaunchKernelGGL(k_advance_frame, dim3(1), dim3(1), 0, c->stream, r->frame_counter);
            }
            r->no_split = false;
            hipError_t e = hipStreamEndCapture(c->stream, &g->graph);
            if (rc == VPT_OK && e != hipSuccess) rc = fail(VPT_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
            if (rc == VPT_OK) { e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0); if (e != hipSuccess) rc = fail(VPT_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e)); }
            if (rc != VPT_OK) { play_graph_free(g); return rc; }
            r->play_graph = g;
        }
        else {
            // a cached graph is replayed without passing through launch_fused: the marchers' pass tracking is told by hand
            for (int i = 0; i < count; i++) VPT_TRY(marcher_track(r, a, true, ((const FrameVar *)frame_vars)[i].mix));
            r->cls.list_now = false;
        }
        {
            Timed t(r, true, (uint32_t)count);       // a replay is timed as a whole: events inside a graph cannot be read back
            HIP_TRY(hipGraphLaunch(g->exec, c->stream));
        }
        g->ran = true; r->tm_valid = false;
    } else if (use_graph == VPT_PLAY_FRAMES && r->kind != VPT_RENDERER_MCM) {
        return fail(VPT_ERR_UNSUPPORTED, "VPT_PLAY_FRAMES is implemented for the MCM renderer");
    } else if (use_graph == VPT_PLAY_FUSED && r->kind != VPT_RENDERER_MCM) {
        // the accumulating renderers: the pass loop lives in their fused kernels (PassArgs.multi_passes)
        if (r->kind == VPT_RENDERER_LAO) return fail(VPT_ERR_UNSUPPORTED, "fused passes are pointless for the LAO renderer: its frames do not accumulate");
        VPT_TRY(play_upload_table(r, frame_vars, count, &a));
        a.multi_passes = (uint32_t)count;
        a.mix = ((const FrameVar *)frame_vars)[0].mix;      // (the kernels take every pass's uniforms from the table; marcher_track wants the first pass's)
        {
            Timed t(r, true, (uint32_t)count);
            VPT_TRY(launch_fused(r, a));
        }
        hipLaunchKernelGGL(k_advance_frames, dim3(1), dim3(1), 0, c->stream, r->frame_counter, (uint32_t)count);
        HIP_TRY(hipGetLastError());
        r->warmed = true;
    } else if (use_graph == VPT_PLAY_FUSED || use_graph == VPT_PLAY_FRAMES) {
        uint2 *ring = nullptr;
        if (use_graph == VPT_PLAY_FRAMES) {
            if (count > VPT_FRAME_SLOTS) return fail(VPT_ERR_INVALID, "VPT_PLAY_FRAMES: %d frames, the ring holds %d", count, VPT_FRAME_SLOTS);
            if (!r->frame_ring) {
                const size_t bytes = (size_t)VPT_FRAME_SLOTS * r->W * r->local_h * 8;
                HIP_TRY(hipMalloc(&r->frame_ring, bytes));
                HIP_TRY(hipMemsetAsync(r->frame_ring, 0, bytes, c->stream));   // a shard's padding rows are never written: zero, as in the render buffer
            }
            ring = r->frame_ring; r->ring_frames = count;
        }
        VPT_TRY(play_upload_table(r, frame_vars, count, &a));
        bool by_class = false;                           // VPT_PLAY_FRAMES where the tile classes are in force: the bucket kernels, one launch per class
        if (ring && count <= VPT_BUCKET_FRAMES) VPT_TRY(mcm_bucket_ready(r, a, &by_class));
        {
            Timed t(r, true, (uint32_t)count);
            if (by_class) VPT_TRY(launch_mcm_bucket(r, a, (const FrameVar *)frame_vars, count, ring, (uint32_t)((size_t)r->W * r->local_h), true));
            else VPT_TRY(launch_mcm_multi(r, a, (uint32_t)count, ring));
        }
        hipLaunchKernelGGL(k_advance_frames, dim3(1), dim3(1), 0, c->stream, r->frame_counter, (uint32_t)count);   // keeps the graph path's counter in step
        HIP_TRY(hipGetLastError());
        r->warmed = true;
    } else {
        const FrameVar *v = (const FrameVar *)frame_vars;
        for (int i = 0; i < count; i++) {
            Timed t(r, true);
            VPT_TRY(launch_fused(r, frame_args(a, v[i])));
        }
        HIP_TRY(hipGetLastError());
        r->warmed = true;
    }
    if (r->kind == VPT_RENDERER_MCM) r->samples_host += r->valid_pixels * (uint64_t)base->steps * (uint64_t)count;
    return VPT_OK;
}

// `count` eager render() passes, frame i into caller memory at first_target + i * stride_bytes (the slots of a bucket a collective
// will move): what `count` x { vpt_renderer_set_render_target; vpt_renderer_render } do, by one call — the host loop around those
// two cost more per frame than a 1/8 share's kernels take (torch.distributed pipeline).  The last target stays
// the renderer's render target.
extern "C" int vpt_renderer_play_into(vpt_renderer *r, const vpt_uniforms *base, const float *frame_vars, int count, void *first_target, size_t stride_bytes) {
    if (!r || !base || !frame_vars || !first_target) return fail(VPT_ERR_INVALID, "null argument");
    if (r->kind == VPT_RENDERER_DOS) return fail(VPT_ERR_UNSUPPORTED, "frame sequences are not defined for the DOS renderer: drive it slice by slice");
    const size_t need = (size_t)r->W * r->local_h * 8;
    if (stride_bytes < need) return fail(VPT_ERR_INVALID, "target stride too small: %zu < %zu", stride_bytes, need);
    if (!r->split_callers) VPT_TRY(join_side(r));
    HIP_TRY(hipSetDevice(r->ctx->device));
    PassArgs a;
    VPT_TRY(play_args(r, base, count, &a));
    const FrameVar *v = (const FrameVar *)frame_vars;
    r->target_is_callers = true;
    int i0 = 0;
    // VPT_OPTION_BUCKET_KERNEL: up to VPT_BUCKET_FRAMES frames per launch of each tile class
    while (r->kind == VPT_RENDERER_MCM && r->bucket_kernel && stride_bytes % 8 == 0 && stride_bytes / 8 <= 0xffffffffull && i0 < count) {
        const int n = std::min(count - i0, VPT_BUCKET_FRAMES);
        bool ready = false;
        VPT_TRY(mcm_bucket_ready(r, a, &ready));
        if (!ready) break;
        uint2 *ring = (uint2 *)((char *)first_target + (size_t)i0 * stride_bytes);
        Timed t(r, true, (uint32_t)n);
        VPT_TRY(launch_mcm_bucket(r, a, v + i0, n, ring, (uint32_t)(stride_bytes / 8), false));
        i0 += n;
        r->render_target = (uint2 *)((char *)first_target + (size_t)(i0 - 1) * stride_bytes);
    }
    for (int i = i0; i < count; i++) {
        r->render_target = (uint2 *)((char *)first_target + (size_t)i * stride_bytes);
        PassArgs f = frame_args(a, v[i]);
        f.render = r->render_target;
        f.tm_table = nullptr;      // (a fused tone mapper follows the renderer's own buffer only)
        Timed t(r, true);
        VPT_TRY(launch_fused(r, f));
    }
    HIP_TRY(hipGetLastError());
    r->warmed = true;
    if (r->kind == VPT_RENDERER_MCM) r->samples_host += r->valid_pixels * (uint64_t)base->steps * (uint64_t)count;
    return VPT_OK;
}

// ---------------------------------------------------------------------------------------------
// read-back, counters, profiling
// ---------------------------------------------------------------------------------------------
extern "C" int vpt_renderer_read(vpt_renderer *r, int buffer, void *dst, size_t nbytes) {
    if (!r || !dst) return fail(VPT_ERR_INVALID, "null argument");
    VPT_TRY(join_side(r));
    vpt_context *c = r->ctx;
    HIP_TRY(hipSetDevice(c->device));
    size_t npix = (size_t)r->W * r->local_h;
    if (buffer == VPT_BUFFER_RENDER) {
        if (nbytes < npix * 8) return fail(VPT_ERR_INVALID, "destination too small: %zu < %zu", nbytes, npix * 8);
        HIP_TRY(hipMemcpyAsync(dst, r->render_target ? r->render_target : r->render, npix * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return VPT_OK;
    }
    const void *src = nullptr; size_t elem = 0;
    if (r->kind == VPT_RENDERER_DOS) {            // row-major already: no tile order to undo
        if (buffer == VPT_BUFFER_ACCUM) { src = r->st[0]; elem = 16; }
        else if (buffer == VPT_BUFFER_DOS_OCCLUSION) { src = r->st[2 + r->dos_cur]; elem = 4; }
        else return fail(VPT_ERR_INVALID, "the DOS renderer holds VPT_BUFFER_ACCUM (colour) and VPT_BUFFER_DOS_OCCLUSION; its frame buffer is never written");
        if (nbytes < npix * elem) return fail(VPT_ERR_INVALID, "destination too small: %zu < %zu", nbytes, npix * elem);
        HIP_TRY(hipMemcpyAsync(dst, src, npix * elem, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return VPT_OK;
    }
    if (buffer == VPT_BUFFER_FRAME || buffer == VPT_BUFFER_ACCUM) {
        elem = frame_elem(r->kind);
        if (!elem) return fail(VPT_ERR_INVALID, "MCM has no frame/accumulation colour buffer; read the MCM state buffers");
        src = (buffer == VPT_BUFFER_FRAME) ? r->frame : r->acc;
    } else if (buffer >= VPT_BUFFER_MCM_POSITION && buffer <= VPT_BUFFER_MCM_RADIANCE) {
        if (r->kind != VPT_RENDERER_MCM) return fail(VPT_ERR_INVALID, "not an MCM renderer");
        if (buffer == VPT_BUFFER_MCM_POSITION || buffer == VPT_BUFFER_MCM_TRANSMITTANCE) VPT_TRY(mcm_materialize(r));
        elem = 16; src = r->st[buffer - VPT_BUFFER_MCM_POSITION];
    } else {
        return fail(VPT_ERR_INVALID, "unknown buffer %d", buffer);
    }
    if (nbytes < npix * elem) return fail(VPT_ERR_INVALID, "destination too small: %zu < %zu", nbytes, npix * elem);
    if (r->scratch_bytes < npix * elem) {
        if (r->scratch) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(r->scratch)); r->scratch = nullptr; }
        HIP_TRY(hipMalloc(&r->scratch, npix * elem));
        r->scratch_bytes = npix * elem;
    }
    PassArgs a;
    VPT_TRY(make_args(r, nullptr, false, &a));
    if (r->kind == VPT_RENDERER_MCM && (buffer == VPT_BUFFER_MCM_POSITION || buffer == VPT_BUFFER_MCM_TRANSMITTANCE))
        hipLaunchKernelGGL(k_detile_mcm3, tile_grid(r), dim3(VPT_BLOCK), 0, c->stream, a.pm, (const f3 *)src, (float4 *)r->scratch);
    else
    hipLaunchKernelGGL(k_detile, tile_grid(r), dim3(VPT_BLOCK), 0, c->stream, a.pm, (const uint8_t *)src, (uint8_t *)r->scratch, (int)elem);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(dst, r->scratch, npix * elem, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VPT_OK;
}
extern "C" int vpt_renderer_read_frame_slot(vpt_renderer *r, int slot, void *dst, size_t nbytes) {
    if (!r || !dst) return fail(VPT_ERR_INVALID, "null argument");
    if (!r->frame_ring || slot < 0 || slot >= r->ring_frames) return fail(VPT_ERR_INVALID, "frame slot %d: the last VPT_PLAY_FRAMES call wrote %d frames", slot, r->ring_frames);
    size_t need = (size_t)r->W * r->local_h * 8;
    if (nbytes < need) return fail(VPT_ERR_INVALID, "buffer too small: %zu < %zu", nbytes, need);
    VPT_TRY(join_side(r));
    HIP_TRY(hipSetDevice(r->ctx->device));
    HIP_TRY(hipMemcpyAsync(dst, (const char *)r->frame_ring + (size_t)slot * need, need, hipMemcpyDeviceToHost, r->ctx->stream));
    HIP_TRY(hipStreamSynchronize(r->ctx->stream));
    return VPT_OK;
}
extern "C" int vpt_renderer_frame_ring_device(vpt_renderer *r, void **ptr, size_t *slot_bytes) {
    if (!r || !ptr || !slot_bytes) return fail(VPT_ERR_INVALID, "null argument");
    *ptr = r->frame_ring; *slot_bytes = (size_t)r->W * r->local_h * 8;
    return VPT_OK;
}
extern "C" int vpt_renderer_render_buffer_device(vpt_renderer *r, void **ptr, size_t *nbytes) {
    if (!r || !ptr || !nbytes) return fail(VPT_ERR_INVALID, "null argument");
    VPT_TRY(join_side(r));
    *ptr = r->render_target ? r->render_target : r->render; *nbytes = (size_t)r->W * r->local_h * 8;
    return VPT_OK;
}
extern "C" int vpt_renderer_bucket_launches(vpt_renderer *r, uint64_t *launches) {
    if (!r || !launches) return fail(VPT_ERR_INVALID, "null argument");
    *launches = r->bucket_launches;
    return VPT_OK;
}
extern "C" int vpt_renderer_join(vpt_renderer *r) {
    if (!r) return fail(VPT_ERR_INVALID, "renderer is null");
    HIP_TRY(hipSetDevice(r->ctx->device));
    return join_side(r);
}
extern "C" int vpt_renderer_set_render_target(vpt_renderer *r, void *ptr, size_t nbytes) {
    if (!r) return fail(VPT_ERR_INVALID, "renderer is null");
    if (!r->split_callers) VPT_TRY(join_side(r));            // (launches in flight carry their target in their arguments)
    size_t need = (size_t)r->W * r->local_h * 8;
    if (ptr && nbytes < need) return fail(VPT_ERR_INVALID, "render target too small: %zu < %zu", nbytes, need);
    r->render_target = (uint2 *)ptr; r->target_is_callers = ptr != nullptr;
    return VPT_OK;
}
// uOcclusionSamples: the RG32F row of DOSRenderer.js:103-140
extern "C" int vpt_renderer_set_occlusion_samples(vpt_renderer *r, const float *xy, int count) {
    if (!r || !xy) return fail(VPT_ERR_INVALID, "null argument");
    if (r->kind != VPT_RENDERER_DOS) return fail(VPT_ERR_INVALID, "not a DOS renderer");
    if (count < 1 || count > 4096) return fail(VPT_ERR_INVALID, "occlusion sample count %d out of range (1..4096)", count);
    vpt_context *c = r->ctx;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (r->dos_samples) { HIP_TRY(hipFree(r->dos_samples)); r->dos_samples = nullptr; r->dos_nsamples = 0; }
    HIP_TRY(hipMalloc(&r->dos_samples, (size_t)count * sizeof(float2)));
    HIP_TRY(hipMemcpy(r->dos_samples, xy, (size_t)count * sizeof(float2), hipMemcpyHostToDevice));
    r->dos_nsamples = count;
    return VPT_OK;
}
// The tiles a DOS slice has to touch.  A pixel whose ray never meets the volume keeps colour 0 and occlusion 1 for the whole
// sweep, and a pixel outside the volume at two consecutive slices already has the right value in the occlusion buffer
// about to be written (it was copied there two slices ago) — so a pass only needs the screen bounding box of the
// volume: the 8 corners of [0,1]^3 taken through the inverse of uMvpInverseMatrix (double precision), padded by one tile
// against the kernel's own fp32 evaluation.  Any corner at or behind the eye plane (w <= 1e-4), or a matrix that does not
// invert, gives the whole image.
static void dos_tile_rect(const vpt_renderer *r, const float *mvp_inverse, int rect[4]) {
    const int tx = r->tiles_x, ty = r->tiles_y;
    rect[0] = 0; rect[1] = 0; rect[2] = tx; rect[3] = ty;
    double a[4][4];
    if (!invert_matrix(mvp_inverse, a)) return;
    double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
    for (int c = 0; c < 8; c++) {
        double p[4] = { (double)(c & 1), (double)((c >> 1) & 1), (double)((c >> 2) & 1), 1.0 }, q[4];
        for (int row = 0; row < 4; row++) q[row] = a[row][0] * p[0] + a[row][1] * p[1] + a[row][2] * p[2] + a[row][3] * p[3];
        if (!(q[3] > 1e-4)) return;
        double x = q[0] / q[3], y = q[1] / q[3];
        if (!(fabs(x) < 1e6) || !(fabs(y) < 1e6)) return;
        xmin = std::min(xmin, x); xmax = std::max(xmax, x); ymin = std::min(ymin, y); ymax = std::max(ymax, y);
    }
    // pixel i has its centre at NDC (2i + 1) / W - 1
    double i0 = floor(((xmin + 1.0) * r->W - 1.0) * 0.5), i1 = ceil(((xmax + 1.0) * r->W - 1.0) * 0.5);
    double j0 = floor(((ymin + 1.0) * r->H - 1.0) * 0.5), j1 = ceil(((ymax + 1.0) * r->H - 1.0) * 0.5);
    int x0 = (int)std::max(0.0, std::min((double)tx, floor(i0 / VPT_TILE) - 1.0)), x1 = (int)std::max(0.0, std::min((double)tx, floor(i1 / VPT_TILE) + 2.0));
    int y0 = (int)std::max(0.0, std::min((double)ty, floor(j0 / VPT_TILE) - 1.0)), y1 = (int)std::max(0.0, std::min((double)ty, floor(j1 / VPT_TILE) + 2.0));
    if (x1 <= x0 || y1 <= y0) { x0 = x1 = y0 = y1 = 0; }                      // the volume is off screen: nothing to launch
    rect[0] = x0; rect[1] = y0; rect[2] = x1; rect[3] = y1;
}
template <typename K>
static int launch_dos_slice(K kernel, vpt_renderer *r, PassArgs &a, const int rect[4]) {
    size_t lds = lds_bytes(r);
    if (lds > 160 * 1024) return fail(VPT_ERR_UNSUPPORTED, "transfer function + volume tables need %zu B of LDS (> 160 KiB)", lds);
    if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    a.dos.tile_x0 = rect[0]; a.dos.tile_y0 = rect[1];
    hipLaunchKernelGGL(kernel, dim3((unsigned)(rect[2] - rect[0]), (unsigned)(rect[3] - rect[1])), dim3(VPT_BLOCK), lds, r->ctx->stream, a);
    return VPT_OK;
}
static int launch_dos(vpt_renderer *r, PassArgs &a, const int rect[4]) {
    if (rect[2] <= rect[0] || rect[3] <= rect[1]) return VPT_OK;
    switch (variant_of(r)) {
        case 0: return launch_dos_slice(k_dos_slice<0>, r, a, rect);
        case 1: return launch_dos_slice(k_dos_slice<1>, r, a, rect);
        case 2: return launch_dos_slice(k_dos_slice<2>, r, a, rect);
        case 3: return launch_dos_slice(k_dos_slice<3>, r, a, rect);
        case 8: return launch_dos_slice(k_dos_slice<8>, r, a, rect);
        case 9: return launch_dos_slice(k_dos_slice<9>, r, a, rect);
        case 10: return launch_dos_slice(k_dos_slice<10>, r, a, rect);
        case 11: return launch_dos_slice(k_dos_slice<11>, r, a, rect);
        case 32: return launch_dos_slice(k_dos_slice<32>, r, a, rect);
        case 33: return launch_dos_slice(k_dos_slice<33>, r, a, rect);
        case 34: return launch_dos_slice(k_dos_slice<34>, r, a, rect);
        case 35: return launch_dos_slice(k_dos_slice<35>, r, a, rect);
        case 40: return launch_dos_slice(k_dos_slice<40>, r, a, rect);
        case 41: return launch_dos_slice(k_dos_slice<41>, r, a, rect);
        case 42: return launch_dos_slice(k_dos_slice<42>, r, a, rect);
        default: return launch_dos_slice(k_dos_slice<43>, r, a, rect);
    }
}
// _integrateFrame of the DOS renderer (DOSRenderer.js:199-259): `count` full-screen passes, pass s with
// (uOcclusionScale.x, uOcclusionScale.y, uDepth) = slices[3s .. 3s+2]; uSliceDistance = u->step_size, uExtinction = u->extinction
extern "C" int vpt_renderer_integrate_slices(vpt_renderer *r, const vpt_uniforms *u, const float *slices, int count) {
    if (!r || !u || (!slices && count > 0)) return fail(VPT_ERR_INVALID, "null argument");
    if (r->kind != VPT_RENDERER_DOS) return fail(VPT_ERR_INVALID, "not a DOS renderer");
    if (count < 0 || count > 65536) return fail(VPT_ERR_INVALID, "slice count %d out of range (0..65536)", count);
    if (!r->dos_samples) return fail(VPT_ERR_INVALID, "no occlusion samples set (vpt_renderer_set_occlusion_samples)");
    HIP_TRY(hipSetDevice(r->ctx->device));
    PassArgs a;
    VPT_TRY(make_args(r, u, true, &a));
    int rect[4], first[4];
    dos_tile_rect(r, u->mvp_inverse, rect);
    memcpy(first, rect, sizeof(rect));
    if (r->dos_rect_valid && r->dos_rect[2] > r->dos_rect[0]) {     // the matrix may have moved since the previous call: its
        if (first[2] <= first[0]) memcpy(first, r->dos_rect, sizeof(first));   // rectangle is swept once more (first slice only)
        else { first[0] = std::min(first[0], r->dos_rect[0]); first[1] = std::min(first[1], r->dos_rect[1]);
               first[2] = std::max(first[2], r->dos_rect[2]); first[3] = std::max(first[3], r->dos_rect[3]); }
    }
    Timed t(r, true, (uint32_t)(count > 0 ? count : 1));
    for (int s = 0; s < count; s++) {
        a.st2 = r->st[2 + r->dos_cur]; a.st3 = r->st[3 - r->dos_cur];
        a.dos = DosParams{ r->dos_samples, r->dos_nsamples, slices[3 * s], slices[3 * s + 1], slices[3 * s + 2], 0, 0 };
        VPT_TRY(launch_dos(r, a, s == 0 ? first : rect));
        r->dos_cur ^= 1;
    }
    if (count > 0) { memcpy(r->dos_rect, rect, sizeof(rect)); r->dos_rect_valid = true; }
    HIP_TRY(hipGetLastError());
    return VPT_OK;
}
extern "C" int vpt_renderer_set_lao_params(vpt_renderer *r, const struct vpt_lao_params *p) {
    if (!r || !p) return fail(VPT_ERR_INVALID, "null argument");
    if (r->kind != VPT_RENDERER_LAO) return fail(VPT_ERR_INVALID, "not an LAO renderer");
    if (p->num_lao_samples < 1 || p->num_lao_samples > 64 || p->num_shadow_samples < 1 || p->num_shadow_samples > 1024)
        return fail(VPT_ERR_INVALID, "sample counts out of range (LAO 1..64, shadows 1..1024)");
    if (!(p->lao_step_size >= 1.0f / 4096.0f)) return fail(VPT_ERR_INVALID, "LAO step size below 1/4096 (the occlusion march would not end)");
    static_assert(sizeof(LaoParams) == sizeof(vpt_lao_params), "parameter block layout");
    memcpy(&r->lao, p, sizeof(r->lao));
    return VPT_OK;
}
extern "C" int vpt_renderer_set_option(vpt_renderer *r, int option, int value) {
    if (!r) return fail(VPT_ERR_INVALID, "renderer is null");
    switch (option) {
#ifdef VPT_WITH_PERSISTENT_KERNELS
        // (the persistent kernels walk every tile from the context's stream: ranges of earlier split passes must be in first)
        case VPT_OPTION_MCS_PERSISTENT: VPT_TRY(join_side(r)); r->mcs_persistent = value != 0; return VPT_OK;
        case VPT_OPTION_MCM_PERSISTENT: VPT_TRY(join_side(r)); r->mcm_persistent = value < 0 ? 0 : (value > 2 ? 2 : value); return VPT_OK;
#else
        case VPT_OPTION_MCS_PERSISTENT: case VPT_OPTION_MCM_PERSISTENT:
            if (value == 0) return VPT_OK;
            return fail(VPT_ERR_UNSUPPORTED, "the persistent-wave kernels (measured slower, DESIGN.md section 5) are not part of this build: make EXTRA=-DVPT_WITH_PERSISTENT_KERNELS");
#endif
        case VPT_OPTION_BOUNDARY_ATLAS: r->boundary_atlas = value != 0; return VPT_OK;
        case VPT_OPTION_SPLIT_STREAMS:
            if (r->kind == VPT_RENDERER_DOS) return fail(VPT_ERR_UNSUPPORTED, "VPT_OPTION_SPLIT_STREAMS: the DOS renderer's slices depend on each other across pixels");
            if (value < 1 || value > VPT_MAX_SPLIT) return fail(VPT_ERR_INVALID, "VPT_OPTION_SPLIT_STREAMS: 1 .. %d", VPT_MAX_SPLIT);
            VPT_TRY(join_side(r));
            HIP_TRY(hipSetDevice(r->ctx->device));
            if (value >= 2 && !r->ev_fork) HIP_TRY(hipEventCreateWithFlags(&r->ev_fork, hipEventDisableTiming));
            for (int i = 0; i + 1 < value; i++) if (!r->side[i]) {
                hipStream_t others[VPT_MAX_SPLIT] = { r->ctx->stream };      // the context's stream and the side streams there are
                for (int k = 0; k < VPT_MAX_SPLIT - 1; k++) others[1 + k] = r->side[k];
                HIP_TRY(hipStreamSynchronize(r->ctx->stream));
                HIP_TRY(create_overlapping_stream(&r->side[i], others, VPT_MAX_SPLIT));
                HIP_TRY(hipEventCreateWithFlags(&r->ev_join[i], hipEventDisableTiming));
            }
            r->split = value; return VPT_OK;
        case VPT_OPTION_SPLIT_CALLER_TARGETS:
            VPT_TRY(join_side(r));
            r->split_callers = value != 0; return VPT_OK;
        case VPT_OPTION_FAST_MATH:
            if (r->kind != VPT_RENDERER_MCM) return fail(VPT_ERR_UNSUPPORTED, "VPT_OPTION_FAST_MATH: only the MCM renderer has a fast-arithmetic variant");
            if ((value != 0) != (r->fast_math != 0)) VPT_TRY(mcm_materialize(r));   // MISS-tile positions in the arithmetic that produced the directions
            r->fast_math = value != 0; return VPT_OK;
        case VPT_OPTION_TILE_CLASSES:
            if (r->kind == VPT_RENDERER_DOS || r->kind == VPT_RENDERER_LAO) return fail(VPT_ERR_UNSUPPORTED, "VPT_OPTION_TILE_CLASSES: not an option of the DOS / LAO renderers");
            r->cls.enabled = value != 0; return VPT_OK;
        case VPT_OPTION_HIT_KERNEL_FORM:
            if (r->kind != VPT_RENDERER_MCM) return fail(VPT_ERR_UNSUPPORTED, "VPT_OPTION_HIT_KERNEL_FORM: an MCM option");
            if (value < 0 || value > 2) return fail(VPT_ERR_INVALID, "VPT_OPTION_HIT_KERNEL_FORM: 0 (automatic), 1 or 2");
            r->hit_form = value; return VPT_OK;
        case VPT_OPTION_BUCKET_KERNEL:
            if (r->kind != VPT_RENDERER_MCM) return fail(VPT_ERR_UNSUPPORTED, "VPT_OPTION_BUCKET_KERNEL: an MCM option");
            r->bucket_kernel = value != 0; return VPT_OK;
        case VPT_OPTION_VERIFY_TILE_CLASSES:
            if (r->kind != VPT_RENDERER_MCM) return fail(VPT_ERR_UNSUPPORTED, "VPT_OPTION_VERIFY_TILE_CLASSES: an MCM option");
            r->cls.verify = value != 0; return VPT_OK;
        default: return fail(VPT_ERR_INVALID, "unknown option %d", option);
    }
}
extern "C" int vpt_renderer_sample_count(vpt_renderer *r, uint64_t *count) {
    if (!r || !count) return fail(VPT_ERR_INVALID, "null argument");
    VPT_TRY(join_side(r));                                   // the ranges of a split pass count into the same slots
    HIP_TRY(hipSetDevice(r->ctx->device));
    unsigned long long slots[VPT_COUNTER_SLOTS * VPT_COUNTER_STRIDE];
    HIP_TRY(hipMemcpyAsync(slots, r->samples, COUNTER_BYTES, hipMemcpyDeviceToHost, r->ctx->stream));
    HIP_TRY(hipStreamSynchronize(r->ctx->stream));
    uint64_t dev = 0;
    for (int i = 0; i < VPT_COUNTER_SLOTS; i++) dev += slots[i * VPT_COUNTER_STRIDE];
    *count = dev + r->samples_host;
    return VPT_OK;
}
extern "C" int vpt_renderer_clear_sample_count(vpt_renderer *r) {
    if (!r) return fail(VPT_ERR_INVALID, "renderer is null");
    VPT_TRY(join_side(r));
    HIP_TRY(hipSetDevice(r->ctx->device));
    HIP_TRY(hipMemsetAsync(r->samples, 0, COUNTER_BYTES, r->ctx->stream));
    r->samples_host = 0;
    return VPT_OK;
}
extern "C" int vpt_renderer_tile_classes(vpt_renderer *r, int *hit_tiles, int *miss_tiles, uint64_t *violations) {
    if (!r) return fail(VPT_ERR_INVALID, "renderer is null");
    if (hit_tiles) *hit_tiles = r->cls.valid ? r->cls.n_hit : r->ntiles;
    if (miss_tiles) *miss_tiles = r->cls.valid ? r->cls.n_miss : 0;
    if (violations) {
        *violations = 0;
        if (r->cls.violations) {
            VPT_TRY(join_side(r));
            HIP_TRY(hipSetDevice(r->ctx->device));
            unsigned long long v = 0;
            HIP_TRY(hipMemcpyAsync(&v, r->cls.violations, sizeof(v), hipMemcpyDeviceToHost, r->ctx->stream));
            HIP_TRY(hipStreamSynchronize(r->ctx->stream));
            *violations = v;
        }
    }
    return VPT_OK;
}
extern "C" int vpt_renderer_set_profiling(vpt_renderer *r, int enabled) {
    if (!r) return fail(VPT_ERR_INVALID, "renderer is null");
    HIP_TRY(hipSetDevice(r->ctx->device));
    HIP_TRY(hipStreamSynchronize(r->ctx->stream));
    r->profiling = enabled != 0;
    r->profile_every = enabled > 1 ? enabled : 1;   // enabled = n > 1: every n-th launch only (events cost ~7 us per launch)
    r->profile_seq = 0;
    r->events_used = 0; r->side_events_used = 0;
    return VPT_OK;
}
extern "C" int vpt_renderer_profile_side(vpt_renderer *r, double *total_ms, uint32_t *launches) {
    if (!r || !total_ms || !launches) return fail(VPT_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(r->ctx->device));
    VPT_TRY(join_side(r));
    HIP_TRY(hipStreamSynchronize(r->ctx->stream));
    double sum = 0.0;
    for (size_t i = 0; i < r->side_events_used; i++) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, r->side_events[i].first, r->side_events[i].second));
        sum += (double)ms;
    }
    *total_ms = sum; *launches = (uint32_t)r->side_events_used;
    return VPT_OK;
}
extern "C" int vpt_renderer_profile(vpt_renderer *r, double *total_ms, uint32_t *launches) {
    if (!r || !total_ms || !launches) return fail(VPT_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(r->ctx->device));
    HIP_TRY(hipStreamSynchronize(r->ctx->stream));
    double sum = 0.0;
    for (size_t i = 0; i < r->events_used; i++) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, r->events[i].first, r->events[i].second));
        sum += (double)ms;
    }
    uint32_t n = 0;
    for (size_t i = 0; i < r->events_used; i++) n += r->event_launches[i];
    *total_ms = sum; *launches = n;
    return VPT_OK;
}

// ---------------------------------------------------------------------------------------------
// probes
// ---------------------------------------------------------------------------------------------
extern "C" int vpt_probe_math(vpt_context *c, int which, const float *in, float *out, size_t n) {
    if (!c || !in || !out) return fail(VPT_ERR_INVALID, "null argument");
    if (which < 0 || which > VPT_PROBE_POW) return fail(VPT_ERR_INVALID, "unknown probe %d", which);
    if (n == 0) return VPT_OK;
    HIP_TRY(hipSetDevice(c->device));
    size_t nin = (which == VPT_PROBE_ATAN2 || which == VPT_PROBE_MIN || which == VPT_PROBE_MAX || which == VPT_PROBE_POW) ? 2 * n : n;
    float *din = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc(&din, nin * sizeof(float)));
    hipError_t e = hipMalloc(&dout, n * sizeof(float));
    if (e != hipSuccess) { hipFree(din); return fail(VPT_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e)); }
    e = hipMemcpyAsync(din, in, nin * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_probe_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, which, din, dout, n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout, n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(din); hipFree(dout);
    if (e != hipSuccess) return fail(VPT_ERR_HIP, "probe: %s", hipGetErrorString(e));
    return VPT_OK;
}
extern "C" int vpt_probe_sample(vpt_renderer *r, const float *xyz, float *rgba, size_t n) {
    if (!r || !xyz || !rgba) return fail(VPT_ERR_INVALID, "null argument");
    if (n == 0) return VPT_OK;
    vpt_context *c = r->ctx;
    HIP_TRY(hipSetDevice(c->device));
    PassArgs a;
    VPT_TRY(make_args(r, nullptr, true, &a));
    float *din = nullptr; float4 *dout = nullptr;
    HIP_TRY(hipMalloc(&din, 3 * n * sizeof(float)));
    hipError_t e = hipMalloc(&dout, n * sizeof(float4));
    if (e != hipSuccess) { hipFree(din); return fail(VPT_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e)); }
    e = hipMemcpyAsync(din, xyz, 3 * n * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        dim3 grid((unsigned)((n + 255) / 256));
        switch (variant_of(r)) {
            case 0: hipLaunchKernelGGL(k_probe_sample<0>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            case 1: hipLaunchKernelGGL(k_probe_sample<1>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            case 2: hipLaunchKernelGGL(k_probe_sample<2>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            case 3: hipLaunchKernelGGL(k_probe_sample<3>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            case 8: hipLaunchKernelGGL(k_probe_sample<8>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            case 9: hipLaunchKernelGGL(k_probe_sample<9>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            case 10: hipLaunchKernelGGL(k_probe_sample<10>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            case 11: hipLaunchKernelGGL(k_probe_sample<11>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            case 32: hipLaunchKernelGGL(k_probe_sample<32>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            case 33: hipLaunchKernelGGL(k_probe_sample<33>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            case 34: hipLaunchKernelGGL(k_probe_sample<34>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            case 35: hipLaunchKernelGGL(k_probe_sample<35>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            case 40: hipLaunchKernelGGL(k_probe_sample<40>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            case 41: hipLaunchKernelGGL(k_probe_sample<41>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            case 42: hipLaunchKernelGGL(k_probe_sample<42>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            default: hipLaunchKernelGGL(k_probe_sample<43>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
        }
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(rgba, dout, n * sizeof(float4), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(din); hipFree(dout);
    if (e != hipSuccess) return fail(VPT_ERR_HIP, "probe: %s", hipGetErrorString(e));
    return VPT_OK;
}

// ---------------------------------------------------------------------------------------------
// tone mappers
// ---------------------------------------------------------------------------------------------
struct vpt_tonemapper {
    vpt_context *ctx;
    int kind, W, H;
    vpt_renderer *source;          // bound renderer (not owned), or null
    uint2 *image; int image_w, image_rows;    // owned source texture (set_source_image), or null
    uint32_t *out; size_t out_pixels;         // RGBA8 target, grown on demand
    int rows;                      // rows of the last render
    int table_mode;                // VPT_TONEMAPPER_TABLE_*
    uint8_t *table; bool table_valid; TonemapParams table_params;   // byte table of the current parameters (vpt_tonemap.h)
    bool fuse;                     // VPT_TONEMAPPER_OPTION_FUSE (default on): arm the bound renderer's fused passes with this table and output
    TmFuse fuse_args; bool fuse_args_valid;   // what the block behind the table holds (vpt_tonemap.h)
};
static void tonemapper_disarm(vpt_tonemapper *t) {
    if (t && t->source && t->source->tm_owner == t) { t->source->tm_owner = nullptr; t->source->tm_mode = 0; t->source->tm_valid = false; }
}
static void renderers_unbind(vpt_context *c, vpt_volume *v) {
    for (vpt_renderer *r : c->renderers) if (r->vol == v) r->vol = nullptr;
}
static void tonemappers_unbind(vpt_context *c, vpt_renderer *r) {
    for (vpt_tonemapper *t : c->tonemappers) if (t->source == r) t->source = nullptr;
    r->tm_owner = nullptr; r->tm_mode = 0;
}
extern "C" int vpt_tonemapper_create(vpt_context *c, int kind, int width, int height, vpt_tonemapper **out) {
    if (!c || !out) return fail(VPT_ERR_INVALID, "null argument");
    if (kind < VPT_TONEMAPPER_ARTISTIC || kind > VPT_TONEMAPPER_UCHIMURA) return fail(VPT_ERR_INVALID, "No suitable class");   // ToneMapperFactory.js:26
    if (width < 1 || height < 1) return fail(VPT_ERR_INVALID, "bad resolution %dx%d", width, height);
    vpt_tonemapper *t = new vpt_tonemapper();
    memset(t, 0, sizeof(*t));
    t->ctx = c; t->kind = kind; t->W = width; t->H = height; t->table_mode = VPT_TONEMAPPER_TABLE_AUTO; t->fuse = true;
    c->tonemappers.push_back(t);
    *out = t;
    return VPT_OK;
}
extern "C" int vpt_tonemapper_destroy(vpt_tonemapper *t) {
    if (!t) return VPT_OK;
    hipSetDevice(t->ctx->device);
    if (t->source) join_side(t->source);
    tonemapper_disarm(t);
    hipStreamSynchronize(t->ctx->stream);
    if (t->image) hipFree(t->image);
    if (t->out) hipFree(t->out);
    if (t->table) hipFree(t->table);
    for (size_t i = 0; i < t->ctx->tonemappers.size(); i++)
        if (t->ctx->tonemappers[i] == t) { t->ctx->tonemappers.erase(t->ctx->tonemappers.begin() + (long)i); break; }
    delete t;
    return VPT_OK;
}
extern "C" int vpt_tonemapper_resize(vpt_tonemapper *t, int width, int height) {
    if (!t) return fail(VPT_ERR_INVALID, "tone mapper is null");
    if (width < 1 || height < 1) return fail(VPT_ERR_INVALID, "bad resolution %dx%d", width, height);
    tonemapper_disarm(t);
    t->W = width; t->H = height; t->rows = 0;
    return VPT_OK;
}
extern "C" int vpt_tonemapper_set_source(vpt_tonemapper *t, vpt_renderer *r) {
    if (!t) return fail(VPT_ERR_INVALID, "tone mapper is null");
    if (r && r->ctx->device != t->ctx->device) return fail(VPT_ERR_INVALID, "renderer and tone mapper live on different devices");
    HIP_TRY(hipSetDevice(t->ctx->device));
    if (t->image) { HIP_TRY(hipStreamSynchronize(t->ctx->stream)); HIP_TRY(hipFree(t->image)); t->image = nullptr; }
    if (t->source) VPT_TRY(join_side(t->source));
    tonemapper_disarm(t);
    t->source = r;
    return VPT_OK;
}
extern "C" int vpt_tonemapper_set_source_image(vpt_tonemapper *t, const void *rgba16f, int width, int rows) {
    if (!t || !rgba16f) return fail(VPT_ERR_INVALID, "null argument");
    if (width < 1 || rows < 1) return fail(VPT_ERR_INVALID, "bad image size %dx%d", width, rows);
    vpt_context *c = t->ctx;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (t->image) { HIP_TRY(hipFree(t->image)); t->image = nullptr; }
    size_t bytes = (size_t)width * rows * 8;
    HIP_TRY(hipMalloc(&t->image, bytes));
    HIP_TRY(hipMemcpyAsync(t->image, rgba16f, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    tonemapper_disarm(t);
    t->image_w = width; t->image_rows = rows; t->source = nullptr;
    return VPT_OK;
}
template <int KIND>
static void launch_tonemap(vpt_tonemapper *t, const uint2 *src, size_t n, const TonemapParams &p) {
    size_t blocks = (n + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;                 // grid-stride beyond 32 workgroups per CU
    // table form (vpt_tonemap.h): for Artistic only at saturation 1 (otherwise its channels are coupled through the mix); in AUTO mode only
    // when the image is large enough to pay for evaluating 65 536 entries, or the table of these parameters already exists
    bool current = t->table && t->table_valid && memcmp(&t->table_params, &p, sizeof(p)) == 0;
    bool use_table = (KIND != VPT_TM_ARTISTIC || p.saturation == 1.0f) &&
                     (t->table_mode == VPT_TONEMAPPER_TABLE_ALWAYS || (t->table_mode == VPT_TONEMAPPER_TABLE_AUTO && (current || n >= 4 * 65536)));
    if (use_table && !t->table && hipMalloc(&t->table, VPT_TM_TABLE_BYTES) != hipSuccess) { t->table = nullptr; use_table = false; (void)hipGetLastError(); }
    // the renderer whose fused passes may carry this map (one context: its streams are ordered against this one by events)
    vpt_renderer *fr = (t->fuse && t->source && src == t->source->render && t->source->ctx == t->ctx) ? t->source : nullptr;
    if (!use_table) {
        tonemapper_disarm(t);
        hipLaunchKernelGGL(k_tonemap<KIND>, dim3((unsigned)blocks), dim3(256), 0, t->ctx->stream, src, t->out, n, p);
        return;
    }
    // VPT_TONEMAPPER_OPTION_FUSE: the renderer's fused passes have been writing this output, with this table, along with the render buffer
    if (fr && current && fr->tm_owner == t && fr->tm_valid && fr->tm_out == t->out) return;
    struct Arm {      // after this pass the output matches the render buffer: from now on the renderer's fused passes keep it so
        vpt_tonemapper *t; vpt_renderer *r; TonemapParams p;
        ~Arm() {
            if (!r) return;
            TmFuse f = { t->out, KIND == VPT_TM_ARTISTIC ? 3 : (KIND == VPT_TM_RANGE ? 2 : 1), p.low, p.high - p.low, 1.0f - p.saturation };
            if (!t->fuse_args_valid || memcmp(&f, &t->fuse_args, sizeof(f)) != 0) {       // the block behind the table (vpt_tonemap.h)
                hipLaunchKernelGGL(k_tonemap_fuse_args, dim3(1), dim3(1), 0, t->ctx->stream, t->table, f);
                t->fuse_args = f; t->fuse_args_valid = true;
            }
            r->tm_owner = t; r->tm_table = t->table; r->tm_out = t->out; r->tm_mode = f.mode;
            r->tm_valid = true;
            r->main_dirty = true;       // side streams of split passes must see the table this stream has just (re)built
        }
    } arm{ t, fr, p };
    if (!current) {
        if (KIND == VPT_TM_ARTISTIC) hipLaunchKernelGGL(k_tonemap_table_artistic, dim3(256), dim3(256), 0, t->ctx->stream, t->table, p);
        else hipLaunchKernelGGL(k_tonemap_table<KIND>, dim3(256), dim3(256), 0, t->ctx->stream, t->table, p);
        t->table_params = p; t->table_valid = true;
    }
    const size_t lds_table = ((VPT_TM_TABLE_ENTRIES + 15) / 16) * 16;
    // table in LDS: up to 2 workgroups of 1024 per CU, grid-stride, >= 4 texels per thread (the global-memory forms below
    // remain as the fallback should the 64 KiB of dynamic LDS be refused)
    const size_t wgs = std::min<size_t>(512, (n + 4095) / 4096);
    if (KIND != VPT_TM_ARTISTIC) {
        auto k = (KIND == VPT_TM_RANGE) ? k_tonemap_apply_table_lds<true> : k_tonemap_apply_table_lds<false>;
        if (hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_table) == hipSuccess) {
            hipLaunchKernelGGL(k, dim3((unsigned)wgs), dim3(1024), lds_table, t->ctx->stream, src, t->out, n, t->table);
            return;
        }
        (void)hipGetLastError();
    } else if (hipFuncSetAttribute((const void *)k_tonemap_apply_table_artistic_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_table) == hipSuccess) {
        hipLaunchKernelGGL(k_tonemap_apply_table_artistic_lds, dim3((unsigned)wgs), dim3(1024), lds_table, t->ctx->stream, src, t->out, n, t->table, p);
        return;
    } else {
        (void)hipGetLastError();
    }
    if (KIND == VPT_TM_ARTISTIC) hipLaunchKernelGGL(k_tonemap_apply_table_artistic, dim3((unsigned)blocks), dim3(256), 0, t->ctx->stream, src, t->out, n, t->table, p);
    else if (KIND == VPT_TM_RANGE) hipLaunchKernelGGL(k_tonemap_apply_table<true>, dim3((unsigned)blocks), dim3(256), 0, t->ctx->stream, src, t->out, n, t->table);
    else hipLaunchKernelGGL(k_tonemap_apply_table<false>, dim3((unsigned)blocks), dim3(256), 0, t->ctx->stream, src, t->out, n, t->table);
}
extern "C" int vpt_tonemapper_set_option(vpt_tonemapper *t, int option, int value) {
    if (!t) return fail(VPT_ERR_INVALID, "tone mapper is null");
    if (option == VPT_TONEMAPPER_OPTION_FUSE) {
        if (t->source) VPT_TRY(join_side(t->source));
        if (!value) tonemapper_disarm(t);
        t->fuse = value != 0;
        return VPT_OK;
    }
    if (option != VPT_TONEMAPPER_OPTION_TABLE) return fail(VPT_ERR_INVALID, "unknown tone mapper option %d", option);
    if (value < VPT_TONEMAPPER_TABLE_NEVER || value > VPT_TONEMAPPER_TABLE_AUTO) return fail(VPT_ERR_INVALID, "bad value %d", value);
    t->table_mode = value;
    return VPT_OK;
}
__global__ void k_fill_u32(uint32_t *dst, size_t n, uint32_t v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;
}
extern "C" int vpt_tonemapper_render(vpt_tonemapper *t, const struct vpt_tonemap_params *params) {
    if (!t || !params) return fail(VPT_ERR_INVALID, "null argument");
    static_assert(sizeof(TonemapParams) == sizeof(vpt_tonemap_params), "parameter block layout");
    TonemapParams p; memcpy(&p, params, sizeof(p));
    // VPT_TONEMAPPER_OPTION_FUSE: the bound renderer's fused passes have been writing this output, with the table of these parameters,
    // along with its render buffer — nothing to launch, and no reason to join the streams of a split pass (vpt_tonemapper_read does)
    if (t->fuse && t->source && t->table && t->table_valid && memcmp(&t->table_params, &p, sizeof(p)) == 0) {
        vpt_renderer *r = t->source;
        if (r->tm_owner == t && r->tm_valid && r->tm_out == t->out && !r->render_target && r->W == t->W && r->H == t->H &&
            (t->kind != VPT_TONEMAPPER_ARTISTIC || p.saturation == 1.0f) && t->table_mode != VPT_TONEMAPPER_TABLE_NEVER) {
            t->rows = r->local_h;
            return VPT_OK;
        }
    }
    VPT_TRY(join_side(t->source));
    vpt_context *c = t->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const uint2 *src = nullptr; int w = t->W, rows = t->H;
    uint2 *white = nullptr;
    if (t->source) {
        vpt_renderer *r = t->source;
        src = r->render_target ? r->render_target : r->render; w = r->W; rows = r->local_h;
        if (r->ctx->stream != c->stream) HIP_TRY(hipStreamSynchronize(r->ctx->stream));      // different contexts: order by waiting
    } else if (t->image) {
        src = t->image; w = t->image_w; rows = t->image_rows;
    }
    if (w != t->W || (!t->source && rows != t->H) || (t->source && t->source->H != t->H))
        return fail(VPT_ERR_UNSUPPORTED, "source is %dx%d, tone mapper %dx%d: resampling between resolutions is not implemented "
                                         "(the reference keeps them equal, RenderingContext.js:219-228)", w, t->source ? t->source->H : rows, t->W, t->H);
    size_t n = (size_t)w * rows;
    if (t->out_pixels < n) {
        tonemapper_disarm(t);                                  // (the armed renderer holds the old output's address)
        if (t->out) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(t->out)); t->out = nullptr; t->out_pixels = 0; }
        HIP_TRY(hipMalloc(&t->out, n * 4));
        t->out_pixels = n;
    }
    if (!src) {                                               // the 1x1 white placeholder texture: a constant image
        HIP_TRY(hipMalloc(&white, n * 8));
        uint64_t one4 = 0x3c003c003c003c00ull;                // half(1) x 4
        static_assert(sizeof(uint2) == 8, "texel size");
        hipLaunchKernelGGL(k_fill_u32, dim3(1024), dim3(256), 0, c->stream, (uint32_t *)white, n * 2, (uint32_t)(one4 & 0xffffffffu));
        src = white;
    }
    switch (t->kind) {
        case VPT_TONEMAPPER_ARTISTIC:   launch_tonemap<VPT_TM_ARTISTIC>(t, src, n, p); break;
        case VPT_TONEMAPPER_RANGE:      launch_tonemap<VPT_TM_RANGE>(t, src, n, p); break;
        case VPT_TONEMAPPER_REINHARD:   launch_tonemap<VPT_TM_REINHARD>(t, src, n, p); break;
        case VPT_TONEMAPPER_REINHARD2:  launch_tonemap<VPT_TM_REINHARD2>(t, src, n, p); break;
        case VPT_TONEMAPPER_UNCHARTED2: launch_tonemap<VPT_TM_UNCHARTED2>(t, src, n, p); break;
        case VPT_TONEMAPPER_FILMIC:     launch_tonemap<VPT_TM_FILMIC>(t, src, n, p); break;
        case VPT_TONEMAPPER_UNREAL:     launch_tonemap<VPT_TM_UNREAL>(t, src, n, p); break;
        case VPT_TONEMAPPER_ACES:       launch_tonemap<VPT_TM_ACES>(t, src, n, p); break;
        case VPT_TONEMAPPER_LOTTES:     launch_tonemap<VPT_TM_LOTTES>(t, src, n, p); break;
        default:                        launch_tonemap<VPT_TM_UCHIMURA>(t, src, n, p); break;
    }
    hipError_t e = hipGetLastError();
    if (white) { hipStreamSynchronize(c->stream); hipFree(white); }
    if (e != hipSuccess) return fail(VPT_ERR_HIP, "tone-map launch: %s", hipGetErrorString(e));
    t->rows = rows;
    return VPT_OK;
}
// `count` render() passes, frame i AS THE ARMED TONE MAPPER SHOWS IT (RGBA8) into caller memory at first_target + i * stride_bytes: the
// bucket a collective moves holds half the bytes of vpt_renderer_play_into's RGBA16F frames.  MCM with the tile classes in force: the
// bucket kernels (one launch per class, the texel through the tone mapper's table in their frame store); otherwise frame by frame
// through the fused pass (which writes the tone mapper's output, VPT_TONEMAPPER_OPTION_FUSE) and a device copy of that output.
extern "C" int vpt_renderer_play_into_display(vpt_renderer *r, vpt_tonemapper *t, const vpt_uniforms *base, const float *frame_vars, int count,
                                              void *first_target, size_t stride_bytes) {
    if (!r || !t || !base || !frame_vars || !first_target) return fail(VPT_ERR_INVALID, "null argument");
    if (r->kind == VPT_RENDERER_DOS) return fail(VPT_ERR_UNSUPPORTED, "frame sequences are not defined for the DOS renderer: drive it slice by slice");
    if (t->source != r || r->tm_owner != t || !r->tm_table || !t->table_valid || !t->out)
        return fail(VPT_ERR_INVALID, "the tone mapper is not armed on this renderer: bind it (vpt_tonemapper_set_source), keep VPT_TONEMAPPER_OPTION_FUSE on and "
                                     "call vpt_tonemapper_render once with the parameters to show (table form)");
    if (r->render_target) return fail(VPT_ERR_INVALID, "a caller-owned render target is set: restore the renderer's own buffer first");
    const size_t need = (size_t)r->W * r->local_h * 4;
    if (stride_bytes < need || stride_bytes % 4) return fail(VPT_ERR_INVALID, "target stride %zu: at least %zu bytes, a multiple of 4", stride_bytes, need);
    HIP_TRY(hipSetDevice(r->ctx->device));
    PassArgs a;
    VPT_TRY(play_args(r, base, count, &a));
    const FrameVar *v = (const FrameVar *)frame_vars;
    int i0 = 0;
    while (r->kind == VPT_RENDERER_MCM && stride_bytes / 4 <= 0xffffffffull && i0 < count) {
        const int n = std::min(count - i0, VPT_BUCKET_FRAMES);
        bool ready = false;
        VPT_TRY(mcm_bucket_ready(r, a, &ready));
        if (!ready) break;
        Timed tm(r, true, (uint32_t)n);
        VPT_TRY(launch_mcm_bucket(r, a, v + i0, n, (char *)first_target + (size_t)i0 * stride_bytes, (uint32_t)(stride_bytes / 4), false, r->tm_table));
        i0 += n;
    }
    for (int i = i0; i < count; i++) {
        PassArgs f = frame_args(a, v[i]);
        {
            Timed tm(r, true);
            VPT_TRY(launch_fused(r, f));                   // armed: the pass writes the tone mapper's output next to the render buffer
        }
        // (a pass that did not keep the output current — the texels of tiles it skipped are from before a bucket launch — : the separate pass)
        if (!r->tm_valid) VPT_TRY(vpt_tonemapper_render(t, (const vpt_tonemap_params *)&t->table_params));
        VPT_TRY(join_side(r));
        HIP_TRY(hipMemcpyAsync((char *)first_target + (size_t)i * stride_bytes, t->out, need, hipMemcpyDeviceToDevice, r->ctx->stream));
    }
    HIP_TRY(hipGetLastError());
    if (!r->split_callers) VPT_TRY(join_side(r));            // (with VPT_OPTION_SPLIT_CALLER_TARGETS the caller joins, once per bucket)
    r->warmed = true;
    if (r->kind == VPT_RENDERER_MCM) r->samples_host += r->valid_pixels * (uint64_t)base->steps * (uint64_t)count;
    return VPT_OK;
}
extern "C" int vpt_tonemapper_rows(vpt_tonemapper *t, int *rows) {
    if (!t || !rows) return fail(VPT_ERR_INVALID, "null argument");
    *rows = t->rows ? t->rows : (t->source ? t->source->local_h : t->H);
    return VPT_OK;
}
extern "C" int vpt_tonemapper_read(vpt_tonemapper *t, void *dst, size_t nbytes) {
    if (!t || !dst) return fail(VPT_ERR_INVALID, "null argument");
    if (!t->rows) return fail(VPT_ERR_INVALID, "nothing rendered yet");
    size_t need = (size_t)t->W * t->rows * 4;
    if (nbytes < need) return fail(VPT_ERR_INVALID, "destination too small: %zu < %zu", nbytes, need);
    HIP_TRY(hipSetDevice(t->ctx->device));
    VPT_TRY(join_side(t->source));                             // (a fused renderer's split passes write the output from their own streams)
    if (t->source && t->source->ctx->stream != t->ctx->stream) HIP_TRY(hipStreamSynchronize(t->source->ctx->stream));
    HIP_TRY(hipMemcpyAsync(dst, t->out, need, hipMemcpyDeviceToHost, t->ctx->stream));
    HIP_TRY(hipStreamSynchronize(t->ctx->stream));
    return VPT_OK;
}
extern "C" int vpt_tonemapper_output_device(vpt_tonemapper *t, void **ptr, size_t *nbytes) {
    if (!t || !ptr || !nbytes) return fail(VPT_ERR_INVALID, "null argument");
    if (!t->rows) return fail(VPT_ERR_INVALID, "nothing rendered yet");
    VPT_TRY(join_side(t->source));
    *ptr = t->out; *nbytes = (size_t)t->W * t->rows * 4;
    return VPT_OK;
}

extern "C" int vpt_probe_stream_read(vpt_context *c, size_t nbytes, int iterations, double *gb_per_s) {
    if (!c || !gb_per_s) return fail(VPT_ERR_INVALID, "null argument");
    if (nbytes < (1u << 20) || iterations < 1) return fail(VPT_ERR_INVALID, "need at least 1 MiB and one iteration");
    HIP_TRY(hipSetDevice(c->device));
    size_t n16 = nbytes / 16;
    uint4 *buf = nullptr; uint32_t *sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc(&buf, n16 * 16);
    if (e == hipSuccess) e = hipMalloc(&sink, 4);
    if (e == hipSuccess) e = hipMemsetAsync(buf, 0, n16 * 16, c->stream);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float ms = 0.0f;
    if (e == hipSuccess) {
        dim3 grid(256 * 16);                                  // 16 workgroups per CU
        hipLaunchKernelGGL(k_stream_read, grid, dim3(VPT_BLOCK), 0, c->stream, buf, n16, sink);    // warm-up
        e = hipEventRecord(e0, c->stream);
        for (int i = 0; i < iterations && e == hipSuccess; i++) {
            hipLaunchKernelGGL(k_stream_read, grid, dim3(VPT_BLOCK), 0, c->stream, buf, n16, sink);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    }
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    if (buf) hipFree(buf);
    if (sink) hipFree(sink);
    if (e != hipSuccess) return fail(VPT_ERR_HIP, "stream probe: %s", hipGetErrorString(e));
    *gb_per_s = (double)(n16 * 16) * iterations / ((double)ms * 1e-3) / 1e9;
    return VPT_OK;
}

// ---------------------------------------------------------------------------------------------
// multi-GPU frame gather over RCCL (dlopen'ed: the library stays loadable where RCCL is absent)
// ---------------------------------------------------------------------------------------------
typedef struct ncclComm *ncclComm_t_;
typedef struct { char internal[128]; } ncclUniqueId_;
struct Rccl {
    void *handle;
    int (*GetUniqueId)(ncclUniqueId_ *);
    int (*CommInitRank)(ncclComm_t_ *, int, ncclUniqueId_, int);
    int (*CommDestroy)(ncclComm_t_);
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t_, hipStream_t);
    int (*Send)(const void *, size_t, int, int, ncclComm_t_, hipStream_t);
    int (*Recv)(void *, size_t, int, int, ncclComm_t_, hipStream_t);
    int (*GroupStart)(void);
    int (*GroupEnd)(void);
    const char *(*GetErrorString)(int);
};
static Rccl g_rccl = {};
static int rccl_load() {
    if (g_rccl.handle) return VPT_OK;
    const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    void *h = nullptr;
    for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) return fail(VPT_ERR_UNSUPPORTED, "RCCL not loadable: %s", dlerror());
    g_rccl.GetUniqueId = (int (*)(ncclUniqueId_ *))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(ncclComm_t_ *, int, ncclUniqueId_, int))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (int (*)(ncclComm_t_))dlsym(h, "ncclCommDestroy");
    g_rccl.AllGather = (int (*)(const void *, void *, size_t, int, ncclComm_t_, hipStream_t))dlsym(h, "ncclAllGather");
    g_rccl.Send = (int (*)(const void *, size_t, int, int, ncclComm_t_, hipStream_t))dlsym(h, "ncclSend");
    g_rccl.Recv = (int (*)(void *, size_t, int, int, ncclComm_t_, hipStream_t))dlsym(h, "ncclRecv");
    g_rccl.GroupStart = (int (*)(void))dlsym(h, "ncclGroupStart");
    g_rccl.GroupEnd = (int (*)(void))dlsym(h, "ncclGroupEnd");
    g_rccl.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather || !g_rccl.GetErrorString ||
        !g_rccl.Send || !g_rccl.Recv || !g_rccl.GroupStart || !g_rccl.GroupEnd)
        return fail(VPT_ERR_UNSUPPORTED, "RCCL library lacks an expected symbol");
    g_rccl.handle = h;
    return VPT_OK;
}
#define RCCL_TRY(expr) do { int e_ = (expr); if (e_ != 0) \
    return fail(VPT_ERR_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(e_)); } while (0)

#ifndef VPT_GATHER_RING
#define VPT_GATHER_RING 16            // even; 1080p: 16 x (2 MB + 16.6 MB), 2160p: 16 x (8.3 MB + 66 MB) of 288 GB
#endif
struct vpt_gather {
    vpt_renderer *r;
    int rank, world;
    int root;                          // -1: every rank receives the frame (all_gather); else only this rank does
    ncclComm_t_ comm;
    hipStream_t comm_stream;
    size_t send_bytes;                 // W * local_h * 8
    // A ring of VPT_GATHER_RING send / receive buffers: frame k uses buffer k % ring.  Per frame the streams exchange ONE
    // event (kernel done -> the communication stream may send); the reverse edge (buffer free again -> the compute stream
    // may overwrite it) is needed only once per half ring: the gather that ends a half records gathered[half parity], and
    // the compute stream waits for it when it re-enters that half a whole ring later.  (With two buffers the reverse edge
    // was paid every frame: ~11 us of event traffic per frame at a 24 us kernel.)
    void *send[VPT_GATHER_RING], *recv[VPT_GATHER_RING];
    hipEvent_t rendered[2][VPT_MAX_SPLIT], gathered[2];   // rendered: per tile-row range (stream) of a split pass
    uint64_t frames;
    void *assembled;                   // [H][W] RGBA16F scratch for read_frame
};

// gathered [world][local_h][W] -> [H][W]: global row j lives on rank (j / R) % G at local row ((j / R) / G) * R + j % R
__global__ void k_assemble_rows(const uint2 *gathered, uint2 *out, int W, int H, int local_h, int G, int R) {
    int j = (int)blockIdx.y;
    int b = j / R;
    int rank = (G == 1) ? 0 : b % G;
    int lrow = (G == 1) ? j : (b / G) * R + (j - b * R);
    const uint2 *src = gathered + ((size_t)rank * local_h + lrow) * W;
    for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < W; i += (int)(gridDim.x * blockDim.x)) out[(size_t)j * W + i] = src[i];
}

extern "C" int vpt_probe_assemble_rows(vpt_context *c, const void *gathered, int width, int height, int local_rows, int world,
                                       int rows_per_block, void *out) {
    if (!c || !gathered || !out) return fail(VPT_ERR_INVALID, "null argument");
    if (width < 1 || height < 1 || local_rows < 1 || world < 1 || rows_per_block < 1) return fail(VPT_ERR_INVALID, "bad geometry");
    // every global row must exist in its owner's block of the gathered buffer
    int blocks = (height + rows_per_block - 1) / rows_per_block;
    int max_local = ((blocks + world - 1) / world) * rows_per_block;
    if (local_rows < max_local) return fail(VPT_ERR_INVALID, "local_rows %d < %d needed for %d rows over %d ranks", local_rows, max_local, height, world);
    HIP_TRY(hipSetDevice(c->device));
    size_t in_bytes = (size_t)world * local_rows * width * 8, out_bytes = (size_t)width * height * 8;
    void *din = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc(&din, in_bytes));
    hipError_t e = hipMalloc(&dout, out_bytes);
    if (e == hipSuccess) e = hipMemcpyAsync(din, gathered, in_bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_assemble_rows, dim3((unsigned)((width + 255) / 256), (unsigned)height), dim3(256), 0, c->stream,
                           (const uint2 *)din, (uint2 *)dout, width, height, local_rows, world, rows_per_block);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout, out_bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(din); if (dout) hipFree(dout);
    if (e != hipSuccess) return fail(VPT_ERR_HIP, "assemble probe: %s", hipGetErrorString(e));
    return VPT_OK;
}
extern "C" int vpt_gather_unique_id(void *id128) {
    if (!id128) return fail(VPT_ERR_INVALID, "id is null");
    VPT_TRY(rccl_load());
    ncclUniqueId_ id;
    RCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return VPT_OK;
}
extern "C" int vpt_gather_destroy(vpt_gather *g) {
    if (!g) return VPT_OK;
    hipSetDevice(g->r->ctx->device);
    join_side(g->r);                                         // ranges of split passes still rendering into the ring
    hipStreamSynchronize(g->r->ctx->stream);
    if (g->comm_stream) hipStreamSynchronize(g->comm_stream);
    vpt_renderer_set_render_target(g->r, nullptr, 0);
    if (g->comm) g_rccl.CommDestroy(g->comm);
    for (int b = 0; b < VPT_GATHER_RING; b++) {
        if (g->send[b]) hipFree(g->send[b]);
        if (g->recv[b]) hipFree(g->recv[b]);
    }
    for (int b = 0; b < 2; b++) {
        for (int i = 0; i < VPT_MAX_SPLIT; i++) if (g->rendered[b][i]) hipEventDestroy(g->rendered[b][i]);
        if (g->gathered[b]) hipEventDestroy(g->gathered[b]);
    }
    if (g->assembled) hipFree(g->assembled);
    if (g->comm_stream) hipStreamDestroy(g->comm_stream);
    delete g;
    return VPT_OK;
}
extern "C" int vpt_gather_create(vpt_renderer *r, const void *id128, int rank, int world, vpt_gather **out) {
    if (!r || !id128 || !out) return fail(VPT_ERR_INVALID, "null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(VPT_ERR_INVALID, "bad rank %d / world %d", rank, world);
    if (r->G != world || r->g != rank) return fail(VPT_ERR_INVALID, "renderer is sharded %d/%d, gather asked for %d/%d", r->g, r->G, rank, world);
    VPT_TRY(rccl_load());
    HIP_TRY(hipSetDevice(r->ctx->device));
    vpt_gather *g = new vpt_gather();
    memset(g, 0, sizeof(*g));
    g->r = r; g->rank = rank; g->world = world; g->root = -1;
    g->send_bytes = (size_t)r->W * r->local_h * 8;
    int rc = VPT_OK;
    hipError_t e;
    {   // the communication stream must overlap the streams the passes run on
        { int jr = join_side(r); if (jr != VPT_OK) { delete g; return jr; } }
        hipStream_t others[VPT_MAX_SPLIT] = { r->ctx->stream };
        for (int k = 0; k < VPT_MAX_SPLIT - 1; k++) others[1 + k] = r->side[k];
        hipStreamSynchronize(r->ctx->stream);
        e = create_overlapping_stream(&g->comm_stream, others, VPT_MAX_SPLIT);
    }
    for (int b = 0; b < 2 && e == hipSuccess; b++) {
        for (int i = 0; i < VPT_MAX_SPLIT && e == hipSuccess; i++) e = hipEventCreateWithFlags(&g->rendered[b][i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g->gathered[b], hipEventDisableTiming);
    }
    for (int b = 0; b < VPT_GATHER_RING && e == hipSuccess; b++) {
        e = hipMalloc(&g->send[b], g->send_bytes);
        if (e == hipSuccess) e = hipMalloc(&g->recv[b], g->send_bytes * world);
        if (e == hipSuccess) e = hipMemset(g->send[b], 0, g->send_bytes);
    }
    if (e == hipSuccess) e = hipMalloc(&g->assembled, (size_t)r->W * r->H * 8);
    if (e != hipSuccess) rc = fail(VPT_ERR_HIP, "gather buffers: %s", hipGetErrorString(e));
    if (rc == VPT_OK) {
        ncclUniqueId_ id;
        memcpy(&id, id128, sizeof(id));
        int ne = g_rccl.CommInitRank(&g->comm, world, id, rank);
        if (ne != 0) { g->comm = nullptr; rc = fail(VPT_ERR_HIP, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(ne)); }
    }
    if (rc != VPT_OK) { char keep[512]; strncpy(keep, g_err, sizeof(keep)); keep[511] = 0; vpt_gather_destroy(g); strncpy(g_err, keep, sizeof(g_err)); return rc; }
    *out = g;
    return VPT_OK;
}
// per-launch timing (vpt_renderer_set_profiling) applies to the pipeline's kernel as well
static void profile_events(vpt_renderer *r, hipEvent_t *t0, hipEvent_t *t1) {
    *t0 = *t1 = nullptr;
    if (!(r->profiling && (r->profile_seq++ % (uint64_t)r->profile_every) == 0)) return;
    if (r->events_used == r->events.size()) {
        hipEvent_t e0, e1;
        if (hipEventCreate(&e0) != hipSuccess) return;
        if (hipEventCreate(&e1) != hipSuccess) { hipEventDestroy(e0); return; }
        r->events.push_back({ e0, e1 }); r->event_launches.push_back(1);
    }
    *t0 = r->events[r->events_used].first; *t1 = r->events[r->events_used].second;
    r->event_launches[r->events_used] = 1; r->events_used++;
}
// The schedule of one frame of the gather pipeline as a PURE function of (frame index, rank, world, root): which ring buffer,
// which event edges, where the kernel renders, which RCCL operations.  gather_enqueue_frame executes exactly this plan, and
// tests/test_gather_schedule.py checks it on the CPU for world 2..8 (matching send / receive pairs, disjoint receive slots
// that cover the frame, no ring buffer rewritten before its gather was waited for) — the `world > 1` operations cannot be
// exercised on a one-GPU box.
extern "C" int vpt_gather_plan(uint64_t frame, int rank, int world, int root, uint64_t send_bytes, vpt_gather_step *out) {
    if (!out) return fail(VPT_ERR_INVALID, "null argument");
    if (world < 1 || rank < 0 || rank >= world || root < -1 || root >= world) return fail(VPT_ERR_INVALID, "bad rank / world / root");
    static_assert(VPT_GATHER_RING >= 2 && VPT_GATHER_RING % 2 == 0, "the ring is split into two halves");
    const uint64_t half = VPT_GATHER_RING / 2;
    memset(out, 0, sizeof(*out));
    out->ring = VPT_GATHER_RING;
    out->buffer = (int)(frame % VPT_GATHER_RING);
    out->parity = (int)((frame / half) & 1);
    // entering a half of the ring again: the gathers that used these buffers a ring ago must have drained
    out->wait_gathered = (frame % half == 0 && frame >= VPT_GATHER_RING) ? 1 : 0;
    // this half's last gather publishes "buffers of this half are free again"
    out->record_gathered = ((frame + 1) % half == 0) ? 1 : 0;
    out->rendered_event = out->buffer & 1;
    // the receiving rank of a rooted gather renders straight into its own slot of the receive buffer
    out->in_place = (root == rank) ? 1 : 0;
    out->render_offset = out->in_place ? (uint64_t)rank * send_bytes : 0;
    if (root < 0) {
        out->op = VPT_GATHER_OP_ALLGATHER;
    } else if (world == 1) {
        out->op = VPT_GATHER_OP_NONE;
    } else if (out->in_place) {
        out->op = VPT_GATHER_OP_RECV;                 // one grouped receive per peer, slot p of the receive buffer
        out->npeers = world - 1;
    } else {
        out->op = VPT_GATHER_OP_SEND;
        out->peer = root;
    }
    return VPT_OK;
}
// the i-th receive of a VPT_GATHER_OP_RECV step: peer rank and byte offset of its slot in the receive buffer
extern "C" int vpt_gather_plan_recv(const vpt_gather_step *st, int rank, int i, uint64_t send_bytes, int *peer, uint64_t *offset) {
    if (!st || !peer || !offset) return fail(VPT_ERR_INVALID, "null argument");
    if (st->op != VPT_GATHER_OP_RECV || i < 0 || i >= st->npeers) return fail(VPT_ERR_INVALID, "not a receive of this step");
    int p = i < rank ? i : i + 1;                     // every rank but this one, ascending
    *peer = p; *offset = (uint64_t)p * send_bytes;
    return VPT_OK;
}
static int gather_enqueue_frame(vpt_gather *g, PassArgs &a, hipEvent_t t0, hipEvent_t t1, uint32_t fused_passes);
extern "C" int vpt_gather_render(vpt_gather *g, const vpt_uniforms *u) {
    if (!g || !u) return fail(VPT_ERR_INVALID, "null argument");
    vpt_renderer *r = g->r;
    HIP_TRY(hipSetDevice(r->ctx->device));
    if (r->kind == VPT_RENDERER_MIP || r->kind == VPT_RENDERER_EAM) VPT_TRY(check_step(u));
    PassArgs a;
    VPT_TRY(make_args(r, u, true, &a));
    hipEvent_t t0, t1;
    profile_events(r, &t0, &t1);
    VPT_TRY(gather_enqueue_frame(g, a, t0, t1, 0));
    if (r->kind == VPT_RENDERER_MCM) r->samples_host += r->valid_pixels * (uint64_t)u->steps;
    r->warmed = true;
    return VPT_OK;
}
// one frame of the gather pipeline on (compute stream cs, communication stream)
// fused_passes > 0: the frame is the result of that many MCM passes run by one k_mcm_multi launch (a carries the frame table)
static int gather_enqueue_frame(vpt_gather *g, PassArgs &a, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr, uint32_t fused_passes = 0) {
    vpt_renderer *r = g->r;
    hipStream_t cs = r->ctx->stream;
    if ((size_t)r->W * r->local_h * 8 != g->send_bytes || r->G != g->world || r->g != g->rank)
        return fail(VPT_ERR_INVALID, "the renderer was resized or re-sharded after the gather was created: destroy and re-create the gather");
    vpt_gather_step st;
    VPT_TRY(vpt_gather_plan(g->frames, g->rank, g->world, g->root, g->send_bytes, &st));
    const int b = st.buffer;
    if (st.wait_gathered) {                                                  // on every stream that may carry a range of this frame
        HIP_TRY(hipStreamWaitEvent(cs, g->gathered[st.parity], 0));
        for (int i = 0; i < VPT_MAX_SPLIT - 1; i++) if (r->side[i]) HIP_TRY(hipStreamWaitEvent(r->side[i], g->gathered[st.parity], 0));
    }
    r->last_ranges = 1;
    // the "rendered" events ride on the dispatches themselves (hipExtLaunchKernel stop events): a hipEventRecord behind the kernel
    // is a barrier packet of its own on the compute queue, 3-4.5 us per frame at every frame size (tools/r02_exp24.sh)
    r->stop_events = fused_passes ? nullptr : g->rendered[st.rendered_event]; r->stop_used = false;
    a.render = st.in_place ? (uint2 *)((char *)g->recv[b] + st.render_offset) : (uint2 *)g->send[b];
    r->render_target = a.render;                                             // vpt_renderer_read(RENDER) returns the last frame's rows
    a.tm_table = nullptr; r->tm_valid = false;                               // (a fused tone mapper follows the renderer's own buffer only)
    if (t0) HIP_TRY(hipEventRecord(t0, cs));
    if (fused_passes) {
        VPT_TRY(launch_mcm_multi(r, a, fused_passes));
        hipLaunchKernelGGL(k_advance_frames, dim3(1), dim3(1), 0, cs, r->frame_counter, fused_passes);
    } else {
        VPT_TRY(launch_fused(r, a));
    }
    if (t1) HIP_TRY(hipEventRecord(t1, cs));
    // A split pass (VPT_OPTION_SPLIT_STREAMS): the communication stream waits for every range; the ranges' streams are NOT joined,
    // so range i of the next frame starts behind range i of this one, whatever the other ranges and the gather are doing.
    const bool rode = r->stop_used;                                          // the events were attached to the launches themselves
    r->stop_events = nullptr; r->stop_used = false;
    for (int i = 0; i < r->last_ranges; i++) {
        hipStream_t s = i == 0 ? cs : r->side[i - 1];
        if (!rode) HIP_TRY(hipEventRecord(g->rendered[st.rendered_event][i], s));
        HIP_TRY(hipStreamWaitEvent(g->comm_stream, g->rendered[st.rendered_event][i], 0));
    }
    if (st.op == VPT_GATHER_OP_ALLGATHER) {
        RCCL_TRY(g_rccl.AllGather(g->send[b], g->recv[b], g->send_bytes, /*ncclUint8*/ 1, g->comm, g->comm_stream));
    } else if (st.op != VPT_GATHER_OP_NONE) {
        // gather to the display rank: its 7 peers send over 7 distinct xGMI links at once
        RCCL_TRY(g_rccl.GroupStart());
        int ne = 0;
        if (st.op == VPT_GATHER_OP_RECV) {
            for (int i = 0; i < st.npeers && ne == 0; i++) {
                int p; uint64_t off;
                VPT_TRY(vpt_gather_plan_recv(&st, g->rank, i, g->send_bytes, &p, &off));
                ne = g_rccl.Recv((char *)g->recv[b] + off, g->send_bytes, /*ncclUint8*/ 1, p, g->comm, g->comm_stream);
            }
        } else {
            ne = g_rccl.Send(g->send[b], g->send_bytes, /*ncclUint8*/ 1, st.peer, g->comm, g->comm_stream);
        }
        int ge = g_rccl.GroupEnd();
        if (ne != 0) return fail(VPT_ERR_HIP, "ncclSend/ncclRecv failed: %s", g_rccl.GetErrorString(ne));
        if (ge != 0) return fail(VPT_ERR_HIP, "ncclGroupEnd failed: %s", g_rccl.GetErrorString(ge));
    }
    if (st.record_gathered) HIP_TRY(hipEventRecord(g->gathered[st.parity], g->comm_stream));   // this half's last gather
    g->frames++;
    return VPT_OK;
}
// `count` frames by one call.  A captured hipGraph holding the RCCL all-gathers was measured 6x slower per frame than
// this eager enqueue and unstable over many replays on ROCm 7.0 / RCCL 2.26, so the sequence is
// always enqueued eagerly: two stream operations per frame on each of the two streams.
extern "C" int vpt_gather_play(vpt_gather *g, const vpt_uniforms *base, const float *frame_vars, int count, int mode) {
    if (!g || !base || !frame_vars) return fail(VPT_ERR_INVALID, "null argument");
    if (mode != VPT_PLAY_EAGER && mode != VPT_PLAY_FUSED) return fail(VPT_ERR_UNSUPPORTED, "the gather pipeline plays eagerly or with fused passes (no graph replay)");
    vpt_renderer *r = g->r;
    HIP_TRY(hipSetDevice(r->ctx->device));
    PassArgs a;
    VPT_TRY(play_args(r, base, count, &a));
    const FrameVar *v = (const FrameVar *)frame_vars;
    if (mode == VPT_PLAY_FUSED) {
        // `count` passes in one launch, then ONE gather of the resulting frame
        if (r->kind != VPT_RENDERER_MCM) return fail(VPT_ERR_UNSUPPORTED, "fused passes are implemented for the MCM renderer only");
        VPT_TRY(play_upload_table(r, frame_vars, count, &a));
        hipEvent_t t0 = nullptr, t1 = nullptr;
        profile_events(r, &t0, &t1);
        if (t0) r->event_launches[r->events_used - 1] = (uint32_t)count;
        VPT_TRY(gather_enqueue_frame(g, a, t0, t1, (uint32_t)count));
    }
    for (int i = 0; i < count && mode == VPT_PLAY_EAGER; i++) {
        PassArgs f = frame_args(a, v[i]);
        hipEvent_t t0 = nullptr, t1 = nullptr;
        profile_events(r, &t0, &t1);
        VPT_TRY(gather_enqueue_frame(g, f, t0, t1, 0));
    }
    HIP_TRY(hipGetLastError());
    r->warmed = true;
    if (r->kind == VPT_RENDERER_MCM) r->samples_host += r->valid_pixels * (uint64_t)base->steps * (uint64_t)count;
    return VPT_OK;
}
extern "C" int vpt_gather_set_root(vpt_gather *g, int root) {
    if (!g) return fail(VPT_ERR_INVALID, "gather is null");
    if (root < -1 || root >= g->world) return fail(VPT_ERR_INVALID, "root %d outside [-1, %d)", root, g->world);
    VPT_TRY(vpt_gather_synchronize(g));                     // frames in flight keep the mode they were enqueued with
    g->root = root;
    return VPT_OK;
}
extern "C" int vpt_gather_synchronize(vpt_gather *g) {
    if (!g) return fail(VPT_ERR_INVALID, "gather is null");
    HIP_TRY(hipSetDevice(g->r->ctx->device));
    VPT_TRY(join_side(g->r));
    HIP_TRY(hipStreamSynchronize(g->r->ctx->stream));
    HIP_TRY(hipStreamSynchronize(g->comm_stream));
    return VPT_OK;
}
extern "C" int vpt_gather_read_frame(vpt_gather *g, void *dst, size_t nbytes) {
    if (!g || !dst) return fail(VPT_ERR_INVALID, "null argument");
    if (g->frames == 0) return fail(VPT_ERR_INVALID, "no frame has been gathered yet");
    if (g->root >= 0 && g->root != g->rank) return fail(VPT_ERR_INVALID, "rank %d does not receive frames: the gather is rooted at rank %d", g->rank, g->root);
    vpt_renderer *r = g->r;
    size_t need = (size_t)r->W * r->H * 8;
    if (nbytes < need) return fail(VPT_ERR_INVALID, "destination too small: %zu < %zu", nbytes, need);
    HIP_TRY(hipSetDevice(r->ctx->device));
    int b = (int)((g->frames - 1) % VPT_GATHER_RING);       // ordered behind that frame's gather by the communication stream itself
    hipLaunchKernelGGL(k_assemble_rows, dim3((unsigned)((r->W + 255) / 256), (unsigned)r->H), dim3(256), 0, g->comm_stream,
                       (const uint2 *)g->recv[b], (uint2 *)g->assembled, r->W, r->H, r->local_h, r->G, r->R);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(dst, g->assembled, need, hipMemcpyDeviceToHost, g->comm_stream));
    HIP_TRY(hipStreamSynchronize(g->comm_stream));
    return VPT_OK;
}
