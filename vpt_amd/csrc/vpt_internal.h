// vpt_internal.h — what the translation units of libvpt_hip.so share: the objects behind the C-ABI handles (include/vpt.h), the
// error plumbing and the launch helpers.  Translation units (Makefile; built in parallel, linked into one library):
//   vpt_core.hip    context, volume (upload, re-layout), renderer life cycle, tile classification, options, read-back, probes
//   vpt_mcm.hip     the MCM passes (vpt_kernels_mcm.h): tile classes, bucket kernels, reset / render / materialise
//   vpt_mcm_hit.hip the MCM integrate kernels of every variant (k_mcm_integrate*), handed to vpt_mcm.hip by variant (vpt_mcm_select.h)
//   vpt_mcm_seq.hip MCM frame sequences in one launch (k_mcm_multi, k_mcm_frames)
//   vpt_march.hip   MIP, EAM, MCS passes (vpt_kernels_march.h)
//   vpt_extra.hip   ISO, Depth, LAO, DOS passes (vpt_kernels_iso_depth.h)
//   vpt_render.hip  the renderer entry points: the four hooks, render(), frame sequences (vpt_renderer_play*)
//   vpt_post.hip    what follows a frame: tone mappers, the RCCL frame gather
// Nothing device-side crosses a translation unit: a kernel is compiled by the unit that names it (the three MCM units share one header).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_ext.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include <string>
#include <utility>
#include <vector>

#include "../../include/vpt.h"
#include "vpt_kernels.h"

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));     // vpt_core.hip; sets vpt_last_error()
char *vpt_error_buffer(void);                                                         // the calling thread's message (512 bytes)
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    return fail(VPT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define VPT_TRY(expr) do { int r_ = (expr); if (r_ != VPT_OK) return r_; } while (0)

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
    uint32_t *atlas;       // boundary atlas: the six outer voxel planes as 2 x 2-footprint cells (vpt_device.h sample_volume_boundary): one dword per cell
                           // and channel (byte volumes) or one float4 (float volumes); channel c's faces 6 * atlas_face cells behind c - 1's
    size_t atlas_dwords;
    bool atlas_ok;         // float volumes: every texel is finite and < 1e37 (k_scan_finite at finalize): else the atlas is not used
    uint32_t *atlas_flag;
    uint32_t atlas_face, atlas_shift;   // dwords per face image (row pitch x rows), log2 of the row pitch
    // column records (vpt_device.h record_addr; one-channel byte volumes): built on the first MCM pass that wants them (volume_records)
    uint8_t *records; size_t rec_bytes; bool rec_valid, rec_wide;
    uint32_t *rtab32, *rtabc;           // RX | RY: byte offsets of the columns / their Z-order codes
};

// Tile classes (vpt_kernels.h, "Tile classes"): per reset the host sorts the 16x16 tiles into those none of whose camera rays
// can meet the cube (MISS) and the rest (HIT).  While every pass uses the reset's matrix (and blur == 0) a MISS tile's photons
// never enter the cube, so its passes run k_mcm_miss on 32 B of state; `stale` says that the position / transmittance arrays
// of the MISS tiles are behind and k_mcm_materialize must run before anything but k_mcm_miss looks at them.
#define VPT_COMPLETE_DESTS 40
struct TileClasses {
    bool enabled, verify;          // VPT_OPTION_TILE_CLASSES (default on), VPT_OPTION_VERIFY_TILE_CLASSES
    bool one_stream;               // VPT_OPTION_TILE_CLASSES = 2: the MCM class kernels also where they must follow each other on one stream (each alone on the chip: measurements)
    bool valid;                    // the lists describe `mvp` for the present geometry, and every pass since that reset used it
    float mvp[16];
    uint32_t *list; int capacity;  // device: n_hit HIT tiles, then n_miss MISS tiles, each tx | ty << 16
    int n_hit, n_miss;
    // what the lists on the device were built for: a reset with the same matrix and geometry (the interactive case: a transfer function or a
    // parameter changed, the camera did not) re-uses them — no classification, no upload (classes_build)
    bool built; float built_mvp[16]; int built_geom[6];
    // uploads go through two pinned staging buffers in turn, no host wait: staged[i] = the copy out of staging[i] has been enqueued and completes
    uint32_t *staging[2]; int staging_capacity[2]; hipEvent_t staged[2]; int stage_next;
    bool stale, stale_fast;        // MISS tiles' position / transmittance arrays are behind; the pass that left them ran the fast variant
    unsigned long long *violations;
    // the accumulating ray marchers (MIP, EAM, ISO, MCS, Depth): see marcher_track
    uint64_t passes, fused_passes; // generate / fused passes since the reset
    bool poisoned;                 // a pass since the reset used another matrix than the first: nothing can be skipped until the next reset
    bool first_mix_one;            // the first pass since the reset was a fused pass with mix == 1 (MCS, Depth: accumulator = frame exactly)
    bool list_now;                 // the launch being enqueued covers the HIT tiles only
    bool reset_seen;               // vpt_renderer_reset has run on the present buffers (zero-filled buffers are not a reset)
    // render destinations (the renderer's own buffer, a caller's target, the slots of a bucket or of the gather ring) that a WHOLE-image fused
    // pass has written since the reset: only there do the skipped tiles hold their final texels (marcher_track)
    const void *complete[VPT_COMPLETE_DESTS]; int n_complete;
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
    bool bucket_call;              // inside vpt_renderer_play_into*: the passes into the caller's bucket may use every stream, the call joins them before it returns
    int last_ranges;               // how many tile-row ranges (streams) the last sampling launch used
    hipEvent_t *stop_events;       // gather pipeline: event i is attached to range i's launch (hipExtLaunchKernel stop event: the
    bool stop_used;                // dispatch packet's own completion signal, no barrier packet behind the kernel)
    int split; bool split_auto;     // split_auto: the stream count is the library's default and follows the launch size (split_for)
    hipStream_t side[VPT_MAX_SPLIT - 1]; hipEvent_t ev_fork, ev_join[VPT_MAX_SPLIT - 1]; bool side_busy, main_dirty;
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
    int hit_form;                  // VPT_HIT_KERNEL_FORM in the environment at creation (A/B and tests): 0 = by the number of HIT tiles, 1 = k_mcm_integrate, 2 = k_mcm_integrate_early
    int column_records;            // VPT_OPTION_COLUMN_RECORDS: 0 = bricks, 1 = column records, 2 (default) = records where the bricks exceed VPT_RECORDS_AUTO_BYTES
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

static const size_t COUNTER_BYTES = (size_t)VPT_COUNTER_SLOTS * VPT_COUNTER_STRIDE * sizeof(unsigned long long);

struct PlayGraph;

static inline size_t frame_elem(int kind) {
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
// shared host functions (vpt_core.hip unless noted)
// ---------------------------------------------------------------------------------------------
int ensure_split_streams(vpt_renderer *r);          // creates the side streams r->split asks for, if they do not exist yet
int join_side(vpt_renderer *r);                     // the side streams' work happens-before everything enqueued on the context's stream from here on
int make_args(vpt_renderer *r, const vpt_uniforms *u, bool need_volume, PassArgs *a);
int volume_records(vpt_volume *v);                  // builds the column records of a finalized one-channel byte volume if they are not current
hipError_t create_overlapping_stream(hipStream_t *out, const hipStream_t *others, int n_others);
bool invert_matrix(const float *m, double out[4][4]);               // column-major float matrix -> its inverse (double); false: singular
int classes_build(vpt_renderer *r, const float *mvp_inverse);       // tile lists of `mvp_inverse` on the device (classify_tiles)
void play_graph_free(PlayGraph *g);                                 // vpt_render.hip
void tonemappers_unbind(vpt_context *c, vpt_renderer *r);           // vpt_post.hip
void advance_frames(vpt_renderer *r, uint32_t n);                   // vpt_render.hip: the device frame counter of the graph path += n

// the renderer families behind the entry points of vpt_render.hip
int march_reset(vpt_renderer *r, const PassArgs &a);                // vpt_march.hip: MIP, EAM, MCS
int march_generate(vpt_renderer *r, const PassArgs &a);
int march_integrate(vpt_renderer *r, const PassArgs &a);
int march_render_frame(vpt_renderer *r, const PassArgs &a);
int march_fused(vpt_renderer *r, const PassArgs &a);
int extra_reset(vpt_renderer *r, const PassArgs &a);                // vpt_extra.hip: ISO, Depth, LAO, DOS
int extra_generate(vpt_renderer *r, const PassArgs &a);
int extra_integrate(vpt_renderer *r, const PassArgs &a);
int extra_render_frame(vpt_renderer *r, const PassArgs &a);
int extra_fused(vpt_renderer *r, const PassArgs &a);
int mcm_reset(vpt_renderer *r, const PassArgs &a, const vpt_uniforms *u);     // vpt_mcm.hip
int mcm_pass(vpt_renderer *r, const PassArgs &a, bool fuse_render);           // one integrate pass (tile classes where they are in force)
int mcm_render_frame(vpt_renderer *r, const PassArgs &a);
int mcm_multi(vpt_renderer *r, const PassArgs &a, uint32_t npasses, uint2 *ring);
int mcm_before_pass(vpt_renderer *r, const PassArgs &a, bool *same_matrix);
int mcm_materialize(vpt_renderer *r);
int mcm_bucket_ready(vpt_renderer *r, const PassArgs &a, bool *ready);
int mcm_bucket(vpt_renderer *r, const PassArgs &a, const FrameVar *v, int count, void *ring, uint32_t slot_pixels, bool last_to_render_buffer,
               const uint8_t *display_table);
int launch_fused(vpt_renderer *r, const PassArgs &a);               // vpt_render.hip: the fused render() launch of the renderer's kind

struct BucketCall {              // scope of a vpt_renderer_play_into* call (vpt_renderer.bucket_call)
    vpt_renderer *r;
    explicit BucketCall(vpt_renderer *r_) : r(r_) { r->bucket_call = true; }
    ~BucketCall() { r->bucket_call = false; }
};
// frame sequences (vpt_render.hip)
int play_args(vpt_renderer *r, const vpt_uniforms *base, int count, PassArgs *a);
int play_upload_table(vpt_renderer *r, const float *vars, int count, PassArgs *a);
int check_step(const vpt_uniforms *u);
static inline PassArgs frame_args(const PassArgs &a, const FrameVar &v) {      // eager frames carry their uniforms in the kernel arguments
    PassArgs f = a;
    f.seed = v.seed; f.offset = v.offset; f.mix = v.mix; f.light = f3{ v.lx, v.ly, v.lz };
    return f;
}

static inline bool is_march_kind(int k) { return k == VPT_RENDERER_MIP || k == VPT_RENDERER_EAM || k == VPT_RENDERER_MCS; }

// dynamic LDS of the sampling kernels: transfer-function pairs + the three brick-offset tables
static inline size_t lds_bytes(const vpt_renderer *r) {
    const vpt_volume *v = r->vol;
    return (size_t)r->tf_w * 2 * sizeof(float4) + (size_t)(v->nx + v->ny + v->nz) * 4;
}
static inline dim3 tile_grid(const vpt_renderer *r) { return dim3((unsigned)(r->tiles_x + 7) / 8u * 8u, (unsigned)r->tiles_y); }
// Ray-marching kernels (MIP, EAM, ISO, Depth, MCS) run as one-wave workgroups when 28 of their LDS images fit a CU: with
// the default camera only ~20 % of the tiles cross the cube, about one resident round of 4-wave workgroups, which the
// dispatcher cannot rebalance (measured: 3.3e11 samples/s against 5.7e11 when every tile crosses the cube).
static inline bool wave_blocks(const vpt_renderer *r) {
    return r->kind != VPT_RENDERER_MCM && r->kind != VPT_RENDERER_DOS && lds_bytes(r) * 28 <= 150 * 1024;
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
// The streams a sampling launch of `tiles` tiles is dealt to.  The library's default counts (vpt_core.hip default_split) are those of a
// 1080p frame; a small frame is a handful of workgroups per stream and the fork / join edges cost more than the overlap returns — measured
// per renderer at 256^2 / 512^2 / 1024^2 (us per frame on 1 | 2 | 3 streams): EAM 39 | 46 | 49, 38 | 37 | 38, 48 (3); MIP 36 | 45 | 54,
// 40 | 34 | 38; MCS 9.8 | 12.0, 10.4 | 12.2, 14.9 | 13.8; ISO 38.8 | 43.6, 39.9 | 42.2, 45.7 | 42.4; Depth 41 | 46 | 43, 41 | 40 | 41.
// A count set through VPT_OPTION_SPLIT_STREAMS is taken as it is.
static inline int split_for(const vpt_renderer *r, int tiles) {
    int k = r->split;
    if (r->split_auto && k > 1) {
        const int per = (r->kind == VPT_RENDERER_MIP || r->kind == VPT_RENDERER_EAM || r->kind == VPT_RENDERER_DEPTH) ? 192 : (r->kind == VPT_RENDERER_LAO ? 32 : 384);
        k = std::min(k, std::max(1, tiles / per));
    }
    return k;
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
    const bool split = r->split >= 2 && !r->no_split && (!r->target_is_callers || r->bucket_call);
    if (split) VPT_TRY(ensure_split_streams(r));
    // (a frame rendered into caller memory — vpt_renderer_set_render_target — is consumed by work the caller enqueues on the
    // context's stream right behind it: such passes stay on that stream unless the caller has taken the join upon itself
    // (vpt_renderer_play_into*: one join per bucket of frames, at the end of the call).  The gather pipeline waits for every range itself.)
    if (r->cls.list_now) {
        // the HIT tiles only (marcher_track): K equal parts of the list on the K streams
        const int k = split ? std::min(split_for(r, r->cls.n_hit), r->cls.n_hit) : 1;
        if (r->side_busy && (r->last_layout != 1 || r->last_ranges != k)) VPT_TRY(join_side(r));      // the tile -> stream map changes
        r->last_layout = 1;
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
    const int kw = split ? split_for(r, r->tiles_x * r->tiles_y) : 1;
    if (r->side_busy && (r->last_layout != 0 || r->last_ranges != kw)) VPT_TRY(join_side(r));     // the previous pass dealt the tiles to the streams in another way
    r->last_layout = 0;
    if (split && kw >= 2 && r->tiles_y >= kw) {
        dim3 g = tile_grid(r);
        const unsigned k = (unsigned)kw;
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
        r->last_ranges = 1;
    }
    return VPT_OK;
}
// the instantiation for (addressing, filter, channels): V = VPT_V_WIDE | VPT_V_NEAREST | VPT_V_RG bits
// MCM on a one-channel byte volume with the LINEAR filter: the in-cube samples come from the column records (VPT_OPTION_COLUMN_RECORDS).
// Measured (round 4, 1080p headline camera, us per frame bricks -> records): 512^3 80.0 -> 83.3, every tile HIT 143.9 -> 157.1, extinction 50
// 80.3 -> 92.9 — 256 MiB of bricks mostly live in the 256 MB Infinity Cache and a dense medium's short steps re-use brick lines, 512 MiB of
// records do neither —; 1024^3 (2 GiB of bricks, beyond every cache) 98.5 -> 96.6, HIT + MISS kernels alone 127.2 -> 122.0.  Hence AUTO.
#define VPT_RECORDS_AUTO_BYTES (512ull << 20)
static inline bool renderer_uses_records(const vpt_renderer *r) {
    const vpt_volume *v = r->vol;
    if (!(r->kind == VPT_RENDERER_MCM && v && v->channels == 1 && !v->f32 && v->filter == VPT_FILTER_LINEAR && v->rtab32 != nullptr)) return false;
    return r->column_records == 1 || (r->column_records == 2 && v->brick_bytes > VPT_RECORDS_AUTO_BYTES);
}
static inline int variant_of(const vpt_renderer *r) {
    return ((r->vol->wide || (renderer_uses_records(r) && r->vol->rec_wide)) ? VPT_V_WIDE : 0) | (r->vol->filter == VPT_FILTER_NEAREST ? VPT_V_NEAREST : 0) | (r->vol->channels == 2 ? VPT_V_RG : 0) |
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
