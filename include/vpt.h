/*
 * vpt.h — C-ABI of the MI355X-native renderer path (libvpt_hip.so).
 *
 * Drop-in boundary for the per-pixel generate / integrate / render / reset passes of the
 * MIP, EAM, MCS and MCM renderers of MOj0/vpt.  Every entry point cites the reference
 * interface (file:line under the reference tree) it replaces.  Plain pointers and sizes
 * only; no C++ or torch types.  All functions return VPT_OK (0) or a negative error code;
 * vpt_last_error() gives the message (the reference only throws Error(msg):
 * WebGL.js:14,28,180; Volume.js:40,103 — bindings convert non-zero into a thrown Error).
 *
 * Threading: like the reference (single JS thread, Ticker.js:5-8) a context is not
 * thread-safe.  Calls enqueue work on the context's HIP stream and return; only
 * vpt_context_synchronize, vpt_renderer_read* and vpt_renderer_sample_count block.
 *
 * Image convention: row 0 is the BOTTOM row (GL framebuffer origin), pixels are x-fastest.
 */
#ifndef VPT_H
#define VPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VPT_API __attribute__((visibility("default")))

#define VPT_OK               0
#define VPT_ERR_INVALID     -1   /* bad argument / bad state */
#define VPT_ERR_HIP         -2   /* HIP runtime error */
#define VPT_ERR_NO_VOLUME   -3   /* renderer has no (ready) volume: Volume.getTexture() == null, Volume.js:107-113 */
#define VPT_ERR_UNSUPPORTED -4

/* RendererFactory.js:10-23: all eight names ('mip' | 'iso' | 'eam' | 'lao' | 'mcs' | 'mcm' | 'dos' | 'depth') */
#define VPT_RENDERER_MIP 0
#define VPT_RENDERER_EAM 1
#define VPT_RENDERER_MCS 2
#define VPT_RENDERER_MCM 3
#define VPT_RENDERER_ISO   4   /* src/js/renderers/ISORenderer.js, src/glsl/renderers/ISORenderer.glsl */
#define VPT_RENDERER_DEPTH 5   /* src/js/renderers/DepthRenderer.js, src/glsl/renderers/DepthRenderer.glsl */
#define VPT_RENDERER_LAO   6   /* src/js/renderers/LAORenderer.js, src/glsl/renderers/LAORenderer.glsl */
#define VPT_RENDERER_DOS   7   /* src/js/renderers/DOSRenderer.js, src/glsl/renderers/DOSRenderer.glsl */

/* Volume.js:115-125 setFilter('linear' | 'nearest') */
#define VPT_FILTER_NEAREST 0
#define VPT_FILTER_LINEAR  1

/* Volume formats (RAWReader.js:36-38: format RED, internalFormat R8, type UNSIGNED_BYTE) */
#define VPT_FORMAT_R8 0
#define VPT_FORMAT_R32F 2        /* format RED, internalFormat R32F (or R16F: widened exactly on upload), type FLOAT / HALF_FLOAT: the texel value
                                  * itself (not normalised), LINEAR-filtered like the byte formats (OES_texture_float_linear, RenderingContext.js:78);
                                  * blocks are uploaded as float32 */
#define VPT_FORMAT_RG8 1         /* format RG, internalFormat RG8, type UNSIGNED_BYTE: two interleaved channels, texture(uVolume, p).rg
                                  * has both and the transfer function is looked up in 2-D (MIPRenderer.glsl:45-49) */
#define VPT_FORMAT_RG32F 3       /* format RG, internalFormat RG32F (or RG16F widened on upload), type FLOAT / HALF_FLOAT: two interleaved float
                                  * channels (Volume.js:58-60 allocates whatever internalFormat the manifest names); blocks are uploaded as
                                  * interleaved float32 pairs */

/* Buffers readable through vpt_renderer_read (SingleBuffer.js / DoubleBuffer.js attachments) */
#define VPT_BUFFER_RENDER 0      /* RGBA16F, 8 B/pixel  (AbstractRenderer.js:142-155, getTexture() :114-116) */
#define VPT_BUFFER_FRAME  1      /* MIP R8 | EAM, LAO RGBA8 | MCS RGBA32F | ISO RGBA16F | Depth R32F  (_getFrameBufferSpec); none for MCM, DOS */
#define VPT_BUFFER_ACCUM  2      /* same formats; DOS: its RGBA32F colour attachment  (_getAccumulationBufferSpec, read side) */
#define VPT_BUFFER_MCM_POSITION      3   /* RGBA32F [pos.xyz, 0]            (MCMRenderer.js:214-263) */
#define VPT_BUFFER_MCM_DIRECTION     4   /* RGBA32F [dir.xyz, bounces]      */
#define VPT_BUFFER_MCM_TRANSMITTANCE 5   /* RGBA32F [transmittance.rgb, 0]  */
#define VPT_BUFFER_MCM_RADIANCE      6   /* RGBA32F [radiance.rgb, samples] */
#define VPT_BUFFER_DOS_OCCLUSION     7   /* R32F, the DOS renderer's second accumulation attachment (DOSRenderer.js:295-305);
                                            its colour attachment (RGBA32F) is VPT_BUFFER_ACCUM */

typedef struct vpt_context  vpt_context;
typedef struct vpt_volume   vpt_volume;
typedef struct vpt_renderer vpt_renderer;

/*
 * The uniforms the reference uploads before each draw (gl.uniform* calls in
 * MIPRenderer.js:82-97, EAMRenderer.js:99-116, MCSRenderer.js:88-117, MCMRenderer.js:91-106,155-175).
 * Values the reference draws with Math.random() (uOffset, uRandSeed, the MCS light direction)
 * are explicit inputs so a run is reproducible ("fixed-seed mode").
 * uInverseResolution is derived by the library from the renderer's size (fl32(1/W), fl32(1/H)).
 */
typedef struct vpt_uniforms {
    float    mvp_inverse[16];   /* uMvpInverseMatrix, column-major (gl-matrix layout) */
    float    rand_seed;         /* uRandSeed */
    float    offset;            /* uOffset            (MIP, EAM) */
    float    step_size;         /* uStepSize = 1/steps (MIP) or 1/slices (EAM, Depth); ISO: fl(1/float(uSteps)), the shader's own division */
    float    extinction;        /* uExtinction        (EAM, MCS, MCM, Depth) */
    float    anisotropy;        /* uAnisotropy        (MCM) */
    uint32_t max_bounces;       /* uMaxBounces        (MCM) */
    uint32_t steps;             /* uSteps             (MCM, ISO) */
    float    light_direction[3];/* uScatteringDirection (MCS) | uLight (ISO, model space, normalised: ISORenderer.js:152-166) */
    float    mix;               /* uMix (EAM, Depth: 1/frameNumber) | uInvFrameNumber (MCS) */
    float    blur;              /* uBlur              (MCM, always 0 in the reference: MCMRenderer.js:93,157) */
    float    isovalue;          /* uIsovalue          (ISO, ISORenderer.js:100) */
    float    gradient_step;     /* uGradientStep      (ISO, 0.005: ISORenderer.js:168) */
    float    threshold;         /* uThreshold         (Depth, DepthRenderer.js:112) */
    float    reserved;          /* keeps the block at 128 bytes */
} vpt_uniforms;

/* ---- context: replaces the WebGL2RenderingContext the reference passes as `gl` (RenderingContext.js:66-106) */
VPT_API int vpt_device_count(int *count);
VPT_API int vpt_context_create(int device_ordinal, vpt_context **out);
/* same, but work is enqueued on a caller-owned HIP stream (e.g. torch.cuda.current_stream().cuda_stream) so that
 * it orders with the caller's other work (RCCL collectives) without host synchronisation; the stream is not destroyed */
VPT_API int vpt_context_create_on_stream(int device_ordinal, void *hip_stream, vpt_context **out);
VPT_API int vpt_context_destroy(vpt_context *ctx);
VPT_API int vpt_context_synchronize(vpt_context *ctx);
VPT_API const char *vpt_last_error(void);
VPT_API const char *vpt_version(void);

/* ---- volume: Volume.js:31-78 (texStorage3D + texSubImage3D per placement), :115-125 (setFilter), :17-22 (destroy) */
VPT_API int vpt_volume_create(vpt_context *ctx, int width, int height, int depth, int format, vpt_volume **out);
VPT_API int vpt_volume_upload_block(vpt_volume *vol, int x, int y, int z, int width, int height, int depth,
                                    const void *host_data, size_t nbytes);
/* same, source already resident in HBM on this context's device (skips PCIe) */
VPT_API int vpt_volume_upload_block_device(vpt_volume *vol, int x, int y, int z, int width, int height, int depth,
                                           const void *device_data, size_t nbytes);
/* builds the bricked Z-order layout from the uploaded blocks; called implicitly by the first pass that needs it */
VPT_API int vpt_volume_finalize(vpt_volume *vol);
VPT_API int vpt_volume_set_filter(vpt_volume *vol, int filter);
VPT_API int vpt_volume_destroy(vpt_volume *vol);
/* force the > 4 GiB addressing variant (brick-code tables + computed in-brick offset; automatic when the bricked layout
 * exceeds 4 GiB); lets the wide kernel variants
 * be checked against the oracle on small volumes */
VPT_API int vpt_volume_set_wide_tables(vpt_volume *vol, int wide);
/* bytes of the bricked layout in HBM (for reporting) */
VPT_API int vpt_volume_bricked_bytes(vpt_volume *vol, uint64_t *nbytes);

/* ---- renderer: AbstractRenderer.js:17-116 and the four subclasses */
/* new R(gl, volume, camera, environmentTexture, {resolution}) — AbstractRenderer.js:17-49; width != height is the
 * documented extension (uInverseResolution = (1/W, 1/H)).  Buffers are allocated as in _rebuildBuffers :78-92. */
VPT_API int vpt_renderer_create(vpt_context *ctx, int kind, int width, int height, vpt_renderer **out);
/* image-plane sharding (no reference counterpart): this renderer owns the row blocks b with b % world == rank,
 * rows_per_block rows each; all buffers then hold only the owned rows (vpt_renderer_local_rows). */
VPT_API int vpt_renderer_set_shard(vpt_renderer *r, int rank, int world, int rows_per_block);
VPT_API int vpt_renderer_local_rows(vpt_renderer *r, int *rows);
/* global row index of local row l (identity when unsharded); -1 for padding rows */
VPT_API int vpt_renderer_global_row(vpt_renderer *r, int local_row, int *global_row);
VPT_API int vpt_renderer_destroy(vpt_renderer *r);                                   /* destroy() :51-58 */
VPT_API int vpt_renderer_set_volume(vpt_renderer *r, vpt_volume *vol);               /* setVolume() :94-97 (caller resets) */
VPT_API int vpt_renderer_set_transfer_function(vpt_renderer *r, const uint8_t *rgba, int width, int height); /* :99-104 */
VPT_API int vpt_renderer_set_environment(vpt_renderer *r, const uint8_t *rgba, int width, int height);       /* RenderingContext.js:90-101,136-141 */
/* The transfer-function widget's canvas as data (src/js/ui/TransferFunction/TransferFunction.js:110-121, src/glsl/TransferFunction.glsl:32-35):
 * `count` Gaussian bumps drawn in order into a cleared width x height RGBA8 target with gl.blendFunc(ONE, ONE_MINUS_SRC_ALPHA) —
 * src = color * exp(-|(position - uv) / size|^2) at the pixel centre uv, dst = src + dst * (1 - src.a), stored as UNORM8 after every
 * bump like the canvas's own 8-bit buffer — then handed over as texImage2D(canvas) hands it to setTransferFunction
 * (AbstractRenderer.js:99-104): texel row 0 = the canvas's TOP row (uv.y = 1 - 0.5 / height), and with `unpremultiply` != 0 the colour
 * divided by alpha again (what a browser does for a premultiplied WebGL canvas when UNPACK_PREMULTIPLY_ALPHA_WEBGL is false, the
 * reference's case; rounding to nearest).  Contract arithmetic (IEEE sqrt and division, the library's exp): one result on every host.
 * Parity unpinned: a browser's exp in `precision mediump` and its un-premultiplication are implementation-defined.
 * rgba_out: width * height * 4 bytes of HOST memory, the layout vpt_renderer_set_transfer_function takes. */
typedef struct vpt_tf_bump {
    float x, y;             /* position.x, position.y in [0, 1]^2 (y = 1: the top of the widget = the first texel row) */
    float sx, sy;           /* size.x, size.y */
    float r, g, b, a;       /* color */
} vpt_tf_bump;
VPT_API int vpt_transfer_function_rasterize(vpt_context *ctx, const vpt_tf_bump *bumps, int count, int width, int height,
                                            int unpremultiply, uint8_t *rgba_out);
VPT_API int vpt_renderer_resize(vpt_renderer *r, int width, int height);             /* setResolution() :106-112 (caller resets) */

/* the four hooks; integrate and reset include the DoubleBuffer swap (AbstractRenderer.js:60-76) */
VPT_API int vpt_renderer_reset(vpt_renderer *r, const vpt_uniforms *u);              /* reset(): _resetFrame + swap */
VPT_API int vpt_renderer_generate(vpt_renderer *r, const vpt_uniforms *u);           /* _generateFrame */
VPT_API int vpt_renderer_integrate(vpt_renderer *r, const vpt_uniforms *u);          /* _integrateFrame + swap */
VPT_API int vpt_renderer_render_frame(vpt_renderer *r, const vpt_uniforms *u);       /* _renderFrame */
/* render(): generate -> integrate -> swap -> renderFrame in ONE launch (same results as the three hooks) */
VPT_API int vpt_renderer_render(vpt_renderer *r, const vpt_uniforms *u);

/* Frame sequences: `count` consecutive render() passes enqueued by ONE call.  `base` holds the uniforms the frames
 * share; frame_vars holds count x 8 floats {rand_seed, offset, mix, 0, light.x, light.y, light.z, 0} — the uniforms
 * that change per frame.  mode VPT_PLAY_EAGER: count launches with the per-frame uniforms in their arguments.
 * VPT_PLAY_GRAPH: the launch sequence is captured once into a hipGraph and replayed (uniforms from a device table read
 * through a device-side frame counter) — where that is the faster form: a captured sequence is whole-image kernels on one stream, so a
 * renderer whose passes run as tile lists and / or on several streams (the defaults) plays the sequence eagerly instead; set
 * VPT_OPTION_SPLIT_STREAMS 1 and VPT_OPTION_TILE_CLASSES 0 to get the graph.  VPT_PLAY_FUSED (MIP, EAM, MCS, MCM, ISO, Depth; MCM below 8 passes: played eagerly, the faster form there): ONE launch runs all `count` passes
 * of a pixel back to back with the photon state (MCM) or the accumulator (MIP, EAM, MCS, ISO, Depth) in registers — no
 * round trip through HBM between passes.  VPT_PLAY_FRAMES (MCM): VPT_PLAY_FUSED that still WRITES EVERY FRAME — pass f of the call
 * goes to slot f of the renderer's frame ring ([VPT_FRAME_SLOTS][local rows][width] RGBA16F, allocated on first use; count <=
 * VPT_FRAME_SLOTS): the images `count` render() calls would have shown one after the other, for a caller that displays or
 * records them later than it asks for them.  In every mode the buffers afterwards are identical to `count` calls of
 * vpt_renderer_render (the render buffer holds the last frame). */
#define VPT_PLAY_EAGER 0
#define VPT_PLAY_GRAPH 1
#define VPT_PLAY_FUSED 2
#define VPT_PLAY_FRAMES 3
#define VPT_FRAME_SLOTS 16
VPT_API int vpt_renderer_play(vpt_renderer *r, const vpt_uniforms *base, const float *frame_vars, int count, int mode);
/* frame `slot` of the last VPT_PLAY_FRAMES call ([local rows][width][4] RGBA16F), and the ring's device address (extensions: the
 * reference renders one frame per animation tick, AbstractRenderer.js:60-70, and has no frame sequences) */
/* (extension) `count` eager render() passes by one call, frame i written to caller-owned device memory at first_target + i * stride_bytes
 * (e.g. the slots of a bucket one collective will move); frame_vars as for vpt_renderer_play.  The same frames as `count` times
 * { vpt_renderer_set_render_target; vpt_renderer_render }, but the bucket's passes run on every stream of a split pass (MCM: the tile
 * classes' two kernels) and are joined ONCE, before the call returns: whatever the caller enqueues on the context's stream next sees all
 * `count` frames.  A later call with the same slots may skip texels that cannot have changed (see vpt_renderer_set_render_target). */
VPT_API int vpt_renderer_play_into(vpt_renderer *r, const vpt_uniforms *base, const float *frame_vars, int count, void *first_target, size_t stride_bytes);
VPT_API int vpt_renderer_read_frame_slot(vpt_renderer *r, int slot, void *host_dst, size_t nbytes);
VPT_API int vpt_renderer_frame_ring_device(vpt_renderer *r, void **device_ptr, size_t *slot_bytes);

/* read-back (the reference never reads back; it hands getTexture() to the tone mapper). Row-major, local rows. */
VPT_API int vpt_renderer_read(vpt_renderer *r, int buffer, void *host_dst, size_t nbytes);
/* device pointer of the row-major RGBA16F render buffer (for the RCCL frame gather) */
VPT_API int vpt_renderer_render_buffer_device(vpt_renderer *r, void **device_ptr, size_t *nbytes);
/* redirect _renderFrame output into caller-owned device memory (>= width*local_rows*8 bytes), e.g. the send buffer of
 * the frame gather; NULL restores the renderer's own render buffer (gl.bindFramebuffer analogue, SingleBuffer.js:28-32).  The first pass into a
 * target after this call writes every texel of it; later passes into the SAME target may skip texels whose value cannot have changed since
 * (the accumulating renderers' tiles that miss the volume): a caller that writes into the memory itself calls this function again */
VPT_API int vpt_renderer_set_render_target(vpt_renderer *r, void *device_ptr, size_t nbytes);
/* implementation switches (results are identical either way).  VPT_OPTION_MCS_PERSISTENT (default 0): run the MCS
 * generate pass as persistent waves with __ballot/__popcll active-ray compaction instead of one thread per pixel
 * (measured slower on MI355X at extinction 1..200 for 512^3 @ 1080p, DESIGN.md section 5: an option, never the default).  One-channel
 * byte volumes; other formats run the default kernels whatever the option says. */
#define VPT_OPTION_MCS_PERSISTENT 0
/* VPT_OPTION_MCM_PERSISTENT (default 0): 1 = run the MCM integrate pass as persistent waves walking several 8x8-pixel
 * segments (LDS tables staged once per workgroup), 2 = the same with the next segment's photon state prefetched under
 * the current segment's events.  Measured 4-8 % slower than one workgroup per tile (register pressure). */
#define VPT_OPTION_MCM_PERSISTENT 1
/* VPT_OPTION_FAST_MATH (default 0; MCM renderer): 1 = run the integrate pass with the arithmetic a GPU driver gives GLSL
 * (v_rcp_f32 / v_rsq_f32 / v_sqrt_f32 / v_log_f32 / v_sin_f32 / v_cos_f32, and algebraically equal shorter forms) instead of
 * the software routines of the bit-exact contract.  Same integer PCG stream, same decisions except within rounding error;
 * results are NOT bit-identical to the contract oracle: checked by first-event agreement and converged-image statistics
 * (tolerance in DESIGN.md section 3).  The persistent-wave option has no fast variant (it takes precedence when set). */
#define VPT_OPTION_FAST_MATH 2
/* VPT_OPTION_BOUNDARY_ATLAS (default 1; MCM renderer, LINEAR filter, one-channel volumes): the sample of an event whose position
 * lies outside the cube — taken at the clamped position and discarded by the shader (MCMRenderer.glsl:132-142) — is fetched from
 * the volume's boundary atlas (the six outer voxel planes, one dword per 2 x 2 footprint) instead of the bricks: the same
 * value bit for bit (a clamped cell has weight 0 along the clamped axis), one aligned 4-byte gather instead of two unaligned
 * 8-byte ones.  0 = always the bricks. */
#define VPT_OPTION_BOUNDARY_ATLAS 3
/* VPT_OPTION_SPLIT_STREAMS (every renderer but DOS; default since round 4: MCM 2 — the HIT | MISS kernels of the tile classes —, MIP / EAM /
 * Depth 3, ISO / MCS / LAO 2: the measured best forms at 1080p, so that a caller who sets nothing gets them (the default follows the launch size: a frame of a few hundred tiles
 * stays on one or two streams; a count set through the option is taken as it is); VPT_DEFAULT_SPLIT=1 in the environment restores 1
 * everywhere): K in 2 .. VPT_MAX_SPLIT = a pass is launched as K tile-row ranges (or K parts of a tile list), all but
 * the first on private side streams (created by the first pass that uses them).  A pixel's pass depends on its own previous pass only, so consecutive passes of the
 * ranges never wait for each other and the launch gap, ramp and tail of one range overlap the body of the others (HIP
 * streams in place of one longer launch).  Every other entry point (reads, reset, tone mapper, gather, synchronize ...) first
 * joins the side streams into the context's stream, so callers see the usual in-order semantics; while the render buffer is
 * redirected into caller memory (vpt_renderer_set_render_target, vpt_gather_*) passes stay on the context's stream, because the
 * caller's own work on that stream reads the frame; a frame sequence captured into a hipGraph (VPT_PLAY_GRAPH) stays on the capturing
 * stream.  Results identical. */
#define VPT_MAX_SPLIT 4
#define VPT_OPTION_SPLIT_STREAMS 4
/* (5: retired in round 4 — vpt_renderer_play_into* split their passes and join once per call by themselves) */
/* VPT_OPTION_TILE_CLASSES (default 1; MCM renderer, LINEAR filter, one-channel volumes with the boundary atlas; extension, no
 * reference counterpart): vpt_renderer_reset() sorts the 16x16 tiles into those none of whose camera rays (jitter included) can meet
 * the unit cube ("MISS") and the rest; while the passes keep the reset's uMvpInverseMatrix and blur == 0, a MISS tile's photon is
 * re-emitted and leaves again at every event (MCMRenderer.glsl:135-141 after resetPhoton :70-78), so its passes run a straight-line
 * kernel on 32 of the 56 state bytes per pixel.  Every buffer a caller can read is identical either way.
 * Value 2 (MCM; a measurement aid): the two class kernels also where a pass must stay on ONE stream, one after the other — each alone on the chip
 * (slower than the general kernel there, which is why 1 does not do it).
 * VPT_OPTION_VERIFY_TILE_CLASSES (default 0): the MISS-tile kernel also counts events that were inside the cube after all
 * (vpt_renderer_tile_classes' `violations`: must stay 0). */
#define VPT_OPTION_TILE_CLASSES 6
#define VPT_OPTION_VERIFY_TILE_CLASSES 7
/* (8: retired in round 4 — the library picks the HIT-tile kernel's form by the number of HIT tiles; VPT_HIT_KERNEL_FORM=1|2 in the environment
 * at renderer creation forces one for A/B measurements and tests) */
/* VPT_OPTION_BUCKET_KERNEL (default 0; MCM renderer; extension): vpt_renderer_play_into() runs up to 16 frames of its bucket by ONE launch
 * per tile class — the photon state stays in registers from the first frame to the last, launch gap, table staging and state traffic are
 * paid once per bucket instead of once per frame (what a rank's small share of a sharded frame mostly pays for) — where the tile classes
 * are in force (VPT_OPTION_TILE_CLASSES with VPT_OPTION_SPLIT_STREAMS >= 2: the defaults); elsewhere frame by
 * frame as without the option.  Every frame is rendered and written to its slot either way; results identical. */
#define VPT_OPTION_BUCKET_KERNEL 9
/* VPT_OPTION_COLUMN_RECORDS (default 2; MCM renderer, LINEAR filter, one-channel byte volumes; extension): 1 = the in-cube samples of the MCM
 * events are fetched from the volume's COLUMN RECORDS — a third layout of the same texels, built on the first MCM pass that wants it: one
 * dword per voxel holding the 2 x 2 x-y footprint of its cell, the records of a voxel column contiguous, so that the eight taps of a
 * trilinear sample are ONE dword-aligned 8-byte gather from one 64-byte sector instead of two byte-aligned windows of a 128-byte brick
 * line.  4 bytes per voxel of HBM.  Same taps, same lerps: results identical.  0 = the apron bricks, as the other renderers; 2 = records
 * where the bricks are beyond the Infinity Cache (> 512 MiB: measured 2-4 % faster there, 4-16 % slower on volumes that fit it). */
#define VPT_OPTION_COLUMN_RECORDS 10
VPT_API int vpt_renderer_set_option(vpt_renderer *r, int option, int value);
/* (extension) how many buckets of frames vpt_renderer_play_into has run through the bucket kernels so far (VPT_OPTION_BUCKET_KERNEL) */
VPT_API int vpt_renderer_bucket_launches(vpt_renderer *r, uint64_t *launches);
/* (extension) tiles of each class under the last reset's matrix (all HIT when no classification is in force) and the
 * VPT_OPTION_VERIFY_TILE_CLASSES counter; any pointer may be null */
VPT_API int vpt_renderer_tile_classes(vpt_renderer *r, int *hit_tiles, int *miss_tiles, uint64_t *violations);
/* (extension, host only: touches no GPU) the classification itself for an image of width x height whose rows are sharded
 * (rank of world, rows_per_block): classes[ty * tiles_x + tx] = 1 where tile (tx, ty) of the rank's LOCAL rows is a MISS tile.
 * `classes` may be null to query the tile grid. */
VPT_API int vpt_classify_tiles(int width, int height, int rank, int world, int rows_per_block, const float *mvp_inverse,
                               uint8_t *classes, size_t nclasses, int *tiles_x, int *tiles_y);
/* (extension, no reference counterpart) Joins the side streams of a split pass (VPT_OPTION_SPLIT_STREAMS) into the context's stream: everything enqueued on that
 * stream afterwards sees every range of the passes enqueued so far.  A no-op when nothing is pending.  Every other entry point
 * that touches the renderer's buffers does this by itself. */
VPT_API int vpt_renderer_join(vpt_renderer *r);
/* the LAO renderer's own uniforms (gl.uniform* calls of LAORenderer.js:159-169; uStepSize and uExtinction travel in
 * vpt_uniforms); defaults are the reference's property defaults (LAORenderer.js:17-108) */
struct vpt_lao_params {
    int   local_ambient_occlusion;  /* uLocalAmbientOcclusion (true) */
    float lao_weight;               /* uLAOWeight (0.69) */
    int   num_lao_samples;          /* uNumLAOSamples (1) */
    float lao_step_size;            /* uLAOStepSize (0.05) */
    int   soft_shadows;             /* uSoftShadows (true) */
    float shadows_weight;           /* uShadowsWeight (0.54) */
    int   num_shadow_samples;       /* uNumShadowSamples (10) */
    float light_radius;             /* uLightRadious (0.19) */
    float light_coefficient;        /* uLightCoeficient (1.0) */
    float light_position[3];        /* uLightPosition (2, 12, 3) */
};
VPT_API int vpt_renderer_set_lao_params(vpt_renderer *r, const struct vpt_lao_params *p);
/* The DOS renderer (directional occlusion shading) sweeps the volume front to back, one view-aligned slice per
 * full-screen pass, each pass reading its neighbours' occlusion from the pass before (DOSRenderer.js:199-259):
 *   vpt_renderer_set_occlusion_samples  the RG32F texel row built by generateOcclusionSamples() (DOSRenderer.js:103-140),
 *                                       count x (x, y);
 *   vpt_renderer_integrate_slices       _integrateFrame(): `count` passes; pass s takes uOcclusionScale = slices[3s], slices[3s+1]
 *                                       and uDepth = slices[3s+2] (the `correction` vector of :243-252), uSliceDistance =
 *                                       u->step_size, uExtinction = u->extinction, uMvpInverseMatrix = u->mvp_inverse.
 * reset / render_frame / read work as for the other renderers (generate is the reference's empty hook); integrate, the
 * single-launch vpt_renderer_render, frame sequences and sharding are refused for this renderer. */
VPT_API int vpt_renderer_set_occlusion_samples(vpt_renderer *r, const float *xy, int count);
VPT_API int vpt_renderer_integrate_slices(vpt_renderer *r, const struct vpt_uniforms *u, const float *slices, int count);
/* volume samples executed since creation / last clear (SURVEY §8d metric) */
VPT_API int vpt_renderer_sample_count(vpt_renderer *r, uint64_t *count);
VPT_API int vpt_renderer_clear_sample_count(vpt_renderer *r);

/* per-launch HIP-event timing of the dominant kernel (generate for MIP/EAM/MCS, integrate for MCM) on the
 * context's stream; enable, run, then query: sum of durations in ms and number of timed launches.
 * enabled = 1 times every launch, enabled = n > 1 every n-th launch (the two event packets cost ~7 us per launch). */
VPT_API int vpt_renderer_set_profiling(vpt_renderer *r, int enabled);
VPT_API int vpt_renderer_profile(vpt_renderer *r, double *total_ms, uint32_t *launches);
/* the same for the first launch the sampled passes put on a side stream (a pass split by VPT_OPTION_SPLIT_STREAMS; with tile classes in
 * force that is the MISS-tile kernel, k_mcm_miss, while vpt_renderer_profile covers the HIT-tile kernel on the context's stream) */
VPT_API int vpt_renderer_profile_side(vpt_renderer *r, double *total_ms, uint32_t *launches);

/* ---- tone mappers (the classes of src/js/tonemappers/ and the shaders of src/glsl/tonemappers/).  One per-pixel pass:
 * RGBA16F render buffer of a renderer (AbstractToneMapper.setTexture, AbstractToneMapper.js:34-36) -> RGBA8
 * (AbstractToneMapper.js:66-79).  Source and target have the same resolution, as RenderingContext.js:184-187,223-227
 * keeps them, so the source is read texel for texel; a sharded renderer's source is its local rows. */
#define VPT_TONEMAPPER_ARTISTIC   0   /* ToneMapperFactory.js:13-24, in its order */
#define VPT_TONEMAPPER_RANGE      1
#define VPT_TONEMAPPER_REINHARD   2
#define VPT_TONEMAPPER_REINHARD2  3
#define VPT_TONEMAPPER_UNCHARTED2 4
#define VPT_TONEMAPPER_FILMIC     5
#define VPT_TONEMAPPER_UNREAL     6
#define VPT_TONEMAPPER_ACES       7
#define VPT_TONEMAPPER_LOTTES     8
#define VPT_TONEMAPPER_UCHIMURA   9
/* the gl.uniform1f values of the _renderFrame() hooks; a mapper reads only its own fields */
struct vpt_tonemap_params {
    float low, mid, high, saturation;   /* Artistic: ArtisticToneMapper.js:75-79 (defaults 0, 0.5, 1, 1) */
    float min, max;                     /* Range: RangeToneMapper.js:58-59 (0, 1) */
    float exposure;                     /* Reinhard ... Uchimura: ReinhardToneMapper.js:53 (1) */
    float gamma;                        /* all (2.2) */
};
typedef struct vpt_tonemapper vpt_tonemapper;
VPT_API int vpt_tonemapper_create(vpt_context *ctx, int kind, int width, int height, vpt_tonemapper **out);
VPT_API int vpt_tonemapper_destroy(vpt_tonemapper *t);
VPT_API int vpt_tonemapper_resize(vpt_tonemapper *t, int width, int height);           /* setResolution, :57-62 */
/* setTexture: the source is the renderer's render buffer (read at render time; NULL = the 1x1 white texture of
 * RenderingContext.js:176-181, i.e. every texel (1,1,1,1)) */
VPT_API int vpt_tonemapper_set_source(vpt_tonemapper *t, vpt_renderer *r);
/* setTexture with an image: [rows][width] RGBA16F copied from the host into a texture owned by the tone mapper */
VPT_API int vpt_tonemapper_set_source_image(vpt_tonemapper *t, const void *rgba16f, int width, int rows);
/* render(): one pass into the RGBA8 target */
VPT_API int vpt_tonemapper_render(vpt_tonemapper *t, const struct vpt_tonemap_params *p);
/* getTexture(): [rows][width] RGBA8 -> host (blocks); rows = the source's rows (the resolution's height when unsharded) */
VPT_API int vpt_tonemapper_read(vpt_tonemapper *t, void *dst, size_t nbytes);
/* (extension) vpt_renderer_play_into with the frames AS THE TONE MAPPER SHOWS THEM: frame i of `count` eager render() passes, tone-mapped to
 * RGBA8 by `t`, into caller-owned device memory at first_target + i * stride_bytes (stride >= width * local rows * 4) — half the bytes
 * of the RGBA16F frames for the collective that moves a bucket of them.  `t` must be armed on `r`: bound with vpt_tonemapper_set_source,
 * VPT_TONEMAPPER_OPTION_FUSE on, vpt_tonemapper_render called once with the parameters to show (its table form).  MCM with the tile
 * classes in force runs the bucket kernels (their frame store looks the texel up in the tone mapper's table; the renderer's own render
 * buffer is not written); otherwise frame by frame through the fused pass and a device copy of the tone mapper's output.  Texels
 * identical to vpt_tonemapper_render on the same frames.  Joins once, before it returns, as vpt_renderer_play_into does. */
VPT_API int vpt_renderer_play_into_display(vpt_renderer *r, vpt_tonemapper *t, const struct vpt_uniforms *base, const float *frame_vars, int count,
                                           void *first_target, size_t stride_bytes);
VPT_API int vpt_tonemapper_rows(vpt_tonemapper *t, int *rows);
VPT_API int vpt_tonemapper_output_device(vpt_tonemapper *t, void **ptr, size_t *nbytes);
/* Range and the eight curve mappers can run through a 65 536-entry byte table (every output byte depends on one
 * half-precision input): bit-identical output, byte gathers instead of log/exp/divisions.  AUTO (default) uses it for
 * images of >= 262 144 pixels or when the table of the current parameters already exists; Artistic does at
 * saturation 1 only (otherwise its channels are coupled). */
#define VPT_TONEMAPPER_OPTION_TABLE  0
#define VPT_TONEMAPPER_TABLE_NEVER   0
#define VPT_TONEMAPPER_TABLE_ALWAYS  1
#define VPT_TONEMAPPER_TABLE_AUTO    2
/* VPT_TONEMAPPER_OPTION_FUSE (default 1; extension): once vpt_tonemapper_render() has run in its table form on a bound renderer, that renderer's
 * fused render() passes write the tone-mapped RGBA8 texel (same table, same lookups: bit-identical) next to every RGBA16F texel they store, and
 * the following vpt_tonemapper_render() calls with unchanged parameters find their output ready and launch nothing — RenderingContext.render()
 * (RenderingContext.js:196-197: renderer.render(); toneMapper.render()) then costs one pass instead of two.  Any pass that writes the render
 * buffer another way (the separate hooks, frame sequences, a caller's render target), new parameters or a resize fall back to the separate pass. */
#define VPT_TONEMAPPER_OPTION_FUSE   1
VPT_API int vpt_tonemapper_set_option(vpt_tonemapper *t, int option, int value);

/* ---- multi-GPU frame gather (no reference counterpart).  One process per GPU; the image plane is sharded
 * with vpt_renderer_set_shard and every frame is all-gathered over RCCL/xGMI.  The pipeline lives below the C ABI so
 * that a frame costs the host two enqueues: kernel k+1 (context stream) overlaps the all_gather of frame k (own
 * communication stream), ordered by HIP events; send/receive buffers are double-buffered. */
typedef struct vpt_gather vpt_gather;
/* RCCL bootstrap: ONE rank obtains the 128-byte id and shares it with the others out of band (e.g. torch.distributed) */
VPT_API int vpt_gather_unique_id(void *id128);
/* collective: every rank of `world` calls it with the same id; the renderer must already be sharded (rank, world) */
VPT_API int vpt_gather_create(vpt_renderer *r, const void *id128, int rank, int world, vpt_gather **out);
VPT_API int vpt_gather_destroy(vpt_gather *g);
/* who receives the frames: root = -1 (default) every rank (RCCL all_gather); root = k only rank k, the display rank
 * (grouped ncclSend/ncclRecv: 1/world of the all_gather traffic).  Collective: every rank passes the same root. */
VPT_API int vpt_gather_set_root(vpt_gather *g, int root);
/* render() of the renderer into the next send buffer + asynchronous gather of it */
VPT_API int vpt_gather_render(vpt_gather *g, const vpt_uniforms *u);
/* `count` passes of the pipeline by one call (frame_vars as for vpt_renderer_play).  VPT_PLAY_EAGER: count frames, each
 * rendered and gathered; VPT_PLAY_FUSED (MCM): the count passes run in one launch and the resulting frame is gathered
 * once.  No graph mode — a captured graph holding RCCL collectives measured slower and unstable on this stack
 * (DESIGN.md section 7). */
VPT_API int vpt_gather_play(vpt_gather *g, const vpt_uniforms *base, const float *frame_vars, int count, int mode);
/* blocks until every enqueued frame has been rendered and gathered */
VPT_API int vpt_gather_synchronize(vpt_gather *g);
/* the most recently gathered frame, rows put back in order: [height][width] RGBA16F -> host (blocks) */
VPT_API int vpt_gather_read_frame(vpt_gather *g, void *host_dst, size_t nbytes);

/* ---- test probes: evaluate device-side building blocks on the GPU (tests/ compares with the oracle) */
#define VPT_PROBE_LOG     0   /* out[i] = log(in[i]) */
#define VPT_PROBE_SIN     1
#define VPT_PROBE_COS     2
#define VPT_PROBE_ASIN    3
#define VPT_PROBE_ATAN2   4   /* in = pairs (y, x) ; n outputs */
#define VPT_PROBE_PCG     5   /* bit patterns: out[i] = pcg(in[i]) */
#define VPT_PROBE_UNIFORM 6   /* bit pattern state in -> float uniform out */
#define VPT_PROBE_F16     7   /* out[i] (low 16 bits) = half(in[i]) */
#define VPT_PROBE_RCP     8   /* software reciprocal rcp_nr */
#define VPT_PROBE_RSQRT   9   /* software reciprocal square root rsqrt_nr */
#define VPT_PROBE_MIN     10  /* in = pairs (a, b) ; n outputs */
#define VPT_PROBE_MAX     11
#define VPT_PROBE_LOG_UNIFORM 12  /* log on the range of random_uniform */
#define VPT_PROBE_RCPZ    13  /* rcp_nrz: reciprocal with 1/(+-0) = +-inf */
#define VPT_PROBE_SQRT    14  /* sqrt_nr = x * rsqrt_nr(x) */
#define VPT_PROBE_EXP     15  /* the tone mappers' exp */
#define VPT_PROBE_POW     16  /* in = pairs (x, y): exp(y * log(x)) ; n outputs */
VPT_API int vpt_probe_math(vpt_context *ctx, int which, const float *in, float *out, size_t n);
/* samples texture(uVolume, p) -> transfer function at n positions (xyz triples); out = n RGBA float4 */
VPT_API int vpt_probe_sample(vpt_renderer *r, const float *xyz, float *rgba, size_t n);
/* the same through the volume's boundary atlas for positions outside the cube (what the MCM kernels execute for out-of-cube events; positions
 * inside go through the bricks): must equal vpt_probe_sample bit for bit in every volume format */
VPT_API int vpt_probe_sample_boundary(vpt_renderer *r, const float *xyz, float *rgba, size_t n);
/* The schedule of frame number `frame` of the gather pipeline as a pure host function (no GPU, no communicator needed): ring
 * buffer, event edges, render destination, RCCL operation.  vpt_gather_render / _play execute exactly this plan; exported so
 * that the multi-rank schedule can be checked without more than one GPU (tests/test_gather_schedule.py). */
#define VPT_GATHER_OP_NONE      0   /* one rank, rooted: nothing to exchange */
#define VPT_GATHER_OP_ALLGATHER 1   /* root -1: ncclAllGather(send[buffer] -> recv[buffer]) */
#define VPT_GATHER_OP_SEND      2   /* rooted, this rank is not the root: ncclSend(send[buffer]) to `peer` */
#define VPT_GATHER_OP_RECV      3   /* rooted, this rank is the root: `npeers` grouped ncclRecv into recv[buffer] (vpt_gather_plan_recv) */
typedef struct vpt_gather_step {
    int      ring;             /* buffers in the ring */
    int      buffer;           /* frame % ring */
    int      parity;           /* which half of the ring: (frame / (ring / 2)) & 1 */
    int      wait_gathered;    /* compute stream waits for gathered[parity] before the kernel (the half is re-entered) */
    int      record_gathered;  /* communication stream records gathered[parity] after this frame's exchange */
    int      rendered_event;   /* rendered[k]: kernel done -> the communication stream may start */
    int      in_place;         /* the kernel renders into recv[buffer] + render_offset (root of a rooted gather) instead of send[buffer] */
    int      op;               /* VPT_GATHER_OP_* */
    int      peer;             /* SEND: destination rank */
    int      npeers;           /* RECV: number of receives */
    uint64_t render_offset;    /* byte offset of the kernel's output inside recv[buffer] when in_place */
} vpt_gather_step;
VPT_API int vpt_gather_plan(uint64_t frame, int rank, int world, int root, uint64_t send_bytes, vpt_gather_step *out);
VPT_API int vpt_gather_plan_recv(const vpt_gather_step *step, int rank, int i, uint64_t send_bytes, int *peer, uint64_t *offset);
/* the row re-assembly of a gathered frame (k_assemble_rows, used by vpt_gather_read_frame) run on host data:
 * gathered = [world][local_rows][width] RGBA16F as the ranks' send buffers arrive, out = [height][width].  Lets a
 * one-GPU box check the multi-rank assembly against an unsharded frame. */
VPT_API int vpt_probe_assemble_rows(vpt_context *ctx, const void *gathered, int width, int height, int local_rows,
                                    int world, int rows_per_block, void *out);
/* measured HBM streaming-read rate: `iterations` grid-stride 16 B/lane reads of an nbytes scratch buffer (choose it far
 * larger than the 256 MB Infinity Cache); the second denominator SURVEY section 8d asks for next to the 8 TB/s peak */
VPT_API int vpt_probe_stream_read(vpt_context *ctx, size_t nbytes, int iterations, double *gb_per_s);

#ifdef __cplusplus
}
#endif
#endif /* VPT_H */
