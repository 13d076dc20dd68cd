// vpt_render.hip — the renderer entry points of the C-ABI: the four hooks of AbstractRenderer.render() / reset()
// (AbstractRenderer.js:60-76), the single-launch render(), and frame sequences (vpt_renderer_play*).  The passes themselves are
// launched by the family units (vpt_mcm.hip, vpt_march.hip, vpt_extra.hip).
#include "vpt_internal.h"

__global__ void k_advance_frame(uint32_t *counter) { *counter = *counter + 1u; }
__global__ void k_advance_frames(uint32_t *counter, uint32_t n) { *counter = *counter + n; }
void advance_frames(vpt_renderer *r, uint32_t n) {
    hipLaunchKernelGGL(k_advance_frames, dim3(1), dim3(1), 0, r->ctx->stream, r->frame_counter, n);
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
        c.n_complete = 0;
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
            // ... and only into a destination whose skipped tiles already hold those values: one that a whole-image fused pass has written
            // since the reset (a caller's render target, the slots of a bucket, the gather's ring: each takes one whole pass first)
            bool complete = false;
            for (int i = 0; i < c.n_complete && !complete; i++) complete = c.complete[i] == (const void *)a.render;
            if (complete) {
                if (!c.valid) VPT_TRY(classes_build(r, c.mvp));
                c.list_now = c.valid && c.n_hit > 0 && c.n_miss > 0;
            }
        }
    }
    if (fused) {
        c.fused_passes++;
        if (!c.list_now && !c.poisoned) {                         // a whole-image pass under the reset's matrix: its destination is complete now
            bool known = false;
            for (int i = 0; i < c.n_complete && !known; i++) known = c.complete[i] == (const void *)a.render;
            if (!known && c.n_complete < VPT_COMPLETE_DESTS) c.complete[c.n_complete++] = (const void *)a.render;
        }
    }
    return VPT_OK;
}

int check_step(const vpt_uniforms *u) {
    // step sizes <= 0 or NaN would never advance t: the reference's spinner enforces min 1 (MIPRenderer.js:24, EAMRenderer.js:34)
    if (!(u->step_size > 0.0f)) return fail(VPT_ERR_INVALID, "step_size must be > 0");
    if (u->step_size < 1.0f / 65536.0f) return fail(VPT_ERR_INVALID, "step_size below 1/65536 (more than 65536 steps per ray)");
    return VPT_OK;
}

static int check_iso(const vpt_uniforms *u) {
    if (u->steps < 1 || u->steps > 65536) return fail(VPT_ERR_INVALID, "ISO steps %u outside [1, 65536]", u->steps);   // ISORenderer.js:20-25: min 1
    return VPT_OK;
}


// the family of a renderer kind: MIP / EAM / MCS (vpt_march.hip), ISO / Depth / LAO / DOS (vpt_extra.hip), MCM (vpt_mcm.hip)
extern "C" int vpt_renderer_reset(vpt_renderer *r, const vpt_uniforms *u) {
    if (!r) return fail(VPT_ERR_INVALID, "renderer is null");
    VPT_TRY(join_side(r));
    if (r->kind == VPT_RENDERER_MCM && !u) return fail(VPT_ERR_INVALID, "MCM reset needs uniforms (uMvpInverseMatrix, uRandSeed)");
    HIP_TRY(hipSetDevice(r->ctx->device));
    PassArgs a;
    VPT_TRY(make_args(r, u, false, &a));
    r->cls.passes = 0; r->cls.fused_passes = 0; r->cls.poisoned = false; r->cls.list_now = false; r->cls.reset_seen = true;
    if (r->kind != VPT_RENDERER_MCM) r->cls.valid = false;
    if (r->kind == VPT_RENDERER_MCM) VPT_TRY(mcm_reset(r, a, u));
    else if (is_march_kind(r->kind)) VPT_TRY(march_reset(r, a));
    else VPT_TRY(extra_reset(r, a));
    HIP_TRY(hipGetLastError());
    return VPT_OK;
}
extern "C" int vpt_renderer_generate(vpt_renderer *r, const vpt_uniforms *u) {
    if (!r) return fail(VPT_ERR_INVALID, "null argument");
    if (r->kind == VPT_RENDERER_DOS) return VPT_OK;                 // DOSRenderer.js has no _generateFrame (AbstractRenderer.js:122-124: empty)
    if (!u) return fail(VPT_ERR_INVALID, "null argument");
    if (r->kind == VPT_RENDERER_MCM) return VPT_OK;                 // MCMRenderer.js:118-119: empty (and nothing to order: the class kernels stay on their streams)
    VPT_TRY(join_side(r));
    HIP_TRY(hipSetDevice(r->ctx->device));
    if (r->kind != VPT_RENDERER_MCS) VPT_TRY(check_step(u));
    if (r->kind == VPT_RENDERER_ISO) VPT_TRY(check_iso(u));
    PassArgs a;
    VPT_TRY(make_args(r, u, true, &a));
    VPT_TRY(marcher_track(r, a, false, 0.0f));
    {
        Timed t(r, true);
        if (is_march_kind(r->kind)) VPT_TRY(march_generate(r, a));
        else VPT_TRY(extra_generate(r, a));
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
    if (r->kind == VPT_RENDERER_MCM) {
        Timed t(r, true);
        VPT_TRY(mcm_pass(r, a, false));
        r->samples_host += r->valid_pixels * (uint64_t)u->steps;   // exactly W*H*steps per pass (MCMRenderer.glsl:129-133)
    } else if (is_march_kind(r->kind)) VPT_TRY(march_integrate(r, a));
    else VPT_TRY(extra_integrate(r, a));
    HIP_TRY(hipGetLastError());
    return VPT_OK;
}
extern "C" int vpt_renderer_render_frame(vpt_renderer *r, const vpt_uniforms *u) {
    if (!r) return fail(VPT_ERR_INVALID, "renderer is null");
    if (r->kind != VPT_RENDERER_MCM) VPT_TRY(join_side(r));    // (MCM: mcm_render_frame — behind a pass of the tile classes each class's texels are written by its own stream)
    r->tm_valid = false;                                       // (the hook kernels do not tone-map: the next vpt_tonemapper_render runs its own pass)
    HIP_TRY(hipSetDevice(r->ctx->device));
    if (r->kind == VPT_RENDERER_ISO && !u) return fail(VPT_ERR_INVALID, "ISO renderFrame needs uniforms (uLight, uGradientStep)");
    PassArgs a;
    VPT_TRY(make_args(r, u, r->kind == VPT_RENDERER_ISO, &a));     // the ISO render pass samples the volume
    if (r->kind == VPT_RENDERER_MCM) VPT_TRY(mcm_render_frame(r, a));
    else if (is_march_kind(r->kind)) VPT_TRY(march_render_frame(r, a));
    else VPT_TRY(extra_render_frame(r, a));
    HIP_TRY(hipGetLastError());
    return VPT_OK;
}
// the fused render() launch of the renderer's kind (generate -> integrate -> renderFrame in one kernel)
int launch_fused(vpt_renderer *r, const PassArgs &a) {
    // (a fused pass writes the armed tone mapper's output with every texel it writes; a pass whose arguments carry no tone map — a caller's
    // render target, the gather ring, a multi-pass or captured sequence — leaves that output behind the render buffer)
    r->tm_valid = r->tm_valid && a.tm_table != nullptr && a.multi_passes <= 1 && !r->no_split;
    VPT_TRY(marcher_track(r, a, true, a.mix));
    struct ListOff { vpt_renderer *r; ~ListOff() { r->cls.list_now = false; } } list_off{ r };
    if (r->kind == VPT_RENDERER_MCM) return mcm_pass(r, a, true);
    if (is_march_kind(r->kind)) return march_fused(r, a);
    return extra_fused(r, a);
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
void play_graph_free(PlayGraph *g) {
    if (!g) return;
    if (g->exec) hipGraphExecDestroy(g->exec);
    if (g->graph) hipGraphDestroy(g->graph);
    delete g;
}
// Appends the per-frame uniforms of the next `count` frames to the device ring (through a pinned staging ring, so the
// copy is asynchronous and the host never waits) and returns the PassArgs shared by the frames.  The device frame
// counter is monotonic: a captured graph needs no per-replay patching.
#define VPT_FRAME_RING 2048
int play_args(vpt_renderer *r, const vpt_uniforms *base, int count, PassArgs *a) {
    if (count < 1 || count > VPT_FRAME_RING / 4) return fail(VPT_ERR_INVALID, "frame count %d out of range [1, %d]", count, VPT_FRAME_RING / 4);
    if (r->kind == VPT_RENDERER_MIP || r->kind == VPT_RENDERER_EAM) VPT_TRY(check_step(base));
    return make_args(r, base, true, a);
}
#define VPT_TABLE_BY_ARGS 32
struct FrameVarBlock { FrameVar v[VPT_TABLE_BY_ARGS]; };
__global__ void k_store_frame_vars(FrameVar *table, FrameVarBlock blk, uint32_t pos, uint32_t count, uint32_t mask) {
    const uint32_t t = threadIdx.x;
    if (t < count) table[(pos + t) & mask] = blk.v[t];
}
int play_upload_table(vpt_renderer *r, const float *vars, int count, PassArgs *a) {
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
    if (count <= VPT_TABLE_BY_ARGS) {
        // short sequences: the entries travel in the arguments of a one-wave kernel (a copy engine's H2D transfer in the middle of the stream
        // costs the sequence 10-20 us of idle chip; measured per play() call of 4 / 16 frames)
        FrameVarBlock blk;
        memcpy(blk.v, src, (size_t)count * sizeof(FrameVar));
        hipLaunchKernelGGL(k_store_frame_vars, dim3(1), dim3(64), 0, c->stream, r->frame_table, blk, (uint32_t)pos, (uint32_t)count, (uint32_t)(VPT_FRAME_RING - 1));
    } else {
        int first = count < VPT_FRAME_RING - pos ? count : VPT_FRAME_RING - pos;
        memcpy(r->frame_staging + pos, src, (size_t)first * sizeof(FrameVar));
        HIP_TRY(hipMemcpyAsync(r->frame_table + pos, r->frame_staging + pos, (size_t)first * sizeof(FrameVar), hipMemcpyHostToDevice, c->stream));
        if (first < count) {
            memcpy(r->frame_staging, src + first, (size_t)(count - first) * sizeof(FrameVar));
            HIP_TRY(hipMemcpyAsync(r->frame_table, r->frame_staging, (size_t)(count - first) * sizeof(FrameVar), hipMemcpyHostToDevice, c->stream));
        }
    }
    r->main_dirty = true;                             // the side streams of a split pass read the table too: they fork behind this upload
    a->frame_base = (uint32_t)r->frames_played;       // == the device counter when the sequence starts (both advance by `count` per sequence)
    r->frames_played += (uint64_t)count;
    a->frame_table = r->frame_table;
    a->frame_counter = r->frame_counter;
    a->frame_mask = VPT_FRAME_RING - 1;
    return VPT_OK;
}
static bool play_key_equal(const PassArgs &x, const PassArgs &y) { return memcmp(&x, &y, sizeof(PassArgs)) == 0; }

extern "C" int vpt_renderer_play(vpt_renderer *r, const vpt_uniforms *base, const float *frame_vars, int count, int use_graph) {
    if (!r || !base || !frame_vars) return fail(VPT_ERR_INVALID, "null argument");
    if (r->kind == VPT_RENDERER_DOS) return fail(VPT_ERR_UNSUPPORTED, "frame sequences are not defined for the DOS renderer: drive it slice by slice");
    vpt_context *c = r->ctx;
    HIP_TRY(hipSetDevice(c->device));
    PassArgs a;
    VPT_TRY(play_args(r, base, count, &a));
    // A captured sequence lives on the capturing stream alone and freezes its grids: whole-image kernels on one stream.  Where the eager
    // passes run as tile lists and / or on several streams (the defaults), they are the faster form — 1080p, us per frame, eager | graph:
    // MCM 94 | 123, EAM 52 | 87, MIP 48 | 77, MCS 15 | 38, ISO 45 | 72, Depth 52 | 75 — and VPT_PLAY_GRAPH asks for the faster form of the
    // same sequence, not for a hipGraph at any price: the graph is kept for renderers set to one stream without tile classes, where the
    // launches are the same and one replay saves the host count - 1 enqueues.
    if (use_graph == VPT_PLAY_GRAPH && (r->cls.enabled || (r->split > 1 && !r->target_is_callers))) use_graph = VPT_PLAY_EAGER;
    // MCM's passes-in-registers kernel is a whole-image launch on one stream: it beats the tile-class passes on two streams from 8 passes per
    // launch on (1080p, us per pass, bit-exact | fast-math: loop 93.0 | 78.7; 2 per launch 109.1 | 100.7, 4: 97.1 | 84.6, 8: 92.5 | 75.1, 16: 90.9 | 70.2)
    if (use_graph == VPT_PLAY_FUSED && r->kind == VPT_RENDERER_MCM && count < 8 && r->cls.enabled && r->split > 1 && !r->target_is_callers) use_graph = VPT_PLAY_EAGER;
    // (eager sequences and the marchers' fused passes are launched exactly as render() launches them: the streams of a split pass are not
    // joined between two calls any more than between two render() calls; a captured graph and MCM's whole-image sequence kernels join)
    if (use_graph == VPT_PLAY_GRAPH || ((use_graph == VPT_PLAY_FUSED || use_graph == VPT_PLAY_FRAMES) && r->kind == VPT_RENDERER_MCM)) VPT_TRY(join_side(r));
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
        advance_frames(r, (uint32_t)count);
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
            if (by_class) VPT_TRY(mcm_bucket(r, a, (const FrameVar *)frame_vars, count, ring, (uint32_t)((size_t)r->W * r->local_h), true, nullptr));
            else VPT_TRY(mcm_multi(r, a, (uint32_t)count, ring));
        }
        advance_frames(r, (uint32_t)count);   // keeps the graph path's counter in step
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
    HIP_TRY(hipSetDevice(r->ctx->device));
    PassArgs a;
    VPT_TRY(play_args(r, base, count, &a));
    const FrameVar *v = (const FrameVar *)frame_vars;
    r->target_is_callers = true;
    // the bucket's passes may use every stream of a split pass (MCM: the HIT | MISS kernels of the tile classes); ONE join at the end of the
    // call puts all of them in front of whatever the caller enqueues on the context's stream next (the collective that moves the bucket)
    BucketCall bucket_call{ r };
    int i0 = 0;
    // VPT_OPTION_BUCKET_KERNEL: up to VPT_BUCKET_FRAMES frames per launch of each tile class
    while (r->kind == VPT_RENDERER_MCM && r->bucket_kernel && stride_bytes % 8 == 0 && stride_bytes / 8 <= 0xffffffffull && i0 < count) {
        const int n = std::min(count - i0, VPT_BUCKET_FRAMES);
        bool ready = false;
        VPT_TRY(mcm_bucket_ready(r, a, &ready));
        if (!ready) break;
        uint2 *ring = (uint2 *)((char *)first_target + (size_t)i0 * stride_bytes);
        Timed t(r, true, (uint32_t)n);
        VPT_TRY(mcm_bucket(r, a, v + i0, n, ring, (uint32_t)(stride_bytes / 8), false, nullptr));
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
    VPT_TRY(join_side(r));
    r->warmed = true;
    if (r->kind == VPT_RENDERER_MCM) r->samples_host += r->valid_pixels * (uint64_t)base->steps * (uint64_t)count;
    return VPT_OK;
}
