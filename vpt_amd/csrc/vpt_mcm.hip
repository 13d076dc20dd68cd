// vpt_mcm.hip — the MCM renderer's passes (vpt_kernels_mcm.h) behind vpt_render.hip's entry points: the tile classes (HIT | MISS kernels
// on two streams), the bucket kernels, reset / render / materialise.  The integrate kernels themselves are compiled in vpt_mcm_hit.hip, the
// frame sequences in one launch in vpt_mcm_seq.hip (three translation units: the build is parallel).  MCMRenderer.js:85-199.
#include "vpt_mcm_select.h"
#define VPT_MCM_PLAIN_KERNELS
#include "vpt_kernels_mcm.h"

// ---- MCM passes over the tile classes ---------------------------------------------------------------------------------
#ifdef VPT_EVENT_TIMING
#define VPT_TIMING_ARG(part, hit) do { if (hit) (part).violations = timing; } while (0)
#else
#define VPT_TIMING_ARG(part, hit) do { } while (0)
#endif
#ifdef VPT_EVENT_TIMING
static unsigned long long *g_timing = nullptr;
static unsigned long long *timing_buffer(vpt_renderer *r) {
    if (!g_timing) {
        if (hipMalloc(&g_timing, (size_t)VPT_TIMING_WAVES * 16 * sizeof(unsigned long long)) != hipSuccess) { g_timing = nullptr; return nullptr; }
        hipMemsetAsync(g_timing, 0, (size_t)VPT_TIMING_WAVES * 16 * sizeof(unsigned long long), r->ctx->stream);
    }
    return g_timing;
}
#endif
static bool mcm_classes_usable(const vpt_renderer *r, const PassArgs &a) {
    return r->cls.enabled && r->cls.valid && a.blur == 0.0f && memcmp(r->cls.mvp, a.mvp_inv.m, sizeof(r->cls.mvp)) == 0;
}
// the kernel side of it: the volume's boundary atlas (what k_mcm_miss samples; every format since round 4), no persistent-wave option
static bool mcm_classes_runnable(const vpt_renderer *r, const PassArgs &a) {
    return a.vol.atlas != nullptr && !r->mcm_persistent;
}
// ... and of the bucket kernels (and of the HIT-tile kernel's early form): LINEAR one-channel byte volumes
static bool mcm_plain_volume(const vpt_renderer *r) { return (variant_of(r) & ~VPT_V_WIDE) == 0; }
template <bool FUSE> static PassKernel format_miss_kernel(int v, bool fast) {
    if (fast) FORMAT_CASES((PassKernel)k_mcm_miss<FUSE, F | VPT_V_FAST, false, true>)
    FORMAT_CASES((PassKernel)k_mcm_miss<FUSE, F, false, true>)
}
// position / transmittance of the MISS tiles, as the last pass's arithmetic would have stored them
int mcm_materialize(vpt_renderer *r) {
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
// an MCM reset with matrix u->mvp_inverse has just been enqueued
static int mcm_classify(vpt_renderer *r, const vpt_uniforms *u) {
    r->cls.valid = false; r->cls.stale = false;               // the reset rewrote every array
    if (!r->cls.enabled || u->blur != 0.0f) return VPT_OK;
    return classes_build(r, u->mvp_inverse);
}

// one MCM pass (integrate, or render() = integrate + renderFrame) as list launches: the HIT tiles through k_mcm_integrate on the
// context's stream, the MISS tiles through k_mcm_miss — with VPT_OPTION_SPLIT_STREAMS = K as K - 1 equal parts on the side streams,
// so that the latency-bound HIT tiles and the arithmetic-bound MISS tiles share the chip for the whole frame
template <bool FUSE>
static int launch_mcm_classes(vpt_renderer *r, const PassArgs &a) {
    const bool fast = r->fast_math != 0, check = r->cls.verify;
    PassKernel kh, km;
    // the HIT tiles: few enough to be resident at once at 5 waves per SIMD (a shard's share) -> the form with the early path end,
    // whose pass is one wave per SIMD walking a chain of dependent latencies; else the 7-waves form (VPT_HIT_KERNEL_FORM in the environment overrides)
    const bool plain = mcm_plain_volume(r);
    const bool early = plain && (r->hit_form == 2 || (r->hit_form == 0 && r->cls.n_hit <= 1280));
    if (plain) kh = mcm_hit_kernel(FUSE, class_variant(r, a), early);
    else kh = mcm_format_hit_kernel(FUSE, variant_of(r), (variant_of(r) & VPT_V_WIDE) != 0, fast);
    // the MISS tiles: the sample consumed after the path end (its gather flies under that arithmetic) — whole frame 80.8 -> 79.3-79.7 us
    // fast-math, 96.1 -> 92.8 bit-exact, rank 3 of 8's share 18.4 -> 17.1 bit-exact but 15.8 -> 16.9 fast-math: there the sample is
    // consumed where the shader takes it
    const bool late = !(fast && early);
    if (!plain) km = format_miss_kernel<FUSE>(variant_of(r), fast);
    else if (check) km = fast ? (late ? (PassKernel)k_mcm_miss<FUSE, VPT_V_FAST, true, true> : (PassKernel)k_mcm_miss<FUSE, VPT_V_FAST, true, false>)
                         : (PassKernel)k_mcm_miss<FUSE, 0, true, true>;
    else km = fast ? (late ? (PassKernel)k_mcm_miss<FUSE, VPT_V_FAST, false, true> : (PassKernel)k_mcm_miss<FUSE, VPT_V_FAST, false, false>)
                   : (PassKernel)k_mcm_miss<FUSE, 0, false, true>;
    const size_t lds_hit = lds_bytes(r), lds_miss = (size_t)r->tf_w * 2 * sizeof(float4);
    if (lds_hit > 160 * 1024) return fail(VPT_ERR_UNSUPPORTED, "transfer function + volume tables need %zu B of LDS (> 160 KiB)", lds_hit);
    if (lds_hit > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)kh, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_hit));
    int k = 1;
    if (r->split >= 2 && !r->no_split && (!r->target_is_callers || r->bucket_call)) k = r->split;
    if (r->side_busy && r->last_layout != 1) VPT_TRY(join_side(r));      // the tile -> stream map changes: order the streams once
    r->last_layout = 1;
    struct Part { PassKernel kernel; const uint32_t *list; int n; size_t lds; };
    Part parts[VPT_MAX_SPLIT]; int np = 0;
    // (measured, 1080p headline frame, us per frame: HIT | MISS on two streams 81.0; HIT | MISS/2 | MISS/2 82.3-83.0; HIT/2 | HIT/2 | MISS
    // 82.7-84.1; four streams 93; one stream, HIT then MISS: 102.  Capping the HIT kernel's residency (dynamic LDS) to 2 / 3 / 4 / 5
    // workgroups per CU so that MISS waves always sit beside its waves: 99 / 91 / 83.4 / 82.2 against 81.6 uncapped — DESIGN.md section 5)
#ifdef VPT_EVENT_TIMING
    unsigned long long *const timing = timing_buffer(r);
#endif
    // (a frame the volume fills — no MISS tile at all — was tried with the HIT list in k parts on the k streams: 143.9 -> 143.7 us, nothing)
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
            part.miss_verify = check ? 1u : 0u;
            VPT_TIMING_ARG(part, i < hit_parts);
            // profiling: the caller's pair (Timed) brackets both launches; the MISS-tile kernel gets the pair vpt_renderer_profile_side reads
            hipEvent_t e1 = nullptr;
            if (i >= hit_parts && i == np - 1 && r->timed_now) {
                if (r->side_events_used == r->side_events.size()) {
                    hipEvent_t a0, a1;
                    if (hipEventCreate(&a0) == hipSuccess && hipEventCreate(&a1) == hipSuccess) r->side_events.push_back({ a0, a1 });
                }
                if (r->side_events_used < r->side_events.size()) {
                    hipEventRecord(r->side_events[r->side_events_used].first, r->ctx->stream);
                    e1 = r->side_events[r->side_events_used++].second;
                }
            }
            if (i + 1 == np) launch_range(parts[i].kernel, r, dim3((unsigned)parts[i].n), dim3(VPT_BLOCK), parts[i].lds, r->ctx->stream, part, 0);
            else hipLaunchKernelGGL(parts[i].kernel, dim3((unsigned)parts[i].n), dim3(VPT_BLOCK), parts[i].lds, r->ctx->stream, part);
            if (e1) hipEventRecord(e1, r->ctx->stream);
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
            part.miss_verify = check ? 1u : 0u;
            VPT_TIMING_ARG(part, i < hit_parts);
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
int mcm_bucket_ready(vpt_renderer *r, const PassArgs &a, bool *ready) {
    bool same = false;
    VPT_TRY(mcm_before_pass(r, a, &same));
    const bool two_streams = r->split >= 2 && !r->no_split && (!r->target_is_callers || r->bucket_call);
    if (two_streams) VPT_TRY(ensure_split_streams(r));
    *ready = same && r->cls.enabled && mcm_classes_runnable(r, a) && mcm_plain_volume(r) && two_streams && !r->cls.verify;
    return VPT_OK;
}
template <bool DISPLAY> static BucketKernel bucket_hit_kernel(int v, bool early) {
    if (early) VARIANT_CASES((BucketKernel)k_mcm_bucket_hit<V, true, DISPLAY>)
    VARIANT_CASES((BucketKernel)k_mcm_bucket_hit<V, false, DISPLAY>)
}
template <bool DISPLAY>
static void bucket_kernels(int v, bool early, BucketKernel *kh, BucketKernel *km) {
    const bool fast = (v & VPT_V_FAST) != 0;
    *kh = bucket_hit_kernel<DISPLAY>(v, early);
    const bool late = !(fast && early);
    *km = fast ? (late ? (BucketKernel)k_mcm_bucket_miss<VPT_V_FAST, true, DISPLAY> : (BucketKernel)k_mcm_bucket_miss<VPT_V_FAST, false, DISPLAY>)
               : (BucketKernel)k_mcm_bucket_miss<0, true, DISPLAY>;
}
// display_table: null = RGBA16F slots; else the armed tone mapper's table — RGBA8 slots (slot_pixels counts texels either way)
int mcm_bucket(vpt_renderer *r, const PassArgs &a, const FrameVar *v, int count, void *ring, uint32_t slot_pixels, bool last_to_render_buffer,
               const uint8_t *display_table) {
    if (count < 1 || count > VPT_BUCKET_FRAMES) return fail(VPT_ERR_INVALID, "a bucket launch holds 1..%d frames", VPT_BUCKET_FRAMES);
    const bool fast = r->fast_math != 0;
    // HIT tiles few enough to be resident at once at the kernel's four waves per SIMD: the form with the early path end (launch_mcm_classes)
    const bool early = r->hit_form == 2 || (r->hit_form == 0 && r->cls.n_hit <= 1024);
    BucketKernel kh, km;
    if (display_table) bucket_kernels<true>(class_variant(r, a), early, &kh, &km);
    else bucket_kernels<false>(class_variant(r, a), early, &kh, &km);
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

// MCM passes with a matrix (or a blur) other than the reset's: the photons of MISS tiles may now enter the cube — the classes are
// void until the next reset.  Whole-image kernels need the MISS tiles' position / transmittance arrays up to date first.
int mcm_before_pass(vpt_renderer *r, const PassArgs &a, bool *same_matrix) {
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
    // vpt_renderer_play_into*, a sequence being captured — runs the general kernel.
    const bool two_streams = r->split >= 2 && !r->no_split && (!r->target_is_callers || r->bucket_call);
    if (two_streams) VPT_TRY(ensure_split_streams(r));
    if (same && r->cls.enabled && mcm_classes_runnable(r, a) && (two_streams || r->cls.one_stream)) return launch_mcm_classes<FUSE>(r, a);
    VPT_TRY(mcm_materialize(r));
    if (r->mcm_persistent && r->vol->channels == 1 && !r->vol->f32) { LAUNCH_MCM_PERSIST(FUSE, r, a); return VPT_OK; }
    return mcm_general_pass(r, a, FUSE);
}

// ---- what vpt_render.hip calls ------------------------------------------------------------------------------------------
#define LAUNCH(kernel, r, a, lds) hipLaunchKernelGGL(kernel, tile_grid(r), dim3(VPT_BLOCK), (lds), (r)->ctx->stream, (a))
int mcm_reset(vpt_renderer *r, const PassArgs &a, const vpt_uniforms *u) {
    LAUNCH(k_mcm_reset, r, a, 0);
    HIP_TRY(hipGetLastError());
    return mcm_classify(r, u);
}
int mcm_pass(vpt_renderer *r, const PassArgs &a, bool fuse_render) {
    return fuse_render ? launch_mcm_pass<true>(r, a) : launch_mcm_pass<false>(r, a);
}
int mcm_render_frame(vpt_renderer *r, const PassArgs &a) {
    // _renderFrame behind an integrate pass of the tile classes that is still on its two streams (render() hook by hook, AbstractRenderer.js:60-70):
    // the HIT tiles' texels by the context's stream, the MISS tiles' by the side stream — each behind its own class kernel, no join, so the
    // next pass's kernels overlap this one's as they do behind the fused call (1080p: 124 -> 104 us per frame hook by hook; fused 92)
    if (r->side_busy && r->last_layout == 1 && r->last_ranges == 2 && r->cls.valid && r->cls.n_hit > 0 && r->cls.n_miss > 0 &&
        !r->target_is_callers && !r->stop_events) {          // (the lists are those of the launch still in flight: a reset joins before it rebuilds them)
        PassArgs part = a;
        part.pm.tile_list = r->cls.list; part.pm.list_n = r->cls.n_hit;
        hipLaunchKernelGGL(k_mcm_render, dim3((unsigned)r->cls.n_hit), dim3(VPT_BLOCK), 0, r->ctx->stream, part);
        part.pm.tile_list = r->cls.list + r->cls.n_hit; part.pm.list_n = r->cls.n_miss;
        hipLaunchKernelGGL(k_mcm_render, dim3((unsigned)r->cls.n_miss), dim3(VPT_BLOCK), 0, r->side[0], part);
        return VPT_OK;
    }
    VPT_TRY(join_side(r));
    LAUNCH(k_mcm_render, r, a, 0);
    return VPT_OK;
}
#ifdef VPT_EVENT_TIMING
// instrumented builds only (tools/r04_event_timing.py binds it by name): the HIT-tile kernel's phase clocks summed over its waves since the last
// call, in 10 ns ticks — [0..4] per event: free path | cell + tables | load flight | blend + transfer function | decision + path end;
// [5] prologue, [6] epilogue, [7] one calibration mark per event, [8] waves
extern "C" VPT_API int vpt_probe_event_timing(vpt_renderer *r, uint64_t *out9) {
    if (!r || !out9 || !g_timing) return fail(VPT_ERR_INVALID, "no timing buffer (run a classified pass first)");
    VPT_TRY(join_side(r));
    HIP_TRY(hipSetDevice(r->ctx->device));
    std::vector<unsigned long long> host((size_t)VPT_TIMING_WAVES * 16);
    HIP_TRY(hipMemcpyAsync(host.data(), g_timing, host.size() * 8, hipMemcpyDeviceToHost, r->ctx->stream));
    HIP_TRY(hipMemsetAsync(g_timing, 0, host.size() * 8, r->ctx->stream));
    HIP_TRY(hipStreamSynchronize(r->ctx->stream));
    for (int k = 0; k < 9; k++) out9[k] = 0;
    for (size_t w = 0; w < VPT_TIMING_WAVES; w++) for (int k = 0; k < 9; k++) out9[k] += host[w * 16 + k];
    // the LAST launch's timeline: when its waves started and ended, relative to the first wave's start (10 ns ticks): out9[9 ..] =
    // { waves, start p50, start p90, start max, end p10, end p50, end p90, end max }
    std::vector<unsigned long long> st, en;
    unsigned long long t0 = ~0ull;
    for (size_t w = 0; w < VPT_TIMING_WAVES; w++) if (host[w * 16 + 8]) { st.push_back(host[w * 16 + 9]); en.push_back(host[w * 16 + 10]); t0 = std::min(t0, host[w * 16 + 9]); }
    for (int k = 9; k < 17; k++) out9[k] = 0;
    if (!st.empty()) {
        std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end());
        const size_t n = st.size();
        out9[9] = n; out9[10] = st[n / 2] - t0; out9[11] = st[n * 9 / 10] - t0; out9[12] = st[n - 1] - t0;
        out9[13] = en[n / 10] - t0; out9[14] = en[n / 2] - t0; out9[15] = en[n * 9 / 10] - t0; out9[16] = en[n - 1] - t0;
    }
    return VPT_OK;
}
#endif
