// vpt_post.hip — what follows a rendered frame: the tone mappers (src/js/tonemappers/*, src/glsl/tonemappers/*)
// and the multi-GPU frame gather over RCCL.
#include <dlfcn.h>
#include "vpt_internal.h"
#include "vpt_tonemap_kernels.h"

// ---------------------------------------------------------------------------------------------
// tone mappers
// ---------------------------------------------------------------------------------------------
static void tonemapper_disarm(vpt_tonemapper *t) {
    if (t && t->source && t->source->tm_owner == t) { t->source->tm_owner = nullptr; t->source->tm_mode = 0; t->source->tm_valid = false; }
}
void tonemappers_unbind(vpt_context *c, vpt_renderer *r) {
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
    BucketCall bucket_call{ r };
    int i0 = 0;
    while (r->kind == VPT_RENDERER_MCM && stride_bytes / 4 <= 0xffffffffull && i0 < count) {
        const int n = std::min(count - i0, VPT_BUCKET_FRAMES);
        bool ready = false;
        VPT_TRY(mcm_bucket_ready(r, a, &ready));
        if (!ready) break;
        Timed tm(r, true, (uint32_t)n);
        VPT_TRY(mcm_bucket(r, a, v + i0, n, (char *)first_target + (size_t)i0 * stride_bytes, (uint32_t)(stride_bytes / 4), false, r->tm_table));
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
    VPT_TRY(join_side(r));                                   // one join per bucket: the caller's collective comes next on the context's stream
    r->warmed = true;
    if (r->kind == VPT_RENDERER_MCM) r->samples_host += r->valid_pixels * (uint64_t)base->steps * (uint64_t)count;
    return VPT_OK;
}
// ---------------------------------------------------------------------------------------------
// the transfer-function widget's canvas (ui/TransferFunction/TransferFunction.js:110-121, glsl/TransferFunction.glsl:32-35) as data:
// one thread per texel walks the bumps in drawing order; the 8-bit target is re-read between two bumps like the canvas's own buffer
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_tf_rasterize(const vpt_tf_bump *bumps, int count, int W, int H, int unpremultiply, uint32_t *out) {
    const int i = (int)(blockIdx.x * 16u + (threadIdx.x & 15u)), j = (int)(blockIdx.y * 16u + (threadIdx.x >> 4));
    if (i >= W || j >= H) return;
    const float u = ((float)i + 0.5f) / (float)W;
    const float v = ((float)(H - 1 - j) + 0.5f) / (float)H;            // texel row 0 = the top row of the canvas
    uint32_t d0 = 0u, d1 = 0u, d2 = 0u, d3 = 0u;                       // gl.clear: (0, 0, 0, 0)
    for (int k = 0; k < count; k++) {
        const vpt_tf_bump b = bumps[k];
        const float dx = (b.x - u) / b.sx, dy = (b.y - v) / b.sy;
        const float r = sqrtf(dx * dx + dy * dy);                       // length()
        const float e = vpt_expf(-(r * r));
        // a fixed-point target: the fragment's colour is clamped to [0, 1] before the blend, the result again before the store
        const float s0 = vclamp01(b.r * e), s1 = vclamp01(b.g * e), s2 = vclamp01(b.b * e), s3 = vclamp01(b.a * e);
        const float k1 = 1.0f - s3;
        d0 = to_unorm8(s0 + ((float)d0 / 255.0f) * k1); d1 = to_unorm8(s1 + ((float)d1 / 255.0f) * k1);
        d2 = to_unorm8(s2 + ((float)d2 / 255.0f) * k1); d3 = to_unorm8(s3 + ((float)d3 / 255.0f) * k1);
    }
    if (unpremultiply) {
        if (d3 == 0u) { d0 = d1 = d2 = 0u; }
        else {
            d0 = min(255u, (d0 * 255u + d3 / 2u) / d3); d1 = min(255u, (d1 * 255u + d3 / 2u) / d3); d2 = min(255u, (d2 * 255u + d3 / 2u) / d3);
        }
    }
    out[(size_t)j * W + i] = d0 | (d1 << 8) | (d2 << 16) | (d3 << 24);
}
extern "C" int vpt_transfer_function_rasterize(vpt_context *c, const vpt_tf_bump *bumps, int count, int width, int height, int unpremultiply,
                                               uint8_t *rgba_out) {
    if (!c || !rgba_out || (count > 0 && !bumps)) return fail(VPT_ERR_INVALID, "null argument");
    if (count < 0 || count > 65536) return fail(VPT_ERR_INVALID, "bump count %d not in [0, 65536]", count);
    if (width < 1 || height < 1 || width > 16384 || height > 16384) return fail(VPT_ERR_INVALID, "transfer function %d x %d not in [1, 16384]^2", width, height);
    for (int k = 0; k < count; k++)
        if (bumps[k].sx == 0.0f || bumps[k].sy == 0.0f) return fail(VPT_ERR_INVALID, "bump %d has a zero size", k);
    HIP_TRY(hipSetDevice(c->device));
    vpt_tf_bump *dev_bumps = nullptr; uint32_t *dev_out = nullptr;
    const size_t nb = (size_t)std::max(count, 1) * sizeof(vpt_tf_bump), no = (size_t)width * height * 4;
    HIP_TRY(hipMalloc(&dev_bumps, nb));
    if (hipMalloc(&dev_out, no) != hipSuccess) { hipFree(dev_bumps); return fail(VPT_ERR_HIP, "hipMalloc of %zu bytes failed", no); }
    int rc = VPT_OK;
    do {
        if (count > 0 && hipMemcpyAsync(dev_bumps, bumps, (size_t)count * sizeof(vpt_tf_bump), hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = fail(VPT_ERR_HIP, "bump upload failed"); break; }
        hipLaunchKernelGGL(k_tf_rasterize, dim3((unsigned)((width + 15) / 16), (unsigned)((height + 15) / 16)), dim3(256), 0, c->stream,
                           dev_bumps, count, width, height, unpremultiply, dev_out);
        if (hipGetLastError() != hipSuccess) { rc = fail(VPT_ERR_HIP, "k_tf_rasterize launch failed"); break; }
        if (hipMemcpyAsync(rgba_out, dev_out, no, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { rc = fail(VPT_ERR_HIP, "transfer function read-back failed"); break; }
    } while (0);
    hipFree(dev_bumps); hipFree(dev_out);
    return rc;
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
        { int jr = ensure_split_streams(r); if (jr == VPT_OK) jr = join_side(r); if (jr != VPT_OK) { delete g; return jr; } }
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
    if (rc != VPT_OK) { char keep[512]; strncpy(keep, vpt_error_buffer(), sizeof(keep)); keep[511] = 0; vpt_gather_destroy(g); strncpy(vpt_error_buffer(), keep, 512); return rc; }
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
        VPT_TRY(mcm_multi(r, a, fused_passes, nullptr));
        advance_frames(r, fused_passes);
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
