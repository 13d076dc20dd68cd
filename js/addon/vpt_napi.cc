// vpt_napi.cc — thin N-API addon over the C-ABI of include/vpt.h (libvpt_hip.so).
//
// One JS function per C entry point, same argument order; handles are napi_external values; every non-zero return
// code becomes a thrown Error carrying vpt_last_error() (the reference only throws Error(msg)).  No logic lives here:
// the host-side mirror of the reference's classes is js/vpt/*.js.  Built with plain g++ against the system's
// node_api.h (N-API v8, Node >= 12) — see js/addon/Makefile.
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "vpt.h"

#define NAPI_OK(call) do { if ((call) != napi_ok) { napi_throw_error(env, nullptr, "N-API call failed: " #call); return nullptr; } } while (0)

static napi_value throw_vpt(napi_env env) {
    napi_throw_error(env, nullptr, vpt_last_error());
    return nullptr;
}
#define VPT_CHECK(call) do { if ((call) != VPT_OK) return throw_vpt(env); } while (0)

static bool get_args(napi_env env, napi_callback_info info, size_t want, napi_value *argv) {
    size_t argc = want;
    if (napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr) != napi_ok || argc < want) {
        napi_throw_type_error(env, nullptr, "wrong number of arguments");
        return false;
    }
    return true;
}
static bool get_i32(napi_env env, napi_value v, int32_t *out) {
    if (napi_get_value_int32(env, v, out) != napi_ok) { napi_throw_type_error(env, nullptr, "expected a number"); return false; }
    return true;
}
template <typename T> static bool get_handle(napi_env env, napi_value v, T **out, bool allow_null = false) {
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok) return false;
    if (allow_null && (t == napi_null || t == napi_undefined)) { *out = nullptr; return true; }
    void *p = nullptr;
    if (t != napi_external || napi_get_value_external(env, v, &p) != napi_ok) {
        napi_throw_type_error(env, nullptr, "expected a native handle");
        return false;
    }
    *out = (T *)p;
    return true;
}
// any TypedArray / ArrayBuffer -> (pointer, byte length); null/undefined allowed when allow_null
static bool get_bytes(napi_env env, napi_value v, void **data, size_t *nbytes, bool allow_null = false) {
    napi_valuetype t;
    napi_typeof(env, v, &t);
    if (allow_null && (t == napi_null || t == napi_undefined)) { *data = nullptr; *nbytes = 0; return true; }
    bool is;
    if (napi_is_typedarray(env, v, &is) == napi_ok && is) {
        napi_typedarray_type ty; size_t len; napi_value ab; size_t off;
        if (napi_get_typedarray_info(env, v, &ty, &len, data, &ab, &off) != napi_ok) return false;
        static const size_t elem[] = { 1, 1, 1, 2, 2, 4, 4, 4, 8, 8, 8 };
        *nbytes = len * elem[ty];
        return true;
    }
    if (napi_is_arraybuffer(env, v, &is) == napi_ok && is) return napi_get_arraybuffer_info(env, v, data, nbytes) == napi_ok;
    if (napi_is_dataview(env, v, &is) == napi_ok && is) {
        napi_value ab; size_t off;
        return napi_get_dataview_info(env, v, nbytes, data, &ab, &off) == napi_ok;
    }
    napi_throw_type_error(env, nullptr, "expected a TypedArray, DataView or ArrayBuffer");
    return false;
}
static bool get_uniforms(napi_env env, napi_value v, const vpt_uniforms **u, bool allow_null) {
    void *p; size_t n;
    if (!get_bytes(env, v, &p, &n, allow_null)) return false;
    if (p && n < sizeof(vpt_uniforms)) { napi_throw_range_error(env, nullptr, "uniform block is smaller than struct vpt_uniforms"); return false; }
    *u = (const vpt_uniforms *)p;
    return true;
}
static napi_value make_external(napi_env env, void *p) {
    napi_value v;
    NAPI_OK(napi_create_external(env, p, nullptr, nullptr, &v));     // explicit destroy() is the contract (AbstractRenderer.js:51-58)
    return v;
}
static napi_value undefined(napi_env env) { napi_value v; napi_get_undefined(env, &v); return v; }
static napi_value number(napi_env env, double d) { napi_value v; napi_create_double(env, d, &v); return v; }

// ---- context ------------------------------------------------------------------------------------------
static napi_value DeviceCount(napi_env env, napi_callback_info) {
    int n = 0;
    VPT_CHECK(vpt_device_count(&n));
    return number(env, n);
}
static napi_value ContextCreate(napi_env env, napi_callback_info info) {
    napi_value a[1]; int32_t dev;
    if (!get_args(env, info, 1, a) || !get_i32(env, a[0], &dev)) return nullptr;
    vpt_context *c = nullptr;
    VPT_CHECK(vpt_context_create(dev, &c));
    return make_external(env, c);
}
static napi_value ContextDestroy(napi_env env, napi_callback_info info) {
    napi_value a[1]; vpt_context *c;
    if (!get_args(env, info, 1, a) || !get_handle(env, a[0], &c)) return nullptr;
    VPT_CHECK(vpt_context_destroy(c));
    return undefined(env);
}
static napi_value ContextSynchronize(napi_env env, napi_callback_info info) {
    napi_value a[1]; vpt_context *c;
    if (!get_args(env, info, 1, a) || !get_handle(env, a[0], &c)) return nullptr;
    VPT_CHECK(vpt_context_synchronize(c));
    return undefined(env);
}
static napi_value Version(napi_env env, napi_callback_info) {
    napi_value v; napi_create_string_utf8(env, vpt_version(), NAPI_AUTO_LENGTH, &v); return v;
}

// ---- volume -------------------------------------------------------------------------------------------
static napi_value VolumeCreate(napi_env env, napi_callback_info info) {
    napi_value a[5]; vpt_context *c; int32_t w, h, d, f;
    if (!get_args(env, info, 5, a) || !get_handle(env, a[0], &c) || !get_i32(env, a[1], &w) || !get_i32(env, a[2], &h) ||
        !get_i32(env, a[3], &d) || !get_i32(env, a[4], &f)) return nullptr;
    vpt_volume *v = nullptr;
    VPT_CHECK(vpt_volume_create(c, w, h, d, f, &v));
    return make_external(env, v);
}
static napi_value VolumeUploadBlock(napi_env env, napi_callback_info info) {
    napi_value a[8]; vpt_volume *v; int32_t p[6]; void *data; size_t n;
    if (!get_args(env, info, 8, a) || !get_handle(env, a[0], &v)) return nullptr;
    for (int i = 0; i < 6; i++) if (!get_i32(env, a[1 + i], &p[i])) return nullptr;
    if (!get_bytes(env, a[7], &data, &n)) return nullptr;
    VPT_CHECK(vpt_volume_upload_block(v, p[0], p[1], p[2], p[3], p[4], p[5], data, n));
    return undefined(env);
}
static napi_value VolumeFinalize(napi_env env, napi_callback_info info) {
    napi_value a[1]; vpt_volume *v;
    if (!get_args(env, info, 1, a) || !get_handle(env, a[0], &v)) return nullptr;
    VPT_CHECK(vpt_volume_finalize(v));
    return undefined(env);
}
static napi_value VolumeSetFilter(napi_env env, napi_callback_info info) {
    napi_value a[2]; vpt_volume *v; int32_t f;
    if (!get_args(env, info, 2, a) || !get_handle(env, a[0], &v) || !get_i32(env, a[1], &f)) return nullptr;
    VPT_CHECK(vpt_volume_set_filter(v, f));
    return undefined(env);
}
static napi_value VolumeDestroy(napi_env env, napi_callback_info info) {
    napi_value a[1]; vpt_volume *v;
    if (!get_args(env, info, 1, a) || !get_handle(env, a[0], &v)) return nullptr;
    VPT_CHECK(vpt_volume_destroy(v));
    return undefined(env);
}
static napi_value VolumeBrickedBytes(napi_env env, napi_callback_info info) {
    napi_value a[1]; vpt_volume *v; uint64_t n = 0;
    if (!get_args(env, info, 1, a) || !get_handle(env, a[0], &v)) return nullptr;
    VPT_CHECK(vpt_volume_bricked_bytes(v, &n));
    return number(env, (double)n);
}

// ---- renderer -----------------------------------------------------------------------------------------
static napi_value RendererCreate(napi_env env, napi_callback_info info) {
    napi_value a[4]; vpt_context *c; int32_t kind, w, h;
    if (!get_args(env, info, 4, a) || !get_handle(env, a[0], &c) || !get_i32(env, a[1], &kind) || !get_i32(env, a[2], &w) ||
        !get_i32(env, a[3], &h)) return nullptr;
    vpt_renderer *r = nullptr;
    VPT_CHECK(vpt_renderer_create(c, kind, w, h, &r));
    return make_external(env, r);
}
static napi_value RendererDestroy(napi_env env, napi_callback_info info) {
    napi_value a[1]; vpt_renderer *r;
    if (!get_args(env, info, 1, a) || !get_handle(env, a[0], &r)) return nullptr;
    VPT_CHECK(vpt_renderer_destroy(r));
    return undefined(env);
}
static napi_value RendererSetShard(napi_env env, napi_callback_info info) {
    napi_value a[4]; vpt_renderer *r; int32_t rank, world, rows;
    if (!get_args(env, info, 4, a) || !get_handle(env, a[0], &r) || !get_i32(env, a[1], &rank) || !get_i32(env, a[2], &world) ||
        !get_i32(env, a[3], &rows)) return nullptr;
    VPT_CHECK(vpt_renderer_set_shard(r, rank, world, rows));
    return undefined(env);
}
static napi_value RendererLocalRows(napi_env env, napi_callback_info info) {
    napi_value a[1]; vpt_renderer *r; int rows = 0;
    if (!get_args(env, info, 1, a) || !get_handle(env, a[0], &r)) return nullptr;
    VPT_CHECK(vpt_renderer_local_rows(r, &rows));
    return number(env, rows);
}
static napi_value RendererGlobalRow(napi_env env, napi_callback_info info) {
    napi_value a[2]; vpt_renderer *r; int32_t l; int g = 0;
    if (!get_args(env, info, 2, a) || !get_handle(env, a[0], &r) || !get_i32(env, a[1], &l)) return nullptr;
    VPT_CHECK(vpt_renderer_global_row(r, l, &g));
    return number(env, g);
}
static napi_value RendererSetVolume(napi_env env, napi_callback_info info) {
    napi_value a[2]; vpt_renderer *r; vpt_volume *v;
    if (!get_args(env, info, 2, a) || !get_handle(env, a[0], &r) || !get_handle(env, a[1], &v, true)) return nullptr;
    VPT_CHECK(vpt_renderer_set_volume(r, v));
    return undefined(env);
}
static napi_value RendererSetImage(napi_env env, napi_callback_info info, bool tf) {
    napi_value a[4]; vpt_renderer *r; void *data; size_t n; int32_t w, h;
    if (!get_args(env, info, 4, a) || !get_handle(env, a[0], &r) || !get_bytes(env, a[1], &data, &n) || !get_i32(env, a[2], &w) ||
        !get_i32(env, a[3], &h)) return nullptr;
    if (w < 1 || h < 1 || n < (size_t)w * (size_t)h * 4) { napi_throw_range_error(env, nullptr, "image data shorter than width*height*4"); return nullptr; }
    if (tf) VPT_CHECK(vpt_renderer_set_transfer_function(r, (const uint8_t *)data, w, h));
    else VPT_CHECK(vpt_renderer_set_environment(r, (const uint8_t *)data, w, h));
    return undefined(env);
}
static napi_value RendererSetTransferFunction(napi_env env, napi_callback_info info) { return RendererSetImage(env, info, true); }
static napi_value RendererSetEnvironment(napi_env env, napi_callback_info info) { return RendererSetImage(env, info, false); }
static napi_value RendererResize(napi_env env, napi_callback_info info) {
    napi_value a[3]; vpt_renderer *r; int32_t w, h;
    if (!get_args(env, info, 3, a) || !get_handle(env, a[0], &r) || !get_i32(env, a[1], &w) || !get_i32(env, a[2], &h)) return nullptr;
    VPT_CHECK(vpt_renderer_resize(r, w, h));
    return undefined(env);
}
typedef int (*pass_fn)(vpt_renderer *, const vpt_uniforms *);
static napi_value RendererPass(napi_env env, napi_callback_info info, pass_fn fn, bool allow_null) {
    napi_value a[2]; vpt_renderer *r; const vpt_uniforms *u;
    if (!get_args(env, info, 2, a) || !get_handle(env, a[0], &r) || !get_uniforms(env, a[1], &u, allow_null)) return nullptr;
    VPT_CHECK(fn(r, u));
    return undefined(env);
}
static napi_value RendererReset(napi_env env, napi_callback_info info) { return RendererPass(env, info, vpt_renderer_reset, true); }
static napi_value RendererGenerate(napi_env env, napi_callback_info info) { return RendererPass(env, info, vpt_renderer_generate, false); }
static napi_value RendererIntegrate(napi_env env, napi_callback_info info) { return RendererPass(env, info, vpt_renderer_integrate, false); }
static napi_value RendererRenderFrame(napi_env env, napi_callback_info info) { return RendererPass(env, info, vpt_renderer_render_frame, true); }
static napi_value RendererRender(napi_env env, napi_callback_info info) { return RendererPass(env, info, vpt_renderer_render, false); }
static napi_value RendererRead(napi_env env, napi_callback_info info) {
    napi_value a[3]; vpt_renderer *r; int32_t which; void *dst; size_t n;
    if (!get_args(env, info, 3, a) || !get_handle(env, a[0], &r) || !get_i32(env, a[1], &which) || !get_bytes(env, a[2], &dst, &n)) return nullptr;
    VPT_CHECK(vpt_renderer_read(r, which, dst, n));
    return undefined(env);
}
static napi_value RendererReadFrameSlot(napi_env env, napi_callback_info info) {      // vpt_renderer_read_frame_slot (VPT_PLAY_FRAMES)
    napi_value a[3]; vpt_renderer *r; int32_t slot; void *dst; size_t n;
    if (!get_args(env, info, 3, a) || !get_handle(env, a[0], &r) || !get_i32(env, a[1], &slot) || !get_bytes(env, a[2], &dst, &n)) return nullptr;
    VPT_CHECK(vpt_renderer_read_frame_slot(r, slot, dst, n));
    return undefined(env);
}
static napi_value RendererSampleCount(napi_env env, napi_callback_info info) {
    napi_value a[1]; vpt_renderer *r; uint64_t n = 0;
    if (!get_args(env, info, 1, a) || !get_handle(env, a[0], &r)) return nullptr;
    VPT_CHECK(vpt_renderer_sample_count(r, &n));
    return number(env, (double)n);
}
static napi_value RendererClearSampleCount(napi_env env, napi_callback_info info) {
    napi_value a[1]; vpt_renderer *r;
    if (!get_args(env, info, 1, a) || !get_handle(env, a[0], &r)) return nullptr;
    VPT_CHECK(vpt_renderer_clear_sample_count(r));
    return undefined(env);
}
static napi_value RendererSetProfiling(napi_env env, napi_callback_info info) {
    napi_value a[2]; vpt_renderer *r; int32_t on;
    if (!get_args(env, info, 2, a) || !get_handle(env, a[0], &r) || !get_i32(env, a[1], &on)) return nullptr;
    VPT_CHECK(vpt_renderer_set_profiling(r, on));
    return undefined(env);
}
static napi_value RendererProfile(napi_env env, napi_callback_info info) {
    napi_value a[1]; vpt_renderer *r; double ms = 0; uint32_t n = 0;
    if (!get_args(env, info, 1, a) || !get_handle(env, a[0], &r)) return nullptr;
    VPT_CHECK(vpt_renderer_profile(r, &ms, &n));
    napi_value o; napi_create_object(env, &o);
    napi_set_named_property(env, o, "totalMs", number(env, ms));
    napi_set_named_property(env, o, "launches", number(env, n));
    return o;
}

static napi_value RendererSetOption(napi_env env, napi_callback_info info) {
    napi_value a[3]; vpt_renderer *r; int32_t opt, val;
    if (!get_args(env, info, 3, a) || !get_handle(env, a[0], &r) || !get_i32(env, a[1], &opt) || !get_i32(env, a[2], &val)) return nullptr;
    VPT_CHECK(vpt_renderer_set_option(r, opt, val));
    return undefined(env);
}
// rendererSetLaoParams(handle, params: a 48-byte view laid out as struct vpt_lao_params)
static napi_value RendererSetLaoParams(napi_env env, napi_callback_info info) {
    napi_value a[2]; vpt_renderer *r; void *p; size_t n;
    if (!get_args(env, info, 2, a) || !get_handle(env, a[0], &r) || !get_bytes(env, a[1], &p, &n)) return nullptr;
    if (n != sizeof(vpt_lao_params)) { napi_throw_range_error(env, nullptr, "LAO parameter block must be sizeof(struct vpt_lao_params) bytes"); return nullptr; }
    VPT_CHECK(vpt_renderer_set_lao_params(r, (const vpt_lao_params *)p));
    return undefined(env);
}
// rendererSetOcclusionSamples(handle, xy: Float32Array of 2 floats per sample) — DOSRenderer.js:103-140
static napi_value RendererSetOcclusionSamples(napi_env env, napi_callback_info info) {
    napi_value a[2]; vpt_renderer *r; void *p; size_t n;
    if (!get_args(env, info, 2, a) || !get_handle(env, a[0], &r) || !get_bytes(env, a[1], &p, &n)) return nullptr;
    if (n % (2 * sizeof(float)) != 0) { napi_throw_range_error(env, nullptr, "occlusion samples are 2 floats each"); return nullptr; }
    VPT_CHECK(vpt_renderer_set_occlusion_samples(r, (const float *)p, (int)(n / (2 * sizeof(float)))));
    return undefined(env);
}
// rendererIntegrateSlices(handle, uniforms, slices: Float32Array of 3 floats per slice) — DOSRenderer.js:199-259
static napi_value RendererIntegrateSlices(napi_env env, napi_callback_info info) {
    napi_value a[3]; vpt_renderer *r; const vpt_uniforms *u; void *p; size_t n;
    if (!get_args(env, info, 3, a) || !get_handle(env, a[0], &r) || !get_uniforms(env, a[1], &u, false) || !get_bytes(env, a[2], &p, &n)) return nullptr;
    if (n % (3 * sizeof(float)) != 0) { napi_throw_range_error(env, nullptr, "slices are 3 floats each"); return nullptr; }
    VPT_CHECK(vpt_renderer_integrate_slices(r, u, (const float *)p, (int)(n / (3 * sizeof(float)))));
    return undefined(env);
}
// rendererPlay(handle, baseUniforms, frameVars: Float32Array of 8 floats per frame, useGraph)
static bool get_frame_vars(napi_env env, napi_value v, const float **vars, int *count) {
    void *p; size_t n;
    if (!get_bytes(env, v, &p, &n)) return false;
    if (n == 0 || n % (8 * sizeof(float)) != 0) { napi_throw_range_error(env, nullptr, "frame variables are 8 floats per frame"); return false; }
    *vars = (const float *)p; *count = (int)(n / (8 * sizeof(float)));
    return true;
}
static napi_value RendererPlay(napi_env env, napi_callback_info info) {
    napi_value a[4]; vpt_renderer *r; const vpt_uniforms *u; const float *vars; int count; int32_t graph;
    if (!get_args(env, info, 4, a) || !get_handle(env, a[0], &r) || !get_uniforms(env, a[1], &u, false) ||
        !get_frame_vars(env, a[2], &vars, &count) || !get_i32(env, a[3], &graph)) return nullptr;
    VPT_CHECK(vpt_renderer_play(r, u, vars, count, graph));
    return undefined(env);
}

// ---- tone mappers -------------------------------------------------------------------------------------
static napi_value TonemapperCreate(napi_env env, napi_callback_info info) {
    napi_value a[4]; vpt_context *c; int32_t kind, w, h;
    if (!get_args(env, info, 4, a) || !get_handle(env, a[0], &c) || !get_i32(env, a[1], &kind) || !get_i32(env, a[2], &w) ||
        !get_i32(env, a[3], &h)) return nullptr;
    vpt_tonemapper *t = nullptr;
    VPT_CHECK(vpt_tonemapper_create(c, kind, w, h, &t));
    return make_external(env, t);
}
static napi_value TonemapperDestroy(napi_env env, napi_callback_info info) {
    napi_value a[1]; vpt_tonemapper *t;
    if (!get_args(env, info, 1, a) || !get_handle(env, a[0], &t)) return nullptr;
    VPT_CHECK(vpt_tonemapper_destroy(t));
    return undefined(env);
}
static napi_value TonemapperResize(napi_env env, napi_callback_info info) {
    napi_value a[3]; vpt_tonemapper *t; int32_t w, h;
    if (!get_args(env, info, 3, a) || !get_handle(env, a[0], &t) || !get_i32(env, a[1], &w) || !get_i32(env, a[2], &h)) return nullptr;
    VPT_CHECK(vpt_tonemapper_resize(t, w, h));
    return undefined(env);
}
static napi_value TonemapperSetSource(napi_env env, napi_callback_info info) {
    napi_value a[2]; vpt_tonemapper *t; vpt_renderer *r;
    if (!get_args(env, info, 2, a) || !get_handle(env, a[0], &t) || !get_handle(env, a[1], &r, true)) return nullptr;
    VPT_CHECK(vpt_tonemapper_set_source(t, r));
    return undefined(env);
}
static napi_value TonemapperSetSourceImage(napi_env env, napi_callback_info info) {
    napi_value a[4]; vpt_tonemapper *t; void *data; size_t n; int32_t w, rows;
    if (!get_args(env, info, 4, a) || !get_handle(env, a[0], &t) || !get_bytes(env, a[1], &data, &n) || !get_i32(env, a[2], &w) ||
        !get_i32(env, a[3], &rows)) return nullptr;
    if (w < 1 || rows < 1 || n < (size_t)w * (size_t)rows * 8) { napi_throw_range_error(env, nullptr, "image data shorter than width*rows*8"); return nullptr; }
    VPT_CHECK(vpt_tonemapper_set_source_image(t, data, w, rows));
    return undefined(env);
}
// tonemapperRender(handle, params: Float32Array(8) = low, mid, high, saturation, min, max, exposure, gamma)
static napi_value TonemapperRender(napi_env env, napi_callback_info info) {
    napi_value a[2]; vpt_tonemapper *t; void *p; size_t n;
    if (!get_args(env, info, 2, a) || !get_handle(env, a[0], &t) || !get_bytes(env, a[1], &p, &n)) return nullptr;
    if (n < sizeof(vpt_tonemap_params)) { napi_throw_range_error(env, nullptr, "parameter block is smaller than struct vpt_tonemap_params"); return nullptr; }
    VPT_CHECK(vpt_tonemapper_render(t, (const vpt_tonemap_params *)p));
    return undefined(env);
}
static napi_value TonemapperSetOption(napi_env env, napi_callback_info info) {
    napi_value a[3]; vpt_tonemapper *t; int32_t opt, val;
    if (!get_args(env, info, 3, a) || !get_handle(env, a[0], &t) || !get_i32(env, a[1], &opt) || !get_i32(env, a[2], &val)) return nullptr;
    VPT_CHECK(vpt_tonemapper_set_option(t, opt, val));
    return undefined(env);
}
static napi_value TonemapperRows(napi_env env, napi_callback_info info) {
    napi_value a[1]; vpt_tonemapper *t; int rows = 0;
    if (!get_args(env, info, 1, a) || !get_handle(env, a[0], &t)) return nullptr;
    VPT_CHECK(vpt_tonemapper_rows(t, &rows));
    return number(env, rows);
}
static napi_value TonemapperRead(napi_env env, napi_callback_info info) {
    napi_value a[2]; vpt_tonemapper *t; void *dst; size_t n;
    if (!get_args(env, info, 2, a) || !get_handle(env, a[0], &t) || !get_bytes(env, a[1], &dst, &n)) return nullptr;
    VPT_CHECK(vpt_tonemapper_read(t, dst, n));
    return undefined(env);
}

// transferFunctionRasterize(ctx, bumps: Float32Array [count][8], width, height, unpremultiply, out: Uint8Array [height][width][4])
static napi_value TransferFunctionRasterize(napi_env env, napi_callback_info info) {
    napi_value a[6]; vpt_context *c; void *bumps, *dst; size_t nb, nd; int32_t w, h, unpre;
    if (!get_args(env, info, 6, a) || !get_handle(env, a[0], &c) || !get_bytes(env, a[1], &bumps, &nb) || !get_i32(env, a[2], &w) ||
        !get_i32(env, a[3], &h) || !get_i32(env, a[4], &unpre) || !get_bytes(env, a[5], &dst, &nd)) return nullptr;
    if (w < 1 || h < 1 || nd < (size_t)w * (size_t)h * 4) { napi_throw_range_error(env, nullptr, "the output holds width * height * 4 bytes"); return nullptr; }
    VPT_CHECK(vpt_transfer_function_rasterize(c, (const vpt_tf_bump *)bumps, (int)(nb / sizeof(vpt_tf_bump)), w, h, unpre, (uint8_t *)dst));
    return undefined(env);
}

// ---- multi-GPU frame gather ---------------------------------------------------------------------------
static napi_value GatherUniqueId(napi_env env, napi_callback_info info) {
    (void)info;
    void *data; napi_value ab;
    NAPI_OK(napi_create_arraybuffer(env, 128, &data, &ab));
    VPT_CHECK(vpt_gather_unique_id(data));
    return ab;
}
static napi_value GatherCreate(napi_env env, napi_callback_info info) {
    napi_value a[4]; vpt_renderer *r; void *id; size_t n; int32_t rank, world;
    if (!get_args(env, info, 4, a) || !get_handle(env, a[0], &r) || !get_bytes(env, a[1], &id, &n) || !get_i32(env, a[2], &rank) ||
        !get_i32(env, a[3], &world)) return nullptr;
    if (n < 128) { napi_throw_range_error(env, nullptr, "the RCCL unique id is 128 bytes"); return nullptr; }
    vpt_gather *g = nullptr;
    VPT_CHECK(vpt_gather_create(r, id, rank, world, &g));
    return make_external(env, g);
}
static napi_value GatherDestroy(napi_env env, napi_callback_info info) {
    napi_value a[1]; vpt_gather *g;
    if (!get_args(env, info, 1, a) || !get_handle(env, a[0], &g)) return nullptr;
    VPT_CHECK(vpt_gather_destroy(g));
    return undefined(env);
}
static napi_value GatherSetRoot(napi_env env, napi_callback_info info) {
    napi_value a[2]; vpt_gather *g; int32_t root;
    if (!get_args(env, info, 2, a) || !get_handle(env, a[0], &g) || !get_i32(env, a[1], &root)) return nullptr;
    VPT_CHECK(vpt_gather_set_root(g, root));
    return undefined(env);
}
static napi_value GatherRender(napi_env env, napi_callback_info info) {
    napi_value a[2]; vpt_gather *g; const vpt_uniforms *u;
    if (!get_args(env, info, 2, a) || !get_handle(env, a[0], &g) || !get_uniforms(env, a[1], &u, false)) return nullptr;
    VPT_CHECK(vpt_gather_render(g, u));
    return undefined(env);
}
static napi_value GatherPlay(napi_env env, napi_callback_info info) {
    napi_value a[4]; vpt_gather *g; const vpt_uniforms *u; const float *vars; int count; int32_t mode;
    if (!get_args(env, info, 4, a) || !get_handle(env, a[0], &g) || !get_uniforms(env, a[1], &u, false) ||
        !get_frame_vars(env, a[2], &vars, &count) || !get_i32(env, a[3], &mode)) return nullptr;
    VPT_CHECK(vpt_gather_play(g, u, vars, count, mode));
    return undefined(env);
}
static napi_value GatherSynchronize(napi_env env, napi_callback_info info) {
    napi_value a[1]; vpt_gather *g;
    if (!get_args(env, info, 1, a) || !get_handle(env, a[0], &g)) return nullptr;
    VPT_CHECK(vpt_gather_synchronize(g));
    return undefined(env);
}
static napi_value GatherReadFrame(napi_env env, napi_callback_info info) {
    napi_value a[2]; vpt_gather *g; void *dst; size_t n;
    if (!get_args(env, info, 2, a) || !get_handle(env, a[0], &g) || !get_bytes(env, a[1], &dst, &n)) return nullptr;
    VPT_CHECK(vpt_gather_read_frame(g, dst, n));
    return undefined(env);
}

#define EXPORT(name, fn) do { napi_value f; napi_create_function(env, name, NAPI_AUTO_LENGTH, fn, nullptr, &f); \
                              napi_set_named_property(env, exports, name, f); } while (0)
#define CONST(name) do { napi_set_named_property(env, exports, #name, number(env, name)); } while (0)

static napi_value Init(napi_env env, napi_value exports) {
    EXPORT("deviceCount", DeviceCount); EXPORT("contextCreate", ContextCreate); EXPORT("contextDestroy", ContextDestroy);
    EXPORT("contextSynchronize", ContextSynchronize); EXPORT("version", Version);
    EXPORT("volumeCreate", VolumeCreate); EXPORT("volumeUploadBlock", VolumeUploadBlock); EXPORT("volumeFinalize", VolumeFinalize);
    EXPORT("volumeSetFilter", VolumeSetFilter); EXPORT("volumeDestroy", VolumeDestroy); EXPORT("volumeBrickedBytes", VolumeBrickedBytes);
    EXPORT("rendererCreate", RendererCreate); EXPORT("rendererDestroy", RendererDestroy); EXPORT("rendererSetShard", RendererSetShard);
    EXPORT("rendererLocalRows", RendererLocalRows); EXPORT("rendererGlobalRow", RendererGlobalRow);
    EXPORT("rendererSetVolume", RendererSetVolume); EXPORT("rendererSetTransferFunction", RendererSetTransferFunction);
    EXPORT("rendererSetEnvironment", RendererSetEnvironment); EXPORT("rendererResize", RendererResize);
    EXPORT("rendererReset", RendererReset); EXPORT("rendererGenerate", RendererGenerate); EXPORT("rendererIntegrate", RendererIntegrate);
    EXPORT("rendererRenderFrame", RendererRenderFrame); EXPORT("rendererRender", RendererRender); EXPORT("rendererRead", RendererRead); EXPORT("rendererReadFrameSlot", RendererReadFrameSlot);
    EXPORT("rendererSampleCount", RendererSampleCount); EXPORT("rendererClearSampleCount", RendererClearSampleCount);
    EXPORT("rendererSetProfiling", RendererSetProfiling); EXPORT("rendererProfile", RendererProfile);
    EXPORT("rendererSetOption", RendererSetOption); EXPORT("rendererSetLaoParams", RendererSetLaoParams);
    EXPORT("rendererSetOcclusionSamples", RendererSetOcclusionSamples); EXPORT("rendererIntegrateSlices", RendererIntegrateSlices); EXPORT("rendererPlay", RendererPlay);
    EXPORT("tonemapperCreate", TonemapperCreate); EXPORT("tonemapperDestroy", TonemapperDestroy); EXPORT("tonemapperResize", TonemapperResize);
    EXPORT("tonemapperSetSource", TonemapperSetSource); EXPORT("tonemapperSetSourceImage", TonemapperSetSourceImage);
    EXPORT("tonemapperSetOption", TonemapperSetOption); CONST(VPT_TONEMAPPER_OPTION_TABLE); CONST(VPT_TONEMAPPER_OPTION_FUSE); CONST(VPT_TONEMAPPER_TABLE_NEVER);
    CONST(VPT_TONEMAPPER_TABLE_ALWAYS); CONST(VPT_TONEMAPPER_TABLE_AUTO);
    EXPORT("tonemapperRender", TonemapperRender); EXPORT("tonemapperRows", TonemapperRows); EXPORT("tonemapperRead", TonemapperRead);
    EXPORT("transferFunctionRasterize", TransferFunctionRasterize);
    CONST(VPT_TONEMAPPER_ARTISTIC); CONST(VPT_TONEMAPPER_RANGE); CONST(VPT_TONEMAPPER_REINHARD); CONST(VPT_TONEMAPPER_REINHARD2);
    CONST(VPT_TONEMAPPER_UNCHARTED2); CONST(VPT_TONEMAPPER_FILMIC); CONST(VPT_TONEMAPPER_UNREAL); CONST(VPT_TONEMAPPER_ACES);
    CONST(VPT_TONEMAPPER_LOTTES); CONST(VPT_TONEMAPPER_UCHIMURA);
    EXPORT("gatherUniqueId", GatherUniqueId); EXPORT("gatherCreate", GatherCreate); EXPORT("gatherDestroy", GatherDestroy);
    EXPORT("gatherSetRoot", GatherSetRoot); EXPORT("gatherRender", GatherRender); EXPORT("gatherPlay", GatherPlay); EXPORT("gatherSynchronize", GatherSynchronize);
    EXPORT("gatherReadFrame", GatherReadFrame);
    CONST(VPT_OPTION_MCS_PERSISTENT); CONST(VPT_OPTION_MCM_PERSISTENT); CONST(VPT_OPTION_FAST_MATH); CONST(VPT_OPTION_BOUNDARY_ATLAS); CONST(VPT_OPTION_SPLIT_STREAMS); CONST(VPT_OPTION_TILE_CLASSES); CONST(VPT_OPTION_VERIFY_TILE_CLASSES); CONST(VPT_OPTION_BUCKET_KERNEL); CONST(VPT_OPTION_COLUMN_RECORDS);
    CONST(VPT_PLAY_EAGER); CONST(VPT_PLAY_GRAPH); CONST(VPT_PLAY_FUSED); CONST(VPT_PLAY_FRAMES); CONST(VPT_FRAME_SLOTS);
    CONST(VPT_RENDERER_MIP); CONST(VPT_RENDERER_EAM); CONST(VPT_RENDERER_MCS); CONST(VPT_RENDERER_MCM);
    CONST(VPT_RENDERER_ISO); CONST(VPT_RENDERER_DEPTH); CONST(VPT_RENDERER_LAO); CONST(VPT_RENDERER_DOS); CONST(VPT_BUFFER_DOS_OCCLUSION);
    CONST(VPT_FILTER_NEAREST); CONST(VPT_FILTER_LINEAR); CONST(VPT_FORMAT_R8); CONST(VPT_FORMAT_RG8); CONST(VPT_FORMAT_R32F); CONST(VPT_FORMAT_RG32F);
    CONST(VPT_BUFFER_RENDER); CONST(VPT_BUFFER_FRAME); CONST(VPT_BUFFER_ACCUM);
    CONST(VPT_BUFFER_MCM_POSITION); CONST(VPT_BUFFER_MCM_DIRECTION); CONST(VPT_BUFFER_MCM_TRANSMITTANCE); CONST(VPT_BUFFER_MCM_RADIANCE);
    napi_set_named_property(env, exports, "UNIFORMS_BYTES", number(env, (double)sizeof(vpt_uniforms)));
    return exports;
}
NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
