// vpt_extra.hip — the ISO, Depth, LAO and DOS renderers' passes (vpt_kernels_iso_depth.h) behind
// vpt_render.hip's entry points, and the DOS renderer's own entry points (its slices depend on each other across pixels).
#include "vpt_internal.h"
#include "vpt_kernels_iso_depth.h"

// the tap windows as dword-aligned 12-byte loads + v_alignbyte (VPT_V_ALIGNED), as MIP / EAM: round 4, after the aligned form stopped going
// through flat_load (vpt_device.h aligned_taps) — 256^3 at 1080p, us per frame: ISO 56.4 -> 43.0, Depth 56.6 -> 46.8, LAO 1963 -> 1675
#ifndef VPT_EXTRA_TAPS
#define VPT_EXTRA_TAPS VPT_V_ALIGNED
#endif
#ifndef VPT_DOS_TAPS
#define VPT_DOS_TAPS VPT_V_ALIGNED     // (the DOS sweep: 2.144 -> 2.107 ms; MCS, whose samples are not coherent: 16.2 -> 16.2, stays unaligned)
#endif
#define K_ISO0(V) (k_iso<0, V | VPT_EXTRA_TAPS>)
#define K_ISO1(V) (k_iso<1, V | VPT_EXTRA_TAPS>)
#define K_ISOR(V) (k_iso_render<V>)
#define K_DEPTH0(V) (k_depth<0, V | VPT_EXTRA_TAPS>)
#define K_DEPTH1(V) (k_depth<1, V | VPT_EXTRA_TAPS>)
#define K_LAO0(V) (k_lao<0, V | VPT_EXTRA_TAPS>)
#define K_LAO1(V) (k_lao<1, V | VPT_EXTRA_TAPS>)
#define LAUNCH(kernel, r, a, lds) hipLaunchKernelGGL(kernel, tile_grid(r), dim3(VPT_BLOCK), (lds), (r)->ctx->stream, (a))

int extra_reset(vpt_renderer *r, const PassArgs &a) {
    switch (r->kind) {
        case VPT_RENDERER_ISO: LAUNCH(k_iso_reset, r, a, 0); break;
        case VPT_RENDERER_DEPTH: LAUNCH(k_depth_reset, r, a, 0); break;
        case VPT_RENDERER_LAO: LAUNCH(k_lao_reset, r, a, 0); break;           // LAORenderer.glsl:285-287: (0, 0, 0, 1) into RGBA8
        default: LAUNCH(k_dos_reset, r, a, 0); r->dos_rect_valid = false; break;
    }
    return VPT_OK;
}
int extra_generate(vpt_renderer *r, const PassArgs &a) {
    switch (r->kind) {
        case VPT_RENDERER_ISO: LAUNCH_S(K_ISO0, r, a); break;
        case VPT_RENDERER_DEPTH: LAUNCH_S(K_DEPTH0, r, a); break;
        case VPT_RENDERER_LAO: LAUNCH_S(K_LAO0, r, a); break;
        default: break;                                                        // DOSRenderer.js has no _generateFrame
    }
    return VPT_OK;
}
int extra_integrate(vpt_renderer *r, const PassArgs &a) {
    switch (r->kind) {
        case VPT_RENDERER_ISO: LAUNCH(k_iso_integrate, r, a, 0); break;
        case VPT_RENDERER_DEPTH: LAUNCH(k_depth_integrate, r, a, 0); break;
        case VPT_RENDERER_LAO: LAUNCH(k_lao_integrate, r, a, 0); break;
        default: break;
    }
    return VPT_OK;
}
int extra_render_frame(vpt_renderer *r, const PassArgs &a) {
    switch (r->kind) {
        case VPT_RENDERER_ISO: LAUNCH_S(K_ISOR, r, a); break;
        case VPT_RENDERER_DEPTH: LAUNCH(k_depth_render, r, a, 0); break;
        case VPT_RENDERER_LAO: LAUNCH(k_lao_render, r, a, 0); break;           // LAORenderer.glsl:259-261
        default: LAUNCH(k_dos_render, r, a, 0); break;
    }
    return VPT_OK;
}
int extra_fused(vpt_renderer *r, const PassArgs &a) {
    switch (r->kind) {
        case VPT_RENDERER_ISO: LAUNCH_S(K_ISO1, r, a); break;
        case VPT_RENDERER_DEPTH: LAUNCH_S(K_DEPTH1, r, a); break;
        case VPT_RENDERER_LAO: LAUNCH_S(K_LAO1, r, a); break;
        default: return fail(VPT_ERR_UNSUPPORTED, "the DOS renderer has no single-launch render()");
    }
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
        case 0: return launch_dos_slice(k_dos_slice<0 | VPT_DOS_TAPS>, r, a, rect);
        case 1: return launch_dos_slice(k_dos_slice<1 | VPT_DOS_TAPS>, r, a, rect);
        case 2: return launch_dos_slice(k_dos_slice<2 | VPT_DOS_TAPS>, r, a, rect);
        case 3: return launch_dos_slice(k_dos_slice<3 | VPT_DOS_TAPS>, r, a, rect);
        case 8: return launch_dos_slice(k_dos_slice<8 | VPT_DOS_TAPS>, r, a, rect);
        case 9: return launch_dos_slice(k_dos_slice<9 | VPT_DOS_TAPS>, r, a, rect);
        case 10: return launch_dos_slice(k_dos_slice<10 | VPT_DOS_TAPS>, r, a, rect);
        case 11: return launch_dos_slice(k_dos_slice<11 | VPT_DOS_TAPS>, r, a, rect);
        case 32: return launch_dos_slice(k_dos_slice<32 | VPT_DOS_TAPS>, r, a, rect);
        case 33: return launch_dos_slice(k_dos_slice<33 | VPT_DOS_TAPS>, r, a, rect);
        case 34: return launch_dos_slice(k_dos_slice<34 | VPT_DOS_TAPS>, r, a, rect);
        case 35: return launch_dos_slice(k_dos_slice<35 | VPT_DOS_TAPS>, r, a, rect);
        case 40: return launch_dos_slice(k_dos_slice<40 | VPT_DOS_TAPS>, r, a, rect);
        case 41: return launch_dos_slice(k_dos_slice<41 | VPT_DOS_TAPS>, r, a, rect);
        case 42: return launch_dos_slice(k_dos_slice<42 | VPT_DOS_TAPS>, r, a, rect);
        default: return launch_dos_slice(k_dos_slice<43 | VPT_DOS_TAPS>, r, a, rect);
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
