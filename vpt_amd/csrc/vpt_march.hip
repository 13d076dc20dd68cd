// vpt_march.hip — the MIP, EAM and MCS renderers' passes (vpt_kernels_march.h) behind vpt_render.hip's entry points.
// MIPRenderer.js:69-100, EAMRenderer.js:88-153, MCSRenderer.js:74-140.
#include "vpt_internal.h"
#include "vpt_kernels_march.h"

// (dword-aligned 12-byte taps + v_alignbyte for MIP / EAM — re-measured in round 3 on the HIT tiles only, 256^3 1080p: EAM 63.7 us
// aligned against 72.3 unaligned on one stream, 52.1 / 63.3 on three; MIP 56.3 / 72.0, 45.1 / 64.1)
#define K_MIP0(V) (k_mip<0, V | VPT_V_ALIGNED>)
#define K_MIP1(V) (k_mip<1, V | VPT_V_ALIGNED>)
#define K_EAM0(V) (k_eam<0, V | VPT_V_ALIGNED>)
#define K_EAM1(V) (k_eam<1, V | VPT_V_ALIGNED>)
#ifndef VPT_MCS_TAPS
#define VPT_MCS_TAPS 0
#endif
#define K_MCS0(V) (k_mcs<0, V | VPT_MCS_TAPS>)
#define K_MCS1(V) (k_mcs<1, V | VPT_MCS_TAPS>)
#define LAUNCH(kernel, r, a, lds) hipLaunchKernelGGL(kernel, tile_grid(r), dim3(VPT_BLOCK), (lds), (r)->ctx->stream, (a))

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
#define LAUNCH_MCS_PERSIST(MODE, r, a) do { \
    int v_ = ((r)->vol->wide ? VPT_V_WIDE : 0) | ((r)->vol->filter == VPT_FILTER_NEAREST ? VPT_V_NEAREST : 0); \
    switch (v_) { \
        case 0: VPT_TRY(launch_mcs_persist((k_mcs_persist<MODE, 0>), (r), (a))); break; \
        case 1: VPT_TRY(launch_mcs_persist((k_mcs_persist<MODE, 1>), (r), (a))); break; \
        case 2: VPT_TRY(launch_mcs_persist((k_mcs_persist<MODE, 2>), (r), (a))); break; \
        default: VPT_TRY(launch_mcs_persist((k_mcs_persist<MODE, 3>), (r), (a))); break; \
    } } while (0)


int march_reset(vpt_renderer *r, const PassArgs &a) {
    switch (r->kind) {
        case VPT_RENDERER_MIP: LAUNCH(k_mip_reset, r, a, 0); break;
        case VPT_RENDERER_EAM: LAUNCH(k_eam_reset, r, a, 0); break;
        default: LAUNCH(k_mcs_reset, r, a, 0); break;
    }
    return VPT_OK;
}
static int launch_mcs(vpt_renderer *r, const PassArgs &a, bool fused) {
    if (r->mcs_persistent && r->vol->channels == 1 && !r->vol->f32) {       // (walks every tile)
        if (fused) LAUNCH_MCS_PERSIST(1, r, a); else LAUNCH_MCS_PERSIST(0, r, a);
        return VPT_OK;
    }
    if (fused) LAUNCH_S(K_MCS1, r, a); else LAUNCH_S(K_MCS0, r, a);
    return VPT_OK;
}
int march_generate(vpt_renderer *r, const PassArgs &a) {                      // _generateFrame
    switch (r->kind) {
        case VPT_RENDERER_MIP: LAUNCH_S(K_MIP0, r, a); break;
        case VPT_RENDERER_EAM: LAUNCH_S(K_EAM0, r, a); break;
        default: return launch_mcs(r, a, false);
    }
    return VPT_OK;
}
int march_integrate(vpt_renderer *r, const PassArgs &a) {                     // _integrateFrame
    switch (r->kind) {
        case VPT_RENDERER_MIP: LAUNCH(k_mip_integrate, r, a, 0); break;
        case VPT_RENDERER_EAM: LAUNCH(k_eam_integrate, r, a, 0); break;
        default: LAUNCH(k_mcs_integrate, r, a, 0); break;
    }
    return VPT_OK;
}
int march_render_frame(vpt_renderer *r, const PassArgs &a) {                  // _renderFrame
    switch (r->kind) {
        case VPT_RENDERER_MIP: LAUNCH(k_mip_render, r, a, 0); break;
        case VPT_RENDERER_EAM: LAUNCH(k_eam_render, r, a, 0); break;
        default: LAUNCH(k_mcs_render, r, a, 0); break;
    }
    return VPT_OK;
}
int march_fused(vpt_renderer *r, const PassArgs &a) {                         // render(): the three hooks in one launch
    switch (r->kind) {
        case VPT_RENDERER_MIP: LAUNCH_S(K_MIP1, r, a); break;
        case VPT_RENDERER_EAM: LAUNCH_S(K_EAM1, r, a); break;
        default: return launch_mcs(r, a, true);
    }
    return VPT_OK;
}
