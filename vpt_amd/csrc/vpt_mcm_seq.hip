// vpt_mcm_seq.hip — MCM frame sequences in one launch (vpt_renderer_play with VPT_PLAY_FUSED / VPT_PLAY_FRAMES): k_mcm_multi and
// k_mcm_frames (vpt_kernels_mcm.h) keep the photon state in registers over the passes.  AbstractRenderer.js:60-70 called n times.
#include "vpt_mcm_select.h"
#include "vpt_kernels_mcm.h"

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
int mcm_multi(vpt_renderer *r, const PassArgs &a, uint32_t npasses, uint2 *ring) {
    r->tm_valid = false;
    VPT_TRY(mcm_before_pass(r, a, nullptr));
    VPT_TRY(mcm_materialize(r));                      // a whole-image kernel: every tile's full photon state
    if (r->side_busy) VPT_TRY(join_side(r));
    if (a.vol.records) {                              // column records: LINEAR one-channel byte volumes
        const int v = class_variant(r, a);
        VARIANT_CASES(ring ? launch_frames(k_mcm_frames<V>, r, a, npasses, ring) : launch_multi(k_mcm_multi<V>, r, a, npasses))
    }
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

