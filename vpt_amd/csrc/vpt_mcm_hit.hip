// vpt_mcm_hit.hip — the MCM integrate kernels (k_mcm_integrate, k_mcm_integrate_early: vpt_kernels_mcm.h) in a translation unit of
// their own: every volume format x wide tables x fast-math x column records, with and without the fused _renderFrame.  vpt_mcm.hip asks for
// one by variant.  MCMRenderer.glsl:116-172.
#include "vpt_mcm_select.h"
#include "vpt_kernels_mcm.h"

#define K_MCM0(V) (k_mcm_integrate<false, V>)
#define K_MCM1(V) (k_mcm_integrate<true, V>)
#define K_MCM0F(V) (k_mcm_integrate<false, V | VPT_V_FAST>)
#define K_MCM1F(V) (k_mcm_integrate<true, V | VPT_V_FAST>)

template <bool FUSE> static PassKernel hit_kernel(int v, bool early) {
    if (early) VARIANT_CASES((PassKernel)k_mcm_integrate_early<FUSE, V>)
    VARIANT_CASES((PassKernel)k_mcm_integrate<FUSE, V>)
}
PassKernel mcm_hit_kernel(bool fuse, int v, bool early) { return fuse ? hit_kernel<true>(v, early) : hit_kernel<false>(v, early); }
template <bool FUSE> static PassKernel format_hit_kernel(int v, bool wide, bool fast) {
    if (wide) { if (fast) FORMAT_CASES((PassKernel)k_mcm_integrate<FUSE, F | VPT_V_WIDE | VPT_V_FAST>) FORMAT_CASES((PassKernel)k_mcm_integrate<FUSE, F | VPT_V_WIDE>) }
    if (fast) FORMAT_CASES((PassKernel)k_mcm_integrate<FUSE, F | VPT_V_FAST>)
    FORMAT_CASES((PassKernel)k_mcm_integrate<FUSE, F>)
}
PassKernel mcm_format_hit_kernel(bool fuse, int v, bool wide, bool fast) {
    return fuse ? format_hit_kernel<true>(v, wide, fast) : format_hit_kernel<false>(v, wide, fast);
}
// one pass over the whole image (no tile classes in force): the general kernel of the renderer's variant, split over the side streams
// like every sampling kernel (launch_sampling)
int mcm_general_pass(vpt_renderer *r, const PassArgs &a, bool fuse) {
    if (a.vol.records) {                                   // (LINEAR one-channel byte volume: variant_of is 0 or VPT_V_WIDE)
        const unsigned g_ = (unsigned)r->ntiles;
        return launch_sampling(mcm_hit_kernel(fuse, class_variant(r, a), false), r, a, g_);
    }
    if (r->fast_math) { if (fuse) LAUNCH_S(K_MCM1F, r, a); else LAUNCH_S(K_MCM0F, r, a); }
    else { if (fuse) LAUNCH_S(K_MCM1, r, a); else LAUNCH_S(K_MCM0, r, a); }
    return VPT_OK;
}
