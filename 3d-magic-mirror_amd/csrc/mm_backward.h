// mm_backward.h -- arguments shared by the two translation units of the pixel-stage backward:
//   mm_pixel_bwd.hip  the pixel-major pass (+ the sweep plan in its grid), compiled like the FORWARD (no contraction, IEEE division): it
//                     recomputes the forward's per-pixel quantities, and a recomputation that rounds differently picks the other bilinear
//                     cell / the other side of torch.clamp for the rare pixel that sits within an ulp of a cell border or of 0 / 1 -- a
//                     different (equally valid, but not the reference's) one-sided derivative there, 1e-3 on a vertex gradient when it happens
//                     (found by profiles/tools/fuzz_parity.py; with identical flags the recomputation is bit-identical and the event is gone)
//   mm_backward.hip   the gathers, held to 1e-4 and bound by instruction issue: fma contraction + 2.5-ulp division (a third fewer instructions)
#pragma once
#include "mm_device.h"

namespace mm {

struct BwdArgs {
    int B, H, W, F, Ht, Wt, knum, blocks_x, blocks_per_image, options;
    float mult, eps, sigmainv, infl;
    float kx, ky, sig2;                                          // multiplier / W, multiplier / H, sigmainv / multiplier^2, formed on the host (IEEE): see soft_factor
    const float4* geo;
    const float* face_uvs;
    const float* fn;
    const float* textures;
    const float* lights;
    const float* bg;
    const int32_t* face_idx;
    const float2* soft;
    const int* fflag;                                            // (B,F,2) {owns a pixel, is in a silhouette product} (raster_fwd): faces with neither get no sweep
                                                                 // items, owners-only are swept over their box without the silhouette margin (sweep_box)
    int sweep_sx, sweep_sy;                                      // that margin in whole pixels, per axis (sweep_shrink)
    const float* grad_rgba;
    float4* gp; float* gp2;
    float* dl_part;
    float* grad_bg;
    unsigned* ticket;
    int* tcur; int* tdrop; const int* trcnt; int* toff; int* tstatus; TexRecord* trec; int ntiles_, trcap;   // texture records (Workspace)
    int* status_flag;                                            // MMRenderDesc.status_flag (may be pinned host memory) or nullptr
    // fused recon_data (gt == nullptr: off)
    const float* gt; const float* rgba; const float* grad_loss; float* loss; float image_weight, contour;
    const long long* ltot;                                       // (B,MM_LSUB,4) fused loss sums of the raster waves (fixed point)
                                                                 // DEFERRED fusion (MMRenderDesc.fused_totals; options & MM_INT_DEFERRED): the SAME field carries
                                                                 // mm_recon_data_forward's per-image totals (B,4) floats instead (deferred_totals below) -- one more
                                                                 // pointer in this struct costs the default pixel kernel eight scalar-register spills
    // gather
    unsigned* gmax;                                              // (B,2) per image: max |K2 number| and max |dL/dalpha| as float bits (pixel pass -> gather)
    const int2* items; const int2* nitems; float* part; int item_cap;   // sweep items {face, chunk} of the plan workgroups; their partial sums
    int2* plan_chunkmap; int2* plan_items; int2* plan_nitems; int plan_wgs;           // ... as the plan workgroups (first B of pixel_bwd's grid) write them
    int ntx, nty;
    float* grad_textures;
};

#define MM_INT_DEFERRED (1 << 30)   // BwdArgs::options, set by launch_raster_bwd (not an MM_OPT_* bit of the ABI)
__device__ inline const float* deferred_totals(const BwdArgs& a) { return reinterpret_cast<const float*>(a.ltot); }

__device__ inline void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#define MM_PLAN_WGS 4              // plan workgroups per image where faces are many (else one), see mm_pixel_bwd.hip
int launch_pixel_bwd(const BwdArgs& a, const MMRenderDesc* d, hipStream_t s);     // mm_pixel_bwd.hip

}  // namespace mm
