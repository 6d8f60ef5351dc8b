// mm_pixel_bwd.hip -- pixel-major pass of the render path's backward for gfx950 (see mm_backward.hip for the scheme, mm_backward.h for why
// this half is compiled with the forward's floating-point flags).
#include "mm_pixel_pass.h"

MM_TIMELINE_STORAGE(pixel_bwd)
MM_PP_STORAGE(pixel_bwd)        // 0 loss totals + g4, 1 shading recompute + stores, 2 record append, 3 dlights reduction

namespace mm {

// ---------------------------------------------------------------------------------------------------------------------
// 1. pixel-major pass (the per-wave function: mm_pixel_pass.h)
// ---------------------------------------------------------------------------------------------------------------------
#ifndef MM_PIXEL_LB
#define MM_PIXEL_LB 5             // waves per SIMD the register allocation is held to: 96 VGPRs without spills (the light gradients are carried as scalar + normal, not
#endif                            // as nine products); 5 workgroups of 28.9 KB LDS (the plan workgroups' staging) fit a CU as well
template <bool kNoMask>
__global__ __launch_bounds__(256, MM_PIXEL_LB) void pixel_bwd_kernel(BwdArgs a) {
    MM_TIMELINE_BEGIN();
    __shared__ __attribute__((aligned(16))) unsigned char s_plan[MM_PLAN_LDS_BYTES];
    __shared__ float s_gm[MM_BLOCK_WAVES][2];
    if ((int)blockIdx.x < a.plan_wgs * a.B) { plan_sweep_items(a, blockIdx.x / a.plan_wgs, blockIdx.x % a.plan_wgs, s_plan); return; }   // (workgroup-uniform)
    int b, blk;
    map_block(blockIdx.x - a.plan_wgs * a.B, a.B, a.blocks_per_image, b, blk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    PixelWaveOut o;
    pixel_backward_wave<kNoMask>(a, b, blk, wave, lane, MM_HF_LOAD, o);
    if (lane == 0) { s_gm[wave][0] = o.m2; s_gm[wave][1] = o.m4; }
    __syncthreads();
    if (threadIdx.x >= 64 && threadIdx.x < 66) {                 // non-negative floats order like their bit patterns: integer max, one atomic per
        const int k = threadIdx.x - 64;                          // workgroup and kind (NaN / inf gradients end up as an inf scale = zero sums)
        const float m = fmaxf(fmaxf(s_gm[0][k], s_gm[1][k]), fmaxf(s_gm[2][k], s_gm[3][k]));
        // one atomic per workgroup and kind, spread over MM_GSHARD words per image on separate 32-byte sectors (thousands of
        // workgroups per image on ONE word queue at the memory side: measured +75 % on this kernel at 512x512)
        if (m > 0.f) atomicMax(a.gmax + ((size_t)b * MM_GSHARD + (blk & (MM_GSHARD - 1))) * 8 + k, __float_as_uint(m));
    }
    MM_TIMELINE_END(pixel_bwd);
}

int launch_pixel_bwd(const BwdArgs& a, const MMRenderDesc* d, hipStream_t s) {
    ProfScope p(d->prof_events, MM_PROF_PIXEL_BWD, s);
    dim3 grid(a.blocks_per_image * d->B + a.plan_wgs * d->B);    // + the plan workgroups, in front
    if (d->no_mask) hipLaunchKernelGGL(pixel_bwd_kernel<true>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(pixel_bwd_kernel<false>, grid, dim3(256), 0, s, a);
    return MM_OK;
}

}  // namespace mm
