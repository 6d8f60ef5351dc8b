// mm_loss.hip -- DiffRender.recon_data (/root/reference/networks.py:364-390) for gfx950: masked L1 + kaolin mask_iou
// (+ optional contour term), forward and backward.  SURVEY.md 8(a) rows a13/a14, gradient Appendix A.4.
//
// Forward: grid (chunks, B); every workgroup reduces its pixels of one image to four partial sums
//   {sum|pi-gi|, sum p*g, sum p+g-p*g, sum (c(p)-c(g))^2} written to the workspace (no atomics, no memset), then a
//   single small workgroup folds the partials in a fixed order into per-image totals and the scalar loss.
// Backward: one thread per pixel, reading the per-image totals.
#include "mm_device.h"

namespace mm {

#define MM_LOSS_CHUNKS 8

struct LossArgs {
    int B, H, W;
    const float* pred; long long ps[4];
    const float* gt;
    float image_weight, contour;
    float* partial;          // (B, MM_LOSS_CHUNKS, 4)
    float* totals;           // (B, 4)
    float* loss;
    const float* grad_loss;
    float* grad_pred;
};

__device__ inline int near_src(int dst, int in, int out) {   // torch 'nearest' interpolate index
    const float scale = (float)in / (float)out;
    const int s = (int)floorf((float)dst * scale);
    return s < in - 1 ? s : in - 1;
}

__device__ inline void contour_src(int y, int x, int H, int W, int& ys, int& xs) {
    const int h4 = H / 4, w4 = W / 4;
    ys = near_src(near_src(y, h4, H), H, h4);
    xs = near_src(near_src(x, w4, W), W, w4);
}

__global__ __launch_bounds__(256) void recon_partial_kernel(LossArgs a) {
    __shared__ float s_red[4][4];
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int hw = a.H * a.W;
    const float* gt = a.gt + (size_t)b * 4 * hw;
    float l1 = 0.f, up = 0.f, down = 0.f, cs = 0.f;
    for (int i = chunk * 256 + tid; i < hw; i += MM_LOSS_CHUNKS * 256) {
        const int y = i / a.W, x = i - y * a.W;
        const size_t po = (size_t)(a.ps[0] * b + a.ps[2] * y + a.ps[3] * x);
        const float gm = gt[3 * hw + i], pm = a.pred[po + a.ps[1] * 3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float gi = gt[c * hw + i] * gm + 1.f * (1.f - gm);
            const float pi = a.pred[po + a.ps[1] * c] * gm + 1.f * (1.f - gm);
            l1 += fabsf(pi - gi);
        }
        const float mul = pm * gm;
        up += mul; down += (pm + gm) - mul;
        if (a.contour > 0.f) {
            int ys, xs;
            contour_src(y, x, a.H, a.W, ys, xs);
            const float cp = fabsf(pm - a.pred[(size_t)(a.ps[0] * b + a.ps[1] * 3 + a.ps[2] * ys + a.ps[3] * xs)]);
            const float cg = fabsf(gm - gt[3 * hw + ys * a.W + xs]);
            cs += (cp - cg) * (cp - cg);
        }
    }
    l1 = wave_sum(l1); up = wave_sum(up); down = wave_sum(down); cs = wave_sum(cs);
    if ((tid & 63) == 0) { s_red[tid >> 6][0] = l1; s_red[tid >> 6][1] = up; s_red[tid >> 6][2] = down; s_red[tid >> 6][3] = cs; }
    __syncthreads();
    if (tid < 4) a.partial[((size_t)b * MM_LOSS_CHUNKS + chunk) * 4 + tid] = ((s_red[0][tid] + s_red[1][tid]) + s_red[2][tid]) + s_red[3][tid];
}

__global__ __launch_bounds__(64) void recon_final_kernel(LossArgs a) {
    const int lane = threadIdx.x;
    float l1 = 0.f, iou = 0.f, cs = 0.f;
    for (int b = lane; b < a.B; b += 64) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < MM_LOSS_CHUNKS; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] += a.partial[((size_t)b * MM_LOSS_CHUNKS + c) * 4 + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) a.totals[b * 4 + k] = t[k];
        l1 += t[0];
        iou += t[1] / (t[2] + 1e-10f);
        cs += t[3];
    }
    l1 = wave_sum(l1); iou = wave_sum(iou); cs = wave_sum(cs);
    if (lane == 0) {
        const float cnt = (float)a.B * 3.f * (float)a.H * (float)a.W;
        float loss_mask = 1.f - iou / (float)a.B;
        if (a.contour > 0.f) loss_mask += (cs / ((float)a.B * (float)a.H * (float)a.W)) * a.contour;
        a.loss[0] = a.image_weight * (l1 / cnt) + 1.f * loss_mask;
    }
}

__global__ __launch_bounds__(256) void recon_bwd_kernel(LossArgs a) {
    const int b = blockIdx.y;
    const int hw = a.H * a.W;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= hw) return;
    const float gs = a.grad_loss ? a.grad_loss[0] : 1.f;
    const float* gt = a.gt + (size_t)b * 4 * hw;
    const int y = i / a.W, x = i - y * a.W;
    const size_t po = (size_t)(a.ps[0] * b + a.ps[2] * y + a.ps[3] * x);
    const float cnt = (float)a.B * 3.f * (float)a.H * (float)a.W;
    const float gm = gt[3 * hw + i];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float gi = gt[c * hw + i] * gm + 1.f * (1.f - gm);
        const float pi = a.pred[po + a.ps[1] * c] * gm + 1.f * (1.f - gm);
        const float df = pi - gi, sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
        a.grad_pred[po + a.ps[1] * c] = gs * a.image_weight * sg * gm / cnt;
    }
    const float up = a.totals[b * 4 + 1], U = a.totals[b * 4 + 2] + 1e-10f;
    a.grad_pred[po + a.ps[1] * 3] = gs * (-(1.f / (float)a.B) * (gm / U - up * (1.f - gm) / (U * U)));
}

// contour term (networks.py:379-387): runs after recon_bwd_kernel, adds into the alpha channel of grad_pred.
// The term couples every pixel p with the pixel s(p) that the two nearest-neighbour resamplings select for it: p receives +g_p, s(p) receives
// -g_p.  No atomics: one thread per pixel forms EVERYTHING its pixel receives -- its own g, and, if it is the selected pixel of a cell of the
// (H/4, W/4) grid, minus the g of every pixel of that cell, visited in row-major order -- and adds it to the gradient with one plain
// read-modify-write (nobody else writes that element in this launch).  Bitwise reproducible; the result is the sum the reference's autograd
// forms, in a fixed order.
__device__ inline float contour_pixel_grad(const LossArgs& a, const float* gt, int b, int y, int x, float k2) {
    int ys, xs;
    contour_src(y, x, a.H, a.W, ys, xs);
    const size_t hw = (size_t)a.H * a.W;
    const float d = a.pred[(size_t)(a.ps[0] * b + a.ps[1] * 3 + a.ps[2] * y + a.ps[3] * x)] -
                    a.pred[(size_t)(a.ps[0] * b + a.ps[1] * 3 + a.ps[2] * ys + a.ps[3] * xs)];
    const float cp = fabsf(d), cg = fabsf(gt[3 * hw + (size_t)y * a.W + x] - gt[3 * hw + (size_t)ys * a.W + xs]);
    const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    return k2 * (cp - cg) * sg;
}
// the cell of the (in/4) grid whose selected pixel is `v`, or -1: near_src(c, in, in4) == v for at most one c (the map is strictly increasing)
__device__ inline int contour_cell_of_source(int v, int in, int in4) {
    const int c0 = (int)floorf((float)v * ((float)in4 / (float)in));
    for (int c = max(c0 - 1, 0); c <= min(c0 + 2, in4 - 1); ++c)
        if (near_src(c, in, in4) == v) return c;
    return -1;
}
__global__ __launch_bounds__(256) void recon_contour_bwd_kernel(LossArgs a) {
    const int b = blockIdx.y;
    const int hw = a.H * a.W;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= hw) return;
    const float gs = a.grad_loss ? a.grad_loss[0] : 1.f;
    const float* gt = a.gt + (size_t)b * 4 * hw;
    const int y = i / a.W, x = i - y * a.W;
    const float n = (float)a.B * (float)a.H * (float)a.W;
    const float k2 = gs * a.contour * 2.f / n;
    float acc = contour_pixel_grad(a, gt, b, y, x, k2);
    const int h4 = a.H / 4, w4 = a.W / 4;
    const int cy = contour_cell_of_source(y, a.H, h4), cx = contour_cell_of_source(x, a.W, w4);
    if (cy >= 0 && cx >= 0) {
        // the cell's pixels: rows / columns r with near_src(r, in4, in) == c -- a contiguous run around c * in / in4
        const int y0 = max((int)floorf((float)cy * ((float)a.H / (float)h4)) - 2, 0), y1 = min((int)ceilf((float)(cy + 1) * ((float)a.H / (float)h4)) + 2, a.H);
        const int x0 = max((int)floorf((float)cx * ((float)a.W / (float)w4)) - 2, 0), x1 = min((int)ceilf((float)(cx + 1) * ((float)a.W / (float)w4)) + 2, a.W);
        float minus = 0.f;
        for (int yy = y0; yy < y1; ++yy) {
            if (near_src(yy, h4, a.H) != cy) continue;
            for (int xx = x0; xx < x1; ++xx) {
                if (near_src(xx, w4, a.W) != cx) continue;
                minus += contour_pixel_grad(a, gt, b, yy, xx, k2);
            }
        }
        acc -= minus;
    }
    if (acc != 0.f) a.grad_pred[(size_t)(a.ps[0] * b + a.ps[1] * 3 + a.ps[2] * y + a.ps[3] * x)] += acc;
}

static LossArgs make_loss_args(const MMReconDesc* d) {
    LossArgs a;
    a.B = d->B; a.H = d->H; a.W = d->W;
    a.pred = d->pred;
    for (int i = 0; i < 4; ++i) a.ps[i] = d->pred_strides[i];
    a.gt = d->gt; a.image_weight = d->image_weight; a.contour = d->contour;
    a.partial = (float*)d->workspace;
    a.totals = a.partial + (size_t)d->B * MM_LOSS_CHUNKS * 4;
    a.loss = d->loss; a.grad_loss = d->grad_loss; a.grad_pred = d->grad_pred;
    return a;
}

size_t recon_workspace_bytes(const MMReconDesc* d) {
    return align256(((size_t)d->B * MM_LOSS_CHUNKS * 4 + (size_t)d->B * 4) * sizeof(float));
}

const float* recon_totals(const MMReconDesc* d) { return (const float*)d->workspace + (size_t)d->B * MM_LOSS_CHUNKS * 4; }   // (make_loss_args' `totals`)

int launch_recon_fwd(const MMReconDesc* d, hipStream_t s) {
    LossArgs a = make_loss_args(d);
    { ProfScope p(d->prof_events, MM_PROF_RECON_PARTIAL, s);
      hipLaunchKernelGGL(recon_partial_kernel, dim3(MM_LOSS_CHUNKS, d->B), dim3(256), 0, s, a); }
    { ProfScope p(d->prof_events, MM_PROF_RECON_FINAL, s);
      hipLaunchKernelGGL(recon_final_kernel, dim3(1), dim3(64), 0, s, a); }
    return launch_ok("recon_fwd");
}

int launch_recon_bwd(const MMReconDesc* d, hipStream_t s) {
    LossArgs a = make_loss_args(d);
    dim3 grid((d->H * d->W + 255) / 256, d->B);
    { ProfScope p(d->prof_events, MM_PROF_RECON_BWD, s);
      hipLaunchKernelGGL(recon_bwd_kernel, grid, dim3(256), 0, s, a); }
    if (d->contour > 0.f) {
        ProfScope p(d->prof_events, MM_PROF_RECON_CONTOUR, s);
        hipLaunchKernelGGL(recon_contour_bwd_kernel, grid, dim3(256), 0, s, a);
    }
    return launch_ok("recon_bwd");
}

}  // namespace mm
