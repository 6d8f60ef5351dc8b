// mm_texflow.hip -- texture-flow sampling of the texture encoder for gfx950.
//
// Replaces the tail of TextureEncoder.forward (/root/reference/network/model_res.py:597-612, makeup == 0):
//   uv_sampler = texture_flow.permute(0, 2, 3, 1)
//   textures   = F.grid_sample(img, uv_sampler, mode='bicubic', align_corners=True)          # zeros padding
//   textures   = torch.cat([textures, textures.flip([2])], dim=2)                              # back = mirrored front
// i.e. the step that produces the (B,3,2H,W) texture the render path consumes.  One thread per sampled texel: 16 taps x C
// channels, both mirrored rows written from registers (the flipped copy is never read back); the backward folds the two
// rows' gradients, differentiates the cubic weights analytically for the flow and scatters to the image only if asked.
// Bicubic convention = ATen's (A = -0.75, taps outside the image read 0, align_corners=True: ix = (gx + 1)/2 * (W - 1)).
#include "mm_device.h"

namespace mm {

struct TexFlowArgs {
    int B, C, H, W, Ho, Wo;
    const float* image;      // (B,C,H,W)
    const float* flow;       // (B,2,Ho,Wo): channel 0 = x, 1 = y in [-1,1]
    float* textures;         // (B,C,2Ho,Wo)
    const float* g_tex;      // (B,C,2Ho,Wo)
    float* g_flow;           // (B,2,Ho,Wo)
    float* g_image;          // (B,C,H,W) or null
};

#define MM_CUBIC_A (-0.75f)

__device__ inline float cubic_w1(float x) { return ((MM_CUBIC_A + 2.f) * x - (MM_CUBIC_A + 3.f)) * x * x + 1.f; }                 // |x| <= 1
__device__ inline float cubic_w2(float x) { return ((MM_CUBIC_A * x - 5.f * MM_CUBIC_A) * x + 8.f * MM_CUBIC_A) * x - 4.f * MM_CUBIC_A; }   // 1 < |x| < 2
__device__ inline float cubic_d1(float x) { return (3.f * (MM_CUBIC_A + 2.f) * x - 2.f * (MM_CUBIC_A + 3.f)) * x; }
__device__ inline float cubic_d2(float x) { return (3.f * MM_CUBIC_A * x - 10.f * MM_CUBIC_A) * x + 8.f * MM_CUBIC_A; }

__device__ inline void cubic_coeffs(float t, float* c) {
    c[0] = cubic_w2(t + 1.f); c[1] = cubic_w1(t); c[2] = cubic_w1(1.f - t); c[3] = cubic_w2(2.f - t);
}
__device__ inline void cubic_coeffs_grad(float t, float* d) {      // d c[k] / d t
    d[0] = cubic_d2(t + 1.f); d[1] = cubic_d1(t); d[2] = -cubic_d1(1.f - t); d[3] = -cubic_d2(2.f - t);
}

struct Taps { int x0, y0; float cx[4], cy[4], tx, ty; };

__device__ inline Taps flow_taps(const TexFlowArgs& a, int b, int oy, int ox) {
    const size_t plane = (size_t)a.Ho * a.Wo;
    const float gx = a.flow[((size_t)b * 2 + 0) * plane + (size_t)oy * a.Wo + ox];
    const float gy = a.flow[((size_t)b * 2 + 1) * plane + (size_t)oy * a.Wo + ox];
    const float ix = ((gx + 1.f) / 2.f) * (float)(a.W - 1), iy = ((gy + 1.f) / 2.f) * (float)(a.H - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    Taps t;
    t.tx = ix - fx; t.ty = iy - fy;
    // saturating float -> int conversion keeps wild (inf / huge) coordinates out of bounds; NaN taps read 0 like ATen's bounds test
    t.x0 = (fx >= -2e9f && fx <= 2e9f) ? (int)fx - 1 : INT_MIN / 2;
    t.y0 = (fy >= -2e9f && fy <= 2e9f) ? (int)fy - 1 : INT_MIN / 2;
    cubic_coeffs(t.tx, t.cx); cubic_coeffs(t.ty, t.cy);
    return t;
}

// The sixteen taps of a channel are loaded UNCONDITIONALLY from clamped (always valid) addresses and masked afterwards: a tap inside a
// per-lane `if (in bounds)` is a branch + wait of its own, i.e. up to 48 dependent trips to memory per texel instead of one.
__device__ inline void tap_window(const TexFlowArgs& a, const Taps& t, int (&xs)[4], int (&ys)[4], bool (&inx)[4], bool (&iny)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = t.x0 + j, y = t.y0 + j;
        inx[j] = x >= 0 && x < a.W; iny[j] = y >= 0 && y < a.H;
        xs[j] = min(max(x, 0), a.W - 1); ys[j] = min(max(y, 0), a.H - 1);
    }
}

// the four x taps of one row: ONE 16-byte load (4-byte aligned: gfx950 global loads need no more) of the window clamped into the row, the
// taps picked out of it; a tap outside the image is masked by the caller.  Rows narrower than four texels take four scalar loads.
struct __attribute__((packed, aligned(4))) Row4 { float v[4]; };
__device__ inline void load_row(const float* row, int W, int x0, const int (&xs)[4], float (&v)[4]) {
    if (W >= 4) {
        const int xw = min(max(x0, 0), W - 4), sh = x0 - xw;     // tap j sits at window slot j + sh (in 0..3 whenever the tap is inside)
        const Row4 r = *(const Row4*)(row + xw);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = j + sh;
            v[j] = k <= 0 ? r.v[0] : (k == 1 ? r.v[1] : (k == 2 ? r.v[2] : r.v[3]));
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = row[xs[j]];
    }
}

__global__ __launch_bounds__(256) void texflow_fwd_kernel(TexFlowArgs a) {
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63), oy = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    if (ox >= a.Wo || oy >= a.Ho) return;
    const Taps t = flow_taps(a, b, oy, ox);
    int xs[4], ys[4];
    bool inx[4], iny[4];
    tap_window(a, t, xs, ys, inx, iny);
    const size_t oplane = (size_t)2 * a.Ho * a.Wo;
    for (int c = 0; c < a.C; ++c) {
        const float* img = a.image + ((size_t)b * a.C + c) * a.H * a.W;
        float v[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) load_row(img + (size_t)ys[i] * a.W, a.W, t.x0, xs, v[i]);
        float rows[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = (inx[j] && iny[i]) ? v[i][j] : 0.f;
            rows[i] = ((w[0] * t.cx[0] + w[1] * t.cx[1]) + w[2] * t.cx[2]) + w[3] * t.cx[3];
        }
        const float out = ((rows[0] * t.cy[0] + rows[1] * t.cy[1]) + rows[2] * t.cy[2]) + rows[3] * t.cy[3];
        float* o = a.textures + ((size_t)b * a.C + c) * oplane;
        o[(size_t)oy * a.Wo + ox] = out;
        o[(size_t)(2 * a.Ho - 1 - oy) * a.Wo + ox] = out;         // textures.flip([2]) half
    }
}

__global__ __launch_bounds__(256) void texflow_bwd_kernel(TexFlowArgs a) {
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63), oy = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    if (ox >= a.Wo || oy >= a.Ho) return;
    const Taps t = flow_taps(a, b, oy, ox);
    float dx[4], dy[4];
    cubic_coeffs_grad(t.tx, dx); cubic_coeffs_grad(t.ty, dy);
    const size_t oplane = (size_t)2 * a.Ho * a.Wo;
    float gix = 0.f, giy = 0.f;
    int xs[4], ys[4];
    bool inx[4], iny[4];
    tap_window(a, t, xs, ys, inx, iny);
    for (int c = 0; c < a.C; ++c) {
        const float* g = a.g_tex + ((size_t)b * a.C + c) * oplane;
        const float go = g[(size_t)oy * a.Wo + ox] + g[(size_t)(2 * a.Ho - 1 - oy) * a.Wo + ox];      // both mirrored rows
        const float* img = a.image + ((size_t)b * a.C + c) * a.H * a.W;
        float* gi = a.g_image ? a.g_image + ((size_t)b * a.C + c) * a.H * a.W : nullptr;
        float v[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) load_row(img + (size_t)ys[i] * a.W, a.W, t.x0, xs, v[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (inx[j] && iny[i]) {
                    gix += go * v[i][j] * (dx[j] * t.cy[i]);
                    giy += go * v[i][j] * (t.cx[j] * dy[i]);
                    if (gi) atomicAdd(gi + (size_t)ys[i] * a.W + xs[j], go * (t.cx[j] * t.cy[i]));
                }
            }
        }
    }
    const size_t plane = (size_t)a.Ho * a.Wo;
    a.g_flow[((size_t)b * 2 + 0) * plane + (size_t)oy * a.Wo + ox] = gix * ((float)(a.W - 1) / 2.f);
    a.g_flow[((size_t)b * 2 + 1) * plane + (size_t)oy * a.Wo + ox] = giy * ((float)(a.H - 1) / 2.f);
}

static TexFlowArgs texflow_args(const MMTexFlowDesc* d) {
    TexFlowArgs a;
    a.B = d->B; a.C = d->C; a.H = d->H; a.W = d->W; a.Ho = d->Ho; a.Wo = d->Wo;
    a.image = d->image; a.flow = d->flow; a.textures = d->textures;
    a.g_tex = nullptr; a.g_flow = nullptr; a.g_image = nullptr;
    return a;
}

int launch_texflow_fwd(const MMTexFlowDesc* d, hipStream_t s) {
    const TexFlowArgs a = texflow_args(d);
    hipLaunchKernelGGL(texflow_fwd_kernel, dim3((d->Wo + 63) / 64, (d->Ho + 3) / 4, d->B), dim3(256), 0, s, a);
    return launch_ok("texflow_fwd");
}

int launch_texflow_bwd(const MMTexFlowDesc* d, const MMTexFlowGrads* g, hipStream_t s) {
    TexFlowArgs a = texflow_args(d);
    a.g_tex = g->grad_textures; a.g_flow = g->grad_flow; a.g_image = g->grad_image;
    if (g->grad_image && hipMemsetAsync(g->grad_image, 0, (size_t)d->B * d->C * d->H * d->W * sizeof(float), s) != hipSuccess) {
        last_launch_error() = {hipGetLastError(), "texflow_bwd memset"};
        return MM_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(texflow_bwd_kernel, dim3((d->Wo + 63) / 64, (d->Ho + 3) / 4, d->B), dim3(256), 0, s, a);
    return launch_ok("texflow_bwd");
}

}  // namespace mm
