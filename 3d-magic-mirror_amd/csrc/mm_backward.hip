// mm_backward.hip -- pixel-stage backward of the render path for gfx950, without a single global atomic.
//
// kaolin's backward kernels (rasterize_backward_cuda, dibr_soft_mask_backward_cuda) and torch's grid_sampler backward
// SCATTER per-pixel contributions with atomicAdd.  On MI355X an agent-scope float atomic is executed at the memory side
// of the fabric (the eight XCD L2s are not coherent with each other), ~17 G atomics/s measured, and this path would issue
// ~5 M of them per batch.  The backward is therefore organised as two GATHER passes (SURVEY.md Appendix A for the math):
//
//   1. pixel_bwd   pixel-major, one lane per pixel: re-shades the pixel, writes dL/dbg, reduces dL/dlights per
//                  workgroup (plain stores of partials), and leaves for every covered pixel the nine numbers the gather
//                  needs: d/d(texture sample rgb), d/d(mask), d/d(u,v), d/d(normal).
//   2a. texture_gather  one workgroup per (image, 32x32-texel texture tile), the tile's accumulators in LDS.  The faces
//                  that can sample the tile are a STATIC list (mm_build_uv_tiles).  16 lanes sweep each face's screen box;
//                  the pixels it owns add their bilinear footprint to the LDS tile (LDS float adds).  The tile is then
//                  written once with plain stores: no zero-fill pass over grad_textures.
//   2b. face_gather  16 lanes per (image, face) sweep the face's inflated screen box: pixels it owns give the K2
//                  barycentric gradient, uncovered pixels that hold the face among their first knum soft-mask faces give
//                  K4.  dL/d(face xy) and dL/d(face normal) are written once per face with plain stores.
#include "mm_device.h"

namespace mm {

struct BwdArgs {
    int B, H, W, F, Ht, Wt, knum, blocks_x, blocks_per_image;
    float mult, eps, sigmainv, infl;
    const float4* geo;
    const float* face_uvs;
    const float* fn;
    const float* textures;
    const float* lights;
    const float* bg;
    const int32_t* face_idx;
    const float* softq;
    const int* lastf;
    const float* grad_rgba;
    float4* gp0; float4* gp1; float* gp2;
    float* dl_part;
    float* grad_bg;
    float* dTacc; unsigned* ticket;
    // gather
    const int32_t* uvt_offsets; const int32_t* uvt_faces;
    int ntx, nty;
    float* grad_textures;
    float* dfxy;
    float* dfn;
};

// ---------------------------------------------------------------------------------------------------------------------
// 1. pixel-major pass
// ---------------------------------------------------------------------------------------------------------------------
template <bool kNoMask>
__global__ __launch_bounds__(256) void pixel_bwd_kernel(BwdArgs a) {
    __shared__ float s_dl[MM_BLOCK_WAVES][9];
    int b, blk;
    map_block(blockIdx.x, a.B, a.blocks_per_image, b, blk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bx = blk % a.blocks_x, by = blk / a.blocks_x;
    const int px = bx * MM_BLOCK_PX + (wave & 1) * MM_TILE + (lane & 7), py = by * MM_BLOCK_PX + (wave >> 1) * MM_TILE + (lane >> 3);
    const bool in_img = px < a.W && py < a.H;
    const float x0 = pixel_x(px, a.W, a.mult), y0 = pixel_y(py, a.H, a.mult);
    const size_t hw = (size_t)a.H * a.W, pin = (size_t)py * a.W + px;
    const size_t pix = (size_t)b * hw + pin;
    if (blk == 0) {                                              // accumulators of the vertex backward, used after this kernel
        if (threadIdx.x < 12) a.dTacc[b * 12 + threadIdx.x] = 0.f;
        if (threadIdx.x == 12) a.ticket[b] = 0u;
    }

    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
    int hf = -1;
    if (in_img) { g4 = *(const float4*)(a.grad_rgba + pix * 4); hf = a.face_idx[pix]; }
    const float gin[3] = {g4.x, g4.y, g4.z};
    float dl[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) dl[i] = 0.f;

    if (in_img && (hf >= 0 || kNoMask)) {
        // recompute the forward quantities of this pixel (only face_idx and the soft-mask state were saved)
        float w0 = 0.f, w1 = 0.f, w2 = 0.f, nrm = 1.f, m = 0.f, u = 0.f, v = 0.f, nx = 0.f, ny = 0.f, nz = 0.f;
        float fu[6] = {0, 0, 0, 0, 0, 0}, n0 = 0.f, n1 = 0.f, n2 = 0.f;
        if (hf >= 0) {
            const float4* geo = a.geo + ((size_t)b * a.F + hf) * 3;
            const float4 p0 = geo[0], p1 = geo[1];
            edge_weights(p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, x0, y0, a.eps, w0, w1, w2, nrm);
            w0 /= nrm; w1 /= nrm; w2 /= nrm;
            const float* fuv = a.face_uvs + (size_t)hf * 6;
#pragma unroll
            for (int i = 0; i < 6; ++i) fu[i] = fuv[i];
            const float* nn = a.fn + ((size_t)b * a.F + hf) * 3;
            n0 = nn[0]; n1 = nn[1]; n2 = nn[2];
            m = (w0 + w1) + w2;
            u = (w0 * fu[0] + w1 * fu[2]) + w2 * fu[4];
            v = (w0 * fu[1] + w1 * fu[3]) + w2 * fu[5];
            nx = (w0 * n0 + w1 * n0) + w2 * n0;
            ny = (w0 * n1 + w1 * n1) + w2 * n1;
            nz = (w0 * n2 + w1 * n2) + w2 * n2;
        }
        const Bilin s = bilin_setup(u, v, a.Ht, a.Wt);
        const bool inw = s.x0 < a.Wt && s.y0 < a.Ht, ine = s.x1 < a.Wt && s.y0 < a.Ht;
        const bool isw = s.x0 < a.Wt && s.y1 < a.Ht, ise = s.x1 < a.Wt && s.y1 < a.Ht;
        float bnd[9];
        sh_bands(nx, ny, nz, bnd);
        const float* L = a.lights + b * 9;
        float coef = 0.f;
#pragma unroll
        for (int i = 0; i < 9; ++i) coef += bnd[i] * L[i];

        float dm = 0.f, dc = 0.f, gix = 0.f, giy = 0.f, dtcv[3];
        const float ex = 1.f - s.tx, ey = 1.f - s.ty;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* tex = a.textures + ((size_t)b * 3 + c) * a.Ht * a.Wt;
            const float tnw = inw ? tex[(size_t)s.y0 * a.Wt + s.x0] : 0.f, tne = ine ? tex[(size_t)s.y0 * a.Wt + s.x1] : 0.f;
            const float tsw = isw ? tex[(size_t)s.y1 * a.Wt + s.x0] : 0.f, tse = ise ? tex[(size_t)s.y1 * a.Wt + s.x1] : 0.f;
            float tc = 0.f;
            if (inw) tc += tnw * s.wnw;
            if (ine) tc += tne * s.wne;
            if (isw) tc += tsw * s.wsw;
            if (ise) tc += tse * s.wse;
            float pre, dtc;
            if (kNoMask) {
                const float bgv = a.bg[((size_t)b * 3 + c) * hw + pin];
                const float base = tc * m + bgv * (1.f - m);
                pre = base * coef;
                const float g = (pre >= 0.f && pre <= 1.f) ? gin[c] : 0.f;      // torch.clamp backward mask
                dc += g * base;
                const float dbase = g * coef;
                dtc = dbase * m;
                a.grad_bg[((size_t)b * 3 + c) * hw + pin] = dbase * (1.f - m);
                dm += dbase * (tc - bgv);
            } else {
                pre = (tc * m) * coef + 1.f * (1.f - m);
                const float g = (pre >= 0.f && pre <= 1.f) ? gin[c] : 0.f;
                dc += g * (tc * m);
                dtc = (g * coef) * m;
                dm += g * (tc * coef - 1.f);
            }
            dtcv[c] = dtc;
            gix += dtc * ((tne - tnw) * ey + (tse - tsw) * s.ty);
            giy += dtc * ((tsw - tnw) * ex + (tse - tne) * s.tx);
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) dl[i] = dc * bnd[i];
        if (hf >= 0) {
            const float du = gix * s.mx * ((float)a.Wt / 2.f) * 2.f;
            const float dv = giy * s.my * ((float)a.Ht / 2.f) * -2.f;
            const float dnx = dc * (((MM_SH_C1 * L[1] + MM_SH_C4 * ny * L[4]) + MM_SH_C7 * nz * L[7]) + 2.f * MM_SH_C8 * nx * L[8]);
            const float dny = dc * (((MM_SH_C1 * L[3] + MM_SH_C4 * nx * L[4]) + MM_SH_C4 * nz * L[5]) - 2.f * MM_SH_C8 * ny * L[8]);
            const float dnz = dc * (((MM_SH_C1 * L[2] + MM_SH_C4 * ny * L[5]) + 2.f * MM_SH_C6 * nz * L[6]) + MM_SH_C7 * nx * L[7]);
            a.gp0[pix] = make_float4(dtcv[0], dtcv[1], dtcv[2], dm);
            a.gp1[pix] = make_float4(du, dv, dnx, dny);
            a.gp2[pix] = dnz;
        }
    }

    // d lights: wave butterfly -> one LDS row per wave -> fixed-order partial of this workgroup (summed by vertex_bwd)
#pragma unroll
    for (int i = 0; i < 9; ++i) dl[i] = wave_sum(dl[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) s_dl[wave][i] = dl[i];
    }
    __syncthreads();
    if (threadIdx.x < 9)
        a.dl_part[((size_t)b * a.blocks_per_image + blk) * 12 + threadIdx.x] =
            ((s_dl[0][threadIdx.x] + s_dl[1][threadIdx.x]) + s_dl[2][threadIdx.x]) + s_dl[3][threadIdx.x];
}

// ---------------------------------------------------------------------------------------------------------------------
// 2. gathers.  Both sweep a face's screen box with 16 lanes, four pixels per lane per trip with the loads of a trip
//    issued together (these loops are latency-bound: a dependent HBM/L2 load per step).
// ---------------------------------------------------------------------------------------------------------------------
#define MM_TS MM_UV_TILE
#define MM_SWEEP 4

__device__ inline float group16_sum(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 16);
    return v;
}

struct FaceBox { float4 p0, p1; float xmin, ymin, xmax, ymax; int px0, py0, bw, npx; };

__device__ inline FaceBox face_box(const BwdArgs& a, size_t o, float pad) {
    FaceBox fb;
    fb.p0 = a.geo[o * 3 + 0]; fb.p1 = a.geo[o * 3 + 1];
    fb.xmin = fminf(fminf(fb.p0.x, fb.p0.z), fb.p1.x); fb.ymin = fminf(fminf(fb.p0.y, fb.p0.w), fb.p1.y);
    fb.xmax = fmaxf(fmaxf(fb.p0.x, fb.p0.z), fb.p1.x); fb.ymax = fmaxf(fmaxf(fb.p0.y, fb.p0.w), fb.p1.y);
    int px1, py1;
    pixel_range(fb.xmin - pad, fb.xmax + pad, a.mult, a.W, false, fb.px0, px1);
    pixel_range(fb.ymin - pad, fb.ymax + pad, a.mult, a.H, true, fb.py0, py1);
    fb.bw = px1 - fb.px0 + 1;
    const int bh = py1 - fb.py0 + 1;
    fb.npx = (fb.bw > 0 && bh > 0) ? fb.bw * bh : 0;
    return fb;
}

// 2a. texture gradient: one workgroup per (image, 32x32-texel tile), accumulators in LDS, every texel written once.
__global__ __launch_bounds__(256) void texture_gather_kernel(BwdArgs a) {
    __shared__ float s_acc[3][MM_TS * MM_TS];
    const int ntiles = a.ntx * a.nty;
    int b, T;
    map_block(blockIdx.x, a.B, ntiles, b, T);
    const int tid = threadIdx.x;
    for (int i = tid; i < 3 * MM_TS * MM_TS; i += 256) (&s_acc[0][0])[i] = 0.f;
    __syncthreads();
    const int tx0 = (T % a.ntx) * MM_TS, ty0 = (T / a.ntx) * MM_TS;
    const int beg = a.uvt_offsets[T], end = a.uvt_offsets[T + 1];
    const int grp = tid >> 4, sl = tid & 15;
    const size_t hw = (size_t)a.H * a.W;

    for (int k = beg + grp; k < end; k += 16) {
        const int f = a.uvt_faces[k] & 0x7FFFFFFF;
        const size_t o = (size_t)b * a.F + f;
        const FaceBox fb = face_box(a, o, 0.f);                  // owned pixels lie inside the face's own box
        const float* fu = a.face_uvs + (size_t)f * 6;
        for (int base = 0; base < fb.npx; base += 16 * MM_SWEEP) {
            size_t pixv[MM_SWEEP]; int fiv[MM_SWEEP]; float4 q0v[MM_SWEEP]; int pxv[MM_SWEEP], pyv[MM_SWEEP];
#pragma unroll
            for (int i = 0; i < MM_SWEEP; ++i) {
                const int idx = base + i * 16 + sl;
                const int yy = idx / fb.bw;
                pxv[i] = fb.px0 + (idx - yy * fb.bw); pyv[i] = fb.py0 + yy;
                pixv[i] = (size_t)b * hw + (size_t)pyv[i] * a.W + pxv[i];
                fiv[i] = idx < fb.npx ? a.face_idx[pixv[i]] : -2;
            }
#pragma unroll
            for (int i = 0; i < MM_SWEEP; ++i) q0v[i] = fiv[i] == f ? a.gp0[pixv[i]] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < MM_SWEEP; ++i) {
                if (fiv[i] != f) continue;
                const float x0 = pixel_x(pxv[i], a.W, a.mult), y0 = pixel_y(pyv[i], a.H, a.mult);
                float w0, w1, w2, nrm;
                edge_weights(fb.p0.x, fb.p0.y, fb.p0.z, fb.p0.w, fb.p1.x, fb.p1.y, x0, y0, a.eps, w0, w1, w2, nrm);
                w0 /= nrm; w1 /= nrm; w2 /= nrm;
                const float u = (w0 * fu[0] + w1 * fu[2]) + w2 * fu[4];
                const float v = (w0 * fu[1] + w1 * fu[3]) + w2 * fu[5];
                const Bilin s = bilin_setup(u, v, a.Ht, a.Wt);
                const int lx0 = s.x0 - tx0, lx1 = s.x1 - tx0, ly0 = s.y0 - ty0, ly1 = s.y1 - ty0;
                const bool cx0 = lx0 >= 0 && lx0 < MM_TS, cx1 = lx1 >= 0 && lx1 < MM_TS && s.x1 < a.Wt;
                const bool cy0 = ly0 >= 0 && ly0 < MM_TS, cy1 = ly1 >= 0 && ly1 < MM_TS && s.y1 < a.Ht;
                const float dt[3] = {q0v[i].x, q0v[i].y, q0v[i].z};
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if (dt[c] != 0.f) {
                        if (cx0 && cy0) atomicAdd(&s_acc[c][ly0 * MM_TS + lx0], dt[c] * s.wnw);
                        if (cx1 && cy0) atomicAdd(&s_acc[c][ly0 * MM_TS + lx1], dt[c] * s.wne);
                        if (cx0 && cy1) atomicAdd(&s_acc[c][ly1 * MM_TS + lx0], dt[c] * s.wsw);
                        if (cx1 && cy1) atomicAdd(&s_acc[c][ly1 * MM_TS + lx1], dt[c] * s.wse);
                    }
                }
            }
        }
    }
    __syncthreads();
    // write the tile once (also where nothing landed: no separate zero-fill of grad_textures)
    for (int i = tid; i < 3 * MM_TS * MM_TS; i += 256) {
        const int c = i / (MM_TS * MM_TS), r = i - c * (MM_TS * MM_TS);
        const int ly = r / MM_TS, lx = r - ly * MM_TS;
        const int x = tx0 + lx, y = ty0 + ly;
        if (x < a.Wt && y < a.Ht) a.grad_textures[(((size_t)b * 3 + c) * a.Ht + y) * a.Wt + x] = s_acc[c][r];
    }
}

// 2b. per-face gradients: 16 lanes per (image, face) sweep the face's inflated box; pixels it owns give the K2 barycentric
//     gradient, uncovered pixels that hold it among their first knum soft-mask faces give K4.  One plain store per face.
__global__ __launch_bounds__(256) void face_gather_kernel(BwdArgs a) {
    const int sl = threadIdx.x & 15;
    const long long gid = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (gid >= (long long)a.B * a.F) return;
    const int b = (int)(gid / a.F), f = (int)(gid - (long long)b * a.F);
    const size_t o = (size_t)gid, hw = (size_t)a.H * a.W;
    const float s2 = a.mult * a.mult;
    const FaceBox fb = face_box(a, o, a.infl);
    const float* fu = a.face_uvs + (size_t)f * 6;
    const float* nn = a.fn + o * 3;
    const float n0 = nn[0], n1 = nn[1], n2 = nn[2];
    const float4 p0 = fb.p0, p1 = fb.p1;
    float gxy[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gn[3] = {0.f, 0.f, 0.f};

    for (int base = 0; base < fb.npx; base += 16 * MM_SWEEP) {
        size_t pixv[MM_SWEEP]; int fiv[MM_SWEEP], pxv[MM_SWEEP], pyv[MM_SWEEP];
        float4 q0v[MM_SWEEP], q1v[MM_SWEEP]; float q2v[MM_SWEEP], sqv[MM_SWEEP], gav[MM_SWEEP]; int lfv[MM_SWEEP];
#pragma unroll
        for (int i = 0; i < MM_SWEEP; ++i) {
            const int idx = base + i * 16 + sl;
            const int yy = idx / fb.bw;
            pxv[i] = fb.px0 + (idx - yy * fb.bw); pyv[i] = fb.py0 + yy;
            pixv[i] = (size_t)b * hw + (size_t)pyv[i] * a.W + pxv[i];
            fiv[i] = idx < fb.npx ? a.face_idx[pixv[i]] : -2;
        }
#pragma unroll
        for (int i = 0; i < MM_SWEEP; ++i) {
            const bool own = fiv[i] == f, opn = fiv[i] == -1;
            q0v[i] = own ? a.gp0[pixv[i]] : make_float4(0.f, 0.f, 0.f, 0.f);
            q1v[i] = own ? a.gp1[pixv[i]] : make_float4(0.f, 0.f, 0.f, 0.f);
            q2v[i] = own ? a.gp2[pixv[i]] : 0.f;
            sqv[i] = opn ? a.softq[pixv[i]] : 0.f;
            lfv[i] = opn ? a.lastf[pixv[i]] : -1;
            gav[i] = opn ? a.grad_rgba[pixv[i] * 4 + 3] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < MM_SWEEP; ++i) {
            const float x0 = pixel_x(pxv[i], a.W, a.mult), y0 = pixel_y(pyv[i], a.H, a.mult);
            if (fiv[i] == f) {
                // ---- K2 (Appendix A.1): features per corner k = (1, u_k, v_k, n)
                float w0, w1, w2, nrm;
                edge_weights(p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, x0, y0, a.eps, w0, w1, w2, nrm);
                w0 /= nrm; w1 /= nrm; w2 /= nrm;
                const float dm = q0v[i].w, du = q1v[i].x, dv = q1v[i].y, dnx = q1v[i].z, dny = q1v[i].w, dnz = q2v[i];
                const float gnn = (dnx * n0 + dny * n1) + dnz * n2;
                const float G0 = ((dm + du * fu[0]) + dv * fu[1]) + gnn;
                const float G1 = ((dm + du * fu[2]) + dv * fu[3]) + gnn;
                const float G2 = ((dm + du * fu[4]) + dv * fu[5]) + gnn;
                const float Gm = (w0 * G0 + w1 * G1) + w2 * G2;
                const float dw0 = (G0 - Gm) / nrm, dw1 = (G1 - Gm) / nrm, dw2 = (G2 - Gm) / nrm;
                const float aex = p0.x - x0, aey = p0.y - y0, bex = p0.z - x0, bey = p0.w - y0, cex = p1.x - x0, cey = p1.y - y0;
                gxy[0] += (dw1 * (-cey) + dw2 * bey) * a.mult;
                gxy[1] += (dw1 * cex + dw2 * (-bex)) * a.mult;
                gxy[2] += (dw0 * cey + dw2 * (-aey)) * a.mult;
                gxy[3] += (dw0 * (-cex) + dw2 * aex) * a.mult;
                gxy[4] += (dw0 * (-bey) + dw1 * aey) * a.mult;
                gxy[5] += (dw0 * bex + dw1 * (-aex)) * a.mult;
                gn[0] += (w0 * dnx + w1 * dnx) + w2 * dnx;
                gn[1] += (w0 * dny + w1 * dny) + w2 * dny;
                gn[2] += (w0 * dnz + w1 * dnz) + w2 * dnz;
            } else if (fiv[i] == -1) {
                // ---- K4 (Appendix A.2): an uncovered pixel that may hold this face among its first knum soft-mask faces
                const float sq = sqv[i], ga = gav[i];
                if (sq != 0.f && sq != 1.f && ga != 0.f && f <= lfv[i] &&
                    !(x0 < fb.xmin - a.infl || x0 > fb.xmax + a.infl || y0 < fb.ymin - a.infl || y0 > fb.ymax + a.infl)) {
                    int r, ty;
                    float d = seg_dist2(x0, y0, p0.x, p0.y, p0.z, p0.w, ty);
                    const float d1 = seg_dist2(x0, y0, p0.z, p0.w, p1.x, p1.y, r); if (d1 < d) { d = d1; ty = 3 + r; }
                    const float d2 = seg_dist2(x0, y0, p1.x, p1.y, p0.x, p0.y, r); if (d2 < d) { d = d2; ty = 6 + r; }
                    const float p = expf(-((d / s2) * a.sigmainv));
                    const float q = 1.f - p;
                    const float qnz = fabsf(sq);
                    const bool onezero = sq < 0.f;
                    const float excl = (q != 0.f) ? (onezero ? 0.f : qnz / q) : (onezero ? qnz : 0.f);
                    const float gd = ga * excl * (-(p * a.sigmainv) / s2);
                    if (gd != 0.f) {
                        const int e = ty / 3, reg = ty - e * 3;
                        const float ux = e == 0 ? p0.x : (e == 1 ? p0.z : p1.x), uy = e == 0 ? p0.y : (e == 1 ? p0.w : p1.y);
                        const float wx = e == 0 ? p0.z : (e == 1 ? p1.x : p0.x), wy = e == 0 ? p0.w : (e == 1 ? p1.y : p0.y);
                        float dux = 0.f, duy = 0.f, dvx = 0.f, dvy = 0.f;
                        if (reg == 0) { dux = -2.f * (x0 - ux); duy = -2.f * (y0 - uy); }
                        else if (reg == 2) { dvx = -2.f * (x0 - wx); dvy = -2.f * (y0 - wy); }
                        else {
                            const float ex = wx - ux, ey = wy - uy, rx = x0 - ux, ry = y0 - uy;
                            const float tt = (rx * ex + ry * ey) / (ex * ex + ey * ey);
                            const float qx = x0 - (ux + tt * ex), qy = y0 - (uy + tt * ey);
                            dux = -2.f * (1.f - tt) * qx; duy = -2.f * (1.f - tt) * qy;
                            dvx = -2.f * tt * qx; dvy = -2.f * tt * qy;
                        }
                        const float sux = gd * dux * a.mult, suy = gd * duy * a.mult, svx = gd * dvx * a.mult, svy = gd * dvy * a.mult;
                        // edge e runs from corner e to corner (e+1)%3
                        if (e == 0) { gxy[0] += sux; gxy[1] += suy; gxy[2] += svx; gxy[3] += svy; }
                        else if (e == 1) { gxy[2] += sux; gxy[3] += suy; gxy[4] += svx; gxy[5] += svy; }
                        else { gxy[4] += sux; gxy[5] += suy; gxy[0] += svx; gxy[1] += svy; }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) gxy[i] = group16_sum(gxy[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) gn[i] = group16_sum(gn[i]);
    if (sl == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) a.dfxy[o * 6 + i] = gxy[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) a.dfn[o * 3 + i] = gn[i];
    }
}

int launch_raster_bwd(const MMRenderDesc* d, const MMRenderGrads* g, const Workspace& w, hipStream_t s) {
    BwdArgs a;
    a.B = d->B; a.H = d->H; a.W = d->W; a.F = d->F; a.Ht = d->Ht; a.Wt = d->Wt; a.knum = d->knum;
    a.blocks_x = (d->W + MM_BLOCK_PX - 1) / MM_BLOCK_PX;
    a.blocks_per_image = w.blocks_per_image;
    a.mult = d->multiplier; a.eps = d->eps; a.sigmainv = d->sigmainv; a.infl = d->boxlen * d->multiplier;
    a.geo = w.geo; a.face_uvs = d->face_uvs; a.fn = d->face_normals; a.textures = d->textures; a.lights = d->lights; a.bg = d->bg;
    a.face_idx = d->face_idx; a.softq = w.softq; a.lastf = w.lastf; a.grad_rgba = g->grad_rgba;
    a.gp0 = w.gp0; a.gp1 = w.gp1; a.gp2 = w.gp2; a.dl_part = w.dl_part; a.grad_bg = g->grad_bg;
    a.dTacc = w.dTacc; a.ticket = w.ticket;
    a.uvt_offsets = d->uvt_offsets; a.uvt_faces = d->uvt_faces;
    a.ntx = (d->Wt + MM_TS - 1) / MM_TS; a.nty = (d->Ht + MM_TS - 1) / MM_TS;
    a.grad_textures = g->grad_textures; a.dfxy = w.dfxy; a.dfn = w.dfn;
    {
        ProfScope p(d->prof_events, MM_PROF_PIXEL_BWD, s);
        dim3 grid(a.blocks_per_image * d->B);
        if (d->no_mask) hipLaunchKernelGGL(pixel_bwd_kernel<true>, grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(pixel_bwd_kernel<false>, grid, dim3(256), 0, s, a);
    }
    if (hipGetLastError() != hipSuccess) return MM_ERR_LAUNCH;
    {
        ProfScope p(d->prof_events, MM_PROF_GATHER_BWD, s);
        hipLaunchKernelGGL(texture_gather_kernel, dim3(a.ntx * a.nty * d->B), dim3(256), 0, s, a);
        hipLaunchKernelGGL(face_gather_kernel, dim3((unsigned)(((long long)d->B * d->F + 15) / 16)), dim3(256), 0, s, a);
    }
    return hipGetLastError() == hipSuccess ? MM_OK : MM_ERR_LAUNCH;
}

}  // namespace mm
