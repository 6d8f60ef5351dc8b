// mm_pixel_pass.h -- the pixel-major pass of the render path's backward as a per-WAVE device function (one lane per pixel of an 8x8 tile), and the
// plan of the face sweep.  Two callers with identical results:
//   mm_pixel_bwd.hip   pixel_bwd_kernel: the stand-alone launch of mm_render_backward (four tiles of a 16x16 block per workgroup)
//   mm_raster.hip      raster_fwd_kernel<.., kStep>: mm_render_step folds the pass into the forward's walk kernel -- a tile's wave runs it right
//                      behind its epilogue, on operands that are still in the caches: one launch and one pass over the screen less per step
// Both translation units are compiled with the forward's floating-point flags (mm_backward.h: the pass re-forms the forward's per-pixel quantities).
#pragma once
#include "mm_backward.h"

namespace mm {

// ---------------------------------------------------------------------------------------------------------------------
// 0. plan of the face sweep (the first MM_PLAN_WGS * B workgroups of pixel_bwd's grid; nothing in the pixel pass depends on it and the
//    gather launch behind it finds it done): every face's inflated pixel box cut into chunks of MM_CHUNK_PX pixels, numbered in face
//    order by an exclusive scan of the chunk counts.  Thread t owns the contiguous faces [t*per, (t+1)*per): it adds up their counts, ONE
//    block scan gives its first item, and it numbers its faces' chunks from there.  Should the items run out (more than sixteen screens'
//    worth of box pixels in one image), the image's chunk size doubles until they fit (item_cap >= F, so it ends).  It used to run
//    between the vertex stage and the walk, on the forward's critical path (63 us at 13 776 faces); here it costs the step nothing.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef MM_PLAN_LDS_FACES
#define MM_PLAN_LDS_FACES 14336   // 28 KiB of LDS: five workgroups per CU stay possible
#endif
#define MM_PLAN_LDS_BYTES (sizeof(int) * MM_PLAN_WGS * 4 + sizeof(unsigned short) * MM_PLAN_LDS_FACES)
// (MM_PLAN_WGS workgroups per image where faces are many, else one: each counts every face -- cheap, from LDS -- and writes the items of its share)
__device__ inline void plan_sweep_items(const BwdArgs& a, int b, int q, void* lds) {
    const int nwg = a.plan_wgs;                                   // 1 or MM_PLAN_WGS
    // lds: MM_PLAN_LDS_BYTES of the caller's LDS (the pixel pass's own array, or the walk kernel's candidate staging, which a plan workgroup never uses)
    int (*s_wave)[4] = reinterpret_cast<int (*)[4]>(lds);
    // the faces' chunk counts at the base chunk size are staged in LDS (2 bytes a face, read once, coalesced, eight loads in flight per
    // thread): with thousands of faces per thread-range the passes below were a chain of dependent trips to memory, one per face.
    // ceil(ceil(n / c) / 2^k) = ceil(n / (c 2^k)): the doubled chunk sizes need nothing else.
    unsigned short* s_nch = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(lds) + sizeof(int) * MM_PLAN_WGS * 4);
    const int tid = threadIdx.x;
    const bool staged = a.F <= MM_PLAN_LDS_FACES;                 // (more faces than that: the counts are re-read from the face records)
    auto box_px = [&](int f) {                                   // pixels of the face's sweep box; 0: the box misses the image, or no pixel refers to the face
        const float4 q2 = a.geo[((size_t)b * a.F + f) * 3 + 2];   //  (most faces of a fine, overlapping mesh: nothing to sweep)
        int own = 1, taken = 1;
        if (a.fflag) { const int2 fl = reinterpret_cast<const int2*>(a.fflag)[(size_t)b * a.F + f]; own = fl.x; taken = fl.y; }
        if (!(own | taken)) return 0;
        int px0, py0, bw, bh;
        sweep_box(__float_as_uint(q2.z), __float_as_uint(q2.w), taken != 0, a.sweep_sx, a.sweep_sy, a.W, a.H, px0, py0, bw, bh);
        return bw * bh;
    };
    if (staged) {
        for (int f0 = tid; f0 < a.F; f0 += 8 * 256) {
            int px[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) px[u] = f0 + u * 256 < a.F ? box_px(f0 + u * 256) : 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) if (f0 + u * 256 < a.F) s_nch[f0 + u * 256] = (unsigned short)min((px[u] + MM_CHUNK_PX - 1) / MM_CHUNK_PX, 65535);
        }
        __syncthreads();
    }
    auto chunks = [&](int f, int shift) {                        // the face's items at chunk size MM_CHUNK_PX << shift
        if (staged) return ((int)s_nch[f] + (1 << shift) - 1) >> shift;
        const int chunk = MM_CHUNK_PX << shift;
        return (box_px(f) + chunk - 1) / chunk;
    };
    // the faces are cut into MM_PLAN_WGS * 256 contiguous ranges; range (k, t) = faces of thread t of workgroup k.  Every workgroup
    // counts all of them (so that it knows the total and what lies in front of its own quarter) and writes only its own.
    const int per = (a.F + nwg * 256 - 1) / (nwg * 256);
    int shift = 0, first = 0, total = 0;
    for (;; ++shift) {
        int mine[MM_PLAN_WGS], pre = 0;
#pragma unroll
        for (int k = 0; k < MM_PLAN_WGS; ++k) {
            if (k >= nwg) { if ((tid & 63) == 63) s_wave[k][tid >> 6] = 0; continue; }      // (workgroup-uniform)
            const int f0 = min(a.F, (k * 256 + tid) * per), f1 = min(a.F, f0 + per);
            mine[k] = 0;
            for (int f = f0; f < f1; ++f) mine[k] += chunks(f, shift);
            int wsum;
            const int inc = wave_prefix_excl(mine[k], tid & 63, wsum) + mine[k];
            if (k == q) pre = inc - mine[k];
            if (k == 0) __syncthreads();                         // (s_wave of the previous round has been read)
            if ((tid & 63) == 63) s_wave[k][tid >> 6] = inc;
        }
        __syncthreads();
        first = pre; total = 0;
#pragma unroll
        for (int k = 0; k < MM_PLAN_WGS; ++k) {
            const int tk = ((s_wave[k][0] + s_wave[k][1]) + s_wave[k][2]) + s_wave[k][3];
            if (k < q) first += tk;
            if (k == q) for (int w = 0; w < (tid >> 6); ++w) first += s_wave[k][w];
            total += tk;
        }
        if (total <= a.item_cap || shift >= 20) break;           // workgroup-uniform (and the same in the image's other workgroups)
    }
    const int chunk = MM_CHUNK_PX << shift;
    const int f0 = min(a.F, (q * 256 + tid) * per), f1 = min(a.F, f0 + per);
    for (int f = f0; f < f1; ++f) {
        const int nch = chunks(f, shift);
        a.plan_chunkmap[(size_t)b * a.F + f] = make_int2(first, nch);
        for (int c = 0; c < nch; ++c) a.plan_items[(size_t)b * a.item_cap + first + c] = make_int2(f, c);
        first += nch;
    }
    if (q == 0 && tid == 0) a.plan_nitems[b] = make_int2(total, chunk);
}

// ---------------------------------------------------------------------------------------------------------------------
// 1. pixel-major pass
// ---------------------------------------------------------------------------------------------------------------------

// what a wave's pass leaves in registers for its caller to reduce / store
struct PixelWaveOut { float dl[9]; float m2, m4; };

// alpha_from_gt (BwdArgs): fused loss only -- dL/dalpha of the uncovered pixels is NOT written (gp2) nor bounded (m4) here: the face gather forms
// it from the ground-truth mask and the image's loss totals itself.  That keeps this pass free of the totals, which is what lets it run inside the
// forward's walk kernel, before the totals are complete.
// hf_in: the pixel's winning face if the caller holds it (the walk kernel, right behind its epilogue), MM_HF_LOAD: read it from face_idx
#define MM_HF_LOAD (-2)
template <bool kNoMask>
__device__ inline void pixel_backward_wave(const BwdArgs& a, int b, int blk, int wave, int lane, int hf_in, PixelWaveOut& out) {
    const int bx = blk % a.blocks_x, by = blk / a.blocks_x;
    const int px = bx * MM_BLOCK_PX + (wave & 1) * MM_TILE + (lane & 7), py = by * MM_BLOCK_PX + (wave >> 1) * MM_TILE + (lane >> 3);
    const bool in_img = px < a.W && py < a.H;
    const float x0 = pixel_x_k(px, a.W, a.kx), y0 = pixel_y_k(py, a.H, a.ky);                  // (host-formed IEEE quotients: the forward's centres)
    const size_t hw = (size_t)a.H * a.W, pin = (size_t)py * a.W + px;
    const size_t pix = (size_t)b * hw + pin;
    // The pass is a chain of dependent trips to memory; it is written so that four remain: (1) everything addressed by the pixel
    // alone -- face_idx, prediction, ground truth, background; (2) what the winner's id addresses -- geometry, normal, corner uvs;
    // (3) the twelve texels, unconditionally from clamped addresses; (4) nothing: the record is stored at the pixel's own place.  (Loads left inside per-lane
    // branches or behind stores that might alias them each cost the wave a trip of their own.)
    float bgv[3] = {0.f, 0.f, 0.f};
    if (kNoMask && in_img) {
#pragma unroll
        for (int c = 0; c < 3; ++c) bgv[c] = a.bg[((size_t)b * 3 + c) * hw + pin];
    }
    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
    int hf = -1;
    // fused recon_data backward (Appendix A.4): dL/dpred_c = kl1 * sign(pred_c' - gt_c') * gm needs the PREDICTION -- which this pass
    // recomputes anyway, bit for bit (it is compiled like the forward for that reason: mm_backward.h), so the forward image is not read
    // back (16 bytes per pixel of a bandwidth-bound kernel); the sign is taken where the pixel's colour is re-formed (grad_colour below).
    float gi3[3] = {0.f, 0.f, 0.f}, gmv = 0.f, kl1 = 0.f;
    const bool fused = a.gt != nullptr;
    if (fused) {
        // dL/dpred_c = kl1 * sign * gm: the same coefficient for every pixel of the batch.  dL/dalpha = ka * gm + kb * (1 - gm) needs the image's
        // loss totals and only matters to uncovered pixels: the face gather forms it (alpha_from_gt, mm_backward.hip: alpha_gradient)
        const float gs = a.grad_loss ? a.grad_loss[0] : 1.f;
        kl1 = gs * a.image_weight / ((float)a.B * 3.f * (float)a.H * (float)a.W);
        if (in_img) {
            hf = hf_in == MM_HF_LOAD ? a.face_idx[pix] : hf_in;
            const float* g = a.gt + (size_t)b * 4 * hw;
            const float gm = g[3 * hw + pin];
            gmv = gm;
#pragma unroll
            for (int c = 0; c < 3; ++c) gi3[c] = g[c * hw + pin] * gm + 1.f * (1.f - gm);
        }
    } else if (in_img) { g4 = *(const float4*)(a.grad_rgba + pix * 4); hf = hf_in == MM_HF_LOAD ? a.face_idx[pix] : hf_in; }
    const float gin[3] = {g4.x, g4.y, g4.z};
    // dL/d(colour c of this pixel) given its un-clamped value `pre`: the caller's gradient, or the fused loss's (the forward's clamp and
    // masking expressions, shade_store / shade_empty_tiles + networks.py:370-377)
    auto grad_colour = [&](int c, float pre) -> float {
        if (!fused) return gin[c];
        const float pc = pre < 0.f ? 0.f : (pre > 1.f ? 1.f : pre);
        const float pi = pc * gmv + 1.f * (1.f - gmv);
        const float df = pi - gi3[c], sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
        return kl1 * sg * gmv;
    };
    float m2 = 0.f, m4 = 0.f;                                    // this lane's largest |K2 number| / |dL/dalpha|: the gather's fixed-point scale
    if (!fused && in_img && hf < 0) { a.gp2[pix] = g4.w; m4 = fabsf(g4.w); }   // the face gather (K4) needs dL/dalpha of uncovered pixels (fused: it forms it itself)
    // dL/dlights of this pixel = dcs * sh_bands(normal): kept as the scalar and the normal (4 registers, not 9, across the record append below)
    float dcs = 0.f, snx = 0.f, sny = 0.f, snz = 0.f;
    TexRecord rec; rec.xy = MM_TREC_NONE; rec.tx = rec.ty = rec.d0 = rec.d1 = rec.d2 = 0.f;
    unsigned bx0 = 255u, by0 = 255u; int bx1 = 0, by1 = 0;      // texture tiles under this pixel's bilinear footprint (none)

    // Tiles without a covered pixel (more than half of them): m = 0 and n = 0 in every lane, so only the background and the two
    // constant SH bands receive gradient -- none of the uv / bilinear / texel / barycentric work below is needed.
    const bool any_covered = __ballot(in_img && hf >= 0) != 0;   // wave-uniform
    if (!any_covered) {
        if (kNoMask && in_img) {
            const float* L = a.lights + b * 9;                   // (bands 0 and 6 only: the same lights whatever the band order)
            const float coef = MM_SH_C0 * L[0] + (0.f - MM_SH_C6B) * L[6];
            float dc = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float pre = bgv[c] * coef;
                const float g = (pre >= 0.f && pre <= 1.f) ? grad_colour(c, pre) : 0.f;      // torch.clamp backward mask
                dc += g * bgv[c];
                a.grad_bg[((size_t)b * 3 + c) * hw + pin] = g * coef;
            }
            dcs = dc;                                            // (normal 0: bands 0 and 6 only)
        }
    } else if (in_img && (hf >= 0 || kNoMask)) {
        // recompute the forward quantities of this pixel (only face_idx and the soft-mask state were saved)
        float w0 = 0.f, w1 = 0.f, w2 = 0.f, nrm = 1.f, m = 0.f, u = 0.f, v = 0.f, nx = 0.f, ny = 0.f, nz = 0.f;
        float fu[6] = {0, 0, 0, 0, 0, 0}, n0 = 0.f, n1 = 0.f, n2 = 0.f;
        float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0;
        {   // trip 2 (uncovered lanes of a covered tile read face 0's records and ignore them)
            const int fs = max(hf, 0);
            const float4* geo = a.geo + ((size_t)b * a.F + fs) * 3;
            const float4 q0 = geo[0], q1 = geo[1];
            const float2* fuv = (const float2*)(a.face_uvs + (size_t)fs * 6);
            const float2 u0 = fuv[0], u1 = fuv[1], u2 = fuv[2];
            const float* nn = a.fn + ((size_t)b * a.F + fs) * 3;
            const float m0 = nn[0], m1 = nn[1], m2 = nn[2];
            if (hf >= 0) {
                p0 = q0; p1 = q1;
                fu[0] = u0.x; fu[1] = u0.y; fu[2] = u1.x; fu[3] = u1.y; fu[4] = u2.x; fu[5] = u2.y;
                n0 = m0; n1 = m1; n2 = m2;
            }
        }
        if (hf >= 0) {
            // (MM_OPT_BARY_ONE_MINUS changes the weights by O(eps); the derivative below stays that of the default form)
            bary_weights(p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, x0, y0, a.eps, (a.options & MM_OPT_BARY_ONE_MINUS) != 0, w0, w1, w2, nrm);
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
        // trip 3: twelve loads in flight together
        float tq[3][4];
        {
            const int cx0 = min(max(s.x0, 0), a.Wt - 1), cx1 = min(max(s.x1, 0), a.Wt - 1);
            const int cy0 = min(max(s.y0, 0), a.Ht - 1), cy1 = min(max(s.y1, 0), a.Ht - 1);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float* tex = a.textures + ((size_t)b * 3 + c) * a.Ht * a.Wt;
                tq[c][0] = tex[(size_t)cy0 * a.Wt + cx0]; tq[c][1] = tex[(size_t)cy0 * a.Wt + cx1];
                tq[c][2] = tex[(size_t)cy1 * a.Wt + cx0]; tq[c][3] = tex[(size_t)cy1 * a.Wt + cx1];
            }
        }
        float bnd[9];
        sh_bands(nx, ny, nz, bnd);
        float L[9];                                              // lights in sh_bands' order (see shade_store)
#pragma unroll
        for (int i = 0; i < 9; ++i) L[i] = a.lights[b * 9 + i];
        if (a.options & MM_OPT_SH_ORDER_XYZ) { const float tmp = L[2]; L[2] = L[3]; L[3] = tmp; }
        float coef = 0.f;
#pragma unroll
        for (int i = 0; i < 9; ++i) coef += bnd[i] * L[i];

        float dm = 0.f, dc = 0.f, gix = 0.f, giy = 0.f, dtcv[3];
        const float ex = 1.f - s.tx, ey = 1.f - s.ty;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float tnw = inw ? tq[c][0] : 0.f, tne = ine ? tq[c][1] : 0.f;
            const float tsw = isw ? tq[c][2] : 0.f, tse = ise ? tq[c][3] : 0.f;
            float tc = 0.f;
            if (inw) tc += tnw * s.wnw;
            if (ine) tc += tne * s.wne;
            if (isw) tc += tsw * s.wsw;
            if (ise) tc += tse * s.wse;
            float pre, dtc;
            if (kNoMask) {
                const float bgvc = bgv[c];
                const float base = tc * m + bgvc * (1.f - m);
                pre = base * coef;
                const float g = (pre >= 0.f && pre <= 1.f) ? grad_colour(c, pre) : 0.f;      // torch.clamp backward mask
                dc += g * base;
                const float dbase = g * coef;
                dtc = dbase * m;
                a.grad_bg[((size_t)b * 3 + c) * hw + pin] = dbase * (1.f - m);
                dm += dbase * (tc - bgvc);
            } else {
                pre = (tc * m) * coef + 1.f * (1.f - m);
                const float g = (pre >= 0.f && pre <= 1.f) ? grad_colour(c, pre) : 0.f;
                dc += g * (tc * m);
                dtc = (g * coef) * m;
                dm += g * (tc * coef - 1.f);
            }
            dtcv[c] = dtc;
            gix += dtc * ((tne - tnw) * ey + (tse - tsw) * s.ty);
            giy += dtc * ((tsw - tnw) * ex + (tse - tne) * s.tx);
        }
        dcs = dc; snx = nx; sny = ny; snz = nz;                  // dL/dlights = dc * bands(normal): formed at the end
        if (hf >= 0) {
            const float du = gix * s.mx * ((float)a.Wt / 2.f) * 2.f;
            const float dv = giy * s.my * ((float)a.Ht / 2.f) * -2.f;
            const float dnx = dc * (((MM_SH_C1 * L[1] + MM_SH_C4 * ny * L[4]) + MM_SH_C7 * nz * L[7]) + 2.f * MM_SH_C8 * nx * L[8]);
            const float dny = dc * (((MM_SH_C1 * L[3] + MM_SH_C4 * nx * L[4]) + MM_SH_C4 * nz * L[5]) - 2.f * MM_SH_C8 * ny * L[8]);
            const float dnz = dc * (((MM_SH_C1 * L[2] + MM_SH_C4 * ny * L[5]) + 2.f * MM_SH_C6 * nz * L[6]) + MM_SH_C7 * nx * L[7]);
            // K2 (Appendix A.1): this pixel's contribution to its face's corner and normal gradients; corner features are
            // (1, u_k, v_k, n).  Left per pixel; the face gather only has to add them up.
            const float gnn = (dnx * n0 + dny * n1) + dnz * n2;
            const float G0 = ((dm + du * fu[0]) + dv * fu[1]) + gnn;
            const float G1 = ((dm + du * fu[2]) + dv * fu[3]) + gnn;
            const float G2 = ((dm + du * fu[4]) + dv * fu[5]) + gnn;
            const float Gm = (w0 * G0 + w1 * G1) + w2 * G2;
            const float inrm = 1.f / nrm;
            const float dw0 = (G0 - Gm) * inrm, dw1 = (G1 - Gm) * inrm, dw2 = (G2 - Gm) * inrm;
            const float aex = p0.x - x0, aey = p0.y - y0, bex = p0.z - x0, bey = p0.w - y0, cex = p1.x - x0, cey = p1.y - y0;
            const float4 k0 = make_float4((dw1 * (-cey) + dw2 * bey) * a.mult, (dw1 * cex + dw2 * (-bex)) * a.mult,
                                          (dw0 * cey + dw2 * (-aey)) * a.mult, (dw0 * (-cex) + dw2 * aex) * a.mult);
            const float4 k1 = make_float4((dw0 * (-bey) + dw1 * aey) * a.mult, (dw0 * bex + dw1 * (-aex)) * a.mult,
                                          (w0 * dnx + w1 * dnx) + w2 * dnx, (w0 * dny + w1 * dny) + w2 * dny);
            const float k2 = (w0 * dnz + w1 * dnz) + w2 * dnz;
            a.gp[pix * 2 + 0] = k0; a.gp[pix * 2 + 1] = k1; a.gp2[pix] = k2;
            m2 = fmaxf(fmaxf(fmaxf(fabsf(k0.x), fabsf(k0.y)), fmaxf(fabsf(k0.z), fabsf(k0.w))),
                       fmaxf(fmaxf(fmaxf(fabsf(k1.x), fabsf(k1.y)), fmaxf(fabsf(k1.z), fabsf(k1.w))), fabsf(k2)));
            if (dtcv[0] != 0.f || dtcv[1] != 0.f || dtcv[2] != 0.f) {
                rec.xy = (unsigned)s.x0 | ((unsigned)s.y0 << 16); rec.tx = s.tx; rec.ty = s.ty;
                rec.d0 = dtcv[0]; rec.d1 = dtcv[1]; rec.d2 = dtcv[2];
                // texture tiles under the bilinear footprint: up to 2x2 when it straddles a tile border
                bx0 = (unsigned)(s.x0 / MM_UV_TILE); by0 = (unsigned)(s.y0 / MM_UV_TILE);
                bx1 = (s.x1 < a.Wt ? s.x1 : s.x0) / MM_UV_TILE; by1 = (s.y1 < a.Ht ? s.y1 : s.y0) / MM_UV_TILE;
            }
        }
    }
    // The pixel's texture-gradient record goes to the pixel's OWN place, screen-tile-major (tile slot = block * 4 + quadrant, lane = pixel: the
    // wave's 64 records are 1.5 KB in a row), and the wave leaves the box of texture tiles its footprints touch.  No lists, no slot atomics, no
    // offsets to wait for, nothing that can overflow: the texture tile's workgroup (mm_backward.hip) reads the boxes and streams the records of
    // the screen tiles whose box holds it.  A tile without any record writes its (empty) box only.
    {
        const size_t slot = (size_t)b * a.nst + (size_t)blk * 4 + wave;
        unsigned box = MM_TBOX_EMPTY;
        if (any_covered) {                                       // (wave-uniform)
            const unsigned x0m = wave_min_u32(bx0), y0m = wave_min_u32(by0);
            const int x1m = wave_max_i32(bx1), y1m = wave_max_i32(by1);
            if (x0m != 255u) {                                   // some lane has a record
                box = x0m | (y0m << 8) | ((unsigned)x1m << 16) | ((unsigned)y1m << 24);
                a.trec[slot * 64 + lane] = rec;                  // (every lane: a pixel without one says so, xy = MM_TREC_NONE)
            }
        }
        if (lane == 0) a.tbox[slot] = box;
    }
    // (the slots are on their way: the wave's light-gradient sums are formed meanwhile, the records stored after them)
    // d lights: wave butterfly -> one LDS row per wave -> fixed-order partial of this workgroup (summed by vertex_bwd)
    float dl[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) dl[i] = 0.f;
    if (any_covered) {
        float bnd9[9];
        sh_bands(snx, sny, snz, bnd9);
#pragma unroll
        for (int i = 0; i < 9; ++i) dl[i] = wave_sum(dcs * bnd9[i]);
    } else { dl[0] = wave_sum(dcs * MM_SH_C0); dl[6] = wave_sum(dcs * (0.f - MM_SH_C6B)); }     // the other seven are zero
    out.m2 = wave_max(m2); out.m4 = wave_max(m4);
    if (a.options & MM_OPT_SH_ORDER_XYZ) { const float tmp = dl[2]; dl[2] = dl[3]; dl[3] = tmp; }   // back to the user's light order
#pragma unroll
    for (int i = 0; i < 9; ++i) out.dl[i] = dl[i];
    // the wave's partial of dL/dlights: one 48-byte row per tile slot (summed in index order by the vertex backward)
    if (lane == 0) {
        float4* row = reinterpret_cast<float4*>(a.dl_part + ((size_t)b * a.nst + (size_t)blk * 4 + wave) * 12);
        row[0] = make_float4(dl[0], dl[1], dl[2], dl[3]); row[1] = make_float4(dl[4], dl[5], dl[6], dl[7]); row[2] = make_float4(dl[8], 0.f, 0.f, 0.f);
    }
}

}  // namespace mm
