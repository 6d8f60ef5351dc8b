// mm_pixel_bwd.hip -- pixel-major pass of the render path's backward for gfx950 (see mm_backward.hip for the scheme, mm_backward.h for why
// this half is compiled with the forward's floating-point flags).
#include "mm_backward.h"

MM_TIMELINE_STORAGE(pixel_bwd)
MM_PP_STORAGE(pixel_bwd)        // 0 loss totals + g4, 1 shading recompute + stores, 2 record append, 3 dlights reduction

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
// (MM_PLAN_WGS workgroups per image where faces are many, else one: each counts every face -- cheap, from LDS -- and writes the items of its share)
__device__ inline void plan_sweep_items(const BwdArgs& a, int b, int q) {
    const int nwg = a.plan_wgs;                                   // 1 or MM_PLAN_WGS
    __shared__ int s_wave[MM_PLAN_WGS][4];
    // the faces' chunk counts at the base chunk size are staged in LDS (2 bytes a face, read once, coalesced, eight loads in flight per
    // thread): with thousands of faces per thread-range the passes below were a chain of dependent trips to memory, one per face.
    // ceil(ceil(n / c) / 2^k) = ceil(n / (c 2^k)): the doubled chunk sizes need nothing else.
    __shared__ unsigned short s_nch[MM_PLAN_LDS_FACES];
    const int tid = threadIdx.x;
    // First (the pixel workgroups behind this one in the grid want it two trips to memory into their lives): where each texture tile's record list
    // starts in the image's packed array = exclusive scan of the forward's per-tile counts.  Stored + 1: the words are zero until now (cleared with
    // the backward's counters), which is how a pixel lane that got there first knows to ask again.
    if (q == 0) {
#ifdef MM_DBG_LATE_TOFF                                         // (test builds: a plan workgroup that gets going ~0.3 ms late -- every pixel lane has given
        for (int i = 0; i < 100; ++i) __builtin_amdgcn_s_sleep(127);   //  up waiting by then and formed its offset itself; the texture gather still finds these)
#endif
        const int nt = a.ntiles_, per4 = (nt + 255) >> 8, t0 = tid * per4;
        const int* cnt = a.trcnt + (size_t)b * nt;
        int mine = 0;
        for (int i = 0; i < per4; ++i) mine += t0 + i < nt ? cnt[t0 + i] : 0;
        int tot;
        int run = wave_prefix_excl(mine, tid & 63, tot);
        if ((tid & 63) == 0) s_wave[0][tid >> 6] = tot;
        __syncthreads();
        for (int w2 = 0; w2 < (tid >> 6); ++w2) run += s_wave[0][w2];
        for (int i = 0; i < per4; ++i) {
            if (t0 + i < nt) {
                __hip_atomic_store(a.toff + (size_t)b * nt + t0 + i, run + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                run += cnt[t0 + i];
            }
        }
        __syncthreads();                                         // (s_wave is used again below)
    }
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
#ifndef MM_TOFF_SPIN_MAX
#define MM_TOFF_SPIN_MAX (1 << 12)   // polls of a tile's list offset (each a trip to memory: a few ms in all) before the lane forms the offset itself
#endif
#ifndef MM_PIXEL_LB
#define MM_PIXEL_LB 5             // waves per SIMD the register allocation is held to: 96 VGPRs without spills (the light gradients are carried as scalar + normal, not
#endif                            // as nine products); 5 workgroups of 28.9 KB LDS (the plan workgroups' staging) fit a CU as well
// kContour: the fused loss carries recon_data's contour term (MMRenderDesc.fused_contour > 0).  The reference's default is --lambda_contour 0
// (train.py:115, trainer.py:441): the default caller gets the instantiation without the term's code (24 vector instructions per wave and its
// registers), chosen by the host.
// kDeferred: DEFERRED fusion (BwdArgs::ltot as deferred_totals, MMRenderDesc.fused_totals): recon_data ran on its own on the image this render wrote; dL/d rgba is formed
// here with mm_recon_data_backward's expressions (csrc/mm_loss.hip: recon_bwd_kernel), in their order, from that call's per-image totals -- the bits
// that kernel would have written -- plus the caller's grad_rgba if there is one.  Never together with kContour.
template <bool kNoMask, bool kContour, bool kDeferred>
__global__ __launch_bounds__(256, MM_PIXEL_LB) void pixel_bwd_kernel(BwdArgs a) {
    MM_TIMELINE_BEGIN();
    __shared__ float s_dl[MM_BLOCK_WAVES][9];
    __shared__ float s_gm[MM_BLOCK_WAVES][2];
    if ((int)blockIdx.x < a.plan_wgs * a.B) { plan_sweep_items(a, blockIdx.x / a.plan_wgs, blockIdx.x % a.plan_wgs); return; }   // (workgroup-uniform)
    int b, blk;
    map_block(blockIdx.x - a.plan_wgs * a.B, a.B, a.blocks_per_image, b, blk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bx = blk % a.blocks_x, by = blk / a.blocks_x;
    const int px = bx * MM_BLOCK_PX + (wave & 1) * MM_TILE + (lane & 7), py = by * MM_BLOCK_PX + (wave >> 1) * MM_TILE + (lane >> 3);
    const bool in_img = px < a.W && py < a.H;
    const float x0 = pixel_x_k(px, a.W, a.kx), y0 = pixel_y_k(py, a.H, a.ky);                  // (host-formed IEEE quotients: the forward's centres)
    const size_t hw = (size_t)a.H * a.W, pin = (size_t)py * a.W + px;
    const size_t pix = (size_t)b * hw + pin;
    if (blk == 0 && threadIdx.x == 0) a.ticket[b] = 0u;           // arrival counter of the vertex backward, used after this kernel
    // The pass is a chain of dependent trips to memory; it is written so that four remain: (1) everything addressed by the pixel
    // alone -- face_idx, prediction, ground truth, background; (2) what the winner's id addresses -- geometry, normal, corner uvs;
    // (3) the twelve texels, unconditionally from clamped addresses; (4) the record-slot atomics.  (Loads left inside per-lane
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
    float gsw = 0.f, cnt = 1.f;                                  // kDeferred: gs * image_weight and B*3*H*W, as recon_bwd_kernel forms them
    if (kDeferred) {
        const float gs = a.grad_loss ? a.grad_loss[0] : 1.f;
        gsw = gs * a.image_weight;
        cnt = (float)a.B * 3.f * (float)a.H * (float)a.W;
        const float up = deferred_totals(a)[b * 4 + 1], U = deferred_totals(a)[b * 4 + 2] + 1e-10f;
        if (in_img) {
            hf = a.face_idx[pix];
            const float* g = a.gt + (size_t)b * 4 * hw;
            const float gm = g[3 * hw + pin];
            gmv = gm;
#pragma unroll
            for (int c = 0; c < 3; ++c) gi3[c] = g[c * hw + pin] * gm + 1.f * (1.f - gm);
            g4.w = gs * (-(1.f / (float)a.B) * (gm / U - up * (1.f - gm) / (U * U)));
            if (a.grad_rgba) {                                   // the image's other consumers (autograd would have added the two gradients)
                const float4 ext = *(const float4*)(a.grad_rgba + pix * 4);
                g4.x = ext.x; g4.y = ext.y; g4.z = ext.z; g4.w += ext.w;
            }
        }
    } else if (fused) {
        // the image's totals are exact integer sums left by its raster waves
        float l1s, up, un;
        loss_totals(a.ltot, b, l1s, up, un);
        const float U = un + 1e-10f;
        const float gs = a.grad_loss ? a.grad_loss[0] : 1.f;
        // everything that is the same for all pixels of the image is folded into three coefficients (wave-uniform arithmetic
        // once, instead of four divisions per lane): dL/dpred_c = kl1 * sign * gm,  dL/dalpha = ka * gm + kb * (1 - gm)
        kl1 = gs * a.image_weight / ((float)a.B * 3.f * (float)a.H * (float)a.W);
        const float ka = -gs / ((float)a.B * U), kb = gs * up / ((float)a.B * U * U);
        if (in_img) {
            hf = a.face_idx[pix];
            const float* g = a.gt + (size_t)b * 4 * hw;
            const float gm = g[3 * hw + pin];
            gmv = gm;
#pragma unroll
            for (int c = 0; c < 3; ++c) gi3[c] = g[c * hw + pin] * gm + 1.f * (1.f - gm);
            g4.w = ka * gm + kb * (1.f - gm);
        }
        if (kContour) {                                          // the contour term, networks.py:379-388 (forward: contour_term, mm_raster_common.h)
            // d/dalpha of  kc * sum_p (|alpha_p - alpha_s(p)| - |gm_p - gm_s(p)|)^2,  s(p) = top-left pixel of p's 4x4 block = lane (lane & 0x24) of this
            // wave: pixel p gets +g_p, its block's corner pixel -sum of the block's g (its own g is 0: |0| has gradient 0, as torch.abs has).
            // alpha is re-formed from the saved soft-mask state exactly as shade_store formed it: covered 1, else 1 - keepprod.
            const float kc = gs * a.contour / ((float)a.B * (float)a.H * (float)a.W);
            float al = 0.f;
            if (in_img) { const float sx = a.soft[pix].x; al = hf >= 0 ? 1.f : 1.f - (sx > 0.f ? sx : 0.f); }
            const int src = lane & 0x24;
            const float as = __shfl(al, src, 64), gms = __shfl(gmv, src, 64);
            float gc = 0.f;
            if (in_img) {
                const float dd = al - as, cp = fabsf(dd), cg = fabsf(gmv - gms);
                gc = (kc * 2.f * (cp - cg)) * (dd > 0.f ? 1.f : (dd < 0.f ? -1.f : 0.f));
            }
            float bs = gc;                                       // the 4x4 block's sum, fixed order: columns (lane bits 0, 1), then rows (bits 3, 4)
            bs += __shfl_xor(bs, 1, 64); bs += __shfl_xor(bs, 2, 64); bs += __shfl_xor(bs, 8, 64); bs += __shfl_xor(bs, 16, 64);
            if (in_img) g4.w += lane == src ? -bs : gc;
        }
    } else if (in_img) { g4 = *(const float4*)(a.grad_rgba + pix * 4); hf = a.face_idx[pix]; }
    const float gin[3] = {g4.x, g4.y, g4.z};
    // dL/d(colour c of this pixel) given its un-clamped value `pre`: the caller's gradient, or the fused loss's (the forward's clamp and
    // masking expressions, shade_store / shade_empty_tiles + networks.py:370-377)
    auto grad_colour = [&](int c, float pre) -> float {
        if (!fused) return gin[c];
        const float pc = pre < 0.f ? 0.f : (pre > 1.f ? 1.f : pre);
        const float pi = pc * gmv + 1.f * (1.f - gmv);
        const float df = pi - gi3[c], sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
        if (kDeferred) return gsw * sg * gmv / cnt + gin[c];     // recon_bwd_kernel: gs * image_weight * sg * gm / cnt  (+ the caller's own gradient, 0 if none)
        return kl1 * sg * gmv;
    };
    float m2 = 0.f, m4 = 0.f;                                    // this lane's largest |K2 number| / |dL/dalpha|: the gather's fixed-point scale
    if (in_img && hf < 0) { a.gp2[pix] = g4.w; m4 = fabsf(g4.w); }   // the face gather (K4) needs dL/dalpha of uncovered pixels
    // dL/dlights of this pixel = dcs * sh_bands(normal): kept as the scalar and the normal (4 registers, not 9, across the record append below)
    float dcs = 0.f, snx = 0.f, sny = 0.f, snz = 0.f;
    TexRecord rec; rec.xy = 0; rec.tx = rec.ty = rec.d0 = rec.d1 = rec.d2 = 0.f;
    int rtile[4] = {-1, -1, -1, -1};

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
                const int tcx0 = s.x0 / MM_UV_TILE, tcy0 = s.y0 / MM_UV_TILE;
                const int tcx1 = (s.x1 < a.Wt ? s.x1 : s.x0) / MM_UV_TILE, tcy1 = (s.y1 < a.Ht ? s.y1 : s.y0) / MM_UV_TILE;
                rtile[0] = tcy0 * a.ntx + tcx0;
                rtile[1] = tcx1 != tcx0 ? tcy0 * a.ntx + tcx1 : -1;
                rtile[2] = tcy1 != tcy0 ? tcy1 * a.ntx + tcx0 : -1;
                rtile[3] = (tcx1 != tcx0 && tcy1 != tcy0) ? tcy1 * a.ntx + tcx1 : -1;
            }
        }
    }
    // append the records: one returning atomic per (wave, distinct tile), lanes of the same tile take consecutive slots of the tile's list, which
    // starts at the tile's offset in the image's packed record array -- read by the group's leader in the same trip as its atomic (written by the
    // image's plan workgroup at the top of its life, + 1: a zero means "not yet", the first microseconds of the launch, and is asked for again).
    // The grouping is pure lane arithmetic; all leaders then issue their atomics in ONE instruction per footprint corner (and the four corners'
    // atomics are in flight together), so a wave pays one fabric round trip, not one per tile.
    int leader[4], rank[4], base[4], toff[4], room[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        leader[c] = -1; rank[c] = 0; base[c] = 0; toff[c] = 1; room[c] = 0;
        if (!any_covered) continue;                              // wave-uniform: nothing to append
        int size = 0;
        unsigned long long pending = __ballot(rtile[c] >= 0);
        while (pending) {
            const int ld = __ffsll((unsigned long long)pending) - 1;
            const int tile = __builtin_amdgcn_readlane(rtile[c], ld);   // (`ld` is wave-uniform: a scalar lane select, no LDS-crossbar round trip per tile)
            const unsigned long long m = __ballot(rtile[c] == tile);
            if (rtile[c] == tile) { leader[c] = ld; rank[c] = ballot_rank(m); size = __popcll(m); }
            pending &= ~m;
        }
        if (leader[c] == lane) {
            toff[c] = __hip_atomic_load(a.toff + (size_t)b * a.ntiles_ + rtile[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            room[c] = a.trcnt[(size_t)b * a.ntiles_ + rtile[c]];    // what the forward counted for this tile: the length of its list
            base[c] = atomicAdd(a.tcur + (size_t)b * a.ntiles_ + rtile[c], size);
        }
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
    m2 = wave_max(m2); m4 = wave_max(m4);
    if (lane == 0) {
        if (a.options & MM_OPT_SH_ORDER_XYZ) { const float tmp = dl[2]; dl[2] = dl[3]; dl[3] = tmp; }   // back to the user's light order
#pragma unroll
        for (int i = 0; i < 9; ++i) s_dl[wave][i] = dl[i];
        s_gm[wave][0] = m2; s_gm[wave][1] = m4;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (!any_covered) break;
        // (the image's plan workgroup has a LOWER workgroup index and writes the offsets first thing: it is in flight before this wave exists,
        //  and the wait below is the first microseconds of a launch.  Progress does NOT depend on that order: the offset is a pure function of
        //  the forward's per-tile counts, which are complete before this launch -- should it not arrive within the bound (a dispatcher that
        //  does not start workgroups in index order, CU masking, a debugger), the lane adds the counts up itself: the same number, later.)
        int spins = 0;
        while (__builtin_expect(leader[c] == lane && toff[c] == 0, 0)) {
            if (++spins > MM_TOFF_SPIN_MAX) {
                const int* cnt = a.trcnt + (size_t)b * a.ntiles_;
                int run = 1;                                     // (stored + 1, as the plan workgroup stores it)
                for (int t = 0; t < rtile[c]; ++t) run += cnt[t];
                toff[c] = run;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
            toff[c] = __hip_atomic_load(a.toff + (size_t)b * a.ntiles_ + rtile[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        room[c] = toff[c] == 0 ? 0 : room[c] - base[c];           // places left in the tile's list from this group's first one (<= 0: none)
        base[c] += toff[c] - 1;
    }
    int ndrop = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (!any_covered) break;
        const int bs = __shfl(base[c], leader[c] < 0 ? lane : leader[c], 64), rm = __shfl(room[c], leader[c] < 0 ? lane : leader[c], 64);
        if (rtile[c] >= 0) {
            const int pos = bs + rank[c];
            // inside the image's array AND inside the tile's own list (the forward's count is an upper bound of what this pass appends as
            // long as both recompute the same footprints; a record beyond it would land in the NEXT tile's list: dropped and reported instead)
            if (pos < a.trcap && rank[c] < rm) a.trec[(size_t)b * a.trcap + pos] = rec;
            else ++ndrop;                                        // counted, and the texture gather poisons the image's gradient
        }
    }
    if (__builtin_expect(__ballot(ndrop != 0) != 0ull, 0)) { if (ndrop) atomicAdd(a.tdrop + b, ndrop); }

    __syncthreads();
    if (threadIdx.x >= 64 && threadIdx.x < 66) {                 // non-negative floats order like their bit patterns: integer max, one atomic per
        const int k = threadIdx.x - 64;                          // workgroup and kind (NaN / inf gradients end up as an inf scale = zero sums)
        const float m = fmaxf(fmaxf(s_gm[0][k], s_gm[1][k]), fmaxf(s_gm[2][k], s_gm[3][k]));
        // one atomic per workgroup and kind, spread over MM_GSHARD words per image on separate 32-byte sectors (thousands of
        // workgroups per image on ONE word queue at the memory side: measured +75 % on this kernel at 512x512)
        if (m > 0.f) atomicMax(a.gmax + ((size_t)b * MM_GSHARD + (blk & (MM_GSHARD - 1))) * 8 + k, __float_as_uint(m));
    }
    if (threadIdx.x < 9)
        a.dl_part[((size_t)b * a.blocks_per_image + blk) * 12 + threadIdx.x] =
            ((s_dl[0][threadIdx.x] + s_dl[1][threadIdx.x]) + s_dl[2][threadIdx.x]) + s_dl[3][threadIdx.x];
    MM_TIMELINE_END(pixel_bwd);
}

int launch_pixel_bwd(const BwdArgs& a, const MMRenderDesc* d, hipStream_t s) {
    ProfScope p(d->prof_events, MM_PROF_PIXEL_BWD, s);
    dim3 grid(a.blocks_per_image * d->B + a.plan_wgs * d->B);    // + the plan workgroups, in front
    const bool contour = a.gt != nullptr && a.contour > 0.f;
    if (a.options & MM_INT_DEFERRED) {                           // deferred fusion (never with the contour term: check_render)
        if (d->no_mask) hipLaunchKernelGGL((pixel_bwd_kernel<true, false, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((pixel_bwd_kernel<false, false, true>), grid, dim3(256), 0, s, a);
        return MM_OK;
    }
    if (d->no_mask) { if (contour) hipLaunchKernelGGL((pixel_bwd_kernel<true, true, false>), grid, dim3(256), 0, s, a); else hipLaunchKernelGGL((pixel_bwd_kernel<true, false, false>), grid, dim3(256), 0, s, a); }
    else { if (contour) hipLaunchKernelGGL((pixel_bwd_kernel<false, true, false>), grid, dim3(256), 0, s, a); else hipLaunchKernelGGL((pixel_bwd_kernel<false, false, false>), grid, dim3(256), 0, s, a); }
    return MM_OK;
}

}  // namespace mm
