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
//                  It also APPENDS each covered pixel's texture contribution to the record list of every texture tile its
//                  bilinear footprint touches (one returning atomic per wave and distinct tile, lanes take consecutive slots).
//   2a. texture_gather  one workgroup per (image, 32x32-texel texture tile) streams its record list into LDS accumulators
//                  (LDS float adds) and writes the tile once with plain stores: no zero-fill pass over grad_textures.
//   2b. face_accum  one 1024-thread workgroup per (image, face range) keeps its faces' gradients in LDS: adds the K2
//                  contributions of all covered pixels, replays the forward's soft-mask batches for K4, stores once per face.
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
    int* tcnt; TexRecord* trec; TexSpill* tspill; int ntiles_;
    const int* sb_cnt; const uint64_t* sb_mask; const int* sb_ids; const int* sb_over; int tiles8_x, tiles8;
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
    TexRecord rec; rec.xy = 0; rec.tx = rec.ty = rec.d0 = rec.d1 = rec.d2 = 0.f;
    int rtile[4] = {-1, -1, -1, -1};

    if (in_img && (hf >= 0 || kNoMask)) {
        // recompute the forward quantities of this pixel (only face_idx and the soft-mask state were saved)
        float w0 = 0.f, w1 = 0.f, w2 = 0.f, nrm = 1.f, m = 0.f, u = 0.f, v = 0.f, nx = 0.f, ny = 0.f, nz = 0.f;
        float fu[6] = {0, 0, 0, 0, 0, 0}, n0 = 0.f, n1 = 0.f, n2 = 0.f;
        float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0;
        if (hf >= 0) {
            const float4* geo = a.geo + ((size_t)b * a.F + hf) * 3;
            p0 = geo[0]; p1 = geo[1];
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
            // K2 (Appendix A.1): this pixel's contribution to its face's corner and normal gradients; corner features are
            // (1, u_k, v_k, n).  Stored per pixel; the per-image accumulation pass adds them up per face.
            const float gnn = (dnx * n0 + dny * n1) + dnz * n2;
            const float G0 = ((dm + du * fu[0]) + dv * fu[1]) + gnn;
            const float G1 = ((dm + du * fu[2]) + dv * fu[3]) + gnn;
            const float G2 = ((dm + du * fu[4]) + dv * fu[5]) + gnn;
            const float Gm = (w0 * G0 + w1 * G1) + w2 * G2;
            const float dw0 = (G0 - Gm) / nrm, dw1 = (G1 - Gm) / nrm, dw2 = (G2 - Gm) / nrm;
            const float aex = p0.x - x0, aey = p0.y - y0, bex = p0.z - x0, bey = p0.w - y0, cex = p1.x - x0, cey = p1.y - y0;
            a.gp0[pix] = make_float4((dw1 * (-cey) + dw2 * bey) * a.mult, (dw1 * cex + dw2 * (-bex)) * a.mult,
                                     (dw0 * cey + dw2 * (-aey)) * a.mult, (dw0 * (-cex) + dw2 * aex) * a.mult);
            a.gp1[pix] = make_float4((dw0 * (-bey) + dw1 * aey) * a.mult, (dw0 * bex + dw1 * (-aex)) * a.mult,
                                     (w0 * dnx + w1 * dnx) + w2 * dnx, (w0 * dny + w1 * dny) + w2 * dny);
            a.gp2[pix] = (w0 * dnz + w1 * dnz) + w2 * dnz;
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
    // append the records: one returning atomic per (wave, distinct tile), lanes of the same tile take consecutive slots
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        unsigned long long pending = __ballot(rtile[c] >= 0);
        while (pending) {
            const int leader = __ffsll((unsigned long long)pending) - 1;
            const int tile = __shfl(rtile[c], leader, 64);
            const unsigned long long m = __ballot(rtile[c] == tile);
            int base = 0;
            if (lane == leader) base = atomicAdd(a.tcnt + (size_t)b * a.ntiles_ + tile, __popcll(m));
            base = __shfl(base, leader, 64);
            if (rtile[c] == tile) {
                const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
                if (slot < MM_TREC_CAP) a.trec[((size_t)b * a.ntiles_ + tile) * MM_TREC_CAP + slot] = rec;
                else {                                          // full tile (rare): the image's spill list
                    const int os = atomicAdd(a.tcnt + (size_t)a.B * a.ntiles_ + b, 1);
                    TexSpill sp; sp.r = rec; sp.tile = tile; sp.pad = 0;
                    a.tspill[(size_t)b * 4 * a.H * a.W + os] = sp;
                }
            }
            pending &= ~m;
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
// 2a. texture gradient: one workgroup per (image, 32x32-texel tile) streams the records the pixel pass appended for the
//     tile into LDS accumulators and writes every texel of the tile once (no zero-fill pass, no HBM atomics).
// ---------------------------------------------------------------------------------------------------------------------
#define MM_TS MM_UV_TILE

__device__ inline void tex_accumulate(const BwdArgs& a, float (*s_acc)[MM_TS * MM_TS], const TexRecord& rc, int tx0, int ty0) {
    const int x0 = (int)(rc.xy & 0xFFFFu), y0 = (int)(rc.xy >> 16);
    const int lx0 = x0 - tx0, lx1 = lx0 + 1, ly0 = y0 - ty0, ly1 = ly0 + 1;
    const bool cx0 = lx0 >= 0 && lx0 < MM_TS, cx1 = lx1 >= 0 && lx1 < MM_TS && x0 + 1 < a.Wt;
    const bool cy0 = ly0 >= 0 && ly0 < MM_TS, cy1 = ly1 >= 0 && ly1 < MM_TS && y0 + 1 < a.Ht;
    const float ex = 1.f - rc.tx, ey = 1.f - rc.ty;
    const float wnw = ex * ey, wne = rc.tx * ey, wsw = ex * rc.ty, wse = rc.tx * rc.ty;
    const float dt[3] = {rc.d0, rc.d1, rc.d2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (dt[c] != 0.f) {
            if (cx0 && cy0) atomicAdd(&s_acc[c][ly0 * MM_TS + lx0], dt[c] * wnw);
            if (cx1 && cy0) atomicAdd(&s_acc[c][ly0 * MM_TS + lx1], dt[c] * wne);
            if (cx0 && cy1) atomicAdd(&s_acc[c][ly1 * MM_TS + lx0], dt[c] * wsw);
            if (cx1 && cy1) atomicAdd(&s_acc[c][ly1 * MM_TS + lx1], dt[c] * wse);
        }
    }
}

__global__ __launch_bounds__(256) void texture_gather_kernel(BwdArgs a) {
    __shared__ float s_acc[3][MM_TS * MM_TS];
    const int ntiles = a.ntx * a.nty;
    int b, T;
    map_block(blockIdx.x, a.B, ntiles, b, T);
    const int tid = threadIdx.x;
    for (int i = tid; i < 3 * MM_TS * MM_TS; i += 256) (&s_acc[0][0])[i] = 0.f;
    __syncthreads();
    const int tx0 = (T % a.ntx) * MM_TS, ty0 = (T / a.ntx) * MM_TS;
    const int nrec = a.tcnt[(size_t)b * ntiles + T];
    const TexRecord* recs = a.trec + ((size_t)b * ntiles + T) * MM_TREC_CAP;
    for (int r = tid; r < min(nrec, MM_TREC_CAP); r += 256) tex_accumulate(a, s_acc, recs[r], tx0, ty0);
    if (nrec > MM_TREC_CAP) {                                    // the tile's list was full: its remaining records are in the spill list
        const int nsp = a.tcnt[(size_t)a.B * ntiles + b];
        const TexSpill* sp = a.tspill + (size_t)b * 4 * a.H * a.W;
        for (int r = tid; r < nsp; r += 256) if (sp[r].tile == T) tex_accumulate(a, s_acc, sp[r].r, tx0, ty0);
    }
    __syncthreads();
    for (int i = tid; i < 3 * MM_TS * MM_TS; i += 256) {
        const int c = i / (MM_TS * MM_TS), r = i - c * (MM_TS * MM_TS);
        const int ly = r / MM_TS, lx = r - ly * MM_TS;
        const int x = tx0 + lx, y = ty0 + ly;
        if (x < a.Wt && y < a.Ht) a.grad_textures[(((size_t)b * 3 + c) * a.Ht + y) * a.Wt + x] = s_acc[c][r];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 2b. per-face gradients: one 1024-thread workgroup per (image, range of <= Fr faces) keeps dL/d(face corner xy) and
//     dL/d(face normal) of its faces in LDS and adds into them with LDS float adds:
//       K2: every covered pixel's nine contributions, computed by the pixel pass;
//       K4 (Appendix A.2): for every uncovered pixel with a live soft-mask gradient, the faces it took in the forward --
//           replayed from the (tile, batch) masks and id tables the forward kept, one wave per 8x8 tile.
//     Then one plain store per face.  Images in which a tile needed more than MM_SB_CAP batches redo the forward's rule
//     directly (all faces in index order, first knum whose inflated box holds the pixel).
// ---------------------------------------------------------------------------------------------------------------------
#define MM_FA_THREADS 1024
#define MM_FA_WAVES (MM_FA_THREADS / 64)

struct __attribute__((aligned(16))) K4Stage { float4 p0[64], p1[64]; int ids[64]; };

// adds d(soft)/d(face f) for pixel (x0,y0): ga = dL/dalpha, sq = forward product state, p0/p1 = face xy
__device__ inline void k4_accumulate(const BwdArgs& a, float* acc9, float x0, float y0, const float4& p0, const float4& p1, float ga, float sq) {
    const float s2 = a.mult * a.mult;
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
    if (gd == 0.f) return;
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
    const int iu = e * 2, iv = (e == 2 ? 0 : e + 1) * 2;          // edge e runs from corner e to corner (e+1)%3
    atomicAdd(acc9 + iu, gd * dux * a.mult); atomicAdd(acc9 + iu + 1, gd * duy * a.mult);
    atomicAdd(acc9 + iv, gd * dvx * a.mult); atomicAdd(acc9 + iv + 1, gd * dvy * a.mult);
}

__global__ __launch_bounds__(MM_FA_THREADS) void face_accum_kernel(BwdArgs a, int Fr, int R) {
    extern __shared__ __attribute__((aligned(16))) char s_raw[];
    K4Stage* stage = (K4Stage*)s_raw;                            // one per wave
    float* acc = (float*)(s_raw + MM_FA_WAVES * sizeof(K4Stage)); // (Fr, 9)
    const int b = blockIdx.x / R, r = blockIdx.x - b * R;
    const int f_lo = r * Fr, f_hi = min(a.F, f_lo + Fr), nf = f_hi - f_lo;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hw = a.H * a.W;
    for (int i = tid; i < nf * 9; i += MM_FA_THREADS) acc[i] = 0.f;
    __syncthreads();

    // ---- K2: stream the image's pixels, four per thread per trip so that the loads of a trip are in flight together
    for (int base = 0; base < hw; base += MM_FA_THREADS * 4) {
        int fi[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int p = base + k * MM_FA_THREADS + tid; fi[k] = p < hw ? a.face_idx[(size_t)b * hw + p] : -1; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (fi[k] >= f_lo && fi[k] < f_hi) {
                const size_t pix = (size_t)b * hw + base + k * MM_FA_THREADS + tid;
                const float4 q0 = a.gp0[pix], q1 = a.gp1[pix];
                const float q2 = a.gp2[pix];
                float* o = acc + (size_t)(fi[k] - f_lo) * 9;
                atomicAdd(o + 0, q0.x); atomicAdd(o + 1, q0.y); atomicAdd(o + 2, q0.z); atomicAdd(o + 3, q0.w);
                atomicAdd(o + 4, q1.x); atomicAdd(o + 5, q1.y); atomicAdd(o + 6, q1.z); atomicAdd(o + 7, q1.w);
                atomicAdd(o + 8, q2);
            }
        }
    }

    // ---- K4: one wave per 8x8 pixel tile, lane = pixel
    const bool slow = a.sb_over[b] != 0;
    K4Stage* st = stage + wave;
    for (int tile = wave; tile < a.tiles8; tile += MM_FA_WAVES) {
        const int nb = a.sb_cnt[(size_t)b * a.tiles8 + tile];
        if (nb == 0 && !slow) continue;
        const int px = (tile % a.tiles8_x) * MM_TILE + (lane & 7), py = (tile / a.tiles8_x) * MM_TILE + (lane >> 3);
        const bool in_img = px < a.W && py < a.H;
        const size_t pix = (size_t)b * hw + (size_t)py * a.W + px;
        float sq = 0.f, ga = 0.f;
        if (in_img && a.face_idx[pix] < 0) { sq = a.softq[pix]; ga = a.grad_rgba[pix * 4 + 3]; }
        const bool active = sq != 0.f && sq != 1.f && ga != 0.f;
        if (__ballot(active) == 0) continue;
        const float x0 = pixel_x(px, a.W, a.mult), y0 = pixel_y(py, a.H, a.mult);
        if (!slow) {
            const size_t row = ((size_t)b * a.tiles8 + tile) * MM_SB_CAP;
            for (int i = 0; i < nb; ++i) {
                uint64_t m = active ? a.sb_mask[(row + i) * 64 + lane] : 0;
                const int id = a.sb_ids[(row + i) * 64 + lane];
                st->ids[lane] = id;
                if (id >= 0) { st->p0[lane] = a.geo[((size_t)b * a.F + id) * 3]; st->p1[lane] = a.geo[((size_t)b * a.F + id) * 3 + 1]; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                while (__ballot(m != 0)) {
                    if (m) {
                        const int j = __ffsll((unsigned long long)m) - 1;
                        m &= m - 1;
                        const int f = st->ids[j];
                        if (f >= f_lo && f < f_hi) k4_accumulate(a, acc + (size_t)(f - f_lo) * 9, x0, y0, st->p0[j], st->p1[j], ga, sq);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        } else if (active) {
            // slow path: the forward's rule restated -- faces in index order, first knum whose inflated box holds the pixel
            int cnt = 0;
            for (int f = 0; f < a.F && cnt < a.knum; ++f) {
                const float4 p0 = a.geo[((size_t)b * a.F + f) * 3], p1 = a.geo[((size_t)b * a.F + f) * 3 + 1];
                const float xmin = fminf(fminf(p0.x, p0.z), p1.x), ymin = fminf(fminf(p0.y, p0.w), p1.y);
                const float xmax = fmaxf(fmaxf(p0.x, p0.z), p1.x), ymax = fmaxf(fmaxf(p0.y, p0.w), p1.y);
                if (x0 < xmin - a.infl || x0 > xmax + a.infl || y0 < ymin - a.infl || y0 > ymax + a.infl) continue;
                ++cnt;
                if (f >= f_lo && f < f_hi) k4_accumulate(a, acc + (size_t)(f - f_lo) * 9, x0, y0, p0, p1, ga, sq);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < nf * 9; i += MM_FA_THREADS) {
        const int f = i / 9, c = i - f * 9;
        const size_t o = (size_t)b * a.F + f_lo + f;
        if (c < 6) a.dfxy[o * 6 + c] = acc[i]; else a.dfn[o * 3 + (c - 6)] = acc[i];
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
    a.tcnt = w.tcnt; a.trec = w.trec; a.tspill = w.tspill; a.ntiles_ = w.ntiles;
    a.sb_cnt = w.sb_cnt; a.sb_mask = w.sb_mask; a.sb_ids = w.sb_ids; a.sb_over = w.sb_over; a.tiles8_x = w.tiles8_x; a.tiles8 = w.tiles8;
    a.uvt_offsets = d->uvt_offsets; a.uvt_faces = d->uvt_faces;
    a.ntx = (d->Wt + MM_TS - 1) / MM_TS; a.nty = (d->Ht + MM_TS - 1) / MM_TS;
    a.grad_textures = g->grad_textures; a.dfxy = w.dfxy; a.dfn = w.dfn;
    if (hipMemsetAsync(w.tcnt, 0, ((size_t)d->B * w.ntiles + d->B) * sizeof(int), s) != hipSuccess) return MM_ERR_LAUNCH;
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
        // faces per workgroup: as many as fit in LDS beside the 16 per-wave K4 stages (160 KiB per CU, keep two resident)
        const int Fr = d->F < 1536 ? d->F : 1536;
        const int R = (d->F + Fr - 1) / Fr;
        const size_t lds = MM_FA_WAVES * sizeof(K4Stage) + (size_t)Fr * 9 * sizeof(float);
        if (lds > 64 * 1024 &&
            hipFuncSetAttribute((const void*)face_accum_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return MM_ERR_LAUNCH;
        hipLaunchKernelGGL(face_accum_kernel, dim3(d->B * R), dim3(MM_FA_THREADS), lds, s, a, Fr, R);
    }
    return hipGetLastError() == hipSuccess ? MM_OK : MM_ERR_LAUNCH;
}

}  // namespace mm
