// mm_dibr.hip -- kaolin.render.mesh.dibr_rasterization as its own operator for gfx950 (the op boundary the
// reference imports, /root/reference/networks.py:10, call site :297-299).  Upstream (NVIDIAGameWorks/kaolin v0.12.0, not vendored)
// this is rasterize() [kaolin._C packed_rasterize_forward_cuda + rasterize_backward_cuda, K1/K2] followed by dibr_soft_mask()
// [dibr_soft_mask_forward_cuda + _backward_cuda, K3/K4] with arbitrary per-corner feature channels.
//
// Forward, three launches:
//   dibr_pack     face_vertices_image / _z / normals_z  ->  the packed face records + screen-bin candidate masks the walk consumes
//                 (the same records the fused vertex stage writes: x multiplier, inflated pixel box, one wave = one mask word)
//   order         heavy tiles first (shared with the fused path)
//   raster_dibr   THE SAME candidate walk as the fused render kernel (mm_raster_walk.h: tile_walk) -- face_idx, barycentrics and
//                 the soft-mask state are bit-identical between the two boundaries -- then the generic epilogue: interpolate D
//                 feature channels, soft mask, int64 face_idx.
// Backward, one launch: face-major GATHER (no float atomics on HBM).  16 lanes per (image, face) sweep the face's inflated pixel
//   box: pixels the face owns give K2 (d/d features and, through the barycentrics, d/d face_vertices_image), uncovered pixels
//   that hold the face among their first knum soft-mask faces give K4; per-face sums live in LDS and are stored once.
#include "mm_raster_walk.h"

namespace mm {

struct DibrWorkspace {
    float4* geo; uint64_t* binmask; unsigned short* order; int* nheavy; int* bincount; float2* soft; int32_t* fidx;
    int bin_shift, nbx, nby, words, blocks_per_image;
    size_t bytes;
};

static DibrWorkspace carve_dibr(void* base, int B, int F, int H, int W) {
    DibrWorkspace w;
    char* p = (char*)base;
    size_t o = 0;
    w.bin_shift = bin_shift_for(H, W, F);
    w.nbx = (W + (1 << w.bin_shift) - 1) >> w.bin_shift;
    w.nby = (H + (1 << w.bin_shift) - 1) >> w.bin_shift;
    w.words = (F + 63) / 64;
    w.blocks_per_image = ((W + MM_BLOCK_PX - 1) / MM_BLOCK_PX) * ((H + MM_BLOCK_PX - 1) / MM_BLOCK_PX);
    w.geo = (float4*)(p + o);            o += align256((size_t)B * F * 3 * sizeof(float4));
    w.binmask = (uint64_t*)(p + o);      o += align256((size_t)B * w.nbx * w.nby * w.words * sizeof(uint64_t));
    w.order = (unsigned short*)(p + o);  o += align256((size_t)B * 4 * w.blocks_per_image * sizeof(unsigned short));
    w.nheavy = (int*)(p + o);            o += align256((size_t)B * 4 * sizeof(int));
    w.bincount = (int*)(p + o);          o += align256((size_t)B * w.nbx * w.nby * sizeof(int));
    w.soft = (float2*)(p + o);           o += align256((size_t)B * H * W * sizeof(float2));
    w.fidx = (int32_t*)(p + o);          o += align256((size_t)B * H * W * sizeof(int32_t));
    w.bytes = o;
    return w;
}

struct PackArgs {
    int B, F, H, W;
    float mult, infl;
    const float* fz; const float* fvi; const float* fnz;
    float4* geo;
    int bin_shift, nbx, nby, words;
    uint64_t* mask;
};

__global__ __launch_bounds__(256) void dibr_pack_kernel(PackArgs a) {
    const int b = blockIdx.y, tid = threadIdx.x, f = blockIdx.x * 256 + tid;
    int bx0 = 0, by0 = 0, bw = 0, bh = 0;
    if (f < a.F) {
        const size_t o = (size_t)b * a.F + f;
        const float* q = a.fvi + o * 6;
        const float ax = q[0] * a.mult, ay = q[1] * a.mult, bx = q[2] * a.mult, by = q[3] * a.mult, cx = q[4] * a.mult, cy = q[5] * a.mult;
        const float* z = a.fz + o * 3;
        unsigned org, ext;
        face_pixel_box(ax, ay, bx, by, cx, cy, a.infl, a.mult, a.W, a.H, bx0, by0, bw, bh, org, ext);
        a.geo[o * 3 + 0] = make_float4(ax, ay, bx, by);
        a.geo[o * 3 + 1] = make_float4(cx, cy, z[0], z[1]);
        a.geo[o * 3 + 2] = make_float4(z[2], a.fnz[o], __uint_as_float(org), __uint_as_float(ext));
    }
    const int c = blockIdx.x * 4 + (tid >> 6);
    if (c >= a.words) return;
    bin_wave_faces(a.mask, b, a.nbx, a.nby, a.words, a.bin_shift, c, tid & 63, bx0, by0, bw, bh);
}

// the fused kernel's walk (four tiles per workgroup, or one heavy tile walked by its four waves), the generic epilogue
template <bool kBlock, bool kQueue>
__global__ __launch_bounds__(kBlock ? 256 : 64) void raster_dibr_kernel(RasterArgs a) {
    __shared__ WaveStage s_stage[kBlock ? 4 : 1];
    const int wv = kBlock ? threadIdx.x >> 6 : 0;
    bool valid, coop;
    const TileCtx t = make_tile<kBlock>(a, wv, -1, valid, coop, 4 * a.blocks_per_image);
    unsigned long long key;
    Hit h;
    SoftState ss;
    MM_PP_BEGIN();
    if (kBlock && coop) {
        tile_walk_coop(a, t, s_stage, wv, key, ss MM_PP_PASS);
        if (wv != 0) return;
    } else {
        if (!valid) return;
        if (kQueue) tile_walk(a, t, &s_stage[wv], key, ss MM_PP_PASS);
        else tile_walk_batch(a, t, &s_stage[wv], key, ss MM_PP_PASS);
    }
    winner(a, t, key, h);
    if (!t.in_img) return;
    const size_t pix = ((size_t)t.b * a.H + t.py) * a.W + t.px;
    a.face_idx[pix] = h.f;
    a.face_idx64[pix] = (long long)h.f;
    const float keepprod = ss.zeros > 0 ? 0.f : ss.qnz;
    a.soft_out[pix] = (h.f >= 0) ? 1.f : (1.f - keepprod);
    a.soft[pix] = make_float2((h.f >= 0 || ss.zeros >= 2) ? 0.f : (ss.zeros == 1 ? -ss.qnz : ss.qnz), __int_as_float(ss.lastf));
    float* out = a.interp + pix * a.D;
    if (h.f >= 0) {
        const float* ff = a.feats + ((size_t)t.b * a.F + h.f) * 3 * a.D;
        for (int d = 0; d < a.D; ++d) out[d] = (h.w0 * ff[d] + h.w1 * ff[a.D + d]) + h.w2 * ff[2 * a.D + d];
    } else {
        for (int d = 0; d < a.D; ++d) out[d] = 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------------
struct DibrBwdArgs {
    int B, H, W, F, D, options;
    float mult, eps, sigmainv, infl, kx, ky;
    const float4* geo; const int32_t* fidx; const float2* soft; const float* feats;
    const float* g_interp; const float* g_soft;
    float* dfvi; float* dfeat;
};

#define MM_DB_FL 16                       // lanes per face
#define MM_DB_FPW (64 / MM_DB_FL)         // faces per wave
#define MM_DB_ACC (6 + 3 * MM_DIBR_MAX_D)

__device__ inline float seg_nearest_t(float px, float py, float ux, float uy, float vx, float vy, float& qx, float& qy, float& d2) {
    const float ex = vx - ux, ey = vy - uy, rx = px - ux, ry = py - uy;
    const float len2 = ex * ex + ey * ey;
    float t = (len2 > 0.f) ? (rx * ex + ry * ey) / len2 : 0.f;
    t = fminf(fmaxf(t, 0.f), 1.f);                               // clamped projection = kaolin's three regions in one form
    qx = rx - t * ex; qy = ry - t * ey;
    d2 = qx * qx + qy * qy;
    return t;
}

// The sweep: MM_DB_FL lanes per face walk the face's inflated pixel box; lane sl takes box pixels sl, sl + 16, ... -- a FIXED assignment.
// kRegs (up to MM_DB_REG_D feature channels: the reference has 6): every lane adds its own pixels' contributions up in registers, in its own
// fixed order, and the sixteen partial sums of a face are added by a fixed butterfly: no atomics at all, the gradients are bitwise
// reproducible, and the LDS float atomics this kernel used to issue (~81 ns of the CU's LDS unit per wave-instruction, whatever the
// addresses: profiles/r02_lds_atomic_calibration.txt) are gone.  More channels than that: per-face LDS accumulators with float atomics
// (order-dependent in the last bits), as before.
#define MM_DB_REG_D 8
template <bool kRegs>
__global__ __launch_bounds__(256) void dibr_bwd_kernel(DibrBwdArgs a) {
    __shared__ float s_acc[kRegs ? 1 : 4][kRegs ? 1 : MM_DB_FPW][kRegs ? 1 : MM_DB_ACC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = lane / MM_DB_FL, sl = lane % MM_DB_FL;
    const long long wid = (long long)blockIdx.x * 4 + wave;      // (face quartet, image): the waves of a workgroup sweep one image's neighbours
    const int b = (int)(wid % a.B);
    const int f_raw = (int)(wid / a.B) * MM_DB_FPW + grp;
    const bool live = f_raw < a.F;
    const int f = live ? f_raw : 0;
    float* acc = kRegs ? nullptr : s_acc[wave][grp];
    float racc[kRegs ? 6 + 3 * MM_DB_REG_D : 1];
#pragma unroll
    for (int k = 0; k < (kRegs ? 6 + 3 * MM_DB_REG_D : 1); ++k) racc[k] = 0.f;
    const int nacc = 6 + 3 * a.D;
    if (!kRegs) for (int k = sl; k < nacc; k += MM_DB_FL) acc[k] = 0.f;
    const size_t o = (size_t)b * a.F + f, hw = (size_t)a.H * a.W;
    const float4 p0 = a.geo[o * 3 + 0], p1 = a.geo[o * 3 + 1], g2 = a.geo[o * 3 + 2];
    const unsigned org = __float_as_uint(g2.z), ext = __float_as_uint(g2.w);
    const int px0 = (int)(org & 0xFFFFu), py0 = (int)(org >> 16), bw = (int)(ext & 0xFFFFu);
    const bool front = (a.options & MM_OPT_CULL_STRICT) ? g2.y > 0.f : g2.y >= 0.f;     // g2.y = face_normals_z
    const int npx = live && (front || !(a.options & MM_OPT_SOFT_SKIP_CULLED)) ? bw * (int)(ext >> 16) : 0;
    const float xmin = fminf(fminf(p0.x, p0.z), p1.x), ymin = fminf(fminf(p0.y, p0.w), p1.y);
    const float xmax = fmaxf(fmaxf(p0.x, p0.z), p1.x), ymax = fmaxf(fmaxf(p0.y, p0.w), p1.y);
    const float s2 = a.mult * a.mult;
    int nmax = npx;
    static_assert(MM_DB_FL == 16, "the exchange strides below start at the lanes-per-face count");
    nmax = max(nmax, (int)lane_xchg<16>((unsigned)nmax, threadIdx.x & 63)); nmax = max(nmax, (int)lane_xchg<32>((unsigned)nmax, threadIdx.x & 63));
    wave_lds_sync();
    auto add = [&](int k, float v) { if (kRegs) racc[k] += v; else atomicAdd(&acc[k], v); };   // (k is a compile-time constant at every call with kRegs)
    for (int base = 0; base < nmax; base += MM_DB_FL) {
        const int idx = base + sl;
        if (idx >= npx) continue;
        const int yy = idx / bw;
        const int px = px0 + (idx - yy * bw), py = py0 + yy;
        const size_t pix = (size_t)b * hw + (size_t)py * a.W + px;
        const int fi = a.fidx[pix];
        const float x0 = pixel_x_k(px, a.W, a.kx), y0 = pixel_y_k(py, a.H, a.ky);
        if (fi == f && a.g_interp) {
            // K2 (Appendix A.1)
            float w0, w1, w2, nrm;
            bary_weights(p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, x0, y0, a.eps, (a.options & MM_OPT_BARY_ONE_MINUS) != 0, w0, w1, w2, nrm);
            const float* g = a.g_interp + pix * a.D;
            const float* ff = a.feats + o * 3 * a.D;
            float G0 = 0.f, G1 = 0.f, G2 = 0.f;
            if (kRegs) {
#pragma unroll
                for (int d = 0; d < MM_DB_REG_D; ++d) {
                    if (d < a.D) {
                        const float gd = g[d];
                        G0 += gd * ff[d]; G1 += gd * ff[a.D + d]; G2 += gd * ff[2 * a.D + d];
                        if (a.dfeat) { racc[6 + d] += w0 * gd; racc[6 + MM_DB_REG_D + d] += w1 * gd; racc[6 + 2 * MM_DB_REG_D + d] += w2 * gd; }
                    }
                }
            } else {
                for (int d = 0; d < a.D; ++d) {
                    const float gd = g[d];
                    G0 += gd * ff[d]; G1 += gd * ff[a.D + d]; G2 += gd * ff[2 * a.D + d];
                    if (a.dfeat && gd != 0.f) {
                        atomicAdd(&acc[6 + d], w0 * gd); atomicAdd(&acc[6 + a.D + d], w1 * gd); atomicAdd(&acc[6 + 2 * a.D + d], w2 * gd);
                    }
                }
            }
            const float Gm = (w0 * G0 + w1 * G1) + w2 * G2;
            const float dw0 = (G0 - Gm) / nrm, dw1 = (G1 - Gm) / nrm, dw2 = (G2 - Gm) / nrm;
            const float aex = p0.x - x0, aey = p0.y - y0, bex = p0.z - x0, bey = p0.w - y0, cex = p1.x - x0, cey = p1.y - y0;
            add(0, (dw1 * (-cey) + dw2 * bey) * a.mult); add(1, (dw1 * cex + dw2 * (-bex)) * a.mult);
            add(2, (dw0 * cey + dw2 * (-aey)) * a.mult); add(3, (dw0 * (-cex) + dw2 * aex) * a.mult);
            add(4, (dw0 * (-bey) + dw1 * aey) * a.mult); add(5, (dw0 * bex + dw1 * (-aex)) * a.mult);
        } else if (fi == -1 && a.g_soft) {
            // K4 (Appendix A.2): this face is among the pixel's first knum soft-mask faces iff its inflated box holds the pixel
            // and its index does not exceed the knum-th face the forward took
            const float ga = a.g_soft[pix];
            const float2 st = a.soft[pix];
            const float sq = st.x;
            const int lf = __float_as_int(st.y);
            const int bm = box_mode(a.options);
            const bool inbox = !(box_reject(x0, xmin - a.infl, xmax + a.infl, bm) || box_reject(y0, ymin - a.infl, ymax + a.infl, bm));
            if (sq != 0.f && (MM_K4_KEEP_ONES || sq != 1.f) && ga != 0.f && f <= lf && inbox) {
                float qx, qy, d2, qx1, qy1, d21;
                float t = seg_nearest_t(x0, y0, p0.x, p0.y, p0.z, p0.w, qx, qy, d2);        // edge 0: a -> b
                int e = 0;
                float t1 = seg_nearest_t(x0, y0, p0.z, p0.w, p1.x, p1.y, qx1, qy1, d21);    // edge 1: b -> c
                if (d21 < d2) { d2 = d21; qx = qx1; qy = qy1; t = t1; e = 1; }
                t1 = seg_nearest_t(x0, y0, p1.x, p1.y, p0.x, p0.y, qx1, qy1, d21);          // edge 2: c -> a
                if (d21 < d2) { d2 = d21; qx = qx1; qy = qy1; t = t1; e = 2; }
                const float p = expf(-((d2 / s2) * a.sigmainv));
                const float q = soft_factor(x0, y0, p0, p1, a.sigmainv / s2);    // the factor exactly as the walk folded it into the product
                const float qnz = fabsf(sq);
                const bool onezero = sq < 0.f;
                const float excl = (q != 0.f) ? (onezero ? 0.f : qnz / q) : (onezero ? qnz : 0.f);
                const float gd = ga * excl * (-(p * a.sigmainv) / s2) * a.mult;
                if (gd != 0.f) {
                    // edge e runs from corner e to corner (e + 1) % 3
                    const float cu = -2.f * (1.f - t) * gd, cv = -2.f * t * gd;
                    if (kRegs) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) {             // (corner c receives cu if it starts the edge, cv if it ends it: selects, no indexed registers)
                            const float wgt = (c == e ? cu : 0.f) + (c == (e == 2 ? 0 : e + 1) ? cv : 0.f);
                            racc[2 * c] += wgt * qx; racc[2 * c + 1] += wgt * qy;
                        }
                    } else {
                        const int iu = e * 2, iv = (e == 2 ? 0 : e + 1) * 2;
                        atomicAdd(&acc[iu], cu * qx); atomicAdd(&acc[iu + 1], cu * qy);
                        atomicAdd(&acc[iv], cv * qx); atomicAdd(&acc[iv + 1], cv * qy);
                    }
                }
            }
        }
    }
    if (kRegs) {
        // the face's sixteen lanes: a fixed butterfly (every lane ends with the total)
#pragma unroll
        for (int k = 0; k < 6 + 3 * MM_DB_REG_D; ++k) {
            float v = racc[k];
            v += xchg_f32<8>(v, lane); v += xchg_f32<4>(v, lane); v += xchg_f32<2>(v, lane); v += xchg_f32<1>(v, lane);
            racc[k] = v;
        }
        if (live && sl == 0) {
#pragma unroll
            for (int k = 0; k < 6; ++k) a.dfvi[o * 6 + k] = racc[k];
            if (a.dfeat) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int d = 0; d < MM_DB_REG_D; ++d)
                        if (d < a.D) a.dfeat[o * 3 * a.D + c * a.D + d] = racc[6 + c * MM_DB_REG_D + d];
            }
        }
        return;
    }
    wave_lds_sync();
    if (live) {
        for (int k = sl; k < 6; k += MM_DB_FL) a.dfvi[o * 6 + k] = acc[k];
        if (a.dfeat) for (int k = sl; k < 3 * a.D; k += MM_DB_FL) a.dfeat[o * 3 * a.D + k] = acc[6 + k];
    }
}

static int check_dibr(const MMDibrDesc* d) {
    if (!d) return MM_ERR_NULL_POINTER;
    if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->F <= 0 || d->D <= 0) return MM_ERR_BAD_SHAPE;
    if (d->knum <= 0 || d->H > 65535 || d->W > 65535 || d->D > MM_DIBR_MAX_D) return MM_ERR_UNSUPPORTED;
    if (!d->face_vertices_z || !d->face_vertices_image || !d->face_features || !d->face_normals_z) return MM_ERR_NULL_POINTER;
    if (!d->workspace || ((uintptr_t)d->workspace & 255) || d->workspace_bytes < carve_dibr(nullptr, d->B, d->F, d->H, d->W).bytes) return MM_ERR_WORKSPACE;
    return MM_OK;
}

}  // namespace mm

extern "C" {

size_t mm_dibr_query_workspace(const MMDibrDesc* d) {
    if (!d || d->B <= 0 || d->F <= 0 || d->H <= 0 || d->W <= 0) return 0;
    return mm::carve_dibr(nullptr, d->B, d->F, d->H, d->W).bytes;
}

int mm_dibr_rasterization_forward(const MMDibrDesc* d, mm_stream_t stream) {
    using namespace mm;
    int st = check_dibr(d);
    if (st != MM_OK) return st;
    if (!d->interpolated_features || !d->soft_mask || !d->face_idx) return MM_ERR_NULL_POINTER;
    hipStream_t s = (hipStream_t)stream;
    const DibrWorkspace w = carve_dibr(d->workspace, d->B, d->F, d->H, d->W);
    clear_stale_error();
    PackArgs p;
    p.B = d->B; p.F = d->F; p.H = d->H; p.W = d->W; p.mult = d->multiplier; p.infl = d->boxlen * d->multiplier;
    p.fz = d->face_vertices_z; p.fvi = d->face_vertices_image; p.fnz = d->face_normals_z; p.geo = w.geo;
    p.bin_shift = w.bin_shift; p.nbx = w.nbx; p.nby = w.nby; p.words = w.words; p.mask = w.binmask;
    hipLaunchKernelGGL(dibr_pack_kernel, dim3((d->F + 255) / 256, d->B), dim3(256), 0, s, p);
    if (launch_ok("dibr_pack") != MM_OK) return MM_ERR_LAUNCH;
    RasterArgs a = {};
    a.B = d->B; a.H = d->H; a.W = d->W; a.F = d->F; a.knum = d->knum;
    a.blocks_x = (d->W + MM_BLOCK_PX - 1) / MM_BLOCK_PX; a.blocks_per_image = w.blocks_per_image;
    a.bin_shift = w.bin_shift; a.nbx = w.nbx; a.nby = w.nby; a.words = w.words;
    a.mult = d->multiplier; a.eps = d->eps; a.sigmainv = d->sigmainv; a.infl = d->boxlen * d->multiplier;
    a.kx = d->multiplier / (float)d->W; a.ky = d->multiplier / (float)d->H;
    a.geo = w.geo; a.binmask = w.binmask; a.soft = w.soft; a.face_idx = w.fidx; a.options = d->options;
    a.feats = d->face_features; a.D = d->D; a.interp = d->interpolated_features; a.soft_out = d->soft_mask;
    a.face_idx64 = (long long*)d->face_idx;
    a.order = launch_order(a, w.order, w.nheavy, w.bincount, d->B, nullptr, s);
    a.spread = walk_spread(a);
    a.nheavy = w.nheavy;
    const bool block = walk_block_mode(a);
    const bool queue = walk_queue_mode(a);
    if (block) { if (queue) hipLaunchKernelGGL((raster_dibr_kernel<true, true>), dim3(walk_grid(a, true)), dim3(256), 0, s, a);
                 else hipLaunchKernelGGL((raster_dibr_kernel<true, false>), dim3(walk_grid(a, true)), dim3(256), 0, s, a); }
    else { if (queue) hipLaunchKernelGGL((raster_dibr_kernel<false, true>), dim3(walk_grid(a, false)), dim3(64), 0, s, a);
           else hipLaunchKernelGGL((raster_dibr_kernel<false, false>), dim3(walk_grid(a, false)), dim3(64), 0, s, a); }
    return launch_ok("raster_dibr");
}

int mm_dibr_rasterization_backward(const MMDibrDesc* d, const MMDibrGrads* g, mm_stream_t stream) {
    using namespace mm;
    int st = check_dibr(d);
    if (st != MM_OK) return st;
    if (!g || !g->grad_face_vertices_image) return MM_ERR_NULL_POINTER;
    const DibrWorkspace w = carve_dibr(d->workspace, d->B, d->F, d->H, d->W);
    DibrBwdArgs a;
    a.B = d->B; a.H = d->H; a.W = d->W; a.F = d->F; a.D = d->D; a.options = d->options;
    a.mult = d->multiplier; a.eps = d->eps; a.sigmainv = d->sigmainv; a.infl = d->boxlen * d->multiplier;
    a.kx = d->multiplier / (float)d->W; a.ky = d->multiplier / (float)d->H;
    a.geo = w.geo; a.fidx = w.fidx; a.soft = w.soft; a.feats = d->face_features;
    a.g_interp = g->grad_interpolated_features; a.g_soft = g->grad_soft_mask;
    a.dfvi = g->grad_face_vertices_image; a.dfeat = g->grad_face_features;
    clear_stale_error();
    const long long nwaves = (long long)d->B * ((d->F + MM_DB_FPW - 1) / MM_DB_FPW);
    if (d->D <= MM_DB_REG_D) hipLaunchKernelGGL(dibr_bwd_kernel<true>, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);   // deterministic
    else hipLaunchKernelGGL(dibr_bwd_kernel<false>, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
    return launch_ok("dibr_bwd");
}

}  // extern "C"
