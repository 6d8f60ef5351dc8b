// mm_raster.hip -- pixel stage of the render path for gfx950 (forward and backward).
//
// Replaces, fused into one launch per direction, what the reference reaches through kaolin for every pixel
// (call sites /root/reference/networks.py:297-317; semantics SURVEY.md 8(a) rows a8-a11, gradients Appendix A):
//   packed_rasterize_forward  (K1)  nearest front-facing face per pixel, barycentric interpolation
//   dibr_soft_mask_forward    (K3)  1 - prod(1 - exp(-sigma d^2)) over the first <= knum nearby faces
//   texture_mapping / grid_sample, spherical_harmonic_lighting, composite, clamp, cat
// and their backward kernels (K2, K4, grid_sampler backward, SH backward).
//
// Design (not kaolin's pixel-major brute force):  one wave owns an 8x8 pixel tile, one lane per pixel.  The wave
// streams the image's face bounding boxes 64 at a time (one coalesced float4 per lane), tests box-vs-tile,
// compacts the survivors IN FACE ORDER with ballot + popcount-prefix into its LDS slot array, and only then do the
// lanes run the per-pixel edge functions against the short list.  Face order is preserved end to end, which is
// what kaolin's tie rule (lowest index wins) and the soft mask's "first knum faces" rule need.  A 256-thread
// workgroup is four such waves side by side (a 32x8 strip: 128-byte output rows).
#include "mm_device.h"

namespace mm {

struct RasterArgs {
    int B, H, W, F, Ht, Wt, knum, tiles_x, tiles_per_image;
    float mult, eps, sigmainv, infl;        // infl = boxlen * multiplier
    const float4* bbox;
    const float4* geo;
    const uint64_t* valid;
    const float* face_uvs;
    const float* fn;                        // (B,F,3) unit normals
    const float* textures;
    const float* lights;
    const float* bg;
    // forward outputs
    float* rgba;
    int32_t* face_idx;
    float* imnormal;
    // backward
    const float* grad_rgba;
    float* grad_textures;
    float* grad_lights;
    float* grad_bg;
    float* dfxy;
    float* dfn;
};

// One staged face: everything a lane needs for the hard test and the soft distance (64 bytes, read as broadcasts).
struct __attribute__((aligned(16))) Slot { float4 bb, p0, p1, p2; };   // p2 = {cz, nz, fidx(bits), 0}

struct TileCtx {
    int b, px, py, lane, wave;
    bool in_img;
    float x0, y0;
    float txlo, txhi, tylo, tyhi;           // pixel-centre extent of this wave's tile (multiplier units)
};

__device__ inline TileCtx make_tile(const RasterArgs& a) {
    TileCtx t;
    int tile;
    map_block(blockIdx.x, a.B, a.tiles_per_image, t.b, tile);
    t.lane = threadIdx.x & 63; t.wave = threadIdx.x >> 6;
    const int bx = tile % a.tiles_x, by = tile / a.tiles_x;
    const int tx0 = bx * (MM_TILE_W * MM_BLOCK_WAVES) + t.wave * MM_TILE_W, ty0 = by * MM_TILE_H;
    t.px = tx0 + (t.lane & 7); t.py = ty0 + (t.lane >> 3);
    t.in_img = t.px < a.W && t.py < a.H;
    t.x0 = pixel_x(t.px, a.W, a.mult); t.y0 = pixel_y(t.py, a.H, a.mult);
    // the same monotone formula bounds every pixel centre of the tile, so box-vs-tile rejection is exactly conservative
    t.txlo = pixel_x(tx0, a.W, a.mult); t.txhi = pixel_x(tx0 + MM_TILE_W - 1, a.W, a.mult);
    t.tyhi = pixel_y(ty0, a.H, a.mult); t.tylo = pixel_y(ty0 + MM_TILE_H - 1, a.H, a.mult);
    return t;
}

// Stream the face boxes of image b through the wave.  Survivors of chunk [base, base+64) are written, in face order,
// to slots[0..n) and `body(n)` is invoked (wave-uniform).  kSoft selects the inflated, un-culled candidate set.
template <bool kSoft, class Body>
__device__ inline void scan_faces(const RasterArgs& a, const TileCtx& t, Slot* slots, Body&& body) {
    const float4* bbox = a.bbox + (size_t)t.b * a.F;
    const float4* geo = a.geo + (size_t)t.b * a.F * 3;
    const uint64_t* valid = a.valid + (size_t)t.b * ((a.F + 63) / 64);
    const float pad = kSoft ? a.infl : 0.f;
    for (int base = 0; base < a.F; base += 64) {
        const int f = base + t.lane;
        bool hit = false;
        float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < a.F) {
            bb = bbox[f];
            hit = !((bb.z + pad) < t.txlo || (bb.x - pad) > t.txhi || (bb.w + pad) < t.tylo || (bb.y - pad) > t.tyhi);
        }
        uint64_t m = __ballot(hit);
        if (!kSoft) m &= valid[base >> 6];                       // back-face cull applies to colour only (a8)
        if (m == 0) continue;
        const bool keep = (m >> t.lane) & 1ull;
        if (keep) {
            const int pos = __popcll(m & ((1ull << t.lane) - 1ull));
            const float4 g0 = geo[(size_t)f * 3 + 0], g1 = geo[(size_t)f * 3 + 1], g2 = geo[(size_t)f * 3 + 2];
            Slot s;
            s.bb = bb; s.p0 = g0; s.p1 = g1; s.p2 = make_float4(g2.x, g2.y, __int_as_float(f), 0.f);
            slots[pos] = s;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (!body(__popcll(m))) return;
        __builtin_amdgcn_wave_barrier();
    }
}

struct Hit { float best; int f; float w0, w1, w2; };

// K1 per pixel: faces arrive in index order; strict z > best keeps the lowest index on ties.
__device__ inline void raster_pixels(const RasterArgs& a, const TileCtx& t, Slot* slots, Hit& h) {
    scan_faces<false>(a, t, slots, [&](int n) {
        for (int j = 0; j < n; ++j) {
            const float4 bb = slots[j].bb;
            if (t.x0 < bb.x || t.x0 > bb.z || t.y0 < bb.y || t.y0 > bb.w) continue;
            const float4 p0 = slots[j].p0, p1 = slots[j].p1, p2 = slots[j].p2;
            float w0, w1, w2, nrm;
            edge_weights(p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, t.x0, t.y0, a.eps, w0, w1, w2, nrm);
            // cheap exact pre-reject: w/nrm < 0 whenever w and nrm have opposite signs and the quotient cannot
            // underflow to -0; everything else takes the IEEE divisions the oracle takes.
            const float sg = nrm < 0.f ? -1.f : 1.f;
            if (fabsf(nrm) < 1e10f && fminf(fminf(w0 * sg, w1 * sg), w2 * sg) < -1e-30f) continue;
            w0 /= nrm; w1 /= nrm; w2 /= nrm;
            if (w0 < 0.f || w1 < 0.f || w2 < 0.f) continue;
            const float z0 = (w0 * p1.z + w1 * p1.w) + w2 * p2.x;
            if (!(z0 > h.best)) continue;
            h.best = z0; h.f = __float_as_int(p2.z); h.w0 = w0; h.w1 = w1; h.w2 = w2;
        }
        return true;
    });
}

// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
template <bool kNoMask>
__global__ __launch_bounds__(256) void raster_fwd_kernel(RasterArgs a) {
    __shared__ Slot s_slots[MM_BLOCK_WAVES][64];
    const TileCtx t = make_tile(a);
    Slot* slots = s_slots[t.wave];

    Hit h; h.best = -INFINITY; h.f = -1; h.w0 = h.w1 = h.w2 = 0.f;
    raster_pixels(a, t, slots, h);

    // K3: soft silhouette for the lanes no front face covers
    float keepprod = 1.f;
    const bool open = t.in_img && h.f < 0;
    if (__ballot(open)) {
        int cnt = 0;
        const float s2 = a.mult * a.mult;
        scan_faces<true>(a, t, slots, [&](int n) {
            for (int j = 0; j < n; ++j) {
                const float4 bb = slots[j].bb;
                const bool in = open && cnt < a.knum &&
                                !(t.x0 < bb.x - a.infl || t.x0 > bb.z + a.infl || t.y0 < bb.y - a.infl || t.y0 > bb.w + a.infl);
                if (!in) continue;
                const float4 p0 = slots[j].p0, p1 = slots[j].p1;
                int r;
                float d = seg_dist2(t.x0, t.y0, p0.x, p0.y, p0.z, p0.w, r);
                const float d1 = seg_dist2(t.x0, t.y0, p0.z, p0.w, p1.x, p1.y, r); if (d1 < d) d = d1;
                const float d2 = seg_dist2(t.x0, t.y0, p1.x, p1.y, p0.x, p0.y, r); if (d2 < d) d = d2;
                const float p = expf(-((d / s2) * a.sigmainv));
                keepprod = keepprod * (1.f - p);
                ++cnt;
            }
            return __ballot(open && cnt < a.knum) != 0;          // every open lane already holds knum faces: stop
        });
    }
    if (!t.in_img) return;

    // ---- shading (a9-a11).  Uncovered pixels carry zero features exactly like kaolin's interpolated_features.
    const size_t pix = ((size_t)t.b * a.H + t.py) * a.W + t.px;
    float m = 0.f, u = 0.f, v = 0.f, nx = 0.f, ny = 0.f, nz = 0.f;
    if (h.f >= 0) {
        const float* fu = a.face_uvs + (size_t)h.f * 6;
        const float* nn = a.fn + ((size_t)t.b * a.F + h.f) * 3;
        m = (h.w0 + h.w1) + h.w2;
        u = (h.w0 * fu[0] + h.w1 * fu[2]) + h.w2 * fu[4];
        v = (h.w0 * fu[1] + h.w1 * fu[3]) + h.w2 * fu[5];
        const float n0 = nn[0], n1 = nn[1], n2 = nn[2];
        nx = (h.w0 * n0 + h.w1 * n0) + h.w2 * n0;
        ny = (h.w0 * n1 + h.w1 * n1) + h.w2 * n1;
        nz = (h.w0 * n2 + h.w1 * n2) + h.w2 * n2;
    }
    const Bilin s = bilin_setup(u, v, a.Ht, a.Wt);
    const bool inw = s.x0 < a.Wt && s.y0 < a.Ht, ine = s.x1 < a.Wt && s.y0 < a.Ht;
    const bool isw = s.x0 < a.Wt && s.y1 < a.Ht, ise = s.x1 < a.Wt && s.y1 < a.Ht;
    float bnd[9];
    sh_bands(nx, ny, nz, bnd);
    const float* L = a.lights + t.b * 9;
    float coef = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) coef += bnd[i] * L[i];
    float out[4];
    const size_t hw = (size_t)a.H * a.W, pin = (size_t)t.py * a.W + t.px;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* tex = a.textures + ((size_t)t.b * 3 + c) * a.Ht * a.Wt;
        float tc = 0.f;
        if (inw) tc += tex[(size_t)s.y0 * a.Wt + s.x0] * s.wnw;
        if (ine) tc += tex[(size_t)s.y0 * a.Wt + s.x1] * s.wne;
        if (isw) tc += tex[(size_t)s.y1 * a.Wt + s.x0] * s.wsw;
        if (ise) tc += tex[(size_t)s.y1 * a.Wt + s.x1] * s.wse;
        float val;
        if (kNoMask) {
            const float g = a.bg[((size_t)t.b * 3 + c) * hw + pin];
            val = (tc * m + g * (1.f - m)) * coef;
        } else {
            val = (tc * m) * coef + 1.f * (1.f - m);
        }
        out[c] = val < 0.f ? 0.f : (val > 1.f ? 1.f : val);
    }
    out[3] = (h.f >= 0) ? 1.f : (1.f - keepprod);
    *(float4*)(a.rgba + pix * 4) = make_float4(out[0], out[1], out[2], out[3]);
    a.face_idx[pix] = h.f;
    if (a.imnormal) { a.imnormal[pix * 3] = nx; a.imnormal[pix * 3 + 1] = ny; a.imnormal[pix * 3 + 2] = nz; }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------------
template <bool kNoMask>
__global__ __launch_bounds__(256) void raster_bwd_kernel(RasterArgs a) {
    __shared__ Slot s_slots[MM_BLOCK_WAVES][64];
    __shared__ float s_dl[MM_BLOCK_WAVES][9];
    const TileCtx t = make_tile(a);
    Slot* slots = s_slots[t.wave];
    const size_t hw = (size_t)a.H * a.W, pin = (size_t)t.py * a.W + t.px;
    const size_t pix = (size_t)t.b * hw + pin;

    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
    int hf = -1;
    if (t.in_img) { g4 = *(const float4*)(a.grad_rgba + pix * 4); hf = a.face_idx[pix]; }
    const float gin[3] = {g4.x, g4.y, g4.z};
    float dl[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) dl[i] = 0.f;

    if (t.in_img && (hf >= 0 || kNoMask)) {
        // recompute the forward quantities of this pixel (nothing but face_idx was saved)
        float w0 = 0.f, w1 = 0.f, w2 = 0.f, nrm = 1.f, m = 0.f, u = 0.f, v = 0.f, nx = 0.f, ny = 0.f, nz = 0.f;
        float4 p0 = make_float4(0, 0, 0, 0), p1 = p0;
        float fu[6] = {0, 0, 0, 0, 0, 0}, n0 = 0.f, n1 = 0.f, n2 = 0.f;
        if (hf >= 0) {
            const float4* geo = a.geo + ((size_t)t.b * a.F + hf) * 3;
            p0 = geo[0]; p1 = geo[1];
            edge_weights(p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, t.x0, t.y0, a.eps, w0, w1, w2, nrm);
            w0 /= nrm; w1 /= nrm; w2 /= nrm;
            const float* fuv = a.face_uvs + (size_t)hf * 6;
#pragma unroll
            for (int i = 0; i < 6; ++i) fu[i] = fuv[i];
            const float* nn = a.fn + ((size_t)t.b * a.F + hf) * 3;
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
        const float* L = a.lights + t.b * 9;
        float coef = 0.f;
#pragma unroll
        for (int i = 0; i < 9; ++i) coef += bnd[i] * L[i];

        float dm = 0.f, dc = 0.f, gix = 0.f, giy = 0.f;
        const float ex = 1.f - s.tx, ey = 1.f - s.ty;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const size_t tb = ((size_t)t.b * 3 + c) * a.Ht * a.Wt;
            const float* tex = a.textures + tb;
            const float tnw = inw ? tex[(size_t)s.y0 * a.Wt + s.x0] : 0.f, tne = ine ? tex[(size_t)s.y0 * a.Wt + s.x1] : 0.f;
            const float tsw = isw ? tex[(size_t)s.y1 * a.Wt + s.x0] : 0.f, tse = ise ? tex[(size_t)s.y1 * a.Wt + s.x1] : 0.f;
            float tc = 0.f;
            if (inw) tc += tnw * s.wnw;
            if (ine) tc += tne * s.wne;
            if (isw) tc += tsw * s.wsw;
            if (ise) tc += tse * s.wse;
            float pre, dtc;
            if (kNoMask) {
                const float bgv = a.bg[((size_t)t.b * 3 + c) * hw + pin];
                const float base = tc * m + bgv * (1.f - m);
                pre = base * coef;
                const float g = (pre >= 0.f && pre <= 1.f) ? gin[c] : 0.f;      // torch.clamp backward mask
                dc += g * base;
                const float dbase = g * coef;
                dtc = dbase * m;
                a.grad_bg[((size_t)t.b * 3 + c) * hw + pin] = dbase * (1.f - m);
                dm += dbase * (tc - bgv);
            } else {
                pre = (tc * m) * coef + 1.f * (1.f - m);
                const float g = (pre >= 0.f && pre <= 1.f) ? gin[c] : 0.f;
                dc += g * (tc * m);
                dtc = (g * coef) * m;
                dm += g * (tc * coef - 1.f);
            }
            if (hf >= 0 && dtc != 0.f) {
                float* gt = a.grad_textures + tb;
                if (inw) atomicAdd(gt + (size_t)s.y0 * a.Wt + s.x0, dtc * s.wnw);
                if (ine) atomicAdd(gt + (size_t)s.y0 * a.Wt + s.x1, dtc * s.wne);
                if (isw) atomicAdd(gt + (size_t)s.y1 * a.Wt + s.x0, dtc * s.wsw);
                if (ise) atomicAdd(gt + (size_t)s.y1 * a.Wt + s.x1, dtc * s.wse);
                gix += dtc * ((tne - tnw) * ey + (tse - tsw) * s.ty);
                giy += dtc * ((tsw - tnw) * ex + (tse - tne) * s.tx);
            }
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) dl[i] = dc * bnd[i];
        if (hf >= 0) {
            const float du = gix * s.mx * ((float)a.Wt / 2.f) * 2.f;
            const float dv = giy * s.my * ((float)a.Ht / 2.f) * -2.f;
            const float dnx = dc * (((MM_SH_C1 * L[1] + MM_SH_C4 * ny * L[4]) + MM_SH_C7 * nz * L[7]) + 2.f * MM_SH_C8 * nx * L[8]);
            const float dny = dc * (((MM_SH_C1 * L[3] + MM_SH_C4 * nx * L[4]) + MM_SH_C4 * nz * L[5]) - 2.f * MM_SH_C8 * ny * L[8]);
            const float dnz = dc * (((MM_SH_C1 * L[2] + MM_SH_C4 * ny * L[5]) + 2.f * MM_SH_C6 * nz * L[6]) + MM_SH_C7 * nx * L[7]);
            // K2 (Appendix A.1): features per corner k = (1, u_k, v_k, n)
            const float gn = (dnx * n0 + dny * n1) + dnz * n2;
            const float G0 = ((dm + du * fu[0]) + dv * fu[1]) + gn;
            const float G1 = ((dm + du * fu[2]) + dv * fu[3]) + gn;
            const float G2 = ((dm + du * fu[4]) + dv * fu[5]) + gn;
            const float Gm = (w0 * G0 + w1 * G1) + w2 * G2;
            const float dw0 = (G0 - Gm) / nrm, dw1 = (G1 - Gm) / nrm, dw2 = (G2 - Gm) / nrm;
            const float aex = p0.x - t.x0, aey = p0.y - t.y0, bex = p0.z - t.x0, bey = p0.w - t.y0, cex = p1.x - t.x0, cey = p1.y - t.y0;
            float* dq = a.dfxy + ((size_t)t.b * a.F + hf) * 6;
            atomicAdd(dq + 0, (dw1 * (-cey) + dw2 * bey) * a.mult);
            atomicAdd(dq + 1, (dw1 * cex + dw2 * (-bex)) * a.mult);
            atomicAdd(dq + 2, (dw0 * cey + dw2 * (-aey)) * a.mult);
            atomicAdd(dq + 3, (dw0 * (-cex) + dw2 * aex) * a.mult);
            atomicAdd(dq + 4, (dw0 * (-bey) + dw1 * aey) * a.mult);
            atomicAdd(dq + 5, (dw0 * bex + dw1 * (-aex)) * a.mult);
            float* dn = a.dfn + ((size_t)t.b * a.F + hf) * 3;
            atomicAdd(dn + 0, (w0 * dnx + w1 * dnx) + w2 * dnx);
            atomicAdd(dn + 1, (w0 * dny + w1 * dny) + w2 * dny);
            atomicAdd(dn + 2, (w0 * dnz + w1 * dnz) + w2 * dnz);
        }
    }

    // d lights: wave butterfly -> one LDS row per wave -> 9 atomics per workgroup
#pragma unroll
    for (int i = 0; i < 9; ++i) dl[i] = wave_sum(dl[i]);
    if (t.lane == 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) s_dl[t.wave][i] = dl[i];
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < MM_BLOCK_WAVES; ++w) sum += s_dl[w][threadIdx.x];
        if (sum != 0.f) atomicAdd(a.grad_lights + t.b * 9 + threadIdx.x, sum);
    }

    // K4 (Appendix A.2): soft-mask gradient of the uncovered lanes.  Two passes over the same ordered candidate walk
    // as the forward: pass 1 rebuilds prod(1-p) (split into non-zero factors and a zero count), pass 2 scatters.
    const bool open = t.in_img && hf < 0 && g4.w != 0.f;
    if (__ballot(open) == 0) return;
    const float s2 = a.mult * a.mult;
    float qnz = 1.f;
    int zeros = 0, cnt = 0;
    auto candidate = [&](const Slot& sl, int c, float& p, int& ty) -> bool {
        const float4 bb = sl.bb;
        if (!(open && c < a.knum) || t.x0 < bb.x - a.infl || t.x0 > bb.z + a.infl || t.y0 < bb.y - a.infl || t.y0 > bb.w + a.infl) return false;
        const float4 q0 = sl.p0, q1 = sl.p1;
        int r, reg;
        float d = seg_dist2(t.x0, t.y0, q0.x, q0.y, q0.z, q0.w, reg); ty = reg;
        const float d1 = seg_dist2(t.x0, t.y0, q0.z, q0.w, q1.x, q1.y, r); if (d1 < d) { d = d1; ty = 3 + r; }
        const float d2 = seg_dist2(t.x0, t.y0, q1.x, q1.y, q0.x, q0.y, r); if (d2 < d) { d = d2; ty = 6 + r; }
        p = expf(-((d / s2) * a.sigmainv));
        return true;
    };
    scan_faces<true>(a, t, slots, [&](int n) {
        for (int j = 0; j < n; ++j) {
            float p; int ty;
            if (!candidate(slots[j], cnt, p, ty)) continue;
            const float q = 1.f - p;
            if (q == 0.f) ++zeros; else qnz = qnz * q;
            ++cnt;
        }
        return __ballot(open && cnt < a.knum) != 0;
    });
    cnt = 0;
    scan_faces<true>(a, t, slots, [&](int n) {
        for (int j = 0; j < n; ++j) {
            float p; int ty;
            if (!candidate(slots[j], cnt, p, ty)) continue;
            ++cnt;
            const float q = 1.f - p;
            const float excl = (q != 0.f) ? (zeros == 0 ? qnz / q : 0.f) : (zeros == 1 ? qnz : 0.f);
            const float gd = g4.w * excl * (-(p * a.sigmainv) / s2);
            if (gd == 0.f) continue;
            const int e = ty / 3, reg = ty - e * 3;
            const float4 q0 = slots[j].p0, q1 = slots[j].p1;
            const float vx[3] = {q0.x, q0.z, q1.x}, vy[3] = {q0.y, q0.w, q1.y};
            const int iu = e, iv = (e == 2) ? 0 : e + 1;
            const float ux = vx[iu], uy = vy[iu], wx = vx[iv], wy = vy[iv];
            float dux = 0.f, duy = 0.f, dvx = 0.f, dvy = 0.f;
            if (reg == 0) { dux = -2.f * (t.x0 - ux); duy = -2.f * (t.y0 - uy); }
            else if (reg == 2) { dvx = -2.f * (t.x0 - wx); dvy = -2.f * (t.y0 - wy); }
            else {
                const float ex = wx - ux, ey = wy - uy, rx = t.x0 - ux, ry = t.y0 - uy;
                const float tt = (rx * ex + ry * ey) / (ex * ex + ey * ey);
                const float qx = t.x0 - (ux + tt * ex), qy = t.y0 - (uy + tt * ey);
                dux = -2.f * (1.f - tt) * qx; duy = -2.f * (1.f - tt) * qy;
                dvx = -2.f * tt * qx; dvy = -2.f * tt * qy;
            }
            const int f = __float_as_int(slots[j].p2.z);
            float* dq = a.dfxy + ((size_t)t.b * a.F + f) * 6;
            atomicAdd(dq + iu * 2, gd * dux * a.mult); atomicAdd(dq + iu * 2 + 1, gd * duy * a.mult);
            atomicAdd(dq + iv * 2, gd * dvx * a.mult); atomicAdd(dq + iv * 2 + 1, gd * dvy * a.mult);
        }
        return __ballot(open && cnt < a.knum) != 0;
    });
}

// Zero-fill of everything the backward accumulates into, in one launch (float4 grid-stride over three ranges).
__global__ __launch_bounds__(256) void zero3_kernel(float4* p0, size_t n0, float4* p1, size_t n1, float4* p2, size_t n2,
                                                    float* tail, size_t ntail) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n0; i += stride) p0[i] = z;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n1; i += stride) p1[i] = z;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) p2[i] = z;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ntail; i += stride) tail[i] = 0.f;
}

static RasterArgs make_args(const MMRenderDesc* d, const Workspace& w) {
    RasterArgs a;
    a.B = d->B; a.H = d->H; a.W = d->W; a.F = d->F; a.Ht = d->Ht; a.Wt = d->Wt; a.knum = d->knum;
    a.tiles_x = (d->W + MM_TILE_W * MM_BLOCK_WAVES - 1) / (MM_TILE_W * MM_BLOCK_WAVES);
    a.tiles_per_image = a.tiles_x * ((d->H + MM_TILE_H - 1) / MM_TILE_H);
    a.mult = d->multiplier; a.eps = d->eps; a.sigmainv = d->sigmainv; a.infl = d->boxlen * d->multiplier;
    a.bbox = w.bbox; a.geo = w.geo; a.valid = w.valid;
    a.face_uvs = d->face_uvs; a.fn = d->face_normals; a.textures = d->textures; a.lights = d->lights; a.bg = d->bg;
    a.rgba = d->rgba; a.face_idx = d->face_idx; a.imnormal = d->imnormal;
    a.grad_rgba = nullptr; a.grad_textures = nullptr; a.grad_lights = nullptr; a.grad_bg = nullptr;
    a.dfxy = w.dfxy; a.dfn = w.dfn;
    return a;
}

int launch_raster_fwd(const MMRenderDesc* d, const Workspace& w, hipStream_t s) {
    RasterArgs a = make_args(d, w);
    dim3 grid(a.tiles_per_image * d->B);
    ProfScope ps(d->prof_events, MM_PROF_RASTER_FWD, s);
    if (d->no_mask) hipLaunchKernelGGL(raster_fwd_kernel<true>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(raster_fwd_kernel<false>, grid, dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? MM_OK : MM_ERR_LAUNCH;
}

int launch_raster_bwd(const MMRenderDesc* d, const MMRenderGrads* g, const Workspace& w, hipStream_t s) {
    RasterArgs a = make_args(d, w);
    a.grad_rgba = g->grad_rgba; a.grad_textures = g->grad_textures; a.grad_lights = g->grad_lights; a.grad_bg = g->grad_bg;
    // zero: grad_textures | dfxy+dfn (contiguous in the workspace up to alignment padding) | grad_lights
    const size_t ntex = (size_t)d->B * 3 * d->Ht * d->Wt;
    const size_t nacc = ((char*)w.dfn - (char*)w.dfxy) / sizeof(float) + (size_t)d->B * d->F * 3;
    const size_t nl = (size_t)d->B * 9;
    {
    ProfScope pz(d->prof_events, MM_PROF_ZERO, s);
    if ((ntex % 4) != 0 || ((uintptr_t)g->grad_textures % 16) != 0) {
        if (hipMemsetAsync(g->grad_textures, 0, ntex * sizeof(float), s) != hipSuccess) return MM_ERR_LAUNCH;
        hipLaunchKernelGGL(zero3_kernel, dim3(256), dim3(256), 0, s, (float4*)w.dfxy, (nacc + 3) / 4, (float4*)nullptr, (size_t)0,
                           (float4*)nullptr, (size_t)0, g->grad_lights, nl);
    } else {
        const int blocks = (int)((ntex / 4 + 255) / 256 < 2048 ? (ntex / 4 + 255) / 256 : 2048);
        hipLaunchKernelGGL(zero3_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, (float4*)g->grad_textures, ntex / 4,
                           (float4*)w.dfxy, (nacc + 3) / 4, (float4*)nullptr, (size_t)0, g->grad_lights, nl);
    }
    }
    if (hipGetLastError() != hipSuccess) return MM_ERR_LAUNCH;
    dim3 grid(a.tiles_per_image * d->B);
    ProfScope pb(d->prof_events, MM_PROF_RASTER_BWD, s);
    if (d->no_mask) hipLaunchKernelGGL(raster_bwd_kernel<true>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(raster_bwd_kernel<false>, grid, dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? MM_OK : MM_ERR_LAUNCH;
}

}  // namespace mm
