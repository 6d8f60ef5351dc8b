// mm_raster_common.h -- pieces shared by the forward pixel kernels of libmm_render.so (gfx950):
//   mm_raster.hip   the fused render kernel (walk + texture + SH + composite + fused loss sums)
//   mm_dibr.hip     the un-fused kaolin-shaped dibr_rasterization (same walk, generic feature interpolation)
// Both evaluate the SAME fp32 expressions on the same values.
#pragma once
#include "mm_device.h"
#include "mm_order.h"

namespace mm {

struct RasterArgs {
    int B, H, W, F, Ht, Wt, knum, blocks_x, blocks_per_image;
    int bin_shift, nbx, nby, words;
    float mult, eps, sigmainv, infl;        // infl = boxlen * multiplier
    float kx, ky;                           // multiplier / W, multiplier / H: IEEE fp32 quotients formed once on the host -- the factor of the pixel-centre
                                            // convention (pixel_x / pixel_y); the kernels multiply, they never divide per tile or per pair
    const float4* geo;
    const uint64_t* binmask;                // candidates per bin: faces whose inflated pixel box touches it
    const float* face_uvs;
    const float* fn;                        // (B,F,3) unit normals
    const float* textures;
    const float* lights;
    const float* bg;
    float2* soft;                           // per pixel {soft-mask product state, id of the knum-th face taken (int bits)}
    int* fflag;                             // (B,F,2) [0] set for a face that wins a pixel, [1] for one taken into an uncovered pixel's silhouette product (nullptr: not wanted)
    const float* gt; long long* ltot;        // fused recon_data sums (gt == nullptr: off)
    float contour;                           // ... its contour weight (0: no contour term; > 0: H and W are multiples of 4)
    int* trcnt; int ntx_tex, ntiles_tex;     // (B,ntiles) covered pixels per 32x32-texel tile under their bilinear footprint: sizes the backward's record lists
    const unsigned short* order;            // (B, 4*blocks) tile slots, heavy first; nullptr: natural order
    const int* nheavy;                      // (B,4) plan kernel: how many of an image's first tiles are walked cooperatively; how many tiles are not empty
    int spread;                             // sorted order: eight consecutive workgroups = eight ranks of one image (walk_image_rank); 2: the four tiles of a block on one XCD
    int block_sort;                         // the order kernel sorts 16x16 blocks, a block's four tiles stay together (bins of 16 pixels or more)
    const int* bincount;                    // (B,nbins) candidates per screen bin, or nullptr (small screens: the order kernel counts the mask bits itself)
    // outputs
    float* rgba;
    int32_t* face_idx;
    float* imnormal;
    // un-fused dibr_rasterization entry point only (mm_dibr.hip)
    const float* feats; int D;              // (B,F,3,D) per-corner features
    float* interp; float* soft_out; long long* face_idx64;
    int options;                            // MM_OPT_* bits
};

#ifndef MM_PAIR_ROUND
#define MM_PAIR_ROUND 512
#endif
#ifndef MM_HEAVY_CAND
#define MM_HEAVY_CAND 192     // a tile with at least this many candidates (three batches) is walked by four waves together ...
#endif
#ifndef MM_HEAVY_MAX
#define MM_HEAVY_MAX 32       // ... if it is among the image's MM_HEAVY_MAX heaviest
#endif

struct TileCtx {
    int b, blk, px, py, tx0, ty0, lane, wave;   // wave = quadrant of the 16x16 block `blk`
    bool in_img;
    bool empty;                             // no face can touch the tile (known from the order kernel): nothing to walk
    float x0, y0;
    float xf0, yf0;                         // (float)(2 tx0 + 1 - W), (float)(H - 2 ty0 - 1): the tile's first column / row in the pixel-centre convention;
                                            // column i is kx * (xf0 + 2 i), row i is ky * (yf0 - 2 i)  (exact integers: the same floats as pixel_x_k / pixel_y_k)
    const uint64_t* mask;                   // this wave's bin row of candidate bits: `words` 64-bit words
};

__device__ inline void tile_pixels(const RasterArgs& a, TileCtx& t) {
    t.px = t.tx0 + (t.lane & 7); t.py = t.ty0 + (t.lane >> 3);
    t.in_img = t.px < a.W && t.py < a.H;
    t.xf0 = (float)(2 * t.tx0 + 1 - a.W); t.yf0 = (float)(a.H - 2 * t.ty0 - 1);
    if (t.empty) { t.x0 = t.y0 = 0.f; return; }                 // wave-uniform: an empty tile never looks at pixel centres
    t.x0 = pixel_x_k(t.px, a.W, a.kx); t.y0 = pixel_y_k(t.py, a.H, a.ky);
}

__device__ inline void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// box-vs-tile for ONE candidate (this lane's), both of its boxes in one pass over the tile's 8 columns and 8 rows: bit (r*8+c) set iff
// pixel (row r, column c) of the tile passes the separable box test  !(x < lo || x > hi)  -- the same comparisons on the same floats as
// the per-pixel test.  mh: the face's own box (colour; only if `hard`), ms: the box inflated by the silhouette margin (only if `soft`).
// The tile's pixel centres are re-formed here (an exact integer add and the convention's one multiplication each) instead of living in
// sixteen registers across the whole walk.
__device__ inline uint64_t outer_mask(unsigned col, unsigned row) {
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        lo |= ((row >> r) & 1u) ? (col << (8 * r)) : 0u;
        hi |= ((row >> (r + 4)) & 1u) ? (col << (8 * r)) : 0u;
    }
    return ((uint64_t)hi << 32) | lo;
}
// the float just above / just below v (finite v):  x <= v  <=>  x < next_up(v),   x >= v  <=>  x > next_down(v)
__device__ inline float next_up(float v) {
    const unsigned b = __float_as_uint(v + 0.f);                 // (-0 -> +0)
    return __uint_as_float((b & 0x80000000u) ? b - 1u : b + 1u);
}
__device__ inline float next_down(float v) {
    const unsigned b = __float_as_uint(v + 0.f);
    return __uint_as_float(b == 0u ? 0x80000001u : ((b & 0x80000000u) ? b + 1u : b - 1u));
}
__device__ inline void box_masks(const RasterArgs& a, const TileCtx& t, float xlo, float ylo, float xhi, float yhi, bool hard, bool soft, int mode,
                                 uint64_t& mh, uint64_t& ms) {
MM_FP_EXACT
    float xlo2 = xlo - a.infl, ylo2 = ylo - a.infl, xhi2 = xhi + a.infl, yhi2 = yhi + a.infl;
    // An OPEN border (MM_OPT_BBOX_HALF_OPEN / MM_OPT_BBOX_MIN_CLOSED_MAX_OPEN: a centre exactly on it is outside) is the closed test against
    // the neighbouring float: the same decisions as  x <= lo / x >= hi, and one code path for the three forms (the kernel's register
    // allocation is that of its hungriest path, taken or not)
    if (mode & 1) { xlo = next_up(xlo); ylo = next_up(ylo); xlo2 = next_up(xlo2); ylo2 = next_up(ylo2); }                   // (wave-uniform)
    if (mode & 2) { xhi = next_down(xhi); yhi = next_down(yhi); xhi2 = next_down(xhi2); yhi2 = next_down(yhi2); }
    unsigned ch = 0, rh = 0, cs = 0, rs = 0;
    float xf0 = t.xf0, yf0 = t.yf0;
    asm volatile("" : "+v"(xf0), "+v"(yf0));                     // (opaque: keeps the sixteen centres from being hoisted out of the walk's loop into registers)
    if (__ballot(soft)) {                                        // (wave-uniform) some candidate of the batch may still matter to the silhouette
#pragma unroll
        for (int i = 0; i < MM_TILE; ++i) {
            const float x = a.kx * (xf0 + (float)(2 * i)), y = a.ky * (yf0 - (float)(2 * i));
            ch |= (unsigned)(!(x < xlo || x > xhi)) << i; rh |= (unsigned)(!(y < ylo || y > yhi)) << i;
            cs |= (unsigned)(!(x < xlo2 || x > xhi2)) << i; rs |= (unsigned)(!(y < ylo2 || y > yhi2)) << i;
        }
    } else {                                                     // colour only: half the comparisons
#pragma unroll
        for (int i = 0; i < MM_TILE; ++i) {
            const float x = a.kx * (xf0 + (float)(2 * i)), y = a.ky * (yf0 - (float)(2 * i));
            ch |= (unsigned)(!(x < xlo || x > xhi)) << i; rh |= (unsigned)(!(y < ylo || y > yhi)) << i;
        }
    }
    mh = hard ? outer_mask(ch, rh) : 0ull;
    ms = soft ? outer_mask(cs, rs) : 0ull;
}

// Balanced evaluation of a batch's (row, column) pairs: every lane owns one ROW of the bit matrix `m` (a candidate, or a
// pixel) and the set bits are its columns.  The pairs of all lanes are laid out row-major in LDS and evaluated 64 at a
// time by WHICHEVER lane, eval(row, col); results are combined by the caller through commutative, exact LDS atomics
// (64-bit max / integer add), so the wave's critical path is pairs/64 evaluations, not its busiest lane, and the outcome
// does not depend on evaluation order.
template <class Stage, class Eval>
__device__ inline void pair_parallel(const TileCtx& t, Stage* st, uint64_t m, Eval&& eval) {
    int total;
    int k = wave_prefix_excl(__popcll(m), t.lane, total);       // index of this lane's next unwritten pair
    uint64_t rem = m;
    for (int base = 0; base < total; base += MM_PAIR_ROUND) {
        const int lim = min(MM_PAIR_ROUND, total - base);
        while (rem && k < base + lim) {                          // every set bit is visited exactly once overall
            const int j = __ffsll((unsigned long long)rem) - 1;
            rem &= rem - 1;
            st->pairs[k - base] = (unsigned short)((t.lane << 8) | j);
            ++k;
        }
        wave_lds_sync();
        if (lim <= 64) {                                         // wave-uniform: one pair per lane at most, no dummy second evaluation
            const bool live = t.lane < lim;
            const unsigned pr = st->pairs[live ? t.lane : 0];
            eval((int)(pr >> 8), (int)(pr & 255u), live);
        } else {
            for (int p = t.lane; p < lim; p += 128) {            // two independent pairs per trip: ILP for a lone wave
                const unsigned pr0 = st->pairs[p];
                const bool two = p + 64 < lim;
                const unsigned pr1 = two ? st->pairs[p + 64] : pr0;
                eval((int)(pr0 >> 8), (int)(pr0 & 255u), true);
                if (__ballot(two)) eval((int)(pr1 >> 8), (int)(pr1 & 255u), two);   // wave-uniform
            }
        }
        wave_lds_sync();
    }
}

// (z, -rank) packed so that an unsigned 64-bit max is kaolin's "strict z > best, lowest index on ties"; rank = any value
// that grows with the face index (the face id itself, or its position in an index-ordered list)
__device__ inline unsigned depth_ord(float z) {                  // unsigned order of a float: a < b  <=>  depth_ord(a) < depth_ord(b)
    const unsigned bits = __float_as_uint(z + 0.f);              // -0 -> +0: equal depths must tie
    return (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
}
__device__ inline unsigned long long depth_key(float z, int rank) {
    return ((unsigned long long)depth_ord(z) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)rank);
}
// Upper bound of the depth a face can give ANY pixel, as a depth_ord: inside the face the barycentrics are a convex combination (up to
// a few ulps: every w_i in [0, 1], their sum <= 1 + 3 ulp), so the interpolated z never exceeds the largest corner z by more than
// ~6e-7 of the largest |z|; 1e-5 of it is added.  A face whose bound lies below what a pixel already holds can never win that pixel
// (strictly below: equal depths, which the lower face index wins, are never cut) -- the walk's early-z, exact.
__device__ inline unsigned depth_bound(float az, float bz, float cz) {
MM_FP_EXACT
    const float zmax = fmaxf(fmaxf(az, bz), cz), amax = fmaxf(fmaxf(fabsf(az), fabsf(bz)), fabsf(cz));
    const float zb = zmax + 1e-5f * amax;
    return (zb == zb && amax < INFINITY) ? depth_ord(zb) : 0xFFFFFFFFu;     // (NaN / inf corners: never cut)
}
__device__ inline int depth_key_rank(unsigned long long k) { return (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull)); }

struct Hit { int f; float w0, w1, w2; };

// K1, one (candidate j, pixel l) pair of the staged batch: barycentrics, inside test, depth -> 64-bit LDS max.
// straight-line on purpose (two of these are interleaved per trip): the IEEE divisions the oracle takes
template <class Stage>
__device__ inline void hard_pair(const RasterArgs& a, const TileCtx& t, Stage* st, Stage* acc, int j, int l, bool live) {   // st: staged candidates; acc: the tile's results
    const float x0 = pixel_x_k(t.tx0 + (l & 7), a.W, a.kx), y0 = pixel_y_k(t.ty0 + (l >> 3), a.H, a.ky);
    const float4 p0 = st->p0[j], p1 = st->p1[j], p2 = st->p2[j];
    float w0, w1, w2, nrm;
    bary_weights(p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, x0, y0, a.eps, (a.options & MM_OPT_BARY_ONE_MINUS) != 0, w0, w1, w2, nrm);
    const float z0 = (w0 * p1.z + w1 * p1.w) + w2 * p2.x;
    if (live && !(w0 < 0.f || w1 < 0.f || w2 < 0.f) && z0 > -INFINITY)
        atomicMax(&acc->key[l], depth_key(z0, __float_as_int(p2.z)));
}

// K1 for ALL (candidate, pixel) pairs of a flush, in two phases.  Where faces overlap heavily (a fine mesh with per-vertex noise: hundreds of
// boxes over every pixel) most box pairs fail the inside test, and the three IEEE divisions of the barycentrics were spent on them too.
//   phase A  every pair, two per lane in packed fp32 (v_pk_*): the three edge functions k0, k1, k2 and their padded sum -- the SAME
//            expressions, in the same order, as edge_weights / bary_weights -- and a sign test that rejects a pair only where the
//            quotient k_i / sum is certainly negative (opposite signs, neither operand within 40 orders of magnitude of the format's
//            ends): exactly the pairs the oracle's `w_i < 0` rejects, minus freak magnitudes, which stay in.  Also EARLY-Z: a pair that
//            cannot reach the depth its pixel already holds (an upper bound of its interpolated depth lies strictly below it) is
//            dropped -- faces come in index order, not depth order, so after the first few a pixel under forty overlapping boxes
//            lets only the occasional nearer face through.  Never cuts a winner: equal depths (the lower index wins) stay in.
//            Survivors are ballot-compacted IN PLACE at the front of the pair list (writes never pass the read cursor).
//   phase B  the survivors, densely: hard_pair as before (divisions, inside test, depth, 64-bit LDS max).
// m: the flush's colour masks, rows = candidates (by_cand) or pixels; results do not depend on the order of evaluation.
#ifndef MM_HARD_DIRECT
#define MM_HARD_DIRECT 192
#endif
#ifndef MM_PAIR_Z
#define MM_PAIR_Z 0            // 1: early-z per PAIR (interpolated depth with a hardware reciprocal) instead of per face (its largest corner depth): fewer
#endif                         // survivors, six more registers -- and this kernel lives on its occupancy
typedef float mm_f2 __attribute__((ext_vector_type(2)));
template <class Stage>
__device__ inline void hard_pairs(const RasterArgs& a, const TileCtx& t, Stage* st, uint64_t m, bool by_cand) {
    int total;
    int k = wave_prefix_excl(__popcll(m), t.lane, total);       // index of this lane's next unwritten pair
    uint64_t rem = m;
    const bool one_minus = (a.options & MM_OPT_BARY_ONE_MINUS) != 0;
    for (int base = 0; base < total; base += MM_PAIR_ROUND) {
        const int lim = min(MM_PAIR_ROUND, total - base);
        while (rem && k < base + lim) {                          // every set bit is visited exactly once overall
            const int j = __ffsll((unsigned long long)rem) - 1;
            rem &= rem - 1;
            st->pairs[k - base] = (unsigned short)(by_cand ? ((t.lane << 8) | j) : ((j << 8) | t.lane));   // (candidate << 8) | pixel
            ++k;
        }
        wave_lds_sync();
        int w = 0;                                               // survivors so far (wave-uniform): they occupy pairs[0, w)
        if (total <= MM_HARD_DIRECT) w = lim;                    // (wave-uniform) a short list: the filter pass would cost more trips than it saves
        else
        for (int q0 = 0; q0 < lim; q0 += 128) {
            const bool v0 = q0 + t.lane < lim, v1 = q0 + 64 + t.lane < lim;
            const unsigned e0 = st->pairs[v0 ? q0 + t.lane : 0], e1 = st->pairs[v1 ? q0 + 64 + t.lane : 0];
            const int j0 = (int)(e0 >> 8), j1 = (int)(e1 >> 8), l0 = (int)(e0 & 63u), l1 = (int)(e1 & 63u);
            const float4 A0 = st->p0[j0], B0 = st->p1[j0], A1 = st->p0[j1], B1 = st->p1[j1];
            // early-z: what the pixel holds by now (the high word of its key; 0 = nothing yet) against an upper bound of the depth this pair
            // would give it -- the interpolation with a hardware reciprocal (1 ulp) plus 2e-5 of the corner depths' magnitudes
#if MM_PAIR_Z
            const float cz0 = st->p2[j0].x, cz1 = st->p2[j1].x;
#else
            const unsigned zb0 = __float_as_uint(st->p2[j0].w), zb1 = __float_as_uint(st->p2[j1].w);   // the face's depth bound
#endif
            const unsigned kh0 = (unsigned)(st->key[l0] >> 32), kh1 = (unsigned)(st->key[l1] >> 32);
            bool pass0, pass1;
            {
MM_FP_EXACT
                const mm_f2 x0 = {a.kx * (t.xf0 + (float)(2 * (l0 & 7))), a.kx * (t.xf0 + (float)(2 * (l1 & 7)))};
                const mm_f2 y0 = {a.ky * (t.yf0 - (float)(2 * (l0 >> 3))), a.ky * (t.yf0 - (float)(2 * (l1 >> 3)))};
                const mm_f2 ax = {A0.x, A1.x}, ay = {A0.y, A1.y}, bx = {A0.z, A1.z}, by = {A0.w, A1.w}, cx = {B0.x, B1.x}, cy = {B0.y, B1.y};
                const mm_f2 aex = ax - x0, aey = ay - y0, bex = bx - x0, bey = by - y0, cex = cx - x0, cey = cy - y0;
                const mm_f2 k0 = bex * cey - bey * cex, k1 = cex * aey - cey * aex, k2 = aex * bey - aey * bex;
                mm_f2 nrm = (k0 + k1) + k2;
                nrm.x += one_minus ? a.eps : copysignf(a.eps, nrm.x); nrm.y += one_minus ? a.eps : copysignf(a.eps, nrm.y);
                // certainly negative quotient: opposite signs and magnitudes far from the ends of the format (1e-12 / 1e12 >= 1e-24: normal)
                auto neg = [](float kk, float nn) { return ((kk < 0.f) != (nn < 0.f)) && fabsf(kk) >= 1e-12f; };
                const bool sane0 = fabsf(nrm.x) <= 1e12f, sane1 = fabsf(nrm.y) <= 1e12f;
                const bool rej0 = sane0 && ((!one_minus && neg(k0.x, nrm.x)) || neg(k1.x, nrm.x) || neg(k2.x, nrm.x));
                const bool rej1 = sane1 && ((!one_minus && neg(k0.y, nrm.y)) || neg(k1.y, nrm.y) || neg(k2.y, nrm.y));
#if MM_PAIR_Z
                const mm_f2 az = {B0.z, B1.z}, bz = {B0.w, B1.w}, cz = {cz0, cz1};
                const mm_f2 zn = (k0 * az + k1 * bz) + k2 * cz;
                const float zu0 = zn.x * __builtin_amdgcn_rcpf(nrm.x) + 2e-5f * ((fabsf(az.x) + fabsf(bz.x)) + fabsf(cz.x));
                const float zu1 = zn.y * __builtin_amdgcn_rcpf(nrm.y) + 2e-5f * ((fabsf(az.y) + fabsf(bz.y)) + fabsf(cz.y));
                const bool near0 = !(zu0 == zu0) || depth_ord(zu0) >= kh0, near1 = !(zu1 == zu1) || depth_ord(zu1) >= kh1;   // (NaN: stays in)
#else
                const bool near0 = zb0 >= kh0, near1 = zb1 >= kh1;
#endif
                pass0 = v0 && !rej0 && near0; pass1 = v1 && !rej1 && near1;
            }
            const uint64_t b0 = __ballot(pass0), b1 = __ballot(pass1);
            wave_lds_sync();                                     // (every read of this trip is done; the writes stay behind the read cursor)
            if (pass0) st->pairs[w + ballot_rank(b0)] = (unsigned short)e0;
            if (pass1) st->pairs[w + __popcll(b0) + ballot_rank(b1)] = (unsigned short)e1;
            w += __popcll(b0) + __popcll(b1);
        }
        wave_lds_sync();
        for (int q = t.lane; q < w; q += 64) {
            const unsigned pr = st->pairs[q];
            hard_pair(a, t, st, st, (int)(pr >> 8), (int)(pr & 63u), true);
        }
        wave_lds_sync();
    }
}

// K3, one (pixel l, candidate j) pair: factor q = 1 - exp(-sigma d^2) folded into the pixel's integer log2 sum.
// sig2 = sigmainv / multiplier^2 (d is in multiplier units).
template <class Stage>
__device__ inline void soft_pair(const RasterArgs& a, const TileCtx& t, Stage* st, Stage* acc, float sig2, int l, int j, bool live) {
#ifdef MM_BOUND_NOSOFT                                          // (bound experiment, WRONG results: the silhouette pairs cost nothing)
    return;
#endif
    const float x0 = pixel_x_k(t.tx0 + (l & 7), a.W, a.kx), y0 = pixel_y_k(t.ty0 + (l >> 3), a.H, a.ky);
    const float4 p0 = st->p0[j], p1 = st->p1[j];
    const float q = soft_factor(x0, y0, p0, p1, sig2);
    if (live) {
        if (q == 0.f) atomicAdd(&acc->zeros[l], 1);
        else {                                                   // log2(q) in 2^-32 fixed point: floor part and 32 fraction bits
            const float x = __builtin_amdgcn_logf(q);
            const float hi = floorf(x);
            const unsigned lo = (unsigned)((x - hi) * 4294967296.f);
            atomicAdd((unsigned long long*)&acc->logsum[l], ((unsigned long long)(long long)(int)hi << 32) + lo);
        }
    }
}

// K3 for all (pixel, candidate) pairs of a flush, two per lane in packed fp32: the segment distances of soft_factor / seg_dist2_fast step
// by step on both halves at once (v_pk_add / v_pk_mul are the same IEEE operations as their scalar forms, the reciprocal, exp2 and log2
// are issued per half): q is BIT-IDENTICAL to soft_factor's, which the backward divides the stored product by.
// sm: this lane's pixel, bit j = queued candidate j it takes.
__device__ inline mm_f2 seg_dist2_pk(mm_f2 px, mm_f2 py, mm_f2 ux, mm_f2 uy, mm_f2 vx, mm_f2 vy) {
MM_FP_EXACT
    const mm_f2 ex = vx - ux, ey = vy - uy, rx = px - ux, ry = py - uy;
    const mm_f2 len2 = ex * ex + ey * ey;
    const mm_f2 dot = rx * ex + ry * ey;
    mm_f2 tt;
    tt.x = (len2.x > 0.f) ? dot.x * __builtin_amdgcn_rcpf(len2.x) : 0.f;
    tt.y = (len2.y > 0.f) ? dot.y * __builtin_amdgcn_rcpf(len2.y) : 0.f;
    tt.x = fminf(fmaxf(tt.x, 0.f), 1.f); tt.y = fminf(fmaxf(tt.y, 0.f), 1.f);
    const mm_f2 qx = rx - tt * ex, qy = ry - tt * ey;
    return qx * qx + qy * qy;
}
template <class Stage>
__device__ inline void soft_pairs(const RasterArgs& a, const TileCtx& t, Stage* st, uint64_t sm, float sig2) {
#ifdef MM_BOUND_NOSOFT
    return;
#endif
    int total;
    int k = wave_prefix_excl(__popcll(sm), t.lane, total);      // index of this lane's next unwritten pair
    uint64_t rem = sm;
    for (int base = 0; base < total; base += MM_PAIR_ROUND) {
        const int lim = min(MM_PAIR_ROUND, total - base);
        while (rem && k < base + lim) {                          // every set bit is visited exactly once overall
            const int j = __ffsll((unsigned long long)rem) - 1;
            rem &= rem - 1;
            st->pairs[k - base] = (unsigned short)((j << 8) | t.lane);   // (candidate << 8) | pixel
            ++k;
        }
        wave_lds_sync();
        for (int q0 = 0; q0 < lim; q0 += 128) {
            const bool v0 = q0 + t.lane < lim, v1 = q0 + 64 + t.lane < lim;
            const unsigned e0 = st->pairs[v0 ? q0 + t.lane : 0], e1 = st->pairs[v1 ? q0 + 64 + t.lane : 0];
            const int j0 = (int)(e0 >> 8), j1 = (int)(e1 >> 8), l0 = (int)(e0 & 63u), l1 = (int)(e1 & 63u);
            const float4 A0 = st->p0[j0], B0 = st->p1[j0], A1 = st->p0[j1], B1 = st->p1[j1];
            float q[2];
            {
MM_FP_EXACT
                const mm_f2 x0 = {a.kx * (t.xf0 + (float)(2 * (l0 & 7))), a.kx * (t.xf0 + (float)(2 * (l1 & 7)))};
                const mm_f2 y0 = {a.ky * (t.yf0 - (float)(2 * (l0 >> 3))), a.ky * (t.yf0 - (float)(2 * (l1 >> 3)))};
                const mm_f2 ax = {A0.x, A1.x}, ay = {A0.y, A1.y}, bx = {A0.z, A1.z}, by = {A0.w, A1.w}, cx = {B0.x, B1.x}, cy = {B0.y, B1.y};
                const mm_f2 d0 = seg_dist2_pk(x0, y0, ax, ay, bx, by), d1 = seg_dist2_pk(x0, y0, bx, by, cx, cy), d2 = seg_dist2_pk(x0, y0, cx, cy, ax, ay);
#ifdef MM_BOUND_SOFT1                                           // (bound experiment, WRONG results: one edge instead of three)
                const float dx = d0.x, dy = d0.y;
#else
                const float dx = fminf(fminf(d0.x, d1.x), d2.x), dy = fminf(fminf(d0.y, d1.y), d2.y);
#endif
                q[0] = 1.f - __builtin_amdgcn_exp2f(-(dx * sig2) * 1.4426950408889634f);
                q[1] = 1.f - __builtin_amdgcn_exp2f(-(dy * sig2) * 1.4426950408889634f);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bool live = u ? v1 : v0;
                const int l = u ? l1 : l0;
                if (live) {
                    if (q[u] == 0.f) atomicAdd(&st->zeros[l], 1);
                    else {                                       // log2(q) in 2^-32 fixed point: floor part and 32 fraction bits
                        const float x = __builtin_amdgcn_logf(q[u]);
                        const float hi = floorf(x);
                        const unsigned lo = (unsigned)((x - hi) * 4294967296.f);
                        atomicAdd((unsigned long long*)&st->logsum[l], ((unsigned long long)(long long)(int)hi << 32) + lo);
                    }
                }
            }
        }
        wave_lds_sync();
    }
}

// soft-mask candidates of this lane in the staged batch: its inflated-box hits, in order, truncated so that the lane
// never takes more than `room` further faces (kaolin keeps the first knum).
__device__ inline uint64_t soft_take(uint64_t sm, bool open, int room) {
    if (!open || room <= 0) return 0;
    if (__popcll(sm) > room) {                                   // keep the first `room` set bits (rare)
        uint64_t kept = 0;
        for (int i = 0; i < room; ++i) { const uint64_t low = sm & (~sm + 1); kept |= low; sm ^= low; }
        sm = kept;
    }
    return sm;
}

// x >= 0 (a sum of at most 64 terms of size <= 3) as 32.32 fixed point, exactly: integer part and 32 fraction bits
__device__ inline unsigned long long fixed32(float x) {
    const unsigned hi = (unsigned)x;
    return ((unsigned long long)hi << 32) | (unsigned)((x - (float)hi) * 4294967296.f);
}

struct SoftState { float qnz; int zeros, lastf; };

// recon_data's contour term of one pixel (networks.py:379-387): contour(m) = |m - up4(down4(m))| for the rendered and the ground-truth mask,
// term = (contour(pred) - contour(gt))^2.  The two nearest-neighbour resamplings of F.interpolate pick, for H and W multiples of 4, the
// top-left pixel of the pixel's 4x4 block (index 4 * (y >> 2): down4 takes row floor(j * 4.0) = 4 j, up4 takes floor(y * 0.25) = y >> 2, both
// exact in torch's float arithmetic) -- a pixel of the same 8x8 tile, i.e. lane (lane & 0x24) of this wave (lane = 8 * row + column).
// Every lane of the wave must call.
__device__ inline float contour_term(float alpha, float gm, bool in_img, int lane) {
    const int src = lane & 0x24;
    const float as = __shfl(alpha, src, 64), gs = __shfl(gm, src, 64);
    if (!in_img) return 0.f;
    const float cp = fabsf(alpha - as), cg = fabsf(gm - gs);
    return (cp - cg) * (cp - cg);
}

// ---- shading (a9-a11), stores, fused recon_data partial sums.  Uncovered pixels carry zero features exactly like kaolin's
// interpolated_features.  key = the pixel's depth key after the walk (0: uncovered).
// (lanes outside a ragged image only stay for the fused loss reduction: they address a clamped pixel and store nothing)
//
// The epilogue is a chain of DEPENDENT trips to memory (each ~1-2.5 us under load, and 40 % of this kernel's wave time when
// every `if (corner inside) tc += tex[..]` was its own branch + wait): it is written so that exactly two remain --
//   trip 1  everything that depends on the winner's id alone: its geometry, normal and corner uvs, plus background / ground truth
//   trip 2  the twelve texels, from clamped (always valid) addresses, unconditionally; a corner outside contributes an exact zero
// The arithmetic (expressions, order, roundings) is unchanged.
// kContour: the fused loss carries recon_data's contour term (a.contour > 0; chosen by the host -- the reference's default is none, train.py:115)
template <bool kNoMask, bool kContour>
__device__ inline void shade_store(const RasterArgs& a, const TileCtx& t, unsigned long long key, const SoftState& ss) {
    if (!t.in_img && !a.gt) return;
    const int cpx = min(t.px, a.W - 1), cpy = min(t.py, a.H - 1);
    const size_t pix = ((size_t)t.b * a.H + cpy) * a.W + cpx;
    const size_t hw = (size_t)a.H * a.W, pin = (size_t)cpy * a.W + cpx;
    Hit h;
    h.f = key != 0ull ? depth_key_rank(key) : -1; h.w0 = h.w1 = h.w2 = 0.f;
    const bool any = __ballot(h.f >= 0) != 0;                    // wave-uniform: more than half of all tiles have no covered pixel
    // ---- trip 1
    float bgv[3] = {0.f, 0.f, 0.f}, gtv[4] = {0.f, 0.f, 0.f, 0.f};
    if (kNoMask) {
#pragma unroll
        for (int c = 0; c < 3; ++c) bgv[c] = a.bg[((size_t)t.b * 3 + c) * hw + pin];
    }
    if (a.gt) {                                                  // (clamped pixel: a valid address in every lane)
#pragma unroll
        for (int c = 0; c < 4; ++c) gtv[c] = a.gt[((size_t)t.b * 4 + c) * hw + pin];
    }
    float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0;
    float fu[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, n0 = 0.f, n1 = 0.f, n2 = 0.f;
    if (any) {                                                   // uncovered lanes read face 0's records and ignore them
        const int fs = max(h.f, 0);
        const float4* geo = a.geo + ((size_t)t.b * a.F + fs) * 3;
        p0 = geo[0]; p1 = geo[1];
        const float2* fuv = (const float2*)(a.face_uvs + (size_t)fs * 6);
        const float2 u0 = fuv[0], u1 = fuv[1], u2 = fuv[2];
        fu[0] = u0.x; fu[1] = u0.y; fu[2] = u1.x; fu[3] = u1.y; fu[4] = u2.x; fu[5] = u2.y;
        const float* nn = a.fn + ((size_t)t.b * a.F + fs) * 3;
        n0 = nn[0]; n1 = nn[1]; n2 = nn[2];
    }
    float nx = 0.f, ny = 0.f, nz = 0.f;
    int fpx0 = -1, fpy0 = 0, fpx1 = 0, fpy1 = 0;                 // texel footprint of a covered in-image pixel (fpx0 < 0: none)
    float out[4];
    float L[9];                                                  // lights in the order of sh_bands (x, z, y): MM_OPT_SH_ORDER_XYZ pairs
#pragma unroll                                                   // the user's lights 2 / 3 with the y / z bands instead
    for (int i = 0; i < 9; ++i) L[i] = a.lights[t.b * 9 + i];
    if (a.options & MM_OPT_SH_ORDER_XYZ) { const float tmp = L[2]; L[2] = L[3]; L[3] = tmp; }
    if (!any) {
        // No lane of the tile is covered.  The general path below then computes, per lane,
        //   m = 0, n = 0  ->  coef = C0*L0 + (0 - C6B)*L6   (the other seven bands are products with 0)
        //   no_mask: (tc*0 + g*(1-0)) * coef = g * coef;   white: (tc*0)*coef + 1*(1-0) = 1        (finite texels / lights)
        // -- the same roundings without the uv -> bilinear -> twelve-texel chain.
        const float coef = MM_SH_C0 * L[0] + (0.f - MM_SH_C6B) * L[6];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float val = kNoMask ? bgv[c] * coef : 1.f;
            out[c] = val < 0.f ? 0.f : (val > 1.f ? 1.f : val);
        }
    } else {
        float m = 0.f, u = 0.f, v = 0.f;
        {
            float w0, w1, w2, nrm;                               // barycentrics of the winner (same expressions, same values as the walk's)
            bary_weights(p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, t.x0, t.y0, a.eps, (a.options & MM_OPT_BARY_ONE_MINUS) != 0, w0, w1, w2, nrm);
            if (h.f >= 0) { h.w0 = w0; h.w1 = w1; h.w2 = w2; }
        }
        if (h.f >= 0) {
            m = (h.w0 + h.w1) + h.w2;
            u = (h.w0 * fu[0] + h.w1 * fu[2]) + h.w2 * fu[4];
            v = (h.w0 * fu[1] + h.w1 * fu[3]) + h.w2 * fu[5];
            nx = (h.w0 * n0 + h.w1 * n0) + h.w2 * n0;
            ny = (h.w0 * n1 + h.w1 * n1) + h.w2 * n1;
            nz = (h.w0 * n2 + h.w1 * n2) + h.w2 * n2;
        }
        const Bilin s = bilin_setup(u, v, a.Ht, a.Wt);
        const bool inw = s.x0 < a.Wt && s.y0 < a.Ht, ine = s.x1 < a.Wt && s.y0 < a.Ht;
        const bool isw = s.x0 < a.Wt && s.y1 < a.Ht, ise = s.x1 < a.Wt && s.y1 < a.Ht;
        if (h.f >= 0 && t.in_img) { fpx0 = s.x0; fpy0 = s.y0; fpx1 = s.x1; fpy1 = s.y1; }
        // ---- trip 2: twelve loads in flight together
        const int cx0 = min(max(s.x0, 0), a.Wt - 1), cx1 = min(max(s.x1, 0), a.Wt - 1);
        const int cy0 = min(max(s.y0, 0), a.Ht - 1), cy1 = min(max(s.y1, 0), a.Ht - 1);
        float tq[3][4];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* tex = a.textures + ((size_t)t.b * 3 + c) * a.Ht * a.Wt;
            tq[c][0] = tex[(size_t)cy0 * a.Wt + cx0]; tq[c][1] = tex[(size_t)cy0 * a.Wt + cx1];
            tq[c][2] = tex[(size_t)cy1 * a.Wt + cx0]; tq[c][3] = tex[(size_t)cy1 * a.Wt + cx1];
        }
        float bnd[9];
        sh_bands(nx, ny, nz, bnd);
        float coef = 0.f;
#pragma unroll
        for (int i = 0; i < 9; ++i) coef += bnd[i] * L[i];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float tc = 0.f;                                      // (x + 0*w = x exactly: a corner outside the texture leaves the sum as it was)
            tc += (inw ? tq[c][0] : 0.f) * s.wnw;
            tc += (ine ? tq[c][1] : 0.f) * s.wne;
            tc += (isw ? tq[c][2] : 0.f) * s.wsw;
            tc += (ise ? tq[c][3] : 0.f) * s.wse;
            float val;
            if (kNoMask) {
                const float g = bgv[c];
                val = (tc * m + g * (1.f - m)) * coef;
            } else {
                val = (tc * m) * coef + 1.f * (1.f - m);
            }
            out[c] = val < 0.f ? 0.f : (val > 1.f ? 1.f : val);
        }
    }
    const float keepprod = ss.zeros > 0 ? 0.f : ss.qnz;
    out[3] = (h.f >= 0) ? 1.f : (1.f - keepprod);
    if (t.in_img) {
        *(float4*)(a.rgba + pix * 4) = make_float4(out[0], out[1], out[2], out[3]);
        a.face_idx[pix] = h.f;
#ifndef MM_NO_OWN_FLAG
        if (h.f >= 0 && a.fflag) a.fflag[((size_t)t.b * a.F + h.f) * 2] = 1;     // "owns a pixel" (idempotent plain store: the backward sweeps this face)
#endif
        a.soft[pix] = make_float2((h.f >= 0 || ss.zeros >= 2) ? 0.f : (ss.zeros == 1 ? -ss.qnz : ss.qnz), __int_as_float(ss.lastf));
        if (a.imnormal) { a.imnormal[pix * 3] = nx; a.imnormal[pix * 3 + 1] = ny; a.imnormal[pix * 3 + 2] = nz; }
    }
    if (a.gt) {                                                  // recon_data terms of this tile (networks.py:370-377)
        float l1 = 0.f, up = 0.f, down = 0.f, cs = 0.f;
        if (t.in_img) {
            const float gm = gtv[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float gi = gtv[c] * gm + 1.f * (1.f - gm);
                const float pi = out[c] * gm + 1.f * (1.f - gm);
                l1 += fabsf(pi - gi);
            }
            up = out[3] * gm; down = (out[3] + gm) - up;
        }
        if (kContour) cs = contour_term(out[3], gtv[3], t.in_img, t.lane);   // (every lane takes part in the exchange)
        l1 = wave_sum(l1); up = wave_sum(up); down = wave_sum(down);
        if (kContour) cs = wave_sum(cs);
        if (t.lane == 0) {                                       // exact in 2^-32 fixed point; integer adds commute: deterministic totals
            unsigned long long* row = (unsigned long long*)(a.ltot + ((size_t)t.b * MM_LSUB + ((t.blk * 4 + t.wave) & (MM_LSUB - 1))) * 4);
            atomicAdd(row + 0, fixed32(l1));
            if (up != 0.f) atomicAdd(row + 1, fixed32(up));
            if (down != 0.f) atomicAdd(row + 2, fixed32(down));
            if (kContour && cs != 0.f) atomicAdd(row + 3, fixed32(cs));
        }
    }
    // Last (nothing waits for these): the texture tiles under the covered pixels' bilinear footprints, one non-returning add per (wave, tile) -- the
    // lengths (an upper bound: a pixel with no texture gradient appends nothing) of the backward's record lists, which it packs by their prefix sums.
    // The same tiles, from the same footprint, as the pixel backward's rtile[] (mm_pixel_bwd.hip).
    if (any && a.trcnt) {
        const bool has = fpx0 >= 0;
        const int tcx0 = fpx0 / MM_UV_TILE, tcy0 = fpy0 / MM_UV_TILE;
        const int tcx1 = (fpx1 < a.Wt ? fpx1 : fpx0) / MM_UV_TILE, tcy1 = (fpy1 < a.Ht ? fpy1 : fpy0) / MM_UV_TILE;
        auto count_corner = [&](int rt) {
            unsigned long long pending = __ballot(rt >= 0);
            while (pending) {
                const int ld = __ffsll((unsigned long long)pending) - 1;
                const int tile = __builtin_amdgcn_readlane(rt, ld);
                const unsigned long long m = __ballot(rt == tile);
                if (t.lane == ld) atomicAdd(a.trcnt + (size_t)t.b * a.ntiles_tex + tile, __popcll(m));
                pending &= ~m;
            }
        };
        count_corner(has ? tcy0 * a.ntx_tex + tcx0 : -1);
        count_corner(has && tcx1 != tcx0 ? tcy0 * a.ntx_tex + tcx1 : -1);
        count_corner(has && tcy1 != tcy0 ? tcy1 * a.ntx_tex + tcx0 : -1);
        count_corner(has && tcx1 != tcx0 && tcy1 != tcy0 ? tcy1 * a.ntx_tex + tcx1 : -1);
    }
}

// Tiles no face can touch (three quarters of all tiles at 128x128; the plan kernel sorts them behind the others), FOUR per wave: a
// tile of its own wave pays the wave's fixed costs (launch, two dependent trips to memory, the store drain) for ~40 instructions of
// work.  Here the four tiles' loads are in flight together.  Per pixel exactly what shade_store's uncovered-tile path computes
// (m = 0, n = 0, soft-mask state "nothing taken"); the four tiles' recon_data terms go to ltot as one exact integer add per sum.
template <bool kNoMask, bool kContour>
__device__ inline void shade_empty_tiles(const RasterArgs& a, int b, int e0, int ne, int lane) {
    const int nslot = 4 * a.blocks_per_image;
    const size_t hw = (size_t)a.H * a.W;
    unsigned sl[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) sl[q] = a.order[(size_t)b * nslot + min(e0 + q, nslot - 1)] & 0x7FFFu;
#ifdef MM_BOUND_NO_EMPTY                                        // BOUND EXPERIMENT (WRONG results, never in the product; profiles/r06_empty_shading_bound.md): of an empty tile's
    for (int q = 0; q < 4; ++q) {                                // 60 bytes per pixel only the 12 the backward cannot do without (face_idx = -1, soft-mask state) are written
        const int blk = (int)sl[q] >> 2, quad = (int)sl[q] & 3;
        const int px = (blk % a.blocks_x) * MM_BLOCK_PX + (quad & 1) * MM_TILE + (lane & 7);
        const int py = (blk / a.blocks_x) * MM_BLOCK_PX + (quad >> 1) * MM_TILE + (lane >> 3);
        if (q < ne && px < a.W && py < a.H) {
            const size_t pix = (size_t)b * hw + (size_t)py * a.W + px;
            a.face_idx[pix] = -1;
            a.soft[pix] = make_float2(1.f, __int_as_float(0x7FFFFFFF));
        }
    }
    return;
#endif
    const float coef = MM_SH_C0 * a.lights[b * 9] + (0.f - MM_SH_C6B) * a.lights[b * 9 + 6];   // (bands 0 and 6: the same lights whatever the band order)
    bool in[4];
    size_t pin[4];
    float bgv[4][3], gtv[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int blk = (int)sl[q] >> 2, quad = (int)sl[q] & 3;
        const int px = (blk % a.blocks_x) * MM_BLOCK_PX + (quad & 1) * MM_TILE + (lane & 7);
        const int py = (blk / a.blocks_x) * MM_BLOCK_PX + (quad >> 1) * MM_TILE + (lane >> 3);
        in[q] = q < ne && px < a.W && py < a.H;
        pin[q] = (size_t)min(py, a.H - 1) * a.W + min(px, a.W - 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) bgv[q][c] = kNoMask ? a.bg[((size_t)b * 3 + c) * hw + pin[q]] : 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) gtv[q][c] = a.gt ? a.gt[((size_t)b * 4 + c) * hw + pin[q]] : 0.f;
    }
    float l1 = 0.f, down = 0.f, cs = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (kContour && a.gt) cs += contour_term(0.f, gtv[q][3], in[q], lane);   // (alpha = 0 everywhere here: the ground truth's contour alone)
        float out[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float val = kNoMask ? bgv[q][c] * coef : 1.f;
            out[c] = val < 0.f ? 0.f : (val > 1.f ? 1.f : val);
        }
        if (!in[q]) continue;
        const size_t pix = (size_t)b * hw + pin[q];
        *(float4*)(a.rgba + pix * 4) = make_float4(out[0], out[1], out[2], 1.f - 1.f);
        a.face_idx[pix] = -1;
        a.soft[pix] = make_float2(1.f, __int_as_float(0x7FFFFFFF));
        if (a.imnormal) { a.imnormal[pix * 3] = 0.f; a.imnormal[pix * 3 + 1] = 0.f; a.imnormal[pix * 3 + 2] = 0.f; }
        if (a.gt) {
            const float gm = gtv[q][3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float gi = gtv[q][c] * gm + 1.f * (1.f - gm);
                const float pi = out[c] * gm + 1.f * (1.f - gm);
                l1 += fabsf(pi - gi);
            }
            down += (0.f + gm) - 0.f * gm;                       // alpha = 0: up = 0, down = gm
        }
    }
    if (a.gt) {
        l1 = wave_sum(l1); down = wave_sum(down);
        if (kContour) cs = wave_sum(cs);
        if (lane == 0) {
            unsigned long long* row = (unsigned long long*)(a.ltot + ((size_t)b * MM_LSUB + (sl[0] & (MM_LSUB - 1))) * 4);
            if (l1 != 0.f) atomicAdd(row + 0, fixed32(l1));
            if (down != 0.f) atomicAdd(row + 2, fixed32(down));
            if (kContour && cs != 0.f) atomicAdd(row + 3, fixed32(cs));
        }
    }
}

// launch plumbing shared by the kernels' translation units
RasterArgs make_raster_args(const MMRenderDesc* d, const Workspace& w);
// plan kernel (mm_raster.hip), one workgroup per image between the vertex stage and the walk: (a) heavy-first tile order for the
// walk kernels, (b) the sweep items (face, box chunk) of the backward.  Returns the order buffer to put in
// RasterArgs::order, or nullptr where the sort does not pay (then the natural order is used).
const unsigned short* launch_order(RasterArgs& a, unsigned short* order, int* nheavy, int* bincount, int B, void** prof_events, hipStream_t s);
// The walk kernels come in two workgroup shapes with identical results: 256 threads (four tiles per workgroup, heavy tiles walked by
// the four waves together) and 64 threads (one tile per workgroup).  Measured (profiles/r02_walk_shapes.md), raster_fwd us, block /
// wave: 128x128 1280 faces 45.6 / 55.3; 128x64 38.4 / 51.5; 512x512 13 776 faces (no tile sort: a workgroup = the 2x2 tiles of a
// block) 546 / 613; 256x256 1280 faces 147 / 134.  The first wins where single tiles are heavy enough to be the kernel's tail
// (8-pixel bins: the mesh folds into few tiles) or where the four tiles are neighbours (unsorted); the second where tiles are many,
// even and taken in sorted order.  MM_OPT_WALK_BLOCK / MM_OPT_WALK_WAVE force one (tuning / tests).
#ifndef MM_WAVE_SHAPE_MIN_TILES
#define MM_WAVE_SHAPE_MIN_TILES 49152      // tiles per launch (B x tiles per image) from which 8-pixel-bin shapes take the one-tile-per-workgroup walk: B = 192 at 128x128
#endif
inline bool walk_block_mode(const RasterArgs& a) {
    if (a.options & MM_OPT_WALK_BLOCK) return true;
    if (a.options & MM_OPT_WALK_WAVE) return false;
    if (a.order == nullptr) return true;                         // no tile sort (a screen beyond MM_ORDER_MAX_SLOTS tiles)
    if (a.bin_shift != 3) return false;
    if (a.options & MM_OPT_MANY_IN_FLIGHT) return false;            // the caller says the chip is shared by several calls: the large-batch shape (below)
    // 8-pixel bins: the 256-thread shape with its cooperative heavy tiles, whose point is the launch's TAIL -- unless the batch runs the chip in many
    // rounds, where the tail is a small share and one tile per workgroup packs better (r06, profiles/r06_large_batch_shapes.md: raster_fwd -5 ... -9 % at
    // B = 256 / 384 with 128x128 images for near, SURVEY-8(d) and far cameras alike; at B = 128 far cameras still lose 14 % without the cooperative walk)
    return (long long)a.B * 4 * a.blocks_per_image < MM_WAVE_SHAPE_MIN_TILES;
}
// Sorted launch order: workgroup i -> (image, rank j of the workgroup inside the image).  The dispatcher deals consecutive workgroups to
// the eight XCDs in turn (observed).  `i % B` keeps an image on one XCD (B a multiple of 8): its face records and bin masks are read
// through one L2 -- right for small screens (128x128: raster_fwd 37.2 us against 39.3).  But an image's weight varies several-fold with
// its camera distance, and with many tiles per image the launch then lasts as long as the XCD that drew the heaviest images while the
// others idle (13 776 faces at 512x512: half of all wave slots empty over the second half of the launch).  With RasterArgs::spread
// eight consecutive workgroups are eight consecutive ranks of the SAME image instead: every XCD takes an eighth of every image, heavy
// tiles first everywhere (512x512: 361 -> 261 us; 256x256: 105 -> 96).
// spread == 2 (block-sorted order, one tile per workgroup): 32 consecutive workgroups = 32 consecutive ranks of one image, dealt so that
// ranks 4k .. 4k+3 -- the four tiles of one block, which walk the same bin's candidates -- are the four workgroups of the 32 that land on
// the SAME XCD (i, i + 8, i + 16, i + 24): dispatched within a microsecond of each other, they share that XCD's L2.
__device__ inline void walk_image_rank(int i, int B, int spread, int& b, int& j) {
    if (spread == 2) { const int g = i >> 5; b = g % B; j = (g / B) * 32 + (i & 7) * 4 + ((i >> 3) & 3); }
    else if (spread) { const int g = i >> 3; b = g % B; j = (g / B) * 8 + (i & 7); }
    else { b = i % B; j = i / B; }
}
inline bool walk_queue_mode(const RasterArgs& a) { return walk_queue_mode(a.options, a.bin_shift); }
// blocks instead of tiles are sorted where a block's four tiles share a bin (bins of 16 pixels or more) and tiles are walked one per workgroup
inline bool walk_block_sort(const RasterArgs& a) {
#ifdef MM_NO_BLOCK_SORT
    return false;
#else
#ifdef MM_BLOCK_SORT_IN_BLOCK_SHAPE                             // (A/B: the bin's four tiles as the four waves of ONE workgroup -- one CU, one L1)
    return a.bin_shift >= 4;
#else
    return a.bin_shift >= 4 && !(a.options & MM_OPT_WALK_BLOCK);
#endif
#endif
}
inline int walk_spread(const RasterArgs& a) {
    if (a.order == nullptr || 4 * a.blocks_per_image < 1024) return 0;
    return a.block_sort && !walk_block_mode(a) ? 2 : 1;
}
inline unsigned walk_grid(const RasterArgs& a, bool block) {
    if (!a.order) return (unsigned)a.B * (unsigned)a.blocks_per_image * (block ? 1u : 4u);
    const unsigned per_image = block ? (unsigned)(MM_HEAVY_MAX + (4 * a.blocks_per_image + 3) / 4) : (unsigned)a.blocks_per_image * 4u;
    return (unsigned)a.B * ((per_image + 31u) & ~31u);           // (ranks beyond an image's last workgroup exit at once)
}

}  // namespace mm
