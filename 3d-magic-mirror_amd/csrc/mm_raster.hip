// mm_raster.hip -- pixel stage of the render path for gfx950 (forward and backward).
//
// Replaces, fused into one launch per direction, what the reference reaches through kaolin for every pixel
// (call sites /root/reference/networks.py:297-317; semantics SURVEY.md 8(a) rows a8-a11, gradients Appendix A):
//   packed_rasterize_forward  (K1)  nearest front-facing face per pixel, barycentric interpolation
//   dibr_soft_mask_forward    (K3)  1 - prod(1 - exp(-sigma d^2)) over the first <= knum nearby faces
//   texture_mapping / grid_sample, spherical_harmonic_lighting, composite, clamp, cat
// and their backward kernels (K2, K4, grid_sampler backward, SH backward).
//
// Design (not kaolin's pixel-major brute force over all faces):
//   * a wave owns an 8x8 pixel tile, one lane per pixel; a 256-thread workgroup is 2x2 such tiles (16x16 px).
//   * the bin kernel left, per screen bin, bit-per-face masks of the faces whose box may touch it (front faces for
//     colour, inflated boxes of all faces for the silhouette).  The wave loads its bin's mask words coalesced, turns the
//     set bits into an ORDERED candidate list with popcount + wave prefix sum (face order = bit order, which the soft
//     mask's "first knum faces" rule needs) and stages 64 candidates at a time in LDS (struct-of-arrays float4 rows).
//   * per batch, lane j tests candidate j's box against the tile's 8 pixel columns and 8 rows (separable closed-box
//     test, the same float comparisons as a per-pixel test) -> a 64-bit pixel mask per candidate; a 6-stage wave
//     butterfly transposes that 64x64 bit matrix when the per-pixel view is needed.
//   * the (pixel, candidate) pairs of the batch are then evaluated 64 at a time by whichever lane and combined with
//     exact, commutative LDS atomics: 64-bit max of (orderable z, ~face id) for colour -- argmax over (z, -index) is
//     exactly kaolin's "strict z > best in index order" -- and an integer sum of log2(1-p) for the silhouette.  A wave's
//     critical path is pairs/64 evaluations, not its busiest pixel, and results do not depend on evaluation order.
#include "mm_device.h"

namespace mm {

struct RasterArgs {
    int B, H, W, F, Ht, Wt, knum, blocks_x, blocks_per_image;
    int bin_shift, nbx, nby, words;
    float mult, eps, sigmainv, infl;        // infl = boxlen * multiplier
    const float4* geo;
    const uint64_t* binmask;                // soft candidates (inflated boxes, all faces)
    const uint64_t* binmask_hard;           // colour candidates (front faces)
    const float* face_uvs;
    const float* fn;                        // (B,F,3) unit normals
    const float* textures;
    const float* lights;
    const float* bg;
    float* softq;
    int* lastf;
    const float* gt; float4* lpart;          // fused recon_data partial sums (gt == nullptr: off)
    const unsigned short* order;            // (B, 4*blocks) tile slots, heavy first; nullptr: natural order
    // outputs
    float* rgba;
    int32_t* face_idx;
    float* imnormal;
};

#define MM_PAIR_ROUND 512

// per-wave LDS staging: 64 candidates as three float4 rows + the id list of one mask group
struct __attribute__((aligned(16))) WaveStage {
    float4 p0[64];      // ax, ay, bx, by   (multiplier units)
    float4 p1[64];      // cx, cy, az, bz
    float4 p2[64];      // cz, unit normal z, face id (bits), 0
    unsigned short ids[MM_GROUP_WORDS * 64];
    unsigned short pairs[MM_PAIR_ROUND];    // (candidate << 8) | pixel, or (owner lane << 8) | candidate
    unsigned long long key[64];             // per pixel: best (orderable z << 32 | ~face) so far; 0 = none
    long long logsum[64];                   // per pixel: sum of log2(1-p) in 2^-32 fixed point (integer adds commute)
    int zeros[64];                          // per pixel: number of factors (1-p) that are exactly 0
};

struct TileCtx {
    int b, blk, px, py, tx0, ty0, lane, wave;
    bool in_img;
    float x0, y0;
    float xs[MM_TILE], ys[MM_TILE];         // pixel-centre columns / rows of the tile (same in every lane)
    const uint64_t* mask;                   // this wave's bin row (soft candidates): `words` 64-bit words
    const uint64_t* mask_hard;              // same bin, colour candidates
};

__device__ inline TileCtx make_tile(const RasterArgs& a) {
    TileCtx t;
    int blk;
    // one wave per workgroup (a slow tile then never pins the LDS of finished neighbours).  Launch order: the tiles with
    // the most candidates first (order_kernel), images interleaved -- the kernel's duration is set by its slowest waves, so
    // they must not start last.  Workgroup i -> image i % B (XCD i % 8 = image % 8 when 8 | B), rank i / B.
    if (a.order) {
        t.b = blockIdx.x % a.B;
        const int slot = a.order[(size_t)t.b * 4 * a.blocks_per_image + blockIdx.x / a.B];
        blk = slot >> 2; t.wave = slot & 3;
    } else {
        map_block(blockIdx.x >> 2, a.B, a.blocks_per_image, t.b, blk);
        t.wave = blockIdx.x & 3;
    }
    t.blk = blk;
    t.lane = threadIdx.x & 63;
    const int bx = blk % a.blocks_x, by = blk / a.blocks_x;
    const int tx0 = bx * MM_BLOCK_PX + (t.wave & 1) * MM_TILE, ty0 = by * MM_BLOCK_PX + (t.wave >> 1) * MM_TILE;
    t.tx0 = tx0; t.ty0 = ty0;
    t.px = tx0 + (t.lane & 7); t.py = ty0 + (t.lane >> 3);
    t.in_img = t.px < a.W && t.py < a.H;
    t.x0 = pixel_x(t.px, a.W, a.mult); t.y0 = pixel_y(t.py, a.H, a.mult);
#pragma unroll
    for (int i = 0; i < MM_TILE; ++i) { t.xs[i] = pixel_x(tx0 + i, a.W, a.mult); t.ys[i] = pixel_y(ty0 + i, a.H, a.mult); }
    // tiles never straddle bins (bin edge is 8, 16 or 32); a tile fully outside the image borrows the last bin
    const int binx = min(tx0 >> a.bin_shift, a.nbx - 1), biny = min(ty0 >> a.bin_shift, a.nby - 1);
    const size_t mrow = ((size_t)t.b * a.nbx * a.nby + (size_t)biny * a.nbx + binx) * a.words;
    t.mask = a.binmask + mrow; t.mask_hard = a.binmask_hard + mrow;
    return t;
}

__device__ inline int wave_prefix_excl(int v, int lane, int& total) {
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int n = __shfl_up(inc, o, 64);
        if (lane >= o) inc += n;
    }
    total = __shfl(inc, 63, 64);
    return inc - v;
}

__device__ inline void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 64x64 bit-matrix transpose across the wave: lane i holds row i on entry and column i on exit (6 butterfly stages).
template <int S>
__device__ inline uint64_t transpose_stage(uint64_t x, int lane) {
    // m: bit positions whose index has bit S clear
    constexpr uint64_t m = S == 32 ? 0x00000000FFFFFFFFull : S == 16 ? 0x0000FFFF0000FFFFull : S == 8 ? 0x00FF00FF00FF00FFull
                         : S == 4 ? 0x0F0F0F0F0F0F0F0Full : S == 2 ? 0x3333333333333333ull : 0x5555555555555555ull;
    const unsigned lo = __shfl_xor((unsigned)x, S, 64), hi = __shfl_xor((unsigned)(x >> 32), S, 64);
    const uint64_t y = ((uint64_t)hi << 32) | lo;
    return (lane & S) ? (((y >> S) & m) | (x & ~m)) : ((x & m) | ((y & m) << S));
}

__device__ inline uint64_t wave_transpose64(uint64_t x, int lane) {
    x = transpose_stage<32>(x, lane);
    x = transpose_stage<16>(x, lane);
    x = transpose_stage<8>(x, lane);
    x = transpose_stage<4>(x, lane);
    x = transpose_stage<2>(x, lane);
    x = transpose_stage<1>(x, lane);
    return x;
}

// box-vs-tile for ONE candidate (this lane's): bit (r*8+c) set iff pixel (row r, column c) of the tile passes the
// separable closed-box test  !(x < lo || x > hi)  -- the same comparisons on the same floats as the per-pixel test.
__device__ inline uint64_t box_pixels(const TileCtx& t, float xlo, float ylo, float xhi, float yhi) {
    unsigned col = 0, row = 0;
#pragma unroll
    for (int i = 0; i < MM_TILE; ++i) {
        col |= (unsigned)(!(t.xs[i] < xlo || t.xs[i] > xhi)) << i;
        row |= (unsigned)(!(t.ys[i] < ylo || t.ys[i] > yhi)) << i;
    }
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        lo |= ((row >> r) & 1u) ? (col << (8 * r)) : 0u;
        hi |= ((row >> (r + 4)) & 1u) ? (col << (8 * r)) : 0u;
    }
    return ((uint64_t)hi << 32) | lo;
}

// Walk the bin's candidates in face order, 64 at a time.  For every batch the candidates are staged in st->p0/p1/p2[0..n)
// and body(n, m) receives the hit masks of the batch -- kHard: front faces whose box contains the pixel; else: all faces whose
// inflated box contains it -- either candidate-major (lane j = candidate j, bit p = pixel p) or, with kTranspose,
// pixel-major (this lane's pixel, bit j = candidate j).  body returns false to stop (wave-uniform).
template <bool kHard, bool kTranspose, class Body>
__device__ inline void for_each_batch(const RasterArgs& a, const TileCtx& t, WaveStage* st, Body&& body) {
    const float4* geo = a.geo + (size_t)t.b * a.F * 3;
    const uint64_t* mask = kHard ? t.mask_hard : t.mask;
    const float pad = kHard ? 0.f : a.infl;
    for (int wbase = 0; wbase < a.words; wbase += MM_GROUP_WORDS) {
        uint64_t w = 0;
        if (t.lane < MM_GROUP_WORDS && wbase + t.lane < a.words) w = mask[wbase + t.lane];
        int total;
        int pos = wave_prefix_excl(__popcll(w), t.lane, total);
        if (total == 0) continue;
        while (w) {                                              // <= MM_GROUP_WORDS lanes, <= 64 iterations
            const int bit = __ffsll((unsigned long long)w) - 1;
            w &= w - 1;
            st->ids[pos++] = (unsigned short)(t.lane * 64 + bit);
        }
        wave_lds_sync();
        for (int k0 = 0; k0 < total; k0 += 64) {
            const int n = min(64, total - k0);
            uint64_t mc = 0;                                     // candidate-major: lane j = candidate j, bit p = pixel p
            if (t.lane < n) {
                const int f = wbase * 64 + st->ids[k0 + t.lane];
                const float4 g0 = geo[(size_t)f * 3 + 0], g1 = geo[(size_t)f * 3 + 1], g2 = geo[(size_t)f * 3 + 2];
                st->p0[t.lane] = g0; st->p1[t.lane] = g1;
                st->p2[t.lane] = make_float4(g2.x, g2.y, __int_as_float(f), 0.f);
                const float xmin = fminf(fminf(g0.x, g0.z), g1.x), ymin = fminf(fminf(g0.y, g0.w), g1.y);
                const float xmax = fmaxf(fmaxf(g0.x, g0.z), g1.x), ymax = fmaxf(fmaxf(g0.y, g0.w), g1.y);
                if (!kHard || g2.y >= 0.f) mc = box_pixels(t, xmin - pad, ymin - pad, xmax + pad, ymax + pad);
            }
            const uint64_t m = kTranspose ? wave_transpose64(mc, t.lane) : mc;
            wave_lds_sync();
            const bool go = body(n, m);
            wave_lds_sync();
            if (!go) return;
        }
    }
}

// Balanced evaluation of a batch's (row, column) pairs: every lane owns one ROW of the bit matrix `m` (a candidate, or a
// pixel) and the set bits are its columns.  The pairs of all lanes are laid out row-major in LDS and evaluated 64 at a
// time by WHICHEVER lane, eval(row, col); results are combined by the caller through commutative, exact LDS atomics
// (64-bit max / integer add), so the wave's critical path is pairs/64 evaluations, not its busiest lane, and the outcome
// does not depend on evaluation order.
template <class Eval>
__device__ inline void pair_parallel(const TileCtx& t, WaveStage* st, uint64_t m, Eval&& eval) {
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
        for (int p = t.lane; p < lim; p += 128) {                // two independent pairs per trip: ILP for a lone wave
            const unsigned pr0 = st->pairs[p];
            const bool two = p + 64 < lim;
            const unsigned pr1 = two ? st->pairs[p + 64] : pr0;
            eval((int)(pr0 >> 8), (int)(pr0 & 255u), true);
            eval((int)(pr1 >> 8), (int)(pr1 & 255u), two);
        }
        wave_lds_sync();
    }
}

__device__ inline unsigned long long depth_key(float z, int f) {
    const unsigned bits = __float_as_uint(z + 0.f);              // -0 -> +0: equal depths must tie
    const unsigned ord = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
    return ((unsigned long long)ord << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)f);
}

struct Hit { int f; float w0, w1, w2; };

// K1: nearest front face per pixel.  kaolin walks faces in index order and keeps strict z > best, i.e. the winner is
// argmax over (z, -index); that maximum is taken here with a 64-bit LDS atomic max per (pixel, face) pair, which is
// exact and order-free.  NaN and -inf depths never win, as in the reference.
__device__ inline void raster_pixels(const RasterArgs& a, const TileCtx& t, WaveStage* st, Hit& h) {
    st->key[t.lane] = 0ull;
    wave_lds_sync();
    for_each_batch<true, false>(a, t, st, [&](int n, uint64_t mc) {
        pair_parallel(t, st, mc, [&](int j, int l, bool live) {  // candidate j of the batch, pixel l of the tile
            const float x0 = pixel_x(t.tx0 + (l & 7), a.W, a.mult), y0 = pixel_y(t.ty0 + (l >> 3), a.H, a.mult);
            const float4 p0 = st->p0[j], p1 = st->p1[j], p2 = st->p2[j];
            float w0, w1, w2, nrm;
            edge_weights(p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, x0, y0, a.eps, w0, w1, w2, nrm);
            // straight-line on purpose (two of these are interleaved per trip): the IEEE divisions the oracle takes
            w0 /= nrm; w1 /= nrm; w2 /= nrm;
            const float z0 = (w0 * p1.z + w1 * p1.w) + w2 * p2.x;
            if (live && !(w0 < 0.f || w1 < 0.f || w2 < 0.f) && z0 > -INFINITY)
                atomicMax(&st->key[l], depth_key(z0, __float_as_int(p2.z)));
        });
        return true;
    });
    wave_lds_sync();
    const unsigned long long k = st->key[t.lane];
    h.f = -1; h.w0 = h.w1 = h.w2 = 0.f;
    if (k != 0ull) {                                             // barycentrics of the winner (same expressions, same values)
        h.f = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
        const float4* geo = a.geo + ((size_t)t.b * a.F + h.f) * 3;
        const float4 p0 = geo[0], p1 = geo[1];
        float nrm;
        edge_weights(p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, t.x0, t.y0, a.eps, h.w0, h.w1, h.w2, nrm);
        h.w0 /= nrm; h.w1 /= nrm; h.w2 /= nrm;
    }
}

// closest of the three edge segments: squared distance (multiplier units) and type = edge*3 + region
__device__ inline float tri_dist2(float x0, float y0, const float4& p0, const float4& p1, int& ty) {
    int r;
    float d = seg_dist2(x0, y0, p0.x, p0.y, p0.z, p0.w, ty);
    const float d1 = seg_dist2(x0, y0, p0.z, p0.w, p1.x, p1.y, r); if (d1 < d) { d = d1; ty = 3 + r; }
    const float d2 = seg_dist2(x0, y0, p1.x, p1.y, p0.x, p0.y, r); if (d2 < d) { d = d2; ty = 6 + r; }
    return d;
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

// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
template <bool kNoMask>
__global__ __launch_bounds__(64) void raster_fwd_kernel(RasterArgs a) {
    __shared__ WaveStage s_stage;
    const TileCtx t = make_tile(a);
    WaveStage* st = &s_stage;

    Hit h;
    raster_pixels(a, t, st, h);

    // K3: soft silhouette for the lanes no front face covers.  prod(1-p) is order-free, so it is accumulated per pixel
    // as an integer sum of log2(1-p) in 2^-32 fixed point (exact, commutative LDS adds) plus a count of exact zeros.
    float qnz = 1.f;
    int zeros = 0, lastf = 0x7FFFFFFF;
    const bool open = t.in_img && h.f < 0;
    if (__ballot(open)) {
        int cnt = 0;
        const float s2 = a.mult * a.mult;
        st->logsum[t.lane] = 0ll; st->zeros[t.lane] = 0;
        wave_lds_sync();
        for_each_batch<false, true>(a, t, st, [&](int n, uint64_t sm) {
            sm = soft_take(sm, open, a.knum - cnt);              // pixel-major: the first knum hits of this pixel, in order
            cnt += __popcll(sm);
            if (sm != 0 && cnt >= a.knum) lastf = __float_as_int(st->p2[63 - __clzll((unsigned long long)sm)].z);   // knum-th face taken
            pair_parallel(t, st, sm, [&](int l, int j, bool live) {   // pixel l of the tile, candidate j of the batch
                const float x0 = pixel_x(t.tx0 + (l & 7), a.W, a.mult), y0 = pixel_y(t.ty0 + (l >> 3), a.H, a.mult);
                int ty;
                const float d = tri_dist2(x0, y0, st->p0[j], st->p1[j], ty);
                const float q = 1.f - expf(-((d / s2) * a.sigmainv));
                if (live) {
                    if (q == 0.f) atomicAdd(&st->zeros[l], 1);
                    else atomicAdd((unsigned long long*)&st->logsum[l], (unsigned long long)(long long)((double)log2f(q) * 4294967296.0));
                }
            });
            return __ballot(open && cnt < a.knum) != 0;          // every open lane already holds knum faces: stop
        });
        wave_lds_sync();
        zeros = st->zeros[t.lane];
        qnz = exp2f((float)((double)st->logsum[t.lane] * (1.0 / 4294967296.0)));
    }
    if (!t.in_img && !a.gt) return;

    // ---- shading (a9-a11).  Uncovered pixels carry zero features exactly like kaolin's interpolated_features.
    // (lanes outside a ragged image only stay for the fused loss reduction: they address a clamped pixel and store nothing)
    const int cpx = min(t.px, a.W - 1), cpy = min(t.py, a.H - 1);
    const size_t pix = ((size_t)t.b * a.H + cpy) * a.W + cpx;
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
    const size_t hw = (size_t)a.H * a.W, pin = (size_t)cpy * a.W + cpx;
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
    const float keepprod = zeros > 0 ? 0.f : qnz;
    out[3] = (h.f >= 0) ? 1.f : (1.f - keepprod);
    if (t.in_img) {
        *(float4*)(a.rgba + pix * 4) = make_float4(out[0], out[1], out[2], out[3]);
        a.face_idx[pix] = h.f;
        a.softq[pix] = (h.f >= 0 || zeros >= 2) ? 0.f : (zeros == 1 ? -qnz : qnz);
        if (h.f < 0) a.lastf[pix] = lastf;
        if (a.imnormal) { a.imnormal[pix * 3] = nx; a.imnormal[pix * 3 + 1] = ny; a.imnormal[pix * 3 + 2] = nz; }
    }
    if (a.gt) {                                                  // recon_data terms of this tile (networks.py:370-377)
        float l1 = 0.f, up = 0.f, down = 0.f;
        if (t.in_img) {
            const float* g = a.gt + (size_t)t.b * 4 * hw;
            const float gm = g[3 * hw + pin];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float gi = g[c * hw + pin] * gm + 1.f * (1.f - gm);
                const float pi = out[c] * gm + 1.f * (1.f - gm);
                l1 += fabsf(pi - gi);
            }
            up = out[3] * gm; down = (out[3] + gm) - up;
        }
        l1 = wave_sum(l1); up = wave_sum(up); down = wave_sum(down);
        if (t.lane == 0) a.lpart[((size_t)t.b * a.blocks_per_image + t.blk) * 4 + t.wave] = make_float4(l1, up, down, 0.f);
    }
}

// Ranks the tile slots (16x16 block * 4 + quadrant) of every image by their soft-mask candidate count, descending
// (rank by counting, all pairs, in LDS: slots per image <= 1024).  Only the launch ORDER of raster_fwd depends on it.
__global__ __launch_bounds__(256) void order_kernel(RasterArgs a, unsigned short* order) {
    __shared__ __attribute__((aligned(16))) int s_cnt[1024];
    const int b = blockIdx.x, nslot = 4 * a.blocks_per_image;          // a multiple of 4
    for (int slot = threadIdx.x; slot < nslot; slot += 256) {
        const int blk = slot >> 2, q = slot & 3;
        const int tx0 = (blk % a.blocks_x) * MM_BLOCK_PX + (q & 1) * MM_TILE, ty0 = (blk / a.blocks_x) * MM_BLOCK_PX + (q >> 1) * MM_TILE;
        int c = 0;
        if (tx0 < a.W && ty0 < a.H) {
            const uint64_t* row = a.binmask + ((size_t)b * a.nbx * a.nby + (size_t)(ty0 >> a.bin_shift) * a.nbx + (tx0 >> a.bin_shift)) * a.words;
            for (int w0 = 0; w0 < a.words; w0 += 8) {             // eight loads in flight per trip
                uint64_t r[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) r[k] = (w0 + k < a.words) ? row[w0 + k] : 0ull;
#pragma unroll
                for (int k = 0; k < 8; ++k) c += __popcll(r[k]);
            }
        }
        s_cnt[slot] = c;
    }
    __syncthreads();
    for (int slot = threadIdx.x; slot < nslot; slot += 256) {
        const int c = s_cnt[slot];
        int rank = 0;
        for (int j = 0; j < nslot; j += 16) {                      // four 16-byte LDS reads in flight per trip
            int4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = (j + 4 * k < nslot) ? *(const int4*)&s_cnt[j + 4 * k] : make_int4(-1, -1, -1, -1);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int jj = j + 4 * k;
                rank += (v[k].x > c) || (v[k].x == c && jj < slot);
                rank += (v[k].y > c) || (v[k].y == c && jj + 1 < slot);
                rank += (v[k].z > c) || (v[k].z == c && jj + 2 < slot);
                rank += (v[k].w > c) || (v[k].w == c && jj + 3 < slot);
            }
        }
        order[(size_t)b * nslot + rank] = (unsigned short)slot;
    }
}

static RasterArgs make_args(const MMRenderDesc* d, const Workspace& w) {
    RasterArgs a;
    a.B = d->B; a.H = d->H; a.W = d->W; a.F = d->F; a.Ht = d->Ht; a.Wt = d->Wt; a.knum = d->knum;
    a.blocks_x = (d->W + MM_BLOCK_PX - 1) / MM_BLOCK_PX;
    a.blocks_per_image = w.blocks_per_image;
    a.bin_shift = w.bin_shift; a.nbx = w.nbx; a.nby = w.nby; a.words = w.words;
    a.mult = d->multiplier; a.eps = d->eps; a.sigmainv = d->sigmainv; a.infl = d->boxlen * d->multiplier;
    a.geo = w.geo; a.binmask = w.binmask; a.binmask_hard = w.binmask_hard; a.softq = w.softq; a.lastf = w.lastf; a.gt = d->fused_gt; a.lpart = w.lpart;
    a.face_uvs = d->face_uvs; a.fn = d->face_normals; a.textures = d->textures; a.lights = d->lights; a.bg = d->bg;
    a.rgba = d->rgba; a.face_idx = d->face_idx; a.imnormal = d->imnormal;
    return a;
}

int launch_raster_fwd(const MMRenderDesc* d, const Workspace& w, hipStream_t s) {
    RasterArgs a = make_args(d, w);
    dim3 grid(a.blocks_per_image * d->B * 4);
    a.order = nullptr;
    if (4 * a.blocks_per_image <= 1024 && a.words <= 64) {       // heavy-first launch order (skipped where the sort would not pay)
        ProfScope po(d->prof_events, MM_PROF_ORDER, s);
        hipLaunchKernelGGL(order_kernel, dim3(d->B), dim3(256), 0, s, a, w.order);
        a.order = w.order;
    }
    ProfScope ps(d->prof_events, MM_PROF_RASTER_FWD, s);
    if (d->no_mask) hipLaunchKernelGGL(raster_fwd_kernel<true>, grid, dim3(64), 0, s, a);
    else hipLaunchKernelGGL(raster_fwd_kernel<false>, grid, dim3(64), 0, s, a);
    return launch_ok("raster_fwd");
}

}  // namespace mm
