// mm_backward.hip -- the GATHER half of the render path's backward for gfx950: no floating-point atomic anywhere, on HBM or in LDS.
//
// kaolin's backward kernels (rasterize_backward_cuda, dibr_soft_mask_backward_cuda) and torch's grid_sampler backward SCATTER per-pixel
// contributions with atomicAdd.  On MI355X an agent-scope float atomic is executed at the memory side of the fabric (the eight XCD L2s are
// not coherent with each other), ~17 G atomics/s measured, and this path would issue ~5 M of them per batch; an LDS float atomic costs
// ~81 ns of the CU's LDS unit per wave-instruction whatever the addresses (profiles/r02_lds_atomic_calibration.txt).  The backward is
// therefore two passes (SURVEY.md Appendix A for the math):
//
//   1. pixel_bwd (mm_pixel_bwd.hip, compiled like the forward: see mm_backward.h)   pixel-major, one lane per pixel: re-shades the pixel,
//      writes dL/dbg, reduces dL/dlights per workgroup, leaves for every covered pixel the nine K2 numbers of its face (gp, gp2), for every
//      uncovered one dL/dalpha, per image the maxima that fix the gather's fixed-point scales, and APPENDS the pixel's texture contribution
//      to the record list of every 32x32-texel tile under its bilinear footprint.  Its first workgroups plan the face sweep (sweep items).
//   2. gather_bwd (this file), one launch, two kinds of workgroup:
//      a. texture tiles   one workgroup per (image, tile): streams the tile's record list into INT32 fixed-point LDS accumulators (per-tile
//         power-of-two scale) and writes the tile once with plain stores: no zero-fill pass over grad_textures.
//      b. face sweep      8 lanes per sweep item (a 128-pixel chunk of a face's inflated screen box; an image's items are dealt to its waves
//         round-robin): pixels the face owns give K2 (add the pixel pass's numbers), uncovered pixels that hold the face among their first
//         knum silhouette faces give K4; hits are ballot-compacted over the wave and finished by all 64 lanes into INT64 fixed-point per-item
//         LDS sums; one plain store per item, added up per face in index order by the vertex backward.
//   Integer adds commute exactly: the whole backward is bitwise reproducible.
#include <cstdlib>
#include "mm_backward.h"

MM_TIMELINE_STORAGE(gather_bwd)
MM_PP_STORAGE(gather_face)      // 0 setup, 1 sweep (face_idx loads), 2 compaction, 3 item loads, 4 item arithmetic + LDS adds, 5 stores; counts: trips, items
MM_PP_STORAGE(gather_tex)       // 0 count + first record, 1 clear, 2 records, 3 tile store; counts: records

#ifndef MM_ITEM_UNROLL
#define MM_ITEM_UNROLL 1        // hit items per lane whose loads are in flight together.  2 (as up to r02i) hides a trip per 128 hits but costs ten VGPRs: at 64
                                // the kernel holds 8 waves per SIMD instead of 6, and occupancy is what this latency-bound launch lives on (gather_bwd us at
                                // configs 2 / 3 / 5: 33.3 / 91.3 / 303 with 2, 31.4 / 85.3 / 279 with 1; 2 at 7 waves per SIMD: 32.4 / 87.3 / 287)
#endif

namespace mm {


// ---------------------------------------------------------------------------------------------------------------------
// 2. gathers.  The face gather sweeps a face's screen box with MM_FL lanes, MM_SWEEP pixels per lane per trip with the loads of a trip
//    issued together (these loops are latency-bound: a dependent HBM/L2 load per step).
// ---------------------------------------------------------------------------------------------------------------------
#define MM_TS MM_UV_TILE
#ifndef MM_SWEEP
#define MM_SWEEP 16             // pixels per lane per trip
#endif
#ifndef MM_FL                  // (4 / 16 lanes and 8 / 4 pixels per lane and trip measured in r06, profiles/r06_sweep_shape_ab.md: 8 x 16 is the best shape at every
                               //  one-batch size; 4 x 16 wins 5 % of this kernel at B=384 only)
#define MM_FL 8                 // lanes per sweep item: a trip covers MM_FL * MM_SWEEP pixels of each of the wave's items
#endif
#define MM_FPW (64 / MM_FL)     // items per wave
#ifndef MM_FL4_MIN_B
#define MM_FL4_MIN_B 128         // batches from this size on sweep with FOUR lanes per item (see launch_raster_bwd)
#endif
static_assert(MM_CHUNK_PX % (MM_FL * MM_SWEEP) == 0, "a chunk is a whole number of trips");

// squared distance from p to segment u-v by clamped projection: t in [0,1] is where the nearest point lies, q = p - nearest.
// With t clamped, d(d^2)/du = -2 (1-t) q and d(d^2)/dv = -2 t q hold in all three regions of kaolin's case split (t = 0: the
// nearest point is u, t = 1: it is v).  Hardware reciprocal: the backward is held to 1e-4, not to the bit.
// (x, y) pairs as two-lane vectors: gfx950 executes v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 on both halves at once
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ inline float dot2(f2 a, f2 b) { const f2 m = a * b; return m.x + m.y; }
struct SegHit { float d2, t; f2 q; };
__device__ inline SegHit seg_nearest(f2 p, f2 u, f2 v) {
    const f2 e = v - u, r = p - u;
    const float len2 = dot2(e, e);
    float t = (len2 > 0.f) ? dot2(r, e) * __builtin_amdgcn_rcpf(len2) : 0.f;
    t = fminf(fmaxf(t, 0.f), 1.f);
    SegHit h;
    h.t = t; h.q = r - t * e;
    h.d2 = dot2(h.q, h.q);
    return h;
}

struct FaceBox { float4 p0, p1; float xmin, ymin, xmax, ymax; int px0, py0, bw, npx; float inv_bw, nz; };

__device__ inline FaceBox face_box(const BwdArgs& a, size_t o) {
    FaceBox fb;
    fb.p0 = a.geo[o * 3 + 0]; fb.p1 = a.geo[o * 3 + 1];
    const float4 g2 = a.geo[o * 3 + 2];                           // z, w: the inflated pixel box, packed by the vertex stage
    fb.xmin = fminf(fminf(fb.p0.x, fb.p0.z), fb.p1.x); fb.ymin = fminf(fminf(fb.p0.y, fb.p0.w), fb.p1.y);
    fb.xmax = fmaxf(fmaxf(fb.p0.x, fb.p0.z), fb.p1.x); fb.ymax = fmaxf(fmaxf(fb.p0.y, fb.p0.w), fb.p1.y);
    int taken = 1, bh;
    if (a.fflag) taken = a.fflag[o * 2 + 1];                      // (the same box the sweep plan cut into this face's items)
    sweep_box(__float_as_uint(g2.z), __float_as_uint(g2.w), taken != 0, a.sweep_sx, a.sweep_sy, a.W, a.H, fb.px0, fb.py0, fb.bw, bh);
    fb.npx = fb.bw * bh;
    fb.inv_bw = 1.f / (float)(fb.bw > 0 ? fb.bw : 1);
    fb.nz = g2.y;
    return fb;
}

// pixel #idx of a box swept row-major (idx < 2^22: the float quotient is exact enough to be fixed up by one compare)
__device__ inline void box_pixel(int idx, int px0, int py0, int bw, float inv_bw, int& px, int& py) {
    int yy = (int)(((float)idx + 0.5f) * inv_bw);
    if (yy * bw > idx) --yy;
    if ((yy + 1) * bw <= idx) ++yy;
    px = px0 + (idx - yy * bw); py = py0 + yy;
}

// What a wave keeps in LDS about the eight faces its 8-lane groups sweep, so that ANY lane can finish a compacted work item of
// any of them.  The per-face sums are 64-bit INTEGERS (fixed point at a per-image power-of-two scale): an LDS float atomic costs
// ~81 ns of the CU's LDS unit per wave-instruction whatever the addresses (ds_add_f32 takes the 64 lanes one after the other:
// measured, profiles/r02_lds_atomic_calibration.txt), an integer one 2-7 ns, and nine float ones per item round had made this
// kernel LDS-bound.  Integer adds also commute exactly: the face gradients are bitwise reproducible.
struct __attribute__((aligned(16))) FaceSlot {
    float4 p0, p1;                   // ax,ay,bx,by | cx,cy,az,bz  (multiplier units)
    float box[4];                    // xmin, ymin, xmax, ymax
    int px0, py0, bw, f;
    float inv_bw;
    int lo;                          // first box pixel of this sweep (0, or the chunk's start)
    long long acc[9];                // dL/d(ax,ay,bx,by,cx,cy), dL/d(n), fixed point
};

template <int FL>
struct __attribute__((aligned(16))) SweepStageT {
    FaceSlot slot[64 / FL];
    unsigned short items[MM_SWEEP * 64];   // owned << 15 | sweep slot << 6 | lane
};

// ballot-compaction of the lanes' hits into an ordered LDS item list; returns the item count (wave-uniform)
__device__ inline int compact_hits(const bool (&own)[MM_SWEEP], const bool (&opn)[MM_SWEEP], int lane, unsigned short* items) {
    int base = 0;
#pragma unroll
    for (int i = 0; i < MM_SWEEP; ++i) {
        const bool hit = own[i] || opn[i];
        const unsigned long long m = __ballot(hit);
        if (hit) items[base + ballot_rank(m)] = (unsigned short)((own[i] ? 0x8000 : 0) | (i << 6) | lane);
        base += __popcll(m);
    }
    return base;
}


// Fixed-point scale of an image's face sums.  The pixel pass left max |K2 number| and max |dL/dalpha| of the image (gmax); a K4
// contribution is bounded by |dL/dalpha| * mult * sqrt(2 sigma' / e), sigma' = sigmainv / mult^2 (the maximum of d exp(-sigma' d^2)
// times the constant factors of Appendix A.2).  A sum belongs to ONE sweep item and takes at most one contribution per pixel of the
// item's chunk (chunk_px = MM_CHUNK_PX << k pixels in this image, nitems[b].y): the largest contribution is placed at 2^(62 - L),
// L = ceil(log2 chunk_px), so that 2^L of them fit a 63-bit sum -- 2^55 for the usual 128-pixel chunk, i.e. the unit is 2^-55 of the
// bound (rounds 2-5 placed it at 2^40 whatever the chunk, room for a 2048 x 2048 chunk nobody cuts: contributions below 2^-41 of the
// bound were rounded to zero, which randomised cases with saturated silhouettes kept finding -- r06 seed 8809 case 254).
__device__ inline float face_sum_scale(const BwdArgs& a, int b, int chunk_px, float& inv) {
    float m2 = 0.f, m4 = 0.f;
#pragma unroll
    for (int sh = 0; sh < MM_GSHARD; ++sh) {
        m2 = fmaxf(m2, __uint_as_float(a.gmax[((size_t)b * MM_GSHARD + sh) * 8]));
        m4 = fmaxf(m4, __uint_as_float(a.gmax[((size_t)b * MM_GSHARD + sh) * 8 + 1]));
    }
    const float sig = a.sigmainv / (a.mult * a.mult);
    const float M = fmaxf(m2, m4 * a.mult * sqrtf(2.f * sig * 0.36787944f) * 1.0001f);
    if (!(M > 0.f) || !(M < INFINITY)) { inv = 0.f; return 0.f; }
    int e;
    (void)frexpf(M, &e);                                         // M < 2^e
    const int L = 32 - __clz(max(chunk_px, 2) - 1);              // ceil(log2 chunk_px) in [1, 31]: a chunk is at most MM_CHUNK_PX << 20 pixels (plan kernel)
    const int k = min(max(62 - L - e, -80), 126);
    inv = ldexpf(1.f, -k);
    return ldexpf(1.f, k);
}
__device__ inline void fixed_add(long long* p, float v, float scale) { atomicAdd((unsigned long long*)p, (unsigned long long)__float2ll_rn(v * scale)); }

// The tile's accumulators are INTEGERS (fixed point at a per-tile power-of-two scale): an LDS float atomic costs ~81 ns of the
// CU's LDS unit per wave-instruction whatever the addresses, an integer one 2-7 ns (profiles/r02_lds_atomic_calibration.txt), and
// twelve of them per record made this half of the gather kernel LDS-bound.  Integer adds also commute exactly, so the texture
// gradient is bitwise reproducible.  Scale: the largest |contribution| of the tile's records times their number bounds every
// texel's sum; it is placed just below 2^30.  Quantisation: half a unit = (records * max) * 2^-31 per add.
__device__ inline void tex_accumulate(const BwdArgs& a, int (*s_acc)[MM_TS * MM_TS], const TexRecord& rc, int tx0, int ty0, float scale) {
    const int x0 = (int)(rc.xy & 0xFFFFu), y0 = (int)(rc.xy >> 16);
    const int lx0 = x0 - tx0, lx1 = lx0 + 1, ly0 = y0 - ty0, ly1 = ly0 + 1;
    const bool cx0 = lx0 >= 0 && lx0 < MM_TS, cx1 = lx1 >= 0 && lx1 < MM_TS && x0 + 1 < a.Wt;
    const bool cy0 = ly0 >= 0 && ly0 < MM_TS, cy1 = ly1 >= 0 && ly1 < MM_TS && y0 + 1 < a.Ht;
    const float ex = 1.f - rc.tx, ey = 1.f - rc.ty;
    const float wnw = ex * ey, wne = rc.tx * ey, wsw = ex * rc.ty, wse = rc.tx * rc.ty;
    const float dt[3] = {rc.d0 * scale, rc.d1 * scale, rc.d2 * scale};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (dt[c] != 0.f) {
            if (cx0 && cy0) atomicAdd(&s_acc[c][ly0 * MM_TS + lx0], __float2int_rn(dt[c] * wnw));
            if (cx1 && cy0) atomicAdd(&s_acc[c][ly0 * MM_TS + lx1], __float2int_rn(dt[c] * wne));
            if (cx0 && cy1) atomicAdd(&s_acc[c][ly1 * MM_TS + lx0], __float2int_rn(dt[c] * wsw));
            if (cx1 && cy1) atomicAdd(&s_acc[c][ly1 * MM_TS + lx1], __float2int_rn(dt[c] * wse));
        }
    }
}

__device__ inline float tex_record_max(const TexRecord& rc) { return fmaxf(fmaxf(fabsf(rc.d0), fabsf(rc.d1)), fabsf(rc.d2)); }

// 2a. texture gradient: one workgroup per (image, 32x32-texel tile) streams the records the pixel pass appended for the
//     tile into LDS accumulators and writes every texel of the tile once.
__device__ inline void texture_gather_block(const BwdArgs& a, int block, int (*s_acc)[MM_TS * MM_TS]) {
    __shared__ float s_max[4];
    const int ntiles = a.ntx * a.nty;
    int b, T;
    map_block(block, a.B, ntiles, b, T);
    const int tid = threadIdx.x;
    const int tx0 = (T % a.ntx) * MM_TS, ty0 = (T / a.ntx) * MM_TS;
    MM_PP_BEGIN();
    const int nall = a.tcur[(size_t)b * ntiles + T], off = a.toff[(size_t)b * ntiles + T] - 1;
    const int dropped = a.tdrop[b];                            // records of the image its array had no room for (pixel_bwd)
    const int nrec = max(0, min(nall, a.trcap - off));           // (the list is cut where the array ends)
    const TexRecord* recs = a.trec + (size_t)b * a.trcap + off;
    MM_PP_MARK(0);
    // largest contribution of the tile's records (first pass; the second one below re-reads them from L2)
    float mx = 0.f;
    for (int r = tid; r < nrec; r += 256) mx = fmaxf(mx, tex_record_max(recs[r]));
    mx = wave_max(mx);
    if ((tid & 63) == 0) s_max[tid >> 6] = mx;
    for (int i = tid; i < 3 * MM_TS * MM_TS / 4; i += 256) ((int4*)&s_acc[0][0])[i] = make_int4(0, 0, 0, 0);   // (16-byte LDS stores)
    __syncthreads();
    mx = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
    MM_PP_MARK(1);
    float inv = 0.f;
    if (mx > 0.f && mx < INFINITY) {                             // workgroup-uniform; nothing to add up otherwise (about half of all tiles)
        int e;
        (void)frexpf((float)nrec * mx, &e);                      // records * max < 2^e
        const int k2 = min(max(30 - e, -100), 120);
        const float scale = ldexpf(1.f, k2);
        inv = ldexpf(1.f, -k2);
        for (int r = tid; r < nrec; r += 256) tex_accumulate(a, s_acc, recs[r], tx0, ty0, scale);
        __syncthreads();
    }
    if (dropped != 0) {                                          // the image lost records: its texture gradient is NOT a gradient -- say so in every texel
        inv = __builtin_nanf("");
        if (T == 0 && tid == 0) {
            a.tstatus[b] = dropped;
            if (a.status_flag) __hip_atomic_fetch_add(a.status_flag, dropped, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (the host may be polling it)
        }
    }
    MM_PP_MARK(2);
    MM_PP_COUNT(nrec, 0);
    // write the tile once (also where nothing landed: no separate zero-fill of grad_textures): four texels of a row per thread and store
    if ((a.Wt & 3) == 0 && (((size_t)a.grad_textures) & 15) == 0) {
        for (int i = tid; i < 3 * MM_TS * MM_TS / 4; i += 256) {
            const int c = i / (MM_TS * MM_TS / 4), r = i - c * (MM_TS * MM_TS / 4);
            const int ly = r / (MM_TS / 4), lx = (r - ly * (MM_TS / 4)) * 4;
            const int x = tx0 + lx, y = ty0 + ly;
            if (x < a.Wt && y < a.Ht) {
                const int4 v = *(const int4*)&s_acc[c][ly * MM_TS + lx];
                *(float4*)(a.grad_textures + (((size_t)b * 3 + c) * a.Ht + y) * a.Wt + x) =
                    make_float4((float)v.x * inv, (float)v.y * inv, (float)v.z * inv, (float)v.w * inv);
            }
        }
    } else {
        for (int i = tid; i < 3 * MM_TS * MM_TS; i += 256) {
            const int c = i / (MM_TS * MM_TS), r = i - c * (MM_TS * MM_TS);
            const int ly = r / MM_TS, lx = r - ly * MM_TS;
            const int x = tx0 + lx, y = ty0 + ly;
            if (x < a.Wt && y < a.Ht) a.grad_textures[(((size_t)b * 3 + c) * a.Ht + y) * a.Wt + x] = (float)s_acc[c][r] * inv;
        }
    }
    MM_PP_MARK(3);
    MM_PP_FLUSH(gather_tex, (long long)block * 4 + (threadIdx.x >> 6));
}

struct ItemLoad { float4 q0, q1; float q2, sq; int lf, px, py, g; bool owned, live; };

// one compacted hit of the sweep, its loads done: owned pixel -> add the pixel pass's nine K2 numbers; open pixel -> K4 (Appendix A.2)
__device__ inline void item_finish(const BwdArgs& a, FaceSlot& fs, const ItemLoad& ld, float s2, float scale) {
    if (ld.owned) {
        const float4 q0 = ld.q0, q1 = ld.q1;
        fixed_add(&fs.acc[0], q0.x, scale); fixed_add(&fs.acc[1], q0.y, scale); fixed_add(&fs.acc[2], q0.z, scale);
        fixed_add(&fs.acc[3], q0.w, scale); fixed_add(&fs.acc[4], q1.x, scale); fixed_add(&fs.acc[5], q1.y, scale);
        fixed_add(&fs.acc[6], q1.z, scale); fixed_add(&fs.acc[7], q1.w, scale); fixed_add(&fs.acc[8], ld.q2, scale);
        return;
    }
    const float ga = ld.q2, sq = ld.sq;                           // uncovered pixels: the pixel pass left dL/dalpha here
    const float x0 = pixel_x_k(ld.px, a.W, a.kx), y0 = pixel_y_k(ld.py, a.H, a.ky);      // (the forward's centres, bit for bit; no division per item)
    const int bm = box_mode(a.options);
    const bool inbox = bm ? !(box_reject(x0, fs.box[0] - a.infl, fs.box[2] + a.infl, bm) || box_reject(y0, fs.box[1] - a.infl, fs.box[3] + a.infl, bm))
                          : !(x0 < fs.box[0] - a.infl || x0 > fs.box[2] + a.infl || y0 < fs.box[1] - a.infl || y0 > fs.box[3] + a.infl);
    if (sq != 0.f && (MM_K4_KEEP_ONES || sq != 1.f) && ga != 0.f && fs.f <= ld.lf && inbox) {
        const float4 p0 = fs.p0, p1 = fs.p1;
        const f2 pp = {x0, y0}, ca = {p0.x, p0.y}, cb = {p0.z, p0.w}, cc = {p1.x, p1.y};
        SegHit h = seg_nearest(pp, ca, cb);               // edge 0: corner a -> b
        int e = 0;
        const SegHit h1 = seg_nearest(pp, cb, cc);        // edge 1: b -> c
        if (h1.d2 < h.d2) { h = h1; e = 1; }
        const SegHit h2 = seg_nearest(pp, cc, ca);        // edge 2: c -> a
        if (h2.d2 < h.d2) { h = h2; e = 2; }
        const float p = __builtin_amdgcn_exp2f(-(h.d2 * (a.sigmainv / s2)) * 1.4426950408889634f);
        // the factor exactly as the forward folded it into the stored product (bit for bit): dividing by anything else is a large error
        // where the pixel centre lies almost on an edge (q of a few ulps)
        const float q = soft_factor(x0, y0, p0, p1, a.sig2);
        const float qnz = fabsf(sq);
        const bool onezero = sq < 0.f;
        const float excl = (q != 0.f) ? (onezero ? 0.f : qnz * __builtin_amdgcn_rcpf(q)) : (onezero ? qnz : 0.f);
        const float gd = ga * excl * (-(p * a.sigmainv) / s2) * a.mult;
        if (gd != 0.f) {
            // edge e runs from corner e to corner (e+1)%3
            const int iu = e * 2, iv = (e == 2 ? 0 : e + 1) * 2;
            const float cu = -2.f * (1.f - h.t) * gd, cv = -2.f * h.t * gd;
            const f2 gu = cu * h.q, gv = cv * h.q;
            fixed_add(&fs.acc[iu], gu.x, scale); fixed_add(&fs.acc[iu + 1], gu.y, scale);
            fixed_add(&fs.acc[iv], gv.x, scale); fixed_add(&fs.acc[iv + 1], gv.y, scale);
        }
    }
}

// 2b. per-face gradients: MM_FL lanes per (image, face) sweep the face's inflated box; pixels it owns give the K2 barycentric
//     gradient, uncovered pixels that hold it among their first knum soft-mask faces give K4.  The hits of a trip are
//     ballot-compacted over the whole wave and finished by all 64 lanes (one round of loads per trip) into the per-face
//     fixed-point LDS sums.  This lane's face is `f` of image `b`, and its group sweeps box pixels [lo, hi) of it.
template <int FL>
__device__ inline void face_sweep(const BwdArgs& a, SweepStageT<FL>* st, int b, int f, int lane, const FaceBox& fb, int lo, int hi, float scale MM_PP_ARG) {
    const int grp = lane / FL, sl = lane % FL;
    const size_t hw = (size_t)a.H * a.W;
    const float s2 = a.mult * a.mult;
    if (sl == 0) {
        FaceSlot& fs = st->slot[grp];
        fs.p0 = fb.p0; fs.p1 = fb.p1; fs.box[0] = fb.xmin; fs.box[1] = fb.ymin; fs.box[2] = fb.xmax; fs.box[3] = fb.ymax;
        fs.px0 = fb.px0; fs.py0 = fb.py0; fs.bw = fb.bw; fs.inv_bw = fb.inv_bw; fs.lo = lo; fs.f = f;
    }
    for (int k = sl; k < 9; k += FL) st->slot[grp].acc[k] = 0ll;
    int nmax = hi - lo;
    if (FL <= 4) nmax = max(nmax, (int)lane_xchg<4>((unsigned)nmax, lane));
    if (FL <= 8) nmax = max(nmax, (int)lane_xchg<8>((unsigned)nmax, lane));
    nmax = max(nmax, (int)lane_xchg<16>((unsigned)nmax, lane)); nmax = max(nmax, (int)lane_xchg<32>((unsigned)nmax, lane));
    static_assert(FL == 16 || FL == 8 || FL == 4, "the exchange strides above start at the lanes-per-item count");
    wave_sync_lds();
    MM_PP_MARK(0);

    // A lane's pixels of a trip are MM_FL apart in the row-major box: the first one by division, the others by stepping (column += MM_FL mod
    // width, row += MM_FL div width, one wrap at most) -- five instructions instead of the twelve of a division, sixteen times per trip.
    int step_c, step_r;
    box_pixel(FL, 0, 0, fb.bw, fb.inv_bw, step_c, step_r);
    const unsigned step_off = (unsigned)step_r * (unsigned)a.W + (unsigned)step_c, wrap_off = (unsigned)a.W - (unsigned)fb.bw;   // a wrap: one row down, width back
    for (int base = 0; base < nmax; base += FL * MM_SWEEP) {
        bool own[MM_SWEEP], opn[MM_SWEEP];
        int col, row;
        box_pixel(lo + base + sl, 0, 0, fb.bw, fb.inv_bw, col, row);
        const int32_t* fimg = a.face_idx + (size_t)b * hw;
        unsigned off = (unsigned)(fb.py0 + row) * (unsigned)a.W + (unsigned)(fb.px0 + col);   // (H, W <= 65535: fits 32 bits)
#pragma unroll
        for (int i = 0; i < MM_SWEEP; ++i) {
            const int idx = lo + base + i * FL + sl;
            const int fi = idx < hi ? fimg[off] : -2;
            own[i] = fi == f; opn[i] = fi == -1;
            col += step_c; off += step_off;
            if (col >= fb.bw) { col -= fb.bw; off += wrap_off; }
        }
        MM_PP_MARK(1);
        // one compacted item list for both kinds of hit: pixels these faces own (K2: add the pixel pass's contributions)
        // and uncovered pixels that may hold one of these faces among their first knum soft-mask faces (K4, Appendix A.2)
        const int n = compact_hits(own, opn, lane, st->items);
        wave_sync_lds();
        MM_PP_MARK(2);
        MM_PP_COUNT(1, n);
        // MM_ITEM_UNROLL items per lane and trip (each round is a dependent trip to memory; see the macro for why it is 1)
        for (int j0 = 0; j0 < n; j0 += 64 * MM_ITEM_UNROLL) {
            ItemLoad ld[MM_ITEM_UNROLL];
#pragma unroll
            for (int u = 0; u < MM_ITEM_UNROLL; ++u) {
                const int j = j0 + u * 64 + lane;
                ld[u].live = j < n;
                if (u >= 1 && j0 + 64 * u >= n) break;               // wave-uniform: a short list has no second half
                const unsigned it = st->items[ld[u].live ? j : 0];
                const int l = it & 63, i = (it >> 6) & 0x1FF;
                ld[u].g = l / FL;
                const FaceSlot& fs = st->slot[ld[u].g];
                box_pixel(fs.lo + base + i * FL + (l % FL), fs.px0, fs.py0, fs.bw, fs.inv_bw, ld[u].px, ld[u].py);
                const size_t pix = (size_t)b * hw + (size_t)ld[u].py * a.W + ld[u].px;   // every group of the wave sweeps the same image
                // the sweep already knows which kind of hit this is: only the three loads that kind needs are issued
                ld[u].owned = (it & 0x8000u) != 0;
                ld[u].q0 = make_float4(0.f, 0.f, 0.f, 0.f); ld[u].q1 = ld[u].q0; ld[u].sq = 0.f; ld[u].lf = 0; ld[u].q2 = 0.f;
                if (ld[u].live) {
                    ld[u].q2 = a.gp2[pix];
                    if (ld[u].owned) { ld[u].q0 = a.gp[pix * 2 + 0]; ld[u].q1 = a.gp[pix * 2 + 1]; }
                    else { const float2 sl2 = a.soft[pix]; ld[u].sq = sl2.x; ld[u].lf = __float_as_int(sl2.y); }
                }
            }
            MM_PP_MARK(3);
#pragma unroll
            for (int u = 0; u < MM_ITEM_UNROLL; ++u) {
                if (u >= 1 && j0 + 64 * u >= n) break;
                if (ld[u].live) item_finish(a, st->slot[ld[u].g], ld[u], s2, scale);
            }
        }
        wave_sync_lds();
        MM_PP_MARK(4);
    }
}

// a wave takes MM_FPW consecutive sweep items of ONE image (consecutive faces, or consecutive chunks of a big face: neighbours on
// the screen); waves walk the images round-robin.  The item's partial sums go to part[item]; the vertex backward adds the items
// of a face up.
template <int FL>
__device__ inline void face_gather_block(const BwdArgs& a, int block, SweepStageT<FL>* s_stage) {
    constexpr int FPW = 64 / FL;                                 // items per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = lane / FL, sl = lane % FL;
    SweepStageT<FL>* st = &s_stage[wave];
    const long long wid = (long long)block * 4 + MM_WAVE_UNIFORM(wave);   // wave index over (item octet, image)
    // (wave-uniform by construction; saying so makes the image's scalars -- item count, chunk size, the 32 shards of the fixed-point scale -- SCALAR
    //  loads from the scalar cache instead of 34 vector loads of one address per wave: 8 000 waves x 34 wave-loads were ~a third of the kernel's
    //  vector-memory instructions)
    const int b = MM_WAVE_UNIFORM(wid % a.B);
    const int2 ni = a.nitems[b];                                  // items of the image, pixels per chunk
    // The image's items are dealt to its waves ROUND-ROBIN (wave w takes items w, w + nw, w + 2 nw, ...): the chunks of a close-up face are
    // consecutive items, most of their pixels owned, and a wave holding eight of them in a row (a thousand hits, sixteen dependent rounds
    // of hit loads) was the tail of this kernel; spread out, every wave gets at most one or two of them.
    const int w_img = (int)(wid / a.B), nw = (ni.x + FPW - 1) / FPW;
    if (w_img >= nw) return;                                      // wave-uniform: the image has fewer items (the grid is sized for the cap)
    const int item = grp * nw + w_img;
    const bool live = item < ni.x;
    const int2 e = a.items[(size_t)b * a.item_cap + (live ? item : 0)];      // face, chunk
    const size_t o = (size_t)b * a.F + e.x;
    const FaceBox fb = face_box(a, o);
    const int lo = e.y * ni.y;
    // MM_OPT_SOFT_SKIP_CULLED: a culled face owns no pixel and took no part in the soft mask -> nothing to sweep
    const bool front = (a.options & MM_OPT_CULL_STRICT) ? fb.nz > 0.f : fb.nz >= 0.f;
    const int hi = live && (front || !(a.options & MM_OPT_SOFT_SKIP_CULLED)) ? min(fb.npx, lo + ni.y) : lo;
    MM_PP_BEGIN();
    float inv;
    const float scale = face_sum_scale(a, b, ni.y, inv);
    face_sweep(a, st, b, e.x, lane, fb, lo, hi, scale MM_PP_PASS);
    if (live) for (int k = sl; k < 9; k += FL) a.part[((size_t)b * a.item_cap + item) * 12 + k] = (float)st->slot[grp].acc[k] * inv;
    MM_PP_MARK(5);
    MM_PP_FLUSH(gather_face, wid);
}

// (this wave's part of the value; networks.py:376-389: image_weight * loss_image + 1 * (loss_mask + contour * loss_contour))
__device__ inline float fused_loss_value(const long long* ltot, int B, int H, int W, float image_weight, float contour, int lane) {
    float l1 = 0.f, iou = 0.f, cs = 0.f;
    for (int bb = lane; bb < B; bb += 64) {
        float s0, s1, s2;
        loss_totals(ltot, bb, s0, s1, s2);
        l1 += s0; iou += s1 / (s2 + 1e-10f);
        if (contour > 0.f) cs += loss_contour_total(ltot, bb);
    }
    l1 = wave_sum(l1); iou = wave_sum(iou);
    float loss_mask = 1.f - iou / (float)B;
    if (contour > 0.f) loss_mask += (wave_sum(cs) / ((float)B * (float)H * (float)W)) * contour;
    return image_weight * (l1 / ((float)B * 3.f * (float)H * (float)W)) + 1.f * loss_mask;
}

// One launch for both gathers: they only depend on the pixel pass, and each is latency-bound with a long tail, so their
// workgroups are interleaved in a single grid (texture tiles first: they are the heavier ones).
#ifndef MM_GATHER_LB
#define MM_GATHER_LB 8            // waves per SIMD the register allocation is held to (64 VGPRs, no spills)
#endif
template <int FL>
__global__ __launch_bounds__(256, MM_GATHER_LB) void gather_bwd_kernel(BwdArgs a, int ntex, int dbg_skip) {
    MM_TIMELINE_BEGIN();
#ifdef MM_PHASE_PROF                                            // timing experiments only (results are wrong): leave one kind of workgroup out
    if ((dbg_skip & 1) && (int)blockIdx.x < ntex) return;
    if ((dbg_skip & 2) && (int)blockIdx.x >= ntex) return;
#endif
    // the two kinds of workgroup never coexist in one workgroup: their LDS is overlaid (more workgroups per CU)
    constexpr size_t kLds = sizeof(float) * 3 * MM_TS * MM_TS > sizeof(SweepStageT<FL>) * 4 ? sizeof(float) * 3 * MM_TS * MM_TS : sizeof(SweepStageT<FL>) * 4;
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[kLds];
    int (*s_acc)[MM_TS * MM_TS] = reinterpret_cast<int (*)[MM_TS * MM_TS]>(s_raw);
    SweepStageT<FL>* s_stage = reinterpret_cast<SweepStageT<FL>*>(s_raw);
    if (a.gt && a.loss && blockIdx.x == gridDim.x - 1 && threadIdx.x < 64) {  // fused recon_data value: fixed-order sum over images
        const float v = fused_loss_value(a.ltot, a.B, a.H, a.W, a.image_weight, a.contour, threadIdx.x);
        if (threadIdx.x == 0) a.loss[0] = v;
    }
    // Texture tiles first, then the faces.  (r04, re-measured on today's kernel: the face workgroups that usually have items first, then the tiles,
    //  then the rest of the face grid: 32.1 / 185.4 us against 29.7 / 176.6 at 128x128 B=48 / B=384.)  Measured alternatives (gather_bwd us at configs 2 / 3 / 5; this order 38.5 / 93.6 / 304): faces
    // first 35.9 / 104.4 / 293.5; the two kinds alternating 43.8 / 114.5 / 417 -- a CU that runs both code paths at once loses more than
    // the earlier start of the slowest workgroups gains.
    if ((int)blockIdx.x < ntex) texture_gather_block(a, blockIdx.x, s_acc);
    else face_gather_block(a, blockIdx.x - ntex, s_stage);
    MM_TIMELINE_END(gather_bwd);
}

// the fused recon_data value on its own (mm_render_fused_loss): the same fixed-order sum over images the gather kernel's last
// workgroup performs, for callers that need the loss before the backward
__global__ __launch_bounds__(64) void fused_loss_kernel(const long long* ltot, int B, int H, int W, float image_weight, float contour, float* loss) {
    const float v = fused_loss_value(ltot, B, H, W, image_weight, contour, threadIdx.x);
    if (threadIdx.x == 0) loss[0] = v;
}

int launch_fused_loss(const MMRenderDesc* d, const Workspace& w, hipStream_t s) {
    hipLaunchKernelGGL(fused_loss_kernel, dim3(1), dim3(64), 0, s, w.ltot, d->B, d->H, d->W, d->fused_image_weight, d->fused_contour, d->fused_loss);
    return launch_ok("fused_loss");
}

int launch_raster_bwd(const MMRenderDesc* d, const MMRenderGrads* g, const Workspace& w, hipStream_t s) {
    BwdArgs a;
    a.B = d->B; a.H = d->H; a.W = d->W; a.F = d->F; a.Ht = d->Ht; a.Wt = d->Wt; a.knum = d->knum;
    a.blocks_x = (d->W + MM_BLOCK_PX - 1) / MM_BLOCK_PX;
    a.blocks_per_image = w.blocks_per_image; a.options = d->options;
    a.mult = d->multiplier; a.eps = d->eps; a.sigmainv = d->sigmainv; a.infl = d->boxlen * d->multiplier;
    a.kx = d->multiplier / (float)d->W; a.ky = d->multiplier / (float)d->H; a.sig2 = d->sigmainv / (d->multiplier * d->multiplier);
    a.geo = w.geo; a.face_uvs = d->face_uvs; a.fn = d->face_normals; a.textures = d->textures; a.lights = d->lights; a.bg = d->bg;
    a.face_idx = d->face_idx; a.soft = w.soft; a.fflag = walk_flags_mode(d->options, w.bin_shift) ? w.fflag : nullptr;
    a.sweep_sx = sweep_shrink(d->boxlen, d->W); a.sweep_sy = sweep_shrink(d->boxlen, d->H);
    a.grad_rgba = g->grad_rgba;   // (face flags: only the compacting walk of the forward sets them)
    a.gp = w.gp; a.gp2 = w.gp2; a.dl_part = w.dl_part; a.grad_bg = g->grad_bg;
    a.ticket = w.ticket;
    a.tcur = w.tcur; a.tdrop = w.tdrop; a.trcnt = w.trcnt; a.toff = w.toff; a.tstatus = w.tstatus; a.trec = w.trec; a.ntiles_ = w.ntiles; a.trcap = w.trcap;
    a.gmax = w.gmax;                                             // (B, MM_GSHARD, 8): two maxima per 32-byte sector
    a.status_flag = d->status_flag;
    a.gt = d->fused_gt; a.rgba = d->rgba; a.grad_loss = d->fused_grad_loss; a.loss = d->fused_totals ? nullptr : d->fused_loss;
    a.image_weight = d->fused_image_weight; a.contour = d->fused_gt ? d->fused_contour : 0.f; a.ltot = w.ltot;
    if (d->fused_gt && d->fused_totals) { a.options |= MM_INT_DEFERRED; a.ltot = reinterpret_cast<const long long*>(d->fused_totals); }   // (deferred fusion: see BwdArgs::ltot)
    a.items = w.items; a.nitems = w.nitems; a.part = w.part; a.item_cap = w.item_cap;
    a.plan_chunkmap = w.chunkmap; a.plan_items = w.items; a.plan_nitems = w.nitems; a.plan_wgs = d->F > 4096 ? MM_PLAN_WGS : 1;
    a.ntx = (d->Wt + MM_TS - 1) / MM_TS; a.nty = (d->Ht + MM_TS - 1) / MM_TS;
    a.grad_textures = g->grad_textures;
    // w.tcnt is zero here: cleared by the vertex stage of the forward and again by every vertex backward (no memset launch)
    launch_pixel_bwd(a, d, s);
    if (launch_ok("pixel_bwd") != MM_OK) return MM_ERR_LAUNCH;
    {
        ProfScope p(d->prof_events, MM_PROF_GATHER_BWD, s);
        const int ntex = a.ntx * a.nty * d->B;
        // lanes per sweep item: 8 (eight items per wave) where one batch is in flight; 4 (sixteen items per wave, half the waves, each with twice the
        // hits) for the large batches that run the chip in many rounds -- r06, profiles/r06_sweep_shape_ab.md: -5 % of this kernel at B=384, +4 to +15 % at B=48
        // (MM_OPT_MANY_IN_FLIGHT: several calls share the chip -- the same regime, said by the caller: +1 % with four B=48 steps on four streams)
        const bool fl4 = MM_FL == 8 && (d->B >= MM_FL4_MIN_B || (d->options & MM_OPT_MANY_IN_FLIGHT) != 0);
        const int fpw = fl4 ? 16 : MM_FPW;
        const long long nwaves = (long long)d->B * ((w.item_cap + fpw - 1) / fpw);          // (item group, image), sized for the cap: waves
        const unsigned nface = (unsigned)((nwaves + 3) / 4);                                // beyond an image's item count exit at once
        int dbg_skip = 0;
#ifdef MM_PHASE_PROF
        if (const char* e = getenv("MM_DBG_GATHER")) dbg_skip = atoi(e);
#endif
        if (fl4) hipLaunchKernelGGL(gather_bwd_kernel<4>, dim3(ntex + nface), dim3(256), 0, s, a, ntex, dbg_skip);
        else hipLaunchKernelGGL(gather_bwd_kernel<MM_FL>, dim3(ntex + nface), dim3(256), 0, s, a, ntex, dbg_skip);
    }
    return launch_ok("raster_bwd");
}

}  // namespace mm
