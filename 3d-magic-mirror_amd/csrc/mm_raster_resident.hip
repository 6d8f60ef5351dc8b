// mm_raster_resident.hip -- forward pixel stage for templates that fit a workgroup's LDS (gfx950).
//
// Same contract and the same arithmetic as mm_raster.hip (kaolin packed_rasterize_forward + dibr_soft_mask_forward +
// texture_mapping + spherical_harmonic_lighting + composite, /root/reference/networks.py:297-317),
// organised around what is slow on this machine: a dependent trip to HBM/L2 costs a wave about a microsecond, and the
// streamed kernel takes roughly a dozen of them per tile (tile order -> bin masks -> face records, twice, -> winner ->
// normals/uvs -> texels).  The reference's templates are tiny (642 vertices, 1280 faces: 13 KB of transformed vertices), so:
//   * one 256-thread workgroup renders a 32x32 pixel REGION of one image.  It transforms the image's vertices itself
//     (same expressions as the vertex stage, so the same bits) into LDS: screen xy, camera xyz.
//   * its four waves sweep all faces once (coalesced read of the static index buffer), test each face's box -- plain and
//     inflated by the soft-mask margin -- against the region, and leave an index-ordered list of the region's faces in
//     LDS (vertex ids + face id + "front facing & touches" bit, 8 bytes each).  No bin masks, no sort kernel.
//   * each wave then renders four of the region's sixteen 8x8 tiles.  Per tile the list is walked 64 entries at a
//     time, box-tested against the tile (candidate-parallel, separable), hits are compacted in order into the wave's
//     staging rows and evaluated exactly like the streamed kernel: pair-parallel, 64-bit LDS max of (z, -rank) for colour,
//     integer log2 sums for the silhouette.  Everything on that path is LDS.
//   * only the shading touches memory: uvs of the winning face (static, L2), then the texels; background and ground truth
//     are requested up front.  Normals of winners are recomputed from the LDS vertices.
// The vertex stage still runs first (it produces T, the face records the backward reads, and attributes['face_normals']).
#include "mm_raster_common.h"

namespace mm {

#define MM_REGION_PX 32
#define MM_REGION_TILES 16     // 4 x 4 tiles of 8 x 8 pixels
#define MM_RES_WAVES 4
#define MM_RES_MAXCH 8         // 64-face chunks per wave in the region sweep: F <= 4 * 8 * 64

// -DMM_RES_PROF: per-phase cycle totals over all waves (debug builds only; read back with mm_debug_resident_prof)
#ifdef MM_RES_PROF
__device__ unsigned long long g_res_prof[4096 * 8];             // per wave, per phase: cycles (plain stores, no atomics)
#define MM_PROF_ARG , unsigned long long* prof_acc_
#define MM_PROF_PASS , prof_acc_
#define MM_PROF_MARK(slot) do { const unsigned long long now_ = clock64(); prof_acc_[slot] += now_ - prof_t_; prof_t_ = clock64(); } while (0)
#else
#define MM_PROF_ARG
#define MM_PROF_PASS
#define MM_PROF_MARK(slot) do { } while (0)
#endif

struct __attribute__((aligned(16))) ResidentStage {
    float4 p0[64];      // ax, ay, bx, by   (multiplier units)
    float4 p1[64];      // cx, cy, az, bz
    float4 p2[64];      // cz, 0, rank (position in the region list, bits), face id (bits)
    unsigned short pairs[MM_PAIR_ROUND];
    unsigned long long key[64];
    long long logsum[64];
    int zeros[64];
    unsigned short cand[64];                // region-list positions of the batch being formed
};

struct ResidentLds {
    ResidentStage* stage;                   // [MM_RES_WAVES]
    float *sx, *sy, *cz, *cx, *cy;          // [V] screen xy (multiplier units), camera xyz
    uint64_t* list;                         // [F] region faces, ascending: i0 | i1 << 16 | i2 << 32 | face << 48 | front << 63
    unsigned* tmask;                        // [F] per entry: bit t = inflated box may touch tile t of the region (4x4, row-major);
                                            //     bit 16 + t = front facing and plain box may touch it
    int* ctl;                               // [0..3] per-wave list counts, [4] tile tickets, [8..23] tile counts, [24..39] tile order
};

__host__ __device__ inline size_t resident_lds_bytes(int V, int F) {
    const size_t Vp = (size_t)((V + 3) & ~3);
    return MM_RES_WAVES * sizeof(ResidentStage) + 20 * Vp + 12 * (size_t)F + 192;
}

__device__ inline ResidentLds carve_lds(unsigned char* raw, int V, int F) {
    ResidentLds L;
    const int Vp = (V + 3) & ~3;
    L.stage = (ResidentStage*)raw;
    L.sx = (float*)(raw + MM_RES_WAVES * sizeof(ResidentStage));
    L.sy = L.sx + Vp; L.cz = L.sy + Vp; L.cx = L.cz + Vp; L.cy = L.cx + Vp;
    L.list = (uint64_t*)(L.cy + Vp);
    L.tmask = (unsigned*)(L.list + F);
    L.ctl = (int*)(L.tmask + F);
    return L;
}

struct Tri { float ax, ay, bx, by, cx, cy; int i0, i1, i2; };

__device__ inline Tri list_triangle(const ResidentLds& L, uint64_t e) {
    Tri r;
    r.i0 = (int)(e & 0xFFFFull); r.i1 = (int)((e >> 16) & 0xFFFFull); r.i2 = (int)((e >> 32) & 0xFFFFull);
    r.ax = L.sx[r.i0]; r.ay = L.sy[r.i0]; r.bx = L.sx[r.i1]; r.by = L.sy[r.i1]; r.cx = L.sx[r.i2]; r.cy = L.sy[r.i2];
    return r;
}

// unit normal of a triangle from the camera-space vertices in LDS: the vertex stage's expressions (a5/a6)
__device__ inline void list_normal(const ResidentLds& L, int i0, int i1, int i2, float& n0, float& n1, float& n2) {
    const float Ax = L.cx[i0], Ay = L.cy[i0], Az = L.cz[i0];
    const float e0[3] = {L.cx[i1] - Ax, L.cy[i1] - Ay, L.cz[i1] - Az};
    const float e1[3] = {L.cx[i2] - Ax, L.cy[i2] - Ay, L.cz[i2] - Az};
    float n[3];
    cross3(e0, e1, n);
    const float len = sqrtf((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
    const float den = len + 1e-10f;
    n0 = n[0] / den; n1 = n[1] / den; n2 = n[2] / den;
}

// Stage region-list entry `idx` in slot `slot` of the wave's batch and return its exact pixel masks for this tile:
// ms = inflated box (every face), mh = plain box of a front face -- the per-pixel closed-box tests of a8.
__device__ inline void stage_entry(const RasterArgs& a, const TileCtx& t, ResidentStage* st, const ResidentLds& L, int idx, int slot,
                                   bool want_soft, bool want_hard, uint64_t& ms, uint64_t& mh) {
    const uint64_t e = L.list[idx];
    const Tri r = list_triangle(L, e);
    st->p0[slot] = make_float4(r.ax, r.ay, r.bx, r.by);
    st->p1[slot] = make_float4(r.cx, r.cy, L.cz[r.i0], L.cz[r.i1]);
    st->p2[slot] = make_float4(L.cz[r.i2], 0.f, __int_as_float(idx), __int_as_float((int)((e >> 48) & 0x7FFFull)));
    const float xmin = fminf(fminf(r.ax, r.bx), r.cx), ymin = fminf(fminf(r.ay, r.by), r.cy);
    const float xmax = fmaxf(fmaxf(r.ax, r.bx), r.cx), ymax = fmaxf(fmaxf(r.ay, r.by), r.cy);
    ms = want_soft ? box_pixels(t, xmin - a.infl, ymin - a.infl, xmax + a.infl, ymax + a.infl) : 0ull;
    mh = (want_hard && (e >> 63)) ? box_pixels(t, xmin - 0.f, ymin - 0.f, xmax + 0.f, ymax + 0.f) : 0ull;
}

// General walk (tiles with more than 64 candidates): the region's entries whose tile bit is set are gathered, in order, into
// batches of 64; body(n, m) sees each batch's hit masks candidate-major or, with kTranspose, pixel-major -- the contract of
// the streamed kernel's walk.
template <bool kHard, bool kTranspose, class Body>
__device__ inline void walk_tile(const RasterArgs& a, const TileCtx& t, ResidentStage* st, const ResidentLds& L, int nr, int tbit, Body&& body) {
    const int shift = kHard ? 16 + tbit : tbit;
    int staged = 0;
    auto flush = [&](int n) -> bool {
        wave_lds_sync();
        uint64_t ms = 0, mh = 0;
        if (t.lane < n) stage_entry(a, t, st, L, st->cand[t.lane], t.lane, !kHard, kHard, ms, mh);
        const uint64_t mc = kHard ? mh : ms;
        const uint64_t m = kTranspose ? wave_transpose64(mc, t.lane) : mc;
        wave_lds_sync();
        const bool go = body(n, m);
        wave_lds_sync();
        return go;
    };
    for (int c0 = 0; c0 < nr; c0 += 64) {
        const int idx = c0 + t.lane;
        const bool hit = idx < nr && ((L.tmask[idx] >> shift) & 1u);
        const uint64_t bal = __ballot(hit);
        if (bal == 0) continue;
        const int p = staged + ballot_rank(bal);
        if (hit && p < 64) st->cand[p] = (unsigned short)idx;
        const int total = staged + __popcll(bal);
        if (total < 64) { staged = total; continue; }
        if (!flush(64)) return;
        if (hit && p >= 64) st->cand[p - 64] = (unsigned short)idx;
        staged = total - 64;
    }
    if (staged > 0) (void)flush(staged);
}

template <bool kNoMask>
__device__ inline void render_tile(const RasterArgs& a, const TileCtx& t, ResidentStage* st, const ResidentLds& L, int nr, int tbit MM_PROF_ARG) {
#ifdef MM_RES_PROF
    unsigned long long prof_t_ = clock64();
#endif
    st->key[t.lane] = 0ull; st->logsum[t.lane] = 0ll; st->zeros[t.lane] = 0;
    // the tile's candidates (inflated boxes: a superset of the colour candidates), by their bit in the region sweep's masks
    int nt = 0;
    for (int c0 = 0; c0 < nr; c0 += 64) {
        const int idx = c0 + t.lane;
        const bool hit = idx < nr && ((L.tmask[idx] >> tbit) & 1u);
        const uint64_t bal = __ballot(hit);
        const int p = nt + ballot_rank(bal);
        if (hit && p < 64) st->cand[p] = (unsigned short)idx;
        nt += __popcll(bal);
    }
    wave_lds_sync();
    MM_PROF_MARK(2);
    const bool single = nt <= 64;                                // the common case: one batch serves both passes
    uint64_t ms = 0, mh = 0;
    if (single) {
        if (t.lane < nt) stage_entry(a, t, st, L, st->cand[t.lane], t.lane, true, true, ms, mh);
        wave_lds_sync();
    }

    MM_PROF_MARK(3);
    // K1: argmax over (z, -rank), rank = position in the index-ordered region list
    if (single) {
        if (__ballot(mh != 0)) pair_parallel(t, st, mh, [&](int j, int l, bool live) { hard_pair(a, t, st, j, l, live); });
    } else {
        walk_tile<true, false>(a, t, st, L, nr, tbit, [&](int n, uint64_t mc) {
            pair_parallel(t, st, mc, [&](int j, int l, bool live) { hard_pair(a, t, st, j, l, live); });
            return true;
        });
    }
    wave_lds_sync();
    MM_PROF_MARK(4);
    const unsigned long long k = st->key[t.lane];
    Hit h;
    h.f = -1; h.w0 = h.w1 = h.w2 = 0.f;
    float n0 = 0.f, n1 = 0.f, n2 = 0.f;
    if (k != 0ull) {                                             // barycentrics + unit normal of the winner (vertex-stage expressions)
        const uint64_t e = L.list[depth_key_rank(k)];
        h.f = (int)((e >> 48) & 0x7FFFull);
        const Tri r = list_triangle(L, e);
        float nrm;
        edge_weights(r.ax, r.ay, r.bx, r.by, r.cx, r.cy, t.x0, t.y0, a.eps, h.w0, h.w1, h.w2, nrm);
        h.w0 /= nrm; h.w1 /= nrm; h.w2 /= nrm;
        list_normal(L, r.i0, r.i1, r.i2, n0, n1, n2);
    }

    // K3: soft silhouette of the uncovered lanes, as in the streamed kernel
    SoftState ss = {1.f, 0, 0x7FFFFFFF};
    const bool open = t.in_img && h.f < 0;
    if (__ballot(open) && nt > 0) {
        int cnt = 0;
        const float s2 = a.sigmainv / (a.mult * a.mult);
        auto take = [&](uint64_t sm) {                            // pixel-major: the first knum hits of this pixel, in order
            sm = soft_take(sm, open, a.knum - cnt);
            cnt += __popcll(sm);
            if (sm != 0 && cnt >= a.knum) ss.lastf = __float_as_int(st->p2[63 - __clzll((unsigned long long)sm)].w);
            pair_parallel(t, st, sm, [&](int l, int j, bool live) { soft_pair(a, t, st, s2, l, j, live); });
        };
        if (single) {
            take(wave_transpose64(ms, t.lane));
        } else {
            walk_tile<false, true>(a, t, st, L, nr, tbit, [&](int n, uint64_t sm) {
                take(sm);
                return __ballot(open && cnt < a.knum) != 0;      // every open lane already holds knum faces: stop
            });
        }
        wave_lds_sync();
        ss.zeros = st->zeros[t.lane];
        ss.qnz = exp2f((float)((double)st->logsum[t.lane] * (1.0 / 4294967296.0)));
    }
    MM_PROF_MARK(5);
    shade_store<kNoMask>(a, t, h, n0, n1, n2, ss);
    MM_PROF_MARK(6);
}

template <bool kNoMask>
__global__ __launch_bounds__(MM_RES_WAVES * 64) void raster_fwd_resident_kernel(RasterArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const ResidentLds L = carve_lds(s_raw, a.V, a.F);
#ifdef MM_RES_PROF
    unsigned long long prof_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long prof_t_ = clock64();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b, region;
    map_block(blockIdx.x, a.B, a.regions_per_image, b, region);     // all regions of an image on one XCD (its L2 holds the texture)

    // ---- vertices -> LDS (prepare_vertices, a5): the vertex stage's expressions, so the same values
    {
        float T[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = a.T[b * 12 + i];
        const float* vb = a.vertices + (size_t)b * a.V * 3;
        for (int v = tid; v < a.V; v += MM_RES_WAVES * 64) {
            const Float3 c = to_camera(vb + (size_t)v * 3, T);
            const float pz = c.z * a.proj2;
            L.sx[v] = ((c.x * a.proj0) / pz) * a.mult;
            L.sy[v] = ((c.y * a.proj1) / pz) * a.mult;
            L.cx[v] = c.x; L.cy[v] = c.y; L.cz[v] = c.z;
        }
    }

    // ---- region sweep: which faces can touch which of the region's 4x4 tiles (closed-box test against each tile column's /
    // row's pixel-centre extent: the bin stage's exactly conservative test), kept in index order
    const int rx = region % a.regions_x, ry = region / a.regions_x;
    const int rpx0 = rx * MM_REGION_PX, rpy0 = ry * MM_REGION_PX;
    float xl[4], xh[4], yl[4], yh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int px0 = rpx0 + i * MM_TILE, py0 = rpy0 + i * MM_TILE;
        xl[i] = pixel_x(px0, a.W, a.mult); xh[i] = pixel_x(min(px0 + MM_TILE - 1, a.W - 1), a.W, a.mult);
        yh[i] = pixel_y(py0, a.H, a.mult); yl[i] = pixel_y(min(py0 + MM_TILE - 1, a.H - 1), a.H, a.mult);
    }
    const int per = ((a.F + MM_RES_WAVES * 64 - 1) / (MM_RES_WAVES * 64)) * 64;      // faces per wave
    uint64_t ent[MM_RES_MAXCH];
    unsigned tms[MM_RES_MAXCH];
    int run = 0;
    int fi[MM_RES_MAXCH][3];
#pragma unroll
    for (int k = 0; k < MM_RES_MAXCH; ++k) {                       // index loads do not wait for the vertices
        const int f = wave * per + k * 64 + lane;
        const bool on = k * 64 < per && f < a.F;
        fi[k][0] = on ? a.faces[(size_t)f * 3 + 0] : -1;
        fi[k][1] = on ? a.faces[(size_t)f * 3 + 1] : 0;
        fi[k][2] = on ? a.faces[(size_t)f * 3 + 2] : 0;
    }
    __syncthreads();
    MM_PROF_MARK(0);
#pragma unroll
    for (int k = 0; k < MM_RES_MAXCH; ++k) {
        ent[k] = 0ull; tms[k] = 0u;
        if (fi[k][0] >= 0) {
            const int f = wave * per + k * 64 + lane;
            const int i0 = fi[k][0], i1 = fi[k][1], i2 = fi[k][2];
            const float ax = L.sx[i0], ay = L.sy[i0], bx = L.sx[i1], by = L.sy[i1], cx = L.sx[i2], cy = L.sy[i2];
            const float xmin = fminf(fminf(ax, bx), cx), ymin = fminf(fminf(ay, by), cy);
            const float xmax = fmaxf(fmaxf(ax, bx), cx), ymax = fmaxf(fmaxf(ay, by), cy);
            unsigned cols = 0, rows = 0, colh = 0, rowh = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                cols |= (unsigned)(!(xmax + a.infl < xl[i] || xmin - a.infl > xh[i])) << i;
                rows |= (unsigned)(!(ymax + a.infl < yl[i] || ymin - a.infl > yh[i])) << i;
                colh |= (unsigned)(!(xmax < xl[i] || xmin > xh[i])) << i;
                rowh |= (unsigned)(!(ymax < yl[i] || ymin > yh[i])) << i;
            }
            unsigned ts = 0, th = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ts |= ((rows >> r) & 1u) ? (cols << (4 * r)) : 0u;
                th |= ((rowh >> r) & 1u) ? (colh << (4 * r)) : 0u;
            }
            if (th) {                                            // colour only sees front faces: unit normal z >= 0 (a8)
                float n0, n1, n2;
                list_normal(L, i0, i1, i2, n0, n1, n2);
                if (!(n2 >= 0.f)) th = 0;
            }
            tms[k] = ts | (th << 16);
            ent[k] = (uint64_t)(unsigned)i0 | ((uint64_t)(unsigned)i1 << 16) | ((uint64_t)(unsigned)i2 << 32) |
                     ((uint64_t)(unsigned)f << 48) | ((uint64_t)(th != 0) << 63);
        }
        run += __popcll(__ballot(tms[k] != 0));
    }
    if (lane == 0) L.ctl[wave] = run;
    __syncthreads();
    int base = 0, nr = 0;
#pragma unroll
    for (int w = 0; w < MM_RES_WAVES; ++w) { const int c = L.ctl[w]; nr += c; if (w < wave) base += c; }
#pragma unroll
    for (int k = 0; k < MM_RES_MAXCH; ++k) {
        const bool hit = tms[k] != 0;
        const uint64_t bal = __ballot(hit);
        if (hit) { const int p = base + ballot_rank(bal); L.list[p] = ent[k]; L.tmask[p] = tms[k]; }
        base += __popcll(bal);
    }
    __syncthreads();

    // ---- tile schedule: the region's 16 tiles, most candidates first, handed out from a shared counter (longest-processing-
    // time-first keeps the four waves level however the silhouette cuts the region).  ctl[8..23] counts, ctl[24..39] order.
    {
        int c[4] = {0, 0, 0, 0};                                  // wave w counts tiles 4w .. 4w+3
        for (int q = 0; q < nr; q += 64) {
            const unsigned tm = (q + lane < nr ? L.tmask[q + lane] : 0u) >> (4 * wave);
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] += __popcll(__ballot((tm >> j) & 1u));
        }
        if (lane < 4) L.ctl[8 + 4 * wave + lane] = lane == 0 ? c[0] : lane == 1 ? c[1] : lane == 2 ? c[2] : c[3];
        if (tid == 0) L.ctl[4] = 0;
    }
    __syncthreads();
    if (tid < MM_REGION_TILES) {
        const int mine = L.ctl[8 + tid];
        int rank = 0;
#pragma unroll
        for (int j = 0; j < MM_REGION_TILES; ++j) { const int o = L.ctl[8 + j]; rank += (o > mine) || (o == mine && j < tid); }
        L.ctl[24 + rank] = tid;
    }
    __syncthreads();
    MM_PROF_MARK(1);

    ResidentStage* st = L.stage + wave;
    const int blocks_y = a.blocks_per_image / a.blocks_x;
    for (;;) {
        // every lane takes part in the LDS atomic (lane 0 adds 1, the others 0): no divergent region around the ticket draw
        const int drawn = atomicAdd(&L.ctl[4], lane == 0 ? 1 : 0);
        const int slot = __shfl(drawn, 0, 64);
        if (slot >= MM_REGION_TILES) break;
        const int ti = L.ctl[24 + slot];
        TileCtx t;
        t.b = b; t.lane = lane;
        t.tx0 = rpx0 + (ti & 3) * MM_TILE; t.ty0 = rpy0 + (ti >> 2) * MM_TILE;
        const int bx = t.tx0 / MM_BLOCK_PX, by = t.ty0 / MM_BLOCK_PX;
        if (bx < a.blocks_x && by < blocks_y) {                  // else: beyond the image's last 16x16 block, nothing to write
            t.blk = by * a.blocks_x + bx;
            t.wave = ((t.ty0 / MM_TILE) & 1) * 2 + ((t.tx0 / MM_TILE) & 1);
            t.mask = nullptr;
            t.empty = false;
            tile_pixels(a, t);
            render_tile<kNoMask>(a, t, st, L, nr, ti MM_PROF_PASS);
            MM_PROF_MARK(7);
        }
    }
#ifdef MM_RES_PROF
    if (lane == 0 && blockIdx.x * MM_RES_WAVES + wave < 4096)
        for (int i = 0; i < 8; ++i) g_res_prof[(blockIdx.x * MM_RES_WAVES + wave) * 8 + i] = prof_acc_[i];
#endif
}

bool resident_path(const MMRenderDesc* d) {
    return (d->options & MM_OPT_RESIDENT) && !(d->options & MM_OPT_STREAMED) && d->V <= 65536 && d->F <= MM_RES_WAVES * MM_RES_MAXCH * 64 &&
           resident_lds_bytes(d->V, d->F) <= 64 * 1024;
}

int launch_raster_fwd_resident(const MMRenderDesc* d, const Workspace& w, hipStream_t s) {
    RasterArgs a = make_raster_args(d, w);
    a.regions_x = (d->W + MM_REGION_PX - 1) / MM_REGION_PX;
    a.regions_per_image = a.regions_x * ((d->H + MM_REGION_PX - 1) / MM_REGION_PX);
    const dim3 grid((unsigned)(d->B * a.regions_per_image));
    const size_t lds = resident_lds_bytes(d->V, d->F);
    ProfScope ps(d->prof_events, MM_PROF_RASTER_FWD, s);
    if (d->no_mask) hipLaunchKernelGGL(raster_fwd_resident_kernel<true>, grid, dim3(MM_RES_WAVES * 64), lds, s, a);
    else hipLaunchKernelGGL(raster_fwd_resident_kernel<false>, grid, dim3(MM_RES_WAVES * 64), lds, s, a);
    return launch_ok("raster_fwd_resident");
}

}  // namespace mm

#ifdef MM_RES_PROF
extern "C" int mm_debug_resident_prof(unsigned long long* out, int nwaves) {      // out[nwaves * 8]
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(mm::g_res_prof), (size_t)nwaves * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
