// mm_raster_walk.h -- the candidate walk of the streamed pixel kernels (gfx950): bin mask -> ordered raw batches of 64 candidates ->
// box tests against the tile -> BALLOT / PREFIX COMPACTION of the candidates that touch it into an ordered LDS queue -> (when the queue
// is full, or at the end) balanced (pixel, candidate) pair evaluation with exact LDS atomics.  Shared by the fused render kernel
// (mm_raster.hip) and by the un-fused kaolin-shaped dibr_rasterization entry point (mm_dibr.hip): both run the SAME walk, so face_idx /
// barycentrics / soft-mask state are bit-identical between the two boundaries.
#pragma once
#include "mm_raster_common.h"

namespace mm {


// per-wave LDS staging: the QUEUE of the candidates that touch the tile (three float4 rows + their two pixel masks, face order) + the id
// list of one mask window
struct __attribute__((aligned(16))) WaveStage {
    float4 p0[64];      // ax, ay, bx, by   (multiplier units)
    float4 p1[64];      // cx, cy, az, bz
    float4 p2[64];      // cz, unit normal z, face id (bits), depth bound (depth_ord bits)
    unsigned long long qm[2][64];           // queue: per candidate its pixel masks (bit p = pixel p of the tile): [0] front-face box (colour),
                                            // [1] inflated box (silhouette).  The cooperative walk keeps per-pixel ints here instead (coop_cnt / coop_lastf)
    unsigned short ids[MM_GROUP_WORDS * 64];
    unsigned short pairs[MM_PAIR_ROUND];    // (row << 8) | column of the bit matrix being evaluated (candidate, pixel) or (pixel, candidate)
    unsigned long long key[64];             // per pixel: best (orderable z << 32 | ~face) so far; 0 = none
    long long logsum[64];                   // per pixel: sum of log2(1-p) in 2^-32 fixed point (integer adds commute)
#ifdef MM_LDS_DIET
    int zeros[16];
#else
    int zeros[64];                          // per pixel: number of factors (1-p) that are exactly 0
#endif
    unsigned long long takenw[64];          // faces of the current chunk of 4096 that some pixel took into its silhouette product: bit b of word w = face
                                            // chunk * 4096 + w * 64 + b (LDS ors; written to RasterArgs::fflag when the chunk has been walked)
    int npair[4];                           // cooperative walk: this wave's colour / silhouette pair counts of the round
};
#ifndef MM_NO_STAGE_ASSERT
static_assert(sizeof(WaveStage) <= 8192, "four of them must fit 32 KiB: five workgroups per CU");
#endif
// cooperative walk: per pixel, inflated-box hits of this wave's batch of the current round / id of the knum-th silhouette face taken
__device__ inline int* coop_cnt(WaveStage* st) { return reinterpret_cast<int*>(&st->qm[0][0]); }
__device__ inline int* coop_lastf(WaveStage* st) { return reinterpret_cast<int*>(&st->qm[1][0]); }

// The candidates some pixel of the tile has just taken into its silhouette product (sm: this lane's pixel, bit j = staged candidate j):
// their faces are flagged for the backward's sweep.  NOT with a store on the spot: vector-memory operations retire in order, so a store
// issued in the middle of the walk holds up the candidate records fetched behind it for a full trip to memory (measured: +3 us at 128x128,
// +8 us at 256x256 for one store per flush).  The faces are noted in an LDS bitmap over the current chunk of 4096 faces instead and
// written out once the chunk has been walked (flush_taken) -- for meshes of up to 4096 faces that is after the tile's last fetch.
// cbase: first mask word of the chunk being walked; a queued candidate of an EARLIER chunk (carried over a chunk border) is stored directly.
__device__ inline void mark_taken(const RasterArgs& a, const TileCtx& t, const WaveStage* st, WaveStage* acc, uint64_t ms, uint64_t openm, int cbase) {
    // ms: THIS LANE'S CANDIDATE, bit p = pixel p lies in its inflated box; openm: the pixels that are uncovered and still taking faces.
    // (A superset of the faces actually taken -- a pixel's "first knum" cut may leave this one out -- which costs the backward a sweep
    //  item now and then, and this walk no cross-lane reduction.)
    if (!a.fflag || !(ms & openm)) return;
    const int f = __float_as_int(st->p2[t.lane].z), rel = f - cbase * 64;
    if (rel >= 0) atomicOr(&acc->takenw[rel >> 6], 1ull << (rel & 63));
    else a.fflag[((size_t)t.b * a.F + f) * 2 + 1] = 1;
}
// lane = word of the chunk: one idempotent store per noted face, then the word is cleared for the next chunk
__device__ inline void flush_taken(const RasterArgs& a, const TileCtx& t, WaveStage* acc, int cbase) {
    if (!a.fflag) return;
    unsigned long long w = acc->takenw[t.lane];
    acc->takenw[t.lane] = 0ull;
    while (w) {
        const int bit = __ffsll(w) - 1;
        w &= w - 1;
        a.fflag[((size_t)t.b * a.F + (size_t)(cbase + t.lane) * 64 + bit) * 2 + 1] = 1;
    }
}

// the last chunk's noted faces, after the tile's epilogue (behind every load of the wave: nothing waits for these stores)
__device__ inline void flush_taken_last(const RasterArgs& a, const TileCtx& t, WaveStage* acc) {
    if (!a.fflag || t.empty) return;
    wave_lds_sync();
    flush_taken(a, t, acc, ((a.words - 1) / 64) * 64);
}

#ifndef MM_TAKEN_EXACT
#define MM_TAKEN_EXACT 1
#endif
#ifndef MM_WALK_TWICE
#define MM_WALK_TWICE 0
#endif
// Work of a 256-thread workgroup: FOUR tiles, one per wave (nothing shared), or ONE heavy tile walked by its four waves together
// (tile_walk_coop).  With the plan kernel's order (tiles of an image by decreasing candidate count, the first nheavy of them heavy):
// workgroup j of image b takes heavy tile j, or -- behind the heavy ones -- the four tiles nheavy + 4 (j - nheavy) + wave.  Launch
// order = heavy first, images interleaved (workgroup i -> image i % B): the kernel's duration is set by its slowest waves, so they
// must not start last.  Without an order (huge meshes / screens): the 2x2 tiles of a 16x16 pixel block.
//   valid: this wave has a tile;  coop: the workgroup's waves share it (wv = this wave's index among them).
//   kBlock = false: the one-wave workgroup variant of the same kernels (one tile per workgroup, never cooperative): a slow tile
//   then never pins the LDS and the wave slots of finished neighbours -- better where tiles are many and none is heavy.
//   limit: tiles [0, limit) of the order are walked by this mapping (the fused kernel shades the empty ones behind them four per wave)
//   rank: the workgroup's index among its image's walking workgroups if the caller has worked it out (the fused kernel interleaves them
//   with the workgroups that shade empty tiles), -1: straight from blockIdx
template <bool kBlock>
__device__ inline TileCtx make_tile(const RasterArgs& a, int wv, int rank, bool& valid, bool& coop, int limit) {
    TileCtx t;
    int blk;
    valid = true; coop = false;
    if (a.order) {
        const int nslot = 4 * a.blocks_per_image;
        int j;
        walk_image_rank((int)blockIdx.x, a.B, a.spread, t.b, j);
        if (rank >= 0) j = rank;
        const int nh = kBlock ? a.nheavy[4 * t.b] : 0;
        int idx;
        if (!kBlock) idx = j;
        else if (j < nh) { idx = j; coop = true; }
        else idx = nh + (j - nh) * 4 + wv;
        valid = idx < limit;
        const unsigned e = a.order[(size_t)t.b * nslot + (valid ? idx : 0)];
        const int slot = (int)(e & 0x7FFFu);
        t.empty = (e >> 15) != 0;                                // the plan kernel counted no candidate at all for this tile
        blk = slot >> 2; t.wave = slot & 3;
    } else {
        map_block(kBlock ? blockIdx.x : blockIdx.x >> 2, a.B, a.blocks_per_image, t.b, blk);
        t.wave = kBlock ? wv : (int)(blockIdx.x & 3);
        t.empty = false;
    }
    t.blk = blk;
    t.lane = threadIdx.x & 63;
    const int bx = blk % a.blocks_x, by = blk / a.blocks_x;
    const int tx0 = bx * MM_BLOCK_PX + (t.wave & 1) * MM_TILE, ty0 = by * MM_BLOCK_PX + (t.wave >> 1) * MM_TILE;
    t.tx0 = tx0; t.ty0 = ty0;
    tile_pixels(a, t);
    // tiles never straddle bins (bin edge is 8, 16 or 32); a tile fully outside the image borrows the last bin
    const int binx = min(tx0 >> a.bin_shift, a.nbx - 1), biny = min(ty0 >> a.bin_shift, a.nby - 1);
    const size_t mrow = ((size_t)t.b * a.nbx * a.nby + (size_t)biny * a.nbx + binx) * a.words;
    t.mask = a.binmask + mrow;
    return t;
}

// The bin's candidate bits -> ordered id lists in st->ids.  A CHUNK is 64 mask words (4096 faces), loaded with one instruction (lane =
// word); it is walked in WINDOWS of whole words holding at most MM_GROUP_WORDS * 64 ids -- nearly always the whole chunk at once, so
// a tile pays one trip to memory for its mask and its candidates fill whole batches whatever words they come from.  Ids are relative
// to the chunk's first face.
struct IdWindows { uint64_t w; int pc, pre, tot, first, done; };
__device__ inline uint64_t idw_load(const RasterArgs& a, const TileCtx& t, int cbase) { return (cbase + t.lane < a.words) ? t.mask[cbase + t.lane] : 0ull; }
__device__ inline void idw_begin(IdWindows& iw, const RasterArgs& a, const TileCtx& t, uint64_t word) {
    iw.w = word;
    iw.pc = __popcll(iw.w);
    iw.pre = wave_prefix_excl(iw.pc, t.lane, iw.tot);
    iw.first = 0; iw.done = 0;
}
// next window: number of ids now in st->ids (0: the chunk is exhausted).  Wave-uniform.
__device__ inline int idw_next(IdWindows& iw, const TileCtx& t, WaveStage* st) {
    if (iw.done >= iw.tot) return 0;
    const int end = iw.pre + iw.pc;
    const bool part = t.lane >= iw.first && end - iw.done <= MM_GROUP_WORDS * 64;   // whole words, contiguous from `first` (a word holds <= 64 ids)
    const int last = 63 - __clzll((unsigned long long)__ballot(part));
    const int wend = __builtin_amdgcn_readlane(end, last);         // (`last` is wave-uniform: a scalar lane select, not an LDS-crossbar shuffle)
    if (part) {
        uint64_t w = iw.w;
        int pos = iw.pre - iw.done;
        while (w) {                                              // <= 64 iterations
            const int bit = __ffsll((unsigned long long)w) - 1;
            w &= w - 1;
            st->ids[pos++] = (unsigned short)(t.lane * 64 + bit);
        }
    }
    const int n = wend - iw.done;
    iw.done = wend; iw.first = last + 1;
    return n;
}

#ifndef MM_HARD_ROW_MAX
#define MM_HARD_ROW_MAX 6
#endif
#ifndef MM_NEAR_PASSES
#define MM_NEAR_PASSES 2          // depth bands a flush's colour candidates are evaluated in, nearest first (1: one pass in index order)
#endif
#ifndef MM_NEAR_MIN_PAIRS
#define MM_NEAR_MIN_PAIRS 256     // ... if the flush has more (pixel, candidate) box pairs than this
#endif
// the two pixel masks of this lane's candidate (front-face box: colour; inflated box: silhouette), candidate-major
//   zfloor: the smallest depth_ord any in-image pixel of the tile holds (0 while one of them holds nothing): a front face whose depth bound
//   lies below it cannot win any pixel of the tile and is not a colour candidate at all;  zb: the face's depth bound (kept with the candidate)
__device__ inline void candidate_masks(const RasterArgs& a, const TileCtx& t, const float4& g0, const float4& g1, const float4& g2, bool soft, int bmode,
                                       unsigned zfloor, uint64_t& mh, uint64_t& ms, unsigned& zb) {
    const float xmin = fminf(fminf(g0.x, g0.z), g1.x), ymin = fminf(fminf(g0.y, g0.w), g1.y);
    const float xmax = fmaxf(fmaxf(g0.x, g0.z), g1.x), ymax = fmaxf(fmaxf(g0.y, g0.w), g1.y);
    // kaolin rasterises the faces with face_normals_z >= 0 (MM_OPT_CULL_STRICT: > 0); its soft mask looks at ALL faces
    // (MM_OPT_SOFT_SKIP_CULLED: only at those)  -- SURVEY Appendix C-1
    const bool front = (a.options & MM_OPT_CULL_STRICT) ? g2.y > 0.f : g2.y >= 0.f;
    zb = depth_bound(g1.z, g1.w, g2.x);
    box_masks(a, t, xmin, ymin, xmax, ymax, front && zb >= zfloor, soft && (front || !(a.options & MM_OPT_SOFT_SKIP_CULLED)), bmode, mh, ms);
}
__device__ inline void stage_slot(WaveStage* st, int slot, int f, const float4& g0, const float4& g1, const float4& g2, unsigned zb) {
    st->p0[slot] = g0; st->p1[slot] = g1;
    st->p2[slot] = make_float4(g2.x, g2.y, __int_as_float(f), __uint_as_float(zb));   // cz, nz, face id, depth bound
}

// Walk the bin's candidates in face order, 64 at a time (a RAW batch: whatever the bin's mask lists).  Per raw batch lane j tests
// candidate j's two boxes against the tile's 8 columns and 8 rows; candidates that touch no pixel of the tile -- most of them where the
// bin is larger than the tile, where the silhouette margin inflates the bin's catchment, or once no pixel can take another silhouette
// face -- are dropped on the spot, the others are BALLOT-COMPACTED (v_mbcnt rank = exclusive prefix over the ballot) behind the ones
// already waiting in the LDS queue, in face order.  flush(n) is called with n <= 64 queued candidates in st->p0/p1/p2 and their masks in
// st->qm whenever the next batch's survivors would not fit, and once at the end.  A raw batch costs its fetch + the box tests; staging
// and pair work are paid per SURVIVOR: a far-away mesh folded into a few tiles (thousands of candidates per bin, a handful of pixel
// hits per batch) costs a tenth of what it did when every batch was staged, transposed and paired.
//   want_soft(): wave-uniform, asked once per raw batch: can any pixel still take a silhouette face?  (Monotone: once false, always false.)
template <class WantSoft, class Flush>
__device__ inline void scan_candidates(const RasterArgs& a, const TileCtx& t, WaveStage* st, const unsigned& zfloor, int& cur_cbase, WantSoft&& want_soft, Flush&& flush MM_PP_ARG) {
    const float4* geo = a.geo + (size_t)t.b * a.F * 3;
    const int bmode = box_mode(a.options);
    int qn = 0;                                                  // candidates waiting in the queue (wave-uniform)
    uint64_t next_word = idw_load(a, t, 0);
    MM_PP_MARK(8);                                               // (phase build: the first mask load's latency alone)
    for (int cbase = 0; cbase < a.words; cbase += 64) {
      IdWindows iw;
      cur_cbase = cbase;
      idw_begin(iw, a, t, next_word);
      if (cbase + 64 < a.words) next_word = idw_load(a, t, cbase + 64);   // the next chunk's mask words travel while this chunk is walked (a mesh of
                                                                          // 13 776 faces has four chunks: three dependent trips to memory less per tile)
      for (bool more = true; more;) {
        const int total = idw_next(iw, t, st);                   // ids now in st->ids (0: the chunk is exhausted)
        more = total != 0 && iw.done < iw.tot;                   // another window of this chunk follows
        // the walk's LAST pass gets one more, empty batch: the final flush happens at the loop's only flush site (the flush is the bulk
        // of this kernel's code; expanded twice it no longer fits the instruction cache)
        const bool final = !more && cbase + 64 >= a.words;
        if (total == 0 && !final) break;
        const int wbase = cbase;
        wave_lds_sync();
        MM_PP_MARK(1);
        MM_PP_COUNT(total, 0);
        // the face records of batch k+1 are requested before batch k is tested (a tile with hundreds of candidates would otherwise pay a
        // dependent trip to memory per batch)
        float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f), n1 = n0, n2 = n0;
        int nf = 0;
        auto fetch = [&](int k0) {                               // (unconditional: lanes beyond the list read its first face and ignore it; every call
            nf = wbase * 64 + st->ids[k0 + t.lane < total ? k0 + t.lane : 0];     //  redefines all thirteen registers, so none is carried across a flush)
            n0 = geo[(size_t)nf * 3 + 0]; n1 = geo[(size_t)nf * 3 + 1]; n2 = geo[(size_t)nf * 3 + 2];
        };
        if (total) fetch(0);
        MM_PP_MARK(9);                                           // (phase build: a window's first record fetch, its latency alone)
        const int nbatch = (total + 63) / 64 + (final ? 1 : 0);
        for (int i = 0; i < nbatch; ++i) {
            const int k0 = i * 64, n = max(0, min(64, total - k0));
            float4 g0, g1, g2;
            int f, slot = 0, ns = 0;
            uint64_t mh = 0, ms = 0;                             // candidate-major: lane j = candidate j, bit p = pixel p
            unsigned zb = 0;
            bool keep = false;
            // (at most two passes: if this batch's survivors do not fit behind the queued ones, the queue is flushed and the batch is
            // taken again from its -- cache-hot -- records, instead of carrying seventeen registers per lane across the flush)
            for (int pass = 0; pass < 2; ++pass) {
                if (pass == 1) fetch(k0);
                g0 = n0; g1 = n1; g2 = n2; f = nf;
                if (k0 + 64 < total) fetch(k0 + 64);
                const bool soft = want_soft();
                mh = 0; ms = 0;
                if (t.lane < n) candidate_masks(a, t, g0, g1, g2, soft, bmode, zfloor, mh, ms, zb);
                keep = (mh | ms) != 0;
                const uint64_t surv = __ballot(keep);
                ns = __popcll(surv);
                slot = ballot_rank(surv);
                MM_PP_MARK(2);
                if (!(n == 0 ? qn > 0 : qn + ns > 64)) break;    // (wave-uniform) not the end of the walk, and the survivors fit
                flush(qn); qn = 0;
                if (n == 0) break;
            }
            if (keep) {
                stage_slot(st, qn + slot, f, g0, g1, g2, zb);
                st->qm[0][qn + slot] = mh; st->qm[1][qn + slot] = ms;
            }
            qn += ns;
        }
      }
      wave_lds_sync();
      if (cbase + 64 < a.words) flush_taken(a, t, st, cbase);    // (the LAST chunk's faces are written by the kernel after its own last load: flush_taken_last)
    }
}

__device__ inline void winner(const RasterArgs& a, const TileCtx& t, unsigned long long k, Hit& h) {
    h.f = -1; h.w0 = h.w1 = h.w2 = 0.f;
    if (k != 0ull) {                                             // barycentrics of the winner (same expressions, same values)
        h.f = depth_key_rank(k);
        const float4* geo = a.geo + ((size_t)t.b * a.F + h.f) * 3;
        const float4 p0 = geo[0], p1 = geo[1];
        float nrm;
        bary_weights(p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, t.x0, t.y0, a.eps, (a.options & MM_OPT_BARY_ONE_MINUS) != 0, h.w0, h.w1, h.w2, nrm);
    }
}


// ONE walk over the tile's candidates serves both rules.
// K1, nearest front face per pixel: kaolin walks faces in index order and keeps strict z > best, i.e. the winner is
// argmax over (z, -index); that maximum is taken with a 64-bit LDS atomic max per (pixel, face) pair, which is exact
// and order-free -- the pairs are taken straight from the candidate-major masks (no transpose: no order to keep).
// NaN and -inf depths never win, as in the reference.
// K3, soft silhouette of the pixels no front face covers: prod(1-p) over the pixel's first knum nearby faces is
// accumulated as an integer sum of log2(1-p) in 2^-32 fixed point (exact, commutative LDS adds) plus a count of exact
// zeros.  The "first knum in face order" rule needs the per-pixel view: one 64x64 bit transpose per flush, only while
// some pixel can still take a face.  A pixel takes silhouette faces while no face flushed SO FAR covers it: for a pixel
// that stays uncovered that is every candidate, in order -- exactly the two-pass result; whatever a pixel gathered
// before a later flush covered it is never looked at.
__device__ inline void tile_walk(const RasterArgs& a, const TileCtx& t, WaveStage* st, unsigned long long& key, SoftState& ss MM_PP_ARG) {
    key = 0ull;
    ss.qnz = 1.f; ss.zeros = 0; ss.lastf = 0x7FFFFFFF;
    if (t.empty) return;                                         // wave-uniform: more than half of all tiles are empty
    const float s2 = a.sigmainv / (a.mult * a.mult);
#if MM_WALK_TWICE
    // BOUND EXPERIMENT (profiles/r05_depth_order_bound.md; never in the product): the tile is walked twice by the same code (a loop, not a second
    // call site).  1: the second walk starts from the FIRST walk's winners -- every pixel already holds its final depth, the tile's depth floor is
    // final, only truly uncovered pixels take silhouette faces: what a perfect nearest-first order of the candidates could at best leave of the
    // colour pairs.  2: the second walk starts from nothing (calibration: the cost of one whole walk).  Results are those of the second walk,
    // which are the first's (the winner is an idempotent maximum; the silhouette state is reset).
    unsigned long long seed = 0ull;
    int lastf = 0x7FFFFFFF;
    for (int rep = 0; rep < 2; ++rep) {
    st->key[t.lane] = MM_WALK_TWICE == 1 ? seed : 0ull; st->logsum[t.lane] = 0ll; st->zeros[t.lane] = 0; st->takenw[t.lane] = 0ull;
    wave_lds_sync();
    int cnt = 0, cbase = 0;
    lastf = 0x7FFFFFFF;
    bool open = t.in_img && (MM_WALK_TWICE == 1 ? seed == 0ull : true);
    unsigned zfloor = MM_WALK_TWICE == 1 ? wave_min_u32(t.in_img ? (unsigned)(seed >> 32) : 0xFFFFFFFFu) : 0u;
#ifdef MM_PHASE_PROF
    if (rep == 1) { pp_.c0 = 0; pp_.c1 = 0; for (int i = 1; i <= 4; ++i) pp_.acc[i] = 0; }   // the counters and walk phases of the second walk alone
#endif
#else
    st->key[t.lane] = 0ull; st->logsum[t.lane] = 0ll; st->zeros[t.lane] = 0; st->takenw[t.lane] = 0ull;
    wave_lds_sync();
    int cnt = 0, lastf = 0x7FFFFFFF, cbase = 0;
    bool open = t.in_img;
    unsigned zfloor = 0;                                         // smallest depth_ord held by an in-image pixel of the tile (wave-uniform; 0: some pixel holds nothing)
#endif
    scan_candidates(a, t, st, zfloor, cbase, [&]() { return __ballot(open && cnt < a.knum) != 0; }, [&](int n) {
        wave_lds_sync();                                         // the queue's stores
        const uint64_t mh = t.lane < n ? st->qm[0][t.lane] : 0ull, ms = t.lane < n ? st->qm[1][t.lane] : 0ull;
#ifdef MM_PHASE_PROF
        { int th; (void)wave_prefix_excl(__popcll(mh), t.lane, th); MM_PP_COUNT(1ull << 32, (unsigned long long)th); }   // flushes | colour pairs
#endif
        if (__ballot(mh != 0)) {
            // NEAREST FIRST.  The winner is an order-free maximum, so the flush's candidates may be evaluated in any order -- and the order decides
            // how much early-z saves: in index order a pixel under forty overlapping boxes meets its nearest face at a random place of the list.
            // The queued candidates are therefore taken in MM_NEAR_PASSES bands of their depth bound, nearest band first (band edges evenly
            // spaced between the flush's smallest and largest bound: two wave reductions, no sort); after every band the tile's depth floor is
            // refreshed and the candidates of the farther bands that lie entirely behind what EVERY pixel already holds lose their masks before
            // a single pair of theirs is listed; the others meet tighter per-pixel depths in the pair filter.  (Strictly behind only: equal
            // depths, which the lower index wins, are never cut -- kaolin's "strict z > best in index order")
            const unsigned zbd = t.lane < n ? __float_as_uint(st->p2[t.lane].w) : 0u;
            int npass = 1;
            unsigned zlo = 0u, zhi = 0u;
#if MM_NEAR_PASSES > 1
            {
                int pairs_total;
                (void)wave_prefix_excl(__popcll(mh), t.lane, pairs_total);
                if (pairs_total > MM_NEAR_MIN_PAIRS) {            // (wave-uniform) a short list: one pass, as before
                    zhi = (unsigned)wave_max_i32((int)(zbd >> 1)) << 1;                         // (31-bit reductions: the bands need no last bit)
                    zlo = wave_min_u32(mh != 0 ? zbd : 0xFFFFFFFFu);
                    npass = zhi > zlo ? MM_NEAR_PASSES : 1;
                }
            }
#endif
            uint64_t left = mh;
            for (int pass = 0; pass < npass; ++pass) {
                // band `pass` holds the bounds in (edge[pass + 1], edge[pass]]: edge[0] = +inf, edge[npass] = -inf
                const unsigned lo_edge = pass + 1 < npass ? zhi - (unsigned)(((unsigned long long)(zhi - zlo) * (unsigned)(pass + 1)) / (unsigned)npass) : 0u;
                uint64_t m = (zbd >= lo_edge) ? left : 0ull;
                left &= ~m;
                if (pass > 0 && zbd < zfloor) m = 0ull;           // entirely behind the tile by now (zfloor: refreshed below)
                if (__ballot(m != 0)) {
                    // The pair list is written row by row, every lane its own row: that costs as many trips as the LONGEST row has bits.  Rows =
                    // candidates (the masks as they are) suit small faces (a few pixels each, many candidates); a close-up face covers the whole
                    // tile (64 bits) while a pixel lies in a handful of boxes: then rows = pixels, at the price of one bit transpose.
                    const bool by_cand = wave_max_i32(__popcll(m)) <= MM_HARD_ROW_MAX;     // (wave-uniform; one instantiation of the pair code for both)
                    hard_pairs(a, t, st, by_cand ? m : wave_transpose64(m, t.lane), by_cand);
                    const unsigned long long kk = st->key[t.lane];
                    open = t.in_img && kk == 0ull;
                    zfloor = wave_min_u32(t.in_img ? (unsigned)(kk >> 32) : 0xFFFFFFFFu);
                }
            }
            MM_PP_MARK(3);
        }
        const uint64_t openm = __ballot(open && cnt < a.knum);
        if (__ballot(ms != 0) && openm) {
            const uint64_t ps = wave_transpose64(ms, t.lane);    // pixel-major: this lane's pixel, bit j = queued candidate j
            const uint64_t sm = soft_take(ps, open, a.knum - cnt);   // the first knum hits of this pixel, in order
#if MM_TAKEN_EXACT
            mark_taken(a, t, st, st, (wave_or_u64(sm) >> t.lane) & 1ull, 1ull, cbase);   // exactly the candidates some pixel took (bit 0 = this lane's)
#else
            mark_taken(a, t, st, st, ms, openm, cbase);          // a superset: every candidate whose inflated box holds a pixel that is still taking
#endif
            cnt += __popcll(sm);
            if (sm != 0 && cnt >= a.knum) lastf = __float_as_int(st->p2[63 - __clzll((unsigned long long)sm)].z);   // knum-th face taken
            if (__ballot(sm != 0)) soft_pairs(a, t, st, sm, s2);
#ifdef MM_PHASE_PROF
            { int ts; (void)wave_prefix_excl(__popcll(sm), t.lane, ts); MM_PP_COUNT(0, (unsigned long long)ts << 32); }   // silhouette pairs
#endif
            MM_PP_MARK(4);
        }
        wave_lds_sync();                                         // the queue is free again
    } MM_PP_PASS);
    wave_lds_sync();
#if MM_WALK_TWICE
    seed = st->key[t.lane];
    wave_lds_sync();
    }
#endif
    key = st->key[t.lane];
    ss.zeros = st->zeros[t.lane];
    ss.qnz = exp2f((float)((double)st->logsum[t.lane] * (1.0 / 4294967296.0)));
    ss.lastf = lastf;
}

// The PER-BATCH walk, for 8-pixel screen bins (the bin IS the tile: nearly every candidate of the bin touches the tile, so there is nothing
// to compact, and with a few dozen candidates per tile the launch lasts as long as one tile's chain of dependent steps -- the queue, the
// filter pass and the face flags of tile_walk only lengthen it: 128x128 with 1 280 faces, raster_fwd 32.5 us against 40).  Every batch of 64
// candidates is staged, box-tested, transposed to the per-pixel view and its pairs evaluated at once; same results bit for bit.
__device__ inline void tile_walk_batch(const RasterArgs& a, const TileCtx& t, WaveStage* st, unsigned long long& key, SoftState& ss MM_PP_ARG) {
    key = 0ull;
    ss.qnz = 1.f; ss.zeros = 0; ss.lastf = 0x7FFFFFFF;
    if (t.empty) return;                                         // wave-uniform: more than half of all tiles are empty
    st->key[t.lane] = 0ull; st->logsum[t.lane] = 0ll; st->zeros[t.lane] = 0; st->takenw[t.lane] = 0ull;
    wave_lds_sync();
    const float s2 = a.sigmainv / (a.mult * a.mult);
    const float4* geo = a.geo + (size_t)t.b * a.F * 3;
    const int bmode = box_mode(a.options);
    int cnt = 0, lastf = 0x7FFFFFFF;
    bool open = t.in_img;
    unsigned zfloor = 0;
    uint64_t next_word = idw_load(a, t, 0);
    MM_PP_MARK(8);
    for (int cbase = 0; cbase < a.words; cbase += 64) {
      IdWindows iw;
      idw_begin(iw, a, t, next_word);
      if (cbase + 64 < a.words) next_word = idw_load(a, t, cbase + 64);
      for (int total = idw_next(iw, t, st); total != 0; total = idw_next(iw, t, st)) {
        const int wbase = cbase;
        wave_lds_sync();
        MM_PP_MARK(1);
        MM_PP_COUNT(total, 0);
        // the face records of batch k+1 are requested before batch k is evaluated
        float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f), n1 = n0, n2 = n0;
        int nf = 0;
        auto fetch = [&](int k0) {
            if (k0 + t.lane < total) {
                nf = wbase * 64 + st->ids[k0 + t.lane];
                n0 = geo[(size_t)nf * 3 + 0]; n1 = geo[(size_t)nf * 3 + 1]; n2 = geo[(size_t)nf * 3 + 2];
            }
        };
        fetch(0);
        MM_PP_MARK(9);
        for (int k0 = 0; k0 < total; k0 += 64) {
#ifdef MM_BOUND_CAP                                             // BOUND EXPERIMENT (WRONG results, never in the product; profiles/r06_semi_heavy_bound.md): a single-wave tile stops after
            if (k0 >= MM_BOUND_CAP) break;                       // this many candidates -- what the launch would last if no single-wave tile were heavier than that
#endif
            const int n = min(64, total - k0);
            const float4 g0 = n0, g1 = n1, g2 = n2;
            const int f = nf;
            if (k0 + 64 < total) fetch(k0 + 64);
            const bool soft = __ballot(open && cnt < a.knum) != 0;
            uint64_t mh = 0, ms = 0;                             // candidate-major: lane j = candidate j, bit p = pixel p
            if (t.lane < n) {
                unsigned zb;
                candidate_masks(a, t, g0, g1, g2, soft, bmode, zfloor, mh, ms, zb);
                stage_slot(st, t.lane, f, g0, g1, g2, zb);
            }
            const uint64_t ph = __ballot(mh != 0) ? wave_transpose64(mh, t.lane) : 0ull;
            const uint64_t ps = __ballot(ms != 0) ? wave_transpose64(ms, t.lane) : 0ull;
            wave_lds_sync();
            MM_PP_MARK(2);
            if (__ballot(ph != 0)) {
#ifdef MM_BATCH_FILTER                                          // (A/B, profiles/r05_batch_walk_ab.md: the compacting walk's two-phase pair evaluation -- packed sign filter +
                hard_pairs(a, t, st, ph, false);                 //  early-z, then the divisions for the survivors -- in the per-batch walk; lists of <= MM_HARD_DIRECT pairs go direct)
#else
                pair_parallel(t, st, ph, [&](int l, int j, bool live) { hard_pair(a, t, st, st, j, l, live); });   // pixel l, candidate j
#endif
                const unsigned long long kk = st->key[t.lane];
                open = t.in_img && kk == 0ull;
                zfloor = wave_min_u32(t.in_img ? (unsigned)(kk >> 32) : 0xFFFFFFFFu);
                MM_PP_MARK(3);
            }
            const uint64_t sm = soft_take(ps, open, a.knum - cnt);   // the first knum hits of this pixel, in order
            cnt += __popcll(sm);
            if (sm != 0 && cnt >= a.knum) lastf = __float_as_int(st->p2[63 - __clzll((unsigned long long)sm)].z);   // knum-th face taken
            // r06 A/B (-DMM_BATCH_FLAGS=1; a.fflag is null otherwise): the face flags for the backward here too, as the compacting walk sets them.
            // Costs this kernel more than it saves the gather at 128x128 (profiles/r06_batch_walk_flags_ab.md): off by default.
#if MM_BATCH_FLAGS
            if (a.fflag && __ballot(sm != 0)) mark_taken(a, t, st, st, (wave_or_u64(sm) >> t.lane) & 1ull, 1ull, cbase);
#endif
#ifdef MM_BATCH_SOFT_PACKED                                             // (two pairs per lane in packed fp32, as the compacting walk does: measured SLOWER
            if (__ballot(sm != 0)) soft_pairs(a, t, st, sm, s2);         //  here, raster_fwd 34.4 / 92.8 / 191.7 us against 33.8 / 88.4 / 184.7 at 128x128 / 256x256 / B=384)
#else
            if (__ballot(sm != 0)) pair_parallel(t, st, sm, [&](int l, int j, bool live) { soft_pair(a, t, st, st, s2, l, j, live); });
#endif
            MM_PP_MARK(4);
            wave_lds_sync();
        }
      }
#if MM_BATCH_FLAGS
      if (cbase + 64 < a.words) { wave_lds_sync(); flush_taken(a, t, st, cbase); }   // (meshes beyond 4096 faces; the LAST chunk's faces: flush_taken_last)
#endif
    }
    wave_lds_sync();
    key = st->key[t.lane];
    ss.zeros = st->zeros[t.lane];
    ss.qnz = exp2f((float)((double)st->logsum[t.lane] * (1.0 / 4294967296.0)));
    ss.lastf = lastf;
}

// A HEAVY tile (hundreds of candidates: a far-away mesh folded into a few tiles) walked by the four waves of its workgroup.  Alone,
// its wave would be the kernel's tail: one wave issues a vector instruction every ~5 cycles at best, and the tile's pair evaluations
// (up to 64 x knum silhouette pairs, most of them in its first batches) added up to 40 us while the chip drained.  Here the batches
// of a ROUND of four are staged by the four waves (one each), and the round's PAIRS -- whichever batch they come from -- are dealt
// evenly to all 256 lanes: every pair is (staging wave, candidate, pixel), colour pairs first, and any lane can evaluate any of them
// because the four stages live in the workgroup's LDS and the results are combined by exact, commutative LDS atomics in stage[0].
// The silhouette's "first knum faces in index order" rule is kept exactly: every wave publishes its batch's per-pixel inflated-box
// hit counts, and a batch takes what is left of knum after the batches before it -- for a pixel that stays uncovered that is the
// sequential result bit for bit (what a pixel gathered before it was covered is never read, as in tile_walk).
#define MM_COOP_WINDOW (4 * MM_PAIR_ROUND)    // pairs dealt per pass: the four stages' pair buffers side by side
__device__ inline void tile_walk_coop(const RasterArgs& a, const TileCtx& t, WaveStage* stage, int wv, unsigned long long& key, SoftState& ss MM_PP_ARG) {
    WaveStage* st = &stage[wv];
    WaveStage* acc = &stage[0];
    key = 0ull;
    ss.qnz = 1.f; ss.zeros = 0; ss.lastf = 0x7FFFFFFF;
    if (wv == 0) { acc->key[t.lane] = 0ull; acc->logsum[t.lane] = 0ll; acc->zeros[t.lane] = 0; acc->takenw[t.lane] = 0ull; coop_lastf(acc)[t.lane] = 0x7FFFFFFF; }
    const int bmode = box_mode(a.options);
    __syncthreads();
    const float s2 = a.sigmainv / (a.mult * a.mult);
    const float4* geo = a.geo + (size_t)t.b * a.F * 3;
    int base_cnt = 0;                                            // this pixel's inflated-box hits in the rounds so far
    bool open = t.in_img;
    for (int cbase = 0; cbase < a.words; cbase += 64) {
      IdWindows iw;
      idw_begin(iw, a, t, idw_load(a, t, cbase));                // every wave expands the same lists into its own stage
      for (int total = idw_next(iw, t, st); total != 0; total = idw_next(iw, t, st)) {   // (the same in the four waves)
        const int wbase = cbase;
        wave_lds_sync();
        MM_PP_MARK(1);
        MM_PP_COUNT(total, 0);
        for (int r0 = 0; r0 < total; r0 += 4 * 64) {
            const int k0 = r0 + wv * 64;
            const int n = max(0, min(64, total - k0));
            const bool soft = __ballot(open && base_cnt < a.knum) != 0;      // the same in the four waves
            uint64_t mh = 0, ms = 0;
            if (t.lane < n) {
                const int f = wbase * 64 + st->ids[k0 + t.lane];
                const float4 g0 = geo[(size_t)f * 3 + 0], g1 = geo[(size_t)f * 3 + 1], g2 = geo[(size_t)f * 3 + 2];
                unsigned zb;
                candidate_masks(a, t, g0, g1, g2, soft, bmode, 0u, mh, ms, zb);
                stage_slot(st, t.lane, f, g0, g1, g2, zb);
            }
            const uint64_t ph = __ballot(mh != 0) ? wave_transpose64(mh, t.lane) : 0ull;
            const uint64_t ps = __ballot(ms != 0) ? wave_transpose64(ms, t.lane) : 0ull;
            mark_taken(a, t, st, acc, ms, __ballot(open && base_cnt < a.knum), cbase);
            coop_cnt(st)[t.lane] = __popcll(ps);
            MM_PP_MARK(2);
            __syncthreads();
            int before = base_cnt, round_total = 0;              // hits of the batches before this wave's, in index order
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) { const int c = coop_cnt(&stage[w2])[t.lane]; before += w2 < wv ? c : 0; round_total += c; }
            const uint64_t sm = soft_take(ps, open, a.knum - before);
            if (sm != 0 && before + __popcll(sm) >= a.knum) coop_lastf(acc)[t.lane] = __float_as_int(st->p2[63 - __clzll((unsigned long long)sm)].z);
            base_cnt += round_total;
            // this wave's pairs of the round: ph (colour) and sm (silhouette), pixel-major; their positions in the round's list
            int nh, ns;
            int kh = wave_prefix_excl(__popcll(ph), t.lane, nh);
            int ks = wave_prefix_excl(__popcll(sm), t.lane, ns);
            if (t.lane == 0) { st->npair[0] = nh; st->npair[1] = ns; }
            __syncthreads();
            int TH = 0, TS = 0;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) {
                const int c0 = stage[w2].npair[0], c1 = stage[w2].npair[1];
                if (w2 < wv) { kh += c0; ks += c1; }
                TH += c0; TS += c1;
            }
            ks += TH;                                            // silhouette pairs behind all colour pairs
            const int T = TH + TS;
            MM_PP_MARK(6);
            MM_PP_COUNT(0, T);
            uint64_t remh = ph, rems = sm;
            for (int base = 0; base < T; base += MM_COOP_WINDOW) {
                const int lim = min(MM_COOP_WINDOW, T - base);
                while (remh && kh < base + lim) {                // every set bit is visited exactly once overall
                    const int j = __ffsll((unsigned long long)remh) - 1;
                    remh &= remh - 1;
                    const int q = kh - base;
                    stage[q / MM_PAIR_ROUND].pairs[q % MM_PAIR_ROUND] = (unsigned short)((wv << 12) | (j << 6) | t.lane);
                    ++kh;
                }
                while (rems && ks < base + lim) {
                    const int j = __ffsll((unsigned long long)rems) - 1;
                    rems &= rems - 1;
                    const int q = ks - base;
                    stage[q / MM_PAIR_ROUND].pairs[q % MM_PAIR_ROUND] = (unsigned short)(0x4000 | (wv << 12) | (j << 6) | t.lane);
                    ++ks;
                }
                __syncthreads();
                MM_PP_MARK(3);
                for (int q = threadIdx.x; q < lim; q += 256) {
                    const unsigned pr = stage[q / MM_PAIR_ROUND].pairs[q % MM_PAIR_ROUND];
                    WaveStage* src = &stage[(pr >> 12) & 3];
                    const int j = (pr >> 6) & 63, l = pr & 63;
                    if (pr & 0x4000u) soft_pair(a, t, src, acc, s2, l, j, true);
                    else hard_pair(a, t, src, acc, j, l, true);
                }
                MM_PP_MARK(4);
                __syncthreads();
                MM_PP_MARK(7);
            }
            open = t.in_img && acc->key[t.lane] == 0ull;
        }
      }
      if (cbase + 64 < a.words) {                                // (the last chunk's faces: flush_taken_last, by the wave that shades the tile)
          __syncthreads();                                       // the chunk's silhouette faces are all noted
          if (wv == 0) flush_taken(a, t, acc, cbase);
          __syncthreads();
      }
    }
    __syncthreads();
    if (wv != 0) return;
    key = acc->key[t.lane];
    ss.zeros = acc->zeros[t.lane];
    ss.qnz = exp2f((float)((double)acc->logsum[t.lane] * (1.0 / 4294967296.0)));
    ss.lastf = coop_lastf(acc)[t.lane];
}

}  // namespace mm
