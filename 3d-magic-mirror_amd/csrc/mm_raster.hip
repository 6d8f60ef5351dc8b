// mm_raster.hip -- pixel stage of the render path's FORWARD for gfx950: the fused walk + shade kernel and the tile sort (the backward is
// mm_pixel_bwd.hip + mm_backward.hip).
//
// Replaces, fused into one launch per direction, what the reference reaches through kaolin for every pixel
// (call sites /root/reference/networks.py:297-317; semantics SURVEY.md 8(a) rows a8-a11, gradients Appendix A):
//   packed_rasterize_forward  (K1)  nearest front-facing face per pixel, barycentric interpolation
//   dibr_soft_mask_forward    (K3)  1 - prod(1 - exp(-sigma d^2)) over the first <= knum nearby faces
//   texture_mapping / grid_sample, spherical_harmonic_lighting, composite, clamp, cat   [+ the forward of recon_data when fused]
//
// Design (not kaolin's pixel-major brute force over all faces):
//   * a wave owns an 8x8 pixel tile, one lane per pixel; tiles are taken in the order the sort kernel left (most candidates first: a launch
//     lasts as long as its slowest tile), heavy tiles by the four waves of a workgroup together, empty tiles four per wave.
//   * the vertex stage left, per screen bin, a bit-per-face mask of the faces whose pixel box, inflated by the silhouette
//     margin, touches it.  The wave loads its bin's mask words coalesced, turns the set bits into an ORDERED candidate
//     list with popcount + wave prefix sum (face order = bit order, which the soft mask's "first knum faces" rule needs)
//     and stages 64 candidates at a time in LDS (struct-of-arrays float4 rows) -- ONE walk serves colour and silhouette.
//   * per batch, lane j tests candidate j's box (front faces) and inflated box (all faces) against the tile's 8 pixel
//     columns and 8 rows (separable closed-box test, the same float comparisons as a per-pixel test) -> two 64-bit pixel
//     masks per candidate; a 6-stage wave butterfly transposes each 64x64 bit matrix to the per-pixel view.
//   * the (pixel, candidate) pairs of the batch are then evaluated 64 at a time by whichever lane and combined with
//     exact, commutative LDS atomics: 64-bit max of (orderable z, ~face id) for colour -- argmax over (z, -index) is
//     exactly kaolin's "strict z > best in index order" -- and an integer sum of log2(1-p) for the silhouette.  A wave's
//     critical path is pairs/64 evaluations, not its busiest pixel, and results do not depend on evaluation order.
//   * the epilogue (winner -> uv -> texels -> shade -> store) is written as TWO dependent trips to memory (mm_raster_common.h: shade_store).
#include "mm_raster_walk.h"
#include "mm_order.h"

MM_TIMELINE_STORAGE(raster_fwd)
MM_PP_STORAGE(raster_fwd)       // 0 tile setup, 1 mask -> id list, 2 fetch + stage + box tests + transposes, 3 colour pairs, 4 silhouette pairs, 5 shade + store, 6 / 7 coop barriers, 8 first mask load (latency alone), 9 first record fetch of a window (latency alone)

namespace mm {

// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
#ifndef MM_RASTER_LB
#define MM_RASTER_LB 5
#endif
#ifndef MM_RASTER_WPE
#define MM_RASTER_WPE 5
#endif
// kQueue: the compacting walk (tile_walk) for screen bins larger than a tile; otherwise the per-batch walk (tile_walk_batch)
// kContour: the fused loss carries recon_data's contour term (host: fused_gt && fused_contour > 0)
template <bool kNoMask, bool kBlock, bool kQueue, bool kContour>
__global__ __launch_bounds__(kBlock ? 256 : 64) __attribute__((amdgpu_waves_per_eu(MM_RASTER_WPE, MM_RASTER_WPE))) void raster_fwd_kernel(RasterArgs a_) {   // kBlock: 5 waves per SIMD = 96 VGPRs, 5 x 32 KiB LDS per CU
#if defined(MM_ARGS_BY_VALUE) || !defined(__HIP_DEVICE_COMPILE__)
    const RasterArgs& a = a_;
#else
    // The arguments are READ FROM THE KERNARG SEGMENT WHERE THEY ARE USED (scalar loads from constant memory, through the scalar cache) instead of
    // being loaded into ~75 scalar registers at the kernel's entry and kept alive across the whole walk: that is what spilled 35-79 of them into
    // vector-register lanes (v_writelane / v_readlane on an issue-bound kernel; verdict r05 item 3).
    (void)a_;
    const RasterArgs& a = *(const RasterArgs*)__builtin_amdgcn_kernarg_segment_ptr();   // (the kernel's only parameter: the segment starts with it)
#endif
    MM_TIMELINE_BEGIN();
    __shared__ WaveStage s_stage[kBlock ? 4 : 1];
    MM_PP_BEGIN();
    const int wv = kBlock ? threadIdx.x >> 6 : 0;               // (MM_WAVE_UNIFORM here and on the order entry: 83 instead of 96 VGPRs, but 1-3 % SLOWER at every size)
    int limit = 4 * a.blocks_per_image, rank = -1;               // rank: this workgroup's index among its image's walking workgroups (-1: from blockIdx)
    if (a.order) {
        // workgroups of image b, in launch order: heavy tiles (one each), the other non-empty tiles (four each, or one), then the empty
        // tiles four per WAVE (shade_empty_tiles); the grid is sized for "no tile is empty", workgroups behind the last one exit
        int b, j;
        walk_image_rank((int)blockIdx.x, a.B, a.spread, b, j);
        const int nh = kBlock ? a.nheavy[4 * b] : 0, nne = a.nheavy[4 * b + 1];
        const int W1 = kBlock ? nh + (max(nne - nh, 0) + 3) / 4 : nne;       // workgroups that walk: heavy tiles one each, the others four each (or one)
        const int per = kBlock ? 16 : 4, W2 = (4 * a.blocks_per_image - nne + per - 1) / per;   // workgroups that shade empty tiles
        limit = nne;
        if (j >= W1) {                                           // (interleaving the two kinds of workgroup evenly was measured: no gain at 512x512,
            if (j - W1 >= W2) return;                            //  slower at 128x128, where every walking workgroup is resident from the start)
            const int e0 = nne + (j - W1) * per + wv * 4, ne = min(4, 4 * a.blocks_per_image - e0);
            if (ne > 0) shade_empty_tiles<kNoMask, kContour>(a, b, e0, ne, threadIdx.x & 63);
            return;
        }
        rank = j;
    }
    bool valid, coop;
    const TileCtx t = make_tile<kBlock>(a, wv, rank, valid, coop, limit);    // coop is workgroup-uniform; !valid only in the last workgroup of the non-empty tiles
    unsigned long long key;
    SoftState ss;
    MM_PP_MARK(0);
    if (kBlock && coop) {
        tile_walk_coop(a, t, s_stage, wv, key, ss MM_PP_PASS);
        if (wv != 0) return;                                     // the tile's pixels are shaded once
    } else {
        if (!valid) return;
        if (kQueue) tile_walk(a, t, &s_stage[wv], key, ss MM_PP_PASS);
        else tile_walk_batch(a, t, &s_stage[wv], key, ss MM_PP_PASS);
    }
    shade_store<kNoMask, kContour>(a, t, key, ss);
    flush_taken_last(a, t, &s_stage[(kBlock && coop) ? 0 : wv]);
    MM_PP_MARK(5);
    MM_PP_FLUSH(raster_fwd, (long long)blockIdx.x * (kBlock ? 4 : 1) + wv);
    MM_TIMELINE_END(raster_fwd);
}

// Candidates per screen bin for big screens / meshes (the order kernel below counts the mask bits itself where a tile's mask row is a few
// words): one WAVE per bin row, lanes = words (coalesced), popcount + wave sum.  13 776 faces at 512x512: 16 384 rows of 216 words.
__global__ __launch_bounds__(256) void bincount_kernel(const uint64_t* binmask, int* bincount, int nrows, int words) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= nrows) return;
    const uint64_t* m = binmask + (size_t)row * words;
    int c = 0;
    for (int w0 = 0; w0 < words; w0 += 256) {                     // four loads in flight
        uint64_t r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = (w0 + 64 * k + lane < words) ? m[w0 + 64 * k + lane] : 0ull;
#pragma unroll
        for (int k = 0; k < 4; ++k) c += __popcll(r[k]);
    }
    int total;
    (void)wave_prefix_excl(c, lane, total);
    if (lane == 0) bincount[row] = total;
}

// Orders the tile slots (16x16 block * 4 + quadrant) of every image by their soft-mask candidate count, descending: a counting
// sort in LDS (keys clipped to 1023; slots per image <= MM_ORDER_MAX_SLOTS), linear in the slots.  Only the launch ORDER of raster_fwd
// depends on it -- slots with equal counts may come out in any order, no result does.  Bit 15 of an entry marks a tile that no
// face can touch (count 0), which raster_fwd then never walks.
#define MM_ORDER_MAX_SLOTS 16384      // 1024x1024 pixels; bigger screens are walked in natural order
__global__ __launch_bounds__(256) void order_kernel(RasterArgs a, unsigned short* order, int* nheavy) {
    __shared__ unsigned short s_key[MM_ORDER_MAX_SLOTS];
    __shared__ int s_start[1024];          // histogram, then the first output position of every key
    __shared__ int s_wave[4];
    const int b = blockIdx.x, tid = threadIdx.x, nslot = 4 * a.blocks_per_image;
    if (!order) return;
    for (int i = tid; i < 1024; i += 256) s_start[i] = 0;
    __syncthreads();
    for (int slot = tid; slot < nslot; slot += 256) {
        const int blk = slot >> 2, q = slot & 3;
        const int tx0 = (blk % a.blocks_x) * MM_BLOCK_PX + (q & 1) * MM_TILE, ty0 = (blk / a.blocks_x) * MM_BLOCK_PX + (q >> 1) * MM_TILE;
        int c = 0;
        if (tx0 < a.W && ty0 < a.H) {
            const size_t bin = (size_t)b * a.nbx * a.nby + (size_t)(ty0 >> a.bin_shift) * a.nbx + (tx0 >> a.bin_shift);
            if (a.bincount) c = a.bincount[bin];
            else {
                const uint64_t* row = a.binmask + bin * a.words;
                for (int w0 = 0; w0 < a.words; w0 += 24) {        // 24 loads in flight per trip: the 1 280-face templates' rows in ONE trip
                    uint64_t r[24];
#pragma unroll
                    for (int k = 0; k < 24; ++k) r[k] = (w0 + k < a.words) ? row[w0 + k] : 0ull;
#pragma unroll
                    for (int k = 0; k < 24; ++k) c += __popcll(r[k]);
                }
            }
        }
        c = min(c, 1023);
        s_key[slot] = (unsigned short)c;
        if (!a.block_sort) atomicAdd(&s_start[c], 1);
    }
    if (a.block_sort) {
        // Screen bins of 16 pixels or more: the four tiles of a 16x16 block lie in ONE bin and walk the same candidate list.  The BLOCKS are
        // sorted (key = the bin's count) and a block's four tiles stay together in the order, so that raster_fwd can hand them to four
        // workgroups of one XCD at the same time (walk_image_rank, spread == 2): the bin's mask row and its candidates' records are then
        // fetched from memory once and found in that XCD's L2 by the other three.
        __syncthreads();
        for (int blk = tid; blk < a.blocks_per_image; blk += 256) {
            const int c = max(max(s_key[4 * blk], s_key[4 * blk + 1]), max(s_key[4 * blk + 2], s_key[4 * blk + 3]));
            s_key[4 * blk] = s_key[4 * blk + 1] = s_key[4 * blk + 2] = s_key[4 * blk + 3] = (unsigned short)c;
            atomicAdd(&s_start[c], 4);
        }
        tile_sort_scatter<256, 4>(nslot, s_key, s_start, s_wave, order + (size_t)b * nslot, nheavy + 4 * b);
    } else
        tile_sort_scatter<256, 1>(nslot, s_key, s_start, s_wave, order + (size_t)b * nslot, nheavy + 4 * b);
}

RasterArgs make_raster_args(const MMRenderDesc* d, const Workspace& w) {
    RasterArgs a;
    a.B = d->B; a.H = d->H; a.W = d->W; a.F = d->F; a.Ht = d->Ht; a.Wt = d->Wt; a.knum = d->knum;
    a.blocks_x = (d->W + MM_BLOCK_PX - 1) / MM_BLOCK_PX;
    a.blocks_per_image = w.blocks_per_image;
    a.bin_shift = w.bin_shift; a.nbx = w.nbx; a.nby = w.nby; a.words = w.words;
    a.mult = d->multiplier; a.eps = d->eps; a.sigmainv = d->sigmainv; a.infl = d->boxlen * d->multiplier;
    a.kx = d->multiplier / (float)d->W; a.ky = d->multiplier / (float)d->H;
    a.bincount = nullptr; a.spread = 0;
    a.geo = w.geo; a.binmask = w.binmask; a.soft = w.soft; a.fflag = w.fflag; a.gt = d->fused_gt; a.ltot = w.ltot; a.contour = d->fused_gt ? d->fused_contour : 0.f;
    a.trcnt = w.trcnt; a.ntx_tex = (d->Wt + MM_UV_TILE - 1) / MM_UV_TILE; a.ntiles_tex = w.ntiles;
    a.face_uvs = d->face_uvs; a.fn = d->face_normals; a.textures = d->textures; a.lights = d->lights; a.bg = d->bg;
    a.rgba = d->rgba; a.face_idx = d->face_idx; a.imnormal = d->imnormal;
    a.order = nullptr;
    a.nheavy = nullptr;
    a.block_sort = 0;
    a.feats = nullptr; a.D = 0; a.interp = nullptr; a.soft_out = nullptr; a.face_idx64 = nullptr; a.options = d->options;
    return a;
}

const unsigned short* launch_order(RasterArgs& a, unsigned short* order, int* nheavy, int* bincount, int B, void** prof_events, hipStream_t s) {
    const int nslot = 4 * a.blocks_per_image;
    a.order = nullptr; a.bincount = nullptr;
    a.block_sort = walk_block_sort(a) ? 1 : 0;
    if (nslot > MM_ORDER_MAX_SLOTS || nslot > 0x7FFF) return nullptr;          // (entries are 15 bits + the empty flag)
    ProfScope po(prof_events, MM_PROF_ORDER, s);
    if (nslot > 1024 || a.words > 64) {                          // big screen / mesh: the bins' candidate counts by a parallel kernel first
        const int nrows = B * a.nbx * a.nby;
        hipLaunchKernelGGL(bincount_kernel, dim3((nrows + 3) / 4), dim3(256), 0, s, a.binmask, bincount, nrows, a.words);
        a.bincount = bincount;
    }
    hipLaunchKernelGGL(order_kernel, dim3(B), dim3(256), 0, s, a, order, nheavy);
    return order;
}

int launch_raster_fwd(const MMRenderDesc* d, const Workspace& w, hipStream_t s) {
    RasterArgs a = make_raster_args(d, w);
    a.order = launch_order(a, w.order, w.nheavy, w.bincount, d->B, d->prof_events, s);     // heavy-first launch order
    a.spread = walk_spread(a);
    a.nheavy = w.nheavy;
    const bool block = walk_block_mode(a);
    const dim3 grid(walk_grid(a, block));
    ProfScope ps(d->prof_events, MM_PROF_RASTER_FWD, s);
    // 8-pixel bins: the bin is the tile, nothing to compact -> the per-batch walk, no face flags (every face gets its sweep items;
    // -DMM_BATCH_FLAGS=1: the r06 A/B in which this walk sets them too)
    const bool queue = walk_queue_mode(a);
    if (!queue && !MM_BATCH_FLAGS) a.fflag = nullptr;
#define MM_LAUNCH_RASTER2(NM, BL, QU, CO) hipLaunchKernelGGL((raster_fwd_kernel<NM, BL, QU, CO>), grid, dim3(BL ? 256 : 64), 0, s, a)
#define MM_LAUNCH_RASTER(NM, BL, QU) do { if (a.contour > 0.f) MM_LAUNCH_RASTER2(NM, BL, QU, true); else MM_LAUNCH_RASTER2(NM, BL, QU, false); } while (0)
    if (block) {
        if (queue) { if (d->no_mask) MM_LAUNCH_RASTER(true, true, true); else MM_LAUNCH_RASTER(false, true, true); }
        else { if (d->no_mask) MM_LAUNCH_RASTER(true, true, false); else MM_LAUNCH_RASTER(false, true, false); }
    } else {
        if (queue) { if (d->no_mask) MM_LAUNCH_RASTER(true, false, true); else MM_LAUNCH_RASTER(false, false, true); }
        else { if (d->no_mask) MM_LAUNCH_RASTER(true, false, false); else MM_LAUNCH_RASTER(false, false, false); }
    }
#undef MM_LAUNCH_RASTER2
#undef MM_LAUNCH_RASTER
    return launch_ok("raster_fwd");
}

}  // namespace mm
