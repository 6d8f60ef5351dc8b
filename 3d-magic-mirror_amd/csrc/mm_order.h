// mm_order.h -- the tile sort of the forward walk (gfx950), used by the order kernel (mm_raster.hip).  (A per-image vertex stage that sorted its
// own image was built in round 4, measured slower and removed: profiles/r04_per_image_stages_ab.md; the sort stayed in this header.)
//
// Orders the tile slots (16x16 block * 4 + quadrant) of ONE image by their candidate count, descending: a counting sort in LDS (keys clipped
// to 1023), linear in the slots.  Only the launch ORDER of raster_fwd depends on it -- slots with equal counts may come out in any order,
// no result does.  Bit 15 of an entry marks a tile that no face can touch (count 0), which raster_fwd then never walks.
#pragma once
#include "mm_device.h"

namespace mm {

#ifndef MM_HEAVY_CAND
#define MM_HEAVY_CAND 192     // a tile with at least this many candidates (three batches) is walked by four waves together ...
#endif
#ifndef MM_HEAVY_MAX
#define MM_HEAVY_MAX 32       // ... if it is among the image's MM_HEAVY_MAX heaviest
#endif

// s_key: the slots' clipped counts (written by the caller's threads before the call, one __syncthreads behind them is taken here);
// s_start: 1024 ints (histogram, then the first output position of every key), zeroed by the caller before the counts were added;
// s_wave: NT / 64 ints.  NT threads, all of which must call.  order: the image's row (nslot entries); nheavy: the image's pair.
template <int NT, int GROUP>
__device__ inline void tile_sort_scatter(int nslot, const unsigned short* s_key, int* s_start, int* s_wave, unsigned short* order, int* nheavy) {
    static_assert(NT == 256 || NT == 1024, "keys per thread below");
    constexpr int KPT = 1024 / NT;                                // keys per thread: thread t owns keys 1023 - KPT t .. 1024 - KPT (t + 1), descending
    const int tid = threadIdx.x;
    __syncthreads();
    int h[KPT], mine = 0;
#pragma unroll
    for (int j = 0; j < KPT; ++j) { h[j] = s_start[1023 - (KPT * tid + j)]; mine += h[j]; }
    int wtot;
    int before = wave_prefix_excl(mine, tid & 63, wtot);
    if ((tid & 63) == 63) s_wave[tid >> 6] = wtot;
    __syncthreads();
    for (int w = 0; w < (tid >> 6); ++w) before += s_wave[w];
#pragma unroll
    for (int j = 0; j < KPT; ++j) { s_start[1023 - (KPT * tid + j)] = before; before += h[j]; }
    __syncthreads();
    // tiles with at least MM_HEAVY_CAND candidates come first: the slots in front of key MM_HEAVY_CAND - 1
    if (tid == 0) { nheavy[0] = min(s_start[MM_HEAVY_CAND - 1], MM_HEAVY_MAX); nheavy[1] = s_start[0]; }   // slots in front of key 0: not empty
    __syncthreads();
    // GROUP consecutive slots (the tiles of a block: equal keys, set by the caller) take GROUP consecutive places
    for (int g = tid; g * GROUP < nslot; g += NT) {
        const int slot = g * GROUP, key = s_key[slot];
        const int pos = atomicAdd(&s_start[key], GROUP);
#pragma unroll
        for (int q = 0; q < GROUP; ++q) order[pos + q] = (unsigned short)((slot + q) | (key == 0 ? 0x8000 : 0));
    }
}

}  // namespace mm
