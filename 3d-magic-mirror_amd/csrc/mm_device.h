// mm_device.h -- device-side helpers shared by the gfx950 kernels of libmm_render.so.
//
// Exactness contract: everything that decides WHICH face wins a pixel (camera, vertex transform, projection, bbox,
// edge functions, normalisation, z interpolation) is written as explicit fp32 expressions in a fixed order and the
// library is compiled with -ffp-contract=off, IEEE-rounded '/' and sqrtf (hipcc's default for HIP).  The CPU oracle
// (oracle/mm_oracle.inc) evaluates the same expressions in the same order, which is what makes face_idx comparable
// bit for bit.  Semantics: SURVEY.md section 8(a); reference call sites /root/reference/networks.py:258-324.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mm_render.h"

#define MM_WAVE 64
// MM_FP_EXACT at the head of a block: every fp32 expression inside rounds as written, whatever the translation unit's flags (the relaxed backward
// includes these functions too).  -DMM_ALLOW_CONTRACT (bound experiments only, wrong bits): the block follows the translation unit's flags.
#ifdef MM_ALLOW_CONTRACT
#define MM_FP_EXACT
#else
#define MM_FP_EXACT _Pragma("clang fp contract(off)")
#endif
// a value that is the same in all lanes of the wave BY CONSTRUCTION (derived from threadIdx.x >> 6 and the like), said so: the compiler then
// keeps it in a scalar register, and what is addressed by it becomes scalar arithmetic and scalar-cache loads.  -DMM_NO_SCALAR_WAVE: A/B switch.
#ifdef MM_NO_SCALAR_WAVE
#define MM_WAVE_UNIFORM(x) (x)
#else
#define MM_WAVE_UNIFORM(x) __builtin_amdgcn_readfirstlane((int)(x))
#endif
#define MM_TILE 8             // a wave owns an 8x8 pixel tile, one lane per pixel
#define MM_BLOCK_PX 16        // a 256-thread workgroup renders a 16x16 pixel block: 2x2 wave tiles
#define MM_BLOCK_WAVES 4
#define MM_LSUB 8             // sub-accumulators per image for the fused loss sums (one 32-byte row each: spreads same-address atomics)
#define MM_UV_TILE 32         // texture-gradient tiles: 32x32 texels, one workgroup and one record list each
#define MM_GSHARD 16          // per image, the pixel backward's maxima are spread over this many words (see mm_backward.hip)
#ifndef MM_CHUNK_PX
#define MM_CHUNK_PX 128          // (256: gather_bwd +3.5 / +4.6 / +10 us at configs 2 / 3 / 5: the waves full of owned chunks are its tail)
#endif                        // the backward sweeps every face's inflated pixel box in chunks of this many pixels, one 8-lane group each:
                              // an even load whatever the box sizes (perspective blow-ups, close-ups at high resolution)
#ifndef MM_GROUP_WORDS
#define MM_GROUP_WORDS 8
#endif                        // ids expanded per window of bin-mask words: 512 -> 1 KiB of uint16 ids per wave (a window never splits a word)

namespace mm {

struct Float3 { float x, y, z; };
// three consecutive 4-byte values at a 4-byte-aligned address, as ONE 12-byte load (global_load_dwordx3)
struct __attribute__((packed, aligned(4))) Packed3 { float x, y, z; };
struct __attribute__((packed, aligned(4))) PackedI3 { int x, y, z; };

// ---- screen bins ------------------------------------------------------------------------------------------------------
// The vertex stage records, per screen bin, which faces' boxes (inflated by the soft-mask margin) may touch it, as one
// bit per face: face order stays implicit in the bit order, which is what kaolin's tie rule and the soft mask's
// "first knum faces" rule need.  Bin edge 8, 16 or 32 px, the smallest whose masks stay under 1/16 of the path's
// algorithmic bytes per image (A = 140 F + 128 HW).
__host__ __device__ inline int bin_shift_for(int H, int W, int F) {
    const size_t words = (size_t)(F + 63) / 64;
    const size_t A = (size_t)140 * F + (size_t)128 * H * W;
    for (int s = 3; s < 5; ++s) {
        const size_t nb = (size_t)((W + (1 << s) - 1) >> s) * ((H + (1 << s) - 1) >> s);
        if (nb * words * 8 * 16 <= A) return s;
    }
    return 5;
}

// One covered pixel's contribution to the texture gradient, appended by the pixel backward to the list of every texture
// tile its bilinear footprint touches; the tile's workgroup streams its list (no search, no atomics on HBM).
struct TexRecord { unsigned xy; float tx, ty, d0, d1, d2; };       // xy = x0 | y0 << 16 (top-left texel)
// The lists of an image are PACKED into one array: the forward counts, per tile, the covered pixels whose footprint touches it (raster_fwd's
// epilogue: an upper bound of the records, exact when no texture gradient is zero), every workgroup of the pixel backward turns the counts
// into the tiles' offsets (the plan workgroup of the image: mm_pixel_bwd.hip), and a tile's records go to [offset, offset + count).  Nothing is sized per tile -- the visible
// surface lands in a few tiles of the texture (a close-up puts four fifths of an image's records into one of 128).
// which form of the forward walk a shape gets (mm_raster_walk.h): the compacting queue (+ the face flags the backward's sweep plan reads) for
// screen bins larger than a tile, the per-batch walk for 8-pixel bins; MM_OPT_WALK_QUEUE / MM_OPT_WALK_BATCH force one (identical results)
#ifndef MM_BATCH_FLAGS
#define MM_BATCH_FLAGS 0     // 1: the per-batch walk (8-pixel bins) flags the faces that receive gradient, like the compacting walk -- built and measured in r06: raster_fwd +2.5-3 us, gather_bwd -1.4 us at 128x128 (profiles/r06_batch_walk_flags_ab.md): off
#endif
inline bool walk_flags_mode(int options, int bin_shift);
inline bool walk_queue_mode(int options, int bin_shift) {
    if (options & MM_OPT_WALK_QUEUE) return true;
    if (options & MM_OPT_WALK_BATCH) return false;
    return bin_shift != 3;
}
// does the forward walk of this shape leave face flags (Workspace::fflag) for the backward's sweep plan?
inline bool walk_flags_mode(int options, int bin_shift) { return MM_BATCH_FLAGS ? true : walk_queue_mode(options, bin_shift); }

// ---- workspace carving (all offsets multiples of 256 bytes) ---------------------------------------------------------
struct Workspace {
    float* T;              // (B,12)     camera transform [R;t], row-major (4,3)
    float* cam;            // (B,48)     the forward's whole Camera record (36 floats used): the backward's camera chain starts from it
                           //            instead of redoing the fp64 trigonometry and the look-at on its critical path
    float4* geo;           // (B,F,3)    {ax,ay,bx,by} {cx,cy,az,bz} {cz, unit normal z, box origin, box extent}; xy in multiplier
                           //            units; box = inflated pixel box px0 | py0 << 16, w | h << 16 (as int bits)
    uint64_t* binmask;     // (B,nbins,ceil(F/64)) bit f: the pixel box of face f, inflated by the soft-mask margin, touches the bin
    float2* soft;          // (B,H,W)    .x = soft-mask product state of uncovered pixels: +prod(1-p) if no factor is 0,
                           //            -prod(non-zero factors) if exactly one factor is 0, 0 if two or more are
    float* dTpart;         // (B,ceil(V/32),12) per-workgroup partial sums of dL/d camera transform (vertex backward; summed in index order)
    unsigned* ticket;      // (B)        arrival counter of the vertex-backward workgroups of an image (same)
                           //            .y (as int bits) = id of the knum-th soft-mask face taken, INT_MAX if fewer were
    float4* gp;            // (B,H,W,2)  covered pixels: K2 contributions of the pixel to its face {d/d(ax,ay,bx,by)} {d/d(cx,cy), d/d(nx,ny)}:
                           //            one 32-byte record = one cache line per item
    float* gp2;            // (B,H,W)    covered pixels: d/d(nz); uncovered pixels: dL/dalpha
    float* dl_part;        // (B,blocks,12) per-workgroup partial sums of dL/dlights (9 used)
    int blocks_per_image;
    unsigned short* order; // (B,4*blocks) raster tiles of an image, most soft-mask candidates first (launch order = heavy first)
    int* nheavy;           // (B,4)      how many of an image's first tiles (in that order) are walked by four waves together; how many are not empty
    int* bincount;         // (B,nbins)  candidates per screen bin (big screens / meshes only: bincount_kernel -> order_kernel)
    int* fflag;            // (B,F,2)    [0] = 1: the face won a pixel; [1] = 1: an uncovered pixel took it into its silhouette product.  Only such faces
                           //            receive gradient from the pixels, and only they are swept by the backward -- a face that only OWNS pixels over its
                           //            own box, not over the box inflated by the silhouette margin (cleared by vertex_fwd, set by raster_fwd)
    long long* ltot;       // (B,MM_LSUB,4) fused loss: per image {sum|pi-gi|, sum p*g, sum p+g-p*g, -} in 2^-32 fixed point, spread over
                           //            MM_LSUB sub-accumulators (64-bit integer atomics of the raster waves: exact, order-free); zeroed by vertex_fwd
    int* tcnt;             // the counters of the backward, ntcnt ints in all, zeroed by the vertex stage of the forward and by every vertex backward for the
    int ntcnt;             //            next one: tcur, toff, tdrop, gmax (below), in this order
    int* tcur;             // (B,ntiles) records appended to a texture tile's list so far (pixel_bwd)
    int* tdrop;            // (B)        records an image's array had no room for (pixel_bwd counts; the texture gather poisons the image and sets tstatus)
    unsigned* gmax;        // (B,MM_GSHARD,8) per-image maxima of the pixel backward (float bits: max |K2 number|, max |dL/dalpha|)
    int* tstatus;          // (B)        records dropped by the last backward: zeroed by vertex_fwd, set by the texture gather, which also poisons the
                           //            image's texture gradient with NaN (mm_render_status reads it)
    int* trcnt;            // (B,ntiles) covered pixels whose bilinear footprint touches the tile: zeroed by vertex_fwd, counted by raster_fwd
    int* toff;             // (B,ntiles) 1 + offset of the tile's list in the image's record array (pixel_bwd's plan workgroup of the image; 0 = not yet)
    TexRecord* trec;       // (B,trcap)  the images' record arrays.  trcap = 9/8 H W by default -- every pixel covered and one footprint in eight across
                           //            a tile border; a larger workspace_bytes enlarges it
    int trcap;
    int2* chunkmap;        // (B,F)      {first sweep item, number of items} of every face (plan kernel, every forward)
    int2* items;           // (B,item_cap) sweep items {face, chunk of its box}
    int2* nitems;          // (B)        {items listed, pixels per chunk in this image (MM_CHUNK_PX << k)}
    float* part;           // (B,item_cap,12) per-item partial sums of dL/d(face xy) (6) and dL/d(unit normal) (3); the vertex backward adds a
                           //            face's items up in index order
    int item_cap;          // F + 16 H W / MM_CHUNK_PX: every face has an item, and sixteen screens' worth of box pixels are cut into chunks
    int ntiles;
    int bin_shift, nbx, nby, words;
    size_t binmask_bytes;
    size_t bytes;
};

__host__ __device__ inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// Small templates in LARGE batches: the vertex backward as one workgroup per image, face-major then vertex-major through LDS
// (vertex_image_bwd_kernel: no ticket, no partial-row round trip).  One workgroup = one CU's load / store pipeline per image: with
// fewer images than a fraction of the chip's 256 CUs the 8-lanes-per-vertex grid of vertex_bwd_kernel (21 workgroups per image) is as
// fast or faster (B=48: 16.2 vs 16.1 us at 128x128, 28.5 vs 18.0 at 256x256 where faces have several sweep items); at B=384 it is 29.6
// against 52.5 us (profiles/r04_per_image_stages_ab.md).
#define MM_VIMG_BWD_MAX_FACES 1700
// K4 (silhouette backward): a pixel whose stored product is EXACTLY 1 took only faces whose factor 1 - exp(-sigma d^2) rounds to 1 in fp32 (exp below 2^-24:
// the forward's alpha is 0 to the last bit).  Their derivative terms are not zero, only < 6e-8 of a term at exp ~ 0.5.  Rounds 1-4 skipped such pixels
// (MM_K4_KEEP_ONES 0); since round 5 they are evaluated like any other: in an image that shows NOTHING ELSE -- a far-away mesh on an 8x8 screen -- they are the
// whole geometry gradient, which the scale-aware parity bar sees (fuzz case 82 of seed 6106), and the skip bought no time (profiles/r05_k4_ones_ab.md).
#ifndef MM_K4_KEEP_ONES
#define MM_K4_KEEP_ONES 1
#endif
#ifndef MM_VIMG_BWD_MIN_B
#define MM_VIMG_BWD_MIN_B 128
#endif
inline bool vertex_bwd_per_image(int B, int F, int vc_stride) {
    return B >= MM_VIMG_BWD_MIN_B && F <= MM_VIMG_BWD_MAX_FACES && vc_stride <= 64;
}

// avail: bytes the caller really has (0 = the minimum, what mm_query_workspace reports): what is beyond the minimum goes to the images' record arrays
// geo_only (MMRenderDesc.geometry_only: nothing is rasterised): only what the vertex stage touches -- camera, face records, face flags, the counters it
// clears, the dL/dT partials and ONE (never written, never used) row of item sums per image for the backward's unconditional loads; every other block
// is empty.  3 MB instead of 71 MB at B=48, 1 280 faces, 128x128 (advisor r05: the lean trainer step keeps such a workspace alive per geometry render).
__host__ __device__ inline Workspace carve_workspace(void* base, int B, int V, int F, int H, int W, int Ht, int Wt, size_t avail = 0, bool geo_only = false) {
    Workspace w;
    char* p = (char*)base;
    size_t o = 0;
    const size_t px = geo_only ? 0 : (size_t)H * W;              // per-pixel blocks
    w.bin_shift = bin_shift_for(H, W, F);
    w.nbx = (W + (1 << w.bin_shift) - 1) >> w.bin_shift;
    w.nby = (H + (1 << w.bin_shift) - 1) >> w.bin_shift;
    w.words = (F + 63) / 64;
    w.binmask_bytes = geo_only ? 0 : (size_t)B * w.nbx * w.nby * w.words * sizeof(uint64_t);
    w.T = (float*)(p + o);          o += align256((size_t)B * 12 * sizeof(float));
    w.cam = (float*)(p + o);        o += align256((size_t)B * 48 * sizeof(float));
    w.geo = (float4*)(p + o);       o += align256((size_t)B * F * 3 * sizeof(float4));
    w.binmask = (uint64_t*)(p + o); o += align256(w.binmask_bytes);
    w.soft = (float2*)(p + o);      o += align256((size_t)B * px * sizeof(float2));
    w.dTpart = (float*)(p + o);     o += align256((size_t)B * ((V + 31) / 32) * 12 * sizeof(float));
    w.ticket = (unsigned*)(p + o);  o += align256((size_t)B * sizeof(unsigned));
    w.gp = (float4*)(p + o);        o += align256((size_t)B * px * 2 * sizeof(float4));
    w.gp2 = (float*)(p + o);        o += align256((size_t)B * px * sizeof(float));
    w.blocks_per_image = ((W + MM_BLOCK_PX - 1) / MM_BLOCK_PX) * ((H + MM_BLOCK_PX - 1) / MM_BLOCK_PX);
    w.dl_part = (float*)(p + o);    o += align256(geo_only ? 0 : (size_t)B * w.blocks_per_image * 12 * sizeof(float));
    w.ltot = (long long*)(p + o);   o += align256((size_t)B * MM_LSUB * 4 * sizeof(long long));
    w.order = (unsigned short*)(p + o); o += align256(geo_only ? 0 : (size_t)B * 4 * w.blocks_per_image * sizeof(unsigned short));
    w.nheavy = (int*)(p + o);       o += align256(geo_only ? 0 : (size_t)B * 4 * sizeof(int));
    w.bincount = (int*)(p + o);     o += align256(geo_only ? 0 : (size_t)B * w.nbx * w.nby * sizeof(int));
    w.fflag = (int*)(p + o);        o += align256((size_t)B * F * 2 * sizeof(int));   // (ints, not bytes: a byte store may alias every later load in the compiler's eyes)
    w.ntiles = ((Wt + MM_UV_TILE - 1) / MM_UV_TILE) * ((Ht + MM_UV_TILE - 1) / MM_UV_TILE);
    const size_t ndrop = ((size_t)B * w.ntiles * 2 + B + 7) / 8 * 8 - (size_t)B * w.ntiles * 2;   // (gmax starts on a 32-byte sector)
    w.ntcnt = (int)((size_t)B * w.ntiles * 2 + ndrop + (size_t)B * MM_GSHARD * 8);
    w.tcnt = (int*)(p + o);         o += align256(((size_t)w.ntcnt + B + (size_t)B * w.ntiles) * sizeof(int));
    w.tcur = w.tcnt;
    w.toff = w.tcur + (size_t)B * w.ntiles;
    w.tdrop = w.toff + (size_t)B * w.ntiles;
    w.gmax = (unsigned*)(w.tdrop + ndrop);
    w.tstatus = w.tcnt + w.ntcnt;
    w.trcnt = w.tstatus + B;
    w.item_cap = geo_only ? 1 : F + (int)(((size_t)16 * H * W + MM_CHUNK_PX - 1) / MM_CHUNK_PX);
    w.chunkmap = (int2*)(p + o);    o += align256(geo_only ? 0 : (size_t)B * F * sizeof(int2));
    w.items = (int2*)(p + o);       o += align256(geo_only ? 0 : (size_t)B * w.item_cap * sizeof(int2));
    w.nitems = (int2*)(p + o);      o += align256(geo_only ? 0 : (size_t)B * sizeof(int2));
    w.part = (float*)(p + o);       o += align256((size_t)B * w.item_cap * 12 * sizeof(float));
    w.trec = (TexRecord*)(p + o);                                 // (last: the record arrays take what the caller gives beyond the minimum)
    const size_t rc_min = geo_only ? 0 : ((size_t)H * W * 9 / 8 + 255) & ~(size_t)255;
    size_t rc = rc_min;
    const size_t need = o + align256((size_t)B * rc_min * sizeof(TexRecord));
    if (avail > need && !geo_only) rc += (avail - need) / ((size_t)B * sizeof(TexRecord));
    if (rc > ((size_t)1 << 30)) rc = (size_t)1 << 30;
    w.trcap = (int)rc;
    o += align256((size_t)B * rc_min * sizeof(TexRecord));
    w.bytes = o;
    return w;
}

// ---- camera: smr_utils.py:257-311 + networks.py:278-282 -----------------------------------------------------------
struct Camera {
    float ce, se, ca, sa;
    float cam[3], zv[3], zl, z[3], xv[3], xl, x[3], y[3];
    float T[12];
};
static_assert(sizeof(Camera) == 36 * sizeof(float), "stored as 36 floats per image (Workspace::cam)");

__device__ inline void cross3(const float* a, const float* b, float* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

#define MM_DEG2RAD 0.017453292519943295f

// trig values are produced by the caller (fp64 sin/cos rounded to fp32, one lane per value)
__device__ inline void camera_build(float dist, float ce, float se, float ca, float sa, float bx, float by, Camera& c) {
    const float up[3] = {0.f, 1.f, 0.f};
    c.ce = ce; c.se = se; c.ca = ca; c.sa = sa;
    c.cam[0] = (dist * ce) * sa;
    c.cam[1] = dist * se;
    c.cam[2] = (dist * ce) * ca;
    const float at[3] = {bx, by, 0.f};
    for (int i = 0; i < 3; ++i) c.zv[i] = c.cam[i] - at[i];
    c.zl = sqrtf((c.zv[0] * c.zv[0] + c.zv[1] * c.zv[1]) + c.zv[2] * c.zv[2]);
    for (int i = 0; i < 3; ++i) c.z[i] = c.zv[i] / c.zl;
    cross3(up, c.z, c.xv);
    c.xl = sqrtf((c.xv[0] * c.xv[0] + c.xv[1] * c.xv[1]) + c.xv[2] * c.xv[2]);
    for (int i = 0; i < 3; ++i) c.x[i] = c.xv[i] / c.xl;
    cross3(c.z, c.x, c.y);
    for (int i = 0; i < 3; ++i) { c.T[i * 3 + 0] = c.x[i]; c.T[i * 3 + 1] = c.y[i]; c.T[i * 3 + 2] = c.z[i]; }
    for (int j = 0; j < 3; ++j)
        c.T[9 + j] = ((-c.cam[0]) * c.T[0 + j] + (-c.cam[1]) * c.T[3 + j]) + (-c.cam[2]) * c.T[6 + j];
}

// dT (4,3) -> d(dist, elev[deg], azim[deg], bias)
__device__ inline void camera_backward(float dist, const Camera& c, const float* dT, float* ddist, float* delev,
                                       float* dazim, float* dbias) {
    const float up[3] = {0.f, 1.f, 0.f};
    float dx[3], dy[3], dz[3], dcam[3], tmp[3];
    for (int i = 0; i < 3; ++i) {
        dx[i] = dT[i * 3 + 0] - c.cam[i] * dT[9 + 0];
        dy[i] = dT[i * 3 + 1] - c.cam[i] * dT[9 + 1];
        dz[i] = dT[i * 3 + 2] - c.cam[i] * dT[9 + 2];
        dcam[i] = -((c.T[i * 3 + 0] * dT[9 + 0] + c.T[i * 3 + 1] * dT[9 + 1]) + c.T[i * 3 + 2] * dT[9 + 2]);
    }
    cross3(c.x, dy, tmp); for (int i = 0; i < 3; ++i) dz[i] += tmp[i];
    cross3(dy, c.z, tmp); for (int i = 0; i < 3; ++i) dx[i] += tmp[i];
    float xd = (c.x[0] * dx[0] + c.x[1] * dx[1]) + c.x[2] * dx[2];
    float dxv[3]; for (int i = 0; i < 3; ++i) dxv[i] = (dx[i] - c.x[i] * xd) / c.xl;
    cross3(dxv, up, tmp); for (int i = 0; i < 3; ++i) dz[i] += tmp[i];
    float zd = (c.z[0] * dz[0] + c.z[1] * dz[1]) + c.z[2] * dz[2];
    float dzv[3]; for (int i = 0; i < 3; ++i) dzv[i] = (dz[i] - c.z[i] * zd) / c.zl;
    for (int i = 0; i < 3; ++i) dcam[i] += dzv[i];
    dbias[0] = -dzv[0]; dbias[1] = -dzv[1];
    *ddist = (dcam[0] * (c.ce * c.sa) + dcam[1] * c.se) + dcam[2] * (c.ce * c.ca);
    float de = (dcam[0] * (-(dist * c.se) * c.sa) + dcam[1] * (dist * c.ce)) + dcam[2] * (-(dist * c.se) * c.ca);
    float da = dcam[0] * ((dist * c.ce) * c.ca) + dcam[2] * (-(dist * c.ce) * c.sa);
    *delev = de * MM_DEG2RAD; *dazim = da * MM_DEG2RAD;
}

// ---- prepare_vertices pieces ----------------------------------------------------------------------
__device__ inline Float3 to_camera(const float* __restrict__ v, const float* T) {
    Float3 r;
    r.x = ((v[0] * T[0] + v[1] * T[3]) + v[2] * T[6]) + T[9];
    r.y = ((v[0] * T[1] + v[1] * T[4]) + v[2] * T[7]) + T[10];
    r.z = ((v[0] * T[2] + v[1] * T[5]) + v[2] * T[8]) + T[11];
    return r;
}

// ---- pixel centre convention of kaolin's rasteriser -----------------------------------------------
__device__ inline float pixel_x(int px, int W, float mult) { return (mult / (float)W) * (float)(2 * px + 1 - W); }
__device__ inline float pixel_y(int py, int H, float mult) { return (mult / (float)H) * (float)(H - 2 * py - 1); }

// ---- one factor of the silhouette product:  q = 1 - exp(-sigma' d^2),  d = distance from the pixel centre to the
// nearest edge of the face, sigma' = sigmainv / multiplier^2.  Held to 1e-4, not to the bit: hardware reciprocal / exp2 (about 1 ulp each).
// But it must be BIT-IDENTICAL in the forward and in the backward, whatever flags their translation units are compiled with: the forward
// stores the product of the factors of a pixel, and the backward gets "the product over the OTHER faces" by dividing by the factor it
// recomputes.  For a pixel centre almost exactly on an edge, q is a few ulps of 1 - p, and a last-bit difference between the two
// evaluations is a 10-20 % error in that pixel's gradient (found by profiles/tools/fuzz_parity.py: 1e-3 absolute on one vertex).
// Hence: no contraction inside (pragma), no division (the flags change how `/` rounds), pixel centre and sigma' passed in as the
// forward computes them (IEEE: multiplier / W and sigmainv / multiplier^2 are formed on the host for the backward).
__device__ inline float seg_dist2_fast(float px, float py, float ux, float uy, float vx, float vy) {
MM_FP_EXACT
    const float ex = vx - ux, ey = vy - uy, rx = px - ux, ry = py - uy;
    const float len2 = ex * ex + ey * ey;
    const float dot = rx * ex + ry * ey;
    float t = (len2 > 0.f) ? dot * __builtin_amdgcn_rcpf(len2) : 0.f;
    t = fminf(fmaxf(t, 0.f), 1.f);                                // clamped projection: the three regions of seg_dist2 in one form
    const float qx = rx - t * ex, qy = ry - t * ey;
    return qx * qx + qy * qy;
}
__device__ inline float soft_factor(float x0, float y0, const float4& p0, const float4& p1, float sig2) {
MM_FP_EXACT
#ifdef MM_BOUND_SOFT1                                           // (bound experiment, WRONG results: one edge instead of three)
    const float d = seg_dist2_fast(x0, y0, p0.x, p0.y, p0.z, p0.w);
#else
    const float d = fminf(fminf(seg_dist2_fast(x0, y0, p0.x, p0.y, p0.z, p0.w), seg_dist2_fast(x0, y0, p0.z, p0.w, p1.x, p1.y)),
                          seg_dist2_fast(x0, y0, p1.x, p1.y, p0.x, p0.y));
#endif
    return 1.f - __builtin_amdgcn_exp2f(-(d * sig2) * 1.4426950408889634f);
}
// pixel centre from the host-formed factor k = multiplier / W (or / H): the same float as pixel_x / pixel_y in a translation unit with
// IEEE division, in any translation unit
__device__ inline float pixel_x_k(int px, int W, float kx) {
MM_FP_EXACT
    return kx * (float)(2 * px + 1 - W);
}
__device__ inline float pixel_y_k(int py, int H, float ky) {
MM_FP_EXACT
    return ky * (float)(H - 2 * py - 1);
}

// Which box borders are OPEN (a pixel centre exactly on them is outside) -- SURVEY Appendix C-4, the three forms upstream may have:
//   0                                   closed box            reject  x <  lo || x >  hi     (default)
//   MM_OPT_BBOX_MIN_CLOSED_MAX_OPEN     [lo, hi)              reject  x <  lo || x >= hi
//   MM_OPT_BBOX_HALF_OPEN               (lo, hi)              reject  x <= lo || x >= hi
// bit 0: the min border is open, bit 1: the max border is open
__host__ __device__ inline int box_mode(int options) {
    return ((options & MM_OPT_BBOX_HALF_OPEN) ? 3 : 0) | ((options & MM_OPT_BBOX_MIN_CLOSED_MAX_OPEN) ? 2 : 0);
}
__device__ inline bool box_reject(float x, float lo, float hi, int mode) {                       // (straight-line on purpose: no branch per border)
    return (x < lo) | (x > hi) | (((mode & 1) != 0) & (x == lo)) | (((mode & 2) != 0) & (x == hi));
}

// conservative pixel range [lo, hi] whose centres can satisfy  lo_v <= centre <= hi_v  (a 0.02 px slack covers the
// rounding of this closed form by orders of magnitude; callers re-test every pixel exactly).  flip: centres fall with the index (y).
__device__ inline void pixel_range(float lo_v, float hi_v, float mult, int n, bool flip, int& lo, int& hi) {
    const float a = (lo_v / mult) * (float)n, c = (hi_v / mult) * (float)n;
    float flo, fhi;
    if (!flip) { flo = (a + (float)(n - 1)) * 0.5f; fhi = (c + (float)(n - 1)) * 0.5f; }
    else { flo = ((float)(n - 1) - c) * 0.5f; fhi = ((float)(n - 1) - a) * 0.5f; }
    if (!(fabsf(flo) < 1e9f) || !(fabsf(fhi) < 1e9f)) { lo = 0; hi = n - 1; return; }         // inf / NaN: every pixel
    lo = (int)ceilf(flo - 0.02f); hi = (int)floorf(fhi + 0.02f);                                // 0.02 px >> rounding of flo/fhi
    lo = lo < 0 ? 0 : lo; hi = hi > n - 1 ? n - 1 : hi;
}

// barycentric weights exactly as packed_rasterize_forward (edge functions, copysign(eps) normalisation)
__device__ inline void edge_weights(float ax, float ay, float bx, float by, float cx, float cy, float x0, float y0, float eps,
                                    float& w0, float& w1, float& w2, float& nrm) {
    float aex = ax - x0, aey = ay - y0, bex = bx - x0, bey = by - y0, cex = cx - x0, cey = cy - y0;
    w0 = bex * cey - bey * cex;
    w1 = cex * aey - cey * aex;
    w2 = aex * bey - aey * bex;
    nrm = (w0 + w1) + w2;
    nrm += copysignf(eps, nrm);
}

// normalised barycentrics.  Default: kaolin's three edge functions over their copysign(eps)-padded sum (edge_weights above + IEEE
// divisions: the expressions the oracle evaluates).  MM_OPT_BARY_ONE_MINUS: w1, w2 over (sum + eps), w0 = 1 - w1 - w2.
__device__ inline void bary_weights(float ax, float ay, float bx, float by, float cx, float cy, float x0, float y0, float eps, bool one_minus,
                                    float& w0, float& w1, float& w2, float& nrm) {
    if (!one_minus) {
        edge_weights(ax, ay, bx, by, cx, cy, x0, y0, eps, w0, w1, w2, nrm);
#ifdef MM_BOUND_FASTDIV                                         // (bound experiment, WRONG bits: one hardware reciprocal instead of three IEEE divisions)
        const float r = __builtin_amdgcn_rcpf(nrm);
        w0 *= r; w1 *= r; w2 *= r;
#else
        w0 /= nrm; w1 /= nrm; w2 /= nrm;
#endif
        return;
    }
    float aex = ax - x0, aey = ay - y0, bex = bx - x0, bey = by - y0, cex = cx - x0, cey = cy - y0;
    const float k0 = bex * cey - bey * cex, k1 = cex * aey - cey * aex, k2 = aex * bey - aey * bex;
    nrm = (k0 + k1) + k2;
    nrm += eps;
    w1 = k1 / nrm; w2 = k2 / nrm; w0 = (1.f - w1) - w2;
}

// ---- bilinear texture fetch = grid_sample(align_corners=False, padding_mode='border') -------------
struct Bilin { int x0, y0, x1, y1; float wnw, wne, wsw, wse, tx, ty, mx, my; };

__device__ inline Bilin bilin_setup(float u, float v, int Ht, int Wt) {
    Bilin s;
    float gx = u * 2.f - 1.f;
    float gy = -(v * 2.f - 1.f);
    float ix = ((gx + 1.f) * (float)Wt - 1.f) / 2.f;
    float iy = ((gy + 1.f) * (float)Ht - 1.f) / 2.f;
    if (ix <= 0.f) { ix = 0.f; s.mx = 0.f; } else if (ix >= (float)(Wt - 1)) { ix = (float)(Wt - 1); s.mx = 0.f; } else s.mx = 1.f;
    if (iy <= 0.f) { iy = 0.f; s.my = 0.f; } else if (iy >= (float)(Ht - 1)) { iy = (float)(Ht - 1); s.my = 0.f; } else s.my = 1.f;
    float fx = floorf(ix), fy = floorf(iy);
    s.x0 = (int)fx; s.y0 = (int)fy; s.x1 = s.x0 + 1; s.y1 = s.y0 + 1;
    s.tx = ix - fx; s.ty = iy - fy;
    float ex = (fx + 1.f) - ix, ey = (fy + 1.f) - iy;
    s.wnw = ex * ey; s.wne = s.tx * ey; s.wsw = ex * s.ty; s.wse = s.tx * s.ty;
    return s;
}

// ---- spherical harmonics -------------------------------------------------------------------------
#define MM_SH_C0 0.28209479177f
#define MM_SH_C1 0.4886025119f
#define MM_SH_C4 1.09254843059f
#define MM_SH_C6 0.94617469575f
#define MM_SH_C6B 0.31539156525f
#define MM_SH_C7 0.77254840404f
#define MM_SH_C8 0.38627420202f

__device__ inline void sh_bands(float x, float y, float z, float* b) {
    b[0] = MM_SH_C0;
    b[1] = MM_SH_C1 * x;
    b[2] = MM_SH_C1 * z;
    b[3] = MM_SH_C1 * y;
    b[4] = MM_SH_C4 * (x * y);
    b[5] = MM_SH_C4 * (y * z);
    b[6] = MM_SH_C6 * (z * z) - MM_SH_C6B;
    b[7] = MM_SH_C7 * (x * z);
    b[8] = MM_SH_C8 * (x * x - y * y);
}

// ---- squared distance from p to segment u->v; region 0: clamped at u, 1: interior, 2: clamped at v ----------------
__device__ inline float seg_dist2(float px, float py, float ux, float uy, float vx, float vy, int& region) {
    float ex = vx - ux, ey = vy - uy, rx = px - ux, ry = py - uy;
    float len2 = ex * ex + ey * ey;
    float dot = rx * ex + ry * ey;
    float t = (len2 > 0.f) ? dot / len2 : 0.f;
    if (t <= 0.f) { region = 0; return rx * rx + ry * ry; }
    if (t >= 1.f) { region = 2; float sx = px - vx, sy = py - vy; return sx * sx + sy * sy; }
    region = 1;
    float qx = px - (ux + t * ex), qy = py - (uy + t * ey);
    return qx * qx + qy * qy;
}

// ---- workgroup -> (image, strip) mapping that keeps all strips of one image on one XCD ------------------------------
// The dispatcher places workgroup i on XCD i % 8 (observed, used for L2 locality only: nothing depends on it for
// correctness).  Images are dealt to XCDs in groups of 8 so that the face records of an image are read through a
// single XCD's L2 instead of all eight.
__device__ inline void map_block(int linear, int B, int tiles_per_image, int& b, int& tile) {
    const int full = (B / 8) * 8 * tiles_per_image;
    if (linear < full) {
        const int xcd = linear & 7, k = linear >> 3;
        b = (k / tiles_per_image) * 8 + xcd;
        tile = k % tiles_per_image;
    } else {
        const int r = linear - full;
        b = (B / 8) * 8 + r / tiles_per_image;
        tile = r % tiles_per_image;
    }
}

// optional per-kernel event pair (MMRenderDesc.prof_events / MMReconDesc.prof_events)
struct ProfScope {
    void** ev; int slot; hipStream_t s;
    ProfScope(void** e, int sl, hipStream_t st) : ev(e), slot(sl), s(st) { if (ev) (void)hipEventRecord((hipEvent_t)ev[2 * slot], s); }
    ~ProfScope() { if (ev) (void)hipEventRecord((hipEvent_t)ev[2 * slot + 1], s); }
};

// -DMM_TIMELINE (debug builds, profiles/tools/timeline.py): wall-clock begin / end of every workgroup of a kernel, 100 MHz ticks
// comparable across CUs.  MM_TIMELINE_STORAGE(name) in the kernel's translation unit defines the buffer and its C getter
// mm_debug_timeline_<name>(out[MM_TIMELINE_MAX][2]).  Everything compiles to nothing otherwise.
#ifdef MM_TIMELINE
#define MM_TIMELINE_MAX 81920
// third word: where the workgroup's first wave ran -- HW_REG_HW_ID (gfx9 layout: wave slot [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13]) in the
// low half, HW_REG_XCC_ID [3:0] (the XCD) in the high half: profiles/tools/timeline.py turns it into the per-CU placement of the launch
#define MM_TIMELINE_STORAGE(name)                                                                                         \
    namespace mm { __device__ unsigned long long g_tl_##name[MM_TIMELINE_MAX][3]; }                                          \
    extern "C" int mm_debug_timeline_##name(unsigned long long* out) {                                                       \
        return hipMemcpyFromSymbol(out, HIP_SYMBOL(mm::g_tl_##name), sizeof(unsigned long long) * MM_TIMELINE_MAX * 3) == hipSuccess ? 0 : -1; \
    }
#define MM_TIMELINE_BEGIN() const unsigned long long tl_begin_ = wall_clock64()
#define MM_TIMELINE_END(name) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x < MM_TIMELINE_MAX) {                 \
    mm::g_tl_##name[blockIdx.x][0] = tl_begin_; mm::g_tl_##name[blockIdx.x][1] = wall_clock64();                             \
    mm::g_tl_##name[blockIdx.x][2] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) |                         \
                                     ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (3 << 11)) << 32); } } while (0)
#else
#define MM_TIMELINE_STORAGE(name)
#define MM_TIMELINE_BEGIN() do { } while (0)
#define MM_TIMELINE_END(name) do { } while (0)
#endif

// -DMM_PHASE_PROF (debug builds, profiles/tools/phase_prof.py): per-wave wall-clock totals of a kernel's phases.  A PhaseProf object
// accumulates the time between successive mark(i) calls into slot i (every mark first waits for everything the wave has issued,
// so a phase is charged with the latency of its own loads; that serialisation is why this is a debug build only) and flush()
// stores the slots + the wave's total + two free counters.  MM_PP_STORAGE(name) in the kernel's translation unit defines the
// buffer and its C getter mm_debug_pp_<name>(out[MM_PP_MAX][MM_PP_SLOTS + 3]).  Compiles to nothing otherwise.
#ifdef MM_PHASE_PROF
#define MM_PP_MAX 16384
#define MM_PP_SLOTS 10
#define MM_PP_STORAGE(name)                                                                                               \
    namespace mm { __device__ unsigned long long g_pp_##name[MM_PP_MAX][MM_PP_SLOTS + 3]; }                                  \
    extern "C" int mm_debug_pp_##name(unsigned long long* out) {                                                             \
        return hipMemcpyFromSymbol(out, HIP_SYMBOL(mm::g_pp_##name), sizeof(unsigned long long) * MM_PP_MAX * (MM_PP_SLOTS + 3)) == hipSuccess ? 0 : -1; \
    }
struct PhaseProf {
    unsigned long long t0, t, acc[MM_PP_SLOTS], c0, c1;
    __device__ PhaseProf() : c0(0), c1(0) {
        for (int i = 0; i < MM_PP_SLOTS; ++i) acc[i] = 0;
        t0 = t = wall_clock64();
    }
    __device__ void mark(int i) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long n = wall_clock64(); acc[i] += n - t; t = n; }
    __device__ void count(unsigned long long a, unsigned long long b) { c0 += a; c1 += b; }
};
#define MM_PP_BEGIN() mm::PhaseProf pp_
#define MM_PP_MARK(i) pp_.mark(i)
#define MM_PP_COUNT(a, b) pp_.count(a, b)
#define MM_PP_ARG , mm::PhaseProf& pp_
#define MM_PP_PASS , pp_
#define MM_PP_FLUSH(name, wave_index) do { const long long wi_ = (wave_index); if ((threadIdx.x & 63) == 0 && wi_ < MM_PP_MAX) {     \
    for (int i_ = 0; i_ < MM_PP_SLOTS; ++i_) mm::g_pp_##name[wi_][i_] = pp_.acc[i_];                                          \
    mm::g_pp_##name[wi_][MM_PP_SLOTS] = wall_clock64() - pp_.t0; mm::g_pp_##name[wi_][MM_PP_SLOTS + 1] = pp_.c0; mm::g_pp_##name[wi_][MM_PP_SLOTS + 2] = pp_.c1; } } while (0)
#else
#define MM_PP_STORAGE(name)
#define MM_PP_BEGIN() do { } while (0)
#define MM_PP_MARK(i) do { } while (0)
#define MM_PP_COUNT(a, b) do { } while (0)
#define MM_PP_ARG
#define MM_PP_PASS
#define MM_PP_FLUSH(name, wave_index) do { } while (0)
#endif

// launch check shared by every launcher: the HIP error (if any) is kept per host thread for mm_last_error_detail()
struct LaunchError { hipError_t code; const char* what; };
inline LaunchError& last_launch_error() { static thread_local LaunchError e = {hipSuccess, ""}; return e; }
inline int launch_ok(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return MM_OK;
    last_launch_error() = {e, what};
    return MM_ERR_LAUNCH;
}
// errors left behind by the caller's own earlier runtime calls (e.g. hipEventQuery -> hipErrorNotReady) are not ours
inline void clear_stale_error() { (void)hipGetLastError(); }

// number of set bits of a ballot below this lane: two v_mbcnt instructions (a 64-bit shift/and/popcount sequence costs ~10x)
__device__ inline int ballot_rank(uint64_t bal) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
}

// lane <-> lane ^ S exchange of a 32-bit value WITHOUT the LDS crossbar: DPP lane selects for S = 1, 2, 4, 8 (quad permutes, half-row
// mirror, row rotate) and gfx950's v_permlane16_swap / v_permlane32_swap for S = 16, 32.  A ds_bpermute costs an address VGPR, an LDS
// instruction and ~100 cycles of latency on the wave's critical path; a 64x64 bit transpose made twelve of them, dependent in pairs.
// (checked against the definition on the box: profiles/tools/xchg_test.hip)
typedef unsigned mm_v2u __attribute__((ext_vector_type(2)));
template <int CTRL>
__device__ inline unsigned dpp_u32(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false); }
template <int S>
__device__ inline unsigned lane_xchg(unsigned v, int lane) {
    if constexpr (S == 1) return dpp_u32<0xB1>(v);                       // quad_perm [1,0,3,2]
    else if constexpr (S == 2) return dpp_u32<0x4E>(v);                  // quad_perm [2,3,0,1]
    else if constexpr (S == 4) return dpp_u32<0x1B>(dpp_u32<0x141>(v));  // row_half_mirror (^7) then quad_perm [3,2,1,0] (^3)
    else if constexpr (S == 8) return dpp_u32<0x128>(v);                 // row_ror:8
    else if constexpr (S == 16) { const mm_v2u r = __builtin_amdgcn_permlane16_swap(v, v, false, false); return (lane & 16) ? r.x : r.y; }
    else { const mm_v2u r = __builtin_amdgcn_permlane32_swap(v, v, false, false); return (lane & 32) ? r.x : r.y; }
}

template <int S>
__device__ inline float xchg_f32(float v, int lane) { return __uint_as_float(lane_xchg<S>(__float_as_uint(v), lane)); }

// 64x64 bit-matrix transpose across the wave: lane i holds row i on entry and column i on exit (6 butterfly stages).
template <int S>
__device__ inline uint64_t transpose_stage(uint64_t x, int lane) {
    // m: bit positions whose index has bit S clear
    constexpr uint64_t m = S == 32 ? 0x00000000FFFFFFFFull : S == 16 ? 0x0000FFFF0000FFFFull : S == 8 ? 0x00FF00FF00FF00FFull
                         : S == 4 ? 0x0F0F0F0F0F0F0F0Full : S == 2 ? 0x3333333333333333ull : 0x5555555555555555ull;
    const unsigned lo = lane_xchg<S>((unsigned)x, lane), hi = lane_xchg<S>((unsigned)(x >> 32), lane);
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

// Exclusive prefix sum over the wave (total = the wave's sum, in every lane): the classic DPP scan -- row_shr 1, 2, 4, 8 inside the
// 16-lane rows (lanes shifted in from outside a row read 0), then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3.
// Six VALU instructions with cross-lane operands instead of six dependent ds_bpermute round trips (~100 cycles each); the walk kernels
// scan twice per batch of candidates.  Integer adds: the result is the same whatever the order.
template <int CTRL, int ROWS, bool BOUND>
__device__ inline int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROWS, 0xF, BOUND); }
__device__ inline int wave_prefix_excl(int v, int lane, int& total) {
    (void)lane;
    int inc = v;
    inc += dpp_i32<0x111, 0xF, true>(inc);                       // row_shr:1
    inc += dpp_i32<0x112, 0xF, true>(inc);                       // row_shr:2
    inc += dpp_i32<0x114, 0xF, true>(inc);                       // row_shr:4
    inc += dpp_i32<0x118, 0xF, true>(inc);                       // row_shr:8
    inc += dpp_i32<0x142, 0xA, false>(inc);                      // row_bcast:15 -> rows 1, 3
    inc += dpp_i32<0x143, 0xC, false>(inc);                      // row_bcast:31 -> rows 2, 3
    total = __builtin_amdgcn_readlane(inc, 63);
    return inc - v;
}

// ---- screen binning of one wave's 64 faces = mask word c of every bin -------------------------------------------------
// A lane turns its face's pixel box (inflated by the soft-mask margin, conservative; bw <= 0: none) into bin column / row
// ranges.  Per block of 8x8 bins the lane's coverage is a 64-bit row-major bit matrix (rows x columns outer product); ONE wave
// transpose turns the 64 faces' coverage words into the 64 bins' mask words, stored plainly -- every word of every bin is
// written (blocks no face touches skip the transpose): no atomics, no zero-fill, no second pass over the face records.
// The raster kernel re-tests every (pixel, face) pair exactly, so a conservative mask changes no result.
__device__ inline void bin_wave_faces(uint64_t* mask, int b, int nbx, int nby, int words, int bin_shift, int c, int lane,
                                      int bx0, int by0, int bw, int bh) {
    int c0 = 0, c1 = -1, r0 = 0, r1 = -1;                         // bin columns / rows the box touches (none)
    if (bw > 0 && bh > 0) {
        c0 = bx0 >> bin_shift; c1 = (bx0 + bw - 1) >> bin_shift;
        r0 = by0 >> bin_shift; r1 = (by0 + bh - 1) >> bin_shift;
    }
    const int sbx = (nbx + 7) >> 3, sby = (nby + 7) >> 3;
    for (int s = 0; s < sbx * sby; ++s) {
        const int kx0 = (s % sbx) * 8, ky0 = (s / sbx) * 8;
        const int clo = max(c0 - kx0, 0), chi = min(c1 - kx0, 7), rlo = max(r0 - ky0, 0), rhi = min(r1 - ky0, 7);
        const unsigned col = chi >= clo ? ((2u << chi) - (1u << clo)) : 0u;       // bits clo..chi
        const unsigned row = rhi >= rlo ? ((2u << rhi) - (1u << rlo)) : 0u;
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            lo |= ((row >> r) & 1u) ? (col << (8 * r)) : 0u;
            hi |= ((row >> (r + 4)) & 1u) ? (col << (8 * r)) : 0u;
        }
        const uint64_t cov = ((uint64_t)hi << 32) | lo;           // bit (r*8+c): this face touches bin (ky0+r, kx0+c)
        uint64_t word = 0;
        if (__ballot(cov != 0)) word = wave_transpose64(cov, lane);               // lane j: bit i = face c*64+i touches bin j
        const int kx = kx0 + (lane & 7), ky = ky0 + (lane >> 3);
        if (kx < nbx && ky < nby) mask[((size_t)b * nbx * nby + (size_t)ky * nbx + kx) * words + c] = word;
    }
}

// the face's pixel box inflated by the soft-mask margin (conservative, see pixel_range), packed for the face sweeps of the
// backward: org = px0 | py0 << 16, ext = width | height << 16 (0 x 0 if it misses the image)
__device__ inline void face_pixel_box(float ax, float ay, float bx, float by, float cx, float cy, float infl, float mult, int W, int H,
                                      int& bx0, int& by0, int& bw, int& bh, unsigned& org, unsigned& ext) {
    int bx1, by1;
    pixel_range(fminf(fminf(ax, bx), cx) - infl, fmaxf(fmaxf(ax, bx), cx) + infl, mult, W, false, bx0, bx1);
    pixel_range(fminf(fminf(ay, by), cy) - infl, fmaxf(fmaxf(ay, by), cy) + infl, mult, H, true, by0, by1);
    bw = bx1 - bx0 + 1; bh = by1 - by0 + 1;
    const bool hit = bw > 0 && bh > 0;
    org = hit ? ((unsigned)bx0 | ((unsigned)by0 << 16)) : 0u;
    ext = hit ? ((unsigned)bw | ((unsigned)bh << 16)) : 0u;
}

// The pixel box the backward sweeps for a face.  The record holds the box inflated by the silhouette margin (what an uncovered pixel needs
// to find the faces of its product, K4).  A face that is in NO silhouette product and only owns pixels (K2) is swept over that box shrunk
// again by the margin's whole pixels minus one: lo_infl = ceil(f_tight - margin_px - 0.02) (pixel_range), so lo_infl + s <= ceil(f_tight) for
// any s <= margin_px -- the owned pixels (centres inside the triangle, hence inside its own box) stay inside.  Sides cut off by the image
// border are left alone (the clipped edge says nothing about where the box would have begun).  sx, sy: the shrink per axis.
__host__ __device__ inline int sweep_shrink(float boxlen, int n) { const int s = (int)(boxlen * (float)n * 0.5f) - 1; return s > 0 ? s : 0; }
__host__ __device__ inline void sweep_box(unsigned org, unsigned ext, bool taken, int sx, int sy, int W, int H, int& px0, int& py0, int& bw, int& bh) {
    px0 = (int)(org & 0xFFFFu); py0 = (int)(org >> 16); bw = (int)(ext & 0xFFFFu); bh = (int)(ext >> 16);
#ifdef MM_DBG_NO_INFLATE                                        // (timing experiment only, results WRONG: every face swept over its own box -- what the K4 sweeps of the
    taken = false;                                               //  inflated boxes cost, profiles/r05_gather_k4_bound.md)
#endif
    if (taken || bw <= 0 || bh <= 0) return;
    const int l = px0 > 0 ? sx : 0, r = px0 + bw < W ? sx : 0, t = py0 > 0 ? sy : 0, b = py0 + bh < H ? sy : 0;
    if (bw - l - r <= 0 || bh - t - b <= 0) return;               // (cannot happen for a face that owns a pixel; never sweep nothing on a rounding doubt)
    px0 += l; bw -= l + r; py0 += t; bh -= t + b;
}

// fused recon_data totals of image b: {sum|pi-gi|, sum p*g, sum p+g-p*g} (exact integer sums of the raster waves' partials)
__device__ inline void loss_totals(const long long* ltot, int b, float& l1, float& up, float& un) {
    long long s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
    for (int k = 0; k < MM_LSUB; ++k) { const long long* r = ltot + ((size_t)b * MM_LSUB + k) * 4; s0 += r[0]; s1 += r[1]; s2 += r[2]; }
    l1 = (float)s0 * (1.f / 4294967296.f); up = (float)s1 * (1.f / 4294967296.f); un = (float)s2 * (1.f / 4294967296.f);   // one rounding each
}

// ... and the fourth: the contour term's sum of squares (MMRenderDesc.fused_contour > 0; zero otherwise)
__device__ inline float loss_contour_total(const long long* ltot, int b) {
    long long s3 = 0;
#pragma unroll
    for (int k = 0; k < MM_LSUB; ++k) s3 += ltot[((size_t)b * MM_LSUB + k) * 4 + 3];
    return (float)s3 * (1.f / 4294967296.f);
}

// Wave-wide sum / max, the result in every lane.  Four DPP steps (VALU cross-lane operands: no LDS crossbar traffic, no
// ds_bpermute latency chain) leave every 16-lane row holding its own total; the four row totals are read as scalars.  Fixed order.
template <int CTRL>
__device__ inline float dpp_move(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false)); }
__device__ inline float row16_sum(float v) {
    v += dpp_move<0xB1>(v);        // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);        // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);       // row_half_mirror
    v += dpp_move<0x140>(v);       // row_mirror
    return v;
}
__device__ inline float lane_value(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ inline float wave_sum(float v) {
    v = row16_sum(v);
    return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
__device__ inline int wave_max_i32(int v) {                      // (same four DPP steps; the result in every lane)
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false)); v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false)); v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ inline unsigned wave_min_u32(unsigned v) {
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false)); v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false)); v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false));
    return min(min((unsigned)__builtin_amdgcn_readlane((int)v, 0), (unsigned)__builtin_amdgcn_readlane((int)v, 16)),
               min((unsigned)__builtin_amdgcn_readlane((int)v, 32), (unsigned)__builtin_amdgcn_readlane((int)v, 48)));
}
__device__ inline unsigned wave_or_u32(unsigned v) {
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false); v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false); v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);
    return ((unsigned)__builtin_amdgcn_readlane((int)v, 0) | (unsigned)__builtin_amdgcn_readlane((int)v, 16)) |
           ((unsigned)__builtin_amdgcn_readlane((int)v, 32) | (unsigned)__builtin_amdgcn_readlane((int)v, 48));
}
__device__ inline uint64_t wave_or_u64(uint64_t v) { return ((uint64_t)wave_or_u32((unsigned)(v >> 32)) << 32) | wave_or_u32((unsigned)v); }
__device__ inline float wave_max(float v) {
    v = fmaxf(v, dpp_move<0xB1>(v)); v = fmaxf(v, dpp_move<0x4E>(v)); v = fmaxf(v, dpp_move<0x141>(v)); v = fmaxf(v, dpp_move<0x140>(v));
    return fmaxf(fmaxf(lane_value(v, 0), lane_value(v, 16)), fmaxf(lane_value(v, 32), lane_value(v, 48)));
}

}  // namespace mm
