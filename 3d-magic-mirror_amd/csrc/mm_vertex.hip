// mm_vertex.hip -- vertex stage of the render path (forward and backward) for gfx950.
//
// Forward  (replaces smr_utils camera math + kaolin prepare_vertices + face_normals, networks.py:278-295):
//   one thread per (image, face): camera transform (built once per workgroup), 3 vertex transforms, perspective
//   divide, x multiplier, inflated pixel box, unit normal.  Writes the packed face records the pixel stage streams plus
//   attributes['face_normals'], and -- the 64 faces of a wave being exactly one word of the screen-bin candidate mask --
//   that mask: per block of 8x8 bins one wave transpose of the lanes' coverage bits (no separate binning pass).
// Backward (reverse of the above): one workgroup per image walks the static vertex->corner CSR, so per-vertex
//   gradients are gathered in a fixed order (no atomics), reduces dT in LDS and finishes with the camera chain.
#include "mm_device.h"

MM_PP_STORAGE(vertex_fwd)       // 0 counters cleared, 1 camera (fp64 trig + look-at), 2 face records, 3 binning
MM_PP_STORAGE(vertex_bwd)       // 0 loads of T + vertex, 1 corner gather, 2 group / wave reductions + partial store, 3 ticket, 4 last workgroup: lights, 5 camera chain

namespace mm {

struct VertexFwdArgs {
    int B, V, F, H, W;
    float proj0, proj1, proj2, mult, infl;
    const int32_t* faces;
    const float* vertices;
    const float *azim, *elev, *dist, *bias;
    float* T;
    float* cam;
    float4* geo;
    float* face_normals;
    int* tcnt; int ntcnt;    // texture-record counters of the backward: cleared here for the first backward after this forward
    long long* ltot; int nltot;   // fused-loss sums of the raster waves: cleared here
    int bin_shift, nbx, nby, words;
    uint64_t* mask;          // (B,nbins,words) screen-bin candidate mask, written here (nullptr: not wanted)
    int* fflag;              // (B,F) "this face receives gradient from the pixels": cleared here, set by raster_fwd
    unsigned* ticket;        // (B) arrival counter of the vertex backward's workgroups: cleared here (and by its last workgroup after use)
};

__device__ inline void block_camera(const float* azim, const float* elev, const float* dist, const float* bias, int b,
                                    float* s_trig, Camera* s_cam) {
    const int tid = threadIdx.x;
    if (tid < 4) {
        const float ang = MM_DEG2RAD * (tid < 2 ? elev[b] : azim[b]);
        // fp64 sin/cos rounded to fp32: the oracle does the same, so both sides see correctly rounded values
#ifdef MM_BOUND_FAST_TRIG                                       // BOUND EXPERIMENT (wrong last bits, never in the product): what the fp64 library trig costs the forward's head
        s_trig[tid] = (tid & 1) ? __sinf(ang) : __cosf(ang);
#else
        s_trig[tid] = (tid & 1) ? (float)sin((double)ang) : (float)cos((double)ang);
#endif
    }
    __syncthreads();
    if (tid == 0) camera_build(dist[b], s_trig[0], s_trig[1], s_trig[2], s_trig[3], bias[2 * b], bias[2 * b + 1], *s_cam);
    __syncthreads();
}

// one face: camera transform of its three vertices, perspective divide, x multiplier, unit normal, inflated pixel box -> the packed
// record the pixel stage streams (geo), attributes['face_normals'], cleared face flags.  The expressions of SURVEY 8(a)-a5, in their order.
__device__ inline void face_record(const VertexFwdArgs& a, int b, int f, const float* pa, const float* pb, const float* pc, const float* T,
                                   int& bx0, int& by0, int& bw, int& bh) {
    const Float3 A = to_camera(pa, T);
    const Float3 Bv = to_camera(pb, T);
    const Float3 C = to_camera(pc, T);
    // perspective_camera: (x*px)/(z*pz), then x multiplier (kaolin rasterises in multiplier units)
    const float apz = A.z * a.proj2, bpz = Bv.z * a.proj2, cpz = C.z * a.proj2;
    const float ax = ((A.x * a.proj0) / apz) * a.mult, ay = ((A.y * a.proj1) / apz) * a.mult;
    const float bx = ((Bv.x * a.proj0) / bpz) * a.mult, by = ((Bv.y * a.proj1) / bpz) * a.mult;
    const float cx = ((C.x * a.proj0) / cpz) * a.mult, cy = ((C.y * a.proj1) / cpz) * a.mult;
    const float e0[3] = {Bv.x - A.x, Bv.y - A.y, Bv.z - A.z};
    const float e1[3] = {C.x - A.x, C.y - A.y, C.z - A.z};
    float n[3];
    cross3(e0, e1, n);
    const float len = sqrtf((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
    const float den = len + 1e-10f;
    const float nx = n[0] / den, ny = n[1] / den, nz = n[2] / den;
    const size_t o = (size_t)b * a.F + f;
    a.geo[o * 3 + 0] = make_float4(ax, ay, bx, by);
    a.geo[o * 3 + 1] = make_float4(cx, cy, A.z, Bv.z);
    // the pixel box of the face inflated by the soft-mask margin (conservative, see pixel_range), packed for the backward's
    // face sweep: x = px0 | py0 << 16, y = width | height << 16 (0 x 0 if it misses the image)
    unsigned org, ext;
    face_pixel_box(ax, ay, bx, by, cx, cy, a.infl, a.mult, a.W, a.H, bx0, by0, bw, bh, org, ext);
    a.geo[o * 3 + 2] = make_float4(C.z, nz, __uint_as_float(org), __uint_as_float(ext));
    a.face_normals[o * 3 + 0] = nx; a.face_normals[o * 3 + 1] = ny; a.face_normals[o * 3 + 2] = nz;
    reinterpret_cast<int2*>(a.fflag)[o] = make_int2(0, 0);
}

__global__ __launch_bounds__(256) void vertex_fwd_kernel(VertexFwdArgs a) {
    __shared__ float s_trig[4];
    __shared__ Camera s_cam;
    const int b = blockIdx.y, tid = threadIdx.x;
    MM_PP_BEGIN();
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + tid; i < a.ntcnt; i += gridDim.x * gridDim.y * 256) a.tcnt[i] = 0;
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + tid; i < a.nltot; i += gridDim.x * gridDim.y * 256) a.ltot[i] = 0;
    MM_PP_MARK(0);
    // this lane's face: its corner ids and their raw positions depend on nothing the camera produces -- both trips to memory are in
    // flight while four lanes do the fp64 trigonometry and one builds the look-at
    const int f = blockIdx.x * 256 + tid;
    float pa[3] = {0.f, 0.f, 0.f}, pb[3] = {0.f, 0.f, 0.f}, pc[3] = {0.f, 0.f, 0.f};
    if (f < a.F) {
        const int i0 = a.faces[f * 3 + 0], i1 = a.faces[f * 3 + 1], i2 = a.faces[f * 3 + 2];
        const float* vb = a.vertices + (size_t)b * a.V * 3;
#pragma unroll
        for (int j = 0; j < 3; ++j) { pa[j] = vb[(size_t)i0 * 3 + j]; pb[j] = vb[(size_t)i1 * 3 + j]; pc[j] = vb[(size_t)i2 * 3 + j]; }
    }
    block_camera(a.azim, a.elev, a.dist, a.bias, b, s_trig, &s_cam);
    MM_PP_MARK(1);
    if (blockIdx.x == 0 && tid < 12) a.T[b * 12 + tid] = s_cam.T[tid];
    if (blockIdx.x == 0 && tid == 12) a.ticket[b] = 0u;
    if (blockIdx.x == 0 && tid >= 64 && tid < 100) a.cam[b * 48 + tid - 64] = reinterpret_cast<const float*>(&s_cam)[tid - 64];
    float T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = s_cam.T[i];

    int bx0 = 0, by0 = 0, bw = 0, bh = 0;                         // inflated pixel box of this lane's face (none)
    if (f < a.F) face_record(a, b, f, pa, pb, pc, T, bx0, by0, bw, bh);

    // ---- screen binning: this wave's 64 faces are exactly mask word c (bin_wave_faces, mm_device.h) ---------------------------
    MM_PP_MARK(2);
    const int c = blockIdx.x * 4 + (tid >> 6);
    if (a.mask != nullptr && c < a.words) bin_wave_faces(a.mask, b, a.nbx, a.nby, a.words, a.bin_shift, c, tid & 63, bx0, by0, bw, bh);
    MM_PP_MARK(3);
    MM_PP_FLUSH(vertex_fwd, (long long)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + (tid >> 6));
}

struct VertexBwdArgs {
    int B, V, F;
    float proj0, proj1, proj2;
    const int4* vc_table;   // (V,vc_stride) {face*3 + corner, the face's three vertex ids}, padded with -1
    int vc_stride;
    const int32_t* faces;   // (F,3) vertex ids (the per-image form walks the faces)
    const float* vertices;
    const float *azim, *elev, *dist, *bias;
    const float* T;         // (B,12) saved by the forward
    const float* cam;       // (B,48) the forward's Camera record
    const int2* chunkmap;   // (B,F) {first sweep item, items} of every face
    const float* part;      // (B,item_cap,12) the items' partial sums: dL/d(face xy) (6), dL/d(unit normal) (3)
    int item_cap;
    const float* gfn;       // (B,F,3) external gradient of attributes['face_normals'] or NULL
    float* dTpart;          // (B,groups,12) per-workgroup partial sums of dL/dT
    unsigned* ticket;       // (B) zeroed arrival counter
    int* tcnt; int ntcnt;   // texture-record counters, consumed by the gather before this kernel: cleared for the next backward
    const float* dl_part;   // (B,blocks,12) partial dL/dlights of the pixel backward
    int geometry_only;      // nothing was rasterised (MMRenderDesc.geometry_only): no face has sweep items, no light gradient is written
    int blocks_per_image;
    float* grad_lights;
    float* grad_vertices;
    float *grad_azim, *grad_elev, *grad_dist, *grad_bias;
};

#ifndef MM_VBWD_ROWS
#define MM_VBWD_ROWS 4      // item rows per corner and trip (2: -2.5 us at 256x256, -1.5 at 512x512 where faces have several items; equal at 128x128: r06_batch_walk_flags_ab.md)
#endif
// Eight lanes per vertex, grid (ceil(V/32), B).  Per-vertex gradients are gathered through the static vertex->corner
// CSR (no atomics); dT is reduced per workgroup and added to the image's accumulator; the LAST
// workgroup of an image to arrive (agent-scope release / ticket / acquire, cdna_hip_programming.md G16) runs the
// camera chain.
__global__ __launch_bounds__(256) void vertex_bwd_kernel(VertexBwdArgs a) {
    __shared__ Camera s_cam;
    __shared__ float s_red[4][12];
    __shared__ float s_part[21][12];
    __shared__ int s_last;
    const int b = blockIdx.y, tid = threadIdx.x;
    MM_PP_BEGIN();
#ifndef MM_DBG_KEEP_TCNT                                            // (profiles/tools/tex_records.py reads the counts of the step it ran)
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + tid; i < a.ntcnt; i += gridDim.x * gridDim.y * 256) a.tcnt[i] = 0;
#endif
    float T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = a.T[b * 12 + i];
    const float* vb = a.vertices + (size_t)b * a.V * 3;
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.f;

    // 8 lanes per vertex: lane c of the group takes corners c, c+8, ... of the vertex (valence is ~6), then a 3-step
    // butterfly inside the group; 32 vertices per workgroup
    const int v = blockIdx.x * 32 + (tid >> 3), cl = tid & 7;
    if (v < a.V) {
        const float p[3] = {vb[v * 3], vb[v * 3 + 1], vb[v * 3 + 2]};
        const Float3 me = to_camera(p, T);
        const float pz = me.z * a.proj2;
        const float ipz = 1.f / pz;                               // one division per vertex, not three per corner
        const float xi = (me.x * a.proj0) * ipz, yi = (me.y * a.proj1) * ipz;
        float d[3] = {0.f, 0.f, 0.f};
        MM_PP_MARK(0);
        // The vertex's corners from the fixed-stride table: its address depends on nothing but the vertex, so it travels with T and the
        // vertex itself, and the entry carries the face's three vertex ids -- the CSR (offsets -> items -> faces) was two trips more.
        for (int sl8 = cl; sl8 < a.vc_stride; sl8 += 8) {
            const int4 ent = a.vc_table[(size_t)v * a.vc_stride + sl8];
            if (ent.x < 0) break;                                 // padding: this vertex has no further corner
            const int item = ent.x;
            const int f = item / 3, k = item - f * 3;
            const size_t o = (size_t)b * a.F + f;
            // the face's gradients = its sweep items' partial sums, added in index order (one item for most faces)
            float gx = 0.f, gy = 0.f, g[3] = {0.f, 0.f, 0.f};
#ifndef MM_DBG_VBWD_SKIP
#define MM_DBG_VBWD_SKIP 0       // traffic break-down builds (WRONG results, profiles/r06_vertex_bwd_traffic.md): 1 no item rows, 2 no face vertices, 4 no chunk map
#endif
            const int2 cm = (a.geometry_only || (MM_DBG_VBWD_SKIP & 4) != 0) ? make_int2(0, (MM_DBG_VBWD_SKIP & 4) ? 1 : 0) : a.chunkmap[o];
            // the face's three vertices ride along with chunkmap: loaded whether or not the normal gradient below turns out to be zero
            // -- inside that branch they would cost a dependent trip to memory of their own
            const int i0 = ent.y, i1 = ent.z, i2 = ent.w;
            float pa[3], pb[3], pc[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (MM_DBG_VBWD_SKIP & 2) { pa[j] = p[j]; pb[j] = p[j] + 1.f; pc[j] = p[j] - 1.f; }
                else { pa[j] = vb[(size_t)i0 * 3 + j]; pb[j] = vb[(size_t)i1 * 3 + j]; pc[j] = vb[(size_t)i2 * 3 + j]; }
            }
            const float* part = a.part + ((size_t)b * a.item_cap + cm.x) * 12;
            // the first MM_VBWD_ROWS items' sums in ONE trip (clamped addresses, selected afterwards: a loop over a per-lane count costs a dependent
            // trip per item); the rest MM_VBWD_ROWS at a time.  Added in index order either way.  (A face has ONE item at 128x128 and the spare rows are
            // re-reads of it that hit the L1: two rows instead of four were measured equal there and slower where faces have several items.)
            float pk[MM_VBWD_ROWS][5];
#pragma unroll
            for (int c = 0; c < MM_VBWD_ROWS; ++c) {
                const float* pc4 = a.part + ((size_t)b * a.item_cap + min(cm.x + min(c, max(cm.y - 1, 0)), a.item_cap - 1)) * 12;   // (a face without items still addresses a valid row)
                if (MM_DBG_VBWD_SKIP & 1) { pk[c][0] = pk[c][1] = pk[c][2] = pk[c][3] = pk[c][4] = (float)item; continue; }
                pk[c][0] = pc4[k * 2]; pk[c][1] = pc4[k * 2 + 1]; pk[c][2] = pc4[6]; pk[c][3] = pc4[7]; pk[c][4] = pc4[8];
            }
#pragma unroll
            for (int c = 0; c < MM_VBWD_ROWS; ++c) {
                if (c < cm.y) { gx += pk[c][0]; gy += pk[c][1]; g[0] += pk[c][2]; g[1] += pk[c][3]; g[2] += pk[c][4]; }
            }
            for (int c0 = MM_VBWD_ROWS; c0 < cm.y; c0 += MM_VBWD_ROWS) {   // (big faces -- a close-up, a crumpled fine mesh)
#pragma unroll
                for (int c = 0; c < MM_VBWD_ROWS; ++c) {
                    const float* pc4 = part + (size_t)min(c0 + c, cm.y - 1) * 12;
                    pk[c][0] = pc4[k * 2]; pk[c][1] = pc4[k * 2 + 1]; pk[c][2] = pc4[6]; pk[c][3] = pc4[7]; pk[c][4] = pc4[8];
                }
#pragma unroll
                for (int c = 0; c < MM_VBWD_ROWS; ++c) {
                    if (c0 + c < cm.y) { gx += pk[c][0]; gy += pk[c][1]; g[0] += pk[c][2]; g[1] += pk[c][3]; g[2] += pk[c][4]; }
                }
            }
            // through face_vertices_image
            d[0] += gx * a.proj0 * ipz;
            d[1] += gy * a.proj1 * ipz;
            d[2] += -(gx * xi + gy * yi) * a.proj2 * ipz;
            // through the unit face normal
            if (a.gfn) { g[0] += a.gfn[o * 3]; g[1] += a.gfn[o * 3 + 1]; g[2] += a.gfn[o * 3 + 2]; }
            if (g[0] != 0.f || g[1] != 0.f || g[2] != 0.f) {
                const Float3 A = to_camera(pa, T), Bv = to_camera(pb, T), C = to_camera(pc, T);
                const float e0[3] = {Bv.x - A.x, Bv.y - A.y, Bv.z - A.z};
                const float e1[3] = {C.x - A.x, C.y - A.y, C.z - A.z};
                float n[3];
                cross3(e0, e1, n);
                const float len = sqrtf((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
                const float den = len + 1e-10f;
                const float ng = (n[0] * g[0] + n[1] * g[1]) + n[2] * g[2];
                // d/dn of g . n / (|n| + 1e-10):  g / den - (n . g) / den^2 * n / |n|   (two divisions for the three components)
                const float iden = 1.f / den, c = (len > 0.f) ? (ng * iden * iden) / len : 0.f;
                float dn[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) dn[j] = g[j] * iden - c * n[j];
                float de0[3], de1[3];
                cross3(e1, dn, de0);
                cross3(dn, e0, de1);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (k == 0) d[j] -= de0[j] + de1[j];
                    else if (k == 1) d[j] += de0[j];
                    else d[j] += de1[j];
                }
            }
        }
        MM_PP_MARK(1);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            d[j] += xchg_f32<4>(d[j], tid); d[j] += xchg_f32<2>(d[j], tid); d[j] += xchg_f32<1>(d[j], tid);
        }
        if (cl == 0) {
            float* gv = a.grad_vertices + ((size_t)b * a.V + v) * 3;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                gv[i] = (T[i * 3 + 0] * d[0] + T[i * 3 + 1] * d[1]) + T[i * 3 + 2] * d[2];
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[i * 3 + j] = p[i] * d[j];
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[9 + j] = d[j];
        }
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = wave_sum(acc[i]);
    if ((tid & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) s_red[tid >> 6][i] = acc[i];
    }
    __syncthreads();
    if (tid < 12) {
        // This workgroup's partial of dL/dT, WRITE-THROUGH (agent-scope relaxed atomic store = sc1: it lands at the memory side, not in
        // this XCD's L2), so that the image's last workgroup can read it with agent-scope loads and NO fence on either side
        // (cdna_hip_programming.md G16, form "sc1 stores and sc1 loads on both sides").  No float atomics: the sum below runs
        // over the workgroups in index order, so the camera gradients are bitwise reproducible.
        __hip_atomic_store(a.dTpart + ((size_t)b * gridDim.x + blockIdx.x) * 12 + tid,
                           ((s_red[0][tid] + s_red[1][tid]) + s_red[2][tid]) + s_red[3][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (blockIdx.x == 0 && !a.geometry_only) {   // dL/dlights (needs nothing of this kernel: the image's FIRST workgroup adds it up, off the last one's tail): sum of the pixel-backward workgroup partials; waves 1..3 take 3 components each, lanes stride over
        // the partials (independent loads), fixed butterfly order
        const int wv = tid >> 6, ln = tid & 63;
        if (wv >= 1) {
            float sum[3] = {0.f, 0.f, 0.f};                      // the three components' loads of four rows in flight together
            for (int k0 = ln; k0 < a.blocks_per_image; k0 += 256) {
                float r[4][3];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k0 + 64 * u;
                    const float* row = a.dl_part + ((size_t)b * a.blocks_per_image + min(k, a.blocks_per_image - 1)) * 12 + (wv - 1) * 3;
                    const bool ok = k < a.blocks_per_image;
                    r[u][0] = ok ? row[0] : 0.f; r[u][1] = ok ? row[1] : 0.f; r[u][2] = ok ? row[2] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) { sum[0] += r[u][0]; sum[1] += r[u][1]; sum[2] += r[u][2]; }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                sum[i] = wave_sum(sum[i]);
                if (ln == 0) a.grad_lights[b * 9 + (wv - 1) * 3 + i] = sum[i];
            }
        }
    }
    // ---- publish, take a ticket; the last workgroup of this image finishes the camera chain
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the storing wave drains its stores before the ticket is drawn
    __syncthreads();
    MM_PP_MARK(2);
    if (tid == 0) {
        // No agent-scope fences here: a release fence writes back the XCD's whole L2 and an acquire invalidates it, once per
        // workgroup (removing them took this kernel from 213 to 53 us at B=384).
        const unsigned prev = __hip_atomic_fetch_add(a.ticket + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (prev == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    MM_PP_MARK(3);
    if (!s_last) { MM_PP_FLUSH(vertex_bwd, (long long)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + (tid >> 6)); return; }
    MM_PP_MARK(4);
    if (tid >= 64 && tid < 100) reinterpret_cast<float*>(&s_cam)[tid - 64] = a.cam[b * 48 + tid - 64];   // the forward's camera (no trig here)
    {   // dL/dT = sum of the workgroups' partials in a FIXED order: thread (gl, comp) adds rows gl, gl + 21, gl + 42, ... of its component --
        // all of its loads in flight together: ONE trip to memory however many workgroups the image has (6 890 vertices: 216 rows; a pass of
        // 21 rows at a time was eleven dependent trips, ~20 us of serial tail per image) -- then twelve threads add the 21 partial sums in
        // index order.  No float atomics, the same order every run: bitwise reproducible.
        const int gl = tid / 12, comp = tid - gl * 12;
        float sum = 0.f;
        if (gl < 21) {
            for (unsigned g0 = gl; g0 < gridDim.x; g0 += 21 * 8) {
                float r[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const unsigned g = g0 + 21u * u;
                    r[u] = g < gridDim.x ? __hip_atomic_load(a.dTpart + ((size_t)b * gridDim.x + g) * 12 + comp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) sum += r[u];
            }
            s_part[gl][comp] = sum;
        }
        __syncthreads();
        if (tid < 12) {
            float tot = 0.f;
            for (int k = 0; k < 21; ++k) tot += s_part[k][tid];
            s_red[0][tid] = tot;
        }
    }
    __syncthreads();
    if (tid == 0) {
        float dT[12];
        for (int i = 0; i < 12; ++i) dT[i] = s_red[0][i];
        float dd, de, da, db[2];
        camera_backward(a.dist[b], s_cam, dT, &dd, &de, &da, db);
        a.grad_dist[b] = dd; a.grad_elev[b] = de; a.grad_azim[b] = da;
        a.grad_bias[2 * b] = db[0]; a.grad_bias[2 * b + 1] = db[1];
        a.ticket[b] = 0u;                                        // every workgroup of the image has drawn: ready for the next backward on this workspace
    }
    MM_PP_MARK(5);
    MM_PP_FLUSH(vertex_bwd, (long long)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + (tid >> 6));
}

// ---------------------------------------------------------------------------------------------------------------------
// The vertex backward PER IMAGE, for small templates (at most MM_VIMG_BWD_MAX_FACES faces): ONE workgroup of 1024 threads per image, two
// phases through LDS instead of the ticket / partial-row round trip of vertex_bwd_kernel:
//   A. face-major   a thread per face adds up the face's sweep-item sums (one item for most faces: the first two rows travel with the face's
//                   vertices) and turns them ONCE per face -- not once per corner -- into dL/d(camera-space position) of its three corners:
//                   the screen-space part through the perspective divide, the normal part through the cross product.  Nine floats per face
//                   into LDS.
//   B. vertex-major a thread per vertex adds its corners up in the static table's (ascending) order, writes dL/dvertex and its twelve terms
//                   of dL/dT; one fixed-order workgroup reduction; thread 0 runs the camera chain from the forward's saved record.
// No atomics, no ticket, no write-through partials, no second trip for the image's last workgroup: the chain is {ids, chunk map} ->
// {vertices, item sums} -> LDS -> outputs.  Bitwise reproducible (fixed orders throughout).
// ---------------------------------------------------------------------------------------------------------------------
// (MM_VIMG_BWD_MAX_FACES = 1700, mm_device.h)   36 bytes of LDS per face: 61 KB of dynamic LDS at most, below the 64 KB a launch gets without opting in
__global__ __launch_bounds__(1024) void vertex_image_bwd_kernel(VertexBwdArgs a) {
    extern __shared__ float s_d[];                                // (F, 3 corners, 3)
    __shared__ Camera s_cam;
    __shared__ float s_red[16][12];
    const int b = blockIdx.x, tid = threadIdx.x;
#ifndef MM_DBG_KEEP_TCNT
    for (int i = b * 1024 + tid; i < a.ntcnt; i += gridDim.x * 1024) a.tcnt[i] = 0;
#endif
    const float* vb = a.vertices + (size_t)b * a.V * 3;
    // ---- trip 1: what depends on nothing (T, the camera record, this thread's faces' ids and chunk maps, its vertex and its corner list)
    float T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = a.T[b * 12 + i];
    if (tid >= 64 && tid < 100) reinterpret_cast<float*>(&s_cam)[tid - 64] = a.cam[b * 48 + tid - 64];
    int fid[2][3]; int2 cm[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int f = tid + 1024 * k;
        fid[k][0] = fid[k][1] = fid[k][2] = 0; cm[k] = make_int2(0, 0);
        if (f < a.F) {
            const PackedI3 i3 = *(const PackedI3*)(a.faces + f * 3);
            fid[k][0] = i3.x; fid[k][1] = i3.y; fid[k][2] = i3.z;
            if (!a.geometry_only) cm[k] = a.chunkmap[(size_t)b * a.F + f];
        }
    }
    int it0[8]; float p0[3] = {0.f, 0.f, 0.f};                    // phase B's first vertex: its corner list and position ride along with trip 1
#pragma unroll
    for (int u = 0; u < 8; ++u) it0[u] = (tid < a.V && u < a.vc_stride) ? a.vc_table[(size_t)tid * a.vc_stride + u].x : -1;
    if (tid < a.V) { const Packed3 v3 = *(const Packed3*)(vb + tid * 3); p0[0] = v3.x; p0[1] = v3.y; p0[2] = v3.z; }
    // ---- phase A
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int f = tid + 1024 * k;
        if (f >= a.F) continue;
        const size_t o = (size_t)b * a.F + f;
        float P[3][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { const Packed3 v3 = *(const Packed3*)(vb + (size_t)fid[k][c] * 3); P[c][0] = v3.x; P[c][1] = v3.y; P[c][2] = v3.z; }
        // the face's gradients = its sweep items' partial sums, added in index order: the first two rows in this trip (clamped addresses,
        // selected afterwards; rows are 48 bytes, 16-byte aligned: three wide loads each), the rare rest one by one
        float r[2][9];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float4* row = (const float4*)(a.part + ((size_t)b * a.item_cap + min(cm[k].x + min(c, max(cm[k].y - 1, 0)), a.item_cap - 1)) * 12);
            const float4 r0 = row[0], r1 = row[1]; const float r2 = a.part[((size_t)b * a.item_cap + min(cm[k].x + min(c, max(cm[k].y - 1, 0)), a.item_cap - 1)) * 12 + 8];
            r[c][0] = r0.x; r[c][1] = r0.y; r[c][2] = r0.z; r[c][3] = r0.w; r[c][4] = r1.x; r[c][5] = r1.y; r[c][6] = r1.z; r[c][7] = r1.w; r[c][8] = r2;
        }
        float gn[3] = {0.f, 0.f, 0.f};
        if (a.gfn) { const Packed3 g3 = *(const Packed3*)(a.gfn + o * 3); gn[0] = g3.x; gn[1] = g3.y; gn[2] = g3.z; }
        float s[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; ++c)
            if (c < cm[k].y) {
#pragma unroll
                for (int j = 0; j < 9; ++j) s[j] += r[c][j];
            }
        for (int c = 2; c < cm[k].y; ++c) {                       // (a close-up: faces of many 128-pixel chunks)
            const float* row = a.part + ((size_t)b * a.item_cap + cm[k].x + c) * 12;
#pragma unroll
            for (int j = 0; j < 9; ++j) s[j] += row[j];
        }
        const float g[3] = {s[6] + gn[0], s[7] + gn[1], s[8] + gn[2]};
        Float3 cam3[3];
        float d[3][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            cam3[c] = to_camera(P[c], T);
            const float pz = cam3[c].z * a.proj2;
            const float ipz = 1.f / pz;
            const float xi = (cam3[c].x * a.proj0) * ipz, yi = (cam3[c].y * a.proj1) * ipz;
            const float gx = s[c * 2], gy = s[c * 2 + 1];
            d[c][0] = gx * a.proj0 * ipz;
            d[c][1] = gy * a.proj1 * ipz;
            d[c][2] = -(gx * xi + gy * yi) * a.proj2 * ipz;
        }
        if (g[0] != 0.f || g[1] != 0.f || g[2] != 0.f) {          // through the unit face normal
            const float e0[3] = {cam3[1].x - cam3[0].x, cam3[1].y - cam3[0].y, cam3[1].z - cam3[0].z};
            const float e1[3] = {cam3[2].x - cam3[0].x, cam3[2].y - cam3[0].y, cam3[2].z - cam3[0].z};
            float n[3];
            cross3(e0, e1, n);
            const float len = sqrtf((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
            const float den = len + 1e-10f;
            const float ng = (n[0] * g[0] + n[1] * g[1]) + n[2] * g[2];
            const float iden = 1.f / den, cc = (len > 0.f) ? (ng * iden * iden) / len : 0.f;
            float dn[3], de0[3], de1[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) dn[j] = g[j] * iden - cc * n[j];
            cross3(e1, dn, de0);
            cross3(dn, e0, de1);
#pragma unroll
            for (int j = 0; j < 3; ++j) { d[0][j] -= de0[j] + de1[j]; d[1][j] += de0[j]; d[2][j] += de1[j]; }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int j = 0; j < 3; ++j) s_d[(f * 3 + c) * 3 + j] = d[c][j];
    }
    __syncthreads();
    // ---- phase B
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.f;
    for (int v = tid; v < a.V; v += 1024) {
        const bool first = v == tid;
        float p[3] = {p0[0], p0[1], p0[2]};
        if (!first) { p[0] = vb[v * 3]; p[1] = vb[v * 3 + 1]; p[2] = vb[v * 3 + 2]; }
        float d[3] = {0.f, 0.f, 0.f};
        for (int sl0 = 0; sl0 < a.vc_stride; sl0 += 8) {          // eight entries per trip (valence is ~6)
            int it[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) it[u] = it0[u];
            if (!first || sl0 != 0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) it[u] = sl0 + u < a.vc_stride ? a.vc_table[(size_t)v * a.vc_stride + sl0 + u].x : -1;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (it[u] >= 0) { d[0] += s_d[it[u] * 3]; d[1] += s_d[it[u] * 3 + 1]; d[2] += s_d[it[u] * 3 + 2]; }
        }
        float* gv = a.grad_vertices + ((size_t)b * a.V + v) * 3;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            gv[i] = (T[i * 3 + 0] * d[0] + T[i * 3 + 1] * d[1]) + T[i * 3 + 2] * d[2];
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i * 3 + j] += p[i] * d[j];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[9 + j] += d[j];
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = wave_sum(acc[i]);
    if ((tid & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) s_red[tid >> 6][i] = acc[i];
    }
    // dL/dlights: the pixel backward's per-workgroup partials, added by waves 13..15 (three components each; the last waves hold the fewest
    // faces and vertices), lanes stride over the rows, fixed butterfly order
    {
        const int wv = tid >> 6, ln = tid & 63;
        if (wv >= 13 && !a.geometry_only) {
            float sum[3] = {0.f, 0.f, 0.f};
            for (int k0 = ln; k0 < a.blocks_per_image; k0 += 256) {
                float r4[4][3];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kk = k0 + 64 * u;
                    const float* row = a.dl_part + ((size_t)b * a.blocks_per_image + min(kk, a.blocks_per_image - 1)) * 12 + (wv - 13) * 3;
                    const bool ok = kk < a.blocks_per_image;
                    r4[u][0] = ok ? row[0] : 0.f; r4[u][1] = ok ? row[1] : 0.f; r4[u][2] = ok ? row[2] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) { sum[0] += r4[u][0]; sum[1] += r4[u][1]; sum[2] += r4[u][2]; }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                sum[i] = wave_sum(sum[i]);
                if (ln == 0) a.grad_lights[b * 9 + (wv - 13) * 3 + i] = sum[i];
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        float dT[12];
        for (int i = 0; i < 12; ++i) {
            float tot = 0.f;
            for (int w = 0; w < 16; ++w) tot += s_red[w][i];
            dT[i] = tot;
        }
        float dd, de, da, db[2];
        camera_backward(a.dist[b], s_cam, dT, &dd, &de, &da, db);
        a.grad_dist[b] = dd; a.grad_elev[b] = de; a.grad_azim[b] = da;
        a.grad_bias[2 * b] = db[0]; a.grad_bias[2 * b + 1] = db[1];
    }
}

int launch_vertex_fwd(const MMRenderDesc* d, const Workspace& w, hipStream_t s) {
    VertexFwdArgs a;
    a.B = d->B; a.V = d->V; a.F = d->F; a.H = d->H; a.W = d->W;
    a.proj0 = d->proj[0]; a.proj1 = d->proj[1]; a.proj2 = d->proj[2]; a.mult = d->multiplier; a.infl = d->boxlen * d->multiplier;
    a.faces = d->faces; a.vertices = d->vertices;
    a.azim = d->azimuths; a.elev = d->elevations; a.dist = d->distances; a.bias = d->biases;
    a.T = w.T; a.cam = w.cam; a.geo = w.geo; a.face_normals = d->face_normals;
    a.tcnt = w.tcnt; a.ntcnt = w.ntcnt + d->B + d->B * w.ntiles;   // (+ the status words and the forward's per-tile counts: cleared by the forward only)
    a.ltot = w.ltot; a.nltot = d->B * MM_LSUB * 4;
    a.bin_shift = w.bin_shift; a.nbx = w.nbx; a.nby = w.nby; a.words = w.words;
    a.mask = d->geometry_only ? nullptr : w.binmask; a.fflag = w.fflag;      // (geometry only: nothing walks the screen bins)
    a.ticket = w.ticket;
    dim3 grid((d->F + 255) / 256, d->B);
    { ProfScope ps(d->prof_events, MM_PROF_VERTEX_FWD, s);
      hipLaunchKernelGGL(vertex_fwd_kernel, grid, dim3(256), 0, s, a); }
    return launch_ok("vertex_fwd");
}

int launch_vertex_bwd(const MMRenderDesc* d, const MMRenderGrads* g, const Workspace& w, hipStream_t s) {
    VertexBwdArgs a;
    a.B = d->B; a.V = d->V; a.F = d->F;
    a.proj0 = d->proj[0]; a.proj1 = d->proj[1]; a.proj2 = d->proj[2];
    a.vc_table = (const int4*)d->vc_table; a.vc_stride = d->vc_stride; a.vertices = d->vertices;
    a.azim = d->azimuths; a.elev = d->elevations; a.dist = d->distances; a.bias = d->biases;
    a.T = w.T; a.cam = w.cam; a.chunkmap = w.chunkmap; a.part = w.part; a.item_cap = w.item_cap; a.gfn = g->grad_face_normals;
    a.dTpart = w.dTpart; a.ticket = w.ticket;
    a.tcnt = w.tcnt; a.ntcnt = w.ntcnt;
    a.dl_part = w.dl_part; a.blocks_per_image = w.blocks_per_image; a.grad_lights = g->grad_lights;
    a.grad_vertices = g->grad_vertices;
    a.grad_azim = g->grad_azimuths; a.grad_elev = g->grad_elevations; a.grad_dist = g->grad_distances; a.grad_bias = g->grad_biases;
    a.faces = d->faces; a.geometry_only = d->geometry_only;
    if (vertex_bwd_per_image(d->B, d->F, d->vc_stride)) {
        ProfScope ps(d->prof_events, MM_PROF_VERTEX_BWD, s);
        hipLaunchKernelGGL(vertex_image_bwd_kernel, dim3(d->B), dim3(1024), (size_t)d->F * 9 * sizeof(float), s, a);
        return launch_ok("vertex_image_bwd");
    }
    { ProfScope ps(d->prof_events, MM_PROF_VERTEX_BWD, s);
      hipLaunchKernelGGL(vertex_bwd_kernel, dim3((d->V + 31) / 32, d->B), dim3(256), 0, s, a); }
    return launch_ok("vertex_bwd");
}

}  // namespace mm
