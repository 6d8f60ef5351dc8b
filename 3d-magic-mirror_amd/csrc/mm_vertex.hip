// mm_vertex.hip -- vertex stage of the render path (forward and backward) for gfx950.
//
// Forward  (replaces smr_utils camera math + kaolin prepare_vertices + face_normals, networks.py:278-295):
//   one thread per (image, face): camera transform (built once per workgroup), 3 vertex transforms, perspective
//   divide, x multiplier, bbox, unit normal, front-facing bit.  Writes the packed face records the pixel stage
//   streams (bbox | geo) plus attributes['face_normals'].
// Backward (reverse of the above): one workgroup per image walks the static vertex->corner CSR, so per-vertex
//   gradients are gathered in a fixed order (no atomics), reduces dT in LDS and finishes with the camera chain.
#include "mm_device.h"

namespace mm {

struct VertexFwdArgs {
    int B, V, F, H, W;
    int bin_shift, nbx, nby, words;
    float proj0, proj1, proj2, mult, infl;
    const int32_t* faces;
    const float* vertices;
    const float *azim, *elev, *dist, *bias;
    float* T;
    float4* geo;
    uint64_t* binmask;
    float* face_normals;
};

__device__ inline void block_camera(const float* azim, const float* elev, const float* dist, const float* bias, int b,
                                    float* s_trig, Camera* s_cam) {
    const int tid = threadIdx.x;
    if (tid < 4) {
        const float ang = MM_DEG2RAD * (tid < 2 ? elev[b] : azim[b]);
        // fp64 sin/cos rounded to fp32: the oracle does the same, so both sides see correctly rounded values
        s_trig[tid] = (tid & 1) ? (float)sin((double)ang) : (float)cos((double)ang);
    }
    __syncthreads();
    if (tid == 0) camera_build(dist[b], s_trig[0], s_trig[1], s_trig[2], s_trig[3], bias[2 * b], bias[2 * b + 1], *s_cam);
    __syncthreads();
}

// conservative pixel range [lo, hi] whose centres can satisfy  lo_v <= centre <= hi_v  (one pixel of slack either side
// covers the rounding of this closed form; the raster stage re-tests every pixel exactly)
__device__ inline void pixel_range(float lo_v, float hi_v, float mult, int n, bool flip, int& lo, int& hi) {
    float a = (lo_v / mult) * (float)n, c = (hi_v / mult) * (float)n;
    float flo, fhi;
    if (!flip) { flo = (a + (float)(n - 1)) * 0.5f; fhi = (c + (float)(n - 1)) * 0.5f; }      // x: centre grows with px
    else { flo = ((float)(n - 1) - c) * 0.5f; fhi = ((float)(n - 1) - a) * 0.5f; }            // y: centre falls with py
    if (!(fabsf(flo) < 1e9f) || !(fabsf(fhi) < 1e9f)) { lo = 0; hi = n - 1; return; }         // inf / NaN: every pixel
    lo = (int)floorf(flo) - 1; hi = (int)ceilf(fhi) + 1;
    lo = lo < 0 ? 0 : lo; hi = hi > n - 1 ? n - 1 : hi;
}

__global__ __launch_bounds__(256) void vertex_fwd_kernel(VertexFwdArgs a) {
    __shared__ float s_trig[4];
    __shared__ Camera s_cam;
    const int b = blockIdx.y, tid = threadIdx.x;
    block_camera(a.azim, a.elev, a.dist, a.bias, b, s_trig, &s_cam);
    if (blockIdx.x == 0 && tid < 12) a.T[b * 12 + tid] = s_cam.T[tid];
    float T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = s_cam.T[i];

    const int f = blockIdx.x * 256 + tid;
    if (f >= a.F) return;
    const int i0 = a.faces[f * 3 + 0], i1 = a.faces[f * 3 + 1], i2 = a.faces[f * 3 + 2];
    const float* vb = a.vertices + (size_t)b * a.V * 3;
    const Float3 A = to_camera(vb + (size_t)i0 * 3, T);
    const Float3 Bv = to_camera(vb + (size_t)i1 * 3, T);
    const Float3 C = to_camera(vb + (size_t)i2 * 3, T);
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
    a.geo[o * 3 + 2] = make_float4(C.z, nz, 0.f, 0.f);
    a.face_normals[o * 3 + 0] = nx; a.face_normals[o * 3 + 1] = ny; a.face_normals[o * 3 + 2] = nz;

    // screen binning of the soft-mask box (the hard box is inside it)
    const float xmin = fminf(fminf(ax, bx), cx) - a.infl, xmax = fmaxf(fmaxf(ax, bx), cx) + a.infl;
    const float ymin = fminf(fminf(ay, by), cy) - a.infl, ymax = fmaxf(fmaxf(ay, by), cy) + a.infl;
    int px0, px1, py0, py1;
    pixel_range(xmin, xmax, a.mult, a.W, false, px0, px1);
    pixel_range(ymin, ymax, a.mult, a.H, true, py0, py1);
    if (px0 > px1 || py0 > py1) return;                      // entirely off screen
    const int bx0 = px0 >> a.bin_shift, bx1 = px1 >> a.bin_shift, by0 = py0 >> a.bin_shift, by1 = py1 >> a.bin_shift;
    const unsigned long long bit = 1ull << (f & 63);
    unsigned long long* base = (unsigned long long*)a.binmask + (size_t)b * a.nbx * a.nby * a.words + (f >> 6);
    for (int yy = by0; yy <= by1; ++yy)
        for (int xx = bx0; xx <= bx1; ++xx) atomicOr(base + (size_t)(yy * a.nbx + xx) * a.words, bit);
}

struct VertexBwdArgs {
    int B, V, F;
    float proj0, proj1, proj2;
    const int32_t* faces;
    const int32_t* vc_offsets;
    const int32_t* vc_items;
    const float* vertices;
    const float *azim, *elev, *dist, *bias;
    const float* T;         // (B,12) saved by the forward
    const float* dfxy;      // (B,F,3,2)
    const float* dfn;       // (B,F,3)
    const float* gfn;       // (B,F,3) external gradient of attributes['face_normals'] or NULL
    float* dTacc;           // (B,12) zeroed accumulator
    unsigned* ticket;       // (B) zeroed arrival counter
    float* grad_vertices;
    float *grad_azim, *grad_elev, *grad_dist, *grad_bias;
};

// One thread per vertex, grid (ceil(V/256), B).  Per-vertex gradients are gathered through the static vertex->corner
// CSR in a fixed order (no atomics); dT is reduced per workgroup and added to the image's accumulator; the LAST
// workgroup of an image to arrive (agent-scope release / ticket / acquire, cdna_hip_programming.md G16) runs the
// camera chain.
__global__ __launch_bounds__(256) void vertex_bwd_kernel(VertexBwdArgs a) {
    __shared__ float s_trig[4];
    __shared__ Camera s_cam;
    __shared__ float s_red[4][12];
    __shared__ int s_last;
    const int b = blockIdx.y, tid = threadIdx.x;
    float T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = a.T[b * 12 + i];
    const float* vb = a.vertices + (size_t)b * a.V * 3;
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.f;

    const int v = blockIdx.x * 256 + tid;
    if (v < a.V) {
        const float p[3] = {vb[v * 3], vb[v * 3 + 1], vb[v * 3 + 2]};
        const Float3 me = to_camera(p, T);
        const float pz = me.z * a.proj2;
        const float xi = (me.x * a.proj0) / pz, yi = (me.y * a.proj1) / pz;
        float d[3] = {0.f, 0.f, 0.f};
        const int beg = a.vc_offsets[v], end = a.vc_offsets[v + 1];
        for (int it = beg; it < end; ++it) {
            const int item = a.vc_items[it];
            const int f = item / 3, k = item - f * 3;
            const size_t o = (size_t)b * a.F + f;
            // through face_vertices_image
            const float gx = a.dfxy[o * 6 + k * 2], gy = a.dfxy[o * 6 + k * 2 + 1];
            d[0] += gx * a.proj0 / pz;
            d[1] += gy * a.proj1 / pz;
            d[2] += -(gx * xi + gy * yi) * a.proj2 / pz;
            // through the unit face normal
            float g[3] = {a.dfn[o * 3], a.dfn[o * 3 + 1], a.dfn[o * 3 + 2]};
            if (a.gfn) { g[0] += a.gfn[o * 3]; g[1] += a.gfn[o * 3 + 1]; g[2] += a.gfn[o * 3 + 2]; }
            if (g[0] != 0.f || g[1] != 0.f || g[2] != 0.f) {
                const int i0 = a.faces[f * 3], i1 = a.faces[f * 3 + 1], i2 = a.faces[f * 3 + 2];
                const Float3 A = to_camera(vb + (size_t)i0 * 3, T), Bv = to_camera(vb + (size_t)i1 * 3, T), C = to_camera(vb + (size_t)i2 * 3, T);
                const float e0[3] = {Bv.x - A.x, Bv.y - A.y, Bv.z - A.z};
                const float e1[3] = {C.x - A.x, C.y - A.y, C.z - A.z};
                float n[3];
                cross3(e0, e1, n);
                const float len = sqrtf((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
                const float den = len + 1e-10f;
                const float ng = (n[0] * g[0] + n[1] * g[1]) + n[2] * g[2];
                float dn[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float dirj = (len > 0.f) ? n[j] / len : 0.f;
                    dn[j] = g[j] / den - (ng / (den * den)) * dirj;
                }
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
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = wave_sum(acc[i]);
    if ((tid & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) s_red[tid >> 6][i] = acc[i];
    }
    __syncthreads();
    if (tid < 12) atomicAdd(a.dTacc + b * 12 + tid, ((s_red[0][tid] + s_red[1][tid]) + s_red[2][tid]) + s_red[3][tid]);
    // ---- publish, take a ticket; the last workgroup of this image finishes the camera chain
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned prev = __hip_atomic_fetch_add(a.ticket + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (prev == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    block_camera(a.azim, a.elev, a.dist, a.bias, b, s_trig, &s_cam);
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        float dT[12];
        for (int i = 0; i < 12; ++i) dT[i] = __hip_atomic_load(a.dTacc + b * 12 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float dd, de, da, db[2];
        camera_backward(a.dist[b], s_cam, dT, &dd, &de, &da, db);
        a.grad_dist[b] = dd; a.grad_elev[b] = de; a.grad_azim[b] = da;
        a.grad_bias[2 * b] = db[0]; a.grad_bias[2 * b + 1] = db[1];
    }
}

int launch_vertex_fwd(const MMRenderDesc* d, const Workspace& w, hipStream_t s) {
    VertexFwdArgs a;
    a.B = d->B; a.V = d->V; a.F = d->F; a.H = d->H; a.W = d->W;
    a.bin_shift = w.bin_shift; a.nbx = w.nbx; a.nby = w.nby; a.words = w.words;
    a.proj0 = d->proj[0]; a.proj1 = d->proj[1]; a.proj2 = d->proj[2]; a.mult = d->multiplier;
    a.infl = d->boxlen * d->multiplier;
    a.faces = d->faces; a.vertices = d->vertices;
    a.azim = d->azimuths; a.elev = d->elevations; a.dist = d->distances; a.bias = d->biases;
    a.T = w.T; a.geo = w.geo; a.binmask = w.binmask; a.face_normals = d->face_normals;
    dim3 grid((d->F + 255) / 256, d->B);
    { ProfScope ps(d->prof_events, MM_PROF_VERTEX_FWD, s);
      hipLaunchKernelGGL(vertex_fwd_kernel, grid, dim3(256), 0, s, a); }
    return hipGetLastError() == hipSuccess ? MM_OK : MM_ERR_LAUNCH;
}

int launch_vertex_bwd(const MMRenderDesc* d, const MMRenderGrads* g, const Workspace& w, hipStream_t s) {
    VertexBwdArgs a;
    a.B = d->B; a.V = d->V; a.F = d->F;
    a.proj0 = d->proj[0]; a.proj1 = d->proj[1]; a.proj2 = d->proj[2];
    a.faces = d->faces; a.vc_offsets = d->vc_offsets; a.vc_items = d->vc_items; a.vertices = d->vertices;
    a.azim = d->azimuths; a.elev = d->elevations; a.dist = d->distances; a.bias = d->biases;
    a.T = w.T; a.dfxy = w.dfxy; a.dfn = w.dfn; a.gfn = g->grad_face_normals;
    a.dTacc = w.dTacc; a.ticket = w.ticket;
    a.grad_vertices = g->grad_vertices;
    a.grad_azim = g->grad_azimuths; a.grad_elev = g->grad_elevations; a.grad_dist = g->grad_distances; a.grad_bias = g->grad_biases;
    { ProfScope ps(d->prof_events, MM_PROF_VERTEX_BWD, s);
      hipLaunchKernelGGL(vertex_bwd_kernel, dim3((d->V + 255) / 256, d->B), dim3(256), 0, s, a); }
    return hipGetLastError() == hipSuccess ? MM_OK : MM_ERR_LAUNCH;
}

}  // namespace mm
