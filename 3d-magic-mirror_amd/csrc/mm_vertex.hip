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
    int B, V, F;
    float proj0, proj1, proj2, mult;
    const int32_t* faces;
    const float* vertices;
    const float *azim, *elev, *dist, *bias;
    float* T;
    float4* bbox;
    float4* geo;
    uint64_t* valid;
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
    bool front = false;
    if (f < a.F) {
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
        front = nz >= 0.f;
        const size_t o = (size_t)b * a.F + f;
        a.bbox[o] = make_float4(fminf(fminf(ax, bx), cx), fminf(fminf(ay, by), cy), fmaxf(fmaxf(ax, bx), cx), fmaxf(fmaxf(ay, by), cy));
        a.geo[o * 3 + 0] = make_float4(ax, ay, bx, by);
        a.geo[o * 3 + 1] = make_float4(cx, cy, A.z, Bv.z);
        a.geo[o * 3 + 2] = make_float4(C.z, nz, 0.f, 0.f);
        a.face_normals[o * 3 + 0] = nx; a.face_normals[o * 3 + 1] = ny; a.face_normals[o * 3 + 2] = nz;
    }
    const uint64_t m = __ballot(front);
    const int fbase = blockIdx.x * 256 + (tid & ~63);
    if ((tid & 63) == 0 && fbase < a.F) a.valid[(size_t)b * ((a.F + 63) / 64) + (fbase >> 6)] = m;
}

struct VertexBwdArgs {
    int B, V, F;
    float proj0, proj1, proj2;
    const int32_t* faces;
    const int32_t* vc_offsets;
    const int32_t* vc_items;
    const float* vertices;
    const float *azim, *elev, *dist, *bias;
    const float* dfxy;      // (B,F,3,2)
    const float* dfn;       // (B,F,3)
    const float* gfn;       // (B,F,3) external gradient of attributes['face_normals'] or NULL
    float* grad_vertices;
    float *grad_azim, *grad_elev, *grad_dist, *grad_bias;
};

#define MM_VB_THREADS 512

__global__ __launch_bounds__(MM_VB_THREADS) void vertex_bwd_kernel(VertexBwdArgs a) {
    __shared__ float s_trig[4];
    __shared__ Camera s_cam;
    __shared__ float s_red[MM_VB_THREADS / 64][12];
    const int b = blockIdx.x, tid = threadIdx.x;
    block_camera(a.azim, a.elev, a.dist, a.bias, b, s_trig, &s_cam);
    float T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = s_cam.T[i];
    const float* vb = a.vertices + (size_t)b * a.V * 3;
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.f;

    for (int v = tid; v < a.V; v += MM_VB_THREADS) {
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
            for (int j = 0; j < 3; ++j) acc[i * 3 + j] += p[i] * d[j];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[9 + j] += d[j];
    }
    // dT: wave butterfly, then a fixed-order sum over the waves of the workgroup
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = wave_sum(acc[i]);
    if ((tid & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) s_red[tid >> 6][i] = acc[i];
    }
    __syncthreads();
    if (tid == 0) {
        float dT[12];
        for (int i = 0; i < 12; ++i) {
            float s = 0.f;
            for (int w = 0; w < MM_VB_THREADS / 64; ++w) s += s_red[w][i];
            dT[i] = s;
        }
        float dd, de, da, db[2];
        camera_backward(a.dist[b], s_cam, dT, &dd, &de, &da, db);
        a.grad_dist[b] = dd; a.grad_elev[b] = de; a.grad_azim[b] = da;
        a.grad_bias[2 * b] = db[0]; a.grad_bias[2 * b + 1] = db[1];
    }
}

int launch_vertex_fwd(const MMRenderDesc* d, const Workspace& w, hipStream_t s) {
    VertexFwdArgs a;
    a.B = d->B; a.V = d->V; a.F = d->F;
    a.proj0 = d->proj[0]; a.proj1 = d->proj[1]; a.proj2 = d->proj[2]; a.mult = d->multiplier;
    a.faces = d->faces; a.vertices = d->vertices;
    a.azim = d->azimuths; a.elev = d->elevations; a.dist = d->distances; a.bias = d->biases;
    a.T = w.T; a.bbox = w.bbox; a.geo = w.geo; a.valid = w.valid; a.face_normals = d->face_normals;
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
    a.dfxy = w.dfxy; a.dfn = w.dfn; a.gfn = g->grad_face_normals;
    a.grad_vertices = g->grad_vertices;
    a.grad_azim = g->grad_azimuths; a.grad_elev = g->grad_elevations; a.grad_dist = g->grad_distances; a.grad_bias = g->grad_biases;
    { ProfScope ps(d->prof_events, MM_PROF_VERTEX_BWD, s);
      hipLaunchKernelGGL(vertex_bwd_kernel, dim3(d->B), dim3(MM_VB_THREADS), 0, s, a); }
    return hipGetLastError() == hipSuccess ? MM_OK : MM_ERR_LAUNCH;
}

}  // namespace mm
