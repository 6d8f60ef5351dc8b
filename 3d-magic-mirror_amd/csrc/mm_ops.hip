// mm_ops.hip -- the small kaolin operators of the reference's import boundary as stand-alone gfx950 kernels:
//   kaolin.render.mesh.prepare_vertices            (call site /root/reference/networks.py:284-287)
//   kaolin.ops.mesh.face_normals                   (:289)
//   kaolin.render.mesh.texture_mapping             (:305)   = F.grid_sample(align_corners=False, padding_mode='border') on (2u-1, -(2v-1))
//   kaolin.render.mesh.spherical_harmonic_lighting (:306)
//   kaolin.metrics.render.mask_iou                 (:377, trainer.py:793,933)
// Upstream these are compositions of ATen ops (10-20 launches each); here each is one launch per direction (two where a
// fixed-order reduction needs a second, tiny one), with the same device functions the fused render kernels use (to_camera,
// bilin_setup, sh_bands), so values agree bit for bit with the fused path's intermediate quantities.
// Semantics: SURVEY.md 8(a) rows a5, a6, a9, a10, a14; gradients Appendix A.3 and exact chain rules of the forwards.
#include <algorithm>
#include "mm_device.h"

namespace mm {

// ---------------------------------------------------------------------------------------------------------------------
// prepare_vertices
// ---------------------------------------------------------------------------------------------------------------------
// camera_proj: the three floats of the descriptor, or of the caller's device tensor (MMPrepareDesc.proj_device)
__device__ inline void prepare_proj(const MMPrepareDesc& d, float* pj) {
    if (d.proj_device) { pj[0] = d.proj_device[0]; pj[1] = d.proj_device[1]; pj[2] = d.proj_device[2]; }
    else { pj[0] = d.proj[0]; pj[1] = d.proj[1]; pj[2] = d.proj[2]; }
}
__device__ inline int clamp_vertex(int i, int V) { return min(max(i, 0), V - 1); }      // (ids outside the cloud are reported by the CSR builder, never dereferenced)

__global__ __launch_bounds__(256) void prepare_fwd_kernel(MMPrepareDesc d) {
    const int b = blockIdx.y, f = blockIdx.x * 256 + threadIdx.x;
    if (f >= d.F) return;
    float pj[3];
    prepare_proj(d, pj);
    float T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = d.transform[b * 12 + i];
    const float* vb = d.vertices + (size_t)b * d.V * 3;
    const size_t o = (size_t)b * d.F + f;
    Float3 c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        c[k] = to_camera(vb + (size_t)clamp_vertex(d.faces[f * 3 + k], d.V) * 3, T);
        const float pz = c[k].z * pj[2];
        float* fc = d.face_vertices_camera + (o * 3 + k) * 3;
        fc[0] = c[k].x; fc[1] = c[k].y; fc[2] = c[k].z;
        d.face_vertices_image[(o * 3 + k) * 2 + 0] = (c[k].x * pj[0]) / pz;
        d.face_vertices_image[(o * 3 + k) * 2 + 1] = (c[k].y * pj[1]) / pz;
    }
    const float e0[3] = {c[1].x - c[0].x, c[1].y - c[0].y, c[1].z - c[0].z};
    const float e1[3] = {c[2].x - c[0].x, c[2].y - c[0].y, c[2].z - c[0].z};
    float n[3];
    cross3(e0, e1, n);
    const float len = sqrtf((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
    const float den = len + 1e-10f;
    d.face_normals[o * 3 + 0] = n[0] / den; d.face_normals[o * 3 + 1] = n[1] / den; d.face_normals[o * 3 + 2] = n[2] / den;
}

// d/d(corner positions) of g . n / (|n| + 1e-10), n = (B - A) x (C - A): adds to dA, dB, dC
__device__ inline void normal_backward(const Float3& A, const Float3& Bv, const Float3& C, const float* g, bool unit, float* dA, float* dB, float* dC) {
    const float e0[3] = {Bv.x - A.x, Bv.y - A.y, Bv.z - A.z};
    const float e1[3] = {C.x - A.x, C.y - A.y, C.z - A.z};
    float n[3], dn[3];
    cross3(e0, e1, n);
    if (unit) {
        const float len = sqrtf((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
        const float den = len + 1e-10f;
        const float ng = (n[0] * g[0] + n[1] * g[1]) + n[2] * g[2];
        const float iden = 1.f / den, c = (len > 0.f) ? (ng * iden * iden) / len : 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) dn[j] = g[j] * iden - c * n[j];
    } else {
#pragma unroll
        for (int j = 0; j < 3; ++j) dn[j] = g[j];
    }
    float de0[3], de1[3];
    cross3(e1, dn, de0);
    cross3(dn, e0, de1);
#pragma unroll
    for (int j = 0; j < 3; ++j) { dB[j] += de0[j]; dC[j] += de1[j]; dA[j] -= de0[j] + de1[j]; }
}

// Eight lanes per vertex gather through the static vertex->corner CSR (no atomics); each workgroup leaves its partial of
// dL/dT in the workspace and prepare_final sums the partials of an image in index order: deterministic.
__global__ __launch_bounds__(256) void prepare_bwd_kernel(MMPrepareDesc d, MMPrepareGrads g) {
    __shared__ float s_red[4][12];
    const int b = blockIdx.y, tid = threadIdx.x;
    float T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = d.transform[b * 12 + i];
    const float* vb = d.vertices + (size_t)b * d.V * 3;
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.f;
    const int v = blockIdx.x * 32 + (tid >> 3), cl = tid & 7;
    float pj[3];
    prepare_proj(d, pj);
    if (v < d.V) {
        const float p[3] = {vb[v * 3], vb[v * 3 + 1], vb[v * 3 + 2]};
        const Float3 me = to_camera(p, T);
        const float pz = me.z * pj[2];
        const float xi = (me.x * pj[0]) / pz, yi = (me.y * pj[1]) / pz;
        float dv[3] = {0.f, 0.f, 0.f};
        const int beg = d.vc_offsets[v], end = d.vc_offsets[v + 1];
        for (int it = beg + cl; it < end; it += 8) {
            const int item = d.vc_items[it];
            const int f = item / 3, k = item - f * 3;
            const size_t o = (size_t)b * d.F + f;
            if (g.grad_face_vertices_camera) {
                const float* q = g.grad_face_vertices_camera + (o * 3 + k) * 3;
                dv[0] += q[0]; dv[1] += q[1]; dv[2] += q[2];
            }
            if (g.grad_face_vertices_image) {
                const float gx = g.grad_face_vertices_image[(o * 3 + k) * 2], gy = g.grad_face_vertices_image[(o * 3 + k) * 2 + 1];
                dv[0] += gx * pj[0] / pz;
                dv[1] += gy * pj[1] / pz;
                dv[2] += -(gx * xi + gy * yi) * pj[2] / pz;
            }
            if (g.grad_face_normals) {
                const float gn[3] = {g.grad_face_normals[o * 3], g.grad_face_normals[o * 3 + 1], g.grad_face_normals[o * 3 + 2]};
                if (gn[0] != 0.f || gn[1] != 0.f || gn[2] != 0.f) {
                    const Float3 A = to_camera(vb + (size_t)clamp_vertex(d.faces[f * 3], d.V) * 3, T),
                                 Bv = to_camera(vb + (size_t)clamp_vertex(d.faces[f * 3 + 1], d.V) * 3, T),
                                 C = to_camera(vb + (size_t)clamp_vertex(d.faces[f * 3 + 2], d.V) * 3, T);
                    float dA[3] = {0.f, 0.f, 0.f}, dB[3] = {0.f, 0.f, 0.f}, dC[3] = {0.f, 0.f, 0.f};
                    normal_backward(A, Bv, C, gn, true, dA, dB, dC);
                    const float* mine = k == 0 ? dA : (k == 1 ? dB : dC);
                    dv[0] += mine[0]; dv[1] += mine[1]; dv[2] += mine[2];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) { dv[j] += xchg_f32<4>(dv[j], tid); dv[j] += xchg_f32<2>(dv[j], tid); dv[j] += xchg_f32<1>(dv[j], tid); }
        if (cl == 0) {
            float* gv = g.grad_vertices + ((size_t)b * d.V + v) * 3;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                gv[i] = (T[i * 3 + 0] * dv[0] + T[i * 3 + 1] * dv[1]) + T[i * 3 + 2] * dv[2];
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[i * 3 + j] = p[i] * dv[j];
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[9 + j] = dv[j];
        }
    }
    if (!g.grad_transform) return;
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = wave_sum(acc[i]);
    if ((tid & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) s_red[tid >> 6][i] = acc[i];
    }
    __syncthreads();
    if (tid < 12) ((float*)d.workspace)[((size_t)b * gridDim.x + blockIdx.x) * 12 + tid] = ((s_red[0][tid] + s_red[1][tid]) + s_red[2][tid]) + s_red[3][tid];
}

// The vertex -> corner CSR of a face list, on the device (mm_build_vertex_corner_csr_device): ONE workgroup -- histogram of the corners'
// vertices in LDS, block scan, scatter through LDS cursors, then every vertex's (short) list is put in ascending order by its own thread,
// so the lists -- and with them the order of every gather that walks them -- are those of the host builder, run to run.
#define MM_CSR_MAX_V 12288
__global__ __launch_bounds__(1024) void csr_build_kernel(int V, int F, const int32_t* __restrict__ faces, int32_t* offsets, int32_t* items, int32_t* status) {
    __shared__ int s_cnt[MM_CSR_MAX_V];
    __shared__ int s_wave[16];
    __shared__ int s_bad;
    const int tid = threadIdx.x, n = 3 * F;
    for (int v = tid; v < V; v += 1024) s_cnt[v] = 0;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    int bad = 0;
    for (int i = tid; i < n; i += 1024) {
        const int v = faces[i];
        if (v >= 0 && v < V) atomicAdd(&s_cnt[v], 1); else ++bad;
    }
    if (bad) atomicAdd(&s_bad, bad);
    __syncthreads();
    // exclusive scan: thread t owns vertices [t * per, (t + 1) * per)
    const int per = (V + 1023) / 1024, v0 = min(V, tid * per), v1 = min(V, v0 + per);
    int mine = 0;
    for (int v = v0; v < v1; ++v) mine += s_cnt[v];
    int wtot;
    int run = wave_prefix_excl(mine, tid & 63, wtot);
    if ((tid & 63) == 63) s_wave[tid >> 6] = wtot;
    __syncthreads();
    for (int w = 0; w < (tid >> 6); ++w) run += s_wave[w];
    for (int v = v0; v < v1; ++v) { const int c = s_cnt[v]; offsets[v] = run; s_cnt[v] = run; run += c; }   // s_cnt becomes the cursor
    if (tid == 1023) offsets[V] = run;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        const int v = faces[i];
        if (v >= 0 && v < V) __hip_atomic_store(items + atomicAdd(&s_cnt[v], 1), i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __threadfence();
    __syncthreads();
    // ascending order inside every list.  The lists are contiguous: vertex v's is [cursor[v-1], cursor[v]) now that every cursor stands at its
    // list's end.  Up to twelve entries (every template of the reference: valence <= 10) are fetched in ONE trip, sorted in registers by a fixed
    // network and stored back; a longer list is insertion-sorted in place (agent-scope accesses: the scatter above went around this CU's L1).
    for (int v = tid; v < V; v += 1024) {
        const int beg = v ? s_cnt[v - 1] : 0, end = s_cnt[v], len = end - beg;
        if (len <= 12) {
            int e[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) e[k] = k < len ? __hip_atomic_load(items + beg + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7FFFFFFF;
#pragma unroll
            for (int pass = 0; pass < 12; ++pass)
#pragma unroll
                for (int k = pass & 1; k + 1 < 12; k += 2) { const int lo = min(e[k], e[k + 1]), hi = max(e[k], e[k + 1]); e[k] = lo; e[k + 1] = hi; }   // odd-even transposition
#pragma unroll
            for (int k = 0; k < 12; ++k) if (k < len) items[beg + k] = e[k];
            continue;
        }
        for (int i = beg + 1; i < end; ++i) {
            const int key = __hip_atomic_load(items + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int j = i - 1;
            for (; j >= beg; --j) {
                const int q = __hip_atomic_load(items + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (q <= key) break;
                __hip_atomic_store(items + j + 1, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __hip_atomic_store(items + j + 1, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid == 0 && status && s_bad) __hip_atomic_fetch_add(status, s_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(64) void prepare_final_kernel(const float* partial, int groups, float* grad_transform) {
    const int b = blockIdx.x, lane = threadIdx.x;
    for (int i = 0; i < 12; ++i) {
        float s = 0.f;
        for (int k = lane; k < groups; k += 64) s += partial[((size_t)b * groups + k) * 12 + i];
        s = wave_sum(s);
        if (lane == 0) grad_transform[b * 12 + i] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// face_normals
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void face_normals_fwd_kernel(long long n, int unit, const float* fv, float* out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* q = fv + i * 9;
    const float e0[3] = {q[3] - q[0], q[4] - q[1], q[5] - q[2]};
    const float e1[3] = {q[6] - q[0], q[7] - q[1], q[8] - q[2]};
    float nn[3];
    cross3(e0, e1, nn);
    if (unit) {
        const float den = sqrtf((nn[0] * nn[0] + nn[1] * nn[1]) + nn[2] * nn[2]) + 1e-10f;
        nn[0] = nn[0] / den; nn[1] = nn[1] / den; nn[2] = nn[2] / den;
    }
    out[i * 3] = nn[0]; out[i * 3 + 1] = nn[1]; out[i * 3 + 2] = nn[2];
}

__global__ __launch_bounds__(256) void face_normals_bwd_kernel(long long n, int unit, const float* fv, const float* gout, float* gfv) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* q = fv + i * 9;
    const Float3 A = {q[0], q[1], q[2]}, Bv = {q[3], q[4], q[5]}, C = {q[6], q[7], q[8]};
    const float g[3] = {gout[i * 3], gout[i * 3 + 1], gout[i * 3 + 2]};
    float dA[3] = {0.f, 0.f, 0.f}, dB[3] = {0.f, 0.f, 0.f}, dC[3] = {0.f, 0.f, 0.f};
    normal_backward(A, Bv, C, g, unit != 0, dA, dB, dC);
    float* o = gfv + i * 9;
#pragma unroll
    for (int j = 0; j < 3; ++j) { o[j] = dA[j]; o[3 + j] = dB[j]; o[6 + j] = dC[j]; }
}

// ---------------------------------------------------------------------------------------------------------------------
// texture_mapping
// ---------------------------------------------------------------------------------------------------------------------
// grid_sample 'nearest' with border padding: unnormalise, clip to [0, size-1], round half to even
__device__ inline void nearest_texel(float u, float v, int Ht, int Wt, int& x, int& y) {
    const float gx = u * 2.f - 1.f, gy = -(v * 2.f - 1.f);
    float ix = ((gx + 1.f) * (float)Wt - 1.f) / 2.f, iy = ((gy + 1.f) * (float)Ht - 1.f) / 2.f;
    ix = fminf(fmaxf(ix, 0.f), (float)(Wt - 1)); iy = fminf(fmaxf(iy, 0.f), (float)(Ht - 1));
    x = (int)nearbyintf(ix); y = (int)nearbyintf(iy);
}

__global__ __launch_bounds__(256) void texmap_fwd_kernel(MMTexMapDesc d) {
    const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    if (n >= d.N) return;
    const size_t p = (size_t)b * d.N + n;
    const float u = d.uv[p * 2], v = d.uv[p * 2 + 1];
    const size_t plane = (size_t)d.Ht * d.Wt;
    if (d.mode == MM_TEXMAP_NEAREST) {
        int x, y;
        nearest_texel(u, v, d.Ht, d.Wt, x, y);
        for (int c = 0; c < d.C; ++c) d.out[p * d.C + c] = d.textures[((size_t)b * d.C + c) * plane + (size_t)y * d.Wt + x];
        return;
    }
    const Bilin s = bilin_setup(u, v, d.Ht, d.Wt);
    const bool inw = s.x0 < d.Wt && s.y0 < d.Ht, ine = s.x1 < d.Wt && s.y0 < d.Ht;
    const bool isw = s.x0 < d.Wt && s.y1 < d.Ht, ise = s.x1 < d.Wt && s.y1 < d.Ht;
    for (int c = 0; c < d.C; ++c) {
        const float* tex = d.textures + ((size_t)b * d.C + c) * plane;
        float tc = 0.f;
        if (inw) tc += tex[(size_t)s.y0 * d.Wt + s.x0] * s.wnw;
        if (ine) tc += tex[(size_t)s.y0 * d.Wt + s.x1] * s.wne;
        if (isw) tc += tex[(size_t)s.y1 * d.Wt + s.x0] * s.wsw;
        if (ise) tc += tex[(size_t)s.y1 * d.Wt + s.x1] * s.wse;
        d.out[p * d.C + c] = tc;
    }
}

// The texture gradient is a scatter.  The fused path gathers it through per-tile record lists; this stand-alone operator keeps
// kaolin/ATen's formulation (hardware float atomics into a zero-filled gradient) but never issues an atomic for a zero
// contribution: pixels no face covers all sample uv = (0,0) and would otherwise pile onto one texel.
// Scratch of the deterministic texture-gradient scatter: per image max |grad_out| (float bits), then one 64-bit fixed-point accumulator per texel.
__host__ __device__ inline size_t texmap_acc_offset(int B) { return align256((size_t)B * sizeof(unsigned)); }
__global__ __launch_bounds__(256) void texmap_max_kernel(MMTexMapDesc d, const float* grad_out, unsigned* gmax) {
    const int b = blockIdx.y;
    const size_t n = (size_t)d.N * d.C;
    // The maximum is taken over the BIT PATTERNS of |grad_out| (an unsigned max): finite < inf < NaN in that order, so a single non-finite
    // upstream value survives the reduction (fmaxf would drop a NaN) and the image's texture gradient is poisoned below -- as ATen's
    // grid_sampler backward and the float-atomic path propagate it -- instead of silently coming out as zeros.
    unsigned m = 0u;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = max(m, __float_as_uint(grad_out[(size_t)b * n + i]) & 0x7FFFFFFFu);
    __shared__ unsigned s_m[4];
    m = (unsigned)wave_max_i32((int)m);                            // (31-bit values: the signed maximum is the unsigned one)
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    // ONE atomic per workgroup (a few per image: thousands of waves on 48 words queued at the memory side took 60 us)
    if (threadIdx.x == 0) {
        m = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
        if (m != 0u) atomicMax(gmax + b, m);
    }
}
__device__ inline bool texmap_nonfinite(unsigned maxbits) { return maxbits >= 0x7F800000u; }   // some upstream value of the image is inf or NaN
// every contribution is |grad_out| * weight with weight <= 1: the image's largest is placed at 2^40, 2^22 of them fit a 63-bit sum
__device__ inline float texmap_scale(unsigned maxbits, float& inv) {
    const float M = __uint_as_float(maxbits);
    if (!(M > 0.f) || !(M < INFINITY)) { inv = 0.f; return 0.f; }
    int e;
    (void)frexpf(M, &e);
    const int k = min(max(40 - e, -80), 126);
    inv = ldexpf(1.f, -k);
    return ldexpf(1.f, k);
}
__global__ __launch_bounds__(256) void texmap_finish_kernel(MMTexMapDesc d, const unsigned* gmax, const long long* acc, float* grad_textures) {
    const int b = blockIdx.y;
    const size_t n = (size_t)d.C * d.Ht * d.Wt;
    float inv;
    (void)texmap_scale(gmax[b], inv);
    // a non-finite upstream gradient: no fixed-point scale exists -- every texel of the image says so (NaN), never a silent zero
    if (texmap_nonfinite(gmax[b])) inv = __builtin_nanf("");
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) grad_textures[(size_t)b * n + i] = (float)acc[(size_t)b * n + i] * inv;
}

template <bool kFixed>
__global__ __launch_bounds__(256) void texmap_bwd_kernel(MMTexMapDesc d, MMTexMapGrads g) {
    const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    if (n >= d.N) return;
    float scale = 1.f, inv_unused;
    long long* acc64 = nullptr;
    if (kFixed) {
        scale = texmap_scale(((const unsigned*)g.workspace)[b], inv_unused);
        acc64 = (long long*)((char*)g.workspace + texmap_acc_offset(d.B));
    }
    // one contribution to texel `idx` of the (B,C,Ht,Wt) gradient: a 64-bit integer add (exact, commutative) or a float atomic
    auto scatter = [&](size_t idx, float v) {
        if (kFixed) { if (scale != 0.f) atomicAdd((unsigned long long*)(acc64 + idx), (unsigned long long)__float2ll_rn(v * scale)); }   // (scale 0: all-zero or non-finite
        else atomicAdd(g.grad_textures + idx, v);                                                                                       //  upstream -- nothing to convert; finish writes 0 / NaN)
    };
    const size_t p = (size_t)b * d.N + n;
    const float u = d.uv[p * 2], v = d.uv[p * 2 + 1];
    const size_t plane = (size_t)d.Ht * d.Wt;
    if (d.mode == MM_TEXMAP_NEAREST) {
        if (g.grad_uv) { g.grad_uv[p * 2] = 0.f; g.grad_uv[p * 2 + 1] = 0.f; }
        if (g.grad_textures) {
            int x, y;
            nearest_texel(u, v, d.Ht, d.Wt, x, y);
            for (int c = 0; c < d.C; ++c) {
                const float go = g.grad_out[p * d.C + c];
                if (go != 0.f) scatter(((size_t)b * d.C + c) * plane + (size_t)y * d.Wt + x, go);
            }
        }
        return;
    }
    const Bilin s = bilin_setup(u, v, d.Ht, d.Wt);
    const bool inw = s.x0 < d.Wt && s.y0 < d.Ht, ine = s.x1 < d.Wt && s.y0 < d.Ht;
    const bool isw = s.x0 < d.Wt && s.y1 < d.Ht, ise = s.x1 < d.Wt && s.y1 < d.Ht;
    const float ex = 1.f - s.tx, ey = 1.f - s.ty;
    float gix = 0.f, giy = 0.f;
    for (int c = 0; c < d.C; ++c) {
        const float* tex = d.textures + ((size_t)b * d.C + c) * plane;
        const float go = g.grad_out[p * d.C + c];
        const float tnw = inw ? tex[(size_t)s.y0 * d.Wt + s.x0] : 0.f, tne = ine ? tex[(size_t)s.y0 * d.Wt + s.x1] : 0.f;
        const float tsw = isw ? tex[(size_t)s.y1 * d.Wt + s.x0] : 0.f, tse = ise ? tex[(size_t)s.y1 * d.Wt + s.x1] : 0.f;
        if (g.grad_textures && go != 0.f) {
            const size_t dt = ((size_t)b * d.C + c) * plane;
            if (inw) scatter(dt + (size_t)s.y0 * d.Wt + s.x0, go * s.wnw);
            if (ine) scatter(dt + (size_t)s.y0 * d.Wt + s.x1, go * s.wne);
            if (isw) scatter(dt + (size_t)s.y1 * d.Wt + s.x0, go * s.wsw);
            if (ise) scatter(dt + (size_t)s.y1 * d.Wt + s.x1, go * s.wse);
        }
        gix += go * ((tne - tnw) * ey + (tse - tsw) * s.ty);
        giy += go * ((tsw - tnw) * ex + (tse - tne) * s.tx);
    }
    if (g.grad_uv) {
        g.grad_uv[p * 2 + 0] = gix * s.mx * ((float)d.Wt / 2.f) * 2.f;
        g.grad_uv[p * 2 + 1] = giy * s.my * ((float)d.Ht / 2.f) * -2.f;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// spherical_harmonic_lighting
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sh_fwd_kernel(MMShDesc d) {
    const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    if (n >= d.N) return;
    const size_t p = (size_t)b * d.N + n;
    float bnd[9];
    sh_bands(d.normals[p * 3], d.normals[p * 3 + 1], d.normals[p * 3 + 2], bnd);
    const float* L = d.lights + b * 9;
    float coef = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) coef += bnd[i] * L[i];
    d.out[p] = coef;
}

// Backward in two launches (advisor r05: one 1024-thread workgroup per image did both halves -- B CUs busy, N / 1024 dependent trips per thread):
//   sh_bwd_normals_kernel   dL/dnormals is elementwise: a thread per point over the whole grid.
//   sh_bwd_lights_kernel    the nine light gradients are sums over ALL of an image's points -- per thread over its points in index order, a fixed
//                           butterfly per wave, the sixteen waves in index order: no atomics (2 304 float atomics per image on nine addresses took
//                           87 us at B=48, 128x128), no zero-fill, no scratch (the ABI gives this operator none), bitwise reproducible.  Still one
//                           workgroup per image, but with EIGHT points of every thread in flight per trip: N / 8192 dependent trips instead of
//                           N / 1024 (512x512: 32 instead of 256).
__global__ __launch_bounds__(256) void sh_bwd_normals_kernel(MMShDesc d, MMShGrads g) {
    const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    if (n >= d.N) return;
    const float* L = d.lights + b * 9;
    const size_t p = (size_t)b * d.N + n;
    const float x = d.normals[p * 3], y = d.normals[p * 3 + 1], z = d.normals[p * 3 + 2];
    const float go = g.grad_out[p];
    g.grad_normals[p * 3 + 0] = go * (((MM_SH_C1 * L[1] + MM_SH_C4 * y * L[4]) + MM_SH_C7 * z * L[7]) + 2.f * MM_SH_C8 * x * L[8]);
    g.grad_normals[p * 3 + 1] = go * (((MM_SH_C1 * L[3] + MM_SH_C4 * x * L[4]) + MM_SH_C4 * z * L[5]) - 2.f * MM_SH_C8 * y * L[8]);
    g.grad_normals[p * 3 + 2] = go * (((MM_SH_C1 * L[2] + MM_SH_C4 * y * L[5]) + 2.f * MM_SH_C6 * z * L[6]) + MM_SH_C7 * x * L[7]);
}

#define MM_SH_UNROLL 8
__global__ __launch_bounds__(1024) void sh_bwd_lights_kernel(MMShDesc d, MMShGrads g) {
    __shared__ float s_red[16][9];
    const int b = blockIdx.x, tid = threadIdx.x;
    float acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = 0.f;
    for (int n0 = tid; n0 < d.N; n0 += 1024 * MM_SH_UNROLL) {     // the trip's loads first (clamped addresses), then the sums in index order
        float xs[MM_SH_UNROLL], ys[MM_SH_UNROLL], zs[MM_SH_UNROLL], gs[MM_SH_UNROLL];
#pragma unroll
        for (int u = 0; u < MM_SH_UNROLL; ++u) {
            const int n = n0 + 1024 * u;
            const size_t p = (size_t)b * d.N + min(n, d.N - 1);
            xs[u] = d.normals[p * 3]; ys[u] = d.normals[p * 3 + 1]; zs[u] = d.normals[p * 3 + 2];
            gs[u] = n < d.N ? g.grad_out[p] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < MM_SH_UNROLL; ++u) {
            if (n0 + 1024 * u < d.N) {
                float bnd[9];
                sh_bands(xs[u], ys[u], zs[u], bnd);
#pragma unroll
                for (int i = 0; i < 9; ++i) acc[i] += gs[u] * bnd[i];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const float s = wave_sum(acc[i]);
        if ((tid & 63) == 0) s_red[tid >> 6][i] = s;
    }
    __syncthreads();
    if (tid < 9) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) s += s_red[w][tid];
        g.grad_lights[b * 9 + tid] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// mask_iou: one 1024-thread workgroup per image reduces in a fixed order (deterministic); a single wave folds the images
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void mask_iou_sums_kernel(MMMaskIouDesc d) {
    __shared__ float s_red[16][2];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* l = d.lhs + (size_t)b * d.N; const float* r = d.rhs + (size_t)b * d.N;
    float up = 0.f, down = 0.f;
    for (int i = tid; i < d.N; i += 1024) { const float mul = l[i] * r[i]; up += mul; down += (l[i] + r[i]) - mul; }
    up = wave_sum(up); down = wave_sum(down);
    if ((tid & 63) == 0) { s_red[tid >> 6][0] = up; s_red[tid >> 6][1] = down; }
    __syncthreads();
    if (tid < 2) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) s += s_red[w][tid];
        d.sums[b * 2 + tid] = s;
    }
}

__global__ __launch_bounds__(64) void mask_iou_final_kernel(MMMaskIouDesc d) {
    float iou = 0.f;
    for (int b = threadIdx.x; b < d.B; b += 64) iou += d.sums[b * 2] / (d.sums[b * 2 + 1] + 1e-10f);
    iou = wave_sum(iou);
    if (threadIdx.x == 0) d.loss[0] = 1.f - iou / (float)d.B;
}

__global__ __launch_bounds__(256) void mask_iou_bwd_kernel(MMMaskIouDesc d, const float* grad_loss, float* gl, float* gr) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= d.N) return;
    const float up = d.sums[b * 2], U = d.sums[b * 2 + 1] + 1e-10f;
    const float gs = grad_loss ? grad_loss[0] : 1.f;
    const float ka = -gs / ((float)d.B * U), kb = gs * up / ((float)d.B * U * U);
    const size_t p = (size_t)b * d.N + i;
    const float l = d.lhs[p], r = d.rhs[p];
    // d(up)/dl = r, d(down)/dl = 1 - r  (and symmetrically for r)
    if (gl) gl[p] = ka * r + kb * (1.f - r);
    if (gr) gr[p] = ka * l + kb * (1.f - l);
}

}  // namespace mm

extern "C" {

size_t mm_prepare_vertices_query_workspace(const MMPrepareDesc* d) {
    if (!d || d->B <= 0 || d->V <= 0) return 0;
    return mm::align256((size_t)d->B * ((d->V + 31) / 32) * 12 * sizeof(float));
}

static int check_prepare(const MMPrepareDesc* d) {
    if (!d) return MM_ERR_NULL_POINTER;
    if (d->B <= 0 || d->V <= 0 || d->F <= 0) return MM_ERR_BAD_SHAPE;
    if (!d->faces || !d->vertices || !d->transform) return MM_ERR_NULL_POINTER;
    return MM_OK;
}

int mm_prepare_vertices_forward(const MMPrepareDesc* d, mm_stream_t stream) {
    const int st = check_prepare(d);
    if (st != MM_OK) return st;
    if (!d->face_vertices_camera || !d->face_vertices_image || !d->face_normals) return MM_ERR_NULL_POINTER;
    mm::clear_stale_error();
    hipLaunchKernelGGL(mm::prepare_fwd_kernel, dim3((d->F + 255) / 256, d->B), dim3(256), 0, (hipStream_t)stream, *d);
    return mm::launch_ok("prepare_vertices_fwd");
}

int mm_prepare_vertices_backward(const MMPrepareDesc* d, const MMPrepareGrads* g, mm_stream_t stream) {
    const int st = check_prepare(d);
    if (st != MM_OK) return st;
    if (!g || !g->grad_vertices || !d->vc_offsets || !d->vc_items) return MM_ERR_NULL_POINTER;
    if (g->grad_transform && (!d->workspace || d->workspace_bytes < mm_prepare_vertices_query_workspace(d))) return MM_ERR_WORKSPACE;
    mm::clear_stale_error();
    const int groups = (d->V + 31) / 32;
    hipLaunchKernelGGL(mm::prepare_bwd_kernel, dim3(groups, d->B), dim3(256), 0, (hipStream_t)stream, *d, *g);
    if (mm::launch_ok("prepare_vertices_bwd") != MM_OK) return MM_ERR_LAUNCH;
    if (g->grad_transform)
        hipLaunchKernelGGL(mm::prepare_final_kernel, dim3(d->B), dim3(64), 0, (hipStream_t)stream, (const float*)d->workspace, groups, g->grad_transform);
    return mm::launch_ok("prepare_vertices_final");
}

int mm_build_vertex_corner_csr_device(int32_t V, int32_t F, const int32_t* faces_dev, int32_t* offsets_dev, int32_t* items_dev,
                                      int32_t* status_flag, mm_stream_t stream) {
    if (!faces_dev || !offsets_dev || !items_dev) return MM_ERR_NULL_POINTER;
    if (V <= 0 || F <= 0) return MM_ERR_BAD_SHAPE;
    if (V > MM_CSR_MAX_V) return MM_ERR_UNSUPPORTED;
    mm::clear_stale_error();
    hipLaunchKernelGGL(mm::csr_build_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, V, F, faces_dev, offsets_dev, items_dev, status_flag);
    return mm::launch_ok("vertex_corner_csr");
}

int mm_face_normals_forward(int64_t n, int32_t unit, const float* fv, float* normals, mm_stream_t stream) {
    if (!fv || !normals) return MM_ERR_NULL_POINTER;
    if (n <= 0 || n > (int64_t)0x7FFFFFFF * 256) return MM_ERR_BAD_SHAPE;
    mm::clear_stale_error();
    hipLaunchKernelGGL(mm::face_normals_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long long)n, unit, fv, normals);
    return mm::launch_ok("face_normals_fwd");
}

int mm_face_normals_backward(int64_t n, int32_t unit, const float* fv, const float* gout, float* gfv, mm_stream_t stream) {
    if (!fv || !gout || !gfv) return MM_ERR_NULL_POINTER;
    if (n <= 0 || n > (int64_t)0x7FFFFFFF * 256) return MM_ERR_BAD_SHAPE;
    mm::clear_stale_error();
    hipLaunchKernelGGL(mm::face_normals_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long long)n, unit, fv, gout, gfv);
    return mm::launch_ok("face_normals_bwd");
}

static int check_texmap(const MMTexMapDesc* d) {
    if (!d) return MM_ERR_NULL_POINTER;
    if (d->B <= 0 || d->N <= 0 || d->C <= 0 || d->Ht <= 0 || d->Wt <= 0 || d->B > 65535) return MM_ERR_BAD_SHAPE;
    if (d->mode != MM_TEXMAP_NEAREST && d->mode != MM_TEXMAP_BILINEAR) return MM_ERR_UNSUPPORTED;
    if (!d->uv || !d->textures) return MM_ERR_NULL_POINTER;
    return MM_OK;
}

int mm_texture_mapping_forward(const MMTexMapDesc* d, mm_stream_t stream) {
    const int st = check_texmap(d);
    if (st != MM_OK) return st;
    if (!d->out) return MM_ERR_NULL_POINTER;
    mm::clear_stale_error();
    hipLaunchKernelGGL(mm::texmap_fwd_kernel, dim3((d->N + 255) / 256, d->B), dim3(256), 0, (hipStream_t)stream, *d);
    return mm::launch_ok("texture_mapping_fwd");
}

int mm_texture_mapping_backward(const MMTexMapDesc* d, const MMTexMapGrads* g, mm_stream_t stream) {
    const int st = check_texmap(d);
    if (st != MM_OK) return st;
    if (!g || !g->grad_out || (!g->grad_uv && !g->grad_textures)) return MM_ERR_NULL_POINTER;
    mm::clear_stale_error();
    hipStream_t s = (hipStream_t)stream;
    const size_t texels = (size_t)d->B * d->C * d->Ht * d->Wt;
    if (g->grad_textures && g->workspace) {                      // deterministic: max |grad_out| -> 64-bit fixed-point scatter -> convert
        if (g->workspace_bytes < mm_texture_mapping_backward_query_workspace(d) || ((uintptr_t)g->workspace & 255)) return MM_ERR_WORKSPACE;
        if (hipMemsetAsync(g->workspace, 0, mm_texture_mapping_backward_query_workspace(d), s) != hipSuccess) return MM_ERR_LAUNCH;
        const unsigned nb = (unsigned)std::min<size_t>(((size_t)d->N * d->C + 255) / 256, 8);
        hipLaunchKernelGGL(mm::texmap_max_kernel, dim3(nb, d->B), dim3(256), 0, s, *d, g->grad_out, (unsigned*)g->workspace);
        hipLaunchKernelGGL(mm::texmap_bwd_kernel<true>, dim3((d->N + 255) / 256, d->B), dim3(256), 0, s, *d, *g);
        const unsigned nf = (unsigned)std::min<size_t>(((size_t)d->C * d->Ht * d->Wt + 255) / 256, 256);
        hipLaunchKernelGGL(mm::texmap_finish_kernel, dim3(nf, d->B), dim3(256), 0, s, *d, (const unsigned*)g->workspace,
                           (const long long*)((const char*)g->workspace + mm::texmap_acc_offset(d->B)), g->grad_textures);
        return mm::launch_ok("texture_mapping_bwd");
    }
    if (g->grad_textures && hipMemsetAsync(g->grad_textures, 0, sizeof(float) * texels, s) != hipSuccess) return MM_ERR_LAUNCH;
    hipLaunchKernelGGL(mm::texmap_bwd_kernel<false>, dim3((d->N + 255) / 256, d->B), dim3(256), 0, s, *d, *g);
    return mm::launch_ok("texture_mapping_bwd");
}

size_t mm_texture_mapping_backward_query_workspace(const MMTexMapDesc* d) {
    if (!d || d->B <= 0 || d->C <= 0 || d->Ht <= 0 || d->Wt <= 0) return 0;
    return mm::texmap_acc_offset(d->B) + mm::align256((size_t)d->B * d->C * d->Ht * d->Wt * sizeof(long long));
}

static int check_sh(const MMShDesc* d) {
    if (!d) return MM_ERR_NULL_POINTER;
    if (d->B <= 0 || d->N <= 0 || d->B > 65535) return MM_ERR_BAD_SHAPE;
    if (!d->normals || !d->lights) return MM_ERR_NULL_POINTER;
    return MM_OK;
}

int mm_sh_lighting_forward(const MMShDesc* d, mm_stream_t stream) {
    const int st = check_sh(d);
    if (st != MM_OK) return st;
    if (!d->out) return MM_ERR_NULL_POINTER;
    mm::clear_stale_error();
    hipLaunchKernelGGL(mm::sh_fwd_kernel, dim3((d->N + 255) / 256, d->B), dim3(256), 0, (hipStream_t)stream, *d);
    return mm::launch_ok("sh_lighting_fwd");
}

int mm_sh_lighting_backward(const MMShDesc* d, const MMShGrads* g, mm_stream_t stream) {
    const int st = check_sh(d);
    if (st != MM_OK) return st;
    if (!g || !g->grad_out || (!g->grad_normals && !g->grad_lights)) return MM_ERR_NULL_POINTER;
    mm::clear_stale_error();
    if (g->grad_normals) hipLaunchKernelGGL(mm::sh_bwd_normals_kernel, dim3((d->N + 255) / 256, d->B), dim3(256), 0, (hipStream_t)stream, *d, *g);
    if (g->grad_lights) hipLaunchKernelGGL(mm::sh_bwd_lights_kernel, dim3(d->B), dim3(1024), 0, (hipStream_t)stream, *d, *g);
    return mm::launch_ok("sh_lighting_bwd");
}

static int check_iou(const MMMaskIouDesc* d) {
    if (!d) return MM_ERR_NULL_POINTER;
    if (d->B <= 0 || d->N <= 0 || d->B > 65535) return MM_ERR_BAD_SHAPE;
    if (!d->lhs || !d->rhs || !d->sums) return MM_ERR_NULL_POINTER;
    return MM_OK;
}

int mm_mask_iou_forward(const MMMaskIouDesc* d, mm_stream_t stream) {
    const int st = check_iou(d);
    if (st != MM_OK) return st;
    if (!d->loss) return MM_ERR_NULL_POINTER;
    mm::clear_stale_error();
    hipLaunchKernelGGL(mm::mask_iou_sums_kernel, dim3(d->B), dim3(1024), 0, (hipStream_t)stream, *d);
    if (mm::launch_ok("mask_iou_sums") != MM_OK) return MM_ERR_LAUNCH;
    hipLaunchKernelGGL(mm::mask_iou_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, *d);
    return mm::launch_ok("mask_iou_final");
}

int mm_mask_iou_backward(const MMMaskIouDesc* d, const float* grad_loss, float* gl, float* gr, mm_stream_t stream) {
    const int st = check_iou(d);
    if (st != MM_OK) return st;
    if (!gl && !gr) return MM_ERR_NULL_POINTER;
    mm::clear_stale_error();
    hipLaunchKernelGGL(mm::mask_iou_bwd_kernel, dim3((d->N + 255) / 256, d->B), dim3(256), 0, (hipStream_t)stream, *d, grad_loss, gl, gr);
    return mm::launch_ok("mask_iou_bwd");
}

}  // extern "C"
