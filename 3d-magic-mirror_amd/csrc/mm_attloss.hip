// mm_attloss.hip -- attribute-reconstruction losses of the training step for gfx950 (second half).
//
// Replaces DiffRender.recon_att (/root/reference/networks.py:326-362, chamfer=False terms): seven mean |a-b| (L1) or mean (a-b)^2
// reductions -- azimuths and elevations through angle2xy (cos, sin of the angle in degrees), distances, biases, vertices,
// textures, lights -- that the reference computes with ~25 torch launches and three passes over the (B,3,Ht,Wt) textures.
// Here: one streaming launch for the seven sums (fixed-order partials, last workgroup adds them up: reproducible run to run)
// and one for every gradient.
#include "mm_device.h"

namespace mm {

struct AttArgs {
    int B, V, T, l1;                 // T = 3*Ht*Wt texels per image
    const float *p_az, *p_el, *p_di, *p_bi, *p_ve, *p_te, *p_li;
    const float *t_az, *t_el, *t_di, *t_bi, *t_ve, *t_te, *t_li;
    float* losses;                   // (7) means: azim, elev, dist, bias, shape, texture, light
    float* partial;                  // (blocks, 8)
    unsigned* ticket;
    const float* weights;            // (7) dL/d losses[k]
    float *g_az, *g_el, *g_di, *g_bi, *g_ve, *g_te, *g_li;      // d/d pred (any may be null)
    float *h_az, *h_el, *h_di, *h_bi, *h_ve, *h_te, *h_li;      // d/d target (any may be null)
};

__device__ inline float att_term(float d, int l1) { return l1 ? fabsf(d) : d * d; }
__device__ inline float att_dterm(float d, int l1) { return l1 ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 2.f * d; }

__device__ inline float block_sum4(float v, float* s_red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
}

#define MM_DEG2RAD_F 0.017453292519943295f

__global__ __launch_bounds__(256) void att_fwd_kernel(AttArgs a) {
    __shared__ float s_red[4];
    __shared__ int s_last;
    const int tid = threadIdx.x;
    const size_t gid = (size_t)blockIdx.x * 256 + tid, gstride = (size_t)gridDim.x * 256;
    float part[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // the big ones, grid-strided (float4 where the pointers allow it is not worth the alignment cases: the launch is latency bound)
    float s = 0.f;
    {
        const size_t n = (size_t)a.B * a.T;
        if ((n & 3) == 0 && ((((size_t)a.p_te) | ((size_t)a.t_te)) & 15) == 0) {          // 16-byte loads (torch tensors always qualify)
            const float4* p4 = (const float4*)a.p_te; const float4* t4 = (const float4*)a.t_te;
            const size_t n4 = n >> 2;
            size_t i = gid;
            for (; i + gstride < n4; i += 2 * gstride) {         // four 16-byte loads in flight per trip
                const float4 p0 = p4[i], p1 = p4[i + gstride], t0 = t4[i], t1 = t4[i + gstride];
                s += ((att_term(p0.x - t0.x, a.l1) + att_term(p0.y - t0.y, a.l1)) + att_term(p0.z - t0.z, a.l1)) + att_term(p0.w - t0.w, a.l1);
                s += ((att_term(p1.x - t1.x, a.l1) + att_term(p1.y - t1.y, a.l1)) + att_term(p1.z - t1.z, a.l1)) + att_term(p1.w - t1.w, a.l1);
            }
            for (; i < n4; i += gstride) {
                const float4 p0 = p4[i], t0 = t4[i];
                s += ((att_term(p0.x - t0.x, a.l1) + att_term(p0.y - t0.y, a.l1)) + att_term(p0.z - t0.z, a.l1)) + att_term(p0.w - t0.w, a.l1);
            }
        } else {
            size_t i = gid;
            for (; i + 3 * gstride < n; i += 4 * gstride) {      // eight loads in flight per trip
                const float p0 = a.p_te[i], p1 = a.p_te[i + gstride], p2 = a.p_te[i + 2 * gstride], p3 = a.p_te[i + 3 * gstride];
                const float t0 = a.t_te[i], t1 = a.t_te[i + gstride], t2 = a.t_te[i + 2 * gstride], t3 = a.t_te[i + 3 * gstride];
                s += att_term(p0 - t0, a.l1); s += att_term(p1 - t1, a.l1); s += att_term(p2 - t2, a.l1); s += att_term(p3 - t3, a.l1);
            }
            for (; i < n; i += gstride) s += att_term(a.p_te[i] - a.t_te[i], a.l1);
        }
    }
    part[5] = block_sum4(s, s_red);
    s = 0.f;
    for (size_t i = gid; i < (size_t)a.B * a.V * 3; i += gstride) s += att_term(a.p_ve[i] - a.t_ve[i], a.l1);
    part[4] = block_sum4(s, s_red);
    if (blockIdx.x == 0) {                                        // the small ones
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s6 = 0.f;
        for (int i = tid; i < a.B; i += 256) {
            const float pa = a.p_az[i] * MM_DEG2RAD_F, ta = a.t_az[i] * MM_DEG2RAD_F;
            s0 += att_term(cosf(pa) - cosf(ta), a.l1) + att_term(sinf(pa) - sinf(ta), a.l1);
            const float pe = a.p_el[i] * MM_DEG2RAD_F, te = a.t_el[i] * MM_DEG2RAD_F;
            s1 += att_term(cosf(pe) - cosf(te), a.l1) + att_term(sinf(pe) - sinf(te), a.l1);
            s2 += att_term(a.p_di[i] - a.t_di[i], a.l1);
        }
        for (int i = tid; i < a.B * 2; i += 256) s3 += att_term(a.p_bi[i] - a.t_bi[i], a.l1);
        for (int i = tid; i < a.B * 9; i += 256) s6 += att_term(a.p_li[i] - a.t_li[i], a.l1);
        part[0] = block_sum4(s0, s_red); part[1] = block_sum4(s1, s_red); part[2] = block_sum4(s2, s_red);
        part[3] = block_sum4(s3, s_red); part[6] = block_sum4(s6, s_red);
    }
    if (tid < 7) {
        float mine = 0.f;
#pragma unroll
        for (int k = 0; k < 7; ++k) if (k == tid) mine = part[k];
        const float old = __hip_atomic_exchange(a.partial + (size_t)blockIdx.x * 8 + tid, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" :: "v"(old));                            // returning: performed before the ticket below (see vertex_bwd)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned prev = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = prev == gridDim.x - 1;
        if (s_last) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    // the last workgroup adds the partials up: rows strided over its 256 threads (seven agent-scope loads per row, all rows of a thread in
    // flight together), then a fixed-order block reduction -- reproducible run to run.  (Seven threads walking the up to 1024 rows one
    // dependent load after the other took 140 of this kernel's 155 us.)
    float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (unsigned i = tid; i < gridDim.x; i += 256) {
#pragma unroll
        for (int k = 0; k < 7; ++k) acc[k] += __hip_atomic_load(a.partial + (size_t)i * 8 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < 7; ++k) { const float t = block_sum4(acc[k], s_red); if (k == tid) tot = t; }
    if (tid >= 7) return;
    const float n[7] = {(float)a.B * 2.f, (float)a.B * 2.f, (float)a.B, (float)a.B * 2.f, (float)a.B * (float)a.V * 3.f,
                        (float)a.B * (float)a.T, (float)a.B * 9.f};
    a.losses[tid] = tot / n[tid];
}

__device__ inline void att_grad_store(float* g, float* h, size_t i, float v) {
    if (g) g[i] = v;
    if (h) h[i] = -v;
}

__global__ __launch_bounds__(256) void att_bwd_kernel(AttArgs a) {
    const int tid = threadIdx.x;
    const size_t gid = (size_t)blockIdx.x * 256 + tid, gstride = (size_t)gridDim.x * 256;
    float w[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) w[k] = a.weights[k];
    if (a.g_te || a.h_te) {
        const float c = w[5] / ((float)a.B * (float)a.T);
        for (size_t i = gid; i < (size_t)a.B * a.T; i += gstride) att_grad_store(a.g_te, a.h_te, i, c * att_dterm(a.p_te[i] - a.t_te[i], a.l1));
    }
    if (a.g_ve || a.h_ve) {
        const float c = w[4] / ((float)a.B * (float)a.V * 3.f);
        for (size_t i = gid; i < (size_t)a.B * a.V * 3; i += gstride) att_grad_store(a.g_ve, a.h_ve, i, c * att_dterm(a.p_ve[i] - a.t_ve[i], a.l1));
    }
    if (blockIdx.x != 0) return;
    for (int i = tid; i < a.B; i += 256) {
        // angle2xy: d/d angle[deg] of term(cos - cos_t) + term(sin - sin_t)
        const float ca = w[0] / ((float)a.B * 2.f), ce = w[1] / ((float)a.B * 2.f);
        const float pa = a.p_az[i] * MM_DEG2RAD_F, ta = a.t_az[i] * MM_DEG2RAD_F;
        const float dc = att_dterm(cosf(pa) - cosf(ta), a.l1), ds = att_dterm(sinf(pa) - sinf(ta), a.l1);
        if (a.g_az) a.g_az[i] = ca * (dc * (-sinf(pa)) + ds * cosf(pa)) * MM_DEG2RAD_F;
        if (a.h_az) a.h_az[i] = ca * (dc * sinf(ta) - ds * cosf(ta)) * MM_DEG2RAD_F;
        const float pe = a.p_el[i] * MM_DEG2RAD_F, te = a.t_el[i] * MM_DEG2RAD_F;
        const float ec = att_dterm(cosf(pe) - cosf(te), a.l1), es = att_dterm(sinf(pe) - sinf(te), a.l1);
        if (a.g_el) a.g_el[i] = ce * (ec * (-sinf(pe)) + es * cosf(pe)) * MM_DEG2RAD_F;
        if (a.h_el) a.h_el[i] = ce * (ec * sinf(te) - es * cosf(te)) * MM_DEG2RAD_F;
        att_grad_store(a.g_di, a.h_di, i, (w[2] / (float)a.B) * att_dterm(a.p_di[i] - a.t_di[i], a.l1));
    }
    for (int i = tid; i < a.B * 2; i += 256) att_grad_store(a.g_bi, a.h_bi, i, (w[3] / ((float)a.B * 2.f)) * att_dterm(a.p_bi[i] - a.t_bi[i], a.l1));
    for (int i = tid; i < a.B * 9; i += 256) att_grad_store(a.g_li, a.h_li, i, (w[6] / ((float)a.B * 9.f)) * att_dterm(a.p_li[i] - a.t_li[i], a.l1));
}

static int att_blocks(const MMAttLossDesc* d) {
    const size_t n = (size_t)d->B * 3 * d->Ht * d->Wt;
    size_t b = (n + 256 * 16 - 1) / (256 * 16);                 // ~16 texels per thread
    if (b < 1) b = 1;
    if (b > 1024) b = 1024;
    return (int)b;
}

size_t att_workspace_bytes(const MMAttLossDesc* d) { return align256((size_t)att_blocks(d) * 8 * sizeof(float)) + 256; }

static AttArgs att_args(const MMAttLossDesc* d) {
    AttArgs a;
    a.B = d->B; a.V = d->V; a.T = 3 * d->Ht * d->Wt; a.l1 = d->l1;
    a.p_az = d->pred.azimuths; a.p_el = d->pred.elevations; a.p_di = d->pred.distances; a.p_bi = d->pred.biases;
    a.p_ve = d->pred.vertices; a.p_te = d->pred.textures; a.p_li = d->pred.lights;
    a.t_az = d->target.azimuths; a.t_el = d->target.elevations; a.t_di = d->target.distances; a.t_bi = d->target.biases;
    a.t_ve = d->target.vertices; a.t_te = d->target.textures; a.t_li = d->target.lights;
    a.losses = d->losses;
    a.partial = (float*)d->workspace;
    a.ticket = (unsigned*)((char*)d->workspace + align256((size_t)att_blocks(d) * 8 * sizeof(float)));
    a.weights = nullptr;
    a.g_az = a.g_el = a.g_di = a.g_bi = a.g_ve = a.g_te = a.g_li = nullptr;
    a.h_az = a.h_el = a.h_di = a.h_bi = a.h_ve = a.h_te = a.h_li = nullptr;
    return a;
}

int launch_att_fwd(const MMAttLossDesc* d, hipStream_t s) {
    const AttArgs a = att_args(d);
    hipLaunchKernelGGL(att_fwd_kernel, dim3(att_blocks(d)), dim3(256), 0, s, a);
    return launch_ok("att_fwd");
}

int launch_att_bwd(const MMAttLossDesc* d, const MMAttLossGrads* g, hipStream_t s) {
    AttArgs a = att_args(d);
    a.weights = g->weights;
    a.g_az = g->pred.azimuths; a.g_el = g->pred.elevations; a.g_di = g->pred.distances; a.g_bi = g->pred.biases;
    a.g_ve = g->pred.vertices; a.g_te = g->pred.textures; a.g_li = g->pred.lights;
    a.h_az = g->target.azimuths; a.h_el = g->target.elevations; a.h_di = g->target.distances; a.h_bi = g->target.biases;
    a.h_ve = g->target.vertices; a.h_te = g->target.textures; a.h_li = g->target.lights;
    hipLaunchKernelGGL(att_bwd_kernel, dim3(att_blocks(d)), dim3(256), 0, s, a);
    return launch_ok("att_bwd");
}

}  // namespace mm
