// mm_reg.hip -- mesh regularisers of the training step for gfx950.
//
// Replaces, in one launch per direction, what the reference computes with ~60 small torch launches per attribute set
// (/root/reference/networks.py:392-491, orchestrated by trainer.py:54-74):
//   calc_reg_loss   laplacian term mean((L dv)^2) * V * 3  (L = the dense (V,V) uniform laplacian, :249, here as CSR)
//                   + flat term mean((n_e1 . n_e2 - 1)^2) * E over the edge -> two faces table (:422-431)
//   calc_reg_edge   0.1 * mean_b || len_e - mean_e(len_e) ||_2                                  (:453-461)
//   calc_reg_depth / depthR / depthC                                                            (:463-485)
//   calc_reg_deform mean |dv|                                                                   (:487-491)
//   recon_flip(L1=False)  mean( |dv_v - S dv_flip(v)| * mask_f )                                (:392-410)
// One 256-thread workgroup per image (the data of an image is a few tens of KB and stays in L2); fixed-order reductions, so
// results are reproducible run to run.  The backward gathers through static transposed tables (no atomics).
#include <cstdio>

#include "mm_device.h"

namespace mm {

struct RegArgs {
    int B, V, F, E;
    unsigned terms;
    const int32_t *lap_offsets, *lap_cols; const float* lap_vals;
    const int32_t *lapT_offsets, *lapT_cols; const float* lapT_vals;
    const int32_t *edges, *edge2faces, *ve_offsets, *ve_items, *fe_offsets, *fe_items, *flip_index, *flipT_offsets, *flipT_items;
    const float* sign_init;
    const float *vertices, *delta, *fn;
    float ratio, temp, eps;
    float* losses;
    // workspace
    float* y;          // (B,V,3) L dv
    float* elen;       // (B,E)   edge lengths
    float* estat;      // (B,2)   mean length, || len - mean ||
    float* partial;    // (B,8)
    unsigned* ticket;  // (1)
    // backward
    const float* weights;
    float *g_vertices, *g_delta, *g_fn;
};

// fixed-order sum over the workgroup (4 waves): butterfly inside a wave, then waves 0..3 in order; every thread gets the total
__device__ inline float block_sum(float v, float* s_red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
}

__device__ inline float depth_weight(const RegArgs& a, int term, float x, float y) {
    const float yr = y / a.ratio;
    const float r2 = x * x + yr * yr;
    return term == MM_REG_DEPTHR ? expf(a.temp * r2) : r2;
}

__global__ __launch_bounds__(256) void mesh_reg_fwd_kernel(RegArgs a) {
    __shared__ float s_red[4];
    __shared__ int s_last;
    const int b = blockIdx.x, tid = threadIdx.x;
    // four workgroups per image, each with its share of the eight terms (every term is a chain of dependent gathers + a block reduction:
    // one workgroup doing all of them one after the other was 33 us of pure latency for 1.5 MB of operands)
    const unsigned group = blockIdx.y == 0 ? (1u << MM_REG_LAPLACIAN)
                         : blockIdx.y == 1 ? (1u << MM_REG_FLAT) | (1u << MM_REG_DEFORM)
                         : blockIdx.y == 2 ? (1u << MM_REG_EDGE) | (1u << MM_REG_FLIP)
                                           : (1u << MM_REG_DEPTH) | (1u << MM_REG_DEPTHR) | (1u << MM_REG_DEPTHC);
    const unsigned terms = a.terms & group;
    const float* dv = a.delta ? a.delta + (size_t)b * a.V * 3 : nullptr;
    const float* vv = a.vertices ? a.vertices + (size_t)b * a.V * 3 : nullptr;
    const float* fn = a.fn ? a.fn + (size_t)b * a.F * 3 : nullptr;
    float part[MM_REG_TERMS];
#pragma unroll
    for (int k = 0; k < MM_REG_TERMS; ++k) part[k] = 0.f;

    if (terms & (1u << MM_REG_LAPLACIAN)) {
        float s = 0.f;
        for (int v = tid; v < a.V; v += 256) {
            float y0 = 0.f, y1 = 0.f, y2 = 0.f;
            for (int it = a.lap_offsets[v]; it < a.lap_offsets[v + 1]; ++it) {
                const int c = a.lap_cols[it];
                const float w = a.lap_vals[it];
                y0 += w * dv[c * 3]; y1 += w * dv[c * 3 + 1]; y2 += w * dv[c * 3 + 2];
            }
            float* y = a.y + ((size_t)b * a.V + v) * 3;
            y[0] = y0; y[1] = y1; y[2] = y2;
            s += (y0 * y0 + y1 * y1) + y2 * y2;
        }
        part[MM_REG_LAPLACIAN] = block_sum(s, s_red);
    }
    if (terms & (1u << MM_REG_FLAT)) {
        float s = 0.f;
        for (int e = tid; e < a.E; e += 256) {
            const float* n1 = fn + (size_t)a.edge2faces[e * 2] * 3;
            const float* n2 = fn + (size_t)a.edge2faces[e * 2 + 1] * 3;
            const float c = ((n1[0] * n2[0] + n1[1] * n2[1]) + n1[2] * n2[2]) - 1.f;
            s += c * c;
        }
        part[MM_REG_FLAT] = block_sum(s, s_red);
    }
    if (terms & (1u << MM_REG_EDGE)) {
        float s = 0.f;
        for (int e = tid; e < a.E; e += 256) {
            const float* p = vv + (size_t)a.edges[e * 2] * 3;
            const float* q = vv + (size_t)a.edges[e * 2 + 1] * 3;
            const float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
            const float len = sqrtf((dx * dx + dy * dy) + dz * dz);
            a.elen[(size_t)b * a.E + e] = len;
            s += len;
        }
        const float mean = block_sum(s, s_red) / (float)a.E;
        float s2 = 0.f;
        for (int e = tid; e < a.E; e += 256) {                     // every thread re-reads the lengths it wrote itself
            const float d = a.elen[(size_t)b * a.E + e] - mean;
            s2 += d * d;
        }
        const float nrm = sqrtf(block_sum(s2, s_red));
        if (tid == 0) { a.estat[b * 2] = mean; a.estat[b * 2 + 1] = nrm; }
        part[MM_REG_EDGE] = nrm;
    }
    if (a.terms & ((1u << MM_REG_DEPTH) | (1u << MM_REG_DEPTHR) | (1u << MM_REG_DEPTHC))) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int v = tid; v < a.V; v += 256) {
            const float x = vv[v * 3], y = vv[v * 3 + 1], z = vv[v * 3 + 2];
            s0 += z * z;
            const float zs = z - (a.sign_init[v] >= 0.f ? a.eps : -a.eps);          // (:472, :483) keeps the side of the plane
            if (terms & (1u << MM_REG_DEPTHR)) s1 += (zs * zs) * depth_weight(a, MM_REG_DEPTHR, x, y);
            if (terms & (1u << MM_REG_DEPTHC)) s2 += (zs * zs) * depth_weight(a, MM_REG_DEPTHC, x, y);
        }
        if (terms & (1u << MM_REG_DEPTH)) part[MM_REG_DEPTH] = block_sum(s0, s_red);
        if (terms & (1u << MM_REG_DEPTHR)) part[MM_REG_DEPTHR] = block_sum(s1, s_red);
        if (terms & (1u << MM_REG_DEPTHC)) part[MM_REG_DEPTHC] = block_sum(s2, s_red);
    }
    if (terms & (1u << MM_REG_DEFORM)) {
        float s = 0.f;
        for (int v = tid; v < a.V; v += 256) s += sqrtf((dv[v * 3] * dv[v * 3] + dv[v * 3 + 1] * dv[v * 3 + 1]) + dv[v * 3 + 2] * dv[v * 3 + 2]);
        part[MM_REG_DEFORM] = block_sum(s, s_red);
    }
    if (terms & (1u << MM_REG_FLIP)) {
        float s = 0.f;
        for (int v = tid; v < a.V; v += 256) {
            const int u = a.flip_index[v];
            const float rx = dv[v * 3] - dv[u * 3], ry = dv[v * 3 + 1] - dv[u * 3 + 1], rz = dv[v * 3 + 2] + dv[u * 3 + 2];
            const float zu = dv[u * 3 + 2];
            const float sg = (zu > 0.f ? 1.f : (zu < 0.f ? -1.f : 0.f)) * a.sign_init[u];     // mask_f = mask_a[flip(v)]
            if (sg > 0.f) s += sg * sqrtf((rx * rx + ry * ry) + rz * rz);
        }
        part[MM_REG_FLIP] = block_sum(s, s_red);
    }

    // per-image partial sums -> the last workgroup to arrive adds them up in image order.  Agent-scope atomic stores / loads
    // and a returning ticket: no L2-wide fence (see vertex_bwd).
    if (tid < MM_REG_TERMS && (group >> tid & 1u)) {
        float mine = 0.f;
#pragma unroll
        for (int k = 0; k < MM_REG_TERMS; ++k) if (k == tid) mine = part[k];
        // RETURNING exchange: its value only comes back once the write has been performed at the memory side
        const float old = __hip_atomic_exchange(a.partial + b * MM_REG_TERMS + tid, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" :: "v"(old));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned prev = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = prev == gridDim.x * gridDim.y - 1;
        if (s_last) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next call
    }
    __syncthreads();
    if (!s_last) return;
    // the partials of all images, fetched by all threads at once (image = thread index, 256 per pass), parked in LDS and added up by one
    // thread per term in image order: bitwise reproducible, one trip to memory instead of B
    __shared__ float s_part[256][MM_REG_TERMS + 1];
    float tot = 0.f;
    for (int i0 = 0; i0 < a.B; i0 += 256) {
        if (i0 + tid < a.B) {
#pragma unroll
            for (int k = 0; k < MM_REG_TERMS; ++k)
                s_part[tid][k] = __hip_atomic_load(a.partial + (size_t)(i0 + tid) * MM_REG_TERMS + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (tid < MM_REG_TERMS) for (int i = 0; i < 256 && i0 + i < a.B; ++i) tot += s_part[i][tid];
        __syncthreads();
    }
    if (tid >= MM_REG_TERMS) return;
    const float fb = (float)a.B, fbv = (float)a.B * (float)a.V;
    float out = 0.f;
    if (tid == MM_REG_LAPLACIAN || tid == MM_REG_FLAT) out = tot / fb;       // mean(.)*V*3 and mean(.)*E
    else if (tid == MM_REG_EDGE) out = 0.1f * (tot / fb);
    else out = tot / fbv;
    a.losses[tid] = (a.terms & (1u << tid)) ? out : 0.f;
}

// backward: d(sum_k weights[k] * losses[k]) / d(vertices, delta_vertices, face_normals); every output element is written.
// Output elements are independent (gathers only): grid (image, 256-element chunk).
__global__ __launch_bounds__(256) void mesh_reg_bwd_kernel(RegArgs a) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* dv = a.delta ? a.delta + (size_t)b * a.V * 3 : nullptr;
    const float* vv = a.vertices ? a.vertices + (size_t)b * a.V * 3 : nullptr;
    const float* fn = a.fn ? a.fn + (size_t)b * a.F * 3 : nullptr;
    float w[MM_REG_TERMS];
#pragma unroll
    for (int k = 0; k < MM_REG_TERMS; ++k) w[k] = (a.terms & (1u << k)) ? a.weights[k] : 0.f;
    const float fb = (float)a.B, fbv = (float)a.B * (float)a.V;

    if (a.g_delta) {
        for (int v = blockIdx.y * 256 + tid; v < a.V; v += gridDim.y * 256) {
            float g0 = 0.f, g1 = 0.f, g2 = 0.f;
            if (a.terms & (1u << MM_REG_LAPLACIAN)) {             // (2/B) L^T (L dv)
                float t0 = 0.f, t1 = 0.f, t2 = 0.f;
                for (int it = a.lapT_offsets[v]; it < a.lapT_offsets[v + 1]; ++it) {
                    const float* y = a.y + ((size_t)b * a.V + a.lapT_cols[it]) * 3;
                    const float lw = a.lapT_vals[it];
                    t0 += lw * y[0]; t1 += lw * y[1]; t2 += lw * y[2];
                }
                const float c = w[MM_REG_LAPLACIAN] * 2.f / fb;
                g0 += c * t0; g1 += c * t1; g2 += c * t2;
            }
            if (a.terms & (1u << MM_REG_DEFORM)) {
                const float x = dv[v * 3], y = dv[v * 3 + 1], z = dv[v * 3 + 2];
                const float n = sqrtf((x * x + y * y) + z * z);
                if (n > 0.f) { const float c = w[MM_REG_DEFORM] / (fbv * n); g0 += c * x; g1 += c * y; g2 += c * z; }
            }
            if (a.terms & (1u << MM_REG_FLIP)) {
                const float c = w[MM_REG_FLIP] / fbv;
                {   // as Na: pair (v, flip(v))
                    const int u = a.flip_index[v];
                    const float rx = dv[v * 3] - dv[u * 3], ry = dv[v * 3 + 1] - dv[u * 3 + 1], rz = dv[v * 3 + 2] + dv[u * 3 + 2];
                    const float zu = dv[u * 3 + 2];
                    const float sg = (zu > 0.f ? 1.f : (zu < 0.f ? -1.f : 0.f)) * a.sign_init[u];
                    const float n = sqrtf((rx * rx + ry * ry) + rz * rz);
                    if (sg > 0.f && n > 0.f) { const float k = c * sg / n; g0 += k * rx; g1 += k * ry; g2 += k * rz; }
                }
                // as the mirrored partner Nf = S dv_v of every t with flip(t) = v; that pair's mask is mask_a[v]
                const float zv = dv[v * 3 + 2];
                const float sgv = (zv > 0.f ? 1.f : (zv < 0.f ? -1.f : 0.f)) * a.sign_init[v];
                if (sgv > 0.f) {
                    for (int it = a.flipT_offsets[v]; it < a.flipT_offsets[v + 1]; ++it) {
                        const int t = a.flipT_items[it];
                        const float rx = dv[t * 3] - dv[v * 3], ry = dv[t * 3 + 1] - dv[v * 3 + 1], rz = dv[t * 3 + 2] + dv[v * 3 + 2];
                        const float n = sqrtf((rx * rx + ry * ry) + rz * rz);
                        if (n > 0.f) { const float k = c * sgv / n; g0 -= k * rx; g1 -= k * ry; g2 += k * rz; }
                    }
                }
            }
            float* g = a.g_delta + ((size_t)b * a.V + v) * 3;
            g[0] = g0; g[1] = g1; g[2] = g2;
        }
    }
    if (a.g_vertices) {
        const float mean = (a.terms & (1u << MM_REG_EDGE)) ? a.estat[b * 2] : 0.f;
        const float nrm = (a.terms & (1u << MM_REG_EDGE)) ? a.estat[b * 2 + 1] : 0.f;
        for (int v = blockIdx.y * 256 + tid; v < a.V; v += gridDim.y * 256) {
            float g0 = 0.f, g1 = 0.f, g2 = 0.f;
            const float x = vv[v * 3], y = vv[v * 3 + 1], z = vv[v * 3 + 2];
            if ((a.terms & (1u << MM_REG_EDGE)) && nrm > 0.f) {
                const float c = w[MM_REG_EDGE] * 0.1f / (fb * nrm);
                for (int it = a.ve_offsets[v]; it < a.ve_offsets[v + 1]; ++it) {
                    const int item = a.ve_items[it], e = item >> 1;
                    const int o = a.edges[e * 2 + ((item & 1) ^ 1)];                 // the other end
                    const float len = a.elen[(size_t)b * a.E + e];
                    if (len > 0.f) {
                        const float k = c * (len - mean) / len;
                        g0 += k * (x - vv[o * 3]); g1 += k * (y - vv[o * 3 + 1]); g2 += k * (z - vv[o * 3 + 2]);
                    }
                }
            }
            if (a.terms & (1u << MM_REG_DEPTH)) g2 += w[MM_REG_DEPTH] * 2.f * z / fbv;
            const float zs = z - (a.sign_init[v] >= 0.f ? a.eps : -a.eps);
            if (a.terms & (1u << MM_REG_DEPTHR)) g2 += w[MM_REG_DEPTHR] * 2.f * zs * depth_weight(a, MM_REG_DEPTHR, x, y) / fbv;   // x, y detached (:467)
            if (a.terms & (1u << MM_REG_DEPTHC)) g2 += w[MM_REG_DEPTHC] * 2.f * zs * depth_weight(a, MM_REG_DEPTHC, x, y) / fbv;
            float* g = a.g_vertices + ((size_t)b * a.V + v) * 3;
            g[0] = g0; g[1] = g1; g[2] = g2;
        }
    }
    if (a.g_fn) {
        for (int f = blockIdx.y * 256 + tid; f < a.F; f += gridDim.y * 256) {
            float g0 = 0.f, g1 = 0.f, g2 = 0.f;
            if (a.terms & (1u << MM_REG_FLAT)) {
                const float c = w[MM_REG_FLAT] * 2.f / fb;
                const float* me = fn + (size_t)f * 3;
                for (int it = a.fe_offsets[f]; it < a.fe_offsets[f + 1]; ++it) {
                    const int item = a.fe_items[it], e = item >> 1;
                    const float* ot = fn + (size_t)a.edge2faces[e * 2 + ((item & 1) ^ 1)] * 3;
                    const float cs = ((me[0] * ot[0] + me[1] * ot[1]) + me[2] * ot[2]) - 1.f;
                    g0 += c * cs * ot[0]; g1 += c * cs * ot[1]; g2 += c * cs * ot[2];
                }
            }
            float* g = a.g_fn + ((size_t)b * a.F + f) * 3;
            g[0] = g0; g[1] = g1; g[2] = g2;
        }
    }
}

static size_t reg_carve(const MMMeshRegDesc* d, RegArgs* a) {
    size_t o = 0;
    char* p = (char*)d->workspace;
    auto take = [&](size_t bytes) { char* r = p ? p + o : nullptr; o += align256(bytes); return r; };
    float* y = (float*)take((size_t)d->B * d->V * 3 * sizeof(float));
    float* elen = (float*)take((size_t)d->B * d->E * sizeof(float));
    float* estat = (float*)take((size_t)d->B * 2 * sizeof(float));
    float* partial = (float*)take((size_t)d->B * MM_REG_TERMS * sizeof(float));
    unsigned* ticket = (unsigned*)take(sizeof(unsigned));
    if (a) { a->y = y; a->elen = elen; a->estat = estat; a->partial = partial; a->ticket = ticket; }
    return o;
}

size_t reg_workspace_bytes(const MMMeshRegDesc* d) { return reg_carve(d, nullptr); }

static RegArgs reg_args(const MMMeshRegDesc* d) {
    RegArgs a;
    a.B = d->B; a.V = d->V; a.F = d->F; a.E = d->E; a.terms = d->terms;
    a.lap_offsets = d->lap_offsets; a.lap_cols = d->lap_cols; a.lap_vals = d->lap_vals;
    a.lapT_offsets = d->lapT_offsets; a.lapT_cols = d->lapT_cols; a.lapT_vals = d->lapT_vals;
    a.edges = d->edges; a.edge2faces = d->edge2faces; a.ve_offsets = d->ve_offsets; a.ve_items = d->ve_items;
    a.fe_offsets = d->fe_offsets; a.fe_items = d->fe_items; a.flip_index = d->flip_index;
    a.flipT_offsets = d->flipT_offsets; a.flipT_items = d->flipT_items; a.sign_init = d->sign_init;
    a.vertices = d->vertices; a.delta = d->delta_vertices; a.fn = d->face_normals;
    a.ratio = d->ratio; a.temp = d->temp; a.eps = d->eps; a.losses = d->losses;
    a.weights = nullptr; a.g_vertices = a.g_delta = a.g_fn = nullptr;
    reg_carve(d, &a);
    return a;
}

int launch_reg_fwd(const MMMeshRegDesc* d, hipStream_t s) {
    RegArgs a = reg_args(d);
    hipLaunchKernelGGL(mesh_reg_fwd_kernel, dim3(d->B, 4), dim3(256), 0, s, a);
    return launch_ok("mesh_reg_fwd");
}

int launch_reg_bwd(const MMMeshRegDesc* d, const MMMeshRegGrads* g, hipStream_t s) {
    RegArgs a = reg_args(d);
    a.weights = g->weights; a.g_vertices = g->grad_vertices; a.g_delta = g->grad_delta_vertices; a.g_fn = g->grad_face_normals;
    const int chunks = ((d->V > d->F ? d->V : d->F) + 255) / 256;
    hipLaunchKernelGGL(mesh_reg_bwd_kernel, dim3(d->B, chunks), dim3(256), 0, s, a);
    return launch_ok("mesh_reg_bwd");
}

}  // namespace mm
