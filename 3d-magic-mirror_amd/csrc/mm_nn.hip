// mm_nn.hip -- brute-force nearest neighbour between two point clouds per batch item, for the chamfer loss the
// reference takes from pytorch3d (pytorch3d.loss.chamfer_distance -> knn_points(K=1); call sites
// /root/reference/networks.py:342,356 and trainer.py:445,469,483; pytorch3d 0.7.0 has no ROCm build).
//
// Shape of the work: B x (N + M) queries x the other cloud = 40 M distance evaluations at B=48, 642 x 642 -- vector-issue work, 1 MB of
// data.  The kernel is laid out for the issue rate, not for memory:
//   * BOTH directions (x -> y and y -> x) in one launch: the chamfer loss always needs both, and one launch fills the chip twice as well.
//   * a workgroup owns 128 queries, two per lane, held as the two halves of packed-fp32 registers: v_pk_add / v_pk_mul do both queries'
//     differences, squares and sums in one instruction each -- the same IEEE operations, in the same order, as the scalar form
//     (dx*dx + dy*dy) + dz*dz without contraction.
//   * the other cloud is staged once per workgroup in LDS as float4 {x, y, z, -}; every wave reads a point with ONE ds_read_b128 whose
//     address is the same in all lanes (a broadcast), amortised over the wave's 128 evaluations.  The eight waves scan an eighth of the
//     cloud each; their results are merged through LDS in index order.
//   * the running minimum is kept per GROUP of four points (three v_min + one compare per four evaluations instead of a compare and two
//     selects per evaluation); the winning group's four distances are recomputed once at the end to name the point.
// Ties keep the lowest index (strict < in ascending order everywhere), NaN distances never win -- as before.
#include <algorithm>
#include "mm_device.h"

namespace mm {

typedef float nn_f2 __attribute__((ext_vector_type(2)));

#define MM_NN_Q 128          // queries per workgroup
#define MM_NN_MAXPTS 2048    // points of the other cloud held in LDS per pass (32 KiB); larger clouds take several passes

__device__ inline nn_f2 nn_dist2(nn_f2 px, nn_f2 py, nn_f2 pz, const float4 q) {
#pragma clang fp contract(off)
    const nn_f2 dx = px - q.x, dy = py - q.y, dz = pz - q.z;
    return (dx * dx + dy * dy) + dz * dz;
}

#define MM_NN_WAVES 8        // waves per workgroup: each scans an eighth of the other cloud (more waves per SIMD: the loop is LDS-latency-bound with few)
__global__ __launch_bounds__(64 * MM_NN_WAVES) void nn_pair_kernel(int N, int M, const float* __restrict__ x, const float* __restrict__ y,
                                                      float* __restrict__ dist_x, int32_t* __restrict__ idx_x,
                                                      float* __restrict__ dist_y, int32_t* __restrict__ idx_y, int nxq) {
    extern __shared__ float4 s_pts[];                             // min(Mo, MM_NN_MAXPTS) points, padded to a multiple of 16
    __shared__ float s_best[MM_NN_WAVES][MM_NN_Q];
    __shared__ int s_idx[MM_NN_WAVES][MM_NN_Q];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // direction: the first nxq workgroups of a row search x's points in y, the others y's points in x
    const bool fwd = (int)blockIdx.x < nxq;
    const int Nq = fwd ? N : M, Mo = fwd ? M : N;                 // queries / points of the other cloud
    const float* qs = (fwd ? x : y) + (size_t)b * Nq * 3;
    const float* os = (fwd ? y : x) + (size_t)b * Mo * 3;
    float* dist = fwd ? dist_x : dist_y;
    int32_t* idx = fwd ? idx_x : idx_y;
    const int q0 = ((int)blockIdx.x - (fwd ? 0 : nxq)) * MM_NN_Q;
    // this lane's two queries (lane and lane + 64 of the workgroup's 128)
    nn_f2 px = {0.f, 0.f}, py = px, pz = px;
    {
        const int qa = q0 + lane, qb = q0 + 64 + lane;
        if (qa < Nq) { px.x = qs[(size_t)qa * 3]; py.x = qs[(size_t)qa * 3 + 1]; pz.x = qs[(size_t)qa * 3 + 2]; }
        if (qb < Nq) { px.y = qs[(size_t)qb * 3]; py.y = qs[(size_t)qb * 3 + 1]; pz.y = qs[(size_t)qb * 3 + 2]; }
    }
    nn_f2 best = {INFINITY, INFINITY};
    int bga = 0, bgb = 0;                                         // winning group (absolute index of its first point / 4) of either query
    for (int p0 = 0; p0 < Mo; p0 += MM_NN_MAXPTS) {               // (one pass for every cloud the reference has)
        const int np = min(MM_NN_MAXPTS, Mo - p0), np16 = (np + 15) & ~15;
        __syncthreads();
        for (int k = tid; k < np16; k += 64 * MM_NN_WAVES) {
            float4 v = make_float4(INFINITY, INFINITY, INFINITY, 0.f);   // padding: infinitely far, never nearer than a real point
            if (k < np) { const float* p = os + (size_t)(p0 + k) * 3; v = make_float4(p[0], p[1], p[2], 0.f); }
            s_pts[k] = v;
        }
        __syncthreads();
        // wave wv scans groups [g0, g1) of this pass: its share of them, in ascending order.  The next group's four points are requested from
        // LDS before this group's arithmetic (the loop is a chain  read -> 40 instructions -> read  otherwise, with two waves per SIMD to hide it)
        const int ng = np16 >> 2, per = (ng + MM_NN_WAVES - 1) / MM_NN_WAVES, g0 = wv * per, g1 = min(ng, g0 + per);
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
        if (g0 < g1) { a0 = s_pts[4 * g0]; a1 = s_pts[4 * g0 + 1]; a2 = s_pts[4 * g0 + 2]; a3 = s_pts[4 * g0 + 3]; }
        for (int g = g0; g < g1; ++g) {
            const float4 c0 = a0, c1 = a1, c2 = a2, c3 = a3;
            const int gn = min(g + 1, g1 - 1);
            a0 = s_pts[4 * gn]; a1 = s_pts[4 * gn + 1]; a2 = s_pts[4 * gn + 2]; a3 = s_pts[4 * gn + 3];
            const nn_f2 d0 = nn_dist2(px, py, pz, c0), d1 = nn_dist2(px, py, pz, c1), d2 = nn_dist2(px, py, pz, c2), d3 = nn_dist2(px, py, pz, c3);
            const float ma = fminf(fminf(d0.x, d1.x), fminf(d2.x, d3.x)), mb = fminf(fminf(d0.y, d1.y), fminf(d2.y, d3.y));
            if (ma < best.x) { best.x = ma; bga = (p0 >> 2) + g; }
            if (mb < best.y) { best.y = mb; bgb = (p0 >> 2) + g; }
        }
    }
    // the winning point of either query: the first point of the winning group at the winning distance (re-read from memory: 4 points)
    int ia = bga * 4, ib = bgb * 4;
    {
        float da[4], db[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ja = min(bga * 4 + k, Mo - 1), jb = min(bgb * 4 + k, Mo - 1);
            const float* pa = os + (size_t)ja * 3; const float* pb = os + (size_t)jb * 3;
            const nn_f2 d = nn_dist2(px, py, pz, make_float4(pa[0], pa[1], pa[2], 0.f)), e = nn_dist2(px, py, pz, make_float4(pb[0], pb[1], pb[2], 0.f));
            da[k] = d.x; db[k] = e.y;
        }
        int ka = 0, kb = 0;
#pragma unroll
        for (int k = 3; k >= 0; --k) { if (da[k] == best.x) ka = k; if (db[k] == best.y) kb = k; }
        ia += ka; ib += kb;
    }
    s_best[wv][lane] = best.x; s_best[wv][64 + lane] = best.y;
    s_idx[wv][lane] = ia; s_idx[wv][64 + lane] = ib;
    __syncthreads();
    if (tid < MM_NN_Q && q0 + tid < Nq) {                         // merge the waves' shares in index order (strict <: the lowest index on ties)
        float bb = s_best[0][tid]; int bi = s_idx[0][tid];
#pragma unroll
        for (int w = 1; w < MM_NN_WAVES; ++w) { const float c = s_best[w][tid]; if (c < bb) { bb = c; bi = s_idx[w][tid]; } }
        if (!(bb < INFINITY)) bi = 0;                             // (nothing finite: index 0, as a plain scan from "best = inf, index 0" leaves it)
        dist[(size_t)b * Nq + q0 + tid] = bb; idx[(size_t)b * Nq + q0 + tid] = min(bi, Mo - 1);
    }
}

static int launch_nn_pair(int B, int N, int M, const float* x, const float* y, float* dx, int32_t* ix, float* dy, int32_t* iy, bool both, hipStream_t s) {
    const int nxq = (N + MM_NN_Q - 1) / MM_NN_Q, nyq = both ? (M + MM_NN_Q - 1) / MM_NN_Q : 0;
    const int pts = std::min(MM_NN_MAXPTS, std::max(N, M));
    const size_t lds = (size_t)((pts + 15) & ~15) * sizeof(float4);
    hipLaunchKernelGGL(nn_pair_kernel, dim3(nxq + nyq, B), dim3(64 * MM_NN_WAVES), lds, s, N, M, x, y, dx, ix, dy, iy, nxq);
    return launch_ok("nearest_neighbour");
}

int launch_nn(int B, int N, int M, const float* x, const float* y, float* dist, int32_t* idx, hipStream_t s) {
    return launch_nn_pair(B, N, M, x, y, dist, idx, nullptr, nullptr, false, s);
}
int launch_nn_both(int B, int N, int M, const float* x, const float* y, float* dx, int32_t* ix, float* dy, int32_t* iy, hipStream_t s) {
    return launch_nn_pair(B, N, M, x, y, dx, ix, dy, iy, true, s);
}

}  // namespace mm
