// mm_nn.hip -- brute-force nearest neighbour between two point clouds per batch item, for the chamfer loss the
// reference takes from pytorch3d (pytorch3d.loss.chamfer_distance -> knn_points(K=1); call sites
// /root/reference/networks.py:342,356 and trainer.py:445,469,483; pytorch3d 0.7.0 has no ROCm build).
//
// Shape of the work: B x (N + M) queries x the other cloud = 40 M distance evaluations at B=48, 642 x 642 -- vector-issue work, 1 MB of
// data.  The kernel is laid out for the issue rate, not for memory:
//   * BOTH directions (x -> y and y -> x) in one launch: the chamfer loss always needs both, and one launch fills the chip twice as well.
//   * a workgroup owns 128 queries, two per lane, held as the two halves of packed-fp32 registers: v_pk_add / v_pk_mul / v_pk_fma do both
//     queries' differences, squares and sums in one instruction each.
//   * the points of the other cloud are WAVE-UNIFORM operands: every lane of a wave measures its queries against the same point at the
//     same time.  They are therefore read by SCALAR loads (s_load_dwordx8 / x16 through the scalar cache, eight points = 24 dwords at a
//     time, straight from the caller's packed x y z array) into SGPRs and enter the packed instructions as scalar operands: no LDS
//     staging pass, no barrier before the scan, no LDS bandwidth in the loop (r04's first form read every point by a broadcast
//     ds_read_b128: four of them per 32 packed instructions kept the CU's LDS port 80 % busy beside the vector ALUs).  The loads are
//     issued a group AHEAD of their use (nn_sload / nn_swait: two sets of 24 scalar registers) -- left to the compiler every scalar
//     load is followed by its wait, a trip to the scalar cache per group on a wave's critical path.
//   * the eight waves of a workgroup scan an eighth of the cloud each (latency hiding needs waves, and 576 workgroups of one wave
//     would be 0.6 waves per SIMD); their results are merged through LDS in index order.
//   * the running minimum is kept per GROUP of eight points (four v_min3 / v_min + one compare per eight evaluations instead of a
//     compare and two selects per evaluation); the winning group's eight distances are recomputed once at the end to name the point.
// Distance expression: fma(dz, dz, fma(dy, dy, dx * dx)) -- the accumulation  dist += diff * diff  over the three coordinates as
// pytorch3d's knn kernel writes it (csrc/knn/knn.cu, restated from its published source: the package is not in this image), which a
// CUDA compiler contracts to exactly these two fused operations.  Ties keep the lowest index (strict < in ascending order everywhere),
// NaN distances never win.
#include <algorithm>
#include "mm_device.h"

namespace mm {

typedef float nn_f2 __attribute__((ext_vector_type(2)));

#define MM_NN_Q 128          // queries per workgroup
#ifndef MM_NN_WAVES
#define MM_NN_WAVES 8        // waves per workgroup: each scans an eighth of the other cloud
#endif
#define MM_NN_GROUP 8        // points per step of the scan (24 dwords of scalar registers)

__device__ inline nn_f2 nn_dist2(nn_f2 px, nn_f2 py, nn_f2 pz, float qx, float qy, float qz) {
    const nn_f2 dx = px - qx, dy = py - qy, dz = pz - qz;
    return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
}
typedef float nn_f16 __attribute__((ext_vector_type(16)));
typedef float nn_f8 __attribute__((ext_vector_type(8)));
// eight points = 24 dwords into scalar registers, WITHOUT waiting for them (the compiler's own scalar loads are followed by their wait at
// once; here the next group's points travel while this group's are measured).  nn_swait is the wait, and because it names the registers
// as in/out operands no use of them can be scheduled above it.
__device__ inline void nn_sload(nn_f16& a, nn_f8& b, const float* p) {
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx8 %1, %2, 0x40" : "=&s"(a), "=&s"(b) : "s"(p));
    __builtin_amdgcn_sched_barrier(0);                            // (the arithmetic of the group in hand stays BEHIND the request)
}
__device__ inline void nn_swait(nn_f16& a, nn_f8& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b)); }

__device__ inline float nn_min8(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
    return fminf(fminf(fminf(a0, a1), fminf(a2, a3)), fminf(fminf(a4, a5), fminf(a6, a7)));
}

__global__ __launch_bounds__(64 * MM_NN_WAVES) void nn_pair_kernel(int N, int M, const float* __restrict__ x, const float* __restrict__ y,
                                                      float* __restrict__ dist_x, int32_t* __restrict__ idx_x,
                                                      float* __restrict__ dist_y, int32_t* __restrict__ idx_y, int nxq) {
    __shared__ float s_best[MM_NN_WAVES][MM_NN_Q];
    __shared__ int s_grp[MM_NN_WAVES][MM_NN_Q];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wv = MM_WAVE_UNIFORM(tid >> 6);                     // (a scalar: the scan's loop counter, addresses and loads are the scalar unit's)
    // direction: the first nxq workgroups of a row search x's points in y, the others y's points in x
    const bool fwd = (int)blockIdx.x < nxq;
    const int Nq = fwd ? N : M, Mo = fwd ? M : N;                 // queries / points of the other cloud
    const float* __restrict__ qs = (fwd ? x : y) + (size_t)b * Nq * 3;
    const float* __restrict__ os = (fwd ? y : x) + (size_t)b * Mo * 3;
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
    int bga = 0, bgb = 0;                                         // winning group (index of its first point / MM_NN_GROUP) of either query
    // wave wv scans groups [g0, g1): its share of the cloud's groups, in ascending order.  Whole groups in the loop; the cloud's last, partial
    // group (if any) is taken once, by the wave whose share ends with it, from clamped addresses.
    const int ng = (Mo + MM_NN_GROUP - 1) / MM_NN_GROUP, nfull = Mo / MM_NN_GROUP;
    const int per = (ng + MM_NN_WAVES - 1) / MM_NN_WAVES, g0 = wv * per, g1 = min(ng, g0 + per);
    auto take = [&](const float (&q)[3 * MM_NN_GROUP], int g) {
        nn_f2 d[MM_NN_GROUP];
#pragma unroll
        for (int k = 0; k < MM_NN_GROUP; ++k) d[k] = nn_dist2(px, py, pz, q[3 * k], q[3 * k + 1], q[3 * k + 2]);
        const float ma = nn_min8(d[0].x, d[1].x, d[2].x, d[3].x, d[4].x, d[5].x, d[6].x, d[7].x);
        const float mb = nn_min8(d[0].y, d[1].y, d[2].y, d[3].y, d[4].y, d[5].y, d[6].y, d[7].y);
        if (ma < best.x) { best.x = ma; bga = g; }
        if (mb < best.y) { best.y = mb; bgb = g; }
    };
    static_assert(MM_NN_GROUP == 8, "nn_min8");
    const int gend = min(g1, nfull);
#if !defined(MM_NN_NO_PREFETCH)
    if (g0 < gend) {
        nn_f16 a16, b16; nn_f8 a8, b8;
        auto take_v = [&](const nn_f16& v16, const nn_f8& v8, int g) {
            float q[3 * MM_NN_GROUP];
#pragma unroll
            for (int k = 0; k < 16; ++k) q[k] = v16[k];
#pragma unroll
            for (int k = 0; k < 8; ++k) q[16 + k] = v8[k];
            take(q, g);
        };
        nn_sload(a16, a8, os + (size_t)g0 * (3 * MM_NN_GROUP));
        for (int g = g0; g < gend; g += 2) {
            nn_swait(a16, a8);
            nn_sload(b16, b8, os + (size_t)min(g + 1, gend - 1) * (3 * MM_NN_GROUP));
            take_v(a16, a8, g);
            nn_swait(b16, b8);
            nn_sload(a16, a8, os + (size_t)min(g + 2, gend - 1) * (3 * MM_NN_GROUP));
            if (g + 1 < gend) take_v(b16, b8, g + 1);
        }
        nn_swait(a16, a8);                                        // (the last request lands before its registers are anything else's)
    }
#else                                                            // plain C++ (the compiler's scalar loads, each followed by its wait): B=48 642x642 one direction
    for (int g = g0; g < gend; ++g) {                            // 8.1 us against 6.6 (a hand-pipelined plain-C++ loop compiles to the same waits)
        float q[3 * MM_NN_GROUP];
        const float* __restrict__ p = os + (size_t)g * (3 * MM_NN_GROUP);
#pragma unroll
        for (int k = 0; k < 3 * MM_NN_GROUP; ++k) q[k] = p[k];     // wave-uniform address, read-only memory: scalar loads
        take(q, g);
    }
#endif
    if (g1 == ng && nfull < ng && g0 < g1) {                      // (wave-uniform) the partial group: points beyond the cloud are infinitely far
        float q[3 * MM_NN_GROUP];
#pragma unroll
        for (int k = 0; k < MM_NN_GROUP; ++k) {
            const int j = nfull * MM_NN_GROUP + k;
            const float* __restrict__ p = os + (size_t)min(j, Mo - 1) * 3;
            q[3 * k] = j < Mo ? p[0] : INFINITY; q[3 * k + 1] = p[1]; q[3 * k + 2] = p[2];
        }
        take(q, nfull);
    }
    // the waves' shares are merged in index order (strict <: the lowest group on ties); the winning POINT is then named once per query, by the
    // thread that writes it: the first point of the winning group at the winning distance (the group's eight points re-read from memory and
    // measured with the same expression)
    s_best[wv][lane] = best.x; s_best[wv][64 + lane] = best.y;
    s_grp[wv][lane] = bga; s_grp[wv][64 + lane] = bgb;
    __syncthreads();
    if (tid < MM_NN_Q && q0 + tid < Nq) {
        float bb = s_best[0][tid]; int bg = s_grp[0][tid];
#pragma unroll
        for (int w = 1; w < MM_NN_WAVES; ++w) { const float c = s_best[w][tid]; if (c < bb) { bb = c; bg = s_grp[w][tid]; } }
        // (thread tid < 64 holds query q0 + tid in the first halves of its registers, thread 64 + l -- lane l of wave 1 -- query q0 + 64 + l in the second)
        const nn_f2 qx = {tid < 64 ? px.x : px.y, 0.f}, qy = {tid < 64 ? py.x : py.y, 0.f}, qz = {tid < 64 ? pz.x : pz.y, 0.f};
        int bi = 0;
        if (bb < INFINITY) {                                      // (nothing finite: index 0, as a plain scan from "best = inf, index 0" leaves it)
            int kk = 0;
#pragma unroll
            for (int k = MM_NN_GROUP - 1; k >= 0; --k) {
                const float* pa = os + (size_t)min(bg * MM_NN_GROUP + k, Mo - 1) * 3;
                if (nn_dist2(qx, qy, qz, pa[0], pa[1], pa[2]).x == bb) kk = k;
            }
            bi = bg * MM_NN_GROUP + kk;
        }
        dist[(size_t)b * Nq + q0 + tid] = bb; idx[(size_t)b * Nq + q0 + tid] = min(bi, Mo - 1);
    }
}

static int launch_nn_pair(int B, int N, int M, const float* x, const float* y, float* dx, int32_t* ix, float* dy, int32_t* iy, bool both, hipStream_t s) {
    const int nxq = (N + MM_NN_Q - 1) / MM_NN_Q, nyq = both ? (M + MM_NN_Q - 1) / MM_NN_Q : 0;
    hipLaunchKernelGGL(nn_pair_kernel, dim3(nxq + nyq, B), dim3(64 * MM_NN_WAVES), 0, s, N, M, x, y, dx, ix, dy, iy, nxq);
    return launch_ok("nearest_neighbour");
}

int launch_nn(int B, int N, int M, const float* x, const float* y, float* dist, int32_t* idx, hipStream_t s) {
    return launch_nn_pair(B, N, M, x, y, dist, idx, nullptr, nullptr, false, s);
}
int launch_nn_both(int B, int N, int M, const float* x, const float* y, float* dx, int32_t* ix, float* dy, int32_t* iy, hipStream_t s) {
    return launch_nn_pair(B, N, M, x, y, dx, ix, dy, iy, true, s);
}

}  // namespace mm
