// mm_nn.hip -- brute-force nearest neighbour between two point clouds per batch item, for the chamfer loss the
// reference takes from pytorch3d (pytorch3d.loss.chamfer_distance -> knn_points(K=1); call sites
// /root/reference/networks.py:342,356 and trainer.py:445,469,483; pytorch3d 0.7.0 has no ROCm build).
// One thread per query point; the other cloud is streamed through LDS in tiles of 256 points.  Ties keep the lowest index.
#include "mm_device.h"

namespace mm {

__global__ __launch_bounds__(256) void nn_kernel(int B, int N, int M, const float* __restrict__ x, const float* __restrict__ y,
                                                 float* __restrict__ dist, int32_t* __restrict__ idx) {
    __shared__ float s_y[256 * 3];
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (i < N) { const float* p = x + ((size_t)b * N + i) * 3; px = p[0]; py = p[1]; pz = p[2]; }
    float best = INFINITY; int bi = 0;
    for (int j0 = 0; j0 < M; j0 += 256) {
        const int nj = min(256, M - j0);
        __syncthreads();
        for (int k = threadIdx.x; k < nj * 3; k += 256) s_y[k] = y[((size_t)b * M + j0) * 3 + k];
        __syncthreads();
        for (int j = 0; j < nj; ++j) {
            const float dx = px - s_y[j * 3], dy = py - s_y[j * 3 + 1], dz = pz - s_y[j * 3 + 2];
            const float d = (dx * dx + dy * dy) + dz * dz;
            if (d < best) { best = d; bi = j0 + j; }
        }
    }
    if (i < N) { dist[(size_t)b * N + i] = best; idx[(size_t)b * N + i] = bi; }
}

int launch_nn(int B, int N, int M, const float* x, const float* y, float* dist, int32_t* idx, hipStream_t s) {
    hipLaunchKernelGGL(nn_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s, B, N, M, x, y, dist, idx);
    return launch_ok("nearest_neighbour");
}

}  // namespace mm
