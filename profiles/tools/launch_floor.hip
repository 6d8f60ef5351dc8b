// launch_floor.hip -- what a kernel boundary costs on this box: the back-to-back period of N dependent launches in one stream for
//   (a) an empty kernel of one workgroup, (b) an empty kernel of 4 608 workgroups of 256 threads (raster_fwd's grid at config 2),
//   (c) one workgroup whose wave makes ONE dependent trip to memory (load -> store), (d) 4 608 workgroups doing the same,
//   (e) a chain of K dependent trips in one wave (the per-trip latency a wave's chain is made of).
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/launch_floor profiles/tools/launch_floor.hip ; run on a GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void k_empty() {}
__global__ void k_trip(const int* in, int* out) { const int i = blockIdx.x * blockDim.x + threadIdx.x; out[i] = in[i] + 1; }
__global__ void k_chain(const int* next, int* out, int K) {          // pointer chase: K dependent loads (each a trip to L2 / HBM)
    int p = threadIdx.x;
    for (int k = 0; k < K; ++k) p = next[p];
    out[threadIdx.x] = p;
}
template <class F> static double period_us(F&& launch, int n, hipStream_t s) {
    for (int i = 0; i < 50; ++i) launch();
    (void)hipStreamSynchronize(s);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < n; ++i) launch();
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3 / n;
}
int main() {
    hipStream_t s; (void)hipStreamCreate(&s);
    const int NWG = 4608, NT = 256;
    int *in, *out, *next;
    (void)hipMalloc(&in, sizeof(int) * NWG * NT); (void)hipMalloc(&out, sizeof(int) * NWG * NT); (void)hipMalloc(&next, sizeof(int) * (1 << 22));
    (void)hipMemset(in, 0, sizeof(int) * NWG * NT);
    {   // a permutation with a long stride so that consecutive loads of the chase miss the L1 and mostly the L2 line just fetched
        int* h = (int*)malloc(sizeof(int) * (1 << 22));
        for (int i = 0; i < (1 << 22); ++i) h[i] = (int)(((long long)i * 1048583 + 12345) & ((1 << 22) - 1));
        (void)hipMemcpy(next, h, sizeof(int) * (1 << 22), hipMemcpyHostToDevice); free(h);
    }
    const int n = 2000;
    printf("back-to-back period of dependent launches in one stream (HIP events around %d launches), us per launch:\n", n);
    printf("  (a) empty kernel, 1 workgroup x 64            %.2f\n", period_us([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); }, n, s));
    printf("  (b) empty kernel, 4608 workgroups x 256       %.2f\n", period_us([&] { hipLaunchKernelGGL(k_empty, dim3(NWG), dim3(NT), 0, s); }, n, s));
    printf("  (b2) empty kernel, 1280 workgroups x 256      %.2f\n", period_us([&] { hipLaunchKernelGGL(k_empty, dim3(1280), dim3(NT), 0, s); }, n, s));
    printf("  (c) one trip (load -> store), 1 x 64          %.2f\n", period_us([&] { hipLaunchKernelGGL(k_trip, dim3(1), dim3(64), 0, s, in, out); }, n, s));
    printf("  (d) one trip, 4608 x 256 (9.4 MB moved)       %.2f\n", period_us([&] { hipLaunchKernelGGL(k_trip, dim3(NWG), dim3(NT), 0, s, in, out); }, n, s));
    for (int K : {1, 2, 4, 8, 16, 32}) {
        printf("  (e) chain of %2d dependent loads, 1 x 64      %.2f\n", K, period_us([&] { hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, s, next, out, K); }, n, s));
    }
    // six empty kernels per "step": the floor of a six-launch decomposition
    printf("  six empty launches (1 workgroup each)          %.2f per six\n", 6 * period_us([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); }, 6 * 500, s));
    return 0;
}
