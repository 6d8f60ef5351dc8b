// fetch_calib.hip -- what rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the access patterns of THIS path, against byte counts known
// in advance (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access pattern before trusting an absolute").
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib profiles/tools/fetch_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE  -d out/f -o f -- /tmp/fetch_calib     (and again with --pmc WRITE_SIZE)
//   python profiles/tools/fetch_calib.py out  ->  profiles/r04_fetch_calibration.json
//
// Every kernel touches an array of 768 MiB (three times the Infinity Cache, 24 times the L2s) exactly once, so that re-use in a cache cannot
// hide a fetch.  The program prints, per kernel, the useful bytes and the 64-byte / 128-byte lines its accesses touch (computed on the host).
//   stream16      16 B per lane, coalesced (the guide's calibrated case: FETCH_SIZE reads 1/2)
//   stream4       4 B per lane, coalesced (pixel-addressed operands: face_idx, background / ground-truth planes)
//   record48      48-byte records gathered through a permuted id list, three 16-byte loads per lane (the face records of the walk)
//   texel4        4-byte texels: per lane a 2x2 bilinear footprint at a random place of a 256-wide plane (the texture fetch)
//   wstream16     16 B per lane written, coalesced (rgba)
//   wstream4      4 B per lane written, coalesced (face_idx)
//   wscatter32    32-byte records written through a permuted id list (gp)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
#include <set>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void stream16(const float4* __restrict__ a, size_t n, float* sink) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 123.456f) sink[0] = s;
}
__global__ void stream4(const float* __restrict__ a, size_t n, float* sink) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
    if (s == 123.456f) sink[0] = s;
}
__global__ void record48(const float4* __restrict__ a, const int* __restrict__ ids, size_t n, float* sink) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t f = (size_t)ids[i];
        const float4 p0 = a[f * 3], p1 = a[f * 3 + 1], p2 = a[f * 3 + 2];
        s += p0.x + p1.y + p2.z;
    }
    if (s == 123.456f) sink[0] = s;
}
__global__ void texel4(const float* __restrict__ a, const int* __restrict__ pos, size_t n, int Wt, float* sink) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = (size_t)pos[i];
        s += a[p] + a[p + 1] + a[p + Wt] + a[p + Wt + 1];
    }
    if (s == 123.456f) sink[0] = s;
}
__global__ void wstream16(float4* a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void wstream4(float* a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = (float)i;
}
__global__ void wscatter32(float4* a, const int* __restrict__ ids, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t f = (size_t)ids[i];
        a[f * 2] = make_float4(1.f, 2.f, 3.f, 4.f); a[f * 2 + 1] = make_float4(5.f, 6.f, 7.f, (float)i);
    }
}

int main() {
    const size_t BYTES = (size_t)768 << 20;
    void *buf, *ids, *sink;
    CK(hipMalloc(&buf, BYTES + 4096)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0, BYTES + 4096));
    std::mt19937_64 rng(1);
    const dim3 grid(256 * 16), block(256);
    printf("{\n");
    auto report = [&](const char* name, double useful, double lines64, double lines128, bool last = false) {
        printf(" \"%s\": {\"useful_bytes\": %.0f, \"bytes_in_64B_lines\": %.0f, \"bytes_in_128B_lines\": %.0f}%s\n", name, useful, lines64, lines128, last ? "" : ",");
    };
    // --- streaming reads
    hipLaunchKernelGGL(stream16, grid, block, 0, 0, (const float4*)buf, BYTES / 16, (float*)sink);
    report("stream16", (double)BYTES, (double)BYTES, (double)BYTES);
    hipLaunchKernelGGL(stream4, grid, block, 0, 0, (const float*)buf, BYTES / 4, (float*)sink);
    report("stream4", (double)BYTES, (double)BYTES, (double)BYTES);
    // --- 48-byte records through a permuted id list: the permutation is LOCAL (ids shuffled inside windows of 1 280 records = one image's faces),
    //     as a tile's candidate list is -- a wave's 64 records come from one image's 61 KB of records, not from anywhere in memory
    {
        const size_t n = BYTES / 48;
        std::vector<int> h(n);
        for (size_t i = 0; i < n; ++i) h[i] = (int)i;
        for (size_t w = 0; w + 1280 <= n; w += 1280) std::shuffle(h.begin() + w, h.begin() + w + 1280, rng);
        CK(hipMalloc(&ids, n * 4)); CK(hipMemcpy(ids, h.data(), n * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(record48, grid, block, 0, 0, (const float4*)buf, (const int*)ids, n, (float*)sink);
        CK(hipDeviceSynchronize()); CK(hipFree(ids));
        report("record48", (double)n * 48, (double)BYTES, (double)BYTES);          // every byte of the array is used exactly once
    }
    // --- bilinear texel footprints: lane i reads a 2x2 footprint; footprints of consecutive lanes are neighbours along a row with a random jitter
    //     (a tile's pixels map to neighbouring texels), rows visited once
    {
        const int Wt = 256;
        const size_t rows = BYTES / 4 / Wt, n = rows / 2 * (Wt / 2);              // one footprint per 2x2 texels: every texel read exactly once
        std::vector<int> h(n);
        size_t k = 0;
        for (size_t r = 0; r + 1 < rows; r += 2)
            for (int c = 0; c + 1 < Wt; c += 2) h[k++] = (int)(r * Wt + c);
        for (size_t w = 0; w + 64 <= n; w += 64) std::shuffle(h.begin() + w, h.begin() + w + 64, rng);   // (lanes of a wave in random order)
        CK(hipMalloc(&ids, n * 4)); CK(hipMemcpy(ids, h.data(), n * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(texel4, grid, block, 0, 0, (const float*)buf, (const int*)ids, n, Wt, (float*)sink);
        CK(hipDeviceSynchronize()); CK(hipFree(ids));
        report("texel4", (double)n * 16, (double)BYTES, (double)BYTES);
    }
    // --- writes
    hipLaunchKernelGGL(wstream16, grid, block, 0, 0, (float4*)buf, BYTES / 16);
    report("wstream16", (double)BYTES, (double)BYTES, (double)BYTES);
    hipLaunchKernelGGL(wstream4, grid, block, 0, 0, (float*)buf, BYTES / 4);
    report("wstream4", (double)BYTES, (double)BYTES, (double)BYTES);
    {
        const size_t n = BYTES / 32;
        std::vector<int> h(n);
        for (size_t i = 0; i < n; ++i) h[i] = (int)i;
        for (size_t w = 0; w + 4096 <= n; w += 4096) std::shuffle(h.begin() + w, h.begin() + w + 4096, rng);
        CK(hipMalloc(&ids, n * 4)); CK(hipMemcpy(ids, h.data(), n * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(wscatter32, grid, block, 0, 0, (float4*)buf, (const int*)ids, n);
        CK(hipDeviceSynchronize()); CK(hipFree(ids));
        report("wscatter32", (double)n * 32, (double)BYTES, (double)BYTES, true);
    }
    printf("}\n");
    CK(hipDeviceSynchronize());
    return 0;
}
