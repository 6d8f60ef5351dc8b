// valu_calib.hip -- how many cycles does a wave64 VALU instruction occupy its SIMD on this chip?
//
// N independent v_fma_f32 per wave (8 accumulators, no dependence between neighbours), W waves per SIMD, every CU busy.
// Each wave brackets its loop with s_memtime (shader clock), so the answer does not depend on knowing the DVFS clock:
//     cycles per wave-instruction per SIMD = (loop cycles of one wave) / (instructions per wave * waves per SIMD)
// (W waves share the SIMD's issue port; W = 8 saturates it).  Also reported: the chip-wide instruction rate from HIP
// events, i.e. the effective clock = rate * cycles / (1024 SIMDs).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_calib profiles/tools/valu_calib.hip && /tmp/valu_calib
// Prints one JSON object.  MI355X_MICROARCH.md says 2 (SIMD-32); GCN-lineage SIMD-16 would give 4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kUnroll = 8;          // independent accumulators
constexpr int kInner = 16;          // fma groups per loop trip
constexpr int kTrips = 512;         // loop trips  -> 8 * 16 * 512 = 65536 v_fma_f32 per wave

__global__ __launch_bounds__(256) void fma_kernel(float* out, unsigned long long* cycles, float b, float c) {
    float a0 = threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < kTrips; ++t) {
#pragma unroll
        for (int i = 0; i < kInner; ++i) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                         "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    out[gid] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    if ((threadIdx.x & 63) == 0) cycles[gid >> 6] = t1 - t0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const long long insts_per_wave = (long long)kUnroll * kInner * kTrips;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz_reported\": %d, \"insts_per_wave\": %lld, \"runs\": [", prop.gcnArchName, cus, prop.clockRate, insts_per_wave);
    bool first = true;
    for (int waves_per_simd : {1, 2, 4, 8}) {
        const int blocks = cus * waves_per_simd;                // 256 threads = 4 waves = one wave per SIMD of a CU
        const size_t nthreads = (size_t)blocks * 256;
        float* out; unsigned long long* cyc;
        CHECK(hipMalloc(&out, nthreads * sizeof(float)));
        CHECK(hipMalloc(&cyc, nthreads / 64 * sizeof(unsigned long long)));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fma_kernel, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0000001f, 1e-9f);
        CHECK(hipDeviceSynchronize());
        const int reps = 10;
        CHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(fma_kernel, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0000001f, 1e-9f);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(nthreads / 64);
        CHECK(hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2];
        // s_memtime ticks are shader cycles on gfx950 (MI355X_MICROARCH.md, constants table); the event-timed rate is the cross-check
        const double us_per_launch = ms * 1e3 / reps;
        const double wave_insts_per_us_per_simd = (double)insts_per_wave * waves_per_simd / us_per_launch;   // per SIMD (every SIMD holds waves_per_simd)
        printf("%s{\"waves_per_simd\": %d, \"us_per_launch\": %.2f, \"wave_insts_per_us_per_simd\": %.1f, \"memtime_ticks_median\": %.0f, "
               "\"cycles_per_inst_if_2400MHz\": %.3f, \"cycles_per_inst_by_memtime\": %.3f, \"implied_clock_MHz\": %.0f}",
               first ? "" : ", ", waves_per_simd, us_per_launch, wave_insts_per_us_per_simd, med,
               2400.0 / wave_insts_per_us_per_simd, med / ((double)insts_per_wave * waves_per_simd),
               wave_insts_per_us_per_simd * med / ((double)insts_per_wave * waves_per_simd));
        first = false;
        CHECK(hipFree(out)); CHECK(hipFree(cyc));
    }
    printf("]}\n");
    return 0;
}
