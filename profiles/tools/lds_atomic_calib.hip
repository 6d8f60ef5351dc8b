// lds_atomic_calib.hip -- what does an LDS atomic cost on gfx950, by type and by how many lanes hit one address?
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o /tmp/lds_atomic_calib profiles/tools/lds_atomic_calib.hip && /tmp/lds_atomic_calib
// W waves per SIMD on every CU (256-thread workgroups); each wave issues N atomics whose 64 lanes are spread over `distinct` words.
// Reports nanoseconds of LDS-unit time per wave-instruction per CU (kernel time x CUs / wave-instructions): the number a kernel's
// atomic count has to be multiplied by.  Finding (MI355X): ds_add_f32 ~81 ns whatever the conflict degree (the float adder takes
// the 64 lanes one after the other); see profiles/r02_lds_atomic_calibration.txt for the integer forms.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int distinct, int iters) {
    __shared__ unsigned long long s[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    s[wave][lane] = 0ull;
    __syncthreads();
    void* p = &s[wave][lane % distinct];
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { atomicAdd((float*)p, 1.0f); atomicAdd((float*)p, 2.0f); atomicAdd((float*)p, 3.0f); atomicAdd((float*)p, 4.0f); }
        if (KIND == 1) { atomicAdd((unsigned*)p, 1u); atomicAdd((unsigned*)p, 2u); atomicAdd((unsigned*)p, 3u); atomicAdd((unsigned*)p, 4u); }
        if (KIND == 2) { atomicAdd((unsigned long long*)p, 1ull); atomicAdd((unsigned long long*)p, 2ull); atomicAdd((unsigned long long*)p, 3ull); atomicAdd((unsigned long long*)p, 4ull); }
        if (KIND == 3) { atomicMax((unsigned long long*)p, (unsigned long long)(i * 4 + lane)); atomicMax((unsigned long long*)p, (unsigned long long)(i * 4 + 1 + lane));
                         atomicMax((unsigned long long*)p, (unsigned long long)(i * 4 + 2 + lane)); atomicMax((unsigned long long*)p, (unsigned long long)(i * 4 + 3 + lane)); }
        if (KIND == 4) { volatile unsigned* q = (volatile unsigned*)p; q[0] = i; q[0] = i + 1; q[0] = i + 2; q[0] = i + 3; }     // plain ds_write_b32
        if (KIND == 5) { float v = 1.f; asm volatile("ds_add_rtn_f32 %0, %1, %0\\n\\ts_waitcnt lgkmcnt(0)" : "+v"(v) : "v"((unsigned)(size_t)p) : "memory"); (void)v;
                         atomicAdd((float*)p, 2.0f); atomicAdd((float*)p, 3.0f); atomicAdd((float*)p, 4.0f); }
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = (float)s[wave][lane];
}
template <int KIND>
void run(const char* name, float* out) {
    const int iters = 256;
    for (int wps : {1, 8}) {
        const int blocks = 256 * wps;
        for (int d : {64, 8, 1}) {
            hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, d, iters);
            hipDeviceSynchronize();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, d, iters);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-14s waves/SIMD %d  lanes per address %2d : %6.1f ns per wave-instruction per CU\n", name, wps, 64 / d, ms * 1e6 / (iters * 4.0 * wps * 4));
        }
    }
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("ds_add_f32", out); run<1>("ds_add_u32", out); run<2>("ds_add_u64", out); run<3>("ds_max_u64", out); run<4>("ds_write_b32", out);
    return 0;
}
