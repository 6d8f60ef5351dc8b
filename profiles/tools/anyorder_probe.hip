// anyorder_probe.hip -- does hipExtAnyOrderLaunch let the NEXT kernel of a stream start while the previous one is still running on gfx950?
// (hip_ext.h says the flag is not supported on GFX9xx boards; this measures it.)   hipcc --offload-arch=gfx950 -O2 -o /tmp/anyorder profiles/tools/anyorder_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void slow(unsigned long long* t, int us) {           // one workgroup that stays for `us` microseconds
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0) t[0] = wall_clock64();                 // its end
}
__global__ void mark(unsigned long long* t) { if (threadIdx.x == 0 && blockIdx.x == 0) t[1] = wall_clock64(); }   // its start

int main() {
    unsigned long long* d; unsigned long long h[2];
    CK(hipMalloc(&d, 16));
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int flag = 0; flag < 2; ++flag) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(d, 0, 16, s));
            hipLaunchKernelGGL(slow, dim3(1), dim3(64), 0, s, d, 200);
            hipExtLaunchKernelGGL(mark, dim3(256), dim3(64), 0, s, nullptr, nullptr, flag ? hipExtAnyOrderLaunch : 0, d);
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
            printf("flags=%d: next kernel started %.1f us %s the slow kernel's end\n", flag, (double)((long long)h[1] - (long long)h[0]) / 100.0,
                   h[1] >= h[0] ? "AFTER" : "BEFORE");
        }
    }
    return 0;
}
