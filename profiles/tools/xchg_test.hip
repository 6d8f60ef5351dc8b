// Lane exchanges lane <-> lane ^ S without the LDS crossbar (ds_bpermute): DPP for S = 1, 2, 4, 8 and gfx950's v_permlane16_swap /
// v_permlane32_swap for S = 16, 32.  Checks every stride against the definition, the 64x64 bit transpose built from them against a
// host transpose, and the DPP prefix scan against a serial sum.   hipcc --offload-arch=gfx950 -O3 -I3d-magic-mirror_amd/csrc -Iinclude profiles/tools/xchg_test.hip -o /tmp/xchg_test && /tmp/xchg_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "mm_device.h"
__global__ void k(unsigned* out, uint64_t* tin, uint64_t* tout) {
    const int lane = threadIdx.x & 63;
    unsigned v = lane * 3 + 7;
    out[lane + 0 * 64] = mm::lane_xchg<1>(v, lane); out[lane + 1 * 64] = mm::lane_xchg<2>(v, lane); out[lane + 2 * 64] = mm::lane_xchg<4>(v, lane);
    out[lane + 3 * 64] = mm::lane_xchg<8>(v, lane); out[lane + 4 * 64] = mm::lane_xchg<16>(v, lane); out[lane + 5 * 64] = mm::lane_xchg<32>(v, lane);
    tout[lane] = mm::wave_transpose64(tin[lane], lane);
    int tot; const int pre = mm::wave_prefix_excl((int)(tin[lane] & 1023), lane, tot);
    out[lane + 6 * 64] = (unsigned)pre; out[lane + 7 * 64] = (unsigned)tot;
}
int main() {
    unsigned* d; uint64_t *ti, *to; hipMalloc(&d, 8 * 64 * 4); hipMalloc(&ti, 512); hipMalloc(&to, 512);
    uint64_t hin[64], hout[64]; uint64_t s = 88172645463325252ull;
    for (int i = 0; i < 64; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; hin[i] = s; }
    hipMemcpy(ti, hin, 512, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, ti, to); unsigned h[8 * 64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(hout, to, 512, hipMemcpyDeviceToHost);
    int bad = 0; const int S[6] = {1, 2, 4, 8, 16, 32};
    for (int s = 0; s < 6; ++s) for (int l = 0; l < 64; ++l) if (h[s * 64 + l] != (unsigned)((l ^ S[s]) * 3 + 7)) { if (bad < 10) printf("S=%d lane %d got %u want %u\n", S[s], l, h[s*64+l], (l ^ S[s]) * 3 + 7); ++bad; }
    for (int i = 0; i < 64; ++i) { uint64_t w = 0; for (int j = 0; j < 64; ++j) w |= ((hin[j] >> i) & 1ull) << j; if (w != hout[i]) { if (bad < 20) printf("transpose row %d wrong\n", i); ++bad; } }
    { unsigned run = 0; for (int l = 0; l < 64; ++l) { if (h[6 * 64 + l] != run) { if (bad < 30) printf("prefix lane %d got %u want %u\n", l, h[6*64+l], run); ++bad; } run += (unsigned)(hin[l] & 1023); }
      for (int l = 0; l < 64; ++l) if (h[7 * 64 + l] != run) { if (bad < 40) printf("total lane %d got %u want %u\n", l, h[7*64+l], run); ++bad; } }
    printf("bad %d\n", bad); return bad != 0;
}
