// Is one v_mfma_f32_16x16x32_bf16 over [tile A | tile B] k slots equal to two chained 16x16x16 MFMAs?  (window.hip
// mfma32_pair).  Build: hipcc --offload-arch=gfx950 -O2 tools/microbench_mfma_pair.hip -o /tmp/mfma_pair && /tmp/mfma_pair
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_a;
__device__ unsigned short f2bf(float f) { union { __bf16 h; unsigned short u; } c; c.h = (__bf16)f; return c.u; }
__global__ void k(float* out) {
    const int lane = threadIdx.x;
    bf16x4 a0, a1, b0, b1;
    for (int r = 0; r < 4; ++r) {
        a0[r] = (short)f2bf((float)((lane * 7 + r * 3) % 11) - 5.0f);
        a1[r] = (short)f2bf((float)((lane * 5 + r * 2) % 13) - 6.0f);
        b0[r] = (short)f2bf((float)((lane * 3 + r * 5) % 7) - 3.0f);
        b1[r] = (short)f2bf((float)((lane * 11 + r) % 9) - 4.0f);
    }
    f32x4 z = {0, 0, 0, 0};
    f32x4 c16 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0, b0, z, 0, 0, 0);
    c16 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1, b1, c16, 0, 0, 0);
    union { struct { bf16x4 lo, hi; } p; bf16x8_a v; } fa, fb;
    fa.p.lo = a0; fa.p.hi = a1; fb.p.lo = b0; fb.p.hi = b1;
    f32x4 c32 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.v, fb.v, z, 0, 0, 0);
    for (int r = 0; r < 4; ++r) { out[lane * 8 + r] = c16[r]; out[lane * 8 + 4 + r] = c32[r]; }
}
int main() {
    float* d; hipMalloc(&d, 64 * 8 * 4);
    k<<<1, 64>>>(d);
    float h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (h[l * 8 + r] != h[l * 8 + 4 + r]) ++bad;
    printf("mismatches %d of 256; lane0: %g %g %g %g | %g %g %g %g\n", bad, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    return 0;
}
