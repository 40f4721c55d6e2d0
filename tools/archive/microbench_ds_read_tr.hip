#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int mode, unsigned short* out) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) ((volatile unsigned short*)lds)[i] = (unsigned short)i;
    __syncthreads();
    int lane = threadIdx.x;
    unsigned addr = (unsigned)(size_t)(&lds[0]);   // low 32 bits = LDS offset
    if (mode == 0) addr += 0;
    else if (mode == 1) addr += lane * 8;
    else if (mode == 2) addr += (lane & 15) * 2 + (lane >> 4) * 128;
    else addr += (lane & 15) * 64 + (lane >> 4) * 8;
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[lane * 4 + 0] = v & 0xffff; out[lane * 4 + 1] = (v >> 16) & 0xffff;
    out[lane * 4 + 2] = (v >> 32) & 0xffff; out[lane * 4 + 3] = (v >> 48) & 0xffff;
}
int main() {
    unsigned short* d; (void)hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, mode, d);
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) { if (l < 20 || l == 32 || l == 48 || l==63) printf("L%02d:[%4d %4d %4d %4d] ", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); if (l % 4 == 3 && l < 20) printf("\n"); }
        printf("\n");
    }
    return 0;
}
