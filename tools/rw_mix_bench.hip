// What the memory system delivers for a READ + WRITE mix (the layer kernels of the decoders write 1-3 bytes per byte they read;
// tools/lds_dma_bench.hip measures reads only): 256-thread workgroups, each thread loads RD float4 and stores WR float4 per
// step, contiguous 4-KB pieces per wave and array, grid-stride.   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/rw_mix_bench.hip -o tools/rw_mix_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int RD, int WR>
__global__ __launch_bounds__(256) void rw_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t steps_total) {
    // step s of the launch: the workgroup's 256 threads read RD consecutive 4-KB pieces of src and write WR pieces of dst
    for (size_t s = blockIdx.x; s < steps_total; s += gridDim.x) {
        f32x4 v[RD > 0 ? RD : 1];
        f32x4 acc = {1.f, 2.f, 3.f, 4.f};
#pragma unroll
        for (int i = 0; i < RD; ++i) v[i] = src[(s * RD + i) * 256 + threadIdx.x];
#pragma unroll
        for (int i = 0; i < RD; ++i) acc += v[i];
#pragma unroll
        for (int i = 0; i < WR; ++i) dst[(s * WR + i) * 256 + threadIdx.x] = acc * (float)(i + 1);
        if (WR == 0 && acc[0] == 12345.678f) dst[threadIdx.x] = acc;       // (keeps the loads of the read-only form alive)
    }
}

int main() {
    const size_t bytes = (size_t)3 << 30;                 // 3 GB per array (far beyond the 256 MB of MALL)
    f32x4 *src, *dst;
    CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes)); CK(hipMemset(src, 0, bytes)); CK(hipMemset(dst, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto kernel, int rd, int wr, int wgs) {
        const size_t per_step = (size_t)(rd + wr) * 4096;
        const size_t steps = (size_t)1 << 18;             // (rd + wr) GB per launch
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kernel, dim3(wgs), dim3(256), 0, 0, (const f32x4*)src, dst, steps);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kernel, dim3(wgs), dim3(256), 0, 0, (const f32x4*)src, dst, steps);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s %5d workgroups: %7.1f GB/s (read + write)\n", name, wgs, per_step * steps * 5 / (ms * 1e-3) * 1e-9);
    };
    for (int wgs : {512, 1024, 2048, 4096}) {
        run("read only (2 : 0)", rw_kernel<2, 0>, 2, 0, wgs);
        run("read 3 : write 1", rw_kernel<3, 1>, 3, 1, wgs);
        run("read 1 : write 1", rw_kernel<1, 1>, 1, 1, wgs);
        run("read 1 : write 2", rw_kernel<1, 2>, 1, 2, wgs);
        run("read 1 : write 3", rw_kernel<1, 3>, 1, 3, wgs);
        run("write only (0 : 2)", rw_kernel<0, 2>, 0, 2, wgs);
    }
    return 0;
}
