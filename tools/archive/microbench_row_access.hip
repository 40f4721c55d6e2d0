// Cost of the T-layout's token-strided row accesses vs. row-contiguous ones, at layer-kernel shape:
// 4 waves x 16 tokens per workgroup, [n,128] bf16 stores (8 x b64 per lane vs 4 x b128 per lane through an LDS
// transposition-free contiguous pattern) and [n,128] fp32 loads (8 x b128 per lane, strided vs contiguous).
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench_row_access.hip -o /tmp/mb_row && /tmp/mb_row
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

template <int MODE, int REP>
__global__ __launch_bounds__(256) void k_store(unsigned short* dst, int n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x * 4 + wave;
    const __amdgpu_buffer_rsrc_t r = rsrc(dst, n * 256);
    for (int rep = 0; rep < REP; ++rep) {
        if (MODE == 0) {                 // T-layout: lane (t = l & 15, g = l >> 4) writes 8 B at row t, col 16 ct + 4 g
            const int off = (tile * 16 + (lane & 15)) * 256 + 8 * (lane >> 4);
#pragma unroll
            for (int ct = 0; ct < 8; ++ct)
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{(unsigned)lane, (unsigned)(ct + rep)}, r, off + 32 * ct, 0, 0);
        } else {                         // contiguous: lane writes 16 B, 16 lanes = one 256-byte row, 4 rows per instruction
            const int off = (tile * 16 + (lane >> 4)) * 256 + 16 * (lane & 15);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{(unsigned)lane, (unsigned)q, (unsigned)rep, 0u}, r, off + 4 * 256 * q, 0, 0);
        }
    }
}

template <int MODE, int REP>
__global__ __launch_bounds__(256) void k_load(const float* src, int n, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x * 4 + wave;
    const __amdgpu_buffer_rsrc_t r = rsrc(src, n * 512);
    float acc = 0.f;
    for (int rep = 0; rep < REP; ++rep) {
        u32x4 v[8];
        if (MODE == 0) {
            const int off = (tile * 16 + (lane & 15)) * 512 + 16 * (lane >> 4);
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) v[ct] = __builtin_amdgcn_raw_buffer_load_b128(r, off + 64 * ct, 0, 0);
        } else {                         // 32 lanes = one 512-byte row, 2 rows per instruction
            const int off = (tile * 16 + (lane >> 5)) * 512 + 16 * (lane & 31);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = __builtin_amdgcn_raw_buffer_load_b128(r, off + 2 * 512 * q, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += __uint_as_float(v[q][0]) + __uint_as_float(v[q][3]);
        asm volatile("" : "+v"(acc));
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <typename F>
static float time_it(F f, int iters = 50) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return 1e3f * ms / iters;
}

int main() {
    for (int n : {6720, 22336, 131072}) {
        unsigned short* d; float* s; float* sink;
        hipMalloc(&d, (size_t)n * 256); hipMalloc(&s, (size_t)n * 512); hipMalloc(&sink, 4);
        hipMemset(s, 0, (size_t)n * 512);
        const int g = n / 64;
        printf("n=%d (%d workgroups)\n", n, g);
        printf("  bf16 row stores x1 : T-layout %.1f us, contiguous %.1f us\n",
               time_it([&] { hipLaunchKernelGGL((k_store<0, 1>), dim3(g), dim3(256), 0, 0, d, n); }),
               time_it([&] { hipLaunchKernelGGL((k_store<1, 1>), dim3(g), dim3(256), 0, 0, d, n); }));
        printf("  bf16 row stores x8 : T-layout %.1f us, contiguous %.1f us\n",
               time_it([&] { hipLaunchKernelGGL((k_store<0, 8>), dim3(g), dim3(256), 0, 0, d, n); }),
               time_it([&] { hipLaunchKernelGGL((k_store<1, 8>), dim3(g), dim3(256), 0, 0, d, n); }));
        printf("  fp32 row loads  x1 : T-layout %.1f us, contiguous %.1f us\n",
               time_it([&] { hipLaunchKernelGGL((k_load<0, 1>), dim3(g), dim3(256), 0, 0, s, n, sink); }),
               time_it([&] { hipLaunchKernelGGL((k_load<1, 1>), dim3(g), dim3(256), 0, 0, s, n, sink); }));
        printf("  fp32 row loads  x8 : T-layout %.1f us, contiguous %.1f us\n",
               time_it([&] { hipLaunchKernelGGL((k_load<0, 8>), dim3(g), dim3(256), 0, 0, s, n, sink); }),
               time_it([&] { hipLaunchKernelGGL((k_load<1, 8>), dim3(g), dim3(256), 0, 0, s, n, sink); }));
        hipFree(d); hipFree(s); hipFree(sink);
    }
    return 0;
}
