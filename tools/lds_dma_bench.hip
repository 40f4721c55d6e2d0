// What one CU's global -> LDS paths deliver: a workgroup of 512 threads streams a contiguous range in 24-KB slabs through a ring,
// (A) with buffer_load_dwordx4 ... lds (the LDS-direct path of csrc/dw_device.h), (B) with buffer_load_dwordx4 into registers
// and ds_write_b128, (C) registers only (no LDS write).  Nothing is computed; one dword per slab is read back so that the
// loads cannot be dropped.   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lds_dma_bench.hip -o tools/lds_dma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void lds_void;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kSlab = 24576, kNI = 3;          // 3 x 1 KB per wave and slab, 8 waves

template <int RING>
__global__ __launch_bounds__(512, 2) void dma_kernel(const char* src, size_t bytes_per_wg, unsigned* out) {
    __shared__ __attribute__((aligned(1024))) char lds[RING * 32768];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = src + (size_t)blockIdx.x * bytes_per_wg;
    const int nslab = (int)(bytes_per_wg / kSlab);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)bytes_per_wg, 0x00020000);
    auto issue = [&](int slab) {
        char* dst = lds + (slab % RING) * 32768 + w * 1024;
        const int vo = slab < nslab ? slab * kSlab + w * 1024 + lane * 16 : 0x7fff0000;
#pragma unroll
        for (int i = 0; i < kNI; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(dst + 8192 * i), 16, vo + 8192 * i, 0, 0, 0);
    };
    for (int s = 0; s < RING - 1; ++s) issue(s);
    unsigned acc = 0;
    for (int k = 0; k < nslab; ++k) {
        if (RING == 2) __builtin_amdgcn_s_waitcnt(0x0f70);
        else if (RING == 3) __builtin_amdgcn_s_waitcnt(0x0f73);
        else if (RING == 4) __builtin_amdgcn_s_waitcnt(0x0f76);
        __builtin_amdgcn_s_barrier();
        issue(k + RING - 1);
        acc += *reinterpret_cast<const unsigned*>(lds + (k % RING) * 32768 + threadIdx.x * 4);
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}

template <int DEPTH, bool WRITE>
__global__ __launch_bounds__(512, 2) void reg_kernel(const char* src, size_t bytes_per_wg, unsigned* out) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * 32768];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = src + (size_t)blockIdx.x * bytes_per_wg;
    const int nslab = (int)(bytes_per_wg / kSlab);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)bytes_per_wg, 0x00020000);
    u32x4 r[DEPTH][kNI];
    auto issue = [&](int slab, u32x4 (&v)[kNI]) {
        const int vo = slab < nslab ? slab * kSlab + w * 1024 + lane * 16 : 0x7fff0000;
#pragma unroll
        for (int i = 0; i < kNI; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo + 8192 * i, 0, 0);
    };
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) issue(s, r[s]);
    unsigned acc = 0;
    for (int k0 = 0; k0 < nslab; k0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int k = k0 + d;
            char* dst = lds + (k & 1) * 32768 + w * 1024 + lane * 16;
            if (WRITE) {
#pragma unroll
                for (int i = 0; i < kNI; ++i) *reinterpret_cast<u32x4*>(dst + 8192 * i) = r[d][i];
            } else {
#pragma unroll
                for (int i = 0; i < kNI; ++i) acc += r[d][i].x ^ r[d][i].w;
            }
            issue(k + DEPTH, r[d]);
            if (WRITE) {
                __syncthreads();
                acc += *reinterpret_cast<const unsigned*>(lds + (k & 1) * 32768 + ((threadIdx.x * 4 + 2048) & 24575));
            }
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
    const size_t per_wg = (size_t)kSlab * 256;        // 6 MB per workgroup
    char* src; unsigned* out;
    CK(hipMalloc(&src, per_wg * 256)); CK(hipMemset(src, 1, per_wg * 256)); CK(hipMalloc(&out, 256 * 512 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, auto kernel, int wgs) {
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kernel, dim3(wgs), dim3(512), 0, 0, (const char*)src, per_wg, out);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kernel, dim3(wgs), dim3(512), 0, 0, (const char*)src, per_wg, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double gbs = per_wg * wgs * 10 / (ms * 1e-3) * 1e-9;
        printf("%-44s %3d workgroups: %7.1f GB/s = %5.1f GB/s per workgroup\n", name, wgs, gbs, gbs / wgs);
    };
    for (int wgs : {16, 80, 192, 256}) {
        time("LDS-direct, ring 2", dma_kernel<2>, wgs);
        time("LDS-direct, ring 3", dma_kernel<3>, wgs);
        time("LDS-direct, ring 4", dma_kernel<4>, wgs);
        time("registers (2 slabs ahead) + ds_write_b128", reg_kernel<2, true>, wgs);
        time("registers (3 slabs ahead) + ds_write_b128", reg_kernel<3, true>, wgs);
        time("registers (4 slabs ahead) + ds_write_b128", reg_kernel<4, true>, wgs);
        time("registers only (3 slabs ahead)", reg_kernel<3, false>, wgs);
    }
    return 0;
}
