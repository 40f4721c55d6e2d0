// Shared helpers for libgeomae_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define GEOMAE_OK 0
#define GEOMAE_ERR_ARG (-1)
#define GEOMAE_ERR_HIP (-2)
#define GEOMAE_ERR_WORKSPACE (-3)
#define GEOMAE_ERR_CONFIG (-4)

namespace geomae {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return GEOMAE_ERR_HIP;
    }
    return GEOMAE_OK;
}

#define GEOMAE_REQUIRE(cond, ...)                 \
    do {                                          \
        if (!(cond)) {                            \
            geomae::set_error(__VA_ARGS__);       \
            return GEOMAE_ERR_ARG;                \
        }                                         \
    } while (0)

#define GEOMAE_HIP(call)                                                      \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) {                                               \
            geomae::set_error("%s: %s", #call, hipGetErrorString(e_));        \
            return GEOMAE_ERR_HIP;                                            \
        }                                                                     \
    } while (0)

constexpr int kWave = 64;  // CDNA wavefront

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// memory-bound streaming kernels: cap the grid and grid-stride (guide G11)
inline int stream_grid(int64_t work_items, int block) {
    int64_t g = (work_items + block - 1) / block;
    const int64_t cap = 256 * 8;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace geomae
