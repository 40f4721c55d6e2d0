// Dynamic voxelization for gfx950 -- bit-exact with the reference's dynamic_voxelize
// (mmdet3d/ops/voxel/src/voxelization_cuda.cu:22-63 kernel, :375-377 grid size; CPU twin
// voxelization_cpu.cpp:6-40).  This fork CLAMPS out-of-range points into the border cell.
//
// MI355X design: one pass over the AoS [N, stride] fp32 rows.  A 256-thread workgroup stages
// 256 rows (256*stride contiguous dwords) through LDS with fully coalesced dword loads (the
// reference reads them with a 20-byte lane stride), then every lane emits all three
// resolutions at once as one 16-byte (b, z, y, x) store per level.  HBM-bound:
// 4*stride B read + 16 B written per point-level.  IEEE fp32 subtract / divide / floor:
// this file must never be built with -ffast-math (hipcc's default keeps fp32 division
// correctly rounded).
#include "common.h"
#include "../../include/geomae_hip.h"

namespace geomae {

struct LevelParam {
    float vs[3];   // x, y, z
    int grid[3];   // x, y, z
};

struct VoxParam {
    float lo[3];
    LevelParam lv[3];
    int n_levels;
};

__device__ __forceinline__ int cell(float p, float lo, float vs, int grid) {
    float q = floorf((p - lo) / vs);
    if (!(q >= 0.0f)) return 0;          // negative or NaN
    if (q >= (float)grid) return grid - 1;
    return (int)q;
}

constexpr int kVoxBlock = 256;
constexpr int kMaxStride = 8;

// out[l]: [n, 4] int32 (b, z, y, x) when with_batch, else [n, 3] (z, y, x)
template <bool WITH_BATCH>
__global__ __launch_bounds__(kVoxBlock) void voxelize_kernel(
    const float* __restrict__ pts, int64_t n, int stride, const int32_t* __restrict__ batch_offsets,
    int n_batch, VoxParam P, int32_t* __restrict__ out0, int32_t* __restrict__ out1,
    int32_t* __restrict__ out2) {
    __shared__ float s[kVoxBlock * kMaxStride];
    const int64_t n_tiles = (n + kVoxBlock - 1) / kVoxBlock;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t base = tile * kVoxBlock;
        const int rows = (int)((n - base) < kVoxBlock ? (n - base) : kVoxBlock);
        const int words = rows * stride;
        const float* src = pts + base * stride;
        for (int w = threadIdx.x; w < words; w += kVoxBlock) s[w] = src[w];
        __syncthreads();
        if ((int)threadIdx.x < rows) {
            const int64_t i = base + threadIdx.x;
            const float x = s[threadIdx.x * stride + 0];
            const float y = s[threadIdx.x * stride + 1];
            const float z = s[threadIdx.x * stride + 2];
            int b = 0;
            if (WITH_BATCH) {
                // batch_offsets is tiny (B+1 entries, scalar-cache resident)
                for (int k = 1; k < n_batch; ++k) b += (i >= (int64_t)batch_offsets[k]) ? 1 : 0;
            }
            int32_t* outs[3] = {out0, out1, out2};
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                if (l < P.n_levels) {
                    const LevelParam& L = P.lv[l];
                    const int cx = cell(x, P.lo[0], L.vs[0], L.grid[0]);
                    const int cy = cell(y, P.lo[1], L.vs[1], L.grid[1]);
                    const int cz = cell(z, P.lo[2], L.vs[2], L.grid[2]);
                    if (WITH_BATCH) {
                        reinterpret_cast<int4*>(outs[l])[i] = make_int4(b, cz, cy, cx);
                    } else {
                        int32_t* o = outs[l] + i * 3;
                        o[0] = cz; o[1] = cy; o[2] = cx;
                    }
                }
            }
        }
        __syncthreads();
    }
}

static int fill_level(LevelParam& L, const float* vs, const float* range) {
    for (int j = 0; j < 3; ++j) {
        if (!(vs[j] > 0.0f)) {
            set_error("voxel_size[%d] must be > 0", j);
            return GEOMAE_ERR_ARG;
        }
        L.vs[j] = vs[j];
        // fp32 arithmetic, exactly as the reference's host code (voxelization_cuda.cu:375-377)
        volatile float span = range[3 + j] - range[j];
        volatile float q = span / vs[j];
        L.grid[j] = (int)ceilf(q);
        if (L.grid[j] <= 0) {
            set_error("empty grid on axis %d", j);
            return GEOMAE_ERR_ARG;
        }
    }
    return GEOMAE_OK;
}

}  // namespace geomae

using namespace geomae;

extern "C" int geomae_grid_size(const float* voxel_size, const float* coors_range, int32_t* grid_xyz) {
    GEOMAE_REQUIRE(voxel_size && coors_range && grid_xyz, "geomae_grid_size: null argument");
    LevelParam L;
    int rc = fill_level(L, voxel_size, coors_range);
    if (rc) return rc;
    for (int j = 0; j < 3; ++j) grid_xyz[j] = L.grid[j];
    return GEOMAE_OK;
}

extern "C" int geomae_dynamic_voxelize(const float* points, int64_t num_points, int32_t num_features,
                                       const float* voxel_size, const float* coors_range,
                                       int32_t* coors, hipStream_t stream) {
    GEOMAE_REQUIRE(num_points >= 0, "dynamic_voxelize: negative num_points");
    if (num_points == 0) return GEOMAE_OK;
    GEOMAE_REQUIRE(points && coors && voxel_size && coors_range, "dynamic_voxelize: null argument");
    GEOMAE_REQUIRE(num_features >= 3 && num_features <= kMaxStride,
                   "dynamic_voxelize: num_features must be in [3, %d], got %d", kMaxStride, num_features);
    VoxParam P;
    P.n_levels = 1;
    for (int j = 0; j < 3; ++j) P.lo[j] = coors_range[j];
    int rc = fill_level(P.lv[0], voxel_size, coors_range);
    if (rc) return rc;
    P.lv[1] = P.lv[0];
    P.lv[2] = P.lv[0];
    const int grid = stream_grid(num_points, kVoxBlock);
    hipLaunchKernelGGL(voxelize_kernel<false>, dim3(grid), dim3(kVoxBlock), 0, stream, points, num_points,
                       num_features, nullptr, 0, P, coors, nullptr, nullptr);
    return check_launch("voxelize_kernel");
}

extern "C" int geomae_voxelize_batch3(const float* points, int64_t num_points, int32_t num_features,
                                      const int32_t* batch_offsets, int32_t batch_size,
                                      const float* voxel_size_top, const float* voxel_size_med,
                                      const float* voxel_size_low, const float* coors_range,
                                      int32_t* coors_top, int32_t* coors_med, int32_t* coors_low,
                                      hipStream_t stream) {
    GEOMAE_REQUIRE(num_points >= 0 && batch_size >= 1, "voxelize_batch3: bad sizes");
    if (num_points == 0) return GEOMAE_OK;
    GEOMAE_REQUIRE(points && batch_offsets && coors_top && coors_med && coors_low, "voxelize_batch3: null argument");
    GEOMAE_REQUIRE(num_features >= 3 && num_features <= kMaxStride,
                   "voxelize_batch3: num_features must be in [3, %d], got %d", kMaxStride, num_features);
    VoxParam P;
    P.n_levels = 3;
    for (int j = 0; j < 3; ++j) P.lo[j] = coors_range[j];
    int rc;
    if ((rc = fill_level(P.lv[0], voxel_size_top, coors_range))) return rc;
    if ((rc = fill_level(P.lv[1], voxel_size_med, coors_range))) return rc;
    if ((rc = fill_level(P.lv[2], voxel_size_low, coors_range))) return rc;
    const int grid = stream_grid(num_points, kVoxBlock);
    hipLaunchKernelGGL(voxelize_kernel<true>, dim3(grid), dim3(kVoxBlock), 0, stream, points, num_points,
                       num_features, batch_offsets, batch_size, P, coors_top, coors_med, coors_low);
    return check_launch("voxelize_kernel<batch3>");
}
