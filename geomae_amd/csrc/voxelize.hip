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

// ---- the step's entry: B frames -> concatenated [N, stride] point rows + the three coordinate arrays, in ONE launch.
// Replaces the reference's torch.cat of the per-sample tensors (ssl.py:320-329; B device-to-device copies + one
// host-to-device copy of the batch offsets in round 2) followed by geomae_voxelize_batch3, and clears up to two
// buffers on the side (the pillar table and the scan state of geomae_pillar_segment_ex: the memset kernels that used
// to sit between the dependent kernels of stage 1).  Frame pointers and offsets travel as kernel arguments.
constexpr int kMaxFrames = 32;
struct FrameTable {
    const float* ptr[kMaxFrames];
    long long off[kMaxFrames + 1];      // row offsets of the frames in the concatenated order
    int n;
};
struct ZeroJobs { uint4* ptr[2]; long long n16[2]; };     // 16-byte units

__global__ __launch_bounds__(kVoxBlock) void voxelize_frames_kernel(FrameTable F, int64_t n, int stride, VoxParam P,
                                                                    float* __restrict__ pts_out,
                                                                    int32_t* __restrict__ offs_out,
                                                                    int32_t* __restrict__ out0, int32_t* __restrict__ out1,
                                                                    int32_t* __restrict__ out2, ZeroJobs Z) {
    __shared__ float s[kVoxBlock * kMaxStride];
    if (blockIdx.x == 0 && threadIdx.x <= F.n && offs_out) offs_out[threadIdx.x] = (int32_t)F.off[threadIdx.x];
    const int64_t n_tiles = (n + kVoxBlock - 1) / kVoxBlock;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t base = tile * kVoxBlock;
        const int rows = (int)((n - base) < kVoxBlock ? (n - base) : kVoxBlock);
        const int words = rows * stride;
        // frame of the tile's first row (scalar search); a tile that straddles frames takes the per-word path
        int b0 = 0;
        for (int k = 1; k < F.n; ++k) b0 += (base >= F.off[k]) ? 1 : 0;
        const bool one_frame = base + rows <= F.off[b0 + 1];
        if (one_frame) {
            const float* src = F.ptr[b0] + (base - F.off[b0]) * stride;
            for (int w = threadIdx.x; w < words; w += kVoxBlock) s[w] = src[w];
        } else {
            for (int w = threadIdx.x; w < words; w += kVoxBlock) {
                const int64_t row = base + w / stride;
                int b = b0;
                while (row >= F.off[b + 1]) ++b;
                s[w] = F.ptr[b][(row - F.off[b]) * stride + w % stride];
            }
        }
        __syncthreads();
        float* dst = pts_out + base * stride;
        for (int w = threadIdx.x; w < words; w += kVoxBlock) dst[w] = s[w];
        if ((int)threadIdx.x < rows) {
            const int64_t i = base + threadIdx.x;
            const float x = s[threadIdx.x * stride + 0];
            const float y = s[threadIdx.x * stride + 1];
            const float z = s[threadIdx.x * stride + 2];
            int b = b0;
            while (i >= F.off[b + 1]) ++b;
            int32_t* outs[3] = {out0, out1, out2};
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                const LevelParam& L = P.lv[l];
                const int cx = cell(x, P.lo[0], L.vs[0], L.grid[0]);
                const int cy = cell(y, P.lo[1], L.vs[1], L.grid[1]);
                const int cz = cell(z, P.lo[2], L.vs[2], L.grid[2]);
                reinterpret_cast<int4*>(outs[l])[i] = make_int4(b, cz, cy, cx);
            }
        }
        __syncthreads();
    }
    const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int q = 0; q < 2; ++q)
        for (int64_t e = (int64_t)blockIdx.x * kVoxBlock + threadIdx.x; e < Z.n16[q]; e += (int64_t)gridDim.x * kVoxBlock)
            Z.ptr[q][e] = z4;
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

extern "C" int geomae_voxelize_frames3(const float* const* frame_points, const int64_t* frame_sizes, int32_t batch_size,
                                       int32_t num_features, const float* voxel_size_top, const float* voxel_size_med,
                                       const float* voxel_size_low, const float* coors_range, float* points_out,
                                       int32_t* batch_offsets_out, int32_t* coors_top, int32_t* coors_med,
                                       int32_t* coors_low, void* zero_a, int64_t zero_a_bytes, void* zero_b,
                                       int64_t zero_b_bytes, hipStream_t stream) {
    GEOMAE_REQUIRE(frame_points && frame_sizes && batch_size >= 1 && batch_size <= kMaxFrames,
                   "voxelize_frames3: 1..%d frames", kMaxFrames);
    GEOMAE_REQUIRE(num_features >= 3 && num_features <= kMaxStride,
                   "voxelize_frames3: num_features must be in [3, %d], got %d", kMaxStride, num_features);
    GEOMAE_REQUIRE(points_out && coors_top && coors_med && coors_low, "voxelize_frames3: null output");
    GEOMAE_REQUIRE(zero_a_bytes >= 0 && zero_b_bytes >= 0 && zero_a_bytes % 16 == 0 && zero_b_bytes % 16 == 0 &&
                   (zero_a || zero_a_bytes == 0) && (zero_b || zero_b_bytes == 0) &&
                   ((uintptr_t)zero_a % 16) == 0 && ((uintptr_t)zero_b % 16) == 0,
                   "voxelize_frames3: zero ranges must be 16-byte aligned multiples of 16 bytes");
    FrameTable F;
    F.n = batch_size;
    int64_t N = 0;
    for (int i = 0; i < kMaxFrames; ++i) { F.ptr[i] = nullptr; F.off[i + 1] = 0; }
    F.off[0] = 0;
    for (int i = 0; i < batch_size; ++i) {
        GEOMAE_REQUIRE(frame_sizes[i] >= 0 && (frame_points[i] || frame_sizes[i] == 0), "voxelize_frames3: bad frame %d", i);
        F.ptr[i] = frame_points[i];
        N += frame_sizes[i];
        F.off[i + 1] = N;
    }
    for (int i = batch_size; i < kMaxFrames; ++i) F.off[i + 1] = N;
    GEOMAE_REQUIRE(N < (int64_t)1 << 31, "voxelize_frames3: too many points");
    VoxParam P;
    P.n_levels = 3;
    for (int j = 0; j < 3; ++j) P.lo[j] = coors_range[j];
    int rc;
    if ((rc = fill_level(P.lv[0], voxel_size_top, coors_range))) return rc;
    if ((rc = fill_level(P.lv[1], voxel_size_med, coors_range))) return rc;
    if ((rc = fill_level(P.lv[2], voxel_size_low, coors_range))) return rc;
    ZeroJobs Z;
    Z.ptr[0] = (uint4*)zero_a; Z.n16[0] = zero_a_bytes / 16;
    Z.ptr[1] = (uint4*)zero_b; Z.n16[1] = zero_b_bytes / 16;
    const int64_t work = N > (zero_a_bytes + zero_b_bytes) / 16 ? N : (zero_a_bytes + zero_b_bytes) / 16;
    if (work == 0) return GEOMAE_OK;
    const int grid = stream_grid(work, kVoxBlock);
    hipLaunchKernelGGL(voxelize_frames_kernel, dim3(grid), dim3(kVoxBlock), 0, stream, F, N, num_features, P, points_out,
                       batch_offsets_out, coors_top, coors_med, coors_low, Z);
    return check_launch("voxelize_frames_kernel");
}
