// DynamicScatter native op: the in-tree pybind functions
//   dynamic_point_to_voxel_forward(feats, coors, reduce_type) -> [reduced_feats, out_coors, coors_map, reduce_count]
//   dynamic_point_to_voxel_backward(grad_feats, grad_reduced_feats, feats, reduced_feats, coors_map, reduce_count, reduce_type)
// of mmdet3d/ops/voxel/src/voxelization.h:112-154 / scatter_points_cuda.cu:183-310, used by
// mmdet3d/ops/voxel/scatter_points.py:11-49 (DynamicScatter; DynamicVFE and the Waymo SST configs).
//
// Reference: at::unique_dim over the coordinates (a multi-pass device sort + host sync for the output size),
// then one thread per point doing C atomicCAS-max / atomicAdd into the voxel rows, a division kernel for the
// mean, and for the max backward an atomicMin pass + a scatter.  Here the grouping is the counting sort of
// segment.hip over the dense voxel grid (the caller knows the grid: DynamicScatter is constructed with
// voxel_size and point_cloud_range), rows come out in the same lexicographic order unique_dim(sorted=True)
// gives, and the reductions walk contiguous segments: no float atomics, sums accumulated in fp64 so that the
// result does not depend on the order of the points inside a voxel.
#include "common.h"
#include "../../include/geomae_hip.h"

namespace geomae {

enum { kSum = 0, kMean = 1, kMax = 2 };

// one wave per voxel, lane = channel (+64 per pass)
__global__ __launch_bounds__(256) void scatter_reduce_kernel(const float* __restrict__ feats, int C,
                                                             const int32_t* __restrict__ order,
                                                             const int32_t* __restrict__ seg_start,
                                                             const int32_t* __restrict__ num_voxels,
                                                             const int32_t* __restrict__ coors4, int ndim, int mode,
                                                             float* __restrict__ out, int32_t* __restrict__ out_coors,
                                                             int32_t* __restrict__ count) {
    const int V = num_voxels[0];
    const int lane = threadIdx.x & 63;
    for (int p = blockIdx.x * 4 + (threadIdx.x >> 6); p < V; p += gridDim.x * 4) {
        const int s = seg_start[p], e = seg_start[p + 1];
        for (int c = lane; c < C; c += 64) {
            float r;
            if (mode == kMax) {
                float best = -INFINITY;
                for (int j = s; j < e; ++j) best = fmaxf(best, feats[(int64_t)order[j] * C + c]);
                r = best;
            } else {
                double acc = 0.0;
                for (int j = s; j < e; ++j) acc += (double)feats[(int64_t)order[j] * C + c];
                r = mode == kMean ? (float)(acc / (double)(e - s)) : (float)acc;
            }
            out[(int64_t)p * C + c] = r;
        }
        if (lane == 0) count[p] = e - s;
        if (lane < ndim) out_coors[(int64_t)p * ndim + lane] = coors4[(int64_t)p * 4 + (4 - ndim) + lane];
    }
}

// sum / mean: grad_feats[i] = grad_reduced[map[i]] (/ count)
__global__ __launch_bounds__(256) void scatter_bwd_add_kernel(float* __restrict__ grad_feats,
                                                              const float* __restrict__ grad_reduced,
                                                              const int32_t* __restrict__ map,
                                                              const int32_t* __restrict__ count, int64_t n, int C,
                                                              int mode) {
    const int64_t total = n * C;
    for (int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t i = t / C;
        const int c = (int)(t - i * C);
        const int m = map[i];
        float g = 0.f;
        if (m >= 0) {
            g = grad_reduced[(int64_t)m * C + c];
            if (mode == kMean) g = g / (float)count[m];
        }
        grad_feats[t] = g;
    }
}

// max: the gradient of voxel (m, c) goes to the LOWEST-index point whose feature equals the maximum
// (scatter_points_cuda.cu:135-160: equality + atomicMin), every other point gets zero
__global__ __launch_bounds__(256) void scatter_bwd_argmin_kernel(const float* __restrict__ feats,
                                                                 const float* __restrict__ reduced,
                                                                 const int32_t* __restrict__ map, int64_t n, int C,
                                                                 int32_t* __restrict__ reduce_from) {
    const int64_t total = n * C;
    for (int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t i = t / C;
        const int c = (int)(t - i * C);
        const int m = map[i];
        if (m >= 0 && feats[t] == reduced[(int64_t)m * C + c]) atomicMin(&reduce_from[(int64_t)m * C + c], (int32_t)i);
    }
}
__global__ __launch_bounds__(256) void scatter_bwd_max_kernel(float* __restrict__ grad_feats,
                                                              const float* __restrict__ grad_reduced,
                                                              const int32_t* __restrict__ map,
                                                              const int32_t* __restrict__ reduce_from, int64_t n, int C) {
    const int64_t total = n * C;
    for (int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t i = t / C;
        const int c = (int)(t - i * C);
        const int m = map[i];
        float g = 0.f;
        if (m >= 0 && reduce_from[(int64_t)m * C + c] == (int32_t)i) g = grad_reduced[(int64_t)m * C + c];
        grad_feats[t] = g;
    }
}

struct ScatterWs { int64_t table, coors4, order, seg, sample, seg_ws, total; };
static ScatterWs scatter_ws(int64_t n, int cap, int nb, int gz, int gy, int gx) {
    auto al = [](int64_t b) { return (b + 255) / 256 * 256; };
    ScatterWs w;
    int64_t p = 0;
    w.table = p;  p += al((int64_t)nb * gz * gy * gx * 4);
    w.coors4 = p; p += al((int64_t)(cap > 0 ? cap : 1) * 16);
    w.order = p;  p += al((n > 0 ? n : 1) * 4);
    w.seg = p;    p += al(((int64_t)cap + 2) * 4);
    w.sample = p; p += al(((int64_t)nb + 1) * 4);
    w.seg_ws = p; p += geomae_pillar_segment_workspace_bytes(n, nb, gz, gy, gx);
    w.total = p;
    return w;
}

}  // namespace geomae

using namespace geomae;

extern "C" int64_t geomae_dynamic_point_to_voxel_workspace_bytes(int64_t num_points, int32_t max_voxels, int32_t batch_size,
                                                                 int32_t gz, int32_t gy, int32_t gx) {
    return scatter_ws(num_points, max_voxels, batch_size, gz, gy, gx).total;
}

extern "C" int geomae_dynamic_point_to_voxel_forward(const float* feats, const int32_t* coors, int64_t num_points,
                                                     int32_t channels, int32_t ndim, int32_t batch_size, int32_t gz,
                                                     int32_t gy, int32_t gx, int32_t reduce_type, int32_t max_voxels,
                                                     float* reduced_feats, int32_t* out_coors, int32_t* coors_map,
                                                     int32_t* reduce_count, int32_t* num_voxels, void* workspace,
                                                     int64_t workspace_bytes, hipStream_t stream) {
    GEOMAE_REQUIRE(reduce_type >= 0 && reduce_type <= 2, "dynamic_point_to_voxel_forward: reduce_type is 0 sum, 1 mean, 2 max");
    GEOMAE_REQUIRE(channels >= 1 && max_voxels >= 0 && num_voxels, "dynamic_point_to_voxel_forward: bad argument");
    const ScatterWs w = scatter_ws(num_points, max_voxels, batch_size, gz, gy, gx);
    if (workspace_bytes < w.total || !workspace) {
        set_error("dynamic_point_to_voxel_forward: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)w.total);
        return GEOMAE_ERR_WORKSPACE;
    }
    char* ws = (char*)workspace;
    int32_t *table = (int32_t*)(ws + w.table), *coors4 = (int32_t*)(ws + w.coors4), *order = (int32_t*)(ws + w.order),
            *seg = (int32_t*)(ws + w.seg), *sample = (int32_t*)(ws + w.sample);
    int rc = geomae_pillar_segment_nd(coors, ndim, num_points, batch_size, gz, gy, gx, table, coors4, coors_map, order, seg,
                                      sample, num_voxels, ws + w.seg_ws, w.total - w.seg_ws, stream);
    if (rc) return rc;
    if (max_voxels == 0 || num_points == 0) return GEOMAE_OK;
    GEOMAE_REQUIRE(feats && reduced_feats && out_coors && reduce_count, "dynamic_point_to_voxel_forward: null argument");
    const int grid = max_voxels / 4 + 1 < 256 * 16 ? max_voxels / 4 + 1 : 256 * 16;
    hipLaunchKernelGGL(scatter_reduce_kernel, dim3(grid), dim3(256), 0, stream, feats, channels, order, seg, num_voxels,
                       coors4, ndim, reduce_type, reduced_feats, out_coors, reduce_count);
    return check_launch("scatter_reduce_kernel");
}

extern "C" int geomae_dynamic_point_to_voxel_backward(float* grad_feats, const float* grad_reduced_feats, const float* feats,
                                                      const float* reduced_feats, const int32_t* coors_map,
                                                      const int32_t* reduce_count, int64_t num_points, int32_t num_voxels,
                                                      int32_t channels, int32_t reduce_type, int32_t* reduce_from_ws,
                                                      hipStream_t stream) {
    GEOMAE_REQUIRE(reduce_type >= 0 && reduce_type <= 2, "dynamic_point_to_voxel_backward: reduce_type is 0 sum, 1 mean, 2 max");
    if (num_points <= 0) return GEOMAE_OK;
    GEOMAE_REQUIRE(grad_feats && channels >= 1, "dynamic_point_to_voxel_backward: bad argument");
    const int64_t total = num_points * channels;
    if (num_voxels <= 0) {                    // grad_feats.fill_(0) and return (scatter_points_cuda.cu:262-265)
        GEOMAE_HIP(hipMemsetAsync(grad_feats, 0, total * sizeof(float), stream));
        return GEOMAE_OK;
    }
    GEOMAE_REQUIRE(grad_reduced_feats && coors_map && reduce_count, "dynamic_point_to_voxel_backward: null argument");
    const int grid = stream_grid(total, 256);
    if (reduce_type != kMax) {
        hipLaunchKernelGGL(scatter_bwd_add_kernel, dim3(grid), dim3(256), 0, stream, grad_feats, grad_reduced_feats, coors_map,
                           reduce_count, num_points, channels, reduce_type);
        return check_launch("scatter_bwd_add_kernel");
    }
    GEOMAE_REQUIRE(feats && reduced_feats && reduce_from_ws, "dynamic_point_to_voxel_backward: max needs feats, reduced_feats, workspace");
    GEOMAE_HIP(hipMemsetAsync(reduce_from_ws, 0x7f, (size_t)num_voxels * channels * sizeof(int32_t), stream));
    hipLaunchKernelGGL(scatter_bwd_argmin_kernel, dim3(grid), dim3(256), 0, stream, feats, reduced_feats, coors_map, num_points,
                       channels, reduce_from_ws);
    hipLaunchKernelGGL(scatter_bwd_max_kernel, dim3(grid), dim3(256), 0, stream, grad_feats, grad_reduced_feats, coors_map,
                       reduce_from_ws, num_points, channels);
    return check_launch("scatter_bwd_max_kernel");
}
