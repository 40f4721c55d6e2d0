// Hard voxelization: voxel_layer.hard_voxelize of mmdet3d/ops/voxel/src/voxelization.h:58-76,
// semantics of voxelization_cpu.cpp:42-100 (the CPU path is the one that compiles here; oracle/_ref):
//   grid = round((max - min) / voxel_size)                      (ROUND, not the dynamic path's ceil, :113-116)
//   c = clamp(floor((p - min) / voxel_size), 0, grid - 1)       (this fork clamps instead of dropping, :22-31)
//   walk the points in index order: a voxel is created when its cell is first seen, at most max_voxels
//   voxels (points of later cells are dropped); a voxel keeps its first max_points points, in index order.
//   -> voxels [max_voxels, max_points, C] (zero padded), coors [max_voxels, 3] (z,y,x), num_points_per_voxel,
//      return value = number of voxels.
// The reference's device version does this with an O(N^2) point_to_voxelidx kernel and a <<<1,1>>> serial
// kernel (voxelization_cuda.cu:208-350).  Here: the counting sort of segment.hip groups the points by cell,
// each cell's point list is put in index order, "first seen" order of the cells is an exclusive scan of the
// is-first-point-of-its-cell flags over the point index, and one wave per cell copies its rows.
#include "common.h"
#include "../../include/geomae_hip.h"

namespace geomae {

__global__ __launch_bounds__(256) void hv_coords_kernel(const float* __restrict__ pts, int64_t n, int C, float vx, float vy,
                                                        float vz, float x0, float y0, float z0, int gx, int gy, int gz,
                                                        int4* __restrict__ coors4) {
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float* p = pts + i * C;
        int cx = (int)floorf(__fdiv_rn(__fsub_rn(p[0], x0), vx));
        int cy = (int)floorf(__fdiv_rn(__fsub_rn(p[1], y0), vy));
        int cz = (int)floorf(__fdiv_rn(__fsub_rn(p[2], z0), vz));
        cx = cx < 0 ? 0 : (cx >= gx ? gx - 1 : cx);
        cy = cy < 0 ? 0 : (cy >= gy ? gy - 1 : cy);
        cz = cz < 0 ? 0 : (cz >= gz ? gz - 1 : cz);
        coors4[i] = make_int4(0, cz, cy, cx);
    }
}

// every cell's point list in ascending point index (rank sort, one wave per cell)
__global__ __launch_bounds__(64) void hv_seg_sort_kernel(const int32_t* __restrict__ seg_start,
                                                         const int32_t* __restrict__ num_cells,
                                                         const int32_t* __restrict__ order_in, int32_t* __restrict__ order_out) {
    __shared__ int a[2048];
    const int V = num_cells[0];
    for (int p = blockIdx.x; p < V; p += gridDim.x) {
        const int s = seg_start[p], n = seg_start[p + 1] - s;
        if (n <= 2048) {
            for (int t = threadIdx.x; t < n; t += 64) a[t] = order_in[s + t];
            __syncthreads();
            for (int t = threadIdx.x; t < n; t += 64) {
                const int v = a[t];
                int r = 0;
                for (int u = 0; u < n; ++u) r += a[u] < v;
                order_out[s + r] = v;
            }
            __syncthreads();
        } else {                                    // a very dense cell: same rank sort against global memory
            for (int t = threadIdx.x; t < n; t += 64) {
                const int v = order_in[s + t];
                int r = 0;
                for (int u = 0; u < n; ++u) r += order_in[s + u] < v;
                order_out[s + r] = v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void hv_flag_kernel(const int32_t* __restrict__ inv, const int32_t* __restrict__ seg_start,
                                                      const int32_t* __restrict__ order, int64_t n, int32_t* __restrict__ flag) {
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        flag[i] = order[seg_start[inv[i]]] == (int32_t)i;
}

// exclusive scan of n ints, three kernels (tile = 1024)
__device__ __forceinline__ int hv_block_scan(int v, int* total, int* sm) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) sm[w] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int s = sm[k];
        if (k < w) woff += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return woff + incl - v;
}
__global__ __launch_bounds__(256) void hv_scan_reduce_kernel(const int32_t* __restrict__ v, int64_t n, int32_t* __restrict__ tile_sum) {
    __shared__ int sm[4];
    const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;
    int s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += base + k < n ? v[base + k] : 0;
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}
__global__ __launch_bounds__(256) void hv_scan_tiles_kernel(int32_t* __restrict__ tile_sum, int n_tiles) {
    __shared__ int sm[4];
    int run = 0;
    for (int base = 0; base < n_tiles; base += 256) {
        const int t = base + threadIdx.x;
        const int v = t < n_tiles ? tile_sum[t] : 0;
        int tot;
        const int e = hv_block_scan(v, &tot, sm);
        if (t < n_tiles) tile_sum[t] = run + e;
        run += tot;
    }
}
__global__ __launch_bounds__(256) void hv_scan_emit_kernel(int32_t* __restrict__ v, int64_t n, const int32_t* __restrict__ tile_sum) {
    __shared__ int sm[4];
    const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;
    int x[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { x[k] = base + k < n ? v[base + k] : 0; s += x[k]; }
    int tot;
    int e = hv_block_scan(s, &tot, sm) + tile_sum[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (base + k < n) v[base + k] = e;
        e += x[k];
    }
}

// one wave per cell: voxel id = number of cells whose first point comes earlier
__global__ __launch_bounds__(256) void hv_fill_kernel(const float* __restrict__ pts, int C, const int32_t* __restrict__ seg_start,
                                                      const int32_t* __restrict__ num_cells, const int32_t* __restrict__ order,
                                                      const int32_t* __restrict__ pos, const int4* __restrict__ cell_coors,
                                                      int max_points, int max_voxels, float* __restrict__ voxels,
                                                      int32_t* __restrict__ coors, int32_t* __restrict__ num_points,
                                                      int32_t* __restrict__ voxel_num) {
    const int V = num_cells[0];
    const int lane = threadIdx.x & 63;
    if (blockIdx.x == 0 && threadIdx.x == 0) voxel_num[0] = V < max_voxels ? V : max_voxels;
    for (int p = blockIdx.x * 4 + (threadIdx.x >> 6); p < V; p += gridDim.x * 4) {
        const int s = seg_start[p], cnt = seg_start[p + 1] - s;
        const int vid = pos[order[s]];
        if (vid >= max_voxels) continue;
        const int keep = cnt < max_points ? cnt : max_points;
        if (lane == 0) {
            const int4 c = cell_coors[p];
            coors[vid * 3 + 0] = c.y; coors[vid * 3 + 1] = c.z; coors[vid * 3 + 2] = c.w;
            num_points[vid] = keep;
        }
        float* dst = voxels + (int64_t)vid * max_points * C;
        for (int e = lane; e < keep * C; e += 64) {
            const int r = e / C, c = e - r * C;
            dst[e] = pts[(int64_t)order[s + r] * C + c];
        }
    }
}

// in-place exclusive scan of n int32 (tile_ws: n / 1024 + 2 ints); shared with points_pipeline.hip
int exclusive_scan_i32(int32_t* values, int64_t n, int32_t* tile_ws, hipStream_t stream) {
    if (n <= 0) return GEOMAE_OK;
    const int n_tiles = cdiv(n, 1024);
    hipLaunchKernelGGL(hv_scan_reduce_kernel, dim3(n_tiles), dim3(256), 0, stream, values, n, tile_ws);
    hipLaunchKernelGGL(hv_scan_tiles_kernel, dim3(1), dim3(256), 0, stream, tile_ws, n_tiles);
    hipLaunchKernelGGL(hv_scan_emit_kernel, dim3(n_tiles), dim3(256), 0, stream, values, n, tile_ws);
    return check_launch("exclusive_scan_i32");
}

struct HvWs { int64_t coors4, table, cell_coors, inv, order0, order1, seg, sample, nv, flag, tiles, seg_ws, total; };
static HvWs hv_ws(int64_t n, int gz, int gy, int gx) {
    auto al = [](int64_t b) { return (b + 255) / 256 * 256; };
    const int64_t cells = (int64_t)gz * gy * gx, cap = n < cells ? n : cells, n1 = n > 0 ? n : 1;
    HvWs w;
    int64_t p = 0;
    w.coors4 = p;     p += al(n1 * 16);
    w.table = p;      p += al(cells * 4);
    w.cell_coors = p; p += al((cap + 1) * 16);
    w.inv = p;        p += al(n1 * 4);
    w.order0 = p;     p += al(n1 * 4);
    w.order1 = p;     p += al(n1 * 4);
    w.seg = p;        p += al((cap + 2) * 4);
    w.sample = p;     p += al(2 * 4);
    w.nv = p;         p += al(4);
    w.flag = p;       p += al(n1 * 4);
    w.tiles = p;      p += al((n1 / 1024 + 2) * 4);
    w.seg_ws = p;     p += geomae_pillar_segment_workspace_bytes(n, 1, gz, gy, gx);
    w.total = p;
    return w;
}
static void hv_grid(const float* vs, const float* r, int* g) {
    for (int i = 0; i < 3; ++i) g[i] = (int)roundf((r[3 + i] - r[i]) / vs[i]);      // voxelization_cpu.cpp:113-116
}

}  // namespace geomae

using namespace geomae;

extern "C" int64_t geomae_hard_voxelize_workspace_bytes(int64_t num_points, const float* voxel_size, const float* coors_range) {
    if (!voxel_size || !coors_range) return -1;
    int g[3];
    hv_grid(voxel_size, coors_range, g);
    if (g[0] < 1 || g[1] < 1 || g[2] < 1) return -1;
    return hv_ws(num_points, g[2], g[1], g[0]).total;
}

extern "C" int geomae_hard_voxelize(const float* points, int64_t num_points, int32_t num_features, const float* voxel_size,
                                    const float* coors_range, int32_t max_points, int32_t max_voxels, float* voxels,
                                    int32_t* coors, int32_t* num_points_per_voxel, int32_t* voxel_num, void* workspace,
                                    int64_t workspace_bytes, hipStream_t stream) {
    GEOMAE_REQUIRE(voxel_size && coors_range && voxels && coors && num_points_per_voxel && voxel_num, "hard_voxelize: null argument");
    GEOMAE_REQUIRE(num_points >= 0 && num_features >= 3 && max_points >= 1 && max_voxels >= 1,
                   "hard_voxelize: needs num_features >= 3, max_points >= 1, max_voxels >= 1 (-1 selects dynamic voxelization)");
    int g[3];
    hv_grid(voxel_size, coors_range, g);
    GEOMAE_REQUIRE(g[0] >= 1 && g[1] >= 1 && g[2] >= 1, "hard_voxelize: empty grid");
    const int gx = g[0], gy = g[1], gz = g[2];
    GEOMAE_REQUIRE((int64_t)gx * gy * gz < ((int64_t)1 << 31), "hard_voxelize: grid too large");
    const HvWs w = hv_ws(num_points, gz, gy, gx);
    if (workspace_bytes < w.total || !workspace) {
        set_error("hard_voxelize: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)w.total);
        return GEOMAE_ERR_WORKSPACE;
    }
    GEOMAE_HIP(hipMemsetAsync(voxels, 0, (size_t)max_voxels * max_points * num_features * sizeof(float), stream));
    GEOMAE_HIP(hipMemsetAsync(coors, 0, (size_t)max_voxels * 3 * sizeof(int32_t), stream));
    GEOMAE_HIP(hipMemsetAsync(num_points_per_voxel, 0, (size_t)max_voxels * sizeof(int32_t), stream));
    GEOMAE_HIP(hipMemsetAsync(voxel_num, 0, sizeof(int32_t), stream));
    if (num_points == 0) return GEOMAE_OK;
    GEOMAE_REQUIRE(points, "hard_voxelize: null points");
    char* ws = (char*)workspace;
    int4* coors4 = (int4*)(ws + w.coors4);
    int32_t *table = (int32_t*)(ws + w.table), *inv = (int32_t*)(ws + w.inv), *order0 = (int32_t*)(ws + w.order0),
            *order1 = (int32_t*)(ws + w.order1), *seg = (int32_t*)(ws + w.seg), *sample = (int32_t*)(ws + w.sample),
            *nv = (int32_t*)(ws + w.nv), *flag = (int32_t*)(ws + w.flag), *tiles = (int32_t*)(ws + w.tiles);
    int4* cell_coors = (int4*)(ws + w.cell_coors);
    hipLaunchKernelGGL(hv_coords_kernel, dim3(stream_grid(num_points, 256)), dim3(256), 0, stream, points, num_points,
                       num_features, voxel_size[0], voxel_size[1], voxel_size[2], coors_range[0], coors_range[1], coors_range[2],
                       gx, gy, gz, coors4);
    int rc = geomae_pillar_segment_nd((const int32_t*)coors4, 4, num_points, 1, gz, gy, gx, table, (int32_t*)cell_coors, inv,
                                      order0, seg, sample, nv, ws + w.seg_ws, w.total - w.seg_ws, stream);
    if (rc) return rc;
    const int64_t cells = (int64_t)gz * gy * gx;
    const int cap = (int)(num_points < cells ? num_points : cells);
    hipLaunchKernelGGL(hv_seg_sort_kernel, dim3(cap < 8192 ? cap : 8192), dim3(64), 0, stream, seg, nv, order0, order1);
    hipLaunchKernelGGL(hv_flag_kernel, dim3(stream_grid(num_points, 256)), dim3(256), 0, stream, inv, seg, order1, num_points, flag);
    if ((rc = exclusive_scan_i32(flag, num_points, tiles, stream))) return rc;
    hipLaunchKernelGGL(hv_fill_kernel, dim3(cap / 4 + 1 < 4096 ? cap / 4 + 1 : 4096), dim3(256), 0, stream, points, num_features,
                       seg, nv, order1, flag, cell_coors, max_points, max_voxels, voxels, coors, num_points_per_voxel, voxel_num);
    return check_launch("hard_voxelize");
}
