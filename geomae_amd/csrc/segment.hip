// Pillar segment build + segmented reductions (gfx950).
//
// Replaces the reference's six torch.unique(coors, dim=0) sorts per iteration
// (mmdet3d/ops/sst/sst_ops.py:8-39 scatter_v2, called 3x by DynamicScatterVFE
// voxel_encoder.py:375,407; detectors/multi_sub_voxel_dynamic_voxelnet_ssl.py:749 called 3x)
// and torch_scatter's mean/max by ONE counting sort per batch over the dense cell table
// (B*gz*gy*gx int32, 2.5 MB for the nuScenes config):
//   1. hist   : rank_in_cell[i] = atomicAdd(table[key_i], 1)          (1 atomic / point)
//   2. scan   : exclusive scans of (occupied, count) over cells -> pillar id, segment start;
//               pillar ids increase with the cell key, i.e. exactly the lexicographic
//               (b, z, y, x) order torch.unique(dim=0) returns; table[cell] <- pillar id | -1
//               (kept: it is the neighbour / parent lookup table of the target kernels)
//   3. place  : inv[i] = table[key_i];  order[start[inv[i]] + rank_in_cell[i]] = i
// Segmented reductions then walk contiguous point ranges (no float atomics):
//   mean  -> 2^-32 fixed-point int64 sums: order-independent, deterministic
//   max   -> exact in any order; arg-max kept for the backward pass
#include "common.h"
#include "../../include/geomae_hip.h"

namespace geomae {

constexpr int kBlk = 256;
constexpr int kItems = 4;                 // cells per thread in the scan kernels
constexpr int kTile = kBlk * kItems;      // 1024 cells per workgroup

__device__ __forceinline__ int64_t key_of(const int4 c, int gz, int gy, int gx) {
    return (((int64_t)c.x * gz + c.y) * gy + c.z) * gx + c.w;
}

// row i of coors [n, ndim]: (b, z, y, x) for ndim 4, (z, y, x) with b = 0 for ndim 3.  A row with a negative or
// out-of-grid coordinate is invalid (vanilla dynamic voxelization marks out-of-range points with -1,
// voxelization_cuda.cu:35-57 upstream; scatter_points_cuda.cu:199 drops them): key -1.
__device__ __forceinline__ int64_t row_key(const int32_t* __restrict__ coors, int64_t i, int ndim, int nb, int gz, int gy,
                                           int gx) {
    int4 c;
    if (ndim == 4) c = reinterpret_cast<const int4*>(coors)[i];
    else c = make_int4(0, coors[i * 3 + 0], coors[i * 3 + 1], coors[i * 3 + 2]);
    const bool ok = c.x >= 0 && c.x < nb && c.y >= 0 && c.y < gz && c.z >= 0 && c.z < gy && c.w >= 0 && c.w < gx;
    return ok ? key_of(c, gz, gy, gx) : -1;
}

__global__ __launch_bounds__(kBlk) void hist_kernel(const int32_t* __restrict__ coors, int64_t n, int ndim, int nb, int gz,
                                                    int gy, int gx, int32_t* __restrict__ table,
                                                    int32_t* __restrict__ rank_in_cell) {
    for (int64_t i = blockIdx.x * (int64_t)kBlk + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlk) {
        const int64_t k = row_key(coors, i, ndim, nb, gz, gy, gx);
        rank_in_cell[i] = k >= 0 ? atomicAdd(&table[k], 1) : -1;
    }
}

// block-wide exclusive scan of one int per thread (256 threads = 4 waves)
__device__ __forceinline__ int block_excl_scan(int v, int* total, int* smem /*>= 8 ints*/) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) smem[w] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < kBlk / 64; ++k) {
        int s = smem[k];
        if (k < w) woff += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return woff + incl - v;
}

// pass 1: per-tile totals of (occupied cells, points)
__global__ __launch_bounds__(kBlk) void scan_reduce_kernel(const int32_t* __restrict__ table, int64_t cells,
                                                           int2* __restrict__ tile_sums) {
    __shared__ int sm[2 * (kBlk / 64)];
    const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kItems;
    int occ = 0, cnt = 0;
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const int64_t c = base + k;
        const int v = c < cells ? table[c] : 0;
        occ += v > 0;
        cnt += v;
    }
    occ = wave_sum(occ);
    cnt = wave_sum(cnt);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sm[w] = occ; sm[kBlk / 64 + w] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int o = 0, c = 0;
        for (int k = 0; k < kBlk / 64; ++k) { o += sm[k]; c += sm[kBlk / 64 + k]; }
        tile_sums[blockIdx.x] = make_int2(o, c);
    }
}

// pass 2: one workgroup scans the tile totals in place (exclusive) and publishes V
__global__ __launch_bounds__(kBlk) void scan_tiles_kernel(int2* __restrict__ tile_sums, int n_tiles,
                                                          int32_t* __restrict__ num_pillars) {
    __shared__ int sm[8];
    int run_o = 0, run_c = 0;
    for (int base = 0; base < n_tiles; base += kBlk) {
        const int t = base + threadIdx.x;
        int2 v = t < n_tiles ? tile_sums[t] : make_int2(0, 0);
        int tot_o, tot_c;
        const int eo = block_excl_scan(v.x, &tot_o, sm);
        const int ec = block_excl_scan(v.y, &tot_c, sm);
        if (t < n_tiles) tile_sums[t] = make_int2(run_o + eo, run_c + ec);
        run_o += tot_o;
        run_c += tot_c;
    }
    if (threadIdx.x == 0) {
        num_pillars[0] = run_o;
        tile_sums[n_tiles] = make_int2(run_o, run_c);      // totals: pillars, valid points
    }
}

// pass 3: per-cell pillar id / segment start; emit pillar coordinates; table <- pillar id | -1
__global__ __launch_bounds__(kBlk) void scan_emit_kernel(int32_t* __restrict__ table, int64_t cells,
                                                         const int2* __restrict__ tile_sums, int gz, int gy,
                                                         int gx, int n_batch, int32_t* __restrict__ voxel_coors,
                                                         int32_t* __restrict__ seg_start,
                                                         int32_t* __restrict__ sample_start, int64_t n_points,
                                                         const int32_t* __restrict__ num_pillars) {
    __shared__ int sm[8];
    const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kItems;
    int v[kItems];
    int occ = 0, cnt = 0;
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const int64_t c = base + k;
        v[k] = c < cells ? table[c] : 0;
        occ += v[k] > 0;
        cnt += v[k];
    }
    int tot;
    int eo = block_excl_scan(occ, &tot, sm);
    int ec = block_excl_scan(cnt, &tot, sm);
    const int2 off = tile_sums[blockIdx.x];
    eo += off.x;
    ec += off.y;
    const int64_t cells_per_sample = (int64_t)gz * gy * gx;
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const int64_t c = base + k;
        if (c >= cells) break;
        if (c % cells_per_sample == 0) sample_start[c / cells_per_sample] = eo;
        if (v[k] > 0) {
            int64_t r = c;
            const int x = (int)(r % gx); r /= gx;
            const int y = (int)(r % gy); r /= gy;
            const int z = (int)(r % gz); r /= gz;
            reinterpret_cast<int4*>(voxel_coors)[eo] = make_int4((int)r, z, y, x);
            seg_start[eo] = ec;
            table[c] = eo;
            ++eo;
            ec += v[k];
        } else {
            table[c] = -1;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const int V = num_pillars[0];
        sample_start[n_batch] = V;
        seg_start[V] = tile_sums[gridDim.x].y;             // number of valid points (= n_points when none is dropped)
    }
}

__global__ __launch_bounds__(kBlk) void place_kernel(const int32_t* __restrict__ coors, int64_t n, int ndim, int nb,
                                                     int gz, int gy, int gx, const int32_t* __restrict__ table,
                                                     const int32_t* __restrict__ rank_in_cell,
                                                     const int32_t* __restrict__ seg_start,
                                                     int32_t* __restrict__ inv, int32_t* __restrict__ order) {
    for (int64_t i = blockIdx.x * (int64_t)kBlk + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlk) {
        const int r = rank_in_cell[i];
        if (r < 0) { inv[i] = -1; continue; }
        const int p = table[row_key(coors, i, ndim, nb, gz, gy, gx)];
        inv[i] = p;
        order[seg_start[p] + r] = (int32_t)i;
    }
}

// ---------------------------------------------------------------- segmented max (+argmax)
// feat rows are given in ORIGINAL point order and read through `order`; one workgroup of C
// lanes walks one pillar, every row read is a coalesced C*4-byte line.
__global__ void seg_max_fwd_kernel(const float* __restrict__ feat, int C, const int32_t* __restrict__ order,
                                   const int32_t* __restrict__ seg_start,
                                   const int32_t* __restrict__ num_pillars, float* __restrict__ out,
                                   int32_t* __restrict__ argmax) {
    const int V = num_pillars[0];
    const int c = threadIdx.x;
    for (int p = blockIdx.x; p < V; p += gridDim.x) {
        const int s = seg_start[p], e = seg_start[p + 1];
        float best = -INFINITY;
        int arg = -1;
        for (int j = s; j < e; ++j) {
            const int i = order[j];
            const float v = feat[(int64_t)i * C + c];
            // ties: lowest point index wins (torch_scatter.scatter_max keeps one arg per cell)
            if (v > best || (v == best && i < arg) || arg < 0) { best = v; arg = i; }
        }
        out[(int64_t)p * C + c] = best;
        argmax[(int64_t)p * C + c] = arg;
    }
}

// grad_feat[i, c] = grad_out[inv[i], c] if argmax[inv[i], c] == i else 0   (dense, coalesced)
__global__ void seg_max_bwd_kernel(const float* __restrict__ grad_out, const int32_t* __restrict__ argmax,
                                   const int32_t* __restrict__ inv, int64_t n, int C,
                                   float* __restrict__ grad_feat) {
    const int64_t total = n * C;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / C;
        const int c = (int)(t - i * C);
        const int64_t o = (int64_t)inv[i] * C + c;
        grad_feat[t] = (argmax[o] == (int32_t)i) ? grad_out[o] : 0.0f;
    }
}

}  // namespace geomae

using namespace geomae;

extern "C" int64_t geomae_pillar_segment_workspace_bytes(int64_t num_points, int32_t batch_size, int32_t gz,
                                                         int32_t gy, int32_t gx) {
    const int64_t cells = (int64_t)batch_size * gz * gy * gx;
    const int64_t tiles = (cells + kTile - 1) / kTile;
    // rank_in_cell [n] int32 + tile sums [tiles + 1] int2, 256-byte aligned pieces
    auto al = [](int64_t b) { return (b + 255) / 256 * 256; };
    return al(num_points * 4) + al((tiles + 1) * 8);
}

extern "C" int geomae_pillar_segment(const int32_t* coors, int64_t num_points, int32_t batch_size, int32_t gz,
                                     int32_t gy, int32_t gx, int32_t* cell_table, int32_t* voxel_coors,
                                     int32_t* inv, int32_t* order, int32_t* seg_start, int32_t* sample_start,
                                     int32_t* num_pillars, void* workspace, int64_t workspace_bytes,
                                     hipStream_t stream) {
    return geomae_pillar_segment_nd(coors, 4, num_points, batch_size, gz, gy, gx, cell_table, voxel_coors, inv, order,
                                    seg_start, sample_start, num_pillars, workspace, workspace_bytes, stream);
}

extern "C" int geomae_pillar_segment_nd(const int32_t* coors, int32_t ndim, int64_t num_points, int32_t batch_size,
                                        int32_t gz, int32_t gy, int32_t gx, int32_t* cell_table, int32_t* voxel_coors,
                                        int32_t* inv, int32_t* order, int32_t* seg_start, int32_t* sample_start,
                                        int32_t* num_pillars, void* workspace, int64_t workspace_bytes,
                                        hipStream_t stream) {
    GEOMAE_REQUIRE(ndim == 4 || (ndim == 3 && batch_size == 1), "pillar_segment: coors are [N,4] (b,z,y,x) or [N,3] (z,y,x)");
    GEOMAE_REQUIRE(num_points >= 0 && batch_size >= 1 && gz >= 1 && gy >= 1 && gx >= 1, "pillar_segment: bad sizes");
    GEOMAE_REQUIRE(cell_table && voxel_coors && seg_start && sample_start && num_pillars,
                   "pillar_segment: null output");
    const int64_t cells = (int64_t)batch_size * gz * gy * gx;
    GEOMAE_REQUIRE(cells < (int64_t)1 << 31 && num_points < (int64_t)1 << 31, "pillar_segment: table too large");
    const int64_t need = geomae_pillar_segment_workspace_bytes(num_points, batch_size, gz, gy, gx);
    if (workspace_bytes < need || (!workspace && need > 0)) {
        set_error("pillar_segment: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
        return GEOMAE_ERR_WORKSPACE;
    }
    auto al = [](int64_t b) { return (b + 255) / 256 * 256; };
    int32_t* rank_in_cell = (int32_t*)workspace;
    int2* tile_sums = (int2*)((char*)workspace + al(num_points * 4));
    const int tiles = (int)((cells + kTile - 1) / kTile);
    GEOMAE_HIP(hipMemsetAsync(cell_table, 0, cells * sizeof(int32_t), stream));
    if (num_points > 0) {
        GEOMAE_REQUIRE(coors && inv && order, "pillar_segment: null argument");
        hipLaunchKernelGGL(hist_kernel, dim3(stream_grid(num_points, kBlk)), dim3(kBlk), 0, stream, coors, num_points,
                           ndim, batch_size, gz, gy, gx, cell_table, rank_in_cell);
    }
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(tiles), dim3(kBlk), 0, stream, cell_table, cells, tile_sums);
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(kBlk), 0, stream, tile_sums, tiles, num_pillars);
    hipLaunchKernelGGL(scan_emit_kernel, dim3(tiles), dim3(kBlk), 0, stream, cell_table, cells, tile_sums, gz, gy,
                       gx, batch_size, voxel_coors, seg_start, sample_start, num_points, num_pillars);
    if (num_points > 0) {
        hipLaunchKernelGGL(place_kernel, dim3(stream_grid(num_points, kBlk)), dim3(kBlk), 0, stream, coors, num_points,
                           ndim, batch_size, gz, gy, gx, cell_table, rank_in_cell, seg_start, inv, order);
    }
    return check_launch("pillar_segment");
}

extern "C" int geomae_segment_max_forward(const float* feat, int32_t channels, const int32_t* order,
                                          const int32_t* seg_start, const int32_t* num_pillars,
                                          int32_t max_pillars, float* out, int32_t* argmax, hipStream_t stream) {
    if (max_pillars <= 0) return GEOMAE_OK;
    GEOMAE_REQUIRE(feat && order && seg_start && num_pillars && out && argmax, "segment_max_forward: null argument");
    GEOMAE_REQUIRE(channels >= 1 && channels <= 1024, "segment_max_forward: channels must be in [1,1024]");
    const int grid = max_pillars < 256 * 32 ? max_pillars : 256 * 32;
    hipLaunchKernelGGL(seg_max_fwd_kernel, dim3(grid), dim3(channels), 0, stream, feat, channels, order, seg_start,
                       num_pillars, out, argmax);
    return check_launch("seg_max_fwd_kernel");
}

extern "C" int geomae_segment_max_backward(const float* grad_out, const int32_t* argmax, const int32_t* inv,
                                           int64_t num_points, int32_t channels, float* grad_feat,
                                           hipStream_t stream) {
    if (num_points <= 0) return GEOMAE_OK;
    GEOMAE_REQUIRE(grad_out && argmax && inv && grad_feat, "segment_max_backward: null argument");
    hipLaunchKernelGGL(seg_max_bwd_kernel, dim3(stream_grid(num_points * channels, kBlk)), dim3(kBlk), 0, stream,
                       grad_out, argmax, inv, num_points, channels, grad_feat);
    return check_launch("seg_max_bwd_kernel");
}
