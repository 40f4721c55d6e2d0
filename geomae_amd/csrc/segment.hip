// Pillar segment build + segmented reductions (gfx950).
//
// Replaces the reference's six torch.unique(coors, dim=0) sorts per iteration
// (mmdet3d/ops/sst/sst_ops.py:8-39 scatter_v2, called 3x by DynamicScatterVFE
// voxel_encoder.py:375,407; detectors/multi_sub_voxel_dynamic_voxelnet_ssl.py:749 called 3x)
// and torch_scatter's mean/max by ONE counting sort per batch over the dense cell table
// (B*gz*gy*gx int32, 2.5 MB for the nuScenes config):
//   1. hist   : rank_in_cell[i] = atomicAdd(table[key_i], 1)          (1 atomic / point)
//   2. scan   : ONE pass over the cells (decoupled look-back): exclusive scans of (occupied, count) -> pillar id,
//               segment start; pillar ids increase with the cell key, i.e. exactly the lexicographic
//               (b, z, y, x) order torch.unique(dim=0) returns; table[cell] <- pillar id | -1
//               (kept: it is the neighbour / parent lookup table of the target kernels)
//   3. place  : inv[i] = table[key_i];  order[start[inv[i]] + rank_in_cell[i]] = i
// Three launches (round 2: five -- tile totals, a single-workgroup scan of them, emit -- plus a memset kernel).
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

// (Round 4, measured and rejected: one returned atomic per RUN of equal keys inside a wave -- head lanes by ballot, the base
//  handed on by shuffle.  The frames' point order is firing order (every beam of one azimuth step, then the next step) or,
//  behind the training pipeline's PointShuffle, random: points of one pillar are never neighbours in the array, the runs
//  have length one and the kernel kept its 138 us at 1.02 M points.)
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

// ---- single-pass scan of (occupied, count) over the cell table with decoupled look-back.
// Tile t publishes one 64-bit status word: [flag:2 | occupied:31 | points:31], flag 1 = the tile's own totals
// (published as soon as they are known), flag 2 = the inclusive prefix up to and including the tile.  A tile finds its
// exclusive prefix by walking back over the status words: inclusive prefixes end the walk, aggregates are added and the
// walk goes on, zero (= not yet published) is polled.  The value travels INSIDE the word that carries the flag, so no
// fence is needed (an agent-scope fence is an L2 write-back on this 8-XCD part, ~20 us per launch: docs/LAB_NOTES.md, rounds 1-3);
// the words are written and read with agent-scope atomics, which bypass the non-coherent per-XCD L2.  Tiles are handed
// out by a ticket counter, so a tile's predecessors have always started (no dependence on the dispatch order).
// status [tiles] + ticket live in the caller's workspace and must be ZERO at launch.
constexpr unsigned long long kFlagAgg = 1ull << 62, kFlagIncl = 2ull << 62, kValMask = (1ull << 31) - 1;
__device__ __forceinline__ unsigned long long pack_status(unsigned long long flag, int occ, int cnt) {
    return flag | ((unsigned long long)(unsigned)occ << 31) | (unsigned long long)(unsigned)cnt;
}

__global__ __launch_bounds__(kBlk) void scan_lookback_kernel(int32_t* __restrict__ table, int64_t cells, int n_tiles,
                                                             unsigned long long* __restrict__ status,
                                                             unsigned int* __restrict__ ticket, int gz, int gy, int gx,
                                                             int n_batch, int32_t* __restrict__ voxel_coors,
                                                             int32_t* __restrict__ seg_start,
                                                             int32_t* __restrict__ sample_start,
                                                             int32_t* __restrict__ sample_start_mirror,
                                                             int32_t* __restrict__ num_pillars) {
    __shared__ int sm[8];
    __shared__ int s_tile, s_off_occ, s_off_cnt;
    if (threadIdx.x == 0) s_tile = (int)atomicAdd(ticket, 1u);
    __syncthreads();
    const int tile = s_tile;
    const int64_t base = (int64_t)tile * kTile + (int64_t)threadIdx.x * kItems;
    int v[kItems];
    if (base + kItems <= cells) {
        const int4 q = *reinterpret_cast<const int4*>(table + base);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
        for (int k = 0; k < kItems; ++k) v[k] = base + k < cells ? table[base + k] : 0;
    }
    int occ = 0, cnt = 0;
#pragma unroll
    for (int k = 0; k < kItems; ++k) { occ += v[k] > 0; cnt += v[k]; }
    int tot_occ, tot_cnt;
    int eo = block_excl_scan(occ, &tot_occ, sm);
    int ec = block_excl_scan(cnt, &tot_cnt, sm);
    if (threadIdx.x < 64) {                      // wave 0: publish, then look back 64 tiles at a time
        const int lane = threadIdx.x;
        if (lane == 0)
            __hip_atomic_store(status + tile, pack_status(tile == 0 ? kFlagIncl : kFlagAgg, tot_occ, tot_cnt), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        int run_occ = 0, run_cnt = 0;
        int hi = tile;                           // tiles [hi - 64, hi) are inspected next
        while (hi > 0) {
            const int t = hi - 1 - lane;         // lane 0 = nearest predecessor
            unsigned long long w = kFlagIncl;    // (lanes before tile 0 count as a zero inclusive prefix)
            if (t >= 0) {
                do {
                    w = __hip_atomic_load(status + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } while ((w >> 62) == 0);
            }
            // nearest lane holding an inclusive prefix ends the walk: sum the aggregates in front of it and add it
            const unsigned long long incl_mask = __ballot((w >> 62) == 2);
            const int stop = incl_mask ? __ffsll((long long)incl_mask) - 1 : 64;      // first such lane
            const bool use = lane <= stop && lane < 64;
            int o = use ? (int)((w >> 31) & kValMask) : 0, c = use ? (int)(w & kValMask) : 0;
            o = wave_sum(o);
            c = wave_sum(c);
            run_occ += o;
            run_cnt += c;
            if (incl_mask) break;
            hi -= 64;
        }
        if (lane == 0) {
            s_off_occ = run_occ;
            s_off_cnt = run_cnt;
            if (tile > 0)
                __hip_atomic_store(status + tile, pack_status(kFlagIncl, run_occ + tot_occ, run_cnt + tot_cnt), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    eo += s_off_occ;
    ec += s_off_cnt;
    const int64_t cells_per_sample = (int64_t)gz * gy * gx;
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const int64_t c = base + k;
        if (c >= cells) break;
        if (c % cells_per_sample == 0) {
            sample_start[c / cells_per_sample] = eo;
            if (sample_start_mirror) sample_start_mirror[c / cells_per_sample] = eo;
        }
        const int pts = v[k];
        if (pts > 0) {
            int64_t r = c;
            const int x = (int)(r % gx); r /= gx;
            const int y = (int)(r % gy); r /= gy;
            const int z = (int)(r % gz); r /= gz;
            reinterpret_cast<int4*>(voxel_coors)[eo] = make_int4((int)r, z, y, x);
            seg_start[eo] = ec;
            v[k] = eo;                            // table[cell] <- pillar id
            ++eo;
            ec += pts;
        } else {
            v[k] = -1;
        }
    }
    if (base + kItems <= cells) {
        *reinterpret_cast<int4*>(table + base) = make_int4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int k = 0; k < kItems; ++k)
            if (base + k < cells) table[base + k] = v[k];
    }
    if (tile == n_tiles - 1 && threadIdx.x == 0) {
        const int V = s_off_occ + tot_occ;
        num_pillars[0] = V;
        sample_start[n_batch] = V;
        if (sample_start_mirror) sample_start_mirror[n_batch] = V;
        seg_start[V] = s_off_cnt + tot_cnt;                 // number of valid points (= n_points when none is dropped)
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

// workspace layout: [status (tiles x 8 B) | ticket (8 B) | pad to 256 B][rank_in_cell (N x 4 B)]
static int64_t scan_state_bytes(int64_t cells) {
    const int64_t tiles = (cells + kTile - 1) / kTile;
    return (tiles * 8 + 8 + 255) / 256 * 256;
}

extern "C" int64_t geomae_pillar_segment_workspace_bytes(int64_t num_points, int32_t batch_size, int32_t gz,
                                                         int32_t gy, int32_t gx) {
    const int64_t cells = (int64_t)batch_size * gz * gy * gx;
    // scan state (tile status words + ticket) first, then rank_in_cell [n] int32; 256-byte aligned pieces
    auto al = [](int64_t b) { return (b + 255) / 256 * 256; };
    return scan_state_bytes(cells) + al(num_points * 4);
}

extern "C" int geomae_pillar_segment(const int32_t* coors, int64_t num_points, int32_t batch_size, int32_t gz,
                                     int32_t gy, int32_t gx, int32_t* cell_table, int32_t* voxel_coors,
                                     int32_t* inv, int32_t* order, int32_t* seg_start, int32_t* sample_start,
                                     int32_t* num_pillars, void* workspace, int64_t workspace_bytes,
                                     hipStream_t stream) {
    return geomae_pillar_segment_nd(coors, 4, num_points, batch_size, gz, gy, gx, cell_table, voxel_coors, inv, order,
                                    seg_start, sample_start, num_pillars, workspace, workspace_bytes, stream);
}

extern "C" int64_t geomae_pillar_segment_scan_state_bytes(int32_t batch_size, int32_t gz, int32_t gy, int32_t gx) {
    return scan_state_bytes((int64_t)batch_size * gz * gy * gx);
}

extern "C" int geomae_pillar_segment_nd(const int32_t* coors, int32_t ndim, int64_t num_points, int32_t batch_size,
                                        int32_t gz, int32_t gy, int32_t gx, int32_t* cell_table, int32_t* voxel_coors,
                                        int32_t* inv, int32_t* order, int32_t* seg_start, int32_t* sample_start,
                                        int32_t* num_pillars, void* workspace, int64_t workspace_bytes,
                                        hipStream_t stream) {
    return geomae_pillar_segment_ex(coors, ndim, num_points, batch_size, gz, gy, gx, cell_table, voxel_coors, inv, order,
                                    seg_start, sample_start, num_pillars, workspace, workspace_bytes, nullptr, 0, stream);
}

extern "C" int geomae_pillar_segment_ex(const int32_t* coors, int32_t ndim, int64_t num_points, int32_t batch_size,
                                        int32_t gz, int32_t gy, int32_t gx, int32_t* cell_table, int32_t* voxel_coors,
                                        int32_t* inv, int32_t* order, int32_t* seg_start, int32_t* sample_start,
                                        int32_t* num_pillars, void* workspace, int64_t workspace_bytes,
                                        int32_t* sample_start_mirror, int32_t prezeroed, hipStream_t stream) {
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
    const int64_t state = scan_state_bytes(cells);
    const int tiles = (int)((cells + kTile - 1) / kTile);
    unsigned long long* status = (unsigned long long*)workspace;
    unsigned int* ticket = (unsigned int*)(status + tiles);
    int32_t* rank_in_cell = (int32_t*)((char*)workspace + state);
    if (!prezeroed) {
        GEOMAE_HIP(hipMemsetAsync(cell_table, 0, cells * sizeof(int32_t), stream));
        GEOMAE_HIP(hipMemsetAsync(workspace, 0, (size_t)state, stream));
    }
    if (num_points > 0) {
        GEOMAE_REQUIRE(coors && inv && order, "pillar_segment: null argument");
        hipLaunchKernelGGL(hist_kernel, dim3(stream_grid(num_points, kBlk)), dim3(kBlk), 0, stream, coors, num_points,
                           ndim, batch_size, gz, gy, gx, cell_table, rank_in_cell);
    }
    hipLaunchKernelGGL(scan_lookback_kernel, dim3(tiles), dim3(kBlk), 0, stream, cell_table, cells, tiles, status, ticket,
                       gz, gy, gx, batch_size, voxel_coors, seg_start, sample_start, sample_start_mirror, num_pillars);
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
