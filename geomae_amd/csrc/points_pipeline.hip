// Input pipeline on the GPU: everything between the raw sweep files and `points` of
// forward_train, for a whole batch in three launches.
//
// Reference (CPU, 4 workers per GPU; configs/mae_sst/...6x_1e-5.py:167-197):
//   LoadPointsFromMultiSweeps (datasets/pipelines/loading.py:184-233): key frame dt := 0; per sweep: remove_close
//     (|x| < r and |y| < r in the SENSOR frame, :162-182), xyz = xyz @ R^T + t (fp64 products rounded into the fp32
//     array, then an in-place fp64 add rounded again, :224-226), dt = ts - sweep_ts (:227), concatenate;
//   GlobalRotScaleTrans (transforms_3d.py:734-757): rotate about z by a uniform angle (core/points/base_points.py:
//     139-179: xyz @ [[c, s, 0], [-s, c, 0], [0, 0, 1]] in fp32), scale xyz, translate;
//   RandomFlip3D (core/points/lidar_points.py:28-33): horizontal y = -y, vertical x = -x;
//   PointsRangeFilter (base_points.py:207-229): strict open interval on x, y, z;
//   PointShuffle (base_points.py:129-137): a random permutation.
// The random draws stay on the host (they are a handful of scalars per frame and must follow numpy's stream to
// reproduce a reference run); the per-point work is
//   pp_transform_kernel : one pass over the raw points -> transformed rows + keep flag
//   exclusive scan      : kept-point ranks (stable: keeps the concatenation order)
//   pp_scatter_kernel   : compaction + per-frame pseudo-random permutation (a 4-round Feistel bijection on
//                         [0, n_frame) with cycle walking: no sort, no extra pass)
#include "common.h"
#include "../../include/geomae_hip.h"

namespace geomae {

int exclusive_scan_i32(int32_t* values, int64_t n, int32_t* tile_ws, hipStream_t stream);     // hard_voxelize.hip

__device__ __forceinline__ int upper_idx(const int32_t* __restrict__ offs, int count, int i) {
    int lo = 0, hi = count;                      // last k with offs[k] <= i
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offs[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void pp_transform_kernel(const float* __restrict__ raw, int nf, int n,
                                                           const int32_t* __restrict__ sweep_offsets,
                                                           const GeomaeSweepInfo* __restrict__ sweeps, int num_sweeps,
                                                           const GeomaeFrameAug* __restrict__ frames, float x0, float y0,
                                                           float z0, float x1, float y1, float z1, float close_radius,
                                                           float* __restrict__ tmp, int32_t* __restrict__ flag) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const GeomaeSweepInfo S = sweeps[upper_idx(sweep_offsets, num_sweeps, i)];
        const GeomaeFrameAug A = frames[S.frame];
        const float* p = raw + (int64_t)i * nf;
        float x = p[0], y = p[1], z = p[2];
        bool keep = true;
        if (S.remove_close) keep = !(fabsf(x) < close_radius && fabsf(y) < close_radius);
        if (S.has_transform) {
            // fp32 row @ fp64 R^T -> fp64, stored into the fp32 array; then += fp64 t (computed in fp64, stored fp32)
            const double dx = (double)x, dy = (double)y, dz = (double)z;
            const float tx = (float)(dx * S.rot[0] + dy * S.rot[1] + dz * S.rot[2]);
            const float ty = (float)(dx * S.rot[3] + dy * S.rot[4] + dz * S.rot[5]);
            const float tz = (float)(dx * S.rot[6] + dy * S.rot[7] + dz * S.rot[8]);
            x = (float)((double)tx + S.trans[0]);
            y = (float)((double)ty + S.trans[1]);
            z = (float)((double)tz + S.trans[2]);
        }
        // GlobalRotScaleTrans: x' = x c - y s ; y' = x s + y c ; scale ; translate
        const float rx = __fsub_rn(__fmul_rn(x, A.rot_cos), __fmul_rn(y, A.rot_sin));
        const float ry = __fadd_rn(__fmul_rn(x, A.rot_sin), __fmul_rn(y, A.rot_cos));
        x = __fadd_rn(__fmul_rn(rx, A.scale), A.trans[0]);
        y = __fadd_rn(__fmul_rn(ry, A.scale), A.trans[1]);
        z = __fadd_rn(__fmul_rn(z, A.scale), A.trans[2]);
        if (A.flip_horizontal) y = -y;
        if (A.flip_vertical) x = -x;
        keep = keep && x > x0 && y > y0 && z > z0 && x < x1 && y < y1 && z < z1;
        float* o = tmp + (int64_t)i * nf;
        o[0] = x; o[1] = y; o[2] = z;
        for (int c = 3; c < nf; ++c) o[c] = p[c];
        if (nf > 4) o[4] = S.dt;                      // key frame 0, sweeps ts - sweep_ts
        flag[i] = keep ? 1 : 0;
    }
}

__device__ __forceinline__ uint32_t pp_mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// bijection on [0, n): 4 Feistel rounds on the next power of 4, cycle-walked back into range
__device__ __forceinline__ uint32_t pp_permute(uint32_t i, uint32_t n, uint32_t seed_lo, uint32_t seed_hi) {
    if (n <= 1) return i;
    int half = 1;
    while ((1u << (2 * half)) < n) ++half;
    const uint32_t mask = (1u << half) - 1u;
    uint32_t v = i;
    do {
        uint32_t l = v >> half, r = v & mask;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t f = pp_mix(r ^ seed_lo ^ (0x9e3779b9u * (uint32_t)(k + 1)) ^ pp_mix(seed_hi + (uint32_t)k)) & mask;
            const uint32_t nl = r;
            r = l ^ f;
            l = nl;
        }
        v = (l << half) | r;
    } while (v >= n);
    return v;
}

__global__ __launch_bounds__(256) void pp_scatter_kernel(const float* __restrict__ tmp, int nf, int n,
                                                         const int32_t* __restrict__ flag_raw,       // 0 / 1 before the scan
                                                         const int32_t* __restrict__ rank,            // exclusive scan
                                                         const int32_t* __restrict__ frame_first,     // [B+1] raw point offsets
                                                         const GeomaeFrameAug* __restrict__ frames, int num_frames,
                                                         float* __restrict__ out, int32_t* __restrict__ out_offsets) {
    const int total = rank[n - 1] + flag_raw[n - 1];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        if (i <= num_frames) out_offsets[i] = i < num_frames ? (frame_first[i] < n ? rank[frame_first[i]] : total) : total;
        if (!flag_raw[i]) continue;
        const int b = upper_idx(frame_first, num_frames, i);
        const int start = rank[frame_first[b]];
        const int end = (b + 1 < num_frames && frame_first[b + 1] < n) ? rank[frame_first[b + 1]] : total;
        int pos = rank[i] - start;
        const GeomaeFrameAug A = frames[b];
        if (A.shuffle_seed_lo | A.shuffle_seed_hi) pos = (int)pp_permute((uint32_t)pos, (uint32_t)(end - start), A.shuffle_seed_lo, A.shuffle_seed_hi);
        const float* s = tmp + (int64_t)i * nf;
        float* o = out + (int64_t)(start + pos) * nf;
        for (int c = 0; c < nf; ++c) o[c] = s[c];
    }
}

}  // namespace geomae

using namespace geomae;

extern "C" int64_t geomae_points_pipeline_workspace_bytes(int64_t num_points, int32_t num_features) {
    auto al = [](int64_t b) { return (b + 255) / 256 * 256; };
    const int64_t n1 = num_points > 0 ? num_points : 1;
    return al(n1 * num_features * 4) + 2 * al(n1 * 4) + al((n1 / 1024 + 2) * 4);
}

extern "C" int geomae_points_pipeline(const float* raw_points, int64_t num_points, int32_t num_features,
                                      const int32_t* sweep_offsets, const GeomaeSweepInfo* sweeps, int32_t num_sweeps,
                                      const int32_t* frame_offsets, const GeomaeFrameAug* frames, int32_t num_frames,
                                      const float* point_cloud_range, float close_radius, float* out_points,
                                      int32_t* out_offsets, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
    GEOMAE_REQUIRE(num_points >= 0 && num_points < ((int64_t)1 << 31) && num_features >= 3 && num_sweeps >= 1 && num_frames >= 1,
                   "points_pipeline: bad sizes");
    GEOMAE_REQUIRE(sweep_offsets && sweeps && frame_offsets && frames && point_cloud_range && out_offsets,
                   "points_pipeline: null argument");
    const int64_t need = geomae_points_pipeline_workspace_bytes(num_points, num_features);
    if (workspace_bytes < need || !workspace) {
        set_error("points_pipeline: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
        return GEOMAE_ERR_WORKSPACE;
    }
    if (num_points == 0) {
        GEOMAE_HIP(hipMemsetAsync(out_offsets, 0, (size_t)(num_frames + 1) * sizeof(int32_t), stream));
        return GEOMAE_OK;
    }
    GEOMAE_REQUIRE(raw_points && out_points, "points_pipeline: null points");
    auto al = [](int64_t b) { return (b + 255) / 256 * 256; };
    char* ws = (char*)workspace;
    float* tmp = (float*)ws;                           ws += al(num_points * num_features * 4);
    int32_t* flag = (int32_t*)ws;                      ws += al(num_points * 4);
    int32_t* rank = (int32_t*)ws;                      ws += al(num_points * 4);
    int32_t* tiles = (int32_t*)ws;
    const int n = (int)num_points;
    const float* r = point_cloud_range;
    hipLaunchKernelGGL(pp_transform_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, stream, raw_points, num_features, n,
                       sweep_offsets, sweeps, num_sweeps, frames, r[0], r[1], r[2], r[3], r[4], r[5], close_radius, tmp, flag);
    GEOMAE_HIP(hipMemcpyAsync(rank, flag, (size_t)n * 4, hipMemcpyDeviceToDevice, stream));
    int rc = exclusive_scan_i32(rank, n, tiles, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(pp_scatter_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, stream, tmp, num_features, n, flag, rank,
                       frame_offsets, frames, num_frames, out_points, out_offsets);
    return check_launch("points_pipeline");
}
