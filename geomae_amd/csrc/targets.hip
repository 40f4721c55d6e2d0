// Geometric pre-training targets (gfx950): per-(sub-)voxel centroids, occupancy, and surface
// normal + curvature from the 3x3-pillar neighbourhood covariance.
//
// Reference path (mmdet3d/models/detectors/multi_sub_voxel_dynamic_voxelnet_ssl.py):
//   get_centroid_per_voxel x3 (726-768: three more torch.unique sorts + scatter_add_),
//   get_multi_voxel_id_to_tensor_id_for_curv (643-671), spconv get_indice_pairs_implicit_gemm
//   (192-207), cal_regular_voxel_nor_and_curv (575-610: [V,144,3] gathers + batched torch.svd),
//   normalize_centroid_sub_voxel x3 (626-641), get_multi_voxel_id_to_tensor_id_ori (673-722).
// Here: two kernels over the pillar segments built by geomae_pillar_segment.
//   T1 one wave per pillar: walk the pillar's points once, accumulate the 128 low / 16 med
//      sub-voxel sums in LDS with 2^-32 fixed-point int64 atomics (order independent, so the
//      result is deterministic although the in-segment point order is not), emit the
//      normalised dense targets + occupancy for masked rows only, and the raw med/top
//      centroids for every pillar.
//   T2 one wave per masked pillar: 9 neighbour lookups in the dense cell table, 144-candidate
//      scatter matrix by wave reduction, 3x3 symmetric eigen-solve (Jacobi, fp64) in lane 0.
// Component order of every vector is (z, y, x), as in the reference (ssl.py:185-187).
// HBM-bound: T1 reads 20+32+4 B per point and writes (128+16)*13 B per masked pillar.
#include "common.h"
#include "../../include/geomae_hip.h"

namespace geomae {

struct TargetCfg {
    int gy, gx;               // top grid (gz == 1)
    int rl[3], rm[3];         // sub-voxel ratios (z, y, x) low / med
    float cs_top[3], cs_med[3], cs_low[3];   // cell sizes in (z, y, x) order
    float start[3];           // range minimum in (z, y, x) order
};

__device__ __forceinline__ long long fx32(float v) { return __double2ll_rn((double)v * 4294967296.0); }

__device__ __forceinline__ float normalise(float c, int coor, float cs, float start) {
    // (c - (coor * cs + start)) / cs with every operation rounded separately (ssl.py:639-640)
    const float origin = __fadd_rn(__fmul_rn((float)coor, cs), start);
    return __fsub_rn(c, origin) / cs;
}

constexpr int kLowMax = 128, kMedMax = 16;

__global__ __launch_bounds__(64) void centroid_targets_kernel(
    const float* __restrict__ pts, int stride, const int32_t* __restrict__ order,
    const int32_t* __restrict__ seg_start, const int32_t* __restrict__ num_pillars,
    const int4* __restrict__ voxel_coors, const int4* __restrict__ coors_med,
    const int4* __restrict__ coors_low, const int32_t* __restrict__ token_row,
    const int32_t* __restrict__ mask_counts, TargetCfg cfg, float* __restrict__ centroid_low,
    uint8_t* __restrict__ mask_low, float* __restrict__ centroid_med, uint8_t* __restrict__ mask_med,
    float* __restrict__ centroid_top, float* __restrict__ top_raw, float* __restrict__ med_raw,
    uint8_t* __restrict__ med_raw_mask, int32_t* __restrict__ occ_clear) {
    if (occ_clear && blockIdx.x == 0 && threadIdx.x < 2) occ_clear[threadIdx.x] = 0;      // accumulators of the later occ_count_kernel
    __shared__ unsigned long long s_low[kLowMax * 3];
    __shared__ unsigned long long s_med[kMedMax * 3];
    __shared__ int c_low[kLowMax];
    __shared__ int c_med[kMedMax];
    const int lane = threadIdx.x;
    const int V = num_pillars[0];
    const int n_low = cfg.rl[0] * cfg.rl[1] * cfg.rl[2];
    const int n_med = cfg.rm[0] * cfg.rm[1] * cfg.rm[2];
    const int n_keep = token_row ? mask_counts[0] : 0;
    for (int p = blockIdx.x; p < V; p += gridDim.x) {
        for (int t = lane; t < n_low * 3; t += 64) s_low[t] = 0ULL;
        for (int t = lane; t < n_low; t += 64) c_low[t] = 0;
        if (lane < n_med * 3) s_med[lane] = 0ULL;
        if (lane < n_med) c_med[lane] = 0;
        __syncthreads();
        const int s = seg_start[p], e = seg_start[p + 1];
        long long tz = 0, ty = 0, tx = 0;
        for (int j = s + lane; j < e; j += 64) {
            const int i = order[j];
            const float* q = pts + (int64_t)i * stride;
            const long long fz = fx32(q[2]), fy = fx32(q[1]), fxx = fx32(q[0]);
            tz += fz; ty += fy; tx += fxx;
            const int4 cl = coors_low[i];
            const int4 cm = coors_med[i];
            const int sl = (cl.y % cfg.rl[0]) * (cfg.rl[1] * cfg.rl[2]) + (cl.z % cfg.rl[1]) * cfg.rl[2] + cl.w % cfg.rl[2];
            const int sm = (cm.y % cfg.rm[0]) * (cfg.rm[1] * cfg.rm[2]) + (cm.z % cfg.rm[1]) * cfg.rm[2] + cm.w % cfg.rm[2];
            atomicAdd(&s_low[sl * 3 + 0], (unsigned long long)fz);
            atomicAdd(&s_low[sl * 3 + 1], (unsigned long long)fy);
            atomicAdd(&s_low[sl * 3 + 2], (unsigned long long)fxx);
            atomicAdd(&c_low[sl], 1);
            atomicAdd(&s_med[sm * 3 + 0], (unsigned long long)fz);
            atomicAdd(&s_med[sm * 3 + 1], (unsigned long long)fy);
            atomicAdd(&s_med[sm * 3 + 2], (unsigned long long)fxx);
            atomicAdd(&c_med[sm], 1);
        }
        tz = wave_sum(tz); ty = wave_sum(ty); tx = wave_sum(tx);
        __syncthreads();
        const int4 vc = voxel_coors[p];
        int row = p;
        if (token_row) row = token_row[p] - n_keep;          // < 0: visible pillar, no dense target rows
        const double k32 = 1.0 / 4294967296.0;
        // --- top centroid
        if (lane < 3) {
            const long long tsum = lane == 0 ? tz : (lane == 1 ? ty : tx);
            const float c = (float)((double)tsum * k32 / (double)(e - s));
            top_raw[(int64_t)p * 3 + lane] = c;
            if (row >= 0) {
                const int coor = lane == 0 ? vc.y : (lane == 1 ? vc.z : vc.w);
                centroid_top[(int64_t)row * 3 + lane] = normalise(c, coor, cfg.cs_top[lane], cfg.start[lane]);
            }
        }
        // --- med sub-voxels: raw (every pillar, for the curvature neighbourhood) + normalised rows
        if (lane < n_med) {
            const int cnt = c_med[lane];
            const int oz = lane / (cfg.rm[1] * cfg.rm[2]);
            const int oy = (lane / cfg.rm[2]) % cfg.rm[1];
            const int ox = lane % cfg.rm[2];
            const int coor[3] = {vc.y * cfg.rm[0] + oz, vc.z * cfg.rm[1] + oy, vc.w * cfg.rm[2] + ox};
            med_raw_mask[(int64_t)p * n_med + lane] = cnt > 0;
            if (row >= 0) mask_med[(int64_t)row * n_med + lane] = cnt > 0;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                float c = 0.0f, nc = 0.0f;
                if (cnt > 0) {
                    c = (float)((double)(long long)s_med[lane * 3 + d] * k32 / (double)cnt);
                    nc = normalise(c, coor[d], cfg.cs_med[d], cfg.start[d]);
                }
                med_raw[((int64_t)p * n_med + lane) * 3 + d] = c;
                if (row >= 0) centroid_med[((int64_t)row * n_med + lane) * 3 + d] = nc;
            }
        }
        // --- low sub-voxels (masked rows only)
        if (row >= 0) {
            for (int sl = lane; sl < n_low; sl += 64) {
                const int cnt = c_low[sl];
                const int oz = sl / (cfg.rl[1] * cfg.rl[2]);
                const int oy = (sl / cfg.rl[2]) % cfg.rl[1];
                const int ox = sl % cfg.rl[2];
                const int coor[3] = {vc.y * cfg.rl[0] + oz, vc.z * cfg.rl[1] + oy, vc.w * cfg.rl[2] + ox};
                mask_low[(int64_t)row * n_low + sl] = cnt > 0;
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    float nc = 0.0f;
                    if (cnt > 0) {
                        const float c = (float)((double)(long long)s_low[sl * 3 + d] * k32 / (double)cnt);
                        nc = normalise(c, coor[d], cfg.cs_low[d], cfg.start[d]);
                    }
                    centroid_low[((int64_t)row * n_low + sl) * 3 + d] = nc;
                }
            }
        }
        __syncthreads();
    }
}

// occupied cells over the output rows (the normalisers of the two masked-MSE losses): byte sums of the
// 0/1 occupancy arrays; one atomic per workgroup (per-pillar atomics on two words serialise: 360 us)
__global__ __launch_bounds__(256) void occ_count_kernel(const uint8_t* __restrict__ mask_low, int64_t n_low_bytes,
                                                        const uint8_t* __restrict__ mask_med, int64_t n_med_bytes,
                                                        int32_t* __restrict__ occ_counts) {
    __shared__ int sm[2][4];
    // the arrays hold 0 / 1 bytes: 16 of them per load, counted with popcount (byte loads made this a 17 us kernel)
    auto count = [&](const uint8_t* __restrict__ m, int64_t n) {
        int c = 0;
        const int64_t n16 = ((uintptr_t)m & 15) == 0 ? n >> 4 : 0;
        const uint4* m16 = reinterpret_cast<const uint4*>(m);
        for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
            const uint4 v = m16[i];
            c += __builtin_popcount(v.x) + __builtin_popcount(v.y) + __builtin_popcount(v.z) + __builtin_popcount(v.w);
        }
        for (int64_t i = (n16 << 4) + blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) c += m[i];
        return c;
    };
    int a = count(mask_low, n_low_bytes), b = count(mask_med, n_med_bytes);
    a = wave_sum(a);
    b = wave_sum(b);
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = a; sm[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&occ_counts[0], sm[0][0] + sm[0][1] + sm[0][2] + sm[0][3]);
        atomicAdd(&occ_counts[1], sm[1][0] + sm[1][1] + sm[1][2] + sm[1][3]);
    }
}

// symmetric 3x3 eigen-decomposition, cyclic Jacobi in fp64; a = [xx, xy, xz, yy, yz, zz]
__device__ void eig3(const double a_in[6], double w[3], double v[3][3]) {
    double a[3][3] = {{a_in[0], a_in[1], a_in[2]}, {a_in[1], a_in[3], a_in[4]}, {a_in[2], a_in[4], a_in[5]}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) v[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        const double diag = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
        if (off <= 1e-300 || off <= 1e-22 * diag) break;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0;
            const int q = pq == 0 ? 1 : 2;
            if (a[p][q] == 0.0) continue;
            const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            const double app = a[p][p], aqq = a[q][q], apq = a[p][q];
            a[p][p] = app - t * apq;
            a[q][q] = aqq + t * apq;
            a[p][q] = a[q][p] = 0.0;
            const int r = 3 - p - q;
            const double arp = a[r][p], arq = a[r][q];
            a[r][p] = a[p][r] = c * arp - s * arq;
            a[r][q] = a[q][r] = s * arp + c * arq;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double vkp = v[k][p], vkq = v[k][q];
                v[k][p] = c * vkp - s * vkq;
                v[k][q] = s * vkp + c * vkq;
            }
        }
    }
    w[0] = a[0][0]; w[1] = a[1][1]; w[2] = a[2][2];
}

// eigen-decomposition of one pillar's 3x3 scatter matrix -> unit normal (canonical sign) and the curvature triple
__device__ __forceinline__ void normal_curv_from_cov(const float (&c)[6], int64_t row, float* __restrict__ normal,
                                                     double* __restrict__ curv) {
    const double a[6] = {c[0], c[1], c[2], c[3], c[4], c[5]};
    double w[3], v[3][3];
    eig3(a, w, v);
    // singular values of the PSD scatter matrix = |eigenvalues|, descending
    int idx[3] = {0, 1, 2};
    double s[3] = {fabs(w[0]), fabs(w[1]), fabs(w[2])};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2 - i; ++j)
            if (s[j] < s[j + 1]) {
                double ts = s[j]; s[j] = s[j + 1]; s[j + 1] = ts;
                int ti = idx[j]; idx[j] = idx[j + 1]; idx[j + 1] = ti;
            }
    float nz = (float)v[0][idx[2]], ny = (float)v[1][idx[2]], nx = (float)v[2][idx[2]];
    const float len = sqrtf(nz * nz + ny * ny + nx * nx);
    nz /= len; ny /= len; nx /= len;
    // canonical sign: the component of largest magnitude is positive (ties: lowest index)
    const float az = fabsf(nz), ay = fabsf(ny), ax = fabsf(nx);
    const float lead = (az >= ay && az >= ax) ? nz : (ay >= ax ? ny : nx);
    if (lead < 0.0f) { nz = -nz; ny = -ny; nx = -nx; }
    normal[(int64_t)row * 3 + 0] = nz;
    normal[(int64_t)row * 3 + 1] = ny;
    normal[(int64_t)row * 3 + 2] = nx;
    // est_curv = (S.double() + 1e-9) / sum  (ssl.py:604-607); S is fp32 in the reference
    const double e0 = (double)(float)s[0] + 1e-9, e1 = (double)(float)s[1] + 1e-9, e2 = (double)(float)s[2] + 1e-9;
    const double tot = e0 + e1 + e2;
    curv[(int64_t)row * 3 + 0] = e0 / tot;
    curv[(int64_t)row * 3 + 1] = e1 / tot;
    curv[(int64_t)row * 3 + 2] = e2 / tot;
}

__global__ __launch_bounds__(64) void normal_curv_kernel(
    const int32_t* __restrict__ num_pillars, const int4* __restrict__ voxel_coors,
    const int32_t* __restrict__ cell_table, const int32_t* __restrict__ token_row,
    const int32_t* __restrict__ mask_counts, TargetCfg cfg, int n_batch, const float* __restrict__ top_raw,
    const float* __restrict__ med_raw, const uint8_t* __restrict__ med_raw_mask, float* __restrict__ normal,
    double* __restrict__ curv, float* __restrict__ cov_out, int cov_only) {
    const int lane = threadIdx.x;
    const int V = num_pillars[0];
    const int n_med = cfg.rm[0] * cfg.rm[1] * cfg.rm[2];
    const int n_keep = token_row ? mask_counts[0] : 0;
    for (int p = blockIdx.x; p < V; p += gridDim.x) {
        int row = p;
        if (token_row) row = token_row[p] - n_keep;
        if (row < 0) continue;
        const int4 vc = voxel_coors[p];
        const float oz = top_raw[(int64_t)p * 3 + 0], oy = top_raw[(int64_t)p * 3 + 1], ox = top_raw[(int64_t)p * 3 + 2];
        float c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
        for (int t = lane; t < 9 * n_med; t += 64) {
            const int k = t / n_med, s = t - k * n_med;
            const int y = vc.z + k / 3 - 1, x = vc.w + k % 3 - 1;
            if (y < 0 || y >= cfg.gy || x < 0 || x >= cfg.gx) continue;
            const int q = cell_table[((int64_t)vc.x * cfg.gy + y) * cfg.gx + x];
            if (q < 0) continue;
            if (!med_raw_mask[(int64_t)q * n_med + s]) continue;
            const float* m = med_raw + ((int64_t)q * n_med + s) * 3;
            const float dz = m[0] - oz, dy = m[1] - oy, dx = m[2] - ox;
            c00 += dz * dz; c01 += dz * dy; c02 += dz * dx;
            c11 += dy * dy; c12 += dy * dx; c22 += dx * dx;
        }
        c00 = wave_sum(c00); c01 = wave_sum(c01); c02 = wave_sum(c02);
        c11 = wave_sum(c11); c12 = wave_sum(c12); c22 = wave_sum(c22);
        if (lane == 0) {
            const float c[6] = {c00, c01, c02, c11, c12, c22};
            if (!cov_only) normal_curv_from_cov(c, row, normal, curv);
            if (cov_out) {
                float* co = cov_out + (int64_t)row * 6;
                co[0] = c00; co[1] = c01; co[2] = c02; co[3] = c11; co[4] = c12; co[5] = c22;
            }
        }
    }
}

// Second half of the normal / curvature targets when the scatter matrices went through memory (cov_only above): one
// THREAD per target row.  In the fused kernel lane 0 of every pillar's wave runs the fp64 Jacobi iteration alone --
// ~11 k issue cycles with 63 lanes idle, 15 k waves: that was the kernel's whole duration (53 us), and it ran beside the
// latency-bound encoder forward.  Same arithmetic per pillar, so the results are bit-identical.
__global__ __launch_bounds__(64) void normal_eig_kernel(const int32_t* __restrict__ num_pillars,
                                                        const int32_t* __restrict__ mask_counts,
                                                        const float* __restrict__ cov, float* __restrict__ normal,
                                                        double* __restrict__ curv) {
    const int rows = mask_counts ? mask_counts[1] : num_pillars[0];
    const int row = blockIdx.x * 64 + threadIdx.x;
    if (row >= rows) return;
    const float* co = cov + (int64_t)row * 6;
    const float c[6] = {co[0], co[1], co[2], co[3], co[4], co[5]};
    normal_curv_from_cov(c, row, normal, curv);
}

static int fill_cfg(TargetCfg& c, const GeomaeTargetConfig* g) {
    GEOMAE_REQUIRE(g, "geometry_targets: null config");
    GEOMAE_REQUIRE(g->grid_size[0] == 1, "geometry_targets: the top grid must be one cell high (pillars), got gz=%d "
                   "(same restriction as the reference, ssl.py:645)", g->grid_size[0]);
    c.gy = g->grid_size[1];
    c.gx = g->grid_size[2];
    int nl = 1, nm = 1;
    for (int d = 0; d < 3; ++d) {
        c.rl[d] = g->ratio_low[d];
        c.rm[d] = g->ratio_med[d];
        GEOMAE_REQUIRE(c.rl[d] >= 1 && c.rm[d] >= 1, "geometry_targets: ratios must be >= 1");
        nl *= c.rl[d];
        nm *= c.rm[d];
        // (z, y, x) <- (x, y, z)
        c.cs_top[d] = g->voxel_size_top[2 - d];
        c.cs_med[d] = g->voxel_size_med[2 - d];
        c.cs_low[d] = g->voxel_size_low[2 - d];
        c.start[d] = g->coors_range[2 - d];
        // nested cells: ratio * sub-cell must equal the pillar cell exactly in fp32, so that the
        // parent of a sub-voxel (coor // ratio, ssl.py:659,696) is the pillar the point fell in
        GEOMAE_REQUIRE((float)c.rl[d] * c.cs_low[d] == c.cs_top[d] && (float)c.rm[d] * c.cs_med[d] == c.cs_top[d],
                       "geometry_targets: sub-voxel sizes must nest exactly in the pillar size (axis %d)", 2 - d);
    }
    GEOMAE_REQUIRE(nl <= kLowMax && nm <= kMedMax && nm * 9 <= 64 * 3, "geometry_targets: too many sub-voxels per pillar");
    return GEOMAE_OK;
}

}  // namespace geomae

using namespace geomae;

extern "C" int geomae_geometry_targets(const float* points, int32_t num_features, const int32_t* order,
                                       const int32_t* seg_start, const int32_t* num_pillars, int32_t max_pillars,
                                       const int32_t* voxel_coors, const int32_t* coors_med,
                                       const int32_t* coors_low, const int32_t* cell_table, int32_t batch_size,
                                       const int32_t* token_row, const int32_t* mask_counts,
                                       const GeomaeTargetConfig* config, float* centroid_low, uint8_t* mask_low,
                                       float* centroid_med, uint8_t* mask_med, float* centroid_top, float* normal,
                                       double* curv, float* top_raw, float* med_raw, uint8_t* med_raw_mask,
                                       float* cov_out, int32_t* occ_counts, int32_t num_rows, hipStream_t stream) {
    if (max_pillars <= 0) return GEOMAE_OK;
    GEOMAE_REQUIRE(points && order && seg_start && num_pillars && voxel_coors && coors_med && coors_low && cell_table,
                   "geometry_targets: null input");
    GEOMAE_REQUIRE(centroid_low && mask_low && centroid_med && mask_med && centroid_top && normal && curv && top_raw &&
                   med_raw && med_raw_mask, "geometry_targets: null output");
    GEOMAE_REQUIRE((token_row == nullptr) == (mask_counts == nullptr), "geometry_targets: token_row and mask_counts go together");
    TargetCfg c;
    int rc = fill_cfg(c, config);
    if (rc) return rc;
    const int grid = max_pillars < 256 * 64 ? max_pillars : 256 * 64;
    hipLaunchKernelGGL(centroid_targets_kernel, dim3(grid), dim3(64), 0, stream, points, num_features, order,
                       seg_start, num_pillars, (const int4*)voxel_coors, (const int4*)coors_med,
                       (const int4*)coors_low, token_row, mask_counts, c, centroid_low, mask_low, centroid_med,
                       mask_med, centroid_top, top_raw, med_raw, med_raw_mask, occ_counts);
    // with a scatter-matrix buffer from the caller the eigen-decompositions run one per thread in a second launch
    hipLaunchKernelGGL(normal_curv_kernel, dim3(grid), dim3(64), 0, stream, num_pillars, (const int4*)voxel_coors,
                       cell_table, token_row, mask_counts, c, batch_size, top_raw, med_raw, med_raw_mask, normal, curv,
                       cov_out, cov_out ? 1 : 0);
    if (cov_out)
        hipLaunchKernelGGL(normal_eig_kernel, dim3((max_pillars + 63) / 64), dim3(64), 0, stream, num_pillars,
                           mask_counts, (const float*)cov_out, normal, curv);
    if (occ_counts) {
        // rows: masked pillars (device count) or all pillars; the host-side upper bound is max_pillars,
        // rows beyond the real count were never written, so count over the rows the caller allocated
        GEOMAE_REQUIRE(num_rows >= 0, "geometry_targets: num_rows needed for occ_counts");
        const int n_low = c.rl[0] * c.rl[1] * c.rl[2], n_med = c.rm[0] * c.rm[1] * c.rm[2];
        hipLaunchKernelGGL(occ_count_kernel, dim3(128), dim3(256), 0, stream, mask_low, (int64_t)num_rows * n_low,
                           mask_med, (int64_t)num_rows * n_med, occ_counts);
    }
    return check_launch("geometry_targets");
}
