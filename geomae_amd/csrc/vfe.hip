// Fused DynamicScatterVFE for gfx950, forward and backward.
//
// Reference: DynamicScatterVFE.forward (mmdet3d/models/voxel_encoders/voxel_encoder.py:358-419):
//   f = [p(5) | xyz - pillar mean | xyz - pillar centre]                       (11 features)
//   h0 = ReLU(BN64(W0 f)) ; m0 = scatter_max(h0) ; g = [h0 | m0[pillar]]      (128)
//   h1 = ReLU(BN128(W1 g)) ; voxel_feats = scatter_max(h1)
// run there as three torch.unique sorts + torch_scatter + cuBLAS + BatchNorm kernels with every [N, C]
// intermediate in HBM (and ~100 autograd nodes in the backward).  Here the points are walked in pillar
// order (geomae_pillar_segment), 16 at a time per wave in the transposed-product MFMA layout of
// sst_device.h, in EXACT fp32 (v_mfma_f32_16x16x4_f32: raw coordinates do not survive bf16), and nothing
// per-point is stored except what the weight-gradient contraction needs:
//   * BatchNorm needs full-batch statistics before it can normalise, so each layer is two sweeps
//     (statistics, then apply + segmented max); y = W f is recomputed instead of stored (11x64 and
//     128x128 MACs per point are cheaper than a 256 / 512 B round trip per point);
//   * a wave owns 64 consecutive points of the pillar-sorted order; a 16 x C tile goes through LDS and lane c
//     scans its channel over the 16 points, carrying the open pillar in registers.  Pillars inside the
//     wave's range are plain stores; the (at most two) pillars cut by the range boundaries are combined
//     across waves with atomics, and every consumer of a pooled row is a later kernel;
//   * the max-pool backward needs no arg-max: the forward value is recomputed bit-identically (same
//     device function, same MFMA order) and the gradient is routed where h == max (ties at 0 are killed
//     by ReLU'; exact duplicates of a point would both receive it);
//   * statistics cross workgroups as fp64 atomics on 2C words; naiveSyncBN1d's cross-rank exchange
//     (mmdet3d/ops/norm.py:54-86) happens between the kernels on those words (host side, RCCL).
#include "common.h"
#include "../../include/geomae_hip.h"
#include "sst_device.h"

namespace geomae {

constexpr int kVfeBlk = 512;            // 8 waves share one LDS copy of W1: 2 waves per SIMD hide each other's latency
constexpr int kVfeWaves = kVfeBlk / 64;
constexpr int kW1bLd = 128 + 8;         // bf16 LDS row of one half (hi or lo) of the split W1 (+16 B pad: conflict-free b128 reads)
constexpr int kW1Ld = kW1bLd;           // the W1 buffer is declared as float [128][kW1Ld] = two bf16 [128][kW1bLd] halves
constexpr int kTileLd = 128 + 4;        // fp32 LDS row of the per-wave [16 x 128] tile
constexpr int kTile0Ld = 80 + 4;        // ... of the [16 x 64 (+16 features)] tile
constexpr int kVfePts = 64;             // points per wave: 4 tiles of 16

__device__ __forceinline__ f32x4 mfma_f32(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

struct VfeGeo {
    const float* feat;            // [N,16] decorated features of the points in pillar order (vfe_prepare)
    const int32_t* pid;           // [N] pillar of each sorted point
    const int32_t* seg_start;
    int n_points;
};

// every wave owns kVfePts consecutive sorted points -- all waves do the same work however the points are
// distributed over pillars (a wave per range of WHOLE pillars made the wave with the 284-point pillar run
// 4.5x longer than the rest, profiles/r01q_step_timeline.txt)
struct WaveRange { int j_lo, j_hi; };

__device__ __forceinline__ WaveRange wave_range(const VfeGeo& G, int wave_id) {
    WaveRange r;
    const int64_t lo = (int64_t)wave_id * kVfePts;
    r.j_lo = lo < G.n_points ? (int)lo : G.n_points;
    r.j_hi = r.j_lo + kVfePts < G.n_points ? r.j_lo + kVfePts : G.n_points;
    return r;
}

// A zero the optimiser cannot see through.  Added to the weight / BN-parameter addresses inside the point
// loops so that LICM does not hoist ~250 registers of loop-invariant operands out of them (that hoisting is
// what pinned these kernels at 1 wave/SIMD).
__device__ __forceinline__ int opaque_zero() {
    int z = 0;
    asm volatile("" : "+s"(z));
    return z;
}

__device__ __forceinline__ int pillar_of(const VfeGeo& G, int j) { return G.pid[j]; }

// the lane's 4 of the 16 (11 used) decorated features of its point, T-layout: feature index 4g + r.
// One coalesced 16-byte load: the gathers (order -> point -> pillar mean / centre) were hoisted into
// vfe_prepare_kernel, because as per-tile dependent loads they cost ~5 us per tile in every sweep.
__device__ __forceinline__ void build_features(const VfeGeo& G, int j, bool valid, int g, float (&f)[4]) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) v = *reinterpret_cast<const float4*>(G.feat + (int64_t)j * 16 + 4 * g);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
}

// Moments of the 11 decorated features over all points: S1[k] = sum f_k, S2[k][l] = sum f_k f_l  (fp64, [16] + [11][11]).
// Layer 0 is linear without bias (y0 = W0 f), so everything its BatchNorm needs from the batch follows from them:
//   sum y0_c = W0[c] . S1,  sum y0_c^2 = W0[c] S2 W0[c]^T        (the forward statistics: no sweep over the points)
//   sum yhat0_c f = invstd_c (W0[c] S2 - mean_c S1)               (the BatchNorm backward term of dW0: no second sweep)
// They depend on the points only, so they are computed with the features, in stage 1 of the batch (a step earlier).
constexpr int kMomS2 = 16, kMomCount = 16 + 121;        // S1 padded to 16, then S2 row-major [11][11]
constexpr int kMomPad = 144;                            // doubles per partial row
constexpr int kMomBlocks = 256;                         // grid cap of the prepare kernel when it accumulates moments

// features of every point, written in pillar order: [x y z i | dt x-mx y-my z-mz | x-cx y-cy z-cz 0 | 0 0 0 0]
// partials (or null): [gridDim.x][kMomPad] doubles, this workgroup's sums of the 11 features and their 66 products
__global__ __launch_bounds__(256) void vfe_prepare_kernel(const float* __restrict__ pts, int stride, int64_t n,
                                                          const int32_t* __restrict__ order,
                                                          const int32_t* __restrict__ inv,
                                                          const float* __restrict__ mean,
                                                          const int4* __restrict__ voxel_coors, float vx, float vy,
                                                          float vz, float xo, float yo, float zo,
                                                          float* __restrict__ feat, int32_t* __restrict__ pid_out,
                                                          double* __restrict__ partials) {
    float s1[11], s2[66];
#pragma unroll
    for (int k = 0; k < 11; ++k) s1[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 66; ++k) s2[k] = 0.f;
    for (int64_t j = blockIdx.x * 256 + threadIdx.x; j < n; j += (int64_t)gridDim.x * 256) {
        const int i = order[j];
        const int p = inv[i];
        const float* q = pts + (int64_t)i * stride;
        const float x = q[0], y = q[1], z = q[2];
        const float* m = mean + (int64_t)p * 3;
        const int4 c = voxel_coors[p];
        float f[11];
        f[0] = x; f[1] = y; f[2] = z; f[3] = q[3]; f[4] = q[4];
        f[5] = x - m[0]; f[6] = y - m[1]; f[7] = z - m[2];
        // x - (coor * v + offset): every operation rounded separately, as the reference's tensor ops
        f[8] = __fsub_rn(x, __fadd_rn(__fmul_rn((float)c.w, vx), xo));
        f[9] = __fsub_rn(y, __fadd_rn(__fmul_rn((float)c.z, vy), yo));
        f[10] = __fsub_rn(z, __fadd_rn(__fmul_rn((float)c.y, vz), zo));
        float4* o = reinterpret_cast<float4*>(feat + j * 16);
        o[0] = make_float4(f[0], f[1], f[2], f[3]);
        o[1] = make_float4(f[4], f[5], f[6], f[7]);
        o[2] = make_float4(f[8], f[9], f[10], 0.f);
        o[3] = make_float4(0.f, 0.f, 0.f, 0.f);
        pid_out[j] = p;
        if (partials) {
            int e = 0;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                s1[k] += f[k];
#pragma unroll
                for (int l = k; l < 11; ++l) s2[e++] += f[k] * f[l];
            }
        }
    }
    if (!partials) return;
    __shared__ double red[4][80];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        const float v = wave_sum(s1[k]);
        if (lane == 0) red[wave][k] = (double)v;
    }
#pragma unroll
    for (int k = 0; k < 66; ++k) {
        const float v = wave_sum(s2[k]);
        if (lane == 0) red[wave][11 + k] = (double)v;
    }
    __syncthreads();
    if (threadIdx.x < 77)
        partials[(int64_t)blockIdx.x * kMomPad + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] +
                                                                red[3][threadIdx.x];
}

// partial rows -> moments [kMomCount]: one workgroup, sums in a fixed order (deterministic)
__global__ __launch_bounds__(1024) void vfe_moments_reduce_kernel(const double* __restrict__ partials, int rows,
                                                                  double* __restrict__ moments) {
    __shared__ double acc[12][80];
    const int e = threadIdx.x % 80, slice = threadIdx.x / 80;            // 12 slices x 80 entries = 960 threads
    if (slice < 12) {
        double s = 0.0;
        if (e < 77)
            for (int r = slice; r < rows; r += 12) s += partials[(int64_t)r * kMomPad + e];
        acc[slice][e] = s;
    }
    __syncthreads();
    if (threadIdx.x < 77) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 12; ++k) s += acc[k][threadIdx.x];
        acc[0][threadIdx.x] = s;
    }
    __syncthreads();
    // unpack: S1 [16] (5 zeros), S2 [11][11] symmetric from the 66 upper-triangle sums
    if (threadIdx.x < 16) moments[threadIdx.x] = threadIdx.x < 11 ? acc[0][threadIdx.x] : 0.0;
    if (threadIdx.x < 121) {
        const int k = threadIdx.x / 11, l = threadIdx.x % 11;
        const int a = k < l ? k : l, b = k < l ? l : k;
        const int idx = a * 11 - a * (a - 1) / 2 + (b - a);             // position of (a, b), a <= b, in the row-wise triangle
        moments[kMomS2 + threadIdx.x] = acc[0][11 + idx];
    }
}

// sums0 [128] (sum y0, sum y0^2 per channel) from the moments: what vfe_stats0_kernel measures with a sweep
__global__ __launch_bounds__(64) void vfe_stats0_from_moments_kernel(const double* __restrict__ moments,
                                                                     const float* __restrict__ w0, double* __restrict__ sums0) {
    const int c = threadIdx.x;
    double w[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) w[k] = (double)w0[c * 11 + k];
    double m1 = 0.0, m2 = 0.0;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        m1 += w[k] * moments[k];
        double r = 0.0;
#pragma unroll
        for (int l = 0; l < 11; ++l) r += w[l] * moments[kMomS2 + k * 11 + l];
        m2 += w[k] * r;
    }
    sums0[c] = m1;
    sums0[64 + c] = m2;
}

// y0[t][16*ot + 4g + r] = sum_k W0[.][k] f[t][k]; W0s: LDS [64][16] fp32 (columns >= 11 are zero)
__device__ __forceinline__ void layer0_linear(const float* __restrict__ W0s, const float (&f)[4], f32x4 (&y)[4], int lane) {
    const int o = lane & 15, g = lane >> 4;
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) {
        const float4 a = *reinterpret_cast<const float4*>(W0s + (16 * ot + o) * 16 + 4 * g);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = mfma_f32(a.x, f[0], acc);
        acc = mfma_f32(a.y, f[1], acc);
        acc = mfma_f32(a.z, f[2], acc);
        acc = mfma_f32(a.w, f[3], acc);
        y[ot] = acc;
    }
}

// The 128 x 128 layer-1 GEMMs in "bf16 x 3": x = x_hi + x_lo and w = w_hi + w_lo with bf16 halves (x_lo =
// bf16(x - x_hi): 16 mantissa bits together), y = w_hi x_hi + w_lo x_hi + w_hi x_lo accumulated in fp32 -- the dropped
// w_lo x_lo term and the 2^-17 representation errors leave ~2^-16 relative error per product, far inside the fp32-
// accumulation noise of a 128-term sum of O(1) terms times the test tolerance (2e-4).  Three 16x16x32 bf16 MFMAs
// (32 cycles each) replace eight 16x16x4 fp32 MFMAs (32 cycles each) per 32-wide k step: 96 instead of 256 MFMA
// issues per [16 x 128] tile, and these sweeps were MFMA-issue bound (272-528 fp32 MFMAs per tile).
// Both the forward and the backward's recomputation call THIS function on the same operands, so the recomputed
// activations stay bit-identical to the forward's (the max-pool backward routes gradients by equality).
struct BSplit { uint4 h[4], l[4]; };     // B operand of a [16 tokens x 128 channels] T-layout tile: 4 k-steps of 32

__device__ __forceinline__ void split2(float x, float y, unsigned int* hi, unsigned int* lo) {
    const unsigned int h = pack2(x, y);
    *hi = h;
    *lo = pack2(x - bf_lo(h), y - bf_hi(h));
}
__device__ __forceinline__ BSplit split_operand(const f32x4 (&v)[8]) {
    BSplit s;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        split2(v[2 * kk][0], v[2 * kk][1], &s.h[kk].x, &s.l[kk].x);
        split2(v[2 * kk][2], v[2 * kk][3], &s.h[kk].y, &s.l[kk].y);
        split2(v[2 * kk + 1][0], v[2 * kk + 1][1], &s.h[kk].z, &s.l[kk].z);
        split2(v[2 * kk + 1][2], v[2 * kk + 1][3], &s.h[kk].w, &s.l[kk].w);
    }
    return s;
}

// y1[t][16*ot + 4g + r] = sum_k W1[.][k] gin[t][k]; W1s: the split, K-permuted LDS copy written by stage_w1.
// NG output tiles starting at ot0 advance together over k (independent accumulators back to back).  The order of
// the terms inside every accumulator is fixed (k step, then hi*hi, lo*hi, hi*lo), so any grouping gives
// bit-identical results.
// PLAIN (GeomaeVfeArgs.layer1_bf16, the bf16 compute mode of the step): the hi * hi product alone -- one MFMA per k step instead of
// three, half the LDS weight reads; the operands are then plain bf16 roundings of g / dy1 and W1 (2^-9 relative per term, the
// grade of the SST layers' GEMMs), still the same function in every sweep: the recomputed activations stay bit-identical.
template <int NG, bool PLAIN = false>
__device__ __forceinline__ void layer1_group(const float* __restrict__ W1s, const BSplit& x, int ot0, f32x4 (&y)[NG],
                                             int lane) {
    const int o = lane & 15, g = lane >> 4;
    const bf16_t* wh = reinterpret_cast<const bf16_t*>(W1s) + (16 * ot0 + o) * kW1bLd + 8 * g;
    const bf16_t* wl = wh + 128 * kW1bLd;
#pragma unroll
    for (int u = 0; u < NG; ++u) y[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        uint4 ah[NG], al[NG];
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            ah[u] = *reinterpret_cast<const uint4*>(wh + 16 * u * kW1bLd + 32 * kk);
            if (!PLAIN) al[u] = *reinterpret_cast<const uint4*>(wl + 16 * u * kW1bLd + 32 * kk);
        }
#pragma unroll
        for (int u = 0; u < NG; ++u) y[u] = mfma32(ah[u], x.h[kk], y[u]);
        if (!PLAIN) {
#pragma unroll
            for (int u = 0; u < NG; ++u) y[u] = mfma32(al[u], x.h[kk], y[u]);
#pragma unroll
            for (int u = 0; u < NG; ++u) y[u] = mfma32(ah[u], x.l[kk], y[u]);
        }
    }
}
// The transposed product from the SAME LDS copy: dg[t][16*ot + 4g + r] = sum_o W1[o][.] dy1[t][o].  The contraction now
// runs over W1's ROW index, so the A fragment of lane (m = lane & 15, g) -- eight o's of column k = 16 ot + m, in the
// K order of the T-layout operand: o = 32 kk + {4g .. 4g+3, 16+4g .. 16+4g+3} -- is two groups of four consecutive
// rows of one column: what ds_read_b64_tr_b16 returns when lane m points at row (m >> 2) and at the 4-column chunk
// (m & 3) of the tile (chunk q of a 32-column block sits at position 8 (q & 3) + 4 (q >> 2) of the K-permuted row).
// Same products in the same order as a GEMM against a transposed copy, without the copy (a second 69.6 KB that does not
// fit beside the first): the backward sweep no longer parks dy1 in HBM between a W1 pass and a W1^T pass.
template <int NG, bool PLAIN = false>
__device__ __forceinline__ void layer1_group_t(const float* __restrict__ W1s, const BSplit& x, int ot0, f32x4 (&y)[NG],
                                               int lane) {
    const int m = lane & 15, g = lane >> 4;
    const bf16_t* wh = reinterpret_cast<const bf16_t*>(W1s) + (4 * g + (m >> 2)) * kW1bLd + 8 * (m & 3);
    const bf16_t* wl = wh + 128 * kW1bLd;
#pragma unroll
    for (int u = 0; u < NG; ++u) y[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        uint4 ah[NG], al[NG];
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const int ot = ot0 + u;
            const int col = 32 * (ot >> 1) + 4 * (ot & 1);
            const bf16_t* ph = wh + 32 * kk * kW1bLd + col;
            const bf16_t* pl = wl + 32 * kk * kW1bLd + col;
            const uint2 h0 = tr_read(ph), h1 = tr_read(ph + 16 * kW1bLd);
            ah[u] = make_uint4(h0.x, h0.y, h1.x, h1.y);
            if (!PLAIN) {
                const uint2 l0 = tr_read(pl), l1 = tr_read(pl + 16 * kW1bLd);
                al[u] = make_uint4(l0.x, l0.y, l1.x, l1.y);
            }
        }
#pragma unroll
        for (int u = 0; u < NG; ++u) y[u] = mfma32(ah[u], x.h[kk], y[u]);
        if (!PLAIN) {
#pragma unroll
            for (int u = 0; u < NG; ++u) y[u] = mfma32(al[u], x.h[kk], y[u]);
#pragma unroll
            for (int u = 0; u < NG; ++u) y[u] = mfma32(ah[u], x.l[kk], y[u]);
        }
    }
}
template <bool PLAIN>
__device__ __forceinline__ void layer1_linear(const float* __restrict__ W1s, const BSplit& x, f32x4 (&y)[8], int lane) {
    layer1_group<4, PLAIN>(W1s, x, 0, reinterpret_cast<f32x4(&)[4]>(y[0]), lane);
    layer1_group<4, PLAIN>(W1s, x, 4, reinterpret_cast<f32x4(&)[4]>(y[4]), lane);
}

template <int NT>
__device__ __forceinline__ void bn_relu(const f32x4 (&y)[NT], const float* __restrict__ scale,
                                        const float* __restrict__ shift, f32x4 (&h)[NT], int lane) {
    const int g = lane >> 4;
#pragma unroll
    for (int ot = 0; ot < NT; ++ot) {
        const float4 s = *reinterpret_cast<const float4*>(scale + 16 * ot + 4 * g);
        const float4 b = *reinterpret_cast<const float4*>(shift + 16 * ot + 4 * g);
        h[ot][0] = fmaxf(y[ot][0] * s.x + b.x, 0.f);
        h[ot][1] = fmaxf(y[ot][1] * s.y + b.y, 0.f);
        h[ot][2] = fmaxf(y[ot][2] * s.z + b.z, 0.f);
        h[ot][3] = fmaxf(y[ot][3] * s.w + b.w, 0.f);
    }
}

template <int NT>
__device__ __forceinline__ void bn_affine(const f32x4 (&y)[NT], const float* __restrict__ scale,
                                          const float* __restrict__ shift, f32x4 (&h)[NT], int lane) {
    const int g = lane >> 4;
#pragma unroll
    for (int ot = 0; ot < NT; ++ot) {
        const float4 s = *reinterpret_cast<const float4*>(scale + 16 * ot + 4 * g);
        const float4 b = *reinterpret_cast<const float4*>(shift + 16 * ot + 4 * g);
        h[ot][0] = y[ot][0] * s.x + b.x;
        h[ot][1] = y[ot][1] * s.y + b.y;
        h[ot][2] = y[ot][2] * s.z + b.z;
        h[ot][3] = y[ot][3] * s.w + b.w;
    }
}

__device__ __forceinline__ void stage_w0(const float* __restrict__ w0, float* W0s) {
    for (int e = threadIdx.x; e < 64 * 16; e += kVfeBlk) {
        const int r = e >> 4, c = e & 15;
        W0s[e] = c < 11 ? w0[r * 11 + c] : 0.f;
    }
}
// W1 [128 out][128 in] fp32 -> LDS, split into bf16 hi / lo halves, K-permuted for the MFMA operand order of the
// T-layout (sst_device.h kperm): dst[o][p] = W1[o][kperm(p)], or with `transpose` (dg = dy1 W1: outputs are the
// input channels) dst[k][p] = W1[kperm(p)][k].  kperm^-1(32 b + 16 h + 4 q + e) = 32 b + 8 q + 4 h + e: four
// consecutive source columns stay consecutive.
__device__ __forceinline__ void stage_w1(const float* __restrict__ w1, float* W1s, bool transpose, bool with_lo = true) {
    bf16_t* wh = reinterpret_cast<bf16_t*>(W1s);
    bf16_t* wl = wh + 128 * kW1bLd;
    for (int e = threadIdx.x; e < 128 * 32; e += kVfeBlk) {
        const int r = e >> 5, c4 = (e & 31) * 4;
        const float4 v = *reinterpret_cast<const float4*>(w1 + r * 128 + c4);
        unsigned int h01, l01, h23, l23;
        split2(v.x, v.y, &h01, &l01);
        split2(v.z, v.w, &h23, &l23);
        if (!transpose) {
            const int p = (c4 & ~31) + 8 * ((c4 >> 2) & 3) + 4 * ((c4 >> 4) & 1);
            *reinterpret_cast<uint2*>(wh + r * kW1bLd + p) = make_uint2(h01, h23);
            if (with_lo) *reinterpret_cast<uint2*>(wl + r * kW1bLd + p) = make_uint2(l01, l23);
        } else {
            const int p = (r & ~31) + 8 * ((r >> 2) & 3) + 4 * ((r >> 4) & 1) + (r & 3);
            wh[(c4 + 0) * kW1bLd + p] = (bf16_t)(h01 & 0xffffu); wh[(c4 + 1) * kW1bLd + p] = (bf16_t)(h01 >> 16);
            wh[(c4 + 2) * kW1bLd + p] = (bf16_t)(h23 & 0xffffu); wh[(c4 + 3) * kW1bLd + p] = (bf16_t)(h23 >> 16);
            wl[(c4 + 0) * kW1bLd + p] = (bf16_t)(l01 & 0xffffu); wl[(c4 + 1) * kW1bLd + p] = (bf16_t)(l01 >> 16);
            wl[(c4 + 2) * kW1bLd + p] = (bf16_t)(l23 & 0xffffu); wl[(c4 + 3) * kW1bLd + p] = (bf16_t)(l23 >> 16);
        }
    }
}

// BatchNorm vectors are read per tile by every lane: keep them in LDS (one VGPR address + immediate offsets
// instead of a 64-bit global pointer per vector per tile, which is what spilled vfe_bwd_stats1_kernel)
__device__ __forceinline__ void stage_vec(const float* __restrict__ src, float* dst, int n) {
    for (int c = threadIdx.x; c < n; c += kVfeBlk) dst[c] = src[c];
}
#define VFE_STAGE_BN0_FWD(W, Wl)                                   \
    __shared__ __attribute__((aligned(16))) float bn0f_s[2][64];   \
    stage_vec((W).scale0, bn0f_s[0], 64);                          \
    stage_vec((W).shift0, bn0f_s[1], 64);                          \
    VfeW Wl = (W);                                                 \
    Wl.scale0 = bn0f_s[0];                                         \
    Wl.shift0 = bn0f_s[1];
#define VFE_STAGE_BN1(bn, bnl)                                     \
    __shared__ __attribute__((aligned(16))) float bn1_s[4][128];   \
    stage_vec((bn).scale, bn1_s[0], 128);                          \
    stage_vec((bn).shift, bn1_s[1], 128);                          \
    stage_vec((bn).mean, bn1_s[2], 128);                           \
    stage_vec((bn).invstd, bn1_s[3], 128);                         \
    const Bn1 bnl = {bn1_s[0], bn1_s[1], bn1_s[2], bn1_s[3]};
#define VFE_STAGE_BN0(bn, bnl)                                     \
    __shared__ __attribute__((aligned(16))) float bn0_s[4][64];    \
    stage_vec((bn).scale, bn0_s[0], 64);                           \
    stage_vec((bn).shift, bn0_s[1], 64);                           \
    stage_vec((bn).mean, bn0_s[2], 64);                            \
    stage_vec((bn).invstd, bn0_s[3], 64);                          \
    const Bn0 bnl = {bn0_s[0], bn0_s[1], bn0_s[2], bn0_s[3]};

// per-wave tile helpers: T-layout registers -> LDS tile[t][c] (fp32)
template <int NT, int LD>
__device__ __forceinline__ void tile_store(float* tile, const f32x4 (&v)[NT], int lane) {
    const int t = lane & 15, g = lane >> 4;
#pragma unroll
    for (int ot = 0; ot < NT; ++ot)
        *reinterpret_cast<float4*>(tile + t * LD + 16 * ot + 4 * g) = make_float4(v[ot][0], v[ot][1], v[ot][2], v[ot][3]);
}
// LDS hand-off inside ONE wave (tile written by its lanes, scanned by its lanes): DS operations of a wave
// execute in order, so it is enough to drain the LDS counter and stop the compiler from moving accesses
// across.  (A workgroup-scope fence here also waited for the wave's outstanding global stores: ~1 us per tile.)
__device__ __forceinline__ void wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
// global hand-off inside one wave (rows stored in pass A, read back in pass B)
__device__ __forceinline__ void wave_global_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// Segmented reduction of a [16 x 64*CPL] tile over the tile's points, lane = channel (+64 per extra channel).
// The carry holds the open pillar's partial result across tiles; a finished pillar's row goes to out[pid][c]: a plain
// store when the pillar lies inside the wave's point range, an atomic combine when it straddles the range (its
// other parts belong to neighbouring waves; `out` is zero-filled).  Max rows are >= 0 (post-ReLU), so the
// integer atomicMax on the float bits is exact and order independent (and 0 is the neutral start of a run); sums use
// float atomicAdd (a pillar cut in more than two parts may round differently from run to run in the last bit).
// Everything that steers the walk over the 16 points is WAVE-UNIFORM and lives in SGPRs: the run ends are a ballot of
// (pillar of point j + 1 differs), a finished run's pillar id a v_readlane, the branches scalar.  (Until round 4 every
// step read the point's pillar id back from LDS and branched on it as a vector value: 16 dependent LDS round trips
// per tile, 5.5 k cycles = 45 % of the layer-1 forward sweep (tools/vfe_time.py).  The segmented scan in registers --
// T-layout rows are DPP rows: row_shr 1 / 2 / 4 / 8 under a same-pillar mask -- was measured too: 480 VALU
// instructions per tile instead of 32, 6 k cycles.)
// fmaxf of two values that may come straight from memory compiles to THREE instructions (llvm.maxnum canonicalises both
// inputs: v_max x, x, x twice); the instruction itself already returns the other operand for a NaN
__device__ __forceinline__ float vmax_f32(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int CPL>
struct SegCarry {
    float cur[CPL];
    int first_pid, last_pid;         // the pillars of the wave's first / last point: they may continue in a neighbour's range
    bool first_shared, last_shared;
    unsigned long long tie;          // (TRACK) the open run: lanes in whose channels two points shared the running maximum
    // (TRACK) the wave's parts of its (at most two) straddling pillars, combined by seg_resolve at the END of the sweep with
    // atomics that RETURN what was stored (a part that finds its own maximum there: a tie across waves); issued and
    // compared inside the walk each of them cost the wave a memory round trip
    float pend_val[2][CPL];
    int pend_pid[2];
    __device__ __forceinline__ void init(const VfeGeo& G, const WaveRange& R, float start = 0.f) {
        tie = 0ull;
        pend_pid[0] = pend_pid[1] = -1;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            cur[k] = start;
            pend_val[0][k] = pend_val[1][k] = 0.f;
        }
        first_pid = last_pid = -1;
        first_shared = last_shared = false;
        if (R.j_lo < R.j_hi) {
            first_pid = __builtin_amdgcn_readfirstlane(G.pid[R.j_lo]);
            last_pid = __builtin_amdgcn_readfirstlane(G.pid[R.j_hi - 1]);
            first_shared = __builtin_amdgcn_readfirstlane(G.seg_start[first_pid]) < R.j_lo;
            last_shared = __builtin_amdgcn_readfirstlane(G.seg_start[last_pid + 1]) > R.j_hi;
        }
    }
};
// the pillar of point j + 1, or -1 behind the wave's last point (loaded beside the point's own pillar id)
__device__ __forceinline__ int pillar_next(const VfeGeo& G, int j, const WaveRange& R) { return j + 1 < R.j_hi ? G.pid[j + 1] : -1; }

// ties (max only, or null): ties[p] = 1 for every pillar whose maximum MAY be held by more than one point in some channel
// -- conservatively: any equality with the RUNNING maximum of the run counts, and so does a part of a straddling pillar
// that finds its own maximum already stored by another part.  A pillar that stays 0 has exactly one arg-max per channel.
// With TRACK the tile holds the values BEFORE the ReLU clamp (max and clamp commute) and a run starts at -inf: clamped
// zeros would be equal to each other all the time, unclamped values coincide as rarely as positive ones -- one compare
// per value instead of two, and the clamp moves from every value to every finished run.
template <int CPL, bool IS_MAX, int LD, bool TRACK = false>
__device__ __forceinline__ void seg_scan(const float* tile, int pid, int pid_next, bool valid, int C, float* __restrict__ out,
                                         SegCarry<CPL>& c, int lane, unsigned char* __restrict__ ties = nullptr) {
    const unsigned int live = (unsigned int)__ballot(valid) & 0xffffu;                       // lanes 0..15 hold the tile's points
    const unsigned int ends = (unsigned int)__ballot(valid && pid_next != pid) & 0xffffu;    // ... the last point of a pillar's run
    float x[16][CPL];
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int k = 0; k < CPL; ++k) x[t][k] = tile[t * LD + lane + 64 * k];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        if (!((live >> t) & 1u)) break;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            if (TRACK) c.tie |= __ballot(x[t][k] == c.cur[k]);        // (a lane mask in SGPRs: v_cmp + s_or)
            c.cur[k] = IS_MAX ? vmax_f32(c.cur[k], x[t][k]) : c.cur[k] + x[t][k];
        }
        if ((ends >> t) & 1u) {
            const int p = __builtin_amdgcn_readlane(pid, t);
            const bool shared = (p == c.first_pid && c.first_shared) || (p == c.last_pid && c.last_shared);
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                float* dst = out + (int64_t)p * C + lane + 64 * k;
                const float v = TRACK ? fmaxf(c.cur[k], 0.f) : c.cur[k];
                if (!shared) *dst = v;
                else if (IS_MAX && TRACK) {
                    if (p == c.first_pid && c.first_shared) { c.pend_val[0][k] = v; c.pend_pid[0] = p; }
                    else { c.pend_val[1][k] = v; c.pend_pid[1] = p; }
                }
                else if (IS_MAX) atomicMax(reinterpret_cast<int*>(dst), __float_as_int(v));
                else atomicAdd(dst, v);
                c.cur[k] = TRACK ? -INFINITY : 0.f;
            }
            if (TRACK) {
                if (ties && c.tie != 0ull && lane == 0) ties[p] = 1;
                c.tie = 0ull;
            }
        }
    }
}

// (TRACK) the straddling pillars' parts: combine, and mark the pillar if a part finds its own maximum (> 0) already stored
template <int CPL>
__device__ __forceinline__ void seg_resolve(const SegCarry<CPL>& c, float* __restrict__ out, int C,
                                            unsigned char* __restrict__ ties, int lane) {
    int old[2][CPL];
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
        if (c.pend_pid[sl] >= 0) {                                    // wave-uniform
#pragma unroll
            for (int k = 0; k < CPL; ++k)
                old[sl][k] = atomicMax(reinterpret_cast<int*>(out + (int64_t)c.pend_pid[sl] * C + lane + 64 * k),
                                       __float_as_int(c.pend_val[sl][k]));
        }
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
        if (c.pend_pid[sl] >= 0) {
            bool eq = false;
#pragma unroll
            for (int k = 0; k < CPL; ++k) eq |= old[sl][k] == __float_as_int(c.pend_val[sl][k]) && c.pend_val[sl][k] > 0.f;
            if (ties && __any(eq) && lane == 0) ties[c.pend_pid[sl]] = 1;
        }
}

// End of a statistics sweep: the lanes' partial sums s1 / s2 (T-layout: lane = point slot t of every tile the wave swept,
// lane group g = 4 channels of every 16) -> per-channel sums over all points, added to out[2C] in fp64.
// The sum over the 16 point lanes goes through LDS: the wave stores a tensor token-major into its slice of `scratch`
// (NT x ds_write_b128), every lane then sums one channel column over the 16 rows; the waves' columns meet in `red`
// and one thread per channel adds the workgroup's sum with a single fp64 atomic.  (In registers the same reduction is
// 4 DPP steps for each of the 8 NT values of a lane plus a leader-lane store per value: ~800 instructions, 10.9 k
// cycles at the end of vfe_bwd_stats1_kernel = 17 % of it, tools/vfe_time.py.)
// scratch: LDS, kVfeWaves * 16 * (16 NT + 4) floats, free at the call (the caller's weight / tile buffer); the function
// starts with the workgroup barrier that makes it so.
template <int NT>
__device__ __forceinline__ void flush_channel_sums(const f32x4 (&s1)[NT], const f32x4 (&s2)[NT], double* __restrict__ out, int C,
                                                   float* red /* LDS [waves][2*C] */, float* scratch, int lane, int wave) {
    constexpr int LD = 16 * NT + 4;
    float* mine = scratch + wave * 16 * LD;
    __syncthreads();                                                  // every wave is done with the buffer behind `scratch`
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        tile_store<NT, LD>(mine, half == 0 ? s1 : s2, lane);
        wave_sync();
#pragma unroll
        for (int k = 0; k < (16 * NT + 63) / 64; ++k) {
            const int c = lane + 64 * k;
            if (c < 16 * NT) {
                float sum = 0.f;
#pragma unroll
                for (int t = 0; t < 16; ++t) sum += mine[t * LD + c];
                red[wave * 2 * C + half * C + c] = sum;
            }
        }
        wave_sync();
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * C; e += kVfeBlk) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < kVfeWaves; ++w) s += red[w * 2 * C + e];
        atomicAdd(out + e, (double)s);
    }
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vfe_mean_accum_kernel(const float* __restrict__ pts, int stride, int64_t n,
                                                             const int32_t* __restrict__ inv,
                                                             unsigned long long* __restrict__ sum64) {
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float* q = pts + i * stride;
        unsigned long long* s = sum64 + (int64_t)inv[i] * 3;
        atomicAdd(s + 0, (unsigned long long)__double2ll_rn((double)q[0] * 4294967296.0));
        atomicAdd(s + 1, (unsigned long long)__double2ll_rn((double)q[1] * 4294967296.0));
        atomicAdd(s + 2, (unsigned long long)__double2ll_rn((double)q[2] * 4294967296.0));
    }
}
__global__ __launch_bounds__(256) void vfe_mean_final_kernel(const unsigned long long* __restrict__ sum64,
                                                             const int32_t* __restrict__ seg_start,
                                                             const int32_t* __restrict__ num_pillars,
                                                             float* __restrict__ mean) {
    const int V = num_pillars[0];
    for (int e = blockIdx.x * 256 + threadIdx.x; e < V * 3; e += gridDim.x * 256) {
        const int p = e / 3;
        const double inv_n = 1.0 / (4294967296.0 * (double)(seg_start[p + 1] - seg_start[p]));
        mean[e] = (float)((double)(long long)sum64[e] * inv_n);
    }
}

// The same means from the pillar-sorted point list (order / seg_start of geomae_pillar_segment): 16 lanes per pillar
// add the 2^-32 fixed-point coordinates of its points and combine by shuffles -- no atomics, no workspace, no second
// kernel.  Integer sums are order independent, so the result is bit-identical to the atomic version (24 + 5 us there,
// 312 k 64-bit atomics; it runs beside the encoder forward as part of the next batch's stage 1).
__global__ __launch_bounds__(256) void vfe_mean_sorted_kernel(const float* __restrict__ pts, int stride,
                                                              const int32_t* __restrict__ order,
                                                              const int32_t* __restrict__ seg_start,
                                                              const int32_t* __restrict__ num_pillars,
                                                              float* __restrict__ mean) {
    const int V = num_pillars[0];
    const int sub = threadIdx.x & 15;
    for (int g0 = blockIdx.x * 16; g0 < V; g0 += gridDim.x * 16) {      // 16 pillars per workgroup and pass
        const int p = g0 + (threadIdx.x >> 4);
        const bool valid = p < V;
        const int s = valid ? seg_start[p] : 0, e = valid ? seg_start[p + 1] : 0;
        long long a0 = 0, a1 = 0, a2 = 0;
        for (int j = s + sub; j < e; j += 16) {
            const float* q = pts + (int64_t)order[j] * stride;
            a0 += __double2ll_rn((double)q[0] * 4294967296.0);
            a1 += __double2ll_rn((double)q[1] * 4294967296.0);
            a2 += __double2ll_rn((double)q[2] * 4294967296.0);
        }
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) {
            a0 += __shfl_xor(a0, m);
            a1 += __shfl_xor(a1, m);
            a2 += __shfl_xor(a2, m);
        }
        if (valid && sub == 0) {
            const double inv_n = 1.0 / (4294967296.0 * (double)(e - s));
            mean[(int64_t)p * 3 + 0] = (float)((double)a0 * inv_n);
            mean[(int64_t)p * 3 + 1] = (float)((double)a1 * inv_n);
            mean[(int64_t)p * 3 + 2] = (float)((double)a2 * inv_n);
        }
    }
}

// BatchNorm bookkeeping, one workgroup.  moments (mean, mean of squares) come either from the local fp64
// sums (single process) or from the caller (after the cross-rank average of naiveSyncBN1d).
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ sums, double count,
                                                          const float* __restrict__ moments_in, int C,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, float momentum, int unbiased_running,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          float* __restrict__ scale, float* __restrict__ shift,
                                                          float* __restrict__ invstd_out, float* __restrict__ moments_out,
                                                          long long* __restrict__ num_batches_tracked) {
    if (num_batches_tracked && threadIdx.x == 0) num_batches_tracked[0] += 1;
    for (int c = threadIdx.x; c < C; c += 256) {
        float mean, msq;
        if (moments_in) { mean = moments_in[c]; msq = moments_in[C + c]; }
        else { mean = (float)(sums[c] / count); msq = (float)(sums[C + c] / count); }
        if (moments_out) { moments_out[c] = mean; moments_out[C + c] = msq; }
        if (scale) {
            float var = msq - mean * mean;
            if (!moments_in) {   // single process: variance from fp64 sums (== nn.BatchNorm1d's two-pass value)
                const double m = sums[c] / count;
                var = (float)(sums[C + c] / count - m * m);
            }
            const float invstd = rsqrtf(var + eps);
            const float s = gamma[c] * invstd;
            scale[c] = s;
            shift[c] = beta[c] - mean * s;
            invstd_out[c] = invstd;
            if (running_mean) {
                const float rv = unbiased_running ? var * (float)(count / (count - 1.0)) : var;
                running_mean[c] += momentum * (mean - running_mean[c]);
                running_var[c] += momentum * (rv - running_var[c]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
struct VfeW {
    const float *w0, *w1;                 // [64,11], [128,128]
    const float *scale0, *shift0, *scale1, *shift1;   // BN folded: y * scale + shift
};

__device__ __forceinline__ VfeW shifted(const VfeW& W, int z) {
    VfeW o = W;
    o.scale0 += z; o.shift0 += z; o.scale1 += z; o.shift1 += z;
    return o;
}
struct Bn1 { const float *scale, *shift, *mean, *invstd; };
struct Bn0 { const float *scale, *shift, *mean, *invstd; };
__device__ __forceinline__ Bn1 shifted(const Bn1& b, int z) { return Bn1{b.scale + z, b.shift + z, b.mean + z, b.invstd + z}; }
__device__ __forceinline__ Bn0 shifted(const Bn0& b, int z) { return Bn0{b.scale + z, b.shift + z, b.mean + z, b.invstd + z}; }

// phase timing of the point loops (timing build only): wave 0 of the last 512 workgroups accumulates the cycles between
// the marks of every tile; slots: 0 staging, 1 + k = from mark k to mark k + 1, 7 = whole kernel
#ifdef GEOMAE_PHASE_TIMING
#define VFE_T_ENTRY() const unsigned long long vt_entry = clock64()
#define VFE_T_BEGIN() unsigned long long vt_acc[6] = {0, 0, 0, 0, 0, 0}; unsigned long long vt_last = clock64(); const unsigned long long vt_begin = vt_last
#define VFE_T(k) do { const unsigned long long c_ = clock64(); if ((k) > 0) vt_acc[(k) - 1] += c_ - vt_last; vt_last = c_; } while (0)
#define VFE_T_END() VFE_T_END_AT(0)
#define VFE_T_END_AT(base_)                                                                                    \
    do {                                                                                                       \
        if (threadIdx.x == 0 && blockIdx.x + GEOMAE_STAMP_BLOCKS >= gridDim.x) {                               \
            unsigned long long* o_ = geomae_stamps + (blockIdx.x % GEOMAE_STAMP_BLOCKS) * GEOMAE_STAMP_SLOTS + (base_);  \
            o_[0] = vt_begin - vt_entry;                                                                       \
            for (int q_ = 0; q_ < 6; ++q_) o_[1 + q_] = vt_acc[q_];                                            \
            o_[7] = clock64() - vt_entry;                                                                      \
        }                                                                                                      \
    } while (0)
#else
#define VFE_T_ENTRY() do {} while (0)
#define VFE_T_BEGIN() do {} while (0)
#define VFE_T(k) do {} while (0)
#define VFE_T_END() do {} while (0)
#define VFE_T_END_AT(base_) do {} while (0)
#endif

// sweep 1 of layer 0: per-channel sum / sum of squares of y0 = W0 f over all points
__global__ __launch_bounds__(kVfeBlk) void vfe_stats0_kernel(VfeGeo G, VfeW W, double* __restrict__ sums0) {
    __shared__ float W0s[64 * 16];
    __shared__ float red[kVfeWaves * 2 * 64];
    __shared__ __attribute__((aligned(16))) float fl_s[kVfeWaves * 16 * 68];
    stage_w0(W.w0, W0s);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const WaveRange R = wave_range(G, blockIdx.x * kVfeWaves + wave);
    f32x4 s1[4], s2[4];
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) { s1[ot] = f32x4{0, 0, 0, 0}; s2[ot] = f32x4{0, 0, 0, 0}; }
    for (int j0 = R.j_lo; j0 < R.j_hi; j0 += 16) {
        const int j = j0 + (lane & 15);
        const bool valid = j < R.j_hi;
        float f[4];
        build_features(G, j, valid, g, f);
        f32x4 y[4];
        layer0_linear(W0s, f, y, lane);
        if (valid) {
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) { s1[ot] += y[ot]; s2[ot] += y[ot] * y[ot]; }
        }
    }
    flush_channel_sums<4>(s1, s2, sums0, 64, red, fl_s, lane, wave);
}

// BatchNorm finalisation folded into the head of a sweep (GeomaeBnFold): channel c's (sum, sum of squares) -> scale / shift
// into LDS for this workgroup; workgroup 0 also writes what bn_finalize_kernel writes.  Same arithmetic as that kernel
// (variance from the fp64 sums, unbiased running variance).
struct BnFoldDev {
    double count;
    const float *gamma, *beta;
    float eps, momentum;
    float *running_mean, *running_var, *scale, *shift, *invstd, *moments;
    long long* nbt;
    bool on;
};
__device__ __forceinline__ void bn_fold_channel(const BnFoldDev& F, int C, int c, double s1, double s2, float* scale_lds,
                                                float* shift_lds) {
    const double m = s1 / F.count;
    const float mean = (float)m, msq = (float)(s2 / F.count);
    const float var = (float)(s2 / F.count - m * m);
    const float invstd = rsqrtf(var + F.eps);
    const float sc = F.gamma[c] * invstd, sh = F.beta[c] - mean * sc;
    scale_lds[c] = sc;
    shift_lds[c] = sh;
    if (blockIdx.x == 0) {
        F.scale[c] = sc; F.shift[c] = sh; F.invstd[c] = invstd;
        F.moments[c] = mean; F.moments[C + c] = msq;
        if (F.running_mean) {
            const float rv = var * (float)(F.count / (F.count - 1.0));
            F.running_mean[c] += F.momentum * (mean - F.running_mean[c]);
            F.running_var[c] += F.momentum * (rv - F.running_var[c]);
        }
        if (F.nbt && c == 0) F.nbt[0] += 1;
    }
}

// sweep 2 of layer 0: h0 = ReLU(BN(y0)), m0 = segmented max (m0 zero-filled by the caller: rows of pillars
// that straddle waves are combined with integer atomicMax, exact because h0 >= 0)
__global__ __launch_bounds__(kVfeBlk) void vfe_layer0_kernel(VfeGeo G, VfeW W, float* __restrict__ m0, BnFoldDev F,
                                                             const double* __restrict__ moments) {
    __shared__ float W0s[64 * 16];
    __shared__ __attribute__((aligned(16))) float tiles[kVfeWaves][16 * kTile0Ld];
    stage_w0(W.w0, W0s);
    VFE_STAGE_BN0_FWD(W, Wl)
    if (F.on && threadIdx.x < 64) {
        // y0 = W0 f is linear: sum y0_c = W0[c] . S1, sum y0_c^2 = W0[c] S2 W0[c]^T (vfe_stats0_from_moments_kernel)
        const int c = threadIdx.x;
        double w[11];
#pragma unroll
        for (int k = 0; k < 11; ++k) w[k] = (double)W.w0[c * 11 + k];
        double m1 = 0.0, m2 = 0.0;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            m1 += w[k] * moments[k];
            double r = 0.0;
#pragma unroll
            for (int l = 0; l < 11; ++l) r += w[l] * moments[kMomS2 + k * 11 + l];
            m2 += w[k] * r;
        }
        bn_fold_channel(F, 64, c, m1, m2, bn0f_s[0], bn0f_s[1]);     // (overwrites what VFE_STAGE_BN0_FWD staged)
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const WaveRange R = wave_range(G, blockIdx.x * kVfeWaves + wave);
    float* tile = tiles[wave];
    SegCarry<1> carry;
    carry.init(G, R);
    for (int j0 = R.j_lo; j0 < R.j_hi; j0 += 16) {
        const int j = j0 + (lane & 15);
        const bool valid = j < R.j_hi;
        const int pid = valid ? pillar_of(G, j) : 0;
        const int pid_next = valid ? pillar_next(G, j, R) : -1;
        float f[4];
        build_features(G, j, valid, g, f);
        f32x4 y[4], h[4];
        layer0_linear(W0s, f, y, lane);
        bn_relu<4>(y, Wl.scale0, Wl.shift0, h, lane);
        tile_store<4, kTile0Ld>(tile, h, lane);
        wave_sync();
        seg_scan<1, true, kTile0Ld>(tile, pid, pid_next, valid, 64, m0, carry, lane);
        wave_sync();
    }
}

// recompute h0 and g = [h0 | m0[pid]] for the lane's point (shared by every later sweep so that the values
// are bit-identical to the forward's)
__device__ __forceinline__ void recompute_g(const VfeGeo& G, const VfeW& W, const float* W0s, const float* __restrict__ m0,
                                            int j, int pid, bool valid, int lane, f32x4 (&y0)[4], f32x4 (&gin)[8]) {
    const int g = lane >> 4;
    float f[4];
    build_features(G, j, valid, g, f);
    layer0_linear(W0s, f, y0, lane);
    bn_relu<4>(y0, W.scale0, W.shift0, reinterpret_cast<f32x4(&)[4]>(gin), lane);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        float4 v = make_float4(0, 0, 0, 0);
        if (valid) v = *reinterpret_cast<const float4*>(m0 + (int64_t)pid * 64 + 16 * ct + 4 * g);
        gin[4 + ct] = f32x4{v.x, v.y, v.z, v.w};
    }
}

// The same from inputs that were loaded a tile AHEAD: a wave walks its 4 tiles one after the other, and with the loads at the
// top of each iteration every tile began with a memory round trip (features / pillar id, then the dependent gather of the
// pillar's layer-0 maxima).  The ids and features of tile k + 1 are requested at the top of iteration k, its gather in the
// middle of it.  (Worth 1.4 k cycles of the phase it removes, of which the layer-1 forward sweep keeps nothing -- its
// second wave per SIMD already covered that wait -- and the statistics sweep ~6 %: 21.7 -> 20.4 us at config 2.)
struct TileIn {
    float f[4];
    float4 m[4];                  // m0[pid][16 ct + 4 g ..]
    int pid, pid_next;
    bool valid;
};
__device__ __forceinline__ void tile_ids_issue(const VfeGeo& G, const WaveRange& R, int j0, int lane, bool want_next, TileIn& t) {
    const int j = j0 + (lane & 15);
    t.valid = j < R.j_hi;
    t.pid = t.valid ? pillar_of(G, j) : 0;
    t.pid_next = want_next && t.valid ? pillar_next(G, j, R) : -1;
    build_features(G, j, t.valid, lane >> 4, t.f);
}
__device__ __forceinline__ void tile_m0_issue(const float* __restrict__ m0, int lane, TileIn& t) {
    const int g = lane >> 4;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        t.m[ct] = make_float4(0, 0, 0, 0);
        if (t.valid) t.m[ct] = *reinterpret_cast<const float4*>(m0 + (int64_t)t.pid * 64 + 16 * ct + 4 * g);
    }
}
__device__ __forceinline__ void recompute_g_from(const VfeW& W, const float* W0s, const TileIn& t, int lane, f32x4 (&y0)[4],
                                                 f32x4 (&gin)[8]) {
    layer0_linear(W0s, t.f, y0, lane);
    bn_relu<4>(y0, W.scale0, W.shift0, reinterpret_cast<f32x4(&)[4]>(gin), lane);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) gin[4 + ct] = f32x4{t.m[ct].x, t.m[ct].y, t.m[ct].z, t.m[ct].w};
}

// sweep 1 of layer 1: statistics of y1 = W1 [h0 | m0]
template <bool PLAIN>
__global__ __launch_bounds__(kVfeBlk) void vfe_stats1_kernel(VfeGeo G, VfeW W, const float* __restrict__ m0,
                                                             double* __restrict__ sums1) {
    __shared__ float W0s[64 * 16];
    __shared__ __attribute__((aligned(16))) float W1s[128 * kW1Ld];
    __shared__ float red[kVfeWaves * 2 * 128];
    stage_w0(W.w0, W0s);
    stage_w1(W.w1, W1s, false, !PLAIN);
    VFE_STAGE_BN0_FWD(W, Ws)
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const WaveRange R = wave_range(G, blockIdx.x * kVfeWaves + wave);
    f32x4 s1[8], s2[8];
#pragma unroll
    for (int ot = 0; ot < 8; ++ot) { s1[ot] = f32x4{0, 0, 0, 0}; s2[ot] = f32x4{0, 0, 0, 0}; }
    TileIn cur, nxt;
    tile_ids_issue(G, R, R.j_lo, lane, false, cur);
    tile_m0_issue(m0, lane, cur);
    for (int j0 = R.j_lo; j0 < R.j_hi; j0 += 16) {
        const bool more = j0 + 16 < R.j_hi;
        if (more) tile_ids_issue(G, R, j0 + 16, lane, false, nxt);
        const bool valid = cur.valid;
        const int oz = opaque_zero();
        const float* W1l = W1s + oz;
        f32x4 y0[4], gin[8];
        recompute_g_from(shifted(Ws, oz), W0s + oz, cur, lane, y0, gin);
        const BSplit gs = split_operand(gin);
        if (more) tile_m0_issue(m0, lane, nxt);
#pragma unroll
        for (int ot0 = 0; ot0 < 8; ot0 += 2) {
            f32x4 y2[2];
            layer1_group<2, PLAIN>(W1l, gs, ot0, y2, lane);
            if (valid) {
#pragma unroll
                for (int u = 0; u < 2; ++u) { s1[ot0 + u] += y2[u]; s2[ot0 + u] += y2[u] * y2[u]; }
            }
        }
        cur = nxt;
    }
    static_assert(128 * kW1Ld >= kVfeWaves * 16 * 132, "the W1 buffer must hold the flush scratch");
    flush_channel_sums<8>(s1, s2, sums1, 128, red, W1s, lane, wave);
}

// sweep 2 of layer 1: h1 = ReLU(BN(y1)), voxel_feats = segmented max (zero-filled by the caller)
template <bool PLAIN>
__global__ __launch_bounds__(kVfeBlk) void vfe_layer1_kernel(VfeGeo G, VfeW W, const float* __restrict__ m0,
                                                             float* __restrict__ vf, unsigned char* __restrict__ ties,
                                                             BnFoldDev F, const double* __restrict__ sums1) {
    __shared__ float W0s[64 * 16];
    __shared__ __attribute__((aligned(16))) float W1s[128 * kW1Ld];
    __shared__ __attribute__((aligned(16))) float tiles[kVfeWaves][16 * kTileLd];
    __shared__ __attribute__((aligned(16))) float bn1f_s[2][128];
    VFE_T_ENTRY();
    stage_w0(W.w0, W0s);
    stage_w1(W.w1, W1s, false, !PLAIN);
    VFE_STAGE_BN0_FWD(W, Ws)
    if (F.on) {
        if (threadIdx.x < 128) bn_fold_channel(F, 128, threadIdx.x, sums1[threadIdx.x], sums1[128 + threadIdx.x], bn1f_s[0], bn1f_s[1]);
    } else {
        stage_vec(W.scale1, bn1f_s[0], 128);
        stage_vec(W.shift1, bn1f_s[1], 128);
    }
    Ws.scale1 = bn1f_s[0];
    Ws.shift1 = bn1f_s[1];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    VFE_T_BEGIN();
    const WaveRange R = wave_range(G, blockIdx.x * kVfeWaves + wave);
    float* tile = tiles[wave];
    SegCarry<2> carry;
    carry.init(G, R, -INFINITY);
    TileIn cur, nxt;
    tile_ids_issue(G, R, R.j_lo, lane, true, cur);
    tile_m0_issue(m0, lane, cur);
    for (int j0 = R.j_lo; j0 < R.j_hi; j0 += 16) {
        const bool more = j0 + 16 < R.j_hi;
        if (more) tile_ids_issue(G, R, j0 + 16, lane, true, nxt);
        const bool valid = cur.valid;
        const int pid = cur.pid, pid_next = cur.pid_next;
        const int oz = opaque_zero();
        const VfeW Wl = shifted(Ws, oz);
        f32x4 y0[4], gin[8], y1[8], h1[8];
        VFE_T(0);
        recompute_g_from(Wl, W0s + oz, cur, lane, y0, gin);
        VFE_T(1);
        layer1_linear<PLAIN>(W1s + oz, split_operand(gin), y1, lane);
        if (more) tile_m0_issue(m0, lane, nxt);
        VFE_T(2);
        bn_affine<8>(y1, Wl.scale1, Wl.shift1, h1, lane);             // (the clamp follows the max: seg_scan TRACK)
        tile_store<8, kTileLd>(tile, h1, lane);
        wave_sync();
        VFE_T(3);
        seg_scan<2, true, kTileLd, true>(tile, pid, pid_next, valid, 128, vf, carry, lane, ties);
        wave_sync();
        VFE_T(4);
        cur = nxt;
    }
    seg_resolve(carry, vf, 128, ties, lane);
    VFE_T_END();
}

// ------------------------------------------------------------------------------------------------ backward
// one output tile of the layer-1 backward inputs: dh[t][c] = dvf[pid][c] where h1[t][c] == vf[pid][c] > 0
// (max-pool + ReLU routing on the recomputed, bit-identical forward value), yhat = (y1 - mean) * invstd
// (m, d: the pillar's rows of vf / dvf for this channel tile, gathered by routed_rows at the TOP of the tile's iteration:
// inside the channel-tile loop each pair of gathers was waited for where it was issued, 4 x ~3.5 k cycles per tile --
// 58 % of the layer-1 backward sweep, tools/vfe_time.py)
__device__ __forceinline__ void routed_tile(const f32x4 y, const Bn1& bn, const float4 m, const float4 d, bool valid, int ot,
                                            int lane, f32x4* dh, f32x4* yhat) {
    const int g = lane >> 4, c0 = 16 * ot + 4 * g;
    const float4 s = *reinterpret_cast<const float4*>(bn.scale + c0);
    const float4 b = *reinterpret_cast<const float4*>(bn.shift + c0);
    const float4 mu = *reinterpret_cast<const float4*>(bn.mean + c0);
    const float4 is = *reinterpret_cast<const float4*>(bn.invstd + c0);
    const float sc[4] = {s.x, s.y, s.z, s.w}, sh[4] = {b.x, b.y, b.z, b.w}, mn[4] = {mu.x, mu.y, mu.z, mu.w},
                iv[4] = {is.x, is.y, is.z, is.w}, mx[4] = {m.x, m.y, m.z, m.w}, dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float h = fmaxf(y[r] * sc[r] + sh[r], 0.f);          // exactly bn_relu's expression
        (*dh)[r] = (valid && h > 0.f && h == mx[r]) ? dd[r] : 0.f;
        (*yhat)[r] = (y[r] - mn[r]) * iv[r];
    }
}

template <int NT>
__device__ __forceinline__ void routed_rows(const float* __restrict__ vf, const float* __restrict__ dvf, int pid, bool valid,
                                            int ot0, int lane, float4 (&m)[NT], float4 (&d)[NT]) {
    const int g = lane >> 4;
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        m[u] = make_float4(0, 0, 0, 0);
        d[u] = m[u];
        if (valid) {
            m[u] = *reinterpret_cast<const float4*>(vf + (int64_t)pid * 128 + 16 * (ot0 + u) + 4 * g);
            d[u] = *reinterpret_cast<const float4*>(dvf + (int64_t)pid * 128 + 16 * (ot0 + u) + 4 * g);
        }
    }
}

// The same two sums from the pillar rows alone, for the pillars with exactly one arg-max point per channel (ties[p] == 0,
// GeomaeVfeArgs.pillar_ties): the routed gradient of channel c over the pillar's points is d_vf[p][c] at that one point if
// vf[p][c] > 0 and nothing otherwise, and the point's yhat is ((vf - shift) / scale - mean) * invstd.  [V,128] x 2 rows
// instead of a sweep of layer 0 + the 128 x 128 GEMM over all points; the sweep below handles the flagged pillars' points
// (two distinct points DO produce the same fp32 maximum now and then: ~1 in 10^6 (pillar, channel) pairs on LiDAR frames).
__global__ __launch_bounds__(1024) void vfe_bwd_stats1_pillars_kernel(const float* __restrict__ vf, const float* __restrict__ dvf,
                                                                       int num_pillars, Bn1 bn,
                                                                       const unsigned char* __restrict__ ties,
                                                                       double* __restrict__ bsums1) {
    // 32 pillar lanes x 32 channel quads per workgroup (16 bytes per load, four pillars' loads in flight per thread), FEW
    // workgroups: every workgroup ends in 256 fp64 atomics on the same 256 words, and same-address atomics serialise
    __shared__ float part[32][2][128];
    const int q = threadIdx.x & 31, pl = threadIdx.x >> 5, c = 4 * q;
    const float4 sc = *reinterpret_cast<const float4*>(bn.scale + c), sh = *reinterpret_cast<const float4*>(bn.shift + c);
    const float4 mu = *reinterpret_cast<const float4*>(bn.mean + c), is = *reinterpret_cast<const float4*>(bn.invstd + c);
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const int stride = gridDim.x * 32;
    for (int p0 = blockIdx.x * 32 + pl; p0 < num_pillars; p0 += 4 * stride) {
        float4 h[4], d[4];
        bool use[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = p0 + k * stride;
            const bool in = p < num_pillars;
            h[k] = in ? *reinterpret_cast<const float4*>(vf + (int64_t)p * 128 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            d[k] = in ? *reinterpret_cast<const float4*>(dvf + (int64_t)p * 128 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            use[k] = in && ties[p] == 0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float hv[4] = {h[k].x, h[k].y, h[k].z, h[k].w}, dv[4] = {d[k].x, d[k].y, d[k].z, d[k].w};
            const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
            const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, isv[4] = {is.x, is.y, is.z, is.w};
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (hv[r] > 0.f && use[k]) {
                    s1[r] += dv[r];
                    // gamma == 0 (scale == 0): every point of the channel has the same y, yhat cannot be recovered from the
                    // pooled value (0 / 0): such a channel contributes yhat = 0 instead of a NaN that would poison the step
                    const float yh = fabsf(scv[r]) > 1e-30f ? ((hv[r] - shv[r]) / scv[r] - muv[r]) * isv[r] : 0.f;
                    s2[r] += dv[r] * yh;
                }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { part[pl][0][c + r] = s1[r]; part[pl][1][c + r] = s2[r]; }
    __syncthreads();
    if (threadIdx.x < 256) {
        const int k = threadIdx.x >> 7, ch = threadIdx.x & 127;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 32; ++w) sum += part[w][k][ch];
        atomicAdd(bsums1 + 128 * k + ch, (double)sum);
    }
}

template <bool PLAIN>
__global__ __launch_bounds__(kVfeBlk) void vfe_bwd_stats1_kernel(VfeGeo G, VfeW W, const float* __restrict__ m0,
                                                                 const float* __restrict__ vf, const float* __restrict__ dvf,
                                                                 Bn1 bn, double* __restrict__ bsums1,
                                                                 const unsigned char* __restrict__ ties) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const WaveRange R = wave_range(G, blockIdx.x * kVfeWaves + wave);
    // ties: only the points of flagged pillars count (vfe_bwd_stats1_pillars_kernel has done the rest); a wave sweeps only
    // the tiles that hold such a point, a workgroup without one leaves before it stages anything
    unsigned int tile_mask = 0xfu;
    if (ties) {
        tile_mask = 0u;
#pragma unroll
        for (int k = 0; k < kVfePts / 16; ++k) {
            const int j = R.j_lo + 16 * k + (lane & 15);
            const bool f = j < R.j_hi && ties[pillar_of(G, j)] != 0;
            if (__any(f)) tile_mask |= 1u << k;
        }
        if (!__syncthreads_or(tile_mask != 0u)) return;
    }
    __shared__ float W0s[64 * 16];
    VFE_T_ENTRY();
    __shared__ __attribute__((aligned(16))) float W1s[128 * kW1Ld];
    __shared__ float red[kVfeWaves * 2 * 128];
    stage_w0(W.w0, W0s);
    stage_w1(W.w1, W1s, false, !PLAIN);
    VFE_STAGE_BN0_FWD(W, Ws)
    VFE_STAGE_BN1(bn, bns)
    __syncthreads();
    // two passes over the points, 4 output tiles each: 16 instead of 32 accumulator vectors live across the
    // loop (the single-pass version spilled); layer 0 is recomputed twice, layer 1's MFMA count is unchanged
    VFE_T_BEGIN();
    f32x4 s1[8], s2[8];
#pragma unroll
    for (int ot = 0; ot < 8; ++ot) { s1[ot] = f32x4{0, 0, 0, 0}; s2[ot] = f32x4{0, 0, 0, 0}; }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        for (int j0 = R.j_lo; j0 < R.j_hi; j0 += 16) {
            if (!((tile_mask >> ((j0 - R.j_lo) >> 4)) & 1u)) continue;
            const int j = j0 + (lane & 15);
            const int pid = j < R.j_hi ? pillar_of(G, j) : 0;
            const bool valid = j < R.j_hi && (!ties || ties[pid] != 0);
            const int oz = opaque_zero();
            const Bn1 bnl = shifted(bns, oz);
            VFE_T(0);
            float4 mrow[4], drow[4];
            routed_rows<4>(vf, dvf, pid, valid, 4 * half, lane, mrow, drow);
            f32x4 y0[4], gin[8];
            recompute_g(G, shifted(Ws, oz), W0s + oz, m0, j, pid, valid, lane, y0, gin);
            VFE_T(1);
            const BSplit gs = split_operand(gin);
#pragma unroll
            for (int q = 0; q < 4; q += 2) {
                const int ot0 = 4 * half + q;
                f32x4 y4[2];
                layer1_group<2, PLAIN>(W1s + oz, gs, ot0, y4, lane);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    f32x4 dh, yh;
                    routed_tile(y4[u], bnl, mrow[q + u], drow[q + u], valid, ot0 + u, lane, &dh, &yh);
                    s1[ot0 + u] += dh;
                    s2[ot0 + u] += dh * yh;
                }
            }
            VFE_T(2);
        }
    }
    VFE_T(0);
    flush_channel_sums<8>(s1, s2, bsums1, 128, red, W1s, lane, wave);
    VFE_T(3);
    VFE_T_END_AT(16);
}

// layer-1 backward sweep:
//   dy1 = invstd1 * (dyh - S1/n - yhat * S2/n)  -> bf16 copy + g bf16 copy (operands of dW1 = dy1^T g)
//   dg = dy1 W1 ; dh0_direct = dg[:, :64] (stored fp32) ; dm0 = segmented sum of dg[:, 64:] (zero-filled by
//   the caller; pillars that straddle waves are combined with float atomics)
template <bool PLAIN>
__global__ __launch_bounds__(kVfeBlk) void vfe_bwd_layer1_kernel(
    VfeGeo G, VfeW W, const float* __restrict__ m0, const float* __restrict__ vf, const float* __restrict__ dvf, Bn1 bn,
    const double* __restrict__ bsums1, float n_eff, bf16_t* __restrict__ dy1_b, bf16_t* __restrict__ g_b,
    float* __restrict__ dy1_f, float* __restrict__ dh0, float* __restrict__ dm0, float* __restrict__ d_beta1,
    float* __restrict__ d_gamma1) {
    __shared__ float W0s[64 * 16];
    VFE_T_ENTRY();
    __shared__ __attribute__((aligned(16))) float W1s[128 * kW1Ld];      // W1, then W1^T (dg = dy1 W1)
    __shared__ __attribute__((aligned(16))) float tiles[kVfeWaves][16 * kTile0Ld];
    // dy1 = scale * (dh - S1/n - yhat * S2/n) with yhat = (y - mean) * invstd, folded per channel into
    // dy1 = scale * dh + A + y * B:  B = -scale * (S2/n) * invstd,  A = -scale * (S1/n) - mean * B
    __shared__ __attribute__((aligned(16))) float bn1s[2][128];            // A, B
    stage_w0(W.w0, W0s);
    VFE_STAGE_BN0_FWD(W, Ws)
    VFE_STAGE_BN1(bn, bns)
    for (int c = threadIdx.x; c < 128; c += kVfeBlk) {
        const float t1 = (float)(bsums1[c] / (double)n_eff), t2 = (float)(bsums1[128 + c] / (double)n_eff);
        const float bq = -bn.scale[c] * t2 * bn.invstd[c];
        bn1s[1][c] = bq;
        bn1s[0][c] = -bn.scale[c] * t1 - bn.mean[c] * bq;
        if (d_beta1 && blockIdx.x == 0) {          // single process: the sums ARE d beta / d gamma (one writer: no atomics)
            d_beta1[c] += (float)bsums1[c];
            d_gamma1[c] += (float)bsums1[128 + c];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const WaveRange R = wave_range(G, blockIdx.x * kVfeWaves + wave);
    float* tile = tiles[wave];
    // Both W1 (recompute y1) and W1^T (dg = dy1 W1) are needed; the second product reads its operand from the SAME split
    // copy with transposing LDS reads (layer1_group_t), so dy1 stays in registers between the two: one sweep.  (Until
    // round 2 the sweep was split: A1 computed dy1 with W1 and parked it in HBM as fp32, the workgroup re-staged W1^T,
    // A2 re-read dy1 -- 108 MB of the kernel's 234 MB.)
    stage_w1(W.w1, W1s, false, !PLAIN);
    __syncthreads();
    SegCarry<1> carry;
    carry.init(G, R);
    VFE_T_BEGIN();
    // ids / features / layer-0 maxima a tile ahead (tile_ids_issue); the 16 row gathers of the routing are issued BEHIND them:
    // the memory counter retires in order, and what the tile needs first must not queue behind what it needs last
    TileIn cur, nxt;
    tile_ids_issue(G, R, R.j_lo, lane, true, cur);
    tile_m0_issue(m0, lane, cur);
    for (int j0 = R.j_lo; j0 < R.j_hi; j0 += 16) {
        const bool more = j0 + 16 < R.j_hi;
        if (more) tile_ids_issue(G, R, j0 + 16, lane, true, nxt);
        const int j = j0 + (lane & 15);
        const bool valid = cur.valid;
        const int pid = cur.pid, pid_next = cur.pid_next;
        const int oz = opaque_zero();
        const Bn1 bnl = shifted(bns, oz);
        const float* cA = bn1s[0] + oz;
        const float* cB = bn1s[1] + oz;
        f32x4 dy1[8];
        VFE_T(0);
        float4 mrow[8], drow[8];
        routed_rows<8>(vf, dvf, pid, valid, 0, lane, mrow, drow);
        {
            f32x4 y0[4], gin[8];
            recompute_g_from(shifted(Ws, oz), W0s + oz, cur, lane, y0, gin);
            store_rows_bf16<128>(g_b, G.n_points, j, 128, 0, gin, lane, true);   // tile-blocked (pad rows of the last tile: unread)
            VFE_T(1);
            const BSplit gs = split_operand(gin);
            // the MFMAs of channel-tile pair k + 1 are issued before the routing arithmetic of pair k
            f32x4 ya[2], yb[2];
            layer1_group<2, PLAIN>(W1s + oz, gs, 0, ya, lane);
            if (more) tile_m0_issue(m0, lane, nxt);
#pragma unroll
            for (int ot0 = 0; ot0 < 8; ot0 += 2) {
                if (ot0 + 2 < 8) layer1_group<2, PLAIN>(W1s + oz, gs, ot0 + 2, yb, lane);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int ot = ot0 + u;
                    const int c0 = 16 * ot + 4 * g;
                    const float4 sc = *reinterpret_cast<const float4*>(bnl.scale + c0);    // gamma * invstd
                    const float4 sh = *reinterpret_cast<const float4*>(bnl.shift + c0);
                    const float4 a4 = *reinterpret_cast<const float4*>(cA + c0);
                    const float4 b4 = *reinterpret_cast<const float4*>(cB + c0);
                    const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
                    const float av[4] = {a4.x, a4.y, a4.z, a4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
                    const float mx[4] = {mrow[ot].x, mrow[ot].y, mrow[ot].z, mrow[ot].w};
                    const float dd[4] = {drow[ot].x, drow[ot].y, drow[ot].z, drow[ot].w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float y = ya[u][r];
                        const float h = fmaxf(y * scv[r] + shv[r], 0.f);          // exactly bn_relu's expression
                        const float dh = (h > 0.f && h == mx[r]) ? dd[r] : 0.f;   // max-pool + ReLU routing
                        dy1[ot][r] = valid ? scv[r] * dh + (av[r] + y * bv[r]) : 0.f;
                    }
                    if (valid) *reinterpret_cast<uint2*>(dy1_b + ((int64_t)(j >> 4) * 8 + ot) * 256 + (j & 15) * 16 + 4 * g) = pack4(dy1[ot]);   // operand of dW1, tile-blocked
                }
                ya[0] = yb[0];
                ya[1] = yb[1];
            }
        }
        // dg[t][k] = sum_o W1[o][k] dy1[t][o]; two output tiles at a time
        const float* W1l = W1s + opaque_zero();
        VFE_T(2);
        const BSplit ds = split_operand(dy1);
#pragma unroll
        for (int ot0 = 0; ot0 < 8; ot0 += 2) {
            f32x4 d2[2];
            layer1_group_t<2, PLAIN>(W1l, ds, ot0, d2, lane);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ct = ot0 + u;
                if (ct < 4) {
                    if (valid) *reinterpret_cast<float4*>(dh0 + (int64_t)j * 64 + 16 * ct + 4 * g) =
                                   make_float4(d2[u][0], d2[u][1], d2[u][2], d2[u][3]);
                } else {
                    *reinterpret_cast<float4*>(tile + (lane & 15) * kTile0Ld + 16 * (ct - 4) + 4 * g) =
                        make_float4(d2[u][0], d2[u][1], d2[u][2], d2[u][3]);
                }
            }
        }
        wave_sync();
        VFE_T(3);
        seg_scan<1, false, kTile0Ld>(tile, pid, pid_next, valid, 64, dm0, carry, lane);
        wave_sync();
        VFE_T(4);
        cur = nxt;
    }
    VFE_T_END();
    (void)dy1_f;                                   // (kept in the C ABI; no longer written)
}

// layer-0 routing sweep (needs the complete dm0): dh0 = dh0_direct + dm0[pid] where h0 == m0[pid] > 0 ;
// sums of dh0 and dh0 * yhat0 for the BN0 backward
// ACC: also contract A[o][k] = sum_t dh0[t][o] * ft[t][k] over the points, ft = [f - S1/N (11 centred features) | 0 0 0 0 | 1]
// (column 15 = the plain sum of dh0), into dw0_acc [64][16] (zero-filled by the caller) -- with the moments of the features
// that is all the layer-0 weight gradient needs (vfe_dw0_finalize_kernel), so dh0 is not written back and no further
// sweep reads it.  Centred features keep the accumulated magnitudes near those of the gradient itself.
template <bool ACC>
__global__ __launch_bounds__(kVfeBlk) void vfe_bwd_route0_kernel(VfeGeo G, VfeW W, const float* __restrict__ m0,
                                                                 Bn0 bn0, const float* __restrict__ dm0,
                                                                 float* __restrict__ dh0, double* __restrict__ bsums0,
                                                                 const double* __restrict__ moments, float inv_n,
                                                                 float* __restrict__ dw0_acc) {
    __shared__ float W0s[64 * 16];
    VFE_T_ENTRY();
    __shared__ float red[kVfeWaves * 2 * 64];
    __shared__ __attribute__((aligned(16))) float tiles[kVfeWaves][16 * kTile0Ld];   // [t][0..63] dh0, [t][64..79] ft; then the flush's scratch
    __shared__ float mu_s[16];
    stage_w0(W.w0, W0s);
    VFE_STAGE_BN0(bn0, bn0l)
    if (ACC) {
        if (threadIdx.x < 16) mu_s[threadIdx.x] = threadIdx.x < 11 ? (float)(moments[threadIdx.x] * (double)inv_n) : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const WaveRange R = wave_range(G, blockIdx.x * kVfeWaves + wave);
    float* tile = tiles[wave];
    f32x4 s1[4], s2[4], dw[4];
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) { s1[ot] = f32x4{0, 0, 0, 0}; s2[ot] = f32x4{0, 0, 0, 0}; dw[ot] = f32x4{0, 0, 0, 0}; }
    VFE_T_BEGIN();
    // ids / features / layer-0 maxima a tile ahead (TileIn); the tile's own gathers (dm0 rows, dh0) at the top of its iteration
    TileIn cur, nxt;
    tile_ids_issue(G, R, R.j_lo, lane, false, cur);
    tile_m0_issue(m0, lane, cur);
    for (int j0 = R.j_lo; j0 < R.j_hi; j0 += 16) {
        VFE_T(0);
        const bool more = j0 + 16 < R.j_hi;
        if (more) tile_ids_issue(G, R, j0 + 16, lane, false, nxt);
        const int j = j0 + (lane & 15);
        const bool valid = cur.valid;
        const int pid = cur.pid;
        const int oz = opaque_zero();
        const Bn0 b0 = shifted(bn0l, oz);
        float4 dmr[4], ddr[4];
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) {
            dmr[ot] = make_float4(0, 0, 0, 0);
            ddr[ot] = dmr[ot];
            if (valid) {
                dmr[ot] = *reinterpret_cast<const float4*>(dm0 + (int64_t)pid * 64 + 16 * ot + 4 * g);
                ddr[ot] = *reinterpret_cast<const float4*>(dh0 + (int64_t)j * 64 + 16 * ot + 4 * g);
            }
        }
        float f[4] = {cur.f[0], cur.f[1], cur.f[2], cur.f[3]};
        f32x4 y0[4];
        layer0_linear(W0s + oz, f, y0, lane);
        if (more) tile_m0_issue(m0, lane, nxt);
        VFE_T(1);
        f32x4 dhv[4];
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) {
            const int c0 = 16 * ot + 4 * g;
            const float4 s = *reinterpret_cast<const float4*>(b0.scale + c0);
            const float4 b = *reinterpret_cast<const float4*>(b0.shift + c0);
            const float4 mu = *reinterpret_cast<const float4*>(b0.mean + c0);
            const float4 is = *reinterpret_cast<const float4*>(b0.invstd + c0);
            const float4 m = cur.m[ot], dm = dmr[ot], dd = ddr[ot];
            const float sc[4] = {s.x, s.y, s.z, s.w}, sh[4] = {b.x, b.y, b.z, b.w}, mn[4] = {mu.x, mu.y, mu.z, mu.w},
                        iv[4] = {is.x, is.y, is.z, is.w}, mx[4] = {m.x, m.y, m.z, m.w}, dmv[4] = {dm.x, dm.y, dm.z, dm.w},
                        ddv[4] = {dd.x, dd.y, dd.z, dd.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float h = fmaxf(y0[ot][r] * sc[r] + sh[r], 0.f);
                float dh = 0.f;
                if (valid && h > 0.f) dh = ddv[r] + (h == mx[r] ? dmv[r] : 0.f);
                const float yh = (y0[ot][r] - mn[r]) * iv[r];
                dhv[ot][r] = dh;
                s1[ot][r] += dh;
                s2[ot][r] += dh * yh;
            }
            if (!ACC && valid) *reinterpret_cast<float4*>(dh0 + (int64_t)j * 64 + c0) = make_float4(dhv[ot][0], dhv[ot][1], dhv[ot][2], dhv[ot][3]);
        }
        VFE_T(2);
        if (ACC) {
            // A[o][k] += sum_t dh0[t][o] ft[t][k]: token contraction -> tile through LDS, MFMA with k = t
            tile_store<4, kTile0Ld>(tile, dhv, lane);
            float4 ft = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid) {
                const float4 mu4 = *reinterpret_cast<const float4*>(mu_s + 4 * g);
                ft = make_float4(f[0] - mu4.x, f[1] - mu4.y, f[2] - mu4.z, f[3] - mu4.w);
                if (g == 2) ft.w = 0.f;                       // feature 11 is padding (mu 0, f 0): keep it exactly 0
                if (g == 3) ft = make_float4(0.f, 0.f, 0.f, 1.f);     // columns 12-14 unused, column 15 = 1
            }
            *reinterpret_cast<float4*>(tile + (lane & 15) * kTile0Ld + 64 + 4 * g) = ft;
            wave_sync();
            VFE_T(3);
            const int o = lane & 15;
#pragma unroll
            for (int ot = 0; ot < 4; ++ot)
#pragma unroll
                for (int sgrp = 0; sgrp < 4; ++sgrp) {
                    const float a = tile[(4 * sgrp + g) * kTile0Ld + 16 * ot + o];      // A[row = o][k = t = 4s + g]
                    const float b = tile[(4 * sgrp + g) * kTile0Ld + 64 + o];           // B[k = t][col = feature o]
                    dw[ot] = mfma_f32(a, b, dw[ot]);
                }
            wave_sync();
            VFE_T(4);
        }
        cur = nxt;
    }
    VFE_T(5);
    // the waves' partial contractions meet in LDS as plain stores into per-wave slices of the (now free) tile buffer and are
    // summed by the flush below.  (LDS float atomics into one [64][16] array serialise: 31 k of the kernel's 60 k cycles.)
    float* part = &tiles[0][0];                                      // [kVfeWaves][64 * 16]
    static_assert(kVfeWaves * 16 * kTile0Ld >= kVfeWaves * 64 * 16, "the tile buffer must hold the partials");
    flush_channel_sums<4>(s1, s2, bsums0, 64, red, part, lane, wave);    // (its barriers: every wave done with its tile / its scratch)
    if (ACC) {
        // dw[ot][r] = A[16*ot + 4g + r][feature = lane & 15]
#pragma unroll
        for (int ot = 0; ot < 4; ++ot)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[wave * 1024 + (16 * ot + 4 * g + r) * 16 + (lane & 15)] = dw[ot][r];
        __syncthreads();
    }
    if (ACC)
        for (int e = threadIdx.x; e < 64 * 16; e += kVfeBlk) {
            float sum = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < kVfeWaves; ++w8) sum += part[w8 * 1024 + e];
            atomicAdd(dw0_acc + e, sum);
        }
    if (ACC) { VFE_T_END_AT(8); }
}

// dW0 from the accumulated contraction and the feature moments (one workgroup; fp64 arithmetic):
//   dW0[o][k] += sc_o * ( A'[o][k] - (T2_o / n) * invstd_o * (W0[o] Cov)[k] + sdy_o * mu[k] )
//   Cov = S2 - S1 S1^T / N,  mu = S1 / N,  A' = sum dh0 (f - mu)^T,
//   sdy_o = sum_t dy0[t][o] / sc_o = A[o][15] - N T1_o / n - (T2_o / n) invstd_o (W0[o] . S1 - N mean_o)
// (N: local points, n: points of all ranks; T1, T2: the -- at world > 1 all-reduced -- sums of dh0 and dh0 * yhat0.  With
// one process sdy vanishes up to rounding.)  Derivation: dy0 = sc (dh0 - T1/n - yhat0 T2/n), yhat0 = invstd (W0 f - mean).
__global__ __launch_bounds__(1024) void vfe_dw0_finalize_kernel(const float* __restrict__ acc, const double* __restrict__ moments,
                                                                const double* __restrict__ bsums0, double n_local, double n_eff,
                                                                const float* __restrict__ w0, Bn0 bn0, float* __restrict__ dw0,
                                                                float* __restrict__ d_beta0, float* __restrict__ d_gamma0) {
    __shared__ double cov[11][11], mu[11];
    const int o = threadIdx.x >> 4, k = threadIdx.x & 15;
    if (threadIdx.x < 11) mu[threadIdx.x] = moments[threadIdx.x] / n_local;
    if (threadIdx.x < 121) {
        const int a = threadIdx.x / 11, b = threadIdx.x % 11;
        cov[a][b] = moments[kMomS2 + threadIdx.x] - moments[a] * moments[b] / n_local;
    }
    __syncthreads();
    const double t1 = bsums0[o], t2 = bsums0[64 + o];
    const double invstd = (double)bn0.invstd[o], mean = (double)bn0.mean[o], sc = (double)bn0.scale[o];
    if (k < 11) {
        double wc = 0.0, ws1 = 0.0;
#pragma unroll
        for (int j = 0; j < 11; ++j) {
            const double w = (double)w0[o * 11 + j];
            wc += w * cov[j][k];
            ws1 += w * moments[j];
        }
        const double sdy = (double)acc[o * 16 + 15] - n_local * t1 / n_eff - (t2 / n_eff) * invstd * (ws1 - n_local * mean);
        const double v = (double)acc[o * 16 + k] - (t2 / n_eff) * invstd * wc + sdy * mu[k];
        dw0[o * 11 + k] += (float)(sc * v);
    }
    if (k == 15 && d_beta0) {                 // single process: the sums ARE d beta / d gamma
        d_beta0[o] += (float)t1;
        d_gamma0[o] += (float)t2;
    }
}

// layer-0 backward: dy0 = invstd0 * (dyh0 - T1/n - yhat0 * T2/n) ; dW0 += dy0^T f   (64 x 11)
__global__ __launch_bounds__(kVfeBlk) void vfe_bwd_layer0_kernel(VfeGeo G, VfeW W, const float* __restrict__ dh0, Bn0 bn0,
                                                                 const double* __restrict__ bsums0, float n_eff,
                                                                 float* __restrict__ dw0, float* __restrict__ d_beta0,
                                                                 float* __restrict__ d_gamma0) {
    __shared__ float W0s[64 * 16];
    __shared__ __attribute__((aligned(16))) float tiles[kVfeWaves][16 * kTile0Ld];     // [t][0..63] dy0, [t][64..79] f
    __shared__ float bn0s[2][64];
    stage_w0(W.w0, W0s);
    VFE_STAGE_BN0(bn0, bn0l)
    for (int c = threadIdx.x; c < 64; c += kVfeBlk) {
        bn0s[0][c] = (float)(bsums0[c] / (double)n_eff);
        bn0s[1][c] = (float)(bsums0[64 + c] / (double)n_eff);
        if (d_beta0 && blockIdx.x == 0) {
            d_beta0[c] += (float)bsums0[c];
            d_gamma0[c] += (float)bsums0[64 + c];
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const WaveRange R = wave_range(G, blockIdx.x * kVfeWaves + wave);
    float* tile = tiles[wave];
    f32x4 dw[4];                                   // C layout: row = out channel 16*ot' ... see below
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) dw[ot] = f32x4{0, 0, 0, 0};
    for (int j0 = R.j_lo; j0 < R.j_hi; j0 += 16) {
        const int j = j0 + (lane & 15);
        const bool valid = j < R.j_hi;
        float f[4];
        build_features(G, j, valid, g, f);
        f32x4 y0[4], dy0[4];
        layer0_linear(W0s, f, y0, lane);
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) {
            const int c0 = 16 * ot + 4 * g;
            const float4 s = *reinterpret_cast<const float4*>(bn0l.scale + c0);
            const float4 mu = *reinterpret_cast<const float4*>(bn0l.mean + c0);
            const float4 is = *reinterpret_cast<const float4*>(bn0l.invstd + c0);
            float4 dd = make_float4(0, 0, 0, 0);
            if (valid) dd = *reinterpret_cast<const float4*>(dh0 + (int64_t)j * 64 + c0);
            const float sc[4] = {s.x, s.y, s.z, s.w}, mn[4] = {mu.x, mu.y, mu.z, mu.w}, iv[4] = {is.x, is.y, is.z, is.w},
                        ddv[4] = {dd.x, dd.y, dd.z, dd.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float yh = (y0[ot][r] - mn[r]) * iv[r];
                dy0[ot][r] = valid ? sc[r] * (ddv[r] - bn0s[0][c0 + r] - yh * bn0s[1][c0 + r]) : 0.f;
            }
        }
        // dW0[o][k] += sum_t dy0[t][o] f[t][k]: token contraction -> tile through LDS, MFMA with k = t
        tile_store<4, kTile0Ld>(tile, dy0, lane);
        *reinterpret_cast<float4*>(tile + (lane & 15) * kTile0Ld + 64 + 4 * g) = make_float4(f[0], f[1], f[2], f[3]);
        wave_sync();
        const int o = lane & 15;
#pragma unroll
        for (int ot = 0; ot < 4; ++ot)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float a = tile[(4 * s + g) * kTile0Ld + 16 * ot + o];      // A[row = o][k = t = 4s + g]
                const float b = tile[(4 * s + g) * kTile0Ld + 64 + o];           // B[k = t][col = feature o]
                dw[ot] = mfma_f32(a, b, dw[ot]);
            }
        wave_sync();
    }
    // dw[ot][r] = dW0[16*ot + 4g + r][feature = lane & 15]: per-wave slices of the tile buffer, summed below (no LDS atomics,
    // as in vfe_bwd_route0_kernel)
    __syncthreads();                                                 // every wave is done with its tile
    float* part = &tiles[0][0];                                      // [kVfeWaves][64 * 16]
#pragma unroll
    for (int ot = 0; ot < 4; ++ot)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave * 1024 + (16 * ot + 4 * g + r) * 16 + (lane & 15)] = dw[ot][r];
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 16; e += kVfeBlk) {
        const int r = e >> 4, c = e & 15;
        float sum = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < kVfeWaves; ++w8) sum += part[w8 * 1024 + e];
        if (c < 11) atomicAdd(dw0 + r * 11 + c, sum);
    }
}

}  // namespace geomae

using namespace geomae;

static int vfe_common(const GeomaeVfeArgs* a, VfeGeo* G, VfeW* W, const char* who) {
    GEOMAE_REQUIRE(a, "%s: null args", who);
    GEOMAE_REQUIRE(a->feat_sorted && a->pid_sorted && a->seg_start && a->w0 && a->w1, "%s: null pointer in args", who);
    GEOMAE_REQUIRE(a->num_points >= 1 && a->num_points < (1ll << 31) - kVfePts && a->max_pillars >= 1, "%s: bad sizes", who);
    G->feat = a->feat_sorted; G->pid = a->pid_sorted; G->seg_start = a->seg_start; G->n_points = (int)a->num_points;
    W->w0 = a->w0; W->w1 = a->w1; W->scale0 = a->scale0; W->shift0 = a->shift0; W->scale1 = a->scale1; W->shift1 = a->shift1;
    return GEOMAE_OK;
}
// the layer-1 sweeps' two forms: bf16 x 3 split products (fp32 grade) | plain bf16 products (GeomaeVfeArgs.layer1_bf16)
#define VFE_L1_FORM(a, kernel) ((a)->layer1_bf16 ? kernel<true> : kernel<false>)
static dim3 vfe_grid(const GeomaeVfeArgs* a) { return dim3(cdiv(cdiv(a->num_points, kVfePts), kVfeWaves)); }

#ifdef GEOMAE_PHASE_TIMING
extern "C" void geomae_debug_read_vfe_stamps(unsigned long long* out, int clear) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(geomae::geomae_stamps), sizeof(unsigned long long) * GEOMAE_STAMP_BLOCKS * GEOMAE_STAMP_SLOTS);
    if (clear) {
        static unsigned long long zeros[GEOMAE_STAMP_BLOCKS * GEOMAE_STAMP_SLOTS];
        (void)hipMemcpyToSymbol(HIP_SYMBOL(geomae::geomae_stamps), zeros, sizeof(zeros));
    }
}
#endif

extern "C" int geomae_vfe_prepare(const float* points, int32_t num_features, int64_t num_points, const int32_t* order,
                                  const int32_t* inv, const float* pillar_mean, const int32_t* voxel_coors,
                                  const float* voxel_size, const float* center_offset, float* feat_sorted,
                                  int32_t* pid_sorted, hipStream_t stream) {
    if (num_points <= 0) return GEOMAE_OK;
    GEOMAE_REQUIRE(points && order && inv && pillar_mean && voxel_coors && voxel_size && center_offset && feat_sorted &&
                   pid_sorted, "vfe_prepare: null argument");
    GEOMAE_REQUIRE(num_features >= 5, "vfe_prepare: points need 5 features (x, y, z, intensity, dt)");
    hipLaunchKernelGGL(vfe_prepare_kernel, dim3(stream_grid(num_points, 256)), dim3(256), 0, stream, points, num_features,
                       num_points, order, inv, pillar_mean, (const int4*)voxel_coors, voxel_size[0], voxel_size[1],
                       voxel_size[2], center_offset[0], center_offset[1], center_offset[2], feat_sorted, pid_sorted,
                       (double*)nullptr);
    return check_launch("vfe_prepare_kernel");
}

extern "C" int64_t geomae_vfe_moments_workspace_bytes(void) { return (int64_t)kMomBlocks * kMomPad * sizeof(double); }

extern "C" int geomae_vfe_prepare_moments(const float* points, int32_t num_features, int64_t num_points, const int32_t* order,
                                          const int32_t* inv, const float* pillar_mean, const int32_t* voxel_coors,
                                          const float* voxel_size, const float* center_offset, float* feat_sorted,
                                          int32_t* pid_sorted, void* workspace, double* moments, hipStream_t stream) {
    if (num_points <= 0) return GEOMAE_OK;
    GEOMAE_REQUIRE(points && order && inv && pillar_mean && voxel_coors && voxel_size && center_offset && feat_sorted &&
                   pid_sorted && workspace && moments, "vfe_prepare_moments: null argument");
    GEOMAE_REQUIRE(num_features >= 5, "vfe_prepare_moments: points need 5 features (x, y, z, intensity, dt)");
    int grid = stream_grid(num_points, 256);
    if (grid > kMomBlocks) grid = kMomBlocks;
    hipLaunchKernelGGL(vfe_prepare_kernel, dim3(grid), dim3(256), 0, stream, points, num_features, num_points, order, inv,
                       pillar_mean, (const int4*)voxel_coors, voxel_size[0], voxel_size[1], voxel_size[2], center_offset[0],
                       center_offset[1], center_offset[2], feat_sorted, pid_sorted, (double*)workspace);
    hipLaunchKernelGGL(vfe_moments_reduce_kernel, dim3(1), dim3(1024), 0, stream, (const double*)workspace, grid, moments);
    return check_launch("vfe_prepare_moments");
}

extern "C" int geomae_segment_mean_xyz(const float* points, int32_t num_features, int64_t num_points,
                                       const int32_t* inv, const int32_t* seg_start, const int32_t* num_pillars,
                                       int32_t max_pillars, void* sum_workspace, float* mean, hipStream_t stream) {
    if (max_pillars <= 0) return GEOMAE_OK;
    GEOMAE_REQUIRE(points && inv && seg_start && num_pillars && sum_workspace && mean, "segment_mean_xyz: null argument");
    GEOMAE_ZERO(sum_workspace, (size_t)max_pillars * 3 * sizeof(unsigned long long), stream);
    if (num_points > 0)
        hipLaunchKernelGGL(vfe_mean_accum_kernel, dim3(stream_grid(num_points, 256)), dim3(256), 0, stream, points,
                           num_features, num_points, inv, (unsigned long long*)sum_workspace);
    hipLaunchKernelGGL(vfe_mean_final_kernel, dim3(stream_grid((int64_t)max_pillars * 3, 256)), dim3(256), 0, stream,
                       (const unsigned long long*)sum_workspace, seg_start, num_pillars, mean);
    return check_launch("segment_mean_xyz");
}

extern "C" int geomae_segment_mean_xyz_sorted(const float* points, int32_t num_features, const int32_t* order,
                                              const int32_t* seg_start, const int32_t* num_pillars, int32_t max_pillars,
                                              float* mean, hipStream_t stream) {
    GEOMAE_REQUIRE(num_features >= 3 && max_pillars >= 0, "segment_mean_xyz_sorted: bad argument");
    if (max_pillars <= 0) return GEOMAE_OK;
    GEOMAE_REQUIRE(points && order && seg_start && num_pillars && mean, "segment_mean_xyz_sorted: null argument");
    hipLaunchKernelGGL(vfe_mean_sorted_kernel, dim3(stream_grid((int64_t)max_pillars * 16, 256)), dim3(256), 0, stream,
                       points, num_features, order, seg_start, num_pillars, mean);
    return check_launch("vfe_mean_sorted_kernel");
}

extern "C" int geomae_bn_finalize(const double* sums, double count, const float* moments_in, int32_t channels,
                                  const float* gamma, const float* beta, float eps, float momentum,
                                  int32_t unbiased_running_var, float* running_mean, float* running_var, float* scale,
                                  float* shift, float* invstd, float* moments_out, int64_t* num_batches_tracked,
                                  hipStream_t stream) {
    GEOMAE_REQUIRE((sums || moments_in) && channels >= 1 && channels <= 1024, "bn_finalize: bad argument");
    GEOMAE_REQUIRE(!scale || (gamma && beta && shift && invstd), "bn_finalize: scale needs gamma, beta, shift, invstd");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(256), 0, stream, sums, count, moments_in, channels, gamma, beta,
                       eps, momentum, unbiased_running_var, running_mean, running_var, scale, shift, invstd, moments_out,
                       (long long*)num_batches_tracked);
    return check_launch("bn_finalize_kernel");
}

extern "C" int geomae_vfe_stats0(const GeomaeVfeArgs* a, double* sums0, hipStream_t stream) {
    VfeGeo G; VfeW W;
    int rc = vfe_common(a, &G, &W, "vfe_stats0");
    if (rc) return rc;
    GEOMAE_REQUIRE(sums0, "vfe_stats0: null output");
    if (a->moments) {          // y0 = W0 f is linear: its batch statistics follow from the feature moments (no sweep)
        hipLaunchKernelGGL(vfe_stats0_from_moments_kernel, dim3(1), dim3(64), 0, stream, a->moments, a->w0, sums0);
        return check_launch("vfe_stats0_from_moments_kernel");
    }
    GEOMAE_ZERO(sums0, 128 * sizeof(double), stream);
    hipLaunchKernelGGL(vfe_stats0_kernel, vfe_grid(a), dim3(kVfeBlk), 0, stream, G, W, sums0);
    return check_launch("vfe_stats0_kernel");
}

extern "C" int geomae_vfe_layer0(const GeomaeVfeArgs* a, float* m0, double* sums1, hipStream_t stream) {
    VfeGeo G; VfeW W;
    int rc = vfe_common(a, &G, &W, "vfe_layer0");
    if (rc) return rc;
    GEOMAE_REQUIRE(m0 && sums1 && a->scale0 && a->shift0, "vfe_layer0: null argument");
    GEOMAE_ZERO(sums1, 256 * sizeof(double), stream);
    GEOMAE_ZERO(m0, (size_t)a->max_pillars * 64 * sizeof(float), stream);
    hipLaunchKernelGGL(vfe_layer0_kernel, vfe_grid(a), dim3(kVfeBlk), 0, stream, G, W, m0, BnFoldDev{}, (const double*)nullptr);
    if ((rc = check_launch("vfe_layer0_kernel"))) return rc;
    hipLaunchKernelGGL(VFE_L1_FORM(a, vfe_stats1_kernel), vfe_grid(a), dim3(kVfeBlk), 0, stream, G, W, (const float*)m0, sums1);
    return check_launch("vfe_stats1_kernel");
}

static int bn_fold_of(const GeomaeBnFold* f, BnFoldDev* d, const char* who) {
    GEOMAE_REQUIRE(f && f->count > 1.0 && f->gamma && f->beta && f->scale && f->shift && f->invstd && f->moments,
                   "%s: incomplete GeomaeBnFold", who);
    GEOMAE_REQUIRE((f->running_mean == nullptr) == (f->running_var == nullptr), "%s: pass both running statistics or none", who);
    d->count = f->count; d->gamma = f->gamma; d->beta = f->beta; d->eps = f->eps; d->momentum = f->momentum;
    d->running_mean = f->running_mean; d->running_var = f->running_var; d->scale = f->scale; d->shift = f->shift;
    d->invstd = f->invstd; d->moments = f->moments; d->nbt = (long long*)f->num_batches_tracked; d->on = true;
    return GEOMAE_OK;
}

extern "C" int geomae_vfe_layer0_bn(const GeomaeVfeArgs* a, const GeomaeBnFold* bn0, float* m0, double* sums1,
                                    hipStream_t stream) {
    VfeGeo G; VfeW W;
    int rc = vfe_common(a, &G, &W, "vfe_layer0_bn");
    if (rc) return rc;
    BnFoldDev F;
    if ((rc = bn_fold_of(bn0, &F, "vfe_layer0_bn"))) return rc;
    GEOMAE_REQUIRE(m0 && sums1 && a->moments && a->scale0 == bn0->scale && a->shift0 == bn0->shift,
                   "vfe_layer0_bn: needs the feature moments, and args->scale0 / shift0 = the fold's outputs");
    GEOMAE_ZERO(sums1, 256 * sizeof(double), stream);
    GEOMAE_ZERO(m0, (size_t)a->max_pillars * 64 * sizeof(float), stream);
    hipLaunchKernelGGL(vfe_layer0_kernel, vfe_grid(a), dim3(kVfeBlk), 0, stream, G, W, m0, F, a->moments);
    if ((rc = check_launch("vfe_layer0_kernel"))) return rc;
    hipLaunchKernelGGL(VFE_L1_FORM(a, vfe_stats1_kernel), vfe_grid(a), dim3(kVfeBlk), 0, stream, G, W, (const float*)m0, sums1);
    return check_launch("vfe_stats1_kernel");
}

extern "C" int geomae_vfe_layer1(const GeomaeVfeArgs* a, const float* m0, float* voxel_feats, hipStream_t stream) {
    VfeGeo G; VfeW W;
    int rc = vfe_common(a, &G, &W, "vfe_layer1");
    if (rc) return rc;
    GEOMAE_REQUIRE(m0 && voxel_feats && a->scale0 && a->shift0 && a->scale1 && a->shift1, "vfe_layer1: null argument");
    GEOMAE_ZERO(voxel_feats, (size_t)a->max_pillars * 128 * sizeof(float), stream);
    if (a->pillar_ties) GEOMAE_ZERO(a->pillar_ties, (size_t)a->max_pillars, stream);
    hipLaunchKernelGGL(VFE_L1_FORM(a, vfe_layer1_kernel), vfe_grid(a), dim3(kVfeBlk), 0, stream, G, W, m0, voxel_feats, a->pillar_ties,
                       BnFoldDev{}, (const double*)nullptr);
    return check_launch("vfe_layer1_kernel");
}

extern "C" int geomae_vfe_layer1_bn(const GeomaeVfeArgs* a, const GeomaeBnFold* bn1, const double* sums1, const float* m0,
                                    float* voxel_feats, hipStream_t stream) {
    VfeGeo G; VfeW W;
    int rc = vfe_common(a, &G, &W, "vfe_layer1_bn");
    if (rc) return rc;
    BnFoldDev F;
    if ((rc = bn_fold_of(bn1, &F, "vfe_layer1_bn"))) return rc;
    GEOMAE_REQUIRE(m0 && voxel_feats && sums1 && a->scale0 && a->shift0 && a->scale1 == bn1->scale && a->shift1 == bn1->shift,
                   "vfe_layer1_bn: null argument, or args->scale1 / shift1 are not the fold's outputs");
    GEOMAE_ZERO(voxel_feats, (size_t)a->max_pillars * 128 * sizeof(float), stream);
    if (a->pillar_ties) GEOMAE_ZERO(a->pillar_ties, (size_t)a->max_pillars, stream);
    hipLaunchKernelGGL(VFE_L1_FORM(a, vfe_layer1_kernel), vfe_grid(a), dim3(kVfeBlk), 0, stream, G, W, m0, voxel_feats, a->pillar_ties, F, sums1);
    return check_launch("vfe_layer1_kernel");
}

static int bn_of(const GeomaeBnState* b, int which, const float** scale, const float** shift, const float** mean,
                 const float** invstd) {
    GEOMAE_REQUIRE(b, "vfe backward: null bn state");
    *scale = which ? b->scale1 : b->scale0; *shift = which ? b->shift1 : b->shift0;
    *mean = which ? b->mean1 : b->mean0; *invstd = which ? b->invstd1 : b->invstd0;
    GEOMAE_REQUIRE(*scale && *shift && *mean && *invstd, "vfe backward: null bn state pointer");
    return GEOMAE_OK;
}

extern "C" int geomae_vfe_backward_stats(const GeomaeVfeArgs* a, const GeomaeBnState* bnst, const float* m0,
                                         const float* voxel_feats, const float* d_voxel_feats, double* bsums1,
                                         hipStream_t stream) {
    VfeGeo G; VfeW W;
    int rc = vfe_common(a, &G, &W, "vfe_backward_stats");
    if (rc) return rc;
    Bn1 bn;
    if ((rc = bn_of(bnst, 1, &bn.scale, &bn.shift, &bn.mean, &bn.invstd))) return rc;
    GEOMAE_REQUIRE(m0 && voxel_feats && d_voxel_feats && bsums1, "vfe_backward_stats: null argument");
    GEOMAE_ZERO(bsums1, 256 * sizeof(double), stream);
    if (a->pillar_ties) {
        int gx = cdiv(a->max_pillars, 32 * 8);                        // ~8 pillars per pillar lane, <= 256 workgroups
        gx = gx < 1 ? 1 : (gx > 256 ? 256 : gx);
        hipLaunchKernelGGL(vfe_bwd_stats1_pillars_kernel, dim3(gx), dim3(1024), 0, stream, voxel_feats, d_voxel_feats,
                           (int)a->max_pillars, bn, (const unsigned char*)a->pillar_ties, bsums1);
        const int rc2 = check_launch("vfe_bwd_stats1_pillars_kernel");
        if (rc2) return rc2;
    }
    hipLaunchKernelGGL(VFE_L1_FORM(a, vfe_bwd_stats1_kernel), vfe_grid(a), dim3(kVfeBlk), 0, stream, G, W, m0, voxel_feats,
                       d_voxel_feats, bn, bsums1, (const unsigned char*)a->pillar_ties);
    return check_launch("vfe_bwd_stats1_kernel");
}

extern "C" int geomae_vfe_backward_layer1(const GeomaeVfeArgs* a, const GeomaeBnState* bnst, const float* m0,
                                          const float* voxel_feats, const float* d_voxel_feats,
                                          const double* bsums1_global, float n_eff, void* dy1_bf16, void* g_bf16,
                                          float* dy1_f32, float* dh0, float* dm0, double* bsums0,
                                          float* d_beta1, float* d_gamma1, hipStream_t stream) {
    VfeGeo G; VfeW W;
    int rc = vfe_common(a, &G, &W, "vfe_backward_layer1");
    if (rc) return rc;
    Bn1 bn; Bn0 bn0;
    if ((rc = bn_of(bnst, 1, &bn.scale, &bn.shift, &bn.mean, &bn.invstd))) return rc;
    if ((rc = bn_of(bnst, 0, &bn0.scale, &bn0.shift, &bn0.mean, &bn0.invstd))) return rc;
    GEOMAE_REQUIRE(m0 && voxel_feats && d_voxel_feats && bsums1_global && dy1_bf16 && g_bf16 && dh0 && dm0 &&
                   bsums0 && n_eff > 0, "vfe_backward_layer1: null argument");
    GEOMAE_ZERO(bsums0, 128 * sizeof(double), stream);
    GEOMAE_ZERO(dm0, (size_t)a->max_pillars * 64 * sizeof(float), stream);
    hipLaunchKernelGGL(VFE_L1_FORM(a, vfe_bwd_layer1_kernel), vfe_grid(a), dim3(kVfeBlk), 0, stream, G, W, m0, voxel_feats, d_voxel_feats,
                       bn, bsums1_global, n_eff, (bf16_t*)dy1_bf16, (bf16_t*)g_bf16, dy1_f32, dh0, dm0, d_beta1, d_gamma1);
    hipEvent_t mid = take_mid_launch_event();          // dy1 / g are complete here: the caller's dW1 contraction may start
    hipStream_t side = take_mid_launch_side();         // ... and the routing sweep may leave the caller's stream
    if ((rc = check_launch("vfe_bwd_layer1_kernel"))) return rc;
    if (mid) GEOMAE_HIP(hipEventRecord(mid, stream));
    if (mid && side) {
        GEOMAE_HIP(hipStreamWaitEvent(side, mid, 0));
        stream = side;
    }
    if (a->moments && a->dw0_acc) {
        GEOMAE_ZERO(a->dw0_acc, 64 * 16 * sizeof(float), stream);
        hipLaunchKernelGGL(vfe_bwd_route0_kernel<true>, vfe_grid(a), dim3(kVfeBlk), 0, stream, G, W, m0, bn0, (const float*)dm0,
                           dh0, bsums0, a->moments, 1.0f / (float)a->num_points, a->dw0_acc);
    } else {
        hipLaunchKernelGGL(vfe_bwd_route0_kernel<false>, vfe_grid(a), dim3(kVfeBlk), 0, stream, G, W, m0, bn0, (const float*)dm0,
                           dh0, bsums0, (const double*)nullptr, 0.f, (float*)nullptr);
    }
    return check_launch("vfe_bwd_route0_kernel");
}

extern "C" int geomae_vfe_backward_layer0(const GeomaeVfeArgs* a, const GeomaeBnState* bnst, const float* dh0,
                                          const double* bsums0_global, float n_eff, int64_t num_points,
                                          const void* dy1_bf16, const void* g_bf16, float* dw0, float* dw1,
                                          float* d_beta0, float* d_gamma0, hipStream_t stream) {
    VfeGeo G; VfeW W;
    int rc = vfe_common(a, &G, &W, "vfe_backward_layer0");
    if (rc) return rc;
    Bn0 bn0;
    if ((rc = bn_of(bnst, 0, &bn0.scale, &bn0.shift, &bn0.mean, &bn0.invstd))) return rc;
    GEOMAE_REQUIRE(dh0 && bsums0_global && dw0 && n_eff > 0 && (!dw1 || (dy1_bf16 && g_bf16)),
                   "vfe_backward_layer0: null argument");
    GEOMAE_REQUIRE((d_beta0 == nullptr) == (d_gamma0 == nullptr), "vfe_backward_layer0: pass both BN gradients or none");
    if (a->moments && a->dw0_acc) {
        hipLaunchKernelGGL(vfe_dw0_finalize_kernel, dim3(1), dim3(1024), 0, stream, (const float*)a->dw0_acc, a->moments,
                           bsums0_global, (double)a->num_points, (double)n_eff, a->w0, bn0, dw0, d_beta0, d_gamma0);
    } else {
        hipLaunchKernelGGL(vfe_bwd_layer0_kernel, vfe_grid(a), dim3(kVfeBlk), 0, stream, G, W, dh0, bn0, bsums0_global, n_eff,
                           dw0, d_beta0, d_gamma0);
    }
    rc = check_launch("vfe_bwd_layer0_kernel");
    if (rc || !dw1) return rc;                       // dw1 == NULL: the caller runs geomae_vfe_weight_grad1 itself
    return geomae_vfe_weight_grad1(dy1_bf16, g_bf16, num_points, dw1, stream);
}

// The layer-form contraction addresses its operand slabs with 32-bit byte offsets inside a buffer descriptor of
// n16 * ldblk * 512 bytes (dw_device.h dl_job_body; ldblk = 8 for 128-channel operands): 256 bytes per row must stay below
// 2^31 -- 8.38 M rows per call.  Above it the SPLIT job is not taken.
static constexpr int64_t kDwSplitMaxRows = ((1ll << 31) - (1 << 20)) / 256;

extern "C" int geomae_vfe_weight_grad1(const void* dy1_bf16, const void* g_bf16, int64_t num_points, float* dw1,
                                       hipStream_t stream) {
    if (num_points <= 0) return GEOMAE_OK;
    GEOMAE_REQUIRE(dy1_bf16 && g_bf16 && dw1, "vfe_weight_grad1: null argument");
    DwTasks T;
    T.t[0] = {(const bf16_t*)dy1_bf16, 128, 0, (const bf16_t*)g_bf16, 128, 0, dw1, 128, 0, 0, nullptr, 128};
    T.blocked = 1;                   // (as vfe_bwd_layer1_kernel writes them)
    T.partial = dw_partial();        // a caller's split-K workspace (csrc/engine.hip), summed by its next geomae_flush_weight_grad
    // with a workspace: the layer-form contraction (LDS-direct operand slabs, its own reduction launch); GEOMAE_DW_LAYER_FORM=0:
    // the old kernel (A/B)
    const bool layer_form = tuning().dw_layer_form != 0;
    if (layer_form && T.partial && num_points <= kDwSplitMaxRows)
        return launch_dw_split((const bf16_t*)dy1_bf16, (const bf16_t*)g_bf16, (int)num_points, dw1, T.partial, stream);
    GEOMAE_REQUIRE(num_points < (1ll << 31) - 64, "vfe_weight_grad1: too many points");
    return launch_dw(T, 1, (int)num_points, stream);
}

extern "C" int64_t geomae_vfe_weight_grad1_workspace_bytes(void) { return 2 * kDwPartialBytes; }

extern "C" int geomae_vfe_weight_grad1_ws(const void* dy1_bf16, const void* g_bf16, int64_t num_points, float* dw1,
                                          void* workspace, int64_t workspace_bytes, hipStream_t stream) {
    if (num_points <= 0) return GEOMAE_OK;
    GEOMAE_REQUIRE(dy1_bf16 && g_bf16 && dw1 && workspace, "vfe_weight_grad1_ws: null argument");
    GEOMAE_REQUIRE(num_points <= kDwSplitMaxRows, "vfe_weight_grad1_ws: more than 8.38 M points per call (32-bit slab offsets)");

    if (workspace_bytes < 2 * kDwPartialBytes) {
        set_error("vfe_weight_grad1_ws: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)(2 * kDwPartialBytes));
        return GEOMAE_ERR_WORKSPACE;
    }
    return launch_dw_split((const bf16_t*)dy1_bf16, (const bf16_t*)g_bf16, (int)num_points, dw1, (float*)workspace, stream);
}

