// Fused decoder heads + the six GeoMAE losses + their backward (gfx950).
//
// Reference: the six nn.Linear heads on the masked rows of the two decoder outputs
// (mmdet3d/models/backbones/multi_mae_sst_spearate_top_only.py:279-300) followed by forward_loss
// (mmdet3d/models/detectors/multi_sub_voxel_dynamic_voxelnet_ssl.py:837-902): masked MSE on the
// low / med / top sub-voxel centroids, MSE on the surface normal ("loss_curv_around"), and two
// sigmoid cross-entropies on the occupancy logits (mmdet CrossEntropyLoss(use_sigmoid=True)).
// That is ~18 GEMMs and ~60 elementwise / reduction / boolean-index launches per iteration (with host
// syncs for the masked sizes).  Here: ONE kernel.  A wave owns 16 masked pillars in the T-layout of
// sst_layer.hip; the 800 head outputs are produced 128 at a time from an LDS-staged weight chunk, turned
// into loss terms and d(logit) in registers, and immediately contracted back through the SAME LDS tile
// into d(decoder output) -- the [M,726] prediction tensors never exist.  Head weight gradients are the
// token contraction dl^T x, done by dw_kernel from the bf16 dl / x copies this kernel writes.
#include <type_traits>
#include "common.h"
#include "../../include/geomae_hip.h"
#include "sst_device.h"

namespace geomae {

// output layout (rows of the packed head matrix, columns of dl):
//   [0,384) reg_low | [384,640) cls_low | [640,688) reg_med | [688,720) cls_med | [720,723) reg_top |
//   [723,768) zero  | [768,771) nor_top (input = density decoder) | [771,800) zero
constexpr int kHeadRows = 800;
constexpr int kDlLd = 896;          // leading dimension of dl (padded so 128-wide dw blocks never overrun)

struct HeadArgs {
    const float* cen; const float* den;       // [n,128] decoder outputs; masked rows start at n_keep
    int n_keep, M;
    const bf16_t* wp;                         // packed [800][128]
    const float* bias;                        // [800]
    const float* t_low; const uint8_t* m_low; // [M,384], [M,128]
    const float* t_med; const uint8_t* m_med; // [M,48], [M,16]
    const float* t_top; const float* t_nor;   // [M,3]
    const int32_t* occ;                       // [2] occupied low / med cells over the M rows
    float w_low, w_med, w_top, w_nor, w_cls_low, w_cls_med;
    float* loss;                              // [6] curv_around, centroid_low, centroid_med, centroid_top, cls_low, cls_med
    float* d_cen; float* d_den;               // [n,128], rows >= n_keep written (others pre-zeroed by caller)
    float* d_cen2;                            // split mode (see heads_loss_kernel): second summand of d_cen, or nullptr
    bf16_t* dl; bf16_t* cm_b; bf16_t* dm_b;   // [M,896], [M,128], [M,128]
    int part = 0;                             // 0: every head; 1: the centroid decoder's heads (chunks 0-5, split in two
                                              //    workgroup kinds); 2: the density decoder's head (chunk 6) -- see
                                              //    geomae_heads_loss_centroid_accumulate / _density_accumulate
};


// BCE with logits against y in {0,1}: value and d/dx
__device__ __forceinline__ void bce(float x, float y, float* l, float* d) {
    const float e = __expf(-fabsf(x));
    *l = fmaxf(x, 0.f) - x * y + __logf(1.f + e);
    const float r = __builtin_amdgcn_rcpf(1.f + e);   // v_rcp_f32 (1 ulp); the IEEE divide is a 12-instruction sequence
    const float sig = x >= 0.f ? r : e * r;
    *d = sig - y;
}

// dX (normal-orientation C layout: lane (c = l&15, g) holds tokens 4g+r of channel tile ct) +=
//   dl[16 x 128 outputs] * W[128 outputs x 128 channels], W read from the staged chunk (K-permuted columns)
// The B operand is a COLUMN of the staged row-major chunk (8 output rows of one LDS column per lane): read with
// ds_read_b64_tr_b16 (two per MFMA; it was eight 2-byte LDS reads plus their repacking).  LDS column j holds
// channel kperm(j) (the chunk is staged K-permuted for the forward GEMM), so column c of accumulator tile ct is
// channel kperm(16 ct + c): the caller un-permutes in its store index.
template <int KK = 4>                     // 32-row groups of the staged chunk that hold weights (the normal head's chunk: 1)
__device__ __forceinline__ void accumulate_dx(const bf16_t* __restrict__ smem, const uint2 (&dlb)[8],
                                              f32x4 (&acc)[8], int lane) {
    constexpr int LD = 128 + kPad;
    const int m = lane & 15, g = lane >> 4;
    // lane m of a 16-lane group addresses row (m >> 2), columns 4 (m & 3).. of a [4 rows x 16 columns] block and
    // receives column m of it: rows = this lane group's output rows 32 kk + 4 g + 0..3 (lo) and + 16 (hi)
    const bf16_t* base = smem + (4 * g + (m >> 2)) * LD + 4 * (m & 3);
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        const uint4 a = make_uint4(dlb[2 * kk].x, dlb[2 * kk].y, dlb[2 * kk + 1].x, dlb[2 * kk + 1].y);
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
            const uint2 lo = tr_read(base + (32 * kk) * LD + 16 * ct), hi = tr_read(base + (32 * kk + 16) * LD + 16 * ct);
            acc[ct] = mfma32(a, make_uint4(lo.x, lo.y, hi.x, hi.y), acc[ct]);
        }
    }
}

// MODE 0: all seven 128-output chunks in one workgroup (the original form).  MODE 1 / 2: the two halves of the split
// form -- chunks 0-2 (the 384 low-level regression outputs) and chunks 3-6 (class logits, med / top outputs, the
// density decoder's normal head).  At 15 k masked pillars the single form is 241 workgroups of one wave per SIMD, each a
// 72 us chain of 7 x (GEMM -> loss arithmetic -> dX GEMM); the split form is 482 workgroups, two per CU, each half as
// long.  Both halves contribute to d(centroid decoder output): they write SEPARATE zero-initialised buffers (d_cen,
// d_cen2) that the decoder's backward sums while loading (geomae_sst_stack_backward dz + dz_add) -- no atomics.
// MODE 3: chunk 6 alone (the only head that reads the density decoder); MODE 4: chunks 3-5 (MODE 2 without chunk 6): with
// MODE 1 + 4 in one launch and MODE 3 in another, each decoder stream runs its own heads and goes straight on into its
// backward -- no stream waits for the other decoder's forward.
template <int MODE>
__device__ __forceinline__ void heads_loss_body(const HeadArgs& A, bf16_t* __restrict__ smem, float (*red)[6]) {
    constexpr int kFirst = MODE == 3 ? 6 : ((MODE == 2 || MODE == 4) ? 3 : 0), kLast = MODE == 1 ? 3 : (MODE == 4 ? 6 : 7);
    float* const d_cen_out = (MODE == 2 || MODE == 4) ? A.d_cen2 : A.d_cen;
    constexpr int SB = MODE == 1 ? 0 : (MODE == 4 ? 10 : 20);         // phase stamps (timing build): per kind of workgroup
    GEOMAE_STAMP(SB);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4;
    const int tile = blockIdx.x * (kLayerBlk / 64) + wave;
    const int64_t row = (int64_t)tile * 16 + (lane & 15);     // masked-row index
    const bool valid = row < A.M;
    const int64_t tok = A.n_keep + row;
    const float inv_low = MODE == 3 ? 0.f : A.w_low / fmaxf(1.f, (float)A.occ[0]);      // (MODE 3: no occupancy counts passed)
    const float inv_med = MODE == 3 ? 0.f : A.w_med / fmaxf(1.f, (float)A.occ[1]);
    const float inv_top = A.w_top / (float)A.M, inv_nor = A.w_nor / (float)A.M;
    const float inv_cl = A.w_cls_low / ((float)A.M * 256.f), inv_cm = A.w_cls_med / ((float)A.M * 32.f);
    float l_nor = 0.f, l_low = 0.f, l_med = 0.f, l_top = 0.f, l_cl = 0.f, l_cm = 0.f;

    // the first chunk's weights before anything else: the oldest entry of the in-order memory counter, in flight together
    // with the rows (the rows first, then the weights: 28-35 k cycles until the first GEMM was done)
    WStage<128, 128> st;                                        // (chunk 6 stages its 32 rows through the first quarter)
    if (kFirst < 6) stage_issue<128, 128>(A.wp + (size_t)(128 * kFirst) * 128, st);
    else stage_issue<128, 32>(A.wp + (size_t)768 * 128, reinterpret_cast<WStage<128, 32>&>(st));
    __builtin_amdgcn_sched_barrier(0);
    uint2 xb[8];
    if (MODE != 3) {
        f32x4 x[8];
        load_rows_f32<128>(A.cen, A.n_keep + A.M, (int)tok, x, lane);
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) xb[ct] = pack4(x[ct]);
        if (MODE != 2 && MODE != 4) store_rows_bf16<128>(A.cm_b, A.M, (int)row, 128, 0, x, lane);
    }
    f32x4 dx[8];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) dx[ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Software pipeline over the chunks (everything below is resolved at compile time: `chunk` is a constant of the unrolled
    // loop).  The weights of chunk c + 1 are requested right after chunk c's matrix is in LDS (their staging registers are
    // free then) and its targets after chunk c's loss arithmetic (their registers are free then): both fly under the rest of
    // chunk c, and the weights are OLDER than the targets in the in-order memory counter, so the LDS commit of chunk c + 1
    // does not wait for target rows.  (Before: targets, then weights, requested at the top of their own chunk -- "loads +
    // GEMM" 10-12 k cycles per chunk for a 32-MFMA GEMM -- and the medium / top targets of chunk 5 loaded element by
    // element inside the loss loop, a mask load, a branch and a dependent target load each: 30 k cycles,
    // tools/heads_time.py.)
    f32x4 tl[8];                                                // regression targets of this lane's 32 outputs (chunks 0-2, 5)
    unsigned int mk[8];                                         // per ct: mask / class bytes (see targets_issue)
    auto targets_issue = [&](auto CH) {
        constexpr int chunk = decltype(CH)::value;
        if (chunk < 5) {
            const __amdgpu_buffer_rsrc_t mr = rows_rsrc(A.m_low, A.M, 128);
            if (chunk < 3) {
                const RowAddr ta = row_addr<4>(A.t_low, A.M, (int)row, 384, 128 * chunk, lane, false);
#pragma unroll
                for (int ct = 0; ct < 8; ++ct) {
                    tl[ct] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ta.r, ta.voff, ct * ta.ct_stride, 0));
                    // outputs og .. og+3 belong to sub-voxels og/3 and (og+3)/3 (equal or consecutive)
                    const int og = 128 * chunk + 16 * ct + 4 * g;
                    const unsigned int b0 = __builtin_amdgcn_raw_buffer_load_b8(mr, (int)row * 128 + og / 3, 0, 0);
                    const unsigned int b1 = __builtin_amdgcn_raw_buffer_load_b8(mr, (int)row * 128 + (og + 3) / 3, 0, 0);
                    mk[ct] = b0 | (b1 << 8);
                }
            } else {
#pragma unroll
                for (int ct = 0; ct < 8; ++ct) {                // class outputs og .. og+3: sub-voxels og/2, og/2 + 1
                    const int og = 128 * (chunk - 3) + 16 * ct + 4 * g;
                    mk[ct] = __builtin_amdgcn_raw_buffer_load_b16(mr, (int)row * 128 + (og >> 1), 0, 0);
                }
            }
        } else if (chunk == 5) {
            // outputs 0-47: medium regression (targets t_med [M,48], masks m_med [M,16]: bytes o/3, (o+3)/3);
            // 48-79: medium class logits (bytes (o-48)/2, +1); 80-82: the top centroid (t_top [M,3])
            const __amdgpu_buffer_rsrc_t mm = rows_rsrc(A.m_med, A.M, 16), tm = rows_rsrc(A.t_med, A.M, 192);
            const __amdgpu_buffer_rsrc_t tt = rows_rsrc(A.t_top, A.M, 12);
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) {
                const int o = 16 * ct + 4 * g;
                tl[ct] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(tm, (int)row * 192 + 4 * o, 0, 0));
                const unsigned int b0 = __builtin_amdgcn_raw_buffer_load_b8(mm, (int)row * 16 + o / 3, 0, 0);
                const unsigned int b1 = __builtin_amdgcn_raw_buffer_load_b8(mm, (int)row * 16 + (o + 3) / 3, 0, 0);
                mk[ct] = b0 | (b1 << 8);
            }
#pragma unroll
            for (int ct = 3; ct < 5; ++ct) mk[ct] = __builtin_amdgcn_raw_buffer_load_b16(mm, (int)row * 16 + ((16 * ct + 4 * g - 48) >> 1), 0, 0);
            tl[5] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (g == 0) {                                       // outputs 80, 81, 82 (+ one unused)
                tl[5][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(tt, (int)row * 12, 0, 0));
                tl[5][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(tt, (int)row * 12 + 4, 0, 0));
                tl[5][2] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(tt, (int)row * 12 + 8, 0, 0));
            }
        } else {
            tl[0] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (g == 0) {                                       // the normal's three components
                const __amdgpu_buffer_rsrc_t tn = rows_rsrc(A.t_nor, A.M, 12);
                tl[0][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(tn, (int)row * 12, 0, 0));
                tl[0][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(tn, (int)row * 12 + 4, 0, 0));
                tl[0][2] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(tn, (int)row * 12 + 8, 0, 0));
            }
        }
    };
    targets_issue(std::integral_constant<int, kFirst>{});
#pragma unroll
    for (int chunk = kFirst; chunk < kLast; ++chunk) {
        const int row0 = chunk < 6 ? 128 * chunk : 768;         // first output row of this chunk
        if (chunk == 6 && MODE != 3) {
            // flush d_cen, switch the input to the density decoder
#pragma unroll
            for (int ct = 0; ct < 8; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t rr = (int64_t)tile * 16 + 4 * g + r;
                    if (rr < A.M) d_cen_out[(A.n_keep + rr) * 128 + kperm(16 * ct + (lane & 15))] = dx[ct][r];
                }
        }
        if (chunk == 6) {
            f32x4 x[8];
            load_rows_f32<128>(A.den, A.n_keep + A.M, (int)tok, x, lane);
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) {
                xb[ct] = pack4(x[ct]);
                dx[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            store_rows_bf16<128>(A.dm_b, A.M, (int)row, 128, 0, x, lane);
        }
        f32x4 z[8];
        if (chunk < 6) {
            load_bias<128>(A.bias + row0, z, lane);
            gemm_staged<128, 128>(st, smem, xb, z, lane);
        } else {
            // last chunk holds 32 rows (nor_top + zero padding); the other 96 rows of the tile are stale
            // weights multiplied by dl = 0 below
            load_bias<32>(A.bias + row0, reinterpret_cast<f32x4(&)[2]>(z), lane);
#pragma unroll
            for (int ct = 2; ct < 8; ++ct) z[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
            gemm_staged<128, 32>(reinterpret_cast<WStage<128, 32>&>(st), smem, xb, reinterpret_cast<f32x4(&)[2]>(z), lane);
        }
        if (chunk + 1 < kLast) {                                // the next chunk's weights: in flight under this chunk's arithmetic
            if (chunk + 1 < 6) stage_issue<128, 128>(A.wp + (size_t)(128 * (chunk + 1)) * 128, st);
            else stage_issue<128, 32>(A.wp + (size_t)768 * 128, reinterpret_cast<WStage<128, 32>&>(st));
        }
        GEOMAE_STAMP(SB + 1 + 3 * ((chunk - kFirst) % 3));
        // ---- loss terms and d(logit) for the 32 outputs this lane holds
        uint2 dlb[8];
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
            if (valid) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = 16 * ct + 4 * g + r;          // output within the chunk
                    const float x = z[ct][r];
                    if (chunk < 3) {
                        const int og = 128 * chunk + o;
                        const unsigned int m = ((og / 3 == (og - r) / 3) ? mk[ct] : (mk[ct] >> 8)) & 0xffu;
                        if (m) {
                            const float df = x - tl[ct][r];
                            l_low += df * df * (1.f / 3.f);
                            d[r] = df * (2.f / 3.f) * inv_low;
                        }
                    } else if (chunk < 5) {
                        const int og = 128 * (chunk - 3) + o;
                        const int cls = (int)((mk[ct] >> (8 * (r >> 1))) & 0xffu);
                        const float y = (cls == (og & 1)) ? 1.f : 0.f;
                        float l, dd;
                        bce(x, y, &l, &dd);
                        l_cl += l;
                        d[r] = dd * inv_cl;
                    } else if (chunk == 5) {
                        if (ct < 3) {                            // o < 48
                            const unsigned int m = ((o / 3 == (o - r) / 3) ? mk[ct] : (mk[ct] >> 8)) & 0xffu;
                            if (m) {
                                const float df = x - tl[ct][r];
                                l_med += df * df * (1.f / 3.f);
                                d[r] = df * (2.f / 3.f) * inv_med;
                            }
                        } else if (ct < 5) {                     // 48 <= o < 80
                            const int og = o - 48;
                            const int cls = (int)((mk[ct] >> (8 * (r >> 1))) & 0xffu);
                            const float y = (cls == (og & 1)) ? 1.f : 0.f;
                            float l, dd;
                            bce(x, y, &l, &dd);
                            l_cm += l;
                            d[r] = dd * inv_cm;
                        } else if (ct == 5 && g == 0 && r < 3) { // 80 <= o < 83
                            const float df = x - tl[5][r];
                            l_top += df * df * (1.f / 3.f);
                            d[r] = df * (2.f / 3.f) * inv_top;
                        }
                    } else if (ct == 0 && g == 0 && r < 3) {     // chunk 6: o < 3
                        const float df = x - tl[0][r];
                        l_nor += df * df * (1.f / 3.f);
                        d[r] = df * (2.f / 3.f) * inv_nor;
                    }
                }
            }
            dlb[ct] = pack4(d);
            if (valid && (chunk < 6 || ct < 2))
                *reinterpret_cast<uint2*>(A.dl + row * kDlLd + row0 + 16 * ct + 4 * g) = dlb[ct];
        }
        GEOMAE_STAMP(SB + 2 + 3 * ((chunk - kFirst) % 3));
        // the next chunk's targets: their registers are free now, they fly under the dX GEMM and the next forward GEMM
        if (chunk + 1 < kLast) {
            switch (chunk + 1) {
                case 1: targets_issue(std::integral_constant<int, 1>{}); break;
                case 2: targets_issue(std::integral_constant<int, 2>{}); break;
                case 3: targets_issue(std::integral_constant<int, 3>{}); break;
                case 4: targets_issue(std::integral_constant<int, 4>{}); break;
                case 5: targets_issue(std::integral_constant<int, 5>{}); break;
                default: targets_issue(std::integral_constant<int, 6>{}); break;
            }
        }
        // (chunk 6 stages 32 rows; in a launch of its own -- MODE 3 -- the other 96 rows of the LDS tile are whatever the last
        //  kernel left there, and 0 x NaN is NaN: only the 32 rows are read.  Behind chunk 5 they are stale finite weights
        //  against dl = 0, as before.)
        if (MODE == 3) accumulate_dx<1>(smem, dlb, dx, lane);
        else accumulate_dx(smem, dlb, dx, lane);
        GEOMAE_STAMP(SB + 3 + 3 * ((chunk - kFirst) % 3));
    }
    float* const d_last = MODE == 1 ? A.d_cen : (MODE == 4 ? A.d_cen2 : A.d_den);   // MODE 1 / 4 end on the centroid decoder's chunks
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t rr = (int64_t)tile * 16 + 4 * g + r;
            if (rr < A.M) d_last[(A.n_keep + rr) * 128 + kperm(16 * ct + (lane & 15))] = dx[ct][r];
        }
    // zero the padding columns of dl that dw_kernel will read: [771,896)
    if (valid && MODE != 1 && MODE != 4)
        for (int cidx = 800 + 4 * g; cidx < kDlLd; cidx += 16)
            *reinterpret_cast<uint2*>(A.dl + row * kDlLd + cidx) = make_uint2(0u, 0u);
    // ---- losses: wave reduce, block reduce, one atomic per loss
    float ls[6] = {l_nor * inv_nor, l_low * inv_low, l_med * inv_med, l_top * inv_top, l_cl * inv_cl, l_cm * inv_cm};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float v = wave_sum(ls[k]);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 6) atomicAdd(A.loss + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] +
                                                          red[2][threadIdx.x] + red[3][threadIdx.x]);
    GEOMAE_STAMP(MODE == 1 ? 30 : (MODE == 4 ? 31 : 29));
}

__global__ __launch_bounds__(kLayerBlk, 2) void heads_loss_kernel(HeadArgs A) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[kWeightLds];
    __shared__ float red[4][6];
    if (A.part == 2) heads_loss_body<3>(A, smem, red);
    else if (A.part == 1) { if (blockIdx.y == 0) heads_loss_body<1>(A, smem, red); else heads_loss_body<4>(A, smem, red); }
    else if (A.d_cen2 == nullptr) heads_loss_body<0>(A, smem, red);
    else if (blockIdx.y == 0) heads_loss_body<1>(A, smem, red);
    else heads_loss_body<2>(A, smem, red);
}

}  // namespace geomae

using namespace geomae;

#ifdef GEOMAE_PHASE_TIMING
extern "C" void geomae_debug_read_heads_stamps(unsigned long long* out, int clear) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(geomae::geomae_stamps), sizeof(unsigned long long) * GEOMAE_STAMP_BLOCKS * GEOMAE_STAMP_SLOTS);
    if (clear) {
        static unsigned long long zeros[GEOMAE_STAMP_BLOCKS * GEOMAE_STAMP_SLOTS];
        (void)hipMemcpyToSymbol(HIP_SYMBOL(geomae::geomae_stamps), zeros, sizeof(zeros));
    }
}
#endif

static int heads_loss_launch(const float* dec_centroid, const float* dec_density, int32_t num_keep,
                             int32_t num_mask, const void* head_w_packed, const float* head_bias,
                             const float* centroid_low, const uint8_t* mask_low, const float* centroid_med,
                             const uint8_t* mask_med, const float* centroid_top, const float* normal,
                             const int32_t* occ_counts, const float* loss_weights, float* losses,
                             float* d_dec_centroid, float* d_dec_centroid2, float* d_dec_density, void* dlogits_bf16,
                             void* cm_bf16, void* dm_bf16, hipStream_t stream, int part = 0) {
    if (num_mask <= 0) return GEOMAE_OK;
    GEOMAE_REQUIRE(head_w_packed && head_bias && loss_weights && losses && dlogits_bf16, "heads_loss: null argument");
    GEOMAE_REQUIRE(part == 2 || (dec_centroid && centroid_low && mask_low && centroid_med && mask_med && centroid_top &&
                                 occ_counts && d_dec_centroid && cm_bf16), "heads_loss: null argument (centroid heads)");
    GEOMAE_REQUIRE(part == 1 || (dec_density && normal && d_dec_density && dm_bf16), "heads_loss: null argument (density head)");
    GEOMAE_REQUIRE(part != 1 || d_dec_centroid2, "heads_loss: the centroid part needs both summand buffers");
    HeadArgs A;
    A.part = part;
    A.cen = dec_centroid; A.den = dec_density; A.n_keep = num_keep; A.M = num_mask;
    A.wp = (const bf16_t*)head_w_packed; A.bias = head_bias;
    A.t_low = centroid_low; A.m_low = mask_low; A.t_med = centroid_med; A.m_med = mask_med;
    A.t_top = centroid_top; A.t_nor = normal; A.occ = occ_counts;
    A.w_nor = loss_weights[0]; A.w_low = loss_weights[1]; A.w_med = loss_weights[2]; A.w_top = loss_weights[3];
    A.w_cls_low = loss_weights[4]; A.w_cls_med = loss_weights[5];
    A.loss = losses; A.d_cen = d_dec_centroid; A.d_den = d_dec_density; A.d_cen2 = d_dec_centroid2;
    A.dl = (bf16_t*)dlogits_bf16; A.cm_b = (bf16_t*)cm_bf16; A.dm_b = (bf16_t*)dm_bf16;
    const int tiles = cdiv(num_mask, 16);
    hipLaunchKernelGGL(heads_loss_kernel, dim3(cdiv(tiles, kLayerBlk / 64), part == 2 ? 1 : (d_dec_centroid2 ? 2 : 1)),
                       dim3(kLayerBlk), 0, stream, A);
    return check_launch("heads_loss_kernel");
}

// `losses` is ACCUMULATED into (atomics): the caller zeroes it, e.g. off the critical path
extern "C" int geomae_heads_loss_accumulate(const float* dec_centroid, const float* dec_density, int32_t num_keep,
                                 int32_t num_mask, const void* head_w_packed, const float* head_bias,
                                 const float* centroid_low, const uint8_t* mask_low, const float* centroid_med,
                                 const uint8_t* mask_med, const float* centroid_top, const float* normal,
                                 const int32_t* occ_counts, const float* loss_weights, float* losses,
                                 float* d_dec_centroid, float* d_dec_density, void* dlogits_bf16, void* cm_bf16,
                                 void* dm_bf16, hipStream_t stream) {
    return heads_loss_launch(dec_centroid, dec_density, num_keep, num_mask, head_w_packed, head_bias, centroid_low, mask_low,
                             centroid_med, mask_med, centroid_top, normal, occ_counts, loss_weights, losses, d_dec_centroid,
                             nullptr, d_dec_density, dlogits_bf16, cm_bf16, dm_bf16, stream);
}

// split form: twice the workgroups, each half the chain.  d_dec_centroid and d_dec_centroid2 (both [n,128], zeroed by
// the caller) are the two summands of the centroid decoder's output gradient: hand them to
// geomae_sst_stack_backward as dz and dz_add.
extern "C" int geomae_heads_loss_split_accumulate(const float* dec_centroid, const float* dec_density, int32_t num_keep,
                                 int32_t num_mask, const void* head_w_packed, const float* head_bias,
                                 const float* centroid_low, const uint8_t* mask_low, const float* centroid_med,
                                 const uint8_t* mask_med, const float* centroid_top, const float* normal,
                                 const int32_t* occ_counts, const float* loss_weights, float* losses,
                                 float* d_dec_centroid, float* d_dec_centroid2, float* d_dec_density, void* dlogits_bf16,
                                 void* cm_bf16, void* dm_bf16, hipStream_t stream) {
    GEOMAE_REQUIRE(d_dec_centroid2, "heads_loss_split: null d_dec_centroid2");
    return heads_loss_launch(dec_centroid, dec_density, num_keep, num_mask, head_w_packed, head_bias, centroid_low, mask_low,
                             centroid_med, mask_med, centroid_top, normal, occ_counts, loss_weights, losses, d_dec_centroid,
                             d_dec_centroid2, d_dec_density, dlogits_bf16, cm_bf16, dm_bf16, stream);
}

// The heads by decoder: everything that reads the centroid decoder (reg_low, cls_low, reg_med, cls_med, reg_top: five of
// the six losses) and the one head of the density decoder (nor_top: loss_curv_around), as launches of their own -- each on
// its decoder's stream, so that neither stack's backward waits for the other stack's forward.  Disjoint outputs (columns
// [0,768) / [768,896) of dlogits, cm / dm, d_dec_centroid(2) / d_dec_density); `losses` accumulated with atomics.
extern "C" int geomae_heads_loss_centroid_accumulate(const float* dec_centroid, int32_t num_keep, int32_t num_mask,
                                 const void* head_w_packed, const float* head_bias, const float* centroid_low,
                                 const uint8_t* mask_low, const float* centroid_med, const uint8_t* mask_med,
                                 const float* centroid_top, const int32_t* occ_counts, const float* loss_weights,
                                 float* losses, float* d_dec_centroid, float* d_dec_centroid2, void* dlogits_bf16,
                                 void* cm_bf16, hipStream_t stream) {
    return heads_loss_launch(dec_centroid, nullptr, num_keep, num_mask, head_w_packed, head_bias, centroid_low, mask_low,
                             centroid_med, mask_med, centroid_top, nullptr, occ_counts, loss_weights, losses, d_dec_centroid,
                             d_dec_centroid2, nullptr, dlogits_bf16, cm_bf16, nullptr, stream, 1);
}

extern "C" int geomae_heads_loss_density_accumulate(const float* dec_density, int32_t num_keep, int32_t num_mask,
                                 const void* head_w_packed, const float* head_bias, const float* normal,
                                 const float* loss_weights, float* losses, float* d_dec_density, void* dlogits_bf16,
                                 void* dm_bf16, hipStream_t stream) {
    return heads_loss_launch(nullptr, dec_density, num_keep, num_mask, head_w_packed, head_bias, nullptr, nullptr, nullptr,
                             nullptr, nullptr, normal, nullptr, loss_weights, losses, nullptr, nullptr, d_dec_density,
                             dlogits_bf16, nullptr, dm_bf16, stream, 2);
}

extern "C" int geomae_heads_loss(const float* dec_centroid, const float* dec_density, int32_t num_keep,
                                 int32_t num_mask, const void* head_w_packed, const float* head_bias,
                                 const float* centroid_low, const uint8_t* mask_low, const float* centroid_med,
                                 const uint8_t* mask_med, const float* centroid_top, const float* normal,
                                 const int32_t* occ_counts, const float* loss_weights, float* losses,
                                 float* d_dec_centroid, float* d_dec_density, void* dlogits_bf16, void* cm_bf16,
                                 void* dm_bf16, hipStream_t stream) {
    GEOMAE_REQUIRE(losses, "heads_loss: null argument");
    GEOMAE_HIP(hipMemsetAsync(losses, 0, 6 * sizeof(float), stream));
    return geomae_heads_loss_accumulate(dec_centroid, dec_density, num_keep, num_mask, head_w_packed, head_bias,
                                        centroid_low, mask_low, centroid_med, mask_med, centroid_top, normal, occ_counts,
                                        loss_weights, losses, d_dec_centroid, d_dec_density, dlogits_bf16, cm_bf16,
                                        dm_bf16, stream);
}

extern "C" int geomae_heads_weight_grad(int32_t num_mask, const void* dlogits_bf16, const void* cm_bf16,
                                        const void* dm_bf16, const GeomaeHeadGrads* g, hipStream_t stream) {
    if (num_mask <= 0) return GEOMAE_OK;
    GEOMAE_REQUIRE(dlogits_bf16 && cm_bf16 && dm_bf16 && g, "heads_weight_grad: null argument");
    GEOMAE_REQUIRE(g->reg_low_w && g->reg_low_b && g->cls_low_w && g->cls_low_b && g->reg_med_w && g->reg_med_b &&
                   g->cls_med_w && g->cls_med_b && g->reg_top_w && g->reg_top_b && g->nor_top_w && g->nor_top_b,
                   "heads_weight_grad: null gradient pointer");
    const bf16_t *dl = (const bf16_t*)dlogits_bf16, *cm = (const bf16_t*)cm_bf16, *dm = (const bf16_t*)dm_bf16;
    DwTasks T;
    //          A   lda    a0   B   ldb b0  C             ldc  r0   c0 dbias         rows
    T.t[0] = {dl, kDlLd, 0,   cm, 128, 0, g->reg_low_w, 128, 0,   0, g->reg_low_b, 128};
    T.t[1] = {dl, kDlLd, 128, cm, 128, 0, g->reg_low_w, 128, 128, 0, g->reg_low_b, 128};
    T.t[2] = {dl, kDlLd, 256, cm, 128, 0, g->reg_low_w, 128, 256, 0, g->reg_low_b, 128};
    T.t[3] = {dl, kDlLd, 384, cm, 128, 0, g->cls_low_w, 128, 0,   0, g->cls_low_b, 128};
    T.t[4] = {dl, kDlLd, 512, cm, 128, 0, g->cls_low_w, 128, 128, 0, g->cls_low_b, 128};
    T.t[5] = {dl, kDlLd, 640, cm, 128, 0, g->reg_med_w, 128, 0,   0, g->reg_med_b, 48};
    T.t[6] = {dl, kDlLd, 688, cm, 128, 0, g->cls_med_w, 128, 0,   0, g->cls_med_b, 32};
    T.t[7] = {dl, kDlLd, 720, cm, 128, 0, g->reg_top_w, 128, 0,   0, g->reg_top_b, 3};
    T.t[8] = {dl, kDlLd, 768, dm, 128, 0, g->nor_top_w, 128, 0,   0, g->nor_top_b, 3};
    return launch_dw(T, 9, num_mask, stream);
}
