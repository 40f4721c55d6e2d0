// Fused SST encoder-layer kernels for gfx950 (forward and backward), everything except the
// in-window attention core (window.hip).
//
// Reference layer (mmdet3d/models/sst/sst_basic_block.py:63-102 EncoderLayer, :26-61 WindowAttention):
//   q = k = (x + pos) W_qk^T + b ; v = x W_v^T + b ; a = MHA(q,k,v) W_o^T + b_o
//   y = LN1(x + a) ; z = LN2(y + W_2 GELU(W_1 y + b_1) + b_2)
// which the reference runs as ~30 small ATen/cuBLAS kernels per layer forward+backward with the
// activations round-tripping HBM in fp32, and whose weight-gradient GEMMs (K = tokens, M x N = 128..384
// wide) land on 4 workgroups in hipBLASLt (profiles/r01a).
//
// Design (every token row is independent outside attention, so no workgroup ever synchronises in the
// forward/backward data path):
//  * one WAVE owns 16 tokens and keeps them in registers in "T-layout": lane l holds token t = l & 15 and,
//    for every 16-channel tile ct, channels 16*ct + 4*(l>>4) + {0,1,2,3}.  That is exactly the C/D layout
//    of v_mfma_f32_16x16x32_bf16 when the product is computed TRANSPOSED (Y^T = W X^T: A = weights,
//    B = activations), and -- with the contraction index permuted identically in the pre-packed weights --
//    also its B-operand layout.  So projections, residuals, LayerNorm, GELU and their backward chain from
//    registers to registers: fp32 residual stream, bf16 MFMA operands, fp32 accumulation.
//  * weights are packed once per step to bf16 (plain and transposed, K-permuted) by pack_weights_kernel;
//    a wave reads its A fragments as 16-byte lines straight from L2 (a layer's 512 KB of packed weights is
//    L2 resident; there is nothing to stage, so no LDS and no barriers).
//  * weight gradients are token-contractions: dW = dY^T X.  dw_kernel stages 32-token slabs of both
//    operands transposed in LDS, accumulates 128 x 128 output blocks over a token chunk in MFMA
//    accumulators and flushes them with coalesced fp32 atomics (~300 G atomics/s measured,
//    profiles/r01_microbench_atomics.txt); bias gradients are the column sums of the same slabs.
#include "common.h"
#include "../../include/geomae_hip.h"

#include "sst_device.h"
#include "dw_device.h"
#include <vector>

namespace geomae {

// ------------------------------------------------------------------------------------------------
// weight packing: desc[d] = {src_off, rows, cols, transpose, dst_off}; dst [R][K] bf16 with
// dst[r][p] = W[r][kperm(p)] (K = cols) or, transposed (tr & 1), dst[j][p] = W[kperm(p)][j] (R = cols, K = rows);
// tr & 4: the same matrix stored fragment-major (below); tr == 2: plain fp32 gather into aux
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ flat,
                                                           const int64_t* __restrict__ desc,
                                                           bf16_t* __restrict__ packed, float* __restrict__ aux) {
    const int64_t* d = desc + (int64_t)blockIdx.y * 5;
    const int64_t src = d[0], rows = d[1], cols = d[2], tr = d[3], dst = d[4];
    const int64_t total = rows * cols;
    if (tr == 2) {        // plain fp32 gather into the aux vector (e.g. concatenated biases)
        for (int64_t e = blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) aux[dst + e] = flat[src + e];
        return;
    }
    // One thread = 8 consecutive positions of one output row = ONE 16-byte store (K is a multiple of 32; in the fragment-major
    // form the 8 positions of a lane are contiguous too).  Round 4's form -- one 2-byte store per thread, scattered for the
    // transposed and the fragment-major copies -- showed as 110 MB of write traffic per step for 30 MB of copies.  32-bit
    // index arithmetic (a matrix has at most 384 x 256 elements).
    const bool tp = tr & 1, frag = tr & 4;
    const int K = (int)(tp ? rows : cols), C = (int)cols, T8 = (int)(total >> 3);
    const float* __restrict__ w = flat + src;
    bf16_t* __restrict__ out = packed + dst;
    if ((K & 7) || (dst & 7) || (frag && ((rows | cols) & 31))) {
        // a shape the 16-byte form does not cover (none in this library: 128 / 256 / 384 / 800 rows x 128 / 256 columns): element-wise
        const int T = (int)total;
        for (int e = blockIdx.x * 256 + threadIdx.x; e < T; e += gridDim.x * 256) {
            const int r = e / K, p = e - r * K, k = kperm(p);
            const int d = frag ? ((r >> 4) * (K >> 5) + (p >> 5)) * 512 + ((((p >> 3) & 3) << 4) + (r & 15)) * 8 + (p & 7) : e;
            out[d] = (bf16_t)f2bf_bits(tp ? w[k * C + r] : w[r * C + k]);
        }
        return;
    }
    for (int e8 = blockIdx.x * 256 + threadIdx.x; e8 < T8; e8 += gridDim.x * 256) {
        const int e = e8 * 8;
        const int r = e / K;
        const int p = e - r * K;
        unsigned int q[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k0 = kperm(p + 2 * i), k1 = kperm(p + 2 * i + 1);
            const float v0 = tp ? w[k0 * C + r] : w[r * C + k0];
            const float v1 = tp ? w[k1 * C + r] : w[r * C + k1];
            q[i] = (unsigned int)f2bf_bits(v0) | ((unsigned int)f2bf_bits(v1) << 16);
        }
        // fragment-major (tr & 4; the one-launch layer kernels, sst_fused.hip): the 16 rows x 32 positions that one MFMA A
        // fragment of 64 lanes covers are one contiguous 1-KB piece [out tile][k step][lane = 16 g + row][8 positions]
        const int d = frag ? ((r >> 4) * (K >> 5) + (p >> 5)) * 512 + ((((p >> 3) & 3) << 4) + (r & 15)) * 8 : e;
        *reinterpret_cast<uint4*>(out + d) = make_uint4(q[0], q[1], q[2], q[3]);
    }
}

// layout flags of the layer kernels (sst_device.h "Row layouts"); 0 = everything row-major (the plain C-ABI calls)
constexpr int kLayBlocked = 1;    // tensors only these kernels exchange: qkv, attn, the saved activations, the backward slabs
constexpr int kLayXBlocked = 2;   // the layer input x (residual stream)
constexpr int kLayZBlocked = 4;   // the layer output z
constexpr int kLaySavedBf16 = 8;  // the saved normalised activations xhat1 / xhat2 are bf16 [n,128] (the stacks; the per-op
                                  // C ABI documents fp32): 1 KB less per token and layer each way, the backward reads them as
                                  // the LayerNorm's x-hat and re-derives y = xhat1 * g1 + be1 -- a bf16 GEMM operand anyway


// ------------------------------------------------------------------------------------------------
// F1: qkv = [(x + pos) Wqk^T + bqk | x Wv^T + bv]  ->  bf16 [n, 384]
// ------------------------------------------------------------------------------------------------
// saved x-hat rows: fp32 (C ABI) or bf16 (kLaySavedBf16); `half_cols`: the pair form's 64-column halves
__device__ __forceinline__ void store_xhat(float* dst, int n, int tok, const f32x4 (&u)[8], int lane, bool blk, bool as_bf16) {
    if (as_bf16) store_rows_bf16<128>(reinterpret_cast<bf16_t*>(dst), n, tok, 128, 0, u, lane, blk);
    else store_rows_f32<128>(dst, n, tok, u, lane, blk);
}
__device__ __forceinline__ void store_xhat_half(float* dst, int n, int tok, int h, const f32x4 (&u)[4], int lane, bool blk,
                                                bool as_bf16) {
    if (as_bf16) store_rows_bf16<64>(reinterpret_cast<bf16_t*>(dst), n, tok, 128, 64 * h, u, lane, blk);
    else store_rows_f32_cols<64>(dst, n, tok, 128, 64 * h, u, lane, blk);
}
__device__ __forceinline__ void load_xhat(const float* src, int n, int tok, f32x4 (&u)[8], int lane, bool blk, bool as_bf16) {
    if (as_bf16) {
        uint2 p[8];
        load_rows_bf16<128>(reinterpret_cast<const bf16_t*>(src), n, tok, 128, 0, p, lane, blk);
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) u[ct] = unpack4(p[ct]);
    } else {
        load_rows_f32<128>(src, n, tok, u, lane, blk);
    }
}

__global__ __launch_bounds__(kLayerBlk, 2) void sst_qkv_fwd_kernel(const float* __restrict__ x,
                                                                const int32_t* __restrict__ tok_pos,
                                                                const float* __restrict__ pos_table, LayerW W,
                                                                int n, bf16_t* __restrict__ qkv,
                                                                bf16_t* __restrict__ x_b, bf16_t* __restrict__ xp_b,
                                                                int lay, SstInputMap M) {
    const bool blk = lay & kLayBlocked;
    __shared__ __attribute__((aligned(16))) bf16_t smem[kWeightLds];
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * (kLayerBlk / 64) + (threadIdx.x >> 6);
    const int t = lane & 15, g = lane >> 4;
    const int tok = tile * 16 + t;
    uint2 xb[8], xpb[8];
    const int p = __builtin_amdgcn_raw_buffer_load_b32(rows_rsrc(tok_pos, n, 4), tok * 4, 0, 0);   // 0 past the end
    WStage<128, 256> s_qk;
    WStage<128, 128> s_v;
    stage_issue<128, 256>(W.wqkv, s_qk);
    {
        f32x4 xv[8];
        if (M.src == nullptr) {
            load_rows_f32<128>(x, n, tok, xv, lane, lay & kLayXBlocked);
        } else {
            // the stack's input conversion (common.h SstInputMap): gather / fill from the row-major source, and leave
            // the tile-blocked copy the other kernels read.  Rows past n_src (and the pad rows of the last tile) are
            // sent out of the descriptor's range: the hardware returns zeros.
            int srow = tok;
            if (M.rows) srow = __builtin_amdgcn_raw_buffer_load_b32(rows_rsrc(M.rows, M.n_src, 4), tok * 4, 0, 0);
            const __amdgpu_buffer_rsrc_t sr =
                __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(M.src), 0, 0x7fffffff, 0x00020000);
            const int voff = tok < M.n_src ? srow * 512 + 16 * g : 0x7fffffff;
            const bool take_fill = M.fill != nullptr && tok >= M.n_src && tok < n;
            const __amdgpu_buffer_rsrc_t fr = table_rsrc(M.fill ? M.fill : M.src);
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) {
                const f32x4 a = buf_load_f32x4(sr, voff + (voff < 0x7fffff00 ? 64 * ct : 0));
                const f32x4 f = buf_load_f32x4(fr, 64 * ct + 16 * g);
                xv[ct] = take_fill ? f : a;
            }
            store_rows_f32<128>(const_cast<float*>(x), n, tok, xv, lane, true);
        }
        const __amdgpu_buffer_rsrc_t pr = table_rsrc(pos_table);
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
            const f32x4 pv = buf_load_f32x4(pr, p * 512 + 64 * ct + 16 * g);
            xb[ct] = pack4(xv[ct]);
            xpb[ct] = pack4(xv[ct] + pv);
        }
    }
    if (x_b) {                       // the bf16 operands of this layer's weight-gradient contraction (dW_qk, dW_v)
        store_rows_packed<128>(x_b, n, tok, 128, 0, xb, lane, blk);
        store_rows_packed<128>(xp_b, n, tok, 128, 0, xpb, lane, blk);
    }
    {
        f32x4 acc[16];
        load_bias<256>(W.bqkv, acc, lane);
        gemm_staged<128, 256>(s_qk, smem, xpb, acc, lane);
        stage_issue<128, 128>(W.wqkv + 256 * 128, s_v);               // in flight under the q/k stores
        store_rows_bf16<256>(qkv, n, tok, 384, 0, acc, lane, blk);
    }
    {
        f32x4 acc[8];
        load_bias<128>(W.bqkv + 256, acc, lane);
        gemm_staged<128, 128>(s_v, smem, xb, acc, lane);
        store_rows_bf16<128>(qkv, n, tok, 384, 256, acc, lane, blk);
    }
}

// ------------------------------------------------------------------------------------------------
// F3: z = LN2(y + FFN(y)),  y = LN1(x + attn Wo^T + bo).  When training it also saves what the backward
// needs instead of recomputing three GEMMs there: the two normalised residuals (fp32), the FFN
// pre-activation (bf16) and the two 1/sigma per token.
// ------------------------------------------------------------------------------------------------
// Vertical fusion: when `N.wqkv` is set the kernel continues with F1 of the NEXT layer on the z it still holds in
// registers (q/k/v projection with the next layer's positional term) -- a 10 us kernel at encoder size whose
// duration is launch ramp + first-load latency + store drain rather than work (a weight-stationary rewrite of F1
// that cut its LDS traffic 6x did not get under that floor), so merging it removes one such floor per layer.
struct NextQkv {
    const bf16_t* wqkv;           // packed [384][128] of the next layer, or nullptr
    const float* bqkv;
    const int32_t* tok_pos;       // the next layer's window layout (the shift alternates)
    const float* pos_table;
    bf16_t* qkv;                  // [n, 384] of the next layer
    bf16_t *x_b, *xp_b;           // bf16 copies of z and z + pos (weight-gradient operands of the next layer), or null
};

// The fp32 parameter vectors of F3 (+ the next layer's in-projection bias) go through LDS, fetched once per workgroup at
// kernel start: [bo | g1 | be1 | b1 (256) | b2 | g2 | be2 | bqkv of the next layer (384)].  Read from global memory at
// their point of use, each one was a round trip the wave sat through (one wave per SIMD, nothing else to run), and --
// vmcnt retires in order -- every such wait also drained the weight-matrix prefetch issued before it: the ISA had
// eight serial (2 loads, s_waitcnt vmcnt(0)) pairs in the LayerNorm affine alone (tools/isa_mix.py).  LDS reads wait on
// lgkmcnt and leave the prefetches alone.
constexpr int kPrmBo = 0, kPrmG1 = 128, kPrmBe1 = 256, kPrmB1 = 384, kPrmB2 = 640, kPrmG2 = 768, kPrmBe2 = 896, kPrmQkv = 1024;
constexpr int kPrmFloats = 1408;
// (clang vector types: with HIP's float4 struct this pair stayed an in-memory aggregate in the pair kernel, the compiler
//  promoted it to LDS and the kernel ran 2x slower)
struct PrmRegs { f32x4 a, b; };
__device__ __forceinline__ void ffn_params_issue(const LayerW& W, const float* __restrict__ next_bqkv, PrmRegs& r) {
    const int s = threadIdx.x;                                        // float4 slot 0..255 of the first 1024 floats
    const float* src = s < 32 ? W.bo + 4 * s : s < 64 ? W.g1 + 4 * (s - 32) : s < 96 ? W.be1 + 4 * (s - 64)
                     : s < 160 ? W.b1 + 4 * (s - 96) : s < 192 ? W.b2 + 4 * (s - 160) : s < 224 ? W.g2 + 4 * (s - 192)
                     : W.be2 + 4 * (s - 224);
    r.a = *reinterpret_cast<const f32x4*>(src);
    r.b = f32x4{0.f, 0.f, 0.f, 0.f};
    if (next_bqkv && s < 96) r.b = *reinterpret_cast<const f32x4*>(next_bqkv + 4 * s);
}
__device__ __forceinline__ void ffn_params_commit(float* __restrict__ prm, bool has_next, const PrmRegs& r) {
    *reinterpret_cast<f32x4*>(prm + 4 * threadIdx.x) = r.a;
    if (has_next && threadIdx.x < 96) *reinterpret_cast<f32x4*>(prm + kPrmQkv + 4 * threadIdx.x) = r.b;
}

__global__ __launch_bounds__(kLayerBlk, 2) void sst_ffn_fwd_kernel(const float* __restrict__ x,
                                                                const bf16_t* __restrict__ attn, LayerW W, int n,
                                                                float eps, float* __restrict__ z,
                                                                float* __restrict__ xh1_out,
                                                                float* __restrict__ xh2_out,
                                                                bf16_t* __restrict__ hp_out,
                                                                float* __restrict__ rstd_out, NextQkv N, int lay,
                                                                int wg0 /* first workgroup (64 tokens each) that runs */) {
    const bool blk = lay & kLayBlocked;
#ifdef FFN_ABL_NO_SAVE                      // (timing ablation: what the saved-activation stores cost this kernel)
    xh1_out = nullptr; xh2_out = nullptr; hp_out = nullptr; rstd_out = nullptr; N.x_b = nullptr;
#endif
    __shared__ __attribute__((aligned(16))) bf16_t smem[kWeightLds];
#ifdef FFN_ABL_ONE_PER_CU
    __shared__ float abl_pad[5120];
    if (n < 0) { abl_pad[threadIdx.x] = 1.f; __syncthreads(); z[0] = abl_pad[threadIdx.x ^ 1]; }
#endif
    __shared__ __attribute__((aligned(16))) float prm[kPrmFloats];
    PrmRegs prm_r;
    ffn_params_issue(W, N.wqkv ? N.bqkv : nullptr, prm_r);
    const int lane = threadIdx.x & 63;
    const int tile = (blockIdx.x + wg0) * (kLayerBlk / 64) + (threadIdx.x >> 6);
    const int tok = tile * 16 + (lane & 15);
    GEOMAE_FSTAMP(0);
    f32x4 u[8], y[8];
    float r1, r2;
    WStage<128, 256> s_w1;
    {
        WStage<128, 128> s_wo;
        stage_issue<128, 128>(W.wo, s_wo);
        uint2 ob[8];
        load_rows_bf16<128>(attn, n, tok, 128, 0, ob, lane, blk);
        f32x4 xr[8];
        load_rows_f32<128>(x, n, tok, xr, lane, lay & kLayXBlocked);                 // needed after the GEMM: in flight under it
        ffn_params_commit(prm, N.wqkv != nullptr, prm_r);
        gemm_staged_then<128, 128>(s_wo, smem, ob, u, lane, [&] { stage_issue<128, 256>(W.w1, s_w1); }, -100, prm + kPrmBo);
        GEOMAE_FSTAMP(1);                                             // (W1: requested behind Wo's LDS commit)
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) u[ct] += xr[ct];
    }
    layer_norm_t(u, eps, &r1);
    if (xh1_out) store_xhat(xh1_out, n, tok, u, lane, blk, lay & kLaySavedBf16);
    affine_t(u, prm + kPrmG1, prm + kPrmBe1, y, lane);
    uint2 hb[16];
    WStage<256, 128> s_w2;
    {
        uint2 yb[8];
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) yb[ct] = pack4(y[ct]);
        GEOMAE_FSTAMP(2);
        // W1 in two halves of 128 hidden channels: 32 accumulator registers at a time instead of 64, which makes room to
        // request W2 right behind W1's LDS commit (behind the whole GEMM it arrived ~8 k cycles late at decoder size, inside
        // the one-piece GEMM it cost 89 spills); same accumulation order per output tile: bit-identical results
        gemm_commit<128, 256>(s_w1, smem);
        stage_issue<256, 128>(W.w2, s_w2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 hp[8];
            load_bias<128>(prm + kPrmB1 + 128 * half, hp, lane);
            if (half == 0) gemm_run<128, 0, 8>(smem, yb, hp, lane);
            else gemm_run<128, 8, 8>(smem, yb, hp, lane);
            if (half == 1) GEOMAE_FSTAMP(3);
            if (hp_out) store_rows_bf16<128>(hp_out, n, tok, 256, 128 * half, hp, lane, blk);
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) hb[8 * half + ct] = pack4(gelu4(hp[ct]));
        }
    }
    load_bias<128>(prm + kPrmB2, u, lane);
    GEOMAE_FSTAMP(4);
    const bool has_next = N.wqkv != nullptr;
    WStage<128, 256> s_qk;
    gemm_staged<256, 128>(s_w2, smem, hb, u, lane);
    GEOMAE_FSTAMP(5);
    if (has_next) stage_issue<128, 256>(N.wqkv, s_qk);                // lands under the LayerNorm arithmetic
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) u[ct] += y[ct];
    layer_norm_t(u, eps, &r2);
    if (xh2_out) store_xhat(xh2_out, n, tok, u, lane, blk, lay & kLaySavedBf16);
    if (rstd_out && (lane >> 4) == 0)
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(r1), __float_as_uint(r2)}, rows_rsrc(rstd_out, n, 8),
                                              tok * 8, 0, 0);
    affine_t(u, prm + kPrmG2, prm + kPrmBe2, y, lane);
    store_rows_f32<128>(z, n, tok, y, lane, lay & kLayZBlocked);
    GEOMAE_FSTAMP(6);
    if (!has_next) return;
    // ---- F1 of the next layer on z = y (registers)
    const int g = lane >> 4;
    const int p = __builtin_amdgcn_raw_buffer_load_b32(rows_rsrc(N.tok_pos, n, 4), tok * 4, 0, 0);
    uint2 xb[8], xpb[8];
    {
        const __amdgpu_buffer_rsrc_t pr = table_rsrc(N.pos_table);
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
            const f32x4 pv = buf_load_f32x4(pr, p * 512 + 64 * ct + 16 * g);
            xb[ct] = pack4(y[ct]);
            xpb[ct] = pack4(y[ct] + pv);
        }
    }
    if (N.xp_b) {                                  // (x_b alone may be null: geomae::set_skip_x_copy)
        if (N.x_b) store_rows_packed<128>(N.x_b, n, tok, 128, 0, xb, lane, blk);
        store_rows_packed<128>(N.xp_b, n, tok, 128, 0, xpb, lane, blk);
    }
    WStage<128, 128> s_v;
    {
        f32x4 acc[16];
        load_bias<256>(prm + kPrmQkv, acc, lane);
        GEOMAE_FSTAMP(7);
        gemm_staged_then<128, 256>(s_qk, smem, xpb, acc, lane, [&] { stage_issue<128, 128>(N.wqkv + 256 * 128, s_v); });
        GEOMAE_FSTAMP(8);
        store_rows_bf16<256>(N.qkv, n, tok, 384, 0, acc, lane, blk);
    }
    {
        f32x4 acc[8];
        load_bias<128>(prm + kPrmQkv + 256, acc, lane);
        gemm_staged<128, 128>(s_v, smem, xb, acc, lane);
        GEOMAE_FSTAMP(9);
        store_rows_bf16<128>(N.qkv, n, tok, 384, 256, acc, lane, blk);
        GEOMAE_FSTAMP(10);
    }
}

// Pair form of F3 (+ the fused F1) for SMALL token counts.  At encoder size (6.5 k kept tokens) the kernel above is 102
// workgroups of one wave per SIMD on a 256-CU chip: 60 % of the CUs idle while each wave walks a ~50 k-cycle dependent
// chain (256 MFMAs + the elementwise phases of 16 tokens x 128..384 channels).  Here a workgroup takes 32 tokens and two
// waves share each 16-token tile: wave half h computes half of every GEMM's output channels from the full-K operand and
// does the elementwise work (GELU, loads, stores) of its half; the halves meet through LDS three times (the two
// pre-LayerNorm sums in fp32, the GELU output in bf16), after which both hold the full row for the LayerNorm
// statistics and the next GEMM's operand.  Twice the workgroups, half the chain per wave; every value is computed by
// the same instruction sequence as in the single-wave form, so the outputs are bit-identical to it.
__global__ __launch_bounds__(kLayerBlk, 1) void sst_ffn_fwd_pair_kernel(const float* __restrict__ x,
                                                                     const bf16_t* __restrict__ attn, LayerW W, int n,
                                                                     float eps, float* __restrict__ z,
                                                                     float* __restrict__ xh1_out,
                                                                     float* __restrict__ xh2_out,
                                                                     bf16_t* __restrict__ hp_out,
                                                                     float* __restrict__ rstd_out, NextQkv N, int lay) {
    const bool blk = lay & kLayBlocked;
    __shared__ __attribute__((aligned(16))) bf16_t smem[kWeightLds];
    __shared__ uint4 xch[4 * 4 * 64];
    __shared__ __attribute__((aligned(16))) float prm[kPrmFloats];
    PrmRegs prm_r;
    ffn_params_issue(W, N.wqkv ? N.bqkv : nullptr, prm_r);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = wave >> 1;                                          // which half of the output channels
    const int tile = blockIdx.x * 2 + (wave & 1);
    const int tok = tile * 16 + (lane & 15);
    GEOMAE_FSTAMP(0);
    f32x4 u[8], y[8];
    float r1, r2;
    WStage<128, 256> s_w1;
    {
        WStage<128, 128> s_wo;
        stage_issue<128, 128>(W.wo, s_wo);
        uint2 ob[8];
        load_rows_bf16<128>(attn, n, tok, 128, 0, ob, lane, blk);
        f32x4 xr[4], uh[4], other[4];
        load_rows_f32_cols<64>(x, n, tok, 128, 64 * h, xr, lane, lay & kLayXBlocked);
        ffn_params_commit(prm, N.wqkv != nullptr, prm_r);
        gemm_staged_half<128, 128>(s_wo, smem, ob, uh, lane, h, prm + kPrmBo + 64 * h);
        GEOMAE_FSTAMP(1);
        stage_issue<128, 256>(W.w1, s_w1);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) uh[ct] += xr[ct];
        pair_exchange<4>(xch, wave, lane, reinterpret_cast<const uint4(&)[4]>(uh), reinterpret_cast<uint4(&)[4]>(other));
        join_halves<4>(uh, other, h, u);
    }
    layer_norm_t(u, eps, &r1);
    if (xh1_out) {
        f32x4 mine[4];
        half_of<4>(u, h, mine);
        store_xhat_half(xh1_out, n, tok, h, mine, lane, blk, lay & kLaySavedBf16);
    }
    affine_t(u, prm + kPrmG1, prm + kPrmBe1, y, lane);
    uint2 hb[16];
    WStage<256, 128> s_w2;
    {
        uint2 yb[8];
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) yb[ct] = pack4(y[ct]);
        f32x4 hp[8];
        load_bias<128>(prm + kPrmB1 + 128 * h, hp, lane);
        GEOMAE_FSTAMP(2);
        gemm_staged_half<128, 256>(s_w1, smem, yb, hp, lane, h);
        GEOMAE_FSTAMP(3);
        stage_issue<256, 128>(W.w2, s_w2);
        if (hp_out) store_rows_bf16<128>(hp_out, n, tok, 256, 128 * h, hp, lane, blk);
        uint4 mine[4], other[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint2 a, b;
            {
                const f32x4 v = hp[2 * i];
                const f32x4 hv = gelu4(v);
                a = pack4(hv);
            }
            {
                const f32x4 v = hp[2 * i + 1];
                const f32x4 hv = gelu4(v);
                b = pack4(hv);
            }
            mine[i] = make_uint4(a.x, a.y, b.x, b.y);
        }
        pair_exchange<4>(xch, wave, lane, mine, other);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 lo = h ? other[i] : mine[i], hi = h ? mine[i] : other[i];
            hb[2 * i] = make_uint2(lo.x, lo.y);
            hb[2 * i + 1] = make_uint2(lo.z, lo.w);
            hb[8 + 2 * i] = make_uint2(hi.x, hi.y);
            hb[8 + 2 * i + 1] = make_uint2(hi.z, hi.w);
        }
    }
    const bool has_next = N.wqkv != nullptr;
    WStage<128, 256> s_qk;
    {
        f32x4 uh[4], yh[4], other[4];
        load_bias<64>(prm + kPrmB2 + 64 * h, uh, lane);
        GEOMAE_FSTAMP(4);
        gemm_staged_half<256, 128>(s_w2, smem, hb, uh, lane, h);
        GEOMAE_FSTAMP(5);
        if (has_next) stage_issue<128, 256>(N.wqkv, s_qk);
        half_of<4>(y, h, yh);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) uh[ct] += yh[ct];
        pair_exchange<4>(xch, wave, lane, reinterpret_cast<const uint4(&)[4]>(uh), reinterpret_cast<uint4(&)[4]>(other));
        join_halves<4>(uh, other, h, u);
    }
    layer_norm_t(u, eps, &r2);
    if (xh2_out) {
        f32x4 mine[4];
        half_of<4>(u, h, mine);
        store_xhat_half(xh2_out, n, tok, h, mine, lane, blk, lay & kLaySavedBf16);
    }
    if (rstd_out && h == 0 && (lane >> 4) == 0)
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(r1), __float_as_uint(r2)}, rows_rsrc(rstd_out, n, 8),
                                              tok * 8, 0, 0);
    affine_t(u, prm + kPrmG2, prm + kPrmBe2, y, lane);
    {
        f32x4 mine[4];
        half_of<4>(y, h, mine);
        store_rows_f32_cols<64>(z, n, tok, 128, 64 * h, mine, lane, lay & kLayZBlocked);
    }
    GEOMAE_FSTAMP(6);
    if (!has_next) return;
    // ---- F1 of the next layer on z = y (registers, both waves hold the full row)
    const int g = lane >> 4;
    const int p = __builtin_amdgcn_raw_buffer_load_b32(rows_rsrc(N.tok_pos, n, 4), tok * 4, 0, 0);
    uint2 xb[8], xpb[8];
    {
        const __amdgpu_buffer_rsrc_t pr = table_rsrc(N.pos_table);
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
            const f32x4 pv = buf_load_f32x4(pr, p * 512 + 64 * ct + 16 * g);
            xb[ct] = pack4(y[ct]);
            xpb[ct] = pack4(y[ct] + pv);
        }
    }
    if (N.xp_b) {
        uint2 m0[4], m1[4];
        half_of<4>(xb, h, m0);
        half_of<4>(xpb, h, m1);
        if (N.x_b) store_rows_packed<64>(N.x_b, n, tok, 128, 64 * h, m0, lane, blk);
        store_rows_packed<64>(N.xp_b, n, tok, 128, 64 * h, m1, lane, blk);
    }
    WStage<128, 128> s_v;
    {
        f32x4 acc[8];
        load_bias<128>(prm + kPrmQkv + 128 * h, acc, lane);
        GEOMAE_FSTAMP(7);
        gemm_staged_half<128, 256>(s_qk, smem, xpb, acc, lane, h);
        GEOMAE_FSTAMP(8);
        stage_issue<128, 128>(N.wqkv + 256 * 128, s_v);
        store_rows_bf16<128>(N.qkv, n, tok, 384, 128 * h, acc, lane, blk);
    }
    {
        f32x4 acc[4];
        load_bias<64>(prm + kPrmQkv + 256 + 64 * h, acc, lane);
        gemm_staged_half<128, 128>(s_v, smem, xb, acc, lane, h);
        GEOMAE_FSTAMP(9);
        store_rows_bf16<64>(N.qkv, n, tok, 384, 256 + 64 * h, acc, lane, blk);
        GEOMAE_FSTAMP(10);
    }
}

// ------------------------------------------------------------------------------------------------
// B3: backward of F3 from the saved (xhat1, xhat2, hp, rstd).  Emits
//   dx_res [n,128] f32 (gradient reaching x through the residual = d(x + a)),  dattn [n,128] bf16,
//   bf16 row-major operands of the weight-gradient GEMMs: du, dv, dhp [n,256], y, h [n,256]
//   and the LayerNorm parameter gradients (atomics, one flush per workgroup).
// ------------------------------------------------------------------------------------------------
struct FfnBwdArgs {
    const float *xh1_in, *xh2_in;
    const bf16_t* hp_in;
    const float *rstd_in, *dz;
    LayerW W;
    int n;
    float* dx_res;
    bf16_t *dattn, *du_b, *dv_b, *dhp_b, *y_b, *h_b;
    float *dg1, *dbe1, *dg2, *dbe2;
    // optional head: B1 of the layer ABOVE (l+1) -- its dx is this layer's dz and stays in registers
    const bf16_t* up_dqkv;            // [n,384] or nullptr (then dz is read from memory)
    const float* up_dx_res;           // [n,128]
    const bf16_t *up_wqkT, *up_wvT;   // packed transposed in-projection of layer l+1
    int lay;                          // kLayBlocked: everything except dz (always row-major: it comes from outside)
    const float* dz_add;              // optional second summand of dz (the other decoder's input gradient), row-major
    int wg0;                          // workgroups below it own DEAD rows (zero dz, set_first_live_row): they store zeros
};

// B1 arithmetic: acc = dx_res + dqkv[:, :256] Wqk + dqkv[:, 256:] Wv
__device__ __forceinline__ void qkv_bwd_rows(const bf16_t* __restrict__ dqkv, const float* __restrict__ dx_res,
                                             const bf16_t* __restrict__ wqkT, const bf16_t* __restrict__ wvT, int n,
                                             int tok, bf16_t* __restrict__ smem, f32x4 (&acc)[8], int lane, bool blk) {
    WStage<256, 128> s_qk;
    WStage<128, 128> s_v;
    stage_issue<256, 128>(wqkT, s_qk);
    load_rows_f32<128>(dx_res, n, tok, acc, lane, blk);
    uint2 dv_rows[8];
    {
        uint2 d[16];
        load_rows_bf16<256>(dqkv, n, tok, 384, 0, d, lane, blk);
        load_rows_bf16<128>(dqkv, n, tok, 384, 256, dv_rows, lane, blk);  // operand of the second GEMM: in flight under the first
        stage_issue<128, 128>(wvT, s_v);
        gemm_staged<256, 128>(s_qk, smem, d, acc, lane);
    }
    gemm_staged<128, 128>(s_v, smem, dv_rows, acc, lane);
}

__device__ __forceinline__ void ffn_bwd_body(const FfnBwdArgs& A, int block, bf16_t* __restrict__ smem,
                                             float (*red)[4][128] /* [wave][tensor][channel] */) {
    const float* __restrict__ xh1_in = A.xh1_in; const float* __restrict__ xh2_in = A.xh2_in;
    const bf16_t* __restrict__ hp_in = A.hp_in; const float* __restrict__ rstd_in = A.rstd_in;
    const float* __restrict__ dz = A.dz; const LayerW& W = A.W; const int n = A.n;
    float* __restrict__ dx_res = A.dx_res; bf16_t* __restrict__ dattn = A.dattn; bf16_t* __restrict__ du_b = A.du_b;
    bf16_t* __restrict__ dv_b = A.dv_b; bf16_t* __restrict__ dhp_b = A.dhp_b; bf16_t* __restrict__ y_b = A.y_b;
    bf16_t* __restrict__ h_b = A.h_b;
    float* __restrict__ dg1 = A.dg1; float* __restrict__ dbe1 = A.dbe1; float* __restrict__ dg2 = A.dg2;
    float* __restrict__ dbe2 = A.dbe2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4;
    const int tile = block * (kLayerBlk / 64) + wave;
    const int tok = tile * 16 + (lane & 15);
    const bool blk = A.lay & kLayBlocked, valid = tok < n;
    float* red_scratch = reinterpret_cast<float*>(smem) + wave * kRedWaveFloats;   // see ln_param_grads_t
    if (block < A.wg0) {              // dead rows: every output of this kernel is zero for them (workgroup-uniform)
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        f32x4 z8[8], z16[16];
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) z8[ct] = z4;
#pragma unroll
        for (int ct = 0; ct < 16; ++ct) z16[ct] = z4;
        store_rows_f32<128>(dx_res, n, tok, z8, lane, blk);
        store_rows_bf16<128>(dattn, n, tok, 128, 0, z8, lane, blk);
        store_rows_bf16<128>(du_b, n, tok, 128, 0, z8, lane, blk);
        store_rows_bf16<128>(dv_b, n, tok, 128, 0, z8, lane, blk);
        if (y_b) store_rows_bf16<128>(y_b, n, tok, 128, 0, z8, lane, blk);
        store_rows_bf16<256>(dhp_b, n, tok, 256, 0, z16, lane, blk);
        store_rows_bf16<256>(h_b, n, tok, 256, 0, z16, lane, blk);
        return;
    }
    GEOMAE_STAMP(0);
    const uint2 rs = buf_load_b64(rows_rsrc(rstd_in, n, 8), tok * 8);
    const float r1 = __uint_as_float(rs.x), r2 = __uint_as_float(rs.y);
    f32x4 dv[8];
    if (A.up_dqkv) qkv_bwd_rows(A.up_dqkv, A.up_dx_res, A.up_wqkT, A.up_wvT, n, tok, smem, dv, lane, blk);
    else {
        load_rows_f32<128>(dz, n, tok, dv, lane);
        if (A.dz_add) {
            f32x4 d2[8];
            load_rows_f32<128>(A.dz_add, n, tok, d2, lane);
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) dv[ct] += d2[ct];
        }
    }
    GEOMAE_STAMP(20);
    WStage<128, 256> s_w2T;
    // ---- LN2 backward
    {
        f32x4 xh2[8];
        load_xhat(xh2_in, n, tok, xh2, lane, blk, A.lay & kLaySavedBf16);
        stage_issue<128, 256>(W.w2T, s_w2T);                          // lands under the LayerNorm arithmetic
        if (A.up_dqkv) __syncthreads();                               // B1's last matrix consumed by every wave
        ln_param_grads_t(dv, xh2, red_scratch, red[wave], 0, lane, valid);   // d gamma2, d beta2
        layer_norm_bwd_t(dv, xh2, W.g2, r2, lane);                    // dv = d(y + f)
    }
    store_rows_bf16<128>(dv_b, n, tok, 128, 0, dv, lane, blk);
    GEOMAE_STAMP(1);
    // ---- FFN backward: dh = dv W2 ; dhp = dh * gelu'(hp) ; dy = dv + dhp W1
    uint2 dhpb[16];
    WStage<256, 128> s_w1T;
    {
        uint2 hpb[16];
        load_rows_bf16<256>(hp_in, n, tok, 256, 0, hpb, lane, blk);           // needed after the GEMM: in flight under it
        uint2 dvb[8];
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) dvb[ct] = pack4(dv[ct]);
        f32x4 dh[16];
#pragma unroll
        for (int ct = 0; ct < 16; ++ct) dh[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        gemm_staged<128, 256>(s_w2T, smem, dvb, dh, lane, 2);
        GEOMAE_STAMP(5);
        stage_issue<256, 128>(W.w1T, s_w1T);                          // lands under the GELU arithmetic
        const RowAddr h_a = row_addr<2>(h_b, n, tok, 256, 0, lane, blk), dhp_a = row_addr<2>(dhp_b, n, tok, 256, 0, lane, blk);
#pragma unroll
        for (int ct = 0; ct < 16; ++ct) {
            const f32x4 hp = unpack4(hpb[ct]);
            f32x4 h, gr;
            gelu_fwd_bwd4(hp, &h, &gr);
            dh[ct] *= gr;
            dhpb[ct] = pack4(dh[ct]);
            const uint2 hb2 = pack4(h);
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{hb2.x, hb2.y}, h_a.r, h_a.voff, ct * h_a.ct_stride, 0);
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{dhpb[ct].x, dhpb[ct].y}, dhp_a.r, dhp_a.voff, ct * dhp_a.ct_stride, 0);
        }
    }
    GEOMAE_STAMP(6);
    f32x4 xh1[8];
    load_xhat(xh1_in, n, tok, xh1, lane, blk, A.lay & kLaySavedBf16);  // in flight under the GEMM
    gemm_staged<256, 128>(s_w1T, smem, dhpb, dv, lane, 7);            // dv now holds dy
    GEOMAE_STAMP(10);
    WStage<128, 128> s_woT;
    stage_issue<128, 128>(W.woT, s_woT);                              // lands under the LayerNorm arithmetic
    // ---- LN1 backward
    {
        if (y_b) {                                                    // (null: the contraction forms y from the saved xhat1)
            f32x4 y[8];
            affine_t(xh1, W.g1, W.be1, y, lane);
            store_rows_bf16<128>(y_b, n, tok, 128, 0, y, lane, blk);
        }
        __syncthreads();                                              // w1T consumed by every wave
        ln_param_grads_t(dv, xh1, red_scratch, red[wave], 2, lane, valid);   // d gamma1, d beta1
        layer_norm_bwd_t(dv, xh1, W.g1, r1, lane);                    // dv now holds du = d(x + a)
    }
    store_rows_f32<128>(dx_res, n, tok, dv, lane, blk);
    store_rows_bf16<128>(du_b, n, tok, 128, 0, dv, lane, blk);
    GEOMAE_STAMP(11);
    {
        uint2 dub[8];
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) dub[ct] = pack4(dv[ct]);
        f32x4 da[8];
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) da[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        gemm_staged<128, 128>(s_woT, smem, dub, da, lane, 12);
        GEOMAE_STAMP(15);
        store_rows_bf16<128>(dattn, n, tok, 128, 0, da, lane, blk);
    }
    GEOMAE_STAMP(16);
    // ---- flush LayerNorm parameter gradients (invalid rows contributed zeros: dz was loaded as 0)
    __syncthreads();
    for (int e = threadIdx.x; e < 4 * 128; e += kLayerBlk) {
        const int k = e >> 7, c = e & 127;
        const float s = red[0][k][c] + red[1][k][c] + red[2][k][c] + red[3][k][c];
        float* dst = k == 0 ? dg2 : (k == 1 ? dbe2 : (k == 2 ? dg1 : dbe1));
        atomicAdd(dst + c, s);
    }
    GEOMAE_STAMP(17);
}

__global__ __launch_bounds__(kLayerBlk, 2) void sst_ffn_bwd_kernel(FfnBwdArgs A) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[kWeightLds];
    __shared__ float red[4][4][128];
#ifdef FFN_ABL_ONE_PER_CU                  // (timing ablation: 16 KB more LDS = ONE workgroup per CU instead of two -- what occupancy is worth)
    __shared__ float abl_pad[4096];
    if (A.n < 0) abl_pad[threadIdx.x] = 1.f;
#endif
    ffn_bwd_body(A, blockIdx.x, smem, red);
#ifdef FFN_ABL_ONE_PER_CU
    if (A.n < 0) A.dx_res[0] = abl_pad[threadIdx.x ^ 1];
#endif
}

// (Round 5: a PAIR form of this kernel -- 32 tokens per workgroup, two waves per 16-token tile each computing half of every
// GEMM's output channels, three LDS exchanges, like sst_ffn_fwd_pair_kernel -- was built again now that the launches carry no
// contraction, checked against this one (operands identical, dx within 1 ulp of fp32) and measured: 24.0 us per encoder-size
// launch against 19.8-21 us for this kernel, enc_bwd 0.49 -> 0.55 ms with the contraction launches beside it.  Both waves of a
// pair run the two LayerNorm backwards on full rows and one of them the parameter-gradient reduction while the other waits:
// the elementwise phases, not the GEMMs, are this kernel's chain.  Removed; docs/LAB_NOTES.md round 5.)

// ------------------------------------------------------------------------------------------------
// B1: dx = dx_res + dqk Wqk + dv Wv.  Stand-alone only for the first layer of a stack; for every other layer it is
// the head of the ffn-backward kernel of the layer below (FfnBwdArgs.up_*).  The bf16 copies of x and x + pos that
// the weight-gradient contraction needs are written by the forward (F1), not here.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kLayerBlk, 2) void sst_qkv_bwd_kernel(const bf16_t* __restrict__ dqkv,
                                                                const float* __restrict__ dx_res, LayerW W, int n,
                                                                float* __restrict__ dx, int lay,
                                                                const int32_t* __restrict__ out_rows, int n_out,
                                                                float* __restrict__ tail_sum, int tail_from) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[kWeightLds];
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * (kLayerBlk / 64) + (threadIdx.x >> 6);
    const int tok = tile * 16 + (lane & 15);
    f32x4 acc[8];
    qkv_bwd_rows(dqkv, dx_res, W.wqkT, W.wvT, n, tok, smem, acc, lane, lay & kLayBlocked);
    if (!out_rows) {
        store_rows_f32<128>(dx, n, tok, acc, lane);                   // the stack's output: row-major
    } else {                                                          // ... scattered: token t -> row out_rows[t] of [n_out, 128]
        const int orow = tok < n ? out_rows[tok] : n_out;             // (past the end: dropped by the range check)
        store_rows_f32<128>(dx, n_out, orow, acc, lane);
    }
    // column sums of the rows >= tail_from (the mask tokens' rows: their gradient is one shared row, bb.py:239-246)
    if (tail_sum && (int)(blockIdx.x + 1) * kLayerBlk / 4 > tail_from) {            // workgroup-uniform
        __syncthreads();                                              // every wave is done with the weights in smem
        float* red = reinterpret_cast<float*>(smem);                  // [waves][128]
        const bool in_tail = tok >= tail_from && tok < n;
        const int g = lane >> 4, wave = threadIdx.x >> 6;
#pragma unroll
        for (int ct = 0; ct < 8; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s = row16_sum(in_tail ? acc[ct][r] : 0.0f);
                if ((lane & 15) == 0) red[wave * 128 + 16 * ct + 4 * g + r] = s;
            }
        __syncthreads();
        if (threadIdx.x < 128) {
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < kLayerBlk / 64; ++w) s += red[w * 128 + threadIdx.x];
            atomicAdd(tail_sum + threadIdx.x, s);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// DW: C[c_row0 + i][c_col0 + j] += sum_t A[t][a_col0 + i] * B[t][b_col0 + j],  i, j < 128
//     dbias[c_row0 + i]         += sum_t A[t][a_col0 + i]                        (if dbias)
// ------------------------------------------------------------------------------------------------
constexpr int kDwTok = 64;             // tokens per LDS slab: two MFMA K-steps of 32 between one pair of barriers
constexpr int kDwPieces = kDwTok * 16 / 256;   // 16-byte pieces per thread, operand and slab
constexpr int kDwLd = 128 + 8;         // LDS row: 128 channels + 16 B pad (token-major, no transposition)

// element offset of (token, column) in a [n, ld] bf16 operand: row-major or tile-blocked [n/16][ld/16][16][16]
__device__ __forceinline__ int64_t dw_elem(int tok, int ld, int col, bool blk) {
    return blk ? ((int64_t)(tok >> 4) * (ld >> 4) + (col >> 4)) * 256 + (tok & 15) * 16 + (col & 15)
               : (int64_t)tok * ld + col;
}

// Split-K.  Every workgroup of a task ends with a [128,128] fp32 partial sum.  Added to the gradient with float atomics
// (16 k per workgroup, 1.7 M per encoder-size launch, 3 M at decoder size, all through the memory-side atomic units)
// the additions drain for ~16 us AFTER the contraction loop, past the end of the ffn-backward workgroups the
// contraction rides with, and set the launch's duration.  With a workspace (`partial`, geomae_sst_stack_backward's
// scratch) the workgroups STORE their partials -- accumulator order, 4 KB per instruction -- and a LATER launch of the
// same stream sums them in chunk order and adds them to the gradient (dw_reduce_body, 64 workgroups riding in the next
// ffn-backward launch): no atomics, and a result that no longer depends on arrival order.
__device__ __forceinline__ void dw_body(const DwTask& T, int n, int chunk, int bx, bf16_t* __restrict__ As,
                                        bf16_t* __restrict__ Bs, float (*bred)[128], bool blk,
                                        float* __restrict__ partial /* of this task: [gx][128*128], or null */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int o = lane & 15, g = lane >> 4;
    const int t_begin = bx * chunk;
    const int t_end = t_begin + chunk < n ? t_begin + chunk : n;
    if (t_begin >= n) return;
    GEOMAE_STAMP(24);
    f32x4 acc[2][8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // staging: 1024 16-byte pieces per operand slab, four per thread (tokens tk0 + 16 k, channel chunk cch).
    // Row-major operands: token = q >> 4, chunk = q & 15 (16 threads read one 256-byte row piece).  Tile-blocked
    // operands: [16-channel block][half][token]: 32 threads read one contiguous 512-byte block (with the row-major
    // mapping they gathered 8 separate 32-byte pieces per token and the contraction ran 29 % slower), and the 16
    // token lanes of a half write 16 LDS rows 4 banks apart (conflict-free; [block][token][half] was 2-way conflicted).
    const int cch = blk ? ((threadIdx.x >> 5) & 7) * 2 + ((threadIdx.x >> 4) & 1) : (threadIdx.x & 15);
    const int tk0 = blk ? threadIdx.x & 15 : threadIdx.x >> 4;
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.f;
    // operand B formed on load (DwTask.b_scale): this thread's eight channels' constants
    float bsc[8], bsh[8];
    const bool b_affine = T.b_scale != nullptr;                       // workgroup-uniform
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        bsc[e] = b_affine ? T.b_scale[T.b_col0 + 8 * cch + e] : 1.f;
        bsh[e] = b_affine ? T.b_shift[T.b_col0 + 8 * cch + e] : 0.f;
    }
    // Operand slabs travel global -> registers -> LDS.  A register ring of kDwStages slabs keeps kDwStages - 1 slabs
    // of loads in flight: with a single stage the loads issued under one slab's MFMAs (~0.3 us) were needed one
    // slab later and each slab paid most of the memory latency.  Slabs of 64 tokens: a slab's life is a serial chain
    // (barrier, LDS writes, barrier, transposing LDS reads, MFMAs) that cost 2.1 k cycles per 32 tokens for 512 cycles of
    // MFMA (tools/phase_dw.py); two K-steps per chain halve the number of chains.
    constexpr int kDwStages = 3;
    u32x4 ra[kDwStages][kDwPieces], rb[kDwStages][kDwPieces];
    auto fetch = [&](int t0, u32x4 (&qa)[kDwPieces], u32x4 (&qb)[kDwPieces]) {
#pragma unroll
        for (int k = 0; k < kDwPieces; ++k) {
            const int tok = t0 + tk0 + 16 * k;
            qa[k] = u32x4{0u, 0u, 0u, 0u};
            qb[k] = qa[k];
            if (tok < t_end) {
                qa[k] = *reinterpret_cast<const u32x4*>(T.A + dw_elem(tok, T.lda, T.a_col0 + 8 * cch, blk));
                qb[k] = *reinterpret_cast<const u32x4*>(T.B + dw_elem(tok, T.ldb, T.b_col0 + 8 * cch, blk));
            }
        }
    };
#pragma unroll
    for (int st = 0; st < kDwStages - 1; ++st) fetch(t_begin + st * kDwTok, ra[st], rb[st]);
    const int m = o;
    const bf16_t* arow = As + (8 * g + (m >> 2)) * kDwLd + 4 * (m & 3);
    const bf16_t* brow = Bs + (8 * g + (m >> 2)) * kDwLd + 4 * (m & 3);
    for (int t0 = t_begin; t0 < t_end; t0 += kDwStages * kDwTok) {
#pragma unroll
        for (int st = 0; st < kDwStages; ++st) {
            const int cur = t0 + st * kDwTok;
            if (cur < t_end) {                                    // workgroup-uniform
                // the slab kDwStages - 1 ahead goes into the stage consumed in the previous iteration
                fetch(cur + (kDwStages - 1) * kDwTok, ra[(st + kDwStages - 1) % kDwStages], rb[(st + kDwStages - 1) % kDwStages]);
                __syncthreads();      // previous slab fully consumed
#pragma unroll
                for (int k = 0; k < kDwPieces; ++k) {
                    *reinterpret_cast<u32x4*>(As + (tk0 + 16 * k) * kDwLd + 8 * cch) = ra[st][k];
                    u32x4 vb = rb[st][k];
                    if (b_affine) {
                        // (rows past the chunk: A is zero there.  Rows the forward SKIPPED -- the dead rows of a decoder's last
                        //  layer -- hold whatever the buffer held; their A rows are zero, but 0 x NaN is NaN: clamp to finite)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float lo = fminf(fmaxf(bf_lo(vb[e]), -65504.f), 65504.f);
                            const float hi = fminf(fmaxf(bf_hi(vb[e]), -65504.f), 65504.f);
                            vb[e] = pack2(lo * bsc[2 * e] + bsh[2 * e], hi * bsc[2 * e + 1] + bsh[2 * e + 1]);
                        }
                    }
                    *reinterpret_cast<u32x4*>(Bs + (tk0 + 16 * k) * kDwLd + 8 * cch) = vb;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        bsum[2 * e] += bf_lo(ra[st][k][e]);
                        bsum[2 * e + 1] += bf_hi(ra[st][k][e]);
                    }
                }
                __syncthreads();
#pragma unroll
                for (int ks = 0; ks < kDwTok / 32; ++ks) {            // MFMA K-steps of 32 tokens
                    const bf16_t* ar = arow + 32 * ks * kDwLd;
                    const bf16_t* br = brow + 32 * ks * kDwLd;
                    uint4 af[2];
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const uint2 lo = tr_read(ar + 32 * wave + 16 * it), hi = tr_read(ar + 4 * kDwLd + 32 * wave + 16 * it);
                        af[it] = make_uint4(lo.x, lo.y, hi.x, hi.y);
                    }
#pragma unroll
                    for (int jt = 0; jt < 8; ++jt) {
                        const uint2 lo = tr_read(br + 16 * jt), hi = tr_read(br + 4 * kDwLd + 16 * jt);
                        const uint4 bf = make_uint4(lo.x, lo.y, hi.x, hi.y);
#pragma unroll
                        for (int it = 0; it < 2; ++it) acc[it][jt] = mfma32(af[it], bf, acc[it][jt]);
                    }
                }
            }
        }
    }
    GEOMAE_STAMP(25);
    // C layout: row i = 32*wave + 16*it + 4*g + r, col j = 16*jt + o.  Added to the gradient with float atomics
    // (memory-side units): issued straight from this layout an instruction touches four 64-byte pieces of four rows.
    // The wave first transposes in registers (ds_bpermute: target lane L pulls column 16*(L>>4) + (L&15) of one row
    // from the 16 lanes that hold that row) so that every atomic instruction covers 256 contiguous bytes of ONE row.
    if (partial) {
        // accumulator order [it][jt][thread][r]: one 16-byte store per lane, 4 KB per instruction
        f32x4* mine = reinterpret_cast<f32x4*>(partial) + (size_t)bx * 4096 + threadIdx.x;
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int jt = 0; jt < 8; ++jt) mine[(it * 8 + jt) * 256] = acc[it][jt];
    } else {
        const int sel = lane >> 4;
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int gt = 0; gt < 4; ++gt) {
                    const int src = ((lane & 15) + 16 * gt) * 4;          // byte address of the source lane
                    const int i = 32 * wave + 16 * it + 4 * gt + r;
                    float* crow = T.C + (int64_t)(T.c_row0 + i) * T.ldc + T.c_col0 + lane;
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        float v[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            v[q] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(acc[it][4 * half + q][r])));
                        const float mine = sel == 0 ? v[0] : sel == 1 ? v[1] : sel == 2 ? v[2] : v[3];
                        if (i < T.rows_valid) atomicAdd(crow + 64 * half, mine);
                    }
                }
    }
    GEOMAE_STAMP(26);
    if (T.dbias) {
        // column sums of A: thread holds 8 channels (chunk cch) of tokens tk0 + 16k (+32 per slab)
        if (!blk) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = bsum[e];
                v = rows4_sum(v);
                if (lane < 16) bred[wave][8 * cch + e] = v;
            }
            __syncthreads();
            if (threadIdx.x < 128 && threadIdx.x < T.rows_valid)
                atomicAdd(T.dbias + T.c_row0 + threadIdx.x,
                          bred[0][threadIdx.x] + bred[1][threadIdx.x] + bred[2][threadIdx.x] + bred[3][threadIdx.x]);
        } else {                 // the 16 token lanes of a chunk are one DPP row, and each chunk lives in exactly one row
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = row16_sum(bsum[e]);
                if ((lane & 15) == 0) bred[0][8 * cch + e] = v;
            }
            __syncthreads();
            if (threadIdx.x < 128 && threadIdx.x < T.rows_valid) atomicAdd(T.dbias + T.c_row0 + threadIdx.x, bred[0][threadIdx.x]);
        }
    }
}

// Sum of the partials an EARLIER launch left (dw_body), added to the gradients with plain loads and stores (this launch
// is the only writer).  Block rb of nrb; one 16-byte accumulator slot (it, jt, thread) per thread and step.
__device__ __forceinline__ void dw_reduce_body(const DwReduce& R, int rb, int nrb) {
    const int slots = R.num_tasks * 4096;
    for (int slot = rb * 256 + threadIdx.x; slot < slots; slot += nrb * 256) {
        const int ti = slot >> 12, q = slot & 4095;
        const int th = q & 255, it = q >> 11, jt = (q >> 8) & 7;
        const DwReduceTask& T = R.t[ti];
        const int wave = th >> 6, g = (th >> 4) & 3, o = th & 15;
        // the four gradient words first, then every partial: one memory round trip for the whole slot (the workgroup's
        // life is latency: issued one after the other these loads made a 64-block reduction last 10-19 us)
        float* c[4];
        float cv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 32 * wave + 16 * it + 4 * g + r, j = 16 * jt + o;
            c[r] = T.C + (int64_t)(T.c_row0 + (i < T.rows_valid ? i : 0)) * T.ldc + T.c_col0 + j;
            cv[r] = *c[r];
        }
        const f32x4* p = reinterpret_cast<const f32x4*>(R.partial) + (size_t)ti * R.gx * 4096 + q;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < R.gx; k0 += 12) {                          // <= 24 chunks: at most two rounds
            f32x4 v[12];
#pragma unroll
            for (int u = 0; u < 12; ++u) v[u] = p[(size_t)(k0 + u < R.gx ? k0 + u : k0) * 4096];
#pragma unroll
            for (int u = 0; u < 12; ++u) s += k0 + u < R.gx ? v[u] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 32 * wave + 16 * it + 4 * g + r;
            if (i < T.rows_valid) *c[r] = cv[r] + s[r];
        }
    }
}

// grid (chunks, tasks [+ rows of reduce blocks when R carries a previous launch's partials])
__global__ __launch_bounds__(256) void dw_kernel(DwTasks tasks, int n, int chunk, int num_tasks, DwReduce R) {
    __shared__ __attribute__((aligned(16))) bf16_t As[kDwTok * kDwLd];
    __shared__ __attribute__((aligned(16))) bf16_t Bs[kDwTok * kDwLd];
    __shared__ float bred[4][128];
    if ((int)blockIdx.y >= num_tasks) {                  // rows of reduce blocks behind the task rows
        dw_reduce_body(R, ((int)blockIdx.y - num_tasks) * gridDim.x + blockIdx.x, ((int)gridDim.y - num_tasks) * gridDim.x);
        return;
    }
    dw_body(tasks.t[blockIdx.y], n, chunk, blockIdx.x, As, Bs, bred, tasks.blocked != 0,
            tasks.partial ? tasks.partial + (size_t)blockIdx.y * gridDim.x * 16384 : nullptr);
}

__global__ __launch_bounds__(256) void dw_reduce_kernel(DwReduce R) { dw_reduce_body(R, blockIdx.x, gridDim.x); }

// Horizontal fusion for the backward of a layer stack: the data-gradient kernel of layer l and the weight-
// gradient contraction of layer l+1 (whose operands the previous three kernels left in the other scratch
// set) are independent, and at encoder size neither fills the chip (105 and 112 workgroups on 256 CUs;
// running the dw kernel on a second stream gained only 0.03 ms/step because of the cross-queue event hops).
// One launch carries both: workgroups [0, n_ffn) run ffn_bwd_body, the rest run dw_body, sharing the LDS
// allocation of the larger body.
__global__ __launch_bounds__(kLayerBlk, 2) void sst_ffn_bwd_dw_kernel(FfnBwdArgs A, int n_ffn, DwTasks tasks, int dw_n,
                                                                     int dw_chunk, int dw_gx, int dw_blocks, DwReduce R) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[kWeightLds];
    __shared__ float red[4][4][128];
    static_assert(2 * kDwTok * kDwLd <= kWeightLds, "dw slabs must fit in the weight buffer");
    const int wkind = (int)blockIdx.x < n_ffn ? 1 : ((int)blockIdx.x < n_ffn + dw_blocks ? 2 : 3);
    (void)wkind;
    GEOMAE_WSTAMP(28, wkind);
    if ((int)blockIdx.x < n_ffn) {
        ffn_bwd_body(A, blockIdx.x, smem, red);
    } else if ((int)blockIdx.x < n_ffn + dw_blocks) {
        const int b = blockIdx.x - n_ffn, ti = b / dw_gx;
        dw_body(tasks.t[ti], dw_n, dw_chunk, b % dw_gx, smem, smem + kDwTok * kDwLd,
                reinterpret_cast<float (*)[128]>(&red[0][0][0]), tasks.blocked != 0,
                tasks.partial ? tasks.partial + (size_t)ti * dw_gx * 16384 : nullptr);
    } else {
        dw_reduce_body(R, blockIdx.x - n_ffn - dw_blocks, kDwReduceBlocks);
    }
    GEOMAE_WSTAMP(29, wkind);
}

// enough workgroups to fill the chip whatever the number of tasks (the VFE's single dW1 task ran on 32)
static void dw_grid(int num_tasks, int num_tokens, int* gx, int* chunk_out, bool split_k = false) {
    // 512 tokens per workgroup: every workgroup ends in 16 k float atomics, and halving the chunk at encoder size (twice
    // the workgroups, twice the atomics) made the carrying ffn-backward launches 9 us slower
    int G = cdiv(num_tokens, 512);
    // (8 tasks x 24 = 192 workgroups beside the carrying launch's ffn workgroups: 12 / 16 / 20 / 24 / 32 chunks per task
    // measured 2.385 / 2.361 / 2.345 / 2.344 / 2.356 ms per step -- flat, the kernels wait, they do not queue)
    int cap = num_tasks >= 8 ? 24 : (256 / num_tasks < 128 ? 256 / num_tasks : 128);
    // one task over very many tokens (the VFE's layer-1 gradient: 106 k points, 54 MB of operands) is bandwidth-bound:
    // without atomics to pay for, more workgroups pull harder (what the split-K workspace holds: 192 partials)
    // (join wait of the VFE backward phase: 49 / 43 / 40 us at 96 / 128 / 192 workgroups)
    if (split_k && num_tasks == 1) { cap = (int)(kDwPartialBytes / 65536); G = cdiv(num_tokens, 256); }
    if (G > cap) G = cap;
    int chunk = cdiv(num_tokens, G);
    chunk = (chunk + kDwTok - 1) / kDwTok * kDwTok;
    *gx = cdiv(num_tokens, chunk);
    *chunk_out = chunk;
}

// hand-over between geomae_sst_weight_grad (deferred) and the next geomae_sst_ffn_backward on the same host
// thread; only sst_stack_backward uses it (defer_next_weight_grad), the plain C-ABI calls launch immediately
// `layer`: the same contraction as the four jobs of the layer-form kernel (dw_device.h; stacks only: blocked operands +
// a split-K workspace), `partial_base` the stack's whole workspace (both buffers)
struct PendingDw { DwTasks tasks; int num_tasks, num_tokens; bool active; bool has_layer = false; DlJob layer[4]; float* partial_base = nullptr; };
static thread_local PendingDw g_pending_dw = {{}, 0, 0, false};
static int launch_dw_layers(const PendingDw* P, int count, hipStream_t stream);
static thread_local bool t_defer_weight_grad = false;
// "defer all" mode (common.h set_defer_all_weight_grads): every deferred contraction of this host thread is QUEUED instead
// of riding in the next ffn-backward launch; geomae_flush_weight_grad launches the queue on its stream.  A stack's
// backward then consists of ffn / attention kernels only, and its weight-gradient contractions -- read by nothing but
// the optimizer -- run later on another stream beside whatever the main stream does next.
static thread_local bool t_defer_all = false;
static thread_local std::vector<PendingDw> t_dw_queue;
// partials a launched contraction left in its workspace: the next launch of this host thread (the next ffn-backward of
// the stack, or the flush) sums them
static thread_local DwReduce g_pending_reduce;
static DwReduce take_pending_reduce() {
    const DwReduce R = g_pending_reduce;
    g_pending_reduce = DwReduce();
    return R;
}
static void note_partials(const DwTasks& T, int num_tasks, int gx) {
    if (!T.partial) return;
    DwReduce R;
    R.partial = T.partial; R.gx = gx; R.num_tasks = num_tasks;
    for (int k = 0; k < num_tasks; ++k) R.t[k] = DwReduceTask{T.t[k].C, T.t[k].ldc, T.t[k].c_row0, T.t[k].c_col0, T.t[k].rows_valid};
    g_pending_reduce = R;
}


}  // namespace geomae

using namespace geomae;

#ifdef GEOMAE_PHASE_TIMING
extern "C" int geomae_debug_read_stamps(unsigned long long* host, int clear) {
    hipDeviceSynchronize();
    if (host) hipMemcpyFromSymbol(host, HIP_SYMBOL(geomae_stamps), sizeof(unsigned long long) * GEOMAE_STAMP_BLOCKS * GEOMAE_STAMP_SLOTS);
    if (clear) {
        static unsigned long long zeros[GEOMAE_STAMP_BLOCKS * GEOMAE_STAMP_SLOTS];
        hipMemcpyToSymbol(HIP_SYMBOL(geomae_stamps), zeros, sizeof(zeros));
    }
    return 0;
}
#endif

extern "C" int geomae_pack_weights(const float* flat_params, const int64_t* desc, int32_t num_desc,
                                   int64_t max_elems, void* packed_bf16, float* aux_f32, hipStream_t stream) {
    if (num_desc <= 0) return GEOMAE_OK;
    GEOMAE_REQUIRE(desc && packed_bf16 && max_elems > 0 && max_elems < (1ll << 30), "pack_weights: bad argument");
    int gx = (int)((max_elems / 8 + 255) / 256);          // a thread packs 8 elements
    if (gx > 128) gx = 128;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(gx, num_desc), dim3(256), 0, stream, flat_params, desc,
                       (bf16_t*)packed_bf16, aux_f32);
    return check_launch("pack_weights_kernel");
}

// the layer kernels address token rows with 32-bit byte offsets (raw buffer descriptors, sst_device.h): the widest
// row is 768 bytes (bf16 qkv) and a launch touches up to 63 rows past the end
constexpr int kMaxLayerTokens = 2700000;
#define GEOMAE_CHECK_TOKENS(n, who) GEOMAE_REQUIRE((n) <= kMaxLayerTokens, who ": more than 2.7 M tokens per call")

// The pair kernels pay when one workgroup per CU covers the whole launch (one wave per SIMD either way, twice the CUs
// busy); past that the single-wave form's lower total work wins.  GeomaeTuning.pair_kernels = 0 / 1 (or
// geomae_sst_set_pair_kernels) force the choice (A/B runs, the bit-identity test).
// rows the stack marked dead for the NEXT ffn forward / backward call of this host thread (sst_stack.hip)
#define t_skip_rows (geomae::first_live_row())
static bool use_pair_kernels(int tiles) {
    const int mode = tuning().pair_kernels;
    if (mode >= 0) return mode != 0;
    return cdiv(tiles, 2) <= 256;
}
extern "C" void geomae_sst_set_pair_kernels(int32_t mode) { tuning_mut().pair_kernels = mode < 0 ? -1 : (mode != 0); }

extern "C" int geomae_sst_qkv_forward(const float* x, const int32_t* tok_pos, const float* pos_table,
                                      const GeomaeSstLayerWeights* w, int32_t num_tokens, void* qkv_bf16,
                                      void* x_bf16, void* xp_bf16, hipStream_t stream) {
    if (num_tokens <= 0) return GEOMAE_OK;
    int rc = check_weights(w, "sst_qkv_forward");
    if (rc) return rc;
    GEOMAE_CHECK_TOKENS(num_tokens, "sst_qkv_forward");
    GEOMAE_REQUIRE(x && tok_pos && pos_table && qkv_bf16, "sst_qkv_forward: null argument");
    GEOMAE_REQUIRE((x_bf16 == nullptr) == (xp_bf16 == nullptr), "sst_qkv_forward: pass both operand copies or none");
    const int tiles = cdiv(num_tokens, 16);
    const SstInputMap M = input_map();
    GEOMAE_REQUIRE(!M.src || (layer_layout() & kLayXBlocked), "sst_qkv_forward: an input map needs the blocked layout");
    hipLaunchKernelGGL(sst_qkv_fwd_kernel, dim3(cdiv(tiles, kLayerBlk / 64)), dim3(kLayerBlk), 0, stream, x, tok_pos,
                       pos_table, to_layer(w), num_tokens, (bf16_t*)qkv_bf16, (bf16_t*)x_bf16, (bf16_t*)xp_bf16,
                       layer_layout(), M);
    return check_launch("sst_qkv_fwd_kernel");
}

extern "C" int geomae_sst_ffn_qkv_forward(const float* x, const void* attn_bf16, const GeomaeSstLayerWeights* w,
                                          int32_t num_tokens, float* z, float* xhat1, float* xhat2, void* hp_bf16,
                                          float* rstd, const GeomaeSstLayerWeights* next_w, const int32_t* next_tok_pos,
                                          const float* pos_table, void* next_qkv_bf16, void* next_x_bf16,
                                          void* next_xp_bf16, hipStream_t stream) {
    if (num_tokens <= 0) return GEOMAE_OK;
    int rc = check_weights(w, "sst_ffn_forward");
    if (rc) return rc;
    GEOMAE_CHECK_TOKENS(num_tokens, "sst_ffn_forward");
    GEOMAE_REQUIRE(x && attn_bf16 && z, "sst_ffn_forward: null argument");
    const bool save = xhat1 || xhat2 || hp_bf16 || rstd;
    GEOMAE_REQUIRE(!save || (xhat1 && xhat2 && hp_bf16 && rstd), "sst_ffn_forward: pass all four save buffers or none");
    NextQkv N = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (next_w) {
        if ((rc = check_weights(next_w, "sst_ffn_qkv_forward"))) return rc;
        GEOMAE_REQUIRE(next_tok_pos && pos_table && next_qkv_bf16, "sst_ffn_qkv_forward: null argument for the next layer");
        GEOMAE_REQUIRE((next_x_bf16 == nullptr) == (next_xp_bf16 == nullptr), "sst_ffn_qkv_forward: pass both operand copies or none");
        N = NextQkv{(const bf16_t*)next_w->wqkv_p, next_w->bqkv, next_tok_pos, pos_table, (bf16_t*)next_qkv_bf16,
                    skip_x_copy() ? nullptr : (bf16_t*)next_x_bf16, (bf16_t*)next_xp_bf16};
    }
    const int tiles = cdiv(num_tokens, 16);
    if (use_pair_kernels(tiles) && !(t_skip_rows >= 64 && !next_w)) {
        hipLaunchKernelGGL(sst_ffn_fwd_pair_kernel, dim3(cdiv(tiles, 2)), dim3(kLayerBlk), 0, stream, x,
                           (const bf16_t*)attn_bf16, to_layer(w), num_tokens, w->ln_eps, z, xhat1, xhat2,
                           (bf16_t*)hp_bf16, rstd, N, layer_layout());
        set_last_kernel_variant(1);
        return check_launch("sst_ffn_fwd_pair_kernel");
    }
    // rows the caller never reads (set_first_live_row, the last layer of a decoder stack): whole workgroups are skipped
    const int wg0 = next_w ? 0 : t_skip_rows / 64;
    hipLaunchKernelGGL(sst_ffn_fwd_kernel, dim3(cdiv(tiles, kLayerBlk / 64) - wg0), dim3(kLayerBlk), 0, stream, x,
                       (const bf16_t*)attn_bf16, to_layer(w), num_tokens, w->ln_eps, z, xhat1, xhat2,
                       (bf16_t*)hp_bf16, rstd, N, layer_layout(), wg0);
    return check_launch("sst_ffn_fwd_kernel");
}

extern "C" int geomae_sst_ffn_forward(const float* x, const void* attn_bf16, const GeomaeSstLayerWeights* w,
                                      int32_t num_tokens, float* z, float* xhat1, float* xhat2, void* hp_bf16,
                                      float* rstd, hipStream_t stream) {
    return geomae_sst_ffn_qkv_forward(x, attn_bf16, w, num_tokens, z, xhat1, xhat2, hp_bf16, rstd, nullptr, nullptr, nullptr,
                                      nullptr, nullptr, nullptr, stream);
}

// geomae_sst_stack_backward (saved activations in bf16): the next geomae_sst_ffn_backward of this thread stores no y copy and
// the next geomae_sst_weight_grad takes its `y_bf16` argument as the SAVED xhat1 and forms y = g1 * xhat1 + be1 on load
static thread_local bool t_y_from_xhat = false;
static thread_local const float *t_y_gamma = nullptr, *t_y_beta = nullptr;
static thread_local bool t_x_from_xhat = false, t_skip_x_copy = false;
static thread_local const float *t_x_gamma = nullptr, *t_x_beta = nullptr;
void geomae::set_x_from_xhat(bool on, const float* gamma2, const float* beta2) {
    t_x_from_xhat = on; t_x_gamma = gamma2; t_x_beta = beta2;
}
void geomae::set_skip_x_copy(bool on) { t_skip_x_copy = on; }
bool geomae::skip_x_copy() { return t_skip_x_copy; }
void geomae::set_y_from_xhat(bool on, const float* gamma1, const float* beta1) {
    t_y_from_xhat = on; t_y_gamma = gamma1; t_y_beta = beta1;
}

extern "C" int geomae_sst_ffn_backward(const float* xhat1, const float* xhat2, const void* hp_bf16,
                                       const float* rstd, const float* dz, const GeomaeSstLayerWeights* w,
                                       int32_t num_tokens, float* dx_res, void* dattn_bf16, void* du_bf16,
                                       void* dv_bf16, void* dhp_bf16, void* y_bf16, void* h_bf16,
                                       const GeomaeSstLayerGrads* grads, const void* up_dqkv_bf16, const float* up_dx_res,
                                       const GeomaeSstLayerWeights* up_w, hipStream_t stream) {
    if (num_tokens <= 0) return GEOMAE_OK;
    int rc = check_weights(w, "sst_ffn_backward");
    if (rc) return rc;
    GEOMAE_CHECK_TOKENS(num_tokens, "sst_ffn_backward");
    GEOMAE_REQUIRE(xhat1 && xhat2 && hp_bf16 && rstd && dx_res && dattn_bf16 && du_bf16 && dv_bf16 && dhp_bf16 && y_bf16 &&
                   h_bf16, "sst_ffn_backward: null argument");
    GEOMAE_REQUIRE((dz != nullptr) != (up_dqkv_bf16 != nullptr), "sst_ffn_backward: pass dz OR the upper layer's dqkv");
    if (up_dqkv_bf16) {
        GEOMAE_REQUIRE(up_dx_res && up_w, "sst_ffn_backward: the fused B1 head needs up_dx_res and up_w");
        if ((rc = check_weights(up_w, "sst_ffn_backward(up)"))) return rc;
    }
    GEOMAE_REQUIRE(grads && grads->ln1_w && grads->ln1_b && grads->ln2_w && grads->ln2_b, "sst_ffn_backward: null grads");
    const FfnBwdArgs A = {xhat1, xhat2, (const bf16_t*)hp_bf16, rstd, dz, to_layer(w), num_tokens, dx_res,
                          (bf16_t*)dattn_bf16, (bf16_t*)du_bf16, (bf16_t*)dv_bf16, (bf16_t*)dhp_bf16,
                          t_y_from_xhat ? nullptr : (bf16_t*)y_bf16,
                          (bf16_t*)h_bf16, grads->ln1_w, grads->ln1_b, grads->ln2_w, grads->ln2_b,
                          (const bf16_t*)up_dqkv_bf16, up_dx_res, up_w ? (const bf16_t*)up_w->wqkT_p : nullptr,
                          up_w ? (const bf16_t*)up_w->wvT_p : nullptr, layer_layout(), dz ? dz_addend() : nullptr,
                          dz ? t_skip_rows / 64 : 0};
    const int n_ffn = cdiv(cdiv(num_tokens, 16), kLayerBlk / 64);
    if (!g_pending_dw.active) {
        hipLaunchKernelGGL(sst_ffn_bwd_kernel, dim3(n_ffn), dim3(kLayerBlk), 0, stream, A);
        return check_launch("sst_ffn_bwd_kernel");
    }
    // geomae::fuse_next_ffn_backward_with(): this call also carries a weight-gradient contraction
    const PendingDw P = g_pending_dw;
    g_pending_dw.active = false;
    int gx, chunk;
    dw_grid(P.num_tasks, P.num_tokens, &gx, &chunk);
    GEOMAE_REQUIRE(!P.tasks.partial || (long long)gx * P.num_tasks * 65536 <= kDwPartialBytes, "weight_grad: split-K workspace too small");
    const DwReduce Rd = take_pending_reduce();          // the partials of the contraction before this one
    const int dw_blocks = gx * P.num_tasks;
    hipLaunchKernelGGL(sst_ffn_bwd_dw_kernel, dim3(n_ffn + dw_blocks + (Rd.partial ? kDwReduceBlocks : 0)), dim3(kLayerBlk), 0,
                       stream, A, n_ffn, P.tasks, P.num_tokens, chunk, gx, dw_blocks, Rd);
    note_partials(P.tasks, P.num_tasks, gx);
    set_last_kernel_variant(1);
    return check_launch("sst_ffn_bwd_dw_kernel");
}

extern "C" int geomae_sst_qkv_backward(const void* dqkv_bf16, const float* dx_res, const GeomaeSstLayerWeights* w,
                                       int32_t num_tokens, float* dx, hipStream_t stream) {
    if (num_tokens <= 0) return GEOMAE_OK;
    int rc = check_weights(w, "sst_qkv_backward");
    if (rc) return rc;
    GEOMAE_CHECK_TOKENS(num_tokens, "sst_qkv_backward");
    GEOMAE_REQUIRE(dqkv_bf16 && dx_res && dx, "sst_qkv_backward: null argument");
    const int tiles = cdiv(num_tokens, 16);
    int n_out = 0;
    const int32_t* orows = output_rows(&n_out);
    int tail_from = 0;
    float* tsum = tail_sum(&tail_from);
    hipLaunchKernelGGL(sst_qkv_bwd_kernel, dim3(cdiv(tiles, kLayerBlk / 64)), dim3(kLayerBlk), 0, stream,
                       (const bf16_t*)dqkv_bf16, dx_res, to_layer(w), num_tokens, dx, layer_layout(), orows, n_out, tsum, tail_from);
    return check_launch("sst_qkv_bwd_kernel");
}

extern "C" int geomae_sst_weight_grad(int32_t num_tokens, const void* dqkv_bf16, const void* xp_bf16,
                                      const void* x_bf16, const void* du_bf16, const void* attn_bf16,
                                      const void* dhp_bf16, const void* y_bf16, const void* dv_bf16,
                                      const void* h_bf16, const GeomaeSstLayerGrads* g, hipStream_t stream) {
    if (num_tokens <= 0) return GEOMAE_OK;
    GEOMAE_REQUIRE(dqkv_bf16 && xp_bf16 && x_bf16 && du_bf16 && attn_bf16 && dhp_bf16 && y_bf16 && dv_bf16 && h_bf16,
                   "sst_weight_grad: null argument");
    GEOMAE_REQUIRE(g && g->wqkv && g->bqkv && g->wo && g->bo && g->w1 && g->b1 && g->w2 && g->b2,
                   "sst_weight_grad: null grads");
    const bf16_t *dqkv = (const bf16_t*)dqkv_bf16, *xp = (const bf16_t*)xp_bf16, *xb = (const bf16_t*)x_bf16,
                 *du = (const bf16_t*)du_bf16, *at = (const bf16_t*)attn_bf16, *dhp = (const bf16_t*)dhp_bf16,
                 *y = (const bf16_t*)y_bf16, *dv = (const bf16_t*)dv_bf16, *h = (const bf16_t*)h_bf16;
    DwTasks T;
    T.blocked = (layer_layout() & kLayBlocked) ? 1 : 0;
    if (float* ws = dw_partial()) {                     // the stack's split-K workspace: two buffers, alternating
        static thread_local int flip = 0;
        flip ^= 1;
        T.partial = ws + (size_t)flip * (kDwPartialBytes / 4);
    }
    //          A     lda a0   B   ldb b0  C        ldc  r0   c0  dbias     rows
    T.t[0] = {dqkv, 384, 0,   xp, 128, 0, g->wqkv, 128, 0,   0,  g->bqkv, 128};   // dWq
    T.t[1] = {dqkv, 384, 128, xp, 128, 0, g->wqkv, 128, 128, 0,  g->bqkv, 128};   // dWk
    T.t[2] = {dqkv, 384, 256, xb, 128, 0, g->wqkv, 128, 256, 0,  g->bqkv, 128};   // dWv
    if (t_x_from_xhat) {                                // `x` is the saved xhat2 of the layer below: x = g2 * xhat2 + be2 on load
        GEOMAE_REQUIRE(t_x_gamma && t_x_beta, "sst_weight_grad: x from xhat2 needs the LayerNorm-2 parameters of the layer below");
        T.t[2].b_scale = t_x_gamma;
        T.t[2].b_shift = t_x_beta;
    }
    T.t[3] = {du,   128, 0,   at, 128, 0, g->wo,   128, 0,   0,  g->bo,   128};   // dWo
    T.t[4] = {dhp,  256, 0,   y,  128, 0, g->w1,   128, 0,   0,  g->b1,   128};   // dW1 rows 0..127
    T.t[5] = {dhp,  256, 128, y,  128, 0, g->w1,   128, 128, 0,  g->b1,   128};   // dW1 rows 128..255
    if (t_y_from_xhat) {                                // `y` is the saved xhat1: y = g1 * xhat1 + be1 formed on load
        GEOMAE_REQUIRE(t_y_gamma && t_y_beta, "sst_weight_grad: y from xhat1 needs the LayerNorm-1 parameters");
        T.t[4].b_scale = T.t[5].b_scale = t_y_gamma;
        T.t[4].b_shift = T.t[5].b_shift = t_y_beta;
    }
    T.t[6] = {dv,   128, 0,   h,  256, 0,   g->w2, 256, 0,   0,   g->b2,   128};  // dW2 cols 0..127
    T.t[7] = {dv,   128, 0,   h,  256, 128, g->w2, 256, 0,   128, nullptr, 128};  // dW2 cols 128..255
    PendingDw P{T, 8, num_tokens, true};
    const int dead = take_dw_dead_rows();
    const bool layer_form = tuning().dw_layer_form != 0;   // (A/B)
    if (layer_form && T.blocked && dw_partial()) {      // the stacks' contractions: four jobs of the layer-form kernel
        P.has_layer = true;
        P.partial_base = dw_partial();
        DlJob* J = P.layer;
        memset(J, 0, sizeof(P.layer));
        const int dead64 = dead / 64 * 64;              // (the forward skips whole 64-token workgroups)
        J[0].kind = kDlTall; J[0].tok_begin = 0;                                     // [dq | dk]^T (x + pos)
        J[0].s[0] = {dqkv, 24, 0, 16}; J[0].s[1] = {xp, 8, 0, 8};
        J[0].out[0] = {g->wqkv, g->bqkv, nullptr, nullptr, 128, 0};
        J[1].kind = kDlTall; J[1].tok_begin = dead64;                                // dhp^T y
        J[1].s[0] = {dhp, 16, 0, 16}; J[1].s[1] = {y, 8, 0, 8};
        J[1].out[0] = {g->w1, g->b1, t_y_from_xhat ? t_y_gamma : nullptr, t_y_from_xhat ? t_y_beta : nullptr, 128, 0};
        J[2].kind = kDlWide; J[2].tok_begin = dead64;                                // dv^T gelu(hp)
        J[2].s[0] = {dv, 8, 0, 8}; J[2].s[1] = {h, 16, 0, 16};
        J[2].out[0] = {g->w2, g->b2, nullptr, nullptr, 256, 0};
        J[3].kind = kDlDual; J[3].tok_begin = 0;                                     // dv_^T x  |  du^T attn
        J[3].s[0] = {dqkv, 24, 16, 8}; J[3].s[1] = {xb, 8, 0, 8};
        J[3].s[2] = {du, 8, 0, 8}; J[3].s[3] = {at, 8, 0, 8};
        J[3].out[0] = {g->wqkv + 256 * 128, g->bqkv + 256, t_x_from_xhat ? t_x_gamma : nullptr, t_x_from_xhat ? t_x_beta : nullptr, 128, 0};
        J[3].out[1] = {g->wo, g->bo, nullptr, nullptr, 128, 0};
    }
    if (t_defer_weight_grad && t_defer_all) {          // queued for geomae_flush_weight_grad
        t_defer_weight_grad = false;
        t_dw_queue.push_back(P);
        return GEOMAE_OK;
    }
    if (t_defer_weight_grad) {          // sst_stack_backward: ride on the next layer's ffn-backward launch (or wait for the flush)
        t_defer_weight_grad = false;
        g_pending_dw = P;
        return GEOMAE_OK;
    }
    if (P.has_layer) return launch_dw_layers(&P, 1, stream);
    return launch_dw(T, 8, num_tokens, stream);
}

void geomae::defer_next_weight_grad() { t_defer_weight_grad = true; }
void geomae::drop_pending_weight_grads() {
    t_defer_weight_grad = false;
    g_pending_dw.active = false;
    t_dw_queue.clear();
    g_pending_reduce = DwReduce();
}
void geomae::set_defer_all_weight_grads(bool on) { t_defer_all = on; }
bool geomae::defer_all_weight_grads() { return t_defer_all; }
extern "C" int geomae_flush_weight_grad(hipStream_t stream) { return geomae::flush_pending_weight_grad(stream); }
int geomae::flush_pending_weight_grad(hipStream_t stream) {
    t_defer_weight_grad = false;
    int rc = GEOMAE_OK;
    if (g_pending_dw.active) {
        g_pending_dw.active = false;
        rc = g_pending_dw.has_layer ? launch_dw_layers(&g_pending_dw, 1, stream)
                                    : launch_dw(g_pending_dw.tasks, g_pending_dw.num_tasks, g_pending_dw.num_tokens, stream);
    }
    if (!t_dw_queue.empty()) {                           // "defer all": every contraction queued since the last flush
        // (one launch per layer.  Four layers of a stack in ONE launch -- 32 tasks, 768 workgroups -- were measured: the
        // queue is shorter, but such a launch takes the whole chip and the latency-bound encoder backward beside it lost
        // 0.09 instead of 0.05 ms)
        // (... and with FEWER workgroups per queued launch -- 16 / 12 / 8 token chunks per task instead of 24 -- the step
        // took 2.023 / 2.044 / 2.127 instead of 2.02 ms: the contractions then sit beside the encoder for longer)
        // (round 5, layer form: up to FOUR layers of a stack are one launch of <= 16 jobs x 12 chunks = 192 workgroups of one
        // per CU -- the fixed costs of a launch once per stack, a quarter of the split-K partials; dw_device.h)
        for (size_t k = 0; k < t_dw_queue.size() && rc == GEOMAE_OK;) {
            const PendingDw& P = t_dw_queue[k];
            if (!P.has_layer) { rc = launch_dw(P.tasks, P.num_tasks, P.num_tokens, stream); ++k; continue; }
            size_t m = 1;
            while (m < 4 && k + m < t_dw_queue.size() && t_dw_queue[k + m].has_layer && t_dw_queue[k + m].num_tokens == P.num_tokens &&
                   t_dw_queue[k + m].partial_base == P.partial_base)
                ++m;
            rc = launch_dw_layers(&t_dw_queue[k], (int)m, stream);
            k += m;
        }
        t_dw_queue.clear();
    }
    if (g_pending_reduce.partial) {                      // the last contraction's own partials
        const DwReduce Rd = take_pending_reduce();
        hipLaunchKernelGGL(dw_reduce_kernel, dim3(4 * kDwReduceBlocks), dim3(256), 0, stream, Rd);      // alone on its stream: wider
        const int rc2 = check_launch("dw_reduce_kernel");
        if (rc == GEOMAE_OK) rc = rc2;
    }
    return rc;
}

// `count` <= 4 layers of one stack (same token set, same workspace) as ONE launch of the layer-form contraction + its reduction
int geomae::launch_dw_layers(const PendingDw* P, int count, hipStream_t stream) {
    GEOMAE_REQUIRE(count >= 1 && 4 * count <= kDlMaxJobs, "weight_grad: more than four layers per launch");
    // partials an old-form launch left in the same workspace: summed first (stream order keeps them intact until then)
    if (g_pending_reduce.partial) {
        const DwReduce Rd = take_pending_reduce();
        hipLaunchKernelGGL(dw_reduce_kernel, dim3(4 * kDwReduceBlocks), dim3(256), 0, stream, Rd);
        const int rc0 = check_launch("dw_reduce_kernel");
        if (rc0) return rc0;
    }
    DlArgs A;
    memset(&A, 0, sizeof(A));
    A.njobs = 4 * count; A.n = P[0].num_tokens; A.partial = P[0].partial_base;
    for (int l = 0; l < count; ++l) memcpy(&A.job[4 * l], P[l].layer, sizeof(P[l].layer));
    // workgroups = jobs x token chunks, one per CU (129 KB of LDS).  Alone, 192 fill the memory system (tools/dw_bench.hip: 4
    // layers at 22 k tokens 87 / 71 / 65 / 65 us at 6 / 8 / 12 / 16 chunks; one layer 58 / 44 / 37 / 33 us at 8 / 12 / 16 / 24) --
    // but these launches run on a side stream BESIDE the backward of the next stack, which needs the CUs more than they do:
    // in the step 96 workgroups are the optimum at every size (config 2: 1.860 / 1.860 / 1.86 / 1.91 ms at 4 / 6 / 8 / 12 chunks
    // of 16 jobs; config 3: 7.29 / 6.37 / 6.07 / 5.91 / 6.00 / 6.15 ms at 2 / 3 / 4 / 6 / 8 / 12 -- fewer do not finish before
    // the step's join, more slow the main stream's kernels)
    const int g_env = tuning().dw_chunks;                // (A/B)
    // (round 5, with the one-launch encoder backward: its ~165 workgroups hold a CU each too, and 165 + 96 > 256 sent some of
    //  them to a second round -- 80 workgroups at the sizes where that kernel runs beside these launches: config 2
    //  1.735 / 1.700 / 1.71 / 1.81 ms at 96 / 80 / 64 / 48)
    // (GeomaeTuning.dw_budget_mid = w: the budget of launches of 12289-32768 tokens -- config 2's decoders, which run beside the OTHER
    //  decoder's backward, not beside the one-launch encoder kernels -- A/B)
    const int mid_env = tuning().dw_budget_mid;
    const int hint = take_dw_budget_hint();
    const int budget = hint > 0 ? hint : A.n <= 12288 ? 80 : A.n <= 32768 ? (mid_env > 0 ? mid_env : 80) : 96;
    int G = budget / A.njobs;
    if (G > 24) G = 24;
    if (g_env > 0) G = g_env;
    const int by_tokens = cdiv(A.n, 2 * kDlSlabTok);                   // at least two slabs per workgroup
    if (G > by_tokens) G = by_tokens;
    if (G > kDlMaxChunks) G = kDlMaxChunks;
    while (G > 1 && (long long)A.njobs * G * kDlPartialFloats * 4 > 2 * kDwPartialBytes) --G;
    if (G < 1) G = 1;
    A.G = G;
    void* prof = thread_profiler();
    const bool timed = profiler_begin(prof, GEOMAE_KERNEL_DW, stream);
    hipLaunchKernelGGL(dw_layer_kernel, dim3(A.njobs * G), dim3(kDlThreads), 0, stream, A);
    if (timed) profiler_end(prof, stream);
    int rc = check_launch("dw_layer_kernel");
    if (rc) return rc;
    DlReduce R;
    R.partial = A.partial; R.njobs = A.njobs; R.G = G;
    for (int j = 0; j < A.njobs; ++j) { R.job[j].kind = A.job[j].kind; R.job[j].pad_ = 0; R.job[j].out[0] = A.job[j].out[0]; R.job[j].out[1] = A.job[j].out[1]; }
    hipLaunchKernelGGL(dw_layer_reduce_kernel, dim3(cdiv(A.njobs * kDlTileSlots, 256)), dim3(256), 0, stream, R);
    return check_launch("dw_layer_reduce_kernel");
}

// ONE [128,128] product over n tokens of two tile-blocked [n,128] operands (the VFE's layer-1 weight gradient) as a SPLIT
// job of the layer-form kernel + its reduction: C += A^T B.  tools/dw_bench.hip at 106 k points: 29 us with the reduction at
// 96 workgroups (32 / 48 / 64: 40 / 33 / 30) against 33 + 11 us for dw_kernel + dw_reduce_kernel; 1.03 M points: 133 us.
int geomae::launch_dw_split(const bf16_t* A, const bf16_t* B, int n, float* C, float* partial, hipStream_t stream) {
    if (g_pending_reduce.partial) {                      // (an old-form launch's partials in the same workspace first)
        const DwReduce Rd = take_pending_reduce();
        hipLaunchKernelGGL(dw_reduce_kernel, dim3(4 * kDwReduceBlocks), dim3(256), 0, stream, Rd);
        const int rc0 = check_launch("dw_reduce_kernel");
        if (rc0) return rc0;
    }
    DlArgs S;
    memset(&S, 0, sizeof(S));
    S.njobs = 1; S.n = n; S.partial = partial;
    S.job[0].kind = kDlSplit;
    S.job[0].s[0] = {A, 8, 0, 8}; S.job[0].s[1] = {B, 8, 0, 8}; S.job[0].s[2] = {A, 8, 16, 8}; S.job[0].s[3] = {B, 8, 16, 8};
    S.job[0].out[0] = {C, nullptr, nullptr, nullptr, 128, 0};
    int G = cdiv(n, 4 * 2 * kDlSlabTok);                 // at least four 64-token slabs per workgroup
    if (G > kDlMaxChunks) G = kDlMaxChunks;
    while (G > 1 && (long long)G * kDlPartialFloats * 4 > 2 * kDwPartialBytes) --G;
    if (G < 1) G = 1;
    S.G = G;
    void* prof = thread_profiler();
    const bool timed = profiler_begin(prof, GEOMAE_KERNEL_DW, stream);
    hipLaunchKernelGGL(dw_layer_kernel, dim3(G), dim3(kDlThreads), 0, stream, S);
    if (timed) profiler_end(prof, stream);
    int rc = check_launch("dw_layer_kernel");
    if (rc) return rc;
    DlReduce R;
    R.partial = partial; R.njobs = 1; R.G = G;
    R.job[0].kind = kDlSplit; R.job[0].pad_ = 0; R.job[0].out[0] = S.job[0].out[0]; R.job[0].out[1] = S.job[0].out[1];
    const bool two_level = tuning().dw_split_reduce != 0;   // (0: A/B)
    if (!two_level) {
        hipLaunchKernelGGL(dw_layer_reduce_kernel, dim3(cdiv(kDlTileSlots, 256)), dim3(256), 0, stream, R);
        return check_launch("dw_layer_reduce_kernel");
    }
    hipLaunchKernelGGL(dw_split_reduce_kernel, dim3(kDlTileSlots / 2 / 16), dim3(256), 0, stream, R);
    return check_launch("dw_split_reduce_kernel");
}

int geomae::launch_dw(const DwTasks& T, int num_tasks, int num_tokens, hipStream_t stream) {
    int G, chunk;
    dw_grid(num_tasks, num_tokens, &G, &chunk, T.partial != nullptr);
    GEOMAE_REQUIRE(!T.partial || (long long)G * num_tasks * 65536 <= kDwPartialBytes, "weight_grad: split-K workspace too small");
    const DwReduce Rd = take_pending_reduce();
    void* prof = thread_profiler();
    const bool timed = profiler_begin(prof, GEOMAE_KERNEL_DW, stream);
    hipLaunchKernelGGL(dw_kernel, dim3(G, num_tasks + (Rd.partial ? cdiv(2 * kDwReduceBlocks, G) : 0)), dim3(256), 0, stream, T, num_tokens, chunk,
                       num_tasks, Rd);
    if (timed) profiler_end(prof, stream);
    note_partials(T, num_tasks, G);
    return check_launch("dw_kernel");
}
