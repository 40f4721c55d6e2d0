// Device helpers shared by the one-launch SST layer kernels (sst_fused.hip: one bundle per workgroup; sst_ws.hip: the
// weight-stationary form that loops over bundles).  See sst_fused.hip for the design notes.
#pragma once
#include "common.h"
#include "sst_device.h"

namespace geomae {

typedef __attribute__((ext_vector_type(4))) short bf16x4_s;

__device__ __forceinline__ bf16x4_s as_bf4(uint2 v) { return __builtin_bit_cast(bf16x4_s, v); }
__device__ __forceinline__ f32x4 mfma16(uint2 a, uint2 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(as_bf4(a), as_bf4(b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma32_2(uint2 a0, uint2 a1, uint2 b0, uint2 b1, f32x4 c) {
    return mfma32(make_uint4(a0.x, a0.y, a1.x, a1.y), make_uint4(b0.x, b0.y, b1.x, b1.y), c);
}

// phase stamps (tools/fused_layer_time.py builds a second library with -DGEOMAE_PHASE_TIMING; a no-op in the product build)
#ifdef GEOMAE_PHASE_TIMING
#define FUSED_STAMP(i)                                                                                          \
    do {                                                                                                        \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < GEOMAE_STAMP_BLOCKS)                                        \
            geomae_stamps[blockIdx.x * GEOMAE_STAMP_SLOTS + (threadIdx.x >> 8) * 16 + (i)] = clock64();           \
    } while (0)
#else
#define FUSED_STAMP(i) do {} while (0)
#endif

constexpr int kFusedThreads = 512;
constexpr int kFMaxT = 144;              // tokens per bundle (a 12 x 12 window)
constexpr int kFRow = 2 * (128 + 8);     // bytes of one bf16 row of 128 channels in LDS (+16 B: conflict-free b128 reads)
constexpr int kFRowH = 2 * (256 + 8);    // ... of 256 channels
constexpr int kFOor = 0x7fff0000;        // a byte offset past every buffer: loads return 0, stores are dropped
// LDS: [X / Y | XP | O] (H aliases XP + O) | LayerNorm statistics [token][wave][2] | window start / end per position
constexpr int kFLdsX = 0, kFLdsXP = kFMaxT * kFRow, kFLdsO = 2 * kFMaxT * kFRow, kFLdsRed = 3 * kFMaxT * kFRow;
constexpr int kFLdsWl = kFLdsRed + kFMaxT * 64, kFLdsWh = kFLdsWl + kFMaxT * 4, kFLdsPrm = kFLdsWh + kFMaxT * 4;
constexpr int kFLdsBytes = kFLdsPrm + 1408 * 4;
static_assert(kFMaxT * kFRowH <= 2 * kFMaxT * kFRow, "H must fit in XP + O");

struct FusedFwd {
    const float* x;              // layer input, token order: tile-blocked [ceil16(n)][128] fp32 -- unless M.src is set
    SstInputMap M;               // first layer of a stack: row-major source rows (+ row map, + fill row), common.h
    const int32_t* bun_tok;      // [NB + 1] bundle b covers plan positions [bun_tok[b], bun_tok[b + 1])
    const int4* plan;            // per position: (token, in-window position, window start, window end)
    const int32_t* num_bundles;
    const float* pos_table;      // [wx * wy][128]
    LayerW W;
    int n;
    float eps;
    float* z;                    // layer output [n][128] fp32: tile-blocked (z_blocked) or row-major
    int z_blocked;
    const int4* items = nullptr;         // forward work items (window.hip item_pack: first position, positions, first query tile,
    const int32_t* num_items = nullptr;  // query tiles), or null / 0 items: the kernel walks bun_tok
    int big_follows = 1;         // the launch for bundles of more than four tiles follows (0: the caller promised there is none)
    // saved for the backward (token order, tile-blocked; all or none)
    bf16_t *qkv, *attn, *xh1, *xh2, *hp, *xb, *xp;
    float *lse, *rstd;
};

// byte offset of lane (token, g)'s 4 channels of channel tile ct in a tile-blocked [.][ld] tensor of E-byte elements
template <int E>
__device__ __forceinline__ int blk_off(int tok, int ld, int ct, int g) {
    return (tok >> 4) * (16 * ld * E) + ct * (256 * E) + (tok & 15) * (16 * E) + g * (4 * E);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t whole_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(p), 0, kFOor, 0x00020000);   // (offsets >= kFOor are out of range)
}
// (timing ablations: FUSED_ABL_NO_SAVE drops every store of the saved activations -- descriptors of zero records --,
// FUSED_ABL_NO_Z the layer's output too: what the bytes a launch leaves dirty in L2 cost at its end)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t saved_rsrc(const void* p) {
#ifdef FUSED_ABL_NO_SAVE
    return __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(p), 0, 0, 0x00020000);
#else
    return whole_rsrc(p);
#endif
}
__device__ __forceinline__ void buf_store_f32x4(__amdgpu_buffer_rsrc_t r, int off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, 0);
}
__device__ __forceinline__ void buf_store_f32(__amdgpu_buffer_rsrc_t r, int off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, off, 0, 0);
}
__device__ __forceinline__ uint4 lds_b128(const char* p) { return *reinterpret_cast<const uint4*>(p); }

// A fragments of output tile `ot` of a FRAGMENT-MAJOR packed [N][K] matrix (pack_weights_kernel, tr & 4): one contiguous
// 1-KB piece per (tile, k step).  (Read as 16 rows x 64 B from the row-major copy the same fetch took 8 k cycles longer
// per workgroup: every 128-byte line was requested twice, by two different instructions.)
template <int K>
__device__ __forceinline__ void load_wfrag(const bf16_t* __restrict__ Wf, int ot, int lane, uint4 (&f)[K / 32]) {
#ifdef FUSED_ABL_NO_WEIGHTS          // (timing ablation: what the weight fetch costs the chain)
#pragma unroll
    for (int kk = 0; kk < K / 32; ++kk) f[kk] = make_uint4(lane, ot, kk, 0x3c003c00u);
    return;
#endif
    const bf16_t* p = Wf + (size_t)ot * (16 * K) + 8 * lane;
#pragma unroll
    for (int kk = 0; kk < K / 32; ++kk) f[kk] = *reinterpret_cast<const uint4*>(p + 512 * kk);
}
__device__ __forceinline__ f32x4 load_f4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// LayerNorm over the 128 channels of a token that 8 waves hold 16 channels each of: every wave leaves (sum, sum of squares)
// of its 16 channels -- two INDEPENDENT cross-lane reductions (the Welford form, mean first and then the centred squares,
// was one dependent chain twice as long on the critical path of two phases) -- and every wave merges the eight pairs.
// The inputs are residual sums of LayerNorm outputs (|mean| of the order of sigma), far from the cancellation regime of
// E[x^2] - E[x]^2 in fp32.
__device__ __forceinline__ void ln_merge(const float* red_tok, float eps, float* mean_out, float* rstd_out) {
    f32x4 a[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const f32x4*>(red_tok + 4 * i);
    const f32x4 s = (a[0] + a[1]) + (a[2] + a[3]);
    const float mean = (s[0] + s[2]) * (1.0f / 128.0f);
    const float var = fmaxf((s[1] + s[3]) * (1.0f / 128.0f) - mean * mean, 0.0f);
    *mean_out = mean;
    *rstd_out = rsqrtf(var + eps);
}
// this wave's (sum, sum of squares) of the 16 channels it holds of token (lane & 15): u = 4 channels per lane group
__device__ __forceinline__ void ln_partial(const f32x4 u, float* red_tok_wave, int g) {
    const float s = rows4_sum((u[0] + u[1]) + (u[2] + u[3]));
    const float q = rows4_sum((u[0] * u[0] + u[1] * u[1]) + (u[2] * u[2] + u[3] * u[3]));
    if (g == 0) *reinterpret_cast<float2*>(red_tok_wave) = make_float2(s, q);
}

// fp32 parameter vectors of the layer, fetched once per workgroup into LDS (read back with ds_read: a global load at the
// point of use waits behind every store the wave issued before it -- vmcnt retires in order on gfx9)
constexpr int kPBq = 0, kPBo = 384, kPG1 = 512, kPBe1 = 640, kPB1 = 768, kPB2 = 1024, kPG2 = 1152, kPBe2 = 1280, kPFloats = 1408;
__device__ __forceinline__ f32x4 params_issue(const LayerW& W) {
    const int s = threadIdx.x;                                        // float4 slot 0..351
    const float* src = s < 96 ? W.bqkv + 4 * s : s < 128 ? W.bo + 4 * (s - 96) : s < 160 ? W.g1 + 4 * (s - 128)
                     : s < 192 ? W.be1 + 4 * (s - 160) : s < 256 ? W.b1 + 4 * (s - 192) : s < 288 ? W.b2 + 4 * (s - 256)
                     : s < 320 ? W.g2 + 4 * (s - 288) : W.be2 + 4 * (s - 320);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (s < kPFloats / 4) v = *reinterpret_cast<const f32x4*>(src);
    return v;
}

}  // namespace geomae
