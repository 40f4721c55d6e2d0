// Weight-gradient contraction of one SST layer, stand-alone form (round 5): dW = dY^T X over the tokens of a stack,
// for the seven weight matrices of EncoderLayer (reference autograd of mmdet3d/models/sst/sst_basic_block.py:85-102).
//
// What the round-1..4 kernel (dw_body, sst_layer.hip) was: eight [128,128] tasks per layer, 256-thread workgroups, operand
// slabs global -> registers (a three-slab ring, 96 VGPRs) -> LDS -> transposing reads; 2.3 TB/s of counter bytes, 1.6x the
// algorithmic ones.  This one:
//  * FOUR jobs per layer instead of eight tasks -- task pairs that share an operand are one job: QK = [dq | dk]^T xp
//    (256 x 128), W1 = dhp^T y (256 x 128), W2 = dv^T h (128 x 256), VO = dv_^T x and du^T attn side by side.  3.25 KB of
//    operands per token and layer instead of 4 KB.
//  * operand slabs go global -> LDS DIRECTLY (buffer_load ... lds, 16 B per lane, 1 KB per wave instruction): the stacks'
//    tile-blocked layout [n/16][C/16][16 tokens][16 channels] keeps every 16-token x C piece contiguous in memory, and the
//    LDS image is that piece as it lies -- a 16-lane group of ds_read_b64_tr_b16 reads 4 tokens x 16 channels = 128
//    contiguous bytes, two groups per 256-byte bank row: conflict-free without a swizzle.  No staging registers; a ring
//    of four 32-token slabs (24-32 KB each) with counted vmcnt waits and ONE raw barrier per slab keeps 2-3 slabs in
//    flight per CU.
//  * 512 threads, wave = a 64 x 64 tile of the job's 256 x 128 outputs: 8 fragment reads per 16 MFMAs (the old 32 x 128
//    wave tile: 10).
//  * operands "formed on load" (y = gamma1 xhat1 + beta1, x = gamma2 xhat2 + beta2) are not formed at all: the
//    contraction is linear, dY^T (xhat diag(gamma) + 1 beta^T) = (dY^T xhat) diag(gamma) + (sum_t dY) beta^T, and the column
//    sums of dY are the bias gradient the kernel computes anyway (one MFMA against a ones fragment) -- the affine is
//    applied to the [64,64] accumulator tile in the epilogue, in fp32.
//  * the dead rows of a decoder's last layer (set_first_live_row) are SKIPPED by the jobs whose dY operand is zero there
//    (W1, W2) instead of being read and clamped.
//  * the layers of a stack whose contractions are deferred anyway (the decoders': they run on the geometry stream behind
//    their stack's backward) are ONE launch: up to 16 jobs x G token chunks.  The fixed costs of a launch (dispatch of
//    129-KB-LDS workgroups, first-slab latency, the 128-KB partial store: ~7 us) are paid once per stack, and a job's token
//    range is split over 8-16 workgroups instead of 24: a quarter of the split-K partial traffic.
//  * split-K through the caller's workspace as before (accumulator order, 16 B per lane), summed by ONE reduction launch
//    behind the contraction (dw_layer_reduce_kernel: all partials of a slot in flight at once).  Bias partials travel the
//    same way: no atomics at all, results independent of arrival order.  (Measured and dropped: the previous launch's
//    partials summed by this launch's workgroups under their first slabs' latency -- 9 us per launch, three dependent
//    rounds of loads in two thirds of the workgroups.)
#pragma once
#include "sst_device.h"

namespace geomae {

constexpr int kDlThreads = 512;
constexpr int kDlSlabTok = 32;                    // one MFMA K-step
#ifndef GEOMAE_DL_RING
#define GEOMAE_DL_RING 4
#endif
constexpr int kDlRing = GEOMAE_DL_RING;          // slabs of the LDS ring (tools/dw_bench.hip built with 2: 119 instead of 83 us)
constexpr int kDlSlabBytes = 32768;               // VO: 4 streams x 2 token blocks x 4 KB; QK / W1 / W2 use 24 KB of it
constexpr int kDlLdsBias = kDlRing * kDlSlabBytes;
constexpr int kDlLdsBytes = kDlLdsBias + 256 * 4;
constexpr int kDlTileSlots = 16 * kDlThreads;     // f32x4 slots of one workgroup's partial: 16 tiles x 512 threads
constexpr int kDlPartialFloats = kDlTileSlots * 4 + 256;   // + the job's 256 column sums of dY
constexpr int kDlMaxChunks = 96;             // (one SPLIT job over 10^5-10^6 points: up to 96 workgroups)
constexpr int kDlOor = 0x7fff0000;

enum { kDlTall = 0 /* [256 x 128]: A 16 blocks, B 8 */, kDlWide = 1 /* [128 x 256]: A 8, B 16 */,
       kDlDual = 2 /* two [128 x 128]: (A0, B0, A1, B1), 8 blocks each */,
       kDlSplit = 3 /* ONE [128 x 128] over 64-token slabs: waves 0-3 take tokens 0..31 of a slab, waves 4-7 tokens 32..63 --
                       the DUAL image with (A1, B1) = (A0, B0) two token blocks further on (blk0 + 2 ldblk); the two halves'
                       partials are summed by the reduction.  The VFE's layer-1 weight gradient: one product over 10^5-10^6
                       points (vfe.hip geomae_vfe_weight_grad1) */ };

constexpr int kDlMaxJobs = 16;                    // four layers of a stack per launch
struct DlStream { const bf16_t* base; int ldblk; short blk0, nblk; };     // tile-blocked [n16][ldblk][16][16]; blocks blk0 .. blk0 + nblk
struct DlOut {
    float* C;                        // the output tile set's origin: C[i * ldc + j]
    float* dbias;                    // += column sums of the A operand (rows of C), or null
    const float* b_scale;            // B operand is (scale * B + shift): applied to the accumulators, or null
    const float* b_shift;
    int ldc, pad_;
};
struct DlJob { int kind, tok_begin; DlStream s[4]; DlOut out[2]; };
struct DlArgs { DlJob job[kDlMaxJobs]; int njobs, n, G; float* partial; };      // partial: [job][chunk][kDlPartialFloats]
struct DlReduceJob { int kind, pad_; DlOut out[2]; };
struct DlReduce { const float* partial = nullptr; int njobs = 0, G = 0; DlReduceJob job[kDlMaxJobs]; };
static_assert(sizeof(DlArgs) <= 3072 && sizeof(DlReduce) <= 3072, "kernel arguments must stay well below the 4 KB segment");

// wave w of a job: which 64 x 64 tile (half = which product of a DUAL job)
__device__ __forceinline__ void dl_wave_role(int kind, int w, int* half, int* wr, int* wc) {
    if (kind == kDlTall) { *half = 0; *wr = w >> 1; *wc = w & 1; }
    else if (kind == kDlWide) { *half = 0; *wr = w >> 2; *wc = w & 3; }
    else { *half = w >> 2; *wr = (w >> 1) & 1; *wc = w & 1; }           // DUAL, SPLIT
}

__device__ __forceinline__ void dl_wait_vm(int n) {          // s_waitcnt vmcnt(n) only (gfx9 encoding)
    switch (n) {
        case 0: __builtin_amdgcn_s_waitcnt(0x0f70); break;
        case 3: __builtin_amdgcn_s_waitcnt(0x0f73); break;
        case 4: __builtin_amdgcn_s_waitcnt(0x0f74); break;
        case 6: __builtin_amdgcn_s_waitcnt(0x0f76); break;
        case 8: __builtin_amdgcn_s_waitcnt(0x0f78); break;
        default: __builtin_amdgcn_s_waitcnt(0x0f70); break;
    }
}

typedef __attribute__((address_space(3))) void dl_lds_void;

// Sum of the G partials a contraction launch left per (job, tile, thread) slot, added to the gradients with plain loads and
// stores (the only writer of those gradients on its stream); then the bias rows.  One slot per thread: the gradient words
// and ALL partials of the slot are requested before the first is used (one memory round trip per slot).
__device__ __forceinline__ void dl_reduce(const DlReduce& R, int first, int stride) {
    const int mslots = R.njobs * kDlTileSlots;
    for (int slot = first; slot < mslots; slot += stride) {
        const int jb = slot / kDlTileSlots, q = slot - jb * kDlTileSlots;
        const int tile = q >> 9, th = q & 511;
        const int a = tile >> 2, b = tile & 3, w = th >> 6, lane = th & 63, g = lane >> 4, o = lane & 15;
        int half, wr, wc;
        dl_wave_role(R.job[jb].kind, w, &half, &wr, &wc);
        const DlOut& O = R.job[jb].out[half];
        float* c0 = O.C + (int64_t)(64 * wr + 16 * a + 4 * g) * O.ldc + 64 * wc + 16 * b + o;
        float cv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) cv[r] = c0[(int64_t)r * O.ldc];
        const bool split = R.job[jb].kind == kDlSplit;
        if (split && half) continue;                               // (the first half's thread sums both halves' partials)
        const float* p = R.partial + (size_t)jb * R.G * kDlPartialFloats + 4 * q;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int u0 = 0; u0 < R.G; u0 += 16) {                     // 16 (SPLIT: 32) partials of the slot in flight at once
            f32x4 v[16], v2[16];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (u0 + u < R.G) {
                    v[u] = *reinterpret_cast<const f32x4*>(p + (size_t)(u0 + u) * kDlPartialFloats);
                    if (split) v2[u] = *reinterpret_cast<const f32x4*>(p + (size_t)(u0 + u) * kDlPartialFloats + 4 * 256);
                }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (u0 + u < R.G) {
                    s += v[u];
                    if (split) s += v2[u];
                }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) c0[(int64_t)r * O.ldc] = cv[r] + s[r];
    }
    const int bslots = R.njobs * 256;
    for (int slot = first; slot < bslots; slot += stride) {
        const int jb = slot >> 8, row = slot & 255;
        const int kind = R.job[jb].kind;
        float* dst = nullptr;
        if (kind == kDlTall) dst = R.job[jb].out[0].dbias ? R.job[jb].out[0].dbias + row : nullptr;
        else if (kind == kDlWide || kind == kDlSplit) dst = (row < 128 && R.job[jb].out[0].dbias) ? R.job[jb].out[0].dbias + row : nullptr;
        else dst = R.job[jb].out[row >> 7].dbias ? R.job[jb].out[row >> 7].dbias + (row & 127) : nullptr;
        if (!dst) continue;
        const float* p = R.partial + (size_t)jb * R.G * kDlPartialFloats + kDlTileSlots * 4 + row;
        float s = 0.f;
        for (int k = 0; k < R.G; ++k) s += p[(size_t)k * kDlPartialFloats];
        *dst += s;
    }
}

// NI: 1-KB load instructions per wave and slab (3: 24-KB slabs, 4: 32-KB slabs)
template <int NI>
__device__ __forceinline__ void dl_job_body(const DlJob& J, const int n, const int chunk_idx, const int G, float* __restrict__ pout,
                                            char* lds) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 15, g = lane >> 4;
    const int kind = J.kind;
    int half, wr, wc;
    dl_wave_role(kind, w, &half, &wr, &wc);

    // ---- this workgroup's token range: [t_begin, t_end), multiples of 32 but for the stack's end
    const int slab_tok = kind == kDlSplit ? 2 * kDlSlabTok : kDlSlabTok;     // SPLIT: each half of the waves takes 32 of 64 tokens
    const int span = n - J.tok_begin;
    int chunk = (span + G - 1) / G;
    chunk = (chunk + slab_tok - 1) / slab_tok * slab_tok;
    const int t_begin = J.tok_begin + chunk_idx * chunk;
    const int t_end = t_begin + chunk < n ? t_begin + chunk : n;
    const int nslab = t_end > t_begin ? (t_end - t_begin + slab_tok - 1) / slab_tok : 0;

    // ---- the LDS image of a slab: streams in order, two 16-token blocks each, nblk x 512 B per block.  Wave w issues the
    // 1-KB pieces q = w, w + 8, ...; piece q of stream s, token block tb, KB kb comes from
    //   base_s + ((T16 + tb) * ldblk_s + blk0_s) * 512 + kb * 1024   (T16 = slab's first token / 16)
    int voff[NI];                 // byte offset of this lane's 16 B of piece i at slab 0 of the chunk
    int vstep[NI];                // ... advance per slab
    __amdgpu_buffer_rsrc_t rs[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int q = w + 8 * i;
        int s = 0, start = 0;
        // (kind-dependent stream sizes in KB: TALL 16, 8; WIDE 8, 16; DUAL 8, 8, 8, 8)
        int size0 = J.s[0].nblk;
        while (q >= start + size0) { start += size0; ++s; size0 = J.s[s].nblk; }
        const DlStream S = J.s[s];
        const int r = q - start, per_tb = S.nblk >> 1, tb = r / per_tb, kb = r - tb * per_tb;
        const int n16 = (n + 15) >> 4;
        rs[i] = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(S.base), 0, n16 * S.ldblk * 512, 0x00020000);
        voff[i] = (((t_begin >> 4) + tb) * S.ldblk + S.blk0) * 512 + kb * 1024 + lane * 16;
        vstep[i] = (slab_tok >> 4) * S.ldblk * 512;
    }
    auto issue = [&](int slab) {                 // slab index within the chunk; past its end: out-of-range loads (no traffic)
        char* dst = lds + (slab & (kDlRing - 1)) * kDlSlabBytes + w * 1024;
        const bool live = slab < nslab;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int vo = live ? voff[i] + slab * vstep[i] : kDlOor;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[i], (dl_lds_void*)(dst + 8192 * i), 16, vo, 0, 0, 0);
        }
    };
#pragma unroll
    for (int s = 0; s < kDlRing - 1; ++s) issue(s);

    // ---- fragment addresses of this wave's 64 x 64 tile inside a slab image
    int offA, nbA, offB, nbB, ab, bb;
    if (kind == kDlTall) { offA = 0; nbA = 16; offB = 16384; nbB = 8; }
    else if (kind == kDlWide) { offA = 0; nbA = 8; offB = 8192; nbB = 16; }
    else { offA = half * 16384; nbA = 8; offB = offA + 8192; nbB = 8; }
    ab = 4 * wr; bb = 4 * wc;
    const int lane_in = (g & 1) * 128 + (m >> 2) * 32 + (m & 3) * 8;
    const int laneA = offA + (g >> 1) * nbA * 512 + ab * 512 + lane_in;
    const int laneB = offB + (g >> 1) * nbB * 512 + bb * 512 + lane_in;
    // bias (column sums of the A operand): the waves that share an A row group split its four tiles among them
    const int nshare = kind == kDlWide ? 4 : 2;
    f32x4 acc[4][4], bacc[2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    bacc[0] = bacc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint4 ones = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);

    for (int k = 0; k < nslab; ++k) {
        dl_wait_vm((kDlRing - 2) * NI);                       // slab k landed (this wave's pieces); k + 1, k + 2 in flight
        __builtin_amdgcn_s_barrier();                         // ... everybody's; and slab k - 1 is consumed
        issue(k + kDlRing - 1);                               // into the slot of slab k - 1
        const char* slab = lds + (k & (kDlRing - 1)) * kDlSlabBytes;
        uint4 af[4], bf[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const uint2 lo = tr_read(reinterpret_cast<const bf16_t*>(slab + laneA + a * 512));
            const uint2 hi = tr_read(reinterpret_cast<const bf16_t*>(slab + laneA + a * 512 + 256));
            af[a] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint2 lo = tr_read(reinterpret_cast<const bf16_t*>(slab + laneB + b * 512));
            const uint2 hi = tr_read(reinterpret_cast<const bf16_t*>(slab + laneB + b * 512 + 256));
            bf[b] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
        const int t0 = t_begin + k * slab_tok;
        if (t0 + slab_tok > t_end) {                          // the stack's last slab: tokens >= n hold anything (workgroup-uniform)
            const int tlo = t0 + (kind == kDlSplit ? kDlSlabTok * half : 0) + 16 * (g >> 1) + 4 * (g & 1);
            unsigned int mk[4];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    const int tk = tlo + 8 * h + 2 * e2;
                    mk[2 * h + e2] = (tk < t_end ? 0x0000ffffu : 0u) | (tk + 1 < t_end ? 0xffff0000u : 0u);
                }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                af[a].x &= mk[0]; af[a].y &= mk[1]; af[a].z &= mk[2]; af[a].w &= mk[3];
                bf[a].x &= mk[0]; bf[a].y &= mk[1]; bf[a].z &= mk[2]; bf[a].w &= mk[3];
            }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = mfma32(af[a], bf[b], acc[a][b]);
        // (constant fragment indices under wave-uniform branches: a select between array elements sent af[] to scratch)
        if (nshare == 2) {
            if (wc == 0) { bacc[0] = mfma32(af[0], ones, bacc[0]); bacc[1] = mfma32(af[1], ones, bacc[1]); }
            else { bacc[0] = mfma32(af[2], ones, bacc[0]); bacc[1] = mfma32(af[3], ones, bacc[1]); }
        } else {
            if (wc == 0) bacc[0] = mfma32(af[0], ones, bacc[0]);
            else if (wc == 1) bacc[0] = mfma32(af[1], ones, bacc[0]);
            else if (wc == 2) bacc[0] = mfma32(af[2], ones, bacc[0]);
            else bacc[0] = mfma32(af[3], ones, bacc[0]);
        }
    }
    dl_wait_vm(0);                                            // (the dummy loads behind the last slab)
    __builtin_amdgcn_s_barrier();

    // ---- epilogue: the job's 256 column sums of dY meet in LDS (row index = position among the job's A rows)
    float* bs = reinterpret_cast<float*>(lds + kDlLdsBias);
    const int arow0 = (kind == kDlDual || kind == kDlSplit ? 128 * half : 0) + 64 * wr;
    if (m == 0) {
        if (nshare == 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) bs[arow0 + 16 * (2 * wc + u) + 4 * g + r] = bacc[u][r];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) bs[arow0 + 16 * wc + 4 * g + r] = bacc[0][r];
        }
    }
    if (kind == kDlWide && threadIdx.x >= 128 && threadIdx.x < 256) bs[threadIdx.x] = 0.f;
    __syncthreads();
    const DlOut& O = J.out[half];
    if (O.b_scale) {                                          // B = scale * B + shift, applied to the accumulated product
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int j = 64 * wc + 16 * b + m;
            const float sc = O.b_scale[j], sh = O.b_shift[j];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f32x4 colsum = *reinterpret_cast<const f32x4*>(bs + arow0 + 16 * a + 4 * g);
                acc[a][b] = acc[a][b] * sc + colsum * sh;
            }
        }
    }
    f32x4* mine = reinterpret_cast<f32x4*>(pout) + threadIdx.x;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) mine[(a * 4 + b) * kDlThreads] = acc[a][b];
    if (threadIdx.x < 256) pout[kDlTileSlots * 4 + threadIdx.x] = bs[threadIdx.x];
}

// grid: njobs * G workgroups; block b = (chunk b / njobs, job b % njobs) so that all jobs advance together
__global__ __launch_bounds__(kDlThreads, 2) void dw_layer_kernel(DlArgs A) {
    __shared__ __attribute__((aligned(1024))) char lds[kDlLdsBytes];
    const int jb = blockIdx.x % A.njobs, chunk = blockIdx.x / A.njobs;
    float* pout = A.partial + (size_t)(jb * A.G + chunk) * kDlPartialFloats;
    if (A.job[jb].kind == kDlDual || A.job[jb].kind == kDlSplit) dl_job_body<4>(A.job[jb], A.n, chunk, A.G, pout, lds);
    else dl_job_body<3>(A.job[jb], A.n, chunk, A.G, pout, lds);
}

// one slot per thread: njobs * 8192 matrix slots (+ njobs * 256 bias rows)
__global__ __launch_bounds__(256) void dw_layer_reduce_kernel(DlReduce R) {
    dl_reduce(R, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

// A SPLIT job's reduction (one [128 x 128] product, up to 96 chunks x 2 halves per slot): with one slot per thread the sum
// was 6 dependent rounds of 32 loads in 4096 threads = 13 us on the step's tail (the VFE's layer-1 weight gradient is the
// last contraction of a step).  Two levels instead: a workgroup = 16 slots x 16 groups, thread (slot, k) sums chunks k, k + 16,
// ... (both halves) -- at most 12 loads, all in flight --, the groups meet in LDS and are added in group order: the result
// still does not depend on arrival order.  256 workgroups.
__global__ __launch_bounds__(256) void dw_split_reduce_kernel(DlReduce R) {
    __shared__ f32x4 red[16][16];
    const int sl = threadIdx.x & 15, k = threadIdx.x >> 4;
    const int sa = blockIdx.x * 16 + sl;                      // active slot 0..4095: (tile, thread of the first half)
    const int tile = sa >> 8, th = sa & 255;
    const int q = tile * kDlThreads + th;
    const float* p = R.partial + 4 * q;
    f32x4 v[2 * (kDlMaxChunks / 16)];
#pragma unroll
    for (int i = 0; i < kDlMaxChunks / 16; ++i) {
        const int u = k + 16 * i;
        v[2 * i] = v[2 * i + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (u < R.G) {
            v[2 * i] = *reinterpret_cast<const f32x4*>(p + (size_t)u * kDlPartialFloats);
            v[2 * i + 1] = *reinterpret_cast<const f32x4*>(p + (size_t)u * kDlPartialFloats + 4 * 256);
        }
    }
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2 * (kDlMaxChunks / 16); ++i) s += v[i];
    red[k][sl] = s;
    __syncthreads();
    if (k == 0) {
        const int a = tile >> 2, b = tile & 3, w = th >> 6, lane = th & 63, g = lane >> 4, o = lane & 15;
        int half, wr, wc;
        dl_wave_role(kDlSplit, w, &half, &wr, &wc);
        const DlOut& O = R.job[0].out[0];
        float* c0 = O.C + (int64_t)(64 * wr + 16 * a + 4 * g) * O.ldc + 64 * wc + 16 * b + o;
        f32x4 t = red[0][sl];
#pragma unroll
        for (int j = 1; j < 16; ++j) t += red[j][sl];
#pragma unroll
        for (int r = 0; r < 4; ++r) c0[(int64_t)r * O.ldc] += t[r];
    }
    if (R.job[0].out[0].dbias && blockIdx.x == 0 && threadIdx.x < 128) {      // column sums of the A operand
        const float* pb = R.partial + kDlTileSlots * 4 + threadIdx.x;
        float sb = 0.f;
        for (int u = 0; u < R.G; ++u) sb += pb[(size_t)u * kDlPartialFloats];
        R.job[0].out[0].dbias[threadIdx.x] += sb;
    }
}

}  // namespace geomae
