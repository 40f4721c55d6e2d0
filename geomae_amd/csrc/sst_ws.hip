// One launch per SST layer at the LARGE token sets (the decoders, config 3's encoder): the one-launch layer of sst_fused.hip
// re-shaped for windows of up to 144 positions and for workgroups that LOOP over bundles.
//
// Reference: EncoderLayer.forward / WindowAttention.forward (mmdet3d/models/sst/sst_basic_block.py:26-61, 85-102); the block
// loops of bb.py:236-303.
//
// STATUS (round 6): correct (tests/test_gpu_fused.py::test_weight_stationary_layer_matches_three_launch_form, five cases incl.
// 9-tile windows, fill rows, row maps) and OFF by default (GeomaeTuning.ws_layers = 0): alone it takes 58 us per layer at
// 22 k tokens against 54 us for the three-launch form and 205-215 against 160-170 us at 93 k; in the step dec_fwd goes
// 0.277 -> 0.305 ms at config 2 and 0.98 -> 1.17 ms at config 3 (docs/LAB_NOTES.md "Round 6" has the phase stamps).  Why: a
// layer is ~12.5 k cycles per 16-token tile and CU in EITHER form, 2.5x the VALU floor of the softmax / GELU / LayerNorm
// arithmetic; this form removes the qkv / attn round trips through HBM (2 of 4.4 KB per token and layer) but runs ONE
// workgroup per CU (136 KB of LDS, 225 VGPRs) where the three-launch kernels run two, and a 144-position window is one
// 110 k-cycle chain.  The name says what the first version was: it held the layer's 256 KB of weight fragments in registers
// across bundles (128 per lane) -- 13 k cycles per tile with ONE tile in flight per phase, because nothing was left for the
// tiles' chains to overlap in; this version fetches them per bundle, a phase or two ahead, like sst_fused.hip.
//
// sst_fused.hip runs one workgroup per bundle of <= 48 positions with straight-line bodies for 1-4 tiles; at 22 k - 93 k
// tokens it lost to the three-launch form (the 5-9-tile windows of the decoders -- 30 % of their tokens at config 2, 60 % at
// config 3 -- ran in a generic body that spilled).  This file is the other shape:
//  * at most one workgroup per CU (8 waves, wave w = head w = channel tile w, as in sst_fused.hip) that LOOPS over bundles;
//    the NEXT bundle's plan records are fetched into a second copy of the per-position tables while a bundle computes;
//  * a bundle is whole windows up to a soft cap, or ONE window of up to 144 positions (9 tiles): the in-projection and the
//    attention see the whole bundle (k and v^T of head w for up to 9 tiles: 36 registers; the queries go two tiles at a time
//    in a run-time loop with an ONLINE softmax over the key tiles, so no per-tile state grows with the window), the row-wise
//    rest -- out-projection, LayerNorm, FFN, LayerNorm -- runs in groups of four tiles whose residual rows are re-read from
//    L2 a phase ahead (they were read for the in-projection a few microseconds earlier) instead of being held in registers
//    across the attention;
//  * straight-line code per size class (2 / 4 / 6 / 9 tiles in the projections, groups of 2 / 4 in the rows): a tile past the
//    bundle carries token -1 (loads return zeros, stores are dropped).  With per-tile branches every loaded row stayed live
//    across a dozen basic blocks and the allocator spilled 60-90 registers behind s_waitcnt vmcnt(0);
//  * q / k / v / P never leave the registers, the attention output and the hidden activations cross waves through LDS; what is
//    stored is what the backward reads (the same saved tensors, token order, tile-blocked), so the two-launch backward follows.
#include "common.h"
#include "../../include/geomae_hip.h"

#include "sst_device.h"

#include "fused_device.h"
#include <type_traits>

namespace geomae {

constexpr int kWsThreads = 512;
constexpr int kWsTiles = kFMaxT / 16;                       // 9
constexpr int kWsGroup = 4;                                 // tiles of a row-phase group
// LDS: X (x, later y) | XP (x + pos; later the hidden rows of a group) | O | LayerNorm partials [2][64 rows][8 waves][2] |
// window start / end / token / in-window position per bundle position | the layer's fp32 parameter vectors
constexpr int kWLdsX = 0, kWLdsXP = kFMaxT * kFRow, kWLdsO = 2 * kFMaxT * kFRow, kWLdsRed = 3 * kFMaxT * kFRow;
// (the four per-position tables twice: the NEXT bundle's plan records land in the other copy while this bundle computes)
constexpr int kWLdsTab = kWLdsRed + 2 * 64 * 64, kWTabBytes = 4 * kFMaxT * 4;
constexpr int kWLdsPrm = kWLdsTab + 2 * kWTabBytes;
constexpr int kWLdsBytes = kWLdsPrm + kPFloats * 4;
static_assert(16 * kWsGroup * kFRowH <= kFMaxT * kFRow, "the hidden rows of a group must fit the XP region");
static_assert(kWLdsBytes <= 160 * 1024, "LDS of a CU");

// this wave's 16 channels (tile w, lane group g) of token tk's fp32 input row: the stack's tile-blocked x, or (first layer)
// the row-major source rows, gathered / filled (common.h SstInputMap)
template <bool MAPPED>
__device__ __forceinline__ f32x4 ws_load_x(const FusedFwd& A, __amdgpu_buffer_rsrc_t xres, int tk, int w, int g) {
    int off;
    if (MAPPED) {
        const bool in = tk >= 0 && tk < A.M.n_src;
        int srow = tk;
        if (A.M.rows && in) srow = A.M.rows[tk];
        off = in ? srow * 512 + 64 * w + 16 * g : kFOor;
    } else {
        off = tk >= 0 ? blk_off<4>(tk, 128, w, g) : kFOor;
    }
    return buf_load_f32x4(xres, off);
}

// phase stamps of every workgroup's FIRST bundle (tools/ws_phase_time.py with the timing build; a no-op in the product build)
#ifdef GEOMAE_PHASE_TIMING
#define WS_STAMP(i)                                                                                             \
    do {                                                                                                        \
        if (threadIdx.x == 0 && b == (int)blockIdx.x && blockIdx.x < GEOMAE_STAMP_BLOCKS)                       \
            geomae_stamps[blockIdx.x * GEOMAE_STAMP_SLOTS + (i)] = clock64();                                    \
    } while (0)
#define WS_STAMP_G(i) do { if (it0 == 0) WS_STAMP(i); } while (0)   /* (inside row_group) */
#else
#define WS_STAMP(i) do {} while (0)
#define WS_STAMP_G(i) do {} while (0)
#endif

// A value the optimiser may not look through: LDS addresses are built as (opaque per-lane base) + (compile-time constant), so
// that the constant rides in the instruction's 16-bit offset field.  Left to itself the compiler forms one base register per
// distinct (buffer, tile) constant above 64 KB -- dozens of loop-invariant address registers that it then spills.
__device__ __forceinline__ int opq(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ uint4 lds_b128_at(const char* lds, int off) { return *reinterpret_cast<const uint4*>(lds + off); }

// plan records of bundle positions [s0, s0 + T) into one copy of the per-position tables (threads 0..143)
__device__ __forceinline__ int4 ws_plan_issue(const FusedFwd& A, int s0, int T) {
    const int idx = threadIdx.x;
    int4 rec = make_int4(-1, 0, -1 - idx, 0);
    if (idx < T) rec = A.plan[s0 + idx];
    return rec;
}
__device__ __forceinline__ void ws_plan_commit(char* tab, const int4 rec) {
    if (threadIdx.x < kFMaxT) {
        int* p = reinterpret_cast<int*>(tab) + threadIdx.x;
        p[0] = rec.z; p[kFMaxT] = rec.w; p[2 * kFMaxT] = rec.x; p[3 * kFMaxT] = rec.y;
    }
}

// `dead_end`: token rows below it are DEAD in this layer's output (the top layer of a decoder stack, common.h
// set_first_live_row, in the 64-row granularity of the three-launch form): their out-projection / FFN results are not stored
// MAPPED: the stack's first layer (row-major source rows, row map, fill row: common.h SstInputMap) -- compiled apart, so that
// the other layers' loads carry no conditional side loads (behind one the compiler waits with vmcnt(0): for the weights too)
template <bool MAPPED>
// `min_tiles`: bundles of fewer tiles are left alone (5: the second launch of geomae_sst_layer_forward, whose first kernel ran the
// bundles of 1-4 tiles with its straight-line bodies)
__global__ __launch_bounds__(kWsThreads, 2) void sst_layer_fwd_ws_kernel(FusedFwd A, int dead_end, int min_tiles) {
    __shared__ __attribute__((aligned(16))) char lds[kWLdsBytes];
    const int NB = A.num_bundles[0];
    if ((int)blockIdx.x >= NB) return;                                // (workgroup-uniform)
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t = lane & 15, g = lane >> 4;
    // the layer's parameter vectors in LDS: this lane's four channels of tile w (of tile 2w for the hidden layer's bias); the
    // offset is made opaque where it is used, so that the loads stay inside their phase (hoisted out of the bundle loop
    // they are 48 loop-invariant registers)
#define PRM4(k) load_f4(reinterpret_cast<const float*>(lds + opq(kWLdsPrm + 64 * w + 16 * g)) + (k))
#define PRM4H(k) load_f4(reinterpret_cast<const float*>(lds + opq(kWLdsPrm + 128 * w + 16 * g)) + (k))
    const bool save = A.qkv != nullptr;
    const LayerW& W = A.W;
    const int perm_b = 2 * (32 * (w >> 1) + 8 * g + 4 * (w & 1));
    // per-lane LDS byte offsets (opq above): its 16-byte piece of row t as a B operand (r), the 8 bytes of the channel tile it
    // produces (w); 128-wide rows X / XP / O, the 256-wide hidden rows H (over XP), the LayerNorm partials
    const int bXr = opq(kWLdsX + t * kFRow + 16 * g), bXPr = opq(kWLdsXP + t * kFRow + 16 * g), bOr = opq(kWLdsO + t * kFRow + 16 * g);
    const int bXw = opq(kWLdsX + t * kFRow + perm_b), bXPw = opq(kWLdsXP + t * kFRow + perm_b), bOw = opq(kWLdsO + t * kFRow + perm_b);
    const int bHr = opq(kWLdsXP + t * kFRowH + 16 * g), bHw = opq(kWLdsXP + t * kFRowH + 2 * (32 * w + 8 * g));
    const int bRed = opq(kWLdsRed + t * 64);

    // ---- the first bundle's plan records and the layer's parameter vectors: once per workgroup
    int s0 = A.bun_tok[blockIdx.x];
    int T = A.bun_tok[blockIdx.x + 1] - s0;
    if (T > kFMaxT) T = kFMaxT;              // (the host refuses layouts of windows above 144 positions)
    {
        const int4 rec0 = ws_plan_issue(A, s0, T);
        const f32x4 prm_r = params_issue(W);
        if (threadIdx.x < kPFloats / 4) *reinterpret_cast<f32x4*>(lds + kWLdsPrm + 16 * threadIdx.x) = prm_r;
        ws_plan_commit(lds + kWLdsTab, rec0);
    }
    const bool has_fill = MAPPED && A.M.fill;
    __syncthreads();                                                  // tables of the first bundle + parameters in LDS

    int par = 0;
    for (int b = blockIdx.x; b < NB; b += gridDim.x, par ^= 1) {
        const int nt = (T + 15) >> 4;
        // this bundle's tables: window start | end | token | in-window position per position (kFMaxT ints each)
        const int tb = kWLdsTab + par * kWTabBytes;
        const int tbT = opq(tb + 4 * t), tbG = opq(tb + 16 * g);
#define WL_AT(i) (*reinterpret_cast<const int*>(lds + tb + 4 * (i)))
#define WH_AT(i) (*reinterpret_cast<const int*>(lds + tb + 4 * kFMaxT + 4 * (i)))
#define WL_T(tile) (*reinterpret_cast<const int*>(lds + tbT + 64 * (tile)))
#define TOK_T(tile) (*reinterpret_cast<const int*>(lds + tbT + 8 * kFMaxT + 64 * (tile)))
#define POS_T(tile) (*reinterpret_cast<const int*>(lds + tbT + 12 * kFMaxT + 64 * (tile)))
#define WL_G4(tile) (*reinterpret_cast<const int4*>(lds + tbG + 64 * (tile)))
        // the NEXT bundle's plan records: in flight under phase A, into the other copy of the tables behind barrier (1)
        // (bundle bounds through the scalar cache: as a vector load its wait was vmcnt(0) -- every store of the previous bundle)
        const int bn = __builtin_amdgcn_readfirstlane(b + (int)gridDim.x);
        int s0n = 0, Tn = 0;
        if (bn < NB) {
            const int32_t* __restrict__ bt = reinterpret_cast<const int32_t*>(uniform_ptr(A.bun_tok));
            s0n = __builtin_amdgcn_readfirstlane(bt[bn]);
            Tn = __builtin_amdgcn_readfirstlane(bt[bn + 1]) - s0n;
            if (Tn > kFMaxT) Tn = kFMaxT;
        }
        const int4 recn = ws_plan_issue(A, s0n, Tn);
        if (nt < min_tiles) {                                        // (uniform) not this launch's bundle: only hand the tables on
            ws_plan_commit(lds + kWLdsTab + (par ^ 1) * kWTabBytes, recn);
            __syncthreads();
            s0 = s0n; T = Tn;
            continue;
        }
        WS_STAMP(0);

        // The weights are fetched per bundle, a phase or two ahead of their first use (fragment-major, L2): a bundle is up to
        // nine tiles, which is what amortises them -- held in registers across bundles they left the tiles' chains no room to
        // overlap (measured: 13 k cycles per tile with the weights stationary and one tile in flight per phase).
        uint4 wq[4], wk[4], wv[4], wo[4], w1a[4], w1b[4], w2[8];

        // ---- phase A: this wave's channel tile of every row: bf16 x and x + pos into LDS.  Straight-line chunks of NC tiles picked by
        // the bundle's size (a tile past the bundle has token -1 in the tables: its loads return zeros, its stores are dropped) --
        // per-tile branches left every loaded row live across a dozen basic blocks, and the allocator spilled them behind
        // s_waitcnt vmcnt(0), one load at a time.
        {
            const __amdgpu_buffer_rsrc_t xres = whole_rsrc(MAPPED ? A.M.src : A.x), pres = whole_rsrc(A.pos_table);
            const __amdgpu_buffer_rsrc_t xb_r = saved_rsrc(A.xb), xp_r = saved_rsrc(A.xp);
            auto rows_in = [&](auto nc, const int it_lo, const bool with_weights) {
                constexpr int NC = decltype(nc)::value;
                f32x4 xr[NC], pv[NC];
                int tkc[NC], psc[NC];
#pragma unroll
                for (int k = 0; k < NC; ++k) { tkc[k] = TOK_T(it_lo + k); psc[k] = POS_T(it_lo + k); }
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    xr[k] = ws_load_x<MAPPED>(A, xres, tkc[k], w, g);
                    pv[k] = buf_load_f32x4(pres, tkc[k] >= 0 ? psc[k] * 512 + 64 * w + 16 * g : kFOor);
                }
                if (with_weights) {                                   // the in-projection's weights BEHIND the first rows
                    __builtin_amdgcn_sched_barrier(0);
                    load_wfrag<128>(W.frag + kOffWqkv, 8 + w, lane, wk);
                    load_wfrag<128>(W.frag + kOffWqkv, 16 + w, lane, wv);
                    load_wfrag<128>(W.frag + kOffWqkv, w, lane, wq);
                }
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    const int tk = tkc[k];
                    if (has_fill && tk >= A.M.n_src) xr[k] = load_f4(A.M.fill + 16 * w + 4 * g);
                    const uint2 xb = pack4(xr[k]), xpb = pack4(xr[k] + pv[k]);
                    *reinterpret_cast<uint2*>(lds + bXw + 16 * (it_lo + k) * kFRow) = xb;
                    *reinterpret_cast<uint2*>(lds + bXPw + 16 * (it_lo + k) * kFRow) = xpb;
                    if (save) {
                        const int o2 = tk >= 0 ? blk_off<2>(tk, 128, w, g) : kFOor;
                        if (A.xb) buf_store_b64(xb_r, o2, xb);
                        buf_store_b64(xp_r, o2, xpb);
                    }
                }
            };
            using std::integral_constant;
            // (all rows of the bundle in flight together: with the weights fetched per bundle phase A has the registers)
            if (nt <= 2) rows_in(integral_constant<int, 2>{}, 0, true);
            else if (nt <= 4) rows_in(integral_constant<int, 4>{}, 0, true);
            else if (nt <= 6) rows_in(integral_constant<int, 6>{}, 0, true);
            else rows_in(integral_constant<int, 9>{}, 0, true);
        }
        WS_STAMP(1);
        __syncthreads();                                                                       // (1) rows in LDS
        WS_STAMP(2);
        ws_plan_commit(lds + kWLdsTab + (par ^ 1) * kWTabBytes, recn);   // (read behind the barriers that follow)

        // ---- phase B: k, v (for the backward), v^T of head w for every tile of the bundle (straight-line for 2 / 4 / 6 / 9 tiles)
        uint2 kf[kWsTiles], vtf[kWsTiles];
        {
            const __amdgpu_buffer_rsrc_t qkv_r = saved_rsrc(A.qkv);
            const f32x4 bk = PRM4(kPBq + 128), bv = PRM4(kPBq + 256);
            const float bvn = *(reinterpret_cast<const float*>(lds + opq(kWLdsPrm + 64 * w + 4 * t)) + kPBq + 256);
            auto project = [&](auto ntc) {
                constexpr int NTC = decltype(ntc)::value;
#pragma unroll
                for (int it = 0; it < kWsTiles; ++it) {
                    kf[it] = make_uint2(0u, 0u); vtf[it] = make_uint2(0u, 0u);
                    if (it < NTC) {
                        const char* xrow = lds + bXr + 16 * it * kFRow;
                        const char* xprow = lds + bXPr + 16 * it * kFRow;
                        f32x4 ak = bk, av = bv, avt = {bvn, bvn, bvn, bvn};
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            const uint4 bxp = lds_b128(xprow + 64 * kk), bx = lds_b128(xrow + 64 * kk);
                            ak = mfma32(wk[kk], bxp, ak);
                            avt = mfma32(bx, wv[kk], avt);
                            av = mfma32(wv[kk], bx, av);
                        }
                        kf[it] = pack4(ak);
                        vtf[it] = pack4(avt);
                        if (save) {
                            const int tk = TOK_T(it);
                            const int o2 = tk >= 0 ? blk_off<2>(tk, 384, w, g) : kFOor;
                            buf_store_b64(qkv_r, o2 + 8 * 512, kf[it]);
                            buf_store_b64(qkv_r, o2 + 16 * 512, pack4(av));
                        }
                    }
                }
            };
            using std::integral_constant;
            if (nt <= 2) project(integral_constant<int, 2>{});
            else if (nt <= 4) project(integral_constant<int, 4>{});
            else if (nt <= 6) project(integral_constant<int, 6>{});
            else project(integral_constant<int, 9>{});
        }
        WS_STAMP(3);
        load_wfrag<128>(W.frag + kOffWo, w, lane, wo);                // first use: phase D
        load_wfrag<128>(W.frag + kOffW1, 2 * w, lane, w1a);           // first use: phase E
        load_wfrag<128>(W.frag + kOffW1, 2 * w + 1, lane, w1b);
        // the fp32 residual rows of the FIRST row group (L2: read for phase A a few microseconds ago): in flight under the attention
        f32x4 xg[kWsGroup];
        {
            const __amdgpu_buffer_rsrc_t xres = whole_rsrc(MAPPED ? A.M.src : A.x);
#pragma unroll
            for (int lt = 0; lt < kWsGroup; ++lt) xg[lt] = ws_load_x<MAPPED>(A, xres, TOK_T(lt), w, g);
        }

        // ---- phase C: the queries two tiles at a time (two independent chains per wave): q of head w, then an ONLINE softmax
        // over the key tiles the pair's windows span -- the state is (m, l, o) per query tile whatever the window's size
        {
            const __amdgpu_buffer_rsrc_t qkv_r = saved_rsrc(A.qkv), attn_r = saved_rsrc(A.attn), lse_r = saved_rsrc(A.lse);
            const f32x4 bq = PRM4(kPBq);
            // scores in the log2 domain: exp(s / 4 - m) = exp2(s c - m'), c = log2(e) / sqrt(16)
            const float scale = 0.25f * 1.4426950408889634f;
            for (int it = 0; it < nt; it += 2) {
                const bool two = it + 1 < nt;                        // (uniform)
                const int first = 16 * it;
                const int tbGi = opq(tbG);                           // (per pair: the key tiles' window ids are NOT hoisted -- 36 registers)
                f32x4 aq0 = bq, aq1 = bq;
                {
                    const char* xp0 = lds + bXPr + first * kFRow;
                    const char* xp1 = xp0 + (two ? 16 * kFRow : 0);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        aq0 = mfma32(wq[kk], lds_b128(xp0 + 64 * kk), aq0);
                        aq1 = mfma32(wq[kk], lds_b128(xp1 + 64 * kk), aq1);
                    }
                }
                const uint2 qf0 = pack4(aq0), qf1 = pack4(aq1);
                const int tk0 = TOK_T(it), tk1 = two ? TOK_T(it + 1) : -1;
                if (save) {
                    buf_store_b64(qkv_r, tk0 >= 0 ? blk_off<2>(tk0, 384, w, g) : kFOor, qf0);
                    buf_store_b64(qkv_r, tk1 >= 0 ? blk_off<2>(tk1, 384, w, g) : kFOor, qf1);
                }
                const int lastq = first + (two ? 31 : 15);
                const int last = lastq < T ? lastq : T - 1;
                const int jlo = __builtin_amdgcn_readfirstlane((WL_AT(first) - s0) >> 4);
                const int jhi = __builtin_amdgcn_readfirstlane((WH_AT(last) - 1 - s0) >> 4);
                const int wq0 = WL_T(it), wq1 = two ? WL_T(it + 1) : -0x40000000;
                float m0 = -INFINITY, l0 = 0.f, m1 = -INFINITY, l1 = 0.f;
                f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0;
#pragma unroll
                for (int j = 0; j < kWsTiles; ++j) {
                    if (j >= jlo && j <= jhi) {
                        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                        f32x4 sa = mfma16(kf[j], qf0, z4);            // S^T: rows = keys 4g + r of tile j, column = query t
                        f32x4 sb = mfma16(kf[j], qf1, z4);
                        const int4 W4 = *reinterpret_cast<const int4*>(lds + tbGi + 64 * j);
                        const int Wr[4] = {W4.x, W4.y, W4.z, W4.w};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            sa[r] = (Wr[r] == wq0) ? sa[r] : -INFINITY;
                            sb[r] = (Wr[r] == wq1) ? sb[r] : -INFINITY;
                        }
                        // (the maxima of the RAW scores, scaled once per tile; scale and subtraction of the exponent in one fma)
                        const float mxa = rows4_max(fmaxf(fmaxf(sa[0], sa[1]), fmaxf(sa[2], sa[3]))) * scale;
                        const float mxb = rows4_max(fmaxf(fmaxf(sb[0], sb[1]), fmaxf(sb[2], sb[3]))) * scale;
                        const float na = fmaxf(m0, mxa), nb = fmaxf(m1, mxb);
                        const float ua = na == -INFINITY ? 0.f : na, ub = nb == -INFINITY ? 0.f : nb;   // (nothing of its window yet)
                        const float ala = __builtin_amdgcn_exp2f(m0 - ua), alb = __builtin_amdgcn_exp2f(m1 - ub);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            sa[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sa[r], scale, -ua));
                            sb[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sb[r], scale, -ub));
                        }
                        l0 = l0 * ala + ((sa[0] + sa[1]) + (sa[2] + sa[3]));
                        l1 = l1 * alb + ((sb[0] + sb[1]) + (sb[2] + sb[3]));
                        o0 *= ala; o1 *= alb;
                        o0 = mfma16(vtf[j], pack4(sa), o0);           // O^T = V^T P^T: rows = channels 4g + r of head w, column = query t
                        o1 = mfma16(vtf[j], pack4(sb), o1);
                        m0 = na; m1 = nb;
                    }
                }
                l0 = rows4_sum(l0); l1 = rows4_sum(l1);
                o0 *= 1.0f / l0; o1 *= 1.0f / l1;
                const uint2 ob0 = pack4(o0), ob1 = pack4(o1);
                *reinterpret_cast<uint2*>(lds + bOw + first * kFRow) = ob0;
                if (two) *reinterpret_cast<uint2*>(lds + bOw + (first + 16) * kFRow) = ob1;
                if (save) {
                    const float ln2 = 0.6931471805599453f;             // the saved log-sum-exp is a natural logarithm
                    buf_store_b64(attn_r, tk0 >= 0 ? blk_off<2>(tk0, 128, w, g) : kFOor, ob0);
                    buf_store_b64(attn_r, tk1 >= 0 ? blk_off<2>(tk1, 128, w, g) : kFOor, ob1);
                    if (g == 0) {
                        buf_store_f32(lse_r, tk0 >= 0 ? (tk0 * 8 + w) * 4 : kFOor, (m0 + __builtin_amdgcn_logf(l0)) * ln2);
                        buf_store_f32(lse_r, tk1 >= 0 ? (tk1 * 8 + w) * 4 : kFOor, (m1 + __builtin_amdgcn_logf(l1)) * ln2);
                    }
                }
            }
        }
        load_wfrag<256>(W.frag + kOffW2, w, lane, w2);                // first use: phase F
        WS_STAMP(4);
        __syncthreads();                                                                       // (2) attention output in LDS
        WS_STAMP(5);

        // ---- the row-wise rest in groups of four (a last group of one or two: two) tiles, straight-line: a tile past the bundle
        // has token -1 (loads return zeros, global stores are dropped); only its LDS row writes are branched around
        // (the residual rows of a group arrive in xg; those of the NEXT group are requested behind barrier (4) and handed back in xg)
        auto row_group = [&](auto gnc, const int it0, const int gi) {
            constexpr int GN = decltype(gnc)::value;
            const int rb = bRed + (gi & 1) * (64 * 64);                       // this group's copy of the LayerNorm partials
#define RED_ROW(lt) reinterpret_cast<float*>(lds + rb + 1024 * (lt))
#define FOR_GROUP(lt) _Pragma("unroll") for (int lt = 0; lt < GN; ++lt)
            int tk[GN];
            f32x4 xr[GN];
            FOR_GROUP(lt) {
                tk[lt] = it0 + lt < kWsTiles ? TOK_T(it0 + lt) : -1;
                xr[lt] = xg[lt];
            }
            // ---- phase D: u = x + attn Wo^T + bo (channel tile w), LayerNorm-1 partials
            {
                const f32x4 bo = PRM4(kPBo);
                FOR_GROUP(lt) {
                    const int itc = it0 + lt < kWsTiles ? it0 + lt : kWsTiles - 1;
                    const char* orow = lds + bOr + 16 * itc * kFRow;
                    f32x4 au = bo;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) au = mfma32(wo[kk], lds_b128(orow + 64 * kk), au);
                    if (has_fill && tk[lt] >= A.M.n_src) xr[lt] = load_f4(A.M.fill + 16 * w + 4 * g);
                    xr[lt] += au;
                    ln_partial(xr[lt], RED_ROW(lt) + 2 * w, g);
                }
            }
            WS_STAMP_G(6);
            __syncthreads();                                                                   // (3) LayerNorm-1 statistics
            WS_STAMP_G(7);
            {
                const __amdgpu_buffer_rsrc_t xh1_r = saved_rsrc(A.xh1), rstd_r = saved_rsrc(A.rstd);
                const f32x4 g1 = PRM4(kPG1), be1 = PRM4(kPBe1);
                FOR_GROUP(lt) {
                    float mean, rstd;
                    ln_merge(RED_ROW(lt), A.eps, &mean, &rstd);
                    const f32x4 xh = (xr[lt] - mean) * rstd;
                    if (save) {
                        const bool st = tk[lt] >= dead_end;
                        buf_store_b64(xh1_r, st ? blk_off<2>(tk[lt], 128, w, g) : kFOor, pack4(xh));
                        if (w == 0 && g == 0) buf_store_f32(rstd_r, st ? tk[lt] * 8 : kFOor, rstd);
                    }
                    xr[lt] = xh * g1 + be1;                              // y: the FFN's input and its residual
                    if (it0 + lt < nt) *reinterpret_cast<uint2*>(lds + bXw + 16 * (it0 + lt) * kFRow) = pack4(xr[lt]);
                }
            }
            WS_STAMP_G(8);
            __syncthreads();                                                                   // (4) y in LDS
            WS_STAMP_G(9);
            if (it0 + kWsGroup < nt) {                               // the next group's residual rows: in flight under phases E, F
                const __amdgpu_buffer_rsrc_t xres = whole_rsrc(MAPPED ? A.M.src : A.x);
#pragma unroll
                for (int lt = 0; lt < kWsGroup; ++lt)
                    xg[lt] = ws_load_x<MAPPED>(A, xres, it0 + kWsGroup + lt < kWsTiles ? TOK_T(it0 + kWsGroup + lt) : -1, w, g);
            }
            // ---- phase E: h = gelu(y W1^T + b1), channel tiles 2w, 2w + 1
            {
                const __amdgpu_buffer_rsrc_t hp_r = saved_rsrc(A.hp);
                const f32x4 b1a = PRM4H(kPB1), b1b = PRM4H(kPB1 + 16);
                FOR_GROUP(lt) {
                    const int itc = it0 + lt < kWsTiles ? it0 + lt : kWsTiles - 1;
                    const char* yrow = lds + bXr + 16 * itc * kFRow;
                    f32x4 ha = b1a, hb = b1b;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const uint4 by = lds_b128(yrow + 64 * kk);
                        ha = mfma32(w1a[kk], by, ha);
                        hb = mfma32(w1b[kk], by, hb);
                    }
                    if (save) {
                        const int o2 = tk[lt] >= dead_end ? blk_off<2>(tk[lt], 256, 2 * w, g) : kFOor;
                        buf_store_b64(hp_r, o2, pack4(ha));
                        buf_store_b64(hp_r, o2 + 512, pack4(hb));
                    }
                    const uint2 pa = pack4(gelu4(ha)), pb = pack4(gelu4(hb));
                    *reinterpret_cast<uint4*>(lds + bHw + 16 * lt * kFRowH) = make_uint4(pa.x, pa.y, pb.x, pb.y);
                }
            }
            WS_STAMP_G(10);
            __syncthreads();                                                                   // (5) gelu output in LDS
            WS_STAMP_G(11);
            // ---- phase F: v = y + h W2^T + b2, LayerNorm-2 partials
            {
                const f32x4 b2 = PRM4(kPB2);
                FOR_GROUP(lt) {
                    const char* hrow = lds + bHr + 16 * lt * kFRowH;
                    f32x4 a = b2;
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) a = mfma32(w2[kk], lds_b128(hrow + 64 * kk), a);
                    xr[lt] += a;
                    ln_partial(xr[lt], RED_ROW(lt) + 2 * w, g);
                }
            }
            WS_STAMP_G(12);
            __syncthreads();                                                                   // (6) LayerNorm-2 statistics
            WS_STAMP_G(13);
            {
                const __amdgpu_buffer_rsrc_t xh2_r = saved_rsrc(A.xh2), rstd_r = saved_rsrc(A.rstd), z_r = whole_rsrc(A.z);
                const f32x4 g2 = PRM4(kPG2), be2 = PRM4(kPBe2);
                FOR_GROUP(lt) {
                    float mean, rstd;
                    ln_merge(RED_ROW(lt), A.eps, &mean, &rstd);
                    const f32x4 xh = (xr[lt] - mean) * rstd;
                    const int tkk = tk[lt];
                    const bool st = tkk >= dead_end && tkk >= 0;
                    if (save) {
                        buf_store_b64(xh2_r, st ? blk_off<2>(tkk, 128, w, g) : kFOor, pack4(xh));
                        if (w == 0 && g == 0) buf_store_f32(rstd_r, st ? tkk * 8 + 4 : kFOor, rstd);
                    }
                    const f32x4 zz = xh * g2 + be2;
                    const int zo = !st ? kFOor : (A.z_blocked ? blk_off<4>(tkk, 128, w, g) : tkk * 512 + 64 * w + 16 * g);
                    buf_store_f32x4(z_r, zo, zz);
                }
            }
            WS_STAMP_G(14);
#undef FOR_GROUP
#undef RED_ROW
        };
        {
            int gi = 0;
            for (int it0 = 0; it0 < nt; it0 += kWsGroup, ++gi) {
                if (nt - it0 <= 2) row_group(std::integral_constant<int, 2>{}, it0, gi);
                else row_group(std::integral_constant<int, kWsGroup>{}, it0, gi);
            }
        }
        WS_STAMP(15);
#ifdef GEOMAE_PHASE_TIMING
        if (threadIdx.x == 0 && b == (int)blockIdx.x && blockIdx.x < GEOMAE_STAMP_BLOCKS) geomae_stamps[blockIdx.x * GEOMAE_STAMP_SLOTS + 16] = T;
#endif
        s0 = s0n; T = Tn;
#undef WL_AT
#undef WH_AT
#undef WL_T
#undef TOK_T
#undef POS_T
#undef WL_G4
    }
#undef PRM4
#undef PRM4H
}

}  // namespace geomae

using namespace geomae;

#ifdef GEOMAE_PHASE_TIMING
extern "C" int geomae_debug_read_ws_stamps(unsigned long long* host, int clear) {
    hipDeviceSynchronize();
    if (host) hipMemcpyFromSymbol(host, HIP_SYMBOL(geomae_stamps), sizeof(unsigned long long) * GEOMAE_STAMP_BLOCKS * GEOMAE_STAMP_SLOTS);
    if (clear) {
        static unsigned long long zeros[GEOMAE_STAMP_BLOCKS * GEOMAE_STAMP_SLOTS];
        hipMemcpyToSymbol(HIP_SYMBOL(geomae_stamps), zeros, sizeof(zeros));
    }
    return 0;
}
#endif

// The forward of one layer as ONE weight-stationary launch (sst_layer_fwd_ws_kernel).  Internal: geomae_sst_stack_forward
// calls it per layer for the token sets above the one-bundle-per-workgroup form's range (sst_stack.hip).  Same arguments and
// saved tensors as geomae_sst_layer_forward; `dead_rows`: common.h set_first_live_row of the stack's LAST layer; `min_tiles`:
// 1, or 5 as the second launch of geomae_sst_layer_forward (the bundles its straight-line bodies leave).
int geomae::sst_layer_forward_ws(const float* x, const SstInputMap& M, int num_tokens, const GeomaeSstLayerWeights* w,
                                 const GeomaeSstStackLayout* layout, const float* pos_table, float* z, bool z_blocked,
                                 void* qkv, void* attn, float* lse, void* xh1, void* xh2, void* hp, float* rstd, void* xb,
                                 void* xp, int dead_rows, int max_workgroups, int min_tiles, hipStream_t stream) {
    GEOMAE_REQUIRE(w && w->frag_p && layout && layout->fbun_tok && layout->pos_info && layout->num_fbundles && layout->max_bundles >= 1,
                   "sst_layer_forward_ws: plan / fragment-major weights missing");
    GEOMAE_REQUIRE(num_tokens > 0 && num_tokens <= 2700000, "sst_layer_forward_ws: token count out of range");
    FusedFwd A;
    A.x = x; A.M = M; A.bun_tok = layout->fbun_tok; A.plan = (const int4*)layout->pos_info; A.num_bundles = layout->num_fbundles;
    A.pos_table = pos_table; A.W = to_layer(w); A.n = num_tokens; A.eps = w->ln_eps; A.z = z; A.z_blocked = z_blocked ? 1 : 0;
    A.qkv = (bf16_t*)qkv; A.attn = (bf16_t*)attn; A.xh1 = (bf16_t*)xh1; A.xh2 = (bf16_t*)xh2; A.hp = (bf16_t*)hp;
    A.xb = (bf16_t*)xb; A.xp = (bf16_t*)xp; A.lse = lse; A.rstd = rstd;
    int grid = layout->max_bundles < max_workgroups ? layout->max_bundles : max_workgroups;
    if (grid < 1) grid = 1;
    if (M.src) hipLaunchKernelGGL(sst_layer_fwd_ws_kernel<true>, dim3(grid), dim3(kWsThreads), 0, stream, A, (dead_rows / 64) * 64, min_tiles);
    else hipLaunchKernelGGL(sst_layer_fwd_ws_kernel<false>, dim3(grid), dim3(kWsThreads), 0, stream, A, (dead_rows / 64) * 64, min_tiles);
    return check_launch("sst_layer_fwd_ws_kernel");
}
