// One launch per SST encoder layer (forward): in-projection with the positional term -> window attention -> out-projection
// + residual + LayerNorm + FFN(GELU) + residual + LayerNorm, for one BUNDLE of windows per workgroup.
//
// Reference: EncoderLayer.forward / WindowAttention.forward (mmdet3d/models/sst/sst_basic_block.py:26-61, 85-102), which
// pads every window to a bucket size and runs nn.MultiheadAttention + ~25 ATen kernels per layer.  The unfused build of this
// repository (sst_layer.hip + window.hip) ran a layer as three launches -- q/k/v, the attention output and the residual
// stream making a round trip through HBM between them -- and the attention launch, with d_head = 16, was a chain of
// dependent gathers around a handful of tiny MFMAs (1 % of the MFMA roof for three rounds).
//
// Design (gfx950; 512 threads = 8 waves = 8 heads per workgroup, one workgroup per bundle of <= 144 tokens):
//  * WEIGHT-STATIONARY, N-SPLIT GEMMs.  Wave w owns a slice of every GEMM's OUTPUT channels for ALL token tiles of the
//    bundle: head w's 16 q / k / v channels, channel tile w of the out-projection and of FFN2, channel tiles 2w, 2w+1 of
//    FFN1.  Its weight fragments (A operands of the transposed product Y^T = W X^T) are read from L2 ONCE into registers;
//    the activations (B operands) come from an LDS copy of the bundle's rows that all waves share.  Per-wave work is
//    (32 + attention) MFMAs per 16-token tile whatever the bundle size: the dependent chain of a wave scales with the
//    bundle, not with a fixed 16-token x 256-MFMA tile as in the one-wave-per-tile kernels.
//  * ATTENTION IN REGISTERS.  With the transposed product the q / k accumulators of head w (lane = token, 4 channels per
//    lane group) ARE the B / A operands of S^T = K Q^T (v_mfma_f32_16x16x16_bf16); V is projected a second time in the
//    untransposed orientation (4 MFMAs per tile), whose accumulator is the A operand of O^T = V^T P^T; and O^T comes out
//    in the T-layout again.  q, k, v and P never touch LDS; the softmax row of a query lives in one lane (x 4 lane groups).
//  * the only exchanges between waves: the bf16 rows x / x + pos (in), the attention output, y = LN1(..), gelu(..) -- each
//    one LDS write + barrier + fragment reads -- and the two LayerNorm statistics (per-wave (mean, M2) of its 16 channels,
//    merged Welford-style: one barrier each).  Six barriers per layer.
//  * token rows are addressed through the build's plan records (window.hip pos_info: token, in-window position, window
//    start, window end per bundle position); every tensor keeps the TOKEN-ORDER tile-blocked layout of the stacks, so the
//    saved activations are exactly what the (unfused) backward kernels read.
#include "common.h"
#include "../../include/geomae_hip.h"

#include "sst_device.h"

#include "fused_device.h"

namespace geomae {

// NT: tiles the body is unrolled for.  EXACT: the bundle has exactly NT tiles -- straight-line code (the scheduler overlaps
// the tiles' chains), all tile pairs of the attention computed (the mask does the block-diagonal); the only form instantiated
// since round 6 (bundles of more than four tiles: sst_ws.hip).  !EXACT: nt <= NT tiles
// behind scalar branches, key-tile ranges.  Weight fragments are fetched two phases ahead of their first use.
// A bundle of more than four tiles that NO launch will run -- the forward's second kernel was skipped, or the one-launch
// backward met one -- is a broken promise of the caller (common.h set_fused_big_layouts: "this layout holds none"): the
// tokens would silently keep stale values.  The kernels count such bundles here; geomae_sst_fused_dropped_bundles reads the
// count (tests assert 0; round 5 shipped NaN losses for an hour through exactly this hole).
static __device__ unsigned int g_fused_dropped = 0;

// Q0, NQ (round 6): the QUERY tiles [Q0, Q0 + NQ) this work item owns.  Rows, k and v^T are made for all NT tiles of the bundle
// (the attention needs every key of a window), q / attention / the row-wise rest and every store only for the owned tiles: a
// bundle of three or four tiles is two work items in two workgroups (window.hip item_pack), each a chain of about two tiles.
template <int NT, bool EXACT, int Q0 = 0, int NQ = NT>
__device__ __forceinline__ void fused_fwd_body(const FusedFwd& A, const int s0, const int T, const int nt_in, char* lds) {
    static_assert(Q0 >= 0 && NQ >= 1 && Q0 + NQ <= NT, "query tiles inside the bundle");
    const int nt = EXACT ? NT : nt_in;
#define FOR_TILES(it) _Pragma("unroll") for (int it = 0; it < NT; ++it) if (EXACT || it < nt)
#define FOR_OWN(it) _Pragma("unroll") for (int it = Q0; it < Q0 + NQ; ++it) if (EXACT || it < nt)
#define OWN(it) ((it) >= Q0 && (it) < Q0 + NQ)
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // head = channel tile of this wave
    const int t = lane & 15, g = lane >> 4;
    char* X = lds + kFLdsX;
    char* XP = lds + kFLdsXP;
    char* O = lds + kFLdsO;
    char* H = lds + kFLdsXP;
    float* red = reinterpret_cast<float*>(lds + kFLdsRed);
    int* wl = reinterpret_cast<int*>(lds + kFLdsWl);
    int* wh = reinterpret_cast<int*>(lds + kFLdsWh);
    const float* prm = reinterpret_cast<const float*>(lds + kFLdsPrm);
    const bool save = A.qkv != nullptr;
    const LayerW& W = A.W;
    // where this wave's 16 channels sit in a K-permuted bf16 row (sst_device.h kperm): 8 bytes per lane group
    const int perm_b = 2 * (32 * (w >> 1) + 8 * g + 4 * (w & 1));
    FUSED_STAMP(0);

    // ---- the plan records first (the head of the longest dependent chain), then the weights
    int4 rec[NT];
    FOR_TILES(it) {
        const int idx = 16 * it + t;
        rec[it] = make_int4(-1, 0, -1 - idx, 0);
        if (idx < T) rec[it] = A.plan[s0 + idx];
#ifdef FUSED_ABL_NO_PLAN             // (timing ablation: what the plan hop costs the chain -- records from arithmetic)
        if (idx < T) rec[it] = make_int4(s0 + idx, idx, s0, s0 + T);
#endif
    }
    const f32x4 prm_r = params_issue(W);
    uint4 wq[4], wk[4], wv[4], wo[4], w1a[4], w1b[4], w2[8];

    // ---- phase A: this wave's channel tile of every row: fp32 (kept for the residual), bf16 x and x + pos into LDS
    int tok[NT];
    f32x4 xr[NT];
    {
        const __amdgpu_buffer_rsrc_t xres = whole_rsrc(A.M.src ? A.M.src : A.x);
        const __amdgpu_buffer_rsrc_t pres = whole_rsrc(A.pos_table);
        const __amdgpu_buffer_rsrc_t xb_r = saved_rsrc(A.xb), xp_r = saved_rsrc(A.xp);
        f32x4 pv[NT];
        FOR_TILES(it) {
            const int tk = rec[it].x;
            tok[it] = tk;
            int off;
            if (A.M.src) {                    // the stack's input conversion: row-major rows, gathered / filled
                int srow = tk;
                if (A.M.rows && tk >= 0 && tk < A.M.n_src) srow = A.M.rows[tk];
                off = (tk >= 0 && tk < A.M.n_src) ? srow * 512 + 64 * w + 16 * g : kFOor;
            } else {
                off = tk >= 0 ? blk_off<4>(tk, 128, w, g) : kFOor;
            }
            xr[it] = buf_load_f32x4(xres, off);
            pv[it] = buf_load_f32x4(pres, tk >= 0 ? rec[it].y * 512 + 64 * w + 16 * g : kFOor);
#ifdef FUSED_ABL_NO_ROWS             // (timing ablation: the row hop)
            xr[it] = f32x4{0.01f * off, 0.5f, -0.5f, 0.25f}; pv[it] = f32x4{0.f, 0.1f, 0.2f, 0.3f};
#endif
        }
        // the in-projection's weights BEHIND the row loads: the rows (the longer dependent chain: plan -> row) are not
        // queued behind 96 KB of fragments, and the fragments land under the LDS writes and the barrier
        __builtin_amdgcn_sched_barrier(0);                          // (the scheduler otherwise hoists them above the plan wait)
        load_wfrag<128>(W.frag + kOffWqkv, w, lane, wq);
        load_wfrag<128>(W.frag + kOffWqkv, 8 + w, lane, wk);
        load_wfrag<128>(W.frag + kOffWqkv, 16 + w, lane, wv);
        if (A.M.src && A.M.fill) {
            const f32x4 fill = load_f4(A.M.fill + 16 * w + 4 * g);
            FOR_TILES(it) if (tok[it] >= A.M.n_src) xr[it] = fill;
        }
        if (threadIdx.x < kPFloats / 4) *reinterpret_cast<f32x4*>(lds + kFLdsPrm + 16 * threadIdx.x) = prm_r;
        FOR_TILES(it) {
            const int idx = 16 * it + t;
            const uint2 xb = pack4(xr[it]), xpb = pack4(xr[it] + pv[it]);
            *reinterpret_cast<uint2*>(X + idx * kFRow + perm_b) = xb;
            *reinterpret_cast<uint2*>(XP + idx * kFRow + perm_b) = xpb;
            if (save && OWN(it)) {
                const int o2 = tok[it] >= 0 ? blk_off<2>(tok[it], 128, w, g) : kFOor;
                if (A.xb) buf_store_b64(xb_r, o2, xb);             // (null: geomae::set_skip_x_copy -- layers above the first)
                buf_store_b64(xp_r, o2, xpb);
            }
            if (w == 0 && g == 0) { wl[idx] = rec[it].z; wh[idx] = rec[it].w; }
        }
    }
    FUSED_STAMP(1);
    __syncthreads();                                                                           // (1) rows in LDS
    FUSED_STAMP(2);
    load_wfrag<128>(W.frag + kOffWo, w, lane, wo);                   // first use: phase D

    // ---- phase B: q, k (T-layout), v (T-layout for the backward, untransposed for P V) of head w
    uint2 qf[NT], kf[NT], vtf[NT];
    {
        const __amdgpu_buffer_rsrc_t qkv_r = saved_rsrc(A.qkv);
        const f32x4 bq = load_f4(prm + kPBq + 16 * w + 4 * g), bk = load_f4(prm + kPBq + 128 + 16 * w + 4 * g);
        const f32x4 bv = load_f4(prm + kPBq + 256 + 16 * w + 4 * g);
        const float bvn = prm[kPBq + 256 + 16 * w + t];
        FOR_TILES(it) {
            const char* xrow = X + (16 * it + t) * kFRow + 16 * g;
            const char* xprow = XP + (16 * it + t) * kFRow + 16 * g;
            uint4 bx[4], bxp[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) { bxp[kk] = lds_b128(xprow + 64 * kk); bx[kk] = lds_b128(xrow + 64 * kk); }
            f32x4 aq = bq, ak = bk, av = bv, avt = {bvn, bvn, bvn, bvn};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (OWN(it)) aq = mfma32(wq[kk], bxp[kk], aq);       // (q and the T-layout v only for the owned query tiles)
                ak = mfma32(wk[kk], bxp[kk], ak);
                avt = mfma32(bx[kk], wv[kk], avt);
                if (OWN(it)) av = mfma32(wv[kk], bx[kk], av);
            }
            qf[it] = pack4(aq);
            kf[it] = pack4(ak);
            vtf[it] = pack4(avt);
            if (save && OWN(it)) {
                const int o2 = tok[it] >= 0 ? blk_off<2>(tok[it], 384, w, g) : kFOor;
                buf_store_b64(qkv_r, o2, qf[it]);
                buf_store_b64(qkv_r, o2 + 8 * 512, kf[it]);
                buf_store_b64(qkv_r, o2 + 16 * 512, pack4(av));
            }
        }
    }
    FUSED_STAMP(3);
    load_wfrag<128>(W.frag + kOffW1, 2 * w, lane, w1a);                         // in flight under the attention (first use: phase E)
    load_wfrag<128>(W.frag + kOffW1, 2 * w + 1, lane, w1b);

    // ---- phase C: attention of head w over the bundle, block-diagonal by window
    {
        const __amdgpu_buffer_rsrc_t attn_r = saved_rsrc(A.attn), lse_r = saved_rsrc(A.lse);
        const float scale = 0.25f;                                   // 1 / sqrt(16)
#ifdef FUSED_SKIP_PAIRS               // (A/B: tile pairs that no window spans are skipped behind scalar branches in the exact bodies too)
        constexpr bool RANGES = true;
#else
        constexpr bool RANGES = !EXACT;
#endif
        FOR_OWN(it) {
            const int first = 16 * it;
            int jlo = 0, jhi = NT - 1;
            if (RANGES) {
                const int last = first + 15 < T ? first + 15 : T - 1;
                jlo = __builtin_amdgcn_readfirstlane((wl[first] - s0) >> 4);
                jhi = __builtin_amdgcn_readfirstlane((wh[last] - 1 - s0) >> 4);
            }
            const int wq_key = wl[first + t];
            f32x4 st[NT];
            float m = -INFINITY;
#pragma unroll
            for (int jt = 0; jt < NT; ++jt)
                if (!RANGES || (jt >= jlo && jt <= jhi)) {
                    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                    st[jt] = mfma16(kf[jt], qf[it], z4);          // S^T: rows = keys 4g + r, column = query t
                    const int4 W4 = *reinterpret_cast<const int4*>(wl + 16 * jt + 4 * g);
                    const int Wr[4] = {W4.x, W4.y, W4.z, W4.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        st[jt][r] = (Wr[r] == wq_key) ? st[jt][r] * scale : -INFINITY;
                        m = fmaxf(m, st[jt][r]);
                    }
                }
            m = rows4_max(m);
            float sum = 0.f;
            uint2 pb[NT];
#pragma unroll
            for (int jt = 0; jt < NT; ++jt)
                if (!RANGES || (jt >= jlo && jt <= jhi)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = __expf(st[jt][r] - m);
                        st[jt][r] = p;
                        sum += p;
                    }
                    pb[jt] = pack4(st[jt]);
                }
            sum = rows4_sum(sum);
            // O^T = V^T P^T: rows = channels 4g + r of head w, column = query t; two key tiles per K = 32 MFMA
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jp = 0; jp < (NT + 1) / 2; ++jp) {
                const int j0 = 2 * jp, j1 = 2 * jp + 1;
                const bool a0 = !RANGES || (j0 >= jlo && j0 <= jhi);
                const bool a1 = j1 < NT && (!RANGES || (j1 >= jlo && j1 <= jhi));
                if (a0 || a1) {
                    const uint2 zz = make_uint2(0u, 0u);
                    const int j1c = j1 < NT ? j1 : j0;
                    o = mfma32_2(a0 ? vtf[j0] : zz, a1 ? vtf[j1c] : zz, a0 ? pb[j0] : zz, a1 ? pb[j1c] : zz, o);
                }
            }
            const float inv = 1.0f / sum;
            o *= inv;
            const uint2 ob = pack4(o);
            *reinterpret_cast<uint2*>(O + (first + t) * kFRow + perm_b) = ob;
            if (save) {
                buf_store_b64(attn_r, tok[it] >= 0 ? blk_off<2>(tok[it], 128, w, g) : kFOor, ob);
                if (g == 0) buf_store_f32(lse_r, tok[it] >= 0 ? (tok[it] * 8 + w) * 4 : kFOor, m + __logf(sum));
            }
        }
    }
    load_wfrag<256>(W.frag + kOffW2, w, lane, w2);                              // first use: phase F
    FUSED_STAMP(4);
    __syncthreads();                                                                           // (2) attention output in LDS
    FUSED_STAMP(5);

    // ---- phase D: u = x + attn Wo^T + bo (channel tile w), LayerNorm 1
    {
        const f32x4 bo = load_f4(prm + kPBo + 16 * w + 4 * g);
        FOR_OWN(it) {
            const char* orow = O + (16 * it + t) * kFRow + 16 * g;
            f32x4 au = bo;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) au = mfma32(wo[kk], lds_b128(orow + 64 * kk), au);
            xr[it] += au;
            ln_partial(xr[it], red + (16 * it + t) * 16 + 2 * w, g);
        }
    }
    FUSED_STAMP(6);
    __syncthreads();                                                                           // (3) LayerNorm-1 statistics
    FUSED_STAMP(7);
    {
        const __amdgpu_buffer_rsrc_t xh1_r = saved_rsrc(A.xh1), rstd_r = saved_rsrc(A.rstd);
        const f32x4 g1 = load_f4(prm + kPG1 + 16 * w + 4 * g), be1 = load_f4(prm + kPBe1 + 16 * w + 4 * g);
        FOR_OWN(it) {
            float mean, rstd;
            ln_merge(red + (16 * it + t) * 16, A.eps, &mean, &rstd);
            f32x4 xh = (xr[it] - mean) * rstd;
            if (save) {
                buf_store_b64(xh1_r, tok[it] >= 0 ? blk_off<2>(tok[it], 128, w, g) : kFOor, pack4(xh));
                if (w == 0 && g == 0) buf_store_f32(rstd_r, tok[it] >= 0 ? tok[it] * 8 : kFOor, rstd);
            }
            xr[it] = xh * g1 + be1;                                // y: the FFN's input and its residual
            *reinterpret_cast<uint2*>(X + (16 * it + t) * kFRow + perm_b) = pack4(xr[it]);
        }
    }
    FUSED_STAMP(8);
    __syncthreads();                                                                           // (4) y in LDS
    FUSED_STAMP(9);

    // ---- phase E: h = gelu(y W1^T + b1), channel tiles 2w, 2w + 1
    {
        const __amdgpu_buffer_rsrc_t hp_r = saved_rsrc(A.hp);
        const f32x4 b1a = load_f4(prm + kPB1 + 32 * w + 4 * g), b1b = load_f4(prm + kPB1 + 32 * w + 16 + 4 * g);
        FOR_OWN(it) {
            const char* yrow = X + (16 * it + t) * kFRow + 16 * g;
            f32x4 ha = b1a, hb = b1b;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const uint4 by = lds_b128(yrow + 64 * kk);
                ha = mfma32(w1a[kk], by, ha);
                hb = mfma32(w1b[kk], by, hb);
            }
            if (save) {
                const int o2 = tok[it] >= 0 ? blk_off<2>(tok[it], 256, 2 * w, g) : kFOor;
                buf_store_b64(hp_r, o2, pack4(ha));
                buf_store_b64(hp_r, o2 + 512, pack4(hb));
            }
            const f32x4 ga = gelu4(ha), gb = gelu4(hb);
            const uint2 pa = pack4(ga), pb = pack4(gb);
            *reinterpret_cast<uint4*>(H + (16 * it + t) * kFRowH + 2 * (32 * w + 8 * g)) = make_uint4(pa.x, pa.y, pb.x, pb.y);
        }
    }
    FUSED_STAMP(10);
    __syncthreads();                                                                           // (5) gelu output in LDS
    FUSED_STAMP(11);

    // ---- phase F: v = y + h W2^T + b2, LayerNorm 2
    {
        const f32x4 b2 = load_f4(prm + kPB2 + 16 * w + 4 * g);
        FOR_OWN(it) {
            const char* hrow = H + (16 * it + t) * kFRowH + 16 * g;
            f32x4 a = b2;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) a = mfma32(w2[kk], lds_b128(hrow + 64 * kk), a);
            xr[it] += a;
            ln_partial(xr[it], red + (16 * it + t) * 16 + 2 * w, g);
        }
    }
    FUSED_STAMP(12);
    __syncthreads();                                                                           // (6) LayerNorm-2 statistics
    FUSED_STAMP(13);
    {
        const __amdgpu_buffer_rsrc_t xh2_r = saved_rsrc(A.xh2), rstd_r = saved_rsrc(A.rstd), z_r = whole_rsrc(A.z);
        const f32x4 g2 = load_f4(prm + kPG2 + 16 * w + 4 * g), be2 = load_f4(prm + kPBe2 + 16 * w + 4 * g);
        FOR_OWN(it) {
            float mean, rstd;
            ln_merge(red + (16 * it + t) * 16, A.eps, &mean, &rstd);
            const f32x4 xh = (xr[it] - mean) * rstd;
            const int tk = tok[it];
            if (save) {
                buf_store_b64(xh2_r, tk >= 0 ? blk_off<2>(tk, 128, w, g) : kFOor, pack4(xh));
                if (w == 0 && g == 0) buf_store_f32(rstd_r, tk >= 0 ? tk * 8 + 4 : kFOor, rstd);
            }
            const f32x4 zz = xh * g2 + be2;
            const int zo = tk < 0 ? kFOor : (A.z_blocked ? blk_off<4>(tk, 128, w, g) : tk * 512 + 64 * w + 16 * g);
#ifndef FUSED_ABL_NO_Z
            buf_store_f32x4(z_r, zo, zz);
#endif
        }
    }
    FUSED_STAMP(14);
#undef FOR_TILES
#undef FOR_OWN
#undef OWN
}

// The exact bodies run the bundles of 1-4 tiles: every bundle of the packing but a single window that kept more than 64 pillars.
// Those (5-9 tiles) are a SECOND launch.  Round 5 had a generic 9-tile instance of the body above for them: inside this kernel it
// set the register allocation of everything (256 VGPRs, 69 spilled, 26.3 us per layer against 21.4 us for the exact bodies alone),
// as a kernel of its own it still carried 74 spilled registers and 300 B of scratch per lane.  Since round 6 the second launch is
// the looping kernel of sst_ws.hip with min_tiles = 5 (run-time tile loops, online softmax, no scratch): every workgroup looks at
// the bundles of its stride and leaves the small ones alone; a layout without a large bundle costs that launch a scan (~2 us),
// which the step engine avoids by knowing the fullest window a step ahead (window.hip window_max_keep).
__global__ __launch_bounds__(kFusedThreads, 2) void sst_layer_fwd_kernel(FusedFwd A) {
    __shared__ __attribute__((aligned(16))) char lds[kFLdsBytes];
    // ---- work items (window.hip item_pack: bundles of a 32-position packing, those of three / four tiles split by query tile)
    const int NI = A.items ? A.num_items[0] : 0;
    if (NI > 0) {
        for (int i = blockIdx.x; i < NI; i += gridDim.x) {
            const int4 item = A.items[i];
            const int s0 = item.x, T = item.y, nt = (T + 15) >> 4;
            switch (nt * 16 + item.z * 4 + item.w) {
                case 1 * 16 + 0 * 4 + 1: fused_fwd_body<1, true>(A, s0, T, nt, lds); break;
                case 2 * 16 + 0 * 4 + 2: fused_fwd_body<2, true>(A, s0, T, nt, lds); break;
                case 3 * 16 + 0 * 4 + 3: fused_fwd_body<3, true>(A, s0, T, nt, lds); break;
                case 3 * 16 + 0 * 4 + 2: fused_fwd_body<3, true, 0, 2>(A, s0, T, nt, lds); break;
                case 3 * 16 + 2 * 4 + 1: fused_fwd_body<3, true, 2, 1>(A, s0, T, nt, lds); break;
                case 4 * 16 + 0 * 4 + 4: fused_fwd_body<4, true>(A, s0, T, nt, lds); break;
                case 4 * 16 + 0 * 4 + 2: fused_fwd_body<4, true, 0, 2>(A, s0, T, nt, lds); break;
                case 4 * 16 + 2 * 4 + 2: fused_fwd_body<4, true, 2, 2>(A, s0, T, nt, lds); break;
                default:                                     // 5-9 tiles: the second launch (sst_ws.hip, min_tiles = 5)
                    if (nt <= 4 || !A.big_follows) { if (threadIdx.x == 0) atomicAdd(&g_fused_dropped, 1u); }
                    break;
            }
            if (i + (int)gridDim.x < NI) __syncthreads();
        }
        return;
    }
    // ---- no item list (more bundles than CUs, a caller without one): the bundles of the second packing
    // (bun_tok holds max_bundles + 1 >= gridDim.x + 1 words: read before the bundle count is known, one round trip less)
    const int NB = A.num_bundles[0];
    for (int b = blockIdx.x; b < NB; b += gridDim.x) {
#ifdef FUSED_ABL_NO_BUNTOK           // (timing ablation: the bundle-table hop)
        const int s0 = 40 * b, T = 40;
#else
        const int s0 = A.bun_tok[b];
        const int T = A.bun_tok[b + 1] - s0;
#endif
        const int nt = (T + 15) >> 4;
        switch (nt) {
            case 1: fused_fwd_body<1, true>(A, s0, T, nt, lds); break;
            case 2: fused_fwd_body<2, true>(A, s0, T, nt, lds); break;
            case 3: fused_fwd_body<3, true>(A, s0, T, nt, lds); break;
            case 4: fused_fwd_body<4, true>(A, s0, T, nt, lds); break;
            default:                                         // 5-9 tiles: the second launch (sst_ws.hip, min_tiles = 5)
                if (!A.big_follows && threadIdx.x == 0) atomicAdd(&g_fused_dropped, 1u);
                break;
        }
        if (b + (int)gridDim.x < NB) __syncthreads();
    }
}

// ==================================================================================================================
// One launch per SST layer, BACKWARD (round 5): LayerNorm-2 backward -> FFN backward (GELU') -> LayerNorm-1 backward ->
// out-projection backward -> window attention backward -> in-projection backward + residual, for one bundle of windows per
// workgroup.  Reference: the autograd of EncoderLayer / WindowAttention (mmdet3d/models/sst/sst_basic_block.py:26-61, 85-147).
// The unfused build runs a layer's backward as two launches (sst_ffn_bwd_kernel with the in-projection backward of the layer
// above as its head, win_attn_bwd_kernel), with dattn, dqkv and dx_res making a round trip through HBM between them.
//
// Same shape as the forward kernel above: 8 waves, wave w = channel tile w = head w; GEMMs N-split and weight-stationary on
// the TRANSPOSED fragment-major matrices (W2^T, W1^T, Wo^T, Wqk^T, Wv^T: the packed block holds them), activations as B
// operands from LDS rows; the two LayerNorm backwards merge their per-token sums across waves through LDS.  The attention
// backward of head w runs inside wave w: q, k, v, dO as T-layout registers (S^T = K Q^T, dP^T = V dO^T), the transposed
// operands of the three token contractions (K^T, Q^T, dO^T) by ds_read_b64_tr_b16 from a wave-private LDS copy of the head's
// slices -- no workgroup barrier inside the attention.  dq, dk, dv leave as bf16 rows (the operands of the weight-gradient
// contraction) and feed the in-projection backward from LDS; the layer's input gradient is written IN PLACE over its output
// gradient (a bundle reads and writes the same token rows).
// Bundles of 1-4 tiles only: a stack whose layouts may hold a larger one keeps the unfused backward (the engine knows a step
// ahead, window.hip window_max_keep).
constexpr int kBRows = 64;                                           // token rows of a bundle (<= 4 tiles)
constexpr int kBRowQ = 2 * (384 + 8);                                // bytes of a bf16 row of 384 channels (+16 B)
constexpr int kBLdsRA = 0, kBLdsRH = kBRows * kFRow;                 // 128-wide rows | 256-wide rows; the 384-wide rows alias both
constexpr int kBLdsRed = kBLdsRH + kBRows * kFRowH;
static_assert(kBRows * kBRowQ <= kBLdsRed, "the dqkv rows must fit over the two row buffers");
constexpr int kBLdsWl = kBLdsRed + kBRows * 64;
constexpr int kBLdsHead = kBLdsWl + kBRows * 4;                      // per wave: K, Q, dO slices [64][16] bf16, L, D [64] f32
constexpr int kBHeadBytes = 3 * kBRows * 32 + 2 * kBRows * 4;
constexpr int kBLdsPg = kBLdsHead + 8 * kBHeadBytes;                 // LayerNorm parameter gradients of the workgroup: [4][128] f32
constexpr int kBLdsBytes = kBLdsPg + 4 * 128 * 4;

struct FusedBwd {
    const float* dz;              // gradient of the layer's output: token order tile-blocked, or (dz_rowmajor) row-major [n][128]
    const float* dz_add;          // optional second summand (row-major only)
    int dz_rowmajor;
    float* dx;                    // gradient of the layer's input: tile-blocked (may alias dz); or (dx_rowmajor) row-major [n][128],
    const int32_t* out_rows; int n_out;   // ... optionally scattered: row out_rows[token] of [n_out][128]
    int dx_rowmajor;
    const int32_t* bun_tok; const int4* plan; const int32_t* num_bundles;
    LayerW W;
    int n;
    const bf16_t *qkv, *attn, *xh1, *xh2, *hp;                       // saved by the forward (token order, tile-blocked)
    const float *lse, *rstd;
    bf16_t *dqkv, *du, *dv, *dhp, *h;                                // operands of the weight-gradient contraction (written)
    float *dg1, *dbe1, *dg2, *dbe2;                                  // LayerNorm parameter gradients (accumulated)
};

__device__ __forceinline__ uint2 lds_tr(const char* slice, int row0, int lane16) {
    return tr_read(reinterpret_cast<const bf16_t*>(slice + (row0 + (lane16 >> 2)) * 32 + 8 * (lane16 & 3)));
}

template <int NT>
__device__ __forceinline__ void fused_bwd_body(const FusedBwd& A, const int s0, const int T, char* lds) {
#define FOR_TILES(it) _Pragma("unroll") for (int it = 0; it < NT; ++it)
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t = lane & 15, g = lane >> 4;
    char* RA = lds + kBLdsRA;
    char* RH = lds + kBLdsRH;
    char* RQ = lds + kBLdsRA;
    float* red = reinterpret_cast<float*>(lds + kBLdsRed);
    int* wl = reinterpret_cast<int*>(lds + kBLdsWl);
    char* hs = lds + kBLdsHead + w * kBHeadBytes;                    // this wave's head slices
    char* Ks = hs; char* Qs = hs + kBRows * 32; char* Os = hs + 2 * kBRows * 32;
    float* Ls = reinterpret_cast<float*>(hs + 3 * kBRows * 32);
    float* Ds = Ls + kBRows;
    float* pgrad = reinterpret_cast<float*>(lds + kBLdsPg);
    const LayerW& W = A.W;
    const int perm_b = 2 * (32 * (w >> 1) + 8 * g + 4 * (w & 1));

    FUSED_STAMP(0);
    // ---- plan records, then everything that hangs on the token ids
    int tok[NT];
    FOR_TILES(it) {
        const int idx = 16 * it + t;
        int4 rec = make_int4(-1, 0, -1 - idx, 0);
        if (idx < T) rec = A.plan[s0 + idx];
        tok[it] = rec.x;
        if (w == 0 && g == 0) wl[idx] = rec.z;
    }
    f32x4 d[NT];                                                     // the running gradient: this wave's 16 channels of every row
    uint2 xh2[NT];
    float r1[NT], r2[NT];                                             // (rstd of LayerNorm 1 is fetched where it is used: four registers
                                                                      //  fewer across the FFN backward, the NT = 4 body's spills)
    {
        const __amdgpu_buffer_rsrc_t zres = whole_rsrc(A.dz), ares = whole_rsrc(A.dz_add ? A.dz_add : A.dz);
        const __amdgpu_buffer_rsrc_t x2r = whole_rsrc(A.xh2), rsr = whole_rsrc(A.rstd);
        FOR_TILES(it) {
            const int tk = tok[it];
            const int off = tk < 0 ? kFOor : (A.dz_rowmajor ? tk * 512 + 64 * w + 16 * g : blk_off<4>(tk, 128, w, g));
            d[it] = buf_load_f32x4(zres, off);
            if (A.dz_add) d[it] += buf_load_f32x4(ares, off);
            xh2[it] = buf_load_b64(x2r, tk < 0 ? kFOor : blk_off<2>(tk, 128, w, g));
            r2[it] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsr, tk < 0 ? kFOor : tk * 8 + 4, 0, 0));
        }
    }
    uint4 w2a[4], w2b[4];
    load_wfrag<128>(W.frag + kOffW2T, 2 * w, lane, w2a);
    load_wfrag<128>(W.frag + kOffW2T, 2 * w + 1, lane, w2b);
    const f32x4 g2 = load_f4(W.g2 + 16 * w + 4 * g), g1 = load_f4(W.g1 + 16 * w + 4 * g);

    // ---- LayerNorm-2 backward: d <- rstd2 (d g2 - mean(d g2) - xhat2 mean(d g2 xhat2)); parameter gradients of this tile
    f32x4 pg = {0.f, 0.f, 0.f, 0.f}, pb = {0.f, 0.f, 0.f, 0.f};       // sum_t d * xhat, sum_t d (this lane's token column)
    FOR_TILES(it) {
        const f32x4 x = unpack4(xh2[it]);
        pg += d[it] * x;
        pb += d[it];
        d[it] *= g2;
        const float a = rows4_sum((d[it][0] + d[it][1]) + (d[it][2] + d[it][3]));
        const float b = rows4_sum((d[it][0] * x[0] + d[it][1] * x[1]) + (d[it][2] * x[2] + d[it][3] * x[3]));
        if (g == 0) *reinterpret_cast<float2*>(red + (16 * it + t) * 16 + 2 * w) = make_float2(a, b);
    }
    FUSED_STAMP(1);
    __syncthreads();                                                                           // (1) LN2 sums
    FUSED_STAMP(2);
    FOR_TILES(it) {
        f32x4 a4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a4[i] = *reinterpret_cast<const f32x4*>(red + (16 * it + t) * 16 + 4 * i);
        const f32x4 sm = (a4[0] + a4[1]) + (a4[2] + a4[3]);
        const float m1 = (sm[0] + sm[2]) * (1.0f / 128.0f), m2 = (sm[1] + sm[3]) * (1.0f / 128.0f);
        d[it] = (d[it] - m1 - unpack4(xh2[it]) * m2) * r2[it];        // d(y + f)
        *reinterpret_cast<uint2*>(RA + (16 * it + t) * kFRow + perm_b) = pack4(d[it]);
    }
    // the forward's saved pre-activations of this wave's two hidden tiles, and xhat1: in flight under the barrier + GEMM
    uint2 hpa[NT], hpb[NT], xh1[NT];
    {
        const __amdgpu_buffer_rsrc_t hpr = whole_rsrc(A.hp), x1r = whole_rsrc(A.xh1);
        FOR_TILES(it) {
            const int o2 = tok[it] >= 0 ? blk_off<2>(tok[it], 256, 2 * w, g) : kFOor;
            hpa[it] = buf_load_b64(hpr, o2);
            hpb[it] = buf_load_b64(hpr, o2 + 512);
            xh1[it] = buf_load_b64(x1r, tok[it] >= 0 ? blk_off<2>(tok[it], 128, w, g) : kFOor);
        }
    }
    // (loads first: vmcnt retires in order, a load issued behind these stores would wait for them where it is used)
    {   // dv operand of dW2 + LayerNorm-2 parameter gradients
        const __amdgpu_buffer_rsrc_t dvr = whole_rsrc(A.dv);
        FOR_TILES(it) buf_store_b64(dvr, tok[it] >= 0 ? blk_off<2>(tok[it], 128, w, g) : kFOor, pack4(d[it]));
        // (parked in LDS -- ds_add_f32 without a return: no round trip -- and flushed once per workgroup at the end of the
        //  kernel: issued to memory here, ~170 workgroups x 64 atomics on the same few cache lines sit in front of every later
        //  load of the wave; as a read-modify-write of LDS they were 8 dependent round trips, ~2 k cycles per LayerNorm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float sg = row16_sum(pg[r]), sb = row16_sum(pb[r]);
            if (t == 0) { atomicAdd(&pgrad[0 * 128 + 16 * w + 4 * g + r], sg); atomicAdd(&pgrad[1 * 128 + 16 * w + 4 * g + r], sb); }
        }
    }
    uint4 w1[8];
    load_wfrag<256>(W.frag + kOffW1T, w, lane, w1);
    FUSED_STAMP(3);
    __syncthreads();                                                                           // (2) d(y + f) rows in LDS
    FUSED_STAMP(4);

    uint4 wo[4];
    load_wfrag<128>(W.frag + kOffWoT, w, lane, wo);                  // (first use behind the next barrier; ahead of this phase's stores)
    // ---- FFN backward: dh = d W2 (hidden tiles 2w, 2w + 1), dhp = dh gelu'(hp), h = gelu(hp)
    {
        const __amdgpu_buffer_rsrc_t dhr = whole_rsrc(A.dhp), hr = whole_rsrc(A.h);
        FOR_TILES(it) {
            const char* row = RA + (16 * it + t) * kFRow + 16 * g;
            f32x4 ha = {0.f, 0.f, 0.f, 0.f}, hb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const uint4 b = lds_b128(row + 64 * kk);
                ha = mfma32(w2a[kk], b, ha);
                hb = mfma32(w2b[kk], b, hb);
            }
            f32x4 va, ga, vb, gb;
            gelu_fwd_bwd4(unpack4(hpa[it]), &va, &ga);
            gelu_fwd_bwd4(unpack4(hpb[it]), &vb, &gb);
            const uint2 pa = pack4(ha * ga), pbk = pack4(hb * gb);
            *reinterpret_cast<uint4*>(RH + (16 * it + t) * kFRowH + 2 * (32 * w + 8 * g)) = make_uint4(pa.x, pa.y, pbk.x, pbk.y);
            const int o2 = tok[it] >= 0 ? blk_off<2>(tok[it], 256, 2 * w, g) : kFOor;
            buf_store_b64(dhr, o2, pa);
            buf_store_b64(dhr, o2 + 512, pbk);
            buf_store_b64(hr, o2, pack4(va));
            buf_store_b64(hr, o2 + 512, pack4(vb));
        }
    }
    FUSED_STAMP(5);
    __syncthreads();                                                                           // (3) dhp rows in LDS
    FUSED_STAMP(6);

    // ---- dy = d(y + f) + dhp W1 (channel tile w); LayerNorm-1 backward
    pg = f32x4{0.f, 0.f, 0.f, 0.f}; pb = pg;
    FOR_TILES(it) {
        const char* row = RH + (16 * it + t) * kFRowH + 16 * g;
        f32x4 a = d[it];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) a = mfma32(w1[kk], lds_b128(row + 64 * kk), a);
        const f32x4 x = unpack4(xh1[it]);
        pg += a * x;
        pb += a;
        a *= g1;
        d[it] = a;
        const float sa = rows4_sum((a[0] + a[1]) + (a[2] + a[3]));
        const float sb = rows4_sum((a[0] * x[0] + a[1] * x[1]) + (a[2] * x[2] + a[3] * x[3]));
        if (g == 0) *reinterpret_cast<float2*>(red + (16 * it + t) * 16 + 2 * w) = make_float2(sa, sb);
    }
    // this head's q, k, v, attention output and log-sum-exp: in flight under the barrier and the out-projection backward
    uint2 qf[NT], kf[NT], vf[NT], of[NT];
    float lse[NT];
    {
        const __amdgpu_buffer_rsrc_t qr = whole_rsrc(A.qkv), orr = whole_rsrc(A.attn), lr = whole_rsrc(A.lse), rs1 = whole_rsrc(A.rstd);
        FOR_TILES(it) {
            const int o2 = tok[it] >= 0 ? blk_off<2>(tok[it], 384, w, g) : kFOor;
            qf[it] = buf_load_b64(qr, o2);
            kf[it] = buf_load_b64(qr, o2 + 8 * 512);
            vf[it] = buf_load_b64(qr, o2 + 16 * 512);
            of[it] = buf_load_b64(orr, tok[it] >= 0 ? blk_off<2>(tok[it], 128, w, g) : kFOor);
            lse[it] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(lr, tok[it] >= 0 ? (tok[it] * 8 + w) * 4 : kFOor, 0, 0));
            r1[it] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs1, tok[it] >= 0 ? tok[it] * 8 : kFOor, 0, 0));
        }
    }
    FUSED_STAMP(7);
    __syncthreads();                                                                           // (4) LN1 sums
    FUSED_STAMP(8);
    FOR_TILES(it) {
        f32x4 a4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a4[i] = *reinterpret_cast<const f32x4*>(red + (16 * it + t) * 16 + 4 * i);
        const f32x4 sm = (a4[0] + a4[1]) + (a4[2] + a4[3]);
        const float m1 = (sm[0] + sm[2]) * (1.0f / 128.0f), m2 = (sm[1] + sm[3]) * (1.0f / 128.0f);
        d[it] = (d[it] - m1 - unpack4(xh1[it]) * m2) * r1[it];        // du = d(x + attention branch): also the residual part of dx
        *reinterpret_cast<uint2*>(RA + (16 * it + t) * kFRow + perm_b) = pack4(d[it]);
    }
    {
        const __amdgpu_buffer_rsrc_t dur = whole_rsrc(A.du);
        FOR_TILES(it) buf_store_b64(dur, tok[it] >= 0 ? blk_off<2>(tok[it], 128, w, g) : kFOor, pack4(d[it]));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float sg = row16_sum(pg[r]), sb = row16_sum(pb[r]);
            if (t == 0) { atomicAdd(&pgrad[2 * 128 + 16 * w + 4 * g + r], sg); atomicAdd(&pgrad[3 * 128 + 16 * w + 4 * g + r], sb); }
        }
    }
    FUSED_STAMP(9);
    __syncthreads();                                                                           // (5) du rows in LDS
    FUSED_STAMP(10);

    // ---- dO of head w = du Wo (channel tile w), then the attention backward of head w inside this wave
    uint2 dof[NT];
    FOR_TILES(it) {
        const char* row = RA + (16 * it + t) * kFRow + 16 * g;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) a = mfma32(wo[kk], lds_b128(row + 64 * kk), a);
        dof[it] = pack4(a);
        // delta_i = sum_d dO[i][d] O[i][d] over the head's 16 channels (dO as the bf16 the contractions see)
        const f32x4 dob = unpack4(dof[it]), ob = unpack4(of[it]);
        const float dl = rows4_sum((dob[0] * ob[0] + dob[1] * ob[1]) + (dob[2] * ob[2] + dob[3] * ob[3]));
        const int idx = 16 * it + t;
        *reinterpret_cast<uint2*>(Ks + idx * 32 + 8 * g) = kf[it];
        *reinterpret_cast<uint2*>(Qs + idx * 32 + 8 * g) = qf[it];
        *reinterpret_cast<uint2*>(Os + idx * 32 + 8 * g) = dof[it];
        if (g == 0) { Ls[idx] = idx < T ? lse[it] : INFINITY; Ds[idx] = dl; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    FUSED_STAMP(11);
    const float scale = 0.25f;
    uint2 dqb[NT], dkb[NT], dvb[NT];
    // pass 1, dQ: lane = query i (column), rows = keys 4g + r of tile jt
    FOR_TILES(it) {
        const int wq = wl[16 * it + t];
        const float Li = Ls[16 * it + t], Di = Ds[16 * it + t];
        f32x4 dq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            const f32x4 sT = mfma16(kf[jt], qf[it], z4);
            const f32x4 dpT = mfma16(vf[jt], dof[it], z4);
            const int4 W4 = *reinterpret_cast<const int4*>(wl + 16 * jt + 4 * g);
            const int Wr[4] = {W4.x, W4.y, W4.z, W4.w};
            f32x4 ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = (Wr[r] == wq) ? __expf(sT[r] * scale - Li) : 0.0f;
                ds[r] = p * (dpT[r] - Di) * scale;
            }
            dq = mfma16(lds_tr(Ks, 16 * jt + 4 * g, t), pack4(ds), dq);          // dQ^T[d][i] += K^T[d][j] dS^T[j][i]
        }
        dqb[it] = pack4(dq);
        __builtin_amdgcn_sched_barrier(0);                               // (keeps the next tile's LDS reads from piling up: NT = 4 spilled)
    }
    FUSED_STAMP(12);
    uint4 wqk[8], wv[4];                                             // the in-projection's weights: in flight under pass 2
    load_wfrag<256>(W.frag + kOffWqkT, w, lane, wqk);
    // pass 2, dK and dV: lane = key j (column), rows = queries 4g + r of tile it
    FOR_TILES(jt) {
        const int wk = wl[16 * jt + t];
        f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int it = 0; it < NT; ++it) {
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            const f32x4 sS = mfma16(qf[it], kf[jt], z4);
            const f32x4 dp = mfma16(dof[it], vf[jt], z4);
            const int i0 = 16 * it + 4 * g;
            const f32x4 L4 = *reinterpret_cast<const f32x4*>(Ls + i0), D4 = *reinterpret_cast<const f32x4*>(Ds + i0);
            const int4 W4 = *reinterpret_cast<const int4*>(wl + i0);
            const int Wr[4] = {W4.x, W4.y, W4.z, W4.w};
            f32x4 pp, ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = (Wr[r] == wk) ? __expf(sS[r] * scale - L4[r]) : 0.0f;
                pp[r] = p;
                ds[r] = p * (dp[r] - D4[r]) * scale;
            }
            dv = mfma16(lds_tr(Os, i0, t), pack4(pp), dv);                       // dV^T[d][j] += dO^T[d][i] P[i][j]
            dk = mfma16(lds_tr(Qs, i0, t), pack4(ds), dk);                       // dK^T[d][j] += Q^T[d][i] dS[i][j]
        }
        dkb[jt] = pack4(dk);
        dvb[jt] = pack4(dv);
        __builtin_amdgcn_sched_barrier(0);
    }
    FUSED_STAMP(13);
    load_wfrag<128>(W.frag + kOffWvT, w, lane, wv);
    __syncthreads();                                                                           // (6) du rows consumed by every wave
    {
        const __amdgpu_buffer_rsrc_t dqr = whole_rsrc(A.dqkv);
        FOR_TILES(it) {
            char* row = RQ + (16 * it + t) * kBRowQ + perm_b;
            *reinterpret_cast<uint2*>(row) = dqb[it];
            *reinterpret_cast<uint2*>(row + 256) = dkb[it];
            *reinterpret_cast<uint2*>(row + 512) = dvb[it];
            const int o2 = tok[it] >= 0 ? blk_off<2>(tok[it], 384, w, g) : kFOor;
            buf_store_b64(dqr, o2, dqb[it]);
            buf_store_b64(dqr, o2 + 8 * 512, dkb[it]);
            buf_store_b64(dqr, o2 + 16 * 512, dvb[it]);
        }
    }
    __syncthreads();                                                                           // (7) dqkv rows in LDS
    FUSED_STAMP(14);

    // ---- dx = du + dqk Wqk + dv Wv (channel tile w)
    {
        const __amdgpu_buffer_rsrc_t xr = whole_rsrc(A.dx);
        FOR_TILES(it) {
            const char* row = RQ + (16 * it + t) * kBRowQ + 16 * g;
            f32x4 a = d[it];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) a = mfma32(wqk[kk], lds_b128(row + 64 * kk), a);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) a = mfma32(wv[kk], lds_b128(row + 512 + 64 * kk), a);
            const int tk = tok[it];
            int off = kFOor;
            if (tk >= 0) {
                if (A.out_rows) {
                    const int orow = A.out_rows[tk];
                    off = orow < A.n_out ? orow * 512 + 64 * w + 16 * g : kFOor;
                } else if (A.dx_rowmajor) {
                    off = tk * 512 + 64 * w + 16 * g;
                } else {
                    off = blk_off<4>(tk, 128, w, g);
                }
            }
            buf_store_f32x4(xr, off, a);
        }
    }
    FUSED_STAMP(15);
#undef FOR_TILES
}

__global__ __launch_bounds__(kFusedThreads, 2) void sst_layer_bwd_kernel(FusedBwd A) {
    __shared__ __attribute__((aligned(16))) char lds[kBLdsBytes];
    const int NB = A.num_bundles[0];
    reinterpret_cast<float*>(lds + kBLdsPg)[threadIdx.x] = 0.f;          // 512 threads = [4][128]
    bool any = false;
    for (int b = blockIdx.x; b < NB; b += gridDim.x) {
        any = true;
        const int s0 = A.bun_tok[b];
        const int T = A.bun_tok[b + 1] - s0;
        const int nt = (T + 15) >> 4;
        switch (nt) {
            case 1: fused_bwd_body<1>(A, s0, T, lds); break;
            case 2: fused_bwd_body<2>(A, s0, T, lds); break;
            case 3: fused_bwd_body<3>(A, s0, T, lds); break;
            case 4: fused_bwd_body<4>(A, s0, T, lds); break;
            default:                                         // (the host keeps the unfused backward for such layouts)
                if (threadIdx.x == 0) atomicAdd(&g_fused_dropped, 1u);
                break;
        }
        if (b + (int)gridDim.x < NB) __syncthreads();
    }
    if (!any) return;                                        // (workgroup-uniform)
    __syncthreads();
    {
        const int k = threadIdx.x >> 7, c = threadIdx.x & 127;
        float* dst = k == 0 ? A.dg2 : (k == 1 ? A.dbe2 : (k == 2 ? A.dg1 : A.dbe1));
        atomicAdd(dst + c, reinterpret_cast<const float*>(lds + kBLdsPg)[threadIdx.x]);
    }
}

// one workgroup per bundle; the bundle count lives on the device, its bound from the greedy packing is 2 n / cap + 1
static int fused_grid(int num_tokens, int max_bundles, int cap) {
    int64_t nb = 2 * (int64_t)num_tokens / (cap > 0 ? cap : 1) + 2;
    if (nb > max_bundles) nb = max_bundles;
    if (nb > 4096) nb = 4096;
    return (int)(nb < 1 ? 1 : nb);
}

}  // namespace geomae

using namespace geomae;

#ifdef GEOMAE_PHASE_TIMING
extern "C" int geomae_debug_read_fused_stamps(unsigned long long* host, int clear) {
    hipDeviceSynchronize();
    if (host) hipMemcpyFromSymbol(host, HIP_SYMBOL(geomae_stamps), sizeof(unsigned long long) * GEOMAE_STAMP_BLOCKS * GEOMAE_STAMP_SLOTS);
    if (clear) {
        static unsigned long long zeros[GEOMAE_STAMP_BLOCKS * GEOMAE_STAMP_SLOTS];
        hipMemcpyToSymbol(HIP_SYMBOL(geomae_stamps), zeros, sizeof(zeros));
    }
    return 0;
}
#endif

// The backward of one layer as ONE launch (sst_layer_bwd_kernel).  Internal: geomae_sst_stack_backward calls it per layer when
// the stack qualifies (sst_stack.hip).  dz: tile-blocked token order, or row-major (+ optional dz_add); dx likewise (+ scatter).
int geomae::sst_layer_backward_fused(const float* dz, const float* dz_add, bool dz_rowmajor, float* dx, bool dx_rowmajor,
                                     const int32_t* out_rows, int n_out, int num_tokens, const GeomaeSstLayerWeights* w,
                                     const GeomaeSstLayerGrads* g, const GeomaeSstStackLayout* layout, int bundle_cap,
                                     const void* qkv, const void* attn, const float* lse, const void* xh1, const void* xh2,
                                     const void* hp, const float* rstd, void* dqkv, void* du, void* dv, void* dhp, void* h,
                                     hipStream_t stream) {
    GEOMAE_REQUIRE(w && w->frag_p && g && layout && layout->fbun_tok && layout->pos_info && layout->num_fbundles,
                   "sst_layer_backward_fused: plan / fragment-major weights missing");
    FusedBwd A;
    memset(&A, 0, sizeof(A));
    A.dz = dz; A.dz_add = dz_add; A.dz_rowmajor = dz_rowmajor ? 1 : 0;
    A.dx = dx; A.dx_rowmajor = dx_rowmajor ? 1 : 0; A.out_rows = out_rows; A.n_out = n_out;
    A.bun_tok = layout->fbun_tok; A.plan = (const int4*)layout->pos_info; A.num_bundles = layout->num_fbundles;
    A.W = to_layer(w); A.n = num_tokens;
    A.qkv = (const bf16_t*)qkv; A.attn = (const bf16_t*)attn; A.xh1 = (const bf16_t*)xh1; A.xh2 = (const bf16_t*)xh2;
    A.hp = (const bf16_t*)hp; A.lse = lse; A.rstd = rstd;
    A.dqkv = (bf16_t*)dqkv; A.du = (bf16_t*)du; A.dv = (bf16_t*)dv; A.dhp = (bf16_t*)dhp; A.h = (bf16_t*)h;
    A.dg1 = g->ln1_w; A.dbe1 = g->ln1_b; A.dg2 = g->ln2_w; A.dbe2 = g->ln2_b;
    const int grid = fused_grid(num_tokens, layout->max_bundles, bundle_cap);
    hipLaunchKernelGGL(sst_layer_bwd_kernel, dim3(grid), dim3(kFusedThreads), 0, stream, A);
    return check_launch("sst_layer_bwd_kernel");
}

extern "C" int geomae_sst_fused_dropped_bundles(int64_t* out, int32_t reset) {
    GEOMAE_REQUIRE(out, "sst_fused_dropped_bundles: null argument");
    unsigned int v = 0;
    GEOMAE_HIP(hipDeviceSynchronize());
    GEOMAE_HIP(hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_fused_dropped), sizeof(v)));
    if (reset) {
        const unsigned int z = 0;
        GEOMAE_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_fused_dropped), &z, sizeof(z)));
    }
    *out = v;
    return GEOMAE_OK;
}

extern "C" int geomae_sst_layer_forward(const float* x, int32_t num_tokens, const GeomaeSstLayerWeights* w,
                                        const GeomaeSstStackLayout* layout, int32_t bundle_cap, const float* pos_table,
                                        float* z, int32_t z_blocked, void* qkv_bf16, void* attn_bf16, float* lse,
                                        void* xhat1_bf16, void* xhat2_bf16, void* hp_bf16, float* rstd, void* x_bf16,
                                        void* xp_bf16, hipStream_t stream) {
    const bool big_possible = take_fused_big_next();
    if (num_tokens <= 0) return GEOMAE_OK;
    int rc = check_weights(w, "sst_layer_forward");
    if (rc) return rc;
    GEOMAE_REQUIRE(num_tokens <= 2700000, "sst_layer_forward: more than 2.7 M tokens per call");
    GEOMAE_REQUIRE(layout && layout->fbun_tok && layout->pos_info && layout->num_fbundles && layout->max_bundles >= 1,
                   "sst_layer_forward: the layout needs the build's plan and second packing (pos_info, fbun_tok, num_fbundles)");
    GEOMAE_REQUIRE(pos_table && z, "sst_layer_forward: null argument");
    // a bundle is whole windows up to the soft cap, or ONE window larger than it: the kernel's LDS holds kFMaxT = 144 rows
    // (a 12 x 12 window).  A layout packed with a larger cap would index past them (the kernel also clamps, below).
    GEOMAE_REQUIRE(bundle_cap >= 1 && bundle_cap <= kFMaxT, "sst_layer_forward: bundle_cap must be in [1, 144]");
    GEOMAE_REQUIRE(w->frag_p, "sst_layer_forward: the layer has no fragment-major packed weights (frag_p)");
    const SstInputMap M = input_map();
    GEOMAE_REQUIRE(x || M.src, "sst_layer_forward: null input");
    const bool save = qkv_bf16 || attn_bf16 || lse || xhat1_bf16 || xhat2_bf16 || hp_bf16 || rstd || x_bf16 || xp_bf16;
    GEOMAE_REQUIRE(!save || (qkv_bf16 && attn_bf16 && lse && xhat1_bf16 && xhat2_bf16 && hp_bf16 && rstd && x_bf16 && xp_bf16),
                   "sst_layer_forward: pass all nine save buffers or none");
    FusedFwd A;
    A.x = x; A.M = M; A.bun_tok = layout->fbun_tok; A.plan = (const int4*)layout->pos_info; A.num_bundles = layout->num_fbundles;
    A.pos_table = pos_table; A.W = to_layer(w); A.n = num_tokens; A.eps = w->ln_eps; A.z = z; A.z_blocked = z_blocked;
    A.qkv = (bf16_t*)qkv_bf16; A.attn = (bf16_t*)attn_bf16; A.xh1 = (bf16_t*)xhat1_bf16; A.xh2 = (bf16_t*)xhat2_bf16;
    A.hp = (bf16_t*)hp_bf16; A.xb = skip_x_copy() ? nullptr : (bf16_t*)x_bf16; A.xp = (bf16_t*)xp_bf16; A.lse = lse; A.rstd = rstd;
    A.big_follows = big_possible ? 1 : 0;
    if (layout->fitems && layout->num_fitems && tuning().fwd_item_cap > 0) {
        A.items = (const int4*)layout->fitems; A.num_items = layout->num_fitems;
    }
    // (an item list exists only when it fits one round of workgroups; without one the kernel's loop covers any bundle count)
    const int grid = fused_grid(num_tokens, layout->max_bundles, A.items ? (tuning().fwd_item_cap < bundle_cap ? tuning().fwd_item_cap : bundle_cap) : bundle_cap);
    hipLaunchKernelGGL(sst_layer_fwd_kernel, dim3(grid), dim3(kFusedThreads), 0, stream, A);
    rc = check_launch("sst_layer_fwd_kernel");
    if (rc) return rc;
    // the bundles of 5-9 tiles (a window that kept more than 64 pillars): rare in the token sets this form is chosen for, the
    // rule at the decoders' sizes (geomae_sst_set_fused_layers(2), tests).  Since round 6 the looping kernel of sst_ws.hip runs
    // them (run-time tile loops, no scratch; the generic 9-tile body it replaces held 74 spilled registers and 300 B of scratch
    // per lane): every workgroup looks at the bundles of its stride and leaves those of at most four tiles alone.
    if (!big_possible) return GEOMAE_OK;         // (the caller knows this layout's fullest window: common.h set_fused_big_layouts)
    return sst_layer_forward_ws(x, M, num_tokens, w, layout, pos_table, z, z_blocked != 0, qkv_bf16, attn_bf16, lse, xhat1_bf16,
                                xhat2_bf16, hp_bf16, rstd, A.xb, xp_bf16, 0, grid < 256 ? grid : 256, 5, stream);
}
