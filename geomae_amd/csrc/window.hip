// SST window partition + per-window variable-length multi-head attention (gfx950).
//
// Reference: MultiMAESSTSPChoose.window_partition / get_voxel_keep_inds / get_flat2win_inds /
// get_inner_win_inds (mmdet3d/models/backbones/multi_mae_sst_spearate_top_only.py:413-681),
// flat2window / window2flat (mmdet3d/ops/sst/sst_ops.py:98-135,225-251) and WindowAttention
// (mmdet3d/models/sst/sst_basic_block.py:26-61), which zero-pads every window to 56 or 144
// tokens and runs nn.MultiheadAttention with a key-padding mask per bucket.
//
// Here a window is a CSR segment: no padding to a bucket size, no [W,T,C] tensors, no
// .item() syncs.  window_build is a counting sort of tokens by window id over the dense
// (B * nwx * nwy) window table; attention runs one wavefront per (window, head) with the
// head's Q/K/V slices staged in LDS (<= 144 x 16 bf16 each) and the 16x16x16 bf16 MFMA
// for QK^T, PV and all backward products, softmax in fp32 registers.  Tiles are 16 tokens
// wide, so a window of n tokens costs ceil(n/16)^2 tiles instead of 56^2 or 144^2.
#include "common.h"
#include "../../include/geomae_hip.h"

namespace geomae {

constexpr int kWBlk = 256;

struct WinGeom {
    int wx, wy;        // window shape (x, y)
    int nwx, nwy;      // windows per sample along x, y (incl. the +1 for shifts)
    int shift_x, shift_y;
};

__device__ __forceinline__ void win_of(const int4 c, const WinGeom g, int* win, int* pos) {
    const int sx = c.w + (g.shift_x > 0 ? g.wx - g.shift_x : 0);
    const int sy = c.z + (g.shift_y > 0 ? g.wy - g.shift_y : 0);
    *win = c.x * (g.nwx * g.nwy) + (sx / g.wx) * g.nwy + sy / g.wy;
    *pos = (sx % g.wx) * g.wy + (sy % g.wy);
}

__device__ __forceinline__ void win_hist_body(const int4* __restrict__ coors, int n, WinGeom g,
                                              int32_t* __restrict__ table, int32_t* __restrict__ rank,
                                              int32_t* __restrict__ tok_win_id, int32_t* __restrict__ tok_pos) {
    for (int i = blockIdx.x * kWBlk + threadIdx.x; i < n; i += gridDim.x * kWBlk) {
        int w, p;
        win_of(coors[i], g, &w, &p);
        tok_win_id[i] = w;
        tok_pos[i] = p;
        rank[i] = atomicAdd(&table[w], 1);
    }
}
__global__ __launch_bounds__(kWBlk) void win_hist_kernel(const int4* __restrict__ coors, int n, WinGeom g,
                                                         int32_t* __restrict__ table, int32_t* __restrict__ rank,
                                                         int32_t* __restrict__ tok_win_id,
                                                         int32_t* __restrict__ tok_pos) {
    win_hist_body(coors, n, g, table, rank, tok_win_id, tok_pos);
}

// One launch of each build stage serves up to kMaxWinJobs layouts (blockIdx.y = job): a training step builds four
// (encoder / decoder tokens x unshifted / shifted windows), and five tiny dependent kernels per layout, two of them
// single-workgroup, were 24 launches = ~0.3 ms of a serial chain that gated the encoder (tools/archive/phase_events.py).
constexpr int kMaxWinJobs = 4;
struct WinJob {
    const int4* coors;
    int n, slots, cap;
    WinGeom g;
    int32_t *table, *rank, *win_start, *win_tokens, *tok_win, *tok_pos, *num_windows, *bun_start, *num_bundles;
    int fcap;                // second packing (the one-launch layer kernel's bundles): cap, starts, count; or fbun_tok null
    int32_t *fbun_tok, *num_fbundles;
    int icap, ibudget;       // the one-launch forward's work items: cap of their packing, workgroups of one round (the CU count)
    int4* fitems;            // ... (first position, positions, first query tile, query tiles) per item, or null
    int32_t* num_fitems;
    int32_t* bun_tok;        // optional attention plan (see bundle_setup): token position where each bundle starts
    int4* pos_info;          // ... and per position of win_tokens: (token, in-window position = pos-embed row, window start, window end)
};
struct WinJobs { WinJob j[kMaxWinJobs]; };

__global__ __launch_bounds__(kWBlk) void win_hist_jobs_kernel(WinJobs J) {
    const WinJob& j = J.j[blockIdx.y];
    win_hist_body(j.coors, j.n, j.g, j.table, j.rank, j.tok_win, j.tok_pos);
}

// one workgroup: compact the non-empty windows (ascending id) and exclusive-scan their sizes
__global__ __launch_bounds__(1024) void win_scan_jobs_kernel(WinJobs J) {
    const WinJob& j = J.j[blockIdx.y];
    int32_t* __restrict__ table = j.table; const int n_slots = j.slots; int32_t* __restrict__ win_start = j.win_start;
    int32_t* __restrict__ num_windows = j.num_windows; const int n_tokens = j.n;
    __shared__ int sm[40];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int run_occ = 0, run_cnt = 0;
    for (int base = 0; base < n_slots; base += 1024) {
        const int t = base + threadIdx.x;
        const int v = t < n_slots ? table[t] : 0;
        int occ = v > 0, cnt = v;
        int io = occ, ic = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int a = __shfl_up(io, o, 64), b = __shfl_up(ic, o, 64);
            if (lane >= o) { io += a; ic += b; }
        }
        if (lane == 63) { sm[w] = io; sm[16 + w] = ic; }
        __syncthreads();
        int oo = 0, oc = 0, to = 0, tc = 0;
        for (int k = 0; k < 16; ++k) {
            if (k < w) { oo += sm[k]; oc += sm[16 + k]; }
            to += sm[k]; tc += sm[16 + k];
        }
        __syncthreads();
        const int eo = run_occ + oo + io - occ, ec = run_cnt + oc + ic - cnt;
        if (t < n_slots) {
            if (v > 0) { win_start[eo] = ec; table[t] = eo; } else { table[t] = -1; }
        }
        run_occ += to;
        run_cnt += tc;
    }
    if (threadIdx.x == 0) { num_windows[0] = run_occ; win_start[run_occ] = n_tokens; }
}

__global__ __launch_bounds__(kWBlk) void win_place_jobs_kernel(WinJobs J) {
    const WinJob& j = J.j[blockIdx.y];
    const int n = j.n; const int32_t* __restrict__ table = j.table; const int32_t* __restrict__ rank = j.rank;
    const int32_t* __restrict__ win_start = j.win_start; int32_t* __restrict__ tok_win_id = j.tok_win;
    int32_t* __restrict__ win_tokens = j.win_tokens;
    for (int i = blockIdx.x * kWBlk + threadIdx.x; i < n; i += gridDim.x * kWBlk) {
        const int w = table[tok_win_id[i]];
        tok_win_id[i] = w;     // now the compact (CSR) window index
        win_tokens[win_start[w] + rank[i]] = i;
    }
}

// make the in-window token order deterministic (ascending token index): rank sort, one wave / window
__global__ __launch_bounds__(64) void win_sort_jobs_kernel(WinJobs J) {
    const WinJob& j = J.j[blockIdx.y];
    const int32_t* __restrict__ win_start = j.win_start; const int32_t* __restrict__ num_windows = j.num_windows;
    int32_t* __restrict__ win_tokens = j.win_tokens;
    __shared__ int a[1024];
    const int W = num_windows[0];
    for (int w = blockIdx.x; w < W; w += gridDim.x) {
        const int s = win_start[w], n = win_start[w + 1] - s;
        if (n <= 1024) {
            for (int t = threadIdx.x; t < n; t += 64) a[t] = win_tokens[s + t];
            __syncthreads();
            for (int t = threadIdx.x; t < n; t += 64) {
                const int v = a[t];
                int r = 0;
                for (int u = 0; u < n; ++u) r += a[u] < v;
                win_tokens[s + r] = v;
                if (j.pos_info) j.pos_info[s + r] = make_int4(v, j.tok_pos[v], s, s + n);
            }
            __syncthreads();
        } else if (j.pos_info) {
            for (int t = threadIdx.x; t < n; t += 64) {
                const int v = win_tokens[s + t];
                j.pos_info[s + t] = make_int4(v, j.tok_pos[v], s, s + n);
            }
        }
    }
}

// greedy packing of consecutive windows into bundles of at most `cap` tokens (a window larger than the cap is a bundle of
// its own): most windows hold a handful of pillars, and one wavefront per (window, head) drowned in fixed per-wave latency
// (profiles/r01d: 16k waves per launch).  Two packings per layout: the attention kernels' (cap = a whole window, bun_start /
// bun_tok / num_bundles) and, optionally, the one-launch layer kernel's (sst_fused.hip: fcap, fbun_tok / num_fbundles).
struct BundleLds { int ws[8192]; int nx[8192]; int ja[8192 + 1]; int jb[8192 + 1]; int n_anchors; };

__device__ __forceinline__ void bundle_pack(const WinJob& j, int cap, int32_t* __restrict__ bun_start /*or null*/,
                                            int32_t* __restrict__ bun_tok /*or null*/, int32_t* __restrict__ num_bundles,
                                            BundleLds& L, int W, bool in_lds) {
    const int32_t* __restrict__ win_start = j.win_start;
    int32_t* __restrict__ nxt_ws = j.rank;      // reused: the window counting sort is done with it
    // nxt[w] = end (exclusive) of the greedy bundle that starts at window w: the largest e with
    // win_start[e] - win_start[w] <= cap (binary search, all threads); then one thread follows the chain
    // 0 -> nxt[0] -> ... (one dependent read per BUNDLE instead of per window).
    for (int w = threadIdx.x; w < W; w += 1024) {
        const int limit = (in_lds ? L.ws[w] : win_start[w]) + cap;
        int lo = w + 1, hi = W;                          // answer in [w+1, W] (w+1: a window larger than the cap)
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            const int v = in_lds ? L.ws[mid] : win_start[mid];
            if (v <= limit) lo = mid; else hi = mid - 1;
        }
        if (in_lds) L.nx[w] = lo; else nxt_ws[w] = lo;
    }
    __syncthreads();
    if (!in_lds) {                                       // huge window tables: the plain serial walk
        if (threadIdx.x == 0) {
            int nb = 0, w = 0;
            while (w < W) {
                if (bun_tok) bun_tok[nb] = win_start[w];
                if (bun_start) bun_start[nb] = w;
                ++nb;
                w = nxt_ws[w];
            }
            if (bun_start) bun_start[nb] = W;
            if (bun_tok) bun_tok[nb] = j.n;
            num_bundles[0] = nb;
        }
        __syncthreads();
        return;
    }
    // The walk 0 -> nx[0] -> nx[nx[0]] ... is one dependent LDS read per bundle (~300 at decoder size: most of this
    // kernel's 27 us, on the chain that gates the encoder).  Four rounds of pointer doubling give the 16-hop map;
    // one thread walks THAT (~20 reads) and records the anchors, then every anchor's thread fills its 16 bundles.
    if (threadIdx.x == 0) L.nx[W] = W;                    // sentinel: the walk stops at W
    __syncthreads();
    for (int w = threadIdx.x; w <= W; w += 1024) L.ja[w] = L.nx[L.nx[w]];          // 2 hops
    __syncthreads();
    for (int w = threadIdx.x; w <= W; w += 1024) L.jb[w] = L.ja[L.ja[w]];          // 4
    __syncthreads();
    for (int w = threadIdx.x; w <= W; w += 1024) L.ja[w] = L.jb[L.jb[w]];          // 8
    __syncthreads();
    for (int w = threadIdx.x; w <= W; w += 1024) L.jb[w] = L.ja[L.ja[w]];          // 16
    __syncthreads();
    if (threadIdx.x == 0) {
        int na = 0, w = 0;
        while (w < W) { L.ja[na++] = w; w = L.jb[w]; }    // ja is free again: it now holds the anchors
        L.n_anchors = na;
        if (na == 0) { if (bun_start) bun_start[0] = W; if (bun_tok) bun_tok[0] = j.n; num_bundles[0] = 0; }
    }
    __syncthreads();
    const int na = L.n_anchors;
    for (int a = threadIdx.x; a < na; a += 1024) {
        int w = L.ja[a], k = 0;
        for (; k < 16 && w < W; ++k) {
            if (bun_start) bun_start[16 * a + k] = w;
            if (bun_tok) bun_tok[16 * a + k] = L.ws[w];
            w = L.nx[w];
        }
        if (a == na - 1) {                                // the last anchor's thread knows the bundle count
            const int nb = 16 * a + k;
            if (bun_start) bun_start[nb] = W;
            if (bun_tok) bun_tok[nb] = j.n;
            num_bundles[0] = nb;
        }
    }
    __syncthreads();
}

// Work items of the one-launch layer FORWARD (sst_fused.hip; round 6).  Its launch lasts as long as its longest work item --
// a 3-tile bundle's chain is 29 k cycles, a 4-tile bundle's 38 k, a 2-tile bundle's 21 k (tools/fused_layer_time.py) -- while a
// small token set has fewer bundles than the device has CUs (163 at config 2's encoder).  So: a THIRD packing with a cap of
// `icap` (32) positions -- mostly two-tile bundles -- whose three- and four-tile bundles (a window that kept 33-64 pillars) are
// split by QUERY tile into two items: each computes k / v for the whole bundle, attention and the row-wise rest for its own
// tiles.  Everything must still fit ONE round of workgroups (`ibudget` = CUs, one 133-KB-LDS workgroup each): the four-tile
// bundles are split first, the three-tile ones if the count allows; with more bundles than CUs no item list is made
// (num_fitems = 0) and the kernel walks the second packing as before.  The backward keeps the second packing: it cannot split
// (dK / dV sum over all queries of a window) and shares the chip with the contraction launches.
__device__ __forceinline__ void item_pack(const WinJob& j, BundleLds& L, int W, bool in_lds) {
    if (!in_lds || j.icap <= 0) {
        if (threadIdx.x == 0) j.num_fitems[0] = 0;
        return;
    }
    // the packing's bundle starts go to the tail of the item buffer (it holds 8 (mw + 1) ints; items grow from its head)
    const int mw = j.n < j.slots ? j.n : j.slots;
    int32_t* tmp = reinterpret_cast<int32_t*>(j.fitems) + 7 * ((mw < 1 ? 1 : mw) + 1);
    bundle_pack(j, j.icap, nullptr, tmp, j.num_fitems, L, W, in_lds);
    __threadfence_block();
    __syncthreads();
    const int nb = j.num_fitems[0];
    __shared__ int cnt3, cnt4, total;
    if (threadIdx.x == 0) { cnt3 = 0; cnt4 = 0; total = 0; }
    // bundle starts into LDS (ja: free after the packing): the items written below may reach the tail of the buffer
    for (int b = threadIdx.x; b <= nb && b <= 8192; b += 1024) L.ja[b] = tmp[b];
    __syncthreads();
    if (nb > j.ibudget || nb > 8191) {                   // (uniform) more bundles than one round of workgroups
        if (threadIdx.x == 0) j.num_fitems[0] = 0;
        return;
    }
    int c3 = 0, c4 = 0;
    for (int b = threadIdx.x; b < nb; b += 1024) {
        const int nt = (L.ja[b + 1] - L.ja[b] + 15) >> 4;
        c3 += nt == 3; c4 += nt == 4;
    }
    if (c3) atomicAdd(&cnt3, c3);
    if (c4) atomicAdd(&cnt4, c4);
    __syncthreads();
    const bool split4 = nb + cnt4 <= j.ibudget, split3 = nb + cnt4 + cnt3 <= j.ibudget;
    // item offsets: an exclusive scan of the parts per bundle (nb <= 8191: eight bundles per thread, then one scan of 1024 sums)
    int first = threadIdx.x * 8, mine = 0, parts[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int b = first + k;
        int p = 0;
        if (b < nb) {
            const int nt = (L.ja[b + 1] - L.ja[b] + 15) >> 4;
            p = 1 + ((split3 && nt == 3) || (split4 && nt == 4) ? 1 : 0);
        }
        parts[k] = p; mine += p;
    }
    L.jb[threadIdx.x] = mine;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = threadIdx.x >= d ? L.jb[threadIdx.x - d] : 0;
        __syncthreads();
        L.jb[threadIdx.x] += v;
        __syncthreads();
    }
    int off = L.jb[threadIdx.x] - mine;
    if (threadIdx.x == 1023) total = L.jb[1023];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int b = first + k;
        if (b < nb) {
            const int s0 = L.ja[b], T = L.ja[b + 1] - s0, nt = (T + 15) >> 4;
            if (parts[k] == 2) {
                j.fitems[off] = make_int4(s0, T, 0, 2);
                j.fitems[off + 1] = make_int4(s0, T, 2, nt - 2);
            } else {
                j.fitems[off] = make_int4(s0, T, 0, nt);
            }
            off += parts[k];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) j.num_fitems[0] = total;
}

__global__ __launch_bounds__(1024) void win_bundle_jobs_kernel(WinJobs J) {
    const WinJob& j = J.j[blockIdx.y];
    __shared__ BundleLds L;
    const int W = j.num_windows[0];
    const bool in_lds = W < 8192;
    if (in_lds)
        for (int t = threadIdx.x; t <= W; t += 1024) L.ws[t] = j.win_start[t];
    __syncthreads();
    bundle_pack(j, j.cap, j.bun_start, j.bun_tok, j.num_bundles, L, W, in_lds);
    if (j.fbun_tok) bundle_pack(j, j.fcap, nullptr, j.fbun_tok, j.num_fbundles, L, W, in_lds);
    if (j.fitems) item_pack(j, L, W, in_lds);
}

// =====================================================================================
// attention core: one wavefront per (bundle of windows, head); d_head = 16.  Attention is block
// diagonal inside a bundle (a token attends to the tokens of its own window); tile pairs that no
// window spans are skipped.
// =====================================================================================
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int kDh = 16;
constexpr int kMaxT = 144;                 // 12 x 12 window
constexpr int kMaxTiles = kMaxT / 16;      // 9

__device__ __forceinline__ unsigned short f2bf(float f) {      // v_cvt_pk_bf16_f32: round to nearest even
    union { __bf16 h; unsigned short u; } c;
    c.h = (__bf16)f;
    return c.u;
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned int)h) << 16); }

__device__ __forceinline__ f32x4 mfma16(bf16x4 a, bf16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}

// Token contractions (P V, dS K, P^T dO, dS^T Q) run over the tokens of a window: two 16-token tiles make one
// v_mfma_f32_16x16x32_bf16.  The k slots of lane group g are [tile A tokens 4g..4g+3 | tile B tokens 4g..4g+3]: exactly
// what the lane already holds from the two tiles' S^T accumulators (A operand) and what two transposing LDS reads
// deliver (B operand) -- the contraction does not care about the order of its index as long as both operands agree.
// Only Q K^T and dO V^T contract over d_head = 16 and stay 16x16x16.  An absent tile B contributes zeros on BOTH sides
// (stale LDS rows may hold anything, and 0 * NaN is not 0).
// Used by the FORWARD kernel (P V: decoder-size launches 28.1 -> 25.3 us, encoder-size 8.3 -> 8.0).  The backward
// kernel's three token contractions (dS K, P^T dO, dS^T Q) were built the same way and measured: 24.1 instead of
// 21.5 us per launch -- holding two tiles' P / dS fragments costs 22 registers (70 -> 92 VGPRs, 7 -> 5 waves per SIMD)
// in a kernel that waits on loads 70 % of its time and issues MFMAs 5 % of it; it keeps one K = 16 MFMA per tile.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_a;
__device__ __forceinline__ f32x4 mfma32_pair(bf16x4 a0, bf16x4 a1, bf16x4 b0, bf16x4 b1, f32x4 c) {
    union { struct { bf16x4 lo, hi; } p; bf16x8_a v; } fa, fb;
    fa.p.lo = a0; fa.p.hi = a1;
    fb.p.lo = b0; fb.p.hi = b1;
    const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.v, fb.v, c, 0, 0, 0);
    // Precaution: both operands stay alive past the instruction, so the register allocator cannot place the RESULT on
    // top of an operand that dies here (it did, with a constant-zero accumulator: v_mfma_f32_16x16x32_bf16 v[42:45],
    // v[42:45], v[46:49], 0).  The wrong tiles seen with that build were traced to the predicated transposing LDS read
    // (keep_if below), not to this form; the constraint costs nothing and stays.
    asm volatile("" ::"v"(fa.v), "v"(fb.v));
    return d;
}

// The transposing LDS read (ds_read_b64_tr_b16) must not sit in a PREDICATED block: compiled as "s_and_saveexec;
// ds_read_b64_tr_b16; s_or exec" without a branch around it, a false condition still left stale LDS contents in the
// destination (gfx950; NaN rows in windows with an odd tile count, reproduced by polluting LDS with NaNs,
// tools/archive/dbg_attn.py): 0 * NaN in the absent half of a tile pair.  Two guards: the tile ranges live in SGPRs
// (tile_range), so the conditions around these reads are scalar branches that skip the block; and the operand of an
// absent tile passes through this select, whatever the read left behind.
__device__ __forceinline__ bf16x4 keep_if(bool cond, bf16x4 v) {
    const bf16x4 z = {0, 0, 0, 0};
    return cond ? v : z;
}

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
// phase timing (tools/archive/phase_timing_attn.py; a no-op in the product build)
#ifdef GEOMAE_PHASE_TIMING
static __device__ unsigned long long attn_stamps[512 * 16];
#define ATTN_STAMP(i)                                                                                   \
    do {                                                                                                \
        if (threadIdx.x == 0 && blockIdx.x < 512) attn_stamps[blockIdx.x * 16 + (i)] = clock64();       \
    } while (0)
#else
#define ATTN_STAMP(i) do {} while (0)
#endif
constexpr int kAttnBlk = 512;                           // 8 waves share one (bundle, head): query / key tiles round-robin (4 waves: +0.8 % step time, 10: +2 %)
constexpr int kStageIters = (kMaxT * 2 + kAttnBlk - 1) / kAttnBlk;   // 16-byte pieces per thread per [T,16] head slice (2)

// Head slice [T, 16] of a [n, ld] bf16 matrix (columns col0..col0+15): lane handles (row, half) pieces.
// Loads for ALL slices are issued before any LDS write so that their latencies overlap.
// element offset of (token, column) in a [n, ld] bf16 matrix: row-major, or the layer stacks' tile-blocked layout
// [n/16][ld/16][16 tokens][16 channels] (sst_device.h "Row layouts"), where a head slice is one 32-byte block row
__device__ __forceinline__ int64_t tok_elem(int tok, int ld, int col, bool blk) {
    return blk ? ((int64_t)(tok >> 4) * (ld >> 4) + (col >> 4)) * 256 + (tok & 15) * 16 + (col & 15)
               : (int64_t)tok * ld + col;
}

__device__ __forceinline__ void stage_load(const unsigned short* __restrict__ src, int ld, int col0,
                                           const int (&tk)[kStageIters], int T, u32x4 (&v)[kStageIters], bool blk) {
#pragma unroll
    for (int k = 0; k < kStageIters; ++k) {
        const int t = k * kAttnBlk + threadIdx.x;
        const int row = t >> 1, half = t & 1;
        v[k] = u32x4{0u, 0u, 0u, 0u};
        if (row < T) v[k] = *reinterpret_cast<const u32x4*>(src + tok_elem(tk[k], ld, col0 + half * 8, blk));
    }
}

// row-major rm[Tp][16] and/or transposed tr[16][Tp]; rows in [T, Tp) are written as zeros
__device__ __forceinline__ void stage_store(const u32x4 (&v)[kStageIters], int Tp, unsigned short* rm,
                                            unsigned short* tr) {
#pragma unroll
    for (int k = 0; k < kStageIters; ++k) {
        const int t = k * kAttnBlk + threadIdx.x;
        const int row = t >> 1, half = t & 1;
        if (row < Tp) {
            if (rm) *reinterpret_cast<u32x4*>(rm + row * kDh + half * 8) = v[k];
            if (tr) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    tr[(half * 8 + 2 * e) * Tp + row] = (unsigned short)(v[k][e] & 0xffffu);
                    tr[(half * 8 + 2 * e + 1) * Tp + row] = (unsigned short)(v[k][e] >> 16);
                }
            }
        }
    }
}

// rows of an LDS tile rm[Tp][16] -> global [n, ld] at columns col0.. (16-byte stores)
__device__ __forceinline__ void unstage_store(const unsigned short* rm, unsigned short* __restrict__ dst, int ld,
                                              int col0, const int* toks, int T, bool blk) {
#pragma unroll
    for (int k = 0; k < kStageIters; ++k) {
        const int t = k * kAttnBlk + threadIdx.x;
        const int row = t >> 1, half = t & 1;
        if (row < T)
            *reinterpret_cast<u32x4*>(dst + tok_elem(toks[row], ld, col0 + half * 8, blk)) =
                *reinterpret_cast<const u32x4*>(rm + row * kDh + half * 8);
    }
}

__device__ __forceinline__ float dot8_bf16(const u32x4 a, const u32x4 b) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        s += __uint_as_float(a[e] << 16) * __uint_as_float(b[e] << 16);
        s += __uint_as_float(a[e] & 0xffff0000u) * __uint_as_float(b[e] & 0xffff0000u);
    }
    return s;
}

__device__ __forceinline__ bf16x4 lds4(const unsigned short* p) { return *reinterpret_cast<const bf16x4*>(p); }

// B operand (k = 4 consecutive tokens row0 .. row0+3, n = channel c = lane & 15) of a token contraction, read
// straight from a ROW-MAJOR [T][16] head slice with ds_read_b64_tr_b16: lane m of a 16-lane group points at row
// row0 + (m >> 2), channel chunk 4 * (m & 3); it receives channel m of the four rows (lane mapping measured in
// tools/archive/microbench_ds_read_tr.hip).  Replaces the transposed LDS copies (Q^T, K^T, dO^T, V^T) that cost
// 8 two-byte LDS stores per 16-byte piece.
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__device__ __forceinline__ bf16x4 lds4_tr(const unsigned short* rm, int row0, int c) {
    const unsigned short* p = rm + (row0 + (c >> 2)) * kDh + 4 * (c & 3);
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
}

struct BundleCtx {
    int s0, T, nt, Tp;
};

// Loads the bundle's token list and, per token, its window (CSR index) and that window's position range into LDS.
// From the CSR arrays this is a chain of four dependent global loads (bun_start -> win_start -> win_tokens ->
// tok_win, then win_start again per tile) in front of the operand gather -- most of a workgroup's life at d_head =
// 16.  The optional attention plan written by the window build (bun_tok [NB+1], pos_info [n] = (token, in-window
// position, window start, window end) per position) makes it two: bun_tok[b], then one 16-byte record per position.
struct AttnPlan {
    const int32_t *bun_start, *win_start, *win_tokens, *tok_win;     // CSR form (always valid)
    const int32_t* bun_tok; const int4* pos_info;                    // plan form, or null
};
__device__ __forceinline__ BundleCtx bundle_setup(int b, const AttnPlan& P, int* toks, int* wid, int* wlo, int* whi) {
    BundleCtx c;
    if (P.pos_info) {
        c.s0 = P.bun_tok[b];
        c.T = P.bun_tok[b + 1] - c.s0;
    } else {
        c.s0 = P.win_start[P.bun_start[b]];
        c.T = P.win_start[P.bun_start[b + 1]] - c.s0;
    }
    c.nt = (c.T + 15) >> 4;
    c.Tp = c.nt * 16;
    for (int t = threadIdx.x; t < c.Tp; t += kAttnBlk) {
        int4 pi = make_int4(-1, -1 - t, 0, 0);            // padded rows: a window id nothing else has
        if (t < c.T) {
            if (P.pos_info) {
                pi = P.pos_info[c.s0 + t];
            } else {
                pi.x = P.win_tokens[c.s0 + t];
                pi.y = P.tok_win[pi.x];
                pi.z = P.win_start[pi.y];
                pi.w = P.win_start[pi.y + 1];
            }
        }
        // (plan form: a window is identified by the position where it starts)
        toks[t] = pi.x; wid[t] = (P.pos_info && t < c.T) ? pi.z : pi.y; wlo[t] = pi.z; whi[t] = pi.w;
    }
    return c;
}

// key-tile range [lo, hi] that the windows touching query tile `it` span
__device__ __forceinline__ void tile_range(const BundleCtx& c, int it, const int* wlo, const int* whi, int* lo, int* hi) {
    const int first = it * 16;
    const int last = (first + 15 < c.T ? first + 15 : c.T - 1);
    // wave-uniform by construction; pinned to SGPRs so that every branch on them is a scalar branch (a block the wave
    // does not take is then SKIPPED, not run under an empty EXEC mask: see keep_if)
    *lo = __builtin_amdgcn_readfirstlane((wlo[first] - c.s0) >> 4);
    *hi = __builtin_amdgcn_readfirstlane((whi[last] - 1 - c.s0) >> 4);
}

// this thread's staging rows -> their token ids.  With the build's attention plan they come straight from global
// memory (one 16-byte record per position), so the operand gathers are issued without the LDS round trip and the
// barrier behind bundle_setup; from the CSR arrays they are read back from `toks` after that barrier.
__device__ __forceinline__ void stage_tokens(const AttnPlan& P, const BundleCtx& B, const int* toks, int (&tk)[kStageIters]) {
    if (!P.pos_info) __syncthreads();                 // workgroup-uniform
#pragma unroll
    for (int k = 0; k < kStageIters; ++k) {
        const int row = (k * kAttnBlk + threadIdx.x) >> 1;
        tk[k] = row < B.T ? (P.pos_info ? P.pos_info[B.s0 + row].x : toks[row]) : -1;
#ifdef ATTN_ABL_IDENTITY            // (timing ablation: rows addressed by bundle position -- what slot-order q / k / v would cost)
        tk[k] = row < B.T ? B.s0 + row : -1;
#endif
    }
}

// qkv: [n, 3*C] bf16 (q | k | v, C = heads*16);  out: [n, C] bf16;  lse: [n, heads] fp32
// Work item -> (bundle, head), XCD-aware.  Workgroups are dealt round-robin to the 8 XCDs (item % 8), each with its
// own L2.  A head reads 32-byte slices of the 768-byte qkv rows of its bundle's tokens, so the 8 heads of a bundle
// share every cache line: with item = bundle * heads + head they landed on 8 different XCDs and each line was
// fetched by up to 4 of them (measured: 47 MB per forward launch for 13 MB of operands, profiles/r01_pmc_traffic.json).
// Here all heads of bundle b run on XCD b % 8, in consecutive slots of that XCD.
constexpr int kXcds = 8;
__device__ __forceinline__ int attn_items(int NB, int n_heads) { return (NB + kXcds - 1) / kXcds * kXcds * n_heads; }
__device__ __forceinline__ bool attn_item(int item, int NB, int n_heads, int* b, int* h) {
    const int xcd = item % kXcds, slot = item / kXcds;
    *b = slot / n_heads * kXcds + xcd;
    *h = slot % n_heads;
    return *b < NB;
}

// H heads of one bundle per workgroup (H = 1, 2, 4; 8 / H waves per head).  H = 1 is the form for small launches (one
// round of workgroups: the encoder); at decoder size the launch is 3-4 rounds of workgroups whose life is a chain of
// dependent loads (bundle -> token records -> operand gather -> ... -> store), and H = 4 quarters the number of chains
// (2 workgroups x 4 heads per CU instead of 3 x 1).
template <int H>
__global__ __launch_bounds__(kAttnBlk, H == 1 ? 6 : (H == 2 ? 5 : 4)) void win_attn_fwd_kernel(const unsigned short* __restrict__ qkv, int n_heads,
                                                          const int32_t* __restrict__ win_start,
                                                          const int32_t* __restrict__ win_tokens,
                                                          const int32_t* __restrict__ tok_win,
                                                          const int32_t* __restrict__ bun_start,
                                                          const int32_t* __restrict__ num_bundles, float scale,
                                                          unsigned short* __restrict__ out,
                                                          float* __restrict__ lse, bool blk,
                                                          const int32_t* __restrict__ bun_tok,
                                                          const int4* __restrict__ pos_info) {
    constexpr int kSlice = kMaxT * kDh;                     // one head's [T, 16] tile
    __shared__ __attribute__((aligned(16))) unsigned short Qs_all[H * kSlice];
    __shared__ __attribute__((aligned(16))) unsigned short Ks_all[H * kSlice];
    __shared__ __attribute__((aligned(16))) unsigned short Vs_all[H * kSlice];
    __shared__ __attribute__((aligned(16))) unsigned short Os_all[H * kSlice];
    __shared__ __attribute__((aligned(16))) int toks[kMaxT], wid[kMaxT], wlo[kMaxT], whi[kMaxT];
    const AttnPlan P = {bun_start, win_start, win_tokens, tok_win, bun_tok, pos_info};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int NB = num_bundles[0];
    const int C = n_heads * kDh;
    const int groups = n_heads / H;
    const int hh = wave % H;                                 // this wave's head inside the group
    const int ho = hh * kSlice;                              // (0 when H = 1)
    for (int item = blockIdx.x; item < attn_items(NB, groups); item += gridDim.x) {
        int b, hg;
        if (!attn_item(item, NB, groups, &b, &hg)) continue;
        const int h0 = hg * H, h = h0 + hh;
        const BundleCtx B = bundle_setup(b, P, toks, wid, wlo, whi);
        const int T = B.T, nt = B.nt, Tp = B.Tp;
        {
            int tk[kStageIters];
            stage_tokens(P, B, toks, tk);
            u32x4 rq[H][kStageIters], rk[H][kStageIters], rv[H][kStageIters];
#pragma unroll
            for (int e = 0; e < H; ++e) {
                stage_load(qkv, 3 * C, (h0 + e) * kDh, tk, T, rq[e], blk);
                stage_load(qkv, 3 * C, C + (h0 + e) * kDh, tk, T, rk[e], blk);
                stage_load(qkv, 3 * C, 2 * C + (h0 + e) * kDh, tk, T, rv[e], blk);
            }
#pragma unroll
            for (int e = 0; e < H; ++e) {
                stage_store(rq[e], Tp, Qs_all + e * kSlice, nullptr);
                stage_store(rk[e], Tp, Ks_all + e * kSlice, nullptr);
                stage_store(rv[e], Tp, Vs_all + e * kSlice, nullptr);
            }
        }
        __syncthreads();
        for (int it = wave / H; it < nt; it += kAttnBlk / 64 / H) {
            int jlo, jhi;
            tile_range(B, it, wlo, whi, &jlo, &jhi);
            // S^T tiles: A = K rows (keys), B = Q^T (queries): lane holds query i = it*16 + c,
            // keys j = jt*16 + 4*g + r
            const bf16x4 qb = lds4(Qs_all + ho + (it * 16 + c) * kDh + 4 * g);
            const int wq = wid[it * 16 + c];
            f32x4 st[kMaxTiles];
            float m = -INFINITY;
#pragma unroll
            for (int jt = 0; jt < kMaxTiles; ++jt) {
                if (jt >= jlo && jt <= jhi) {
                    const bf16x4 ka = lds4(Ks_all + ho + (jt * 16 + c) * kDh + 4 * g);
                    f32x4 z = {0, 0, 0, 0};
                    st[jt] = mfma16(ka, qb, z);
                    const int4 W4 = *reinterpret_cast<const int4*>(wid + jt * 16 + 4 * g);
                    const int Wr[4] = {W4.x, W4.y, W4.z, W4.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        st[jt][r] = (Wr[r] == wq) ? st[jt][r] * scale : -INFINITY;
                        m = fmaxf(m, st[jt][r]);
                    }
                }
            }
            m = rows4_max(m);
            float sum = 0.0f;
#pragma unroll
            for (int jt = 0; jt < kMaxTiles; ++jt) {
                if (jt >= jlo && jt <= jhi) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = __expf(st[jt][r] - m);
                        st[jt][r] = p;
                        sum += p;
                    }
                }
            }
            sum = rows4_sum(sum);
            // O tile = P V : A = P (row i = c, k = key), B = V (k = key, col = d); two key tiles per K = 32 MFMA
            f32x4 o = {0, 0, 0, 0};
#pragma unroll
            for (int jp = 0; jp < (kMaxTiles + 1) / 2; ++jp) {
                const int j0 = 2 * jp, j1 = 2 * jp + 1;
                const bool a0 = j0 >= jlo && j0 <= jhi;
                const bool a1 = j1 < kMaxTiles && j1 >= jlo && j1 <= jhi;
                if (a0 || a1) {
                    const bf16x4 zero4 = {0, 0, 0, 0};
                    bf16x4 pa0 = zero4, pa1 = zero4, vb0 = zero4, vb1 = zero4;
                    if (a0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) pa0[r] = (short)f2bf(st[j0][r]);
                        vb0 = lds4_tr(Vs_all + ho, j0 * 16 + 4 * g, c);
                    }
                    if (a1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) pa1[r] = (short)f2bf(st[j1 < kMaxTiles ? j1 : j0][r]);
                        vb1 = lds4_tr(Vs_all + ho, j1 * 16 + 4 * g, c);
                    }
                    o = mfma32_pair(pa0, pa1, keep_if(a0, vb0), keep_if(a1, vb1), o);
                }
            }
            // C layout: row i = 4g + r, col d = c ; the row statistics live in lane (i & 15)
            const float inv = 1.0f / sum;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = it * 16 + 4 * g + r;
                const float inv_i = __shfl(inv, 4 * g + r, 64);
                Os_all[ho + i * kDh + c] = f2bf(o[r] * inv_i);
            }
            const int iq = it * 16 + c;
            if (g == 0 && iq < T) lse[(int64_t)toks[iq] * n_heads + h] = m + __logf(sum);
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < H; ++e) unstage_store(Os_all + e * kSlice, out, C, (h0 + e) * kDh, toks, T, blk);
        __syncthreads();
    }
}

// backward: dqkv [n, 3C] bf16 from dout [n, C] bf16, qkv, out, lse.  Two passes over the tile pairs of a (bundle, head),
// their tiles dealt round-robin to the 8 waves: a query-major one (dQ: a lane holds a query) and a key-major one (dK, dV:
// a lane holds a key) -- dQ contracts over the keys and dK / dV over the queries, so with the tiles of one head spread
// over waves one of the sums would cross waves in a single pass.  Measured alternatives (round 4, tools/attn_time.py):
// one pass with dQ summed through LDS float atomics (ds_add_f32 serialises: 30 -> 180..470 us per decoder-size launch);
// one WAVE per (bundle, head) walking every pair once with the P^T / dS^T tiles turned through LDS scratch (5 MFMAs per
// pair, no barrier, no atomics -- but an eighth of the waves and a 64-thread gather: 75 us).  Round 4 keeps the two passes
// and computes all three products TRANSPOSED, so that they come out in the T-layout (lane = token, 4 channels per lane
// group) and leave as 8-byte row pieces straight from the registers: the two LDS output tiles, two of the three barriers
// and the three un-stage passes are gone.
__global__ __launch_bounds__(kAttnBlk) void win_attn_bwd_kernel(const unsigned short* __restrict__ qkv,
                                                          const unsigned short* __restrict__ out,
                                                          const unsigned short* __restrict__ dout,
                                                          const float* __restrict__ lse, int n_heads,
                                                          const int32_t* __restrict__ win_start,
                                                          const int32_t* __restrict__ win_tokens,
                                                          const int32_t* __restrict__ tok_win,
                                                          const int32_t* __restrict__ bun_start,
                                                          const int32_t* __restrict__ num_bundles, float scale,
                                                          unsigned short* __restrict__ dqkv, bool blk,
                                                          const int32_t* __restrict__ bun_tok,
                                                          const int4* __restrict__ pos_info) {
    __shared__ __attribute__((aligned(16))) unsigned short Qs[kMaxT * kDh], Ks[kMaxT * kDh], Vs[kMaxT * kDh],
        dOs[kMaxT * kDh];
    __shared__ __attribute__((aligned(16))) float Ls[kMaxT], Ds[kMaxT];
    __shared__ __attribute__((aligned(16))) int toks[kMaxT], wid[kMaxT], wlo[kMaxT], whi[kMaxT];
    const AttnPlan P = {bun_start, win_start, win_tokens, tok_win, bun_tok, pos_info};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int NB = num_bundles[0];
    const int C = n_heads * kDh;
    for (int item = blockIdx.x; item < attn_items(NB, n_heads); item += gridDim.x) {
        ATTN_STAMP(0);
        int b, h;
        if (!attn_item(item, NB, n_heads, &b, &h)) continue;
        const BundleCtx B = bundle_setup(b, P, toks, wid, wlo, whi);
        const int T = B.T, nt = B.nt, Tp = B.Tp;
        ATTN_STAMP(1);
        {
            int tk[kStageIters];
            stage_tokens(P, B, toks, tk);
            u32x4 rq[kStageIters], rk[kStageIters], rv[kStageIters], rdo[kStageIters], ro[kStageIters];
            stage_load(qkv, 3 * C, h * kDh, tk, T, rq, blk);
            stage_load(qkv, 3 * C, C + h * kDh, tk, T, rk, blk);
            stage_load(qkv, 3 * C, 2 * C + h * kDh, tk, T, rv, blk);
            stage_load(dout, C, h * kDh, tk, T, rdo, blk);
            stage_load(out, C, h * kDh, tk, T, ro, blk);
            stage_store(rq, Tp, Qs, nullptr);
            stage_store(rk, Tp, Ks, nullptr);
            stage_store(rv, Tp, Vs, nullptr);
            stage_store(rdo, Tp, dOs, nullptr);
            // delta_i = sum_d dO[i,d] * O[i,d] (two 8-wide halves per row, adjacent lanes) ; L_i
#pragma unroll
            for (int k = 0; k < kStageIters; ++k) {
                const int t = k * kAttnBlk + threadIdx.x;
                const int row = t >> 1;
                float d = dot8_bf16(rdo[k], ro[k]);
                d += dpp_mov<kDppXor1>(d);
                if ((t & 1) == 0 && row < Tp) {
                    Ds[row] = d;                                                    // zero for padded rows
                    Ls[row] = row < T ? lse[(int64_t)tk[k] * n_heads + h] : INFINITY;       // P = exp(s - inf) = 0
                }
            }
        }
        __syncthreads();
        ATTN_STAMP(2);
        // ---- pass 1: dQ.  S^T orientation: lane holds query i = it*16 + c, keys j = jt*16 + 4g + r
        for (int it = wave; it < nt; it += kAttnBlk / 64) {
            int jlo, jhi;
            tile_range(B, it, wlo, whi, &jlo, &jhi);
            const bf16x4 qb = lds4(Qs + (it * 16 + c) * kDh + 4 * g);
            const bf16x4 dob = lds4(dOs + (it * 16 + c) * kDh + 4 * g);
            const float Li = Ls[it * 16 + c], Di = Ds[it * 16 + c];
            const int wq = wid[it * 16 + c];
            f32x4 dq = {0, 0, 0, 0};
            for (int jt = jlo; jt <= jhi; ++jt) {
                const bf16x4 ka = lds4(Ks + (jt * 16 + c) * kDh + 4 * g);
                const bf16x4 va = lds4(Vs + (jt * 16 + c) * kDh + 4 * g);
                f32x4 z = {0, 0, 0, 0};
                const f32x4 s = mfma16(ka, qb, z);       // [j][i]
                const f32x4 dp = mfma16(va, dob, z);     // dP^T[j][i] = sum_d V[j,d] dO[i,d]
                bf16x4 dsa;
                const int4 W4 = *reinterpret_cast<const int4*>(wid + jt * 16 + 4 * g);
                const int Wr[4] = {W4.x, W4.y, W4.z, W4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = (Wr[r] == wq) ? __expf(s[r] * scale - Li) : 0.0f;
                    dsa[r] = (short)f2bf(p * (dp[r] - Di) * scale);
                }
                // dQ^T[d][i] += sum_j K^T[d][j] dS^T[j][i]: A = K^T (lane = channel d: K[4g..4g+3][d], the transposing read),
                // B = this lane's registers -> rows d = 4g + r, column i = c: the T-layout of query i (computed as
                // dQ = dS K it came out [queries 4g + r][channel c] and went through an LDS tile, a barrier and an
                // un-stage pass to reach memory in row pieces)
                const bf16x4 kt = lds4_tr(Ks, jt * 16 + 4 * g, c);
                dq = mfma16(kt, dsa, dq);
            }
            if (it * 16 + c < T) {
                union { bf16x4 v; uint2 u; } pq;
#pragma unroll
                for (int r = 0; r < 4; ++r) pq.v[r] = (short)f2bf(dq[r]);
                *reinterpret_cast<uint2*>(dqkv + tok_elem(toks[it * 16 + c], 3 * C, h * kDh + 4 * g, blk)) = pq.u;
            }
        }
        ATTN_STAMP(3);
        ATTN_STAMP(4);
        // ---- pass 2: dK, dV.  S orientation: lane holds key j = jt*16 + c, queries i = it*16 + 4g + r
        for (int jt = wave; jt < nt; jt += kAttnBlk / 64) {
            int ilo, ihi;
            tile_range(B, jt, wlo, whi, &ilo, &ihi);
            const bf16x4 kb = lds4(Ks + (jt * 16 + c) * kDh + 4 * g);
            const bf16x4 vb = lds4(Vs + (jt * 16 + c) * kDh + 4 * g);
            const int wk = wid[jt * 16 + c];
            f32x4 dk = {0, 0, 0, 0}, dv = {0, 0, 0, 0};
            for (int it = ilo; it <= ihi; ++it) {
                const bf16x4 qa = lds4(Qs + (it * 16 + c) * kDh + 4 * g);
                const bf16x4 doa = lds4(dOs + (it * 16 + c) * kDh + 4 * g);
                f32x4 z = {0, 0, 0, 0};
                const f32x4 s = mfma16(qa, kb, z);       // [i][j]
                const f32x4 dp = mfma16(doa, vb, z);     // dP[i][j]
                bf16x4 pa, dsa;
                const int i0 = it * 16 + 4 * g;                       // 4 consecutive queries: one 16-byte LDS read each
                const float4 L4 = *reinterpret_cast<const float4*>(Ls + i0), D4 = *reinterpret_cast<const float4*>(Ds + i0);
                const int4 W4 = *reinterpret_cast<const int4*>(wid + i0);
                const float Lr[4] = {L4.x, L4.y, L4.z, L4.w}, Dr[4] = {D4.x, D4.y, D4.z, D4.w};
                const int Wr[4] = {W4.x, W4.y, W4.z, W4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = (Wr[r] == wk) ? __expf(s[r] * scale - Lr[r]) : 0.0f;
                    pa[r] = (short)f2bf(p);
                    dsa[r] = (short)f2bf(p * (dp[r] - Dr[r]) * scale);
                }
                // dV^T[d][j] += sum_i dO^T[d][i] P[i][j] ; dK^T[d][j] += sum_i Q^T[d][i] dS[i][j]: A = the transposing read of the
                // row tile (lane = channel d), B = this lane's registers (lane = key j) -> the T-layout of key j
                const bf16x4 dot = lds4_tr(dOs, it * 16 + 4 * g, c);
                dv = mfma16(dot, pa, dv);
                const bf16x4 qt = lds4_tr(Qs, it * 16 + 4 * g, c);
                dk = mfma16(qt, dsa, dk);
            }
            if (jt * 16 + c < T) {
                const int tkj = toks[jt * 16 + c];
                union { bf16x4 v; uint2 u; } pk, pv;
#pragma unroll
                for (int r = 0; r < 4; ++r) { pk.v[r] = (short)f2bf(dk[r]); pv.v[r] = (short)f2bf(dv[r]); }
                *reinterpret_cast<uint2*>(dqkv + tok_elem(tkj, 3 * C, C + h * kDh + 4 * g, blk)) = pk.u;
                *reinterpret_cast<uint2*>(dqkv + tok_elem(tkj, 3 * C, 2 * C + h * kDh + 4 * g, blk)) = pv.u;
            }
        }
        ATTN_STAMP(5);
        __syncthreads();                                   // (the next item's staging overwrites the tiles)
        ATTN_STAMP(6);
    }
}

// ---- SSTInputLayer voxel drop (fine-tune path): keep a voxel iff its arrival rank inside its
// window is below the max_tokens of the window's drop level (level = the range (lower, upper] its token count
// falls in; mmdet3d/models/middle_encoders/sst_input_layer.py:213-238 drop_single_shift).  The reference ranks by a
// sort after a random shuffle of the voxels, i.e. which voxels survive in an over-full window is random; here
// the rank is the atomic arrival order.
struct DropLevels { int n; int max_tokens[8]; int lower[8]; int upper[8]; };
__global__ __launch_bounds__(kWBlk) void win_drop_kernel(int n, const int32_t* __restrict__ table,
                                                         const int32_t* __restrict__ rank,
                                                         const int32_t* __restrict__ tok_win_id, DropLevels L,
                                                         uint8_t* __restrict__ keep, int32_t* __restrict__ level) {
    for (int i = blockIdx.x * kWBlk + threadIdx.x; i < n; i += gridDim.x * kWBlk) {
        const int cnt = table[tok_win_id[i]];
        int l = L.n - 1;                      // a count above the last range is treated as the last level
        for (int k = 0; k < L.n; ++k)
            if (cnt > L.lower[k] && cnt <= L.upper[k]) { l = k; break; }
        keep[i] = rank[i] < L.max_tokens[l];
        if (level) level[i] = l;
    }
}

// ---- recover_bev (sst_second_pretrained_v1.py:243-280): scatter token rows into a dense BEV canvas.  The canvas
// is stored channels-last ([B, ny, nx, C] in memory, i.e. an NCHW tensor in torch's channels_last format): a
// token is one coalesced C*4-byte row, and MIOpen's convolutions take the layout as is.
__global__ __launch_bounds__(256) void bev_scatter_kernel(const float* __restrict__ feat, const int4* __restrict__ coors,
                                                          int64_t n, int C, int ny, int nx, float* __restrict__ canvas) {
    const int64_t total = n * (C / 4);
    for (int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t i = t / (C / 4);
        const int c4 = (int)(t - i * (C / 4));
        const int4 c = coors[i];
        reinterpret_cast<float4*>(canvas + (((int64_t)c.x * ny + c.z) * nx + c.w) * C)[c4] =
            reinterpret_cast<const float4*>(feat + i * C)[c4];
    }
}
__global__ __launch_bounds__(256) void bev_gather_kernel(const float* __restrict__ canvas, const int4* __restrict__ coors,
                                                         int64_t n, int C, int ny, int nx, float* __restrict__ out) {
    const int64_t total = n * (C / 4);
    for (int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t i = t / (C / 4);
        const int c4 = (int)(t - i * (C / 4));
        const int4 c = coors[i];
        reinterpret_cast<float4*>(out + i * C)[c4] =
            reinterpret_cast<const float4*>(canvas + (((int64_t)c.x * ny + c.z) * nx + c.w) * C)[c4];
    }
}

}  // namespace geomae

using namespace geomae;

#ifdef GEOMAE_PHASE_TIMING
extern "C" int geomae_debug_read_attn_stamps(unsigned long long* host) {
    hipDeviceSynchronize();
    static unsigned long long zeros[512 * 16];
    hipMemcpyFromSymbol(host, HIP_SYMBOL(attn_stamps), sizeof(zeros));
    hipMemcpyToSymbol(HIP_SYMBOL(attn_stamps), zeros, sizeof(zeros));
    return 0;
}
#endif

// Bundle size of the SECOND packing of a layout (fbun_tok: the one-launch layer kernel's work items, sst_fused.hip; the
// attention kernels keep whole-window bundles).  That kernel runs one workgroup per bundle whose dependent chain grows with
// the bundle, on a 256-CU chip: small token sets want many small bundles.  GeomaeTuning.bundle_cap overrides.
extern "C" int32_t geomae_window_bundle_cap(int32_t num_tokens, int32_t max_window_tokens) {
    const GeomaeTuning& tn = tuning();
    // <= fused_max_tokens (12288: the token sets the one-bundle-per-workgroup kernel takes, sst_stack.hip): three 16-token tiles
    // per bundle (measured at 6.6 k tokens: 26.8 / 25.3 / 27.8 us per layer at caps 40 / 48 / 56); above: the weight-stationary
    // kernel's work items (sst_ws.hip: the weights stay in registers, so a bundle only has to amortise its six barriers --
    // GeomaeTuning.ws_bundle_cap), or whole windows where that form is off
    int cap = tn.bundle_cap > 0 ? tn.bundle_cap
            : (num_tokens <= tn.fused_max_tokens ? 48 : (tn.ws_layers ? tn.ws_bundle_cap : max_window_tokens));
    if (cap < 16) cap = 16;
    if (cap > max_window_tokens) cap = max_window_tokens;
    return cap;
}

static int win_geom(const GeomaeWindowConfig* cfg, int shift_index, WinGeom* g, int* slots_per_sample);

// How many KEPT pillars the fullest window of each of the two layouts (unshifted / shifted) holds: one workgroup, the window
// tables of both shifts in LDS.  The step engine runs this with the random mask, a step ahead of the encoder that will pack
// these pillars into bundles, and reads the two numbers back with the pillar counts: the one-launch layer's second kernel
// (sst_ws.hip sst_layer_fwd_ws_kernel with min_tiles = 5: bundles of more than 64 positions = a window that kept more than 64 pillars) is
// only launched for a layout that has such a window.
constexpr int kMaxKeepSlots = 12288;             // window slots of BOTH shifts that fit the LDS tables (48 KB)
__global__ __launch_bounds__(1024) void win_max_keep_kernel(const int32_t* __restrict__ ids_keep, const int32_t* __restrict__ counts,
                                                            const int4* __restrict__ voxel_coors, WinGeom g0, WinGeom g1, int slots,
                                                            int32_t* __restrict__ out /* pinned host memory: [2] */) {
    __shared__ int table[kMaxKeepSlots];
    __shared__ int best[2];
    for (int i = threadIdx.x; i < 2 * slots; i += 1024) table[i] = 0;
    if (threadIdx.x < 2) best[threadIdx.x] = 0;
    __syncthreads();
    const int nk = counts[0];
    for (int i = threadIdx.x; i < nk; i += 1024) {
        const int4 c = voxel_coors[ids_keep[i]];
        int w, p;
        win_of(c, g0, &w, &p);
        atomicAdd(&table[w], 1);
        win_of(c, g1, &w, &p);
        atomicAdd(&table[slots + w], 1);
    }
    __syncthreads();
    int m0 = 0, m1 = 0;
    for (int i = threadIdx.x; i < slots; i += 1024) { m0 = max(m0, table[i]); m1 = max(m1, table[slots + i]); }
    atomicMax(&best[0], m0);
    atomicMax(&best[1], m1);
    __syncthreads();
    if (threadIdx.x < 2) out[threadIdx.x] = best[threadIdx.x];
}

// -> out[0], out[1] (device-visible host memory): the largest number of kept pillars in one window, unshifted / shifted
// layout; -1, -1 when the window tables do not fit (the caller then assumes large windows exist)
namespace geomae {
int window_max_keep(const int32_t* ids_keep, const int32_t* counts, const int32_t* voxel_coors, int batch_size,
                    const GeomaeWindowConfig* cfg, int32_t* out, hipStream_t stream);
}
int geomae::window_max_keep(const int32_t* ids_keep, const int32_t* counts, const int32_t* voxel_coors, int batch_size,
                            const GeomaeWindowConfig* cfg, int32_t* out, hipStream_t stream) {
    WinGeom g0, g1;
    int sps = 0;
    int rc = win_geom(cfg, 0, &g0, &sps);
    if (rc) return rc;
    if ((rc = win_geom(cfg, 1, &g1, &sps))) return rc;
    const int slots = batch_size * sps;
    if (2 * slots > kMaxKeepSlots) {
        out[0] = out[1] = -1;                    // (host write: the caller reads it behind its own event anyway)
        return GEOMAE_OK;
    }
    hipLaunchKernelGGL(win_max_keep_kernel, dim3(1), dim3(1024), 0, stream, ids_keep, counts, (const int4*)voxel_coors, g0, g1, slots, out);
    return check_launch("win_max_keep_kernel");
}

static int win_geom(const GeomaeWindowConfig* cfg, int shift_index, WinGeom* g, int* slots_per_sample) {
    GEOMAE_REQUIRE(cfg, "window: null config");
    GEOMAE_REQUIRE(shift_index == 0 || shift_index == 1, "window: shift_index must be 0 or 1");
    GEOMAE_REQUIRE(cfg->window_shape[0] >= 1 && cfg->window_shape[1] >= 1, "window: bad window_shape");
    g->wx = cfg->window_shape[0];
    g->wy = cfg->window_shape[1];
    // bb.py:637-642: ceil(bev / win) + 1 windows per axis ("plus one to meet the needs of shift")
    g->nwx = (cfg->bev_shape[0] + g->wx - 1) / g->wx + 1;
    g->nwy = (cfg->bev_shape[1] + g->wy - 1) / g->wy + 1;
    g->shift_x = shift_index ? cfg->shift[0] : 0;
    g->shift_y = shift_index ? cfg->shift[1] : 0;
    *slots_per_sample = g->nwx * g->nwy;
    return GEOMAE_OK;
}

extern "C" int64_t geomae_window_build_workspace_bytes(int32_t num_tokens, int32_t batch_size,
                                                       const GeomaeWindowConfig* cfg) {
    WinGeom g;
    int sps;
    if (win_geom(cfg, 0, &g, &sps)) return -1;
    auto al = [](int64_t b) { return (b + 255) / 256 * 256; };
    return al((int64_t)batch_size * sps * 4) + al((int64_t)num_tokens * 4);
}

extern "C" int64_t geomae_window_build_batch_workspace_bytes(const int32_t* num_tokens, int32_t num_jobs,
                                                             int32_t batch_size, const GeomaeWindowConfig* cfg) {
    if (!num_tokens || num_jobs < 1 || num_jobs > kMaxWinJobs) return -1;
    int64_t total = 0;
    for (int k = 0; k < num_jobs; ++k) {
        const int64_t b = geomae_window_build_workspace_bytes(num_tokens[k], batch_size, cfg);
        if (b < 0) return -1;
        total += b;
    }
    return total;
}

static int device_cus() {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        return n > 0 ? n : 256;
    }();
    return cus;
}

extern "C" int64_t geomae_window_build_batch_table_bytes(const int32_t* num_tokens, int32_t num_jobs, int32_t batch_size,
                                                         const GeomaeWindowConfig* cfg) {
    if (!num_tokens || num_jobs < 1 || num_jobs > kMaxWinJobs) return -1;
    WinGeom g;
    int sps;
    if (win_geom(cfg, 0, &g, &sps)) return -1;
    return (int64_t)num_jobs * (((int64_t)batch_size * sps * 4 + 255) / 256 * 256);
}

extern "C" int geomae_window_build_batch(const GeomaeWindowBuildJob* jobs, int32_t num_jobs, int32_t batch_size,
                                         const GeomaeWindowConfig* cfg, void* workspace, int64_t workspace_bytes,
                                         hipStream_t stream) {
    GEOMAE_REQUIRE(jobs && num_jobs >= 1 && num_jobs <= kMaxWinJobs && batch_size >= 1,
                   "window_build_batch: 1..4 jobs, batch_size >= 1");
    auto al = [](int64_t b) { return (b + 255) / 256 * 256; };
    WinJobs J;
    int64_t table_bytes = 0, need = 0;
    int max_n = 0, max_w = 1;
    for (int k = 0; k < num_jobs; ++k) {
        const GeomaeWindowBuildJob& in = jobs[k];
        WinJob& j = J.j[k];
        int sps;
        int rc = win_geom(cfg, in.shift_index, &j.g, &sps);
        if (rc) return rc;
        GEOMAE_REQUIRE(in.num_tokens >= 0, "window_build: bad sizes");
        GEOMAE_REQUIRE(in.win_start && in.num_windows && in.bun_start && in.num_bundles, "window_build: null output");
        GEOMAE_REQUIRE(in.num_tokens == 0 || (in.coors && in.win_tokens && in.tok_win && in.tok_pos),
                       "window_build: null argument");
        j.coors = (const int4*)in.coors; j.n = in.num_tokens; j.slots = batch_size * sps; j.cap = j.g.wx * j.g.wy;
        j.fcap = geomae_window_bundle_cap(in.num_tokens, j.cap);
        GEOMAE_REQUIRE((in.fbun_tok == nullptr) == (in.num_fbundles == nullptr), "window_build: pass both arrays of the second packing or none");
        GEOMAE_REQUIRE(!in.fbun_tok || in.pos_info, "window_build: the second packing needs the attention plan");
        j.fbun_tok = in.fbun_tok; j.num_fbundles = in.num_fbundles;
        GEOMAE_REQUIRE((in.fitems == nullptr) == (in.num_fitems == nullptr), "window_build: pass both arrays of the forward work items or none");
        GEOMAE_REQUIRE(!in.fitems || in.fbun_tok, "window_build: the forward work items need the second packing");
        j.fitems = (int4*)in.fitems; j.num_fitems = in.num_fitems;
        // (only the token sets the one-launch forward takes get items: the others' layers never read them)
        j.icap = (in.fitems && in.num_tokens <= tuning().fused_max_tokens) ? tuning().fwd_item_cap : 0;
        j.ibudget = device_cus();
        j.win_start = in.win_start; j.win_tokens = in.win_tokens; j.tok_win = in.tok_win; j.tok_pos = in.tok_pos;
        j.num_windows = in.num_windows; j.bun_start = in.bun_start; j.num_bundles = in.num_bundles;
        GEOMAE_REQUIRE((in.bun_tok == nullptr) == (in.pos_info == nullptr), "window_build: pass both attention-plan arrays or none");
        j.bun_tok = in.bun_tok; j.pos_info = (int4*)in.pos_info;
        table_bytes += al((int64_t)j.slots * 4);
        need += al((int64_t)j.slots * 4) + al((int64_t)j.n * 4);
        if (j.n > max_n) max_n = j.n;
        const int w = j.n < j.slots ? j.n : j.slots;
        if (w > max_w) max_w = w;
    }
    if (workspace_bytes < need || !workspace) {
        set_error("window_build: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
        return GEOMAE_ERR_WORKSPACE;
    }
    // workspace = [table_0 | table_1 | ...][rank_0 | rank_1 | ...]: the window tables are zeroed by one memset
    char* tp = (char*)workspace;
    char* rp = (char*)workspace + table_bytes;
    for (int k = 0; k < num_jobs; ++k) {
        J.j[k].table = (int32_t*)tp; tp += al((int64_t)J.j[k].slots * 4);
        J.j[k].rank = (int32_t*)rp;  rp += al((int64_t)J.j[k].n * 4);
    }
    for (int k = num_jobs; k < kMaxWinJobs; ++k) J.j[k] = J.j[0];
    if (!window_tables_prezeroed()) GEOMAE_HIP(hipMemsetAsync(workspace, 0, (size_t)table_bytes, stream));
    const dim3 tok_grid(stream_grid(max_n > 0 ? max_n : 1, kWBlk), num_jobs);
    if (max_n > 0) hipLaunchKernelGGL(win_hist_jobs_kernel, tok_grid, dim3(kWBlk), 0, stream, J);
    hipLaunchKernelGGL(win_scan_jobs_kernel, dim3(1, num_jobs), dim3(1024), 0, stream, J);
    if (max_n > 0) {
        hipLaunchKernelGGL(win_place_jobs_kernel, tok_grid, dim3(kWBlk), 0, stream, J);
        hipLaunchKernelGGL(win_sort_jobs_kernel, dim3(max_w < 4096 ? max_w : 4096, num_jobs), dim3(64), 0, stream, J);
    }
    hipLaunchKernelGGL(win_bundle_jobs_kernel, dim3(1, num_jobs), dim3(1024), 0, stream, J);
    return check_launch("window_build");
}

extern "C" int geomae_window_build(const int32_t* coors, int32_t num_tokens, int32_t batch_size,
                                   const GeomaeWindowConfig* cfg, int32_t shift_index, int32_t* win_start,
                                   int32_t* win_tokens, int32_t* tok_win, int32_t* tok_pos,
                                   int32_t* num_windows, int32_t* bun_start, int32_t* num_bundles,
                                   void* workspace, int64_t workspace_bytes, hipStream_t stream) {
    GeomaeWindowBuildJob job = {coors, num_tokens, shift_index, win_start, win_tokens, tok_win, tok_pos,
                                num_windows, bun_start, num_bundles, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    return geomae_window_build_batch(&job, 1, batch_size, cfg, workspace, workspace_bytes, stream);
}

// Workgroups to launch: one per (bundle, head).  The bundle count lives on the device; its bound from the
// greedy packing (two consecutive bundles always hold more than `cap` tokens, win_bundle_kernel) is
// 2 n / cap + 1, far below max_bundles = min(n, window slots) -- the old grid of 4096 was mostly workgroups
// that only read num_bundles and left.
static int attn_grid(int num_tokens, int num_heads, int max_bundles, int cap) {
    int64_t nb = 2 * (int64_t)num_tokens / (cap > 0 ? cap : 1) + 2;
    if (nb > max_bundles) nb = max_bundles;
    const int64_t items = (nb + 7) / 8 * 8 * num_heads;               // a multiple of 8: item % 8 (the XCD) is loop-invariant
    return (int)(items < 256 * 16 ? items : 256 * 16);
}

// heads per attention workgroup (see win_attn_fwd_kernel); GeomaeTuning.attn_heads = 1/2/4 forces it (A/B runs)
static int attn_heads_per_wg(int num_tokens, int num_heads) {
    const int forced = tuning().attn_heads;
    int h = forced ? forced : (num_tokens > 8192 ? 4 : 1);      // decoder forward phase 0.342 / 0.332 / 0.327 ms at 1 / 2 / 4
    while (h > 1 && num_heads % h) h >>= 1;
    return h == 4 || h == 2 ? h : 1;
}

extern "C" int geomae_window_attention_forward(const void* qkv_bf16, int32_t num_tokens, int32_t num_heads,
                                               int32_t head_dim, const int32_t* win_start,
                                               const int32_t* win_tokens, const int32_t* tok_win,
                                               const int32_t* bun_start, const int32_t* num_bundles,
                                               int32_t max_bundles, int32_t max_window_tokens, void* out_bf16,
                                               float* lse, const int32_t* bun_tok, const int32_t* pos_info,
                                               hipStream_t stream) {
    if (num_tokens <= 0 || max_bundles <= 0) return GEOMAE_OK;
    GEOMAE_REQUIRE(qkv_bf16 && win_start && win_tokens && tok_win && bun_start && num_bundles && out_bf16 && lse,
                   "window_attention_forward: null argument");
    GEOMAE_REQUIRE(head_dim == kDh, "window_attention_forward: head_dim must be %d", kDh);
    GEOMAE_REQUIRE(max_window_tokens <= kMaxT, "window_attention_forward: windows hold at most %d tokens", kMaxT);
    const int hpw = attn_heads_per_wg(num_tokens, num_heads);
    const int grid = attn_grid(num_tokens, num_heads / hpw, max_bundles, max_window_tokens);
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(kAttnBlk), 0, stream, (const unsigned short*)qkv_bf16,
                           num_heads, win_start, win_tokens, tok_win, bun_start, num_bundles,
                           1.0f / sqrtf((float)head_dim), (unsigned short*)out_bf16, lse, (layer_layout() & 1) != 0,
                           pos_info ? bun_tok : nullptr, (const int4*)(bun_tok ? pos_info : nullptr));
    };
    if (hpw == 4) launch(win_attn_fwd_kernel<4>);
    else if (hpw == 2) launch(win_attn_fwd_kernel<2>);
    else launch(win_attn_fwd_kernel<1>);
    return check_launch("win_attn_fwd_kernel");
}

extern "C" int geomae_window_attention_backward(const void* qkv_bf16, const void* out_bf16, const void* dout_bf16,
                                                const float* lse, int32_t num_tokens, int32_t num_heads,
                                                int32_t head_dim, const int32_t* win_start,
                                                const int32_t* win_tokens, const int32_t* tok_win,
                                                const int32_t* bun_start, const int32_t* num_bundles,
                                                int32_t max_bundles, int32_t max_window_tokens, void* dqkv_bf16,
                                                const int32_t* bun_tok, const int32_t* pos_info, hipStream_t stream) {
    if (num_tokens <= 0 || max_bundles <= 0) return GEOMAE_OK;
    GEOMAE_REQUIRE(qkv_bf16 && out_bf16 && dout_bf16 && lse && win_start && win_tokens && tok_win && bun_start &&
                   num_bundles && dqkv_bf16, "window_attention_backward: null argument");
    GEOMAE_REQUIRE(head_dim == kDh, "window_attention_backward: head_dim must be %d", kDh);
    GEOMAE_REQUIRE(max_window_tokens <= kMaxT, "window_attention_backward: windows hold at most %d tokens", kMaxT);
    const int grid = attn_grid(num_tokens, num_heads, max_bundles, max_window_tokens);
    hipLaunchKernelGGL(win_attn_bwd_kernel, dim3(grid), dim3(kAttnBlk), 0, stream, (const unsigned short*)qkv_bf16,
                       (const unsigned short*)out_bf16, (const unsigned short*)dout_bf16, lse, num_heads, win_start,
                       win_tokens, tok_win, bun_start, num_bundles, 1.0f / sqrtf((float)head_dim),
                       (unsigned short*)dqkv_bf16, (layer_layout() & 1) != 0, pos_info ? bun_tok : nullptr,
                       (const int4*)(bun_tok ? pos_info : nullptr));
    return check_launch("win_attn_bwd_kernel");
}

extern "C" int64_t geomae_window_drop_workspace_bytes(int32_t num_tokens, int32_t batch_size, const GeomaeWindowConfig* cfg) {
    WinGeom g;
    int sps;
    if (win_geom(cfg, 0, &g, &sps)) return -1;
    auto al = [](int64_t b) { return (b + 255) / 256 * 256; };
    return al((int64_t)batch_size * sps * 4) + 3 * al((int64_t)num_tokens * 4);
}

extern "C" int geomae_window_drop(const int32_t* coors, int32_t num_tokens, int32_t batch_size, const GeomaeWindowConfig* cfg,
                                  int32_t shift_index, int32_t num_levels, const int32_t* max_tokens,
                                  const int32_t* range_lower, const int32_t* range_upper, uint8_t* keep,
                                  int32_t* drop_level, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
    if (num_tokens <= 0) return GEOMAE_OK;
    WinGeom g;
    int sps;
    int rc = win_geom(cfg, shift_index, &g, &sps);
    if (rc) return rc;
    GEOMAE_REQUIRE(coors && keep && max_tokens && range_lower && range_upper && num_levels >= 1 && num_levels <= 8,
                   "window_drop: bad argument (1..8 drop levels)");
    const int64_t need = geomae_window_drop_workspace_bytes(num_tokens, batch_size, cfg);
    if (workspace_bytes < need || !workspace) {
        set_error("window_drop: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
        return GEOMAE_ERR_WORKSPACE;
    }
    const int slots = batch_size * sps;
    auto al = [](int64_t b) { return (b + 255) / 256 * 256; };
    char* ws = (char*)workspace;
    int32_t* table = (int32_t*)ws;  ws += al((int64_t)slots * 4);
    int32_t* rank = (int32_t*)ws;   ws += al((int64_t)num_tokens * 4);
    int32_t* wid = (int32_t*)ws;    ws += al((int64_t)num_tokens * 4);
    int32_t* pos = (int32_t*)ws;
    GEOMAE_HIP(hipMemsetAsync(table, 0, (size_t)slots * 4, stream));
    hipLaunchKernelGGL(win_hist_kernel, dim3(stream_grid(num_tokens, kWBlk)), dim3(kWBlk), 0, stream, (const int4*)coors,
                       num_tokens, g, table, rank, wid, pos);
    DropLevels L;
    L.n = num_levels;
    for (int k = 0; k < num_levels; ++k) { L.max_tokens[k] = max_tokens[k]; L.lower[k] = range_lower[k]; L.upper[k] = range_upper[k]; }
    hipLaunchKernelGGL(win_drop_kernel, dim3(stream_grid(num_tokens, kWBlk)), dim3(kWBlk), 0, stream, num_tokens, table, rank,
                       wid, L, keep, drop_level);
    return check_launch("win_drop_kernel");
}

extern "C" int geomae_recover_bev_forward(const float* feat, const int32_t* coors, int64_t num_tokens, int32_t channels,
                                          int32_t batch_size, int32_t ny, int32_t nx, float* canvas, hipStream_t stream) {
    GEOMAE_REQUIRE(canvas && channels >= 4 && channels % 4 == 0 && batch_size >= 1 && ny >= 1 && nx >= 1,
                   "recover_bev_forward: bad argument (channels must be a multiple of 4)");
    GEOMAE_HIP(hipMemsetAsync(canvas, 0, (size_t)batch_size * ny * nx * channels * sizeof(float), stream));
    if (num_tokens <= 0) return GEOMAE_OK;
    GEOMAE_REQUIRE(feat && coors, "recover_bev_forward: null argument");
    hipLaunchKernelGGL(bev_scatter_kernel, dim3(stream_grid(num_tokens * (channels / 4), 256)), dim3(256), 0, stream, feat,
                       (const int4*)coors, num_tokens, channels, ny, nx, canvas);
    return check_launch("bev_scatter_kernel");
}

extern "C" int geomae_recover_bev_backward(const float* grad_canvas, const int32_t* coors, int64_t num_tokens, int32_t channels,
                                           int32_t batch_size, int32_t ny, int32_t nx, float* grad_feat, hipStream_t stream) {
    if (num_tokens <= 0) return GEOMAE_OK;
    GEOMAE_REQUIRE(grad_canvas && coors && grad_feat && channels >= 4 && channels % 4 == 0, "recover_bev_backward: bad argument");
    hipLaunchKernelGGL(bev_gather_kernel, dim3(stream_grid(num_tokens * (channels / 4), 256)), dim3(256), 0, stream, grad_canvas,
                       (const int4*)coors, num_tokens, channels, ny, nx, grad_feat);
    return check_launch("bev_gather_kernel");
}
