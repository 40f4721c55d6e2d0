// Random token masking on device (gfx950).
//
// Reference: get_vanilla_mask_index (mmdet3d/models/detectors/multi_sub_voxel_dynamic_voxelnet_ssl.py:
// 287-304): per sample, L pillars, len_keep = int(L * (1 - ratio)), torch.randperm(L) -> first
// len_keep kept, the rest masked.  Only the SET matters downstream (attention is permutation
// equivariant, every loss is a mean over masked tokens), so instead of a permutation we draw a
// uniformly random subset of exactly len_keep pillars: counter-based 32-bit keys, a 3-pass
// LDS radix select of the len_keep-th smallest key, then an order-preserving compaction --
// ids_keep / ids_mask come out ascending, which keeps later gathers coalesced.
// One workgroup per sample; no host sync.  cuda/torch RNG streams cannot be reproduced, so
// parity tests inject ids (the Python boundary accepts them).
#include "common.h"
#include "../../include/geomae_hip.h"

namespace geomae {

constexpr int kMaskBlk = 1024;

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t mask_key(uint64_t seed, int b, int i) {
    uint32_t h = mix32((uint32_t)seed ^ mix32((uint32_t)(seed >> 32) + 0x9E3779B9U * (uint32_t)(b + 1)));
    return mix32(mix32((uint32_t)i * 0x9E3779B1U + h) ^ (h * 0x85EBCA6BU + 0x27D4EB2FU));
}

__device__ int block_scan_1024(int v, int* total, int* sm /*>=17*/) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) sm[w] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
    for (int k = 0; k < kMaskBlk / 64; ++k) {
        int s = sm[k];
        if (k < w) woff += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return woff + incl - v;
}

__global__ __launch_bounds__(kMaskBlk) void random_mask_kernel(const int32_t* __restrict__ sample_start,
                                                               int n_batch, double keep_frac, uint64_t seed,
                                                               int32_t* __restrict__ ids_keep,
                                                               int32_t* __restrict__ ids_mask,
                                                               int32_t* __restrict__ token_row,
                                                               int32_t* __restrict__ counts) {
    __shared__ int hist[4096];
    __shared__ int sm[20];
    __shared__ uint32_t s_prefix;
    __shared__ int s_need;
    const int b = blockIdx.x;
    const int p0 = sample_start[b];
    const int L = sample_start[b + 1] - p0;
    const int K = (int)((double)L * keep_frac);
    int keep_base = 0, mask_base = 0, keep_total = 0, mask_total = 0;
    for (int k = 0; k < n_batch; ++k) {
        const int l = sample_start[k + 1] - sample_start[k];
        const int kk = (int)((double)l * keep_frac);
        if (k < b) { keep_base += kk; mask_base += l - kk; }
        keep_total += kk;
        mask_total += l - kk;
    }
    if (b == 0 && threadIdx.x == 0) { counts[0] = keep_total; counts[1] = mask_total; }

    // ---- radix select: find T = K-th smallest key (0-based rank K-1); need = how many keys == T to keep
    uint32_t prefix = 0, prefix_mask = 0;
    int need = K;                       // rank still to resolve inside the current prefix bucket
    const int shifts[3] = {20, 8, 0};
    const int bits[3] = {12, 12, 8};
    if (K > 0) {
        for (int pass = 0; pass < 3; ++pass) {
            const int nb = 1 << bits[pass];
            for (int t = threadIdx.x; t < nb; t += kMaskBlk) hist[t] = 0;
            __syncthreads();
            for (int i = threadIdx.x; i < L; i += kMaskBlk) {
                const uint32_t key = mask_key(seed, b, i);
                if ((key & prefix_mask) == prefix) atomicAdd(&hist[(key >> shifts[pass]) & (nb - 1)], 1);
            }
            __syncthreads();
            {   // parallel search of the bin holding rank `need`: 4 consecutive bins per thread
                int h[4], v = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int bin = threadIdx.x * 4 + k;
                    h[k] = bin < nb ? hist[bin] : 0;
                    v += h[k];
                }
                int tot;
                int acc = block_scan_1024(v, &tot, sm);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (acc < need && acc + h[k] >= need) {
                        s_prefix = prefix | ((uint32_t)(threadIdx.x * 4 + k) << shifts[pass]);
                        s_need = need - acc;
                    }
                    acc += h[k];
                }
            }
            __syncthreads();
            prefix = s_prefix;
            need = s_need;
            prefix_mask |= (uint32_t)(nb - 1) << shifts[pass];
            __syncthreads();
        }
    }
    const uint32_t T = prefix;          // keys < T are kept, the first `need` keys == T are kept
    // ---- order-preserving compaction
    int run_keep = 0, run_eq = 0;
    for (int base = 0; base < L; base += kMaskBlk) {
        const int i = base + threadIdx.x;
        uint32_t key = 0;
        int lt = 0, eq = 0;
        if (i < L && K > 0) {
            key = mask_key(seed, b, i);
            lt = key < T;
            eq = key == T;
        }
        int tot_eq, tot_keep;
        const int eq_rank = run_eq + block_scan_1024(eq, &tot_eq, sm);
        const int keep = lt | (eq & (eq_rank < need));
        const int keep_rank = run_keep + block_scan_1024(keep, &tot_keep, sm);
        if (i < L) {
            const int p = p0 + i;
            if (keep) {
                ids_keep[keep_base + keep_rank] = p;
                token_row[p] = keep_base + keep_rank;
            } else {
                const int r = mask_base + (i - keep_rank);
                ids_mask[r] = p;
                token_row[p] = keep_total + r;
            }
        }
        run_eq += tot_eq;
        run_keep += tot_keep;
    }
}

// ---- the same draw, tokens emitted WINDOW-MAJOR.  The subset is the one random_mask_kernel draws from the same seed;
// only the ORDER of ids_keep / ids_mask differs: grouped by the (unshifted) SST window of the pillar, windows ascending,
// pillars ascending inside a window.  The token order of the SST stacks is the order of these lists, and the stacks'
// activations live in 16-token tiles: with tokens in pillar order (ascending (b, y, x)) the 16 tokens of a tile are
// spread over the 4-5 windows a row of pillars crosses, so a (bundle, head) attention workgroup used 1-2 of the 4
// token slices in every 128-byte line it fetched (27 MB per forward launch for 12.7 MB of operands,
// profiles/r02_pmc_traffic.json).  Window-major, a tile belongs to one or two windows for BOTH shifts (a shifted
// window overlaps four unshifted ones, a tile covers a row or two of one).  Nothing downstream depends on the order:
// attention is permutation equivariant, every loss is a mean over the masked set (the reference's own order is a
// random permutation, ssl.py:287-304).
constexpr int kMaxWinLds = 2048;          // window slots per sample held in LDS (35^2 nuScenes, 40^2 Waymo geometry)
constexpr int kSortCap = 256;             // tokens of one kind in one window that the in-window sort handles

struct MaskWinGeom { int wx, wy, nwx, nwy; };

__global__ __launch_bounds__(kMaskBlk) void random_mask_win_kernel(const int32_t* __restrict__ sample_start, int n_batch,
                                                                   double keep_frac, uint64_t seed,
                                                                   const int4* __restrict__ voxel_coors, MaskWinGeom g,
                                                                   int32_t* __restrict__ ids_keep,
                                                                   int32_t* __restrict__ ids_mask,
                                                                   int32_t* __restrict__ token_row,
                                                                   int32_t* __restrict__ counts) {
    __shared__ int cur[2 * kMaxWinLds];       // radix-select histogram first, then per (kind, window) cursors
    __shared__ int base[2 * kMaxWinLds];      // per (kind, window) first slot
    __shared__ int scratch[kMaskBlk / 64][kSortCap];
    __shared__ int sm[20];
    __shared__ uint32_t s_prefix;
    __shared__ int s_need;
    int* hist = cur;
    const int b = blockIdx.x;
    const int p0 = sample_start[b];
    const int L = sample_start[b + 1] - p0;
    const int K = (int)((double)L * keep_frac);
    int keep_base = 0, mask_base = 0, keep_total = 0, mask_total = 0;
    for (int k = 0; k < n_batch; ++k) {
        const int l = sample_start[k + 1] - sample_start[k];
        const int kk = (int)((double)l * keep_frac);
        if (k < b) { keep_base += kk; mask_base += l - kk; }
        keep_total += kk;
        mask_total += l - kk;
    }
    if (b == 0 && threadIdx.x == 0) { counts[0] = keep_total; counts[1] = mask_total; }
    // ---- radix select of the K-th smallest key: identical to random_mask_kernel
    uint32_t prefix = 0, prefix_mask = 0;
    int need = K;
    const int shifts[3] = {20, 8, 0};
    const int bits[3] = {12, 12, 8};
    if (K > 0) {
        for (int pass = 0; pass < 3; ++pass) {
            const int nb = 1 << bits[pass];
            for (int t = threadIdx.x; t < nb; t += kMaskBlk) hist[t] = 0;
            __syncthreads();
            for (int i = threadIdx.x; i < L; i += kMaskBlk) {
                const uint32_t key = mask_key(seed, b, i);
                if ((key & prefix_mask) == prefix) atomicAdd(&hist[(key >> shifts[pass]) & (nb - 1)], 1);
            }
            __syncthreads();
            {
                int h[4], v = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int bin = threadIdx.x * 4 + k;
                    h[k] = bin < nb ? hist[bin] : 0;
                    v += h[k];
                }
                int tot;
                int acc = block_scan_1024(v, &tot, sm);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (acc < need && acc + h[k] >= need) {
                        s_prefix = prefix | ((uint32_t)(threadIdx.x * 4 + k) << shifts[pass]);
                        s_need = need - acc;
                    }
                    acc += h[k];
                }
            }
            __syncthreads();
            prefix = s_prefix;
            need = s_need;
            prefix_mask |= (uint32_t)(nb - 1) << shifts[pass];
            __syncthreads();
        }
    }
    const uint32_t T = prefix;
    const int nwin = g.nwx * g.nwy;
    // (clamped: bev_shape comes from the caller, the coordinates from the voxelizer's fp32 grid -- a pillar at or past
    //  bev_shape must not index outside the LDS tables)
    auto window_of = [&](int p) {
        const int4 c = voxel_coors[p];
        const int a = c.w / g.wx, b2 = c.z / g.wy;
        return (a < g.nwx ? (a < 0 ? 0 : a) : g.nwx - 1) * g.nwy + (b2 < g.nwy ? (b2 < 0 ? 0 : b2) : g.nwy - 1);
    };
    // ---- pass A: keep flag per pillar (parked in token_row), pillars per (kind, window)
    for (int t = threadIdx.x; t < 2 * nwin; t += kMaskBlk) cur[t] = 0;
    __syncthreads();
    int run_eq = 0;
    for (int i0 = 0; i0 < L; i0 += kMaskBlk) {
        const int i = i0 + threadIdx.x;
        int lt = 0, eq = 0;
        if (i < L && K > 0) {
            const uint32_t key = mask_key(seed, b, i);
            lt = key < T;
            eq = key == T;
        }
        int tot_eq;
        const int eq_rank = run_eq + block_scan_1024(eq, &tot_eq, sm);
        run_eq += tot_eq;
        if (i < L) {
            const int keep = lt | (eq & (eq_rank < need));
            token_row[p0 + i] = keep;
            atomicAdd(&cur[(keep ? 0 : nwin) + window_of(p0 + i)], 1);
        }
    }
    __syncthreads();
    // ---- exclusive scan over (kind, window): 2 * nwin <= 4096 entries, 4 per thread
    {
        int h[4], v = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = threadIdx.x * 4 + k;
            h[k] = e < 2 * nwin ? cur[e] : 0;
            v += h[k];
        }
        int tot;
        int acc = block_scan_1024(v, &tot, sm);
        const int kept = K;                           // entries [0, nwin) sum to K: the masked part restarts at 0
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = threadIdx.x * 4 + k;
            if (e < 2 * nwin) {
                const int v0 = e < nwin ? acc : acc - kept;
                base[e] = v0;
                cur[e] = v0;
            }
            acc += h[k];
        }
    }
    __syncthreads();
    // ---- pass B: place (arrival order inside a (kind, window) run; made deterministic by pass C)
    for (int i = threadIdx.x; i < L; i += kMaskBlk) {
        const int p = p0 + i;
        const int keep = token_row[p];
        const int slot = atomicAdd(&cur[(keep ? 0 : nwin) + window_of(p)], 1);
        if (keep) ids_keep[keep_base + slot] = p;
        else ids_mask[mask_base + slot] = p;
    }
    __syncthreads();
    // ---- pass C: ascending pillar order inside every run (rank sort, one wave per run), token rows
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int* sc = scratch[wave];
    for (int e = wave; e < 2 * nwin; e += kMaskBlk / 64) {
        const int s0 = base[e], n = cur[e] - s0;
        if (n <= 0) continue;
        int32_t* ids = e < nwin ? ids_keep + keep_base : ids_mask + mask_base;
        const int row0 = e < nwin ? keep_base : keep_total + mask_base;
        if (n <= kSortCap) {
            for (int t = lane; t < n; t += 64) sc[t] = ids[s0 + t];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the wave's LDS writes before its reads
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int t = lane; t < n; t += 64) {
                const int v = sc[t];
                int r = 0;
                for (int u = 0; u < n; ++u) r += sc[u] < v;
                ids[s0 + r] = v;
                token_row[v] = row0 + s0 + r;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else {
            for (int t = lane; t < n; t += 64) token_row[ids[s0 + t]] = row0 + s0 + t;
        }
    }
}

// coordinates of the decoder's token list (kept pillars, then masked pillars) + the kept ids widened to int64 (the
// index dtype torch's gather / index_copy want): one launch instead of two casts, two gathers and a concatenation
__global__ __launch_bounds__(256) void gather_token_coors_kernel(const int32_t* __restrict__ ids_keep, int n_keep,
                                                                 const int32_t* __restrict__ ids_mask, int n_mask,
                                                                 const int4* __restrict__ voxel_coors,
                                                                 int4* __restrict__ coors_out,
                                                                 long long* __restrict__ ids_keep_i64,
                                                                 uint4* __restrict__ zero, long long zero_n16) {
    for (long long e = blockIdx.x * 256 + threadIdx.x; e < zero_n16; e += (long long)gridDim.x * 256)
        zero[e] = make_uint4(0u, 0u, 0u, 0u);          // (the window tables of the build that follows)
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_keep + n_mask; i += gridDim.x * 256) {
        const int id = i < n_keep ? ids_keep[i] : ids_mask[i - n_keep];
        coors_out[i] = voxel_coors[id];
        if (ids_keep_i64 && i < n_keep) ids_keep_i64[i] = id;
    }
}

}  // namespace geomae

using namespace geomae;

extern "C" int geomae_gather_token_coors(const int32_t* ids_keep, int32_t num_keep, const int32_t* ids_mask,
                                         int32_t num_mask, const int32_t* voxel_coors, int32_t* coors_out,
                                         int64_t* ids_keep_i64, hipStream_t stream) {
    return geomae_gather_token_coors_zero(ids_keep, num_keep, ids_mask, num_mask, voxel_coors, coors_out, ids_keep_i64,
                                          nullptr, 0, stream);
}

extern "C" int geomae_gather_token_coors_zero(const int32_t* ids_keep, int32_t num_keep, const int32_t* ids_mask,
                                              int32_t num_mask, const int32_t* voxel_coors, int32_t* coors_out,
                                              int64_t* ids_keep_i64, void* zero, int64_t zero_bytes, hipStream_t stream) {
    GEOMAE_REQUIRE(num_keep >= 0 && num_mask >= 0, "gather_token_coors: bad sizes");
    GEOMAE_REQUIRE(zero_bytes >= 0 && zero_bytes % 16 == 0 && (zero || zero_bytes == 0) && ((uintptr_t)zero % 16) == 0,
                   "gather_token_coors: the zero range must be a 16-byte aligned multiple of 16 bytes");
    if (num_keep + num_mask == 0 && zero_bytes == 0) return GEOMAE_OK;
    GEOMAE_REQUIRE(num_keep + num_mask == 0 || ((ids_keep || num_keep == 0) && (ids_mask || num_mask == 0) && voxel_coors &&
                   coors_out), "gather_token_coors: null argument");
    const int64_t work = num_keep + num_mask > zero_bytes / 16 ? num_keep + num_mask : zero_bytes / 16;
    hipLaunchKernelGGL(gather_token_coors_kernel, dim3(stream_grid(work, 256)), dim3(256), 0, stream, ids_keep,
                       num_keep, ids_mask, num_mask, (const int4*)voxel_coors, (int4*)coors_out, (long long*)ids_keep_i64,
                       (uint4*)zero, (long long)(zero_bytes / 16));
    return check_launch("gather_token_coors_kernel");
}

extern "C" int geomae_random_mask_windowed(const int32_t* sample_start, int32_t batch_size, double keep_fraction,
                                           uint64_t seed, const int32_t* voxel_coors, const GeomaeWindowConfig* window,
                                           int32_t* ids_keep, int32_t* ids_mask, int32_t* token_row, int32_t* counts,
                                           hipStream_t stream) {
    GEOMAE_REQUIRE(sample_start && ids_keep && ids_mask && token_row && counts && voxel_coors && window,
                   "random_mask_windowed: null argument");
    GEOMAE_REQUIRE(batch_size >= 1 && keep_fraction >= 0.0 && keep_fraction <= 1.0, "random_mask_windowed: bad arguments");
    GEOMAE_REQUIRE(window->window_shape[0] >= 1 && window->window_shape[1] >= 1 && window->bev_shape[0] >= 1 &&
                   window->bev_shape[1] >= 1, "random_mask_windowed: bad window configuration");
    MaskWinGeom g;
    g.wx = window->window_shape[0];
    g.wy = window->window_shape[1];
    g.nwx = (window->bev_shape[0] + g.wx - 1) / g.wx;
    g.nwy = (window->bev_shape[1] + g.wy - 1) / g.wy;
    // a window table that does not fit in LDS, or windows larger than the rank sort of pass C handles (its runs would keep
    // their atomic arrival order: not reproducible run to run): pillar order
    if ((int64_t)g.nwx * g.nwy > kMaxWinLds || g.wx * g.wy > kSortCap)
        return geomae_random_mask(sample_start, batch_size, keep_fraction, seed, ids_keep, ids_mask, token_row, counts, stream);
    hipLaunchKernelGGL(random_mask_win_kernel, dim3(batch_size), dim3(kMaskBlk), 0, stream, sample_start, batch_size,
                       keep_fraction, seed, (const int4*)voxel_coors, g, ids_keep, ids_mask, token_row, counts);
    return check_launch("random_mask_win_kernel");
}

extern "C" int geomae_random_mask(const int32_t* sample_start, int32_t batch_size, double keep_fraction,
                                  uint64_t seed, int32_t* ids_keep, int32_t* ids_mask, int32_t* token_row,
                                  int32_t* counts, hipStream_t stream) {
    GEOMAE_REQUIRE(sample_start && ids_keep && ids_mask && token_row && counts, "random_mask: null argument");
    GEOMAE_REQUIRE(batch_size >= 1 && keep_fraction >= 0.0 && keep_fraction <= 1.0, "random_mask: bad arguments");
    hipLaunchKernelGGL(random_mask_kernel, dim3(batch_size), dim3(kMaskBlk), 0, stream, sample_start, batch_size,
                       keep_fraction, seed, ids_keep, ids_mask, token_row, counts);
    return check_launch("random_mask_kernel");
}
