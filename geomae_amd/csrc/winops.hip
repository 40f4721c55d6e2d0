// Operator-level window plumbing of mmdet3d.ops (ops/__init__.py:22-26 exports flat2window, window2flat,
// get_flat2win_inds, get_inner_win_inds, make_continuous_inds; ops/sst/sst_ops.py:57-135, 225-251, 271-319, 371-388).
// The pre-training path never materialises the padded [W, T, C] tensors these functions exchange (it works on the CSR
// layout of window.hip); they exist so that other reference modules that import them keep working on this library:
//   * geomae_window_rank: make_continuous_inds + get_inner_win_inds from the pillar-segment counting sort
//     (segment.hip) over the window ids -- no torch.sort / unique / bincount / cumsum chain;
//   * geomae_rows_scatter / geomae_rows_gather: the index copies of flat2window / window2flat, any element type.
#include "common.h"
#include "../../include/geomae_hip.h"

namespace geomae {
namespace {

// position of every token inside the segment-sorted order -> rank inside its window; inv = continuous window id
__global__ __launch_bounds__(256) void window_rank_kernel(const int32_t* __restrict__ order, const int32_t* __restrict__ inv,
                                                          const int32_t* __restrict__ seg_start, int64_t n,
                                                          long long* __restrict__ conti, long long* __restrict__ inner) {
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (int64_t)gridDim.x * 256) {
        const int32_t t = order[j];
        const int32_t w = inv[t];
        conti[t] = w;
        inner[t] = j - seg_start[w];
    }
}

// one row per group of `lanes` threads; rows are copied in 16-byte (or 4-byte) pieces
template <typename V, bool kScatter>
__global__ __launch_bounds__(256) void rows_copy_kernel(const V* __restrict__ src, const long long* __restrict__ idx,
                                                        int64_t n, int pieces, V* __restrict__ dst) {
    const int64_t total = n * pieces;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int64_t r = q / pieces;
        const int p = (int)(q - r * pieces);
        const int64_t o = (int64_t)idx[r];
        if (kScatter) dst[o * pieces + p] = src[q];
        else dst[q] = src[o * pieces + p];
    }
}

template <bool kScatter>
int rows_copy(const void* src, const int64_t* idx, int64_t n, int32_t row_bytes, void* dst, hipStream_t stream,
              const char* who) {
    GEOMAE_REQUIRE(n >= 0 && row_bytes > 0, "%s: bad sizes", who);
    if (n == 0) return GEOMAE_OK;
    GEOMAE_REQUIRE(src && idx && dst, "%s: null argument", who);
    const bool wide = row_bytes % 16 == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0);
    if (wide) {
        const int pieces = row_bytes / 16;
        hipLaunchKernelGGL((rows_copy_kernel<uint4, kScatter>), dim3(stream_grid(n * pieces, 256)), dim3(256), 0, stream,
                           (const uint4*)src, (const long long*)idx, n, pieces, (uint4*)dst);
    } else if (row_bytes % 4 == 0) {
        const int pieces = row_bytes / 4;
        hipLaunchKernelGGL((rows_copy_kernel<uint32_t, kScatter>), dim3(stream_grid(n * pieces, 256)), dim3(256), 0, stream,
                           (const uint32_t*)src, (const long long*)idx, n, pieces, (uint32_t*)dst);
    } else {
        hipLaunchKernelGGL((rows_copy_kernel<uint8_t, kScatter>), dim3(stream_grid(n * row_bytes, 256)), dim3(256), 0, stream,
                           (const uint8_t*)src, (const long long*)idx, n, row_bytes, (uint8_t*)dst);
    }
    return check_launch(who);
}

}  // namespace
}  // namespace geomae

using namespace geomae;

extern "C" int geomae_window_rank(const int32_t* order, const int32_t* inv, const int32_t* seg_start, int64_t num_tokens,
                                  int64_t* continuous_inds, int64_t* inner_inds, hipStream_t stream) {
    GEOMAE_REQUIRE(num_tokens >= 0, "window_rank: bad size");
    if (num_tokens == 0) return GEOMAE_OK;
    GEOMAE_REQUIRE(order && inv && seg_start && continuous_inds && inner_inds, "window_rank: null argument");
    hipLaunchKernelGGL(window_rank_kernel, dim3(stream_grid(num_tokens, 256)), dim3(256), 0, stream, order, inv, seg_start,
                       num_tokens, (long long*)continuous_inds, (long long*)inner_inds);
    return check_launch("window_rank_kernel");
}

extern "C" int geomae_rows_scatter(const void* src, const int64_t* row_index, int64_t num_rows, int32_t row_bytes, void* dst,
                                   hipStream_t stream) {
    return rows_copy<true>(src, row_index, num_rows, row_bytes, dst, stream, "rows_scatter");
}

extern "C" int geomae_rows_gather(const void* src, const int64_t* row_index, int64_t num_rows, int32_t row_bytes, void* dst,
                                  hipStream_t stream) {
    return rows_copy<false>(src, row_index, num_rows, row_bytes, dst, stream, "rows_gather");
}
