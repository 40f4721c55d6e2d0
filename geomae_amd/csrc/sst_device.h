// Device helpers shared by the fused SST layer kernels and the fused heads+loss kernel (gfx950).
// See sst_layer.hip for the design notes (T-layout, K-permuted packed weights, LDS-shared weight tiles).
#pragma once
#include "common.h"

namespace geomae {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ f32x4 mfma32(uint4 a, uint4 b, f32x4 c) {
    union { uint4 u; bf16x8_t v; } fa, fb;
    fa.u = a;
    fb.u = b;
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.v, fb.v, c, 0, 0, 0);
}

__device__ __forceinline__ unsigned int f2bf_bits(float f) {
    unsigned int u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
// v_cvt_pk_bf16_f32 (gfx950): round-to-nearest-even like f2bf_bits, one instruction per pair
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ unsigned int pack2(float a, float b) {
    const f32x2_t v = {a, b};
    union { bf16x2_t h; unsigned int u; } c;
    c.h = __builtin_convertvector(v, bf16x2_t);
    return c.u;
}
__device__ __forceinline__ uint2 pack4(const f32x4 v) { return make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3])); }
__device__ __forceinline__ float bf_lo(unsigned int w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned int w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ f32x4 unpack4(const uint2 p) {
    f32x4 v = {bf_lo(p.x), bf_hi(p.x), bf_lo(p.y), bf_hi(p.y)};
    return v;
}

// packed position p (inside a row of K) <-> original contraction index k
__host__ __device__ __forceinline__ int kperm(int p) { return (p & ~31) + 16 * ((p >> 2) & 1) + 4 * ((p >> 3) & 3) + (p & 3); }

// ------------------------------------------------------------------------------------------------
// Y^T[N x 16 tokens] += Wp[N x K] * X^T : acc[ot][r] = Y[t][16*ot + 4*g + r]  (T-layout in, T-layout out)
// ------------------------------------------------------------------------------------------------
constexpr int kLayerBlk = 256;   // 4 waves = 4 token tiles (64 tokens) per workgroup
constexpr int kPad = 8;          // bf16 elements (16 B) of row padding in LDS: b128 fragment reads conflict-free
constexpr int kWeightLds = 256 * (128 + kPad);   // largest staged matrix: [256][128] (also covers [128][256])

// The workgroup's 4 waves share every weight matrix through LDS: one cooperative copy (L2 -> LDS, 16 B per
// lane) per matrix per 64 tokens, then each wave reads its A fragments with ds_read_b128.  Without this
// every wave pulled the whole matrix through its own vector-memory pipe with 1-2 loads in flight (the
// kernels ran at ~260 cycles per MFMA, profiles/r01b).  All waves of the block must call this together.
// phase timing instrumentation (tools/phase_timing.py builds a second library with -DGEOMAE_PHASE_TIMING;
// a no-op in the product build)
#ifdef GEOMAE_PHASE_TIMING
#define GEOMAE_STAMP_SLOTS 32
#ifndef GEOMAE_STAMP_BLOCKS
#define GEOMAE_STAMP_BLOCKS 512
#endif
#ifndef GEOMAE_STAMP_MAX_GRID
#define GEOMAE_STAMP_MAX_GRID 1000000     // -DGEOMAE_STAMP_MAX_GRID=300: only encoder-size launches leave stamps
#endif
#ifndef GEOMAE_STAMP_MIN_GRID
#define GEOMAE_STAMP_MIN_GRID 0           // -DGEOMAE_STAMP_MIN_GRID=300: only decoder-size launches leave stamps
#endif
static __device__ unsigned long long geomae_stamps[GEOMAE_STAMP_BLOCKS * GEOMAE_STAMP_SLOTS];
#define GEOMAE_STAMP(i)                                                                                   \
    do {                                                                                                  \
        if (threadIdx.x == 0 && blockIdx.x + GEOMAE_STAMP_BLOCKS >= gridDim.x && (i) >= 0 &&              \
            gridDim.x <= GEOMAE_STAMP_MAX_GRID && gridDim.x >= GEOMAE_STAMP_MIN_GRID)                              \
            geomae_stamps[(blockIdx.x % GEOMAE_STAMP_BLOCKS) * GEOMAE_STAMP_SLOTS + (i)] = clock64();     \
    } while (0)
#else
#define GEOMAE_STAMP(i) do {} while (0)
#endif
// wall-clock stamps (s_memrealtime, 100 MHz, one clock for the whole device): when each workgroup of a launch started and
// ended -- clock64 differs between XCDs, so only this shows the shape of a launch (tools/launch_shape.py)
#ifdef GEOMAE_PHASE_TIMING
#define GEOMAE_WSTAMP(slot, kind)                                                                          \
    do {                                                                                                  \
        if (threadIdx.x == 0 && blockIdx.x < GEOMAE_STAMP_BLOCKS && gridDim.x <= GEOMAE_STAMP_MAX_GRID &&  \
            gridDim.x >= GEOMAE_STAMP_MIN_GRID) {                                                          \
            geomae_stamps[blockIdx.x * GEOMAE_STAMP_SLOTS + (slot)] = wall_clock64();                      \
            geomae_stamps[blockIdx.x * GEOMAE_STAMP_SLOTS + 30] = (kind);                                  \
        }                                                                                                 \
    } while (0)
#else
#define GEOMAE_WSTAMP(slot, kind) do {} while (0)
#endif
// -DGEOMAE_STAMP_FWD: the stamps go to the FFN-forward kernels instead (first launch after a clear keeps its stamps)
#if defined(GEOMAE_PHASE_TIMING) && defined(GEOMAE_STAMP_FWD)
#undef GEOMAE_STAMP
#define GEOMAE_STAMP(i) do {} while (0)
#define GEOMAE_FSTAMP(i)                                                                                  \
    do {                                                                                                  \
        if (threadIdx.x == 0 && blockIdx.x < GEOMAE_STAMP_BLOCKS && gridDim.x <= GEOMAE_STAMP_MAX_GRID && \
            gridDim.x >= GEOMAE_STAMP_MIN_GRID && geomae_stamps[blockIdx.x * GEOMAE_STAMP_SLOTS + (i)] == 0) \
            geomae_stamps[blockIdx.x * GEOMAE_STAMP_SLOTS + (i)] = clock64();                             \
    } while (0)
#else
#define GEOMAE_FSTAMP(i) do {} while (0)
#endif

// The copy is split in two so that a kernel can issue the global loads of the NEXT matrix (stage_issue) before
// the elementwise phase that precedes its GEMM: phase timing (tools/phase_timing.py) showed each GEMM waiting
// ~2-3 k cycles for its weights and each elementwise phase waiting as long for its activation rows, with one
// wave per SIMD and nothing else to run.
template <int K, int N>
struct WStage { u32x4 r[N * (K / 8) / kLayerBlk]; };

template <int K, int N>
__device__ __forceinline__ void stage_issue(const bf16_t* __restrict__ Wp, WStage<K, N>& st) {
    constexpr int CH = K / 8;                  // 16-byte chunks per row
    constexpr int PASSES = N * CH / kLayerBlk;
    static_assert(N * CH % kLayerBlk == 0, "matrix must tile over the block");
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int c = p * kLayerBlk + threadIdx.x;
        st.r[p] = *reinterpret_cast<const u32x4*>(Wp + (size_t)(c / CH) * K + 8 * (c % CH));
    }
}

template <int N>
__device__ __forceinline__ void load_bias(const float* __restrict__ b, f32x4 (&acc)[N / 16], int lane);

// `bias`: optional [N] floats the accumulators start from, read AFTER the barriers -- the way to hand over a vector
// that sits in LDS and was written by other threads since the last barrier (stage_params below)
// gemm_staged_then: `after_commit()` runs between the LDS commit of the matrix and its MFMA loop -- the place to request
// the NEXT matrix: the staging registers of this one are free from there on, and the request gets the whole GEMM as extra
// lead (requested behind the GEMM, the W2 stage of sst_ffn_fwd_kernel arrived ~8 k cycles late at decoder size:
// tools/ffn_fwd_time.py).
struct NoHook { __device__ __forceinline__ void operator()() const {} };
// the two halves of a staged GEMM: the matrix into LDS (both barriers), and the MFMA loop over NOT output tiles from OT0
template <int K, int N>
__device__ __forceinline__ void gemm_commit(const WStage<K, N>& st, bf16_t* __restrict__ smem, int sb = -100) {
    constexpr int LD = K + kPad;
    constexpr int CH = K / 8;
    constexpr int PASSES = N * CH / kLayerBlk;
    static_assert(N * LD <= kWeightLds, "LDS weight buffer too small");
    GEOMAE_STAMP(sb);
    __syncthreads();                           // previous matrix fully consumed by every wave
    GEOMAE_STAMP(sb + 1);
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int c = p * kLayerBlk + threadIdx.x;
        *reinterpret_cast<u32x4*>(smem + (c / CH) * LD + 8 * (c % CH)) = st.r[p];
    }
    __syncthreads();
    GEOMAE_STAMP(sb + 2);
}
template <int K, int OT0, int NOT>
__device__ __forceinline__ void gemm_run(const bf16_t* __restrict__ smem, const uint2 (&xb)[K / 16], f32x4 (&acc)[NOT], int lane) {
    constexpr int LD = K + kPad;
    const int o = lane & 15, g = lane >> 4;
    // groups of 4 output tiles advance together over K: consecutive MFMAs hit independent accumulators, so the
    // dependent-accumulator latency of a chain (kk inner loop: 38 % issue stalls in profiles/r01 PMC) is hidden
    constexpr int GRP = NOT < 4 ? NOT : 4;
#pragma unroll
    for (int ot0 = 0; ot0 < NOT; ot0 += GRP) {
#pragma unroll
        for (int kk = 0; kk < K / 32; ++kk) {
            const uint4 b = make_uint4(xb[2 * kk].x, xb[2 * kk].y, xb[2 * kk + 1].x, xb[2 * kk + 1].y);
#pragma unroll
            for (int u = 0; u < GRP; ++u) {
                const uint4 a = *reinterpret_cast<const uint4*>(smem + (16 * (OT0 + ot0 + u) + o) * LD + 8 * g + 32 * kk);
                acc[ot0 + u] = mfma32(a, b, acc[ot0 + u]);
            }
        }
    }
}
template <int K, int N, typename F>
__device__ __forceinline__ void gemm_staged_then(const WStage<K, N>& st, bf16_t* __restrict__ smem,
                                                 const uint2 (&xb)[K / 16], f32x4 (&acc)[N / 16], int lane, F&& after_commit,
                                                 int sb = -100, const float* bias = nullptr) {
    gemm_commit<K, N>(st, smem, sb);
    after_commit();
    if (bias) load_bias<N>(bias, acc, lane);
    gemm_run<K, 0, N / 16>(smem, xb, acc, lane);
}

template <int K, int N>
__device__ __forceinline__ void gemm_staged(const WStage<K, N>& st, bf16_t* __restrict__ smem,
                                            const uint2 (&xb)[K / 16], f32x4 (&acc)[N / 16], int lane, int sb = -100,
                                            const float* bias = nullptr) {
    gemm_staged_then<K, N>(st, smem, xb, acc, lane, NoHook{}, sb, bias);
}

// Column-split form (the "pair" kernels of sst_layer.hip): two waves share one 16-token tile, wave half h computes the
// output tiles [h * N/32, (h + 1) * N/32) from the FULL-K operand -- half the MFMA chain and half the LDS fragment reads
// per wave.  Same staging copy and barriers as gemm_staged; same accumulation order over K, so the results are
// bit-identical to gemm_staged's.
template <int K, int N>
__device__ __forceinline__ void gemm_staged_half(const WStage<K, N>& st, bf16_t* __restrict__ smem,
                                                 const uint2 (&xb)[K / 16], f32x4 (&acc)[N / 32], int lane, int h,
                                                 const float* bias = nullptr /* of this half */) {
    constexpr int LD = K + kPad;
    constexpr int CH = K / 8;
    constexpr int PASSES = N * CH / kLayerBlk;
    static_assert(N * LD <= kWeightLds, "LDS weight buffer too small");
    __syncthreads();
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int c = p * kLayerBlk + threadIdx.x;
        *reinterpret_cast<u32x4*>(smem + (c / CH) * LD + 8 * (c % CH)) = st.r[p];
    }
    __syncthreads();
    if (bias) load_bias<N / 2>(bias, acc, lane);
    const int o = lane & 15, g = lane >> 4;
    const bf16_t* base = smem + h * (N / 2) * LD + o * LD + 8 * g;
    constexpr int NH = N / 32;
    constexpr int GRP = NH < 4 ? NH : 4;
#pragma unroll
    for (int ot0 = 0; ot0 < NH; ot0 += GRP) {
#pragma unroll
        for (int kk = 0; kk < K / 32; ++kk) {
            const uint4 b = make_uint4(xb[2 * kk].x, xb[2 * kk].y, xb[2 * kk + 1].x, xb[2 * kk + 1].y);
#pragma unroll
            for (int u = 0; u < GRP; ++u) {
                const uint4 a = *reinterpret_cast<const uint4*>(base + 16 * (ot0 + u) * LD + 32 * kk);
                acc[ot0 + u] = mfma32(a, b, acc[ot0 + u]);
            }
        }
    }
}

// The two waves of a pair (wave w and w ^ 2 of the 4-wave workgroup) swap NV 16-byte values per lane through LDS:
// lane l of one wave receives what lane l of the other wrote.  `xch` holds 4 waves x NV x 64 lanes x 16 B.  The caller
// keeps a workgroup barrier between two exchanges (every gemm_staged* has two).
template <int NV>
__device__ __forceinline__ void pair_exchange(uint4* __restrict__ xch, int wave, int lane, const uint4 (&mine)[NV],
                                              uint4 (&theirs)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) xch[(wave * NV + i) * 64 + lane] = mine[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) theirs[i] = xch[((wave ^ 2) * NV + i) * 64 + lane];
}
// full[NT tiles of half 0 | NT tiles of half 1] from this wave's half (h) and its partner's
template <int NT>
__device__ __forceinline__ void join_halves(const f32x4 (&mine)[NT], const f32x4 (&theirs)[NT], int h, f32x4 (&full)[2 * NT]) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        full[i] = h ? theirs[i] : mine[i];
        full[NT + i] = h ? mine[i] : theirs[i];
    }
}
template <int NT, typename T>
__device__ __forceinline__ void half_of(const T (&full)[2 * NT], int h, T (&mine)[NT]) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        // (both values first: `h ? full[NT + i] : full[i]` became a select of two ADDRESSES in sst_ffn_bwd_pair_kernel and
        //  sent the whole array to scratch, 211 scratch accesses)
        const T a = full[i], b = full[NT + i];
        mine[i] = h ? b : a;
    }
}

template <int K, int N>
__device__ __forceinline__ void gemm_t(const bf16_t* __restrict__ Wp, bf16_t* __restrict__ smem,
                                       const uint2 (&xb)[K / 16], f32x4 (&acc)[N / 16], int lane, int sb = -100) {
    WStage<K, N> st;
    stage_issue<K, N>(Wp, st);
    gemm_staged<K, N>(st, smem, xb, acc, lane, sb);
}

template <int N>
__device__ __forceinline__ void load_bias(const float* __restrict__ b, f32x4 (&acc)[N / 16], int lane) {
    const int g = lane >> 4;
#pragma unroll
    for (int ot = 0; ot < N / 16; ++ot) {
        const float4 v = *reinterpret_cast<const float4*>(b + 16 * ot + 4 * g);
        acc[ot][0] = v.x; acc[ot][1] = v.y; acc[ot][2] = v.z; acc[ot][3] = v.w;
    }
}

// ------------------------------------------------------------------------------------------------
// Token rows through raw buffer descriptors.  A [n, C] activation is addressed as a buffer of n * row_bytes bytes:
// rows at or past n (the tail of the last 16-token tile, up to 63 rows) read as ZERO and their stores are dropped
// by the hardware's range check.  That removes the per-access `if (valid)` (an s_and_saveexec / branch / s_or
// triple around every load and store, ~270 of them in sst_ffn_bwd_kernel, which also fenced the scheduler) and
// the 64-bit address arithmetic: one 32-bit byte offset per tensor, the tile / column steps ride in the 12-bit
// immediate.  Host side guarantees n * 768 < 2^31 (geomae_sst_* entry points).
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
// Every descriptor in these kernels is wave-uniform (kernel arguments, or a task picked by blockIdx), but the compiler
// loses that whenever a component passes through a select / phi it keeps in VGPRs (the `blk ? .. : ..` row counts, a
// task struct indexed by blockIdx.y): it then wraps EVERY buffer access in a waterfall loop (4 v_readfirstlane +
// 2 v_cmp + s_and_saveexec + branch; 193 of them in sst_ffn_bwd_dw_kernel).  Pinning base and size to SGPRs here
// removes the loops.
__device__ __forceinline__ void* uniform_ptr(const void* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc(const void* base, int n, int row_bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(base), 0, __builtin_amdgcn_readfirstlane(n * row_bytes), 0x00020000);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t table_rsrc(const void* base) {   // no bound known / needed
    return __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(base), 0, -1, 0x00020000);
}
__device__ __forceinline__ uint2 buf_load_b64(__amdgpu_buffer_rsrc_t r, int off) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
    return make_uint2(v[0], v[1]);
}
__device__ __forceinline__ void buf_store_b64(__amdgpu_buffer_rsrc_t r, int off, uint2 v) {
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{v.x, v.y}, r, off, 0, 0);
}
__device__ __forceinline__ f32x4 buf_load_f32x4(__amdgpu_buffer_rsrc_t r, int off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

// Row layouts.  Row-major [n, ld] is what the C ABI exchanges.  The T-layout access of a wave -- lane (t = l & 15,
// g = l >> 4) touches 8 or 16 bytes of token t -- hits 16 different rows per instruction there: 64 separate
// requests in the memory pipeline; tools/microbench_row_access.hip measures 64 such store instructions per wave at
// 8.5 us against 3.0 us for row-contiguous ones (105 workgroups), and a layer kernel issues ~120 of them per wave.
// The tensors that only these kernels exchange (inside geomae_sst_stack_*) therefore use the TILE-BLOCKED layout
//     [n/16][ld/16][16 tokens][16 channels],
// in which the 64 lanes of one access instruction cover one contiguous 512-byte (bf16) or 1-KB (fp32) block.  Both
// layouts have the same tile stride (16 * ld elements); `blk` selects the in-tile strides at run time (wave-uniform:
// scalar selects, and the per-tile step rides in the instruction's scalar offset).  Row-major relies on the buffer
// range check for the tail rows of the last tile; blocked buffers are allocated for ceil16(n) rows, whose pad rows
// hold finite-or-garbage values that never leave their own token (every per-token op is column-independent in the
// T-layout; the two reductions over tokens mask them: ln_param_grads_t here, the token bound in dw_body).
struct RowAddr {
    __amdgpu_buffer_rsrc_t r;
    int voff;        // byte offset of this lane's piece in tile ct = 0
    int ct_stride;   // bytes between consecutive 16-channel tiles
};
template <int ELEM>
__device__ __forceinline__ RowAddr row_addr(const void* base, int n, int tok, int ld, int col0, int lane, bool blk) {
    RowAddr a;
    const int rowb = ld * ELEM, g = lane >> 4;
    a.r = rows_rsrc(base, blk ? ((n + 15) & ~15) : n, rowb);
    a.voff = blk ? (tok >> 4) * (16 * rowb) + (col0 >> 4) * (256 * ELEM) + (tok & 15) * (16 * ELEM) + 4 * ELEM * g
                 : tok * rowb + col0 * ELEM + 4 * ELEM * g;
    a.ct_stride = blk ? 256 * ELEM : 16 * ELEM;
    return a;
}

// lane (t = l & 15, g = l >> 4) holds channels 16 ct + 4 g + {0..3} of token `tok` (T-layout accumulator order)
template <int C>
__device__ __forceinline__ void load_rows_f32(const float* __restrict__ src, int n, int tok, f32x4 (&v)[C / 16], int lane,
                                              bool blk = false) {
    const RowAddr a = row_addr<4>(src, n, tok, C, 0, lane, blk);
#pragma unroll
    for (int ct = 0; ct < C / 16; ++ct)
        v[ct] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a.r, a.voff, ct * a.ct_stride, 0));
}

// C channels starting at column col0 of a [n, ld] bf16 matrix
template <int C>
__device__ __forceinline__ void load_rows_bf16(const bf16_t* __restrict__ src, int n, int tok, int ld, int col0,
                                               uint2 (&v)[C / 16], int lane, bool blk = false) {
    const RowAddr a = row_addr<2>(src, n, tok, ld, col0, lane, blk);
#pragma unroll
    for (int ct = 0; ct < C / 16; ++ct) {
        const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(a.r, a.voff, ct * a.ct_stride, 0);
        v[ct] = make_uint2(t[0], t[1]);
    }
}

template <int C>
__device__ __forceinline__ void store_rows_f32(float* __restrict__ dst, int n, int tok, const f32x4 (&v)[C / 16], int lane,
                                               bool blk = false) {
    const RowAddr a = row_addr<4>(dst, n, tok, C, 0, lane, blk);
#pragma unroll
    for (int ct = 0; ct < C / 16; ++ct)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[ct]), a.r, a.voff, ct * a.ct_stride, 0);
}

// C channels starting at column col0 of a [n, ld] fp32 matrix
template <int C>
__device__ __forceinline__ void load_rows_f32_cols(const float* __restrict__ src, int n, int tok, int ld, int col0,
                                                   f32x4 (&v)[C / 16], int lane, bool blk = false) {
    const RowAddr a = row_addr<4>(src, n, tok, ld, col0, lane, blk);
#pragma unroll
    for (int ct = 0; ct < C / 16; ++ct)
        v[ct] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a.r, a.voff, ct * a.ct_stride, 0));
}
template <int C>
__device__ __forceinline__ void store_rows_f32_cols(float* __restrict__ dst, int n, int tok, int ld, int col0,
                                                    const f32x4 (&v)[C / 16], int lane, bool blk = false) {
    const RowAddr a = row_addr<4>(dst, n, tok, ld, col0, lane, blk);
#pragma unroll
    for (int ct = 0; ct < C / 16; ++ct)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[ct]), a.r, a.voff, ct * a.ct_stride, 0);
}

template <int C>
__device__ __forceinline__ void store_rows_packed(bf16_t* __restrict__ dst, int n, int tok, int ld, int col0,
                                                  const uint2 (&v)[C / 16], int lane, bool blk = false) {
    const RowAddr a = row_addr<2>(dst, n, tok, ld, col0, lane, blk);
#pragma unroll
    for (int ct = 0; ct < C / 16; ++ct)
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{v[ct].x, v[ct].y}, a.r, a.voff, ct * a.ct_stride, 0);
}

template <int C>
__device__ __forceinline__ void store_rows_bf16(bf16_t* __restrict__ dst, int n, int tok, int ld, int col0,
                                                const f32x4 (&v)[C / 16], int lane, bool blk = false) {
    const RowAddr a = row_addr<2>(dst, n, tok, ld, col0, lane, blk);
#pragma unroll
    for (int ct = 0; ct < C / 16; ++ct) {
        const uint2 p = pack4(v[ct]);
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{p.x, p.y}, a.r, a.voff, ct * a.ct_stride, 0);
    }
}

// sum over the 128 channels of a token (spread over 8 tiles x 4 regs in-lane and the 4 lanes of group g)
__device__ __forceinline__ float row_sum(float v) { return rows4_sum(v); }

// LayerNorm over 128 channels in T-layout; returns xhat in place, rstd out
__device__ __forceinline__ void layer_norm_t(f32x4 (&u)[8], float eps, float* rstd_out) {
    float s = 0.f;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) s += (u[ct][0] + u[ct][1]) + (u[ct][2] + u[ct][3]);
    const float mean = row_sum(s) * (1.0f / 128.0f);
    float q = 0.f;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = u[ct][r] - mean;
            u[ct][r] = d;
            q += d * d;
        }
    const float var = row_sum(q) * (1.0f / 128.0f);
    const float rstd = rsqrtf(var + eps);
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) u[ct][r] *= rstd;
    *rstd_out = rstd;
}

__device__ __forceinline__ void affine_t(const f32x4 (&xhat)[8], const float* __restrict__ w,
                                         const float* __restrict__ b, f32x4 (&y)[8], int lane) {
    const int g = lane >> 4;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
        const float4 wv = *reinterpret_cast<const float4*>(w + 16 * ct + 4 * g);
        const float4 bv = *reinterpret_cast<const float4*>(b + 16 * ct + 4 * g);
        y[ct][0] = xhat[ct][0] * wv.x + bv.x;
        y[ct][1] = xhat[ct][1] * wv.y + bv.y;
        y[ct][2] = xhat[ct][2] * wv.z + bv.z;
        y[ct][3] = xhat[ct][3] * wv.w + bv.w;
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma     (in place on dy)
__device__ __forceinline__ void layer_norm_bwd_t(f32x4 (&dy)[8], const f32x4 (&xhat)[8], const float* __restrict__ w,
                                                 float rstd, int lane) {
    const int g = lane >> 4;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
        const float4 wv = *reinterpret_cast<const float4*>(w + 16 * ct + 4 * g);
        dy[ct][0] *= wv.x; dy[ct][1] *= wv.y; dy[ct][2] *= wv.z; dy[ct][3] *= wv.w;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s1 += dy[ct][r];
            s2 += dy[ct][r] * xhat[ct][r];
        }
    }
    const float m1 = row_sum(s1) * (1.0f / 128.0f), m2 = row_sum(s2) * (1.0f / 128.0f);
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) dy[ct][r] = rstd * (dy[ct][r] - m1 - xhat[ct][r] * m2);
}

// GELU (erf form, as F.gelu) and its derivative.  erf by Abramowitz-Stegun 7.1.26 (|err| < 1.5e-7): one
// exp + one rcp instead of libm erff (~60 instructions); exp(-x^2/2) is shared with the pdf term.
__device__ __forceinline__ void gelu_parts(float x, float* cdf, float* pdf) {
    const float z = fabsf(x) * 0.70710678118654752f;
    // v_rcp_f32 (1 ulp).  __frcp_rn is the correctly rounded reciprocal: under -fno-fast-math it expands to the
    // div_scale / div_fmas / div_fixup sequence (~12 instructions per element, 768 per kernel), far more precision
    // than the 1.5e-7 polynomial around it needs.
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float e = __expf(-z * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.0f - poly * e;
    *cdf = 0.5f * (1.0f + copysignf(erf_abs, x));
    *pdf = 0.3989422804014327f * e;
}
__device__ __forceinline__ float gelu_f(float x) {
    float c, p;
    gelu_parts(x, &c, &p);
    return x * c;
}
// The same on the four values of an accumulator register group, two at a time on the packed fp32 pipe (v_pk_mul_f32 /
// v_pk_fma_f32 process two floats per lane and issue slot): 17 full-rate + 4 quarter-rate instructions per PAIR instead
// of 14 + 2 per value.  The GELU arithmetic is VALU-bound: 64 evaluations per lane and kernel were 7.2 k of the 46 k
// cycles of a workgroup in sst_ffn_bwd_kernel (tools/archive/phase_timing.py).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_parts2(f32x2 x, f32x2* cdf, f32x2* pdf) {
    const f32x2 ax = {__builtin_fabsf(x.x), __builtin_fabsf(x.y)};
    const f32x2 z = ax * 0.70710678118654752f;
    const f32x2 den = z * 0.3275911f + 1.0f;
    const f32x2 t = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    const f32x2 a2 = z * z * -1.4426950408889634f;                    // exp(-z^2) = exp2(-z^2 log2 e)
    const f32x2 e = {__builtin_amdgcn_exp2f(a2.x), __builtin_amdgcn_exp2f(a2.y)};
    const f32x2 poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const f32x2 erf_abs = 1.0f - poly * e;
    const f32x2 sg = {__builtin_copysignf(erf_abs.x, x.x), __builtin_copysignf(erf_abs.y, x.y)};
    *cdf = sg * 0.5f + 0.5f;
    *pdf = e * 0.3989422804014327f;
}
__device__ __forceinline__ f32x4 gelu4(const f32x4 x) {
    f32x2 c0, c1, p0, p1;
    gelu_parts2(f32x2{x[0], x[1]}, &c0, &p0);
    gelu_parts2(f32x2{x[2], x[3]}, &c1, &p1);
    return f32x4{x[0] * c0.x, x[1] * c0.y, x[2] * c1.x, x[3] * c1.y};
}
// h = x * cdf(x) and d(x gelu)/dx = cdf + x * pdf
__device__ __forceinline__ void gelu_fwd_bwd4(const f32x4 x, f32x4* h, f32x4* grad) {
    f32x2 c0, c1, p0, p1;
    const f32x2 x0 = {x[0], x[1]}, x1 = {x[2], x[3]};
    gelu_parts2(x0, &c0, &p0);
    gelu_parts2(x1, &c1, &p1);
    const f32x2 h0 = x0 * c0, h1 = x1 * c1, g0 = c0 + x0 * p0, g1 = c1 + x1 * p1;
    *h = f32x4{h0.x, h0.y, h1.x, h1.y};
    *grad = f32x4{g0.x, g0.y, g1.x, g1.y};
}
__device__ __forceinline__ float gelu_grad(float x) {
    float c, p;
    gelu_parts(x, &c, &p);
    return c + x * p;
}

// sum over the 16 tokens of the wave (lanes with equal g); result valid in every lane
__device__ __forceinline__ float tok_sum(float v) { return row16_sum(v); }

// LayerNorm parameter gradients of one wave's token tile: red[k0][c] = sum_t dy[c][t] * xhat[c][t],
// red[k0 + 1][c] = sum_t dy[c][t].  In T-layout the token index is the lane (l & 15), so a register-level
// reduction costs 4 DPP steps for each of the 2 x 32 values of a lane plus a leader-lane branch per value
// (~700 instructions per LayerNorm, a fifth of sst_ffn_bwd_kernel).  Instead the wave writes both tensors
// token-major into a private LDS scratch (16 x ds_write_b128), and every lane sums a 4-channel column
// of one tensor over the 16 token rows (16 x ds_read_b128 + 15 float4 adds): ~90 instructions, no branches.
// Row stride 132 floats: the 16 lanes of a token-group start 4 banks apart (conflict-free b128 writes), the
// reads of a row are contiguous.  `scratch` is the (idle) weight buffer: the caller has a workgroup barrier
// between the last GEMM that read it and this call; the next gemm_staged barriers before overwriting it.
constexpr int kRedLd = 132;
constexpr int kRedWaveFloats = 2 * 16 * kRedLd;              // per wave: two tensors x 16 tokens
static_assert(4 * kRedWaveFloats * 4 <= kWeightLds * 2, "token-reduction scratch must fit the weight buffer");
__device__ __forceinline__ void ln_param_grads_t(const f32x4 (&dy)[8], const f32x4 (&xhat)[8], float* __restrict__ scratch,
                                                 float (*red_wave)[128] /* [tensor][channel] of this wave */, int k0,
                                                 int lane, bool valid) {
    const int t = lane & 15, g = lane >> 4;
    float* row = scratch + t * kRedLd + 4 * g;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {                 // rows past the token count contribute nothing (pad rows of a
        *reinterpret_cast<f32x4*>(row + 16 * ct) = valid ? dy[ct] * xhat[ct] : zero;       // blocked buffer are not zero)
        *reinterpret_cast<f32x4*>(row + 16 * kRedLd + 16 * ct) = valid ? dy[ct] : zero;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int k = lane >> 5, c4 = lane & 31;
    const float* col = scratch + k * 16 * kRedLd + 4 * c4;
    f32x4 s0 = *reinterpret_cast<const f32x4*>(col), s1 = *reinterpret_cast<const f32x4*>(col + kRedLd);
#pragma unroll
    for (int tt = 2; tt < 16; tt += 2) {
        s0 += *reinterpret_cast<const f32x4*>(col + tt * kRedLd);
        s1 += *reinterpret_cast<const f32x4*>(col + (tt + 1) * kRedLd);
    }
    *reinterpret_cast<f32x4*>(&red_wave[k0 + k][4 * c4]) = s0 + s1;
}



typedef __attribute__((ext_vector_type(4))) short s16x4;
// ds_read_b64_tr_b16: within a 16-lane group, lane 4a+b receives element b of lanes a, 4+a, 8+a, 12+a.
// With lane m pointing at row (m>>2), 4-element column chunk (m&3) of a row-major [4 x 16] block, lane c gets
// column c of the block: 4 consecutive TOKENS of one channel -- the MFMA fragment of a token contraction,
// read straight from the token-major slab (measured on gfx950, round 1).
__device__ __forceinline__ uint2 tr_read(const bf16_t* p) {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    union { s16x4 s; uint2 u; } c;
    c.s = v;
    return c.u;
}

// weight-gradient (token contraction) tasks, see dw_kernel in sst_layer.hip
struct DwTask {
    const bf16_t* A; int lda, a_col0;
    const bf16_t* B; int ldb, b_col0;
    float* C; int ldc, c_row0, c_col0;
    float* dbias;
    int rows_valid;          // only rows i < rows_valid of the 128-row block exist
    // B formed on load: B[t][c] = b_scale[c] * B[t][c] + b_shift[c] (or null).  How dW1 = dhp^T y reads the SAVED xhat1 of the
    // forward instead of a y = affine(xhat1) copy that the ffn backward used to store for it (256 of the 2.5 KB it wrote per
    // token: the decoder-size launches run at the HBM roof)
    const float* b_scale = nullptr;
    const float* b_shift = nullptr;
};
constexpr int kMaxDwTasks = 12;
struct DwTasks {
    DwTask t[kMaxDwTasks];
    int blocked = 0;         // operands A, B in the tile-blocked layout (the layer stacks' slabs) instead of row-major
    float* partial = nullptr;    // split-K through memory: [task][chunk][128 x 128] fp32 partial sums (dw_body), or atomics
};
// The sum of a previous launch's partials, carried by a later launch (dw_reduce_body)
struct DwReduceTask { float* C; int ldc, c_row0, c_col0, rows_valid; };
struct DwReduce {
    const float* partial = nullptr;          // nullptr: nothing to reduce
    int gx = 0, num_tasks = 0;
    DwReduceTask t[8];
};
constexpr int kDwReduceBlocks = 64;          // workgroups a carrying launch adds for the reduction
int launch_dw(const DwTasks& tasks, int num_tasks, int num_tokens, hipStream_t stream);
// C[128,128] += A^T B over n tokens, A / B tile-blocked [n,128] bf16, through the layer-form kernel (dw_device.h SPLIT job);
// `partial`: the caller's split-K workspace (2 * kDwPartialBytes)
int launch_dw_split(const bf16_t* A, const bf16_t* B, int n, float* C, float* partial, hipStream_t stream);
// the next geomae_sst_weight_grad call of this host thread only records its tasks; the following
// geomae_sst_ffn_backward launches them inside its own kernel (sst_ffn_bwd_dw_kernel)
void defer_next_weight_grad();
int flush_pending_weight_grad(hipStream_t stream);      // launches a recorded-but-unlaunched contraction, if any

// the packed weights / fp32 parameter vectors of one layer as the kernels take them (GeomaeSstLayerWeights, host side)
struct LayerW {
    const bf16_t *wqkv, *wqkT, *wvT, *wo, *woT, *w1, *w1T, *w2, *w2T;
    const float *bqkv, *bo, *b1, *b2, *g1, *be1, *g2, *be2;
    const bf16_t* frag;      // the nine matrices again, fragment-major, at the element offsets below (or null)
};
// element offsets of the matrices inside a layer's packed block (row-major block and fragment-major block alike)
constexpr int kOffWqkv = 0, kOffWqkT = 49152, kOffWvT = 81920, kOffWo = 98304, kOffWoT = 114688, kOffW1 = 131072,
              kOffW1T = 163840, kOffW2 = 196608, kOffW2T = 229376, kPackedPerLayer = 262144;
#ifdef GEOMAE_HIP_H
// sst_fused.hip: the backward of one layer as ONE launch (bundles of at most four tiles); called per layer by sst_stack.hip
int sst_layer_backward_fused(const float* dz, const float* dz_add, bool dz_rowmajor, float* dx, bool dx_rowmajor,
                             const int32_t* out_rows, int n_out, int num_tokens, const GeomaeSstLayerWeights* w,
                             const GeomaeSstLayerGrads* g, const GeomaeSstStackLayout* layout, int bundle_cap, const void* qkv,
                             const void* attn, const float* lse, const void* xh1, const void* xh2, const void* hp, const float* rstd,
                             void* dqkv, void* du, void* dv, void* dhp, void* h, hipStream_t stream);
// sst_ws.hip: the forward of one layer as ONE weight-stationary launch (workgroups loop over bundles; windows of up to 144 positions)
int sst_layer_forward_ws(const float* x, const SstInputMap& M, int num_tokens, const GeomaeSstLayerWeights* w,
                         const GeomaeSstStackLayout* layout, const float* pos_table, float* z, bool z_blocked, void* qkv,
                         void* attn, float* lse, void* xh1, void* xh2, void* hp, float* rstd, void* xb, void* xp,
                         int dead_rows, int max_workgroups, int min_tiles, hipStream_t stream);
inline LayerW to_layer(const GeomaeSstLayerWeights* w) {
    LayerW L;
    L.wqkv = (const bf16_t*)w->wqkv_p; L.wqkT = (const bf16_t*)w->wqkT_p; L.wvT = (const bf16_t*)w->wvT_p;
    L.wo = (const bf16_t*)w->wo_p; L.woT = (const bf16_t*)w->woT_p; L.w1 = (const bf16_t*)w->w1_p;
    L.w1T = (const bf16_t*)w->w1T_p; L.w2 = (const bf16_t*)w->w2_p; L.w2T = (const bf16_t*)w->w2T_p;
    L.bqkv = w->bqkv; L.bo = w->bo; L.b1 = w->b1; L.b2 = w->b2;
    L.g1 = w->ln1_w; L.be1 = w->ln1_b; L.g2 = w->ln2_w; L.be2 = w->ln2_b;
    L.frag = (const bf16_t*)w->frag_p;
    return L;
}
inline int check_weights(const GeomaeSstLayerWeights* w, const char* who) {
    GEOMAE_REQUIRE(w, "%s: null weights", who);
    GEOMAE_REQUIRE(w->wqkv_p && w->wqkT_p && w->wvT_p && w->wo_p && w->woT_p && w->w1_p && w->w1T_p && w->w2_p &&
                   w->w2T_p && w->bqkv && w->bo && w->b1 && w->b2 && w->ln1_w && w->ln1_b && w->ln2_w && w->ln2_b,
                   "%s: null weight pointer", who);
    GEOMAE_REQUIRE(w->d_model == 128 && w->d_ffn == 256, "%s: kernels are built for d_model=128, d_ffn=256", who);
    return GEOMAE_OK;
}
#endif

}  // namespace geomae
