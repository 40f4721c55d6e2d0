// Shared helpers for libgeomae_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define GEOMAE_OK 0
#define GEOMAE_ERR_ARG (-1)
#define GEOMAE_ERR_HIP (-2)
#define GEOMAE_ERR_WORKSPACE (-3)
#define GEOMAE_ERR_CONFIG (-4)

namespace geomae {

void set_error(const char* fmt, ...);
}  // namespace geomae
struct GeomaeTuning;
namespace geomae {
// the process-wide tuning surface (include/geomae_hip.h GeomaeTuning; common.hip)
const GeomaeTuning& tuning();
GeomaeTuning& tuning_mut();          // (the older per-switch setters of the C ABI write through this)

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return GEOMAE_ERR_HIP;
    }
    return GEOMAE_OK;
}

#define GEOMAE_REQUIRE(cond, ...)                 \
    do {                                          \
        if (!(cond)) {                            \
            geomae::set_error(__VA_ARGS__);       \
            return GEOMAE_ERR_ARG;                \
        }                                         \
    } while (0)

// Layout flags (sst_layer.hip kLay*) that the SST layer / attention entry points apply to the tensors they exchange.
// Per host thread, 0 = row-major everywhere (what the C ABI documents); only geomae_sst_stack_forward / _backward set
// it, around their own calls, to keep their internal buffers in the tile-blocked layout (sst_device.h "Row layouts").
void set_layer_layout(int flags);
int layer_layout();
// Row map of the NEXT geomae_sst_qkv_backward of this host thread (set by geomae_sst_stack_backward around its last
// kernel only): token t's gradient row goes to row rows[t] of a [num_rows_out, 128] buffer instead of row t.
void set_output_rows(const int32_t* rows, int num_rows_out);
const int32_t* output_rows(int* num_rows_out);
// Second summand of the NEXT geomae_sst_ffn_backward's dz on this host thread (dz + dz2, row-major like dz): set by
// geomae_sst_stack_backward around its top layer only.
void set_dz_addend(const float* dz2);
void set_y_from_xhat(bool on, const float* gamma1 = nullptr, const float* beta1 = nullptr);   // sst_layer.hip
void set_x_from_xhat(bool on, const float* gamma2 = nullptr, const float* beta2 = nullptr);   // sst_layer.hip: dW_v's operand
void set_skip_x_copy(bool on);      // the next layer forward of this thread stores no bf16 copy of x (only x + pos)
bool skip_x_copy();
const float* dz_addend();
// Column sums of the rows >= from_row of the NEXT geomae_sst_qkv_backward's output on this host thread, ADDED into
// sum[128] (set by geomae_sst_stack_backward around its last kernel only: the decoders' mask-token gradient).
void set_tail_sum(float* sum, int from_row);
float* tail_sum(int* from_row);
// First LIVE token row of the stacks this host thread runs next (geomae_sst_stack_forward / _backward; 0 = all rows):
// the caller promises that it never reads the stack's output rows below it and that the output GRADIENT of those rows is
// zero.  The last layer then skips their out-projection / FFN (forward) and its backward (whose outputs for those rows are
// zeros).  The decoders of the pre-training step: only masked pillars reach the heads (bb.py:300-335), the kept ones --
// 30 % of the tokens -- come first.
void set_first_live_row(int row);
int first_live_row();
// an event that the next multi-kernel entry point of this thread records BETWEEN its launches (geomae_vfe_backward_layer1:
// behind its layer-1 sweep, before the routing sweep), so that a caller can start work on another stream that needs only
// the first kernel's outputs.  One-shot: taken (and cleared) by the callee.
void set_mid_launch_event(hipEvent_t ev);
hipEvent_t take_mid_launch_event();
// ... and a stream that the launches BEHIND that event move to (it waits for the event first): the callee's remaining kernels
// leave the caller's stream, which goes on with what needs only the first kernel's outputs.  One-shot as the event; ignored
// without one.
void set_mid_launch_side(hipStream_t side);
hipStream_t take_mid_launch_side();
// which variant of a layer kernel the last launch on this thread used (0 = plain; 1 = sst_ffn_fwd_pair_kernel /
// sst_ffn_bwd_dw_kernel): the stack's per-kernel timer (bench.py's roofline) keeps a launch's events only if it was the
// kernel asked for, so that its averages are those of ONE kernel of the rocprofv3 table
void set_last_kernel_variant(int v);
int last_kernel_variant();
// The per-kernel launch timer (geomae_profiler_create, sst_stack.hip) of the step this host thread is enqueueing, for
// launches made outside the stack calls that take it as an argument: the deferred weight-gradient contractions, which
// geomae_flush_weight_grad launches on ANOTHER stream -- their event pairs are recorded there, around each dw_kernel.
void set_thread_profiler(void* prof);
void* thread_profiler();
bool profiler_begin(void* prof, int kernel_id, hipStream_t s);      // true: a start event was recorded, call profiler_end
void profiler_end(void* prof, hipStream_t s);
// Split-K workspace of the geomae_sst_weight_grad calls of this host thread (set by geomae_sst_stack_backward for its own
// layers): two buffers of kDwPartialBytes where the contraction's workgroups leave their partial sums instead of
// atomically adding them to the gradients (sst_layer.hip dw_body); nullptr = atomics.
// (round 5: sized for the layer-form contraction, csrc/dw_device.h -- 16 jobs x 8 token chunks x (16 tiles x 512 threads x 16 B
//  + 256 bias floats) per buffer; the two buffers together hold a four-layer launch at 16 chunks per job.  The old form's
//  8 tasks x 24 chunks x [128,128] fp32 = 12.6 MB fits in one.)
constexpr long long kDwPartialBytes = 16ll * 8 * (16 * 512 * 16 + 1024);
void set_dw_partial(float* ws);
float* dw_partial();
// rows [0, r) of the NEXT geomae_sst_weight_grad's token range are DEAD (the top layer of a decoder stack: their dY rows are
// zero and their saved forward rows were never written, set_first_live_row): the layer-form contraction starts behind them
// in the jobs whose dY operand is zero there.  Thread-local, consumed by that call.
// The one-launch layer forward (sst_fused.hip) runs bundles of more than four tiles in a SECOND launch per layer.  A caller that
// knows a layout has no such bundle (the step engine: window.hip window_max_keep, a step ahead) says so before
// geomae_sst_stack_forward: bit s = layout s (unshifted / shifted) may hold one.  Thread-local, consumed by that call;
// default: both may.
void set_fused_big_layouts(int mask);
int take_fused_big_layouts();
// which form the LAST geomae_sst_stack_forward / _backward of this host thread took (include/geomae_hip.h GEOMAE_STACK_FORM_*):
// what geomae_sst_last_stack_forms and the step engine's geomae_pretrain_step_forms report, so that a test can assert that
// the kernel it means to pin actually ran
void set_last_stack_form(bool backward, int form);
int last_stack_form(bool backward);
void set_fused_big_next(bool possible);          // ... handed on to the next geomae_sst_layer_forward of this thread
bool take_fused_big_next();
constexpr int kStackSyncBytes = 2048;      // the persistent stack forward's grid-barrier counters, behind a stack's saved tensors
void set_dw_dead_rows(int rows);
int take_dw_dead_rows();
// Input map of the NEXT geomae_sst_qkv_forward of this host thread (set by geomae_sst_stack_forward around F1 of its first
// layer): instead of reading its tile-blocked x, the kernel gathers token t from row rows[t] (or t) of the row-major
// `src` for t < n_src and takes `fill` for the tokens behind, and WRITES the tile-blocked x (what the stack's other
// kernels read) -- the stack's input conversion without a launch of its own.
struct SstInputMap { const float* src; int n_src; const float* fill; const int32_t* rows; };
void set_input_map(const SstInputMap& m);
SstInputMap input_map();
// geomae_window_build_batch of this host thread finds its window tables (the first table bytes of its workspace) already
// zero: the step engine clears them with the token-coordinate gather that precedes the build (no memset kernel).
void set_window_tables_prezeroed(bool on);
bool window_tables_prezeroed();
// "Defer all" mode of the weight-gradient contractions of geomae_sst_stack_backward on this host thread (sst_layer.hip):
// they are queued and launched by the next geomae_flush_weight_grad on ITS stream.  The stack then needs one set of
// operand slabs per layer (geomae_sst_stack_scratch_bytes_layers).
void set_defer_all_weight_grads(bool on);
// "defer all" with flushes on the way: every `every` layers geomae_sst_stack_backward records `ev` on its stream, makes `side`
// wait for it and launches the contractions queued so far there (geomae_flush_weight_grad(side)) -- the caller flushes the
// rest behind the stack.  How the ENCODER's contractions leave its backward launches (round 5): merged launches of four layers
// on the geometry stream beside the layers still to come, instead of riding in every ffn-backward launch.
struct DwMidFlush { hipStream_t side = nullptr; hipEvent_t ev = nullptr; int every = 0; int budget = 0; };   // budget: workgroups of a flush on the way (0 = default)
// the workgroup budget of the NEXT contraction launch of this thread (one-shot; 0 = by size, sst_layer.hip launch_dw_layers)
void set_dw_budget_hint(int workgroups);
int take_dw_budget_hint();
void set_dw_mid_flush(const DwMidFlush& f);
DwMidFlush dw_mid_flush();
bool defer_all_weight_grads();
// forget every recorded-but-unlaunched contraction of this host thread (error paths; the start of a step)
void drop_pending_weight_grads();
struct LayerLayoutScope {
    explicit LayerLayoutScope(int flags) { set_layer_layout(flags); }
    ~LayerLayoutScope() { set_layer_layout(0); }
};

// Accumulator / atomics-target buffers are zeroed by the entry point that fills them (hipMemsetAsync: a ~5 us fill
// kernel each, a dozen of them between dependent kernels of a training step) unless the calling thread declared, with
// geomae_set_accumulators_prezeroed(1), that it hands in buffers that are already zero (one arena fill per step,
// done off the critical path).  Entry points that honour the mode say so in include/geomae_hip.h.
bool accumulators_prezeroed();
#define GEOMAE_ZERO(ptr, bytes, stream)                                                 \
    do {                                                                                \
        if (!geomae::accumulators_prezeroed()) GEOMAE_HIP(hipMemsetAsync(ptr, 0, bytes, stream)); \
    } while (0)

#define GEOMAE_HIP(call)                                                      \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) {                                               \
            geomae::set_error("%s: %s", #call, hipGetErrorString(e_));        \
            return GEOMAE_ERR_HIP;                                            \
        }                                                                     \
    } while (0)

constexpr int kWave = 64;  // CDNA wavefront

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// memory-bound streaming kernels: cap the grid and grid-stride (guide G11)
inline int stream_grid(int64_t work_items, int block) {
    int64_t g = (work_items + block - 1) / block;
    const int64_t cap = 256 * 8;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// ---- cross-lane reductions without the LDS crossbar.  __shfl_xor compiles to ds_bpermute_b32 (~100 cycles of
// latency each and a dependent chain per reduction: phase timing showed 27 k cycles per LayerNorm backward in
// sst_ffn_bwd_kernel spent in them).  Inside a row of 16 lanes DPP modifiers do the exchange in the VALU; across
// rows gfx950's v_permlane16_swap / v_permlane32_swap do.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;

// sum / max over the 16 lanes of a row (lanes l & ~15 .. | 15); result in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<kDppXor1>(v);
    v += dpp_mov<kDppXor2>(v);
    v += dpp_mov<kDppHalfMirror>(v);
    v += dpp_mov<kDppMirror>(v);
    return v;
}
__device__ __forceinline__ int row16_sum(int v) {
    v += dpp_mov<kDppXor1>(v);
    v += dpp_mov<kDppXor2>(v);
    v += dpp_mov<kDppHalfMirror>(v);
    v += dpp_mov<kDppMirror>(v);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_mov<kDppXor1>(v));
    v = fmaxf(v, dpp_mov<kDppXor2>(v));
    v = fmaxf(v, dpp_mov<kDppHalfMirror>(v));
    v = fmaxf(v, dpp_mov<kDppMirror>(v));
    return v;
}
// combine lane l with lanes l^16, l^32, l^48 (the same column of the four rows); result in every lane
__device__ __forceinline__ float rows4_sum(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ int rows4_sum(int v) {
    const auto a = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    v = (int)a[0] + (int)a[1];
    const auto b = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return (int)b[0] + (int)b[1];
}
__device__ __forceinline__ float rows4_max(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

__device__ __forceinline__ float wave_sum(float v) { return rows4_sum(row16_sum(v)); }
__device__ __forceinline__ int wave_sum(int v) { return rows4_sum(row16_sum(v)); }
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {            // 64-bit types: the generic LDS-crossbar shuffle
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) { return rows4_max(row16_max(v)); }

}  // namespace geomae
