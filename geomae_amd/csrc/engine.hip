// The pre-training step as ONE C call: geomae_pretrain_step enqueues every kernel of
//   forward_train (ssl.py:126-242) + backward + clip + AdamW (configs/_base_/schedules/cosine_2x.py:1-17)
// on three HIP streams, from a caller-provided workspace, with one host wait (the pillar-count readback that the
// previous step already enqueued).  Same kernels, same order, same cross-stream events as the Python explicit
// schedule it replaces (geomae_amd/detector.py train_step_explicit + train.py Trainer.train_step): that schedule
// needed 1.2-1.6 ms of Python / ctypes / allocator time per 2.4 ms step on a 5 GHz host and was host-bound on
// anything slower (round-1 driver run: 3.6 ms/step with identical kernel durations).
//
// Streams (the caller's `stream` = main, plus the two side streams given at creation):
//   main     zero arena -> VFE forward -> [wait layouts] -> encoder -> decoder A -> its heads + losses -> decoder A backward ->
//            [join B] -> encoder backward -> VFE backward -> [join geometry] -> clip + AdamW
//   geometry zero arena of everything behind the VFE forward -> token coordinates -> four window layouts; later every
//            weight-gradient contraction (heads, the three stacks, VFE layer 1) and, at world > 1, the gradient-exchange hooks
//   dec_b    geometric targets -> NEXT batch's stage 1 -> [fork behind the encoder] decoder B forward -> its head + loss ->
//            decoder B backward -> (after the optimizer) bf16 re-pack of the updated weights
// Every wait of the main stream on another stream's event puts a barrier packet into its queue: 6-8 us even when the event
// completed long ago (the neighbouring kernels no longer overlap) -- three are left per step.
#include "common.h"
#include "../../include/geomae_hip.h"
#include <chrono>
#include <new>
#include <vector>

namespace geomae {
// window.hip: kept pillars in the fullest window of the unshifted / shifted layout -> out[2] (pinned host memory)
int window_max_keep(const int32_t* ids_keep, const int32_t* counts, const int32_t* voxel_coors, int batch_size,
                    const GeomaeWindowConfig* cfg, int32_t* out, hipStream_t stream);
namespace {

__global__ void bn_param_grad_add_kernel(const double* __restrict__ bsums, int C, float* __restrict__ d_beta,
                                         float* __restrict__ d_gamma) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        d_beta[c] += (float)bsums[c];
        d_gamma[c] += (float)bsums[C + c];
    }
}
// token_row / counts for a caller-supplied mask (geomae_pretrain_set_mask): what random_mask_kernel writes beside the ids
__global__ void token_rows_from_ids_kernel(const int32_t* __restrict__ ids_keep, int n_keep,
                                           const int32_t* __restrict__ ids_mask, int n_mask,
                                           int32_t* __restrict__ token_row, int32_t* __restrict__ counts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { counts[0] = n_keep; counts[1] = n_mask; }
    if (i < n_keep) token_row[ids_keep[i]] = i;
    else if (i < n_keep + n_mask) token_row[ids_mask[i - n_keep]] = i;
}
// arena clears: one launch of our own per arena (hipMemsetAsync is a runtime kernel of the same cost; this one can be
// seen and accounted for in the profile like every other kernel of the step)
__global__ __launch_bounds__(256) void zero_arena_kernel(uint4* __restrict__ p, int64_t n16) {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) p[i] = z;
}
int zero_arena(void* p, int64_t bytes, hipStream_t stream) {
    if (bytes <= 0) return GEOMAE_OK;
    if ((bytes & 15) || ((uintptr_t)p & 15)) { GEOMAE_HIP(hipMemsetAsync(p, 0, (size_t)bytes, stream)); return GEOMAE_OK; }
    hipLaunchKernelGGL(zero_arena_kernel, dim3(stream_grid(bytes / 16, 256)), dim3(256), 0, stream, (uint4*)p, bytes / 16);
    return check_launch("zero_arena_kernel");
}

inline int64_t al256(int64_t b) { return (b + 255) / 256 * 256; }
// the distributed schedule (hooks, separate optimizer call): world_size > 1, or forced at world size 1 (exchange_always:
// how the RCCL code path is exercised on a one-GPU box)
inline bool exchanges(const GeomaePretrainConfig& c) { return c.world_size > 1 || c.exchange_always != 0; }

struct Arena {
    char* base = nullptr;
    int64_t cap = 0, off = 0;
    bool overflow = false;
    void reset(char* b, int64_t c) { base = b; cap = c; off = 0; overflow = false; }
    char* bytes(int64_t n) {
        n = al256(n < 1 ? 1 : n);
        if (off + n > cap) { overflow = true; off += n; return base; }
        char* p = base + off;
        off += n;
        return p;
    }
    template <typename T> T* take(int64_t count) { return reinterpret_cast<T*>(bytes(count * (int64_t)sizeof(T))); }
};

// stage 1 of one batch: everything that depends on the points only
struct Batch {
    bool valid = false, counts_read = false;
    int64_t N = 0;
    int32_t cap = 0;
    float* points = nullptr;
    int32_t *boffs = nullptr, *coors_top = nullptr, *coors_med = nullptr, *coors_low = nullptr;
    int32_t *cell_table = nullptr, *voxel_coors = nullptr, *inv = nullptr, *order = nullptr, *seg_start = nullptr;
    int32_t *sample_start = nullptr, *num_pillars = nullptr;
    float *mean = nullptr, *feat = nullptr;
    double* moments = nullptr;         // feature moments of the batch (geomae_vfe_prepare_moments)
    int32_t* pid = nullptr;
    int32_t *ids_keep = nullptr, *ids_mask = nullptr, *token_row = nullptr, *counts = nullptr;
    int32_t* host_offs = nullptr;      // pinned [B + 1]
    int32_t* host_counts = nullptr;    // pinned [B + 1]
    hipEvent_t readback = nullptr;
    int32_t* host_maxkeep = nullptr;   // pinned [2]: kept pillars in the fullest window, unshifted / shifted (window_max_keep)
    hipEvent_t readback2 = nullptr;    // ... written behind the random mask of stage 1
    bool mask_injected = false;        // geomae_pretrain_set_mask replaced the drawn mask: host_maxkeep says nothing about it
    int32_t V = 0, n_keep = 0, n_mask = 0;
    bool moments_exchanged = false;    // the rank-averaged feature moments of this batch are in bn_sync_feat_moments[slot]
};

struct WinLayout {
    int32_t n = 0, max_windows = 1;
    int32_t *win_start, *win_tokens, *tok_win, *tok_pos, *num_windows, *bun_start, *num_bundles, *bun_tok, *pos_info;
    int32_t *fbun_tok, *num_fbundles;
    int32_t *fitems, *num_fitems;
};

enum Ev { kStepEnd, kLayouts, kVfeDone, kForkDec, kJoinDecFwd, kHeads, kAuxBwd, kMainDecBwd, kEncBwd, kVfeL1, kGeoDone,
          kPacked, kFirstMain, kNextReady, kMoments0, kMoments1, kZeroLate, kEncMid, kDecMidA, kDecMidB, kVfeSide, kNumEv };
enum Phase { pStart, pVfeFwd, pLayouts, pEncFwd, pDecFwd, pHeads, pDecBwd, pEncBwd, pVfeStats, pVfeL1, pVfeL0, pVfeBwd, pOpt, kNumPhase };

struct Engine {
    GeomaePretrainConfig cfg;
    GeomaePretrainModel m;
    std::vector<GeomaeSstLayerWeights> layers;
    std::vector<GeomaeSstLayerGrads> grads;
    char* ws = nullptr;
    int64_t ws_bytes = 0;
    // persistent carve
    double* sumsq_ring = nullptr;      // [2]
    float* gnorm = nullptr;            // [1]
    float* losses_ring = nullptr;      // [4][8]
    int64_t persistent_bytes = 0, stage_bytes = 0;
    Batch batch[2];
    int pending = -1;                  // index of the batch submitted last, -1 = none
    uint64_t mask_draws = 0;           // steps begun so far: the batch consumed by step i (0-based) drew mask i + 1
    hipStream_t geo = nullptr, aux = nullptr;
    hipEvent_t ev[kNumEv];
    hipEvent_t phase_ev[kNumPhase];
    bool phase_timing = false;
    int phase_last = -1;              // last phase event recorded by the most recent step, -1 = none
    bool have_step_end = false, packed_fresh = false;
    int64_t opt_steps = 0, steps = 0;
    int sumsq_slot = 0;
    void* profiler = nullptr;
    GeomaePretrainHook hook = nullptr;
    void* hook_user = nullptr;
    // results of the last step
    int64_t last_N = 0;
    int32_t last_V = 0, last_keep = 0, last_mask = 0;
    int64_t off_losses = 0, off_ids_keep = 0, off_ids_mask = 0, off_voxel_coors = 0;
    int32_t cells = 0, gz = 1, gy = 1, gx = 1, s_low = 1, s_med = 1;
    int last_forms[6] = {-1, -1, -1, -1, -1, -1};           // GEOMAE_STACK_FORM_* of the last step's stacks: forward enc / den / cen, backward
    int last_maxkeep[2] = {-1, -1}, last_big_layouts = 3;   // the last step's fullest windows / which layouts took the second launch
    double host_step_s = 0.0, host_blocked_s = 0.0;      // cumulative wall time inside step calls / in the readback wait
    // a step that fails AFTER its first launch leaves streams, events and the workspace in an undefined state
    bool enqueue_started = false, poisoned = false;
};

inline double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int64_t stage_region_bytes(const GeomaePretrainConfig& c, int64_t N) {
    const int64_t cells = (int64_t)c.batch_size * c.targets.grid_size[0] * c.targets.grid_size[1] * c.targets.grid_size[2];
    const int64_t cap = (N < cells ? N : cells) < 1 ? 1 : (N < cells ? N : cells);
    int64_t b = 0;
    b += al256(N * c.num_features * 4);                 // points
    b += al256((c.batch_size + 1) * 4);                 // boffs
    b += 3 * al256(N * 16);                             // coors x3
    b += al256(cells * 4);                              // cell table
    b += al256(cap * 16) + 2 * al256(N * 4) + al256((cap + 1) * 4) + al256((c.batch_size + 1) * 4) + 256;
    b += al256(geomae_pillar_segment_workspace_bytes(N, c.batch_size, c.targets.grid_size[0], c.targets.grid_size[1],
                                                     c.targets.grid_size[2]));
    b += al256(cap * 12) + al256(N * 64) + al256(N * 4);   // mean, feat, pid
    b += al256(144 * 8) + al256(geomae_vfe_moments_workspace_bytes());   // feature moments + their partial sums
    b += 3 * al256(cap * 4) + 256;                      // mask
    return b + 4096;
}

int64_t window_layout_bytes(const GeomaePretrainConfig& c, int64_t n) {
    const int nwx = (c.window.bev_shape[0] + c.window.window_shape[0] - 1) / c.window.window_shape[0] + 1;
    const int nwy = (c.window.bev_shape[1] + c.window.window_shape[1] - 1) / c.window.window_shape[1] + 1;
    int64_t slots = (int64_t)c.batch_size * nwx * nwy;
    int64_t mw = n < slots ? n : slots;
    if (mw < 1) mw = 1;
    const int64_t n1 = n < 1 ? 1 : n;
    return 4 * al256((mw + 1) * 4) + 3 * al256(n1 * 4) + 3 * 256 + al256(n1 * 16);
}

int64_t step_region_bytes(const GeomaePretrainConfig& c, int64_t N, int64_t V) {
    const int64_t s_low = (int64_t)c.targets.ratio_low[0] * c.targets.ratio_low[1] * c.targets.ratio_low[2];
    const int64_t s_med = (int64_t)c.targets.ratio_med[0] * c.targets.ratio_med[1] * c.targets.ratio_med[2];
    const int64_t n = V < 1 ? 1 : V, M = n, nk = n;
    int64_t b = 0;
    b += al256(256 * 8) + al256(128 * 8) + 3 * al256(n * 512) + al256(n * 512) + al256(n * 256) + al256(4096);   // zeros_late
    b += 6 * al256(128 * 4) + al256(256 * 4) * 2 + al256(128 * 8) + al256(n * 256) + al256(256 * 8) + al256(n * 512) + al256(n);  // zeros_fwd
    b += al256(n * 16);                                                                                    // coors_all
    b += 2 * window_layout_bytes(c, nk) + 2 * window_layout_bytes(c, n);
    int32_t ns[4] = {(int32_t)nk, (int32_t)nk, (int32_t)n, (int32_t)n};
    b += al256(geomae_window_build_batch_workspace_bytes(ns, 4, c.batch_size, &c.window));
    b += al256(M * s_low * 12) + al256(M * s_low) + al256(M * s_med * 12) + al256(M * s_med) + 2 * al256(M * 12) +
         al256(M * 24) + al256(n * 12) + al256(n * s_med * 12) + al256(n * s_med) + al256(M * 24) + 256;   // targets
    b += al256(geomae_sst_stack_saved_bytes((int32_t)nk, c.encoder_layers, c.num_heads)) +
         2 * al256(geomae_sst_stack_saved_bytes((int32_t)n, c.decoder_layers, c.num_heads));
    b += al256(geomae_sst_stack_scratch_bytes_layers((int32_t)nk, c.encoder_layers)) + 2 * al256(geomae_sst_stack_scratch_bytes_layers((int32_t)n, c.decoder_layers));
    b += al256(nk * 512) + 4 * al256(n * 512);                                                             // z_enc, cen, den, dxa, dxb
    b += al256(M * 896 * 2) + 2 * al256(M * 128 * 2);                                                      // heads
    b += 2 * al256((N + 15) / 16 * 16 * 128 * 2) + al256(N * 64 * 4) + al256(2 * kDwPartialBytes);                         // VFE backward
    return b + 8192;
}

constexpr int64_t kPersistentBytes = 4096;

struct Prezeroed {
    bool was;
    Prezeroed() : was(accumulators_prezeroed()) { geomae_set_accumulators_prezeroed(1); }
    ~Prezeroed() { geomae_set_accumulators_prezeroed(was ? 1 : 0); }
};

#define ENG_CALL(expr)                 \
    do {                               \
        int rc_ = (expr);              \
        if (rc_ != GEOMAE_OK) return rc_; \
    } while (0)

inline int order_after(Engine* e, Ev which, hipStream_t producer, hipStream_t consumer) {
    GEOMAE_HIP(hipEventRecord(e->ev[which], producer));
    GEOMAE_HIP(hipStreamWaitEvent(consumer, e->ev[which], 0));
    return GEOMAE_OK;
}

inline void mark(Engine* e, Phase p, hipStream_t s) {
    if (e->phase_timing) { (void)hipEventRecord(e->phase_ev[p], s); e->phase_last = p; }
}

// stage 1 of a batch on `s` (voxelize x3 -> pillar sort -> count readback -> VFE front -> random mask)
int run_stage1(Engine* e, int which, const float* const* frames, const int64_t* sizes, uint64_t draw, hipStream_t s) {
    const GeomaePretrainConfig& c = e->cfg;
    Batch& b = e->batch[which];
    b.valid = false;
    GEOMAE_REQUIRE(frames && sizes, "pretrain: null frame list");
    int64_t N = 0;
    for (int i = 0; i < c.batch_size; ++i) {
        GEOMAE_REQUIRE(sizes[i] >= 0 && (frames[i] || sizes[i] == 0), "pretrain: bad frame %d", i);
        b.host_offs[i] = (int32_t)N;
        N += sizes[i];
    }
    b.host_offs[c.batch_size] = (int32_t)N;
    GEOMAE_REQUIRE(N > 0 && N < (int64_t)1 << 31, "pretrain: empty or oversized batch");
    Arena a;
    a.reset(e->ws + e->persistent_bytes + which * e->stage_bytes, e->stage_bytes);
    const int64_t cap = N < e->cells ? N : e->cells;
    b.N = N;
    b.cap = (int32_t)cap;
    b.points = a.take<float>(N * c.num_features);
    b.boffs = a.take<int32_t>(c.batch_size + 1);
    b.coors_top = a.take<int32_t>(N * 4);
    b.coors_med = a.take<int32_t>(N * 4);
    b.coors_low = a.take<int32_t>(N * 4);
    b.cell_table = a.take<int32_t>(e->cells);
    b.voxel_coors = a.take<int32_t>(cap * 4);
    b.inv = a.take<int32_t>(N);
    b.order = a.take<int32_t>(N);
    b.seg_start = a.take<int32_t>(cap + 1);
    b.sample_start = a.take<int32_t>(c.batch_size + 1);
    b.num_pillars = a.take<int32_t>(1);
    const int64_t wsb = geomae_pillar_segment_workspace_bytes(N, c.batch_size, e->gz, e->gy, e->gx);
    char* seg_ws = a.bytes(wsb);
    b.mean = a.take<float>(cap * 3);
    b.feat = a.take<float>(N * 16);
    b.pid = a.take<int32_t>(N);
    b.moments = a.take<double>(144);
    char* mom_ws = a.bytes(geomae_vfe_moments_workspace_bytes());
    b.ids_keep = a.take<int32_t>(cap);
    b.ids_mask = a.take<int32_t>(cap);
    b.token_row = a.take<int32_t>(cap);
    b.counts = a.take<int32_t>(2);
    if (a.overflow) {
        set_error("pretrain: a batch of %lld points does not fit the workspace (stage region %lld bytes)", (long long)N,
                  (long long)e->stage_bytes);
        return GEOMAE_ERR_WORKSPACE;
    }
    // the reference concatenates the per-sample tensors (ssl.py:320-329 torch.cat) and voxelizes three times; here ONE
    // launch reads the frames where they lie, writes the concatenated rows + the three coordinate arrays and clears the
    // pillar table and the scan state on the side; the pillar sort is three more launches, and the per-sample pillar
    // counts land in pinned host memory from the scan kernel itself (no copy command).  Round 2: B + 1 copies, a
    // memset, voxelize, five segment kernels, a device-to-host copy.
    const int64_t scan_state = geomae_pillar_segment_scan_state_bytes(c.batch_size, e->gz, e->gy, e->gx);
    if (c.batch_size <= 32) {
        ENG_CALL(geomae_voxelize_frames3(frames, sizes, c.batch_size, c.num_features, c.targets.voxel_size_top,
                                         c.targets.voxel_size_med, c.targets.voxel_size_low, c.targets.coors_range, b.points,
                                         b.boffs, b.coors_top, b.coors_med, b.coors_low, b.cell_table,
                                         al256((int64_t)e->cells * 4), seg_ws, scan_state, s));
        ENG_CALL(geomae_pillar_segment_ex(b.coors_top, 4, N, c.batch_size, e->gz, e->gy, e->gx, b.cell_table, b.voxel_coors,
                                          b.inv, b.order, b.seg_start, b.sample_start, b.num_pillars, seg_ws, wsb,
                                          b.host_counts, 1, s));
    } else {
        for (int i = 0; i < c.batch_size; ++i)
            if (sizes[i] > 0)
                GEOMAE_HIP(hipMemcpyAsync(b.points + (int64_t)b.host_offs[i] * c.num_features, frames[i],
                                          (size_t)sizes[i] * c.num_features * 4, hipMemcpyDeviceToDevice, s));
        GEOMAE_HIP(hipMemcpyAsync(b.boffs, b.host_offs, (size_t)(c.batch_size + 1) * 4, hipMemcpyHostToDevice, s));
        ENG_CALL(geomae_voxelize_batch3(b.points, N, c.num_features, b.boffs, c.batch_size, c.targets.voxel_size_top,
                                        c.targets.voxel_size_med, c.targets.voxel_size_low, c.targets.coors_range,
                                        b.coors_top, b.coors_med, b.coors_low, s));
        ENG_CALL(geomae_pillar_segment(b.coors_top, N, c.batch_size, e->gz, e->gy, e->gx, b.cell_table, b.voxel_coors, b.inv,
                                       b.order, b.seg_start, b.sample_start, b.num_pillars, seg_ws, wsb, s));
        GEOMAE_HIP(hipMemcpyAsync(b.host_counts, b.sample_start, (size_t)(c.batch_size + 1) * 4, hipMemcpyDeviceToHost, s));
    }
    GEOMAE_HIP(hipEventRecord(b.readback, s));
    ENG_CALL(geomae_segment_mean_xyz_sorted(b.points, c.num_features, b.order, b.seg_start, b.num_pillars, b.cap, b.mean, s));
    ENG_CALL(geomae_vfe_prepare_moments(b.points, c.num_features, N, b.order, b.inv, b.mean, b.voxel_coors, c.vfe_voxel_size,
                                        c.vfe_center_offset, b.feat, b.pid, mom_ws, b.moments, s));
    // the mask index is a function of the step that will consume the batch (not of how many stage 1s ran: a replaced
    // submission or a re-created engine must not shift the stream; geomae_pretrain_set_mask_draws)
    // window-major token lists: a 16-token tile of the stacks' activations then belongs to one or two attention windows
    ENG_CALL(geomae_random_mask_windowed(b.sample_start, c.batch_size, c.keep_fraction, (c.mask_seed << 32) + draw,
                                         b.voxel_coors, &c.window, b.ids_keep, b.ids_mask, b.token_row, b.counts, s));
    // the fullest window's kept pillars, both layouts, into pinned memory: by the time this batch is consumed the host knows
    // whether the one-launch encoder layers need their second launch (window.hip window_max_keep)
    ENG_CALL(window_max_keep(b.ids_keep, b.counts, b.voxel_coors, c.batch_size, &c.window, b.host_maxkeep, s));
    GEOMAE_HIP(hipEventRecord(b.readback2, s));
    b.mask_injected = false;
    b.valid = true;
    b.counts_read = false;
    return GEOMAE_OK;
}

// naiveSyncBN1d of the first VFE layer, exchanged AHEAD: y0 = W0 f is linear in the point features, so the equal-weight
// rank average of (mean y0, mean y0^2) is W0 applied to the rank average of the per-rank NORMALISED feature moments
// (S1 / N_r, S2 / N_r).  Those depend on the batch only -- not on the weights -- and are complete with its stage 1, one
// step before the VFE forward that needs them: the all-reduce leaves the main stream's critical path (the in-line form,
// BN_FWD0, stops the VFE forward for a collective round trip).  Raised behind every other hook of the step, so the
// process group sees the same order of collectives on every rank.
__global__ void scale_moments_kernel(const double* __restrict__ in, double* __restrict__ out, double inv, int count) {
    const int i = threadIdx.x;
    out[i] = i < count ? in[i] * inv : 0.0;
}

constexpr int kFeatMoments = 144;      // doubles per slot (csrc/vfe.hip: S1 padded to 16, S2 [11][11], padding)

int exchange_moments(Engine* e, int which, hipStream_t s) {
    const GeomaePretrainConfig& c = e->cfg;
    Batch& b = e->batch[which];
    b.moments_exchanged = false;
    if (!exchanges(c) || !c.sync_bn || !e->m.bn_sync_feat_moments) return GEOMAE_OK;
    GEOMAE_REQUIRE(e->hook, "pretrain: world_size > 1 needs a hook");
    double* buf = e->m.bn_sync_feat_moments + (int64_t)kFeatMoments * which;
    hipLaunchKernelGGL(scale_moments_kernel, dim3(1), dim3(kFeatMoments), 0, s, (const double*)b.moments, buf,
                       1.0 / ((double)b.N * (double)c.world_size), 16 + 121);
    GEOMAE_HIP(hipGetLastError());
    e->hook(e->hook_user, which == 0 ? GEOMAE_HOOK_FEAT_MOMENTS0 : GEOMAE_HOOK_FEAT_MOMENTS1, s);
    GEOMAE_HIP(hipEventRecord(e->ev[which == 0 ? kMoments0 : kMoments1], s));
    b.moments_exchanged = true;
    return GEOMAE_OK;
}

int read_counts(Engine* e, Batch& b) {
    if (b.counts_read) return GEOMAE_OK;
    const double t0 = now_s();
    GEOMAE_HIP(hipEventSynchronize(b.readback));
    e->host_blocked_s += now_s() - t0;
    const int B = e->cfg.batch_size;
    b.V = b.host_counts[B];
    int nk = 0;
    for (int i = 0; i < B; ++i) nk += (int)((double)(b.host_counts[i + 1] - b.host_counts[i]) * e->cfg.keep_fraction);
    b.n_keep = nk;
    b.n_mask = b.V - nk;
    b.counts_read = true;
    return GEOMAE_OK;
}

void carve_layout(Arena& a, const GeomaePretrainConfig& c, int32_t n, WinLayout* L) {
    const int nwx = (c.window.bev_shape[0] + c.window.window_shape[0] - 1) / c.window.window_shape[0] + 1;
    const int nwy = (c.window.bev_shape[1] + c.window.window_shape[1] - 1) / c.window.window_shape[1] + 1;
    const int64_t slots = (int64_t)c.batch_size * nwx * nwy;
    int64_t mw = n < slots ? n : slots;
    if (mw < 1) mw = 1;
    const int64_t n1 = n < 1 ? 1 : n;
    L->n = n;
    L->max_windows = (int32_t)mw;
    L->win_start = a.take<int32_t>(mw + 1);
    L->bun_start = a.take<int32_t>(mw + 1);
    L->bun_tok = a.take<int32_t>(mw + 1);
    L->win_tokens = a.take<int32_t>(n1);
    L->tok_win = a.take<int32_t>(n1);
    L->tok_pos = a.take<int32_t>(n1);
    L->num_windows = a.take<int32_t>(1);
    L->num_bundles = a.take<int32_t>(1);
    L->pos_info = a.take<int32_t>(n1 * 4);
    L->fbun_tok = a.take<int32_t>(mw + 1);
    L->num_fbundles = a.take<int32_t>(1);
    L->fitems = a.take<int32_t>(8 * (mw + 1));       // [2 (mw + 1)][4]: the one-launch forward's work items
    L->num_fitems = a.take<int32_t>(1);
}

void stack_layouts(const WinLayout* L, GeomaeSstStackLayout* out) {
    for (int i = 0; i < 2; ++i) {
        out[i].win_start = L[i].win_start; out[i].win_tokens = L[i].win_tokens; out[i].tok_win = L[i].tok_win;
        out[i].tok_pos = L[i].tok_pos; out[i].bun_start = L[i].bun_start; out[i].num_bundles = L[i].num_bundles;
        out[i].max_bundles = L[i].max_windows; out[i].bun_tok = L[i].bun_tok; out[i].pos_info = L[i].pos_info;
        out[i].fbun_tok = L[i].fbun_tok; out[i].num_fbundles = L[i].num_fbundles;
        out[i].fitems = L[i].fitems; out[i].num_fitems = L[i].num_fitems;
    }
}

int pack_weights(Engine* e, hipStream_t s) {
    return geomae_pack_weights(nullptr, e->m.pack_desc, e->m.num_pack_desc, e->m.pack_max_elems, e->m.packed,
                               e->m.pack_aux, s);
}

// BatchNorm statistics -> folded scale / shift (+ running stats); naiveSyncBN1d's equal-weight cross-rank average of
// (mean, mean of squares) at world > 1 (mmdet3d/ops/norm.py:64-76: biased running variance, no batch counter)
int bn_forward(Engine* e, int layer, const double* sums, double count, float* scale, float* shift, float* invstd,
               float* moments, hipStream_t s) {
    const GeomaePretrainConfig& c = e->cfg;
    const GeomaePretrainModel& m = e->m;
    const int C = layer == 0 ? 64 : 128;
    if (!exchanges(c) || !c.sync_bn)
        return geomae_bn_finalize(sums, count, nullptr, C, m.bn_gamma[layer], m.bn_beta[layer], c.bn_eps, c.bn_momentum, 1,
                                  m.bn_running_mean[layer], m.bn_running_var[layer], scale, shift, invstd, moments,
                                  m.bn_num_batches[layer], s);
    float* mom = layer == 0 ? m.bn_sync_moments0 : m.bn_sync_moments1;
    GEOMAE_REQUIRE(mom && e->hook, "pretrain: world_size > 1 needs bn_sync buffers and a hook");
    // local (mean, mean of squares) already divided by the world size (count * world as the divisor), so that the SUM
    // all-reduce delivers naiveSyncBN1d's equal-weight average and no scaling kernel sits behind the collective
    ENG_CALL(geomae_bn_finalize(sums, count * (double)c.world_size, nullptr, C, nullptr, nullptr, 0.f, 0.f, 0, nullptr, nullptr,
                                nullptr, nullptr, nullptr, mom, nullptr, s));
    e->hook(e->hook_user, layer == 0 ? GEOMAE_HOOK_BN_FWD0 : GEOMAE_HOOK_BN_FWD1, s);
    return geomae_bn_finalize(nullptr, count, mom, C, m.bn_gamma[layer], m.bn_beta[layer], c.bn_eps, c.bn_momentum, 0,
                              m.bn_running_mean[layer], m.bn_running_var[layer], scale, shift, invstd, moments, nullptr, s);
}

int run_optimizer(Engine* e, float lr, float grad_scale, hipStream_t main) {
    const GeomaePretrainConfig& c = e->cfg;
    const GeomaePretrainModel& m = e->m;
    Prezeroed pz;
    e->opt_steps += 1;
    double* cur = e->sumsq_ring + e->sumsq_slot;
    double* nxt = e->sumsq_ring + (1 - e->sumsq_slot);
    e->sumsq_slot = 1 - e->sumsq_slot;
    ENG_CALL(geomae_grad_sumsq(m.grads, m.num_params, cur, main));
    ENG_CALL(geomae_adamw_step(m.params, m.grads, m.exp_avg, m.exp_avg_sq, m.num_params, m.no_decay_prefix, lr, c.beta1,
                               c.beta2, c.adam_eps, c.weight_decay, e->opt_steps, c.max_grad_norm, cur, grad_scale, 1,
                               e->gnorm, m.no_decay2_start, m.no_decay2_count, nxt, main));
    mark(e, pOpt, main);
    // bf16 MFMA-layout copies of the updated weights for the next step, off its critical path: on the decoder-B
    // stream, idle until the next step's VFE forward is done
    GEOMAE_HIP(hipEventRecord(e->ev[kStepEnd], main));
    GEOMAE_HIP(hipStreamWaitEvent(e->aux, e->ev[kStepEnd], 0));
    ENG_CALL(pack_weights(e, e->aux));
    GEOMAE_HIP(hipEventRecord(e->ev[kPacked], e->aux));
    e->have_step_end = true;
    e->packed_fresh = true;
    return GEOMAE_OK;
}

int run_step(Engine* e, const float* const* next_frames, const int64_t* next_sizes, float lr, float grad_scale,
             int do_opt, hipStream_t main) {
    const GeomaePretrainConfig& c = e->cfg;
    const GeomaePretrainModel& m = e->m;
    GEOMAE_REQUIRE(e->pending >= 0 && e->batch[e->pending].valid, "pretrain_step: no batch submitted");
    Batch& b = e->batch[e->pending];
    ENG_CALL(read_counts(e, b));
    const int64_t N = b.N;
    const int32_t V = b.V, nk = b.n_keep, nm = b.n_mask, n = nk + nm;
    GEOMAE_REQUIRE(V >= 1 && nk >= 1 && nm >= 1, "pretrain_step: degenerate batch (V=%d keep=%d mask=%d)", V, nk, nm);
    if (next_frames) {
        GEOMAE_REQUIRE(next_sizes, "pretrain_step: next_frame_sizes is null");
        int64_t nn = 0;
        for (int i = 0; i < c.batch_size; ++i) nn += next_sizes[i] > 0 ? next_sizes[i] : 0;
        if (stage_region_bytes(c, nn) > e->stage_bytes) {
            set_error("pretrain_step: the next batch (%lld points) does not fit the workspace", (long long)nn);
            return GEOMAE_ERR_WORKSPACE;
        }
    }
    hipStream_t geo = e->geo, aux = e->aux;
    const int nh = c.num_heads, ne = c.encoder_layers, nd = c.decoder_layers;
    drop_pending_weight_grads();       // (a step that failed between queueing and flushing leaves entries into a dead workspace)
    struct ThreadProfilerScope {       // the contractions launched by geomae_flush_weight_grad are timed where they run
        explicit ThreadProfilerScope(void* p) { set_thread_profiler(p); }
        ~ThreadProfilerScope() { set_thread_profiler(nullptr); }
    } prof_scope(e->profiler);

    // ---------------- carve the step's buffers (sizes are all known on the host)
    Arena a;
    a.reset(e->ws + e->persistent_bytes + 2 * e->stage_bytes, e->ws_bytes - e->persistent_bytes - 2 * e->stage_bytes);
    // zero arena of everything behind the VFE forward (one fill on the decoder-B stream)
    char* zl0 = a.base + a.off;
    double* bs1 = a.take<double>(256);
    double* bs0 = a.take<double>(128);
    float* d_cen = a.take<float>((int64_t)n * 128);
    float* d_cen2 = a.take<float>((int64_t)n * 128);       // second summand of d_cen (geomae_heads_loss_split_accumulate)
    float* d_den = a.take<float>((int64_t)n * 128);
    float* d_vf = a.take<float>((int64_t)V * 128);
    float* dm0 = a.take<float>((int64_t)V * 64);
    float* dw0_acc = a.take<float>(64 * 16);               // layer-0 weight-gradient contraction of the VFE backward
    const int64_t zl_bytes = (a.base + a.off) - zl0;
    // zero arena of the VFE forward (one fill on the main stream)
    char* zf0 = a.base + a.off;
    float* bn_scale0 = a.take<float>(128); float* bn_shift0 = a.take<float>(128); float* bn_invstd0 = a.take<float>(128);
    float* bn_mom0 = a.take<float>(256);
    float* bn_scale1 = a.take<float>(128); float* bn_shift1 = a.take<float>(128); float* bn_invstd1 = a.take<float>(128);
    float* bn_mom1 = a.take<float>(256);
    double* sums0 = a.take<double>(128);
    float* m0 = a.take<float>((int64_t)V * 64);
    double* sums1 = a.take<double>(256);
    float* vf = a.take<float>((int64_t)V * 128);
    uint8_t* vfe_ties = a.take<uint8_t>((int64_t)V);       // pillars whose maximum two points may share (GeomaeVfeArgs.pillar_ties)
    const int64_t zf_bytes = (a.base + a.off) - zf0;
    int32_t* coors_all = a.take<int32_t>((int64_t)n * 4);
    WinLayout lay[4];
    carve_layout(a, c, nk, &lay[0]);
    carve_layout(a, c, nk, &lay[1]);
    carve_layout(a, c, n, &lay[2]);
    carve_layout(a, c, n, &lay[3]);
    int32_t ns[4] = {nk, nk, n, n};
    const int64_t win_wsb = geomae_window_build_batch_workspace_bytes(ns, 4, c.batch_size, &c.window);
    GEOMAE_REQUIRE(win_wsb >= 0, "pretrain_step: bad window configuration");
    char* win_ws = a.bytes(win_wsb);
    const int64_t M = nm, SL = e->s_low, SM = e->s_med;
    float* t_clow = a.take<float>(M * SL * 3);
    uint8_t* t_mlow = a.take<uint8_t>(M * SL);
    float* t_cmed = a.take<float>(M * SM * 3);
    uint8_t* t_mmed = a.take<uint8_t>(M * SM);
    float* t_ctop = a.take<float>(M * 3);
    float* t_normal = a.take<float>(M * 3);
    double* t_curv = a.take<double>(M * 3);
    float* t_top_raw = a.take<float>((int64_t)V * 3);
    float* t_med_raw = a.take<float>((int64_t)V * SM * 3);
    uint8_t* t_med_raw_mask = a.take<uint8_t>((int64_t)V * SM);
    float* t_cov = a.take<float>(M * 6);
    int32_t* t_occ = a.take<int32_t>(2);
    const int64_t sb_enc = geomae_sst_stack_saved_bytes(nk, ne, nh), sb_dec = geomae_sst_stack_saved_bytes(n, nd, nh);
    const int64_t wb_enc = geomae_sst_stack_scratch_bytes_layers(nk, ne), wb_dec = geomae_sst_stack_scratch_bytes_layers(n, nd);
    // The weight-gradient contractions of the two DECODER stacks run on the geometry stream after their stack's backward,
    // beside the encoder backward (GEOMAE_DW_DEFER_ALL=0: riding in the stacks' ffn-backward launches, round 2's form).
    // The decoder backward is at the memory system's roof (two stacks, 7 TB/s between L2 and fabric): without the
    // contractions' 4 KB per token and layer its phase is 0.49 -> 0.41 ms, while the encoder backward -- one workgroup
    // chain per CU, half the chip idle -- takes them in for +0.02 ms.  The ENCODER's own contractions keep riding: queued
    // behind its backward they are twelve launches of one round each, 0.11 ms that nothing is left to hide.
    const bool defer_dec_dw = tuning().dw_defer_all != 0;
    struct DeferAllScope {
        explicit DeferAllScope(bool on) { set_defer_all_weight_grads(on); }
        ~DeferAllScope() { set_defer_all_weight_grads(false); }
    };
    char* s_enc = a.bytes(sb_enc); char* s_cen = a.bytes(sb_dec); char* s_den = a.bytes(sb_dec);
    char* w_enc = a.bytes(wb_enc); char* w_cen = a.bytes(wb_dec); char* w_den = a.bytes(wb_dec);
    float* z_enc = a.take<float>((int64_t)nk * 128);
    float* cen = a.take<float>((int64_t)n * 128);
    float* den = a.take<float>((int64_t)n * 128);
    float* dxa = a.take<float>((int64_t)n * 128);
    float* dxb = a.take<float>((int64_t)n * 128);
    char* h_dl = a.bytes(M * 896 * 2); char* h_cm = a.bytes(M * 128 * 2); char* h_dm = a.bytes(M * 128 * 2);
    char* dy1_b = a.bytes((N + 15) / 16 * 16 * 128 * 2); char* g_b = a.bytes((N + 15) / 16 * 16 * 128 * 2);   // tile-blocked
    float* dw1_partial = (float*)a.bytes(2 * kDwPartialBytes);      // split-K workspace of the layer-1 weight gradient
    float* dh0 = a.take<float>(N * 64);
    if (a.overflow) {
        set_error("pretrain_step: N=%lld V=%d needs %lld bytes of step workspace, %lld available", (long long)N, V,
                  (long long)a.off, (long long)a.cap);
        return GEOMAE_ERR_WORKSPACE;
    }
    float* losses = e->losses_ring + 8 * (e->steps & 3);
    e->off_losses = (char*)losses - e->ws;
    e->off_ids_keep = (char*)b.ids_keep - e->ws;
    e->off_ids_mask = (char*)b.ids_mask - e->ws;
    e->off_voxel_coors = (char*)b.voxel_coors - e->ws;
    e->last_N = N; e->last_V = V; e->last_keep = nk; e->last_mask = nm;
    e->steps += 1;
    e->mask_draws += 1;

    Prezeroed pz;
    const GeomaeSstLayerWeights* L_enc = e->layers.data();
    const GeomaeSstLayerWeights* L_cen = L_enc + ne;
    const GeomaeSstLayerWeights* L_den = L_cen + nd;
    const GeomaeSstLayerGrads* G_enc = e->grads.data();
    const GeomaeSstLayerGrads* G_cen = G_enc + ne;
    const GeomaeSstLayerGrads* G_den = G_cen + nd;
    GeomaeSstStackLayout lay_enc[2], lay_dec[2];
    stack_layouts(&lay[0], lay_enc);
    stack_layouts(&lay[2], lay_dec);
    const int max_tokens = c.window.window_shape[0] * c.window.window_shape[1];

    // ---------------- side streams start behind the previous step's optimizer
    e->enqueue_started = true;
    if (e->have_step_end) {
        GEOMAE_HIP(hipStreamWaitEvent(geo, e->ev[kStepEnd], 0));      // (dec_b waited for it when it packed ahead)
    } else {
        GEOMAE_HIP(hipEventRecord(e->ev[kFirstMain], main));
        GEOMAE_HIP(hipStreamWaitEvent(geo, e->ev[kFirstMain], 0));
        GEOMAE_HIP(hipStreamWaitEvent(aux, e->ev[kFirstMain], 0));
    }
    // geometry stream: (mask: drawn with the batch's stage 1) -> token coordinates -> four window layouts
    const bool was_packed = e->packed_fresh;
    if (!was_packed) ENG_CALL(pack_weights(e, geo));
    e->packed_fresh = false;
    const int64_t win_tables = geomae_window_build_batch_table_bytes(ns, 4, c.batch_size, &c.window);
    GEOMAE_REQUIRE(win_tables >= 0 && win_tables <= win_wsb, "pretrain_step: bad window table size");
    // The zero arena of everything behind the VFE forward (gradient rows of the decoders' outputs, BatchNorm backward sums, the
    // losses) is filled HERE, on the geometry stream, in front of the layouts: the main stream's wait for the layouts covers
    // it, and the decoder-B stream forks from the main stream behind that wait.  (Until round 5 the decoder-B stream filled it
    // behind the VFE forward: one more event for the main stream to wait on in front of the heads -- a satisfied wait still
    // costs the queue a barrier packet, ~6-8 us with the lost overlap of the neighbouring kernels -- and one for dec_b.)
    // GeomaeTuning.zero_late_aux: the earlier placement (A/B).
    const bool zero_on_aux = tuning().zero_late_aux != 0;
    if (!zero_on_aux) {
        ENG_CALL(zero_arena(zl0, zl_bytes, geo));
        ENG_CALL(zero_arena(losses, 32, geo));
    }
    ENG_CALL(geomae_gather_token_coors_zero(b.ids_keep, nk, b.ids_mask, nm, b.voxel_coors, coors_all, nullptr, win_ws,
                                            win_tables, geo));
    {
        GeomaeWindowBuildJob jobs[4];
        for (int k = 0; k < 4; ++k) {
            jobs[k].coors = coors_all; jobs[k].num_tokens = lay[k].n; jobs[k].shift_index = k & 1;
            jobs[k].win_start = lay[k].win_start; jobs[k].win_tokens = lay[k].win_tokens; jobs[k].tok_win = lay[k].tok_win;
            jobs[k].tok_pos = lay[k].tok_pos; jobs[k].num_windows = lay[k].num_windows; jobs[k].bun_start = lay[k].bun_start;
            jobs[k].num_bundles = lay[k].num_bundles; jobs[k].bun_tok = lay[k].bun_tok; jobs[k].pos_info = lay[k].pos_info;
            jobs[k].fbun_tok = lay[k].fbun_tok; jobs[k].num_fbundles = lay[k].num_fbundles;
            jobs[k].fitems = lay[k].fitems; jobs[k].num_fitems = lay[k].num_fitems;
        }
        set_window_tables_prezeroed(true);         // (cleared by the gather above)
        const int rc_win = geomae_window_build_batch(jobs, 4, c.batch_size, &c.window, win_ws, win_wsb, geo);
        set_window_tables_prezeroed(false);
        ENG_CALL(rc_win);
    }
    if (was_packed) GEOMAE_HIP(hipStreamWaitEvent(geo, e->ev[kPacked], 0));
    GEOMAE_HIP(hipEventRecord(e->ev[kLayouts], geo));

    // ---------------- dec_b stream, from the start of the step: this batch's geometric targets, then the NEXT batch's stage 1
    // (in this order: the other one measured 7 us slower).
    // Neither depends on anything the step computes; whatever runs beside the latency-bound encoder forward is paid
    // there almost 1:1, beside the VFE forward (8-wave workgroups, one per CU) much less: with both of them behind the
    // VFE forward (until round 2) the encoder forward took 0.356 ms, now 0.317 (VFE forward 0.101 -> 0.107).
    ENG_CALL(geomae_geometry_targets(b.points, c.num_features, b.order, b.seg_start, b.num_pillars, V, b.voxel_coors,
                                     b.coors_med, b.coors_low, b.cell_table, c.batch_size, b.token_row, b.counts,
                                     &c.targets, t_clow, t_mlow, t_cmed, t_mmed, t_ctop, t_normal, t_curv, t_top_raw,
                                     t_med_raw, t_med_raw_mask, t_cov, t_occ, (int32_t)M, aux));
    if (next_frames) {
        // the next batch's frames were produced on the CALLER's stream (H2D copies, augmentation kernels): the decoder-B
        // stream reads them, so it is ordered behind what `main` holds at this point (the event sits behind the previous
        // step's optimizer, which dec_b already waited for: the wait is free unless the caller enqueued a loader)
        ENG_CALL(order_after(e, kNextReady, main, aux));
        ENG_CALL(run_stage1(e, 1 - e->pending, next_frames, next_sizes, e->mask_draws + 1, aux));
    }
    // ---------------- main: VFE forward
    e->phase_last = -1;
    mark(e, pStart, main);
    ENG_CALL(zero_arena(zf0, zf_bytes, main));
    GeomaeVfeArgs va;
    va.feat_sorted = b.feat; va.pid_sorted = b.pid; va.seg_start = b.seg_start;
    va.num_points = N; va.max_pillars = V;
    va.w0 = m.vfe_w0; va.w1 = m.vfe_w1;
    va.scale0 = bn_scale0; va.shift0 = bn_shift0; va.scale1 = bn_scale1; va.shift1 = bn_shift1;
    va.moments = b.moments; va.dw0_acc = dw0_acc; va.pillar_ties = vfe_ties;
    va.layer1_bf16 = c.vfe_bf16;
    if (b.moments_exchanged) {
        // (mean, mean of squares) of all ranks straight from the averaged moments: no collective in the VFE forward
        GEOMAE_HIP(hipStreamWaitEvent(main, e->ev[e->pending == 0 ? kMoments0 : kMoments1], 0));
        GeomaeVfeArgs vs = va;
        vs.moments = m.bn_sync_feat_moments + (int64_t)kFeatMoments * e->pending;
        ENG_CALL(geomae_vfe_stats0(&vs, sums0, main));
        ENG_CALL(geomae_bn_finalize(sums0, 1.0, nullptr, 64, m.bn_gamma[0], m.bn_beta[0], c.bn_eps, c.bn_momentum, 0,
                                    m.bn_running_mean[0], m.bn_running_var[0], bn_scale0, bn_shift0, bn_invstd0, bn_mom0,
                                    nullptr, main));
    }
    // single process (nothing to exchange between the statistics and their use): both BatchNorm finalisations ride in the
    // heads of the sweeps that use them (GeomaeBnFold) -- three single-workgroup launches fewer on the critical path
    const bool fold_bn = !b.moments_exchanged && (!exchanges(c) || !c.sync_bn) && b.moments != nullptr;
    auto fold_of = [&](int layer, float* scale, float* shift, float* invstd, float* mom) {
        GeomaeBnFold f;
        f.count = (double)N; f.gamma = m.bn_gamma[layer]; f.beta = m.bn_beta[layer]; f.eps = c.bn_eps; f.momentum = c.bn_momentum;
        f.running_mean = m.bn_running_mean[layer]; f.running_var = m.bn_running_var[layer];
        f.scale = scale; f.shift = shift; f.invstd = invstd; f.moments = mom; f.num_batches_tracked = m.bn_num_batches[layer];
        return f;
    };
    if (fold_bn) {
        const GeomaeBnFold f0 = fold_of(0, bn_scale0, bn_shift0, bn_invstd0, bn_mom0);
        const GeomaeBnFold f1 = fold_of(1, bn_scale1, bn_shift1, bn_invstd1, bn_mom1);
        ENG_CALL(geomae_vfe_layer0_bn(&va, &f0, m0, sums1, main));
        ENG_CALL(geomae_vfe_layer1_bn(&va, &f1, sums1, m0, vf, main));
    } else {
        if (!b.moments_exchanged) {
            ENG_CALL(geomae_vfe_stats0(&va, sums0, main));
            ENG_CALL(bn_forward(e, 0, sums0, (double)N, bn_scale0, bn_shift0, bn_invstd0, bn_mom0, main));
        }
        ENG_CALL(geomae_vfe_layer0(&va, m0, sums1, main));
        ENG_CALL(bn_forward(e, 1, sums1, (double)N, bn_scale1, bn_shift1, bn_invstd1, bn_mom1, main));
        ENG_CALL(geomae_vfe_layer1(&va, m0, vf, main));
    }
    mark(e, pVfeFwd, main);
    if (zero_on_aux) {
        ENG_CALL(order_after(e, kVfeDone, main, aux));
        ENG_CALL(zero_arena(zl0, zl_bytes, aux));
        ENG_CALL(zero_arena(losses, 32, aux));
        GEOMAE_HIP(hipEventRecord(e->ev[kZeroLate], aux));
    }
    const int nxt = 1 - e->pending;

    // ---------------- main: encoder, decoders
    GEOMAE_HIP(hipStreamWaitEvent(main, e->ev[kLayouts], 0));
    mark(e, pLayouts, main);
    {
        // which of the encoder's two layouts hold a window with more than 64 kept pillars (= a bundle of more than four tiles:
        // the one-launch layer's second kernel): known since the batch's stage 1, a step ago
        int big = 3;
        const bool skip_big = tuning().fused_skip_big != 0;
        bool counted = false;                                        // (host_maxkeep is valid only behind readback2)
        // (a bundle packs whole windows up to its cap: only with a cap of at most 64 positions does "no window above 64" mean "no
        //  bundle above four tiles" -- the packing of token sets above 12288 and GeomaeTuning.bundle_cap may use larger caps)
        if (skip_big && !b.mask_injected && b.host_maxkeep && geomae_window_bundle_cap(nk, max_tokens) <= 64) {
            GEOMAE_HIP(hipEventSynchronize(b.readback2));            // (stage 1 of this batch ended during the previous step)
            counted = true;
            big = (b.host_maxkeep[0] < 0 || b.host_maxkeep[0] > 64 ? 1 : 0) | (b.host_maxkeep[1] < 0 || b.host_maxkeep[1] > 64 ? 2 : 0);
        }
        set_fused_big_layouts(big);
        e->last_big_layouts = big;
        e->last_maxkeep[0] = counted ? b.host_maxkeep[0] : -1;      // (-1: not waited for -- injected mask, the skip switched off, a larger cap)
        e->last_maxkeep[1] = counted ? b.host_maxkeep[1] : -1;
    }
    ENG_CALL(geomae_sst_stack_forward(vf, nk, L_enc, ne, lay_enc, m.pos_table, nh, max_tokens, s_enc, sb_enc, z_enc, nk,
                                      nullptr, b.ids_keep, e->profiler, main));
    e->last_forms[0] = last_stack_form(false);
    mark(e, pEncFwd, main);
    ENG_CALL(order_after(e, kForkDec, main, aux));
    set_first_live_row((int)nk);                 // only the masked pillars' rows reach the heads
    ENG_CALL(geomae_sst_stack_forward(z_enc, n, L_den, nd, lay_dec, m.pos_table, nh, max_tokens, s_den, sb_dec, den, nk,
                                      m.mask_token, nullptr, e->profiler, aux));
    e->last_forms[1] = last_stack_form(false);
    set_first_live_row((int)nk);                 // only the masked pillars' rows reach the heads
    ENG_CALL(geomae_sst_stack_forward(z_enc, n, L_cen, nd, lay_dec, m.pos_table, nh, max_tokens, s_cen, sb_dec, cen, nk,
                                      m.mask_token, nullptr, e->profiler, main));
    e->last_forms[2] = last_stack_form(false);
    mark(e, pDecFwd, main);
    // ---------------- heads + losses BY DECODER: each decoder's stream runs the heads that read its stack and goes straight
    // on into that stack's backward -- neither waits for the other stack's forward (one launch for all heads needed the
    // decoder-B stream to join the main stream in front of it and to fork again behind it: two cross-queue hand-overs of
    // ~13 us on the critical path).  The density decoder feeds ONE head (nor_top -> loss_curv_around), the centroid
    // decoder the other five.  GeomaeTuning.heads_joint: the joint launch.
    const bool heads_joint = tuning().heads_joint != 0;
    if (heads_joint) {
        ENG_CALL(order_after(e, kJoinDecFwd, aux, main));
        ENG_CALL(geomae_heads_loss_split_accumulate(cen, den, nk, nm, m.head_w_packed, m.head_bias, t_clow, t_mlow, t_cmed,
                                                    t_mmed, t_ctop, t_normal, t_occ, c.loss_weights, losses, d_cen, d_cen2,
                                                    d_den, h_dl, h_cm, h_dm, main));
        ENG_CALL(order_after(e, kHeads, main, geo));
        GEOMAE_HIP(hipStreamWaitEvent(aux, e->ev[kHeads], 0));
    } else {
        // (d_cen / d_cen2 / d_den / losses were zeroed on the geometry stream in front of the layouts: both streams are behind them)
        if (zero_on_aux) GEOMAE_HIP(hipStreamWaitEvent(main, e->ev[kZeroLate], 0));
        ENG_CALL(geomae_heads_loss_centroid_accumulate(cen, nk, nm, m.head_w_packed, m.head_bias, t_clow, t_mlow, t_cmed, t_mmed,
                                                       t_ctop, t_occ, c.loss_weights, losses, d_cen, d_cen2, h_dl, h_cm, main));
        ENG_CALL(geomae_heads_loss_density_accumulate(den, nk, nm, m.head_w_packed, m.head_bias, t_normal, c.loss_weights,
                                                      losses, d_den, h_dl, h_dm, aux));
        ENG_CALL(order_after(e, kHeads, main, geo));
        ENG_CALL(order_after(e, kJoinDecFwd, aux, geo));            // (the geometry stream needs both halves of dl)
    }
    // ---------------- backward.  Contractions that only the optimizer reads go to the geometry stream.
    ENG_CALL(geomae_heads_weight_grad(nm, h_dl, h_cm, h_dm, &m.head_grads, geo));
    mark(e, pHeads, main);
    set_first_live_row((int)nk);                 // only the masked pillars' rows reach the heads
    // The decoders' contractions go to the geometry stream every two layers while the frame is small enough for the 80-workgroup
    // contraction budget (n <= 32768: 1.680 vs 1.689 ms at config 2, five alternated runs), and behind the whole stack above that
    // (Waymo geometry 4.275 vs 4.247 ms with the flush, config 3 flat).  GeomaeTuning.dec_dw_every = k overrides (0 = behind the stack).
    const int dec_every_env = tuning().dec_dw_every;
    const int dec_every = dec_every_env >= 0 ? dec_every_env : (n <= 32768 ? 2 : 0);
    const int dec_mid_budget = tuning().dec_mid_budget;   // (A/B)
    struct MidFlushScopeD {
        explicit MidFlushScopeD(const DwMidFlush& f) { set_dw_mid_flush(f); }
        ~MidFlushScopeD() { set_dw_mid_flush(DwMidFlush()); }
    };
    {
        DeferAllScope defer(defer_dec_dw);
        DwMidFlush mfd;
        if (defer_dec_dw && dec_every > 0) { mfd.side = geo; mfd.ev = e->ev[kDecMidB]; mfd.every = dec_every; mfd.budget = dec_mid_budget; }
        MidFlushScopeD mid(mfd);
        ENG_CALL(geomae_sst_stack_backward(d_den, nullptr, n, L_den, G_den, nd, lay_dec, m.pos_table, nh, max_tokens, s_den,
                                           w_den, wb_dec, dxb, nullptr, 0, m.mask_token_grad, nk, 1, e->profiler, aux));
        e->last_forms[4] = last_stack_form(true);
    }
    GEOMAE_HIP(hipEventRecord(e->ev[kAuxBwd], aux));
    GEOMAE_HIP(hipStreamWaitEvent(geo, e->ev[kAuxBwd], 0));
    ENG_CALL(geomae_flush_weight_grad(geo));
    set_first_live_row((int)nk);                 // only the masked pillars' rows reach the heads
    {
        DeferAllScope defer(defer_dec_dw);
        DwMidFlush mfd;
        if (defer_dec_dw && dec_every > 0) { mfd.side = geo; mfd.ev = e->ev[kDecMidA]; mfd.every = dec_every; mfd.budget = dec_mid_budget; }
        MidFlushScopeD mid(mfd);
        ENG_CALL(geomae_sst_stack_backward(d_cen, d_cen2, n, L_cen, G_cen, nd, lay_dec, m.pos_table, nh, max_tokens, s_cen,
                                           w_cen, wb_dec, dxa, nullptr, 0, m.mask_token_grad, nk, 1, e->profiler, main));
        e->last_forms[5] = last_stack_form(true);
    }
    ENG_CALL(order_after(e, kMainDecBwd, main, geo));
    ENG_CALL(geomae_flush_weight_grad(geo));
    GEOMAE_HIP(hipStreamWaitEvent(main, e->ev[kAuxBwd], 0));
    mark(e, pDecBwd, main);
    if (e->hook && exchanges(c)) e->hook(e->hook_user, GEOMAE_HOOK_GRADS_EARLY, geo);
    {
        // (round 5) the ENCODER's contractions leave its backward launches too: queued per layer and flushed to the geometry
        // stream every `every` layers as one merged launch of the layer-form contraction (csrc/dw_device.h) -- the ffn-backward
        // launches carry no riders (GeomaeTuning.enc_dw_defer = 0: riding as in rounds 2-4; four layers per flush: 2 / 3 / 4 / 6
        // measured in round 5, docs/LAB_NOTES.md)
        const bool defer_enc_dw = tuning().enc_dw_defer != 0;
        const int enc_every = 4;
        DeferAllScope defer(defer_enc_dw);
        struct MidFlushScope {
            explicit MidFlushScope(const DwMidFlush& f) { set_dw_mid_flush(f); }
            ~MidFlushScope() { set_dw_mid_flush(DwMidFlush()); }
        };
        DwMidFlush mf;
        if (defer_enc_dw && enc_every > 0) { mf.side = geo; mf.ev = e->ev[kEncMid]; mf.every = enc_every; }
        MidFlushScope mid(mf);
        set_fused_big_layouts(e->last_big_layouts);      // (as for the forward: no bundle of more than four tiles -> one launch per layer)
        ENG_CALL(geomae_sst_stack_backward(dxa, dxb, nk, L_enc, G_enc, ne, lay_enc, m.pos_table, nh, max_tokens, s_enc, w_enc,
                                           wb_enc, d_vf, b.ids_keep, V, nullptr, 0, 1, e->profiler, main));
        e->last_forms[3] = last_stack_form(true);
    }
    ENG_CALL(order_after(e, kEncBwd, main, geo));
    ENG_CALL(geomae_flush_weight_grad(geo));
    if (e->hook && exchanges(c)) e->hook(e->hook_user, GEOMAE_HOOK_GRADS_ENCODER, geo);
    mark(e, pEncBwd, main);

    // ---------------- VFE backward
    GeomaeBnState bn;
    bn.scale0 = bn_scale0; bn.shift0 = bn_shift0; bn.mean0 = bn_mom0; bn.invstd0 = bn_invstd0;
    bn.scale1 = bn_scale1; bn.shift1 = bn_shift1; bn.mean1 = bn_mom1; bn.invstd1 = bn_invstd1;
    const bool fold = !exchanges(c) || !c.sync_bn;
    double* use_bs1 = fold ? bs1 : m.bn_sync_bsums1;
    double* use_bs0 = fold ? bs0 : m.bn_sync_bsums0;
    if (!fold) {
        GEOMAE_REQUIRE(use_bs1 && use_bs0, "pretrain: world_size > 1 needs bn_sync_bsums buffers");
        ENG_CALL(zero_arena(use_bs1, 256 * 8, main));
        ENG_CALL(zero_arena(use_bs0, 128 * 8, main));
    }
    ENG_CALL(geomae_vfe_backward_stats(&va, &bn, m0, vf, d_vf, use_bs1, main));
    mark(e, pVfeStats, main);
    float n_eff = (float)N;
    if (!fold) {
        ENG_CALL(geomae_bn_param_grad_add(use_bs1, 128, m.bn_dbeta[1], m.bn_dgamma[1], main));
        e->hook(e->hook_user, GEOMAE_HOOK_BN_BWD1, main);
        n_eff = (float)((double)c.world_size * (double)N);
    }
    // The step's tail.  Behind the layer-1 sweep two branches remain: the layer-1 weight-gradient contraction (needs dy1 / g:
    // a SPLIT job + its reduction, ~17 + 5 us) and the routing sweep + layer-0 finalize (~22 + 5 us).  Three placements
    // measured in round 5 (same box, alternating, ms per step): contraction on the geometry stream beside the routing sweep 1.690
    // (kept); everything in line on the main stream 1.696; contraction on the main stream and the routing sweep + finalize on
    // the decoder-B stream, idle at that point, 1.695.  An event hop between two streams costs 8-15 us on this stack (the
    // kernel trace shows the waiting queue starting that late): a branch moved to another stream pays two of them, about what
    // running it in line costs.  The switch of that experiment is gone; the first placement is what runs.
    set_mid_launch_event(e->ev[kVfeL1]);
    const int rc_l1 = geomae_vfe_backward_layer1(&va, &bn, m0, vf, d_vf, use_bs1, n_eff, dy1_b, g_b, nullptr, dh0, dm0, use_bs0,
                                                 fold ? m.bn_dbeta[1] : nullptr, fold ? m.bn_dgamma[1] : nullptr, main);
    (void)take_mid_launch_event();                       // (an early error return leaves them set)
    (void)take_mid_launch_side();
    ENG_CALL(rc_l1);
    mark(e, pVfeL1, main);
    // one [128,128] output contracted over all N points: through the split-K workspace + a reduction launch
    hipStream_t dw1_stream = geo;
    GEOMAE_HIP(hipStreamWaitEvent(geo, e->ev[kVfeL1], 0));
    set_dw_partial(dw1_partial);
    int rc_dw1 = geomae_vfe_weight_grad1(dy1_b, g_b, N, m.vfe_dw1, dw1_stream);
    set_dw_partial(nullptr);
    ENG_CALL(rc_dw1);
    ENG_CALL(geomae_flush_weight_grad(dw1_stream));
    if (!fold) {
        ENG_CALL(geomae_bn_param_grad_add(use_bs0, 64, m.bn_dbeta[0], m.bn_dgamma[0], main));
        e->hook(e->hook_user, GEOMAE_HOOK_BN_BWD0, main);
    }
    hipStream_t l0_stream = main;
    ENG_CALL(geomae_vfe_backward_layer0(&va, &bn, dh0, use_bs0, n_eff, N, dy1_b, g_b, m.vfe_dw0, nullptr,
                                        fold ? m.bn_dbeta[0] : nullptr, fold ? m.bn_dgamma[0] : nullptr, l0_stream));
    mark(e, pVfeL0, main);
    ENG_CALL(order_after(e, kGeoDone, geo, main));
    mark(e, pVfeBwd, main);

    // the next batch's feature moments -> all ranks (behind every other collective of this step)
    if (next_frames) ENG_CALL(exchange_moments(e, nxt, aux));
    // the consumed batch's slot is free again; the next batch (if any) is pending
    b.valid = false;
    e->pending = next_frames ? nxt : -1;
    if (do_opt) return run_optimizer(e, lr, grad_scale, main);
    // without the optimizer: the side streams of the next step still need an ordering point
    GEOMAE_HIP(hipEventRecord(e->ev[kStepEnd], main));
    GEOMAE_HIP(hipStreamWaitEvent(aux, e->ev[kStepEnd], 0));
    e->have_step_end = true;
    return GEOMAE_OK;
}

}  // namespace
}  // namespace geomae

using namespace geomae;

extern "C" int geomae_bn_param_grad_add(const double* bsums, int32_t channels, float* d_beta, float* d_gamma,
                                        hipStream_t stream) {
    GEOMAE_REQUIRE(bsums && d_beta && d_gamma && channels >= 1, "bn_param_grad_add: bad argument");
    hipLaunchKernelGGL(bn_param_grad_add_kernel, dim3(cdiv(channels, 256)), dim3(256), 0, stream, bsums, channels, d_beta,
                       d_gamma);
    return check_launch("bn_param_grad_add_kernel");
}

static int check_cfg(const GeomaePretrainConfig* c) {
    GEOMAE_REQUIRE(c, "pretrain: null config");
    GEOMAE_REQUIRE(c->batch_size >= 1 && c->num_features >= 4 && c->num_heads >= 1 && c->encoder_layers >= 1 &&
                   c->decoder_layers >= 1, "pretrain: bad sizes in the config");
    GEOMAE_REQUIRE(c->targets.grid_size[0] == 1 && c->targets.grid_size[1] >= 1 && c->targets.grid_size[2] >= 1,
                   "pretrain: the top grid must be (1, ny, nx)");
    GEOMAE_REQUIRE(c->keep_fraction > 0.0 && c->keep_fraction < 1.0, "pretrain: keep_fraction must be in (0, 1)");
    return GEOMAE_OK;
}

extern "C" int64_t geomae_pretrain_workspace_bytes(const GeomaePretrainConfig* cfg, int64_t max_points,
                                                   int32_t max_pillars) {
    if (check_cfg(cfg) != GEOMAE_OK || max_points < 1 || max_pillars < 1) return -1;
    return kPersistentBytes + 2 * stage_region_bytes(*cfg, max_points) + step_region_bytes(*cfg, max_points, max_pillars);
}

extern "C" void* geomae_pretrain_create(const GeomaePretrainConfig* cfg, const GeomaePretrainModel* model, void* workspace,
                                        int64_t workspace_bytes, int64_t max_points, int32_t max_pillars,
                                        const hipStream_t* side_streams) {
    if (check_cfg(cfg) != GEOMAE_OK) return nullptr;
    if (!model || !workspace || !side_streams || !model->layers || !model->layer_grads || !model->params || !model->grads) {
        set_error("pretrain_create: null argument");
        return nullptr;
    }
    Engine* e = new (std::nothrow) Engine();
    if (!e) { set_error("pretrain_create: out of host memory"); return nullptr; }
    e->cfg = *cfg;
    e->m = *model;
    const int nl = cfg->encoder_layers + 2 * cfg->decoder_layers;
    e->layers.assign(model->layers, model->layers + nl);
    e->grads.assign(model->layer_grads, model->layer_grads + nl);
    e->m.layers = e->layers.data();
    e->m.layer_grads = e->grads.data();
    e->ws = (char*)workspace;
    e->ws_bytes = workspace_bytes;
    e->gz = cfg->targets.grid_size[0]; e->gy = cfg->targets.grid_size[1]; e->gx = cfg->targets.grid_size[2];
    e->cells = cfg->batch_size * e->gz * e->gy * e->gx;
    e->s_low = cfg->targets.ratio_low[0] * cfg->targets.ratio_low[1] * cfg->targets.ratio_low[2];
    e->s_med = cfg->targets.ratio_med[0] * cfg->targets.ratio_med[1] * cfg->targets.ratio_med[2];
    e->geo = side_streams[0];
    e->aux = side_streams[1];
    e->persistent_bytes = kPersistentBytes;
    e->stage_bytes = stage_region_bytes(*cfg, max_points);
    if (max_points < 1 || max_pillars < 1 ||
        workspace_bytes < geomae_pretrain_workspace_bytes(cfg, max_points, max_pillars)) {
        set_error("pretrain_create: workspace of %lld bytes < geomae_pretrain_workspace_bytes(%lld points, %d pillars)",
                  (long long)workspace_bytes, (long long)max_points, max_pillars);
        delete e;
        return nullptr;
    }
    bool ok = true;
    for (int i = 0; i < kNumEv; ++i) ok = ok && hipEventCreateWithFlags(&e->ev[i], hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < kNumPhase; ++i) ok = ok && hipEventCreate(&e->phase_ev[i]) == hipSuccess;
    for (int k = 0; k < 2; ++k) {
        ok = ok && hipEventCreateWithFlags(&e->batch[k].readback, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&e->batch[k].readback2, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipHostMalloc((void**)&e->batch[k].host_maxkeep, 2 * 4, hipHostMallocDefault) == hipSuccess;
        ok = ok && hipHostMalloc((void**)&e->batch[k].host_offs, (size_t)(cfg->batch_size + 1) * 4, hipHostMallocDefault) == hipSuccess;
        ok = ok && hipHostMalloc((void**)&e->batch[k].host_counts, (size_t)(cfg->batch_size + 1) * 4, hipHostMallocDefault) == hipSuccess;
    }
    Arena a;
    a.reset(e->ws, kPersistentBytes);
    e->sumsq_ring = a.take<double>(2);
    e->gnorm = a.take<float>(1);
    e->losses_ring = a.take<float>(32);
    ok = ok && hipMemset(e->ws, 0, kPersistentBytes) == hipSuccess;
    if (!ok) {
        set_error("pretrain_create: HIP resource creation failed");
        geomae_pretrain_destroy(e);
        return nullptr;
    }
    return e;
}

extern "C" void geomae_pretrain_destroy(void* engine) {
    Engine* e = (Engine*)engine;
    if (!e) return;
    for (int i = 0; i < kNumEv; ++i) if (e->ev[i]) (void)hipEventDestroy(e->ev[i]);
    for (int i = 0; i < kNumPhase; ++i) if (e->phase_ev[i]) (void)hipEventDestroy(e->phase_ev[i]);
    for (int k = 0; k < 2; ++k) {
        if (e->batch[k].readback) (void)hipEventDestroy(e->batch[k].readback);
        if (e->batch[k].readback2) (void)hipEventDestroy(e->batch[k].readback2);
        if (e->batch[k].host_maxkeep) (void)hipHostFree(e->batch[k].host_maxkeep);
        if (e->batch[k].host_offs) (void)hipHostFree(e->batch[k].host_offs);
        if (e->batch[k].host_counts) (void)hipHostFree(e->batch[k].host_counts);
    }
    delete e;
}

extern "C" int geomae_pretrain_set_hook(void* engine, GeomaePretrainHook hook, void* user) {
    GEOMAE_REQUIRE(engine, "pretrain: null engine");
    ((Engine*)engine)->hook = hook;
    ((Engine*)engine)->hook_user = user;
    return GEOMAE_OK;
}
extern "C" int geomae_pretrain_set_profiler(void* engine, void* profiler) {
    GEOMAE_REQUIRE(engine, "pretrain: null engine");
    ((Engine*)engine)->profiler = profiler;
    return GEOMAE_OK;
}
extern "C" int geomae_pretrain_set_phase_timing(void* engine, int32_t enabled) {
    GEOMAE_REQUIRE(engine, "pretrain: null engine");
    ((Engine*)engine)->phase_timing = enabled != 0;
    if (!enabled) ((Engine*)engine)->phase_last = -1;
    return GEOMAE_OK;
}
extern "C" int32_t geomae_pretrain_phase_times(void* engine, float* ms_out, int32_t capacity) {
    Engine* e = (Engine*)engine;
    if (!e || !ms_out || e->phase_last < 1) return 0;
    if (hipEventSynchronize(e->phase_ev[e->phase_last]) != hipSuccess) return 0;
    int n = 0;
    for (int i = 0; i < e->phase_last && n < capacity; ++i, ++n)
        if (hipEventElapsedTime(&ms_out[n], e->phase_ev[i], e->phase_ev[i + 1]) != hipSuccess) return n;
    return n;
}
extern "C" int geomae_pretrain_invalidate_packed(void* engine) {
    GEOMAE_REQUIRE(engine, "pretrain: null engine");
    ((Engine*)engine)->packed_fresh = false;
    return GEOMAE_OK;
}

extern "C" int32_t geomae_pretrain_pending_slot(void* engine) {
    Engine* e = (Engine*)engine;
    return e ? e->pending : -1;
}

extern "C" int geomae_pretrain_submit(void* engine, const float* const* frame_points, const int64_t* frame_sizes,
                                      hipStream_t stream) {
    return geomae_pretrain_submit_ex(engine, frame_points, frame_sizes, 0, stream);
}

extern "C" int geomae_pretrain_submit_ex(void* engine, const float* const* frame_points, const int64_t* frame_sizes,
                                         int32_t flags, hipStream_t stream) {
    Engine* e = (Engine*)engine;
    GEOMAE_REQUIRE(e, "pretrain: null engine");
    // a batch submitted out of band: its slot must not be the one a pending batch occupies
    const int which = e->pending >= 0 ? 1 - e->pending : 0;
    if (e->pending >= 0) e->batch[e->pending].valid = false;      // replaced
    // order behind everything of the previous step (its kernels may still read the slot's previous batch)
    if (e->have_step_end) GEOMAE_HIP(hipStreamWaitEvent(stream, e->ev[kStepEnd], 0));
    int rc = run_stage1(e, which, frame_points, frame_sizes, e->mask_draws + 1, stream);
    if (rc == GEOMAE_OK) {
        if ((flags & GEOMAE_SUBMIT_MOMENTS_EXCHANGED) && exchanges(e->cfg) && e->cfg.sync_bn && e->m.bn_sync_feat_moments) {
            // a RE-submission (the caller re-created the engine for a larger workspace on THIS rank only): the batch's
            // rank-averaged feature moments were exchanged when it was first submitted and the caller put them back
            // into this slot of bn_sync_feat_moments -- a second all-reduce here would have no partner on the ranks that
            // did not grow (every later SyncBN collective would then pair with the wrong one)
            GEOMAE_HIP(hipEventRecord(e->ev[which == 0 ? kMoments0 : kMoments1], stream));
            e->batch[which].moments_exchanged = true;
        } else {
            rc = exchange_moments(e, which, stream);
        }
    }
    if (rc != GEOMAE_OK) { e->pending = -1; return rc; }
    e->pending = which;
    // the step's side streams read the batch: order them behind this stream's stage 1
    GEOMAE_HIP(hipEventRecord(e->ev[kFirstMain], stream));
    GEOMAE_HIP(hipStreamWaitEvent(e->geo, e->ev[kFirstMain], 0));
    GEOMAE_HIP(hipStreamWaitEvent(e->aux, e->ev[kFirstMain], 0));
    return GEOMAE_OK;
}

extern "C" int geomae_pretrain_set_mask(void* engine, const int32_t* ids_keep, int32_t num_keep, const int32_t* ids_mask,
                                        int32_t num_mask, hipStream_t stream) {
    Engine* e = (Engine*)engine;
    GEOMAE_REQUIRE(e, "pretrain: null engine");
    GEOMAE_REQUIRE(e->pending >= 0 && e->batch[e->pending].valid, "pretrain_set_mask: no batch submitted");
    GEOMAE_REQUIRE(ids_keep && ids_mask && num_keep >= 1 && num_mask >= 1, "pretrain_set_mask: bad argument");
    Batch& b = e->batch[e->pending];
    ENG_CALL(read_counts(e, b));
    GEOMAE_REQUIRE(num_keep + num_mask == b.V, "pretrain_set_mask: %d kept + %d masked ids for a batch of %d pillars", num_keep,
                   num_mask, b.V);
    GEOMAE_HIP(hipMemcpyAsync(b.ids_keep, ids_keep, (size_t)num_keep * 4, hipMemcpyDeviceToDevice, stream));
    GEOMAE_HIP(hipMemcpyAsync(b.ids_mask, ids_mask, (size_t)num_mask * 4, hipMemcpyDeviceToDevice, stream));
    hipLaunchKernelGGL(token_rows_from_ids_kernel, dim3(cdiv(b.V, 256)), dim3(256), 0, stream, b.ids_keep, num_keep, b.ids_mask,
                       num_mask, b.token_row, b.counts);
    ENG_CALL(check_launch("token_rows_from_ids_kernel"));
    b.n_keep = num_keep;
    b.n_mask = num_mask;
    // the fullest windows of THIS mask (the drawn mask's numbers no longer apply): counted again, read back before the step
    ENG_CALL(window_max_keep(b.ids_keep, b.counts, b.voxel_coors, e->cfg.batch_size, &e->cfg.window, b.host_maxkeep, stream));
    GEOMAE_HIP(hipEventRecord(b.readback2, stream));
    b.mask_injected = false;
    // the step's side streams read the mask: order them behind this stream
    GEOMAE_HIP(hipEventRecord(e->ev[kFirstMain], stream));
    GEOMAE_HIP(hipStreamWaitEvent(e->geo, e->ev[kFirstMain], 0));
    GEOMAE_HIP(hipStreamWaitEvent(e->aux, e->ev[kFirstMain], 0));
    return GEOMAE_OK;
}

extern "C" int geomae_pretrain_set_mask_draws(void* engine, uint64_t steps_begun) {
    GEOMAE_REQUIRE(engine, "pretrain: null engine");
    ((Engine*)engine)->mask_draws = steps_begun;
    return GEOMAE_OK;
}

extern "C" int geomae_pretrain_step(void* engine, const float* const* next_frame_points, const int64_t* next_frame_sizes,
                                    float lr, float grad_scale, int32_t run_opt, hipStream_t stream) {
    Engine* e = (Engine*)engine;
    GEOMAE_REQUIRE(e, "pretrain: null engine");
    GEOMAE_REQUIRE(!e->poisoned, "pretrain_step: an earlier step failed after its first launch; destroy this engine");
    const double t0 = now_s();
    e->enqueue_started = false;
    const int rc = run_step(e, next_frame_points, next_frame_sizes, lr, grad_scale, run_opt, stream);
    e->host_step_s += now_s() - t0;
    if (rc != GEOMAE_OK && e->enqueue_started) e->poisoned = true;      // (failures before the first launch are clean)
    return rc;
}

extern "C" int geomae_pretrain_optimizer(void* engine, float lr, float grad_scale, hipStream_t stream) {
    Engine* e = (Engine*)engine;
    GEOMAE_REQUIRE(e, "pretrain: null engine");
    return run_optimizer(e, lr, grad_scale, stream);
}

extern "C" int64_t geomae_pretrain_result_offset(void* engine, int32_t what) {
    Engine* e = (Engine*)engine;
    if (!e) return -1;
    switch (what) {
        case 0: return e->off_losses;
        case 1: return (char*)e->gnorm - e->ws;
        case 2: return e->off_ids_keep;
        case 3: return e->off_ids_mask;
        case 4: return e->off_voxel_coors;
        default: return -1;
    }
}

extern "C" int geomae_pretrain_host_times(void* engine, double* out) {
    Engine* e = (Engine*)engine;
    GEOMAE_REQUIRE(e && out, "pretrain: null argument");
    out[0] = e->host_step_s; out[1] = e->host_blocked_s; out[2] = (double)e->steps;
    return GEOMAE_OK;
}

extern "C" int geomae_pretrain_set_optimizer_steps(void* engine, int64_t steps_taken) {
    GEOMAE_REQUIRE(engine && steps_taken >= 0, "pretrain: bad argument");
    ((Engine*)engine)->opt_steps = steps_taken;
    return GEOMAE_OK;
}

extern "C" int geomae_pretrain_last_sizes_n(void* engine, int64_t* out, int32_t capacity) {
    GEOMAE_REQUIRE(engine && out && capacity >= 0, "pretrain: null argument");
    int64_t all[9];
    const int rc = geomae_pretrain_last_sizes(engine, all);
    if (rc) return rc;
    const int n = capacity < 9 ? capacity : 9;
    for (int i = 0; i < n; ++i) out[i] = all[i];
    return n;
}
extern "C" int geomae_pretrain_step_forms(void* engine, int32_t* out, int32_t capacity) {
    Engine* e = (Engine*)engine;
    GEOMAE_REQUIRE(e && out && capacity >= 0, "pretrain: null argument");
    const int n = capacity < 6 ? capacity : 6;
    for (int i = 0; i < n; ++i) out[i] = e->last_forms[i];
    return n;
}
extern "C" int geomae_pretrain_last_sizes(void* engine, int64_t* out) {
    Engine* e = (Engine*)engine;
    GEOMAE_REQUIRE(e && out, "pretrain: null argument");
    out[0] = e->last_N; out[1] = e->last_V; out[2] = e->last_keep; out[3] = e->last_mask; out[4] = e->opt_steps;
    out[5] = (int64_t)e->mask_draws;
    out[6] = e->last_maxkeep[0]; out[7] = e->last_maxkeep[1]; out[8] = e->last_big_layouts;
    return GEOMAE_OK;
}
