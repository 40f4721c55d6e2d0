/* libgeomae_hip -- C ABI of the MI355X-native GeoMAE-SST pre-training hot path.
 *
 * Drop-in boundary: these entry points are what the reference's
 * native bindings for this path would bind instead of its pybind module `voxel_layer`
 * (mmdet3d/ops/voxel/src/voxelization.cpp:5-11, voxelization.h:58-154), torch_scatter
 * (call sites mmdet3d/ops/sst/sst_ops.py:30,32) and spconv's index-pair generator
 * (call site mmdet3d/models/detectors/multi_sub_voxel_dynamic_voxelnet_ssl.py:192-207).
 *
 * Conventions (same as the reference's native ops): the CALLER allocates every output and
 * workspace (torch owns all memory); all pointers are DEVICE pointers unless marked host;
 * row-major contiguous; work is enqueued on `stream` and nothing blocks the host; nothing is
 * allocated and no global state is kept.  Return 0, or a negative GEOMAE_ERR_* code with a
 * thread-local message in geomae_last_error().  Coordinates are int32 (b, z, y, x); geometric
 * vectors are fp32 (z, y, x), as in the reference.
 */
#ifndef GEOMAE_HIP_H
#define GEOMAE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* geomaeStream_t; /* == hipStream_t */

#define GEOMAE_ABI_VERSION 1

const char* geomae_last_error(void);
/* Per host thread.  enabled != 0: the caller promises that the accumulator / atomics-target OUTPUT buffers it passes
 * from now on are already zero, and the entry points stop enqueueing their own memsets: geomae_segment_mean_xyz
 * (sum_workspace), geomae_vfe_stats0 (sums0), geomae_vfe_layer0 (m0, sums1), geomae_vfe_layer1 (voxel_feats),
 * geomae_vfe_backward_stats (bsums1), geomae_vfe_backward_layer1 (bsums0, dm0), geomae_grad_sumsq (sumsq).
 * A training step zeroes one arena off the critical path instead of a dozen small fills between dependent kernels. */
int geomae_set_accumulators_prezeroed(int32_t enabled);

/* ------------------------------------------------------------------ tuning surface
 * Every switch that selects a kernel form, a schedule or a launch shape lives HERE: one process-wide struct with defaults,
 * read by the library where the choice is made (no getenv under csrc/).  geomae_get_tuning fills the caller's struct with
 * the current values; geomae_set_tuning replaces them (fields outside their range are clamped; `size` must be
 * sizeof(GeomaeTuning)).  The Python binding reads the GEOMAE_* environment variables of the same names ONCE at load
 * (geomae_amd/_lib.py TUNING_ENV) and applies them through this call; tests and A/B tools call it in-process.  DESIGN.md
 * section 9 is the table of the fields with what each was measured against.  Set it between steps, not while one is enqueued. */
typedef struct GeomaeTuning {
    int32_t size;                  /* sizeof(GeomaeTuning) */
    /* which form a layer stack takes (sst_stack.hip) */
    int32_t fused_layers;          /* 1  | 0: never the one-launch layer of sst_fused.hip, 1: token sets <= fused_max_tokens, 2: always, 3: as 1 (kept for A/B scripts) */
    int32_t fused_max_tokens;      /* 12288 */
    int32_t fused_bwd;             /* 1  | one-launch backward where the forward took the one-launch form */
    int32_t ws_layers;             /* 0  | the looping one-launch layer of sst_ws.hip (windows of up to 144 positions): 0 never, 1 token sets >
                                      fused_max_tokens, 2 always.  OFF: measured 5-9 % slower than the three-launch form in the step
                                      (docs/LAB_NOTES.md "Round 6"); kept as a tested alternative */
    int32_t ws_bwd;                /* 0  | reserved: a backward of that form (not built: the forward did not win) */
    int32_t ws_bundle_cap;         /* 144 | soft cap (positions) of the bundles of the token sets the ws form takes */
    int32_t ws_max_workgroups;     /* 0  | 0 = one per CU */
    int32_t bundle_cap;            /* 0  | != 0 overrides geomae_window_bundle_cap for every token set */
    int32_t saved_f32;             /* 0  | 1: x-hat rows saved in fp32 (three-launch form only; parity checks) */
    int32_t x_from_xhat;           /* 1  | dW_v's operand x of the layers above the first formed from the saved xhat2 of the layer below */
    int32_t y_from_xhat;           /* 1  | dW1's operand y formed from the saved xhat1 */
    int32_t pair_kernels;          /* -1 | the 32-token pair form of the ffn forward: -1 by size, 0 never, 1 always */
    int32_t attn_heads;            /* 0  | heads per workgroup of the window-attention kernels: 0 by size, 1 / 2 / 4 */
    /* weight-gradient contractions (sst_layer.hip, dw_device.h) */
    int32_t dw_layer_form;         /* 1  | the layer-form contraction (0: the round-4 task form) */
    int32_t dw_chunks;             /* 0  | token chunks per job (0 = by size) */
    int32_t dw_budget_mid;         /* 0  | workgroup budget of a flush on the way (0 = default 80) */
    int32_t dw_split_reduce;       /* 1  | two-level reduction of the SPLIT job's partials */
    /* step engine schedule (engine.hip) */
    int32_t dw_defer_all;          /* 1  | the decoders' contractions leave their backward launches for the geometry stream */
    int32_t dec_dw_every;          /* -1 | flush period (layers) of the decoders' contractions: -1 by size, 0 behind the stack */
    int32_t dec_mid_budget;        /* 0  | workgroup budget of those flushes (0 = default) */
    int32_t enc_dw_defer;          /* 1  | the encoder's contractions on the geometry stream, flushed every four layers */
    int32_t zero_late_aux;         /* 0  | 1: the zero arena filled on the decoder-B stream (the round-4 placement) */
    int32_t fused_skip_big;        /* 1  | skip the one-launch layer's second kernel when no window kept more than 64 pillars */
    int32_t heads_joint;           /* 0  | 1: all six heads in one launch on the main stream */
    int32_t fwd_item_cap;          /* 0  | != 0 (32): the one-launch forward walks WORK ITEMS -- a packing of that cap whose bundles of three / four
                                      tiles are split by query tile (GeomaeWindowBuildJob.fitems).  OFF: sst_layer_fwd_kernel 18.3 -> 17.0 us
                                      per launch, the step within noise (docs/LAB_NOTES.md round 6); 0: the second packing, as the backward */
    int32_t reserved[7];
} GeomaeTuning;
int geomae_get_tuning(GeomaeTuning* out);
int geomae_set_tuning(const GeomaeTuning* in);
int32_t geomae_abi_version(void);

/* ------------------------------------------------------------------ A1 dynamic voxelization
 * replaces voxel_layer.dynamic_voxelize(points, coors, voxel_size, coors_range, NDim=3)
 * (voxelization.h:97-109 -> voxelization_cuda.cu:22-63,352-393).  Bit-exact, including this
 * fork's clamp of out-of-range points into the border cell.  coors: [N, 3] int32 (z, y, x). */
int geomae_grid_size(const float* voxel_size /*host[3] x,y,z*/, const float* coors_range /*host[6]*/,
                     int32_t* grid_xyz /*host[3]*/);
int geomae_dynamic_voxelize(const float* points, int64_t num_points, int32_t num_features,
                            const float* voxel_size /*host[3]*/, const float* coors_range /*host[6]*/,
                            int32_t* coors, geomaeStream_t stream);

/* Fused form of detector.voxelize / sub_voxelize_med / sub_voxelize_low (ssl.py:307-377): one pass
 * over the concatenated batch, three resolutions, batch index prepended: coors_*: [N, 4] int32.
 * batch_offsets: device int32 [B + 1] (row offsets of the samples in `points`). */
int geomae_voxelize_batch3(const float* points, int64_t num_points, int32_t num_features,
                           const int32_t* batch_offsets, int32_t batch_size,
                           const float* voxel_size_top, const float* voxel_size_med,
                           const float* voxel_size_low, const float* coors_range /*all host*/,
                           int32_t* coors_top, int32_t* coors_med, int32_t* coors_low,
                           geomaeStream_t stream);
/* The step's entry point: B frames -> the concatenated point rows (the reference's torch.cat, ssl.py:320-329) and the
 * three coordinate arrays of geomae_voxelize_batch3, in ONE launch.  frame_points / frame_sizes: HOST arrays (device
 * pointers, row counts; at most 32 frames).  points_out [N, num_features], batch_offsets_out [B + 1] int32 (or NULL).
 * zero_a / zero_b: up to two device ranges (16-byte aligned, multiples of 16 bytes, or NULL / 0) cleared by the same
 * launch -- the pillar table and the scan state of geomae_pillar_segment_ex (prezeroed = 1). */
int geomae_voxelize_frames3(const float* const* frame_points, const int64_t* frame_sizes, int32_t batch_size,
                            int32_t num_features, const float* voxel_size_top, const float* voxel_size_med,
                            const float* voxel_size_low, const float* coors_range, float* points_out,
                            int32_t* batch_offsets_out, int32_t* coors_top, int32_t* coors_med, int32_t* coors_low,
                            void* zero_a, int64_t zero_a_bytes, void* zero_b, int64_t zero_b_bytes, geomaeStream_t stream);

/* ------------------------------------------------------------------ A2 pillar segments
 * replaces torch.unique(coors, dim=0, return_inverse=True) of scatter_v2 (sst_ops.py:8-39) and of
 * get_centroid_per_voxel (ssl.py:749).  voxel_coors comes out in the same lexicographic
 * (b, z, y, x) order.  Outputs sized for the worst case (V <= min(N, cells)):
 *   cell_table  [B*gz*gy*gx]  pillar id of every cell or -1 (kept for neighbour lookups)
 *   voxel_coors [cap, 4]      inv [N] (point -> pillar)      order [N] (points grouped by pillar)
 *   seg_start   [cap + 1]     sample_start [B + 1] (pillar offsets per sample)    num_pillars [1] */
int64_t geomae_pillar_segment_workspace_bytes(int64_t num_points, int32_t batch_size, int32_t gz,
                                              int32_t gy, int32_t gx);
int geomae_pillar_segment(const int32_t* coors /*[N,4]*/, int64_t num_points, int32_t batch_size,
                          int32_t gz, int32_t gy, int32_t gx, int32_t* cell_table,
                          int32_t* voxel_coors, int32_t* inv, int32_t* order, int32_t* seg_start,
                          int32_t* sample_start, int32_t* num_pillars, void* workspace,
                          int64_t workspace_bytes, geomaeStream_t stream);
/* same for coors [N, ndim]: ndim 4 = (b,z,y,x), ndim 3 = (z,y,x) with batch_size 1.  Rows with a negative or
 * out-of-grid coordinate are dropped: inv[i] = -1, they do not appear in `order`, and seg_start[V] = number of
 * valid points (upstream dynamic voxelization marks out-of-range points with -1; this fork's clamps instead,
 * voxelization_cuda.cu:35-57, so the pre-training path never has any). */
int geomae_pillar_segment_nd(const int32_t* coors, int32_t ndim, int64_t num_points, int32_t batch_size,
                             int32_t gz, int32_t gy, int32_t gx, int32_t* cell_table, int32_t* voxel_coors,
                             int32_t* inv, int32_t* order, int32_t* seg_start, int32_t* sample_start,
                             int32_t* num_pillars, void* workspace, int64_t workspace_bytes, geomaeStream_t stream);
/* geomae_pillar_segment_nd plus: sample_start_mirror (or NULL) -- a second copy of sample_start written by the same
 * kernel, e.g. pinned host memory the device can address (the step's one count readback without a copy command);
 * prezeroed != 0 -- the caller hands in cell_table and the first geomae_pillar_segment_scan_state_bytes(...) bytes of
 * the workspace already ZERO (geomae_voxelize_frames3 clears them on the side), so no memset is enqueued. */
int geomae_pillar_segment_ex(const int32_t* coors, int32_t ndim, int64_t num_points, int32_t batch_size, int32_t gz,
                             int32_t gy, int32_t gx, int32_t* cell_table, int32_t* voxel_coors, int32_t* inv,
                             int32_t* order, int32_t* seg_start, int32_t* sample_start, int32_t* num_pillars,
                             void* workspace, int64_t workspace_bytes, int32_t* sample_start_mirror, int32_t prezeroed,
                             geomaeStream_t stream);
int64_t geomae_pillar_segment_scan_state_bytes(int32_t batch_size, int32_t gz, int32_t gy, int32_t gx);

/* torch_scatter.scatter(reduce='mean') of the xyz columns (voxel_encoder.py:375): mean [cap, 3].
 * 2^-32 fixed-point int64 atomics per point (order independent); sum_workspace: cap * 3 * 8 bytes. */
int geomae_segment_mean_xyz(const float* points, int32_t num_features, int64_t num_points, const int32_t* inv,
                            const int32_t* seg_start, const int32_t* num_pillars, int32_t max_pillars,
                            void* sum_workspace, float* mean, geomaeStream_t stream);
/* The same means (bit-identical) from the pillar-sorted point list of geomae_pillar_segment: no atomics, no workspace. */
int geomae_segment_mean_xyz_sorted(const float* points, int32_t num_features, const int32_t* order,
                                   const int32_t* seg_start, const int32_t* num_pillars, int32_t max_pillars,
                                   float* mean, geomaeStream_t stream);
/* torch_scatter.scatter_max (voxel_encoder.py:407): feat [N, C] in point order -> out [cap, C],
 * argmax [cap, C] (point index); backward routes grad_out to the arg-max rows: grad_feat [N, C]. */
int geomae_segment_max_forward(const float* feat, int32_t channels, const int32_t* order,
                               const int32_t* seg_start, const int32_t* num_pillars, int32_t max_pillars,
                               float* out, int32_t* argmax, geomaeStream_t stream);
int geomae_segment_max_backward(const float* grad_out, const int32_t* argmax, const int32_t* inv,
                                int64_t num_points, int32_t channels, float* grad_feat,
                                geomaeStream_t stream);

/* ------------------------------------------------------------------ A6 random masking
 * replaces get_vanilla_mask_index (ssl.py:287-304).  Per sample keeps int(L * keep_fraction)
 * pillars chosen uniformly at random (counter-based RNG on `seed`).  ids_keep / ids_mask:
 * ascending pillar ids, [cap] each; token_row [cap]: position of the pillar in the decoder's
 * token list (kept tokens first, then masked); counts [2] = (n_keep, n_mask). */
int geomae_random_mask(const int32_t* sample_start, int32_t batch_size, double keep_fraction,
                       uint64_t seed, int32_t* ids_keep, int32_t* ids_mask, int32_t* token_row,
                       int32_t* counts, geomaeStream_t stream);


/* coors_out [num_keep + num_mask, 4] = voxel_coors rows of the kept pillars followed by the masked pillars (the
 * decoder's token order; its head is the encoder's token list, bb.py:227-246 torch.cat of the two gathers), and
 * optionally ids_keep widened to int64. */
int geomae_gather_token_coors(const int32_t* ids_keep, int32_t num_keep, const int32_t* ids_mask, int32_t num_mask,
                              const int32_t* voxel_coors /*[V,4]*/, int32_t* coors_out, int64_t* ids_keep_i64 /*or NULL*/,
                              geomaeStream_t stream);
/* the same, and clears `zero_bytes` bytes at `zero` (16-byte aligned, a multiple of 16; or NULL / 0) on the side: the step
 * engine hands it the window tables of the geomae_window_build_batch that follows (geomae_window_build_batch_table_bytes
 * bytes at the start of that call's workspace), which then enqueues no memset */
int geomae_gather_token_coors_zero(const int32_t* ids_keep, int32_t num_keep, const int32_t* ids_mask, int32_t num_mask,
                                   const int32_t* voxel_coors, int32_t* coors_out, int64_t* ids_keep_i64, void* zero,
                                   int64_t zero_bytes, geomaeStream_t stream);

/* ------------------------------------------------------------------ A5,A7-A11 geometric targets */
typedef struct GeomaeTargetConfig {
    int32_t grid_size[3];       /* top grid (z, y, x), z must be 1           (config grid_size)        */
    int32_t ratio_low[3];       /* sub_voxel_ratio_low (z, y, x)                                      */
    int32_t ratio_med[3];       /* sub_voxel_ratio_med (z, y, x)                                      */
    float voxel_size_top[3];    /* (x, y, z)                                                          */
    float voxel_size_med[3];
    float voxel_size_low[3];
    float coors_range[6];       /* point_cloud_range                                                  */
} GeomaeTargetConfig;

/* replaces get_centroid_per_voxel x3, get_multi_voxel_id_to_tensor_id_for_curv, the spconv 3x3
 * neighbour table, cal_regular_voxel_nor_and_curv, normalize_centroid_sub_voxel x3 and
 * get_multi_voxel_id_to_tensor_id_ori (ssl.py:575-768).  Rows of the first seven outputs are the
 * masked pillars in ids_mask order (row = token_row[p] - counts[0]); pass token_row = mask_counts
 * = NULL to get one row per pillar.  S_low / S_med = prod(ratio).
 *   centroid_low [M,S_low,3] f32   mask_low [M,S_low] u8   centroid_med [M,S_med,3]   mask_med [M,S_med]
 *   centroid_top [M,3]   normal [M,3] f32 (canonical sign)   curv [M,3] f64
 *   top_raw [cap,3], med_raw [cap,S_med,3], med_raw_mask [cap,S_med]: un-normalised centroids of
 *   every pillar (also scratch for the neighbourhood pass);  cov_out [M,6] optional (may be NULL): the 3x3 scatter
 *   matrices (upper triangle) of cal_regular_voxel_nor_and_curv; when given, the eigen-decompositions run in a second
 *   launch, one per thread, instead of on one lane of each pillar's wave (same results, 5x shorter) */
int geomae_geometry_targets(const float* points, int32_t num_features, const int32_t* order,
                            const int32_t* seg_start, const int32_t* num_pillars, int32_t max_pillars,
                            const int32_t* voxel_coors, const int32_t* coors_med,
                            const int32_t* coors_low, const int32_t* cell_table, int32_t batch_size,
                            const int32_t* token_row, const int32_t* mask_counts,
                            const GeomaeTargetConfig* config /*host*/, float* centroid_low,
                            uint8_t* mask_low, float* centroid_med, uint8_t* mask_med,
                            float* centroid_top, float* normal, double* curv, float* top_raw,
                            float* med_raw, uint8_t* med_raw_mask, float* cov_out,
                            int32_t* occ_counts /* [2] occupied low / med cells over the output rows, may be NULL */,
                            int32_t num_rows /* rows of the M-sized outputs (needed with occ_counts) */,
                            geomaeStream_t stream);

/* ------------------------------------------------------------------ A12-A16 windows */
typedef struct GeomaeWindowConfig {
    int32_t window_shape[2];    /* (x, y) pillars per window, e.g. 12, 12                            */
    int32_t shift[2];           /* second layout's shift, e.g. 6, 6 (shift_index 1)                   */
    int32_t bev_shape[2];       /* pillars along (x, y), e.g. 400, 400                                */
} GeomaeWindowConfig;

/* The same subset as geomae_random_mask draws from the same seed, with ids_keep / ids_mask emitted WINDOW-MAJOR: grouped
 * by the unshifted SST window of the pillar (windows ascending, pillars ascending inside a window) instead of ascending
 * pillar index.  The lists' order is the token order of the SST stacks, whose activations live in 16-token tiles: this
 * order keeps a tile inside one or two attention windows for both shifts (csrc/mask.hip).  voxel_coors [V,4] int32
 * (b,z,y,x) of the pillars.  Falls back to geomae_random_mask when a sample has more than 2048 window slots. */
int geomae_random_mask_windowed(const int32_t* sample_start, int32_t batch_size, double keep_fraction, uint64_t seed,
                                const int32_t* voxel_coors, const GeomaeWindowConfig* window, int32_t* ids_keep,
                                int32_t* ids_mask, int32_t* token_row, int32_t* counts, geomaeStream_t stream);

/* replaces window_partition + get_voxel_keep_inds + get_flat2win_inds (bb.py:413-681): tokens
 * grouped by window as CSR.  coors [n, 4] int32.  Outputs: win_start [min(n, slots) + 1],
 * win_tokens [n] (token ids grouped by window, ascending inside a window), tok_win [n] (CSR window
 * of each token), tok_pos [n] (in-window position cx * wy + cy, the pos-embed row), num_windows [1],
 * and the packing of consecutive windows into bundles of <= wx*wy tokens that the attention kernels
 * iterate over: bun_start [min(n, slots) + 1] (window index ranges), num_bundles [1]. */
int64_t geomae_window_build_workspace_bytes(int32_t num_tokens, int32_t batch_size,
                                            const GeomaeWindowConfig* cfg);
int geomae_window_build(const int32_t* coors, int32_t num_tokens, int32_t batch_size,
                        const GeomaeWindowConfig* cfg /*host*/, int32_t shift_index, int32_t* win_start,
                        int32_t* win_tokens, int32_t* tok_win, int32_t* tok_pos, int32_t* num_windows,
                        int32_t* bun_start, int32_t* num_bundles, void* workspace, int64_t workspace_bytes,
                        geomaeStream_t stream);

/* Several layouts in one call (<= 4 jobs: a pre-training step builds encoder / decoder tokens x unshifted / shifted
 * windows): every build stage is ONE launch over all jobs instead of one per layout, so the serial chain of small
 * dependent kernels is as long as for a single layout.  Same outputs per job as geomae_window_build. */
typedef struct GeomaeWindowBuildJob {
    const int32_t* coors;       /* [num_tokens, 4] (b, z, y, x) */
    int32_t num_tokens, shift_index;
    int32_t *win_start, *win_tokens, *tok_win, *tok_pos, *num_windows, *bun_start, *num_bundles;
    /* optional "attention plan" (both or neither): bun_tok [min(n, slots) + 1] = position in win_tokens where each
     * bundle starts, pos_info [n][4] = (token, in-window position, window start, window end) per position.  The attention kernels
     * then reach their operands through two dependent loads instead of five. */
    int32_t *bun_tok, *pos_info;
    /* optional second packing (both or neither; needs the plan): the bundles of the one-launch layer kernel,
     * geomae_sst_layer_forward -- at most geomae_window_bundle_cap(num_tokens, wx * wy) tokens each unless a single window is
     * larger: fbun_tok [min(n, slots) + 1] positions where they start, num_fbundles [1] */
    int32_t *fbun_tok, *num_fbundles;
    /* optional WORK ITEMS of the one-launch layer FORWARD (both or neither; needs the second packing; round 6): a third packing
     * with a cap of GeomaeTuning.fwd_item_cap (0 = none made; 32 when switched on) positions whose bundles of three / four tiles are SPLIT by query tile into
     * two items -- a launch lasts as long as its longest work item, and the device has more CUs than a small token set has
     * bundles.  fitems [2 * (min(n, slots) + 1)][4] = (first position, positions, first query tile, query tiles) per item,
     * num_fitems [1]; 0 items = "not applicable" (more bundles than CUs, huge window tables): the kernel then walks fbun_tok */
    int32_t *fitems, *num_fitems;
} GeomaeWindowBuildJob;
int64_t geomae_window_build_batch_workspace_bytes(const int32_t* num_tokens /*host [num_jobs]*/, int32_t num_jobs,
                                                  int32_t batch_size, const GeomaeWindowConfig* cfg);
int geomae_window_build_batch(const GeomaeWindowBuildJob* jobs /*host [num_jobs]*/, int32_t num_jobs,
                              int32_t batch_size, const GeomaeWindowConfig* cfg /*host*/, void* workspace,
                              int64_t workspace_bytes, geomaeStream_t stream);
/* bytes of window tables at the start of geomae_window_build_batch's workspace (what the call clears first) */
int64_t geomae_window_build_batch_table_bytes(const int32_t* num_tokens, int32_t num_jobs, int32_t batch_size,
                                              const GeomaeWindowConfig* cfg);
/* Tokens per bundle of the SECOND packing (fbun_tok) that geomae_window_build_batch makes for a layout of num_tokens tokens
 * (a window larger than the cap is a bundle of its own): small bundles for small token sets, whole windows
 * (max_window_tokens = wx * wy) for large ones.  geomae_sst_layer_forward sizes its grid from it. */
int32_t geomae_window_bundle_cap(int32_t num_tokens, int32_t max_window_tokens);

/* Operator-level window plumbing (mmdet3d/ops/__init__.py:22-26; ops/sst/sst_ops.py:57-135, 225-251, 271-319, 371-388):
 * the hot path works on the CSR layout above and never calls these; they back the reference's function names
 * (geomae_amd.ops.get_inner_win_inds / make_continuous_inds / get_flat2win_inds / flat2window / window2flat).
 * geomae_window_rank: given geomae_pillar_segment_nd over the window ids as (z,y,x) = (0,0,id) -- order, inv,
 * seg_start -- writes continuous_inds[t] = rank of token t's window among the occupied ids (make_continuous_inds) and
 * inner_inds[t] = a ranking 0..n_w-1 of the tokens of each window (get_inner_win_inds: any order is valid, the
 * reference's follows an unstable sort).  geomae_rows_scatter: dst[row_index[i]] = src[i]; geomae_rows_gather:
 * dst[i] = src[row_index[i]]; rows of row_bytes bytes, any element type (flat2window / window2flat copies). */
int geomae_window_rank(const int32_t* order, const int32_t* inv, const int32_t* seg_start, int64_t num_tokens,
                       int64_t* continuous_inds, int64_t* inner_inds, geomaeStream_t stream);
int geomae_rows_scatter(const void* src, const int64_t* row_index, int64_t num_rows, int32_t row_bytes, void* dst,
                        geomaeStream_t stream);
int geomae_rows_gather(const void* src, const int64_t* row_index, int64_t num_rows, int32_t row_bytes, void* dst,
                       geomaeStream_t stream);

/* ------------------------------------------------------------------ A19 windowed attention core
 * replaces flat2window -> nn.MultiheadAttention(key_padding_mask) -> window2flat
 * (sst_basic_block.py:36-59) between the in-projection and the out-projection.
 * qkv [n, 3*H*16] bf16 (q | k | v, already projected, q NOT pre-scaled); out [n, H*16] bf16;
 * lse [n, H] fp32 (log-sum-exp of the scaled scores, kept for the backward pass). */
int geomae_window_attention_forward(const void* qkv_bf16, int32_t num_tokens, int32_t num_heads,
                                    int32_t head_dim, const int32_t* win_start,
                                    const int32_t* win_tokens, const int32_t* tok_win,
                                    const int32_t* bun_start, const int32_t* num_bundles,
                                    int32_t max_bundles, int32_t max_window_tokens, void* out_bf16,
                                    float* lse, const int32_t* bun_tok /*or NULL*/,
                                    const int32_t* pos_info /*or NULL: the build's attention plan*/,
                                    geomaeStream_t stream);
int geomae_window_attention_backward(const void* qkv_bf16, const void* out_bf16, const void* dout_bf16,
                                     const float* lse, int32_t num_tokens, int32_t num_heads,
                                     int32_t head_dim, const int32_t* win_start,
                                     const int32_t* win_tokens, const int32_t* tok_win,
                                     const int32_t* bun_start, const int32_t* num_bundles,
                                     int32_t max_bundles, int32_t max_window_tokens, void* dqkv_bf16,
                                     const int32_t* bun_tok /*or NULL*/, const int32_t* pos_info /*or NULL*/,
                                     geomaeStream_t stream);

/* ------------------------------------------------------------------ A19-A22 fused SST encoder layer
 * replaces EncoderLayer.forward (sst_basic_block.py:85-102) and its autograd: in-projection with the
 * positional embedding folded in, out-projection + residual + LayerNorm + FFN(GELU) + residual +
 * LayerNorm, forward and backward, for d_model = 128, d_ffn = 256.  Residual stream fp32, MFMA operands
 * bf16, fp32 accumulation.  The attention core between qkv and attn is geomae_window_attention_*.
 * Packed weights come from geomae_pack_weights (bf16, contraction index permuted for the MFMA B layout). */
typedef struct GeomaeSstLayerWeights {
    const void *wqkv_p, *wqkT_p, *wvT_p, *wo_p, *woT_p, *w1_p, *w1T_p, *w2_p, *w2T_p; /* packed bf16            */
    const float *bqkv, *bo, *b1, *b2, *ln1_w, *ln1_b, *ln2_w, *ln2_b;                 /* the fp32 parameters   */
    int32_t d_model, d_ffn;                                                           /* must be 128, 256       */
    float ln_eps;
    /* the same nine matrices FRAGMENT-MAJOR (pack descriptor transpose | 4), one block of 262144 bf16 with wqkv at element
     * 0, wqkT 49152, wvT 81920, wo 98304, woT 114688, w1 131072, w1T 163840, w2 196608, w2T 229376 -- what
     * geomae_sst_layer_forward reads; NULL: the one-launch layer kernels are not used. */
    const void* frag_p;
} GeomaeSstLayerWeights;
typedef struct GeomaeSstLayerGrads { /* fp32 gradient buffers, ACCUMULATED into (views of .grad) */
    float *wqkv, *bqkv, *wo, *bo, *w1, *b1, *w2, *b2, *ln1_w, *ln1_b, *ln2_w, *ln2_b;
} GeomaeSstLayerGrads;

/* desc: device int64 [num_desc, 5] rows {src_offset, rows, cols, transpose, dst_offset} (elements);
 * writes bf16 dst[r][p] = W[r][perm(p)] or, transposed (transpose & 1), dst[c][p] = W[perm(p)][c]; transpose & 4: the
 * same matrix stored fragment-major ([row / 16][p / 32][lane = 16 ((p / 8) % 4) + row % 16][p % 8]: the 1-KB piece one
 * MFMA A fragment covers is contiguous).  Packed matrices: with the contraction length a multiple of 8 and dst_offset a
 * multiple of 8 elements a thread writes 8 consecutive bf16 with one 16-byte store (every matrix of this library); other
 * shapes take an element-wise path.  src_offset is relative
 * to flat_params; flat_params may be NULL with src_offset = (device address / 4) for scattered tensors. */
int geomae_pack_weights(const float* flat_params, const int64_t* desc, int32_t num_desc, int64_t max_elems,
                        void* packed_bf16, float* aux_f32 /* target of transpose==2 rows: plain fp32 gather */,
                        geomaeStream_t stream);

/* The layer kernels exist in two forms with bit-identical results: one wave per 16-token tile (64 tokens per workgroup),
 * and a PAIR form (two waves share a tile, 32 tokens per workgroup: half the dependent chain per wave, twice the
 * workgroups) that is chosen when one workgroup per CU covers the launch.  mode: -1 automatic (default), 0 never, 1 always.
 * Process-wide; meant for A/B measurements and tests. */
void geomae_sst_set_pair_kernels(int32_t mode);
/* qkv [n,384] bf16 = [(x + pos_table[tok_pos]) Wqk^T + b | x Wv^T + b];  x [n,128] fp32.  For training also pass
 * x_bf16, xp_bf16 [n,128]: bf16(x) and bf16(x + pos), the operands of this layer's dW_v / dW_qk contraction
 * (geomae_sst_weight_grad); both NULL for inference. */
int geomae_sst_qkv_forward(const float* x, const int32_t* tok_pos, const float* pos_table,
                           const GeomaeSstLayerWeights* w /*host*/, int32_t num_tokens, void* qkv_bf16,
                           void* x_bf16, void* xp_bf16, geomaeStream_t stream);
/* z = LN2(y + W2 gelu(W1 y + b1) + b2), y = LN1(x + attn Wo^T + bo);  attn [n,128] bf16, z [n,128] fp32.
 * For training pass the four save buffers (else all NULL): xhat1, xhat2 [n,128] f32 (normalised residuals),
 * hp [n,256] bf16 (FFN pre-activation), rstd [n,2] f32. */
int geomae_sst_ffn_forward(const float* x, const void* attn_bf16, const GeomaeSstLayerWeights* w,
                           int32_t num_tokens, float* z, float* xhat1, float* xhat2, void* hp_bf16,
                           float* rstd, geomaeStream_t stream);
/* the same, followed -- when next_w != NULL -- by geomae_sst_qkv_forward of the NEXT layer on the z still held in
 * registers (next_tok_pos: window layout of that layer; next_qkv_bf16 [n,384]): one launch instead of two */
int geomae_sst_ffn_qkv_forward(const float* x, const void* attn_bf16, const GeomaeSstLayerWeights* w,
                               int32_t num_tokens, float* z, float* xhat1, float* xhat2, void* hp_bf16, float* rstd,
                               const GeomaeSstLayerWeights* next_w, const int32_t* next_tok_pos, const float* pos_table,
                               void* next_qkv_bf16, void* next_x_bf16, void* next_xp_bf16, geomaeStream_t stream);
/* backward of geomae_sst_ffn_forward from its saved tensors: dx_res [n,128] f32, dattn [n,128] bf16, and the
 * bf16 operands of the weight-gradient GEMMs du,dv,y [n,128], dhp,h [n,256]; LayerNorm parameter
 * gradients are accumulated into grads->ln*.  The incoming gradient is either dz [n,128] (then up_* are NULL) or
 * -- vertical fusion -- computed in the same launch as geomae_sst_qkv_backward of the layer ABOVE:
 * dz = up_dx_res + up_dqkv[:, :256] Wqk(up_w) + up_dqkv[:, 256:] Wv(up_w), never written to memory (dz NULL).
 * up_dx_res may alias dx_res (every wave reads its rows before it writes them). */
int geomae_sst_ffn_backward(const float* xhat1, const float* xhat2, const void* hp_bf16, const float* rstd,
                            const float* dz, const GeomaeSstLayerWeights* w, int32_t num_tokens,
                            float* dx_res, void* dattn_bf16, void* du_bf16, void* dv_bf16, void* dhp_bf16,
                            void* y_bf16, void* h_bf16, const GeomaeSstLayerGrads* grads,
                            const void* up_dqkv_bf16, const float* up_dx_res, const GeomaeSstLayerWeights* up_w,
                            geomaeStream_t stream);
/* dx = dx_res + dqkv[:, :256] Wqk + dqkv[:, 256:] Wv (stand-alone: the first layer of a stack) */
int geomae_sst_qkv_backward(const void* dqkv_bf16, const float* dx_res, const GeomaeSstLayerWeights* w,
                            int32_t num_tokens, float* dx, geomaeStream_t stream);
/* all weight + bias gradients of the layer (token contractions), accumulated into grads */
int geomae_sst_weight_grad(int32_t num_tokens, const void* dqkv_bf16, const void* xp_bf16, const void* x_bf16,
                           const void* du_bf16, const void* attn_bf16, const void* dhp_bf16, const void* y_bf16,
                           const void* dv_bf16, const void* h_bf16, const GeomaeSstLayerGrads* grads,
                           geomaeStream_t stream);

/* ------------------------------------------------------------------ A22 heads + A23 losses, fused
 * replaces the six decoder head Linears on the masked rows (bb.py:279-300) and forward_loss
 * (ssl.py:837-902), forward AND backward, in one kernel.  dec_centroid / dec_density [n,128] f32 are the
 * two decoder outputs (masked rows = num_keep .. num_keep + num_mask - 1).  head_w_packed [800,128] bf16
 * (geomae_pack_weights, rows: reg_low 0.. | cls_low 384.. | reg_med 640.. | cls_med 688.. | reg_top 720.. |
 * zeros | nor_top 768.. | zeros), head_bias [800] f32 in the same order.  Targets as produced by
 * geomae_geometry_targets; occ_counts [2] = occupied low / med cells.  loss_weights (host) [6] and losses
 * (device) [6] in the order curv_around, centroid_low, centroid_med, centroid_top, cls_low, cls_med.
 * Outputs: d_dec_* [n,128] f32 (masked rows written; caller zeroes the rest), and the bf16 operands of
 * the head weight gradients: dlogits [M,896], cm, dm [M,128]. */
int geomae_heads_loss(const float* dec_centroid, const float* dec_density, int32_t num_keep, int32_t num_mask,
                      const void* head_w_packed, const float* head_bias, const float* centroid_low,
                      const uint8_t* mask_low, const float* centroid_med, const uint8_t* mask_med,
                      const float* centroid_top, const float* normal, const int32_t* occ_counts,
                      const float* loss_weights /*host*/, float* losses, float* d_dec_centroid,
                      float* d_dec_density, void* dlogits_bf16, void* cm_bf16, void* dm_bf16,
                      geomaeStream_t stream);
/* same, but `losses` is accumulated into instead of being zeroed first (the caller zeroes it off the critical path) */
int geomae_heads_loss_accumulate(const float* dec_centroid, const float* dec_density, int32_t num_keep,
                                 int32_t num_mask, const void* head_w_packed, const float* head_bias,
                                 const float* centroid_low, const uint8_t* mask_low, const float* centroid_med,
                                 const uint8_t* mask_med, const float* centroid_top, const float* normal,
                                 const int32_t* occ_counts, const float* loss_weights /*host*/, float* losses,
                                 float* d_dec_centroid, float* d_dec_density, void* dlogits_bf16, void* cm_bf16,
                                 void* dm_bf16, geomaeStream_t stream);
/* split form of the same launch: two workgroups per 64 masked pillars (the low-level regression outputs | everything else),
 * each half the dependent chain.  The centroid decoder's output gradient comes back as TWO summands, d_dec_centroid and
 * d_dec_centroid2 (both [num_keep+num_mask,128], zeroed by the caller): pass them to geomae_sst_stack_backward as dz and
 * dz_add.  What geomae_pretrain_step uses. */
int geomae_heads_loss_split_accumulate(const float* dec_centroid, const float* dec_density, int32_t num_keep,
                                       int32_t num_mask, const void* head_w_packed, const float* head_bias,
                                       const float* centroid_low, const uint8_t* mask_low, const float* centroid_med,
                                       const uint8_t* mask_med, const float* centroid_top, const float* normal,
                                       const int32_t* occ_counts, const float* loss_weights /*host*/, float* losses,
                                       float* d_dec_centroid, float* d_dec_centroid2, float* d_dec_density,
                                       void* dlogits_bf16, void* cm_bf16, void* dm_bf16, geomaeStream_t stream);
/* the heads BY DECODER, as two launches with disjoint outputs (dlogits columns [0,768) | [768,896), cm | dm, the two
 * summands of d_dec_centroid | d_dec_density; `losses` accumulated with atomics): the five heads that read the centroid
 * decoder, and the density decoder's normal head.  Each can run on its decoder's stream right behind that decoder's
 * forward, and that stream goes straight on into the decoder's backward.  What geomae_pretrain_step uses. */
int geomae_heads_loss_centroid_accumulate(const float* dec_centroid, int32_t num_keep, int32_t num_mask,
                                          const void* head_w_packed, const float* head_bias, const float* centroid_low,
                                          const uint8_t* mask_low, const float* centroid_med, const uint8_t* mask_med,
                                          const float* centroid_top, const int32_t* occ_counts,
                                          const float* loss_weights /*host*/, float* losses, float* d_dec_centroid,
                                          float* d_dec_centroid2, void* dlogits_bf16, void* cm_bf16, geomaeStream_t stream);
int geomae_heads_loss_density_accumulate(const float* dec_density, int32_t num_keep, int32_t num_mask,
                                         const void* head_w_packed, const float* head_bias, const float* normal,
                                         const float* loss_weights /*host*/, float* losses, float* d_dec_density,
                                         void* dlogits_bf16, void* dm_bf16, geomaeStream_t stream);
typedef struct GeomaeHeadGrads { /* fp32 gradient buffers of the six head Linears, accumulated into */
    float *reg_low_w, *reg_low_b, *cls_low_w, *cls_low_b, *reg_med_w, *reg_med_b, *cls_med_w, *cls_med_b,
          *reg_top_w, *reg_top_b, *nor_top_w, *nor_top_b;
} GeomaeHeadGrads;
int geomae_heads_weight_grad(int32_t num_mask, const void* dlogits_bf16, const void* cm_bf16,
                             const void* dm_bf16, const GeomaeHeadGrads* grads, geomaeStream_t stream);

/* ------------------------------------------------------------------ A3/A4 fused DynamicScatterVFE
 * replaces DynamicScatterVFE.forward (voxel_encoder.py:358-419) + DynamicVFELayer (utils.py:130-144) and
 * their autograd, for in_channels 5 (+3 cluster +3 centre), feat_channels [64, 128], mode 'max', in exact
 * fp32.  BatchNorm is training-mode over all points: each layer is a statistics sweep, a host-visible
 * [2C] fp64 sum vector (where naiveSyncBN1d's cross-rank average happens, ops/norm.py:64-70), and an apply
 * sweep.  Every wave owns 64 consecutive points of the pillar-sorted order; pillars cut by a wave boundary
 * are combined with atomics (max: exact; sums: float).  The [V,*] outputs are zero-filled by the calls. */
typedef struct GeomaeVfeArgs {
    const float* feat_sorted;                        /* [N,16] from geomae_vfe_prepare                   */
    const int32_t* pid_sorted;                       /* [N]                                              */
    const int32_t* seg_start;                        /* from geomae_pillar_segment                       */
    int64_t num_points; int32_t max_pillars;         /* N, and the row count of the [V,*] outputs        */
    const float *w0, *w1;                            /* vfe_layers.0.linear.weight [64,11], .1 [128,128] */
    const float *scale0, *shift0, *scale1, *shift1;  /* folded BatchNorm (geomae_bn_finalize); may be NULL
                                                        for the sweeps that do not need them            */
    const double* moments;                           /* [16 + 121] feature moments of geomae_vfe_prepare_moments, or NULL.
                                                        With them geomae_vfe_stats0 needs no sweep over the points, and
                                                        together with dw0_acc the layer-0 backward needs none either */
    float* dw0_acc;                                  /* [64,16] scratch of the layer-0 weight gradient, or NULL        */
    uint8_t* pillar_ties;                            /* [max_pillars] bytes, cleared by geomae_vfe_layer1 like its other
                                                        outputs, or NULL.  The layer-1 sweep marks every pillar whose maximum
                                                        may be held by more than one of its points in some channel (exact
                                                        duplicates of a point, or two points that round to the same fp32 value:
                                                        ~1 in 10^6 (pillar, channel) pairs).  For an unmarked pillar the
                                                        max-pool routes the gradient of a channel to ONE point, so its share
                                                        of the BatchNorm-backward sums of geomae_vfe_backward_stats follows
                                                        from its [128] rows alone (sum d_vf, sum d_vf * yhat(vf) where vf > 0):
                                                        only the marked pillars' points are swept.  NULL: all points are */
    int32_t layer1_bf16;                             /* 0: the two 128 x 128 layer-1 GEMMs (utils.py:130-144's Linear; forward,
                                                        recomputation, dg = dy1 W1) as bf16 x 3 split products, fp32 grade --
                                                        what compute_dtype 'fp32' and the tight parity tests use.  != 0: ONE
                                                        bf16 product (operands rounded to bf16, fp32 accumulation; BatchNorm
                                                        sums, max-pool routing and everything of layer 0 stay fp32 / fp64):
                                                        the bf16 compute mode of the pre-training step, a third of the MFMA
                                                        issue of those sweeps.  Every sweep of one forward / backward pair must
                                                        see the same value (the backward routes by equality with the recomputed
                                                        forward value) */
} GeomaeVfeArgs;
/* decorated point features [x y z i dt | xyz - pillar mean | xyz - pillar centre | 0...] in pillar order
 * (voxel_encoder.py:372-397); voxel_size (vx,vy,vz) and center_offset = v/2 + range_min are host arrays */
int geomae_vfe_prepare(const float* points, int32_t num_features, int64_t num_points, const int32_t* order,
                       const int32_t* inv, const float* pillar_mean, const int32_t* voxel_coors,
                       const float* voxel_size, const float* center_offset, float* feat_sorted,
                       int32_t* pid_sorted, geomaeStream_t stream);
/* the same + the moments of the 11 decorated features over all points (fp64: S1 [16], S2 [11][11]): layer 0 is linear
 * without bias, so its BatchNorm statistics (forward) and the BatchNorm term of its weight gradient (backward) follow
 * from them -- GeomaeVfeArgs.moments.  workspace: geomae_vfe_moments_workspace_bytes() bytes. */
int64_t geomae_vfe_moments_workspace_bytes(void);
int geomae_vfe_prepare_moments(const float* points, int32_t num_features, int64_t num_points, const int32_t* order,
                               const int32_t* inv, const float* pillar_mean, const int32_t* voxel_coors,
                               const float* voxel_size, const float* center_offset, float* feat_sorted,
                               int32_t* pid_sorted, void* workspace, double* moments /*[137]*/, geomaeStream_t stream);
/* sums [2C] fp64 (sum, sum of squares) and count -> (mean, mean of squares) in moments_out [2C] and/or, when
 * scale != NULL, the folded affine scale/shift, invstd and the running-stat update.  Pass moments_in [2C]
 * instead of sums to finalize from externally averaged moments (naiveSyncBN1d). */
int geomae_bn_finalize(const double* sums, double count, const float* moments_in, int32_t channels,
                       const float* gamma, const float* beta, float eps, float momentum,
                       int32_t unbiased_running_var, float* running_mean, float* running_var, float* scale,
                       float* shift, float* invstd, float* moments_out,
                       int64_t* num_batches_tracked /* += 1 when not NULL (nn.BatchNorm1d's counter) */,
                       geomaeStream_t stream);
int geomae_vfe_stats0(const GeomaeVfeArgs* args /*host*/, double* sums0 /*[128]*/, geomaeStream_t stream);
int geomae_vfe_layer0(const GeomaeVfeArgs* args, float* m0 /*[V,64]*/, double* sums1 /*[256]*/, geomaeStream_t stream);
int geomae_vfe_layer1(const GeomaeVfeArgs* args, const float* m0, float* voxel_feats /*[V,128]*/, geomaeStream_t stream);
/* The same two sweeps with the BatchNorm finalisation FOLDED into them (single-process training: no exchange between the
 * statistics and their use): every workgroup derives scale / shift itself -- layer 0 from the feature moments of the
 * args (W0 S W0^T, fp64), layer 1 from the [256] sums of the statistics sweep -- and workgroup 0 writes what
 * geomae_bn_finalize would have written (scale, shift, invstd, moments = (mean, mean of squares), running statistics with
 * the unbiased variance, the batch counter).  Three single-workgroup launches fewer per forward (vfe_stats0_from_moments,
 * two bn_finalize): ~25 us of a 110 us VFE forward at BASELINE config 2.  args->scale0 / shift0 / scale1 / shift1 must be
 * the `scale` / `shift` arrays of the fold structs (the later sweeps read them there). */
typedef struct GeomaeBnFold {
    double count;                                    /* points behind the statistics */
    const float *gamma, *beta;                       /* [C] */
    float eps, momentum;
    float *running_mean, *running_var;               /* [C], or NULL */
    float *scale, *shift, *invstd, *moments;         /* outputs: [C], [C], [C], [2C] */
    int64_t* num_batches_tracked;                    /* += 1, or NULL */
} GeomaeBnFold;
int geomae_vfe_layer0_bn(const GeomaeVfeArgs* args /* moments != NULL */, const GeomaeBnFold* bn0, float* m0 /*[V,64]*/,
                         double* sums1 /*[256]*/, geomaeStream_t stream);
int geomae_vfe_layer1_bn(const GeomaeVfeArgs* args, const GeomaeBnFold* bn1, const double* sums1 /*[256]*/, const float* m0,
                         float* voxel_feats /*[V,128]*/, geomaeStream_t stream);
/* backward.  GeomaeBnState = what geomae_bn_finalize produced in the forward (scale = gamma * invstd,
 * shift = beta - mean * scale, mean, invstd) for the two BatchNorms.  bsums [2C] fp64 = (sum dh, sum dh * yhat)
 * over the local points: they ARE d beta and d gamma; for naiveSyncBN1d the caller all-reduces them before the
 * next call and passes n_eff = world_size * N (ops/norm.py:21-24), else n_eff = N. */
typedef struct GeomaeBnState {
    const float *scale0, *shift0, *mean0, *invstd0, *scale1, *shift1, *mean1, *invstd1;
} GeomaeBnState;
int geomae_vfe_backward_stats(const GeomaeVfeArgs* args, const GeomaeBnState* bn, const float* m0,
                              const float* voxel_feats, const float* d_voxel_feats, double* bsums1 /*[256]*/,
                              geomaeStream_t stream);
int geomae_vfe_backward_layer1(const GeomaeVfeArgs* args, const GeomaeBnState* bn, const float* m0,
                               const float* voxel_feats, const float* d_voxel_feats, const double* bsums1_global,
                               float n_eff, void* dy1_bf16 /*[ceil16(N),128] tile-blocked*/, void* g_bf16 /*same*/,
                               float* dy1_f32 /* unused since the one-sweep kernel: may be NULL */, float* dh0 /*[N,64]*/, float* dm0 /*[V,64]*/,
                               double* bsums0 /*[128]*/, float* d_beta1 /*[128] += or NULL*/,
                               float* d_gamma1 /*[128] += or NULL*/, geomaeStream_t stream);
int geomae_vfe_backward_layer0(const GeomaeVfeArgs* args, const GeomaeBnState* bn, const float* dh0,
                               const double* bsums0_global, float n_eff, int64_t num_points, const void* dy1_bf16,
                               const void* g_bf16, float* dw0 /*[64,11] +=*/, float* dw1 /*[128,128] +=*/,
                               float* d_beta0 /*[64] += or NULL*/, float* d_gamma0 /*[64] += or NULL*/,
                               geomaeStream_t stream);
/* dy1_bf16 / g_bf16 are scratch operands between geomae_vfe_backward_layer1 and the dW1 contraction, in the TILE-BLOCKED
 * layout of the layer stacks' slabs: [ceil(N/16)][8 channel tiles][16 points][16 channels] bf16 -- (N + 15) / 16 * 16 rows
 * of 256 bytes; the sweep's T-layout lanes then store whole 512-byte blocks (row-major: 32-byte pieces of 16 rows).
 * dw1 += dy1^T g (the layer-1 weight gradient): geomae_vfe_backward_layer0 runs it last unless its dw1 is NULL; a
 * caller may instead launch it on another stream right after geomae_vfe_backward_layer1, beside the two kernels of
 * the layer-0 backward (it is read only by the optimizer). */
int geomae_vfe_weight_grad1(const void* dy1_bf16, const void* g_bf16, int64_t num_points, float* dw1 /*[128,128] +=*/,
                            geomaeStream_t stream);
/* The same product through a caller's split-K workspace (what the step engine runs): up to 96 workgroups each contract a
 * chunk of the points (operand slabs global -> LDS by buffer_load ... lds, csrc/dw_device.h SPLIT job) and leave fp32
 * partials in `workspace`; a second launch sums them in a fixed order and adds the sum to dw1 -- no atomics, the result does
 * not depend on arrival order.  workspace_bytes >= geomae_vfe_weight_grad1_workspace_bytes(); rows of dy1 / g past
 * num_points (the last 16-point block's padding) may hold anything. */
int64_t geomae_vfe_weight_grad1_workspace_bytes(void);
int geomae_vfe_weight_grad1_ws(const void* dy1_bf16, const void* g_bf16, int64_t num_points, float* dw1 /*[128,128] +=*/,
                               void* workspace, int64_t workspace_bytes, geomaeStream_t stream);
/* d_beta / d_gamma (both or neither): single-process callers let the kernels add bsums (= d beta, d gamma) to the
 * BatchNorm parameter gradients; with naiveSyncBN1d the caller adds the LOCAL sums itself before all-reducing them. */

/* ------------------------------------------------------------------ a stack of SST layers in one call
 * replaces the Python block loops of forward_encoder / forward_decoder (bb.py:227-277) and their autograd
 * graph: one call enqueues every kernel of `num_layers` consecutive layers (layer i uses layouts[i & 1]: the
 * unshifted / shifted windows alternate, sst_basic_block.py:133-145).  `saved` (geomae_sst_stack_saved_bytes)
 * carries the activations from forward to backward; `scratch` (geomae_sst_stack_scratch_bytes) is reused by
 * every layer of the backward.  layers / grads are HOST arrays of structs.
 * NOTE on the `saved` blob: it is private to the stack pair (forward writes, backward reads).  With bf16 saved activations
 * (the default) the stack forward stores the bf16 copy of x ONLY for layer 0: the backward forms dW_v's operand of the
 * layers above from the saved xhat2 of the layer below (x = gamma2 * xhat2 + beta2 while the contraction loads its slabs),
 * and dW1's operand y from the saved xhat1.  The x / y slots of those layers are UNWRITTEN: do not pair
 * geomae_sst_stack_forward with op-level geomae_sst_weight_grad calls on slices of the blob (use geomae_sst_stack_backward;
 * GEOMAE_X_FROM_XHAT=0 / GEOMAE_Y_FROM_XHAT=0 restore the stored copies for A/B runs). */
typedef struct GeomaeSstStackLayout {       /* the CSR arrays of geomae_window_build for one shift */
    const int32_t *win_start, *win_tokens, *tok_win, *tok_pos, *bun_start, *num_bundles;
    int32_t max_bundles;
    const int32_t *bun_tok, *pos_info;      /* the build's attention plan, or both NULL */
    const int32_t *fbun_tok, *num_fbundles; /* the build's second packing (one-launch layer kernel), or both NULL */
    const int32_t *fitems, *num_fitems;     /* the build's forward work items (GeomaeWindowBuildJob.fitems), or both NULL */
} GeomaeSstStackLayout;
/* ------------------------------------------------------------------ A19-A22 one launch per layer (forward)
 * EncoderLayer.forward + WindowAttention.forward (sst_basic_block.py:26-61, 85-102) as ONE kernel, one workgroup per bundle
 * of windows: in-projection with the positional term, window attention (q, k, v, P in registers), out-projection +
 * residual + LayerNorm + FFN(GELU) + residual + LayerNorm (csrc/sst_fused.hip).  x [ceil16(n),128] fp32 TILE-BLOCKED
 * ([n/16][8][16 tokens][16 channels]); z likewise, or row-major [n,128] with z_blocked = 0.  `layout` must carry the
 * build's plan and second packing (pos_info, fbun_tok, num_fbundles); bundle_cap = geomae_window_bundle_cap(num_tokens,
 * wx * wy).  The nine save
 * buffers (all or none; what geomae_sst_ffn_backward / geomae_window_attention_backward / geomae_sst_weight_grad read,
 * tile-blocked bf16: qkv [n,384], attn, xhat1, xhat2, x, x + pos [n,128], hp [n,256]; fp32 lse [n,8], rstd [n,2]). */
int geomae_sst_layer_forward(const float* x, int32_t num_tokens, const GeomaeSstLayerWeights* w /*host*/,
                             const GeomaeSstStackLayout* layout /*host*/, int32_t bundle_cap, const float* pos_table,
                             float* z, int32_t z_blocked, void* qkv_bf16, void* attn_bf16, float* lse, void* xhat1_bf16,
                             void* xhat2_bf16, void* hp_bf16, float* rstd, void* x_bf16, void* xp_bf16,
                             geomaeStream_t stream);
/* How geomae_sst_stack_forward runs its layers.  1 (default): through geomae_sst_layer_forward when the layouts carry the
 * plan, the layers have fragment-major weights and the token set is small enough for the one-launch kernel to win (<= 12288
 * tokens); 2: whenever plan and weights allow; 0: always the three-launch form (qkv / attention / ffn kernels).
 * Process-wide; A/B measurements and tests. */
void geomae_sst_set_fused_layers(int32_t mode);

/* which form the last geomae_sst_stack_forward (out[0]) / geomae_sst_stack_backward (out[1]) of THIS host thread took; -1: none
 * yet.  Forward: THREE_LAUNCH = qkv / attention / ffn kernels, ONE_LAUNCH = sst_layer_fwd_kernel (+ its second kernel for
 * bundles of more than four tiles), LOOPING = sst_layer_fwd_ws_kernel.  Backward: THREE_LAUNCH = the ffn / attention backward
 * launches per layer, ONE_LAUNCH = sst_layer_bwd_kernel. */
/* Per host thread, consumed by the NEXT geomae_sst_stack_forward / _backward (default: both): bit s = layout s (unshifted /
 * shifted) MAY hold a bundle of more than four tiles in its second packing.  A caller that knows better (the step engine counts
 * the kept pillars of the fullest window a step ahead, geomae_window_max_keep) clears the bits: the one-launch forward then skips
 * its second kernel for that layout and the backward may take its one-launch form.  The promise is checked on the device:
 * geomae_sst_fused_dropped_bundles (host-synchronising; measurement / tests) counts bundles that no launch ran. */
void geomae_sst_set_big_bundle_layouts(int32_t mask);
int geomae_sst_fused_dropped_bundles(int64_t* out /*host*/, int32_t reset);
enum { GEOMAE_STACK_FORM_THREE_LAUNCH = 0, GEOMAE_STACK_FORM_ONE_LAUNCH = 1, GEOMAE_STACK_FORM_LOOPING = 2 };
int geomae_sst_last_stack_forms(int32_t* out /*host [2]*/);

int64_t geomae_sst_stack_saved_bytes(int32_t num_tokens, int32_t num_layers, int32_t num_heads);
int64_t geomae_sst_stack_scratch_bytes(int32_t num_tokens);
/* scratch for a backward whose weight-gradient contractions are all deferred to geomae_flush_weight_grad (the step
 * engine's schedule): one set of operand slabs per layer instead of two alternating ones */
int64_t geomae_sst_stack_scratch_bytes_layers(int32_t num_tokens, int32_t num_layers);
/* x_in holds num_input_rows rows; the remaining num_tokens - num_input_rows input rows are copies of fill_row [128]
 * (the decoders' mask token, bb.py:239-246).  fill_row == NULL: x_in holds all num_tokens rows.
 * input_rows != NULL: token t (t < num_input_rows) reads row input_rows[t] of x_in (the gather of the kept voxels,
 * bb.py:178, folded into the stack's input conversion). */
int geomae_sst_stack_forward(const float* x_in, int32_t num_tokens, const GeomaeSstLayerWeights* layers,
                             int32_t num_layers, const GeomaeSstStackLayout* layouts /*[2]*/,
                             const float* pos_table, int32_t num_heads, int32_t max_window_tokens, void* saved,
                             int64_t saved_bytes, float* z_out, int32_t num_input_rows, const float* fill_row /*or NULL*/,
                             const int32_t* input_rows /*or NULL*/, void* profiler /*or NULL*/, geomaeStream_t stream);
/* output_rows != NULL: token t's input-gradient row is written to row output_rows[t] of dx_out [num_output_rows, 128]
 * (rows not named keep their contents: the caller zeroes them), the transpose of input_rows above. */
/* dz_add != NULL: the output gradient is dz + dz_add (two consumers of the stack's output, e.g. the two decoders on
 * the encoder's), summed in the top layer's first kernel instead of by the caller. */
int geomae_sst_stack_backward(const float* dz, const float* dz_add /*or NULL*/, int32_t num_tokens, const GeomaeSstLayerWeights* layers,
                              const GeomaeSstLayerGrads* grads, int32_t num_layers,
                              const GeomaeSstStackLayout* layouts, const float* pos_table, int32_t num_heads,
                              int32_t max_window_tokens, const void* saved, void* scratch, int64_t scratch_bytes,
                              float* dx_out, const int32_t* output_rows /*or NULL*/, int32_t num_output_rows,
                              float* tail_sum /*or NULL*/, int32_t tail_from, int32_t defer_last_weight_grad,
                              void* profiler /*or NULL*/, geomaeStream_t stream);
/* tail_sum != NULL: the column sums of the input-gradient rows >= tail_from are ADDED into tail_sum[128] (atomics) by
 * the stack's last data kernel: the gradient of the fill row of geomae_sst_stack_forward (the decoders' mask token). */
/* defer_last_weight_grad != 0: the weight-gradient contraction of the stack's FIRST layer (the last kernel of the
 * backward, read only by the optimizer) is recorded instead of launched; geomae_flush_weight_grad(other_stream)
 * launches it there (order other_stream behind `stream` first), beside whatever the caller enqueues next on `stream`.
 * defer_last_weight_grad == 2: EVERY layer's contraction is recorded ("defer all": what the step engine runs) -- the scratch
 * must then hold one set of operand slabs per layer (geomae_sst_stack_scratch_bytes_layers) and stay alive until the flush;
 * only in this mode, with no bundle of more than four tiles promised (geomae_sst_set_big_bundle_layouts(0)) and a token set
 * the one-launch forward took, does the backward take its one-launch form (geomae_sst_last_stack_forms tells). */
int geomae_flush_weight_grad(geomaeStream_t stream);

/* ------------------------------------------------------------------ N3 DynamicScatter native op
 * replaces the pybind functions dynamic_point_to_voxel_forward / _backward (ops/voxel/src/voxelization.h:112-154,
 * scatter_points_cuda.cu:183-310; Python wrapper ops/voxel/scatter_points.py:11-49).  coors [N, ndim] int32, ndim 3
 * (z,y,x; batch_size 1) or 4 (b,z,y,x); rows with a negative (or out-of-grid) coordinate are dropped like the
 * reference's masked_fill(-1) rows.  Outputs (caller allocated, max_voxels rows; *num_voxels = M on the device):
 * reduced_feats [M,C], out_coors [M,ndim] in lexicographic order (= at::unique_dim sorted), coors_map [N] (voxel
 * row or -1), reduce_count [M].  reduce_type: 0 sum, 1 mean, 2 max.  The grid (gz,gy,gx) bounds the coordinates:
 * DynamicScatter knows it from voxel_size / point_cloud_range.
 * backward: the reference's signature; grad_feats [N,C] is overwritten.  max routes each voxel/channel gradient to
 * the lowest-index point that equals the maximum; reduce_from_ws: num_voxels * C int32 (max only). */
int64_t geomae_dynamic_point_to_voxel_workspace_bytes(int64_t num_points, int32_t max_voxels, int32_t batch_size,
                                                      int32_t gz, int32_t gy, int32_t gx);
int geomae_dynamic_point_to_voxel_forward(const float* feats, const int32_t* coors, int64_t num_points,
                                          int32_t channels, int32_t ndim, int32_t batch_size, int32_t gz, int32_t gy,
                                          int32_t gx, int32_t reduce_type, int32_t max_voxels, float* reduced_feats,
                                          int32_t* out_coors, int32_t* coors_map, int32_t* reduce_count,
                                          int32_t* num_voxels, void* workspace, int64_t workspace_bytes,
                                          geomaeStream_t stream);
int geomae_dynamic_point_to_voxel_backward(float* grad_feats, const float* grad_reduced_feats, const float* feats,
                                           const float* reduced_feats, const int32_t* coors_map,
                                           const int32_t* reduce_count, int64_t num_points, int32_t num_voxels,
                                           int32_t channels, int32_t reduce_type, int32_t* reduce_from_ws,
                                           geomaeStream_t stream);

/* hard voxelization: voxel_layer.hard_voxelize (ops/voxel/src/voxelization.h:58-76; semantics of the CPU path
 * voxelization_cpu.cpp:42-100, wrapper ops/voxel/voxelize.py:44-58).  grid = round((max - min) / voxel_size);
 * coordinates clamped into the grid (this fork); voxels numbered in order of first appearance (point index order),
 * at most max_voxels; each keeps its first max_points points.  Outputs (caller allocated, zero-filled by the call):
 * voxels [max_voxels, max_points, num_features], coors [max_voxels, 3] (z,y,x), num_points_per_voxel [max_voxels],
 * *voxel_num on the device (the pybind function returns it; the caller reads it back to slice). */
int64_t geomae_hard_voxelize_workspace_bytes(int64_t num_points, const float* voxel_size, const float* coors_range);
int geomae_hard_voxelize(const float* points, int64_t num_points, int32_t num_features, const float* voxel_size,
                         const float* coors_range, int32_t max_points, int32_t max_voxels, float* voxels,
                         int32_t* coors, int32_t* num_points_per_voxel, int32_t* voxel_num, void* workspace,
                         int64_t workspace_bytes, geomaeStream_t stream);

/* ------------------------------------------------------------------ N1 fine-tune path pieces
 * geomae_window_drop: SSTInputLayer's region batching drop (models/middle_encoders/sst_input_layer.py:213-238
 * drop_single_shift): keep[i] = (rank of voxel i inside its window) < max_tokens[level], level = first k with
 * range_lower[k] < window count <= range_upper[k] (a count above every range uses the last level).  The rank is
 * the atomic arrival order (the reference shuffles the voxels first, so WHICH voxels of an over-full window
 * survive is random there too).  drop_level [n] (or NULL) receives the level.
 * geomae_recover_bev_*: SSTSecondPretrainedv1.recover_bev (models/backbones/sst_second_pretrained_v1.py:243-280):
 * canvas [B, ny, nx, C] fp32 (an NCHW tensor in channels_last memory format), zero-filled, row (b, y, x) <- feat[i];
 * backward gathers grad_feat[i] <- grad_canvas row. */
int64_t geomae_window_drop_workspace_bytes(int32_t num_tokens, int32_t batch_size, const GeomaeWindowConfig* cfg);
int geomae_window_drop(const int32_t* coors, int32_t num_tokens, int32_t batch_size, const GeomaeWindowConfig* cfg,
                       int32_t shift_index, int32_t num_levels, const int32_t* max_tokens /*host*/,
                       const int32_t* range_lower /*host*/, const int32_t* range_upper /*host*/, uint8_t* keep,
                       int32_t* drop_level, void* workspace, int64_t workspace_bytes, geomaeStream_t stream);
int geomae_recover_bev_forward(const float* feat, const int32_t* coors, int64_t num_tokens, int32_t channels,
                               int32_t batch_size, int32_t ny, int32_t nx, float* canvas, geomaeStream_t stream);
int geomae_recover_bev_backward(const float* grad_canvas, const int32_t* coors, int64_t num_tokens, int32_t channels,
                                int32_t batch_size, int32_t ny, int32_t nx, float* grad_feat, geomaeStream_t stream);

/* ------------------------------------------------------------------ N2 input pipeline
 * replaces, per batch, the CPU transforms of the train pipeline (configs/mae_sst/...6x_1e-5.py:167-197):
 * LoadPointsFromMultiSweeps' per-sweep work (datasets/pipelines/loading.py:184-233: remove_close, sensor->lidar
 * transform, time lag, concatenation), GlobalRotScaleTrans, RandomFlip3D, PointsRangeFilter, PointShuffle
 * (datasets/pipelines/transforms_3d.py:734-757, 125-160, 849-884, 770-797).  The random draws are made by the
 * caller (GeomaeFrameAug); file reading stays on the host.
 * raw_points [N, F] = the sweeps of all frames concatenated (key frame first, then its sweeps), device;
 * sweep_offsets [S+1], frame_offsets [B+1] (raw point index where each sweep / frame starts), sweeps [S],
 * frames [B]: device arrays.  out_points [<= N, F] = frames concatenated, out_offsets [B+1] on the device. */
typedef struct GeomaeSweepInfo {
    double rot[9];            /* sensor2lidar_rotation, row major (applied as xyz @ rot^T)  */
    double trans[3];          /* sensor2lidar_translation                                    */
    float dt;                 /* value of column 4: 0 for the key frame, ts - sweep_ts       */
    int32_t frame;            /* which frame of the batch this sweep belongs to              */
    int32_t remove_close;     /* drop |x| < r and |y| < r (sensor frame)                     */
    int32_t has_transform;    /* 0 for the key frame                                         */
} GeomaeSweepInfo;
typedef struct GeomaeFrameAug {
    float rot_cos, rot_sin, scale;        /* GlobalRotScaleTrans (fp32 sin/cos of the drawn angle) */
    float trans[3];
    int32_t flip_horizontal, flip_vertical;
    uint32_t shuffle_seed_lo, shuffle_seed_hi;   /* both 0: keep the concatenation order */
} GeomaeFrameAug;
int64_t geomae_points_pipeline_workspace_bytes(int64_t num_points, int32_t num_features);
int geomae_points_pipeline(const float* raw_points, int64_t num_points, int32_t num_features,
                           const int32_t* sweep_offsets, const GeomaeSweepInfo* sweeps, int32_t num_sweeps,
                           const int32_t* frame_offsets, const GeomaeFrameAug* frames, int32_t num_frames,
                           const float* point_cloud_range /*host [6]*/, float close_radius, float* out_points,
                           int32_t* out_offsets, void* workspace, int64_t workspace_bytes, geomaeStream_t stream);

/* ------------------------------------------------------------------ N4 optimizer step
 * replaces mmcv OptimizerHook.clip_grads (torch.nn.utils.clip_grad_norm_, max_norm 10, L2) + torch.optim.AdamW
 * as configured by configs/_base_/schedules/cosine_2x.py:1-17, on flat fp32 buffers (16-byte aligned) whose first
 * `num_no_decay` elements are the parameters exempt from weight decay (names containing 'norm').
 * geomae_grad_sumsq: *sumsq = sum g^2 (fp64).  geomae_adamw_step: g *= grad_scale * min(1, max_norm / (norm + 1e-6))
 * with norm = grad_scale * sqrt(*grad_sumsq) (max_norm <= 0: no clipping, grad_sumsq may be NULL), then the
 * AdamW update of torch's single-tensor path op by op in fp32; `step` counts from 1; zero_grad != 0 clears the
 * gradient buffer in the same pass; grad_norm_out (or NULL) receives the pre-clip norm.  A second no-decay range
 * [no_decay2_start, + no_decay2_count) covers flat buffers laid out as two segments (each with its own no-decay
 * prefix) in ONE launch; zero_after (or NULL) is a fp64 word cleared for the next step's geomae_grad_sumsq (two
 * alternating words: no memset between the last backward kernel and the norm reduction). */
int geomae_grad_sumsq(const float* grad, int64_t num_elems, double* sumsq, geomaeStream_t stream);
int geomae_adamw_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t num_elems,
                      int64_t num_no_decay, float lr, float beta1, float beta2, float eps, float weight_decay,
                      int64_t step, float max_norm, const double* grad_sumsq, float grad_scale, int32_t zero_grad,
                      float* grad_norm_out, int64_t no_decay2_start, int64_t no_decay2_count, double* zero_after,
                      geomaeStream_t stream);

/* BatchNorm parameter gradients from the LOCAL backward sums of the fused VFE (naiveSyncBN1d at world > 1: they are
 * added before the sums are all-reduced for the input gradient): d_beta[c] += bsums[c], d_gamma[c] += bsums[C + c]. */
int geomae_bn_param_grad_add(const double* bsums /*[2C]*/, int32_t channels, float* d_beta, float* d_gamma,
                             geomaeStream_t stream);

/* ------------------------------------------------------------------ A24 the whole pre-training step in ONE call
 * replaces the loop body of mmcv's EpochBasedRunner for this detector: MultiSubVoxelDynamicVoxelNetSSL.forward_train
 * (ssl.py:126-242) + backward + OptimizerHook (clip 10, AdamW; configs/_base_/schedules/cosine_2x.py:1-17).  The
 * per-op entry points above stay the operator-level boundary; the engine is the step-level one: the host enqueues
 * ~140 launches on three HIP streams per step, and doing that from Python cost 1.5 ms of a 2.4 ms step (the step was
 * host-bound on any slower CPU).  geomae_pretrain_step issues the same kernels in the same order from C.
 *
 * Memory: ONE caller-allocated workspace (geomae_pretrain_workspace_bytes for the largest batch: max_points points,
 * max_pillars non-empty pillars), carved by the engine per step with the actual sizes; a step that does not fit
 * returns GEOMAE_ERR_WORKSPACE before anything is enqueued.  Host-side state (events, pinned staging buffers for the
 * batch offsets and the per-sample pillar counts) belongs to the engine object.
 * Schedule: stage 1 of batch k+1 (concatenation, voxelize x3, pillar sort, count readback, VFE front, random mask) is
 * enqueued inside step k (`next_*` arguments); the only host wait of a step is that readback's event. */
typedef struct GeomaePretrainConfig {
    int32_t batch_size;              /* frames per step                                                     */
    int32_t num_features;            /* columns of a point row (5)                                          */
    GeomaeTargetConfig targets;      /* grid (z,y,x), sub-voxel ratios, the three voxel sizes, range        */
    GeomaeWindowConfig window;
    int32_t num_heads;               /* 8                                                                   */
    int32_t encoder_layers;          /* 12 (= 2 x encoder_num_blocks)                                       */
    int32_t decoder_layers;          /* 4 per decoder stack                                                 */
    double keep_fraction;            /* 1 - random_mask_ratio                                               */
    uint64_t mask_seed;              /* step i's batch draws its mask with seed (mask_seed << 32) + i + 1   */
    float loss_weights[6];           /* curv_around, centroid_low, centroid_med, centroid_top, cls_low, cls_med */
    float vfe_voxel_size[3];         /* DynamicScatterVFE (vx, vy, vz)                                      */
    float vfe_center_offset[3];      /* v / 2 + range_min                                                   */
    float bn_eps, bn_momentum;
    float beta1, beta2, adam_eps, weight_decay, max_grad_norm;
    int32_t world_size;              /* > 1: gradient-segment hooks; with sync_bn also naiveSyncBN1d's exchanges */
    int32_t sync_bn;                 /* the VFE's norm layers are naiveSyncBN1d (cross-rank statistics at world_size > 1);
                                        0 = plain BatchNorm1d statistics of the local batch whatever the world size */
    int32_t exchange_always;         /* != 0: run the world_size > 1 schedule (hooks, SyncBN exchanges, optimizer as a
                                        separate call) even at world_size 1 -- exercises the RCCL path on one GPU */
    int32_t vfe_bf16;                /* GeomaeVfeArgs.layer1_bf16 of every VFE sweep of the step: != 0 = the voxel encoder's
                                        layer-1 GEMMs as plain bf16 products (the bf16 compute mode, BASELINE config 2) */
} GeomaePretrainConfig;

/* Device pointers of the model (parameters and gradients are views of the flat buffers; all of them must stay where
 * they are for the life of the engine).  layers / layer_grads: HOST arrays, order encoder | centroid decoder |
 * density decoder.  bn_sync_*: caller-owned communication buffers of naiveSyncBN1d (world_size > 1 only). */
typedef struct GeomaePretrainModel {
    const GeomaeSstLayerWeights* layers;
    const GeomaeSstLayerGrads* layer_grads;
    GeomaeHeadGrads head_grads;
    const void* head_w_packed;  const float* head_bias;
    const float* pos_table;
    const float* mask_token;    float* mask_token_grad;
    const int64_t* pack_desc;   int32_t num_pack_desc;   int64_t pack_max_elems;   void* packed;   float* pack_aux;
    const float *vfe_w0, *vfe_w1;   float *vfe_dw0, *vfe_dw1;
    const float *bn_gamma[2], *bn_beta[2];   float *bn_dgamma[2], *bn_dbeta[2];
    float *bn_running_mean[2], *bn_running_var[2];   int64_t* bn_num_batches[2];
    float *params, *grads, *exp_avg, *exp_avg_sq;   int64_t num_params;
    int64_t no_decay_prefix, no_decay2_start, no_decay2_count;
    float *bn_sync_moments0 /*[128]*/, *bn_sync_moments1 /*[256]*/;
    double *bn_sync_bsums1 /*[256]*/, *bn_sync_bsums0 /*[128]*/;
    double *bn_sync_feat_moments /*[2][144]*/;   /* or NULL: layer-0 statistics exchanged in line (BN_FWD0) */
} GeomaePretrainModel;

/* hook(user, what, stream): called from inside geomae_pretrain_step at world_size > 1, with the stream behind which
 * the named data is complete.  BN_*: all-reduce (sum) the matching bn_sync_* buffer on `stream` before returning (the
 * engine divides by world_size itself); GRADS_*: the gradient segment is complete in `stream`'s order -- start its
 * exchange (the engine does not wait for it; the caller does, before geomae_pretrain_optimizer).
 * FEAT_MOMENTS0 / 1 (only with bn_sync_feat_moments): all-reduce (sum) the 144 doubles of slot 0 / 1 of that buffer.  The
 * first VFE layer is linear, so the cross-rank statistics of its BatchNorm follow from the rank-averaged moments of the
 * point features, which depend on the BATCH only: they are exchanged when the batch's stage 1 is done -- at the end of the
 * previous step, or inside geomae_pretrain_submit -- instead of in the middle of the VFE forward (BN_FWD0 is then never
 * raised).  Every rank raises the hooks of one process group in the same order as long as all ranks step together. */
enum { GEOMAE_HOOK_BN_FWD0 = 0, GEOMAE_HOOK_BN_FWD1 = 1, GEOMAE_HOOK_BN_BWD1 = 2, GEOMAE_HOOK_BN_BWD0 = 3,
       GEOMAE_HOOK_GRADS_EARLY = 4, GEOMAE_HOOK_GRADS_ENCODER = 5, GEOMAE_HOOK_FEAT_MOMENTS0 = 6,
       GEOMAE_HOOK_FEAT_MOMENTS1 = 7 };
typedef void (*GeomaePretrainHook)(void* user, int32_t what, geomaeStream_t stream);

int64_t geomae_pretrain_workspace_bytes(const GeomaePretrainConfig* cfg, int64_t max_points, int32_t max_pillars);
/* side_streams[2]: the geometry stream and the decoder-B stream (created by the caller, first used in this order) */
void* geomae_pretrain_create(const GeomaePretrainConfig* cfg, const GeomaePretrainModel* model, void* workspace,
                             int64_t workspace_bytes, int64_t max_points, int32_t max_pillars,
                             const geomaeStream_t* side_streams);
void geomae_pretrain_destroy(void* engine);
int geomae_pretrain_set_hook(void* engine, GeomaePretrainHook hook, void* user);
/* kernel profiler of the stack calls (geomae_profiler_create) or NULL; phase timing on/off: HIP events with timing at
 * the phase boundaries of the main stream (VFE forward | wait for the window layouts | encoder | decoders | heads+loss |
 * decoders backward | encoder backward | VFE backward | clip+AdamW); geomae_pretrain_phase_times waits for the last
 * step's final event and returns the phase durations in ms (up to 9) */
int geomae_pretrain_set_profiler(void* engine, void* profiler);
int geomae_pretrain_set_phase_timing(void* engine, int32_t enabled);
int32_t geomae_pretrain_phase_times(void* engine, float* ms_out /*host*/, int32_t capacity);
/* the bf16 MFMA-layout weight copies are re-packed by the engine after its own optimizer step; call this after
 * anything else wrote the parameters (load_state_dict ...) */
int geomae_pretrain_invalidate_packed(void* engine);
/* stage 1 of a batch, on `stream`, without a step to hide behind (the first batch of a run).  frame_points: HOST
 * array of batch_size DEVICE pointers ([n_b, num_features] fp32 each), frame_sizes: HOST array of row counts. */
int geomae_pretrain_submit(void* engine, const float* const* frame_points, const int64_t* frame_sizes,
                           geomaeStream_t stream);
/* geomae_pretrain_submit with flags.  GEOMAE_SUBMIT_MOMENTS_EXCHANGED: the batch is RE-submitted (to an engine re-created
 * with a larger workspace after geomae_pretrain_step returned GEOMAE_ERR_WORKSPACE on this rank only) and its rank-averaged
 * feature moments -- exchanged at its first submission -- were copied by the caller into the slot of bn_sync_feat_moments
 * this submission uses (slot 0 of a fresh engine): no FEAT_MOMENTS hook is raised, so the sequence of collectives stays
 * the same on every rank.  geomae_pretrain_pending_slot: slot (0 / 1) of the batch submitted last, -1 = none. */
#define GEOMAE_SUBMIT_MOMENTS_EXCHANGED 1
int geomae_pretrain_submit_ex(void* engine, const float* const* frame_points, const int64_t* frame_sizes, int32_t flags,
                              geomaeStream_t stream);
int32_t geomae_pretrain_pending_slot(void* engine);
/* replace the random mask of the batch submitted last by caller-supplied pillar ids (DEVICE int32 arrays, any order,
 * together a permutation of 0..V-1; the reference's get_vanilla_mask_index output, ssl.py:287-304): what parity tests
 * use to run the ENGINE on the reference's own mask.  Waits for the batch's pillar-count readback (host).  Call it
 * between geomae_pretrain_submit and geomae_pretrain_step, on the stream the step will be given. */
int geomae_pretrain_set_mask(void* engine, const int32_t* ids_keep, int32_t num_keep, const int32_t* ids_mask,
                             int32_t num_mask, geomaeStream_t stream);
/* the batch consumed by the i-th step (0-based) of a RUN draws its mask with seed (mask_seed << 32) + i + 1, however
 * often its stage 1 was enqueued (a replaced submission does not shift the stream).  An engine counts the steps it
 * began itself; the counter belongs to the CALLER across engine re-creations (workspace growth, another batch size,
 * checkpoint resume): restore it here (geomae_pretrain_last_sizes out[5] reads it) BEFORE submitting the next batch. */
int geomae_pretrain_set_mask_draws(void* engine, uint64_t steps_begun);
/* Ordering contract for the frames: `frame_points` of geomae_pretrain_submit are read on `stream`; `next_frame_points`
 * of geomae_pretrain_step are read on the decoder-B side stream, which the engine orders behind everything enqueued on
 * `stream` before the call -- so a loader that produces the next batch on the caller's stream (non-blocking H2D
 * copies, augmentation kernels) needs no event of its own.  The frames must stay alive until the step has run.
 *
 * one training step on the batch submitted last (by geomae_pretrain_submit or as the previous step's next_*):
 * forward, backward and -- run_optimizer != 0 -- clip + AdamW with learning rate lr and gradients scaled by
 * grad_scale (1 / world_size).  next_frame_points / next_frame_sizes (or NULL): the following batch.
 * Byte offsets of the step's results inside the workspace: geomae_pretrain_result_offset. */
int geomae_pretrain_step(void* engine, const float* const* next_frame_points, const int64_t* next_frame_sizes,
                         float lr, float grad_scale, int32_t run_optimizer, geomaeStream_t stream);
int geomae_pretrain_optimizer(void* engine, float lr, float grad_scale, geomaeStream_t stream);
/* what: 0 = losses of the LAST step ([6] f32; one of 4 ring slots: valid until three more steps were enqueued),
 * 1 = pre-clip gradient norm ([1] f32), 2 = ids_keep, 3 = ids_mask of the last step ([n_keep] / [n_mask] int32),
 * 4 = voxel_coors of the last step's batch ([V, 4] int32 (b, z, y, x): valid until the step after next reuses the slot) */
int64_t geomae_pretrain_result_offset(void* engine, int32_t what);
/* measurement: cumulative host wall time (seconds) spent inside geomae_pretrain_step calls, the part of it spent
 * waiting for the pillar-count readback, and the number of steps: out[0..2] */
int geomae_pretrain_host_times(void* engine, double* out /*host [3]*/);
/* AdamW's step counter (bias correction): set it when optimizer state is loaded from a checkpoint */
int geomae_pretrain_set_optimizer_steps(void* engine, int64_t steps_taken);
/* host-side sizes of the last step: out[0..5] = N, V, n_keep, n_mask, optimizer steps taken, masks drawn; out[6..7] = kept
 * pillars in the fullest window of the unshifted / shifted layout as counted a step ahead with the random mask (-1: unknown --
 * injected mask, window tables too large), out[8] = which of the two layouts ran the one-launch layer's second kernel for
 * bundles of more than four tiles (bit s = layout s) */
int geomae_pretrain_last_sizes(void* engine, int64_t* out /*host [9]*/);
/* the same with the caller's capacity stated: writes min(capacity, 9) words and returns the number written (negative: error).
 * New callers use this one; the array above grew from 6 to 9 words in round 5 with no way for an older caller to notice. */
int geomae_pretrain_last_sizes_n(void* engine, int64_t* out, int32_t capacity);
/* which kernel form each layer stack of the LAST step took (GEOMAE_STACK_FORM_*): out[0..2] = forward of the encoder, the
 * density decoder, the centroid decoder; out[3..5] = their backward.  Writes min(capacity, 6) words, returns the number written. */
int geomae_pretrain_step_forms(void* engine, int32_t* out, int32_t capacity);

/* measurement only: HIP events recorded on the launch stream around every launch of ONE kernel of the stack
 * calls (bench.py's roofline).  read() synchronises on the events and returns the launch durations in ms. */
enum { GEOMAE_KERNEL_QKV_FWD = 1, GEOMAE_KERNEL_ATTN_FWD = 2, GEOMAE_KERNEL_FFN_FWD = 3, GEOMAE_KERNEL_FFN_BWD = 4,
       GEOMAE_KERNEL_ATTN_BWD = 5, GEOMAE_KERNEL_QKV_BWD = 6, GEOMAE_KERNEL_DW = 7,
       GEOMAE_KERNEL_FFN_BWD_DW = 8 /* the ffn-backward launches that carry a weight-gradient contraction
                                       (sst_ffn_bwd_dw_kernel); 4 = those that do not (sst_ffn_bwd_kernel) */,
       GEOMAE_KERNEL_FFN_FWD_PAIR = 9 /* sst_ffn_fwd_pair_kernel launches; 3 = sst_ffn_fwd_kernel launches */,
       GEOMAE_KERNEL_LAYER_FWD = 10 /* sst_layer_fwd_kernel: the one-launch layer forward */,
       GEOMAE_KERNEL_LAYER_BWD = 11 /* sst_layer_bwd_kernel: the one-launch layer backward */ };
void* geomae_profiler_create(int32_t kernel_id, int32_t max_launches);
int32_t geomae_profiler_read(void* profiler, float* ms_out, int32_t capacity);
void geomae_profiler_destroy(void* profiler);

#ifdef __cplusplus
}
#endif
#endif /* GEOMAE_HIP_H */
