// Host-side orchestration of a stack of fused SST layers (encoder: 12, each decoder: 4) in ONE C call.
//
// Reference: BasicShiftBlock / the block loops of MultiMAESSTSPChoose.forward_encoder / forward_decoder
// (mmdet3d/models/sst/sst_basic_block.py:119-147; backbones/multi_mae_sst_spearate_top_only.py:227-277),
// which Python drives layer by layer (and the autograd engine node by node).  With the kernels at 13-60 us
// the per-layer Python / ctypes / allocator work (~100 us a layer) had become the bottleneck of the step, so
// the layer loop lives here: one call enqueues 3 kernels per layer forward, 4 per layer backward, on the
// caller's stream, carving every intermediate out of two caller-provided buffers (no allocation, no sync).
//
// saved   (kept from forward to backward), per layer, 256-byte aligned pieces:
//           x_in f32 [n,128] | qkv bf16 [n,384] | attn bf16 [n,128] | lse f32 [n,H] |
//           xhat1 f32 [n,128] | xhat2 f32 [n,128] | hp bf16 [n,256] | rstd f32 [n,2] | xb, xp bf16 [n,128] each
// scratch (backward only, reused by every layer): dx_res f32 [n,128] | dattn bf16 [n,128] |
//           two sets of bf16 slabs du, dv, y [n,128] each, dhp, h [n,256] each, dqkv [n,384] |
//           two split-K workspaces of the weight-gradient contraction (12.6 MB each)
//
// Kernel chain (vertical fusion: every ~10-50 us kernel of a small stack pays a fixed launch-ramp / first-load /
// store-drain floor, so the two projection kernels ride on their neighbours):
//   forward :  F1(0) | attn(0) | F3(0)+F1(1) | attn(1) | F3(1)+F1(2) | ... | F3(L-1)
//   backward:  B3(L-1) | battn(L-1) | B1(L-1)+B3(L-2) [+ dW(L-1)] | battn(L-2) | ... | B1(1)+B3(0) [+ dW(1)] | battn(0) |
//              B1(0) | dW(0)
#include "common.h"
#include "../../include/geomae_hip.h"
#include <vector>

namespace geomae {
// sst_ws.hip: the forward of one layer as ONE weight-stationary launch (workgroups loop over bundles; windows of up to 144 positions)
int sst_layer_forward_ws(const float* x, const SstInputMap& M, int num_tokens, const GeomaeSstLayerWeights* w,
                         const GeomaeSstStackLayout* layout, const float* pos_table, float* z, bool z_blocked, void* qkv,
                         void* attn, float* lse, void* xh1, void* xh2, void* hp, float* rstd, void* xb, void* xp,
                         int dead_rows, int max_workgroups, int min_tiles, hipStream_t stream);
// sst_fused.hip: the backward of one layer as ONE launch (bundles of at most four tiles)
int sst_layer_backward_fused(const float* dz, const float* dz_add, bool dz_rowmajor, float* dx, bool dx_rowmajor,
                             const int32_t* out_rows, int n_out, int num_tokens, const GeomaeSstLayerWeights* w,
                             const GeomaeSstLayerGrads* g, const GeomaeSstStackLayout* layout, int bundle_cap, const void* qkv,
                             const void* attn, const float* lse, const void* xh1, const void* xh2, const void* hp, const float* rstd,
                             void* dqkv, void* du, void* dv, void* dhp, void* h, hipStream_t stream);
}

namespace geomae {
// sst_layer.hip: lets the weight-gradient contraction of a layer ride inside the next ffn-backward launch
void defer_next_weight_grad();
int flush_pending_weight_grad(hipStream_t stream);
}

namespace geomae {

static inline int64_t al256(int64_t b) { return (b + 255) / 256 * 256; }

// GeomaeTuning.fused_layers (geomae_sst_set_fused_layers) = 0 (never) | 1 (automatic) | 2 (always)
// The one-launch layer kernel is a LATENCY design: one workgroup of 8 waves per bundle and CU (134 KB of LDS), every
// wave walking the whole layer for its channel slice.  It wins where a layer's launches do not fill the chip anyway (the
// encoder's kept pillars at BASELINE configs 1 / 2: 26.1 vs 27.7 us per layer); at 28 k tokens (config 3's encoder) the
// three-launch form, with two workgroups per CU, moves more tokens per us (52 vs 72 us per layer), and so it does at the
// decoders' sizes (tools/fused_layer_time.py).  Automatic = token sets of at most kFusedMaxTokens.
static bool fused_layers_enabled(int num_tokens) {
    const GeomaeTuning& t = tuning();
    return t.fused_layers == 2 || ((t.fused_layers == 1 || t.fused_layers == 3) && num_tokens <= t.fused_max_tokens);
}
// The weight-stationary one-launch layer (sst_ws.hip) takes the token sets ABOVE that range (GeomaeTuning.ws_layers = 1), or
// every token set (2); fused_layers == 2 ("always the one-bundle-per-workgroup form": tests, A/B) wins over it.
static bool ws_layers_enabled(int num_tokens) {
    const GeomaeTuning& t = tuning();
    if (t.fused_layers == 2) return false;
    return t.ws_layers == 2 || (t.ws_layers == 1 && num_tokens > t.fused_max_tokens);
}
static int ws_workgroups() {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        return n > 0 ? n : 256;
    }();
    const int m = tuning().ws_max_workgroups;
    return m > 0 ? m : cus;
}

struct SavedOffsets {
    int64_t x, qkv, attn, lse, xh1, xh2, hp, rstd, xb, xp, stride;
};
// layout flags of sst_layer.hip (kLayBlocked / kLayXBlocked / kLayZBlocked)
constexpr int kBlk = 1, kXBlk = 2, kZBlk = 4, kSavedBf16 = 8;
// bf16 saved x-hat rows (sst_layer.hip kLaySavedBf16); GeomaeTuning.saved_f32 keeps them fp32 (A/B runs, parity checks)
static int saved_flag() { return tuning().saved_f32 ? 0 : kSavedBf16; }

// dW_v's operand x of the layers above the first: formed by the contraction from the layer below's saved xhat2 (bf16 saves
// only; GeomaeTuning.x_from_xhat = 0: the stored copy, A/B runs).  The forward then stores no x copy for those layers.
static bool x_from_xhat_enabled() { return tuning().x_from_xhat && saved_flag() == kSavedBf16; }

// The stack's own buffers are tile-blocked ([n/16][C/16][16][16], sst_device.h "Row layouts"): sized for ceil16(n) rows
static SavedOffsets saved_offsets(int64_t n, int heads) {
    SavedOffsets o;
    n = (n + 15) / 16 * 16;
    int64_t p = 0;
    o.x = p;    p += al256(n * 128 * 4);
    o.qkv = p;  p += al256(n * 384 * 2);
    o.attn = p; p += al256(n * 128 * 2);
    o.lse = p;  p += al256(n * heads * 4);
    o.xh1 = p;  p += al256(n * 128 * 4);
    o.xh2 = p;  p += al256(n * 128 * 4);
    o.hp = p;   p += al256(n * 256 * 2);
    o.rstd = p; p += al256(n * 2 * 4);
    o.xb = p;   p += al256(n * 128 * 2);
    o.xp = p;   p += al256(n * 128 * 2);
    o.stride = p;
    return o;
}

struct ScratchOffsets {
    int64_t dx_res, dattn;
    int64_t set0;                               // first of two identical slab sets read by the weight-gradient kernel
    int64_t du, dv, y, dhp, h, dqkv;            // offsets inside a set
    int64_t set_bytes, dw_partial, total;
};
static ScratchOffsets scratch_offsets(int64_t n, int sets = 2) {
    ScratchOffsets o;
    n = (n + 15) / 16 * 16;
    int64_t p = 0;
    o.dx_res = p; p += al256(n * 128 * 4);
    o.dattn = p;  p += al256(n * 128 * 2);
    o.set0 = p;
    int64_t q = 0;
    o.du = q;     q += al256(n * 128 * 2);
    o.dv = q;     q += al256(n * 128 * 2);
    o.y = q;      q += al256(n * 128 * 2);
    o.dhp = q;    q += al256(n * 256 * 2);
    o.h = q;      q += al256(n * 256 * 2);
    o.dqkv = q;   q += al256(n * 384 * 2);
    o.set_bytes = q;
    o.dw_partial = p + (int64_t)sets * q;       // split-K workspace of the weight-gradient contraction: two buffers
    o.total = o.dw_partial + 2 * kDwPartialBytes;
    return o;
}

// optional per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline)
struct Profiler {
    int kernel_id;
    std::vector<hipEvent_t> ev;     // pairs
    int used = 0;
};
// ids 8 / 9 are the second variants of the ffn launches (include/geomae_hip.h): which one a launch turned out to be is
// known after it (last_kernel_variant), so the pair is kept or dropped in the destructor
struct Timed {
    Profiler* p; hipStream_t s; bool on; int want;
    Timed(void* prof, int id, hipStream_t st) : p((Profiler*)prof), s(st), want(0) {
        int base = p ? p->kernel_id : 0;
        if (base == GEOMAE_KERNEL_FFN_BWD_DW) { base = GEOMAE_KERNEL_FFN_BWD; want = 1; }
        if (base == GEOMAE_KERNEL_FFN_FWD_PAIR) { base = GEOMAE_KERNEL_FFN_FWD; want = 1; }
        on = p && base == id && p->used + 2 <= (int)p->ev.size();
        set_last_kernel_variant(0);
        if (on) (void)hipEventRecord(p->ev[p->used], s);
    }
    ~Timed() {
        if (on && last_kernel_variant() == want) { (void)hipEventRecord(p->ev[p->used + 1], s); p->used += 2; }
    }
};

static thread_local void* t_thread_profiler = nullptr;
void set_thread_profiler(void* prof) { t_thread_profiler = prof; }
void* thread_profiler() { return t_thread_profiler; }
bool profiler_begin(void* prof, int kernel_id, hipStream_t s) {
    Profiler* p = (Profiler*)prof;
    if (!p || p->kernel_id != kernel_id || p->used + 2 > (int)p->ev.size()) return false;
    (void)hipEventRecord(p->ev[p->used], s);
    return true;
}
void profiler_end(void* prof, hipStream_t s) {
    Profiler* p = (Profiler*)prof;
    (void)hipEventRecord(p->ev[p->used + 1], s);
    p->used += 2;
}

}  // namespace geomae

using namespace geomae;

extern "C" void geomae_sst_set_fused_layers(int32_t mode) { tuning_mut().fused_layers = mode < 0 ? 0 : (mode > 2 ? 2 : mode); }

extern "C" void* geomae_profiler_create(int32_t kernel_id, int32_t max_launches) {
    Profiler* p = new Profiler();
    p->kernel_id = kernel_id;
    p->ev.resize((size_t)max_launches * 2);
    for (auto& e : p->ev)
        if (hipEventCreate(&e) != hipSuccess) { delete p; set_error("profiler_create: hipEventCreate failed"); return nullptr; }
    return p;
}
extern "C" int32_t geomae_profiler_read(void* prof, float* ms_out, int32_t capacity) {
    Profiler* p = (Profiler*)prof;
    if (!p) return 0;
    const int n = p->used / 2 < capacity ? p->used / 2 : capacity;
    for (int i = 0; i < n; ++i) {
        (void)hipEventSynchronize(p->ev[2 * i + 1]);
        (void)hipEventElapsedTime(&ms_out[i], p->ev[2 * i], p->ev[2 * i + 1]);
    }
    p->used = 0;
    return n;
}
extern "C" void geomae_profiler_destroy(void* prof) {
    Profiler* p = (Profiler*)prof;
    if (!p) return;
    for (auto& e : p->ev) (void)hipEventDestroy(e);
    delete p;
}

extern "C" int64_t geomae_sst_stack_saved_bytes(int32_t num_tokens, int32_t num_layers, int32_t num_heads) {
    return saved_offsets(num_tokens, num_heads).stride * num_layers + kStackSyncBytes;   // (+ 2 KB kept from the round-5 layout)
}
extern "C" int64_t geomae_sst_stack_scratch_bytes(int32_t num_tokens) { return scratch_offsets(num_tokens).total; }
extern "C" int64_t geomae_sst_stack_scratch_bytes_layers(int32_t num_tokens, int32_t num_layers) {
    return scratch_offsets(num_tokens, num_layers < 2 ? 2 : num_layers).total;
}

static int check_stack(const GeomaeSstLayerWeights* layers, int n_layers, const GeomaeSstStackLayout* lay, const char* who) {
    GEOMAE_REQUIRE(layers && n_layers >= 1 && lay, "%s: null layers / layouts", who);
    for (int s = 0; s < 2; ++s)
        GEOMAE_REQUIRE(lay[s].win_start && lay[s].win_tokens && lay[s].tok_win && lay[s].tok_pos && lay[s].bun_start &&
                       lay[s].num_bundles && lay[s].max_bundles >= 1, "%s: incomplete window layout %d", who, s);
    return GEOMAE_OK;
}

extern "C" int geomae_sst_stack_forward(const float* x_in, int32_t num_tokens, const GeomaeSstLayerWeights* layers,
                                        int32_t num_layers, const GeomaeSstStackLayout* layouts, const float* pos_table,
                                        int32_t num_heads, int32_t max_window_tokens, void* saved, int64_t saved_bytes,
                                        float* z_out, int32_t num_input_rows, const float* fill_row,
                                        const int32_t* input_rows, void* profiler, hipStream_t stream) {
    const int live_row = first_live_row();          // caller's promise (common.h): consumed first, applied per layer below
    set_first_live_row(0);
    const int big_layouts = take_fused_big_layouts();    // (likewise: which layouts may hold a bundle of more than four tiles)
    if (num_tokens <= 0) return GEOMAE_OK;
    int rc = check_stack(layers, num_layers, layouts, "sst_stack_forward");
    if (rc) return rc;
    GEOMAE_REQUIRE(x_in && pos_table && saved && z_out, "sst_stack_forward: null argument");
    if (!fill_row && !input_rows) num_input_rows = num_tokens;
    GEOMAE_REQUIRE(num_input_rows >= 0 && num_input_rows <= num_tokens, "sst_stack_forward: num_input_rows out of range");
    const SavedOffsets so = saved_offsets(num_tokens, num_heads);
    if (saved_bytes < so.stride * num_layers) {
        set_error("sst_stack_forward: saved buffer %lld < %lld bytes", (long long)saved_bytes, (long long)(so.stride * num_layers));
        return GEOMAE_ERR_WORKSPACE;
    }
    char* base = (char*)saved;
    struct LiveRowScope {
        explicit LiveRowScope(int r) { set_first_live_row(r); }
        ~LiveRowScope() { set_first_live_row(0); }
    };
    struct SkipXCopyScope {                         // thread-local switch of the next layer forward: cleared on every exit path
        explicit SkipXCopyScope(bool on) { set_skip_x_copy(on); }
        ~SkipXCopyScope() { set_skip_x_copy(false); }
    };
    // ---- one weight-stationary launch per layer (sst_ws.hip) for the large token sets: the decoders, config 3's encoder
    if (ws_layers_enabled(num_tokens) && layouts[0].fbun_tok && layouts[0].pos_info && layouts[1].fbun_tok && layouts[1].pos_info &&
        saved_flag() == kSavedBf16 && num_heads == 8 && max_window_tokens <= 144 && layers[0].frag_p) {
        const int wgs = ws_workgroups();
        set_last_stack_form(false, GEOMAE_STACK_FORM_LOOPING);
        for (int l = 0; l < num_layers; ++l) {
            char* sv = base + so.stride * l;
            const bool next = l + 1 < num_layers;
            float* z = next ? (float*)(sv + so.stride + so.x) : z_out;
            Timed t(profiler, GEOMAE_KERNEL_LAYER_FWD, stream);
            const SstInputMap M = l == 0 ? SstInputMap{x_in, num_input_rows, fill_row, input_rows} : SstInputMap{nullptr, 0, nullptr, nullptr};
            const bool skip_x = l > 0 && x_from_xhat_enabled();   // (the contraction forms x from the layer below's saved xhat2)
            rc = sst_layer_forward_ws((const float*)(sv + so.x), M, num_tokens, &layers[l], &layouts[l & 1], pos_table, z, next,
                                      sv + so.qkv, sv + so.attn, (float*)(sv + so.lse), sv + so.xh1, sv + so.xh2, sv + so.hp,
                                      (float*)(sv + so.rstd), skip_x ? nullptr : sv + so.xb, sv + so.xp, next ? 0 : live_row, wgs, 1, stream);
            if (rc) return rc;
        }
        return GEOMAE_OK;
    }
    // ---- one launch per layer (sst_fused.hip) when the layouts carry the build's plan.  Same saved tensors, same layouts
    // as the three-launch form below (which stays for layouts without a plan, GEOMAE_SAVED_F32=1 and A/B runs).
    if (fused_layers_enabled(num_tokens) && layouts[0].fbun_tok && layouts[0].pos_info && layouts[1].fbun_tok && layouts[1].pos_info &&
        saved_flag() == kSavedBf16 && num_heads == 8 && max_window_tokens <= 144 && layers[0].frag_p) {
        const int cap = geomae_window_bundle_cap(num_tokens, max_window_tokens);
        set_last_stack_form(false, GEOMAE_STACK_FORM_ONE_LAUNCH);
        for (int l = 0; l < num_layers; ++l) {
            char* sv = base + so.stride * l;
            const bool next = l + 1 < num_layers;
            float* z = next ? (float*)(sv + so.stride + so.x) : z_out;
            Timed t(profiler, GEOMAE_KERNEL_LAYER_FWD, stream);
            if (l == 0) set_input_map(SstInputMap{x_in, num_input_rows, fill_row, input_rows});
            SkipXCopyScope skip(l > 0 && x_from_xhat_enabled());  // (the contraction forms x from the layer below's saved xhat2)
            set_fused_big_next((big_layouts >> (l & 1)) & 1);
            rc = geomae_sst_layer_forward((const float*)(sv + so.x), num_tokens, &layers[l], &layouts[l & 1], cap, pos_table, z,
                                          next ? 1 : 0, sv + so.qkv, sv + so.attn, (float*)(sv + so.lse), sv + so.xh1,
                                          sv + so.xh2, sv + so.hp, (float*)(sv + so.rstd), sv + so.xb, sv + so.xp, stream);
            if (l == 0) set_input_map(SstInputMap{nullptr, 0, nullptr, nullptr});
            if (rc) return rc;
        }
        return GEOMAE_OK;
    }
    set_last_stack_form(false, GEOMAE_STACK_FORM_THREE_LAUNCH);
    // the input conversion (row-major x_in [gathered by input_rows, followed by fill_row] -> tile-blocked x of layer 0)
    // is done by F1 of layer 0 itself (common.h SstInputMap); it used to be a 5-13 us launch in front of every stack
    // F1 of layer l+1 rides at the end of F3 of layer l (geomae_sst_ffn_qkv_forward): 2 launches per layer
    for (int l = 0; l < num_layers; ++l) {
        char* sv = base + so.stride * l;
        const GeomaeSstStackLayout& L = layouts[l & 1];
        const float* x = (const float*)(sv + so.x);
        float* z = (l + 1 < num_layers) ? (float*)(sv + so.stride + so.x) : z_out;
        const bool next = l + 1 < num_layers;
        // everything between the stack's input copy and its output is tile-blocked; z of the last layer is the output
        LayerLayoutScope lay(kBlk | kXBlk | (next ? kZBlk : 0) | saved_flag());
        if (l == 0) {
            Timed t(profiler, GEOMAE_KERNEL_QKV_FWD, stream);
            set_input_map(SstInputMap{x_in, num_input_rows, fill_row, input_rows});
            rc = geomae_sst_qkv_forward(x, L.tok_pos, pos_table, &layers[l], num_tokens, sv + so.qkv, sv + so.xb,
                                        sv + so.xp, stream);
            set_input_map(SstInputMap{nullptr, 0, nullptr, nullptr});
            if (rc) return rc;
        }
        {
            Timed t(profiler, GEOMAE_KERNEL_ATTN_FWD, stream);
            if ((rc = geomae_window_attention_forward(sv + so.qkv, num_tokens, num_heads, 128 / num_heads, L.win_start,
                                                      L.win_tokens, L.tok_win, L.bun_start, L.num_bundles, L.max_bundles,
                                                      max_window_tokens, sv + so.attn, (float*)(sv + so.lse), L.bun_tok,
                                                      L.pos_info, stream)))
                return rc;
        }
        {
            Timed t(profiler, GEOMAE_KERNEL_FFN_FWD, stream);
            LiveRowScope live(next ? 0 : live_row);               // only the LAST layer's output rows can be dead
            SkipXCopyScope skip(x_from_xhat_enabled());           // (layer l + 1 >= 1: its x copy is never read)
            rc = geomae_sst_ffn_qkv_forward(x, sv + so.attn, &layers[l], num_tokens, z, (float*)(sv + so.xh1),
                                                 (float*)(sv + so.xh2), sv + so.hp, (float*)(sv + so.rstd),
                                                 next ? &layers[l + 1] : nullptr, next ? layouts[(l + 1) & 1].tok_pos : nullptr,
                                                 pos_table, next ? sv + so.stride + so.qkv : nullptr,
                                                 next ? sv + so.stride + so.xb : nullptr, next ? sv + so.stride + so.xp : nullptr,
                                                 stream);
            if (rc) return rc;
        }
    }
    return GEOMAE_OK;
}

extern "C" int geomae_sst_stack_backward(const float* dz, const float* dz_add, int32_t num_tokens, const GeomaeSstLayerWeights* layers,
                                         const GeomaeSstLayerGrads* grads, int32_t num_layers,
                                         const GeomaeSstStackLayout* layouts, const float* pos_table, int32_t num_heads,
                                         int32_t max_window_tokens, const void* saved, void* scratch,
                                         int64_t scratch_bytes, float* dx_out, const int32_t* output_rows,
                                         int32_t num_output_rows, float* tail_sum, int32_t tail_from,
                                         int32_t defer_last_weight_grad, void* profiler, hipStream_t stream) {
    const int live_row = first_live_row();          // as in the forward: the TOP layer's dead rows
    set_first_live_row(0);
    const int big_layouts = take_fused_big_layouts();    // (which layouts may hold a bundle of more than four tiles; default: both)
    if (num_tokens <= 0) return GEOMAE_OK;
    int rc = check_stack(layers, num_layers, layouts, "sst_stack_backward");
    if (rc) return rc;
    GEOMAE_REQUIRE(dz && grads && pos_table && saved && scratch && dx_out, "sst_stack_backward: null argument");
    const SavedOffsets so = saved_offsets(num_tokens, num_heads);
    // "defer all": the contractions run after the whole stack, so every layer keeps its own operand slabs
    // (defer_last_weight_grad == 2: the caller's "defer all" -- every layer's contraction queued for geomae_flush_weight_grad,
    //  scratch sized by geomae_sst_stack_scratch_bytes_layers -- what the step engine sets through common.h)
    const bool defer_all = defer_all_weight_grads() || defer_last_weight_grad == 2;
    struct DeferAllArgScope {
        bool on;
        explicit DeferAllArgScope(bool o) : on(o) { if (on) set_defer_all_weight_grads(true); }
        ~DeferAllArgScope() { if (on) set_defer_all_weight_grads(false); }
    } defer_arg_scope(defer_last_weight_grad == 2 && !defer_all_weight_grads());
    const int n_sets = defer_all ? (num_layers < 2 ? 2 : num_layers) : 2;
    const ScratchOffsets sc = scratch_offsets(num_tokens, n_sets);
    if (scratch_bytes < sc.total) {
        set_error("sst_stack_backward: scratch %lld < %lld bytes", (long long)scratch_bytes, (long long)sc.total);
        return GEOMAE_ERR_WORKSPACE;
    }
    // The weight-gradient contraction of layer l only feeds .grad: it rides inside the ffn-backward launch of layer
    // l-1 (sst_ffn_bwd_dw_kernel), reading the slab set the three kernels of layer l left behind while layer l-1
    // writes the other set.  (Running it on a second stream instead was tried and rejected: the cross-queue event
    // hops cost more than the overlap gained, docs/LAB_NOTES.md, rounds 1-3.)
    const char* base = (const char*)saved;
    char* w = (char*)scratch;
    struct DwPartialScope {                         // this stack's contractions sum through the scratch, not atomics
        explicit DwPartialScope(float* ws) { set_dw_partial(ws); }
        ~DwPartialScope() { set_dw_partial(nullptr); }
    } dw_scope((float*)(w + sc.dw_partial));
    struct OperandFormScope {                       // y / x "formed on load" switches: cleared on every exit path
        ~OperandFormScope() { set_y_from_xhat(false); set_x_from_xhat(false); (void)take_dw_dead_rows(); }
    } operand_scope;
    // ---- one launch per layer (sst_fused.hip sst_layer_bwd_kernel) where the forward took its one-launch form, every
    // contraction is queued (defer all: per-layer operand slabs) and no layout holds a bundle of more than four tiles (the
    // caller says so: common.h set_fused_big_layouts; the step engine knows a step ahead).  GEOMAE_FUSED_BWD=0: never.
    const bool fused_bwd_switch = tuning().fused_bwd != 0, y_switch0 = tuning().y_from_xhat != 0;
    const bool fused_bwd = fused_bwd_switch && defer_all && big_layouts == 0 && live_row == 0 && !tail_sum && fused_layers_enabled(num_tokens) &&
                           layouts[0].fbun_tok && layouts[0].pos_info && layouts[1].fbun_tok && layouts[1].pos_info &&
                           saved_flag() == kSavedBf16 && y_switch0 && num_heads == 8 && max_window_tokens <= 144 && layers[0].frag_p;
    set_last_stack_form(true, fused_bwd ? GEOMAE_STACK_FORM_ONE_LAUNCH : GEOMAE_STACK_FORM_THREE_LAUNCH);
    for (int l = num_layers - 1; fused_bwd && l >= 0 && rc == GEOMAE_OK; --l) {
        const char* sv = base + so.stride * l;
        char* ws = w + sc.set0 + l * sc.set_bytes;
        const bool top = l + 1 == num_layers;
        const int cap = geomae_window_bundle_cap(num_tokens, max_window_tokens);
        float* dxbuf = (float*)(w + sc.dx_res);                    // the running input gradient: tile-blocked, rewritten in place
        {
            Timed t(profiler, GEOMAE_KERNEL_LAYER_BWD, stream);
            rc = sst_layer_backward_fused(top ? dz : dxbuf, top ? dz_add : nullptr, top, l == 0 ? dx_out : dxbuf, l == 0,
                                          l == 0 ? output_rows : nullptr, l == 0 ? num_output_rows : 0, num_tokens, &layers[l], &grads[l],
                                          &layouts[l & 1], cap, sv + so.qkv, sv + so.attn, (const float*)(sv + so.lse), sv + so.xh1,
                                          sv + so.xh2, sv + so.hp, (const float*)(sv + so.rstd), ws + sc.dqkv, ws + sc.du, ws + sc.dv,
                                          ws + sc.dhp, ws + sc.h, stream);
        }
        if (rc) break;
        LayerLayoutScope lay(kBlk | saved_flag());
        set_y_from_xhat(true, layers[l].ln1_w, layers[l].ln1_b);
        const bool x_from_xhat = l > 0 && x_from_xhat_enabled();
        set_x_from_xhat(x_from_xhat, x_from_xhat ? layers[l - 1].ln2_w : nullptr, x_from_xhat ? layers[l - 1].ln2_b : nullptr);
        const char* x_operand = x_from_xhat ? base + so.stride * (l - 1) + so.xh2 : sv + so.xb;
        defer_next_weight_grad();                                  // (defer all: queued for geomae_flush_weight_grad)
        rc = geomae_sst_weight_grad(num_tokens, ws + sc.dqkv, sv + so.xp, x_operand, ws + sc.du, sv + so.attn, ws + sc.dhp, sv + so.xh1,
                                    ws + sc.dv, ws + sc.h, &grads[l], stream);
        const DwMidFlush mf = dw_mid_flush();
        if (rc == GEOMAE_OK && l > 0 && mf.side && mf.ev && mf.every > 0 && (num_layers - l) % mf.every == 0) {
            GEOMAE_HIP(hipEventRecord(mf.ev, stream));
            GEOMAE_HIP(hipStreamWaitEvent(mf.side, mf.ev, 0));
            rc = flush_pending_weight_grad(mf.side);
        }
    }
    if (fused_bwd) return rc;
    for (int l = num_layers - 1; l >= 0 && rc == GEOMAE_OK; --l) {
        const char* sv = base + so.stride * l;
        const GeomaeSstStackLayout& L = layouts[l & 1];
        const int set = defer_all ? l : (l & 1);
        char* ws = w + sc.set0 + set * sc.set_bytes;
        const bool top = l + 1 == num_layers;
        LayerLayoutScope lay(kBlk | saved_flag());  // dz (top layer) and dx_out are the row-major boundary tensors
        const char* ws_up = w + sc.set0 + (defer_all ? (l + 1 < num_layers ? l + 1 : l) : ((l + 1) & 1)) * sc.set_bytes;   // slabs of the layer above
        // saved activations in bf16: the ffn backward stores no y = affine(xhat1) copy for dW1, the contraction forms it from
        // the forward's saved xhat1 while it loads its slabs (DwTask.b_scale)
        const bool y_switch = tuning().y_from_xhat != 0;   // (A/B)
        const bool y_from_xhat = y_switch && saved_flag() == kSavedBf16;
        set_y_from_xhat(y_from_xhat, layers[l].ln1_w, layers[l].ln1_b);
        // ... and dW_v's x of the layers above the first from the saved xhat2 of the layer below (x = z of that layer)
        const bool x_from_xhat = l > 0 && x_from_xhat_enabled();
        set_x_from_xhat(x_from_xhat, x_from_xhat ? layers[l - 1].ln2_w : nullptr, x_from_xhat ? layers[l - 1].ln2_b : nullptr);
        const char* x_operand = x_from_xhat ? base + so.stride * (l - 1) + so.xh2 : sv + so.xb;
        {
            // B3(l); for l < L-1 its head is B1(l+1) (dz stays in registers) and dW(l+1) rides in the same launch
            Timed t(profiler, GEOMAE_KERNEL_FFN_BWD, stream);
            if (top) set_dz_addend(dz_add);
            if (top) set_first_live_row(live_row);
            rc = geomae_sst_ffn_backward((const float*)(sv + so.xh1), (const float*)(sv + so.xh2), sv + so.hp,
                                         (const float*)(sv + so.rstd), top ? dz : nullptr, &layers[l], num_tokens,
                                         (float*)(w + sc.dx_res), w + sc.dattn, ws + sc.du, ws + sc.dv, ws + sc.dhp,
                                         ws + sc.y, ws + sc.h, &grads[l], top ? nullptr : ws_up + sc.dqkv,
                                         top ? nullptr : (const float*)(w + sc.dx_res), top ? nullptr : &layers[l + 1], stream);
            set_dz_addend(nullptr);
            set_first_live_row(0);
        }
        if (rc) break;
        {
            Timed t(profiler, GEOMAE_KERNEL_ATTN_BWD, stream);
            rc = geomae_window_attention_backward(sv + so.qkv, sv + so.attn, w + sc.dattn, (const float*)(sv + so.lse),
                                                  num_tokens, num_heads, 128 / num_heads, L.win_start, L.win_tokens,
                                                  L.tok_win, L.bun_start, L.num_bundles, L.max_bundles,
                                                  max_window_tokens, ws + sc.dqkv, L.bun_tok, L.pos_info, stream);
        }
        if (rc) break;
        if (l == 0) {
            Timed t(profiler, GEOMAE_KERNEL_QKV_BWD, stream);
            set_output_rows(output_rows, num_output_rows);
            set_tail_sum(tail_sum, tail_from);
            rc = geomae_sst_qkv_backward(ws + sc.dqkv, (const float*)(w + sc.dx_res), &layers[0], num_tokens, dx_out, stream);
            set_output_rows(nullptr, 0);
            set_tail_sum(nullptr, 0);
            if (rc) break;
        }
        if (top && live_row > 0) set_dw_dead_rows(live_row);       // (consumed by the geomae_sst_weight_grad below)
        if (l > 0) {
            defer_next_weight_grad();               // recorded now, launched inside B3(l-1)
            rc = geomae_sst_weight_grad(num_tokens, ws + sc.dqkv, sv + so.xp, x_operand, ws + sc.du, sv + so.attn,
                                        ws + sc.dhp, y_from_xhat ? sv + so.xh1 : ws + sc.y, ws + sc.dv, ws + sc.h, &grads[l], stream);
            // flushes on the way (common.h DwMidFlush): the layers queued so far go to the side stream now
            const DwMidFlush mf = dw_mid_flush();
            if (rc == GEOMAE_OK && defer_all && mf.side && mf.ev && mf.every > 0 && (num_layers - l) % mf.every == 0) {
                GEOMAE_HIP(hipEventRecord(mf.ev, stream));
                GEOMAE_HIP(hipStreamWaitEvent(mf.side, mf.ev, 0));
                set_dw_budget_hint(mf.budget);
                rc = flush_pending_weight_grad(mf.side);
                (void)take_dw_budget_hint();
            }
        } else {
            // the first layer's contraction feeds nothing but the optimizer: a caller with another stream to spare
            // leaves it recorded and launches it there (geomae_flush_weight_grad), beside whatever follows on `stream`
            if (defer_last_weight_grad || defer_all) defer_next_weight_grad();
            // (timed inside launch_dw, on the stream it really runs on: thread_profiler)
            rc = geomae_sst_weight_grad(num_tokens, ws + sc.dqkv, sv + so.xp, x_operand, ws + sc.du, sv + so.attn,
                                        ws + sc.dhp, y_from_xhat ? sv + so.xh1 : ws + sc.y, ws + sc.dv, ws + sc.h, &grads[l], stream);
        }
    }
    if (!((defer_last_weight_grad || defer_all) && rc == GEOMAE_OK) && flush_pending_weight_grad(stream) != GEOMAE_OK && rc == GEOMAE_OK)
        rc = GEOMAE_ERR_HIP;                                                                         // error paths only
    return rc;
}
