// Error reporting + ABI version for libgeomae_hip.
#include "common.h"
#include "../../include/geomae_hip.h"
#include <stdarg.h>

namespace geomae {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
static thread_local int g_layer_layout = 0;
void set_layer_layout(int flags) { g_layer_layout = flags; }
int layer_layout() { return g_layer_layout; }
static thread_local const int32_t* g_out_rows = nullptr;
static thread_local int g_out_rows_n = 0;
void set_output_rows(const int32_t* rows, int num_rows_out) { g_out_rows = rows; g_out_rows_n = num_rows_out; }
const int32_t* output_rows(int* num_rows_out) { *num_rows_out = g_out_rows_n; return g_out_rows; }
static thread_local const float* g_dz_addend = nullptr;
void set_dz_addend(const float* dz2) { g_dz_addend = dz2; }
const float* dz_addend() { return g_dz_addend; }
static thread_local float* g_tail_sum = nullptr;
static thread_local int g_tail_from = 0;
void set_tail_sum(float* sum, int from_row) { g_tail_sum = sum; g_tail_from = from_row; }
float* tail_sum(int* from_row) { *from_row = g_tail_from; return g_tail_sum; }
static thread_local SstInputMap g_input_map = {nullptr, 0, nullptr, nullptr};
void set_input_map(const SstInputMap& m) { g_input_map = m; }
SstInputMap input_map() { return g_input_map; }
static thread_local int g_first_live_row = 0;
void set_first_live_row(int row) { g_first_live_row = row > 0 ? row : 0; }
int first_live_row() { return g_first_live_row; }
static thread_local hipEvent_t g_mid_launch_event = nullptr;
void set_mid_launch_event(hipEvent_t ev) { g_mid_launch_event = ev; }
hipEvent_t take_mid_launch_event() { hipEvent_t ev = g_mid_launch_event; g_mid_launch_event = nullptr; return ev; }
static thread_local hipStream_t g_mid_launch_side = nullptr;
void set_mid_launch_side(hipStream_t side) { g_mid_launch_side = side; }
hipStream_t take_mid_launch_side() { hipStream_t s = g_mid_launch_side; g_mid_launch_side = nullptr; return s; }
static thread_local int g_last_kernel_variant = 0;
void set_last_kernel_variant(int v) { g_last_kernel_variant = v; }
int last_kernel_variant() { return g_last_kernel_variant; }
static thread_local float* g_dw_partial = nullptr;
void set_dw_partial(float* ws) { g_dw_partial = ws; }
float* dw_partial() { return g_dw_partial; }
static thread_local int g_fused_big_layouts = 3;
void set_fused_big_layouts(int mask) { g_fused_big_layouts = mask & 3; }
int take_fused_big_layouts() { const int m = g_fused_big_layouts; g_fused_big_layouts = 3; return m; }
static thread_local int g_last_stack_form[2] = {-1, -1};
void set_last_stack_form(bool backward, int form) { g_last_stack_form[backward ? 1 : 0] = form; }
int last_stack_form(bool backward) { return g_last_stack_form[backward ? 1 : 0]; }
static thread_local bool g_fused_big_next = true;
void set_fused_big_next(bool possible) { g_fused_big_next = possible; }
bool take_fused_big_next() { const bool b = g_fused_big_next; g_fused_big_next = true; return b; }
static thread_local DwMidFlush g_dw_mid_flush;
void set_dw_mid_flush(const DwMidFlush& f) { g_dw_mid_flush = f; }
DwMidFlush dw_mid_flush() { return g_dw_mid_flush; }
static thread_local int g_dw_budget_hint = 0;
void set_dw_budget_hint(int w) { g_dw_budget_hint = w; }
int take_dw_budget_hint() { const int w = g_dw_budget_hint; g_dw_budget_hint = 0; return w; }
static thread_local int g_dw_dead_rows = 0;
void set_dw_dead_rows(int rows) { g_dw_dead_rows = rows > 0 ? rows : 0; }
int take_dw_dead_rows() { const int r = g_dw_dead_rows; g_dw_dead_rows = 0; return r; }
static thread_local bool g_win_prezeroed = false;
void set_window_tables_prezeroed(bool on) { g_win_prezeroed = on; }
bool window_tables_prezeroed() { return g_win_prezeroed; }
static thread_local bool g_prezeroed = false;
bool accumulators_prezeroed() { return g_prezeroed; }
}  // namespace geomae

extern "C" int geomae_set_accumulators_prezeroed(int32_t enabled) {
    geomae::g_prezeroed = enabled != 0;
    return GEOMAE_OK;
}

namespace geomae {
static GeomaeTuning default_tuning() {
    GeomaeTuning t;
    memset(&t, 0, sizeof(t));
    t.size = (int32_t)sizeof(GeomaeTuning);
    t.fused_layers = 1; t.fused_max_tokens = 12288; t.fused_bwd = 1;
    t.ws_layers = 0; t.ws_bwd = 0; t.ws_bundle_cap = 144; t.ws_max_workgroups = 0; t.bundle_cap = 0;
    t.saved_f32 = 0; t.x_from_xhat = 1; t.y_from_xhat = 1; t.pair_kernels = -1; t.attn_heads = 0;
    t.dw_layer_form = 1; t.dw_chunks = 0; t.dw_budget_mid = 0; t.dw_split_reduce = 1;
    t.dw_defer_all = 1; t.dec_dw_every = -1; t.dec_mid_budget = 0; t.enc_dw_defer = 1; t.zero_late_aux = 0;
    t.fused_skip_big = 1; t.heads_joint = 0; t.fwd_item_cap = 0;
    return t;
}
static GeomaeTuning g_tuning = default_tuning();
const GeomaeTuning& tuning() { return g_tuning; }
GeomaeTuning& tuning_mut() { return g_tuning; }
}  // namespace geomae

extern "C" int geomae_get_tuning(GeomaeTuning* out) {
    GEOMAE_REQUIRE(out, "get_tuning: null argument");
    *out = geomae::g_tuning;
    return GEOMAE_OK;
}
extern "C" int geomae_set_tuning(const GeomaeTuning* in) {
    GEOMAE_REQUIRE(in && in->size == (int32_t)sizeof(GeomaeTuning), "set_tuning: null argument or a struct of another size (%d, expected %d)",
                   in ? in->size : -1, (int)sizeof(GeomaeTuning));
    GeomaeTuning t = *in;
    auto clamp = [](int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); };
    t.fused_layers = clamp(t.fused_layers, 0, 3); t.fused_max_tokens = clamp(t.fused_max_tokens, 0, 1 << 30);
    t.ws_layers = clamp(t.ws_layers, 0, 2); t.ws_bundle_cap = clamp(t.ws_bundle_cap, 16, 144);
    t.ws_max_workgroups = clamp(t.ws_max_workgroups, 0, 4096); t.bundle_cap = clamp(t.bundle_cap, 0, 144);
    t.pair_kernels = clamp(t.pair_kernels, -1, 1);
    t.attn_heads = (t.attn_heads == 1 || t.attn_heads == 2 || t.attn_heads == 4) ? t.attn_heads : 0;
    t.dw_chunks = clamp(t.dw_chunks, 0, 64); t.dw_budget_mid = clamp(t.dw_budget_mid, 0, 4096);
    t.dec_dw_every = clamp(t.dec_dw_every, -1, 64); t.dec_mid_budget = clamp(t.dec_mid_budget, 0, 4096);
    t.fwd_item_cap = t.fwd_item_cap <= 0 ? 0 : clamp(t.fwd_item_cap, 16, 64);
    geomae::g_tuning = t;
    return GEOMAE_OK;
}

extern "C" void geomae_sst_set_big_bundle_layouts(int32_t mask) { geomae::set_fused_big_layouts(mask); }

extern "C" int geomae_sst_last_stack_forms(int32_t* out) {
    GEOMAE_REQUIRE(out, "sst_last_stack_forms: null argument");
    out[0] = geomae::last_stack_form(false);
    out[1] = geomae::last_stack_form(true);
    return GEOMAE_OK;
}

extern "C" const char* geomae_last_error(void) { return geomae::g_err; }
extern "C" int32_t geomae_abi_version(void) { return GEOMAE_ABI_VERSION; }
