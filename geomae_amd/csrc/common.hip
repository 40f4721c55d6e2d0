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

extern "C" const char* geomae_last_error(void) { return geomae::g_err; }
extern "C" int32_t geomae_abi_version(void) { return GEOMAE_ABI_VERSION; }
