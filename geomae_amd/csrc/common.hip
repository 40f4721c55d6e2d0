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
}  // namespace geomae

extern "C" const char* geomae_last_error(void) { return geomae::g_err; }
extern "C" int32_t geomae_abi_version(void) { return GEOMAE_ABI_VERSION; }
