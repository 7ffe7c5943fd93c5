// C-ABI housekeeping: version + thread-local error string (include/lpm_hip.h).
#include "lpm_common.h"

namespace lpm {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace lpm

extern "C" int lpm_version(void) { return LPM_VERSION; }
extern "C" const char* lpm_last_error(void) { return lpm::g_err; }
