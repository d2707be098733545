// core.hip -- version, thread-local error string.
#include <stdarg.h>
#include <stdlib.h>

#include "common.h"

namespace ssdk {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
// Deterministic mode (ssdk_set_deterministic; default from the environment, SSDK_DETERMINISTIC=1): -1 = not yet read.
static int g_deterministic = -1;
bool deterministic() {
    int v = __atomic_load_n(&g_deterministic, __ATOMIC_RELAXED);
    if (v < 0) {
        const char* e = getenv("SSDK_DETERMINISTIC");
        v = (e && *e && *e != '0') ? 1 : 0;
        __atomic_store_n(&g_deterministic, v, __ATOMIC_RELAXED);
    }
    return v != 0;
}
}  // namespace ssdk

extern "C" int ssdk_set_deterministic(int enabled) {
    const int prev = ssdk::deterministic() ? 1 : 0;
    __atomic_store_n(&ssdk::g_deterministic, enabled ? 1 : 0, __ATOMIC_RELAXED);
    return prev;
}
extern "C" int ssdk_get_deterministic(void) { return ssdk::deterministic() ? 1 : 0; }

extern "C" int ssdk_version(void) { return SSDK_VERSION; }
extern "C" const char* ssdk_last_error_string(void) { return ssdk::g_err; }
