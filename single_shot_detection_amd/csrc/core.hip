// core.hip -- version, thread-local error string.
#include <stdarg.h>

#include "common.h"

namespace ssdk {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace ssdk

extern "C" int ssdk_version(void) { return SSDK_VERSION; }
extern "C" const char* ssdk_last_error_string(void) { return ssdk::g_err; }
