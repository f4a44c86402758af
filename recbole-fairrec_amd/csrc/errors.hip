// Error string plumbing and version of libfairrec_hip.so.
#include <stdarg.h>

#include "common.hpp"

namespace fr {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace fr

extern "C" int fr_version(void) { return 1; }
extern "C" const char* fr_last_error(void) { return fr::g_err; }
