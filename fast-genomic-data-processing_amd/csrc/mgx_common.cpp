// mgx_common.cpp -- thread-local last-error string for the C ABI.
#include "mgx_common.h"

#include <cstdarg>
#include <cstdio>

namespace {
thread_local char g_err[512] = "";
}

namespace mgx {
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
}  // namespace mgx

extern "C" const char* mgx_last_error(void) { return g_err; }
