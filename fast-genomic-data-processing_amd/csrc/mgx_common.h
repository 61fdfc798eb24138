// mgx_common.h -- error reporting shared by the C-ABI translation units.
#pragma once
namespace mgx {
// printf-style; stores a thread-local message returned by mgx_last_error().
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
}  // namespace mgx
