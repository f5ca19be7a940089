#include "err.h"
#include <cstdarg>
#include <cstdio>

namespace dgsct {
static thread_local char g_err[512] = {0};
static thread_local bool g_has = false;

void set_error(const char* fmt, ...) {
  if (g_has) return;  // keep the first error of a call
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  g_has = true;
}
const char* last_error() { return g_err; }
bool has_error() { return g_has; }
void clear_error() { g_has = false; g_err[0] = 0; }
}  // namespace dgsct
