// Thread-local error text for the C ABI (include/dib.h: dib_last_error).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/dib.h"

namespace dib {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace dib

extern "C" int dib_abi_version(void) { return DIB_ABI_VERSION; }
extern "C" const char *dib_last_error(void) { return dib::g_err; }
