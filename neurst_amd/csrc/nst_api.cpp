// Error plumbing + ABI version for libneurst_hip.so.
#include <stdarg.h>

#include "nst_common.h"

static thread_local char g_err[512] = "";

void nst_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int nst_abi_version(void) { return NST_ABI_VERSION; }
extern "C" const char* nst_last_error_string(void) { return g_err; }
