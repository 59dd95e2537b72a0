// Error plumbing + ABI version for the C-ABI in include/sf_hip.h.
#include <cstdarg>
#include <cstdio>
#include "../../include/sf_hip.h"

static thread_local char g_err[512] = "";

extern "C" void sf_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* sf_last_error(void) { return g_err; }
extern "C" int sf_abi_version(void) { return SF_HIP_ABI_VERSION; }
