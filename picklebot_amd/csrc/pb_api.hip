// Error reporting and ABI version for the C-ABI library.
#include "pb_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void pb_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* pb_last_error(void) { return g_err; }
extern "C" int pb_abi_version(void) { return PB_ABI_VERSION; }
