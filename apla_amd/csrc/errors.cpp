// Thread-local error message + version for the C-ABI.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/apla_hip.h"

static thread_local char g_err[512] = "";

void apla_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* apla_last_error(void) { return g_err; }
extern "C" int apla_version(void) { return 100; }

extern "C" int apla_operand_dtype(void) {
#if defined(APLA_FP16)
  return APLA_F16;
#else
  return APLA_BF16;
#endif
}
