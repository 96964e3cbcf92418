// Thread-local error message for the C-ABI (include/mgd_hip.h: mgd_last_error).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/mgd_hip.h"

static thread_local char g_err[512] = "";

int mgd_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

thread_local const char* mgd_last_launch_name = "";

extern "C" const char* mgd_last_error(void) { return g_err; }
extern "C" const char* mgd_last_kernel(void) { return mgd_last_launch_name; }
extern "C" int mgd_version(void) { return 1; }
