// Error plumbing + version for libvacnic_hip.so (C-ABI never throws/aborts: status + message).
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "ok";

void vacnic_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vacnic_last_error_string(void) { return g_err; }
extern "C" int vacnic_version(void) { return 100; }
