// Error plumbing and misc entry points of libdiffnorm_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

static thread_local char g_err[512] = "";

void dn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* dn_last_error(void) { return g_err; }
extern "C" int dn_version(void) { return 100; }
