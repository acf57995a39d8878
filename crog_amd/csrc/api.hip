// Library-level entry points: version and error text.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void crog_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int crog_hip_version(void) { return 100; }
extern "C" const char* crog_last_error(void) { return g_err; }
