// Error reporting and version of libmusicgan_hip.so.
#include "mg_common.h"

namespace {
thread_local char g_err[512] = "";
}

void mg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int mg_version(void) { return 100; }
extern "C" const char* mg_last_error(void) { return g_err; }
