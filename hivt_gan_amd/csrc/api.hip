// Library identity and per-thread error text.
#include "common.h"

#include <string.h>

namespace hivt {
namespace {
thread_local char g_error[512] = {0};
}

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}
}  // namespace hivt

extern "C" int hivt_version(void) { return 100; /* 0.1.0 */ }

extern "C" const char* hivt_last_error(void) { return hivt::g_error; }
