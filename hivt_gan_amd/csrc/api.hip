// Library identity and per-thread error text.
#include "common.h"

#include <string.h>

namespace hivt {
namespace {
thread_local char g_error[512] = {0};
}

__global__ void zero_words_kernel(uint32_t* __restrict__ p, int64_t words) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (int64_t)gridDim.x * blockDim.x)
    p[i] = 0u;
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
