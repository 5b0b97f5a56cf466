// Shared declarations for libpdae_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PDAE_OK 0
#define PDAE_EINVAL (-1)

// sets the thread-local error string returned by pdae_last_error()
void pdae_set_error(const char* fmt, ...);

#define PDAE_CHECK_ARG(cond, ...)          \
  do {                                     \
    if (!(cond)) {                         \
      pdae_set_error(__VA_ARGS__);         \
      return PDAE_EINVAL;                  \
    }                                      \
  } while (0)

// launch-error check without synchronising the stream
static inline int pdae_launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    pdae_set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return PDAE_OK;
}

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
