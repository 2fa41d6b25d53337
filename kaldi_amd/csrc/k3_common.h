// k3_common.h -- error plumbing shared by the libk3hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "../../include/k3hip.h"

namespace k3 {
void set_error(const char *fmt, ...);
inline int fail(int code, const char *what, const char *file, int line) {
  set_error("%s (%s:%d)", what, file, line);
  return code;
}
}  // namespace k3

#define K3_HIP_CHECK(expr)                                                              \
  do {                                                                                  \
    hipError_t e__ = (expr);                                                            \
    if (e__ != hipSuccess) {                                                            \
      k3::set_error("HIP error %s at %s:%d: %s", hipGetErrorName(e__), __FILE__, __LINE__, #expr); \
      return K3_ERR_HIP;                                                                \
    }                                                                                   \
  } while (0)
#define K3_REQUIRE(cond, msg)                                   \
  do {                                                          \
    if (!(cond)) return k3::fail(K3_ERR_ARG, msg, __FILE__, __LINE__); \
  } while (0)
