// Shared helpers for libnbdt_hip.so (gfx950 only -- no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/nbdt_hip.h"

namespace nbdt {

extern thread_local char g_err[512];

inline int fail(int code, const char* fmt, const char* a = "", const char* b = "") {
  snprintf(g_err, sizeof(g_err), fmt, a, b);
  return code;
}

#define NBDT_HIP_CHECK(expr)                                                              \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) return nbdt::fail(NBDT_EHIP, "%s: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

#define NBDT_REQUIRE(cond, msg)                                       \
  do {                                                                \
    if (!(cond)) return nbdt::fail(NBDT_EINVAL, "%s (%s)", msg, #cond); \
  } while (0)

#define NBDT_LAUNCH_CHECK() NBDT_HIP_CHECK(hipGetLastError())

typedef unsigned short bf16_t;  // raw bf16 bits

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((unsigned)v) << 16);
}
// round-to-nearest-even, NaN preserved
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  return (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
}

}  // namespace nbdt
