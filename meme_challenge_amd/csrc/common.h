// Shared host/device helpers for libuniter_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include "../../include/uniter_hip.h"

#define WAVE 64

void uniter_set_error(const char* fmt, ...);

#define UCHECK_ARG(cond, ...)                                   \
  do { if (!(cond)) { uniter_set_error(__VA_ARGS__); return UNITER_E_ARG; } } while (0)
#define UCHECK_SHAPE(cond, ...)                                 \
  do { if (!(cond)) { uniter_set_error(__VA_ARGS__); return UNITER_E_SHAPE; } } while (0)
#define UCHECK_LAUNCH()                                         \
  do { hipError_t e_ = hipGetLastError();                       \
       if (e_ != hipSuccess) { uniter_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, \
                               hipGetErrorString(e_)); return (int)e_; } } while (0)
#define UCHECK_HIP(call)                                        \
  do { hipError_t e_ = (call);                                  \
       if (e_ != hipSuccess) { uniter_set_error("%s:%d %s: %s", __FILE__, __LINE__, #call, \
                               hipGetErrorString(e_)); return (int)e_; } } while (0)
#define UCHECK_RC(call) do { int rc_ = (call); if (rc_ != 0) return rc_; } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

#ifdef __HIPCC__
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float gelu_erf(float x) {
  return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
}
// d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
__device__ __forceinline__ float dgelu_erf(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
#endif
