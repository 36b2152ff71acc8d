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

// In-kernel launch stamps (uniter_prof_enable_stamps; bench.py's per-family roofline): when non-null, the NEXT GEMM
// launch stores every workgroup's start and end time (100 MHz real-time clock) into its own words of this device slot
// ([2][STAMP_WGS]: starts, then ends) -- two plain 8-byte stores per workgroup, reduced on the host afterwards.
// Measured alternatives: a HIP event pair around each launch costs ~7 us on the stream and breaks the overlap of the
// two backward streams (fp32 step +11 %, bf16 +27 %); 64-bit atomicMin / atomicMax into ONE slot serialise at the
// memory side (~50 ns each, thousands per launch: +16 % / +35 %).  The launch that reads the pointer resets it.
// thread_local: the model call that sets it and the launch that takes it run on one host thread (the schedule of a model call
// is one C function); another thread's launches have their own.
#define STAMP_WGS 1024
extern thread_local unsigned long long* g_uniter_stamp_slot;
static inline unsigned long long* take_stamp_slot() {
  unsigned long long* s = g_uniter_stamp_slot;
  g_uniter_stamp_slot = nullptr;
  return s;
}

// Wave priority of the NEXT matrix-kernel launch (0 = default, 1..3 = s_setprio level at kernel entry; the launch that reads
// it resets it).  fp32 matrix and vector instructions share one issue pipe per SIMD and two kernels that share a CU are
// arbitrated wave by wave by priority, then age: the model's schedule gives the kernels of its critical path (forward,
// input-gradient chain, attention, row passes) a higher level than the weight-gradient launches of the side stream, which have
// slack (HIP stream priorities do not reach this arbitration: measured no effect).  thread_local, as the stamp slot.
extern thread_local int g_uniter_launch_prio;
static inline int take_launch_prio() {
  const int p = g_uniter_launch_prio;
  g_uniter_launch_prio = 0;
  return p;
}

// Dropout keep flags drawn ahead for the NEXT LayerNorm row pass (model.cpp sets it, the launch that reads it resets it): see
// DropCfg::bits.  thread_local, as the channels above.
extern thread_local const unsigned char* g_uniter_drop_bits;
static inline const unsigned char* take_drop_bits() {
  const unsigned char* p = g_uniter_drop_bits;
  g_uniter_drop_bits = nullptr;
  return p;
}

// CUs the persistent matrix kernels leave free (uniter_model_set_cu_reserve: the collectives of a data-parallel exchange run beside the
// backward pass and need CUs of their own -- a persistent launch that counts on all of them leaves its last workgroups queued behind
// the collective's and their whole share of the tiles late).  Set by every model call from its handle; thread_local as the channels above.
extern thread_local int g_uniter_cu_reserve;

#ifdef __HIPCC__
// uniform branch around the immediate-operand instruction
__device__ __forceinline__ void set_wave_prio(int level) {
  if (level >= 3) __builtin_amdgcn_s_setprio(3);
  else if (level == 2) __builtin_amdgcn_s_setprio(2);
  else if (level == 1) __builtin_amdgcn_s_setprio(1);
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void stamp_begin(unsigned long long* slot) {
  if (slot && threadIdx.x == 0 && blockIdx.x < STAMP_WGS) slot[blockIdx.x] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
}
// call where every thread of the workgroup arrives (the end of the kernel): the barrier makes thread 0's clock read the
// end of the workgroup's last wave (its LDS is held until then anyway); stores still in flight are not waited for
__device__ __forceinline__ void stamp_end(unsigned long long* slot) {
  if (slot) {
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x < STAMP_WGS) slot[STAMP_WGS + blockIdx.x] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
  }
}
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
// Branch-free erf for the GEMM epilogues: odd rational x P(x^2) / Q(x^2) on [-4, 4] (|erf| = 1 to
// fp32 beyond), 13 FMAs + one reciprocal; max abs error 4.5e-7 (checked against float64 erf over
// [-6, 6]).  The libm erff is ~3x the instructions, and the FFN epilogues evaluate 8 M of them per
// layer -- VALU time the MFMAs cannot hide because all co-resident tiles finish together.
__device__ __forceinline__ float erf_fast(float a) {
  const float x = fminf(fmaxf(a, -4.0f), 4.0f);
  const float x2 = x * x;
  float p = -2.72614225801306e-10f;
  p = p * x2 + 2.77068142495902e-08f;
  p = p * x2 + -2.10102402082508e-06f;
  p = p * x2 + -5.69250639462346e-05f;
  p = p * x2 + -7.34990630326855e-04f;
  p = p * x2 + -2.95459980854025e-03f;
  p = p * x2 + -1.60960333262415e-02f;
  float q = -1.45660718464996e-05f;
  q = q * x2 + -2.13374055278905e-04f;
  q = q * x2 + -1.68282697438203e-03f;
  q = q * x2 + -7.37332916720468e-03f;
  q = q * x2 + -1.42647390514189e-02f;
  return x * p * __builtin_amdgcn_rcpf(q);
}
// gelu(x) and gelu'(x) for the GEMM epilogues, sharing ONE exponential: with E = exp(-x^2/2),
//   erfc(|x|/sqrt 2) = P(t) E,  t = 1 / (1 + p |x| / sqrt 2)   (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7),
// so Phi(x) = 1 - H (x >= 0) or H (x < 0) with H = P E / 2, and phi(x) = E / sqrt(2 pi).  17 VALU operations, two of
// them transcendental (v_rcp, v_exp), against 25 for erf_fast + a separate exp.  Checked against float64 over
// [-12, 12]: |Phi| 3.0e-7, |gelu| 4.2e-7, |gelu'| 3.0e-7 (the tail x < 0 is formed without cancellation).
__device__ __forceinline__ void gelu_pair_fast(float x, float& g, float& dg) {
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(x), 0.23164189f, 1.0f));     // p / sqrt 2
  float P = 0.5307027145f;                       // a5 / 2 .. a1 / 2
  P = __builtin_fmaf(P, t, -0.7265760135f);
  P = __builtin_fmaf(P, t, 0.7107068705f);
  P = __builtin_fmaf(P, t, -0.142248368f);
  P = __builtin_fmaf(P, t, 0.127414796f);
  const float E = __builtin_amdgcn_exp2f(x * x * -0.72134752044f);          // exp(-x^2 / 2)
  const float H = P * t * E;
  const float cdf = x >= 0.f ? 1.0f - H : H;
  g = x * cdf;
  // one fused op on purpose: with separate mul + add the compiler pairs neighbouring elements into
  // v_pk_mul_f32 / v_pk_add_f32 (op_sel), and that sequence dropped the cdf term on a few lanes per launch in the
  // 128x128 bf16 kernels on gfx950 (tests/tools/gemm_glitch_screen.py: 12 of 12 launches bad before, none with the fma)
  dg = __builtin_fmaf(x, E * 0.39894228040143267794f, cdf);
}
// d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
__device__ __forceinline__ float dgelu_erf(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
#endif
