// Riders of the grouped weight-gradient launches (gemm_split3.hip: precision fp32x3; gemm_bf16_dma.hip: precision bf16): the
// column-reduction items that the workgroups of the launch run beside their tiles (uniter_x3_riders_t, include/uniter_hip.h).
#pragma once
#include "common.h"

#ifdef __HIPCC__
// One item: out[c] += sum_p part[p * stride + c] for 64 columns, one 16-column strip per wave of the four that call it (lane =
// 16 partial-row slices x 4 sixteen-byte column groups; the slices meet by wave shuffles: no LDS, no workgroup barrier).  Returns
// the sum of squares of what the lane wrote.
__device__ __forceinline__ float riders_reduce_item(const uniter_x3_riders_t& x, int r, int wave, int lane) {
  int j = 0;
  while (j + 1 < x.njobs && r >= x.first_item[j + 1]) ++j;
  const int q = lane & 3, sl = lane >> 2;
  const int col = (r - x.first_item[j]) * 64 + wave * 16 + q * 4;
  const int n = x.n[j], nparts = x.nparts[j];
  const float* __restrict__ part = x.part[j];
  const size_t stride = (size_t)x.stride[j];
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  if (col < n) {
    int p = sl;
    for (; p + 48 < nparts; p += 64) {        // four independent 16-byte loads in flight per lane
      a0 += *reinterpret_cast<const f32x4*>(part + (size_t)p * stride + col);
      a1 += *reinterpret_cast<const f32x4*>(part + (size_t)(p + 16) * stride + col);
      a2 += *reinterpret_cast<const f32x4*>(part + (size_t)(p + 32) * stride + col);
      a3 += *reinterpret_cast<const f32x4*>(part + (size_t)(p + 48) * stride + col);
    }
    for (; p < nparts; p += 16) a0 += *reinterpret_cast<const f32x4*>(part + (size_t)p * stride + col);
  }
  f32x4 t = (a0 + a1) + (a2 + a3);
#pragma unroll
  for (int o = 4; o < 64; o <<= 1)
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] += __shfl_xor(t[e], o, 64);
  float ss = 0.f;
  if (sl == 0 && col < n) {
    const int seg = x.seg[j];
    float* o = x.out[j][col / seg];
    if (o) {
      f32x4 v = *reinterpret_cast<f32x4*>(o + col % seg);
      v += t;
      *reinterpret_cast<f32x4*>(o + col % seg) = v;
      ss = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
  }
  return ss;
}

// the wave's sum of squares into its slot: `nw` slots per workgroup (4, or the 8 compute waves of a 128 x 256 tile)
__device__ __forceinline__ void riders_store_ssq(const uniter_x3_riders_t& x, double wss, int wave, int lane, int nw = 4) {
  if (x.ssq) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wss += __shfl_xor(wss, o, 64);
    if (lane == 0) x.ssq[(size_t)blockIdx.x * nw + wave] = wss;
  }
}
#endif

// host side: checks a riders block and fills in first_item / nred (grid is the caller's)
int riders_prepare(uniter_x3_riders_t& x, const char* who);
