// Row-wise building blocks shared by the LayerNorm and embedding kernels.
// One 64-lane wave owns one row of H floats; lane `l` holds the 16-byte chunks
// l, l+64, ... (NV = ceil(H/256) chunks per lane), so every global access is a
// fully coalesced 1-KiB wave transaction.
#pragma once
#include "common.h"
#include "philox.h"

#ifdef __HIPCC__
template <int NV>
__device__ __forceinline__ void row_load(f32x4 (&v)[NV], const float* __restrict__ p, int H4, int lane) {
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = lane + 64 * k;
    v[k] = c < H4 ? reinterpret_cast<const f32x4*>(p)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
  }
}
template <int NV>
__device__ __forceinline__ void row_store(const f32x4 (&v)[NV], float* __restrict__ p, int H4, int lane) {
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = lane + 64 * k;
    if (c < H4) reinterpret_cast<f32x4*>(p)[c] = v[k];
  }
}
// bf16 (round-to-nearest-even) copy of a row: 4 elements = 8 bytes per lane and vector
template <int NV>
__device__ __forceinline__ void row_store_bf16(const f32x4 (&v)[NV], unsigned short* __restrict__ p, int H4, int lane) {
  typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = lane + 64 * k;
    if (c < H4) {
      bf16x4_t o = {(__bf16)v[k][0], (__bf16)v[k][1], (__bf16)v[k][2], (__bf16)v[k][3]};
      reinterpret_cast<bf16x4_t*>(p)[c] = o;
    }
  }
}
// the three bf16 pieces of the row (x = x1 + x2 + x3 exactly, round-to-nearest residuals: csrc/gemm_split3.hip) at
// p, p + H, p + 2 H: the operand copy of an fp32-accurate product on the bf16 matrix pipe
template <int NV>
__device__ __forceinline__ void row_store_x3(const f32x4 (&v)[NV], unsigned short* __restrict__ p, int H, int H4, int lane) {
  typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = lane + 64 * k;
    if (c < H4) {
      f32x4 r = v[k];
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) {
        const bf16x4_t o = {(__bf16)r[0], (__bf16)r[1], (__bf16)r[2], (__bf16)r[3]};
        reinterpret_cast<bf16x4_t*>(p + (size_t)pc * H)[c] = o;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] -= (float)o[e];
      }
    }
  }
}
// multiply by the dropout keep-mask * 1/(1-p); group0 = index of the row's first 4-element group
template <int NV>
__device__ __forceinline__ void row_dropout(f32x4 (&v)[NV], const DropCfg& d, uint64_t group0, int H4, int lane) {
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = lane + 64 * k;
    if (c < H4) {
      float m[4];
      drop_mult4(d, group0 + (uint64_t)c, m);
      v[k][0] *= m[0]; v[k][1] *= m[1]; v[k][2] *= m[2]; v[k][3] *= m[3];
    }
  }
}
// biased variance, eps = 1e-12 inside the sqrt (apex FusedLayerNorm / F.layer_norm)
template <int NV>
__device__ __forceinline__ void row_stats(const f32x4 (&v)[NV], int H, int H4, int lane, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
  mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if (lane + 64 * k < H4) {
      const float a = v[k][0] - mean, b = v[k][1] - mean, c = v[k][2] - mean, d = v[k][3] - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float var = wave_sum(q) / (float)H;
  rstd = 1.0f / sqrtf(var + 1e-12f);
}
template <int NV>
__device__ __forceinline__ void row_affine(f32x4 (&v)[NV], const float* __restrict__ gamma,
                                           const float* __restrict__ beta, float mean, float rstd,
                                           int H4, int lane) {
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = lane + 64 * k;
    if (c < H4) {
      const f32x4 g = reinterpret_cast<const f32x4*>(gamma)[c];
      const f32x4 b = reinterpret_cast<const f32x4*>(beta)[c];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[k][e] = (v[k][e] - mean) * rstd * g[e] + b[e];
    }
  }
}
// LayerNorm backward for one row.  in: d = dy, zx = z (pre-norm input); out: d = dz.
// Accumulates dgamma += dy * xhat, dbeta += dy per lane-owned column.
template <int NV>
__device__ __forceinline__ void row_ln_bwd(f32x4 (&d)[NV], f32x4 (&zx)[NV], const f32x4 (&g)[NV],
                                           float mu, float rs, f32x4 (&dg)[NV], f32x4 (&db)[NV],
                                           int H, int H4, int lane) {
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const bool ok = lane + 64 * k < H4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = ok ? (zx[k][e] - mu) * rs : 0.f;
      const float dyv = ok ? d[k][e] : 0.f;
      zx[k][e] = xh;
      db[k][e] += dyv;
      dg[k][e] += dyv * xh;
      const float gd = g[k][e] * dyv;
      d[k][e] = gd;
      s1 += gd;
      s2 += gd * xh;
    }
  }
  const float c1 = wave_sum(s1) / (float)H, c2 = wave_sum(s2) / (float)H;
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) d[k][e] = rs * (d[k][e] - c1 - zx[k][e] * c2);
}
// sum the per-wave column partials of a WPB-wave workgroup and store one row of H floats.
// red: >= WPB*NV*256 floats of LDS.
template <int NV, int WPB = 4>
__device__ __forceinline__ void block_col_reduce_store(const f32x4 (&acc)[NV], float* red,
                                                       float* __restrict__ out_row, int H4, int lane, int wave) {
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k)
    *reinterpret_cast<f32x4*>(red + ((wave * NV + k) * 64 + lane) * 4) = acc[k];
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = lane + 64 * k;
      if (c < H4) {
        f32x4 t = acc[k];
#pragma unroll
        for (int w = 1; w < WPB; ++w) t += *reinterpret_cast<const f32x4*>(red + ((w * NV + k) * 64 + lane) * 4);
        reinterpret_cast<f32x4*>(out_row)[c] = t;
      }
    }
  }
}
#endif

// host: out[n] (+)= sum_p part[p*stride + n]   (layernorm.hip)
int finalize_partials(const float* part, int nparts, size_t stride, float* out, int N, int beta,
                      hipStream_t st);
// outs[j][c] += sum_p part[p*stride + j*H + c] for j < nout (<= 8); NULL outputs skipped
int finalize_partials_multi(const float* part, int nparts, size_t stride, float* const* outs, int nout, int H,
                            hipStream_t st);
