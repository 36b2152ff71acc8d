// Fused dropout + residual + LayerNorm forward / backward, and column sums.
//
// Replaces nn.Dropout + residual add + Apex FusedLayerNorm(eps=1e-12) of
// BertSelfOutput / BertOutput (model/layer.py:113-114,154-155) and their
// autograd, and the bias-gradient reductions of every nn.Linear.
//
// HBM-bound row kernels: one 64-lane wave owns one row, 16-byte loads, fp32
// statistics with a two-pass variance held in registers, wave-shuffle reduces.
// Column reductions (dgamma, dbeta, bias grads) are two-stage and deterministic:
// per-workgroup partial rows in a workspace, then a finalize kernel.
#include <stdlib.h>
#include "common.h"
#include "philox.h"
#include "rowops.h"

namespace {

// The row passes are HBM-bound streams with a few hundred vector instructions per row.  Inside the training step they run
// beside the other stream's fp32 GEMMs, whose v_mfma_f32_32x32x2_f32 occupy the SIMD's vector pipe for 64 cycles each (fp32
// matrix and vector instructions share one pipe on gfx950: tests/tools/mfma_valu_coissue.hip): at equal priority every
// vector instruction of a row pass queues behind them (the pass took 45-65 us beside a weight-gradient GEMM against 13 alone
// and 23 beside a full-rate HBM copy, tests/tools/ln_beside_lab.py).  Raised wave priority lets its few instructions go first.
#ifndef UNITER_LN_PRIO
#define UNITER_LN_PRIO 3
#endif
#define LN_SETPRIO() __builtin_amdgcn_s_setprio(UNITER_LN_PRIO)

template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x,
                                                     const float* __restrict__ res,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta,
                                                     float* __restrict__ z_out, float* __restrict__ y,
                                                     float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, int M, int H,
                                                     DropCfg drop, unsigned short* __restrict__ y_b16,
                                                     int nslab, size_t slab_stride, int pieces) {
  LN_SETPRIO();
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int H4 = H >> 2;
  f32x4 v[NV];
  row_load<NV>(v, x + (size_t)row * H, H4, lane);
  for (int s = 1; s < nslab; ++s) {       // split-K partial sums of the producing GEMM (gemm_bf16_dma.hip)
    f32x4 r[NV];
    row_load<NV>(r, x + s * slab_stride + (size_t)row * H, H4, lane);
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] += r[k];
  }
  if (drop.active) row_dropout<NV>(v, drop, (uint64_t)row * H4, H4, lane);
  if (res) {
    f32x4 r[NV];
    row_load<NV>(r, res + (size_t)row * H, H4, lane);
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] += r[k];
  }
  if (z_out) row_store<NV>(v, z_out + (size_t)row * H, H4, lane);
  float mean, rstd;
  row_stats<NV>(v, H, H4, lane, mean, rstd);
  if (mean_out && lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
  row_affine<NV>(v, gamma, beta, mean, rstd, H4, lane);
  row_store<NV>(v, y + (size_t)row * H, H4, lane);
  if (y_b16) {      // operand copy for the next GEMM: bf16, or the three bf16 pieces [M][3][H] of the fp32-accurate products
    if (pieces == 3) row_store_x3<NV>(v, y_b16 + (size_t)row * 3 * H, H, H4, lane);
    else row_store_bf16<NV>(v, y_b16 + (size_t)row * H, H4, lane);
  }
}

// grid: nblk workgroups of LNB_WAVES waves; wave w of block b walks rows b*W+w, +W*nblk, ...  Eight waves and (up to
// 4096 rows) ONE row per wave: the pass is a latency-bound stream -- with two rows per wave on 4-wave workgroups only
// 1312 waves (1.3 per SIMD) had loads in flight and it ran at 1.2 TB/s; the column partials stay one row per workgroup.
template <int NV, int LNB_WAVES>
__global__ __launch_bounds__(64 * LNB_WAVES) void ln_bwd_kernel(const float* __restrict__ dy,
                                                     const float* __restrict__ z,
                                                     const float* __restrict__ mean,
                                                     const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma,
                                                     float* __restrict__ dz, float* __restrict__ dx,
                                                     float* __restrict__ part, int M, int H,
                                                     DropCfg drop, int want_dbias,
                                                     unsigned short* __restrict__ dx_b16, int nslab,
                                                     size_t slab_stride, int pieces) {
  __shared__ __attribute__((aligned(16))) float red[LNB_WAVES * NV * 256];
  LN_SETPRIO();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int H4 = H >> 2;
  f32x4 g[NV], dg[NV], db[NV], dbx[NV];     // dbx: column sum of dx = bias gradient of the producing Linear
  row_load<NV>(g, gamma, H4, lane);
#pragma unroll
  for (int k = 0; k < NV; ++k) { dg[k] = f32x4{0, 0, 0, 0}; db[k] = f32x4{0, 0, 0, 0}; dbx[k] = f32x4{0, 0, 0, 0}; }
  for (int row = blockIdx.x * LNB_WAVES + wave; row < M; row += gridDim.x * LNB_WAVES) {
    f32x4 d[NV], xh[NV];
    row_load<NV>(d, dy + (size_t)row * H, H4, lane);
    for (int s = 1; s < nslab; ++s) {
      row_load<NV>(xh, dy + s * slab_stride + (size_t)row * H, H4, lane);
#pragma unroll
      for (int k = 0; k < NV; ++k) d[k] += xh[k];
    }
    row_load<NV>(xh, z + (size_t)row * H, H4, lane);
    const float mu = mean[row], rs = rstd[row];
    row_ln_bwd<NV>(d, xh, g, mu, rs, dg, db, H, H4, lane);      // d <- dz, accumulates dg/db
    if (dz) row_store<NV>(d, dz + (size_t)row * H, H4, lane);
    if (drop.active) row_dropout<NV>(d, drop, (uint64_t)row * H4, H4, lane);
    if (dx && (dx != dz || drop.active)) row_store<NV>(d, dx + (size_t)row * H, H4, lane);
    if (dx_b16) {
      if (pieces == 3) row_store_x3<NV>(d, dx_b16 + (size_t)row * 3 * H, H, H4, lane);
      else row_store_bf16<NV>(d, dx_b16 + (size_t)row * H, H4, lane);
    }
    if (want_dbias) {
#pragma unroll
      for (int k = 0; k < NV; ++k) if (lane + 64 * k < H4) dbx[k] += d[k];
    }
  }
  // cross-wave reduce of the column partials, then one partial row per workgroup
  block_col_reduce_store<NV, LNB_WAVES>(dg, red, part + (size_t)blockIdx.x * 3 * H, H4, lane, wave);
  block_col_reduce_store<NV, LNB_WAVES>(db, red, part + (size_t)blockIdx.x * 3 * H + H, H4, lane, wave);
  if (want_dbias) block_col_reduce_store<NV, LNB_WAVES>(dbx, red, part + (size_t)blockIdx.x * 3 * H + 2 * H, H4, lane, wave);
}

// out[n] (+)= sum_p part[p*stride + n].  Block = 32 columns x 8 partial-slices: the slices
// stride through the partial rows in parallel and meet in LDS (a single thread per column
// walking 512 partial rows serially cost 80 us; this shape costs a few).
__global__ __launch_bounds__(256) void finalize_partials_kernel(const float* __restrict__ part, int nparts,
                                                                size_t stride, float* __restrict__ out, int N,
                                                                int beta) {
  __shared__ float red[8][33];
  const int cx = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int n = blockIdx.x * 32 + cx;
  float s = 0.f;
  if (n < N)
    for (int p = sl; p < nparts; p += 8) s += part[(size_t)p * stride + n];
  red[sl][cx] = s;
  __syncthreads();
  if (sl == 0 && n < N) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][cx];
    out[n] = beta ? out[n] + t : t;
  }
}

// several outputs of H columns each out of one partial buffer whose rows are [nout][H]:
// outs[j][c] += sum_p part[p*stride + j*H + c]   (one launch instead of one per output)
struct MultiOut { float* out[8]; };
// block = 16 columns x 16 partial-slices.  256-thread blocks: a 1024-thread block needs a CU with 16
// free wave slots, which the persistent GEMMs of the other stream rarely leave (24 us in situ for a
// few microseconds of work, on the critical dgrad chain).
__global__ __launch_bounds__(256) void finalize_multi_kernel(const float* __restrict__ part, int nparts,
                                                             size_t stride, MultiOut outs, int nout, int H) {
  __shared__ float red[16][17];
  const int cx = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int n = blockIdx.x * 16 + cx, N = nout * H;
  float s = 0.f;
  if (n < N) {
    // four independent loads in flight per thread (a dependent chain of ~10 cost 28 us)
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int p = sl;
    for (; p + 48 < nparts; p += 64) {
      a0 += part[(size_t)p * stride + n];
      a1 += part[(size_t)(p + 16) * stride + n];
      a2 += part[(size_t)(p + 32) * stride + n];
      a3 += part[(size_t)(p + 48) * stride + n];
    }
    for (; p < nparts; p += 16) a0 += part[(size_t)p * stride + n];
    s = (a0 + a1) + (a2 + a3);
  }
  red[sl][cx] = s;
  __syncthreads();
  if (sl == 0 && n < N) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][cx];
    float* o = outs.out[n / H];
    if (o) o[n % H] += t;
  }
}

// several independent column reductions in ONE launch (per encoder layer: the two LayerNorm backward passes'
// [dgamma | dbeta | dbias] partial rows, the attention backward's per-sample query|key|value bias partials and, in the
// fp32 mode, the dU column partials): out[c] += sum_p part[p * stride + c] for c < n.  Same 16 x 16 block shape as
// finalize_multi_kernel; blocks are dealt to jobs by their prefix of block counts.
struct FinJobs {
  const float* part[4]; int nparts[4]; size_t stride[4]; int n[4];
  float* out[4][3]; int seg[4];        // output j of a job covers columns [j * seg, (j + 1) * seg)
  int first_block[5];
};
__global__ __launch_bounds__(256) void finalize_jobs_kernel(const FinJobs J, int njobs) {
  __shared__ float red[16][17];
  int j = 0;
  while (j + 1 < njobs && (int)blockIdx.x >= J.first_block[j + 1]) ++j;
  const int cx = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int n = (blockIdx.x - J.first_block[j]) * 16 + cx, N = J.n[j], nparts = J.nparts[j];
  const float* __restrict__ part = J.part[j];
  const size_t stride = J.stride[j];
  float s = 0.f;
  if (n < N) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int p = sl;
    for (; p + 48 < nparts; p += 64) {
      a0 += part[(size_t)p * stride + n];
      a1 += part[(size_t)(p + 16) * stride + n];
      a2 += part[(size_t)(p + 32) * stride + n];
      a3 += part[(size_t)(p + 48) * stride + n];
    }
    for (; p < nparts; p += 16) a0 += part[(size_t)p * stride + n];
    s = (a0 + a1) + (a2 + a3);
  }
  red[sl][cx] = s;
  __syncthreads();
  if (sl == 0 && n < N) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][cx];
    float* o = J.out[j][n / J.seg[j]];
    if (o) o[n % J.seg[j]] += t;
  }
}

// column sums: block (bx, by) covers columns [bx*256, bx*256+256) and rows by, by+gridDim.y, ...
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int M, int N, int ld,
                                                     float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) float red[4 * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + lane * 4;
  f32x4 s = {0, 0, 0, 0};
  if (c < N) {
    for (int r = blockIdx.y * 4 + wave; r < M; r += gridDim.y * 4)
      s += *reinterpret_cast<const f32x4*>(X + (size_t)r * ld + c);
  }
  *reinterpret_cast<f32x4*>(red + wave * 256 + lane * 4) = s;
  __syncthreads();
  if (wave == 0 && c < N) {
    f32x4 t = s;
#pragma unroll
    for (int w = 1; w < 4; ++w) t += *reinterpret_cast<const f32x4*>(red + w * 256 + lane * 4);
    *reinterpret_cast<f32x4*>(part + (size_t)blockIdx.y * N + c) = t;
  }
}

// column sums of a bf16 matrix, added into out (fp32): block (bx, by) covers 512 columns (64 lanes x 8) and the rows
// by*RPB .. +RPB; four row-lanes meet in LDS, then one fp32 atomic per column and block (the bias gradient of
// intermediate.dense from the bf16 dU: 16 MB read once, on the side stream)
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const unsigned short* __restrict__ X, int M, int N, int ld,
                                                          float* __restrict__ out, int rows_per_block) {
  __shared__ float red[4][64 * 8 + 8];
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 512 + lane * 8;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c < N) {
    const int r1 = min(M, (int)(blockIdx.y + 1) * rows_per_block);
    for (int r = blockIdx.y * rows_per_block + wave; r < r1; r += 4) {
      const u32x4_t v = *reinterpret_cast<const u32x4_t*>(X + (size_t)r * ld + c);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        s[2 * k] += __builtin_bit_cast(float, v[k] << 16);
        s[2 * k + 1] += __builtin_bit_cast(float, v[k] & 0xffff0000u);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[wave][lane * 8 + k] = s[k];
  __syncthreads();
  for (int j = threadIdx.x; j < 512; j += 256) {
    const int col = blockIdx.x * 512 + j;
    if (col < N) atomicAdd(out + col, (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]));
  }
}

// out[i] += sum_s slab[s * stride + i]: the k-pieces of a split-K weight-gradient GEMM (fp32 partial tiles written with
// plain 16-byte stores) folded into the gradient buffer -- one streaming pass instead of float atomics at 1.3 TB/s
__global__ __launch_bounds__(256) void slab_reduce_add_kernel(const float* __restrict__ slabs, int nslab, size_t stride4,
                                                             float* __restrict__ out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    f32x4 a = reinterpret_cast<const f32x4*>(slabs)[i];
    for (int s = 1; s < nslab; ++s) a += reinterpret_cast<const f32x4*>(slabs)[s * stride4 + i];
    reinterpret_cast<f32x4*>(out)[i] += a;
  }
}

__global__ void add_kernel(float* __restrict__ out, const float* __restrict__ a, const float* __restrict__ b, size_t n4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) reinterpret_cast<f32x4*>(out)[i] = reinterpret_cast<const f32x4*>(a)[i] + reinterpret_cast<const f32x4*>(b)[i];
}

// waves per workgroup / rows per wave of the backward row pass (A/B switches: UNITER_LNB_WAVES = 4 | 8, UNITER_LNB_ROWS)
inline int lnb_waves() { static const int w = [] { const char* e = getenv("UNITER_LNB_WAVES"); return (e && atoi(e) == 8) ? 8 : 4; }(); return w; }
inline int lnb_rows() { static const int r = [] { const char* e = getenv("UNITER_LNB_ROWS"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : v; }(); return r; }
inline int ln_bwd_blocks(int M) {
  const int per = lnb_waves() * lnb_rows();
  int b = (M + per - 1) / per;
  return b < 1024 ? (b < 1 ? 1 : b) : 1024;
}
inline int colsum_splits(int M) {
  int s = (M + 31) / 32;
  return s < 64 ? (s < 1 ? 1 : s) : 64;
}

}  // namespace

// out = a + b (n multiple of 4)
int launch_add_f32(float* out, const float* a, const float* b, size_t n, hipStream_t st) {
  const size_t n4 = n / 4;
  if (n4 == 0) return 0;
  hipLaunchKernelGGL(add_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, out, a, b, n4);
  UCHECK_LAUNCH();
  return 0;
}

// accumulate nout (<= 8) outputs of H columns each; NULL outputs are skipped
int finalize_partials_multi(const float* part, int nparts, size_t stride, float* const* outs, int nout, int H,
                            hipStream_t st) {
  MultiOut mo = {};
  for (int j = 0; j < nout && j < 8; ++j) mo.out[j] = outs[j];
  hipLaunchKernelGGL(finalize_multi_kernel, dim3((nout * H + 15) / 16), dim3(256), 0, st, part, nparts, stride,
                     mo, nout, H);
  UCHECK_LAUNCH();
  return 0;
}

// Up to 4 jobs: job i accumulates nout[i] (<= 3) outputs of seg[i] columns each out of part[i] (nparts[i] rows of
// `stride[i]` floats).  NULL outputs are skipped; jobs with part == NULL are dropped.
int finalize_partials_jobs(int njobs, const float* const* part, const int* nparts, const size_t* stride,
                           float* const (*outs)[3], const int* nout, const int* seg, hipStream_t st) {
  FinJobs J = {};
  int k = 0, blocks = 0;
  for (int i = 0; i < njobs && k < 4; ++i) {
    if (!part[i] || nparts[i] <= 0) continue;
    J.part[k] = part[i]; J.nparts[k] = nparts[i]; J.stride[k] = stride[i]; J.seg[k] = seg[i]; J.n[k] = nout[i] * seg[i];
    for (int o = 0; o < 3; ++o) J.out[k][o] = o < nout[i] ? outs[i][o] : nullptr;
    J.first_block[k] = blocks;
    blocks += (J.n[k] + 15) / 16;
    ++k;
  }
  J.first_block[k] = blocks;
  if (k == 0) return 0;
  hipLaunchKernelGGL(finalize_jobs_kernel, dim3(blocks), dim3(256), 0, st, J, k);
  UCHECK_LAUNCH();
  return 0;
}

int finalize_partials(const float* part, int nparts, size_t stride, float* out, int N, int beta,
                      hipStream_t st) {
  hipLaunchKernelGGL(finalize_partials_kernel, dim3((N + 31) / 32), dim3(256), 0, st, part, nparts,
                     stride, out, N, beta);
  UCHECK_LAUNCH();
  return 0;
}

extern "C" size_t uniter_colsum_ws_bytes(int M, int N) {
  return (size_t)colsum_splits(M) * N * sizeof(float);
}

extern "C" int uniter_colsum_f32(const float* X, int M, int N, int ld, float* out, int beta, void* ws,
                                 size_t ws_bytes, void* stream) {
  UCHECK_ARG(X && out && ws, "colsum: null pointer");
  UCHECK_SHAPE(N % 4 == 0 && ld % 4 == 0, "colsum: N and ld must be multiples of 4");
  UCHECK_ARG(ws_bytes >= uniter_colsum_ws_bytes(M, N), "colsum: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int splits = colsum_splits(M);
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 255) / 256, splits), dim3(256), 0, st, X, M, N, ld,
                     (float*)ws);
  UCHECK_LAUNCH();
  return finalize_partials((const float*)ws, splits, (size_t)N, out, N, beta, st);
}

extern "C" int uniter_slab_reduce_add(const float* slabs, int nslab, size_t slab_stride, float* out, size_t n, void* stream) {
  UCHECK_ARG(slabs && out && nslab >= 1, "slab_reduce_add: bad argument");
  UCHECK_SHAPE(n % 4 == 0 && slab_stride % 4 == 0 && ((uintptr_t)slabs & 15) == 0 && ((uintptr_t)out & 15) == 0,
               "slab_reduce_add: n and the slab stride must be multiples of 4, buffers 16-byte aligned");
  if (n == 0) return 0;
  const size_t n4 = n / 4;
  const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(slab_reduce_add_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, slabs, nslab, slab_stride / 4, out, n4);
  UCHECK_LAUNCH();
  return 0;
}

// out[n] += sum_m X[m, n] for a bf16 X (N % 8 == 0, ld % 8 == 0, 16-byte aligned)
extern "C" int uniter_colsum_bf16_add(const void* X, int M, int N, int ld, float* out, void* stream) {
  UCHECK_ARG(X && out && M > 0 && N > 0, "colsum_bf16: bad argument");
  UCHECK_SHAPE(N % 8 == 0 && ld % 8 == 0 && ((uintptr_t)X & 15) == 0, "colsum_bf16: N, ld multiples of 8 and a 16-byte aligned operand required");
  const int rpb = 64;
  hipLaunchKernelGGL(colsum_bf16_kernel, dim3((N + 511) / 512, (M + rpb - 1) / rpb), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)X, M, N, ld, out, rpb);
  UCHECK_LAUNCH();
  return 0;
}

#define LN_DISPATCH(NVv, KERNEL, GRID, ...) LN_DISPATCH_T(NVv, KERNEL, GRID, 256, __VA_ARGS__)
#define LN_DISPATCH_T(NVv, KERNEL, GRID, THREADS, ...)                                           \
  switch (NVv) {                                                                                 \
    case 1: hipLaunchKernelGGL((KERNEL<1>), GRID, dim3(THREADS), 0, st, __VA_ARGS__); break;     \
    case 2: hipLaunchKernelGGL((KERNEL<2>), GRID, dim3(THREADS), 0, st, __VA_ARGS__); break;     \
    case 3: hipLaunchKernelGGL((KERNEL<3>), GRID, dim3(THREADS), 0, st, __VA_ARGS__); break;     \
    case 4: hipLaunchKernelGGL((KERNEL<4>), GRID, dim3(THREADS), 0, st, __VA_ARGS__); break;     \
    default: uniter_set_error("layernorm: hidden size %d unsupported (max 1024)", H);            \
             return UNITER_E_SHAPE;                                                              \
  }

extern "C" int uniter_ln_fwd(const float* x, const float* res, const float* gamma, const float* beta,
                             float* z_out, float* y, float* mean, float* rstd, int M, int H,
                             float p_drop, uint64_t seed, uint32_t offset, uint32_t site,
                             void* stream) {
  return uniter_ln_fwd_b16(x, res, gamma, beta, z_out, y, nullptr, mean, rstd, M, H, p_drop, seed, offset, site, stream);
}

extern "C" int uniter_ln_fwd_b16(const float* x, const float* res, const float* gamma, const float* beta,
                                 float* z_out, float* y, void* y_bf16, float* mean, float* rstd, int M, int H,
                                 float p_drop, uint64_t seed, uint32_t offset, uint32_t site, void* stream) {
  return uniter_ln_fwd_slabs(x, 1, 0, res, gamma, beta, z_out, y, y_bf16, mean, rstd, M, H, p_drop, seed, offset, site, stream);
}

static int ln_fwd_run(const float* x, int nslab, size_t slab_stride, const float* res, const float* gamma,
                      const float* beta, float* z_out, float* y, void* y_bf16, int pieces, float* mean, float* rstd,
                      int M, int H, float p_drop, uint64_t seed, uint32_t offset, uint32_t site, void* stream) {
  const unsigned char* ahead = take_drop_bits();      // (first thing: a call that fails below must not leave the pointer to the next one)
  UCHECK_ARG(x && gamma && beta && y, "ln_fwd: null pointer");
  UCHECK_ARG(nslab >= 1 && (nslab == 1 || slab_stride >= (size_t)M * H), "ln_fwd: bad slab count / stride");
  UCHECK_ARG((mean == nullptr) == (rstd == nullptr), "ln_fwd: mean/rstd must both be given or NULL");
  UCHECK_SHAPE(H % 4 == 0 && H >= 4, "ln_fwd: H must be a multiple of 4");
  UCHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "ln_fwd: bad dropout p");
  if (M <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  DropCfg drop = make_drop(p_drop, seed, offset, site);
  drop.bits = drop.active ? ahead : nullptr;
  const int nv = (H / 4 + 63) / 64;
  LN_DISPATCH(nv, ln_fwd_kernel, dim3((M + 3) / 4), x, res, gamma, beta, z_out, y, mean, rstd, M, H, drop,
              (unsigned short*)y_bf16, nslab, slab_stride, pieces);
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_ln_fwd_slabs(const float* x, int nslab, size_t slab_stride, const float* res, const float* gamma,
                                   const float* beta, float* z_out, float* y, void* y_bf16, float* mean, float* rstd,
                                   int M, int H, float p_drop, uint64_t seed, uint32_t offset, uint32_t site,
                                   void* stream) {
  return ln_fwd_run(x, nslab, slab_stride, res, gamma, beta, z_out, y, y_bf16, 1, mean, rstd, M, H, p_drop, seed, offset, site, stream);
}

extern "C" int uniter_ln_fwd_slabs_x3(const float* x, int nslab, size_t slab_stride, const float* res, const float* gamma,
                                      const float* beta, float* z_out, float* y, void* y_x3, float* mean, float* rstd,
                                      int M, int H, float p_drop, uint64_t seed, uint32_t offset, uint32_t site,
                                      void* stream) {
  UCHECK_SHAPE(H % 8 == 0 && ((uintptr_t)y_x3 & 15) == 0, "ln_fwd: the x3 copy needs H %% 8 == 0 and a 16-byte aligned buffer");
  return ln_fwd_run(x, nslab, slab_stride, res, gamma, beta, z_out, y, y_x3, 3, mean, rstd, M, H, p_drop, seed, offset, site, stream);
}

extern "C" size_t uniter_ln_bwd_ws_bytes(int M, int H) {
  return (size_t)ln_bwd_blocks(M) * 3 * H * sizeof(float);
}

extern "C" int uniter_ln_bwd(const float* dy, const float* z, const float* mean, const float* rstd,
                             const float* gamma, float* dz, float* dx, float* dgamma, float* dbeta,
                             float* dbias, int M, int H, float p_drop, uint64_t seed, uint32_t offset,
                             uint32_t site, void* ws, size_t ws_bytes, void* stream) {
  return uniter_ln_bwd_b16(dy, z, mean, rstd, gamma, dz, dx, nullptr, dgamma, dbeta, dbias, M, H, p_drop, seed, offset,
                           site, ws, ws_bytes, stream);
}

extern "C" int uniter_ln_bwd_b16(const float* dy, const float* z, const float* mean, const float* rstd,
                                 const float* gamma, float* dz, float* dx, void* dx_bf16, float* dgamma,
                                 float* dbeta, float* dbias, int M, int H, float p_drop, uint64_t seed,
                                 uint32_t offset, uint32_t site, void* ws, size_t ws_bytes, void* stream) {
  UCHECK_ARG(dgamma && dbeta, "ln_bwd: null pointer");
  UCHECK_RC(uniter_ln_bwd_rows(dy, z, mean, rstd, gamma, dz, dx, dx_bf16, dbias != nullptr, M, H, p_drop, seed, offset,
                               site, ws, ws_bytes, stream));
  return uniter_ln_bwd_finalize(ws, ws_bytes, M, H, dgamma, dbeta, dbias, stream);
}

// The two halves of uniter_ln_bwd_b16 as separate calls: the row pass produces everything the
// backward CHAIN needs (dz, dx); the column reduction only feeds parameter gradients and can run
// later on another stream, out of the critical path (the caller keeps `ws` untouched in between).
extern "C" int uniter_ln_bwd_rows(const float* dy, const float* z, const float* mean, const float* rstd,
                                  const float* gamma, float* dz, float* dx, void* dx_bf16, int want_dbias, int M,
                                  int H, float p_drop, uint64_t seed, uint32_t offset, uint32_t site, void* ws,
                                  size_t ws_bytes, void* stream) {
  return uniter_ln_bwd_rows_slabs(dy, 1, 0, z, mean, rstd, gamma, dz, dx, dx_bf16, want_dbias, M, H, p_drop, seed, offset,
                                  site, ws, ws_bytes, stream);
}

static int ln_bwd_rows_run(const float* dy, int nslab, size_t slab_stride, const float* z, const float* mean,
                           const float* rstd, const float* gamma, float* dz, float* dx, void* dx_bf16, int pieces,
                           int want_dbias, int M, int H, float p_drop, uint64_t seed, uint32_t offset,
                           uint32_t site, void* ws, size_t ws_bytes, void* stream) {
  const unsigned char* ahead = take_drop_bits();      // (first thing, as in the forward pass)
  UCHECK_ARG(dy && z && mean && rstd && gamma && ws, "ln_bwd: null pointer");
  UCHECK_ARG(nslab >= 1 && (nslab == 1 || slab_stride >= (size_t)M * H), "ln_bwd: bad slab count / stride");
  UCHECK_ARG(dz || dx || dx_bf16, "ln_bwd: need dz or dx");
  UCHECK_SHAPE(H % 4 == 0 && H >= 4, "ln_bwd: H must be a multiple of 4");
  UCHECK_ARG(ws_bytes >= uniter_ln_bwd_ws_bytes(M, H), "ln_bwd: workspace too small");
  if (M <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  DropCfg drop = make_drop(p_drop, seed, offset, site);
  drop.bits = drop.active ? ahead : nullptr;
  const int nv = (H / 4 + 63) / 64;
  const int nblk = ln_bwd_blocks(M);
  float* part = (float*)ws;
#define LNB_ARGS dy, z, mean, rstd, gamma, dz, dx, part, M, H, drop, want_dbias != 0, (unsigned short*)dx_bf16, nslab, slab_stride, pieces
  if (lnb_waves() == 8) {
    switch (nv) {
      case 1: hipLaunchKernelGGL((ln_bwd_kernel<1, 8>), dim3(nblk), dim3(512), 0, st, LNB_ARGS); break;
      case 2: hipLaunchKernelGGL((ln_bwd_kernel<2, 8>), dim3(nblk), dim3(512), 0, st, LNB_ARGS); break;
      case 3: hipLaunchKernelGGL((ln_bwd_kernel<3, 8>), dim3(nblk), dim3(512), 0, st, LNB_ARGS); break;
      case 4: hipLaunchKernelGGL((ln_bwd_kernel<4, 8>), dim3(nblk), dim3(512), 0, st, LNB_ARGS); break;
      default: uniter_set_error("layernorm: hidden size %d unsupported (max 1024)", H); return UNITER_E_SHAPE;
    }
  } else {
    switch (nv) {
      case 1: hipLaunchKernelGGL((ln_bwd_kernel<1, 4>), dim3(nblk), dim3(256), 0, st, LNB_ARGS); break;
      case 2: hipLaunchKernelGGL((ln_bwd_kernel<2, 4>), dim3(nblk), dim3(256), 0, st, LNB_ARGS); break;
      case 3: hipLaunchKernelGGL((ln_bwd_kernel<3, 4>), dim3(nblk), dim3(256), 0, st, LNB_ARGS); break;
      case 4: hipLaunchKernelGGL((ln_bwd_kernel<4, 4>), dim3(nblk), dim3(256), 0, st, LNB_ARGS); break;
      default: uniter_set_error("layernorm: hidden size %d unsupported (max 1024)", H); return UNITER_E_SHAPE;
    }
  }
#undef LNB_ARGS
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_ln_bwd_rows_slabs(const float* dy, int nslab, size_t slab_stride, const float* z, const float* mean,
                                        const float* rstd, const float* gamma, float* dz, float* dx, void* dx_bf16,
                                        int want_dbias, int M, int H, float p_drop, uint64_t seed, uint32_t offset,
                                        uint32_t site, void* ws, size_t ws_bytes, void* stream) {
  return ln_bwd_rows_run(dy, nslab, slab_stride, z, mean, rstd, gamma, dz, dx, dx_bf16, 1, want_dbias, M, H, p_drop, seed,
                         offset, site, ws, ws_bytes, stream);
}

extern "C" int uniter_ln_bwd_rows_slabs_x3(const float* dy, int nslab, size_t slab_stride, const float* z, const float* mean,
                                           const float* rstd, const float* gamma, float* dz, float* dx, void* dx_x3,
                                           int want_dbias, int M, int H, float p_drop, uint64_t seed, uint32_t offset,
                                           uint32_t site, void* ws, size_t ws_bytes, void* stream) {
  UCHECK_SHAPE(H % 8 == 0 && ((uintptr_t)dx_x3 & 15) == 0, "ln_bwd: the x3 copy needs H %% 8 == 0 and a 16-byte aligned buffer");
  return ln_bwd_rows_run(dy, nslab, slab_stride, z, mean, rstd, gamma, dz, dx, dx_x3, 3, want_dbias, M, H, p_drop, seed,
                         offset, site, ws, ws_bytes, stream);
}

// internal: the partial-row count / row stride of a row pass over M rows (for finalize_partials_jobs)
int ln_bwd_partial_rows(int M) { return ln_bwd_blocks(M); }

extern "C" int uniter_ln_bwd_finalize(const void* ws, size_t ws_bytes, int M, int H, float* dgamma, float* dbeta,
                                      float* dbias, void* stream) {
  UCHECK_ARG(ws && dgamma && dbeta && ws_bytes >= uniter_ln_bwd_ws_bytes(M, H), "ln_bwd_finalize: bad argument");
  if (M <= 0) return 0;
  float* outs[3] = {dgamma, dbeta, dbias};
  return finalize_partials_multi((const float*)ws, ln_bwd_blocks(M), (size_t)3 * H, outs, dbias ? 3 : 2, H,
                                 (hipStream_t)stream);
}

// ---- hidden-dropout keep flags drawn ahead (round 5) ----------------------------------------------------------------------
// The row passes draw their dropout flags with ten Philox rounds per 4-element group (1.5 us of a 9-us forward pass, 1.8 of an
// 11-us backward pass, twice each per layer): a function of (seed, offset, site, index) alone, so all of a step's sites are drawn in ONE
// launch on the auxiliary stream beside the head of the forward pass, as nibbles -- bits[s][g >> 1] >> 4 (g & 1) = the keep flags
// of group g of site s -- and the passes read 3 bytes per lane and row instead (DropCfg::bits, through g_uniter_drop_bits).
// Same flags bit for bit (tests/test_layernorm_gpu.py::test_keep_flags_drawn_ahead_are_the_row_passes_own).
__global__ __launch_bounds__(256) void hidden_keep_bits_kernel(unsigned* __restrict__ bits, size_t site_words, int nsites, uint32_t site_a0,
                                                               uint32_t site_b0, uint32_t site_step, size_t words, DropCfg d) {
  const size_t w = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int s = blockIdx.y;
  if (w >= words || s >= nsites) return;
  d.site = ((s & 1) ? site_b0 : site_a0) + (uint32_t)(s >> 1) * site_step;
  unsigned word = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) word |= drop_bits4(d, (uint64_t)w * 8 + j) << (4 * j);
  bits[(size_t)s * site_words + w] = word;
}

extern "C" size_t uniter_hidden_keep_bits_bytes(size_t elements) { return ((elements / 4 + 7) / 8) * 4; }

// sites site_a0 + k * site_step (even s = 2 k) and site_b0 + k * site_step (odd s = 2 k + 1), `elements` values each
extern "C" int uniter_hidden_keep_bits_gen(void* bits, size_t site_stride_bytes, int nsites, uint32_t site_a0, uint32_t site_b0,
                                           uint32_t site_step, size_t elements, float p_drop, uint64_t seed, uint32_t offset,
                                           void* stream) {
  UCHECK_ARG(bits && nsites >= 1 && elements > 0 && p_drop > 0.f && p_drop < 1.f, "hidden_keep_bits_gen: bad argument");
  const size_t words = (elements / 4 + 7) / 8;
  UCHECK_SHAPE(elements % 4 == 0 && site_stride_bytes % 4 == 0 && site_stride_bytes >= words * 4 && ((uintptr_t)bits & 3) == 0,
               "hidden_keep_bits_gen: elements %% 4, a 4-byte aligned buffer and site stride >= uniter_hidden_keep_bits_bytes");
  const DropCfg d = make_drop(p_drop, seed, offset, 0);
  hipLaunchKernelGGL(hidden_keep_bits_kernel, dim3((unsigned)((words + 255) / 256), nsites), dim3(256), 0, (hipStream_t)stream,
                     (unsigned*)bits, site_stride_bytes / 4, nsites, site_a0, site_b0, site_step, words, d);
  UCHECK_LAUNCH();
  return 0;
}

// the keep flags the NEXT LayerNorm row pass of this host thread reads instead of drawing them (NULL = draw); tests and callers
// that drive the row passes themselves
extern "C" int uniter_ln_set_next_keep_bits(const void* site_bits) {
  g_uniter_drop_bits = (const unsigned char*)site_bits;
  return 0;
}
