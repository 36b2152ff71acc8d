// fp32-accurate GEMM from PRE-SPLIT operands ("planes"): the split of csrc/gemm_split.hip
//     x = x1 + x2 + x3  (three bf16 pieces, exact),   a*b ~= six bf16 x bf16 products
// done ONCE per tensor instead of once per output tile.  A tensor's planes are three bf16 arrays
// [3][rows][ld] (plane p at + p * plane_stride elements).  Weights are split once per optimizer
// step (plus a transposed copy for dgrad, so that every weight operand is k-contiguous);
// activations are split by their producers.  The consumer kernel below does no arithmetic
// besides MFMAs: bf16 tiles go global -> registers -> LDS unchanged, and the loop is bound by the
// bf16 matrix pipe (6 x 32x32x16 MFMAs per 32x32x16 block = 417 TFLOP/s of fp32-equivalent work at
// pipe peak) instead of the 157 TFLOP/s fp32 pipe.
//
// C = epi(A @ B^T):  A planes [3][M][lda], B planes [3][N][ldb], both k-contiguous.
#include <stdlib.h>
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned short bf16_t;

constexpr int BKP = 32;
constexpr int LDP = 40;          // LDS row stride (bf16 elements): 80 B, conflict-free ds_read_b128

__device__ __forceinline__ void split1(float x, unsigned& h1, unsigned& h2, unsigned& h3) {
  const unsigned ux = __builtin_bit_cast(unsigned, x);
  h1 = ux & 0xffff0000u;
  const float r = x - __builtin_bit_cast(float, h1);
  h2 = __builtin_bit_cast(unsigned, r) & 0xffff0000u;
  const float s = r - __builtin_bit_cast(float, h2);
  h3 = __builtin_bit_cast(unsigned, s) & 0xffff0000u;
}

// planes[p][r][c] = piece p of x[r][c]; 4 consecutive columns per thread
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, int rows, int cols, int ld,
                                                           bf16_t* __restrict__ planes, int pld, size_t pstride) {
  const int c4 = blockIdx.x * 64 + (threadIdx.x & 63);
  const int r = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (r >= rows || c4 * 4 >= cols) return;
  const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)r * ld + c4 * 4);
  unsigned a[4][3];
#pragma unroll
  for (int e = 0; e < 4; ++e) split1(v[e], a[e][0], a[e][1], a[e][2]);
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    u32x2_t o = {(a[0][p] >> 16) | a[1][p], (a[2][p] >> 16) | a[3][p]};
    *reinterpret_cast<u32x2_t*>(planes + p * pstride + (size_t)r * pld + c4 * 4) = o;
  }
}

// transposed: planes[p][c][r] = piece p of x[r][c]  (64 x 64 tiles through LDS)
__global__ __launch_bounds__(256) void split_planes_t_kernel(const float* __restrict__ x, int rows, int cols, int ld,
                                                             bf16_t* __restrict__ planes, int pld, size_t pstride) {
  __shared__ float t[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int k = threadIdx.x; k < 64 * 16; k += 256) {
    const int r = k >> 4, c4 = k & 15;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r0 + r < rows && c0 + c4 * 4 < cols) v = *reinterpret_cast<const f32x4*>(x + (size_t)(r0 + r) * ld + c0 + c4 * 4);
    t[r][c4 * 4 + 0] = v[0]; t[r][c4 * 4 + 1] = v[1]; t[r][c4 * 4 + 2] = v[2]; t[r][c4 * 4 + 3] = v[3];
  }
  __syncthreads();
  for (int k = threadIdx.x; k < 64 * 16; k += 256) {
    const int c = k >> 4, r4 = k & 15;              // output row = source column c, 4 consecutive source rows
    if (c0 + c >= cols || r0 + r4 * 4 >= rows) continue;
    unsigned a[4][3];
#pragma unroll
    for (int e = 0; e < 4; ++e) split1(t[r4 * 4 + e][c], a[e][0], a[e][1], a[e][2]);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      u32x2_t o = {(a[0][p] >> 16) | a[1][p], (a[2][p] >> 16) | a[3][p]};
      *reinterpret_cast<u32x2_t*>(planes + p * pstride + (size_t)(c0 + c) * pld + r0 + r4 * 4) = o;
    }
  }
}

struct GemmArgsP {
  int M, N, K;
  const bf16_t* A; int lda; size_t psa;
  const bf16_t* B; int ldb; size_t psb;
  float* C; int ldc;
  bf16_t* Cp; int ldcp; size_t psc;     // optional planes of the stored C
  int epi;
  const float* bias;
  const float* aux_in;
  float* aux_out;
  int ld_aux;
  int beta;
  int tiles_m, tiles_n, band_h;
  float* colsum_part;
};

__device__ __forceinline__ void tile_coords_p(int t, int tiles_m, int tiles_n, int band_h, int& tm, int& tn) {
  const int full = band_h * tiles_n;
  const int band = t / full;
  const int rem = t - band * full;
  const int bh = min(band_h, tiles_m - band * band_h);
  tn = rem / bh;
  tm = band * band_h + (rem - tn * bh);
}

// 16-B loads of one operand tile: R rows x 32 k x 3 planes; 4 threads per row, 64 rows per pass
template <int R>
__device__ __forceinline__ void tile_load_p(u32x4_t (&reg)[3 * R / 64], __amdgpu_buffer_rsrc_t rsrc, int ld, size_t ps,
                                            int row0, int koff_bytes, int tid) {
  const int c = tid & 3, rr = tid >> 2;
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int q = 0; q < R / 64; ++q) {
      const int off = (int)(p * ps * 2) + ((row0 + rr + 64 * q) * ld + c * 8) * 2;
      reg[p * (R / 64) + q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, koff_bytes, 0);
    }
}
template <int R>
__device__ __forceinline__ void tile_store_p(const u32x4_t (&reg)[3 * R / 64], bf16_t* s, int tid) {
  const int c = tid & 3, rr = tid >> 2;
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int q = 0; q < R / 64; ++q)
      *reinterpret_cast<u32x4_t*>(s + p * R * LDP + (rr + 64 * q) * LDP + c * 8) = reg[p * (R / 64) + q];
}
__device__ __forceinline__ bf16x8 frag_read_p(const bf16_t* s, int r0, int ks, int i, int h) {
  return *reinterpret_cast<const bf16x8*>(s + (r0 + i) * LDP + ks * 16 + 8 * h);
}

template <int BM, int BN, int WPS>
__global__ __launch_bounds__(256, WPS) void gemm_planes_kernel(const GemmArgsP g) {
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  constexpr int PA = BM * LDP, PB = BN * LDP;
  constexpr int STAGE = 3 * (PA + PB);
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntiles = g.tiles_m * g.tiles_n;
  const int nk = g.K / BKP;

  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  const int q = ntiles >> 3, r = ntiles & 7;
  const int chunk0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int chunk_n = q + (xcd < r ? 1 : 0);
  if (idx >= chunk_n) return;
  const int my_tiles = (chunk_n - idx + per_xcd - 1) / per_xcd;
  const int total_units = my_tiles * nk;

  // one descriptor per operand covering all three planes
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(g.A), 0, (int)((2 * g.psa + (size_t)g.M * g.lda) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(g.B), 0, (int)((2 * g.psb + (size_t)g.N * g.ldb) * 2), 0x00020000);

  int lt = idx, lk = 0, lm0, ln0;
  {
    int tmi, tni;
    tile_coords_p(chunk0 + lt, g.tiles_m, g.tiles_n, g.band_h, tmi, tni);
    lm0 = tmi * BM; ln0 = tni * BN;
  }
  int loaded = 0;
#define LOAD_UNIT(RA, RB)                                                                \
  do {                                                                                   \
    if (loaded < total_units) {                                                          \
      tile_load_p<BM>(RA, rsA, g.lda, g.psa, lm0, lk * BKP * 2, tid);                    \
      tile_load_p<BN>(RB, rsB, g.ldb, g.psb, ln0, lk * BKP * 2, tid);                    \
      ++loaded;                                                                          \
      if (++lk == nk) {                                                                  \
        lk = 0; lt += per_xcd;                                                           \
        if (loaded < total_units) {                                                      \
          int tmi_, tni_;                                                                \
          tile_coords_p(chunk0 + lt, g.tiles_m, g.tiles_n, g.band_h, tmi_, tni_);        \
          lm0 = tmi_ * BM; ln0 = tni_ * BN;                                              \
        }                                                                                \
      }                                                                                  \
    }                                                                                    \
  } while (0)

  u32x4_t ra[3 * BM / 64], rb[3 * BN / 64];
  bf16x8 fa0[TM][3], fb0[TN][3], fa1[TM][3], fb1[TN][3];

  LOAD_UNIT(ra, rb);
  tile_store_p<BM>(ra, smem, tid);
  tile_store_p<BN>(rb, smem + 3 * PA, tid);
  LOAD_UNIT(ra, rb);
  __syncthreads();

#define READ_FRAGS(FA, FB, ST, KS)                                                                      \
  _Pragma("unroll") for (int a = 0; a < TM; ++a)                                                        \
  _Pragma("unroll") for (int p = 0; p < 3; ++p)                                                         \
      FA[a][p] = frag_read_p((ST) + p * PA, wm * WM + a * 32, KS, i, h);                                \
  _Pragma("unroll") for (int b = 0; b < TN; ++b)                                                        \
  _Pragma("unroll") for (int p = 0; p < 3; ++p)                                                         \
      FB[b][p] = frag_read_p((ST) + 3 * PA + p * PB, wn * WN + b * 32, KS, i, h);
// product index outermost: consecutive MFMAs go to DIFFERENT accumulators (a dependent chain of six
// on one accumulator does not issue back to back)
#define MFMA_ONE(FA, FB, PA_, PB_)                                                                      \
  _Pragma("unroll") for (int a = 0; a < TM; ++a)                                                        \
  _Pragma("unroll") for (int b = 0; b < TN; ++b)                                                        \
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[a][PA_], FB[b][PB_], acc[a][b], 0, 0, 0);
#define MFMA_BLOCK(FA, FB)                                                                              \
  MFMA_ONE(FA, FB, 0, 2) MFMA_ONE(FA, FB, 1, 1) MFMA_ONE(FA, FB, 2, 0)                                  \
  MFMA_ONE(FA, FB, 0, 1) MFMA_ONE(FA, FB, 1, 0) MFMA_ONE(FA, FB, 0, 0)

  READ_FRAGS(fa0, fb0, smem, 0)

  int ct = idx, ck = 0;
  int tmi0, tni0;
  tile_coords_p(chunk0 + ct, g.tiles_m, g.tiles_n, g.band_h, tmi0, tni0);
  int m0 = tmi0 * BM, n0 = tni0 * BN;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) acc[a][b][rr] = 0.f;

  for (int u = 0; u < total_units; ++u) {
    const bf16_t* sS = smem + (u & 1) * STAGE;
    bf16_t* dS = smem + ((u + 1) & 1) * STAGE;
    const bool more = u + 1 < total_units;
    __builtin_amdgcn_sched_barrier(0);
    READ_FRAGS(fa1, fb1, sS, 1)
    MFMA_BLOCK(fa0, fb0)
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
      tile_store_p<BM>(ra, dS, tid);
      tile_store_p<BN>(rb, dS + 3 * PA, tid);
    }
    __builtin_amdgcn_sched_barrier(0);
    LOAD_UNIT(ra, rb);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    if (more) { READ_FRAGS(fa0, fb0, dS, 0) }
    MFMA_BLOCK(fa1, fb1)
    __builtin_amdgcn_sched_barrier(0);
    if (++ck == nk) {
#pragma unroll
      for (int a = 0; a < TM; ++a) {
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          const int col = n0 + wn * WN + b * 32 + i;
          const bool cok = col < g.N;
          const float bv = (cok && (g.epi == UNITER_EPI_BIAS || g.epi == UNITER_EPI_BIAS_GELU)) ? g.bias[col] : 0.f;
          float csum = 0.f;
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) {
            const int row = m0 + wm * WM + a * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
            if (cok && row < g.M) {
              float v = acc[a][b][rr] + bv;
              if (g.epi == UNITER_EPI_BIAS_GELU) {
                if (g.aux_out) g.aux_out[(size_t)row * g.ld_aux + col] = v;
                v = gelu_erf(v);
              } else if (g.epi == UNITER_EPI_DGELU) {
                v *= dgelu_erf(g.aux_in[(size_t)row * g.ld_aux + col]);
              } else if (g.epi == UNITER_EPI_ADD) {
                v += g.aux_in[(size_t)row * g.ld_aux + col];
              }
              csum += v;
              if (g.C) {
                float* c = g.C + (size_t)row * g.ldc + col;
                if (g.beta) v += *c;
                *c = v;
              }
              if (g.Cp) {       // planes of the stored value for the next contraction
                unsigned h1, h2, h3;
                split1(v, h1, h2, h3);
                bf16_t* cp = g.Cp + (size_t)row * g.ldcp + col;
                cp[0] = (bf16_t)(h1 >> 16); cp[g.psc] = (bf16_t)(h2 >> 16); cp[2 * g.psc] = (bf16_t)(h3 >> 16);
              }
            }
            acc[a][b][rr] = 0.f;
          }
          if (g.colsum_part) {
            csum += __shfl_xor(csum, 32, 64);
            if (h == 0 && cok && (m0 + wm * WM + a * 32) < g.M)
              g.colsum_part[(size_t)((m0 + wm * WM + a * 32) >> 5) * g.N + col] = csum;
          }
        }
      }
      ck = 0; ct += per_xcd;
      if (more) {
        tile_coords_p(chunk0 + ct, g.tiles_m, g.tiles_n, g.band_h, tmi0, tni0);
        m0 = tmi0 * BM; n0 = tni0 * BN;
      }
    }
  }
#undef MFMA_BLOCK
#undef MFMA_ONE
#undef READ_FRAGS
#undef LOAD_UNIT
}

template <int BM, int BN, int WPS>
int launch_p(GemmArgsP g, hipStream_t st, int slots) {
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = (g.N + BN - 1) / BN;
  const int tiles = g.tiles_m * g.tiles_n;
  const long panel = (long)BM * g.K * 6;
  long bh = (3l << 19) / (panel > 0 ? panel : 1);
  g.band_h = (int)(bh < 1 ? 1 : (bh > 16 ? 16 : bh));
  if (g.band_h > g.tiles_m) g.band_h = g.tiles_m;
  const int grid = tiles < slots ? (tiles + 7) / 8 * 8 : slots;
  hipLaunchKernelGGL((gemm_planes_kernel<BM, BN, WPS>), dim3(grid), dim3(256), 0, st, g);
  UCHECK_LAUNCH();
  return 0;
}

}  // namespace

int split_planes_run(const float* x, int rows, int cols, int ld, unsigned short* planes, int pld, size_t pstride,
                     int transpose, hipStream_t st) {
  UCHECK_ARG(x && planes && rows > 0 && cols > 0, "split_planes: bad argument");
  UCHECK_SHAPE(cols % 4 == 0 && ld % 4 == 0 && pld % 4 == 0 && (!transpose || rows % 4 == 0),
               "split_planes: dimensions must be multiples of 4");
  if (!transpose)
    hipLaunchKernelGGL(split_planes_kernel, dim3((cols / 4 + 63) / 64, (rows + 3) / 4), dim3(256), 0, st, x, rows, cols,
                       ld, planes, pld, pstride);
  else
    hipLaunchKernelGGL(split_planes_t_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, st, x, rows, cols,
                       ld, planes, pld, pstride);
  UCHECK_LAUNCH();
  return 0;
}

int gemm_planes_run(int cfg, int M, int N, int K, const unsigned short* A, int lda, size_t psa,
                    const unsigned short* B, int ldb, size_t psb, float* C, int ldc, unsigned short* Cp, int ldcp,
                    size_t psc, int epilogue, const float* bias, const float* aux_in, float* aux_out, int ld_aux,
                    int beta, float* colsum_part, hipStream_t st) {
  UCHECK_ARG(M > 0 && N > 0 && K > 0 && A && B && (C || Cp), "gemm_planes: bad argument");
  UCHECK_SHAPE(K % BKP == 0 && lda % 8 == 0 && ldb % 8 == 0, "gemm_planes: K %% 32 and ld %% 8 required");
  UCHECK_ARG(((2 * psa + (size_t)M * lda) * 2) < (1ull << 31) && ((2 * psb + (size_t)N * ldb) * 2) < (1ull << 31),
             "gemm_planes: operand too large for 32-bit buffer offsets");
  UCHECK_ARG(epilogue >= 0 && epilogue <= UNITER_EPI_ADD, "gemm_planes: bad epilogue");
  GemmArgsP g;
  g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.psa = psa; g.B = B; g.ldb = ldb; g.psb = psb;
  g.C = C; g.ldc = ldc; g.Cp = Cp; g.ldcp = ldcp; g.psc = psc; g.epi = epilogue; g.bias = bias; g.aux_in = aux_in;
  g.aux_out = aux_out; g.ld_aux = ld_aux; g.beta = beta; g.tiles_m = g.tiles_n = 0; g.band_h = 1;
  g.colsum_part = colsum_part;
  if (cfg == 0) {
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    cfg = t128 >= 200 ? 1 : 4;
  }
  switch (cfg) {
    case 1: return launch_p<128, 128, 1>(g, st, 256);
    case 2: return launch_p<64, 128, 1>(g, st, 256);
    case 3: return launch_p<128, 64, 1>(g, st, 256);
    case 4: return launch_p<64, 64, 2>(g, st, 512);
    default: uniter_set_error("gemm_planes: bad cfg %d", cfg); return UNITER_E_ARG;
  }
}

extern "C" int uniter_split_planes(const float* x, int rows, int cols, int ld, void* planes, int pld,
                                   size_t plane_stride, int transpose, void* stream) {
  return split_planes_run(x, rows, cols, ld, (unsigned short*)planes, pld, plane_stride, transpose, (hipStream_t)stream);
}

extern "C" int uniter_gemm_planes_cfg(int cfg, int M, int N, int K, const void* A, int lda, size_t psa, const void* B,
                                      int ldb, size_t psb, float* C, int ldc, int epilogue, const float* bias,
                                      const float* aux_in, float* aux_out, int ld_aux, int beta, void* stream) {
  UCHECK_ARG(!(epilogue == UNITER_EPI_BIAS || epilogue == UNITER_EPI_BIAS_GELU) || bias, "gemm_planes: epilogue needs bias");
  UCHECK_ARG(!(epilogue == UNITER_EPI_DGELU || epilogue == UNITER_EPI_ADD) || aux_in, "gemm_planes: epilogue needs aux_in");
  return gemm_planes_run(cfg, M, N, K, (const unsigned short*)A, lda, psa, (const unsigned short*)B, ldb, psb, C, ldc,
                         nullptr, 0, 0, epilogue, bias, aux_in, aux_out, ld_aux, beta, nullptr, (hipStream_t)stream);
}
