// fp32-ACCURATE GEMM on the bf16 matrix pipe of gfx950 ("3 x bf16 split, 6 products").
//
// Every fp32 operand value x is split EXACTLY into three bf16 pieces by truncation,
//     x = x1 + x2 + x3,   x1 = hi16(x), x2 = hi16(x - x1), x3 = hi16(x - x1 - x2)
// (8 + 8 + 8 = 24 significant bits, all pieces carry x's sign, the two subtractions are exact in
// fp32), and the product is assembled from the six bf16 x bf16 products whose weight is >= 2^-16:
//     a*b ~= a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1)          (dropped: <= 2^-24 |a b|)
// Each bf16 x bf16 product is exact in fp32 and v_mfma_f32_32x32x16_bf16 accumulates in fp32, so the
// result carries the same O(2^-24 * sum|a b|) error as the native fp32 MFMA chain it replaces
// (tests/test_gemm_split_gpu.py measures both against float64).  The bf16 pipe is 16x the fp32
// pipe (2.5 PFLOP/s vs 157 TFLOP/s), so six products still leave 2.7x: 417 TFLOP/s of
// fp32-equivalent work at pipe peak.
//
// Structure = gemm_bf16.hip (persistent workgroups, banded tile order, buffer loads one full
// iteration ahead, transposing staging for k-major operands) with three LDS planes per operand.
// Unlike the one-product bf16 kernel this one is matrix-pipe bound again (48 MFMAs per wave per
// 32-deep k-tile), and the split arithmetic (6 VALU ops per element) rides in the MFMA shadow.
#include <stdlib.h>
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

struct GemmArgsS {
  int M, N, K;
  const float* A; int lda;
  const float* B; int ldb;
  float* C; int ldc;
  int epi;
  const float* bias;
  const float* aux_in;
  float* aux_out;
  int ld_aux;
  int beta;
  int tiles_m, tiles_n, band_h;
  float* colsum_part;
};

constexpr int BKS = 32;            // k-tile depth
constexpr int LDS16 = 40;          // LDS row stride in bf16 elements (80 B: 5 16-B slots, 5i mod 16 bijective)

// exact 3-way split of two floats, packed as bf16 pairs (lo = first float) for the three planes
__device__ __forceinline__ void split2(float x, float y, unsigned& p1, unsigned& p2, unsigned& p3) {
  const unsigned ux = __builtin_bit_cast(unsigned, x), uy = __builtin_bit_cast(unsigned, y);
  const unsigned x1 = ux & 0xffff0000u, y1 = uy & 0xffff0000u;
  const float rx = x - __builtin_bit_cast(float, x1), ry = y - __builtin_bit_cast(float, y1);
  const unsigned x2 = __builtin_bit_cast(unsigned, rx) & 0xffff0000u, y2 = __builtin_bit_cast(unsigned, ry) & 0xffff0000u;
  const float sx = rx - __builtin_bit_cast(float, x2), sy = ry - __builtin_bit_cast(float, y2);
  const unsigned x3 = __builtin_bit_cast(unsigned, sx) & 0xffff0000u, y3 = __builtin_bit_cast(unsigned, sy) & 0xffff0000u;
  p1 = (x1 >> 16) | y1;
  p2 = (x2 >> 16) | y2;
  p3 = (x3 >> 16) | y3;
}

__device__ __forceinline__ void tile_coords_s(int t, int tiles_m, int tiles_n, int band_h, int& tm, int& tn) {
  const int full = band_h * tiles_n;
  const int band = t / full;
  const int rem = t - band * full;
  const int bh = min(band_h, tiles_m - band * band_h);
  tn = rem / bh;
  tm = band * band_h + (rem - tn * bh);
}

template <int R, bool KM>
__device__ __forceinline__ void tile_offsets_s(int (&voff)[R / 32], int ld, int row0, int tid) {
  if constexpr (!KM) {
    const int c4 = tid & 7, rr = tid >> 3;           // 8 threads cover one row's 32 k
#pragma unroll
    for (int p = 0; p < R / 32; ++p) voff[p] = ((row0 + rr + 32 * p) * ld + c4 * 4) * 4;
  } else {
    // item = (4 columns, k-pair): columns fastest over 4 lanes, then 16 k-pairs
#pragma unroll
    for (int p = 0; p < R / 64; ++p) {
      const int id = tid + 256 * p;
      const int c4 = ((id >> 6) << 2) | (id & 3), kp = (id >> 2) & 15;
      voff[2 * p] = ((2 * kp) * ld + row0 + c4 * 4) * 4;
      voff[2 * p + 1] = voff[2 * p] + ld * 4;
    }
  }
}

template <int R>
__device__ __forceinline__ void tile_load_s(f32x4 (&reg)[R / 32], __amdgpu_buffer_rsrc_t rsrc,
                                            const int (&voff)[R / 32], int soff) {
#pragma unroll
  for (int p = 0; p < R / 32; ++p)
    reg[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[p], soff, 0));
}

// split + write one staged tile into its three LDS planes, each [R][LDS16] bf16
template <int R, bool KM>
__device__ __forceinline__ void tile_store_s(const f32x4 (&reg)[R / 32], unsigned short* s, int tid) {
  constexpr int PL = R * LDS16;      // plane size in bf16 elements
  if constexpr (!KM) {
    const int c4 = tid & 7, rr = tid >> 3;
#pragma unroll
    for (int p = 0; p < R / 32; ++p) {
      unsigned a1, a2, a3, b1, b2, b3;
      split2(reg[p][0], reg[p][1], a1, a2, a3);
      split2(reg[p][2], reg[p][3], b1, b2, b3);
      unsigned short* d = s + (rr + 32 * p) * LDS16 + c4 * 4;
      *reinterpret_cast<u32x2_t*>(d) = u32x2_t{a1, b1};
      *reinterpret_cast<u32x2_t*>(d + PL) = u32x2_t{a2, b2};
      *reinterpret_cast<u32x2_t*>(d + 2 * PL) = u32x2_t{a3, b3};
    }
  } else {
#pragma unroll
    for (int p = 0; p < R / 64; ++p) {
      const int id = tid + 256 * p;
      const int c4 = ((id >> 6) << 2) | (id & 3), kp = (id >> 2) & 15;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned q1, q2, q3;
        split2(reg[2 * p][e], reg[2 * p + 1][e], q1, q2, q3);      // (k, k+1) pair of column 4*c4+e
        unsigned short* d = s + (c4 * 4 + e) * LDS16 + 2 * kp;
        *reinterpret_cast<unsigned*>(d) = q1;
        *reinterpret_cast<unsigned*>(d + PL) = q2;
        *reinterpret_cast<unsigned*>(d + 2 * PL) = q3;
      }
    }
  }
}

__device__ __forceinline__ bf16x8 frag_read_s(const unsigned short* s, int r0, int ks, int i, int h) {
  return *reinterpret_cast<const bf16x8*>(s + (r0 + i) * LDS16 + ks * 16 + 8 * h);
}

template <int BM, int BN, bool AKM, bool BKM, int WPS>
__global__ __launch_bounds__(256, WPS) void gemm_split_kernel(const GemmArgsS g) {
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  constexpr int PA = BM * LDS16, PB = BN * LDS16;              // one plane, bf16 elements
  constexpr int STAGE = 3 * (PA + PB);
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntiles = g.tiles_m * g.tiles_n;
  const int nk = g.K / BKS;

  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  const int q = ntiles >> 3, r = ntiles & 7;
  const int chunk0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int chunk_n = q + (xcd < r ? 1 : 0);
  if (idx >= chunk_n) return;
  const int my_tiles = (chunk_n - idx + per_xcd - 1) / per_xcd;
  const int total_units = my_tiles * nk;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(g.A), 0, (AKM ? g.K : g.M) * g.lda * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(g.B), 0, (BKM ? g.K : g.N) * g.ldb * 4, 0x00020000);
  const int kstepA = (AKM ? BKS * g.lda : BKS) * 4, kstepB = (BKM ? BKS * g.ldb : BKS) * 4;

  int lt = idx, lk = 0;
  int voA[BM / 32], voB[BN / 32];
  {
    int tmi, tni;
    tile_coords_s(chunk0 + lt, g.tiles_m, g.tiles_n, g.band_h, tmi, tni);
    tile_offsets_s<BM, AKM>(voA, g.lda, tmi * BM, tid);
    tile_offsets_s<BN, BKM>(voB, g.ldb, tni * BN, tid);
  }
  int loaded = 0;
// branch-free (keeps the MFMA / VALU interleave in one scheduling region): loads past the end of
// the unit sequence use an out-of-range scalar offset, for which buffer loads return 0
#define LOAD_UNIT(RA, RB)                                                                \
  do {                                                                                   \
    const bool live_ = loaded < total_units;                                             \
    tile_load_s<BM>(RA, rsA, voA, live_ ? lk * kstepA : 0x7ffffff0);                     \
    tile_load_s<BN>(RB, rsB, voB, live_ ? lk * kstepB : 0x7ffffff0);                     \
    ++loaded;                                                                            \
    if (++lk == nk) {                                                                    \
      lk = 0; lt += per_xcd;                                                             \
      int tmi_, tni_;                                                                    \
      tile_coords_s(min(chunk0 + lt, ntiles - 1), g.tiles_m, g.tiles_n, g.band_h, tmi_, tni_); \
      tile_offsets_s<BM, AKM>(voA, g.lda, tmi_ * BM, tid);                               \
      tile_offsets_s<BN, BKM>(voB, g.ldb, tni_ * BN, tid);                               \
    }                                                                                    \
  } while (0)

  // two raw staging sets: while one (unit u+1) is being split + written to LDS during iteration u,
  // the other receives unit u+2 -- every global load has a full iteration to land
  f32x4 ra0[BM / 32], rb0[BN / 32], ra1[BM / 32], rb1[BN / 32];
  bf16x8 fa0[TM][3], fb0[TN][3], fa1[TM][3], fb1[TN][3];

  LOAD_UNIT(ra0, rb0);                                   // unit 0
  LOAD_UNIT(ra1, rb1);                                   // unit 1
  tile_store_s<BM, AKM>(ra0, smem, tid);
  tile_store_s<BN, BKM>(rb0, smem + 3 * PA, tid);
  __syncthreads();

#define READ_FRAGS(FA, FB, ST, KS)                                                                      \
  _Pragma("unroll") for (int a = 0; a < TM; ++a)                                                        \
  _Pragma("unroll") for (int p = 0; p < 3; ++p)                                                         \
      FA[a][p] = frag_read_s((ST) + p * PA, wm * WM + a * 32, KS, i, h);                                \
  _Pragma("unroll") for (int b = 0; b < TN; ++b)                                                        \
  _Pragma("unroll") for (int p = 0; p < 3; ++p)                                                         \
      FB[b][p] = frag_read_s((ST) + 3 * PA + p * PB, wn * WN + b * 32, KS, i, h);
// the six products (smallest weights first), split in two groups so that the tail of the k-tile's
// MFMAs can run behind the barrier and cover the first fragment reads of the next stage
#define MFMA_P03(FA, FB)                                                                                \
  _Pragma("unroll") for (int a = 0; a < TM; ++a)                                                        \
  _Pragma("unroll") for (int b = 0; b < TN; ++b) {                                                      \
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[a][0], FB[b][2], acc[a][b], 0, 0, 0);      \
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[a][1], FB[b][1], acc[a][b], 0, 0, 0);      \
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[a][2], FB[b][0], acc[a][b], 0, 0, 0);      \
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[a][0], FB[b][1], acc[a][b], 0, 0, 0);      \
  }
#define MFMA_P45(FA, FB)                                                                                \
  _Pragma("unroll") for (int a = 0; a < TM; ++a)                                                        \
  _Pragma("unroll") for (int b = 0; b < TN; ++b) {                                                      \
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[a][1], FB[b][0], acc[a][b], 0, 0, 0);      \
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[a][0], FB[b][0], acc[a][b], 0, 0, 0);      \
  }

  READ_FRAGS(fa0, fb0, smem, 0)

  int ct = idx, ck = 0;
  int tmi0, tni0;
  tile_coords_s(chunk0 + ct, g.tiles_m, g.tiles_n, g.band_h, tmi0, tni0);
  int m0 = tmi0 * BM, n0 = tni0 * BN;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) acc[a][b][rr] = 0.f;

#define EPILOGUE()                                                                                      \
  _Pragma("unroll") for (int a = 0; a < TM; ++a) {                                                      \
    _Pragma("unroll") for (int b = 0; b < TN; ++b) {                                                    \
      const int col = n0 + wn * WN + b * 32 + i;                                                        \
      const bool cok = col < g.N;                                                                       \
      const float bv = (cok && (g.epi == UNITER_EPI_BIAS || g.epi == UNITER_EPI_BIAS_GELU)) ? g.bias[col] : 0.f; \
      float csum = 0.f;                                                                                 \
      _Pragma("unroll") for (int rr = 0; rr < 16; ++rr) {                                               \
        const int row = m0 + wm * WM + a * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;                       \
        if (cok && row < g.M) {                                                                         \
          float v = acc[a][b][rr] + bv;                                                                 \
          if (g.epi == UNITER_EPI_BIAS_GELU) {                                                          \
            if (g.aux_out) g.aux_out[(size_t)row * g.ld_aux + col] = v;                                 \
            v = gelu_erf(v);                                                                            \
          } else if (g.epi == UNITER_EPI_DGELU) {                                                       \
            v *= dgelu_erf(g.aux_in[(size_t)row * g.ld_aux + col]);                                     \
          } else if (g.epi == UNITER_EPI_ADD) {                                                         \
            v += g.aux_in[(size_t)row * g.ld_aux + col];                                                \
          }                                                                                             \
          csum += v;                                                                                    \
          float* c = g.C + (size_t)row * g.ldc + col;                                                   \
          if (g.beta) v += *c;                                                                          \
          *c = v;                                                                                       \
        }                                                                                               \
        acc[a][b][rr] = 0.f;                                                                            \
      }                                                                                                 \
      if (g.colsum_part) {                                                                              \
        csum += __shfl_xor(csum, 32, 64);                                                               \
        if (h == 0 && cok && (m0 + wm * WM + a * 32) < g.M)                                             \
          g.colsum_part[(size_t)((m0 + wm * WM + a * 32) >> 5) * g.N + col] = csum;                     \
      }                                                                                                 \
    }                                                                                                   \
  }

// iteration u: unit u is in LDS stage u&1; RS* hold unit u+1 (split + stored now, spread under the
// MFMAs of the whole k-tile); RL* receive unit u+2
#define K_ITERATION(U, RSA, RSB, RLA, RLB)                                                              \
  {                                                                                                     \
    const unsigned short* sS = smem + ((U) & 1) * STAGE;                                                \
    unsigned short* dS = smem + (((U) + 1) & 1) * STAGE;                                                \
    const bool more = (U) + 1 < total_units;                                                            \
    LOAD_UNIT(RLA, RLB);                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    READ_FRAGS(fa1, fb1, sS, 1)                                                                         \
    MFMA_P03(fa0, fb0)                                                                                  \
    MFMA_P45(fa0, fb0)                                                                                  \
    MFMA_P03(fa1, fb1)                                                                                  \
    tile_store_s<BM, AKM>(RSA, dS, tid);      /* (after the last unit: a stage nobody reads) */         \
    tile_store_s<BN, BKM>(RSB, dS + 3 * PA, tid);                                                       \
    _Pragma("unroll") for (int sgi = 0; sgi < 10 * TM * TN; ++sgi) {                                    \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      /* 1 MFMA            */                   \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      /* 1 LDS read        */                   \
      __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);      /* 5 VALU (split)    */                   \
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      /* 1 LDS write       */                   \
    }                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    __syncthreads();                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    if (more) { READ_FRAGS(fa0, fb0, dS, 0) }                                                           \
    MFMA_P45(fa1, fb1)                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    if (++ck == nk) {                                                                                   \
      EPILOGUE();                                                                                       \
      ck = 0; ct += per_xcd;                                                                            \
      if (more) {                                                                                       \
        tile_coords_s(chunk0 + ct, g.tiles_m, g.tiles_n, g.band_h, tmi0, tni0);                         \
        m0 = tmi0 * BM; n0 = tni0 * BN;                                                                 \
      }                                                                                                 \
    }                                                                                                   \
  }

  for (int u = 0; u < total_units; u += 2) {
    K_ITERATION(u, ra1, rb1, ra0, rb0)
    if (u + 1 < total_units) K_ITERATION(u + 1, ra0, rb0, ra1, rb1)
  }
#undef K_ITERATION
#undef EPILOGUE
#undef MFMA_P03
#undef MFMA_P45
#undef READ_FRAGS
#undef LOAD_UNIT
}

template <int BM, int BN, bool AKM, bool BKM, int WPS>
int launch_s(GemmArgsS g, hipStream_t st, int slots) {
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = (g.N + BN - 1) / BN;
  const int tiles = g.tiles_m * g.tiles_n;
  const long panel = (long)BM * g.K * 4;
  long bh = (3l << 19) / (panel > 0 ? panel : 1);
  g.band_h = (int)(bh < 1 ? 1 : (bh > 16 ? 16 : bh));
  if (g.band_h > g.tiles_m) g.band_h = g.tiles_m;
  const int grid = tiles < slots ? (tiles + 7) / 8 * 8 : slots;
  hipLaunchKernelGGL((gemm_split_kernel<BM, BN, AKM, BKM, WPS>), dim3(grid), dim3(256), 0, st, g);
  UCHECK_LAUNCH();
  return 0;
}

template <bool AKM, bool BKM>
int dispatch_s(int cfg, const GemmArgsS& g, hipStream_t st) {
  switch (cfg) {
    case 1: return launch_s<128, 128, AKM, BKM, 1>(g, st, 256);     // 120 KB LDS: one workgroup per CU
    case 2: return launch_s<64, 128, AKM, BKM, 1>(g, st, 256);      // 90 KB
    case 3: return launch_s<128, 64, AKM, BKM, 1>(g, st, 256);
    case 4: return launch_s<64, 64, AKM, BKM, 2>(g, st, 512);       // 60 KB: two per CU
    default: uniter_set_error("gemm_split: bad cfg %d", cfg); return UNITER_E_ARG;
  }
}

}  // namespace

int gemm_f32_run(int cfg, int tag, int a_kmajor, int b_kmajor, int M, int N, int K, const float* A, int lda,
                 const float* B, int ldb, float* C, int ldc, int epilogue, const float* bias,
                 const float* aux_in, float* aux_out, int ld_aux, int beta, float* colsum_part, void* stream);

int gemm_split_run(int cfg, int a_kmajor, int b_kmajor, int M, int N, int K, const float* A, int lda,
                   const float* B, int ldb, float* C, int ldc, int epilogue, const float* bias,
                   const float* aux_in, float* aux_out, int ld_aux, int beta, float* colsum_part, void* stream) {
  UCHECK_ARG(M > 0 && N > 0 && K > 0 && A && B && C, "gemm_split: bad argument");
  const bool ok = K % BKS == 0 && lda % 4 == 0 && ldb % 4 == 0 &&
                  (a_kmajor ? M % 4 == 0 : true) && (b_kmajor ? N % 4 == 0 : true) &&
                  (size_t)(a_kmajor ? K : M) * lda * 4 < (1ull << 31) &&
                  (size_t)(b_kmajor ? K : N) * ldb * 4 < (1ull << 31) &&
                  ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0;
  if (!ok)
    return gemm_f32_run(0, 0, a_kmajor, b_kmajor, M, N, K, A, lda, B, ldb, C, ldc, epilogue, bias, aux_in, aux_out,
                        ld_aux, beta, colsum_part, stream);
  UCHECK_ARG(epilogue >= 0 && epilogue <= UNITER_EPI_ADD, "gemm_split: bad epilogue %d", epilogue);
  GemmArgsS g;
  g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.epi = epilogue; g.bias = bias; g.aux_in = aux_in; g.aux_out = aux_out; g.ld_aux = ld_aux; g.beta = beta;
  g.tiles_m = g.tiles_n = 0; g.band_h = 1; g.colsum_part = colsum_part;
  if (cfg == 0) cfg = 4;
  hipStream_t st = (hipStream_t)stream;
  if (!a_kmajor && !b_kmajor) return dispatch_s<false, false>(cfg, g, st);
  if (!a_kmajor && b_kmajor) return dispatch_s<false, true>(cfg, g, st);
  if (a_kmajor && b_kmajor) return dispatch_s<true, true>(cfg, g, st);
  return dispatch_s<true, false>(cfg, g, st);
}

extern "C" int uniter_gemm_f32x3_cfg(int cfg, int a_kmajor, int b_kmajor, int M, int N, int K, const float* A,
                                     int lda, const float* B, int ldb, float* C, int ldc, int epilogue,
                                     const float* bias, const float* aux_in, float* aux_out, int ld_aux, int beta,
                                     void* stream) {
  UCHECK_ARG(!(epilogue == UNITER_EPI_BIAS || epilogue == UNITER_EPI_BIAS_GELU) || bias, "gemm_split: epilogue needs bias");
  UCHECK_ARG(!(epilogue == UNITER_EPI_DGELU || epilogue == UNITER_EPI_ADD) || aux_in, "gemm_split: epilogue needs aux_in");
  return gemm_split_run(cfg, a_kmajor, b_kmajor, M, N, K, A, lda, B, ldb, C, ldc, epilogue, bias, aux_in, aux_out,
                        ld_aux, beta, nullptr, stream);
}
