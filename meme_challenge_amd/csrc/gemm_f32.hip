// fp32 MFMA GEMM for gfx950:  C[M,N] (+)= epi( sum_k A(m,k) B(k,n) ).
//
// Replaces the cuBLAS sgemm calls behind nn.Linear / its autograd in the
// reference (model/layer.py:76-78,112,140,153; model/model.py:267).
//
// Design (MI355X_MICROARCH.md "Matrix cores", cdna_hip_programming.md 3) of the kernel the model runs,
// gemm_f32_v3_kernel (gemm_f32_kernel further down is the one-tile-per-workgroup fallback for K % 32 != 0):
//  * v_mfma_f32_32x32x2_f32: exact fp32, 64 cycles/SIMD per instruction, peak 157.3 TFLOP/s -- 1/16 of the bf16
//    rate and EQUAL to the fp32 vector rate, so the kernel is matrix-pipe bound and every VALU instruction beside
//    the MFMAs costs its full issue time.
//  * PERSISTENT 64 x 64 tiles: 256-thread workgroups (4 waves, one 32 x 32 accumulator each), four resident per CU
//    (71 - 105 VGPRs, 36 KB LDS each; __launch_bounds__(256, 2) only states the floor of two), 1024 slots walking an
//    XCD-chunked, L2-banded tile order; the k-tiles of all of a workgroup's tiles form one flat sequence of 32-deep
//    units, so tiles finish out of phase and one tile's epilogue overlaps other tiles' MFMAs.  128 x 128 tiles
//    (512 slots) from 3072 tiles of 64 x 64 up.
//  * the MFMA K index is a free permutation: lane-half h of instruction t in an 8-deep k-block takes
//    k = 8*kb + 4*h + t for BOTH operands.  A k-contiguous operand fragment is then ONE ds_read_b128 (4 consecutive
//    k) per 4 MFMAs.
//  * k-contiguous tiles sit in LDS as [rows][32+4] floats: the 36-dword stride maps the 16 lanes of every
//    ds_read_b128 lane group to 16 distinct 4-bank slots (9*i mod 16 is a bijection) -> conflict free.
//  * k-major tiles (dgrad's W, wgrad's dY and X) sit as [32][cols+4]; a fragment is 4 ds_read_b32 with lanes 0..31
//    on consecutive dwords -> conflict free.
//  * operands: buffer_load_dwordx4 with the hardware range check (no bounds branches) -> staging registers -> LDS,
//    written mid-iteration, refetched three units ahead; the k-loop is unrolled by two so every LDS address is an
//    immediate; one __syncthreads per unit.
//  * stream-K (SK = true) for C += A.B with few tiles and a long K (weight gradients): an XCD's unit sequence is cut
//    into equal pieces, partial tiles added with buffer_atomic_add_f32.
//  * epilogues through buffer instructions with scalar row offsets; optional per-32-row column sums.
#include <stdlib.h>
#include "common.h"

namespace {

struct GemmArgs {
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
  int tiles_m, tiles_n;
  int band_h;
  float* colsum_part;   // optional [ceil(M/32)][N]: per-32-row-block column sums of the stored C (v3 only)
  unsigned long long* stamp;    // optional {first start, last end} slot (common.h)
  int prio;                     // wave priority (common.h: g_uniter_launch_prio)
};

__device__ __forceinline__ float buf_ld_f32(__amdgpu_buffer_rsrc_t r, int voff, int soff, int aux) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_st_f32(float v, __amdgpu_buffer_rsrc_t r, int voff, int soff, int aux) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

constexpr int BK = 32;
constexpr int LDK = BK + 4;   // row stride (floats) of a k-contiguous tile

template <int R, bool KM> struct TileSize {
  static constexpr int value = KM ? BK * (R + 4) : R * LDK;
};

// issue the global loads of one operand tile into registers
template <int R, bool KM>
__device__ __forceinline__ void tile_load(f32x4 (&reg)[R / 32], const float* __restrict__ P,
                                          int ld, int row0, int rows, int k0, int K, int tid) {
  if constexpr (!KM) {
    const int c4 = tid & 7, rr = tid >> 3;
#pragma unroll
    for (int p = 0; p < R / 32; ++p) {
      const int r = row0 + rr + 32 * p;
      const int k = k0 + c4 * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < rows && k < K) v = *reinterpret_cast<const f32x4*>(P + (size_t)r * ld + k);
      reg[p] = v;
    }
  } else {
    constexpr int TPR = R / 4;          // threads per k-row
    constexpr int RPP = 256 / TPR;      // k-rows per pass
    const int c4 = tid % TPR, kk0 = tid / TPR;
#pragma unroll
    for (int p = 0; p < R / 32; ++p) {
      const int k = k0 + kk0 + RPP * p;
      const int c = row0 + c4 * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (k < K && c < rows) v = *reinterpret_cast<const f32x4*>(P + (size_t)k * ld + c);
      reg[p] = v;
    }
  }
}

// Branch-free variant: raw buffer loads with hardware range checking (out-of-range rows of
// a k-contiguous operand and out-of-range k-rows of a k-major operand read as 0).  Needs
// K % 32 == 0 for k-contiguous operands (no in-row k tail).  voff[] holds the per-pass byte
// offsets for k0 = 0; the k advance goes in the scalar offset.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

template <int R, bool KM>
__device__ __forceinline__ void tile_offsets(int (&voff)[R / 32], int ld, int row0, int tid) {
  if constexpr (!KM) {
    const int c4 = tid & 7, rr = tid >> 3;
#pragma unroll
    for (int p = 0; p < R / 32; ++p) voff[p] = ((row0 + rr + 32 * p) * ld + c4 * 4) * 4;
  } else {
    constexpr int TPR = R / 4;
    constexpr int RPP = 256 / TPR;
    const int c4 = tid % TPR, kk0 = tid / TPR;
#pragma unroll
    for (int p = 0; p < R / 32; ++p) voff[p] = ((kk0 + RPP * p) * ld + row0 + c4 * 4) * 4;
  }
}

template <int R>
__device__ __forceinline__ void tile_load_buf(f32x4 (&reg)[R / 32], __amdgpu_buffer_rsrc_t rsrc,
                                              const int (&voff)[R / 32], int soff) {
#pragma unroll
  for (int p = 0; p < R / 32; ++p)
    reg[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[p], soff, 0));
}

template <int R, bool KM>
__device__ __forceinline__ void tile_store(const f32x4 (&reg)[R / 32], float* s, int tid) {
  if constexpr (!KM) {
    const int c4 = tid & 7, rr = tid >> 3;
#pragma unroll
    for (int p = 0; p < R / 32; ++p)
      *reinterpret_cast<f32x4*>(s + (rr + 32 * p) * LDK + c4 * 4) = reg[p];
  } else {
    constexpr int TPR = R / 4;
    constexpr int RPP = 256 / TPR;
    const int c4 = tid % TPR, kk0 = tid / TPR;
#pragma unroll
    for (int p = 0; p < R / 32; ++p)
      *reinterpret_cast<f32x4*>(s + (kk0 + RPP * p) * (R + 4) + c4 * 4) = reg[p];
  }
}

// fragment of the 32-row sub-tile starting at row `r0` for k-block kb
template <int R, bool KM>
__device__ __forceinline__ f32x4 frag_read(const float* s, int r0, int kb, int i, int h) {
  if constexpr (!KM) {
    return *reinterpret_cast<const f32x4*>(s + (r0 + i) * LDK + kb * 8 + 4 * h);
  } else {
    const float* p = s + (kb * 8 + 4 * h) * (R + 4) + r0 + i;
    f32x4 v;
    v[0] = p[0];
    v[1] = p[R + 4];
    v[2] = p[2 * (R + 4)];
    v[3] = p[3 * (R + 4)];
    return v;
  }
}

template <int BM, int BN, bool AKM, bool BKM, int TAG>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const GemmArgs g) {
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  constexpr int SA = TileSize<BM, AKM>::value, SB = TileSize<BN, BKM>::value;
  __shared__ __attribute__((aligned(16))) float smem[2 * (SA + SB)];

  // XCD-aware bijective remap: blocks b, b+8, ... share an XCD -> contiguous tile chunk
  const int nwg = gridDim.x;
  int tile;
  {
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = tile / g.tiles_n, tile_n = tile - tile_m * g.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  f32x4 ra[BM / 32], rb[BN / 32];
  const int nk = (g.K + BK - 1) / BK;

  tile_load<BM, AKM>(ra, g.A, g.lda, m0, g.M, 0, g.K, tid);
  tile_load<BN, BKM>(rb, g.B, g.ldb, n0, g.N, 0, g.K, tid);
  tile_store<BM, AKM>(ra, smem, tid);
  tile_store<BN, BKM>(rb, smem + SA, tid);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const float* sA = smem + (kt & 1) * (SA + SB);
    const float* sB = sA + SA;
    if (kt + 1 < nk) {
      tile_load<BM, AKM>(ra, g.A, g.lda, m0, g.M, (kt + 1) * BK, g.K, tid);
      tile_load<BN, BKM>(rb, g.B, g.ldb, n0, g.N, (kt + 1) * BK, g.K, tid);
    }
#pragma unroll
    for (int kb = 0; kb < BK / 8; ++kb) {
      f32x4 fa[TM], fb[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a) fa[a] = frag_read<BM, AKM>(sA, wm * WM + a * 32, kb, i, h);
#pragma unroll
      for (int b = 0; b < TN; ++b) fb[b] = frag_read<BN, BKM>(sB, wn * WN + b * 32, kb, i, h);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][t], fb[b][t], acc[a][b], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      float* dA = smem + ((kt + 1) & 1) * (SA + SB);
      tile_store<BM, AKM>(ra, dA, tid);
      tile_store<BN, BKM>(rb, dA + SA, tid);
    }
    __syncthreads();
  }

  // epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int a = 0; a < TM; ++a) {
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int col = n0 + wn * WN + b * 32 + i;
      if (col >= g.N) continue;
      const float bv = (g.epi == UNITER_EPI_BIAS || g.epi == UNITER_EPI_BIAS_GELU || g.epi == UNITER_EPI_BIAS_GELU_D) ? g.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= g.M) continue;
        float v = acc[a][b][r] + bv;
        if (g.epi == UNITER_EPI_BIAS_GELU) {
          if (g.aux_out) g.aux_out[(size_t)row * g.ld_aux + col] = v;
          v = gelu_erf(v);
        } else if (g.epi == UNITER_EPI_DGELU) {
          v *= dgelu_erf(g.aux_in[(size_t)row * g.ld_aux + col]);
        } else if (g.epi == UNITER_EPI_ADD) {
          v += g.aux_in[(size_t)row * g.ld_aux + col];
        } else if (g.epi == UNITER_EPI_MUL) {
          v *= g.aux_in[(size_t)row * g.ld_aux + col];
        } else if (g.epi == UNITER_EPI_BIAS_GELU_D) {
          float dg;
          gelu_pair_fast(v, v, dg);
          if (g.aux_out) g.aux_out[(size_t)row * g.ld_aux + col] = dg;
        }
        float* c = g.C + (size_t)row * g.ldc + col;
        if (g.beta) v += *c;
        *c = v;
      }
    }
  }
}

template <int BM, int BN, bool AKM, bool BKM, int TAG>
int launch(GemmArgs g, hipStream_t st) {
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = (g.N + BN - 1) / BN;
  const int tiles = g.tiles_m * g.tiles_n;
  hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, AKM, BKM, TAG>), dim3(tiles), dim3(256), 0, st, g);
  UCHECK_LAUNCH();
  return 0;
}


// ---------------------------------------------------------------------------
// Persistent, software-pipelined across the barrier (schedule of the v3 kernel below).
//   * each workgroup walks a static sequence of output tiles; the first k-tile
//     of the NEXT output tile is fetched during the last k-iteration of the
//     current one, so the epilogue stores overlap the next tile's prologue
//   * per k-iteration the four 8-deep k-blocks are issued as
//         [kb0 MFMAs | read kb1] [kb1 MFMAs | read kb2] [write next tile -> LDS]
//         [kb2 MFMAs | read kb3] [barrier] [kb3 MFMAs | read next tile's kb0]
//     so fragment reads always have a 16-MFMA (1024-cycle) shadow, the
//     ds_writes complete under kb2's MFMAs, and the first MFMA after the barrier
//     never waits for LDS.
// ---------------------------------------------------------------------------
// L2-aware tile order.  The workgroups of one XCD run ~128 consecutive tiles of the linear
// order at once (32 CUs x 4 resident workgroups).  Row-major order makes that window 2-3 tile
// rows x all tile columns: the whole B operand streams through the 4 MiB L2 again for every few
// rows (measured on FFN-up: 68 % L2 hit rate, ~250 MB fetched beyond L2 for 17 MB of operands).
// Banded order: bands of `band_h` tile rows whose A panel stays L2-resident, tile columns swept
// inside a band with the row index fastest, so the window is band_h x ~(128/band_h) tiles.
__device__ __forceinline__ void tile_coords(int t, int tiles_m, int tiles_n, int band_h, int& tm, int& tn) {
  const int full = band_h * tiles_n;
  const int band = t / full;
  const int rem = t - band * full;
  const int bh = min(band_h, tiles_m - band * band_h);
  tn = rem / bh;
  tm = band * band_h + (rem - tn * bh);
}

// ---------------------------------------------------------------------------
// v3: the schedule above with a full-iteration prefetch distance.  The k-tiles a workgroup will consume
// (across all of its output tiles) form one flat sequence of "units"; in the middle of iteration u
// the staging registers (unit u+1, fetched during iteration u-1) are written to LDS and at once
// refilled with unit u+2, so every global load has a full k-iteration (>= 64 MFMAs of this wave,
// times the waves sharing the SIMD) to land instead of half of one.  A lone wave per SIMD lost ~19 % to vmcnt stalls with
// the one-deep version.  Needs the branch-free buffer-load path (K % 32 == 0).
// ---------------------------------------------------------------------------
//
// SK = true is the stream-K form for C += A.B (weight gradients: few output tiles, long K): an XCD's
// unit sequence (its tiles x their k-tiles) is cut into equal contiguous pieces, one per workgroup,
// so every slot of the chip gets the same number of k-iterations whatever the tile count; a piece
// that covers part of a tile's K range adds its partial sum with a float atomic.
template <int BM, int BN, bool AKM, bool BKM, int TAG, bool SK>
__global__ __launch_bounds__(256, 2) void gemm_f32_v3_kernel(const GemmArgs g) {
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  constexpr int SA = TileSize<BM, AKM>::value, SB = TileSize<BN, BKM>::value;
  __shared__ __attribute__((aligned(16))) float smem[2 * (SA + SB)];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntiles = g.tiles_m * g.tiles_n;
  const int nk = (g.K + BK - 1) / BK;      // a ragged last k-tile is legal for k-major operands only: rows >= K read as 0

  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  const int q = ntiles >> 3, r = ntiles & 7;
  const int chunk0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int chunk_n = q + (xcd < r ? 1 : 0);
  int total_units, t_first, k_first;
  if (SK) {
    const long U = (long)chunk_n * nk;
    const int u0 = (int)(U * idx / per_xcd), u1 = (int)(U * (idx + 1) / per_xcd);
    total_units = u1 - u0;
    t_first = u0 / nk; k_first = u0 - t_first * nk;
  } else {
    const int my_tiles = idx < chunk_n ? (chunk_n - idx + per_xcd - 1) / per_xcd : 0;
    total_units = my_tiles * nk;
    t_first = idx; k_first = 0;
  }
  if (total_units == 0) return;
  set_wave_prio(g.prio);
  stamp_begin(g.stamp);
  const int t_step = SK ? 1 : per_xcd;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(g.A), 0, (AKM ? g.K : g.M) * g.lda * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(g.B), 0, (BKM ? g.K : g.N) * g.ldb * 4, 0x00020000);
  const int kstepA = (AKM ? BK * g.lda : BK) * 4, kstepB = (BKM ? BK * g.ldb : BK) * 4;

  // load cursor: (tile, k) of the next unit to fetch
  int lt = t_first, lk = k_first, lm0, ln0;
  int voA[BM / 32], voB[BN / 32];
  {
    int tmi, tni;
    tile_coords(chunk0 + lt, g.tiles_m, g.tiles_n, g.band_h, tmi, tni);
    lm0 = tmi * BM; ln0 = tni * BN;
    tile_offsets<BM, AKM>(voA, g.lda, lm0, tid);
    tile_offsets<BN, BKM>(voB, g.ldb, ln0, tid);
  }
  int loaded = 0;     // units fetched so far
#define ADVANCE_LOAD_CURSOR()                                                            \
  do {                                                                                   \
    ++loaded;                                                                            \
    if (++lk == nk) {                                                                    \
      lk = 0; lt += t_step;                                                              \
      if (loaded < total_units) {                                                        \
        int tmi_, tni_;                                                                  \
        tile_coords(chunk0 + lt, g.tiles_m, g.tiles_n, g.band_h, tmi_, tni_);            \
        lm0 = tmi_ * BM; ln0 = tni_ * BN;                                                \
        tile_offsets<BM, AKM>(voA, g.lda, lm0, tid);                                     \
        tile_offsets<BN, BKM>(voB, g.ldb, ln0, tid);                                     \
      }                                                                                  \
    }                                                                                    \
  } while (0)
#define LOAD_UNIT(RA, RB)                                                                \
  do {                                                                                   \
    if (loaded < total_units) {                                                          \
      tile_load_buf<BM>(RA, rsA, voA, lk * kstepA);                                      \
      tile_load_buf<BN>(RB, rsB, voB, lk * kstepB);                                      \
      ADVANCE_LOAD_CURSOR();                                                             \
    }                                                                                    \
  } while (0)

  f32x4 ra0[BM / 32], rb0[BN / 32], ra1[BM / 32], rb1[BN / 32];
  f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];

  LOAD_UNIT(ra0, rb0);                       // unit 0
  tile_store<BM, AKM>(ra0, smem, tid);
  tile_store<BN, BKM>(rb0, smem + SA, tid);
  LOAD_UNIT(ra1, rb1);                       // unit 1 (stays in registers until iteration 0 stores it)
  __syncthreads();
#pragma unroll
  for (int a = 0; a < TM; ++a) fa0[a] = frag_read<BM, AKM>(smem, wm * WM + a * 32, 0, i, h);
#pragma unroll
  for (int b = 0; b < TN; ++b) fb0[b] = frag_read<BN, BKM>(smem + SA, wn * WN + b * 32, 0, i, h);

  // compute cursor
  int ct = t_first, ck = k_first;
  int tmi0, tni0;
  tile_coords(chunk0 + ct, g.tiles_m, g.tiles_n, g.band_h, tmi0, tni0);
  int m0 = tmi0 * BM, n0 = tni0 * BN;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) acc[a][b][rr] = 0.f;

  // ADD / MUL epilogues read a second [M, N] operand: fetch this wave's 16 values per tile two
  // k-iterations before the tile ends, so the epilogue does not start with an exposed HBM round trip
  const bool has_aux = g.epi == UNITER_EPI_ADD || g.epi == UNITER_EPI_MUL;
  bool aux_ready = false;
  float auxr[TM][TN][16];
#define AUXR(a, b, rr) auxr[a][b][rr]
#define PREFETCH_AUX()                                                                                  \
  {                                                                                                     \
    const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc(                                \
        const_cast<float*>(g.aux_in), 0, g.M * g.ld_aux * 4, 0x00020000);                               \
    _Pragma("unroll") for (int a = 0; a < TM; ++a)                                                      \
    _Pragma("unroll") for (int b = 0; b < TN; ++b) {                                                    \
      const int col = n0 + wn * WN + b * 32 + i;                                                        \
      const int r0 = m0 + wm * WM + a * 32 + 4 * h;                                                     \
      const int voX = col < g.N ? (r0 * g.ld_aux + col) * 4 : 0x7ffffff0;                               \
      _Pragma("unroll") for (int rr = 0; rr < 16; ++rr)                                                 \
        AUXR(a, b, rr) = buf_ld_f32(rsI, voX, ((rr & 3) + 8 * (rr >> 2)) * g.ld_aux * 4, 0); \
    }                                                                                                   \
    aux_ready = true;                                                                                   \
  }
#define MFMA_BLOCK(FA, FB)                                                                              \
  _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                         \
  _Pragma("unroll") for (int a = 0; a < TM; ++a)                                                        \
  _Pragma("unroll") for (int b = 0; b < TN; ++b)                                                        \
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[a][t], FB[b][t], acc[a][b], 0, 0, 0);
#define READ_FRAGS(FA, FB, SAp, SBp, KB)                                                                \
  _Pragma("unroll") for (int a = 0; a < TM; ++a) FA[a] = frag_read<BM, AKM>(SAp, wm * WM + a * 32, KB, i, h); \
  _Pragma("unroll") for (int b = 0; b < TN; ++b) FB[b] = frag_read<BN, BKM>(SBp, wn * WN + b * 32, KB, i, h);

// one k-iteration: unit u is in LDS stage (u&1); RS* = registers holding unit u+1 (stored now);
// RL* = registers that receive unit u+2
#define K_ITERATION(U, STAGE, RSA, RSB, RLA, RLB)                                                       \
  {                                                                                                     \
    const float* sA = smem + (STAGE) * (SA + SB);                                                       \
    const float* sB = sA + SA;                                                                          \
    float* dA = smem + (1 - (STAGE)) * (SA + SB);                                                       \
    const bool more = (U) + 1 < total_units;                                                            \
    if (!SK && has_aux && ck + 2 == nk) { PREFETCH_AUX(); }                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    READ_FRAGS(fa1, fb1, sA, sB, 1)                                                                     \
    MFMA_BLOCK(fa0, fb0)                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    READ_FRAGS(fa0, fb0, sA, sB, 2)                                                                     \
    MFMA_BLOCK(fa1, fb1)                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    if (more) {                                                                                         \
      tile_store<BM, AKM>(RSA, dA, tid);                                                                \
      tile_store<BN, BKM>(RSB, dA + SA, tid);                                                           \
    }                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    LOAD_UNIT(RSA, RSB);   /* registers just stored are free: fetch unit u+3 ... */                     \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    READ_FRAGS(fa1, fb1, sA, sB, 3)                                                                     \
    MFMA_BLOCK(fa0, fb0)                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    __syncthreads();                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    if (more) { READ_FRAGS(fa0, fb0, dA, (dA + SA), 0) }                                                \
    MFMA_BLOCK(fa1, fb1)                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    if (++ck == nk || (SK && !more)) {                                                                  \
      EPILOGUE();                                                                                       \
      aux_ready = false;                                                                                \
      ck = 0; ct += t_step;                                                                             \
      if (more) {                                                                                       \
        tile_coords(chunk0 + ct, g.tiles_m, g.tiles_n, g.band_h, tmi0, tni0);                          \
        m0 = tmi0 * BM; n0 = tni0 * BN;                                                                 \
      }                                                                                                 \
    }                                                                                                   \
  }

// Output addressing costs VALU time that the fp32 MFMAs cannot hide (on gfx950 the fp32 matrix
// instructions and the vector ALU do not overlap within a SIMD: parking a finished tile and writing
// it out element by element during the next tile's k-loop made the kernel 13 % SLOWER).  So the
// epilogue uses buffer instructions: one voffset per lane and column tile, the 16 row steps as
// scalar offsets, rows >= M and columns >= N dropped by the hardware range check -- no 64-bit
// address arithmetic, compares or exec-mask branches per element.
#define EPILOGUE()                                                                                      \
  {                                                                                                     \
    constexpr int OOB = 0x7ffffff0;                                                                     \
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, g.M * g.ldc * 4, 0x00020000); \
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(                                \
        g.aux_out, 0, g.aux_out ? g.M * g.ld_aux * 4 : 0, 0x00020000);                                  \
    const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc(                                \
        const_cast<float*>(g.aux_in), 0, g.aux_in ? g.M * g.ld_aux * 4 : 0, 0x00020000);                \
    /* loads and stores share vmcnt on gfx9: a bias load issued after a block's stores would wait for all of */ \
    /* them, so the bias of every column block is fetched before the first store                            */ \
    const bool has_bias = g.epi == UNITER_EPI_BIAS || g.epi == UNITER_EPI_BIAS_GELU || g.epi == UNITER_EPI_BIAS_GELU_D; \
    float bvv[TN];                                                                                      \
    _Pragma("unroll") for (int b = 0; b < TN; ++b) {                                                    \
      const int colb = n0 + wn * WN + b * 32 + i;                                                       \
      bvv[b] = (has_bias && colb < g.N) ? g.bias[colb] : 0.f;                                           \
    }                                                                                                   \
    _Pragma("unroll") for (int a = 0; a < TM; ++a) {                                                    \
      _Pragma("unroll") for (int b = 0; b < TN; ++b) {                                                  \
        const int col = n0 + wn * WN + b * 32 + i;                                                      \
        const bool cok = col < g.N;                                                                     \
        const int r0 = m0 + wm * WM + a * 32 + 4 * h;                                                   \
        const int voC = cok ? (r0 * g.ldc + col) * 4 : OOB;                                             \
        const int voX = cok ? (r0 * g.ld_aux + col) * 4 : OOB;                                          \
        const float bv = bvv[b];                                                                        \
        float csum = 0.f;                                                                               \
        _Pragma("unroll") for (int rr = 0; rr < 16; ++rr) {                                             \
          const int kr = (rr & 3) + 8 * (rr >> 2);                                                      \
          float v = acc[a][b][rr] + bv;                                                                 \
          if (g.epi == UNITER_EPI_BIAS_GELU) {                                                          \
            buf_st_f32(v, rsX, voX, kr * g.ld_aux * 4, 0);                   \
            v = gelu_erf(v);                                                                            \
          } else if (g.epi == UNITER_EPI_DGELU) {                                                       \
            v *= dgelu_erf(buf_ld_f32(rsI, voX, kr * g.ld_aux * 4, 0));       \
          } else if (g.epi == UNITER_EPI_ADD) {                                                         \
            v += aux_ready ? AUXR(a, b, rr) : buf_ld_f32(rsI, voX, kr * g.ld_aux * 4, 0); \
          } else if (g.epi == UNITER_EPI_MUL) {                                                         \
            v *= aux_ready ? AUXR(a, b, rr) : buf_ld_f32(rsI, voX, kr * g.ld_aux * 4, 0); \
          } else if (g.epi == UNITER_EPI_BIAS_GELU_D) {                                                 \
            float dg_;                                                                                  \
            gelu_pair_fast(v, v, dg_);                                                                  \
            buf_st_f32(dg_, rsX, voX, kr * g.ld_aux * 4, 0);                 \
          }                                                                                             \
          if (!SK && g.colsum_part) csum += (r0 + kr < g.M) ? v : 0.f;                                  \
          if (SK || g.beta) {   /* C += without reading C: no load to wait for between the stores */   \
            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, rsC, voC, kr * g.ldc * 4, 0);            \
          } else {                                                                                      \
            buf_st_f32(v, rsC, voC, kr * g.ldc * 4, 0);                      \
          }                                                                                             \
          acc[a][b][rr] = 0.f;                                                                          \
        }                                                                                               \
        if (!SK && g.colsum_part) {                                                                     \
          csum += __shfl_xor(csum, 32, 64);                                                             \
          if (h == 0 && cok && (m0 + wm * WM + a * 32) < g.M)                                           \
            g.colsum_part[(size_t)((m0 + wm * WM + a * 32) >> 5) * g.N + col] = csum;                   \
        }                                                                                               \
      }                                                                                                 \
    }                                                                                                   \
  }

  // One register set suffices: the set is written to LDS in the middle of iteration u and refilled
  // immediately, and is not needed again until the middle of iteration u+1.
  // unrolled by two so that the LDS stage of every read / write is a compile-time constant (the
  // stage-dependent address adds were ~10 VALU instructions per k-iteration, and VALU time is not
  // hidden behind fp32 MFMAs)
  int u = 0;
  for (; u + 1 < total_units; u += 2) {
    K_ITERATION(u, 0, ra1, rb1, ra1, rb1)
    K_ITERATION(u + 1, 1, ra1, rb1, ra1, rb1)
  }
  if (u < total_units) K_ITERATION(u, 0, ra1, rb1, ra1, rb1)
#undef K_ITERATION
#undef EPILOGUE
#undef PREFETCH_AUX
#undef AUXR
#undef MFMA_BLOCK
#undef READ_FRAGS
#undef LOAD_UNIT
#undef ADVANCE_LOAD_CURSOR
  stamp_end(g.stamp);
}


// ---------------------------------------------------------------------------
// Grouped weight gradients: up to four products dW_p[M_p, N_p] (+)= A_p^T B_p of ONE reduction length K (A_p [K, M_p], B_p [K, N_p],
// both k-major: the four weight gradients of an encoder layer, K = rows of the batch) as ONE persistent launch of whole-K
// 64 x 64 tiles.  Their tiles are numbered through (1728 for UNITER-base) and walked by 1024 workgroup slots like the tiles of
// one product in gemm_f32_v3_kernel -- same k-loop, same prefetch distance -- so a layer's weight gradients cost one
// prologue / last round instead of four, and 1728 tiles balance over the slots better than 576 + 576 + 144 + 432 do launch by
// launch.  No partial tiles: no float atomics (overwrite = a plain store, accumulate = one atomic add per element, one adder).
// ---------------------------------------------------------------------------
struct GemmGroup {
  const float* A[4]; const float* B[4]; float* C[4];
  int M[4], N[4];              // output shape of product p (lda = M, ldb = ldc = N)
  int tiles_m[4], tiles_n[4], band_h[4];
  int start[5];                // first tile of product p; start[n .. 4] = total
  int K, overwrite, prio;
  unsigned long long* stamp;
};

__global__ __launch_bounds__(256, 2) void gemm_f32_wgrad_group_kernel(const GemmGroup G) {
  constexpr int BM = 64, BN = 64;
  constexpr bool AKM = true, BKM = true;
  constexpr int WM = BM / 2, WN = BN / 2, TM = 1, TN = 1;
  constexpr int SA = TileSize<BM, AKM>::value, SB = TileSize<BN, BKM>::value;
  __shared__ __attribute__((aligned(16))) float smem[2 * (SA + SB)];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntiles = G.start[4];
  const int nk = (G.K + BK - 1) / BK;      // rows >= K of a k-major operand read as 0 (buffer range check)

  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  const int q = ntiles >> 3, r = ntiles & 7;
  const int chunk0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int chunk_n = q + (xcd < r ? 1 : 0);
  const int my_tiles = idx < chunk_n ? (chunk_n - idx + per_xcd - 1) / per_xcd : 0;
  const int total_units = my_tiles * nk;
  if (total_units == 0) return;
  set_wave_prio(G.prio);
  stamp_begin(G.stamp);

  // (product, tile row, tile column) of global tile T
#define LOCATE(T, P_, M0_, N0_)                                                          \
  do {                                                                                   \
    const int T_ = (T);                                                                  \
    P_ = (T_ >= G.start[1]) + (T_ >= G.start[2]) + (T_ >= G.start[3]);                   \
    int tm_, tn_;                                                                        \
    tile_coords(T_ - G.start[P_], G.tiles_m[P_], G.tiles_n[P_], G.band_h[P_], tm_, tn_); \
    M0_ = tm_ * BM; N0_ = tn_ * BN;                                                      \
  } while (0)

  // load cursor
  int lt = idx, lk = 0, lp, lm0, ln0;
  int voA[BM / 32], voB[BN / 32];
  __amdgpu_buffer_rsrc_t rsA, rsB;
  int kstepA, kstepB;
#define BIND_LOAD()                                                                      \
  do {                                                                                   \
    LOCATE(chunk0 + lt, lp, lm0, ln0);                                                   \
    rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G.A[lp]), 0, G.K * G.M[lp] * 4, 0x00020000); \
    rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G.B[lp]), 0, G.K * G.N[lp] * 4, 0x00020000); \
    kstepA = BK * G.M[lp] * 4; kstepB = BK * G.N[lp] * 4;                                \
    tile_offsets<BM, AKM>(voA, G.M[lp], lm0, tid);                                       \
    tile_offsets<BN, BKM>(voB, G.N[lp], ln0, tid);                                       \
  } while (0)
  BIND_LOAD();
  int loaded = 0;
#define LOAD_UNIT(RA, RB)                                                                \
  do {                                                                                   \
    if (loaded < total_units) {                                                          \
      tile_load_buf<BM>(RA, rsA, voA, lk * kstepA);                                      \
      tile_load_buf<BN>(RB, rsB, voB, lk * kstepB);                                      \
      ++loaded;                                                                          \
      if (++lk == nk) {                                                                  \
        lk = 0; lt += per_xcd;                                                           \
        if (loaded < total_units) BIND_LOAD();                                           \
      }                                                                                  \
    }                                                                                    \
  } while (0)

  f32x4 ra0[BM / 32], rb0[BN / 32], ra1[BM / 32], rb1[BN / 32];
  f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
  LOAD_UNIT(ra0, rb0);
  tile_store<BM, AKM>(ra0, smem, tid);
  tile_store<BN, BKM>(rb0, smem + SA, tid);
  LOAD_UNIT(ra1, rb1);
  __syncthreads();
  fa0[0] = frag_read<BM, AKM>(smem, wm * WM, 0, i, h);
  fb0[0] = frag_read<BN, BKM>(smem + SA, wn * WN, 0, i, h);

  // compute cursor
  int ct = idx, ck = 0, cp, m0, n0;
  LOCATE(chunk0 + ct, cp, m0, n0);
  f32x16 acc;
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) acc[rr] = 0.f;

#define MFMA4(FA, FB)                                                                    \
  _Pragma("unroll") for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[0][t], FB[0][t], acc, 0, 0, 0);
#define RDF(FA, FB, SAp, SBp, KB)                                                        \
  FA[0] = frag_read<BM, AKM>(SAp, wm * WM, KB, i, h);                                    \
  FB[0] = frag_read<BN, BKM>(SBp, wn * WN, KB, i, h);
#define GROUP_EPILOGUE()                                                                 \
  {                                                                                      \
    constexpr int OOB = 0x7ffffff0;                                                      \
    const int Mp = G.M[cp], Np = G.N[cp];                                                \
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(G.C[cp], 0, Mp * Np * 4, 0x00020000); \
    const int col = n0 + wn * WN + i;                                                    \
    const int r0 = m0 + wm * WM + 4 * h;                                                 \
    const int voC = col < Np ? (r0 * Np + col) * 4 : OOB;                                \
    _Pragma("unroll") for (int rr = 0; rr < 16; ++rr) {                                  \
      const int kr = (rr & 3) + 8 * (rr >> 2);                                           \
      if (G.overwrite) buf_st_f32(acc[rr], rsC, voC, kr * Np * 4, 0);                    \
      else __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc[rr], rsC, voC, kr * Np * 4, 0); \
      acc[rr] = 0.f;                                                                     \
    }                                                                                    \
  }
#define K_ITER(U, STAGE)                                                                 \
  {                                                                                      \
    const float* sA = smem + (STAGE) * (SA + SB);                                        \
    const float* sB = sA + SA;                                                           \
    float* dA = smem + (1 - (STAGE)) * (SA + SB);                                        \
    const bool more = (U) + 1 < total_units;                                             \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    RDF(fa1, fb1, sA, sB, 1)                                                             \
    MFMA4(fa0, fb0)                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    RDF(fa0, fb0, sA, sB, 2)                                                             \
    MFMA4(fa1, fb1)                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    if (more) {                                                                          \
      tile_store<BM, AKM>(ra1, dA, tid);                                                 \
      tile_store<BN, BKM>(rb1, dA + SA, tid);                                            \
    }                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    LOAD_UNIT(ra1, rb1);                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    RDF(fa1, fb1, sA, sB, 3)                                                             \
    MFMA4(fa0, fb0)                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    __syncthreads();                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    if (more) { RDF(fa0, fb0, dA, (dA + SA), 0) }                                        \
    MFMA4(fa1, fb1)                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    if (++ck == nk) {                                                                    \
      GROUP_EPILOGUE();                                                                  \
      ck = 0; ct += per_xcd;                                                             \
      if (more) LOCATE(chunk0 + ct, cp, m0, n0);                                         \
    }                                                                                    \
  }
  int u = 0;
  for (; u + 1 < total_units; u += 2) {
    K_ITER(u, 0)
    K_ITER(u + 1, 1)
  }
  if (u < total_units) K_ITER(u, 0)
#undef K_ITER
#undef GROUP_EPILOGUE
#undef RDF
#undef MFMA4
#undef LOAD_UNIT
#undef BIND_LOAD
#undef LOCATE
  stamp_end(G.stamp);
}

template <int BM, int BN, bool AKM, bool BKM, int TAG>
int launch_v3(GemmArgs g, hipStream_t st, int slots, bool allow_sk = true) {
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = (g.N + BN - 1) / BN;
  const int tiles = g.tiles_m * g.tiles_n;
  {
    const long panel = (long)BM * g.K * 4;
    long bh = (3l << 19) / (panel > 0 ? panel : 1);
    g.band_h = (int)(bh < 1 ? 1 : (bh > 16 ? 16 : bh));
    if (g.band_h > g.tiles_m) g.band_h = g.tiles_m;
  }
  int grid = tiles < slots ? (tiles + 7) / 8 * 8 : slots;
  // stream-K when the tiles fill the slots unevenly (only legal for a plain accumulating GEMM)
  static const int sk_mode = env_int("UNITER_GEMM_SK", 1);     // 0 never, 1 heuristic, 2 whenever legal
  if constexpr (TAG == 0) if (allow_sk && sk_mode && g.beta == 1 && g.epi == UNITER_EPI_NONE && !g.colsum_part && tiles >= 8) {
    const int rounds = (tiles + slots - 1) / slots;
    const bool uneven = (long)tiles * 100 < (long)rounds * slots * 88;
    const long units = (long)tiles * ((g.K + BK - 1) / BK);
    // >= 8 k-iterations per piece; a product too small for that on every slot (the image projection: 108 tiles x 64 units)
    // takes as many slots, in steps of 256, as get eight units each
    int sk_n = slots;
    if (units < 8l * slots) sk_n = (int)(units / 8 / 256) * 256;
    if ((uneven || sk_mode == 2) && sk_n >= 256) {
      slots = sk_n;
      static const int sk_slots = env_int("UNITER_WGRAD_SLOTS_F32", 0);      // A/B: pieces of the stream-K form
      if (sk_slots > 0) slots = sk_slots;
      hipLaunchKernelGGL((gemm_f32_v3_kernel<BM, BN, AKM, BKM, 0, true>), dim3(slots), dim3(256), 0, st, g);
      UCHECK_LAUNCH();
      return 0;
    }
  }
  hipLaunchKernelGGL((gemm_f32_v3_kernel<BM, BN, AKM, BKM, TAG, false>), dim3(grid), dim3(256), 0, st, g);
  UCHECK_LAUNCH();
  return 0;
}

template <bool AKM, bool BKM, int TAG>
int dispatch_cfg(int cfg, const GemmArgs& g, hipStream_t st) {
  switch (cfg) {
    case 1: return launch<128, 128, AKM, BKM, TAG>(g, st);
    case 2: return launch<64, 128, AKM, BKM, TAG>(g, st);
    case 3: return launch<128, 64, AKM, BKM, TAG>(g, st);
    case 4: return launch<64, 64, AKM, BKM, TAG>(g, st);
    case 21: case 24: case 25: {      // 25 = 24 without the stream-K form (whole-K tiles: operand panels shared k-synchronously)
      const bool fast = (g.K % BK == 0 || (AKM && BKM)) && (size_t)(AKM ? g.K : g.M) * g.lda * 4 < (1ull << 31) &&
                        (size_t)(BKM ? g.K : g.N) * g.ldb * 4 < (1ull << 31) &&
                        ((size_t)g.M + 128) * g.ldc * 4 < (1ull << 31) &&
                        ((size_t)g.M + 128) * (g.ld_aux > 0 ? g.ld_aux : 1) * 4 < (1ull << 31);
      if (!fast) return dispatch_cfg<AKM, BKM, TAG>(cfg == 21 ? 1 : 4, g, st);      // one-tile-per-workgroup kernel
      if (cfg == 21) return launch_v3<128, 128, AKM, BKM, TAG>(g, st, 512);
      return launch_v3<64, 64, AKM, BKM, TAG>(g, st, 1024, cfg != 25);
    }
    default: uniter_set_error("gemm: bad cfg %d", cfg); return UNITER_E_ARG;
  }
}

// Tile choice, from measurements on MI355X (tests/bench_gemm_gpu.py): the persistent 64x64
// kernel (4 workgroups per CU, tiles finish out of phase so epilogues overlap other tiles' MFMAs)
// wins on every model shape (M = 2624 / 576, N,K in {768, 2304, 3072, 2048}); 128x128 wins once
// there are >= ~3000 64x64 tiles (e.g. 4096^3: 137 vs 128 TFLOP/s).
int choose_cfg(int M, int N) {
  const long t64 = (long)((M + 63) / 64) * ((N + 63) / 64);
  return t64 >= 3072 ? 21 : 24;      // v3 kernels; they fall back to the v1 kernel when K % 32 != 0
}

}  // namespace

// dW_p[Mo_p, No_p] (+)= A_p^T B_p, p < n <= 4, A_p [K, Mo_p], B_p [K, No_p] fp32 (gemm_f32_wgrad_group_kernel); returns
// UNITER_E_SHAPE for shapes the grouped kernel does not take (the caller then launches the products one by one)
int gemm_f32_wgrad_group(int n, const int* Mo, const int* No, int K, const float* const* A, const float* const* B,
                         float* const* dW, int overwrite, void* stream) {
  UCHECK_ARG(n >= 1 && n <= 4 && K > 0 && Mo && No && A && B && dW, "wgrad_group_f32: bad argument");
  GemmGroup G = {};
  int total = 0;
  for (int p = 0; p < 4; ++p) {
    G.start[p] = total;
    if (p >= n) { G.A[p] = G.A[0]; G.B[p] = G.B[0]; G.C[p] = G.C[0]; G.M[p] = G.M[0]; G.N[p] = G.N[0];
                  G.tiles_m[p] = G.tiles_m[0]; G.tiles_n[p] = G.tiles_n[0]; G.band_h[p] = G.band_h[0]; continue; }
    UCHECK_ARG(Mo[p] > 0 && No[p] > 0 && A[p] && B[p] && dW[p], "wgrad_group_f32: bad product %d", p);
    const bool ok = Mo[p] % 4 == 0 && No[p] % 4 == 0 && ((uintptr_t)A[p] & 15) == 0 && ((uintptr_t)B[p] & 15) == 0 &&
                    (size_t)K * Mo[p] * 4 < (1ull << 31) && (size_t)K * No[p] * 4 < (1ull << 31) &&
                    ((size_t)Mo[p] + 128) * No[p] * 4 < (1ull << 31);
    if (!ok) { uniter_set_error("wgrad_group_f32: product %d (%d x %d, K = %d) outside the grouped kernel's range", p, Mo[p], No[p], K); return UNITER_E_SHAPE; }
    G.A[p] = A[p]; G.B[p] = B[p]; G.C[p] = dW[p]; G.M[p] = Mo[p]; G.N[p] = No[p];
    G.tiles_m[p] = (Mo[p] + 63) / 64; G.tiles_n[p] = (No[p] + 63) / 64;
    const long panel = 64l * K * 4;
    long bh = (3l << 19) / (panel > 0 ? panel : 1);
    G.band_h[p] = (int)(bh < 1 ? 1 : (bh > 16 ? 16 : bh));
    if (G.band_h[p] > G.tiles_m[p]) G.band_h[p] = G.tiles_m[p];
    total += G.tiles_m[p] * G.tiles_n[p];
  }
  G.start[4] = total;
  for (int p = n; p < 4; ++p) G.start[p] = total;
  G.K = K; G.overwrite = overwrite; G.prio = take_launch_prio(); G.stamp = take_stamp_slot();
  static const int slots_env = env_int("UNITER_WGRAD_GROUP_F32_SLOTS", 1024);
  const int slots = slots_env >= 8 ? slots_env / 8 * 8 : 1024;
  const int grid = total < slots ? (total + 7) / 8 * 8 : slots;
  hipLaunchKernelGGL(gemm_f32_wgrad_group_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, G);
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_wgrad_f32_group(int n, const int* M, const int* N, int K, const float* const* A, const float* const* B,
                                      float* const* dW, int overwrite, void* stream) {
  return gemm_f32_wgrad_group(n, M, N, K, A, B, dW, overwrite, stream);
}

// tag != 0 selects a separately named instantiation of the x @ W^T kernel (TAG template
// argument) so that one call site (the FFN-up forward GEMM) is its own row in rocprofv3 --stats.
int gemm_f32_run(int cfg, int tag, int a_kmajor, int b_kmajor, int M, int N, int K, const float* A,
                 int lda, const float* B, int ldb, float* C, int ldc, int epilogue, const float* bias,
                 const float* aux_in, float* aux_out, int ld_aux, int beta, float* colsum_part,
                 void* stream) {
  UCHECK_ARG(M > 0 && N > 0 && K > 0, "gemm: bad dims %d %d %d", M, N, K);
  UCHECK_ARG(A && B && C, "gemm: null operand");
  UCHECK_ARG(epilogue >= 0 && epilogue <= UNITER_EPI_MUL, "gemm: bad epilogue %d", epilogue);
  UCHECK_ARG(!(epilogue == UNITER_EPI_BIAS || epilogue == UNITER_EPI_BIAS_GELU || epilogue == UNITER_EPI_BIAS_GELU_D) || bias,
             "gemm: epilogue needs bias");
  UCHECK_ARG(!(epilogue == UNITER_EPI_DGELU || epilogue == UNITER_EPI_ADD || epilogue == UNITER_EPI_MUL) || aux_in,
             "gemm: epilogue needs aux_in");
  // 16-byte vector loads along the contiguous dimension
  if (!a_kmajor) UCHECK_SHAPE(K % 4 == 0 && lda % 4 == 0, "gemm: K/lda must be multiples of 4 (A k-contiguous)");
  else           UCHECK_SHAPE(M % 4 == 0 && lda % 4 == 0, "gemm: M/lda must be multiples of 4 (A k-major)");
  if (!b_kmajor) UCHECK_SHAPE(K % 4 == 0 && ldb % 4 == 0, "gemm: K/ldb must be multiples of 4 (B k-contiguous)");
  else           UCHECK_SHAPE(N % 4 == 0 && ldb % 4 == 0, "gemm: N/ldb must be multiples of 4 (B k-major)");
  UCHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "gemm: operands must be 16-byte aligned");
  GemmArgs g;
  g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.epi = epilogue; g.bias = bias; g.aux_in = aux_in; g.aux_out = aux_out; g.ld_aux = ld_aux;
  g.beta = beta; g.tiles_m = g.tiles_n = 0; g.band_h = 1; g.colsum_part = colsum_part;
  g.stamp = take_stamp_slot();
  g.prio = take_launch_prio();
  if (cfg == 0) cfg = choose_cfg(M, N);
  if (colsum_part) {
    const bool fast = (K % BK == 0 || (a_kmajor && b_kmajor)) && (size_t)(a_kmajor ? K : M) * lda * 4 < (1ull << 31) &&
                      (size_t)(b_kmajor ? K : N) * ldb * 4 < (1ull << 31);
    UCHECK_SHAPE(fast && (cfg == 21 || cfg == 24 || cfg == 25) && beta == 0, "gemm: fused column sums need the v3 kernel (K %% 32 == 0)");
  }
  hipStream_t st = (hipStream_t)stream;
  if (!a_kmajor && !b_kmajor)
    return tag ? dispatch_cfg<false, false, 1>(cfg, g, st) : dispatch_cfg<false, false, 0>(cfg, g, st);
  if (!a_kmajor && b_kmajor) return dispatch_cfg<false, true, 0>(cfg, g, st);
  if (a_kmajor && b_kmajor) return dispatch_cfg<true, true, 0>(cfg, g, st);
  return dispatch_cfg<true, false, 0>(cfg, g, st);
}

extern "C" int uniter_gemm_f32_cfg(int cfg, int a_kmajor, int b_kmajor, int M, int N, int K,
                                   const float* A, int lda, const float* B, int ldb, float* C,
                                   int ldc, int epilogue, const float* bias, const float* aux_in,
                                   float* aux_out, int ld_aux, int beta, void* stream) {
  return gemm_f32_run(cfg, 0, a_kmajor, b_kmajor, M, N, K, A, lda, B, ldb, C, ldc, epilogue, bias, aux_in,
                      aux_out, ld_aux, beta, nullptr, stream);
}

extern "C" int uniter_gemm_f32(int a_kmajor, int b_kmajor, int M, int N, int K, const float* A,
                               int lda, const float* B, int ldb, float* C, int ldc, int epilogue,
                               const float* bias, const float* aux_in, float* aux_out, int ld_aux,
                               int beta, void* stream) {
  return uniter_gemm_f32_cfg(0, a_kmajor, b_kmajor, M, N, K, A, lda, B, ldb, C, ldc, epilogue, bias,
                             aux_in, aux_out, ld_aux, beta, stream);
}
