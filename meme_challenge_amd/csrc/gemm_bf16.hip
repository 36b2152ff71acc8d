// Mixed-precision GEMM for gfx950: fp32 operands in HBM, converted to bf16 while they are
// staged into LDS, multiplied on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, fp32
// accumulate), fp32 epilogue and output.
//
// This is the "bf16 MFMA used only for the dense QKV / FFN GEMMs" mode of BASELINE config 3: all
// tensors stay fp32 in memory (master weights, activations, gradients, LayerNorm / softmax / loss
// arithmetic), only the contraction inputs are rounded to bf16 (round-to-nearest-even,
// v_cvt_pk_bf16_f32).  The bf16 pipe is 16x the fp32 one, so unlike gemm_f32 this kernel is bound
// by operand delivery (L2 -> LDS), not by the matrix pipe: tiles are large (128 x 128 x 64) and
// the schedule is v3's (persistent workgroups, banded L2-aware tile order, branch-free
// buffer loads with a full-iteration prefetch distance, LDS written mid-iteration).
//
// LDS image: every operand tile is [rows][64 k] bf16 with a 144-byte row stride (conflict-free
// ds_read_b128 of 8 consecutive k per lane).  Operands whose k index is the slow memory index
// (dgrad's W, wgrad's dY and X) are TRANSPOSED ON THE WAY IN: a thread loads 2 k-rows x 4
// consecutive columns, packs (k, k+1) pairs and issues 4 ds_write_b32 -- so the MFMA loop is
// identical for all three layouts and no transposed LDS read is needed.
#include <stdlib.h>
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

struct GemmArgsB {
  int M, N, K;
  const void* A; int lda;      // fp32 (converted while staged) or, in the RES kernels, bf16 as stored
  const void* B; int ldb;
  float* C; int ldc;
  unsigned short* Cb; int ldcb;   // optional bf16 copy of C (the next GEMM's operand)
  int epi;
  const float* bias;
  const float* aux_in;
  float* aux_out;
  int ld_aux;
  int beta;
  int tiles_m, tiles_n, band_h;
  float* colsum_part;
  unsigned long long* stamp;    // optional {first start, last end} slot (common.h)
};

__device__ __forceinline__ float buf_ld_f32(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_st_f32(float v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}

constexpr int BKB = 64;            // k-tile depth
constexpr int LDB16 = 72;          // LDS row stride in bf16 elements (144 B)

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}

__device__ __forceinline__ void tile_coords_b(int t, int tiles_m, int tiles_n, int band_h, int& tm, int& tn) {
  const int full = band_h * tiles_n;
  const int band = t / full;
  const int rem = t - band * full;
  const int bh = min(band_h, tiles_m - band * band_h);
  tn = rem / bh;
  tm = band * band_h + (rem - tn * bh);
}

// per-thread byte offsets of the float4 loads of one tile (k0 = 0)
template <int R, bool KM>
__device__ __forceinline__ void tile_offsets_b(int (&voff)[R / 16], int ld, int row0, int tid) {
  if constexpr (!KM) {
    const int c4 = tid & 15, rr = tid >> 4;          // 16 threads cover one row's 64 k
#pragma unroll
    for (int p = 0; p < R / 16; ++p) voff[p] = ((row0 + rr + 16 * p) * ld + c4 * 4) * 4;
  } else {
    // item = (4 columns, k-pair): id = tid + 256*pass; columns fastest over 4 lanes, then 32 k-pairs
#pragma unroll
    for (int p = 0; p < R / 32; ++p) {
      const int id = tid + 256 * p;
      const int c4 = ((id >> 7) << 2) | (id & 3), kp = (id >> 2) & 31;
      voff[2 * p] = ((2 * kp) * ld + row0 + c4 * 4) * 4;
      voff[2 * p + 1] = voff[2 * p] + ld * 4;
    }
  }
}

template <int R>
__device__ __forceinline__ void tile_load_b(f32x4 (&reg)[R / 16], __amdgpu_buffer_rsrc_t rsrc,
                                            const int (&voff)[R / 16], int soff) {
#pragma unroll
  for (int p = 0; p < R / 16; ++p)
    reg[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[p], soff, 0));
}

// convert + write one staged tile into its LDS image [R][LDB16] bf16
template <int R, bool KM>
__device__ __forceinline__ void tile_store_b(const f32x4 (&reg)[R / 16], unsigned short* s, int tid) {
  if constexpr (!KM) {
    const int c4 = tid & 15, rr = tid >> 4;
#pragma unroll
    for (int p = 0; p < R / 16; ++p) {
      u32x2_t v = {pack_bf16(reg[p][0], reg[p][1]), pack_bf16(reg[p][2], reg[p][3])};
      *reinterpret_cast<u32x2_t*>(s + (rr + 16 * p) * LDB16 + c4 * 4) = v;
    }
  } else {
#pragma unroll
    for (int p = 0; p < R / 32; ++p) {
      const int id = tid + 256 * p;
      const int c4 = ((id >> 7) << 2) | (id & 3), kp = (id >> 2) & 31;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        *reinterpret_cast<unsigned*>(s + (c4 * 4 + e) * LDB16 + 2 * kp) = pack_bf16(reg[2 * p][e], reg[2 * p + 1][e]);
    }
  }
}

// operand fragment of the 32-row sub-tile at r0 for k16-step ks: 8 consecutive k of row r0 + (lane&31)
__device__ __forceinline__ bf16x8 frag_read_b(const unsigned short* s, int r0, int ks, int i, int h) {
  return *reinterpret_cast<const bf16x8*>(s + (r0 + i) * LDB16 + ks * 16 + 8 * h);
}

// SK = stream-K for C += A.B (weight gradients), as in gemm_f32.hip: equal contiguous pieces of an
// XCD's unit sequence per workgroup, partial tiles added with buffer_atomic_add_f32.
// ---- operand staging: global -> registers -> LDS image [R][LDB16] bf16 -------------------------
// RES = false: fp32 in memory, rounded to bf16 on the way (the functions above).
// RES = true : bf16 in memory (resident activations / weight mirror): half the bytes, no conversion.
//   k-contiguous rows: 8 lanes x 16 B per 64-k row, written as whole 16-B slots;
//   k-major: a lane takes 8 consecutive columns of a k-pair of rows and writes 8 (k, k+1) words.
//   The 16-B slots of a row are XOR-swizzled by (row >> 3) & 3 so that the four lanes that differ
//   only in their column group do not share a bank on those word writes.
template <int R, bool KM, bool RES> struct Stager;

template <int R, bool KM> struct Stager<R, KM, false> {
  static constexpr int IMG = R * LDB16;
  int voff[R / 16];
  f32x4 reg[R / 16];
  static __device__ __forceinline__ int kstep(int ld) { return (KM ? BKB * ld : BKB) * 4; }
  static __device__ __forceinline__ size_t bytes(int rows, int ld) { return (size_t)rows * ld * 4; }
  __device__ __forceinline__ void offsets(int ld, int row0, int tid) { tile_offsets_b<R, KM>(voff, ld, row0, tid); }
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int soff) { tile_load_b<R>(reg, rs, voff, soff); }
  __device__ __forceinline__ void store(unsigned short* s, int tid) const { tile_store_b<R, KM>(reg, s, tid); }
  static __device__ __forceinline__ bf16x8 frag(const unsigned short* s, int r0, int ks, int i, int h) {
    return frag_read_b(s, r0, ks, i, h);
  }
};

template <int R> struct Stager<R, false, true> {
  static constexpr int NP = R / 32;
  static constexpr int IMG = R * LDB16;
  int voff[NP];
  u32x4_t reg[NP];
  static __device__ __forceinline__ int kstep(int) { return BKB * 2; }
  static __device__ __forceinline__ size_t bytes(int rows, int ld) { return (size_t)rows * ld * 2; }
  __device__ __forceinline__ void offsets(int ld, int row0, int tid) {
    const int c8 = tid & 7, rr = tid >> 3;
#pragma unroll
    for (int p = 0; p < NP; ++p) voff[p] = ((row0 + rr + 32 * p) * ld + c8 * 8) * 2;
  }
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int soff) {
#pragma unroll
    for (int p = 0; p < NP; ++p) reg[p] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[p], soff, 0);
  }
  __device__ __forceinline__ void store(unsigned short* s, int tid) const {
    const int c8 = tid & 7, rr = tid >> 3;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int row = rr + 32 * p;
      *reinterpret_cast<u32x4_t*>(s + row * LDB16 + ((c8 ^ ((row >> 3) & 3)) << 3)) = reg[p];
    }
  }
  static __device__ __forceinline__ bf16x8 frag(const unsigned short* s, int r0, int ks, int i, int h) {
    const int row = r0 + i;
    return *reinterpret_cast<const bf16x8*>(s + row * LDB16 + (((2 * ks + h) ^ ((row >> 3) & 3)) << 3));
  }
};

// k-major resident operand ([k][n] in memory, n contiguous: dgrad's weight, both wgrad operands).  The
// tile goes to LDS AS IT IS -- [64 k][R n] rows of 16-byte pieces, no register transposition, no
// per-element LDS writes -- and the MFMA operand (8 consecutive k of one n per lane) is gathered by
// ds_read_b64_tr_b16: per 16-lane group the instruction reads a 4 (k) x 16 (n) block and hands lane i
// column i.  Lane 4q+p of a group addresses row k0+q, columns n0+4p..+3; two reads (k0, k0+4) make
// one bf16x8 fragment.  Row stride 192 B (R = 64) / 320 B (R = 128): the four rows a 32-lane half
// touches fall into disjoint 16-bank windows.
template <int R> struct Stager<R, true, true> {
  static constexpr int NP = R / 32;                   // 16-B pieces per thread: 64 rows x R/8 pieces / 256 threads
  static constexpr int LDK = R == 64 ? 96 : 160;      // LDS row stride in bf16 elements
  static constexpr int IMG = BKB * LDK;
  int voff[NP];
  u32x4_t reg[NP];
  static __device__ __forceinline__ int kstep(int ld) { return BKB * ld * 2; }
  static __device__ __forceinline__ size_t bytes(int rows, int ld) { return (size_t)rows * ld * 2; }
  __device__ __forceinline__ void offsets(int ld, int row0, int tid) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int id = tid + 256 * p;
      const int c8 = id % (R / 8), k = id / (R / 8);
      voff[p] = (k * ld + row0 + c8 * 8) * 2;
    }
  }
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int soff) {
#pragma unroll
    for (int p = 0; p < NP; ++p) reg[p] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[p], soff, 0);
  }
  __device__ __forceinline__ void store(unsigned short* s, int tid) const {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int id = tid + 256 * p;
      const int c8 = id % (R / 8), k = id / (R / 8);
      *reinterpret_cast<u32x4_t*>(s + k * LDK + c8 * 8) = reg[p];
    }
  }
  static __device__ __forceinline__ bf16x8 frag(const unsigned short* s, int r0, int ks, int i, int h) {
    typedef short short4v __attribute__((ext_vector_type(4)));
    typedef short4v __attribute__((address_space(3))) * lds_s4;
    const int lane16 = i & 15, q = lane16 >> 2, p = lane16 & 3;
    const unsigned short* a0 = s + (ks * 16 + 8 * h + q) * LDK + r0 + (i & 16) + 4 * p;
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(a0));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(a0 + 4 * LDK));
    typedef short short8v __attribute__((ext_vector_type(8)));
    const short8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  }
};

// workgroups per CU the register allocator has to leave room for: 64x64 tiles run 3 deep (<= 168 VGPRs, and
// the k-major resident LDS images are 43 - 49 KB), everything larger 2 deep
template <int BM, int BN, bool KM, bool RES> constexpr int wgs_per_cu() {
  return (BM == 64 && BN == 64) ? 3 : 2;
}

template <int BM, int BN, bool AKM, bool BKM, int TAG, bool SK, bool RES = false>
__global__ __launch_bounds__(256, (wgs_per_cu<BM, BN, AKM || BKM, RES>())) void gemm_bf16_kernel(const GemmArgsB g) {
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  // aux operand of an epilogue prefetched per 32x32 block (16 registers) -- not in the 128x128 convert-in-flight
  // kernels, whose fp32 staging registers leave no room for it
  constexpr bool AUXPF = RES || BM * BN < 128 * 128;
  constexpr int SA = Stager<BM, AKM, RES>::IMG, SB = Stager<BN, BKM, RES>::IMG;      // bf16 elements per LDS image
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * (SA + SB)];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntiles = g.tiles_m * g.tiles_n;
  const int nk = (g.K + BKB - 1) / BKB;     // ragged last k-tile: k-major operands only (rows >= K read as 0)

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
  stamp_begin(g.stamp);
  const int t_step = SK ? 1 : per_xcd;

  using StA = Stager<BM, AKM, RES>;
  using StB = Stager<BN, BKM, RES>;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(g.A), 0, (int)StA::bytes(AKM ? g.K : g.M, g.lda), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(g.B), 0, (int)StB::bytes(BKM ? g.K : g.N, g.ldb), 0x00020000);
  const int kstepA = StA::kstep(g.lda), kstepB = StB::kstep(g.ldb);

  int lt = t_first, lk = k_first;
  StA sa;
  StB sb;
  {
    int tmi, tni;
    tile_coords_b(chunk0 + lt, g.tiles_m, g.tiles_n, g.band_h, tmi, tni);
    sa.offsets(g.lda, tmi * BM, tid);
    sb.offsets(g.ldb, tni * BN, tid);
  }
  int loaded = 0;
#define LOAD_UNIT()                                                                      \
  do {                                                                                   \
    if (loaded < total_units) {                                                          \
      sa.load(rsA, lk * kstepA);                                                         \
      sb.load(rsB, lk * kstepB);                                                         \
      ++loaded;                                                                          \
      if (++lk == nk) {                                                                  \
        lk = 0; lt += t_step;                                                            \
        if (loaded < total_units) {                                                      \
          int tmi_, tni_;                                                                \
          tile_coords_b(chunk0 + lt, g.tiles_m, g.tiles_n, g.band_h, tmi_, tni_);        \
          sa.offsets(g.lda, tmi_ * BM, tid);                                             \
          sb.offsets(g.ldb, tni_ * BN, tid);                                             \
        }                                                                                \
      }                                                                                  \
    }                                                                                    \
  } while (0)

  bf16x8 fa0[TM], fb0[TN], fa1[TM], fb1[TN];

  LOAD_UNIT();
  sa.store(smem, tid);
  sb.store(smem + SA, tid);
  LOAD_UNIT();
  __syncthreads();
#pragma unroll
  for (int a = 0; a < TM; ++a) fa0[a] = StA::frag(smem, wm * WM + a * 32, 0, i, h);
#pragma unroll
  for (int b = 0; b < TN; ++b) fb0[b] = StB::frag(smem + SA, wn * WN + b * 32, 0, i, h);

  int ct = t_first, ck = k_first;
  int tmi0, tni0;
  tile_coords_b(chunk0 + ct, g.tiles_m, g.tiles_n, g.band_h, tmi0, tni0);
  int m0 = tmi0 * BM, n0 = tni0 * BN;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) acc[a][b][rr] = 0.f;

#define MFMA_BLOCK(FA, FB)                                                                              \
  _Pragma("unroll") for (int a = 0; a < TM; ++a)                                                        \
  _Pragma("unroll") for (int b = 0; b < TN; ++b)                                                        \
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[a], FB[b], acc[a][b], 0, 0, 0);
#define READ_FRAGS(FA, FB, SAp, SBp, KS)                                                                \
  _Pragma("unroll") for (int a = 0; a < TM; ++a) FA[a] = StA::frag(SAp, wm * WM + a * 32, KS, i, h);    \
  _Pragma("unroll") for (int b = 0; b < TN; ++b) FB[b] = StB::frag(SBp, wn * WN + b * 32, KS, i, h);

// Epilogue ordering: on gfx9 loads and stores share vmcnt and the compiler has to assume they retire out of
// order, so ANY load result needed while stores are in flight becomes s_waitcnt vmcnt(0) -- a full drain of
// the stores (measured: 13 - 28 us per 128x128 tile when bias / aux / beta loads sat between the stores).
// Hence: every load of the tile (bias, the aux operand) is issued before its first store, and C += is a
// buffer_atomic_add (no read).
#define AUXV(rr) (AUXPF ? auxv[rr] : buf_ld_f32(rsI, voX, (((rr) & 3) + 8 * ((rr) >> 2)) * stX))
#define EPILOGUE_B()                                                                                    \
  {                                                                                                     \
    constexpr int OOB = 0x7ffffff0;                                                                     \
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, g.C ? g.M * g.ldc * 4 : 0, 0x00020000); \
    const __amdgpu_buffer_rsrc_t rsCb = __builtin_amdgcn_make_buffer_rsrc(g.Cb, 0, g.Cb ? g.M * g.ldcb * 2 : 0, 0x00020000); \
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(                                \
        g.aux_out, 0, g.aux_out ? g.M * g.ld_aux * 4 : 0, 0x00020000);                                  \
    const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc(                                \
        const_cast<float*>(g.aux_in), 0, g.aux_in ? g.M * g.ld_aux * 4 : 0, 0x00020000);                \
    const bool has_bias = g.epi == UNITER_EPI_BIAS || g.epi == UNITER_EPI_BIAS_GELU || g.epi == UNITER_EPI_BIAS_GELU_D; \
    const bool has_aux = g.epi == UNITER_EPI_DGELU || g.epi == UNITER_EPI_ADD || g.epi == UNITER_EPI_MUL; \
    float bvv[TN];                                                                                      \
    _Pragma("unroll") for (int b = 0; b < TN; ++b) {                                                    \
      const int col = n0 + wn * WN + b * 32 + i;                                                        \
      bvv[b] = (has_bias && col < g.N) ? g.bias[col] : 0.f;                                             \
    }                                                                                                   \
    /* rows of a 32x32 accumulator block: kr(rr) = (rr & 3) + 8 (rr >> 2); the addresses walk down them in a */ \
    /* VGPR (+1 row, or +5 after every fourth) -- 16 scalar row offsets per stream do not fit the SGPR file  */ \
    const int stC = g.ldc * 4, stX = g.ld_aux * 4, stCb = g.ldcb * 2;                                   \
    _Pragma("unroll") for (int a = 0; a < TM; ++a) {                                                    \
      _Pragma("unroll") for (int b = 0; b < TN; ++b) {                                                  \
        const int col = n0 + wn * WN + b * 32 + i;                                                      \
        const bool cok = col < g.N;                                                                     \
        const int r0 = m0 + wm * WM + a * 32 + 4 * h;                                                   \
        const int voC = cok ? (r0 * g.ldc + col) * 4 : OOB;                                             \
        const int voX = cok ? (r0 * g.ld_aux + col) * 4 : OOB;                                          \
        const int voCb = cok ? (r0 * g.ldcb + col) * 2 : OOB;                                           \
        /* the aux operand of this 32x32 block (all of a 64x64 tile's per-wave output; a 128x128 tile */  \
        /* waits for the previous block's stores before each of its other three) */                     \
        float auxv[16];                                                                                 \
        if (AUXPF && has_aux) {                                                                         \
          int p_ = voX;                                                                                 \
          _Pragma("unroll") for (int rr = 0; rr < 16; ++rr) {                                           \
            auxv[rr] = buf_ld_f32(rsI, p_, 0);                                                          \
            p_ += ((rr & 3) == 3 ? 5 : 1) * stX;                                                        \
          }                                                                                             \
        }                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        const float bv = bvv[b];                                                                        \
        /* the epilogue kind and the output set are uniform: branch once per 32x32 block, not per element */ \
        f32x16& v = acc[a][b];                                                                          \
        _Pragma("unroll") for (int rr = 0; rr < 16; ++rr) v[rr] += bv;                                  \
        if (g.epi == UNITER_EPI_BIAS_GELU_D) {                                                          \
          int p_ = voX;                                                                                 \
          _Pragma("unroll") for (int rr = 0; rr < 16; ++rr) {                                           \
            float dg_;                                                                                  \
            float x_ = v[rr], y_;                                                                       \
            gelu_pair_fast(x_, y_, dg_);                                                                \
            v[rr] = y_;                                                                                 \
            buf_st_f32(dg_, rsX, p_, 0);                                                                \
            p_ += ((rr & 3) == 3 ? 5 : 1) * stX;                                                        \
            if ((rr & 3) == 3) __builtin_amdgcn_sched_barrier(0);                                       \
          }                                                                                             \
        } else if (g.epi == UNITER_EPI_MUL) {                                                           \
          _Pragma("unroll") for (int rr = 0; rr < 16; ++rr) v[rr] *= AUXV(rr);                          \
        } else if (g.epi == UNITER_EPI_ADD) {                                                           \
          _Pragma("unroll") for (int rr = 0; rr < 16; ++rr) v[rr] += AUXV(rr);                          \
        } else if (g.epi == UNITER_EPI_BIAS_GELU) {                                                     \
          int p_ = voX;                                                                                 \
          _Pragma("unroll") for (int rr = 0; rr < 16; ++rr) {                                           \
            buf_st_f32(v[rr], rsX, p_, 0);                                                              \
            p_ += ((rr & 3) == 3 ? 5 : 1) * stX;                                                        \
            v[rr] = gelu_erf(v[rr]);                                                                    \
          }                                                                                             \
        } else if (g.epi == UNITER_EPI_DGELU) {                                                         \
          _Pragma("unroll") for (int rr = 0; rr < 16; ++rr) v[rr] *= dgelu_erf(AUXV(rr));               \
        }                                                                                               \
        if (!SK && g.colsum_part) {                                                                     \
          float csum = 0.f;                                                                             \
          _Pragma("unroll") for (int rr = 0; rr < 16; ++rr)                                             \
            csum += (r0 + (rr & 3) + 8 * (rr >> 2) < g.M) ? v[rr] : 0.f;                                \
          csum += __shfl_xor(csum, 32, 64);                                                             \
          if (h == 0 && cok && (m0 + wm * WM + a * 32) < g.M)                                           \
            g.colsum_part[(size_t)((m0 + wm * WM + a * 32) >> 5) * g.N + col] = csum;                   \
        }                                                                                               \
        if (SK || g.beta) {                                                                             \
          int p_ = voC;                                                                                 \
          _Pragma("unroll") for (int rr = 0; rr < 16; ++rr) {                                           \
            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v[rr], rsC, p_, 0, 0);                      \
            p_ += ((rr & 3) == 3 ? 5 : 1) * stC;                                                        \
          }                                                                                             \
        } else {                                                                                        \
          if (g.C) {                                                                                    \
            int p_ = voC;                                                                               \
            _Pragma("unroll") for (int rr = 0; rr < 16; ++rr) {                                         \
              buf_st_f32(v[rr], rsC, p_, 0);                                                            \
              p_ += ((rr & 3) == 3 ? 5 : 1) * stC;                                                      \
            }                                                                                           \
          }                                                                                             \
          if (g.Cb) {                                                                                   \
            int p_ = voCb;                                                                              \
            _Pragma("unroll") for (int rr = 0; rr < 16; ++rr) {                                         \
              __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (__bf16)v[rr]), rsCb, p_, 0, 0); \
              p_ += ((rr & 3) == 3 ? 5 : 1) * stCb;                                                     \
            }                                                                                           \
          }                                                                                             \
        }                                                                                               \
        _Pragma("unroll") for (int rr = 0; rr < 16; ++rr) v[rr] = 0.f;                                  \
      }                                                                                                 \
    }                                                                                                   \
  }

// one k-iteration with a compile-time LDS stage (the loop is unrolled by two: no address VALU)
#define K_ITERATION_B(U, STAGE)                                                                         \
  {                                                                                                     \
    const unsigned short* sA = smem + (STAGE) * (SA + SB);                                              \
    const unsigned short* sB = sA + SA;                                                                 \
    unsigned short* dA = smem + (1 - (STAGE)) * (SA + SB);                                              \
    const bool more = (U) + 1 < total_units;                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    READ_FRAGS(fa1, fb1, sA, sB, 1)                                                                     \
    MFMA_BLOCK(fa0, fb0)                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    READ_FRAGS(fa0, fb0, sA, sB, 2)                                                                     \
    MFMA_BLOCK(fa1, fb1)                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    if (more) {                                                                                         \
      sa.store(dA, tid);                                                                                \
      sb.store(dA + SA, tid);                                                                           \
    }                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    LOAD_UNIT();                                                                                        \
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
      EPILOGUE_B();                                                                                     \
      ck = 0; ct += t_step;                                                                             \
      if (more) {                                                                                       \
        tile_coords_b(chunk0 + ct, g.tiles_m, g.tiles_n, g.band_h, tmi0, tni0);                         \
        m0 = tmi0 * BM; n0 = tni0 * BN;                                                                 \
      }                                                                                                 \
    }                                                                                                   \
  }

  int u = 0;
  for (; u + 1 < total_units; u += 2) {
    K_ITERATION_B(u, 0)
    K_ITERATION_B(u + 1, 1)
  }
  if (u < total_units) K_ITERATION_B(u, 0)
#undef K_ITERATION_B
#undef EPILOGUE_B
#undef AUXV
#undef MFMA_BLOCK
#undef READ_FRAGS
#undef LOAD_UNIT
  stamp_end(g.stamp);
}

template <int BM, int BN, bool AKM, bool BKM, bool RES = false>
int launch_b(GemmArgsB g, hipStream_t st, int slots) {
  if (BM == 64 && BN == 64) slots = 768;     // three 64x64 workgroups per CU (wgs_per_cu)
  static const int sk_slots = [] { const char* e = getenv("UNITER_WGRAD_SLOTS"); return e ? atoi(e) : 0; }();   // A/B: stream-K pieces
  // weight gradients run on the side stream BESIDE the input-gradient chain: 512 pieces (two per CU, ~96 KB of LDS)
  // leave room for a 64-KB workgroup of the other stream on every CU, 768 (three per CU) fill the LDS and the two
  // streams take turns instead (bf16 step 2898 -> 2944 samples/s; 384: 2913, 256: 2798)
  if (g.beta == 1 && AKM && BKM && BM == 64 && BN == 64) slots = sk_slots > 0 ? sk_slots : 512;
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = (g.N + BN - 1) / BN;
  const int tiles = g.tiles_m * g.tiles_n;
  const long panel = (long)BM * g.K * 4;
  long bh = (3l << 19) / (panel > 0 ? panel : 1);
  g.band_h = (int)(bh < 1 ? 1 : (bh > 16 ? 16 : bh));
  if (g.band_h > g.tiles_m) g.band_h = g.tiles_m;
  const int grid = tiles < slots ? (tiles + 7) / 8 * 8 : slots;
  if (g.beta == 1 && g.epi == UNITER_EPI_NONE && !g.colsum_part && tiles >= 8) {
    const int rounds = (tiles + slots - 1) / slots;
    const bool uneven = (long)tiles * 100 < (long)rounds * slots * 88;
    const long units = (long)tiles * ((g.K + BKB - 1) / BKB);
    if (uneven && units >= 8l * slots) {
      hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, AKM, BKM, 0, true, RES>), dim3(slots), dim3(256), 0, st, g);
      UCHECK_LAUNCH();
      return 0;
    }
  }
  hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, AKM, BKM, 0, false, RES>), dim3(grid), dim3(256), 0, st, g);
  UCHECK_LAUNCH();
  return 0;
}

template <bool AKM, bool BKM>
int dispatch_b(int cfg, const GemmArgsB& g, hipStream_t st) {
  switch (cfg) {
    case 1: return launch_b<128, 128, AKM, BKM>(g, st, 512);
    case 2: return launch_b<64, 128, AKM, BKM>(g, st, 512);
    case 3: return launch_b<128, 64, AKM, BKM>(g, st, 512);
    case 4: return launch_b<64, 64, AKM, BKM>(g, st, 1024);
    default: uniter_set_error("gemm_bf16: bad cfg %d", cfg); return UNITER_E_ARG;
  }
}

template <bool AKM, bool BKM>
int dispatch_r(int cfg, const GemmArgsB& g, hipStream_t st) {
  switch (cfg) {
    case 1: return launch_b<128, 128, AKM, BKM, true>(g, st, 512);
    case 4: return launch_b<64, 64, AKM, BKM, true>(g, st, 1024);
    default: uniter_set_error("gemm_bf16res: bad cfg %d (1 or 4)", cfg); return UNITER_E_ARG;
  }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                                        size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
    u32x2_t o = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
    reinterpret_cast<u32x2_t*>(dst)[i] = o;
  }
}

}  // namespace

// All-bf16 operands (resident activations / weight mirror), fp32 accumulate, fp32 and / or bf16 output.
int gemm_bf16res_run(int cfg, int a_kmajor, int b_kmajor, int M, int N, int K, const void* A, int lda, const void* B,
                     int ldb, float* C, int ldc, void* Cb, int ldcb, int epilogue, const float* bias,
                     const float* aux_in, float* aux_out, int ld_aux, int beta, float* colsum_part, void* stream) {
  UCHECK_ARG(M > 0 && N > 0 && K > 0 && A && B && (C || Cb), "gemm_bf16res: bad argument");
  UCHECK_ARG(epilogue >= 0 && epilogue <= UNITER_EPI_MUL, "gemm_bf16res: bad epilogue %d", epilogue);
  UCHECK_ARG(!(a_kmajor && !b_kmajor), "gemm_bf16res: layout (A k-major, B k-contiguous) is not built");
  UCHECK_ARG(!beta || (C && !Cb), "gemm_bf16res: C += needs the fp32 output and has no bf16 copy (the sum is formed by atomics)");
  UCHECK_SHAPE((K % BKB == 0 || (a_kmajor && b_kmajor)) && lda % 8 == 0 && ldb % 8 == 0 &&
               (a_kmajor ? M % 8 == 0 : true) && (b_kmajor ? N % 8 == 0 : true) &&
               ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0,
               "gemm_bf16res: K %% 64, leading dimensions %% 8 and 16-byte aligned operands required (M=%d N=%d K=%d)", M, N, K);
  UCHECK_SHAPE((size_t)(a_kmajor ? K : M) * lda * 2 < (1ull << 31) && (size_t)(b_kmajor ? K : N) * ldb * 2 < (1ull << 31) &&
               ((size_t)M + 128) * (ldc > 0 ? ldc : 1) * 4 < (1ull << 31) &&
               ((size_t)M + 128) * (ld_aux > 0 ? ld_aux : 1) * 4 < (1ull << 31), "gemm_bf16res: operand beyond 31-bit offsets");
  GemmArgsB g;
  g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.Cb = (unsigned short*)Cb; g.ldcb = ldcb;
  g.epi = epilogue; g.bias = bias; g.aux_in = aux_in; g.aux_out = aux_out; g.ld_aux = ld_aux; g.beta = beta;
  g.tiles_m = g.tiles_n = 0; g.band_h = 1; g.colsum_part = colsum_part;
  g.stamp = take_stamp_slot();
  (void)take_launch_prio();
  if (cfg == 0) {
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    // measured (tests/tools/gemm_lab.py at M = 1424 / 2624 / 5248, profiles/r01_gemm_bf16_resident_tiles.txt): a
    // full wave of 128x128 tiles moves ~1.45x the flops per unit time of a full wave of 64x64 ones (half the
    // L2 -> LDS traffic per flop), so what decides is how well each tiling fills its last round of persistent
    // slots (512 resp. 768).  The weight gradients (both operands k-major, stream-K) stay on 64x64.
    const long t64 = (long)((M + 63) / 64) * ((N + 63) / 64);
    const double e128 = 1.45 * (double)t128 / (512.0 * ((t128 + 511) / 512));
    const double e64 = (double)t64 / (768.0 * ((t64 + 767) / 768));
    cfg = (t128 >= 1024 || (!a_kmajor && e128 > e64)) ? 1 : 4;
  }
  hipStream_t st = (hipStream_t)stream;
  if (!a_kmajor && !b_kmajor) return dispatch_r<false, false>(cfg, g, st);
  if (!a_kmajor && b_kmajor) return dispatch_r<false, true>(cfg, g, st);
  return dispatch_r<true, true>(cfg, g, st);
}

extern "C" int uniter_gemm_bf16res_cfg(int cfg, int a_kmajor, int b_kmajor, int M, int N, int K, const void* A,
                                       int lda, const void* B, int ldb, float* C, int ldc, void* C_bf16, int ldcb,
                                       int epilogue, const float* bias, const float* aux_in, float* aux_out,
                                       int ld_aux, int beta, void* stream) {
  UCHECK_ARG(!(epilogue == UNITER_EPI_BIAS || epilogue == UNITER_EPI_BIAS_GELU || epilogue == UNITER_EPI_BIAS_GELU_D) || bias,
             "gemm_bf16res: epilogue needs bias");
  UCHECK_ARG(!(epilogue == UNITER_EPI_DGELU || epilogue == UNITER_EPI_ADD || epilogue == UNITER_EPI_MUL) || aux_in,
             "gemm_bf16res: epilogue needs aux_in");
  return gemm_bf16res_run(cfg, a_kmajor, b_kmajor, M, N, K, A, lda, B, ldb, C, ldc, C_bf16, ldcb, epilogue, bias,
                          aux_in, aux_out, ld_aux, beta, nullptr, stream);
}

extern "C" int uniter_cast_bf16(const float* src, void* dst, size_t n, void* stream) {
  UCHECK_ARG(src && dst && n % 4 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 7) == 0,
             "cast_bf16: n must be a multiple of 4 and the buffers aligned");
  if (n == 0) return 0;
  const size_t n4 = n / 4;
  const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (unsigned short*)dst, n4);
  UCHECK_LAUNCH();
  return 0;
}

int gemm_f32_run(int cfg, int tag, int a_kmajor, int b_kmajor, int M, int N, int K, const float* A, int lda,
                 const float* B, int ldb, float* C, int ldc, int epilogue, const float* bias,
                 const float* aux_in, float* aux_out, int ld_aux, int beta, float* colsum_part, void* stream);

// Same contract as gemm_f32_run, contraction on the bf16 matrix pipe.  Shapes the bf16 kernel does
// not cover (K % 64 != 0, offsets beyond 31 bits) run on the exact fp32 kernel instead.
int gemm_bf16_run(int cfg, int a_kmajor, int b_kmajor, int M, int N, int K, const float* A, int lda,
                  const float* B, int ldb, float* C, int ldc, int epilogue, const float* bias,
                  const float* aux_in, float* aux_out, int ld_aux, int beta, float* colsum_part, void* stream) {
  UCHECK_ARG(M > 0 && N > 0 && K > 0 && A && B && C, "gemm_bf16: bad argument");
  const bool ok = (K % BKB == 0 || (a_kmajor && b_kmajor)) && lda % 4 == 0 && ldb % 4 == 0 &&
                  ((size_t)M + 128) * ldc * 4 < (1ull << 31) && ((size_t)M + 128) * (ld_aux > 0 ? ld_aux : 1) * 4 < (1ull << 31) &&
                  (a_kmajor ? M % 4 == 0 : true) && (b_kmajor ? N % 4 == 0 : true) &&
                  (size_t)(a_kmajor ? K : M) * lda * 4 < (1ull << 31) &&
                  (size_t)(b_kmajor ? K : N) * ldb * 4 < (1ull << 31) &&
                  ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0;
  if (!ok)
    return gemm_f32_run(0, 0, a_kmajor, b_kmajor, M, N, K, A, lda, B, ldb, C, ldc, epilogue, bias, aux_in, aux_out,
                        ld_aux, beta, colsum_part, stream);
  UCHECK_ARG(epilogue >= 0 && epilogue <= UNITER_EPI_MUL, "gemm_bf16: bad epilogue %d", epilogue);
  GemmArgsB g;
  g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.Cb = nullptr; g.ldcb = 0;
  g.epi = epilogue; g.bias = bias; g.aux_in = aux_in; g.aux_out = aux_out; g.ld_aux = ld_aux; g.beta = beta;
  g.tiles_m = g.tiles_n = 0; g.band_h = 1; g.colsum_part = colsum_part;
  g.stamp = take_stamp_slot();
  (void)take_launch_prio();
  if (cfg == 0) {
    // operand delivery bound: the biggest tile that still fills the chip
    // measured on MI355X (tests/tools/gemm_bf16_exp.py): 128x128 only pays once it fills the chip
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    cfg = t128 >= 448 ? 1 : 4;
  }
  hipStream_t st = (hipStream_t)stream;
  if (!a_kmajor && !b_kmajor) return dispatch_b<false, false>(cfg, g, st);
  if (!a_kmajor && b_kmajor) return dispatch_b<false, true>(cfg, g, st);
  if (a_kmajor && b_kmajor) return dispatch_b<true, true>(cfg, g, st);
  return dispatch_b<true, false>(cfg, g, st);
}

extern "C" int uniter_gemm_bf16_cfg(int cfg, int a_kmajor, int b_kmajor, int M, int N, int K, const float* A,
                                    int lda, const float* B, int ldb, float* C, int ldc, int epilogue,
                                    const float* bias, const float* aux_in, float* aux_out, int ld_aux, int beta,
                                    void* stream) {
  UCHECK_ARG(!(epilogue == UNITER_EPI_BIAS || epilogue == UNITER_EPI_BIAS_GELU || epilogue == UNITER_EPI_BIAS_GELU_D) || bias,
             "gemm_bf16: epilogue needs bias");
  UCHECK_ARG(!(epilogue == UNITER_EPI_DGELU || epilogue == UNITER_EPI_ADD || epilogue == UNITER_EPI_MUL) || aux_in,
             "gemm_bf16: epilogue needs aux_in");
  return gemm_bf16_run(cfg, a_kmajor, b_kmajor, M, N, K, A, lda, B, ldb, C, ldc, epilogue, bias, aux_in, aux_out,
                       ld_aux, beta, nullptr, stream);
}
