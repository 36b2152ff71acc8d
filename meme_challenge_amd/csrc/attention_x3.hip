// Fused self-attention of the fp32x3 mode: every product of BertSelfAttention (model/layer.py:80-100) and of its autograd runs on
// the bf16 matrix pipe with fp32 results -- each operand value as three bf16 pieces (x = x1 + x2 + x3 exactly), six
// v_mfma_f32_16x16x32_bf16 products per block (gemm_split3.hip has the arithmetic) -- instead of on the fp32 MFMAs of
// attention_f32.hip, which run at 1/16 of the bf16 rate.  Scores, softmax, dropout, LSE and deltas are fp32 as there.
//
// Work decomposition (L <= 192): one workgroup per (batch, head), ONE WAVE PER 16 ROWS -- 12 waves at L = 164..192, three per SIMD,
// no partial results to merge.  With the 16x16x32 MFMA the accumulator of
//     S^T[key][query] = K . Q^T      (lane = query l & 15, registers = keys 4 (l >> 4) + r of a 16-key block)
// is, two 16-key blocks together, the B operand of the 32-deep step of
//     O^T[d][query] = V^T . P^T      (element j of lane group g: key 16 (j >> 2) + 4 g + (j & 3))
// with no lane movement; the A operand V^T comes out of the ROW-MAJOR image of V by ds_read_b64_tr_b16 in exactly that key order
// (two reads of 4 keys x 16 d).  So each operand has ONE LDS image per piece -- [row][64 d] bf16, 128-byte rows, 16-byte chunk c of
// row r at c ^ (r & 6): row reads (ds_read_b128) and transposed reads both conflict-free (searched over the linear swizzles
// with the bank rules of MI355X_MICROARCH.md) -- and K, V (forward, dQ pass) or Q, dO (dK / dV pass) fit as 2 x 3 images of 24 KB in the 160-KB LDS.
//
// Backward, one launch, NO hand-over through memory: pass 1 (wave = 16 queries, all keys) recomputes S^T and dP^T = V . dO^T and
// accumulates dQ^T = K^T . dS^T; pass 2 (wave = 16 keys, all queries; Q and dO restaged over K and V) recomputes
// S = Q . K^T and dP = dO . V^T in the other orientation and accumulates dV^T = dO^T . Pd and dK^T = Q^T . dS.  The recomputation
// costs 48 of the 168 MFMAs per 32 x 32 tile pair -- on the bf16 pipe less than the 2 x 28 MB of scratch traffic it replaces.
//
// The kernels are templates over the number of pieces: NP = 3 is the fp32x3 mode (uniter_attn_x3_fwd / _bwd), NP = 1 the bf16
// mode in the same decomposition (uniter_attn_b16x_fwd / _bwd: operands rounded to bf16 where they become MFMA operands, one
// product per block, Q / K / V optionally bf16 in memory; its forward keeps attention_bf16.hip's two online-softmax ranges so that
// probabilities are rounded where the bf16 oracle rounds them).
//
// Dropout: the keep flags are READ (uniter_attn_keep_bits_gen draws them ahead of the forward pass); both passes of the backward
// read the same words.  Layouts as attention_f32.hip: qkv [rows, 3H], ctx / dctx [rows, H], lse / delta [B, nh, L], x3 outputs
// [rows][3][ld].  head_dim == 64.
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;
typedef unsigned char u8;

constexpr int D = 64;
constexpr int MAXL = 192;
constexpr int ROWB = 128;                 // bytes per image row (64 bf16)
constexpr int IMG = MAXL * ROWB;          // bytes per piece image, whatever the length: piece and block offsets are instruction immediates
constexpr float NEG_INF = -__builtin_huge_valf();

struct Args {
  const void* qkv;         // [rows, 3H] fp32 (bf16 in the bf16 mode's QB16 kernels)
  const float* mask;       // [B, L] or NULL (varlen)
  const int* cu;           // [B+1] or NULL
  float* ctx; u16* ctx_x3; // forward outputs (backward: ctx is an input)
  float* lse;              // [B, nh, L]
  const float* dctx;       // [dctx_slabs][rows, H]: k-pieces of the attention-output input gradient, summed while they are read
  int dctx_slabs; size_t dctx_slab_stride;
  float* dqkv; u16* dqkv_x3;
  float* bias_part;        // optional [B, 3H] per-sample column sums of dqkv
  float* delta;            // [B, nh, L]
  const u16* keep_bits;    // [B*nh, L, Lr/32, 2] keep flags (attention_f32.hip, attn_keep_bits_kernel), NULL without dropout
  float drop_scale;        // 1 / (1 - p)
  int B, L, nh, H, prio;
  float scale;
};

struct Span { int row0, Lb, nb; };
__device__ __forceinline__ Span span_of(const Args& a, int b) {
  Span s;
  s.row0 = a.cu ? a.cu[b] : b * a.L;
  s.Lb = a.cu ? a.cu[b + 1] - s.row0 : a.L;
  s.nb = (s.Lb + 31) >> 5;
  return s;
}

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v = {(__bf16)lo, (__bf16)hi};      // v_cvt_pk_bf16_f32: round to nearest even
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bflo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bfhi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
// the three bf16 pieces of two fp32 values (exact: every residual is representable)
__device__ __forceinline__ void split2(float x0, float x1, unsigned& w1, unsigned& w2, unsigned& w3) {
  w1 = pack2(x0, x1);
  float r0 = x0 - bflo(w1), r1 = x1 - bfhi(w1);
  w2 = pack2(r0, r1);
  r0 -= bflo(w2); r1 -= bfhi(w2);
  w3 = pack2(r0, r1);
}
// pieces of 8 values as MFMA fragments (NP = 3: the exact split; NP = 1: rounded to bf16, the bf16 mode's operands)
template <int NP>
__device__ __forceinline__ void split8(const f32x4& x, const f32x4& y, bf16x8 (&f)[NP]) {
  if constexpr (NP == 1) {
    f[0] = __builtin_bit_cast(bf16x8, u32x4{pack2(x[0], x[1]), pack2(x[2], x[3]), pack2(y[0], y[1]), pack2(y[2], y[3])});
    return;
  }
  unsigned w[3][4];
  split2(x[0], x[1], w[0][0], w[1][0], w[2][0]);
  split2(x[2], x[3], w[0][1], w[1][1], w[2][1]);
  split2(y[0], y[1], w[0][2], w[1][2], w[2][2]);
  split2(y[2], y[3], w[0][3], w[1][3], w[2][3]);
#pragma unroll
  for (int p = 0; p < NP; ++p) f[p] = __builtin_bit_cast(bf16x8, u32x4{w[p][0], w[p][1], w[p][2], w[p][3]});
}

// chunk swizzle of the images (header)
__device__ __forceinline__ int swz(int r) { return r & 6; }

// One [L, 64] operand on its way into its NP piece images: this thread's 16-byte pieces are all loaded before anything is
// written (blockDim.x = 4 Lr threads).  Rows >= L are zero.  B16: the source is bf16 and is copied as stored (NP == 1).
template <int NP, bool B16>
struct Stage {
  f32x4 v[B16 ? 2 : 4];
  __device__ __forceinline__ void load(const void* __restrict__ base_, int ld, int L, int Lr, int tid, int nthr) {
    if constexpr (B16) {
      const u16* base = static_cast<const u16*>(base_);
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int idx = tid + it * nthr, r = idx >> 3, c8 = idx & 7;
        v[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (idx < Lr * 8 && r < L) v[it] = *reinterpret_cast<const f32x4*>(base + (size_t)r * ld + c8 * 8);
      }
    } else {
      const float* base = static_cast<const float*>(base_);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int idx = tid + it * nthr, r = idx >> 4, c4 = idx & 15;
        v[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (idx < Lr * 16 && r < L) v[it] = *reinterpret_cast<const f32x4*>(base + (size_t)r * ld + c4 * 4);
      }
    }
  }
  __device__ __forceinline__ void store(u8* img, int Lr, int tid, int nthr) const {
    if constexpr (B16) {
      static_assert(NP == 1, "a bf16 source has one piece");
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int idx = tid + it * nthr, r = idx >> 3, c8 = idx & 7;
        if (idx < Lr * 8) *reinterpret_cast<f32x4*>(img + r * ROWB + 16 * (c8 ^ swz(r))) = v[it];
      }
    } else {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int idx = tid + it * nthr, r = idx >> 4, c4 = idx & 15;
        if (idx < Lr * 16) {
          u8* p = img + r * ROWB + 16 * ((c4 >> 1) ^ swz(r)) + 8 * (c4 & 1);
          if constexpr (NP == 1) {
            *reinterpret_cast<u32x2*>(p) = u32x2{pack2(v[it][0], v[it][1]), pack2(v[it][2], v[it][3])};
          } else {
            unsigned a1, a2, a3, b1, b2, b3;
            split2(v[it][0], v[it][1], a1, a2, a3);
            split2(v[it][2], v[it][3], b1, b2, b3);
            *reinterpret_cast<u32x2*>(p) = u32x2{a1, b1};
            *reinterpret_cast<u32x2*>(p + IMG) = u32x2{a2, b2};
            *reinterpret_cast<u32x2*>(p + 2 * IMG) = u32x2{a3, b3};
          }
        }
      }
    }
  }
};

#ifdef UNITER_X3_LAB
// Measurement build only (LAB & 8: "what if Q, K, V arrived as pieces"): the operand is read as [rows][3][ld] bf16 -- three 8-byte
// words per four values, no vector work on the way into the images.  Run over a buffer 1.5 x the fp32 one (tests/tools/attn_x3_lab.py);
// the values are meaningless, the instruction and byte counts are those of the real thing.
struct StagePieces {
  u32x2 w[4][3];
  __device__ __forceinline__ void load(const u16* __restrict__ base, int ld, int L, int Lr, int tid, int nthr) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = tid + it * nthr, r = idx >> 4, c4 = idx & 15;
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        w[it][p] = u32x2{0u, 0u};
        if (idx < Lr * 16 && r < L) w[it][p] = *reinterpret_cast<const u32x2*>(base + ((size_t)r * 3 + p) * ld + c4 * 4);
      }
    }
  }
  __device__ __forceinline__ void store(u8* img, int Lr, int tid, int nthr) const {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = tid + it * nthr, r = idx >> 4, c4 = idx & 15;
      if (idx < Lr * 16) {
        u8* q = img + r * ROWB + 16 * ((c4 >> 1) ^ swz(r)) + 8 * (c4 & 1);
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x2*>(q + p * IMG) = w[it][p];
      }
    }
  }
};
__device__ __forceinline__ void row_frags_pieces(bf16x8 (&f)[2][3], const u16* __restrict__ row, int ld, bool valid, int g) {
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      f32x4 x = {0.f, 0.f, 0.f, 0.f};
      if (valid) x = *reinterpret_cast<const f32x4*>(row + (size_t)p * ld + 32 * s + 8 * g);
      f[s][p] = __builtin_bit_cast(bf16x8, x);
    }
}
#endif

// B operand of a product that sums over d, from the lane's own row: step s, lane group g -> d = 32 s + 8 g .. + 7
template <int NP, bool B16>
__device__ __forceinline__ void row_frags(bf16x8 (&f)[2][NP], const void* __restrict__ row_, bool valid, int g) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    if constexpr (B16) {
      f32x4 x = {0.f, 0.f, 0.f, 0.f};
      if (valid) x = *reinterpret_cast<const f32x4*>(static_cast<const u16*>(row_) + 32 * s + 8 * g);
      f[s][0] = __builtin_bit_cast(bf16x8, x);
    } else {
      const float* row = static_cast<const float*>(row_);
      f32x4 x = {0.f, 0.f, 0.f, 0.f}, y = x;
      if (valid) {
        x = *reinterpret_cast<const f32x4*>(row + 32 * s + 8 * g);
        y = *reinterpret_cast<const f32x4*>(row + 32 * s + 8 * g + 4);
      }
      split8<NP>(x, y, f[s]);
    }
  }
}

// per-lane byte offsets into an image: row reads (A operand, k = d) and transposed reads (A operand, k = image rows)
template <int NP>
struct LaneOffs {
  int row[2];      // step s: row i, chunk 4 s + g
  int tr[4];       // d-block db: row 4 g + (i >> 2), columns 16 db + 4 (i & 3) .. + 3
  int row2[2], tr2[4];   // the same into the second operand's images (+ NP IMG): registers of their own, so that piece and block
                         // offsets stay within the 16-bit immediates of the LDS instructions for both operands
  __device__ __forceinline__ void init(int i, int g) {
    const int fi = swz(i);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      row[s] = i * ROWB + 16 * ((4 * s + g) ^ fi);
      row2[s] = row[s] + NP * IMG;
      asm volatile("" : "+v"(row2[s]));
    }
    const int tr_row = 4 * g + (i >> 2), p4 = i & 3, ft = swz(tr_row);
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      tr[db] = tr_row * ROWB + 16 * ((2 * db + (p4 >> 1)) ^ ft) + 8 * (p4 & 1);
      tr2[db] = tr[db] + NP * IMG;
      asm volatile("" : "+v"(tr2[db]));
    }
  }
};

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
// rows r0 .. r0 + 15 of piece image `img`: the lane's row, 8 consecutive d of step s
__device__ __forceinline__ bf16x8 frag_row(const u8* img, int r0, int off) {
  return *reinterpret_cast<const bf16x8*>(img + r0 * ROWB + off);
}
// image rows r0 .. r0 + 31 as the 32-deep k of an A operand whose rows are 16 image columns: accumulator k order
// (element j of lane group g: row r0 + 16 (j >> 2) + 4 g + (j & 3)).  Needs EXEC all ones (wave-uniform control flow around it).
__device__ __forceinline__ bf16x8 frag_tr(const u8* img, int r0, int off) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(img + r0 * ROWB + off));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(img + (r0 + 16) * ROWB + off));
  return __builtin_bit_cast(bf16x8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
}

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
// c += a . b for operands in three pieces: the six products above 2^-24 |a b|
template <int NP, int LAB = 0>
__device__ __forceinline__ void mac(f32x4& c, const bf16x8 (&a)[NP], const bf16x8 (&b)[NP]) {
  if constexpr (NP == 1) {      // bf16 operands: one product
    c = MFMA16(a[0], b[0], c);
  } else if constexpr (LAB & 4) {      // measurement build only: the fragments are read, the products are not issued
    asm volatile("" ::"v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(b[0]), "v"(b[1]), "v"(b[2]));
  } else {
    c = MFMA16(a[0], b[0], c);
    c = MFMA16(a[0], b[1], c);
    c = MFMA16(a[1], b[0], c);
    c = MFMA16(a[1], b[1], c);
    c = MFMA16(a[0], b[2], c);
    c = MFMA16(a[2], b[0], c);
  }
}

// sums over the four lane groups (the lanes that share l & 15)
__device__ __forceinline__ float groups_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float groups_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

// one output row (lane's query or key): acc[db][r] = element d = 16 db + 4 g + r; fp32 and / or the three pieces.  The pieces
// leave two d-blocks at a time: v_permlane16_swap hands lane group 1's (3's) values of block db to group 0 (2) and group 0's
// (2's) values of block db + 1 to group 1 (3), so every lane stores 8 consecutive d -- one 16-byte store per piece instead of two
// of 8 bytes (the store tail of an attention kernel is issue-bound, MI355X_MICROARCH.md).  Every lane of the wave must call it
// (the swap crosses lanes); `valid` masks the stores only.
template <int NP>
__device__ __forceinline__ void store_row(bool valid, float* __restrict__ row, u16* __restrict__ row_x3, int ps,
                                          const f32x4 (&acc)[4], float mul, int g) {
  f32x4 v[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    v[db] = f32x4{acc[db][0] * mul, acc[db][1] * mul, acc[db][2] * mul, acc[db][3] * mul};
    if (valid && row) *reinterpret_cast<f32x4*>(row + 16 * db + 4 * g) = v[db];
  }
  if (row_x3) {
#pragma unroll
    for (int db = 0; db < 4; db += 2) {
      unsigned wa[3][2], wb[3][2];
      if constexpr (NP == 1) {      // the bf16 copy of the bf16 mode
        wa[0][0] = pack2(v[db][0], v[db][1]); wa[0][1] = pack2(v[db][2], v[db][3]);
        wb[0][0] = pack2(v[db + 1][0], v[db + 1][1]); wb[0][1] = pack2(v[db + 1][2], v[db + 1][3]);
      } else {
        split2(v[db][0], v[db][1], wa[0][0], wa[1][0], wa[2][0]);
        split2(v[db][2], v[db][3], wa[0][1], wa[1][1], wa[2][1]);
        split2(v[db + 1][0], v[db + 1][1], wb[0][0], wb[1][0], wb[2][0]);
        split2(v[db + 1][2], v[db + 1][3], wb[0][1], wb[1][1], wb[2][1]);
      }
      u16* dst = row_x3 + (db + (g & 1)) * 16 + 8 * (g >> 1);
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const auto r0 = __builtin_amdgcn_permlane16_swap(wa[p][0], wb[p][0], false, false);
        const auto r1 = __builtin_amdgcn_permlane16_swap(wa[p][1], wb[p][1], false, false);
        if (valid) *reinterpret_cast<u32x4*>(dst + p * ps) = u32x4{r0[0], r1[0], r0[1], r1[1]};
      }
    }
  }
}

// column sums of an output block over the wave's valid rows -> red[0..63] (LDS, pre-zeroed)
__device__ __forceinline__ void acc_colsum(float* red, const f32x4 (&acc)[4], float mul, bool valid, int i, int g) {
#pragma unroll
  for (int db = 0; db < 4; ++db) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = valid ? acc[db][r] * mul : 0.f;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (i == 0) atomicAdd(red + 16 * db + 4 * g + r, v);
    }
  }
}

extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

__device__ __forceinline__ void stage_mask(float* mb, const Args& a, int b, int Lb, int Lr, int tid, int nthr) {
  for (int k = tid; k < Lr; k += nthr)
    mb[k] = k < Lb ? (a.mask ? (1.0f - a.mask[(size_t)b * a.L + k]) * -10000.0f : 0.f) : NEG_INF;
}

// keep flags of a 2 x 16-row accumulator pair whose LANE is the query: one word (attention_f32.hip: bit 4 gg + t of word
// (bh, q, key block, half) = key 32 kb + 8 gg + 4 half + t): keys 16 b2 + 4 g + r -> word g & 1, bit 4 (2 b2 + (g >> 1)) + r
__device__ __forceinline__ float keep_mult(unsigned word, int b2, int g, int r, float scale) {
  return ((word >> (4 * (2 * b2 + (g >> 1)) + r)) & 1u) ? scale : 0.f;
}

// ---------------------------------------------------------------- forward ---
// LDS: K pieces | V pieces | mask bias
template <int NP, bool QB16>
__global__ __launch_bounds__(768) void attn_x3_fwd_kernel(const Args a, int Lr) {
  set_wave_prio(a.prio);
  u8* Ki = smem_raw;
  u8* Vi = smem_raw + NP * IMG;
  float* mb = reinterpret_cast<float*>(smem_raw + 2 * NP * IMG);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
  const int i = lane & 15, g = lane >> 4;
  const int nblk = Lr >> 5;
  const int bh = blockIdx.x, b = bh / a.nh, head = bh - b * a.nh;
  const Span sp = span_of(a, b);
  const int Lb = sp.Lb, ld = 3 * a.H;
  typedef typename std::conditional<QB16, u16, float>::type src_t;
  const src_t* base = static_cast<const src_t*>(a.qkv) + (size_t)sp.row0 * ld + head * D;
  const int q = wave * 16 + i;
  const bool vq = q < Lb;
  bf16x8 qf[2][NP];
  {
    Stage<NP, QB16> sk, sv;
    sk.load(base + a.H, ld, Lb, Lr, tid, nthr);
    sv.load(base + 2 * a.H, ld, Lb, Lr, tid, nthr);
    row_frags<NP, QB16>(qf, base + (size_t)q * ld, vq, g);
    stage_mask(mb, a, b, Lb, Lr, tid, nthr);
    sk.store(Ki, Lr, tid, nthr);
    sv.store(Vi, Lr, tid, nthr);
  }
  __syncthreads();
  LaneOffs<NP> lo;
  lo.init(i, g);
  const int nk = wave * 16 < Lb ? sp.nb : 0;           // wave-uniform
  const bool drop = a.keep_bits != nullptr;
  // the keep words of this lane's query for every key chunk, ahead of the loop (one memory latency instead of one per chunk)
  unsigned words[MAXL / 32];
#pragma unroll
  for (int kc = 0; kc < MAXL / 32; ++kc) {
    words[kc] = 0xffffu;
    if (drop && kc < nblk) words[kc] = a.keep_bits[(((size_t)bh * a.L + (vq ? q : 0)) * nblk + kc) * 2 + (g & 1)];
  }
  f32x4 o[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) o[db] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = NEG_INF, l_run = 0.f;
  // NP == 1 (bf16 mode): the key range as TWO online softmaxes (first ceil(nb / 2) chunks, the rest) merged at the end -- the
  // blocking of attention_bf16.hip's forward kernel, which the bf16 oracle restates (oracle/uniter_oracle.py,
  // _online_softmax_pv_b16): a probability is rounded to bf16 relative to ITS range's running maximum, and a rounding
  // modelled differently decorrelates every later bf16 rounding within a few layers
  const int hb = NP == 1 ? (sp.nb + 1) >> 1 : nk;
  f32x4 o_a[NP == 1 ? 4 : 1];
  float m_a = NEG_INF, l_a = 0.f;
  for (int kc = 0; kc < nk; ++kc) {
    if constexpr (NP == 1) {
      if (kc == hb) {
#pragma unroll
        for (int db = 0; db < 4; ++db) { o_a[db] = o[db]; o[db] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        m_a = m_run; l_a = l_run; m_run = NEG_INF; l_run = 0.f;
      }
    }
    const int k0 = kc * 32;
    unsigned word = words[0];
#pragma unroll
    for (int c = 1; c < MAXL / 32; ++c) word = kc == c ? words[c] : word;
    f32x4 s[2];
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2) {                     // S^T[key][query] = K . Q^T
      s[b2] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        bf16x8 ka[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) ka[p] = frag_row(Ki + p * IMG, k0 + 16 * b2, lo.row[st]);
        mac<NP>(s[b2], ka, qf[st]);
      }
    }
    float mx = NEG_INF;
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2) {
      const f32x4 bias = *reinterpret_cast<const f32x4*>(mb + k0 + 16 * b2 + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[b2][r] = s[b2][r] * a.scale + bias[r];
        mx = fmaxf(mx, s[b2][r]);
      }
    }
    mx = groups_max(mx);
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);
    float ls = 0.f;
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
      for (int r = 0; r < 4; ++r) { s[b2][r] = __expf(s[b2][r] - m_new); ls += s[b2][r]; }
    l_run = l_run * alpha + ls;
    m_run = m_new;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[db][r] *= alpha;
    if (drop) {
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[b2][r] *= keep_mult(word, b2, g, r, a.drop_scale);
    }
    bf16x8 pb[NP];
    split8<NP>(s[0], s[1], pb);
#pragma unroll
    for (int db = 0; db < 4; ++db) {                     // O^T[d][query] += V^T . Pd^T
      bf16x8 va[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) va[p] = frag_tr(Ki + p * IMG, k0, lo.tr2[db]);
      mac<NP>(o[db], va, pb);
    }
  }
  if constexpr (NP == 1) {
    if (nk > hb) {
      const float m_new = fmaxf(m_a, m_run);
      const float wa = __expf(m_a - m_new), wb = m_run == NEG_INF ? 0.f : __expf(m_run - m_new);
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[db][r] = o_a[db][r] * wa + o[db][r] * wb;
      l_run = l_a * wa + l_run * wb;
      m_run = m_new;
    }
  }
  const float l_tot = groups_sum(l_run);
  {
    const size_t row = (size_t)sp.row0 + (vq ? q : 0);
    store_row<NP>(vq, a.ctx ? a.ctx + row * a.H + head * D : nullptr, a.ctx_x3 ? a.ctx_x3 + row * NP * a.H + head * D : nullptr, a.H, o,
              1.0f / l_tot, g);
    if (vq && g == 0 && a.lse) a.lse[(size_t)bh * a.L + q] = m_run + __logf(l_tot);
  }
}

// --------------------------------------------------------------- backward ---
// LDS: operand pieces 1 | operand pieces 2 | mask bias | lse | delta | column sums | keep words
// LAB (measurement builds, -DUNITER_X3_LAB + UNITER_ATTN_X3_LAB=bits): 1 = no pass-1 loop, 2 = no pass-2 loop, 4 = no MFMAs,
// 8 = Q, K, V read as pieces (StagePieces)
template <int NP, bool QB16, int LAB>
__global__ __launch_bounds__(768) void attn_x3_bwd_kernel(const Args a, int Lr) {
  set_wave_prio(a.prio);
  u8* I1 = smem_raw;                    // pass 1: K, pass 2: Q
  u8* I2 = smem_raw + NP * IMG;         // pass 1: V, pass 2: dO
  float* mb = reinterpret_cast<float*>(smem_raw + 2 * NP * IMG);
  float* lse_s = mb + Lr;
  float* delta_s = lse_s + Lr;
  float* red = delta_s + Lr;
  u16* kb_s = reinterpret_cast<u16*>(red + 192);      // the head's keep words [query][key chunk][2] (both passes read them)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
  const int i = lane & 15, g = lane >> 4;
  const int nblk = Lr >> 5;
  const int bh = blockIdx.x, b = bh / a.nh, head = bh - b * a.nh;
  const Span sp = span_of(a, b);
  const int Lb = sp.Lb, ld = 3 * a.H;
  typedef typename std::conditional<QB16, u16, float>::type src_t;
  const src_t* base = static_cast<const src_t*>(a.qkv) + (size_t)sp.row0 * ld + head * D;
  const float* dobase = a.dctx + (size_t)sp.row0 * a.H + head * D;
  const int rw = wave * 16 + i;          // this lane's row: a query in pass 1, a key in pass 2
  const bool vr = rw < Lb;
  const bool drop = a.keep_bits != nullptr;
  const int nloop = wave * 16 < Lb ? sp.nb : 0;         // wave-uniform: 32-row chunks of the other index
  LaneOffs<NP> lo;
  lo.init(i, g);

  // ---- pass 1: dQ (and delta) for queries rw; keys from the images of K and V
  bf16x8 qf[2][NP], dof[2][NP];
  {
    float delta = 0.f;
    {
#ifdef UNITER_X3_LAB
      typedef typename std::conditional<(LAB & 8) != 0, StagePieces, Stage<NP, QB16>>::type stage_t;
      stage_t sk, sv;
      if constexpr (LAB & 8) {
        const u16* b16 = static_cast<const u16*>(a.qkv) + (size_t)sp.row0 * 3 * ld + head * D;
        sk.load(b16 + a.H, ld, Lb, Lr, tid, nthr);
        sv.load(b16 + 2 * a.H, ld, Lb, Lr, tid, nthr);
        row_frags_pieces(qf, b16 + (size_t)rw * 3 * ld, ld, vr, g);
      } else {
        sk.load(base + a.H, ld, Lb, Lr, tid, nthr);
        sv.load(base + 2 * a.H, ld, Lb, Lr, tid, nthr);
        row_frags<NP, QB16>(qf, base + (size_t)rw * ld, vr, g);
      }
#else
      Stage<NP, QB16> sk, sv;
      sk.load(base + a.H, ld, Lb, Lr, tid, nthr);
      sv.load(base + 2 * a.H, ld, Lb, Lr, tid, nthr);
      row_frags<NP, QB16>(qf, base + (size_t)rw * ld, vr, g);
#endif
      // this lane's 16 of the 64 values of its dO and O rows (d = 32 s + 8 g + 0..7): the k-slots of the dO operand
      const float* dorow = dobase + (size_t)rw * a.H;
      const float* orow = a.ctx + ((size_t)sp.row0 + rw) * a.H + head * D;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        f32x4 x = {0.f, 0.f, 0.f, 0.f}, y = x, ox = x, oy = x;
        if (vr) {
          x = *reinterpret_cast<const f32x4*>(dorow + 32 * s + 8 * g);
          y = *reinterpret_cast<const f32x4*>(dorow + 32 * s + 8 * g + 4);
          for (int sl = 1; sl < a.dctx_slabs; ++sl) {
            x += *reinterpret_cast<const f32x4*>(dorow + sl * a.dctx_slab_stride + 32 * s + 8 * g);
            y += *reinterpret_cast<const f32x4*>(dorow + sl * a.dctx_slab_stride + 32 * s + 8 * g + 4);
          }
          ox = *reinterpret_cast<const f32x4*>(orow + 32 * s + 8 * g);
          oy = *reinterpret_cast<const f32x4*>(orow + 32 * s + 8 * g + 4);
        }
        delta += x[0] * ox[0] + x[1] * ox[1] + x[2] * ox[2] + x[3] * ox[3] + y[0] * oy[0] + y[1] * oy[1] + y[2] * oy[2] + y[3] * oy[3];
        split8<NP>(x, y, dof[s]);
      }
      stage_mask(mb, a, b, Lb, Lr, tid, nthr);
      for (int t = tid; t < 192; t += nthr) red[t] = 0.f;
      if (drop) {
        const unsigned* src = reinterpret_cast<const unsigned*>(a.keep_bits + (size_t)bh * a.L * nblk * 2);
        for (int t = tid; t < Lb * nblk; t += nthr) reinterpret_cast<unsigned*>(kb_s)[t] = src[t];
      }
      sk.store(I1, Lr, tid, nthr);
      sv.store(I2, Lr, tid, nthr);
    }
    delta = groups_sum(delta);
    const float lse = vr ? a.lse[(size_t)bh * a.L + rw] : -NEG_INF;      // +inf: every probability of a padded query is 0
    if (g == 0) {
      if (vr) a.delta[(size_t)bh * a.L + rw] = delta;
      lse_s[rw] = lse;
      delta_s[rw] = vr ? delta : 0.f;
    }
    __syncthreads();

    const u16* kbits = kb_s + (vr ? rw : 0) * nblk * 2 + (g & 1);
    f32x4 dq[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) dq[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kc = 0; kc < ((LAB & 1) ? 0 : nloop); ++kc) {
      const int k0 = kc * 32;
      unsigned word = 0xffffu;
      if (drop) word = kbits[kc * 2];
      f32x4 s[2], dp[2];
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) {
        s[b2] = f32x4{0.f, 0.f, 0.f, 0.f};
        dp[b2] = s[b2];
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          bf16x8 ka[NP], va[NP];
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            ka[p] = frag_row(I1 + p * IMG, k0 + 16 * b2, lo.row[st]);
            va[p] = frag_row(I1 + p * IMG, k0 + 16 * b2, lo.row2[st]);
          }
          mac<NP, LAB>(s[b2], ka, qf[st]);                       // S^T[key][query] = K . Q^T
          mac<NP, LAB>(dp[b2], va, dof[st]);                     // dP^T[key][query] = V . dO^T
        }
      }
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) {
        const f32x4 bias = *reinterpret_cast<const f32x4*>(mb + k0 + 16 * b2 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __expf(s[b2][r] * a.scale + bias[r] - lse);
          const float m = drop ? keep_mult(word, b2, g, r, a.drop_scale) : 1.f;
          s[b2][r] = p * (dp[b2][r] * m - delta);      // x scale (a power of two) when dQ leaves
        }
      }
      bf16x8 db3[NP];
      split8<NP>(s[0], s[1], db3);
#pragma unroll
      for (int db = 0; db < 4; ++db) {                   // dQ^T[d][query] += K^T . dS^T
        bf16x8 kt[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) kt[p] = frag_tr(I1 + p * IMG, k0, lo.tr[db]);
        mac<NP, LAB>(dq[db], kt, db3);
      }
    }
    {
      const size_t row = (size_t)sp.row0 + (vr ? rw : 0);
      store_row<NP>(vr, a.dqkv ? a.dqkv + row * ld + head * D : nullptr, a.dqkv_x3 ? a.dqkv_x3 + row * NP * ld + head * D : nullptr, ld,
                dq, a.scale, g);
    }
    if (a.bias_part) acc_colsum(red, dq, a.scale, vr, i, g);
  }

  // ---- pass 2: dK, dV for keys rw; queries from the images of Q and dO.  Nothing is loaded again: this wave's K and V rows
  // come out of the images before they are overwritten, and the images of Q and dO are the row fragments the waves already
  // hold -- lane (i, g) has exactly chunk 4 s + g of row rw of every piece.
  bf16x8 kf[2][NP], vf[2][NP];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      kf[s][p] = frag_row(I1 + p * IMG, wave * 16, lo.row[s]);
      vf[s][p] = frag_row(I1 + p * IMG, wave * 16, lo.row2[s]);
    }
  __syncthreads();                                       // every wave is done with K and V
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      *reinterpret_cast<bf16x8*>(I1 + p * IMG + wave * 16 * ROWB + lo.row[s]) = qf[s][p];
      *reinterpret_cast<bf16x8*>(I1 + p * IMG + wave * 16 * ROWB + lo.row2[s]) = dof[s][p];
    }
  __syncthreads();
  {
    const float bias = mb[rw];
    // keep flag of (query, this lane's key): word (key >> 5, (key >> 2) & 1) of the query, bit 4 ((key & 31) >> 3) + (key & 3)
    const u16* kbits = kb_s + (rw >> 5) * 2 + ((rw >> 2) & 1);
    const int kbit = 4 * ((rw & 31) >> 3) + (rw & 3);
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) { dk[db] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[db] = dk[db]; }
    for (int qc = 0; qc < ((LAB & 2) ? 0 : nloop); ++qc) {
      const int q0 = qc * 32;
      f32x4 s[2], dp[2];
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) {
        s[b2] = f32x4{0.f, 0.f, 0.f, 0.f};
        dp[b2] = s[b2];
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          bf16x8 qa[NP], da[NP];
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            qa[p] = frag_row(I1 + p * IMG, q0 + 16 * b2, lo.row[st]);
            da[p] = frag_row(I1 + p * IMG, q0 + 16 * b2, lo.row2[st]);
          }
          mac<NP, LAB>(s[b2], qa, kf[st]);                       // S[query][key] = Q . K^T
          mac<NP, LAB>(dp[b2], da, vf[st]);                      // dP[query][key] = dO . V^T
        }
      }
      f32x4 pd[2];
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) {
        const int qr = q0 + 16 * b2 + 4 * g;             // rows (queries) qr .. qr + 3 of this lane
        const f32x4 lse4 = *reinterpret_cast<const f32x4*>(lse_s + qr);
        const f32x4 dl4 = *reinterpret_cast<const f32x4*>(delta_s + qr);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float m = 1.f;
          if (drop) {
            const int qq = qr + r < Lb ? qr + r : 0;
            m = ((kbits[qq * nblk * 2] >> kbit) & 1u) ? a.drop_scale : 0.f;
          }
          const float p = __expf(s[b2][r] * a.scale + bias - lse4[r]);
          pd[b2][r] = p * m;
          s[b2][r] = p * (dp[b2][r] * m - dl4[r]);     // x scale when dK leaves
        }
      }
      // the two output products one after the other (each with its own pieces and fragments live: 168 registers per lane)
      __builtin_amdgcn_sched_barrier(0);
      {
        bf16x8 pb3[NP];
        split8<NP>(pd[0], pd[1], pb3);
#pragma unroll
        for (int db = 0; db < 4; ++db) {                 // dV^T[d][key] += dO^T . Pd
          bf16x8 dt[NP];
#pragma unroll
          for (int p = 0; p < NP; ++p) dt[p] = frag_tr(I1 + p * IMG, q0, lo.tr2[db]);
          mac<NP, LAB>(dv[db], dt, pb3);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      {
        bf16x8 db3[NP];
        split8<NP>(s[0], s[1], db3);
#pragma unroll
        for (int db = 0; db < 4; ++db) {                 // dK^T[d][key] += Q^T . dS
          bf16x8 qt[NP];
#pragma unroll
          for (int p = 0; p < NP; ++p) qt[p] = frag_tr(I1 + p * IMG, q0, lo.tr[db]);
          mac<NP, LAB>(dk[db], qt, db3);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      const size_t row = (size_t)sp.row0 + (vr ? rw : 0);
      store_row<NP>(vr, a.dqkv ? a.dqkv + row * ld + a.H + head * D : nullptr,
                a.dqkv_x3 ? a.dqkv_x3 + row * NP * ld + a.H + head * D : nullptr, ld, dk, a.scale, g);
      store_row<NP>(vr, a.dqkv ? a.dqkv + row * ld + 2 * a.H + head * D : nullptr,
                a.dqkv_x3 ? a.dqkv_x3 + row * NP * ld + 2 * a.H + head * D : nullptr, ld, dv, 1.0f, g);
    }
    if (a.bias_part) {
      acc_colsum(red + 64, dk, a.scale, vr, i, g);
      acc_colsum(red + 128, dv, 1.0f, vr, i, g);
      __syncthreads();
      for (int t = tid; t < 192; t += nthr)
        a.bias_part[(size_t)b * 3 * a.H + (t >> 6) * a.H + head * D + (t & 63)] = red[t];
    }
  }
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
  UCHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)bytes));
  return 0;
}

int fill(Args& a, int B, int L, int nh, float p_drop, const void* keep_bits, const char* who) {
  UCHECK_ARG(B > 0 && L > 0 && nh > 0, "%s: bad dims B=%d L=%d nh=%d", who, B, L, nh);
  UCHECK_SHAPE(L <= MAXL, "%s: L %d > %d", who, L, MAXL);
  UCHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "%s: bad dropout p", who);
  UCHECK_ARG(p_drop == 0.f || keep_bits, "%s: dropout needs the keep flags (uniter_attn_keep_bits_gen)", who);
  a.B = B; a.L = L; a.nh = nh; a.H = nh * D;
  a.scale = 0.125f;        // 1/sqrt(64), model/layer.py:86
  a.keep_bits = p_drop > 0.f ? static_cast<const u16*>(keep_bits) : nullptr;
  a.drop_scale = 1.0f / (1.0f - p_drop);
  static const int prio = [] { const char* e = getenv("UNITER_ATTN_PRIO"); return e ? atoi(e) : 2; }();
  a.prio = prio;
  return 0;
}

template <int NP, bool QB16>
int launch_fwd(const Args& a, hipStream_t st) {
  const int Lr = (a.L + 31) / 32 * 32;
  const size_t lds = (size_t)2 * NP * IMG + (size_t)Lr * 4;
  UCHECK_RC(set_lds(attn_x3_fwd_kernel<NP, QB16>, lds));
  hipLaunchKernelGGL((attn_x3_fwd_kernel<NP, QB16>), dim3(a.B * a.nh), dim3(Lr * 4), lds, st, a, Lr);
  UCHECK_LAUNCH();
  return 0;
}

template <int NP, bool QB16>
int launch_bwd(const Args& a, hipStream_t st) {
  const int Lr = (a.L + 31) / 32 * 32;
  const size_t lds = (size_t)2 * NP * IMG + (size_t)3 * Lr * 4 + 192 * 4 + (size_t)Lr * (Lr / 32) * 4;
#ifdef UNITER_X3_LAB
  if constexpr (NP == 3 && !QB16) {
    const char* e = getenv("UNITER_ATTN_X3_LAB");
    const int lab = e ? atoi(e) : 0;
#define X3A_LAB_CASE(N)                                                                                                  \
    if (lab == N) {                                                                                                      \
      UCHECK_RC(set_lds(attn_x3_bwd_kernel<3, false, N>, lds));                                                          \
      hipLaunchKernelGGL((attn_x3_bwd_kernel<3, false, N>), dim3(a.B * a.nh), dim3(Lr * 4), lds, st, a, Lr);             \
      UCHECK_LAUNCH();                                                                                                   \
      return 0;                                                                                                          \
    }
    X3A_LAB_CASE(1) X3A_LAB_CASE(2) X3A_LAB_CASE(3) X3A_LAB_CASE(4) X3A_LAB_CASE(5) X3A_LAB_CASE(6) X3A_LAB_CASE(8) X3A_LAB_CASE(11)
#undef X3A_LAB_CASE
  }
#endif
  UCHECK_RC(set_lds(attn_x3_bwd_kernel<NP, QB16, 0>, lds));
  hipLaunchKernelGGL((attn_x3_bwd_kernel<NP, QB16, 0>), dim3(a.B * a.nh), dim3(Lr * 4), lds, st, a, Lr);
  UCHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" int uniter_attn_x3_max_len(void) { return MAXL; }

extern "C" int uniter_attn_x3_fwd(const float* qkv, const float* attn_mask, const int32_t* cu_seqlens, float* ctx, void* ctx_x3,
                                  float* lse, const void* keep_bits, int B, int L, int nh, float p_drop, void* stream) {
  UCHECK_ARG(qkv && (ctx || ctx_x3) && ((attn_mask != nullptr) != (cu_seqlens != nullptr)),
             "attn_x3_fwd: null pointer, or not exactly one of attn_mask / cu_seqlens");
  UCHECK_ARG(((uintptr_t)ctx_x3 & 15) == 0 && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)ctx & 15) == 0, "attn_x3_fwd: misaligned pointer");
  Args a = {};
  UCHECK_RC(fill(a, B, L, nh, p_drop, keep_bits, "attn_x3_fwd"));
  a.qkv = qkv; a.mask = attn_mask; a.cu = cu_seqlens; a.ctx = ctx; a.ctx_x3 = (u16*)ctx_x3; a.lse = lse;
  return launch_fwd<3, false>(a, (hipStream_t)stream);
}

extern "C" int uniter_attn_x3_bwd(const float* qkv, const float* attn_mask, const int32_t* cu_seqlens, const float* ctx,
                                  const float* lse, const float* dctx, int dctx_slabs, size_t dctx_slab_stride, float* dqkv,
                                  void* dqkv_x3, float* bias_part, const void* keep_bits, float* delta, int B, int L, int nh,
                                  float p_drop, void* stream) {
  UCHECK_ARG(dctx_slabs >= 1 && dctx_slabs <= 4 && (dctx_slabs == 1 || dctx_slab_stride % 4 == 0),
             "attn_x3_bwd: dctx_slabs %d (1..4) / slab stride not a multiple of 4 elements", dctx_slabs);
  UCHECK_ARG(qkv && ctx && lse && dctx && (dqkv || dqkv_x3) && delta && ((attn_mask != nullptr) != (cu_seqlens != nullptr)),
             "attn_x3_bwd: null pointer, or not exactly one of attn_mask / cu_seqlens");
  UCHECK_ARG(((uintptr_t)dqkv_x3 & 15) == 0 && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)ctx & 15) == 0 &&
                 ((uintptr_t)dctx & 15) == 0 && ((uintptr_t)dqkv & 15) == 0, "attn_x3_bwd: misaligned pointer");
  Args a = {};
  UCHECK_RC(fill(a, B, L, nh, p_drop, keep_bits, "attn_x3_bwd"));
  a.qkv = qkv; a.mask = attn_mask; a.cu = cu_seqlens; a.ctx = const_cast<float*>(ctx); a.lse = const_cast<float*>(lse);
  a.dctx = dctx; a.dctx_slabs = dctx_slabs; a.dctx_slab_stride = dctx_slab_stride;
  a.dqkv = dqkv; a.dqkv_x3 = (u16*)dqkv_x3; a.bias_part = bias_part; a.delta = delta;
  return launch_bwd<3, false>(a, (hipStream_t)stream);
}

// The bf16 mode's attention in the same decomposition (one wave per 16 rows, one LDS image per operand read row-wise and
// transposed, backward without the scratch hand-over): operands rounded to bf16 where they become MFMA operands, ONE product per
// block -- the arithmetic of attention_bf16.hip (uniter_attn_bf16_fwd_pre / uniter_attn_bf16_bwd), whose kernels remain the
// path for keep flags drawn inside the kernel.  ctx_bf16 [rows][H], dqkv_bf16 [rows][3H].
extern "C" int uniter_attn_b16x_fwd(const void* qkv, int qkv_is_bf16, const float* attn_mask, const int32_t* cu_seqlens, float* ctx,
                                    void* ctx_bf16, float* lse, const void* keep_bits, int B, int L, int nh, float p_drop,
                                    void* stream) {
  UCHECK_ARG(qkv && (ctx || ctx_bf16) && ((attn_mask != nullptr) != (cu_seqlens != nullptr)),
             "attn_b16x_fwd: null pointer, or not exactly one of attn_mask / cu_seqlens");
  UCHECK_ARG(((uintptr_t)ctx_bf16 & 15) == 0 && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)ctx & 15) == 0, "attn_b16x_fwd: misaligned pointer");
  Args a = {};
  UCHECK_RC(fill(a, B, L, nh, p_drop, keep_bits, "attn_b16x_fwd"));
  a.qkv = qkv; a.mask = attn_mask; a.cu = cu_seqlens; a.ctx = ctx; a.ctx_x3 = (u16*)ctx_bf16; a.lse = lse;
  return qkv_is_bf16 ? launch_fwd<1, true>(a, (hipStream_t)stream) : launch_fwd<1, false>(a, (hipStream_t)stream);
}

extern "C" int uniter_attn_b16x_bwd(const void* qkv, int qkv_is_bf16, const float* attn_mask, const int32_t* cu_seqlens,
                                    const float* ctx, const float* lse, const float* dctx, float* dqkv, void* dqkv_bf16,
                                    float* bias_part, const void* keep_bits, float* delta, int B, int L, int nh, float p_drop,
                                    void* stream) {
  UCHECK_ARG(qkv && ctx && lse && dctx && (dqkv || dqkv_bf16) && delta && ((attn_mask != nullptr) != (cu_seqlens != nullptr)),
             "attn_b16x_bwd: null pointer, or not exactly one of attn_mask / cu_seqlens");
  UCHECK_ARG(((uintptr_t)dqkv_bf16 & 15) == 0 && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)ctx & 15) == 0 &&
                 ((uintptr_t)dctx & 15) == 0 && ((uintptr_t)dqkv & 15) == 0, "attn_b16x_bwd: misaligned pointer");
  Args a = {};
  UCHECK_RC(fill(a, B, L, nh, p_drop, keep_bits, "attn_b16x_bwd"));
  a.qkv = qkv; a.mask = attn_mask; a.cu = cu_seqlens; a.ctx = const_cast<float*>(ctx); a.lse = const_cast<float*>(lse);
  a.dctx = dctx; a.dctx_slabs = 1; a.dctx_slab_stride = 0;
  a.dqkv = dqkv; a.dqkv_x3 = (u16*)dqkv_bf16; a.bias_part = bias_part; a.delta = delta;
  return qkv_is_bf16 ? launch_bwd<1, true>(a, (hipStream_t)stream) : launch_bwd<1, false>(a, (hipStream_t)stream);
}
