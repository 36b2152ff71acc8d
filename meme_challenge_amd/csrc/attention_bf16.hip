// Fused self-attention on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, fp32 accumulate) for the
// bf16 precision mode: Q, K, V (fp32 in memory, the QKV GEMM's output) are rounded to bf16 while they
// are staged, scores / softmax / dropout / LSE / deltas stay fp32, the probabilities and score
// gradients are rounded to bf16 where they become MFMA operands -- the usual mixed-precision
// attention.  Replaces the same reference code as attention_f32.hip (model/layer.py:80-100) and is
// drop-in for its L <= 192 "split" kernels: same work decomposition (one workgroup per (batch,
// head), two waves per 32-row block meeting in LDS), same transposed orientation (the accumulator
// of S^T = K.Q^T feeds O^T = V^T.P^T from registers), same Philox element indices, same varlen
// (cu_seqlens) support, same outputs (fp32 + optional bf16 copies).
//
// What changes with the 16-deep bf16 MFMA:
//   * a 32x32 score tile costs 4 MFMAs of 32 cycles instead of 32 of 64: the kernels are VALU-bound
//     (softmax, Philox), not matrix-bound;
//   * operand k-slots: lane-half h of k16-step s supplies 8 values.  For operands that come out of
//     an accumulator (P^T, dS^T: lane = query, registers = keys) the 8 values are registers
//     8s .. 8s+7, i.e. keys 16s + 8(t>>2) + 4h + (t&3): the partner operand (V^T or K^T) is therefore
//     staged TRANSPOSED in LDS ([64 d][keys], 400-B rows) and read as two 8-byte pieces per MFMA at
//     columns 16s+4h and 16s+8+4h -- conflict-free, no transposed-read instruction needed;
//   * the dK/dV kernel contracts over queries with both operands in natural order: Q^T / dO^T
//     transposed in LDS (one ds_read_b128 per MFMA) against Pd / dS read from the bf16 scratch the dQ
//     kernel wrote ([B*nh][key][query], half the bytes of the fp32 kernels' scratch).
#include <type_traits>
#include <stdlib.h>
#include "common.h"
#include "philox.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

constexpr int D = 64;
constexpr int KLD = 72;          // row-major image [row][64 d] bf16, 144-B rows (conflict-free ds_read_b128)
constexpr int TLD = 200;         // transposed image [64 d][<= 192 rows] bf16, 400-B rows
constexpr int MAXL = 192;
constexpr int XROW = 34;         // floats exchanged per lane between the two halves (32 accumulators + m, l)
constexpr float NEG_INF = -__builtin_huge_valf();

struct Args {
  const void* qkv;         // [rows, 3H] fp32, or bf16 when qb16 (the QKV GEMM's bf16 output, read as stored)
  int qb16;
  const float* mask;       // [B, L] or NULL (varlen)
  const int* cu;           // [B+1] or NULL
  float* ctx; u16* ctx_b16;
  float* lse;              // [B, nh, L]
  const float* dctx;
  float* dqkv; u16* dqkv_b16;
  u16* keep_bits;          // optional [B*nh, L, Lr/32, 2] dropout keep flags written by the forward pass, read by dQ
  int prio;                // wave priority (UNITER_ATTN_PRIO, default 2)
  int keep_ready;          // the forward pass reads keep_bits (uniter_attn_keep_bits_gen drew them ahead) instead of drawing them
  float* bias_part;        // optional [B, 3H] per-sample column sums of dqkv (QKV bias gradient partials)
  float* delta;            // [B, nh, L]
  u16* pd_ws; u16* ds_ws;  // [B*nh][Lr][Lr] bf16, indexed [key][query]
  int B, L, nh, H, Lp4;
  float scale;
  DropCfg drop;
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
  bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}

// One [L, 64] operand on its way into LDS: the 16-byte pieces of this thread (blockDim.x = 4 Lr: four pieces of an fp32
// source, two of a bf16 source) are ALL loaded before anything else happens -- the kernels issue the loads of every
// operand and of their row fragments first and write LDS afterwards (piece by piece, each write waited for its own
// load: ~10 us of memory latency in a row at the head of every attention kernel, tests/tools/attn_phase_lab.py).
// store_rm: bf16 image [row][KLD] (rows >= L zero); store_tr: transposed image [d][TLD] (columns >= L zero).
struct StageF {       // fp32 source, rounded to bf16 on the way
  f32x4 v[4];
  __device__ __forceinline__ void load(const float* __restrict__ base, int ld, int L, int Lr, int tid, int nthr) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = tid + it * nthr, r = idx >> 4, c4 = idx & 15;
      v[it] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (idx < Lr * 16 && r < L) v[it] = *reinterpret_cast<const f32x4*>(base + (size_t)r * ld + c4 * 4);
    }
  }
  __device__ __forceinline__ void store_rm(u16* s, int Lr, int tid, int nthr) const {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = tid + it * nthr, r = idx >> 4, c4 = idx & 15;
      if (idx < Lr * 16) *reinterpret_cast<u32x2*>(s + r * KLD + c4 * 4) = u32x2{pack2(v[it][0], v[it][1]), pack2(v[it][2], v[it][3])};
    }
  }
  __device__ __forceinline__ void store_tr(u16* s, int Lr, int tid, int nthr) const {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = tid + it * nthr, r = idx >> 4, c4 = idx & 15;
      if (idx < Lr * 16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) s[(c4 * 4 + e) * TLD + r] = __builtin_bit_cast(u16, (__bf16)v[it][e]);
      }
    }
  }
};
struct StageH {       // bf16 source, copied as stored
  typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
  u16x8 v[2];
  __device__ __forceinline__ void load(const u16* __restrict__ base, int ld, int L, int Lr, int tid, int nthr) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int idx = tid + it * nthr, r = idx >> 3, c8 = idx & 7;
      v[it] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (idx < Lr * 8 && r < L) v[it] = *reinterpret_cast<const u16x8*>(base + (size_t)r * ld + c8 * 8);
    }
  }
  __device__ __forceinline__ void store_rm(u16* s, int Lr, int tid, int nthr) const {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int idx = tid + it * nthr, r = idx >> 3, c8 = idx & 7;
      if (idx < Lr * 8) *reinterpret_cast<u16x8*>(s + r * KLD + c8 * 8) = v[it];
    }
  }
  __device__ __forceinline__ void store_tr(u16* s, int Lr, int tid, int nthr) const {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int idx = tid + it * nthr, r = idx >> 3, c8 = idx & 7;
      if (idx < Lr * 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s[(c8 * 8 + e) * TLD + r] = v[it][e];
      }
    }
  }
};
__device__ __forceinline__ void row_frags(bf16x8 (&f)[4], const u16* __restrict__ row, bool valid, int h) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    bf16x8 v = {};
    if (valid) v = *reinterpret_cast<const bf16x8*>(row + 16 * s + 8 * h);
    f[s] = v;
  }
}

// B operand from a global fp32 row (64 values): step s, half h -> d = 16s + 8h .. +7
__device__ __forceinline__ void row_frags(bf16x8 (&f)[4], const float* __restrict__ row, bool valid, int h) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
    if (valid) {
      a = *reinterpret_cast<const f32x4*>(row + 16 * s + 8 * h);
      b = *reinterpret_cast<const f32x4*>(row + 16 * s + 8 * h + 4);
    }
    f[s] = bf16x8{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3], (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
  }
}
// A operand out of a row-major image: row r, step s, half h
__device__ __forceinline__ bf16x8 frag_rm(const u16* s, int r, int step, int h) {
  return *reinterpret_cast<const bf16x8*>(s + r * KLD + step * 16 + 8 * h);
}
// A operand out of a transposed image in ACCUMULATOR k-order: columns c0+16s+4h..+3 and c0+16s+8+4h..+3
__device__ __forceinline__ bf16x8 frag_t_acc(const u16* s, int d, int c0, int step, int h) {
  const u16* p = s + d * TLD + c0 + 16 * step + 4 * h;
  const bf16x4 lo = *reinterpret_cast<const bf16x4*>(p), hi = *reinterpret_cast<const bf16x4*>(p + 8);
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// A operand out of a transposed image in natural k-order: columns c0+16s+8h .. +7
__device__ __forceinline__ bf16x8 frag_t_nat(const u16* s, int d, int c0, int step, int h) {
  return *reinterpret_cast<const bf16x8*>(s + d * TLD + c0 + 16 * step + 8 * h);
}
// B operand out of an accumulator: registers 8s .. 8s+7
__device__ __forceinline__ bf16x8 pack_acc(const f32x16& v, int step) {
  const int o = 8 * step;
  return bf16x8{(__bf16)v[o], (__bf16)v[o + 1], (__bf16)v[o + 2], (__bf16)v[o + 3],
                (__bf16)v[o + 4], (__bf16)v[o + 5], (__bf16)v[o + 6], (__bf16)v[o + 7]};
}
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// transposed accumulator pair (rows = d) -> one row of the [rows, H] / [rows, 3H] output, fp32 and bf16
__device__ __forceinline__ void store_rowT(float* __restrict__ row, u16* __restrict__ row_b, const f32x16& a0,
                                           const f32x16& a1, float mul, int h) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 v0 = {a0[4 * g] * mul, a0[4 * g + 1] * mul, a0[4 * g + 2] * mul, a0[4 * g + 3] * mul};
    const f32x4 v1 = {a1[4 * g] * mul, a1[4 * g + 1] * mul, a1[4 * g + 2] * mul, a1[4 * g + 3] * mul};
    if (row) {
      *reinterpret_cast<f32x4*>(row + 8 * g + 4 * h) = v0;
      *reinterpret_cast<f32x4*>(row + 32 + 8 * g + 4 * h) = v1;
    }
    if (row_b) {
      *reinterpret_cast<bf16x4*>(row_b + 8 * g + 4 * h) = bf16x4{(__bf16)v0[0], (__bf16)v0[1], (__bf16)v0[2], (__bf16)v0[3]};
      *reinterpret_cast<bf16x4*>(row_b + 32 + 8 * g + 4 * h) = bf16x4{(__bf16)v1[0], (__bf16)v1[1], (__bf16)v1[2], (__bf16)v1[3]};
    }
  }
}

// sum over the 32 lanes of each wave half; column sums of a transposed accumulator pair -> red[0..63] (LDS)
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ void acc_colsum(float* red, const f32x16& a0, const f32x16& a1, bool valid, int i, int h) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float s0 = half_sum(valid ? a0[r] : 0.f), s1 = half_sum(valid ? a1[r] : 0.f);
    if (i == 0) {
      const int d = (r & 3) + 8 * (r >> 2) + 4 * h;
      atomicAdd(red + d, s0);
      atomicAdd(red + 32 + d, s1);
    }
  }
}

extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

__device__ __forceinline__ void stage_mask(float* mb, const Args& a, int b, int Lb, int Lr, int tid, int nthr) {
  for (int k = tid; k < Lr; k += nthr)
    mb[k] = k < Lb ? (a.mask ? (1.0f - a.mask[(size_t)b * a.L + k]) * -10000.0f : 0.f) : NEG_INF;
}

// ---------------------------------------------------------------- forward ---
// LDS: K row-major | V transposed | mask bias; afterwards the (O, m, l) exchange aliases it
__global__ __launch_bounds__(768) void attn_b16_fwd_kernel(const Args a, int Lr) {
  set_wave_prio(a.prio);      // critical path of the step: ahead of the side stream's weight gradients (common.h)
  u16* Kb = reinterpret_cast<u16*>(smem_raw);
  u16* Vt = Kb + Lr * KLD;
  float* mb = reinterpret_cast<float*>(Vt + D * TLD);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
  const int i = lane & 31, h = lane >> 5;
  const int nblk = Lr >> 5;
  const int qb = wave % nblk, half = wave / nblk;
  const int bh = blockIdx.x, b = bh / a.nh, head = bh - b * a.nh;
  const Span sp = span_of(a, b);
  const int Lb = sp.Lb, ld = 3 * a.H;
  const size_t boff = (size_t)sp.row0 * ld + head * D;
  const int q = qb * 32 + i;
  const bool vq = q < Lb;
  bf16x8 qf[4];
  if (a.qb16) {
    const u16* base = static_cast<const u16*>(a.qkv) + boff;
    StageH sk, sv;
    sk.load(base + a.H, ld, Lb, Lr, tid, nthr);
    sv.load(base + 2 * a.H, ld, Lb, Lr, tid, nthr);
    row_frags(qf, base + (size_t)q * ld, vq, h);
    stage_mask(mb, a, b, Lb, Lr, tid, nthr);
    sk.store_rm(Kb, Lr, tid, nthr);
    sv.store_tr(Vt, Lr, tid, nthr);
  } else {
    const float* base = static_cast<const float*>(a.qkv) + boff;
    StageF sk, sv;
    sk.load(base + a.H, ld, Lb, Lr, tid, nthr);
    sv.load(base + 2 * a.H, ld, Lb, Lr, tid, nthr);
    row_frags(qf, base + (size_t)q * ld, vq, h);
    stage_mask(mb, a, b, Lb, Lr, tid, nthr);
    sk.store_rm(Kb, Lr, tid, nthr);
    sv.store_tr(Vt, Lr, tid, nthr);
  }
  __syncthreads();

  const int kmid = ((sp.nb + 1) >> 1) * 32;
  const int kbeg = half ? kmid : 0, kend = qb < sp.nb ? (half ? sp.nb * 32 : kmid) : 0;
  f32x16 o0 = zero16(), o1 = zero16();
  float m_run = NEG_INF, l_run = 0.f;
  for (int k0 = kbeg; k0 < kend; k0 += 32) {
    f32x16 s = zero16();
#pragma unroll
    for (int st = 0; st < 4; ++st) s = MFMA(frag_rm(Kb, k0 + i, st, h), qf[st], s);     // S^T[key][query]
    float mx = NEG_INF;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 bias = *reinterpret_cast<const f32x4*>(mb + k0 + 8 * g + 4 * h);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        s[4 * g + t] = s[4 * g + t] * a.scale + bias[t];
        mx = fmaxf(mx, s[4 * g + t]);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);
    float ls = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = __expf(s[r] - m_new); ls += s[r]; }
    l_run = l_run * alpha + ls;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
    if (a.drop.active && vq) {
      unsigned bits = 0;
      const size_t kidx = (((size_t)bh * a.L + q) * nblk + (k0 >> 5)) * 2 + h;
      if (a.keep_ready) {
        bits = a.keep_bits[kidx];
      } else {
        const uint64_t grow = ((uint64_t)bh * a.L + q) * a.Lp4 + (k0 >> 2);
#pragma unroll
        for (int g = 0; g < 4; ++g) bits |= drop_bits4(a.drop, grow + 2 * g + h) << (4 * g);
        if (a.keep_bits) a.keep_bits[kidx] = (u16)bits;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] *= ((bits >> r) & 1u) ? a.drop.scale : 0.f;
    }
#pragma unroll
    for (int st = 0; st < 2; ++st) {                                   // O^T[d][query] += V^T . Pd^T
      const bf16x8 pb = pack_acc(s, st);
      o0 = MFMA(frag_t_acc(Vt, i, k0, st, h), pb, o0);
      o1 = MFMA(frag_t_acc(Vt, 32 + i, k0, st, h), pb, o1);
    }
  }
  __syncthreads();
  float* xb = reinterpret_cast<float*>(smem_raw) + (size_t)qb * XROW * 64;
  if (half) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { xb[r * 64 + lane] = o0[r]; xb[(16 + r) * 64 + lane] = o1[r]; }
    xb[32 * 64 + lane] = m_run;
    xb[33 * 64 + lane] = l_run;
  }
  __syncthreads();
  if (half) return;
  {
    const float m_b = xb[32 * 64 + lane], l_b = xb[33 * 64 + lane];
    const float m_new = fmaxf(m_run, m_b);
    const float wa = __expf(m_run - m_new), wb = m_b == NEG_INF ? 0.f : __expf(m_b - m_new);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      o0[r] = o0[r] * wa + xb[r * 64 + lane] * wb;
      o1[r] = o1[r] * wa + xb[(16 + r) * 64 + lane] * wb;
    }
    l_run = l_run * wa + l_b * wb;
    m_run = m_new;
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (vq) {
    const size_t off = ((size_t)sp.row0 + q) * a.H + head * D;
    store_rowT(a.ctx + off, a.ctx_b16 ? a.ctx_b16 + off : nullptr, o0, o1, 1.0f / l_tot, h);
    if (h == 0 && a.lse) a.lse[(size_t)bh * a.L + q] = m_run + __logf(l_tot);
  }
}

// ------------------------------------------------- backward: dQ, delta, Pd / dS scratch ---
// LDS: K row-major | V row-major | K transposed | mask bias
template <bool QB16>
__device__ __forceinline__ void b16_dq_body(const Args& a, int Lr, int red_off) {
  u16* Kb = reinterpret_cast<u16*>(smem_raw);
  u16* Vb = Kb + Lr * KLD;
  u16* Kt = Vb + Lr * KLD;
  float* mb = reinterpret_cast<float*>(Kt + D * TLD);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
  const int i = lane & 31, h = lane >> 5;
  const int nblk = Lr >> 5;
  const int qb = wave % nblk, half = wave / nblk;
  const int bh = blockIdx.x, b = bh / a.nh, head = bh - b * a.nh;
  const Span sp = span_of(a, b);
  const int Lb = sp.Lb, ld = 3 * a.H;
  const size_t boff = (size_t)sp.row0 * ld + head * D;
  const int q = qb * 32 + i;
  const bool vq = q < Lb;
  bf16x8 qf[4], dof[4];
  const float* dorow = a.dctx + ((size_t)sp.row0 + q) * a.H + head * D;
  const float* orow = a.ctx + ((size_t)sp.row0 + q) * a.H + head * D;
  // this lane's 32 of the 64 values of its dO and O rows (d = 16 s + 8 h + 4 j + 0..3 at index 2 s + j: the k-slots of the
  // dO operand); delta = O . dO is the sum of both halves' partial dot products
  f32x4 dob[8], ob[8];
  float* red = reinterpret_cast<float*>(smem_raw + red_off);
  typedef typename std::conditional<QB16, u16, float>::type src_t;
  typename std::conditional<QB16, StageH, StageF>::type sk, sv;
  {
    // every global load of the prologue is issued before the first LDS write (K is loaded once for both of its images)
    const src_t* base = static_cast<const src_t*>(a.qkv) + boff;
    sk.load(base + a.H, ld, Lb, Lr, tid, nthr);
    sv.load(base + 2 * a.H, ld, Lb, Lr, tid, nthr);
    row_frags(qf, base + (size_t)q * ld, vq, h);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      dob[c] = f32x4{0.f, 0.f, 0.f, 0.f};
      ob[c] = dob[c];
      if (vq) {
        dob[c] = *reinterpret_cast<const f32x4*>(dorow + 16 * (c >> 1) + 8 * h + 4 * (c & 1));
        ob[c] = *reinterpret_cast<const f32x4*>(orow + 16 * (c >> 1) + 8 * h + 4 * (c & 1));
      }
    }
    stage_mask(mb, a, b, Lb, Lr, tid, nthr);
    for (int t = tid; t < 192; t += nthr) red[t] = 0.f;        // (a 32-row workgroup has only 128 threads)
    sk.store_rm(Kb, Lr, tid, nthr);
    sk.store_tr(Kt, Lr, tid, nthr);
    sv.store_rm(Vb, Lr, tid, nthr);
  }
  float delta = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) delta += ob[c][0] * dob[c][0] + ob[c][1] * dob[c][1] + ob[c][2] * dob[c][2] + ob[c][3] * dob[c][3];
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    const f32x4 x = dob[2 * st], y = dob[2 * st + 1];
    dof[st] = bf16x8{(__bf16)x[0], (__bf16)x[1], (__bf16)x[2], (__bf16)x[3], (__bf16)y[0], (__bf16)y[1], (__bf16)y[2], (__bf16)y[3]};
  }
  delta += __shfl_xor(delta, 32, 64);
  if (vq && h == 0 && half == 0) a.delta[(size_t)bh * a.L + q] = delta;
  const float lse = vq ? a.lse[(size_t)bh * a.L + q] : -NEG_INF;      // +inf: every probability of a padded query is 0
  __syncthreads();

  const int kmid = ((sp.nb + 1) >> 1) * 32;
  const int kbeg = half ? kmid : 0, kend = qb < sp.nb ? (half ? sp.nb * 32 : kmid) : 0;
  u16* pdw = a.pd_ws + (size_t)bh * Lr * Lr + qb * 32 + i;
  u16* dsw = a.ds_ws + (size_t)bh * Lr * Lr + qb * 32 + i;
  f32x16 dq0 = zero16(), dq1 = zero16();
  for (int k0 = kbeg; k0 < kend; k0 += 32) {
    f32x16 s = zero16(), dp = zero16();
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      s = MFMA(frag_rm(Kb, k0 + i, st, h), qf[st], s);
      dp = MFMA(frag_rm(Vb, k0 + i, st, h), dof[st], dp);               // dP^T[key][query] = V . dO^T
    }
    unsigned bits = 0xffffu;
    const bool stored = a.drop.active && vq && a.keep_bits;
    if (stored) bits = a.keep_bits[(((size_t)bh * a.L + q) * nblk + (k0 >> 5)) * 2 + h];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 bias = *reinterpret_cast<const f32x4*>(mb + k0 + 8 * g + 4 * h);
      float m4[4] = {1.f, 1.f, 1.f, 1.f};
      if (stored) {
#pragma unroll
        for (int t = 0; t < 4; ++t) m4[t] = ((bits >> (4 * g + t)) & 1u) ? a.drop.scale : 0.f;
      } else if (a.drop.active && vq) {
        drop_mult4(a.drop, ((uint64_t)bh * a.L + q) * a.Lp4 + (k0 >> 2) + 2 * g + h, m4);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float p = __expf(s[4 * g + t] * a.scale + bias[t] - lse);
        const float pdv = p * m4[t];
        const float dsv = p * (dp[4 * g + t] * m4[t] - delta) * a.scale;
        s[4 * g + t] = dsv;
        const size_t off = (size_t)(k0 + 8 * g + 4 * h + t) * Lr;         // row = key, 32 lanes = 64 contiguous bytes
        pdw[off] = __builtin_bit_cast(u16, (__bf16)pdv);
        dsw[off] = __builtin_bit_cast(u16, (__bf16)dsv);
      }
    }
#pragma unroll
    for (int st = 0; st < 2; ++st) {                                    // dQ^T[d][query] += K^T . dS^T
      const bf16x8 db = pack_acc(s, st);
      dq0 = MFMA(frag_t_acc(Kt, i, k0, st, h), db, dq0);
      dq1 = MFMA(frag_t_acc(Kt, 32 + i, k0, st, h), db, dq1);
    }
  }
  __syncthreads();
  float* xb = reinterpret_cast<float*>(smem_raw) + (size_t)qb * XROW * 64;
  if (half) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { xb[r * 64 + lane] = dq0[r]; xb[(16 + r) * 64 + lane] = dq1[r]; }
  }
  __syncthreads();
  if (!half) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq0[r] += xb[r * 64 + lane]; dq1[r] += xb[(16 + r) * 64 + lane]; }
    if (vq) {
      const size_t off = ((size_t)sp.row0 + q) * ld + head * D;
      store_rowT(a.dqkv ? a.dqkv + off : nullptr, a.dqkv_b16 ? a.dqkv_b16 + off : nullptr, dq0, dq1, 1.0f, h);
    }
    if (a.bias_part) acc_colsum(red, dq0, dq1, vq, i, h);
  }
  if (a.bias_part) {
    __syncthreads();
    if (tid < 64) a.bias_part[(size_t)b * 3 * a.H + head * D + tid] = red[tid];
  }
}

template <bool QB16>
__global__ __launch_bounds__(768) void attn_b16_dq_kernel(const Args a, int Lr, int red_off) {
  set_wave_prio(a.prio);      // critical path of the step: ahead of the side stream's weight gradients (common.h)
  b16_dq_body<QB16>(a, Lr, red_off);
}

// ------------------------------------------------------------ backward: dK, dV ---
// LDS: Q transposed | dO transposed.  dV^T[d][key] = sum_q dO[q][d] Pd[q][key], dK^T = sum_q Q[q][d] dS[q][key]
__device__ __forceinline__ void b16_dkv_body(const Args& a, int Lr, int red_off) {
  u16* Qt = reinterpret_cast<u16*>(smem_raw);
  u16* dOt = Qt + D * TLD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
  const int i = lane & 31, h = lane >> 5;
  const int nblk = Lr >> 5;
  const int kb = wave % nblk, half = wave / nblk;
  const int bh = blockIdx.x, b = bh / a.nh, head = bh - b * a.nh;
  const Span sp = span_of(a, b);
  const int Lb = sp.Lb, ld = 3 * a.H;
  const size_t boff = (size_t)sp.row0 * ld + head * D;
  float* red = reinterpret_cast<float*>(smem_raw + red_off);
  {
    StageF sdo;
    sdo.load(a.dctx + (size_t)sp.row0 * a.H + head * D, a.H, Lb, Lr, tid, nthr);
    if (a.qb16) {
      StageH sq;
      sq.load(static_cast<const u16*>(a.qkv) + boff, ld, Lb, Lr, tid, nthr);
      sq.store_tr(Qt, Lr, tid, nthr);
    } else {
      StageF sq;
      sq.load(static_cast<const float*>(a.qkv) + boff, ld, Lb, Lr, tid, nthr);
      sq.store_tr(Qt, Lr, tid, nthr);
    }
    for (int t = tid; t < 192; t += nthr) red[t] = 0.f;        // (a 32-row workgroup has only 128 threads)
    sdo.store_tr(dOt, Lr, tid, nthr);
  }
  const int key = kb * 32 + i;
  const bool vk = key < Lb;
  const int qmid = ((sp.nb + 1) >> 1) * 32;
  const int qbeg = half ? qmid : 0, qend = kb < sp.nb ? (half ? sp.nb * 32 : qmid) : 0;
  const u16* pdr = a.pd_ws + ((size_t)bh * Lr + key) * Lr + 8 * h;      // lane's key row, natural query order
  const u16* dsr = a.ds_ws + ((size_t)bh * Lr + key) * Lr + 8 * h;
  __syncthreads();

  f32x16 dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();
  for (int q0 = qbeg; q0 < qend; q0 += 32) {
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const bf16x8 pb = *reinterpret_cast<const bf16x8*>(pdr + q0 + 16 * st);
      const bf16x8 db = *reinterpret_cast<const bf16x8*>(dsr + q0 + 16 * st);
      dv0 = MFMA(frag_t_nat(dOt, i, q0, st, h), pb, dv0);
      dv1 = MFMA(frag_t_nat(dOt, 32 + i, q0, st, h), pb, dv1);
      dk0 = MFMA(frag_t_nat(Qt, i, q0, st, h), db, dk0);
      dk1 = MFMA(frag_t_nat(Qt, 32 + i, q0, st, h), db, dk1);
    }
  }
  __syncthreads();
  float* xb = reinterpret_cast<float*>(smem_raw) + (size_t)kb * 64 * 64;
  if (half) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      xb[r * 64 + lane] = dk0[r]; xb[(16 + r) * 64 + lane] = dk1[r];
      xb[(32 + r) * 64 + lane] = dv0[r]; xb[(48 + r) * 64 + lane] = dv1[r];
    }
  }
  __syncthreads();
  if (!half) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      dk0[r] += xb[r * 64 + lane]; dk1[r] += xb[(16 + r) * 64 + lane];
      dv0[r] += xb[(32 + r) * 64 + lane]; dv1[r] += xb[(48 + r) * 64 + lane];
    }
    if (vk) {
      const size_t off = ((size_t)sp.row0 + key) * ld + head * D;
      store_rowT(a.dqkv ? a.dqkv + off + a.H : nullptr, a.dqkv_b16 ? a.dqkv_b16 + off + a.H : nullptr, dk0, dk1, 1.0f, h);
      store_rowT(a.dqkv ? a.dqkv + off + 2 * a.H : nullptr, a.dqkv_b16 ? a.dqkv_b16 + off + 2 * a.H : nullptr, dv0, dv1, 1.0f, h);
    }
    if (a.bias_part) { acc_colsum(red + 64, dk0, dk1, vk, i, h); acc_colsum(red + 128, dv0, dv1, vk, i, h); }
  }
  if (a.bias_part) {
    __syncthreads();
    for (int t = tid; t < 128; t += nthr)
      a.bias_part[(size_t)b * 3 * a.H + (1 + (t >> 6)) * a.H + head * D + (t & 63)] = red[64 + t];
  }
}

__global__ __launch_bounds__(768) void attn_b16_dkv_kernel(const Args a, int Lr, int red_off) {
  set_wave_prio(a.prio);      // critical path of the step: ahead of the side stream's weight gradients (common.h)
  b16_dkv_body(a, Lr, red_off);
}

// Both passes in one launch (attention_f32.hip, attn_bwd_fused_split_kernel): a workgroup's dK / dV pass reads only the Pd / dS
// blocks its own dQ pass wrote, so it follows behind a workgroup barrier instead of behind a second launch.
template <bool QB16>
__global__ __launch_bounds__(768) void attn_b16_bwd_fused_kernel(const Args a, int Lr, int red_dq, int red_dkv) {
  set_wave_prio(a.prio);
  b16_dq_body<QB16>(a, Lr, red_dq);
  __syncthreads();
  b16_dkv_body(a, Lr, red_dkv);
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
  UCHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)bytes));
  return 0;
}
size_t max3(size_t x, size_t y, size_t z) { return x > y ? (x > z ? x : z) : (y > z ? y : z); }

int fill(Args& a, int B, int L, int nh, float p_drop, uint64_t seed, uint32_t offset, uint32_t site) {
  UCHECK_ARG(B > 0 && L > 0 && nh > 0, "attn_bf16: bad dims B=%d L=%d nh=%d", B, L, nh);
  UCHECK_SHAPE(L <= MAXL, "attn_bf16: L %d > %d", L, MAXL);
  UCHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "attn_bf16: bad dropout p");
  a.B = B; a.L = L; a.nh = nh; a.H = nh * D; a.Lp4 = (L + 3) / 4;
  a.scale = 0.125f;        // 1/sqrt(64), model/layer.py:86
  a.drop = make_drop(p_drop, seed, offset, site);
  static const int prio = [] { const char* e = getenv("UNITER_ATTN_PRIO"); return e ? atoi(e) : 2; }();
  a.prio = prio;
  return 0;
}

}  // namespace

extern "C" size_t uniter_attn_bf16_bwd_ws_bytes(int B, int L, int nh) {
  const int Lr = (L + 31) / 32 * 32;
  if (B <= 0 || nh <= 0 || L > MAXL) return 0;
  return (size_t)2 * B * nh * Lr * Lr * sizeof(unsigned short);
}

extern "C" int uniter_attn_bf16_fwd(const void* qkv, int qkv_is_bf16, const float* attn_mask, const int32_t* cu_seqlens, float* ctx,
                                    void* ctx_bf16, float* lse, void* keep_bits, int B, int L, int nh, float p_drop,
                                    uint64_t seed, uint32_t offset, uint32_t site, void* stream) {
  return uniter_attn_bf16_fwd_pre(qkv, qkv_is_bf16, attn_mask, cu_seqlens, ctx, ctx_bf16, lse, keep_bits, 0, B, L, nh, p_drop,
                                  seed, offset, site, stream);
}

extern "C" int uniter_attn_bf16_fwd_pre(const void* qkv, int qkv_is_bf16, const float* attn_mask, const int32_t* cu_seqlens,
                                        float* ctx, void* ctx_bf16, float* lse, void* keep_bits, int keep_bits_ready, int B,
                                        int L, int nh, float p_drop, uint64_t seed, uint32_t offset, uint32_t site,
                                        void* stream) {
  UCHECK_ARG(qkv && ctx && ((attn_mask != nullptr) != (cu_seqlens != nullptr)),
             "attn_bf16_fwd: need attn_mask or cu_seqlens (not both)");
  UCHECK_ARG(!keep_bits_ready || keep_bits, "attn_bf16_fwd_pre: keep_bits_ready without keep_bits");
  Args a = {};
  UCHECK_RC(fill(a, B, L, nh, p_drop, seed, offset, site));
  a.qkv = qkv; a.qb16 = qkv_is_bf16; a.mask = attn_mask; a.cu = cu_seqlens; a.ctx = ctx; a.ctx_b16 = (u16*)ctx_bf16; a.lse = lse;
  a.keep_bits = (u16*)keep_bits; a.keep_ready = keep_bits_ready;
  const int Lr = (L + 31) / 32 * 32, nblk = Lr / 32;
  const size_t lds = max3((size_t)(Lr * KLD + D * TLD) * 2 + Lr * 4, (size_t)nblk * XROW * 64 * 4, 0);
  UCHECK_RC(set_lds(attn_b16_fwd_kernel, lds));
  hipLaunchKernelGGL(attn_b16_fwd_kernel, dim3(B * nh), dim3(Lr * 4), lds, (hipStream_t)stream, a, Lr);
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_attn_bf16_bwd(const void* qkv, int qkv_is_bf16, const float* attn_mask, const int32_t* cu_seqlens,
                                    const float* ctx, const float* lse, const float* dctx, float* dqkv,
                                    void* dqkv_bf16, float* bias_part, const void* keep_bits, float* delta, int B, int L,
                                    int nh, float p_drop, uint64_t seed, uint32_t offset, uint32_t site, void* ws,
                                    size_t ws_bytes, void* stream) {
  UCHECK_ARG(qkv && ctx && lse && dctx && (dqkv || dqkv_bf16) && delta && ws && ((attn_mask != nullptr) != (cu_seqlens != nullptr)),
             "attn_bf16_bwd: null pointer, or not exactly one of attn_mask / cu_seqlens");
  UCHECK_ARG(ws_bytes >= uniter_attn_bf16_bwd_ws_bytes(B, L, nh), "attn_bf16_bwd: workspace too small");
  Args a = {};
  UCHECK_RC(fill(a, B, L, nh, p_drop, seed, offset, site));
  a.qkv = qkv; a.qb16 = qkv_is_bf16; a.mask = attn_mask; a.cu = cu_seqlens; a.ctx = const_cast<float*>(ctx); a.lse = const_cast<float*>(lse);
  a.dctx = dctx; a.dqkv = dqkv; a.dqkv_b16 = (u16*)dqkv_bf16; a.bias_part = bias_part; a.keep_bits = (u16*)keep_bits; a.delta = delta;
  const int Lr = (L + 31) / 32 * 32, nblk = Lr / 32;
  a.pd_ws = (u16*)ws;
  a.ds_ws = a.pd_ws + (size_t)B * nh * Lr * Lr;
  // the 192-float column-sum area sits behind both the staged images and the exchange area that later aliases them
  const size_t red_dq = max3((size_t)(2 * Lr * KLD + D * TLD) * 2 + Lr * 4, (size_t)nblk * XROW * 64 * 4, 0);
  const size_t red_dkv = max3((size_t)(2 * D * TLD) * 2, (size_t)nblk * 64 * 64 * 4, 0);
  const size_t lds_dq = red_dq + 192 * 4, lds_dkv = red_dkv + 192 * 4;
  static const bool fused = [] { const char* e = getenv("UNITER_ATTN_BWD_FUSED"); return !(e && e[0] == '0'); }();
  if (fused) {
    const size_t lds = lds_dq > lds_dkv ? lds_dq : lds_dkv;
    if (a.qb16) {
      UCHECK_RC(set_lds(attn_b16_bwd_fused_kernel<true>, lds));
      hipLaunchKernelGGL(attn_b16_bwd_fused_kernel<true>, dim3(B * nh), dim3(Lr * 4), lds, (hipStream_t)stream, a, Lr, (int)red_dq,
                         (int)red_dkv);
    } else {
      UCHECK_RC(set_lds(attn_b16_bwd_fused_kernel<false>, lds));
      hipLaunchKernelGGL(attn_b16_bwd_fused_kernel<false>, dim3(B * nh), dim3(Lr * 4), lds, (hipStream_t)stream, a, Lr, (int)red_dq,
                         (int)red_dkv);
    }
    UCHECK_LAUNCH();
    return 0;
  }
  UCHECK_RC(set_lds(attn_b16_dkv_kernel, lds_dkv));
  if (a.qb16) {
    UCHECK_RC(set_lds(attn_b16_dq_kernel<true>, lds_dq));
    hipLaunchKernelGGL(attn_b16_dq_kernel<true>, dim3(B * nh), dim3(Lr * 4), lds_dq, (hipStream_t)stream, a, Lr, (int)red_dq);
  } else {
    UCHECK_RC(set_lds(attn_b16_dq_kernel<false>, lds_dq));
    hipLaunchKernelGGL(attn_b16_dq_kernel<false>, dim3(B * nh), dim3(Lr * 4), lds_dq, (hipStream_t)stream, a, Lr, (int)red_dq);
  }
  UCHECK_LAUNCH();
  hipLaunchKernelGGL(attn_b16_dkv_kernel, dim3(B * nh), dim3(Lr * 4), lds_dkv, (hipStream_t)stream, a, Lr, (int)red_dkv);
  UCHECK_LAUNCH();
  return 0;
}
