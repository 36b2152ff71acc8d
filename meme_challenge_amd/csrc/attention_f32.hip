// Fused fp32 self-attention forward / backward over the joint [text|region]
// sequence for gfx950.
//
// Replaces BertSelfAttention.forward after the Q/K/V projections
// (model/layer.py:80-100: transpose_for_scores, QK^T / sqrt(d), + mask, softmax,
// dropout on the probabilities, P.V, permute+view) and its autograd.  The
// [B,h,L,L] score tensor never exists in HBM: forward keeps a running
// (max, sum) per query and stores only the log-sum-exp; backward recomputes the
// probabilities from Q, K and the LSE and regenerates the dropout mask from the
// Philox counter.
//
// Everything is computed in the TRANSPOSED orientation so that no tile crosses
// LDS between products (cdna_hip_programming.md 3, "An accumulator tile as the
// next MFMA's operand"): with v_mfma_f32_32x32x2_f32 the accumulator of
//     S^T[key][query] = K . Q^T
// holds, in lane (query j, half h), keys 8g+4h+t in register 4g+t -- exactly the
// k-slot that lane-half h must feed as the B operand of step (g,t) of
//     O^T[d][query] = V^T . P^T .
// One wave owns 32 queries (fwd, dQ) or 32 keys (dK/dV); row max / sum are 15
// in-register ops plus one cross-half shuffle.  K/V (or Q/dO) tiles of 32 rows
// are staged in LDS with a 68-dword row stride: row reads are conflict-free
// ds_read_b128, column reads conflict-free ds_read_b32 off the same image.
//
// Layouts: qkv [B*L, 3H] (columns [Q|K|V], head h at h*64), ctx/dctx [B*L, H],
// lse/delta [B, nh, L].  head_dim == 64.
#include <stdlib.h>
#include "common.h"
#include "philox.h"

namespace {

constexpr int D = 64;        // head dim
constexpr int LDT = 68;      // LDS row stride (floats) of a 32 x 64 tile
constexpr int NW = 2;        // waves per workgroup (each owns 32 queries / keys)
constexpr float NEG_INF = -__builtin_huge_valf();

struct AttnArgs {
  const float* qkv;
  const float* mask;    // [B, L]
  float* ctx;           // fwd out
  float* lse;           // [B, nh, L]
  const float* dctx;
  float* dqkv;
  float* delta;         // [B, nh, L]
  unsigned short* ctx_b16;   // optional bf16 copies of ctx / dqkv (operands of bf16-resident GEMMs)
  unsigned short* dqkv_b16;
  int b16_pieces;            // 3: the copies are x3 pieces [rows][3][ld] (operands of the fp32-accurate products, csrc/gemm_split3.hip)
  unsigned short* keep_bits; // optional [B*nh, L, Lr/32, 2]: dropout keep flags of a (query, key block, lane half), written by the
                             // forward pass and read by dQ instead of a second Philox evaluation
  int prio;                  // wave priority of the L <= 192 kernels (UNITER_ATTN_PRIO, default 2)
  int keep_ready;            // the forward pass READS keep_bits (uniter_attn_keep_bits_gen filled them) instead of drawing them
  float* bias_part;          // optional [B, 3H]: per-sample column sums of dqkv (the QKV bias gradient, reduced over B later)
  const int* cu;        // [B+1] prefix sums of per-sample lengths (packed rows), or NULL: sample b owns rows b*L .. b*L+L-1
  int B, L, nh, H, Lp4; // Lp4 = roundup(L,4)/4
  float scale;
  DropCfg drop;
};

// cooperative load of rows [r0, r0+32) x 64 floats (row stride ld) into s[32][LDT]; rows >= L zeroed
__device__ __forceinline__ void stage_tile(float* s, const float* __restrict__ base, int ld, int r0,
                                           int L, int tid) {
  // 32 rows x 16 float4 = 512 float4 over NW*64 = 128 threads -> 4 each
#pragma unroll
  for (int p = 0; p < 512 / (NW * 64); ++p) {
    const int idx = tid + p * NW * 64;
    const int r = idx >> 4, c4 = idx & 15;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r0 + r < L) v = *reinterpret_cast<const f32x4*>(base + (size_t)(r0 + r) * ld + c4 * 4);
    *reinterpret_cast<f32x4*>(s + r * LDT + c4 * 4) = v;
  }
}

// per-lane operand fragments of one row (64 floats): element kb holds k = 8kb+4h .. +3
__device__ __forceinline__ void load_row_frags(f32x4 (&f)[8], const float* __restrict__ row, bool valid, int h) {
#pragma unroll
  for (int kb = 0; kb < 8; ++kb)
    f[kb] = valid ? *reinterpret_cast<const f32x4*>(row + kb * 8 + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
}

// acc[rows of tile s][lane's column] = sum_d tile[row][d] * frag[d]
__device__ __forceinline__ f32x16 tile_times_frag(const float* s, const f32x4 (&f)[8], int i, int h) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int kb = 0; kb < 8; ++kb) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(s + i * LDT + kb * 8 + 4 * h);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], f[kb][t], acc, 0, 0, 0);
  }
  return acc;
}

// out{0,1}[d][lane's column] += sum_rows tile[row][d] * w[row]   (w in accumulator layout)
__device__ __forceinline__ void tileT_times_acc(const float* s, const f32x16& w, f32x16& out0, f32x16& out1,
                                                int i, int h) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float* p = s + (8 * g + 4 * h + t) * LDT + i;
      out0 = __builtin_amdgcn_mfma_f32_32x32x2f32(p[0], w[4 * g + t], out0, 0, 0, 0);
      out1 = __builtin_amdgcn_mfma_f32_32x32x2f32(p[32], w[4 * g + t], out1, 0, 0, 0);
    }
  }
}

// store a transposed accumulator pair (rows = d, column = lane's row of the output matrix)
__device__ __forceinline__ void store_rowT(float* __restrict__ row, const f32x16& a0, const f32x16& a1,
                                           float mul, int h) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 v0 = {a0[4 * g] * mul, a0[4 * g + 1] * mul, a0[4 * g + 2] * mul, a0[4 * g + 3] * mul};
    f32x4 v1 = {a1[4 * g] * mul, a1[4 * g + 1] * mul, a1[4 * g + 2] * mul, a1[4 * g + 3] * mul};
    *reinterpret_cast<f32x4*>(row + 8 * g + 4 * h) = v0;
    *reinterpret_cast<f32x4*>(row + 32 + 8 * g + 4 * h) = v1;
  }
}

__device__ __forceinline__ void store_rowT_bf16(unsigned short* __restrict__ row, const f32x16& a0, const f32x16& a1,
                                                float mul, int h) {
  typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    bf16x4_t v0 = {(__bf16)(a0[4 * g] * mul), (__bf16)(a0[4 * g + 1] * mul), (__bf16)(a0[4 * g + 2] * mul), (__bf16)(a0[4 * g + 3] * mul)};
    bf16x4_t v1 = {(__bf16)(a1[4 * g] * mul), (__bf16)(a1[4 * g + 1] * mul), (__bf16)(a1[4 * g + 2] * mul), (__bf16)(a1[4 * g + 3] * mul)};
    *reinterpret_cast<bf16x4_t*>(row + 8 * g + 4 * h) = v0;
    *reinterpret_cast<bf16x4_t*>(row + 32 + 8 * g + 4 * h) = v1;
  }
}

// the three bf16 pieces of the row (x = x1 + x2 + x3 exactly, round-to-nearest residuals) at row, row + ps, row + 2 ps
__device__ __forceinline__ void store_rowT_x3(unsigned short* __restrict__ row, int ps, const f32x16& a0, const f32x16& a1,
                                              float mul, int h) {
  typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 r0 = {a0[4 * g] * mul, a0[4 * g + 1] * mul, a0[4 * g + 2] * mul, a0[4 * g + 3] * mul};
    f32x4 r1 = {a1[4 * g] * mul, a1[4 * g + 1] * mul, a1[4 * g + 2] * mul, a1[4 * g + 3] * mul};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const bf16x4_t v0 = {(__bf16)r0[0], (__bf16)r0[1], (__bf16)r0[2], (__bf16)r0[3]};
      const bf16x4_t v1 = {(__bf16)r1[0], (__bf16)r1[1], (__bf16)r1[2], (__bf16)r1[3]};
      *reinterpret_cast<bf16x4_t*>(row + p * ps + 8 * g + 4 * h) = v0;
      *reinterpret_cast<bf16x4_t*>(row + p * ps + 32 + 8 * g + 4 * h) = v1;
#pragma unroll
      for (int e = 0; e < 4; ++e) { r0[e] -= (float)v0[e]; r1[e] -= (float)v1[e]; }
    }
  }
}
// bf16 copy or x3 pieces of one 64-wide head slice of row `r` of a [rows][ld] tensor
__device__ __forceinline__ void store_rowT_copy(unsigned short* __restrict__ base, int pieces, size_t r, int ld, int col,
                                                const f32x16& a0, const f32x16& a1, float mul, int h) {
  if (pieces == 3) store_rowT_x3(base + r * 3 * ld + col, ld, a0, a1, mul, h);
  else store_rowT_bf16(base + r * ld + col, a0, a1, mul, h);
}

__device__ __forceinline__ void stage_mask_bias(float* mb, const float* __restrict__ mask_row, int k0, int L, int tid) {
  if (tid < 32) {
    const int k = k0 + tid;
    mb[tid] = k < L ? (1.0f - mask_row[k]) * -10000.0f : NEG_INF;
  }
}

// ---------------------------------------------------------------- forward ---
__global__ __launch_bounds__(NW * 64) void attn_fwd_kernel(const AttnArgs a) {
  __shared__ __attribute__((aligned(16))) float Ks[32 * LDT];
  __shared__ __attribute__((aligned(16))) float Vs[32 * LDT];
  __shared__ __attribute__((aligned(16))) float mb[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.nh, head = bh - b * a.nh;
  const int ld = 3 * a.H;
  const float* base = a.qkv + (size_t)b * a.L * ld + head * D;
  const int q = (blockIdx.x * NW + wave) * 32 + i;
  const bool vq = q < a.L;

  f32x4 qf[8];
  load_row_frags(qf, base + (size_t)q * ld, vq, h);

  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = NEG_INF, l_run = 0.f;

  for (int k0 = 0; k0 < a.L; k0 += 32) {
    __syncthreads();
    stage_tile(Ks, base + a.H, ld, k0, a.L, tid);
    stage_tile(Vs, base + 2 * a.H, ld, k0, a.L, tid);
    stage_mask_bias(mb, a.mask + (size_t)b * a.L, k0, a.L, tid);
    __syncthreads();

    f32x16 s = tile_times_frag(Ks, qf, i, h);     // S^T[key][query]
    float mx = NEG_INF;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 bias = *reinterpret_cast<const f32x4*>(mb + 8 * g + 4 * h);
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
      const uint64_t grow = ((uint64_t)bh * a.L + q) * a.Lp4 + (k0 >> 2);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float m4[4];
        drop_mult4(a.drop, grow + 2 * g + h, m4);
#pragma unroll
        for (int t = 0; t < 4; ++t) s[4 * g + t] *= m4[t];
      }
    }
    tileT_times_acc(Vs, s, o0, o1, i, h);         // O^T[d][query] += V^T . Pd^T
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (vq) {
    store_rowT(a.ctx + ((size_t)b * a.L + q) * a.H + head * D, o0, o1, 1.0f / l_tot, h);
    if (h == 0 && a.lse) a.lse[(size_t)bh * a.L + q] = m_run + __logf(l_tot);
  }
}

// ------------------------------------------------------- backward: dQ, delta ---
__global__ __launch_bounds__(NW * 64) void attn_bwd_dq_kernel(const AttnArgs a) {
  __shared__ __attribute__((aligned(16))) float Ks[32 * LDT];
  __shared__ __attribute__((aligned(16))) float Vs[32 * LDT];
  __shared__ __attribute__((aligned(16))) float mb[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.nh, head = bh - b * a.nh;
  const int ld = 3 * a.H;
  const float* base = a.qkv + (size_t)b * a.L * ld + head * D;
  const int q = (blockIdx.x * NW + wave) * 32 + i;
  const bool vq = q < a.L;

  f32x4 qf[8], dof[8];
  load_row_frags(qf, base + (size_t)q * ld, vq, h);
  load_row_frags(dof, a.dctx + ((size_t)b * a.L + q) * a.H + head * D, vq, h);
  float delta = 0.f;
  {
    f32x4 of[8];
    load_row_frags(of, a.ctx + ((size_t)b * a.L + q) * a.H + head * D, vq, h);
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
      for (int t = 0; t < 4; ++t) delta += of[kb][t] * dof[kb][t];
    delta += __shfl_xor(delta, 32, 64);
    if (vq && h == 0) a.delta[(size_t)bh * a.L + q] = delta;
  }
  const float lse = vq ? a.lse[(size_t)bh * a.L + q] : 0.f;

  f32x16 dq0, dq1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }

  for (int k0 = 0; k0 < a.L; k0 += 32) {
    __syncthreads();
    stage_tile(Ks, base + a.H, ld, k0, a.L, tid);
    stage_tile(Vs, base + 2 * a.H, ld, k0, a.L, tid);
    stage_mask_bias(mb, a.mask + (size_t)b * a.L, k0, a.L, tid);
    __syncthreads();

    f32x16 s = tile_times_frag(Ks, qf, i, h);      // S^T[key][query]
    f32x16 dp = tile_times_frag(Vs, dof, i, h);    // dPd^T[key][query] = V . dO^T
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 bias = *reinterpret_cast<const f32x4*>(mb + 8 * g + 4 * h);
      float m4[4] = {1.f, 1.f, 1.f, 1.f};
      if (a.drop.active && vq)
        drop_mult4(a.drop, ((uint64_t)bh * a.L + q) * a.Lp4 + (k0 >> 2) + 2 * g + h, m4);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float p = __expf(s[4 * g + t] * a.scale + bias[t] - lse);
        s[4 * g + t] = p * (dp[4 * g + t] * m4[t] - delta) * a.scale;   // dS^T * scale
      }
    }
    tileT_times_acc(Ks, s, dq0, dq1, i, h);        // dQ^T[d][query] += K^T . dS^T
  }
  if (vq) store_rowT(a.dqkv + ((size_t)b * a.L + q) * ld + head * D, dq0, dq1, 1.0f, h);
}

// ------------------------------------------------------ backward: dK, dV ---
__global__ __launch_bounds__(NW * 64) void attn_bwd_dkv_kernel(const AttnArgs a) {
  __shared__ __attribute__((aligned(16))) float Qs[32 * LDT];
  __shared__ __attribute__((aligned(16))) float dOs[32 * LDT];
  __shared__ __attribute__((aligned(16))) float lse_s[32];
  __shared__ __attribute__((aligned(16))) float del_s[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5, u = lane & 3;
  const int bh = blockIdx.y, b = bh / a.nh, head = bh - b * a.nh;
  const int ld = 3 * a.H;
  const float* base = a.qkv + (size_t)b * a.L * ld + head * D;
  const int key = (blockIdx.x * NW + wave) * 32 + i;
  const bool vk = key < a.L;

  f32x4 kf[8], vf[8];
  load_row_frags(kf, base + a.H + (size_t)key * ld, vk, h);
  load_row_frags(vf, base + 2 * a.H + (size_t)key * ld, vk, h);
  const float bias = vk ? (1.0f - a.mask[(size_t)b * a.L + key]) * -10000.0f : NEG_INF;

  f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk0[r] = 0.f; dk1[r] = 0.f; dv0[r] = 0.f; dv1[r] = 0.f; }

  for (int q0 = 0; q0 < a.L; q0 += 32) {
    __syncthreads();
    stage_tile(Qs, base, ld, q0, a.L, tid);
    stage_tile(dOs, a.dctx + (size_t)b * a.L * a.H + head * D, a.H, q0, a.L, tid);
    if (tid < 32) {
      const int qq = q0 + tid;
      lse_s[tid] = qq < a.L ? a.lse[(size_t)bh * a.L + qq] : -NEG_INF;   // +inf -> p = 0
      del_s[tid] = qq < a.L ? a.delta[(size_t)bh * a.L + qq] : 0.f;
    }
    __syncthreads();

    f32x16 s = tile_times_frag(Qs, kf, i, h);      // S[query][key]
    f32x16 dp = tile_times_frag(dOs, vf, i, h);    // dPd[query][key] = dO . V^T
    f32x16 pd;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 lse4 = *reinterpret_cast<const f32x4*>(lse_s + 8 * g + 4 * h);
      const f32x4 del4 = *reinterpret_cast<const f32x4*>(del_s + 8 * g + 4 * h);
      float m4[4] = {1.f, 1.f, 1.f, 1.f};
      if (a.drop.active) {
        // the 4 lanes of a quad hold keys 4c..4c+3; quad-lane u draws the Philox group of
        // (query 8g+4h+u, keys 4c..4c+3) and the keep bits are exchanged across the quad
        const int qq = q0 + 8 * g + 4 * h + u;
        const u32x4 w = drop_words(a.drop, ((uint64_t)bh * a.L + qq) * a.Lp4 + (key >> 2));
        const int bits = (w.x >= a.drop.thresh ? 1 : 0) | (w.y >= a.drop.thresh ? 2 : 0) |
                         (w.z >= a.drop.thresh ? 4 : 0) | (w.w >= a.drop.thresh ? 8 : 0);
        const int b0 = __builtin_amdgcn_mov_dpp(bits, 0x00, 0xf, 0xf, true);
        const int b1 = __builtin_amdgcn_mov_dpp(bits, 0x55, 0xf, 0xf, true);
        const int b2 = __builtin_amdgcn_mov_dpp(bits, 0xAA, 0xf, 0xf, true);
        const int b3 = __builtin_amdgcn_mov_dpp(bits, 0xFF, 0xf, 0xf, true);
        m4[0] = ((b0 >> u) & 1) ? a.drop.scale : 0.f;
        m4[1] = ((b1 >> u) & 1) ? a.drop.scale : 0.f;
        m4[2] = ((b2 >> u) & 1) ? a.drop.scale : 0.f;
        m4[3] = ((b3 >> u) & 1) ? a.drop.scale : 0.f;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float p = __expf(s[4 * g + t] * a.scale + bias - lse4[t]);
        pd[4 * g + t] = p * m4[t];
        s[4 * g + t] = p * (dp[4 * g + t] * m4[t] - del4[t]) * a.scale;   // dS * scale
      }
    }
    tileT_times_acc(dOs, pd, dv0, dv1, i, h);      // dV^T[d][key] += dO^T . Pd
    tileT_times_acc(Qs, s, dk0, dk1, i, h);        // dK^T[d][key] += Q^T . dS
  }
  if (vk) {
    float* row = a.dqkv + ((size_t)b * a.L + key) * ld + head * D;
    store_rowT(row + a.H, dk0, dk1, 1.0f, h);
    store_rowT(row + 2 * a.H, dv0, dv1, 1.0f, h);
  }
}


// ---------------------------------------------------------------------------
// Resident variants (L <= 256): one workgroup per (batch, head) with one wave per 32-row block.
// The whole K and V (or Q and dO) of the head -- 2 x L x 64 fp32 = 84 KB at L = 164 -- are staged
// ONCE into the 160 KB LDS; the key/query loop then runs without a single barrier or global load.
// The streaming kernels above (32-row tiles re-staged per block, two barriers each) remain the
// general path for longer sequences.
// ---------------------------------------------------------------------------
extern __shared__ __attribute__((aligned(16))) float dyn_smem[];

// Two [L, 64] operands -> LDS images [Lr][LDT] (rows >= L zero).  IT = Lr * 16 / blockDim.x 16-byte pieces per thread and
// operand (4 with the split kernels' Lr * 4 threads, 8 with the resident kernels' Lr * 2): ALL 2 * IT loads are issued
// before the first LDS write (and the split kernels issue their row-fragment loads in between) -- one piece at a time (load, wait, write) left 8 memory latencies in a row, 9.8 of the
// 57 us of the dQ kernel (tests/tools/attn_phase_lab.py).
template <int IT>
struct Stage2 {
  f32x4 v1[IT], v2[IT];
  __device__ __forceinline__ void load(const float* __restrict__ base1, int ld1, const float* __restrict__ base2, int ld2,
                                       int L, int Lr, int tid, int nthr) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = tid + it * nthr, r = idx >> 4, c4 = idx & 15;
      v1[it] = f32x4{0.f, 0.f, 0.f, 0.f};
      v2[it] = v1[it];
      if (idx < Lr * 16 && r < L) {
        v1[it] = *reinterpret_cast<const f32x4*>(base1 + (size_t)r * ld1 + c4 * 4);
        v2[it] = *reinterpret_cast<const f32x4*>(base2 + (size_t)r * ld2 + c4 * 4);
      }
    }
  }
  __device__ __forceinline__ void store(float* s1, float* s2, int Lr, int tid, int nthr) const {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = tid + it * nthr, r = idx >> 4, c4 = idx & 15;
      if (idx < Lr * 16) {
        *reinterpret_cast<f32x4*>(s1 + r * LDT + c4 * 4) = v1[it];
        *reinterpret_cast<f32x4*>(s2 + r * LDT + c4 * 4) = v2[it];
      }
    }
  }
};
template <int IT>
__device__ __forceinline__ void stage_rows2(float* s1, const float* __restrict__ base1, int ld1, float* s2,
                                            const float* __restrict__ base2, int ld2, int L, int Lr, int tid, int nthr) {
  Stage2<IT> st;
  st.load(base1, ld1, base2, ld2, L, Lr, tid, nthr);
  st.store(s1, s2, Lr, tid, nthr);
}

__global__ __launch_bounds__(512) void attn_fwd_res_kernel(const AttnArgs a, int Lr) {
  float* Ks = dyn_smem;
  float* Vs = Ks + Lr * LDT;
  float* mb = Vs + Lr * LDT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
  const int i = lane & 31, h = lane >> 5;
  const int bh = blockIdx.x, b = bh / a.nh, head = bh - b * a.nh;
  const int ld = 3 * a.H;
  const float* base = a.qkv + (size_t)b * a.L * ld + head * D;
  stage_rows2<8>(Ks, base + a.H, ld, Vs, base + 2 * a.H, ld, a.L, Lr, tid, nthr);
  for (int k = tid; k < Lr; k += nthr)
    mb[k] = k < a.L ? (1.0f - a.mask[(size_t)b * a.L + k]) * -10000.0f : NEG_INF;
  const int q = wave * 32 + i;
  const bool vq = q < a.L;
  f32x4 qf[8];
  load_row_frags(qf, base + (size_t)q * ld, vq, h);
  __syncthreads();

  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = NEG_INF, l_run = 0.f;
  for (int k0 = 0; k0 < a.L; k0 += 32) {
    f32x16 s = tile_times_frag(Ks + k0 * LDT, qf, i, h);
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
      const uint64_t grow = ((uint64_t)bh * a.L + q) * a.Lp4 + (k0 >> 2);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float m4[4];
        drop_mult4(a.drop, grow + 2 * g + h, m4);
#pragma unroll
        for (int t = 0; t < 4; ++t) s[4 * g + t] *= m4[t];
      }
    }
    tileT_times_acc(Vs + k0 * LDT, s, o0, o1, i, h);
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (vq) {
    store_rowT(a.ctx + ((size_t)b * a.L + q) * a.H + head * D, o0, o1, 1.0f / l_tot, h);
    if (h == 0 && a.lse) a.lse[(size_t)bh * a.L + q] = m_run + __logf(l_tot);
  }
}

__global__ __launch_bounds__(512) void attn_bwd_dq_res_kernel(const AttnArgs a, int Lr) {
  float* Ks = dyn_smem;
  float* Vs = Ks + Lr * LDT;
  float* mb = Vs + Lr * LDT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
  const int i = lane & 31, h = lane >> 5;
  const int bh = blockIdx.x, b = bh / a.nh, head = bh - b * a.nh;
  const int ld = 3 * a.H;
  const float* base = a.qkv + (size_t)b * a.L * ld + head * D;
  stage_rows2<8>(Ks, base + a.H, ld, Vs, base + 2 * a.H, ld, a.L, Lr, tid, nthr);
  for (int k = tid; k < Lr; k += nthr)
    mb[k] = k < a.L ? (1.0f - a.mask[(size_t)b * a.L + k]) * -10000.0f : NEG_INF;
  const int q = wave * 32 + i;
  const bool vq = q < a.L;
  f32x4 qf[8], dof[8];
  load_row_frags(qf, base + (size_t)q * ld, vq, h);
  load_row_frags(dof, a.dctx + ((size_t)b * a.L + q) * a.H + head * D, vq, h);
  float delta = 0.f;
  {
    f32x4 of[8];
    load_row_frags(of, a.ctx + ((size_t)b * a.L + q) * a.H + head * D, vq, h);
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
      for (int t = 0; t < 4; ++t) delta += of[kb][t] * dof[kb][t];
    delta += __shfl_xor(delta, 32, 64);
    if (vq && h == 0) a.delta[(size_t)bh * a.L + q] = delta;
  }
  const float lse = vq ? a.lse[(size_t)bh * a.L + q] : 0.f;
  __syncthreads();

  f32x16 dq0, dq1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }
  for (int k0 = 0; k0 < a.L; k0 += 32) {
    f32x16 s = tile_times_frag(Ks + k0 * LDT, qf, i, h);
    f32x16 dp = tile_times_frag(Vs + k0 * LDT, dof, i, h);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 bias = *reinterpret_cast<const f32x4*>(mb + k0 + 8 * g + 4 * h);
      float m4[4] = {1.f, 1.f, 1.f, 1.f};
      if (a.drop.active && vq)
        drop_mult4(a.drop, ((uint64_t)bh * a.L + q) * a.Lp4 + (k0 >> 2) + 2 * g + h, m4);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float p = __expf(s[4 * g + t] * a.scale + bias[t] - lse);
        s[4 * g + t] = p * (dp[4 * g + t] * m4[t] - delta) * a.scale;
      }
    }
    tileT_times_acc(Ks + k0 * LDT, s, dq0, dq1, i, h);
  }
  if (vq) store_rowT(a.dqkv + ((size_t)b * a.L + q) * ld + head * D, dq0, dq1, 1.0f, h);
}

__global__ __launch_bounds__(512) void attn_bwd_dkv_res_kernel(const AttnArgs a, int Lr) {
  float* Qs = dyn_smem;
  float* dOs = Qs + Lr * LDT;
  float* lse_s = dOs + Lr * LDT;
  float* del_s = lse_s + Lr;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
  const int i = lane & 31, h = lane >> 5, u = lane & 3;
  const int bh = blockIdx.x, b = bh / a.nh, head = bh - b * a.nh;
  const int ld = 3 * a.H;
  const float* base = a.qkv + (size_t)b * a.L * ld + head * D;
  stage_rows2<8>(Qs, base, ld, dOs, a.dctx + (size_t)b * a.L * a.H + head * D, a.H, a.L, Lr, tid, nthr);
  for (int k = tid; k < Lr; k += nthr) {
    lse_s[k] = k < a.L ? a.lse[(size_t)bh * a.L + k] : -NEG_INF;
    del_s[k] = k < a.L ? a.delta[(size_t)bh * a.L + k] : 0.f;
  }
  const int key = wave * 32 + i;
  const bool vk = key < a.L;
  f32x4 kf[8], vf[8];
  load_row_frags(kf, base + a.H + (size_t)key * ld, vk, h);
  load_row_frags(vf, base + 2 * a.H + (size_t)key * ld, vk, h);
  const float bias = vk ? (1.0f - a.mask[(size_t)b * a.L + key]) * -10000.0f : NEG_INF;
  __syncthreads();

  f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk0[r] = 0.f; dk1[r] = 0.f; dv0[r] = 0.f; dv1[r] = 0.f; }
  for (int q0 = 0; q0 < a.L; q0 += 32) {
    f32x16 s = tile_times_frag(Qs + q0 * LDT, kf, i, h);
    f32x16 dp = tile_times_frag(dOs + q0 * LDT, vf, i, h);
    f32x16 pd;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 lse4 = *reinterpret_cast<const f32x4*>(lse_s + q0 + 8 * g + 4 * h);
      const f32x4 del4 = *reinterpret_cast<const f32x4*>(del_s + q0 + 8 * g + 4 * h);
      float m4[4] = {1.f, 1.f, 1.f, 1.f};
      if (a.drop.active) {
        const int qq = q0 + 8 * g + 4 * h + u;
        const u32x4 w = drop_words(a.drop, ((uint64_t)bh * a.L + qq) * a.Lp4 + (key >> 2));
        const int bits = (w.x >= a.drop.thresh ? 1 : 0) | (w.y >= a.drop.thresh ? 2 : 0) |
                         (w.z >= a.drop.thresh ? 4 : 0) | (w.w >= a.drop.thresh ? 8 : 0);
        const int b0 = __builtin_amdgcn_mov_dpp(bits, 0x00, 0xf, 0xf, true);
        const int b1 = __builtin_amdgcn_mov_dpp(bits, 0x55, 0xf, 0xf, true);
        const int b2 = __builtin_amdgcn_mov_dpp(bits, 0xAA, 0xf, 0xf, true);
        const int b3 = __builtin_amdgcn_mov_dpp(bits, 0xFF, 0xf, 0xf, true);
        m4[0] = ((b0 >> u) & 1) ? a.drop.scale : 0.f;
        m4[1] = ((b1 >> u) & 1) ? a.drop.scale : 0.f;
        m4[2] = ((b2 >> u) & 1) ? a.drop.scale : 0.f;
        m4[3] = ((b3 >> u) & 1) ? a.drop.scale : 0.f;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float p = __expf(s[4 * g + t] * a.scale + bias - lse4[t]);
        pd[4 * g + t] = p * m4[t];
        s[4 * g + t] = p * (dp[4 * g + t] * m4[t] - del4[t]) * a.scale;
      }
    }
    tileT_times_acc(dOs + q0 * LDT, pd, dv0, dv1, i, h);
    tileT_times_acc(Qs + q0 * LDT, s, dk0, dk1, i, h);
  }
  if (vk) {
    float* row = a.dqkv + ((size_t)b * a.L + key) * ld + head * D;
    store_rowT(row + a.H, dk0, dk1, 1.0f, h);
    store_rowT(row + 2 * a.H, dv0, dv1, 1.0f, h);
  }
}

// ---------------------------------------------------------------------------
// Split variants (L <= 192): still one workgroup per (batch, head) with K/V (or Q/dO) resident in
// LDS, but TWO waves per 32-row block, each covering half of the loop range -- 12 waves at L = 164,
// three per SIMD, instead of six waves spread 2/2/1/1.  Partial results meet in LDS (the staged
// operands are dead by then): forward merges the two (max, sum, O) triples, dQ / dK / dV add.
// The dQ kernel also writes the dropped probabilities Pd and the score gradients dS to a scratch
// tensor in HBM ([B*nh][key][query], 2 x 28 MB at B = 16): the dK/dV kernel then is two plain
// products (dV = Pd^T dO, dK = dS^T Q) -- half the MFMAs of recomputing S and dP, no exp, no Philox.
// ---------------------------------------------------------------------------
constexpr int SPLIT_MAX_LR = 192;
#ifdef ATTN_STAMPS      // measurement builds only (tests/tools/attn_phase_lab.py): per-workgroup phase clocks
__device__ unsigned long long g_attn_stamps[3 * 8 * 1024];
#define ASTAMP(kern, k) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_attn_stamps[((kern) * 1024 + blockIdx.x) * 8 + (k)] = (unsigned long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ASTAMP(kern, k) do { } while (0)
#endif
constexpr int XROW = 34;             // floats exchanged per lane: 32 accumulators + (m, l)

// Packed (varlen) batches: with a.cu set, sample b owns rows cu[b] .. cu[b+1]-1 of qkv / ctx / dqkv,
// every key of a sample is valid (no mask) and waves whose 32-row block lies beyond the sample's
// length only keep the barriers company.  lse / delta / the scratch stay indexed by the maximum length.
// sum over the 32 lanes of each wave half
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// column sums of a transposed accumulator pair over the wave's valid lanes -> red[0..63] (LDS, pre-zeroed)
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

struct SampleSpan { int row0, Lb, nb; };
__device__ __forceinline__ SampleSpan sample_span(const AttnArgs& a, int b) {
  SampleSpan s;
  s.row0 = a.cu ? a.cu[b] : b * a.L;
  s.Lb = a.cu ? a.cu[b + 1] - s.row0 : a.L;
  s.nb = (s.Lb + 31) >> 5;
  return s;
}
__device__ __forceinline__ void stage_mask(float* mb, const AttnArgs& a, int b, int Lb, int Lr, int tid, int nthr) {
  for (int k = tid; k < Lr; k += nthr)
    mb[k] = k < Lb ? (a.mask ? (1.0f - a.mask[(size_t)b * a.L + k]) * -10000.0f : 0.f) : NEG_INF;
}

__global__ __launch_bounds__(768) void attn_fwd_split_kernel(const AttnArgs a, int Lr) {
  set_wave_prio(a.prio);      // critical path of the step: ahead of the side stream's weight gradients (common.h)
  float* Ks = dyn_smem;
  float* Vs = Ks + Lr * LDT;
  float* mb = Vs + Lr * LDT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
  const int i = lane & 31, h = lane >> 5;
  const int nblk = Lr >> 5;
  const int qb = wave % nblk, half = wave / nblk;
  const int bh = blockIdx.x, b = bh / a.nh, head = bh - b * a.nh;
  const SampleSpan sp = sample_span(a, b);
  const int Lb = sp.Lb;
  const int ld = 3 * a.H;
  const float* base = a.qkv + (size_t)sp.row0 * ld + head * D;
  Stage2<4> stg;
  stg.load(base + a.H, ld, base + 2 * a.H, ld, Lb, Lr, tid, nthr);
  const int q = qb * 32 + i;
  const bool vq = q < Lb;
  f32x4 qf[8];
  load_row_frags(qf, base + (size_t)q * ld, vq, h);
  stage_mask(mb, a, b, Lb, Lr, tid, nthr);
  stg.store(Ks, Vs, Lr, tid, nthr);
  __syncthreads();

  const int kmid = ((sp.nb + 1) >> 1) * 32;
  const int kbeg = half ? kmid : 0, kend = qb < sp.nb ? (half ? sp.nb * 32 : kmid) : 0;
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = NEG_INF, l_run = 0.f;
  for (int k0 = kbeg; k0 < kend; k0 += 32) {
    f32x16 s = tile_times_frag(Ks + k0 * LDT, qf, i, h);
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
        bits = a.keep_bits[kidx];       // drawn ahead by attn_keep_bits_kernel: ten dependent Philox rounds per 4 keys cost a
                                        // 3-waves-per-SIMD attention kernel far more than a full-occupancy elementwise pass
      } else {
        const uint64_t grow = ((uint64_t)bh * a.L + q) * a.Lp4 + (k0 >> 2);
#pragma unroll
        for (int g = 0; g < 4; ++g) bits |= drop_bits4(a.drop, grow + 2 * g + h) << (4 * g);
        if (a.keep_bits) a.keep_bits[kidx] = (unsigned short)bits;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] *= ((bits >> r) & 1u) ? a.drop.scale : 0.f;
    }
    tileT_times_acc(Vs + k0 * LDT, s, o0, o1, i, h);
  }
  // second-half waves hand (O, m, l) to their partner through LDS
  __syncthreads();
  float* xb = dyn_smem + (size_t)qb * XROW * 64;
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
    // a half without any key (one key block) carries m = -inf, l = 0: its weight is exp(-inf) = 0
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
    store_rowT(a.ctx + ((size_t)sp.row0 + q) * a.H + head * D, o0, o1, 1.0f / l_tot, h);
    if (a.ctx_b16) store_rowT_copy(a.ctx_b16, a.b16_pieces, (size_t)sp.row0 + q, a.H, head * D, o0, o1, 1.0f / l_tot, h);
    if (h == 0 && a.lse) a.lse[(size_t)bh * a.L + q] = m_run + __logf(l_tot);
  }
}

// (no __restrict__ on the scratch pointers of the two bodies: the fused kernel runs both on the same memory)
__device__ __forceinline__ void bwd_dq_split_body(const AttnArgs& a, int Lr, float* pd_ws, float* ds_ws) {
  float* Ks = dyn_smem;
  float* Vs = Ks + Lr * LDT;
  float* mb = Vs + Lr * LDT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
  const int i = lane & 31, h = lane >> 5;
  const int nblk = Lr >> 5;
  const int qb = wave % nblk, half = wave / nblk;
  const int bh = blockIdx.x, b = bh / a.nh, head = bh - b * a.nh;
  const SampleSpan sp = sample_span(a, b);
  const int Lb = sp.Lb;
  const int ld = 3 * a.H;
  const float* base = a.qkv + (size_t)sp.row0 * ld + head * D;
  ASTAMP(1, 0);
  Stage2<4> stg;
  stg.load(base + a.H, ld, base + 2 * a.H, ld, Lb, Lr, tid, nthr);
  const int q = qb * 32 + i;
  const bool vq = q < Lb;
  f32x4 qf[8], dof[8];
  load_row_frags(qf, base + (size_t)q * ld, vq, h);
  load_row_frags(dof, a.dctx + ((size_t)sp.row0 + q) * a.H + head * D, vq, h);
  float delta = 0.f;
  {
    f32x4 of[8];
    load_row_frags(of, a.ctx + ((size_t)sp.row0 + q) * a.H + head * D, vq, h);
    stage_mask(mb, a, b, Lb, Lr, tid, nthr);
    for (int t = tid; t < 192; t += nthr) dyn_smem[2 * Lr * LDT + 2 * Lr + t] = 0.f;      // (a 32-row workgroup has only 128 threads)
    stg.store(Ks, Vs, Lr, tid, nthr);
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
      for (int t = 0; t < 4; ++t) delta += of[kb][t] * dof[kb][t];
    delta += __shfl_xor(delta, 32, 64);
    if (vq && h == 0 && half == 0) a.delta[(size_t)bh * a.L + q] = delta;
  }
  // an out-of-range query has lse = +inf: every probability (and so Pd, dS) of its column is 0
  const float lse = vq ? a.lse[(size_t)bh * a.L + q] : -NEG_INF;
  ASTAMP(1, 1);
  __syncthreads();
  ASTAMP(1, 2);

  // blocks [0, nb) x [0, nb) of the scratch are written in full (zeros where query or key >= Lb):
  // that is exactly what the dK/dV kernel reads
  const int kmid = ((sp.nb + 1) >> 1) * 32;
  const int kbeg = half ? kmid : 0, kend = qb < sp.nb ? (half ? sp.nb * 32 : kmid) : 0;
  float* pdw = pd_ws + (size_t)bh * Lr * Lr + qb * 32 + i;
  float* dsw = ds_ws + (size_t)bh * Lr * Lr + qb * 32 + i;
  f32x16 dq0, dq1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }
  for (int k0 = kbeg; k0 < kend; k0 += 32) {
    f32x16 s = tile_times_frag(Ks + k0 * LDT, qf, i, h);
    f32x16 dp = tile_times_frag(Vs + k0 * LDT, dof, i, h);
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
        const size_t off = (size_t)(k0 + 8 * g + 4 * h + t) * Lr;     // row = key, 32 lanes = 128 contiguous bytes
        pdw[off] = pdv;
        dsw[off] = dsv;
      }
    }
    tileT_times_acc(Ks + k0 * LDT, s, dq0, dq1, i, h);
  }
  ASTAMP(1, 3);
  __syncthreads();
  ASTAMP(1, 4);
  float* xb = dyn_smem + (size_t)qb * XROW * 64;
  if (half) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { xb[r * 64 + lane] = dq0[r]; xb[(16 + r) * 64 + lane] = dq1[r]; }
  }
  __syncthreads();
  ASTAMP(1, 5);
  float* red = dyn_smem + 2 * Lr * LDT + 2 * Lr;        // 192 floats behind everything else (res_lds_bytes)
  if (!half) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq0[r] += xb[r * 64 + lane]; dq1[r] += xb[(16 + r) * 64 + lane]; }
    if (vq) {
      if (a.dqkv) store_rowT(a.dqkv + ((size_t)sp.row0 + q) * ld + head * D, dq0, dq1, 1.0f, h);
      if (a.dqkv_b16) store_rowT_copy(a.dqkv_b16, a.b16_pieces, (size_t)sp.row0 + q, ld, head * D, dq0, dq1, 1.0f, h);
    }
    if (a.bias_part) acc_colsum(red, dq0, dq1, vq, i, h);
  }
  if (a.bias_part) {
    __syncthreads();
    if (tid < 64) a.bias_part[(size_t)b * 3 * a.H + head * D + tid] = red[tid];
  }
  ASTAMP(1, 6);
}

__global__ __launch_bounds__(768) void attn_bwd_dq_split_kernel(const AttnArgs a, int Lr, float* __restrict__ pd_ws,
                                                                float* __restrict__ ds_ws) {
  set_wave_prio(a.prio);      // critical path of the step: ahead of the side stream's weight gradients (common.h)
  bwd_dq_split_body(a, Lr, pd_ws, ds_ws);
}

// dV^T[d][key] = sum_q dO[q][d] Pd[q][key],  dK^T[d][key] = sum_q Q[q][d] dS[q][key]
__device__ __forceinline__ void bwd_dkv_split_body(const AttnArgs& a, int Lr, const float* pd_ws, const float* ds_ws) {
  float* Qs = dyn_smem;
  float* dOs = Qs + Lr * LDT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
  const int i = lane & 31, h = lane >> 5;
  const int nblk = Lr >> 5;
  const int kb = wave % nblk, half = wave / nblk;
  const int bh = blockIdx.x, b = bh / a.nh, head = bh - b * a.nh;
  const SampleSpan sp = sample_span(a, b);
  const int Lb = sp.Lb;
  const int ld = 3 * a.H;
  const float* base = a.qkv + (size_t)sp.row0 * ld + head * D;
  Stage2<4> stg;
  stg.load(base, ld, a.dctx + (size_t)sp.row0 * a.H + head * D, a.H, Lb, Lr, tid, nthr);
  for (int t = tid; t < 192; t += nthr) dyn_smem[2 * Lr * LDT + 2 * Lr + t] = 0.f;      // (a 32-row workgroup has only 128 threads)
  const int key = kb * 32 + i;
  const bool vk = key < Lb;
  const int qmid = ((sp.nb + 1) >> 1) * 32;
  const int qbeg = half ? qmid : 0, qend = kb < sp.nb ? (half ? sp.nb * 32 : qmid) : 0;
  // lane (key i, half h) needs, for q-tile q0, the 4 consecutive queries q0 + 8g + 4h .. +3 of its key row
  const float* pdr = pd_ws + ((size_t)bh * Lr + key) * Lr + 4 * h;
  const float* dsr = ds_ws + ((size_t)bh * Lr + key) * Lr + 4 * h;
  f32x4 wp[4], wd[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) { wp[g] = f32x4{0.f, 0.f, 0.f, 0.f}; wd[g] = wp[g]; }
  if (qbeg < qend) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      wp[g] = *reinterpret_cast<const f32x4*>(pdr + qbeg + 8 * g);
      wd[g] = *reinterpret_cast<const f32x4*>(dsr + qbeg + 8 * g);
    }
  }
  stg.store(Qs, dOs, Lr, tid, nthr);
  __syncthreads();

  f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk0[r] = 0.f; dk1[r] = 0.f; dv0[r] = 0.f; dv1[r] = 0.f; }
  for (int q0 = qbeg; q0 < qend; q0 += 32) {
    f32x16 pd, ds;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int t = 0; t < 4; ++t) { pd[4 * g + t] = wp[g][t]; ds[4 * g + t] = wd[g][t]; }
    if (q0 + 32 < qend) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        wp[g] = *reinterpret_cast<const f32x4*>(pdr + q0 + 32 + 8 * g);
        wd[g] = *reinterpret_cast<const f32x4*>(dsr + q0 + 32 + 8 * g);
      }
    }
    tileT_times_acc(dOs + q0 * LDT, pd, dv0, dv1, i, h);
    tileT_times_acc(Qs + q0 * LDT, ds, dk0, dk1, i, h);
  }
  __syncthreads();
  float* xb = dyn_smem + (size_t)kb * 64 * 64;
  if (half) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      xb[r * 64 + lane] = dk0[r]; xb[(16 + r) * 64 + lane] = dk1[r];
      xb[(32 + r) * 64 + lane] = dv0[r]; xb[(48 + r) * 64 + lane] = dv1[r];
    }
  }
  __syncthreads();
  float* red = dyn_smem + 2 * Lr * LDT + 2 * Lr;
  if (!half) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      dk0[r] += xb[r * 64 + lane]; dk1[r] += xb[(16 + r) * 64 + lane];
      dv0[r] += xb[(32 + r) * 64 + lane]; dv1[r] += xb[(48 + r) * 64 + lane];
    }
    if (vk) {
      if (a.dqkv) {
        float* row = a.dqkv + ((size_t)sp.row0 + key) * ld + head * D;
        store_rowT(row + a.H, dk0, dk1, 1.0f, h);
        store_rowT(row + 2 * a.H, dv0, dv1, 1.0f, h);
      }
      if (a.dqkv_b16) {
        store_rowT_copy(a.dqkv_b16, a.b16_pieces, (size_t)sp.row0 + key, ld, a.H + head * D, dk0, dk1, 1.0f, h);
        store_rowT_copy(a.dqkv_b16, a.b16_pieces, (size_t)sp.row0 + key, ld, 2 * a.H + head * D, dv0, dv1, 1.0f, h);
      }
    }
    if (a.bias_part) { acc_colsum(red + 64, dk0, dk1, vk, i, h); acc_colsum(red + 128, dv0, dv1, vk, i, h); }
  }
  if (a.bias_part) {
    __syncthreads();
    for (int t = tid; t < 128; t += nthr)
      a.bias_part[(size_t)b * 3 * a.H + (1 + (t >> 6)) * a.H + head * D + (t & 63)] = red[64 + t];
  }
}

__global__ __launch_bounds__(768) void attn_bwd_dkv_split_kernel(const AttnArgs a, int Lr,
                                                                 const float* __restrict__ pd_ws,
                                                                 const float* __restrict__ ds_ws) {
  set_wave_prio(a.prio);      // critical path of the step: ahead of the side stream's weight gradients (common.h)
  bwd_dkv_split_body(a, Lr, pd_ws, ds_ws);
}

// Both passes in ONE launch: a workgroup's dK / dV pass reads only what its own dQ pass left in the scratch (the Pd / dS blocks
// of its (batch, head)), so it can follow at once -- behind a workgroup-wide barrier and a fence that makes the scratch stores
// visible inside the workgroup -- instead of behind the slowest workgroup of a first launch.  In the training step the pair then takes a CU ONCE: a
// 12-wave, 107-KB workgroup needs an empty CU, and between two launches the other stream's GEMM workgroups move in.
__global__ __launch_bounds__(768) void attn_bwd_fused_split_kernel(const AttnArgs a, int Lr, float* pd_ws, float* ds_ws) {
  set_wave_prio(a.prio);
  bwd_dq_split_body(a, Lr, pd_ws, ds_ws);
  // workgroup scope is enough (and an agent-scope fence is a write-back of the L2): the waves of a workgroup share a CU, the
  // vector L1 writes through, and no wave has read these scratch lines in this launch
  __syncthreads();
  bwd_dkv_split_body(a, Lr, pd_ws, ds_ws);
}

// Dropout keep flags of the attention probabilities for `nlayers` layers at once, in the layout the L <= 192 kernels store
// and read ([B*nh, L, Lr/32, 2] 16-bit words: bit 4g + t of word (bh, q, key block kb, half h) = key 32 kb + 8g + 4h + t kept).
// One thread per word (four Philox4x32-10 calls): a full-occupancy elementwise pass hides the ten dependent rounds that
// stall the attention kernels' three waves per SIMD (forward with dropout 17.4 vs 11.5 us per layer in the bf16 mode).
__global__ __launch_bounds__(256) void attn_keep_bits_kernel(unsigned short* __restrict__ out, size_t layer_stride,
                                                             size_t words, int L, int nblk, int Lp4, DropCfg d,
                                                             uint32_t site_step) {
  const size_t w = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (w >= words) return;
  d.site += site_step * blockIdx.y;
  const int h = (int)(w & 1);
  const size_t r = w >> 1;                    // (bh * L + q) * nblk + kb
  const int kb = (int)(r % nblk);
  const uint64_t grow = (uint64_t)(r / nblk) * Lp4 + (uint64_t)kb * 8;
  unsigned bits = 0;
#pragma unroll
  for (int g = 0; g < 4; ++g) bits |= drop_bits4(d, grow + 2 * g + h) << (4 * g);
  out[(size_t)blockIdx.y * layer_stride + w] = (unsigned short)bits;
}

constexpr int RES_MAX_LR = 256;      // 8 waves (2 per SIMD: 256 VGPRs each); 2*256*68*4 B = 139 KB LDS

inline size_t res_lds_bytes(int Lr) { return (size_t)(2 * Lr * LDT + 2 * Lr + 192) * sizeof(float); }

template <typename K>
int set_dyn_lds(K kernel, size_t bytes) {
  UCHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return 0;
}

// UNITER_ATTN_SPLIT=0 keeps the one-wave-per-block resident kernels (A/B measurements)
bool split_enabled() {
  static const bool on = [] { const char* e = getenv("UNITER_ATTN_SPLIT"); return !(e && e[0] == '0'); }();
  return on;
}

// UNITER_ATTN_BWD_FUSED=0: the two passes of the L <= 192 backward as two launches (A/B measurements)
bool bwd_fused() {
  static const bool on = [] { const char* e = getenv("UNITER_ATTN_BWD_FUSED"); return !(e && e[0] == '0'); }();
  return on;
}

int make_args(AttnArgs& a, int B, int L, int nh, float p_drop, uint64_t seed, uint32_t offset,
              uint32_t site) {
  UCHECK_ARG(B > 0 && L > 0 && nh > 0, "attention: bad dims B=%d L=%d nh=%d", B, L, nh);
  UCHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "attention: bad dropout p");
  a.B = B; a.L = L; a.nh = nh; a.H = nh * D; a.Lp4 = (L + 3) / 4;
  a.scale = 0.125f;   // 1/sqrt(64), model/layer.py:86
  a.drop = make_drop(p_drop, seed, offset, site);
  static const int prio = [] { const char* e = getenv("UNITER_ATTN_PRIO"); return e ? atoi(e) : 2; }();
  a.prio = prio;
  return 0;
}

int launch_bwd_split(const AttnArgs& a, int Lr, float* pd_ws, float* ds_ws, hipStream_t st) {
  const size_t lds = res_lds_bytes(Lr);
  if (bwd_fused()) {
    UCHECK_RC(set_dyn_lds(attn_bwd_fused_split_kernel, lds));
    hipLaunchKernelGGL(attn_bwd_fused_split_kernel, dim3(a.B * a.nh), dim3(Lr * 4), lds, st, a, Lr, pd_ws, ds_ws);
    UCHECK_LAUNCH();
    return 0;
  }
  UCHECK_RC(set_dyn_lds(attn_bwd_dq_split_kernel, lds));
  UCHECK_RC(set_dyn_lds(attn_bwd_dkv_split_kernel, lds));
  hipLaunchKernelGGL(attn_bwd_dq_split_kernel, dim3(a.B * a.nh), dim3(Lr * 4), lds, st, a, Lr, pd_ws, ds_ws);
  UCHECK_LAUNCH();
  hipLaunchKernelGGL(attn_bwd_dkv_split_kernel, dim3(a.B * a.nh), dim3(Lr * 4), lds, st, a, Lr, pd_ws, ds_ws);
  UCHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" int uniter_attn_fwd(const float* qkv, const float* attn_mask, float* ctx, float* lse, int B,
                               int L, int nh, float p_drop, uint64_t seed, uint32_t offset,
                               uint32_t site, void* stream) {
  UCHECK_ARG(qkv && attn_mask && ctx, "attn_fwd: null pointer");
  AttnArgs a = {};
  UCHECK_RC(make_args(a, B, L, nh, p_drop, seed, offset, site));
  a.qkv = qkv; a.mask = attn_mask; a.ctx = ctx; a.lse = lse;
  const int Lr = (L + 31) / 32 * 32;
  if (Lr <= SPLIT_MAX_LR && split_enabled()) {
    const size_t lds = res_lds_bytes(Lr);
    UCHECK_RC(set_dyn_lds(attn_fwd_split_kernel, lds));
    hipLaunchKernelGGL(attn_fwd_split_kernel, dim3(B * nh), dim3(Lr * 4), lds, (hipStream_t)stream, a, Lr);
  } else if (Lr <= RES_MAX_LR) {
    const size_t lds = res_lds_bytes(Lr);
    UCHECK_RC(set_dyn_lds(attn_fwd_res_kernel, lds));
    hipLaunchKernelGGL(attn_fwd_res_kernel, dim3(B * nh), dim3(Lr * 2), lds, (hipStream_t)stream, a, Lr);
  } else {
    dim3 grid((L + 32 * NW - 1) / (32 * NW), B * nh);
    hipLaunchKernelGGL(attn_fwd_kernel, grid, dim3(NW * 64), 0, (hipStream_t)stream, a);
  }
  UCHECK_LAUNCH();
  return 0;
}

// Packed batches (uniter_hip.h): the split kernels only -- Lmax <= 192.
extern "C" int uniter_attn_varlen_max_len(void) { return split_enabled() ? SPLIT_MAX_LR : 0; }

extern "C" int uniter_attn_fwd_varlen(const float* qkv, const int32_t* cu_seqlens, float* ctx, float* lse, int B,
                                      int Lmax, int nh, float p_drop, uint64_t seed, uint32_t offset,
                                      uint32_t site, void* stream) {
  UCHECK_ARG(qkv && cu_seqlens && ctx, "attn_fwd_varlen: null pointer");
  AttnArgs a = {};
  UCHECK_RC(make_args(a, B, Lmax, nh, p_drop, seed, offset, site));
  a.qkv = qkv; a.mask = nullptr; a.cu = cu_seqlens; a.ctx = ctx; a.lse = lse;
  const int Lr = (Lmax + 31) / 32 * 32;
  UCHECK_SHAPE(Lr <= uniter_attn_varlen_max_len(), "attn_fwd_varlen: Lmax %d > %d", Lmax, uniter_attn_varlen_max_len());
  const size_t lds = res_lds_bytes(Lr);
  UCHECK_RC(set_dyn_lds(attn_fwd_split_kernel, lds));
  hipLaunchKernelGGL(attn_fwd_split_kernel, dim3(B * nh), dim3(Lr * 4), lds, (hipStream_t)stream, a, Lr);
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_attn_bwd_varlen(const float* qkv, const int32_t* cu_seqlens, const float* ctx,
                                      const float* lse, const float* dctx, float* dqkv, float* delta, int B,
                                      int Lmax, int nh, float p_drop, uint64_t seed, uint32_t offset,
                                      uint32_t site, void* ws, size_t ws_bytes, void* stream) {
  UCHECK_ARG(qkv && cu_seqlens && ctx && lse && dctx && dqkv && delta && ws, "attn_bwd_varlen: null pointer");
  const int Lr = (Lmax + 31) / 32 * 32;
  UCHECK_SHAPE(Lr <= uniter_attn_varlen_max_len(), "attn_bwd_varlen: Lmax %d > %d", Lmax, uniter_attn_varlen_max_len());
  UCHECK_ARG(ws_bytes >= (size_t)2 * B * nh * Lr * Lr * sizeof(float), "attn_bwd_varlen: workspace too small");
  AttnArgs a = {};
  UCHECK_RC(make_args(a, B, Lmax, nh, p_drop, seed, offset, site));
  a.qkv = qkv; a.mask = nullptr; a.cu = cu_seqlens; a.ctx = const_cast<float*>(ctx); a.lse = const_cast<float*>(lse);
  a.dctx = dctx; a.dqkv = dqkv; a.delta = delta;
  float* pd_ws = (float*)ws;
  float* ds_ws = pd_ws + (size_t)B * nh * Lr * Lr;
  return launch_bwd_split(a, Lr, pd_ws, ds_ws, (hipStream_t)stream);
}

// General forms: mask XOR cu_seqlens, optional bf16 copies of the outputs (split kernels only: L <= 192).
extern "C" size_t uniter_attn_keep_bits_bytes(int B, int L, int nh) {
  const int Lr = (L + 31) / 32 * 32;
  return (size_t)B * nh * L * (Lr / 32) * 2 * sizeof(unsigned short);
}

extern "C" int uniter_attn_keep_bits_gen(void* keep_bits, size_t layer_stride_bytes, int nlayers, int B, int L, int nh,
                                         float p_drop, uint64_t seed, uint32_t offset, uint32_t site0, uint32_t site_step,
                                         void* stream) {
  UCHECK_ARG(keep_bits && nlayers >= 1 && nlayers <= 65535 && B > 0 && L > 0 && nh > 0 && layer_stride_bytes % 2 == 0,
             "attn_keep_bits_gen: bad argument");
  UCHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "attn_keep_bits_gen: bad dropout p");
  const int Lr = (L + 31) / 32 * 32, nblk = Lr / 32;
  UCHECK_SHAPE(Lr <= SPLIT_MAX_LR, "attn_keep_bits_gen: L %d > %d", L, SPLIT_MAX_LR);
  const size_t words = (size_t)B * nh * L * nblk * 2;
  UCHECK_ARG(nlayers == 1 || layer_stride_bytes >= words * 2, "attn_keep_bits_gen: layers overlap");
  const DropCfg d = make_drop(p_drop, seed, offset, site0);
  hipLaunchKernelGGL(attn_keep_bits_kernel, dim3((unsigned)((words + 255) / 256), nlayers), dim3(256), 0, (hipStream_t)stream,
                     (unsigned short*)keep_bits, layer_stride_bytes / 2, words, L, nblk, (L + 3) / 4, d, site_step);
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_attn_fwd_ex(const float* qkv, const float* attn_mask, const int32_t* cu_seqlens, float* ctx,
                                  void* ctx_bf16, float* lse, void* keep_bits, int B, int L, int nh, float p_drop,
                                  uint64_t seed, uint32_t offset, uint32_t site, void* stream) {
  return uniter_attn_fwd_pre(qkv, attn_mask, cu_seqlens, ctx, ctx_bf16, lse, keep_bits, 0, B, L, nh, p_drop, seed, offset, site,
                             stream);
}

static int attn_fwd_pre_run(const float* qkv, const float* attn_mask, const int32_t* cu_seqlens, float* ctx,
                            void* ctx_bf16, int pieces, float* lse, void* keep_bits, int keep_bits_ready, int B, int L, int nh,
                            float p_drop, uint64_t seed, uint32_t offset, uint32_t site, void* stream);

extern "C" int uniter_attn_fwd_pre(const float* qkv, const float* attn_mask, const int32_t* cu_seqlens, float* ctx,
                                   void* ctx_bf16, float* lse, void* keep_bits, int keep_bits_ready, int B, int L, int nh,
                                   float p_drop, uint64_t seed, uint32_t offset, uint32_t site, void* stream) {
  return attn_fwd_pre_run(qkv, attn_mask, cu_seqlens, ctx, ctx_bf16, 1, lse, keep_bits, keep_bits_ready, B, L, nh, p_drop, seed,
                          offset, site, stream);
}

extern "C" int uniter_attn_fwd_pre_x3(const float* qkv, const float* attn_mask, const int32_t* cu_seqlens, float* ctx,
                                      void* ctx_x3, float* lse, void* keep_bits, int keep_bits_ready, int B, int L, int nh,
                                      float p_drop, uint64_t seed, uint32_t offset, uint32_t site, void* stream) {
  UCHECK_ARG(ctx_x3 && ((uintptr_t)ctx_x3 & 7) == 0, "attn_fwd_pre_x3: ctx_x3 is NULL or misaligned");
  return attn_fwd_pre_run(qkv, attn_mask, cu_seqlens, ctx, ctx_x3, 3, lse, keep_bits, keep_bits_ready, B, L, nh, p_drop, seed,
                          offset, site, stream);
}

static int attn_fwd_pre_run(const float* qkv, const float* attn_mask, const int32_t* cu_seqlens, float* ctx,
                            void* ctx_bf16, int pieces, float* lse, void* keep_bits, int keep_bits_ready, int B, int L, int nh,
                            float p_drop, uint64_t seed, uint32_t offset, uint32_t site, void* stream) {
  UCHECK_ARG(qkv && ctx && ((attn_mask != nullptr) != (cu_seqlens != nullptr)), "attn_fwd_ex: need attn_mask or cu_seqlens (not both)");
  UCHECK_ARG(!keep_bits_ready || keep_bits, "attn_fwd_pre: keep_bits_ready without keep_bits");
  AttnArgs a = {};
  UCHECK_RC(make_args(a, B, L, nh, p_drop, seed, offset, site));
  a.qkv = qkv; a.mask = attn_mask; a.cu = cu_seqlens; a.ctx = ctx; a.ctx_b16 = (unsigned short*)ctx_bf16; a.b16_pieces = pieces; a.lse = lse;
  a.keep_bits = (unsigned short*)keep_bits; a.keep_ready = keep_bits_ready;
  const int Lr = (L + 31) / 32 * 32;
  UCHECK_SHAPE(Lr <= uniter_attn_varlen_max_len(), "attn_fwd_ex: L %d > %d", L, uniter_attn_varlen_max_len());
  const size_t lds = res_lds_bytes(Lr);
  UCHECK_RC(set_dyn_lds(attn_fwd_split_kernel, lds));
  hipLaunchKernelGGL(attn_fwd_split_kernel, dim3(B * nh), dim3(Lr * 4), lds, (hipStream_t)stream, a, Lr);
  UCHECK_LAUNCH();
  return 0;
}

static int attn_bwd_ex_run(const float* qkv, const float* attn_mask, const int32_t* cu_seqlens,
                           const float* ctx, const float* lse, const float* dctx, float* dqkv,
                           void* dqkv_bf16, int pieces, float* bias_part, const void* keep_bits, float* delta, int B, int L,
                           int nh, float p_drop, uint64_t seed, uint32_t offset, uint32_t site, void* ws,
                           size_t ws_bytes, void* stream);

extern "C" int uniter_attn_bwd_ex(const float* qkv, const float* attn_mask, const int32_t* cu_seqlens,
                                  const float* ctx, const float* lse, const float* dctx, float* dqkv,
                                  void* dqkv_bf16, float* bias_part, const void* keep_bits, float* delta, int B, int L,
                                  int nh, float p_drop, uint64_t seed, uint32_t offset, uint32_t site, void* ws,
                                  size_t ws_bytes, void* stream) {
  UCHECK_ARG(dqkv, "attn_bwd_ex: dqkv is NULL");
  return attn_bwd_ex_run(qkv, attn_mask, cu_seqlens, ctx, lse, dctx, dqkv, dqkv_bf16, 1, bias_part, keep_bits, delta, B, L, nh,
                         p_drop, seed, offset, site, ws, ws_bytes, stream);
}

extern "C" int uniter_attn_bwd_ex_x3(const float* qkv, const float* attn_mask, const int32_t* cu_seqlens,
                                     const float* ctx, const float* lse, const float* dctx, float* dqkv,
                                     void* dqkv_x3, float* bias_part, const void* keep_bits, float* delta, int B, int L,
                                     int nh, float p_drop, uint64_t seed, uint32_t offset, uint32_t site, void* ws,
                                     size_t ws_bytes, void* stream) {
  UCHECK_ARG(dqkv_x3 && ((uintptr_t)dqkv_x3 & 7) == 0, "attn_bwd_ex_x3: dqkv_x3 is NULL or misaligned");
  return attn_bwd_ex_run(qkv, attn_mask, cu_seqlens, ctx, lse, dctx, dqkv, dqkv_x3, 3, bias_part, keep_bits, delta, B, L, nh,
                         p_drop, seed, offset, site, ws, ws_bytes, stream);
}

static int attn_bwd_ex_run(const float* qkv, const float* attn_mask, const int32_t* cu_seqlens,
                           const float* ctx, const float* lse, const float* dctx, float* dqkv,
                           void* dqkv_bf16, int pieces, float* bias_part, const void* keep_bits, float* delta, int B, int L,
                           int nh, float p_drop, uint64_t seed, uint32_t offset, uint32_t site, void* ws,
                           size_t ws_bytes, void* stream) {
  UCHECK_ARG(qkv && ctx && lse && dctx && (dqkv || dqkv_bf16) && delta && ws && ((attn_mask != nullptr) != (cu_seqlens != nullptr)),
             "attn_bwd_ex: null pointer, or not exactly one of attn_mask / cu_seqlens");
  const int Lr = (L + 31) / 32 * 32;
  UCHECK_SHAPE(Lr <= uniter_attn_varlen_max_len(), "attn_bwd_ex: L %d > %d", L, uniter_attn_varlen_max_len());
  UCHECK_ARG(ws_bytes >= (size_t)2 * B * nh * Lr * Lr * sizeof(float), "attn_bwd_ex: workspace too small");
  AttnArgs a = {};
  UCHECK_RC(make_args(a, B, L, nh, p_drop, seed, offset, site));
  a.qkv = qkv; a.mask = attn_mask; a.cu = cu_seqlens; a.ctx = const_cast<float*>(ctx); a.lse = const_cast<float*>(lse);
  a.dctx = dctx; a.dqkv = dqkv; a.dqkv_b16 = (unsigned short*)dqkv_bf16; a.b16_pieces = pieces; a.bias_part = bias_part; a.keep_bits = (unsigned short*)keep_bits; a.delta = delta;
  float* pd_ws = (float*)ws;
  float* ds_ws = pd_ws + (size_t)B * nh * Lr * Lr;
  return launch_bwd_split(a, Lr, pd_ws, ds_ws, (hipStream_t)stream);
}

extern "C" size_t uniter_attn_bwd_ws_bytes(int B, int L, int nh) {
  const int Lr = (L + 31) / 32 * 32;
  if (B <= 0 || nh <= 0 || Lr > SPLIT_MAX_LR || !split_enabled()) return 0;
  return (size_t)2 * B * nh * Lr * Lr * sizeof(float);
}

extern "C" int uniter_attn_bwd(const float* qkv, const float* attn_mask, const float* ctx,
                               const float* lse, const float* dctx, float* dqkv, float* delta, int B,
                               int L, int nh, float p_drop, uint64_t seed, uint32_t offset,
                               uint32_t site, void* ws, size_t ws_bytes, void* stream) {
  UCHECK_ARG(qkv && attn_mask && ctx && lse && dctx && dqkv && delta, "attn_bwd: null pointer");
  UCHECK_ARG(ws_bytes >= uniter_attn_bwd_ws_bytes(B, L, nh) && (ws || ws_bytes == 0 || uniter_attn_bwd_ws_bytes(B, L, nh) == 0),
             "attn_bwd: workspace too small (uniter_attn_bwd_ws_bytes)");
  AttnArgs a = {};
  UCHECK_RC(make_args(a, B, L, nh, p_drop, seed, offset, site));
  a.qkv = qkv; a.mask = attn_mask; a.ctx = const_cast<float*>(ctx); a.lse = const_cast<float*>(lse);
  a.dctx = dctx; a.dqkv = dqkv; a.delta = delta;
  const int Lr = (L + 31) / 32 * 32;
  if (Lr <= SPLIT_MAX_LR && split_enabled()) {
    float* pd_ws = (float*)ws;
    float* ds_ws = pd_ws + (size_t)B * nh * Lr * Lr;
    return launch_bwd_split(a, Lr, pd_ws, ds_ws, (hipStream_t)stream);
  }
  if (Lr <= RES_MAX_LR) {
    const size_t lds = res_lds_bytes(Lr);
    UCHECK_RC(set_dyn_lds(attn_bwd_dq_res_kernel, lds));
    UCHECK_RC(set_dyn_lds(attn_bwd_dkv_res_kernel, lds));
    hipLaunchKernelGGL(attn_bwd_dq_res_kernel, dim3(B * nh), dim3(Lr * 2), lds, (hipStream_t)stream, a, Lr);
    UCHECK_LAUNCH();
    hipLaunchKernelGGL(attn_bwd_dkv_res_kernel, dim3(B * nh), dim3(Lr * 2), lds, (hipStream_t)stream, a, Lr);
    UCHECK_LAUNCH();
    return 0;
  }
  dim3 grid((L + 32 * NW - 1) / (32 * NW), B * nh);
  hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(NW * 64), 0, (hipStream_t)stream, a);
  UCHECK_LAUNCH();
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, dim3(NW * 64), 0, (hipStream_t)stream, a);
  UCHECK_LAUNCH();
  return 0;
}

#ifdef ATTN_STAMPS
extern "C" int uniter_dbg_attn_stamps(unsigned long long* out, size_t n) {
  UCHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_stamps), n * sizeof(unsigned long long)));
  return 0;
}
#endif
