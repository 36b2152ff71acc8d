// Embedding kernels: text (3 table lookups + LN + dropout), image-region
// epilogue (3 LayerNorms + 7-d position projection + type embedding + dropout),
// the gather that compacts [text|regions] per sample, and their backward.
//
// Replaces UniterTextEmbeddings.forward (model/model.py:232-245),
// UniterImageEmbeddings.forward after img_linear (:261-272), the type-embedding
// lookup of _compute_img_embeddings (:311-319) and cat + torch.gather
// (:330-333), plus their autograd.
//
// HBM/latency-bound row kernels: one wave per row of H floats, 16-byte accesses,
// all three LayerNorms of an image row stay in registers.  Backward recomputes
// the cheap forward pieces instead of saving them; embedding-table gradients are
// scattered with fp32 atomics issued as 256-B contiguous wave transactions (the
// fast shape, MI355X_MICROARCH.md "Global float atomics"); affine / type grads use
// deterministic two-stage column reductions.
#include "common.h"
#include "philox.h"
#include "rowops.h"

namespace {

__device__ __forceinline__ int clampi(long long v, int hi) { return v < 0 ? 0 : (v >= hi ? hi - 1 : (int)v); }

template <int NV>
__device__ __forceinline__ void row_add(f32x4 (&v)[NV], const f32x4 (&w)[NV]) {
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] += w[k];
}

// p[c] = sum_k pos7[k] * Wp[c*7+k] + bp[c] for the lane's columns
template <int NV>
__device__ __forceinline__ void pos_linear(f32x4 (&p)[NV], const float* __restrict__ pos7,
                                           const float* __restrict__ Wp, const float* __restrict__ bp,
                                           int H4, int lane) {
  float x[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) x[k] = pos7[k];
#pragma unroll
  for (int kk = 0; kk < NV; ++kk) {
    const int c4 = lane + 64 * kk;
    if (c4 < H4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = c4 * 4 + e;
        float s = bp[c];
#pragma unroll
        for (int k = 0; k < 7; ++k) s += x[k] * Wp[c * 7 + k];
        p[kk][e] = s;
      }
    } else {
      p[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
}

struct TxtArgs {
  const int64_t* ids; const int64_t* pos_ids; const int64_t* type_ids;
  const float* word; const float* pos; const float* type;
  const float* gamma; const float* beta;
  float* cat; const float* dcat;
  float* dword; float* dpos; float* dtype; float* part;
  int B, T, S, H, vocab, max_pos, type_vocab, pos_bcast;
  DropCfg drop;
};

template <int NV>
__device__ __forceinline__ void txt_row_sum(const TxtArgs& a, int row, int t, f32x4 (&e)[NV], int& id, int& pid,
                                            int& tt, int H4, int lane) {
  id = clampi(a.ids[row], a.vocab);
  pid = clampi(a.pos_ids[a.pos_bcast ? t : row], a.max_pos);
  tt = a.type_ids ? clampi(a.type_ids[row], a.type_vocab) : 0;
  f32x4 w[NV];
  row_load<NV>(e, a.word + (size_t)id * a.H, H4, lane);
  row_load<NV>(w, a.pos + (size_t)pid * a.H, H4, lane);
  row_add<NV>(e, w);
  row_load<NV>(w, a.type + (size_t)tt * a.H, H4, lane);
  row_add<NV>(e, w);
}

template <int NV>
__global__ __launch_bounds__(256) void txt_embed_fwd_kernel(const TxtArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.B * a.T) return;
  const int b = row / a.T, t = row - b * a.T, H4 = a.H >> 2;
  f32x4 e[NV];
  int id, pid, tt;
  txt_row_sum<NV>(a, row, t, e, id, pid, tt, H4, lane);
  float mean, rstd;
  row_stats<NV>(e, a.H, H4, lane, mean, rstd);
  row_affine<NV>(e, a.gamma, a.beta, mean, rstd, H4, lane);
  if (a.drop.active) row_dropout<NV>(e, a.drop, (uint64_t)row * H4, H4, lane);
  row_store<NV>(e, a.cat + ((size_t)b * a.S + t) * a.H, H4, lane);
}

// scatter-add one row (held as float4 chunks) as contiguous 256-B atomic transactions via LDS
template <int NV>
__device__ __forceinline__ void row_atomic_add(const f32x4 (&v)[NV], float* buf, float* __restrict__ dst,
                                               int H, int H4, int lane) {
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = lane + 64 * k;
    if (c < H4) *reinterpret_cast<f32x4*>(buf + c * 4) = v[k];
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
  for (int c = lane; c < H; c += 64) atomicAdd(dst + c, buf[c]);
  __builtin_amdgcn_wave_barrier();
}

template <int NV>
__global__ __launch_bounds__(256) void txt_embed_bwd_kernel(const TxtArgs a) {
  __shared__ __attribute__((aligned(16))) float red[4 * NV * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, H4 = a.H >> 2;
  float* abuf = red + wave * NV * 256;   // per-wave staging for atomics (reused for the reduce)
  f32x4 g[NV], dg[NV], db[NV], dt0[NV];
  row_load<NV>(g, a.gamma, H4, lane);
#pragma unroll
  for (int k = 0; k < NV; ++k) { dg[k] = f32x4{0, 0, 0, 0}; db[k] = dg[k]; dt0[k] = dg[k]; }
  const int rows = a.B * a.T;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const int b = row / a.T, t = row - b * a.T;
    f32x4 d[NV], e[NV];
    row_load<NV>(d, a.dcat + ((size_t)b * a.S + t) * a.H, H4, lane);
    if (a.drop.active) row_dropout<NV>(d, a.drop, (uint64_t)row * H4, H4, lane);
    int id, pid, tt;
    txt_row_sum<NV>(a, row, t, e, id, pid, tt, H4, lane);
    float mean, rstd;
    row_stats<NV>(e, a.H, H4, lane, mean, rstd);
    row_ln_bwd<NV>(d, e, g, mean, rstd, dg, db, a.H, H4, lane);     // d <- d(sum of the three lookups)
    if (id != 0) row_atomic_add<NV>(d, abuf, a.dword + (size_t)id * a.H, a.H, H4, lane);  // padding_idx=0
    row_atomic_add<NV>(d, abuf, a.dpos + (size_t)pid * a.H, a.H, H4, lane);
    if (a.type_ids) row_atomic_add<NV>(d, abuf, a.dtype + (size_t)tt * a.H, a.H, H4, lane);
    else row_add<NV>(dt0, d);
  }
  float* part = a.part + (size_t)blockIdx.x * 3 * a.H;
  block_col_reduce_store<NV>(dg, red, part, H4, lane, wave);
  block_col_reduce_store<NV>(db, red, part + a.H, H4, lane, wave);
  block_col_reduce_store<NV>(dt0, red, part + 2 * a.H, H4, lane, wave);
}

struct ImgArgs {
  const float* imgfc; const float* pos7; const int64_t* type_ids;
  const float* Wp; const float* bp; const float* type;
  const float* g_i; const float* b_i; const float* g_p; const float* b_p; const float* g_f; const float* b_f;
  float* cat; float* stats; const float* dcat;
  float* d_imgfc; float* d_posfc; float* dtype; float* part;
  int B, R, T0, S, H, type_vocab;
  DropCfg drop;
};

template <int NV>
__global__ __launch_bounds__(256) void img_embed_fwd_kernel(const ImgArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.B * a.R) return;
  const int b = row / a.R, r = row - b * a.R, H4 = a.H >> 2;
  f32x4 x[NV], p[NV];
  float st[6];
  row_load<NV>(x, a.imgfc + (size_t)row * a.H, H4, lane);
  row_stats<NV>(x, a.H, H4, lane, st[0], st[1]);
  row_affine<NV>(x, a.g_i, a.b_i, st[0], st[1], H4, lane);
  pos_linear<NV>(p, a.pos7 + (size_t)row * 7, a.Wp, a.bp, H4, lane);
  row_stats<NV>(p, a.H, H4, lane, st[2], st[3]);
  row_affine<NV>(p, a.g_p, a.b_p, st[2], st[3], H4, lane);
  row_add<NV>(x, p);
  const int tid = a.type_ids ? clampi(a.type_ids[row], a.type_vocab) : 1;
  row_load<NV>(p, a.type + (size_t)tid * a.H, H4, lane);
  row_add<NV>(x, p);
  // lanes beyond H4 hold garbage-free zeros only if every addend was zero there: enforce
#pragma unroll
  for (int k = 0; k < NV; ++k) if (lane + 64 * k >= H4) x[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  row_stats<NV>(x, a.H, H4, lane, st[4], st[5]);
  row_affine<NV>(x, a.g_f, a.b_f, st[4], st[5], H4, lane);
  if (a.drop.active) row_dropout<NV>(x, a.drop, (uint64_t)row * H4, H4, lane);
  row_store<NV>(x, a.cat + ((size_t)b * a.S + a.T0 + r) * a.H, H4, lane);
  if (a.stats && lane < 6) a.stats[(size_t)row * 6 + lane] = st[lane];
}

template <int NV>
__global__ __launch_bounds__(256) void img_embed_bwd_kernel(const ImgArgs a) {
  __shared__ __attribute__((aligned(16))) float red[4 * NV * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, H4 = a.H >> 2;
  float* abuf = red + wave * NV * 256;
  f32x4 acc[7][NV];      // dg_f, db_f, dg_i, db_i, dg_p, db_p, dtype[1]
#pragma unroll
  for (int s = 0; s < 7; ++s)
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[s][k] = f32x4{0, 0, 0, 0};
  f32x4 gf[NV], gi[NV], gp[NV];
  row_load<NV>(gf, a.g_f, H4, lane);
  row_load<NV>(gi, a.g_i, H4, lane);
  row_load<NV>(gp, a.g_p, H4, lane);
  const int rows = a.B * a.R;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const int b = row / a.R, r = row - b * a.R;
    const float* st = a.stats + (size_t)row * 6;
    f32x4 d[NV], x[NV], p[NV], f[NV];
    row_load<NV>(d, a.dcat + ((size_t)b * a.S + a.T0 + r) * a.H, H4, lane);
    if (a.drop.active) row_dropout<NV>(d, a.drop, (uint64_t)row * H4, H4, lane);
    // recompute f = LN_i(imgfc) + LN_p(pos proj) + type
    row_load<NV>(x, a.imgfc + (size_t)row * a.H, H4, lane);
    pos_linear<NV>(p, a.pos7 + (size_t)row * 7, a.Wp, a.bp, H4, lane);
    const int tid = a.type_ids ? clampi(a.type_ids[row], a.type_vocab) : 1;
    row_load<NV>(f, a.type + (size_t)tid * a.H, H4, lane);
    {
      f32x4 t[NV];
#pragma unroll
      for (int k = 0; k < NV; ++k) t[k] = x[k];
      row_affine<NV>(t, a.g_i, a.b_i, st[0], st[1], H4, lane);
      row_add<NV>(f, t);
#pragma unroll
      for (int k = 0; k < NV; ++k) t[k] = p[k];
      row_affine<NV>(t, a.g_p, a.b_p, st[2], st[3], H4, lane);
      row_add<NV>(f, t);
    }
    row_ln_bwd<NV>(d, f, gf, st[4], st[5], acc[0], acc[1], a.H, H4, lane);      // d <- df
    if (a.type_ids) row_atomic_add<NV>(d, abuf, a.dtype + (size_t)tid * a.H, a.H, H4, lane);
    else row_add<NV>(acc[6], d);
    f32x4 d2[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) d2[k] = d[k];
    row_ln_bwd<NV>(d, x, gi, st[0], st[1], acc[2], acc[3], a.H, H4, lane);      // d <- d_imgfc
    row_store<NV>(d, a.d_imgfc + (size_t)row * a.H, H4, lane);
    row_ln_bwd<NV>(d2, p, gp, st[2], st[3], acc[4], acc[5], a.H, H4, lane);     // d2 <- d_posfc
    row_store<NV>(d2, a.d_posfc + (size_t)row * a.H, H4, lane);
  }
  float* part = a.part + (size_t)blockIdx.x * 7 * a.H;
#pragma unroll
  for (int s = 0; s < 7; ++s) block_col_reduce_store<NV>(acc[s], red, part + (size_t)s * a.H, H4, lane, wave);
}

// dWp[c][k] += sum_rows d_posfc[row][c] * pos7[row][k];  dbp[c] += sum_rows d_posfc[row][c]
__global__ __launch_bounds__(256) void pos_linear_wgrad_kernel(const float* __restrict__ d_posfc,
                                                               const float* __restrict__ pos7,
                                                               float* __restrict__ dWp, float* __restrict__ dbp,
                                                               int rows, int H) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int r0 = blockIdx.y * 32, r1 = min(rows, r0 + 32);
  if (c >= H) return;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int r = r0; r < r1; ++r) {
    const float dv = d_posfc[(size_t)r * H + c];
#pragma unroll
    for (int k = 0; k < 7; ++k) s[k] += dv * pos7[(size_t)r * 7 + k];
    s[7] += dv;
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) atomicAdd(dWp + (size_t)c * 7 + k, s[k]);
  atomicAdd(dbp + c, s[7]);
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ cat,
                                                          const int64_t* __restrict__ gi, float* __restrict__ out,
                                                          int B, int S, int Lout, int H) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B * Lout) return;
  const int b = row / Lout, j = row - b * Lout;
  const int s = gi ? clampi(gi[row], S) : j;
  const f32x4* src = reinterpret_cast<const f32x4*>(cat + ((size_t)b * S + s) * H);
  f32x4* dst = reinterpret_cast<f32x4*>(out + (size_t)row * H);
  for (int c = lane; c < (H >> 2); c += 64) dst[c] = src[c];
}

// the same gather with the operand copies of layer 0's first product written in the same pass (round 6: the gather and the split were
// two launches of the chain that keeps the first encoder product waiting behind the optimizer -- 11 + 8 us): MODE 1 = the three bf16
// pieces of the fp32x3 mode ([row][3][H], what uniter_split3 writes), MODE 2 = the bf16 copy of the bf16 mode (uniter_cast_bf16)
template <int NV, int MODE>
__global__ __launch_bounds__(256) void gather_rows_ex_kernel(const float* __restrict__ cat, const int64_t* __restrict__ gi,
                                                             float* __restrict__ out, unsigned short* __restrict__ outb, int B, int S,
                                                             int Lout, int H) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B * Lout) return;
  const int b = row / Lout, j = row - b * Lout;
  const int s = gi ? clampi(gi[row], S) : j;
  const int H4 = H >> 2;
  f32x4 v[NV];
  row_load<NV>(v, cat + ((size_t)b * S + s) * H, H4, lane);
  row_store<NV>(v, out + (size_t)row * H, H4, lane);
  if constexpr (MODE == 1) row_store_x3<NV>(v, outb + (size_t)row * 3 * H, H, H4, lane);
  else row_store_bf16<NV>(v, outb + (size_t)row * H, H4, lane);
}

// source-indexed: dcat[b,s] = sum over j with gi[b,j]==s of dout[b,j]  (no atomics, deterministic)
__global__ __launch_bounds__(256) void gather_rows_bwd_kernel(const float* __restrict__ dout,
                                                              const int64_t* __restrict__ gi,
                                                              float* __restrict__ dcat, int B, int S, int Lout, int H) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B * S) return;
  const int b = row / S, s = row - b * S;
  f32x4* dst = reinterpret_cast<f32x4*>(dcat + (size_t)row * H);
  const int H4 = H >> 2;
  if (!gi) {
    const f32x4* src = reinterpret_cast<const f32x4*>(dout + ((size_t)b * Lout + s) * H);
    for (int c = lane; c < H4; c += 64) dst[c] = s < Lout ? src[c] : f32x4{0.f, 0.f, 0.f, 0.f};
    return;
  }
  for (int c0 = 0; c0 < H4; c0 += 64) {
    const int c = c0 + lane;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int j0 = 0; j0 < Lout; j0 += 64) {
      const int j = j0 + lane;
      const bool hit = j < Lout && gi[(size_t)b * Lout + j] == (int64_t)s;
      unsigned long long m = __ballot(hit);
      while (m) {
        const int jj = j0 + __builtin_ctzll(m);
        m &= m - 1;
        if (c < H4) acc += reinterpret_cast<const f32x4*>(dout + ((size_t)b * Lout + jj) * H)[c];
      }
    }
    if (c < H4) dst[c] = acc;
  }
}

__global__ void img_mask_add_kernel(const float* __restrict__ feat, const int64_t* __restrict__ masks,
                                    const float* __restrict__ mask_emb, float* __restrict__ out, int rows, int D4) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)rows * D4) return;
  const int row = (int)(idx / D4), c = (int)(idx - (size_t)row * D4);
  f32x4 v = reinterpret_cast<const f32x4*>(feat)[idx];
  const int64_t m = masks[row];
  if (m != 0) v += reinterpret_cast<const f32x4*>(mask_emb + (size_t)(m > 1 ? 1 : m) * D4 * 4)[c];   // row 0 == 0
  reinterpret_cast<f32x4*>(out)[idx] = v;
}

// out[c] += sum over rows with masks[row] != 0 of x[row][c]   (gradient of mask_embedding row 1)
__global__ void masked_rowsum_kernel(const float* __restrict__ x, const int64_t* __restrict__ masks,
                                     float* __restrict__ out, int rows, int D) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= D) return;
  float s = 0.f;
  for (int r = 0; r < rows; ++r)
    if (masks[r] != 0) s += x[(size_t)r * D + c];
  out[c] += s;
}

inline int bwd_blocks(int rows) {
  int b = (rows + 3) / 4;
  return b < 256 ? (b < 1 ? 1 : b) : 256;
}

}  // namespace

int launch_masked_rowsum(const float* x, const int64_t* masks, float* out, int rows, int D, hipStream_t st) {
  hipLaunchKernelGGL(masked_rowsum_kernel, dim3((D + 255) / 256), dim3(256), 0, st, x, masks, out, rows, D);
  UCHECK_LAUNCH();
  return 0;
}

#define NV_DISPATCH(KERNEL, GRID, ARG)                                                         \
  switch ((H / 4 + 63) / 64) {                                                                 \
    case 1: hipLaunchKernelGGL((KERNEL<1>), GRID, dim3(256), 0, st, ARG); break;               \
    case 2: hipLaunchKernelGGL((KERNEL<2>), GRID, dim3(256), 0, st, ARG); break;               \
    case 3: hipLaunchKernelGGL((KERNEL<3>), GRID, dim3(256), 0, st, ARG); break;               \
    case 4: hipLaunchKernelGGL((KERNEL<4>), GRID, dim3(256), 0, st, ARG); break;               \
    default: uniter_set_error("embed: hidden size %d unsupported (max 1024)", H);              \
             return UNITER_E_SHAPE;                                                            \
  }

extern "C" size_t uniter_embed_bwd_ws_bytes(int rows, int H) {
  return (size_t)bwd_blocks(rows) * 7 * H * sizeof(float);
}

extern "C" int uniter_txt_embed_fwd(const int64_t* input_ids, const int64_t* position_ids,
                                    const int64_t* type_ids, const float* word, const float* pos,
                                    const float* type, const float* gamma, const float* beta, float* cat,
                                    int B, int T, int S, int H, int vocab, int max_pos, int type_vocab,
                                    int pos_bcast, float p_drop, uint64_t seed, uint32_t offset,
                                    void* stream) {
  UCHECK_ARG(input_ids && position_ids && word && pos && type && gamma && beta && cat, "txt_embed_fwd: null pointer");
  UCHECK_SHAPE(H % 4 == 0 && T <= S, "txt_embed_fwd: bad shape");
  if (B * T <= 0) return 0;
  TxtArgs a = {};
  a.ids = input_ids; a.pos_ids = position_ids; a.type_ids = type_ids; a.word = word; a.pos = pos;
  a.type = type; a.gamma = gamma; a.beta = beta; a.cat = cat; a.B = B; a.T = T; a.S = S; a.H = H;
  a.vocab = vocab; a.max_pos = max_pos; a.type_vocab = type_vocab; a.pos_bcast = pos_bcast;
  a.drop = make_drop(p_drop, seed, offset, SITE_TXT_EMB);
  hipStream_t st = (hipStream_t)stream;
  NV_DISPATCH(txt_embed_fwd_kernel, dim3((B * T + 3) / 4), a);
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_txt_embed_bwd(const float* dcat, const int64_t* input_ids, const int64_t* position_ids,
                                    const int64_t* type_ids, const float* word, const float* pos,
                                    const float* type, const float* gamma, float* dword, float* dpos,
                                    float* dtype, float* dgamma, float* dbeta, int B, int T, int S, int H,
                                    int vocab, int max_pos, int type_vocab, int pos_bcast, float p_drop,
                                    uint64_t seed, uint32_t offset, void* ws, size_t ws_bytes, void* stream) {
  UCHECK_ARG(dcat && input_ids && position_ids && word && pos && type && gamma && dword && dpos && dtype &&
             dgamma && dbeta && ws, "txt_embed_bwd: null pointer");
  UCHECK_SHAPE(H % 4 == 0 && T <= S, "txt_embed_bwd: bad shape");
  UCHECK_ARG(ws_bytes >= uniter_embed_bwd_ws_bytes(B * T, H), "txt_embed_bwd: workspace too small");
  if (B * T <= 0) return 0;
  TxtArgs a = {};
  a.ids = input_ids; a.pos_ids = position_ids; a.type_ids = type_ids; a.word = word; a.pos = pos;
  a.type = type; a.gamma = gamma; a.dcat = dcat; a.dword = dword; a.dpos = dpos; a.dtype = dtype;
  a.part = (float*)ws; a.B = B; a.T = T; a.S = S; a.H = H; a.vocab = vocab; a.max_pos = max_pos;
  a.type_vocab = type_vocab; a.pos_bcast = pos_bcast;
  a.drop = make_drop(p_drop, seed, offset, SITE_TXT_EMB);
  hipStream_t st = (hipStream_t)stream;
  const int nblk = bwd_blocks(B * T);
  NV_DISPATCH(txt_embed_bwd_kernel, dim3(nblk), a);
  UCHECK_LAUNCH();
  float* outs[3] = {dgamma, dbeta, type_ids ? nullptr : dtype};
  return finalize_partials_multi(a.part, nblk, (size_t)3 * H, outs, 3, H, st);
}

extern "C" int uniter_img_embed_fwd(const float* imgfc, const float* pos7, const int64_t* img_type_ids,
                                    const float* Wp, const float* bp, const float* type, const float* g_i,
                                    const float* b_i, const float* g_p, const float* b_p, const float* g_f,
                                    const float* b_f, float* cat, float* stats, int B, int R, int T0, int S,
                                    int H, int type_vocab, float p_drop, uint64_t seed, uint32_t offset,
                                    void* stream) {
  UCHECK_ARG(imgfc && pos7 && Wp && bp && type && g_i && b_i && g_p && b_p && g_f && b_f && cat,
             "img_embed_fwd: null pointer");
  UCHECK_SHAPE(H % 4 == 0 && T0 + R <= S && type_vocab >= 2, "img_embed_fwd: bad shape");
  if (B * R <= 0) return 0;
  ImgArgs a = {};
  a.imgfc = imgfc; a.pos7 = pos7; a.type_ids = img_type_ids; a.Wp = Wp; a.bp = bp; a.type = type;
  a.g_i = g_i; a.b_i = b_i; a.g_p = g_p; a.b_p = b_p; a.g_f = g_f; a.b_f = b_f; a.cat = cat;
  a.stats = stats; a.B = B; a.R = R; a.T0 = T0; a.S = S; a.H = H; a.type_vocab = type_vocab;
  a.drop = make_drop(p_drop, seed, offset, SITE_IMG_EMB);
  hipStream_t st = (hipStream_t)stream;
  NV_DISPATCH(img_embed_fwd_kernel, dim3((B * R + 3) / 4), a);
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_img_embed_bwd(const float* dcat, const float* imgfc, const float* pos7,
                                    const int64_t* img_type_ids, const float* Wp, const float* bp,
                                    const float* type, const float* g_i, const float* b_i, const float* g_p,
                                    const float* b_p, const float* g_f, const float* stats, float* d_imgfc,
                                    float* d_posfc, float* dWp, float* dbp, float* dtype, float* dg_i,
                                    float* db_i, float* dg_p, float* db_p, float* dg_f, float* db_f, int B,
                                    int R, int T0, int S, int H, int type_vocab, float p_drop, uint64_t seed,
                                    uint32_t offset, void* ws, size_t ws_bytes, void* stream) {
  UCHECK_ARG(dcat && imgfc && pos7 && Wp && bp && type && g_i && b_i && g_p && b_p && g_f && stats &&
             d_imgfc && d_posfc && dWp && dbp && dtype && dg_i && db_i && dg_p && db_p && dg_f && db_f && ws,
             "img_embed_bwd: null pointer");
  UCHECK_SHAPE(H % 4 == 0 && T0 + R <= S && type_vocab >= 2, "img_embed_bwd: bad shape");
  UCHECK_ARG(ws_bytes >= uniter_embed_bwd_ws_bytes(B * R, H), "img_embed_bwd: workspace too small");
  if (B * R <= 0) return 0;
  ImgArgs a = {};
  a.imgfc = imgfc; a.pos7 = pos7; a.type_ids = img_type_ids; a.Wp = Wp; a.bp = bp; a.type = type;
  a.g_i = g_i; a.b_i = b_i; a.g_p = g_p; a.b_p = b_p; a.g_f = g_f; a.stats = const_cast<float*>(stats);
  a.dcat = dcat; a.d_imgfc = d_imgfc; a.d_posfc = d_posfc; a.dtype = dtype; a.part = (float*)ws;
  a.B = B; a.R = R; a.T0 = T0; a.S = S; a.H = H; a.type_vocab = type_vocab;
  a.drop = make_drop(p_drop, seed, offset, SITE_IMG_EMB);
  hipStream_t st = (hipStream_t)stream;
  const int nblk = bwd_blocks(B * R);
  NV_DISPATCH(img_embed_bwd_kernel, dim3(nblk), a);
  UCHECK_LAUNCH();
  float* outs[7] = {dg_f, db_f, dg_i, db_i, dg_p, db_p, img_type_ids ? nullptr : dtype + H};
  UCHECK_RC(finalize_partials_multi(a.part, nblk, (size_t)7 * H, outs, 7, H, st));
  hipLaunchKernelGGL(pos_linear_wgrad_kernel, dim3((H + 255) / 256, (B * R + 31) / 32), dim3(256), 0, st,
                     d_posfc, pos7, dWp, dbp, B * R, H);
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_gather_rows(const float* cat, const int64_t* gather_index, float* out, int B, int S,
                                  int Lout, int H, void* stream) {
  UCHECK_ARG(cat && out, "gather_rows: null pointer");
  UCHECK_SHAPE(H % 4 == 0 && (gather_index || Lout <= S), "gather_rows: bad shape");
  if (B * Lout <= 0) return 0;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((B * Lout + 3) / 4), dim3(256), 0, (hipStream_t)stream, cat,
                     gather_index, out, B, S, Lout, H);
  UCHECK_LAUNCH();
  return 0;
}

// uniter_gather_rows with the joint rows' operand copy in the same launch: mode 1 = three bf16 pieces per row ([row][3][H], as
// uniter_split3 with row stride 3 H and piece stride H), mode 2 = one bf16 copy ([row][H], as uniter_cast_bf16).  H up to 1024;
// UNITER_E_SHAPE beyond (the caller then runs the two launches).
extern "C" int uniter_gather_rows_ex(const float* cat, const int64_t* gather_index, float* out, void* out_b16, int mode, int B, int S,
                                     int Lout, int H, void* stream) {
  UCHECK_ARG(cat && out && out_b16 && (mode == 1 || mode == 2), "gather_rows_ex: bad argument");
  UCHECK_SHAPE(H % 4 == 0 && H <= 1024 && (gather_index || Lout <= S), "gather_rows_ex: bad shape (H %% 4, H <= 1024)");
  if (B * Lout <= 0) return 0;
  const dim3 grid((B * Lout + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  unsigned short* ob = (unsigned short*)out_b16;
  const int nv = (H / 4 + 63) / 64;
#define UNITER_GRX(NV_)                                                                                                       \
  if (mode == 1) hipLaunchKernelGGL((gather_rows_ex_kernel<NV_, 1>), grid, block, 0, st, cat, gather_index, out, ob, B, S, Lout, H); \
  else hipLaunchKernelGGL((gather_rows_ex_kernel<NV_, 2>), grid, block, 0, st, cat, gather_index, out, ob, B, S, Lout, H);
  if (nv == 1) { UNITER_GRX(1) } else if (nv == 2) { UNITER_GRX(2) } else if (nv == 3) { UNITER_GRX(3) } else { UNITER_GRX(4) }
#undef UNITER_GRX
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_gather_rows_bwd(const float* dout, const int64_t* gather_index, float* dcat, int B,
                                      int S, int Lout, int H, void* stream) {
  UCHECK_ARG(dout && dcat, "gather_rows_bwd: null pointer");
  UCHECK_SHAPE(H % 4 == 0 && (gather_index || Lout <= S), "gather_rows_bwd: bad shape");
  if (B * S <= 0) return 0;
  hipLaunchKernelGGL(gather_rows_bwd_kernel, dim3((B * S + 3) / 4), dim3(256), 0, (hipStream_t)stream, dout,
                     gather_index, dcat, B, S, Lout, H);
  UCHECK_LAUNCH();
  return 0;
}

// out[m][:] = bias for m < M: the starting value of a product that is then ACCUMULATED on top (C += x W^T by the stream-K
// form of the GEMM, which has no bias epilogue)
__global__ __launch_bounds__(256) void bias_rows_kernel(const float* __restrict__ bias, float* __restrict__ out, int M, int N4) {
  const size_t n = (size_t)M * N4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    reinterpret_cast<f32x4*>(out)[i] = reinterpret_cast<const f32x4*>(bias)[i % N4];
}

extern "C" int uniter_bias_rows(const float* bias, float* out, int M, int N, void* stream) {
  UCHECK_ARG(bias && out && M > 0 && N > 0, "bias_rows: bad argument");
  UCHECK_SHAPE(N % 4 == 0, "bias_rows: N must be a multiple of 4");
  const size_t n = (size_t)M * (N / 4);
  const int nb = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(bias_rows_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, bias, out, M, N / 4);
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_img_mask_add(const float* feat, const int64_t* img_masks, const float* mask_emb,
                                   float* feat_out, int rows, int D, void* stream) {
  UCHECK_ARG(feat && img_masks && mask_emb && feat_out, "img_mask_add: null pointer");
  UCHECK_SHAPE(D % 4 == 0, "img_mask_add: D must be a multiple of 4");
  const size_t n = (size_t)rows * (D / 4);
  if (n == 0) return 0;
  hipLaunchKernelGGL(img_mask_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, feat, img_masks, mask_emb, feat_out, rows, D / 4);
  UCHECK_LAUNCH();
  return 0;
}
