// Fused optimizer step over flat fp32 buffers.
//
// Replaces, in one HBM pass, TrainerTemplate.average_gradients
// (train_template.py:89-92), torch.nn.utils.clip_grad_norm_ (:104), the
// per-tensor torch.optim.Adam / AdamW step configured by get_optimizer
// (utils/optim_utils.py:9-46) and optimizer.zero_grad (:107): 212 tensors x ~5
// elementwise launches in the reference.  Algorithmic traffic: read p,g,m,v +
// write p,m,v (+ zeroed g) = 32 B per parameter.
//
// The global gradient norm is reduced on the device into a double and consumed
// by the step kernel from device memory: no host synchronisation.
// Per-64-element chunk flags carry the parameter-group split:
//   0 = parameter received no gradient this step (torch skips grad=None params),
//   1 = no weight decay ('bias' / 'LayerNorm.*' names), 2 = weight decay;
//   + 4 = leave the gradient as it is (zero_grads notwithstanding): the next backward pass OVERWRITES this chunk
//         (the encoder's weight gradients, uniter_model_set_wgrad_overwrite) -- 28 instead of 32 bytes per parameter.
#include "common.h"

namespace {

constexpr int CHUNK = 64;

__device__ __forceinline__ f32x4 widen4(const unsigned short* __restrict__ g16, size_t i) {
  typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
  const u32x2_t w = reinterpret_cast<const u32x2_t*>(g16)[i];
  return f32x4{__builtin_bit_cast(float, w[0] << 16), __builtin_bit_cast(float, w[0] & 0xffff0000u),
               __builtin_bit_cast(float, w[1] << 16), __builtin_bit_cast(float, w[1] & 0xffff0000u)};
}

// g16 (optional): the gradients as bf16 (the data-parallel exchange's reduced payload) instead of the fp32 buffer
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g,
                                                    const uint8_t* __restrict__ flags, size_t n4,
                                                    double* __restrict__ part,
                                                    const unsigned short* __restrict__ g16) {
  __shared__ double red[4];
  double acc = 0.0;
  // two 16-byte pieces per thread and iteration, both loads issued first (the pass is a pure HBM stream: 440 MB)
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += 2 * stride) {
    const size_t j = i + stride;
    const bool f0 = !flags || flags[(i * 4) / CHUNK] != 0, f1 = j < n4 && (!flags || flags[(j * 4) / CHUNK] != 0);
    f32x4 v = {0.f, 0.f, 0.f, 0.f}, w = v;
    if (f0) v = g16 ? widen4(g16, i) : reinterpret_cast<const f32x4*>(g)[i];
    if (f1) w = g16 ? widen4(g16, j) : reinterpret_cast<const f32x4*>(g)[j];
    acc += (double)(v[0] * v[0] + v[1] * v[1]) + (double)(v[2] * v[2] + v[3] * v[3]);
    acc += (double)(w[0] * w[0] + w[1] * w[1]) + (double)(w[2] * w[2] + w[3] * w[3]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// (round 5: the fp32x3 backward leaves ~1000 partial sums per layer -- 1024 threads, four independent loads in flight each; the
// order of the additions depends on n alone: bit-reproducible)
__global__ __launch_bounds__(1024) void sumsq_final_kernel(const double* __restrict__ part, int n,
                                                           double* __restrict__ out) {
  __shared__ double red[16];
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int i = threadIdx.x;
  for (; i + 3 * 1024 < n; i += 4 * 1024) { a0 += part[i]; a1 += part[i + 1024]; a2 += part[i + 2048]; a3 += part[i + 3072]; }
  for (; i < n; i += 1024) a0 += part[i];
  double acc = (a0 + a1) + (a2 + a3);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w];
    out[0] = t;
  }
}

struct AdamArgs {
  float* p; float* g; float* m; float* v;
  const uint8_t* flags;
  size_t n4;
  const double* sumsq;
  float gscale, max_norm, lr, b1, b2, eps, wd, step_size, inv_sqrt_bc2;
  int adamw, zero_grads;
  unsigned short* mirror;     // optional bf16 copy of the updated parameters (precision 'bf16' weight mirror)
  size_t mirror_ps;           // > 0: the mirror holds the THREE bf16 pieces of every parameter, piece p at mirror + p * mirror_ps (precision 'fp32x3')
  const unsigned short* g16;  // optional: read the gradient from this bf16 buffer (reduced data-parallel payload); g is still zeroed
  // row split of a table (round 6, uniter_adam_step_rows): chunk c belongs to row c / row_chunks; only the rows whose mask byte is
  // (!= 0) == (rows_want != 0) are updated.  rows_want == 0 also means "these rows received no gradient": g is taken as zero
  // without being read (and is not cleared)
  const uint8_t* rowmask;
  int row_chunks, rows_want;
  // paired-row weight mirror (round 6, uniter_adam_step_x3p): the launch walks the buffer in the MIRROR's order.  vsrc[c] = {s0, s1}
  // (elements from the flat buffers' start; s0 < 0 / no table: chunk c is its own source): the two 32-element units of chunk c of the
  // mirror are the parameters s0 .. s0 + 31 and s1 .. s1 + 31 -- one unit of row 2 q and the same unit of row 2 q + 1.  p, g, m, v are
  // then read and written in whole 128-byte lines (a unit is 32 floats) and the mirror in whole lines too (two neighbouring 64-byte
  // units); walking in the parameters' order instead left half-written mirror lines behind every wave (adam_kernel 76 -> 95 us).
  // The table pointer is that of this launch's first chunk; vsrc_base = the launch's first element (its offsets are absolute)
  const int2* vsrc;
  long vsrc_base;
};

#define NT_LOAD(base, idx) __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base) + (idx))
__device__ __forceinline__ void adam_update4(const AdamArgs& a, float coef, float wd, f32x4& p, const f32x4& g, f32x4& m, f32x4& v) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float gg = g[e] * coef;
    float pp = p[e];
    if (a.adamw) pp *= 1.0f - a.lr * wd;
    else gg += wd * pp;
    m[e] = a.b1 * m[e] + (1.0f - a.b1) * gg;
    v[e] = a.b2 * v[e] + (1.0f - a.b2) * gg * gg;
    const float denom = sqrtf(v[e]) * a.inv_sqrt_bc2 + a.eps;
    p[e] = pp - a.step_size * (m[e] / denom);
  }
}
__device__ __forceinline__ void adam_store4(const AdamArgs& a, size_t i, size_t si, const f32x4& p, const f32x4& m, const f32x4& v, bool clear) {
  // the update streams 34 bytes per parameter once: non-temporal accesses, so that it does not evict the operand panels of
  // the forward kernels that share the chip with it from the L2s.  si: the item's place in the parameter buffers, i: in the mirror
  __builtin_nontemporal_store(p, reinterpret_cast<f32x4*>(a.p) + si);
  if (a.mirror) {
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    const bf16x4_t o = {(__bf16)p[0], (__bf16)p[1], (__bf16)p[2], (__bf16)p[3]};
    bf16x4_t* m0 = reinterpret_cast<bf16x4_t*>(a.mirror) + i;      // (i: the item's place in the mirror's order)
    *m0 = o;
    if (a.mirror_ps) {      // x = x1 + x2 + x3 exactly (round-to-nearest residuals): the operands of csrc/gemm_split3.hip
      f32x4 r = {p[0] - (float)o[0], p[1] - (float)o[1], p[2] - (float)o[2], p[3] - (float)o[3]};
      const bf16x4_t o2 = {(__bf16)r[0], (__bf16)r[1], (__bf16)r[2], (__bf16)r[3]};
      *reinterpret_cast<bf16x4_t*>(reinterpret_cast<unsigned short*>(m0) + a.mirror_ps) = o2;
      r = f32x4{r[0] - (float)o2[0], r[1] - (float)o2[1], r[2] - (float)o2[2], r[3] - (float)o2[3]};
      *reinterpret_cast<bf16x4_t*>(reinterpret_cast<unsigned short*>(m0) + 2 * a.mirror_ps) = bf16x4_t{(__bf16)r[0], (__bf16)r[1], (__bf16)r[2], (__bf16)r[3]};
    }
  }
  __builtin_nontemporal_store(m, reinterpret_cast<f32x4*>(a.m) + si);
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(a.v) + si);
  // zero_grad: only where something was written -- four of five rows of the embeddings' block (the word table away from
  // the batch's tokens) hold zeros already
  if (clear) __builtin_nontemporal_store(f32x4{0.f, 0.f, 0.f, 0.f}, reinterpret_cast<f32x4*>(a.g) + si);
}
__device__ __forceinline__ bool any_nonzero(const f32x4& g) { return g[0] != 0.f || g[1] != 0.f || g[2] != 0.f || g[3] != 0.f; }

// Two 16-byte elements per thread and iteration, all eight loads issued before the arithmetic: a grid of one or two
// workgroups per CU (the launch that shares the chip with the next forward, see trainer.FusedAdam) still keeps
// 32-64 KB per CU in flight.
__global__ __launch_bounds__(256) void adam_kernel(const AdamArgs a) {
  float coef = a.gscale;
  if (a.max_norm > 0.f && a.sumsq) {
    const float total = (float)sqrt(a.sumsq[0]) * a.gscale;         // norm of the averaged grads
    const float c = a.max_norm / (total + 1e-6f);                   // clip_grad_norm_
    coef *= c < 1.0f ? c : 1.0f;
  }
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n4; i += 2 * stride) {
    const size_t j = i + stride;
    const bool two = j < a.n4;
    // where the item sits in the parameter buffers: its own place, or -- a chunk of a paired tensor, walked in the mirror's order --
    // one of the two source units the table names
    size_t si = i, sj = j;
    if (a.vsrc) {
      const int2 vi = a.vsrc[(i * 4) / CHUNK];
      if (vi.x >= 0) si = (size_t)((long)((((i * 4) >> 5) & 1) ? vi.y : vi.x) - a.vsrc_base + (long)((i * 4) & 31)) / 4;
      if (two) {
        const int2 vj = a.vsrc[(j * 4) / CHUNK];
        if (vj.x >= 0) sj = (size_t)((long)((((j * 4) >> 5) & 1) ? vj.y : vj.x) - a.vsrc_base + (long)((j * 4) & 31)) / 4;
      }
    }
    uint8_t f0 = a.flags[(si * 4) / CHUNK];
    uint8_t f1 = two ? a.flags[(sj * 4) / CHUNK] : 0;
    if (a.rowmask) {      // (one table, split by rows: this launch takes the rows on its side of the mask)
      if (f0 && (a.rowmask[((si * 4) / CHUNK) / a.row_chunks] != 0) != (a.rows_want != 0)) f0 = 0;
      if (f1 && (a.rowmask[((sj * 4) / CHUNK) / a.row_chunks] != 0) != (a.rows_want != 0)) f1 = 0;
    }
    const bool no_g = a.rowmask && a.rows_want == 0;      // rows without a gradient this step: g == 0, unread
    f32x4 p0, g0 = {0.f, 0.f, 0.f, 0.f}, m0, v0, p1, g1 = g0, m1, v1;
    if (f0) {
      p0 = NT_LOAD(a.p, si); if (!no_g) g0 = a.g16 ? widen4(a.g16, si) : NT_LOAD(a.g, si); m0 = NT_LOAD(a.m, si); v0 = NT_LOAD(a.v, si);
    }
    if (f1) {
      p1 = NT_LOAD(a.p, sj); if (!no_g) g1 = a.g16 ? widen4(a.g16, sj) : NT_LOAD(a.g, sj); m1 = NT_LOAD(a.m, sj); v1 = NT_LOAD(a.v, sj);
    }
    // (gradients read from the bf16 payload: the fp32 buffer holds this rank's own sums, cleared whatever the payload says)
    if (f0) {
      const bool c0 = !no_g && a.zero_grads && !(f0 & 4) && (a.g16 != nullptr || any_nonzero(g0));
      adam_update4(a, coef, (f0 & 3) == 2 ? a.wd : 0.f, p0, g0, m0, v0); adam_store4(a, i, si, p0, m0, v0, c0);
    }
    if (f1) {
      const bool c1 = !no_g && a.zero_grads && !(f1 & 4) && (a.g16 != nullptr || any_nonzero(g1));
      adam_update4(a, coef, (f1 & 3) == 2 ? a.wd : 0.f, p1, g1, m1, v1); adam_store4(a, j, sj, p1, m1, v1, c1);
    }
  }
}

inline int sumsq_blocks(size_t n) {
  size_t b = (n / 4 + 255) / 256;
  return (int)(b < 2048 ? (b < 1 ? 1 : b) : 2048);
}

}  // namespace

extern "C" size_t uniter_grad_sumsq_ws_bytes(size_t n) { return (size_t)sumsq_blocks(n) * sizeof(double); }

extern "C" int uniter_grad_sumsq(const float* grads, const uint8_t* chunk_flags, size_t n, double* sumsq,
                                 void* ws, size_t ws_bytes, void* stream) {
  UCHECK_ARG(grads && chunk_flags && sumsq && ws, "grad_sumsq: null pointer");
  UCHECK_SHAPE(n % CHUNK == 0, "grad_sumsq: n must be a multiple of 64");
  UCHECK_ARG(ws_bytes >= uniter_grad_sumsq_ws_bytes(n), "grad_sumsq: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int nb = sumsq_blocks(n);
  hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, st, grads, chunk_flags, n / 4, (double*)ws, (const unsigned short*)nullptr);
  UCHECK_LAUNCH();
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(1024), 0, st, (const double*)ws, nb, sumsq);
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_grad_sumsq_bf16(const void* grads_bf16, const uint8_t* chunk_flags, size_t n, double* sumsq,
                                      void* ws, size_t ws_bytes, void* stream) {
  UCHECK_ARG(grads_bf16 && chunk_flags && sumsq && ws, "grad_sumsq_bf16: null pointer");
  UCHECK_SHAPE(n % CHUNK == 0 && ((uintptr_t)grads_bf16 & 7) == 0, "grad_sumsq_bf16: n must be a multiple of 64, 8-byte aligned");
  UCHECK_ARG(ws_bytes >= uniter_grad_sumsq_ws_bytes(n), "grad_sumsq_bf16: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int nb = sumsq_blocks(n);
  hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, st, (const float*)nullptr, chunk_flags, n / 4, (double*)ws,
                     (const unsigned short*)grads_bf16);
  UCHECK_LAUNCH();
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(1024), 0, st, (const double*)ws, nb, sumsq);
  UCHECK_LAUNCH();
  return 0;
}

// One slice of the norm, unreduced: `nblocks` workgroups leave their partial sums in parts[0 .. nblocks); the caller joins
// the slices' partials with uniter_sumsq_combine.  chunk_flags NULL = every chunk counts (a parameter without a gradient
// this step holds zeros: the buffer is cleared by the optimizer step).
extern "C" int uniter_grad_sumsq_part(const float* grads, const uint8_t* chunk_flags, size_t n, double* parts, int nblocks,
                                      void* stream) {
  UCHECK_ARG(grads && parts && nblocks >= 1 && nblocks <= 2048, "grad_sumsq_part: bad argument");
  UCHECK_SHAPE(n % 4 == 0 && ((uintptr_t)grads & 15) == 0, "grad_sumsq_part: n must be a multiple of 4, 16-byte aligned");
  hipLaunchKernelGGL(sumsq_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, grads, chunk_flags, n / 4, parts,
                     (const unsigned short*)nullptr);
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_sumsq_combine(const double* parts, int n, double* out, void* stream) {
  UCHECK_ARG(parts && out && n >= 1, "sumsq_combine: bad argument");
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, parts, n, out);
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                                const uint8_t* chunk_flags, size_t n, const double* sumsq, float grad_scale,
                                float max_norm, float lr, float beta1, float beta2, float eps,
                                float weight_decay, int step, int adamw, int zero_grads, void* stream) {
  return uniter_adam_step_mirror(params, grads, exp_avg, exp_avg_sq, chunk_flags, n, sumsq, grad_scale, max_norm, lr,
                                 beta1, beta2, eps, weight_decay, step, adamw, zero_grads, nullptr, stream);
}

extern "C" int uniter_adam_step_mirror(float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                                       const uint8_t* chunk_flags, size_t n, const double* sumsq, float grad_scale,
                                       float max_norm, float lr, float beta1, float beta2, float eps,
                                       float weight_decay, int step, int adamw, int zero_grads,
                                       void* mirror_bf16, void* stream) {
  return uniter_adam_step_ex(params, grads, exp_avg, exp_avg_sq, chunk_flags, n, sumsq, grad_scale, max_norm, lr, beta1,
                             beta2, eps, weight_decay, step, adamw, zero_grads, mirror_bf16, 0, stream);
}

extern "C" int uniter_adam_step_ex(float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                                   const uint8_t* chunk_flags, size_t n, const double* sumsq, float grad_scale,
                                   float max_norm, float lr, float beta1, float beta2, float eps,
                                   float weight_decay, int step, int adamw, int zero_grads,
                                   void* mirror_bf16, int max_workgroups, void* stream) {
  return uniter_adam_step_g16(params, grads, nullptr, exp_avg, exp_avg_sq, chunk_flags, n, sumsq, grad_scale, max_norm, lr,
                              beta1, beta2, eps, weight_decay, step, adamw, zero_grads, mirror_bf16, max_workgroups, stream);
}

extern "C" int uniter_adam_step_g16(float* params, float* grads, const void* grads_bf16, float* exp_avg,
                                    float* exp_avg_sq, const uint8_t* chunk_flags, size_t n, const double* sumsq,
                                    float grad_scale, float max_norm, float lr, float beta1, float beta2, float eps,
                                    float weight_decay, int step, int adamw, int zero_grads, void* mirror_bf16,
                                    int max_workgroups, void* stream) {
  return uniter_adam_step_x3(params, grads, grads_bf16, exp_avg, exp_avg_sq, chunk_flags, n, sumsq, grad_scale, max_norm, lr,
                             beta1, beta2, eps, weight_decay, step, adamw, zero_grads, mirror_bf16, 0, max_workgroups, stream);
}

static int adam_step_impl(float* params, float* grads, const void* grads_bf16, float* exp_avg,
                          float* exp_avg_sq, const uint8_t* chunk_flags, size_t n, const double* sumsq,
                          float grad_scale, float max_norm, float lr, float beta1, float beta2, float eps,
                          float weight_decay, int step, int adamw, int zero_grads, void* mirror_bf16,
                          size_t mirror_piece_stride, int max_workgroups, const uint8_t* row_mask, int row_chunks, int rows_want,
                          const void* vsrc, long vsrc_base, void* stream);

extern "C" int uniter_adam_step_x3(float* params, float* grads, const void* grads_bf16, float* exp_avg,
                                   float* exp_avg_sq, const uint8_t* chunk_flags, size_t n, const double* sumsq,
                                   float grad_scale, float max_norm, float lr, float beta1, float beta2, float eps,
                                   float weight_decay, int step, int adamw, int zero_grads, void* mirror_bf16,
                                   size_t mirror_piece_stride, int max_workgroups, void* stream) {
  return adam_step_impl(params, grads, grads_bf16, exp_avg, exp_avg_sq, chunk_flags, n, sumsq, grad_scale, max_norm, lr, beta1, beta2,
                        eps, weight_decay, step, adamw, zero_grads, mirror_bf16, mirror_piece_stride, max_workgroups, nullptr, 0, 0, nullptr, 0, stream);
}

// uniter_adam_step_x3 that writes the weight pieces in the PAIRED-ROW layout (round 6) by walking the buffer in the MIRROR's order:
// pair_src points at the source table's entry for this launch's first 64-element chunk -- two int32 per chunk, the flat-buffer
// element offsets of the chunk's two 32-element units ({s0 < 0, ..} = the chunk is its own source; ParamStore.pair_src);
// first_element = the flat-buffer offset of `params` (the table's offsets are absolute).
extern "C" int uniter_adam_step_x3p(float* params, float* grads, const void* grads_bf16, float* exp_avg,
                                    float* exp_avg_sq, const uint8_t* chunk_flags, size_t n, const double* sumsq,
                                    float grad_scale, float max_norm, float lr, float beta1, float beta2, float eps,
                                    float weight_decay, int step, int adamw, int zero_grads, void* mirror,
                                    size_t mirror_piece_stride, const int* pair_src, size_t first_element, int max_workgroups, void* stream) {
  UCHECK_ARG(!pair_src || (mirror && mirror_piece_stride > 0 && first_element % CHUNK == 0 && ((uintptr_t)pair_src & 7) == 0),
             "adam_step_x3p: a source table needs the x3 mirror, a launch that starts on a chunk and an 8-byte aligned table");
  return adam_step_impl(params, grads, grads_bf16, exp_avg, exp_avg_sq, chunk_flags, n, sumsq, grad_scale, max_norm, lr, beta1, beta2,
                        eps, weight_decay, step, adamw, zero_grads, mirror, mirror_piece_stride, max_workgroups, nullptr, 0, 0, pair_src,
                        (long)first_element, stream);
}

// The update of ONE table split by rows (round 6): a fine-tuning step touches at most B x T of the word-embedding table's 28996 rows
// (model/model.py:220-221), yet torch.optim.Adam with weight_decay updates every row (utils/optim_utils.py:33-40: g += wd p) -- 0.62 of
// the step's 3.18 GB -- and the next forward's text branch waits for all of it.  A row WITHOUT a gradient is updated from g = 0, which
// neither the clip coefficient nor the gradient exchange can change (0 x c = 0): its update does not depend on this step's backward
// pass at all.  So the table's launch is two: rows_touched = 0 -- the rows whose mask byte is 0, g taken as zero (never read), any
// time after the previous step's update, e.g. beside the backward pass; rows_touched = 1 -- the masked rows with their gradients,
// clipped, behind the backward pass as before (a few per cent of the table).  Same arithmetic per element: bit-identical parameters.
// params .. chunk_flags point at the TABLE (row 0); n = rows x row_len with row_len % 64 == 0; row_mask one byte per row.
extern "C" int uniter_adam_step_rows(float* params, float* grads, float* exp_avg, float* exp_avg_sq, const uint8_t* chunk_flags,
                                     size_t n, const double* sumsq, float grad_scale, float max_norm, float lr, float beta1,
                                     float beta2, float eps, float weight_decay, int step, int adamw, int zero_grads,
                                     const uint8_t* row_mask, int row_len, int rows_touched, int max_workgroups, void* stream) {
  UCHECK_ARG(row_mask && row_len > 0 && row_len % CHUNK == 0 && n % (size_t)row_len == 0, "adam_step_rows: row_len must be a multiple of 64 dividing n");
  return adam_step_impl(params, grads, nullptr, exp_avg, exp_avg_sq, chunk_flags, n, rows_touched ? sumsq : nullptr, grad_scale,
                        rows_touched ? max_norm : 0.f, lr, beta1, beta2, eps, weight_decay, step, adamw, zero_grads, nullptr, 0,
                        max_workgroups, row_mask, row_len / CHUNK, rows_touched ? 1 : 0, nullptr, 0, stream);
}

static int adam_step_impl(float* params, float* grads, const void* grads_bf16, float* exp_avg,
                          float* exp_avg_sq, const uint8_t* chunk_flags, size_t n, const double* sumsq,
                          float grad_scale, float max_norm, float lr, float beta1, float beta2, float eps,
                          float weight_decay, int step, int adamw, int zero_grads, void* mirror_bf16,
                          size_t mirror_piece_stride, int max_workgroups, const uint8_t* row_mask, int row_chunks, int rows_want,
                          const void* vsrc, long vsrc_base, void* stream) {
  UCHECK_SHAPE(mirror_piece_stride % 4 == 0 && (mirror_piece_stride == 0 || mirror_bf16), "adam_step: bad mirror piece stride");
  UCHECK_ARG(params && grads && exp_avg && exp_avg_sq && chunk_flags, "adam_step: null pointer");
  UCHECK_SHAPE(((uintptr_t)grads_bf16 & 7) == 0, "adam_step: bf16 gradients must be 8-byte aligned");
  UCHECK_SHAPE(n % CHUNK == 0, "adam_step: n must be a multiple of 64");
  UCHECK_ARG(step >= 1, "adam_step: step must be >= 1");
  UCHECK_ARG(max_norm <= 0.f || sumsq, "adam_step: clipping needs sumsq");
  AdamArgs a;
  a.p = params; a.g = grads; a.m = exp_avg; a.v = exp_avg_sq; a.flags = chunk_flags; a.n4 = n / 4;
  a.sumsq = sumsq; a.gscale = grad_scale; a.max_norm = max_norm; a.lr = lr; a.b1 = beta1; a.b2 = beta2;
  a.eps = eps; a.wd = weight_decay; a.adamw = adamw; a.zero_grads = zero_grads;
  a.mirror = (unsigned short*)mirror_bf16;
  a.mirror_ps = mirror_piece_stride;
  a.g16 = (const unsigned short*)grads_bf16;
  a.rowmask = row_mask; a.row_chunks = row_chunks > 0 ? row_chunks : 1; a.rows_want = rows_want;
  a.vsrc = (const int2*)vsrc; a.vsrc_base = vsrc_base;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  a.step_size = (float)((double)lr / bc1);
  a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  int nb = sumsq_blocks(n);
  if (max_workgroups > 0) {
    // an explicit grid: up to the one in which every thread takes its two items once (short-lived workgroups: what a persistent
    // product of the forward pass that wants the CU waits for is ONE workgroup's lifetime)
    const size_t full = (a.n4 + 511) / 512;
    nb = (int)(full < (size_t)max_workgroups ? (full < 1 ? 1 : full) : (size_t)max_workgroups);
  }
  hipLaunchKernelGGL(adam_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, a);
  UCHECK_LAUNCH();
  return 0;
}
