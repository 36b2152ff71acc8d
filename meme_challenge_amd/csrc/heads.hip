// Building blocks of the pretraining heads (BASELINE config 5: ITM / MLM / MRFR; MRC / MRC-kl):
// row gather / scatter for "_compute_masked_hidden" (model/pretrain.py:129-133), per-row
// cross-entropy over the vocabulary (F.cross_entropy(..., reduction='none'), :122-124,:199) and
// element-wise MSE (F.mse_loss(..., reduction='none'), :150-151), forward and backward.
// The dense / tied-decoder contractions use uniter_gemm_f32, the LayerNorm uniter_ln_*.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void row_gather_kernel(const float* __restrict__ src,
                                                         const int64_t* __restrict__ idx,
                                                         float* __restrict__ dst, int n, int H4, int nsrc) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  long long s = idx[row];
  s = s < 0 ? 0 : (s >= nsrc ? nsrc - 1 : s);
  const f32x4* sp = reinterpret_cast<const f32x4*>(src) + (size_t)s * H4;
  f32x4* dp = reinterpret_cast<f32x4*>(dst) + (size_t)row * H4;
  for (int c = lane; c < H4; c += 64) dp[c] = sp[c];
}

// dst[idx[r]] += src[r]  (indices must be unique: one wave owns one destination row)
__global__ __launch_bounds__(256) void row_scatter_add_kernel(const float* __restrict__ src,
                                                              const int64_t* __restrict__ idx,
                                                              float* __restrict__ dst, int n, int H4, int ndst) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const long long s = idx[row];
  if (s < 0 || s >= ndst) return;
  const f32x4* sp = reinterpret_cast<const f32x4*>(src) + (size_t)row * H4;
  f32x4* dp = reinterpret_cast<f32x4*>(dst) + (size_t)s * H4;
  for (int c = lane; c < H4; c += 64) dp[c] += sp[c];
}

// one 256-thread workgroup per row: lse = log sum exp(logits), loss = lse - logits[target]
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits,
                                                     const int64_t* __restrict__ targets,
                                                     float* __restrict__ loss, float* __restrict__ lse,
                                                     int C, int ld) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* x = logits + (size_t)row * ld;
  float mx = -__builtin_huge_valf();
  for (int c = tid; c < C; c += 256) mx = fmaxf(mx, x[c]);
  mx = wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int c = tid; c < C; c += 256) s += expf(x[c] - mx);
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) {
    const float l = mx + logf(red[0] + red[1] + red[2] + red[3]);
    lse[row] = l;
    long long t = targets[row];
    t = t < 0 ? 0 : (t >= C ? C - 1 : t);
    loss[row] = l - x[t];
  }
}

// dlogits[r][c] = (softmax - onehot) * dloss[r]
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits,
                                                     const int64_t* __restrict__ targets,
                                                     const float* __restrict__ lse,
                                                     const float* __restrict__ dloss,
                                                     float* __restrict__ dlogits, int C, int ld) {
  const int row = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float g = dloss[row];
  const float p = expf(logits[(size_t)row * ld + c] - lse[row]);
  dlogits[(size_t)row * ld + c] = (p - (c == (int)targets[row] ? 1.0f : 0.0f)) * g;
}

// row-wise log-sum-exp shared by the KL kernels (256 threads per row)
__device__ __forceinline__ float row_lse_256(const float* __restrict__ x, int C, float* red) {
  const int tid = threadIdx.x;
  float mx = -__builtin_huge_valf();
  for (int c = tid; c < C; c += 256) mx = fmaxf(mx, x[c]);
  mx = wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int c = tid; c < C; c += 256) s += expf(x[c] - mx);
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float l = mx + logf(red[0] + red[1] + red[2] + red[3]);
  __syncthreads();
  return l;
}

// kl_div(log_softmax(x), t, 'none'): t (log t - log_softmax(x)), and 0 where t == 0 (xlogy)
__global__ __launch_bounds__(256) void kl_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ target,
                                                     float* __restrict__ loss, float* __restrict__ lse, int C, int ld) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* x = logits + (size_t)row * ld;
  const float* t = target + (size_t)row * ld;
  const float l = row_lse_256(x, C, red);
  if (tid == 0) lse[row] = l;
  for (int c = tid; c < C; c += 256) {
    const float tv = t[c];
    loss[(size_t)row * ld + c] = tv > 0.f ? tv * (logf(tv) - (x[c] - l)) : 0.f;
  }
}

// dx[j] = softmax[j] * sum_c(dloss[c] t[c]) - dloss[j] t[j]
__global__ __launch_bounds__(256) void kl_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ target,
                                                     const float* __restrict__ lse, const float* __restrict__ dloss,
                                                     float* __restrict__ dlogits, int C, int ld) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const size_t o = (size_t)row * ld;
  float s = 0.f;
  for (int c = tid; c < C; c += 256) s += dloss[o + c] * target[o + c];
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  s = red[0] + red[1] + red[2] + red[3];
  const float l = lse[row];
  for (int c = tid; c < C; c += 256)
    dlogits[o + c] = expf(logits[o + c] - l) * s - dloss[o + c] * target[o + c];
}

// one wave per row: first maximum of x[row, c0:C]
__global__ __launch_bounds__(256) void row_argmax_kernel(const float* __restrict__ x, int n, int C, int ld, int c0,
                                                         int64_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  float best = -__builtin_huge_valf();
  int at = 0x7fffffff;
  for (int c = c0 + lane; c < C; c += 64) {
    const float v = x[(size_t)row * ld + c];
    if (v > best) { best = v; at = c; }        // ascending c per lane: keeps the lane's first maximum
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oa = __shfl_xor(at, o, 64);
    if (ob > best || (ob == best && oa < at)) { best = ob; at = oa; }
  }
  if (lane == 0) out[row] = at == 0x7fffffff ? c0 : at;
}

__global__ void mse_fwd_kernel(const float* __restrict__ p, const float* __restrict__ t,
                               float* __restrict__ loss, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const float d = p[i] - t[i]; loss[i] = d * d; }
}
__global__ void mse_bwd_kernel(const float* __restrict__ p, const float* __restrict__ t,
                               const float* __restrict__ dloss, float* __restrict__ dp, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dp[i] = 2.0f * (p[i] - t[i]) * dloss[i];
}

__global__ void dgelu_mul_kernel(const float* __restrict__ dy, const float* __restrict__ u,
                                 float* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = dy[i] * dgelu_erf(u[i]);
}

}  // namespace

extern "C" int uniter_dgelu_mul(const float* dy, const float* u, float* out, size_t n, void* stream) {
  UCHECK_ARG(dy && u && out, "dgelu_mul: null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(dgelu_mul_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, u,
                     out, n);
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_row_gather(const float* src, const int64_t* idx, float* dst, int n, int H, int nsrc,
                                 void* stream) {
  UCHECK_ARG(src && idx && dst && n >= 0 && nsrc > 0, "row_gather: bad argument");
  UCHECK_SHAPE(H % 4 == 0, "row_gather: H must be a multiple of 4");
  if (n == 0) return 0;
  hipLaunchKernelGGL(row_gather_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, src, idx, dst, n,
                     H / 4, nsrc);
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_row_scatter_add(const float* src, const int64_t* idx, float* dst, int n, int H, int ndst,
                                      void* stream) {
  UCHECK_ARG(src && idx && dst && n >= 0 && ndst > 0, "row_scatter_add: bad argument");
  UCHECK_SHAPE(H % 4 == 0, "row_scatter_add: H must be a multiple of 4");
  if (n == 0) return 0;
  hipLaunchKernelGGL(row_scatter_add_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, src, idx, dst,
                     n, H / 4, ndst);
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_cross_entropy_fwd(const float* logits, const int64_t* targets, float* loss, float* lse,
                                        int n, int C, int ld, void* stream) {
  UCHECK_ARG(logits && targets && loss && lse && n >= 0 && C > 0 && ld >= C, "cross_entropy_fwd: bad argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(ce_fwd_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, logits, targets, loss, lse, C, ld);
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_cross_entropy_bwd(const float* logits, const int64_t* targets, const float* lse,
                                        const float* dloss, float* dlogits, int n, int C, int ld, void* stream) {
  UCHECK_ARG(logits && targets && lse && dloss && dlogits && n >= 0 && C > 0 && ld >= C,
             "cross_entropy_bwd: bad argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(ce_bwd_kernel, dim3((C + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, logits, targets,
                     lse, dloss, dlogits, C, ld);
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_kl_div_fwd(const float* logits, const float* target, float* loss, float* lse, int n, int C,
                                 int ld, void* stream) {
  UCHECK_ARG(logits && target && loss && lse && n >= 0 && C > 0 && ld >= C, "kl_div_fwd: bad argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(kl_fwd_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, logits, target, loss, lse, C, ld);
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_kl_div_bwd(const float* logits, const float* target, const float* lse, const float* dloss,
                                 float* dlogits, int n, int C, int ld, void* stream) {
  UCHECK_ARG(logits && target && lse && dloss && dlogits && n >= 0 && C > 0 && ld >= C, "kl_div_bwd: bad argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(kl_bwd_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, logits, target, lse, dloss, dlogits,
                     C, ld);
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_row_argmax(const float* x, int n, int C, int ld, int c0, int64_t* out, void* stream) {
  UCHECK_ARG(x && out && n >= 0 && C > 0 && ld >= C && c0 >= 0 && c0 < C, "row_argmax: bad argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(row_argmax_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, n, C, ld, c0, out);
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_mse_fwd(const float* pred, const float* target, float* loss, size_t n, void* stream) {
  UCHECK_ARG(pred && target && loss, "mse_fwd: null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(mse_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pred,
                     target, loss, n);
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_mse_bwd(const float* pred, const float* target, const float* dloss, float* dpred, size_t n,
                              void* stream) {
  UCHECK_ARG(pred && target && dloss && dpred, "mse_bwd: null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(mse_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pred,
                     target, dloss, dpred, n);
  UCHECK_LAUNCH();
  return 0;
}
