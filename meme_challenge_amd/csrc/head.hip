// Pooler + classification head + loss.
//
// Replaces BertPooler.forward (model/layer.py:179-185: h[:,0] -> dense -> tanh),
// MemeUniter.linear (model/meme_uniter.py:19-20), nn.BCEWithLogitsLoss(pos_weight)
// (train_template.py:65,98-99) and their autograd.  B is small (16): these are
// latency-bound VALU kernels, one wave per output row of the weight.
#include "common.h"

namespace {

// pooled[b][n] = tanh(sum_k h0[b][k] W[n][k] + bias[n]); one wave per n, loops over b
__global__ __launch_bounds__(256) void pooler_fwd_kernel(const float* __restrict__ hidden,
                                                         const float* __restrict__ W,
                                                         const float* __restrict__ bias,
                                                         float* __restrict__ pooled, int B, int L, int H) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= H) return;
  const float* w = W + (size_t)n * H;
  const float bn = bias[n];
  for (int b = 0; b < B; ++b) {
    const float* h0 = hidden + (size_t)b * L * H;
    float s = 0.f;
    for (int k = lane; k < H; k += 64) s += h0[k] * w[k];
    s = wave_sum(s);
    if (lane == 0) pooled[(size_t)b * H + n] = tanhf(s + bn);
  }
}

// one wave per n: dbias[n] += sum_b dpre[b][n]; dW[n][:] += sum_b dpre[b][n] * h0[b][:]
__global__ __launch_bounds__(256) void pooler_bwd_w_kernel(const float* __restrict__ dpooled,
                                                           const float* __restrict__ pooled,
                                                           const float* __restrict__ hidden,
                                                           float* __restrict__ dW, float* __restrict__ dbias,
                                                           int B, int L, int H) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= H) return;
  float sb = 0.f;
  for (int k = lane; k < H; k += 64) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
      const float p = pooled[(size_t)b * H + n];
      const float dpre = dpooled[(size_t)b * H + n] * (1.0f - p * p);
      s += dpre * hidden[(size_t)b * L * H + k];
      if (k == lane) sb += dpre;
    }
    dW[(size_t)n * H + k] += s;
  }
  if (lane == 0) dbias[n] += sb;
}

// one thread per (b,k): dh0[b][k] (+)= sum_n dpre[b][n] W[n][k]
__global__ __launch_bounds__(256) void pooler_bwd_x_kernel(const float* __restrict__ dpooled,
                                                           const float* __restrict__ pooled,
                                                           const float* __restrict__ W,
                                                           float* __restrict__ dhidden, int B, int L, int H,
                                                           int beta) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (k >= H) return;
  float s = 0.f;
  for (int n = 0; n < H; ++n) {
    const float p = pooled[(size_t)b * H + n];
    s += dpooled[(size_t)b * H + n] * (1.0f - p * p) * W[(size_t)n * H + k];
  }
  float* d = dhidden + (size_t)b * L * H + k;
  *d = beta ? *d + s : s;
}

// y[b][c] = x[b] . W[c] + bias[c]; one wave per (b,c)
__global__ __launch_bounds__(256) void linear_small_fwd_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ W,
                                                               const float* __restrict__ bias,
                                                               float* __restrict__ y, int B, int H, int Cn) {
  const int lane = threadIdx.x & 63;
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= B * Cn) return;
  const int b = o / Cn, c = o - b * Cn;
  float s = 0.f;
  for (int k = lane; k < H; k += 64) s += x[(size_t)b * H + k] * W[(size_t)c * H + k];
  s = wave_sum(s);
  if (lane == 0) y[o] = s + bias[c];
}

// thread per k: dx[b][k] = sum_c dy[b][c] W[c][k] (all b);  dW[c][k] += sum_b dy[b][c] x[b][k];
// thread k < Cn additionally: db[k] += sum_b dy[b][k]
__global__ __launch_bounds__(256) void linear_small_bwd_kernel(const float* __restrict__ dy,
                                                               const float* __restrict__ x,
                                                               const float* __restrict__ W,
                                                               float* __restrict__ dx, float* __restrict__ dW,
                                                               float* __restrict__ db, int B, int H, int Cn) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < Cn && db) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dy[(size_t)b * Cn + k];
    db[k] += s;
  }
  if (k >= H) return;
  for (int b = 0; b < B && dx; ++b) {
    float s = 0.f;
    for (int c = 0; c < Cn; ++c) s += dy[(size_t)b * Cn + c] * W[(size_t)c * H + k];
    dx[(size_t)b * H + k] = s;
  }
  for (int c = 0; c < Cn && dW; ++c) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dy[(size_t)b * Cn + c] * x[(size_t)b * H + k];
    dW[(size_t)c * H + k] += s;
  }
}

// single workgroup; loss = mean_b [ (1-y) x + (1 + (pw-1) y) softplus(-x) ]
__global__ __launch_bounds__(256) void bce_logits_kernel(const float* __restrict__ logits,
                                                         const int64_t* __restrict__ labels, float pw,
                                                         float* __restrict__ loss, float* __restrict__ probs,
                                                         float* __restrict__ dlogits, float gscale, int B) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float x = logits[b];
    const float y = (float)labels[b];
    const float lw = 1.0f + (pw - 1.0f) * y;
    const float sp = log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.f);     // softplus(-x)
    acc += (1.0f - y) * x + lw * sp;
    const float sg = 1.0f / (1.0f + expf(-x));
    if (probs) probs[b] = sg;
    if (dlogits) dlogits[b] = ((1.0f - y) - lw * (1.0f - sg)) * (gscale / (float)B);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0 && loss) *loss = (red[0] + red[1] + red[2] + red[3]) / (float)B;
}

}  // namespace

extern "C" int uniter_pooler_fwd(const float* hidden, const float* Wp, const float* bp, float* pooled, int B,
                                 int L, int H, void* stream) {
  UCHECK_ARG(hidden && Wp && bp && pooled && B > 0 && L > 0 && H > 0, "pooler_fwd: bad argument");
  hipLaunchKernelGGL(pooler_fwd_kernel, dim3((H + 3) / 4), dim3(256), 0, (hipStream_t)stream, hidden, Wp, bp,
                     pooled, B, L, H);
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_pooler_bwd(const float* dpooled, const float* pooled, const float* hidden,
                                 const float* Wp, float* dWp, float* dbp, float* dhidden, int B, int L, int H,
                                 int beta_dhidden, void* stream) {
  UCHECK_ARG(dpooled && pooled && hidden && Wp && dWp && dbp && B > 0 && L > 0 && H > 0,
             "pooler_bwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(pooler_bwd_w_kernel, dim3((H + 3) / 4), dim3(256), 0, st, dpooled, pooled, hidden, dWp,
                     dbp, B, L, H);
  UCHECK_LAUNCH();
  if (dhidden) {
    hipLaunchKernelGGL(pooler_bwd_x_kernel, dim3((H + 255) / 256, B), dim3(256), 0, st, dpooled, pooled, Wp,
                       dhidden, B, L, H, beta_dhidden);
    UCHECK_LAUNCH();
  }
  return 0;
}

extern "C" int uniter_linear_small_fwd(const float* x, const float* W, const float* b, float* y, int B, int H,
                                       int Cn, void* stream) {
  UCHECK_ARG(x && W && b && y && B > 0 && H > 0 && Cn > 0, "linear_small_fwd: bad argument");
  hipLaunchKernelGGL(linear_small_fwd_kernel, dim3((B * Cn + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, W,
                     b, y, B, H, Cn);
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_linear_small_bwd(const float* dy, const float* x, const float* W, float* dx, float* dW,
                                       float* db, int B, int H, int Cn, void* stream) {
  UCHECK_ARG(dy && x && W && B > 0 && H > 0 && Cn > 0 && Cn <= H, "linear_small_bwd: bad argument");
  hipLaunchKernelGGL(linear_small_bwd_kernel, dim3((H + 255) / 256), dim3(256), 0, (hipStream_t)stream, dy, x,
                     W, dx, dW, db, B, H, Cn);
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_bce_logits(const float* logits, const int64_t* labels, float pos_weight, float* loss,
                                 float* probs, float* dlogits, float grad_scale, int B, void* stream) {
  UCHECK_ARG(logits && labels && B > 0, "bce_logits: bad argument");
  hipLaunchKernelGGL(bce_logits_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, labels,
                     pos_weight, loss, probs, dlogits, grad_scale, B);
  UCHECK_LAUNCH();
  return 0;
}
