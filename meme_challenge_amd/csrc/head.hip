// Pooler + classification head + loss.
//
// Replaces BertPooler.forward (model/layer.py:179-185: h[:,0] -> dense -> tanh),
// MemeUniter.linear (model/meme_uniter.py:19-20), nn.BCEWithLogitsLoss(pos_weight)
// (train_template.py:65,98-99) and their autograd.  B is small (16): these are
// latency-bound VALU kernels, one wave per output row of the weight.
#include "common.h"

namespace {

// pooled[b][n] = tanh(sum_k h0[b][k] W[n][k] + bias[n]); one wave per (n, group of 4 batch rows):
// the weight row is read once per wave and the four dot products reduce together
__global__ __launch_bounds__(256) void pooler_fwd_kernel(const float* __restrict__ hidden,
                                                         const float* __restrict__ W,
                                                         const float* __restrict__ bias,
                                                         float* __restrict__ pooled, int B, int L, int H) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int b0 = blockIdx.y * 4;
  if (n >= H) return;
  const float* w = W + (size_t)n * H;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if ((H & 3) == 0) {        // 16-byte loads: 3 iterations instead of 12 at H = 768 (the kernel is a chain of load latencies)
    const int H4 = H >> 2;
    for (int k4 = lane; k4 < H4; k4 += 64) {
      const f32x4 wk = reinterpret_cast<const f32x4*>(w)[k4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (b0 + j < B) {
          const f32x4 x = reinterpret_cast<const f32x4*>(hidden + (size_t)(b0 + j) * L * H)[k4];
          s[j] += (x[0] * wk[0] + x[1] * wk[1]) + (x[2] * wk[2] + x[3] * wk[3]);
        }
    }
  } else {
    for (int k = lane; k < H; k += 64) {
      const float wk = w[k];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (b0 + j < B) s[j] += hidden[(size_t)(b0 + j) * L * H + k] * wk;
    }
  }
  const float bn = bias[n];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float t = wave_sum(s[j]);
    if (lane == 0 && b0 + j < B) pooled[(size_t)(b0 + j) * H + n] = tanhf(t + bn);
  }
}

// one wave per n: dbias[n] += sum_b dpre[b][n]; dW[n][:] += sum_b dpre[b][n] * h0[b][:]
// dpre[b][n] lives in lane b (chunks of 64 batch rows) and is broadcast by shuffles
__global__ __launch_bounds__(256) void pooler_bwd_w_kernel(const float* __restrict__ dpooled,
                                                           const float* __restrict__ pooled,
                                                           const float* __restrict__ hidden,
                                                           float* __restrict__ dW, float* __restrict__ dbias,
                                                           int B, int L, int H) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= H) return;
  float sb = 0.f;
  for (int bc = 0; bc < B; bc += 64) {
    const int nb = min(64, B - bc);
    float dpre = 0.f;
    if (lane < nb) {
      const float p = pooled[(size_t)(bc + lane) * H + n];
      dpre = dpooled[(size_t)(bc + lane) * H + n] * (1.0f - p * p);
    }
    sb += wave_sum(dpre);
    if ((H & 3) == 0) {
      const int H4 = H >> 2;
      for (int k4 = lane; k4 < H4; k4 += 64) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int b = 0; b < nb; ++b) {
          const float d = __shfl(dpre, b, 64);
          const f32x4 x = reinterpret_cast<const f32x4*>(hidden + (size_t)(bc + b) * L * H)[k4];
          acc[0] += d * x[0]; acc[1] += d * x[1]; acc[2] += d * x[2]; acc[3] += d * x[3];
        }
        f32x4* o = reinterpret_cast<f32x4*>(dW + (size_t)n * H) + k4;
        f32x4 t = *o;
        t[0] += acc[0]; t[1] += acc[1]; t[2] += acc[2]; t[3] += acc[3];
        *o = t;
      }
    } else {
      for (int k = lane; k < H; k += 64) {
        float s = 0.f;
        for (int b = 0; b < nb; ++b) s += __shfl(dpre, b, 64) * hidden[(size_t)(bc + b) * L * H + k];
        dW[(size_t)n * H + k] += s;
      }
    }
  }
  if (lane == 0) dbias[n] += sb;
}

// dh0[b][k] (+)= sum_n dpre[b][n] W[n][k]: workgroup = (64 columns k, one b); its 4 waves split n
// 16 waves (was 4): the loop over n is a chain of dependent load latencies, 48 iterations instead of 192 at H = 768
constexpr int PBX_WAVES = 16;
__global__ __launch_bounds__(64 * PBX_WAVES) void pooler_bwd_x_kernel(const float* __restrict__ dpooled,
                                                           const float* __restrict__ pooled,
                                                           const float* __restrict__ W,
                                                           float* __restrict__ dhidden, int B, int L, int H,
                                                           int beta) {
  __shared__ float red[PBX_WAVES][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane;
  const int b = blockIdx.y;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (k < H) {
    const float* dp = dpooled + (size_t)b * H;
    const float* pp = pooled + (size_t)b * H;
    constexpr int S = PBX_WAVES;
    int n = wv;
    for (; n + 3 * S < H; n += 4 * S) {
      const float p0 = pp[n], p1 = pp[n + S], p2 = pp[n + 2 * S], p3 = pp[n + 3 * S];
      s0 += dp[n] * (1.0f - p0 * p0) * W[(size_t)n * H + k];
      s1 += dp[n + S] * (1.0f - p1 * p1) * W[(size_t)(n + S) * H + k];
      s2 += dp[n + 2 * S] * (1.0f - p2 * p2) * W[(size_t)(n + 2 * S) * H + k];
      s3 += dp[n + 3 * S] * (1.0f - p3 * p3) * W[(size_t)(n + 3 * S) * H + k];
    }
    for (; n < H; n += S) {
      const float p0 = pp[n];
      s0 += dp[n] * (1.0f - p0 * p0) * W[(size_t)n * H + k];
    }
  }
  red[wv][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (wv == 0 && k < H) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < PBX_WAVES; ++w) s += red[w][lane];
    float* d = dhidden + (size_t)b * L * H + k;
    *d = beta ? *d + s : s;
  }
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
  hipLaunchKernelGGL(pooler_fwd_kernel, dim3((H + 3) / 4, (B + 3) / 4), dim3(256), 0, (hipStream_t)stream, hidden, Wp, bp,
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
    hipLaunchKernelGGL(pooler_bwd_x_kernel, dim3((H + 63) / 64, B), dim3(64 * PBX_WAVES), 0, st, dpooled, pooled, Wp,
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
