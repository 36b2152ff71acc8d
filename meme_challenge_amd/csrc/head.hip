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
    // (the loops' trip counts are wave-uniform: the shuffles below read lanes that a per-lane bound would have switched off --
    // H / 4 < 64 with more batch rows than that)
    if ((H & 3) == 0) {
      const int H4 = H >> 2;
      for (int k0 = 0; k0 < H4; k0 += 64) {
        const int k4 = k0 + lane;
        const bool on = k4 < H4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int b = 0; b < nb; ++b) {
          const float d = __shfl(dpre, b, 64);
          if (on) {
            const f32x4 x = reinterpret_cast<const f32x4*>(hidden + (size_t)(bc + b) * L * H)[k4];
            acc[0] += d * x[0]; acc[1] += d * x[1]; acc[2] += d * x[2]; acc[3] += d * x[3];
          }
        }
        if (on) {
          f32x4* o = reinterpret_cast<f32x4*>(dW + (size_t)n * H) + k4;
          f32x4 t = *o;
          t[0] += acc[0]; t[1] += acc[1]; t[2] += acc[2]; t[3] += acc[3];
          *o = t;
        }
      }
    } else {
      for (int k0 = 0; k0 < H; k0 += 64) {
        const int k = k0 + lane;
        float s = 0.f;
        for (int b = 0; b < nb; ++b) {
          const float d = __shfl(dpre, b, 64);
          if (k < H) s += d * hidden[(size_t)(bc + b) * L * H + k];
        }
        if (k < H) dW[(size_t)n * H + k] += s;
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


// ---- round 6: pooler + classifier as ONE forward launch and ONE backward launch ------------------------------------------------
// Between the encoder's last LayerNorm and the backward pass's first one the step ran seven latency-sized launches (pooler, classifier,
// loss, classifier backward, a fill of the encoder output's gradient, pooler backward x 2: 67 us of a 9.1-ms step with nothing beside
// them, profiles/r06_timeline_f32x3.txt).  The loss stays the trainer's (train_template.py:98-99); the two modules of
// model/meme_uniter.py:19-21 become one launch each way.

// forward: pooler_fwd_kernel's waves, then the LAST workgroup to finish (a ticket counter) computes the logits from the pooled rows.
// The pooled values cross workgroups -- and XCDs, whose L2s are not coherent with each other inside a launch -- so they are stored and
// re-read device-coherently (sc1), the idiom of the balanced walk's partial sums (gemm_split3.hip).
__global__ __launch_bounds__(256) void pool_head_fwd_kernel(const float* __restrict__ hidden, const float* __restrict__ W,
                                                            const float* __restrict__ bias, const float* __restrict__ Wl,
                                                            const float* __restrict__ bl, float* __restrict__ pooled,
                                                            float* __restrict__ logits, unsigned* __restrict__ ticket, int B, int L,
                                                            int H, int Cn) {
  __shared__ unsigned last_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 4 + wave;
  const int b0 = blockIdx.y * 4;
  if (n < H) {
    const float* w = W + (size_t)n * H;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if ((H & 3) == 0) {
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
      if (lane == 0 && b0 + j < B)
        __hip_atomic_store(pooled + (size_t)(b0 + j) * H + n, tanhf(t + bn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's pooled values are in memory ...
  __syncthreads();                                        // ... and so are the other three waves'
  if (threadIdx.x == 0) {
    // Two levels of counters (16 groups 256 bytes apart, then one): 768 read-modify-writes of ONE address are served one after the other
    // (23 ns each: 18 us behind the pooler's own 10); 48 per group address in parallel, then 16.  Every counter is left zero.
    const unsigned total = gridDim.x * gridDim.y, id = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned g = id & 15u, gsize = (total + 15u - g) >> 4;          // workgroups whose id is g modulo 16
    unsigned* tg = ticket + g * 64u;
    unsigned last = 0u;
    if (__hip_atomic_fetch_add(tg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1u) {
      __hip_atomic_store(tg, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned* tt = ticket + 16u * 64u;
      if (__hip_atomic_fetch_add(tt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (total < 16u ? total : 16u) - 1u) {
        __hip_atomic_store(tt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = 1u;
      }
    }
    last_s = last;
  }
  __syncthreads();
  if (last_s == 0u) return;
  asm volatile("" ::: "memory");
  // logits[b][c] = pooled[b] . Wl[c] + bl[c]: one wave per (b, c), the accumulation order of linear_small_fwd_kernel.  The pooled rows
  // come through a buffer descriptor with the sc1 bit (device-coherent loads the compiler is free to issue back to back: the per-element
  // atomic loads of a first version were a chain of 48 memory latencies, 34 us for the launch)
  constexpr int AUX_SC1 = 16;
  const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(pooled, 0, (int)((size_t)B * H * 4), 0x00020000);
  // every task's loads are issued before the first is waited for: two tasks (2 x up to 16 loads per lane) at a time
  constexpr int PF = 16;
  for (int o0 = wave; o0 < B * Cn; o0 += 8) {
    float pv[2][PF], acc[2] = {0.f, 0.f};
    for (int k0 = 0; k0 < H; k0 += 64 * PF) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int o = o0 + 4 * t;
        const int b = o < B * Cn ? o / Cn : 0;
#pragma unroll
        for (int j = 0; j < PF; ++j) {
          const int k = k0 + lane + 64 * j;
          // (a column beyond H: an offset beyond the descriptor's range reads as zero)
          pv[t][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsP, k < H ? (b * H + k) * 4 : 0x7ffffff0, 0, AUX_SC1));
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int o = o0 + 4 * t;
        if (o < B * Cn) {
          const int c = o - (o / Cn) * Cn;
#pragma unroll
          for (int j = 0; j < PF; ++j) {
            const int k = k0 + lane + 64 * j;
            if (k < H) acc[t] += pv[t][j] * Wl[(size_t)c * H + k];
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int o = o0 + 4 * t;
      const float a = wave_sum(acc[t]);
      if (lane == 0 && o < B * Cn) logits[o] = a + bl[o - (o / Cn) * Cn];
    }
  }
}

// backward: one launch of 16-wave workgroups in three roles (by blockIdx.x):
//   [0, nW)          one wave (of four) per pooler output n: dWp[n][:] += sum_b dpre[b][n] h0[b][:], dbp[n] += sum_b dpre[b][n]   (pooler_bwd_w_kernel)
//   [nW, nW + nX)    (64 columns k, one sample b): dh0[b][k] = sum_n dpre[b][n] Wp[n][k]                                  (pooler_bwd_x_kernel)
//   nW + nX ..       thread per k: dWl[c][k] += sum_b dlogits[b][c] pooled[b][k]; dbl[c] += sum_b dlogits[b][c]           (linear_small_bwd_kernel)
// every role recomputes the pooled rows' gradient it needs from the logits' gradient on the fly (Cn multiply-adds per element):
//   dpooled[b][n] = sum_c dlogits[b][c] Wl[c][n],  dpre[b][n] = dpooled[b][n] (1 - pooled[b][n]^2)
constexpr int PHB_WAVES = 16;
constexpr int PHB_MAX_H = 4096;     // (the X role keeps one sample's row of H gradients in LDS)
__device__ __forceinline__ float phb_dpre(const float* __restrict__ dlogits, const float* __restrict__ Wl,
                                          const float* __restrict__ pooled, int b, int n, int H, int Cn) {
  float s = 0.f;
  for (int c = 0; c < Cn; ++c) s += dlogits[(size_t)b * Cn + c] * Wl[(size_t)c * H + n];
  const float p = pooled[(size_t)b * H + n];
  return s * (1.0f - p * p);
}

__global__ __launch_bounds__(64 * PHB_WAVES) void pool_head_bwd_kernel(const float* __restrict__ dlogits,
                                                                       const float* __restrict__ pooled,
                                                                       const float* __restrict__ hidden,
                                                                       const float* __restrict__ Wp, const float* __restrict__ Wl,
                                                                       float* __restrict__ dWp, float* __restrict__ dbp,
                                                                       float* __restrict__ dWl, float* __restrict__ dbl,
                                                                       float* __restrict__ dhidden, int B, int L, int H, int Cn,
                                                                       int nW, int nX, int beta) {
  __shared__ float red[PHB_WAVES][64];
  __shared__ float dpre_s[PHB_MAX_H];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int blk = blockIdx.x;
  if (blk < nW) {
    // (four of the sixteen waves: one pooler output each, spread over H / 4 workgroups -- the wave's loop over the batch rows of
    // hidden is a chain of load latencies that sixteen waves on one CU only queue behind each other)
    if (wv >= 4) return;
    const int n = blk * 4 + wv;
    if (n >= H) return;
    float sb = 0.f;
    for (int bc = 0; bc < B; bc += 64) {
      const int nb = min(64, B - bc);
      const float dpre = lane < nb ? phb_dpre(dlogits, Wl, pooled, bc + lane, n, H, Cn) : 0.f;
      sb += wave_sum(dpre);
      if ((H & 3) == 0) {                     // (wave-uniform trip counts around the shuffles, as in pooler_bwd_w_kernel)
        const int H4 = H >> 2;
        for (int k0 = 0; k0 < H4; k0 += 64) {
          const int k4 = k0 + lane;
          const bool on = k4 < H4;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          for (int b = 0; b < nb; ++b) {
            const float d = __shfl(dpre, b, 64);
            if (on) {
              const f32x4 x = reinterpret_cast<const f32x4*>(hidden + (size_t)(bc + b) * L * H)[k4];
              acc[0] += d * x[0]; acc[1] += d * x[1]; acc[2] += d * x[2]; acc[3] += d * x[3];
            }
          }
          if (on) {
            f32x4* o = reinterpret_cast<f32x4*>(dWp + (size_t)n * H) + k4;
            f32x4 t = *o;
            t[0] += acc[0]; t[1] += acc[1]; t[2] += acc[2]; t[3] += acc[3];
            *o = t;
          }
        }
      } else {
        for (int k0 = 0; k0 < H; k0 += 64) {
          const int k = k0 + lane;
          float s = 0.f;
          for (int b = 0; b < nb; ++b) {
            const float d = __shfl(dpre, b, 64);
            if (k < H) s += d * hidden[(size_t)(bc + b) * L * H + k];
          }
          if (k < H) dWp[(size_t)n * H + k] += s;
        }
      }
    }
    if (lane == 0) dbp[n] += sb;
    return;
  }
  blk -= nW;
  if (blk < nX) {
    // the sample's 'dpre' row once per workgroup through LDS (every thread one or a few n), then sixteen waves split the sum over n:
    // the loop is a stream of Wp loads with nothing to wait for in between
    const int kb = (H + 63) / 64;
    const int b = blk / kb;
    const int k = (blk - b * kb) * 64 + lane;
    for (int n = threadIdx.x; n < H; n += 64 * PHB_WAVES) dpre_s[n] = phb_dpre(dlogits, Wl, pooled, b, n, H, Cn);
    __syncthreads();
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (k < H) {
      constexpr int S = PHB_WAVES;
      int n = wv;
      for (; n + 3 * S < H; n += 4 * S) {        // (the summation order of pooler_bwd_x_kernel)
        s0 += dpre_s[n] * Wp[(size_t)n * H + k];
        s1 += dpre_s[n + S] * Wp[(size_t)(n + S) * H + k];
        s2 += dpre_s[n + 2 * S] * Wp[(size_t)(n + 2 * S) * H + k];
        s3 += dpre_s[n + 3 * S] * Wp[(size_t)(n + 3 * S) * H + k];
      }
      for (; n < H; n += S) s0 += dpre_s[n] * Wp[(size_t)n * H + k];
    }
    red[wv][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wv == 0 && k < H) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < PHB_WAVES; ++w) s += red[w][lane];
      float* d = dhidden + (size_t)b * L * H + k;
      *d = beta ? *d + s : s;
    }
    return;
  }
  blk -= nX;
  const int k = blk * 64 * PHB_WAVES + threadIdx.x;
  if (k < Cn) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dlogits[(size_t)b * Cn + k];
    dbl[k] += s;
  }
  if (k >= H) return;
  for (int c = 0; c < Cn; ++c) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dlogits[(size_t)b * Cn + c] * pooled[(size_t)b * H + k];
    dWl[(size_t)c * H + k] += s;
  }
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

// Pooler + classifier, one launch: pooled = tanh(hidden[:, 0] Wp^T + bp) (model/layer.py:179-185), logits = pooled Wl^T + bl
// (model/meme_uniter.py:19-21).  `ticket`: UNITER_POOL_HEAD_TICKET_WORDS zero-initialised unsigneds the launch leaves zero again (its last
// workgroup computes the logits).  Results equal uniter_pooler_fwd followed by uniter_linear_small_fwd.
extern "C" int uniter_pool_head_fwd(const float* hidden, const float* Wp, const float* bp, const float* Wl, const float* bl,
                                    float* pooled, float* logits, unsigned* ticket, int B, int L, int H, int Cn, void* stream) {
  UCHECK_ARG(hidden && Wp && bp && Wl && bl && pooled && logits && ticket && B > 0 && L > 0 && H > 0 && Cn > 0,
             "pool_head_fwd: bad argument");
  hipLaunchKernelGGL(pool_head_fwd_kernel, dim3((H + 3) / 4, (B + 3) / 4), dim3(256), 0, (hipStream_t)stream, hidden, Wp, bp, Wl,
                     bl, pooled, logits, ticket, B, L, H, Cn);
  UCHECK_LAUNCH();
  return 0;
}

// Backward of uniter_pool_head_fwd from the logits' gradient, one launch: dWp, dbp, dWl, dbl are ACCUMULATED into, dhidden (optional) gets
// the pooler's input gradient in the first row of every sample (assigned, or added with beta_dhidden != 0; the other rows are not touched).
// Results equal uniter_linear_small_bwd followed by uniter_pooler_bwd.
extern "C" int uniter_pool_head_bwd(const float* dlogits, const float* pooled, const float* hidden, const float* Wp,
                                    const float* Wl, float* dWp, float* dbp, float* dWl, float* dbl, float* dhidden, int B, int L,
                                    int H, int Cn, int beta_dhidden, void* stream) {
  UCHECK_ARG(dlogits && pooled && hidden && Wp && Wl && dWp && dbp && dWl && dbl && B > 0 && L > 0 && H > 0 && Cn > 0 && Cn <= H &&
                 H <= PHB_MAX_H, "pool_head_bwd: bad argument (hidden size up to 4096)");
  const int nW = (H + 3) / 4;
  const int nX = dhidden ? ((H + 63) / 64) * B : 0;
  const int nL = (H + 64 * PHB_WAVES - 1) / (64 * PHB_WAVES);
  hipLaunchKernelGGL(pool_head_bwd_kernel, dim3(nW + nX + nL), dim3(64 * PHB_WAVES), 0, (hipStream_t)stream, dlogits, pooled, hidden,
                     Wp, Wl, dWp, dbp, dWl, dbl, dhidden, B, L, H, Cn, nW, nX, beta_dhidden);
  UCHECK_LAUNCH();
  return 0;
}
