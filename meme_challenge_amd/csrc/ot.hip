// Optimal-transport distance of the ITM pretraining task (IPOT): replaces optimal_transport_dist / cost_matrix_cosine / ipot of
// the reference's model/ot.py (:11-21, :36-66, :69-85; called from model/pretrain.py:168-193 on the text and region rows of
// the encoder output).  One workgroup per sample; the cosine cost matrix C [M][N], A = exp(-C / beta) and the transport plan
// T [N][M] live in LDS through all `iteration` proximal steps (M text rows x N region rows: 3 M N + 2 (M + N) floats, 55 KB at
// 128 x 36), fp32 like the reference ("run in fp32 for stability", pretrain.py:188).  The distance's gradient reaches the
// embeddings through C only (the reference detaches T): dC = g T^T, then through the row normalisation of F.normalize.
// Not on the fine-tuning path: a correct, LDS-resident form, not a tuned one.
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int DC = 32;                 // embedding columns staged per step of the cost product
constexpr int MAX_MN = 12288;          // cost-matrix entries a workgroup holds (48 accumulators per thread)

struct OtArgs {
  const float* x; const float* y;              // [B, M, D], [B, N, D]
  const unsigned char* x_pad; const unsigned char* y_pad;      // [B, M], [B, N]: 1 = padding
  float* dist;                                 // [B]
  float* T;                                    // [B, N, M] transport plan (forward: output, optional; backward: input)
  const float* g;                              // [B] gradient of the distances (backward)
  float* dx; float* dy;                        // backward outputs
  int B, M, N, D, iteration;
  float beta, eps;
};

extern __shared__ __attribute__((aligned(16))) float smem[];

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// row norms (clamped at eps as F.normalize does) of the sample's x and y rows -> nx[M], ny[N]
__device__ void row_norms(const OtArgs& a, int b, float* nx, float* ny) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < a.M + a.N; r += NT / 64) {
    const float* row = r < a.M ? a.x + ((size_t)b * a.M + r) * a.D : a.y + ((size_t)b * a.N + (r - a.M)) * a.D;
    float s = 0.f;
    for (int d = lane; d < a.D; d += 64) s += row[d] * row[d];
    s = wave_sum(s);
    if (lane == 0) (r < a.M ? nx[r] : ny[r - a.M]) = fmaxf(sqrtf(s), a.eps);
  }
}

// LDS: C [M][N] | A [N][M] | T [N][M] | sigma [M] | delta [N] | nx [M] | ny [N] | xs [M][DC + 1] | ys [N][DC + 1] | red [NT / 64]
__global__ __launch_bounds__(NT) void ot_fwd_kernel(const OtArgs a) {
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int M = a.M, N = a.N, MN = M * N;
  float* C = smem;
  float* A = C + MN;
  float* T = A + MN;
  float* sigma = T + MN;
  float* delta = sigma + M;
  float* nx = delta + N;
  float* ny = nx + M;
  float* xs = ny + N;
  float* ys = xs + M * (DC + 1);
  float* red = ys + N * (DC + 1);
  const unsigned char* xp = a.x_pad + (size_t)b * M;
  const unsigned char* yp = a.y_pad + (size_t)b * N;
  row_norms(a, b, nx, ny);
  // x . y^T in DC-column steps through LDS; entry e = m * N + n of this thread in acc[e / NT]
  float acc[MAX_MN / NT];
#pragma unroll
  for (int j = 0; j < MAX_MN / NT; ++j) acc[j] = 0.f;
  for (int d0 = 0; d0 < a.D; d0 += DC) {
    __syncthreads();
    for (int e = tid; e < (M + N) * DC; e += NT) {
      const int r = e / DC, c = e - r * DC;
      const float v = d0 + c < a.D ? (r < M ? a.x[((size_t)b * M + r) * a.D + d0 + c] : a.y[((size_t)b * N + (r - M)) * a.D + d0 + c]) : 0.f;
      (r < M ? xs[r * (DC + 1) + c] : ys[(r - M) * (DC + 1) + c]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MAX_MN / NT; ++j) {
      const int e = tid + j * NT;
      if (e < MN) {
        const int m = e / N, n = e - m * N;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < DC; ++c) s += xs[m * (DC + 1) + c] * ys[n * (DC + 1) + c];
        acc[j] += s;
      }
    }
  }
  __syncthreads();
  int xlen = 0, ylen = 0;
  for (int m = 0; m < M; ++m) xlen += xp[m] ? 0 : 1;
  for (int n = 0; n < N; ++n) ylen += yp[n] ? 0 : 1;
  const float xl = (float)xlen, yl = (float)ylen;
#pragma unroll
  for (int j = 0; j < MAX_MN / NT; ++j) {
    const int e = tid + j * NT;
    if (e < MN) {
      const int m = e / N, n = e - m * N;
      const bool pad = xp[m] || yp[n];
      const float c = pad ? 0.f : 1.f - acc[j] / (nx[m] * ny[n]);        // ot.py:19-20, :74
      C[e] = c;
      A[n * M + m] = pad ? 0.f : __expf(-c / a.beta);                    // ot.py:43, :49
      T[n * M + m] = pad ? 0.f : 1.f;                                    // ot.py:42, :48
    }
  }
  for (int m = tid; m < M; m += NT) sigma[m] = xp[m] ? 0.f : 1.f / xl;    // ot.py:40, :46
  __syncthreads();
  for (int it = 0; it < a.iteration; ++it) {
    for (int e = tid; e < MN; e += NT) T[e] *= A[e];                      // Q = A * T (ot.py:60), in place
    __syncthreads();
    for (int n = wave; n < N; n += NT / 64) {                             // delta (ot.py:62)
      float s = 0.f;
      for (int m = lane; m < M; m += 64) s += T[n * M + m] * sigma[m];
      s = wave_sum(s);
      if (lane == 0) delta[n] = 1.f / (yl * s + (yp[n] ? 1e4f : 0.f));
    }
    __syncthreads();
    for (int m = tid; m < M; m += NT) {                                   // sigma (ot.py:63)
      float s = 0.f;
      for (int n = 0; n < N; ++n) s += delta[n] * T[n * M + m];
      sigma[m] = 1.f / (xl * s + (xp[m] ? 1e4f : 0.f));
    }
    __syncthreads();
    for (int e = tid; e < MN; e += NT) {                                  // T = delta * Q * sigma (ot.py:64)
      const int n = e / M, m = e - n * M;
      T[e] = delta[n] * T[e] * sigma[m];
    }
    __syncthreads();
  }
  // padded entries back to zero (ot.py:65), distance = trace(C T) = sum_mn C[m][n] T[n][m] (ot.py:84)
  float s = 0.f;
  for (int e = tid; e < MN; e += NT) {
    const int n = e / M, m = e - n * M;
    const float t = (xp[m] || yp[n]) ? 0.f : T[e];
    if (a.T) a.T[(size_t)b * MN + e] = t;
    s += C[m * N + n] * t;
  }
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < NT / 64; ++w) t += red[w];
    a.dist[b] = t;
  }
}

// dx, dy of sum_b g[b] dist[b]: dC[m][n] = g T[n][m]; C = 1 - xn . yn; xn = x / max(|x|, eps)
// LDS: T [N][M] | nx [M] | ny [N]
__global__ __launch_bounds__(NT) void ot_bwd_kernel(const OtArgs a) {
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int M = a.M, N = a.N, MN = M * N;
  float* T = smem;
  float* nx = T + MN;
  float* ny = nx + M;
  for (int e = tid; e < MN; e += NT) T[e] = a.T[(size_t)b * MN + e];
  row_norms(a, b, nx, ny);
  __syncthreads();
  const float g = a.g[b];
  // one wave per output row; lane owns columns d = lane, lane + 64, ..
  for (int r = wave; r < M + N; r += NT / 64) {
    const bool isx = r < M;
    const int i = isx ? r : r - M, K = isx ? N : M;
    const float* own = isx ? a.x + ((size_t)b * M + i) * a.D : a.y + ((size_t)b * N + i) * a.D;
    const float* oth = isx ? a.y + (size_t)b * N * a.D : a.x + (size_t)b * M * a.D;
    const float nown = isx ? nx[i] : ny[i];
    float* out = isx ? a.dx + ((size_t)b * M + i) * a.D : a.dy + ((size_t)b * N + i) * a.D;
    for (int d0 = 0; d0 < a.D; d0 += 64 * 4) {
      float dn[4] = {0.f, 0.f, 0.f, 0.f};                 // d(normalised own row) at d0 + lane + 64 j
      for (int k = 0; k < K; ++k) {
        const float t = isx ? T[k * M + i] : T[i * M + k];
        if (t == 0.f) continue;                            // wave-uniform (padding, or an empty plan entry)
        const float w = -g * t / (isx ? ny[k] : nx[k]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int d = d0 + lane + 64 * j;
          if (d < a.D) dn[j] += w * oth[(size_t)k * a.D + d];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int d = d0 + lane + 64 * j;
        if (d < a.D) out[d] = dn[j];                        // first pass: d(normalised row); finished below
      }
    }
    // through x / max(|x|, eps): (dn - xn (xn . dn)) / |x| above the clamp, dn / eps at it
    float dot = 0.f, sq = 0.f;
    for (int d = lane; d < a.D; d += 64) { dot += own[d] * out[d]; sq += own[d] * own[d]; }
    dot = wave_sum(dot); sq = wave_sum(sq);
    const bool clamped = sqrtf(sq) < a.eps;
    for (int d = lane; d < a.D; d += 64) {
      const float dn = out[d];
      out[d] = clamped ? dn / a.eps : (dn - own[d] * (dot / (nown * nown))) / nown;
    }
  }
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
  UCHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return 0;
}

int check(const char* who, int B, int M, int N, int D, int iteration, float beta) {
  UCHECK_ARG(B > 0 && M > 0 && N > 0 && D > 0 && iteration >= 0 && beta > 0.f, "%s: bad dims B=%d M=%d N=%d D=%d / iteration / beta", who, B, M, N, D);
  UCHECK_SHAPE((long)M * N <= MAX_MN, "%s: M x N = %d x %d > %d cost-matrix entries per sample (the LDS-resident form)", who, M, N, MAX_MN);
  return 0;
}

}  // namespace

extern "C" int uniter_ot_dist_fwd(const float* txt_emb, const float* img_emb, const unsigned char* txt_pad,
                                  const unsigned char* img_pad, float* dist, float* T, int B, int M, int N, int D, float beta,
                                  int iteration, void* stream) {
  UCHECK_ARG(txt_emb && img_emb && txt_pad && img_pad && dist, "ot_dist_fwd: null pointer");
  UCHECK_RC(check("ot_dist_fwd", B, M, N, D, iteration, beta));
  OtArgs a = {};
  a.x = txt_emb; a.y = img_emb; a.x_pad = txt_pad; a.y_pad = img_pad; a.dist = dist; a.T = T;
  a.B = B; a.M = M; a.N = N; a.D = D; a.iteration = iteration; a.beta = beta; a.eps = 1e-5f;      // eps: ot.py:11
  const size_t lds = ((size_t)3 * M * N + 2 * (M + N) + (size_t)(M + N) * (DC + 1) + NT / 64) * sizeof(float);
  UCHECK_SHAPE(lds <= 160 * 1024, "ot_dist_fwd: M = %d, N = %d need %zu bytes of LDS (160 KB per workgroup)", M, N, lds);
  UCHECK_RC(set_lds(ot_fwd_kernel, lds));
  hipLaunchKernelGGL(ot_fwd_kernel, dim3(B), dim3(NT), lds, (hipStream_t)stream, a);
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_ot_dist_bwd(const float* txt_emb, const float* img_emb, const float* T, const float* grad_dist,
                                  float* d_txt, float* d_img, int B, int M, int N, int D, void* stream) {
  UCHECK_ARG(txt_emb && img_emb && T && grad_dist && d_txt && d_img, "ot_dist_bwd: null pointer");
  UCHECK_RC(check("ot_dist_bwd", B, M, N, D, 0, 1.f));
  OtArgs a = {};
  a.x = txt_emb; a.y = img_emb; a.T = const_cast<float*>(T); a.g = grad_dist; a.dx = d_txt; a.dy = d_img;
  a.B = B; a.M = M; a.N = N; a.D = D; a.eps = 1e-5f;
  const size_t lds = ((size_t)M * N + M + N) * sizeof(float);
  UCHECK_SHAPE(lds <= 160 * 1024, "ot_dist_bwd: M = %d, N = %d need %zu bytes of LDS", M, N, lds);
  UCHECK_RC(set_lds(ot_bwd_kernel, lds));
  hipLaunchKernelGGL(ot_bwd_kernel, dim3(B), dim3(NT), lds, (hipStream_t)stream, a);
  UCHECK_LAUNCH();
  return 0;
}
