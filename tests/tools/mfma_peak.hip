// Diagnostic (not part of the product): what fp32 MFMA rate and shader clock does THIS
// device hold?  (a) bare v_mfma_f32_32x32x2_f32 loop, 1 or 2 waves per SIMD;
// (b) the same with LDS fragment reads in the loop.  Clock = d(s_memtime)/d(s_memrealtime)*100MHz.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool LDS>
__global__ __launch_bounds__(256) void probe(float* out, unsigned long long* stamps, int iters) {
  __shared__ __attribute__((aligned(16))) float s[128 * 36];
  for (int k = threadIdx.x; k < 128 * 36; k += 256) s[k] = (float)(k % 7) * 0.25f;
  __syncthreads();
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  const int lane = threadIdx.x & 63;
  float x = lane * 0.001f, y = 1.0f - lane * 0.002f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    f32x4 fa = {x, y, x, y}, fb = {y, x, y, x};
    if (LDS) {
      fa = *reinterpret_cast<const f32x4*>(s + (lane & 31) * 36 + ((it & 3) * 8) + 4 * (lane >> 5));
      fb = *reinterpret_cast<const f32x4*>(s + (64 + (lane & 31)) * 36 + ((it & 3) * 8) + 4 * (lane >> 5));
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t], fb[t], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t], fb[3 - t], a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[t], fa[t], a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[3 - t], fa[t], a3, 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float acc = 0;
  for (int r = 0; r < 16; ++r) acc += a0[r] + a1[r] + a2[r] + a3[r];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <bool LDS>
void run(int blocks_per_cu, const char* name) {
  const int nb = 256 * blocks_per_cu, iters = 20000;
  float* out; unsigned long long* st;
  hipMalloc(&out, nb * 256 * 4); hipMalloc(&st, nb * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<LDS>, dim3(nb), dim3(256), 0, 0, out, st, 1000);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<LDS>, dim3(nb), dim3(256), 0, 0, out, st, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
  double flops = (double)nb * 4 * iters * 16.0 * 32 * 32 * 2 * 2;
  printf("%-28s %d blk/CU: %.1f TFLOP/s, %.3f ms, in-kernel clock %.0f MHz\n", name, blocks_per_cu,
         flops / ms / 1e9, ms, (double)h[0] / (double)h[1] * 100.0);
  hipFree(out); hipFree(st);
}

int main() {
  run<false>(1, "bare MFMA f32 32x32x2");
  run<false>(2, "bare MFMA f32 32x32x2");
  run<true>(1, "MFMA + ds_read_b128 frags");
  run<true>(2, "MFMA + ds_read_b128 frags");
  return 0;
}
