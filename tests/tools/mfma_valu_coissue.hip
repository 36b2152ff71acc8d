// Diagnostic (not part of the product): how much vector-ALU work hides behind an fp32 MFMA on gfx950?
// The fp32 matrix instructions run at the fp32 VECTOR rate (64 FLOP/clk/SIMD); DESIGN.md section 4 records that VALU
// work beside them ADDS to the kernel time in the fp32 GEMM.  This probe measures it directly: per loop iteration
// 16 independent v_mfma_f32_32x32x2_f32 (or 32 v_mfma_f32_16x16x4_f32), each followed by V independent VALU
// instructions (v_fma_f32, or v_exp_f32), at 1 / 2 / 4 waves per SIMD.  Output: shader cycles per MFMA
// (s_memtime), so the number of VALU issue cycles that are free per MFMA can be read off the knee.
//   hipcc -O3 --offload-arch=gfx950 tests/tools/mfma_valu_coissue.hip -o /tmp/coissue && /tmp/coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V, int KIND>   // KIND 0: v_fma_f32 fillers, 1: v_exp_f32 fillers, 2: 16x16x4 MFMAs + v_fma fillers
__global__ __launch_bounds__(1024) void probe(float* out, unsigned long long* stamps, int iters) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0}, c4 = {0}, c5 = {0}, c6 = {0}, c7 = {0};
  const int lane = threadIdx.x & 63;
  float x = lane * 0.001f, y = 1.0f - lane * 0.002f;
  float f[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) f[k] = x + k;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#define FILL()                                                                                         \
  _Pragma("unroll") for (int v = 0; v < V; ++v) {                                                      \
    if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(f[v % 16]));                                 \
    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[v % 16]) : "v"(y), "v"(x));                  \
  }                                                                                                    \
  __builtin_amdgcn_sched_barrier(0);
      if (KIND == 2) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c0, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); FILL()
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, c1, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); FILL()
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, c2, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); FILL()
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, c3, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); FILL()
        c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c4, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); FILL()
        c5 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, c5, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); FILL()
        c6 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, c6, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); FILL()
        c7 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, c7, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); FILL()
      } else {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); FILL()
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); FILL()
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); FILL()
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); FILL()
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float acc = 0;
  for (int r = 0; r < 16; ++r) acc += a0[r] + a1[r] + a2[r] + a3[r] + f[r];
  for (int r = 0; r < 4; ++r) acc += c0[r] + c1[r] + c2[r] + c3[r] + c4[r] + c5[r] + c6[r] + c7[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int V, int KIND>
void run(int waves_per_simd) {
  const int nb = 256, nthr = 256 * waves_per_simd, iters = 2000;
  float* out; unsigned long long* st;
  hipMalloc(&out, nb * nthr * 4); hipMalloc(&st, nb * 16 * 8);
  hipLaunchKernelGGL((probe<V, KIND>), dim3(nb), dim3(nthr), 0, 0, out, st, 200);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<V, KIND>), dim3(nb), dim3(nthr), 0, 0, out, st, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[16]; hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
  const int mfma_per_it = KIND == 2 ? 32 : 16;
  // cycles of one wave per MFMA it issued; per SIMD the pipe sees waves_per_simd times as many
  const double cyc = (double)h[0] / ((double)iters * mfma_per_it);
  const double flop = (double)nb * nthr / 64 * iters * mfma_per_it * (KIND == 2 ? 2048.0 : 4096.0);
  printf("%s V=%2d %s fillers, %d waves/SIMD: %6.1f wave-cycles per MFMA = %5.1f SIMD-cycles per MFMA, %6.1f TFLOP/s\n",
         KIND == 2 ? "16x16x4" : "32x32x2", V, KIND == 1 ? "v_exp" : "v_fma", waves_per_simd, cyc, cyc / waves_per_simd,
         flop / ms / 1e9);
  hipFree(out); hipFree(st);
}

int main() {
  for (int w = 1; w <= 4; w *= 2) {
    run<0, 0>(w); run<2, 0>(w); run<4, 0>(w); run<8, 0>(w); run<12, 0>(w); run<16, 0>(w); run<24, 0>(w);
    run<4, 1>(w); run<8, 1>(w);
    run<0, 2>(w); run<2, 2>(w); run<4, 2>(w); run<8, 2>(w);
  }
  return 0;
}
