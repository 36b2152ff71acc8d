// Diagnostic (not part of the product): what does each non-MFMA instruction of the fp32 GEMM's k-loop cost beside
// v_mfma_f32_32x32x2_f32?  Geometry of gemm_f32_v3_kernel: 256-thread workgroups, four resident per CU (one wave of each
// per SIMD), ONE 32 x 32 accumulator per wave (a dependent MFMA chain), 16 MFMAs per 32-deep unit.  Each variant adds
// one ingredient of the real loop to the bare chain; the last one adds them all.  Output: TFLOP/s and the fraction of
// the 157.3 TFLOP/s peak, i.e. the price of that ingredient.
//   hipcc -O3 --offload-arch=gfx950 tests/tools/mfma_f32_pricelist.hip -o tests/tools/mfma_f32_pricelist.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

enum { BARE = 0, RD8 = 1, WR4 = 2, LD4 = 4, BAR = 8, RD4 = 16, WAIT = 32, TWOACC = 64 };

template <int F>
__global__ __launch_bounds__(256, 2) void probe(const float* __restrict__ src, float* out, int iters, int ld) {
  __shared__ __attribute__((aligned(16))) float s[2 * 128 * 36];       // 36 KB: four workgroups per CU
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5, wave = tid >> 6;
  for (int k = tid; k < 2 * 128 * 36; k += 256) s[k] = (float)(k % 7) * 0.25f;
  __syncthreads();
  f32x16 acc = {0}, acc2 = {0};
  f32x4 fa = {0.5f, 0.25f, 1.f, 2.f}, fb = {1.f, 0.5f, 0.25f, 2.f};
  f32x4 st[4];
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1 << 28, 0x00020000);
  int voff[4];
  for (int p = 0; p < 4; ++p) voff[p] = ((((blockIdx.x & 15) * 128 + 64 * (p >> 1) + (tid >> 3) + 32 * (p & 1)) * ld) + (tid & 7) * 4) * 4;
  for (int p = 0; p < 4; ++p) st[p] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* sA = s + (wave >> 1) * 32 * 36;
  const float* sB = s + 64 * 36 + (wave & 1) * 32 * 36;
  for (int it = 0; it < iters; ++it) {
    const int stage = (it & 1) * 128 * 36;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      if (F & (RD8 | RD4)) {
        if (!(F & RD4) || !(kb & 1)) {
          fa = *reinterpret_cast<const f32x4*>(sA + stage + i * 36 + kb * 8 + 4 * h);
          fb = *reinterpret_cast<const f32x4*>(sB + stage + i * 36 + kb * 8 + 4 * h);
        }
      }
      if ((F & WR4) && kb == 2) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
          *reinterpret_cast<f32x4*>(s + (128 * 36 - stage) + ((tid >> 3) + 32 * p) * 36 + (tid & 7) * 4) = st[p];
      }
      if ((F & LD4) && kb == 2) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
          st[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[p], (it & 63) * 128, 0));
      }
      if ((F & BAR) && kb == 3) __syncthreads();
      if ((F & WAIT) && kb == 3) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if ((F & TWOACC) && (t & 1)) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t], fb[t], acc2, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t], fb[t], acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float r = 0;
  for (int k = 0; k < 16; ++k) r += acc[k] + acc2[k];
  for (int p = 0; p < 4; ++p) r += st[p][0];
  out[blockIdx.x * 256 + tid] = r;
}

template <int F>
void run(const char* name, const float* src, float* out) {
  const int nb = 1024, iters = 4000, ld = 768;
  hipLaunchKernelGGL(probe<F>, dim3(nb), dim3(256), 0, 0, src, out, 400, ld);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<F>, dim3(nb), dim3(256), 0, 0, src, out, iters, ld);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)nb * 4 * iters * 16 * 4096.0;
  printf("%-58s %6.1f TFLOP/s  %.3f of peak\n", name, flop / ms / 1e9, flop / ms / 1e9 / 157.3);
}

int main() {
  float *src, *out;
  hipMalloc(&src, 1 << 28); hipMemset(src, 0, 1 << 28);
  hipMalloc(&out, 1024 * 256 * 4);
  run<BARE>("bare chain: 16 MFMA / unit, one accumulator", src, out);
  run<TWOACC>("two accumulators", src, out);
  run<RD4>("+ 4 ds_read_b128 / unit", src, out);
  run<RD8>("+ 8 ds_read_b128 / unit (the kernel's)", src, out);
  run<WR4>("+ 4 ds_write_b128 / unit", src, out);
  run<LD4>("+ 4 buffer_load_dwordx4 / unit (L2 hits)", src, out);
  run<BAR>("+ 1 s_barrier / unit", src, out);
  run<RD8 | BAR>("+ reads + barrier", src, out);
  run<RD8 | WR4 | BAR>("+ reads + writes + barrier", src, out);
  run<RD8 | WR4 | LD4 | BAR>("+ reads + writes + loads + barrier (the whole loop)", src, out);
  run<RD8 | WR4 | LD4 | BAR | TWOACC>("the whole loop, two accumulators", src, out);
  return 0;
}
