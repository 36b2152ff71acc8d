// Diagnostic (not part of the product): what bounds the k-loop of the LDS-DMA bf16 GEMM (gemm_bf16_dma.hip) -- and does a larger
// register tile PER WAVE move it?  The loop's three ingredients, in the kernel's geometry and with its swizzled LDS images:
// LDS-DMA fills (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction), ds_read_b128 fragment reads, 32x32x16 bf16 MFMAs.
// A wave owns TM x TN accumulator blocks of 32 x 32: per 16-deep step it reads TM + TN fragments for TM * TN MFMAs.
//   variant            tile        waves  per wave  reads / MFMA  LDS per stage  stages  workgroups per CU
//   the kernel's       128 x 128   4      2 x 2     1.0           32 KB          2       2
//   one wave per SIMD  128 x 128   4      2 x 2     1.0           32 KB          3       1
//   wide wave tile     128 x 256   4      4 x 2     0.75          48 KB          3       1
//   square wave tile   256 x 256   4      4 x 4     0.5           64 KB          2       1
// Every workgroup reads the SAME operand rows (all L2 hits): the figure is the CU-side ceiling, not a GEMM's rate.
//   hipcc -O3 --offload-arch=gfx950 tests/tools/mfma_bf16_tile_lab.hip -o tests/tools/mfma_bf16_tile_lab.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// flags: 1 = no DMA (operands stay as they are), 2 = no LDS reads (fragments stay in registers)
template <int BM, int BN, int TM, int TN, int ST, int WPC, int FLAGS>
__global__ __launch_bounds__(256, WPC) void probe(const unsigned short* __restrict__ src, float* out, int nk, int ld) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int IMG_A = BM * 128, STAGE = (BM + BN) * 128;
  constexpr int NIA = BM / 8 / 4, NIB = BN / 8 / 4, NDMA = NIA + NIB;
  constexpr int WN = BN / (32 * TN);                       // waves along N
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i5 = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(src), 0, 1 << 30, 0x00020000);
  int voA[NIA], voB[NIB];
#pragma unroll
  for (int t = 0; t < NIA; ++t) {
    const int j = wave + 4 * t, row = 8 * j + (lane >> 3), c = (lane & 7) ^ ((4 * (j & 1) + (lane >> 4)) & 7);
    voA[t] = row * ld * 2 + c * 16;
  }
#pragma unroll
  for (int t = 0; t < NIB; ++t) {
    const int j = wave + 4 * t, row = 8 * j + (lane >> 3), c = (lane & 7) ^ ((4 * (j & 1) + (lane >> 4)) & 7);
    voB[t] = (BM + row) * ld * 2 + c * 16;
  }
  const int o0 = i5 * 128, o1 = h ^ ((i5 >> 1) & 7);
  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  for (int k = tid; k < ST * STAGE / 4; k += 256) reinterpret_cast<unsigned*>(smem)[k] = 0x3c003c00u + k % 5;
  __syncthreads();
#define ISSUE(KT, STG)                                                                                       \
  if (!(FLAGS & 1) && !(FLAGS & 4)) {                                                                                        \
    _Pragma("unroll") for (int t = 0; t < NIA; ++t)                                                          \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + (STG) * STAGE + (wave + 4 * t) * 1024), 16, voA[t], ((KT) & 31) * 128, 0, 0); \
    _Pragma("unroll") for (int t = 0; t < NIB; ++t)                                                          \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + (STG) * STAGE + IMG_A + (wave + 4 * t) * 1024), 16, voB[t], ((KT) & 31) * 128, 0, 0); \
  }
#define ISSUE0(KT, STG)                                                                                      \
  if (!(FLAGS & 1)) {                                                                                        \
    _Pragma("unroll") for (int t = 0; t < NIA; ++t)                                                          \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + (STG) * STAGE + (wave + 4 * t) * 1024), 16, voA[t], ((KT) & 31) * 128, 0, 0); \
    _Pragma("unroll") for (int t = 0; t < NIB; ++t)                                                          \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + (STG) * STAGE + IMG_A + (wave + 4 * t) * 1024), 16, voB[t], ((KT) & 31) * 128, 0, 0); \
  }
  bf16x8 xa[2][TM], xb[2][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a) { xa[0][a] = bf16x8{1, 1, 1, 1, 1, 1, 1, 1}; xa[1][a] = xa[0][a]; }
#pragma unroll
  for (int b = 0; b < TN; ++b) { xb[0][b] = bf16x8{1, 1, 1, 1, 1, 1, 1, 1}; xb[1][b] = xb[0][b]; }
#define RD(KS, BUF, SA, SB)                                                                                  \
  if (!(FLAGS & 2)) {                                                                                        \
    _Pragma("unroll") for (int a = 0; a < TM; ++a)                                                           \
      xa[BUF][a] = *reinterpret_cast<const bf16x8*>((SA) + o0 + (wm * TM + a) * 32 * 128 + (((2 * (KS)) ^ o1) << 4)); \
    _Pragma("unroll") for (int b = 0; b < TN; ++b)                                                           \
      xb[BUF][b] = *reinterpret_cast<const bf16x8*>((SB) + o0 + (wn * TN + b) * 32 * 128 + (((2 * (KS)) ^ o1) << 4)); \
  }
#define MM(BUF)                                                                                              \
  _Pragma("unroll") for (int a = 0; a < TM; ++a)                                                             \
  _Pragma("unroll") for (int b = 0; b < TN; ++b)                                                             \
    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb[BUF][b], xa[BUF][a], acc[a][b], 0, 0, 0);
// quarter Q of the next k-tile's fills (FLAGS & 4: the fills are issued between the steps instead of in front of them)
#define ISSUE_Q(KT, STG, Q)                                                                                  \
  if ((FLAGS & 4) && !(FLAGS & 1)) {                                                                         \
    _Pragma("unroll") for (int t = 0; t < NIA; ++t) if ((t & 3) == (Q))                                      \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + (STG) * STAGE + (wave + 4 * t) * 1024), 16, voA[t], ((KT) & 31) * 128, 0, 0); \
    _Pragma("unroll") for (int t = 0; t < NIB; ++t) if ((t & 3) == (Q))                                      \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + (STG) * STAGE + IMG_A + (wave + 4 * t) * 1024), 16, voB[t], ((KT) & 31) * 128, 0, 0); \
  }
#define COMPUTE(STG, KTN, STGN)                                                                              \
  {                                                                                                          \
    const unsigned char* sA = smem + (STG) * STAGE;                                                          \
    const unsigned char* sB = sA + IMG_A;                                                                    \
    RD(0, 0, sA, sB)                                                                                         \
    RD(1, 1, sA, sB) MM(0) ISSUE_Q(KTN, STGN, 0)                                                             \
    RD(2, 0, sA, sB) MM(1) ISSUE_Q(KTN, STGN, 1)                                                             \
    RD(3, 1, sA, sB) MM(0) ISSUE_Q(KTN, STGN, 2)                                                             \
    MM(1) ISSUE_Q(KTN, STGN, 3)                                                                              \
  }
  if (ST == 3) {
    ISSUE0(0, 0); ISSUE0(1, 1);
    for (int kt = 0; kt < nk; kt += 3) {
#define STEP3(O, STG)                                                            \
      wait_vm<NDMA>(); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); \
      ISSUE(kt + (O) + 2, ((STG) + 2) % 3) COMPUTE(STG, kt + (O) + 2, ((STG) + 2) % 3) __builtin_amdgcn_sched_barrier(0);
      STEP3(0, 0) STEP3(1, 1) STEP3(2, 2)
    }
  } else {
    ISSUE0(0, 0);
    for (int kt = 0; kt < nk; kt += 2) {
#define STEP2(O, STG)                                                            \
      wait_vm<0>(); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); \
      ISSUE(kt + (O) + 1, 1 - (STG)) COMPUTE(STG, kt + (O) + 1, 1 - (STG)) __builtin_amdgcn_sched_barrier(0);
      STEP2(0, 0) STEP2(1, 1)
    }
  }
  wait_vm<0>();
  float r = 0;
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) r += acc[a][b][q];
  out[blockIdx.x * 256 + tid] = r;
#endif
}

template <int BM, int BN, int TM, int TN, int ST, int WPC, int FLAGS>
void run(const char* name, const unsigned short* src, float* out) {
  const int nb = 256 * WPC, nk = 3000, ld = 2048;
  const size_t lds = (size_t)ST * (BM + BN) * 128;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<BM, BN, TM, TN, ST, WPC, FLAGS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((probe<BM, BN, TM, TN, ST, WPC, FLAGS>), dim3(nb), dim3(256), lds, 0, src, out, 300, ld);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<BM, BN, TM, TN, ST, WPC, FLAGS>), dim3(nb), dim3(256), lds, 0, src, out, nk, ld);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)nb * nk * 2.0 * BM * BN * 64;
  const double us_per_ktile_cu = ms * 1e3 / nk;                 // all workgroups of a CU advance one k-tile
  printf("%-64s %7.1f TFLOP/s  %.3f of 2.5 PF  %.3f us per k-tile of a CU's tiles (%d KB staged)\n", name, flop / ms / 1e9,
         flop / ms / 1e9 / 2500.0, us_per_ktile_cu, WPC * (BM + BN) * 128 / 1024);
}

int main() {
  unsigned short* src; float* out;
  hipMalloc(&src, 1 << 26); hipMemset(src, 0x3c, 1 << 26);
  hipMalloc(&out, 1024 * 256 * 4);
  run<128, 128, 2, 2, 2, 2, 0>("128x128, 2x2 per wave, 2 stages, 2 workgroups per CU (the kernel)", src, out);
  run<128, 128, 2, 2, 2, 2, 1>("  the same without the DMA fills", src, out);
  run<128, 128, 2, 2, 2, 2, 2>("  the same without the fragment reads", src, out);
  run<128, 128, 2, 2, 2, 2, 3>("  the same with neither (MFMA chain + barrier)", src, out);
  run<128, 128, 2, 2, 2, 2, 4>("  the kernel with the fills issued between the four steps", src, out);
  run<128, 64, 2, 1, 2, 3, 0>("128x64, 2x1 per wave, 2 stages, 3 workgroups per CU", src, out);
  run<64, 128, 1, 2, 2, 3, 0>("64x128, 1x2 per wave, 2 stages, 3 workgroups per CU", src, out);
  run<128, 128, 2, 2, 3, 1, 0>("128x128, 2x2 per wave, 3 stages, 1 workgroup per CU", src, out);
  run<128, 256, 4, 2, 3, 1, 0>("128x256, 4x2 per wave, 3 stages, 1 workgroup per CU", src, out);
  run<128, 256, 4, 2, 3, 1, 1>("  the same without the DMA fills", src, out);
  run<128, 256, 4, 2, 3, 1, 2>("  the same without the fragment reads", src, out);
  run<256, 128, 4, 2, 3, 1, 0>("256x128, 4x2 per wave (waves along M), 3 stages, 1 workgroup per CU", src, out);
  run<256, 256, 4, 4, 2, 1, 0>("256x256, 4x4 per wave, 2 stages, 1 workgroup per CU", src, out);
  run<256, 256, 4, 4, 2, 1, 1>("  the same without the DMA fills", src, out);
  return 0;
}
