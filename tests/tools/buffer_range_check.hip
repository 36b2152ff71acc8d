// Does the raw-buffer range check on gfx950 include the scalar offset (soffset)?  The GEMM epilogue
// and the k-major operand loads rely on it (rows addressed as voffset + soffset, rows >= M dropped).
// Prints what a load / store at voffset = 0, soffset = 128 does against a 64-byte descriptor.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(float* buf, float* out) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 64, 0x00020000);
  const int v = threadIdx.x * 4;
  unsigned a = __builtin_amdgcn_raw_buffer_load_b32(r, v, 128, 0);          // in range by voffset, out by voffset+soffset
  unsigned b = __builtin_amdgcn_raw_buffer_load_b32(r, v + 128, 0, 0);      // out of range by voffset alone
  unsigned c = __builtin_amdgcn_raw_buffer_load_b32(r, v, 32, 0);           // in range either way (v < 32)
  __builtin_amdgcn_raw_buffer_store_b32(0x42280000u /* 42.0f */, r, v, 128, 0);
  out[threadIdx.x * 3 + 0] = __builtin_bit_cast(float, a);
  out[threadIdx.x * 3 + 1] = __builtin_bit_cast(float, b);
  out[threadIdx.x * 3 + 2] = __builtin_bit_cast(float, c);
}
int main() {
  float h[64]; for (int i = 0; i < 64; ++i) h[i] = 100.f + i;
  float *d, *o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 8 * 3 * 4);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(8), 0, 0, d, o);
  float r[24], back[64];
  hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost); hipMemcpy(back, d, sizeof(back), hipMemcpyDeviceToHost);
  printf("load voff=0 soff=128 (element 32 = %g if NOT range-checked): %g\n", h[32], r[0]);
  printf("load voff=128 soff=0: %g   load voff=0 soff=32 (expect %g): %g\n", r[1], h[8], r[2]);
  printf("store voff=0 soff=128: element 32 is now %g (42 = store went through)\n", back[32]);
  return 0;
}
