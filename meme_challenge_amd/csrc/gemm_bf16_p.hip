// bf16-resident GEMM for gfx950, third generation (round 6): the persistent loader / compute form of gemm_split3.hip for
// operands of ONE bf16 piece.
//
// What bounds a bf16 product at the model's sizes is the CU's operand intake (MI355X_MICROARCH.md: ~70 GB/s per CU for any
// 16-byte-per-lane path from L2; gemm_bf16_dma.hip's two co-resident 128 x 128 workgroups already stage their 64 KB per 64-deep
// k-tile at 69 GB/s), so the lever is fewer staged bytes per product: a 128 x 256 tile stages 48 KB per k-tile for the products of
// two 128 x 128 tiles' 64 KB, 128 x 192 (the query|key|value projection: 252 tiles for 256 CUs) 40 KB.  A single-piece operand needs a
// third of the x3 kernel's fragment registers, so eight 64 x 64 compute waves + four loader waves fit 168 registers with BOTH
// 32-deep steps of a k-tile resident.  Structure as gemm_s3p_kernel:
//   * loader waves (LDS-DMA only: buffer_load_dwordx4 ... lds, full 128-byte source lines for k-contiguous operands) run ST - 1
//     k-tiles ahead of the compute waves ACROSS work items of the persistent workgroup; one s_barrier per k-tile between the roles;
//   * compute waves: fragment reads (inline assembly, hand-counted lgkmcnt) + v_mfma_f32_16x16x32_bf16 + epilogue, never a vector
//     memory instruction inside the k-loop; weights as the MFMA's first operand, so a lane owns one output ROW and four
//     consecutive columns per 16 x 16 block: 16-byte fp32 stores, 16-byte bf16 stores after a v_permlane16_swap of two blocks;
//   * XCD-chunked, banded tile walk; split-K as fp32 slabs summed by the consuming row pass.
//
// Replaces cuBLAS behind nn.Linear forward, input-gradient and (grouped per layer) weight-gradient products of
// model/layer.py:76-78 (query / key / value), :112 (attention output), :140 (intermediate) and :153 (output) in the bf16 mode.
//
// LDS images per operand and stage (64-deep k-tiles, as gemm_bf16_dma.hip):
//   k-contiguous ([rows][K]): [R][64] bf16, 128-B rows, 16-B chunk c of row r at c ^ ((r >> 1) & 7) -- conflict-free for the
//     16x16x32 fragment read (lane = row l & 15, k-octet 4 s + (l >> 4) of step s) as well: every ds_read_b128 lane group holds
//     8 rows of one k-octet and the 8 complementary rows of the next, whose chunk numbers cover 0..7 in each row parity.
//   k-major ([K][cols]): 256-B segments, chunk c of k-row k at c ^ (((k & 3) << 2) | ((k >> 2) & 3)), gathered by
//     ds_read_b64_tr_b16; the 64-deep image is two 32-deep images of gemm_split3.hip behind each other.
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <utility>
#include "common.h"
#include "riders.h"

int gemm_chip_cus();      // gemm_split3.hip: CUs of the current device, a multiple of 8

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int KT = 64;                 // k-tile depth: two 32-deep MFMA steps
#define OOB 0x7ffffff0                 /* buffer offset beyond every descriptor: load returns 0, store is dropped */

struct P1Args {
  int M, N, K;
  const void* A; int lda;        // bf16; k-contiguous: rows = M, k-major: rows = K
  const void* B; int ldb;        // bf16; k-contiguous: rows = N, k-major: rows = K
  float* C; int ldc;             // fp32 output (optional); slab s of a split-K launch at C + s * c_split_stride
  long c_split_stride;
  unsigned short* Cb; int ldcb;  // bf16 output (optional)
  const float* bias;
  const void* aux_in; int aux_in_bf16;
  void* aux_out; int aux_out_bf16;
  int ld_aux;
  int tiles_m, tiles_n, band_h, nsplit;
  unsigned long long* stamp;
  int prio;
  float* colpart;     // optional (P1_MUL): partial column sums of the bf16 values the epilogue stores, one row of N floats per 64 output rows
};

struct P1Group {
  P1Args p[4];
  int start[5];
  uniter_x3_riders_t x;     // side work of a grouped weight-gradient launch (kernels instantiated with XTR only)
};

enum { P1_NONE = 0, P1_BIAS = 1, P1_ADD = 4, P1_BIAS_GELU_D = 5, P1_MUL = 6 };

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ void tile_coords_p(int t, int tiles_m, int tiles_n, int band_h, int& tm, int& tn) {
  const int full = band_h * tiles_n;
  const int band = t / full;
  const int rem = t - band * full;
  const int bh = min(band_h, tiles_m - band * band_h);
  tn = rem / bh;
  tm = band * band_h + (rem - tn * bh);
}

__device__ __forceinline__ unsigned pack2p(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v = {(__bf16)lo, (__bf16)hi};      // v_cvt_pk_bf16_f32: round to nearest even
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf_lo_p(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi_p(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// work item of this workgroup, XCD-chunked (blocks b and b + 8 share an XCD's L2)
__device__ __forceinline__ int xcd_work_item_p(int nwork, int round) {
  const int xcd = blockIdx.x & 7, idx = (blockIdx.x >> 3) + round * (int)(gridDim.x >> 3);
  const int q8 = nwork >> 3, r8 = nwork & 7;
  const int chunk0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int chunk_n = q8 + (xcd < r8 ? 1 : 0);
  return idx < chunk_n ? chunk0 + idx : -1;
}

// ---- LDS-DMA fill of one operand image (the layouts of gemm_bf16_dma.hip) ---------------------------------------------
template <int R, bool KM, int NW>
struct DmaP {
  static constexpr int NI = R / 8 / NW;          // 1-KiB wave-instructions per loader wave and k-tile
  static_assert(NI >= 1 && NI * 8 * NW == R, "tile rows must be a multiple of 8 x loader waves");
  static_assert(!KM || R == 128 || R == 256, "k-major tiles are 128 or 256 wide");
  int voff[NI];
  static __device__ __forceinline__ int kstep(int ld) { return (KM ? KT * ld : KT) * 2; }
  __device__ __forceinline__ void offsets(int ld, int rc0, int wave, int lane) {
#pragma unroll
    for (int t = 0; t < NI; ++t) {
      const int j = wave + NW * t;
      if constexpr (!KM) {
        // eight rows x one 128-byte line per instruction: the eight lanes of a row fetch the whole line
        const int row = 8 * j + (lane >> 3);
        const int c = (lane & 7) ^ ((4 * (j & 1) + (lane >> 4)) & 7);
        voff[t] = (rc0 + row) * ld * 2 + c * 16;
      } else if constexpr (R == 128) {
        const int k = 4 * j + (lane >> 4);
        const int c = (lane & 15) ^ (((lane >> 4) << 2) | (j & 3));
        voff[t] = (k * ld + rc0) * 2 + c * 16;
      } else {
        const int k = 2 * j + (lane >> 5);
        const int sw = (((2 * (j & 1) + (lane >> 5)) & 3) << 2) | ((j >> 1) & 3);
        const int c = (lane & 15) ^ sw;
        voff[t] = (k * ld + rc0) * 2 + ((lane >> 4) & 1) * 256 + c * 16;
      }
    }
  }
  template <int T>
  __device__ __forceinline__ void issue1(__amdgpu_buffer_rsrc_t rs, unsigned char* img, int soff, int wave) const {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(img + (wave + NW * T) * 1024), 16, voff[T], soff, 0, 0);
  }
};

template <int OFF>
__device__ __forceinline__ void lds_read_b128_p(u32x4_t& out, unsigned addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(out) : "v"(addr), "n"(OFF) : "memory");
#endif
}
template <int OFF>
__device__ __forceinline__ void lds_read_tr_p(u32x2_t& out, unsigned addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(out) : "v"(addr), "n"(OFF) : "memory");
#endif
}
__device__ __forceinline__ void tie_p(u32x4_t& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(v)::"memory");
#endif
}
__device__ __forceinline__ void tie2_p(u32x2_t& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(v)::"memory");
#endif
}
template <int N> __device__ __forceinline__ void lgkm_wait_p() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
#endif
}
template <int N> __device__ __forceinline__ void wait_vm_p() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---- fragments of v_mfma_f32_16x16x32_bf16: lane l holds row l & 15 of a 16-row block, k = 8 (l >> 4) .. + 7 of a 32-deep step ----
template <int R, bool KM, int NB>
struct FragP {
  unsigned ka[2];           // k-contiguous: offset of step s in the wave's first 16-row block (block t: + t * 2048 B)
  unsigned tr[NB][2];       // k-major: offsets of the two transposed reads of block t (step s: + s * 32 k-rows)
  __device__ __forceinline__ void init(int lane, int blk0) {
    const int r = lane & 15, g = lane >> 4;
    if constexpr (!KM) {
#pragma unroll
      for (int s = 0; s < 2; ++s) ka[s] = (blk0 * 16 + r) * 128 + (((4 * s + g) ^ ((r >> 1) & 7)) << 4);
    } else {
      const int q = r >> 2, p = r & 3;
      const int s1 = (q << 2) | ((2 * g) & 3), s2 = (q << 2) | ((2 * g + 1) & 3);
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        const int cb = blk0 + t;                                // 16-column block of the tile
        const int c = 2 * (cb & 7) + (p >> 1);                  // 16-byte chunk of the block's columns inside its 256-byte segment
        const int seg = (cb >> 3) * 256;
        tr[t][0] = (8 * g + q) * (R * 2) + seg + 8 * (p & 1) + ((c ^ s1) << 4);
        tr[t][1] = (8 * g + 4 + q) * (R * 2) + seg + 8 * (p & 1) + ((c ^ s2) << 4);
      }
    }
  }
};
// the fragments of one 32-deep step
template <int R, bool KM, int NB> struct RegsP;
template <int R, int NB>
struct RegsP<R, false, NB> {
  u32x4_t v[NB];
  static constexpr int READS = NB;
  template <int S>
  __device__ __forceinline__ void read(const FragP<R, false, NB>& f, unsigned img_addr) {
    static_for<0, NB>([&](auto tc) {
      constexpr int T = decltype(tc)::value;
      lds_read_b128_p<T * 2048>(v[T], img_addr + f.ka[S]);
    });
  }
  __device__ __forceinline__ void tie_all() {
#pragma unroll
    for (int t = 0; t < NB; ++t) tie_p(v[t]);
  }
  template <int T> __device__ __forceinline__ bf16x8 get() const { return __builtin_bit_cast(bf16x8, v[T]); }
};
template <int R, int NB>
struct RegsP<R, true, NB> {
  u32x2_t lo[NB], hi[NB];
  static constexpr int READS = 2 * NB;
  template <int S>
  __device__ __forceinline__ void read(const FragP<R, true, NB>& f, unsigned img_addr) {
    static_for<0, NB>([&](auto tc) {
      constexpr int T = decltype(tc)::value;
      lds_read_tr_p<S * 32 * R * 2>(lo[T], img_addr + f.tr[T][0]);
      lds_read_tr_p<S * 32 * R * 2>(hi[T], img_addr + f.tr[T][1]);
    });
  }
  __device__ __forceinline__ void tie_all() {
#pragma unroll
    for (int t = 0; t < NB; ++t) { tie2_p(lo[t]); tie2_p(hi[t]); }
  }
  template <int T> __device__ __forceinline__ bf16x8 get() const {
    return __builtin_bit_cast(bf16x8, u32x4_t{lo[T][0], lo[T][1], hi[T][0], hi[T][1]});
  }
};

// epilogue of one output tile held as 16 x 16 accumulator blocks: lane l owns output row 16 a + (l & 15) and columns
// 16 b + 4 (l >> 4) .. + 3 of block (a, b).  Returns (XTR) the sum of squares of what the lane stored to C.
template <int WM, int WN, int EPI, bool XTR>
__device__ __forceinline__ float p1_epilogue(const P1Args& g, int piece, int m0, int n0, int wm, int wn, int lane,
                                             f32x4 (&acc)[WM / 16][WN / 16]) {
  float ss = 0.f;
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int MB = WM / 16, NB = WN / 16;
  const int r = lane & 15, gq = lane >> 4;
  float* Cp = g.C ? g.C + (size_t)piece * g.c_split_stride : nullptr;
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(Cp, 0, Cp ? g.M * g.ldc * 4 : 0, 0x00020000);
  const bool first = piece == 0;          // the other k-pieces store plain partial sums
  const __amdgpu_buffer_rsrc_t rsCb = __builtin_amdgcn_make_buffer_rsrc(g.Cb, 0, g.Cb ? g.M * g.ldcb * 2 : 0, 0x00020000);
  const int axe_i = g.aux_in_bf16 ? 2 : 4, axe_o = g.aux_out_bf16 ? 2 : 4;
  const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(g.aux_in), 0, g.aux_in ? g.M * g.ld_aux * axe_i : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(g.aux_out, 0, g.aux_out ? g.M * g.ld_aux * axe_o : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsBias = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(g.bias), 0, g.bias ? g.N * 4 : 0, 0x00020000);
  constexpr bool HAS_BIAS = EPI == P1_BIAS || EPI == P1_BIAS_GELU_D;
  constexpr bool HAS_AUX = EPI == P1_ADD || EPI == P1_MUL;
  constexpr bool TWO = EPI == P1_BIAS_GELU_D;
  f32x4 bv[HAS_BIAS ? NB : 1], ax[HAS_AUX ? MB : 1][HAS_AUX ? NB : 1];
  // every load of the tile in front of its first store (loads and stores share vmcnt)
  if (HAS_BIAS && first) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int n = n0 + wn * WN + b * 16 + 4 * gq;
      bv[b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsBias, n < g.N ? n * 4 : OOB, 0, 0));
    }
  }
  if (HAS_AUX && first) {
    // one uniform branch around ALL loads (a per-load branch makes hipcc wait vmcnt(0) after each of them)
    if (g.aux_in_bf16) {
      u32x2_t raw[MB][NB];
#pragma unroll
      for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int m = m0 + wm * WM + a * 16 + r;
          const int n = n0 + wn * WN + b * 16 + 4 * gq;
          raw[a][b] = __builtin_amdgcn_raw_buffer_load_b64(rsI, n < g.N ? (m * g.ld_aux + n) * 2 : OOB, 0, 0);
        }
#pragma unroll
      for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const u32x2_t t = raw[a][b];
          ax[a][b] = f32x4{bf_lo_p(t[0]), bf_hi_p(t[0]), bf_lo_p(t[1]), bf_hi_p(t[1])};
        }
    } else {
#pragma unroll
      for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int m = m0 + wm * WM + a * 16 + r;
          const int n = n0 + wn * WN + b * 16 + 4 * gq;
          ax[a][b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsI, n < g.N ? (m * g.ld_aux + n) * 4 : OOB, 0, 0));
        }
    }
  }
  f32x4 csum[EPI == P1_MUL ? NB : 1];                        // column sums of this wave's 64 rows (g.colpart)
#pragma unroll
  for (int b = 0; b < (EPI == P1_MUL ? NB : 1); ++b) csum[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < MB; ++a) {
    const int m = m0 + wm * WM + a * 16 + r;                 // this lane's output row
    f32x4 x2[NB];                                            // second output (gelu') of the row's blocks
    unsigned pk[NB][2], pk2[TWO ? NB : 1][2];                // the row's bf16 outputs, packed in pairs
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      f32x4& v = acc[a][b];
      if (first) {
        if constexpr (HAS_BIAS) v += bv[b];
        if constexpr (EPI == P1_BIAS_GELU_D) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { float y_, d_; gelu_pair_fast(v[j], y_, d_); v[j] = y_; x2[b][j] = d_; }
        } else if constexpr (EPI == P1_MUL) {
          v *= ax[a][b];
        } else if constexpr (EPI == P1_ADD) {
          v += ax[a][b];
        }
      }
      const int n = n0 + wn * WN + b * 16 + 4 * gq;
      if (Cp) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rsC, n < g.N ? (m * g.ldc + n) * 4 : OOB, 0, 0);
        if constexpr (XTR) {      // only what was stored counts (a k-major operand's overhang columns read the next row, not zeros)
          const float s4 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
          ss += (m < g.M && n < g.N) ? s4 : 0.f;
        }
      }
      if (TWO && !g.aux_out_bf16)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, x2[b]), rsX, n < g.N ? (m * g.ld_aux + n) * 4 : OOB, 0, 0);
      pk[b][0] = pack2p(v[0], v[1]); pk[b][1] = pack2p(v[2], v[3]);
      if constexpr (TWO) { pk2[b][0] = pack2p(x2[b][0], x2[b][1]); pk2[b][1] = pack2p(x2[b][2], x2[b][3]); }
      if constexpr (EPI == P1_MUL) {
        // the bias gradient that belongs to this dY sums the STORED (rounded) values, as a pass over the bf16 tensor would
        if (g.colpart && first) csum[b] += f32x4{bf_lo_p(pk[b][0]), bf_hi_p(pk[b][0]), bf_lo_p(pk[b][1]), bf_hi_p(pk[b][1])};
      }
    }
    // bf16 outputs: two column blocks at a time -- v_permlane16_swap hands the odd 16-lane rows of block b to the even rows and the
    // even rows of block b + 1 to the odd ones, so every lane ends with 8 consecutive columns: one 16-byte store; an odd last block
    // (128 x 192 tiles: three blocks per wave) leaves as 8-byte stores
    const bool two_b16 = TWO && g.aux_out_bf16;
    if (g.Cb || two_b16) {
#pragma unroll
      for (int b = 0; b + 1 < NB; b += 2) {
        const int n8 = n0 + wn * WN + (b + (gq & 1)) * 16 + 8 * (gq >> 1);
        const bool ok = n8 < g.N;
        if (g.Cb) {
          const auto r0 = __builtin_amdgcn_permlane16_swap(pk[b][0], pk[b + 1][0], false, false);
          const auto r1 = __builtin_amdgcn_permlane16_swap(pk[b][1], pk[b + 1][1], false, false);
          const u32x4_t o = {r0[0], r1[0], r0[1], r1[1]};
          __builtin_amdgcn_raw_buffer_store_b128(o, rsCb, ok ? (m * g.ldcb + n8) * 2 : OOB, 0, 0);
        }
        if constexpr (TWO) {
          if (two_b16) {
            const auto r0 = __builtin_amdgcn_permlane16_swap(pk2[b][0], pk2[b + 1][0], false, false);
            const auto r1 = __builtin_amdgcn_permlane16_swap(pk2[b][1], pk2[b + 1][1], false, false);
            const u32x4_t o = {r0[0], r1[0], r0[1], r1[1]};
            __builtin_amdgcn_raw_buffer_store_b128(o, rsX, ok ? (m * g.ld_aux + n8) * 2 : OOB, 0, 0);
          }
        }
      }
      if constexpr (NB % 2 == 1) {
        constexpr int b = NB - 1;
        const int n = n0 + wn * WN + b * 16 + 4 * gq;
        if (g.Cb) __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{pk[b][0], pk[b][1]}, rsCb, n < g.N ? (m * g.ldcb + n) * 2 : OOB, 0, 0);
        if constexpr (TWO) {
          if (two_b16) __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{pk2[b][0], pk2[b][1]}, rsX, n < g.N ? (m * g.ld_aux + n) * 2 : OOB, 0, 0);
        }
      }
    }
  }
  if constexpr (EPI == P1_MUL) if (g.colpart && first) {
    // the 16 lanes of a quarter wave hold the 16 rows of a block: four shuffle steps inside the quarter, then lane r == 0 of each
    // quarter stores its four columns of every block (rows beyond M hold zeros: zero operand rows, zero aux)
    static_assert(WM == 64, "column partials: one row of partial sums per 64 output rows");
    float* prow = g.colpart + (size_t)((m0 / 64) + wm) * g.N;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      f32x4 t = csum[b];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] += __shfl_xor(t[e], o, 64);
      const int n = n0 + wn * WN + b * 16 + 4 * gq;
      if (r == 0 && n < g.N && m0 + wm * 64 < g.M) *reinterpret_cast<f32x4*>(prow + n) = t;
    }
  }
#endif
  return ss;
}

// ---- the kernel ---------------------------------------------------------------------------------------------------------
//   loader:   [issue k-tiles 0 .. ST-2]  for u: { vmcnt: k-tile u landed;  barrier B_u;  issue k-tile u + ST - 1 -> stage (u - 1) % ST }
//   compute:                            for u: { barrier B_u;  read stage u % ST (both 32-deep steps), 2 x MB x NB MFMAs;  (last k-tile of an item: epilogue) }
// B_u orders k-tile u's LDS-DMA before its reads (every loader waited for its own instructions) and the reads of k-tile u - 1
// (each compute wave waits lgkmcnt(0) before its last MFMAs) before the LDS-DMA that overwrites their stage.
template <int BM, int BN, int WM, int WN, bool AKM, bool BKM, int ST, int NWL, int EPI, bool XTR = false>
__global__ __launch_bounds__(64 * ((BM / WM) * (BN / WN) + NWL), ((BM / WM) * (BN / WN) + NWL + 3) / 4)
void gemm_b1p_kernel(const P1Group G) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int WGN = BN / WN, NWC = (BM / WM) * WGN;
  constexpr int IMG_A = BM * KT * 2, IMG_B = BN * KT * 2, STAGE = IMG_A + IMG_B;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[ST * STAGE];
  const int nwork = G.start[4];
  if (!XTR && xcd_work_item_p(nwork, 0) < 0) return;        // (with riders every workgroup stays: it owns sum-of-squares slots)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  stamp_begin(G.p[0].stamp);

  struct Item { int p, piece, m0, n0, kb, ke; bool valid; };
  auto item = [&](int round) -> Item {
    Item it;
    const int w = xcd_work_item_p(nwork, round);
    it.valid = w >= 0;
    if (!it.valid) { it.p = 0; it.piece = it.m0 = it.n0 = it.kb = it.ke = 0; return it; }
    it.p = (w >= G.start[1]) + (w >= G.start[2]) + (w >= G.start[3]);
    const P1Args& g = G.p[it.p];
    const int local = w - G.start[it.p];
    const int tile = local / g.nsplit;
    it.piece = local - tile * g.nsplit;
    int tmi, tni;
    tile_coords_p(tile, g.tiles_m, g.tiles_n, g.band_h, tmi, tni);
    it.m0 = tmi * BM; it.n0 = tni * BN;
    const int nk = (g.K + KT - 1) / KT;
    it.kb = (int)((long)nk * it.piece / g.nsplit);
    it.ke = (int)((long)nk * (it.piece + 1) / g.nsplit);
    return it;
  };

  if (wave >= NWC) {
    // ------------------------------------------------------------------ loader waves ----
    set_wave_prio(2);
    typedef DmaP<BM, AKM, NWL> DA;
    typedef DmaP<BN, BKM, NWL> DB;
    constexpr int NDL = DA::NI + DB::NI;        // LDS-DMA instructions per loader wave and k-tile
    static_assert((ST - 2) * NDL <= 63, "vmcnt is six bits");
    const int lw = wave - NWC;
    DA da;
    DB db;
    int ir = 0;                   // issue cursor: item, k-tile
    Item it = item(0);
    int ikt = it.kb;
    __amdgpu_buffer_rsrc_t rsA, rsB;
    int kstepA = 0, kstepB = 0;
    auto bind = [&]() {
      const P1Args& g = G.p[it.p];
      rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), 0, (AKM ? g.K : g.M) * g.lda * 2, 0x00020000);
      rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.B), 0, (BKM ? g.K : g.N) * g.ldb * 2, 0x00020000);
      da.offsets(g.lda, it.m0, lw, lane);
      db.offsets(g.ldb, it.n0, lw, lane);
      kstepA = DA::kstep(g.lda); kstepB = DB::kstep(g.ldb);
    };
    if (it.valid) bind();
    int istage = 0, issued = 0, consumed = 0;
    auto issue_next = [&]() {
      while (it.valid && ikt >= it.ke) {
        it = item(++ir);
        ikt = it.kb;
        if (it.valid) bind();
      }
      if (!it.valid) return;
      unsigned char* sbase = smem + istage * STAGE;
      static_for<0, DA::NI>([&](auto ic) { da.template issue1<decltype(ic)::value>(rsA, sbase, ikt * kstepA, lw); });
      static_for<0, DB::NI>([&](auto ic) { db.template issue1<decltype(ic)::value>(rsB, sbase + IMG_A, ikt * kstepB, lw); });
      ++ikt; ++issued;
      istage = istage == ST - 1 ? 0 : istage + 1;
    };
#pragma unroll
    for (int s_ = 0; s_ < ST - 1; ++s_) issue_next();
    for (int r = 0;; ++r) {
      const Item c = item(r);
      if (!c.valid) break;
      for (int kt = c.kb; kt < c.ke; ++kt) {
        // k-tile `consumed` has landed once at most the k-tiles issued after it are outstanding
        if (issued - consumed - 1 >= ST - 2) wait_vm_p<(ST - 2) * NDL>(); else wait_vm_p<0>();
        __builtin_amdgcn_s_barrier();
        issue_next();
        ++consumed;
      }
    }
  } else {
    // ------------------------------------------------------------------ compute waves ----
    set_wave_prio(G.p[0].prio);
    constexpr int MB = WM / 16, NB = WN / 16;
    const int wm = wave / WGN, wn = wave % WGN;
    FragP<BM, AKM, MB> fa;
    FragP<BN, BKM, NB> fb;
    fa.init(lane, wm * MB);
    fb.init(lane, wn * NB);
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;
    typedef RegsP<BM, AKM, MB> RA_;
    typedef RegsP<BN, BKM, NB> RB_;
    constexpr int RS = RA_::READS + RB_::READS;        // LDS reads per 32-deep step
    f32x4 acc[MB][NB];
    auto mma = [&](const RA_& pa, const RB_& pb) {
      static_for<0, MB>([&](auto ac) {
        constexpr int A_ = decltype(ac)::value;
        static_for<0, NB>([&](auto bc) {
          constexpr int B_ = decltype(bc)::value;
          acc[A_][B_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pb.template get<B_>(), pa.template get<A_>(), acc[A_][B_], 0, 0, 0);
        });
      });
    };
    // riders (XTR): the sum of squares of everything this wave writes, in double; the column-reduction items of this workgroup --
    // dealt from the END of the grid, where the workgroups with one tile less sit -- run first, while the loader waves fill the
    // first stages of the tile loop
    double wss = 0.0;
    if constexpr (XTR) {
      if (wave < 4)        // (an item is four 16-column strips: the first four compute waves)
        for (int r = (int)gridDim.x - 1 - (int)blockIdx.x; r < G.x.nred; r += (int)gridDim.x)
          wss += (double)riders_reduce_item(G.x, r, wave, lane);
    }
    int stg = 0;
    for (int r = 0;; ++r) {
      const Item c = item(r);
      if (!c.valid) break;
#pragma unroll
      for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int kt = c.kb; kt < c.ke; ++kt) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned sA = lds0 + stg * STAGE, sB = sA + IMG_A;
        RA_ a0, a1;
        RB_ b0, b1;
        a0.template read<0>(fa, sA); b0.template read<0>(fb, sB);
        a1.template read<1>(fa, sA); b1.template read<1>(fb, sB);
        lgkm_wait_p<(RS > 15 ? 15 : RS)>();
        a0.tie_all(); b0.tie_all();
        mma(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        lgkm_wait_p<0>();          // every read of the stage is complete in front of the next barrier (the loaders overwrite it behind it)
        a1.tie_all(); b1.tie_all();
        mma(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        stg = stg == ST - 1 ? 0 : stg + 1;
      }
      const float ss = p1_epilogue<WM, WN, EPI, XTR>(G.p[c.p], c.piece, c.m0, c.n0, wm, wn, lane, acc);
      if constexpr (XTR) wss += (double)ss;
    }
    if constexpr (XTR) riders_store_ssq(G.x, wss, wave, lane, NWC);
  }
  stamp_end(G.p[0].stamp);
#endif
}

// band of the tile walk: an XCD's ~32 concurrent tiles as a sqrt(32 BN / BM)-row rectangle (gemm_split3.hip, plan_tiles3)
void plan_tiles_p(P1Args& g, int BM, int BN) {
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = (g.N + BN - 1) / BN;
  long bh = 1;
  while ((bh + 1) * (bh + 1) * (long)BM <= 32l * BN) ++bh;
  g.band_h = (int)(bh > 16 ? 16 : bh);
  if (g.band_h > g.tiles_m) g.band_h = g.tiles_m;
}

// workgroups of a persistent launch over `nwork` items: a multiple of 8 (one chunk of the work per XCD), one per CU at most
int p1_grid(int nwork, int max_wgs) {
  int grid = (nwork + 7) / 8 * 8;
  const int cus = gemm_chip_cus();
  int cap = max_wgs >= 8 ? max_wgs / 8 * 8 : cus;
  if (g_uniter_cu_reserve > 0) {                          // CUs left to the data-parallel exchange's kernels
    const int room = (cus - g_uniter_cu_reserve) / 8 * 8;
    if (room >= 8 && cap > room) cap = room;
  }
  return grid > cap ? cap : grid;
}

template <int BM, int BN, int WM, int WN, bool AKM, bool BKM, int ST, int EPI, bool XTR = false>
int launch_p1(const P1Group& G, int max_wgs, hipStream_t st) {
  const int grid = p1_grid(G.start[4], max_wgs);
  hipLaunchKernelGGL((gemm_b1p_kernel<BM, BN, WM, WN, AKM, BKM, ST, 4, EPI, XTR>), dim3(grid), dim3(64 * ((BM / WM) * (BN / WN) + 4)), 0, st, G);
  UCHECK_LAUNCH();
  return 0;
}

// geometry: 6 = 128 x 128 tiles (4 compute waves of 64 x 64 + 4 loaders, three 32-KB stages), 7 = 128 x 256 (8 compute waves, three
// 48-KB stages = 144 KB: one workgroup per CU), 8 = 128 x 192 (8 compute waves of 64 x 48, three 40-KB stages; forward layout only)
template <bool BKM, int EPI>
int dispatch_geo(int cfg, const P1Group& G, hipStream_t st) {
  switch (cfg) {
    case 6: return launch_p1<128, 128, 64, 64, false, BKM, 3, EPI>(G, 0, st);
    case 7: return launch_p1<128, 256, 64, 64, false, BKM, 3, EPI>(G, 0, st);
    case 8:
      if constexpr (!BKM) return launch_p1<128, 192, 64, 48, false, BKM, 3, EPI>(G, 0, st);
    default: uniter_set_error("gemm_bf16p: bad cfg %d (6 = 128 x 128, 7 = 128 x 256, 8 = 128 x 192 with k-contiguous weights only)", cfg); return UNITER_E_ARG;
  }
}

}  // namespace

// Tile geometry and k-pieces of a forward / input-gradient product on `avail` CUs: rounds x k-tiles per work item x time per
// 64-deep k-tile (the staged bytes at ~67 GB/s per CU: 32 KB 0.48 us, 40 KB 0.60, 48 KB 0.72) + 3 us per slab (an 8-MB write and
// an 8-MB read in the consuming row pass at configs[1]).  nsplit_fixed > 0: the caller's k-pieces (bf16 outputs cannot be split).
void b1p_choose(int M, int N, int K, int avail, int nsplit_fixed, bool b_kmajor, int* cfg_out, int* ns_out) {
  if (avail < 8) avail = 8;
  const long tm = (M + 127) / 128, t128 = tm * ((N + 127) / 128), t256 = tm * ((N + 255) / 256), t192 = tm * ((N + 191) / 192);
  const int nk = (K + KT - 1) / KT;
  double best = 1e30;
  int bc = 6, bn = nsplit_fixed > 0 ? nsplit_fixed : 1;
  const int ns_lo = nsplit_fixed > 0 ? nsplit_fixed : 1, ns_hi = nsplit_fixed > 0 ? nsplit_fixed : ((N <= 1024 && K >= 512) ? 4 : 1);
  for (int ns = ns_lo; ns <= ns_hi; ++ns) {
    if (nk / ns < 6 && ns > 1) continue;                     // pieces shorter than six k-tiles do not pay for their prologue
    for (int c = 6; c <= 8; ++c) {
      if (c == 7 && N < 256) continue;
      if (c == 8 && (b_kmajor || N % 192 != 0)) continue;
      const long items = (c == 7 ? t256 : c == 8 ? t192 : t128) * ns;
      const long rounds = (items + avail - 1) / avail;
      const double t = (double)rounds * ((nk + ns - 1) / ns) * (c == 7 ? 0.72 : c == 8 ? 0.60 : 0.48) + 3.0 * ns;
      if (t < best - 1e-9) { best = t; bc = c; bn = ns; }
    }
  }
  *cfg_out = bc; *ns_out = bn;
}

// What uniter_gemm_bf16v2_cfg (cfg 0 with the persistent kernels on) and the model's plan choose for a forward / input-gradient
// product of the bf16 mode on `avail_cus` CUs (0 = the chip's): geometry 6 / 7 / 8 and k-pieces.  Host arithmetic, no launch.
extern "C" int uniter_gemm_bf16p_plan(int M, int N, int K, int avail_cus, int nsplit_fixed, int b_kmajor, int* cfg, int* nsplit) {
  UCHECK_ARG(M > 0 && N > 0 && K > 0 && cfg && nsplit && nsplit_fixed >= 0 && nsplit_fixed <= 8, "gemm_bf16p_plan: bad argument");
  b1p_choose(M, N, K, avail_cus > 0 ? avail_cus : p1_grid(1 << 20, 0), nsplit_fixed, b_kmajor != 0, cfg, nsplit);
  return 0;
}

int gemm_b1p_pick_split(int M, int N, int K, int avail) {
  int c, n;
  b1p_choose(M, N, K, avail > 0 ? avail : p1_grid(1 << 20, 0), 0, false, &c, &n);
  return n;
}

// C / Cb = epi(A . B^T) on bf16 operands (fp32 accumulate): the persistent loader / compute kernels.  cfg 6..8 (dispatch_geo), 0 = choose.
int gemm_b1p_run(int cfg, int nsplit, int b_kmajor, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                 float* C, int ldc, long c_split_stride, void* Cb, int ldcb, int epilogue, const float* bias,
                 const void* aux_in, int aux_in_bf16, void* aux_out, int aux_out_bf16, int ld_aux, float* colpart, void* stream) {
  UCHECK_ARG(M > 0 && N > 0 && K > 0 && A && B && (C || Cb), "gemm_bf16p: bad argument");
  UCHECK_ARG(epilogue == UNITER_EPI_NONE || epilogue == UNITER_EPI_BIAS || epilogue == UNITER_EPI_ADD || epilogue == UNITER_EPI_BIAS_GELU_D ||
             epilogue == UNITER_EPI_MUL, "gemm_bf16p: epilogue %d is not built for the persistent kernels (none, bias, + aux, bias + GELU + gelu', x aux)", epilogue);
  UCHECK_ARG(nsplit >= 1 && nsplit <= 8 && (nsplit == 1 || (C && !Cb && c_split_stride >= (long)M * ldc)),
             "gemm_bf16p: split-K needs fp32 slabs (no bf16 output)");
  UCHECK_ARG(!(epilogue == UNITER_EPI_BIAS || epilogue == UNITER_EPI_BIAS_GELU_D) || bias, "gemm_bf16p: epilogue needs bias");
  UCHECK_ARG(!(epilogue == UNITER_EPI_ADD || epilogue == UNITER_EPI_MUL) || aux_in, "gemm_bf16p: epilogue needs aux_in");
  UCHECK_ARG(epilogue != UNITER_EPI_BIAS_GELU_D || aux_out, "gemm_bf16p: epilogue needs aux_out");
  UCHECK_ARG(!colpart || (epilogue == UNITER_EPI_MUL && nsplit == 1 && ((uintptr_t)colpart & 15) == 0 && N % 4 == 0),
             "gemm_bf16p: column partials ride on the x aux epilogue, one k-piece");
  UCHECK_SHAPE(K % KT == 0 && lda % 8 == 0 && ldb % 8 == 0 && N % 8 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 &&
               (ldc % 4 == 0) && (ldcb % 8 == 0) && (ld_aux % 4 == 0) && ((uintptr_t)C & 15) == 0 && ((uintptr_t)Cb & 15) == 0 &&
               ((uintptr_t)aux_in & 15) == 0 && ((uintptr_t)aux_out & 15) == 0 && ((uintptr_t)bias & 15) == 0,
               "gemm_bf16p: K %% 64, N %% 8, leading dimensions %% 8 (bf16) / %% 4 (fp32) and 16-byte aligned buffers required (M=%d N=%d K=%d)", M, N, K);
  UCHECK_SHAPE((size_t)(M + 256) * lda * 2 < (1ull << 31) && (size_t)(b_kmajor ? K + 64 : N + 256) * ldb * 2 < (1ull << 31) &&
               ((size_t)M + 256) * (ldc > 0 ? ldc : 1) * 4 < (1ull << 31) && ((size_t)M + 256) * (ld_aux > 0 ? ld_aux : 1) * 4 < (1ull << 31),
               "gemm_bf16p: operand beyond 31-bit offsets");
  if (cfg == 0) {
    int ns_;
    b1p_choose(M, N, K, p1_grid(1 << 20, 0), nsplit, b_kmajor != 0, &cfg, &ns_);
  }
  P1Group G;
  memset(&G.x, 0, sizeof(G.x));
  P1Args& g = G.p[0];
  g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.c_split_stride = c_split_stride;
  g.Cb = (unsigned short*)Cb; g.ldcb = ldcb; g.bias = bias; g.aux_in = aux_in; g.aux_in_bf16 = aux_in_bf16; g.aux_out = aux_out;
  g.aux_out_bf16 = aux_out_bf16; g.ld_aux = ld_aux; g.nsplit = nsplit; g.colpart = colpart;
  g.stamp = take_stamp_slot();
  g.prio = take_launch_prio();
  plan_tiles_p(g, 128, cfg == 7 ? 256 : cfg == 8 ? 192 : 128);
  const int total = g.tiles_m * g.tiles_n * nsplit;
  for (int p = 1; p < 4; ++p) G.p[p] = g;
  G.start[0] = 0;
  for (int p = 1; p <= 4; ++p) G.start[p] = total;
  hipStream_t st = (hipStream_t)stream;
  if (b_kmajor) {
    switch (epilogue) {
      case UNITER_EPI_NONE: return dispatch_geo<true, P1_NONE>(cfg, G, st);
      case UNITER_EPI_ADD: return dispatch_geo<true, P1_ADD>(cfg, G, st);
      case UNITER_EPI_MUL: return dispatch_geo<true, P1_MUL>(cfg, G, st);
      default: uniter_set_error("gemm_bf16p: epilogue %d is not built for k-major weights (none, + aux, x aux)", epilogue); return UNITER_E_ARG;
    }
  }
  switch (epilogue) {
    case UNITER_EPI_NONE: return dispatch_geo<false, P1_NONE>(cfg, G, st);
    case UNITER_EPI_BIAS: return dispatch_geo<false, P1_BIAS>(cfg, G, st);
    case UNITER_EPI_BIAS_GELU_D: return dispatch_geo<false, P1_BIAS_GELU_D>(cfg, G, st);
    default: uniter_set_error("gemm_bf16p: epilogue %d is not built for k-contiguous weights (none, bias, bias + GELU + gelu')", epilogue); return UNITER_E_ARG;
  }
}

// dW_p[M_p, N_p] (+)= A_p^T B_p for up to four products of one reduction length K (A_p [K][M_p], B_p [K][N_p] bf16, dW_p fp32 with
// leading dimension N_p): ONE persistent launch of whole-K 128 x 256 tiles (216 for an encoder layer of UNITER-base: one round),
// no atomics, bit-reproducible; optional riders (include/uniter_hip.h: uniter_x3_riders_t; 8 sum-of-squares slots per workgroup)
static int b1p_wgrad_tiles(int n, const int* Mo, const int* No) {
  int total = 0;
  for (int p = 0; p < n; ++p) total += ((Mo[p] + 127) / 128) * ((No[p] + 255) / 256);
  return total;
}
int gemm_b1p_wgrad_group_slots(int n, const int* Mo, const int* No, int max_wgs) {
  if (!Mo || !No || n < 1 || n > 4) return 0;
  return 8 * p1_grid(b1p_wgrad_tiles(n, Mo, No), max_wgs);
}
int gemm_b1p_wgrad_group(int n, const int* Mo, const int* No, int K, const void* const* A, const void* const* B, float* const* dW,
                         void* stream, int overwrite, int max_wgs, uniter_x3_riders_t* riders) {
  UCHECK_ARG(n >= 1 && n <= 4 && K > 0 && Mo && No && A && B && dW, "wgrad_bf16p_group: bad argument");
  P1Group G;
  memset(&G.x, 0, sizeof(G.x));
  unsigned long long* stamp = take_stamp_slot();
  int total = 0;
  for (int p = 0; p < 4; ++p) {
    G.start[p] = total;
    if (p >= n) { G.p[p] = G.p[0]; continue; }
    UCHECK_ARG(Mo[p] > 0 && No[p] > 0 && A[p] && B[p] && dW[p], "wgrad_bf16p_group: bad product %d", p);
    UCHECK_SHAPE(Mo[p] % 8 == 0 && No[p] % 8 == 0 && ((uintptr_t)A[p] & 15) == 0 && ((uintptr_t)B[p] & 15) == 0 &&
                 ((uintptr_t)dW[p] & 15) == 0 && (size_t)(K + 64) * (Mo[p] > No[p] ? Mo[p] : No[p]) * 2 < (1ull << 31) &&
                 ((size_t)Mo[p] + 256) * No[p] * 4 < (1ull << 31),
                 "wgrad_bf16p_group: M, N %% 8, 16-byte aligned buffers, 31-bit offsets (product %d: %d x %d, K=%d)", p, Mo[p], No[p], K);
    P1Args& g = G.p[p];
    g.M = Mo[p]; g.N = No[p]; g.K = K; g.A = A[p]; g.lda = Mo[p]; g.B = B[p]; g.ldb = No[p]; g.C = dW[p]; g.ldc = No[p];
    g.c_split_stride = 0; g.Cb = nullptr; g.ldcb = 0; g.bias = nullptr;
    g.aux_in = overwrite ? nullptr : dW[p]; g.aux_in_bf16 = 0; g.aux_out = nullptr; g.aux_out_bf16 = 0; g.ld_aux = No[p];
    g.nsplit = 1; g.stamp = stamp; g.prio = 0; g.colpart = nullptr;
    plan_tiles_p(g, 128, 256);
    total += g.tiles_m * g.tiles_n;
  }
  G.start[4] = total;
  for (int p = n; p < 4; ++p) G.start[p] = total;
  hipStream_t st = (hipStream_t)stream;
  if (riders) {
    UCHECK_ARG(!riders->colsum_out, "wgrad_bf16p_group: colsum_out does not ride on the 128 x 256 geometry (take the bias gradient from the "
               "producing product's column partials as a reduction job)");
    riders->grid = p1_grid(total, max_wgs);
    UCHECK_RC(riders_prepare(*riders, "wgrad_bf16p_group"));
    G.x = *riders;
    return overwrite ? launch_p1<128, 256, 64, 64, true, true, 3, P1_NONE, true>(G, max_wgs, st)
                     : launch_p1<128, 256, 64, 64, true, true, 3, P1_ADD, true>(G, max_wgs, st);
  }
  return overwrite ? launch_p1<128, 256, 64, 64, true, true, 3, P1_NONE>(G, max_wgs, st)
                   : launch_p1<128, 256, 64, 64, true, true, 3, P1_ADD>(G, max_wgs, st);
}
