// fp32-accurate dense products on the bf16 matrix pipe of gfx950: "x3" operands.
//
// gfx950's fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at the fp32 VECTOR rate, 1/16 of v_mfma_f32_32x32x16_bf16.  An fp32
// value is EXACTLY the sum of three bf16 pieces x = x1 + x2 + x3 (round-to-nearest residuals: 8 + 8 + 8 significant bits
// and a sign each), so a product a*b is the six bf16 x bf16 products
//     a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1)          (dropped: a2 b3, a3 b2, a3 b3 <= 2^-24 |a b|)
// accumulated in fp32 by the MFMA: six bf16 MFMAs per 32x32x16 block are 6/16 of the fp32 pipe's time for the same block
// (416.7 TFLOP/s of fp32-equivalent work at the bf16 pipe's 2.5 PFLOP/s).  Every bf16 x bf16 product is exact in fp32, so
// the result differs from an fp32 FMA chain only by the dropped 2^-24 terms and the accumulation order; measured against
// float64 on the model's shapes the error is 0.75-0.93 x the native fp32 kernel's (profiles/r04_gemm_x3_lab.txt).
//
// Replaces cuBLAS behind nn.Linear forward, input-gradient and weight-gradient products of model/layer.py:76-78 (query /
// key / value), :112 (attention output), :140 (intermediate) and :153 (output) in the fp32 mode `fp32x3` (precision 3).
//
// x3 form of a [rows][cols] tensor: three bf16 pieces addressed as  piece p of (r, c) = base + r * row_stride + p *
// piece_stride + c  (elements).  Activations are written [rows][3][ld] (row_stride 3 ld, piece_stride ld): ONE buffer
// descriptor covers the tensor and its range check zero-fills whole rows beyond `rows` in all three pieces.  Weights are
// written piece-major [3][rows][ld] by the optimizer (row_stride ld, piece_stride rows * ld: a flat mirror per piece).
//
// Kernel: persistent workgroups of loader waves (LDS-DMA ring as gemm_bf16_dma.hip: buffer_load_dwordx4 ... lds, no staging
// registers) and compute waves (fragment reads + MFMAs + epilogue), 32-deep k-tiles of 3 + 3 piece images per stage, transposed
// accumulator (lane = output row), ALL LDS reads as inline assembly with hand-counted lgkmcnt waits (hipcc drains vmcnt before
// compiler-visible LDS reads while an LDS-DMA is in flight).
// LDS images per piece and stage:
//   k-contiguous operand ([rows][K]): [R][32] bf16, 64-B rows, 16-B chunk c of row r at c ^ ((r >> 2) & 3): the 16 lanes
//     of a ds_read_b128 group cover all 64 banks.
//   k-major operand ([K][cols]): the image of gemm_bf16_dma.hip cut to 32 k-rows: 256-B segments, chunk c of k-row k at
//     c ^ (((k & 3) << 2) | ((k >> 2) & 3)), gathered by ds_read_b64_tr_b16.
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <utility>
#include "common.h"
#include "riders.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct S3Args {
  int M, N, K;
  const void* A; int lda, psa;   // x3: row stride, piece stride (elements); k-contiguous: rows = M, k-major: rows = K
  const void* B; int ldb, psb;   // x3; k-contiguous: rows = N, k-major: rows = K
  float* C; int ldc;             // fp32 output (optional); slab s of a split-K launch at C + s * c_split_stride
  long c_split_stride;
  unsigned short* Cx; int ldcx, pscx;  // x3 output (optional): row stride, piece stride
  const float* bias;
  const float* aux_in;
  float* aux_out;
  int ld_aux;
  int tiles_m, tiles_n, band_h, nsplit;
  unsigned long long* stamp;
  int prio;
  int dbg;
  int b_paired;       // B (the weights) in the PAIRED-ROW layout of the optimizer's mirror (round 6): rows 2 q and 2 q + 1 interleaved in
                      // 64-byte units, so the 32-deep k-tile u of a row pair is ONE 128-byte line at (q * 2 ldb + u * 64) elements
  float* colpart;     // optional (v_mfma_f32_16x16x32_bf16 geometries): partial column sums of the values the epilogue leaves, one row of N
                      // floats per 64 output rows (row 2 (m0 / 128) + wm): the bias gradient that belongs to a dY this product writes
};

#define OOB 0x7ffffff0        /* buffer offset beyond every descriptor: load returns 0, store is dropped */
// cache policy of the 16x16x32 epilogue's output stores (lab switch, -DX3_ST_AUX=2: nt, =16: sc1 write-through, =18: both).  A
// one-round launch of 252 workgroups ends in one store burst (80 MB for FFN-up forward) and the kernel boundary behind it writes the
// L2s' dirty lines back: round 6 measured whether stores that leave the L2 as they are issued shorten that
#ifndef X3_ST_AUX
#define X3_ST_AUX 0
#endif

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ void tile_coords3(int t, int tiles_m, int tiles_n, int band_h, int& tm, int& tn) {
  const int full = band_h * tiles_n;
  const int band = t / full;
  const int rem = t - band * full;
  const int bh = min(band_h, tiles_m - band * band_h);
  tn = rem / bh;
  tm = band * band_h + (rem - tn * bh);
}

__device__ __forceinline__ unsigned pack2r(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v = {(__bf16)lo, (__bf16)hi};      // v_cvt_pk_bf16_f32: round to nearest even
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bflo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bfhi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// the three bf16 pieces of two fp32 values (exact: every residual is representable)
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& w1, unsigned& w2, unsigned& w3) {
  w1 = pack2r(x0, x1);
  float r0 = x0 - bflo(w1), r1 = x1 - bfhi(w1);
  w2 = pack2r(r0, r1);
  r0 -= bflo(w2); r1 -= bfhi(w2);
  w3 = pack2r(r0, r1);
}

// work item of this workgroup, XCD-chunked (blocks b and b + 8 share an XCD's L2)
__device__ __forceinline__ int xcd_work_item3(int nwork, int round = 0) {
  const int xcd = blockIdx.x & 7, idx = (blockIdx.x >> 3) + round * (int)(gridDim.x >> 3);
  const int q8 = nwork >> 3, r8 = nwork & 7;
  const int chunk0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int chunk_n = q8 + (xcd < r8 ? 1 : 0);
  return idx < chunk_n ? chunk0 + idx : -1;
}

// ---- LDS-DMA fill of the three piece images of one operand ---------------------------------------------------------
// chunk swizzle of the k-contiguous image by row group (row >> 2) & 3: the identity for the 32x32x16 fragments (lane = row i5,
// k-half h), the table {0, 2, 3, 1} for the 16x16x32 ones (lane = row l & 15, k-octet l >> 4) -- with either, the 16 lanes of
// every ds_read_b128 group touch all 64 banks once
template <bool M16>
__device__ __forceinline__ int chunk_swz(int rowgroup) { return M16 ? ((0x1320 >> (4 * rowgroup)) & 3) : rowgroup; }

template <int R, bool KM, int NW, int KT, bool M16 = false>
struct Dma3 {
  static constexpr int NP = R * KT / 512;          // 1-KiB wave-instructions per piece image and k-tile
  static_assert(KT == 32, "32-deep k-tiles");
  static_assert(NP >= NW && NP % NW == 0, "piece image must split evenly over the loader waves");
  static_assert(!KM || R == 128 || R == 256, "k-major tiles are 128 or 256 wide");
  static constexpr int NI = NP / NW;
  static constexpr int IMG = R * KT * 2;
  int voff[NI];
  // rs: row stride of the operand (elements)
  static __device__ __forceinline__ int kstep(int rs) { return (KM ? KT * rs : KT) * 2; }
  // paired (k-contiguous operands): rows 2 q and 2 q + 1 of the operand interleaved in 64-byte units -- the 32-deep k-tile u of the
  // pair is ONE 128-byte line at (q * 2 rs + u * 64) elements: the eight lanes of a row pair fetch a whole line instead of two halves
  __device__ __forceinline__ void offsets(int rs, int rc0, int wave, int lane, bool paired = false) {
#pragma unroll
    for (int t = 0; t < NI; ++t) {
      const int j = wave + NW * t;
      if constexpr (!KM) {
        const int row = 16 * j + (lane >> 2), c = (lane & 3) ^ chunk_swz<M16>((lane >> 4) & 3);
        voff[t] = paired ? ((rc0 + row) >> 1) * rs * 4 + ((rc0 + row) & 1) * 64 + c * 16 : (rc0 + row) * rs * 2 + c * 16;
      } else if constexpr (R == 128) {
        const int k = 4 * j + (lane >> 4);
        const int c = (lane & 15) ^ (((lane >> 4) << 2) | (j & 3));
        // paired (a k-major read of the paired-row weight mirror: k = row of the weight): element (k, col) sits at
        // (k >> 1) * 2 rs + (col >> 5) * 64 + (k & 1) * 32 + (col & 31); rc0 is a multiple of 128, a 16-byte chunk never crosses a unit
        voff[t] = paired ? ((k >> 1) * 2 * rs + ((rc0 >> 5) + (c >> 2)) * 64 + (k & 1) * 32 + 8 * (c & 3)) * 2 : (k * rs + rc0) * 2 + c * 16;
      } else {
        const int k = 2 * j + (lane >> 5);
        const int sw = (((2 * (j & 1) + (lane >> 5)) & 3) << 2) | ((j >> 1) & 3);
        const int c = (lane & 15) ^ sw;
        const int cf = ((lane >> 4) & 1) * 16 + c;          // 16-byte chunk of the 512-byte k-row
        voff[t] = paired ? ((k >> 1) * 2 * rs + ((rc0 >> 5) + (cf >> 2)) * 64 + (k & 1) * 32 + 8 * (cf & 3)) * 2
                         : (k * rs + rc0) * 2 + ((lane >> 4) & 1) * 256 + c * 16;
      }
    }
  }
  // piece image P, wave-instruction T of this wave
  template <int P, int T>
  __device__ __forceinline__ void issue1(__amdgpu_buffer_rsrc_t rs, unsigned char* img, int soff, int pstep, int wave) const {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(img + P * IMG + (wave + NW * T) * 1024), 16, voff[T], soff + P * pstep, 0, 0);
  }
};

// ---- MFMA operand fragments: per-lane LDS byte offsets inside one piece image -------------------------------------------
template <int R, bool KM, int NB, int KT>
struct Frag3 {
  unsigned ka[KT / 16];     // k-contiguous: offset of k16-step ks in the wave's first block (block t: + t * 32 rows)
  unsigned tr[NB][2];       // k-major: offsets of the two transposed reads of block t (k16-step ks: + ks * 16 k-rows)
  static constexpr int ROWB = KT * 2;
  __device__ __forceinline__ void init(int i5, int h, int blk0) {
    if constexpr (!KM) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) ka[ks] = (blk0 * 32 + i5) * 64 + ((((2 * ks + h) ^ ((i5 >> 2) & 3))) << 4);
    } else {
      const int l16 = i5 & 15, q = l16 >> 2, p = l16 & 3;
      const int s1 = (q << 2) | (2 * h), s2 = s1 | 1;
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        const int r0 = (blk0 + t) * 32;
        const int seg = r0 >> 7;
        const int c = ((r0 & 127) >> 3) + 2 * (i5 >> 4) + (p >> 1);
        const int base = (8 * h + q) * (R * 2) + 8 * (p & 1) + seg * 256;
        tr[t][0] = base + ((c ^ s1) << 4);
        tr[t][1] = base + 4 * (R * 2) + ((c ^ s2) << 4);
      }
    }
  }
};

template <int OFF>
__device__ __forceinline__ void lds_read_b128_o(u32x4_t& out, unsigned addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(out) : "v"(addr), "n"(OFF) : "memory");
#endif
}
template <int OFF>
__device__ __forceinline__ void lds_read_tr_o(u32x2_t& out, unsigned addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(out) : "v"(addr), "n"(OFF) : "memory");
#endif
}
// ties a fragment's first use to the statements above it (the wait): an empty volatile asm that "rewrites" the register
__device__ __forceinline__ void tie(u32x4_t& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(v)::"memory");
#endif
}
__device__ __forceinline__ void tie2(u32x2_t& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(v)::"memory");
#endif
}
template <int N> __device__ __forceinline__ void lgkm_wait() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
#endif
}
template <int N> __device__ __forceinline__ void wait_vm3() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// One k-contiguous operand's fragments of one k16-step: [piece][block], 8 consecutive k of row (block, i5) per lane half
template <int R, int NB, int KT>
struct FragRegs {
  u32x4_t v[3][NB];
  static constexpr int READS_PER_PIECE = NB;
  // piece P of k16-step KS, every block of the wave
  template <int KS, int P>
  __device__ __forceinline__ void read_piece(const Frag3<R, false, NB, KT>& f, unsigned img_addr) {
    constexpr int IMG = R * KT * 2;
    static_for<0, NB>([&](auto tc) {
      constexpr int T = decltype(tc)::value;
      lds_read_b128_o<P * IMG + T * 32 * KT * 2>(v[P][T], img_addr + f.ka[KS]);
    });
  }
  template <int P>
  __device__ __forceinline__ void tie_piece() {
#pragma unroll
    for (int t = 0; t < NB; ++t) tie(v[P][t]);
  }
  template <int P, int T>
  __device__ __forceinline__ bf16x8 get() const { return __builtin_bit_cast(bf16x8, v[P][T]); }
};

// k-major fragments: the two halves must stay separate registers until the wait (an asm output cannot be half a vector)
template <int R, int NB, int KT>
struct FragRegsKM {
  u32x2_t lo[3][NB], hi[3][NB];
  static constexpr int READS_PER_PIECE = 2 * NB;
  template <int KS, int P>
  __device__ __forceinline__ void read_piece(const Frag3<R, true, NB, KT>& f, unsigned img_addr) {
    constexpr int IMG = R * KT * 2;
    static_for<0, NB>([&](auto tc) {
      constexpr int T = decltype(tc)::value;
      lds_read_tr_o<P * IMG + KS * 16 * R * 2>(lo[P][T], img_addr + f.tr[T][0]);
      lds_read_tr_o<P * IMG + KS * 16 * R * 2>(hi[P][T], img_addr + f.tr[T][1]);
    });
  }
  template <int P>
  __device__ __forceinline__ void tie_piece() {
#pragma unroll
    for (int t = 0; t < NB; ++t) { tie2(lo[P][t]); tie2(hi[P][t]); }
  }
  template <int P, int T>
  __device__ __forceinline__ bf16x8 get() const {
    return __builtin_bit_cast(bf16x8, u32x4_t{lo[P][T][0], lo[P][T][1], hi[P][T][0], hi[P][T][1]});
  }
};

// ---- fragments of v_mfma_f32_16x16x32_bf16: lane l holds row l & 15 of a 16-row block, k = 8 (l >> 4) .. + 7 of the 32-deep k-tile:
// ONE k-step per k-tile, 16 accumulator blocks of 16 x 16 per 64 x 64 wave tile.  Same FLOPs per cycle as the 32x32x16 form, but
// the chip holds a higher clock under it (MI355X_MICROARCH.md, DVFS item 7; measured here: 6-7 % less time per launch)
template <int R, bool KM, int NB>
struct Frag16 {
  unsigned ka;              // k-contiguous: offset in the wave's first 16-row block (block t: + t * 16 rows = t * 1024 B)
  unsigned tr[NB][2];       // k-major: offsets of the two transposed reads of block t
  // k-major, compact form (the lean compute path: 168 registers): tr[t][h] = kb + h * 4 * R * 2 + (xk ^ ((t << 5) | (h << 4))) for
  // a wave whose first block is a multiple of 4 and NB = 4 -- the block index enters the swizzled chunk number on bits the lane's
  // swizzle XORs, so the address is one v_xad_u32 per read instead of eight registers per operand
  unsigned kb, xk;
  __device__ __forceinline__ void init(int lane, int blk0) {
    const int r = lane & 15, g = lane >> 4;
    if constexpr (!KM) {
      ka = (blk0 * 16 + r) * 64 + ((g ^ chunk_swz<true>((r >> 2) & 3)) << 4);
    } else {
      static_assert(!KM || R == 128 || R == 256, "k-major 16x16x32 fragments: 128- or 256-wide tiles");
      const int q = r >> 2, p = r & 3;
      const int s1 = (q << 2) | ((2 * g) & 3), s2 = (q << 2) | ((2 * g + 1) & 3);
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        // a k-row is R * 2 bytes: one (R = 128) or two (R = 256) 256-byte segments of 16 swizzled 16-byte chunks
        const int cb = blk0 + t;                                // 16-column block of the tile
        const int c = 2 * (cb & 7) + (p >> 1);                  // 16-byte chunk of the block's columns inside its segment
        const int seg = (cb >> 3) * 256;
        tr[t][0] = (8 * g + q) * (R * 2) + seg + 8 * (p & 1) + ((c ^ s1) << 4);
        tr[t][1] = (8 * g + 4 + q) * (R * 2) + seg + 8 * (p & 1) + ((c ^ s2) << 4);
      }
      kb = (8 * g + q) * (R * 2) + (blk0 >> 3) * 256 + 8 * (p & 1);
      xk = ((2 * (blk0 & 7) + (p >> 1)) ^ s1) << 4;
    }
  }
};
template <int R, int NB>
struct FragRegs16 {
  u32x4_t v[3][NB];
  static constexpr int READS_PER_PIECE = NB;
  template <int P>
  __device__ __forceinline__ void read_piece(const Frag16<R, false, NB>& f, unsigned img_addr) {
    constexpr int IMG = R * 64;
    static_for<0, NB>([&](auto tc) {
      constexpr int T = decltype(tc)::value;
      lds_read_b128_o<P * IMG + T * 1024>(v[P][T], img_addr + f.ka);
    });
  }
  template <int P>
  __device__ __forceinline__ void tie_piece() {
#pragma unroll
    for (int t = 0; t < NB; ++t) tie(v[P][t]);
  }
  template <int P, int T>
  __device__ __forceinline__ bf16x8 get() const { return __builtin_bit_cast(bf16x8, v[P][T]); }
};
template <int R, int NB>
struct FragRegs16KM {
  u32x2_t lo[3][NB], hi[3][NB];
  static constexpr int READS_PER_PIECE = 2 * NB;
  template <int P>
  __device__ __forceinline__ void read_piece(const Frag16<R, true, NB>& f, unsigned img_addr) {
    constexpr int IMG = R * 64;
    static_for<0, NB>([&](auto tc) {
      constexpr int T = decltype(tc)::value;
      lds_read_tr_o<P * IMG>(lo[P][T], img_addr + f.tr[T][0]);
      lds_read_tr_o<P * IMG>(hi[P][T], img_addr + f.tr[T][1]);
    });
  }
  template <int P>
  __device__ __forceinline__ void tie_piece() {
#pragma unroll
    for (int t = 0; t < NB; ++t) { tie2(lo[P][t]); tie2(hi[P][t]); }
  }
  template <int P, int T>
  __device__ __forceinline__ bf16x8 get() const {
    return __builtin_bit_cast(bf16x8, u32x4_t{lo[P][T][0], lo[P][T][1], hi[P][T][0], hi[P][T][1]});
  }
};

// ONE piece of an operand's fragments (the lean compute path: B one piece at a time in two alternating register sets)
template <int R, bool KM, int NB> struct Piece16;
template <int R, int NB>
struct Piece16<R, false, NB> {
  u32x4_t v[NB];
  static constexpr int READS = NB;
  template <int P>
  __device__ __forceinline__ void read(const Frag16<R, false, NB>& f, unsigned img_addr) {
    static_for<0, NB>([&](auto tc) {
      constexpr int T = decltype(tc)::value;
      lds_read_b128_o<P * R * 64 + T * 1024>(v[T], img_addr + f.ka);
    });
  }
  __device__ __forceinline__ void tie_all() {
#pragma unroll
    for (int t = 0; t < NB; ++t) tie(v[t]);
  }
  template <int T> __device__ __forceinline__ bf16x8 get() const { return __builtin_bit_cast(bf16x8, v[T]); }
};
template <int R, int NB>
struct Piece16<R, true, NB> {
  u32x2_t lo[NB], hi[NB];
  static constexpr int READS = 2 * NB;
  template <int P>
  __device__ __forceinline__ void read(const Frag16<R, true, NB>& f, unsigned img_addr) {
    static_assert(NB == 4, "compact k-major addressing: four 16-column blocks per wave");
    const unsigned base = img_addr + f.kb;
    static_for<0, NB>([&](auto tc) {
      constexpr int T = decltype(tc)::value;
      lds_read_tr_o<P * R * 64>(lo[T], base + (f.xk ^ (unsigned)(T << 5)));
      lds_read_tr_o<P * R * 64 + 4 * R * 2>(hi[T], base + (f.xk ^ (unsigned)((T << 5) | 16)));
    });
  }
  __device__ __forceinline__ void tie_all() {
#pragma unroll
    for (int t = 0; t < NB; ++t) { tie2(lo[t]); tie2(hi[t]); }
  }
  template <int T> __device__ __forceinline__ bf16x8 get() const {
    return __builtin_bit_cast(bf16x8, u32x4_t{lo[T][0], lo[T][1], hi[T][0], hi[T][1]});
  }
};

// epilogue kinds of this kernel
enum { S3_NONE = 0, S3_BIAS = 1, S3_ADD = 4, S3_BIAS_GELU_D = 5, S3_MUL = 6 };

// epilogue of one output tile: lane = output row m, register group gq of block (a, b) = columns nb + 8 gq + 4 h .. + 3
template <int WM, int WN, int EPI>
__device__ __forceinline__ void s3_epilogue(const S3Args& g, int piece, int m0, int n0, int wm, int wn, int i5, int h,
                                            f32x16 (&acc)[WM / 32][WN / 32]) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int AB = WM / 32, BB = WN / 32;
  float* Cp = g.C ? g.C + (size_t)piece * g.c_split_stride : nullptr;
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(Cp, 0, Cp ? g.M * g.ldc * 4 : 0, 0x00020000);
  const bool first = piece == 0;          // the other k-pieces store plain partial sums
  const __amdgpu_buffer_rsrc_t rsCx = __builtin_amdgcn_make_buffer_rsrc(g.Cx, 0, g.Cx ? ((g.M - 1) * g.ldcx + 2 * g.pscx + g.N) * 2 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(g.aux_in), 0, g.aux_in ? g.M * g.ld_aux * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(g.aux_out, 0, g.aux_out ? g.M * g.ld_aux * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsBias = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(g.bias), 0, g.bias ? g.N * 4 : 0, 0x00020000);
  constexpr bool HAS_BIAS = EPI == S3_BIAS || EPI == S3_BIAS_GELU_D;
  constexpr bool HAS_AUX = EPI == S3_ADD || EPI == S3_MUL;
  constexpr bool TWO = EPI == S3_BIAS_GELU_D;
  f32x4 bv[HAS_BIAS ? BB : 1][4], ax[HAS_AUX ? AB : 1][HAS_AUX ? BB : 1][4];
  if (HAS_BIAS && first) {
#pragma unroll
    for (int b = 0; b < BB; ++b)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int n = n0 + wn * WN + b * 32 + 8 * gq + 4 * h;
        bv[b][gq] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsBias, n < g.N ? n * 4 : OOB, 0, 0));
      }
  }
  if (HAS_AUX && first) {
#pragma unroll
    for (int a = 0; a < AB; ++a)
#pragma unroll
      for (int b = 0; b < BB; ++b)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int m = m0 + wm * WM + a * 32 + i5;
          const int n = n0 + wn * WN + b * 32 + 8 * gq + 4 * h;
          ax[a][b][gq] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsI, n < g.N ? (m * g.ld_aux + n) * 4 : OOB, 0, 0));
        }
  }
#pragma unroll
  for (int a = 0; a < AB; ++a) {
    const int m = m0 + wm * WM + a * 32 + i5;                 // this lane's output row
#pragma unroll
    for (int b = 0; b < BB; ++b) {
      const int nb = n0 + wn * WN + b * 32;                   // first column of the block
      f32x16& v = acc[a][b];
      f32x16 x2;     // second output (gelu')
      if (first) {
        if constexpr (HAS_BIAS) {
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) v[rr] += bv[b][rr >> 2][rr & 3];
        }
        if constexpr (EPI == S3_BIAS_GELU_D) {
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) { float y_, d_; gelu_pair_fast(v[rr], y_, d_); v[rr] = y_; x2[rr] = d_; }
        } else if constexpr (EPI == S3_MUL) {
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) v[rr] *= ax[a][b][rr >> 2][rr & 3];
        } else if constexpr (EPI == S3_ADD) {
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) v[rr] += ax[a][b][rr >> 2][rr & 3];
        }
      }
      if (Cp) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int n = nb + 8 * gq + 4 * h;
          const f32x4 o = {v[4 * gq], v[4 * gq + 1], v[4 * gq + 2], v[4 * gq + 3]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rsC, n < g.N ? (m * g.ldc + n) * 4 : OOB, 0, 0);
        }
      }
      if (TWO) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int n = nb + 8 * gq + 4 * h;
          const f32x4 o = {x2[4 * gq], x2[4 * gq + 1], x2[4 * gq + 2], x2[4 * gq + 3]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rsX, n < g.N ? (m * g.ld_aux + n) * 4 : OOB, 0, 0);
        }
      }
      // x3 output: groups (gq, gq + 1) exchanged between the lane halves -> 8 consecutive columns per lane, one 16-byte
      // store per piece
      if (g.Cx) {
#pragma unroll
        for (int gp = 0; gp < 4; gp += 2) {
          const int n8 = nb + 8 * (gp + h);
          const bool ok = n8 < g.N;
          unsigned wa[3][2], wb[3][2];
          split3_pair(v[4 * gp], v[4 * gp + 1], wa[0][0], wa[1][0], wa[2][0]);
          split3_pair(v[4 * gp + 2], v[4 * gp + 3], wa[0][1], wa[1][1], wa[2][1]);
          split3_pair(v[4 * gp + 4], v[4 * gp + 5], wb[0][0], wb[1][0], wb[2][0]);
          split3_pair(v[4 * gp + 6], v[4 * gp + 7], wb[0][1], wb[1][1], wb[2][1]);
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            const auto r0 = __builtin_amdgcn_permlane32_swap(wa[p][0], wb[p][0], false, false);
            const auto r1 = __builtin_amdgcn_permlane32_swap(wa[p][1], wb[p][1], false, false);
            const u32x4_t o = {r0[0], r1[0], r0[1], r1[1]};
            __builtin_amdgcn_raw_buffer_store_b128(o, rsCx, ok ? (m * g.ldcx + p * g.pscx + n8) * 2 : OOB, 0, 0);
          }
        }
      }
    }
  }
#endif
}

// epilogue of one output tile held as 16 x 16 accumulator blocks (v_mfma_f32_16x16x32_bf16, weights as the A operand): lane l
// owns output row 16 a + (l & 15) and columns 16 b + 4 (l >> 4) .. + 3 of block (a, b)
template <int WM, int WN, int EPI>
__device__ __forceinline__ void s3_epilogue16(const S3Args& g, int piece, int m0, int n0, int wm, int wn, int lane,
                                              f32x4 (&acc)[WM / 16][WN / 16]) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int MB = WM / 16, NB = WN / 16;
  // (x3 outputs leave in pairs of column blocks: a geometry with an odd number of blocks per wave -- 128 x 192 tiles -- has fp32 outputs only)
  const int r = lane & 15, gq = lane >> 4;
  float* Cp = g.C ? g.C + (size_t)piece * g.c_split_stride : nullptr;
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(Cp, 0, Cp ? g.M * g.ldc * 4 : 0, 0x00020000);
  const bool first = piece == 0;          // the other k-pieces store plain partial sums
  const __amdgpu_buffer_rsrc_t rsCx = __builtin_amdgcn_make_buffer_rsrc(g.Cx, 0, g.Cx ? ((g.M - 1) * g.ldcx + 2 * g.pscx + g.N) * 2 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(g.aux_in), 0, g.aux_in ? g.M * g.ld_aux * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(g.aux_out, 0, g.aux_out ? g.M * g.ld_aux * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsBias = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(g.bias), 0, g.bias ? g.N * 4 : 0, 0x00020000);
  constexpr bool HAS_BIAS = EPI == S3_BIAS || EPI == S3_BIAS_GELU_D;
  constexpr bool HAS_AUX = EPI == S3_ADD || EPI == S3_MUL;
  constexpr bool TWO = EPI == S3_BIAS_GELU_D;
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
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int m = m0 + wm * WM + a * 16 + r;
        const int n = n0 + wn * WN + b * 16 + 4 * gq;
        ax[a][b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsI, n < g.N ? (m * g.ld_aux + n) * 4 : OOB, 0, 0));
      }
  }
  f32x4 csum[EPI == S3_MUL ? NB : 1];                        // column sums of this wave's 64 rows (g.colpart)
#pragma unroll
  for (int b = 0; b < (EPI == S3_MUL ? NB : 1); ++b) csum[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < MB; ++a) {
    const int m = m0 + wm * WM + a * 16 + r;                 // this lane's output row
    f32x4 x2[NB];                                            // second output (gelu') of the row's blocks
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      f32x4& v = acc[a][b];
      if (first) {
        if constexpr (HAS_BIAS) v += bv[b];
        if constexpr (EPI == S3_BIAS_GELU_D) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { float y_, d_; gelu_pair_fast(v[j], y_, d_); v[j] = y_; x2[b][j] = d_; }
        } else if constexpr (EPI == S3_MUL) {
          v *= ax[a][b];
        } else if constexpr (EPI == S3_ADD) {
          v += ax[a][b];
        }
      }
      const int n = n0 + wn * WN + b * 16 + 4 * gq;
      if (Cp) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rsC, n < g.N ? (m * g.ldc + n) * 4 : OOB, 0, X3_ST_AUX);
      if (TWO) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, x2[b]), rsX, n < g.N ? (m * g.ld_aux + n) * 4 : OOB, 0, X3_ST_AUX);
    }
    if constexpr (EPI == S3_MUL) {      // (the one epilogue whose output is a dY with a bias gradient: dU)
      if (g.colpart && first) {
#pragma unroll
        for (int b = 0; b < NB; ++b) csum[b] += acc[a][b];     // (rows beyond M hold zeros: zero operand rows, zero aux)
      }
    }
    // x3 output: two column blocks at a time -- v_permlane16_swap hands the odd 16-lane rows of block b to the even rows and the
    // even rows of block b + 1 to the odd ones, so every lane ends with 8 consecutive columns: one 16-byte store per piece
    if constexpr (NB % 2 == 0) if (g.Cx) {
#pragma unroll
      for (int b = 0; b < NB; b += 2) {
        unsigned wa[3][2], wb[3][2];
        split3_pair(acc[a][b][0], acc[a][b][1], wa[0][0], wa[1][0], wa[2][0]);
        split3_pair(acc[a][b][2], acc[a][b][3], wa[0][1], wa[1][1], wa[2][1]);
        split3_pair(acc[a][b + 1][0], acc[a][b + 1][1], wb[0][0], wb[1][0], wb[2][0]);
        split3_pair(acc[a][b + 1][2], acc[a][b + 1][3], wb[0][1], wb[1][1], wb[2][1]);
        const int n8 = n0 + wn * WN + (b + (gq & 1)) * 16 + 8 * (gq >> 1);
        const bool ok = n8 < g.N;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const auto r0 = __builtin_amdgcn_permlane16_swap(wa[p][0], wb[p][0], false, false);
          const auto r1 = __builtin_amdgcn_permlane16_swap(wa[p][1], wb[p][1], false, false);
          const u32x4_t o = {r0[0], r1[0], r0[1], r1[1]};
          __builtin_amdgcn_raw_buffer_store_b128(o, rsCx, ok ? (m * g.ldcx + p * g.pscx + n8) * 2 : OOB, 0, X3_ST_AUX);
        }
      }
    }
  }
  if constexpr (EPI == S3_MUL) if (g.colpart && first) {
    // the 16 lanes of a quarter wave hold the 16 rows of a block: four shuffle steps inside the quarter, then lane r == 0 of each
    // quarter stores its four columns of every block
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
      if (r == 0 && n < g.N && m0 + wm * 64 < g.M) *reinterpret_cast<f32x4*>(prow + n) = t;      // (no row of partials beyond ceil(M / 64))
    }
  }
#endif
}

// up to four products of one launch (the weight gradients of an encoder layer; a single product otherwise): the tiles of all
// products are numbered through, start[p] = first work item of product p, start[4] = total
struct S3Group {
  S3Args p[4];
  int start[5];
  uniter_x3_riders_t x;     // side work of a grouped weight-gradient launch (kernels instantiated with XTR only)
  // balanced walk (kernels instantiated with SK only): the k-tiles of ALL tiles form one sequence of sk_units = tiles x sk_nk units
  // dealt evenly over the grid; a tile whose k-range is cut hands its later parts over through sk_part (one BM x BN fp32 slot per
  // workgroup) and sk_flag (one word per workgroup and compute wave; zero before the first launch, left zero by every launch)
  float* sk_part;
  unsigned* sk_flag;
  int sk_units, sk_nk;
};

// ---- persistent form with loader waves ------------------------------------------------------------------------------------
// An LDS-DMA instruction holds its wave's issue until the CU's texture-address unit takes it (35-45 cycles per 1-KiB instruction
// when all of a CU's waves load: ~30 B/clk/CU, the L2 -> LDS ceiling), and a 128 x 128 x3 tile needs 48 such instructions per
// 32-deep k-tile against 1536 cycles of MFMAs per SIMD: waves that both load and multiply spend as long blocked in front of the
// address unit as in the matrix pipe, and the two do not overlap (measured: MFMAs alone 0.84 us per k-tile, LDS-DMA + fragment
// reads alone 0.96, together 1.33).  So the roles are split: NWC compute waves (fragment reads + MFMAs + epilogue, never a vector
// memory instruction inside the k-loop) and NWL loader waves (LDS-DMA only), one s_barrier per k-tile between them.  The workgroup
// is persistent over its work items (XCD-chunked, banded tile order) and the k-tiles of all its items form ONE sequence: the
// loaders run ST - 1 k-tiles ahead across item boundaries, so the next tile's first stages land while the compute waves are in
// the epilogue of the previous one, and that epilogue's stores drain under the next tile's k-loop.
//   loader:   [issue k-tiles 0 .. ST-2]  for u: { vmcnt: k-tile u landed;  barrier B_u;  issue k-tile u + ST - 1 -> stage (u - 1) % ST }
//   compute:                            for u: { barrier B_u;  read stage u % ST, MFMAs;  (last k-tile of an item: epilogue) }
// B_u orders k-tile u's LDS-DMA before its reads (every loader waited for its own instructions) and the reads of k-tile u - 1
// (each compute wave waits lgkmcnt(0) before its last MFMAs) before the LDS-DMA that overwrites their stage.
//
// Balanced walk (SK, round 5): a launch whose tiles do not fill a whole number of rounds -- 216 weight-gradient tiles of 128 x 256 on
// 256 CUs, 189 tiles of the QKV product -- leaves CUs idle for the length of a tile.  With SK the unit of work is the K-TILE: the
// k-tiles of all tiles, in tile order, are cut into gridDim.x equal runs (workgroup l, in XCD-chunked order, takes units
// [l U / G, (l + 1) U / G)), so a workgroup's run is: the LAST k-tiles of a tile another workgroup began (it stores its accumulators
// as a partial sum, write-through, and raises its waves' flags), whole tiles, and the FIRST k-tiles of a tile it owns (it waits for the
// flags of the workgroups that hold the rest -- l + 1, l + 2, ..., which stored theirs at the START of their runs -- adds their
// partial sums in that order and runs the epilogue).  The order of the additions is fixed by the grid: results are reproducible
// run to run.  A wave waits only for the same wave of another workgroup (each reads exactly what its namesake stored): no
// workgroup-wide hand-shake beside the k-tile barrier.  No deadlock: a partial sum is stored before its workgroup waits for anything.
template <int BM, int BN, int WM, int WN, bool AKM, bool BKM, int ST, int KT, int NWL, bool M16, int EPI, bool XTR = false, bool SK = false>
__global__ __launch_bounds__(64 * ((BM / WM) * (BN / WN) + NWL), ((BM / WM) * (BN / WN) + NWL + 3) / 4)
void gemm_s3p_kernel(const S3Group G) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int AB = WM / 32, BB = WN / 32, WGN = BN / WN, NWC = (BM / WM) * WGN;
  constexpr int IMG_A = BM * KT * 2, IMG_B = BN * KT * 2, STAGE = 3 * (IMG_A + IMG_B);
  constexpr int KS = KT / 16;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[ST * STAGE];
  const int nwork = G.start[4];
  if (!XTR && !SK && xcd_work_item3(nwork, 0) < 0) return;        // (with riders every workgroup stays: it owns sum-of-squares slots)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  stamp_begin(G.p[0].stamp);
#ifdef UNITER_X3_LAB
  if ((G.p[0].dbg & 16) && G.p[0].bias && tid == 0) {      // shader clock and real time of this workgroup's lifetime
    unsigned long long* o = (unsigned long long*)G.p[0].bias + (size_t)blockIdx.x * 4;
    o[0] = __builtin_amdgcn_s_memtime(); o[2] = __builtin_amdgcn_s_memrealtime();
  }
#endif

  // work item number `round` of this workgroup: product, tile origin, k-tile range
  struct Item { int p, piece, m0, n0, kb, ke; bool valid; };
  // SK: this workgroup's run of units (XCD-chunked order: consecutive runs share an L2; the grid is a multiple of 8)
  const int sk_l = SK ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : 0;
  auto sk_first = [&](int l) -> int { return (int)((long)G.sk_units * l / (long)gridDim.x); };
  const int sk_u0 = SK ? sk_first(sk_l) : 0, sk_u1 = SK ? sk_first(sk_l + 1) : 0;
  const int sk_t0 = SK ? sk_u0 / G.sk_nk : 0;
  auto item = [&](int round) -> Item {
    Item it;
    int w, skb = 0, ske = 0;
    if constexpr (SK) {
      const int ustart = round == 0 ? sk_u0 : (sk_t0 + round) * G.sk_nk;
      w = ustart < sk_u1 ? sk_t0 + round : -1;
      skb = round == 0 ? sk_u0 - sk_t0 * G.sk_nk : 0;
      ske = min(G.sk_nk, skb + (sk_u1 - ustart));
    } else {
      w = xcd_work_item3(nwork, round);
    }
    it.valid = w >= 0;
    if (!it.valid) { it.p = 0; it.piece = it.m0 = it.n0 = it.kb = it.ke = 0; return it; }
    it.p = (w >= G.start[1]) + (w >= G.start[2]) + (w >= G.start[3]);
    const S3Args& g = G.p[it.p];
    const int local = w - G.start[it.p];
    const int tile = local / g.nsplit;
    it.piece = local - tile * g.nsplit;
    int tmi, tni;
    tile_coords3(tile, g.tiles_m, g.tiles_n, g.band_h, tmi, tni);
    it.m0 = tmi * BM; it.n0 = tni * BN;
    const int nk = (g.K + KT - 1) / KT;
    it.kb = (int)((long)nk * it.piece / g.nsplit);
    it.ke = (int)((long)nk * (it.piece + 1) / g.nsplit);
    if constexpr (SK) { it.kb = skb; it.ke = ske; }        // (nsplit == 1, every product has sk_nk k-tiles: the launcher checks)
    return it;
  };
#ifdef UNITER_X3_LAB
  const int dbg = G.p[0].dbg;       // measurement builds (tests/tools/gemm_x3_lab.py): 1 = no LDS-DMA, 2 = no LDS reads, 4 = no MFMAs, 64 = no barriers
#else
  constexpr int dbg = 0;
#endif

  if (wave >= NWC) {
    // ------------------------------------------------------------------ loader waves ----
    set_wave_prio(2);
    typedef Dma3<BM, AKM, NWL, KT, M16> DA;
    typedef Dma3<BN, BKM, NWL, KT, M16> DB;
    constexpr int NDL = 3 * (DA::NI + DB::NI);        // LDS-DMA instructions per loader wave and k-tile
    static_assert((ST - 2) * NDL <= 63, "vmcnt is six bits");
    const int lw = wave - NWC;
    DA da;
    DB db;
    int ir = 0;                   // issue cursor: item, k-tile
    Item it = item(0);
    int ikt = it.kb;
    __amdgpu_buffer_rsrc_t rsA, rsB;
    int kstepA = 0, kstepB = 0, pstepA = 0, pstepB = 0;
    auto bind = [&]() {
      const S3Args& g = G.p[it.p];
      // records: the last row's third piece ends the operand -- whole rows beyond `rows` read as zeros where the pieces of a
      // row lie behind each other (activations); piece-major weights have no rows beyond theirs in a k-loop
      rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), 0,
                                              (((AKM ? g.K : g.M) - 1) * g.lda + 2 * g.psa + (AKM ? g.M : g.K)) * 2, 0x00020000);
      rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.B), 0,
                                              (((BKM ? g.K : g.N) - 1) * g.ldb + 2 * g.psb + (BKM ? g.N : g.K)) * 2, 0x00020000);
#ifdef UNITER_X3_LAB
      const bool pair_a = !AKM && (G.p[0].dbg & 128);      // timing experiment (wrong data, the real footprint)
      const bool pair_b = (!BKM && (G.p[0].dbg & 32)) || g.b_paired != 0;
#else
      constexpr bool pair_a = false;
      const bool pair_b = g.b_paired != 0;                 // the weights as the optimizer's paired-row mirror holds them
#endif
      da.offsets(g.lda, it.m0, lw, lane, pair_a);
      db.offsets(g.ldb, it.n0, lw, lane, pair_b);
      // (k-contiguous paired: the next 32-deep k-tile of a row pair is the next 128-byte line; k-major paired: 32 k-rows = 16 pairs on)
      kstepA = pair_a ? 128 : DA::kstep(g.lda); kstepB = (pair_b && !BKM) ? 128 : DB::kstep(g.ldb);
      pstepA = g.psa * 2; pstepB = g.psb * 2;
    };
    bind();
    int istage = 0, issued = 0, consumed = 0;
    auto issue_next = [&]() {
      while (it.valid && ikt >= it.ke) {
        it = item(++ir);
        ikt = it.kb;
        if (it.valid) bind();
      }
      if (!it.valid) return;
      if (!(dbg & 1)) {
        unsigned char* sbase = smem + istage * STAGE;
        static_for<0, 3 * DA::NI>([&](auto ic) {
          constexpr int I = decltype(ic)::value;
          da.template issue1<I / DA::NI, I % DA::NI>(rsA, sbase, ikt * kstepA, pstepA, lw);
        });
        static_for<0, 3 * DB::NI>([&](auto ic) {
          constexpr int I = decltype(ic)::value;
          db.template issue1<I / DB::NI, I % DB::NI>(rsB, sbase + 3 * IMG_A, ikt * kstepB, pstepB, lw);
        });
      }
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
        if (issued - consumed - 1 >= ST - 2) wait_vm3<(ST - 2) * NDL>(); else wait_vm3<0>();
        if (!(dbg & 64)) __builtin_amdgcn_s_barrier();
        issue_next();
        ++consumed;
      }
    }
  } else if constexpr (M16) {
    // ------------------------------------------------------------------ compute waves, v_mfma_f32_16x16x32_bf16 ----
    constexpr int MB = WM / 16, NB = WN / 16;
    const int wm = wave / WGN, wn = wave % WGN;
    Frag16<BM, AKM, MB> fa;
    Frag16<BN, BKM, NB> fb;
    fa.init(lane, wm * MB);
    fb.init(lane, wn * NB);
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;
    struct Frs {
      typename std::conditional<AKM, FragRegs16KM<BM, MB>, FragRegs16<BM, MB>>::type a;
      typename std::conditional<BKM, FragRegs16KM<BN, NB>, FragRegs16<BN, NB>>::type b;
    };
    constexpr int RP = decltype(Frs::a)::READS_PER_PIECE + decltype(Frs::b)::READS_PER_PIECE;      // LDS reads per piece pair
    f32x4 acc[MB][NB];
    auto mma = [&](Frs& f, auto pac, auto pbc) {
      constexpr int PA = decltype(pac)::value, PB = decltype(pbc)::value;
      static_for<0, MB>([&](auto ac) {
        constexpr int A_ = decltype(ac)::value;
        static_for<0, NB>([&](auto bc) {
          constexpr int B_ = decltype(bc)::value;
          acc[A_][B_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.b.template get<PB, B_>(), f.a.template get<PA, A_>(), acc[A_][B_], 0, 0, 0);
        });
      });
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    // riders (XTR): the sum of squares of everything this wave writes, in double; the column-reduction items of this workgroup --
    // dealt from the END of the grid, where the workgroups with one tile less sit -- run first, while the loader waves fill the
    // first stages of the tile loop
    double wss = 0.0;
    if constexpr (XTR) {
      if (wave < 4)        // (an item is four 16-column strips: the first four compute waves)
        for (int r = (int)gridDim.x - 1 - (int)blockIdx.x; r < G.x.nred; r += (int)gridDim.x)
          wss += (double)riders_reduce_item(G.x, r, wave, lane);
    }
    f32x4 acc1[(XTR && NWC == 4) ? MB : 1];      // XTR: column sums of the k-major A operand (ones . A on the matrix pipe)
    const bf16x8 ones8 = __builtin_bit_cast(bf16x8, u32x4_t{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
    // riders, behind a tile's epilogue: the sum of squares of what this lane stored (the epilogue left the values in `acc`; a
    // k-major operand's overhang columns read its next piece, not zeros: only what was stored counts), and the column sums
    auto tile_riders = [&](const Item& c, bool colsum) -> float {
      float ss = 0.f;
#pragma unroll
      for (int a = 0; a < MB; ++a) {
        const bool row_ok = c.m0 + wm * WM + a * 16 + (lane & 15) < G.p[c.p].M;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const bool ok = row_ok && c.n0 + wn * WN + b * 16 + 4 * (lane >> 4) < G.p[c.p].N;
          float s4 = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) s4 = __builtin_fmaf(acc[a][b][j], acc[a][b][j], s4);
          ss += ok ? s4 : 0.f;
        }
      }
      if constexpr (XTR && NWC == 4) {
        if (colsum && lane < 16) {
          // every accumulator row of the ones-product holds the same sums: lanes 0..15 own column m of row block a
#pragma unroll
          for (int a = 0; a < MB; ++a) {
            const int m = c.m0 + wm * WM + a * 16 + lane;
            if (m < G.p[0].M) {
              const float o = G.x.colsum_out[m] + acc1[a][0];
              G.x.colsum_out[m] = o;
              ss = __builtin_fmaf(o, o, ss);
            }
          }
        }
      }
      return ss;
    };
    int stg = 0;
    for (int r = 0;; ++r) {
      const Item c = item(r);
      if (!c.valid) break;
#pragma unroll
      for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
      // product 0's tiles of the first tile column also sum the columns of its A operand over k (the bias gradient that belongs to
      // this weight gradient: A = dY): the waves that hold the tile's first 64 columns run 3 more MFMAs per row block and k-tile
      const bool colsum = XTR && AKM && NWC == 4 && G.x.colsum_out && c.p == 0 && c.n0 == 0 && wn == 0;
      if constexpr (XTR && NWC == 4) {
#pragma unroll
        for (int a = 0; a < MB; ++a) acc1[a] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if constexpr (NWC > 4) {
        // ---- lean form (12 waves per workgroup: 168 registers): A's three pieces resident (48 registers), B ONE piece at a time
        // in two alternating sets (2 x 16): products a1 b1, a2 b1, a3 b1, a1 b2, a2 b2, a1 b3
        typedef Piece16<BM, AKM, MB> PA_;
        typedef Piece16<BN, BKM, NB> PB_;
        constexpr int RA = PA_::READS, RB = PB_::READS;
        auto mma1 = [&](const PA_& pa, const PB_& pb) {
          static_for<0, MB>([&](auto ac) {
            constexpr int A_ = decltype(ac)::value;
            static_for<0, NB>([&](auto bc) {
              constexpr int B_ = decltype(bc)::value;
              acc[A_][B_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pb.template get<B_>(), pa.template get<A_>(), acc[A_][B_], 0, 0, 0);
            });
          });
        };
        for (int kt = c.kb; kt < c.ke; ++kt) {
          if (!(dbg & 64)) __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
          const unsigned sA = lds0 + stg * STAGE, sB = sA + 3 * IMG_A;
          PA_ a0, a1, a2;
          PB_ bx, by;
          a0.template read<0>(fa, sA); bx.template read<0>(fb, sB);
          a1.template read<1>(fa, sA); a2.template read<2>(fa, sA);
          by.template read<1>(fb, sB);
          lgkm_wait<(2 * RA + RB > 15 ? 15 : 2 * RA + RB)>();
          a0.tie_all(); bx.tie_all();
          mma1(a0, bx);
          __builtin_amdgcn_sched_barrier(0);
          lgkm_wait<(RA + RB > 15 ? 15 : RA + RB)>();
          a1.tie_all();
          mma1(a1, bx);
          __builtin_amdgcn_sched_barrier(0);
          lgkm_wait<(RB > 15 ? 15 : RB)>();
          a2.tie_all();
          mma1(a2, bx);
          __builtin_amdgcn_sched_barrier(0);
          bx.template read<2>(fb, sB);              // (the third piece takes the first one's registers)
          lgkm_wait<(RB > 15 ? 15 : RB)>();
          by.tie_all();
          mma1(a0, by); mma1(a1, by);
          __builtin_amdgcn_sched_barrier(0);
          lgkm_wait<0>();          // every read of the stage is complete in front of the next barrier
          bx.tie_all();
          mma1(a0, bx);
          __builtin_amdgcn_sched_barrier(0);
          stg = stg == ST - 1 ? 0 : stg + 1;
        }
        if constexpr (SK) {
          // a 16 x 16 block of a wave is 1 KiB: [workgroup][wave][block][lane] x 16 bytes, stored and loaded device-coherently (sc1)
          constexpr int AUX_SC1 = 16;
          constexpr int SLOT = BM * BN * 4, WSLOT = MB * NB * 1024;
          if (c.kb > 0) {                 // the later part of a tile another workgroup owns: hand the partial sum over
            const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<unsigned char*>(G.sk_part) + (size_t)sk_l * SLOT + (size_t)wave * WSLOT, 0, WSLOT, 0x00020000);
#pragma unroll
            for (int a = 0; a < MB; ++a)
#pragma unroll
              for (int b = 0; b < NB; ++b)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[a][b]), rsP, ((a * NB + b) * 64 + lane) * 16, 0, AUX_SC1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(G.sk_flag + sk_l * NWC + wave, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
          }
          if (c.ke < G.sk_nk) {           // the first part of a tile this workgroup owns: add the parts the next workgroups hold
            int cover = c.ke;
            for (int j = sk_l + 1; cover < G.sk_nk; ++j) {
              unsigned* fl = G.sk_flag + j * NWC + wave;
              // (a relaxed device-scope poll: an ACQUIRE would invalidate the XCD's L2 on every turn, under the operand panels of every
              // workgroup of the XCD; the partial sums themselves are loaded device-coherently below)
              while (__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(8);
              asm volatile("" ::: "memory");
              const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(
                  reinterpret_cast<unsigned char*>(G.sk_part) + (size_t)j * SLOT + (size_t)wave * WSLOT, 0, WSLOT, 0x00020000);
              // (two row blocks at a time: 32 registers beside the 64 accumulators -- all 64 at once spill)
#pragma unroll
              for (int a0 = 0; a0 < MB; a0 += 2) {
                f32x4 part[2][NB];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                  for (int b = 0; b < NB; ++b)
                    part[a][b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsP, (((a0 + a) * NB + b) * 64 + lane) * 16, 0, AUX_SC1));
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                  for (int b = 0; b < NB; ++b) acc[a0 + a][b] += part[a][b];
                __builtin_amdgcn_sched_barrier(0);
              }
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              if (lane == 0) __hip_atomic_store(fl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (left zero for the next launch)
              cover += min(G.sk_nk - cover, sk_first(j + 1) - sk_first(j));
            }
          }
        }
        s3_epilogue16<WM, WN, EPI>(G.p[c.p], c.piece, c.m0, c.n0, wm, wn, lane, acc);
        if constexpr (XTR) wss += (double)tile_riders(c, false);
        continue;
      }
      static_assert(!SK || NWC > 4, "the balanced walk is built for the 128 x 256 geometry");
      Frs f;
      for (int kt = c.kb; kt < c.ke; ++kt) {
        if (!(dbg & 64)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned sA = lds0 + stg * STAGE, sB = sA + 3 * IMG_A;
        // the k-tile is ONE 32-deep MFMA step: its fragments piece by piece, the six products in the order the pieces land
        if (!(dbg & 2)) {
          f.a.template read_piece<0>(fa, sA); f.b.template read_piece<0>(fb, sB);
          f.a.template read_piece<1>(fa, sA); f.b.template read_piece<1>(fb, sB);
          f.a.template read_piece<2>(fa, sA); f.b.template read_piece<2>(fb, sB);
        }
        lgkm_wait<(2 * RP > 15 ? 15 : 2 * RP)>();
        f.a.template tie_piece<0>(); f.b.template tie_piece<0>();
        if (!(dbg & 4)) mma(f, I0{}, I0{});
        __builtin_amdgcn_sched_barrier(0);
        lgkm_wait<(RP > 15 ? 15 : RP)>();
        f.a.template tie_piece<1>(); f.b.template tie_piece<1>();
        if (!(dbg & 4)) { mma(f, I0{}, I1{}); mma(f, I1{}, I0{}); mma(f, I1{}, I1{}); }
        __builtin_amdgcn_sched_barrier(0);
        lgkm_wait<0>();          // every read of the stage is complete in front of the next barrier (the loaders overwrite it behind it)
        f.a.template tie_piece<2>(); f.b.template tie_piece<2>();
        if (!(dbg & 4)) { mma(f, I0{}, I2{}); mma(f, I2{}, I0{}); }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (XTR && AKM && NWC == 4) {
          if (colsum) {
            static_for<0, MB>([&](auto ac) {
              constexpr int A_ = decltype(ac)::value;
              acc1[A_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones8, f.a.template get<0, A_>(), acc1[A_], 0, 0, 0);
              acc1[A_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones8, f.a.template get<1, A_>(), acc1[A_], 0, 0, 0);
              acc1[A_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones8, f.a.template get<2, A_>(), acc1[A_], 0, 0, 0);
            });
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        stg = stg == ST - 1 ? 0 : stg + 1;
      }
      s3_epilogue16<WM, WN, EPI>(G.p[c.p], c.piece, c.m0, c.n0, wm, wn, lane, acc);
      if constexpr (XTR) wss += (double)tile_riders(c, colsum);
    }
    if constexpr (XTR) {
      riders_store_ssq(G.x, wss, wave, lane, NWC);
    }
  } else {
    // ------------------------------------------------------------------ compute waves, v_mfma_f32_32x32x16_bf16 ----
    const int i5 = lane & 31, h = lane >> 5;
    const int wm = wave / WGN, wn = wave % WGN;
    Frag3<BM, AKM, AB, KT> fa;
    Frag3<BN, BKM, BB, KT> fb;
    fa.init(i5, h, wm * AB);
    fb.init(i5, h, wn * BB);
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;
    struct Frs {
      typename std::conditional<AKM, FragRegsKM<BM, AB, KT>, FragRegs<BM, AB, KT>>::type a;
      typename std::conditional<BKM, FragRegsKM<BN, BB, KT>, FragRegs<BN, BB, KT>>::type b;
    };
    constexpr int RP = decltype(Frs::a)::READS_PER_PIECE + decltype(Frs::b)::READS_PER_PIECE;      // LDS reads per piece pair
    f32x16 acc[AB][BB];
    // all fragments of one k16-step, piece by piece (A's and B's piece 0 first): LDS returns them in this order
    auto read_step = [&](Frs& f, auto ksc, unsigned sA, unsigned sB) {
      constexpr int KS_ = decltype(ksc)::value;
      f.a.template read_piece<KS_, 0>(fa, sA); f.b.template read_piece<KS_, 0>(fb, sB);
      f.a.template read_piece<KS_, 1>(fa, sA); f.b.template read_piece<KS_, 1>(fb, sB);
      f.a.template read_piece<KS_, 2>(fa, sA); f.b.template read_piece<KS_, 2>(fb, sB);
    };
    // products (piece of A, piece of B) of one k16-step on every accumulator block
    auto mma = [&](Frs& f, auto pac, auto pbc) {
      constexpr int PA = decltype(pac)::value, PB = decltype(pbc)::value;
      static_for<0, AB>([&](auto ac) {
        constexpr int A_ = decltype(ac)::value;
        static_for<0, BB>([&](auto bc) {
          constexpr int B_ = decltype(bc)::value;
          acc[A_][B_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b.template get<PB, B_>(), f.a.template get<PA, A_>(), acc[A_][B_], 0, 0, 0);
        });
      });
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    // One k16-step whose reads are in flight: the six products in the order the pieces land -- (1,1) needs only the first
    // pieces, so the matrix pipe starts after a third of the step's LDS traffic instead of all of it (lgkmcnt counts down in issue
    // order; it holds 15 at most, a wait that cannot be expressed waits for a little more).  `next` (the following step's reads)
    // is issued in front of the last two products.  (sched_barrier: MFMAs are plain register operations to the compiler, free to
    // sink below the volatile waits -- it did, and every wave then sat out its fragment reads with the matrix pipe idle.)
    auto step = [&](Frs& f, auto&& next) {
      lgkm_wait<(2 * RP > 15 ? 15 : 2 * RP)>();
      f.a.template tie_piece<0>(); f.b.template tie_piece<0>();
      if (!(dbg & 4)) mma(f, I0{}, I0{});
      __builtin_amdgcn_sched_barrier(0);
      lgkm_wait<(RP > 15 ? 15 : RP)>();
      f.a.template tie_piece<1>(); f.b.template tie_piece<1>();
      if (!(dbg & 4)) { mma(f, I0{}, I1{}); mma(f, I1{}, I0{}); mma(f, I1{}, I1{}); }
      __builtin_amdgcn_sched_barrier(0);
      lgkm_wait<0>();
      f.a.template tie_piece<2>(); f.b.template tie_piece<2>();
      next();
      __builtin_amdgcn_sched_barrier(0);
      if (!(dbg & 4)) { mma(f, I0{}, I2{}); mma(f, I2{}, I0{}); }
      __builtin_amdgcn_sched_barrier(0);
    };
    static_assert(KS == 2, "two k16-steps per k-tile");
    int stg = 0;
    for (int r = 0;; ++r) {
      const Item c = item(r);
      if (!c.valid) break;
#pragma unroll
      for (int a = 0; a < AB; ++a)
#pragma unroll
        for (int b = 0; b < BB; ++b)
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) acc[a][b][rr] = 0.f;
      Frs f0, f1;
      for (int kt = c.kb; kt < c.ke; ++kt) {
        if (!(dbg & 64)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned sA = lds0 + stg * STAGE, sB = sA + 3 * IMG_A;
        if (!(dbg & 2)) read_step(f0, I0{}, sA, sB);
        step(f0, [&]() { if (!(dbg & 2)) read_step(f1, I1{}, sA, sB); });
        // (the last wait of this step is lgkmcnt(0): every read of the stage is complete in front of the next barrier, behind
        // which the loaders overwrite it)
        step(f1, [&]() {});
        stg = stg == ST - 1 ? 0 : stg + 1;
      }
      s3_epilogue<WM, WN, EPI>(G.p[c.p], c.piece, c.m0, c.n0, wm, wn, i5, h, acc);
    }
    }
#ifdef UNITER_X3_LAB
  if ((dbg & 16) && G.p[0].bias && tid == 0) {
    unsigned long long* o = (unsigned long long*)G.p[0].bias + (size_t)blockIdx.x * 4;
    o[1] = __builtin_amdgcn_s_memtime(); o[3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  stamp_end(G.p[0].stamp);
#endif
}

template <int BM>
void plan_tiles3(S3Args& g, int BN) {
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = (g.N + BN - 1) / BN;
  // Band height of the walk (round 5): an XCD's workgroups run ~32 tiles of its chunk of the walk at a time -- a band_h x (32 / band_h)
  // rectangle of the tile grid -- and its L2 fetches band_h row panels and 32 / band_h column panels for them: least for
  // band_h = sqrt(32 BN / BM).  (Round 4 sized the band for L2 capacity, 1.5 MB of row panels: 2 rows at K = 768, every XCD then
  // fetched EVERY weight panel -- 7.1 x the operand bytes on FFN-up forward.  The panels' k-tiles are consumed k-synchronously, so
  // capacity is not the constraint.  Time is unchanged either way -- the re-fetches are Infinity-Cache hits -- but the fabric moves
  // a third less: profiles/r05_pmc_traffic.json.)
  long bh = 1;
  while ((bh + 1) * (bh + 1) * (long)BM <= 32l * BN) ++bh;
  g.band_h = (int)(bh > 16 ? 16 : bh);
  // UNITER_X3_BAND_H: tile rows per band of the walk (lab switch: how many row panels an XCD's chunk of the walk spans decides how
  // often the eight L2s fetch the same operand panels -- VERDICT r04 item 7)
  static const int band_env = [] { const char* e = getenv("UNITER_X3_BAND_H"); return e ? atoi(e) : 0; }();
  if (band_env > 0) g.band_h = band_env;
  if (g.band_h > g.tiles_m) g.band_h = g.tiles_m;
}

// workgroups of a persistent launch over `nwork` items: a multiple of 8 (one chunk of the work per XCD), one per CU at most
int x3_chip_cus() {
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 8 ? n / 8 * 8 : 256;
  }();
  return cus;
}
int x3_grid(int nwork, int max_wgs) {
  int grid = (nwork + 7) / 8 * 8;
  const int cus = x3_chip_cus();
  int cap = max_wgs >= 8 ? max_wgs / 8 * 8 : cus;       // one workgroup per CU: it owns the CU's LDS
  if (g_uniter_cu_reserve > 0) {                          // CUs left to the data-parallel exchange's kernels
    const int room = (cus - g_uniter_cu_reserve) / 8 * 8;
    if (room >= 8 && cap > room) cap = room;
  }
  return grid > cap ? cap : grid;
}

template <int BM, int BN, int WM, int WN, bool AKM, bool BKM, int ST, int KT, int NWL, bool M16, int EPI, bool XTR = false>
int launch_s3p(const S3Group& G, int max_wgs, hipStream_t st) {
  const int grid = x3_grid(G.start[4], max_wgs);
  hipLaunchKernelGGL((gemm_s3p_kernel<BM, BN, WM, WN, AKM, BKM, ST, KT, NWL, M16, EPI, XTR>), dim3(grid),
                     dim3(64 * ((BM / WM) * (BN / WN) + NWL)), 0, st, G);
  UCHECK_LAUNCH();
  return 0;
}

// ---- balanced walk (SK kernels: 128 x 256 tiles) ------------------------------------------------------------------------------
// workspace: [flags: one word per workgroup and compute wave, ZERO before the first launch -- every launch leaves them zero]
//            [partial sums: one 128 x 256 fp32 tile per workgroup]
constexpr size_t SK_FLAG_BYTES = 16384;           // 512 workgroups x 8 compute waves x 4 bytes
constexpr size_t SK_SLOT_BYTES = 128 * 256 * 4;
size_t sk_ws_bytes_() { return SK_FLAG_BYTES + (size_t)x3_chip_cus() * SK_SLOT_BYTES; }
// grid of the balanced walk for `tiles` tiles of `nk` k-tiles each: every CU the launch may use -- or 0: the classic walk is as
// good (its rounds are full), or the workspace is missing.  The hand-over (a 128-KB partial sum stored, flagged and added) is
// priced at four k-tiles
int x3_sk_grid(int tiles, int nk, int max_wgs, const void* ws, size_t ws_bytes) {
  if (!ws || ws_bytes < sk_ws_bytes_() || ((uintptr_t)ws & 255) != 0 || tiles <= 0 || nk <= 0) return 0;
  const int cap = x3_grid(1 << 20, max_wgs);
  const long units = (long)tiles * nk;
  if (cap > 512 || units >= (1l << 30) || units < 8l * cap) return 0;
  const int g0 = x3_grid(tiles, max_wgs);
  const long classic = (long)((tiles + g0 - 1) / g0) * nk, balanced = (units + cap - 1) / cap + 4;
  return classic > balanced ? cap : 0;
}
template <int BM, int BN, int WM, int WN, bool AKM, bool BKM, int ST, int KT, int NWL, bool M16, int EPI, bool XTR>
int launch_s3p_sk(S3Group& G, int grid, void* ws, hipStream_t st) {
  static_assert(BM == 128 && BN == 256, "the partial-sum slots are 128 x 256 tiles");
  G.sk_flag = (unsigned*)ws;
  G.sk_part = (float*)((char*)ws + SK_FLAG_BYTES);
  G.sk_nk = (G.p[0].K + KT - 1) / KT;
  G.sk_units = G.start[4] * G.sk_nk;
  hipLaunchKernelGGL((gemm_s3p_kernel<BM, BN, WM, WN, AKM, BKM, ST, KT, NWL, M16, EPI, XTR, true>), dim3(grid),
                     dim3(64 * ((BM / WM) * (BN / WN) + NWL)), 0, st, G);
  UCHECK_LAUNCH();
  return 0;
}

// cfg: tile geometry (all: 128 x 128 tiles, three 32-deep stages = 144 KB of LDS, one persistent workgroup per CU)
//   1: 8 compute waves of 64 x 32 + 4 loader waves, v_mfma_f32_32x32x16_bf16
//   2: 4 compute waves of 64 x 64 + 4 loader waves, v_mfma_f32_32x32x16_bf16
//   3: 4 compute waves of 64 x 64 + 4 loader waves, v_mfma_f32_16x16x32_bf16 (default)
//   4: 128 x 256 tiles (round 5): 8 compute waves of 64 x 64 + 4 loader waves, TWO stages of 72 KB, v_mfma_f32_16x16x32_bf16 with the
//      B fragments one piece at a time (168 registers per wave) -- 36 instead of 48 KB staged per 128 x 128 x 32 block of products
//   5: 128 x 192 tiles (forward layout, fp32 output): 8 compute waves of 64 x 48, two stages of 60 KB -- for products whose N is a
//      multiple of 192 and whose 128 x 256 tiles leave CUs idle (the QKV product at configs[1]: 189 tiles of 128 x 256, 252 of 128 x 192)
template <bool AKM, bool BKM, int EPI>
int dispatch_cfg3p(int cfg, const S3Group& G, int max_wgs, hipStream_t st) {
  switch (cfg) {
    case 1: return launch_s3p<128, 128, 64, 32, AKM, BKM, 3, 32, 4, false, EPI>(G, max_wgs, st);
    case 2: return launch_s3p<128, 128, 64, 64, AKM, BKM, 3, 32, 4, false, EPI>(G, max_wgs, st);
    case 3: return launch_s3p<128, 128, 64, 64, AKM, BKM, 3, 32, 4, true, EPI>(G, max_wgs, st);
    case 4:
      if constexpr (!AKM) return launch_s3p<128, 256, 64, 64, AKM, BKM, 2, 32, 4, true, EPI>(G, max_wgs, st);
      // (weight gradients keep the 128 x 128 whole-K tiles)
    case 5:
      if constexpr (!AKM && !BKM) return launch_s3p<128, 192, 64, 48, AKM, BKM, 2, 32, 4, true, EPI>(G, max_wgs, st);
      // (forward layout only)
    default: uniter_set_error("gemm_x3: bad cfg %d (1..5; 4 not for weight gradients, 5 for the forward layout only)", cfg); return UNITER_E_ARG;
  }
}

struct SkWs { void* p; size_t bytes; };

template <bool AKM, bool BKM, int EPI>
int dispatch_cfg3(int cfg, const S3Args& g, hipStream_t st, SkWs sk) {
  S3Group G;
  memset(&G.x, 0, sizeof(G.x));
  G.sk_part = nullptr; G.sk_flag = nullptr; G.sk_units = G.sk_nk = 0;
  for (int p = 0; p < 4; ++p) G.p[p] = g;
  plan_tiles3<128>(G.p[0], cfg == 4 ? 256 : cfg == 5 ? 192 : 128);
  const int total = G.p[0].tiles_m * G.p[0].tiles_n * g.nsplit;
  G.start[0] = 0;
  for (int p = 1; p <= 4; ++p) G.start[p] = total;
  // the balanced walk: built for the forward products with a bias epilogue (the QKV product: 189 tiles of 128 x 256 at configs[1])
  if constexpr (!AKM && !BKM && EPI == S3_BIAS) {
    if (cfg == 4 && g.nsplit == 1) {
      const int grid = x3_sk_grid(total, (g.K + 31) / 32, 0, sk.p, sk.bytes);
      if (grid > 0) return launch_s3p_sk<128, 256, 64, 64, AKM, BKM, 2, 32, 4, true, EPI, false>(G, grid, sk.p, st);
    }
  }
  return dispatch_cfg3p<AKM, BKM, EPI>(cfg, G, 0, st);
}

template <bool AKM, bool BKM>
int dispatch_epi3(int cfg, const S3Args& g, int epi, hipStream_t st, SkWs sk) {
  // forward products (weights k-contiguous): none / bias / bias + GELU; input gradients (weights k-major): none / add / mul;
  // weight gradients (both k-major): none / add
  if (epi == UNITER_EPI_NONE) return dispatch_cfg3<AKM, BKM, S3_NONE>(cfg, g, st, sk);
  if constexpr (!AKM && !BKM) {
    if (epi == UNITER_EPI_BIAS) return dispatch_cfg3<AKM, BKM, S3_BIAS>(cfg, g, st, sk);
    if (epi == UNITER_EPI_BIAS_GELU_D) return dispatch_cfg3<AKM, BKM, S3_BIAS_GELU_D>(cfg, g, st, sk);
  } else {
    if (epi == UNITER_EPI_ADD) return dispatch_cfg3<AKM, BKM, S3_ADD>(cfg, g, st, sk);
    if constexpr (!AKM) {
      if (epi == UNITER_EPI_MUL) return dispatch_cfg3<AKM, BKM, S3_MUL>(cfg, g, st, sk);
    }
  }
  uniter_set_error("gemm_x3: epilogue %d is not built for this operand layout", epi);
  return UNITER_E_ARG;
}

// ---- fp32 -> x3 (every tensor no producer kernel writes in pieces) ------------------------------------------------------
// piece p of x[r][c] -> x3[r * rs + p * ps + c]
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, int rows, int cols, int ld,
                                                     unsigned short* __restrict__ x3, size_t rs, size_t ps) {
  const int c8 = blockIdx.x * 32 + (threadIdx.x & 31);       // 8 consecutive columns per thread
  const int r = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (r >= rows || c8 * 8 >= cols) return;
  const f32x4 v0 = *reinterpret_cast<const f32x4*>(x + (size_t)r * ld + c8 * 8);
  const f32x4 v1 = *reinterpret_cast<const f32x4*>(x + (size_t)r * ld + c8 * 8 + 4);
  unsigned w[3][4];
  split3_pair(v0[0], v0[1], w[0][0], w[1][0], w[2][0]);
  split3_pair(v0[2], v0[3], w[0][1], w[1][1], w[2][1]);
  split3_pair(v1[0], v1[1], w[0][2], w[1][2], w[2][2]);
  split3_pair(v1[2], v1[3], w[0][3], w[1][3], w[2][3]);
#pragma unroll
  for (int p = 0; p < 3; ++p)
    *reinterpret_cast<u32x4_t*>(x3 + (size_t)r * rs + p * ps + c8 * 8) = u32x4_t{w[p][0], w[p][1], w[p][2], w[p][3]};
}

// fp32 -> the weight mirror (three piece-major copies of the flat parameter buffer) with a per-chunk destination table: chunk c (64
// consecutive parameters = two 32-element units of one weight row) goes to dst[c] (first unit) and dst[c] + 64 (second unit) -- the
// paired-row layout of the encoder layers' weights -- or, dst[c] < 0, to its own place
__global__ __launch_bounds__(256) void split3_mirror_kernel(const float* __restrict__ x, size_t n4, size_t first, unsigned short* __restrict__ mirror,
                                                            size_t ps, const int* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;       // one f32x4 per thread
  if (i >= n4) return;
  const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
  unsigned w[3][2];
  split3_pair(v[0], v[1], w[0][0], w[1][0], w[2][0]);
  split3_pair(v[2], v[3], w[0][1], w[1][1], w[2][1]);
  const size_t e = first + 4 * i;                                 // absolute element
  size_t o = e;
  if (dst) {
    const int d = dst[e >> 6];
    if (d >= 0) o = (size_t)d + ((e >> 5) & 1) * 64 + (e & 31);
  }
#pragma unroll
  for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x2_t*>(mirror + p * ps + o) = u32x2_t{w[p][0], w[p][1]};
}

// x3 -> fp32 (tests, and consumers that want the plain tensor back): the exact sum of the three pieces
__global__ __launch_bounds__(256) void join3_kernel(const unsigned short* __restrict__ x3, int rows, int cols, size_t rs, size_t ps,
                                                    float* __restrict__ x, int ld) {
  const int c8 = blockIdx.x * 32 + (threadIdx.x & 31);
  const int r = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (r >= rows || c8 * 8 >= cols) return;
  u32x4_t w[3];
#pragma unroll
  for (int p = 0; p < 3; ++p) w[p] = *reinterpret_cast<const u32x4_t*>(x3 + (size_t)r * rs + p * ps + c8 * 8);
  float o[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    o[2 * e] = (bflo(w[2][e]) + bflo(w[1][e])) + bflo(w[0][e]);
    o[2 * e + 1] = (bfhi(w[2][e]) + bfhi(w[1][e])) + bfhi(w[0][e]);
  }
  *reinterpret_cast<f32x4*>(x + (size_t)r * ld + c8 * 8) = f32x4{o[0], o[1], o[2], o[3]};
  *reinterpret_cast<f32x4*>(x + (size_t)r * ld + c8 * 8 + 4) = f32x4{o[4], o[5], o[6], o[7]};
}

// column sums of an x3 tensor [rows][3][ldx] added to out[cols] (the bias gradient of intermediate.dense from dU, which exists
// only as x3: 48 MB read once, on the weight-gradient stream): block (bx, by) covers 512 columns (64 lanes x 8, three 16-byte
// loads per row and lane) and rows by * RPB .. + RPB; four row-lanes meet in LDS, then one fp32 atomic per column and block
__global__ __launch_bounds__(256) void colsum3_kernel(const unsigned short* __restrict__ x3, int rows, int cols, int ldx,
                                                      float* __restrict__ out, int rows_per_block) {
  __shared__ float red[4][64 * 8 + 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 512 + lane * 8;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c < cols) {
    const int r1 = min(rows, (int)(blockIdx.y + 1) * rows_per_block);
    for (int r = blockIdx.y * rows_per_block + wave; r < r1; r += 4) {
      const unsigned short* p = x3 + (size_t)r * 3 * ldx + c;
      const u32x4_t v0 = *reinterpret_cast<const u32x4_t*>(p), v1 = *reinterpret_cast<const u32x4_t*>(p + ldx),
                    v2 = *reinterpret_cast<const u32x4_t*>(p + 2 * ldx);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        s[2 * k] += (bflo(v2[k]) + bflo(v1[k])) + bflo(v0[k]);
        s[2 * k + 1] += (bfhi(v2[k]) + bfhi(v1[k])) + bfhi(v0[k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[wave][lane * 8 + k] = s[k];
  __syncthreads();
  for (int k = threadIdx.x; k < 512; k += 256) {
    const int cc = blockIdx.x * 512 + k;
    if (cc < cols) atomicAdd(out + cc, (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]));
  }
}

// 31-bit buffer offsets: the largest byte offset an x3 operand of `rows` rows is addressed with (tile overhang included)
bool x3_fits(size_t rows, int rs, int ps, int ext) { return ((rows + 256) * (size_t)rs + 2 * (size_t)ps + ext) * 2 < (1ull << 31); }

}  // namespace

size_t gemm_x3_sk_ws_bytes() { return sk_ws_bytes_(); }
int gemm_chip_cus() { return x3_chip_cus(); }      // CUs of the current device, a multiple of 8 (shared with gemm_bf16_dma.hip)

int riders_prepare(uniter_x3_riders_t& x, const char* who) {
  UCHECK_ARG(x.njobs >= 0 && x.njobs <= 4, "%s: at most 4 column-reduction jobs", who);
  int items = 0;
  for (int j = 0; j < x.njobs; ++j) {
    UCHECK_ARG(x.part[j] && x.nparts[j] > 0 && x.n[j] > 0 && x.seg[j] > 0 && x.stride[j] >= x.n[j], "%s: bad reduction job %d", who, j);
    UCHECK_SHAPE(x.stride[j] % 4 == 0 && x.seg[j] % 4 == 0 && x.n[j] % 4 == 0 && x.n[j] <= 3 * x.seg[j] && ((uintptr_t)x.part[j] & 15) == 0,
                 "%s: reduction job %d: stride, segment and column count must be multiples of 4, 16-byte aligned partials", who, j);
    for (int o = 0; o < 3; ++o)
      UCHECK_SHAPE(((uintptr_t)x.out[j][o] & 15) == 0, "%s: reduction job %d: output %d is not 16-byte aligned", who, j, o);
    x.first_item[j] = items;
    items += (x.n[j] + 63) / 64;
  }
  for (int j = x.njobs; j < 5; ++j) x.first_item[j] = items;
  x.nred = items;
  UCHECK_SHAPE(!x.colsum_out || ((uintptr_t)x.colsum_out & 3) == 0, "%s: colsum_out alignment", who);
  return 0;
}

void x3_choose(int M, int N, int K, int avail, int nsplit_fixed, int* cfg_out, int* ns_out, bool allow192 = false);

// C / Cx = epi(A . B^T) on x3 operands (fp32-accurate, six bf16 MFMA products per block, fp32 accumulate).
int gemm_x3_run(int cfg, int nsplit, int a_kmajor, int b_kmajor, int M, int N, int K, const void* A, int lda, int psa,
                const void* B, int ldb, int psb, float* C, int ldc, long c_split_stride, void* Cx, int ldcx, int pscx,
                int epilogue, const float* bias, const float* aux_in, float* aux_out, int ld_aux, void* stream,
                float* colsum_part, void* sk_ws, size_t sk_ws_bytes) {
  UCHECK_ARG(M > 0 && N > 0 && K > 0 && A && B && (C || Cx), "gemm_x3: bad argument");
  UCHECK_ARG(!a_kmajor || (b_kmajor && !Cx && (epilogue == UNITER_EPI_NONE || epilogue == UNITER_EPI_ADD)),
             "gemm_x3: A k-major only as the weight-gradient layout (both operands k-major, fp32 output, none / add)");
  UCHECK_ARG(nsplit >= 1 && nsplit <= 8 && (nsplit == 1 || (C && !Cx && c_split_stride >= (long)M * ldc)),
             "gemm_x3: split-K needs fp32 slabs (no x3 output)");
  UCHECK_ARG(!(epilogue == UNITER_EPI_BIAS || epilogue == UNITER_EPI_BIAS_GELU_D) || bias, "gemm_x3: epilogue needs bias");
  UCHECK_ARG(!(epilogue == UNITER_EPI_ADD || epilogue == UNITER_EPI_MUL) || aux_in, "gemm_x3: epilogue needs aux_in");
  UCHECK_ARG(epilogue != UNITER_EPI_BIAS_GELU_D || aux_out, "gemm_x3: epilogue needs aux_out");
  UCHECK_SHAPE(((a_kmajor && b_kmajor) || K % 32 == 0) && lda % 8 == 0 && ldb % 8 == 0 && psa % 8 == 0 && psb % 8 == 0 && N % 8 == 0 &&
               (!a_kmajor || M % 8 == 0) && ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 && (ldc % 4 == 0) &&
               (ldcx % 8 == 0) && (pscx % 8 == 0) && (ld_aux % 4 == 0) && ((uintptr_t)C & 15) == 0 && ((uintptr_t)Cx & 15) == 0 &&
               ((uintptr_t)aux_in & 15) == 0 && ((uintptr_t)aux_out & 15) == 0 && ((uintptr_t)bias & 15) == 0,
               "gemm_x3: K %% 32, N %% 8, strides %% 8 (x3) / %% 4 (fp32) and 16-byte aligned buffers required "
               "(M=%d N=%d K=%d)", M, N, K);
  UCHECK_SHAPE(lda > 0 && ldb > 0 && psa >= 0 && psb >= 0 && x3_fits(a_kmajor ? K : M, lda, psa, a_kmajor ? M : K) &&
               x3_fits(b_kmajor ? K : N, ldb, psb, b_kmajor ? N : K) && (!Cx || (ldcx > 0 && pscx >= 0 && x3_fits(M, ldcx, pscx, N))) &&
               ((size_t)M + 256) * (ldc > 0 ? ldc : 1) * 4 < (1ull << 31) &&
               ((size_t)M + 256) * (ld_aux > 0 ? ld_aux : 1) * 4 < (1ull << 31), "gemm_x3: operand beyond 31-bit offsets");
  S3Args g;
  g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.psa = psa; g.B = B; g.ldb = ldb; g.psb = psb; g.C = C; g.ldc = ldc;
  g.c_split_stride = c_split_stride; g.Cx = (unsigned short*)Cx; g.ldcx = ldcx; g.pscx = pscx; g.bias = bias;
  g.aux_in = aux_in; g.aux_out = aux_out; g.ld_aux = ld_aux;
  g.tiles_m = g.tiles_n = 0; g.band_h = 1; g.nsplit = nsplit;
  g.dbg = cfg >> 8; g.b_paired = (cfg >> 6) & 1; cfg &= 0x3f;      // (cfg | 64: B in the paired-row layout of the weight mirror)
  UCHECK_ARG(!g.b_paired || (!a_kmajor && N % 2 == 0 && ldb % 32 == 0), "gemm_x3: paired rows are the layout of a weight (B, forward or input-gradient "
             "product) with an even number of rows and a row length that is a multiple of 32");
  g.stamp = take_stamp_slot();
  g.prio = take_launch_prio();
  UCHECK_ARG((cfg & 0xff) != 5 || (!a_kmajor && !b_kmajor && !Cx), "gemm_x3: cfg 5 (128 x 192 tiles) is built for the forward layout with an fp32 output");
  UCHECK_ARG(!colsum_part || (epilogue == UNITER_EPI_MUL && nsplit == 1 && (cfg == 0 || cfg >= 3) && ((uintptr_t)colsum_part & 15) == 0 && N % 4 == 0),
             "gemm_x3: column partials ride on the x aux epilogue of the 16x16x32 geometries (cfg 0, 3, 4), one k-piece");
  g.colpart = colsum_part;
  // in the step: 4 x (64 x 64) compute waves + 4 loaders are +1.2 % over cfg 1 (8 x (64 x 32)); the 16x16x32 MFMA shape another
  // +3 % (10.51 -> 10.21 ms, same box): the chip holds a higher clock under it
  if (cfg == 0) {
    cfg = 3;
    // round 5: 128 x 256 tiles where they take fewer rounds' worth of time.  The k-loop is bound by the LDS-DMA issue rate (one
    // 1-KiB instruction per ~45 cycles and CU), and a 128 x 256 tile stages 72 KB per k-tile for twice the products of a 128 x 128
    // one's 48 KB: measured 1.75 x the time per k-tile for 2 x the work (profiles/r05_gemm_x3_lab.txt: QKV forward 61.0 -> 54.9 us,
    // FFN-up forward 76.7 -> 69.5, FFN-down input gradient 72.9 -> 65.2).  x3_choose prices both geometries on the CUs this
    // launch may use (all of them, or what a data-parallel exchange leaves: g_uniter_cu_reserve)
    if (!a_kmajor) {
      int ns_;
      x3_choose(M, N, K, x3_grid(1 << 20, 0), nsplit, &cfg, &ns_, !b_kmajor && !Cx);
    }
  }
  hipStream_t st = (hipStream_t)stream;
  const SkWs sk = {sk_ws, sk_ws_bytes};
  if (a_kmajor) return dispatch_epi3<true, true>(cfg, g, epilogue, st, sk);
  return b_kmajor ? dispatch_epi3<false, true>(cfg, g, epilogue, st, sk) : dispatch_epi3<false, false>(cfg, g, epilogue, st, sk);
}

// Tile geometry (cfg 3 = 128 x 128 persistent, cfg 4 = 128 x 256 one-round) and k-pieces of a forward / input-gradient product on
// `avail` CUs: the cost model of profiles/r05_gemm_x3_lab.txt -- rounds x k-tiles per work item x time per k-tile (1.2 us for a
// 128 x 128 tile, 2.1 us for a 128 x 256 one) plus 4 us per slab (an 8-MB write and an 8-MB read in the consuming row pass at
// configs[1]).  nsplit_fixed > 0: the caller chose the k-pieces (x3 outputs cannot be split); 0: pieces 1..4 compete (N <= 1024
// only: more tiles than that fill the chip without them).  Written for `avail` < the chip's CUs too: while a data-parallel
// exchange holds CUs (uniter_model_set_cu_reserve) the 252-item forms that exactly fit 256 CUs would run two rounds on 240.
void x3_choose(int M, int N, int K, int avail, int nsplit_fixed, int* cfg_out, int* ns_out, bool allow192) {
  static const bool wide_on = [] { const char* e = getenv("UNITER_X3_WIDE"); return !(e && e[0] == '0'); }();
  static const bool on192 = [] { const char* e = getenv("UNITER_X3_192"); return !(e && e[0] == '0'); }();
  if (avail < 8) avail = 8;
  const long tm = (M + 127) / 128, t128 = tm * ((N + 127) / 128), t256 = tm * ((N + 255) / 256), t192 = tm * ((N + 191) / 192);
  const int nk = (K + 31) / 32;
  double best = 1e30;
  int bc = 3, bn = nsplit_fixed > 0 ? nsplit_fixed : 1;
  const int ns_lo = nsplit_fixed > 0 ? nsplit_fixed : 1, ns_hi = nsplit_fixed > 0 ? nsplit_fixed : ((N <= 1024 && K >= 512) ? 4 : 1);
  for (int ns = ns_lo; ns <= ns_hi; ++ns) {
    if (nk / ns < 6 && ns > 1) continue;                     // pieces shorter than six k-tiles do not pay for their prologue
    for (int c = 3; c <= 5; ++c) {
      if (c == 4 && (!wide_on || N < 256)) continue;
      // 128 x 192 tiles (forward layout, fp32 output): 60 KB staged per k-tile against the 128 x 256 tile's 72 -- 1.75 us
      if (c == 5 && (!allow192 || !on192 || !wide_on || N % 192 != 0)) continue;
      const long items = (c == 4 ? t256 : c == 5 ? t192 : t128) * ns;
      const long rounds = (items + avail - 1) / avail;
      const double t = (double)rounds * ((nk + ns - 1) / ns) * (c == 4 ? 2.1 : c == 5 ? 1.75 : 1.2) + 4.0 * ns;
      if (t < best - 1e-9) { best = t; bc = c; bn = ns; }
    }
  }
  *cfg_out = bc; *ns_out = bn;
}

extern "C" int uniter_gemm_x3_plan(int M, int N, int K, int avail_cus, int nsplit_fixed, int* cfg, int* nsplit) {
  UCHECK_ARG(M > 0 && N > 0 && K > 0 && cfg && nsplit && nsplit_fixed >= 0 && nsplit_fixed <= 8, "gemm_x3_plan: bad argument");
  x3_choose(M, N, K, avail_cus > 0 ? avail_cus : x3_grid(1 << 20, 0), nsplit_fixed, cfg, nsplit);
  return 0;
}

// the same for a forward product (both operands k-contiguous) with an fp32 output: 128 x 192 tiles (cfg 5) compete too
extern "C" int uniter_gemm_x3_plan_fwd32(int M, int N, int K, int avail_cus, int nsplit_fixed, int* cfg, int* nsplit) {
  UCHECK_ARG(M > 0 && N > 0 && K > 0 && cfg && nsplit && nsplit_fixed >= 0 && nsplit_fixed <= 8, "gemm_x3_plan_fwd32: bad argument");
  x3_choose(M, N, K, avail_cus > 0 ? avail_cus : x3_grid(1 << 20, 0), nsplit_fixed, cfg, nsplit, true);
  return 0;
}

// k-pieces of a product whose output goes to fp32 slabs, on `avail` CUs (0 = the chip's)
int gemm_x3_pick_split_on(int M, int N, int K, int avail) {
  int c, n;
  x3_choose(M, N, K, avail > 0 ? avail : x3_grid(1 << 20, 0), 0, &c, &n);
  return n;
}

// Pieces for the split-K slab form of the products whose N is the hidden size (126 tiles of 128 x 128 for 256 persistent
// workgroups at M = 2624): two pieces fill the chip (K = 3072: 107 -> 68 us, K = 768: 32 -> 24 us; four pieces lose again:
// profiles/r04_gemm_x3_lab.txt); the consumer's row pass adds the slabs (uniter_ln_fwd_slabs / uniter_ln_bwd_rows_slabs).
int gemm_x3_pick_split(int M, int N, int K) {
  const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
  return (tiles <= 160 && K >= 512) ? 2 : 1;
}

extern "C" int uniter_gemm_x3_cfg(int cfg, int nsplit, int a_kmajor, int b_kmajor, int M, int N, int K, const void* A,
                                  int lda, int psa, const void* B, int ldb, int psb, float* C, int ldc, long c_split_stride,
                                  void* C_x3, int ldcx, int pscx, int epilogue, const float* bias, const float* aux_in,
                                  float* aux_out, int ld_aux, void* stream) {
  return gemm_x3_run(cfg, nsplit, a_kmajor, b_kmajor, M, N, K, A, lda, psa, B, ldb, psb, C, ldc, c_split_stride, C_x3, ldcx,
                     pscx, epilogue, bias, aux_in, aux_out, ld_aux, stream, nullptr, nullptr, 0);
}

// the same, with the workspace of the balanced walk (uniter_gemm_x3_balanced_ws_bytes; its first 16 KB zero before the first launch,
// every launch leaves them zero; one workspace per stream that launches concurrently): a 128 x 256-tile forward product with a bias
// epilogue whose tiles do not fill whole rounds of the chip is cut into equal runs of k-tiles instead (gemm_s3p_kernel, "balanced walk")
extern "C" size_t uniter_gemm_x3_balanced_ws_bytes(void) { return gemm_x3_sk_ws_bytes(); }
extern "C" int uniter_gemm_x3_cfg_ws(int cfg, int nsplit, int a_kmajor, int b_kmajor, int M, int N, int K, const void* A,
                                     int lda, int psa, const void* B, int ldb, int psb, float* C, int ldc, long c_split_stride,
                                     void* C_x3, int ldcx, int pscx, int epilogue, const float* bias, const float* aux_in,
                                     float* aux_out, int ld_aux, void* ws, size_t ws_bytes, void* stream) {
  return gemm_x3_run(cfg, nsplit, a_kmajor, b_kmajor, M, N, K, A, lda, psa, B, ldb, psb, C, ldc, c_split_stride, C_x3, ldcx,
                     pscx, epilogue, bias, aux_in, aux_out, ld_aux, stream, nullptr, ws, ws_bytes);
}

// the same product, also leaving partial column sums of its output: colsum_part[(i, n)] = sum of the output rows 64 i .. 64 i + 63
// of column n ([(M + 63) / 64][N] floats; UNITER_EPI_MUL only: the product that writes dU, whose column sums are intermediate.dense's
// bias gradient, model/layer.py:140 backward) -- a column-reduction job of the weight-gradient launch's riders finishes them
extern "C" int uniter_gemm_x3_colpart(int cfg, int a_kmajor, int b_kmajor, int M, int N, int K, const void* A, int lda, int psa,
                                      const void* B, int ldb, int psb, float* C, int ldc, void* C_x3, int ldcx, int pscx,
                                      const float* aux_in, int ld_aux, float* colsum_part, void* stream) {
  return gemm_x3_run(cfg, 1, a_kmajor, b_kmajor, M, N, K, A, lda, psa, B, ldb, psb, C, ldc, 0, C_x3, ldcx, pscx, UNITER_EPI_MUL,
                     nullptr, aux_in, nullptr, ld_aux, stream, colsum_part, nullptr, 0);
}

extern "C" int uniter_split3(const float* x, int rows, int cols, int ld, void* x3, size_t row_stride, size_t piece_stride,
                             void* stream) {
  UCHECK_ARG(x && x3 && rows > 0 && cols > 0, "split3: bad argument");
  UCHECK_SHAPE(cols % 8 == 0 && ld % 4 == 0 && row_stride % 8 == 0 && piece_stride % 8 == 0 && ((uintptr_t)x & 15) == 0 &&
               ((uintptr_t)x3 & 15) == 0, "split3: cols %% 8, ld %% 4, strides %% 8 and 16-byte aligned buffers required");
  hipLaunchKernelGGL(split3_kernel, dim3((cols / 8 + 31) / 32, (rows + 7) / 8), dim3(256), 0, (hipStream_t)stream, x, rows, cols, ld,
                     (unsigned short*)x3, row_stride, piece_stride);
  UCHECK_LAUNCH();
  return 0;
}

// Refresh elements [first, first + n) of the x3 weight mirror (piece p of element e at mirror + p * piece_stride + e, or, with a
// destination table, chunk-wise at pair_dst[e / 64] (+ 64 for the chunk's second 32-element unit): the paired-row layout
// ParamStore.mirror_pair_dst describes -- what uniter_adam_step_x3p writes and uniter_gemm_x3_cfg (cfg | 64) reads).
extern "C" int uniter_mirror_refresh_x3(const float* params_base, size_t first, size_t n, void* mirror, size_t piece_stride,
                                        const int* pair_dst, void* stream) {
  UCHECK_ARG(params_base && mirror && n > 0, "mirror_refresh_x3: bad argument");
  UCHECK_SHAPE(first % 64 == 0 && n % 4 == 0 && piece_stride % 4 == 0 && ((uintptr_t)params_base & 15) == 0 && ((uintptr_t)mirror & 7) == 0,
               "mirror_refresh_x3: first %% 64, n %% 4, 16-byte aligned parameters required");
  const size_t n4 = n / 4;
  hipLaunchKernelGGL(split3_mirror_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params_base + first, n4, first,
                     (unsigned short*)mirror, piece_stride, pair_dst);
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_join3(const void* x3, int rows, int cols, size_t row_stride, size_t piece_stride, float* x, int ld,
                            void* stream) {
  UCHECK_ARG(x && x3 && rows > 0 && cols > 0, "join3: bad argument");
  UCHECK_SHAPE(cols % 8 == 0 && ld % 4 == 0 && row_stride % 8 == 0 && piece_stride % 8 == 0 && ((uintptr_t)x & 15) == 0 &&
               ((uintptr_t)x3 & 15) == 0, "join3: cols %% 8, ld %% 4, strides %% 8 and 16-byte aligned buffers required");
  hipLaunchKernelGGL(join3_kernel, dim3((cols / 8 + 31) / 32, (rows + 7) / 8), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)x3, rows, cols, row_stride, piece_stride, x, ld);
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_colsum_x3_add(const void* x3, int rows, int cols, int ldx, float* out, void* stream) {
  UCHECK_ARG(x3 && out && rows > 0 && cols > 0 && ldx >= cols, "colsum_x3_add: bad argument");
  UCHECK_SHAPE(cols % 8 == 0 && ldx % 8 == 0 && ((uintptr_t)x3 & 15) == 0, "colsum_x3_add: cols, ldx multiples of 8 and a 16-byte aligned operand required");
  // ~2048 workgroups: enough loads in flight for the HBM stream, few enough atomics per column (as uniter_colsum_bf16_add)
  const int cb = (cols + 511) / 512;
  int rb = (2048 + cb - 1) / cb;
  if (rb > (rows + 3) / 4) rb = (rows + 3) / 4;
  if (rb < 1) rb = 1;
  const int rpb = (rows + rb - 1) / rb;
  hipLaunchKernelGGL(colsum3_kernel, dim3(cb, (rows + rpb - 1) / rpb), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)x3, rows, cols, ldx, out, rpb);
  UCHECK_LAUNCH();
  return 0;
}

// dW_p[M_p, N_p] (+)= A_p^T B_p for up to four products of one reduction length K (A_p [K][3][M_p], B_p [K][3][N_p] x3,
// dW_p fp32 with leading dimension N_p), one launch of whole-K tiles.
// tile width of the grouped weight-gradient launch: cfg 4 = 128 x 256 tiles (round 5), else 128 x 128
static int wgrad_bn(int cfg) { return cfg == 4 ? 256 : 128; }
static int wgrad_tiles(int cfg, int n, const int* Mo, const int* No) {
  int total = 0;
  for (int p = 0; p < n; ++p) total += ((Mo[p] + 127) / 128) * ((No[p] + wgrad_bn(cfg) - 1) / wgrad_bn(cfg));
  return total;
}
// the default geometry of the grouped launch (UNITER_X3_WGRAD_CFG = 3 | 4 overrides)
int gemm_x3_wgrad_default_cfg() {
  static const int c = [] { const char* e = getenv("UNITER_X3_WGRAD_CFG"); const int v = e ? atoi(e) : 0; return (v == 3 || v == 4) ? v : 4; }();
  return c;
}

int gemm_x3_wgrad_group(int cfg, int n, const int* Mo, const int* No, int K, const void* const* A, const void* const* B,
                        float* const* dW, void* stream, int overwrite, int max_wgs, uniter_x3_riders_t* riders,
                        void* sk_ws, size_t sk_ws_bytes) {
  UCHECK_ARG(n >= 1 && n <= 4 && K > 0 && Mo && No && A && B && dW, "wgrad_x3_group: bad argument");
  if (cfg == 0) cfg = gemm_x3_wgrad_default_cfg();
  S3Group G;
  memset(&G.x, 0, sizeof(G.x));
  G.sk_part = nullptr; G.sk_flag = nullptr; G.sk_units = G.sk_nk = 0;
  unsigned long long* stamp = take_stamp_slot();
  int total = 0;
  for (int p = 0; p < 4; ++p) {
    G.start[p] = total;
    if (p >= n) { G.p[p] = G.p[0]; continue; }
    UCHECK_ARG(Mo[p] > 0 && No[p] > 0 && A[p] && B[p] && dW[p], "wgrad_x3_group: bad product %d", p);
    UCHECK_SHAPE(Mo[p] % 8 == 0 && No[p] % 8 == 0 && ((uintptr_t)A[p] & 15) == 0 && ((uintptr_t)B[p] & 15) == 0 &&
                 ((uintptr_t)dW[p] & 15) == 0 && x3_fits(K, 3 * Mo[p], Mo[p], Mo[p]) && x3_fits(K, 3 * No[p], No[p], No[p]) &&
                 ((size_t)Mo[p] + 256) * No[p] * 4 < (1ull << 31),
                 "wgrad_x3_group: M, N %% 8, 16-byte aligned buffers, 31-bit offsets (product %d: %d x %d, K=%d)", p, Mo[p], No[p], K);
    S3Args& g = G.p[p];
    g.M = Mo[p]; g.N = No[p]; g.K = K; g.A = A[p]; g.lda = 3 * Mo[p]; g.psa = Mo[p]; g.B = B[p]; g.ldb = 3 * No[p]; g.psb = No[p];
    g.C = dW[p]; g.ldc = No[p];
    g.c_split_stride = 0; g.Cx = nullptr; g.ldcx = 0; g.pscx = 0; g.bias = nullptr;
    g.aux_in = overwrite ? nullptr : dW[p]; g.aux_out = nullptr; g.ld_aux = No[p];
    g.nsplit = 1; g.stamp = stamp; g.prio = 0; g.dbg = 0; g.colpart = nullptr; g.b_paired = 0;
    plan_tiles3<128>(g, wgrad_bn(cfg));
    total += g.tiles_m * g.tiles_n;
  }
  G.start[4] = total;
  for (int p = n; p < 4; ++p) G.start[p] = total;
  hipStream_t st = (hipStream_t)stream;
  // the balanced walk (128 x 256 tiles, a workspace given): 216 tiles for UNITER-base on 256 CUs -- 82 k-tiles per workgroup become 70
  const int sk_grid = cfg == 4 ? x3_sk_grid(total, (K + 31) / 32, max_wgs, sk_ws, sk_ws_bytes) : 0;
  if (riders) {
    // riders ride on the v_mfma_f32_16x16x32_bf16 geometries: 4 compute waves of 64 x 64 (cfg 3) or the 128 x 256 tile's 8 (cfg 4)
    UCHECK_ARG(cfg == 3 || cfg == 4, "wgrad_x3_group: riders need cfg 3 or 4");
    uniter_x3_riders_t& x = *riders;
    UCHECK_ARG(cfg != 4 || !x.colsum_out, "wgrad_x3_group: colsum_out rides on cfg 3 only (the 128 x 256 geometry has no registers for it: "
               "take the bias gradient from the producing product's column partials, uniter_gemm_x3_colpart, as a reduction job)");
    x.grid = sk_grid > 0 ? sk_grid : x3_grid(total, max_wgs);
    UCHECK_RC(riders_prepare(x, "wgrad_x3_group"));
    G.x = x;
    if (sk_grid > 0)
      return overwrite ? launch_s3p_sk<128, 256, 64, 64, true, true, 2, 32, 4, true, S3_NONE, true>(G, sk_grid, sk_ws, st)
                       : launch_s3p_sk<128, 256, 64, 64, true, true, 2, 32, 4, true, S3_ADD, true>(G, sk_grid, sk_ws, st);
    if (cfg == 4)
      return overwrite ? launch_s3p<128, 256, 64, 64, true, true, 2, 32, 4, true, S3_NONE, true>(G, max_wgs, st)
                       : launch_s3p<128, 256, 64, 64, true, true, 2, 32, 4, true, S3_ADD, true>(G, max_wgs, st);
    return overwrite ? launch_s3p<128, 128, 64, 64, true, true, 3, 32, 4, true, S3_NONE, true>(G, max_wgs, st)
                     : launch_s3p<128, 128, 64, 64, true, true, 3, 32, 4, true, S3_ADD, true>(G, max_wgs, st);
  }
  if (sk_grid > 0)
    return overwrite ? launch_s3p_sk<128, 256, 64, 64, true, true, 2, 32, 4, true, S3_NONE, false>(G, sk_grid, sk_ws, st)
                     : launch_s3p_sk<128, 256, 64, 64, true, true, 2, 32, 4, true, S3_ADD, false>(G, sk_grid, sk_ws, st);
  if (cfg == 4)
    return overwrite ? launch_s3p<128, 256, 64, 64, true, true, 2, 32, 4, true, S3_NONE>(G, max_wgs, st)
                     : launch_s3p<128, 256, 64, 64, true, true, 2, 32, 4, true, S3_ADD>(G, max_wgs, st);
  return overwrite ? dispatch_cfg3p<true, true, S3_NONE>(cfg, G, max_wgs, st) : dispatch_cfg3p<true, true, S3_ADD>(cfg, G, max_wgs, st);
}

// sum-of-squares slots a launch with riders writes (one per compute wave: 4 per workgroup, 8 with 128 x 256 tiles) for these products;
// K > 0 and a workspace size: the launch is given the balanced walk's workspace (it then runs on every CU it may use)
int gemm_x3_wgrad_group_slots(int cfg, int n, const int* Mo, const int* No, int max_wgs, int K, size_t sk_ws_bytes) {
  if (!Mo || !No || n < 1 || n > 4) return 0;
  if (cfg == 0) cfg = gemm_x3_wgrad_default_cfg();
  const int total = wgrad_tiles(cfg, n, Mo, No);
  // (x3_sk_grid checks the pointer's alignment only: any aligned non-null value stands for the workspace here)
  const int sk_grid = (cfg == 4 && K > 0) ? x3_sk_grid(total, (K + 31) / 32, max_wgs, (const void*)256, sk_ws_bytes) : 0;
  return (cfg == 4 ? 8 : 4) * (sk_grid > 0 ? sk_grid : x3_grid(total, max_wgs));
}

// the smallest grid (a multiple of 8) on which these products' tiles take no more rounds than on one workgroup per CU
int gemm_x3_wgrad_group_balanced_wgs(int cfg, int n, const int* Mo, const int* No) {
  if (!Mo || !No || n < 1 || n > 4) return 0;
  if (cfg == 0) cfg = gemm_x3_wgrad_default_cfg();
  const int total = wgrad_tiles(cfg, n, Mo, No);
  const int full = x3_grid(total, 0);
  const int rounds = (total + full - 1) / full;
  const int wgs = ((total + rounds - 1) / rounds + 7) / 8 * 8;
  return wgs < full ? wgs : full;
}

extern "C" int uniter_wgrad_x3_group_riders(int cfg, int n, const int* M, const int* N, int K, const void* const* A,
                                            const void* const* B, float* const* dW, int overwrite, int max_wgs,
                                            uniter_x3_riders_t* riders, void* stream) {
  return gemm_x3_wgrad_group(cfg, n, M, N, K, A, B, dW, stream, overwrite, max_wgs, riders, nullptr, 0);
}
extern "C" int uniter_wgrad_x3_group_slots(int cfg, int n, const int* M, const int* N, int max_wgs) {
  return gemm_x3_wgrad_group_slots(cfg, n, M, N, max_wgs, 0, 0);
}
// the same two with the workspace of the balanced walk (uniter_gemm_x3_balanced_ws_bytes, rules as uniter_gemm_x3_cfg_ws; riders may
// be NULL): with 128 x 256 tiles that do not fill whole rounds the launch is cut into equal runs of k-tiles and runs on every CU it
// may use -- the slot count follows (K = the reduction length, ws_bytes = the size of the workspace the launch will be given)
extern "C" int uniter_wgrad_x3_group_ws(int cfg, int n, const int* M, const int* N, int K, const void* const* A,
                                        const void* const* B, float* const* dW, int overwrite, int max_wgs,
                                        uniter_x3_riders_t* riders, void* ws, size_t ws_bytes, void* stream) {
  return gemm_x3_wgrad_group(cfg, n, M, N, K, A, B, dW, stream, overwrite, max_wgs, riders, ws, ws_bytes);
}
extern "C" int uniter_wgrad_x3_group_slots_ws(int cfg, int n, const int* M, const int* N, int K, int max_wgs, size_t ws_bytes) {
  return gemm_x3_wgrad_group_slots(cfg, n, M, N, max_wgs, K, ws_bytes);
}

extern "C" int uniter_wgrad_x3_group(int cfg, int n, const int* M, const int* N, int K, const void* const* A,
                                     const void* const* B, float* const* dW, int overwrite, int max_wgs, void* stream) {
  return gemm_x3_wgrad_group(cfg, n, M, N, K, A, B, dW, stream, overwrite, max_wgs, nullptr, nullptr, 0);
}
