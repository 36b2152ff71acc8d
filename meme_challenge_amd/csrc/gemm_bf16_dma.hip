// bf16-resident GEMM for gfx950, second generation: operands go global -> LDS by LDS-DMA
// (buffer_load_dwordx4 ... lds), no staging registers and no ds_write traffic; a ring of LDS stages with
// counted s_waitcnt vmcnt(N) and one raw s_barrier per 64-deep k-tile keeps two k-tiles of loads in flight
// across the barrier; the accumulator is held TRANSPOSED (MFMA A operand = the weight rows, B operand = the
// activation rows) so that each lane owns one output row and 4 consecutive columns per register group:
// fp32 outputs leave as 16-byte stores, bf16 outputs as 16-byte stores after a v_permlane32_swap of
// neighbouring groups, the bias / residual / gelu' operands arrive as 16- or 8-byte loads.
//
// Replaces cuBLAS behind nn.Linear forward and input-gradient products of model/layer.py:76-78 (query / key /
// value), :112 (attention output), :140 (intermediate) and :153 (output) in the bf16 mode.
//
// LDS images (per operand and stage, R = rows of the tile, 64 k per k-tile, no padding: LDS-DMA writes
// 1 KiB = 64 lanes x 16 B contiguously, so the swizzle sits on the per-lane SOURCE address and on the read):
//   k-contiguous operand ([rows][K] in memory): [R][64] bf16, 128-B rows, 16-B chunk c of row r stored at
//     chunk c ^ ((r >> 1) & 7): 16 consecutive rows x one k-chunk cover all 64 banks (ds_read_b128).
//   k-major operand ([K][cols] in memory: the weight of an input-gradient product): [64 k][R] bf16 as
//     256-B segments, chunk c (0..15) of k-row k stored at c ^ (((k & 3) << 2) | ((k >> 2) & 3)); the
//     MFMA operand is gathered by ds_read_b64_tr_b16 (4 k-rows x 16 columns per 16-lane group), whose 32
//     lanes per half then touch 32 distinct 8-byte units.
//
// Split-K: a tile's k-range may be cut into `nsplit` pieces computed by different workgroups; piece s
// stores its fp32 partial tile to C + s * c_split_stride (piece 0 applies the epilogue); the CONSUMER
// (LayerNorm forward / backward row pass) adds the slabs -- no atomics, no in-kernel hand-off.
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "riders.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct GArgsD {
  int M, N, K;
  const void* A; int lda;      // bf16
  const void* B; int ldb;      // bf16
  float* C; int ldc;           // fp32 output (optional); slab s of a split-K launch at C + s * c_split_stride
  long c_split_stride;
  unsigned short* Cb; int ldcb;   // bf16 output (optional)
  int epi;
  const float* bias;
  const void* aux_in; int aux_in_bf16;
  void* aux_out; int aux_out_bf16;
  int ld_aux;
  int tiles_m, tiles_n, band_h, nsplit;
  unsigned long long* stamp;    // optional {first start, last end} slot (common.h)
  int prio;      // wave priority (common.h: g_uniter_launch_prio)
  int dbg;       // measurement builds only (tests/tools/gemm_v2_lab.py): 1 = drop every output store, 2 = skip the k-loop
};

constexpr int KT = 64;                 // k-tile depth
constexpr int OOB = 0x7ffffff0;        // buffer offset beyond every descriptor: load returns 0, store is dropped

__device__ __forceinline__ void tile_coords_d(int t, int tiles_m, int tiles_n, int band_h, int& tm, int& tn) {
  const int full = band_h * tiles_n;
  const int band = t / full;
  const int rem = t - band * full;
  const int bh = min(band_h, tiles_m - band * band_h);
  tn = rem / bh;
  tm = band * band_h + (rem - tn * bh);
}

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// ---- LDS-DMA fill of one operand image --------------------------------------------------------------
template <int R, bool KM, int NW>
struct Dma {
  static constexpr int NI = R / 8 / NW;          // 1-KiB wave-instructions per wave and k-tile
  static_assert(NI >= 1 && NI * 8 * NW == R, "tile rows must be a multiple of 8 x waves");
  static_assert(!KM || R == 128 || R == 256, "k-major tiles are 128 or 256 wide");
  int voff[NI];
  static __device__ __forceinline__ int kstep(int ld) { return (KM ? KT * ld : KT) * 2; }
  __device__ __forceinline__ void offsets(int ld, int rc0, int wave, int lane) {
#pragma unroll
    for (int t = 0; t < NI; ++t) {
      const int j = wave + NW * t;
      if constexpr (!KM) {
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
  __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t rs, unsigned char* img, int soff, int wave) const {
#pragma unroll
    for (int t = 0; t < NI; ++t)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(img + (wave + NW * t) * 1024), 16, voff[t], soff, 0, 0);
  }
};

// ---- MFMA operand fragments ---------------------------------------------------------------------------
// 8 consecutive k (k16-step ks, lane half h) of row / column 32 blk + i5 of the operand image.
// k-contiguous images are read by plain 16-byte LDS loads (the compiler schedules and counts them).
// k-major images are read by ds_read_b64_tr_b16 issued as INLINE ASSEMBLY with hand-counted lgkmcnt waits:
// behind the builtin form of that read hipcc (ROCm 7.2) waits s_waitcnt vmcnt(0) whenever an LDS-DMA is in
// flight, which would drain the two-tile prefetch every k-tile (checked in the .s).
template <int R, bool KM>
struct Frag {
  int o0, o1;         // k-contiguous: row byte offset, swizzle key
  unsigned ka[4];     // k-contiguous, asm form: image-relative byte address of k16-step ks (first block of the wave)
  unsigned tr[2][2];  // k-major: image-relative byte address of the (first, second) read of the wave's two blocks
  __device__ __forceinline__ void init(int i5, int h, int blk0) {
    if constexpr (!KM) {
      o0 = i5 * 128; o1 = (h ^ ((i5 >> 1) & 7));      // (xor with 2 ks below)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) ka[ks] = o0 + blk0 * 32 * 128 + (((2 * ks) ^ o1) << 4);
    } else {
      const int l16 = i5 & 15, q = l16 >> 2, p = l16 & 3;
      const int s1 = (q << 2) | (2 * h), s2 = s1 | 1;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int r0 = (blk0 + t) * 32;
        const int seg = r0 >> 7;
        const int c = ((r0 & 127) >> 3) + 2 * (i5 >> 4) + (p >> 1);
        const int base = (8 * h + q) * (R * 2) + 8 * (p & 1) + seg * 256;
        tr[t][0] = base + ((c ^ s1) << 4);
        tr[t][1] = base + 4 * (R * 2) + ((c ^ s2) << 4);
      }
    }
  }
  __device__ __forceinline__ bf16x8 read(const unsigned char* img, int blk, int ks) const {
    static_assert(!KM, "k-major images are read through read_tr");
    const int off = o0 + blk * 32 * 128 + (((2 * ks) ^ o1) << 4);
    return *reinterpret_cast<const bf16x8*>(img + off);
  }
  // asm form of read() for the kernels whose other operand is k-major: one kind of LDS read per loop, all counted by hand
  template <int KS, int T>
  __device__ __forceinline__ void read_asm(unsigned img_addr, u32x4_t& out) const {
#if defined(__HIP_DEVICE_COMPILE__)      // the host pass parses this body too and knows no "v" constraint
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(out) : "v"(img_addr + ka[KS]), "n"(T * 32 * 128) : "memory");
#endif
  }
  // raw halves of the fragment of block t, k16-step KS; complete only after the caller's lgkmcnt wait
  template <int KS>
  __device__ __forceinline__ void read_tr(unsigned img_addr, int t, u32x2_t& lo, u32x2_t& hi) const {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(img_addr + tr[t][0]), "n"(KS * 16 * R * 2) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(img_addr + tr[t][1]), "n"(KS * 16 * R * 2) : "memory");
#endif
  }
};

__device__ __forceinline__ bf16x8 join_halves(u32x2_t lo, u32x2_t hi) {
  const u32x4_t v = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(bf16x8, v);
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N>
__device__ __forceinline__ void wait_lgkm(u32x2_t& a, u32x2_t& b, u32x2_t& c, u32x2_t& d, u32x4_t& e, u32x4_t& f) {
#if defined(__HIP_DEVICE_COMPILE__)
  // naming every destination "+v" ties their first use to this wait (and keeps register-only MFMAs below it)
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "n"(N) : "memory");
#endif
}

// BM x BN tile, one 64 x 64 sub-tile (2 x 2 accumulator blocks of 32 x 32) per wave, ST LDS stages.
// SWAP: accumulator transposed (lane = output row); !SWAP: lane = output column (fp32 atomics for C +=).
template <int N>
__device__ __forceinline__ void wait_lgkm8(u32x2_t& a, u32x2_t& b, u32x2_t& c, u32x2_t& d, u32x2_t& e, u32x2_t& f,
                                           u32x2_t& g, u32x2_t& h) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "n"(N) : "memory");
#endif
}

// work item of this workgroup, XCD-chunked (blocks b and b + 8 share an XCD's L2: consecutive items go to one XCD);
// -1 for the padding blocks of the grid.  round > 0: the items a workgroup of a grid SMALLER than the work takes after
// its first one (the grouped weight-gradient launch on a capped grid), still inside its XCD's chunk
__device__ __forceinline__ int xcd_work_item(int nwork, int round = 0) {
  const int xcd = blockIdx.x & 7, idx = (blockIdx.x >> 3) + round * (int)(gridDim.x >> 3);
  const int q8 = nwork >> 3, r8 = nwork & 7;
  const int chunk0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int chunk_n = q8 + (xcd < r8 ? 1 : 0);
  return idx < chunk_n ? chunk0 + idx : -1;
}

// one work item w = (tile, k-piece) of the product g
// what a tile of a launch with riders hands back / is asked for (gemm_dma_wgrad_group_kernel<.., XTR = true>)
struct TileRider {
  float* colsum_out;   // non-null: also sum this tile's A operand (k-major) over k, out[m] += .. (the tile's first 64-column strip only)
  int colsum_M;
  float ss;            // out: this lane's sum of squares of what the tile stored (and of the column sums it wrote)
};

template <int BM, int BN, bool AKM, bool BKM, bool SWAP, int ST, int EPI, bool XTR = false>
__device__ __forceinline__ void gemm_dma_tile(const GArgsD& g, const int w, unsigned char* smem, TileRider* rt = nullptr) {
#if defined(__HIP_DEVICE_COMPILE__)      // device-only builtins / asm: the host pass gets an empty body (it only needs the launch stub)
  constexpr int WNN = BN / 64, NW = (BM / 64) * WNN;
  constexpr int IMG_A = BM * 128, IMG_B = BN * 128, STAGE = IMG_A + IMG_B;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i5 = lane & 31, h = lane >> 5;
  const int wm = wave / WNN, wn = wave % WNN;

  const int tile = w / g.nsplit, piece = w - tile * g.nsplit;
  int tmi, tni;
  tile_coords_d(tile, g.tiles_m, g.tiles_n, g.band_h, tmi, tni);
  const int m0 = tmi * BM, n0 = tni * BN;
  const int nk = (g.K + KT - 1) / KT;
  const int kb = (int)((long)nk * piece / g.nsplit), ke = (g.dbg & 2) ? kb : (int)((long)nk * (piece + 1) / g.nsplit);

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(g.A), 0, (AKM ? g.K : g.M) * g.lda * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(g.B), 0, (BKM ? g.K : g.N) * g.ldb * 2, 0x00020000);
  Dma<BM, AKM, NW> da;
  Dma<BN, BKM, NW> db;
  da.offsets(g.lda, m0, wave, lane);
  db.offsets(g.ldb, n0, wave, lane);
  const int kstepA = Dma<BM, AKM, NW>::kstep(g.lda), kstepB = Dma<BN, BKM, NW>::kstep(g.ldb);
  constexpr int NDMA = Dma<BM, AKM, NW>::NI + Dma<BN, BKM, NW>::NI;

  Frag<BM, AKM> fa;
  Frag<BN, BKM> fb;
  fa.init(i5, h, wm * 2);
  fb.init(i5, h, wn * 2);
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) acc[a][b][rr] = 0.f;
  // riders: the column sums of the k-major A operand over k, as ones . A on the matrix pipe (two more MFMAs per k16-step for
  // the waves that hold the tile's first 64 columns, in the tiles of the first tile column)
  f32x16 acc1[XTR ? 2 : 1];
  bool colsum = false;
  if constexpr (XTR) {
    colsum = rt->colsum_out != nullptr && n0 == 0 && wn == 0;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) acc1[a][rr] = 0.f;
  }
  const bf16x8 ones8 = __builtin_bit_cast(bf16x8, u32x4_t{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});

#define ISSUE(KTILE, STG)                                                        \
  do {                                                                           \
    da.issue(rsA, smem + (STG) * STAGE, (KTILE) * kstepA, wave);                 \
    db.issue(rsB, smem + (STG) * STAGE + IMG_A, (KTILE) * kstepB, wave);         \
  } while (0)

// One k-tile: four k16-steps, fragments of step s+1 read while step s multiplies.  LDS instructions per step
// of a k-major B: 2 + 4 asm reads and no compiler-counted LDS read at all (a compiler read between them would be
// waited for with a count that ignores the asm reads, i.e. far too early in the queue); every asm statement
// clobbers "memory", so the issue order is the program order and lgkmcnt(6) at step s leaves exactly step
// s+1's six reads outstanding.
#define RD_STEP(KS, BUF)                                                                                 \
  {                                                                                                      \
    if constexpr (AKM && BKM) {           /* weight gradients: both operands k-major, 8 transposed reads per step */ \
      fa.template read_tr<KS>(sAaddr, 0, rA[BUF][0][0], rA[BUF][0][1]);                                  \
      fa.template read_tr<KS>(sAaddr, 1, rA[BUF][1][0], rA[BUF][1][1]);                                  \
      fb.template read_tr<KS>(sBaddr, 0, rb[BUF][0][0], rb[BUF][0][1]);                                  \
      fb.template read_tr<KS>(sBaddr, 1, rb[BUF][1][0], rb[BUF][1][1]);                                  \
    } else if constexpr (BKM) {                                                                          \
      fa.template read_asm<KS, 0>(sAaddr, ra[BUF][0]);                                                   \
      fa.template read_asm<KS, 1>(sAaddr, ra[BUF][1]);                                                   \
      fb.template read_tr<KS>(sBaddr, 0, rb[BUF][0][0], rb[BUF][0][1]);                                  \
      fb.template read_tr<KS>(sBaddr, 1, rb[BUF][1][0], rb[BUF][1][1]);                                  \
    } else {                                                                                             \
      xa[BUF][0] = fa.read(sA, wm * 2, KS); xa[BUF][1] = fa.read(sA, wm * 2 + 1, KS);                    \
      xb[BUF][0] = fb.read(sB, wn * 2, KS); xb[BUF][1] = fb.read(sB, wn * 2 + 1, KS);                    \
    }                                                                                                    \
  }
#define MM_STEP(BUF, NLATER)                                                                             \
  {                                                                                                      \
    if constexpr (AKM && BKM) {                                                                          \
      wait_lgkm8<(NLATER) ? 8 : 0>(rA[BUF][0][0], rA[BUF][0][1], rA[BUF][1][0], rA[BUF][1][1],           \
                                   rb[BUF][0][0], rb[BUF][0][1], rb[BUF][1][0], rb[BUF][1][1]);          \
      xa[BUF][0] = join_halves(rA[BUF][0][0], rA[BUF][0][1]);                                            \
      xa[BUF][1] = join_halves(rA[BUF][1][0], rA[BUF][1][1]);                                            \
      xb[BUF][0] = join_halves(rb[BUF][0][0], rb[BUF][0][1]);                                            \
      xb[BUF][1] = join_halves(rb[BUF][1][0], rb[BUF][1][1]);                                            \
    } else if constexpr (BKM) {                                                                          \
      wait_lgkm<NLATER>(rb[BUF][0][0], rb[BUF][0][1], rb[BUF][1][0], rb[BUF][1][1], ra[BUF][0], ra[BUF][1]); \
      xa[BUF][0] = __builtin_bit_cast(bf16x8, ra[BUF][0]);                                               \
      xa[BUF][1] = __builtin_bit_cast(bf16x8, ra[BUF][1]);                                               \
      xb[BUF][0] = join_halves(rb[BUF][0][0], rb[BUF][0][1]);                                            \
      xb[BUF][1] = join_halves(rb[BUF][1][0], rb[BUF][1][1]);                                            \
    }                                                                                                    \
    _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                        \
    _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                        \
      acc[a][b] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb[BUF][b], xa[BUF][a], acc[a][b], 0, 0, 0) \
                       : __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[BUF][a], xb[BUF][b], acc[a][b], 0, 0, 0); \
    if constexpr (XTR && AKM && BKM && SWAP) {                                                           \
      if (colsum) {                                                                                      \
        acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones8, xa[BUF][0], acc1[0], 0, 0, 0);          \
        acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones8, xa[BUF][1], acc1[1], 0, 0, 0);          \
      }                                                                                                  \
    }                                                                                                    \
  }
#define COMPUTE(STG)                                                                                     \
  {                                                                                                      \
    const unsigned char* sA = smem + (STG) * STAGE;                                                      \
    const unsigned char* sB = sA + IMG_A;                                                                \
    const unsigned sAaddr = lds0 + (STG) * STAGE, sBaddr = sAaddr + IMG_A;                               \
    (void)sA; (void)sB; (void)sAaddr; (void)sBaddr;                                                      \
    bf16x8 xa[2][2], xb[2][2];                                                                           \
    u32x2_t rb[2][2][2], rA[2][2][2];                                                                    \
    u32x4_t ra[2][2];                                                                                    \
    RD_STEP(0, 0)                                                                                        \
    RD_STEP(1, 1) MM_STEP(0, 6)                                                                          \
    RD_STEP(2, 0) MM_STEP(1, 6)                                                                          \
    RD_STEP(3, 1) MM_STEP(0, 6)                                                                          \
    MM_STEP(1, 0)                                                                                        \
  }

  if (kb < ke) {
    if constexpr (ST == 3) {
      ISSUE(kb, 0);
      if (kb + 1 < ke) ISSUE(kb + 1, 1);
      int kt = kb;
#define STEP3(STG)                                                               \
  {                                                                              \
    if (kt + 1 < ke) wait_vm<NDMA>(); else wait_vm<0>();                         \
    __builtin_amdgcn_s_barrier();                                                \
    __builtin_amdgcn_sched_barrier(0);                                           \
    if (kt + 2 < ke) ISSUE(kt + 2, ((STG) + 2) % 3);                             \
    COMPUTE(STG)                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                           \
  }
      while (true) {
        STEP3(0) if (++kt == ke) break;
        STEP3(1) if (++kt == ke) break;
        STEP3(2) if (++kt == ke) break;
      }
#undef STEP3
    } else {
      ISSUE(kb, 0);
      int kt = kb;
#define STEP2(STG)                                                               \
  {                                                                              \
    wait_vm<0>();                                                                \
    __builtin_amdgcn_s_barrier();                                                \
    __builtin_amdgcn_sched_barrier(0);                                           \
    if (kt + 1 < ke) ISSUE(kt + 1, 1 - (STG));                                   \
    COMPUTE(STG)                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                           \
  }
      while (true) {
        STEP2(0) if (++kt == ke) break;
        STEP2(1) if (++kt == ke) break;
      }
#undef STEP2
    }
  }
#undef ISSUE
#undef COMPUTE
#undef RD_STEP
#undef MM_STEP

  // ---------------------------------------------------------------- epilogue ----
  // The epilogue kind is a template parameter (one code path per kernel: the unrolled blocks are long), and every
  // load of the tile (bias, aux operand of all four blocks) is issued before the first store: loads and stores share
  // vmcnt, so a load result needed behind stores in flight would drain them (s_waitcnt vmcnt(0)).
  float* Cp = g.C ? g.C + (size_t)piece * g.c_split_stride : nullptr;
  const int keep = (g.dbg & 1) ? 0 : 1;     // zero records: the range check drops the stores, the instruction stream stays
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(Cp, 0, Cp ? keep * g.M * g.ldc * 4 : 0, 0x00020000);
  if constexpr (SWAP) {
    const bool first = piece == 0;          // the other k-pieces store plain partial sums
    const __amdgpu_buffer_rsrc_t rsCb = __builtin_amdgcn_make_buffer_rsrc(g.Cb, 0, g.Cb ? keep * g.M * g.ldcb * 2 : 0, 0x00020000);
    const int axe_i = g.aux_in_bf16 ? 2 : 4, axe_o = g.aux_out_bf16 ? 2 : 4;
    const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(g.aux_in), 0, g.aux_in ? g.M * g.ld_aux * axe_i : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        g.aux_out, 0, g.aux_out ? keep * g.M * g.ld_aux * axe_o : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsBias = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.bias), 0, g.bias ? g.N * 4 : 0, 0x00020000);
    constexpr bool HAS_BIAS = EPI == UNITER_EPI_BIAS || EPI == UNITER_EPI_BIAS_GELU || EPI == UNITER_EPI_BIAS_GELU_D;
    constexpr bool HAS_AUX = EPI == UNITER_EPI_DGELU || EPI == UNITER_EPI_ADD || EPI == UNITER_EPI_MUL;
    constexpr bool TWO = EPI == UNITER_EPI_BIAS_GELU_D || EPI == UNITER_EPI_BIAS_GELU;
    // register group gq of block (a, b) holds columns nb(b) + 8 gq + 4 h .. + 3 of row m(a)
    f32x4 bv[HAS_BIAS ? 2 : 1][4], ax[HAS_AUX ? 2 : 1][HAS_AUX ? 2 : 1][4];
    if (HAS_BIAS && first) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int n = n0 + wn * 64 + b * 32 + 8 * gq + 4 * h;
          bv[b][gq] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsBias, n < g.N ? n * 4 : OOB, 0, 0));
        }
    }
    if (HAS_AUX && first) {
      // one uniform branch around ALL loads (a per-load branch makes hipcc wait vmcnt(0) after each of them)
      if (g.aux_in_bf16) {
        u32x2_t raw[2][2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              const int m = m0 + wm * 64 + a * 32 + i5;
              const int n = n0 + wn * 64 + b * 32 + 8 * gq + 4 * h;
              raw[a][b][gq] = __builtin_amdgcn_raw_buffer_load_b64(rsI, n < g.N ? (m * g.ld_aux + n) * 2 : OOB, 0, 0);
            }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              const u32x2_t t = raw[a][b][gq];
              ax[a][b][gq] = f32x4{bf_lo(t[0]), bf_hi(t[0]), bf_lo(t[1]), bf_hi(t[1])};
            }
      } else {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              const int m = m0 + wm * 64 + a * 32 + i5;
              const int n = n0 + wn * 64 + b * 32 + 8 * gq + 4 * h;
              ax[a][b][gq] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsI, n < g.N ? (m * g.ld_aux + n) * 4 : OOB, 0, 0));
            }
      }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int m = m0 + wm * 64 + a * 32 + i5;                 // this lane's output row
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int nb = n0 + wn * 64 + b * 32;                   // first column of the block
        f32x16& v = acc[a][b];
        f32x16 x2;     // second output (gelu' or the pre-activation)
        if (first) {
          if constexpr (HAS_BIAS) {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) v[rr] += bv[b][rr >> 2][rr & 3];
          }
          if constexpr (EPI == UNITER_EPI_BIAS_GELU_D) {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) { float y_, d_; gelu_pair_fast(v[rr], y_, d_); v[rr] = y_; x2[rr] = d_; }
          } else if constexpr (EPI == UNITER_EPI_BIAS_GELU) {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) { x2[rr] = v[rr]; v[rr] = gelu_erf(v[rr]); }
          } else if constexpr (EPI == UNITER_EPI_MUL) {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) v[rr] *= ax[a][b][rr >> 2][rr & 3];
          } else if constexpr (EPI == UNITER_EPI_ADD) {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) v[rr] += ax[a][b][rr >> 2][rr & 3];
          } else if constexpr (EPI == UNITER_EPI_DGELU) {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) v[rr] *= dgelu_erf(ax[a][b][rr >> 2][rr & 3]);
          }
        }
        // fp32 outputs: one 16-byte store per register group
        if (Cp) {
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int n = nb + 8 * gq + 4 * h;
            const f32x4 o = {v[4 * gq], v[4 * gq + 1], v[4 * gq + 2], v[4 * gq + 3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rsC, n < g.N ? (m * g.ldc + n) * 4 : OOB, 0, 0);
            if constexpr (XTR) {      // only what was stored counts (a k-major operand's overhang columns read the next row, not zeros)
              const float s4 = (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
              rt->ss += (m < g.M && n < g.N) ? s4 : 0.f;
            }
          }
        }
        if (TWO && !g.aux_out_bf16) {
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int n = nb + 8 * gq + 4 * h;
            const f32x4 o = {x2[4 * gq], x2[4 * gq + 1], x2[4 * gq + 2], x2[4 * gq + 3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rsX, n < g.N ? (m * g.ld_aux + n) * 4 : OOB, 0, 0);
          }
        }
        // bf16 outputs: groups (gq, gq + 1) exchanged between the lane halves -> 8 consecutive columns per lane
        if (g.Cb || (TWO && g.aux_out_bf16)) {
#pragma unroll
          for (int gp = 0; gp < 4; gp += 2) {
            const int n8 = nb + 8 * (gp + h);
            const bool ok = n8 < g.N;
            if (g.Cb) {
              unsigned a0 = pack2(v[4 * gp], v[4 * gp + 1]), a1 = pack2(v[4 * gp + 2], v[4 * gp + 3]);
              unsigned b0 = pack2(v[4 * gp + 4], v[4 * gp + 5]), b1 = pack2(v[4 * gp + 6], v[4 * gp + 7]);
              const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
              const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
              const u32x4_t o = {r0[0], r1[0], r0[1], r1[1]};
              __builtin_amdgcn_raw_buffer_store_b128(o, rsCb, ok ? (m * g.ldcb + n8) * 2 : OOB, 0, 0);
            }
            if (TWO && g.aux_out_bf16) {
              unsigned a0 = pack2(x2[4 * gp], x2[4 * gp + 1]), a1 = pack2(x2[4 * gp + 2], x2[4 * gp + 3]);
              unsigned b0 = pack2(x2[4 * gp + 4], x2[4 * gp + 5]), b1 = pack2(x2[4 * gp + 6], x2[4 * gp + 7]);
              const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
              const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
              const u32x4_t o = {r0[0], r1[0], r0[1], r1[1]};
              __builtin_amdgcn_raw_buffer_store_b128(o, rsX, ok ? (m * g.ld_aux + n8) * 2 : OOB, 0, 0);
            }
          }
        }
      }
    }
    if constexpr (XTR) {
      if (colsum && h == 0) {
        // every accumulator row of the ones-product holds the same sums: lanes 0..31 own column m of row block a
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int m = m0 + wm * 64 + a * 32 + i5;
          if (m < rt->colsum_M) {
            const float o = rt->colsum_out[m] + acc1[a][0];
            rt->colsum_out[m] = o;
            rt->ss = __builtin_fmaf(o, o, rt->ss);
          }
        }
      }
    }
  } else {
    // lane = output column: C += by fp32 atomics (two 128-byte row segments per wave-instruction)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int col = n0 + wn * 64 + b * 32 + i5;
        const int r0 = m0 + wm * 64 + a * 32 + 4 * h;
        int p_ = col < g.N ? (r0 * g.ldc + col) * 4 : OOB;
        const int stC = g.ldc * 4;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc[a][b][rr], rsC, p_, 0, 0);
          p_ += ((rr & 3) == 3 ? 5 : 1) * stC;
        }
      }
  }
#endif
}

template <int BM, int BN, bool AKM, bool BKM, bool SWAP, int ST, int EPI>
__global__ __launch_bounds__(64 * (BM / 64) * (BN / 64), (ST * (BM + BN) * 128 <= 80 * 1024 ? 2 : 1)) void gemm_dma_kernel(const GArgsD g) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[ST * (BM + BN) * 128];
  // work item -> (tile, k-piece), banded tile order inside an XCD's chunk
  const int w = xcd_work_item(g.tiles_m * g.tiles_n * g.nsplit);
  if (w < 0) return;
  set_wave_prio(g.prio);
  stamp_begin(g.stamp);
  gemm_dma_tile<BM, BN, AKM, BKM, SWAP, ST, EPI>(g, w, smem);
  stamp_end(g.stamp);
}

// The weight gradients of one encoder layer as ONE launch: up to four products dW_p[M_p, N_p] += A_p^T B_p (both
// operands k-major, K = rows of the batch) whose 128 x 128 tiles are numbered through.  A layer of UNITER-base has
// 144 + 144 + 108 + 36 = 432 tiles for the chip's 512 workgroup slots: every tile is owned by one workgroup over
// the whole K, so there are no partial sums to exchange (no atomics, no slabs) and dW += is a plain
// read-modify-write in the epilogue (UNITER_EPI_ADD with the output as its own aux operand).
struct GGroupD {
  GArgsD p[4];
  int start[5];     // first work item of product p; start[n..4] = total
  uniter_x3_riders_t x;   // riders (kernels instantiated with XTR only): include/uniter_hip.h
};

template <int ST, bool ACC, bool XTR = false>
__global__ __launch_bounds__(256, (ST == 2 ? 2 : 1)) void gemm_dma_wgrad_group_kernel(const GGroupD G) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[ST * 256 * 128];
  // riders (XTR, round 5): the column-reduction items of this workgroup first -- dealt from the END of the grid, where the
  // workgroups with one tile less sit -- then, per tile, the sum of squares of what it stored (and, for product 0's first
  // tile column, the column sums of its A operand)
  double wss = 0.0;
  if constexpr (XTR) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = (int)gridDim.x - 1 - (int)blockIdx.x; r < G.x.nred; r += (int)gridDim.x)
      wss += (double)riders_reduce_item(G.x, r, wave, lane);
  }
  // a grid smaller than the tile count (G.max_wgs): the workgroup walks its XCD's chunk in strides of the grid -- fewer
  // workgroups of this launch resident per CU, so the input-gradient chain on the other stream finds free slots
  bool any = false;
  for (int round = 0;; ++round) {
    const int w = xcd_work_item(G.start[4], round);
    if (w < 0) break;
    if (!any) stamp_begin(G.p[0].stamp);
    else __syncthreads();                 // the slowest wave has left the previous tile's last LDS stage
    any = true;
    const int p = (w >= G.start[1]) + (w >= G.start[2]) + (w >= G.start[3]);
    // ACC: dW += (the output is its own aux operand); !ACC: dW = (the first backward pass after an optimizer step that left
    // the gradient uncleared: no read of dW, and the optimizer wrote no zeros -- uniter_model_set_wgrad_overwrite)
    if constexpr (XTR) {
      TileRider rt;
      rt.colsum_out = p == 0 ? G.x.colsum_out : nullptr; rt.colsum_M = G.p[0].M; rt.ss = 0.f;
      gemm_dma_tile<128, 128, true, true, true, ST, ACC ? UNITER_EPI_ADD : UNITER_EPI_NONE, true>(G.p[p], w - G.start[p], smem, &rt);
      wss += (double)rt.ss;
    } else {
      gemm_dma_tile<128, 128, true, true, true, ST, ACC ? UNITER_EPI_ADD : UNITER_EPI_NONE>(G.p[p], w - G.start[p], smem);
    }
  }
  if constexpr (XTR) riders_store_ssq(G.x, wss, threadIdx.x >> 6, threadIdx.x & 63);
  if (any) stamp_end(G.p[0].stamp);
}

template <int BM>
void plan_tiles(GArgsD& g, int BN) {
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = (g.N + BN - 1) / BN;
  const long panel = (long)BM * g.K * 2;
  long bh = (3l << 19) / (panel > 0 ? panel : 1);
  g.band_h = (int)(bh < 1 ? 1 : (bh > 16 ? 16 : bh));
  if (g.band_h > g.tiles_m) g.band_h = g.tiles_m;
}

template <int BM, int BN, bool AKM, bool BKM, bool SWAP, int ST, int EPI>
int launch_d(GArgsD g, hipStream_t st) {
  plan_tiles<BM>(g, BN);
  const int nwork = g.tiles_m * g.tiles_n * g.nsplit;
  const int grid = (nwork + 7) / 8 * 8;
  hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, AKM, BKM, SWAP, ST, EPI>), dim3(grid), dim3(64 * (BM / 64) * (BN / 64)), 0, st, g);
  UCHECK_LAUNCH();
  return 0;
}

// weight gradients (both operands k-major): 128 x 128 tiles only
template <bool SWAP>
int dispatch_wgrad(int cfg, const GArgsD& g, hipStream_t st) {
  switch (cfg) {
    case 1: return launch_d<128, 128, true, true, SWAP, 2, UNITER_EPI_NONE>(g, st);
    case 4: return launch_d<128, 128, true, true, SWAP, 3, UNITER_EPI_NONE>(g, st);
    default: uniter_set_error("gemm_bf16v2: weight-gradient layout runs cfg 1 or 4 (got %d)", cfg); return UNITER_E_ARG;
  }
}

template <bool BKM, bool SWAP, int EPI>
int dispatch_cfg(int cfg, const GArgsD& g, hipStream_t st) {
  switch (cfg) {
    case 1: return launch_d<128, 128, false, BKM, SWAP, 2, EPI>(g, st);     // 4 waves, two workgroups per CU
    case 2: return launch_d<128, 256, false, BKM, SWAP, 3, EPI>(g, st);     // 8 waves, one workgroup per CU
    case 3: return launch_d<256, 128, false, BKM, SWAP, 3, EPI>(g, st);
    case 4: return launch_d<128, 128, false, BKM, SWAP, 3, EPI>(g, st);     // 4 waves, one workgroup per CU, deeper ring
    case 5: return launch_d<64, 128, false, BKM, SWAP, 2, EPI>(g, st);      // 2 waves: twice the tiles for the few-tile shapes (N = hidden)
    default: uniter_set_error("gemm_bf16v2: bad cfg %d (1..5)", cfg); return UNITER_E_ARG;
  }
}

template <bool BKM>
int dispatch_epi(int cfg, const GArgsD& g, hipStream_t st) {
  switch (g.epi) {
    case UNITER_EPI_NONE: return dispatch_cfg<BKM, true, UNITER_EPI_NONE>(cfg, g, st);
    case UNITER_EPI_BIAS: return dispatch_cfg<BKM, true, UNITER_EPI_BIAS>(cfg, g, st);
    case UNITER_EPI_BIAS_GELU: return dispatch_cfg<BKM, true, UNITER_EPI_BIAS_GELU>(cfg, g, st);
    case UNITER_EPI_DGELU: return dispatch_cfg<BKM, true, UNITER_EPI_DGELU>(cfg, g, st);
    case UNITER_EPI_ADD: return dispatch_cfg<BKM, true, UNITER_EPI_ADD>(cfg, g, st);
    case UNITER_EPI_BIAS_GELU_D: return dispatch_cfg<BKM, true, UNITER_EPI_BIAS_GELU_D>(cfg, g, st);
    case UNITER_EPI_MUL: return dispatch_cfg<BKM, true, UNITER_EPI_MUL>(cfg, g, st);
    default: uniter_set_error("gemm_bf16v2: bad epilogue %d", g.epi); return UNITER_E_ARG;
  }
}

}  // namespace

// the persistent loader / compute kernels (gemm_bf16_p.hip, round 6): cfg 6 = 128 x 128 tiles, 7 = 128 x 256, 8 = 128 x 192
int gemm_b1p_run(int cfg, int nsplit, int b_kmajor, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                 float* C, int ldc, long c_split_stride, void* Cb, int ldcb, int epilogue, const float* bias,
                 const void* aux_in, int aux_in_bf16, void* aux_out, int aux_out_bf16, int ld_aux, float* colpart, void* stream);
int gemm_b1p_wgrad_group(int n, const int* Mo, const int* No, int K, const void* const* A, const void* const* B, float* const* dW,
                         void* stream, int overwrite, int max_wgs, uniter_x3_riders_t* riders);
int gemm_b1p_wgrad_group_slots(int n, const int* Mo, const int* No, int max_wgs);

// cfg: 1 = 128x128 (2 stages), 2 = 128x256, 3 = 256x128, 4 = 128x128 (3 stages), 5 = 64x128 (2 waves); 0 = choose;
// 6 / 7 / 8 = the persistent loader / compute kernels of gemm_bf16_p.hip (128 x 128 / 128 x 256 / 128 x 192 tiles).
// beta = 1: C += A.B through fp32 atomics (lane = column orientation; no epilogue, no bf16 output).
int gemm_bf16v2_run(int cfg, int nsplit, int a_kmajor, int b_kmajor, int M, int N, int K, const void* A, int lda,
                    const void* B, int ldb, float* C, int ldc, long c_split_stride, void* Cb, int ldcb, int epilogue,
                    const float* bias, const void* aux_in, int aux_in_bf16, void* aux_out, int aux_out_bf16,
                    int ld_aux, int beta, void* stream) {
  UCHECK_ARG(M > 0 && N > 0 && K > 0 && A && B && (C || Cb), "gemm_bf16v2: bad argument");
  UCHECK_ARG(epilogue >= 0 && epilogue <= UNITER_EPI_MUL, "gemm_bf16v2: bad epilogue %d", epilogue);
  UCHECK_ARG(!a_kmajor || (b_kmajor && epilogue == UNITER_EPI_NONE && !Cb),
             "gemm_bf16v2: A k-major only as the weight-gradient layout (both operands k-major, no epilogue, fp32 output)");
  UCHECK_ARG(nsplit >= 1 && nsplit <= 8 && (nsplit == 1 || (C && !Cb && !beta && c_split_stride >= (long)M * ldc)),
             "gemm_bf16v2: split-K needs fp32 slabs (no bf16 output, no accumulate)");
  UCHECK_ARG(!beta || (C && !Cb && epilogue == UNITER_EPI_NONE), "gemm_bf16v2: C += has no epilogue and no bf16 copy");
  UCHECK_SHAPE((K % KT == 0 || (a_kmajor && b_kmajor)) && lda % 8 == 0 && ldb % 8 == 0 && N % 8 == 0 && (!a_kmajor || M % 8 == 0) &&
               ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 && (ldc % 4 == 0) && (ldcb % 8 == 0) &&
               (ld_aux % 4 == 0) && ((uintptr_t)C & 15) == 0 && ((uintptr_t)Cb & 15) == 0 &&
               ((uintptr_t)aux_in & 15) == 0 && ((uintptr_t)aux_out & 15) == 0 && ((uintptr_t)bias & 15) == 0,
               "gemm_bf16v2: K %% 64, N %% 8, leading dimensions %% 8 (bf16) / %% 4 (fp32) and 16-byte aligned "
               "buffers required (M=%d N=%d K=%d)", M, N, K);
  UCHECK_SHAPE((size_t)(a_kmajor ? K : M + 256) * lda * 2 < (1ull << 31) && (size_t)(b_kmajor ? K + 64 : N + 256) * ldb * 2 < (1ull << 31) &&
               ((size_t)M + 256) * (ldc > 0 ? ldc : 1) * 4 < (1ull << 31) &&
               ((size_t)M + 256) * (ld_aux > 0 ? ld_aux : 1) * 4 < (1ull << 31), "gemm_bf16v2: operand beyond 31-bit offsets");
  if ((cfg & 0xff) >= 6) {
    UCHECK_ARG(!a_kmajor && !beta, "gemm_bf16v2: cfg %d (persistent kernels) runs forward / input-gradient layouts without accumulate "
               "(weight gradients: uniter_wgrad_bf16_group, cfg 7)", cfg & 0xff);
    return gemm_b1p_run(cfg & 0xff, nsplit, b_kmajor, M, N, K, A, lda, B, ldb, C, ldc, c_split_stride, Cb, ldcb, epilogue, bias, aux_in,
                        aux_in_bf16, aux_out, aux_out_bf16, ld_aux, nullptr, stream);
  }
  GArgsD g;
  g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.c_split_stride = c_split_stride; g.Cb = (unsigned short*)Cb; g.ldcb = ldcb; g.epi = epilogue; g.bias = bias;
  g.aux_in = aux_in; g.aux_in_bf16 = aux_in_bf16; g.aux_out = aux_out; g.aux_out_bf16 = aux_out_bf16; g.ld_aux = ld_aux;
  g.tiles_m = g.tiles_n = 0; g.band_h = 1; g.nsplit = nsplit;
  g.dbg = cfg >> 8; cfg &= 0xff;
  g.stamp = take_stamp_slot();
  g.prio = take_launch_prio();
  if (cfg == 0) {
    // few tiles and one k-piece (the attention-output products: 126 tiles of 128 x 128 for 256 CUs): 64 x 128 tiles,
    // twice the workgroups (12.0 -> 10.6 us forward, 11.7 -> 10.0 us input gradient; profiles/r02_gemm_bf16_v2.txt)
    const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
    cfg = (!a_kmajor && nsplit == 1 && tiles <= 160) ? 5 : 1;
  }
  hipStream_t st = (hipStream_t)stream;
  if (a_kmajor) return beta ? dispatch_wgrad<false>(cfg, g, st) : dispatch_wgrad<true>(cfg, g, st);
  if (beta) return b_kmajor ? dispatch_cfg<true, false, UNITER_EPI_NONE>(cfg, g, st) : dispatch_cfg<false, false, UNITER_EPI_NONE>(cfg, g, st);
  return b_kmajor ? dispatch_epi<true>(cfg, g, st) : dispatch_epi<false>(cfg, g, st);
}

// dW_p[M_p, N_p] += A_p^T B_p for up to four products of one reduction length K (A_p [K, M_p], B_p [K, N_p] bf16,
// dW_p fp32 with leading dimension N_p), one launch; cfg 1 = two LDS stages (two workgroups per CU), 4 = three.
// workgroups of the grouped launch over `total` tiles: a multiple of 8, capped (max_wgs > 0: by the caller; else
// UNITER_WGRAD_GROUP_WGS, default 256 = one workgroup per CU walking its tiles)
int gemm_chip_cus();
static int wgrad_group_grid(int total, int max_wgs) {
  int grid = (total + 7) / 8 * 8;
  // UNITER_WGRAD_GROUP_WGS: cap of the grid (a multiple of 8; 0 = one workgroup per tile).  Default 256 = one workgroup of this
  // launch per CU, walking its tiles: the launch alone takes what two co-resident workgroups take (a lone 4-wave tile runs
  // its k-loop at 0.57 us per k-tile against 0.93 for two), and every CU keeps a slot for the other stream's kernels.
  // Measured again once the attention backward took its CU in one launch (four same-box rounds): UNITER-base 4.75 -> 4.71 ms,
  // UNITER-large 9.65 -> 9.49, config 5 at B = 32 8.19 -> 8.00 ms per step; 192 / 320 / 384 workgroups 4.77-4.80, 128 5.13
  static const int cap_env = [] { const char* e = getenv("UNITER_WGRAD_GROUP_WGS"); return e ? atoi(e) / 8 * 8 : 256; }();
  int cap = max_wgs >= 8 ? max_wgs / 8 * 8 : cap_env;
  if (g_uniter_cu_reserve > 0 && cap >= 8) {              // CUs left to the data-parallel exchange's kernels (common.h)
    const int room = (gemm_chip_cus() - g_uniter_cu_reserve) / 8 * 8;      // (the device's CU count: gemm_split3.hip)
    if (room >= 8 && cap > room) cap = room;
  }
  if (cap >= 8 && grid > cap) grid = cap;
  return grid;
}
static int wgrad_group_tiles(int n, const int* Mo, const int* No) {
  int total = 0;
  for (int p = 0; p < n; ++p) total += ((Mo[p] + 127) / 128) * ((No[p] + 127) / 128);
  return total;
}
// sum-of-squares slots a launch with riders writes (4 per workgroup)
int gemm_bf16v2_wgrad_group_slots(int n, const int* Mo, const int* No, int max_wgs) {
  if (!Mo || !No || n < 1 || n > 4) return 0;
  return 4 * wgrad_group_grid(wgrad_group_tiles(n, Mo, No), max_wgs);
}
// the smallest grid on which the tiles take no more rounds than on one workgroup per CU
int gemm_bf16v2_wgrad_group_balanced_wgs(int n, const int* Mo, const int* No) {
  if (!Mo || !No || n < 1 || n > 4) return 0;
  const int total = wgrad_group_tiles(n, Mo, No);
  const int full = wgrad_group_grid(total, 0);
  const int rounds = (total + full - 1) / full;
  const int wgs = ((total + rounds - 1) / rounds + 7) / 8 * 8;
  return wgs < full ? wgs : full;
}

int gemm_bf16v2_wgrad_group(int cfg, int n, const int* Mo, const int* No, int K, const void* const* A,
                            const void* const* B, float* const* dW, void* stream, int overwrite, int max_wgs,
                            uniter_x3_riders_t* riders) {
  UCHECK_ARG(n >= 1 && n <= 4 && K > 0 && Mo && No && A && B && dW, "wgrad_group: bad argument");
  if (cfg == 7) return gemm_b1p_wgrad_group(n, Mo, No, K, A, B, dW, stream, overwrite, max_wgs, riders);      // 128 x 256 tiles, persistent
  GGroupD G;
  memset(&G.x, 0, sizeof(G.x));
  unsigned long long* stamp = take_stamp_slot();
  int total = 0;
  for (int p = 0; p < 4; ++p) {
    G.start[p] = total;
    if (p >= n) { G.p[p] = G.p[0]; continue; }
    UCHECK_ARG(Mo[p] > 0 && No[p] > 0 && A[p] && B[p] && dW[p], "wgrad_group: bad product %d", p);
    UCHECK_SHAPE(Mo[p] % 8 == 0 && No[p] % 8 == 0 && ((uintptr_t)A[p] & 15) == 0 && ((uintptr_t)B[p] & 15) == 0 &&
                 ((uintptr_t)dW[p] & 15) == 0 && (size_t)(K + 64) * (Mo[p] > No[p] ? Mo[p] : No[p]) * 2 < (1ull << 31) &&
                 ((size_t)Mo[p] + 256) * No[p] * 4 < (1ull << 31),
                 "wgrad_group: M, N %% 8, 16-byte aligned buffers, 31-bit offsets (product %d: %d x %d, K=%d)", p, Mo[p], No[p], K);
    GArgsD& g = G.p[p];
    g.M = Mo[p]; g.N = No[p]; g.K = K; g.A = A[p]; g.lda = Mo[p]; g.B = B[p]; g.ldb = No[p]; g.C = dW[p]; g.ldc = No[p];
    g.c_split_stride = 0; g.Cb = nullptr; g.ldcb = 0; g.epi = overwrite ? UNITER_EPI_NONE : UNITER_EPI_ADD; g.bias = nullptr;
    g.aux_in = overwrite ? nullptr : dW[p]; g.aux_in_bf16 = 0; g.aux_out = nullptr; g.aux_out_bf16 = 0; g.ld_aux = No[p];
    g.nsplit = 1; g.stamp = stamp; g.dbg = 0; g.prio = 0;
    plan_tiles<128>(g, 128);
    total += g.tiles_m * g.tiles_n;
  }
  G.start[4] = total;
  for (int p = n; p < 4; ++p) G.start[p] = total;
  const int grid = wgrad_group_grid(total, max_wgs);
  hipStream_t st = (hipStream_t)stream;
  if (riders) {
    UCHECK_ARG(cfg != 4, "wgrad_group: riders ride on the two-stage form (cfg 1)");
    riders->grid = grid;
    UCHECK_RC(riders_prepare(*riders, "wgrad_bf16_group"));
    G.x = *riders;
    if (overwrite) hipLaunchKernelGGL((gemm_dma_wgrad_group_kernel<2, false, true>), dim3(grid), dim3(256), 0, st, G);
    else hipLaunchKernelGGL((gemm_dma_wgrad_group_kernel<2, true, true>), dim3(grid), dim3(256), 0, st, G);
    UCHECK_LAUNCH();
    return 0;
  }
  if (cfg == 4) {
    if (overwrite) hipLaunchKernelGGL((gemm_dma_wgrad_group_kernel<3, false>), dim3(grid), dim3(256), 0, st, G);
    else hipLaunchKernelGGL((gemm_dma_wgrad_group_kernel<3, true>), dim3(grid), dim3(256), 0, st, G);
  } else {
    if (overwrite) hipLaunchKernelGGL((gemm_dma_wgrad_group_kernel<2, false>), dim3(grid), dim3(256), 0, st, G);
    else hipLaunchKernelGGL((gemm_dma_wgrad_group_kernel<2, true>), dim3(grid), dim3(256), 0, st, G);
  }
  UCHECK_LAUNCH();
  return 0;
}

extern "C" int uniter_wgrad_bf16_group(int cfg, int n, const int* M, const int* N, int K, const void* const* A,
                                       const void* const* B, float* const* dW, void* stream) {
  return gemm_bf16v2_wgrad_group(cfg, n, M, N, K, A, B, dW, stream, 0, 0, nullptr);
}

extern "C" int uniter_wgrad_bf16_group_riders(int cfg, int n, const int* M, const int* N, int K, const void* const* A,
                                              const void* const* B, float* const* dW, int overwrite, int max_wgs,
                                              uniter_x3_riders_t* riders, void* stream) {
  return gemm_bf16v2_wgrad_group(cfg, n, M, N, K, A, B, dW, stream, overwrite, max_wgs, riders);
}
extern "C" int uniter_wgrad_bf16_group_slots(int n, const int* M, const int* N, int max_wgs) {
  return gemm_bf16v2_wgrad_group_slots(n, M, N, max_wgs);
}
// the same for a given geometry of the grouped launch: cfg 7 (persistent 128 x 256 tiles, gemm_bf16_p.hip) writes 8 slots per workgroup
extern "C" int uniter_wgrad_bf16_group_slots_cfg(int cfg, int n, const int* M, const int* N, int max_wgs) {
  return cfg == 7 ? gemm_b1p_wgrad_group_slots(n, M, N, max_wgs) : gemm_bf16v2_wgrad_group_slots(n, M, N, max_wgs);
}

// Split-K choice for the GEMMs whose N is the hidden size (measured on MI355X, tests/tools/gemm_v2_lab.py,
// profiles/r02_gemm_bf16_v2.txt): with 128 x 128 tiles M = 2624, N = 768 gives 126 tiles for 256 CUs x 2 resident
// workgroups; two k-pieces per tile fill the chip (FFN-down 33.7 -> 22.3 us).  Four pieces are faster still in the
// GEMM (21.0 us) but every extra slab costs the consumer an 8 MB read, so four only where two leave half the slots
// empty.  Pieces shorter than 12 k-tiles do not pay for their prologue.
int gemm_bf16v2_pick_split(int M, int N, int K) {
  const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
  const int nk = (K + KT - 1) / KT;
  if (tiles <= 128 && nk >= 48) return 4;
  if (tiles <= 320 && nk >= 24) return 2;
  return 1;
}

// Weight gradients dW[M, N] += dY^T X (both operands k-major, K = rows of the batch): pieces of the split-K slab form,
// or 0 = keep the stream-K + float-atomics kernel of gemm_bf16.hip.  Measured at K = 2624
// (profiles/r02_gemm_bf16_v2.txt): 768 x 3072 39.2 -> 30.9 us with two pieces, 2304 x 768 26.0 -> 24.7 (two),
// 768 x 768 17.9 -> 16.8 (four); 3072 x 768 stays on stream-K (29.0 vs 31.2).
// In the training step the gain does not survive (2870 -> 2855 samples/s: the weight gradients share the chip with the
// input-gradient chain, whose idle slots the perfectly balanced stream-K pieces fill better), so the slab form is an
// opt-in (UNITER_WGRAD_SLABS=1) and the default stays stream-K.
int gemm_bf16v2_wgrad_pieces(int M, int N, int K) {
  static const bool on = [] { const char* e = getenv("UNITER_WGRAD_SLABS"); return e && e[0] == '1'; }();
  if (!on || K < 1024 || M % 8 || N % 8) return 0;
  const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
  if (tiles <= 48) return 4;
  if (M <= N || tiles <= 128) return 2;
  return 0;
}

extern "C" int uniter_gemm_bf16v2_cfg(int cfg, int nsplit, int a_kmajor, int b_kmajor, int M, int N, int K,
                                      const void* A, int lda, const void* B, int ldb, float* C, int ldc,
                                      long c_split_stride, void* C_bf16, int ldcb, int epilogue, const float* bias,
                                      const void* aux_in, int aux_in_bf16, void* aux_out, int aux_out_bf16,
                                      int ld_aux, int beta, void* stream) {
  UCHECK_ARG(!(epilogue == UNITER_EPI_BIAS || epilogue == UNITER_EPI_BIAS_GELU || epilogue == UNITER_EPI_BIAS_GELU_D) || bias,
             "gemm_bf16v2: epilogue needs bias");
  UCHECK_ARG(!(epilogue == UNITER_EPI_DGELU || epilogue == UNITER_EPI_ADD || epilogue == UNITER_EPI_MUL) || aux_in,
             "gemm_bf16v2: epilogue needs aux_in");
  UCHECK_ARG(!(epilogue == UNITER_EPI_BIAS_GELU || epilogue == UNITER_EPI_BIAS_GELU_D) || aux_out,
             "gemm_bf16v2: epilogue needs aux_out");
  return gemm_bf16v2_run(cfg, nsplit, a_kmajor, b_kmajor, M, N, K, A, lda, B, ldb, C, ldc, c_split_stride, C_bf16, ldcb,
                         epilogue, bias, aux_in, aux_in_bf16, aux_out, aux_out_bf16, ld_aux, beta, stream);
}
