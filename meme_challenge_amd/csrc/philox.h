// Counter-based dropout masks: Philox4x32-10.  Specification: oracle/philox.py
//   r = philox(counter = (lo32(e>>2), hi32(e>>2), site, offset), key = (lo32(seed), hi32(seed)))[e & 3]
//   keep(e) = r >= floor(p * 2^32)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// dropout-site ids (mirrors oracle/philox.py)
#define SITE_TXT_EMB 0u
#define SITE_IMG_EMB 1u
#define SITE_ATTN_PROBS(l) (2u + 4u * (uint32_t)(l))
#define SITE_ATTN_OUT(l)   (3u + 4u * (uint32_t)(l))
#define SITE_FFN_OUT(l)    (4u + 4u * (uint32_t)(l))

struct DropCfg {
  uint32_t k0, k1;       // seed lo/hi
  uint32_t site, offset;
  uint32_t thresh;       // keep iff r >= thresh
  float scale;           // 1/(1-p)
  int active;
  // keep flags drawn ahead (round 5: uniter_hidden_keep_bits_gen): one nibble per 4-element group of the site's index space, two groups
  // per byte; NULL = draw them here (ten Philox rounds per group)
  const unsigned char* bits;
};

static inline DropCfg make_drop(float p, uint64_t seed, uint32_t offset, uint32_t site) {
  DropCfg d;
  d.k0 = (uint32_t)(seed & 0xffffffffu);
  d.k1 = (uint32_t)(seed >> 32);
  d.site = site;
  d.offset = offset;
  d.active = p > 0.0f;
  double t = (double)p * 4294967296.0;
  d.thresh = t >= 4294967295.0 ? 0xffffffffu : (t <= 0.0 ? 0u : (uint32_t)t);
  d.scale = d.active ? 1.0f / (1.0f - p) : 1.0f;
  d.bits = nullptr;
  return d;
}

#ifdef __HIPCC__
struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return u32x4{c0, c1, c2, c3};
}

// 4 random words for elements 4*g .. 4*g+3 of the site's index space
__device__ __forceinline__ u32x4 drop_words(const DropCfg& d, uint64_t group) {
  return philox4x32_10((uint32_t)group, (uint32_t)(group >> 32), d.site, d.offset, d.k0, d.k1);
}
// multipliers (0 or scale) for the 4 elements of a group
__device__ __forceinline__ void drop_mult4(const DropCfg& d, uint64_t group, float m[4]) {
  if (d.bits) {      // (uniform per launch)
    const unsigned nib = (unsigned)d.bits[group >> 1] >> (4u * (unsigned)(group & 1));
    m[0] = (nib & 1u) ? d.scale : 0.0f;
    m[1] = (nib & 2u) ? d.scale : 0.0f;
    m[2] = (nib & 4u) ? d.scale : 0.0f;
    m[3] = (nib & 8u) ? d.scale : 0.0f;
    return;
  }
  const u32x4 r = drop_words(d, group);
  m[0] = r.x >= d.thresh ? d.scale : 0.0f;
  m[1] = r.y >= d.thresh ? d.scale : 0.0f;
  m[2] = r.z >= d.thresh ? d.scale : 0.0f;
  m[3] = r.w >= d.thresh ? d.scale : 0.0f;
}
// keep flags of the group's 4 elements as a 4-bit mask (bit t = element t kept)
__device__ __forceinline__ unsigned drop_bits4(const DropCfg& d, uint64_t group) {
  const u32x4 r = drop_words(d, group);
  return (r.x >= d.thresh ? 1u : 0u) | (r.y >= d.thresh ? 2u : 0u) | (r.z >= d.thresh ? 4u : 0u) | (r.w >= d.thresh ? 8u : 0u);
}
#endif
