// Shared device/host helpers for libneurst_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/neurst_hip.h"

typedef unsigned short bf16_t;  // raw bf16 bits

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(4))) float floatx4_t;

#define NST_WAVE 64

// ---------------------------------------------------------------------------
// error plumbing: never abort, never throw across the ABI
// ---------------------------------------------------------------------------
void nst_set_error(const char* fmt, ...);

#define NST_CHECK_ARG(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      nst_set_error(__VA_ARGS__);                \
      return NST_ERR_INVALID_ARG;                \
    }                                            \
  } while (0)

#define NST_CHECK_LAUNCH(name)                                                    \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) {                                                      \
      nst_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));       \
      return NST_ERR_LAUNCH;                                                      \
    }                                                                             \
  } while (0)

#define NST_CHECK_HIP(expr)                                                       \
  do {                                                                            \
    hipError_t e__ = (expr);                                                      \
    if (e__ != hipSuccess) {                                                      \
      nst_set_error("%s failed: %s", #expr, hipGetErrorString(e__));              \
      return NST_ERR_LAUNCH;                                                      \
    }                                                                             \
  } while (0)

// Device scalar every dropout-drawing kernel ADDS to its seed argument when it runs (nst_dropout_seed_offset_set / _add):
// a captured HIP graph then draws new masks at every replay.  Allocated once per process (8 bytes), zero by default.
const uint64_t* nst_seed_offset_devptr();

static inline int nst_dtype_size(int dt) { return dt == NST_BF16 ? 2 : 4; }
static inline bool nst_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---------------------------------------------------------------------------
// bf16 <-> f32 (round to nearest even)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// gfx950 converts in hardware (v_cvt_pk_bf16_f32, round-to-nearest-even): one instruction per PAIR of values
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_hw_t;
typedef __attribute__((ext_vector_type(2))) float floatx2_hw_t;
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const floatx2_hw_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_hw_t));
}

template <typename T>
__device__ __forceinline__ float to_f32(T v);
template <>
__device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return bf16_to_f32(v); }

template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return f32_to_bf16(v); }

// ---------------------------------------------------------------------------
// wavefront (64-lane) reductions by shuffle
// ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Fast all-lanes sum: 4 DPP steps reduce each 16-lane row in place (no LDS crossbar: __shfl_xor lowers to
// ds_bpermute, ~100 cycles of dependent latency per step), then the 4 row totals are read with v_readlane.
// The result is wave-uniform (lives in an SGPR).
__device__ __forceinline__ float dpp_add(float v, const int ctrl) {
  switch (ctrl) {  // dpp_ctrl must be an immediate
    case 0: return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    case 1: return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    case 2: return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));  // row_half_mirror
    default: return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true)); // row_mirror
  }
}
__device__ __forceinline__ float wave_sum_fast(float v) {
  v = dpp_add(v, 0);
  v = dpp_add(v, 1);
  v = dpp_add(v, 2);
  v = dpp_add(v, 3);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// seed + *offset read through the scalar cache (never a vector load: a pending VMEM load would make the compiler drain
// the LDS-DMA prefetch of the GEMM kernels in front of it)
__device__ __forceinline__ uint64_t seed_with_offset(uint64_t seed, const uint64_t* offset_dev) {
  uint64_t off;
  asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(off) : "s"(offset_dev) : "memory");
  return seed + off;
}

// ---------------------------------------------------------------------------
// Philox4x32-7 counter RNG for dropout (7 rounds pass BigCrush; the 10-round default only adds safety margin).  One call yields 4 x u32 for the
// element group (idx/4); element idx uses word idx%4.  The same
// (seed, stream, idx) triple regenerates the same mask in backward.
// ---------------------------------------------------------------------------
struct Philox4 {
  uint32_t x, y, z, w;
};
__device__ __forceinline__ Philox4 philox4x32_10(uint64_t seed, uint64_t stream, uint64_t ctr) {
  uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = (uint32_t)stream, c3 = (uint32_t)(stream >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    // one 32x32 -> 64 multiply per product (v_mad_u64_u32) instead of a mul_hi / mul_lo pair
    const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  Philox4 o; o.x = c0; o.y = c1; o.z = c2; o.w = c3;
  return o;
}
// Dropout masks outside attention use 16-bit fields: one Philox call (4 x u32 = 8 x u16) serves the element group
// idx/8; element idx compares field idx%8 (word (idx%8)/2, low half first) against thresh16 = round(p * 65536).
// The drop probability is therefore quantised to 1/65536 and inv_keep = 65536/(65536 - thresh16) keeps the mask
// exactly unbiased.  keep-multiplier: 0 (dropped) or inv_keep.
__device__ __forceinline__ float drop_field(uint32_t word, int half, uint32_t thresh16, float inv_keep) {
  const uint32_t f = half ? (word >> 16) : (word & 0xffffu);
  return f >= thresh16 ? inv_keep : 0.0f;
}
__device__ __forceinline__ float dropout_keep_scale(uint64_t seed, uint64_t stream, uint64_t idx, uint32_t thresh16,
                                                    float inv_keep) {
  Philox4 r = philox4x32_10(seed, stream, idx >> 3);
  const uint32_t sel = (uint32_t)(idx & 7), ws = sel >> 1;
  const uint32_t w = ws == 0 ? r.x : (ws == 1 ? r.y : (ws == 2 ? r.z : r.w));
  return drop_field(w, sel & 1, thresh16, inv_keep);
}
// 8 consecutive elements starting at idx (multiple of 8): one Philox call
__device__ __forceinline__ void dropout_keep8(uint64_t seed, uint64_t stream, uint64_t idx, uint32_t thresh16,
                                              float inv_keep, float (&m)[8]) {
  const Philox4 r = philox4x32_10(seed, stream, idx >> 3);
  const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int j = 0; j < 8; ++j) m[j] = drop_field(w[j >> 1], j & 1, thresh16, inv_keep);
}
// 4 consecutive elements starting at idx (multiple of 4): the lower or upper half of one Philox call
__device__ __forceinline__ void dropout_keep4(uint64_t seed, uint64_t stream, uint64_t idx, uint32_t thresh16,
                                              float inv_keep, float (&m)[4]) {
  const Philox4 r = philox4x32_10(seed, stream, idx >> 3);
  const bool hi = (idx >> 2) & 1;
  const uint32_t w0 = hi ? r.z : r.x, w1 = hi ? r.w : r.y;
  m[0] = drop_field(w0, 0, thresh16, inv_keep);
  m[1] = drop_field(w0, 1, thresh16, inv_keep);
  m[2] = drop_field(w1, 0, thresh16, inv_keep);
  m[3] = drop_field(w1, 1, thresh16, inv_keep);
}
// host: p -> (thresh16, inv_keep)
static inline void nst_dropout_params16(float p, uint32_t* thresh16, float* inv_keep) {
  double t = (double)p * 65536.0 + 0.5;
  if (t < 0) t = 0;
  if (t > 65535.0) t = 65535.0;
  *thresh16 = (uint32_t)t;
  *inv_keep = (float)(65536.0 / (65536.0 - (double)*thresh16));
}

