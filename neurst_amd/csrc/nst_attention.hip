// Fused scaled-dot-product attention, forward and backward (flash style: the [B,H,Tq,Tk] probability
// matrix is never written to HBM; only the per-row log-sum-exp is saved).
//   replaces q*=dh^-0.5 ; einsum(BTNH,BFNH->BNFT) ; +bias ; softmax ; dropout ; einsum(BNFT,BTNH->BFNH)
//   of neurst/layers/attentions/multi_head_attention.py:124-164, 203-215 and their TF gradients.
//
// One workgroup (4 waves) per (64-row tile, head, batch); each wave owns 16 rows.  K/V (or Q/dO) tiles of
// 64 rows x 64 head-dim are staged in LDS, row-major with a 16-byte pad; a tile serves both as a
// reduction-contiguous MFMA operand (ds_read_b128) and as a reduction-major one (ds_read_b64_tr_b16).
// Softmax statistics live in registers: a 16x16 MFMA C fragment keeps row (lane>>4)*4+reg, column lane&15,
// so row reductions are shuffles inside 16-lane groups.  P is handed from C layout to A layout through a
// wave-private LDS tile.  Additive masks follow the reference exactly (bias added in fp32, FLOAT_MIN not -inf),
// keys beyond Tk are excluded.  Dropout mask = Philox(seed, stream_id, ((b*H+h)*Tq+q)*Tk+k), regenerated in bwd.
#include "nst_gemm_core.h"

#include <stdlib.h>

using nstgemm::Mma;

namespace {

constexpr int DH = 64;   // padded head dim (dh <= 64)
constexpr int TR = 64;   // tile rows (queries or keys)

template <typename T>
struct AT {
  static constexpr int RS = DH * (int)sizeof(T) + 16;  // LDS row stride in bytes
  static constexpr int E = 16 / (int)sizeof(T);
  static constexpr int CPR = DH * (int)sizeof(T) / 16;  // 16-byte chunks per row
  static constexpr int KS = Mma<T>::KS;
  static constexpr int NK = DH / KS;                    // MFMA steps over a 64-long reduction
  static constexpr int TILE_BYTES = TR * RS;
};

template <typename T>
struct FragT;
template <>
struct FragT<bf16_t> { typedef bf16x8_t type; };
template <>
struct FragT<float> { typedef float type; };

// reduction-contiguous fragment: rows row0+(l&15), reduction kk + ...
template <typename T>
__device__ __forceinline__ typename FragT<T>::type rc_frag(const char* tile, int row0, int kk, int lane);
template <>
__device__ __forceinline__ bf16x8_t rc_frag<bf16_t>(const char* tile, int row0, int kk, int lane) {
  return *reinterpret_cast<const bf16x8_t*>(tile + (row0 + (lane & 15)) * AT<bf16_t>::RS + (kk + (lane >> 4) * 8) * 2);
}
template <>
__device__ __forceinline__ float rc_frag<float>(const char* tile, int row0, int kk, int lane) {
  return *reinterpret_cast<const float*>(tile + (row0 + (lane & 15)) * AT<float>::RS + (kk + (lane >> 4)) * 4);
}
// reduction-major fragment: tile is [reduction rows][output cols]; output col col0+(l&15)
template <typename T, bool USE_TR>
__device__ __forceinline__ typename FragT<T>::type oc_frag(const char* tile, int col0, int kk, int lane);
template <>
__device__ __forceinline__ bf16x8_t oc_frag<bf16_t, true>(const char* tile, int col0, int kk, int lane) {
  const int ii = lane & 15;
  const char* p = tile + (kk + (lane >> 4) * 8 + (ii >> 2)) * AT<bf16_t>::RS + (col0 + (ii & 3) * 4) * 2;
  typedef short4_t __attribute__((address_space(3))) * lds_ptr_t;
  short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(p));
  short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(p + 4 * AT<bf16_t>::RS));
  union { short s[8]; bf16x8_t f; } u;
  u.s[0] = lo[0]; u.s[1] = lo[1]; u.s[2] = lo[2]; u.s[3] = lo[3];
  u.s[4] = hi[0]; u.s[5] = hi[1]; u.s[6] = hi[2]; u.s[7] = hi[3];
  return u.f;
}
template <>
__device__ __forceinline__ bf16x8_t oc_frag<bf16_t, false>(const char* tile, int col0, int kk, int lane) {
  const char* p = tile + (kk + (lane >> 4) * 8) * AT<bf16_t>::RS + (col0 + (lane & 15)) * 2;
  union { short s[8]; bf16x8_t f; } u;
#pragma unroll
  for (int j = 0; j < 8; ++j) u.s[j] = *reinterpret_cast<const short*>(p + j * AT<bf16_t>::RS);
  return u.f;
}
template <>
__device__ __forceinline__ float oc_frag<float, true>(const char* tile, int col0, int kk, int lane) {
  return *reinterpret_cast<const float*>(tile + (kk + (lane >> 4)) * AT<float>::RS + (col0 + (lane & 15)) * 4);
}
template <>
__device__ __forceinline__ float oc_frag<float, false>(const char* tile, int col0, int kk, int lane) {
  return oc_frag<float, true>(tile, col0, kk, lane);
}

// stage rows [row0, row0+64) x cols [0, DH) of a [nrows, dh] matrix (row stride ld elements) into LDS, zero filled
template <typename T>
__device__ __forceinline__ void stage_tile(char* lds, const T* __restrict__ base, int64_t ld, int row0, int nrows, int dh,
                                           int vec, int tid) {
  constexpr int CH = TR * AT<T>::CPR;
#pragma unroll
  for (int c = tid; c < CH; c += 256) {
    const int row = c / AT<T>::CPR, cc = c % AT<T>::CPR;
    const int col = cc * AT<T>::E;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row0 + row < nrows && col < dh) {
      const T* p = base + (int64_t)(row0 + row) * ld + col;
      if (vec) {
        v = *reinterpret_cast<const uint4*>(p);
      } else {
        T tmp[AT<T>::E];
#pragma unroll
        for (int e = 0; e < AT<T>::E; ++e) tmp[e] = (col + e < dh) ? p[e] : (T)0;
        memcpy(&v, tmp, 16);
      }
    }
    *reinterpret_cast<uint4*>(lds + row * AT<T>::RS + cc * 16) = v;
  }
}

// register-staged variant: issue the global loads of a tile early (tile_load), write them to LDS later (tile_store),
// so the HBM/L2 latency of tile t+1 hides under the MFMAs of tile t.
template <typename T>
struct TileRegs {
  uint4 v[TR * AT<T>::CPR / 256];
};
template <typename T>
__device__ __forceinline__ void tile_load(TileRegs<T>& regs, const T* __restrict__ base, int64_t ld, int row0, int nrows, int dh,
                                          int vec, int tid) {
  constexpr int N = TR * AT<T>::CPR / 256;
#pragma unroll
  for (int s = 0; s < N; ++s) {
    const int c = tid + s * 256;
    const int row = c / AT<T>::CPR, cc = c % AT<T>::CPR;
    const int col = cc * AT<T>::E;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row0 + row < nrows && col < dh) {
      const T* p = base + (int64_t)(row0 + row) * ld + col;
      if (vec) {
        v = *reinterpret_cast<const uint4*>(p);
      } else {
        T tmp[AT<T>::E];
#pragma unroll
        for (int e = 0; e < AT<T>::E; ++e) tmp[e] = (col + e < dh) ? p[e] : (T)0;
        memcpy(&v, tmp, 16);
      }
    }
    regs.v[s] = v;
  }
}
template <typename T>
__device__ __forceinline__ void tile_store(char* lds, const TileRegs<T>& regs, int tid) {
  constexpr int N = TR * AT<T>::CPR / 256;
#pragma unroll
  for (int s = 0; s < N; ++s) {
    const int c = tid + s * 256;
    const int row = c / AT<T>::CPR, cc = c % AT<T>::CPR;
    *reinterpret_cast<uint4*>(lds + row * AT<T>::RS + cc * 16) = regs.v[s];
  }
}

__device__ __forceinline__ float group16_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 1, 64)); v = fmaxf(v, __shfl_xor(v, 2, 64));
  v = fmaxf(v, __shfl_xor(v, 4, 64)); v = fmaxf(v, __shfl_xor(v, 8, 64));
  return v;
}
__device__ __forceinline__ float group16_sum(float v) {
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
  return v;
}

struct AttnParams {
  const void *q, *k, *v, *out, *dout;
  const float* key_bias;
  void *o, *dq, *dk, *dv;
  float *lse, *delta;
  int B, H, Tq, Tk, dh;
  int64_t ldq, ldk, ldv, ldo;
  float scale, float_min;
  int causal;
  uint32_t drop_thresh;
  float drop_inv_keep;
  uint64_t seed, stream_id;
  int vq, vk, vv, vo;  // 16-byte vector loads legal for q / k / v / (out,dout)
};

// Dropout mask of attention probabilities.  One Philox call serves the 4 keys {64*kt + 16*f + lc, f = 0..3} of one
// query row -- exactly the 4 C-fragment columns a lane owns in the forward / dQ kernels -- so those kernels pay one
// Philox per 4 probabilities.  counter = ((bh*Tq + q) * ceil(Tk/64) + kt) * 16 + lc ; word f.
__device__ __forceinline__ Philox4 attn_drop4(const AttnParams& p, int64_t bh_row_base, int qg, int kt, int lc) {
  const int nkt = (p.Tk + TR - 1) / TR;
  const uint64_t ctr = (uint64_t)(((bh_row_base + qg) * nkt + kt) * 16 + lc);
  return philox4x32_10(p.seed, p.stream_id, ctr);
}
__device__ __forceinline__ float drop_scale(const AttnParams& p, uint32_t word) {
  return word >= p.drop_thresh ? p.drop_inv_keep : 0.f;
}

// biased logit of (query qg, key kg) from the raw dot product -- the reference's fp32 order of operations
__device__ __forceinline__ float biased_logit(const AttnParams& p, float raw, float kbias, int qg, int kg) {
  float v = raw * p.scale + kbias;
  if (p.causal && kg > qg) v += p.float_min;
  return v;
}

// =============================================================================================
// forward
// =============================================================================================
template <typename T, bool USE_TR>
__global__ void __launch_bounds__(256) attn_fwd_kernel(AttnParams p) {
  typedef typename FragT<T>::type Frag;
  __shared__ __attribute__((aligned(16))) char smem[3 * AT<T>::TILE_BYTES];
  char* Ks = smem;
  char* Vs = smem + AT<T>::TILE_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  char* Ps = smem + 2 * AT<T>::TILE_BYTES + wave * 16 * AT<T>::RS;
  const int q0 = blockIdx.x * TR, h = blockIdx.y, b = blockIdx.z;
  const T* qb = (const T*)p.q + (int64_t)b * p.Tq * p.ldq + h * p.dh;
  const T* kb = (const T*)p.k + (int64_t)b * p.Tk * p.ldk + h * p.dh;
  const T* vb = (const T*)p.v + (int64_t)b * p.Tk * p.ldv + h * p.dh;
  const int lq = lane >> 4, lc = lane & 15;

  stage_tile<T>(Ks, qb, p.ldq, q0, p.Tq, p.dh, p.vq, tid);
  __syncthreads();
  Frag qf[AT<T>::NK];
#pragma unroll
  for (int s = 0; s < AT<T>::NK; ++s) qf[s] = rc_frag<T>(Ks, wave * 16, s * AT<T>::KS, lane);
  __syncthreads();

  float m_run[4], l_run[4];
  floatx4_t o[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { m_run[r] = -INFINITY; l_run[r] = 0.f; }
#pragma unroll
  for (int f = 0; f < 4; ++f) o[f] = floatx4_t{0.f, 0.f, 0.f, 0.f};

  int nkt = (p.Tk + TR - 1) / TR;
  if (p.causal) { const int lim = (q0 + TR - 1) / TR + 1; if (lim < nkt) nkt = lim; }
  const int64_t drop_row_base = ((int64_t)b * p.H + h) * p.Tq;

  TileRegs<T> kreg, vreg;
  tile_load<T>(kreg, kb, p.ldk, 0, p.Tk, p.dh, p.vk, tid);
  tile_load<T>(vreg, vb, p.ldv, 0, p.Tk, p.dh, p.vv, tid);
  for (int kt = 0; kt < nkt; ++kt) {
    const int k0 = kt * TR;
    tile_store<T>(Ks, kreg, tid);
    tile_store<T>(Vs, vreg, tid);
    __syncthreads();
    if (kt + 1 < nkt) {
      tile_load<T>(kreg, kb, p.ldk, k0 + TR, p.Tk, p.dh, p.vk, tid);
      tile_load<T>(vreg, vb, p.ldv, k0 + TR, p.Tk, p.dh, p.vv, tid);
    }
    floatx4_t s[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) s[f] = floatx4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < AT<T>::NK; ++st)
#pragma unroll
      for (int f = 0; f < 4; ++f) s[f] = Mma<T>::run(qf[st], rc_frag<T>(Ks, f * 16, st * AT<T>::KS, lane), s[f]);

    float rmax[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int kg = k0 + f * 16 + lc;
      const bool kvalid = kg < p.Tk;
      const float kbias = (kvalid && p.key_bias) ? p.key_bias[(int64_t)b * p.Tk + kg] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qg = q0 + wave * 16 + lq * 4 + r;
        const float v = kvalid ? biased_logit(p, s[f][r], kbias, qg, kg) : -INFINITY;
        s[f][r] = v;
        rmax[r] = fmaxf(rmax[r], v);
      }
    }
    float alpha[4], rsum[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float mn = fmaxf(m_run[r], group16_max(rmax[r]));
      alpha[r] = __expf(m_run[r] - mn);
      m_run[r] = mn;
      rsum[r] = 0.f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qg = q0 + wave * 16 + lq * 4 + r;
      float keep[4] = {1.f, 1.f, 1.f, 1.f};
      if (p.drop_thresh) {
        const Philox4 w4 = attn_drop4(p, drop_row_base, qg, kt, lc);
        keep[0] = drop_scale(p, w4.x); keep[1] = drop_scale(p, w4.y); keep[2] = drop_scale(p, w4.z); keep[3] = drop_scale(p, w4.w);
      }
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const float pv = __expf(s[f][r] - m_run[r]);
        rsum[r] += pv;
        *reinterpret_cast<T*>(Ps + (lq * 4 + r) * AT<T>::RS + (f * 16 + lc) * (int)sizeof(T)) = from_f32<T>(pv * keep[f]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) l_run[r] = l_run[r] * alpha[r] + group16_sum(rsum[r]);
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[f][r] *= alpha[r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int st = 0; st < AT<T>::NK; ++st) {
      const Frag a = rc_frag<T>(Ps, 0, st * AT<T>::KS, lane);
#pragma unroll
      for (int f = 0; f < 4; ++f) o[f] = Mma<T>::run(a, oc_frag<T, USE_TR>(Vs, f * 16, st * AT<T>::KS, lane), o[f]);
    }
    __syncthreads();
  }

  T* ob = (T*)p.o + (int64_t)b * p.Tq * p.ldo + h * p.dh;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qg = q0 + wave * 16 + lq * 4 + r;
    const bool qok = qg < p.Tq;
    const float inv = 1.f / l_run[r];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int d = f * 16 + lc;
      if (qok && d < p.dh) ob[(int64_t)qg * p.ldo + d] = from_f32<T>(o[f][r] * inv);
    }
    if (qok && lc == 0) p.lse[((int64_t)b * p.H + h) * p.Tq + qg] = m_run[r] + logf(l_run[r]);
  }
}

// =============================================================================================
// backward, step 0: delta[b,h,q] = sum_d dout*out
// =============================================================================================
template <typename T>
__global__ void attn_delta_kernel(AttnParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t total = (int64_t)p.B * p.H * p.Tq;
  for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < total; i += (int64_t)gridDim.x * 4) {
    const int qg = (int)(i % p.Tq);
    const int h = (int)((i / p.Tq) % p.H);
    const int b = (int)(i / ((int64_t)p.Tq * p.H));
    const T* o = (const T*)p.out + ((int64_t)b * p.Tq + qg) * p.ldo + h * p.dh;
    const T* g = (const T*)p.dout + ((int64_t)b * p.Tq + qg) * p.ldo + h * p.dh;
    float acc = 0.f;
    for (int d = lane; d < p.dh; d += 64) acc += to_f32<T>(o[d]) * to_f32<T>(g[d]);
    acc = wave_sum(acc);
    if (lane == 0) p.delta[i] = acc;
  }
}

// =============================================================================================
// backward, step 1: dK, dV.  One workgroup per 64-key tile; loops over query tiles.
//   S^T[key][q] = K.Q^T ; P^T = exp(S^T*scale + bias - lse[q]) ; dV += Pdrop^T.dO
//   dP^T[key][q] = V.dO^T ; dS^T = P^T o (keep*dP^T - delta[q]) ; dK += scale * dS^T.Q
// =============================================================================================
template <typename T, bool USE_TR>
__global__ void __launch_bounds__(256) attn_bwd_dkdv_kernel(AttnParams p) {
  typedef typename FragT<T>::type Frag;
  __shared__ __attribute__((aligned(16))) char smem[3 * AT<T>::TILE_BYTES];
  char* Qs = smem;                       // [q][d]
  char* Gs = smem + AT<T>::TILE_BYTES;   // dO [q][d]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  char* Ps = smem + 2 * AT<T>::TILE_BYTES + wave * 16 * AT<T>::RS;  // [16 keys][64 q], wave private
  const int k0 = blockIdx.x * TR, h = blockIdx.y, b = blockIdx.z;
  const T* qb = (const T*)p.q + (int64_t)b * p.Tq * p.ldq + h * p.dh;
  const T* kb = (const T*)p.k + (int64_t)b * p.Tk * p.ldk + h * p.dh;
  const T* vb = (const T*)p.v + (int64_t)b * p.Tk * p.ldv + h * p.dh;
  const T* gb = (const T*)p.dout + (int64_t)b * p.Tq * p.ldo + h * p.dh;
  const int lq = lane >> 4, lc = lane & 15;

  // K and V fragments of this wave's 16 keys stay in registers for the whole kernel
  Frag kf[AT<T>::NK], vf[AT<T>::NK];
  stage_tile<T>(Qs, kb, p.ldk, k0, p.Tk, p.dh, p.vk, tid);
  stage_tile<T>(Gs, vb, p.ldv, k0, p.Tk, p.dh, p.vv, tid);
  __syncthreads();
#pragma unroll
  for (int s = 0; s < AT<T>::NK; ++s) {
    kf[s] = rc_frag<T>(Qs, wave * 16, s * AT<T>::KS, lane);
    vf[s] = rc_frag<T>(Gs, wave * 16, s * AT<T>::KS, lane);
  }
  __syncthreads();

  floatx4_t dk[4], dv[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) { dk[f] = floatx4_t{0.f, 0.f, 0.f, 0.f}; dv[f] = floatx4_t{0.f, 0.f, 0.f, 0.f}; }

  // per-lane key data: this lane's rows of the C fragments are keys k0 + wave*16 + lq*4 + r
  float kbias[4];
  bool kvalid[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int kg = k0 + wave * 16 + lq * 4 + r;
    kvalid[r] = kg < p.Tk;
    kbias[r] = (kvalid[r] && p.key_bias) ? p.key_bias[(int64_t)b * p.Tk + kg] : 0.f;
  }
  const int nqt = (p.Tq + TR - 1) / TR;
  const int qt_first = p.causal ? k0 / TR : 0;  // queries before the key tile never see it
  const int64_t stat_base = ((int64_t)b * p.H + h) * p.Tq;

  TileRegs<T> qreg, greg;
  if (qt_first < nqt) {
    tile_load<T>(qreg, qb, p.ldq, qt_first * TR, p.Tq, p.dh, p.vq, tid);
    tile_load<T>(greg, gb, p.ldo, qt_first * TR, p.Tq, p.dh, p.vo, tid);
  }
  for (int qt = qt_first; qt < nqt; ++qt) {
    const int q0 = qt * TR;
    tile_store<T>(Qs, qreg, tid);
    tile_store<T>(Gs, greg, tid);
    __syncthreads();
    if (qt + 1 < nqt) {
      tile_load<T>(qreg, qb, p.ldq, q0 + TR, p.Tq, p.dh, p.vq, tid);
      tile_load<T>(greg, gb, p.ldo, q0 + TR, p.Tq, p.dh, p.vo, tid);
    }
    floatx4_t st[4], dp[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) { st[f] = floatx4_t{0.f, 0.f, 0.f, 0.f}; dp[f] = floatx4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int s = 0; s < AT<T>::NK; ++s)
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        st[f] = Mma<T>::run(kf[s], rc_frag<T>(Qs, f * 16, s * AT<T>::KS, lane), st[f]);
        dp[f] = Mma<T>::run(vf[s], rc_frag<T>(Gs, f * 16, s * AT<T>::KS, lane), dp[f]);
      }
    // P^T (dropped) -> Ps, then dV += P^T . dO
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int qg = q0 + f * 16 + lc;
      const bool qvalid = qg < p.Tq;
      const float lse = qvalid ? p.lse[stat_base + qg] : 0.f;
      const float dl = qvalid ? p.delta[stat_base + qg] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kg = k0 + wave * 16 + lq * 4 + r;
        float pv = 0.f, keep = 1.f;
        if (qvalid && kvalid[r]) {
          pv = __expf(biased_logit(p, st[f][r], kbias[r], qg, kg) - lse);
          if (p.drop_thresh) {
            const Philox4 w4 = attn_drop4(p, stat_base, qg, blockIdx.x, lq * 4 + r);
            const uint32_t wsel = wave == 0 ? w4.x : (wave == 1 ? w4.y : (wave == 2 ? w4.z : w4.w));
            keep = drop_scale(p, wsel);
          }
        }
        *reinterpret_cast<T*>(Ps + (lq * 4 + r) * AT<T>::RS + (f * 16 + lc) * (int)sizeof(T)) = from_f32<T>(pv * keep);
        st[f][r] = pv * (keep * dp[f][r] - dl) * p.scale;  // dS^T (scaled), kept for the second product
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < AT<T>::NK; ++s) {
      const Frag a = rc_frag<T>(Ps, 0, s * AT<T>::KS, lane);
#pragma unroll
      for (int f = 0; f < 4; ++f) dv[f] = Mma<T>::run(a, oc_frag<T, USE_TR>(Gs, f * 16, s * AT<T>::KS, lane), dv[f]);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *reinterpret_cast<T*>(Ps + (lq * 4 + r) * AT<T>::RS + (f * 16 + lc) * (int)sizeof(T)) = from_f32<T>(st[f][r]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < AT<T>::NK; ++s) {
      const Frag a = rc_frag<T>(Ps, 0, s * AT<T>::KS, lane);
#pragma unroll
      for (int f = 0; f < 4; ++f) dk[f] = Mma<T>::run(a, oc_frag<T, USE_TR>(Qs, f * 16, s * AT<T>::KS, lane), dk[f]);
    }
    __syncthreads();
  }

  T* dkb = (T*)p.dk + (int64_t)b * p.Tk * p.ldk + h * p.dh;
  T* dvb = (T*)p.dv + (int64_t)b * p.Tk * p.ldv + h * p.dh;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int kg = k0 + wave * 16 + lq * 4 + r;
    const bool kok = kg < p.Tk;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int d = f * 16 + lc;
      if (kok && d < p.dh) {
        dkb[(int64_t)kg * p.ldk + d] = from_f32<T>(dk[f][r]);
        dvb[(int64_t)kg * p.ldv + d] = from_f32<T>(dv[f][r]);
      }
    }
  }
}

// =============================================================================================
// backward, step 2: dQ.  One workgroup per 64-query tile; loops over key tiles.
//   S = Q.K^T ; P = exp(S*scale + bias - lse) ; dP = dO.V^T ; dS = P o (keep*dP - delta) ; dQ += scale * dS.K
// =============================================================================================
template <typename T, bool USE_TR>
__global__ void __launch_bounds__(256) attn_bwd_dq_kernel(AttnParams p) {
  typedef typename FragT<T>::type Frag;
  __shared__ __attribute__((aligned(16))) char smem[3 * AT<T>::TILE_BYTES];
  char* Ks = smem;
  char* Vs = smem + AT<T>::TILE_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  char* Ps = smem + 2 * AT<T>::TILE_BYTES + wave * 16 * AT<T>::RS;
  const int q0 = blockIdx.x * TR, h = blockIdx.y, b = blockIdx.z;
  const T* qb = (const T*)p.q + (int64_t)b * p.Tq * p.ldq + h * p.dh;
  const T* kb = (const T*)p.k + (int64_t)b * p.Tk * p.ldk + h * p.dh;
  const T* vb = (const T*)p.v + (int64_t)b * p.Tk * p.ldv + h * p.dh;
  const T* gb = (const T*)p.dout + (int64_t)b * p.Tq * p.ldo + h * p.dh;
  const int lq = lane >> 4, lc = lane & 15;

  Frag qf[AT<T>::NK], gf[AT<T>::NK];
  stage_tile<T>(Ks, qb, p.ldq, q0, p.Tq, p.dh, p.vq, tid);
  stage_tile<T>(Vs, gb, p.ldo, q0, p.Tq, p.dh, p.vo, tid);
  __syncthreads();
#pragma unroll
  for (int s = 0; s < AT<T>::NK; ++s) {
    qf[s] = rc_frag<T>(Ks, wave * 16, s * AT<T>::KS, lane);
    gf[s] = rc_frag<T>(Vs, wave * 16, s * AT<T>::KS, lane);
  }
  __syncthreads();

  const int64_t stat_base = ((int64_t)b * p.H + h) * p.Tq;
  float lse[4], dl[4];
  bool qvalid[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qg = q0 + wave * 16 + lq * 4 + r;
    qvalid[r] = qg < p.Tq;
    lse[r] = qvalid[r] ? p.lse[stat_base + qg] : 0.f;
    dl[r] = qvalid[r] ? p.delta[stat_base + qg] : 0.f;
  }
  floatx4_t dq[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) dq[f] = floatx4_t{0.f, 0.f, 0.f, 0.f};

  int nkt = (p.Tk + TR - 1) / TR;
  if (p.causal) { const int lim = (q0 + TR - 1) / TR + 1; if (lim < nkt) nkt = lim; }
  TileRegs<T> kreg, vreg;
  tile_load<T>(kreg, kb, p.ldk, 0, p.Tk, p.dh, p.vk, tid);
  tile_load<T>(vreg, vb, p.ldv, 0, p.Tk, p.dh, p.vv, tid);
  for (int kt = 0; kt < nkt; ++kt) {
    const int k0 = kt * TR;
    tile_store<T>(Ks, kreg, tid);
    tile_store<T>(Vs, vreg, tid);
    __syncthreads();
    if (kt + 1 < nkt) {
      tile_load<T>(kreg, kb, p.ldk, k0 + TR, p.Tk, p.dh, p.vk, tid);
      tile_load<T>(vreg, vb, p.ldv, k0 + TR, p.Tk, p.dh, p.vv, tid);
    }
    floatx4_t s[4], dp[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) { s[f] = floatx4_t{0.f, 0.f, 0.f, 0.f}; dp[f] = floatx4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int st = 0; st < AT<T>::NK; ++st)
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        s[f] = Mma<T>::run(qf[st], rc_frag<T>(Ks, f * 16, st * AT<T>::KS, lane), s[f]);
        dp[f] = Mma<T>::run(gf[st], rc_frag<T>(Vs, f * 16, st * AT<T>::KS, lane), dp[f]);
      }
    float kbias4[4];
    bool kvalid4[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int kg = k0 + f * 16 + lc;
      kvalid4[f] = kg < p.Tk;
      kbias4[f] = (kvalid4[f] && p.key_bias) ? p.key_bias[(int64_t)b * p.Tk + kg] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qg = q0 + wave * 16 + lq * 4 + r;
      float keep[4] = {1.f, 1.f, 1.f, 1.f};
      if (p.drop_thresh) {
        const Philox4 w4 = attn_drop4(p, stat_base, qg, kt, lc);
        keep[0] = drop_scale(p, w4.x); keep[1] = drop_scale(p, w4.y); keep[2] = drop_scale(p, w4.z); keep[3] = drop_scale(p, w4.w);
      }
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const int kg = k0 + f * 16 + lc;
        float ds = 0.f;
        if (kvalid4[f] && qvalid[r]) {
          const float pv = __expf(biased_logit(p, s[f][r], kbias4[f], qg, kg) - lse[r]);
          ds = pv * (keep[f] * dp[f][r] - dl[r]) * p.scale;
        }
        *reinterpret_cast<T*>(Ps + (lq * 4 + r) * AT<T>::RS + (f * 16 + lc) * (int)sizeof(T)) = from_f32<T>(ds);
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int st = 0; st < AT<T>::NK; ++st) {
      const Frag a = rc_frag<T>(Ps, 0, st * AT<T>::KS, lane);
#pragma unroll
      for (int f = 0; f < 4; ++f) dq[f] = Mma<T>::run(a, oc_frag<T, USE_TR>(Ks, f * 16, st * AT<T>::KS, lane), dq[f]);
    }
    __syncthreads();
  }
  T* dqb = (T*)p.dq + (int64_t)b * p.Tq * p.ldq + h * p.dh;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qg = q0 + wave * 16 + lq * 4 + r;
    const bool qok = qg < p.Tq;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int d = f * 16 + lc;
      if (qok && d < p.dh) dqb[(int64_t)qg * p.ldq + d] = from_f32<T>(dq[f][r]);
    }
  }
}

bool attn_use_tr() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("NST_GEMM_NO_TR"); v = (e && e[0] == '1') ? 0 : 1; }
  return v == 1;
}

int vec_legal(const void* base, int64_t ld, int dh, int esz) {
  const int E = 16 / esz;
  return nst_aligned16(base) && ((ld * esz) % 16 == 0) && (dh % E == 0) && ((dh * esz) % 16 == 0);
}

int fill_params(const NstAttnDesc* d, AttnParams& p) {
  NST_CHECK_ARG(d, "attention: null descriptor");
  NST_CHECK_ARG(d->B > 0 && d->H > 0 && d->Tq > 0 && d->Tk > 0, "attention: bad dims B=%d H=%d Tq=%d Tk=%d", d->B, d->H, d->Tq, d->Tk);
  NST_CHECK_ARG(d->dh > 0 && d->dh <= DH, "attention: dh=%d unsupported (1..%d)", d->dh, DH);
  NST_CHECK_ARG(d->dtype == NST_F32 || d->dtype == NST_BF16, "attention: bad dtype %d", d->dtype);
  NST_CHECK_ARG(d->dropout_p >= 0.f && d->dropout_p < 1.f, "attention: dropout_p=%f", d->dropout_p);
  NST_CHECK_ARG(d->ldq >= (int64_t)d->H * d->dh && d->ldk >= (int64_t)d->H * d->dh && d->ldv >= (int64_t)d->H * d->dh &&
                    d->ldo >= (int64_t)d->H * d->dh, "attention: row stride smaller than H*dh");
  p.B = d->B; p.H = d->H; p.Tq = d->Tq; p.Tk = d->Tk; p.dh = d->dh;
  p.ldq = d->ldq; p.ldk = d->ldk; p.ldv = d->ldv; p.ldo = d->ldo;
  p.scale = d->scale; p.float_min = d->float_min; p.causal = d->causal;
  p.drop_thresh = nst_dropout_threshold(d->dropout_p);
  p.drop_inv_keep = 1.f / (1.f - d->dropout_p);
  p.seed = d->seed; p.stream_id = d->stream_id;
  return NST_OK;
}

}  // namespace

extern "C" int nst_attention_fwd(const NstAttnDesc* d, const void* q, const void* k, const void* v, const float* key_bias,
                                 void* out, float* lse, void* stream) {
  AttnParams p;
  memset(&p, 0, sizeof(p));
  int rc = fill_params(d, p);
  if (rc) return rc;
  NST_CHECK_ARG(q && k && v && out && lse, "attention_fwd: null pointer");
  const int esz = nst_dtype_size(d->dtype);
  p.q = q; p.k = k; p.v = v; p.o = out; p.lse = lse; p.key_bias = key_bias;
  p.vq = vec_legal(q, d->ldq, d->dh, esz); p.vk = vec_legal(k, d->ldk, d->dh, esz); p.vv = vec_legal(v, d->ldv, d->dh, esz);
  dim3 grid((d->Tq + TR - 1) / TR, d->H, d->B);
  hipStream_t st = (hipStream_t)stream;
  const bool tr = attn_use_tr();
  if (d->dtype == NST_F32) attn_fwd_kernel<float, true><<<grid, 256, 0, st>>>(p);
  else if (tr) attn_fwd_kernel<bf16_t, true><<<grid, 256, 0, st>>>(p);
  else attn_fwd_kernel<bf16_t, false><<<grid, 256, 0, st>>>(p);
  NST_CHECK_LAUNCH("attention_fwd");
  return NST_OK;
}

extern "C" int nst_attention_bwd(const NstAttnDesc* d, const void* q, const void* k, const void* v, const float* key_bias,
                                 const void* out, const void* dout, const float* lse, float* delta, void* dq, void* dk,
                                 void* dv, void* stream) {
  AttnParams p;
  memset(&p, 0, sizeof(p));
  int rc = fill_params(d, p);
  if (rc) return rc;
  NST_CHECK_ARG(q && k && v && out && dout && lse && delta && dq && dk && dv, "attention_bwd: null pointer");
  const int esz = nst_dtype_size(d->dtype);
  p.q = q; p.k = k; p.v = v; p.out = out; p.dout = dout; p.lse = const_cast<float*>(lse); p.delta = delta;
  p.key_bias = key_bias; p.dq = dq; p.dk = dk; p.dv = dv;
  p.vq = vec_legal(q, d->ldq, d->dh, esz); p.vk = vec_legal(k, d->ldk, d->dh, esz); p.vv = vec_legal(v, d->ldv, d->dh, esz);
  p.vo = vec_legal(dout, d->ldo, d->dh, esz);
  hipStream_t st = (hipStream_t)stream;
  const bool tr = attn_use_tr();
  {
    int64_t rows = (int64_t)d->B * d->H * d->Tq;
    int blocks = (int)((rows + 3) / 4 > 4096 ? 4096 : (rows + 3) / 4);
    if (d->dtype == NST_F32) attn_delta_kernel<float><<<blocks, 256, 0, st>>>(p);
    else attn_delta_kernel<bf16_t><<<blocks, 256, 0, st>>>(p);
    NST_CHECK_LAUNCH("attention_bwd(delta)");
  }
  dim3 gk((d->Tk + TR - 1) / TR, d->H, d->B), gq((d->Tq + TR - 1) / TR, d->H, d->B);
  if (d->dtype == NST_F32) {
    attn_bwd_dkdv_kernel<float, true><<<gk, 256, 0, st>>>(p);
    attn_bwd_dq_kernel<float, true><<<gq, 256, 0, st>>>(p);
  } else if (tr) {
    attn_bwd_dkdv_kernel<bf16_t, true><<<gk, 256, 0, st>>>(p);
    attn_bwd_dq_kernel<bf16_t, true><<<gq, 256, 0, st>>>(p);
  } else {
    attn_bwd_dkdv_kernel<bf16_t, false><<<gk, 256, 0, st>>>(p);
    attn_bwd_dq_kernel<bf16_t, false><<<gq, 256, 0, st>>>(p);
  }
  NST_CHECK_LAUNCH("attention_bwd");
  return NST_OK;
}
