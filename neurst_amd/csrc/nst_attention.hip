// Fused scaled-dot-product attention, forward and backward (flash style: the [B,H,Tq,Tk] probability
// matrix is never written to HBM; only the per-row log-sum-exp and, under dropout, one mask BIT per probability
// are saved).
//   replaces q*=dh^-0.5 ; einsum(BTNH,BFNH->BNFT) ; +bias ; softmax ; dropout ; einsum(BNFT,BTNH->BFNH)
//   of neurst/layers/attentions/multi_head_attention.py:124-164, 203-215 and their TF gradients.
//
// Register-resident design.  A 16x16 MFMA accumulator keeps, in lane l, column l&15 and rows (l>>4)*4+0..3 of its
// block -- which is exactly the B-operand layout (column l&15, 4..8 consecutive reduction indices) of the NEXT
// product as long as the reduction runs over the accumulator's ROW index.  So every kernel computes its score block
// with the reduction index of the following product on the rows:
//     forward / dQ :  S^T[key][q] = K.Q^T   -> P^T / dS^T feed  O^T[d][q]  = V^T.P^T   and dQ^T[d][q] = K^T.dS^T
//     dK,dV        :  S[q][key]   = Q.K^T   -> P / dS feed      dV^T[d][key] = dO^T.P  and dK^T[d][key] = Q^T.dS
// The probabilities never pass through LDS, the softmax statistics of a query are one scalar per lane (the lane's
// column), and the transposed A operands (V^T, K^T, dO^T, Q^T) come from the row-major LDS tiles through the LDS
// transpose read (ds_read_b64_tr_b16).  The 8 reduction slots of a bf16 MFMA step are fed with rows
// {g*4+0..3, 16+g*4+0..3} of a 32-row step (g = l>>4) on BOTH operands, which is only a permutation of the sum.
// Each wave owns MI blocks of 16 queries (keys in dK/dV), so one K/V (Q/dO) fragment read serves MI MFMAs.
//
// Additive masks follow the reference (bias added in fp32; padding bias stays finite so a fully padded row degrades
// to the same uniform distribution; causal positions and keys beyond Tk get probability exactly 0, as FLOAT_MIN does
// in fp32).  exp() is evaluated as exp2 with log2(e) folded into the logit scale.
//
// Dropout: the forward kernel draws 16-bit Philox fields (two calls per lane, tile and query block), applies the
// mask and stores the 16 keep bits of each lane as one u16:
//     mask[((bh*ceil(Tq/16) + qblk)*ceil(Tk/64) + ktile)*64 + lane]  bit (f*4+r)  <->  query qblk*16 + (lane&15),
//                                                                      key ktile*64 + f*16 + (lane>>4)*4 + r
// The backward kernels read the bits back instead of regenerating random numbers.
#include "nst_gemm_core.h"

#include <stdlib.h>

#include <type_traits>

using nstgemm::Mma;

namespace {

constexpr int DH = 64;   // padded head dim (dh <= 64)
constexpr int TR = 64;   // rows of one staged tile (queries or keys)
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

template <typename T>
struct AT {
  static constexpr int RS = DH * (int)sizeof(T) + 16;  // LDS row stride in bytes
  static constexpr int E = 16 / (int)sizeof(T);
  static constexpr int CPR = DH * (int)sizeof(T) / 16;  // 16-byte chunks per row
  static constexpr int KS = Mma<T>::KS;
  static constexpr int NK = DH / KS;                    // MFMA steps over the head dim
  static constexpr int TILE_BYTES = TR * RS;
  static constexpr int NCH = TR * CPR / 256;            // chunks per thread per tile
};

template <typename T>
struct FragT;
template <>
struct FragT<bf16_t> { typedef bf16x8_t type; };
template <>
struct FragT<float> { typedef float type; };

// head-dim-contiguous fragment (A or B operand of the score products): row row0+(l&15), head dims kk + ...
template <typename T>
__device__ __forceinline__ typename FragT<T>::type rc_frag(const char* tile, int row0, int kk, int lane) {
  if constexpr (sizeof(T) == 2)
    return *reinterpret_cast<const bf16x8_t*>(tile + (row0 + (lane & 15)) * AT<T>::RS + (kk + (lane >> 4) * 8) * 2);
  else
    return *reinterpret_cast<const float*>(tile + (row0 + (lane & 15)) * AT<T>::RS + (kk + (lane >> 4)) * 4);
}

// acc[mi][fd] (+)= sum over the 64 rows of `tile` of  tile[row][fd*16 + i] (A operand, output row i)
//                                                   x  pt[mi][f][r]        (B operand: row f*16 + g*4 + r, column l&15)
// RSV: row stride of `tile` in bytes.  The forward stages V with 160-byte rows (VS_RS): a ds_read_b64_tr_b16 pass serves 32
// lanes = 8 rows x 32 bytes, and a stride of 8 dwords mod 64 spreads them over all 64 banks (with the 144-byte rows that suit
// the ds_read_b128 row reads of K and Q, rows r and r + 7 overlap: 36 % of the forward's LDS cycles were bank conflicts,
// profiles/r04_pmc_attn.json); the backward kernels made the same choice in round 4 (HB_RS).
template <typename T, int MI, int RSV = AT<T>::RS>
__device__ __forceinline__ void tmul_acc(floatx4_t (&acc)[MI][4], const floatx4_t (&pt)[MI][4], const char* tile, int lane) {
  const int g = lane >> 4, lc = lane & 15;
  if constexpr (sizeof(T) == 2) {
    typedef short4_t __attribute__((address_space(3))) * lds_ptr_t;
    // transpose read: lane ii of a 16-lane group addresses row ii>>2, 8-byte chunk ii&3 and receives column ii
    const char* base = tile + (g * 4 + (lc >> 2)) * RSV + (lc & 3) * 8;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      bf16x8_t b[MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const floatx4_t lo = pt[mi][2 * st], hi = pt[mi][2 * st + 1];
        union { uint32_t u[4]; bf16x8_t f; } u;
        u.u[0] = pack_bf16x2(lo[0], lo[1]); u.u[1] = pack_bf16x2(lo[2], lo[3]);
        u.u[2] = pack_bf16x2(hi[0], hi[1]); u.u[3] = pack_bf16x2(hi[2], hi[3]);
        b[mi] = u.f;
      }
#pragma unroll
      for (int fd = 0; fd < 4; ++fd) {
        const char* p = base + st * 32 * RSV + fd * 32;
        const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(p));
        const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(p + 16 * RSV));
        union { short4_t h[2]; bf16x8_t f; } a;
        a.h[0] = lo; a.h[1] = hi;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[mi][fd] = Mma<T>::run(a.f, b[mi], acc[mi][fd]);
      }
    }
  } else {
    const char* base = tile + (g * 4) * RSV + lc * 4;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int fd = 0; fd < 4; ++fd) {
          const float a = *reinterpret_cast<const float*>(base + (f * 16 + r) * RSV + fd * 64);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) acc[mi][fd] = Mma<T>::run(a, pt[mi][f][r], acc[mi][fd]);
        }
  }
}

// ---------------------------------------------------------------------------------------------
// staging: 64 rows x DH of a [nrows, dh] matrix (row stride ld elements) through registers into LDS, zero filled.
// VEC: 16-byte loads are legal (aligned base, ld and dh multiples of a chunk); the loads are branch-free (a chunk
// outside the matrix reads the first chunk of the matrix and is zeroed by a select).
// ---------------------------------------------------------------------------------------------
template <typename T>
struct TileRegs {
  uint4 v[AT<T>::NCH];
};
template <typename T, bool VEC>
__device__ __forceinline__ void tile_load(TileRegs<T>& regs, const T* __restrict__ base, int64_t ld, int row0, int nrows, int dh,
                                          int tid) {
#pragma unroll
  for (int s = 0; s < AT<T>::NCH; ++s) {
    const int c = tid + s * 256;
    const int row = row0 + c / AT<T>::CPR, col = (c % AT<T>::CPR) * AT<T>::E;
    const bool ok = row < nrows && col < dh;
    if (VEC) {
      const T* p = ok ? base + (int64_t)row * ld + col : base;
      const uint4 v = *reinterpret_cast<const uint4*>(p);
      regs.v[s] = ok ? v : make_uint4(0, 0, 0, 0);
    } else {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ok) {
        const T* p = base + (int64_t)row * ld + col;
        T tmp[AT<T>::E];
#pragma unroll
        for (int e = 0; e < AT<T>::E; ++e) tmp[e] = (col + e < dh) ? p[e] : (T)0;
        memcpy(&v, tmp, 16);
      }
      regs.v[s] = v;
    }
  }
}
template <typename T, int RSV = AT<T>::RS>
__device__ __forceinline__ void tile_store(char* lds, const TileRegs<T>& regs, int tid) {
#pragma unroll
  for (int s = 0; s < AT<T>::NCH; ++s) {
    const int c = tid + s * 256;
    *reinterpret_cast<uint4*>(lds + (c / AT<T>::CPR) * RSV + (c % AT<T>::CPR) * 16) = regs.v[s];
  }
}

// Reductions over the 4 lanes {lc, lc+16, lc+32, lc+48} that share a C-fragment column: two gfx950 lane-row swaps
// (v_permlane16_swap / v_permlane32_swap, VALU only -- no LDS crossbar).
//   permlane16_swap(a, b): swaps the odd 16-lane rows of a with the even rows of b;  permlane32_swap: upper half of
//   a with lower half of b.  With a == b == x the two results hold x and its partner row in every lane.
typedef __attribute__((ext_vector_type(2))) unsigned uint2_hw_t;
__device__ __forceinline__ float xgroup_max(float v) {
  uint2_hw_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xgroup_sum(float v) {
  uint2_hw_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// raw v_exp_f32 (no denormal range fix-up: arguments are <= 0 up to rounding, tiny results may flush to 0)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// a recomputed probability: exp2 with the result clamped to [0, 1] (the VOP3 clamp bit of v_exp_f32, no extra instruction).
// Mathematically a no-op; it keeps a row whose saved log-sum-exp lost its low bits next to the -1e9 padding bias (every
// key padded) from turning into inf, and inf * 0 into NaN gradients for the whole (batch, head).
__device__ __forceinline__ float prob_exp2(float x) { return __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(x), 0.f, 1.f); }

struct AttnParams {
  const void *q, *k, *v, *out, *dout;
  const float* key_bias;
  void *o, *dq, *dk, *dv;
  float *lse, *delta;
  uint16_t* mask;  // dropout keep bits (see the header comment), NULL without dropout
  int B, H, Tq, Tk, dh;
  int nqb, nkt;    // ceil(Tq/16), ceil(Tk/64)
  int64_t ldq, ldk, ldv, ldo;
  int64_t bsk, bsv;  // batch strides of k / v (and dk / dv), elements
  float scale, scale2;  // dh^-0.5 and dh^-0.5 * log2(e)
  int causal;
  int coff;        // causal offset: key j is masked for query i when j > i + coff (0 = lower triangle, k-1 = wait-k)
  uint32_t drop_thresh;  // 16-bit threshold, 0 = no dropout
  float drop_inv_keep;
  uint64_t seed, stream_id;
  const uint64_t* seed_dev;   // device scalar added to `seed` when the forward kernel runs (nst_dropout_seed_offset_*)
  bf16_t* dst;     // backward workspace: dS^T [B*H][tkp keys][tqp queries] bf16 (scaled), written by the dK/dV kernel
  int tqp, tkp, ds_kblock;   // padded extents; keys per dK/dV workgroup (the causal skip rule of that kernel)
};

// additive key term of the logits in the log2 domain; keys beyond Tk are excluded
__device__ __forceinline__ float key_bias2(const AttnParams& p, int b, int kg) {
  if (kg >= p.Tk) return -INFINITY;
  const float v = p.key_bias ? p.key_bias[(int64_t)b * p.Tk + kg] * LOG2E : 0.f;
  return fmaxf(v, -3.0e38f);
}

// the same in two halves for software-pipelined loops: the load (raw value, nothing depends on it) a tile ahead, the
// arithmetic when the value is needed -- a use right behind the load costs a full memory round trip every tile
__device__ __forceinline__ float key_bias_raw(const AttnParams& p, int b, int kg) {
  return (p.key_bias && kg < p.Tk) ? p.key_bias[(int64_t)b * p.Tk + kg] : 0.f;
}
__device__ __forceinline__ float key_bias_fin(const AttnParams& p, float raw, int kg) {
  return kg >= p.Tk ? -INFINITY : fmaxf(raw * LOG2E, -3.0e38f);
}

// write 4 consecutive head dims d..d+3 of one row
template <typename T, bool VEC>
__device__ __forceinline__ void store_row4(T* row, int d, int dh, const floatx4_t& v, float mul) {
  if (VEC) {
    if (d < dh) {
      if constexpr (sizeof(T) == 2)
        *reinterpret_cast<uint2*>(row + d) = make_uint2(pack_bf16x2(v[0] * mul, v[1] * mul), pack_bf16x2(v[2] * mul, v[3] * mul));
      else
        *reinterpret_cast<float4*>(row + d) = make_float4(v[0] * mul, v[1] * mul, v[2] * mul, v[3] * mul);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (d + r < dh) row[d + r] = from_f32<T>(v[r] * mul);
  }
}

// =============================================================================================
// forward.  One workgroup = 4 waves = 64*MI queries; wave w owns query blocks {mi*4 + w}.
// =============================================================================================
template <typename T, int MI, bool VEC>
__device__ __forceinline__ void attn_fwd_body(const AttnParams& p) {
  typedef typename FragT<T>::type Frag;
  constexpr int VS_RS = sizeof(T) == 2 ? 160 : AT<T>::RS;   // V rows: laid out for the transpose reads (see tmul_acc)
  __shared__ __attribute__((aligned(16))) char smem[AT<T>::TILE_BYTES + TR * VS_RS + TR * 4];
  char* Ks = smem;
  char* Vs = smem + AT<T>::TILE_BYTES;
  float* kbs = reinterpret_cast<float*>(smem + AT<T>::TILE_BYTES + TR * VS_RS);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, lc = lane & 15;
  const int q0 = blockIdx.x * (TR * MI), h = blockIdx.y, b = blockIdx.z;
  const T* qb = (const T*)p.q + (int64_t)b * p.Tq * p.ldq + h * p.dh;
  const T* kb = (const T*)p.k + (int64_t)b * p.bsk + h * p.dh;
  const T* vb = (const T*)p.v + (int64_t)b * p.bsv + h * p.dh;
  const int64_t bh = (int64_t)b * p.H + h;

  Frag qf[MI][AT<T>::NK];
  {
    TileRegs<T> qreg;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      tile_load<T, VEC>(qreg, qb, p.ldq, q0 + mi * TR, p.Tq, p.dh, tid);
      if (mi) __syncthreads();
      tile_store<T>(Ks, qreg, tid);
      __syncthreads();
#pragma unroll
      for (int s = 0; s < AT<T>::NK; ++s) qf[mi][s] = rc_frag<T>(Ks, wave * 16, s * AT<T>::KS, lane);
    }
    __syncthreads();
  }

  float m_run[MI], l_run[MI];
  floatx4_t o[MI][4];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    m_run[mi] = -INFINITY;
    l_run[mi] = 0.f;
#pragma unroll
    for (int f = 0; f < 4; ++f) o[mi][f] = floatx4_t{0.f, 0.f, 0.f, 0.f};
  }

  int nkt = p.nkt;
  if (p.causal) { const int lim = (q0 + TR * MI - 1 + p.coff) / TR + 1; if (lim < nkt) nkt = lim; }

  TileRegs<T> kreg, vreg;
  float kbreg = 0.f;
  tile_load<T, VEC>(kreg, kb, p.ldk, 0, p.Tk, p.dh, tid);
  tile_load<T, VEC>(vreg, vb, p.ldv, 0, p.Tk, p.dh, tid);
  if (tid < TR) kbreg = key_bias_raw(p, b, tid);
  // dropout-mask words of the previous tile, stored at the top of the next one (behind the barrier, in front of the
  // prefetch loads: vmcnt retires in order, a store issued after the loads would be waited for with them)
  int64_t mpend[MI];   // element index into p.mask (an index, not a pointer: a select with nullptr would turn the
  uint32_t mbits[MI];  // store into a flat_store, which also counts against lgkmcnt)
  bool mok[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) { mpend[mi] = 0; mbits[mi] = 0; mok[mi] = false; }
  for (int kt = 0; kt < nkt; ++kt) {
    const int k0 = kt * TR;
    tile_store<T>(Ks, kreg, tid);
    tile_store<T, VS_RS>(Vs, vreg, tid);
    if (tid < TR) kbs[tid] = key_bias_fin(p, kbreg, k0 + tid);
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
      if (mok[mi]) p.mask[mpend[mi]] = (uint16_t)mbits[mi];
    if (kt + 1 < nkt) {
      tile_load<T, VEC>(kreg, kb, p.ldk, k0 + TR, p.Tk, p.dh, tid);
      tile_load<T, VEC>(vreg, vb, p.ldv, k0 + TR, p.Tk, p.dh, tid);
      if (tid < TR) kbreg = key_bias_raw(p, b, k0 + TR + tid);
    }
    floatx4_t kb4[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) kb4[f] = *reinterpret_cast<const floatx4_t*>(kbs + f * 16 + g * 4);

    floatx4_t s[MI][4];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int f = 0; f < 4; ++f) s[mi][f] = floatx4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < AT<T>::NK; ++st)
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const Frag a = rc_frag<T>(Ks, f * 16, st * AT<T>::KS, lane);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) s[mi][f] = Mma<T>::run(a, qf[mi][st], s[mi][f]);
      }

#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int qblk0 = q0 + mi * TR + wave * 16;  // first query of this block; the lane's query is +lc
      const int qg = qblk0 + lc;
      const bool diag = p.causal && (k0 + TR - 1 > qblk0 + p.coff);
      float mx = -INFINITY;
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = fmaf(s[mi][f][r], p.scale2, kb4[f][r]);
          if (diag && (k0 + f * 16 + g * 4 + r > qg + p.coff)) x = -INFINITY;
          s[mi][f][r] = x;
          mx = fmaxf(mx, x);
        }
      mx = xgroup_max(mx);
      const float mn = fmaxf(m_run[mi], mx);
      const float alpha = fast_exp2(m_run[mi] - mn);
      m_run[mi] = mn;
      float psum = 0.f;
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = fast_exp2(s[mi][f][r] - mn);
          psum += pv;
          s[mi][f][r] = pv;
        }
      l_run[mi] = l_run[mi] * alpha + psum;
      if (p.drop_thresh) {
        // 16 fields of 16 bits for the lane's 16 probabilities: two Philox calls
        const uint64_t ctr = ((uint64_t)((bh * p.Tq + qg) * p.nkt + kt) * 4 + g) * 2;
        const uint64_t seed_eff = seed_with_offset(p.seed, p.seed_dev);
        const Philox4 w0 = philox4x32_10(seed_eff, p.stream_id, ctr), w1 = philox4x32_10(seed_eff, p.stream_id, ctr + 1);
        const uint32_t w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        uint32_t bits = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const uint32_t fld = (e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu);
          const bool keep = fld >= p.drop_thresh;
          bits |= keep ? (1u << e) : 0u;
          s[mi][e >> 2][e & 3] *= keep ? p.drop_inv_keep : 0.f;
        }
        mpend[mi] = ((bh * p.nqb + (qblk0 >> 4)) * p.nkt + kt) * 64 + lane;
        mok[mi] = qblk0 < p.Tq;
        mbits[mi] = bits;
      }
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[mi][f][r] *= alpha;
    }
    tmul_acc<T, MI, VS_RS>(o, s, Vs, lane);
    __syncthreads();
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
    if (mok[mi]) p.mask[mpend[mi]] = (uint16_t)mbits[mi];

  T* ob = (T*)p.o + (int64_t)b * p.Tq * p.ldo + h * p.dh;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int qg = q0 + mi * TR + wave * 16 + lc;
    const float l = xgroup_sum(l_run[mi]);
    if (qg < p.Tq) {
      const float inv = 1.f / l;
#pragma unroll
      for (int fd = 0; fd < 4; ++fd) store_row4<T, VEC>(ob + (int64_t)qg * p.ldo, fd * 16 + g * 4, p.dh, o[mi][fd], inv);
      if (g == 0) p.lse[bh * p.Tq + qg] = m_run[mi] * LN2 + logf(l);
    }
  }
}

template <typename T, int MI, bool VEC>
__global__ void __launch_bounds__(256) attn_fwd_kernel(AttnParams p) {
  attn_fwd_body<T, MI, VEC>(p);
}
// The same body held to OCC waves per SIMD (round 6).  The kernel is latency bound -- a tile is a chain global load -> LDS ->
// barrier -> scores -> softmax -> barrier -- so resident waves are what hides it: the compiler's own choice for bf16 / one query
// block per wave is 148 + 16 registers = three waves; asked for four it finds 122 without a spill and the encoder shape runs
// 41 -> 38 us, the decoder shapes 11.7 -> 10.4 and 23.1 -> 21.0, the step 12.07 -> 12.01 ms (profiles/r06_history/c31_*, c32_*).
// Five / six waves (96 / 80 registers) spill 41 - 86 registers and run 84 / 114 us.  Only this instantiation: the float32 and
// two-block forms spill at 128 registers.
template <bool VEC, int OCC>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) attn_fwd_occ_kernel(AttnParams p) {
  attn_fwd_body<bf16_t, 1, VEC>(p);
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
// dh == 64 with 16-byte-legal rows: 16 lanes x 4 elements cover one head, 64/(16*H) (b,q) rows per wave pass
template <typename T>
__global__ void attn_delta_vec_kernel(AttnParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lpr = p.H * 16, rpw = 64 / lpr;  // lanes per (b,q) row, rows per wave
  const int sub = lane / lpr, h = (lane % lpr) >> 4, d = (lane & 15) * 4;
  const int64_t rows = (int64_t)p.B * p.Tq;
  for (int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * rpw; r0 < rows; r0 += (int64_t)gridDim.x * 4 * rpw) {
    const int64_t row = r0 + sub;
    float acc = 0.f;
    if (row < rows) {
      const T* o = (const T*)p.out + row * p.ldo + h * 64 + d;
      const T* gp = (const T*)p.dout + row * p.ldo + h * 64 + d;
      if constexpr (sizeof(T) == 2) {
        const uint2 a = *reinterpret_cast<const uint2*>(o), c = *reinterpret_cast<const uint2*>(gp);
        acc = __uint_as_float(a.x << 16) * __uint_as_float(c.x << 16) +
              __uint_as_float(a.x & 0xffff0000u) * __uint_as_float(c.x & 0xffff0000u) +
              __uint_as_float(a.y << 16) * __uint_as_float(c.y << 16) +
              __uint_as_float(a.y & 0xffff0000u) * __uint_as_float(c.y & 0xffff0000u);
      } else {
        const float4 a = *reinterpret_cast<const float4*>(o), c = *reinterpret_cast<const float4*>(gp);
        acc = a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
      }
    }
    acc = dpp_add(acc, 0); acc = dpp_add(acc, 1); acc = dpp_add(acc, 2); acc = dpp_add(acc, 3);
    if (row < rows && (lane & 15) == 0) {
      const int64_t b = row / p.Tq, qg = row - b * p.Tq;
      p.delta[(b * p.H + h) * p.Tq + qg] = acc;
    }
  }
}

// keep-multipliers of the 16 probabilities (f*4+r) of one forward lane from its mask word
__device__ __forceinline__ float keep_mul(uint32_t bits, int e, float inv_keep) {
  return (bits >> e) & 1u ? inv_keep : 0.f;
}

// =============================================================================================
// backward, step 1: dK, dV.  One workgroup = 64*MI keys (wave w owns key blocks {mi*4 + w}); loops over query tiles.
//   S[q][key] = Q.K^T ; P = exp(S*scale + bias - lse[q]) ; dV^T += dO^T.Pdrop
//   dP[q][key] = dO.V^T ; dS = P o (keep*dP - delta[q]) ; dK^T += scale * Q^T.dS
// =============================================================================================
template <typename T, int MI, bool VEC, bool WDS = false>
__global__ void __launch_bounds__(256) attn_bwd_dkdv_kernel(AttnParams p) {
  typedef typename FragT<T>::type Frag;
  __shared__ __attribute__((aligned(16))) char smem[2 * AT<T>::TILE_BYTES + 2 * TR * 4];
  char* Qs = smem;                       // [q][d]
  char* Gs = smem + AT<T>::TILE_BYTES;   // dO [q][d]
  float* lss = reinterpret_cast<float*>(smem + 2 * AT<T>::TILE_BYTES);  // lse*log2e of the tile's queries
  float* dls = lss + TR;                                                // delta
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, lc = lane & 15;
  const int k0 = blockIdx.x * (TR * MI), h = blockIdx.y, b = blockIdx.z;
  const T* qb = (const T*)p.q + (int64_t)b * p.Tq * p.ldq + h * p.dh;
  const T* kb = (const T*)p.k + (int64_t)b * p.bsk + h * p.dh;
  const T* vb = (const T*)p.v + (int64_t)b * p.bsv + h * p.dh;
  const T* gb = (const T*)p.dout + (int64_t)b * p.Tq * p.ldo + h * p.dh;
  const int64_t bh = (int64_t)b * p.H + h;

  // K and V fragments (B operands: column = key) of this wave's key blocks stay in registers
  Frag kf[MI][AT<T>::NK], vf[MI][AT<T>::NK];
  float kb2[MI];
  {
    TileRegs<T> r0, r1;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      tile_load<T, VEC>(r0, kb, p.ldk, k0 + mi * TR, p.Tk, p.dh, tid);
      tile_load<T, VEC>(r1, vb, p.ldv, k0 + mi * TR, p.Tk, p.dh, tid);
      if (mi) __syncthreads();
      tile_store<T>(Qs, r0, tid);
      tile_store<T>(Gs, r1, tid);
      __syncthreads();
#pragma unroll
      for (int s = 0; s < AT<T>::NK; ++s) {
        kf[mi][s] = rc_frag<T>(Qs, wave * 16, s * AT<T>::KS, lane);
        vf[mi][s] = rc_frag<T>(Gs, wave * 16, s * AT<T>::KS, lane);
      }
      kb2[mi] = key_bias2(p, b, k0 + mi * TR + wave * 16 + lc);
    }
    __syncthreads();
  }

  floatx4_t dk[MI][4], dv[MI][4];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int f = 0; f < 4; ++f) { dk[mi][f] = floatx4_t{0.f, 0.f, 0.f, 0.f}; dv[mi][f] = floatx4_t{0.f, 0.f, 0.f, 0.f}; }

  const int nqt = (p.Tq + TR - 1) / TR;
  const int qt_first = (p.causal && k0 > p.coff) ? (k0 - p.coff) / TR : 0;  // queries before the first key of the block never see it
  // mask words this lane reads: forward lanes (g*4 + r) + 16*(lc>>2), r = 0..3 -> 4 consecutive u16; bit = f_key*4 + (lc&3)
  const int mlane = g * 4 + 16 * (lc >> 2);
  const int mbit = wave * 4 + (lc & 3);

  // Software pipeline over the query tiles.  vmcnt counts loads and stores in order, so everything this loop waits for
  // is issued a full tile ahead: Q / dO / lse / delta and the dropout-mask words of tile t+1 at the top of tile t, and the
  // dS^T stores of tile t at the top of tile t+1 (issued right behind the loads they would otherwise sit in front of: the
  // next wait then finds stores that have had a whole tile to retire instead of ones issued a moment ago).
  TileRegs<T> qreg, greg;
  float lsreg = 0.f, dlreg = 0.f;
  uint2 mwn[MI][4];            // mask words of the NEXT tile
  uint2 dsp[MI][4];            // packed dS^T of the PREVIOUS tile, not yet stored
  bf16_t* dsrow[MI];
  bool ds_pending = false;
  auto mask_load = [&](int q0n) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int kblk0 = k0 + mi * TR + wave * 16;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        mwn[mi][f] = make_uint2(0xffffffffu, 0xffffffffu);
        if (p.drop_thresh && (q0n >> 4) + f < p.nqb)
          mwn[mi][f] = *reinterpret_cast<const uint2*>(
              p.mask + ((bh * p.nqb + ((q0n >> 4) + f)) * p.nkt + (kblk0 >> 6)) * 64 + mlane);
      }
    }
  };
  auto stat_load = [&](int q) {
    const bool ok = q < p.Tq;
    lsreg = ok ? p.lse[bh * p.Tq + q] : INFINITY;   // raw: the scale by log2(e) waits until the value is needed
    dlreg = ok ? p.delta[bh * p.Tq + q] : 0.f;
  };
  if (qt_first < nqt) {
    tile_load<T, VEC>(qreg, qb, p.ldq, qt_first * TR, p.Tq, p.dh, tid);
    tile_load<T, VEC>(greg, gb, p.ldo, qt_first * TR, p.Tq, p.dh, tid);
    if (tid < TR) stat_load(qt_first * TR + tid);
    mask_load(qt_first * TR);
  }
  for (int qt = qt_first; qt < nqt; ++qt) {
    const int q0 = qt * TR;
    tile_store<T>(Qs, qreg, tid);
    tile_store<T>(Gs, greg, tid);
    if (tid < TR) { lss[tid] = lsreg * LOG2E; dls[tid] = dlreg; }
    uint2 mwc[MI][4];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int f = 0; f < 4; ++f) mwc[mi][f] = mwn[mi][f];
    __syncthreads();
    if constexpr (WDS && sizeof(T) == 2) {
      if (ds_pending) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int f = 0; f < 4; ++f) *reinterpret_cast<uint2*>(dsrow[mi] + f * 16) = dsp[mi][f];
      }
    }
    if (qt + 1 < nqt) {
      tile_load<T, VEC>(qreg, qb, p.ldq, q0 + TR, p.Tq, p.dh, tid);
      tile_load<T, VEC>(greg, gb, p.ldo, q0 + TR, p.Tq, p.dh, tid);
      if (tid < TR) stat_load(q0 + TR + tid);
      mask_load(q0 + TR);
    }
    floatx4_t ls4[4], dl4[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      ls4[f] = *reinterpret_cast<const floatx4_t*>(lss + f * 16 + g * 4);
      dl4[f] = *reinterpret_cast<const floatx4_t*>(dls + f * 16 + g * 4);
    }
    floatx4_t st[MI][4], dp[MI][4];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int f = 0; f < 4; ++f) { st[mi][f] = floatx4_t{0.f, 0.f, 0.f, 0.f}; dp[mi][f] = floatx4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int s = 0; s < AT<T>::NK; ++s)
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const Frag aq = rc_frag<T>(Qs, f * 16, s * AT<T>::KS, lane);
        const Frag ag = rc_frag<T>(Gs, f * 16, s * AT<T>::KS, lane);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          st[mi][f] = Mma<T>::run(aq, kf[mi][s], st[mi][f]);
          dp[mi][f] = Mma<T>::run(ag, vf[mi][s], dp[mi][f]);
        }
      }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int kblk0 = k0 + mi * TR + wave * 16;
      const int kg = kblk0 + lc;
      const bool diag = p.causal && (kblk0 + 15 > q0 + p.coff);
      auto elems = [&](auto DIAG, auto DROP) {
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const uint2 mw = mwc[mi][f];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float pv = prob_exp2(fmaf(st[mi][f][r], p.scale2, kb2[mi] - ls4[f][r]));
            if (decltype(DIAG)::value && (kg > q0 + f * 16 + g * 4 + r + p.coff)) pv = 0.f;
            float keep = 1.f;
            if (decltype(DROP)::value) keep = keep_mul(r < 2 ? mw.x : mw.y, (r & 1) * 16 + mbit, p.drop_inv_keep);
            st[mi][f][r] = pv * keep;                                            // dropped P, feeds dV
            dp[mi][f][r] = pv * (keep * dp[mi][f][r] - dl4[f][r]) * p.scale;      // dS (scaled), feeds dK
          }
        }
      };
      if (p.drop_thresh) { if (diag) elems(std::true_type{}, std::true_type{}); else elems(std::false_type{}, std::true_type{}); }
      else { if (diag) elems(std::true_type{}, std::false_type{}); else elems(std::false_type{}, std::false_type{}); }
      if constexpr (WDS && sizeof(T) == 2) {
        // dS^T[key][q] for the dQ kernel: this lane's key row, 4 consecutive queries per 16-query block (8-byte stores;
        // the four lane groups of a key cover 32 contiguous bytes).  Rows / columns beyond Tk / Tq hold exact zeros.
        dsrow[mi] = p.dst + ((int64_t)bh * p.tkp + kg) * p.tqp + q0 + g * 4;
#pragma unroll
        for (int f = 0; f < 4; ++f)
          dsp[mi][f] = make_uint2(pack_bf16x2(dp[mi][f][0], dp[mi][f][1]), pack_bf16x2(dp[mi][f][2], dp[mi][f][3]));
        ds_pending = true;
      }
    }
    tmul_acc<T, MI>(dv, st, Gs, lane);
    tmul_acc<T, MI>(dk, dp, Qs, lane);
    __syncthreads();
  }
  if constexpr (WDS && sizeof(T) == 2) {
    if (ds_pending) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int f = 0; f < 4; ++f) *reinterpret_cast<uint2*>(dsrow[mi] + f * 16) = dsp[mi][f];
    }
  }

  T* dkb = (T*)p.dk + (int64_t)b * p.bsk + h * p.dh;
  T* dvb = (T*)p.dv + (int64_t)b * p.bsv + h * p.dh;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int kg = k0 + mi * TR + wave * 16 + lc;
    if (kg < p.Tk) {
#pragma unroll
      for (int fd = 0; fd < 4; ++fd) {
        store_row4<T, VEC>(dkb + (int64_t)kg * p.ldk, fd * 16 + g * 4, p.dh, dk[mi][fd], 1.f);
        store_row4<T, VEC>(dvb + (int64_t)kg * p.ldv, fd * 16 + g * 4, p.dh, dv[mi][fd], 1.f);
      }
    }
  }
}

// =============================================================================================
// backward, step 2 (bf16 with a dS workspace): dQ^T[d][q] = sum_key K^T[d][key] . dS^T[key][q] from the (already scaled)
// dS^T the dK/dV kernel left in the workspace.  No score recomputation, no exp, no mask: the kernel that used to redo both
// products and all the element-wise work of the first pass (VALU-issue bound) becomes 8 MFMAs per 64-key tile and wave fed by
// transpose reads.  One workgroup = 64 queries (wave w: 16 of them), loops over the 64-key tiles; both operands are staged
// [key][.] row-major and read with ds_read_b64_tr_b16 (the reduction index runs over LDS rows on both sides).
// =============================================================================================
template <bool VEC>
__global__ void __launch_bounds__(256) attn_bwd_dq_from_ds_kernel(AttnParams p) {
  typedef bf16_t T;
  __shared__ __attribute__((aligned(16))) char smem[2 * AT<T>::TILE_BYTES];
  char* Ks = smem;                       // [key][d]
  char* Ds = smem + AT<T>::TILE_BYTES;   // dS^T [key][q]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, lc = lane & 15;
  const int q0 = blockIdx.x * TR, h = blockIdx.y, b = blockIdx.z;
  const T* kb = (const T*)p.k + (int64_t)b * p.bsk + h * p.dh;
  const int64_t bh = (int64_t)b * p.H + h;
  const T* db = p.dst + bh * p.tkp * (int64_t)p.tqp + q0;   // column block q0.. of every key row
  floatx4_t dq[1][4];
#pragma unroll
  for (int f = 0; f < 4; ++f) dq[0][f] = floatx4_t{0.f, 0.f, 0.f, 0.f};
  const int qt = blockIdx.x;
  // key tiles whose dK/dV workgroup processed this query tile (that kernel skips query tiles wholly above the causal
  // diagonal of its key block: their dS is zero and was never written)
  auto tile_written = [&](int kt) {
    const int k0 = (kt * TR / p.ds_kblock) * p.ds_kblock;
    return !(p.causal && k0 > p.coff) || qt >= (k0 - p.coff) / TR;
  };
  int nkt = p.nkt;
  if (p.causal) { const int lim = (q0 + TR - 1 + p.coff) / TR + 1; if (lim < nkt) nkt = lim; }
  TileRegs<T> kreg, dreg;
  tile_load<T, VEC>(kreg, kb, p.ldk, 0, p.Tk, p.dh, tid);
  tile_load<T, true>(dreg, db, p.tqp, 0, p.tkp, TR, tid);
  for (int kt = 0; kt < nkt; ++kt) {
    tile_store<T>(Ks, kreg, tid);
    tile_store<T>(Ds, dreg, tid);
    __syncthreads();
    if (kt + 1 < nkt) {
      tile_load<T, VEC>(kreg, kb, p.ldk, (kt + 1) * TR, p.Tk, p.dh, tid);
      tile_load<T, true>(dreg, db, p.tqp, (kt + 1) * TR, p.tkp, TR, tid);
    }
    if (tile_written(kt)) {   // workgroup-uniform
      typedef short4_t __attribute__((address_space(3))) * lds_ptr_t;
      const int roff = (g * 4 + (lc >> 2)) * AT<T>::RS + (lc & 3) * 8;
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        union { short4_t hh[2]; bf16x8_t f; } bfr;
        const char* pb = Ds + roff + st * 32 * AT<T>::RS + wave * 32;
        bfr.hh[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(pb));
        bfr.hh[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(pb + 16 * AT<T>::RS));
#pragma unroll
        for (int fd = 0; fd < 4; ++fd) {
          union { short4_t hh[2]; bf16x8_t f; } afr;
          const char* pa = Ks + roff + st * 32 * AT<T>::RS + fd * 32;
          afr.hh[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(pa));
          afr.hh[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(pa + 16 * AT<T>::RS));
          dq[0][fd] = Mma<T>::run(afr.f, bfr.f, dq[0][fd]);
        }
      }
    }
    __syncthreads();
  }
  T* dqb = (T*)p.dq + (int64_t)b * p.Tq * p.ldq + h * p.dh;
  const int qg = q0 + wave * 16 + lc;
  if (qg < p.Tq) {
#pragma unroll
    for (int fd = 0; fd < 4; ++fd) store_row4<T, VEC>(dqb + (int64_t)qg * p.ldq, fd * 16 + g * 4, p.dh, dq[0][fd], 1.f);
  }
}

// =============================================================================================
// backward, one workgroup per (batch, head) (bf16, Tq <= 256 and Tk <= 256; round 4): dK, dV AND dQ from one kernel, EIGHT waves.
//
// The two-kernel path above moves the scaled dS^T through HBM (B*H*Tk'*Tq' bf16 written by the dK/dV kernel, read by the dQ
// kernel: 134 MB of the ~240 MB an encoder layer's attention backward touches at the benchmark shape -- the launch pair runs at
// 5.7 TB/s, it is HBM-bound on traffic the algorithm does not need) and re-reads Q / dO once per key block.  Round 3's first
// fused form (four waves, one per SIMD, Q / dO / K resident in 140 KB of LDS) removed the traffic but walked its 16 dependent
// steps with nothing to hide their latency behind (82 us against 86).  This form:
//   * wave w of 8 owns keys [32 w, 32 w + 32) for the whole kernel -- K and V fragments, the key bias and the dK / dV
//     accumulators of those keys live in its registers (two waves per SIMD: one computes while the other waits);
//   * K of the head stays in LDS (the A operand of dQ = dS . K for EVERY wave), Q and dO stream through it in tiles of 32
//     queries: one 16-byte chunk per thread and tile (threads 0..255 Q, 256..511 dO), loaded a tile ahead into a register,
//     written to the other half of a double buffer before the tile's single barrier;
//   * per tile and wave: S and dP (8 + 8 MFMAs), the element-wise part, dV += P^T dO and dK += dS^T Q (8 + 8), the scaled
//     dS^T of its 32 keys x 32 queries into a double-buffered LDS tile, BARRIER, then its 16 x 16 block of
//     dQ^T = K^T . dS^T over all 256 keys (8 MFMAs) -- stored at once, no accumulator survives the tile.
// 40 MFMAs per wave and tile, one barrier per tile, ~113 KB of LDS, B*H workgroups.
// Encoder shape (B = 128, H = 4, T = 225): 52.7 us against 87 us for the two kernels.  Timing ablations (profiles/r04_history/
// c13 / c14_attn_ablation.log): loads + staging + barriers alone 18.5 us, + stores 28 us (the 107 MB a launch moves at ~4.5 TB/s:
// one workgroup per CU, so a workgroup's loads, compute and stores do not overlap with a neighbour's), element-wise part ~9 us,
// the five products ~12 us -- the compute part is VALU-issue bound (16 probabilities per lane and tile, ~11 instructions each).
// =============================================================================================
constexpr int HB_MAXT = 256, HB_QT = 32;
// Row strides chosen for the TRANSPOSE reads, which outnumber the row reads 6 : 1 here: a ds_read_b64_tr_b16 serves 32 lanes
// per pass = 8 rows x 32 bytes, so a stride of 8 dwords mod 64 (160 B, 96 B) spreads them over all 64 banks; with the
// 144-byte rows of the other kernels two of the eight rows share four banks (PMC: 45 % of the LDS cycles were conflicts)
constexpr int HB_RS = 160;                               // K / Q / dO images: 64 head dims (128 B) + 32
constexpr int HB_RSD = HB_QT * 2 + 32;                   // 96: dS^T image rows (32 queries + 32)
constexpr int HB_K = 0;                                  // K image [256][144]
constexpr int HB_QG = HB_K + HB_MAXT * HB_RS;            // two buffers of { Q tile [32][144] | dO tile [32][144] }
constexpr int HB_QG_BUF = 2 * HB_QT * HB_RS;
constexpr int HB_D = HB_QG + 2 * HB_QG_BUF;              // two dS^T images [256][80]
constexpr int HB_D_BUF = HB_MAXT * HB_RSD;
constexpr int HB_STAT = HB_D + 2 * HB_D_BUF;             // lse * log2(e) [256], delta [256]
constexpr int HB_LDS_BYTES = HB_STAT + 2 * HB_MAXT * 4;

__device__ __forceinline__ bf16x8_t hb_frag(const char* tile, int row0, int kk, int lane) {   // rc_frag on HB_RS rows
  return *reinterpret_cast<const bf16x8_t*>(tile + (row0 + (lane & 15)) * HB_RS + (kk + (lane >> 4) * 8) * 2);
}

__global__ void __launch_bounds__(512, 2) attn_bwd_head8_kernel(AttnParams p) {
  typedef bf16_t T;
  typedef FragT<T>::type Frag;
  typedef short4_t __attribute__((address_space(3))) * lds_ptr_t;
  extern __shared__ __attribute__((aligned(16))) char hsm[];
  float* lss = reinterpret_cast<float*>(hsm + HB_STAT);
  float* dls = lss + HB_MAXT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, lc = lane & 15;
  const int h = blockIdx.x, b = blockIdx.y;
  const T* qb = (const T*)p.q + (int64_t)b * p.Tq * p.ldq + h * p.dh;
  const T* kb = (const T*)p.k + (int64_t)b * p.bsk + h * p.dh;
  const T* vb = (const T*)p.v + (int64_t)b * p.bsv + h * p.dh;
  const T* gb = (const T*)p.dout + (int64_t)b * p.Tq * p.ldo + h * p.dh;
  const int64_t bh = (int64_t)b * p.H + h;
  const int nqt = (p.Tq + HB_QT - 1) / HB_QT;
  const int kw0 = wave * 32;                     // this wave's first key

  // Q / dO tile t: thread -> (which operand, row, 16-byte chunk); one chunk per thread
  const int which = tid >> 8, crow = (tid & 255) >> 3, cchunk = tid & 7;
  const T* qg_src = which ? gb : qb;
  const int64_t qg_ld = which ? p.ldo : p.ldq;
  // every global load below is UNCONDITIONAL at a clamped (always valid) address, out-of-range pieces are zeroed by a select on
  // the VALUE: a load under a branch makes hipcc wait vmcnt(0) right behind it (prologue: eight serialized memory round trips)
  // and again at the loop's back edge (a full round trip per query tile) -- MI355X guide, "three .s-level traps" (c)
  const int ccol = cchunk * 8 < p.dh ? cchunk * 8 : 0;
  auto qg_load = [&](int t) -> uint4 {
    const int row = t * HB_QT + crow, rr = row < p.Tq ? row : p.Tq - 1;
    const uint4 v = *reinterpret_cast<const uint4*>(qg_src + (int64_t)rr * qg_ld + ccol);
    const bool ok = row < p.Tq && cchunk * 8 < p.dh;
    return make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
  };
  auto qg_store = [&](int buf, const uint4& v) {
    *reinterpret_cast<uint4*>(hsm + HB_QG + buf * HB_QG_BUF + which * (HB_QT * HB_RS) + crow * HB_RS + cchunk * 16) = v;
  };

  // ---------------------------------------------------------------- prologue
  uint4 qg_next = qg_load(0);
  uint4 kv[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {                  // K image: 256 rows x 8 chunks, 4 per thread (rows beyond Tk: zeros)
    const int c = tid + s * 512, row = c >> 3, ch = c & 7;
    const int rr = row < p.Tk ? row : p.Tk - 1, cc = ch * 8 < p.dh ? ch * 8 : 0;
    kv[s] = *reinterpret_cast<const uint4*>(kb + (int64_t)rr * p.ldk + cc);
  }
  Frag kf[2][2], vf[2][2];
  float kb2[2];
  uint4 vraw[2][2];
  float kbraw[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {               // V fragments (B operand: column = key) straight from HBM
    const int row = kw0 + mi * 16 + lc, rr = row < p.Tk ? row : p.Tk - 1;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int col = s2 * 32 + g * 8;
      vraw[mi][s2] = *reinterpret_cast<const uint4*>(vb + (int64_t)rr * p.ldv + (col < p.dh ? col : 0));
    }
    const float* kbp = p.key_bias ? p.key_bias + (int64_t)b * p.Tk + rr : p.lse;     // (any readable float when there is no bias)
    kbraw[mi] = *kbp;
  }
  const int sidx = tid & (HB_MAXT - 1), sq = sidx < p.Tq ? sidx : p.Tq - 1;
  const float sraw = tid < HB_MAXT ? p.lse[bh * p.Tq + sq] : p.delta[bh * p.Tq + sq];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int c = tid + s * 512, row = c >> 3, ch = c & 7;
    const bool ok = row < p.Tk && ch * 8 < p.dh;
    *reinterpret_cast<uint4*>(hsm + HB_K + row * HB_RS + ch * 16) =
        make_uint4(ok ? kv[s].x : 0u, ok ? kv[s].y : 0u, ok ? kv[s].z : 0u, ok ? kv[s].w : 0u);
  }
  if (tid < HB_MAXT) lss[sidx] = sidx < p.Tq ? sraw * LOG2E : INFINITY;
  else dls[sidx] = sidx < p.Tq ? sraw : 0.f;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int row = kw0 + mi * 16 + lc;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const bool ok = row < p.Tk && s2 * 32 + g * 8 < p.dh;
      const uint4 v = vraw[mi][s2];
      vf[mi][s2] = __builtin_bit_cast(Frag, make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u));
    }
    kb2[mi] = row >= p.Tk ? -INFINITY : (p.key_bias ? fmaxf(kbraw[mi] * LOG2E, -3.0e38f) : 0.f);
  }
  qg_store(0, qg_next);
  __syncthreads();
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) kf[mi][s2] = hb_frag(hsm + HB_K, kw0 + mi * 16, s2 * AT<T>::KS, lane);

  floatx4_t dk[2][4], dv[2][4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int f = 0; f < 4; ++f) { dk[mi][f] = floatx4_t{0.f, 0.f, 0.f, 0.f}; dv[mi][f] = floatx4_t{0.f, 0.f, 0.f, 0.f}; }

  const int mlane = g * 4 + 16 * (lc >> 2);
  const int roff = g * 4 + (lc >> 2);            // transpose reads: this lane's row inside a 16-row half step ...
  const int coff8 = (lc & 3) * 8;                // ... and its 8-byte chunk
  const int fd_q = wave & 3, fq_q = wave >> 2;   // this wave's 16 x 16 block of dQ^T: head dims 16 fd_q .., queries 16 fq_q ..
  T* dqb = (T*)p.dq + (int64_t)b * p.Tq * p.ldq + h * p.dh;

  // dropout keep words of tile t (two 16-query blocks x this wave's two 16-key blocks), loaded a tile ahead
  uint2 mwn[2][2], mwc[2][2];
  const bool mask_off = p.drop_thresh == 0;
  const uint16_t* mbase = mask_off ? reinterpret_cast<const uint16_t*>(p.lse)      // (no dropout: 8 readable bytes, unused)
                                   : p.mask + bh * p.nqb * (int64_t)p.nkt * 64 + mlane;
  auto mask_load = [&](int t) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        int qblk = t * 2 + f, ktile = (kw0 + mi * 16) >> 6;
        qblk = qblk < p.nqb ? qblk : p.nqb - 1;
        ktile = ktile < p.nkt ? ktile : p.nkt - 1;
        mwn[mi][f] = *reinterpret_cast<const uint2*>(mbase + (mask_off ? 0 : (qblk * p.nkt + ktile) * 64));
      }
  };
  mask_load(0);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int f = 0; f < 2; ++f) mwc[mi][f] = mwn[mi][f];

  for (int t = 0; t < nqt; ++t) {
    const int q0 = t * HB_QT, buf = t & 1;
    const char* Qt = hsm + HB_QG + buf * HB_QG_BUF;
    const char* Gt = Qt + HB_QT * HB_RS;
    char* Dt = hsm + HB_D + buf * HB_D_BUF;
    // the next tile's Q / dO chunk and dropout words: issued here, consumed at the END of this iteration -- no load is pending
    // across the loop's back edge (where hipcc would drain vmcnt(0), the dQ stores included)
    qg_next = qg_load(t + 1 < nqt ? t + 1 : t);
    mask_load(t + 1 < nqt ? t + 1 : t);
    floatx4_t ls4[2], dl4[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      ls4[f] = *reinterpret_cast<const floatx4_t*>(lss + q0 + f * 16 + g * 4);
      dl4[f] = *reinterpret_cast<const floatx4_t*>(dls + q0 + f * 16 + g * 4);
    }
    // ---- S[q][key] = Q K^T and dP[q][key] = dO V^T for 32 queries x this wave's 32 keys
    floatx4_t st[2][2], dp[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int f = 0; f < 2; ++f) { st[mi][f] = floatx4_t{0.f, 0.f, 0.f, 0.f}; dp[mi][f] = floatx4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const Frag aq = hb_frag(Qt, f * 16, s2 * AT<T>::KS, lane);
        const Frag ag = hb_frag(Gt, f * 16, s2 * AT<T>::KS, lane);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          st[mi][f] = Mma<T>::run(aq, kf[mi][s2], st[mi][f]);
          dp[mi][f] = Mma<T>::run(ag, vf[mi][s2], dp[mi][f]);
        }
      }
    // ---- P, dropped P, scaled dS; dS^T of the wave's keys into the exchange image
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int kblk0 = kw0 + mi * 16, kg = kblk0 + lc;
      const int mbit = ((kblk0 >> 4) & 3) * 4 + (lc & 3);
      const bool diag = p.causal && (kblk0 + 15 > q0 + p.coff);
      auto elems = [&](auto DIAG, auto DROP) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const uint2 mw = mwc[mi][f];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float pv = prob_exp2(fmaf(st[mi][f][r], p.scale2, kb2[mi] - ls4[f][r]));
            if (decltype(DIAG)::value && (kg > q0 + f * 16 + g * 4 + r + p.coff)) pv = 0.f;
            float keep = 1.f;
            if (decltype(DROP)::value) keep = keep_mul(r < 2 ? mw.x : mw.y, (r & 1) * 16 + mbit, p.drop_inv_keep);
            st[mi][f][r] = pv * keep;                                           // dropped P, feeds dV
            dp[mi][f][r] = pv * (keep * dp[mi][f][r] - dl4[f][r]) * p.scale;     // dS (scaled), feeds dK and dQ
          }
        }
      };
      if (p.drop_thresh) { if (diag) elems(std::true_type{}, std::true_type{}); else elems(std::false_type{}, std::true_type{}); }
      else { if (diag) elems(std::true_type{}, std::false_type{}); else elems(std::false_type{}, std::false_type{}); }
#pragma unroll
      for (int f = 0; f < 2; ++f)
        *reinterpret_cast<uint2*>(Dt + (kblk0 + lc) * HB_RSD + (f * 16 + g * 4) * 2) =
            make_uint2(pack_bf16x2(dp[mi][f][0], dp[mi][f][1]), pack_bf16x2(dp[mi][f][2], dp[mi][f][3]));
    }
    // ---- dV^T[d][key] += dO^T[d][q] . P[q][key],  dK^T[d][key] += Q^T[d][q] . dS[q][key]   (one 32-query step)
    {
      bf16x8_t bp[2], bs[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        union { uint32_t u[4]; bf16x8_t f; } u0, u1;
        u0.u[0] = pack_bf16x2(st[mi][0][0], st[mi][0][1]); u0.u[1] = pack_bf16x2(st[mi][0][2], st[mi][0][3]);
        u0.u[2] = pack_bf16x2(st[mi][1][0], st[mi][1][1]); u0.u[3] = pack_bf16x2(st[mi][1][2], st[mi][1][3]);
        u1.u[0] = pack_bf16x2(dp[mi][0][0], dp[mi][0][1]); u1.u[1] = pack_bf16x2(dp[mi][0][2], dp[mi][0][3]);
        u1.u[2] = pack_bf16x2(dp[mi][1][0], dp[mi][1][1]); u1.u[3] = pack_bf16x2(dp[mi][1][2], dp[mi][1][3]);
        bp[mi] = u0.f; bs[mi] = u1.f;
      }
      const char* gbase = Gt + roff * HB_RS + coff8;
      const char* qbase = Qt + roff * HB_RS + coff8;
#pragma unroll
      for (int fd = 0; fd < 4; ++fd) {
        union { short4_t hh[2]; bf16x8_t f; } ag, aq;
        ag.hh[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(gbase + fd * 32));
        ag.hh[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(gbase + fd * 32 + 16 * HB_RS));
        aq.hh[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(qbase + fd * 32));
        aq.hh[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(qbase + fd * 32 + 16 * HB_RS));
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          dv[mi][fd] = Mma<T>::run(ag.f, bp[mi], dv[mi][fd]);
          dk[mi][fd] = Mma<T>::run(aq.f, bs[mi], dk[mi][fd]);
        }
      }
    }
    // ---- the next tile's Q / dO go to the other buffer (every wave finished reading it before the previous barrier)
    qg_store(buf ^ 1, qg_next);                         // (behind the last tile: a copy nobody reads)
    // dS^T of all 256 keys (and the next Q / dO tile) are in LDS.  A raw barrier behind lgkmcnt(0) only: __syncthreads() also
    // drains vmcnt, i.e. waits at EVERY tile for the Q / dO prefetch issued a moment ago and for the dQ stores of the previous
    // tile to be acknowledged -- a full memory round trip per tile (the ablation: 22 of 54 us were this skeleton, 12 the stores)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // the next tile's dropout words become current HERE, in front of the dQ stores: behind them the copy would wait for the
    // stores' acknowledgement as well (a store under `if (row < Tq)` makes hipcc's count conservative: vmcnt(0))
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int f = 0; f < 2; ++f) { mwc[mi][f] = mwn[mi][f]; asm volatile("" : "+v"(mwc[mi][f].x), "+v"(mwc[mi][f].y)); }
    // ---- dQ^T[d][q] = K^T[d][key] . dS^T[key][q] over the 256 keys: this wave's 16 head dims x 16 queries
    {
      floatx4_t dq = floatx4_t{0.f, 0.f, 0.f, 0.f};
      const char* ka = hsm + HB_K + roff * HB_RS + coff8 + fd_q * 32;
      const char* da = Dt + roff * HB_RSD + coff8 + fq_q * 32;
#pragma unroll
      for (int ks = 0; ks < HB_MAXT / 32; ++ks) {
        union { short4_t hh[2]; bf16x8_t f; } af, bf;
        af.hh[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(ka + ks * 32 * HB_RS));
        af.hh[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(ka + ks * 32 * HB_RS + 16 * HB_RS));
        bf.hh[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(da + ks * 32 * HB_RSD));
        bf.hh[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(da + ks * 32 * HB_RSD + 16 * HB_RSD));
        dq = Mma<T>::run(af.f, bf.f, dq);
      }
      const int qg = q0 + fq_q * 16 + lc;
      if (qg < p.Tq) store_row4<T, true>(dqb + (int64_t)qg * p.ldq, fd_q * 16 + g * 4, p.dh, dq, 1.f);
    }
  }

  T* dkb = (T*)p.dk + (int64_t)b * p.bsk + h * p.dh;
  T* dvb = (T*)p.dv + (int64_t)b * p.bsv + h * p.dh;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int kg = kw0 + mi * 16 + lc;
    if (kg < p.Tk) {
#pragma unroll
      for (int fd = 0; fd < 4; ++fd) {
        store_row4<T, true>(dkb + (int64_t)kg * p.ldk, fd * 16 + g * 4, p.dh, dk[mi][fd], 1.f);
        store_row4<T, true>(dvb + (int64_t)kg * p.ldv, fd * 16 + g * 4, p.dh, dv[mi][fd], 1.f);
      }
    }
  }
}

// =============================================================================================
// backward, step 2: dQ.  Same blocking as the forward kernel.
//   S^T = K.Q^T ; P^T = exp(S^T*scale + bias - lse[q]) ; dP^T = V.dO^T ; dS^T = P^T o (keep*dP^T - delta[q])
//   dQ^T += scale * K^T.dS^T
// =============================================================================================
template <typename T, int MI, bool VEC>
__global__ void __launch_bounds__(256) attn_bwd_dq_kernel(AttnParams p) {
  typedef typename FragT<T>::type Frag;
  __shared__ __attribute__((aligned(16))) char smem[2 * AT<T>::TILE_BYTES + TR * 4];
  char* Ks = smem;
  char* Vs = smem + AT<T>::TILE_BYTES;
  float* kbs = reinterpret_cast<float*>(smem + 2 * AT<T>::TILE_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, lc = lane & 15;
  const int q0 = blockIdx.x * (TR * MI), h = blockIdx.y, b = blockIdx.z;
  const T* qb = (const T*)p.q + (int64_t)b * p.Tq * p.ldq + h * p.dh;
  const T* kb = (const T*)p.k + (int64_t)b * p.bsk + h * p.dh;
  const T* vb = (const T*)p.v + (int64_t)b * p.bsv + h * p.dh;
  const T* gb = (const T*)p.dout + (int64_t)b * p.Tq * p.ldo + h * p.dh;
  const int64_t bh = (int64_t)b * p.H + h;

  Frag qf[MI][AT<T>::NK], gf[MI][AT<T>::NK];
  float ls2[MI], dl[MI];
  {
    TileRegs<T> r0, r1;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      tile_load<T, VEC>(r0, qb, p.ldq, q0 + mi * TR, p.Tq, p.dh, tid);
      tile_load<T, VEC>(r1, gb, p.ldo, q0 + mi * TR, p.Tq, p.dh, tid);
      if (mi) __syncthreads();
      tile_store<T>(Ks, r0, tid);
      tile_store<T>(Vs, r1, tid);
      __syncthreads();
#pragma unroll
      for (int s = 0; s < AT<T>::NK; ++s) {
        qf[mi][s] = rc_frag<T>(Ks, wave * 16, s * AT<T>::KS, lane);
        gf[mi][s] = rc_frag<T>(Vs, wave * 16, s * AT<T>::KS, lane);
      }
      const int qg = q0 + mi * TR + wave * 16 + lc;
      const bool ok = qg < p.Tq;
      ls2[mi] = ok ? p.lse[bh * p.Tq + qg] * LOG2E : INFINITY;
      dl[mi] = ok ? p.delta[bh * p.Tq + qg] : 0.f;
    }
    __syncthreads();
  }
  floatx4_t dq[MI][4];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int f = 0; f < 4; ++f) dq[mi][f] = floatx4_t{0.f, 0.f, 0.f, 0.f};

  int nkt = p.nkt;
  if (p.causal) { const int lim = (q0 + TR * MI - 1 + p.coff) / TR + 1; if (lim < nkt) nkt = lim; }
  TileRegs<T> kreg, vreg;
  float kbreg = 0.f;
  tile_load<T, VEC>(kreg, kb, p.ldk, 0, p.Tk, p.dh, tid);
  tile_load<T, VEC>(vreg, vb, p.ldv, 0, p.Tk, p.dh, tid);
  if (tid < TR) kbreg = key_bias2(p, b, tid);
  for (int kt = 0; kt < nkt; ++kt) {
    const int k0 = kt * TR;
    tile_store<T>(Ks, kreg, tid);
    tile_store<T>(Vs, vreg, tid);
    if (tid < TR) kbs[tid] = kbreg;
    __syncthreads();
    if (kt + 1 < nkt) {
      tile_load<T, VEC>(kreg, kb, p.ldk, k0 + TR, p.Tk, p.dh, tid);
      tile_load<T, VEC>(vreg, vb, p.ldv, k0 + TR, p.Tk, p.dh, tid);
      if (tid < TR) kbreg = key_bias2(p, b, k0 + TR + tid);
    }
    floatx4_t kb4[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) kb4[f] = *reinterpret_cast<const floatx4_t*>(kbs + f * 16 + g * 4);
    floatx4_t s[MI][4], dp[MI][4];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int f = 0; f < 4; ++f) { s[mi][f] = floatx4_t{0.f, 0.f, 0.f, 0.f}; dp[mi][f] = floatx4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int st = 0; st < AT<T>::NK; ++st)
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const Frag ak = rc_frag<T>(Ks, f * 16, st * AT<T>::KS, lane);
        const Frag av = rc_frag<T>(Vs, f * 16, st * AT<T>::KS, lane);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          s[mi][f] = Mma<T>::run(ak, qf[mi][st], s[mi][f]);
          dp[mi][f] = Mma<T>::run(av, gf[mi][st], dp[mi][f]);
        }
      }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int qblk0 = q0 + mi * TR + wave * 16;
      const int qg = qblk0 + lc;
      const bool diag = p.causal && (k0 + TR - 1 > qblk0 + p.coff);
      uint32_t bits = 0xffffu;
      if (p.drop_thresh && qblk0 < p.Tq) bits = p.mask[((bh * p.nqb + (qblk0 >> 4)) * p.nkt + kt) * 64 + lane];
      auto elems = [&](auto DIAG, auto DROP) {
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float pv = prob_exp2(fmaf(s[mi][f][r], p.scale2, kb4[f][r] - ls2[mi]));
            if (decltype(DIAG)::value && (k0 + f * 16 + g * 4 + r > qg + p.coff)) pv = 0.f;
            float keep = 1.f;
            if (decltype(DROP)::value) keep = keep_mul(bits, f * 4 + r, p.drop_inv_keep);
            s[mi][f][r] = pv * (keep * dp[mi][f][r] - dl[mi]) * p.scale;
          }
      };
      if (p.drop_thresh) { if (diag) elems(std::true_type{}, std::true_type{}); else elems(std::false_type{}, std::true_type{}); }
      else { if (diag) elems(std::true_type{}, std::false_type{}); else elems(std::false_type{}, std::false_type{}); }
    }
    tmul_acc<T, MI>(dq, s, Ks, lane);
    __syncthreads();
  }
  T* dqb = (T*)p.dq + (int64_t)b * p.Tq * p.ldq + h * p.dh;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int qg = q0 + mi * TR + wave * 16 + lc;
    if (qg < p.Tq) {
#pragma unroll
      for (int fd = 0; fd < 4; ++fd) store_row4<T, VEC>(dqb + (int64_t)qg * p.ldq, fd * 16 + g * 4, p.dh, dq[mi][fd], 1.f);
    }
  }
}

int vec_legal(const void* base, int64_t ld, int dh, int esz) {
  const int E = 16 / esz;
  return nst_aligned16(base) && ((ld * esz) % 16 == 0) && (dh % E == 0);
}

int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return (e && e[0]) ? atoi(e) : dflt;
}

int64_t mask_bytes(const NstAttnDesc* d) {
  return (int64_t)d->B * d->H * ((d->Tq + 15) / 16) * ((d->Tk + TR - 1) / TR) * 64 * 2;
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
  p.nqb = (d->Tq + 15) / 16; p.nkt = (d->Tk + TR - 1) / TR;
  p.ldq = d->ldq; p.ldk = d->ldk; p.ldv = d->ldv; p.ldo = d->ldo;
  p.bsk = d->bsk ? d->bsk : (int64_t)d->Tk * d->ldk;
  p.bsv = d->bsv ? d->bsv : (int64_t)d->Tk * d->ldv;
  NST_CHECK_ARG(p.bsk >= (int64_t)d->Tk * d->ldk && p.bsv >= (int64_t)d->Tk * d->ldv, "attention: k/v batch stride smaller than Tk rows");
  p.scale = d->scale; p.scale2 = d->scale * LOG2E; p.causal = d->causal;
  p.coff = d->causal ? d->causal_offset : 0;
  NST_CHECK_ARG(p.coff >= 0, "attention: causal_offset=%d must be >= 0", p.coff);
  nst_dropout_params16(d->dropout_p, &p.drop_thresh, &p.drop_inv_keep);
  if (p.drop_thresh) {
    NST_CHECK_ARG(d->dropout_mask, "attention: dropout_p > 0 needs dropout_mask");
    NST_CHECK_ARG(d->dropout_mask_bytes >= mask_bytes(d) && (((uintptr_t)d->dropout_mask) & 7) == 0,
                  "attention: dropout_mask needs %lld bytes, 8-byte aligned (got %lld)", (long long)mask_bytes(d),
                  (long long)d->dropout_mask_bytes);
    p.mask = reinterpret_cast<uint16_t*>(d->dropout_mask);
  }
  p.seed = d->seed; p.stream_id = d->stream_id;
  p.seed_dev = nst_seed_offset_devptr();
  if (!p.seed_dev) return NST_ERR_LAUNCH;
  return NST_OK;
}

// rows per workgroup = 64*MI.  Measured on MI355X (B=128, H=4, dh=64): the forward kernel gains from two 16-query
// blocks per wave (one K/V fragment read feeds two MFMAs) once there are enough workgroups to fill the chip; the
// backward kernels hold twice the accumulators and lose more to occupancy than they gain, so they stay at MI=1.
int pick_mi(const char* env, int rows, int64_t bh, bool prefer2) {
  const int forced = env_int(env, 0);
  if (forced == 1 || forced == 2) return forced;
  if (!prefer2) return 1;
  return (rows > TR && ((int64_t)((rows + 2 * TR - 1) / (2 * TR)) * bh >= 512)) ? 2 : 1;
}

#define NST_ATTN_LAUNCH(KERNEL, GRID_ROWS, MI_, VEC_)                                          \
  do {                                                                                         \
    dim3 grid_(((GRID_ROWS) + TR * (MI_) - 1) / (TR * (MI_)), d->H, d->B);                      \
    if (d->dtype == NST_F32) KERNEL<float, MI_, VEC_><<<grid_, 256, 0, st>>>(p);               \
    else KERNEL<bf16_t, MI_, VEC_><<<grid_, 256, 0, st>>>(p);                                  \
  } while (0)
#define NST_ATTN_DISPATCH(KERNEL, GRID_ROWS, mi, vec)                                          \
  do {                                                                                         \
    if ((mi) == 2) { if (vec) NST_ATTN_LAUNCH(KERNEL, GRID_ROWS, 2, true); else NST_ATTN_LAUNCH(KERNEL, GRID_ROWS, 2, false); } \
    else { if (vec) NST_ATTN_LAUNCH(KERNEL, GRID_ROWS, 1, true); else NST_ATTN_LAUNCH(KERNEL, GRID_ROWS, 1, false); }           \
  } while (0)

}  // namespace

extern "C" int64_t nst_attention_dropout_mask_bytes(const NstAttnDesc* d) {
  if (!d || d->B <= 0 || d->H <= 0 || d->Tq <= 0 || d->Tk <= 0) return 0;
  return mask_bytes(d);
}

extern "C" int nst_attention_fwd(const NstAttnDesc* d, const void* q, const void* k, const void* v, const float* key_bias,
                                 void* out, float* lse, void* stream) {
  AttnParams p;
  memset(&p, 0, sizeof(p));
  int rc = fill_params(d, p);
  if (rc) return rc;
  NST_CHECK_ARG(q && k && v && out && lse, "attention_fwd: null pointer");
  const int esz = nst_dtype_size(d->dtype);
  p.q = q; p.k = k; p.v = v; p.o = out; p.lse = lse; p.key_bias = key_bias;
  const bool vec = vec_legal(q, d->ldq, d->dh, esz) && vec_legal(k, d->ldk, d->dh, esz) && vec_legal(v, d->ldv, d->dh, esz) &&
                   vec_legal(out, d->ldo, d->dh, esz);
  hipStream_t st = (hipStream_t)stream;
  // (round 4: a one-workgroup-per-head forward with K / V staged once -- the backward's structure -- was built, bit-identical and
  // NOT faster: encoder shape 39.1 vs 41.2 us, decoder shapes slower, step 13.10 vs 13.02 ms; the forward is bound by its
  // element-wise / Philox work, not by the K / V re-reads.  Removed; profiles/r04_history/c15_attn_bench.log)
  // one query block per wave (140 registers, three waves per SIMD) measured 0.07 ms per step faster than two (236
  // registers, two waves) at the benchmark shape: the kernel is latency-bound, occupancy beats reuse (NST_ATTN_MI_FWD=2)
  const int mi = pick_mi("NST_ATTN_MI_FWD", d->Tq, (int64_t)d->B * d->H, false);
  if (d->dtype == NST_BF16 && mi == 1) {   // four waves per SIMD instead of the compiler's three (attn_fwd_occ_kernel)
    const dim3 grid((d->Tq + TR - 1) / TR, d->H, d->B);
    if (vec) attn_fwd_occ_kernel<true, 4><<<grid, 256, 0, st>>>(p);
    else attn_fwd_occ_kernel<false, 4><<<grid, 256, 0, st>>>(p);
  } else {
    NST_ATTN_DISPATCH(attn_fwd_kernel, d->Tq, mi, vec);
  }
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
  NST_CHECK_ARG(q && k && v && dout && lse && delta && dq && dk && dv, "attention_bwd: null pointer");
  const int esz = nst_dtype_size(d->dtype);
  p.q = q; p.k = k; p.v = v; p.out = out; p.dout = dout; p.lse = const_cast<float*>(lse); p.delta = delta;
  p.key_bias = key_bias; p.dq = dq; p.dk = dk; p.dv = dv;
  const bool vo = vec_legal(dout, d->ldo, d->dh, esz) && vec_legal(out, d->ldo, d->dh, esz);
  const bool vec = vec_legal(q, d->ldq, d->dh, esz) && vec_legal(k, d->ldk, d->dh, esz) && vec_legal(v, d->ldv, d->dh, esz) && vo &&
                   vec_legal(dq, d->ldq, d->dh, esz) && vec_legal(dk, d->ldk, d->dh, esz) && vec_legal(dv, d->ldv, d->dh, esz);
  hipStream_t st = (hipStream_t)stream;
  if (out) {  // out == NULL: the caller's GEMM already left delta = rowsum(dout o out) there (NstGemmDesc.rowdot_dst)
    const bool fast = vo && d->dh == 64 && (d->H == 1 || d->H == 2 || d->H == 4);
    const int64_t rows = fast ? ((int64_t)d->B * d->Tq * d->H * 16 + 63) / 64 : (int64_t)d->B * d->H * d->Tq;
    const int blocks = (int)((rows + 3) / 4 > 8192 ? 8192 : (rows + 3) / 4);
    if (fast) {
      if (d->dtype == NST_F32) attn_delta_vec_kernel<float><<<blocks, 256, 0, st>>>(p);
      else attn_delta_vec_kernel<bf16_t><<<blocks, 256, 0, st>>>(p);
    } else {
      if (d->dtype == NST_F32) attn_delta_kernel<float><<<blocks, 256, 0, st>>>(p);
      else attn_delta_kernel<bf16_t><<<blocks, 256, 0, st>>>(p);
    }
    NST_CHECK_LAUNCH("attention_bwd(delta)");
  }
  // sequences up to 256 in bf16: one 8-wave workgroup per (batch, head) -- dK / dV / dQ from one kernel, no dS^T round trip
  // through HBM (attn_bwd_head8_kernel).  NST_ATTN_FUSED_BWD=0: the two-kernel path (the test matrix runs both)
  if (d->dtype == NST_BF16 && vec && d->Tq <= HB_MAXT && d->Tk <= HB_MAXT && env_int("NST_ATTN_FUSED_BWD", 1) != 0) {
    static bool lds_ok = false;
    if (!lds_ok) {
      (void)hipFuncSetAttribute((const void*)attn_bwd_head8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, HB_LDS_BYTES);
      lds_ok = true;
    }
    attn_bwd_head8_kernel<<<dim3(d->H, d->B), 512, HB_LDS_BYTES, st>>>(p);
    NST_CHECK_LAUNCH("attention_bwd(head8)");
    return NST_OK;
  }
  const int mik = pick_mi("NST_ATTN_MI_DKDV", d->Tk, (int64_t)d->B * d->H, false);
  const int miq = pick_mi("NST_ATTN_MI_DQ", d->Tq, (int64_t)d->B * d->H, false);
  // bf16 with a workspace: the dK/dV kernel leaves dS^T behind and dQ is one small product over it (no second pass)
  const int kblock = TR * mik;
  const int64_t tkp = ((int64_t)d->Tk + kblock - 1) / kblock * kblock, tqp = ((int64_t)d->Tq + TR - 1) / TR * TR;
  const int64_t ds_need = (int64_t)d->B * d->H * tkp * tqp * 2;
  const bool use_ds = d->dtype == NST_BF16 && vec && d->ds_workspace && d->ds_workspace_bytes >= ds_need &&
                      nst_aligned16(d->ds_workspace);
  if (use_ds) {
    p.dst = (bf16_t*)d->ds_workspace; p.tqp = (int)tqp; p.tkp = (int)tkp; p.ds_kblock = kblock;
    dim3 gk((d->Tk + kblock - 1) / kblock, d->H, d->B), gq((d->Tq + TR - 1) / TR, d->H, d->B);
    if (mik == 2) attn_bwd_dkdv_kernel<bf16_t, 2, true, true><<<gk, 256, 0, st>>>(p);
    else attn_bwd_dkdv_kernel<bf16_t, 1, true, true><<<gk, 256, 0, st>>>(p);
    attn_bwd_dq_from_ds_kernel<true><<<gq, 256, 0, st>>>(p);
    NST_CHECK_LAUNCH("attention_bwd");
    return NST_OK;
  }
  NST_ATTN_DISPATCH(attn_bwd_dkdv_kernel, d->Tk, mik, vec);
  NST_ATTN_DISPATCH(attn_bwd_dq_kernel, d->Tq, miq, vec);
  NST_CHECK_LAUNCH("attention_bwd");
  return NST_OK;
}
