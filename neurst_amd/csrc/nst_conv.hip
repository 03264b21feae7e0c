// Conv2d-subsampling front end (AudioConv2dSubsamplingLayer.call, neurst/layers/modalities/audio_modalities.py:84-109).
//
// Layer 1 (C_in = 1, 9 taps): HBM-write bound -- one wavefront per output pixel, lanes own channels, the
//   3x3 conv, LayerNorm over channels (wavefront shuffle reductions, fp32) and ReLU are fused so the largest
//   activation of the model ([B, T/2, F/2, C]) is written exactly once and never re-read by a norm pass.
//   Backward recomputes the conv from the (tiny) input instead of saving the pre-norm activation.
// Layer 2 (C_in = C_out = C): implicit GEMM on MFMA (M = B*T2*F2 pixels, N = C, K = 9*C) through the loaders of
//   nst_gemm_core.h: the im2col matrix is never materialised; out-of-image taps read zeros.
//   dgrad splits the input pixels into the 4 stride-2 parity classes so every K step is a real tap
//   (no multiply-by-zero work), wgrad reduces over the pixels with split-K.
#include "nst_gemm_core.h"
#include "nst_gemm256.h"

#include <stdlib.h>

using namespace nstgemm;

namespace {

// =============================================================================================
// layer 1
// =============================================================================================
constexpr int C1_SLOTS = 8;  // channels per lane (C <= 512)

__device__ __forceinline__ int c1_channel(int lane, int s, int vec) { return vec ? lane * 4 + (s >> 2) * 256 + (s & 3) : lane + s * 64; }

constexpr int C1_MAXF = 254;  // feature bins: three padded source rows per wave live in LDS

// Stages the three source rows (ti = 2*to-1 .. 2*to+1) of one output row into wave-private LDS with a zero column
// on both sides, so tap (kh,kw) of output pixel fo is xs[kh][2*fo + kw] with no bounds checks.
__device__ __forceinline__ void c1_stage_rows(float (*xs)[C1_MAXF + 2], const float* __restrict__ src, int b, int to,
                                              int T_, int F, int lane) {
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int ti = 2 * to + r - 1;
    const bool rok = ti >= 0 && ti < T_;
    const float* p = src + ((int64_t)b * T_ + (rok ? ti : 0)) * F;
    for (int f = lane; f < F + 2; f += 64) xs[r][f] = (rok && f >= 1 && f <= F) ? p[f - 1] : 0.f;
  }
}

template <typename T, int S, bool VEC>
__global__ void __launch_bounds__(256) conv1_fwd_kernel(const float* __restrict__ src, const float* __restrict__ w1,
                                                       const float* __restrict__ b1, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, T* __restrict__ out,
                                                       float* __restrict__ mean_out, float* __restrict__ rstd_out, int B,
                                                       int T_, int F, int C, int T1, int F1, int layer_norm, float eps) {
  __shared__ float xs_all[4][3][C1_MAXF + 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float(*xs)[C1_MAXF + 2] = xs_all[wave];
  float w[S][9], bias[S], g[S], be[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int c = c1_channel(lane, s, VEC);
    const bool ok = c < C;
#pragma unroll
    for (int t = 0; t < 9; ++t) w[s][t] = ok ? w1[t * C + c] : 0.f;
    bias[s] = ok ? b1[c] : 0.f;
    g[s] = (ok && layer_norm) ? gamma[c] : 1.f;
    be[s] = (ok && layer_norm) ? beta[c] : 0.f;
  }
  const int nrows = B * T1;
  const float inv_c = 1.f / (float)C;
  for (int row = blockIdx.x * 4 + wave; row < nrows; row += gridDim.x * 4) {
    const int b = row / T1, to = row - b * T1;
    c1_stage_rows(xs, src, b, to, T_, F, lane);
    __builtin_amdgcn_wave_barrier();
    for (int fo = 0; fo < F1; ++fo) {
      float x[9];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) x[kh * 3 + kw] = xs[kh][2 * fo + kw];
      float z[S];
      float sum = 0.f;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        float a = bias[s];
#pragma unroll
        for (int t = 0; t < 9; ++t) a = fmaf(w[s][t], x[t], a);
        z[s] = a;
        sum += c1_channel(lane, s, VEC) < C ? a : 0.f;
      }
      const int64_t pix = (int64_t)row * F1 + fo;
      if (layer_norm) {
        const float mean = wave_sum_fast(sum) * inv_c;
        float sq = 0.f;
#pragma unroll
        for (int s = 0; s < S; ++s) {
          const float d = c1_channel(lane, s, VEC) < C ? z[s] - mean : 0.f;
          sq += d * d;
        }
        const float rstd = rsqrtf(wave_sum_fast(sq) * inv_c + eps);
#pragma unroll
        for (int s = 0; s < S; ++s) z[s] = (z[s] - mean) * rstd * g[s] + be[s];
        if (lane == 0) { mean_out[pix] = mean; rstd_out[pix] = rstd; }
      }
      T* o = out + pix * C;
      if (VEC) {
#pragma unroll
        for (int k = 0; k < S / 4; ++k) {
          const int c = lane * 4 + k * 256;
          if (c < C) {
            const float v0 = fmaxf(z[k * 4], 0.f), v1 = fmaxf(z[k * 4 + 1], 0.f), v2 = fmaxf(z[k * 4 + 2], 0.f), v3 = fmaxf(z[k * 4 + 3], 0.f);
            if (sizeof(T) == 2) {
              uint2 raw;
              raw.x = pack_bf16x2(v0, v1);
              raw.y = pack_bf16x2(v2, v3);
              *reinterpret_cast<uint2*>(o + c) = raw;
            } else {
              *reinterpret_cast<float4*>(o + c) = make_float4(v0, v1, v2, v3);
            }
          }
        }
      } else {
#pragma unroll
        for (int s = 0; s < S; ++s) {
          const int c = lane + s * 64;
          if (c < C) o[c] = from_f32<T>(fmaxf(z[s], 0.f));
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// backward: accumulators per lane for its channels: dw[9], db, dgamma, dbeta
template <typename T, int S, bool VEC>
__global__ void __launch_bounds__(256) conv1_bwd_kernel(const float* __restrict__ src, const float* __restrict__ w1,
                                                       const float* __restrict__ b1, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ mean_in,
                                                       const float* __restrict__ rstd_in, const T* __restrict__ dout,
                                                       float* __restrict__ dw1, float* __restrict__ db1,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int T_,
                                                       int F, int C, int T1, int F1, int layer_norm) {
  __shared__ float xs_all[4][3][C1_MAXF + 2];
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float(*xs)[C1_MAXF + 2] = xs_all[wave];
  float w[S][9], bias[S], g[S], be[S];
  float aw[S][9], ab[S], ag[S], abe[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int c = c1_channel(lane, s, VEC);
    const bool ok = c < C;
#pragma unroll
    for (int t = 0; t < 9; ++t) { w[s][t] = ok ? w1[t * C + c] : 0.f; aw[s][t] = 0.f; }
    bias[s] = ok ? b1[c] : 0.f;
    g[s] = (ok && layer_norm) ? gamma[c] : 1.f;
    be[s] = (ok && layer_norm) ? beta[c] : 0.f;
    ab[s] = 0.f; ag[s] = 0.f; abe[s] = 0.f;
  }
  const int nrows = B * T1;
  const float inv_c = 1.f / (float)C;
  // dout row of output pixel (row, fo): 512 bytes per wave -- far too little per request to cover the HBM latency if
  // loaded when needed, so the rows of the next GRP pixels are requested before the current GRP are processed
  constexpr int GRP = 4;
  auto load_gy = [&](const T* go, float (&gy)[S]) {
    if (VEC) {
#pragma unroll
      for (int k = 0; k < S / 4; ++k) {
        const int c = lane * 4 + k * 256;
        if (c < C) {
          if (sizeof(T) == 2) {
            uint2 raw = *reinterpret_cast<const uint2*>(go + c);
            gy[k * 4 + 0] = bf16_to_f32((bf16_t)(raw.x & 0xffff)); gy[k * 4 + 1] = bf16_to_f32((bf16_t)(raw.x >> 16));
            gy[k * 4 + 2] = bf16_to_f32((bf16_t)(raw.y & 0xffff)); gy[k * 4 + 3] = bf16_to_f32((bf16_t)(raw.y >> 16));
          } else {
            float4 raw = *reinterpret_cast<const float4*>(go + c);
            gy[k * 4 + 0] = raw.x; gy[k * 4 + 1] = raw.y; gy[k * 4 + 2] = raw.z; gy[k * 4 + 3] = raw.w;
          }
        } else {
          gy[k * 4 + 0] = 0.f; gy[k * 4 + 1] = 0.f; gy[k * 4 + 2] = 0.f; gy[k * 4 + 3] = 0.f;
        }
      }
    } else {
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const int c = lane + s * 64;
        gy[s] = c < C ? to_f32<T>(go[c]) : 0.f;
      }
    }
  };
  for (int row = blockIdx.x * 4 + wave; row < nrows; row += gridDim.x * 4) {
    const int b = row / T1, to = row - b * T1;
    c1_stage_rows(xs, src, b, to, T_, F, lane);
    // LayerNorm statistics of the whole output row: lane f keeps pixels f and f+64, broadcast later by a lane read
    float mean_lo = 0.f, mean_hi = 0.f, rstd_lo = 1.f, rstd_hi = 1.f;
    if (layer_norm) {
      const int64_t p0 = (int64_t)row * F1;
      if (lane < F1) { mean_lo = mean_in[p0 + lane]; rstd_lo = rstd_in[p0 + lane]; }
      if (lane + 64 < F1) { mean_hi = mean_in[p0 + lane + 64]; rstd_hi = rstd_in[p0 + lane + 64]; }
    }
    __builtin_amdgcn_wave_barrier();
    float gyn[GRP][S];
#pragma unroll
    for (int u = 0; u < GRP; ++u) {
      if (u < F1) load_gy(dout + ((int64_t)row * F1 + u) * C, gyn[u]);
      else {
#pragma unroll
        for (int s = 0; s < S; ++s) gyn[u][s] = 0.f;
      }
    }
    for (int f0 = 0; f0 < F1; f0 += GRP) {
      float gyc[GRP][S];
#pragma unroll
      for (int u = 0; u < GRP; ++u)
#pragma unroll
        for (int s = 0; s < S; ++s) gyc[u][s] = gyn[u][s];
#pragma unroll
      for (int u = 0; u < GRP; ++u)
        if (f0 + GRP + u < F1) load_gy(dout + ((int64_t)row * F1 + f0 + GRP + u) * C, gyn[u]);
#pragma unroll
      for (int u = 0; u < GRP; ++u) {
        const int fo = f0 + u;
        if (fo < F1) {
          float x[9];
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) x[kh * 3 + kw] = xs[kh][2 * fo + kw];
          const float mean = __shfl(fo < 64 ? mean_lo : mean_hi, fo & 63, 64);
          const float rstd = __shfl(fo < 64 ? rstd_lo : rstd_hi, fo & 63, 64);
          float xh[S], dxh[S];
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int s = 0; s < S; ++s) {
            float a = bias[s];
#pragma unroll
            for (int t = 0; t < 9; ++t) a = fmaf(w[s][t], x[t], a);
            float gv = gyc[u][s];
            if (layer_norm) {
              const float xhat = (a - mean) * rstd;
              const float y = xhat * g[s] + be[s];
              gv = y > 0.f ? gv : 0.f;
              ag[s] += gv * xhat;
              abe[s] += gv;
              const float d = gv * g[s];
              xh[s] = xhat;
              dxh[s] = d;
              s1 += d;
              s2 += d * xhat;
            } else {
              dxh[s] = a > 0.f ? gv : 0.f;
              xh[s] = 0.f;
            }
          }
          float c1 = 0.f, c2 = 0.f;
          if (layer_norm) { c1 = wave_sum_fast(s1) * inv_c; c2 = wave_sum_fast(s2) * inv_c; }
#pragma unroll
          for (int s = 0; s < S; ++s) {
            const float dz = layer_norm ? rstd * (dxh[s] - c1 - xh[s] * c2) : dxh[s];
            const float dzc = c1_channel(lane, s, VEC) < C ? dz : 0.f;
            ab[s] += dzc;
#pragma unroll
            for (int t = 0; t < 9; ++t) aw[s][t] = fmaf(dzc, x[t], aw[s][t]);
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  // reduce the 4 waves of the block, then one atomic per (tap, channel) per block
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int c = c1_channel(lane, s, VEC);
#pragma unroll
    for (int q = 0; q < 12; ++q) {
      const float v = q < 9 ? aw[s][q < 9 ? q : 0] : (q == 9 ? ab[s] : (q == 10 ? ag[s] : abe[s]));
      __syncthreads();
      red[wave][lane] = v;
      __syncthreads();
      if (wave == 0 && c < C) {
        const float t = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
        if (q < 9) atomicAdd(dw1 + q * C + c, t);
        else if (q == 9) atomicAdd(db1 + c, t);
        else if (layer_norm) atomicAdd((q == 10 ? dgamma : dbeta) + c, t);
      }
    }
  }
}

// =============================================================================================
// layer 2 loaders
// =============================================================================================
// im2col element (pixel m, kk = tap*C + c) of a stride-2, pad-1, 3x3 conv over x [B,T1,F1,C].
// Serves the forward A operand (RC: outer = pixel, contig = kk) and the wgrad A operand (OC: outer = pixel, contig = kk).
template <typename T>
struct Im2colLoader {
  const T* x;
  int B, T1, F1, C, T2, F2;
  FastDiv dF2, dT2, dC;
  int outer_limit, contig_limit;  // pixels, 9*C
  int vec;                        // C % E == 0 and 16-byte aligned base
  void init_divs() { dF2.init(F2); dT2.init(T2); dC.init(C); }
  __device__ __forceinline__ const T* addr(int b, int to, int fo, int kk, bool& ok) const {
    uint32_t tap, c;
    dC.divmod((uint32_t)kk, tap, c);
    const int kh = (int)tap / 3, kw = (int)tap - kh * 3;
    const int ti = 2 * to + kh - 1, fi = 2 * fo + kw - 1;
    ok = ti >= 0 && ti < T1 && fi >= 0 && fi < F1;
    return x + (((int64_t)b * T1 + ti) * F1 + fi) * C + (int)c;
  }
  __device__ __forceinline__ const T* ptr(int outer, int contig) const {
    if (outer >= outer_limit || contig >= contig_limit) return nullptr;
    uint32_t q, fo, b, to;
    dF2.divmod((uint32_t)outer, q, fo);
    dT2.divmod(q, b, to);
    bool ok;
    const T* p = addr((int)b, (int)to, (int)fo, contig, ok);
    return ok ? p : nullptr;
  }
  __device__ __forceinline__ uint4 load(int outer, int contig) const {
    uint4 r = make_uint4(0, 0, 0, 0);
    if (outer >= outer_limit || contig >= contig_limit) return r;
    uint32_t q, fo, b, to;
    dF2.divmod((uint32_t)outer, q, fo);
    dT2.divmod(q, b, to);
    if (vec) {
      bool ok;
      const T* p = addr((int)b, (int)to, (int)fo, contig, ok);
      if (ok) r = *reinterpret_cast<const uint4*>(p);
      return r;
    }
    T tmp[Tile<T>::E];
#pragma unroll
    for (int e = 0; e < Tile<T>::E; ++e) {
      tmp[e] = (T)0;
      if (contig + e < contig_limit) {
        bool ok;
        const T* p = addr((int)b, (int)to, (int)fo, contig + e, ok);
        if (ok) tmp[e] = *p;
      }
    }
    memcpy(&r, tmp, 16);
    return r;
  }
};


// =============================================================================================
// layer 1 backward on the matrix cores (bf16 activations, C a multiple of 16, C <= 256).
//
// The VALU kernel above spends 9 + 9 fp32 FMAs per output element (recomputed conv + tap gradient); both are tiny
// GEMMs, so a wave takes 32 consecutive output pixels and runs, per block of 16 channels,
//   (1)  Z[pix, c]   = P[pix, :] . W[:, c]          one v_mfma_f32_16x16x32_bf16 per 16 pixels: the 32 K slots hold
//        x_hi*w_hi + x_hi*w_lo + x_lo*w_hi for the 9 taps plus 1*b_hi + 1*b_lo (bf16 hi/lo splits of the fp32 inputs:
//        the recomputed pre-norm activation agrees with the fp32 forward to ~2^-16 relative);
//   (2)  dW[t, c]   += P^T[t, pix] . dZ[pix, c]      one MFMA per 32 pixels; row 9 of P^T is all ones, so the same
//        instruction accumulates db1.  dZ reaches the B operand straight from the registers that computed it: a
//        16x16 accumulator holds rows (l>>4)*4+r of column l&15, i.e. 4 (per tile) x 2 tiles = the 8 reduction slots
//        of lane l -- the A operand simply enumerates the pixels in the same order.
// The LayerNorm backward needs two per-pixel channel means first, hence two passes over the channel blocks (Z is
// recomputed in the second: two MFMAs).  dgamma / dbeta are accumulated per lane and reduced at the end.
// =============================================================================================
typedef __attribute__((ext_vector_type(2))) unsigned c1_uint2_t;
__device__ __forceinline__ float c1_swap16_add(float v) {
  const c1_uint2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float c1_swap32_add(float v) {
  const c1_uint2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float c1_row16_sum(float v) {  // sum over the 16 lanes that share l>>4
  v = dpp_add(v, 0); v = dpp_add(v, 1); v = dpp_add(v, 2); v = dpp_add(v, 3);
  return v;
}
__device__ __forceinline__ uint32_t c1_relu_bf16x2(uint32_t w) {   // v_pk_max_i16: -0 and every negative value -> +0
  typedef short c1_short2_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(c1_short2_t, w), c1_short2_t{0, 0}));
}

__device__ __forceinline__ void c1_split_bf16(float x, bf16_t& hi, bf16_t& lo) {
  hi = f32_to_bf16(x);
  lo = f32_to_bf16(x - bf16_to_f32(hi));
}

constexpr int C1M_PT = 12;  // floats per pixel in the wave's patch table: 9 taps, -mean*rstd, rstd, pixel mask
constexpr int C1M_WAVES = 4;  // one wave per SIMD: the kernel needs ~300 registers (64 MFMA accumulators + per-pixel state)

// WV waves per workgroup, NB dout-tile buffers per wave.  (4, 2): one wave per SIMD, the next group's tile lands under the
// current group's work.  (8, 1) -- the V2 form below, C = 256: two waves per SIMD in the same 160 KB; a wave's next tile is
// requested when it has finished reading the current one and the SIMD's other wave computes while it lands.
template <int NCB, int WV = C1M_WAVES, int NB = 2>
struct C1mLds {
  uint4 w[NCB][64];                    // B operand of product (1) per channel block
  float2 gb[NCB * 16];                 // (gamma, beta)
  float pt[WV][32][C1M_PT];            // per-wave patch table
  char dst[WV][NB][32 * NCB * 32];     // per-wave dout tiles: 32 pixels x C bf16, 32-byte units XOR-swizzled by (pix>>2)&3
};

// V2 (round 3): the element-wise loops in packed-f32 form on transposed LDS reads.  The first form needs 497 registers (the
// compiler parks 241 of them in the accumulation file: 887 v_accvgpr moves, one wave per SIMD) and ~3800 instructions per
// 32-pixel group, 320 of them 2-byte LDS reads; it is bound by instruction issue (a software pipeline over its channel blocks
// changed nothing).  V2 reads the 4 pixels x 1 channel a lane needs with ONE ds_read_b64_tr_b16, does the arithmetic on float2
// pairs (v_pk_fma_f32 ...), is written branch-free, and is compiled for 256 registers so that eight waves fit a CU.
template <int NCB, bool LN, int WV = C1M_WAVES, int NB = 2, bool V2 = false>
__global__ void __launch_bounds__(WV * 64, WV == 8 ? 2 : 1) conv1_bwd_mfma_kernel(
    const float* __restrict__ src, const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
    const bf16_t* __restrict__ dout, float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dgamma,
    float* __restrict__ dbeta, int T_, int F, int T1, int F1, int64_t npix, FastDiv dF1, FastDiv dT1) {
  constexpr int C = NCB * 16;
  constexpr int ROWB = C * 2;  // bytes per pixel row of dout
  extern __shared__ __attribute__((aligned(16))) char c1m_smem[];
  typedef C1mLds<NCB, WV, NB> Lds;
  Lds& L = *reinterpret_cast<Lds*>(c1m_smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, lc = lane & 15;

  // ---- weights -> MFMA B fragments (once per workgroup).  K slot (g, j) carries, for channel c:
  //   g=0: w_hi[tap j]   g=1: w_lo[tap j]   g=2: w_hi[tap j]   (taps 0..7; the A side holds x_hi, x_hi, x_lo)
  //   g=3: j=0 w_hi[8], j=1 w_lo[8], j=2 w_hi[8], j=3 b_hi, j=4 b_lo, j=5..7 zero
  for (int i = tid; i < NCB * 64; i += WV * 64) {
    const int cb = i >> 6, l = i & 63, gg = l >> 4, c = cb * 16 + (l & 15);
    bf16_t v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      bf16_t hi, lo;
      float x = 0.f;
      bool want_lo = false;
      if (gg < 3) { x = w1[j * C + c]; want_lo = (gg == 1); }
      else if (j < 3) { x = w1[8 * C + c]; want_lo = (j == 1); }
      else if (j < 5) { x = b1[c]; want_lo = (j == 4); }
      c1_split_bf16(x, hi, lo);
      v[j] = want_lo ? lo : hi;
    }
    uint4 pk;
    pk.x = v[0] | ((uint32_t)v[1] << 16); pk.y = v[2] | ((uint32_t)v[3] << 16);
    pk.z = v[4] | ((uint32_t)v[5] << 16); pk.w = v[6] | ((uint32_t)v[7] << 16);
    L.w[cb][l] = pk;
  }
  for (int c = tid; c < C; c += WV * 64) L.gb[c] = LN ? make_float2(gamma[c], beta[c]) : make_float2(1.f, 0.f);
  __syncthreads();

  floatx4_t accw[NCB];
  float ag[NCB], abe[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) { accw[cb] = floatx4_t{0.f, 0.f, 0.f, 0.f}; ag[cb] = 0.f; abe[cb] = 0.f; }
  const float inv_c = 1.f / (float)C;
  float(*pt)[C1M_PT] = L.pt[wave];
  typedef __attribute__((address_space(3))) char* lds_char_ptr;
  const uint32_t dst_addr0 = (uint32_t)(uintptr_t)((lds_char_ptr)L.dst[wave][0]);
  const int64_t ngroups = (npix + 31) / 32;
  const int64_t gstride = (int64_t)gridDim.x * WV;

  // dout tile of a group -> LDS by LDS-DMA (1 KB = 2 pixel rows per instruction); pixels past the end stage zeros
  auto stage_tile = [&](int64_t p0, int buf) {
    constexpr int NI = 32 * ROWB / 1024;  // instructions per tile
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int byte = i * 1024 + lane * 16;
      const int pix = byte / ROWB, chunk = (byte % ROWB) >> 4;
      const int64_t p = p0 + pix;
      const int schunk = chunk ^ (((pix >> 2) & 3) << 1);  // source chunk that lands at this LDS position
      const void* srcp = p < npix ? (const void*)(reinterpret_cast<const char*>(dout) + p * ROWB + schunk * 16)
                                  : (const void*)g_nst_zero16;
      glds16(srcp, dst_addr0 + buf * (32 * ROWB) + i * 1024);
    }
  };
  // patch-table row of pixel p0 + lane (lanes < 32): 9 zero-padded taps, -mean*rstd, rstd, pixel mask
  struct PixRow { float x[9], nmr, rs, msk; };
  auto gather_row = [&](int64_t p0) {
    PixRow q;
#pragma unroll
    for (int t = 0; t < 9; ++t) q.x[t] = 0.f;
    q.nmr = 0.f; q.rs = 0.f; q.msk = 0.f;
    const int64_t p = p0 + lane;
    if (lane < 32 && p < npix) {
      uint32_t row, fo, b, to;
      dF1.divmod((uint32_t)p, row, fo);
      dT1.divmod(row, b, to);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int ti = 2 * (int)to + kh - 1;
        if (ti >= 0 && ti < T_) {
          const float* sp = src + ((int64_t)b * T_ + ti) * F;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int fi = 2 * (int)fo + kw - 1;
            if (fi >= 0 && fi < F) q.x[kh * 3 + kw] = sp[fi];
          }
        }
      }
      q.msk = 1.f;
      if (LN) { q.rs = rstd_in[p]; q.nmr = -mean_in[p] * q.rs; } else { q.rs = 1.f; }
    }
    return q;
  };
  auto write_row = [&](const PixRow& q) {
    if (lane < 32) {
#pragma unroll
      for (int t = 0; t < 9; ++t) pt[lane][t] = q.x[t];
      pt[lane][9] = q.nmr;
      pt[lane][10] = q.rs;
      pt[lane][11] = q.msk;
    }
  };

  int64_t grp = (int64_t)blockIdx.x * WV + wave;
  int buf = 0;
  if (grp < ngroups) {
    stage_tile(grp * 32, 0);
    write_row(gather_row(grp * 32));
  }
  for (; grp < ngroups; grp += gstride) {
    const int64_t p0 = grp * 32;
    const char* dst = L.dst[wave][buf];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // this group's tile has landed, its table is written
    __builtin_amdgcn_wave_barrier();

    // ---- A operands of product (1): pixel u*16 + lc, K slots by g (see the weight layout above)
    bf16x8_t a1[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float* xp = pt[u * 16 + lc];
      bf16_t v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        bf16_t hi, lo;
        c1_split_bf16(xp[j], hi, lo);
        v[j] = g == 2 ? lo : hi;
      }
      if (g == 3) {
        bf16_t hi, lo;
        c1_split_bf16(xp[8], hi, lo);
        v[0] = hi; v[1] = hi; v[2] = lo; v[3] = 0x3F80; v[4] = 0x3F80; v[5] = 0; v[6] = 0; v[7] = 0;
      }
      union { uint32_t w[4]; bf16x8_t f; } pk;
      pk.w[0] = v[0] | ((uint32_t)v[1] << 16); pk.w[1] = v[2] | ((uint32_t)v[3] << 16);
      pk.w[2] = v[4] | ((uint32_t)v[5] << 16); pk.w[3] = v[6] | ((uint32_t)v[7] << 16);
      a1[u] = pk.f;
    }
    // ---- A operand of product (2): row t = lc (taps 0..8, row 9 = ones), K slot j = pixel (j>>2)*16 + g*4 + (j&3)
    bf16x8_t a2;
    {
      bf16_t v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int pix = (j >> 2) * 16 + g * 4 + (j & 3);
        const float x = (lc < 9 ? pt[pix][lc < 9 ? lc : 0] : (lc == 9 ? 1.f : 0.f)) * pt[pix][11];
        v[j] = f32_to_bf16(x);
      }
      union { uint32_t w[4]; bf16x8_t f; } pk;
      pk.w[0] = v[0] | ((uint32_t)v[1] << 16); pk.w[1] = v[2] | ((uint32_t)v[3] << 16);
      pk.w[2] = v[4] | ((uint32_t)v[5] << 16); pk.w[3] = v[6] | ((uint32_t)v[7] << 16);
      a2 = pk.f;
    }
    // ---- per-pixel scalars of this lane's accumulator rows: pixel u*16 + g*4 + r
    float nmr[2][4], rs[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int pix = u * 16 + g * 4 + r;
        nmr[u][r] = pt[pix][9];
        rs[u][r] = pt[pix][10];
      }
    // ---- the table is consumed: start the next group (tile by DMA into the other buffer, table row into registers)
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    const int64_t nxt = grp + gstride;
    PixRow nrow;
    if (nxt < ngroups) {
      if (NB == 2) stage_tile(nxt * 32, buf ^ 1);   // (one buffer: requested below, behind this group's last read of the tile)
      nrow = gather_row(nxt * 32);
    }
    // dout of channel block cb for the lane's 8 pixels from the staged tile: row (u*16+g*4+r)*ROWB, 32-byte unit cb ^ g
    int wofs = lane, gofs = lc;  // laundered per group: keeps the table reads below from being hoisted out of the loop
    int dofs = g * 4 * ROWB + lc * 2;
    asm volatile("" : "+v"(wofs), "+v"(gofs), "+v"(dofs));
    const int xg0 = (0 ^ g) << 5, xg1 = (1 ^ g) << 5, xg2 = (2 ^ g) << 5, xg3 = (3 ^ g) << 5;
    auto load_go = [&](int cb, float (&go)[2][4]) {
      const int k = cb & 3;
      const int unit = ((cb & ~3) << 5) + (k == 0 ? xg0 : (k == 1 ? xg1 : (k == 2 ? xg2 : xg3)));
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bf16_t raw = *reinterpret_cast<const bf16_t*>(dst + dofs + (u * 16 + r) * ROWB + unit);
          go[u][r] = bf16_to_f32(raw);
        }
    };

    if constexpr (V2) {
      typedef float f2 __attribute__((ext_vector_type(2)));
      typedef short4_t __attribute__((address_space(3))) * lds_tr_ptr_t;
      // dout of channel block cb for pixels u*16 + g*4 + 0..3 at channel cb*16 + lc: ONE transposed read per u.  The 16 lanes of a
      // group address the 4 x 16 block (row u*16 + g*4 + (lc >> 2), 8 bytes at channel (lc & 3) * 4 of the 32-byte unit cb ^ g)
      // and receive it transposed.
      const int trofs = (g * 4 + (lc >> 2)) * ROWB + ((lc & 3) << 3);
      int trl = trofs;
      asm volatile("" : "+v"(trl));
      auto load_go2 = [&](int cb, short4_t (&raw)[2]) {
        const int unit = ((cb & ~3) << 5) + (((cb & 3) ^ g) << 5);
#pragma unroll
        for (int u = 0; u < 2; ++u)
          raw[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr_t)(dst + trl + u * 16 * ROWB + unit));
      };
      auto unpack = [&](const short4_t& r, int pr) {   // pixels 2 pr, 2 pr + 1 of the lane's four
        const uint32_t w = pr == 0 ? __builtin_bit_cast(uint2, r).x : __builtin_bit_cast(uint2, r).y;
        return f2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
      };
      f2 rs2[2][2], nm2[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          rs2[u][pr] = f2{rs[u][2 * pr], rs[u][2 * pr + 1]};
          nm2[u][pr] = f2{nmr[u][2 * pr], nmr[u][2 * pr + 1]};
        }
      f2 c1v[2][2], c2v[2][2];
      if (LN) {
        f2 s1[2][2], s2[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) { s1[u][pr] = f2{0.f, 0.f}; s2[u][pr] = f2{0.f, 0.f}; }
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          short4_t raw[2];
          load_go2(cb, raw);
          const bf16x8_t wf = __builtin_bit_cast(bf16x8_t, L.w[cb][wofs]);
          const float2 gbv = L.gb[cb * 16 + gofs];
          const f2 gx = f2{gbv.x, gbv.x}, gy = f2{gbv.y, gbv.y};
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const floatx4_t z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[u], wf, floatx4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
              const f2 xhat = __builtin_elementwise_fma(f2{z[2 * pr], z[2 * pr + 1]}, rs2[u][pr], nm2[u][pr]);
              const f2 y = __builtin_elementwise_fma(xhat, gx, gy);
              const f2 go = unpack(raw[u], pr);
              const f2 d = f2{y.x > 0.f ? go.x : 0.f, y.y > 0.f ? go.y : 0.f} * gx;
              s1[u][pr] += d;
              s2[u][pr] = __builtin_elementwise_fma(d, xhat, s2[u][pr]);
            }
          }
          __builtin_amdgcn_sched_barrier(0);  // bounds how far LDS reads are hoisted (register pressure)
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            c1v[u][pr] = f2{c1_row16_sum(s1[u][pr].x), c1_row16_sum(s1[u][pr].y)} * inv_c;
            c2v[u][pr] = f2{c1_row16_sum(s2[u][pr].x), c1_row16_sum(s2[u][pr].y)} * inv_c;
          }
        asm volatile("" : "+v"(wofs), "+v"(gofs), "+v"(trl));   // the second pass re-reads the tables (see the first form)
      }
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        short4_t raw[2];
        load_go2(cb, raw);
        const bf16x8_t wf = __builtin_bit_cast(bf16x8_t, L.w[cb][wofs]);
        const float2 gbv = L.gb[cb * 16 + gofs];
        const f2 gx = f2{gbv.x, gbv.x}, gy = f2{gbv.y, gbv.y};
        f2 la = f2{0.f, 0.f}, lb = f2{0.f, 0.f};
        uint32_t packed[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const floatx4_t z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[u], wf, floatx4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            const f2 go = unpack(raw[u], pr);
            f2 dz;
            if (LN) {
              const f2 xhat = __builtin_elementwise_fma(f2{z[2 * pr], z[2 * pr + 1]}, rs2[u][pr], nm2[u][pr]);
              const f2 y = __builtin_elementwise_fma(xhat, gx, gy);
              const f2 gv = f2{y.x > 0.f ? go.x : 0.f, y.y > 0.f ? go.y : 0.f};
              la = __builtin_elementwise_fma(gv, xhat, la);
              lb += gv;
              // rs * (gv * gamma - c1 - xhat * c2)
              dz = rs2[u][pr] * (__builtin_elementwise_fma(gv, gx, -c1v[u][pr]) - xhat * c2v[u][pr]);
            } else {
              dz = f2{z[2 * pr] > 0.f ? go.x : 0.f, z[2 * pr + 1] > 0.f ? go.y : 0.f};
            }
            packed[u][pr] = pack_bf16x2(dz.x, dz.y);
          }
        }
        if (LN) { ag[cb] += la.x + la.y; abe[cb] += lb.x + lb.y; }
        union { uint32_t w[4]; bf16x8_t f; } b2;
        b2.w[0] = packed[0][0]; b2.w[1] = packed[0][1]; b2.w[2] = packed[1][0]; b2.w[3] = packed[1][1];
        accw[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2.f, accw[cb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
    float c1[2][4], c2[2][4];
    if (LN) {
      float s1[2][4], s2[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[u][r] = 0.f; s2[u][r] = 0.f; }
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        float go[2][4];
        load_go(cb, go);
        const bf16x8_t wf = __builtin_bit_cast(bf16x8_t, L.w[cb][wofs]);
        const float2 gbv = L.gb[cb * 16 + gofs];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const floatx4_t z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[u], wf, floatx4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float xhat = fmaf(z[r], rs[u][r], nmr[u][r]);
            const float y = fmaf(xhat, gbv.x, gbv.y);
            const float d = (y > 0.f ? go[u][r] : 0.f) * gbv.x;
            s1[u][r] += d;
            s2[u][r] = fmaf(d, xhat, s2[u][r]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);  // bounds how far LDS reads are hoisted (register pressure)
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          c1[u][r] = c1_row16_sum(s1[u][r]) * inv_c;
          c2[u][r] = c1_row16_sum(s2[u][r]) * inv_c;
        }
      // second pass re-reads the tables: without this the compiler keeps all first-pass values (14 VGPRs per block) live
      asm volatile("" : "+v"(wofs), "+v"(gofs), "+v"(dofs));
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      float go[2][4];
      load_go(cb, go);
      const bf16x8_t wf = __builtin_bit_cast(bf16x8_t, L.w[cb][wofs]);
      const float2 gbv = L.gb[cb * 16 + gofs];
      float dz[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const floatx4_t z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[u], wf, floatx4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (LN) {
            const float xhat = fmaf(z[r], rs[u][r], nmr[u][r]);
            const float y = fmaf(xhat, gbv.x, gbv.y);
            const float gv = y > 0.f ? go[u][r] : 0.f;
            const float d = gv * gbv.x;
            ag[cb] = fmaf(gv, xhat, ag[cb]);
            abe[cb] += gv;
            dz[u][r] = rs[u][r] * (d - c1[u][r] - xhat * c2[u][r]);
          } else {
            dz[u][r] = z[r] > 0.f ? go[u][r] : 0.f;
          }
        }
      }
      union { uint32_t w[4]; bf16x8_t f; } b2;
      b2.w[0] = pack_bf16x2(dz[0][0], dz[0][1]); b2.w[1] = pack_bf16x2(dz[0][2], dz[0][3]);
      b2.w[2] = pack_bf16x2(dz[1][0], dz[1][1]); b2.w[3] = pack_bf16x2(dz[1][2], dz[1][3]);
      accw[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2.f, accw[cb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    }
    if (NB == 1 && nxt < ngroups) {
      __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the tile's last reads have returned
      __builtin_amdgcn_wave_barrier();
      stage_tile(nxt * 32, 0);
    }
    if (nxt < ngroups) write_row(nrow);
    if (NB == 2) buf ^= 1;
  }

  // ---- reductions: accw[cb][r] = gradient of (tap g*4+r | db1 at 9, channel cb*16+lc); ag/abe per (cb, lc) summed over g.
  // The dout staging area is free now and doubles as the cross-wave reduction buffer [wave][C].
  __syncthreads();
  float* red = reinterpret_cast<float*>(&L.dst[0][0][0]);
  for (int q = 0; q < 12; ++q) {  // q = 0..8 taps, 9 db1, 10 dgamma, 11 dbeta
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      if (q < 10) {
        if ((q >> 2) == g) {
          const int r = q & 3;
          const float v = r == 0 ? accw[cb][0] : (r == 1 ? accw[cb][1] : (r == 2 ? accw[cb][2] : accw[cb][3]));
          red[wave * C + cb * 16 + lc] = v;
        }
      } else if (LN) {
        float v = q == 10 ? ag[cb] : abe[cb];
        v = c1_swap32_add(c1_swap16_add(v));  // over the 4 lanes that share lc
        if (g == 0) red[wave * C + cb * 16 + lc] = v;
      }
    }
    __syncthreads();
    if (q < 10 || LN) {
      for (int c = tid; c < C; c += WV * 64) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < WV; ++w) t += red[w * C + c];
        if (q < 9) atomicAdd(dw1 + q * C + c, t);
        else if (q == 9) atomicAdd(db1 + c, t);
        else atomicAdd((q == 10 ? dgamma : dbeta) + c, t);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Layer 1 forward on the matrix cores (round 5; bf16 output, C == 256).  Its predecessor (two pixels per wavefront pass on the
// vector ALU: 36 packed FMAs for the taps + the LayerNorm per pair) took 340 us on the benchmark shape, twice the write time of
// its 1.18 GB output; this one 245.  Here the
// conv is product (1) of the backward kernel with the operands swapped: Z^T[channel, pixel] = W^T . P^T, so that a lane of the
// 16 x 16 accumulator holds 4 CHANNELS of ONE pixel -- the LayerNorm sums are in-lane adds plus two lane swaps, and with the
// channel rows of a pair of MFMAs interleaved (row g*4 + r of MFMA j = channel pb*32 + g*8 + j*4 + r) a lane ends up with 8
// consecutive channels: one 16-byte store.  Same K-slot layout as the backward (x_hi*w_hi + x_hi*w_lo + x_lo*w_hi for the nine
// taps, 1*b_hi + 1*b_lo): the pre-norm activation agrees with the fp32 kernels to ~2^-16 relative and is what the backward
// recomputes.  A wave owns 16 consecutive pixels per pass; the 16 weight fragments are shared through LDS.
// ---------------------------------------------------------------------------------------------
template <bool LN>
__global__ void __launch_bounds__(256, 2) conv1_fwd_mfma_kernel(const float* __restrict__ src, const float* __restrict__ w1,
                                                                const float* __restrict__ b1, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, bf16_t* __restrict__ out,
                                                                float* __restrict__ mean_out, float* __restrict__ rstd_out, int T_,
                                                                int F, int T1, int F1, int64_t npix, FastDiv dF1, FastDiv dT1,
                                                                float eps) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  constexpr int C = 256;
  __shared__ __attribute__((aligned(16))) float gb_s[2][C];   // gamma, beta
  __shared__ __attribute__((aligned(16))) float pt_s[4][2][4][16][12];   // per wave: 2 blocks x 4 groups x 16 pixels: the 9 taps
  __shared__ uint4 wf_s[17][64];                                   // the 16 weight fragments (A operands) + the sum fragment
  __shared__ __attribute__((aligned(16))) char tr_s[4 * 16 * 528];   // per wave: the 16 output rows of a pass
  float(*wsum_s)[C] = reinterpret_cast<float(*)[C]>(tr_s);   // prologue only: w1 | b1 rows, then their channel sums in column 0
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, lc = lane & 15;
  for (int c = tid; c < C; c += 256) {
    gb_s[0][c] = LN ? gamma[c] : 1.f;
    gb_s[1][c] = LN ? beta[c] : 0.f;
  }
  // weight fragments: A operand of MFMA (pb, j), lane (g, lc): row lc = channel pb*32 + (lc>>2)*8 + j*4 + (lc&3); K slots of g.
  // Fragment 16 holds the channel SUMS of the weights in every row: its product is 256 x the LayerNorm mean of the pixel, in
  // all four accumulator registers of all four lanes of the pixel (no adds, no lane swaps).
  for (int i = tid; i < 10 * C; i += 256) wsum_s[i / C][i % C] = i < 9 * C ? w1[i] : b1[i - 9 * C];
  __syncthreads();
  if (tid < 10) {
    float t = 0.f;
    for (int c = 0; c < C; ++c) t += wsum_s[tid][c];
    wsum_s[tid][0] = t;
  }
  __syncthreads();
  for (int i = tid; i < 17 * 64; i += 256) {
    const int m = i >> 6, l = i & 63, gg = l >> 4, ll = l & 15;
    const int c = (m >> 1) * 32 + (ll >> 2) * 8 + (m & 1) * 4 + (ll & 3);
    bf16_t v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float x = 0.f;
      bool want_lo = false;
      if (gg < 3) { x = m < 16 ? w1[k * C + c] : wsum_s[k][0]; want_lo = (gg == 1); }
      else if (k < 3) { x = m < 16 ? w1[8 * C + c] : wsum_s[8][0]; want_lo = (k == 1); }
      else if (k < 5) { x = m < 16 ? b1[c] : wsum_s[9][0]; want_lo = (k == 4); }
      bf16_t hi, lo;
      c1_split_bf16(x, hi, lo);
      v[k] = want_lo ? lo : hi;
    }
    wf_s[m][l] = make_uint4(v[0] | ((uint32_t)v[1] << 16), v[2] | ((uint32_t)v[3] << 16), v[4] | ((uint32_t)v[5] << 16),
                            v[6] | ((uint32_t)v[7] << 16));
  }
  __syncthreads();

  const int64_t ngroups = (npix + 15) >> 4;
  const int64_t gstride = (int64_t)gridDim.x * 4;
  const int64_t first = (int64_t)blockIdx.x * 4 + wave;
  // The taps are fetched a BLOCK of four groups at a time (lane quarter q gathers pixel lc of the block's group q: every lane
  // works) and one to two blocks ahead: vector memory operations retire in order, so a wave that waits for a load also waits
  // for every store it issued before that load -- with a look-ahead of one or two groups it sat on its own stores every pass
  // (ablation: 184 us of arithmetic, + 43 with the stores, + 43 with the gather, + 178 with both).
  struct Taps { float x[9]; };
  auto gather = [&](int64_t blk) {   // groups first + (4 blk + q) gstride, q = lane >> 4
    Taps q;
#pragma unroll
    for (int t = 0; t < 9; ++t) q.x[t] = 0.f;
    const int64_t grp_l = first + (4 * blk + g) * gstride;
    const int64_t p = grp_l * 16 + lc;
    if (grp_l < ngroups && p < npix) {
      uint32_t row, fo, b, to;
      dF1.divmod((uint32_t)p, row, fo);
      dT1.divmod(row, b, to);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int ti = 2 * (int)to + kh - 1;
        if (ti >= 0 && ti < T_) {
          const float* sp = src + ((int64_t)b * T_ + ti) * F;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int fi = 2 * (int)fo + kw - 1;
            if (fi >= 0 && fi < F) q.x[kh * 3 + kw] = sp[fi];
          }
        }
      }
    }
    return q;
  };
  auto write_taps = [&](const Taps& q, int buf) {   // table [buf][q][lc]
    float* row = &pt_s[wave][buf][g][lc][0];
    *reinterpret_cast<float4*>(row) = make_float4(q.x[0], q.x[1], q.x[2], q.x[3]);
    *reinterpret_cast<float4*>(row + 4) = make_float4(q.x[4], q.x[5], q.x[6], q.x[7]);
    row[8] = q.x[8];
  };
  const float inv_c = 1.f / (float)C;
  Taps pend;
  write_taps(gather(0), 0);
  pend = gather(1);
  int64_t n = 0;
  for (int64_t grp = first; grp < ngroups; grp += gstride, ++n) {
    const int64_t p0 = grp * 16;
    const int blk = (int)(n >> 2), q4 = (int)(n & 3);
    if (q4 == 0) {   // block blk starts: its tables are in buffer blk & 1; the next block's taps go into the other one
      write_taps(pend, (blk + 1) & 1);
      pend = gather(blk + 2);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the tables are written
    __builtin_amdgcn_wave_barrier();
    const float(*pt)[12] = pt_s[wave][blk & 1][q4];
    // ---- B operand: pixel lc, K slots of g (x_hi, x_hi, x_lo for taps 0..7; g = 3: tap 8 three times, the two ones of the bias)
    bf16x8_t bfrag;
    {
      const float4 xa = *reinterpret_cast<const float4*>(&pt[lc][0]);
      const float4 xb = *reinterpret_cast<const float4*>(&pt[lc][4]);
      const float x8 = pt[lc][8];
      const float xs[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
      bf16_t v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        bf16_t hi, lo;
        c1_split_bf16(xs[k], hi, lo);
        v[k] = g == 2 ? lo : hi;
      }
      if (g == 3) {
        bf16_t hi, lo;
        c1_split_bf16(x8, hi, lo);
        v[0] = hi; v[1] = hi; v[2] = lo; v[3] = 0x3F80; v[4] = 0x3F80; v[5] = 0; v[6] = 0; v[7] = 0;
      }
      union { uint32_t w[4]; bf16x8_t f; } pk;
      pk.w[0] = v[0] | ((uint32_t)v[1] << 16); pk.w[1] = v[2] | ((uint32_t)v[3] << 16);
      pk.w[2] = v[4] | ((uint32_t)v[5] << 16); pk.w[3] = v[6] | ((uint32_t)v[7] << 16);
      bfrag = pk.f;
    }

    floatx4_t z[8][2];
#pragma unroll
    for (int pb = 0; pb < 8; ++pb)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        z[pb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf_s[pb * 2 + j][lane]), bfrag, floatx4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    const int64_t pix = p0 + lc;
    const bool valid = pix < npix;
    f2 rs2 = f2{1.f, 1.f};
    if (LN) {
      const floatx4_t zs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf_s[16][lane]), bfrag,
                                                                    floatx4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      const float mean = zs[0] * inv_c;
      const f2 m2 = f2{mean, mean};
      f2 sq = f2{0.f, 0.f};
#pragma unroll
      for (int pb = 0; pb < 8; ++pb)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f2 d0 = f2{z[pb][j][0], z[pb][j][1]} - m2, d1 = f2{z[pb][j][2], z[pb][j][3]} - m2;
          sq = __builtin_elementwise_fma(d0, d0, sq);
          sq = __builtin_elementwise_fma(d1, d1, sq);
          z[pb][j][0] = d0.x; z[pb][j][1] = d0.y; z[pb][j][2] = d1.x; z[pb][j][3] = d1.y;
        }
      const float rstd = rsqrtf(c1_swap32_add(c1_swap16_add(sq.x + sq.y)) * inv_c + eps);
      rs2 = f2{rstd, rstd};
      if (g == 0 && valid) { mean_out[pix] = mean; rstd_out[pix] = rstd; }
    }
    // ---- normalise, ReLU (on the packed bf16 pair: a negative bf16 is a negative int16), rows through LDS: a lane holds 16-byte
    // pieces of ONE pixel, stored directly every instruction would write 64-byte pieces of 16 rows; through the (528-byte pitch)
    // row buffer an instruction writes two whole 512-byte rows (- 25 us)
    char* trw = tr_s + wave * (16 * 528) + lc * 528 + g * 16;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int pb = 0; pb < 8; ++pb) {
      uint32_t w[4];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f2 y0 = f2{z[pb][j][0], z[pb][j][1]}, y1 = f2{z[pb][j][2], z[pb][j][3]};
        if (LN) {
          const float4 gv = *reinterpret_cast<const float4*>(&gb_s[0][pb * 32 + g * 8 + j * 4]);
          const float4 bv = *reinterpret_cast<const float4*>(&gb_s[1][pb * 32 + g * 8 + j * 4]);
          y0 = __builtin_elementwise_fma(y0 * rs2, f2{gv.x, gv.y}, f2{bv.x, bv.y});
          y1 = __builtin_elementwise_fma(y1 * rs2, f2{gv.z, gv.w}, f2{bv.z, bv.w});
        }
        w[2 * j] = c1_relu_bf16x2(pack_bf16x2(y0.x, y0.y));
        w[2 * j + 1] = c1_relu_bf16x2(pack_bf16x2(y1.x, y1.y));
      }
      *reinterpret_cast<uint4*>(trw + pb * 64) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int prow = 2 * i + (lane >> 5);
      const uint4 v = *reinterpret_cast<const uint4*>(tr_s + wave * (16 * 528) + prow * 528 + (lane & 31) * 16);
      if (p0 + prow < npix) *reinterpret_cast<uint4*>(out + (p0 + prow) * C + (lane & 31) * 8) = v;
    }
  }
}

}  // namespace

// DMA cursor of the im2col operand when it is the row (RC) operand of the implicit GEMM (conv2 forward): the pixel of
// every chunk is fixed for the whole tile, so its (b, t, f) decomposition and the 9 tap-validity bits are computed
// once; a K step then only adds a wave-uniform tap/channel offset (the generic path re-derives three integer
// divisions and a bounds check per 16-byte chunk per K step, which made the kernel VALU-bound on address math).
namespace nstgemm {
template <typename T>
struct TileDma<T, MODE_RC, Im2colLoader<T>> {
  const Im2colLoader<T>* ld;
  int o0, r0, lane;
  bool fastc;            // a K step never straddles a tap (C % BK == 0, aligned start)
  const char* base[4];   // &x[b][2*to-1][2*fo-1][kchunk*E] of the chunk's pixel; only dereferenced under a valid tap
  uint32_t tapmask[4];   // bit kh*3+kw: that tap of the pixel lies inside the image
  __device__ __forceinline__ void init(const Im2colLoader<T>& l, int o0_, int r0_, int wave, int lane_) {
    ld = &l; o0 = o0_; r0 = r0_; lane = lane_;
    fastc = l.vec && (l.C % Tile<T>::BK) == 0 && (r0_ % Tile<T>::BK) == 0;
    if (!fastc) return;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int c = (s * 4 + wave) * 64 + lane;
      const int row = c >> 3, slot = c & 7;
      const int kchunk = slot ^ ((row >> 1) & 7);
      const int pixel = o0 + row;
      uint32_t q, fo, b, to;
      l.dF2.divmod((uint32_t)pixel, q, fo);
      l.dT2.divmod(q, b, to);
      uint32_t mask = 0;
      if (pixel < l.outer_limit) {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int ti = 2 * (int)to + kh - 1, fi = 2 * (int)fo + kw - 1;
            if (ti >= 0 && ti < l.T1 && fi >= 0 && fi < l.F1) mask |= 1u << (kh * 3 + kw);
          }
      }
      tapmask[s] = mask;
      base[s] = reinterpret_cast<const char*>(l.x + (((int64_t)b * l.T1 + (2 * (int)to - 1)) * l.F1 + (2 * (int)fo - 1)) * l.C +
                                              kchunk * Tile<T>::E);
    }
  }
  __device__ __forceinline__ void issue(int t, uint32_t tile_lds_addr, int wave) const {
    if (!fastc) {
      dma_tile<T, MODE_RC, Im2colLoader<T>>(*ld, o0, r0 + t * Tile<T>::BK, tile_lds_addr, wave, lane);
      return;
    }
    const int kk0 = r0 + t * Tile<T>::BK;  // wave-uniform
    const int tap = (int)ld->dC.div((uint32_t)kk0), c0 = kk0 - tap * ld->C;
    const int kh = tap / 3, kw = tap - kh * 3;
    const int64_t off = (((int64_t)kh * ld->F1 + kw) * ld->C + c0) * (int64_t)sizeof(T);
    const bool kin = kk0 < ld->contig_limit;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bool ok = kin && ((tapmask[s] >> tap) & 1u);
      const void* src = ok ? (const void*)(base[s] + off) : (const void*)g_nst_zero16;
      glds16(src, __builtin_amdgcn_readfirstlane(tile_lds_addr + (uint32_t)((s * 4 + wave) * 64) * 16u));
    }
  }
  int t_next;
  __device__ __forceinline__ void begin(int) { t_next = 0; }
  __device__ __forceinline__ void next(uint32_t, uint32_t tile_lds_addr, int wave) { issue(t_next++, tile_lds_addr, wave); }
};

// DMA cursor of the im2col operand when the PIXEL is the reduction index (OC: conv2 weight gradient, tile = 64 pixels x
// 128 consecutive kk).  The four chunks of a lane share their kk column (hence tap and channel offset, fixed for the
// whole unit); their pixels are p, p+16, p+32, p+48 and every K step moves all of them 64 pixels on.  The (b, to, fo)
// decomposition is therefore done once and then ADVANCED: fo += 64 % F2, to += 64 / F2 (+ carry), image wrap by one
// compare.  The generic path re-derives three integer divisions, a tap split and a 64-bit address per chunk per K step
// (~300 VALU instructions per step per lane against 32 MFMAs: the kernel was bound by address arithmetic).
template <typename T>
struct TileDma<T, MODE_OC, Im2colLoader<T>> {
  const Im2colLoader<T>* ld;
  int o0, r0, lane;
  bool fast;
  int fo[4], to[4], grow[4], pix[4];  // grow = b*T1 + 2*to (global input row of tap kh == 1)
  int kh, kw, dq, df, t_cur;
  uint32_t coff;                      // byte offset of the lane's 16-byte channel chunk inside a pixel
  bool cvalid;
  __device__ __forceinline__ void init(const Im2colLoader<T>& l, int o0_, int r0_, int wave, int lane_) {
    ld = &l; o0 = o0_; r0 = r0_; lane = lane_; t_cur = 0;
    dq = Tile<T>::BK / l.F2;
    df = Tile<T>::BK - dq * l.F2;
    fast = sizeof(T) == 2 && l.vec && (l.C % BM) == 0 && l.T2 >= dq + 2;
    if (!fast) return;
    constexpr int CPR = Tile<T>::OC_CPR;
    const int r_in = (lane / CPR) + wave * (64 / CPR);         // row of chunk s = 0; chunk s adds 4 * 64 / CPR rows
    const int c16 = lane % CPR;
    const int g = (r_in & 3) | (((r_in >> 3) & 1) << 2);       // unchanged by + multiples of 16 rows
    const int kk = o0 + (c16 ^ (g << 1)) * Tile<T>::E;
    const int tap = (int)l.dC.div((uint32_t)o0);               // the tile lies inside one tap (C % BM == 0)
    kh = tap / 3; kw = tap - kh * 3;
    cvalid = kk < l.contig_limit;
    coff = (uint32_t)((kk - tap * l.C) * (int)sizeof(T));
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int p = r0 + r_in + s * (4 * 64 / CPR);
      uint32_t q, f, b, t;
      l.dF2.divmod((uint32_t)p, q, f);
      l.dT2.divmod(q, b, t);
      pix[s] = p; fo[s] = (int)f; to[s] = (int)t; grow[s] = (int)b * l.T1 + 2 * (int)t;
    }
  }
  __device__ __forceinline__ void advance() {
    const int F2 = ld->F2, T2 = ld->T2, wrap = ld->T1 - 2 * T2;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      int f = fo[s] + df, inc = dq;
      if (f >= F2) { f -= F2; ++inc; }
      int t = to[s] + inc, gr = grow[s] + 2 * inc;
      if (t >= T2) { t -= T2; gr += wrap; }
      fo[s] = f; to[s] = t; grow[s] = gr; pix[s] += Tile<T>::BK;
    }
    ++t_cur;
  }
  __device__ __forceinline__ void put(uint32_t tile_lds_addr, int wave) const {
    const int T1 = ld->T1, F1 = ld->F1;
    const int64_t pix_bytes = (int64_t)ld->C * (int64_t)sizeof(T);
    const char* xb = reinterpret_cast<const char*>(ld->x) + coff;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int ti = 2 * to[s] + kh - 1, fi = 2 * fo[s] + kw - 1;
      const bool ok = cvalid && pix[s] < ld->outer_limit && (unsigned)ti < (unsigned)T1 && (unsigned)fi < (unsigned)F1;
      const void* src = ok ? (const void*)(xb + ((int64_t)(grow[s] + kh - 1) * F1 + fi) * pix_bytes) : (const void*)g_nst_zero16;
      glds16(src, __builtin_amdgcn_readfirstlane(tile_lds_addr + (uint32_t)((s * 4 + wave) * 64) * 16u));
    }
  }
  // K steps are requested in order (t == t_cur, or t_cur + a few in a prologue)
  __device__ __forceinline__ void issue(int t, uint32_t tile_lds_addr, int wave) {
    if (!fast) {
      dma_tile<T, MODE_OC, Im2colLoader<T>>(*ld, o0, r0 + t * Tile<T>::BK, tile_lds_addr, wave, lane);
      return;
    }
    while (t_cur < t) advance();
    put(tile_lds_addr, wave);
  }
  int t_next;
  __device__ __forceinline__ void begin(int) { t_next = 0; }
  __device__ __forceinline__ void next(uint32_t, uint32_t tile_lds_addr, int wave) { issue(t_next++, tile_lds_addr, wave); }
};
}  // namespace nstgemm

namespace {

// dgrad, one stride parity class (pt, pf): logical row = (b, th, fh) with ti = 2*th+pt, fi = 2*fh+pf.
// Reduction index r = tapidx*C + co over the class's valid taps: kh in {1} (pt=0) or {0,2} (pt=1), same for kw.
template <typename T>
struct DgradALoader {  // RC: outer = class row, contig = r
  const T* dy;
  int B, C, T2, F2, ct, cf, pt, pf, nkw;
  FastDiv dcf, dct, dC;
  int outer_limit, contig_limit, vec;
  void init_divs() { dcf.init(cf); dct.init(ct); dC.init(C); }
  __device__ __forceinline__ const T* addr(int b, int th, int fh, int r, bool& ok) const {
    uint32_t tapidx, co;
    dC.divmod((uint32_t)r, tapidx, co);
    const int ih = nkw == 2 ? (int)(tapidx >> 1) : (int)tapidx, iw = nkw == 2 ? (int)(tapidx & 1) : 0;
    // pt=0: kh=1 -> to = th ; pt=1: kh=0 -> to = th+1 (ih=0), kh=2 -> to = th (ih=1)
    const int to = pt ? (ih == 0 ? th + 1 : th) : th;
    const int fo = pf ? (iw == 0 ? fh + 1 : fh) : fh;
    ok = to < T2 && fo < F2;
    return dy + (((int64_t)b * T2 + to) * F2 + fo) * C + (int)co;
  }
  __device__ __forceinline__ const T* ptr(int outer, int contig) const {
    if (outer >= outer_limit || contig >= contig_limit) return nullptr;
    uint32_t q, fh, b, th;
    dcf.divmod((uint32_t)outer, q, fh);
    dct.divmod(q, b, th);
    bool ok;
    const T* p = addr((int)b, (int)th, (int)fh, contig, ok);
    return ok ? p : nullptr;
  }
  __device__ __forceinline__ uint4 load(int outer, int contig) const {
    uint4 r = make_uint4(0, 0, 0, 0);
    if (outer >= outer_limit || contig >= contig_limit) return r;
    uint32_t q, fh, b, th;
    dcf.divmod((uint32_t)outer, q, fh);
    dct.divmod(q, b, th);
    if (vec) {
      bool ok;
      const T* p = addr((int)b, (int)th, (int)fh, contig, ok);
      if (ok) r = *reinterpret_cast<const uint4*>(p);
      return r;
    }
    T tmp[Tile<T>::E];
#pragma unroll
    for (int e = 0; e < Tile<T>::E; ++e) {
      tmp[e] = (T)0;
      if (contig + e < contig_limit) {
        bool ok;
        const T* p = addr((int)b, (int)th, (int)fh, contig + e, ok);
        if (ok) tmp[e] = *p;
      }
    }
    memcpy(&r, tmp, 16);
    return r;
  }
};
template <typename T>
struct DgradBLoader {  // RC: outer = ci, contig = r ; element = w2[tap][ci][co]
  const T* w2;
  int C, pt, pf, nkw;
  FastDiv dC;
  int outer_limit, contig_limit, vec;
  void init_divs() { dC.init(C); }
  __device__ __forceinline__ const T* addr(int ci, int r) const {
    uint32_t tapidx, co;
    dC.divmod((uint32_t)r, tapidx, co);
    const int ih = nkw == 2 ? (int)(tapidx >> 1) : (int)tapidx, iw = nkw == 2 ? (int)(tapidx & 1) : 0;
    const int kh = pt ? (ih == 0 ? 0 : 2) : 1;
    const int kw = pf ? (iw == 0 ? 0 : 2) : 1;
    return w2 + ((int64_t)(kh * 3 + kw) * C + ci) * C + (int)co;
  }
  __device__ __forceinline__ const T* ptr(int outer, int contig) const {
    if (outer >= outer_limit || contig >= contig_limit) return nullptr;
    return addr(outer, contig);
  }
  __device__ __forceinline__ uint4 load(int outer, int contig) const {
    uint4 r = make_uint4(0, 0, 0, 0);
    if (outer >= outer_limit || contig >= contig_limit) return r;
    if (vec) return *reinterpret_cast<const uint4*>(addr(outer, contig));
    T tmp[Tile<T>::E];
#pragma unroll
    for (int e = 0; e < Tile<T>::E; ++e) tmp[e] = (contig + e < contig_limit) ? *addr(outer, contig + e) : (T)0;
    memcpy(&r, tmp, 16);
    return r;
  }
};
}  // namespace

// Same idea for the two operands of the dgrad implicit GEMM (see the im2col cursor above): per chunk the class row
// (or the input channel) is fixed, a K step adds a wave-uniform (tap, channel) offset.
namespace nstgemm {
template <typename T>
struct TileDma<T, MODE_RC, DgradALoader<T>> {
  const DgradALoader<T>* ld;
  int o0, r0, lane;
  bool fastc;
  const char* base[4];   // &dy[b][th][fh][kchunk*E]
  uint32_t tapmask[4];   // bit tapidx: (to, fo) of that tap lies inside dy
  __device__ __forceinline__ void init(const DgradALoader<T>& l, int o0_, int r0_, int wave, int lane_) {
    ld = &l; o0 = o0_; r0 = r0_; lane = lane_;
    fastc = l.vec && (l.C % Tile<T>::BK) == 0 && (r0_ % Tile<T>::BK) == 0;
    if (!fastc) return;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int c = (s * 4 + wave) * 64 + lane;
      const int row = c >> 3, slot = c & 7;
      const int kchunk = slot ^ ((row >> 1) & 7);
      const int orow = o0 + row;
      uint32_t q, fh, b, th;
      l.dcf.divmod((uint32_t)orow, q, fh);
      l.dct.divmod(q, b, th);
      uint32_t mask = 0;
      if (orow < l.outer_limit) {
#pragma unroll
        for (int tapidx = 0; tapidx < 4; ++tapidx) {
          const int ih = l.nkw == 2 ? (tapidx >> 1) : tapidx, iw = l.nkw == 2 ? (tapidx & 1) : 0;
          const int to = l.pt ? (ih == 0 ? (int)th + 1 : (int)th) : (int)th;
          const int fo = l.pf ? (iw == 0 ? (int)fh + 1 : (int)fh) : (int)fh;
          if (to < l.T2 && fo < l.F2) mask |= 1u << tapidx;
        }
      }
      tapmask[s] = mask;
      base[s] = reinterpret_cast<const char*>(l.dy + (((int64_t)b * l.T2 + (int)th) * l.F2 + (int)fh) * l.C + kchunk * Tile<T>::E);
    }
  }
  __device__ __forceinline__ void issue(int t, uint32_t tile_lds_addr, int wave) const {
    if (!fastc) {
      dma_tile<T, MODE_RC, DgradALoader<T>>(*ld, o0, r0 + t * Tile<T>::BK, tile_lds_addr, wave, lane);
      return;
    }
    const int kk0 = r0 + t * Tile<T>::BK;  // wave-uniform
    const int tapidx = (int)ld->dC.div((uint32_t)kk0), c0 = kk0 - tapidx * ld->C;
    const int ih = ld->nkw == 2 ? (tapidx >> 1) : tapidx, iw = ld->nkw == 2 ? (tapidx & 1) : 0;
    const int dto = (ld->pt && ih == 0) ? 1 : 0, dfo = (ld->pf && iw == 0) ? 1 : 0;
    const int64_t off = (((int64_t)dto * ld->F2 + dfo) * ld->C + c0) * (int64_t)sizeof(T);
    const bool kin = kk0 < ld->contig_limit;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bool ok = kin && ((tapmask[s] >> tapidx) & 1u);
      const void* src = ok ? (const void*)(base[s] + off) : (const void*)g_nst_zero16;
      glds16(src, __builtin_amdgcn_readfirstlane(tile_lds_addr + (uint32_t)((s * 4 + wave) * 64) * 16u));
    }
  }
  int t_next;
  __device__ __forceinline__ void begin(int) { t_next = 0; }
  __device__ __forceinline__ void next(uint32_t, uint32_t tile_lds_addr, int wave) { issue(t_next++, tile_lds_addr, wave); }
};

template <typename T>
struct TileDma<T, MODE_RC, DgradBLoader<T>> {
  const DgradBLoader<T>* ld;
  int o0, r0, lane;
  bool fastc;
  const char* base[4];   // &w2[0][ci][kchunk*E]
  bool ovalid[4];
  __device__ __forceinline__ void init(const DgradBLoader<T>& l, int o0_, int r0_, int wave, int lane_) {
    ld = &l; o0 = o0_; r0 = r0_; lane = lane_;
    fastc = l.vec && (l.C % Tile<T>::BK) == 0 && (r0_ % Tile<T>::BK) == 0;
    if (!fastc) return;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int c = (s * 4 + wave) * 64 + lane;
      const int row = c >> 3, slot = c & 7;
      const int kchunk = slot ^ ((row >> 1) & 7);
      const int ci = o0 + row;
      ovalid[s] = ci < l.outer_limit;
      base[s] = reinterpret_cast<const char*>(l.w2 + (int64_t)ci * l.C + kchunk * Tile<T>::E);
    }
  }
  __device__ __forceinline__ void issue(int t, uint32_t tile_lds_addr, int wave) const {
    if (!fastc) {
      dma_tile<T, MODE_RC, DgradBLoader<T>>(*ld, o0, r0 + t * Tile<T>::BK, tile_lds_addr, wave, lane);
      return;
    }
    const int kk0 = r0 + t * Tile<T>::BK;
    const int tapidx = (int)ld->dC.div((uint32_t)kk0), c0 = kk0 - tapidx * ld->C;
    const int ih = ld->nkw == 2 ? (tapidx >> 1) : tapidx, iw = ld->nkw == 2 ? (tapidx & 1) : 0;
    const int kh = ld->pt ? (ih == 0 ? 0 : 2) : 1, kw = ld->pf ? (iw == 0 ? 0 : 2) : 1;
    const int64_t off = ((int64_t)(kh * 3 + kw) * ld->C * ld->C + c0) * (int64_t)sizeof(T);
    const bool kin = kk0 < ld->contig_limit;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bool ok = kin && ovalid[s];
      const void* src = ok ? (const void*)(base[s] + off) : (const void*)g_nst_zero16;
      glds16(src, __builtin_amdgcn_readfirstlane(tile_lds_addr + (uint32_t)((s * 4 + wave) * 64) * 16u));
    }
  }
  int t_next;
  __device__ __forceinline__ void begin(int) { t_next = 0; }
  __device__ __forceinline__ void next(uint32_t, uint32_t tile_lds_addr, int wave) { issue(t_next++, tile_lds_addr, wave); }
};
}  // namespace nstgemm

namespace {
struct DgradRowMap {  // class row -> pixel row of dx [B*T1*F1]
  int T1, F1, ct, cf, pt, pf;
  FastDiv dcf, dct;
  void init_divs() { dcf.init(cf); dct.init(ct); }
  __device__ __forceinline__ int64_t operator()(int row) const {
    uint32_t q, fh, b, th;
    dcf.divmod((uint32_t)row, q, fh);
    dct.divmod(q, b, th);
    return ((int64_t)b * T1 + (2 * (int)th + pt)) * F1 + (2 * (int)fh + pf);
  }
};

template <typename T, typename OutT, int AMODE, int BMODE, bool USE_TR, typename AL, typename BL, typename RM>
__global__ void __launch_bounds__(THREADS) conv_gemm_kernel(AL la, BL lb, OutT* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                           int tiles_n, int ntiles, int kt_per_split, Epilogue ep, RM rm) {
  __shared__ __attribute__((aligned(16))) char smem[2 * Tile<T>::LDS_BYTES];
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int kt_total = (K + Tile<T>::BK - 1) / Tile<T>::BK;
  const int kt_first = blockIdx.z * kt_per_split;
  int kt_count = kt_total - kt_first;
  if (kt_count > kt_per_split) kt_count = kt_per_split;
  if (kt_count <= 0) return;
  gemm_block<T, OutT, AMODE, BMODE, USE_TR, AL, BL, RM>(la, lb, C + (int64_t)blockIdx.z * ep.slab_stride, ldc, M, N, tm * BM,
                                                        tn * BN, kt_first, kt_count, ep, smem, rm);
}

template <typename T, typename OutT, int AMODE, int BMODE, int NST, typename AL, typename BL, typename RM, bool CS = false>
__global__ void __launch_bounds__(THREADS) conv_gemm_kernel_v2(AL la, BL lb, OutT* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                              int tiles_n, int ntiles, int kt_per_split, Epilogue ep, RM rm) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int kt_total = (K + Tile<T>::BK - 1) / Tile<T>::BK;
  const int kt_first = blockIdx.z * kt_per_split;
  int kt_count = kt_total - kt_first;
  if (kt_count > kt_per_split) kt_count = kt_per_split;
  if (kt_count <= 0) return;
  gemm_block_v2<T, OutT, AMODE, BMODE, NST, AL, BL, RM, CS>(la, lb, C + (int64_t)blockIdx.z * ep.slab_stride, ldc, M, N, tm * BM,
                                                            tn * BN, kt_first, kt_count, ep, smem_dyn, rm);
}

// 128 x 256 tiles (8 waves) for outputs wider than 128 columns: the im2col / dy operand is read once per row panel
template <typename T, typename OutT, int AMODE, int BMODE, typename AL, typename BL, typename RM>
__global__ void __launch_bounds__(V2W_THREADS) conv_gemm_kernel_v2w(AL la, BL lb, OutT* __restrict__ C, int64_t ldc, int M, int N,
                                                                    int K, int tiles_n, int ntiles, Epilogue ep, RM rm) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int kt_total = (K + Tile<T>::BK - 1) / Tile<T>::BK;
  gemm_block_v2w<T, OutT, AMODE, BMODE, AL, BL, RM>(la, lb, C, ldc, M, N, tm * BM, tn * 2 * BN, kt_total, ep, smem_dyn, rm);
}


bool conv_use_wide() { return true; }

// dw2[i] (+)= sum_z slabs[z][i]  (dense [9C, C] output);  db2[j] (+)= sum_z cs_parts[z][j] when cs_parts != NULL
__global__ void __launch_bounds__(256) conv_splitk_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                                                int64_t total4, int split, int accumulate,
                                                                const float* __restrict__ cs_parts, float* __restrict__ cs_out,
                                                                int n, int cs_accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    // 8 slab loads in flight per round (one dependent round trip per slab otherwise); adds stay in slab order
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* src = slabs + i * 4;
    const int64_t zs = total4 * 4;
    int z = 0;
    for (; z + 8 <= split; z += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(src + (z + u) * zs);
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; z < split; ++z) {
      const float4 v = *reinterpret_cast<const float4*>(src + z * zs);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    float* o = out + i * 4;
    if (accumulate) {
      const float4 old = *reinterpret_cast<const float4*>(o);
      acc.x += old.x; acc.y += old.y; acc.z += old.z; acc.w += old.w;
    }
    *reinterpret_cast<float4*>(o) = acc;
  }
  if (cs_parts) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
      float acc = 0.f;
      for (int z = 0; z < split; ++z) acc += cs_parts[(int64_t)z * n + j];
      cs_out[j] = cs_accumulate ? cs_out[j] + acc : acc;
    }
  }
}

int conv_nst() { return 2; }   // stages of the v2 ring (three measured no faster in round 2 and are gone)

bool conv_use_v2() { return true; }

template <typename KernelT>
void conv_allow_big_lds(KernelT kernel, int bytes) {
  static thread_local const void* done[32];
  static thread_local int ndone = 0;
  for (int i = 0; i < ndone; ++i)
    if (done[i] == (const void*)kernel) return;
  (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (ndone < 32) done[ndone++] = (const void*)kernel;
}

#define NST_CONV_LAUNCH_V2(T_, OutT_, AM, BMO, AL, BL, RM, grid, ktps, ...)                                   \
  do {                                                                                                        \
    auto kfn = conv_gemm_kernel_v2<T_, OutT_, AM, BMO, 2, AL, BL, RM>;                                        \
    conv_allow_big_lds(kfn, 2 * V2_STAGE_BYTES);                                                              \
    kfn<<<grid, THREADS, 2 * V2_STAGE_BYTES, st>>>(__VA_ARGS__);                                              \
  } while (0)

bool conv_use_tr() { return true; }   // (the fragment path without ds_read_b64_tr_b16 is no longer instantiated)

Epilogue plain_epilogue() {
  Epilogue ep;
  memset(&ep, 0, sizeof(ep));
  ep.alpha = 1.f;
  ep.drop_inv_keep = 1.f;
  ep.posenc_period = 1;
  ep.emb_scale = 1.f;
  ep.slab_stride = 0;
  return ep;
}



// =============================================================================================
// conv2 on the phase-staggered 256 x 256 tile kernel (nst_gemm256.h), bf16, C == 256 (round 4).
//
// forward: y[pixel, co] = sum over (tap, ci) im2col[pixel, (tap, ci)] . w2[(tap, ci), co] + b2: a workgroup owns 256
// consecutive output pixels x all 256 output channels; a K step is one 64-channel slice of one tap (36 steps).  The A
// half-tile images (128 pixels x 64 channels, 128-byte rows) are gathered by LDS-DMA straight from the NHWC input: per
// (half, piece) a lane keeps the address of its pixel's tap (0, 0) and a 9-bit mask of the taps that fall inside the image;
// a K step adds a wave-uniform (tap, slice) offset, padding taps read the zero block.  Every input element crosses the L2
// interface 2.25 times this way (the LDS-resident patch kernels of rounds 2-3 fetched it once) -- but the loop runs at ~60 % of the MFMA peak
// instead of ~30 %, and the re-reads hit the L2: neighbouring taps of neighbouring pixels a few K steps apart.
// =============================================================================================
struct Im2colDma256 {
  const char* base[2][2];   // [half][piece]: &x[b][2*to - 1][2*fo - 1][kchunk * 8] of the chunk's pixel (never dereferenced at a padding tap)
  uint32_t mask[2];         // [piece]: bits 0..8 = valid taps of half 0's pixel, bits 16..24 of half 1's
  int F1, C;
  int tap[2], c0[2];        // per half (wave-uniform): the (tap, first channel) of the K step issued next
  __device__ __forceinline__ void init(const Im2colLoader<bf16_t>& l, int m0, int wave, int lane) {
    F1 = l.F1; C = l.C;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int c = (s * 8 + wave) * 64 + lane;
      const int row = c >> 3, slot = c & 7;
      const int kchunk = slot ^ ((row >> 1) & 7);
      uint32_t m = 0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int pixel = m0 + h * 128 + row;
        uint32_t q, fo, b, to;
        l.dF2.divmod((uint32_t)pixel, q, fo);
        l.dT2.divmod(q, b, to);
        uint32_t mm = 0;
        if (pixel < l.outer_limit) {
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              const int ti = 2 * (int)to + kh - 1, fi = 2 * (int)fo + kw - 1;
              if (ti >= 0 && ti < l.T1 && fi >= 0 && fi < l.F1) mm |= 1u << (kh * 3 + kw);
            }
        }
        m |= mm << (16 * h);
        base[h][s] = reinterpret_cast<const char*>(l.x + (((int64_t)b * l.T1 + (2 * (int)to - 1)) * l.F1 + (2 * (int)fo - 1)) * l.C +
                                                   kchunk * 8);
      }
      mask[s] = m;
    }
    tap[0] = tap[1] = 0;
    c0[0] = c0[1] = 0;
  }
  template <int H>
  __device__ __forceinline__ void issue(int /*t*/, uint32_t img, int wave) {
    const uint32_t dst = img + (uint32_t)wave * 1024u;
    const int tp = tap[H], kh = (tp * 11) >> 5, kw = tp - kh * 3;
    const int64_t off = (((int64_t)kh * F1 + kw) * C + c0[H]) * 2;   // wave-uniform
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bool ok = (mask[s] >> (16 * H + tp)) & 1u;
      const void* src = ok ? (const void*)(base[H][s] + off) : (const void*)g_nst_zero16;
      glds16(src, __builtin_amdgcn_readfirstlane(dst + (uint32_t)s * 8192u));
    }
    // K-step order: the nine TAPS of one 64-channel slice on consecutive K steps, then the next slice (round 6; it was the four
    // slices of one tap): neighbouring output pixels share input columns between taps (kw = 2 of pixel f is kw = 0 of pixel
    // f + 1), so the re-reads now come two K steps apart -- inside what the XCD's L2 holds -- instead of eight
    if (++tap[H] == 9) { tap[H] = 0; c0[H] += 64; }
  }
};

// The weight operand of the forward in the same K-step order: K step (slice, tap) reads rows [tap * C + c0, + 64) of w2 [9 C, C]
// (reduction-major = OC images, the chunk mapping of Dma256<MODE_OC>).
struct W2FwdDma256 {
  const char* p[2][2];      // [half][piece]: &w2[r][col] of the chunk at (tap 0, slice 0)
  int C;
  int tap[2], c0[2];
  __device__ __forceinline__ void init(const bf16_t* w2, int C_, int wave, int lane) {
    C = C_;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int c = (s * 8 + wave) * 64 + lane;
      const int r = c >> 4, c16 = c & 15;
      const int g = (r & 3) | (((r >> 3) & 1) << 2);
#pragma unroll
      for (int h = 0; h < 2; ++h) p[h][s] = reinterpret_cast<const char*>(w2 + (int64_t)r * C + h * 128 + (c16 ^ (g << 1)) * 8);
    }
    tap[0] = tap[1] = 0;
    c0[0] = c0[1] = 0;
  }
  template <int H>
  __device__ __forceinline__ void issue(int /*t*/, uint32_t img, int wave) {
    const uint32_t dst = img + (uint32_t)wave * 1024u;
    const int64_t off = ((int64_t)(tap[H] * C + c0[H]) * C) * 2;   // wave-uniform
    const char* s0 = p[H][0] + off;
    const char* s1 = p[H][1] + off;
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_add_u32 m0, m0, 0x2000\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(s0), "v"(s1), "s"(dst)
        : "memory", "scc");
    if (++tap[H] == 9) { tap[H] = 0; c0[H] += 64; }
  }
};

struct Conv2Fwd256Args {
  Im2colLoader<bf16_t> la;
  const bf16_t* w2;
  bf16_t* y;
  const float* bias;
  int M;
};

template <int EF>
__global__ void __launch_bounds__(G256_THREADS, 2) conv2_fwd256_kernel(Conv2Fwd256Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int m0 = blockIdx.x * G256_TILE;
  const int C = a.la.C;
  floatx4_t acc[2][4][4], cs[4];
  {
    Im2colDma256 da;
    da.init(a.la, m0, wave, lane);
    W2FwdDma256 db;           // Bop[j = co][r] = w2[r * C + co]  (reduction-major), K steps in the order of Im2colDma256
    db.init(a.w2, C, wave, lane);
    gemm256_mainloop<MODE_RC, MODE_OC, false, 0>(smem_dyn, da, db, (9 * C) >> 6, false, acc, cs);
  }
  asm volatile("" ::: "memory");
  Epilogue ep{};
  ep.vec = 1;
  ep.bias = a.bias;
  float* epi = reinterpret_cast<float*>(smem_dyn + wave * V3_EPI_BYTES_PER_WAVE);
  const IdentityRowMap rowmap;
  epilogue_v3<bf16_t, IdentityRowMap, EF>(acc[0], epi, a.y, (int64_t)C, a.M, C, m0 + wr * 128, wc * 64, ep, rowmap, lane);
  epilogue_v3<bf16_t, IdentityRowMap, EF>(acc[1], epi, a.y, (int64_t)C, a.M, C, m0 + wr * 128 + 64, wc * 64, ep, rowmap, lane);
}

// (the generic implicit-GEMM kernels below serve the shapes the 256 x 256 kernels do not take: C != 256, fp32, unaligned)
bool conv2_use_g256() { return true; }

// returns true when the 256 x 256 kernel handled the call
bool conv2_fwd_g256(const void* x, const void* w2, const float* b2, void* y, int B, int T1, int F1, int C, int relu, hipStream_t st) {
  const int T2 = (T1 + 1) / 2, F2 = (F1 + 1) / 2;
  const int64_t M = (int64_t)B * T2 * F2;
  if (!conv2_use_g256() || C != 256 || !b2 || !nst_aligned16(x) || !nst_aligned16(w2) || !nst_aligned16(y) ||
      (((uintptr_t)b2) & 15) != 0 || M >= (1ll << 30) || M < 256)
    return false;
  Conv2Fwd256Args a;
  a.la.x = (const bf16_t*)x; a.la.B = B; a.la.T1 = T1; a.la.F1 = F1; a.la.C = C; a.la.T2 = T2; a.la.F2 = F2;
  a.la.outer_limit = (int)M; a.la.contig_limit = 9 * C; a.la.vec = 1;
  a.la.init_divs();
  a.w2 = (const bf16_t*)w2; a.y = (bf16_t*)y; a.bias = b2; a.M = (int)M;
  const int grid = (int)((M + G256_TILE - 1) / G256_TILE);
  if (relu) {
    auto kfn = conv2_fwd256_kernel<EF_BIAS | EF_RELU>;
    conv_allow_big_lds(kfn, G256_LDS_BYTES);
    kfn<<<grid, G256_THREADS, G256_LDS_BYTES, st>>>(a);
  } else {
    auto kfn = conv2_fwd256_kernel<EF_BIAS>;
    conv_allow_big_lds(kfn, G256_LDS_BYTES);
    kfn<<<grid, G256_THREADS, G256_LDS_BYTES, st>>>(a);
  }
  return true;
}

template <typename T>
int conv2_fwd_t(const void* x, const void* w2, const float* b2, void* y, int B, int T1, int F1, int C, int relu, hipStream_t st) {
  const int T2 = (T1 + 1) / 2, F2 = (F1 + 1) / 2;
  const int M = B * T2 * F2, N = C, K = 9 * C;
  if constexpr (sizeof(T) == 2) {
    if (conv2_fwd_g256(x, w2, b2, y, B, T1, F1, C, relu, st)) return 0;
  }
  Im2colLoader<T> la;
  la.x = (const T*)x; la.B = B; la.T1 = T1; la.F1 = F1; la.C = C; la.T2 = T2; la.F2 = F2;
  la.outer_limit = M; la.contig_limit = K;
  la.vec = nst_aligned16(x) && (C % Tile<T>::E == 0);
  la.init_divs();
  DenseLoader<T> lb;  // Bop[j=co][r] = w2[r*C + co]  (OC: outer = r, contig = co)
  lb.base = (const T*)w2; lb.ld = C; lb.outer_limit = K; lb.contig_limit = N;
  lb.vec = nst_aligned16(w2) && (C % Tile<T>::E == 0);
  Epilogue ep = plain_epilogue();
  ep.bias = b2;
  ep.relu = relu;
  ep.vec = nst_aligned16(y) && (C % 8 == 0);
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, ntiles = tiles_m * tiles_n;
  const int kt_total = (K + Tile<T>::BK - 1) / Tile<T>::BK;
  dim3 grid(ntiles, 1, 1);
  if (conv_use_v2() && conv_use_wide() && conv_use_tr() && la.vec && lb.vec && N > BN) {
    const int tiles_nw = (N + 2 * BN - 1) / (2 * BN), ntw = tiles_m * tiles_nw;
    auto kfn = conv_gemm_kernel_v2w<T, T, MODE_RC, MODE_OC, Im2colLoader<T>, DenseLoader<T>, IdentityRowMap>;
    conv_allow_big_lds(kfn, V2W_LDS_BYTES);
    kfn<<<ntw, V2W_THREADS, V2W_LDS_BYTES, st>>>(la, lb, (T*)y, (int64_t)C, M, N, K, tiles_nw, ntw, ep, IdentityRowMap());
  } else if (conv_use_v2() && conv_use_tr() && la.vec && lb.vec)
    NST_CONV_LAUNCH_V2(T, T, MODE_RC, MODE_OC, Im2colLoader<T>, DenseLoader<T>, IdentityRowMap, grid, kt_total, la, lb, (T*)y,
                       (int64_t)C, M, N, K, tiles_n, ntiles, kt_total, ep, IdentityRowMap());
  else
    conv_gemm_kernel<T, T, MODE_RC, MODE_OC, true, Im2colLoader<T>, DenseLoader<T>, IdentityRowMap><<<grid, THREADS, 0, st>>>(la, lb, (T*)y, C, M, N, K, tiles_n, ntiles, kt_total, ep, IdentityRowMap());
  return 0;
}

// =============================================================================================
// conv2 data gradient on the phase-staggered 256 x 256 tile kernel (bf16, C == 256, even T1 and F1; round 4).
//
// dx[b, ti, fi, :] = sum over the taps (kh, kw) that reach it:  dy[b, (ti+1-kh)/2, (fi+1-kw)/2, :] . w2[kh, kw]^T.  With even
// T1 / F1 each of the four parity classes (ti & 1, fi & 1) of dx has exactly the T2 x F2 grid of dy, and class pixel (u, v)
// needs dy at (u + du, v + dv), du / dv in {0, 1} (1, 2, 2 and 4 taps).  A workgroup owns 256 consecutive dy pixels and
// produces the four classes of their 2 x 2 dx pixels one after the other (4 + 2 + 2 + 1 taps, 36 K steps of 64 output
// channels); both operands stream through the two K-step buffers of gemm256_mainloop:
//   A (RC images, 128 dy pixels x 64 co): the lane keeps the address of ITS pixel's dy row and a 4-bit mask of the shifts
//     (du, dv) that stay inside the image; a K step adds the wave-uniform ((du * F2 + dv) * C + co0) offset, a shift that
//     leaves the image reads the zero block.  A dy row crosses the L2 interface 9 times (once per tap)
//     -- all but the first are hits, the taps of one class and the classes of one tile follow each other within microseconds.
//   B (RC images, 128 ci x 64 co): w2[tap][ci][co] rows, 288 KB per class set, L2-resident.
// The next class's first loads are issued BEFORE the current class's store epilogue (whose transposition scratch lives in
// buffer 1's A images, the one region those loads do not touch), so only the first class of a tile pays the load latency.
// =============================================================================================
struct DyTapDma256 {
  const char* base;         // &dy[m0 + row][kchunk * 8]: piece s of half h sits (h * 128 + s * 64) rows further (same chunk swizzle)
  uint32_t mask[2];         // [piece]: bits 0..3 = shifts (du * 2 + dv) of half 0's pixel that stay inside, bits 16..19 of half 1's
  int F2, C, pf;
  int tp[2], c0[2];         // per half (wave-uniform): the (tap of the class, first channel) of the K step issued next
  __device__ __forceinline__ void init(const bf16_t* dy, int M, int T2, int F2_, int C_, const FastDiv& dF2, const FastDiv& dT2, int m0,
                                       int wave, int lane) {
    F2 = F2_; C = C_;
    const int row = wave * 8 + (lane >> 3), slot = lane & 7;
    const int kchunk = slot ^ ((row >> 1) & 7);
    base = reinterpret_cast<const char*>(dy + (int64_t)(m0 + row) * C + kchunk * 8);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      uint32_t m = 0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int pixel = m0 + h * 128 + s * 64 + row;
        uint32_t q, v, b, u;
        dF2.divmod((uint32_t)pixel, q, v);
        dT2.divmod(q, b, u);
        uint32_t mm = 0;
        if (pixel < M) {
          const bool un = (int)u + 1 < T2, vn = (int)v + 1 < F2;
          mm = 1u | (vn ? 2u : 0u) | (un ? 4u : 0u) | ((un && vn) ? 8u : 0u);
        }
        m |= mm << (16 * h);
      }
      mask[s] = m;
    }
  }
  int ntap;
  __device__ __forceinline__ void begin(int pt_, int pf_) { pf = pf_; ntap = 1 << (pt_ + pf_); tp[0] = tp[1] = 0; c0[0] = c0[1] = 0; }
  template <int H>
  __device__ __forceinline__ void issue(int /*t*/, uint32_t img, int wave) {
    const uint32_t dst = img + (uint32_t)wave * 1024u;
    const int du = tp[H] >> pf, dv = tp[H] & pf;          // taps of a class: dv fastest (pf == 0: dv == 0, du == tp)
    const int64_t off = ((int64_t)(du * F2 + dv + H * 128) * C + c0[H]) * 2;   // wave-uniform
    const int bit = 16 * H + du * 2 + dv;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bool ok = (mask[s] >> bit) & 1u;
      const void* src = ok ? (const void*)(base + (off + (int64_t)s * 64 * C * 2)) : (const void*)g_nst_zero16;
      glds16(src, __builtin_amdgcn_readfirstlane(dst + (uint32_t)s * 8192u));
    }
    // K-step order inside a class: the TAPS of one 64-channel slice follow each other, then the next slice (round 6; it was the
    // four slices of one tap).  The shifted windows of the taps are the same dy rows but for one pixel row / column, so
    // consecutive K steps of a workgroup re-read what is still in its XCD's L2 -- with the slices inner the re-use distance
    // was 4 K steps x 32 workgroups x 64 KB = twice the L2 and the counters showed dy crossing the HBM interface 7.7 times
    if (++tp[H] == ntap) { tp[H] = 0; c0[H] += 64; }
  }
};

struct W2TapDma256 {
  const char* base;         // &w2[0][row][kchunk * 8]: piece s of half h sits (h * 128 + s * 64) rows (input channels) further
  int C, pt, pf;
  int tp[2], c0[2];
  __device__ __forceinline__ void init(const bf16_t* w2, int C_, int wave, int lane) {
    C = C_;
    const int row = wave * 8 + (lane >> 3), slot = lane & 7;
    const int kchunk = slot ^ ((row >> 1) & 7);
    base = reinterpret_cast<const char*>(w2 + (int64_t)row * C + kchunk * 8);
  }
  int ntap;
  __device__ __forceinline__ void begin(int pt_, int pf_) { pt = pt_; pf = pf_; ntap = 1 << (pt_ + pf_); tp[0] = tp[1] = 0; c0[0] = c0[1] = 0; }
  template <int H>
  __device__ __forceinline__ void issue(int /*t*/, uint32_t img, int wave) {
    const uint32_t dst = img + (uint32_t)wave * 1024u;
    const int du = tp[H] >> pf, dv = tp[H] & pf;
    // dx row ti = 2u + pt takes dy row u + du through tap kh = ti + 1 - 2 (u + du): even rows kh = 1, odd rows kh = 2 - 2 du
    const int kh = pt ? 2 - 2 * du : 1, kw = pf ? 2 - 2 * dv : 1;
    const int64_t off = (((int64_t)(kh * 3 + kw) * C + H * 128) * C + c0[H]) * 2;   // wave-uniform
    const char* s0 = base + off;
    const char* s1 = s0 + (int64_t)64 * C * 2;
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_add_u32 m0, m0, 0x2000\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(s0), "v"(s1), "s"(dst)
        : "memory", "scc");
    if (++tp[H] == ntap) { tp[H] = 0; c0[H] += 64; }    // (the order of DyTapDma256: taps inner, slices outer)
  }
};

struct Conv2Dgrad256Args {
  const bf16_t* dy;
  const bf16_t* w2;
  bf16_t* dx;
  int M, T1, F1, T2, F2, C;
  FastDiv dF2, dT2;
};

__global__ void __launch_bounds__(G256_THREADS) conv2_dgrad256_kernel(Conv2Dgrad256Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int m0 = blockIdx.x * G256_TILE;
  const int C = a.C;
  DyTapDma256 da;
  da.init(a.dy, a.M, a.T2, a.F2, C, a.dF2, a.dT2, m0, wave, lane);
  W2TapDma256 db;
  db.init(a.w2, C, wave, lane);
  DgradRowMap rm;
  rm.T1 = a.T1; rm.F1 = a.F1; rm.ct = a.T2; rm.cf = a.F2; rm.dcf = a.dF2; rm.dct = a.dT2;
  Epilogue ep{};
  ep.vec = 1;
  // transposition scratch of the store epilogue: buffer 1's A images (see the header comment)
  float* epi = reinterpret_cast<float*>(smem_dyn + G256_KT_BYTES + wave * V3_EPI_BYTES_PER_WAVE);
  floatx4_t acc[2][4][4], cs[4];
  // heaviest class first: (odd, odd) 4 taps, then 2, 2, 1
  da.begin(1, 1);
  db.begin(1, 1);
  gemm256_prologue(smem_dyn, da, db, 16);
#pragma unroll 1
  for (int cls = 3; cls >= 0; --cls) {
    const int pt = cls >> 1, pf = cls & 1;
    const int nk = 4 << (pt + pf);
    gemm256_mainloop<MODE_RC, MODE_RC, false, 0, true>(smem_dyn, da, db, nk, false, acc, cs);
    asm volatile("" ::: "memory");
    if (cls > 0) {
      const int npt = (cls - 1) >> 1, npf = (cls - 1) & 1;
      da.begin(npt, npf);
      db.begin(npt, npf);
      gemm256_prologue(smem_dyn, da, db, 4 << (npt + npf));
    }
    rm.pt = pt; rm.pf = pf;
    epilogue_v3<bf16_t, DgradRowMap, 0>(acc[0], epi, a.dx, (int64_t)C, a.M, C, m0 + wr * 128, wc * 64, ep, rm, lane);
    epilogue_v3<bf16_t, DgradRowMap, 0>(acc[1], epi, a.dx, (int64_t)C, a.M, C, m0 + wr * 128 + 64, wc * 64, ep, rm, lane);
  }
}

// returns true when the 256 x 256 kernel handled the call
bool conv2_dgrad_g256(const void* dy, const void* w2, void* dx, int B, int T1, int F1, int C, hipStream_t st) {
  const int T2 = T1 / 2, F2 = F1 / 2;
  if (C != 256 || (T1 & 1) || (F1 & 1) || T1 < 2 || F1 < 2 || !nst_aligned16(dy) || !nst_aligned16(w2) || !nst_aligned16(dx))
    return false;
  const int64_t M = (int64_t)B * T2 * F2;
  if (M >= (1ll << 30) || M < 256) return false;
  Conv2Dgrad256Args a;
  a.dy = (const bf16_t*)dy; a.w2 = (const bf16_t*)w2; a.dx = (bf16_t*)dx;
  a.M = (int)M; a.T1 = T1; a.F1 = F1; a.T2 = T2; a.F2 = F2; a.C = C;
  a.dF2.init(F2); a.dT2.init(T2);
  const int grid = (int)((M + G256_TILE - 1) / G256_TILE);
  conv_allow_big_lds(conv2_dgrad256_kernel, G256_LDS_BYTES);
  conv2_dgrad256_kernel<<<grid, G256_THREADS, G256_LDS_BYTES, st>>>(a);
  return true;
}

template <typename T>
int conv2_dgrad_t(const void* dy, const void* w2, void* dx, int B, int T1, int F1, int C, hipStream_t st) {
  const int T2 = (T1 + 1) / 2, F2 = (F1 + 1) / 2;
  if constexpr (sizeof(T) == 2)
  {
    if (conv2_dgrad_g256(dy, w2, dx, B, T1, F1, C, st)) return 0;
  }
  for (int pt = 0; pt < 2; ++pt)
    for (int pf = 0; pf < 2; ++pf) {
      const int ct = (T1 - pt + 1) / 2, cf = (F1 - pf + 1) / 2;  // rows / cols of this parity
      if (ct <= 0 || cf <= 0) continue;
      const int nkh = pt ? 2 : 1, nkw = pf ? 2 : 1;
      const int M = B * ct * cf, N = C, K = nkh * nkw * C;
      DgradALoader<T> la;
      la.dy = (const T*)dy; la.B = B; la.C = C; la.T2 = T2; la.F2 = F2; la.ct = ct; la.cf = cf; la.pt = pt; la.pf = pf; la.nkw = nkw;
      la.outer_limit = M; la.contig_limit = K; la.vec = nst_aligned16(dy) && (C % Tile<T>::E == 0);
      la.init_divs();
      DgradBLoader<T> lb;
      lb.w2 = (const T*)w2; lb.C = C; lb.pt = pt; lb.pf = pf; lb.nkw = nkw;
      lb.outer_limit = N; lb.contig_limit = K; lb.vec = nst_aligned16(w2) && (C % Tile<T>::E == 0);
      lb.init_divs();
      DgradRowMap rm;
      rm.T1 = T1; rm.F1 = F1; rm.ct = ct; rm.cf = cf; rm.pt = pt; rm.pf = pf;
      rm.init_divs();
      Epilogue ep = plain_epilogue();
      ep.vec = nst_aligned16(dx) && (C % 8 == 0);
      const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, ntiles = tiles_m * tiles_n;
      const int kt_total = (K + Tile<T>::BK - 1) / Tile<T>::BK;
      dim3 grid(ntiles, 1, 1);
      if (conv_use_v2() && la.vec && lb.vec)
        NST_CONV_LAUNCH_V2(T, T, MODE_RC, MODE_RC, DgradALoader<T>, DgradBLoader<T>, DgradRowMap, grid, kt_total, la, lb, (T*)dx,
                           (int64_t)C, M, N, K, tiles_n, ntiles, kt_total, ep, rm);
      else
        conv_gemm_kernel<T, T, MODE_RC, MODE_RC, true, DgradALoader<T>, DgradBLoader<T>, DgradRowMap><<<grid, THREADS, 0, st>>>(la, lb, (T*)dx, C, M, N, K, tiles_n, ntiles, kt_total, ep, rm);
    }
  return 0;
}

// =============================================================================================
// conv2 weight gradient on the phase-staggered 256 x 256 tile kernel (bf16, C == 256; round 4).
//
// dw2[(tap, ci), co] = sum over output pixels p of x[pixel(p, tap), ci] . dy[p, co]: nine products (one per tap) of 256 x 256
// outputs over a reduction of P = B * T2 * F2 pixels -- far too few tiles for the chip, so the reduction is cut into S slices
// (9 S <= CUs units, each >= 32 K steps) whose partial sums go to slabs and through conv_splitk_reduce_kernel (66 MB next to
// 1.5 GB of operands on the benchmark shape).  Both operands are reduction-major:
//   A (OC images, 64 pixels x 128 ci): the rows of a K step are 64 consecutive output pixels; a lane decomposes ITS pixel
//     into (b, to, fo) with two fast divisions per K step and piece, adds the unit's tap and reads the 16-byte chunk of that
//     input pixel (the zero block for padding taps and past the last pixel);
//   B (OC images, 64 pixels x 128 co): dy rows, the dense cursor of the grouped weight gradients.
// The nine taps of a slice read the same dy rows and neighbouring x rows: they are dealt to ONE XCD (workgroup b runs on XCD
// b % 8).  db2 = column sums of dy from the ones-row MFMA of the tap-0 units.
// =============================================================================================
struct Im2colOcDma256 {
  const bf16_t* x;
  int T1, F1, C, P, kh, kw, k0;
  FastDiv dF2, dT2;
  int r[2];       // K-step row (0..63) of piece s
  int col[2];     // first channel of the lane's 16-byte chunk inside a half (swizzled by the row, see Dma256<MODE_OC>)
  __device__ __forceinline__ void init(const bf16_t* x_, int T1_, int F1_, int C_, int P_, const FastDiv& dF2_, const FastDiv& dT2_,
                                       int tap, int k0_, int wave, int lane) {
    x = x_; T1 = T1_; F1 = F1_; C = C_; P = P_; dF2 = dF2_; dT2 = dT2_; k0 = k0_;
    kh = (tap * 11) >> 5; kw = tap - kh * 3;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int c = (s * 8 + wave) * 64 + lane;
      const int rr = c >> 4, c16 = c & 15;
      const int g = (rr & 3) | (((rr >> 3) & 1) << 2);
      r[s] = rr;
      col[s] = (c16 ^ (g << 1)) * 8;
    }
  }
  template <int H>
  __device__ __forceinline__ void issue(int t, uint32_t img, int wave) {
    const uint32_t dst = img + (uint32_t)wave * 1024u;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int p = k0 + t * 64 + r[s];
      uint32_t q, fo, b, to;
      dF2.divmod((uint32_t)p, q, fo);
      dT2.divmod(q, b, to);
      const int ti = 2 * (int)to + kh - 1, fi = 2 * (int)fo + kw - 1;
      const bool ok = p < P && ti >= 0 && ti < T1 && fi >= 0 && fi < F1;
      const bf16_t* src = x + (((int64_t)b * T1 + ti) * F1 + fi) * C + H * 128 + col[s];
      glds16(ok ? (const void*)src : (const void*)g_nst_zero16, __builtin_amdgcn_readfirstlane(dst + (uint32_t)s * 8192u));
    }
  }
};

struct Conv2Wgrad256Args {
  const bf16_t* x;
  const bf16_t* dy;
  float* slabs;      // [nsplit][9 C][C]
  float* cs_parts;   // [nsplit][C] or NULL
  int P, T1, F1, C, KT, kps, nsplit;
  FastDiv dF2, dT2;
};

__global__ void __launch_bounds__(G256_THREADS) conv2_wgrad256_kernel(Conv2Wgrad256Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  // workgroup -> (K slice, tap): the nine taps of a slice on one XCD; slices beyond a multiple of 8 are spread over the XCDs
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, s8 = a.nsplit >> 3;
  int split, tap;
  if (j < 9 * s8) {
    const int jj = (j * 57) >> 9;   // j / 9 for j < 512
    split = xcd + 8 * jj; tap = j - 9 * jj;
  } else {
    const int e = (j - 9 * s8) * 8 + xcd;
    if (e >= 9 * (a.nsplit & 7)) return;
    const int ee = (e * 57) >> 9;
    split = 8 * s8 + ee; tap = e - 9 * ee;
  }
  const int t0 = split * a.kps;
  int nk = a.KT - t0;
  if (nk > a.kps) nk = a.kps;
  const int C = a.C;
  const bool do_cs = a.cs_parts != nullptr && tap == 0 && wr == 0;   // wave-uniform
  floatx4_t acc[2][4][4], cs[4];
  {
    Im2colOcDma256 da;
    da.init(a.x, a.T1, a.F1, C, a.P, a.dF2, a.dT2, tap, t0 * 64, wave, lane);
    DenseLoader<bf16_t> lb;   // dy [P rows][C]: reduction-major, output columns = co
    lb.base = a.dy; lb.ld = C; lb.outer_limit = a.P; lb.contig_limit = C; lb.vec = 1;
    Dma256<MODE_OC> db;
    db.init(lb, 0, t0 * 64, wave, lane);
    gemm256_mainloop<MODE_OC, MODE_OC, true>(smem_dyn, da, db, nk, do_cs, acc, cs);
  }
  asm volatile("" ::: "memory");
  if (do_cs && lane < 16) {
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) a.cs_parts[(int64_t)split * C + wc * 64 + jn * 16 + lane] = cs[jn][0];
  }
  Epilogue ep{};
  ep.vec = 1;
  float* epi = reinterpret_cast<float*>(smem_dyn + wave * V3_EPI_BYTES_PER_WAVE);
  float* Ct = a.slabs + ((int64_t)split * 9 + tap) * C * C;   // the unit's 256 x 256 tile of its slab
  const IdentityRowMap rowmap;
  epilogue_v3<float, IdentityRowMap, 0>(acc[0], epi, Ct, (int64_t)C, 256, C, wr * 128, wc * 64, ep, rowmap, lane);
  epilogue_v3<float, IdentityRowMap, 0>(acc[1], epi, Ct, (int64_t)C, 256, C, wr * 128 + 64, wc * 64, ep, rowmap, lane);
}

// returns true when the 256 x 256 kernel handled the call (slabs + reduce included)
bool conv2_wgrad_g256(const void* x, const void* dy, float* dw2, float* db2, int B, int T1, int F1, int C, int accumulate, void* ws,
                      int64_t ws_bytes, hipStream_t st, bool* db2_done) {
  const int T2 = (T1 + 1) / 2, F2 = (F1 + 1) / 2;
  const int64_t P = (int64_t)B * T2 * F2;
  if (C != 256 || !ws || !nst_aligned16(x) || !nst_aligned16(dy) || !nst_aligned16(dw2) || !nst_aligned16(ws) ||
      P >= (1ll << 30))
    return false;
  const int KT = (int)((P + 63) / 64);
  int cus = 256;
  {
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
  }
  int nsplit = cus / 9;                       // one unit per CU
  if (nsplit > KT / 32) nsplit = KT / 32;     // slices of >= 32 K steps
  if (nsplit > 56) nsplit = 56;
  if (nsplit < 8) return false;               // too little work to fill the chip this way
  const int kps = (KT + nsplit - 1) / nsplit;
  nsplit = (KT + kps - 1) / kps;              // every slice non-empty
  const int64_t M = 9 * (int64_t)C;
  if (ws_bytes < (int64_t)nsplit * (M + 1) * C * 4) return false;
  Conv2Wgrad256Args a;
  a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.slabs = (float*)ws;
  a.cs_parts = db2 ? (float*)ws + (int64_t)nsplit * M * C : nullptr;
  a.P = (int)P; a.T1 = T1; a.F1 = F1; a.C = C; a.KT = KT; a.kps = kps; a.nsplit = nsplit;
  a.dF2.init(F2); a.dT2.init(T2);
  const int s8 = nsplit >> 3, rest = 9 * (nsplit & 7);
  const int grid = 8 * (9 * s8 + (rest + 7) / 8);
  conv_allow_big_lds(conv2_wgrad256_kernel, G256_LDS_BYTES);
  conv2_wgrad256_kernel<<<grid, G256_THREADS, G256_LDS_BYTES, st>>>(a);
  const int64_t total4 = M * C / 4;
  const int blocks = (int)((total4 + 255) / 256 > 2048 ? 2048 : (total4 + 255) / 256);
  conv_splitk_reduce_kernel<<<blocks, 256, 0, st>>>((const float*)ws, dw2, total4, nsplit, accumulate, a.cs_parts, db2, C, accumulate);
  if (db2) *db2_done = true;
  return true;
}

template <typename T>
int conv2_wgrad_t(const void* x, const void* dy, float* dw2, float* db2, int B, int T1, int F1, int C, int accumulate, void* ws,
                  int64_t ws_bytes, hipStream_t st, bool* db2_done) {
  const int T2 = (T1 + 1) / 2, F2 = (F1 + 1) / 2;
  const int P = B * T2 * F2;          // reduction: pixels
  const int M = 9 * C, N = C, K = P;  // dw2[kk][co] = sum_p im2col[p][kk] * dy[p][co]
  if constexpr (sizeof(T) == 2)
    if (conv2_wgrad_g256(x, dy, dw2, db2, B, T1, F1, C, accumulate, ws, ws_bytes, st, db2_done)) return 0;
  Im2colLoader<T> la;                 // OC: outer = pixel (reduction), contig = kk (output row)
  la.x = (const T*)x; la.B = B; la.T1 = T1; la.F1 = F1; la.C = C; la.T2 = T2; la.F2 = F2;
  la.outer_limit = P; la.contig_limit = M;
  la.vec = nst_aligned16(x) && (C % Tile<T>::E == 0);
  la.init_divs();
  DenseLoader<T> lb;                  // Bop[j=co][r=p] = dy[p*C + co] (OC)
  lb.base = (const T*)dy; lb.ld = C; lb.outer_limit = P; lb.contig_limit = N;
  lb.vec = nst_aligned16(dy) && (C % Tile<T>::E == 0);
  Epilogue ep = plain_epilogue();
  ep.vec = nst_aligned16(dw2) && (C % 8 == 0);
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, ntiles = tiles_m * tiles_n;
  const int kt_total = (K + Tile<T>::BK - 1) / Tile<T>::BK;
  int split = (512 + ntiles - 1) / ntiles;  // ~2 workgroups per CU
  const bool dma = conv_use_v2() && conv_use_tr() && la.vec && lb.vec;
  if (split > kt_total) split = kt_total;
  if (split < 1) split = 1;
  const int kt_per_split = (kt_total + split - 1) / split;
  split = (kt_total + kt_per_split - 1) / kt_per_split;
  dim3 grid(ntiles, 1, split);
  const bool slab = ws && nst_aligned16(ws) && ws_bytes >= (int64_t)split * (M + 1) * N * 4 && (N % 8 == 0) && nst_aligned16(dw2);
  float* out = dw2;
  float* cs_parts = nullptr;
  if (slab) {
    ep.slab_stride = (int64_t)M * N;
    ep.vec = 1;
    out = (float*)ws;
    if (db2 && dma && conv_nst() < 3) {  // db2 = column sums of dy, accumulated by the same MFMA loop
      cs_parts = (float*)ws + (int64_t)split * M * N;
      ep.colsum_dst = cs_parts;
      ep.colsum_zstride = N;
    }
  } else {
    ep.atomic = 1;
    if (!accumulate) {
      if (hipMemsetAsync(dw2, 0, sizeof(float) * (size_t)M * N, st) != hipSuccess) return -1;
    }
  }
  if (cs_parts) {
    auto kfn = conv_gemm_kernel_v2<T, float, MODE_OC, MODE_OC, 2, Im2colLoader<T>, DenseLoader<T>, IdentityRowMap, true>;
    conv_allow_big_lds(kfn, 2 * V2_STAGE_BYTES);
    kfn<<<grid, THREADS, 2 * V2_STAGE_BYTES, st>>>(la, lb, out, (int64_t)N, M, N, K, tiles_n, ntiles, kt_per_split, ep, IdentityRowMap());
    *db2_done = true;
  } else if (dma)
    NST_CONV_LAUNCH_V2(T, float, MODE_OC, MODE_OC, Im2colLoader<T>, DenseLoader<T>, IdentityRowMap, grid, kt_per_split, la, lb, out,
                       (int64_t)N, M, N, K, tiles_n, ntiles, kt_per_split, ep, IdentityRowMap());
  else
    conv_gemm_kernel<T, float, MODE_OC, MODE_OC, true, Im2colLoader<T>, DenseLoader<T>, IdentityRowMap><<<grid, THREADS, 0, st>>>(la, lb, out, N, M, N, K, tiles_n, ntiles, kt_per_split, ep, IdentityRowMap());
  if (slab) {
    const int64_t total4 = (int64_t)M * N / 4;
    int blocks = (int)((total4 + 255) / 256 > 2048 ? 2048 : (total4 + 255) / 256);
    conv_splitk_reduce_kernel<<<blocks, 256, 0, st>>>((const float*)ws, dw2, total4, split, accumulate, cs_parts, db2, N, accumulate);
  }
  return 0;
}

int device_cu_count() {
  int cus = 256, dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  return cus > 0 ? cus : 256;
}

int check_conv_dims(const char* name, int B, int T, int F, int C) {
  NST_CHECK_ARG(B > 0 && T > 0 && F > 0 && C > 0, "%s: bad dims B=%d T=%d F=%d C=%d", name, B, T, F, C);
  NST_CHECK_ARG((int64_t)B * T * F < ((int64_t)1 << 30) && (int64_t)9 * C < ((int64_t)1 << 24), "%s: problem too large for 32-bit row indices", name);
  return NST_OK;
}

}  // namespace

extern "C" int nst_conv1_ln_relu_fwd(const float* src, const float* w1, const float* b1, const float* gamma,
                                     const float* beta, void* out, float* mean, float* rstd, int B, int T, int F, int C,
                                     int layer_norm, float eps, int out_dtype, void* stream) {
  NST_CHECK_ARG(src && w1 && b1 && out, "conv1_fwd: null pointer");
  NST_CHECK_ARG(!layer_norm || (gamma && beta && mean && rstd), "conv1_fwd: LayerNorm needs gamma/beta/mean/rstd");
  NST_CHECK_ARG(C > 0 && C <= 64 * C1_SLOTS, "conv1_fwd: C=%d unsupported (1..%d)", C, 64 * C1_SLOTS);
  NST_CHECK_ARG(B > 0 && T > 0 && F > 0, "conv1_fwd: bad dims");
  NST_CHECK_ARG(out_dtype == NST_F32 || out_dtype == NST_BF16, "conv1_fwd: bad dtype %d", out_dtype);
  NST_CHECK_ARG(F <= C1_MAXF, "conv1_fwd: F=%d unsupported (<= %d)", F, C1_MAXF);
  const int T1 = (T + 1) / 2, F1 = (F + 1) / 2;
  const int64_t nrows = (int64_t)B * T1;
  int blocks = (int)((nrows + 3) / 4 > 8192 ? 8192 : (nrows + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (C % 4 == 0) && ((((uintptr_t)out) & 15) == 0);
#define NST_C1F(TT, S, V) conv1_fwd_kernel<TT, S, V><<<blocks, 256, 0, st>>>(src, w1, b1, gamma, beta, (TT*)out, mean, rstd, B, T, F, C, T1, F1, layer_norm, eps)
  if (out_dtype == NST_F32) {
    if (vec) { if (C <= 256) NST_C1F(float, 4, true); else NST_C1F(float, 8, true); }
    else { if (C <= 256) NST_C1F(float, 4, false); else NST_C1F(float, 8, false); }
  } else if (vec && C == 256 && (int64_t)B * T1 * F1 < ((int64_t)1 << 31)) {   // the conv on the matrix cores (see the kernel)
    const int64_t npix = (int64_t)B * T1 * F1;
    FastDiv dF1, dT1;
    dF1.init((uint32_t)F1);
    dT1.init((uint32_t)T1);
    const int64_t wgs = (((npix + 15) >> 4) + 3) / 4;
    const int mb = (int)(wgs > 2 * device_cu_count() ? 2 * device_cu_count() : wgs);   // two workgroups per CU (77 KB of LDS each)
    if (layer_norm)
      conv1_fwd_mfma_kernel<true><<<mb, 256, 0, st>>>(src, w1, b1, gamma, beta, (bf16_t*)out, mean, rstd, T, F, T1, F1, npix, dF1, dT1, eps);
    else
      conv1_fwd_mfma_kernel<false><<<mb, 256, 0, st>>>(src, w1, b1, gamma, beta, (bf16_t*)out, mean, rstd, T, F, T1, F1, npix, dF1, dT1, eps);
  } else {
    if (vec) { if (C <= 256) NST_C1F(bf16_t, 4, true); else NST_C1F(bf16_t, 8, true); }
    else { if (C <= 256) NST_C1F(bf16_t, 4, false); else NST_C1F(bf16_t, 8, false); }
  }
#undef NST_C1F
  NST_CHECK_LAUNCH("conv1_fwd");
  return NST_OK;
}

extern "C" int nst_conv1_ln_relu_bwd(const float* src, const float* w1, const float* b1, const float* gamma,
                                     const float* beta, const float* mean, const float* rstd, const void* dout,
                                     float* dw1, float* db1, float* dgamma, float* dbeta, int B, int T, int F, int C,
                                     int layer_norm, float eps, int dtype, int accumulate, void* stream) {
  (void)eps;
  NST_CHECK_ARG(src && w1 && b1 && dout && dw1 && db1, "conv1_bwd: null pointer");
  NST_CHECK_ARG(!layer_norm || (gamma && beta && mean && rstd && dgamma && dbeta), "conv1_bwd: LayerNorm pointers missing");
  NST_CHECK_ARG(C > 0 && C <= 64 * C1_SLOTS, "conv1_bwd: C=%d unsupported", C);
  NST_CHECK_ARG(dtype == NST_F32 || dtype == NST_BF16, "conv1_bwd: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate) {
    NST_CHECK_HIP(hipMemsetAsync(dw1, 0, sizeof(float) * 9 * C, st));
    NST_CHECK_HIP(hipMemsetAsync(db1, 0, sizeof(float) * C, st));
    if (layer_norm) {
      NST_CHECK_HIP(hipMemsetAsync(dgamma, 0, sizeof(float) * C, st));
      NST_CHECK_HIP(hipMemsetAsync(dbeta, 0, sizeof(float) * C, st));
    }
  }
  NST_CHECK_ARG(F <= C1_MAXF, "conv1_bwd: F=%d unsupported (<= %d)", F, C1_MAXF);
  const int T1 = (T + 1) / 2, F1 = (F + 1) / 2;
  const int64_t nrows = (int64_t)B * T1;
  int blocks = (int)((nrows + 3) / 4 > 1024 ? 1024 : (nrows + 3) / 4);
  const bool vec = (C % 4 == 0) && ((((uintptr_t)dout) & 15) == 0);
  if (dtype == NST_BF16 && (C == 64 || C == 128 || C == 256) && (int64_t)B * T1 * F1 < ((int64_t)1 << 31)) {
    const int64_t npix = (int64_t)B * T1 * F1;
    FastDiv dF1, dT1;
    dF1.init((uint32_t)F1);
    dT1.init((uint32_t)T1);
    const int64_t ngroups = (npix + 31) / 32;
    int cus = 256;
    {
      int dev = 0;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      if (cus <= 0) cus = 256;
    }
    int mb = (int)((ngroups + C1M_WAVES - 1) / C1M_WAVES > cus ? cus : (ngroups + C1M_WAVES - 1) / C1M_WAVES);
#define NST_C1M(NCB, LN_)                                                                                              \
  do {                                                                                                                 \
    auto kfn = conv1_bwd_mfma_kernel<NCB, LN_>;                                                                         \
    conv_allow_big_lds(kfn, (int)sizeof(C1mLds<NCB>));                                                                  \
    kfn<<<mb, C1M_WAVES * 64, sizeof(C1mLds<NCB>), st>>>(src, w1, b1, gamma, beta, mean, rstd, (const bf16_t*)dout, dw1, \
                                                        db1, dgamma, dbeta, T, F, T1, F1, npix, dF1, dT1);             \
  } while (0)
    if (C == 256) {   // eight waves, two per SIMD (the one-wave-per-SIMD form below serves C = 64 / 128)
      typedef C1mLds<16, 8, 1> Lds2;
      const int mb2 = (int)((ngroups + 7) / 8 > cus ? cus : (ngroups + 7) / 8);
      if (layer_norm) {
        auto kfn = conv1_bwd_mfma_kernel<16, true, 8, 1, true>;
        conv_allow_big_lds(kfn, (int)sizeof(Lds2));
        kfn<<<mb2, 512, sizeof(Lds2), st>>>(src, w1, b1, gamma, beta, mean, rstd, (const bf16_t*)dout, dw1, db1, dgamma, dbeta, T, F,
                                            T1, F1, npix, dF1, dT1);
      } else {
        auto kfn = conv1_bwd_mfma_kernel<16, false, 8, 1, true>;
        conv_allow_big_lds(kfn, (int)sizeof(Lds2));
        kfn<<<mb2, 512, sizeof(Lds2), st>>>(src, w1, b1, gamma, beta, mean, rstd, (const bf16_t*)dout, dw1, db1, dgamma, dbeta, T, F,
                                            T1, F1, npix, dF1, dT1);
      }
      NST_CHECK_LAUNCH("conv1_bwd(mfma, v2)");
      return NST_OK;
    }
    if (layer_norm) { if (C == 128) NST_C1M(8, true); else NST_C1M(4, true); }
    else { if (C == 128) NST_C1M(8, false); else NST_C1M(4, false); }
#undef NST_C1M
    NST_CHECK_LAUNCH("conv1_bwd(mfma)");
    return NST_OK;
  }
#define NST_C1B(TT, S, V) conv1_bwd_kernel<TT, S, V><<<blocks, 256, 0, st>>>(src, w1, b1, gamma, beta, mean, rstd, (const TT*)dout, dw1, db1, dgamma, dbeta, B, T, F, C, T1, F1, layer_norm)
  if (dtype == NST_F32) {
    if (vec) { if (C <= 256) NST_C1B(float, 4, true); else NST_C1B(float, 8, true); }
    else { if (C <= 256) NST_C1B(float, 4, false); else NST_C1B(float, 8, false); }
  } else {
    if (vec) { if (C <= 256) NST_C1B(bf16_t, 4, true); else NST_C1B(bf16_t, 8, true); }
    else { if (C <= 256) NST_C1B(bf16_t, 4, false); else NST_C1B(bf16_t, 8, false); }
  }
#undef NST_C1B
  NST_CHECK_LAUNCH("conv1_bwd");
  return NST_OK;
}

extern "C" int nst_conv2_fwd(const void* x, const void* w2, const float* b2, void* y, int B, int T1, int F1, int C, int relu,
                             int dtype, void* stream) {
  NST_CHECK_ARG(x && w2 && y, "conv2_fwd: null pointer");
  int rc = check_conv_dims("conv2_fwd", B, T1, F1, C);
  if (rc) return rc;
  if (dtype == NST_F32) conv2_fwd_t<float>(x, w2, b2, y, B, T1, F1, C, relu, (hipStream_t)stream);
  else if (dtype == NST_BF16) conv2_fwd_t<bf16_t>(x, w2, b2, y, B, T1, F1, C, relu, (hipStream_t)stream);
  else { nst_set_error("conv2_fwd: bad dtype %d", dtype); return NST_ERR_INVALID_ARG; }
  NST_CHECK_LAUNCH("conv2_fwd");
  return NST_OK;
}

extern "C" int nst_conv2_dgrad(const void* dy, const void* w2, void* dx, int B, int T1, int F1, int C, int dtype, void* stream) {
  NST_CHECK_ARG(dy && w2 && dx, "conv2_dgrad: null pointer");
  int rc = check_conv_dims("conv2_dgrad", B, T1, F1, C);
  if (rc) return rc;
  if (dtype == NST_F32) conv2_dgrad_t<float>(dy, w2, dx, B, T1, F1, C, (hipStream_t)stream);
  else if (dtype == NST_BF16) conv2_dgrad_t<bf16_t>(dy, w2, dx, B, T1, F1, C, (hipStream_t)stream);
  else { nst_set_error("conv2_dgrad: bad dtype %d", dtype); return NST_ERR_INVALID_ARG; }
  NST_CHECK_LAUNCH("conv2_dgrad");
  return NST_OK;
}

extern "C" int nst_conv2_wgrad(const void* x, const void* dy, float* dw2, float* db2, int B, int T1, int F1, int C, int dtype,
                               int accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
  NST_CHECK_ARG(x && dy && dw2, "conv2_wgrad: null pointer");
  int rc = check_conv_dims("conv2_wgrad", B, T1, F1, C);
  if (rc) return rc;
  int r = 0;
  bool db2_done = false;
  if (dtype == NST_F32) r = conv2_wgrad_t<float>(x, dy, dw2, db2, B, T1, F1, C, accumulate, workspace, workspace_bytes, (hipStream_t)stream, &db2_done);
  else if (dtype == NST_BF16) r = conv2_wgrad_t<bf16_t>(x, dy, dw2, db2, B, T1, F1, C, accumulate, workspace, workspace_bytes, (hipStream_t)stream, &db2_done);
  else { nst_set_error("conv2_wgrad: bad dtype %d", dtype); return NST_ERR_INVALID_ARG; }
  if (r) { nst_set_error("conv2_wgrad: memset failed"); return NST_ERR_LAUNCH; }
  NST_CHECK_LAUNCH("conv2_wgrad");
  if (db2 && !db2_done) {  // operands not DMA-legal: separate column-sum pass over dy
    const int T2 = (T1 + 1) / 2, F2 = (F1 + 1) / 2;
    return nst_colsum(dy, db2, (int64_t)B * T2 * F2, C, C, dtype, accumulate, nullptr, 0, stream);
  }
  return NST_OK;
}
