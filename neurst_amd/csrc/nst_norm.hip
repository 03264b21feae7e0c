// LayerNorm forward/backward (optionally fused with ReLU) -- HBM-bound, one wavefront per row,
// wavefront-shuffle reductions, fp32 statistics.
//   replaces tf.keras.layers.LayerNormalization at neurst/layers/common_layers.py:64-65,77,
//   transformer_encoder.py:98-100,135, transformer_decoder.py:99-101,225 and (with ReLU)
//   audio_modalities.py:102-104.
#include "nst_common.h"

#include <type_traits>

namespace {

constexpr int LN_MAX_PER_LANE = 16;  // d <= 64*16 = 1024
constexpr int LN_WAVES = 4;

// Values are cached in registers with STATIC indices (runtime-indexed register arrays go to scratch):
// slot c of a lane holds element elem_index(lane, c); slots past the row end are zero and never stored.
template <int VEC>
__device__ __forceinline__ int elem_index(int lane, int c) {
  return VEC == 4 ? (lane * 4 + (c >> 2) * 256 + (c & 3)) : (lane + c * 64);
}

template <typename T, int VEC, int S>
__device__ __forceinline__ void load_row(const T* __restrict__ p, int d, int lane, float (&v)[S]) {
  if (VEC == 4) {
#pragma unroll
    for (int k = 0; k < S / 4; ++k) {
      const int e = lane * 4 + k * 256;
      if (e < d) {
        if (sizeof(T) == 2) {
          uint2 raw = *reinterpret_cast<const uint2*>(p + e);
          v[k * 4 + 0] = bf16_to_f32((bf16_t)(raw.x & 0xffff));
          v[k * 4 + 1] = bf16_to_f32((bf16_t)(raw.x >> 16));
          v[k * 4 + 2] = bf16_to_f32((bf16_t)(raw.y & 0xffff));
          v[k * 4 + 3] = bf16_to_f32((bf16_t)(raw.y >> 16));
        } else {
          float4 raw = *reinterpret_cast<const float4*>(p + e);
          v[k * 4 + 0] = raw.x; v[k * 4 + 1] = raw.y; v[k * 4 + 2] = raw.z; v[k * 4 + 3] = raw.w;
        }
      } else {
        v[k * 4 + 0] = 0.f; v[k * 4 + 1] = 0.f; v[k * 4 + 2] = 0.f; v[k * 4 + 3] = 0.f;
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < S; ++k) {
      const int e = lane + k * 64;
      v[k] = e < d ? to_f32<T>(p[e]) : 0.f;
    }
  }
}

template <typename T, int VEC, int S>
__device__ __forceinline__ void store_row(T* __restrict__ p, int d, int lane, const float (&v)[S]) {
  if (VEC == 4) {
#pragma unroll
    for (int k = 0; k < S / 4; ++k) {
      const int e = lane * 4 + k * 256;
      if (e < d) {
        if (sizeof(T) == 2) {
          uint2 raw;
          raw.x = pack_bf16x2(v[k * 4 + 0], v[k * 4 + 1]);
          raw.y = pack_bf16x2(v[k * 4 + 2], v[k * 4 + 3]);
          *reinterpret_cast<uint2*>(p + e) = raw;
        } else {
          *reinterpret_cast<float4*>(p + e) = make_float4(v[k * 4], v[k * 4 + 1], v[k * 4 + 2], v[k * 4 + 3]);
        }
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < S; ++k) {
      const int e = lane + k * 64;
      if (e < d) p[e] = from_f32<T>(v[k]);
    }
  }
}

template <typename T, int VEC, bool RELU, int S>
__global__ void __launch_bounds__(LN_WAVES * 64) ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, T* __restrict__ y,
                                                              float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                              int64_t rows, int d, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float inv_d = 1.0f / (float)d;
  // two rows in flight per wave: both rows' loads are issued before either reduction chain starts
  const int64_t stride = (int64_t)gridDim.x * LN_WAVES;
  for (int64_t row0 = (int64_t)blockIdx.x * LN_WAVES + wave; row0 < rows; row0 += 2 * stride) {
    const int64_t row1 = row0 + stride;
    const bool has1 = row1 < rows;
    float v[2][S];
    load_row<T, VEC, S>(x + row0 * d, d, lane, v[0]);
    if (has1) load_row<T, VEC, S>(x + row1 * d, d, lane, v[1]);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && !has1) break;
      const int64_t row = u == 0 ? row0 : row1;
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < S; ++c) s += v[u][c];
      const float mean = wave_sum_fast(s) * inv_d;
      float sq = 0.f;
#pragma unroll
      for (int c = 0; c < S; ++c) {
        const float t = elem_index<VEC>(lane, c) < d ? v[u][c] - mean : 0.f;
        sq += t * t;
      }
      const float rstd = rsqrtf(wave_sum_fast(sq) * inv_d + eps);
#pragma unroll
      for (int c = 0; c < S; ++c) {
        const int e = elem_index<VEC>(lane, c);
        if (e < d) {
          float o = (v[u][c] - mean) * rstd * gamma[e] + beta[e];
          if (RELU) o = fmaxf(o, 0.f);
          v[u][c] = o;
        }
      }
      store_row<T, VEC, S>(y + row * d, d, lane, v[u]);
      if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
      }
    }
  }
}

template <typename T, int VEC, bool RELU, int S>
__global__ void __launch_bounds__(LN_WAVES * 64) ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                              const T* __restrict__ yout, const float* __restrict__ gamma,
                                                              const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                              const T* __restrict__ dres, T* __restrict__ dx, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int64_t rows, int d,
                                                              float* __restrict__ partial) {
  __shared__ float red[LN_WAVES][256];  // reused twice (dgamma then dbeta) in 4 slices
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float inv_d = 1.0f / (float)d;
  float g_acc[S], b_acc[S];
#pragma unroll
  for (int c = 0; c < S; ++c) { g_acc[c] = 0.f; b_acc[c] = 0.f; }
  const int64_t stride = (int64_t)gridDim.x * LN_WAVES;
  for (int64_t row0 = (int64_t)blockIdx.x * LN_WAVES + wave; row0 < rows; row0 += 2 * stride) {
    const int64_t row1 = row0 + stride;
    const bool has1 = row1 < rows;
    float xv[2][S], gv[2][S], rv[2][S];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && !has1) break;
      const int64_t row = u == 0 ? row0 : row1;
      load_row<T, VEC, S>(x + row * d, d, lane, xv[u]);
      load_row<T, VEC, S>(dy + row * d, d, lane, gv[u]);
      if (RELU) load_row<T, VEC, S>(yout + row * d, d, lane, rv[u]);
      else if (dres) load_row<T, VEC, S>(dres + row * d, d, lane, rv[u]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && !has1) break;
      const int64_t row = u == 0 ? row0 : row1;
      if (RELU) {
#pragma unroll
        for (int c = 0; c < S; ++c) gv[u][c] = rv[u][c] > 0.f ? gv[u][c] : 0.f;
      }
      const float mean = mean_in[row], rstd = rstd_in[row];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int c = 0; c < S; ++c) {
        const int e = elem_index<VEC>(lane, c);
        if (e < d) {
          const float xhat = (xv[u][c] - mean) * rstd;
          const float dxh = gv[u][c] * gamma[e];
          g_acc[c] += gv[u][c] * xhat;
          b_acc[c] += gv[u][c];
          s1 += dxh;
          s2 += dxh * xhat;
          xv[u][c] = xhat;
          gv[u][c] = dxh;
        }
      }
      const float c1 = wave_sum_fast(s1) * inv_d, c2m = wave_sum_fast(s2) * inv_d;
#pragma unroll
      for (int c = 0; c < S; ++c) xv[u][c] = rstd * (gv[u][c] - c1 - xv[u][c] * c2m);
      if (!RELU && dres) {
#pragma unroll
        for (int c = 0; c < S; ++c) xv[u][c] += rv[u][c];
      }
      store_row<T, VEC, S>(dx + row * d, d, lane, xv[u]);
    }
  }
  // cross-wave reduction in LDS, 4 cached values at a time, then one atomic per column per block
#pragma unroll
  for (int base = 0; base < S; base += 4) {
    if (elem_index<VEC>(0, base) >= d) break;  // block-uniform: no lane owns a column in this slice
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 4; ++j) red[wave][lane * 4 + j] = pass == 0 ? g_acc[base + j] : b_acc[base + j];
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int e = elem_index<VEC>(lane, base + j);
          if (e < d) {
            float t = red[0][lane * 4 + j] + red[1][lane * 4 + j] + red[2][lane * 4 + j] + red[3][lane * 4 + j];
            if (partial) partial[((int64_t)blockIdx.x * 2 + pass) * d + e] = t;   // plain store, reduced by ln_bwd_finalize
            else atomicAdd((pass == 0 ? dgamma : dbeta) + e, t);
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Wide path (d % 8 == 0, 16-byte aligned rows): every lane moves 8 consecutive elements per access (16 bytes of bf16,
// 2 x 16 bytes of f32), LPR lanes cover one row, so a wave pass covers 64/LPR consecutive rows and U passes are in
// flight before the first reduction -- the kernels are HBM-latency bound otherwise (one 512-byte row per wave).
// Row reductions: 4 DPP steps inside a 16-lane row, then gfx950 lane-row swaps (v_permlane16/32_swap).
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(2))) unsigned ln_uint2_t;
__device__ __forceinline__ float swap16_add(float v) {
  const ln_uint2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap32_add(float v) {
  const ln_uint2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
  v = dpp_add(v, 0); v = dpp_add(v, 1); v = dpp_add(v, 2); v = dpp_add(v, 3);
  if (LPR >= 32) v = swap16_add(v);
  if (LPR >= 64) v = swap32_add(v);
  return v;
}
template <typename T>
__device__ __forceinline__ void load8(const T* __restrict__ p, float (&v)[8]) {
  if constexpr (sizeof(T) == 2) {
    const uint4 raw = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(raw.x << 16); v[1] = __uint_as_float(raw.x & 0xffff0000u);
    v[2] = __uint_as_float(raw.y << 16); v[3] = __uint_as_float(raw.y & 0xffff0000u);
    v[4] = __uint_as_float(raw.z << 16); v[5] = __uint_as_float(raw.z & 0xffff0000u);
    v[6] = __uint_as_float(raw.w << 16); v[7] = __uint_as_float(raw.w & 0xffff0000u);
  } else {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
}
template <typename T>
__device__ __forceinline__ void store8(T* __restrict__ p, const float (&v)[8]) {
  if constexpr (sizeof(T) == 2) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                              pack_bf16x2(v[6], v[7]));
  } else {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
}
__device__ __forceinline__ void zero8(float (&v)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = 0.f;
}

// TX / ADD: the fp32 residual stream of the pre-norm bf16 path (nst_add_layernorm_fwd).  x is then read as TX (f32, or bf16
// for the first sub-layer of a stack), the sub-layer contribution `delta` (type T) is added to it, the sum is written once as
// f32 (xout, the value the backward normalises again and the next sub-layer adds to) and normalised into y (type T).
template <typename T, int LPR, int S, bool RELU, int U, typename TX = T, bool ADD = false>
__global__ void __launch_bounds__(256) ln_fwd_wide_kernel(const TX* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, T* __restrict__ y,
                                                         float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                         int64_t rows, int d, float eps, const T* __restrict__ delta = nullptr,
                                                         float* __restrict__ xout = nullptr) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, li = lane % LPR;
  float gm[S][8], bt[S][8];
  bool cok[S];
#pragma unroll
  for (int c = 0; c < S; ++c) {
    const int col = (li + c * LPR) * 8;
    cok[c] = col < d;
    if (cok[c]) { load8<float>(gamma + col, gm[c]); load8<float>(beta + col, bt[c]); }
    else { zero8(gm[c]); zero8(bt[c]); }
  }
  const float inv_d = 1.0f / (float)d;
  const int64_t step = (int64_t)gridDim.x * 4 * RPW * U;
  for (int64_t base = ((int64_t)blockIdx.x * 4 + wave) * RPW * U; base < rows; base += step) {
    float v[U][S][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = base + u * RPW + sub;
#pragma unroll
      for (int c = 0; c < S; ++c) {
        if (row < rows && cok[c]) load8<TX>(x + row * d + (li + c * LPR) * 8, v[u][c]);
        else zero8(v[u][c]);
      }
    }
    if constexpr (ADD) {
      float dl[U][S][8];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t row = base + u * RPW + sub;
#pragma unroll
        for (int c = 0; c < S; ++c) {
          if (row < rows && cok[c]) load8<T>(delta + row * d + (li + c * LPR) * 8, dl[u][c]);
          else zero8(dl[u][c]);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t row = base + u * RPW + sub;
#pragma unroll
        for (int c = 0; c < S; ++c) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[u][c][j] += dl[u][c][j];
          if (xout && row < rows && cok[c]) store8<float>(xout + row * d + (li + c * LPR) * 8, v[u][c]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = base + u * RPW + sub;
      float sum = 0.f;
#pragma unroll
      for (int c = 0; c < S; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += v[u][c][j];
      const float mean = group_sum<LPR>(sum) * inv_d;
      float sq = 0.f;
#pragma unroll
      for (int c = 0; c < S; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float t = cok[c] ? v[u][c][j] - mean : 0.f;
          sq += t * t;
        }
      const float rstd = rsqrtf(group_sum<LPR>(sq) * inv_d + eps);
#pragma unroll
      for (int c = 0; c < S; ++c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float o = (v[u][c][j] - mean) * rstd * gm[c][j] + bt[c][j];
          if (RELU) o = fmaxf(o, 0.f);
          v[u][c][j] = o;
        }
        if (row < rows && cok[c]) store8<T>(y + row * d + (li + c * LPR) * 8, v[u][c]);
      }
      if (li == 0 && row < rows) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
      }
    }
  }
}

template <typename T, int LPR, int S, bool RELU, int U, typename TX = T>
__global__ void __launch_bounds__(256) ln_bwd_wide_kernel(const T* __restrict__ dy, const TX* __restrict__ x,
                                                         const T* __restrict__ yout, const float* __restrict__ gamma,
                                                         const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                         const T* __restrict__ dres, T* __restrict__ dx, int64_t rows, int d,
                                                         float* __restrict__ partial, T* __restrict__ dz, uint32_t dz_thresh,
                                                         float dz_inv_keep, uint64_t dz_seed, uint64_t dz_sid,
                                                         const uint64_t* __restrict__ seed_dev,
                                                         const float* __restrict__ beta = nullptr) {
  if (dz) dz_seed = seed_with_offset(dz_seed, seed_dev);   // wave-uniform
  // RELU with yout == NULL: the gate relu'(LN(x)) is recomputed from x, the saved statistics and beta -- the same expression the
  // forward evaluated -- instead of read back from the saved activation (a quarter of this kernel's traffic)
  const bool regate = RELU && yout == nullptr;
  constexpr int RPW = 64 / LPR;
  constexpr int W = LPR * S * 8;  // padded row width
  __shared__ float red[4][2][W];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, li = lane % LPR;
  float gm[S][8], g_acc[S][8], b_acc[S][8];
  float bt[RELU ? S : 1][8];
  bool cok[S];
#pragma unroll
  for (int c = 0; c < S; ++c) {
    const int col = (li + c * LPR) * 8;
    cok[c] = col < d;
    if (cok[c]) load8<float>(gamma + col, gm[c]); else zero8(gm[c]);
    if constexpr (RELU) {
      if (cok[c] && regate) load8<float>(beta + col, bt[c]); else zero8(bt[c]);
    }
    zero8(g_acc[c]); zero8(b_acc[c]);
  }
  const float inv_d = 1.0f / (float)d;
  const int64_t step = (int64_t)gridDim.x * 4 * RPW * U;
  for (int64_t base = ((int64_t)blockIdx.x * 4 + wave) * RPW * U; base < rows; base += step) {
    float xv[U][S][8], gv[U][S][8], rv[U][S][8];
    float mu[U], rs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = base + u * RPW + sub;
      const bool rok = row < rows;
      mu[u] = rok ? mean_in[row] : 0.f;
      rs[u] = rok ? rstd_in[row] : 0.f;
#pragma unroll
      for (int c = 0; c < S; ++c) {
        const int64_t off = row * d + (li + c * LPR) * 8;
        if (rok && cok[c]) {
          load8<TX>(x + off, xv[u][c]);
          load8<T>(dy + off, gv[u][c]);
          if (RELU) { if (!regate) load8<T>(yout + off, rv[u][c]); else zero8(rv[u][c]); }
          else if (dres) load8<T>(dres + off, rv[u][c]);
          else zero8(rv[u][c]);
        } else {
          zero8(xv[u][c]); zero8(gv[u][c]); zero8(rv[u][c]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = base + u * RPW + sub;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int c = 0; c < S; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float g = gv[u][c][j];
          const float xhat = cok[c] ? (xv[u][c][j] - mu[u]) * rs[u] : 0.f;
          if constexpr (RELU) {
            const bool open_ = regate ? (xhat * gm[c][j] + bt[c][j]) > 0.f : rv[u][c][j] > 0.f;
            g = open_ ? g : 0.f;
          }
          const float dxh = g * gm[c][j];
          g_acc[c][j] += g * xhat;
          b_acc[c][j] += g;
          s1 += dxh;
          s2 += dxh * xhat;
          xv[u][c][j] = xhat;
          gv[u][c][j] = dxh;
        }
      const float c1 = group_sum<LPR>(s1) * inv_d, c2m = group_sum<LPR>(s2) * inv_d;
#pragma unroll
      for (int c = 0; c < S; ++c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float o = rs[u] * (gv[u][c][j] - c1 - xv[u][c][j] * c2m);
          if (!RELU) o += rv[u][c][j];
          xv[u][c][j] = o;
        }
        if (row < rows && cok[c]) {
          const int64_t off = row * d + (li + c * LPR) * 8;
          store8<T>(dx + off, xv[u][c]);
          if (dz) {  // second output: the dropout backward of dx for the sublayer that consumes it next
            float m[8];
            dropout_keep8(dz_seed, dz_sid, (uint64_t)off, dz_thresh, dz_inv_keep, m);
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[u][c][j] *= m[j];
            store8<T>(dz + off, xv[u][c]);
          }
        }
      }
    }
  }
  // lanes with the same li (different rows of a pass) hold partial sums of the same columns
#pragma unroll
  for (int c = 0; c < S; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (LPR <= 16) { g_acc[c][j] = swap16_add(g_acc[c][j]); b_acc[c][j] = swap16_add(b_acc[c][j]); }
      if (LPR <= 32) { g_acc[c][j] = swap32_add(g_acc[c][j]); b_acc[c][j] = swap32_add(b_acc[c][j]); }
    }
  if (sub == 0) {
#pragma unroll
    for (int c = 0; c < S; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        red[wave][0][(li + c * LPR) * 8 + j] = g_acc[c][j];
        red[wave][1][(li + c * LPR) * 8 + j] = b_acc[c][j];
      }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * W; i += 256) {
    const int pass = i / W, e = i % W;
    if (e < d)
      partial[((int64_t)blockIdx.x * 2 + pass) * d + e] = (red[0][pass][e] + red[1][pass][e]) + (red[2][pass][e] + red[3][pass][e]);
  }
}

// dgamma/dbeta (+)= sum over workgroups of partial[block][2][d].  16 columns x 16 row groups per workgroup: every
// thread adds rows rg, rg+16, ... (independent loads), then the 16 row groups are combined through LDS.
// Second stage of the dgamma / dbeta reduction: column e of the [blocks][2 d] partial sums.  1024 threads = 16 columns x 64 row
// groups: with <= 512 partial rows every thread has <= 8 loads, all in flight at once (the 256-thread version walked 32 partial
// rows per thread in dependent batches: 30 us per launch, 0.55 ms per step in round 2).  The summation order is fixed, and the
// same for the immediate and the deferred (multi-job) launch.
__device__ __forceinline__ void ln_finalize_body(const float* __restrict__ partial, float* __restrict__ dgamma,
                                                 float* __restrict__ dbeta, int blocks, int d, int accumulate, float (*sh)[17]) {
  const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + c;  // 0 .. 2*d-1
  float t = 0.f;
  if (e < 2 * d) {
    int b = rg;
    for (; b + 448 < blocks; b += 512) {   // 8 independent loads per round; the adds stay in row order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(b + 64 * u) * 2 * d + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) t += v[u];
    }
    for (; b < blocks; b += 64) t += partial[(int64_t)b * 2 * d + e];
  }
  sh[rg][c] = t;
  __syncthreads();
  if (rg < 4 && e < 2 * d) {   // 64 -> 4 partial sums in parallel, then one thread per column finishes
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += sh[rg * 16 + k][c];
    sh[rg * 16][c] = acc;
  }
  __syncthreads();
  if (rg == 0 && e < 2 * d) {
    const float acc = (sh[0][c] + sh[16][c]) + (sh[32][c] + sh[48][c]);
    float* o = e < d ? dgamma + e : dbeta + (e - d);
    *o = accumulate ? *o + acc : acc;
  }
}

// The same reduction for d % 4 == 0 (every wide-path caller: 16-byte aligned partial rows): a workgroup of 256 threads owns 32
// columns = 8 float4 column groups x 32 row groups, so a wave reads whole 128-byte pieces of 2 partial rows per instruction
// and every thread has its (<= 16 at 512 partial rows) loads in flight at once.  The 1024-thread body above walks 64-byte pieces
// (16 columns) and took 50 - 114 us per 16-job launch in round 5 (0.32 ms per step next to the input-gradient chain).
// The summation order is fixed: rows rg, rg + 32, ... per thread, then the 32 row groups in a fixed tree.
__device__ __forceinline__ void ln_finalize_body4(const float* __restrict__ partial, float* __restrict__ dgamma,
                                                  float* __restrict__ dbeta, int blocks, int d, int accumulate, float4 (*sh)[9]) {
  const int c4 = threadIdx.x & 7, rg = threadIdx.x >> 3;
  const int e = blockIdx.x * 32 + c4 * 4;  // 0 .. 2*d-4, multiple of 4
  float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < 2 * d) {
    int b = rg;
    for (; b + 96 < blocks; b += 128) {   // 4 independent loads per round; the adds stay in row order
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(partial + (int64_t)(b + 32 * u) * 2 * d + e);
#pragma unroll
      for (int u = 0; u < 4; ++u) { t.x += v[u].x; t.y += v[u].y; t.z += v[u].z; t.w += v[u].w; }
    }
    for (; b < blocks; b += 32) {
      const float4 v = *reinterpret_cast<const float4*>(partial + (int64_t)b * 2 * d + e);
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
  }
  sh[rg][c4] = t;
  __syncthreads();
  if (rg < 4) {   // 32 -> 4 partial sums in parallel, then one thread per column group finishes
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 8; ++k) { const float4 v = sh[rg * 8 + k][c4]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    sh[rg * 8][c4] = acc;
  }
  __syncthreads();
  if (rg == 0 && e < 2 * d) {
    const float4 a0 = sh[0][c4], a1 = sh[8][c4], a2 = sh[16][c4], a3 = sh[24][c4];
    const float r[4] = {(a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                        (a0.w + a1.w) + (a2.w + a3.w)};
#pragma unroll
    for (int j = 0; j < 4; ++j) {   // (d % 4 == 0: the four columns lie on one side of the dgamma | dbeta boundary)
      float* o = e < d ? dgamma + e + j : dbeta + (e - d) + j;
      *o = accumulate ? *o + r[j] : r[j];
    }
  }
}

__global__ void __launch_bounds__(1024) ln_bwd_finalize_kernel(const float* __restrict__ partial, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int blocks, int d, int accumulate) {
  __shared__ float sh[64][17];
  ln_finalize_body(partial, dgamma, dbeta, blocks, d, accumulate, sh);
}
__global__ void __launch_bounds__(256) ln_bwd_finalize4_kernel(const float* __restrict__ partial, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int blocks, int d, int accumulate) {
  __shared__ float4 sh[32][9];
  ln_finalize_body4(partial, dgamma, dbeta, blocks, d, accumulate, sh);
}
inline bool ln_finalize_vec_ok(const float* partial, int d) { return d % 4 == 0 && ((((uintptr_t)partial) & 15) == 0); }

// 8 elements per lane access: d multiple of 8 (rows then stay 16-byte aligned for bf16, 32 for f32), aligned bases
template <typename T>
bool wide_ok(int d, const void* a, const void* b, const void* c, const void* e) {
  return d % 8 == 0 && d <= 1024 && ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)e) & 15) == 0);
}

template <typename T, bool RELU>
int launch_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int64_t rows,
               int d, float eps, hipStream_t st) {
  if (wide_ok<T>(d, x, y, gamma, beta)) {
    const int lpr = d <= 128 ? 16 : (d <= 256 ? 32 : 64);
    const int U_ = 4;   // rows in flight per lane group
    int64_t wb = (rows + 4 * (64 / lpr) * U_ - 1) / (4 * (64 / lpr) * U_);
    if (wb > 65535) wb = 65535;
#define NST_LN_FWDW(L, S, UU) ln_fwd_wide_kernel<T, L, S, RELU, UU><<<(int)wb, 256, 0, st>>>((const T*)x, gamma, beta, (T*)y, mean, rstd, rows, d, eps)
    if (lpr == 16) NST_LN_FWDW(16, 1, 4); else if (lpr == 32) NST_LN_FWDW(32, 1, 4);
    else if (d <= 512) NST_LN_FWDW(64, 1, 4); else NST_LN_FWDW(64, 2, 4);
#undef NST_LN_FWDW
    return 0;
  }
  int64_t blocks = (rows + LN_WAVES - 1) / LN_WAVES;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
#define NST_LN_FWD(V, S) ln_fwd_kernel<T, V, RELU, S><<<(int)blocks, LN_WAVES * 64, 0, st>>>((const T*)x, gamma, beta, (T*)y, mean, rstd, rows, d, eps)
  // the fallback (d not a multiple of 8, d > 1024's neighbours, unaligned rows): ONE scalar-access variant for every width
  NST_LN_FWD(1, 16);
#undef NST_LN_FWD
  return 0;
}

// the add + LayerNorm forward of the fp32 residual stream: wide shapes only (the host keeps the bf16 stream otherwise)
template <typename TX>
int launch_add_fwd(const void* x, const void* delta, float* xout, const float* gamma, const float* beta, void* y, float* mean,
                   float* rstd, int64_t rows, int d, float eps, hipStream_t st) {
  typedef bf16_t T;
  const int lpr = d <= 128 ? 16 : (d <= 256 ? 32 : 64);
  // rows in flight per lane group: two row operands per row here, so half of the plain forward's four
  const int U_ = 2;
  int64_t wb = (rows + 4 * (64 / lpr) * U_ - 1) / (4 * (64 / lpr) * U_);
  if (wb > 65535) wb = 65535;
#define NST_LN_ADDW(L, S) ln_fwd_wide_kernel<T, L, S, false, 2, TX, true><<<(int)wb, 256, 0, st>>>((const TX*)x, gamma, beta, (T*)y, mean, rstd, rows, d, eps, (const T*)delta, xout)
  if (lpr == 16) NST_LN_ADDW(16, 1); else if (lpr == 32) NST_LN_ADDW(32, 1);
  else if (d <= 512) NST_LN_ADDW(64, 1); else NST_LN_ADDW(64, 2);
#undef NST_LN_ADDW
  return 0;
}

template <typename T, bool RELU, typename TX = T>
int launch_bwd(const void* dy, const void* x, const void* y, const float* gamma, const float* mean, const float* rstd,
               const void* dres, void* dx, float* dgamma, float* dbeta, int64_t rows, int d, float* partial, int* nblocks,
               hipStream_t st, void* dz, uint32_t dz_thresh, float dz_inv_keep, uint64_t dz_seed, uint64_t dz_sid, bool* dz_done,
               const float* beta = nullptr) {
  if (RELU && !y && !(partial && beta)) return -1;   // the recomputed gate exists on the wide path only
  if (partial && wide_ok<T>(d, x, dy, dx, RELU ? y : dres) && wide_ok<T>(d, gamma, beta, nullptr, nullptr)) {
    const int lpr = d <= 128 ? 16 : (d <= 256 ? 32 : 64);
    // rows in flight per lane group.  d_model = 256: one row (fewer registers, 4 waves per SIMD) measured 0.05 ms per step
    // faster than two
    // (also for the front end's 576 000-row call: 256 us with one row, 267 with two, profiles/r04_history/c32_conv_bench_u*.json)
    const int U_ = (lpr == 32 && d <= 256) ? 1 : 2;
    int64_t wb = (rows + 4 * (64 / lpr) * U_ - 1) / (4 * (64 / lpr) * U_);
    if (wb > 512) wb = 512;
    *nblocks = (int)wb;
#define NST_LN_BWDW(L, S, UU) ln_bwd_wide_kernel<T, L, S, RELU, UU, TX><<<(int)wb, 256, 0, st>>>((const T*)dy, (const TX*)x, (const T*)y, gamma, mean, rstd, (const T*)dres, (T*)dx, rows, d, partial, (T*)dz, dz_thresh, dz_inv_keep, dz_seed, dz_sid, nst_seed_offset_devptr(), beta)
    *dz_done = true;
    if (lpr == 16) NST_LN_BWDW(16, 1, 2); else if (lpr == 32) NST_LN_BWDW(32, 1, 1);
    else if (d <= 512) NST_LN_BWDW(64, 1, 2); else NST_LN_BWDW(64, 2, 2);
#undef NST_LN_BWDW
    return 0;
  }
  if constexpr (!std::is_same<T, TX>::value) return -1;   // mixed input types exist on the wide path only (caller reports it)
  if (RELU && !y) return -1;
  int64_t blocks = (rows + 2 * LN_WAVES - 1) / (2 * LN_WAVES);
  const int64_t cap = partial ? 1024 : 512;  // atomics: one per column per block, keep the fan-in per address small
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  *nblocks = (int)blocks;
#define NST_LN_BWD(V, S) ln_bwd_kernel<T, V, RELU, S><<<(int)blocks, LN_WAVES * 64, 0, st>>>((const T*)dy, (const T*)x, (const T*)y, gamma, mean, rstd, (const T*)dres, (T*)dx, dgamma, dbeta, rows, d, partial)
  if constexpr (std::is_same<T, TX>::value) NST_LN_BWD(1, 16);   // see launch_fwd
#undef NST_LN_BWD
  return 0;
}

int ln_fwd_common(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int64_t rows,
                  int d, float eps, int dtype, void* stream, bool relu) {
  NST_CHECK_ARG(x && gamma && beta && y, "layernorm_fwd: null pointer");
  NST_CHECK_ARG(d > 0 && d <= 64 * LN_MAX_PER_LANE, "layernorm_fwd: d=%d unsupported (1..%d)", d, 64 * LN_MAX_PER_LANE);
  NST_CHECK_ARG(dtype == NST_F32 || dtype == NST_BF16, "layernorm_fwd: bad dtype %d", dtype);
  if (rows <= 0) return NST_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == NST_F32) {
    if (relu) launch_fwd<float, true>(x, gamma, beta, y, mean, rstd, rows, d, eps, st);
    else launch_fwd<float, false>(x, gamma, beta, y, mean, rstd, rows, d, eps, st);
  } else {
    if (relu) launch_fwd<bf16_t, true>(x, gamma, beta, y, mean, rstd, rows, d, eps, st);
    else launch_fwd<bf16_t, false>(x, gamma, beta, y, mean, rstd, rows, d, eps, st);
  }
  NST_CHECK_LAUNCH("layernorm_fwd");
  return NST_OK;
}

// up to 16 deferred finalize stages in one launch: blockIdx.y = job
struct LnJobs { NstLnFinalizeJob j[16]; };
__global__ void __launch_bounds__(1024) ln_bwd_finalize_multi_kernel(LnJobs jobs) {
  const NstLnFinalizeJob& q = jobs.j[blockIdx.y];
  __shared__ float sh[64][17];
  if (blockIdx.x * 16 >= 2 * q.d) return;  // (block-uniform) jobs narrower than the widest one
  ln_finalize_body(q.partial, q.dgamma, q.dbeta, q.nblocks, q.d, q.accumulate, sh);
}
__global__ void __launch_bounds__(256) ln_bwd_finalize4_multi_kernel(LnJobs jobs) {
  const NstLnFinalizeJob& q = jobs.j[blockIdx.y];
  __shared__ float4 sh[32][9];
  if (blockIdx.x * 32 >= 2 * q.d) return;  // (block-uniform) jobs narrower than the widest one
  ln_finalize_body4(q.partial, q.dgamma, q.dbeta, q.nblocks, q.d, q.accumulate, sh);
}

int ln_bwd_common(const void* dy, const void* x, const void* y, const float* gamma, const float* mean,
                  const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta, int64_t rows, int d, int dtype,
                  int accumulate, void* ws, int64_t ws_bytes, void* stream, bool relu, void* dz = nullptr, float dz_p = 0.f,
                  uint64_t dz_seed = 0, uint64_t dz_sid = 0, NstLnFinalizeJob* job_out = nullptr, int x_dtype = -1,
                  const float* beta = nullptr) {
  if (job_out) memset(job_out, 0, sizeof(*job_out));
  if (x_dtype < 0) x_dtype = dtype;
  const bool x32 = x_dtype == NST_F32 && dtype == NST_BF16;   // the saved input of the fp32 residual stream
  NST_CHECK_ARG(x_dtype == dtype || x32, "layernorm_bwd: x_dtype %d with dtype %d (only f32 x with bf16 gradients is mixed)", x_dtype, dtype);
  NST_CHECK_ARG(!(x32 && relu), "layernorm_bwd: the ReLU variant has no mixed-type form");
  NST_CHECK_ARG(dy && x && gamma && mean && rstd && dx && dgamma && dbeta, "layernorm_bwd: null pointer");
  NST_CHECK_ARG(!relu || y || beta, "layernorm_relu_bwd: y (the saved activation) or beta (to recompute the gate) is required");
  NST_CHECK_ARG(d > 0 && d <= 64 * LN_MAX_PER_LANE, "layernorm_bwd: d=%d unsupported", d);
  NST_CHECK_ARG(dtype == NST_F32 || dtype == NST_BF16, "layernorm_bwd: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  NST_CHECK_ARG(dz_p >= 0.f && dz_p < 1.f, "layernorm_bwd: dropout_p=%f", dz_p);
  uint32_t dz_thresh = 0;
  float dz_inv_keep = 1.f;
  nst_dropout_params16(dz_p, &dz_thresh, &dz_inv_keep);
  bool dz_done = false;
  float* partial = (ws && ws_bytes >= (int64_t)2048 * 2 * d * 4 && ((((uintptr_t)ws) & 3) == 0)) ? (float*)ws : nullptr;
  if (!accumulate && (!partial || rows <= 0)) {
    NST_CHECK_HIP(hipMemsetAsync(dgamma, 0, sizeof(float) * d, st));
    NST_CHECK_HIP(hipMemsetAsync(dbeta, 0, sizeof(float) * d, st));
  }
  if (rows <= 0) return NST_OK;
  int nblocks = 0;
  if (x32) {
    if (!partial || launch_bwd<bf16_t, false, float>(dy, x, y, gamma, mean, rstd, dres, dx, dgamma, dbeta, rows, d, partial, &nblocks, st, dz, dz_thresh, dz_inv_keep, dz_seed, dz_sid, &dz_done) != 0) {
      nst_set_error("layernorm_bwd: f32 x with bf16 gradients needs d %% 8 == 0, d <= 1024, 16-byte aligned rows and a workspace");
      return NST_ERR_UNSUPPORTED;
    }
  } else if (relu && !y) {
    const int rc = dtype == NST_F32
        ? launch_bwd<float, true>(dy, x, nullptr, gamma, mean, rstd, dres, dx, dgamma, dbeta, rows, d, partial, &nblocks, st, dz, dz_thresh, dz_inv_keep, dz_seed, dz_sid, &dz_done, beta)
        : launch_bwd<bf16_t, true>(dy, x, nullptr, gamma, mean, rstd, dres, dx, dgamma, dbeta, rows, d, partial, &nblocks, st, dz, dz_thresh, dz_inv_keep, dz_seed, dz_sid, &dz_done, beta);
    if (rc != 0) {
      nst_set_error("layernorm_relu_bwd without y: needs d %% 8 == 0, d <= 1024, 16-byte aligned rows and a workspace");
      return NST_ERR_UNSUPPORTED;
    }
  } else if (dtype == NST_F32) {
    if (relu) launch_bwd<float, true>(dy, x, y, gamma, mean, rstd, dres, dx, dgamma, dbeta, rows, d, partial, &nblocks, st, dz, dz_thresh, dz_inv_keep, dz_seed, dz_sid, &dz_done);
    else launch_bwd<float, false>(dy, x, y, gamma, mean, rstd, dres, dx, dgamma, dbeta, rows, d, partial, &nblocks, st, dz, dz_thresh, dz_inv_keep, dz_seed, dz_sid, &dz_done);
  } else {
    if (relu) launch_bwd<bf16_t, true>(dy, x, y, gamma, mean, rstd, dres, dx, dgamma, dbeta, rows, d, partial, &nblocks, st, dz, dz_thresh, dz_inv_keep, dz_seed, dz_sid, &dz_done);
    else launch_bwd<bf16_t, false>(dy, x, y, gamma, mean, rstd, dres, dx, dgamma, dbeta, rows, d, partial, &nblocks, st, dz, dz_thresh, dz_inv_keep, dz_seed, dz_sid, &dz_done);
  }
  NST_CHECK_LAUNCH("layernorm_bwd");
  if (partial && job_out) {
    job_out->partial = partial; job_out->dgamma = dgamma; job_out->dbeta = dbeta;
    job_out->nblocks = nblocks; job_out->d = d; job_out->accumulate = accumulate;
  } else if (partial) {
    if (ln_finalize_vec_ok(partial, d)) ln_bwd_finalize4_kernel<<<(2 * d + 31) / 32, 256, 0, st>>>(partial, dgamma, dbeta, nblocks, d, accumulate);
    else ln_bwd_finalize_kernel<<<(2 * d + 15) / 16, 1024, 0, st>>>(partial, dgamma, dbeta, nblocks, d, accumulate);
    NST_CHECK_LAUNCH("layernorm_bwd(finalize)");
  }
  if (dz && !dz_done)  // narrow / unaligned rows: separate element-wise pass
    return nst_scale_dropout_bwd(dx, dz, rows * d, 1.0f, dz_p, dz_seed, dz_sid, dtype, stream);
  return NST_OK;
}

}  // namespace

extern "C" int nst_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                 int64_t rows, int d, float eps, int dtype, void* stream) {
  return ln_fwd_common(x, gamma, beta, y, mean, rstd, rows, d, eps, dtype, stream, false);
}
extern "C" int nst_layernorm_relu_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                      float* rstd, int64_t rows, int d, float eps, int dtype, void* stream) {
  return ln_fwd_common(x, gamma, beta, y, mean, rstd, rows, d, eps, dtype, stream, true);
}
extern "C" int nst_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                 const void* dres, void* dx, float* dgamma, float* dbeta, int64_t rows, int d, int dtype,
                                 int accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
  return ln_bwd_common(dy, x, nullptr, gamma, mean, rstd, dres, dx, dgamma, dbeta, rows, d, dtype, accumulate, workspace,
                       workspace_bytes, stream, false);
}
extern "C" int nst_layernorm_bwd_dropout(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                         const void* dres, void* dx, void* dz, float dropout_p, uint64_t seed,
                                         uint64_t stream_id, float* dgamma, float* dbeta, int64_t rows, int d, int dtype,
                                         int accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
  NST_CHECK_ARG(dz, "layernorm_bwd_dropout: null dz");
  return ln_bwd_common(dy, x, nullptr, gamma, mean, rstd, dres, dx, dgamma, dbeta, rows, d, dtype, accumulate, workspace,
                       workspace_bytes, stream, false, dz, dropout_p, seed, stream_id);
}
extern "C" int nst_layernorm_relu_bwd(const void* dy, const void* x, const void* y, const float* gamma, const float* mean,
                                      const float* rstd, void* dx, float* dgamma, float* dbeta, int64_t rows, int d,
                                      int dtype, int accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
  return ln_bwd_common(dy, x, y, gamma, mean, rstd, nullptr, dx, dgamma, dbeta, rows, d, dtype, accumulate, workspace,
                       workspace_bytes, stream, true);
}
extern "C" int nst_layernorm_bwd_deferred(const void* dy, const void* x, const void* y, const float* gamma, const float* mean,
                                          const float* rstd, const void* dres, void* dx, void* dz, float dropout_p,
                                          uint64_t seed, uint64_t stream_id, float* dgamma, float* dbeta, int64_t rows, int d,
                                          int dtype, int accumulate, void* workspace, int64_t workspace_bytes,
                                          NstLnFinalizeJob* job_out, void* stream) {
  NST_CHECK_ARG(job_out, "layernorm_bwd_deferred: null job_out");
  NST_CHECK_ARG(!(y && (dres || dz)), "layernorm_bwd_deferred: the ReLU variant takes neither dres nor dz");
  return ln_bwd_common(dy, x, y, gamma, mean, rstd, dres, dx, dgamma, dbeta, rows, d, dtype, accumulate, workspace,
                       workspace_bytes, stream, y != nullptr, dz, dz ? dropout_p : 0.f, seed, stream_id, job_out);
}
// fp32 residual stream of the pre-norm bf16 path (PrePostProcessingWrapper, neurst/layers/common_layers.py:73-85):
//   x_new = x + delta (f32 sum, written to x_out when it is not null) ;  y = LayerNorm(x_new) in bf16
extern "C" int nst_add_layernorm_fwd(const void* x, int x_dtype, const void* delta, void* x_out, const float* gamma,
                                     const float* beta, void* y, float* mean, float* rstd, int64_t rows, int d, float eps,
                                     int dtype, void* stream) {
  NST_CHECK_ARG(x && delta && gamma && beta && y, "add_layernorm_fwd: null pointer");
  NST_CHECK_ARG(dtype == NST_BF16 && (x_dtype == NST_F32 || x_dtype == NST_BF16), "add_layernorm_fwd: bf16 delta / y, f32 or bf16 x");
  if (rows <= 0) return NST_OK;
  const bool ok = d > 0 && d % 8 == 0 && d <= 1024 &&
                  ((((uintptr_t)x | (uintptr_t)delta | (uintptr_t)x_out | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0);
  if (!ok) {
    nst_set_error("add_layernorm_fwd: needs d %% 8 == 0, d <= 1024 and 16-byte aligned operands (d=%d)", d);
    return NST_ERR_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  if (x_dtype == NST_F32) launch_add_fwd<float>(x, delta, (float*)x_out, gamma, beta, y, mean, rstd, rows, d, eps, st);
  else launch_add_fwd<bf16_t>(x, delta, (float*)x_out, gamma, beta, y, mean, rstd, rows, d, eps, st);
  NST_CHECK_LAUNCH("add_layernorm_fwd");
  return NST_OK;
}
// LayerNorm backward whose saved input x has its own dtype (x_dtype = NST_F32 with dtype = NST_BF16: the fp32 residual stream);
// dz / job_out optional as in nst_layernorm_bwd_dropout / nst_layernorm_bwd_deferred
extern "C" int nst_layernorm_bwd_mixed(const void* dy, const void* x, int x_dtype, const float* gamma, const float* mean,
                                       const float* rstd, const void* dres, void* dx, void* dz, float dropout_p, uint64_t seed,
                                       uint64_t stream_id, float* dgamma, float* dbeta, int64_t rows, int d, int dtype,
                                       int accumulate, void* workspace, int64_t workspace_bytes, NstLnFinalizeJob* job_out,
                                       void* stream) {
  return ln_bwd_common(dy, x, nullptr, gamma, mean, rstd, dres, dx, dgamma, dbeta, rows, d, dtype, accumulate, workspace,
                       workspace_bytes, stream, false, dz, dz ? dropout_p : 0.f, seed, stream_id, job_out, x_dtype);
}
// nst_layernorm_relu_bwd without the saved activation: the gate relu'(LN(x)) is recomputed from x, mean, rstd, gamma and beta
// (job_out optional as in nst_layernorm_bwd_deferred)
extern "C" int nst_layernorm_relu_bwd_regate(const void* dy, const void* x, const float* gamma, const float* beta,
                                             const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta,
                                             int64_t rows, int d, int dtype, int accumulate, void* workspace,
                                             int64_t workspace_bytes, NstLnFinalizeJob* job_out, void* stream) {
  NST_CHECK_ARG(beta, "layernorm_relu_bwd_regate: null beta");
  return ln_bwd_common(dy, x, nullptr, gamma, mean, rstd, nullptr, dx, dgamma, dbeta, rows, d, dtype, accumulate, workspace,
                       workspace_bytes, stream, true, nullptr, 0.f, 0, 0, job_out, -1, beta);
}
extern "C" int nst_ln_finalize_multi(const NstLnFinalizeJob* jobs, int njobs, void* stream) {
  NST_CHECK_ARG(njobs >= 0 && njobs <= 16 && (njobs == 0 || jobs), "ln_finalize_multi: 0..16 jobs");
  LnJobs packed;
  int n = 0, dmax = 0;
  bool vec = true;
  for (int i = 0; i < njobs; ++i) {
    if (jobs[i].nblocks <= 0) continue;  // nothing pending for this call
    NST_CHECK_ARG(jobs[i].partial && jobs[i].dgamma && jobs[i].dbeta && jobs[i].d > 0, "ln_finalize_multi: bad job %d", i);
    packed.j[n++] = jobs[i];
    dmax = jobs[i].d > dmax ? jobs[i].d : dmax;
    vec = vec && ln_finalize_vec_ok(jobs[i].partial, jobs[i].d);
  }
  if (n == 0) return NST_OK;
  if (vec) ln_bwd_finalize4_multi_kernel<<<dim3((2 * dmax + 31) / 32, n), 256, 0, (hipStream_t)stream>>>(packed);
  else ln_bwd_finalize_multi_kernel<<<dim3((2 * dmax + 15) / 16, n), 1024, 0, (hipStream_t)stream>>>(packed);
  NST_CHECK_LAUNCH("ln_finalize_multi");
  return NST_OK;
}
