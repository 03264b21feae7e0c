// Row phase of the whole-row kernels (d_model = 256): a workgroup holds BM complete rows of a product as an LDS tile
// [rows][TILE_LD] f32; wave w owns rows [w RPW, +RPW), 32 lanes per row, 8 consecutive columns per lane -- the access pattern of
// the wide LayerNorm kernels (nst_norm.hip): whole 1 KB / 512-byte row segments per half wave, row reductions inside the half
// wave by DPP + lane-row swaps.  Shared by nst_rowgemm.hip (plain products) and nst_ffn.hip (the feed-forward pair), so that the
// stages of the reference's pre-norm wrapper (neurst/layers/common_layers.py:73-85) that need a whole row run behind whichever
// kernel makes the row:
//   ln_fwd:  v = bf16(dropout(tile + bias));  x_out = x + v (f32);  y = LayerNorm(x + v) (bf16), mean, rstd
//   ln_bwd:  g = bf16(tile);  dx = LayerNorm'(g; x, mean, rstd, gamma) + dres (bf16);  dz = dropout-backward copy of dx;
//            per-workgroup partial sums of dgamma / dbeta -> partial[block][2][256] (nst_ln_finalize_multi reduces them)
#pragma once
#include "nst_common.h"

namespace rowphase {

constexpr int RN = 256;              // columns = d_model
constexpr int TILE_LD = 260;         // floats per tile row (1040 bytes: the 16-lane groups of a 16-byte access hit distinct banks)

struct RowEpi {
  const float* bias;     // [256] or null (forward)
  uint32_t drop_thresh;
  float drop_inv_keep;
  uint64_t stream_id;
  const float* x;        // fwd: the residual stream [M, 256] f32; bwd: the saved LayerNorm input (f32)
  float* x_out;          // fwd: x + delta (nullable)
  const float* gamma;
  const float* beta;
  float eps;
  int reserved0;
  bf16_t* y;             // fwd: LayerNorm output; bwd: dx
  float* mean;           // fwd: out; bwd: in
  float* rstd;
  const bf16_t* dres;    // bwd: [M, 256] or null
  bf16_t* dz;            // bwd: [M, 256] or null: dx under the dropout mask (drop_thresh, drop_inv_keep, seed, stream_id)
  float* partial;        // bwd: [gridDim.x][2][256]
};

typedef __attribute__((ext_vector_type(2))) unsigned rg_uint2_t;
__device__ __forceinline__ float rg_swap16_add(float v) {
  const rg_uint2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float rg_swap32_add(float v) {
  const rg_uint2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// sum over the 32 lanes of a half wave (every lane of the half ends up with it)
__device__ __forceinline__ float half_sum(float v) {
  v = dpp_add(v, 0); v = dpp_add(v, 1); v = dpp_add(v, 2); v = dpp_add(v, 3);
  return rg_swap16_add(v);
}
__device__ __forceinline__ void rg_load8_bf16(const bf16_t* __restrict__ p, float (&v)[8]) {
  const uint4 raw = *reinterpret_cast<const uint4*>(p);
  v[0] = __uint_as_float(raw.x << 16); v[1] = __uint_as_float(raw.x & 0xffff0000u);
  v[2] = __uint_as_float(raw.y << 16); v[3] = __uint_as_float(raw.y & 0xffff0000u);
  v[4] = __uint_as_float(raw.z << 16); v[5] = __uint_as_float(raw.z & 0xffff0000u);
  v[6] = __uint_as_float(raw.w << 16); v[7] = __uint_as_float(raw.w & 0xffff0000u);
}
__device__ __forceinline__ void rg_load8_f32(const float* __restrict__ p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void rg_store8_bf16(bf16_t* __restrict__ p, const float (&v)[8]) {
  *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                            pack_bf16x2(v[6], v[7]));
}
__device__ __forceinline__ void rg_store8_f32(float* __restrict__ p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}


// xpre (PRE): the x rows of this lane, requested by the caller before its K loop in the row phase's layout
// (row m0 + wave RPW + 2 p + (lane >> 5), columns 8 (lane & 31) ..), PASSES = RPW / 2 of them
template <int RPW, bool PRE, int NP>
__device__ __forceinline__ void ln_fwd(const float* __restrict__ tile, const RowEpi& e, uint64_t seed, int m0, int M, int wave,
                                       int lane, const float (&xpre)[NP][8]) {
  constexpr int PASSES = RPW / 2;
  static_assert(RPW % 4 == 0 && (!PRE || NP == PASSES), "two rows per pass, two passes per round");
  const int sub = lane >> 5, li = lane & 31, col = li * 8;
  const float inv_d = 1.0f / (float)RN;
    float gm[8], bt[8], bs[8];
    rg_load8_f32(e.gamma + col, gm);
    rg_load8_f32(e.beta + col, bt);
    if (e.bias) rg_load8_f32(e.bias + col, bs);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) bs[j] = 0.f;
    }
#pragma unroll
    for (int p0 = 0; p0 < PASSES; p0 += 2) {
      float v[2][8], xr[2][8];
      int rowg[2];
      bool ok[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int rl = wave * RPW + (p0 + u) * 2 + sub;
        rowg[u] = m0 + rl;
        ok[u] = rowg[u] < M;
        if constexpr (PRE) {
#pragma unroll
          for (int j = 0; j < 8; ++j) xr[u][j] = xpre[p0 + u][j];
        } else {
          rg_load8_f32(e.x + (int64_t)(ok[u] ? rowg[u] : M - 1) * RN + col, xr[u]);
        }
        const float4 t0 = *reinterpret_cast<const float4*>(tile + rl * TILE_LD + col);
        const float4 t1 = *reinterpret_cast<const float4*>(tile + rl * TILE_LD + col + 4);
        v[u][0] = t0.x + bs[0]; v[u][1] = t0.y + bs[1]; v[u][2] = t0.z + bs[2]; v[u][3] = t0.w + bs[3];
        v[u][4] = t1.x + bs[4]; v[u][5] = t1.y + bs[5]; v[u][6] = t1.z + bs[6]; v[u][7] = t1.w + bs[7];
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (e.drop_thresh) {
          float m[8];
          dropout_keep8(seed, e.stream_id, (uint64_t)rowg[u] * (uint64_t)RN + (uint64_t)col, e.drop_thresh, e.drop_inv_keep, m);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[u][j] *= m[j];
        }
        // the sub-layer's contribution is rounded to bf16 before it joins the stream, as the unfused pair does
        // (GEMM epilogue -> bf16 delta -> nst_add_layernorm_fwd): both paths then produce the same sum bit for bit
#pragma unroll
        for (int j = 0; j < 8; ++j) v[u][j] = bf16_to_f32(f32_to_bf16(v[u][j])) + xr[u][j];
        if (e.x_out && ok[u]) rg_store8_f32(e.x_out + (int64_t)rowg[u] * RN + col, v[u]);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[u][j];
        const float mean = half_sum(s) * inv_d;
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float t = v[u][j] - mean; sq += t * t; }
        const float rstd = rsqrtf(half_sum(sq) * inv_d + e.eps);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (v[u][j] - mean) * rstd * gm[j] + bt[j];
        if (ok[u]) {
          rg_store8_bf16(e.y + (int64_t)rowg[u] * RN + col, o);
          if (li == 0) { e.mean[rowg[u]] = mean; e.rstd[rowg[u]] = rstd; }
        }
      }
    }
}

// red: NW * 2 * 256 floats of LDS behind the tile; tid: thread index in the workgroup of NW waves
template <int NW, int RPW, bool PRE, int NP>
__device__ __forceinline__ void ln_bwd(const float* __restrict__ tile, float* __restrict__ red, const RowEpi& e, uint64_t seed,
                                       int m0, int M, int tid, int wave, int lane, const float (&xpre)[NP][8]) {
  constexpr int PASSES = RPW / 2;
  static_assert(RPW % 4 == 0 && (!PRE || NP == PASSES) && (NW == 4 || NW == 8), "two rows per pass, two passes per round");
  const int sub = lane >> 5, li = lane & 31, col = li * 8;
  const float inv_d = 1.0f / (float)RN;
    float gm[8], g_acc[8], b_acc[8];
    rg_load8_f32(e.gamma + col, gm);
#pragma unroll
    for (int j = 0; j < 8; ++j) { g_acc[j] = 0.f; b_acc[j] = 0.f; }
#pragma unroll
    for (int p0 = 0; p0 < PASSES; p0 += 2) {
      float g[2][8], xv[2][8], rv[2][8], mu[2], rs[2];
      int rowg[2];
      bool ok[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int rl = wave * RPW + (p0 + u) * 2 + sub;
        rowg[u] = m0 + rl;
        ok[u] = rowg[u] < M;
        const int rc = ok[u] ? rowg[u] : M - 1;
        if constexpr (PRE) {
#pragma unroll
          for (int j = 0; j < 8; ++j) xv[u][j] = xpre[p0 + u][j];
        } else {
          rg_load8_f32(e.x + (int64_t)rc * RN + col, xv[u]);
        }
        if (e.dres) rg_load8_bf16(e.dres + (int64_t)rc * RN + col, rv[u]);
        else {
#pragma unroll
          for (int j = 0; j < 8; ++j) rv[u][j] = 0.f;
        }
        mu[u] = e.mean[rc];
        rs[u] = e.rstd[rc];
        const float4 t0 = *reinterpret_cast<const float4*>(tile + rl * TILE_LD + col);
        const float4 t1 = *reinterpret_cast<const float4*>(tile + rl * TILE_LD + col + 4);
        const float keep = ok[u] ? 1.f : 0.f;   // rows past the end add nothing to the column sums
        g[u][0] = t0.x * keep; g[u][1] = t0.y * keep; g[u][2] = t0.z * keep; g[u][3] = t0.w * keep;
        g[u][4] = t1.x * keep; g[u][5] = t1.y * keep; g[u][6] = t1.z * keep; g[u][7] = t1.w * keep;
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float s1 = 0.f, s2 = 0.f, xh[8], dxh[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          // the product's output is rounded to bf16 first, as the unfused pair hands it over (GEMM -> bf16 -> LayerNorm backward)
          const float gj = bf16_to_f32(f32_to_bf16(g[u][j]));
          xh[j] = (xv[u][j] - mu[u]) * rs[u];
          dxh[j] = gj * gm[j];
          g_acc[j] += gj * xh[j];
          b_acc[j] += gj;
          s1 += dxh[j];
          s2 += dxh[j] * xh[j];
        }
        const float c1 = half_sum(s1) * inv_d, c2 = half_sum(s2) * inv_d;
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rs[u] * (dxh[j] - c1 - xh[j] * c2) + rv[u][j];
        if (ok[u]) {
          const int64_t off = (int64_t)rowg[u] * RN + col;
          rg_store8_bf16(e.y + off, o);
          if (e.dz) {
            float m[8];
            dropout_keep8(seed, e.stream_id, (uint64_t)off, e.drop_thresh, e.drop_inv_keep, m);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] *= m[j];
            rg_store8_bf16(e.dz + off, o);
          }
        }
      }
    }
    // column sums: the two half waves hold different rows of the same columns; then the four waves through LDS
#pragma unroll
    for (int j = 0; j < 8; ++j) { g_acc[j] = rg_swap32_add(g_acc[j]); b_acc[j] = rg_swap32_add(b_acc[j]); }
        if (sub == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        red[(wave * 2 + 0) * RN + col + j] = g_acc[j];
        red[(wave * 2 + 1) * RN + col + j] = b_acc[j];
      }
    }
    __syncthreads();
    if (tid < RN) {
      const int c = tid;   // one thread per column
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        float t = (red[(0 * 2 + pass) * RN + c] + red[(1 * 2 + pass) * RN + c]) + (red[(2 * 2 + pass) * RN + c] + red[(3 * 2 + pass) * RN + c]);
        if constexpr (NW == 8)
          t += (red[(4 * 2 + pass) * RN + c] + red[(5 * 2 + pass) * RN + c]) + (red[(6 * 2 + pass) * RN + c] + red[(7 * 2 + pass) * RN + c]);
        e.partial[((int64_t)blockIdx.x * 2 + pass) * RN + c] = t;
      }
    }
}

}  // namespace rowphase
