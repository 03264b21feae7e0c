// HBM-bound kernels of the path: target embedding (+scale, +sinusoid, +dropout) and its scatter-add
// gradient, elementwise scale/posenc/dropout, fused label-smoothed cross entropy, column sums
// (bias gradients), fused Adam and dtype casts.  fp32 math everywhere; bf16 only on the wire.
#include "nst_common.h"

namespace {

template <typename T>
__device__ __forceinline__ void load4(const T* p, float (&v)[4]) {
  if (sizeof(T) == 2) {
    uint2 raw = *reinterpret_cast<const uint2*>(p);
    v[0] = bf16_to_f32((bf16_t)(raw.x & 0xffff)); v[1] = bf16_to_f32((bf16_t)(raw.x >> 16));
    v[2] = bf16_to_f32((bf16_t)(raw.y & 0xffff)); v[3] = bf16_to_f32((bf16_t)(raw.y >> 16));
  } else {
    float4 raw = *reinterpret_cast<const float4*>(p);
    v[0] = raw.x; v[1] = raw.y; v[2] = raw.z; v[3] = raw.w;
  }
}
template <typename T>
__device__ __forceinline__ void store4(T* p, const float (&v)[4]) {
  if (sizeof(T) == 2) {
    uint2 raw;
    raw.x = pack_bf16x2(v[0], v[1]);
    raw.y = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = raw;
  } else {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
}
template <typename T>
bool aligned4(const void* p) { return (((uintptr_t)p) & (sizeof(T) == 2 ? 7 : 15)) == 0; }

__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += sh[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* sh) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float t = sh[0];
  for (int i = 1; i < nw; ++i) t = fmaxf(t, sh[i]);
  return t;
}

// ---------------------------------------------------------------------------------------------
// embedding: out[row,:] = dropout(table[ids[row],:]*scale + posenc[row % L,:])
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void embedding_fwd_kernel(const T* __restrict__ table, const int64_t* __restrict__ ids,
                                     const float* __restrict__ posenc, T* __restrict__ out, int64_t rows, int L, int d,
                                     int V, float scale, uint32_t thresh, float inv_keep, uint64_t seed, uint64_t sid,
                                     const uint64_t* __restrict__ seed_dev) {
  if (thresh) seed = seed_with_offset(seed, seed_dev);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int64_t row = (int64_t)blockIdx.x * nw + wave; row < rows; row += (int64_t)gridDim.x * nw) {
    int64_t id = ids[row];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    const int pos = (int)(row % L);
    for (int e = lane; e < d; e += 64) {
      float v = to_f32<T>(table[id * d + e]) * scale;
      if (posenc) v += posenc[(int64_t)pos * d + e];
      if (thresh) v *= dropout_keep_scale(seed, sid, (uint64_t)row * d + e, thresh, inv_keep);
      out[row * d + e] = from_f32<T>(v);
    }
  }
}

template <typename T>
__global__ void embedding_bwd_kernel(const T* __restrict__ dout, const int64_t* __restrict__ ids, float* __restrict__ dtable,
                                     int64_t rows, int d, int V, float scale, uint32_t thresh, float inv_keep,
                                     uint64_t seed, uint64_t sid, const uint64_t* __restrict__ seed_dev) {
  if (thresh) seed = seed_with_offset(seed, seed_dev);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int64_t row = (int64_t)blockIdx.x * nw + wave; row < rows; row += (int64_t)gridDim.x * nw) {
    int64_t id = ids[row];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    for (int e = lane; e < d; e += 64) {
      float g = to_f32<T>(dout[row * d + e]) * scale;
      if (thresh) g *= dropout_keep_scale(seed, sid, (uint64_t)row * d + e, thresh, inv_keep);
      atomicAdd(dtable + id * d + e, g);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// y = dropout(x*scale + posenc[row % period, :])   /   dx = dy*mask*scale
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ void scale_posenc_dropout_kernel(const T* __restrict__ x, const float* __restrict__ posenc, T* __restrict__ y,
                                            int64_t n, int d, int period, float scale, uint32_t thresh, float inv_keep,
                                            uint64_t seed, uint64_t sid, const uint64_t* __restrict__ seed_dev) {
  if (thresh) seed = seed_with_offset(seed, seed_dev);
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC; i < n;
       i += (int64_t)gridDim.x * blockDim.x * VEC) {
    float v[4];
    if (VEC == 4) load4<T>(x + i, v); else v[0] = to_f32<T>(x[i]);
    float m[4] = {1.f, 1.f, 1.f, 1.f};
    if (thresh) {
      if (VEC == 4) dropout_keep4(seed, sid, (uint64_t)i, thresh, inv_keep, m);
      else m[0] = dropout_keep_scale(seed, sid, (uint64_t)i, thresh, inv_keep);
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int64_t idx = i + j;
      float t = v[j] * scale;
      if (posenc) { const int64_t row = idx / d; t += posenc[(row % period) * d + (idx - row * d)]; }
      v[j] = t * m[j];
    }
    if (VEC == 4) store4<T>(y + i, v); else y[i] = from_f32<T>(v[0]);
  }
}

// ---------------------------------------------------------------------------------------------
// label smoothed cross entropy
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) ls_xent_fwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ labels,
                                                         const float* __restrict__ weights, float* __restrict__ xent,
                                                         float* __restrict__ lse_out, int V, int64_t ldl, float conf,
                                                         float low, float norm_const) {
  __shared__ float sh[8];
  const int64_t row = blockIdx.x;
  const T* x = logits + row * ldl;
  float mx = -INFINITY;
  for (int v = threadIdx.x; v < V; v += blockDim.x) mx = fmaxf(mx, to_f32<T>(x[v]));
  mx = block_max(mx, sh);
  float se = 0.f, sx = 0.f;
  for (int v = threadIdx.x; v < V; v += blockDim.x) {
    const float t = to_f32<T>(x[v]);
    se += __expf(t - mx);
    sx += t;
  }
  se = block_sum(se, sh);
  sx = block_sum(sx, sh);
  if (threadIdx.x == 0) {
    const float lse = mx + logf(se);
    int64_t lab = labels[row];
    lab = lab < 0 ? 0 : (lab >= V ? V - 1 : lab);
    const float xl = to_f32<T>(x[lab]);
    // -sum_v soft_v*(x_v - lse), soft = low everywhere, conf at the label
    float loss = -((conf - low) * (xl - lse) + low * (sx - (float)V * lse)) - norm_const;
    xent[row] = loss * weights[row];
    lse_out[row] = lse;
  }
}

// 16-byte form (aligned rows, V a multiple of the chunk, V <= NCH * 256 chunks): a row is read ONCE -- every thread keeps its
// NCH chunks (8 bf16 / 4 f32 each) in registers between the maximum and the two sums.  The scalar kernel above reads 2 bytes
// per lane and walks the row twice: 115 us for the 9600 x 8008 bf16 logits of the benchmark (154 MB, 1.3 TB/s).
template <typename T, int NCH>
__global__ void __launch_bounds__(256) ls_xent_fwd_vec_kernel(const T* __restrict__ logits, const int64_t* __restrict__ labels,
                                                             const float* __restrict__ weights, float* __restrict__ xent,
                                                             float* __restrict__ lse_out, int V, int64_t ldl, float conf,
                                                             float low, float norm_const) {
  constexpr int CPT = 16 / (int)sizeof(T);
  __shared__ float sh[8];
  const int64_t row = blockIdx.x;
  const T* x = logits + row * ldl;
  const int nchunks = V / CPT;
  float v[NCH][CPT];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int c = threadIdx.x + j * 256;
    if (c < nchunks) {
      const uint4 raw = *reinterpret_cast<const uint4*>(x + (int64_t)c * CPT);
      T tmp[CPT];
      memcpy(tmp, &raw, 16);
#pragma unroll
      for (int e = 0; e < CPT; ++e) { v[j][e] = to_f32<T>(tmp[e]); mx = fmaxf(mx, v[j][e]); }
    } else {
#pragma unroll
      for (int e = 0; e < CPT; ++e) v[j][e] = -INFINITY;
    }
  }
  mx = block_max(mx, sh);
  float se = 0.f, sx = 0.f;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const bool in = threadIdx.x + j * 256 < nchunks;
#pragma unroll
    for (int e = 0; e < CPT; ++e) {
      se += __expf(v[j][e] - mx);          // exp(-inf) = 0 for the chunks past the row
      sx += in ? v[j][e] : 0.f;
    }
  }
  se = block_sum(se, sh);
  sx = block_sum(sx, sh);
  if (threadIdx.x == 0) {
    const float lse = mx + logf(se);
    int64_t lab = labels[row];
    lab = lab < 0 ? 0 : (lab >= V ? V - 1 : lab);
    const float xl = to_f32<T>(x[lab]);
    float loss = -((conf - low) * (xl - lse) + low * (sx - (float)V * lse)) - norm_const;
    xent[row] = loss * weights[row];
    lse_out[row] = lse;
  }
}

template <typename T>
__global__ void __launch_bounds__(256) ls_xent_bwd_vec_kernel(const T* __restrict__ logits, const int64_t* __restrict__ labels,
                                                             const float* __restrict__ weights, const float* __restrict__ lse_in,
                                                             T* __restrict__ dlogits, int V, int64_t ldl, float conf, float low,
                                                             float gscale, const float* __restrict__ gscale_dev) {
  constexpr int CPT = 16 / (int)sizeof(T);
  const int64_t row = blockIdx.x;
  const T* x = logits + row * ldl;
  T* dx = dlogits + row * ldl;
  const float lse = lse_in[row];
  const float w = weights[row] * gscale * (gscale_dev ? gscale_dev[0] : 1.f);
  int64_t lab = labels[row];
  lab = lab < 0 ? 0 : (lab >= V ? V - 1 : lab);
  const int nchunks = V / CPT;
  for (int c = threadIdx.x; c < nchunks; c += 256) {
    const uint4 raw = *reinterpret_cast<const uint4*>(x + (int64_t)c * CPT);
    T tmp[CPT], out[CPT];
    memcpy(tmp, &raw, 16);
#pragma unroll
    for (int e = 0; e < CPT; ++e) {
      const float p = __expf(to_f32<T>(tmp[e]) - lse);
      const float soft = (c * CPT + e) == lab ? conf : low;
      out[e] = from_f32<T>((p - soft) * w);
    }
    uint4 o;
    memcpy(&o, out, 16);
    *reinterpret_cast<uint4*>(dx + (int64_t)c * CPT) = o;
  }
}

template <typename T>
__global__ void __launch_bounds__(256) ls_xent_bwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ labels,
                                                         const float* __restrict__ weights, const float* __restrict__ lse_in,
                                                         T* __restrict__ dlogits, int V, int64_t ldl, float conf, float low,
                                                         float gscale, const float* __restrict__ gscale_dev) {
  const int64_t row = blockIdx.x;
  const T* x = logits + row * ldl;
  T* dx = dlogits + row * ldl;
  const float lse = lse_in[row];
  const float w = weights[row] * gscale * (gscale_dev ? gscale_dev[0] : 1.f);
  int64_t lab = labels[row];
  lab = lab < 0 ? 0 : (lab >= V ? V - 1 : lab);
  for (int v = threadIdx.x; v < V; v += blockDim.x) {
    const float p = __expf(to_f32<T>(x[v]) - lse);
    const float soft = v == lab ? conf : low;
    dx[v] = from_f32<T>((p - soft) * w);
  }
}

// ---------------------------------------------------------------------------------------------
// column sums (bias gradients): out[c] += sum_r x[r, c]
// ---------------------------------------------------------------------------------------------
// Each thread owns CPT consecutive columns (16 bytes of a row) so a wave reads whole 128-byte lines; the block's
// 256 threads cover (256*CPT/ncols_tile) rows per iteration; rows are strided over gridDim.y strips.
template <typename T>
__global__ void __launch_bounds__(256) colsum_kernel(const T* __restrict__ x, float* __restrict__ out, int64_t rows, int n,
                                                    int64_t ldx, int64_t rows_per_block, int vec, float* __restrict__ partial) {
  constexpr int CPT = 16 / (int)sizeof(T);  // columns per thread
  constexpr int TPR = 32;                   // threads per row -> a block tile is 8 rows x (32*CPT) columns
  __shared__ float sh[8][TPR * CPT + 1];
  const int tr = threadIdx.x / TPR, tc = threadIdx.x % TPR;
  const int col0 = blockIdx.x * (TPR * CPT) + tc * CPT;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  float acc[CPT];
#pragma unroll
  for (int j = 0; j < CPT; ++j) acc[j] = 0.f;
  if (col0 < n) {
    for (int64_t r = r0 + tr; r < r1; r += 8) {
      const T* p = x + r * ldx + col0;
      if (vec) {
        uint4 raw = *reinterpret_cast<const uint4*>(p);
        T tmp[CPT];
        memcpy(tmp, &raw, 16);
#pragma unroll
        for (int j = 0; j < CPT; ++j) acc[j] += to_f32<T>(tmp[j]);
      } else {
#pragma unroll
        for (int j = 0; j < CPT; ++j)
          if (col0 + j < n) acc[j] += to_f32<T>(p[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < CPT; ++j) sh[tr][tc * CPT + j] = acc[j];
  __syncthreads();
  for (int c = threadIdx.x; c < TPR * CPT; c += 256) {
    const int col = blockIdx.x * (TPR * CPT) + c;
    if (col < n) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) t += sh[q][c];
      if (partial) partial[(int64_t)blockIdx.y * n + col] = t;
      else atomicAdd(out + col, t);
    }
  }
}

__global__ void __launch_bounds__(256) colsum_finalize_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                             int strips, int n, int accumulate) {
  __shared__ float sh[16][17];
  const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int col = blockIdx.x * 16 + c;
  float t = 0.f;
  if (col < n)
    for (int s = rg; s < strips; s += 16) t += partial[(int64_t)s * n + col];
  sh[rg][c] = t;
  __syncthreads();
  if (rg == 0 && col < n) {
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc += sh[q][c];
    out[col] = accumulate ? out[col] + acc : acc;
  }
}

// ---------------------------------------------------------------------------------------------
// Keras Adam over a flat buffer (+ optional bf16 shadow of the updated parameter)
// ---------------------------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, const float* __restrict__ g,
                            bf16_t* __restrict__ shadow, int64_t n, float lr_t, const float* __restrict__ lr_t_dev, float b1,
                            float b2, float eps, float gscale, const float* __restrict__ ls_state) {
  if (lr_t_dev) lr_t = *lr_t_dev;   // graph replay: the step size of THIS replay lives in device memory
  if (ls_state) {                   // dynamic loss scale: skip the step on non-finite gradients, else unscale
    if (ls_state[2] == 0.f) return;
    gscale /= ls_state[3];
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    const float pi = p[i] - lr_t * mi / (sqrtf(vi) + eps);
    m[i] = mi; v[i] = vi; p[i] = pi;
    if (shadow) shadow[i] = f32_to_bf16(pi);
  }
}

// dynamic loss scale (neurst/training/revised_dynamic_loss_scale.py:48-107): state = {scale, good steps, finite flag of the
// gradients just checked, scale those gradients carry}
__global__ void __launch_bounds__(256) nonfinite_count_kernel(const float* __restrict__ g, int64_t n, unsigned int* __restrict__ count) {
  unsigned int bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t bits = __float_as_uint(g[i]);
    bad += ((bits & 0x7f800000u) == 0x7f800000u) ? 1u : 0u;   // exponent all ones: inf or nan
  }
  bad = __builtin_amdgcn_ballot_w64(bad != 0) != 0 ? 1u : 0u;
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(count, 1u);
}
__global__ void loss_scale_update_kernel(float* __restrict__ st, unsigned int* __restrict__ count, float growth_steps, float multiplier) {
  const bool finite = *count == 0u;
  *count = 0u;
  st[3] = st[0];
  st[2] = finite ? 1.f : 0.f;
  if (finite) {
    if (st[1] + 1.f >= growth_steps) {
      const float grown = st[0] * multiplier;
      if ((__float_as_uint(grown) & 0x7f800000u) != 0x7f800000u) st[0] = grown;   // _assign_if_finite
      st[1] = 0.f;
    } else {
      st[1] += 1.f;
    }
  } else {
    st[0] = fmaxf(st[0] / multiplier, 1.f);
    st[1] = 0.f;
  }
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = f32_to_bf16(in[i]);
}
__global__ void cast_bf16_f32_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = bf16_to_f32(in[i]);
}

int grid_for(int64_t work_items, int per_block, int cap) {
  int64_t b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

}  // namespace

extern "C" int nst_embedding_fwd(const void* table, const int64_t* ids, const float* posenc, void* out, int64_t rows,
                                 int L, int d, int V, float emb_scale, float dropout_p, uint64_t seed, uint64_t stream_id,
                                 int dtype, void* stream) {
  NST_CHECK_ARG(table && ids && out, "embedding_fwd: null pointer");
  NST_CHECK_ARG(L > 0 && d > 0 && V > 0, "embedding_fwd: bad dims L=%d d=%d V=%d", L, d, V);
  NST_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "embedding_fwd: dropout_p=%f", dropout_p);
  if (rows <= 0) return NST_OK;
  uint32_t th; float ik;
  nst_dropout_params16(dropout_p, &th, &ik);
  hipStream_t st = (hipStream_t)stream;
  const int g = grid_for(rows, 4, 4096);
  if (dtype == NST_F32)
    embedding_fwd_kernel<float><<<g, 256, 0, st>>>((const float*)table, ids, posenc, (float*)out, rows, L, d, V, emb_scale, th, ik, seed, stream_id, nst_seed_offset_devptr());
  else if (dtype == NST_BF16)
    embedding_fwd_kernel<bf16_t><<<g, 256, 0, st>>>((const bf16_t*)table, ids, posenc, (bf16_t*)out, rows, L, d, V, emb_scale, th, ik, seed, stream_id, nst_seed_offset_devptr());
  else { nst_set_error("embedding_fwd: bad dtype %d", dtype); return NST_ERR_INVALID_ARG; }
  NST_CHECK_LAUNCH("embedding_fwd");
  return NST_OK;
}

extern "C" int nst_embedding_bwd(const void* dout, const int64_t* ids, float* dtable, int64_t rows, int d, int V,
                                 float emb_scale, float dropout_p, uint64_t seed, uint64_t stream_id, int dtype,
                                 void* stream) {
  NST_CHECK_ARG(dout && ids && dtable, "embedding_bwd: null pointer");
  NST_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "embedding_bwd: dropout_p=%f", dropout_p);
  if (rows <= 0) return NST_OK;
  uint32_t th; float ik;
  nst_dropout_params16(dropout_p, &th, &ik);
  hipStream_t st = (hipStream_t)stream;
  const int g = grid_for(rows, 4, 4096);
  if (dtype == NST_F32)
    embedding_bwd_kernel<float><<<g, 256, 0, st>>>((const float*)dout, ids, dtable, rows, d, V, emb_scale, th, ik, seed, stream_id, nst_seed_offset_devptr());
  else if (dtype == NST_BF16)
    embedding_bwd_kernel<bf16_t><<<g, 256, 0, st>>>((const bf16_t*)dout, ids, dtable, rows, d, V, emb_scale, th, ik, seed, stream_id, nst_seed_offset_devptr());
  else { nst_set_error("embedding_bwd: bad dtype %d", dtype); return NST_ERR_INVALID_ARG; }
  NST_CHECK_LAUNCH("embedding_bwd");
  return NST_OK;
}

template <typename T>
static void launch_spd(const void* x, const float* posenc, void* y, int64_t n, int d, int period, float scale, uint32_t th,
                       float ik, uint64_t seed, uint64_t sid, hipStream_t st) {
  if (n % 4 == 0 && aligned4<T>(x) && aligned4<T>(y) && (d % 4 == 0))
    scale_posenc_dropout_kernel<T, 4><<<grid_for(n / 4, 256, 8192), 256, 0, st>>>((const T*)x, posenc, (T*)y, n, d, period, scale, th, ik, seed, sid, nst_seed_offset_devptr());
  else
    scale_posenc_dropout_kernel<T, 1><<<grid_for(n, 256, 8192), 256, 0, st>>>((const T*)x, posenc, (T*)y, n, d, period, scale, th, ik, seed, sid, nst_seed_offset_devptr());
}

extern "C" int nst_scale_posenc_dropout_fwd(const void* x, const float* posenc, void* y, int64_t rows, int d, int period,
                                            float scale, float dropout_p, uint64_t seed, uint64_t stream_id, int dtype,
                                            void* stream) {
  NST_CHECK_ARG(x && y, "scale_posenc_dropout_fwd: null pointer");
  NST_CHECK_ARG(d > 0 && (posenc == nullptr || period > 0), "scale_posenc_dropout_fwd: bad dims");
  NST_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "scale_posenc_dropout_fwd: dropout_p=%f", dropout_p);
  if (rows <= 0) return NST_OK;
  uint32_t th; float ik;
  nst_dropout_params16(dropout_p, &th, &ik);
  hipStream_t st = (hipStream_t)stream;
  if (period <= 0) period = 1;
  if (dtype == NST_F32) launch_spd<float>(x, posenc, y, rows * d, d, period, scale, th, ik, seed, stream_id, st);
  else if (dtype == NST_BF16) launch_spd<bf16_t>(x, posenc, y, rows * d, d, period, scale, th, ik, seed, stream_id, st);
  else { nst_set_error("scale_posenc_dropout_fwd: bad dtype %d", dtype); return NST_ERR_INVALID_ARG; }
  NST_CHECK_LAUNCH("scale_posenc_dropout_fwd");
  return NST_OK;
}

extern "C" int nst_scale_dropout_bwd(const void* dy, void* dx, int64_t n, float scale, float dropout_p, uint64_t seed,
                                     uint64_t stream_id, int dtype, void* stream) {
  NST_CHECK_ARG(dy && dx, "scale_dropout_bwd: null pointer");
  NST_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "scale_dropout_bwd: dropout_p=%f", dropout_p);
  if (n <= 0) return NST_OK;
  uint32_t th; float ik;
  nst_dropout_params16(dropout_p, &th, &ik);
  hipStream_t st = (hipStream_t)stream;
  // d is only used for the posenc lookup (absent here): pass d=4 so the vector path stays eligible
  if (dtype == NST_F32) launch_spd<float>(dy, nullptr, dx, n, 4, 1, scale, th, ik, seed, stream_id, st);
  else if (dtype == NST_BF16) launch_spd<bf16_t>(dy, nullptr, dx, n, 4, 1, scale, th, ik, seed, stream_id, st);
  else { nst_set_error("scale_dropout_bwd: bad dtype %d", dtype); return NST_ERR_INVALID_ARG; }
  NST_CHECK_LAUNCH("scale_dropout_bwd");
  return NST_OK;
}

static int xent_consts(int V, float ls, float* conf, float* low, float* norm) {
  *conf = 1.0f - ls;
  *low = V > 1 ? ls / (float)(V - 1) : 0.f;
  *norm = 0.f;
  if (ls != 0.f)  // label_smoothed_cross_entropy.py:131-136
    *norm = -(*conf * logf(*conf) + (float)(V - 1) * *low * logf(*low + 1e-20f));
  return 0;
}

static bool xent_vec() { return true; }   // (the scalar kernels serve unaligned / odd vocabularies)

extern "C" int nst_ls_xent_fwd(const void* logits, const int64_t* labels, const float* weights, float* xent, float* lse,
                               int64_t rows, int V, int64_t ldl, float label_smoothing, int dtype, void* stream) {
  NST_CHECK_ARG(logits && labels && weights && xent && lse, "ls_xent_fwd: null pointer");
  NST_CHECK_ARG(V > 0 && ldl >= V, "ls_xent_fwd: bad V=%d ldl=%lld", V, (long long)ldl);
  NST_CHECK_ARG(label_smoothing >= 0.f && label_smoothing < 1.f, "ls_xent_fwd: label_smoothing=%f", label_smoothing);
  if (rows <= 0) return NST_OK;
  float conf, low, norm;
  xent_consts(V, label_smoothing, &conf, &low, &norm);
  hipStream_t st = (hipStream_t)stream;
  const int esz = dtype == NST_F32 ? 4 : 2, cpt = 16 / esz;
  const bool vec = xent_vec() && nst_aligned16(logits) && (ldl * esz) % 16 == 0 && V % cpt == 0;
  const int nchunks = V / cpt;
  if (vec && dtype == NST_BF16 && nchunks <= 4 * 256)
    ls_xent_fwd_vec_kernel<bf16_t, 4><<<(unsigned)rows, 256, 0, st>>>((const bf16_t*)logits, labels, weights, xent, lse, V, ldl, conf, low, norm);
  else if (vec && dtype == NST_BF16 && nchunks <= 8 * 256)
    ls_xent_fwd_vec_kernel<bf16_t, 8><<<(unsigned)rows, 256, 0, st>>>((const bf16_t*)logits, labels, weights, xent, lse, V, ldl, conf, low, norm);
  else if (vec && dtype == NST_F32 && nchunks <= 8 * 256)
    ls_xent_fwd_vec_kernel<float, 8><<<(unsigned)rows, 256, 0, st>>>((const float*)logits, labels, weights, xent, lse, V, ldl, conf, low, norm);
  else if (dtype == NST_F32)
    ls_xent_fwd_kernel<float><<<(unsigned)rows, 256, 0, st>>>((const float*)logits, labels, weights, xent, lse, V, ldl, conf, low, norm);
  else if (dtype == NST_BF16)
    ls_xent_fwd_kernel<bf16_t><<<(unsigned)rows, 256, 0, st>>>((const bf16_t*)logits, labels, weights, xent, lse, V, ldl, conf, low, norm);
  else { nst_set_error("ls_xent_fwd: bad dtype %d", dtype); return NST_ERR_INVALID_ARG; }
  NST_CHECK_LAUNCH("ls_xent_fwd");
  return NST_OK;
}

extern "C" int nst_ls_xent_bwd(const void* logits, const int64_t* labels, const float* weights, const float* lse,
                               void* dlogits, int64_t rows, int V, int64_t ldl, float label_smoothing, float gscale,
                               const float* gscale_dev, int dtype, void* stream) {
  NST_CHECK_ARG(logits && labels && weights && lse && dlogits, "ls_xent_bwd: null pointer");
  NST_CHECK_ARG(V > 0 && ldl >= V, "ls_xent_bwd: bad V=%d ldl=%lld", V, (long long)ldl);
  if (rows <= 0) return NST_OK;
  float conf, low, norm;
  xent_consts(V, label_smoothing, &conf, &low, &norm);
  hipStream_t st = (hipStream_t)stream;
  const int esz = dtype == NST_F32 ? 4 : 2, cpt = 16 / esz;
  const bool vec = xent_vec() && nst_aligned16(logits) && nst_aligned16(dlogits) && (ldl * esz) % 16 == 0 && V % cpt == 0;
  if (vec && dtype == NST_F32)
    ls_xent_bwd_vec_kernel<float><<<(unsigned)rows, 256, 0, st>>>((const float*)logits, labels, weights, lse, (float*)dlogits, V, ldl, conf, low, gscale, gscale_dev);
  else if (vec && dtype == NST_BF16)
    ls_xent_bwd_vec_kernel<bf16_t><<<(unsigned)rows, 256, 0, st>>>((const bf16_t*)logits, labels, weights, lse, (bf16_t*)dlogits, V, ldl, conf, low, gscale, gscale_dev);
  else if (dtype == NST_F32)
    ls_xent_bwd_kernel<float><<<(unsigned)rows, 256, 0, st>>>((const float*)logits, labels, weights, lse, (float*)dlogits, V, ldl, conf, low, gscale, gscale_dev);
  else if (dtype == NST_BF16)
    ls_xent_bwd_kernel<bf16_t><<<(unsigned)rows, 256, 0, st>>>((const bf16_t*)logits, labels, weights, lse, (bf16_t*)dlogits, V, ldl, conf, low, gscale, gscale_dev);
  else { nst_set_error("ls_xent_bwd: bad dtype %d", dtype); return NST_ERR_INVALID_ARG; }
  NST_CHECK_LAUNCH("ls_xent_bwd");
  return NST_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Sequence masks from lengths in ONE launch (model_utils.py:44-75 sequence_mask / 1 - sequence_mask, layer_utils.py:19-32
// padding * FLOAT_MIN, speech_transformer.py:179-189 the length after the two stride-2 convolutions): the host path built them
// from arange / compare / cast / arithmetic launches (11 of the ~30 torch-native launches of a step).
//   out[b][t] = t < len'(b) ? on_token : on_padding,   len' = `halvings` times ceil(len / stride)
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) seq_mask_kernel(const int64_t* __restrict__ lengths, float* __restrict__ out, int B, int T,
                                                      int halvings, int stride, float on_token, float on_padding) {
  const int64_t n = (int64_t)B * T;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / T), t = (int)(i - (int64_t)b * T);
    int64_t len = lengths[b];
    for (int h = 0; h < halvings; ++h) len = (len + stride - 1) / stride;
    out[i] = t < len ? on_token : on_padding;
  }
}

extern "C" int nst_seq_mask(const int64_t* lengths, float* out, int B, int T, int halvings, int stride, float on_token,
                            float on_padding, void* stream) {
  NST_CHECK_ARG(lengths && out && B >= 0 && T >= 0 && halvings >= 0 && stride >= 1, "seq_mask: bad arguments");
  if (B == 0 || T == 0) return NST_OK;
  const int64_t n = (int64_t)B * T;
  const int blocks = (int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
  seq_mask_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(lengths, out, B, T, halvings, stride, on_token, on_padding);
  NST_CHECK_LAUNCH("seq_mask");
  return NST_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Reductions of the criterion in ONE launch (label_smoothed_cross_entropy.py:46-53, 141-157): per sample nll_sum[b] = sum_t
// xent[b][t] and n_tokens[b] = sum_t weights[b][t], the batch loss sum(nll_sum) / sum(n_tokens) and 1 / sum(n_tokens) (the
// factor the backward kernel reads from the device) -- eight torch reductions / element-wise launches before.  One workgroup;
// sums in a fixed order (deterministic).
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) xent_reduce_kernel(const float* __restrict__ xent, const float* __restrict__ weights, int B, int L,
                                                         float* __restrict__ nll_sum, float* __restrict__ n_tokens,
                                                         float* __restrict__ loss, float* __restrict__ inv_tokens) {
  __shared__ float s_nll[256], s_tok[256];
  float a_nll = 0.f, a_tok = 0.f;
  // eight lanes per row, 32 rows per pass: a lane sums positions j, j + 8, ... and the eight partial sums meet in a fixed
  // xor tree (one thread per row walked its L positions as one dependent chain of strided loads: 22 us for 128 x 75)
  const int j = threadIdx.x & 7;
  for (int b0 = 0; b0 < B; b0 += 32) {
    const int b = b0 + ((int)threadIdx.x >> 3);
    float n = 0.f, w = 0.f;
    if (b < B)
      for (int t = j; t < L; t += 8) { n += xent[(int64_t)b * L + t]; w += weights[(int64_t)b * L + t]; }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { n += __shfl_xor(n, o, 64); w += __shfl_xor(w, o, 64); }
    if (j == 0 && b < B) {
      nll_sum[b] = n;
      n_tokens[b] = w;
      a_nll += n;
      a_tok += w;
    }
  }
  s_nll[threadIdx.x] = a_nll;
  s_tok[threadIdx.x] = a_tok;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { s_nll[threadIdx.x] += s_nll[threadIdx.x + o]; s_tok[threadIdx.x] += s_tok[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    loss[0] = s_nll[0] / s_tok[0];
    inv_tokens[0] = 1.0f / s_tok[0];
  }
}

extern "C" int nst_xent_reduce(const float* xent, const float* weights, int B, int L, float* nll_sum, float* n_tokens, float* loss,
                               float* inv_tokens, void* stream) {
  NST_CHECK_ARG(xent && weights && nll_sum && n_tokens && loss && inv_tokens && B > 0 && L > 0, "xent_reduce: bad arguments");
  xent_reduce_kernel<<<1, 256, 0, (hipStream_t)stream>>>(xent, weights, B, L, nll_sum, n_tokens, loss, inv_tokens);
  NST_CHECK_LAUNCH("xent_reduce");
  return NST_OK;
}

extern "C" int nst_colsum(const void* x, float* out, int64_t rows, int n, int64_t ldx, int dtype, int accumulate,
                          void* workspace, int64_t workspace_bytes, void* stream) {
  NST_CHECK_ARG(x && out, "colsum: null pointer");
  NST_CHECK_ARG(n > 0 && ldx >= n, "colsum: bad n=%d ldx=%lld", n, (long long)ldx);
  hipStream_t st = (hipStream_t)stream;
  float* partial = (workspace && workspace_bytes >= (int64_t)256 * n * 4 && ((((uintptr_t)workspace) & 3) == 0))
                       ? (float*)workspace : nullptr;
  if (!accumulate && (!partial || rows <= 0)) NST_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(float) * n, st));
  if (rows <= 0) return NST_OK;
  const int esz = dtype == NST_BF16 ? 2 : 4;
  const int cpt = 16 / esz, tile_cols = 32 * cpt;
  int64_t strips = (rows + 63) / 64;
  if (strips > 256) strips = 256;
  const int64_t rpb = (rows + strips - 1) / strips;
  dim3 grid((n + tile_cols - 1) / tile_cols, (unsigned)((rows + rpb - 1) / rpb));
  const int vec = nst_aligned16(x) && ((ldx * esz) % 16 == 0) && (n % cpt == 0);
  if (dtype == NST_F32) colsum_kernel<float><<<grid, 256, 0, st>>>((const float*)x, out, rows, n, ldx, rpb, vec, partial);
  else if (dtype == NST_BF16) colsum_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)x, out, rows, n, ldx, rpb, vec, partial);
  else { nst_set_error("colsum: bad dtype %d", dtype); return NST_ERR_INVALID_ARG; }
  NST_CHECK_LAUNCH("colsum");
  if (partial) {
    colsum_finalize_kernel<<<(n + 15) / 16, 256, 0, st>>>(partial, out, (int)grid.y, n, accumulate);
    NST_CHECK_LAUNCH("colsum(finalize)");
  }
  return NST_OK;
}

extern "C" int nst_adam_update(float* p, float* m, float* v, const float* g, uint16_t* shadow_bf16, int64_t n, float lr_t,
                               float beta1, float beta2, float eps, float grad_scale, void* stream) {
  NST_CHECK_ARG(p && m && v && g, "adam_update: null pointer");
  if (n <= 0) return NST_OK;
  adam_kernel<<<grid_for(n, 256, 8192), 256, 0, (hipStream_t)stream>>>(p, m, v, g, shadow_bf16, n, lr_t, nullptr, beta1, beta2, eps, grad_scale, nullptr);
  NST_CHECK_LAUNCH("adam_update");
  return NST_OK;
}

extern "C" int nst_adam_update_dev(float* p, float* m, float* v, const float* g, uint16_t* shadow_bf16, int64_t n, float lr_t,
                                   const float* lr_t_dev, float beta1, float beta2, float eps, float grad_scale,
                                   const float* loss_scale_state, void* stream) {
  NST_CHECK_ARG(p && m && v && g, "adam_update_dev: null pointer");
  if (n <= 0) return NST_OK;
  adam_kernel<<<grid_for(n, 256, 8192), 256, 0, (hipStream_t)stream>>>(p, m, v, g, shadow_bf16, n, lr_t, lr_t_dev, beta1, beta2, eps, grad_scale, loss_scale_state);
  NST_CHECK_LAUNCH("adam_update_dev");
  return NST_OK;
}

extern "C" int nst_loss_scale_update(const float* grad, int64_t n, float* state, float growth_steps, float multiplier,
                                     void* workspace, int64_t workspace_bytes, void* stream) {
  NST_CHECK_ARG(grad && state && workspace && workspace_bytes >= 4, "loss_scale_update: null pointer / workspace of >= 4 zeroed bytes");
  NST_CHECK_ARG(growth_steps >= 1.f && multiplier > 1.f, "loss_scale_update: growth_steps=%f multiplier=%f", growth_steps, multiplier);
  unsigned int* count = (unsigned int*)workspace;   // must be zero on entry; the update kernel leaves it zero
  if (n > 0) nonfinite_count_kernel<<<grid_for(n, 256 * 8, 2048), 256, 0, (hipStream_t)stream>>>(grad, n, count);
  loss_scale_update_kernel<<<1, 1, 0, (hipStream_t)stream>>>(state, count, growth_steps, multiplier);
  NST_CHECK_LAUNCH("loss_scale_update");
  return NST_OK;
}

extern "C" int nst_cast_f32_to_bf16(const float* in, uint16_t* out, int64_t n, void* stream) {
  NST_CHECK_ARG(in && out, "cast_f32_to_bf16: null pointer");
  if (n <= 0) return NST_OK;
  cast_f32_bf16_kernel<<<grid_for(n, 256, 8192), 256, 0, (hipStream_t)stream>>>(in, out, n);
  NST_CHECK_LAUNCH("cast_f32_to_bf16");
  return NST_OK;
}
extern "C" int nst_cast_bf16_to_f32(const uint16_t* in, float* out, int64_t n, void* stream) {
  NST_CHECK_ARG(in && out, "cast_bf16_to_f32: null pointer");
  if (n <= 0) return NST_OK;
  cast_bf16_f32_kernel<<<grid_for(n, 256, 8192), 256, 0, (hipStream_t)stream>>>(in, out, n);
  NST_CHECK_LAUNCH("cast_bf16_to_f32");
  return NST_OK;
}

// =============================================================================================
// Gradient clipping on the flat gradient buffer (GradAccumKerasModel.train_step,
// neurst/training/gradaccum_keras_model.py:228-233): after the data-parallel average,
//   clip_value:  g = clamp(g, -c, +c)                      (tf.clip_by_value, every element)
//   clip_norm :  g = g * c / max(||g||_2, c)  PER TENSOR   (tf.clip_by_norm of each gradient, not the global norm)
// One table entry (<= 4096 consecutive elements of one tensor) per workgroup; the norms take two deterministic
// stages (per-entry partial sums of squares, then one workgroup per tensor).
// =============================================================================================
struct ClipEntry {
  int64_t off;
  int32_t n;
  int32_t seg;
};

__global__ void __launch_bounds__(256) clip_value_kernel(float* __restrict__ g, const ClipEntry* __restrict__ table, float pre_scale,
                                                         float clip) {
  const ClipEntry e = table[blockIdx.x];
  float* p = g + e.off;
  for (int i = threadIdx.x; i < e.n; i += 256) {
    const float v = p[i] * pre_scale;
    p[i] = fminf(fmaxf(v, -clip), clip);
  }
}

__global__ void __launch_bounds__(256) clip_sumsq_kernel(const float* __restrict__ g, const ClipEntry* __restrict__ table,
                                                         float* __restrict__ partial) {
  const ClipEntry e = table[blockIdx.x];
  const float* p = g + e.off;
  float acc = 0.f;
  for (int i = threadIdx.x; i < e.n; i += 256) acc = fmaf(p[i], p[i], acc);
  __shared__ float red[4];
  acc = wave_sum_fast(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// factor[seg] = pre_scale * c / max(pre_scale * sqrt(sum of the segment's partials), c)
__global__ void __launch_bounds__(256) clip_factor_kernel(const float* __restrict__ partial, const int32_t* __restrict__ seg_first,
                                                          float* __restrict__ factor, float pre_scale, float clip) {
  const int s = blockIdx.x;
  float acc = 0.f;
  for (int i = seg_first[s] + threadIdx.x; i < seg_first[s + 1]; i += 256) acc += partial[i];
  __shared__ float red[4];
  acc = wave_sum_fast(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float norm = pre_scale * sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    factor[s] = pre_scale * clip / fmaxf(norm, clip);
  }
}

__global__ void __launch_bounds__(256) clip_scale_kernel(float* __restrict__ g, const ClipEntry* __restrict__ table,
                                                         const float* __restrict__ factor) {
  const ClipEntry e = table[blockIdx.x];
  const float f = factor[e.seg];
  float* p = g + e.off;
  for (int i = threadIdx.x; i < e.n; i += 256) p[i] *= f;
}

extern "C" int nst_grad_clip(float* grad, const void* table, int nentries, const int32_t* seg_first, int nseg, float* workspace,
                             int64_t workspace_floats, float pre_scale, float clip_value, float clip_norm, void* stream) {
  NST_CHECK_ARG(grad && table && nentries >= 0, "grad_clip: null pointer");
  NST_CHECK_ARG((clip_value > 0.f) != (clip_norm > 0.f), "grad_clip: exactly one of clip_value / clip_norm must be positive");
  if (nentries == 0) return NST_OK;
  static_assert(sizeof(ClipEntry) == 16, "table entry layout");
  hipStream_t st = (hipStream_t)stream;
  const ClipEntry* t = (const ClipEntry*)table;
  if (clip_value > 0.f) {
    clip_value_kernel<<<nentries, 256, 0, st>>>(grad, t, pre_scale, clip_value);
    NST_CHECK_LAUNCH("grad_clip(value)");
    return NST_OK;
  }
  NST_CHECK_ARG(seg_first && nseg > 0 && workspace && workspace_floats >= (int64_t)nentries + nseg,
                "grad_clip: clip_norm needs the segment table and a workspace of nentries + nseg floats");
  float* partial = workspace;
  float* factor = workspace + nentries;
  clip_sumsq_kernel<<<nentries, 256, 0, st>>>(grad, t, partial);
  clip_factor_kernel<<<nseg, 256, 0, st>>>(partial, seg_first, factor, pre_scale, clip_norm);
  clip_scale_kernel<<<nentries, 256, 0, st>>>(grad, t, factor);
  NST_CHECK_LAUNCH("grad_clip(norm)");
  return NST_OK;
}
