// MFMA GEMM core shared by the dense projections/FFN/logits GEMMs and the implicit-GEMM conv.
//
//   C[i][j] = epilogue( alpha * sum_r Aop[i][r] * Bop[j][r] )        i<M, j<N, r<K
//
// Each operand is described by a LOADER (where element (outer, contig) lives in HBM) and a
// storage MODE:
//   RC  reduction-contiguous : tile staged in LDS as [BM rows i][BK r]   -> fragments by ds_read_b128
//   OC  output-contiguous    : tile staged in LDS as [BK rows r][BM i]   -> fragments by
//                              ds_read_b64_tr_b16 (bf16, the gfx950 LDS transpose read) or ds_read_b32 (f32)
// Staging is global -> VGPR -> LDS in 16-byte chunks along the contiguous dimension with the next tile's
// loads issued before the MFMAs of the current tile (register double buffering).  LDS rows are padded by
// 16 bytes so the 16-lane groups of ds_read_b128 hit distinct banks.
//
// Block = 256 threads = 4 waves (2x2), block tile 128x128, wave tile 64x64 = 4x4 MFMA 16x16 fragments,
// K step = 128 bytes of the reduction dimension (64 bf16 / 32 f32).
//   bf16: v_mfma_f32_16x16x32_bf16      f32: v_mfma_f32_16x16x4_f32 (exact fp32, 1/16 of the bf16 rate)
// Fragment maps (verified on hardware by nst_probe_mfma / tests/test_gpu_probe.py):
//   A/B: lane l holds row (l&15), reduction elements (l>>4)*8..+7 (bf16) or (l>>4) (f32)
//   C/D: lane l, reg j holds row (l>>4)*4+j, col (l&15)
#pragma once
#include "nst_common.h"

namespace nstgemm {

constexpr int BM = 128, BN = 128, KBYTES = 128, THREADS = 256;
constexpr int RS_RC = KBYTES + 16;  // LDS row stride (bytes) of an RC tile

enum { MODE_RC = 0, MODE_OC = 1 };

template <typename T>
struct Tile {
  static constexpr int BK = KBYTES / (int)sizeof(T);          // reduction elements per K step
  static constexpr int E = 16 / (int)sizeof(T);               // elements per 16-byte chunk
  static constexpr int RS_OC = BM * (int)sizeof(T) + 16;      // LDS row stride (bytes) of an OC tile
  static constexpr int OC_CPR = BM * (int)sizeof(T) / 16;     // chunks per OC row
  static constexpr int LDS_BYTES = (BM * RS_RC > BK * RS_OC) ? BM * RS_RC : BK * RS_OC;
};

// Division by a runtime-constant divisor without the ~40-instruction integer divide (Granlund-Montgomery
// round-up method, exact for all 32-bit n): the conv loaders decompose pixel / tap indices with it every K step.
struct FastDiv {
  uint32_t d, mul, sh1, sh2;
  void init(uint32_t div) {
    d = div;
    uint32_t l = 0;
    while ((1ull << l) < div) ++l;
    mul = (uint32_t)(((1ull << 32) * ((1ull << l) - div)) / div + 1);
    sh1 = l < 1 ? l : 1;
    sh2 = l > 0 ? l - 1 : 0;
  }
  __device__ __forceinline__ uint32_t div(uint32_t n) const {
    const uint32_t t = __umulhi(mul, n);
    return (t + ((n - t) >> sh1)) >> sh2;
  }
  __device__ __forceinline__ void divmod(uint32_t n, uint32_t& q, uint32_t& r) const {
    q = div(n);
    r = n - q * d;
  }
};

struct Epilogue {
  int vec;  // 16-byte vector epilogue is legal (host-checked alignment of C / residual / gate, N % 8 == 0)
  float alpha;
  const float* bias;
  int relu;
  uint32_t drop_thresh;
  float drop_inv_keep;
  uint64_t seed, stream_id;
  const uint64_t* seed_dev;  // device scalar added to `seed` when the kernel runs (nst_dropout_seed_offset_*)
  const void* residual;
  int64_t ldr;
  const void* gate_src;
  int64_t ldg;
  float gate_scale;
  const float* posenc;
  int posenc_period;
  float emb_scale;
  int accumulate;
  int atomic;  // split-K: atomicAdd into an f32 C
  int64_t slab_stride;  // split-K with workspace: split z stores its partial tile at C + z*slab_stride (elements)
  // fused column sums of the B operand over the reduction (bias gradient next to a weight gradient): split z writes
  // its partial sum of column j to colsum_dst[z*colsum_zstride + j]; colsum_acc adds the old value (split == 1 only)
  float* colsum_dst;
  int64_t colsum_zstride;
  int colsum_acc;
  // row dots per 64-column head (stream kernel, bf16 output only): rowdot_dst[(b*H + h)*T + t] = sum over the head's 64
  // columns of C[row][.] (as stored, i.e. rounded to bf16) * rowdot_src[row][.], row = b*T + t.  The attention output
  // projection's input gradient uses it to leave delta = rowsum(dO o O) for the attention backward in the same pass.
  const void* rowdot_src;
  int64_t ldrs;
  float* rowdot_dst;
  int rowdot_T, rowdot_H;
};

// ---------------------------------------------------------------------------------------------
// dense loader: element (outer, contig) at base[outer*ld + contig]
// ---------------------------------------------------------------------------------------------
template <typename T>
struct DenseLoader {
  const T* base;
  int64_t ld;
  int outer_limit, contig_limit;
  int vec;  // 16-byte loads are legal (alignment + contig_limit % E == 0)
  // address of the 16-byte chunk (vector-legal loaders only), nullptr when it lies outside the matrix
  __device__ __forceinline__ const T* ptr(int outer, int contig) const {
    if (outer >= outer_limit || contig >= contig_limit) return nullptr;
    return base + (int64_t)outer * ld + contig;
  }
  __device__ __forceinline__ uint4 load(int outer, int contig) const {
    uint4 r = make_uint4(0, 0, 0, 0);
    if (outer >= outer_limit || contig >= contig_limit) return r;
    const T* p = base + (int64_t)outer * ld + contig;
    if (vec) return *reinterpret_cast<const uint4*>(p);
    T tmp[Tile<T>::E];
#pragma unroll
    for (int e = 0; e < Tile<T>::E; ++e) tmp[e] = (contig + e < contig_limit) ? p[e] : (T)0;
    memcpy(&r, tmp, 16);
    return r;
  }
};

// ---------------------------------------------------------------------------------------------
// fragment readers
// ---------------------------------------------------------------------------------------------
template <typename T, int MODE, bool USE_TR>
struct FragReader;

template <bool USE_TR>
struct FragReader<bf16_t, MODE_RC, USE_TR> {
  typedef bf16x8_t Frag;
  // row = tile row (i), kk = reduction offset of this MFMA step inside the tile
  static __device__ __forceinline__ Frag read(const char* tile, int row, int kk, int lane) {
    const char* p = tile + (row + (lane & 15)) * RS_RC + (kk + (lane >> 4) * 8) * 2;
    return *reinterpret_cast<const Frag*>(p);
  }
};
template <>
struct FragReader<bf16_t, MODE_OC, true> {
  typedef bf16x8_t Frag;
  static __device__ __forceinline__ Frag read(const char* tile, int row, int kk, int lane) {
    const int ii = lane & 15;
    const int r = kk + (lane >> 4) * 8 + (ii >> 2);
    const char* p = tile + r * Tile<bf16_t>::RS_OC + (row + (ii & 3) * 4) * 2;
    typedef short4_t __attribute__((address_space(3))) * lds_ptr_t;
    short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(p));
    short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(p + 4 * Tile<bf16_t>::RS_OC));
    union { short s[8]; Frag f; } u;
    u.s[0] = lo[0]; u.s[1] = lo[1]; u.s[2] = lo[2]; u.s[3] = lo[3];
    u.s[4] = hi[0]; u.s[5] = hi[1]; u.s[6] = hi[2]; u.s[7] = hi[3];
    return u.f;
  }
};
template <>
struct FragReader<bf16_t, MODE_OC, false> {
  typedef bf16x8_t Frag;
  static __device__ __forceinline__ Frag read(const char* tile, int row, int kk, int lane) {
    const char* p = tile + (kk + (lane >> 4) * 8) * Tile<bf16_t>::RS_OC + (row + (lane & 15)) * 2;
    union { short s[8]; Frag f; } u;
#pragma unroll
    for (int j = 0; j < 8; ++j) u.s[j] = *reinterpret_cast<const short*>(p + j * Tile<bf16_t>::RS_OC);
    return u.f;
  }
};
template <bool USE_TR>
struct FragReader<float, MODE_RC, USE_TR> {
  typedef float Frag;
  static __device__ __forceinline__ Frag read(const char* tile, int row, int kk, int lane) {
    return *reinterpret_cast<const float*>(tile + (row + (lane & 15)) * RS_RC + (kk + (lane >> 4)) * 4);
  }
};
template <bool USE_TR>
struct FragReader<float, MODE_OC, USE_TR> {
  typedef float Frag;
  static __device__ __forceinline__ Frag read(const char* tile, int row, int kk, int lane) {
    return *reinterpret_cast<const float*>(tile + (kk + (lane >> 4)) * Tile<float>::RS_OC + (row + (lane & 15)) * 4);
  }
};

template <typename T>
struct Mma;
template <>
struct Mma<bf16_t> {
  static constexpr int KS = 32;
  static __device__ __forceinline__ floatx4_t run(bf16x8_t a, bf16x8_t b, floatx4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <>
struct Mma<float> {
  static constexpr int KS = 4;
  static __device__ __forceinline__ floatx4_t run(float a, float b, floatx4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};

// ---------------------------------------------------------------------------------------------
// staging: every thread moves 4 x 16-byte chunks per operand per K step
// ---------------------------------------------------------------------------------------------
template <typename T, int MODE, typename Loader>
__device__ __forceinline__ void stage_load(const Loader& ld, int o0, int r0, int tid, uint4 (&regs)[4]) {
  // o0 = first output index (i or j) of the tile, r0 = first reduction index of the K step
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int c = tid + s * THREADS;
    if (MODE == MODE_RC) {
      const int row = c >> 3, cc = c & 7;  // 8 chunks per 128-byte row
      regs[s] = ld.load(o0 + row, r0 + cc * Tile<T>::E);
    } else {
      const int row = c / Tile<T>::OC_CPR, cc = c % Tile<T>::OC_CPR;
      regs[s] = ld.load(r0 + row, o0 + cc * Tile<T>::E);
    }
  }
}
template <typename T, int MODE>
__device__ __forceinline__ void stage_store(char* tile, int tid, const uint4 (&regs)[4]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int c = tid + s * THREADS;
    if (MODE == MODE_RC) {
      const int row = c >> 3, cc = c & 7;
      *reinterpret_cast<uint4*>(tile + row * RS_RC + cc * 16) = regs[s];
    } else {
      const int row = c / Tile<T>::OC_CPR, cc = c % Tile<T>::OC_CPR;
      *reinterpret_cast<uint4*>(tile + row * Tile<T>::RS_OC + cc * 16) = regs[s];
    }
  }
}

template <typename OutT>
__device__ __forceinline__ void epilogue_store(const Epilogue& ep, OutT* __restrict__ C, int64_t ldc, int row, int col,
                                               int N, float acc, int64_t out_row) {
  float v = acc * ep.alpha;
  if (ep.bias) v += ep.bias[col];
  if (ep.relu) v = fmaxf(v, 0.f);
  if (ep.drop_thresh)
    v *= dropout_keep_scale(ep.seed, ep.stream_id, (uint64_t)row * (uint64_t)N + (uint64_t)col, ep.drop_thresh,
                            ep.drop_inv_keep);
  if (ep.residual) v += to_f32<OutT>(reinterpret_cast<const OutT*>(ep.residual)[(int64_t)row * ep.ldr + col]);
  if (ep.gate_src)
    v *= to_f32<OutT>(reinterpret_cast<const OutT*>(ep.gate_src)[(int64_t)row * ep.ldg + col]) > 0.f ? ep.gate_scale : 0.f;
  if (ep.posenc) v = v * ep.emb_scale + ep.posenc[(int64_t)(row % ep.posenc_period) * N + col];
  OutT* p = C + out_row * ldc + col;
  if (ep.atomic) {
    atomicAdd(reinterpret_cast<float*>(p), v);  // only instantiated meaningfully for OutT=float (host enforces)
  } else {
    if (ep.accumulate) v += to_f32<OutT>(*p);
    *p = from_f32<OutT>(v);
  }
}

// k_tiles_of(z, &first, &count): the K steps this block (blockIdx.z) owns.
// Vector epilogue: one lane finishes 8 consecutive columns of one row (16-byte bf16 / 2x16-byte f32 accesses);
// every optional stage is a wave-uniform branch taken once per 8 elements instead of once per element.
template <typename OutT>
__device__ __forceinline__ void epilogue_store8(const Epilogue& ep, OutT* __restrict__ C, int64_t ldc, int row, int col,
                                                int N, float (&v)[8], const float (&bias8)[8], int64_t out_row) {
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = v[j] * ep.alpha + bias8[j];
  if (ep.relu) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
  }
  if (ep.drop_thresh) {
    float m[8];  // (row*N + col) is a multiple of 8: one Philox call covers the 8 columns
    dropout_keep8(ep.seed, ep.stream_id, (uint64_t)row * (uint64_t)N + (uint64_t)col, ep.drop_thresh, ep.drop_inv_keep, m);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= m[j];
  }
  OutT tmp[8];
  if (ep.residual) {
    const OutT* p = reinterpret_cast<const OutT*>(ep.residual) + (int64_t)row * ep.ldr + col;
    *reinterpret_cast<uint4*>(tmp) = *reinterpret_cast<const uint4*>(p);
    if (sizeof(OutT) == 4) *reinterpret_cast<uint4*>(tmp + 4) = *reinterpret_cast<const uint4*>(p + 4);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += to_f32<OutT>(tmp[j]);
  }
  if (ep.gate_src) {
    const OutT* p = reinterpret_cast<const OutT*>(ep.gate_src) + (int64_t)row * ep.ldg + col;
    *reinterpret_cast<uint4*>(tmp) = *reinterpret_cast<const uint4*>(p);
    if (sizeof(OutT) == 4) *reinterpret_cast<uint4*>(tmp + 4) = *reinterpret_cast<const uint4*>(p + 4);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= to_f32<OutT>(tmp[j]) > 0.f ? ep.gate_scale : 0.f;
  }
  if (ep.posenc) {
    const float* p = ep.posenc + (int64_t)(row % ep.posenc_period) * N + col;
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    const float pe[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = v[j] * ep.emb_scale + pe[j];
  }
  OutT* o = C + out_row * ldc + col;
  if (ep.atomic) {
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(reinterpret_cast<float*>(o) + j, v[j]);
    return;
  }
  if (ep.accumulate) {
    *reinterpret_cast<uint4*>(tmp) = *reinterpret_cast<const uint4*>(o);
    if (sizeof(OutT) == 4) *reinterpret_cast<uint4*>(tmp + 4) = *reinterpret_cast<const uint4*>(o + 4);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += to_f32<OutT>(tmp[j]);
  }
  if (sizeof(OutT) == 2) {
    *reinterpret_cast<uint4*>(o) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                              pack_bf16x2(v[6], v[7]));
  } else {
    *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(o) + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
}

struct IdentityRowMap {
  int unused = 0;  // keeps the struct a dword multiple (kernarg-resident arguments are copied dword-wise)
  __device__ __forceinline__ int64_t operator()(int row) const { return row; }
};

// Shared epilogue of both main loops (see gemm_block).
template <typename OutT, typename RowMap, int NWN = 2>
__device__ __forceinline__ void gemm_epilogue(floatx4_t (&acc)[4][4], OutT* __restrict__ C, int64_t ldc, int M, int N, int m0,
                                              int n0, const Epilogue& ep, char* smem, const RowMap& rowmap) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave / NWN) * 64, wn = (wave % NWN) * 64;  // NWN waves side by side over a tile of NWN*64 columns
  // Epilogue through LDS: each wave transposes its 64x64 accumulator tile in two 32-row halves so that global
  // accesses are row-contiguous.  Fast path (interior tile, aligned): a lane finishes 8 consecutive columns with
  // 16-byte accesses; edge tiles fall back to one element per lane.
  constexpr int EPI_LD = 68;  // floats; multiple of 4 keeps the float4 reads 16-byte aligned
  float* epi = reinterpret_cast<float*>(smem) + wave * (32 * EPI_LD);
  const int lr = (lane >> 4) * 4, lc = lane & 15;
  const bool fast = ep.vec && (m0 + BM <= M) && (n0 + NWN * 64 <= N);
  const int vrow = lane >> 3, vcol = (lane & 7) * 8;
  float bias8[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bias8[j] = (fast && ep.bias) ? ep.bias[n0 + wn + vcol + j] : 0.f;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const floatx4_t a4 = acc[half * 2 + ii][j];
        epi[(ii * 16 + lr + 0) * EPI_LD + j * 16 + lc] = a4[0];
        epi[(ii * 16 + lr + 1) * EPI_LD + j * 16 + lc] = a4[1];
        epi[(ii * 16 + lr + 2) * EPI_LD + j * 16 + lc] = a4[2];
        epi[(ii * 16 + lr + 3) * EPI_LD + j * 16 + lc] = a4[3];
      }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes have landed (wave-private region)
    __builtin_amdgcn_wave_barrier();
    const int row_base = m0 + wm + half * 32;
    if (fast) {
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int rl = pass * 8 + vrow;
        const float4 x0 = *reinterpret_cast<const float4*>(epi + rl * EPI_LD + vcol);
        const float4 x1 = *reinterpret_cast<const float4*>(epi + rl * EPI_LD + vcol + 4);
        float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        const int row = row_base + rl;
        epilogue_store8<OutT>(ep, C, ldc, row, n0 + wn + vcol, N, v, bias8, rowmap(row));
      }
    } else {
      const int col = n0 + wn + lane;
      if (col < N) {
        for (int r = 0; r < 32; ++r) {
          const int row = row_base + r;
          if (row < M) epilogue_store<OutT>(ep, C, ldc, row, col, N, epi[r * EPI_LD + lane], rowmap(row));
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// RowMap: logical output row -> row index in C (identity for dense GEMMs; the conv dgrad scatters its
// parity-class rows back to pixel order).
template <typename T, typename OutT, int AMODE, int BMODE, bool USE_TR, typename ALoader, typename BLoader,
          typename RowMap = IdentityRowMap>
__device__ __forceinline__ void gemm_block(const ALoader& la, const BLoader& lb, OutT* __restrict__ C, int64_t ldc, int M,
                                           int N, int m0, int n0, int kt_first, int kt_count, const Epilogue& ep,
                                           char* smem, const RowMap rowmap = RowMap()) {
  typedef FragReader<T, AMODE, USE_TR> RA;
  typedef FragReader<T, BMODE, USE_TR> RB;
  constexpr int BK = Tile<T>::BK;
  constexpr int KS = Mma<T>::KS;
  char* As = smem;
  char* Bs = smem + Tile<T>::LDS_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

  floatx4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = floatx4_t{0.f, 0.f, 0.f, 0.f};

  uint4 ra[4], rb[4];
  if (kt_count > 0) {
    stage_load<T, AMODE>(la, m0, kt_first * BK, tid, ra);
    stage_load<T, BMODE>(lb, n0, kt_first * BK, tid, rb);
  }
  for (int kt = 0; kt < kt_count; ++kt) {
    stage_store<T, AMODE>(As, tid, ra);
    stage_store<T, BMODE>(Bs, tid, rb);
    __syncthreads();
    if (kt + 1 < kt_count) {
      stage_load<T, AMODE>(la, m0, (kt_first + kt + 1) * BK, tid, ra);
      stage_load<T, BMODE>(lb, n0, (kt_first + kt + 1) * BK, tid, rb);
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += KS) {
      typename RA::Frag a[4];
      typename RB::Frag b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = RA::read(As, wm + i * 16, kk, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = RB::read(Bs, wn + j * 16, kk, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::run(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }

  gemm_epilogue<OutT, RowMap>(acc, C, ldc, M, N, m0, n0, ep, smem, rowmap);
}
// =============================================================================================
// v2 main loop: global -> LDS by LDS-DMA (global_load_lds_dwordx4), NST-stage ring, counted vmcnt.
//
//  * no VGPR staging and no ds_write pass: a lane only computes the global address of its 16-byte chunk; the chunk
//    lands at LDS byte (wave-uniform base + lane*16), i.e. the stage image is lane-linear and UNPADDED;
//  * bank conflicts are removed by an XOR swizzle applied to the SOURCE chunk index and to the fragment read address
//    (same involution on both sides):  RC tile (128-byte rows): 16-byte slot ^= (row>>1)&7
//                                      OC tile (BM*sizeof(T)-byte rows): 32-byte pair ^= (r&3) | ((r>>3)&1)<<2
//  * the loads of tile t+NST-1 are issued right after the barrier that starts tile t, so NST-1 tiles are in flight
//    while tile t is multiplied; the wait before the barrier is a COUNTED s_waitcnt vmcnt(N), never a drain;
//  * the DMA is issued from inline asm so that hipcc's own waitcnt insertion (which would drain vmcnt(0) in front of
//    every ds_read it cannot prove disjoint) does not see it; out-of-range chunks read a 16-byte zero block.
// =============================================================================================
__device__ __attribute__((aligned(16))) const uint4 g_nst_zero16[1] = {{0u, 0u, 0u, 0u}};

__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_byte_addr_uniform) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_byte_addr_uniform)
      : "memory");
}

template <typename T, int MODE>
struct SwzFrag;  // fragment readers for the unpadded, swizzled stage images
template <>
struct SwzFrag<bf16_t, MODE_RC> {
  typedef bf16x8_t Frag;
  static __device__ __forceinline__ Frag read(const char* tile, int row, int kk, int lane) {
    const int r = row + (lane & 15);
    const int slot = ((kk >> 3) + (lane >> 4)) ^ ((r >> 1) & 7);
    return *reinterpret_cast<const Frag*>(tile + r * KBYTES + slot * 16);
  }
};
template <>
struct SwzFrag<bf16_t, MODE_OC> {
  typedef bf16x8_t Frag;
  static __device__ __forceinline__ Frag read(const char* tile, int row, int kk, int lane) {
    constexpr int RB = BM * 2;  // row bytes
    const int ii = lane & 15;
    const int r0 = kk + (lane >> 4) * 8 + (ii >> 2), r1 = r0 + 4;
    const int off = (row + (ii & 3) * 4) * 2;
    const int s0 = ((r0 & 3) | (((r0 >> 3) & 1) << 2)) << 5, s1 = ((r1 & 3) | (((r1 >> 3) & 1) << 2)) << 5;
    typedef short4_t __attribute__((address_space(3))) * lds_ptr_t;
    short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(tile + r0 * RB + (off ^ s0)));
    short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(tile + r1 * RB + (off ^ s1)));
    // register-pair concatenation: element-wise copies through a union became v_bfi_b32 merges behind a full
    // s_waitcnt lgkmcnt(0) -- every fragment read of the K step had to land before the first MFMA
    typedef __attribute__((ext_vector_type(8))) short short8_t;
    return __builtin_bit_cast(Frag, (short8_t)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  }
};
template <>
struct SwzFrag<float, MODE_RC> {
  typedef float Frag;
  static __device__ __forceinline__ Frag read(const char* tile, int row, int kk, int lane) {
    const int r = row + (lane & 15), k = kk + (lane >> 4);
    const int slot = (k >> 2) ^ ((r >> 1) & 7);
    return *reinterpret_cast<const float*>(tile + r * KBYTES + slot * 16 + (k & 3) * 4);
  }
};
template <>
struct SwzFrag<float, MODE_OC> {
  typedef float Frag;
  static __device__ __forceinline__ Frag read(const char* tile, int row, int kk, int lane) {
    constexpr int RB = BM * 4;
    const int r = kk + (lane >> 4);
    const int off = (row + (lane & 15)) * 4;
    const int s = ((r & 3) | (((r >> 3) & 1) << 2)) << 5;
    return *reinterpret_cast<const float*>(tile + r * RB + (off ^ s));
  }
};

// issue the LDS-DMA of one operand tile (16 KB = 4 wave-instructions per wave)
template <typename T, int MODE, typename Loader>
__device__ __forceinline__ void dma_tile(const Loader& ld, int o0, int r0, uint32_t tile_lds_addr, int wave, int lane) {
  constexpr int E = Tile<T>::E;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int cbase = (s * 4 + wave) * 64;  // first chunk of this wave-instruction (wave uniform)
    const int c = cbase + lane;
    const T* p;
    if (MODE == MODE_RC) {
      const int row = c >> 3, slot = c & 7;
      const int kchunk = slot ^ ((row >> 1) & 7);
      p = ld.ptr(o0 + row, r0 + kchunk * E);
    } else {
      constexpr int CPR = Tile<T>::OC_CPR;
      const int r = c / CPR, c16 = c % CPR;
      const int g = (r & 3) | (((r >> 3) & 1) << 2);
      p = ld.ptr(r0 + r, o0 + (c16 ^ (g << 1)) * E);
    }
    const void* src = p ? (const void*)p : (const void*)g_nst_zero16;
    glds16(src, __builtin_amdgcn_readfirstlane(tile_lds_addr + (uint32_t)cbase * 16u));
  }
}

// Per-thread DMA cursor of one operand tile: the 4 chunk pointers are computed once per tile; every K step only adds a
// constant byte stride (dense loaders: the generic path above re-derives 64-bit addresses 8 times per step, which on a
// one-wave-per-SIMD kernel is serialised VALU work next to the MFMAs).
template <typename T, int MODE>
struct DenseDma {
  const char* p[4];
  int kvalid_base[4];   // reduction index of the chunk at K step 0 (RC: element k; OC: row r)
  bool ovalid[4];       // the chunk's fixed (non-reduction) coordinate is inside the matrix
  int64_t step_bytes;
  int k_limit;
  __device__ __forceinline__ void init(const DenseLoader<T>& ld, int o0, int r0, int wave, int lane) {
    constexpr int E = Tile<T>::E;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int c = (s * 4 + wave) * 64 + lane;
      if (MODE == MODE_RC) {
        const int row = c >> 3, slot = c & 7;
        const int kchunk = slot ^ ((row >> 1) & 7);
        ovalid[s] = (o0 + row) < ld.outer_limit;
        kvalid_base[s] = r0 + kchunk * E;
        p[s] = reinterpret_cast<const char*>(ld.base + (int64_t)(o0 + row) * ld.ld + kvalid_base[s]);
      } else {
        constexpr int CPR = Tile<T>::OC_CPR;
        const int r = c / CPR, c16 = c % CPR;
        const int g = (r & 3) | (((r >> 3) & 1) << 2);
        const int col = o0 + (c16 ^ (g << 1)) * E;
        ovalid[s] = col < ld.contig_limit;
        kvalid_base[s] = r0 + r;
        p[s] = reinterpret_cast<const char*>(ld.base + (int64_t)kvalid_base[s] * ld.ld + col);
      }
    }
    step_bytes = MODE == MODE_RC ? (int64_t)Tile<T>::BK * (int64_t)sizeof(T) : (int64_t)Tile<T>::BK * ld.ld * (int64_t)sizeof(T);
    k_limit = MODE == MODE_RC ? ld.contig_limit : ld.outer_limit;
  }
  // v3 streaming interface.  begin(): after init(), decides (wave-uniformly) whether every chunk of every K step of the
  // unit lies inside the matrix; the cursors then only advance by a constant and the 4 DMA pieces go out back to back
  // from one asm block (m0 = LDS base of the piece, +4 KB per piece).  Otherwise next() falls back to issue().
  int n_fast;  // the first n_fast K steps of the unit are entirely inside the matrix (wave-uniform)
  int t_next;
  __device__ __forceinline__ void begin(int kt_count) {
    bool ok = true;
    int kmax = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      ok = ok && ovalid[s];
      kmax = kvalid_base[s] > kmax ? kvalid_base[s] : kmax;
    }
    // step t is inside iff kmax + t*BK < k_limit for the largest kmax of the wave
    int nf = (k_limit - 1 - kmax) >= 0 ? (k_limit - 1 - kmax) / Tile<T>::BK + 1 : 0;
    if (__builtin_amdgcn_ballot_w64(!ok) != 0) nf = 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(nf, o, 64); nf = other < nf ? other : nf; }
    n_fast = __builtin_amdgcn_readfirstlane(nf < kt_count ? nf : kt_count);
    t_next = 0;
  }
  __device__ __forceinline__ void next(uint32_t piece0_lds_addr_uniform, uint32_t tile_lds_addr, int wave) {
    if (t_next < n_fast) {
      uint32_t keep;
      asm volatile(
          "s_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %5\n\t"
          "s_nop 0\n\t"
          "global_load_lds_dwordx4 %1, off\n\t"
          "s_add_u32 m0, m0, 0x1000\n\t"
          "s_nop 0\n\t"
          "global_load_lds_dwordx4 %2, off\n\t"
          "s_add_u32 m0, m0, 0x1000\n\t"
          "s_nop 0\n\t"
          "global_load_lds_dwordx4 %3, off\n\t"
          "s_add_u32 m0, m0, 0x1000\n\t"
          "s_nop 0\n\t"
          "global_load_lds_dwordx4 %4, off\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "s"(piece0_lds_addr_uniform)
          : "memory", "scc");
#pragma unroll
      for (int s = 0; s < 4; ++s) p[s] += step_bytes;
    } else {  // edge steps: per-chunk bounds, out-of-range chunks read the zero block
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bool ok = ovalid[s] && (kvalid_base[s] + t_next * Tile<T>::BK) < k_limit;
        const void* src = ok ? (const void*)p[s] : (const void*)g_nst_zero16;
        glds16(src, __builtin_amdgcn_readfirstlane(tile_lds_addr + (uint32_t)((s * 4 + wave) * 64) * 16u));
        p[s] += step_bytes;
      }
    }
    ++t_next;
  }
  // issue the DMA of K step `t` (relative to the r0 given to init) into the stage at tile_lds_addr
  __device__ __forceinline__ void issue(int t, uint32_t tile_lds_addr, int wave) const {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bool ok = ovalid[s] && (kvalid_base[s] + t * Tile<T>::BK) < k_limit;
      const void* src = ok ? (const void*)(p[s] + (int64_t)t * step_bytes) : (const void*)g_nst_zero16;
      glds16(src, __builtin_amdgcn_readfirstlane(tile_lds_addr + (uint32_t)((s * 4 + wave) * 64) * 16u));
    }
  }
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// TileDma: DenseDma for dense loaders, the generic per-step address derivation for the conv gather loaders
template <typename T, int MODE, typename Loader>
struct TileDma {
  const Loader* ld;
  int o0, r0, lane;
  __device__ __forceinline__ void init(const Loader& l, int o0_, int r0_, int wave, int lane_) { ld = &l; o0 = o0_; r0 = r0_; lane = lane_; }
  __device__ __forceinline__ void issue(int t, uint32_t tile_lds_addr, int wave) const {
    dma_tile<T, MODE, Loader>(*ld, o0, r0 + t * Tile<T>::BK, tile_lds_addr, wave, lane);
  }
  int t_next;
  __device__ __forceinline__ void begin(int) { t_next = 0; }
  __device__ __forceinline__ void next(uint32_t, uint32_t tile_lds_addr, int wave) { issue(t_next++, tile_lds_addr, wave); }
};
template <typename T, int MODE>
struct TileDma<T, MODE, DenseLoader<T>> : DenseDma<T, MODE> {};

constexpr int V2_STAGE_BYTES = 2 * BM * KBYTES;  // A tile + B tile, 32 KB

template <typename T>
__device__ __forceinline__ typename SwzFrag<T, MODE_RC>::Frag ones_frag() {
  if constexpr (sizeof(T) == 2) {
    union { uint32_t u[4]; bf16x8_t f; } o;
    o.u[0] = o.u[1] = o.u[2] = o.u[3] = 0x3F803F80u;  // bf16 1.0 pairs
    return o.f;
  } else {
    return 1.0f;
  }
}

// CS: the first row of tiles (m0 == 0) also accumulates ones^T . Bop -- every row of that accumulator is the column
// sum of the B operand over this block's reduction range (one extra MFMA per B fragment in 2 of the 4 waves).
template <typename T, typename OutT, int AMODE, int BMODE, int NST, typename ALoader, typename BLoader,
          typename RowMap = IdentityRowMap, bool CS = false>
__device__ __forceinline__ void gemm_block_v2(const ALoader& la, const BLoader& lb, OutT* __restrict__ C, int64_t ldc, int M,
                                              int N, int m0, int n0, int kt_first, int kt_count, const Epilogue& ep,
                                              char* smem, const RowMap rowmap = RowMap()) {
  typedef SwzFrag<T, AMODE> RA;
  typedef SwzFrag<T, BMODE> RB;
  constexpr int BK = Tile<T>::BK;
  constexpr int KS = Mma<T>::KS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  typedef __attribute__((address_space(3))) char* lds_char_ptr;
  const uint32_t smem_addr = (uint32_t)(uintptr_t)((lds_char_ptr)smem);

  floatx4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = floatx4_t{0.f, 0.f, 0.f, 0.f};

  floatx4_t cs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) cs[j] = floatx4_t{0.f, 0.f, 0.f, 0.f};
  const bool do_cs = CS && ep.colsum_dst && m0 == 0 && wm == 0;  // wave-uniform
  const typename RA::Frag ones = ones_frag<T>();

  TileDma<T, AMODE, ALoader> da;
  TileDma<T, BMODE, BLoader> db;
  da.init(la, m0, kt_first * BK, wave, lane);
  db.init(lb, n0, kt_first * BK, wave, lane);
  // prologue: NST-1 tiles in flight
#pragma unroll
  for (int s = 0; s < NST - 1; ++s) {
    if (s < kt_count) {
      da.issue(s, smem_addr + s * V2_STAGE_BYTES, wave);
      db.issue(s, smem_addr + s * V2_STAGE_BYTES + BM * KBYTES, wave);
    }
  }
  int stage = 0;
  for (int kt = 0; kt < kt_count; ++kt) {
    // tile kt must have landed; tiles kt+1 .. kt+NST-2 (8 DMA instructions each) may stay in flight
    const int ahead = (kt_count - 1 - kt) < (NST - 2) ? (kt_count - 1 - kt) : (NST - 2);
    if (ahead <= 0) wait_vmcnt<0>();
    else if (ahead == 1) wait_vmcnt<8>();
    else wait_vmcnt<16>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int nxt = kt + NST - 1;
    if (nxt < kt_count) {
      int st2 = stage + NST - 1;
      if (st2 >= NST) st2 -= NST;
      da.issue(nxt, smem_addr + st2 * V2_STAGE_BYTES, wave);
      db.issue(nxt, smem_addr + st2 * V2_STAGE_BYTES + BM * KBYTES, wave);
    }
    const char* As = smem + stage * V2_STAGE_BYTES;
    const char* Bs = As + BM * KBYTES;
#pragma unroll
    for (int kk = 0; kk < BK; kk += KS) {
      typename RA::Frag a[4];
      typename RB::Frag b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = RA::read(As, wm + i * 16, kk, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = RB::read(Bs, wn + j * 16, kk, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::run(a[i], b[j], acc[i][j]);
      if (CS && do_cs) {
#pragma unroll
        for (int j = 0; j < 4; ++j) cs[j] = Mma<T>::run(ones, b[j], cs[j]);
      }
    }
    stage = stage + 1 == NST ? 0 : stage + 1;
  }
  if (CS && do_cs && lane < 16) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n0 + wn + j * 16 + lane;
      if (col < N) {
        float* dst = ep.colsum_dst + (int64_t)blockIdx.z * ep.colsum_zstride + col;
        *dst = ep.colsum_acc ? *dst + cs[j][0] : cs[j][0];
      }
    }
  }
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();  // every wave is done reading the last stage: the epilogue reuses the LDS
  asm volatile("" ::: "memory");
  gemm_epilogue<OutT, RowMap>(acc, C, ldc, M, N, m0, n0, ep, smem, rowmap);
}

// Wide variant of the v2 block for outputs with more than 128 columns (conv2: N = C = 256): one workgroup of 8 waves
// owns a 128 x 256 tile, i.e. the (large, streamed) A operand is read ONCE per row panel instead of once per 128-column
// tile -- the two column tiles of a row panel otherwise fetch the same im2col rows twice from HBM (measured 2.0x the
// algorithmic traffic, which made the conv2 kernels HBM-bound).  Stage = [A | B0 | B1] = 48 KB, two stages; the B
// operand is kept as two independent 128-column tile images so the fragment readers are unchanged.  Waves 0-3 issue
// the DMA of A and B0, waves 4-7 that of B1.
constexpr int V2W_THREADS = 512;
constexpr int V2W_STAGE_BYTES = 3 * BM * KBYTES;
constexpr int V2W_LDS_BYTES = 2 * V2W_STAGE_BYTES;

// kt_first / z: split-K slice (first K step, slab index of the fused column sums).  CS as in gemm_block_v2: the first row
// of tiles also accumulates ones^T . Bop (the bias gradient of a weight-gradient product).
template <typename T, typename OutT, int AMODE, int BMODE, typename ALoader, typename BLoader, typename RowMap, bool CS = false>
__device__ __forceinline__ void gemm_block_v2w(const ALoader& la, const BLoader& lb, OutT* __restrict__ C, int64_t ldc, int M, int N,
                                               int m0, int n0, int kt_count, const Epilogue& ep, char* smem, const RowMap rowmap,
                                               int kt_first = 0, int z = 0) {
  typedef SwzFrag<T, AMODE> RA;
  typedef SwzFrag<T, BMODE> RB;
  constexpr int BK = Tile<T>::BK;
  constexpr int KS = Mma<T>::KS;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int quad = wave >> 2, wq = wave & 3;
  const int wm = (wave >> 2) * 64, wn = (wave & 3) * 64;
  typedef __attribute__((address_space(3))) char* lds_char_ptr;
  const uint32_t smem_addr = (uint32_t)(uintptr_t)((lds_char_ptr)smem);

  floatx4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = floatx4_t{0.f, 0.f, 0.f, 0.f};

  floatx4_t cs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) cs[j] = floatx4_t{0.f, 0.f, 0.f, 0.f};
  const bool do_cs = CS && ep.colsum_dst && m0 == 0 && wm == 0;  // wave-uniform
  const typename RA::Frag ones = ones_frag<T>();

  TileDma<T, AMODE, ALoader> da;
  TileDma<T, BMODE, BLoader> db;
  if (quad == 0) da.init(la, m0, kt_first * BK, wq, lane);
  db.init(lb, n0 + quad * BN, kt_first * BK, wq, lane);
  auto issue = [&](int kt, int stage) {
    const uint32_t sa = smem_addr + stage * V2W_STAGE_BYTES;
    if (quad == 0) {
      da.issue(kt, sa, wq);
      db.issue(kt, sa + BM * KBYTES, wq);
    } else {
      db.issue(kt, sa + 2 * BM * KBYTES, wq);
    }
  };
  if (kt_count > 0) issue(0, 0);
  int stage = 0;
  for (int kt = 0; kt < kt_count; ++kt) {
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + 1 < kt_count) issue(kt + 1, stage ^ 1);
    const char* As = smem + stage * V2W_STAGE_BYTES;
    const char* Bs = As + BM * KBYTES * (1 + (wn >> 7));
    const int wnl = wn & 127;
    if constexpr (BK / KS == 2) {
      // both halves of the K step are requested up front: the second set of fragments lands under the first 16 MFMAs
      typename RA::Frag a0[4], a1[4];
      typename RB::Frag b0[4], b1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b0[j] = RB::read(Bs, wnl + j * 16, 0, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) a0[i] = RA::read(As, wm + i * 16, 0, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) b1[j] = RB::read(Bs, wnl + j * 16, KS, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) a1[i] = RA::read(As, wm + i * 16, KS, lane);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = Mma<T>::run(a0[i], b0[j], acc[i][j]);
      if (CS && do_cs) {
#pragma unroll
        for (int j = 0; j < 4; ++j) cs[j] = Mma<T>::run(ones, b0[j], cs[j]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = Mma<T>::run(a1[i], b1[j], acc[i][j]);
      if (CS && do_cs) {
#pragma unroll
        for (int j = 0; j < 4; ++j) cs[j] = Mma<T>::run(ones, b1[j], cs[j]);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < BK; kk += KS) {
        typename RA::Frag a[4];
        typename RB::Frag b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = RA::read(As, wm + i * 16, kk, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = RB::read(Bs, wnl + j * 16, kk, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::run(a[i], b[j], acc[i][j]);
        if (CS && do_cs) {
#pragma unroll
          for (int j = 0; j < 4; ++j) cs[j] = Mma<T>::run(ones, b[j], cs[j]);
        }
      }
    }
    stage ^= 1;
  }
  if (CS && do_cs && lane < 16) {  // every row of cs holds the column sums; lane = column within the 16-block
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n0 + wn + j * 16 + lane;
      if (col < N) {
        float* dst = ep.colsum_dst + (int64_t)z * ep.colsum_zstride + col;
        *dst = ep.colsum_acc ? *dst + cs[j][0] : cs[j][0];
      }
    }
  }
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();  // every wave is done reading the last stage: the epilogue reuses the LDS
  asm volatile("" ::: "memory");
  gemm_epilogue<OutT, RowMap, 4>(acc, C, ldc, M, N, m0, n0, ep, smem, rowmap);
}

// XCD-aware tile order: consecutive block ids land on different XCDs (id % 8); give each XCD a contiguous
// run of tiles so neighbouring tiles (which share an A row panel) hit the same L2.  Bijective for any count.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7, xcd = bid & 7, idx = bid >> 3;
  const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + idx;
}


// =============================================================================================
// v3: persistent stream of (output tile, K step) work over the same LDS-DMA stage ring.
//
//  * accumulators are kept TRANSPOSED: acc[i][j] = Bfrag[j] (MFMA A operand, rows = n) x Afrag[i] (MFMA B operand,
//    columns = m), so lane l owns row m = i*16 + (l&15) and the 4 CONSECUTIVE columns n = j*16 + (l>>4)*4 + 0..3 of
//    every 16x16 block.  The epilogue is then written straight from registers with 8-byte (bf16) / 16-byte (f32)
//    stores -- no LDS transposition, no barrier, and the LDS stays free for the DMA ring;
//  * a workgroup walks units u = blockIdx.x, +gridDim.x, ... (unit = output tile x split-K slice); the DMA of the
//    first K step of unit u+1 is issued before the last K step of unit u is multiplied, so the global-load latency of
//    a new tile and the stores of the finished one overlap (short-K GEMMs are otherwise dominated by both);
//  * grid = units / ceil(units / resident workgroups): every workgroup gets the same number of units.
// =============================================================================================
// out-of-line copy of the per-element mask (keeps the rarely taken unaligned paths from bloating the kernels)
__device__ __attribute__((noinline)) float dropout_keep_scale_call(uint64_t seed, uint64_t stream, uint64_t idx, uint32_t thresh16,
                                                                   float inv_keep) {
  return dropout_keep_scale(seed, stream, idx, thresh16, inv_keep);
}

// register select acc[i][j][r] by a runtime index (compare/select chain; only the unaligned scalar epilogue uses it)
__device__ __forceinline__ float acc_pick(const floatx4_t (&acc)[4][4], int q) {
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) v = (q == (i * 4 + j) * 4 + r) ? acc[i][j][r] : v;
  return v;
}

// v3 epilogue.  The accumulators of a wave (64x64 outputs, lane l: column j*16 + (l&15), rows i*16 + (l>>4)*4 + r) are
// turned into row-major pieces through a 4 KB wave-private LDS buffer, 16 rows x 64 f32 at a time: afterwards lane l
// owns the 16 consecutive columns (l&3)*16.. of row l>>2, i.e. 4 adjacent lanes cover one 128-byte (bf16) row segment
// and the stores coalesce.  The buffer lives behind the DMA stages (it is never a DMA target), so the prefetch of
// the next unit keeps running underneath.  16-byte chunk index ^= row & 3 keeps the b128 reads conflict free.
// Compile-time epilogue configuration of the v3 stream kernels.  EF < 0 (EF_GENERIC): every stage is a runtime branch on the
// Epilogue fields -- one kernel serves every call, but it keeps ~100 scalars of the argument block alive around its
// epilogue (205-224 SGPR spills in round 2) and pays a branch per stage per 16 outputs.  EF >= 0: a bit mask of the stages
// that EXIST in the instantiation; the dispatcher (nst_gemm.hip) picks it when the call's stages equal the mask exactly, the
// vector epilogue is legal and alpha == 1.  Unused fields of the argument block are then never loaded.
enum : int { EF_BIAS = 1, EF_RELU = 2, EF_DROP = 4, EF_RESID = 8, EF_GATE = 16, EF_POSENC = 32, EF_ACCUM = 64, EF_ROWDOT = 128,
             EF_GENERIC = -1 };
template <int EF, int F>
__device__ __forceinline__ bool ef_on(bool runtime) {
  if constexpr (EF < 0) return runtime;
  else return (EF & F) != 0;
}

constexpr int V3_EPI_BYTES_PER_WAVE = 16 * 64 * 4;
constexpr int V3_LDS_BYTES = 2 * V2_STAGE_BYTES + 4 * V3_EPI_BYTES_PER_WAVE;  // 80 KB: two workgroups fill the 160 KB of a CU

template <typename OutT, int EF = EF_GENERIC>
__device__ __forceinline__ void epi_piece16_v3(float (&v)[16], OutT* __restrict__ C, int64_t ldc, int row, int64_t out_row, int n, int N,
                                               const Epilogue& ep) {
  // v: outputs (row, n .. n+15) with alpha and bias already applied
  if (ef_on<EF, EF_RELU>(ep.relu)) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
  }
  if (ef_on<EF, EF_DROP>(ep.drop_thresh != 0)) {
    float m0[8], m1[8];
    const uint64_t idx = (uint64_t)row * (uint64_t)N + (uint64_t)n;  // multiple of 8 (N % 8 == 0, n % 16 == 0)
    dropout_keep8(ep.seed, ep.stream_id, idx, ep.drop_thresh, ep.drop_inv_keep, m0);
    dropout_keep8(ep.seed, ep.stream_id, idx + 8, ep.drop_thresh, ep.drop_inv_keep, m1);
#pragma unroll
    for (int r = 0; r < 8; ++r) { v[r] *= m0[r]; v[8 + r] *= m1[r]; }
  }
  constexpr int NV = (int)sizeof(OutT);  // 16-byte vectors per 16 outputs: 2 (bf16) or 4 (f32)
  const int nv = (n + 16 <= N) ? NV : NV / 2;  // N % 8 == 0: the last piece of a row may hold only 8 outputs
  OutT tmp[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) tmp[r] = (OutT)0;
  if (ef_on<EF, EF_RESID>(ep.residual != nullptr)) {
    const OutT* p = reinterpret_cast<const OutT*>(ep.residual) + (int64_t)row * ep.ldr + n;
#pragma unroll
    for (int q = 0; q < NV; ++q)
      if (q < nv) reinterpret_cast<uint4*>(tmp)[q] = reinterpret_cast<const uint4*>(p)[q];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] += to_f32<OutT>(tmp[r]);
  }
  if (ef_on<EF, EF_GATE>(ep.gate_src != nullptr)) {
    const OutT* p = reinterpret_cast<const OutT*>(ep.gate_src) + (int64_t)row * ep.ldg + n;
#pragma unroll
    for (int q = 0; q < NV; ++q)
      if (q < nv) reinterpret_cast<uint4*>(tmp)[q] = reinterpret_cast<const uint4*>(p)[q];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] *= to_f32<OutT>(tmp[r]) > 0.f ? ep.gate_scale : 0.f;
  }
  if (ef_on<EF, EF_POSENC>(ep.posenc != nullptr)) {
    const float* p = ep.posenc + (int64_t)(row % ep.posenc_period) * N + n;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (n + q * 4 < N) {
        const float4 pe = reinterpret_cast<const float4*>(p)[q];
        v[q * 4 + 0] = v[q * 4 + 0] * ep.emb_scale + pe.x; v[q * 4 + 1] = v[q * 4 + 1] * ep.emb_scale + pe.y;
        v[q * 4 + 2] = v[q * 4 + 2] * ep.emb_scale + pe.z; v[q * 4 + 3] = v[q * 4 + 3] * ep.emb_scale + pe.w;
      }
    }
  }
  OutT* o = C + out_row * ldc + n;
  if constexpr (EF < 0) {   // atomic split-K exists in the generic kernel only
    if (ep.atomic) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (n + r < N) atomicAdd(reinterpret_cast<float*>(o) + r, v[r]);
      return;
    }
  }
  if (ef_on<EF, EF_ACCUM>(ep.accumulate != 0)) {
#pragma unroll
    for (int q = 0; q < NV; ++q)
      if (q < nv) reinterpret_cast<uint4*>(tmp)[q] = reinterpret_cast<const uint4*>(o)[q];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] += to_f32<OutT>(tmp[r]);
  }
  if (sizeof(OutT) == 2) {
    const uint4 w0 = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
    const uint4 w1 = make_uint4(pack_bf16x2(v[8], v[9]), pack_bf16x2(v[10], v[11]), pack_bf16x2(v[12], v[13]), pack_bf16x2(v[14], v[15]));
    reinterpret_cast<uint4*>(o)[0] = w0;
    if (nv == NV) reinterpret_cast<uint4*>(o)[1] = w1;
    if (ef_on<EF, EF_ROWDOT>(ep.rowdot_dst != nullptr)) {  // host guarantees N % 64 == 0: the 4 lanes of a quad hold the 64 columns of one head of this row
      const bf16_t* sp = reinterpret_cast<const bf16_t*>(ep.rowdot_src) + (int64_t)row * ep.ldrs + n;
      const uint4 s0 = reinterpret_cast<const uint4*>(sp)[0], s1 = reinterpret_cast<const uint4*>(sp)[1];
      const uint32_t cw[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
      const uint32_t sw[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      float dot = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        dot = fmaf(__uint_as_float(cw[q] << 16), __uint_as_float(sw[q] << 16), dot);
        dot = fmaf(__uint_as_float(cw[q] & 0xffff0000u), __uint_as_float(sw[q] & 0xffff0000u), dot);
      }
      dot = dpp_add(dot, 0);   // + lane ^ 1
      dot = dpp_add(dot, 1);   // + lane ^ 2: every lane of the quad holds the head's sum
      if ((n & 63) == 0) {
        const int b = row / ep.rowdot_T, t = row - b * ep.rowdot_T;
        ep.rowdot_dst[((int64_t)b * ep.rowdot_H + (n >> 6)) * ep.rowdot_T + t] = dot;
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < nv) reinterpret_cast<float4*>(o)[q] = make_float4(v[q * 4], v[q * 4 + 1], v[q * 4 + 2], v[q * 4 + 3]);
  }
}

// one 16-row block i of the wave tile: acc_i[j][r] = output (row g*4 + r, column j*16 + lc) of the block
template <typename OutT, typename RowMap, int EF = EF_GENERIC>
__device__ __forceinline__ void epi_block_v3(const floatx4_t (&a)[4], float* __restrict__ epi, OutT* __restrict__ C, int64_t ldc, int M,
                                             int N, int row0, int nw, const float (&bias16)[16], const Epilogue& ep,
                                             const RowMap& rowmap, int lane) {
  const int g = lane >> 4, lc = lane & 15;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = g * 4 + r, chunk = (j * 4 + (lc >> 2)) ^ r;  // row & 3 == r
      epi[row * 64 + chunk * 4 + (lc & 3)] = a[j][r];
    }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes have landed (wave-private region)
  __builtin_amdgcn_wave_barrier();
  const int rr = lane >> 2, c16 = lane & 3;  // this lane: row rr, columns c16*16 .. +15
  float v[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 x = *reinterpret_cast<const float4*>(epi + rr * 64 + (((c16 * 4 + q) ^ (rr & 3)) << 2));
    if constexpr (EF >= 0) {   // alpha == 1 (dispatcher); without a bias stage the sums go through untouched
      if constexpr ((EF & EF_BIAS) != 0) {
        v[q * 4 + 0] = x.x + bias16[q * 4 + 0]; v[q * 4 + 1] = x.y + bias16[q * 4 + 1];
        v[q * 4 + 2] = x.z + bias16[q * 4 + 2]; v[q * 4 + 3] = x.w + bias16[q * 4 + 3];
      } else {
        v[q * 4 + 0] = x.x; v[q * 4 + 1] = x.y; v[q * 4 + 2] = x.z; v[q * 4 + 3] = x.w;
      }
    } else {
      v[q * 4 + 0] = x.x * ep.alpha + bias16[q * 4 + 0]; v[q * 4 + 1] = x.y * ep.alpha + bias16[q * 4 + 1];
      v[q * 4 + 2] = x.z * ep.alpha + bias16[q * 4 + 2]; v[q * 4 + 3] = x.w * ep.alpha + bias16[q * 4 + 3];
    }
  }
  __builtin_amdgcn_wave_barrier();  // every lane has read before the next block overwrites the buffer
  const int row = row0 + rr, n = nw + c16 * 16;
  if (row < M && n < N) epi_piece16_v3<OutT, EF>(v, C, ldc, row, rowmap(row), n, N, ep);
}

template <typename OutT, typename RowMap, int EF = EF_GENERIC>
__device__ __forceinline__ void epilogue_v3(floatx4_t (&acc)[4][4], float* __restrict__ epi, OutT* __restrict__ C, int64_t ldc, int M,
                                            int N, int mw, int nw, const Epilogue& ep, const RowMap& rowmap, int lane) {
  if (EF >= 0 || ep.vec) {  // N % 8 == 0, 16-byte aligned rows; a 16-column piece may be cut to 8 at the right edge
    const int c16 = lane & 3;
    float bias16[16];
    if constexpr (EF >= 0 && (EF & EF_BIAS) == 0) {
#pragma unroll
      for (int q = 0; q < 16; ++q) bias16[q] = 0.f;
    } else {
      const int n = nw + c16 * 16;
      // unconditional loads (zero block without a bias) consumed unconditionally: a load left pending on some path
      // makes the compiler drain vmcnt(0) -- and with it the LDS-DMA prefetch -- inside the K loop
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float* bp = (ef_on<EF, EF_BIAS>(ep.bias != nullptr) && n + q * 4 < N) ? ep.bias + n + q * 4
                                                                                    : reinterpret_cast<const float*>(g_nst_zero16);
        const float4 x = *reinterpret_cast<const float4*>(bp);
        bias16[q * 4] = x.x; bias16[q * 4 + 1] = x.y; bias16[q * 4 + 2] = x.z; bias16[q * 4 + 3] = x.w;
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) asm volatile("" : "+v"(bias16[q]));
    }
    epi_block_v3<OutT, RowMap, EF>(acc[0], epi, C, ldc, M, N, mw, nw, bias16, ep, rowmap, lane);
    epi_block_v3<OutT, RowMap, EF>(acc[1], epi, C, ldc, M, N, mw + 16, nw, bias16, ep, rowmap, lane);
    epi_block_v3<OutT, RowMap, EF>(acc[2], epi, C, ldc, M, N, mw + 32, nw, bias16, ep, rowmap, lane);
    epi_block_v3<OutT, RowMap, EF>(acc[3], epi, C, ldc, M, N, mw + 48, nw, bias16, ep, rowmap, lane);
  } else if constexpr (EF < 0) {  // unaligned / odd-N outputs: one element at a time in a rolled loop
    const int g = lane >> 4, lc = lane & 15;
#pragma unroll 1
    for (int q = 0; q < 64; ++q) {
      const int i = q >> 4, j = (q >> 2) & 3, r = q & 3;
      const int row = mw + i * 16 + g * 4 + r, n = nw + j * 16 + lc;
      const float v = acc_pick(acc, q);
      if (row < M && n < N) epilogue_store<OutT>(ep, C, ldc, row, n, N, v, rowmap(row));
    }
  }
}


// Kernel arguments of the v3 kernels travel as ONE by-value struct, i.e. they sit at offset 0 of the kernarg segment.
// The K loop keeps only its own bookkeeping in SGPRs; everything that is needed once per unit (loaders for the DMA
// cursors, the whole Epilogue) is re-read from the kernarg segment through a laundered pointer at the point of use.
// Left to itself the compiler keeps all ~100 scalars live across the loop and spills SGPRs to VGPR lanes inside it.
#define NST_AS4 __attribute__((address_space(4)))
template <typename OutT, typename ALoader, typename BLoader, typename RowMap>
struct GemmArgs {
  ALoader la;
  BLoader lb;
  OutT* C;
  int64_t ldc;
  int M, N, K, tiles_n, ntiles, split, kt_per_split;
  int z_per_xcd;   // != 0 (needs split % 8 == 0, grid % 8 == 0): K slice z lives on XCD z % 8, see unit_of
  int reserved0;
  Epilogue ep;
  RowMap rowmap;
};
template <typename F>
__device__ __forceinline__ F kload(const NST_AS4 F* p) {  // dword-wise copy: stays in the constant address space (s_load)
  static_assert(sizeof(F) % 4 == 0, "kernarg structs are dword multiples");
  uint32_t w[sizeof(F) / 4];
  const NST_AS4 uint32_t* src = reinterpret_cast<const NST_AS4 uint32_t*>(p);
#pragma unroll
  for (unsigned i = 0; i < sizeof(F) / 4; ++i) w[i] = src[i];
  return __builtin_bit_cast(F, w);
}
template <typename A>
__device__ __forceinline__ const NST_AS4 A* launder(const NST_AS4 A* p) {
  asm volatile("" : "+s"(p));
  return p;
}

template <typename T, typename OutT, int AMODE, int BMODE, typename ALoader, typename BLoader, typename RowMap, bool CS,
          int EF = EF_GENERIC>
__device__ __forceinline__ void gemm_stream_v3(char* smem) {
  typedef GemmArgs<OutT, ALoader, BLoader, RowMap> Args;
  const NST_AS4 Args* ka = (const NST_AS4 Args*)__builtin_amdgcn_kernarg_segment_ptr();
  typedef SwzFrag<T, AMODE> RA;
  typedef SwzFrag<T, BMODE> RB;
  constexpr int BK = Tile<T>::BK;
  constexpr int KS = Mma<T>::KS;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  typedef __attribute__((address_space(3))) char* lds_char_ptr;
  const uint32_t smem_addr = (uint32_t)(uintptr_t)((lds_char_ptr)smem);
  const int ntiles = ka->ntiles, tiles_n = ka->tiles_n, kt_per_split = ka->kt_per_split;
  const int kt_total = (ka->K + BK - 1) / BK;
  const int units = ntiles * ka->split;
  const bool has_cs = CS && ka->ep.colsum_dst != nullptr;

  // unit -> tile origin and K range
  // Split-K units of one K slice read the same rows of both operands.  Workgroup ids go round-robin over the 8 XCDs, so
  // with z_per_xcd the slice index is tied to id % 8: all tiles of a slice run on ONE XCD and the re-reads of the
  // shared rows hit that XCD's L2 (otherwise every XCD fetches every row across the fabric: the conv2 weight gradient
  // moved 10.6 GB per launch that way, 7x its operands).
  const int zx = ka->z_per_xcd;
  struct Unit { int m0, n0, z, kt_first, kt_count; };
  auto unit_of = [&](int u) {
    Unit q;
    int tile;
    if (zx) {
      const int idx = u >> 3, sl = idx / ntiles;
      q.z = (u & 7) + 8 * sl;
      tile = idx - sl * ntiles;
    } else {
      q.z = u / ntiles;
      tile = xcd_remap(u - q.z * ntiles, ntiles);
    }
    const int tm = tile / tiles_n;
    q.m0 = tm * BM;
    q.n0 = (tile - tm * tiles_n) * BN;
    q.kt_first = q.z * kt_per_split;
    q.kt_count = kt_total - q.kt_first;
    if (q.kt_count > kt_per_split) q.kt_count = kt_per_split;
    return q;
  };

  floatx4_t acc[4][4], cs[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = floatx4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) cs[j] = floatx4_t{0.f, 0.f, 0.f, 0.f};
  const typename RA::Frag ones = ones_frag<T>();

  int iu = blockIdx.x;           // unit whose K steps are being issued
  if (iu >= units) return;
  int ikt = 0, ikt_count;        // next K step of iu to issue
  TileDma<T, AMODE, ALoader> da;
  TileDma<T, BMODE, BLoader> db;
  ALoader la_cur;                // loader copies the cursors point at (generic TileDma keeps a pointer to its loader)
  BLoader lb_cur;
  auto cursor_init = [&](int u) {
    const NST_AS4 Args* k2 = launder(ka);
    la_cur = kload(&k2->la);
    lb_cur = kload(&k2->lb);
    const Unit q = unit_of(u);
    ikt_count = q.kt_count;
    da.init(la_cur, q.m0, q.kt_first * BK, wave, lane);
    db.init(lb_cur, q.n0, q.kt_first * BK, wave, lane);
    da.begin(q.kt_count);
    db.begin(q.kt_count);
  };
  cursor_init(iu);
  int cu = iu;                   // unit being multiplied
  Unit cq = unit_of(cu);
  int ckt = 0;

  // the DMA of one K step in two halves (A tile, then B tile + the unit bookkeeping): the main loop puts the halves BEHIND the
  // fragment reads and between its two groups of 16 MFMAs -- issuing the eight DMA instructions back to back in front of the
  // step cost the wave 500-800 cycles with the matrix core idle (measured on the feed-forward kernel, r03_ffn_v2_ablation.log)
  auto issue_a = [&](int stage) {
    const uint32_t sa = smem_addr + stage * V2_STAGE_BYTES;
    da.next(sa + wave * 1024, sa, wave);
  };
  auto issue_b = [&](int stage) {
    const uint32_t sa = smem_addr + stage * V2_STAGE_BYTES;
    db.next(sa + BM * KBYTES + wave * 1024, sa + BM * KBYTES, wave);
    if (++ikt == ikt_count) {
      iu += gridDim.x;
      ikt = 0;
      if (iu < units) cursor_init(iu);
    }
  };
  auto issue_next = [&](int stage) {
    issue_a(stage);
    issue_b(stage);
  };

  issue_next(0);
  int stage = 0;
  while (true) {
    wait_vmcnt<0>();                 // the K step about to be multiplied has landed (and older stores have retired)
    __builtin_amdgcn_s_barrier();    // ... for every wave; every wave is also done reading the other stage
    asm volatile("" ::: "memory");
    if (iu < units) issue_next(stage ^ 1);
    const char* As = smem + stage * V2_STAGE_BYTES;
    const char* Bs = As + BM * KBYTES;
    const bool do_cs = has_cs && cq.m0 == 0 && wm == 0;  // wave-uniform
    if constexpr (BK / KS == 2) {
      // both halves of the K step are requested up front: the second set of fragments lands under the first 16 MFMAs
      typename RA::Frag a0[4], a1[4];
      typename RB::Frag b0[4], b1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b0[j] = RB::read(Bs, wn + j * 16, 0, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) a0[i] = RA::read(As, wm + i * 16, 0, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) b1[j] = RB::read(Bs, wn + j * 16, KS, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) a1[i] = RA::read(As, wm + i * 16, KS, lane);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = Mma<T>::run(a0[i], b0[j], acc[i][j]);
      if (CS && do_cs) {
#pragma unroll
        for (int j = 0; j < 4; ++j) cs[j] = Mma<T>::run(ones, b0[j], cs[j]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = Mma<T>::run(a1[i], b1[j], acc[i][j]);
      if (CS && do_cs) {
#pragma unroll
        for (int j = 0; j < 4; ++j) cs[j] = Mma<T>::run(ones, b1[j], cs[j]);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < BK; kk += KS) {
        typename RA::Frag a[4];
        typename RB::Frag b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = RB::read(Bs, wn + j * 16, kk, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = RA::read(As, wm + i * 16, kk, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i][j] = Mma<T>::run(a[i], b[j], acc[i][j]);
        if (CS && do_cs) {
#pragma unroll
          for (int j = 0; j < 4; ++j) cs[j] = Mma<T>::run(ones, b[j], cs[j]);
        }
      }
    }
    stage ^= 1;
    if (++ckt == cq.kt_count) {
      const NST_AS4 Args* k2 = launder(ka);
      Epilogue ep = kload(&k2->ep);
      if (ef_on<EF, EF_DROP>(ep.drop_thresh != 0)) ep.seed = seed_with_offset(ep.seed, ep.seed_dev);   // wave-uniform
      const RowMap rowmap = kload(&k2->rowmap);
      const int M = k2->M, N = k2->N;
      if (CS && do_cs && lane < 16) {  // every row of cs holds the column sums; lane = column within the 16-block
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int col = cq.n0 + wn + j * 16 + lane;
          if (col < N) {
            float* dst = ep.colsum_dst + (int64_t)cq.z * ep.colsum_zstride + col;
            *dst = ep.colsum_acc ? *dst + cs[j][0] : cs[j][0];
          }
        }
      }
      epilogue_v3<OutT, RowMap, EF>(acc, reinterpret_cast<float*>(smem + 2 * V2_STAGE_BYTES + wave * V3_EPI_BYTES_PER_WAVE),
                                k2->C + (int64_t)cq.z * ep.slab_stride, k2->ldc, M, N, cq.m0 + wm, cq.n0 + wn, ep, rowmap, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) cs[j] = floatx4_t{0.f, 0.f, 0.f, 0.f};
      cu += gridDim.x;
      if (cu >= units) break;
      cq = unit_of(cu);
      ckt = 0;
    }
  }
}

}  // namespace nstgemm
