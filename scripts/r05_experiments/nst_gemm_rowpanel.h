// Row-panel GEMM for reductions of exactly 256 (d_model of the Speech-Transformer): C[M, N] = epilogue(A[M, 256] . Bop)
//
// Why (round 5): the projections of a d_model = 256 layer -- q|k|v, output, their input gradients, the decoder's q / k|v / ffn
// products -- are four 64-deep K steps per 128 x 128 tile in the stream kernel (nst_gemm_core.h: gemm_stream_v3).  With ONE
// stage in flight per workgroup every step pays a whole load latency (~1.8 us under load) for 0.25 us of MFMAs: qkv forward
// 26 us against an HBM floor of 9 (profiles/r04_kernel_bench_vs_hipblaslt.txt).  Here a workgroup owns a PANEL of 128 rows for
// a range of column blocks and
//
//   * keeps the panel's A operand in REGISTERS for the whole unit (wave (r, h) holds rows [32 r, +32) x K = 256 as 16 MFMA
//     fragments = 64 VGPRs, loaded straight from global memory in fragment layout: A is read from HBM exactly once);
//   * streams the B operand (the weights: L2-resident, shared by every workgroup) through a ring of SIX 16 KB tile images
//     [128 columns x 64 k] by LDS-DMA with counted vmcnt waits -- five tiles (80 KB) in flight per CU instead of one
//     32 KB stage per workgroup, and a K step never waits for a load that was issued less than four steps earlier;
//     the fragments of a tile are read one step ahead of its MFMAs (two register sets);
//   * eight waves = 4 row groups x 2 column halves: a wave multiplies its 32 rows by 64 of the tile's 128 columns
//     (16 MFMAs 16x16x32 and 8 fragment reads per tile); two waves share a SIMD, so one wave's fragment reads run under the
//     other's MFMAs.  LDS reads (64 KB per tile over the CU's 128 B / clk) and MFMAs (512 clk per SIMD and tile) balance;
//   * the epilogue of a column block (two 16 x 64 pieces per wave through the wave-private scratch of the v3 epilogue, all
//     compile-time stages of nst_gemm_core.h) runs with the next block's tiles already in flight; the bias of the unit's
//     columns is staged in LDS once (any compiler-visible global load inside the loop would drain the ring: the compiler
//     does not see the DMA and waits for vmcnt(0)).
//
// Units: (panel, part) -- `nsplit` parts cut the column blocks of a panel when there are fewer panels than CUs (decoder:
// 75 panels).  One workgroup per CU (128 KB of LDS), grid = units.
#pragma once
#include "nst_gemm_core.h"
#include "nst_gemm256.h"

namespace nstgemm {

constexpr int RP_THREADS = 512;
constexpr int RP_BM = 128;
constexpr int RP_NST = 6;                                   // ring of B tile images
constexpr int RP_TILE_BYTES = BM * KBYTES;                  // 16 KB
constexpr int RP_MAX_NB = 24;                               // column blocks per unit (the bias of a unit's columns lives in LDS)
constexpr int RP_EPI_OFF = RP_NST * RP_TILE_BYTES;          // wave-private epilogue scratch, 8 x 4 KB
constexpr int RP_BIAS_OFF = RP_EPI_OFF + 8 * V3_EPI_BYTES_PER_WAVE;
constexpr int RP_LDS_BYTES = RP_BIAS_OFF + RP_MAX_NB * BN * 4;   // 96 KB + 32 KB + 12 KB
constexpr int RP_K = 256, RP_KT = RP_K / 64;

template <typename OutT>
struct RpArgs {
  DenseLoader<bf16_t> la;   // A: RC (row-major, reduction contiguous), outer_limit = M, contig_limit = K
  DenseLoader<bf16_t> lb;   // B: RC (element (n, k) at base[n * ld + k]) or OC (element (k, n) at base[k * ld + n])
  OutT* C;
  int64_t ldc;
  int M, N;
  int nblocks;              // ceil(N / 128)
  int nsplit, nb_per;       // a panel's column blocks are cut into nsplit parts of nb_per blocks
  int units;                // panels * nsplit
  int dbg;                  // timing ablations (temporary): 1 no DMA in the loop, 2 no MFMAs, 4 no fragment reads, 8 no epilogue
  Epilogue ep;
};

// LDS-DMA of one B tile image by eight waves (two 1 KB pieces per wave); same source swizzle as dma_tile (4 waves x 4 pieces)
template <int MODE>
__device__ __forceinline__ void rp_dma_tile(const DenseLoader<bf16_t>& ld, int n0, int k0, uint32_t tile_lds_addr, int wave, int lane) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int cbase = (s * 8 + wave) * 64;   // first chunk of this wave instruction (wave uniform)
    const int c = cbase + lane;
    const bf16_t* p;
    if (MODE == MODE_RC) {
      const int row = c >> 3, slot = c & 7;
      const int kchunk = slot ^ ((row >> 1) & 7);
      p = ld.ptr(n0 + row, k0 + kchunk * 8);
    } else {
      const int r = c >> 4, c16 = c & 15;
      const int g = (r & 3) | (((r >> 3) & 1) << 2);
      p = ld.ptr(k0 + r, n0 + (c16 ^ (g << 1)) * 8);
    }
    const void* src = p ? (const void*)p : (const void*)g_nst_zero16;
    glds16(src, __builtin_amdgcn_readfirstlane(tile_lds_addr + (uint32_t)cbase * 16u));
  }
}

// bias_s: this lane's 16 bias values (columns nw + (lane & 3) * 16 ..) in LDS, staged once per unit -- a global load here would
// make the compiler drain vmcnt(0), i.e. wait for the five B tiles in flight, in every epilogue
template <typename OutT, int EF>
__device__ __forceinline__ void rp_epilogue(floatx4_t (&acc)[2][4], float* __restrict__ epi, OutT* __restrict__ C, int64_t ldc, int M,
                                            int N, int mw, int nw, const float* __restrict__ bias_s, const Epilogue& ep, int lane) {
  float bias16[16];
  if constexpr ((EF & EF_BIAS) == 0) {
#pragma unroll
    for (int q = 0; q < 16; ++q) bias16[q] = 0.f;
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 x = *reinterpret_cast<const float4*>(bias_s + q * 4);
      bias16[q * 4] = x.x; bias16[q * 4 + 1] = x.y; bias16[q * 4 + 2] = x.z; bias16[q * 4 + 3] = x.w;
    }
  }
  const IdentityRowMap rowmap;
  epi_block_v3<OutT, IdentityRowMap, EF>(acc[0], epi, C, ldc, M, N, mw, nw, bias16, ep, rowmap, lane);
  epi_block_v3<OutT, IdentityRowMap, EF>(acc[1], epi, C, ldc, M, N, mw + 16, nw, bias16, ep, rowmap, lane);
}

template <typename OutT, int BMODE, int EF>
__device__ __forceinline__ void gemm_rowpanel_block(char* smem) {
  typedef bf16_t T;
  typedef RpArgs<OutT> Args;
  const NST_AS4 Args* ka = (const NST_AS4 Args*)__builtin_amdgcn_kernarg_segment_ptr();
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wh = wave & 1;    // row group (32 rows), column half (64 of the tile's 128 columns)
  typedef __attribute__((address_space(3))) char* lds_char_ptr;
  const uint32_t smem_addr = (uint32_t)(uintptr_t)((lds_char_ptr)smem);

  const int unit = blockIdx.x;
  const int dbg = ka->dbg;
  const int nsplit = ka->nsplit, nb_per = ka->nb_per, nblocks = ka->nblocks;
  const int panel = unit / nsplit, part = unit - panel * nsplit;
  const int nb0 = part * nb_per;
  int nb_count = nblocks - nb0;
  if (nb_count > nb_per) nb_count = nb_per;
  if (nb_count <= 0) return;                  // (workgroup-uniform)
  const int m0 = panel * RP_BM;
  const int S = nb_count * RP_KT;             // B tiles of this unit

  // ---- the bias of the unit's columns -> LDS (zeros beyond N and without a bias); first read behind the loop's barriers
  float* bias_all = reinterpret_cast<float*>(smem + RP_BIAS_OFF);
  if constexpr ((EF & EF_BIAS) != 0) {
    typedef const __attribute__((address_space(1))) float* gfloat_ptr;
    const gfloat_ptr bias = (gfloat_ptr)(uintptr_t)ka->ep.bias;
    const int N = ka->N;
    for (int i = tid; i < nb_count * BN; i += RP_THREADS) {
      const int n = nb0 * BN + i;
      bias_all[i] = n < N ? bias[n] : 0.f;
    }
  }
  // ---- the panel's A operand: 2 row blocks x 8 K steps of 32, in MFMA fragment layout, straight from global memory
  bf16x8_t af[2][RP_KT * 2];
  {
    typedef const __attribute__((address_space(1))) bf16x8_t* gfrag_ptr;
    const DenseLoader<T> la = kload(&ka->la);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int ks = 0; ks < RP_KT * 2; ++ks) {
        const T* p = la.ptr(m0 + wr * 32 + rb * 16 + (lane & 15), ks * 32 + (lane >> 4) * 8);
        const void* src = p ? (const void*)p : (const void*)g_nst_zero16;
        af[rb][ks] = *(gfrag_ptr)(uintptr_t)src;
      }
  }
  // ---- ring prologue: the first NST tiles (tile t lives in stage t % NST)
  const DenseLoader<T> lb = kload(&launder(ka)->lb);
  auto issue = [&](int s) {
    const int jb = s >> 2, kc = s & 3;
    rp_dma_tile<BMODE>(lb, (nb0 + jb) * BN, kc * 64, smem_addr + (uint32_t)(s % RP_NST) * RP_TILE_BYTES, wave, lane);
  };
#pragma unroll 1
  for (int s = 0; s < RP_NST && s < S; ++s) issue(s);
  // the A fragments are "used" HERE: the compiler's wait for them (a vmcnt(0): it does not see the DMA) then sits in front of
  // the loop, once per unit, instead of in front of the first MFMA of every column block.  It also retires the prologue tiles.
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int ks = 0; ks < RP_KT * 2; ++ks) asm volatile("" : "+v"(af[rb][ks]));
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // Fragments are read ONE TILE AHEAD (two register sets, static indices: the K loop is unrolled over the 4 tiles of a column
  // block): a step issues the reads of tile s + 1 and then multiplies tile s, so LDS latency and bandwidth run under the MFMAs
  // of the same wave.  Order of a step:
  //   lgkmcnt(0)      this wave's reads of tile s have landed -- BEFORE the barrier, so that after it anybody may restage
  //   vmcnt(..)       this wave's pieces of tile s + 1 have landed             the stage of tile s
  //   barrier         ... in every wave
  //   DMA             tile s + NST -> stage s % NST (five tiles stay in flight)
  //   reads           tile s + 1 -> the other register set
  //   16 MFMAs        tile s
  float* epi = reinterpret_cast<float*>(smem + RP_EPI_OFF + wave * V3_EPI_BYTES_PER_WAVE);
  bf16x8_t b0[4][2], b1[4][2];
  // (Frag256: the fragment readers of the 256 x 256 kernel -- same 16 KB image, the lane-dependent part of the address
  // computed once; the generic SwzFrag readers re-derive it per fragment and cost this loop 50 more registers)
  Frag256<BMODE> fb;
  fb.init(lane);
  auto read_tile = [&](int t, bf16x8_t (&b)[4][2]) {
    const char* Bs = smem + (t % RP_NST) * RP_TILE_BYTES;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) b[cb][ks] = fb.read(Bs, wh * 4 + cb, ks);
  };
  read_tile(0, b0);
#pragma unroll 1
  for (int jb = 0; jb < nb_count; ++jb) {
    floatx4_t acc[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[rb][cb] = floatx4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < RP_KT; ++kc) {
      const int s = jb * RP_KT + kc;
      __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0)
      if (s + 1 < S) {
        // tile s + 1 has landed once no more than the tiles issued behind it (2 wave instructions each) are outstanding
        const int behind = S - 2 - s;
        if (behind >= RP_NST - 2) wait_vmcnt<2 * (RP_NST - 2)>();
        else if (behind == 3) wait_vmcnt<6>();
        else if (behind == 2) wait_vmcnt<4>();
        else if (behind == 1) wait_vmcnt<2>();
        else wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (s + RP_NST < S && !(dbg & 1)) issue(s + RP_NST);
      if (s + 1 < S && !(dbg & 4)) {
        if ((kc & 1) == 0) read_tile(s + 1, b1);
        else read_tile(s + 1, b0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (dbg & 2) {
      } else if ((kc & 1) == 0) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) acc[rb][cb] = Mma<T>::run(af[rb][kc * 2 + ks], b0[cb][ks], acc[rb][cb]);
      } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) acc[rb][cb] = Mma<T>::run(af[rb][kc * 2 + ks], b1[cb][ks], acc[rb][cb]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!(dbg & 8)) {
      const NST_AS4 Args* k2 = launder(ka);
      Epilogue ep = kload(&k2->ep);
      if ((EF & EF_DROP) != 0) ep.seed = seed_with_offset(ep.seed, ep.seed_dev);   // wave-uniform
      rp_epilogue<OutT, EF>(acc, epi, k2->C, k2->ldc, k2->M, k2->N, m0 + wr * 32, (nb0 + jb) * BN + wh * 64,
                            bias_all + jb * BN + wh * 64 + (lane & 3) * 16, ep, lane);
    }
  }
}

}  // namespace nstgemm
