// Whole-row GEMMs for d_model = 256 (bf16): out[M, 256] = A[M, K] . Bop, every workgroup owns BM COMPLETE output rows, so the
// stages of the reference's pre-norm wrapper that need a whole row run in the epilogue of the product that makes the row
// (neurst/layers/common_layers.py:73-85: inputs + dropout(layer(LayerNorm(inputs)))):
//
//   forward  (nst_gemm_add_layernorm_fwd): the LAST product of a sub-layer (attention output projection
//            multi_head_attention.py:219, feed-forward dense2 common_layers.py:159)
//               x_new = x + dropout(A . W + b)         the float32 residual stream, written once
//               y     = LayerNorm(x_new; gamma, beta)  the NEXT wrapper's normalised input (bf16), mean / rstd saved
//            instead of GEMM -> delta (bf16) -> nst_add_layernorm_fwd: delta never travels (29 MB per encoder sub-layer at the
//            benchmark shape) and one launch disappears;
//   backward (nst_gemm_layernorm_bwd): the FIRST product of the sub-layer's backward chain (qkv / q projection, dense1:
//            g = dZ . W^T is the gradient w.r.t. the LayerNorm output)
//               dx = LayerNorm'(g; x, mean, rstd, gamma) + dres      (+ dz = dropout-backward copy for the next sub-layer)
//               partial sums of dgamma / dbeta per workgroup for the deferred finalize (nst_ln_finalize_multi)
//            instead of GEMM -> g (bf16) -> nst_layernorm_bwd_mixed;
//   rowdot   (nst_gemm_rowdot256): the attention output projection's input gradient d(context) = dZ . Wo^T together with
//            delta[b, h, t] = sum over the head's 64 columns of d(context) o context for the attention backward.
//
// Structure: 256 threads = 4 waves, wave w owns output columns [64 w, 64 w + 64) of all BM rows (BM = 64 or 32; 64 rows keep
// two workgroups on a CU: 2 x 80 KB of LDS).  K is walked in steps of 64 through two LDS stages filled by LDS-DMA
// (global_load_lds_dwordx4; the stage images are the swizzled, unpadded ones of nst_gemm_core.h, so its fragment readers are
// used unchanged: A [BM][64 k], the weights as two images of 128 output columns each).  The MFMAs run "transposed" (weights as
// the A operand), so a lane ends up with 4 consecutive columns of one row and the accumulators go to an LDS tile
// [BM][256] f32 with 16-byte writes.  Behind one barrier the waves change roles: wave w owns rows [w BM/4, (w+1) BM/4), 32
// lanes per row, 8 consecutive columns per lane -- the access pattern of the wide LayerNorm kernels (nst_norm.hip), whole
// 1 KB / 512-byte row segments per half wave, row reductions inside the half wave by DPP + lane-row swaps.
#include "nst_gemm_core.h"
#include "nst_rowphase.h"

#include <stdlib.h>

#include <utility>

using namespace nstgemm;
using namespace rowphase;

namespace {

enum { EPI_LN_FWD = 0, EPI_LN_BWD = 1, EPI_ROWDOT = 2 };

struct RowArgs {
  const bf16_t* A;       // [M, K], row stride lda
  const bf16_t* W;       // OC: [K, 256] (ldb)   RC: [256, K] (ldb)
  int64_t lda, ldb;
  int M, K;
  uint64_t seed;
  const uint64_t* seed_dev;
  RowEpi e;              // the row phase's operands (nst_rowphase.h); e.y is C for the rowdot / plain forms
  // rowdot
  const bf16_t* rd_src;  // [M, 256]
  float* rd_dst;         // [(b * 4 + h) * T + t]
  int rd_T;
  int reserved1;
};

template <int BM, int NST, int NW = 4>
struct RowCfg {
  static constexpr int RG = NW / 4;                     // row groups of waves (4 waves side by side cover the 256 columns)
  static constexpr int GROWS = BM / RG;                 // rows a wave multiplies
  static constexpr int IMG_ROWS = (BM + 31) / 32 * 32;  // rows of the A image (BM = 48: 64, the last 16 repeat row BM - 1)
  static constexpr int A_BYTES = IMG_ROWS * 128;
  static constexpr int STAGE = A_BYTES + 2 * 16384;
  static constexpr int A_IPW = IMG_ROWS / 8 / NW;       // DMA instructions per wave and K step for the A image
  static constexpr int B_IPW = 32 / NW;                 // ... for the two weight images
  static constexpr int IPW = A_IPW + B_IPW;             // ... for the whole stage
  static constexpr int MI = GROWS / 16;
  static constexpr int TILE_BYTES = BM * TILE_LD * 4;
  static constexpr int RED_BYTES = NW * 2 * RN * 4;
  static constexpr int LDS = (NST * STAGE > TILE_BYTES + RED_BYTES) ? NST * STAGE : TILE_BYTES + RED_BYTES;
  static_assert(NST >= 2 && NST <= 4 && (NST - 2) * IPW < 64 && GROWS % 16 == 0 && (BM / NW) % 2 == 0 && (NW == 4 || NW == 8) &&
                    (IMG_ROWS / 8) % NW == 0, "stage ring depth / tile height");
  static_assert(LDS <= 160 * 1024, "LDS of a CU");
  static constexpr int RPW = BM / NW;                   // rows a wave owns in the row phase
  static constexpr int PASSES = RPW / 2;                // two rows (32 lanes each) per pass
};

// one LDS-DMA instruction: 64 lanes x 16 bytes from sbase + voff (per lane) -> LDS [lds_addr_uniform + lane * 16]
__device__ __forceinline__ void rg_glds(const void* sbase, uint32_t voff, uint32_t lds_addr_uniform) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 1\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_addr_uniform)
      : "memory");
}

// Behind the K loop of either kernel: accumulators -> LDS tile [BM][256] f32 (the stages' memory), then the row phase.
template <int BM, int NW, int MI, int EPI, int NP>
__device__ __forceinline__ void rg_finish(char* smem, floatx4_t (&acc)[MI][4], const RowArgs& a, const float (&xpre)[NP][8], int m0,
                                          int M, int tid, int lane, int wave, int wrow, int wc) {
  constexpr int RPW = BM / NW, PASSES = RPW / 2, TILE_BYTES = BM * TILE_LD * 4;
  const int sub = lane >> 5, li = lane & 31, col = li * 8;
  __builtin_amdgcn_s_barrier();      // every wave is done reading the stages: the tile reuses their LDS
  asm volatile("" ::: "memory");

  // ---------------------------------------------------------------- accumulators -> LDS tile [BM][256] f32
  float* tile = reinterpret_cast<float*>(smem);
  {
    const int ml = lane & 15, nq = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const floatx4_t v = acc[i][j];
        *reinterpret_cast<float4*>(tile + (wrow + i * 16 + ml) * TILE_LD + wc * 64 + j * 16 + nq) = make_float4(v[0], v[1], v[2], v[3]);
      }
  }
  __syncthreads();

  // ---------------------------------------------------------------- row phase: wave w owns rows [w RPW, +RPW), 32 lanes per row
  uint64_t seed = a.seed;
  if ((EPI == EPI_LN_FWD || EPI == EPI_LN_BWD) && a.e.drop_thresh) seed = seed_with_offset(a.seed, a.seed_dev);   // wave-uniform

  if constexpr (EPI == EPI_LN_FWD) {
    ln_fwd<RPW, true>(tile, a.e, seed, m0, M, wave, lane, xpre);
  } else if constexpr (EPI == EPI_LN_BWD) {
    ln_bwd<NW, RPW, true>(tile, reinterpret_cast<float*>(smem + TILE_BYTES), a.e, seed, m0, M, tid, wave, lane, xpre);
  } else {
    // EPI_ROWDOT: C = acc as bf16; per head (64 columns = 8 lanes) the sum of C (as stored) o src
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int rl = wave * RPW + p * 2 + sub;
      const int rowg = m0 + rl;
      const bool ok = rowg < M;
      const int rc = ok ? rowg : M - 1;
      const float4 t0 = *reinterpret_cast<const float4*>(tile + rl * TILE_LD + col);
      const float4 t1 = *reinterpret_cast<const float4*>(tile + rl * TILE_LD + col + 4);
      float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
      if (ok) rg_store8_bf16(a.e.y + (int64_t)rowg * RN + col, v);
      float sv[8];
      rg_load8_bf16(a.rd_src + (int64_t)rc * RN + col, sv);
      float dot = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) dot = fmaf(bf16_to_f32(f32_to_bf16(v[j])), sv[j], dot);
      dot = dpp_add(dot, 0); dot = dpp_add(dot, 1); dot = dpp_add(dot, 2);   // the 8 lanes of a head
      if (ok && (li & 7) == 0) {
        const int b = rowg / a.rd_T, t = rowg - b * a.rd_T;
        a.rd_dst[((int64_t)b * 4 + (li >> 3)) * a.rd_T + t] = dot;
      }
    }
  }
}

template <int BM, int NST, int NW, int BMODE, int EPI>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) rowgemm_kernel(RowArgs a) {
  typedef RowCfg<BM, NST, NW> C;
  typedef SwzFrag<bf16_t, MODE_RC> RA;
  typedef SwzFrag<bf16_t, BMODE> RB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) char* lds_char_ptr;
  const uint32_t smem_addr = (uint32_t)(uintptr_t)((lds_char_ptr)smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave & 3, wrow = (wave >> 2) * C::GROWS;   // this wave multiplies rows [wrow, +GROWS) x columns [64 wc, +64)
  const int M = a.M;
  const int m0 = blockIdx.x * BM;
  const int nk = a.K >> 6;

  // ---------------------------------------------------------------- DMA source offsets (per lane, constant over K)
  uint32_t voffA[C::A_IPW], voffB[C::B_IPW];
#pragma unroll
  for (int s = 0; s < C::A_IPW; ++s) {
    const int c = (wave * C::A_IPW + s) * 64 + lane;
    const int row = c >> 3, slot = c & 7;
    const int kch = slot ^ ((row >> 1) & 7);
    int rg = m0 + (row < BM ? row : BM - 1);  // (image rows past the tile: any valid row, never multiplied)
    rg = rg < M ? rg : M - 1;                 // rows past the end read the last row (their results are never stored)
    voffA[s] = (uint32_t)((int64_t)(rg - m0) * a.lda * 2 + kch * 16);
  }
#pragma unroll
  for (int s = 0; s < C::B_IPW; ++s) {
    const int t = wave * C::B_IPW + s;        // 0 .. 31: image t >> 4, 1 KB piece t & 15 of it
    const int h = t >> 4, c = (t & 15) * 64 + lane;
    if (BMODE == MODE_RC) {
      const int row = c >> 3, slot = c & 7;
      const int kch = slot ^ ((row >> 1) & 7);
      voffB[s] = (uint32_t)((int64_t)(128 * h + row) * a.ldb * 2 + kch * 16);
    } else {
      const int r = c >> 4, c16 = c & 15;
      const int g = (r & 3) | (((r >> 3) & 1) << 2);
      voffB[s] = (uint32_t)((int64_t)r * a.ldb * 2 + (128 * h + (c16 ^ (g << 1)) * 8) * 2);
    }
  }
  const char* baseA = reinterpret_cast<const char*>(a.A) + ((int64_t)m0 * a.lda) * 2;
  const char* baseB = reinterpret_cast<const char*>(a.W);
  const int64_t stepB = BMODE == MODE_RC ? 128 : (int64_t)64 * a.ldb * 2;
  auto issue = [&](int kt, int stage) {
    const uint32_t sa = smem_addr + (uint32_t)stage * C::STAGE;
    const char* pa = baseA + (int64_t)kt * 128;
    const char* pb = baseB + (int64_t)kt * stepB;
#pragma unroll
    for (int s = 0; s < C::A_IPW; ++s) rg_glds(pa, voffA[s], sa + (uint32_t)(wave * C::A_IPW + s) * 1024u);
#pragma unroll
    for (int s = 0; s < C::B_IPW; ++s) rg_glds(pb, voffB[s], sa + C::A_BYTES + (uint32_t)(wave * C::B_IPW + s) * 1024u);
  };
  // The f32 rows the row phase adds (forward) / normalises again (backward) are requested FIRST, in the row phase's own lane
  // layout: their HBM latency and transfer then run under the whole K loop instead of behind it (vmcnt retires in issue order,
  // so the first K step waits for them once -- every later wait finds them landed).
  const int sub = lane >> 5, li = lane & 31, col = li * 8;
  float xpre[(EPI == EPI_LN_FWD || EPI == EPI_LN_BWD) ? C::PASSES : 1][8];
  if constexpr (EPI == EPI_LN_FWD || EPI == EPI_LN_BWD) {
#pragma unroll
    for (int p = 0; p < C::PASSES; ++p) {
      int rowg = m0 + wave * C::RPW + p * 2 + sub;
      rowg = rowg < M ? rowg : M - 1;
      rg_load8_f32(a.e.x + (int64_t)rowg * RN + col, xpre[p]);
    }
  }
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) issue(s, s);

  floatx4_t acc[C::MI][4];
#pragma unroll
  for (int i = 0; i < C::MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = floatx4_t{0.f, 0.f, 0.f, 0.f};

  // ---------------------------------------------------------------- main loop: ring of NST stages, NST - 1 K steps in flight
  // One K step = a stream of 8 MI MFMA positions (two halves of 32 k: column block j outer, row block i inner).  The IPW DMA
  // instructions that refill the ring are spread over the positions, ONE behind an MFMA at a time: issued back to back in
  // front of the step they keep the wave out of the matrix core for their whole issue time (call 3 of round 6: the K = 768 /
  // 2048 products ran at 36 - 49 GB/s of DMA per CU where the same ring without MFMAs moves 70 - 117).
  const int bimg = (wc >> 1) * 16384, wnl = (wc & 1) * 64;
  int stage = 0, stage_in = NST - 1;     // stage being multiplied / stage the next issue fills
  constexpr int NPOS = 8 * C::MI;
  auto kstep = [&](auto has_next_tag, int kt) {
    constexpr bool HAS_NEXT = decltype(has_next_tag)::value;
    // K step kt has landed; the (up to NST - 2) younger steps stay in flight: a COUNTED wait, never a drain
    const int ahead = (nk - 1 - kt) < (NST - 2) ? (nk - 1 - kt) : (NST - 2);
    if (NST == 2 || ahead <= 0) wait_vmcnt<0>();
    else if (NST == 3 || ahead == 1) wait_vmcnt<C::IPW>();
    else wait_vmcnt<(NST > 3 ? 2 : 1) * C::IPW>();
    __builtin_amdgcn_s_barrier();      // ... for every wave, and every wave is done reading the stage the next issue fills
    asm volatile("" ::: "memory");
    const uint32_t sa = smem_addr + (uint32_t)stage_in * C::STAGE;
    const char* pa = baseA + (int64_t)(kt + NST - 1) * 128;
    const char* pb = baseB + (int64_t)(kt + NST - 1) * stepB;
    stage_in = stage_in + 1 == NST ? 0 : stage_in + 1;
    const char* As = smem + stage * C::STAGE;
    stage = stage + 1 == NST ? 0 : stage + 1;
    const char* Bs = As + C::A_BYTES + bimg;
    typename RA::Frag af[2][C::MI];
    typename RB::Frag bf[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bf[0][j] = RB::read(Bs, wnl + j * 16, 0, lane);
#pragma unroll
    for (int i = 0; i < C::MI; ++i) af[0][i] = RA::read(As, wrow + i * 16, 0, lane);
    [&]<int... PP>(std::integer_sequence<int, PP...>) {
      ([&] {
        constexpr int P = PP;
        if constexpr (P == 1) {   // the second half's fragments: requested behind the first MFMA, they land under the first half
#pragma unroll
          for (int j = 0; j < 4; ++j) bf[1][j] = RB::read(Bs, wnl + j * 16, 32, lane);
#pragma unroll
          for (int i = 0; i < C::MI; ++i) af[1][i] = RA::read(As, wrow + i * 16, 32, lane);
        }
        constexpr int half = P / (4 * C::MI), j = (P % (4 * C::MI)) / C::MI, i = P % C::MI;
        constexpr int d = (P * C::IPW) / NPOS;
        constexpr bool DMA_HERE = HAS_NEXT && (P == 0 || d != ((P - 1) * C::IPW) / NPOS);
        // (the DMA is inline asm: the scheduler would otherwise sink all of them behind the step's MFMAs -- seen in the ISA)
        if constexpr (DMA_HERE) __builtin_amdgcn_sched_barrier(0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[half][j], af[half][i], acc[i][j], 0, 0, 0);   // D[n][m]: weights are the A operand
        if constexpr (DMA_HERE) {
          if constexpr (d < C::A_IPW) rg_glds(pa, voffA[d], sa + (uint32_t)(wave * C::A_IPW + d) * 1024u);
          else rg_glds(pb, voffB[d - C::A_IPW], sa + C::A_BYTES + (uint32_t)(wave * C::B_IPW + (d - C::A_IPW)) * 1024u);
          __builtin_amdgcn_sched_barrier(0);
        }
      }(), ...);
    }(std::make_integer_sequence<int, NPOS>());
  };
  {
    int kt = 0;
    for (; kt + NST - 1 < nk; ++kt) kstep(std::true_type(), kt);
    for (; kt < nk; ++kt) kstep(std::false_type(), kt);
  }
  rg_finish<BM, NW, C::MI, EPI>(smem, acc, a, xpre, m0, M, tid, lane, wave, wrow, wc);
}

template <typename KernelT>
void rg_allow_lds(KernelT kernel) {
  static thread_local const void* done[32];
  static thread_local int ndone = 0;
  for (int i = 0; i < ndone; ++i)
    if (done[i] == (const void*)kernel) return;
  (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (ndone < 32) done[ndone++] = (const void*)kernel;
}

// (rows per workgroup, stages): 64 rows while that still gives every CU a workgroup, else 32; two stages (two workgroups per CU)
// for the short reductions, a deeper ring (one workgroup per CU, more bytes in flight) for the long ones.
// NST_ROWGEMM_CFG="<rows>,<stages>" overrides (benchmarks).
struct RgCfg { int bm, nst; };
RgCfg rg_pick(int64_t M, int K) {
  static int forced_bm = -1, forced_nst = 0;
  if (forced_bm < 0) {
    forced_bm = 0;
    const char* e = getenv("NST_ROWGEMM_CFG");
    if (e) sscanf(e, "%d,%d", &forced_bm, &forced_nst);
  }
  (void)K;
  RgCfg c;
  // every CU gets a workgroup of 64 rows: two workgroups per CU, two stages each (deeper rings at one workgroup per CU lost
  // for every shape: call 3 of round 6).  Fewer rows than that (the decoder's 9 600): the launch is a chain of L2 round trips
  // per workgroup, so ONE workgroup per CU with the whole LDS as a four-stage ring (three K steps in flight) and 48 rows, if
  // that is a single round of workgroups; 32 rows x two stages otherwise.
  if (M >= 64 * 224) { c.bm = 64; c.nst = 2; }
  else if ((M + 47) / 48 <= 256 && M > 32 * 64) { c.bm = 48; c.nst = 4; }
  else { c.bm = 32; c.nst = 2; }
  if (forced_bm == 64 && forced_nst == 2) { c.bm = 64; c.nst = 2; }
  if (forced_bm == 32 && forced_nst == 2) { c.bm = 32; c.nst = 2; }
  if (forced_bm == 48 && forced_nst == 4) { c.bm = 48; c.nst = 4; }
  return c;
}

template <int BM, int NST, int NW, int BMODE, int EPI>
void rg_launch_one(const RowArgs& a, hipStream_t st, int* nblocks_out) {
  auto k = rowgemm_kernel<BM, NST, NW, BMODE, EPI>;
  rg_allow_lds(k);
  const int nb = (a.M + BM - 1) / BM;
  k<<<nb, 64 * NW, RowCfg<BM, NST, NW>::LDS, st>>>(a);
  if (nblocks_out) *nblocks_out = nb;
}

template <int BMODE, int EPI>
int rg_launch(const RowArgs& a, hipStream_t st, int* nblocks_out) {
  const RgCfg c = rg_pick(a.M, a.K);
  if (c.bm == 64) {
    rg_launch_one<64, 2, 4, BMODE, EPI>(a, st, nblocks_out);
  } else if (c.bm == 48) {
    rg_launch_one<48, 4, 4, BMODE, EPI>(a, st, nblocks_out);
  } else {
    rg_launch_one<32, 2, 4, BMODE, EPI>(a, st, nblocks_out);
  }
  return NST_OK;
}

int rg_check_common(const NstRowGemmDesc* d, const void* A, const void* W, const char* what) {
  NST_CHECK_ARG(d && A && W, "%s: null pointer", what);
  NST_CHECK_ARG(d->rows > 0 && d->rows < (1 << 30), "%s: rows=%lld", what, (long long)d->rows);
  NST_CHECK_ARG(nst_rowgemm_supported(d->n, d->k, d->dtype), "%s: unsupported shape n=%d k=%d dtype=%d (n = 256, k %% 64 == 0, bf16)",
                what, d->n, d->k, d->dtype);
  NST_CHECK_ARG(d->lda >= d->k && d->lda % 8 == 0 && nst_aligned16(A), "%s: A needs 16-byte aligned rows (lda=%lld)", what, (long long)d->lda);
  const int64_t ldb_min = d->trans_b ? d->k : d->n;
  NST_CHECK_ARG(d->ldb >= ldb_min && d->ldb % 8 == 0 && nst_aligned16(W), "%s: W needs 16-byte aligned rows (ldb=%lld)", what, (long long)d->ldb);
  NST_CHECK_ARG(d->lda * 2 * 64 < (1ll << 31) && d->ldb * 2 * 256 < (1ll << 31), "%s: leading dimensions exceed the 32-bit DMA offsets", what);
  return NST_OK;
}

void rg_fill(RowArgs& a, const NstRowGemmDesc* d, const void* A, const void* W) {
  memset(&a, 0, sizeof(a));
  a.A = (const bf16_t*)A; a.W = (const bf16_t*)W;
  a.lda = d->lda; a.ldb = d->ldb;
  a.M = (int)d->rows; a.K = d->k;
}

}  // namespace

extern "C" int nst_rowgemm_supported(int n, int k, int dtype) {
  return (dtype == NST_BF16 && n == RN && k >= 64 && k % 64 == 0 && k <= 16384) ? 1 : 0;
}

extern "C" int nst_gemm_add_layernorm_fwd(const NstRowGemmDesc* d, const void* A, const void* W, const float* bias, const float* x,
                                          float* x_out, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                          void* stream) {
  if (d && d->rows == 0) return NST_OK;
  int rc = rg_check_common(d, A, W, "gemm_add_layernorm_fwd");
  if (rc != NST_OK) return rc;
  NST_CHECK_ARG(x && gamma && beta && y && mean && rstd, "gemm_add_layernorm_fwd: null pointer");
  if (d->trans_b) {   // (only the orientations a training step issues are instantiated: forward products read the kernel as stored)
    nst_set_error("gemm_add_layernorm_fwd: trans_b = 1 is not built (a forward product reads W [k, 256])");
    return NST_ERR_UNSUPPORTED;
  }
  NST_CHECK_ARG(nst_aligned16(x) && nst_aligned16(y) && nst_aligned16(gamma) && nst_aligned16(beta) && (!x_out || nst_aligned16(x_out)) &&
                    (!bias || nst_aligned16(bias)),
                "gemm_add_layernorm_fwd: operands must be 16-byte aligned");
  NST_CHECK_ARG(d->dropout_p >= 0.f && d->dropout_p < 1.f, "gemm_add_layernorm_fwd: dropout_p=%f", d->dropout_p);
  RowArgs a;
  rg_fill(a, d, A, W);
  a.e.bias = bias; a.e.x = x; a.e.x_out = x_out; a.e.gamma = gamma; a.e.beta = beta; a.e.eps = d->eps;
  a.e.y = (bf16_t*)y; a.e.mean = mean; a.e.rstd = rstd;
  nst_dropout_params16(d->dropout_p, &a.e.drop_thresh, &a.e.drop_inv_keep);
  a.seed = d->seed; a.e.stream_id = d->stream_id;
  a.seed_dev = nst_seed_offset_devptr();
  if (!a.seed_dev) return NST_ERR_LAUNCH;
  rg_launch<MODE_OC, EPI_LN_FWD>(a, (hipStream_t)stream, nullptr);
  NST_CHECK_LAUNCH("gemm_add_layernorm_fwd");
  return NST_OK;
}

extern "C" int nst_gemm_layernorm_bwd(const NstRowGemmDesc* d, const void* A, const void* W, const float* x, const float* gamma,
                                      const float* mean, const float* rstd, const void* dres, void* dx, void* dz, float* dgamma,
                                      float* dbeta, int accumulate, void* workspace, int64_t workspace_bytes,
                                      NstLnFinalizeJob* job_out, void* stream) {
  if (job_out) memset(job_out, 0, sizeof(*job_out));
  if (d && d->rows == 0) return NST_OK;
  int rc = rg_check_common(d, A, W, "gemm_layernorm_bwd");
  if (rc != NST_OK) return rc;
  NST_CHECK_ARG(x && gamma && mean && rstd && dx && dgamma && dbeta && workspace, "gemm_layernorm_bwd: null pointer");
  if (!d->trans_b) {
    nst_set_error("gemm_layernorm_bwd: trans_b = 0 is not built (an input-gradient product reads the kernel [256, k] as stored)");
    return NST_ERR_UNSUPPORTED;
  }
  NST_CHECK_ARG(nst_aligned16(x) && nst_aligned16(dx) && nst_aligned16(gamma) && (!dres || nst_aligned16(dres)) && (!dz || nst_aligned16(dz)) &&
                    (((uintptr_t)workspace) & 15) == 0,
                "gemm_layernorm_bwd: operands must be 16-byte aligned");
  NST_CHECK_ARG(d->dropout_p >= 0.f && d->dropout_p < 1.f, "gemm_layernorm_bwd: dropout_p=%f", d->dropout_p);
  const int bm = rg_pick(d->rows, d->k).bm;
  const int64_t nb = (d->rows + bm - 1) / bm;
  if (workspace_bytes < nb * 2 * RN * 4) {
    nst_set_error("gemm_layernorm_bwd: workspace of %lld bytes, %lld needed", (long long)workspace_bytes, (long long)(nb * 2 * RN * 4));
    return NST_ERR_WORKSPACE;
  }
  RowArgs a;
  rg_fill(a, d, A, W);
  a.e.x = x; a.e.gamma = gamma; a.e.mean = const_cast<float*>(mean); a.e.rstd = const_cast<float*>(rstd);
  a.e.dres = (const bf16_t*)dres; a.e.y = (bf16_t*)dx; a.e.dz = (bf16_t*)dz; a.e.partial = (float*)workspace;
  if (dz) {
    nst_dropout_params16(d->dropout_p, &a.e.drop_thresh, &a.e.drop_inv_keep);
    a.seed = d->seed; a.e.stream_id = d->stream_id;
  }
  a.seed_dev = nst_seed_offset_devptr();
  if (!a.seed_dev) return NST_ERR_LAUNCH;
  int nblocks = 0;
  rg_launch<MODE_RC, EPI_LN_BWD>(a, (hipStream_t)stream, &nblocks);
  NST_CHECK_LAUNCH("gemm_layernorm_bwd");
  NstLnFinalizeJob job;
  memset(&job, 0, sizeof(job));
  job.partial = (const float*)workspace; job.dgamma = dgamma; job.dbeta = dbeta;
  job.nblocks = nblocks; job.d = RN; job.accumulate = accumulate;
  if (job_out) {
    *job_out = job;
    return NST_OK;
  }
  return nst_ln_finalize_multi(&job, 1, stream);
}

extern "C" int nst_gemm_rowdot256(const NstRowGemmDesc* d, const void* A, const void* W, void* C_, const void* src, float* dst,
                                  int rows_per_batch, void* stream) {
  if (d && d->rows == 0) return NST_OK;
  int rc = rg_check_common(d, A, W, "gemm_rowdot256");
  if (rc != NST_OK) return rc;
  NST_CHECK_ARG(C_ && nst_aligned16(C_), "gemm_rowdot256: C must be 16-byte aligned");
  NST_CHECK_ARG(src && dst, "gemm_rowdot256: src and dst are required");
  if (!d->trans_b) {
    nst_set_error("gemm_rowdot256: trans_b = 0 is not built (the output projection's input gradient reads the kernel as stored)");
    return NST_ERR_UNSUPPORTED;
  }
  NST_CHECK_ARG(!src || (nst_aligned16(src) && rows_per_batch > 0 && d->rows % rows_per_batch == 0),
                "gemm_rowdot256: rows=%lld is not a multiple of rows_per_batch=%d", (long long)d->rows, rows_per_batch);
  RowArgs a;
  rg_fill(a, d, A, W);
  a.e.y = (bf16_t*)C_;
  a.rd_src = (const bf16_t*)src; a.rd_dst = dst; a.rd_T = rows_per_batch;
  rg_launch<MODE_RC, EPI_ROWDOT>(a, (hipStream_t)stream, nullptr);
  NST_CHECK_LAUNCH("gemm_rowdot256");
  return NST_OK;
}
