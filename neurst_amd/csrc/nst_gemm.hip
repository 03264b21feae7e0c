// Dense GEMM entry point (projections, FFN, front-end dense, tied logits and all their gradients).
#include "nst_gemm_core.h"
#include "nst_gemm256.h"

#include <stdlib.h>

using namespace nstgemm;

namespace {

template <typename T, typename OutT, int AMODE, int BMODE, bool USE_TR>
__global__ void __launch_bounds__(THREADS) dense_gemm_kernel(DenseLoader<T> la, DenseLoader<T> lb, OutT* __restrict__ C,
                                                            int64_t ldc, int M, int N, int K, int tiles_n, int ntiles,
                                                            int kt_per_split, Epilogue ep) {
  __shared__ __attribute__((aligned(16))) char smem[2 * Tile<T>::LDS_BYTES];
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int kt_total = (K + Tile<T>::BK - 1) / Tile<T>::BK;
  const int kt_first = blockIdx.z * kt_per_split;
  int kt_count = kt_total - kt_first;
  if (kt_count > kt_per_split) kt_count = kt_per_split;
  if (kt_count <= 0) return;
  if (ep.drop_thresh) ep.seed = seed_with_offset(ep.seed, ep.seed_dev);
  gemm_block<T, OutT, AMODE, BMODE, USE_TR>(la, lb, C + (int64_t)blockIdx.z * ep.slab_stride, ldc, M, N, tm * BM, tn * BN,
                                            kt_first, kt_count, ep, smem);
}

// EF: compile-time epilogue stages (nst_gemm_core.h: EF_*), EF_GENERIC = every stage a runtime branch
template <typename T, typename OutT, int AMODE, int BMODE, bool CS, int EF = EF_GENERIC>
__global__ void __launch_bounds__(THREADS, 2)
dense_gemm_kernel_v3(GemmArgs<OutT, DenseLoader<T>, DenseLoader<T>, IdentityRowMap> args) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  (void)args;  // read through the kernarg segment, see gemm_stream_v3
  gemm_stream_v3<T, OutT, AMODE, BMODE, DenseLoader<T>, DenseLoader<T>, IdentityRowMap, CS, EF>(smem_dyn);
}

// 256 x 256 tiles, eight waves in two phase-staggered groups (nst_gemm256.h): nst_gemm_wgrad_group.  A SINGLE weight gradient
// stays on the 128 x 128 stream kernel: alone it needs 32 K slices to fill the chip with 256 x 256 tiles, and their slabs cost
// more than the tile saves (ffn1: 68 us against 53; profiles/r04_history/c1_g256_new.json / c1_g256_old.json).
__global__ void __launch_bounds__(G256_THREADS, 2) gemm256_group_kernel(G256GroupArgs args) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  (void)args;
  gemm256_group_block(smem_dyn);
}

// The same 256 x 256 phase-staggered main loop as a plain forward / input-gradient GEMM with the compile-time epilogues of the
// stream kernel (round 5): C[M, N] = epilogue(A[M, K] . Bop), A row-major.  A 256 x 256 tile has 128 FLOP per fetched byte
// where the 128 x 128 stream tile has 64, and every tile kernel here is bound by what a CU can fetch (~35 - 38 GB/s per CU,
// L2 hits included: DESIGN 5f) -- so wherever the tiles fill the chip and the reduction is long enough to amortise the
// 96 KB prologue (d_model >= 512: the text Transformers; nothing of the d_model = 256 speech model qualifies) this kernel
// takes the product.
template <typename OutT>
struct Dense256Args {
  DenseLoader<bf16_t> la;   // RC: outer = M rows, contig = K
  DenseLoader<bf16_t> lb;   // OC: outer = K, contig = N  |  RC: outer = N, contig = K
  OutT* C;
  int64_t ldc;
  int M, N, K;
  int tiles_n, ntiles;
  int reserved0;
  Epilogue ep;
};
// Persistent: a workgroup walks tiles id = blockIdx.x, + gridDim.x, ... (grid = a multiple of 8, so it keeps its XCD and
// xcd_remap hands every XCD a contiguous run of tiles -- neighbours share an A row panel in that L2); the loads a tile starts
// with are issued in FRONT of the previous tile's store epilogue (gemm256_prologue / PRE, as in the conv2 data gradient: the
// epilogue's scratch lives in buffer 1's A images, which the prologue does not touch).
template <typename OutT, int BMODE, int EF>
__global__ void __launch_bounds__(G256_THREADS) dense_gemm256_kernel(Dense256Args<OutT> args) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  typedef Dense256Args<OutT> Args;
  const NST_AS4 Args* ka = (const NST_AS4 Args*)__builtin_amdgcn_kernarg_segment_ptr();
  (void)args;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int ntiles = ka->ntiles, tiles_n = ka->tiles_n;
  const int nk = (ka->K + 63) >> 6;
  int id = blockIdx.x;
  if (id >= ntiles) return;
  auto origin = [&](int i, int& m0, int& n0) {
    const int tile = xcd_remap(i, ntiles);
    const int tm = tile / tiles_n;
    m0 = tm * G256_TILE;
    n0 = (tile - tm * tiles_n) * G256_TILE;
  };
  Dma256<MODE_RC> da;
  Dma256<BMODE> db;
  int m0, n0;
  origin(id, m0, n0);
  {
    const DenseLoader<bf16_t> la = kload(&ka->la), lb = kload(&ka->lb);
    da.init(la, m0, 0, wave, lane);
    db.init(lb, n0, 0, wave, lane);
  }
  gemm256_prologue(smem_dyn, da, db, nk);
  float* epi = reinterpret_cast<float*>(smem_dyn + G256_KT_BYTES + wave * V3_EPI_BYTES_PER_WAVE);
  floatx4_t acc[2][4][4], cs[4];
#pragma unroll 1
  while (true) {
    gemm256_mainloop<MODE_RC, BMODE, false, 0, true>(smem_dyn, da, db, nk, false, acc, cs);
    asm volatile("" ::: "memory");
    const int cm0 = m0, cn0 = n0;
    id += gridDim.x;
    const bool more = id < ntiles;     // workgroup-uniform
    if (more) {
      origin(id, m0, n0);
      const NST_AS4 Args* k1 = launder(ka);
      const DenseLoader<bf16_t> la = kload(&k1->la), lb = kload(&k1->lb);
      da.init(la, m0, 0, wave, lane);
      db.init(lb, n0, 0, wave, lane);
      gemm256_prologue(smem_dyn, da, db, nk);
    }
    const NST_AS4 Args* k2 = launder(ka);
    Epilogue ep = kload(&k2->ep);
    if ((EF & EF_DROP) != 0) ep.seed = seed_with_offset(ep.seed, ep.seed_dev);   // wave-uniform
    const IdentityRowMap rowmap;
    OutT* C = k2->C;
    const int64_t ldc = k2->ldc;
    const int M = k2->M, N = k2->N;
    epilogue_v3<OutT, IdentityRowMap, EF>(acc[0], epi, C, ldc, M, N, cm0 + wr * 128, cn0 + wc * 64, ep, rowmap, lane);
    epilogue_v3<OutT, IdentityRowMap, EF>(acc[1], epi, C, ldc, M, N, cm0 + wr * 128 + 64, cn0 + wc * 64, ep, rowmap, lane);
    if (!more) break;
  }
}

// copies a by-value chunk of the product table into device memory (tables of more than G256_MAX_PROBLEMS products)
__global__ void __launch_bounds__(256) g256_table_write_kernel(G256GroupArgs chunk, G256Problem* dst) {
  const int nw = chunk.nprob * (int)(sizeof(G256Problem) / 4);
  const uint32_t* src = reinterpret_cast<const uint32_t*>(&chunk.p[0]);
  uint32_t* d = reinterpret_cast<uint32_t*>(dst);
  for (int i = threadIdx.x; i < nw; i += blockDim.x) d[i] = src[i];
}

// the stages an Epilogue asks for, as an EF_* mask; -1 when a stage has no compile-time form (alpha, atomics, scalar stores)
int epilogue_mask(const Epilogue& ep) {
  if (!ep.vec || ep.alpha != 1.0f || ep.atomic) return -1;
  int m = 0;
  if (ep.bias) m |= EF_BIAS;
  if (ep.relu) m |= EF_RELU;
  if (ep.drop_thresh) m |= EF_DROP;
  if (ep.residual) m |= EF_RESID;
  if (ep.gate_src) m |= EF_GATE;
  if (ep.posenc) m |= EF_POSENC;
  if (ep.accumulate) m |= EF_ACCUM;
  if (ep.rowdot_dst) m |= EF_ROWDOT;
  return m;
}

// workgroups of the persistent kernels: two per CU are resident (64 KB of LDS each); every workgroup gets the same
// number of units, and the count is a multiple of 8 so that a workgroup keeps its XCD across units
int v3_grid(int units) {
  static int resident = 0;
  if (!resident) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    resident = 2 * (cus > 0 ? cus : 256);
  }
  if (units <= resident) return units;
  const int rounds = (units + resident - 1) / resident;
  int g = (units + rounds - 1) / rounds;
  g = (g + 7) & ~7;
  return g > units ? units : g;
}

template <typename KernelT>
void allow_big_lds(KernelT kernel, int bytes) {
  static thread_local const void* done[64];
  static thread_local int ndone = 0;
  for (int i = 0; i < ndone; ++i)
    if (done[i] == (const void*)kernel) return;
  (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (ndone < 64) done[ndone++] = (const void*)kernel;
}

int device_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
  }
  return cus;
}

bool use_v2() { return true; }   // (the register-staged kernel serves operands the LDS-DMA path cannot take: unaligned, odd widths)

// sum_z src[z * zs .. +4): 8 slabs per round, all 8 loads in flight before the first add (a runtime-bound loop of single
// loads is one dependent memory round trip per slab: the 64 slabs of a 256x256 gradient took ~60 us that way).  The adds
// stay in slab order, so the result does not depend on the unroll.
__device__ __forceinline__ float4 sum_slabs(const float* __restrict__ src, int64_t zs, int split) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
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
  return acc;
}

// C[i] (+)= sum_z slabs[z][i]   (split-K second stage; slabs are [split][M][N] f32, C has leading dimension ldc)
// and, when cs_parts != NULL, cs_out[j] (+)= sum_z cs_parts[z][j]  (the fused column sums)
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ C, int M, int N,
                                                           int64_t ldc, int split, int accumulate,
                                                           const float* __restrict__ cs_parts, float* __restrict__ cs_out,
                                                           int cs_accumulate) {
  const int64_t total4 = (int64_t)M * N / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 acc = sum_slabs(slabs + i * 4, (int64_t)M * N, split);
    const int64_t e = i * 4;
    const int row = (int)(e / N), col = (int)(e - (int64_t)row * N);
    float* o = C + (int64_t)row * ldc + col;
    if (accumulate) {
      const float4 old = *reinterpret_cast<const float4*>(o);
      acc.x += old.x; acc.y += old.y; acc.z += old.z; acc.w += old.w;
    }
    *reinterpret_cast<float4*>(o) = acc;
  }
  if (cs_parts) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < N; j += gridDim.x * blockDim.x) {
      float acc = 0.f;
      for (int z = 0; z < split; ++z) acc += cs_parts[(int64_t)z * N + j];
      cs_out[j] = cs_accumulate ? cs_out[j] + acc : acc;
    }
  }
}

// up to 8 deferred split-K second stages in one launch: blockIdx.y = job
struct SplitkJobs { NstSplitkJob j[8]; };
__global__ void __launch_bounds__(256) splitk_reduce_multi_kernel(SplitkJobs jobs) {
  const NstSplitkJob& q = jobs.j[blockIdx.y];
  const int M = q.M, N = q.N, split = q.split;
  const int64_t total4 = (int64_t)M * N / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 acc = sum_slabs(q.slabs + i * 4, (int64_t)M * N, split);
    const int64_t e = i * 4;
    const int row = (int)(e / N), col = (int)(e - (int64_t)row * N);
    float* o = q.C + (int64_t)row * q.ldc + col;
    if (q.accumulate) {
      const float4 old = *reinterpret_cast<const float4*>(o);
      acc.x += old.x; acc.y += old.y; acc.z += old.z; acc.w += old.w;
    }
    *reinterpret_cast<float4*>(o) = acc;
  }
  if (q.cs_parts) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < N; j += gridDim.x * blockDim.x) {
      float acc = 0.f;
      for (int z = 0; z < split; ++z) acc += q.cs_parts[(int64_t)z * N + j];
      q.cs_out[j] = q.cs_accumulate ? q.cs_out[j] + acc : acc;
    }
  }
}

template <typename T>
DenseLoader<T> make_loader(const void* base, int64_t ld, int mode, int out_extent, int k_extent) {
  DenseLoader<T> l;
  l.base = (const T*)base;
  l.ld = ld;
  if (mode == MODE_RC) { l.outer_limit = out_extent; l.contig_limit = k_extent; }
  else { l.outer_limit = k_extent; l.contig_limit = out_extent; }
  l.vec = nst_aligned16(base) && ((ld * (int64_t)sizeof(T)) % 16 == 0) && (l.contig_limit % Tile<T>::E == 0);
  return l;
}

bool use_tr() { return true; }   // (the fragment path without ds_read_b64_tr_b16 is no longer instantiated)

template <typename T, typename OutT>
int launch(const NstGemmDesc* d, const void* A, const void* B, void* C, const Epilogue& ep, int split, hipStream_t st) {
  // Aop[i][r]: trans_a==0 -> A[i*lda + r] (RC);  trans_a==1 -> A[r*lda + i] (OC)
  // Bop[j][r]: trans_b==1 -> B[j*ldb + r] (RC);  trans_b==0 -> B[r*ldb + j] (OC)
  const int amode = d->trans_a ? MODE_OC : MODE_RC;
  const int bmode = d->trans_b ? MODE_RC : MODE_OC;
  DenseLoader<T> la = make_loader<T>(A, d->lda, amode, d->M, d->K);
  DenseLoader<T> lb = make_loader<T>(B, d->ldb, bmode, d->N, d->K);
  const int tiles_m = (d->M + BM - 1) / BM, tiles_n = (d->N + BN - 1) / BN, ntiles = tiles_m * tiles_n;
  const int kt_total = (d->K + Tile<T>::BK - 1) / Tile<T>::BK;
  if (split > kt_total) split = kt_total;
  if (split < 1) split = 1;
  const bool tr = use_tr();
  const int kt_per_split = (kt_total + split - 1) / split;
  split = (kt_total + kt_per_split - 1) / kt_per_split;
  dim3 grid(ntiles, 1, split);
  // ---- 256 x 256 tiles for bf16 forward / input-gradient products that fill the chip with them (see dense_gemm256_kernel)
  if constexpr (sizeof(T) == 2 && sizeof(OutT) == 2) {
    const int em0 = epilogue_mask(ep);
    const int t256 = ((d->M + 255) / 256) * ((d->N + 255) / 256), cus = device_cus();
    const int rounds = (t256 + cus - 1) / cus;
    if (use_v2() && amode == MODE_RC && la.vec && lb.vec && split == 1 && !ep.colsum_dst && em0 >= 0 && d->K >= 512 &&
        d->M >= 256 && d->N >= 256 && 10 * t256 >= 8 * rounds * cus && d->lda < (1ll << 31) && d->ldb < (1ll << 31)) {
      Dense256Args<OutT> ga;
      ga.la = la; ga.lb = lb; ga.C = (OutT*)C; ga.ldc = d->ldc; ga.M = d->M; ga.N = d->N; ga.K = d->K;
      ga.tiles_n = (d->N + 255) / 256; ga.ntiles = t256; ga.reserved0 = 0; ga.ep = ep;
      const int pgrid = cus & ~7;     // one persistent workgroup per CU (128 KB of LDS), a multiple of 8: it keeps its XCD
#define NST_GEMM_256(BMO, EF_)                                                                       \
  do {                                                                                               \
    auto kfn = dense_gemm256_kernel<OutT, BMO, EF_>;                                                 \
    allow_big_lds(kfn, G256_LDS_BYTES);                                                              \
    kfn<<<t256 < pgrid ? t256 : pgrid, G256_THREADS, G256_LDS_BYTES, st>>>(ga);                      \
    return 0;                                                                                        \
  } while (0)
      if (bmode == MODE_OC) {   // forward projections: x [M, K] . W [K, N]
        switch (em0) {
          case EF_BIAS: NST_GEMM_256(MODE_OC, EF_BIAS);
          case EF_BIAS | EF_DROP: NST_GEMM_256(MODE_OC, EF_BIAS | EF_DROP);
          case EF_BIAS | EF_DROP | EF_RESID: NST_GEMM_256(MODE_OC, EF_BIAS | EF_DROP | EF_RESID);
          case EF_BIAS | EF_RESID: NST_GEMM_256(MODE_OC, EF_BIAS | EF_RESID);
          case EF_BIAS | EF_RELU | EF_DROP: NST_GEMM_256(MODE_OC, EF_BIAS | EF_RELU | EF_DROP);
          case EF_BIAS | EF_RELU: NST_GEMM_256(MODE_OC, EF_BIAS | EF_RELU);
          default: break;
        }
      } else {                  // input gradients and tied logits: dz [M, K] . W^T
        switch (em0) {
          case 0: NST_GEMM_256(MODE_RC, 0);
          case EF_GATE: NST_GEMM_256(MODE_RC, EF_GATE);
          case EF_ROWDOT: NST_GEMM_256(MODE_RC, EF_ROWDOT);
          case EF_RESID: NST_GEMM_256(MODE_RC, EF_RESID);
          default: break;
        }
      }
#undef NST_GEMM_256
    }
  }
  if (use_v2() && tr && la.vec && lb.vec) {  // LDS-DMA stream kernel: needs 16-byte aligned, 8-element granular operands
    const int units = ntiles * split;
    dim3 g3(v3_grid(units), 1, 1);
    GemmArgs<OutT, DenseLoader<T>, DenseLoader<T>, IdentityRowMap> ga;
    ga.la = la; ga.lb = lb; ga.C = (OutT*)C; ga.ldc = d->ldc; ga.M = d->M; ga.N = d->N; ga.K = d->K;
    ga.tiles_n = tiles_n; ga.ntiles = ntiles; ga.split = split; ga.kt_per_split = kt_per_split; ga.ep = ep;
    ga.z_per_xcd = (split >= 8 && split % 8 == 0 && g3.x % 8 == 0) ? 1 : 0;
    ga.reserved0 = 0;
#define NST_GEMM_LAUNCH3E(AM, BMO, CS_, EF_)                                                                          \
  do {                                                                                                               \
    auto kfn = dense_gemm_kernel_v3<T, OutT, AM, BMO, CS_, EF_>;                                                      \
    allow_big_lds(kfn, V3_LDS_BYTES);                                                                                 \
    kfn<<<g3, THREADS, V3_LDS_BYTES, st>>>(ga);                                                                       \
  } while (0)
#define NST_GEMM_LAUNCH3(AM, BMO, CS_) NST_GEMM_LAUNCH3E(AM, BMO, CS_, EF_GENERIC)
    // Specialised instantiations: the epilogue configurations a training step of the Transformer models actually issues
    // (enumerated by tracing a step; everything else takes the generic kernel below).
    // (round 6: the fp32 path -- BASELINE config #2 -- takes them too; its products ran on the generic-epilogue kernels, which
    // keep ~100 scalars of the argument block alive around the epilogue: 205 - 224 SGPR spills each)
    const int em = epilogue_mask(ep);
    if (em >= 0) {
      if constexpr (sizeof(T) == sizeof(OutT)) {
        if (amode == MODE_RC && bmode == MODE_OC && !ep.colsum_dst) {   // forward projections: x [M,K] . W [K,N]
          switch (em) {
            case 0: NST_GEMM_LAUNCH3E(MODE_RC, MODE_OC, false, 0); return 0;
            case EF_BIAS: NST_GEMM_LAUNCH3E(MODE_RC, MODE_OC, false, EF_BIAS); return 0;
            case EF_BIAS | EF_DROP: NST_GEMM_LAUNCH3E(MODE_RC, MODE_OC, false, EF_BIAS | EF_DROP); return 0;   // (fp32 residual stream: the next LayerNorm adds)
            case EF_BIAS | EF_DROP | EF_RESID: NST_GEMM_LAUNCH3E(MODE_RC, MODE_OC, false, EF_BIAS | EF_DROP | EF_RESID); return 0;
            case EF_BIAS | EF_RESID: NST_GEMM_LAUNCH3E(MODE_RC, MODE_OC, false, EF_BIAS | EF_RESID); return 0;
            case EF_BIAS | EF_RELU | EF_DROP: NST_GEMM_LAUNCH3E(MODE_RC, MODE_OC, false, EF_BIAS | EF_RELU | EF_DROP); return 0;
            case EF_BIAS | EF_RELU: NST_GEMM_LAUNCH3E(MODE_RC, MODE_OC, false, EF_BIAS | EF_RELU); return 0;
            case EF_BIAS | EF_POSENC: NST_GEMM_LAUNCH3E(MODE_RC, MODE_OC, false, EF_BIAS | EF_POSENC); return 0;
            default: break;
          }
        }
        if (amode == MODE_RC && bmode == MODE_RC && !ep.colsum_dst) {   // input gradients / tied logits: dz [M,K] . W^T
          switch (em) {
            case 0: NST_GEMM_LAUNCH3E(MODE_RC, MODE_RC, false, 0); return 0;
            case EF_BIAS: NST_GEMM_LAUNCH3E(MODE_RC, MODE_RC, false, EF_BIAS); return 0;
            case EF_GATE: NST_GEMM_LAUNCH3E(MODE_RC, MODE_RC, false, EF_GATE); return 0;
            case EF_ROWDOT: if constexpr (sizeof(T) == 2) { NST_GEMM_LAUNCH3E(MODE_RC, MODE_RC, false, EF_ROWDOT); return 0; } break;
            case EF_ACCUM: NST_GEMM_LAUNCH3E(MODE_RC, MODE_RC, false, EF_ACCUM); return 0;
            case EF_RESID: NST_GEMM_LAUNCH3E(MODE_RC, MODE_RC, false, EF_RESID); return 0;
            default: break;
          }
        }
      }
      if constexpr (sizeof(OutT) == 4) {
        if (amode == MODE_OC && bmode == MODE_OC) {                     // weight gradients: x^T . dz, slabs or in place
          if (ep.colsum_dst) {
            if (em == 0) { NST_GEMM_LAUNCH3E(MODE_OC, MODE_OC, true, 0); return 0; }
            if (em == EF_ACCUM) { NST_GEMM_LAUNCH3E(MODE_OC, MODE_OC, true, EF_ACCUM); return 0; }
          } else {
            if (em == 0) { NST_GEMM_LAUNCH3E(MODE_OC, MODE_OC, false, 0); return 0; }
            if (em == EF_ACCUM) { NST_GEMM_LAUNCH3E(MODE_OC, MODE_OC, false, EF_ACCUM); return 0; }
          }
        }
      }
    }
    if (ep.colsum_dst) {  // host guarantees: OC/OC operands, f32 output
      if constexpr (sizeof(OutT) == 4) { NST_GEMM_LAUNCH3(MODE_OC, MODE_OC, true); return 0; }
    }
    if (amode == MODE_RC && bmode == MODE_RC) NST_GEMM_LAUNCH3(MODE_RC, MODE_RC, false);
    else if (amode == MODE_RC && bmode == MODE_OC) NST_GEMM_LAUNCH3(MODE_RC, MODE_OC, false);
    else if (amode == MODE_OC && bmode == MODE_RC) NST_GEMM_LAUNCH3(MODE_OC, MODE_RC, false);
    else NST_GEMM_LAUNCH3(MODE_OC, MODE_OC, false);
#undef NST_GEMM_LAUNCH3
#undef NST_GEMM_LAUNCH3E
    return 0;
  }
#define NST_GEMM_LAUNCH(AM, BMO, TR)                                                                                   \
  dense_gemm_kernel<T, OutT, AM, BMO, TR><<<grid, THREADS, 0, st>>>(la, lb, (OutT*)C, d->ldc, d->M, d->N, d->K, tiles_n, \
                                                                   ntiles, kt_per_split, ep)
  if (amode == MODE_RC && bmode == MODE_RC) NST_GEMM_LAUNCH(MODE_RC, MODE_RC, true);
  else if (amode == MODE_RC && bmode == MODE_OC) NST_GEMM_LAUNCH(MODE_RC, MODE_OC, true);
  else if (amode == MODE_OC && bmode == MODE_RC) NST_GEMM_LAUNCH(MODE_OC, MODE_RC, true);
  else NST_GEMM_LAUNCH(MODE_OC, MODE_OC, true);
#undef NST_GEMM_LAUNCH
  return 0;
}

}  // namespace

extern "C" int nst_splitk_reduce_multi(const NstSplitkJob* jobs, int njobs, void* stream) {
  NST_CHECK_ARG(njobs >= 0 && njobs <= 8 && (njobs == 0 || jobs), "splitk_reduce_multi: 0..8 jobs");
  if (njobs == 0) return NST_OK;
  SplitkJobs packed;
  int64_t most = 0;
  for (int i = 0; i < njobs; ++i) {
    NST_CHECK_ARG(jobs[i].slabs && jobs[i].C && jobs[i].M > 0 && jobs[i].N > 0 && jobs[i].N % 4 == 0 && jobs[i].split > 0,
                  "splitk_reduce_multi: bad job %d", i);
    packed.j[i] = jobs[i];
    const int64_t t4 = (int64_t)jobs[i].M * jobs[i].N / 4;
    most = t4 > most ? t4 : most;
  }
  int blocks = (int)((most + 255) / 256 > 1024 ? 1024 : (most + 255) / 256);
  splitk_reduce_multi_kernel<<<dim3(blocks, njobs), 256, 0, (hipStream_t)stream>>>(packed);
  NST_CHECK_LAUNCH("splitk_reduce_multi");
  return NST_OK;
}

extern "C" int nst_gemm_wgrad_group(const NstGemmDesc* descs, const void* const* A, const void* const* B, void* const* C, int n,
                                    void* workspace, int64_t workspace_bytes, void* stream) {
  constexpr int MAXN = 1024;
  NST_CHECK_ARG(n >= 0 && n <= MAXN && (n == 0 || (descs && A && B && C)), "gemm_wgrad_group: 0..%d products", MAXN);
  if (n == 0) return NST_OK;
  for (int i = 0; i < n; ++i) {
    const NstGemmDesc* d = &descs[i];
    NST_CHECK_ARG(A[i] && B[i] && C[i] && d->M > 0 && d->N > 0 && d->K > 0, "gemm_wgrad_group: bad product %d", i);
    const bool ok = d->in_dtype == NST_BF16 && d->out_dtype == NST_F32 && d->trans_a && !d->trans_b && d->alpha == 1.0f &&
                    !d->bias && !d->relu && d->dropout_p == 0.f && !d->residual && !d->gate_src && !d->posenc &&
                    !d->rowdot_dst && d->split_k <= 1 && nst_aligned16(A[i]) && nst_aligned16(B[i]) && nst_aligned16(C[i]) &&
                    (d->lda * 2) % 16 == 0 && (d->ldb * 2) % 16 == 0 && (d->ldc * 4) % 16 == 0 && d->M % 8 == 0 &&
                    d->N % 8 == 0 && d->lda >= d->M && d->ldb >= d->N && d->ldc >= d->N && d->lda < (1ll << 31) &&
                    d->ldb < (1ll << 31) && d->ldc < (1ll << 31) && (!d->colsum || ((uintptr_t)d->colsum & 3) == 0);
    if (!ok) {
      nst_set_error("gemm_wgrad_group: product %d is not a plain aligned bf16 weight gradient (trans_a, f32 output)", i);
      return NST_ERR_UNSUPPORTED;
    }
  }
  const bool ext = n > G256_MAX_PROBLEMS;
  NST_CHECK_ARG(!ext || (workspace && nst_aligned16(workspace) && workspace_bytes >= (int64_t)n * (int64_t)sizeof(G256Problem)),
                "gemm_wgrad_group: %d products need a table workspace of %lld bytes", n, (long long)n * (long long)sizeof(G256Problem));
  // longest reductions first (the hardware hands out workgroups in index order), then the products with most tiles
  static thread_local int order[MAXN];
  for (int i = 0; i < n; ++i) order[i] = i;
  auto tiles_of = [&](int i) { return ((descs[i].M + 255) / 256) * ((descs[i].N + 255) / 256); };
  for (int i = 1; i < n; ++i) {   // insertion sort (stable; the host usually hands the products over nearly sorted)
    const int v = order[i];
    int j = i - 1;
    while (j >= 0 && (descs[order[j]].K < descs[v].K || (descs[order[j]].K == descs[v].K && tiles_of(order[j]) < tiles_of(v)))) {
      order[j + 1] = order[j];
      --j;
    }
    order[j + 1] = v;
  }
  hipStream_t st = (hipStream_t)stream;
  G256GroupArgs ga;
  int units = 0;
  for (int base = 0; base < n; base += G256_MAX_PROBLEMS) {
    const int m = n - base < G256_MAX_PROBLEMS ? n - base : G256_MAX_PROBLEMS;
    for (int i = 0; i < m; ++i) {
      const int q = order[base + i];
      const NstGemmDesc* d = &descs[q];
      G256Problem& p = ga.p[i];
      p.A = (const bf16_t*)A[q]; p.B = (const bf16_t*)B[q]; p.C = (float*)C[q]; p.colsum = d->colsum;
      p.lda = (int)d->lda; p.ldb = (int)d->ldb; p.ldc = (int)d->ldc;
      p.M = d->M; p.N = d->N; p.K = d->K;
      p.tiles_n = (d->N + 255) / 256;
      units += tiles_of(q);
      p.unit_end = units;
      p.flags = (d->accumulate ? 1 : 0) | (d->colsum_accumulate ? 2 : 0);
      p.reserved = 0;
    }
    if (ext) {
      ga.nprob = m; ga.nunits = 0; ga.ext = nullptr;
      g256_table_write_kernel<<<1, 256, 0, st>>>(ga, (G256Problem*)workspace + base);
      NST_CHECK_LAUNCH("gemm_wgrad_group(table)");
    }
  }
  ga.nprob = n;
  ga.nunits = units;
  ga.ext = ext ? (const G256Problem*)workspace : nullptr;
  const int grid = (units + 63) / 64 * 64;
  allow_big_lds(gemm256_group_kernel, G256_LDS_BYTES);
  gemm256_group_kernel<<<grid, G256_THREADS, G256_LDS_BYTES, st>>>(ga);
  NST_CHECK_LAUNCH("gemm_wgrad_group");
  return NST_OK;
}

extern "C" int nst_gemm(const NstGemmDesc* d, const void* A, const void* B, void* C, void* stream) {
  NST_CHECK_ARG(d && A && B && C, "gemm: null pointer");
  NST_CHECK_ARG(d->M >= 0 && d->N >= 0 && d->K >= 0, "gemm: negative dims");
  NST_CHECK_ARG(d->in_dtype == NST_F32 || d->in_dtype == NST_BF16, "gemm: bad in_dtype %d", d->in_dtype);
  NST_CHECK_ARG(d->out_dtype == NST_F32 || d->out_dtype == NST_BF16, "gemm: bad out_dtype %d", d->out_dtype);
  NST_CHECK_ARG(!(d->in_dtype == NST_F32 && d->out_dtype == NST_BF16), "gemm: f32 inputs with bf16 output unsupported");
  NST_CHECK_ARG(d->lda >= (d->trans_a ? d->M : d->K), "gemm: lda=%lld too small", (long long)d->lda);
  NST_CHECK_ARG(d->ldb >= (d->trans_b ? d->K : d->N), "gemm: ldb=%lld too small", (long long)d->ldb);
  NST_CHECK_ARG(d->ldc >= d->N, "gemm: ldc=%lld too small", (long long)d->ldc);
  NST_CHECK_ARG(d->dropout_p >= 0.f && d->dropout_p < 1.f, "gemm: dropout_p=%f", d->dropout_p);
  NST_CHECK_ARG(!d->posenc || d->posenc_period > 0, "gemm: posenc needs posenc_period > 0");
  hipStream_t st = (hipStream_t)stream;
  if (d->M == 0 || d->N == 0) return NST_OK;

  Epilogue ep;
  {
    const int osz = nst_dtype_size(d->out_dtype);
    bool v = nst_aligned16(C) && ((d->ldc * osz) % 16 == 0) && (d->N % 8 == 0);
    if (d->residual) v = v && nst_aligned16(d->residual) && ((d->ldr * osz) % 16 == 0);
    if (d->gate_src) v = v && nst_aligned16(d->gate_src) && ((d->ldg * osz) % 16 == 0);
    if (d->posenc) v = v && nst_aligned16(d->posenc);
    if (d->bias) v = v && ((((uintptr_t)d->bias) & 3) == 0);
    ep.vec = v ? 1 : 0;
  }
  ep.alpha = d->alpha;
  ep.bias = d->bias;
  ep.relu = d->relu;
  nst_dropout_params16(d->dropout_p, &ep.drop_thresh, &ep.drop_inv_keep);
  ep.seed = d->seed;
  ep.stream_id = d->stream_id;
  ep.seed_dev = nst_seed_offset_devptr();
  if (!ep.seed_dev) return NST_ERR_LAUNCH;
  ep.residual = d->residual;
  ep.ldr = d->ldr;
  ep.gate_src = d->gate_src;
  ep.ldg = d->ldg;
  ep.gate_scale = d->gate_scale;
  ep.posenc = d->posenc;
  ep.posenc_period = d->posenc_period > 0 ? d->posenc_period : 1;
  ep.emb_scale = d->emb_scale;
  ep.accumulate = d->accumulate;
  ep.atomic = 0;
  ep.slab_stride = 0;
  ep.colsum_dst = nullptr;
  ep.colsum_zstride = 0;
  ep.colsum_acc = 0;
  ep.rowdot_src = nullptr;
  ep.ldrs = 0;
  ep.rowdot_dst = nullptr;
  ep.rowdot_T = 1;
  ep.rowdot_H = 1;
  if (d->rowdot_dst) {
    const DenseLoader<bf16_t> ta = make_loader<bf16_t>(A, d->lda, d->trans_a ? MODE_OC : MODE_RC, d->M, d->K);
    const DenseLoader<bf16_t> tb = make_loader<bf16_t>(B, d->ldb, d->trans_b ? MODE_RC : MODE_OC, d->N, d->K);
    NST_CHECK_ARG(d->in_dtype == NST_BF16 && d->out_dtype == NST_BF16 && d->split_k <= 1 && ep.vec && use_v2() && use_tr() &&
                      ta.vec && tb.vec && d->N % 64 == 0 && d->rowdot_src && nst_aligned16(d->rowdot_src) &&
                      (d->ldrs * 2) % 16 == 0 && d->ldrs >= d->N && d->rowdot_rows > 0 && d->rowdot_heads * 64 == d->N &&
                      d->M % d->rowdot_rows == 0 && !d->accumulate,
                  "gemm: rowdot needs the bf16 stream kernel (aligned operands), N = heads * 64, M = batches * rowdot_rows");
    ep.rowdot_src = d->rowdot_src;
    ep.ldrs = d->ldrs;
    ep.rowdot_dst = d->rowdot_dst;
    ep.rowdot_T = d->rowdot_rows;
    ep.rowdot_H = d->rowdot_heads;
  }
  // column sums of the B operand (bias gradient): fused into the MFMA loop when the LDS-DMA kernel runs on
  // OC/OC operands with an f32 output, otherwise a separate nst_colsum pass at the end
  bool cs_fused = false;
  if (d->colsum) {
    NST_CHECK_ARG(!d->trans_b, "gemm: colsum needs trans_b == 0 (B stored [K,N])");
    const int esz = nst_dtype_size(d->in_dtype), E = 16 / esz;
    cs_fused = d->trans_a && d->out_dtype == NST_F32 && use_v2() && use_tr() && d->K > 0 && nst_aligned16(A) &&
               nst_aligned16(B) && (d->lda * esz) % 16 == 0 && (d->ldb * esz) % 16 == 0 && d->M % E == 0 && d->N % E == 0;
  }

  int split = d->split_k > 1 ? d->split_k : 1;
  if (split > 1) {
    NST_CHECK_ARG(d->out_dtype == NST_F32, "gemm: split_k requires an f32 output");
    NST_CHECK_ARG(!d->relu && d->dropout_p == 0.f && !d->gate_src && !d->posenc && !d->bias && !d->residual,
                  "gemm: split_k supports only the plain alpha*A*B (+accumulate) epilogue");
    const int kt_total = (d->K + (d->in_dtype == NST_BF16 ? 64 : 32) - 1) / (d->in_dtype == NST_BF16 ? 64 : 32);
    if (split > kt_total) split = kt_total;
    const int kps = (kt_total + split - 1) / split;
    split = (kt_total + kps - 1) / kps;  // the split count launch() will actually use
    const int64_t need = (int64_t)split * d->M * d->N * 4 + (cs_fused ? (int64_t)split * d->N * 4 : 0);
    const bool slab = split > 1 && d->workspace && d->workspace_bytes >= need && nst_aligned16(d->workspace) &&
                      (d->N % 4 == 0) && ((d->ldc * 4) % 16 == 0) && nst_aligned16(C);
    if (slab) {
      Epilogue eps = ep;
      eps.accumulate = 0;
      eps.slab_stride = (int64_t)d->M * d->N;
      eps.vec = (d->N % 8 == 0) ? 1 : 0;
      float* cs_parts = cs_fused ? (float*)d->workspace + (int64_t)split * d->M * d->N : nullptr;
      eps.colsum_dst = cs_parts;
      eps.colsum_zstride = d->N;
      NstGemmDesc ds = *d;
      ds.ldc = d->N;  // slabs are dense [M][N]
      if (d->in_dtype == NST_F32) launch<float, float>(&ds, A, B, d->workspace, eps, split, st);
      else launch<bf16_t, float>(&ds, A, B, d->workspace, eps, split, st);
      NST_CHECK_LAUNCH("gemm(split-K partials)");
      if (d->reduce_job_out && (!d->colsum || cs_fused)) {   // deferred second stage: describe it, the caller batches it
        NstSplitkJob* j = d->reduce_job_out;
        j->slabs = (const float*)d->workspace; j->C = (float*)C; j->ldc = d->ldc; j->M = d->M; j->N = d->N; j->split = split;
        j->accumulate = d->accumulate; j->cs_parts = cs_parts; j->cs_out = d->colsum; j->cs_accumulate = d->colsum_accumulate;
        j->reserved = 0;
        return NST_OK;
      }
      const int64_t total4 = (int64_t)d->M * d->N / 4;
      int blocks = (int)((total4 + 255) / 256 > 2048 ? 2048 : (total4 + 255) / 256);
      splitk_reduce_kernel<<<blocks, 256, 0, st>>>((const float*)d->workspace, (float*)C, d->M, d->N, d->ldc, split,
                                                  d->accumulate, cs_parts, d->colsum, d->colsum_accumulate);
      NST_CHECK_LAUNCH("gemm(split-K reduce)");
      if (d->colsum && !cs_fused)
        return nst_colsum(B, d->colsum, d->K, d->N, d->ldb, d->in_dtype, d->colsum_accumulate, nullptr, 0, stream);
      return NST_OK;
    }
    cs_fused = false;  // atomic split-K: the column sums take the separate pass
    if (!d->accumulate)
      NST_CHECK_HIP(hipMemset2DAsync(C, d->ldc * sizeof(float), 0, (size_t)d->N * sizeof(float), d->M, st));
    ep.atomic = 1;
    ep.accumulate = 0;
  }
  if (d->K == 0) {  // empty reduction: C = epilogue(0)
    nst_set_error("gemm: K == 0 unsupported");
    return NST_ERR_UNSUPPORTED;
  }
  if (cs_fused) {
    ep.colsum_dst = d->colsum;
    ep.colsum_acc = d->colsum_accumulate;
  }
  if (d->in_dtype == NST_F32) launch<float, float>(d, A, B, C, ep, split, st);
  else if (d->out_dtype == NST_BF16) launch<bf16_t, bf16_t>(d, A, B, C, ep, split, st);
  else launch<bf16_t, float>(d, A, B, C, ep, split, st);
  NST_CHECK_LAUNCH("gemm");
  if (d->colsum && !cs_fused)
    return nst_colsum(B, d->colsum, d->K, d->N, d->ldb, d->in_dtype, d->colsum_accumulate, nullptr, 0, stream);
  return NST_OK;
}
