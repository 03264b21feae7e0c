// 256 x 256 bf16 tile GEMM with waves in DIFFERENT phases (round 4).
//
// Why: every tile kernel of rounds 1-3 put all waves of a workgroup through load-issue -> barrier -> fragment reads -> MFMA in
// lock-step, so the load path and the matrix cores took turns (20-33 % of the MFMA peak whatever the K loop
// looked like).  Here a workgroup of EIGHT waves owns a 256 x 256 output tile (128 FLOP per fetched byte instead of 64) and its
// two wave groups run HALF A PHASE APART:
//
//   group g = wave >> 2 owns rows [128 g, +128); wave (g, c) owns the 128 x 64 block at columns [64 c, +64)
//   K step = 64, 4 phases per K step; a phase = { fragment reads + ONE half-tile of LDS-DMA | barrier | 16 MFMAs | barrier }
//   group 1 executes one extra barrier before the loop, so while group 0 multiplies, group 1 reads fragments / issues DMA and
//   vice versa: each SIMD holds one wave of either group and its matrix core always has a wave in an MFMA section.
//
//   phase 0: read B-lo (4) + A-lo (8)   DMA A0(t+1)   MFMA A-lo x B-lo          A-lo/hi: rows [0,64) / [64,128) of the wave
//   phase 1: read B-hi (4)              DMA A1(t+1)   MFMA A-lo x B-hi          B-lo/hi: columns [0,32) / [32,64)
//   phase 2: read A-hi (8)              DMA B0(t+2)   MFMA A-hi x B-hi
//   phase 3: -                          DMA B1(t+2)   MFMA A-hi x B-lo          + the one counted vmcnt of the K step
//
// LDS: two K-step buffers of four 16 KB half-tile images [A0 | A1 | B0 | B1] = 128 KB; an image is exactly the 128 x 64 stage
// image of the 128 x 128 kernels (nst_gemm_core.h: lane-linear LDS-DMA destination, XOR swizzle on the SOURCE chunk and on the
// fragment read), staged cooperatively by all eight waves (2 x 1 KB pieces per wave and half-tile).
//
// Hazards (MI355X: nothing orders a ds_read behind a pending LDS-DMA except the issuing wave's vmcnt + a barrier):
//   RAW  the wait sits at the end of phase 3's load section (before its first barrier) in BOTH groups; the buffer is first
//        read in the next phase 0, i.e. behind a barrier that both groups' waits precede;
//   WAR  every wave retires its fragment reads (lgkmcnt(0)) BEFORE the barrier that ends its load section, so a half-tile
//        may be restaged one phase after its last read: A (last read in phase 2) in phases 0 / 1 of the next K step, B (last
//        read in phase 1) in phases 2 / 3 of the same K step.
//   vmcnt(4) at phase 3 leaves B0 / B1 of step t+2 in flight and retires everything of step t+1.
#pragma once
#include "nst_gemm_core.h"

namespace nstgemm {

constexpr int G256_THREADS = 512;
constexpr int G256_TILE = 256;
constexpr int G256_HALF_BYTES = 128 * KBYTES;        // 16 KB
constexpr int G256_KT_BYTES = 4 * G256_HALF_BYTES;   // 64 KB
constexpr int G256_LDS_BYTES = 2 * G256_KT_BYTES;    // 128 KB

// DMA cursors of one operand (two half-tiles), eight waves x two pieces per half-tile.  Chunk c = (s * 8 + wave) * 64 + lane of
// a half-tile image lands at image + c * 16; its source is the swizzled chunk of nst_gemm_core.h::dma_tile.
template <int MODE>
struct Dma256 {
  const char* p[2][2];   // [half][piece]
  int kb[2];             // reduction index of piece s at the unit's first K step
  bool ov[2][2];         // the chunk's fixed (non-reduction) coordinate lies inside the matrix
  bool allv[2];          // wave-uniform: every chunk this wave moves for half h is inside
  int nfull;             // K steps of the unit that lie entirely inside the reduction range
  int k_limit;
  int64_t step_bytes;
  __device__ __forceinline__ void init(const DenseLoader<bf16_t>& ld, int o0, int r0, int wave, int lane) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int c = (s * 8 + wave) * 64 + lane;
      if (MODE == MODE_RC) {
        const int row = c >> 3, slot = c & 7;
        const int kchunk = slot ^ ((row >> 1) & 7);
        kb[s] = r0 + kchunk * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int outer = o0 + h * 128 + row;
          ov[h][s] = outer < ld.outer_limit;
          p[h][s] = reinterpret_cast<const char*>(ld.base + (int64_t)outer * ld.ld + kb[s]);
        }
      } else {
        const int r = c >> 4, c16 = c & 15;
        const int g = (r & 3) | (((r >> 3) & 1) << 2);
        kb[s] = r0 + r;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int col = o0 + h * 128 + (c16 ^ (g << 1)) * 8;
          ov[h][s] = col < ld.contig_limit;
          p[h][s] = reinterpret_cast<const char*>(ld.base + (int64_t)kb[s] * ld.ld + col);
        }
      }
    }
    k_limit = MODE == MODE_RC ? ld.contig_limit : ld.outer_limit;
    step_bytes = MODE == MODE_RC ? (int64_t)128 : (int64_t)128 * ld.ld;
    const int nf = (k_limit - r0) / 64;
    nfull = nf > 0 ? nf : 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) allv[h] = __builtin_amdgcn_ballot_w64(!(ov[h][0] && ov[h][1])) == 0;
  }
  // K step t (relative to the unit) of half H -> the half-tile image at LDS address img (wave-uniform)
  template <int H>
  __device__ __forceinline__ void issue(int t, uint32_t img, int wave) {
    const uint32_t dst = img + (uint32_t)wave * 1024u;
    if (t < nfull && allv[H]) {
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
          : "v"(p[H][0]), "v"(p[H][1]), "s"(dst)
          : "memory", "scc");
    } else {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bool ok = ov[H][s] && (kb[s] + t * 64) < k_limit;
        const void* src = ok ? (const void*)p[H][s] : (const void*)g_nst_zero16;
        glds16(src, __builtin_amdgcn_readfirstlane(dst + (uint32_t)s * 8192u));
      }
    }
    p[H][0] += step_bytes;
    p[H][1] += step_bytes;
  }
};

// Fragment addressing inside a 16 KB half-tile image.  ip = 16-wide block (0..7) of the image's 128 outer indices, kh = half
// of the K step.  Everything that depends on the lane is computed once (off*); a read adds compile-time constants.
template <int MODE>
struct Frag256;
template <>
struct Frag256<MODE_RC> {   // [128 rows][64 k] image, 16-byte slot ^= (row >> 1) & 7; fragments by ds_read_b128
  int off0, off1;
  __device__ __forceinline__ void init(int lane) {
    const int l15 = lane & 15, g = lane >> 4, sw = (l15 >> 1) & 7;
    off0 = l15 * KBYTES + ((g ^ sw) << 4);
    off1 = off0 ^ 64;
  }
  __device__ __forceinline__ bf16x8_t read(const char* img, int ip, int kh) const {
    return *reinterpret_cast<const bf16x8_t*>(img + (kh ? off1 : off0) + ip * (16 * KBYTES));
  }
};
template <>
struct Frag256<MODE_OC> {   // [64 k][128 cols] image, 32-byte pair ^= (r & 3) | ((r >> 3) & 1) << 2; ds_read_b64_tr_b16
  int q;
  __device__ __forceinline__ void init(int lane) {
    const int ii = lane & 15, g = lane >> 4;
    const int sidx = (ii >> 2) | ((g & 1) << 2);
    q = ((8 * g + (ii >> 2)) << 8) | (sidx << 5) | ((ii & 3) << 3);
  }
  __device__ __forceinline__ bf16x8_t read(const char* img, int ip, int kh) const {
    typedef short4_t __attribute__((address_space(3))) * lds_ptr_t;
    const char* a = img + (q ^ (ip << 5)) + kh * 8192;
    const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(a));
    const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(a + 1024));
    typedef __attribute__((ext_vector_type(8))) short short8_t;
    return __builtin_bit_cast(bf16x8_t, (short8_t)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  }
};

__device__ __forceinline__ void g256_wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xc07f); }

// Main loop of one unit: acc (+)= A[m0.., k-range] . B[n0.., k-range]; cs = column sums of the B operand (do_cs, wave-uniform).
// DBG: always 0 (it selected the timing ablations of DESIGN 5e -- no MFMAs / no DMA after the prologue / no fragment reads --
// whose branches are gone; the parameter keeps the instantiation names of the earlier profiles).
// da / db: DMA cursors of the two operands, positioned on the unit's first K step; cursor.issue<H>(t, image, wave) stages
// half-tile H of the unit's K step t (called with t = 0, 1, 2, ... in order for either half).  AMODE / BMODE: layout of their
// half-tile images (fragment readers).
// The loads a unit starts with: K step 0 entirely, B of K step 1 (also callable ahead of the main loop -- PRE -- while the
// previous unit's epilogue runs, provided that epilogue keeps out of buffer 0 and of buffer 1's B images).
template <typename DA, typename DB>
__device__ __forceinline__ void gemm256_prologue(char* smem, DA& da, DB& db, int nk) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  typedef __attribute__((address_space(3))) char* lds_char_ptr;
  const uint32_t smem_addr = (uint32_t)(uintptr_t)((lds_char_ptr)smem);
  da.template issue<0>(0, smem_addr + 0 * G256_HALF_BYTES, wave);
  da.template issue<1>(0, smem_addr + 1 * G256_HALF_BYTES, wave);
  db.template issue<0>(0, smem_addr + 2 * G256_HALF_BYTES, wave);
  db.template issue<1>(0, smem_addr + 3 * G256_HALF_BYTES, wave);
  if (nk > 1) {
    db.template issue<0>(1, smem_addr + G256_KT_BYTES + 2 * G256_HALF_BYTES, wave);
    db.template issue<1>(1, smem_addr + G256_KT_BYTES + 3 * G256_HALF_BYTES, wave);
  }
}

template <int AMODE, int BMODE, bool CS, int DBG = 0, bool PRE = false, typename DA, typename DB>
__device__ __forceinline__ void gemm256_mainloop(char* smem, DA& da, DB& db, int nk, bool do_cs, floatx4_t (&acc)[2][4][4],
                                                 floatx4_t (&cs)[4]) {
  typedef bf16_t T;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  typedef __attribute__((address_space(3))) char* lds_char_ptr;
  const uint32_t smem_addr = (uint32_t)(uintptr_t)((lds_char_ptr)smem);
  Frag256<AMODE> fa;
  Frag256<BMODE> fb;
  fa.init(lane);
  fb.init(lane);

#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[h][i][j] = floatx4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) cs[j] = floatx4_t{0.f, 0.f, 0.f, 0.f};
  const bf16x8_t ones = ones_frag<T>();

  // ---- prologue: K step 0 entirely, B of K step 1
  if constexpr (PRE) {
    // issued by the caller in front of the previous unit's epilogue, whose stores sit between those loads and this point in
    // the counter: drain it (and this wave's epilogue scratch reads, before anybody restages buffer 1's A images)
    wait_vmcnt<0>();
    g256_wait_lgkm0();
  } else {
    gemm256_prologue(smem, da, db, nk);
    if (nk > 1) wait_vmcnt<4>();
    else wait_vmcnt<0>();
  }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();   // group 1 runs half a phase behind from here on
  __builtin_amdgcn_sched_barrier(0);

  const int bip = (wc & 1) * 4;   // first 16-column block of this wave inside its B half-tile
#pragma unroll 1
  for (int t = 0; t < nk; ++t) {
    const uint32_t cur = (uint32_t)(t & 1) * G256_KT_BYTES, nxt = G256_KT_BYTES - cur;
    const char* Ah = smem + cur + wr * G256_HALF_BYTES;
    const char* Bh = smem + cur + (2 + (wc >> 1)) * G256_HALF_BYTES;
    const bool dma_on = true;
    bf16x8_t blo[2][2], bhi[2][2], af[4][2];

    // ---------------- phase 0
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) blo[j][kh] = fb.read(Bh, bip + j, kh);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) af[i][kh] = fa.read(Ah, i, kh);
    __builtin_amdgcn_sched_barrier(0);
    if (dma_on && t + 1 < nk) da.template issue<0>(t + 1, smem_addr + nxt + 0 * G256_HALF_BYTES, wave);
    g256_wait_lgkm0();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[0][i][j] = Mma<T>::run(af[i][kh], blo[j][kh], acc[0][i][j]);
    if (CS && do_cs) {
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int j = 0; j < 2; ++j) cs[j] = Mma<T>::run(ones, blo[j][kh], cs[j]);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // ---------------- phase 1
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) bhi[j][kh] = fb.read(Bh, bip + 2 + j, kh);
    __builtin_amdgcn_sched_barrier(0);
    if (dma_on && t + 1 < nk) da.template issue<1>(t + 1, smem_addr + nxt + 1 * G256_HALF_BYTES, wave);
    g256_wait_lgkm0();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[0][i][2 + j] = Mma<T>::run(af[i][kh], bhi[j][kh], acc[0][i][2 + j]);
    if (CS && do_cs) {
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int j = 0; j < 2; ++j) cs[2 + j] = Mma<T>::run(ones, bhi[j][kh], cs[2 + j]);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // ---------------- phase 2
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) af[i][kh] = fa.read(Ah, 4 + i, kh);
    __builtin_amdgcn_sched_barrier(0);
    if (dma_on && t + 2 < nk) db.template issue<0>(t + 2, smem_addr + cur + 2 * G256_HALF_BYTES, wave);
    g256_wait_lgkm0();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[1][i][2 + j] = Mma<T>::run(af[i][kh], bhi[j][kh], acc[1][i][2 + j]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // ---------------- phase 3
    if (dma_on && t + 2 < nk) {
      db.template issue<1>(t + 2, smem_addr + cur + 3 * G256_HALF_BYTES, wave);
      wait_vmcnt<4>();   // B0 / B1 of step t + 2 stay in flight; everything of step t + 1 has landed
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[1][i][j] = Mma<T>::run(af[i][kh], blo[j][kh], acc[1][i][j]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();   // pairs with group 1's last barrier: nobody reads the stage buffers any more
  asm volatile("" ::: "memory");

}

// -----------------------------------------------------------------------------------------------------------------------
// Grouped weight gradients: MANY products dW_p[M_p, N_p] (+)= X_p^T . dZ_p (both operands reduction-major bf16, f32 output)
// in ONE launch.  The weight gradients of a layer stack do not feed the backward chain, so they can all wait until the stack
// is done; with every 256 x 256 output tile of every product as a unit of the same grid the chip fills WITHOUT split-K: no
// partial-sum slabs, no second stage, and a unit runs the whole reduction (hundreds of K steps) behind one prologue.
// (Stand-alone, one 256 x 2048 gradient over 28 800 rows needs 32 K slices to fill 256 CUs: 67 MB of slabs written and read
// next to 133 MB of operands, which is why the 256 x 256 kernel LOSES to the 128 x 128 one there -- profiles/r04.)
//
// Unit order: the host sorts the products by reduction length (longest first) and the kernel deals "bundles" of 8 consecutive
// logical units round-robin to the 8 XCDs (workgroup b runs on XCD b % 8): the column tiles of one product share its X rows,
// sit in one bundle, start together on one XCD and hit its L2; every XCD gets the same mix of long and short units.
// -----------------------------------------------------------------------------------------------------------------------
constexpr int G256_MAX_PROBLEMS = 56;
struct G256Problem {       // 72 bytes
  const bf16_t* A;         // X  [K rows][lda], output rows M = its columns
  const bf16_t* B;         // dZ [K rows][ldb], output columns N = its columns
  float* C;                // dW [M][ldc]
  float* colsum;           // db [N] or NULL
  int lda, ldb, ldc;
  int M, N, K;
  int tiles_n;
  int unit_end;            // logical units [unit_end of the previous problem, unit_end) belong to this one
  int flags;               // 1: C += , 2: colsum +=
  int reserved;
};
struct G256GroupArgs {
  int nprob, nunits;
  const G256Problem* ext;   // != NULL: the table lives in device memory (more than G256_MAX_PROBLEMS products), p[] is unused
  G256Problem p[G256_MAX_PROBLEMS];
};

__device__ __forceinline__ void gemm256_group_block(char* smem) {
  typedef bf16_t T;
  typedef DenseLoader<T> Loader;
  const NST_AS4 G256GroupArgs* ka = (const NST_AS4 G256GroupArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  // workgroup -> logical unit: bundle ((k / 8) * 8 + xcd), member k % 8, with xcd = b % 8 and k = b / 8
  const int b = blockIdx.x, xcd = b & 7, k = b >> 3;
  const int unit = (((k >> 3) << 3) + xcd) * 8 + (k & 7);
  if (unit >= ka->nunits) return;   // (the grid is padded to whole rounds of 8 bundles; uniform for the workgroup)
  // the table is read with scalar loads either way (it does not change while the kernel runs)
  const NST_AS4 G256Problem* tab = ka->ext ? (const NST_AS4 G256Problem*)(uintptr_t)ka->ext : &ka->p[0];
  int pi = 0, ubase = 0;
  {
    const int np = ka->nprob;
#pragma unroll 1
    for (; pi < np - 1; ++pi) {
      const int ue = tab[pi].unit_end;
      if (unit < ue) break;
      ubase = ue;
    }
  }
  const NST_AS4 G256Problem* pp = &tab[pi];
  const int tile = unit - ubase, tiles_n = pp->tiles_n;
  const int tm = tile / tiles_n;
  const int m0 = tm * G256_TILE, n0 = (tile - tm * tiles_n) * G256_TILE;
  const int M = pp->M, N = pp->N, Kd = pp->K;
  const bool do_cs = pp->colsum != nullptr && m0 == 0 && wr == 0;   // wave-uniform
  floatx4_t acc[2][4][4], cs[4];
  {
    Loader la, lb;   // reduction-major operands: element (r, i) at base[r * ld + i]
    la.base = pp->A; la.ld = pp->lda; la.outer_limit = Kd; la.contig_limit = M; la.vec = 1;
    lb.base = pp->B; lb.ld = pp->ldb; lb.outer_limit = Kd; lb.contig_limit = N; lb.vec = 1;
    Dma256<MODE_OC> da, db;
    da.init(la, m0, 0, wave, lane);
    db.init(lb, n0, 0, wave, lane);
    gemm256_mainloop<MODE_OC, MODE_OC, true>(smem, da, db, (Kd + 63) >> 6, do_cs, acc, cs);
  }
  asm volatile("" ::: "memory");
  const NST_AS4 G256Problem* p2 = launder(pp);
  const int flags = p2->flags;
  if (do_cs && lane < 16) {
    float* csd = p2->colsum;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n0 + wc * 64 + j * 16 + lane;
      if (col < N) csd[col] = (flags & 2) ? csd[col] + cs[j][0] : cs[j][0];
    }
  }
  Epilogue ep{};   // only .vec is read by the compile-time epilogues used here
  ep.vec = 1;
  float* epi = reinterpret_cast<float*>(smem + wave * V3_EPI_BYTES_PER_WAVE);
  float* C = p2->C;
  const int64_t ldc = p2->ldc;
  const IdentityRowMap rowmap;
  if (flags & 1) {
    epilogue_v3<float, IdentityRowMap, EF_ACCUM>(acc[0], epi, C, ldc, M, N, m0 + wr * 128, n0 + wc * 64, ep, rowmap, lane);
    epilogue_v3<float, IdentityRowMap, EF_ACCUM>(acc[1], epi, C, ldc, M, N, m0 + wr * 128 + 64, n0 + wc * 64, ep, rowmap, lane);
  } else {
    epilogue_v3<float, IdentityRowMap, 0>(acc[0], epi, C, ldc, M, N, m0 + wr * 128, n0 + wc * 64, ep, rowmap, lane);
    epilogue_v3<float, IdentityRowMap, 0>(acc[1], epi, C, ldc, M, N, m0 + wr * 128 + 64, n0 + wc * 64, ep, rowmap, lane);
  }
}

}  // namespace nstgemm
