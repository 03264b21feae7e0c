// Fused position-wise feed-forward pair for d_model = 256 (speech_transformer_s / transformer sets with d = 256), bf16.
//
//   forward  (neurst/layers/common_layers.py:145-160 inside the wrapper :73-85):
//       H = dropout1(relu(X @ W1 + b1))            [M, F]   written once (saved for backward), never re-read
//       Y = residual + dropout2(H @ W2 + b2)       [M, 256]
//   backward (input gradient; the weight gradients stay separate reductions over M):
//       dH = (dY @ W2^T) * gate(H)                 [M, F]   written once for the dense1 weight / bias gradient
//       dX = dH @ W1^T (+ residual)                [M, 256]
//
// Both are the SAME two-GEMM chain  Z = mid(Xin @ WA^T) ; Out = Z @ WB^T  with
//       WA [F][256]  (row = hidden unit, the 256-long reduction contiguous)   forward: W1^T     backward: W2
//       WB [256][F]  (row = output column, hidden units contiguous)           forward: W2^T     backward: W1
// so one kernel template serves both; the forward reads transposed bf16 copies of the two FFN weights (made once per
// optimizer step by nst_transpose_bf16), the backward reads the weights as stored.
//
// Why one kernel: with d_model = 256 the two GEMMs of an FFN are HBM / launch bound on the [M, F] hidden tensor
// (118 MB per encoder layer at the benchmark shape): separate kernels write it, read it, and each has only 4 K steps
// (dense1) or a 256-wide output (dense2).  Here a workgroup owns 128 rows, walks the hidden dimension in chunks of 64
// units, and the hidden tile stays on chip between the products (ffn_pair8_kernel below: eight waves, two per SIMD; the
// four-wave kernel of round 2 that this header used to describe is scripts/r06_experiments/ffn_pair_four_wave_kernel.hip.txt).
// Common to both: v_mfma_f32_32x32x16_bf16 with "transposed" accumulators (MFMA A operand = weights, B operand =
// activations), so a lane owns ONE row m = lane & 31 of its 32-row group and 16 CONSECUTIVE columns per 32-column block; the
// A-operand row r of a 32-row block holds hidden unit pi(r) = 16*((r >> 2) & 1) + 4*(r >> 3) + (r & 3), which makes the MFMA
// result rows 8q + 4h + i of lane half h the 16 consecutive hidden units 16h + 4q + i -- bias, ReLU, one Philox call per 8
// units, and the packed bf16 pairs ARE the B operand of the second product; weights arrive by LDS-DMA
// (global_load_lds_dwordx4, no VGPR staging) with the XOR swizzle applied to the DMA source address and to the fragment read.
//
// Round 6: three exits for the 128 x 256 output tile (template parameter OUTM): the final epilogue (bias, dropout, residual,
// bf16 store), the row phase of nst_rowphase.h behind an LDS tile (the wrapper's LayerNorm stages: nst_ffn_add_layernorm_fwd /
// nst_ffn_layernorm_bwd), or raw f32 partial sums of a SLICE of the hidden dimension (gridDim.y slices per row tile, for row
// counts that give fewer tiles than CUs; ffn_slab_rows_kernel adds the slabs and runs the row phase).
#include "nst_common.h"
#include "nst_rowphase.h"

#include <stdlib.h>

#include <utility>

namespace {

typedef __attribute__((ext_vector_type(16))) float floatx16_t;

constexpr int D = 256;            // d_model
constexpr int CH = 128;           // hidden units per chunk
constexpr int PIECE = 16384;      // bytes per ring slot
enum { MODE_FWD = 0, MODE_BWD = 1 };

struct FfnArgs {
  const bf16_t* xin;       // [M, 256]
  const bf16_t* wa;        // [F, 256]
  const bf16_t* wb;        // [256, F]
  const float* bias_a;     // [F]    forward: b1
  const float* bias_b;     // [256]  forward: b2
  const bf16_t* residual;  // [M, 256] or null
  const bf16_t* gate;      // [M, F]   backward: saved activation H
  bf16_t* mid_out;         // [M, F]   forward: H, backward: dH
  bf16_t* out;             // [M, 256]
  const uint64_t* seed_dev;  // optional device scalar added to both seeds (graph replay: the step counter)
  int M, F;
  uint32_t drop1_thresh, drop2_thresh;
  float drop1_inv_keep, drop2_inv_keep;
  float gate_scale;
  uint64_t seed1, stream1, seed2, stream2;
  int rot_mode;   // v2: 0 = chunk order rotated per workgroup, 1 = per XCD (workgroup id & 7), 2 = none
  // v2: the gate of the backward as one bit per hidden element (opaque layout, see ffn_pair8_kernel): written by the forward
  // when non-null, read by the backward INSTEAD of the saved activation when non-null
  uint16_t* gate_bits;
  // LN variants of the eight-wave kernel (nst_ffn_add_layernorm_fwd / nst_ffn_layernorm_bwd): the output tile goes through LDS to
  // the row phase of nst_rowphase.h instead of to `out` directly
  rowphase::RowEpi rp;
  uint64_t rp_seed;
  // OUTM == 2 (hidden dimension split over gridDim.y workgroups per row tile): the raw f32 partial sums of the second product,
  // [gridDim.y][M][256]; ffn_slab_rows_kernel adds the slabs and runs the row phase
  float* slab;
};

__device__ __forceinline__ int pi32(int r) { return (((r >> 2) & 1) << 4) + ((r >> 3) << 2) + (r & 3); }

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// One piece share of a wave: IPW LDS-DMA instructions (64 lanes x 16 bytes each -> LDS [dst + i*1024 + lane*16]) from
// sbase + voff[i] (per lane), issued back to back from one asm block (M0 carries the LDS address; the s_nop 4 covers an
// SGPR base the compiler may have produced with v_readfirstlane right in front of the statement).
template <int IPW, bool NT = false>
__device__ __forceinline__ void glds_piece(const void* sbase, const uint32_t (&voff)[IPW], uint32_t lds_addr_uniform) {
  uint32_t keep;
  if constexpr (NT && IPW == 4) {   // streamed-once data (the backward's gate rows): non-temporal, keeps the weights in L2
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %6\n\t"
        "s_nop 4\n\t"
        "global_load_lds_dwordx4 %1, %5 nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %5 nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, %5 nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %4, %5 nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(sbase), "s"(lds_addr_uniform)
        : "memory", "scc");
  } else if constexpr (IPW == 4) {
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %6\n\t"
        "s_nop 4\n\t"
        "global_load_lds_dwordx4 %1, %5\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %5\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, %5\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %4, %5\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(sbase), "s"(lds_addr_uniform)
        : "memory", "scc");
  } else {
    static_assert(IPW == 8, "4 or 8 DMA instructions per wave and piece");
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %10\n\t"
        "s_nop 4\n\t"
        "global_load_lds_dwordx4 %1, %9\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %9\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, %9\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %4, %9\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %5, %9\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %6, %9\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %7, %9\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %8, %9\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "v"(voff[4]), "v"(voff[5]), "v"(voff[6]), "v"(voff[7]),
          "s"(sbase), "s"(lds_addr_uniform)
        : "memory", "scc");
  }
}

// one LDS-DMA instruction: 64 lanes x 16 bytes from sbase + voff (per lane) -> LDS [lds_addr_uniform + lane * 16]
__device__ __forceinline__ void glds_one(const void* sbase, uint32_t voff, uint32_t lds_addr_uniform) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 4\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_addr_uniform)
      : "memory");
}
// the 4-byte form: 64 lanes x 4 bytes -> LDS [lds_addr_uniform + lane * 4]
__device__ __forceinline__ void glds_one_dword(const void* sbase, uint32_t voff, uint32_t lds_addr_uniform) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 4\n\t"
      "global_load_lds_dword %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_addr_uniform)
      : "memory");
}

// =====================================================================================================================
// v2 (round 3): the same two-GEMM chain on EIGHT waves per workgroup -- two per SIMD -- so that one wave's MFMAs run under
// its partner's epilogue VALU work, fragment reads and waits (the one-wave-per-SIMD kernel above serialises them:
// MFMA 29 us + fragment reads / epilogue 28 us + DMA / stores 26 us of an 87 us launch, profiles/r02_ffn_bench_ablation.json).
//
//   workgroup = 128 rows = 4 row groups of 32 rows; wave w: row group rg = w & 3, half hh = w >> 2 (waves w and w + 4 share
//   a SIMD and a row group).  The hidden dimension is walked in chunks of 64 units:
//     first product  (A phase): wave (rg, hh) computes hidden units 32 hh .. 32 hh + 31 of the chunk for its 32 rows
//                               (16 MFMAs 32x32x16 over K = 256, X fragments resident in 64 registers);
//     mid epilogue            : bias + ReLU + dropout, packed to bf16 -> the P tile [128 rows][64 units] in LDS: the two
//                               halves of a row group exchange their hidden halves through it, and the saved activation is
//                               stored to HBM FROM it in whole 128-byte row segments;
//     second product (B phase): wave (rg, hh) computes output columns 128 hh .. 128 hh + 127 over all 64 units of the chunk
//                               (4 blocks x 4 k-steps = 16 MFMAs; B operand = P fragments, A operand = W2 fragments).
//   256 accumulator/operand registers per wave: X 64 + output 64 + hidden 16 + packed tile 8 + fragments.
//   Software pipeline: iteration c runs  A(c+1) | barrier Y | B(c) interleaved with the mid epilogue of c+1 | barrier X |
//   P(c+1) write + DMA issue -- so the epilogue VALU sits between the second product's MFMAs, and every barrier has a phase of
//   matrix work on either side.
//   LDS (152 KB): W1 chunks double buffered 2 x 32 KB [64 units][256 k] (slot ^= row & 15 within the low 16 slots),
//   W2 chunks 2 x 32 KB [256 out][64 units] and P 16 KB [128 rows][64 units] (128-byte rows, slot ^= (row >> 1) & 7),
//   first-layer bias (pre-scaled) <= 8 KB.  Weights arrive by LDS-DMA a full iteration ahead (issued behind barrier X of
//   iteration c for iterations c+2 / c+3, waited for -- counted, behind the two hidden-tile stores -- before barrier X of
//   iteration c+1).  The hidden-unit permutation pi of the kernel above gives every lane 16 consecutive units / columns.
// =====================================================================================================================
constexpr int CH2 = 64;
constexpr int V2_W1 = 0, V2_W2 = 2 * 32768, V2_P = 4 * 32768, V2_BIAS = V2_P + 16384;
constexpr int V2_ROWS = 128;

// Gate bits (BITS): the backward needs of the saved activation only its sign test (hidden > 0).  The forward packs that test
// for each lane's 16 hidden units of a chunk into 16 bits (unit 2k -> bit 7-k, unit 2k+1 -> bit 15-k, taken from the SAME bf16
// values it stores) and writes them as gate_bits[(2 chunk + hh) * M + row][h] (uint16): one 128-byte store per wave and chunk.
// The backward then fetches 4 bytes per row and half chunk (one LDS-DMA instruction per wave and chunk) instead of 64:
// 7.4 MB instead of 118 MB at the benchmark shape, and the weight-gradient stream running beside it keeps that bandwidth.
// OUTM: 0 = the output tile leaves through the final epilogue below; 1 = through LDS to the row phase (LayerNorm stages of the
// wrapper, nst_rowphase.h); 2 = the workgroup walks only ITS share of the hidden chunks (blockIdx.y of gridDim.y) and stores raw
// f32 partial sums to a slab -- fewer row tiles than CUs (the decoder's 9 600 rows = 75 tiles) then still fill the chip
template <int MODE, int DROP, bool FULL, int DBG = 0, bool BITS = false, int OUTM = 0>
__global__ void __launch_bounds__(512, 2) ffn_pair8_kernel(FfnArgs a) {
  constexpr bool LN = OUTM == 1;
  static_assert(MODE == MODE_FWD || DROP == 0, "the backward has no dropout of its own (the gate carries the forward's mask)");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) char* lds_char_ptr;
  const uint32_t smem_addr = (uint32_t)(uintptr_t)((lds_char_ptr)smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave & 3, hh = wave >> 2;
  const int i_l = lane & 31, h = lane >> 5;
  const int F = a.F, M = a.M;
  const int nch_all = F / CH2;
  const int c_base = OUTM == 2 ? (int)((blockIdx.y * nch_all) / gridDim.y) : 0;
  const int nch = OUTM == 2 ? (int)(((blockIdx.y + 1) * nch_all) / gridDim.y) - c_base : nch_all;   // >= 2 (host)
  const int m0 = blockIdx.x * V2_ROWS;
  // rotated chunk order: the chip writes all columns of the hidden tensor at once (mode 0: per workgroup; mode 1: per XCD --
  // the 28 workgroups of an XCD then stream the SAME weight chunk through their shared L2 at about the same time)
  const int rot = a.rot_mode == 0 ? (int)(blockIdx.x % nch) : (a.rot_mode == 1 ? (int)(blockIdx.x & 7) * (nch >> 3) % nch : 0);
  auto phys = [&](int c) { const int q = c + rot; return c_base + (q >= nch ? q - nch : q); };
  const int row = m0 + rg * 32 + i_l;   // this lane's row of Xin / outputs
  const bool row_ok = FULL || row < M;
  const int row_c = row_ok ? row : M - 1;

  uint64_t seed_off = 0;
  if (MODE == MODE_FWD && DROP != 0) seed_off = seed_with_offset(0, a.seed_dev);

  // ---------------------------------------------------------------- DMA source offsets (per lane, constant)
  uint32_t voff1[4], voff2[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int t = wave * 4 + s;
    {
      const int r = 2 * t + (lane >> 5), sl = lane & 31;                       // W1 chunk: 512-byte rows, 32 slots
      voff1[s] = (uint32_t)(r * (D * 2) + ((sl ^ (r & 15)) << 4));
    }
    {
      const int r = 8 * t + (lane >> 3), sl = lane & 7;                        // W2 chunk: 128-byte pieces of 2F-byte rows
      voff2[s] = (uint32_t)r * (uint32_t)(F * 2) + (uint32_t)((sl ^ ((r >> 1) & 7)) << 4);
    }
  }
  auto issue_w1 = [&](int c_logical) {
    const int c = phys(c_logical);
    glds_piece<4>(reinterpret_cast<const char*>(a.wa) + (int64_t)c * CH2 * D * 2, voff1,
                  smem_addr + V2_W1 + (uint32_t)(c_logical & 1) * 32768u + (uint32_t)wave * 4096u);
  };
  auto issue_w2 = [&](int c_logical) {
    const int c = phys(c_logical);
    glds_piece<4>(reinterpret_cast<const char*>(a.wb) + (int64_t)c * CH2 * 2, voff2,
                  smem_addr + V2_W2 + (uint32_t)(c_logical & 1) * 32768u + (uint32_t)wave * 4096u);
  };
  issue_w1(0);
  issue_w2(0);
  if (nch > 1) issue_w1(1);

  // ---------------------------------------------------------------- Xin fragments (B operand of the first product)
  bf16x8_t xf[16];
  {
    const bf16_t* xr = a.xin + (int64_t)row_c * D + h * 8;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) xf[kb] = *reinterpret_cast<const bf16x8_t*>(xr + kb * 16);
  }
  float* bias_lds = reinterpret_cast<float*>(smem + V2_BIAS);
  if constexpr (MODE == MODE_FWD) {
    const float sc = (DROP & 1) ? a.drop1_inv_keep : 1.0f;   // relu(z) * inv_keep == relu(z * inv_keep)
    for (int i = tid; i < F; i += 512) bias_lds[i] = a.bias_a ? a.bias_a[i] * sc : 0.f;
  }
  // backward: the gate (the saved activation of this wave's 32 rows x 32 units of a chunk, 2 KB) arrives by LDS-DMA in a
  // region PRIVATE to the wave (the 16 KB the forward's bias occupies): issued by the wave right after its epilogue has read
  // the previous chunk's gate, waited for (counted) by the wave before the next epilogue -- no barrier involved
  const uint32_t g_lds = smem_addr + V2_BIAS + (uint32_t)wave * 2048u;
  uint32_t voffg[2] = {0u, 0u};
  if constexpr (MODE == MODE_BWD) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      int rgate = m0 + 32 * rg + 16 * q + (lane >> 2);
      rgate = rgate < M ? rgate : M - 1;
      voffg[q] = (uint32_t)rgate * (uint32_t)(F * 2) + (uint32_t)((lane & 3) << 4);
    }
    if constexpr (BITS) voffg[0] = (uint32_t)row_c * 4u;   // the row's 32 gate bits of a half chunk
  }
  auto issue_g = [&](int c_logical) {   // chunk index clamped: a copy past the end lands in the region nobody reads again
    const int c = phys(c_logical < nch ? c_logical : nch - 1);
    if constexpr (BITS) {
      glds_one_dword(reinterpret_cast<const char*>(a.gate_bits) + (int64_t)(2 * c + hh) * M * 4, voffg[0], g_lds);
      return;
    }
    const char* src = reinterpret_cast<const char*>(a.gate) + ((int64_t)c * CH2 + 32 * hh) * 2;
    glds_one(src, voffg[0], g_lds);
    glds_one(src, voffg[1], g_lds + 1024u);
  };
  if constexpr (MODE == MODE_BWD) issue_g(0);

  // ---------------------------------------------------------------- LDS read offsets (per lane, constant)
  const int pr = pi32(i_l);
  // W1 fragment (k-block kb): row R = 32 hh + pr, slot (2 kb + h) ^ (R & 15) on the low 4 slot bits:
  //   byte = R * 512 + (kb >> 3) * 256 + ((((2 kb) & 15) | h) ^ (pr & 15)) * 16  =  offA ^ (((2 kb) & 15) << 4)  + (kb >> 3) * 256
  const uint32_t offA = (uint32_t)((32 * hh + pr) * 512 + ((h ^ (pr & 15)) << 4));
  // W2 fragment (output block ob, k-step ks): row 128 hh + 32 ob + pr, slot (2 ks + h) ^ ((pr >> 1) & 7)
  uint32_t offB[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) offB[ks] = (uint32_t)((128 * hh + pr) * 128 + (((2 * ks + h) ^ ((pr >> 1) & 7)) << 4));
  // P tile: row 32 rg + i_l, slot ^ ((i_l >> 1) & 7)
  const uint32_t pz = (uint32_t)((i_l >> 1) & 7);
  const uint32_t offP = (uint32_t)((32 * rg + i_l) * 128);
  // hidden-tile store: this wave writes rows 32 rg + 16 hh + 8 q + (lane >> 3), q = 0, 1; 8 lanes = one 128-byte row piece
  const int st_r0 = 32 * rg + 16 * hh + (lane >> 3), st_slot = lane & 7;
  const bool st_ok0 = FULL || (m0 + st_r0) < M, st_ok1 = FULL || (m0 + st_r0 + 8) < M;

  floatx16_t accH, accY[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int v = 0; v < 16; ++v) accY[i][v] = 0.f;
  uint32_t packed[8];
  if constexpr (DBG != 0) {   // ablation builds (NST_FFN_ABLATION): stages may be missing, their registers still hold something defined
#pragma unroll
    for (int v = 0; v < 16; ++v) accH[v] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) packed[e] = 0u;
  }

  // ---- first product of chunk c (logical): 16 dependent MFMAs (same accumulator: the matrix core forwards it)
  auto a_phase = [&](int c_logical) {
    const char* w1 = smem + V2_W1 + (c_logical & 1) * 32768;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
      const uint32_t o = (offA ^ (uint32_t)(((2 * kb) & 15) << 4)) + (uint32_t)((kb >> 3) * 256);
      const bf16x8_t wf = *reinterpret_cast<const bf16x8_t*>(w1 + o);
      if (kb == 0) {
        const floatx16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        accH = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf[0], zero, 0, 0, 0);
      } else {
        accH = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf[kb], accH, 0, 0, 0);
      }
    }
  };
  // ---- mid epilogue of chunk c: accH -> packed (bias, ReLU, dropout, bf16)
  auto mid_epilogue = [&](int c_logical, auto first_tag) {   // first_tag: the prologue's call (everything has been drained)
    const int c = phys(c_logical);
    const int col0 = c * CH2 + 32 * hh + 16 * h;   // this lane's 16 consecutive hidden units
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = accH[e];
    if constexpr ((DBG & 32) != 0) {   // ablation: the conversion only
#pragma unroll
      for (int e = 0; e < 8; ++e) packed[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
      if constexpr (MODE == MODE_BWD) { if constexpr (decltype(first_tag)::value || !FULL) wait_vm<0>(); issue_g(c_logical + 1); }
      return;
    }
    if constexpr (MODE == MODE_BWD) {
      // this lane's 16 gate values: row i_l of the wave's region (64-byte rows), bytes [32 h, +32)
      if constexpr (decltype(first_tag)::value || !FULL) wait_vm<0>(); else wait_vm<10>();   // (2 stores + 8 weight DMAs are younger than the gate copy)
      if constexpr (BITS) {
        // bit -> all-ones / all-zeros word (v_bfe_i32) ANDed onto the scaled value: two VALU operations per element, no VCC
        const int m = (int)(*reinterpret_cast<const uint32_t*>(smem + V2_BIAS + wave * 2048 + lane * 4) >> (16 * h));
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          uint32_t keep;   // (asm: the compiler rewrites a plain sign-extended bit into and / compare / select, five operations per pair)
          asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(keep) : "v"(m), "n"((7 - (e >> 1)) + 8 * (e & 1)));
          v[e] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v[e] * a.gate_scale) & keep);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) packed[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue_g(c_logical + 1);
        return;
      }
      const uint4* gp = reinterpret_cast<const uint4*>(smem + V2_BIAS + wave * 2048 + i_l * 64 + h * 32);
      union { uint4 u[2]; short s[16]; } g;
      g.u[0] = gp[0];
      g.u[1] = gp[1];
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = g.s[e] > 0 ? v[e] * a.gate_scale : 0.f;   // bf16 > 0 <=> its bits as int16 > 0
#pragma unroll
      for (int e = 0; e < 8; ++e) packed[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the gate reads have returned: the region may be overwritten
      issue_g(c_logical + 1);
      return;
    }
    const float4* bp = reinterpret_cast<const float4*>(bias_lds + col0);
    const float sc = (DROP & 1) ? a.drop1_inv_keep : 1.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 b = bp[q];
      v[4 * q + 0] = fmaxf(fmaf(v[4 * q + 0], sc, b.x), 0.f);
      v[4 * q + 1] = fmaxf(fmaf(v[4 * q + 1], sc, b.y), 0.f);
      v[4 * q + 2] = fmaxf(fmaf(v[4 * q + 2], sc, b.z), 0.f);
      v[4 * q + 3] = fmaxf(fmaf(v[4 * q + 3], sc, b.w), 0.f);
    }
    if constexpr ((DROP & 1) != 0) {
      const uint64_t idx = (uint64_t)row * (uint64_t)F + (uint64_t)col0;   // multiple of 8
      const Philox4 r0 = philox4x32_10(a.seed1 + seed_off, a.stream1, idx >> 3);
      const Philox4 r1 = philox4x32_10(a.seed1 + seed_off, a.stream1, (idx >> 3) + 1);
      const uint32_t w[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const uint32_t f = (e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu);
        v[e] = f >= a.drop1_thresh ? v[e] : 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) packed[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
    if constexpr (BITS) {   // the values are +x, +0 or -0: "bf16 > 0" <=> the low 15 bits are not all zero
      uint32_t acc = 0;
      const uint32_t ones = 0x00010001u;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        uint32_t nz;   // min(half, 1) per 16-bit half: 0 / 1 (asm: the generic vector min is legalised into compares and selects)
        asm("v_pk_min_u16 %0, %1, %2" : "=v"(nz) : "v"(packed[k] & 0x7fff7fffu), "s"(ones));
        acc = (acc << 1) | nz;
      }
      const uint32_t m16 = (acc & 0xffu) | ((acc >> 8) & 0xff00u);
      if constexpr ((DBG & 1) == 0)
        if (row_ok) a.gate_bits[((int64_t)(2 * c + hh) * M + row) * 2 + h] = (uint16_t)m16;
    }
  };
  auto write_p = [&]() {   // this lane's 16 units = slots 4 hh + 2 h, + 1 of its row
    if constexpr ((DBG & 64) != 0) return;
    char* prow = smem + V2_P + offP;
    *reinterpret_cast<uint4*>(prow + ((((uint32_t)(4 * hh + 2 * h)) ^ pz) << 4)) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
    *reinterpret_cast<uint4*>(prow + ((((uint32_t)(4 * hh + 2 * h + 1)) ^ pz) << 4)) = make_uint4(packed[4], packed[5], packed[6], packed[7]);
  };
  // ---- the saved activation of chunk c, stored from the P tile in whole row pieces
  // (round 6: reading the two row pieces at the top of the span and storing them four MFMAs later -- no lgkmcnt(0) in front of
  // the stores -- measured no faster forward and 5 % slower backward on the same box: the 7 us that the stage ablation assigns
  // to these stores is the stores, not the wait in front of them)
  auto store_hidden = [&](int c_logical) {
    const int c = phys(c_logical);
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = st_r0 + 8 * q;
      const uint4 d = *reinterpret_cast<const uint4*>(smem + V2_P + r * 128 + ((st_slot ^ ((r >> 1) & 7)) << 4));
      bf16_t* o = a.mid_out + (int64_t)(m0 + r) * F + (c * CH2 + st_slot * 8);
      if constexpr ((DBG & 1) == 0)
        if (q == 0 ? st_ok0 : st_ok1) __builtin_nontemporal_store(u32x4_t{d.x, d.y, d.z, d.w}, reinterpret_cast<u32x4_t*>(o));
    }
  };
  auto barrier = [&]() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // ---------------------------------------------------------------- fragment stream of one span
  // A span (between two barrier pairs) multiplies the NEXT chunk's first product (stream positions 0..15 = k-block) and the
  // CURRENT chunk's second product (positions 16..31 = (k-step, output block)).  Weight fragments travel through a ring of PD
  // register quads read LOOK positions ahead of the MFMA that consumes them -- with one read in flight per wave the LDS
  // latency (not its bandwidth) paced the MFMAs: "no fragment reads" took 16 of 85 us (gpurun_out/r03_ffn_v2_ablation.log).
  constexpr int PD = 8, LOOK = 6;
  bf16x8_t wq[PD], pf[4];
  if constexpr (DBG != 0) {
#pragma unroll
    for (int i = 0; i < PD; ++i) wq[i] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) pf[i] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
  }
  auto read_frag = [&](auto stag, const char* w1, const char* w2) {
    constexpr int S = decltype(stag)::value;
    if constexpr ((DBG & 4) != 0) {
      return;
    } else if constexpr (S < 16) {
      const uint32_t o = (offA ^ (uint32_t)(((2 * S) & 15) << 4)) + (uint32_t)((S >> 3) * 256);
      wq[S % PD] = *reinterpret_cast<const bf16x8_t*>(w1 + o);
    } else {
      constexpr int ks = (S - 16) >> 2, ob = (S - 16) & 3;
      wq[S % PD] = *reinterpret_cast<const bf16x8_t*>(w2 + offB[ks] + ob * 4096);
    }
  };
  auto mfma_at = [&](auto stag) {
    constexpr int S = decltype(stag)::value;
    if constexpr ((DBG & 8) != 0) {
      return;
    } else if constexpr (S == 0) {
      const floatx16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      accH = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[0], xf[0], zero, 0, 0, 0);
    } else if constexpr (S < 16) {
      accH = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[S % PD], xf[S], accH, 0, 0, 0);
    } else {
      constexpr int ks = (S - 16) >> 2, ob = (S - 16) & 3;
      accY[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[S % PD], pf[ks], accY[ob], 0, 0, 0);
    }
  };
  auto read_p_frags = [&]() {
    if constexpr ((DBG & 64) != 0) return;
    const char* prow = smem + V2_P + offP;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      pf[ks] = *reinterpret_cast<const bf16x8_t*>(prow + ((((uint32_t)(2 * ks + h)) ^ pz) << 4));
    }
  };
  // the first LOOK fragments of a span, issued right behind barrier X (the P write, barrier Y and the DMA issue cover their latency)
  auto burst = [&](auto s0tag, const char* w1, const char* w2) {
    constexpr int S0 = decltype(s0tag)::value;
    [&]<int... I>(std::integer_sequence<int, I...>) {
      (read_frag(std::integral_constant<int, S0 + I>(), w1, w2), ...);
    }(std::make_integer_sequence<int, LOOK>());
  };
  // one span: HAS_A = it contains the first product of chunk c_next (all spans but the last)
  auto span = [&](auto has_a_tag, int c_cur, const char* w1, const char* w2) {
    constexpr bool HAS_A = decltype(has_a_tag)::value;
    constexpr int S0 = HAS_A ? 0 : 16;
    // the weights of the spans after the next: W1 chunk c_cur + 3 -> the buffer the PREVIOUS span's first product read,
    // W2 chunk c_cur + 2 -> the buffer its second product read (both free since barrier X).  One DMA instruction rides behind
    // each of the first eight MFMAs (issuing the eight back to back costs the wave 500-800 cycles with the matrix core idle).
    // Chunks past the end are clamped to the last one: the copy lands in a buffer nobody reads again, and the span stays
    // one basic block.
    const int cw1 = phys(c_cur + 2 < nch ? c_cur + 2 : nch - 1), cw2 = phys(c_cur + 1 < nch ? c_cur + 1 : nch - 1);
    const char* src1 = reinterpret_cast<const char*>(a.wa) + (int64_t)cw1 * CH2 * D * 2;
    const char* src2 = reinterpret_cast<const char*>(a.wb) + (int64_t)cw2 * CH2 * 2;
    const uint32_t dst1 = smem_addr + V2_W1 + (uint32_t)(c_cur & 1) * 32768u + (uint32_t)wave * 4096u;
    const uint32_t dst2 = smem_addr + V2_W2 + (uint32_t)((c_cur + 1) & 1) * 32768u + (uint32_t)wave * 4096u;
    read_p_frags();
    store_hidden(c_cur);
    [&]<int... SS>(std::integer_sequence<int, SS...>) {
      ([&] {
        constexpr int S = S0 + SS;
        mfma_at(std::integral_constant<int, S>());
        if constexpr (HAS_A && S >= 1 && S <= 8 && (DBG & 2) == 0) {
          if constexpr (S <= 4) glds_one(src1, voff1[S - 1], dst1 + (uint32_t)(S - 1) * 1024u);
          else glds_one(src2, voff2[S - 5], dst2 + (uint32_t)(S - 5) * 1024u);
        }
        if constexpr (S + LOOK < 32) read_frag(std::integral_constant<int, S + LOOK>(), w1, w2);
        if constexpr (HAS_A && S == 15) mid_epilogue(c_cur + 1, std::false_type());   // program order: behind the first product; scheduled into the slots below
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if constexpr (HAS_A && S >= 16) __builtin_amdgcn_sched_group_barrier(0x002, ((DROP & 1) ? 11 : 4) + (MODE == MODE_FWD && BITS ? 2 : 0), 0);
      }(), ...);
    }(std::make_integer_sequence<int, 32 - S0>());
  };

  // ---------------------------------------------------------------- prologue: chunk 0's first product and epilogue
  wait_vm<0>();
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the bias writes
  barrier();
  a_phase(0);
  mid_epilogue(0, std::true_type());
  barrier();                            // X of "span -1": every wave is done with W1 buffer 0
  if (nch > 1) burst(std::integral_constant<int, 0>(), smem + V2_W1 + 32768, smem + V2_W2);
  else burst(std::integral_constant<int, 16>(), smem + V2_W1, smem + V2_W2);
  write_p();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the P writes have landed before the barrier publishes them
  barrier();                            // Y (span 0 issues the DMAs of W1 chunk 2 and W2 chunk 1 itself)

#pragma unroll 1
  for (int c = 0; c < nch - 1; ++c) {
    const char* w1 = smem + V2_W1 + ((c + 1) & 1) * 32768;
    const char* w2 = smem + V2_W2 + (c & 1) * 32768;
    span(std::true_type(), c, w1, w2);
    // the packed tile is complete HERE (otherwise the compiler sinks the epilogue behind the barrier, next to its only use)
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(packed[e]));
    // the weight DMAs issued early in this span have landed (the two hidden-tile stores are older still); the backward's two
    // gate DMAs of the next chunk are the youngest and stay in flight
    if (MODE == MODE_BWD && FULL) { if constexpr (BITS) wait_vm<1>(); else wait_vm<2>(); } else wait_vm<0>();
    if constexpr ((DBG & 16) == 0) barrier();   // X: P(c), W2 buffer c & 1 and W1 buffer (c+1) & 1 are free; the next span's weights are visible
    if (c + 2 < nch) burst(std::integral_constant<int, 0>(), smem + V2_W1 + (c & 1) * 32768, smem + V2_W2 + ((c + 1) & 1) * 32768);
    else burst(std::integral_constant<int, 16>(), smem + V2_W1, smem + V2_W2 + ((c + 1) & 1) * 32768);
    write_p();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr ((DBG & 16) == 0) barrier();   // Y: P(c+1) visible
  }
  // the residual rows of the final epilogue: requested HERE, so that their HBM latency runs under the last span (no LDS-DMA is
  // outstanding any more; in the loop an ordinary load would make the compiler's vmcnt wait drain the weight prefetch)
  uint4 rres[4][2];
#pragma unroll
  for (int ob = 0; ob < 4; ++ob) rres[ob][0] = rres[ob][1] = make_uint4(0u, 0u, 0u, 0u);
  if (OUTM == 0 && a.residual) {
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
      const uint4* rp = reinterpret_cast<const uint4*>(a.residual + (int64_t)row_c * D + 128 * hh + 32 * ob + 16 * h);
      rres[ob][0] = rp[0];
      rres[ob][1] = rp[1];
    }
  }
  span(std::false_type(), nch - 1, smem + V2_W1, smem + V2_W2 + ((nch - 1) & 1) * 32768);

  if constexpr (MODE == MODE_BWD) wait_vm<0>();   // (the clamped gate copy of "chunk nch")
  if constexpr (LN) {
    // ---------------------------------------------------------------- LN variants: the raw output tile [128][256] f32 goes through
    // LDS (every weight / P buffer is free now) to the row phase shared with the whole-row products: forward bias + dropout + the
    // float32 residual stream + the NEXT LayerNorm; backward the LayerNorm backward of the wrapper around this feed-forward pair
    wait_vm<0>();
    barrier();                          // every wave is done with the last span's fragments
    float* tile = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
      float* tp = tile + (32 * rg + i_l) * rowphase::TILE_LD + 128 * hh + 32 * ob + 16 * h;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(tp + 4 * q) = make_float4(accY[ob][4 * q], accY[ob][4 * q + 1], accY[ob][4 * q + 2], accY[ob][4 * q + 3]);
    }
    __syncthreads();
    const float nopre[1][8] = {{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
    if constexpr (MODE == MODE_FWD) {
      rowphase::ln_fwd<16, false>(tile, a.rp, a.seed2 + seed_off, m0, M, wave, lane, nopre);
    } else {
      uint64_t sd = a.rp_seed;
      if (a.rp.dz) sd = seed_with_offset(a.rp_seed, a.seed_dev);   // wave-uniform
      rowphase::ln_bwd<8, 16, false>(tile, tile + 128 * rowphase::TILE_LD, a.rp, sd, m0, M, tid, wave, lane, nopre);
    }
    return;
  }
  if constexpr (OUTM == 2) {
    if (row_ok) {
      float* sp = a.slab + ((int64_t)blockIdx.y * M + row) * D + 128 * hh + 16 * h;
#pragma unroll
      for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(sp + 32 * ob + 4 * q) = make_float4(accY[ob][4 * q], accY[ob][4 * q + 1], accY[ob][4 * q + 2], accY[ob][4 * q + 3]);
    }
    return;
  }
  // ---------------------------------------------------------------- final epilogue: 16 consecutive output columns per lane and block
#pragma unroll
  for (int ob = 0; ob < 4; ++ob) {
    const int col0 = 128 * hh + 32 * ob + 16 * h;
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = accY[ob][e];
    if (MODE == MODE_FWD && a.bias_b) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b = *reinterpret_cast<const float4*>(a.bias_b + col0 + 4 * q);
        v[4 * q + 0] += b.x; v[4 * q + 1] += b.y; v[4 * q + 2] += b.z; v[4 * q + 3] += b.w;
      }
    }
    if constexpr ((DROP & 2) != 0) {
      const uint64_t idx = (uint64_t)row * (uint64_t)D + (uint64_t)col0;
      float k0[8], k1[8];
      dropout_keep8(a.seed2 + seed_off, a.stream2, idx, a.drop2_thresh, a.drop2_inv_keep, k0);
      dropout_keep8(a.seed2 + seed_off, a.stream2, idx + 8, a.drop2_thresh, a.drop2_inv_keep, k1);
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[e] *= k0[e]; v[8 + e] *= k1[e]; }
    }
    {
      union { uint4 u[2]; bf16_t s[16]; } r;   // zeros without a residual
      r.u[0] = rres[ob][0];
      r.u[1] = rres[ob][1];
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] += bf16_to_f32(r.s[e]);
    }
    if (row_ok) {
      typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
      u32x4_t* o = reinterpret_cast<u32x4_t*>(a.out + (int64_t)row * D + col0);
      __builtin_nontemporal_store(u32x4_t{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])}, o);
      __builtin_nontemporal_store(u32x4_t{pack_bf16x2(v[8], v[9]), pack_bf16x2(v[10], v[11]), pack_bf16x2(v[12], v[13]), pack_bf16x2(v[14], v[15])}, o + 1);
    }
  }
}

// Raises the kernel's dynamic-LDS limit to the whole 160 KB of a CU, once per kernel and thread: the launches of one
// instantiation ask for different sizes (the forward's depends on the filter size), so remembering only "already set" would
// leave a later, larger request above the limit of an earlier, smaller one.
template <typename KernelT>
void allow_lds(KernelT kernel, int /*bytes*/) {
  static thread_local const void* done[16];
  static thread_local int ndone = 0;
  for (int i = 0; i < ndone; ++i)
    if (done[i] == (const void*)kernel) return;
  (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (ndone < 16) done[ndone++] = (const void*)kernel;
}

// v2 (eight waves, two per SIMD): forward, full chip (>= 160 workgroups of 128 rows), F a multiple of 64 whose bias fits the
// 16 KB behind the P tile.
bool use_v2_fwd(const FfnArgs& a) {
  return a.F % CH2 == 0 && a.F >= 2 * CH2 && a.F <= 4096 && a.M >= 128 * 160;
}

// Ablation build (make -C neurst_amd/csrc ablation -> lib/libneurst_hip_ablation.so, -DNST_FFN_ABLATION; never the product
// library): NST_FFN_DBG=<bits> launches the forward (dropout, full tiles, gate bits) / backward (full tiles, gate bits) kernel with
// stages removed -- 1 hidden-tile and gate-bit stores, 2 weight DMA of the loop, 4 weight fragment reads, 8 MFMAs, 16 the loop's
// barriers, 32 the mid epilogue's arithmetic, 64 the P tile's writes and reads.  Results are wrong by construction; only the
// durations mean something (scripts/ffn_ablation.py, profiles/r06_ffn_ablation.json).
#ifdef NST_FFN_ABLATION
#define NST_FFN_DBG_LIST(X) X(1) X(2) X(4) X(8) X(16) X(32) X(64) X(3) X(71) X(76) X(103) X(119) X(127)
int ffn_dbg_bits() {
  const char* e = getenv("NST_FFN_DBG");
  return e ? atoi(e) : 0;
}
template <int MODE, int DROP>
bool launch_ablation(const FfnArgs& a, int lds, hipStream_t st) {
  const int dbg = ffn_dbg_bits();
  if (dbg == 0 || a.M % V2_ROWS != 0 || !a.gate_bits) return false;
#define NST_FFN_DBG_CASE(B)                                      \
  if (dbg == B) {                                                \
    auto k = ffn_pair8_kernel<MODE, DROP, true, B, true>;        \
    allow_lds(k, lds);                                           \
    k<<<a.M / V2_ROWS, 512, lds, st>>>(a);                       \
    return true;                                                 \
  }
  NST_FFN_DBG_LIST(NST_FFN_DBG_CASE)
#undef NST_FFN_DBG_CASE
  return false;
}
#endif

template <int DROP, bool FULL>
int launch_v2_fwd(const FfnArgs& a, hipStream_t st) {
  constexpr int DBG = 0;
  const int lds = V2_BIAS + a.F * 4;
#ifdef NST_FFN_ABLATION
  if constexpr (DROP == 3 && FULL)
    if (launch_ablation<MODE_FWD, 3>(a, lds, st)) return NST_OK;
#endif
  if (a.gate_bits) {
    auto k = ffn_pair8_kernel<MODE_FWD, DROP, FULL, 0, true>;
    allow_lds(k, lds);
    k<<<(a.M + V2_ROWS - 1) / V2_ROWS, 512, lds, st>>>(a);
    return NST_OK;
  }
  auto k = ffn_pair8_kernel<MODE_FWD, DROP, FULL, DBG>;
  allow_lds(k, lds);
  k<<<(a.M + V2_ROWS - 1) / V2_ROWS, 512, lds, st>>>(a);
  return NST_OK;
}

// LN variants: training shapes of the eight-wave kernel with gate bits
template <int DROP, bool FULL>
void launch_v2_fwd_ln(const FfnArgs& a, hipStream_t st) {
  const int lds = V2_BIAS + a.F * 4;
  auto k = ffn_pair8_kernel<MODE_FWD, DROP, FULL, 0, true, 1>;
  allow_lds(k, lds);
  k<<<(a.M + V2_ROWS - 1) / V2_ROWS, 512, lds, st>>>(a);
}
template <bool FULL>
void launch_v2_bwd_ln(const FfnArgs& a, hipStream_t st) {
  const int lds = V2_BIAS + 16384;
  auto k = ffn_pair8_kernel<MODE_BWD, 0, FULL, 0, true, 1>;
  allow_lds(k, lds);
  k<<<(a.M + V2_ROWS - 1) / V2_ROWS, 512, lds, st>>>(a);
}

// ---- hidden dimension split over workgroups (fewer than 160 row tiles of 128 rows): S slices fill the chip with S x tiles
// workgroups, ffn_slab_rows_kernel adds their slabs and runs the row phase on 32-row tiles
int ffn_split(int64_t M, int F) {
  if (M >= 128 * 160 || M < 1024 || F % CH2 != 0 || F < 4 * CH2 || F > 4096) return 0;
  const int tiles = (int)((M + V2_ROWS - 1) / V2_ROWS);
  int S = 240 / tiles;
  if (S > 4) S = 4;
  while (S > 1 && (F / CH2) / S < 2) --S;
  return S >= 2 ? S : 0;
}

struct SlabArgs {
  const float* slab;
  int S, M;
  rowphase::RowEpi rp;
  uint64_t seed;
  const uint64_t* seed_dev;
};

template <int EPI>   // 0: forward row phase, 1: backward
__global__ void __launch_bounds__(256, 2) ffn_slab_rows_kernel(SlabArgs a) {
  __shared__ __attribute__((aligned(16))) float tile[32 * rowphase::TILE_LD + (EPI == 1 ? 4 * 2 * rowphase::RN : 0)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * 32, M = a.M;
  // the f32 rows of the row phase first (its lane layout), then every slab piece of this thread: all loads of the workgroup are
  // in flight before the first add
  float xpre[4][8];
  {
    const int sub = lane >> 5, col = (lane & 31) * 8;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      int rowg = m0 + wave * 8 + p * 2 + sub;
      rowg = rowg < M ? rowg : M - 1;
      const float4 u0 = *reinterpret_cast<const float4*>(a.rp.x + (int64_t)rowg * D + col);
      const float4 u1 = *reinterpret_cast<const float4*>(a.rp.x + (int64_t)rowg * D + col + 4);
      xpre[p][0] = u0.x; xpre[p][1] = u0.y; xpre[p][2] = u0.z; xpre[p][3] = u0.w;
      xpre[p][4] = u1.x; xpre[p][5] = u1.y; xpre[p][6] = u1.z; xpre[p][7] = u1.w;
    }
  }
  float4 v[4][8];
#pragma unroll
  for (int sl = 0; sl < 4; ++sl) {
    if (sl < a.S) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int idx = i * 256 + tid, r = idx >> 6, c4 = (idx & 63) * 4;
        int rowg = m0 + r;
        rowg = rowg < M ? rowg : M - 1;
        v[sl][i] = *reinterpret_cast<const float4*>(a.slab + ((int64_t)sl * M + rowg) * D + c4);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[sl][i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = i * 256 + tid, r = idx >> 6, c4 = (idx & 63) * 4;
    float4 t;   // slices in order 0, 1, 2, 3: a fixed summation order
    t.x = ((v[0][i].x + v[1][i].x) + v[2][i].x) + v[3][i].x;
    t.y = ((v[0][i].y + v[1][i].y) + v[2][i].y) + v[3][i].y;
    t.z = ((v[0][i].z + v[1][i].z) + v[2][i].z) + v[3][i].z;
    t.w = ((v[0][i].w + v[1][i].w) + v[2][i].w) + v[3][i].w;
    *reinterpret_cast<float4*>(tile + r * rowphase::TILE_LD + c4) = t;
  }
  __syncthreads();
  uint64_t sd = a.seed;
  if (a.rp.drop_thresh) sd = seed_with_offset(a.seed, a.seed_dev);
  if constexpr (EPI == 0) rowphase::ln_fwd<8, true>(tile, a.rp, sd, m0, M, wave, lane, xpre);
  else rowphase::ln_bwd<4, 8, true>(tile, tile + 32 * rowphase::TILE_LD, a.rp, sd, m0, M, tid, wave, lane, xpre);
}

template <int DROP, bool FULL>
void launch_v2_fwd_split(const FfnArgs& a, int S, hipStream_t st) {
  const int lds = V2_BIAS + a.F * 4;
  auto k = ffn_pair8_kernel<MODE_FWD, DROP, FULL, 0, true, 2>;
  allow_lds(k, lds);
  k<<<dim3((a.M + V2_ROWS - 1) / V2_ROWS, S), 512, lds, st>>>(a);
}
template <bool FULL>
void launch_v2_bwd_split(const FfnArgs& a, int S, hipStream_t st) {
  const int lds = V2_BIAS + 16384;
  auto k = ffn_pair8_kernel<MODE_BWD, 0, FULL, 0, true, 2>;
  allow_lds(k, lds);
  k<<<dim3((a.M + V2_ROWS - 1) / V2_ROWS, S), 512, lds, st>>>(a);
}

bool use_v2_bwd(const FfnArgs& a) {
  return a.F % CH2 == 0 && a.F >= 2 * CH2 && a.F <= 4096 && a.M >= 128 * 160;
}

int launch_pair_v2_bwd(const FfnArgs& a_in, hipStream_t st) {
  FfnArgs a = a_in;
  a.rot_mode = 0;
  const int lds = V2_BIAS + 16384;   // the eight 2 KB gate regions
#ifdef NST_FFN_ABLATION
  if (launch_ablation<MODE_BWD, 0>(a, lds, st)) return NST_OK;
#endif
  if (a.gate_bits) {
    if (a.M % V2_ROWS == 0) {
      auto k = ffn_pair8_kernel<MODE_BWD, 0, true, 0, true>;
      allow_lds(k, lds);
      k<<<a.M / V2_ROWS, 512, lds, st>>>(a);
    } else {
      auto k = ffn_pair8_kernel<MODE_BWD, 0, false, 0, true>;
      allow_lds(k, lds);
      k<<<(a.M + V2_ROWS - 1) / V2_ROWS, 512, lds, st>>>(a);
    }
    return NST_OK;
  }
  if (a.M % V2_ROWS == 0) {
    auto k = ffn_pair8_kernel<MODE_BWD, 0, true, 0>;
    allow_lds(k, lds);
    k<<<a.M / V2_ROWS, 512, lds, st>>>(a);
  } else {
    auto k = ffn_pair8_kernel<MODE_BWD, 0, false, 0>;
    allow_lds(k, lds);
    k<<<(a.M + V2_ROWS - 1) / V2_ROWS, 512, lds, st>>>(a);
  }
  return NST_OK;
}

int launch_pair_v2_fwd(const FfnArgs& a_in, hipStream_t st) {
  FfnArgs a = a_in;
  a.rot_mode = 0;   // rotated chunk order per workgroup (per-XCD rotation measured no faster)
  const bool full = a.M % V2_ROWS == 0;
  const int drop = (a.drop1_thresh ? 1 : 0) | (a.drop2_thresh ? 2 : 0);
  if (full && drop == 3) return launch_v2_fwd<3, true>(a, st);
  if (full && drop == 0) return launch_v2_fwd<0, true>(a, st);
  return launch_v2_fwd<3, false>(a, st);
}

// [R, C] bf16 -> [C, R] bf16, 64x64 tiles through LDS, a table of matrices per launch
struct TransposeJob { const bf16_t* src; bf16_t* dst; int rows, cols, tiles_c, tile0; };

__global__ void __launch_bounds__(256) transpose_bf16_kernel(const TransposeJob* __restrict__ jobs, int njobs) {
  __shared__ bf16_t tile[64][66];
  int j = 0;
  while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].tile0) ++j;
  const TransposeJob q = jobs[j];
  const int t = blockIdx.x - q.tile0, tr = t / q.tiles_c, tc = t - tr * q.tiles_c;
  const int r0 = tr * 64, c0 = tc * 64;
  // 16-byte global accesses when both shapes allow it (every weight matrix of the models does): 8 elements of a source row
  // in, 8 elements of a destination row (= 8 source rows of one column, gathered from the tile) out; element-wise otherwise
  const bool vec = ((q.rows | q.cols) & 7) == 0 && ((((uintptr_t)q.src) | ((uintptr_t)q.dst)) & 15) == 0;
  if (vec) {
    for (int e = threadIdx.x; e < 64 * 8; e += 256) {
      const int r = e >> 3, c = (e & 7) * 8;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (r0 + r < q.rows && c0 + c < q.cols) v = *reinterpret_cast<const uint4*>(q.src + (int64_t)(r0 + r) * q.cols + c0 + c);
      uint32_t* t32 = reinterpret_cast<uint32_t*>(&tile[r][c]);   // (row pitch 132 bytes: 4-byte aligned)
      t32[0] = v.x; t32[1] = v.y; t32[2] = v.z; t32[3] = v.w;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 8; e += 256) {
      const int c = e >> 3, r = (e & 7) * 8;
      if (r0 + r < q.rows && c0 + c < q.cols) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = (uint32_t)tile[r + 2 * i][c] | ((uint32_t)tile[r + 2 * i + 1][c] << 16);
        *reinterpret_cast<uint4*>(q.dst + (int64_t)(c0 + c) * q.rows + r0 + r) = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
    return;
  }
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    tile[r][c] = (r0 + r < q.rows && c0 + c < q.cols) ? q.src[(int64_t)(r0 + r) * q.cols + c0 + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int c = e >> 6, r = e & 63;
    if (r0 + r < q.rows && c0 + c < q.cols) q.dst[(int64_t)(c0 + c) * q.rows + r0 + r] = tile[r][c];
  }
}

// strided block copies for packed weight copies: job j copies rows x row_bytes from src (pitch src_pitch) to dst (pitch dst_pitch),
// 16 bytes per thread when everything is 16-byte aligned, byte-wise otherwise; block0 = number of 256-thread blocks of earlier jobs
struct Pack2dJob { const char* src; char* dst; int rows, row_bytes; int64_t src_pitch, dst_pitch; int block0, nblocks; };

__global__ void __launch_bounds__(256) pack2d_kernel(const Pack2dJob* __restrict__ jobs, int njobs) {
  int j = 0;
  while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].block0) ++j;
  const Pack2dJob q = jobs[j];
  const int b = blockIdx.x - q.block0;
  const bool vec = (((uintptr_t)q.src | (uintptr_t)q.dst | (uintptr_t)q.src_pitch | (uintptr_t)q.dst_pitch | (uintptr_t)q.row_bytes) & 15) == 0;
  if (vec) {
    const int cpr = q.row_bytes >> 4;
    const int64_t total = (int64_t)q.rows * cpr;
    for (int64_t e = (int64_t)b * 256 + threadIdx.x; e < total; e += (int64_t)q.nblocks * 256) {
      const int r = (int)(e / cpr), c = (int)(e - (int64_t)r * cpr);
      *reinterpret_cast<uint4*>(q.dst + r * q.dst_pitch + c * 16) = *reinterpret_cast<const uint4*>(q.src + r * q.src_pitch + c * 16);
    }
  } else {
    const int64_t total = (int64_t)q.rows * q.row_bytes;
    for (int64_t e = (int64_t)b * 256 + threadIdx.x; e < total; e += (int64_t)q.nblocks * 256) {
      const int r = (int)(e / q.row_bytes), c = (int)(e - (int64_t)r * q.row_bytes);
      q.dst[r * q.dst_pitch + c] = q.src[r * q.src_pitch + c];
    }
  }
}

}  // namespace

extern "C" int nst_pack2d(const NstPack2dJob* jobs_dev, int njobs, int total_blocks, void* stream) {
  NST_CHECK_ARG(njobs >= 0 && total_blocks >= 0 && (njobs == 0 || jobs_dev), "pack2d: bad arguments");
  if (njobs == 0 || total_blocks == 0) return NST_OK;
  static_assert(sizeof(Pack2dJob) == sizeof(NstPack2dJob), "job table layout");
  pack2d_kernel<<<total_blocks, 256, 0, (hipStream_t)stream>>>((const Pack2dJob*)jobs_dev, njobs);
  NST_CHECK_LAUNCH("pack2d");
  return NST_OK;
}

extern "C" int nst_ffn_supported(int d_model, int filter_size, int dtype) {
  return dtype == NST_BF16 && d_model == D && filter_size >= CH && filter_size % CH == 0 && filter_size <= 4096 ? 1 : 0;
}

extern "C" int nst_ffn_fwd(const NstFfnDesc* d, const void* x, const void* w1t, const float* b1, const void* w2t, const float* b2,
                           const void* residual, void* hidden, void* y, void* stream) {
  NST_CHECK_ARG(d && x && w1t && w2t && hidden && y, "ffn_fwd: null pointer");
  NST_CHECK_ARG(nst_ffn_supported(d->d_model, d->filter_size, d->dtype), "ffn_fwd: unsupported shape d=%d ffn=%d dtype=%d",
                d->d_model, d->filter_size, d->dtype);
  NST_CHECK_ARG(d->rows >= 0 && d->rows < (1 << 30), "ffn_fwd: rows=%lld", (long long)d->rows);
  NST_CHECK_ARG(nst_aligned16(x) && nst_aligned16(w1t) && nst_aligned16(w2t) && nst_aligned16(hidden) && nst_aligned16(y) &&
                    (!residual || nst_aligned16(residual)) && (!b2 || nst_aligned16(b2)),
                "ffn_fwd: operands must be 16-byte aligned");
  NST_CHECK_ARG(d->hidden_dropout_p >= 0.f && d->hidden_dropout_p < 1.f && d->output_dropout_p >= 0.f && d->output_dropout_p < 1.f,
                "ffn_fwd: dropout rate");
  if (d->rows == 0) return NST_OK;
  FfnArgs a;
  memset(&a, 0, sizeof(a));
  a.xin = (const bf16_t*)x; a.wa = (const bf16_t*)w1t; a.wb = (const bf16_t*)w2t;
  a.bias_a = b1; a.bias_b = b2; a.residual = (const bf16_t*)residual;
  a.mid_out = (bf16_t*)hidden; a.out = (bf16_t*)y;
  a.seed_dev = d->seed_offset ? d->seed_offset : nst_seed_offset_devptr();   // explicit scalar, else the library's
  if (!a.seed_dev) return NST_ERR_LAUNCH;
  a.M = (int)d->rows; a.F = d->filter_size;
  nst_dropout_params16(d->hidden_dropout_p, &a.drop1_thresh, &a.drop1_inv_keep);
  nst_dropout_params16(d->output_dropout_p, &a.drop2_thresh, &a.drop2_inv_keep);
  a.seed1 = d->hidden_seed; a.stream1 = d->hidden_stream_id; a.seed2 = d->output_seed; a.stream2 = d->output_stream_id;
  if (d->gate_bits) {
    const int64_t need = nst_ffn_gate_bits_bytes(d);
    NST_CHECK_ARG(need > 0 && d->gate_bits_bytes >= need && ((uintptr_t)d->gate_bits & 3) == 0,
                  "ffn_fwd: gate_bits given (%lld bytes) where nst_ffn_gate_bits_bytes says %lld", (long long)d->gate_bits_bytes,
                  (long long)need);
    a.gate_bits = (uint16_t*)d->gate_bits;
  }
  const int rc = launch_pair_v2_fwd(a, (hipStream_t)stream);   // (any row count: the four-wave kernel of round 2 is gone)
  if (rc != NST_OK) return rc;
  NST_CHECK_LAUNCH("ffn_fwd");
  return NST_OK;
}

extern "C" int64_t nst_ffn_gate_bits_bytes(const NstFfnDesc* d) {
  if (!d || !nst_ffn_supported(d->d_model, d->filter_size, d->dtype) || d->rows <= 0 || d->rows >= (1 << 30)) return 0;
  FfnArgs a;
  memset(&a, 0, sizeof(a));
  a.M = (int)d->rows; a.F = d->filter_size;
  return (use_v2_fwd(a) && use_v2_bwd(a)) ? (int64_t)a.M * (a.F / 32) * 4 : 0;   // (what nst_ffn_fwd / nst_ffn_bwd can use)
}

extern "C" int nst_ffn_bwd(const NstFfnDesc* d, const void* dy, const void* hidden, const void* w2, const void* w1,
                           const void* residual, void* dhidden, void* dx, void* stream) {
  NST_CHECK_ARG(d && dy && hidden && w2 && w1 && dhidden && dx, "ffn_bwd: null pointer");
  NST_CHECK_ARG(nst_ffn_supported(d->d_model, d->filter_size, d->dtype), "ffn_bwd: unsupported shape d=%d ffn=%d dtype=%d",
                d->d_model, d->filter_size, d->dtype);
  NST_CHECK_ARG(d->rows >= 0 && d->rows < (1 << 30), "ffn_bwd: rows=%lld", (long long)d->rows);
  // the eight-wave backward addresses its gate (rows x filter x 2 bytes) with 32-bit offsets
  NST_CHECK_ARG((int64_t)d->rows * d->filter_size * 2 < (1ll << 32), "ffn_bwd: rows * filter_size = %lld x %d exceeds the 4 GiB the "
                "gate offsets cover", (long long)d->rows, d->filter_size);
  NST_CHECK_ARG(nst_aligned16(dy) && nst_aligned16(hidden) && nst_aligned16(w1) && nst_aligned16(w2) && nst_aligned16(dhidden) &&
                    nst_aligned16(dx) && (!residual || nst_aligned16(residual)),
                "ffn_bwd: operands must be 16-byte aligned");
  if (d->rows == 0) return NST_OK;
  FfnArgs a;
  memset(&a, 0, sizeof(a));
  a.xin = (const bf16_t*)dy; a.wa = (const bf16_t*)w2; a.wb = (const bf16_t*)w1;
  a.residual = (const bf16_t*)residual; a.gate = (const bf16_t*)hidden;
  a.mid_out = (bf16_t*)dhidden; a.out = (bf16_t*)dx;
  a.M = (int)d->rows; a.F = d->filter_size;
  uint32_t th; float inv;
  nst_dropout_params16(d->hidden_dropout_p, &th, &inv);
  a.gate_scale = th ? inv : 1.0f;
  if (d->gate_bits && use_v2_bwd(a)) {   // (otherwise the saved activation is the gate)
    NST_CHECK_ARG(d->gate_bits_bytes >= (int64_t)a.M * (a.F / 32) * 4 && ((uintptr_t)d->gate_bits & 3) == 0,
                  "ffn_bwd: gate_bits holds %lld bytes", (long long)d->gate_bits_bytes);
    a.gate_bits = (uint16_t*)d->gate_bits;
  }
  const int rc = launch_pair_v2_bwd(a, (hipStream_t)stream);
  if (rc != NST_OK) return rc;
  NST_CHECK_LAUNCH("ffn_bwd");
  return NST_OK;
}

extern "C" int nst_ffn_ln_supported(const NstFfnDesc* d) {
  if (!d || !nst_ffn_supported(d->d_model, d->filter_size, d->dtype) || d->rows <= 0 || d->rows >= (1 << 30)) return 0;
  FfnArgs a;
  memset(&a, 0, sizeof(a));
  a.M = (int)d->rows; a.F = d->filter_size;
  if ((int64_t)d->rows * d->filter_size * 2 >= (1ll << 32)) return 0;
  if (use_v2_fwd(a) && use_v2_bwd(a)) return 1;
  return ffn_split(a.M, a.F) > 0 ? 2 : 0;
}

extern "C" int64_t nst_ffn_ln_slab_bytes(const NstFfnDesc* d) {
  if (nst_ffn_ln_supported(d) != 2) return 0;
  return (int64_t)ffn_split(d->rows, d->filter_size) * d->rows * D * 4;
}

extern "C" int nst_ffn_add_layernorm_fwd(const NstFfnDesc* d, const void* x, const void* w1t, const float* b1, const void* w2t,
                                         const float* b2, const float* x_res, float* x_out, const float* gamma, const float* beta,
                                         float eps, void* hidden, void* y, float* mean, float* rstd, void* slabs, int64_t slabs_bytes,
                                         void* stream) {
  NST_CHECK_ARG(d && x && w1t && w2t && hidden && y && x_res && gamma && beta && mean && rstd, "ffn_add_layernorm_fwd: null pointer");
  const int ln_mode = nst_ffn_ln_supported(d);
  NST_CHECK_ARG(ln_mode && d->gate_bits, "ffn_add_layernorm_fwd: needs the eight-wave kernel's shapes (rows >= 1024) and gate bits");
  NST_CHECK_ARG(ln_mode == 1 || (slabs && (((uintptr_t)slabs) & 15) == 0 && slabs_bytes >= nst_ffn_ln_slab_bytes(d)),
                "ffn_add_layernorm_fwd: %lld bytes of slabs needed (nst_ffn_ln_slab_bytes)", (long long)nst_ffn_ln_slab_bytes(d));
  NST_CHECK_ARG(nst_aligned16(x) && nst_aligned16(w1t) && nst_aligned16(w2t) && nst_aligned16(hidden) && nst_aligned16(y) &&
                    nst_aligned16(x_res) && (!x_out || nst_aligned16(x_out)) && nst_aligned16(gamma) && nst_aligned16(beta) &&
                    (!b2 || nst_aligned16(b2)),
                "ffn_add_layernorm_fwd: operands must be 16-byte aligned");
  NST_CHECK_ARG(d->hidden_dropout_p >= 0.f && d->hidden_dropout_p < 1.f && d->output_dropout_p >= 0.f && d->output_dropout_p < 1.f,
                "ffn_add_layernorm_fwd: dropout rate");
  FfnArgs a;
  memset(&a, 0, sizeof(a));
  a.xin = (const bf16_t*)x; a.wa = (const bf16_t*)w1t; a.wb = (const bf16_t*)w2t;
  a.bias_a = b1; a.bias_b = nullptr;
  a.mid_out = (bf16_t*)hidden; a.out = (bf16_t*)y;
  a.seed_dev = d->seed_offset ? d->seed_offset : nst_seed_offset_devptr();
  if (!a.seed_dev) return NST_ERR_LAUNCH;
  a.M = (int)d->rows; a.F = d->filter_size;
  nst_dropout_params16(d->hidden_dropout_p, &a.drop1_thresh, &a.drop1_inv_keep);
  nst_dropout_params16(d->output_dropout_p, &a.drop2_thresh, &a.drop2_inv_keep);
  a.seed1 = d->hidden_seed; a.stream1 = d->hidden_stream_id; a.seed2 = d->output_seed; a.stream2 = d->output_stream_id;
  const int64_t need = (int64_t)a.M * (a.F / 32) * 4;     // (both forms of this entry write them, also where nst_ffn_fwd cannot)
  NST_CHECK_ARG(need > 0 && d->gate_bits_bytes >= need && ((uintptr_t)d->gate_bits & 3) == 0, "ffn_add_layernorm_fwd: gate_bits holds %lld bytes, %lld needed",
                (long long)d->gate_bits_bytes, (long long)need);
  a.gate_bits = (uint16_t*)d->gate_bits;
  a.rp.bias = b2; a.rp.drop_thresh = a.drop2_thresh; a.rp.drop_inv_keep = a.drop2_inv_keep; a.rp.stream_id = a.stream2;
  a.rp.x = x_res; a.rp.x_out = x_out; a.rp.gamma = gamma; a.rp.beta = beta; a.rp.eps = eps;
  a.rp.y = (bf16_t*)y; a.rp.mean = mean; a.rp.rstd = rstd;
  a.rot_mode = 0;
  const bool full = a.M % V2_ROWS == 0;
  const int drop = (a.drop1_thresh ? 1 : 0) | (a.drop2_thresh ? 2 : 0);
  if (ln_mode == 2) {
    // fewer row tiles than CUs: the hidden dimension is split over S workgroups per tile (f32 slabs), a second launch adds the
    // slabs and runs the row phase
    const int S = ffn_split(a.M, a.F);
    a.slab = (float*)slabs;
    if (full && drop == 3) launch_v2_fwd_split<3, true>(a, S, (hipStream_t)stream);
    else if (full && drop == 0) launch_v2_fwd_split<0, true>(a, S, (hipStream_t)stream);
    else launch_v2_fwd_split<3, false>(a, S, (hipStream_t)stream);
    NST_CHECK_LAUNCH("ffn_add_layernorm_fwd(split)");
    SlabArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.slab = a.slab; sa.S = S; sa.M = a.M; sa.rp = a.rp; sa.seed = a.seed2; sa.seed_dev = a.seed_dev;
    ffn_slab_rows_kernel<0><<<(a.M + 31) / 32, 256, 0, (hipStream_t)stream>>>(sa);
    NST_CHECK_LAUNCH("ffn_add_layernorm_fwd(rows)");
    return NST_OK;
  }
  if (full && drop == 3) launch_v2_fwd_ln<3, true>(a, (hipStream_t)stream);
  else if (full && drop == 0) launch_v2_fwd_ln<0, true>(a, (hipStream_t)stream);
  else launch_v2_fwd_ln<3, false>(a, (hipStream_t)stream);
  NST_CHECK_LAUNCH("ffn_add_layernorm_fwd");
  return NST_OK;
}

extern "C" int nst_ffn_layernorm_bwd(const NstFfnDesc* d, const void* dy, const void* hidden, const void* w2, const void* w1,
                                     const float* x_ln, const float* gamma, const float* mean, const float* rstd, const void* dres,
                                     void* dhidden, void* dx, void* dz, float dz_p, uint64_t dz_seed, uint64_t dz_stream_id,
                                     float* dgamma, float* dbeta, int accumulate, void* workspace, int64_t workspace_bytes,
                                     NstLnFinalizeJob* job_out, void* slabs, int64_t slabs_bytes, void* stream) {
  if (job_out) memset(job_out, 0, sizeof(*job_out));
  NST_CHECK_ARG(d && dy && hidden && w2 && w1 && dhidden && dx && x_ln && gamma && mean && rstd && dgamma && dbeta && workspace,
                "ffn_layernorm_bwd: null pointer");
  const int ln_mode = nst_ffn_ln_supported(d);
  NST_CHECK_ARG(ln_mode && d->gate_bits, "ffn_layernorm_bwd: needs the eight-wave kernel's shapes and gate bits");
  NST_CHECK_ARG(ln_mode == 1 || (slabs && (((uintptr_t)slabs) & 15) == 0 && slabs_bytes >= nst_ffn_ln_slab_bytes(d)),
                "ffn_layernorm_bwd: %lld bytes of slabs needed (nst_ffn_ln_slab_bytes)", (long long)nst_ffn_ln_slab_bytes(d));
  NST_CHECK_ARG(nst_aligned16(dy) && nst_aligned16(hidden) && nst_aligned16(w1) && nst_aligned16(w2) && nst_aligned16(dhidden) &&
                    nst_aligned16(dx) && nst_aligned16(x_ln) && nst_aligned16(gamma) && (!dres || nst_aligned16(dres)) &&
                    (!dz || nst_aligned16(dz)) && (((uintptr_t)workspace) & 15) == 0,
                "ffn_layernorm_bwd: operands must be 16-byte aligned");
  NST_CHECK_ARG(dz_p >= 0.f && dz_p < 1.f, "ffn_layernorm_bwd: dropout_p=%f", dz_p);
  FfnArgs a;
  memset(&a, 0, sizeof(a));
  a.xin = (const bf16_t*)dy; a.wa = (const bf16_t*)w2; a.wb = (const bf16_t*)w1;
  a.gate = (const bf16_t*)hidden;
  a.mid_out = (bf16_t*)dhidden; a.out = (bf16_t*)dx;
  a.M = (int)d->rows; a.F = d->filter_size;
  uint32_t th; float inv;
  nst_dropout_params16(d->hidden_dropout_p, &th, &inv);
  a.gate_scale = th ? inv : 1.0f;
  NST_CHECK_ARG(d->gate_bits_bytes >= (int64_t)a.M * (a.F / 32) * 4 && ((uintptr_t)d->gate_bits & 3) == 0,
                "ffn_layernorm_bwd: gate_bits holds %lld bytes", (long long)d->gate_bits_bytes);
  a.gate_bits = (uint16_t*)d->gate_bits;
  const int nb = ln_mode == 2 ? (a.M + 31) / 32 : (a.M + V2_ROWS - 1) / V2_ROWS;
  if (workspace_bytes < (int64_t)nb * 2 * D * 4) {
    nst_set_error("ffn_layernorm_bwd: workspace of %lld bytes, %lld needed", (long long)workspace_bytes, (long long)nb * 2 * D * 4);
    return NST_ERR_WORKSPACE;
  }
  a.seed_dev = nst_seed_offset_devptr();
  if (!a.seed_dev) return NST_ERR_LAUNCH;
  a.rp.x = x_ln; a.rp.gamma = gamma; a.rp.mean = const_cast<float*>(mean); a.rp.rstd = const_cast<float*>(rstd);
  a.rp.dres = (const bf16_t*)dres; a.rp.y = (bf16_t*)dx; a.rp.dz = (bf16_t*)dz; a.rp.partial = (float*)workspace;
  if (dz) {
    nst_dropout_params16(dz_p, &a.rp.drop_thresh, &a.rp.drop_inv_keep);
    a.rp_seed = dz_seed; a.rp.stream_id = dz_stream_id;
  }
  a.rot_mode = 0;
  if (ln_mode == 2) {
    const int S = ffn_split(a.M, a.F);
    a.slab = (float*)slabs;
    if (a.M % V2_ROWS == 0) launch_v2_bwd_split<true>(a, S, (hipStream_t)stream);
    else launch_v2_bwd_split<false>(a, S, (hipStream_t)stream);
    NST_CHECK_LAUNCH("ffn_layernorm_bwd(split)");
    SlabArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.slab = a.slab; sa.S = S; sa.M = a.M; sa.rp = a.rp; sa.seed = a.rp_seed; sa.seed_dev = a.seed_dev;
    if (!a.rp.dz) sa.rp.drop_thresh = 0;
    ffn_slab_rows_kernel<1><<<nb, 256, 0, (hipStream_t)stream>>>(sa);
  } else if (a.M % V2_ROWS == 0) launch_v2_bwd_ln<true>(a, (hipStream_t)stream);
  else launch_v2_bwd_ln<false>(a, (hipStream_t)stream);
  NST_CHECK_LAUNCH("ffn_layernorm_bwd");
  NstLnFinalizeJob job;
  memset(&job, 0, sizeof(job));
  job.partial = (const float*)workspace; job.dgamma = dgamma; job.dbeta = dbeta;
  job.nblocks = nb; job.d = D; job.accumulate = accumulate;
  if (job_out) {
    *job_out = job;
    return NST_OK;
  }
  return nst_ln_finalize_multi(&job, 1, stream);
}

extern "C" int nst_transpose_bf16(const NstTransposeJob* jobs_dev, int njobs, int total_tiles, void* stream) {
  NST_CHECK_ARG(njobs >= 0 && total_tiles >= 0 && (njobs == 0 || jobs_dev), "transpose_bf16: bad arguments");
  if (njobs == 0 || total_tiles == 0) return NST_OK;
  static_assert(sizeof(TransposeJob) == sizeof(NstTransposeJob), "job table layout");
  transpose_bf16_kernel<<<total_tiles, 256, 0, (hipStream_t)stream>>>((const TransposeJob*)jobs_dev, njobs);
  NST_CHECK_LAUNCH("transpose_bf16");
  return NST_OK;
}
