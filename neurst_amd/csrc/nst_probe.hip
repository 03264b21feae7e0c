// Hardware probes for the lane maps the MFMA kernels assume (tests/test_gpu_kernels.py checks them against numpy):
//   out_c16     [16x16] f32 : D = A.B computed with ONE v_mfma_f32_16x16x32_bf16, fragments loaded and the result
//                             scattered with exactly the index formulas of nst_gemm_core.h
//   out_c16_f32 [16x16] f32 : same for v_mfma_f32_16x16x4_f32
//   out_tr      [64x8] u16  : what FragReader<bf16, OC, true> (ds_read_b64_tr_b16) returns from a [32][16] tile
//                             holding value r*16+i at (reduction row r, column i)
// Inputs are generated in-kernel: A[i][k] = ((i*37 + k*11) % 17 - 8) / 8, B[k][j] = ((k*13 + j*7) % 19 - 9) / 16
// (exactly representable in bf16).
#include "nst_gemm_core.h"

namespace {

__device__ __forceinline__ float a_val(int i, int k) { return (float)((i * 37 + k * 11) % 17 - 8) / 8.0f; }
__device__ __forceinline__ float b_val(int k, int j) { return (float)((k * 13 + j * 7) % 19 - 9) / 16.0f; }

__global__ void probe_kernel(float* out_c16, float* out_c16_f32, uint16_t* out_tr) {
  __shared__ __attribute__((aligned(16))) char lds[2 * nstgemm::Tile<bf16_t>::LDS_BYTES];
  const int lane = threadIdx.x;
  // ---- bf16 16x16x32: A as an RC tile [16 rows][32 k], B as an RC tile [16 cols j][32 k]
  char* As = lds;
  char* Bs = lds + nstgemm::Tile<bf16_t>::LDS_BYTES;
  for (int e = lane; e < 16 * 32; e += 64) {
    const int r = e / 32, k = e % 32;
    *reinterpret_cast<bf16_t*>(As + r * nstgemm::RS_RC + k * 2) = f32_to_bf16(a_val(r, k));
    *reinterpret_cast<bf16_t*>(Bs + r * nstgemm::RS_RC + k * 2) = f32_to_bf16(b_val(k, r));
  }
  __syncthreads();
  {
    bf16x8_t a = nstgemm::FragReader<bf16_t, nstgemm::MODE_RC, true>::read(As, 0, 0, lane);
    bf16x8_t b = nstgemm::FragReader<bf16_t, nstgemm::MODE_RC, true>::read(Bs, 0, 0, lane);
    floatx4_t c = {0.f, 0.f, 0.f, 0.f};
    c = nstgemm::Mma<bf16_t>::run(a, b, c);
    for (int r = 0; r < 4; ++r) out_c16[((lane >> 4) * 4 + r) * 16 + (lane & 15)] = c[r];
  }
  __syncthreads();
  // ---- f32 16x16x4
  for (int e = lane; e < 16 * 4; e += 64) {
    const int r = e / 4, k = e % 4;
    *reinterpret_cast<float*>(As + r * nstgemm::RS_RC + k * 4) = a_val(r, k);
    *reinterpret_cast<float*>(Bs + r * nstgemm::RS_RC + k * 4) = b_val(k, r);
  }
  __syncthreads();
  {
    float a = nstgemm::FragReader<float, nstgemm::MODE_RC, true>::read(As, 0, 0, lane);
    float b = nstgemm::FragReader<float, nstgemm::MODE_RC, true>::read(Bs, 0, 0, lane);
    floatx4_t c = {0.f, 0.f, 0.f, 0.f};
    c = nstgemm::Mma<float>::run(a, b, c);
    for (int r = 0; r < 4; ++r) out_c16_f32[((lane >> 4) * 4 + r) * 16 + (lane & 15)] = c[r];
  }
  __syncthreads();
  // ---- LDS transpose read: OC tile [32 reduction rows][16 cols], value r*16+i
  for (int e = lane; e < 32 * 16; e += 64) {
    const int r = e / 16, i = e % 16;
    *reinterpret_cast<uint16_t*>(As + r * nstgemm::Tile<bf16_t>::RS_OC + i * 2) = (uint16_t)(r * 16 + i);
  }
  __syncthreads();
  {
    bf16x8_t f = nstgemm::FragReader<bf16_t, nstgemm::MODE_OC, true>::read(As, 0, 0, lane);
    union { bf16x8_t f; uint16_t s[8]; } u;
    u.f = f;
    for (int j = 0; j < 8; ++j) out_tr[lane * 8 + j] = u.s[j];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fetch probe: what a CU can pull through its vector-memory path when it does NOTHING else.  Every workgroup (4 waves)
// streams `steps` pieces of 32 KB into a ring of LDS slots exactly like the stream GEMM does (8 LDS-DMA instructions of 1 KB
// per wave and step, counted vmcnt waits, one barrier per step) -- or with plain 16-byte loads into registers (MODE 1) --
// and multiplies nothing.  wg_stride = 0: all workgroups read the same `span` bytes (the weight stream of the feed-forward
// pair / conv2 kernels: L2 hits after the first pass); wg_stride = span: private regions (operands streamed from HBM once).
// The rate this reaches is the ceiling of every kernel whose tile re-fetches its operands per workgroup
// (profiles/r03_fetch_ceiling.json).
template <int MODE, int pat>
__global__ void __launch_bounds__(256) fetch_probe_kernel(const char* __restrict__ src, int64_t wg_stride, int64_t span, int steps,
                                                          int ring, int group_mod, float* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  typedef __attribute__((address_space(3))) char* lds_char_ptr;
  const uint32_t smem_addr = (uint32_t)(uintptr_t)((lds_char_ptr)smem_dyn);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* base = src + (int64_t)(blockIdx.x % group_mod) * wg_stride;
  // piece i of a step: bytes [i KB, i KB + 1 KB) of the 32 KB; wave w issues pieces w, w + 4, ...
  int64_t off = 0;
  float acc = 0.f;
  if constexpr (MODE == 1) {   // plain loads, one step (8 x 16 bytes per lane) in flight behind the one being consumed
    float4 cur[8], nxt[8];
    auto load = [&](float4 (&v)[8]) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const float4*>(base + off + (int64_t)((i * 4 + wave) * 1024 + lane * 16));
      off += 32768;
      if (off >= span) off -= span;
    };
    load(cur);
    for (int s = 0; s < steps; ++s) {
      if (s + 1 < steps) load(nxt);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += cur[i].x;
#pragma unroll
      for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
    }
    if (acc == 12345.678f) sink[0] = acc;
    return;
  }
  // lane -> byte of the 1 KB piece.  pat 0: lane * 16 (contiguous).  pat 1: the reduction-major tile of the stream GEMM -- 4 rows
  // of 256 B, the 16-byte chunks of a row XOR-ed by the row's swizzle (even masks: 32-byte pairs stay together).  pat 2: the
  // row-major tile -- 8 rows of 128 B, chunks XOR-ed by (row >> 1) & 7 (any mask).  pat 3 / 4: as 1 / 2 with rows 4 KB apart
  // (a piece then touches 4 / 8 separate segments, as a tile of a wide matrix does).
  auto issue = [&](int slot) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = i * 4 + wave;
      int64_t byte;
      if (pat == 0) {
        byte = q * 1024 + lane * 16;
      } else if (pat == 1 || pat == 3) {
        const int r = q * 4 + (lane >> 4), c16 = lane & 15;
        const int g = (r & 3) | (((r >> 3) & 1) << 2);
        byte = (pat == 1 ? (int64_t)r * 256 : (int64_t)(r & 7) * 4096 + (r >> 3) * 256) + ((c16 ^ (g << 1)) << 4);
      } else {
        const int r = q * 8 + (lane >> 3), c8 = lane & 7;
        byte = (pat == 2 ? (int64_t)r * 128 : (int64_t)(r & 7) * 4096 + (r >> 3) * 128) + ((c8 ^ ((r >> 1) & 7)) << 4);
      }
      nstgemm::glds16(base + off + byte, __builtin_amdgcn_readfirstlane(smem_addr + (uint32_t)slot * 32768u + (uint32_t)(q * 1024)));
    }
    off += 32768;
    if (off >= span) off -= span;
  };
  int issued = 0, slot = 0;
  for (; issued < ring - 1 && issued < steps; ++issued) { issue(slot); slot = slot + 1 == ring ? 0 : slot + 1; }
  for (int s = 0; s < steps; ++s) {
    const int younger = issued - s - 1;   // steps in flight behind the one waited for (8 instructions each)
    if (younger >= 3) nstgemm::wait_vmcnt<24>();
    else if (younger == 2) nstgemm::wait_vmcnt<16>();
    else if (younger == 1) nstgemm::wait_vmcnt<8>();
    else nstgemm::wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (issued < steps) { issue(slot); slot = slot + 1 == ring ? 0 : slot + 1; ++issued; }
    acc += *reinterpret_cast<const volatile float*>(smem_dyn + ((s % ring) * 32768) + tid * 4);
  }
  if (acc == 12345.678f) sink[0] = acc;   // (keeps the LDS reads alive)
}

}  // namespace

extern "C" int nst_probe_fetch(const void* src, int64_t wg_stride, int64_t span, int steps, int ring, int mode, int workgroups,
                               int group_mod, int pattern, float* sink, void* stream) {
  NST_CHECK_ARG(src && sink && steps > 0 && ring >= 2 && ring <= 4 && span >= 32768 && span % 32768 == 0 && workgroups > 0 &&
                    (mode == 0 || mode == 1) && nst_aligned16(src) && group_mod > 0 && pattern >= 0 && pattern <= 4,
                "probe_fetch: bad arguments");
  const int lds = ring * 32768;
#define NST_PROBE_LAUNCH(M_, P_)                                                                                             \
  do {                                                                                                                     \
    NST_CHECK_HIP(hipFuncSetAttribute((const void*)fetch_probe_kernel<M_, P_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
    fetch_probe_kernel<M_, P_><<<workgroups, 256, lds, (hipStream_t)stream>>>((const char*)src, wg_stride, span, steps, ring,   \
                                                                             group_mod, sink);                             \
  } while (0)
  if (mode == 1) NST_PROBE_LAUNCH(1, 0);
  else if (pattern == 0) NST_PROBE_LAUNCH(0, 0);
  else if (pattern == 1) NST_PROBE_LAUNCH(0, 1);
  else if (pattern == 2) NST_PROBE_LAUNCH(0, 2);
  else if (pattern == 3) NST_PROBE_LAUNCH(0, 3);
  else NST_PROBE_LAUNCH(0, 4);
#undef NST_PROBE_LAUNCH
  NST_CHECK_LAUNCH("probe_fetch");
  return NST_OK;
}

extern "C" int nst_probe_mfma(float* out_c16, float* out_c16_f32, uint16_t* out_tr, void* stream) {
  NST_CHECK_ARG(out_c16 && out_c16_f32 && out_tr, "probe_mfma: null pointer");
  probe_kernel<<<1, 64, 0, (hipStream_t)stream>>>(out_c16, out_c16_f32, out_tr);
  NST_CHECK_LAUNCH("probe_mfma");
  return NST_OK;
}
