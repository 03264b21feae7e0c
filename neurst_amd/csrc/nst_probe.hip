// Hardware probes for the lane maps the MFMA kernels assume (tests/test_gpu_probe.py checks them against numpy):
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

}  // namespace

extern "C" int nst_probe_mfma(float* out_c16, float* out_c16_f32, uint16_t* out_tr, void* stream) {
  NST_CHECK_ARG(out_c16 && out_c16_f32 && out_tr, "probe_mfma: null pointer");
  probe_kernel<<<1, 64, 0, (hipStream_t)stream>>>(out_c16, out_c16_f32, out_tr);
  NST_CHECK_LAUNCH("probe_mfma");
  return NST_OK;
}
