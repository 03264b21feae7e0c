// ds_read_b128 lane-group / bank-conflict probe for candidate LDS layouts: hipcc -O2 --offload-arch=gfx950 scripts/ldsprobe.hip -o /tmp/ldsprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
#include <functional>
#include <string>
__global__ void __launch_bounds__(512) probe(const int* __restrict__ addrs, unsigned long long* out, int npat) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < 131072 / 4; i += 512) ((int*)smem)[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  for (int p = 0; p < npat; ++p) {
    const int a = addrs[p * 64 + lane];
    typedef __attribute__((address_space(3))) char* lp;
    const uint32_t la = (uint32_t)(uintptr_t)((lp)smem) + a;
    uint4 acc = make_uint4(0, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 128; ++it) {
      uint4 v0, v1, v2, v3, v4, v5, v6, v7;
      asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8\n\tds_read_b128 %2, %8\n\tds_read_b128 %3, %8\n\t"
                   "ds_read_b128 %4, %8\n\tds_read_b128 %5, %8\n\tds_read_b128 %6, %8\n\tds_read_b128 %7, %8\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(la) : "memory");
      acc.x ^= v0.x ^ v1.y ^ v2.z ^ v3.w ^ v4.x ^ v5.x ^ v6.x ^ v7.x;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[p] = t1 - t0;
    if (acc.x == 0x12345678u) out[npat] = acc.x;
  }
}
int main() {
  std::vector<int> pats;
  std::vector<std::string> names;
  auto add = [&](std::string n, std::function<int(int)> f) { names.push_back(n); for (int l = 0; l < 64; ++l) pats.push_back(f(l)); };
  add("canonical lane*16", [](int l) { return l * 16; });
  add("broadcast", [](int l) { return 0; });
  add("all same bank (stride 256)", [](int l) { return l * 256; });
  add("rows128 + g*16 (no swz)", [](int l) { return (l & 15) * 128 + (l >> 4) * 16; });
  add("RC swz (gemm A tile)", [](int l) { int r = l & 15; return r * 128 + (((l >> 4)) ^ ((r >> 1) & 7)) * 16; });
  const int PW = 42, F2 = 20;
  // cc32 layouts: phys = pos ^ ((pos>>sb)&1), chunk = g ^ ((pos>>cb)&3), 64 B / position; positions 2 apart (+wrap)
  for (int f0 : {0, 8, 13}) {
    for (int sb = 0; sb < 6; ++sb)
      for (int cb = 0; cb < 6; ++cb) {
        char nm[64]; snprintf(nm, 64, "cc32 f0=%d sb=%d cb=%d", f0, sb, cb);
        add(nm, [=](int l) {
          int j = l & 15, g = l >> 4; int fo = f0 + j, dr = 0; if (fo >= F2) { fo -= F2; dr = 1; }
          int pos = 2 * dr * PW + 2 * fo + 43;
          int phys = sb ? (pos ^ ((pos >> sb) & 1)) : pos;
          return (phys << 6) + ((g ^ ((pos >> cb) & 3)) << 4);
        });
      }
  }
  // cc64 layout (128 B / position): phys = pos ^ ((pos>>4)&1); chunk = (g + 4*kk) ^ ((pos>>1)&7)
  for (int f0 : {0, 8, 13}) {
    char nm[64]; snprintf(nm, 64, "cc64 f0=%d", f0);
    add(nm, [=](int l) {
      int j = l & 15, g = l >> 4; int fo = f0 + j, dr = 0; if (fo >= F2) { fo -= F2; dr = 1; }
      int pos = 2 * dr * PW + 2 * fo + 43;
      return ((pos ^ ((pos >> 4) & 1)) << 7) + ((g ^ ((pos >> 1) & 7)) << 4);
    });
  }
  const int npat = (int)names.size();
  int* d; unsigned long long* o;
  hipMalloc(&d, pats.size() * 4); hipMalloc(&o, (npat + 1) * 8);
  hipMemcpy(d, pats.data(), pats.size() * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  probe<<<1, 512, 131072>>>(d, o, npat);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(npat + 1);
  hipMemcpy(h.data(), o, (npat + 1) * 8, hipMemcpyDeviceToHost);
  for (int p = 0; p < npat; ++p) printf("%-32s %.2f cyc/wave-read\n", names[p].c_str(), h[p] / (128.0 * 8 * 8));
  return 0;
}
