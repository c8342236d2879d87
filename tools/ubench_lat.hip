// dependent-chain latency of v_mad_u64_u32 / v_add_u32 / v_lshl_add_u64 (1 wave per SIMD, one chain)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <int OP> __global__ void lat(uint32_t* out, int iters) {
  uint32_t a = threadIdx.x + 3, b = blockIdx.x + 5; uint64_t q = a; uint32_t w = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      if constexpr (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q) : "v"(a), "v"(b) : "vcc");
      if constexpr (OP == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(w) : "v"(a));
      if constexpr (OP == 2) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q) : "v"(q));
      if constexpr (OP == 3) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_lshrrev_b64 %0, 29, %0" : "+v"(q) : "v"(a), "v"(b) : "vcc");
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)q + w;
}
template <int OP> void run(const char* nm, int per, uint32_t* d, int waves_per_simd) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int blocks = 256 * waves_per_simd, iters = 2000;
  lat<OP><<<blocks, 256>>>(d, 10); hipDeviceSynchronize();
  hipEventRecord(e0); lat<OP><<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 32 * per);
  printf("%-28s waves/SIMD=%d: %.2f cycles per instruction per wave (@2.4GHz)\n", nm, waves_per_simd, cyc);
}
int main() {
  uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  for (int w : {1, 2, 4, 8}) {
    run<0>("dep v_mad_u64_u32", 1, d, w); run<1>("dep v_add_u32", 1, d, w); run<2>("dep v_lshl_add_u64", 1, d, w);
    run<3>("dep mad+lshr64 pair", 2, d, w);
  }
  return 0;
}
