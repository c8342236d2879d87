// Instruction-rate microbenchmark for the integer ops a 256-bit modular multiply is made of (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 ubench_isa.hip -o ubench_isa ; prints wave-instructions/ns/CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template <int OP>
__global__ __launch_bounds__(256) void ub(uint32_t* out, int iters) {
  uint32_t a = threadIdx.x * 2654435761u + 12345u, b = blockIdx.x * 40503u + 7u + threadIdx.x;
  uint64_t q[8];
  uint32_t w[8];
  double f[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { q[k] = a * (k + 1); w[k] = b + k; f[k] = 1.0 + k; }
  double fa = 1.000001, fb = 0.5;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if constexpr (OP == 0) {
#define S(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[k]) : "v"(a), "v"(b) : "vcc");
        REP8(S)
#undef S
      } else if constexpr (OP == 1) {
#define S(k) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q[k]) : "v"(q[(k + 1) & 7]));
        REP8(S)
#undef S
      } else if constexpr (OP == 2) {
#define S(k) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(w[k]) : "v"(a) : "vcc");
        REP8(S)
#undef S
      } else if constexpr (OP == 3) {
#define S(k) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(w[k]) : "v"(a) : "vcc");
        REP8(S)
#undef S
      } else if constexpr (OP == 4) {
#define S(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(w[k]) : "v"(a));
        REP8(S)
#undef S
      } else if constexpr (OP == 5) {
#define S(k) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(w[k]) : "v"(a));
        REP8(S)
#undef S
      } else if constexpr (OP == 6) {
#define S(k) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(w[k]) : "v"(a), "v"(b));
        REP8(S)
#undef S
      } else if constexpr (OP == 7) {
#define S(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f[k]) : "v"(fa), "v"(fb));
        REP8(S)
#undef S
      } else if constexpr (OP == 8) {
#define S(k) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(w[k]) : "v"(a), "v"(b));
        REP8(S)
#undef S
      } else if constexpr (OP == 9) {
#define S(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(w[k]) : "v"(a));
        REP8(S)
#undef S
      } else if constexpr (OP == 10) {
#define S(k) asm volatile("v_mov_b32 %0, %1" : "=v"(w[k]) : "v"(w[(k + 1) & 7]));
        REP8(S)
#undef S
      } else if constexpr (OP == 11) {
#define S(k) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(w[k]) : "v"(a));
        REP8(S)
#undef S
      } else if constexpr (OP == 12) {  // mad with SGPR multiplier operand
#define S(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[k]) : "v"(a), "s"(iters) : "vcc");
        REP8(S)
#undef S
      } else if constexpr (OP == 13) {
#define S(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(w[k]) : "v"(a) : );
        REP8(S)
#undef S
      } else if constexpr (OP == 14) {  // mad followed by dependent addc (Comba step)
#define S(k) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(q[k]), "+v"(w[k]) : "v"(a), "v"(b) : "vcc");
        REP8(S)
#undef S
      } else if constexpr (OP == 15) {
#define S(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(f[k]) : "v"(fa));
        REP8(S)
#undef S
      } else if constexpr (OP == 16) {
#define S(k) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(w[k]) : "v"(a));
        REP8(S)
#undef S
      } else if constexpr (OP == 17) {
#define S(k) asm volatile("v_alignbit_b32 %0, %0, %1, 29" : "+v"(w[k]) : "v"(a));
        REP8(S)
#undef S
      }
    }
  }
  uint64_t s = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += q[k] + w[k] + (uint64_t)f[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}

template <int OP>
int run(const char* name, int insn_per_group, uint32_t* d_out, int blocks, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  ub<OP><<<blocks, 256>>>(d_out, 16);  // warm
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0));
    ub<OP><<<blocks, 256>>>(d_out, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  double waves = (double)blocks * 4;
  double insn = waves * (double)iters * 4 * 8 * insn_per_group;  // wave-instructions
  double per_cu_per_ns = insn / (best * 1e6) / 256.0;
  // cycles per wave-instruction per SIMD assuming 2.4 GHz: 4 SIMDs per CU
  double cyc = 2.4 * 4.0 / per_cu_per_ns;
  printf("%-34s %8.3f ms  %7.3f wave-insn/ns/CU  => %6.2f cyc/wave-insn/SIMD @2.4GHz  (%.3e lane-ops/s chip)\n", name, best,
         per_cu_per_ns, cyc, insn * 64 / (best * 1e-3));
  return 0;
}

int main() {
  int dev = 0; CK(hipSetDevice(dev));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, dev));
  printf("device: %s  CUs=%d  clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  int blocks = p.multiProcessorCount * 8;  // 8 waves per SIMD
  uint32_t* d_out; CK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
  int iters = 4000;
  run<9>("v_add_u32", 1, d_out, blocks, iters);
  run<10>("v_mov_b32", 1, d_out, blocks, iters);
  run<2>("v_add_co_u32", 1, d_out, blocks, iters);
  run<3>("v_addc_co_u32", 1, d_out, blocks, iters);
  run<8>("v_add3_u32", 1, d_out, blocks, iters);
  run<13>("v_cndmask_b32", 1, d_out, blocks, iters);
  run<17>("v_alignbit_b32", 1, d_out, blocks, iters);
  run<1>("v_lshl_add_u64", 1, d_out, blocks, iters);
  run<0>("v_mad_u64_u32", 1, d_out, blocks, iters);
  run<12>("v_mad_u64_u32 (sgpr operand)", 1, d_out, blocks, iters);
  run<14>("v_mad_u64_u32 + v_addc_co_u32", 2, d_out, blocks, iters);
  run<4>("v_mul_lo_u32", 1, d_out, blocks, iters);
  run<5>("v_mul_hi_u32", 1, d_out, blocks, iters);
  run<6>("v_mad_u32_u24", 1, d_out, blocks, iters);
  run<11>("v_mul_hi_u32_u24", 1, d_out, blocks, iters);
  run<16>("v_pk_mul_lo_u16", 1, d_out, blocks, iters);
  run<7>("v_fma_f64", 1, d_out, blocks, iters);
  run<15>("v_mul_f64", 1, d_out, blocks, iters);
  // occupancy sensitivity for the mad: 1, 2, 4 waves per SIMD
  for (int wps : {1, 2, 4}) {
    char nm[64]; snprintf(nm, sizeof nm, "v_mad_u64_u32 @%d waves/SIMD", wps);
    run<0>(nm, 1, d_out, p.multiProcessorCount * wps, iters);
  }
  CK(hipFree(d_out));
  return 0;
}
