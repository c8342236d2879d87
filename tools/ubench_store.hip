// Store-pattern microbenchmark for the party-major output y[n][G] of the evaluation kernels (n = 16,
// G = 2^20, 32-byte elements): which store shape reaches which write bandwidth on MI355X.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_store.hip -o gpurun_out/ubench_store && ./gpurun_out/ubench_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ void nt_store(uint4 v, uint4* p) { v4u t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, (v4u*)p); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int N = 16;
// A: lane g writes its own 32-byte element as two 16-byte stores (stride 32 B between lanes) -- the current shape
template <bool NT> __global__ __launch_bounds__(64) void k_a(uint4* __restrict__ y, size_t G, uint32_t seed) {
    const size_t g = (size_t)blockIdx.x * 64 + threadIdx.x;
    uint4 v = make_uint4(seed, threadIdx.x, blockIdx.x, 7);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        uint4* p = y + ((size_t)j * G + g) * 2;
        v.x += j;
        if (NT) { nt_store(v, p); nt_store(v, p + 1); }
        else { p[0] = v; p[1] = v; }
    }
}
// B: each store instruction writes 1 KiB contiguous (lane l -> 16 bytes at l*16), two instructions per party row
template <bool NT> __global__ __launch_bounds__(64) void k_b(uint4* __restrict__ y, size_t G, uint32_t seed) {
    const size_t g0 = (size_t)blockIdx.x * 64;
    uint4 v = make_uint4(seed, threadIdx.x, blockIdx.x, 7);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        uint4* p = y + ((size_t)j * G + g0) * 2 + threadIdx.x;
        v.x += j;
        if (NT) { nt_store(v, p); nt_store(v, p + 64); }
        else { p[0] = v; p[64] = v; }
    }
}
// C: like A but 256-thread blocks (4 waves), D: like B with 256-thread blocks
__global__ __launch_bounds__(256) void k_c(uint4* __restrict__ y, size_t G, uint32_t seed) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint4 v = make_uint4(seed, threadIdx.x, blockIdx.x, 7);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        uint4* p = y + ((size_t)j * G + g) * 2;
        v.x += j;
        p[0] = v; p[1] = v;
    }
}
// R: read 192 B per lane chunk-major (coalesced 16-byte pieces) + write shape A or B
template <bool SHAPE_B> __global__ __launch_bounds__(64) void k_rw(const uint4* __restrict__ x, uint4* __restrict__ y, size_t G) {
    const size_t g0 = (size_t)blockIdx.x * 64;
    const uint4* src = x + g0 * 12;
    uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 12; ++i) { const uint4 t = src[i * 64 + threadIdx.x]; acc.x ^= t.x; acc.y += t.y; acc.z ^= t.z; acc.w += t.w; }
#pragma unroll
    for (int j = 0; j < N; ++j) {
        acc.x += j;
        if (SHAPE_B) { uint4* p = y + ((size_t)j * G + g0) * 2 + threadIdx.x; p[0] = acc; p[64] = acc; }
        else { uint4* p = y + ((size_t)j * G + g0 + threadIdx.x) * 2; p[0] = acc; p[1] = acc; }
    }
}
template <class F> float time_it(F launch, hipStream_t s) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e0, s);
    for (int i = 0; i < 50; ++i) launch();
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 50;
}
// A with a row stride that is not a power of two (rows of 2^20 elements are 32 MiB apart: do the 16 streams alias
// onto the same HBM channels?)
__global__ __launch_bounds__(64) void k_a_strided(uint4* __restrict__ y, size_t G, size_t stride, uint32_t seed) {
    const size_t g = (size_t)blockIdx.x * 64 + threadIdx.x;
    uint4 v = make_uint4(seed, threadIdx.x, blockIdx.x, 7);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        uint4* p = y + ((size_t)j * stride + g) * 2;
        v.x += j;
        p[0] = v; p[1] = v;
    }
}
__global__ __launch_bounds__(64) void k_rw_strided(const uint4* __restrict__ x, uint4* __restrict__ y, size_t G, size_t stride) {
    const size_t g0 = (size_t)blockIdx.x * 64;
    const uint4* src = x + g0 * 12;
    uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 12; ++i) { const uint4 t = src[i * 64 + threadIdx.x]; acc.x ^= t.x; acc.y += t.y; acc.z ^= t.z; acc.w += t.w; }
#pragma unroll
    for (int j = 0; j < N; ++j) {
        acc.x += j;
        uint4* p = y + ((size_t)j * stride + g0 + threadIdx.x) * 2; p[0] = acc; p[1] = acc;
    }
}
int main() {
    const size_t G = 1 << 20;
    uint4 *y, *x; CK(hipMalloc(&y, (G + 4096) * N * 32)); CK(hipMalloc(&x, G * 6 * 32)); CK(hipMemset(x, 1, G * 6 * 32));
    hipStream_t s; CK(hipStreamCreate(&s));
    const double wb = (double)G * N * 32, rb = (double)G * 6 * 32;
    float ms;
    ms = time_it([&] { hipLaunchKernelGGL(k_a<false>, dim3(G / 64), dim3(64), 0, s, y, G, 1u); }, s); printf("A  lane-owned 2x16B         : %7.1f us  %6.0f GB/s write\n", ms * 1e3, wb / ms / 1e6);
    ms = time_it([&] { hipLaunchKernelGGL(k_a<true>, dim3(G / 64), dim3(64), 0, s, y, G, 1u); }, s);  printf("A  + nontemporal             : %7.1f us  %6.0f GB/s write\n", ms * 1e3, wb / ms / 1e6);
    ms = time_it([&] { hipLaunchKernelGGL(k_b<false>, dim3(G / 64), dim3(64), 0, s, y, G, 1u); }, s); printf("B  1KiB-contiguous per instr : %7.1f us  %6.0f GB/s write\n", ms * 1e3, wb / ms / 1e6);
    ms = time_it([&] { hipLaunchKernelGGL(k_b<true>, dim3(G / 64), dim3(64), 0, s, y, G, 1u); }, s);  printf("B  + nontemporal             : %7.1f us  %6.0f GB/s write\n", ms * 1e3, wb / ms / 1e6);
    ms = time_it([&] { hipLaunchKernelGGL(k_c, dim3(G / 256), dim3(256), 0, s, y, G, 1u); }, s);       printf("C  shape A, 256-thread blocks: %7.1f us  %6.0f GB/s write\n", ms * 1e3, wb / ms / 1e6);
    ms = time_it([&] { hipLaunchKernelGGL(k_rw<false>, dim3(G / 64), dim3(64), 0, s, x, y, G); }, s);  printf("RW shape A (read 192B/lane)  : %7.1f us  %6.0f GB/s total\n", ms * 1e3, (wb + rb) / ms / 1e6);
    ms = time_it([&] { hipLaunchKernelGGL(k_rw<true>, dim3(G / 64), dim3(64), 0, s, x, y, G); }, s);   printf("RW shape B                   : %7.1f us  %6.0f GB/s total\n", ms * 1e3, (wb + rb) / ms / 1e6);
    for (size_t pad : {(size_t)0, (size_t)64, (size_t)192, (size_t)1000, (size_t)4096}) {
        ms = time_it([&] { hipLaunchKernelGGL(k_a_strided, dim3(G / 64), dim3(64), 0, s, y, G, G + pad, 1u); }, s);
        printf("A  row stride G + %-5zu       : %7.1f us  %6.0f GB/s write\n", pad, ms * 1e3, wb / ms / 1e6);
        ms = time_it([&] { hipLaunchKernelGGL(k_rw_strided, dim3(G / 64), dim3(64), 0, s, x, y, G, G + pad); }, s);
        printf("RW row stride G + %-5zu       : %7.1f us  %6.0f GB/s total\n", pad, ms * 1e3, (wb + rb) / ms / 1e6);
    }
    ms = time_it([&] { hipMemsetAsync(y, 0, G * N * 32, s); }, s);                                      printf("hipMemsetAsync               : %7.1f us  %6.0f GB/s write\n", ms * 1e3, wb / ms / 1e6);
    return 0;
}
