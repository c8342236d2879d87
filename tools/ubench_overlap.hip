// ubench_overlap.hip -- does matrix-core work of one wave overlap with vector work of its SIMD partners?
// Each wave loops: NM chained v_mfma_i32_32x32x32_i8, then NV vector instructions (v_mad_u64_u32 or v_add_u32, four
// independent chains) -- optionally consuming the accumulator first (DEP).  No memory traffic inside the loop.
// One block per CU; prints shader cycles per iteration (s_memtime, wave 0 of every block, averaged) for 1, 2 and 4
// waves per SIMD: with perfect overlap W waves cost W * max(matrix, vector); with none W * (matrix + vector).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_overlap.hip -o tools/ubench_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e)); \
            exit(2);                                                               \
        }                                                                          \
    } while (0)

template <int NM, int NV, int KIND, bool DEP>
__global__ __launch_bounds__(1024) void k(int iters, const v4i* in, uint32_t* out, long long* cyc) {
    v4i a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
    v16i acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (int)threadIdx.x + i;
    uint64_t c0 = threadIdx.x, c1 = threadIdx.x * 3, c2 = 7, c3 = 11;
    uint32_t d0 = threadIdx.x, d1 = 5, d2 = 7, d3 = 11;
    const uint32_t m0 = a[0] | 1, m1 = a[1] | 1;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NM; ++i) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
        if (DEP && NM > 0 && NV > 0) {
            c0 += (uint32_t)acc[0];
            d0 += (uint32_t)acc[1];
        }
#pragma unroll
        for (int i = 0; i < NV / 4; ++i) {
            if (KIND == 0) {
                uint64_t cu;
                asm volatile("v_mad_u64_u32 %0, %4, %5, %6, %0\n\tv_mad_u64_u32 %1, %4, %6, %5, %1\n\t"
                             "v_mad_u64_u32 %2, %4, %5, %6, %2\n\tv_mad_u64_u32 %3, %4, %6, %5, %3"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&s"(cu)
                             : "v"(m0), "v"(m1));
            } else {
                asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %5\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %5"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)
                             : "v"(m0), "v"(m1));
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t s = (uint32_t)(c0 + c1 + c2 + c3) + d0 + d1 + d2 + d3;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += (uint32_t)acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static v4i* d_in;
static uint32_t* d_out;
static long long* d_cyc;
template <int NM, int NV, int KIND, bool DEP>
static void run(const char* what) {
    const int iters = 2000, blocks = 256;
    printf("%-46s", what);
    for (int w : {1, 2, 4}) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        hipLaunchKernelGGL((k<NM, NV, KIND, DEP>), dim3(blocks), dim3(256 * w), 0, 0, 10, d_in, d_out, d_cyc);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k<NM, NV, KIND, DEP>), dim3(blocks), dim3(256 * w), 0, 0, iters, d_in, d_out, d_cyc);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<long long> cyc(blocks);
        CK(hipMemcpy(cyc.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost));
        double sum = 0;
        for (long long c : cyc) sum += (double)c;
        printf("  %dw/SIMD: %7.1f cyc/iter (%.3f ms)", w, sum / blocks / iters, ms);
    }
    printf("\n");
    fflush(stdout);
}
int main() {
    CK(hipMalloc(&d_in, 128 * 16));
    CK(hipMemset(d_in, 0x11, 128 * 16));
    CK(hipMalloc(&d_out, 256 * 1024 * 4));
    CK(hipMalloc(&d_cyc, 256 * 8));
    run<11, 0, 0, false>("11 MFMA i8 32x32x32 (chained)");
    run<0, 92, 0, false>("92 v_mad_u64_u32");
    run<0, 92, 1, false>("92 v_add_u32");
    run<11, 92, 0, true>("11 MFMA then 92 v_mad_u64_u32 (dependent)");
    run<11, 92, 0, false>("11 MFMA then 92 v_mad_u64_u32 (independent)");
    run<11, 92, 1, true>("11 MFMA then 92 v_add_u32 (dependent)");
    run<11, 92, 1, false>("11 MFMA then 92 v_add_u32 (independent)");
    run<22, 184, 0, true>("22 MFMA then 184 v_mad_u64_u32 (dependent)");
    return 0;
}
