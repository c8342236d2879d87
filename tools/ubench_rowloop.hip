// ubench_rowloop.hip -- the row loop of the matrix-core kernels (kernels_mfma.hpp) with NO global memory traffic: what the
// SIMD makes of "bias from LDS, M x (A operand from LDS, MFMA), epilogue in registers" for W waves per workgroup.
// One workgroup per CU, 14 table rows of M = 11 slabs in its LDS, the B operands of a tile in registers; every wave walks
// ITER x 14 rows.  Printed: shader cycles per row and SIMD (s_memtime around the loop of wave 0 of every workgroup; rows per
// SIMD = ITER x 14 x waves per SIMD) against the 32 M cycles the matrix pipe needs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_rowloop.hip -o tools/ubench_rowloop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../mpc-protocols_amd/csrc/kernels_mfma.hpp"

using namespace hbmpc;
using namespace hbmpc::mf;
#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e)); \
            exit(2);                                                               \
        }                                                                          \
    } while (0)
constexpr int M = 11, ROWS = 14, SLABS = M * 1024;

// EPI: 0 none (the accumulator is folded into a sink with two XORs), 1 reduce_tile (an output row), 2 verify_tile
// BIAS: accumulator start value from the LDS (as the kernels do) or zero
// DL: LDS ring depth of the A operands; CG: tiles that share an A operand
template <int W, int DL, bool BIAS, int EPI, int CG, bool NOMFMA>
__global__ __launch_bounds__(64 * W) void k_rowloop(int iters, const v4i* seed, uint32_t* out, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    for (int p = threadIdx.x; p < (ROWS * SLABS + ROWS * 128) / 16; p += 64 * W) reinterpret_cast<v4i*>(lds)[p] = seed[p & 1023];
    __syncthreads();
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const Half H = make_half(h);
    v4i data[CG][M];
#pragma unroll
    for (int cg = 0; cg < CG; ++cg)
#pragma unroll
        for (int i = 0; i < M; ++i) data[cg][i] = seed[(threadIdx.x + 7 * i + 13 * cg) & 1023];
    v4i ys = seed[(threadIdx.x * 3) & 1023];
    uint32_t sink = 0;
    const uint8_t* bias_lds = lds + ROWS * SLABS + h * 64;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            v16i acc[CG];
            if constexpr (BIAS) {
                const v4i* bp = reinterpret_cast<const v4i*>(bias_lds + r * 128);
                const v4i b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
#pragma unroll
                for (int cg = 0; cg < CG; ++cg)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[cg][k] = b0[k], acc[cg][4 + k] = b1[k], acc[cg][8 + k] = b2[k], acc[cg][12 + k] = b3[k];
            } else {
#pragma unroll
                for (int cg = 0; cg < CG; ++cg)
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc[cg][k] = 0;
            }
            const uint8_t* tab_lane = lds + (size_t)r * SLABS + lane * 16;
            v4i av[DL];
#pragma unroll
            for (int i = 0; i < DL - 1 && i < M; ++i) av[i] = *reinterpret_cast<const v4i*>(tab_lane + i * 1024);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < M; ++i) {
                if (i + DL - 1 < M) av[(i + DL - 1) % DL] = *reinterpret_cast<const v4i*>(tab_lane + (i + DL - 1) * 1024);
#pragma unroll
                for (int cg = 0; cg < CG; ++cg) {
                    if constexpr (NOMFMA) acc[cg][i & 15] ^= av[i % DL][i & 3] ^ data[cg][i][(i + r) & 3];
                    else acc[cg] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[i % DL], data[cg][i], acc[cg], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) {
                if constexpr (EPI == 0) {
                    sink ^= (uint32_t)acc[cg][0] ^ (uint32_t)acc[cg][15];
                } else if constexpr (EPI == 1) {
                    uint32_t Rw[4];
                    reduce_tile(acc[cg], Rw, H);
                    sink ^= Rw[0] ^ Rw[1] ^ Rw[2] ^ Rw[3];
                } else {
                    sink |= verify_tile(acc[cg], ys, H);
                }
                asm volatile("" : "+v"(sink));
            }
        }
    }
    __syncthreads();  // the workgroup's last wave: shader cycles and 100 MHz ticks of the whole loop -> the clock the chip held
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0, cyc[256 + blockIdx.x] = r1 - r0;
}

static v4i* d_seed;
static uint32_t* d_out;
static long long* d_cyc;
template <int W, int DL, bool BIAS, int EPI, int CG = 1, bool NOMFMA = false>
static void run(const char* what) {
    const int iters = 200 / CG, blocks = 256;
    const size_t shm = ROWS * SLABS + ROWS * 128;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rowloop<W, DL, BIAS, EPI, CG, NOMFMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_rowloop<W, DL, BIAS, EPI, CG, NOMFMA>), dim3(blocks), dim3(64 * W), shm, 0, 3, d_seed, d_out, d_cyc);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_rowloop<W, DL, BIAS, EPI, CG, NOMFMA>), dim3(blocks), dim3(64 * W), shm, 0, iters, d_seed, d_out, d_cyc);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> cyc(2 * blocks);
    CK(hipMemcpy(cyc.data(), d_cyc, 2 * blocks * 8, hipMemcpyDeviceToHost));
    double sum = 0, rsum = 0;
    for (int b = 0; b < blocks; ++b) sum += (double)cyc[b], rsum += (double)cyc[blocks + b];
    const double ghz = sum / rsum * 0.1;
    const double rows_per_simd = (double)iters * ROWS * CG * (W / 4.0);  // 32-chunk row tiles per SIMD
    printf("%-58s W=%2d DL=%d CG=%d: %7.1f cycles per row tile and SIMD (matrix pipe alone: %d), %.0f ns, clock %.2f GHz\n", what, W, DL, CG,
           sum / blocks / rows_per_simd, NOMFMA ? 0 : 32 * M, ms * 1e6 / rows_per_simd, ghz);
    fflush(stdout);
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
}
int main() {
    CK(hipMalloc(&d_seed, 1024 * 16));
    std::vector<uint32_t> s(4096);
    uint64_t x = 88172645463325252ull;
    for (auto& v : s) x ^= x << 13, x ^= x >> 7, x ^= x << 17, v = (uint32_t)x;
    CK(hipMemcpy(d_seed, s.data(), 4096 * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, 256 * 1024 * 4));
    CK(hipMalloc(&d_cyc, 512 * 8));
    for (int rep = 0; rep < 2; ++rep) {
        run<12, 3, false, 0>("MFMA chain, A from LDS, no bias, no epilogue");
        run<12, 3, true, 0>("MFMA chain, A from LDS, bias from LDS, no epilogue");
        run<12, 5, true, 0>("MFMA chain, A from LDS, bias from LDS, no epilogue");
        run<8, 3, true, 0>("MFMA chain, A from LDS, bias from LDS, no epilogue");
        run<4, 3, true, 0>("MFMA chain, A from LDS, bias from LDS, no epilogue");
        run<4, 6, true, 0>("MFMA chain, A from LDS, bias from LDS, no epilogue");
        run<16, 3, true, 0>("MFMA chain, A from LDS, bias from LDS, no epilogue");
        run<12, 3, true, 1, 1, true>("no MFMA: LDS reads + output-row epilogue");
        run<12, 3, true, 2, 1, true>("no MFMA: LDS reads + verify-row epilogue");
        run<12, 3, true, 1>("full output row");
        run<12, 3, true, 2>("full verify row");
        run<12, 5, true, 1>("full output row");
        run<8, 3, true, 1>("full output row");
        run<8, 3, true, 2>("full verify row");
        run<16, 3, true, 1>("full output row");
        run<16, 3, true, 2>("full verify row");
        run<4, 3, true, 1>("full output row");
        run<8, 3, true, 1, 2>("full output row, two tiles per A operand");
        run<8, 3, true, 2, 2>("full verify row, two tiles per A operand");
        run<4, 3, true, 1, 2>("full output row, two tiles per A operand");
    }
    return 0;
}
