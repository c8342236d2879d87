// What this box's HBM delivers for IDEAL streams (16 bytes per lane, consecutive lanes consecutive addresses, one pass over buffers far
// larger than the Infinity Cache) at different read : write mixes -- the yardstick for the headline kernel's 27 % read / 73 % write stream
// (compute_shares n = 16, t = 5: 192 B read and 512 B written per secret).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_hbm_mix tools/ubench_hbm_mix.hip && /tmp/ubench_hbm_mix
// R loads and W stores of 16 bytes per lane and step; R = 0: write only, W = 0: read only (a sum keeps the loads alive).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int R, int W>
__global__ __launch_bounds__(256) void k_mix(const uint4* __restrict__ x, uint4* __restrict__ y, size_t steps, uint4* __restrict__ sink) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (size_t)gridDim.x * blockDim.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t s = tid; s < steps; s += nthreads) {
        uint4 v[R > 0 ? R : 1];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = x[(size_t)r * steps + s];          // R streams, each contiguous across lanes
#pragma unroll
        for (int r = 0; r < R; ++r) acc.x ^= v[r].x, acc.y += v[r].y, acc.z ^= v[r].z, acc.w += v[r].w;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            uint4 o = acc;
            o.x += (uint32_t)w;
            y[(size_t)w * steps + s] = o;                                       // W streams
        }
    }
    if (W == 0 && acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[tid & 63] = acc;  // never true in practice: keeps the loads
}

template <int R, int W>
int run(const uint4* x, uint4* y, uint4* sink, size_t steps, int blocks, const char* what) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_mix<R, W>), dim3(blocks), dim3(256), 0, 0, x, y, steps, sink);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_mix<R, W>), dim3(blocks), dim3(256), 0, 0, x, y, steps, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double bytes = (double)(R + W) * steps * 16;
    printf("%-44s %2d loads : %2d stores  (%4.1f %% read)  %7.1f MB  %7.1f us  %6.0f GB/s\n", what, R, W, 100.0 * R / (R + W), bytes / 1e6, ms * 1e3, bytes / ms / 1e6);
    return 0;
}

int main() {
    const size_t steps = (size_t)1 << 22;  // 64 MB per stream
    uint4 *x, *y, *sink;
    CK(hipMalloc(&x, steps * 16 * 16)); CK(hipMalloc(&y, steps * 16 * 16)); CK(hipMalloc(&sink, 1024));
    CK(hipMemset(x, 1, steps * 16 * 16)); CK(hipMemset(y, 0, steps * 16 * 16));
    for (int blocks : {1024, 2048, 4096}) {
        printf("-- %d workgroups of 256 lanes, 64 MB per stream\n", blocks);
        if (run<8, 0>(x, y, sink, steps, blocks, "read only")) return 1;
        if (run<0, 8>(x, y, sink, steps, blocks, "write only")) return 1;
        if (run<4, 4>(x, y, sink, steps, blocks, "copy")) return 1;
        if (run<3, 8>(x, y, sink, steps, blocks, "the headline's mix (3 : 8)")) return 1;
        if (run<6, 16>(x, y, sink, steps, blocks, "the headline's streams (6 in, 16 out)")) return 1;
        if (run<8, 4>(x, y, sink, steps, blocks, "config 4's mix (2 : 1)")) return 1;
    }
    float ms; hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0)); for (int i = 0; i < 10; ++i) CK(hipMemcpyAsync(y, x, steps * 16 * 8, hipMemcpyDeviceToDevice, 0)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("hipMemcpyAsync device to device, 512 MB: %7.1f us  %6.0f GB/s (read + write)\n", ms * 1e3, 2.0 * steps * 16 * 8 / ms / 1e6);
    CK(hipEventRecord(e0)); for (int i = 0; i < 10; ++i) CK(hipMemsetAsync(y, 0, steps * 16 * 8, 0)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("hipMemsetAsync, 512 MB: %7.1f us  %6.0f GB/s\n", ms * 1e3, 1.0 * steps * 16 * 8 / ms / 1e6);
    return 0;
}
