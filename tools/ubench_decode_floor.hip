// ubench_decode_floor.hip -- what bounds the config 3 decode kernel (k_mfma_rows<11,1,12,11>, two roles): the same launch with
// the epilogue arithmetic, the MFMAs or both compiled out (every load and store stays).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_decode_floor.hip -o tools/ubench_decode_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <random>
#include <vector>

#include "kernels_mfma_lab.hpp"
#include "../mpc-protocols_amd/csrc/tables_mfma.hpp"

using namespace hbmpc;
#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e)); \
            exit(2);                                                               \
        }                                                                          \
    } while (0)
static std::mt19937_64 rng(7);
static void rand_canon(uint64_t c[4]) {
    for (;;) {
        for (int i = 0; i < 4; ++i) c[i] = rng();
        c[3] &= 0x7fffffffffffffffULL;
        if (!HFr::geq(c)) return;
    }
}
template <class F>
static float time_ms(F f, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) f();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}
template <int W, int ABL>
static void launch(mf::MfmaRowsArgs a, int rows) {
    constexpr int M = 11, ROWB = M * 1024 + 128;
    if (!mf::mf_plan_roles(rows, a.nv, (160 * 1024) / ROWB, 256, &a)) exit(3);
    const size_t shm = (size_t)mf::mf_max_role_rows(a) * ROWB;
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mf::k_mfma_rows_lab<M, 1, W, 11, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
    hipLaunchKernelGGL((mf::k_mfma_rows_lab<M, 1, W, 11, ABL>), dim3((unsigned)mf::mf_grid(a)), dim3(64 * W), shm, 0, a);
}
int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 20, reps = argc > 2 ? atoi(argv[2]) : 20;
    const size_t G = (size_t)1 << lg;
    constexpr int M = 11, nv = 10, n = M + nv;
    auto rand_rows = [&](int rows) {
        std::vector<std::vector<HFr>> C(rows, std::vector<HFr>(M));
        for (auto& row : C)
            for (auto& v : row) {
                uint64_t c[4];
                rand_canon(c);
                v = HFr::from_canon(c);
            }
        return C;
    };
    // sender rows that pass the verification: y = [I ; Cv] x from this kernel's own encode, then decode with [Cv ; Co]
    const auto Cv = rand_rows(nv), Co = rand_rows(M);
    std::vector<std::vector<HFr>> Cenc, Cdec = Cv;
    for (int i = 0; i < M; ++i) {
        std::vector<HFr> row(M, HFr::zero());
        row[i] = HFr::one();
        Cenc.push_back(row);
    }
    for (auto& r : Cv) Cenc.push_back(r);
    for (auto& r : Co) Cdec.push_back(r);
    const auto tenc = build_mfma_table(Cenc, M), tab = build_mfma_table(Cdec, M);
    uint8_t *d_te, *d_t, *d_x, *d_y, *d_out, *d_st;
    uint32_t *d_flag, *d_cnt, *d_sum;
    CK(hipMalloc(&d_te, tenc.size() * 4));
    CK(hipMemcpy(d_te, tenc.data(), tenc.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_t, tab.size() * 4));
    CK(hipMemcpy(d_t, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    std::vector<uint64_t> x(G * M * 4);
    for (size_t i = 0; i < G * M; ++i) rand_canon(&x[4 * i]);
    CK(hipMalloc(&d_x, G * M * 32));
    CK(hipMemcpy(d_x, x.data(), G * M * 32, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_y, (size_t)n * G * 32));
    CK(hipMalloc(&d_out, G * M * 32));
    CK(hipMalloc(&d_st, G));
    CK(hipMalloc(&d_flag, G * 4));
    CK(hipMalloc(&d_cnt, 64));
    CK(hipMalloc(&d_sum, 16));
    CK(hipMemset(d_cnt, 0, 64));
    {
        mf::MfmaRowsArgs e = {};
        e.in = d_x, e.G = G, e.in_chunk_major = 1, e.table = d_te, e.nv = 0, e.out = d_y, e.out_party_major = 1, e.out_stride = G;
        launch<12, 0>(e, n);
        CK(hipDeviceSynchronize());
    }
    mf::MfmaRowsArgs a = {};
    a.in = d_y, a.G = G, a.in_chunk_major = 0, a.row_stride = G, a.table = d_t, a.nv = nv, a.out = d_out, a.out_party_major = 0, a.out_stride = M;
    for (int i = 0; i < n; ++i) a.rows.set(i, i);
    a.status = d_st, a.flagged = d_flag, a.counters = d_cnt, a.summary = d_sum;
    auto run = [&](auto f) {
        CK(hipMemset(d_cnt, 0, 64));
        return time_ms([&] { f(); CK(hipMemsetAsync(d_cnt, 0, 64, 0)); }, reps);
    };
    launch<12, 0>(a, n);
    uint32_t cnt[4];
    CK(hipMemcpy(cnt, d_cnt, 16, hipMemcpyDeviceToHost));
    printf("flagged chunks in the full kernel: %u (0 expected)\n", cnt[0]);
    for (int i = 0; i < 1500; ++i) launch<12, 0>(a, n);  // half a second: clocks up
    CK(hipDeviceSynchronize());
    const float f0 = run([&] { launch<12, 0>(a, n); }), f1 = run([&] { launch<12, 1>(a, n); }), f2 = run([&] { launch<12, 2>(a, n); }),
                f3 = run([&] { launch<12, 3>(a, n); });
    const float g0 = run([&] { launch<12, 0>(a, n); }), g3 = run([&] { launch<12, 3>(a, n); });
    printf("again: full %.4f, loads and stores only %.4f\n", g0, g3);
    const double algo = (double)(n + M) * 32 * G, actual = (double)(2 * M + nv + M) * 32 * G;
    printf("config 3 decode, 2^%d chunks, k_mfma_rows<11,1,12,11> (verify role + coefficient role): full %.4f ms | no epilogue arithmetic %.4f | "
           "no MFMA %.4f | loads and stores only %.4f\n", lg, f0, f1, f2, f3);
    printf("   algorithmic %.0f MB (%.2f TB/s at the full time), with the second role's reads %.0f MB (%.2f TB/s; %.2f TB/s in the loads-and-stores run)\n",
           algo / 1e6, algo / f0 / 1e9, actual / 1e6, actual / f0 / 1e9, actual / f3 / 1e9);
    return 0;
}
