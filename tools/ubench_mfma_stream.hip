// ubench_mfma_stream.hip -- the one-role decode with streamed overflow rows (csrc/kernels_mfma_stream.hpp) against the two-role
// k_mfma_rows<11,1,12,11> on config 3's decode shape (n = 31, t = d = 10: 10 verify rows + 11 coefficient rows, 2^20 chunks):
// whole outputs and status bytes compared, then both timed alternately.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_mfma_stream.hip -o tools/ubench_mfma_stream
//   tools/ubench_mfma_stream [log2_chunks=20] [reps=20]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <random>
#include <vector>

#include "kernels_mfma_stream.hpp"
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
static std::mt19937_64 rng(11);
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
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return ms / reps;
}
constexpr int M = 11;
template <int NR, int CG = 1, int W = 12>
static void launch_two_roles(mf::MfmaRowsArgs a, int rows) {
    constexpr int ROWB = M * 1024 + 128;
    if (!mf::mf_plan_roles(rows, a.nv, (160 * 1024) / ROWB, 256, &a)) exit(3);
    const size_t shm = (size_t)mf::mf_max_role_rows(a) * ROWB;
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mf::k_mfma_rows<M, CG, W, NR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
    hipLaunchKernelGGL((mf::k_mfma_rows<M, CG, W, NR>), dim3((unsigned)mf::mf_grid(a)), dim3(64 * W), shm, 0, a);
}
template <int W, int NV, int NO, int NRES, int D, int DL = 3, int YD = 1, bool DB = true>
static void launch_stream(mf::MfmaRowsArgs a, int nwg = 256) {
    const size_t shm = mf::mfs_lds_bytes(M, NRES, NV + NO);
    if (shm > 160 * 1024) { fprintf(stderr, "LDS %zu\n", shm); exit(3); }
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mf::k_mfma_rows_stream<M, W, NV, NO, NRES, D, DL, YD, DB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
    hipLaunchKernelGGL((mf::k_mfma_rows_stream<M, W, NV, NO, NRES, D, DL, YD, DB>), dim3((unsigned)nwg), dim3(64 * W), shm, 0, a);
}
int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 20, reps = argc > 2 ? atoi(argv[2]) : 20;
    const size_t G = ((size_t)1 << lg) + (lg < 20 ? 13 : 0);  // small runs are ragged
    constexpr int nv = 10, n = M + nv;
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
    // sender rows that pass the verification: y = [I ; Cv] x from the two-role kernel's encode, then decode with [Cv ; Co]
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
    uint8_t *d_te, *d_t, *d_x, *d_y, *d_out, *d_out2, *d_st, *d_st2;
    uint32_t *d_flag, *d_cnt, *d_sum, *d_nc, *d_nc2;
    CK(hipMalloc(&d_te, tenc.size() * 4));
    CK(hipMemcpy(d_te, tenc.data(), tenc.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_t, tab.size() * 4));
    CK(hipMemcpy(d_t, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    std::vector<uint64_t> x(G * M * 4);
    for (size_t i = 0; i < G * M; ++i) rand_canon(&x[4 * i]);
    {
        const uint64_t rm1[4] = {HFr::MOD[0] - 1, HFr::MOD[1], HFr::MOD[2], HFr::MOD[3]};
        for (int i = 0; i < M; ++i)
            for (int k = 0; k < 4; ++k) x[(0 * M + i) * 4 + k] = 0, x[(1 * M + i) * 4 + k] = rm1[k], x[(2 * M + i) * 4 + k] = (i & 1) ? rm1[k] : 0;
    }
    CK(hipMalloc(&d_x, G * M * 32));
    CK(hipMemcpy(d_x, x.data(), G * M * 32, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_y, (size_t)n * G * 32));
    CK(hipMalloc(&d_out, G * M * 32));
    CK(hipMalloc(&d_out2, G * M * 32));
    CK(hipMalloc(&d_st, G));
    CK(hipMalloc(&d_st2, G));
    CK(hipMalloc(&d_nc, G * 4));
    CK(hipMalloc(&d_nc2, G * 4));
    CK(hipMalloc(&d_flag, G * 4));
    CK(hipMalloc(&d_cnt, 128));
    CK(hipMalloc(&d_sum, 16));
    CK(hipMemset(d_cnt, 0, 128));
    {
        mf::MfmaRowsArgs e = {};
        e.in = d_x, e.G = G, e.in_chunk_major = 1, e.table = d_te, e.nv = 0, e.out = d_y, e.out_party_major = 1, e.out_stride = G;
        launch_two_roles<11>(e, n);
        CK(hipDeviceSynchronize());
    }
    // three corrupted chunks: a verify row, an interpolation row, the last chunk
    const size_t bad[3] = {G / 3, G / 2 + 1, G - 1};
    auto corrupt = [&](bool undo) {
        for (int k = 0; k < 3; ++k) {
            uint64_t v[4];
            uint8_t* p = d_y + ((size_t)(k == 1 ? 2 : M + 1 + k) * G + bad[k]) * 32;
            CK(hipMemcpy(v, p, 32, hipMemcpyDeviceToHost));
            v[k] ^= 1ull << (7 * k);
            CK(hipMemcpy(p, v, 32, hipMemcpyHostToDevice));
        }
        (void)undo;
    };
    mf::MfmaRowsArgs a = {};
    a.in = d_y, a.G = G, a.in_chunk_major = 0, a.row_stride = G, a.table = d_t, a.nv = nv, a.out = d_out, a.out_party_major = 0, a.out_stride = M;
    for (int i = 0; i < n; ++i) a.rows.set(i, i);
    a.status = d_st, a.ncoeffs = d_nc, a.flagged = d_flag, a.counters = d_cnt, a.summary = d_sum;
    mf::MfmaRowsArgs b = a;
    b.out = d_out2, b.status = d_st2, b.ncoeffs = d_nc2;
    int errors = 0;
    auto compare = [&](const char* what, bool flagged_expected) {
        CK(hipDeviceSynchronize());
        std::vector<uint8_t> o1(G * M * 32), o2(G * M * 32), s1(G), s2(G);
        std::vector<uint32_t> n1(G), n2(G);
        CK(hipMemcpy(o1.data(), d_out, o1.size(), hipMemcpyDeviceToHost));
        CK(hipMemcpy(o2.data(), d_out2, o2.size(), hipMemcpyDeviceToHost));
        CK(hipMemcpy(s1.data(), d_st, G, hipMemcpyDeviceToHost));
        CK(hipMemcpy(s2.data(), d_st2, G, hipMemcpyDeviceToHost));
        CK(hipMemcpy(n1.data(), d_nc, G * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(n2.data(), d_nc2, G * 4, hipMemcpyDeviceToHost));
        size_t diff = 0, sdiff = 0, ndiff = 0, nbad = 0;
        for (size_t g = 0; g < G; ++g) {
            // flagged chunks: the two-role kernel writes their coefficient rows anyway (its second role does not know the verdict), so does this one
            if (memcmp(&o1[g * M * 32], &o2[g * M * 32], M * 32) != 0 && diff++ < 3) fprintf(stderr, "%s: chunk %zu differs\n", what, g);
            sdiff += s1[g] != s2[g];
            if (s1[g] == 0) ndiff += n1[g] != n2[g];
            nbad += s2[g] != 0;
        }
        if (diff || sdiff || ndiff || (flagged_expected ? nbad != 3 : nbad != 0)) {
            fprintf(stderr, "%s: %zu chunks differ, %zu status bytes, %zu ncoeffs; %zu chunks flagged\n", what, diff, sdiff, ndiff, nbad);
            ++errors;
        } else {
            fprintf(stderr, "%s: identical to k_mfma_rows<11,1,12,11> (%zu chunks, %zu flagged)\n", what, G, nbad);
        }
    };
    auto reset = [&] {
        CK(hipMemset(d_out2, 0xEE, G * M * 32));
        CK(hipMemset(d_st2, 0xEE, G));
        CK(hipMemset(d_nc2, 0xEE, G * 4));
        CK(hipMemset(d_cnt, 0, 128));
    };
    corrupt(false);
    CK(hipMemset(d_out, 0xEE, G * M * 32));
    CK(hipMemset(d_st, 0xEE, G));
    CK(hipMemset(d_nc, 0xEE, G * 4));
    launch_two_roles<11>(a, n);
    CK(hipDeviceSynchronize());
#define CHECK(W, NRES, D)                                 \
    reset();                                              \
    launch_stream<W, nv, M, NRES, D>(b);                  \
    compare("stream W=" #W " NRES=" #NRES " D=" #D " (3 corrupted chunks)", true);
#ifndef HBMPC_MFS_TIMING_ONLY_LDS_SLABS
    CHECK(12, 14, 4)
    CHECK(12, 14, 6)
    CHECK(12, 13, 4)
    CHECK(8, 14, 8)
    CHECK(8, 14, 4)
    CHECK(12, 14, 3)
#endif
#ifndef HBMPC_MFS_TIMING_ONLY_LDS_SLABS
    // the clean batch
    {
        mf::MfmaRowsArgs e = {};
        e.in = d_x, e.G = G, e.in_chunk_major = 1, e.table = d_te, e.nv = 0, e.out = d_y, e.out_party_major = 1, e.out_stride = G;
        launch_two_roles<11>(e, n);
        CK(hipMemset(d_cnt, 0, 128));
        launch_two_roles<11>(a, n);
        CK(hipDeviceSynchronize());
    }
    reset();
    launch_stream<12, nv, M, 14, 4>(b);
    compare("stream W=12 NRES=14 D=4 (clean)", false);

#else
    fprintf(stderr, "TIMING ONLY: the streamed slabs are read from the LDS (wrong operands)\n");
    {
        mf::MfmaRowsArgs e = {};
        e.in = d_x, e.G = G, e.in_chunk_major = 1, e.table = d_te, e.nv = 0, e.out = d_y, e.out_party_major = 1, e.out_stride = G;
        launch_two_roles<11>(e, n);
        CK(hipMemset(d_cnt, 0, 128));
        CK(hipDeviceSynchronize());
    }
#endif
    if (lg < 18) {
        fprintf(stderr, errors ? "FAILED: %d\n" : "all checks passed\n", errors);
        return errors != 0;
    }
    for (int i = 0; i < 1200; ++i) launch_two_roles<11>(a, n);  // clocks up
    CK(hipDeviceSynchronize());
    const double algo = (double)(n + M) * 32 * G;
    for (int round = 0; round < 2; ++round) {
        const float t0 = time_ms([&] { launch_two_roles<11>(a, n); }, reps);
        printf("two roles k_mfma_rows<11,1,12,11>: %.4f ms (%.2f TB/s algorithmic)\n", t0, algo / t0 / 1e9);
#define TIME(W, NRES, D, DL, YD, DB)                                                          \
    {                                                                                      \
        const float tt = time_ms([&] { launch_stream<W, nv, M, NRES, D, DL, YD, DB>(b); }, reps); \
        printf("one role, W=" #W " resident=" #NRES " ring=" #D " lds_ring=" #DL " ys_ahead=" #YD " two_sets=" #DB ": %.4f ms (%.2f TB/s algorithmic)\n", tt, algo / tt / 1e9); \
    }
        TIME(12, 14, 4, 3, 1, true)
        TIME(12, 14, 4, 3, 1, false)
        TIME(16, 14, 4, 3, 1, false)
        TIME(8, 14, 4, 3, 1, true)
        fflush(stdout);
    }
    uint32_t cnt[4];
    CK(hipMemcpy(cnt, d_cnt, 16, hipMemcpyDeviceToHost));
    if (cnt[0] != 0) fprintf(stderr, "%u chunks flagged in the clean timed runs\n", cnt[0]), ++errors;
    fprintf(stderr, errors ? "FAILED: %d\n" : "all checks passed\n", errors);
    return errors != 0;
}
