// ubench_mfma_bfly.hip -- the paired-point encode (csrc/kernels_mfma_bfly.hpp) against the plain matrix-core encode
// (k_mfma_rows) on the BASELINE encode shapes: byte comparison of the whole output, a host check of sampled chunks, ms per launch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_mfma_bfly.hip -o tools/ubench_mfma_bfly
//   tools/ubench_mfma_bfly [log2_chunks=20] [reps=20] [workgroups=256]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <random>
#include <vector>

#include "kernels_mfma_bfly_lab.hpp"
#include "kernels_mfma_lab.hpp"
#include "../mpc-protocols_amd/csrc/tables.hpp"
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

static std::mt19937_64 rng(0xB0F1);
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
static int g_nwg = 256;
static int bad_lines = 0;

template <int M, int WAVES, int NR, int ABL = 0>
static void launch_plain(mf::MfmaRowsArgs a, int rows) {
    constexpr int ROWB = M * 1024 + 128;
    if (!mf::mf_plan_roles(rows, 0, (160 * 1024) / ROWB, 256, &a)) exit(3);
    const size_t shm = (size_t)mf::mf_max_role_rows(a) * ROWB;
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mf::k_mfma_rows_lab<M, 1, WAVES, NR, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
    hipLaunchKernelGGL((mf::k_mfma_rows_lab<M, 1, WAVES, NR, ABL>), dim3((unsigned)mf::mf_grid(a)), dim3(64 * WAVES), shm, 0, a);
}
template <int M, int WAVES, int NP, int ABL = 0>
static void launch_bfly(mf::MfmaRowsArgs a, int pairs, int wgs_per_cu) {
    constexpr int ROWB = M * 1024 + 256;
    const int cap = (160 * 1024 / wgs_per_cu) / ROWB;
    if (!mf::mf_plan_roles(pairs, 0, cap, g_nwg * wgs_per_cu, &a)) exit(3);
    const size_t shm = (size_t)mf::mf_max_role_rows(a) * ROWB;
    if (NP > 0 && mf::mf_max_role_rows(a) > NP) exit(4);
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mf::k_mfma_bfly_lab<M, WAVES, NP, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
    hipLaunchKernelGGL((mf::k_mfma_bfly_lab<M, WAVES, NP, ABL>), dim3((unsigned)mf::mf_grid(a)), dim3(64 * WAVES), shm, 0, a);
}

// the fused local product + encode of triple generation: y[P][n][G] from a, b, r2t [P][G][M]
template <int M, int WAVES, int NP, int ABL = 0, int SD = 1>
static void launch_triple(mf::MfmaRowsArgs a, int pairs) {
    constexpr int ROWB = M * 1024 + 256;
    if (!mf::mf_plan_pairs(pairs, (160 * 1024) / ROWB, g_nwg, &a)) exit(3);
    const size_t shm = (size_t)mf::mf_max_role_rows(a) * ROWB;
    if (mf::mf_max_role_rows(a) != NP || a.nroles != 1) exit(4);
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mf::k_mfma_bfly_lab<M, WAVES, NP, ABL, true, SD>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
    hipLaunchKernelGGL((mf::k_mfma_bfly_lab<M, WAVES, NP, ABL, true, SD>), dim3((unsigned)mf::mf_grid(a)), dim3(64 * WAVES), shm, 0, a);
}
template <int M, int NP>
static int run_triple(int n, size_t G, int P, int reps) {
    const size_t size = domain_size(n), half = size / 2;
    std::vector<HFr> el = domain_elements<HFr>(n, n);
    std::vector<std::vector<HFr>> V(n, std::vector<HFr>(M));
    for (int j = 0; j < n; ++j) {
        HFr p = HFr::one();
        for (int k = 0; k < M; ++k) V[j][k] = p, p = p * el[j];
    }
    std::vector<std::vector<HFr>> VR = V;  // the kernel hands over (a b - r2t) / R: the table rows carry R = 2^261
    {
        HFr R = HFr::one();
        const HFr two = HFr::from_u64(2);
        for (int i = 0; i < 261; ++i) R = R * two;
        for (auto& row : VR)
            for (auto& v : row) v = v * R;
    }
    const auto tb = build_mfma_bfly_table(VR, M, half);
    uint8_t *d_tb, *d_a, *d_b, *d_r, *d_y;
    CK(hipMalloc(&d_tb, tb.size() * 4));
    CK(hipMemcpy(d_tb, tb.data(), tb.size() * 4, hipMemcpyHostToDevice));
    const size_t N = (size_t)P * G * M;
    std::vector<uint64_t> av(N * 4), bv(N * 4), rv(N * 4);
    for (size_t i = 0; i < N; ++i) rand_canon(&av[4 * i]), rand_canon(&bv[4 * i]), rand_canon(&rv[4 * i]);
    {
        const uint64_t rm1[4] = {HFr::MOD[0] - 1, HFr::MOD[1], HFr::MOD[2], HFr::MOD[3]};
        for (int i = 0; i < M; ++i)
            for (int k = 0; k < 4; ++k) {
                av[(0 * M + i) * 4 + k] = 0, rv[(0 * M + i) * 4 + k] = rm1[k];                  // 0 * b - (r - 1) = 1
                av[(1 * M + i) * 4 + k] = rm1[k], bv[(1 * M + i) * 4 + k] = rm1[k], rv[(1 * M + i) * 4 + k] = 0;  // (-1)(-1) - 0
                av[(2 * M + i) * 4 + k] = rm1[k], bv[(2 * M + i) * 4 + k] = k == 0, rv[(2 * M + i) * 4 + k] = rm1[k];  // 0
            }
    }
    CK(hipMalloc(&d_a, N * 32)); CK(hipMalloc(&d_b, N * 32)); CK(hipMalloc(&d_r, N * 32));
    CK(hipMalloc(&d_y, (size_t)P * n * G * 32));
    CK(hipMemcpy(d_a, av.data(), N * 32, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_b, bv.data(), N * 32, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_r, rv.data(), N * 32, hipMemcpyHostToDevice));
    CK(hipMemset(d_y, 0xee, (size_t)P * n * G * 32));
    mf::MfmaRowsArgs a = {};
    a.in = d_a, a.in_b = d_b, a.in_r = d_r, a.parties = P, a.G = G, a.in_chunk_major = 1, a.nv = 0, a.out = d_y, a.out_party_major = 1, a.out_stride = G;
    a.table = d_tb, a.half = (int)half, a.nout = n;
    launch_triple<M, 8, NP>(a, (int)half);
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> y((size_t)P * n * G * 4);
    CK(hipMemcpy(y.data(), d_y, y.size() * 8, hipMemcpyDeviceToHost));
    int errors = 0;
    for (size_t s = 0; s < 96; ++s) {
        const size_t pp = s % P, g = s < 8 ? s / 2 : (s < 16 ? G - 1 - s : rng() % G);
        for (int j = 0; j < n; ++j) {
            HFr acc = HFr::zero();
            for (int k = 0; k < M; ++k) {
                const size_t e = ((pp * G + g) * M + k) * 4;
                acc = acc + V[j][k] * (HFr::from_canon(&av[e]) * HFr::from_canon(&bv[e]) - HFr::from_canon(&rv[e]));
            }
            uint64_t w[4];
            acc.to_canon(w);
            for (int k = 0; k < 4; ++k) errors += w[k] != y[((pp * n + j) * G + g) * 4 + k];
        }
    }
    const float t8 = time_ms([&] { launch_triple<M, 8, NP>(a, (int)half); }, reps);
    const float t12 = time_ms([&] { launch_triple<M, 12, NP>(a, (int)half); }, reps);
    const float d2_12 = time_ms([&] { launch_triple<M, 12, NP, 0, 2>(a, (int)half); }, reps);
    const float d2_8 = time_ms([&] { launch_triple<M, 8, NP, 0, 2>(a, (int)half); }, reps);
    const float d3_8 = time_ms([&] { launch_triple<M, 8, NP, 0, 3>(a, (int)half); }, reps);
    printf("slots requested 2 ahead: 12 waves %.4f ms, 8 waves %.4f; 3 ahead, 8 waves %.4f\n", d2_12, d2_8, d3_8);
    const double bytes = (double)P * G * (3 * M + n) * 32;
    printf("fused local product + encode, n=%d m=%d, %d parties x %zu chunks: host check of sampled chunks: %d errors; 8 waves %.4f ms (%.2f TB/s), 12 waves %.4f ms (%.2f TB/s)\n",
           n, M, P, G, errors, t8, bytes / t8 / 1e9, t12, bytes / t12 / 1e9);
#ifdef BFLY_ABLATE
    if (G > 100000) {
        const float a3 = time_ms([&] { launch_triple<M, 12, NP, 3>(a, (int)half); }, reps);
        const float a8 = time_ms([&] { launch_triple<M, 12, NP, 8>(a, (int)half); }, reps);
        const float a11 = time_ms([&] { launch_triple<M, 12, NP, 11>(a, (int)half); }, reps);
        printf("   12 waves: products only (no MFMA, no epilogue arithmetic) %.4f ms | no products %.4f | loads and stores only %.4f\n", a3, a8, a11);
    }
#endif
    CK(hipFree(d_tb)); CK(hipFree(d_a)); CK(hipFree(d_b)); CK(hipFree(d_r)); CK(hipFree(d_y));
    return errors;
}

template <int M, int WAVES, int NP, int ABL = 0>
static void launch_lines(mf::MfmaRowsArgs a, int pairs) {
    constexpr int ROWB = M * 1024 + 256;
    if (!mf::mf_plan_pairs(pairs, (160 * 1024) / ROWB, g_nwg, &a)) exit(3);
    if (mf::mf_max_role_rows(a) != NP || a.nroles != 1) exit(4);
    const size_t shm = (size_t)NP * ROWB + (size_t)WAVES * mf::bfly_slot_bytes<M>();
    if (shm > 160 * 1024) exit(5);
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mf::k_mfma_bfly_lab<M, WAVES, NP, ABL, false, 1, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
    hipLaunchKernelGGL((mf::k_mfma_bfly_lab<M, WAVES, NP, ABL, false, 1, false, true>), dim3((unsigned)mf::mf_grid(a)), dim3(64 * WAVES), shm, 0, a);
}

template <int M, int WP, int NRP, int NP>
static int run(const char* name, int n, size_t G, int reps) {
    const size_t size = domain_size(n), half = size / 2;
    std::vector<HFr> el = domain_elements<HFr>(n, n);
    std::vector<std::vector<HFr>> V(n, std::vector<HFr>(M));
    for (int j = 0; j < n; ++j) {
        HFr p = HFr::one();
        for (int k = 0; k < M; ++k) V[j][k] = p, p = p * el[j];
    }
    const auto tp = build_mfma_table(V, M), tb = build_mfma_bfly_table(V, M, half);
    uint8_t *d_tp, *d_tb, *d_x, *d_y0, *d_y1;
    CK(hipMalloc(&d_tp, tp.size() * 4));
    CK(hipMalloc(&d_tb, tb.size() * 4));
    CK(hipMemcpy(d_tp, tp.data(), tp.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tb, tb.data(), tb.size() * 4, hipMemcpyHostToDevice));
    std::vector<uint64_t> x(G * M * 4);
    for (size_t i = 0; i < G * M; ++i) rand_canon(&x[4 * i]);
    {
        const uint64_t rm1[4] = {HFr::MOD[0] - 1, HFr::MOD[1], HFr::MOD[2], HFr::MOD[3]};
        for (int i = 0; i < M; ++i)
            for (int k = 0; k < 4; ++k) {
                x[(0 * M + i) * 4 + k] = 0;
                x[(1 * M + i) * 4 + k] = rm1[k];
                x[(2 * M + i) * 4 + k] = k == 0 ? 1 : 0;
                x[(3 * M + i) * 4 + k] = (i & 1) ? rm1[k] : 0;
                x[(4 * M + i) * 4 + k] = (i & 1) ? 0 : rm1[k];
            }
    }
    CK(hipMalloc(&d_x, G * M * 32));
    CK(hipMalloc(&d_y0, (size_t)n * G * 32));
    CK(hipMalloc(&d_y1, (size_t)n * G * 32));
    CK(hipMemcpy(d_x, x.data(), G * M * 32, hipMemcpyHostToDevice));
    CK(hipMemset(d_y1, 0xee, (size_t)n * G * 32));
    mf::MfmaRowsArgs a = {};
    a.in = d_x, a.G = G, a.in_chunk_major = 1, a.nv = 0, a.out_party_major = 1, a.out_stride = G;
    mf::MfmaRowsArgs pa = a, ba = a;
    pa.table = d_tp, pa.out = d_y0;
    ba.table = d_tb, ba.out = d_y1, ba.half = (int)half, ba.nout = n;
    launch_plain<M, WP, NRP>(pa, n);
    launch_bfly<M, 8, NP>(ba, (int)half, 1);
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> y0((size_t)n * G * 4), y1((size_t)n * G * 4);
    CK(hipMemcpy(y0.data(), d_y0, y0.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(y1.data(), d_y1, y1.size() * 8, hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < y0.size(); ++i) diff += y0[i] != y1[i];
    int errors = 0;
    for (size_t s = 0; s < 64; ++s) {  // host check of the plain kernel on sampled chunks (the edge chunks first)
        const size_t g = s < 8 ? s : (rng() % G);
        for (int j = 0; j < n; ++j) {
            HFr acc = HFr::zero();
            for (int k = 0; k < M; ++k) acc = acc + V[j][k] * HFr::from_canon(&x[(g * M + k) * 4]);
            uint64_t w[4];
            acc.to_canon(w);
            for (int k = 0; k < 4; ++k) errors += w[k] != y1[((size_t)j * G + g) * 4 + k];
        }
    }
    printf("%s: n=%d m=%d chunks=%zu: paired-point output vs plain: %zu words differ; host check of sampled chunks: %d errors\n", name, n, M, G, diff, errors);
    const float tp_ms = time_ms([&] { launch_plain<M, WP, NRP>(pa, n); }, reps);
    printf("   plain k_mfma_rows<%d,1,%d,%d>: %.4f ms\n", M, WP, NRP, tp_ms);
    const float b8 = time_ms([&] { launch_bfly<M, 8, NP>(ba, (int)half, 1); }, reps);
    const float b12 = time_ms([&] { launch_bfly<M, 12, NP>(ba, (int)half, 1); }, reps);
    const float b16 = time_ms([&] { launch_bfly<M, 16, NP>(ba, (int)half, 1); }, reps);
    printf("   paired points, one workgroup per CU: 8 waves %.4f ms, 12 waves %.4f, 16 waves %.4f\n", b8, b12, b16);
    if (2 * ((size_t)NP * (M * 1024 + 256)) <= 160 * 1024) {
        const float c4 = time_ms([&] { launch_bfly<M, 4, NP>(ba, (int)half, 2); }, reps);
        const float c6 = time_ms([&] { launch_bfly<M, 6, NP>(ba, (int)half, 2); }, reps);
        const float c8 = time_ms([&] { launch_bfly<M, 8, NP>(ba, (int)half, 2); }, reps);
        printf("   paired points, two workgroups per CU: 4 waves %.4f ms, 6 waves %.4f, 8 waves %.4f\n", c4, c6, c8);
    }
#ifdef BFLY_ABLATE
    {
        const float a1 = time_ms([&] { launch_bfly<M, 12, NP, 1>(ba, (int)half, 1); }, reps);
        const float a2 = time_ms([&] { launch_bfly<M, 12, NP, 2>(ba, (int)half, 1); }, reps);
        const float a6 = time_ms([&] { launch_bfly<M, 12, NP, 6>(ba, (int)half, 1); }, reps);
        const float a3 = time_ms([&] { launch_bfly<M, 12, NP, 3>(ba, (int)half, 1); }, reps);
        const float a7 = time_ms([&] { launch_bfly<M, 12, NP, 7>(ba, (int)half, 1); }, reps);
        const float p3 = time_ms([&] { launch_plain<M, WP, NRP, 3>(pa, n); }, reps);
        const float p3s = time_ms([&] { launch_plain<M, WP, (M == 6 ? 16 : NRP), 3>(pa, n); }, reps);
        printf("   plain kernel, neither MFMA nor epilogue: %.4f (static rows %.4f)\n", p3, p3s);
        const float b3 = time_ms([&] { launch_bfly<M, 12, 0, 3>(ba, (int)half, 1); }, reps);
        const float c3 = time_ms([&] { launch_bfly<M, 16, 0, 3>(ba, (int)half, 1); }, reps);
        const float d3 = time_ms([&] { launch_bfly<M, 16, NP, 3>(ba, (int)half, 1); }, reps);
        const float e3 = time_ms([&] { launch_bfly<M, 8, NP, 3>(ba, (int)half, 2); }, reps);
        printf("   neither MFMA nor epilogue: run-time pair loop with plain stores 12 waves %.4f, 16 waves %.4f; static 16 waves %.4f; static 2 x 8 waves %.4f\n", b3, c3, d3, e3);
        printf("   ablations, 12 waves: no epilogue %.4f | no MFMA %.4f | no MFMA, no table reads %.4f | neither MFMA nor epilogue %.4f | memory traffic only %.4f\n", a1, a2, a6, a3, a7);
    }
#endif
    if constexpr (M <= 8) {
        if ((size_t)NP == half) {
            CK(hipMemset(d_y1, 0xee, (size_t)n * G * 32));
            launch_lines<M, 12, NP>(ba, (int)half);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(y1.data(), d_y1, y1.size() * 8, hipMemcpyDeviceToHost));
            size_t dd = 0;
            for (size_t i = 0; i < y0.size(); ++i) dd += y0[i] != y1[i];
            const float l8 = time_ms([&] { launch_lines<M, 8, NP>(ba, (int)half); }, reps);
            const float l12 = time_ms([&] { launch_lines<M, 12, NP>(ba, (int)half); }, reps);
            const float l16 = time_ms([&] { launch_lines<M, 16, NP>(ba, (int)half); }, reps);
#ifdef BFLY_ABLATE
            {
                const float e1 = time_ms([&] { launch_lines<M, 12, NP, 1>(ba, (int)half); }, reps);
                const float e2 = time_ms([&] { launch_lines<M, 12, NP, 2>(ba, (int)half); }, reps);
                const float e3 = time_ms([&] { launch_lines<M, 12, NP, 3>(ba, (int)half); }, reps);
                printf("   whole lines, 12 waves: without epilogue arithmetic %.4f ms, without MFMAs %.4f, loads and stores (and the LDS round trip) only %.4f\n", e1, e2, e3);
            }
#endif
            printf("   inputs as whole lines through a wave-private LDS slot: %zu words differ; 8 waves %.4f ms, 12 waves %.4f, 16 waves %.4f\n", dd, l8, l12, l16);
            bad_lines += dd != 0;
        }
    }
    const double bytes = (double)G * (M + n) * 32;
    printf("   algorithmic bytes %.1f MB: at 8 TB/s %.4f ms\n", bytes / 1e6, bytes / 8e12 * 1e3);
    CK(hipFree(d_tp)); CK(hipFree(d_tb)); CK(hipFree(d_x)); CK(hipFree(d_y0)); CK(hipFree(d_y1));
    return errors + (diff != 0);
}

int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 20, reps = argc > 2 ? atoi(argv[2]) : 20;
    if (argc > 3) g_nwg = atoi(argv[3]);
    const size_t G = (size_t)1 << lg;
    int bad = 0;
    if (argc > 4) {  // config 4's shape: 16 parties x 381 300 chunks of 11 (4 194 300 triples per party)
        bad += run_triple<11, 8>(16, 1000 + 13, 3, 2);
        bad += run_triple<11, 8>(16, 381300, atoi(argv[4]), reps);
        bad += run_triple<6, 8>(16, 381300, atoi(argv[4]), reps);
        printf(bad ? "FAILED\n" : "ok\n");
        return bad != 0;
    }
    bad += run<6, 16, 0, 8>("config 2 (n = 16, t = 5)", 16, G, reps);
    bad += run<11, 12, 11, 8>("config 3 (n = 31, t = 10)", 31, G, reps);
    bad += run<3, 16, 0, 4>("n = 7, t = 2", 7, G, reps);
    bad += bad_lines;
    printf(bad ? "FAILED\n" : "all outputs identical\n");
    return bad != 0;
}
