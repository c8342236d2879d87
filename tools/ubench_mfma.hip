// ubench_mfma.hip -- go / no-go microbenchmark for the matrix-core formulation of the constant-matrix maps
// (VERDICT r1 item 2): runs csrc/kernels_mfma.hpp at the BASELINE shapes on random data, checks sampled chunks against
// host field arithmetic (HFr), and prints ms per launch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_mfma.hip -o tools/ubench_mfma
//   tools/ubench_mfma [log2_chunks=20] [reps=20] [shape mask=127] [workgroups=256]
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

static std::mt19937_64 rng(0xC0FFEE02);
static void rand_canon(uint64_t c[4]) {
    for (;;) {
        for (int i = 0; i < 4; ++i) c[i] = rng();
        c[3] &= 0x7fffffffffffffffULL;
        if (!HFr::geq(c)) return;
    }
}
static HFr rand_fr() {
    uint64_t c[4];
    rand_canon(c);
    return HFr::from_canon(c);
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

static int g_nwg = 256;  // workgroups per launch (one per CU: the table rows of a role fill most of the LDS)
template <int M, int CG, int WAVES, int NR = 0, int ABL = 0, bool PIPE = false>
static void launch_rows(mf::MfmaRowsArgs a, int rows) {
    constexpr int ROWB = M * 1024 + 128;
    const int cap = (160 * 1024) / ROWB;
    if (!mf::mf_plan_roles(rows, a.nv, cap, g_nwg, &a)) {
        fprintf(stderr, "rows do not fit the role plan\n");
        exit(2);
    }
    const size_t shm = (size_t)mf::mf_max_role_rows(a) * ROWB;
    static int shown = 0;
    if (shown < 12 && a.G > 100000) {
        ++shown;
        fprintf(stderr, "   plan rows=%d nv=%d:", rows, a.nv);
        for (int k = 0; k < a.nroles; ++k) fprintf(stderr, " role %d = rows [%d, %d) x %d workgroups;", k, a.role[k].row0, a.role[k].row0 + a.role[k].nrows, a.role_nwg[k]);
        fprintf(stderr, " blocks:");
        for (int j = 0; j < a.nblocks; ++j) fprintf(stderr, "%d", a.blk_role[j]);
        fprintf(stderr, "\n");
    }
    static bool attr_set = false;
    if (!attr_set) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mf::k_mfma_rows_lab<M, CG, WAVES, NR, ABL, PIPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    if (NR > 0 && mf::mf_max_role_rows(a) > NR) {
        fprintf(stderr, "role rows %d exceed the static row count %d\n", mf::mf_max_role_rows(a), NR);
        exit(2);
    }
    hipLaunchKernelGGL((mf::k_mfma_rows_lab<M, CG, WAVES, NR, ABL, PIPE>), dim3((unsigned)mf::mf_grid(a)), dim3(64 * WAVES), shm, 0, a);
}

// encode x[G][M] with [I ; Cv] -> evals[M + nv][G]; decode with verify rows Cv and output rows Co; check vs host
template <int M, int CG, int WAVES>
static int run_shape(const char* name, int nv, size_t G, int reps) {
    std::vector<std::vector<HFr>> Cv(nv, std::vector<HFr>(M)), Co(M, std::vector<HFr>(M)), Cenc;
    for (auto& row : Cv)
        for (auto& v : row) v = rand_fr();
    for (auto& row : Co)
        for (auto& v : row) v = rand_fr();
    for (int i = 0; i < M; ++i) {
        std::vector<HFr> row(M, HFr::zero());
        row[i] = HFr::one();
        Cenc.push_back(row);
    }
    for (auto& row : Cv) Cenc.push_back(row);
    std::vector<std::vector<HFr>> Cdec = Cv;
    for (auto& row : Co) Cdec.push_back(row);
    std::vector<std::vector<HFr>> Cp0 = Cv;
    Cp0.push_back(Co[0]);
    const int n = M + nv;
    auto tenc = build_mfma_table(Cenc, M), tdec = build_mfma_table(Cdec, M), tp0 = build_mfma_table(Cp0, M);
    uint8_t *d_tenc, *d_tdec, *d_tp0, *d_x, *d_y, *d_out, *d_st;
    uint32_t *d_flag, *d_cnt, *d_sum;
    CK(hipMalloc(&d_tenc, tenc.size() * 4));
    CK(hipMalloc(&d_tdec, tdec.size() * 4));
    CK(hipMalloc(&d_tp0, tp0.size() * 4));
    CK(hipMemcpy(d_tenc, tenc.data(), tenc.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tdec, tdec.data(), tdec.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tp0, tp0.data(), tp0.size() * 4, hipMemcpyHostToDevice));
    std::vector<uint64_t> x(G * M * 4);
    for (size_t i = 0; i < G * M; ++i) rand_canon(&x[4 * i]);
    // edge values in the first chunks: 0, r - 1, 1, 2^255-ish patterns
    {
        const uint64_t rm1[4] = {HFr::MOD[0] - 1, HFr::MOD[1], HFr::MOD[2], HFr::MOD[3]};
        for (int i = 0; i < M; ++i) {
            for (int k = 0; k < 4; ++k) x[(0 * M + i) * 4 + k] = 0;
            for (int k = 0; k < 4; ++k) x[(1 * M + i) * 4 + k] = rm1[k];
            for (int k = 0; k < 4; ++k) x[(2 * M + i) * 4 + k] = k == 0 ? 1 : 0;
            for (int k = 0; k < 4; ++k) x[(3 * M + i) * 4 + k] = (i & 1) ? rm1[k] : 0;
        }
    }
    CK(hipMalloc(&d_x, G * M * 32));
    CK(hipMalloc(&d_y, (size_t)n * G * 32));
    CK(hipMalloc(&d_out, G * M * 32));
    CK(hipMalloc(&d_st, G));
    CK(hipMalloc(&d_flag, G * 4));
    CK(hipMalloc(&d_cnt, 16));
    CK(hipMalloc(&d_sum, 16));
    CK(hipMemcpy(d_x, x.data(), G * M * 32, hipMemcpyHostToDevice));
    CK(hipMemset(d_cnt, 0, 16));
    mf::MfmaRowsArgs ea = {};
    ea.in = d_x, ea.G = G, ea.in_chunk_major = 1, ea.table = d_tenc, ea.nv = 0, ea.out = d_y, ea.out_party_major = 1, ea.out_stride = G;
    launch_rows<M, CG, WAVES>(ea, n);
    CK(hipDeviceSynchronize());
    int errors = 0;
    // sample chunks
    std::vector<size_t> samp;
    for (size_t gidx = 0; gidx < 64 && gidx < G; ++gidx) samp.push_back(gidx);
    for (int k = 0; k < 512; ++k) samp.push_back(rng() % G);
    for (size_t gidx = G > 64 ? G - 64 : 0; gidx < G; ++gidx) samp.push_back(gidx);
    std::vector<uint64_t> yrow(4);
    for (size_t gi : samp) {
        HFr xv[M];
        for (int i = 0; i < M; ++i) xv[i] = HFr::from_canon(&x[(gi * M + i) * 4]);
        for (int s = 0; s < n; ++s) {
            HFr acc = HFr::zero();
            for (int i = 0; i < M; ++i) acc = acc + Cenc[s][i] * xv[i];
            uint64_t want[4];
            acc.to_canon(want);
            CK(hipMemcpy(yrow.data(), d_y + ((size_t)s * G + gi) * 32, 32, hipMemcpyDeviceToHost));
            if (memcmp(want, yrow.data(), 32) != 0 && errors++ < 5)
                fprintf(stderr, "%s: encode mismatch chunk %zu row %d: got %016llx.. want %016llx..\n", name, gi, s,
                        (unsigned long long)yrow[0], (unsigned long long)want[0]);
        }
    }
    if (nv > 14) {
        // encode only (e.g. 31 rows = config 3's apply_vandermonde): more verify rows than one decode role holds
        const float ms_e = time_ms([&] { launch_rows<M, CG, WAVES>(ea, n); }, reps);
        const float ms_s = time_ms([&] { launch_rows<M, CG, WAVES, 11>(ea, n); }, reps);
        printf("{\"shape\": \"%s\", \"M\": %d, \"CG\": %d, \"waves\": %d, \"chunks\": %zu, \"errors\": %d, \"encode_%d_rows_ms\": %.4f, "
               "\"encode_%d_rows_static_row_count_ms\": %.4f, \"encode_GBps\": %.0f}\n", name, M, CG, WAVES, G, errors, n, ms_e, n, ms_s,
               (double)(M + n) * 32 * G / (ms_s < ms_e ? ms_s : ms_e) / 1e6);
        fflush(stdout);
        for (void* q : {(void*)d_tenc, (void*)d_tdec, (void*)d_tp0, (void*)d_x, (void*)d_y, (void*)d_out, (void*)d_st, (void*)d_flag, (void*)d_cnt, (void*)d_sum}) (void)hipFree(q);
        return errors;
    }
    // corrupt two chunks: one in a verify row, one in an interpolation row
    const size_t bad1 = G / 3, bad2 = G / 2 + 1;
    {
        uint64_t v[4];
        CK(hipMemcpy(v, d_y + ((size_t)(M + 1) * G + bad1) * 32, 32, hipMemcpyDeviceToHost));
        v[0] ^= 1;
        CK(hipMemcpy(d_y + ((size_t)(M + 1) * G + bad1) * 32, v, 32, hipMemcpyHostToDevice));
        CK(hipMemcpy(v, d_y + ((size_t)2 * G + bad2) * 32, 32, hipMemcpyDeviceToHost));
        v[3] ^= 1ull << 40;
        CK(hipMemcpy(d_y + ((size_t)2 * G + bad2) * 32, v, 32, hipMemcpyHostToDevice));
    }
    mf::MfmaRowsArgs ra = {};
    ra.in = d_y, ra.G = G, ra.in_chunk_major = 0, ra.row_stride = G, ra.table = d_tdec, ra.nv = nv, ra.out = d_out, ra.out_party_major = 0, ra.out_stride = M;
    ra.status = d_st, ra.flagged = d_flag, ra.counters = d_cnt, ra.summary = d_sum;
    for (int s = 0; s < n; ++s) ra.rows.set(s, s);
    CK(hipMemset(d_out, 0xEE, G * M * 32));
    launch_rows<M, CG, WAVES>(ra, nv + M);
    CK(hipDeviceSynchronize());
    uint32_t cnt[4];
    CK(hipMemcpy(cnt, d_cnt, 16, hipMemcpyDeviceToHost));
    std::vector<uint8_t> st(G);
    CK(hipMemcpy(st.data(), d_st, G, hipMemcpyDeviceToHost));
    size_t nbad = 0;
    for (size_t gi = 0; gi < G; ++gi) nbad += st[gi] != 0;
    if (cnt[0] != 2 || nbad != 2 || st[bad1] != 0xff || st[bad2] != 0xff) {
        fprintf(stderr, "%s: flagged %u, status!=0 on %zu chunks (expected exactly the 2 corrupted)\n", name, cnt[0], nbad);
        ++errors;
    }
    std::vector<uint64_t> orow(4 * M);
    for (size_t gi : samp) {
        if (gi == bad1 || gi == bad2) continue;
        HFr xv[M];
        for (int i = 0; i < M; ++i) xv[i] = HFr::from_canon(&x[(gi * M + i) * 4]);
        CK(hipMemcpy(orow.data(), d_out + gi * M * 32, M * 32, hipMemcpyDeviceToHost));
        for (int k = 0; k < M; ++k) {
            HFr acc = HFr::zero();
            for (int i = 0; i < M; ++i) acc = acc + Co[k][i] * xv[i];
            uint64_t want[4];
            acc.to_canon(want);
            if (memcmp(want, &orow[4 * k], 32) != 0 && errors++ < 5)
                fprintf(stderr, "%s: decode mismatch chunk %zu coeff %d\n", name, gi, k);
        }
    }
    // restore the corrupted values so the timed decode takes the optimistic path everywhere
    launch_rows<M, CG, WAVES>(ea, n);
    CK(hipMemset(d_cnt, 0, 16));
    CK(hipDeviceSynchronize());
    const float ms_enc = time_ms([&] { launch_rows<M, CG, WAVES>(ea, n); }, reps);
    const float ms_dec = time_ms([&] { launch_rows<M, CG, WAVES>(ra, nv + M); }, reps);
    // the same launches with a compile-time row count (16 covers every role of these shapes)
    const float ms_enc_s = time_ms([&] { launch_rows<M, CG, WAVES, (M == 11 ? 11 : 16)>(ea, n); }, reps);
    const float ms_dec_s = time_ms([&] { launch_rows<M, CG, WAVES, (M == 11 ? 11 : 16)>(ra, nv + M); }, reps);
    fprintf(stderr, "   static row count: encode %.4f -> %.4f ms, decode %.4f -> %.4f ms\n", ms_enc, ms_enc_s, ms_dec, ms_dec_s);
    if constexpr (M == 6 && CG == 1) {
        // config 2's shape (16 rows x 6 inputs, one role): the pipelined row loop at 8 and 12 waves per workgroup
        uint8_t* d_y2;
        CK(hipMalloc(&d_y2, (size_t)n * G * 32));
        CK(hipMemset(d_y2, 0xEE, (size_t)n * G * 32));
        mf::MfmaRowsArgs ep2 = ea;
        ep2.out = d_y2;
        launch_rows<M, 1, 8, 16, 0, true>(ep2, n);
        CK(hipDeviceSynchronize());
        std::vector<uint8_t> y1((size_t)n * G * 32), y2((size_t)n * G * 32);
        CK(hipMemcpy(y1.data(), d_y, y1.size(), hipMemcpyDeviceToHost));
        CK(hipMemcpy(y2.data(), d_y2, y2.size(), hipMemcpyDeviceToHost));
        if (memcmp(y1.data(), y2.data(), y1.size()) != 0) {
            size_t k = 0;
            while (y1[k] == y2[k]) ++k;
            fprintf(stderr, "%s: pipelined encode differs at byte %zu\n", name, k);
            ++errors;
        }
        CK(hipMemset(d_y2, 0xEE, (size_t)n * G * 32));
        launch_rows<M, 1, 12, 16, 0, true>(ep2, n);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(y2.data(), d_y2, y2.size(), hipMemcpyDeviceToHost));
        if (memcmp(y1.data(), y2.data(), y1.size()) != 0) {
            fprintf(stderr, "%s: pipelined encode (12 waves) differs\n", name);
            ++errors;
        }
        const float p8 = time_ms([&] { launch_rows<M, 1, 8, 16, 0, true>(ep2, n); }, reps);
        const float p12 = time_ms([&] { launch_rows<M, 1, 12, 16, 0, true>(ep2, n); }, reps);
        const float u8 = time_ms([&] { launch_rows<M, 1, 8, 16>(ep2, n); }, reps);
        const float u12 = time_ms([&] { launch_rows<M, 1, 12, 16>(ep2, n); }, reps);
        fprintf(stderr, "   %d-row encode, pipelined row loop: 8 waves %.4f ms, 12 waves %.4f ms (unpipelined, static rows: 8 waves %.4f, 12 waves %.4f, %d waves %.4f)\n", n, p8, p12, u8, u12, WAVES, ms_enc_s);
        printf("{\"shape\": \"%s\", \"encode_rows\": %d, \"pipelined_8_waves_ms\": %.4f, \"pipelined_12_waves_ms\": %.4f, \"unpipelined_8_waves_ms\": %.4f, \"unpipelined_12_waves_ms\": %.4f, \"unpipelined_%d_waves_ms\": %.4f, \"errors\": %d}\n", name, n, p8, p12, u8, u12, WAVES, ms_enc_s, errors);
        (void)hipFree(d_y2);
    }
    if constexpr (M == 11 && CG == 1) {
        // the pipelined row loop (matrix pipe and vector epilogue overlapped inside the wave), 8 waves per workgroup: same bytes?
        uint8_t *d_out2, *d_y2;
        CK(hipMalloc(&d_out2, G * M * 32));
        CK(hipMalloc(&d_y2, (size_t)n * G * 32));
        CK(hipMemset(d_out2, 0xEE, G * M * 32));
        CK(hipMemset(d_y2, 0xEE, (size_t)n * G * 32));
        mf::MfmaRowsArgs rp2 = ra, ep2 = ea;
        rp2.out = d_out2, ep2.out = d_y2;
        launch_rows<M, 1, 8, 11, 0, true>(rp2, nv + M);
        launch_rows<M, 1, 8, 11, 0, true>(ep2, n);
        CK(hipDeviceSynchronize());
        std::vector<uint8_t> h1(G * M * 32), h2(G * M * 32);
        CK(hipMemcpy(h1.data(), d_out, h1.size(), hipMemcpyDeviceToHost));
        CK(hipMemcpy(h2.data(), d_out2, h2.size(), hipMemcpyDeviceToHost));
        if (memcmp(h1.data(), h2.data(), h1.size()) != 0) {
            size_t k = 0;
            while (h1[k] == h2[k]) ++k;
            fprintf(stderr, "%s: pipelined decode differs at byte %zu (chunk %zu coeff %zu)\n", name, k, k / (M * 32), (k / 32) % M);
            ++errors;
        }
        std::vector<uint8_t> y1((size_t)n * G * 32), y2((size_t)n * G * 32);
        CK(hipMemcpy(y1.data(), d_y, y1.size(), hipMemcpyDeviceToHost));
        CK(hipMemcpy(y2.data(), d_y2, y2.size(), hipMemcpyDeviceToHost));
        if (memcmp(y1.data(), y2.data(), y1.size()) != 0) {
            fprintf(stderr, "%s: pipelined encode differs\n", name);
            ++errors;
        }
        uint32_t cnt2[4];
        CK(hipMemcpy(cnt2, d_cnt, 16, hipMemcpyDeviceToHost));
        const float ms_dec_p = time_ms([&] { launch_rows<M, 1, 8, 11, 0, true>(rp2, nv + M); }, reps);
        const float ms_enc_p = time_ms([&] { launch_rows<M, 1, 8, 11, 0, true>(ep2, n); }, reps);
        const float ms_dec_8 = time_ms([&] { launch_rows<M, 1, 8, 11>(rp2, nv + M); }, reps);
        const float ms_dec_p4 = time_ms([&] { launch_rows<M, 1, 4, 11, 0, true>(rp2, nv + M); }, reps);
        const float ms_enc_p4 = time_ms([&] { launch_rows<M, 1, 4, 11, 0, true>(ep2, n); }, reps);
        fprintf(stderr, "   pipelined row loop (4 waves, one per SIMD): decode %.4f ms, encode %.4f ms\n", ms_dec_p4, ms_enc_p4);
        fprintf(stderr, "   pipelined row loop (8 waves): decode %.4f ms (same 8 waves without it: %.4f), encode %.4f ms; flagged so far %u\n", ms_dec_p,
                ms_dec_8, ms_enc_p, cnt2[0]);
        printf("{\"shape\": \"%s\", \"pipelined\": {\"decode_ms\": %.4f, \"decode_8_waves_unpipelined_ms\": %.4f, \"encode_ms\": %.4f}, \"unpipelined_%d_waves\": {\"decode_ms\": %.4f, \"encode_ms\": %.4f}, \"errors\": %d}\n",
               name, ms_dec_p, ms_dec_8, ms_enc_p, WAVES, ms_dec_s, ms_enc_s, errors);
        CK(hipMemset(d_cnt, 0, 16));
        (void)hipFree(d_out2);
        (void)hipFree(d_y2);
    }
    mf::MfmaRowsArgs rp = ra;
    rp.table = d_tp0, rp.out_stride = 1;
    const float ms_p0 = time_ms([&] { launch_rows<M, CG, WAVES>(rp, nv + 1); }, reps);
    CK(hipMemcpy(cnt, d_cnt, 16, hipMemcpyDeviceToHost));
    if (G > 100000) {
        fprintf(stderr, "   ablation %s CG=%d waves=%d: full %.4f ms", name, CG, WAVES, ms_dec);
#define HBMPC_ABL_RUN(A, args, rows) fprintf(stderr, " | abl %d: %.4f", A, time_ms([&] { launch_rows<M, CG, WAVES, 0, A>(args, rows); }, reps))
        HBMPC_ABL_RUN(1, ra, nv + M); HBMPC_ABL_RUN(2, ra, nv + M); HBMPC_ABL_RUN(3, ra, nv + M); HBMPC_ABL_RUN(4, ra, nv + M);
        HBMPC_ABL_RUN(5, ra, nv + M); HBMPC_ABL_RUN(6, ra, nv + M); HBMPC_ABL_RUN(7, ra, nv + M);
        fprintf(stderr, "   (1 = no epilogue arithmetic, 2 = no MFMA, 3 = memory traffic only, 4 = inputs from L2 + no stores)\n");
        fprintf(stderr, "   encode ablation %s CG=%d waves=%d: full %.4f ms", name, CG, WAVES, ms_enc);
        HBMPC_ABL_RUN(1, ea, n); HBMPC_ABL_RUN(2, ea, n); HBMPC_ABL_RUN(3, ea, n); HBMPC_ABL_RUN(4, ea, n);
        HBMPC_ABL_RUN(5, ea, n); HBMPC_ABL_RUN(6, ea, n); HBMPC_ABL_RUN(7, ea, n);
        fprintf(stderr, "\n");
        CK(hipMemcpy(cnt, d_cnt, 16, hipMemcpyDeviceToHost));
    }
    if (cnt[0] != 0) {
        fprintf(stderr, "%s: %u chunks flagged in the clean timed runs\n", name, cnt[0]);
        ++errors;
    }
    const double enc_b = (double)(M + n) * 32 * G, dec_b = (double)(n + M) * 32 * G, p0_b = (double)(n + 1) * 32 * G;
    printf("{\"shape\": \"%s\", \"M\": %d, \"rows_verify\": %d, \"CG\": %d, \"waves\": %d, \"chunks\": %zu, \"errors\": %d, "
           "\"encode_%d_rows_ms\": %.4f, \"encode_GBps\": %.0f, \"decode_ms\": %.4f, \"decode_GBps\": %.0f, "
           "\"decode_p0_ms\": %.4f, \"decode_p0_GBps\": %.0f}\n",
           name, M, nv, CG, WAVES, G, errors, n, ms_enc, enc_b / ms_enc / 1e6, ms_dec, dec_b / ms_dec / 1e6, ms_p0, p0_b / ms_p0 / 1e6);
    fflush(stdout);
    for (void* q : {(void*)d_tenc, (void*)d_tdec, (void*)d_tp0, (void*)d_x, (void*)d_y, (void*)d_out, (void*)d_st, (void*)d_flag, (void*)d_cnt, (void*)d_sum}) (void)hipFree(q);
    return errors;
}

int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 20;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const size_t G = (size_t)1 << lg;
    int errors = 0;
    const int mask = argc > 3 ? atoi(argv[3]) : 0xff;
    // ragged small case first (live masks, partial tiles)
    if (argc > 4) g_nwg = atoi(argv[4]);
    if (mask & 1) errors += run_shape<11, 2, 8>("cfg3_ragged", 10, 1000 + 37, 2);
    if (mask & 1) errors += run_shape<11, 1, 16>("cfg3_ragged", 10, 3000 + 5, 2);
    if (mask & 2) errors += run_shape<11, 1, 8>("cfg3 n=31 t=10 (decode: 21 of the rows)", 10, G, reps);
    if (mask & 4) errors += run_shape<11, 1, 12>("cfg3 n=31 t=10 (decode: 21 of the rows)", 10, G, reps);
    if (mask & 8) errors += run_shape<11, 1, 16>("cfg3 n=31 t=10 (decode: 21 of the rows)", 10, G, reps);
    if (mask & 16) errors += run_shape<11, 2, 8>("cfg3 n=31 t=10 (decode: 21 of the rows)", 10, G, reps);
    if (mask & 128) errors += run_shape<11, 1, 12>("cfg3 encode: apply_vandermonde n=31 d=10 (31 rows)", 20, G, reps);
    if (mask & 32) errors += run_shape<6, 1, 16>("cfg2-like m=6, 10 extra rows", 10, G, reps);
    if (mask & 64) errors += run_shape<6, 2, 8>("cfg2-like m=6, 10 extra rows", 10, G, reps);
    fprintf(stderr, errors ? "FAILED: %d errors\n" : "all checks passed\n", errors);
    return errors != 0;
}
