// ubench_mfma_rt.hip -- microbenchmark + check of csrc/kernels_mfma_rt.hpp (table rows in registers, inputs by LDS-DMA,
// MFMAs interleaved with the previous row's epilogue) next to csrc/kernels_mfma.hpp on the same buffers.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_mfma_rt.hip -o tools/ubench_mfma_rt
//   tools/ubench_mfma_rt [log2_chunks=20] [reps=20] [shape mask] [ring slots=0 (as many as fit)] [workgroups=256]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <random>
#include <vector>

#define HBMPC_RT_PROF 1
#ifndef RT_W
#define RT_W 4
#endif
#ifndef RT_OCC
#define RT_OCC 1
#endif
#include "kernels_mfma_rt.hpp"
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

static int g_nwg = 256, g_nslot = 0, g_lds_cap = 160 * 1024;

// all rows in one role (what the register file allows: 4 RPW rows)
template <int M, int RPW, int ABL = 0, int W = RT_W, int OCC = RT_OCC>
static bool launch_rt(mf::RtArgs a, int rows) {
    if (!mf::mf_plan_roles(rows, a.nv, W * RPW, g_nwg * OCC, &a)) return false;
    const int need = mf::mf_rt_plan(&a, W);
    if (need > RPW) {
        fprintf(stderr, "plan needs %d rows per wave, kernel has %d\n", need, RPW);
        return false;
    }
    int nver_max = 0, rows_max = 0;
    for (int k = 0; k < a.nroles; ++k) {
        const int nver = a.role[k].row0 < a.nv ? std::min(a.nv - a.role[k].row0, a.role[k].nrows) : 0;
        nver_max = std::max(nver_max, nver), rows_max = std::max(rows_max, a.role[k].nrows);
    }
    int nslot = g_nslot ? g_nslot : mf::RT_MAX_SLOTS;
    while (nslot > 2 && mf::mf_rt_lds_bytes(M, nver_max, rows_max, nslot) > (size_t)g_lds_cap / OCC) --nslot;
    const size_t shm = mf::mf_rt_lds_bytes(M, nver_max, rows_max, nslot);
    static bool attr_set = false;
    if (!attr_set) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mf::k_mfma_rt<M, RPW, W, ABL, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
        fprintf(stderr, "   rt<%d,%d> W=%d: rows %d nv %d roles %d wv %d ring slots %d lds %zu\n", M, RPW, W, rows, a.nv, a.nroles, a.role_wv[0], nslot, shm);
    }
    hipLaunchKernelGGL((mf::k_mfma_rt<M, RPW, W, ABL, OCC>), dim3((unsigned)mf::mf_grid(a)), dim3(64 * W), shm, 0, a, nslot);
    return true;
}
template <int M, int W, int NR>
static void launch_lds(mf::MfmaRowsArgs a, int rows) {
    constexpr int ROWB = M * 1024 + 128;
    if (!mf::mf_plan_roles(rows, a.nv, (160 * 1024) / ROWB, g_nwg, &a)) exit(2);
    const size_t shm = (size_t)mf::mf_max_role_rows(a) * ROWB;
    static bool attr_set = false;
    if (!attr_set) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mf::k_mfma_rows<M, 1, W, NR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL((mf::k_mfma_rows<M, 1, W, NR>), dim3((unsigned)mf::mf_grid(a)), dim3(64 * W), shm, 0, a);
}

// encode x[G][M] with [I ; Cv] -> evals[M + nv][G]; decode with verify rows Cv and output rows Co; P(0)-only; all checked
// against host arithmetic on sampled chunks (every chunk of small batches)
template <int M, int RPW_DEC, int RPW_ENC, int W_LDS, int RPW_P0 = RPW_DEC>
static int run_shape(const char* name, int nv, size_t G, int reps, bool time_lds) {
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
    uint8_t *d_tenc, *d_tdec, *d_tp0, *d_x, *d_y, *d_y2, *d_out, *d_st;
    uint32_t *d_flag, *d_cnt, *d_sum, *d_nco;
    CK(hipMalloc(&d_tenc, tenc.size() * 4));
    CK(hipMalloc(&d_tdec, tdec.size() * 4));
    CK(hipMalloc(&d_tp0, tp0.size() * 4));
    CK(hipMemcpy(d_tenc, tenc.data(), tenc.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tdec, tdec.data(), tdec.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tp0, tp0.data(), tp0.size() * 4, hipMemcpyHostToDevice));
    std::vector<uint64_t> x(G * M * 4);
    for (size_t i = 0; i < G * M; ++i) rand_canon(&x[4 * i]);
    {
        const uint64_t rm1[4] = {HFr::MOD[0] - 1, HFr::MOD[1], HFr::MOD[2], HFr::MOD[3]};
        for (int i = 0; i < M && G > 4; ++i) {
            for (int k = 0; k < 4; ++k) x[(0 * M + i) * 4 + k] = 0;
            for (int k = 0; k < 4; ++k) x[(1 * M + i) * 4 + k] = rm1[k];
            for (int k = 0; k < 4; ++k) x[(2 * M + i) * 4 + k] = k == 0 ? 1 : 0;
            for (int k = 0; k < 4; ++k) x[(3 * M + i) * 4 + k] = (i & 1) ? rm1[k] : 0;
        }
    }
    CK(hipMalloc(&d_x, G * M * 32));
    CK(hipMalloc(&d_y, (size_t)n * G * 32));
    CK(hipMalloc(&d_y2, (size_t)n * G * 32));
    CK(hipMalloc(&d_out, G * M * 32));
    CK(hipMalloc(&d_st, G));
    CK(hipMalloc(&d_flag, G * 4));
    CK(hipMalloc(&d_nco, G * 4));
    CK(hipMalloc(&d_cnt, 128));
    CK(hipMalloc(&d_sum, 16));
    CK(hipMemcpy(d_x, x.data(), G * M * 32, hipMemcpyHostToDevice));
    CK(hipMemset(d_cnt, 0, 128));
    CK(hipMemset(d_y, 0xEE, (size_t)n * G * 32));
    mf::RtArgs ea = {};
    ea.in = d_x, ea.G = G, ea.in_chunk_major = 1, ea.table = d_tenc, ea.nv = 0, ea.out = d_y, ea.out_party_major = 1, ea.out_stride = G;
    int errors = 0;
    const bool enc_rt = launch_rt<M, RPW_ENC>(ea, n);
    if (!enc_rt) launch_lds<M, W_LDS, 0>(ea, n);
    CK(hipDeviceSynchronize());
    std::vector<size_t> samp;
    if (G <= 4096) {
        for (size_t gidx = 0; gidx < G; ++gidx) samp.push_back(gidx);
    } else {
        for (size_t gidx = 0; gidx < 64; ++gidx) samp.push_back(gidx);
        for (int k = 0; k < 512; ++k) samp.push_back(rng() % G);
        for (size_t gidx = G - 64; gidx < G; ++gidx) samp.push_back(gidx);
    }
    std::vector<uint64_t> yall((size_t)n * G * 4);
    CK(hipMemcpy(yall.data(), d_y, (size_t)n * G * 32, hipMemcpyDeviceToHost));
    for (size_t gi : samp) {
        HFr xv[M];
        for (int i = 0; i < M; ++i) xv[i] = HFr::from_canon(&x[(gi * M + i) * 4]);
        for (int s = 0; s < n; ++s) {
            HFr acc = HFr::zero();
            for (int i = 0; i < M; ++i) acc = acc + Cenc[s][i] * xv[i];
            uint64_t want[4];
            acc.to_canon(want);
            if (memcmp(want, &yall[((size_t)s * G + gi) * 4], 32) != 0 && errors++ < 5)
                fprintf(stderr, "%s: encode mismatch chunk %zu row %d: got %016llx.. want %016llx..\n", name, gi, s,
                        (unsigned long long)yall[((size_t)s * G + gi) * 4], (unsigned long long)want[0]);
        }
    }
    if (errors) fprintf(stderr, "%s: encode (%s kernel): %d mismatches\n", name, enc_rt ? "rt" : "lds", errors);
    // corrupt two chunks: one in a verify row, one in an interpolation row
    const size_t bad1 = G / 3, bad2 = G / 2 + 1;
    auto corrupt = [&](bool on) {
        (void)on;
        uint64_t v[4];
        CK(hipMemcpy(v, d_y + ((size_t)(M + 1) * G + bad1) * 32, 32, hipMemcpyDeviceToHost));
        v[0] ^= 1;
        CK(hipMemcpy(d_y + ((size_t)(M + 1) * G + bad1) * 32, v, 32, hipMemcpyHostToDevice));
        CK(hipMemcpy(v, d_y + ((size_t)2 * G + bad2) * 32, 32, hipMemcpyDeviceToHost));
        v[3] ^= 1ull << 40;
        CK(hipMemcpy(d_y + ((size_t)2 * G + bad2) * 32, v, 32, hipMemcpyHostToDevice));
    };
    corrupt(true);
    mf::RtArgs ra = {};
    ra.in = d_y, ra.G = G, ra.in_chunk_major = 0, ra.row_stride = G, ra.table = d_tdec, ra.nv = nv, ra.out = d_out, ra.out_party_major = 0, ra.out_stride = M;
    ra.status = d_st, ra.flagged = d_flag, ra.counters = d_cnt, ra.summary = d_sum, ra.ncoeffs = d_nco;
    for (int s = 0; s < n; ++s) ra.rows.set(s, s);
    for (int direct = 0; direct < 2; ++direct) {
        ra.direct = direct;
        CK(hipMemset(d_out, 0xEE, G * M * 32));
        CK(hipMemset(d_st, 0x77, G));
        CK(hipMemset(d_cnt, 0, 128));
        if (!launch_rt<M, RPW_DEC>(ra, nv + M)) {
            fprintf(stderr, "%s: decode does not fit the rt kernel\n", name);
            return errors + 1;
        }
        CK(hipDeviceSynchronize());
        uint32_t cnt[4], sum[4];
        CK(hipMemcpy(cnt, d_cnt, 16, hipMemcpyDeviceToHost));
        CK(hipMemcpy(sum, d_sum, 16, hipMemcpyDeviceToHost));
        std::vector<uint8_t> st(G);
        CK(hipMemcpy(st.data(), d_st, G, hipMemcpyDeviceToHost));
        size_t nbad = 0;
        for (size_t gi = 0; gi < G; ++gi) nbad += st[gi] != 0;
        const uint8_t want_st = direct ? (uint8_t)DecodingError : 0xff;
        const bool cnt_ok = direct ? (cnt[0] == 0 && sum[0] == 2 && sum[2] == (uint32_t)bad1) : cnt[0] == 2;
        if (!cnt_ok || nbad != 2 || st[bad1] != want_st || st[bad2] != want_st) {
            fprintf(stderr, "%s: direct=%d flagged %u summary {%u %u %u %u}, status!=0 on %zu chunks (expected exactly the 2 corrupted), st %02x %02x\n", name,
                    direct, cnt[0], sum[0], sum[1], sum[2], sum[3], nbad, st[bad1], st[bad2]);
            ++errors;
        }
        std::vector<uint64_t> oall(G * M * 4);
        CK(hipMemcpy(oall.data(), d_out, G * M * 32, hipMemcpyDeviceToHost));
        int derr = 0;
        for (size_t gi : samp) {
            const bool isbad = gi == bad1 || gi == bad2;
            if (isbad && !direct) continue;
            HFr xv[M];
            for (int i = 0; i < M; ++i) xv[i] = HFr::from_canon(&x[(gi * M + i) * 4]);
            for (int k = 0; k < M; ++k) {
                HFr acc = HFr::zero();
                for (int i = 0; i < M; ++i) acc = acc + Co[k][i] * xv[i];
                uint64_t want[4] = {0, 0, 0, 0};
                if (!isbad) acc.to_canon(want);
                if (memcmp(want, &oall[(gi * M + k) * 4], 32) != 0 && derr++ < 5)
                    fprintf(stderr, "%s: decode (direct=%d) mismatch chunk %zu coeff %d got %016llx want %016llx\n", name, direct, gi, k,
                            (unsigned long long)oall[(gi * M + k) * 4], (unsigned long long)want[0]);
            }
        }
        if (direct) {  // the zeroed chunks
            for (size_t gi : {bad1, bad2})
                for (int k = 0; k < M * 4; ++k)
                    if (oall[gi * M * 4 + k] != 0 && derr++ < 5) fprintf(stderr, "%s: failed chunk %zu not zeroed\n", name, gi);
        }
        errors += derr;
    }
    ra.direct = 0;
    // P(0) only
    mf::RtArgs rp = ra;
    rp.table = d_tp0, rp.out_stride = 1;
    CK(hipMemset(d_out, 0xEE, G * M * 32));
    CK(hipMemset(d_cnt, 0, 128));
    if (launch_rt<M, RPW_P0>(rp, nv + 1)) {
        CK(hipDeviceSynchronize());
        std::vector<uint64_t> oall(G * 4);
        CK(hipMemcpy(oall.data(), d_out, G * 32, hipMemcpyDeviceToHost));
        int perr = 0;
        for (size_t gi : samp) {
            if (gi == bad1 || gi == bad2) continue;
            HFr acc = HFr::zero();
            for (int i = 0; i < M; ++i) acc = acc + Co[0][i] * HFr::from_canon(&x[(gi * M + i) * 4]);
            uint64_t want[4];
            acc.to_canon(want);
            if (memcmp(want, &oall[gi * 4], 32) != 0 && perr++ < 5) fprintf(stderr, "%s: P(0) mismatch chunk %zu\n", name, gi);
        }
        errors += perr;
    }
    // restore, then time on clean inputs
    corrupt(false);
    CK(hipMemset(d_cnt, 0, 128));
    CK(hipDeviceSynchronize());
    if (reps > 0) {
        long long* d_prof;
        CK(hipMalloc(&d_prof, 8 * 8 * 8 * 8));
        for (int which = 0; which < 3; ++which) {
            mf::RtArgs pa = which == 0 ? ra : which == 1 ? rp : ea;
            pa.prof = d_prof;
            CK(hipMemset(d_prof, 0, 8 * 8 * 8 * 8));
            if (which == 0) launch_rt<M, RPW_DEC>(pa, nv + M);
            else if (which == 1) launch_rt<M, RPW_P0>(pa, nv + 1);
            else if (enc_rt) launch_rt<M, RPW_ENC>(pa, n);
            CK(hipDeviceSynchronize());
            long long pf[8 * 8 * 8];
            CK(hipMemcpy(pf, d_prof, sizeof pf, hipMemcpyDeviceToHost));
            for (int w = 0; w < 4; ++w) {
                const long long* r = pf + (0 * 8 + w * (RT_W / 4)) * 8;  // (with 8 waves: the first four)
                const double nq = r[6] ? (double)r[6] : 1;
                fprintf(stderr, "   prof %s wg0 wave %d (%lld tiles): cycles per tile: barrier %.0f | b-read+setup %.0f | report %.0f | rows %.0f\n",
                        which == 0 ? "decode" : which == 1 ? "p0" : "encode", w, r[6], r[0] / nq, r[1] / nq, r[2] / nq, r[3] / nq);
            }
        }
        (void)hipFree(d_prof);
#ifdef RT_ABLATE
#define ABL_RUN(A) fprintf(stderr, "   abl %2d: decode %.4f ms\n", A, time_ms([&] { launch_rt<M, RPW_DEC, A>(ra, nv + M); }, reps))
        ABL_RUN(6); ABL_RUN(7); ABL_RUN(6 + 64); ABL_RUN(6 + 128); ABL_RUN(6 + 256); ABL_RUN(6 + 512); ABL_RUN(6 + 1024); ABL_RUN(6 + 2048); ABL_RUN(6 + 64 + 128 + 256); ABL_RUN(6 + 512 + 1024 + 2048);
        CK(hipMemset(d_cnt, 0, 128));
#endif
        const float ms_dec = time_ms([&] { launch_rt<M, RPW_DEC>(ra, nv + M); }, reps);
        const float ms_p0 = time_ms([&] { launch_rt<M, RPW_P0>(rp, nv + 1); }, reps);
        float ms_enc = 0;
        if (enc_rt) ms_enc = time_ms([&] { launch_rt<M, RPW_ENC>(ea, n); }, reps);
        float l_dec = 0, l_p0 = 0, l_enc = 0;
        if (time_lds) {
            l_dec = time_ms([&] { launch_lds<M, W_LDS, (M == 11 ? 11 : 0)>(ra, nv + M); }, reps);
            l_p0 = time_ms([&] { launch_lds<M, W_LDS, 0>(rp, nv + 1); }, reps);
            l_enc = time_ms([&] { launch_lds<M, W_LDS, 0>(ea, n); }, reps);
        }
        uint32_t cnt[4];
        CK(hipMemcpy(cnt, d_cnt, 16, hipMemcpyDeviceToHost));
        if (cnt[0] != 0) {
            fprintf(stderr, "%s: %u chunks flagged in the clean timed runs\n", name, cnt[0]);
            ++errors;
        }
        const double enc_b = (double)(M + n) * 32 * G, dec_b = (double)(n + M) * 32 * G, p0_b = (double)(n + 1) * 32 * G;
        printf("{\"shape\": \"%s\", \"M\": %d, \"rows_verify\": %d, \"chunks\": %zu, \"errors\": %d, \"rt\": {\"decode_ms\": %.4f, \"decode_GBps\": %.0f, "
               "\"p0_ms\": %.4f, \"p0_GBps\": %.0f, \"encode_%d_rows_ms\": %.4f, \"encode_GBps\": %.0f}, \"lds\": {\"decode_ms\": %.4f, \"p0_ms\": %.4f, "
               "\"encode_ms\": %.4f}}\n",
               name, M, nv, G, errors, ms_dec, dec_b / ms_dec / 1e6, ms_p0, p0_b / ms_p0 / 1e6, n, ms_enc, ms_enc > 0 ? enc_b / ms_enc / 1e6 : 0.0, l_dec, l_p0,
               l_enc);
        fflush(stdout);
    } else {
        printf("{\"shape\": \"%s\", \"M\": %d, \"rows_verify\": %d, \"chunks\": %zu, \"errors\": %d}\n", name, M, nv, G, errors);
    }
    for (void* q : {(void*)d_tenc, (void*)d_tdec, (void*)d_tp0, (void*)d_x, (void*)d_y, (void*)d_y2, (void*)d_out, (void*)d_st, (void*)d_flag, (void*)d_cnt,
                    (void*)d_sum, (void*)d_nco})
        (void)hipFree(q);
    return errors;
}

// does an LDS-DMA destination beyond 64 KB work (M0 carries the LDS address)?
__global__ void k_probe_m0(const uint8_t* src, uint32_t* out, uint32_t off) {
    extern __shared__ __attribute__((aligned(16))) uint8_t l[];
    for (uint32_t i = threadIdx.x; i < (off + 1024) / 4; i += 64) reinterpret_cast<uint32_t*>(l)[i] = 0xdeadbeefu;
    __syncthreads();
    mf::rt_dma16(src, threadIdx.x * 16u, (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(uintptr_t)l + off)));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const uint32_t* p = reinterpret_cast<const uint32_t*>(l + off) + threadIdx.x * 4;
    out[threadIdx.x] = p[0] ^ p[1] ^ p[2] ^ p[3];
    if (threadIdx.x == 0) out[64] = reinterpret_cast<const uint32_t*>(l)[(off & 0xffffu) / 4];  // where a 16-bit M0 would have written
}
static bool probe_m0(uint32_t off) {
    uint8_t* d_src;
    uint32_t* d_out;
    CK(hipMalloc(&d_src, 1024));
    CK(hipMalloc(&d_out, 65 * 4));
    std::vector<uint32_t> src(256);
    for (int i = 0; i < 256; ++i) src[i] = 0x1000u + i * 77u;
    CK(hipMemcpy(d_src, src.data(), 1024, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_probe_m0), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(k_probe_m0, dim3(1), dim3(64), off + 1024, 0, d_src, d_out, off);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> out(65);
    CK(hipMemcpy(out.data(), d_out, 65 * 4, hipMemcpyDeviceToHost));
    bool ok = true;
    for (int t = 0; t < 64; ++t) ok &= out[t] == (src[4 * t] ^ src[4 * t + 1] ^ src[4 * t + 2] ^ src[4 * t + 3]);
    fprintf(stderr, "LDS-DMA to LDS offset %u: %s (word at the 16-bit alias: %08x)\n", off, ok ? "ok" : "WRONG", out[64]);
    (void)hipFree(d_src);
    (void)hipFree(d_out);
    return ok;
}

int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 20;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const int mask = argc > 3 ? atoi(argv[3]) : 0xff;
    if (argc > 4) g_nslot = atoi(argv[4]);
    if (argc > 5) g_nwg = atoi(argv[5]);
    const size_t G = (size_t)1 << lg;
    int errors = 0;
    probe_m0(1024);
    if (!probe_m0(100 * 1024)) {
        g_lds_cap = 64 * 1024;
        fprintf(stderr, "ring limited to the first 64 KB of LDS\n");
    }
#if RT_OCC == 2
    if (mask & 1) errors += run_shape<6, 5, 4, 16, 3>("cfg2 ragged", 10, 3000 + 5, 0, false);
    if (mask & 2) errors += run_shape<6, 5, 4, 16, 3>("cfg2 n=16 d=5: 16-row encode / t=10 decode", 10, G, reps, true);
    fprintf(stderr, errors ? "FAILED: %d errors\n" : "all checks passed\n", errors);
    return errors != 0;
#elif RT_W == 8
    if (mask & 1) errors += run_shape<11, 3, 3, 12, 2>("cfg3 ragged", 10, 1000 + 37, 0, false);
    if (mask & 2) errors += run_shape<11, 3, 3, 12, 2>("cfg3 n=31 t=10", 10, G, reps, true);
    if (mask & 4) errors += run_shape<11, 3, 2, 12, 1>("cfg4 decode n=16 d=10 t=5", 5, G, reps, true);
    if (mask & 8) errors += run_shape<6, 2, 2, 16, 1>("cfg2 n=16 d=5 (encode) / t=5 decode", 5, G, reps, true);
    if (mask & 16) errors += run_shape<6, 2, 2, 16, 2>("cfg2 n=16 d=5: 16-row encode / t=10 decode", 10, G, reps, true);
    fprintf(stderr, errors ? "FAILED: %d errors\n" : "all checks passed\n", errors);
    return errors != 0;
#else
    if (mask & 1) errors += run_shape<11, 6, 6, 12>("cfg3 ragged", 10, 1000 + 37, 0, false);
    if (mask & 1) errors += run_shape<11, 6, 6, 12>("cfg3 one tile", 10, 29, 0, false);
    if (mask & 1) errors += run_shape<6, 4, 4, 16>("m=6 ragged", 5, 3000 + 5, 0, false);
    if (mask & 2) errors += run_shape<11, 6, 6, 12, 4>("cfg3 n=31 t=10", 10, G, reps, true);
    if (mask & 4) errors += run_shape<11, 5, 4, 12, 2>("cfg4 decode n=16 d=10 t=5", 5, G, reps, true);
    if (mask & 8) errors += run_shape<6, 3, 3, 16, 2>("cfg2 n=16 d=5 (encode) / t=5 decode", 5, G, reps, true);
    if (mask & 16) errors += run_shape<6, 5, 4, 16, 3>("cfg2 n=16 d=5: 16-row encode / t=10 decode", 10, G, reps, true);
    fprintf(stderr, errors ? "FAILED: %d errors\n" : "all checks passed\n", errors);
    return errors != 0;
#endif
}
