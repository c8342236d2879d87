// TripleGenNode for all parties of a SMALL batch in one launch: a workgroup per chunk of 2t + 1 triples
// (triple_gen/triple_generation.rs:304-364 with BatchRecon's two arms, batch_recon.rs:157-165, :384-391, :457-467).
//
// The four launches of the separate steps -- the local products inside the encode, the P(0) decodes of all recipients, the
// coefficient decode of the revealed values, [c] = rt + opened -- cost 7 - 14 us each at a few hundred triples, each a lone wave per
// SIMD behind a memory round trip (profiles/r04_small_batch_fpmul.txt).  With all parties on one device a chunk of 2t + 1 triples
// depends on nothing outside it, so one workgroup of 256 lanes takes it from the shares to [c]: a lane per (party, triple) for the
// local products, a lane per (party, recipient) for the encode, a lane pair per (recipient, table row) and a lane quad per table row
// for the two decodes (dot_shared; over Goldilocks a lane per row: its products are a few instructions).  Covers n = 3t + 1 <= 16 (every recipient then decodes from exactly d + t + 1 = n senders: no
// OEC round, a failed chunk is final -- what the separate decodes do in one launch each).
//
// Every buffer a caller can see gets the bytes of the separate launches (tests/test_gpu_pipelines.py).
#pragma once
#include "kernels_recover.hpp"

namespace hbmpc {

struct TripleGenWgArgs {
    const uint32_t *a, *b, *r2t, *rt;  // [party][N]
    const uint32_t* vmat;              // [n][2t + 1] constants alpha_j^k
    const uint32_t* tab;               // the decodes' table (ids 0 .. n - 1, d = 2t): [t verify rows | 2t + 1 coefficient rows][2t + 1]
    uint32_t *Y, *Z, *opened, *c;      // Y[party][recipient][G], Z[recipient][G], opened[G][2t + 1], c[party][N]
    uint8_t* status;                   // [n G] as the two decodes leave it: [0, G) the second's, [G, n G) recipients 1 .. n - 1 of the first
    uint32_t *summary_first, *summary;
    uint32_t* counters;                // the stream's decode counters ([24], [25]: the first decode)
    size_t G, N;
    int n, t;
    uint32_t r2[9];                    // R^2: canonical -> Montgomery
};

// LDS words: X[n][M] | Y[n][n] | Z[n] | opened[M] (elements in limb form, `ls` words apart) | bad flags [n + 1] | vmat | tab
// (nl words per constant: 9 and ls = 12 for U29, 2 and 2 for Goldilocks)
struct TripleGenWgLds {
    int X, Y, Z, O, flags, vmat, tab, total;
    __host__ __device__ TripleGenWgLds(int n, int t, int ls, int nl) {
        const int M = 2 * t + 1;
        X = 0, Y = X + n * M * ls, Z = Y + n * n * ls, O = Z + n * ls, flags = O + M * ls;
        vmat = flags + ((n + 1 + 3) & ~3);
        tab = vmat + n * M * nl;
        total = tab + (t + M) * M * nl;
    }
};

template <class F>
__global__ __launch_bounds__(256) void k_triplegen_wg(TripleGenWgArgs a) {
    using E = typename F::E;
    constexpr bool SHARE = F::NL == 9;             // U29: a row's products shared by adjacent lanes (dot_shared); Goldilocks' are cheap
    constexpr int NL = F::NL, EW = F::EW, LS = SHARE ? 12 : F::NL;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x, n = a.n, t = a.t, M = 2 * t + 1, nv = t;
    const size_t g = blockIdx.x;
    const TripleGenWgLds L(n, t, LS, NL);
    uint32_t *X = lds + L.X, *Yl = lds + L.Y, *Zl = lds + L.Z, *Ol = lds + L.O, *flags = lds + L.flags, *vmat = lds + L.vmat, *tab = lds + L.tab;
    auto put_limbs = [&](uint32_t* dst, const E& v) {
#pragma unroll
        for (int i = 0; i < NL; ++i) dst[i] = v.l[i];
    };
    auto put_words = [&](uint32_t* dst, const E& canon) {  // canonical value -> the stored form
        if constexpr (SHARE) {
            uint32_t w[8];
            F::to_words(canon, w);
            *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
            *reinterpret_cast<uint4*>(dst + 4) = make_uint4(w[4], w[5], w[6], w[7]);
        } else {
            F::store_lt2r(dst, canon);
        }
    };
    auto dot = [&](auto&& value_of, const uint32_t* row, int lk, int sidx) -> E {
        if constexpr (SHARE) {
            return dot_shared<F>(value_of, row, M, lk, sidx);
        } else {
            typename F::Acc acc;
            F::acc_zero(acc);
            for (int i = 0; i < M; ++i) F::acc_mac(acc, value_of(i), row + i * NL);
            return F::acc_reduce(acc);
        }
    };
    constexpr int LK1 = SHARE ? 1 : 0;

    // ---- loads: a lane per (party, triple of the chunk); the tables by everyone ----------------------------------------------------
    const int pk_p = tid / M, pk_k = tid - pk_p * M;
    const bool pk = tid < n * M;
    const size_t e = pk ? (size_t)pk_p * a.N + g * M + pk_k : 0;
    E va = F::zero(), vb = F::zero(), vr = F::zero(), vrt = F::zero();
    if (pk) va = F::load(a.a + e * EW), vb = F::load(a.b + e * EW), vr = F::load(a.r2t + e * EW), vrt = F::load(a.rt + e * EW);
    for (int w = tid; w < n * M * NL; w += 256) vmat[w] = a.vmat[w];
    for (int w = tid; w < (nv + M) * M * NL; w += 256) tab[w] = a.tab[w];
    if (tid <= n) flags[tid] = 0;
    // [ab - r]_2t = a_i b_i - r2t_i  (triple_generation.rs:333-340), canonical as k_triple_local stores it
    if (pk) {
        const E am = F::mulc(va, a.r2);
        put_limbs(X + (pk_p * M + pk_k) * LS, F::canon_loose(F::template sub<2>(F::mont(vb, am), vr)));
    }
    __syncthreads();

    // ---- encode (batch_recon.rs:157-165): a lane per (party p, recipient j): y = sum_k alpha_j^k x_p[k] --------------------------------
    {
        const int p = tid / n, j = tid - p * n;
        if (tid < n * n) {
            const E y = F::canon_loose(dot([&](int k) { return F::load_const(X + (p * M + k) * LS); }, vmat + (size_t)j * M * NL, 0, 0));
            put_limbs(Yl + (p * n + j) * LS, y);
            put_words(a.Y + (((size_t)p * n + j) * a.G + g) * EW, y);
        }
    }
    __syncthreads();

    // ---- the EvalBatch arm (:384-391): recipient j opens its value from the n senders' y_p[j]: t verify rows and the P(0) row, a lane
    // pair per (recipient, row) ----------------------------------------------------------------------------------------------------
    {
        const int q = tid >> LK1, sidx = tid & ((1 << LK1) - 1), j = q / (nv + 1), r = q - j * (nv + 1);
        E kept = F::zero();
        if (q < n * (nv + 1)) {
            kept = dot([&](int i) { return F::load_const(Yl + (i * n + j) * LS); }, tab + (size_t)(r < nv ? r : nv) * M * NL, LK1, sidx);
            if (r < nv && sidx == 0 && !F::eq_canon(F::canon_loose(kept), F::load_const(Yl + ((M + r) * n + j) * LS))) flags[j] = 1;
        }
        __syncthreads();
        if (q < n * (nv + 1) && r == nv && sidx == 0) {
            const bool ok = flags[j] == 0;
            const E z = ok ? F::canon_loose(kept) : F::zero();
            put_limbs(Zl + j * LS, z);
            put_words(a.Z + ((size_t)j * a.G + g) * EW, z);
            if (a.status && j > 0) a.status[(size_t)j * a.G + g] = ok ? 0 : (uint8_t)DecodingError;  // recipient 0's is rewritten by the second decode
            if (!ok) {
                atomicAdd(a.counters + 24, 1u);
                atomicMax(a.counters + 25, 0xffffffffu - (uint32_t)((size_t)j * a.G + g));
            }
        }
    }
    __syncthreads();

    // ---- the RevealBatch arm (:457-467): everyone interpolates the 2t + 1 opened values from the n broadcast z_j: t verify rows and
    // 2t + 1 coefficient rows, a lane quad per row -----------------------------------------------------------------------------------
    {
        const int lk = !SHARE ? 0 : M >= 4 ? 2 : 1;
        const int r = tid >> lk, sidx = tid & ((1 << lk) - 1);
        E kept = F::zero();
        if (r < nv + M) {
            kept = dot([&](int i) { return F::load_const(Zl + i * LS); }, tab + (size_t)r * M * NL, lk, sidx);
            if (r < nv && sidx == 0 && !F::eq_canon(F::canon_loose(kept), F::load_const(Zl + (M + r) * LS))) flags[n] = 1;
        }
        __syncthreads();
        const bool ok = flags[n] == 0;
        if (r >= nv && r < nv + M && sidx == 0) {
            const E o = ok ? F::canon_loose(kept) : F::zero();
            put_limbs(Ol + (r - nv) * LS, o);
            put_words(a.opened + (g * M + (r - nv)) * EW, o);
        }
        if (tid == 0) {
            if (a.status) a.status[g] = ok ? 0 : (uint8_t)DecodingError;
            if (!ok) {
                atomicAdd(a.counters, 1u);
                atomicMax(a.counters + 1, 0xffffffffu - (uint32_t)g);
            }
        }
    }
    __syncthreads();

    // ---- [c]_t = rt_i + opened  (triple_generation.rs:196-208) ------------------------------------------------------------------------
    if (pk) F::store_loose(a.c + e * EW, F::add(vrt, F::load_const(Ol + pk_k * LS)));

    // ---- the summaries: the last workgroup turns the counters into them and leaves the counters at zero ------------------------------
    __syncthreads();
    if (tid != 0) return;
    __threadfence();
    const unsigned nblocks = gridDim.x, sub = blockIdx.x % DIRECT_FAN, quota = nblocks / DIRECT_FAN + (sub < nblocks % DIRECT_FAN ? 1u : 0u);
    if (atomicAdd(a.counters + 8 + sub, 1u) != quota - 1) return;
    const unsigned groups = nblocks < DIRECT_FAN ? nblocks : DIRECT_FAN;
    if (atomicAdd(a.counters + 3, 1u) != groups - 1) return;
    __threadfence();
#pragma unroll
    for (unsigned k = 0; k < DIRECT_FAN; ++k) store_handoff(a.counters + 8 + k, 0u);
    const uint32_t f1 = load_handoff(a.counters + 24), l1 = load_handoff(a.counters + 25);
    const uint32_t f2 = load_handoff(a.counters), l2 = load_handoff(a.counters + 1);
    if (a.summary_first) {
        a.summary_first[0] = f1, a.summary_first[1] = f1;
        a.summary_first[2] = f1 ? 0xffffffffu - l1 : 0xffffffffu;
        a.summary_first[3] = f1 ? (uint32_t)DecodingError : 0u;
    }
    if (a.summary) {
        a.summary[0] = f2, a.summary[1] = f2;
        a.summary[2] = f2 ? 0xffffffffu - l2 : 0xffffffffu;
        a.summary[3] = f2 ? (uint32_t)DecodingError : 0u;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) store_handoff(a.counters + k, 0u);
    store_handoff(a.counters + 24, 0u);
    store_handoff(a.counters + 25, 0u);
}

}  // namespace hbmpc
