// kernels_eval.hpp -- polynomial evaluation on the roots-of-unity domain (reference rows a3 + a5):
//   y[j][g] = sum_k x[g][k] * alpha_j^k,  j < n,  alpha_j = omega_size^j,  size = next_pow2(n)
// This one linear map is both RobustShare::compute_shares (robust_interpolate.rs:52-82, a size-`size`
// FFT of the zero-padded coefficients) and make_vandermonde+apply_vandermonde
// (common/share/mod.rs:31-76, an n x (d+1) mat-vec).  Field arithmetic is exact, so any evaluation
// order gives the reference's bits.
//
// Fast path: one lane per chunk, a zero-pruned radix-2 DIT FFT held entirely in registers
// (<= 16 points per pass); domains larger than 16 are done in size/16 passes over the outputs
// j = r (mod size/16): pass r twists the coefficients by omega^(r k), folds them mod x^16 - 1 and
// runs the same 16-point FFT.  Pruning means a multiply only happens where the reference's FFT
// would multiply two non-trivial values: n=16,d=5 costs 15 modmuls per secret (direct: 80),
// n=31,d=10 costs 44 per chunk (direct: 341).
// (ys = elements between consecutive output rows; G when the party rows are dense, larger when they are written
// straight into wire payloads, hbmpc_dev_encode_fvec.)
// Layout: input chunk-major x[G][d+1] is staged through LDS with coalesced 16-byte loads (one
// wave-tile = 64 chunks, rows unpadded: LDS capacity, not LDS bank conflicts, is what matters here);
// output party-major y[n][G]: a wave stores 2 KiB contiguous per party, each output canonicalised and
// stored as soon as the last stage produces it.  The same templates are instantiated for the
// Goldilocks field (fr_gold.hpp, 8-byte elements: F::EW = 2).
#pragma once
#include "eval_out.hpp"
#include <type_traits>
#include <utility>

#include "fr_sat.hpp"
#include "fr_u29.hpp"
#include "kernels_recover.hpp"

namespace hbmpc {

// ---------------------------------------------------------------------------------------------
// compile-time structure of the pruned FFT
// ---------------------------------------------------------------------------------------------
constexpr int bitrev_c(int log, int p) {
    int r = 0;
    for (int b = 0; b < log; ++b)
        if (p & (1 << b)) r |= 1 << (log - 1 - b);
    return r;
}
struct Bd {
    int lbu;  // limb bound in units of 2^29 (0: element is identically zero)
    int vb;   // value bound in units of r
};
constexpr int sub_k(int vb) {
    return vb <= 1 ? 2 : vb <= 2 ? 4 : vb <= 4 ? 8 : vb <= 8 ? 16 : vb <= 16 ? 32 : vb <= 32 ? 64 : (1 << 20);
}

// bound of X[idx] after `stage` butterfly stages (stage 0 = bit-reversed inputs).  The code
// generator below takes EXACTLY the same decisions (normalise u when lbu would pass 7, canonicalise
// a lazy v before it is subtracted un-multiplied, normalise it before it is multiplied).
constexpr Bd fft_bd(int LOG, int CNT, int LBU0, int VB0, int stage, int idx) {
    if (stage == 0) return bitrev_c(LOG, idx) < CNT ? Bd{LBU0, VB0} : Bd{0, 0};
    const int B = 1 << (stage - 1);
    const int pos = idx & (2 * B - 1), k = pos & (B - 1);
    const int iu = (idx & ~(2 * B - 1)) + k, iv = iu + B;
    const bool upper = pos >= B;
    const Bd u = fft_bd(LOG, CNT, LBU0, VB0, stage - 1, iu), v = fft_bd(LOG, CNT, LBU0, VB0, stage - 1, iv);
    if (v.lbu == 0) return u;
    // k != 0: t is a mulc output; k == 0: t is v itself when it is tight, else v canonicalised
    const Bd t0 = k != 0 ? Bd{1, 2} : ((v.lbu == 1 && v.vb <= 2) ? v : Bd{1, 1});
    const int K = sub_k(t0.vb);
    if (u.lbu == 0) return upper ? Bd{2, K} : t0;
    const Bd un = (u.lbu + 2 > 7) ? Bd{1, u.vb} : u;
    return upper ? Bd{un.lbu + 2, un.vb + K} : Bd{un.lbu + t0.lbu, un.vb + t0.vb};
}

template <class F, int K>
HB_DEV typename F::E sub_dispatch(const typename F::E& a, const typename F::E& b) {
    return F::template sub<K>(a, b);
}

template <class F, int LOG, int CNT, int LBU0, int VB0, int STAGE, int IDX>
HB_DEV void fft_bfly(typename F::E (&X)[1 << LOG], const uint32_t* __restrict__ tw) {
    using E = typename F::E;
    constexpr int S = 1 << LOG, B = 1 << STAGE;
    constexpr int blk = IDX / B, k = IDX % B, iu = blk * 2 * B + k, iv = iu + B;
    constexpr Bd bu = fft_bd(LOG, CNT, LBU0, VB0, STAGE, iu), bv = fft_bd(LOG, CNT, LBU0, VB0, STAGE, iv);
    if constexpr (bv.lbu == 0) {
        if constexpr (bu.lbu != 0) X[iv] = X[iu];
    } else {
        E t;
        constexpr bool tight = bv.lbu == 1 && bv.vb <= 2;
        if constexpr (k == 0) {
            t = tight ? X[iv] : F::canon_loose(X[iv]);
        } else {
            const E vin = bv.lbu > 4 ? F::normalize(X[iv]) : X[iv];
            t = F::mulc_u(vin, tw + (k * (S / (2 * B))) * F::NL);
        }
        constexpr int tvb = k == 0 ? (tight ? bv.vb : 1) : 2;
        constexpr int K = sub_k(tvb);
        static_assert(K <= 64, "value bound");
        if constexpr (bu.lbu == 0) {
            X[iu] = t;
            X[iv] = sub_dispatch<F, K>(F::zero(), t);
        } else {
            const E u = (bu.lbu + 2 > 7) ? F::normalize(X[iu]) : X[iu];
            X[iu] = F::add(u, t);
            X[iv] = sub_dispatch<F, K>(u, t);
        }
    }
}
template <class F, int LOG, int CNT, int LBU0, int VB0, int STAGE, int... IDX>
HB_DEV void fft_stage(typename F::E (&X)[1 << LOG], const uint32_t* __restrict__ tw,
                      std::integer_sequence<int, IDX...>) {
    (fft_bfly<F, LOG, CNT, LBU0, VB0, STAGE, IDX>(X, tw), ...);
}
template <class F, int LOG, int CNT, int LBU0, int VB0, int... STAGE>
HB_DEV void fft_stages(typename F::E (&X)[1 << LOG], const uint32_t* __restrict__ tw,
                       std::integer_sequence<int, STAGE...>) {
    (fft_stage<F, LOG, CNT, LBU0, VB0, STAGE>(X, tw, std::make_integer_sequence<int, (1 << LOG) / 2>{}), ...);
}
// X: bit-reversed inputs (X[p] = c[bitrev(p)], zero where bitrev(p) >= CNT) -> natural-order
// evaluations at omega_S^i.  tw[q] = omega_S^q (device-constant form), q < S/2.
template <class F, int LOG, int CNT, int LBU0, int VB0>
HB_DEV void fft_pruned(typename F::E (&X)[1 << LOG], const uint32_t* __restrict__ tw) {
    if constexpr (LOG > 0) fft_stages<F, LOG, CNT, LBU0, VB0>(X, tw, std::make_integer_sequence<int, LOG>{});
}
// The same transform, but every output is handed to `sink(index, value)` as soon as its last butterfly is
// done (the last stage produces X[i] and X[i + S/2] together).  The kernels canonicalise and store there, so
// the stores of a tile are spread over the last stage's multiplies instead of bunched behind the arithmetic
// -- the kernel's traffic alone needs ~150 us per 2^20 secrets (profiles/r01_store_pattern_ubench.txt), as
// long as the arithmetic, so the two have to overlap.
// The last TWO stages are walked together, quarter by quarter: for k < S/4 the two stage-(LOG-2) butterflies
// that produce X[k], X[k+S/4], X[S/2+k], X[S/2+k+S/4] are immediately followed by the two last-stage
// butterflies that consume exactly those four values and hand the four finished outputs to the sink.  Same
// butterflies, same results -- but the 16 values of the last stage's input never exist at once: live data
// peaks at the 8 inputs of stage LOG-2 plus the quarter in flight (108 + 36 registers instead of 144 + ...),
// which is what lets these kernels run at 3 waves per SIMD.
template <class F, int LOG, int CNT, int LBU0, int VB0, int K, class Sink>
HB_DEV void fft_last_two_quarter(typename F::E (&X)[1 << LOG], const uint32_t* __restrict__ tw, Sink&& sink) {
    constexpr int S = 1 << LOG, Q = S / 4, H = S / 2;
    fft_bfly<F, LOG, CNT, LBU0, VB0, LOG - 2, K>(X, tw);
    fft_bfly<F, LOG, CNT, LBU0, VB0, LOG - 2, Q + K>(X, tw);
    fft_bfly<F, LOG, CNT, LBU0, VB0, LOG - 1, K>(X, tw);
    sink(std::integral_constant<int, K>{}, X[K]);
    sink(std::integral_constant<int, K + H>{}, X[K + H]);
    fft_bfly<F, LOG, CNT, LBU0, VB0, LOG - 1, K + Q>(X, tw);
    sink(std::integral_constant<int, K + Q>{}, X[K + Q]);
    sink(std::integral_constant<int, K + Q + H>{}, X[K + Q + H]);
}
template <class F, int LOG, int CNT, int LBU0, int VB0, class Sink, int... K>
HB_DEV void fft_last_two_sink(typename F::E (&X)[1 << LOG], const uint32_t* __restrict__ tw, Sink&& sink,
                              std::integer_sequence<int, K...>) {
    (fft_last_two_quarter<F, LOG, CNT, LBU0, VB0, K>(X, tw, sink), ...);
}
// plain order: all of stage LOG-2, then the last stage with the sink after each butterfly
template <class F, int LOG, int CNT, int LBU0, int VB0, class Sink, int... IDX>
HB_DEV void fft_last_stage_sink(typename F::E (&X)[1 << LOG], const uint32_t* __restrict__ tw, Sink&& sink,
                                std::integer_sequence<int, IDX...>) {
    ((fft_bfly<F, LOG, CNT, LBU0, VB0, LOG - 1, IDX>(X, tw), sink(std::integral_constant<int, IDX>{}, X[IDX]),
      sink(std::integral_constant<int, IDX + (1 << (LOG - 1))>{}, X[IDX + (1 << (LOG - 1))])),
     ...);
}
template <class F, int LOG, int CNT, int LBU0, int VB0, class Sink>
HB_DEV void fft_pruned_sink(typename F::E (&X)[1 << LOG], const uint32_t* __restrict__ tw, Sink&& sink) {
    if constexpr (LOG == 0) {
        sink(std::integral_constant<int, 0>{}, X[0]);
    } else if constexpr (LOG == 1) {
        fft_bfly<F, 1, CNT, LBU0, VB0, 0, 0>(X, tw);
        sink(std::integral_constant<int, 0>{}, X[0]);
        sink(std::integral_constant<int, 1>{}, X[1]);
    } else if constexpr (F::template eval_interleave<LOG, CNT>()) {
        if constexpr (LOG > 2) fft_stages<F, LOG, CNT, LBU0, VB0>(X, tw, std::make_integer_sequence<int, LOG - 2>{});
        fft_last_two_sink<F, LOG, CNT, LBU0, VB0>(X, tw, sink, std::make_integer_sequence<int, (1 << LOG) / 4>{});
    } else {  // all 16 values are live anyway (more than 6 non-zero inputs): the plain order schedules better
        fft_stages<F, LOG, CNT, LBU0, VB0>(X, tw, std::make_integer_sequence<int, LOG - 1>{});
        fft_last_stage_sink<F, LOG, CNT, LBU0, VB0>(X, tw, sink, std::make_integer_sequence<int, (1 << LOG) / 2>{});
    }
}
template <int LOG, int CNT, int LBU0, int VB0>
constexpr int fft_max_vb() {
    int m = 0;
    for (int i = 0; i < (1 << LOG); ++i) {
        const Bd b = fft_bd(LOG, CNT, LBU0, VB0, LOG, i);
        if (b.vb > m) m = b.vb;
        if (b.lbu > 7) return 1 << 20;
    }
    return m;
}

// ---------------------------------------------------------------------------------------------
// LDS tile: 64 chunks x DP1 elements, row pitch DP1 * element bytes (+ TILE_PAD words)
// ---------------------------------------------------------------------------------------------
constexpr int EVAL_TILE = 64;  // chunks per wave-tile (= one wavefront)

// EW = u32 words per element (8: Fr, 2: Goldilocks); a piece is 16 bytes (Fr) or one 8-byte element
// Row padding of the tile in u32 words.  32-byte elements: none -- a padded row would make the per-lane
// ds_read_b128 bank-conflict free, but the ~300 LDS reads of a tile are nothing next to its ~14 k VALU
// instructions, while 16 bytes per row cost a whole wave of occupancy (dp1 = 11: 22.5 KB per wave fits 7 waves per
// CU, 23.5 KB only 6): measured -4 % on the config-3 encode, -1 % on config 2.
template <int EW>
constexpr int TILE_PAD = EW >= 4 ? 0 : 2;
template <int EW>
HB_DEV int tile_pitch_words(int dp1) { return dp1 * EW + TILE_PAD<EW>; }

// stage rows [g0, g0+64) of x[G][dp1] into LDS: coalesced 16-byte pieces, wave-uniform bounds.
// All of a lane's loads (NP = ceil(2*dp1) pieces, compile-time) are issued back-to-back BEFORE the first
// wait, so one memory latency is exposed per tile -- a `load; s_waitcnt vmcnt(0); ds_write` loop exposed
// twelve of them per wave (SQ_WAIT_ANY ~47 % of wave cycles, profiles/r01_pmc_compute_shares_v1.txt).
template <int NP, int EW>
HB_DEV void stage_tile(uint32_t* __restrict__ lds, const uint32_t* __restrict__ x, size_t g0, size_t G, int dp1,
                       int lane) {
    constexpr int PW = EW >= 4 ? 4 : 2;  // words per piece
    using piece_t = typename std::conditional<PW == 4, uint4, uint2>::type;
    const int pieces_per_row = dp1 * (EW / PW);
    const size_t rows = G - g0 < (size_t)EVAL_TILE ? G - g0 : (size_t)EVAL_TILE;
    const int total = (int)rows * pieces_per_row;
    const piece_t* src = reinterpret_cast<const piece_t*>(x + g0 * (size_t)dp1 * EW);
    const int pitch = tile_pitch_words<EW>(dp1);
    for (int base = 0; base < total; base += NP * EVAL_TILE) {  // one trip unless dp1 > NP/2 (fold kernels)
        piece_t v[NP];
        int pc[NP];
#pragma unroll
        for (int it = 0; it < NP; ++it) {
            // clamped: lanes past the end re-load (and re-write) the last piece -- no branches, no scratch
            const int p = base + it * EVAL_TILE + lane;
            pc[it] = p < total ? p : total - 1;
            v[it] = src[pc[it]];
        }
#pragma unroll
        for (int it = 0; it < NP; ++it) {
            const int row = pc[it] / pieces_per_row, part = pc[it] - row * pieces_per_row;
            *reinterpret_cast<piece_t*>(lds + row * pitch + part * PW) = v[it];
        }
    }
}

// statically indexed input/output passes (fold expressions: a partially unrolled loop would index
// X[] dynamically and push the whole array to scratch)
template <class F, int LOG, int CNT, int P>
HB_DEV void load_plain_one(typename F::E (&X)[1 << LOG], const uint32_t* __restrict__ row, int dp1, bool fold) {
    constexpr int k = bitrev_c(LOG, P);
    if constexpr (k < CNT) {
        X[P] = F::load(row + k * F::EW);
        if (fold && k + 16 < dp1) X[P] = F::add(X[P], F::load(row + (k + 16) * F::EW));
    }
}
template <class F, int LOG, int CNT, int... P>
HB_DEV void load_plain(typename F::E (&X)[1 << LOG], const uint32_t* __restrict__ row, int dp1, bool fold,
                       std::integer_sequence<int, P...>) {
    (load_plain_one<F, LOG, CNT, P>(X, row, dp1, fold), ...);
}
template <class F, int CNT, int P>
HB_DEV void load_twisted_one(typename F::E (&X)[16], const uint32_t* __restrict__ row,
                             const uint32_t* __restrict__ tr, int dp1, bool fold) {
    constexpr int k = bitrev_c(4, P);
    if constexpr (k < CNT) {
        X[P] = F::mulc_u(F::load(row + k * F::EW), tr + k * F::NL);
        if (fold && k + 16 < dp1) X[P] = F::add(X[P], F::mulc_u(F::load(row + (k + 16) * F::EW), tr + (k + 16) * F::NL));
    }
}
template <class F, int CNT, int... P>
HB_DEV void load_twisted(typename F::E (&X)[16], const uint32_t* __restrict__ row, const uint32_t* __restrict__ tr,
                         int dp1, bool fold, std::integer_sequence<int, P...>) {
    (load_twisted_one<F, CNT, P>(X, row, tr, dp1, fold), ...);
}
// ---------------------------------------------------------------------------------------------
// single-pass kernel: size = 2^LOG <= 16, DP1 = CNT coefficients
// ---------------------------------------------------------------------------------------------
template <class F, int LOG, int CNT>
__global__ __launch_bounds__(EVAL_TILE) __attribute__((amdgpu_waves_per_eu(F::template eval_waves_min<LOG, CNT>(), F::template eval_waves<LOG, CNT>()))) void k_eval_fft1(const uint32_t* __restrict__ x, size_t G, int n,
                                                         const uint32_t* __restrict__ tw, uint32_t* __restrict__ y, size_t ys) {
    using E = typename F::E;
    constexpr int S = 1 << LOG;
    static_assert(CNT <= S && fft_max_vb<LOG, CNT, 1, 1>() <= 64, "bounds");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = threadIdx.x;
    const size_t g0 = (size_t)blockIdx.x * EVAL_TILE;
    x += (size_t)blockIdx.y * G * CNT * F::EW;  // party-batched launches: blockIdx.y = party (x[P][G][CNT] -> y[P][n][G])
    y += (size_t)blockIdx.y * n * ys * F::EW;
    stage_tile<CNT * (F::EW >= 4 ? 2 : 1), F::EW>(lds, x, g0, G, CNT, lane);
    __syncthreads();
    const size_t g = g0 + lane;
    if (g >= G) return;
    const uint32_t* row = lds + lane * tile_pitch_words<F::EW>(CNT);
    E X[S];
    load_plain<F, LOG, CNT>(X, row, CNT, false, std::make_integer_sequence<int, S>{});
    fft_pruned_sink<F, LOG, CNT, 1, 1>(X, tw, [&](auto idx, const E& v) {
        constexpr int j = decltype(idx)::value;
        if (j < n) F::store_loose(y + ((size_t)j * ys + g) * F::EW, v);
    });
}

// ---------------------------------------------------------------------------------------------
// The producers' n x n mixing step in ONE launch of the single-pass kernel (share_gen.rs:401-418, ran_dou_sha/mod.rs:392-403): a lane per
// chunk (party j, batch element k) = g / K, g % K reads its CNT = n inputs from the ROWS they were dealt into (row i at x + i * xs: for a
// lane-per-chunk kernel the rows are the coalesced side), and writes output rows [row0, row0 + rows) into the parties' lists
// ([k][row], up to two slices) and every other row either to y[row][G] or, with `others`, party-major to others[(j nother + r') K + k]
// -- what hbmpc_dev_vandermonde_apply_rows_lists / _split do with a transpose either side where no kernel reads rows (Goldilocks).
// ---------------------------------------------------------------------------------------------
template <class F, int LOG, int CNT, int P>
HB_DEV void load_rows_one(typename F::E (&X)[1 << LOG], const uint32_t* __restrict__ mine, size_t row_words) {
    constexpr int k = bitrev_c(LOG, P);
    if constexpr (k < CNT) X[P] = F::load(mine + (size_t)k * row_words);
}
template <class F, int LOG, int CNT, int... P>
HB_DEV void load_rows(typename F::E (&X)[1 << LOG], const uint32_t* __restrict__ mine, size_t row_words, std::integer_sequence<int, P...>) {
    (load_rows_one<F, LOG, CNT, P>(X, mine, row_words), ...);
}
template <class F, int LOG, int CNT>
__global__ __launch_bounds__(EVAL_TILE) __attribute__((amdgpu_waves_per_eu(F::template eval_waves_min<LOG, CNT>(), F::template eval_waves<LOG, CNT>()))) void k_eval_fft1_mix(
    const uint32_t* __restrict__ x, size_t xs, size_t G, int n, const uint32_t* __restrict__ tw, MixOut o) {
    using E = typename F::E;
    constexpr int S = 1 << LOG, EW = F::EW;
    static_assert(CNT <= S && fft_max_vb<LOG, CNT, 1, 1>() <= 64, "bounds");
    const size_t g = (size_t)blockIdx.x * EVAL_TILE + threadIdx.x;
    if (g >= G) return;
    E X[S];
    load_rows<F, LOG, CNT>(X, x + g * EW, xs * EW, std::make_integer_sequence<int, S>{});
    const size_t j = g / o.K, k = g - j * o.K;
    uint32_t* lrow = nullptr;  // row row0 of this chunk in its party's list (nullptr: no slice holds batch element k)
#pragma unroll
    for (int q = 0; q < 2; ++q)
        if (o.list[q].dst && k - o.list[q].k0 < o.list[q].count) lrow = o.list[q].dst + ((j * o.list[q].stride + (k - o.list[q].k0) * (size_t)o.rows)) * EW;
    const int nother = n - o.rows;
    fft_pruned_sink<F, LOG, CNT, 1, 1>(X, tw, [&](auto idx, const E& v) {
        constexpr int r = decltype(idx)::value;
        if (r >= n) return;
        if (r >= o.row0 && r < o.row0 + o.rows) {
            if (lrow) F::store_loose(lrow + (size_t)(r - o.row0) * EW, v);
        } else if (o.others) {
            const int rp = r < o.row0 ? r : r - o.rows;
            F::store_loose(o.others + ((j * (size_t)nother + rp) * o.K + k) * EW, v);
        } else {
            F::store_loose(o.y + ((size_t)r * G + g) * EW, v);
        }
    });
}

// ---------------------------------------------------------------------------------------------
// TripleGenNode::init_batch in ONE kernel (triple_gen/triple_generation.rs:333-340 followed by the BatchRecon encode,
// batch_recon/batch_recon.rs:157-165): the chunk values x[g][k] = a b - r2t are computed while the tile is staged, so
// the [party][N] array of local products is never written to HBM and read back (config 4: 4.4 of 22 GB per step).
// a, b, r2t: [party][G * CNT] canonical, the flat element order of the chunks; one lane per element while staging.
// ---------------------------------------------------------------------------------------------
struct TripleConsts {
    uint32_t r2[9];  // R^2 mod r in device-constant form (ElemConsts::r2)
};
template <class F, int LOG, int CNT>
__global__ __launch_bounds__(EVAL_TILE) __attribute__((amdgpu_waves_per_eu(F::template eval_waves_min<LOG, CNT>(), F::template eval_waves<LOG, CNT>()))) void k_eval_fft1_triple(
    const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, const uint32_t* __restrict__ r2t, size_t G, int n,
    const uint32_t* __restrict__ tw, uint32_t* __restrict__ y, size_t ys, TripleConsts cs) {
    using E = typename F::E;
    constexpr int S = 1 << LOG;
    static_assert(CNT <= S && fft_max_vb<LOG, CNT, 1, 1>() <= 64, "bounds");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = threadIdx.x;
    const size_t g0 = (size_t)blockIdx.x * EVAL_TILE;
    const size_t poff = (size_t)blockIdx.y * G * CNT * F::EW;  // blockIdx.y = party
    a += poff, b += poff, r2t += poff;
    y += (size_t)blockIdx.y * n * ys * F::EW;
    {
        const size_t rows = G - g0 < (size_t)EVAL_TILE ? G - g0 : (size_t)EVAL_TILE;
        const int total = (int)rows * CNT, pitch = tile_pitch_words<F::EW>(CNT);
        const size_t e0 = g0 * CNT;
#pragma unroll
        for (int it = 0; it < CNT; ++it) {  // rows * CNT elements, 64 per trip: CNT trips
            const int p = it * EVAL_TILE + lane;
            const int pc = p < total ? p : total - 1;  // clamped: lanes past the end redo the last element
            const E am = F::mulc(F::load(a + (e0 + pc) * F::EW), cs.r2);
            const E pr = F::mont(F::load(b + (e0 + pc) * F::EW), am);  // a b, < 2r
            const E c = F::template sub<2>(pr, F::load(r2t + (e0 + pc) * F::EW));
            const int row = pc / CNT, k = pc - row * CNT;
            F::store_loose(lds + row * pitch + k * F::EW, c);
        }
    }
    __syncthreads();
    const size_t g = g0 + lane;
    if (g >= G) return;
    const uint32_t* row = lds + lane * tile_pitch_words<F::EW>(CNT);
    E X[S];
    load_plain<F, LOG, CNT>(X, row, CNT, false, std::make_integer_sequence<int, S>{});
    fft_pruned_sink<F, LOG, CNT, 1, 1>(X, tw, [&](auto idx, const E& v) {
        constexpr int j = decltype(idx)::value;
        if (j < n) F::store_loose(y + ((size_t)j * ys + g) * F::EW, v);
    });
}

// ---------------------------------------------------------------------------------------------
// multi-pass kernel: size = 16 * P (P = 2, 4, 8, 16), DP1 <= 32 coefficients.
// twist[r][k] = omega_size^(r k) (device-constant form), r < P, k < dp1 (row r = 0 unused).
// CNT16 = min(dp1, 16).  FOLD = dp1 > 16.
// ---------------------------------------------------------------------------------------------
template <class F, int CNT16, bool FOLD>
__global__ __launch_bounds__(EVAL_TILE) __attribute__((amdgpu_waves_per_eu(F::template eval_waves_min<4, CNT16 + (FOLD ? 16 : 0)>(), F::template eval_waves<4, CNT16 + (FOLD ? 16 : 0)>()))) void k_eval_fftP(const uint32_t* __restrict__ x, size_t G, int n, int dp1,
                                                         int P, const uint32_t* __restrict__ tw16,
                                                         const uint32_t* __restrict__ twist,
                                                         uint32_t* __restrict__ y, size_t ys) {
    using E = typename F::E;
    constexpr int Q = FOLD ? 2 : 1;
    static_assert(fft_max_vb<4, CNT16, Q, Q>() <= 64 && fft_max_vb<4, CNT16, Q, 2 * Q>() <= 64, "bounds");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = threadIdx.x;
    const size_t g0 = (size_t)blockIdx.x * EVAL_TILE;
    x += (size_t)blockIdx.y * G * dp1 * F::EW;
    y += (size_t)blockIdx.y * n * ys * F::EW;
    stage_tile<(FOLD ? 8 : CNT16) * (F::EW >= 4 ? 2 : 1), F::EW>(lds, x, g0, G, dp1, lane);
    __syncthreads();
    const size_t g = g0 + lane;
    if (g >= G) return;
    const uint32_t* row = lds + lane * tile_pitch_words<F::EW>(dp1);
    {  // pass 0: no twist
        E X[16];
        load_plain<F, 4, CNT16>(X, row, dp1, FOLD, std::make_integer_sequence<int, 16>{});
        fft_pruned_sink<F, 4, CNT16, Q, Q>(X, tw16, [&](auto idx, const E& v) {
            const int j = P * decltype(idx)::value;
            if (j < n) F::store_loose(y + ((size_t)j * ys + g) * F::EW, v);
        });
    }
    for (int r = 1; r < P; ++r) {
        E X[16];
        load_twisted<F, CNT16>(X, row, twist + (size_t)r * dp1 * F::NL, dp1, FOLD, std::make_integer_sequence<int, 16>{});
        fft_pruned_sink<F, 4, CNT16, Q, 2 * Q>(X, tw16, [&](auto idx, const E& v) {
            const int j = r + P * decltype(idx)::value;
            if (j < n) F::store_loose(y + ((size_t)j * ys + g) * F::EW, v);
        });
    }
}

// ---------------------------------------------------------------------------------------------
// generic kernel (any n <= 2^32 domain, any d): Horner per output, coefficients re-read from
// global memory (chunk-major rows are L1/L2 resident while a lane walks its row).  O(n d) modmuls:
// the slow, always-available path for shapes without a specialised kernel.
// alpha[j] = omega^j in device-constant form.
// ---------------------------------------------------------------------------------------------
template <class F>
__global__ __launch_bounds__(256) void k_eval_generic(const uint32_t* __restrict__ x, size_t G, int n, int dp1,
                                                      const uint32_t* __restrict__ alpha, uint32_t* __restrict__ y, size_t ys) {
    using E = typename F::E;
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    x += (size_t)blockIdx.y * G * dp1 * F::EW;
    y += (size_t)blockIdx.y * n * ys * F::EW;
    const uint32_t* row = x + g * (size_t)dp1 * F::EW;
    for (int j = 0; j < n; ++j) {
        const uint32_t* a = alpha + (size_t)j * F::NL;
        E acc = F::load(row + (size_t)(dp1 - 1) * F::EW);
        for (int k = dp1 - 2; k >= 0; --k) {
            acc = F::mulc_u(acc, a);  // < 2r, normalised
            acc = F::add(acc, F::load(row + (size_t)k * F::EW));
            // value < 3r, limbs < 2^30: fine as the next mulc input
        }
        F::store_loose(y + ((size_t)j * ys + g) * F::EW, acc);
    }
}

// Small batches: one wave per chunk, one lane per evaluation point (Horner over the chunk's d+1 coefficients, which
// every lane reads from the same addresses).  The latency of d multiplications instead of a whole per-lane FFT; used
// while the call is too small to fill the chip anyway (see k_batch_recover_wide).  Same results as every other path.
template <class F>
__global__ __launch_bounds__(256) void k_eval_wide(const uint32_t* __restrict__ x, size_t G, int n, int dp1,
                                                   const uint32_t* __restrict__ alpha, uint32_t* __restrict__ y, size_t ys) {
    using E = typename F::E;
    const int lane = threadIdx.x & 63;
    const size_t g = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    x += (size_t)blockIdx.y * G * dp1 * F::EW;
    y += (size_t)blockIdx.y * n * ys * F::EW;
    const uint32_t* row = x + g * (size_t)dp1 * F::EW;
    for (int j = lane; j < n; j += 64) {
        const uint32_t* a = alpha + (size_t)j * F::NL;
        E acc = F::load(row + (size_t)(dp1 - 1) * F::EW);
        for (int k = dp1 - 2; k >= 0; --k) {
            acc = F::mulc(acc, a);
            acc = F::add(acc, F::load(row + (size_t)k * F::EW));
        }
        F::store_loose(y + ((size_t)j * ys + g) * F::EW, acc);
    }
}

// The same for U29 as a table product: a wave per chunk, 2^lk lanes per evaluation point j, y_j = sum_k alpha_j^k x_k with the
// constants alpha_j^k staged in LDS once per workgroup (vmat [n][dp1]) -- depth ceil(dp1 / 2^lk) x 81 + 72 dependent
// v_mad_u64_u32 instead of Horner's (dp1 - 1) x 153 (n = 16, d = 5, four lanes per point: 234 against 765; a lone wave per SIMD
// issues one every ~10 cycles, profiles/r01_isa_rates.txt, r04_small_batch_fpmul.txt).  dot_shared: kernels_recover.hpp.
template <class F>
__global__ __launch_bounds__(256) void k_eval_wide_dot(const uint32_t* __restrict__ x, size_t G, int n, int dp1,
                                                       const uint32_t* __restrict__ vmat, uint32_t* __restrict__ y, size_t ys, int lk) {
    using E = typename F::E;
    extern __shared__ __attribute__((aligned(16))) uint32_t vm[];
    for (int w = threadIdx.x; w < n * dp1 * F::NL; w += 256) vm[w] = vmat[w];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const size_t g = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    x += (size_t)blockIdx.y * G * dp1 * F::EW;
    y += (size_t)blockIdx.y * n * ys * F::EW;
    const uint32_t* row = x + g * (size_t)dp1 * F::EW;
    const int sidx = lane & ((1 << lk) - 1), per = 64 >> lk;
    for (int j0 = 0; j0 < n; j0 += per) {  // whole groups of 2^lk lanes share a point: in or out together
        const int j = j0 + (lane >> lk);
        if (j < n) {
            const E v = dot_shared<F>([&](int k) { return F::load(row + (size_t)k * F::EW); }, vm + (size_t)j * dp1 * F::NL, dp1, lk, sidx);
            if (sidx == 0) F::store_loose(y + ((size_t)j * ys + g) * F::EW, v);
        }
    }
}

}  // namespace hbmpc
