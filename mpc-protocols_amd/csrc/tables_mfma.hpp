// tables_mfma.hpp -- host-side construction of the byte-digit tables of the matrix-core kernels (kernels_mfma.hpp).
//
// A constant-matrix map of the path (verify = Y * VM, coeffs = Y * BC, robust_interpolate.rs:391-427; y = X * V,
// common/share/mod.rs:50-76) multiplies BATCH data by coefficients that depend only on (n, d, t, sender ids).  A
// canonical element already is 32 base-256 digits, so with T[i][a] = c_i * 2^(8a) mod r precomputed on the host
//     sum_i c_i * y_i  ==  sum_{i, a} ybyte[i][a] * T[i][a]      (mod r)
// and, digit by digit (no convolution: the shift 2^(8a) is inside T),
//     L[b] = sum_{i, a} ybyte[i][a] * digit_b(T[i][a]),  b < 32          -- an int8 GEMM [chunks x 32 m] * [32 m x 32]
// followed by ONE carry pass and one small-quotient reduction per output element.
//
// v_mfma_i32_32x32x32_i8 is signed x signed, so the table holds BALANCED digits d in [-128, 127] and the kernel feeds
// data bytes as s = y - 128 (one XOR per dword): sum_k y_k T_k = sum_k s_k T_k + 128 sum_k T_k.  The second term is a
// constant of the row; its residue mod r travels -- as 32 byte digits -- in the accumulator's initial value, together
// with a bias that keeps every digit sum non-negative (the bias digits sum to a multiple of r, so they change nothing
// mod r).  Digit sums therefore stay below 32 m * 32768 + 512 < 0xff0000 for m <= 15: two of them combine into one
// 32-bit value without overflow (v_lshl_add_u32), which halves the 64-bit work of the carry pass.
//
// Row layout (one output row = one coefficient row c_0 .. c_{m-1}):  m slabs of 1024 bytes, slab i = the A operand
// of the MFMA of input i exactly as the 64 lanes hold it (lane (rho, ha): 16 digits, element j <-> data byte
// a = 16 ha + j of digit row rho), then 128 bytes of accumulator bias [lane half][16 registers] as int32.
// The digit b of the result lives in MFMA row rho(b) such that lane half h = b / 16 holds digits 16 h .. 16 h + 15 in
// its 16 accumulator registers in order (C/D map of the 32x32 shapes: row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)).
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

#include "host_fr.hpp"

namespace hbmpc {

constexpr size_t MF_BIAS_BYTES = 128;
inline size_t mf_row_bytes(size_t m) { return m * 1024 + MF_BIAS_BYTES; }
inline int mf_row_of_digit(int b) {
    const int h = b >> 4, reg = b & 15;
    return (reg & 3) + 8 * (reg >> 2) + 4 * h;
}
// per-digit bias magnitude: |sum s d| <= 32 m * 128 * 128
inline uint32_t mf_bias_mag(size_t m) { return (uint32_t)(32 * m * 16384); }
constexpr size_t MF_MAX_M = 15;  // digit sums < 2 * 32 m * 16384 + 512 must stay below 0xff0000

namespace mfdetail {
// plain 256-bit integers mod r (canonical, little-endian u64 words): the table needs 32 m shifted copies of every
// coefficient, and a shift is cheaper as integer arithmetic than as a Montgomery product plus a conversion back
struct U5 {
    uint64_t w[5];
};
inline bool geq(const U5& a, const U5& b) {
    for (int i = 4; i >= 0; --i) {
        if (a.w[i] != b.w[i]) return a.w[i] > b.w[i];
    }
    return true;
}
inline void sub(U5& a, const U5& b) {
    unsigned __int128 br = 0;
    for (int i = 0; i < 5; ++i) {
        const unsigned __int128 d = (unsigned __int128)a.w[i] - b.w[i] - br;
        a.w[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
}
inline U5 modulus() { return U5{{HFr::MOD[0], HFr::MOD[1], HFr::MOD[2], HFr::MOD[3], 0}}; }
inline void mul256(uint64_t c[4]) {  // c <- 256 c mod r
    U5 x{{c[0] << 8, (c[1] << 8) | (c[0] >> 56), (c[2] << 8) | (c[1] >> 56), (c[3] << 8) | (c[2] >> 56), c[3] >> 56}};
    // quotient estimate from the top 72 bits: q <= floor(x / r) <= q + 1
    const unsigned __int128 top = ((unsigned __int128)x.w[4] << 64) | x.w[3];
    const uint64_t q = (uint64_t)(top / ((unsigned __int128)HFr::MOD[3] + 1));
    unsigned __int128 mc = 0, br = 0;
    for (int i = 0; i < 5; ++i) {  // x -= q r
        mc += (unsigned __int128)q * (i < 4 ? HFr::MOD[i] : 0);
        const unsigned __int128 d = (unsigned __int128)x.w[i] - (uint64_t)mc - br;
        x.w[i] = (uint64_t)d;
        br = (d >> 64) & 1;
        mc >>= 64;
    }
    const U5 r = modulus();
    while (geq(x, r)) sub(x, r);
    for (int i = 0; i < 4; ++i) c[i] = x.w[i];
}
inline void add_mod(uint64_t a[4], const uint64_t b[4]) {  // a <- a + b mod r (a, b < r)
    U5 x{{0, 0, 0, 0, 0}};
    unsigned __int128 cy = 0;
    for (int i = 0; i < 4; ++i) {
        cy += (unsigned __int128)a[i] + b[i];
        x.w[i] = (uint64_t)cy;
        cy >>= 64;
    }
    x.w[4] = (uint64_t)cy;
    const U5 r = modulus();
    if (geq(x, r)) sub(x, r);
    for (int i = 0; i < 4; ++i) a[i] = x.w[i];
}
}  // namespace mfdetail

// (Bmag sum_b 256^b) mod r as canonical words: the constant of the bias digits (a function of m only)
inline void mf_bias_e(size_t m, uint64_t e[4]) {
    const HFr s256 = HFr::from_u64(256);
    HFr acc = HFr::zero(), p = HFr::one();
    for (int b = 0; b < 32; ++b) {
        acc = acc + p;
        p = p * s256;
    }
    acc = acc * HFr::from_u64(mf_bias_mag(m));
    acc.to_canon(e);
}
// the rows x m coefficients as canonical words, row-major: the input of the device expansion (kernels_tables.hpp)
inline std::vector<uint32_t> mf_coeff_words(const std::vector<std::vector<HFr>>& C, size_t m) {
    std::vector<uint32_t> w;
    w.reserve(C.size() * m * 8);
    for (const auto& row : C)
        for (size_t i = 0; i < m; ++i) {
            uint64_t c[4];
            row[i].to_canon(c);
            for (int k = 0; k < 4; ++k) w.push_back((uint32_t)c[k]), w.push_back((uint32_t)(c[k] >> 32));
        }
    return w;
}

// C: rows x m coefficient matrix.  Returns rows * mf_row_bytes(m) bytes (as u32 words).
inline std::vector<uint32_t> build_mfma_table(const std::vector<std::vector<HFr>>& C, size_t m) {
    const size_t RB = mf_row_bytes(m);
    std::vector<uint32_t> out(C.size() * RB / 4, 0u);
    uint8_t* base = reinterpret_cast<uint8_t*>(out.data());
    // E = (Bmag * sum_b 256^b) mod r: the bias digits are Bmag - byte_b(E), which sum to a multiple of r
    const uint32_t bmag = mf_bias_mag(m);
    const HFr s256 = HFr::from_u64(256);
    HFr acc = HFr::zero(), p = HFr::one();
    for (int b = 0; b < 32; ++b) {
        acc = acc + p;
        p = p * s256;
    }
    acc = acc * HFr::from_u64(bmag);
    uint64_t e[4];
    acc.to_canon(e);
    for (size_t r = 0; r < C.size(); ++r) {
        uint8_t* row = base + r * RB;
        uint64_t tsum[4] = {0, 0, 0, 0};  // sum_k T_k mod r
        for (size_t i = 0; i < m; ++i) {
            int8_t* tile = reinterpret_cast<int8_t*>(row + i * 1024);
            uint64_t c[4];
            C[r][i].to_canon(c);
            // balanced digits of x = the bytes of x + 0x80..80 (carries propagate), each minus 128: one 256-bit add
            // and one XOR per shifted copy.  (The top byte of a canonical value is <= 0x73: no carry out.)
            int8_t D[32][32];  // D[a][b] = digit b of c * 2^(8a) mod r
            for (int a = 0; a < 32; ++a) {
                mfdetail::add_mod(tsum, c);
                uint64_t y[4];
                unsigned __int128 cy = 0;
                for (int w = 0; w < 4; ++w) {
                    cy += (unsigned __int128)c[w] + 0x8080808080808080ULL;
                    y[w] = (uint64_t)cy ^ 0x8080808080808080ULL;
                    cy >>= 64;
                }
                memcpy(D[a], y, 32);
                mfdetail::mul256(c);
            }
            for (int b = 0; b < 32; ++b)
                for (int ha = 0; ha < 2; ++ha) {
                    int8_t* dst = tile + (mf_row_of_digit(b) + 32 * ha) * 16;
                    for (int j = 0; j < 16; ++j) dst[j] = D[16 * ha + j][b];
                }
        }
        uint64_t cr[4] = {tsum[0], tsum[1], tsum[2], tsum[3]};  // 128 * sum_k T_k mod r
        for (int k = 0; k < 7; ++k) {
            uint64_t dbl[4] = {cr[0], cr[1], cr[2], cr[3]};
            mfdetail::add_mod(cr, dbl);
        }
        int32_t* bias = reinterpret_cast<int32_t*>(row + m * 1024);
        for (int b = 0; b < 32; ++b) {
            const int32_t eb = (int32_t)((e[b >> 3] >> (8 * (b & 7))) & 0xff);
            const int32_t cb = (int32_t)((cr[b >> 3] >> (8 * (b & 7))) & 0xff);
            bias[b] = (int32_t)bmag - eb + cb;
        }
    }
    return out;
}

// The pair table of kernels_mfma_bfly.hpp for the rows C[0 .. n) of a map whose rows k and k + half differ by the sign of
// the odd columns (the Vandermonde rows of a domain of 2 half roots of unity).  Pair p < half: the slabs of row p, then
// bE = (b_p + b') / 2 and bT = (b_p - b') / 2 with b_p the plain bias of row p and b' the plain bias of row p + half in a
// digit representation of the same parity: adding r's bytes flips digit 0 (r is odd), and "digit b + 1, digit b - 1 - 256"
// flips digit b alone; neither changes the value mod r, and the digit sums move by at most 512 -- inside the slack the
// bound of the plain table leaves (|s d| <= 128 * 127 on the negative side).
constexpr size_t MF_BFLY_BIAS_BYTES = 256;
// m = 16 is admitted for the pair tables (n x n maps on a 16-point domain: the preprocessing producers' mixing matrix and its
// inverse): the worst-case bound above exceeds 0xff0000 by 0.4 % there, so the builder PROVES the bound for the table at hand --
// the extremes of every digit sum over all inputs, from the table's own digits -- and returns an empty table when it does
// not hold (the caller then takes the FFT kernels).  For m <= 15 the same proof is an internal check that cannot fail.
constexpr size_t MF_BFLY_MAX_M = 16;
inline size_t mf_bfly_row_bytes(size_t m) { return m * 1024 + MF_BFLY_BIAS_BYTES; }
inline std::vector<uint32_t> build_mfma_bfly_table(const std::vector<std::vector<HFr>>& C, size_t m, size_t half) {
    const size_t n = C.size(), RB = mf_row_bytes(m), PB = mf_bfly_row_bytes(m), pairs = half < n ? half : n;
    const std::vector<uint32_t> plain = build_mfma_table(C, m);
    const uint8_t* src = reinterpret_cast<const uint8_t*>(plain.data());
    std::vector<uint32_t> out(pairs * PB / 4, 0u);
    uint8_t* dst = reinterpret_cast<uint8_t*>(out.data());
    for (size_t p = 0; p < pairs; ++p) {
        memcpy(dst + p * PB, src + p * RB, m * 1024);
        const int32_t* b1 = reinterpret_cast<const int32_t*>(src + p * RB + m * 1024);
        int32_t* bE = reinterpret_cast<int32_t*>(dst + p * PB + m * 1024);
        int32_t* bT = bE + 32;
        const bool partner = p + half < n;
        int32_t b2[32];
        if (!partner) {
            for (int b = 0; b < 32; ++b) bE[b] = b1[b], bT[b] = 0, b2[b] = b1[b];
        } else {
            memcpy(b2, src + (p + half) * RB + m * 1024, sizeof b2);
            if ((b1[0] ^ b2[0]) & 1)
                for (int b = 0; b < 32; ++b) b2[b] += (int32_t)((HFr::MOD[b >> 3] >> (8 * (b & 7))) & 0xff);
            for (int b = 1; b < 32; ++b)
                if ((b1[b] ^ b2[b]) & 1) b2[b] += 1, b2[b - 1] -= 256;
            for (int b = 0; b < 32; ++b) bE[b] = (b1[b] + b2[b]) / 2, bT[b] = (b1[b] - b2[b]) / 2;
        }
        // the bound: data bytes are s in [-128, 127]; digit b of slab (i, a) sits at tile[(row_of_digit(b) + 32 (a >> 4)) * 16 + (a & 15)]
        for (int b = 0; b < 32; ++b) {
            int64_t lo[2] = {0, 0}, hi[2] = {0, 0};
            for (size_t i = 0; i < m; ++i) {
                const int8_t* tile = reinterpret_cast<const int8_t*>(src + p * RB + i * 1024);
                for (int a = 0; a < 32; ++a) {
                    const int64_t v = tile[(mf_row_of_digit(b) + 32 * (a >> 4)) * 16 + (a & 15)];
                    lo[i & 1] += v < 0 ? 127 * v : -128 * v;
                    hi[i & 1] += v < 0 ? -128 * v : 127 * v;
                }
            }
            const int64_t plus_lo = b1[b] + lo[0] + lo[1], plus_hi = b1[b] + hi[0] + hi[1];
            const int64_t minus_lo = b2[b] + lo[0] - hi[1], minus_hi = b2[b] + hi[0] - lo[1];
            if (plus_lo < 0 || plus_hi >= 0xff0000 || (partner && (minus_lo < 0 || minus_hi >= 0xff0000))) return {};
        }
    }
    return out;
}

}  // namespace hbmpc
