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
// data bytes as s = y - 128 (one XOR per dword); sum y d = sum s d + 128 sum d.  The per-digit constant 128 sum d and a
// bias that keeps every digit sum non-negative (a multiple of r in total, so it changes nothing mod r) travel as the
// accumulator's initial value.
//
// Row layout (one output row = one coefficient row c_0 .. c_{m-1}):  m slabs of 1024 bytes, slab i = the A operand
// of the MFMA of input i exactly as the 64 lanes hold it (lane (rho, ha): 16 digits, element j <-> data byte
// a = 16 ha + j of digit row rho), then 128 bytes of accumulator bias [lane half][16 registers] as int32.
// The digit b of the result lives in MFMA row rho(b) such that lane half h = b / 16 holds digits 16 h .. 16 h + 15 in
// its 16 accumulator registers in order (C/D map of the 32x32 shapes: row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)).
#pragma once
#include <stdint.h>

#include <vector>

#include "host_fr.hpp"

namespace hbmpc {

constexpr size_t MF_BIAS_BYTES = 128;
inline size_t mf_row_bytes(size_t m) { return m * 1024 + MF_BIAS_BYTES; }
inline int mf_row_of_digit(int b) {
    const int h = b >> 4, reg = b & 15;
    return (reg & 3) + 8 * (reg >> 2) + 4 * h;
}
// per-digit bias magnitude: |sum y d| <= 32 m * 255 * 128 < 32 m * 32768
inline uint32_t mf_bias_mag(size_t m) { return (uint32_t)(32 * m * 32768); }

// C: rows x m coefficient matrix.  Returns rows * mf_row_bytes(m) bytes (as u32 words).
inline std::vector<uint32_t> build_mfma_table(const std::vector<std::vector<HFr>>& C, size_t m) {
    const size_t RB = mf_row_bytes(m);
    std::vector<uint32_t> out(C.size() * RB / 4, 0u);
    uint8_t* base = reinterpret_cast<uint8_t*>(out.data());
    const HFr s256 = HFr::from_u64(256);
    // E = (Bmag * sum_b 256^b) mod r: the bias digits are Bmag - byte_b(E), which sum to a multiple of r
    const uint32_t bmag = mf_bias_mag(m);
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
        int64_t dsum[32] = {0};
        for (size_t i = 0; i < m; ++i) {
            int8_t* tile = reinterpret_cast<int8_t*>(row + i * 1024);
            HFr v = C[r][i];
            for (int a = 0; a < 32; ++a) {
                uint64_t c[4];
                v.to_canon(c);
                int carry = 0;
                for (int b = 0; b < 32; ++b) {
                    int x = (int)((c[b >> 3] >> (8 * (b & 7))) & 0xff) + carry;
                    carry = x >= 128;
                    if (carry) x -= 256;
                    const int lane = mf_row_of_digit(b) + 32 * (a >> 4);
                    tile[lane * 16 + (a & 15)] = (int8_t)x;
                    dsum[b] += x;
                }
                // the top byte of a canonical value is <= 0x73: the last digit never carries out
                v = v * s256;
            }
        }
        int32_t* bias = reinterpret_cast<int32_t*>(row + m * 1024);
        for (int b = 0; b < 32; ++b) {
            const int32_t eb = (int32_t)((e[b >> 3] >> (8 * (b & 7))) & 0xff);
            bias[b] = (int32_t)(128 * dsum[b]) + (int32_t)bmag - eb;
        }
    }
    return out;
}

}  // namespace hbmpc
