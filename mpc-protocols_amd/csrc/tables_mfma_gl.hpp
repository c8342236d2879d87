// tables_mfma_gl.hpp -- host-side byte-digit tables of the Goldilocks matrix-core kernel (kernels_mfma_gl.hpp).
//
// The formulation of tables_mfma.hpp carried over to p = 2^64 - 2^32 + 1, where it is far cheaper: an element is 8 base-256
// digits, so with T[i][a] = c_i 2^(8a) mod p
//     sum_i c_i y_i  ==  sum_{i, a} ybyte[i][a] T[i][a]   (mod p),      L[b] = sum_{i, a} ybyte[i][a] digit_b(T[i][a]),  b < 8
// is an int8 GEMM with K = 8 m data bytes and 8 result digits per table row: FOUR table rows share one 32-row MFMA tile
// and the 21 rows of a (d = t = 10) decode are 6 tiles x 3 K-steps = 18 MFMAs per 32 chunks.
//
// Balanced digits: v_mfma_i32_32x32x32_i8 is signed x signed, data bytes are fed as y - 128.  T is taken as the
// representative of its class in [-0x8080808080808080, 0x7f7f7f7f7f7f7f7f] (T or T - p: the interval is 2^64 - 1 >= p wide),
// whose 8 balanced digits are the bytes of T + 0x80..80, each minus 128.  128 * sum T mod p travels as 8 byte digits in the
// accumulator's initial value together with a bias Bmag - e_b (e = Bmag * sum_b 256^b mod p, so the bias is a multiple of p)
// that keeps every digit sum non-negative: sums stay below 2 * 8 m * 16384 + 512 < 2^23 for m <= 16.
//
// Layout.  K-step s covers data bytes [32 s, 32 s + 32) = elements 4 s .. 4 s + 3.  Tile mt covers table rows 4 mt .. 4 mt + 3:
// lane half h of the accumulator holds rows 4 mt + 2 h and 4 mt + 2 h + 1, eight digits each, in registers 0..7 and 8..15
// (C/D map of the 32x32 shapes: MFMA row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)).
// One record per tile, so that the first tiles of a table are a table (P(0)-only decode = the first nv + 1 rows):
//   [ksteps] slabs of 1024 bytes: the A operand as the 64 lanes hold it (lane (row, ha): 16 digits, element j <-> data
//            byte 32 s + 16 ha + j)
//   128 bytes: accumulator bias [lane half][16 registers] as int32
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

#include "host_fr.hpp"

namespace hbmpc {

constexpr size_t MFGL_MAX_M = 16;  // K = 8 m <= 128 bytes = 4 K-steps
inline size_t mfgl_ksteps(size_t m) { return (8 * m + 31) / 32; }
inline size_t mfgl_tiles(size_t rows) { return (rows + 3) / 4; }
inline size_t mfgl_table_bytes(size_t rows, size_t m) { return mfgl_tiles(rows) * (mfgl_ksteps(m) * 1024 + 128); }
inline uint32_t mfgl_bias_mag(size_t m) { return (uint32_t)(8 * m * 16384); }

inline std::vector<uint32_t> build_mfma_table_gl(const std::vector<std::vector<HGl>>& C, size_t m) {
    typedef unsigned __int128 u128;
    const size_t rows = C.size(), KS = mfgl_ksteps(m), NT = mfgl_tiles(rows);
    std::vector<uint32_t> out(mfgl_table_bytes(rows, m) / 4, 0u);
    uint8_t* base = reinterpret_cast<uint8_t*>(out.data());
    const size_t TR = KS * 1024 + 128;  // bytes per tile record
    (void)NT;
    const uint32_t bmag = mfgl_bias_mag(m);
    // e = Bmag * (1 + 256 + ... + 256^7) mod p
    uint64_t e;
    {
        u128 s = 0, pw = 1;
        for (int b = 0; b < 8; ++b) {
            s += pw;
            pw <<= 8;
        }
        e = (uint64_t)(((s % HGl::P) * bmag) % HGl::P);
    }
    for (size_t r = 0; r < rows; ++r) {
        const size_t mt = r / 4, h = (r % 4) / 2, el = r % 2;
        uint64_t tsum = 0;  // sum of T mod p
        for (size_t i = 0; i < m; ++i) {
            uint64_t c = C[r][i].v;
            for (int a = 0; a < 8; ++a) {
                tsum = (uint64_t)(((u128)tsum + c) % HGl::P);
                const uint64_t rep = c <= 0x7f7f7f7f7f7f7f7fULL ? c : c - HGl::P;  // two's complement of T - p when negative
                const uint64_t y = (rep + 0x8080808080808080ULL) ^ 0x8080808080808080ULL;
                const size_t kk = 8 * i + a, s = kk / 32, ha = (kk % 32) / 16, j = kk % 16;
                for (int b = 0; b < 8; ++b) {
                    const int reg = (int)(8 * el + b);
                    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (int)h;
                    reinterpret_cast<int8_t*>(base + mt * TR + s * 1024)[((size_t)row + 32 * ha) * 16 + j] = (int8_t)(y >> (8 * b));
                }
                c = (uint64_t)(((u128)c << 8) % HGl::P);
            }
        }
        const uint64_t cr = (uint64_t)(((u128)tsum * 128) % HGl::P);
        int32_t* bias = reinterpret_cast<int32_t*>(base + mt * TR + KS * 1024);
        for (int b = 0; b < 8; ++b)
            bias[h * 16 + 8 * el + b] = (int32_t)bmag - (int32_t)((e >> (8 * b)) & 0xff) + (int32_t)((cr >> (8 * b)) & 0xff);
    }
    return out;
}

}  // namespace hbmpc
