// kernels_tables.hpp -- the byte-digit table of the matrix-core kernels (tables_mfma.hpp) expanded ON THE DEVICE.
//
// A new sender set needs rows x m coefficients (a few KB, computed on the host: tables.hpp) and their 32 shifted copies
// each as balanced byte digits in MFMA operand order -- 239 KB for config 3, 0.4 - 0.6 ms of host loops plus the upload,
// which is what a mid-size decode with a sender set not seen before used to wait for (BatchRecon decodes with the FIRST
// d + t + 1 arrivals, batch_recon.rs:371-389: the set changes from session to session).  Here a thread per (row, input)
// walks c, 256 c, 256^2 c, ... mod r with plain 256-bit integer steps -- the same arithmetic, step for step, as
// tables_mfma.hpp::build_mfma_table, which stays as the reference the tests compare this against -- and a second launch
// writes the bias rows.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hbmpc {
namespace tb {

__device__ static constexpr uint64_t R64[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};

__device__ inline bool geq_r(const uint64_t* x /*5 words*/) {
    if (x[4]) return true;
    for (int i = 3; i >= 0; --i) {
        if (x[i] != R64[i]) return x[i] > R64[i];
    }
    return true;
}
__device__ inline void sub_r(uint64_t* x /*5 words*/) {
    uint64_t br = 0;
    for (int i = 0; i < 5; ++i) {
        const uint64_t m = i < 4 ? R64[i] : 0;
        const uint64_t d = x[i] - m, d2 = d - br;
        br = (x[i] < m) | (d < br);
        x[i] = d2;
    }
}
// c <- 256 c mod r   (c < r)
__device__ inline void mul256(uint64_t* c) {
    uint64_t x[5] = {c[0] << 8, (c[1] << 8) | (c[0] >> 56), (c[2] << 8) | (c[1] >> 56), (c[3] << 8) | (c[2] >> 56), c[3] >> 56};
    // quotient estimate from the top 40 bits: q <= floor(x / r) (at most a few short of it: the loop below finishes)
    const uint64_t top = (x[4] << 32) | (x[3] >> 32);
    const uint64_t q = top / ((R64[3] >> 32) + 1);
    uint64_t carry = 0, br = 0;  // x -= q r
    for (int i = 0; i < 5; ++i) {
        const uint64_t m = i < 4 ? R64[i] : 0;
        const uint64_t lo = q * m, hi = __umul64hi(q, m);
        const uint64_t p = lo + carry;
        carry = hi + (p < lo);
        const uint64_t d = x[i] - p, d2 = d - br;
        br = (x[i] < p) | (d < br);
        x[i] = d2;
    }
    while (geq_r(x)) sub_r(x);
    for (int i = 0; i < 4; ++i) c[i] = x[i];
}
__device__ inline void add_mod(uint64_t* a, const uint64_t* b) {  // a <- a + b mod r   (a, b < r)
    uint64_t x[5], cy = 0;
    for (int i = 0; i < 4; ++i) {
        const uint64_t s = a[i] + b[i], s2 = s + cy;
        cy = (s < a[i]) | (s2 < s);
        x[i] = s2;
    }
    x[4] = cy;
    if (geq_r(x)) sub_r(x);
    for (int i = 0; i < 4; ++i) a[i] = x[i];
}
__device__ inline int row_of_digit(int b) {
    const int h = b >> 4, reg = b & 15;
    return (reg & 3) + 8 * (reg >> 2) + 4 * h;
}

// thread (row, input): the 1 KiB slab of that pair; partial[row * m + input] = sum_a c 256^a mod r
__global__ __launch_bounds__(64) void k_mfma_table_slabs(const uint64_t* __restrict__ coeff, int m, int rows, uint8_t* __restrict__ table,
                                                         uint64_t* __restrict__ partial) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * m) return;
    const int r = idx / m, i = idx % m;
    uint8_t* tile = table + (size_t)r * (m * 1024 + 128) + (size_t)i * 1024;
    uint64_t c[4], tsum[4] = {0, 0, 0, 0};
    for (int k = 0; k < 4; ++k) c[k] = coeff[(size_t)idx * 4 + k];
    for (int a = 0; a < 32; ++a) {
        add_mod(tsum, c);
        // balanced digits of x = the bytes of x + 0x80..80 (carries propagate), each minus 128
        uint64_t y[4], cy = 0;
        for (int w = 0; w < 4; ++w) {
            const uint64_t s = c[w] + 0x8080808080808080ULL, s2 = s + cy;
            cy = (s < c[w]) | (s2 < s);
            y[w] = s2 ^ 0x8080808080808080ULL;
        }
        const int ha = a >> 4, j = a & 15;
        for (int b = 0; b < 32; ++b) tile[(row_of_digit(b) + 32 * ha) * 16 + j] = (uint8_t)(y[b >> 3] >> (8 * (b & 7)));
        mul256(c);
    }
    for (int k = 0; k < 4; ++k) partial[(size_t)idx * 4 + k] = tsum[k];
}
struct TableE {
    uint64_t w[4];  // (Bmag sum_b 256^b) mod r
};
// thread row: bias[b] = Bmag - byte_b(E) + byte_b(128 sum_k T_k mod r)
__global__ __launch_bounds__(64) void k_mfma_table_bias(const uint64_t* __restrict__ partial, int m, int rows, TableE E, uint32_t bmag,
                                                        uint8_t* __restrict__ table) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    uint64_t t[4] = {0, 0, 0, 0};
    for (int i = 0; i < m; ++i) add_mod(t, partial + ((size_t)r * m + i) * 4);
    for (int k = 0; k < 7; ++k) {
        uint64_t dbl[4] = {t[0], t[1], t[2], t[3]};
        add_mod(t, dbl);
    }
    int32_t* bias = reinterpret_cast<int32_t*>(table + (size_t)r * (m * 1024 + 128) + (size_t)m * 1024);
    for (int b = 0; b < 32; ++b) {
        const int32_t eb = (int32_t)((E.w[b >> 3] >> (8 * (b & 7))) & 0xff);
        const int32_t cb = (int32_t)((t[b >> 3] >> (8 * (b & 7))) & 0xff);
        bias[b] = (int32_t)bmag - eb + cb;
    }
}

}  // namespace tb
}  // namespace hbmpc
