// kernels_codec.hpp -- wire payloads of the path, produced/consumed straight from the device rows
// (SURVEY.md section 8(f) row 1, Appendix B).  ark-serialize "compressed" encodings:
//   F                     32-byte little-endian canonical integer          (== our U256 in memory)
//   Vec<F>                u64-LE length, then the elements                 (EvalBatch / RevealBatch payloads,
//                         batch_recon.rs:173-176,396-398; common/utils.rs:3-21)
//   ShamirShare<F,1,P>    share[0] (32 B) | id u64 LE | degree u64 LE = 48 B   (common/mod.rs:92-99)
//   Vec<RobustShare<F>>   u64-LE length, then 48-byte records              (share_gen.rs:262-266)
// Payloads start at arbitrary 8-byte-aligned offsets, so they are moved as u64 words.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hbmpc {

__device__ __forceinline__ bool is_canonical_u64x4(uint64_t a0, uint64_t a1, uint64_t a2, uint64_t a3) {
    // a < r, r = 0x73eda753299d7d48 3339d80809a1d805 53bda402fffe5bfe ffffffff00000001
    const uint64_t r3 = 0x73eda753299d7d48ULL, r2 = 0x3339d80809a1d805ULL, r1 = 0x53bda402fffe5bfeULL,
                   r0 = 0xffffffff00000001ULL;
    if (a3 != r3) return a3 < r3;
    if (a2 != r2) return a2 < r2;
    if (a1 != r1) return a1 < r1;
    return a0 < r0;
}

// rows[r][g] (row stride in elements) -> payload r = [G][elements]; grid (ceil(G/256), n_rows)
__global__ __launch_bounds__(256) void k_pack_fvec(const uint64_t* __restrict__ rows, size_t row_stride, size_t G,
                                                   uint64_t* __restrict__ payloads, size_t payload_stride_words) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t r = blockIdx.y;
    uint64_t* p = payloads + r * payload_stride_words;
    if (g == 0) p[0] = (uint64_t)G;
    if (g >= G) return;
    const uint64_t* src = rows + (r * row_stride + g) * 4;
    const uint64_t a0 = src[0], a1 = src[1], a2 = src[2], a3 = src[3];
    uint64_t* dst = p + 1 + g * 4;
    dst[0] = a0, dst[1] = a1, dst[2] = a2, dst[3] = a3;
}
// payload r -> rows[r][g]; status[r] = 0 ok, 4 (InvalidInput) when the length prefix differs from G or an
// element is not canonical (ark's deserialize_compressed returns InvalidData for both)
__global__ __launch_bounds__(256) void k_unpack_fvec(const uint64_t* __restrict__ payloads, size_t payload_stride_words,
                                                     size_t G, uint64_t* __restrict__ rows, size_t row_stride,
                                                     uint32_t* __restrict__ status) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t r = blockIdx.y;
    const uint64_t* p = payloads + r * payload_stride_words;
    if (p[0] != (uint64_t)G) {
        if (g == 0) atomicMax(&status[r], 4u);
        return;
    }
    if (g >= G) return;
    const uint64_t* src = p + 1 + g * 4;
    const uint64_t a0 = src[0], a1 = src[1], a2 = src[2], a3 = src[3];
    if (!is_canonical_u64x4(a0, a1, a2, a3)) atomicMax(&status[r], 4u);
    uint64_t* dst = rows + (r * row_stride + g) * 4;
    dst[0] = a0, dst[1] = a1, dst[2] = a2, dst[3] = a3;
}
// In-place wire path: when a payload starts 8 bytes before a 32-byte boundary, its elements are 32-byte aligned and
// the encode kernel writes them directly (output row stride = payload stride); only the length prefixes remain.
__global__ __launch_bounds__(256) void k_fvec_prefix(uint64_t* __restrict__ payloads, size_t payload_stride_words, size_t G,
                                                     size_t n_rows) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_rows) payloads[r * payload_stride_words] = (uint64_t)G;
}
// k_unpack_fvec without the copy: the receive side decodes out of the payloads in place (sender rows at the payload
// stride), so deserialisation is only this check -- one read-only pass
__global__ __launch_bounds__(256) void k_validate_fvec(const uint64_t* __restrict__ payloads, size_t payload_stride_words,
                                                       size_t G, uint32_t* __restrict__ status) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t r = blockIdx.y;
    const uint64_t* p = payloads + r * payload_stride_words;
    if (p[0] != (uint64_t)G) {
        if (g == 0) atomicMax(&status[r], 4u);
        return;
    }
    if (g >= G) return;
    const uint64_t* src = p + 1 + g * 4;
    const bool bad = !is_canonical_u64x4(src[0], src[1], src[2], src[3]);
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicMax(&status[r], 4u);  // one atomic per wave that saw one
}
// Goldilocks (ark Fp64: 8-byte little-endian canonical elements, the in-memory form): Vec<F> payloads need no
// alignment trick at all -- any 8-byte-aligned payload has aligned elements
__global__ __launch_bounds__(256) void k_validate_fvec_gl(const uint64_t* __restrict__ payloads, size_t payload_stride_words,
                                                          size_t G, uint32_t* __restrict__ status) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t r = blockIdx.y;
    const uint64_t* p = payloads + r * payload_stride_words;
    if (p[0] != (uint64_t)G) {
        if (g == 0) atomicMax(&status[r], 4u);
        return;
    }
    if (g >= G) return;
    const bool bad = p[1 + g] >= 0xFFFFFFFF00000001ULL;
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicMax(&status[r], 4u);
}
// values[N] + (id, degree) -> payload = [N][48-byte records]
__global__ __launch_bounds__(256) void k_pack_shares(const uint64_t* __restrict__ values, size_t N, uint64_t id,
                                                     uint64_t degree, uint64_t* __restrict__ payload) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) payload[0] = (uint64_t)N;
    if (i >= N) return;
    const uint64_t* src = values + i * 4;
    uint64_t* dst = payload + 1 + i * 6;
    dst[0] = src[0], dst[1] = src[1], dst[2] = src[2], dst[3] = src[3], dst[4] = id, dst[5] = degree;
}
// payload -> values[N]; status[0]: 4 bad length / non-canonical value, 3 (IdMismatch) a record with another id,
// 2 (DegreeMismatch) a record with another degree (max of the codes seen)
__global__ __launch_bounds__(256) void k_unpack_shares(const uint64_t* __restrict__ payload, size_t N, uint64_t id,
                                                       uint64_t degree, uint64_t* __restrict__ values,
                                                       uint32_t* __restrict__ status) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (payload[0] != (uint64_t)N) {
        if (i == 0) atomicMax(status, 4u);
        return;
    }
    if (i >= N) return;
    const uint64_t* src = payload + 1 + i * 6;
    const uint64_t a0 = src[0], a1 = src[1], a2 = src[2], a3 = src[3];
    uint32_t code = 0;
    if (src[5] != degree) code = 2;
    if (src[4] != id) code = 3;
    if (!is_canonical_u64x4(a0, a1, a2, a3)) code = 4;
    if (code) atomicMax(status, code);
    uint64_t* dst = values + i * 4;
    dst[0] = a0, dst[1] = a1, dst[2] = a2, dst[3] = a3;
}
// every element of a[N] canonical?  status[0] = 4 otherwise (the check Fr::from_bigint(..).unwrap() makes)
__global__ __launch_bounds__(256) void k_validate_canonical(const uint64_t* __restrict__ a, size_t N,
                                                            uint32_t* __restrict__ status) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const uint64_t* s = a + i * 4;
    if (!is_canonical_u64x4(s[0], s[1], s[2], s[3])) atomicMax(status, 4u);
}

// degree_out[g] = DensePolynomial::degree() of coeffs[g][0..m): index of the highest non-zero coefficient,
// 0 for the zero polynomial
// (ew64 = 64-bit words per element: 4 for Fr, 1 for Goldilocks)
__global__ __launch_bounds__(256) void k_poly_degree(const uint64_t* __restrict__ coeffs, size_t G, int m, int ew64,
                                                     uint32_t* __restrict__ degree_out, uint64_t* __restrict__ c0_out) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    if (c0_out)  // the constant term beside the degree (the RanDouSha verifier keeps these two of an interpolation)
        for (int w = 0; w < ew64; ++w) c0_out[g * ew64 + w] = coeffs[g * (size_t)m * ew64 + w];
    int deg = 0;
    for (int k = m - 1; k > 0; --k) {
        const uint64_t* c = coeffs + (g * (size_t)m + k) * ew64;
        uint64_t any = 0;
        for (int w = 0; w < ew64; ++w) any |= c[w];
        if (any != 0) {
            deg = k;
            break;
        }
    }
    degree_out[g] = (uint32_t)deg;
}


// ---- layout and verdict kernels of the preprocessing producers (RanSha / RanDouSha) --------------------------------
// dst[b][c][r] = src[b][r][c] for elements of W 64-bit words: the dealers' outputs S[dealer][recipient, k] become the
// recipients' inputs x[recipient, k][dealer] of the n x n Vandermonde (share_gen.rs:401-418, ran_dou_sha/mod.rs:392-403),
// and the selected output rows y[i][party, k] become each party's list [k][i] in the reference's order
// (share_gen.rs:199-203, ran_dou_sha/mod.rs:314-331).  A 16 x 64 tile goes through LDS so that both sides move whole
// runs of elements.
template <int W>
struct LayoutElem {
    uint64_t w[W];
};
template <int W>
__global__ __launch_bounds__(256) void k_transpose(const uint64_t* __restrict__ src, size_t rows, size_t cols, size_t src_row_stride,
                                                   uint64_t* __restrict__ dst, size_t dst_row_stride, size_t src_batch_stride,
                                                   size_t dst_batch_stride) {
    constexpr int TR = 16, TC = 64;
    __shared__ LayoutElem<W> tile[TR][TC + 1];
    const LayoutElem<W>* s = reinterpret_cast<const LayoutElem<W>*>(src) + (size_t)blockIdx.z * src_batch_stride;
    LayoutElem<W>* d = reinterpret_cast<LayoutElem<W>*>(dst) + (size_t)blockIdx.z * dst_batch_stride;
    const size_t r0 = (size_t)blockIdx.y * TR, c0 = (size_t)blockIdx.x * TC;
    for (int k = threadIdx.x; k < TR * TC; k += 256) {
        const int r = k / TC, c = k % TC;
        if (r0 + r < rows && c0 + c < cols) tile[r][c] = s[(r0 + r) * src_row_stride + c0 + c];
    }
    __syncthreads();
    for (int k = threadIdx.x; k < TR * TC; k += 256) {
        const int c = k / TR, r = k % TR;
        if (r0 + r < rows && c0 + c < cols) d[(c0 + c) * dst_row_stride + r0 + r] = tile[r][c];
    }
}
// The rows outside [row0, row0 + rows) of y[row][party, k] (G = parties K elements per row) party-major: row number r' among them of chunk
// (j, k) to dst[(j nother + r') K + k] -- what the mixing kernel writes itself where it covers the shape (MfmaRowsArgs::other_stride)
template <int W>
__global__ __launch_bounds__(256) void k_rows_party_major(const uint64_t* __restrict__ src, size_t G, size_t K, int row0, int rows, int nother,
                                                          uint64_t* __restrict__ dst) {
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= G) return;
    const int rp = blockIdx.y, row = rp < row0 ? rp : rp + rows;
    const size_t j = c / K, k = c - j * K;
    reinterpret_cast<LayoutElem<W>*>(dst)[(j * nother + rp) * K + k] = reinterpret_cast<const LayoutElem<W>*>(src)[(size_t)row * G + c];
}
// degree (DensePolynomial::degree(): 0 for the zero polynomial) of polynomial g, coefficients at coeffs + g * stride
template <int W>
__device__ inline int layout_degree(const uint64_t* coeffs, int m) {
    for (int k = m - 1; k > 0; --k) {
        uint64_t any = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) any |= coeffs[(size_t)k * W + w];
        if (any != 0) return k;
    }
    return 0;
}
// RanSha verifier (share_gen.rs:516-530): counts the chunks whose reconstruction failed (status not 0 / 1) or whose
// polynomial does not have degree exactly `want`.  bad[0] += count, bad[1] = min(first bad chunk).
template <int W>
__global__ __launch_bounds__(256) void k_check_degree(const uint64_t* __restrict__ coeffs, const uint8_t* __restrict__ status, size_t G, int m,
                                                      int want, uint32_t* __restrict__ bad) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool wrong = false;
    if (g < G) wrong = (status && status[g] > 1) || layout_degree<W>(coeffs + g * (size_t)m * W, m) != want;
    const unsigned long long mask = __ballot(wrong);
    if (mask != 0 && (threadIdx.x & 63) == __ffsll((long long)mask) - 1) {
        atomicAdd(bad, (uint32_t)__popcll(mask));
        atomicMin(bad + 1, (uint32_t)g);
    }
}
// The same verdict from the TOP coefficient alone (hbmpc_dev_batch_recover_coeff_strided with k = want): a polynomial of at most
// `want` coefficients + 1 has degree exactly `want` > 0 iff that coefficient is not zero; want = 0 leaves the status test
template <int W>
__global__ __launch_bounds__(256) void k_check_top_coeff(const uint64_t* __restrict__ top, const uint8_t* __restrict__ status, size_t G, int want,
                                                         uint32_t* __restrict__ bad, size_t columns) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool wrong = false;
    if (g < G) {
        uint64_t any = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) any |= top[g * W + w];
        wrong = (status && status[g] > 1) || (want > 0 && any == 0);
    }
    const unsigned long long mask = __ballot(wrong);
    if (mask != 0 && (threadIdx.x & 63) == __ffsll((long long)mask) - 1) {
        atomicAdd(bad, (uint32_t)__popcll(mask));
        atomicMin(bad + 1, (uint32_t)(columns ? g % columns : g));  // several verifiers' columns in one launch: the column
    }
}
// RanDouSha's verifier from the two kept coefficients of each interpolation (hbmpc_dev_interpolate_degree_check_strided): sel_t / sel_2t are
// [G][2] = (constant term, coefficient t resp. 2t); a status above 1 says the points did not lie on a polynomial of that degree at all
template <int W>
__global__ __launch_bounds__(256) void k_check_double_sel(const uint64_t* __restrict__ sel_t, const uint8_t* __restrict__ st_t,
                                                          const uint64_t* __restrict__ sel_2t, const uint8_t* __restrict__ st_2t, size_t G, int t,
                                                          uint32_t* __restrict__ bad, size_t columns) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool wrong = false;
    if (g < G) {
        const uint64_t *a = sel_t + g * 2 * W, *b = sel_2t + g * 2 * W;
        uint64_t top_a = 0, top_b = 0, diff = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) top_a |= a[W + w], top_b |= b[W + w], diff |= a[w] ^ b[w];
        wrong = st_t[g] > 1 || st_2t[g] > 1 || (t > 0 && (top_a == 0 || top_b == 0)) || diff != 0;
    }
    const unsigned long long mask = __ballot(wrong);
    if (mask != 0 && (threadIdx.x & 63) == __ffsll((long long)mask) - 1) {
        atomicAdd(bad, (uint32_t)__popcll(mask));
        atomicMin(bad + 1, (uint32_t)(columns ? g % columns : g));
    }
}
// the same two coefficients and the verdict from a FULL interpolation (coeffs [G][m]): the form of the shapes the selective decode
// does not cover
template <int W>
__global__ __launch_bounds__(256) void k_pick_two(const uint64_t* __restrict__ coeffs, size_t G, int m, int d, uint64_t* __restrict__ sel,
                                                  uint8_t* __restrict__ status) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const uint64_t* c = coeffs + g * (size_t)m * W;
    uint64_t high = 0;
    for (int k = d + 1; k < m; ++k)
#pragma unroll
        for (int w = 0; w < W; ++w) high |= c[(size_t)k * W + w];
    status[g] = high ? (uint8_t)8 : 0;  // ShareErrorCode::DecodingError
#pragma unroll
    for (int w = 0; w < W; ++w) sel[g * 2 * W + w] = high ? 0 : c[w], sel[g * 2 * W + W + w] = high ? 0 : c[(size_t)d * W + w];
}
// RanDouSha verifier (ran_dou_sha/mod.rs:586-589): degree(poly_t) == t, degree(poly_2t) == 2 t and equal constant terms
template <int W>
__global__ __launch_bounds__(256) void k_check_double(const uint64_t* __restrict__ ct, const uint64_t* __restrict__ c2t, size_t G, int m, int t,
                                                      uint32_t* __restrict__ bad) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool wrong = false;
    if (g < G) {
        const uint64_t *a = ct + g * (size_t)m * W, *b = c2t + g * (size_t)m * W;
        wrong = layout_degree<W>(a, m) != t || layout_degree<W>(b, m) != 2 * t;
#pragma unroll
        for (int w = 0; w < W; ++w) wrong |= a[w] != b[w];
    }
    const unsigned long long mask = __ballot(wrong);
    if (mask != 0 && (threadIdx.x & 63) == __ffsll((long long)mask) - 1) {
        atomicAdd(bad, (uint32_t)__popcll(mask));
        atomicMin(bad + 1, (uint32_t)g);
    }
}
// the same tests from what the interpolation kernels leave when the coefficients are not needed: the constant terms c0[G] and the
// degrees[G] of the two polynomials of every column (hbmpc_dev_batch_interpolate_c0)
template <int W>
__global__ __launch_bounds__(256) void k_check_double_c0(const uint64_t* __restrict__ c0t, const uint32_t* __restrict__ degt,
                                                         const uint64_t* __restrict__ c02t, const uint32_t* __restrict__ deg2t, size_t G, int t,
                                                         uint32_t* __restrict__ bad, size_t columns) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool wrong = false;
    if (g < G) {
        wrong = degt[g] != (uint32_t)t || deg2t[g] != (uint32_t)(2 * t);
#pragma unroll
        for (int w = 0; w < W; ++w) wrong |= c0t[g * W + w] != c02t[g * W + w];
    }
    const unsigned long long mask = __ballot(wrong);
    if (mask != 0 && (threadIdx.x & 63) == __ffsll((long long)mask) - 1) {
        atomicAdd(bad, (uint32_t)__popcll(mask));
        atomicMin(bad + 1, (uint32_t)(g % columns));  // several verifiers' columns in one array: the column, not the chunk
    }
}
// c0[g] = coeffs[g][0] of chunk-major coefficient rows (the shapes without a fused kernel)
template <int W>
__global__ __launch_bounds__(256) void k_take_c0(const uint64_t* __restrict__ coeffs, size_t G, int m, uint64_t* __restrict__ c0) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
#pragma unroll
    for (int w = 0; w < W; ++w) c0[g * W + w] = coeffs[g * (size_t)m * W + w];
}

}  // namespace hbmpc
