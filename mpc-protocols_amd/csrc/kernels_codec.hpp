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
                                                     uint32_t* __restrict__ degree_out) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
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

}  // namespace hbmpc
