// kernels_rng.hpp -- seeded generation of the random polynomial coefficients of compute_shares on the device.
// The reference draws them from the caller's `rng: &mut impl Rng` (DensePolynomial::rand,
// robust_interpolate.rs:68); a device path needs a generator whose output is a function of (seed, position) only.
// Contract "hbmpc-chacha20-v1" (restated in oracle/spec.py, seeded_coefficient): coefficient k (1 <= k <= degree) of
// secret number b = first candidate < modulus in the ChaCha20 keystream with key = seed, nonce = b,
// block counter = (k << 32) + attempt, attempt = 0, 1, ...; a 64-byte block holds 64 / element-bytes little-endian
// candidates (Fr: bit 255 cleared first).  ChaCha20 because these coefficients are what hides the secret: they
// must be cryptographically unpredictable, not merely well distributed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fr_consts.h"

namespace hbmpc {

struct SeedArg {
    uint32_t k[8];
};

__device__ __forceinline__ uint32_t rotl32(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }
#define HB_QR(a, b, c, d)                                   \
    a += b; d = rotl32(d ^ a, 16); c += d; b = rotl32(b ^ c, 12); \
    a += b; d = rotl32(d ^ a, 8);  c += d; b = rotl32(b ^ c, 7);

__device__ __forceinline__ void chacha20_block(const SeedArg& key, uint64_t counter, uint64_t nonce, uint32_t (&out)[16]) {
    const uint32_t in[16] = {0x61707865u, 0x3320646Eu, 0x79622D32u, 0x6B206574u, key.k[0], key.k[1], key.k[2], key.k[3],
                             key.k[4], key.k[5], key.k[6], key.k[7], (uint32_t)counter, (uint32_t)(counter >> 32),
                             (uint32_t)nonce, (uint32_t)(nonce >> 32)};
    uint32_t s0 = in[0], s1 = in[1], s2 = in[2], s3 = in[3], s4 = in[4], s5 = in[5], s6 = in[6], s7 = in[7], s8 = in[8], s9 = in[9],
             s10 = in[10], s11 = in[11], s12 = in[12], s13 = in[13], s14 = in[14], s15 = in[15];
#pragma unroll 2
    for (int r = 0; r < 10; ++r) {
        HB_QR(s0, s4, s8, s12) HB_QR(s1, s5, s9, s13) HB_QR(s2, s6, s10, s14) HB_QR(s3, s7, s11, s15)
        HB_QR(s0, s5, s10, s15) HB_QR(s1, s6, s11, s12) HB_QR(s2, s7, s8, s13) HB_QR(s3, s4, s9, s14)
    }
    const uint32_t s[16] = {s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15};
#pragma unroll
    for (int i = 0; i < 16; ++i) out[i] = s[i] + in[i];
}
#undef HB_QR

// coeffs[b][0] = secrets[b] (or seeded coefficient (first_index + b, 0) when secrets is null);
// coeffs[b][k] = seeded coefficient (first_index + b, k), k = 1..d.  One lane per element.
// EW = u32 words per element: 8 (bls12-381 Fr) or 2 (Goldilocks).
template <int EW>
__global__ __launch_bounds__(256) void k_fill_coeffs(SeedArg seed, const uint32_t* __restrict__ secrets, size_t B,
                                                     uint64_t first_index, int dp1, uint32_t* __restrict__ coeffs) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * (size_t)dp1) return;
    const size_t b = e / (size_t)dp1;
    const uint32_t k = (uint32_t)(e - b * (size_t)dp1);
    uint32_t* dst = coeffs + e * EW;
    if (k == 0 && secrets) {  // secrets == nullptr: the secret is drawn too (stream position k = 0): a RanSha dealer
#pragma unroll
        for (int w = 0; w < EW; ++w) dst[w] = secrets[b * EW + w];
        return;
    }
    for (uint32_t attempt = 0;; ++attempt) {
        uint32_t blk[16];
        chacha20_block(seed, ((uint64_t)k << 32) + attempt, first_index + b, blk);
#pragma unroll
        for (int c = 0; c < 16 / EW; ++c) {
            uint32_t v[EW];
#pragma unroll
            for (int w = 0; w < EW; ++w) v[w] = blk[c * EW + w];
            bool lt = false;
            if constexpr (EW == 8) {
                v[7] &= 0x7fffffffu;
                // v < r, most significant word first (S_MOD = r as eight u32)
#pragma unroll
                for (int w = 7; w >= 0; --w) {
                    if (v[w] != consts::S_MOD[w]) {
                        lt = v[w] < consts::S_MOD[w];
                        break;
                    }
                }
            } else {
                const uint64_t x = ((uint64_t)v[1] << 32) | v[0];
                lt = x < 0xFFFFFFFF00000001ull;
            }
            if (lt) {
#pragma unroll
                for (int w = 0; w < EW; ++w) dst[w] = v[w];
                return;
            }
        }
    }
}

}  // namespace hbmpc
