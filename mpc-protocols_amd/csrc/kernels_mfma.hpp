// kernels_mfma.hpp -- the constant-matrix maps of the path on the gfx950 matrix cores (v_mfma_i32_32x32x32_i8).
//
//   decode  (batch_recover_secret, robust_interpolate.rs:391-427): verify rows  Y * VM  and coefficient rows  Y * BC
//   encode  (apply_vandermonde, common/share/mod.rs:50-76; compute_shares = the same map, robust_interpolate.rs:52-82)
//
// Formulation and table layout: tables_mfma.hpp.  One MFMA tile = 32 chunks (columns, B operand = the chunk's canonical
// bytes straight from HBM, sign-flipped with one XOR per dword) x 32 result digits (rows, A operand = table slab);
// K = 32 bytes of one input element, so an output element of a chunk costs m MFMAs per 32 chunks.  A lane pair
// (c, h = 0 / 1) holds the 32 digit sums of chunk c: digits 16 h .. 16 h + 15 in the 16 accumulator registers, i.e.
// each lane owns one 128-bit half of the 256-bit result.  The epilogue stays in registers:
//   gather    digits (< 2^25, spaced 8 bits) -> 4 x 32-bit words + carry per half (16 v_mad_u64_u32)
//   verify    r = 1 (mod 2^32), so  S = y + q r  <=>  q = (S - y) mod 2^32  and  S + q (2^256 - r) = y + q 2^256:
//             one 4-word multiply-add chain per half, exact, no quotient estimate and no conditional subtraction
//   reduce    q' = floor(top 49 bits / (r >> 224) + 1) <= q, R = S - q' r; R < r whenever word 8 cancels and the top
//             word is below r's top word; the (rare) rest takes a wave-uniform slow path of conditional subtractions
// Carries cross from the low half to the high half once per chain (v_permlane32_swap).
// The table row of the current output streams through LDS (double buffered, one barrier per output row); a workgroup
// of 4 waves x CG tiles re-uses it for 128 CG chunks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fr_u29.hpp"
#include "kernels_recover.hpp"

namespace hbmpc {
namespace mf {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// words of r and of 2^256 - r, least significant first
__device__ static constexpr uint32_t R_W[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                               0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
__device__ static constexpr uint32_t NR_W[8] = {0xffffffffu, 0x00000000u, 0x0001a401u, 0xac425bfdu,
                                                0xf65e27fau, 0xccc627f7u, 0xd66282b7u, 0x8c1258acu};
constexpr uint32_t R_TOP = 0x73eda753u;   // r >> 224
constexpr uint32_t Q_RECIP = 0x8d54253au; // floor(2^62 / (R_TOP + 1))

struct Half {
    uint32_t nr[4], rw[4];  // this lane half's words of 2^256 - r and of r
    uint32_t hmask;         // all ones in the high half
    uint32_t k8, k16, k24;  // 2^8, 2^16, 2^24 kept opaque so that the gather stays v_mad_u64_u32
};
HB_DEV Half make_half(int h) {
    Half H;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        H.nr[j] = h ? NR_W[4 + j] : NR_W[j];
        H.rw[j] = h ? R_W[4 + j] : R_W[j];
    }
    H.hmask = h ? 0xffffffffu : 0u;
    uint32_t a = 1u << 8, b = 1u << 16, c = 1u << 24;
    asm volatile("" : "+s"(a), "+s"(b), "+s"(c));
    H.k8 = a, H.k16 = b, H.k24 = c;
    return H;
}
// v_permlane32_swap vdst, src0 exchanges lanes 32..63 of vdst with lanes 0..31 of src0; with both = x the first
// result is the low half's value in every lane, the second the high half's
HB_DEV uint32_t low_bcast(uint32_t x) { return (uint32_t)__builtin_amdgcn_permlane32_swap(x, x, false, false)[0]; }
HB_DEV uint32_t high_bcast(uint32_t x) { return (uint32_t)__builtin_amdgcn_permlane32_swap(x, x, false, false)[1]; }

// 16 digit sums (non-negative, < 2^29, weight 2^(8 b)) -> 4 words + carry out of this half
HB_DEV void gather(const v16i& acc, uint32_t (&W)[4], uint32_t& cout, const Half& H) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        c += (uint32_t)acc[4 * j];
        c = (uint64_t)(uint32_t)acc[4 * j + 1] * H.k8 + c;
        c = (uint64_t)(uint32_t)acc[4 * j + 2] * H.k16 + c;
        c = (uint64_t)(uint32_t)acc[4 * j + 3] * H.k24 + c;
        W[j] = (uint32_t)c;
        c >>= 32;
    }
    cout = (uint32_t)c;
}
// U = S + q (2^256 - r), S given as the un-rippled halves (W, cg).  On return the high half holds words 4..7 of U and
// `top` = everything above 2^256; the low half words 0..3.
HB_DEV void add_q_nr(uint32_t q, const uint32_t (&W)[4], uint32_t cg, uint32_t (&U)[4], uint32_t& top, const Half& H) {
    uint64_t a = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a = (uint64_t)q * H.nr[j] + a;
        a += W[j];
        U[j] = (uint32_t)a;
        a >>= 32;
    }
    top = cg + (uint32_t)a;
    const uint32_t cin = low_bcast(top) & H.hmask;
    uint64_t t = (uint64_t)U[0] + cin;
    U[0] = (uint32_t)t;
#pragma unroll
    for (int j = 1; j < 4; ++j) {
        t = (t >> 32) + U[j];
        U[j] = (uint32_t)t;
    }
    top += (uint32_t)(t >> 32);
}
// nonzero in some lane of the pair iff  sum != ys (mod r);  ys = this half's 4 words of the claimed value (canonical)
HB_DEV uint32_t verify_tile(const v16i& acc, const v4i& ys, const Half& H) {
    uint32_t W[4], cg, U[4], top;
    gather(acc, W, cg, H);
    const uint32_t q = low_bcast(W[0] - (uint32_t)ys[0]);
    add_q_nr(q, W, cg, U, top, H);
    uint32_t bad = (U[0] ^ (uint32_t)ys[0]) | (U[1] ^ (uint32_t)ys[1]) | (U[2] ^ (uint32_t)ys[2]) | (U[3] ^ (uint32_t)ys[3]);
    bad |= (top ^ q) & H.hmask;
    return bad;
}
// canonical residue of the digit sums: this half's 4 words in Rw
HB_DEV void reduce_tile(const v16i& acc, uint32_t (&Rw)[4], const Half& H) {
    uint32_t W[4], cg, top;
    gather(acc, W, cg, H);
    // the high half estimates the quotient from its (un-rippled, so never too large) top 49 bits
    const uint32_t xq = (cg << 15) | (W[3] >> 17);
    const uint32_t q = high_bcast(__umulhi(xq, Q_RECIP) >> 13);
    add_q_nr(q, W, cg, Rw, top, H);
    // exact when word 8 cancels (R = S - q r fits 256 bits) and R's top word is below r's
    const bool fast = H.hmask == 0 || (top == q && Rw[3] < R_TOP);
    if (__builtin_expect(__any(!fast) != 0, 0)) {
        // e = what is left above 2^256 (0 <= e, small); subtract r while e 2^256 + R >= r
        uint32_t e = high_bcast(top - q);
        for (int it = 0; it < 4; ++it) {
            uint32_t D[4];
            uint64_t b = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint64_t d = (uint64_t)Rw[j] - H.rw[j] - b;
                D[j] = (uint32_t)d;
                b = (d >> 32) & 1;
            }
            const uint32_t bin = low_bcast((uint32_t)b) & H.hmask;
            uint64_t d = (uint64_t)D[0] - bin;
            D[0] = (uint32_t)d;
#pragma unroll
            for (int j = 1; j < 4; ++j) {
                d = (uint64_t)D[j] - ((d >> 32) & 1);
                D[j] = (uint32_t)d;
            }
            const uint32_t bout = high_bcast((uint32_t)b + (uint32_t)((d >> 32) & 1));  // 0 or 1 (never both)
            const bool take = e >= bout;  // e 2^256 + R - r >= 0
            if (take) {
#pragma unroll
                for (int j = 0; j < 4; ++j) Rw[j] = D[j];
                e -= bout;
            }
        }
    }
}

HB_DEV v4i flip(v4i x) {
    x[0] ^= 0x80808080, x[1] ^= 0x80808080, x[2] ^= 0x80808080, x[3] ^= 0x80808080;
    return x;
}

// the M MFMAs of one output row for CG tiles; the A operand (table slab) is read from LDS three slabs ahead of its use
template <int M, int CG>
HB_DEV void mfma_row(const uint8_t* tab_lane, const v4i (&data)[CG][M], v16i (&acc)[CG]) {
    constexpr int D = 3;
    v4i av[D];
#pragma unroll
    for (int i = 0; i < D - 1 && i < M; ++i) av[i] = *reinterpret_cast<const v4i*>(tab_lane + i * 1024);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < M; ++i) {
        if (i + D - 1 < M) av[(i + D - 1) % D] = *reinterpret_cast<const v4i*>(tab_lane + (i + D - 1) * 1024);
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) acc[cg] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[i % D], data[cg][i], acc[cg], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);  // hipcc otherwise sinks every read to just before its MFMA
    }
}

// Streams one table row (ROWB bytes) from global memory into an LDS buffer with all 256 threads: issue() starts the
// loads, commit() writes them once the row's MFMAs are done.
template <int ROWB>
struct RowStage {
    static constexpr int PIECES = ROWB / 16;
    static constexpr int PF = (PIECES + 255) / 256;
    v4i r[PF];
    HB_DEV void issue(const uint8_t* __restrict__ src) {
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int p = (int)threadIdx.x + 256 * k;
            if (p < PIECES) r[k] = *reinterpret_cast<const v4i*>(src + (size_t)p * 16);
        }
    }
    HB_DEV void commit(uint8_t* dst) const {
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int p = (int)threadIdx.x + 256 * k;
            if (p < PIECES) *reinterpret_cast<v4i*>(dst + (size_t)p * 16) = r[k];
        }
    }
};

struct MfmaRecoverArgs {
    const uint8_t* evals;   // sender rows, canonical 32-byte elements; row s at evals + rows[s] * row_stride * 32
    size_t G;
    size_t row_stride;      // elements
    RowsArg rows;
    int needed;             // d + t + 1
    const uint8_t* table;   // (needed - M) verify rows, then the output rows (M, or 1 for P(0) only)
    uint32_t* out;          // [G][M] or [G]
    uint32_t* ncoeffs;
    uint8_t* status;
    uint32_t* flagged;
    uint32_t* counters;
    uint32_t* summary;
};

// One workgroup = 4 waves; a wave owns CG tiles of 32 chunks.
template <int M, int CG, bool P0_ONLY>
__global__ __launch_bounds__(256, 2) void k_mfma_recover(MfmaRecoverArgs a) {
    static_assert(M <= 16, "quotient estimate assumes the sum stays below 2^273");
    constexpr int ROWB = M * 1024 + 128;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];  // 2 * ROWB
    if (blockIdx.x == 0 && threadIdx.x < 4) a.summary[threadIdx.x] = threadIdx.x == 2 ? 0xffffffffu : 0u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
    const Half H = make_half(h);
    const size_t g0 = ((size_t)blockIdx.x * 4 + wave) * (32 * CG);
    const int nv = a.needed - M, nrows = nv + (P0_ONLY ? 1 : M);
    constexpr int OW = P0_ONLY ? 1 : M;

    RowStage<ROWB> stage;
    stage.issue(a.table);
    size_t g[CG];
    bool live[CG];
    v4i data[CG][M];
#pragma unroll
    for (int cg = 0; cg < CG; ++cg) {
        const size_t gi = g0 + (size_t)cg * 32 + c;
        live[cg] = gi < a.G;
        g[cg] = live[cg] ? gi : a.G - 1;
#pragma unroll
        for (int i = 0; i < M; ++i)
            data[cg][i] = flip(*reinterpret_cast<const v4i*>(a.evals + ((size_t)a.rows[i] * a.row_stride + g[cg]) * 32 + 16 * h));
    }
    stage.commit(lds);
    __syncthreads();

    uint32_t bad[CG];
#pragma unroll
    for (int cg = 0; cg < CG; ++cg) bad[cg] = 0;
    bool okc[CG];
#pragma unroll
    for (int cg = 0; cg < CG; ++cg) okc[cg] = true;

    for (int r = 0; r < nrows; ++r) {
        const uint8_t* cur = lds + (size_t)(r & 1) * ROWB;
        if (r + 1 < nrows) stage.issue(a.table + (size_t)(r + 1) * ROWB);
        v4i ys[CG];
        if (r < nv) {
#pragma unroll
            for (int cg = 0; cg < CG; ++cg)
                ys[cg] = *reinterpret_cast<const v4i*>(a.evals + ((size_t)a.rows[M + r] * a.row_stride + g[cg]) * 32 + 16 * h);
        }
        v16i acc[CG];
        {
            const v4i* bp = reinterpret_cast<const v4i*>(cur + M * 1024 + h * 64);
            const v4i b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
            v16i bias;
#pragma unroll
            for (int k = 0; k < 4; ++k) bias[k] = b0[k], bias[4 + k] = b1[k], bias[8 + k] = b2[k], bias[12 + k] = b3[k];
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) acc[cg] = bias;
        }
        mfma_row<M, CG>(cur + lane * 16, data, acc);
        if (r < nv) {
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) bad[cg] |= verify_tile(acc[cg], ys[cg], H);
            if (r == nv - 1) {
                // verdict per chunk: both halves of the lane pair must agree on every verify row
#pragma unroll
                for (int cg = 0; cg < CG; ++cg) {
                    const unsigned long long m = __ballot(bad[cg] != 0);
                    const uint32_t m32 = (uint32_t)m | (uint32_t)(m >> 32);
                    okc[cg] = ((m32 >> c) & 1u) == 0;
                    const bool flag = live[cg] && !okc[cg] && h == 0;
                    const unsigned long long fm = __ballot(flag);
                    if (fm != 0) {
                        const int leader = __ffsll((long long)fm) - 1;
                        uint32_t base = 0;
                        if (lane == leader) base = atomicAdd(a.counters, (uint32_t)__popcll(fm));
                        base = __shfl(base, leader);
                        if (flag) a.flagged[base + __popcll(fm & ((1ull << lane) - 1ull))] = (uint32_t)g[cg];
                    }
                    if (live[cg] && h == 0) {
                        if (a.status) a.status[g[cg]] = okc[cg] ? 0 : 0xff;  // 0xff: pending, rewritten by the fallback kernels
                        if (a.ncoeffs && okc[cg]) a.ncoeffs[g[cg]] = M;
                    }
                }
            }
        } else {
            const int k = r - nv;
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) {
                uint32_t Rw[4];
                reduce_tile(acc[cg], Rw, H);
                if (live[cg] && okc[cg])
                    *reinterpret_cast<uint4*>(a.out + ((g[cg] * OW + k) * 8 + 4 * h)) = make_uint4(Rw[0], Rw[1], Rw[2], Rw[3]);
            }
        }
        if (r + 1 < nrows) stage.commit(lds + (size_t)((r + 1) & 1) * ROWB);
        __syncthreads();
    }
    if (nv == 0) {
        // no verify rows (needed == M): every chunk is accepted as it stands
#pragma unroll
        for (int cg = 0; cg < CG; ++cg)
            if (live[cg] && h == 0) {
                if (a.status) a.status[g[cg]] = 0;
                if (a.ncoeffs) a.ncoeffs[g[cg]] = M;
            }
    }
}

struct MfmaEvalArgs {
    const uint8_t* x;      // [G][M] chunk-major canonical elements
    size_t G;
    const uint8_t* table;  // n output rows
    int n;
    uint8_t* y;            // [n][ystride] party-major
    size_t ystride;        // elements between consecutive output rows
};
template <int M, int CG>
__global__ __launch_bounds__(256, 2) void k_mfma_eval(MfmaEvalArgs a) {
    static_assert(M <= 16, "quotient estimate assumes the sum stays below 2^273");
    constexpr int ROWB = M * 1024 + 128;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
    const Half H = make_half(h);
    const size_t g0 = ((size_t)blockIdx.x * 4 + wave) * (32 * CG);
    RowStage<ROWB> stage;
    stage.issue(a.table);
    size_t g[CG];
    bool live[CG];
    v4i data[CG][M];
#pragma unroll
    for (int cg = 0; cg < CG; ++cg) {
        const size_t gi = g0 + (size_t)cg * 32 + c;
        live[cg] = gi < a.G;
        g[cg] = live[cg] ? gi : a.G - 1;
#pragma unroll
        for (int i = 0; i < M; ++i) data[cg][i] = flip(*reinterpret_cast<const v4i*>(a.x + (g[cg] * M + i) * 32 + 16 * h));
    }
    stage.commit(lds);
    __syncthreads();
    for (int r = 0; r < a.n; ++r) {
        const uint8_t* cur = lds + (size_t)(r & 1) * ROWB;
        if (r + 1 < a.n) stage.issue(a.table + (size_t)(r + 1) * ROWB);
        v16i acc[CG];
        {
            const v4i* bp = reinterpret_cast<const v4i*>(cur + M * 1024 + h * 64);
            const v4i b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
            v16i bias;
#pragma unroll
            for (int k = 0; k < 4; ++k) bias[k] = b0[k], bias[4 + k] = b1[k], bias[8 + k] = b2[k], bias[12 + k] = b3[k];
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) acc[cg] = bias;
        }
        mfma_row<M, CG>(cur + lane * 16, data, acc);
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) {
            uint32_t Rw[4];
            reduce_tile(acc[cg], Rw, H);
            if (live[cg])
                *reinterpret_cast<uint4*>(a.y + ((size_t)r * a.ystride + g[cg]) * 32 + 16 * h) = make_uint4(Rw[0], Rw[1], Rw[2], Rw[3]);
        }
        if (r + 1 < a.n) stage.commit(lds + (size_t)((r + 1) & 1) * ROWB);
        __syncthreads();
    }
}

}  // namespace mf
}  // namespace hbmpc
