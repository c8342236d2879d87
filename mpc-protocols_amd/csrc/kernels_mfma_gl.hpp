// kernels_mfma_gl.hpp -- the constant-matrix maps of the path over Goldilocks (p = 2^64 - 2^32 + 1, the reference's small
// field: common/math/goldilocks.rs:4-13, PreprocNodesSmallField honeybadger/mod.rs:316-324) on the matrix cores.
//
//   decode  (batch_recover_secret, robust_interpolate.rs:391-427): verify rows  Y * VM  and coefficient rows  Y * BC
//   encode  (apply_vandermonde, common/share/mod.rs:50-76; compute_shares = the same map)
//
// Formulation and table layout: tables_mfma_gl.hpp.  A wave owns a tile of 32 chunks; the B operand of K-step s is 16 bytes
// per lane = TWO whole 8-byte elements of the lane's chunk (elements 4 s + 2 h and 4 s + 2 h + 1, h = lane >> 5), loaded
// as they lie in HBM and sign-flipped with one XOR per dword.  An MFMA tile carries four table rows, two per lane half,
// eight digit sums each (< 2^23): one lane finishes an element by itself -- two 4-digit gathers, a 128-bit combine and the
// Goldilocks fold 2^64 = 2^32 - 1 -- with no cross-lane step.  The whole table (a few KB) sits in LDS, so ONE workgroup
// kind serves verify and output rows alike: inputs are read once, and a call without OEC rounds is a single launch
// (kernels_recover.hpp: fail_chunk / count_failures / finish_direct).
// The vector-ALU kernels pay 8 instructions per term (231 terms per chunk at n = 31, d = t = 10); this one 18 MFMAs and
// ~250 vector instructions per 32 chunks, which leaves the 256 bytes per chunk of traffic as the bound.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fr_gold.hpp"
#include "kernels_mfma.hpp"  // v4i, v16i
#include "kernels_recover.hpp"

#ifndef HBMPC_GL_YT
#define HBMPC_GL_YT 2
#endif

namespace hbmpc {
namespace mf {

struct MfmaGlArgs {
    const uint8_t* in;     // decode: sender rows (row s at in + rows[s] * row_stride * 8); encode: x[G][m]
    size_t G;
    int in_chunk_major;
    size_t row_stride;     // elements
    RowsArg rows;          // decode: positions of the sorted senders' rows in the caller's array
    const uint8_t* table;  // tables_mfma_gl.hpp
    int m, nrows, nv;      // inputs per chunk; table rows; the first nv of them verify rows (claimed value = sender row m + r)
    uint8_t* out;          // output k of chunk g: encode (party-major) out + (k * out_stride + g) * 8, decode (chunk-major) out + (g * out_stride + k) * 8
    int out_party_major;
    size_t out_stride;
    uint32_t* ncoeffs;     // decode only (nullable): m for accepted chunks
    uint8_t* status;       // decode only (nullable)
    uint32_t* flagged;
    uint32_t* counters;
    uint32_t* summary;     // decode only (nullable): initialised by workgroup 0
    int direct;            // no OEC round exists: failing chunks fail here, the last workgroup writes the summary
};

// digits L[0..8) of one element (register block BASE of the accumulator, each < 2^23) -> canonical residue.
// Two digits fit 32 bits (v_lshl_add_u32), two such pairs one v_mad_u64_u32; the value is w0 + 2^32 w1 + 2^64 w2 with
// w2 < 2^17, and 2^64 = 2^32 - 1 (mod p) folds w2 with one shift, one subtraction and one addition.
template <int BASE>
HB_DEV uint64_t gl_finish(const v16i& acc, uint32_t k16) {
    const uint32_t p01 = ((uint32_t)acc[BASE + 1] << 8) + (uint32_t)acc[BASE], p23 = ((uint32_t)acc[BASE + 3] << 8) + (uint32_t)acc[BASE + 2];
    const uint32_t p45 = ((uint32_t)acc[BASE + 5] << 8) + (uint32_t)acc[BASE + 4], p67 = ((uint32_t)acc[BASE + 7] << 8) + (uint32_t)acc[BASE + 6];
    const uint64_t lo = (uint64_t)p23 * k16 + p01, hi = (uint64_t)p67 * k16 + p45;  // < 2^48 each; value = lo + 2^32 hi
    uint32_t cy;
    const uint32_t w0 = (uint32_t)lo;
    const uint32_t w1 = __builtin_addc((uint32_t)(lo >> 32), (uint32_t)hi, 0u, &cy);
    const uint32_t w2 = (uint32_t)(hi >> 32) + cy;  // < 2^17
    const uint64_t x = ((uint64_t)w1 << 32) | w0;
    const uint64_t t1 = ((uint64_t)w2 << 32) - w2;  // w2 (2^32 - 1) < 2^49
    uint64_t r = x + t1;
    if (r < x) r += Gold::EPS;  // wrapped once: + 2^64 = + EPS (cannot wrap again: r < 2^49 after the wrap)
    return r >= Gold::P ? r - Gold::P : r;
}

// KS = K-steps (ceil(8 m / 32)); 256 threads = 4 waves, several workgroups per CU, grid-stride over 32-chunk tiles
// waves_per_eu: with a register budget of at most 256 per lane hipcc keeps the MFMA accumulator in VGPRs; left to itself
// (a 256-thread workgroup may use 512) it puts it in AGPRs and pays 16 v_accvgpr_write + 16 v_accvgpr_read per tile of
// four rows -- a third of this kernel's vector instructions.
template <int KS, bool ENCODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_mfma_rows_gl(MfmaGlArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int NT = (a.nrows + 3) >> 2;
    const int table_bytes = NT * (KS * 1024 + 128);
    // byte offset of every sender row (the permutation times the row stride): lanes index it by THEIR rows
    uint64_t* rowoff = reinterpret_cast<uint64_t*>(lds + table_bytes);
    if (a.summary && !a.direct && blockIdx.x == 0 && threadIdx.x < 4) a.summary[threadIdx.x] = threadIdx.x == 2 ? 0xffffffffu : 0u;
    for (int p = threadIdx.x; p < table_bytes / 16; p += 256)
        *reinterpret_cast<v4i*>(lds + (size_t)p * 16) = *reinterpret_cast<const v4i*>(a.table + (size_t)p * 16);
    rowoff[threadIdx.x] = (uint64_t)a.rows[(int)threadIdx.x] * a.row_stride * 8;
    __syncthreads();
    uint32_t k16 = 1u << 16;
    asm volatile("" : "+s"(k16));  // opaque, so that the gathers stay v_mad_u64_u32
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
    const size_t ntiles = (a.G + 31) / 32, tstep = (size_t)gridDim.x * 4;
    constexpr int TR = KS * 1024 + 128;  // bytes per tile record (tables_mfma_gl.hpp)

    auto chunk_of = [&](size_t t) __attribute__((always_inline)) {
        const size_t gi = t * 32 + c;
        return gi < a.G ? gi : a.G - 1;
    };
    // what a lane reads for a tile: its two elements of every K-step, and (decode) the claimed values of its verify rows in
    // the first YT tiles -- loaded a whole tile ahead of their use.  A claimed value fetched where it is compared costs one
    // memory latency per verify tile; prefetching all of them costs registers and occupancy.  Measured for 2^20 chunks
    // (n = 31, d = t = 10 / n = 16, d = 10, t = 5), decode call: YT = 0: 82 / 60 us, YT = 2: 77 / 53, YT = 4: 96 / 74.
    constexpr int YT = HBMPC_GL_YT;  // verify rows 0 .. 4 YT - 1 are prefetched; beyond that they are loaded in place
    struct TileIn {
        uint2 x[KS][2];
        uint2 ys[YT > 0 ? YT : 1][2];
    };
    auto load_inputs = [&](size_t t, TileIn& in) __attribute__((always_inline)) {
        uint2 (&dst)[KS][2] = in.x;
        const size_t g = chunk_of(t);
        if (!ENCODE) {
#pragma unroll
            for (int mt = 0; mt < YT; ++mt)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int rho = 4 * mt + 2 * h + e;
                    const int rr = rho < a.nv ? rho : 0;  // a row that exists (nv == 0: position m, never compared)
                    if (4 * mt < a.nv) in.ys[mt][e] = *reinterpret_cast<const uint2*>(a.in + (rowoff[a.m + rr] + g * 8));
                }
        }
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int i = 4 * s + 2 * h + j;
                const int ii = i < a.m ? i : a.m - 1;
                const uint8_t* p = ENCODE ? a.in + (g * (size_t)a.m + ii) * 8 : a.in + (rowoff[ii] + g * 8);
                const uint2 v = *reinterpret_cast<const uint2*>(p);
                dst[s][j] = i < a.m ? v : make_uint2(0u, 0u);
            }
    };
    auto process_tile = [&](size_t t, const TileIn& in) __attribute__((always_inline)) {
        const uint2 (&raw)[KS][2] = in.x;
        const size_t g = chunk_of(t);
        const bool live = t * 32 + c < a.G;
        v4i data[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            data[s][0] = (int)(raw[s][0].x ^ 0x80808080u), data[s][1] = (int)(raw[s][0].y ^ 0x80808080u);
            data[s][2] = (int)(raw[s][1].x ^ 0x80808080u), data[s][3] = (int)(raw[s][1].y ^ 0x80808080u);
        }
        bool bad = false;
        // the pair verdict once the verify rows are behind this lane pair: rows ascend with the tile, so it is final when
        // an output row is reached (both halves of a tile share mt; rows 4 mt + 2 h + e < nv are verify rows)
        auto pair_bad = [&]() __attribute__((always_inline)) {
            const unsigned long long mb = __ballot(bad);
            return ((((uint32_t)mb | (uint32_t)(mb >> 32)) >> c) & 1u) != 0;
        };
        for (int mt = 0; mt < NT; ++mt) {
            v16i acc;
            {
                const v4i* bp = reinterpret_cast<const v4i*>(lds + (size_t)mt * TR + KS * 1024 + h * 64);
                const v4i b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = b0[k], acc[4 + k] = b1[k], acc[8 + k] = b2[k], acc[12 + k] = b3[k];
            }
            const uint8_t* slab = lds + (size_t)mt * TR + lane * 16;
#pragma unroll
            for (int s = 0; s < KS; ++s)
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const v4i*>(slab + (size_t)s * 1024), data[s], acc, 0, 0, 0);
            // both elements of the lane, then: verify rows of the whole tile first, then its output rows
            const int row0 = 4 * mt + 2 * h;
            const uint64_t v0 = gl_finish<0>(acc, k16), v1 = gl_finish<8>(acc, k16);
            if (!ENCODE && 4 * mt < a.nv) {  // some row of this tile is a verify row (uniform)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int rho = row0 + e;
                    if (rho < a.nv) {
                        const uint64_t val = e == 0 ? v0 : v1;
                        uint2 ys;
                        if (mt < YT) {  // prefetched with the inputs; mt is uniform, so this is a scalar select per register
                            ys = in.ys[0][e];
#pragma unroll
                            for (int q = 1; q < YT; ++q)
                                if (mt == q) ys = in.ys[q][e];
                        } else {
                            ys = *reinterpret_cast<const uint2*>(a.in + (rowoff[a.m + rho] + g * 8));
                        }
                        bad = bad || (uint32_t)val != ys.x || (uint32_t)(val >> 32) != ys.y;
                    }
                }
            }
            if (4 * mt + 3 >= a.nv) {  // some row of this tile is an output row (uniform)
                const bool zero_out = !ENCODE && a.direct && a.nv > 0 && pair_bad();
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int rho = row0 + e;
                    if (rho >= a.nv && rho < a.nrows) {
                        const uint64_t val = zero_out ? 0 : e == 0 ? v0 : v1;
                        const size_t k = (size_t)(rho - a.nv);
                        uint8_t* q = ENCODE ? a.out + (k * a.out_stride + g) * 8 : a.out + (g * a.out_stride + k) * 8;
                        if (live) *reinterpret_cast<uint2*>(q) = make_uint2((uint32_t)val, (uint32_t)(val >> 32));
                    }
                }
            }
        }
        if (!ENCODE && (a.status != nullptr || a.flagged != nullptr)) {
            const bool ok = !pair_bad();
            const bool flag = live && !ok && h == 0;
            const unsigned long long fm = __ballot(flag);
            if (fm != 0 && a.direct) {  // count_failures: chunks ascend with the lane
                if (lane == __ffsll((long long)fm) - 1) {
                    atomicAdd(a.counters, (uint32_t)__popcll(fm));
                    atomicMax(a.counters + 1, 0xffffffffu - (uint32_t)g);
                    __threadfence();
                }
            } else if (fm != 0) {
                const int leader = __ffsll((long long)fm) - 1;
                uint32_t base = 0;
                if (lane == leader) base = atomicAdd(a.counters, (uint32_t)__popcll(fm));
                base = __shfl(base, leader);
                const size_t slot = (size_t)base + __popcll(fm & ((1ull << lane) - 1ull));
                if (flag && slot < a.G) a.flagged[slot] = (uint32_t)g;  // the list has G entries (handoff_count)
            }
            if (live && h == 0) {
                if (a.status) a.status[g] = ok ? 0 : a.direct ? (uint8_t)DecodingError : 0xff;  // 0xff: pending, rewritten by the fallback kernels
                if (a.ncoeffs && (ok || a.direct)) a.ncoeffs[g] = ok ? (uint32_t)a.m : 0u;
            }
        }
    };
    // two input register sets: the next tile's loads are issued a whole tile ahead of their use
    TileIn setA, setB;
    size_t t = (size_t)blockIdx.x * 4 + wave;
    if (t < ntiles) load_inputs(t, setA);
    while (t < ntiles) {
        if (t + tstep < ntiles) load_inputs(t + tstep, setB);
        process_tile(t, setA);
        t += tstep;
        if (t >= ntiles) break;
        if (t + tstep < ntiles) load_inputs(t + tstep, setA);
        process_tile(t, setB);
        t += tstep;
    }
    if (a.direct) finish_direct(a.counters, a.summary);
}

}  // namespace mf
}  // namespace hbmpc
