// kernels_mfma_stream.hpp -- the matrix-core decode (kernels_mfma.hpp) with ONE kind of workgroup when the table does not fit
// the LDS: the rows that fit stay resident, the A operands of the others are fetched from the table in global memory.
//
// batch_recover_secret (robust_interpolate.rs:391-427) at config 3's shape has 10 verify rows + 11 coefficient rows of
// 11 KB each: 239 KB of table against 160 KB of LDS.  k_mfma_rows therefore runs two kinds of workgroup (the verify rows;
// the coefficient rows) and each of them reads the 11 input rows of a tile: 1.29x the compulsory HBM traffic.  Here every
// workgroup serves every row: NRES rows live in its LDS, and the 1 KB slab of each (row, input) pair beyond them arrives as
// one coalesced 16-byte-per-lane global load per MFMA.  Those slabs are the same few dozen KB for every wave on the chip, so
// they are L2 hits; a wave keeps D of them in flight in a register ring.  vmcnt retires in order, so the streamed rows come
// LAST in a tile: by then the tile's own HBM requests (the next tile's inputs, the claimed values of the verify rows) are
// long back and a wait for a slab waits for nothing else.
//
// Everything inside the tile loop is unconditional (loads past the end are clamped, stores go through buffer descriptors
// that drop what must not be written) so that hipcc counts the vector memory operations between a request and its use.
#pragma once
#include <utility>
#ifndef HBMPC_MFS_ABL  // timing-only ablations of tools/ubench_mfma_stream.hip: 1 = claimed values not loaded, 2 = no output stores, 4 = every tile re-reads tile 0
#define HBMPC_MFS_ABL 0
#endif

#include "../mpc-protocols_amd/csrc/kernels_mfma.hpp"

namespace hbmpc {
namespace mf {

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): the row index of the tile walk must be a constant in every
// copy of the row body (which rows are resident, which slab of the ring an MFMA reads); a `#pragma unroll` loop of 21 such
// bodies is only partly unrolled by hipcc, which leaves a run-time row index and with it dynamically indexed registers
template <class F, int... I>
HB_DEV void mfs_static_for(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}

constexpr size_t mfs_lds_bytes(int M, int NRES, int NR) { return (size_t)NRES * M * 1024 + (size_t)NR * 128; }

// rows [0, NV): verify rows (claimed value = sender row M + r), rows [NV, NV + NO): output rows; rows [0, NRES) resident
template <int M, int WAVES, int NV, int NO, int NRES, int D, int DL = 3, int YD = 1, bool DB = true>
__global__ __launch_bounds__(64 * WAVES) void k_mfma_rows_stream(MfmaRowsArgs a) {
    static_assert(M >= 2 && M <= 15, "digit sums must stay below 0xff0000 (tables_mfma.hpp)");
    constexpr int NR = NV + NO, ROWB = M * 1024 + 128, SLABS = M * 1024, NT = 64 * WAVES;
    static_assert(NRES >= NV && NRES <= NR, "the verify rows are resident: their claimed values are the tile's last HBM requests");
    constexpr int NSL = (NR - NRES) * M;  // streamed slabs per tile
    static_assert(NSL == 0 || D <= NSL, "ring depth");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];  // NRES * SLABS, then the NR biases
    if (a.summary && !a.direct && blockIdx.x == 0 && threadIdx.x < 4) a.summary[threadIdx.x] = threadIdx.x == 2 ? 0xffffffffu : 0u;
    for (int p = threadIdx.x; p < NRES * (SLABS / 16); p += NT) {
        const int row = p / (SLABS / 16), off = p % (SLABS / 16);
        *reinterpret_cast<v4i*>(lds + (size_t)row * SLABS + off * 16) = *reinterpret_cast<const v4i*>(a.table + (size_t)row * ROWB + off * 16);
    }
    for (int p = threadIdx.x; p < NR * 8; p += NT)
        *reinterpret_cast<v4i*>(lds + (size_t)NRES * SLABS + p * 16) = *reinterpret_cast<const v4i*>(a.table + (size_t)(p >> 3) * ROWB + SLABS + (p & 7) * 16);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
    const Half H = make_half(h);
    const size_t ntiles = (a.G + 31) / 32;
    const size_t tstep = (size_t)gridDim.x * WAVES;
    const uint8_t* bias_lds = lds + (size_t)NRES * SLABS + h * 64;
    // slab n of the streamed part: one buffer load, lane offset in a VGPR, (n / M) * ROWB + (n % M) * 1024 as the scalar offset (no
    // address registers per slab: with global loads hipcc keeps 77 hoisted 64-bit addresses and spills)
    const __amdgpu_buffer_rsrc_t stream_rsrc = rt_rsrc(a.table + (size_t)NRES * ROWB, (uint32_t)((NR - NRES) * ROWB));
    auto load_inputs = [&](size_t t, v4i (&dst)[M]) {
        const size_t gi = ((HBMPC_MFS_ABL & 4) ? (t & 63) : t) * 32 + c;
        const uint32_t g = (uint32_t)(gi < a.G ? gi : a.G - 1);
#pragma unroll
        for (int i = 0; i < M; ++i) {
            uint32_t ri = (uint32_t)a.rows[i];
            asm volatile("" : "+s"(ri));  // recomputed at every use (scalar registers, as in k_mfma_rows)
            dst[i] = *reinterpret_cast<const v4i*>(a.in + (size_t)ri * a.row_stride * 32 + (g * 32u + 16u * h));
        }
    };
    const uint32_t out_bytes = (uint32_t)([&] {
        const size_t b = a.out_party_major ? a.G * 32 : a.G * a.out_stride * 32;
        return b < 0xffffffe0ull ? b : 0xffffffe0ull;
    }());
    const __amdgpu_buffer_rsrc_t status_rsrc = rt_rsrc(a.status, a.status ? (uint32_t)(a.G < 0xffffffe0ull ? a.G : 0xffffffe0ull) : 0u);
    const __amdgpu_buffer_rsrc_t ncoeffs_rsrc = rt_rsrc(a.ncoeffs, a.ncoeffs ? (uint32_t)(a.G * 4 < 0xffffffe0ull ? a.G * 4 : 0xffffffe0ull) : 0u);
    auto process_tile = [&](size_t t, v4i (&data)[M]) {
        const size_t gi = t * 32 + c;
        const bool live = gi < a.G;
        const uint32_t g = (uint32_t)(live ? gi : a.G - 1);
        const uint32_t qo = live ? g * (a.out_party_major ? 32u : (uint32_t)a.out_stride * 32u) + 16u * h : RT_OOB;
#pragma unroll
        for (int i = 0; i < M; ++i) data[i] = flip(data[i]);
        auto load_ys = [&](int r) {  // claimed values of verify row r
            if constexpr ((HBMPC_MFS_ABL & 1) != 0) return data[r % M];
            uint32_t ri = (uint32_t)a.rows[M + r];
            asm volatile("" : "+s"(ri));
            return *reinterpret_cast<const v4i*>(a.in + (size_t)ri * a.row_stride * 32 + (g * 32u + 16u * h));
        };
        auto stream_slab = [&](int n) {
#ifdef HBMPC_MFS_TIMING_ONLY_LDS_SLABS  // tools/ubench_mfma_stream.hip: what the streamed slabs cost (wrong operands from the LDS instead)
            return *reinterpret_cast<const v4i*>(lds + (size_t)((n / M) % NRES) * SLABS + (n % M) * 1024 + lane * 16);
#endif
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(stream_rsrc, lane * 16, (n / M) * ROWB + (n % M) * 1024, 0);
            v4i r;
            r[0] = (int)v[0], r[1] = (int)v[1], r[2] = (int)v[2], r[3] = (int)v[3];
            return r;
        };
        [[maybe_unused]] v4i ys[YD + 1];  // claimed values, requested YD rows ahead of their use
#pragma unroll
        for (int k = 0; k < YD && k < NV; ++k) ys[k] = load_ys(k);
        [[maybe_unused]] v4i sv[D];
        uint32_t bad = 0;
        bool zero_me = false;
        mfs_static_for([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            if constexpr (NSL > 0) {
                if (r == (NRES > 0 ? NRES - 1 : 0)) {  // the first D slabs are requested one row ahead of the streamed part
#pragma unroll
                    for (int n = 0; n < D; ++n) sv[n] = stream_slab(n);
                }
            }
            if constexpr (r + YD < NV) ys[(r + YD) % (YD + 1)] = load_ys(r + YD);
            v16i acc;
            {
                const v4i* bp = reinterpret_cast<const v4i*>(bias_lds + r * 128);
                const v4i b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = b0[k], acc[4 + k] = b1[k], acc[8 + k] = b2[k], acc[12 + k] = b3[k];
            }
            if (r < NRES) {  // A operands from the LDS, three slabs ahead of their use (mfma_row)
                const uint8_t* tab_lane = lds + (size_t)r * SLABS + lane * 16;
                v4i av[DL];
#pragma unroll
                for (int i = 0; i < DL - 1 && i < M; ++i) av[i] = *reinterpret_cast<const v4i*>(tab_lane + i * 1024);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < M; ++i) {
                    if (i + DL - 1 < M) av[(i + DL - 1) % DL] = *reinterpret_cast<const v4i*>(tab_lane + (i + DL - 1) * 1024);
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[i % DL], data[i], acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);  // hipcc otherwise sinks every read to just before its MFMA
                }
            } else {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < M; ++i) {
                    const int n = (r - NRES) * M + i;
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(sv[n % D], data[i], acc, 0, 0, 0);
                    if (n + D < NSL) sv[n % D] = stream_slab(n + D);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (r < NV) {
                bad |= verify_tile(acc, ys[r % (YD + 1)], H);
                asm volatile("" : "+v"(bad));  // one register per tile: left alone, hipcc re-associates the ORs of all rows into a tree and keeps every row's five terms live
                if (r == NV - 1 && a.direct) {
                    const unsigned long long mb = __ballot(bad != 0);
                    zero_me = ((((uint32_t)mb | (uint32_t)(mb >> 32)) >> c) & 1u) != 0;
                }
            } else {
                uint32_t k32 = (uint32_t)(r - NV);
                asm volatile("" : "+s"(k32));  // the output row base is recomputed, not kept per unrolled row
                uint32_t Rw[4];
                reduce_tile(acc, Rw, H);
                if (zero_me) Rw[0] = Rw[1] = Rw[2] = Rw[3] = 0u;  // direct: a chunk that failed the verify rows gets zeros
                uint8_t* qb = a.out_party_major ? a.out + (size_t)k32 * a.out_stride * 32 : a.out + (size_t)k32 * 32;  // wave-uniform
                v4i val;
                val[0] = (int)Rw[0], val[1] = (int)Rw[1], val[2] = (int)Rw[2], val[3] = (int)Rw[3];
                if (!(HBMPC_MFS_ABL & 2) || val[0] == 0x12345) __builtin_amdgcn_raw_buffer_store_b128(val, rt_rsrc(qb, out_bytes), (int)qo, 0, 0);
            }
        }, std::make_integer_sequence<int, NR>{});
        if constexpr ((HBMPC_MFS_ABL & 1) != 0) bad &= (uint32_t)a.in_chunk_major;  // timing only: zero at run time, the verify arithmetic stays
        // the verdict of the chunk (NV == 0: every chunk is accepted)
        const unsigned long long m = __ballot(bad != 0);
        const uint32_t m32 = (uint32_t)m | (uint32_t)(m >> 32);
        const bool ok = ((m32 >> c) & 1u) == 0;
        const bool flag = live && !ok && h == 0;
        const unsigned long long fm = __ballot(flag);
        if (__builtin_expect(fm != 0, 0)) {
            if (a.direct) {  // count_failures: chunks ascend with the lane
                if (lane == __ffsll((long long)fm) - 1) {
                    atomicAdd(a.counters, (uint32_t)__popcll(fm));
                    atomicMax(a.counters + 1, 0xffffffffu - g);
                    __threadfence();
                }
            } else {
                const int leader = __ffsll((long long)fm) - 1;
                uint32_t base = 0;
                if (lane == leader) base = atomicAdd(a.counters, (uint32_t)__popcll(fm));
                base = __shfl(base, leader);
                const size_t slot = (size_t)base + __popcll(fm & ((1ull << lane) - 1ull));
                if (flag && slot < a.G) a.flagged[slot] = g;  // the list has G entries (handoff_count)
            }
        }
        const bool writer = live && h == 0;
        const uint8_t st = ok ? 0 : a.direct ? (uint8_t)DecodingError : 0xff;  // 0xff: pending, rewritten by the fallback kernels
        __builtin_amdgcn_raw_buffer_store_b8(st, status_rsrc, (int)(writer ? g : RT_OOB), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(ok ? (uint32_t)M : 0u, ncoeffs_rsrc, (int)(writer && (ok || a.direct) ? g * 4u : RT_OOB), 0, 0);
    };
    size_t t = (size_t)blockIdx.x * WAVES + wave;
    if constexpr (!DB) {  // one input set: a tile's inputs are requested when the wave reaches it (more waves per SIMD instead of a set in flight)
        v4i set[M];
        for (; t < ntiles; t += tstep) {
            load_inputs(t, set);
            process_tile(t, set);
        }
        if (a.direct) finish_direct(a.counters, a.summary);
        return;
    }
    v4i setA[M], setB[M];
    if (t < ntiles) load_inputs(t, setA);
    while (t < ntiles) {
        load_inputs(t + tstep < ntiles ? t + tstep : ntiles - 1, setB);
        process_tile(t, setA);
        t += tstep;
        if (t >= ntiles) break;
        load_inputs(t + tstep < ntiles ? t + tstep : ntiles - 1, setA);
        process_tile(t, setB);
        t += tstep;
    }
    if (a.direct) finish_direct(a.counters, a.summary);
}

}  // namespace mf
}  // namespace hbmpc
