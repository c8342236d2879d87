// kernels_mfma_lab.hpp -- laboratory forms of k_mfma_rows (csrc/kernels_mfma.hpp), for the microbenchmarks under tools/ only;
// nothing here is compiled into the library.
//   k_mfma_rows_lab<M, CG, WAVES, NR, ABL, PIPE>: the production kernel's text plus
//     ABL   timing-only ablations: 1 = no epilogue arithmetic (loads and stores stay), 2 = no MFMAs (every loaded word is folded
//           into the accumulator, so the loads stay), 4 = every tile re-reads the first tiles (inputs stay in L2), nothing stored
//     PIPE  the software-pipelined row loop of round 3 (bit-exact, slower; profiles/r03_mfma_rows_pipelined*.txt)
//   the staged epilogue (RtEpi, rt_epi_stage, ...) is shared with tools/kernels_mfma_rt.hpp
// Before quoting an ablation's time, check that its instance issues the loads of the full kernel:
//   python3 tools/count_loads.py <file.hip> [-D...]   (global_load / buffer_load counts per kernel instance from hipcc -S)
#pragma once
#include "../mpc-protocols_amd/csrc/kernels_mfma.hpp"

namespace hbmpc {
namespace mf {

// ---- the epilogue of kernels_mfma.hpp in stages: the pieces a software-pipelined row loop deals out over the gaps between MFMAs ----
// The vector epilogue of a row (gather / verify_tile / reduce_tile above) cut into micro-stages, which are dealt out
// over the gaps between the M MFMAs of the next row by weight (~ instructions), so that no gap carries much more than
// its share: an in-order wave cannot make up for a gap whose vector work outlasts its MFMA.
constexpr int RT_NSTAGE = 13;
constexpr int RT_STAGE_W[RT_NSTAGE] = {4, 4, 4, 4, 4, 1, 1, 2, 3, 3, 4, 4, 3};
constexpr int rt_weight_before(int st) {
    int w = 0;
    for (int k = 0; k < st; ++k) w += RT_STAGE_W[k];
    return w;
}
template <int M>
constexpr int rt_stage_gap(int st) { return rt_weight_before(st) * M / rt_weight_before(RT_NSTAGE); }
template <int M>
constexpr int rt_bias_gap() {  // first gap after the one that reads the last sums (stage 1)
    return rt_stage_gap<M>(1) + 1 < M ? rt_stage_gap<M>(1) + 1 : M - 1;
}
// One wave per SIMD has nobody to hide a dependent instruction's latency behind (v_mad_u64_u32: ~10 cycles against ~4 of
// issue), so the instructions of a stage are independent of each other wherever the arithmetic allows (the four digit
// groups side by side), and the carry ripple is three multiply-adds t_k = hi(t_{k-1}) * 1 + Q_k, one per gap, instead
// of a v_addc chain with its wait states.
struct RtEpi {
    uint32_t p0[4], p1[4];
    uint64_t T[4], Q[4];
    uint32_t U[4];
    uint32_t q, top, res, c;  // res: verify -> nonzero when the row disagrees; output -> nonzero when the fast reduction does not hold
};
// where the result of the row being finished goes
struct RtRow {
    uint32_t voff;  // output rows: byte offset of this lane's 16 bytes inside the row (RT_OOB: no store)
    uint32_t soff;  // output rows: byte offset of the row
    uint32_t mask;  // verify rows: all ones when the lane's verdict counts
};
template <bool VERIFY>
HB_DEV void rt_epi_stage(int st, RtEpi& e, const v16i& acc, const v4i& ys, const RtRow& row, const Half& H, uint32_t one,
                         const __amdgpu_buffer_rsrc_t& rsrc, uint32_t& bad) {
    switch (st) {
        case 0:
#pragma unroll
            for (int j = 0; j < 4; ++j) e.p0[j] = ((uint32_t)acc[4 * j + 1] << 8) + (uint32_t)acc[4 * j];
            break;
        case 1:
#pragma unroll
            for (int j = 0; j < 4; ++j) e.p1[j] = ((uint32_t)acc[4 * j + 3] << 8) + (uint32_t)acc[4 * j + 2];
            break;
        case 2:
#pragma unroll
            for (int j = 0; j < 4; ++j) e.T[j] = (uint64_t)e.p1[j] * H.k16 + e.p0[j];
            break;
        case 3:
            if constexpr (VERIFY) {
                e.q = low_bcast((uint32_t)e.T[0] - (uint32_t)ys[0]);  // r = 1 mod 2^32 (kernels_mfma.hpp)
            } else {
                const uint32_t xq = (uint32_t)(e.T[3] >> 17);
                e.q = high_bcast(__umulhi(xq, Q_RECIP) >> 13);
            }
            break;
        case 4:
#pragma unroll
            for (int j = 0; j < 4; ++j) e.Q[j] = (uint64_t)e.q * H.nr[j] + e.T[j];
            break;
        case 5:
            e.Q[1] = (uint64_t)(uint32_t)(e.Q[0] >> 32) * one + e.Q[1];  // < 2^64: q NR_j + T_j + 2^32 stays below it for every q that can pass (kernels_mfma.hpp)
            break;
        case 6:
            e.Q[2] = (uint64_t)(uint32_t)(e.Q[1] >> 32) * one + e.Q[2];
            break;
        case 7:
            e.Q[3] = (uint64_t)(uint32_t)(e.Q[2] >> 32) * one + e.Q[3];
            e.top = (uint32_t)(e.Q[3] >> 32);
            break;
        case 8:
            e.res = low_bcast(e.top) & H.hmask;  // carry into the high half
            break;
        case 9:
            e.U[0] = __builtin_addc((uint32_t)e.Q[0], e.res, 0u, &e.c);
            e.U[1] = __builtin_addc((uint32_t)e.Q[1], 0u, e.c, &e.c);
            break;
        case 10:
            e.U[2] = __builtin_addc((uint32_t)e.Q[2], 0u, e.c, &e.c);
            e.U[3] = __builtin_addc((uint32_t)e.Q[3], 0u, e.c, &e.c);
            e.top += e.c;
            break;
        case 11:
            if constexpr (VERIFY) {
                e.res = (e.U[0] ^ (uint32_t)ys[0]) | (e.U[1] ^ (uint32_t)ys[1]) | (e.U[2] ^ (uint32_t)ys[2]) | (e.U[3] ^ (uint32_t)ys[3]);
            } else {
                // exact when word 8 cancels and the top word is below r's (high half); the rest is repaired after the chain
                e.res = (H.hmask != 0 && (e.top != e.q || e.U[3] >= R_TOP)) ? 1u : 0u;
            }
            break;
        default:
            if constexpr (VERIFY) {
                bad |= (e.res | ((e.top ^ e.q) & H.hmask)) & row.mask;
            } else {
                v4i val;
                val[0] = (int)e.U[0], val[1] = (int)e.U[1], val[2] = (int)e.U[2], val[3] = (int)e.U[3];
                __builtin_amdgcn_raw_buffer_store_b128(val, rsrc, (int)row.voff, (int)row.soff, 0);
            }
            break;
    }
}
// the rare tail of reduce_tile: conditional subtractions where the fast path does not hold (wave-uniform branch by the caller)
HB_DEV void rt_reduce_slow(RtEpi& e, const Half& H) {
    uint32_t rw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rw[j] = H.hmask ? R_W[4 + j] : R_W[j];
    uint32_t ex = high_bcast(e.top - e.q);
    for (int it = 0; it < 5; ++it) {
        uint32_t Dw[4];
        uint64_t b = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint64_t d = (uint64_t)e.U[j] - rw[j] - b;
            Dw[j] = (uint32_t)d;
            b = (d >> 32) & 1;
        }
        const uint32_t bin = low_bcast((uint32_t)b) & H.hmask;
        uint64_t d = (uint64_t)Dw[0] - bin;
        Dw[0] = (uint32_t)d;
#pragma unroll
        for (int j = 1; j < 4; ++j) {
            d = (uint64_t)Dw[j] - ((d >> 32) & 1);
            Dw[j] = (uint32_t)d;
        }
        const uint32_t bout = high_bcast((uint32_t)b + (uint32_t)((d >> 32) & 1));
        const bool take = ex >= bout;
        if (take) {
#pragma unroll
            for (int j = 0; j < 4; ++j) e.U[j] = Dw[j];
            ex -= bout;
        }
    }
}

// NR > 0: every role of the launch has at most NR rows and the row loop is unrolled NR times with a compile-time trip
// count -- hipcc can then count the stores issued after the next tile's loads and wait with vmcnt(#stores) at the tile
// boundary; with a run-time trip count it waits for vmcnt(0), i.e. for every store of the tile to complete.
// ABL: timing-only ablations, instantiated by tools/ubench_mfma.hip alone (the library's instances have ABL = 0 and none
// of that code): 1 = no epilogue arithmetic (loads and stores stay), 2 = no MFMAs (the inputs are folded into the accumulator
// with one XOR each, so their loads stay), 4 = every tile re-reads the first tiles (inputs stay in L2) and nothing is stored.
template <int M, int CG, int WAVES, int NR = 0, int ABL = 0, bool PIPE = false>
__global__ __launch_bounds__(64 * WAVES) void k_mfma_rows_lab(MfmaRowsArgs a) {
    static_assert(M <= 15, "digit sums must stay below 0xff0000 (tables_mfma.hpp) and the sum below 2^273");
    constexpr int ROWB = M * 1024 + 128;
    constexpr int NT = 64 * WAVES;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];  // role.nrows * ROWB
    if (a.summary && !a.direct && blockIdx.x == 0 && threadIdx.x < 4) a.summary[threadIdx.x] = threadIdx.x == 2 ? 0xffffffffu : 0u;
    const int blk8 = (int)blockIdx.x >> 3, role_id = a.blk_role[blk8];
    const int wg_in_role = (int)a.blk_idx[blk8] * 8 + ((int)blockIdx.x & 7);
    MfmaRole role = a.role[0];
    int role_wgs = a.role_nwg[0];
#pragma unroll
    for (int k = 1; k < MF_MAX_ROLES; ++k)
        if (k == role_id) role = a.role[k], role_wgs = a.role_nwg[k];
    {
        const uint8_t* src = a.table + (size_t)role.row0 * ROWB;
        const int pieces = role.nrows * (ROWB / 16);
        for (int p = threadIdx.x; p < pieces; p += NT)
            *reinterpret_cast<v4i*>(lds + (size_t)p * 16) = *reinterpret_cast<const v4i*>(src + (size_t)p * 16);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
    const Half H = make_half(h);
    const size_t ntiles = (a.G + 32 * CG - 1) / (32 * CG);
    const int nver = role.row0 < a.nv ? (a.nv - role.row0 < role.nrows ? a.nv - role.row0 : role.nrows) : 0;  // verify rows of this role
    // Addresses are a wave-uniform 64-bit row base plus a 32-bit lane offset (the host keeps G * 32 * max(M, out
    // width) below 2^32).
    // HBM latency: a wave that loads the M input rows of its tile and then computes on them for ~20 us leaves too few
    // bytes in flight per CU (measured: +0.145 ms on config 3 over the same kernel with its inputs in L2,
    // profiles/r02_ubench_mfma_v3_resident_ablation.txt).  So a wave keeps TWO input register sets and the tile loop is
    // unrolled by two: at the start of a tile all M loads of its NEXT tile are issued into the other set, a whole tile
    // ahead of their use.  (Tried and dropped: spreading those loads over the output rows -- register indices that
    // depend on the row index made hipcc peel the loop and copy the set at every merge; pulling the next tile into L2
    // with LDS-DMA loads into a scratch slot -- vmcnt is in-order, so the next wait for a claimed value then also waits
    // for the prefetch issued just before it: 0.44 ms instead of 0.36.)
    const size_t tstep = (size_t)role_wgs * WAVES;
    const uint32_t in_lane_stride = a.in_chunk_major ? M * 32u : 32u;
    auto tile_chunks = [&](size_t t, uint32_t (&gg)[CG]) {
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) {
            const size_t gi = ((ABL & 4) ? (t & 63) * CG + cg : t * CG + cg) * 32 + c;
            gg[cg] = (uint32_t)(gi < a.G ? gi : a.G - 1);
        }
    };
    auto load_inputs = [&](size_t t, v4i (&dst)[CG][M]) {
        uint32_t gg[CG];
        tile_chunks(t, gg);
#pragma unroll
        for (int i = 0; i < M; ++i) {
            const uint8_t* base = a.in_chunk_major ? a.in + (size_t)i * 32 : a.in + (size_t)a.rows[i] * a.row_stride * 32;
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) dst[cg][i] = *reinterpret_cast<const v4i*>(base + (gg[cg] * in_lane_stride + 16u * h));
        }
    };
    // the role that owns the verify rows (all of them: the host never splits them) gives the verdict per chunk;
    // with no verify rows at all (needed == M) the role of table row 0 accepts every chunk
    auto give_verdict = [&](const uint32_t (&bad)[CG], const uint32_t (&g)[CG], const bool (&live)[CG]) {
        if (a.status != nullptr || a.flagged != nullptr) {
            if (nver > 0 || (a.nv == 0 && role.row0 == 0)) {
#pragma unroll
                for (int cg = 0; cg < CG; ++cg) {
                    const unsigned long long m = __ballot(bad[cg] != 0);
                    const uint32_t m32 = (uint32_t)m | (uint32_t)(m >> 32);
                    const bool ok = ((m32 >> c) & 1u) == 0;
                    const bool flag = live[cg] && !ok && h == 0;
                    const unsigned long long fm = __ballot(flag);
                    if (fm != 0 && a.direct) {  // count_failures: chunks ascend with the lane
                        if (lane == __ffsll((long long)fm) - 1) {
                            atomicAdd(a.counters, (uint32_t)__popcll(fm));
                            atomicMax(a.counters + 1, 0xffffffffu - g[cg]);
                            __threadfence();
                        }
                    } else if (fm != 0) {
                        const int leader = __ffsll((long long)fm) - 1;
                        uint32_t base = 0;
                        if (lane == leader) base = atomicAdd(a.counters, (uint32_t)__popcll(fm));
                        base = __shfl(base, leader);
                        const size_t slot = (size_t)base + __popcll(fm & ((1ull << lane) - 1ull));
                        if (flag && slot < a.G) a.flagged[slot] = g[cg];  // the list has G entries (handoff_count)
                    }
                    if (live[cg] && h == 0) {
                        if (a.status) a.status[g[cg]] = ok ? 0 : a.direct ? (uint8_t)DecodingError : 0xff;  // 0xff: pending, rewritten by the fallback kernels
                        if (a.ncoeffs && (ok || a.direct)) a.ncoeffs[g[cg]] = ok ? M : 0;
                    }
                }
            }
        }
    };
    // one tile from the register set `data` (raw bytes on entry; sign-flipped in place)
    auto process_tile = [&](size_t t, v4i (&data)[CG][M]) {
        uint32_t g[CG];
        bool live[CG];
        tile_chunks(t, g);
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) {
            live[cg] = (((ABL & 4) ? (t & 63) * CG + cg : t * CG + cg) * 32 + c) < a.G;
#pragma unroll
            for (int i = 0; i < M; ++i) data[cg][i] = flip(data[cg][i]);
        }
        auto load_ys = [&](int r, v4i (&ys)[CG]) {  // claimed values of verify row r (table row index)
            uint32_t ri = (uint32_t)a.rows[M + r];
            asm volatile("" : "+s"(ri));  // recomputed at every use: hoisted out of the tile loop, the row bases of an
                                          // unrolled row loop (NR > 0) would take two SGPRs each and spill the scalar file
            const uint8_t* base = a.in + (size_t)ri * a.row_stride * 32;
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) ys[cg] = *reinterpret_cast<const v4i*>(base + (g[cg] * 32u + 16u * h));
        };
        v4i ys_cur[CG], ys_next[CG];
        if (nver > 0) load_ys(role.row0, ys_cur);
        uint32_t bad[CG];
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) bad[cg] = 0;
#pragma unroll
        for (int r = 0; r < (NR > 0 ? NR : role.nrows); ++r) {
            if (NR > 0 && r >= role.nrows) break;
            const uint8_t* cur = lds + (size_t)r * ROWB;
            const int rho = role.row0 + r;
            if (r + 1 < nver) load_ys(rho + 1, ys_next);
            v16i acc[CG];
            {
                const v4i* bp = reinterpret_cast<const v4i*>(cur + M * 1024 + h * 64);
                const v4i b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
                v16i bias;
#pragma unroll
                for (int k = 0; k < 4; ++k) bias[k] = b0[k], bias[4 + k] = b1[k], bias[8 + k] = b2[k], bias[12 + k] = b3[k];
#pragma unroll
                for (int cg = 0; cg < CG; ++cg) acc[cg] = bias;
                if (!(ABL & 2)) mfma_row<M, CG>(cur + lane * 16, data, acc);
                else {  // the inputs stay live (their loads must not be optimised away with the MFMAs)
#pragma unroll
                    for (int cg = 0; cg < CG; ++cg)
#pragma unroll
                        for (int i = 0; i < M; ++i) acc[cg][i & 15] ^= data[cg][i][(r + i) & 3];
                }
            }
            if ((ABL & 1) && r < nver) {
#pragma unroll
                for (int cg = 0; cg < CG; ++cg) bad[cg] |= (uint32_t)acc[cg][0] & (uint32_t)acc[cg][7] & (uint32_t)ys_cur[cg][0] & 0x80000000u;  // digit sums are < 2^24
                if (r + 1 < nver) {
#pragma unroll
                    for (int cg = 0; cg < CG; ++cg) ys_cur[cg] = ys_next[cg];
                }
            } else if (r < nver) {
#pragma unroll
                for (int cg = 0; cg < CG; ++cg) bad[cg] |= verify_tile(acc[cg], ys_cur[cg], H) & (ABL ? 0u : ~0u);
                if (r + 1 < nver) {
#pragma unroll
                    for (int cg = 0; cg < CG; ++cg) ys_cur[cg] = ys_next[cg];
                }
            } else {
                uint32_t k32 = (uint32_t)(rho - a.nv);
                asm volatile("" : "+s"(k32));  // as above: the output row base is recomputed, not kept per unrolled row
                const size_t k = k32;
#pragma unroll
                for (int cg = 0; cg < CG; ++cg) {
                    uint32_t Rw[4];
                    if (ABL & 1) Rw[0] = acc[cg][0], Rw[1] = acc[cg][5], Rw[2] = acc[cg][10], Rw[3] = acc[cg][15];  // no arithmetic, same stores
                    else reduce_tile(acc[cg], Rw, H);
                    if (a.direct) {  // one role, the verify rows are behind us: a chunk that failed them gets zeros
                        const unsigned long long mb = __ballot(bad[cg] != 0);
                        if ((((uint32_t)mb | (uint32_t)(mb >> 32)) >> c) & 1u) Rw[0] = Rw[1] = Rw[2] = Rw[3] = 0u;
                    }
                    uint8_t* qb = a.out_party_major ? a.out + k * a.out_stride * 32 : a.out + k * 32;  // wave-uniform
                    const uint32_t qo = g[cg] * (a.out_party_major ? 32u : (uint32_t)a.out_stride * 32u) + 16u * h;
                    if (live[cg] && (!(ABL & 4) || Rw[0] == 0x12345u)) *reinterpret_cast<uint4*>(qb + qo) = make_uint4(Rw[0], Rw[1], Rw[2], Rw[3]);
                }
            }
        }
        give_verdict(bad, g, live);
    };
    // PIPE: the same tile with the matrix pipe and the vector ALU overlapped INSIDE the wave: two
    // accumulators, row r's M MFMAs issue with the staged epilogue of row r - 1 between them (rt_epi_stage: gather, quotient,
    // q (2^256 - r), carry ripple, compare / store dealt out over the M gaps by weight), the A operands stream from the LDS
    // three slabs ahead across row boundaries, the accumulator a row leaves takes the next row's bias as soon as its sums are
    // gathered, and sched_barrier pins "A request, MFMA, piece of vector work" gap by gap.  For roles of ONE kind of row
    // (every role of a multi-role decode or encode; a single mixed role keeps the loop above) and a compile-time row count.
    auto process_tile_pipe = [&](auto verify_c, auto nrows_c, size_t t, v4i (&data)[CG][M], const __amdgpu_buffer_rsrc_t& rsrc) {
        constexpr bool VERIFY = decltype(verify_c)::value;
        constexpr int NROWS = decltype(nrows_c)::value;  // the role's row count at compile time: the row loop below has no exit
        static_assert(CG == 1, "the pipelined row loop walks one tile");
        constexpr int D = 2;
        uint32_t g[CG];
        bool live[CG];
        tile_chunks(t, g);
        live[0] = (t * 32 + c) < a.G;
#pragma unroll
        for (int i = 0; i < M; ++i) data[0][i] = flip(data[0][i]);
        uint32_t one = 1u;
        asm volatile("" : "+s"(one));
        RtRow row;
        row.mask = live[0] ? ~0u : 0u;
        row.voff = live[0] ? g[0] * (a.out_party_major ? 32u : (uint32_t)a.out_stride * 32u) + 16u * (uint32_t)h : RT_OOB;
        row.soff = 0;
        auto claimed = [&](int r) {  // claimed values of verify row r of this role
            const uint32_t ri = (uint32_t)__builtin_amdgcn_readfirstlane(a.rows[M + role.row0 + r]);
            return *reinterpret_cast<const v4i*>(a.in + (size_t)ri * a.row_stride * 32 + (g[0] * 32u + 16u * h));
        };
        auto bias_of = [&](int r, v16i& acc) {
            const v4i* bp = reinterpret_cast<const v4i*>(lds + (size_t)r * ROWB + M * 1024 + h * 64);
            const v4i b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = b0[k], acc[4 + k] = b1[k], acc[8 + k] = b2[k], acc[12 + k] = b3[k];
        };
        auto slab = [&](int n) {  // A operand of MFMA n of the tile (row n / M, input n % M)
            return *reinterpret_cast<const v4i*>(lds + (size_t)(n / M) * ROWB + (n % M) * 1024 + lane * 16);
        };
        v16i accA, accB;
        bias_of(0, accA);
        // claimed values: two sets; row r + 1's are requested late in row r, into the set row r - 1's epilogue has finished with
        v4i ys[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        if constexpr (VERIFY) ys[0] = claimed(0);
        v4i av[D];
#pragma unroll
        for (int n = 0; n < D - 1; ++n) av[n] = slab(n);
        uint32_t bad[CG] = {0};
        RtEpi e;
#pragma unroll
        for (int r = 0; r < NROWS; ++r) {
            v16i& cur = (r & 1) ? accB : accA;
            v16i& prev = (r & 1) ? accA : accB;
            if constexpr (!VERIFY) {
                const uint32_t k32 = (uint32_t)__builtin_amdgcn_readfirstlane(role.row0 + r - 1 - a.nv);  // the row being finished in this row's gaps
                row.soff = a.out_party_major ? k32 * (uint32_t)a.out_stride * 32u : k32 * 32u;
            }
#pragma unroll
            for (int i = 0; i < M; ++i) {
                const int n = r * M + i;
                if (n + D - 1 < NROWS * M) av[(n + D - 1) % D] = slab(n + D - 1);
                cur = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[n % D], data[0][i], cur, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (r > 0) {
#pragma unroll
                    for (int st = 0; st < RT_NSTAGE; ++st)
                        if (rt_stage_gap<M>(st) == i) rt_epi_stage<VERIFY>(st, e, prev, ys[(r - 1) & 1], row, H, one, rsrc, bad[0]);
                }
                if (i == rt_bias_gap<M>() && r + 1 < NROWS) bias_of(r + 1, prev);
                if (VERIFY && i == M - 1 && r + 1 < NROWS) ys[(r + 1) & 1] = claimed(r + 1);  // row r - 1's epilogue (which read this set) is done
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (!VERIFY) {
                if (r > 0 && __builtin_expect(__any(e.res != 0) != 0, 0)) {
                    rt_reduce_slow(e, H);
                    v4i val;
                    val[0] = (int)e.U[0], val[1] = (int)e.U[1], val[2] = (int)e.U[2], val[3] = (int)e.U[3];
                    __builtin_amdgcn_raw_buffer_store_b128(val, rsrc, (int)row.voff, (int)row.soff, 0);
                }
            }
        }
        {   // the last row's epilogue, on its own
            constexpr int r = NROWS - 1;
            if constexpr (!VERIFY) {
                const uint32_t k32 = (uint32_t)(role.row0 + r - a.nv);
                row.soff = a.out_party_major ? k32 * (uint32_t)a.out_stride * 32u : k32 * 32u;
            }
            const v16i& last = (r & 1) ? accB : accA;
#pragma unroll
            for (int st = 0; st < RT_NSTAGE; ++st) rt_epi_stage<VERIFY>(st, e, last, ys[r & 1], row, H, one, rsrc, bad[0]);
            if constexpr (!VERIFY) {
                if (__builtin_expect(__any(e.res != 0) != 0, 0)) {
                    rt_reduce_slow(e, H);
                    v4i val;
                    val[0] = (int)e.U[0], val[1] = (int)e.U[1], val[2] = (int)e.U[2], val[3] = (int)e.U[3];
                    __builtin_amdgcn_raw_buffer_store_b128(val, rsrc, (int)row.voff, (int)row.soff, 0);
                }
            }
        }
        if constexpr (VERIFY) give_verdict(bad, g, live);
    };
    v4i setA[CG][M], setB[CG][M];
    size_t t = (size_t)wg_in_role * WAVES + wave;
    if (t < ntiles) load_inputs(t, setA);
    // which row loop: the pipelined one for a role of one kind of row (rows ordered verify-first: nver == nrows or nver == 0)
    [[maybe_unused]] const bool pure_verify = nver == role.nrows && nver > 0, pure_output = nver == 0;
    [[maybe_unused]] __amdgpu_buffer_rsrc_t out_rsrc;
    if constexpr (PIPE && NR > 0 && CG == 1) {
        const size_t bytes = a.out_party_major ? ((size_t)(role.row0 + role.nrows - a.nv - 1) * a.out_stride + a.G) * 32 : a.G * a.out_stride * 32;
        out_rsrc = rt_rsrc(a.out, (uint32_t)(bytes < 0xffffffe0ull ? bytes : 0xffffffe0ull));
    }
    // the tile loop: the input sets alternate, the loads of a wave's NEXT tile are issued before it starts on the current one
#define HBMPC_MF_WALK(TILE)                                                \
    while (t < ntiles) {                                                   \
        if (t + tstep < ntiles) load_inputs(t + tstep, setB);              \
        TILE(t, setA);                                                     \
        t += tstep;                                                        \
        if (t >= ntiles) break;                                            \
        if (t + tstep < ntiles) load_inputs(t + tstep, setA);              \
        TILE(t, setB);                                                     \
        t += tstep;                                                        \
    }
    if constexpr (PIPE && NR > 1 && CG == 1) {
        // launched only for roles of one kind with NR or NR - 1 rows (mf_pipe_ok): which of the four row loops this workgroup
        // runs is decided once, outside the tile loop
        using RA = std::integral_constant<int, NR>;
        using RB = std::integral_constant<int, NR - 1>;
#define HBMPC_MF_T0(tt, d) process_tile_pipe(std::true_type{}, RA{}, tt, d, out_rsrc)
#define HBMPC_MF_T1(tt, d) process_tile_pipe(std::true_type{}, RB{}, tt, d, out_rsrc)
#define HBMPC_MF_T2(tt, d) process_tile_pipe(std::false_type{}, RA{}, tt, d, out_rsrc)
#define HBMPC_MF_T3(tt, d) process_tile_pipe(std::false_type{}, RB{}, tt, d, out_rsrc)
        if (pure_verify && role.nrows == NR) {
            HBMPC_MF_WALK(HBMPC_MF_T0)
        } else if (pure_verify) {
            HBMPC_MF_WALK(HBMPC_MF_T1)
        } else if (role.nrows == NR) {
            HBMPC_MF_WALK(HBMPC_MF_T2)
        } else {
            HBMPC_MF_WALK(HBMPC_MF_T3)
        }
#undef HBMPC_MF_T0
#undef HBMPC_MF_T1
#undef HBMPC_MF_T2
#undef HBMPC_MF_T3
    } else {
        HBMPC_MF_WALK(process_tile)
    }
#undef HBMPC_MF_WALK
    if (a.direct) finish_direct(a.counters, a.summary);
}

// can the pipelined instance <.., NR, .., PIPE = true> serve this plan?  every role of one kind of row, with nr or nr - 1 rows
inline bool mf_pipe_ok(const MfmaRowsArgs& a, int nr) {
    if (a.direct || nr < 2) return false;
    for (int k = 0; k < a.nroles; ++k) {
        const MfmaRole& r = a.role[k];
        const int nver = r.row0 < a.nv ? (a.nv - r.row0 < r.nrows ? a.nv - r.row0 : r.nrows) : 0;
        if (nver != 0 && nver != r.nrows) return false;
        if (r.nrows != nr && r.nrows != nr - 1) return false;
    }
    return true;
}

}  // namespace mf
}  // namespace hbmpc
