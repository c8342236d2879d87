// kernels_mfma_team.hpp -- k_mfma_rows for batches too small to give every wave a tile.
//
// k_mfma_rows gives a 32-chunk tile to ONE wave, which walks the role's 10..13 table rows one after the other: ~10 us per
// tile, fine when every wave has a queue of tiles, but it is the whole kernel when there are fewer tiles than waves (a
// 2^14-chunk decode has 512 tiles for 3072 waves: 43 workgroups work, 213 fill their LDS and leave).  Here the WORKGROUP
// owns the tile and its waves share the rows (row r of the role goes to wave r mod WAVES): a tile costs one or two rows of
// latency, every workgroup of the launch has work, and the price -- each wave loads the tile's inputs for itself, out of
// the L2 -- is nothing at these sizes.  The verify verdict of a tile is the OR over the waves, through 32 flag words in
// LDS (one barrier before they are read, two around their reset; roles without verify rows -- every encode -- have none).  Same tables, same row arithmetic (kernels_mfma.hpp), same roles, same hand-off to the fallback
// kernels; results are bit-identical.
#pragma once
#include "kernels_mfma.hpp"

namespace hbmpc {
namespace mf {

// LDS: role rows, then 32 flag words.  grid = 8 * a.nblocks as for k_mfma_rows.
template <int M, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_mfma_rows_team(MfmaRowsArgs a) {
    static_assert(M <= 15, "digit sums must stay below 0xff0000 (tables_mfma.hpp)");
    constexpr int ROWB = M * 1024 + 128;
    constexpr int NT = 64 * WAVES;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    if (a.summary && !a.direct && blockIdx.x == 0 && threadIdx.x < 4) a.summary[threadIdx.x] = threadIdx.x == 2 ? 0xffffffffu : 0u;
    const int blk8 = (int)blockIdx.x >> 3, role_id = a.blk_role[blk8];
    const int wg_in_role = (int)a.blk_idx[blk8] * 8 + ((int)blockIdx.x & 7);
    MfmaRole role = a.role[0];
    int role_wgs = a.role_nwg[0];
#pragma unroll
    for (int k = 1; k < MF_MAX_ROLES; ++k)
        if (k == role_id) role = a.role[k], role_wgs = a.role_nwg[k];
    uint32_t* flags = reinterpret_cast<uint32_t*>(lds + (size_t)role.nrows * ROWB);
    {
        const uint8_t* src = a.table + (size_t)role.row0 * ROWB;
        const int pieces = role.nrows * (ROWB / 16);
        for (int p = threadIdx.x; p < pieces; p += NT)
            *reinterpret_cast<v4i*>(lds + (size_t)p * 16) = *reinterpret_cast<const v4i*>(src + (size_t)p * 16);
        if (threadIdx.x < 32) flags[threadIdx.x] = 0u;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
    const Half H = make_half(h);
    const size_t ntiles = (a.G + 31) / 32;
    const int nver = role.row0 < a.nv ? (a.nv - role.row0 < role.nrows ? a.nv - role.row0 : role.nrows) : 0;  // verify rows of this role
    const bool verdict_role = nver > 0 || (a.nv == 0 && role.row0 == 0);
    const uint32_t in_lane_stride = a.in_chunk_major ? M * 32u : 32u;

    auto chunk_of = [&](size_t t) __attribute__((always_inline)) {
        const size_t gi = t * 32 + c;
        return (uint32_t)(gi < a.G ? gi : a.G - 1);
    };
    // (prefetching a wave's claimed value with the inputs was tried: slower, 12.1 against 10.5 us for 8 000 chunks of config 3)
    auto load_inputs = [&](size_t t, v4i (&dst)[1][M]) __attribute__((always_inline)) {
        const uint32_t g = chunk_of(t);
#pragma unroll
        for (int i = 0; i < M; ++i) {
            const uint8_t* base = a.in_chunk_major ? a.in + (size_t)i * 32 : a.in + (size_t)a.rows[i] * a.row_stride * 32;
            dst[0][i] = *reinterpret_cast<const v4i*>(base + (g * in_lane_stride + 16u * h));
        }
    };
    auto row_acc = [&](int r, const v4i (&data)[1][M], v16i (&acc)[1]) __attribute__((always_inline)) {
        const uint8_t* cur = lds + (size_t)r * ROWB;
        const v4i* bp = reinterpret_cast<const v4i*>(cur + M * 1024 + h * 64);
        const v4i b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[0][k] = b0[k], acc[0][4 + k] = b1[k], acc[0][8 + k] = b2[k], acc[0][12 + k] = b3[k];
        mfma_row<M, 1>(cur + lane * 16, data, acc);
    };
    // one tile, all waves of the workgroup on it.  Every wave passes every barrier.
    auto process_tile = [&](size_t t, v4i (&data)[1][M]) __attribute__((always_inline)) {
        const uint32_t g = chunk_of(t);
        const bool live = t * 32 + c < a.G;
#pragma unroll
        for (int i = 0; i < M; ++i) data[0][i] = flip(data[0][i]);
        // verify rows: wave w takes rows w, w + WAVES, ...
        uint32_t bad = 0;
        for (int r = wave; r < nver; r += WAVES) {
            const uint8_t* base = a.in + (size_t)a.rows[M + role.row0 + r] * a.row_stride * 32;
            const v4i ys = *reinterpret_cast<const v4i*>(base + (g * 32u + 16u * h));
            v16i acc[1];
            row_acc(r, data, acc);
            bad |= verify_tile(acc[0], ys, H);
        }
        bool ok = true;
        if (nver > 0) {
            if (bad != 0) flags[c] = 1u;  // both halves of a pair, any wave: the same value
            __syncthreads();
            ok = flags[c] == 0u;
        }
        // output rows: distributed the same way
        for (int r = nver + wave; r < role.nrows; r += WAVES) {
            v16i acc[1];
            row_acc(r, data, acc);
            uint32_t Rw[4];
            reduce_tile(acc[0], Rw, H);
            if (a.direct && !ok) Rw[0] = Rw[1] = Rw[2] = Rw[3] = 0u;  // one role, no OEC round: a failed chunk gets zeros
            const size_t k = (size_t)(role.row0 + r - a.nv);
            uint8_t* qb = a.out_party_major ? a.out + k * a.out_stride * 32 : a.out + k * 32;
            const uint32_t qo = g * (a.out_party_major ? 32u : (uint32_t)a.out_stride * 32u) + 16u * h;
            if (live) *reinterpret_cast<uint4*>(qb + qo) = make_uint4(Rw[0], Rw[1], Rw[2], Rw[3]);
        }
        // the verdict of the tile: the first wave of the role that owns the verify rows
        if (wave == 0 && verdict_role && (a.status != nullptr || a.flagged != nullptr)) {
            const bool flag = live && !ok && h == 0;
            const unsigned long long fm = __ballot(flag);
            if (fm != 0 && a.direct) {
                if (lane == __ffsll((long long)fm) - 1) {
                    atomicAdd(a.counters, (uint32_t)__popcll(fm));
                    atomicMax(a.counters + 1, 0xffffffffu - g);
                    __threadfence();
                }
            } else if (fm != 0) {
                const int leader = __ffsll((long long)fm) - 1;
                uint32_t base = 0;
                if (lane == leader) base = atomicAdd(a.counters, (uint32_t)__popcll(fm));
                base = __shfl(base, leader);
                const size_t slot = (size_t)base + __popcll(fm & ((1ull << lane) - 1ull));
                if (flag && slot < a.G) a.flagged[slot] = g;  // the list has G entries (handoff_count)
            }
            if (live && h == 0) {
                if (a.status) a.status[g] = ok ? 0 : a.direct ? (uint8_t)DecodingError : 0xff;  // 0xff: pending, rewritten by the fallback kernels
                if (a.ncoeffs && (ok || a.direct)) a.ncoeffs[g] = ok ? M : 0;
            }
        }
        if (nver > 0) {
            __syncthreads();  // everybody has read the flags
            if (threadIdx.x < 32) flags[threadIdx.x] = 0u;
            // the next tile's flag writes come after its own verify rows and the barrier above orders them after this
            // clear only per wave; one more barrier keeps a fast wave's write from being cleared
            __syncthreads();
        }
    };
    // two input register sets as in k_mfma_rows: the next tile's loads go out before this tile's rows
    v4i setA[1][M], setB[1][M];
    const size_t tstep = (size_t)role_wgs;
    size_t t = (size_t)wg_in_role;
    if (t < ntiles) load_inputs(t, setA);
    while (t < ntiles) {  // workgroup-uniform
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
