// kernels_mfma_rt.hpp -- the matrix-core form of the constant-matrix maps with the TABLE IN REGISTERS.
//
// Same formulation, table layout and epilogue arithmetic as kernels_mfma.hpp (decode: robust_interpolate.rs:391-427;
// encode: common/share/mod.rs:50-76).  What changes is where things live and who overlaps with whom:
//
//   k_mfma_rows   table rows resident in LDS (<= 160 KB: config 3's 21 rows do not fit, so two workgroup kinds each read
//                 every input), a wave owns a tile and walks all rows: its 11-MFMA chains and its vector epilogues run
//                 back to back, and the A operand of every MFMA is a 1 KiB LDS read.
//   k_mfma_rt     one wave per SIMD with the whole 512-entry register file: a wave keeps ITS rows of the table in
//                 registers for the lifetime of the workgroup (config 3: 5 - 6 rows x 11 slabs x 4 registers), the four
//                 waves of a workgroup together hold every row (the register files of a CU are 512 KB, three times its
//                 LDS), so every workgroup serves all rows and the inputs are read from HBM ONCE.  The inputs of a
//                 32-chunk tile arrive by LDS-DMA (global_load_lds_dwordx4, no staging registers) into a ring of tile
//                 slots several tiles ahead of their use; the wave that issued a slab flips its sign bits in place (the
//                 MFMA is signed x signed) one tile ahead, so the B operand of an MFMA is a plain conflict-free
//                 ds_read_b128.  Inside a wave the M MFMAs of row j run with the vector epilogue of row j - 1 between
//                 them (two accumulators), and so do the DMA issue, the flips and the wait for the next tile: the matrix
//                 pipe and the vector ALU overlap within ONE instruction stream instead of relying on other waves.
//                 One s_barrier per tile.
//
// Synchronisation of the ring (NSLOT slots, D = NSLOT - 1 tiles in flight), per iteration q of a workgroup:
//   s_waitcnt lgkmcnt(0) ; s_barrier      every wave has issued (and, for the MFMAs that consumed them, received) its LDS
//                                         reads of tile q - 1, and the flipped slabs of tile q are visible
//   rows of tile q, and in their gaps:
//     DMA of tile q + D                   into the slot tile q - 1 occupied
//     s_waitcnt vmcnt(N) ; flip q + 1     N <= (DMA groups issued after tile q + 1's) x (DMA instructions of THIS wave per
//                                         tile): a lower bound of what was issued after them -- the output stores of the
//                                         rows in between also count in vmcnt and complete in order, so the wait can only
//                                         be stricter than needed, never weaker
//   the last row's epilogue is carried into tile q + 1 (its claimed values are read before the next barrier); the verdict
//   of tile q is published in iteration q + 1 and reported in iteration q + 2.
// All global reads inside the loop are asm DMA (hipcc would otherwise drain them with vmcnt(0) at its own waits); the
// stores are buffer stores, whose out-of-range form replaces the `if (live)` branches.  sched_barrier pins the order
// "B operand request, MFMA, piece of vector work" gap by gap.
#pragma once
#include "kernels_mfma_lab.hpp"  // the staged epilogue (RtEpi, rt_epi_stage)

namespace hbmpc {
namespace mf {

// the arguments of k_mfma_rows plus what only this kernel needs
struct RtArgs : MfmaRowsArgs {
    uint8_t role_wv[MF_MAX_ROLES];  // waves [0, wv) of a workgroup take the role's verify rows (mf_rt_plan)
    long long* prof;                // built with HBMPC_RT_PROF: cycles per section
};
constexpr int RT_MAX_SLOTS = 8;
constexpr int RT_VERD_STRIDE = 8;  // verdict words per tile (one per wave, at most 8 waves)
constexpr int RT_BD = 2;           // the B operand of MFMA n is requested in gap n - RT_BD
constexpr int RT_NB = RT_BD + 1;

// s_waitcnt vmcnt(4 * min(n / 4, 15)): an immediate at or below the one asked for only waits longer
HB_DEV void rt_wait_vmcnt(int n) {
    const int k = n >= 60 ? 15 : n >> 2;
#define HBMPC_RT_W(v) asm volatile("s_waitcnt vmcnt(" #v ")" ::: "memory")
    if (k & 8) {
        if (k & 4) {
            if (k & 2) { if (k & 1) HBMPC_RT_W(60); else HBMPC_RT_W(56); }
            else { if (k & 1) HBMPC_RT_W(52); else HBMPC_RT_W(48); }
        } else {
            if (k & 2) { if (k & 1) HBMPC_RT_W(44); else HBMPC_RT_W(40); }
            else { if (k & 1) HBMPC_RT_W(36); else HBMPC_RT_W(32); }
        }
    } else {
        if (k & 4) {
            if (k & 2) { if (k & 1) HBMPC_RT_W(28); else HBMPC_RT_W(24); }
            else { if (k & 1) HBMPC_RT_W(20); else HBMPC_RT_W(16); }
        } else {
            if (k & 2) { if (k & 1) HBMPC_RT_W(12); else HBMPC_RT_W(8); }
            else { if (k & 1) HBMPC_RT_W(4); else HBMPC_RT_W(0); }
        }
    }
#undef HBMPC_RT_W
}

// one LDS-DMA piece: lane l's 16 bytes at sbase + voff land at lds_dst + 16 l (sbase, lds_dst wave-uniform).  M0 is
// compiler-reserved: saved and restored inside the statement.
HB_DEV void rt_dma16(const uint8_t* sbase, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}
HB_DEV void rt_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// timing-only ablations of tools/ubench_mfma_rt.hip (results are wrong with any of them): 1 no epilogue, 2 no DMA inside the
// loop, 4 no wait / flips, 8 no B operand reads, 16 no MFMAs, 32 no bias reloads
#define RT_ABL(a, bit) ((ABL & (bit)) != 0)
constexpr bool rt_abl_stage(int abl, int st) {  // 64: stages 0-2, 128: 3, 256: 4-7, 512: 8-10, 1024: 11, 2048: 12
    return ((abl & 64) && st <= 2) || ((abl & 128) && st == 3) || ((abl & 256) && st >= 4 && st <= 7) || ((abl & 512) && st >= 8 && st <= 10) ||
           ((abl & 1024) && st == 11) || ((abl & 2048) && st == 12);
}

// Row plan of a workgroup (host side, mf_rt_plan): the verify rows of the role go round-robin to waves [0, wv), the
// output rows to waves [wv, 4): a wave's rows are all of one kind, which keeps its row loop free of branches.
struct RtWave {
    int kind;    // 1: verify rows
    int first;   // table row of slot 0
    int stride;  // table rows between slots
    int count;   // real rows (slots beyond are dummies: zero table, results discarded)
};
HB_DEV RtWave rt_wave_rows(const MfmaRole& role, int nver, int wv, int wave, int nwaves) {
    RtWave w;
    if (wave < wv) {
        w.kind = 1, w.first = role.row0 + wave, w.stride = wv;
        w.count = nver > wave ? (nver - wave + wv - 1) / wv : 0;
    } else {
        const int no = role.nrows - nver, wo = nwaves - wv, wi = wave - wv;
        w.kind = 0, w.first = role.row0 + nver + wi, w.stride = wo;
        w.count = no > wi ? (no - wi + wo - 1) / wo : 0;
    }
    return w;
}

struct RtState {
    const uint8_t* ring;     // LDS
    const uint8_t* biasL;    // LDS: bias of role row r at biasL + 128 r
    uint32_t* verd;          // LDS: [4][4] verdict words
    uint32_t slotb;          // bytes per ring slot
    int lane, c, h, wave;
    RtWave w;
    int role_row0, nver;
    __amdgpu_buffer_rsrc_t out_rsrc;
    uint32_t out_lane_stride;  // bytes between consecutive chunks of one output row
};

struct RtCarry {
    v4i ys;      // claimed values of the carried row (verify waves)
    RtRow row;   // where the carried row's result goes
};

// What else a wave does in the gaps of a tile (besides the epilogue of the previous row): per tile, in this order of gaps
//   gap 1 + 2 k           DMA piece k of tile q + D                      (k < pw)
//   gap W0                 wait for this wave's pieces of tile q + 1
//   gap W0 + 1 + k         read slab k of tile q + 1 for its sign flip    (k < fw)
//   gap W0 + 1 + k + LAT   write it back
// Gaps beyond the tile's KA M are run after its last MFMA.
template <int M, int RPW, int W>
struct RtSched {
    static constexpr int PWMAX = (M + W * RPW + W - 1) / W;  // DMA pieces per wave and tile, at most
    static constexpr int FWMAX = (M + W - 1) / W;            // data slabs per wave and tile, at most
    static constexpr int W0 = 2 * PWMAX + 1;
    static constexpr int END = W0 + 1 + FWMAX;
};
template <int M, int RPW, int W>
struct RtExtra {
    // wave constants
    int pw, fw;                                   // DMA pieces / data slabs of this wave per tile
    uint32_t rowidx[RtSched<M, RPW, W>::PWMAX];      // piece k: input row (decode: sender row position; encode: element index)
    uint64_t rstride;                             // bytes between input rows
    const uint8_t* in;
    uint32_t wave_slab0;                          // (wave) * 1024: LDS offset of this wave's first slab inside a slot
    // per tile
    bool dma_on, flip_on;
    uint32_t dma_voff, dma_dst;                   // lane offset inside an input row; LDS address of the slot of tile q + D
    uint32_t flip_base;                           // LDS address of this lane's 16 bytes of slab 0 of tile q + 1's slot
    int wait_n;
};
template <int M, int RPW, int W>
HB_DEV void rt_extra(int n, RtExtra<M, RPW, W>& x) {
    using S = RtSched<M, RPW, W>;
#pragma unroll
    for (int k = 0; k < S::PWMAX; ++k)
        if (n == 1 + 2 * k && x.dma_on && k < x.pw) {
            const uint8_t* sb = x.in + (uint64_t)x.rowidx[k] * x.rstride;
            rt_dma16(sb, x.dma_voff, x.dma_dst + x.wave_slab0 + (uint32_t)k * (W * 1024u));
        }
    if (n == S::W0 && x.flip_on && x.fw > 0) rt_wait_vmcnt(x.wait_n);
#pragma unroll
    for (int k = 0; k < S::FWMAX; ++k)
        if (n == S::W0 + 1 + k && x.flip_on && k < x.fw) {
            // sign flip in place: two LDS atomic XORs of 8 bytes per lane (no return value, no register, nothing to wait for)
            const uint32_t addr = x.flip_base + x.wave_slab0 + (uint32_t)k * (W * 1024u);
            const uint64_t m = 0x8080808080808080ull;
            asm volatile("ds_xor_b64 %0, %1\n\tds_xor_b64 %0, %1 offset:8" : : "v"(addr), "v"(m) : "memory");
        }
}

// The rows of ONE tile for one wave: KA slots of M MFMAs.  Gap n = j M + i of the tile: the B operand of MFMA n + RT_BD
// is requested, MFMA n issues, then a piece of the epilogue of the PREVIOUS row runs (slot 0: the previous tile's last
// row, described by `carry`) and whatever rt_extra has for this gap; once that row's sums have been gathered its
// accumulator takes the bias of the row that will use it next.  On entry b[0 .. RT_BD) hold (or have in flight) the first
// slabs of this tile and the accumulator of slot 0 holds its bias.  P: which accumulator slot 0 takes.
template <int M, int RPW, int W, bool VERIFY, int KA, int P, int ABL>
HB_DEV void rt_tile(const RtArgs& a, const RtState& s, const v4i (&tab)[RPW][M], const uint8_t* slot, v4i (&b)[RT_NB], uint32_t g, bool live,
                    v16i& accA, v16i& accB, RtCarry& carry, uint32_t& bad, uint32_t& bad_carried, const Half& H, RtExtra<M, RPW, W>& x) {
    const uint8_t* bsrc = slot + s.lane * 16;
    uint32_t one = 1u;
    asm volatile("" : "+s"(one));  // opaque: t = hi * 1 + Q stays a v_mad_u64_u32
    v4i ys_prev = carry.ys, ys_cur = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < KA; ++j) {
        v16i& acc = ((j + P) & 1) ? accB : accA;
        v16i& prev = ((j + P) & 1) ? accA : accB;
        // the row being finished in this slot's gaps
        RtRow row;
        if (j == 0) {
            row = carry.row;
        } else {
            const bool real = j - 1 < s.w.count;
            const uint32_t k = (uint32_t)(s.w.first + (j - 1) * s.w.stride - a.nv);
            row.voff = (live && real) ? g * s.out_lane_stride + 16u * (uint32_t)s.h : RT_OOB;
            row.soff = real ? (a.out_party_major ? k * (uint32_t)a.out_stride * 32u : k * 32u) : 0u;
            row.mask = (live && real) ? ~0u : 0u;
        }
        // the row whose bias `prev` takes once its sums are gathered: slot j + 1 (slot 0 of the next tile after the last)
        const int jn = j + 1 < KA ? j + 1 : 0;
        const int rn = jn < s.w.count ? s.w.first + jn * s.w.stride - s.role_row0 : 0;
        const v4i* bp = reinterpret_cast<const v4i*>(s.biasL + (size_t)rn * 128 + s.h * 64);
        RtEpi e;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            const int n = j * M + i;
            if (n + RT_BD < KA * M && !RT_ABL(a, 8)) b[(n + RT_BD) % RT_NB] = *reinterpret_cast<const v4i*>(bsrc + ((n + RT_BD) % M) * 1024);
            if (!RT_ABL(a, 16)) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(tab[j][i], b[n % RT_NB], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int st = 0; st < RT_NSTAGE; ++st)
                if (rt_stage_gap<M>(st) == i && (!RT_ABL(a, 1) || st == RT_NSTAGE - 1) && !rt_abl_stage(ABL, st)) rt_epi_stage<VERIFY>(st, e, prev, ys_prev, row, H, one, s.out_rsrc, j == 0 ? bad_carried : bad);
            if (i == rt_bias_gap<M>() && !RT_ABL(a, 32)) {
                const v4i b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
#pragma unroll
                for (int k = 0; k < 4; ++k) prev[k] = b0[k], prev[4 + k] = b1[k], prev[8 + k] = b2[k], prev[12 + k] = b3[k];
            }
            if (VERIFY && i == M - 1) {
                const int vi = j < s.w.count ? s.w.first + j * s.w.stride - s.role_row0 : 0;  // verify rows come first in the role
                ys_cur = *reinterpret_cast<const v4i*>(bsrc + (size_t)(M + vi) * 1024);
            }
            rt_extra<M, RPW, W>(n, x);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (!VERIFY) {
            if (__builtin_expect(__any(e.res != 0) != 0, 0)) {
                rt_reduce_slow(e, H);
                v4i val;
                val[0] = (int)e.U[0], val[1] = (int)e.U[1], val[2] = (int)e.U[2], val[3] = (int)e.U[3];
                __builtin_amdgcn_raw_buffer_store_b128(val, s.out_rsrc, (int)row.voff, (int)row.soff, 0);
            }
        }
        ys_prev = ys_cur;
    }
#pragma unroll
    for (int n = KA * M; n < RtSched<M, RPW, W>::END; ++n) rt_extra<M, RPW, W>(n, x);
    carry.ys = ys_prev;
    {
        const bool real = KA > 0 && KA - 1 < s.w.count;  // KA = 0: a wave without rows only moves data
        const uint32_t k = (uint32_t)(s.w.first + (KA - 1) * s.w.stride - a.nv);
        carry.row.voff = (live && real) ? g * s.out_lane_stride + 16u * (uint32_t)s.h : RT_OOB;
        carry.row.soff = real ? (a.out_party_major ? k * (uint32_t)a.out_stride * 32u : k * 32u) : 0u;
        carry.row.mask = (live && real) ? ~0u : 0u;
    }
}

template <int M, int RPW, int W, bool VERIFY, int KA, int ABL>
HB_DEV void rt_wave_loop(const RtArgs& a, const RtState& s, const MfmaRole& role, int role_wgs, int wg_in_role, int nslot,
                         const v4i (&tab)[RPW][M], const Half& H) {
    using S = RtSched<M, RPW, W>;
    const size_t ntiles = (a.G + 31) / 32;
    const int nq = ntiles > (size_t)wg_in_role ? (int)((ntiles - wg_in_role + role_wgs - 1) / role_wgs) : 0;
    const int D = nslot - 1;
    const int NS = M + s.nver;
    const uint32_t ring0 = (uint32_t)(uintptr_t)s.ring;
    auto tile_of = [&](int q) { return (size_t)wg_in_role + (size_t)q * role_wgs; };
    const uint32_t in_lane_stride = a.in_chunk_major ? M * 32u : 32u;
    auto in_voff = [&](int q) {
        const size_t gi = tile_of(q) * 32 + s.c;
        return (uint32_t)(gi < a.G ? gi : a.G - 1) * in_lane_stride + 16u * (uint32_t)s.h;
    };
    RtExtra<M, RPW, W> x;
    x.pw = NS > s.wave ? (NS - s.wave + W - 1) / W : 0;
    x.fw = M > s.wave ? (M - s.wave + W - 1) / W : 0;
    x.rstride = a.in_chunk_major ? 32u : (uint64_t)a.row_stride * 32u;
    x.in = a.in;
    x.wave_slab0 = (uint32_t)s.wave * 1024u;
#pragma unroll
    for (int k = 0; k < S::PWMAX; ++k) {
        const int sl = s.wave + W * k;
        uint32_t r = 0;
        if (sl < NS) r = a.in_chunk_major ? (uint32_t)sl : (uint32_t)a.rows[sl < M ? sl : M + s.role_row0 + (sl - M)];
        x.rowidx[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
    }
    x.dma_on = x.flip_on = false;
    const int spt = VERIFY ? 0 : (s.w.count < KA ? s.w.count : KA);  // output stores of this wave per tile
    x.dma_voff = x.dma_dst = 0;
    x.flip_base = 0;
    x.wait_n = 0;
    // DMA of tile q (prologue), flips of tile 0
    auto issue_all = [&](int q) {
        x.dma_on = true, x.flip_on = false;
        x.dma_voff = in_voff(q);
        x.dma_dst = ring0 + (uint32_t)(q % nslot) * s.slotb;
#pragma unroll
        for (int n = 0; n < S::W0; ++n) rt_extra<M, RPW, W>(n, x);
    };
    const bool reports = a.status != nullptr || a.flagged != nullptr;
    const bool role_judges = s.nver > 0 || (a.nv == 0 && role.row0 == 0);
    auto report = [&](int q, const v4i& vw, const v4i& vw2) {
        // verdict of tile q: OR of the four waves' words
        const uint32_t m32 = (uint32_t)__builtin_amdgcn_readfirstlane(vw[0] | vw[1] | vw[2] | vw[3] | vw2[0] | vw2[1] | vw2[2] | vw2[3]);
        const size_t gi = tile_of(q) * 32 + s.c;
        const bool live = gi < a.G;
        const uint32_t g = (uint32_t)gi;
        const bool ok = ((m32 >> s.c) & 1u) == 0;
        if (a.direct && m32 != 0 && !VERIFY) {
            // no OEC round exists: a chunk that failed the verification gets zeros (this wave's rows, behind its own stores)
            for (int j = 0; j < s.w.count; ++j) {
                const uint32_t k = (uint32_t)(s.w.first + j * s.w.stride - a.nv);
                const uint32_t voff = (live && !ok) ? g * s.out_lane_stride + 16u * (uint32_t)s.h : RT_OOB;
                const uint32_t soff = a.out_party_major ? k * (uint32_t)a.out_stride * 32u : k * 32u;
                const v4i z = {0, 0, 0, 0};
                __builtin_amdgcn_raw_buffer_store_b128(z, s.out_rsrc, (int)voff, (int)soff, 0);
            }
        }
        if (!(reports && role_judges) || s.wave != q % W) return;
        const bool flag = live && !ok && s.h == 0;
        const unsigned long long fm = __ballot(flag);
        if (fm != 0 && a.direct) {
            if (s.lane == __ffsll((long long)fm) - 1) {
                atomicAdd(a.counters, (uint32_t)__popcll(fm));
                atomicMax(a.counters + 1, 0xffffffffu - g);
                __threadfence();
            }
        } else if (fm != 0) {
            const int leader = __ffsll((long long)fm) - 1;
            uint32_t base = 0;
            if (s.lane == leader) base = atomicAdd(a.counters, (uint32_t)__popcll(fm));
            base = __shfl(base, leader);
            const size_t slot = (size_t)base + __popcll(fm & ((1ull << s.lane) - 1ull));
            if (flag && slot < a.G) a.flagged[slot] = g;
        }
        if (live && s.h == 0) {
            if (a.status) a.status[g] = ok ? 0 : a.direct ? (uint8_t)DecodingError : 0xff;
            if (a.ncoeffs && (ok || a.direct)) a.ncoeffs[g] = ok ? M : 0;
        }
    };
    auto publish = [&](int q, uint32_t badv) {
        if constexpr (VERIFY) {
            const unsigned long long m = __ballot(badv != 0);
            if (s.lane == 0) s.verd[(q & 3) * RT_VERD_STRIDE + s.wave] = (uint32_t)m | (uint32_t)(m >> 32);
        }
    };

    v16i accA, accB;
    {
        // slot 0 of the first tile takes accA: its bias; accB is the (non-existent) carried row
        const int r0 = s.w.count > 0 ? s.w.first - s.role_row0 : 0;
        const v4i* bp = reinterpret_cast<const v4i*>(s.biasL + (size_t)r0 * 128 + s.h * 64);
        const v4i b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
#pragma unroll
        for (int k = 0; k < 4; ++k) accA[k] = b0[k], accA[4 + k] = b1[k], accA[8 + k] = b2[k], accA[12 + k] = b3[k];
#pragma unroll
        for (int k = 0; k < 16; ++k) accB[k] = 0;
    }
    v4i b[RT_NB];
#pragma unroll
    for (int k = 0; k < RT_NB; ++k) b[k] = v4i{0, 0, 0, 0};
    RtCarry carry;
    carry.ys = v4i{0, 0, 0, 0};
    carry.row.voff = RT_OOB, carry.row.soff = 0, carry.row.mask = 0;
    uint32_t bad_prev = 0;

    int issued = -1;  // last tile whose DMA this wave has issued
    for (int q = 0; q < D && q < nq; ++q) issue_all(q), issued = q;
    if (nq > 0) {
        // tile 0: wait for it and flip it here
        x.dma_on = false, x.flip_on = true;
        x.wait_n = issued * x.pw;
        x.flip_base = ring0 + (uint32_t)s.lane * 16u;
#pragma unroll
        for (int n = S::W0; n < S::END; ++n) rt_extra<M, RPW, W>(n, x);
    }
#ifdef HBMPC_RT_PROF
    long long pf[6] = {0, 0, 0, 0, 0, 0}, pt = __builtin_amdgcn_s_memtime(), pn;
#define RT_STAMP(k) pn = __builtin_amdgcn_s_memtime(), pf[k] += pn - pt, pt = pn
#else
#define RT_STAMP(k)
#endif
    for (int q = 0; q < nq + 2; ++q) {
        rt_barrier();
        RT_STAMP(0);
        const uint8_t* slot = s.ring + (size_t)(q % nslot) * s.slotb;
        if (q < nq) {
#pragma unroll
            for (int n = 0; n < RT_BD && n < KA * M; ++n) b[n] = *reinterpret_cast<const v4i*>(slot + (n % M) * 1024 + s.lane * 16);
        }
        v4i vw = {0, 0, 0, 0}, vw2 = {0, 0, 0, 0};
        if (q >= 2) {
            vw = *reinterpret_cast<const v4i*>(s.verd + ((q - 2) & 3) * RT_VERD_STRIDE);
            if (W > 4) vw2 = *reinterpret_cast<const v4i*>(s.verd + ((q - 2) & 3) * RT_VERD_STRIDE + 4);
        }
        // what rt_extra does in this tile's gaps
        x.dma_on = q + D < nq;
        if (x.dma_on && RT_ABL(a, 2)) issued = q + D, x.dma_on = false;
        if (x.dma_on) {
            issued = q + D;
            x.dma_voff = in_voff(q + D);
            x.dma_dst = ring0 + (uint32_t)((q + D) % nslot) * s.slotb;
        }
        x.flip_on = q + 1 < nq && !RT_ABL(a, 4);
        // issued after tile q + 1's pieces, at least: the pieces of the later tiles and, between two groups of pieces, the
        // output stores of one tile (one per real output row)
        x.wait_n = (issued - (q + 1)) * x.pw + (issued - (q + 1) > 0 ? (issued - (q + 1) - 1) * spt : 0);
        x.flip_base = ring0 + (uint32_t)((q + 1) % nslot) * s.slotb + (uint32_t)s.lane * 16u;
        RT_STAMP(1);
        if (q >= 2) report(q - 2, vw, vw2);
        RT_STAMP(2);
        if (q < nq) {
            const size_t gi = tile_of(q) * 32 + s.c;
            const bool live = gi < a.G;
            const uint32_t g = (uint32_t)(live ? gi : 0);
            uint32_t bad_carried = bad_prev, bad_now = 0;
            if ((KA & 1) && (q & 1)) rt_tile<M, RPW, W, VERIFY, KA, 1, ABL>(a, s, tab, slot, b, g, live, accA, accB, carry, bad_now, bad_carried, H, x);
            else rt_tile<M, RPW, W, VERIFY, KA, 0, ABL>(a, s, tab, slot, b, g, live, accA, accB, carry, bad_now, bad_carried, H, x);
            if (q >= 1) publish(q - 1, bad_carried);
            bad_prev = bad_now;
        } else if (q == nq && nq > 0) {
            // the last tile's last row
            uint32_t bad_carried = bad_prev;
            const bool inB = (((KA - 1) + (((KA & 1) && ((nq - 1) & 1)) ? 1 : 0)) & 1) != 0;
            const v16i& prev = inB ? accB : accA;
            uint32_t one = 1u;
            asm volatile("" : "+s"(one));
            RtEpi e;
#pragma unroll
            for (int st = 0; st < RT_NSTAGE; ++st) rt_epi_stage<VERIFY>(st, e, prev, carry.ys, carry.row, H, one, s.out_rsrc, bad_carried);
            if constexpr (!VERIFY) {
                if (__any(e.res != 0)) {
                    rt_reduce_slow(e, H);
                    v4i val;
                    val[0] = (int)e.U[0], val[1] = (int)e.U[1], val[2] = (int)e.U[2], val[3] = (int)e.U[3];
                    __builtin_amdgcn_raw_buffer_store_b128(val, s.out_rsrc, (int)carry.row.voff, (int)carry.row.soff, 0);
                }
            }
            publish(nq - 1, bad_carried);
        }
        RT_STAMP(3);
    }
#ifdef HBMPC_RT_PROF
    if (a.prof && blockIdx.x < 8 && s.lane == 0) {
        for (int k = 0; k < 6; ++k) a.prof[((size_t)blockIdx.x * 8 + s.wave) * 8 + k] = pf[k];
        a.prof[((size_t)blockIdx.x * 8 + s.wave) * 8 + 6] = nq;
    }
#endif
#undef RT_STAMP
}

// LDS bytes of a launch: the ring, the bias rows, the verdict words
inline size_t mf_rt_lds_bytes(int m, int nver_max, int rows_max, int nslot) {
    return (size_t)nslot * (size_t)(m + nver_max) * 1024 + (size_t)rows_max * 128 + 4 * RT_VERD_STRIDE * 4;
}

// OCC workgroups per CU (each holds every row of its role): with OCC = 2 two independent workgroups share a CU and one
// computes while the other sits in its barrier / set-up phase -- for tables small enough for 256 registers per wave
template <int M, int RPW, int W = 4, int ABL = 0, int OCC = 1>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(OCC * W / 4, OCC * W / 4))) void k_mfma_rt(RtArgs a, int nslot) {
    static_assert(M <= 15, "digit sums must stay below 0xff0000 (tables_mfma.hpp)");
    constexpr int ROWB = M * 1024 + 128;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    if (a.summary && !a.direct && blockIdx.x == 0 && threadIdx.x < 4) a.summary[threadIdx.x] = threadIdx.x == 2 ? 0xffffffffu : 0u;
    const int blk8 = (int)blockIdx.x >> 3, role_id = a.blk_role[blk8];
    const int wg_in_role = (int)a.blk_idx[blk8] * 8 + ((int)blockIdx.x & 7);
    MfmaRole role = a.role[0];
    int role_wgs = a.role_nwg[0], wv = a.role_wv[0];
#pragma unroll
    for (int k = 1; k < MF_MAX_ROLES; ++k)
        if (k == role_id) role = a.role[k], role_wgs = a.role_nwg[k], wv = a.role_wv[k];
    RtState s;
    s.lane = threadIdx.x & 63, s.c = s.lane & 31, s.h = s.lane >> 5;
    s.wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    s.role_row0 = role.row0;
    s.nver = role.row0 < a.nv ? (a.nv - role.row0 < role.nrows ? a.nv - role.row0 : role.nrows) : 0;
    s.slotb = (uint32_t)(M + s.nver) * 1024u;
    s.ring = lds;
    s.biasL = lds + (size_t)nslot * s.slotb;
    s.verd = reinterpret_cast<uint32_t*>(lds + (size_t)nslot * s.slotb + (size_t)role.nrows * 128);
    s.w = rt_wave_rows(role, s.nver, wv, s.wave, W);
    {
        const size_t ow = a.out_party_major ? 0 : a.out_stride;  // chunk-major: G records of out_stride elements
        const size_t bytes = a.out_party_major ? ((size_t)(role.row0 + role.nrows - a.nv - 1) * a.out_stride + a.G) * 32 : a.G * ow * 32;
        s.out_rsrc = rt_rsrc(a.out, (uint32_t)(bytes < 0xffffffe0ull ? bytes : 0xffffffe0ull));
        s.out_lane_stride = a.out_party_major ? 32u : (uint32_t)a.out_stride * 32u;
    }
    // this wave's rows of the table -> registers (dummy slots: zeros)
    v4i tab[RPW][M];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const uint8_t* src = a.table + (size_t)(s.w.first + j * s.w.stride) * ROWB + s.lane * 16;
#pragma unroll
        for (int i = 0; i < M; ++i) tab[j][i] = j < s.w.count ? *reinterpret_cast<const v4i*>(src + i * 1024) : v4i{0, 0, 0, 0};
    }
    for (int p = threadIdx.x; p < role.nrows * 8; p += 64 * W)
        reinterpret_cast<v4i*>(lds + (size_t)nslot * s.slotb)[p] =
            *reinterpret_cast<const v4i*>(a.table + (size_t)(role.row0 + (p >> 3)) * ROWB + M * 1024 + (p & 7) * 16);
    if (threadIdx.x < 4 * RT_VERD_STRIDE) s.verd[threadIdx.x] = 0u;
    __syncthreads();
    const Half H = make_half(s.h);
    // a wave with fewer than RPW rows runs RPW - 1 slots (anything shorter is padded with dummy rows)
    if (s.w.kind) {
        if (s.w.count >= RPW) rt_wave_loop<M, RPW, W, true, RPW, ABL>(a, s, role, role_wgs, wg_in_role, nslot, tab, H);
        else rt_wave_loop<M, RPW, W, true, RPW - 1, ABL>(a, s, role, role_wgs, wg_in_role, nslot, tab, H);
    } else {
        if (s.w.count >= RPW) rt_wave_loop<M, RPW, W, false, RPW, ABL>(a, s, role, role_wgs, wg_in_role, nslot, tab, H);
        else rt_wave_loop<M, RPW, W, false, RPW - 1, ABL>(a, s, role, role_wgs, wg_in_role, nslot, tab, H);
    }
    if (a.direct) finish_direct(a.counters, a.summary);
}

// Host side: split the rows of every role between verify waves and output waves so that the longest wave is as short as
// possible; returns that length (the RPW the launch needs).
inline int mf_rt_plan(RtArgs* a, int nwaves = 4) {
    int need = 0;
    for (int k = 0; k < a->nroles; ++k) {
        const MfmaRole& r = a->role[k];
        const int nver = r.row0 < a->nv ? (a->nv - r.row0 < r.nrows ? a->nv - r.row0 : r.nrows) : 0, no = r.nrows - nver;
        int best = -1, best_len = 1 << 30;
        for (int wv = 0; wv <= nwaves; ++wv) {
            if ((nver > 0) != (wv > 0) || (no > 0 && wv == nwaves)) continue;
            const int lv = wv ? (nver + wv - 1) / wv : 0, lo = no ? (no + (nwaves - wv) - 1) / (nwaves - wv) : 0;
            const int len = lv > lo ? lv : lo;
            if (len < best_len) best_len = len, best = wv;
        }
        a->role_wv[k] = (uint8_t)best;
        need = best_len > need ? best_len : need;
    }
    return need;
}

}  // namespace mf
}  // namespace hbmpc
