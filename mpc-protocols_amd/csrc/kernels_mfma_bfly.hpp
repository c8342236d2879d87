// kernels_mfma_bfly.hpp -- the encode (apply_vandermonde, common/share/mod.rs:50-76; compute_shares,
// robust_interpolate.rs:52-82) on the matrix cores with HALF the MFMAs of kernels_mfma.hpp.
//
// The evaluation points are the powers of a root of unity of order size = 2 half (common/mod.rs:51-68), so
// alpha_{k + half} = -alpha_k and, with the polynomial split by coefficient parity,
//     p(alpha_k)        = E_k + T_k          E_k = sum_{i even} c_i alpha_k^i
//     p(alpha_{k+half}) = E_k - T_k          T_k = sum_{i odd}  c_i alpha_k^i
// E_k and T_k are constant-matrix maps of the even / odd coefficients: in the byte-digit formulation (tables_mfma.hpp)
// their digit sums come from the SAME table slabs as row k of the plain kernel -- slab i goes to the accumulator of its
// parity -- and the sum / difference is taken on the un-normalised digit sums (exact integers, far from overflow)
// before the one carry pass and reduction each output needs anyway.  A pair of outputs so costs M MFMAs and M KB of
// LDS operand traffic per 32 chunks instead of 2 M, plus 16 vector adds.
//
// Table row p (pair p): the M slabs of point p exactly as build_mfma_table lays them out, then two accumulator biases
// [lane half][16] as int32: bE for the even accumulator and bT for the odd one, with bE + bT = the plain bias of row p and
// bE - bT = a bias of row p + half (digit representation adjusted to the parity of the first, see
// tables_mfma.hpp::build_mfma_bfly_table).  A point without a partner (p + half >= nout) has bT = 0 and no second output.
//
// Any n x m map whose rows p and p + half differ by the sign of the odd columns fits: the Vandermonde encode, the same with
// its inputs given as rows (the producers' n x n mixing step), and the inverse transform (Lagrange interpolation through
// all points of a full domain, hbmpc_dev_batch_interpolate).
#pragma once
#include "kernels_mfma.hpp"

namespace hbmpc {
namespace mf {

constexpr int MF_BFLY_BIAS = 256;

// NP > 0: every role has exactly NP pairs (the host plans it so) and the pair loop is unrolled
// (Timing-only ablations and the whole-line read path of round 3 live in tools/kernels_mfma_bfly_lab.hpp.)
// TRIPLE: the encode of triple generation -- the inputs are the local products a b - r2t of three arrays
// (triple_gen/triple_generation.rs:333-340: the product of the parties' shares, masked with the degree-2t randomness before
// it is opened), computed in this kernel: lane (chunk, h) multiplies the coefficients of its chunk whose index has parity h
// (one lazy product and one Montgomery reduction each, fr_u29.hpp) and one v_permlane32_swap per word hands the halves
// round so that every lane ends up with its 16 bytes of every coefficient -- the B operands, without a trip through memory
// or the LDS.  The tile index then runs over parties x tiles (x[P][G][M] -> y[P][n][G]).
// DEG: the outputs of a chunk are the coefficients of a polynomial (out chunk-major, one role) and a.ncoeffs[g] receives its
// degree -- the index of the highest nonzero coefficient, 0 for the zero polynomial (DensePolynomial::degree(), what the
// RanDouSha verifier tests, ran_dou_sha/mod.rs:586-589) -- so the coefficients are not read a second time for it; with
// a.store_rows = 1 only c_0 is stored (the verifier's other test compares the constant terms of its two polynomials).
// LISTS: the producers' mixing step with the parties' output rows written in list order (MfmaRowsArgs::list) and, with other_stride, the rows
// the verifiers receive party-major ([party][verifier][k]: one decode serves all verifiers of a kind).  With no list rows at all the same
// instance is the party-batched encode x[P][G][M] -> y[P][n][G] in one launch (list_K = G, other_stride = n G: the dealers' compute_shares).
template <int M, int WAVES, int NP = 0, bool TRIPLE = false, int TRIPLE_DEPTH = 1, bool DEG = false, bool LISTS = false>
__global__ __launch_bounds__(64 * WAVES) void k_mfma_bfly(MfmaRowsArgs a) {
    static_assert(M >= 2 && M <= 16, "digit sums must stay below 0xff0000 (tables_mfma.hpp: proved per table for M = 16)");
    constexpr int ROWB = M * 1024 + MF_BFLY_BIAS;
    constexpr int NT = 64 * WAVES;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];  // role.nrows * ROWB
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
    const size_t ntiles = (a.G + 31) / 32;
    const size_t tstep = (size_t)role_wgs * WAVES;
    // NP > 0: nothing inside the tile loop is conditional -- the next tile's loads are issued even past the end (clamped to
    // the last chunk) and stores go through a buffer descriptor that drops what must not be written (dead lanes: an offset
    // beyond the buffer; a partner point that does not exist: a buffer of zero bytes) -- so that hipcc knows how many vector
    // memory operations follow a load and waits for a tile's inputs with vmcnt(stores + later loads) instead of vmcnt(0).
    // vmcnt retires in order: with a conditional load or store in between, the wait for the OLDER input set also drains
    // the set requested a moment ago, i.e. every other tile would pay a full HBM round trip.
    constexpr bool STATIC = NP > 0;
    // inputs: chunk-major x[G][M] (compute_shares, apply_vandermonde) or M rows, row i at in + rows[i] * row_stride * 32 (the
    // producers' mixing step reads the dealt shares where the dealers' encodes left them: S[dealer][..])
    auto load_inputs = [&](size_t t, v4i (&dst)[M]) {
        const size_t gi = t * 32 + c;
        const uint32_t g = (uint32_t)(gi < a.G ? gi : a.G - 1);
#pragma unroll
        for (int i = 0; i < M; ++i) {
            if (a.in_chunk_major) {
                dst[i] = *reinterpret_cast<const v4i*>(a.in + (size_t)i * 32 + (g * (M * 32u) + 16u * h));
            } else {
                uint32_t ri = (uint32_t)a.rows[i];
                asm volatile("" : "+s"(ri));  // recomputed at every use (scalar registers, as in k_mfma_rows)
                dst[i] = *reinterpret_cast<const v4i*>(a.in + (size_t)ri * a.row_stride * 32 + (g * 32u + 16u * h));
            }
        }
    };
    const uint32_t row_bytes = (uint32_t)([&] {
        const size_t b = a.out_party_major ? a.G * 32 : a.G * a.out_stride * 32;
        return b < 0xffffffe0ull ? b : 0xffffffe0ull;
    }());
    // party-major other rows: a row's stores end (parties - 1) other_stride + list_K elements behind its base (the host keeps it below 2^32 bytes)
    [[maybe_unused]] const uint32_t other_bytes = LISTS && a.other_stride != 0 ? (uint32_t)(((a.G / a.list_K - 1) * (size_t)a.other_stride + a.list_K) * 32) : 0u;
    // sc: this lane's 16 bytes of list row 0 of its chunk (0: the chunk has no list destination), see MfmaRowsArgs::list
    auto store_row = [&](uint8_t* out, uint32_t k, bool exists, bool live, uint32_t qo, const uint32_t (&Rw)[4], uint64_t sc) {
        if (LISTS && k - (uint32_t)a.list_row0 < (uint32_t)a.list_rows) {  // wave-uniform
            if (exists && sc != 0) *reinterpret_cast<uint4*>(sc + (uint64_t)(k - (uint32_t)a.list_row0) * 32) = make_uint4(Rw[0], Rw[1], Rw[2], Rw[3]);
            return;
        }
        if (DEG && a.store_rows != 0 && k >= (uint32_t)a.store_rows) exists = false;
        uint8_t* qb = a.out_party_major ? out + (size_t)k * a.out_stride * 32 : out + (size_t)k * 32;  // wave-uniform
        uint32_t bound = row_bytes;
        if (LISTS && a.other_stride != 0) {  // the other rows party-major (MfmaRowsArgs::other_stride): qo is the lane's (party, k) offset
            const uint32_t rp = k < (uint32_t)a.list_row0 ? k : k - (uint32_t)a.list_rows;
            qb = out + (size_t)rp * a.list_K * 32;
            bound = other_bytes;
        }
        if constexpr (STATIC) {
            v4i val;
            val[0] = (int)Rw[0], val[1] = (int)Rw[1], val[2] = (int)Rw[2], val[3] = (int)Rw[3];
            __builtin_amdgcn_raw_buffer_store_b128(val, rt_rsrc(qb, exists ? bound : 0u), (int)(live ? qo : RT_OOB), 0, 0);
        } else {
            if (exists && live) *reinterpret_cast<uint4*>(qb + qo) = make_uint4(Rw[0], Rw[1], Rw[2], Rw[3]);
        }
    };
    // the pairs of one tile from its B operands (sign-flipped bytes)
    auto pairs_of_tile = [&](size_t t, const v4i (&data)[M], uint8_t* out) {
        const size_t gi = t * 32 + c;
        const bool live = gi < a.G;
        const uint32_t g = (uint32_t)(live ? gi : a.G - 1);
        uint32_t qo = g * (a.out_party_major ? 32u : (uint32_t)a.out_stride * 32u) + 16u * h;
        uint64_t sc = 0;
        if (LISTS && live) {
            const uint32_t j = g / a.list_K, kk = g - j * a.list_K;
            if (a.other_stride != 0) qo = (j * a.other_stride + kk) * 32u + 16u * h;
            const int s = kk - a.list[0].k0 < a.list[0].count ? 0 : kk - a.list[1].k0 < a.list[1].count ? 1 : -1;
            if (s >= 0) {
                const MfmaRowsArgs::ListSlice sl = s ? a.list[1] : a.list[0];
                sc = (uint64_t)sl.dst + ((uint64_t)j * sl.stride + (uint64_t)(kk - sl.k0) * (uint32_t)a.list_rows) * 32 + 16u * h;
            }
        }
        [[maybe_unused]] uint32_t degree = 0;
        [[maybe_unused]] auto note_degree = [&](uint32_t k, bool exists, const uint32_t (&Rw)[4]) {
            const uint32_t any = Rw[0] | Rw[1] | Rw[2] | Rw[3];
            const auto both = __builtin_amdgcn_permlane32_swap(any, any, false, false);  // [0]: the low half's value, [1]: the high half's
            if (exists && (both[0] | both[1]) != 0 && k > degree) degree = k;
        };
#pragma unroll
        for (int p = 0; p < (STATIC ? NP : role.nrows); ++p) {
            const uint8_t* cur = lds + (size_t)p * ROWB;
            v16i accE, accT;
            {
                const v4i* bp = reinterpret_cast<const v4i*>(cur + M * 1024 + h * 64);
                const v4i e0 = bp[0], e1 = bp[1], e2 = bp[2], e3 = bp[3];
                const v4i t0 = bp[8], t1 = bp[9], t2 = bp[10], t3 = bp[11];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    accE[k] = e0[k], accE[4 + k] = e1[k], accE[8 + k] = e2[k], accE[12 + k] = e3[k];
                    accT[k] = t0[k], accT[4 + k] = t1[k], accT[8 + k] = t2[k], accT[12 + k] = t3[k];
                }
            }
            {   // the M MFMAs: even inputs into accE, odd into accT (two independent chains); A operands three slabs ahead
                constexpr int D = 3;
                const uint8_t* tab_lane = cur + lane * 16;
                v4i av[D];
#pragma unroll
                for (int i = 0; i < D - 1 && i < M; ++i) av[i] = *reinterpret_cast<const v4i*>(tab_lane + i * 1024);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < M; ++i) {
                    if (i + D - 1 < M) av[(i + D - 1) % D] = *reinterpret_cast<const v4i*>(tab_lane + (i + D - 1) * 1024);
                    if (i & 1) accT = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[i % D], data[i], accT, 0, 0, 0);
                    else accE = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[i % D], data[i], accE, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            uint32_t pe[8], pt[8];
            gather_pairs(accE, pe);
            gather_pairs(accT, pt);
            uint32_t k32 = (uint32_t)(role.row0 + p);
            asm volatile("" : "+s"(k32));  // the output row bases are recomputed, not kept per unrolled pair (scalar registers)
            {
                uint64_t T[4];
                uint32_t Rw[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) T[j] = (uint64_t)(pe[2 * j + 1] + pt[2 * j + 1]) * H.k16 + (pe[2 * j] + pt[2 * j]);
                reduce_words(T, Rw, H);
                store_row(out, k32, true, live, qo, Rw, sc);
                if constexpr (DEG) note_degree(k32, true, Rw);
            }
            const bool partner = (int)k32 + a.half < a.nout;
            if (STATIC || partner) {
                uint64_t T[4];
                uint32_t Rw[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) T[j] = (uint64_t)(pe[2 * j + 1] - pt[2 * j + 1]) * H.k16 + (pe[2 * j] - pt[2 * j]);
                reduce_words(T, Rw, H);
                store_row(out, k32 + (uint32_t)a.half, partner, live, qo, Rw, sc);
                if constexpr (DEG) note_degree(k32 + (uint32_t)a.half, partner, Rw);
            }
        }
        if constexpr (DEG) {  // one more counted store: lanes that do not write get an offset beyond the buffer
            const size_t db = a.G * 4;
            __builtin_amdgcn_raw_buffer_store_b32(degree, rt_rsrc(a.ncoeffs, (uint32_t)(db < 0xffffffe0ull ? db : 0xffffffe0ull)),
                                                  (int)(live && h == 0 ? g * 4u : RT_OOB), 0, 0);
        }
    };
    auto process_tile = [&](size_t t, v4i (&data)[M]) {
#pragma unroll
        for (int i = 0; i < M; ++i) data[i] = flip(data[i]);
        pairs_of_tile(t, data, a.out);
    };
    auto dropped_stores = [&] {
        // as many dropped stores as a tile issues: the loop header then sees the same queue behind the first loads on entry
        // as on the back edge (hipcc merges the two states to the stricter wait)
        const v4i z = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 2 * NP + (DEG ? 1 : 0); ++k) __builtin_amdgcn_raw_buffer_store_b128(z, rt_rsrc(a.out, 0u), (int)RT_OOB, 0, 0);
    };
    if constexpr (TRIPLE) {
        using F = U29;
        constexpr int NS = (M + 1) / 2;  // slots: slot j holds coefficient 2 j + h of this lane's chunk
        struct Slot {
            v4i a[2], b[2], r[2];
        };
        // (party, tile) of this wave's current work item, advanced without a division; parties beyond the last are clamped
        // for the loads past the end
        auto load_slot = [&](uint32_t pp, size_t t, int j) {
            const size_t gi = t * 32 + c;
            const size_t g = (size_t)(pp < (uint32_t)a.parties ? pp : (uint32_t)a.parties - 1) * a.G + (gi < a.G ? gi : a.G - 1);
            const int i = 2 * j + h < M ? 2 * j + h : M - 1;  // M odd: the last slot of the upper lanes repeats the last coefficient (unused)
            const size_t off = (g * M + (size_t)i) * 32;
            Slot sl;
            sl.a[0] = *reinterpret_cast<const v4i*>(a.in + off), sl.a[1] = *reinterpret_cast<const v4i*>(a.in + off + 16);
            sl.b[0] = *reinterpret_cast<const v4i*>(a.in_b + off), sl.b[1] = *reinterpret_cast<const v4i*>(a.in_b + off + 16);
            sl.r[0] = *reinterpret_cast<const v4i*>(a.in_r + off), sl.r[1] = *reinterpret_cast<const v4i*>(a.in_r + off + 16);
            return sl;
        };
        auto elem = [](const v4i (&w)[2]) {
            const uint32_t ww[8] = {(uint32_t)w[0][0], (uint32_t)w[0][1], (uint32_t)w[0][2], (uint32_t)w[0][3],
                                    (uint32_t)w[1][0], (uint32_t)w[1][1], (uint32_t)w[1][2], (uint32_t)w[1][3]};
            return F::from_words(ww);
        };
        // x = (a b - r2t) R^-1 (R = 2^261, the Montgomery radix of fr_u29.hpp) as 8 words of a value below 2 r -- any
        // representative below 2^256 serves, the table is linear in the bytes -- from ONE reduction: the 81 limb products of
        // a b and the limbs of 2 r - r2t go into the same 18 columns (U_SUBC2 keeps every limb difference non-negative), then
        // the 72 multiply-adds of the Montgomery reduction.  The factor R^-1 is undone by the TABLE, whose rows are
        // alpha^i R for this kernel (hbmpc_capi.hip: "mfbflyR"), so no operand is ever converted to Montgomery form.
        auto product = [&](const Slot& sl, v4i& lo, v4i& hi) {
            const F::E ea = elem(sl.a), eb = elem(sl.b), er = elem(sl.r);
            F::Acc A;
            F::acc_zero(A);
#pragma unroll
            for (int i = 0; i < 9; ++i) A.c[i] += (uint64_t)(consts::U_SUBC2[i] - er.l[i]);
            F::acc_mac_pinned(A, ea, eb.l);
            const F::E cc = F::canon_loose(F::acc_reduce(A));
            uint32_t w[8];
            F::to_words(cc, w);
#pragma unroll
            for (int k = 0; k < 4; ++k) lo[k] = (int)w[k], hi[k] = (int)w[4 + k];
        };
        uint32_t pp = 0;
        size_t t = (size_t)wg_in_role * WAVES + wave;
        auto settle = [&](uint32_t& q, size_t& u) {
            while (u >= ntiles && q < (uint32_t)a.parties) u -= ntiles, ++q;
        };
        settle(pp, t);
        // slots are requested SD ahead of their use, across the item boundary: while the last slots of an item are multiplied
        // and while its pairs run, the first SD slots of the wave's next item are on their way
        constexpr int SD = TRIPLE_DEPTH;
        static_assert(SD >= 1 && SD <= NS, "slot prefetch depth");
        Slot ring[SD];
        if (pp < (uint32_t)a.parties) {
#pragma unroll
            for (int j = 0; j < SD; ++j) ring[j] = load_slot(pp, t, j);
            if constexpr (STATIC) dropped_stores();
        }
        v4i data[M];
        while (pp < (uint32_t)a.parties) {
            uint32_t ppn = pp;
            size_t tn = t + tstep;
            settle(ppn, tn);
            const size_t tnc = tn < ntiles ? tn : ntiles - 1;
            // positions 0 .. NSP - 1, NSP = the slot count rounded up to the ring size: a slot always returns to the ring entry
            // it had in the previous item (positions >= NS are empty: nothing is multiplied, nothing requested for them)
            constexpr int NSP = (NS + SD - 1) / SD * SD;
#pragma unroll
            for (int j = 0; j < NSP; ++j) {
                v4i lo, hi;
                if (j < NS) product(ring[j % SD], lo, hi);
                if (j + SD < NS) ring[j % SD] = load_slot(pp, t, j + SD);
                else if (j + SD >= NSP) ring[j % SD] = load_slot(ppn, tnc, j + SD - NSP);
                if (j < NS) {
                    // lower lanes hold coefficient 2 j, upper lanes 2 j + 1: the swap leaves (low half | high half) of 2 j in lo
                    // and of 2 j + 1 in hi, each lane with its own 16 bytes
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const auto sw = __builtin_amdgcn_permlane32_swap((uint32_t)lo[k], (uint32_t)hi[k], false, false);
                        lo[k] = (int)sw[0], hi[k] = (int)sw[1];
                    }
                    data[2 * j] = flip(lo);
                    if (2 * j + 1 < M) data[2 * j + 1] = flip(hi);
                }
            }
            pairs_of_tile(t, data, a.out + (size_t)pp * (size_t)a.nout * a.out_stride * 32);
            pp = ppn, t = tn;
        }
    } else {
        // the tile loop of k_mfma_rows: two input register sets, the loads of a wave's NEXT tile issued before it starts on the
        // current one
        v4i setA[M], setB[M];
        size_t t = (size_t)wg_in_role * WAVES + wave;
        if (t < ntiles) {
            load_inputs(t, setA);
            if constexpr (STATIC) dropped_stores();
        }
        while (t < ntiles) {
            if (STATIC || t + tstep < ntiles) load_inputs(t + tstep, setB);
            process_tile(t, setA);
            t += tstep;
            if (t >= ntiles) break;
            if (STATIC || t + tstep < ntiles) load_inputs(t + tstep, setA);
            process_tile(t, setB);
            t += tstep;
        }
    }
}

}  // namespace mf
}  // namespace hbmpc
