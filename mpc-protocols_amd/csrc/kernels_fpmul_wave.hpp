// FPMulNode for all parties of a SMALL batch in one launch: a wave per batch element (fpmul/fpmul.rs:61-110).
//
// At the batch sizes the protocols run a fixed-point multiplication at (a few hundred to a few thousand elements) the five
// launches of the separate steps -- the shares Multiply opens, their decode, finalize_mul + r' + the share TruncPr opens, its
// decode, TruncPr's last step -- cost 4 - 13 us each whatever the batch (profiles/r04_small_batch_fpmul.txt): a lone wave
// per SIMD walks a chain of v_mad_u64_u32 at ~10 cycles each (profiles/r01_isa_rates.txt) behind a memory round trip, five times over.  With all parties on one
// device every step of an element depends on that element alone, so one wave can take it from the triple to the output
// shares: one round trip of loads, and the multiplications of a step spread over the lanes the step would leave idle
// (a lane per table row in the decodes; a lane per (party, product) between them -- every product of a step is the same
// instruction stream, "sum of a_k * c_k, then one REDC", so the lanes stay converged).
//
// The bytes of every buffer a caller can see are those of the separate launches (tests/test_gpu_pipelines.py): each value is
// stored canonical, and a chunk that fails its verification opens to zero and is counted, exactly as there.
#pragma once
#include "kernels_recover.hpp"

namespace hbmpc {

struct FpmulWaveArgs {
    const uint32_t *ta, *tb, *tc, *x, *y;  // [party][N]: the triple's shares, the factors' shares
    const uint32_t *r_bits, *r_int;        // [party][m][N], [party][N]
    const uint32_t* pow2;                  // [m] constants 2^j
    const uint32_t* tab;                   // [t verify rows | P(0) row | P(0) * R row][t + 1] constants (hbmpc_capi.hip, fpmul_wave_table)
    uint32_t *de_out;                      // [2 N]: the opened a - x, then the opened b - y
    uint32_t *z, *r_dash, *open_sh, *out;  // [party][N]
    uint32_t* c_open;                      // [N]
    uint8_t* status;                       // [2 N] as the two decodes leave it: [0, N) the second open's, [N, 2 N) the b - y half of the first
    uint32_t *summary_first, *summary;     // the two opens' summaries
    uint32_t* counters;                    // the stream's decode counters, zero at the start and at the end ([24], [25]: the first open)
    size_t N;
    int parties, m, needed, M;             // needed = 2 t + 1 senders, M = t + 1
    int mask_bits;                         // TruncPr's modulus 2^m as a bit count (min(m, 256))
    int lk1, lk3;                          // log2 of the lanes that share a table row's products in the first / second open (0 .. 2)
    RowsArg rows;                          // rows[i] = party id of the i-th lowest sender
    uint32_t c0[9], c1[9], cinv[9];        // 2^m (constant form), 2^(k-1) (plain limbs), 2^-m (constant form)
};

// LDS words of one workgroup (4 elements): per wave the senders' two values, the parties' operands, the products, the open
// shares and the broadcast values; then pow2 | c0 | the table
struct FpmulWaveLds {
    int per_wave, ysd, yse, ops, res, val, bc, consts, tab, total;
    __host__ __device__ FpmulWaveLds(int needed, int parties, int m, int tab_words) {
        ysd = 0, yse = ysd + needed * 12;       // limbs (12-word stride)
        ops = yse + needed * 12;                // [4 + m][party] canonical words: y, x, c, r_int, bits
        res = ops + (4 + m) * parties * 8;      // [4][party] limbs
        val = res + 4 * parties * 12;           // [party] limbs
        bc = val + parties * 12;                // d, e (8 words each) | d R, e R (12 each) | the second open (8)
        per_wave = bc + 16 + 24 + 8;
        consts = 4 * per_wave;                  // [m + 1][12]: pow2 then c0
        tab = consts + (m + 1) * 12;
        total = tab + ((tab_words + 3) & ~3);
    }
};

template <class F>
__global__ __launch_bounds__(256) void k_fpmul_wave(FpmulWaveArgs a) {
    using E = typename F::E;
    static_assert(F::EW == 8 && F::NL == 9, "U29 only");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t g_raw = (size_t)blockIdx.x * 4 + wave;
    const bool live = g_raw < a.N;
    const size_t g = live ? g_raw : a.N - 1;
    const int M = a.M, nv = a.needed - M, P = a.parties;
    const int tab_words = (nv + 2) * M * 9;
    const FpmulWaveLds L(a.needed, P, a.m, tab_words);
    uint32_t* W = lds + (size_t)wave * L.per_wave;
    uint32_t *ysd = W + L.ysd, *yse = W + L.yse, *ops = W + L.ops, *res = W + L.res, *val = W + L.val, *bc = W + L.bc;
    uint32_t *cst = lds + L.consts, *tab = lds + L.tab;

    auto put_limbs = [&](uint32_t* dst, const E& v) {
#pragma unroll
        for (int i = 0; i < 9; ++i) dst[i] = v.l[i];
    };
    auto put_words = [&](uint32_t* dst, const E& canon) {
        uint32_t w[8];
        F::to_words(canon, w);
        *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
        *reinterpret_cast<uint4*>(dst + 4) = make_uint4(w[4], w[5], w[6], w[7]);
    };

    // ---- every global load of the element, then the LDS writes -------------------------------------------------------------
    const int tq = tab_words >> 2;
    uint4 t0 = make_uint4(0, 0, 0, 0), t1 = t0;
    if ((int)threadIdx.x < tq) t0 = reinterpret_cast<const uint4*>(a.tab)[threadIdx.x];
    if ((int)threadIdx.x + 256 < tq) t1 = reinterpret_cast<const uint4*>(a.tab)[threadIdx.x + 256];
    uint32_t cw = 0;  // pow2 [m][9] and c0 at a 12-word stride
    const int ncw = (a.m + 1) * 9;
    if ((int)threadIdx.x < ncw) cw = (int)threadIdx.x < a.m * 9 ? a.pow2[threadIdx.x] : a.c0[threadIdx.x - a.m * 9];
    E sa = F::zero(), sx = F::zero(), sb = F::zero(), sy = F::zero();
    if (lane < a.needed) {
        const size_t ip = (size_t)row_of_lane(a.rows, lane) * a.N + g;
        sa = F::load(a.ta + ip * 8), sx = F::load(a.x + ip * 8), sb = F::load(a.tb + ip * 8), sy = F::load(a.y + ip * 8);
    }
    // operand item q of party p: 0 y, 1 x, 2 c, 3 r_int, 4 + j bit j
    const int nops = (4 + a.m) * P;
    auto op_src = [&](int it) -> const uint32_t* {
        const int q = it / P, p = it - q * P;
        const uint32_t* base = q == 0 ? a.y : q == 1 ? a.x : q == 2 ? a.tc : a.r_int;
        return q < 4 ? base + ((size_t)p * a.N + g) * 8 : a.r_bits + (((size_t)p * a.m + (q - 4)) * a.N + g) * 8;
    };
    uint4 o0[2] = {t0, t0}, o1[2] = {t0, t0};
    if (lane < nops) {
        const uint32_t* s = op_src(lane);
        o0[0] = *reinterpret_cast<const uint4*>(s), o0[1] = *reinterpret_cast<const uint4*>(s + 4);
    }
    if (lane + 64 < nops) {
        const uint32_t* s = op_src(lane + 64);
        o1[0] = *reinterpret_cast<const uint4*>(s), o1[1] = *reinterpret_cast<const uint4*>(s + 4);
    }
    if ((int)threadIdx.x < tq) reinterpret_cast<uint4*>(tab)[threadIdx.x] = t0;
    if ((int)threadIdx.x + 256 < tq) reinterpret_cast<uint4*>(tab)[threadIdx.x + 256] = t1;
    for (int q = threadIdx.x + 512; q < tq; q += 256) reinterpret_cast<uint4*>(tab)[q] = reinterpret_cast<const uint4*>(a.tab)[q];
    for (int w = (tq << 2) + threadIdx.x; w < tab_words; w += 256) tab[w] = a.tab[w];
    if ((int)threadIdx.x < ncw) cst[(threadIdx.x / 9) * 12 + threadIdx.x % 9] = cw;
    for (int w = threadIdx.x + 256; w < ncw; w += 256) cst[(w / 9) * 12 + w % 9] = w < a.m * 9 ? a.pow2[w] : a.c0[w - a.m * 9];
    if (lane < a.needed) {  // the shares Multiply opens (mul/multiplication.rs:417-426), canonical as k_beaver_open_pair stores them
        put_limbs(ysd + lane * 12, F::canon_loose(F::template sub<2>(sa, sx)));
        put_limbs(yse + lane * 12, F::canon_loose(F::template sub<2>(sb, sy)));
    }
    if (lane < nops) {
        *reinterpret_cast<uint4*>(ops + lane * 8) = o0[0];
        *reinterpret_cast<uint4*>(ops + lane * 8 + 4) = o0[1];
    }
    if (lane + 64 < nops) {
        *reinterpret_cast<uint4*>(ops + (lane + 64) * 8) = o1[0];
        *reinterpret_cast<uint4*>(ops + (lane + 64) * 8 + 4) = o1[1];
    }
    for (int it = lane + 128; it < nops; it += 64) {
        const uint32_t* s = op_src(it);
        *reinterpret_cast<uint4*>(ops + it * 8) = *reinterpret_cast<const uint4*>(s);
        *reinterpret_cast<uint4*>(ops + it * 8 + 4) = *reinterpret_cast<const uint4*>(s + 4);
    }
    __syncthreads();

    // a table row times the chunk's M values (LDS limbs), the products shared by 2^lk adjacent lanes (kernels_recover.hpp)
    auto dot = [&](auto&& y_of, const uint32_t* row, int lk, int sidx) -> E {
        return dot_shared<F>([&](int i) { return F::load_const(y_of(i)); }, row, M, lk, sidx);
    };
    // ---- the first open: a - x in lanes 0 .. 31, b - y in lanes 32 .. 63; row r of the table per lane -----------------------
    // (r < nv verify rows; nv: P(0); nv + 1: P(0) R, what finalize_mul multiplies by)
    {
        const int h = lane >> 5, r = (lane & 31) >> a.lk1, sidx = lane & ((1 << a.lk1) - 1);
        const uint32_t* ys = h ? yse : ysd;
        bool bad = false;
        E kept = F::zero();
        if (r < nv + 2) {  // whole quads: the lanes that share a row are all in or all out
            kept = dot([&](int i) { return ys + i * 12; }, tab + (size_t)r * M * 9, a.lk1, sidx);
            if (r < nv) bad = !F::eq_canon(F::canon_loose(kept), F::load_const(ys + (M + r) * 12));
        }
        const unsigned long long vote = __ballot(bad);
        const bool ok_d = (uint32_t)vote == 0, ok_e = (uint32_t)(vote >> 32) == 0, ok = h ? ok_e : ok_d;
        if (r == nv && sidx == 0) {
            const E v = ok ? F::canon_loose(kept) : F::zero();
            put_words(bc + h * 8, v);
            if (live) put_words(a.de_out + ((size_t)h * a.N + g) * 8, v);
            if (live && h && a.status) a.status[a.N + g] = ok ? 0 : (uint8_t)DecodingError;
        }
        if (r == nv + 1 && sidx == 0) put_limbs(bc + 16 + h * 12, ok ? kept : F::zero());
        if (live && lane == 0 && !(ok_d && ok_e)) {
            atomicAdd(a.counters + 24, (ok_d ? 0u : 1u) + (ok_e ? 0u : 1u));
            atomicMax(a.counters + 25, 0xffffffffu - (uint32_t)(ok_d ? a.N + g : g));
        }
    }
    __syncthreads();

    // ---- finalize_mul (multiplication.rs:57-100), r' (truncpr.rs:277-283), 2^m r_int: a product per lane ---------------------
    //   q = 0: (e + y_p) d R    1: x_p e R    2: r_int_p 2^m    3: sum_j bit_pj 2^j
    for (int task = lane; task < 4 * P; task += 64) {
        const int q = task / P, p = task - q * P;
        const int terms = q == 3 ? a.m : 1;
        const uint32_t* cs = q == 0 ? bc + 16 : q == 1 ? bc + 28 : q == 2 ? cst + a.m * 12 : cst;
        typename F::Acc acc;
        F::acc_zero(acc);
        int pending = 0;
        for (int k = 0; k < terms; ++k) {
            if (pending == F::MAX_DOT_TERMS) {
                F::acc_fold(acc);
                pending = 1;
            }
            E v = F::load(ops + ((q == 0 ? 0 : q == 1 ? 1 : q == 2 ? 3 : 4 + k) * P + p) * 8);
            if (q == 0) v = F::normalize(F::add(v, F::load(bc + 8)));
            F::acc_mac(acc, v, cs + k * 12);
            ++pending;
        }
        F::acc_fold(acc);
        put_limbs(res + (q * P + p) * 12, F::acc_reduce(acc));
    }
    __syncthreads();
    E zc = F::zero(), rd = F::zero();  // party `lane`'s z and r', kept for TruncPr's last step
    for (int p = lane; p < P; p += 64) {
        E acc = F::template sub<4>(F::load(ops + (2 * P + p) * 8), F::load_const(res + p * 12));
        acc = F::template sub<4>(acc, F::load_const(res + (P + p) * 12));
        zc = F::canon_loose(acc);
        rd = F::canon_loose(F::load_const(res + (3 * P + p) * 12));
        E o = F::add(zc, F::load_const(a.c1));
        o = F::add(o, F::load_const(res + (2 * P + p) * 12));
        o = F::canon_loose(F::add(o, rd));
        put_limbs(val + p * 12, o);
        if (live) {
            const size_t ip = (size_t)p * a.N + g;
            put_words(a.z + ip * 8, zc);
            put_words(a.r_dash + ip * 8, rd);
            put_words(a.open_sh + ip * 8, o);
        }
    }
    __syncthreads();

    // ---- the second open (truncpr.rs:215) ---------------------------------------------------------------------------------
    {
        bool bad = false;
        E kept = F::zero();
        const int r = lane >> a.lk3, sidx = lane & ((1 << a.lk3) - 1);
        if (r < nv + 1) {
            kept = dot([&](int i) { return val + a.rows[i] * 12; }, tab + (size_t)r * M * 9, a.lk3, sidx);
            if (r < nv) bad = !F::eq_canon(F::canon_loose(kept), F::load_const(val + row_of_lane(a.rows, M + r) * 12));
        }
        const bool ok = __ballot(bad) == 0;
        if (r == nv && sidx == 0) {
            const E v = ok ? F::canon_loose(kept) : F::zero();
            put_words(bc + 40, v);
            if (live) {
                put_words(a.c_open + g * 8, v);
                if (a.status) a.status[g] = ok ? 0 : (uint8_t)DecodingError;
                if (!ok) {
                    atomicAdd(a.counters, 1u);
                    atomicMax(a.counters + 1, 0xffffffffu - (uint32_t)g);
                }
            }
        }
    }
    __syncthreads();

    // ---- TruncPr's last step (truncpr.rs:216-220, fpmul/mod.rs:381-406): (z - ((c mod 2^m) - r')) 2^-m ------------------------
    if (P <= 64) {
        if (lane < P) {
            uint32_t w[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int lo_bit = 32 * q;
                const uint32_t mask = a.mask_bits >= lo_bit + 32 ? 0xffffffffu : (a.mask_bits <= lo_bit ? 0u : ((1u << (a.mask_bits - lo_bit)) - 1u));
                w[q] = bc[40 + q] & mask;
            }
            E tt = F::template sub<2>(rd, F::from_words(w));
            tt = F::add(tt, zc);
            const E o = F::mulc(tt, a.cinv);
            if (live) F::store_lt2r(a.out + ((size_t)lane * a.N + g) * 8, o);
        }
    }

    // ---- the summaries: the last workgroup turns the counters into them and leaves the counters at zero ------------------------
    __syncthreads();
    if (threadIdx.x != 0) return;
    __threadfence();
    const unsigned nblocks = gridDim.x, sub = blockIdx.x % DIRECT_FAN, quota = nblocks / DIRECT_FAN + (sub < nblocks % DIRECT_FAN ? 1u : 0u);
    if (atomicAdd(a.counters + 8 + sub, 1u) != quota - 1) return;
    const unsigned groups = nblocks < DIRECT_FAN ? nblocks : DIRECT_FAN;
    if (atomicAdd(a.counters + 3, 1u) != groups - 1) return;
    __threadfence();
#pragma unroll
    for (unsigned k = 0; k < DIRECT_FAN; ++k) store_handoff(a.counters + 8 + k, 0u);
    const uint32_t f1 = load_handoff(a.counters + 24), l1 = load_handoff(a.counters + 25);
    const uint32_t f2 = load_handoff(a.counters), l2 = load_handoff(a.counters + 1);
    if (a.summary_first) {
        a.summary_first[0] = f1, a.summary_first[1] = f1;
        a.summary_first[2] = f1 ? 0xffffffffu - l1 : 0xffffffffu;
        a.summary_first[3] = f1 ? (uint32_t)DecodingError : 0u;
    }
    if (a.summary) {
        a.summary[0] = f2, a.summary[1] = f2;
        a.summary[2] = f2 ? 0xffffffffu - l2 : 0xffffffffu;
        a.summary[3] = f2 ? (uint32_t)DecodingError : 0u;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) store_handoff(a.counters + k, 0u);
    store_handoff(a.counters + 24, 0u);
    store_handoff(a.counters + 25, 0u);
}

}  // namespace hbmpc
