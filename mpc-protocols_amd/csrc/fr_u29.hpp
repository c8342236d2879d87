// fr_u29.hpp -- bls12-381 Fr on gfx950 in an UNSATURATED radix-2^29 representation.
//
// Why this shape (measured on MI355X with tools/ubench_isa.hip, profiles/r01_isa_rates.txt):
//   v_mad_u64_u32 issues at ~4.9 cycles per wave-instruction -- the same as v_addc_co_u32 /
//   v_lshl_add_u64 (~4.5) -- and a VCC carry chain additionally needs wait states between the
//   carry write and the carry-in read.  So the cost of a 256-bit modular multiply is its
//   INSTRUCTION COUNT, and carry handling is as expensive as the multiplies themselves.
//   With 9 limbs of 29 bits every 32x32->64 product is < 2^58..2^60, a whole column of a
//   9x9 product (plus the Montgomery m*n column) fits one 64-bit accumulator, and the multiply
//   is a pure chain of v_mad_u64_u32 into a running 64-bit column sum: NO carry instructions.
//   Additions are 9 full-rate v_add_u32 with no carry propagation (lazy limbs).
//
// Representation
//   value = sum l[i] * 2^(29 i), i < 9.   Montgomery radix R = 2^261.
//   "normalised": l[i] < 2^29 (i < 8).    "loose": l[i] < 2^31 (sums of <= 4 normalised values).
//   r = 2^254.86, so 2^261 = 70.5 r: REDC of a T-term dot product of values < r and constants < r
//   is < (T/70.5 + 1) r  -- below 2r for every T <= 70 (the VALUE bound; the 64-bit COLUMN bound is
//   tighter: the accumulator is carry-folded every 6 terms, see MAX_DOT_TERMS).
//   -r^-1 mod 2^29 = 2^29 - 1 (r = 1 mod 2^32), so the Montgomery digit is m = (-acc) & MASK,
//   and r's limb 0 is 1, so m*n0 is an add.
//
// Constants ("C9") are 9 normalised limbs of c*R mod r: mulc(x, C) = x*c mod r with x canonical
// data -- data never needs converting to Montgomery form for constant-matrix maps.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fr_consts.h"

namespace hbmpc {

#define HB_DEV __device__ __forceinline__

struct U29 {
    static constexpr int NL = 9;            // limbs per element / per constant
    static constexpr uint32_t MASK = 0x1fffffffu;
    // A 64-bit column holds 2^64 / 2^58 = 64 products of normalised limbs.  A term adds up to 9 per column,
    // the reduction up to 8 more (m * r) plus carries: 9 T + 9 <= 64  =>  T <= 6 terms between folds.
    static constexpr int EW = 8;  // u32 words per stored element
    // waves per SIMD the register-resident FFT kernels are compiled for: with <= 6 non-zero inputs the inputs of
    // the second-to-last stage are <= 8 distinct values and the interleaved last two stages (kernels_eval.hpp)
    // fit 168 VGPRs; with more inputs all 16 values are live and the kernel stays at 2 waves.
    template <int LOG, int CNT>
    static constexpr int eval_waves() { return (LOG < 4 || CNT <= 6) ? 3 : 2; }
    template <int LOG, int CNT>
    static constexpr int eval_waves_min() { return eval_waves<LOG, CNT>(); }  // forced: the register budget is the point
    template <int LOG, int CNT>
    static constexpr bool eval_interleave() { return eval_waves<LOG, CNT>() == 3; }
    static constexpr int MAX_DOT_TERMS = 6;

    struct E {
        uint32_t l[9];
    };
    struct Acc {
        uint64_t c[18];
    };

    static HB_DEV E zero() {
        E r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.l[i] = 0;
        return r;
    }

    // canonical 8 x u32 words (little endian) -> 9 normalised limbs
    static HB_DEV E from_words(const uint32_t w[8]) {
        E r;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int o = 29 * i, q = o >> 5, s = o & 31;
            uint32_t v = w[q] >> s;
            if (s > 3 && q + 1 < 8) v |= w[q + 1] << (32 - s);
            r.l[i] = v & MASK;
        }
        return r;
    }
    // normalised limbs of a value < 2^256 -> 8 words
    static HB_DEV void to_words(const E& a, uint32_t w[8]) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int o = 32 * q, i = o / 29, s = o % 29;  // s <= 21
            w[q] = (a.l[i] >> s) | (a.l[i + 1] << (29 - s));
        }
    }
    static HB_DEV E load(const uint32_t* __restrict__ p) {  // p: 32-byte aligned canonical element
        const uint4 a = *reinterpret_cast<const uint4*>(p);
        const uint4 b = *reinterpret_cast<const uint4*>(p + 4);
        const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        return from_words(w);
    }
    static HB_DEV E load_const(const uint32_t* __restrict__ p) {  // 9 limbs, already in limb form
        E r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.l[i] = p[i];
        return r;
    }

    // lazy add: limb bounds add, value bounds add
    static HB_DEV E add(const E& a, const E& b) {
        E r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + b.l[i];
        return r;
    }
    // a - b + K*r, b normalised with value < K*r/2 (K = 2: b < r canonical; K = 4: b < 2r; ... K = 64: b < 32r)
    template <int K>
    static HB_DEV E sub(const E& a, const E& b) {
        static_assert(K == 2 || K == 4 || K == 8 || K == 16 || K == 32 || K == 64, "K");
        E r;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const uint32_t c = K == 2    ? consts::U_SUBC2[i]
                               : K == 4  ? consts::U_SUBC4[i]
                               : K == 8  ? consts::U_SUBC8[i]
                               : K == 16 ? consts::U_SUBC16[i]
                               : K == 32 ? consts::U_SUBC32[i]
                                         : consts::U_SUBC64[i];
            r.l[i] = a.l[i] + (c - b.l[i]);
        }
        return r;
    }
    // carry propagation: limbs -> normalised, value unchanged (value must be < 2^261)
    static HB_DEV E normalize(const E& a) {
        E r = a;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            r.l[i + 1] += r.l[i] >> 29;
            r.l[i] &= MASK;
        }
        return r;
    }

    // Montgomery digit of a column sum u that was pre-biased by MASK: ~u & MASK == (-(u - MASK)) & MASK, one
    // v_bfi_b32 ((u & 0) | (~u & MASK)).
    static HB_DEV uint32_t digit_of_biased(uint32_t u_lo) {
        uint32_t m;
        asm("v_bfi_b32 %0, %1, 0, %2" : "=v"(m) : "v"(u_lo), "v"(MASK));
        return m;
    }
    // Montgomery product a*b/R (mod r) by product scanning.  a loose (limbs < 2^31), b normalised
    // (limbs < 2^29).  Result normalised, value < a*b/2^261 + r.
    // Column bound: 9*(2^31*2^29) + 8*2^58 + 2^29 + carry(2^35) < 1.27e19 < 2^64.
    // Digit step.  With t = column + carry the textbook step is m = (-t) & MASK; t += m; carry = t >> 29
    // (five dependent instructions here).  t + m is t rounded UP to a multiple of 2^29, so the carry is
    // (t + MASK) >> 29 whatever m is: every low column starts at MASK instead of 0 (free: it is the addend of
    // its first mad), u = column + carry, carry' = u >> 29 and m = ~u & MASK -- three instructions, and the
    // carry chain is two deep per digit.
    // Plain C++ statement of the product (kept as the readable definition and as a cross-check of the asm
    // version below, which is what the kernels run).
    static HB_DEV E mont_cxx(const E& a, const uint32_t* __restrict__ b) {
        uint64_t c[17];
#pragma unroll
        for (int k = 0; k < 17; ++k) c[k] = k < 9 ? MASK : 0;
#pragma unroll
        for (int j = 0; j < 9; ++j) c[j] += (uint64_t)a.l[0] * b[j];
        uint64_t carry = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const uint64_t u = c[i] + carry;
            const uint32_t m = digit_of_biased((uint32_t)u);
            carry = u >> 29;
            if (i < 8) {
#pragma unroll
                for (int j = 0; j < 9; ++j) c[i + 1 + j] += (uint64_t)a.l[i + 1] * b[j];
            }
#pragma unroll
            for (int j = 1; j < 9; ++j) c[i + j] += (uint64_t)m * consts::U_MOD[j];
        }
        E t;
#pragma unroll
        for (int k = 9; k < 17; ++k) {
            const uint64_t u = c[k] + carry;
            t.l[k - 9] = (uint32_t)u & MASK;
            carry = u >> 29;
        }
        t.l[8] = (uint32_t)carry;
        return t;
    }
    // hipcc regroups the C++ version column by column (one long dependent mad chain per column) whatever
    // the source order: the kernels use the hand-scheduled single-asm-block version (tools/gen_mont_asm.py).
#include "fr_u29_mont_asm.inc"
    static HB_DEV E mont(const E& a, const uint32_t* __restrict__ b) {
        const uint32_t bb[9] = {b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], b[8]};
        return mont_asm_v(a, bb);
    }
    // c must be WAVE-UNIFORM (a table indexed by compile-time or loop constants): it is held in SGPRs
    static HB_DEV E mulc_u(const E& a, const uint32_t* __restrict__ c) {
        const uint32_t bb[9] = {c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8]};
        return mont_asm_s(a, bb);
    }
    static HB_DEV E mulc(const E& a, const uint32_t* __restrict__ c) { return mont(a, c); }
    static HB_DEV E mont(const E& a, const E& b) { return mont(a, b.l); }

    // ---- lazy dot products: acc += a*c (81 mads, no carries); one REDC at the end ------------
    // the nine low columns start at MASK: the bias acc_reduce's digit step expects (see mont; carry-folding
    // moves multiples of 2^29 from column i to column i+1 as units, which leaves every u of the reduction --
    // and so every digit and carry -- unchanged)
    static HB_DEV void acc_zero(Acc& A) {
#pragma unroll
        for (int i = 0; i < 18; ++i) A.c[i] = i < 9 ? MASK : 0;
    }
    // a normalised, c normalised; at most MAX_DOT_TERMS (6) terms between acc_zero/acc_fold
    static HB_DEV void acc_mac(Acc& A, const E& a, const uint32_t* __restrict__ c) {
#pragma unroll
        for (int i = 0; i < 9; ++i)
#pragma unroll
            for (int j = 0; j < 9; ++j) A.c[i + j] += (uint64_t)a.l[i] * c[j];
    }
    // the same, with every product pinned to ONE in-place v_mad_u64_u32 (inline asm): hipcc otherwise
    // re-associates the 81 mads into extra 64-bit temporaries to shorten dependency chains and, in
    // kernels that already hold many elements in registers, spills the accumulators to scratch.
    // Consecutive mads hit different columns (a column is revisited 8 instructions later), so the
    // in-order stream has no dependency stalls to hide.
    static HB_DEV void acc_mac_pinned(Acc& A, const E& a, const uint32_t (&c)[9]) {
        // one asm statement per data limb (9 mads): hipcc pads every inline-asm VALU statement with an
        // s_nop, so fewer, larger statements
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            uint64_t carry_unused;  // the instructions' scalar carry-out (never read)
            asm("v_mad_u64_u32 %0, %9, %10, %11, %0\n\t"
                "v_mad_u64_u32 %1, %9, %10, %12, %1\n\t"
                "v_mad_u64_u32 %2, %9, %10, %13, %2\n\t"
                "v_mad_u64_u32 %3, %9, %10, %14, %3\n\t"
                "v_mad_u64_u32 %4, %9, %10, %15, %4\n\t"
                "v_mad_u64_u32 %5, %9, %10, %16, %5\n\t"
                "v_mad_u64_u32 %6, %9, %10, %17, %6\n\t"
                "v_mad_u64_u32 %7, %9, %10, %18, %7\n\t"
                "v_mad_u64_u32 %8, %9, %10, %19, %8"
                : "+v"(A.c[i]), "+v"(A.c[i + 1]), "+v"(A.c[i + 2]), "+v"(A.c[i + 3]), "+v"(A.c[i + 4]),
                  "+v"(A.c[i + 5]), "+v"(A.c[i + 6]), "+v"(A.c[i + 7]), "+v"(A.c[i + 8]), "=&s"(carry_unused)
                : "v"(a.l[i]), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7]),
                  "v"(c[8]));
        }
    }
    // adds x * R (x loose): REDC then yields (... + x) -- a free "+ x" inside a dot product
    static HB_DEV void acc_add_hi(Acc& A, const E& x) {
#pragma unroll
        for (int i = 0; i < 9; ++i) A.c[9 + i] += x.l[i];
    }
    // column carry propagation (keeps the value, makes room for another MAX_DOT_TERMS terms)
    static HB_DEV void acc_fold(Acc& A) {
#pragma unroll
        for (int i = 0; i < 17; ++i) {
            A.c[i + 1] += A.c[i] >> 29;
            A.c[i] &= MASK;
        }
    }
    // Partial fold for a dot product of M_TOTAL terms: column k receives cnt_k = min(k, 16-k, 8) + 1 products per
    // term, so only columns with cnt_k * M_TOTAL + 9 > 64 can ever overflow; the others are left alone.
    template <int M_TOTAL>
    static HB_DEV void acc_fold_needed(Acc& A) {
#pragma unroll
        for (int i = 0; i < 17; ++i) {
            const int cnt = (i < 8 ? i : (16 - i < 8 ? 16 - i : 8)) + 1;
            if (cnt * M_TOTAL + 9 > 64) {
                A.c[i + 1] += A.c[i] >> 29;
                A.c[i] &= MASK;
            }
        }
    }
    // Montgomery reduction of the 18 columns; result normalised, value < T/R + r
    static HB_DEV E acc_reduce(Acc& A) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const uint32_t m = digit_of_biased((uint32_t)A.c[k]);
            A.c[k + 1] += A.c[k] >> 29;
#pragma unroll
            for (int j = 1; j < 9; ++j) A.c[k + j] += (uint64_t)m * consts::U_MOD[j];
        }
        E t;
#pragma unroll
        for (int k = 9; k < 17; ++k) {
            A.c[k + 1] += A.c[k] >> 29;
            t.l[k - 9] = (uint32_t)A.c[k] & MASK;
        }
        t.l[8] = (uint32_t)A.c[17];
        return t;
    }

    // ---- canonicalisation ---------------------------------------------------------------------
    // x normalised, value < 2r  ->  canonical (< r), normalised
    static HB_DEV E cond_sub_r(const E& x) {
        int32_t d[9];
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int32_t v = (int32_t)x.l[i] - (int32_t)consts::U_MOD[i] + c;
            c = v >> 29;  // arithmetic: 0 or -1 (limb 8 keeps its sign in v itself)
            d[i] = i < 8 ? (v & (int32_t)MASK) : v;
        }
        const uint32_t neg = (uint32_t)(d[8] >> 31);  // all ones when x < r
        E r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.l[i] = ((uint32_t)d[i] & ~neg) | (x.l[i] & neg);
        return r;
    }
    // x loose (limbs < 2^32, value < 2^261): subtract q*r with q estimated from the top limbs.
    // Result normalised and in [0, 2r); it is already < r unless its top limb reaches r's top limb
    // (probability ~2^-19 per value: q is then one too small, or the value sits within 2^232 of r).
    static HB_DEV E reduce_top(const E& x) {
        // top = floor(x / 2^232) up to an error of +1; r_top = 0x73eda7 (floor(r / 2^232)).
        const uint32_t top = x.l[8] + (x.l[7] >> 29);
        // q = floor(top / (r_top + 1)) <= floor(x / r), and >= floor(x / r) - 1
        constexpr uint64_t RECIP = (1ull << 45) / (uint64_t)(consts::U_MOD[8] + 1);
        const int32_t nq = -(int32_t)(uint32_t)(((uint64_t)top * RECIP) >> 45);
        E y;
        int64_t acc = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            acc += (int64_t)x.l[i];
            acc += (int64_t)nq * (int64_t)(int32_t)consts::U_MOD[i];
            if (i < 8) {
                y.l[i] = (uint32_t)acc & MASK;
                acc >>= 29;
            } else {
                y.l[i] = (uint32_t)acc;
            }
        }
        return y;
    }
    static HB_DEV bool maybe_ge_r(const E& y) { return y.l[8] >= consts::U_MOD[8]; }
    static HB_DEV E canon_loose(const E& x) {
        E y = reduce_top(x);
        // exact: y_8 < r_8 implies y < r_8 * 2^232 <= r.  The full conditional subtraction runs only for
        // waves in which some lane hits the rare case.
        if (__builtin_expect(__any(maybe_ge_r(y)) != 0, 0)) y = cond_sub_r(y);
        return y;
    }
    // canonical store of a value that is normalised and < 2r
    static HB_DEV void store_lt2r(uint32_t* __restrict__ p, const E& x) {
        const E c = cond_sub_r(x);
        uint32_t w[8];
        to_words(c, w);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
        *reinterpret_cast<uint4*>(p + 4) = make_uint4(w[4], w[5], w[6], w[7]);
    }
    // canonical store of any loose value < 2^261
    static HB_DEV void store_loose(uint32_t* __restrict__ p, const E& x) {
        const E c = canon_loose(x);
        uint32_t w[8];
        to_words(c, w);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
        *reinterpret_cast<uint4*>(p + 4) = make_uint4(w[4], w[5], w[6], w[7]);
    }
    // exact equality of two values (any loose forms)
    static HB_DEV bool eq_canon(const E& a_canon, const E& b_canon) {
        uint32_t d = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) d |= a_canon.l[i] ^ b_canon.l[i];
        return d == 0;
    }
    static HB_DEV bool is_zero_canon(const E& a) {
        uint32_t d = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) d |= a.l[i];
        return d == 0;
    }
};

}  // namespace hbmpc
