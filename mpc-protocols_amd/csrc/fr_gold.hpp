// fr_gold.hpp -- the Goldilocks field p = 2^64 - 2^32 + 1 on the device (SURVEY.md section 8(f) row 4:
// the reference's small field, common/math/goldilocks.rs:4-13 -- ark Fp64<MontBackend>, generator 7,
// two-adicity 32; its RanSha / RanDouSha / TripleGen / RandBit nodes run in it, honeybadger/mod.rs:316-324).
// Same interface as Sat32 / U29 so the generic kernels (Horner evaluation, batch recover, OEC/Gao, mat-vec,
// element-wise) instantiate unchanged.  Elements are 8 bytes (EW = 2 words), always canonical, and there is
// no Montgomery form: "device-constant form" of c is c itself, mont(a, b) = a*b mod p.
// Reduction: 2^64 = 2^32 - 1 (= EPS) and 2^96 = -1 (mod p), so for x = lo + 2^64 (hl + 2^32 hh):
//   x = lo - hh + hl * EPS (mod p).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hbmpc {

#ifndef HB_DEV
#define HB_DEV __device__ __forceinline__
#endif

struct Gold {
    static constexpr int NL = 2;  // words per constant
    static constexpr int EW = 2;  // words per stored element
    template <int LOG, int CNT>
    static constexpr int eval_waves() { return 8; }  // memory-bound: as many waves as fit ...
    template <int LOG, int CNT>
    static constexpr int eval_waves_min() { return 1; }  // ... but never at the price of spills
    template <int LOG, int CNT>
    static constexpr bool eval_interleave() { return false; }
    static constexpr int MAX_DOT_TERMS = 1 << 30;
    static constexpr uint64_t P = 0xFFFFFFFF00000001ull;
    static constexpr uint64_t EPS = 0xFFFFFFFFull;

    struct E {
        uint32_t l[2];
    };
    // Lazy dot products (the role U29's column accumulators play): a term is NOT reduced.  With a = a1 2^32 + a0 and
    // c = c1 2^32 + c0 the four 32 x 32 partial products go to three COLUMNS (a0 c0 | a0 c1 + a1 c0 | a1 c1), each a
    // 64-bit sum with a 32-bit carry count: one v_mad_u64_u32 (carry out) + one v_addc per product, 8 instructions
    // per term, 2^32 terms of headroom, and ONE reduction per dot product folds the columns:
    // sum = C0 + 2^32 C1 + 2^64 C2 (mod p).  (128-bit accumulators through the compiler cost ~20 issue slots per
    // term -- carry juggling through SGPR pairs with hazard nops -- and a reduced multiply-add per term ~45.)
    struct Acc {
        uint64_t c0, c1, c2;
        uint32_t h0, h1, h2;
    };
    static HB_DEV uint64_t u(const E& a) { return ((uint64_t)a.l[1] << 32) | a.l[0]; }
    static HB_DEV E e(uint64_t v) {
        E r = {{(uint32_t)v, (uint32_t)(v >> 32)}};
        return r;
    }
    static HB_DEV E zero() { return e(0); }
    static HB_DEV E load(const uint32_t* __restrict__ p) {
        const uint2 a = *reinterpret_cast<const uint2*>(p);
        E r = {{a.x, a.y}};
        return r;
    }
    static HB_DEV E load_const(const uint32_t* __restrict__ p) {
        E r = {{p[0], p[1]}};
        return r;
    }
    static HB_DEV uint64_t addm(uint64_t a, uint64_t b) {  // a, b < p
        uint64_t s = a + b;
        if (s < a) s += EPS;  // wrapped: + 2^64 = + EPS (cannot wrap again: a + b <= 2p - 2)
        return s >= P ? s - P : s;
    }
    static HB_DEV uint64_t subm(uint64_t a, uint64_t b) {  // a, b < p
        uint64_t d = a - b;
        if (a < b) d -= EPS;  // wrapped: - 2^64 + p = - EPS
        return d;
    }
    static HB_DEV uint64_t mulm(uint64_t a, uint64_t b) {
        const uint64_t lo = a * b, hi = __umul64hi(a, b);
        const uint64_t hh = hi >> 32, hl = hi & EPS;
        uint64_t t0 = lo - hh;
        if (lo < hh) t0 -= EPS;
        const uint64_t t1 = hl * EPS;  // < 2^64
        uint64_t r = t0 + t1;
        if (r < t0) r += EPS;
        return r >= P ? r - P : r;
    }
    static HB_DEV E add(const E& a, const E& b) { return e(addm(u(a), u(b))); }
    template <int K>
    static HB_DEV E sub(const E& a, const E& b) { return e(subm(u(a), u(b))); }
    static HB_DEV E normalize(const E& a) { return a; }
    static HB_DEV E mont(const E& a, const uint32_t* __restrict__ b) { return e(mulm(u(a), ((uint64_t)b[1] << 32) | b[0])); }
    static HB_DEV E mulc(const E& a, const uint32_t* __restrict__ c) { return mont(a, c); }
    static HB_DEV E mulc_u(const E& a, const uint32_t* __restrict__ c) { return mont(a, c); }
    static HB_DEV E mont(const E& a, const E& b) { return e(mulm(u(a), u(b))); }

    static HB_DEV void acc_zero(Acc& A) { A.c0 = A.c1 = A.c2 = 0, A.h0 = A.h1 = A.h2 = 0; }
    // One term = ONE asm statement (the compiler pads hazard nops between separate asm statements that touch vcc):
    // four (v_mad_u64_u32 with carry out, v_addc into the column's carry count) pairs.
#define HB_GOLD_MAC(CK)                                                                                              \
    asm("v_mad_u64_u32 %0, vcc, %6, %8, %0\n\tv_addc_co_u32_e32 %3, vcc, 0, %3, vcc\n\t"                            \
        "v_mad_u64_u32 %1, vcc, %6, %9, %1\n\tv_addc_co_u32_e32 %4, vcc, 0, %4, vcc\n\t"                            \
        "v_mad_u64_u32 %1, vcc, %7, %8, %1\n\tv_addc_co_u32_e32 %4, vcc, 0, %4, vcc\n\t"                            \
        "v_mad_u64_u32 %2, vcc, %7, %9, %2\n\tv_addc_co_u32_e32 %5, vcc, 0, %5, vcc"                                 \
        : "+v"(A.c0), "+v"(A.c1), "+v"(A.c2), "+v"(A.h0), "+v"(A.h1), "+v"(A.h2)                                     \
        : "v"(a.l[0]), "v"(a.l[1]), CK(c0), CK(c1)                                                                   \
        : "vcc")
    static HB_DEV void acc_mac(Acc& A, const E& a, const uint32_t* __restrict__ c) {
        const uint32_t c0 = c[0], c1 = c[1];
        HB_GOLD_MAC("v");
    }
    // constants from SGPRs: wave-uniform by the caller's promise (they come through the scalar cache)
    static HB_DEV void acc_mac_pinned(Acc& A, const E& a, const uint32_t (&c)[2]) {
        const uint32_t c0 = c[0], c1 = c[1];
        HB_GOLD_MAC("s");
    }
#undef HB_GOLD_MAC
    static HB_DEV void acc_add_hi(Acc& A, const E& x) {
        const uint64_t v = u(x), s = A.c0 + v;
        A.h0 += s < v;
        A.c0 = s;
    }
    static HB_DEV void acc_fold(Acc&) {}
    template <int M_TOTAL>
    static HB_DEV void acc_fold_needed(Acc&) {}
    // x < 2^128 -> canonical residue: x = lo + 2^64 (hl + 2^32 hh) = lo - hh + hl * EPS (mod p)
    static HB_DEV uint64_t reduce128(unsigned __int128 x) {
        const uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
        const uint64_t hh = hi >> 32, hl = hi & EPS;
        uint64_t t0 = lo - hh;
        if (lo < hh) t0 -= EPS;
        const uint64_t t1 = hl * EPS;
        uint64_t r = t0 + t1;
        if (r < t0) r += EPS;
        return r >= P ? r - P : r;
    }
    // a 96-bit column col + 2^64 cnt -> canonical residue
    static HB_DEV uint64_t reduce96(uint64_t col, uint32_t cnt) { return reduce128(((unsigned __int128)cnt << 64) | col); }
    static HB_DEV E acc_reduce(Acc& A) {
        const uint64_t r0 = reduce96(A.c0, A.h0), r1 = reduce96(A.c1, A.h1), r2 = reduce96(A.c2, A.h2);
        // r0 + 2^32 r1 + 2^64 r2 with 2^64 = EPS (mod p): two modular products by constants, one fused 128-bit fold
        const unsigned __int128 t = (unsigned __int128)r0 + ((unsigned __int128)r1 << 32) + (unsigned __int128)r2 * EPS;  // < 2^98
        return e(reduce128(t));
    }

    static HB_DEV E cond_sub_r(const E& x) { return x; }
    static HB_DEV E canon_loose(const E& x) { return x; }
    static HB_DEV void store_lt2r(uint32_t* __restrict__ p, const E& x) { *reinterpret_cast<uint2*>(p) = make_uint2(x.l[0], x.l[1]); }
    static HB_DEV void store_loose(uint32_t* __restrict__ p, const E& x) { store_lt2r(p, x); }
    static HB_DEV bool eq_canon(const E& a, const E& b) { return ((a.l[0] ^ b.l[0]) | (a.l[1] ^ b.l[1])) == 0; }
    static HB_DEV bool is_zero_canon(const E& a) { return (a.l[0] | a.l[1]) == 0; }
};

}  // namespace hbmpc
