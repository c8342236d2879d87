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
    // Lazy dot products (the role U29's column accumulators play): a term is NOT reduced.  With a = a1 2^32 + a0 the
    // two partial products a0*c and a1*c are < 2^96 each and are summed in two 128-bit accumulators -- plain
    // add-with-carry chains, 2^32 terms of headroom -- and one reduction per dot product folds them:
    // sum = lo + 2^32 hi (mod p).  (A reduced multiply-add per term costs ~45 instructions here, this ~14.)
    struct Acc {
        unsigned __int128 lo, hi;
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

    static HB_DEV void acc_zero(Acc& A) { A.lo = 0, A.hi = 0; }
    static HB_DEV void acc_mac_u64(Acc& A, uint64_t a, uint64_t c) {
        A.lo += (unsigned __int128)(uint32_t)a * c;
        A.hi += (unsigned __int128)(uint32_t)(a >> 32) * c;
    }
    static HB_DEV void acc_mac(Acc& A, const E& a, const uint32_t* __restrict__ c) { acc_mac_u64(A, u(a), ((uint64_t)c[1] << 32) | c[0]); }
    static HB_DEV void acc_mac_pinned(Acc& A, const E& a, const uint32_t (&c)[2]) { acc_mac_u64(A, u(a), ((uint64_t)c[1] << 32) | c[0]); }
    static HB_DEV void acc_add_hi(Acc& A, const E& x) { A.lo += u(x); }
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
    static HB_DEV E acc_reduce(Acc& A) { return e(addm(reduce128(A.lo), mulm(reduce128(A.hi), 1ull << 32))); }

    static HB_DEV E cond_sub_r(const E& x) { return x; }
    static HB_DEV E canon_loose(const E& x) { return x; }
    static HB_DEV void store_lt2r(uint32_t* __restrict__ p, const E& x) { *reinterpret_cast<uint2*>(p) = make_uint2(x.l[0], x.l[1]); }
    static HB_DEV void store_loose(uint32_t* __restrict__ p, const E& x) { store_lt2r(p, x); }
    static HB_DEV bool eq_canon(const E& a, const E& b) { return ((a.l[0] ^ b.l[0]) | (a.l[1] ^ b.l[1])) == 0; }
    static HB_DEV bool is_zero_canon(const E& a) { return (a.l[0] | a.l[1]) == 0; }
};

}  // namespace hbmpc
