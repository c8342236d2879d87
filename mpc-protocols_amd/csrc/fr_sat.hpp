// fr_sat.hpp -- bls12-381 Fr on the device with SATURATED 8 x 32-bit limbs (Montgomery radix 2^256).
// The straightforward formulation (operand-scanning CIOS, fully reduced after every operation).
// It exists as the A/B partner and cross-check of fr_u29.hpp: every kernel is a template over the
// field implementation, tests run both, the fast one (U29) is what the C ABI dispatches to.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fr_consts.h"

namespace hbmpc {

#ifndef HB_DEV
#define HB_DEV __device__ __forceinline__
#endif

struct Sat32 {
    static constexpr int NL = 8;
    static constexpr int EW = 8;  // u32 words per stored element
    template <int LOG, int CNT>
    static constexpr int eval_waves() { return 2; }
    template <int LOG, int CNT>
    static constexpr int eval_waves_min() { return 1; }
    template <int LOG, int CNT>
    static constexpr bool eval_interleave() { return false; }
    static constexpr int MAX_DOT_TERMS = 1 << 30;

    struct E {
        uint32_t l[8];
    };
    struct Acc {
        E s;
    };

    static HB_DEV E zero() {
        E r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.l[i] = 0;
        return r;
    }
    static HB_DEV E load(const uint32_t* __restrict__ p) {
        const uint4 a = *reinterpret_cast<const uint4*>(p);
        const uint4 b = *reinterpret_cast<const uint4*>(p + 4);
        E r = {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
        return r;
    }
    static HB_DEV E load_const(const uint32_t* __restrict__ p) {
        E r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.l[i] = p[i];
        return r;
    }
    static HB_DEV E csub(const uint32_t t[8], uint32_t top) {  // t + top*2^256 in [0, 2r) -> [0, r)
        uint32_t s[8];
        uint64_t br = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint64_t d = (uint64_t)t[j] - consts::S_MOD[j] - br;
            s[j] = (uint32_t)d;
            br = (d >> 32) & 1;
        }
        const bool ge = (top != 0) || (br == 0);
        E r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r.l[j] = ge ? s[j] : t[j];
        return r;
    }
    static HB_DEV E add(const E& a, const E& b) {
        uint32_t t[8];
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            c += (uint64_t)a.l[j] + b.l[j];
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        return csub(t, (uint32_t)c);
    }
    template <int K>
    static HB_DEV E sub(const E& a, const E& b) {
        uint32_t t[8];
        uint64_t br = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint64_t d = (uint64_t)a.l[j] - b.l[j] - br;
            t[j] = (uint32_t)d;
            br = (d >> 32) & 1;
        }
        const uint32_t mask = 0u - (uint32_t)br;
        E r;
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            c += (uint64_t)t[j] + (consts::S_MOD[j] & mask);
            r.l[j] = (uint32_t)c;
            c >>= 32;
        }
        return r;
    }
    static HB_DEV E normalize(const E& a) { return a; }
    static HB_DEV E mont(const E& a, const uint32_t* __restrict__ b) {
        uint32_t t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint64_t c = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint64_t x = (uint64_t)a.l[j] * b[i] + t[j] + c;
                t[j] = (uint32_t)x;
                c = x >> 32;
            }
            uint64_t x = (uint64_t)t[8] + c;
            t[8] = (uint32_t)x;
            t[9] = (uint32_t)(x >> 32);
            const uint32_t m = 0u - t[0];  // -r^-1 mod 2^32 = 0xffffffff
            c = ((uint64_t)m * consts::S_MOD[0] + t[0]) >> 32;
#pragma unroll
            for (int j = 1; j < 8; ++j) {
                x = (uint64_t)m * consts::S_MOD[j] + t[j] + c;
                t[j - 1] = (uint32_t)x;
                c = x >> 32;
            }
            x = (uint64_t)t[8] + c;
            t[7] = (uint32_t)x;
            t[8] = t[9] + (uint32_t)(x >> 32);
        }
        return csub(t, t[8]);
    }
    static HB_DEV E mulc(const E& a, const uint32_t* __restrict__ c) { return mont(a, c); }
    static HB_DEV E mulc_u(const E& a, const uint32_t* __restrict__ c) { return mont(a, c); }  // uniform-constant form
    static HB_DEV E mont(const E& a, const E& b) { return mont(a, b.l); }

    static HB_DEV void acc_zero(Acc& A) { A.s = zero(); }
    static HB_DEV void acc_mac(Acc& A, const E& a, const uint32_t* __restrict__ c) { A.s = add(A.s, mont(a, c)); }
    static HB_DEV void acc_mac_pinned(Acc& A, const E& a, const uint32_t (&c)[8]) { A.s = add(A.s, mont(a, c)); }
    static HB_DEV void acc_add_hi(Acc& A, const E& x) { A.s = add(A.s, x); }
    static HB_DEV void acc_fold(Acc&) {}
    template <int M_TOTAL>
    static HB_DEV void acc_fold_needed(Acc&) {}
    static HB_DEV E acc_reduce(Acc& A) { return A.s; }

    static HB_DEV E cond_sub_r(const E& x) { return x; }
    static HB_DEV E canon_loose(const E& x) { return x; }
    static HB_DEV void store_lt2r(uint32_t* __restrict__ p, const E& x) {
        *reinterpret_cast<uint4*>(p) = make_uint4(x.l[0], x.l[1], x.l[2], x.l[3]);
        *reinterpret_cast<uint4*>(p + 4) = make_uint4(x.l[4], x.l[5], x.l[6], x.l[7]);
    }
    static HB_DEV void store_loose(uint32_t* __restrict__ p, const E& x) { store_lt2r(p, x); }
    static HB_DEV bool eq_canon(const E& a, const E& b) {
        uint32_t d = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) d |= a.l[i] ^ b.l[i];
        return d == 0;
    }
    static HB_DEV bool is_zero_canon(const E& a) {
        uint32_t d = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) d |= a.l[i];
        return d == 0;
    }
};

}  // namespace hbmpc
