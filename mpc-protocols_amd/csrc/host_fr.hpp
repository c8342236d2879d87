// host_fr.hpp -- host-side bls12-381 Fr, used ONLY to build the small constant tables the kernels
// stage (domain elements, twiddles, Vandermonde rows, Lagrange bases, verify matrices) and to
// convert them into each device representation.  It never touches batch data: there is no CPU
// data path in this library.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <vector>

namespace hbmpc {

struct HFr {  // Montgomery form, 4 x 64-bit limbs, radix 2^256
    uint64_t l[4];
    typedef unsigned __int128 u128;
    static constexpr uint64_t MOD[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL,
                                        0x73eda753299d7d48ULL};
    static constexpr uint64_t INV = 0xfffffffeffffffffULL;
    static constexpr size_t EBYTES = 32;
    static HFr raw(uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
        HFr r;
        r.l[0] = a, r.l[1] = b, r.l[2] = c, r.l[3] = d;
        return r;
    }
    static HFr zero() { return raw(0, 0, 0, 0); }
    static HFr r2() { return raw(0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL); }
    static HFr one() { return raw(0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL); }
    static bool geq(const uint64_t a[4]) {
        for (int i = 3; i >= 0; --i) {
            if (a[i] > MOD[i]) return true;
            if (a[i] < MOD[i]) return false;
        }
        return true;
    }
    static void subm(uint64_t a[4]) {
        u128 br = 0;
        for (int i = 0; i < 4; ++i) {
            u128 d = (u128)a[i] - MOD[i] - br;
            a[i] = (uint64_t)d;
            br = (d >> 64) & 1;
        }
    }
    bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
    bool operator==(const HFr& o) const { return memcmp(l, o.l, 32) == 0; }
    HFr operator+(const HFr& b) const {
        HFr o;
        u128 c = 0;
        for (int i = 0; i < 4; ++i) {
            c += (u128)l[i] + b.l[i];
            o.l[i] = (uint64_t)c;
            c >>= 64;
        }
        if (geq(o.l)) subm(o.l);
        return o;
    }
    HFr operator-(const HFr& b) const {
        HFr o;
        u128 br = 0;
        for (int i = 0; i < 4; ++i) {
            u128 d = (u128)l[i] - b.l[i] - br;
            o.l[i] = (uint64_t)d;
            br = (d >> 64) & 1;
        }
        if (br) {
            u128 c = 0;
            for (int i = 0; i < 4; ++i) {
                c += (u128)o.l[i] + MOD[i];
                o.l[i] = (uint64_t)c;
                c >>= 64;
            }
        }
        return o;
    }
    HFr neg() const { return zero() - *this; }
    HFr operator*(const HFr& b) const {
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; ++i) {
            u128 c = 0;
            for (int j = 0; j < 4; ++j) {
                c += (u128)l[j] * b.l[i] + t[j];
                t[j] = (uint64_t)c;
                c >>= 64;
            }
            c += t[4];
            t[4] = (uint64_t)c;
            t[5] = (uint64_t)(c >> 64);
            uint64_t m = t[0] * INV;
            c = ((u128)m * MOD[0] + t[0]) >> 64;
            for (int j = 1; j < 4; ++j) {
                c += (u128)m * MOD[j] + t[j];
                t[j - 1] = (uint64_t)c;
                c >>= 64;
            }
            c += t[4];
            t[3] = (uint64_t)c;
            t[4] = t[5] + (uint64_t)(c >> 64);
        }
        HFr o = raw(t[0], t[1], t[2], t[3]);
        if (t[4] || geq(o.l)) subm(o.l);
        return o;
    }
    static HFr from_canon(const uint64_t c[4]) { return raw(c[0], c[1], c[2], c[3]) * r2(); }
    static HFr from_u64(uint64_t v) { return raw(v, 0, 0, 0) * r2(); }
    void to_canon(uint64_t c[4]) const {
        HFr o = *this * raw(1, 0, 0, 0);
        memcpy(c, o.l, 32);
    }
    HFr pow(const uint64_t e[4]) const {
        HFr acc = one();
        for (int i = 255; i >= 0; --i) {
            acc = acc * acc;
            if ((e[i >> 6] >> (i & 63)) & 1) acc = acc * *this;
        }
        return acc;
    }
    HFr pow_u64(uint64_t e) const {
        uint64_t ee[4] = {e, 0, 0, 0};
        return pow(ee);
    }
    static void inv_exponent(uint64_t e[4]) { e[0] = MOD[0] - 2, e[1] = MOD[1], e[2] = MOD[2], e[3] = MOD[3]; }
    HFr inv() const {
        uint64_t e[4];
        inv_exponent(e);
        return pow(e);
    }
    // 7^((r-1)/2^32): GENERATOR = 7, two-adicity 32
    static HFr two_adic_root() {
        const uint64_t rm1[4] = {MOD[0] - 1, MOD[1], MOD[2], MOD[3]};
        uint64_t e[4];
        e[0] = (rm1[0] >> 32) | (rm1[1] << 32);
        e[1] = (rm1[1] >> 32) | (rm1[2] << 32);
        e[2] = (rm1[2] >> 32) | (rm1[3] << 32);
        e[3] = rm1[3] >> 32;
        return from_u64(7).pow(e);
    }
    // ---- device constant formats ----
    // value v (this, Montgomery 2^256) -> limbs of v * Rdev mod r in the device representation
    void to_sat32(uint32_t out[8]) const {  // v * 2^256: the Montgomery limbs themselves
        for (int i = 0; i < 4; ++i) {
            out[2 * i] = (uint32_t)l[i];
            out[2 * i + 1] = (uint32_t)(l[i] >> 32);
        }
    }
    void to_u29(uint32_t out[9]) const {  // v * 2^261 = (v * 2^256) * 32
        HFr x = *this;
        for (int k = 0; k < 5; ++k) x = x + x;  // Montgomery residue of v scaled by 32
        // x.l now holds (v * 2^256 * 32 mod r) as a plain integer
        for (int i = 0; i < 9; ++i) {
            const int o = 29 * i, q = o >> 6, s = o & 63;
            uint64_t v = x.l[q] >> s;
            if (s > 64 - 29 && q + 1 < 4) v |= x.l[q + 1] << (64 - s);
            out[i] = (uint32_t)(v & 0x1fffffffu);
        }
    }
};

}  // namespace hbmpc

namespace hbmpc {

// Goldilocks p = 2^64 - 2^32 + 1 on the host (tables only, like HFr).  Plain canonical values: the device
// representation has no Montgomery form.
struct HGl {
    uint64_t v;
    typedef unsigned __int128 u128;
    static constexpr uint64_t P = 0xFFFFFFFF00000001ULL;
    static constexpr size_t EBYTES = 8;
    static HGl raw(uint64_t x) {
        HGl r;
        r.v = x;
        return r;
    }
    static HGl zero() { return raw(0); }
    static HGl one() { return raw(1); }
    static HGl from_u64(uint64_t x) { return raw(x % P); }
    bool is_zero() const { return v == 0; }
    bool operator==(const HGl& o) const { return v == o.v; }
    HGl operator+(const HGl& b) const { return raw((uint64_t)(((u128)v + b.v) % P)); }
    HGl operator-(const HGl& b) const { return raw((uint64_t)(((u128)v + P - b.v) % P)); }
    HGl neg() const { return zero() - *this; }
    HGl operator*(const HGl& b) const { return raw((uint64_t)(((u128)v * b.v) % P)); }
    HGl pow_u64(uint64_t e) const {
        HGl acc = one(), base = *this;
        for (; e; e >>= 1) {
            if (e & 1) acc = acc * base;
            base = base * base;
        }
        return acc;
    }
    HGl inv() const { return pow_u64(P - 2); }
    void to_canon(uint64_t c[4]) const { c[0] = v, c[1] = c[2] = c[3] = 0; }
    // 7^((p-1)/2^32): GENERATOR = 7, two-adicity 32 (common/math/goldilocks.rs:4-13)
    static HGl two_adic_root() { return from_u64(7).pow_u64((P - 1) >> 32); }
    static void inv_exponent(uint64_t e[4]) { e[0] = P - 2, e[1] = e[2] = e[3] = 0; }
};

}  // namespace hbmpc
