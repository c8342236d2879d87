/*
 * hbmpc_oracle.c -- CPU restatement (plain C) of the HoneyBadgerMPC Shamir hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; the product (mpc-protocols_amd/) never links, loads or calls it.
 *
 * PARITY UNPINNED at the stored-bytes level: the reference (Rust on ark-ff/ark-poly/ark-bls12-381
 * 0.5.x, semver-caret, not vendored; Cargo.toml:17-18,32) cannot be built or run here and its tests
 * hold no golden vectors for this path.  This file is pinned by (i) the field constants of
 * SURVEY.md Appendix A, (ii) the reference's literal-input tests restated in tests/, (iii) an
 * independent Python big-int restatement (oracle/spec.py) and the golden fixtures generated from
 * it (tests/golden/), (iv) exactness of modular integer arithmetic.
 *
 * It keeps the reference's ALGORITHMIC STRUCTURE (so that it is an honest CPU baseline):
 * radix-2 FFT per secret for compute_shares, per-chunk n x (d+1) mat-vec for apply_vandermonde,
 * shared Lagrange basis + per-chunk verify/recover for batch_recover_secret, FNT -> OEC -> Gao for
 * recover_secret.  Arithmetic: 4 x 64-bit-limb Montgomery (CIOS) over bls12-381 Fr, like ark-ff.
 *
 * Compiled a second time with -DORACLE_GOLDILOCKS the same algorithms run over GoldilocksField (common/math/goldilocks.rs:4-13;
 * exported as oracle_gl_*, elements are single u64 words): the reference's code is generic over the field, and so is
 * everything below the field layer of this file.
 *
 * Reference paths are relative to /root/reference/mpc/src/.
 */
#include "hbmpc_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
#ifndef ORACLE_GOLDILOCKS
typedef struct {
    uint64_t l[4];
} fr; /* Montgomery form */
typedef U256 ELEM;
#define ORACLE_FN(name) oracle_##name

/* SURVEY.md Appendix A */
static const uint64_t MOD[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL,
                                0x73eda753299d7d48ULL};
static const uint64_t INV64 = 0xfffffffeffffffffULL; /* -r^-1 mod 2^64 */
static const fr R2 = {{0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL}};
static const fr ONE_M = {{0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL}};
static const fr ZERO = {{0, 0, 0, 0}};

static inline int geq_mod(const uint64_t a[4]) {
    for (int i = 3; i >= 0; --i) {
        if (a[i] > MOD[i]) return 1;
        if (a[i] < MOD[i]) return 0;
    }
    return 1;
}
static inline void sub_mod(uint64_t a[4]) {
    u128 br = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a[i] - MOD[i] - br;
        a[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
}
static inline fr fr_add(fr a, fr b) {
    fr o;
    u128 c = 0;
    for (int i = 0; i < 4; ++i) {
        c += (u128)a.l[i] + b.l[i];
        o.l[i] = (uint64_t)c;
        c >>= 64;
    }
    if (c || geq_mod(o.l)) sub_mod(o.l); /* r < 2^255: no carry out, kept for clarity */
    return o;
}
static inline fr fr_sub(fr a, fr b) {
    fr o;
    u128 br = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a.l[i] - b.l[i] - br;
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
static inline fr fr_neg(fr a) { return fr_sub(ZERO, a); }
static inline int fr_is_zero(fr a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
static inline int fr_eq(fr a, fr b) {
    return a.l[0] == b.l[0] && a.l[1] == b.l[1] && a.l[2] == b.l[2] && a.l[3] == b.l[3];
}
/* CIOS Montgomery product, fully unrolled, with the "no-carry" shortcut ark-ff also uses for moduli
 * whose top bit is free (r < 2^255): the running value stays below 2^256 + small, one spare word. */
#define MAC(lo, hi, a, b, c, d)                         \
    do {                                                \
        u128 p__ = (u128)(a) * (b) + (c) + (d);         \
        (lo) = (uint64_t)p__;                           \
        (hi) = (uint64_t)(p__ >> 64);                   \
    } while (0)
static inline fr fr_mul(fr a, fr b) {
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, c, c2, m, dump;
#define ROUND(bi)                                   \
    MAC(t0, c, a.l[0], bi, t0, 0);                  \
    m = t0 * INV64;                                 \
    MAC(dump, c2, m, MOD[0], t0, 0);                \
    MAC(t1, c, a.l[1], bi, t1, c);                  \
    MAC(t0, c2, m, MOD[1], t1, c2);                 \
    MAC(t2, c, a.l[2], bi, t2, c);                  \
    MAC(t1, c2, m, MOD[2], t2, c2);                 \
    MAC(t3, c, a.l[3], bi, t3, c);                  \
    MAC(t2, c2, m, MOD[3], t3, c2);                 \
    t3 = c + c2;
    ROUND(b.l[0])
    ROUND(b.l[1])
    ROUND(b.l[2])
    ROUND(b.l[3])
#undef ROUND
    (void)dump;
    fr o = {{t0, t1, t2, t3}};
    if (geq_mod(o.l)) sub_mod(o.l);
    return o;
}
static inline fr fr_from_canon(const U256* u) {
    fr a = {{u->data[0], u->data[1], u->data[2], u->data[3]}};
    return fr_mul(a, R2);
}
static inline void fr_to_canon(fr a, U256* u) {
    fr one = {{1, 0, 0, 0}};
    fr o = fr_mul(a, one);
    memcpy(u->data, o.l, 32);
}
static fr fr_from_u64(uint64_t v) {
    U256 u = {{v, 0, 0, 0}};
    return fr_from_canon(&u);
}
static fr fr_pow_limbs(fr a, const uint64_t e[4]) {
    fr acc = ONE_M;
    for (int i = 255; i >= 0; --i) {
        acc = fr_mul(acc, acc);
        if ((e[i >> 6] >> (i & 63)) & 1) acc = fr_mul(acc, a);
    }
    return acc;
}
static fr fr_pow_u64(fr a, uint64_t e) {
    uint64_t ee[4] = {e, 0, 0, 0};
    return fr_pow_limbs(a, ee);
}
static fr fr_inv(fr a) { /* a^(r-2) */
    uint64_t e[4] = {MOD[0] - 2, MOD[1], MOD[2], MOD[3]};
    return fr_pow_limbs(a, e);
}

/* ---- evaluation domain: common/mod.rs:51-68 -> GeneralEvaluationDomain::new(n) = Radix-2 of size
 * next_pow2(n); element(j) = omega^j, omega = 7^((r-1)/size) ---------------------------------- */
static size_t dom_size(size_t n) {
    size_t s = 1;
    while (s < n) s <<= 1;
    return s;
}
static int dom_omega(size_t n, fr* w) {
    size_t size = dom_size(n);
    int lg = 0;
    while (((size_t)1 << lg) < size) ++lg;
    if (lg > 32) return 0;
    /* (r-1)/2^32 */
    uint64_t e[4];
    uint64_t rm1[4] = {MOD[0] - 1, MOD[1], MOD[2], MOD[3]};
    e[0] = (rm1[0] >> 32) | (rm1[1] << 32);
    e[1] = (rm1[1] >> 32) | (rm1[2] << 32);
    e[2] = (rm1[2] >> 32) | (rm1[3] << 32);
    e[3] = rm1[3] >> 32;
    fr root = fr_pow_limbs(fr_from_u64(7), e);
    for (int i = lg; i < 32; ++i) root = fr_mul(root, root);
    *w = root;
    return 1;
}
#else
/* ---- the small field: GoldilocksField = Fp64<MontBackend<.., 1>>, p = 2^64 - 2^32 + 1, generator 7, two-adicity 32
 * (common/math/goldilocks.rs:4-13).  Values are kept canonical ("Montgomery form" with R = 1), an element is one u64,
 * products go through unsigned __int128: a second, independent restatement beside oracle/spec_gl.py. ------------- */
typedef struct {
    uint64_t l[1];
} fr;
typedef uint64_t ELEM;
#define ORACLE_FN(name) oracle_gl_##name
static const uint64_t GL_P = 0xffffffff00000001ULL;
static const fr ZERO = {{0}};
static const fr ONE_M = {{1}};
static inline fr fr_add(fr a, fr b) {
    u128 c = (u128)a.l[0] + b.l[0];
    if (c >= GL_P) c -= GL_P;
    fr o = {{(uint64_t)c}};
    return o;
}
static inline fr fr_sub(fr a, fr b) {
    fr o = {{a.l[0] >= b.l[0] ? a.l[0] - b.l[0] : a.l[0] + (GL_P - b.l[0])}};
    return o;
}
static inline fr fr_neg(fr a) { return fr_sub(ZERO, a); }
static inline int fr_is_zero(fr a) { return a.l[0] == 0; }
static inline int fr_eq(fr a, fr b) { return a.l[0] == b.l[0]; }
static inline fr fr_mul(fr a, fr b) {
    fr o = {{(uint64_t)(((u128)a.l[0] * b.l[0]) % GL_P)}};
    return o;
}
static inline fr fr_from_canon(const ELEM* u) {
    fr a = {{*u}};
    return a;
}
static inline void fr_to_canon(fr a, ELEM* u) { *u = a.l[0]; }
static fr fr_from_u64(uint64_t v) {
    fr a = {{v % GL_P}};
    return a;
}
static fr fr_pow_u64(fr a, uint64_t e) {
    fr acc = ONE_M;
    for (int i = 63; i >= 0; --i) {
        acc = fr_mul(acc, acc);
        if ((e >> i) & 1) acc = fr_mul(acc, a);
    }
    return acc;
}
static fr fr_inv(fr a) { return fr_pow_u64(a, GL_P - 2); }
static size_t dom_size(size_t n) {
    size_t s = 1;
    while (s < n) s <<= 1;
    return s;
}
static int dom_omega(size_t n, fr* w) {
    size_t size = dom_size(n);
    int lg = 0;
    while (((size_t)1 << lg) < size) ++lg;
    if (lg > 32) return 0;
    fr root = fr_pow_u64(fr_from_u64(7), (GL_P - 1) >> 32); /* GENERATOR^((p-1)/2^32) */
    for (int i = lg; i < 32; ++i) root = fr_mul(root, root);
    *w = root;
    return 1;
}
#endif
/* elems[j] = omega^j, j < count */
static int dom_elements(size_t n, size_t count, fr* elems) {
    fr w;
    if (!dom_omega(n, &w)) return 0;
    fr p = ONE_M;
    for (size_t j = 0; j < count; ++j) {
        elems[j] = p;
        p = fr_mul(p, w);
    }
    return 1;
}

/* ---- DensePolynomial (ark-poly) semantics: trailing zeros trimmed, zero poly has len 0 -------- */
#define MAXP 520
typedef struct {
    int len;
    fr c[MAXP];
} poly;

static void p_trim(poly* p) {
    while (p->len > 0 && fr_is_zero(p->c[p->len - 1])) p->len--;
}
static int p_degree(const poly* p) { return p->len == 0 ? 0 : p->len - 1; } /* degree(0) == 0 */
static fr p_eval(const poly* p, fr x) {
    fr acc = ZERO;
    for (int i = p->len - 1; i >= 0; --i) acc = fr_add(fr_mul(acc, x), p->c[i]);
    return acc;
}
static void p_set_const(poly* p, fr v) {
    p->len = 1;
    p->c[0] = v;
    p_trim(p);
}
static void p_mul(poly* out, const poly* a, const poly* b) {
    poly r;
    if (a->len == 0 || b->len == 0) {
        out->len = 0;
        return;
    }
    r.len = a->len + b->len - 1;
    for (int i = 0; i < r.len; ++i) r.c[i] = ZERO;
    for (int i = 0; i < a->len; ++i)
        for (int j = 0; j < b->len; ++j) r.c[i + j] = fr_add(r.c[i + j], fr_mul(a->c[i], b->c[j]));
    p_trim(&r);
    *out = r;
}
/* out = a * (x - root) */
static void p_mul_linear(poly* a, fr root) {
    if (a->len == 0) return;
    fr nr = fr_neg(root);
    a->c[a->len] = ZERO;
    for (int i = a->len; i > 0; --i) a->c[i] = fr_add(fr_mul(a->c[i], nr), a->c[i - 1]);
    a->c[0] = fr_mul(a->c[0], nr);
    a->len++;
    p_trim(a);
}
static void p_scale(poly* out, const poly* a, fr s) {
    if (a->len == 0 || fr_is_zero(s)) {
        out->len = 0;
        return;
    }
    out->len = a->len;
    for (int i = 0; i < a->len; ++i) out->c[i] = fr_mul(a->c[i], s);
    p_trim(out);
}
static void p_addsub(poly* out, const poly* a, const poly* b, int sub) {
    poly r;
    r.len = a->len > b->len ? a->len : b->len;
    for (int i = 0; i < r.len; ++i) {
        fr x = i < a->len ? a->c[i] : ZERO, y = i < b->len ? b->c[i] : ZERO;
        r.c[i] = sub ? fr_sub(x, y) : fr_add(x, y);
    }
    p_trim(&r);
    *out = r;
}
/* DenseOrSparsePolynomial::divide_with_q_and_r (robust_interpolate.rs:183-193).
 * Returns 0 when the divisor is the zero polynomial (ark-poly panics there). */
static int p_divmod(poly* q, poly* rem, const poly* a, const poly* b) {
    if (a->len == 0) {
        q->len = 0;
        rem->len = 0;
        return 1;
    }
    if (b->len == 0) return 0;
    if (p_degree(a) < p_degree(b)) {
        q->len = 0;
        *rem = *a;
        return 1;
    }
    poly r = *a, qq;
    qq.len = a->len - b->len + 1;
    for (int i = 0; i < qq.len; ++i) qq.c[i] = ZERO;
    fr li = fr_inv(b->c[b->len - 1]);
    while (r.len > 0 && r.len >= b->len) {
        fr cq = fr_mul(r.c[r.len - 1], li);
        int d = r.len - b->len;
        qq.c[d] = cq;
        for (int i = 0; i < b->len; ++i) r.c[d + i] = fr_sub(r.c[d + i], fr_mul(cq, b->c[i]));
        p_trim(&r);
    }
    p_trim(&qq);
    *q = qq;
    *rem = r;
    return 1;
}
/* robust_interpolate.rs:164-178 */
static void p_derivative(poly* out, const poly* p) {
    if (p->len <= 1) {
        out->len = 0;
        return;
    }
    out->len = p->len - 1;
    for (int i = 1; i < p->len; ++i) out->c[i - 1] = fr_mul(fr_from_u64((uint64_t)i), p->c[i]);
    p_trim(out);
}

/* ==== a3 compute_shares: robust_interpolate.rs:52-82, shamir.rs:158-196 =======================
 * poly = [secret, rand...]; evals = domain.fft(poly) (radix-2, size = next_pow2(n), zero padded,
 * natural order); shares = first n evals. */
static void fft_inplace(fr* a, size_t size, const fr* tw /* omega^k, k < size/2 */) {
    /* bit-reversal, then iterative Cooley-Tukey DIT: same operation count as ark-poly's radix-2 */
    size_t lg = 0;
    while (((size_t)1 << lg) < size) ++lg;
    for (size_t i = 0; i < size; ++i) {
        size_t r = 0;
        for (size_t b = 0; b < lg; ++b)
            if (i & ((size_t)1 << b)) r |= (size_t)1 << (lg - 1 - b);
        if (r > i) {
            fr tmp = a[i];
            a[i] = a[r];
            a[r] = tmp;
        }
    }
    for (size_t len = 2; len <= size; len <<= 1) {
        size_t half = len >> 1, step = size / len;
        for (size_t s = 0; s < size; s += len)
            for (size_t k = 0; k < half; ++k) {
                fr u = a[s + k], v = fr_mul(a[s + k + half], tw[k * step]);
                a[s + k] = fr_add(u, v);
                a[s + k + half] = fr_sub(u, v);
            }
    }
}

int ORACLE_FN(compute_shares)(const ELEM* coeffs, size_t B, size_t n, size_t d, ELEM* shares_out) {
    if (n <= d) return InvalidInput; /* :59-64 */
    size_t size = dom_size(n);
    fr* tw = (fr*)malloc(sizeof(fr) * (size > 1 ? size : 2));
    fr* buf = (fr*)malloc(sizeof(fr) * size);
    if (!dom_elements(n, size / 2 ? size / 2 : 1, tw)) {
        free(tw);
        free(buf);
        return NoSuitableDomain;
    }
    for (size_t b = 0; b < B; ++b) {
        for (size_t k = 0; k < size; ++k) buf[k] = k <= d ? fr_from_canon(&coeffs[b * (d + 1) + k]) : ZERO;
        fft_inplace(buf, size, tw);
        for (size_t j = 0; j < n; ++j) fr_to_canon(buf[j], &shares_out[j * B + b]);
    }
    free(tw);
    free(buf);
    return ShareSuccess;
}

/* ==== a4 make_vandermonde: common/share/mod.rs:31-45 ======================================== */
static int vandermonde_m(size_t n, size_t d, fr* v /* [n][d+1] */) {
    fr* el = (fr*)malloc(sizeof(fr) * n);
    if (!dom_elements(n, n, el)) {
        free(el);
        return NoSuitableDomain;
    }
    for (size_t j = 0; j < n; ++j) {
        fr p = ONE_M;
        for (size_t k = 0; k <= d; ++k) {
            v[j * (d + 1) + k] = p;
            p = fr_mul(p, el[j]);
        }
    }
    free(el);
    return ShareSuccess;
}
int ORACLE_FN(make_vandermonde)(size_t n, size_t d, ELEM* v_out) {
    fr* v = (fr*)malloc(sizeof(fr) * n * (d + 1));
    int rc = vandermonde_m(n, d, v);
    if (rc == ShareSuccess)
        for (size_t i = 0; i < n * (d + 1); ++i) fr_to_canon(v[i], &v_out[i]);
    free(v);
    return rc;
}
/* ==== a5 apply_vandermonde per chunk: common/share/mod.rs:50-76, loop of batch_recon.rs:160-165.
 * The matrix is built once per call (as init_batch_reconstruct_many does), then every chunk does
 * its n x (d+1) multiply-adds including the *1 of column 0 (:68). */
int ORACLE_FN(vandermonde_apply)(const ELEM* x, size_t G, size_t n, size_t d, ELEM* y_out) {
    size_t m = d + 1;
    fr* v = (fr*)malloc(sizeof(fr) * n * m);
    fr* xs = (fr*)malloc(sizeof(fr) * m);
    int rc = vandermonde_m(n, d, v);
    if (rc != ShareSuccess) {
        free(v);
        free(xs);
        return rc;
    }
    for (size_t g = 0; g < G; ++g) {
        for (size_t k = 0; k < m; ++k) xs[k] = fr_from_canon(&x[g * m + k]);
        for (size_t j = 0; j < n; ++j) {
            fr acc = fr_mul(xs[0], v[j * m]);
            for (size_t k = 1; k < m; ++k) acc = fr_add(acc, fr_mul(xs[k], v[j * m + k]));
            fr_to_canon(acc, &y_out[j * G + g]);
        }
    }
    free(v);
    free(xs);
    return ShareSuccess;
}

/* ==== textbook Lagrange: common/mod.rs:134-165 =============================================== */
static int lagrange_interpolate(poly* out, const fr* xs, const fr* ys, int cnt) {
    for (int i = 0; i < cnt; ++i)
        for (int j = i + 1; j < cnt; ++j)
            if (fr_eq(xs[i], xs[j])) return InvalidInput;
    poly res, num, term;
    res.len = 0;
    for (int j = 0; j < cnt; ++j) {
        p_set_const(&num, ONE_M);
        fr den = ONE_M;
        for (int m = 0; m < cnt; ++m)
            if (m != j) {
                p_mul_linear(&num, xs[m]);
                den = fr_mul(den, fr_sub(xs[j], xs[m]));
            }
        p_scale(&term, &num, fr_mul(ys[j], fr_inv(den)));
        p_addsub(&res, &res, &term, 0);
    }
    *out = res;
    return ShareSuccess;
}

/* ==== a6 robust_interpolate_fnt: robust_interpolate.rs:206-266 =============================== */
typedef struct {
    size_t id;
    fr v;
} sh_t;

static int cmp_sh(const void* a, const void* b) {
    size_t x = ((const sh_t*)a)->id, y = ((const sh_t*)b)->id;
    return x < y ? -1 : x > y;
}

static int robust_interpolate_fnt(poly* out, size_t t, size_t n, const sh_t* shares, size_t cnt, size_t degree,
                                  const fr* el) {
    (void)n;
    size_t m = degree + 1;
    poly a_poly, a_der, basis, rem, lin, term, interp;
    p_set_const(&a_poly, ONE_M);
    for (size_t i = 0; i < m; ++i) p_mul_linear(&a_poly, el[shares[i].id]);
    p_derivative(&a_der, &a_poly);
    interp.len = 0;
    for (size_t i = 0; i < m; ++i) {
        fr x_i = el[shares[i].id];
        fr denom = p_eval(&a_der, x_i);
        if (fr_is_zero(denom)) return PolynomialOperationError;
        fr scalar = fr_mul(shares[i].v, fr_inv(denom));
        lin.len = 2;
        lin.c[0] = fr_neg(x_i);
        lin.c[1] = ONE_M;
        if (!p_divmod(&basis, &rem, &a_poly, &lin)) return PolynomialOperationError;
        if (rem.len != 0) return PolynomialOperationError;
        p_scale(&term, &basis, scalar);
        p_addsub(&interp, &interp, &term, 0);
    }
    size_t valid = 0;
    for (size_t s = 0; s < cnt; ++s)
        if (fr_eq(p_eval(&interp, el[shares[s].id]), shares[s].v)) ++valid;
    if (valid >= degree + t + 1) {
        *out = interp;
        return ShareSuccess;
    }
    return DecodingError;
}

/* ==== a8 gao_rs_decode: robust_interpolate.rs:456-538 (+ compute_g0_from_domain :540-565) ===== */
static int gao_rs_decode(poly* out, const fr* received, size_t k, size_t n, const uint8_t* erased /* [n] */,
                         const fr* el) {
    if (k > n) return InvalidInput;
    size_t s = 0;
    poly s_poly, g0, g1, xa, rem;
    p_set_const(&s_poly, ONE_M);
    for (size_t i = 0; i < n; ++i)
        if (erased[i]) {
            ++s;
            p_mul_linear(&s_poly, el[i]);
        }
    fr xs[MAXP / 2], ys[MAXP / 2];
    int cnt = 0;
    for (size_t i = 0; i < n; ++i)
        if (!erased[i]) {
            xs[cnt] = el[i];
            ys[cnt] = received[i];
            ++cnt;
        }
    int rc = lagrange_interpolate(&g1, xs, ys, cnt);
    if (rc != ShareSuccess) return rc;
    p_set_const(&xa, ONE_M);
    for (size_t i = 0; i < n; ++i) p_mul_linear(&xa, el[i]);
    if (!p_divmod(&g0, &rem, &xa, &s_poly)) return PolynomialOperationError;
    size_t threshold = (n - s + k) / 2;
    poly r0 = g0, r1 = g1, t0, t1, q, tmp, r, tt;
    t0.len = 0;
    p_set_const(&t1, ONE_M);
    while ((size_t)p_degree(&r1) >= threshold) {
        if (!p_divmod(&q, &rem, &r0, &r1)) return PolynomialOperationError; /* ark-poly would panic */
        p_mul(&tmp, &q, &r1);
        p_addsub(&r, &r0, &tmp, 1);
        p_mul(&tmp, &q, &t1);
        p_addsub(&tt, &t0, &tmp, 1);
        r0 = r1;
        r1 = r;
        t0 = t1;
        t1 = tt;
    }
    poly quotient, remainder;
    if (!p_divmod(&quotient, &rem, &r1, &t1)) return PolynomialOperationError;
    p_mul(&tmp, &quotient, &t1);
    p_addsub(&remainder, &r1, &tmp, 1);
    if (remainder.len == 0 && (size_t)p_degree(&quotient) < k) {
        *out = quotient;
        return ShareSuccess;
    }
    return DecodingError;
}

/* ==== a8 oec_decode: robust_interpolate.rs:579-628 =========================================== */
static int oec_decode(poly* out, fr* at0, size_t n, size_t t, const sh_t* shares, size_t cnt, size_t degree,
                      const fr* el) {
    fr* received = (fr*)malloc(sizeof(fr) * n);
    uint8_t* erased = (uint8_t*)malloc(n);
    int result = DecodingError;
    for (size_t r = 1; r <= t; ++r) {
        size_t required = degree + t + 1 + r;
        if (cnt < required) break;
        for (size_t i = 0; i < n; ++i) {
            received[i] = ZERO;
            erased[i] = 1;
        }
        for (size_t i = 0; i < required; ++i) {
            received[shares[i].id] = shares[i].v;
            erased[shares[i].id] = 0;
        }
        poly p;
        if (gao_rs_decode(&p, received, degree + 1, n, erased, el) != ShareSuccess) continue;
        size_t matched = 0;
        for (size_t i = 0; i < required; ++i)
            if (fr_eq(p_eval(&p, el[shares[i].id]), shares[i].v)) ++matched;
        if (matched >= degree + t + 1) {
            *out = p;
            *at0 = p_eval(&p, ZERO);
            result = ShareSuccess;
            break;
        }
    }
    free(received);
    free(erased);
    return result;
}

/* ==== a6 recover_secret: robust_interpolate.rs:94-157 ========================================
 * shares already validated for duplicates / range by the callers below; sorted here. */
static int recover_core(poly* out, fr* at0, sh_t* shares, size_t cnt, size_t n, size_t t, size_t degree,
                        const fr* el) {
    qsort(shares, cnt, sizeof(sh_t), cmp_sh);
    if (robust_interpolate_fnt(out, t, n, shares, degree + t + 1, degree, el) == ShareSuccess) {
        *at0 = p_eval(out, ZERO);
        return ShareSuccess;
    }
    return oec_decode(out, at0, n, t, shares, cnt, degree, el);
}

int ORACLE_FN(recover_secret)(const size_t* ids, const size_t* degrees, const ELEM* vals, size_t S, size_t n, size_t t,
                          ELEM* coeffs_out, size_t* ncoeffs_out, ELEM* secret_out) {
    if (n < 3 * t + 1) return InvalidInput; /* :100 */
    if (S == 0) return InvalidInput;        /* :108 */
    size_t degree = degrees[0];
    for (size_t i = 0; i < S; ++i)
        if (degrees[i] != degree) return DegreeMismatch; /* :114 */
    for (size_t i = 0; i < S; ++i)
        for (size_t j = i + 1; j < S; ++j)
            if (ids[i] == ids[j]) return InvalidInput; /* :118 */
    for (size_t i = 0; i < S; ++i)
        if (ids[i] >= n) return InvalidInput; /* :125 */
    if (S < degree + t + 1) return InvalidInput; /* :135 */
    fr* el = (fr*)malloc(sizeof(fr) * n);
    if (!dom_elements(n, n, el)) {
        free(el);
        return NoSuitableDomain;
    }
    sh_t* sh = (sh_t*)malloc(sizeof(sh_t) * S);
    for (size_t i = 0; i < S; ++i) {
        sh[i].id = ids[i];
        sh[i].v = fr_from_canon(&vals[i]);
    }
    poly p;
    fr at0;
    int rc = recover_core(&p, &at0, sh, S, n, t, degree, el);
    if (rc == ShareSuccess) {
        for (int i = 0; i < p.len; ++i) fr_to_canon(p.c[i], &coeffs_out[i]);
        *ncoeffs_out = (size_t)p.len;
        fr_to_canon(at0, secret_out);
    }
    free(sh);
    free(el);
    return rc;
}

int ORACLE_FN(gao_rs_decode)(const ELEM* received, size_t k, size_t n, const size_t* erasure_positions,
                         size_t n_erasures, ELEM* coeffs_out, size_t* ncoeffs_out) {
    if (k > n) return InvalidInput;
    fr* el = (fr*)malloc(sizeof(fr) * n);
    if (!dom_elements(n, n, el)) {
        free(el);
        return NoSuitableDomain;
    }
    fr* rec = (fr*)malloc(sizeof(fr) * n);
    uint8_t* er = (uint8_t*)calloc(n, 1);
    for (size_t i = 0; i < n; ++i) rec[i] = fr_from_canon(&received[i]);
    for (size_t i = 0; i < n_erasures; ++i)
        if (erasure_positions[i] < n) er[erasure_positions[i]] = 1;
    poly p;
    int rc = gao_rs_decode(&p, rec, k, n, er, el);
    if (rc == ShareSuccess) {
        for (int i = 0; i < p.len; ++i) fr_to_canon(p.c[i], &coeffs_out[i]);
        *ncoeffs_out = (size_t)p.len;
    }
    free(el);
    free(rec);
    free(er);
    return rc;
}

/* ==== a7 batch_recover_secret: robust_interpolate.rs:284-443 ================================= */
typedef struct {
    size_t id;
    size_t pos; /* row in evals */
} snd_t;
static int cmp_snd(const void* a, const void* b) {
    size_t x = ((const snd_t*)a)->id, y = ((const snd_t*)b)->id;
    return x < y ? -1 : x > y;
}

static int batch_recover_impl(const size_t* sender_ids, size_t S, const ELEM* evals, size_t G, size_t n, size_t d,
                              size_t t, ELEM* coeffs_out, uint32_t* ncoeffs_out, uint8_t* status_out, int p0_only) {
    if (n < 3 * t + 1) return InvalidInput; /* :290 */
    if (S == 0) return InvalidInput;        /* :297 */
    if (G == 0) return InvalidInput;        /* :303 */
    snd_t* srt = (snd_t*)malloc(sizeof(snd_t) * S);
    for (size_t i = 0; i < S; ++i) {
        srt[i].id = sender_ids[i];
        srt[i].pos = i;
    }
    qsort(srt, S, sizeof(snd_t), cmp_snd); /* :313-315 */
    for (size_t i = 0; i < S; ++i) {       /* :317-330: duplicate first, then range, in sorted order */
        if (i > 0 && srt[i].id == srt[i - 1].id) {
            free(srt);
            return InvalidInput;
        }
        if (srt[i].id >= n) {
            free(srt);
            return InvalidInput;
        }
    }
    size_t needed = d + t + 1, m = d + 1;
    if (S < needed) {
        free(srt);
        return InvalidInput; /* :333 */
    }
    fr* el = (fr*)malloc(sizeof(fr) * n);
    if (!dom_elements(n, n, el)) {
        free(el);
        free(srt);
        return NoSuitableDomain;
    }
    /* shared Lagrange basis :351-376 */
    poly a_poly, a_der, bp, rem, lin;
    poly* basis = (poly*)malloc(sizeof(poly) * m);
    p_set_const(&a_poly, ONE_M);
    for (size_t i = 0; i < m; ++i) p_mul_linear(&a_poly, el[srt[i].id]);
    p_derivative(&a_der, &a_poly);
    int rc = ShareSuccess;
    for (size_t i = 0; i < m && rc == ShareSuccess; ++i) {
        fr x_i = el[srt[i].id];
        fr denom = p_eval(&a_der, x_i);
        if (fr_is_zero(denom)) {
            rc = PolynomialOperationError;
            break;
        }
        lin.len = 2;
        lin.c[0] = fr_neg(x_i);
        lin.c[1] = ONE_M;
        if (!p_divmod(&bp, &rem, &a_poly, &lin) || rem.len != 0) {
            rc = PolynomialOperationError;
            break;
        }
        p_scale(&basis[i], &bp, fr_inv(denom));
    }
    if (rc != ShareSuccess) {
        free(basis);
        free(el);
        free(srt);
        return rc;
    }
    /* verify matrix :392-399 */
    fr* vm = (fr*)malloc(sizeof(fr) * needed * m);
    for (size_t s = 0; s < needed; ++s)
        for (size_t i = 0; i < m; ++i) vm[s * m + i] = p_eval(&basis[i], el[srt[s].id]);
    fr* bc = (fr*)malloc(sizeof(fr) * m * m); /* bc[k][i] = basis[i].coeffs[k] or 0 (:423) */
    for (size_t k = 0; k < m; ++k)
        for (size_t i = 0; i < m; ++i) bc[k * m + i] = (int)k < basis[i].len ? basis[i].c[k] : ZERO;

    fr* y = (fr*)malloc(sizeof(fr) * needed);
    sh_t* sh = (sh_t*)malloc(sizeof(sh_t) * S);
    size_t ow = p0_only ? 1 : m;
    int first_err = ShareSuccess;
    for (size_t c = 0; c < G; ++c) {
        for (size_t s = 0; s < needed; ++s) y[s] = fr_from_canon(&evals[srt[s].pos * G + c]);
        int ok = 1;
        for (size_t s = 0; s < needed && ok; ++s) { /* :405-415 */
            fr acc = ZERO;
            for (size_t i = 0; i < m; ++i) acc = fr_add(acc, fr_mul(vm[s * m + i], y[i]));
            if (!fr_eq(acc, y[s])) ok = 0;
        }
        if (ok) { /* :419-428 */
            for (size_t k = 0; k < ow; ++k) {
                fr acc = ZERO;
                for (size_t i = 0; i < m; ++i) acc = fr_add(acc, fr_mul(bc[k * m + i], y[i]));
                fr_to_canon(acc, &coeffs_out[c * ow + k]);
            }
            if (ncoeffs_out) ncoeffs_out[c] = (uint32_t)m;
            if (status_out) status_out[c] = 0;
        } else { /* :433-438 */
            for (size_t i = 0; i < S; ++i) {
                sh[i].id = srt[i].id;
                sh[i].v = fr_from_canon(&evals[srt[i].pos * G + c]);
            }
            poly p;
            fr at0;
            int r2 = recover_core(&p, &at0, sh, S, n, t, d, el);
            ELEM zero;
            memset(&zero, 0, sizeof zero);
            for (size_t k = 0; k < ow; ++k) coeffs_out[c * ow + k] = zero;
            if (r2 == ShareSuccess) {
                for (int k = 0; k < p.len && (size_t)k < ow; ++k) fr_to_canon(p.c[k], &coeffs_out[c * ow + k]);
                if (ncoeffs_out) ncoeffs_out[c] = (uint32_t)p.len;
                if (status_out) status_out[c] = 1;
            } else {
                if (ncoeffs_out) ncoeffs_out[c] = 0;
                if (status_out) status_out[c] = (uint8_t)r2;
                if (first_err == ShareSuccess) first_err = r2; /* the reference's `?` stops here (:437) */
            }
        }
    }
    free(sh);
    free(y);
    free(bc);
    free(vm);
    free(basis);
    free(el);
    free(srt);
    return first_err;
}

int ORACLE_FN(batch_recover)(const size_t* sender_ids, size_t S, const ELEM* evals, size_t G, size_t n, size_t d,
                         size_t t, ELEM* coeffs_out, uint32_t* ncoeffs_out, uint8_t* status_out) {
    return batch_recover_impl(sender_ids, S, evals, G, n, d, t, coeffs_out, ncoeffs_out, status_out, 0);
}
int ORACLE_FN(batch_recover_p0)(const size_t* sender_ids, size_t S, const ELEM* evals, size_t G, size_t n, size_t d,
                            size_t t, ELEM* secrets_out, uint8_t* status_out) {
    return batch_recover_impl(sender_ids, S, evals, G, n, d, t, secrets_out, NULL, status_out, 1);
}

/* ==== NonRobustShare::recover_secret: shamir.rs:199-239 ====================================== */
int ORACLE_FN(nonrobust_recover_secret)(const size_t* ids, const size_t* degrees, const ELEM* vals, size_t S, size_t n,
                                    ELEM* coeffs_out, size_t* ncoeffs_out, ELEM* secret_out) {
    if (S == 0) return InvalidInput;
    for (size_t i = 0; i < S; ++i)
        for (size_t j = i + 1; j < S; ++j)
            if (ids[i] == ids[j]) return InvalidInput;
    size_t deg = degrees[0];
    for (size_t i = 0; i < S; ++i)
        if (degrees[i] != deg) return DegreeMismatch;
    if (S < deg + 1) return InsufficientShares;
    fr* el = (fr*)malloc(sizeof(fr) * n);
    if (!dom_elements(n, n, el)) {
        free(el);
        return NoSuitableDomain;
    }
    for (size_t i = 0; i < S; ++i)
        if (ids[i] >= n) {
            free(el);
            return InvalidInput;
        }
    fr* xs = (fr*)malloc(sizeof(fr) * S);
    fr* ys = (fr*)malloc(sizeof(fr) * S);
    for (size_t i = 0; i < S; ++i) {
        xs[i] = el[ids[i]];
        ys[i] = fr_from_canon(&vals[i]);
    }
    poly p;
    int rc = lagrange_interpolate(&p, xs, ys, (int)S);
    if (rc == ShareSuccess && (size_t)p_degree(&p) > deg) rc = DegreeMismatch;
    if (rc == ShareSuccess) {
        for (int i = 0; i < p.len; ++i) fr_to_canon(p.c[i], &coeffs_out[i]);
        *ncoeffs_out = (size_t)p.len;
        ELEM zero;
            memset(&zero, 0, sizeof zero);
        *secret_out = zero; /* reference indexes poly[0]: panics on the zero polynomial */
        if (p.len > 0) fr_to_canon(p.c[0], secret_out);
    }
    free(xs);
    free(ys);
    free(el);
    return rc;
}

/* ==== element-wise (a9, a11, a12, a13).  Canonical in, canonical out. ======================== */
#ifndef ORACLE_GOLDILOCKS
static inline fr ld(const ELEM* u) {
    fr a = {{u->data[0], u->data[1], u->data[2], u->data[3]}};
    return a;
} /* canonical value used directly where only add/sub are needed */
static inline void st(fr a, ELEM* u) { memcpy(u->data, a.l, 32); }
#else
static inline fr ld(const ELEM* u) { return fr_from_canon(u); }
static inline void st(fr a, ELEM* u) { *u = a.l[0]; }
#endif

/* triple_generation.rs:333-340 */
int ORACLE_FN(triple_local)(const ELEM* a, const ELEM* b, const ELEM* r2t, size_t N, ELEM* out) {
    for (size_t i = 0; i < N; ++i) {
        fr p = fr_mul(fr_from_canon(&a[i]), ld(&b[i])); /* (aR)*b/R = ab, canonical */
        st(fr_sub(p, ld(&r2t[i])), &out[i]);
    }
    return ShareSuccess;
}
/* triple_generation.rs:196-208 */
int ORACLE_FN(triple_finalize)(const ELEM* rt, const ELEM* opened, size_t N, ELEM* c_out) {
    for (size_t i = 0; i < N; ++i) st(fr_add(ld(&rt[i]), ld(&opened[i])), &c_out[i]);
    return ShareSuccess;
}
/* multiplication.rs:417-426 */
int ORACLE_FN(beaver_open_shares)(const ELEM* a, const ELEM* b, const ELEM* x, const ELEM* y, size_t N, ELEM* d_sh,
                              ELEM* e_sh) {
    for (size_t i = 0; i < N; ++i) {
        st(fr_sub(ld(&a[i]), ld(&x[i])), &d_sh[i]);
        st(fr_sub(ld(&b[i]), ld(&y[i])), &e_sh[i]);
    }
    return ShareSuccess;
}
/* multiplication.rs:57-100 */
int ORACLE_FN(beaver_finalize)(const ELEM* c, const ELEM* x, const ELEM* y, const ELEM* d, const ELEM* e, size_t N,
                           ELEM* z_out) {
    for (size_t i = 0; i < N; ++i) {
        fr dm = fr_from_canon(&d[i]), em = fr_from_canon(&e[i]);
        fr de = fr_mul(dm, ld(&e[i]));
        fr dy = fr_mul(dm, ld(&y[i]));
        fr ex = fr_mul(em, ld(&x[i]));
        st(fr_sub(fr_sub(fr_sub(ld(&c[i]), de), dy), ex), &z_out[i]);
    }
    return ShareSuccess;
}
#ifndef ORACLE_GOLDILOCKS /* TruncPr's byte logic is written for the 256-bit field (the reference's fpmul runs over Fr) */
/* fpmul/mod.rs:377 */
static fr pow2_m(size_t e) { return fr_pow_u64(fr_from_u64(2), (uint64_t)e); }
/* truncpr.rs:277-283 */
int ORACLE_FN(truncpr_rdash)(const ELEM* r_bits, size_t m, size_t N, ELEM* r_dash_out) {
    for (size_t i = 0; i < N; ++i) {
        fr acc = ZERO;
        for (size_t j = 0; j < m; ++j) acc = fr_add(acc, fr_mul(pow2_m(j), ld(&r_bits[j * N + i])));
        st(acc, &r_dash_out[i]);
    }
    return ShareSuccess;
}
/* truncpr.rs:275,294-297 */
int ORACLE_FN(truncpr_open_share)(const ELEM* a, const ELEM* r_dash, const ELEM* r_int, size_t k, size_t m, size_t N,
                              ELEM* open_out) {
    if (k == 0) return InvalidInput; /* 2^(k-1): usize underflow panics in the reference */
    fr one = {{1, 0, 0, 0}};
    fr p2k = fr_mul(pow2_m(k - 1), one); /* canonical */
    fr p2m = pow2_m(m);
    for (size_t i = 0; i < N; ++i) {
        fr b = fr_add(ld(&a[i]), p2k);
        fr r = fr_add(fr_mul(p2m, ld(&r_int[i])), ld(&r_dash[i]));
        st(fr_add(b, r), &open_out[i]);
    }
    return ShareSuccess;
}
/* fpmul/mod.rs:381-406 on a canonical value */
static fr mod_pow_2(fr x, size_t m) {
    uint8_t bytes[32];
    memcpy(bytes, x.l, 32);
    size_t full = m / 8, extra = m % 8, usable = extra ? full + 1 : full, len = 32;
    if (len > usable) len = usable;
    if (extra > 0 && len > 0) bytes[full] &= (uint8_t)((1u << extra) - 1);
    fr o = ZERO;
    memcpy(o.l, bytes, len);
    if (geq_mod(o.l)) sub_mod(o.l); /* from_le_bytes_mod_order; never taken for x < r */
    return o;
}
/* truncpr.rs:215-220 */
int ORACLE_FN(truncpr_finalize)(const ELEM* a, const ELEM* r_dash, const ELEM* c_open, size_t m, size_t N,
                            ELEM* d_out) {
    if (m % 8 != 0 && m / 8 >= 32) return InvalidInput; /* bytes[full_bytes] out of bounds: reference panics */
    fr inv2m = fr_inv(pow2_m(m));
    for (size_t i = 0; i < N; ++i) {
        fr c_mod = mod_pow_2(ld(&c_open[i]), m);
        fr a_prime = fr_sub(c_mod, ld(&r_dash[i]));
        st(fr_mul(inv2m, fr_sub(ld(&a[i]), a_prime)), &d_out[i]);
    }
    return ShareSuccess;
}

#endif

/* ---- raw field ops for cross-checking the device arithmetic ---------------------------------- */
void ORACLE_FN(fr_mul)(const ELEM* a, const ELEM* b, size_t N, ELEM* out) {
    for (size_t i = 0; i < N; ++i) st(fr_mul(fr_from_canon(&a[i]), ld(&b[i])), &out[i]);
}
void ORACLE_FN(fr_add)(const ELEM* a, const ELEM* b, size_t N, ELEM* out) {
    for (size_t i = 0; i < N; ++i) st(fr_add(ld(&a[i]), ld(&b[i])), &out[i]);
}
void ORACLE_FN(fr_sub)(const ELEM* a, const ELEM* b, size_t N, ELEM* out) {
    for (size_t i = 0; i < N; ++i) st(fr_sub(ld(&a[i]), ld(&b[i])), &out[i]);
}
void ORACLE_FN(fr_inv)(const ELEM* a, size_t N, ELEM* out) {
    for (size_t i = 0; i < N; ++i) fr_to_canon(fr_inv(fr_from_canon(&a[i])), &out[i]);
}
void ORACLE_FN(domain_elements)(size_t n, size_t count, ELEM* out) {
    fr* el = (fr*)malloc(sizeof(fr) * (count ? count : 1));
    if (dom_elements(n, count, el))
        for (size_t i = 0; i < count; ++i) fr_to_canon(el[i], &out[i]);
    free(el);
}

#ifndef ORACLE_GOLDILOCKS
/* SplitMix64(seed) -> 4 limbs -> mod r  (SURVEY.md section 8(d)); conditional subtractions reduce a
 * 256-bit value below r (2^256 < 5r). */
void ORACLE_FN(fill_random)(uint64_t seed, size_t N, ELEM* out) {
    uint64_t s = seed;
    for (size_t i = 0; i < N; ++i) {
        uint64_t l[4];
        for (int k = 0; k < 4; ++k) {
            s += 0x9E3779B97F4A7C15ULL;
            uint64_t z = s;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
            l[k] = z ^ (z >> 31);
        }
        while (geq_mod(l)) sub_mod(l);
        memcpy(out[i].data, l, 32);
    }
}
#else
/* SplitMix64(seed) -> one word, redrawn while >= p (uniform on the field) */
void oracle_gl_fill_random(uint64_t seed, size_t N, ELEM* out) {
    uint64_t s = seed;
    for (size_t i = 0; i < N; ++i) {
        uint64_t v;
        do {
            s += 0x9E3779B97F4A7C15ULL;
            uint64_t z = s;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
            v = z ^ (z >> 31);
        } while (v >= GL_P);
        out[i] = v;
    }
}
#endif
