/* A plain C99 caller of include/hbmpc_hip.h -- the shape of the reference's own C test of its exported ABI
 * (mpc/src/ffi/tests/secret_share.c: create shares of a literal U256 secret for n = 6 parties, recover, compare),
 * restated against this library's entry points, for both share kinds and both fields.  Needs an MI355X. */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "hbmpc_hip.h"

static uint64_t lcg_state = 0x9E3779B97F4A7C15ull;
static uint64_t lcg(void) {
    lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
    return lcg_state;
}
/* a canonical Fr element: top limb below r's top limb */
static U256 rand_fr(void) {
    U256 v = {{lcg(), lcg(), lcg(), lcg() % 0x73eda753299d7d48ull}};
    return v;
}
#define CHECK(c)                                                                  \
    do {                                                                          \
        if (!(c)) {                                                               \
            printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c);                 \
            return 1;                                                             \
        }                                                                         \
    } while (0)

/* RobustShare: secret {3,3,22,22}, n = 6, degree 2, t = 1 (secret_share.c:64-118) */
static int robust_roundtrip(hbmpc_ctx* ctx) {
    const U256 secret = {{3, 3, 22, 22}};
    U256 coeffs[3], shares[6], co[3], rec;
    size_t ids[6] = {0, 1, 2, 3, 4, 5}, deg[6] = {2, 2, 2, 2, 2, 2}, nco = 0;
    coeffs[0] = secret;
    coeffs[1] = rand_fr();
    coeffs[2] = rand_fr();
    CHECK(hbmpc_compute_shares(ctx, coeffs, 1, 6, 2, shares) == ShareSuccess);
    CHECK(hbmpc_recover_secret(ctx, ids, deg, shares, 6, 6, 1, co, &nco, &rec) == ShareSuccess);
    CHECK(nco == 3 && memcmp(&rec, &secret, sizeof rec) == 0 && memcmp(co, coeffs, sizeof co) == 0);
    shares[4].data[0] ^= 1; /* one lie: still recovered (t = 1) */
    CHECK(hbmpc_recover_secret(ctx, ids, deg, shares, 6, 6, 1, co, &nco, &rec) == ShareSuccess);
    CHECK(memcmp(&rec, &secret, sizeof rec) == 0);
    shares[1].data[0] ^= 1; /* two lies: n - (d+1) = 3 redundant points cannot correct them */
    CHECK(hbmpc_recover_secret(ctx, ids, deg, shares, 6, 6, 1, co, &nco, &rec) != ShareSuccess);
    CHECK(hbmpc_compute_shares(ctx, coeffs, 1, 2, 2, shares) == InvalidInput); /* n <= degree */
    return 0;
}
/* NonRobustShare: secret {16,33,44,81}, n = 6, degree 5 (secret_share.c:120-170) */
static int nonrobust_roundtrip(hbmpc_ctx* ctx) {
    const U256 secret = {{16, 33, 44, 81}};
    U256 coeffs[6], shares[6], co[6], rec;
    size_t ids[6] = {0, 1, 2, 3, 4, 5}, deg[6] = {5, 5, 5, 5, 5, 5}, nco = 0;
    int i;
    coeffs[0] = secret;
    for (i = 1; i < 6; ++i) coeffs[i] = rand_fr();
    CHECK(hbmpc_compute_shares(ctx, coeffs, 1, 6, 5, shares) == ShareSuccess);
    CHECK(hbmpc_nonrobust_recover_secret(ctx, ids, deg, shares, 6, 6, co, &nco, &rec) == ShareSuccess);
    CHECK(nco == 6 && memcmp(&rec, &secret, sizeof rec) == 0);
    CHECK(hbmpc_nonrobust_recover_secret(ctx, ids, deg, shares, 5, 6, co, &nco, &rec) == InsufficientShares);
    return 0;
}
/* the same round trip in the small field (common/math/goldilocks.rs) */
static int goldilocks_roundtrip(void) {
    hbmpc_ctx* ctx = NULL;
    uint64_t coeffs[3] = {520, 86, 918520}, shares[6], co[3], rec = 0;
    size_t ids[6] = {0, 1, 2, 3, 4, 5}, deg[6] = {2, 2, 2, 2, 2, 2}, nco = 0;
    U256 dummy[6];
    CHECK(hbmpc_create(0, Goldilocks64, &ctx) == ShareSuccess);
    CHECK(hbmpc_gl_compute_shares(ctx, coeffs, 1, 6, 2, shares) == ShareSuccess);
    shares[2] ^= 1;
    CHECK(hbmpc_gl_recover_secret(ctx, ids, deg, shares, 6, 6, 1, co, &nco, &rec) == ShareSuccess);
    CHECK(nco == 3 && rec == 520 && co[1] == 86 && co[2] == 918520);
    CHECK(hbmpc_compute_shares(ctx, dummy, 1, 6, 2, dummy) == TypeMismatch); /* a context serves one field */
    hbmpc_destroy(ctx);
    return 0;
}

/* a dealer without a host rng (seeded sharing, secrets drawn too), and the in-place wire path: the encode kernel
 * writes the n Vec<F> payloads, the receive side validates and decodes out of them where they lie */
static int seeded_and_wire(hbmpc_ctx* ctx) {
    enum { N = 7, T = 2, D = 2, B = 40, STRIDE = 32 * (B + 1) };
    uint8_t seed[32];
    U256 shares[N * B], again[N * B];
    size_t ids[N], stats[4];
    void *buf = NULL;
    U256 *x_dev, *co_dev;
    uint32_t* st_dev;
    uint8_t* wire;
    uint32_t st[N];
    U256 x[B * (D + 1)], co[B * (D + 1)];
    int i;
    for (i = 0; i < 32; ++i) seed[i] = (uint8_t)(7 * i + 1);
    for (i = 0; i < N; ++i) ids[i] = (size_t)i;
    CHECK(hbmpc_compute_shares_seeded(ctx, seed, NULL, B, 0, N, D, shares) == ShareSuccess);
    CHECK(hbmpc_compute_shares_seeded(ctx, seed, NULL, B, 0, N, D, again) == ShareSuccess);
    CHECK(memcmp(shares, again, sizeof shares) == 0);                       /* a function of (seed, index) only */
    CHECK(hbmpc_compute_shares_seeded(ctx, seed, NULL, B, 1, N, D, again) == ShareSuccess);
    CHECK(memcmp(shares, again, sizeof shares) != 0);
    /* device side: x[B][D+1] -> n payloads in place -> validate -> decode in place */
    for (i = 0; i < B * (D + 1); ++i) x[i] = rand_fr();
    CHECK(hbmpc_dev_alloc(ctx, sizeof x + sizeof co + 64 * sizeof(uint32_t) + (size_t)N * STRIDE + 64, &buf) == ShareSuccess);
    x_dev = (U256*)buf;
    co_dev = x_dev + B * (D + 1);
    st_dev = (uint32_t*)(co_dev + B * (D + 1));
    wire = (uint8_t*)(st_dev + 64);
    wire += (32 - ((uintptr_t)wire & 31)) % 32 + 24;                         /* 8 bytes before a 32-byte boundary */
    CHECK(hbmpc_memcpy_h2d(ctx, x_dev, x, sizeof x, NULL) == ShareSuccess);
    CHECK(hbmpc_dev_encode_fvec(ctx, x_dev, B, N, D, wire, STRIDE, NULL) == ShareSuccess);
    CHECK(hbmpc_dev_validate_fvec(ctx, wire, STRIDE, 8 + 32 * B, B, N, st_dev, NULL) == ShareSuccess);
    CHECK(hbmpc_dev_batch_recover_slots(ctx, ids, ids, N, (const U256*)(wire + 8), STRIDE / 32, B, N, D, T, 0, co_dev, NULL, NULL,
                                        NULL, NULL) == ShareSuccess);
    CHECK(hbmpc_memcpy_d2h(ctx, co, co_dev, sizeof co, NULL) == ShareSuccess);
    CHECK(hbmpc_memcpy_d2h(ctx, st, st_dev, sizeof st, NULL) == ShareSuccess);
    CHECK(hbmpc_stream_sync(ctx, NULL) == ShareSuccess);
    for (i = 0; i < N; ++i) CHECK(st[i] == 0);
    CHECK(memcmp(co, x, sizeof co) == 0);
    CHECK(hbmpc_cache_stats(ctx, stats) == ShareSuccess && stats[0] > 0 && stats[3] == 0);
    CHECK(hbmpc_dev_free(ctx, buf) == ShareSuccess);
    return 0;
}

/* The pipelines behind the C ABI (hbmpc_pipe_*): run_preprocessing's triple part for n = 4, t = 1 from the dealers' polynomials
 * (honeybadger/mod.rs:1239-1393: RanSha -> a, b; DouSha + RanDouSha -> r; TripleGen), checked with the library's own host calls:
 * every opened c must equal the opened a times the opened b, the producers' verdicts must be clean, and the same handle
 * replayed as a HIP graph must give the same bytes. */
#define PRE_N 4
#define PRE_T 1
#define PRE_TRIPLES 6 /* a multiple of 2t + 1, of n - 2t and of t + 1: the producers write straight into TripleGen's arrays */
static int preprocessing_pipeline(hbmpc_ctx* ctx) {
    enum { n = PRE_N, t = PRE_T, N = PRE_TRIPLES, Krs = 2 * N / (n - 2 * t), Krd = N / (t + 1) };
    static U256 co[n * Krs * (t + 1)], ct[n * Krd * (t + 1)], c2t[n * Krd * (2 * t + 1)], a[n * N], b[n * N], c[n * N], c2[n * N];
    hbmpc_pipe *pre = NULL, *rs = NULL, *rd = NULL, *tg = NULL;
    hbmpc_recover_summary sm;
    void* stream = NULL;
    uint32_t verdict[2] = {9, 9};
    size_t ids[n], deg[n], i, p, k, nco = 0, elements = 0;
    void* dev = NULL;
    for (i = 0; i < n; ++i) ids[i] = i, deg[i] = t;
    for (i = 0; i < sizeof co / sizeof co[0]; ++i) co[i] = rand_fr();
    for (i = 0; i < sizeof ct / sizeof ct[0]; ++i) ct[i] = rand_fr();
    for (i = 0; i < sizeof c2t / sizeof c2t[0]; ++i) c2t[i] = rand_fr();
    for (k = 0; k < (size_t)n * Krd; ++k) c2t[k * (2 * t + 1)] = ct[k * (t + 1)]; /* the same secret in both sharings */
    CHECK(hbmpc_stream_create(ctx, &stream) == ShareSuccess);
    CHECK(hbmpc_pipe_preprocessing_create(ctx, n, t, N, stream, &pre) == ShareSuccess);
    CHECK(hbmpc_pipe_part(pre, "ransha", &rs) == ShareSuccess && hbmpc_pipe_part(pre, "randousha", &rd) == ShareSuccess &&
          hbmpc_pipe_part(pre, "triplegen", &tg) == ShareSuccess);
    CHECK(hbmpc_pipe_part(pre, "nonsense", &tg) == InvalidInput && hbmpc_pipe_part(pre, "triplegen", &tg) == ShareSuccess);
    CHECK(hbmpc_pipe_buffer(rs, "coeffs", &dev, &elements) == ShareSuccess && dev != NULL && elements == sizeof co / sizeof co[0]);
    CHECK(hbmpc_pipe_buffer(rs, "no such buffer", &dev, &elements) == InvalidInput);
    CHECK(hbmpc_pipe_upload(rs, "coeffs", co, sizeof co / sizeof co[0]) == ShareSuccess);
    CHECK(hbmpc_pipe_upload(rs, "coeffs", co, sizeof co / sizeof co[0] + 1) == InvalidInput); /* beyond the buffer */
    CHECK(hbmpc_pipe_upload(rd, "coeffs_t", ct, sizeof ct / sizeof ct[0]) == ShareSuccess);
    CHECK(hbmpc_pipe_upload(rd, "coeffs_2t", c2t, sizeof c2t / sizeof c2t[0]) == ShareSuccess);
    CHECK(hbmpc_pipe_set_checked(pre, 1) == ShareSuccess && hbmpc_pipe_run(pre) == ShareSuccess);
    CHECK(hbmpc_pipe_verdict(pre, verdict) == ShareSuccess && verdict[0] == 0);
    CHECK(hbmpc_pipe_summary(tg, &sm) == ShareSuccess && sm.n_failed == 0);
    CHECK(hbmpc_pipe_download(tg, "a", a, (size_t)n * N) == ShareSuccess && hbmpc_pipe_download(tg, "b", b, (size_t)n * N) == ShareSuccess &&
          hbmpc_pipe_download(tg, "c", c, (size_t)n * N) == ShareSuccess);
    for (i = 0; i < N; ++i) { /* open triple i from the n parties' shares: c = a b */
        U256 sa[n], sb[n], sc[n], oa, ob, oc, ab, coeffs[t + 1];
        for (p = 0; p < n; ++p) sa[p] = a[p * N + i], sb[p] = b[p * N + i], sc[p] = c[p * N + i];
        CHECK(hbmpc_recover_secret(ctx, ids, deg, sa, n, n, t, coeffs, &nco, &oa) == ShareSuccess);
        CHECK(hbmpc_recover_secret(ctx, ids, deg, sb, n, n, t, coeffs, &nco, &ob) == ShareSuccess);
        CHECK(hbmpc_recover_secret(ctx, ids, deg, sc, n, n, t, coeffs, &nco, &oc) == ShareSuccess);
        CHECK(hbmpc_fr_op(ctx, 2, &oa, &ob, 1, &ab) == ShareSuccess && memcmp(&ab, &oc, sizeof ab) == 0);
    }
    /* the same handle as a HIP graph: c cleared, replayed, identical */
    CHECK(hbmpc_pipe_capture(pre) == ShareSuccess);
    memset(c2, 0, sizeof c2);
    CHECK(hbmpc_pipe_upload(tg, "c", c2, (size_t)n * N) == ShareSuccess);
    CHECK(hbmpc_pipe_replay(pre) == ShareSuccess && hbmpc_pipe_sync(pre) == ShareSuccess);
    CHECK(hbmpc_pipe_download(tg, "c", c2, (size_t)n * N) == ShareSuccess && memcmp(c, c2, sizeof c) == 0);
    /* a dealer that deals an inconsistent share: RanSha's verifiers must say so (share_gen.rs:516-530) */
    CHECK(hbmpc_pipe_deal(rs) == ShareSuccess);
    CHECK(hbmpc_pipe_buffer(rs, "S", &dev, NULL) == ShareSuccess);
    {
        U256 one;
        U256* at = (U256*)dev + (1 * n + 2) * Krs + 1; /* dealer 1's share for recipient 2, batch element 1 */
        CHECK(hbmpc_memcpy_d2h(ctx, &one, at, sizeof one, stream) == ShareSuccess && hbmpc_stream_sync(ctx, stream) == ShareSuccess);
        one.data[0] ^= 1;
        CHECK(hbmpc_memcpy_h2d(ctx, at, &one, sizeof one, stream) == ShareSuccess);
    }
    CHECK(hbmpc_pipe_finish(rs) == ShareSuccess && hbmpc_pipe_verdict(rs, verdict) == ShareSuccess && verdict[0] >= 1 && verdict[1] == 1);
    /* shapes the reference rejects / wrong handle kinds */
    CHECK(hbmpc_pipe_ransha_create(ctx, 4, 2, 3, 0, stream, &rs) == InvalidInput); /* n <= 2t */
    CHECK(hbmpc_pipe_triplegen_create(ctx, 4, 1, 4, stream, &rs) == InvalidInput); /* N not a multiple of 2t + 1 */
    CHECK(hbmpc_pipe_deal(tg) == InvalidInput && hbmpc_pipe_verdict(tg, verdict) == InvalidInput);
    hbmpc_pipe_destroy(pre);
    CHECK(hbmpc_stream_destroy(ctx, stream) == ShareSuccess);
    return 0;
}

int main(void) {
    hbmpc_ctx* ctx = NULL;
    int bad = 0;
    if (hbmpc_create(0, Bls12_381Fr, &ctx) != ShareSuccess) {
        printf("hbmpc_create failed: %s\n", hbmpc_last_error(NULL));
        return 2;
    }
    bad += robust_roundtrip(ctx);
    bad += nonrobust_roundtrip(ctx);
    bad += seeded_and_wire(ctx);
    bad += preprocessing_pipeline(ctx);
    hbmpc_destroy(ctx);
    bad += goldilocks_roundtrip();
    if (bad == 0) printf("C ABI round trips passed (%s)\n", hbmpc_version());
    return bad;
}
