/* A plain C99 caller of include/hbmpc_hip.h -- the shape of the reference's own C test of its exported ABI
 * (mpc/src/ffi/tests/secret_share.c: create shares of a literal U256 secret for n = 6 parties, recover, compare),
 * restated against this library's entry points, for both share kinds and both fields.  Needs an MI355X. */
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

int main(void) {
    hbmpc_ctx* ctx = NULL;
    int bad = 0;
    if (hbmpc_create(0, Bls12_381Fr, &ctx) != ShareSuccess) {
        printf("hbmpc_create failed: %s\n", hbmpc_last_error(NULL));
        return 2;
    }
    bad += robust_roundtrip(ctx);
    bad += nonrobust_roundtrip(ctx);
    hbmpc_destroy(ctx);
    bad += goldilocks_roundtrip();
    if (bad == 0) printf("C ABI round trips passed (%s)\n", hbmpc_version());
    return bad;
}
