"""CPU tests of the oracle's restatement of the preprocessing producers (oracle/spec.py: ransha, randousha; reference
share_gen/share_gen.rs, double_share/double_share_generation.rs, ran_dou_sha/mod.rs).  What pins it: the properties the
protocols exist for -- the outputs are consistent degree-t (and 2t) sharings of the Vandermonde mix of the dealers'
secrets, double sharings hide equal secrets, the verifiers accept honest runs and reject a dealer that cheats -- in both
fields."""
import random

import pytest

from oracle import spec as SFR
from oracle.spec_gl import S as SGL


def _coeffs(S, rng, n, K, deg, secrets=None):
    return [[[secrets[p][k] if secrets else rng.randrange(S.R_MOD)] + [rng.randrange(S.R_MOD) for _ in range(deg)] for k in range(K)]
            for p in range(n)]


@pytest.mark.parametrize("S", [SFR, SGL], ids=["fr", "goldilocks"])
@pytest.mark.parametrize("n,t,K", [(4, 1, 2), (7, 2, 3), (10, 3, 2)])
def test_ransha_outputs_are_sharings_of_the_mixed_secrets(S, n, t, K):
    rng = random.Random(n * 7 + K)
    co = _coeffs(S, rng, n, K, t)
    out, ok = S.ransha(co, n, t)
    assert ok == [True] * (2 * t) and all(len(o) == (n - 2 * t) * K for o in out)
    vdm = S.make_vandermonde(n, n - 1)
    for k in range(K):
        for idx in range(n - 2 * t):
            i = 2 * t + idx
            poly, sec = S.recover_secret([S.Share(out[j][k * (n - 2 * t) + idx], j, t) for j in range(n)], n, t)
            assert S.p_degree(poly) <= t
            assert sec == sum(vdm[i][p] * co[p][k][0] for p in range(n)) % S.R_MOD   # row i of the Vandermonde mix of the dealers' secrets
    # all n senders to the verifiers: same verdicts
    assert S.ransha(co, n, t, verify_senders=n)[1] == ok


@pytest.mark.parametrize("S", [SFR, SGL], ids=["fr", "goldilocks"])
def test_randousha_double_sharings_and_cheating_dealer(S):
    n, t, K = 7, 2, 3
    rng = random.Random(99)
    ct = _coeffs(S, rng, n, K, t)
    c2t = _coeffs(S, rng, n, K, 2 * t, secrets=[[ct[p][k][0] for k in range(K)] for p in range(n)])
    a, b, ok = S.randousha(ct, c2t, n, t)
    assert ok == [True] * (n - t - 1) and all(len(x) == (t + 1) * K for x in a)
    for k in range(K):
        for i in range(t + 1):
            pa, sa = S.nonrobust_recover_secret([S.Share(a[j][k * (t + 1) + i], j, t) for j in range(n)], n)
            pb, sb = S.nonrobust_recover_secret([S.Share(b[j][k * (t + 1) + i], j, 2 * t) for j in range(n)], n)
            assert sa == sb and S.p_degree(pa) <= t and S.p_degree(pb) <= 2 * t
    c2t[4][2][0] = (c2t[4][2][0] + 1) % S.R_MOD         # different secrets in the two sharings of one dealt element
    assert S.randousha(ct, c2t, n, t)[2] == [False] * (n - t - 1)
