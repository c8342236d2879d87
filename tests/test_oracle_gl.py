"""CPU tests of the Goldilocks oracle (oracle/spec_gl.py = oracle/spec.py with the constants of
mpc/src/common/math/goldilocks.rs:4-13).  What pins it: the published field constants and the algebraic
properties the reference's own tests check, instantiated in the small field."""
import itertools
import random

import pytest

from oracle import spec as SFR
from oracle.spec_gl import P, S


def test_field_constants():
    assert P == 18446744069414584321 == 2 ** 64 - 2 ** 32 + 1          # goldilocks.rs:6
    assert S.GENERATOR == 7 and S.TWO_ADICITY == 32                     # goldilocks.rs:7; p - 1 = 2^32 (2^32 - 1)
    # GENERATOR^((p-1)/2^32): the value every Goldilocks library publishes as its 2^32-th root of unity
    assert S.TWO_ADIC_ROOT == 1753635133440165772
    assert pow(S.TWO_ADIC_ROOT, 1 << 32, P) == 1 and pow(S.TWO_ADIC_ROOT, 1 << 31, P) == P - 1
    assert SFR.R_MOD != P and SFR.TWO_ADIC_ROOT != S.TWO_ADIC_ROOT        # the Fr oracle is untouched


def test_domain_does_not_contain_zero_and_vandermonde():                # common/share/mod.rs tests :84-136
    for n in (4, 6, 16, 31):
        els = [S.domain_element(n, j) for j in range(n)]
        assert 0 not in els and len(set(els)) == n
        v = S.make_vandermonde(n, 3)
        assert all(row[0] == 1 and row[2] == row[1] * row[1] % P for row in v)


@pytest.mark.parametrize("n,t", [(4, 1), (7, 2), (10, 3), (16, 5)])
def test_share_recover_roundtrip_with_errors(n, t):
    rng = random.Random(n)
    coeffs = [rng.randrange(P) for _ in range(t + 1)]
    shares = S.compute_shares(coeffs, n, t)
    got, sec = S.recover_secret(shares, n, t)
    assert got == S.p_norm(coeffs) and sec == coeffs[0]
    for bad in itertools.islice(itertools.combinations(range(n), t), 20):
        cor = [S.Share((s.v + 1 + i) % P if i in bad else s.v, s.id, s.degree) for i, s in enumerate(shares)]
        got, sec = S.recover_secret(cor, n, t)
        assert got == S.p_norm(coeffs)
    with pytest.raises(S.ShareErr):
        S.recover_secret(shares[: 2 * t], n, t)                          # not enough shares


def test_batch_recover_matches_per_chunk():                             # robust_interpolate.rs :897-967 shape
    n, t, d, G = 10, 3, 3, 6
    rng = random.Random(5)
    polys = [[rng.randrange(P) for _ in range(d + 1)] for _ in range(G)]
    ev = [(i, [S.p_eval(p, S.domain_element(n, i)) for p in polys]) for i in range(n)]
    assert S.batch_recover_secret(ev, n, d, t) == polys
    for b in range(t):
        ev[b] = (ev[b][0], [(v + 7 * (c + 1) + b) % P for c, v in enumerate(ev[b][1])])
    out = S.batch_recover_secret(ev, n, d, t)
    assert [o[0] for o in out] == [p[0] for p in polys]
