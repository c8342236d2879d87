"""oracle/spec.py (Python big-int restatement) against every literal-input test the reference
holds for the path and the published field constants."""
from itertools import combinations

import pytest

from oracle import spec as S

R = S.R_MOD


def test_field_constants():
    # SURVEY.md Appendix A; the 2^32-th root of unity is the widely published bls12-381 constant
    # (e.g. the last entry of the KZG "SCALE2_ROOT_OF_UNITY" tables; quoted from memory).
    assert R == 52435875175126190479447740508185965837690552500527637822603658699938581184513
    assert S.TWO_ADIC_ROOT == 0x16A2A19EDFE81F20D09B681922C813B4B63683508C2280B93829971F439F0D2B
    assert S.TWO_ADIC_ROOT == 10238227357739495823651030575849232062558860180284477541189508159991286009131
    for n, size, w in [(4, 4, 0x8D51CCCE760304D0EC030002760300000001000000000000),
                       (16, 16, 0x20B1CE9140267AF9DD1C0AF834CEC32C17BEB312F20B6F7653EA61D87742BCCE),
                       (31, 32, 0x50E0903A157988BAB4BCD40E22F55448BF6E88FB4C38FB8A360C60997369DF4E)]:
        assert S.domain_size(n) == size
        assert S.domain_omega(n) == w
        assert pow(w, size, R) == 1 and pow(w, size // 2, R) == R - 1
    assert pow(2, 256, R) == 0x1824B159ACC5056F998C4FEFECBC4FF55884B7FA0003480200000001FFFFFFFE


def test_domain_does_not_contain_zero():  # shamir.rs:453-458 (n = 100)
    assert all(S.domain_element(100, j) != 0 for j in range(128))


def test_poly_derivative():  # robust_interpolate.rs:636-644
    assert S.poly_derivative([3, 2, 1]) == [2, 2]


def test_make_vandermonde_basic():  # common/share/mod.rs:87-136
    n, t = 4, 2
    v = S.make_vandermonde(n, t)
    assert len(v) == n and all(len(r) == t + 1 for r in v)
    assert v[0] == [1, 1, 1]
    a1 = S.domain_element(n, 1)
    assert v[1] == [1, a1, a1 * a1 % R]
    assert v[2][1] == S.domain_element(n, 2)
    assert v[3][2] == pow(S.domain_element(n, 3), 2, R)


def test_apply_vandermonde_basic():  # common/share/mod.rs:138-172
    n, t = 4, 2
    v = S.make_vandermonde(n, t)
    shares = [S.Share(1, 0, 2), S.Share(2, 0, 2), S.Share(3, 0, 2)]
    y = S.apply_vandermonde(v, shares)
    assert len(y) == n
    for j in range(n):
        a = S.domain_element(n, j)
        assert y[j].v == (1 + 2 * a + 3 * a * a) % R
        assert (y[j].id, y[j].degree) == (0, 2)  # keeps the INPUT id/degree
    with pytest.raises(S.InvalidInput):
        S.apply_vandermonde(v, shares[:2])


def test_robust_interpolate_fnt_optimistic_case():  # robust_interpolate.rs:645-680
    n, t = 16, 2
    shares = S.compute_shares([7, 3, 5], n, t)
    assert S.robust_interpolate_fnt(t, n, shares[: 2 * t + 1]) == [7, 3, 5]


def _rand_shares(seed, secret, n, t):
    rng = S.SplitMix64(seed)
    coeffs = [secret] + [rng.fr() for _ in range(t)]
    return coeffs, S.compute_shares(coeffs, n, t)


def test_reed_solomon_erasure():  # :682-704
    t, n = 2, 8
    _, shares = _rand_shares(1, 42, n, t)
    vals = [s.v for s in shares]
    for i in (1, 2):
        vals[i] = 0
    assert S.gao_rs_decode(vals, t + 1, n, [1, 2])[0] == 42


def test_reed_solomon_error():  # :705-726
    t, n = 2, 10
    _, shares = _rand_shares(2, 42, n, t)
    vals = [s.v for s in shares]
    vals[2] += 5
    vals[4] += 3
    assert S.gao_rs_decode(vals, t + 1, n, [])[0] == 42


def test_reed_solomon_error_all_triples():  # :727-756
    t, n = 3, 10
    coeffs, shares = _rand_shares(3, 42, n, t)
    base = [s.v for s in shares]
    for tr in combinations(range(n), 3):
        c = list(base)
        c[tr[0]] += 5
        c[tr[1]] += 3
        c[tr[2]] += 3
        assert S.gao_rs_decode(c, t + 1, n, []) == coeffs


def test_oec_protocol():  # :757-789
    t, n = 2, 10
    _, shares = _rand_shares(4, 42, n, t)
    shares[0].v = (shares[0].v + 999) % R
    shares[5].v = (shares[5].v + 999) % R
    assert S.oec_decode(n, t, shares)[1] == 42


def test_robust_interpolate_full():  # :790-826
    t, n = 3, 10
    _, shares = _rand_shares(5, 42, n, t)
    for i in (1, 4):
        shares[i] = S.share_add(shares[i], S.Share(7, i, t))
    assert S.recover_secret(shares, n, t)[1] == 42


def test_robust_interpolate_all_corruption_combinations():  # :827-876
    t, n = 2, 7
    coeffs, base = _rand_shares(6, 42, n, t)
    for k in range(1, t + 1):
        for idx in combinations(range(n), k):
            sh = [S.Share(s.v, s.id, s.degree) for s in base]
            for i in idx:
                sh[i].v = (sh[i].v + 999) % R
            c, z = S.recover_secret(sh, n, t)
            assert z == 42 and c == coeffs, idx


def test_batch_recover_secret_matches_per_chunk():  # :880-927
    n, t, degree, batch_len = 10, 3, 3, 16
    rng = S.SplitMix64(7)
    polys = [[rng.fr() for _ in range(degree + 1)] for _ in range(batch_len)]
    ev = [(i, [S.p_eval(p, S.domain_element(n, i)) for p in polys]) for i in range(n)]
    ev.reverse()
    batched = S.batch_recover_secret(ev, n, degree, t)
    assert len(batched) == batch_len
    for c in range(batch_len):
        shares = [S.Share(vals[c], sid, degree) for sid, vals in ev]
        per_chunk, _ = S.recover_secret(shares, n, t)
        per_chunk = per_chunk + [0] * (degree + 1 - len(per_chunk))
        assert batched[c] == per_chunk
        assert batched[c][0] == polys[c][0]


def test_batch_recover_secret_with_corruption():  # :931-967
    n, t, degree, batch_len = 10, 3, 3, 8
    rng = S.SplitMix64(8)
    polys = [[rng.fr() for _ in range(degree + 1)] for _ in range(batch_len)]
    ev = [(i, [S.p_eval(p, S.domain_element(n, i)) for p in polys]) for i in range(n)]
    for bad in range(t):
        for c in range(batch_len):
            ev[bad][1][c] = (ev[bad][1][c] + (c + 1) * 7 + bad) % R
    batched = S.batch_recover_secret(ev, n, degree, t)
    for c in range(batch_len):
        assert batched[c][0] == polys[c][0]


def test_nonrobust_shamir_cases():  # shamir.rs:250-348
    rng = S.SplitMix64(9)
    coeffs = [918520] + [rng.fr() for _ in range(5)]
    shares = S.compute_shares(coeffs, 6, 5)
    assert S.nonrobust_recover_secret(shares, 6)[1] == 918520
    a = S.compute_shares([10] + [rng.fr() for _ in range(5)], 6, 5)
    b = S.compute_shares([20] + [rng.fr() for _ in range(5)], 6, 5)
    assert S.nonrobust_recover_secret([S.share_add(x, y) for x, y in zip(a, b)], 6)[1] == 30
    c = S.compute_shares([55] + [rng.fr() for _ in range(5)], 8, 5)
    assert S.nonrobust_recover_secret([S.share_mul_scalar(x, 3) for x in c], 8)[1] == 165
    shares[2].degree = 4
    with pytest.raises(S.DegreeMismatch):
        S.nonrobust_recover_secret(shares, 6)
    s3 = S.compute_shares([918520, rng.fr(), rng.fr()], 3, 2)
    with pytest.raises(S.InsufficientShares):
        S.nonrobust_recover_secret(s3[1:], 3)
    b[0].id = 6
    with pytest.raises(S.IdMismatch):
        S.share_add(a[0], b[0])


def test_c_abi_smoke_inputs():  # ffi/tests/secret_share.c:64-118 (U256{3,3,22,22}, n=6, degree 2, t=1)
    secret = S.from_limbs([3, 3, 22, 22])
    assert secret < R
    rng = S.SplitMix64(10)
    shares = S.compute_shares([secret, rng.fr(), rng.fr()], 6, 2)
    coeffs, z = S.recover_secret(shares, 6, 1)
    assert z == secret and len(coeffs) == 3


def test_recover_secret_validation_order():  # robust_interpolate.rs:100-142
    n, t = 7, 2
    _, sh = _rand_shares(11, 5, n, t)
    with pytest.raises(S.InvalidInput):
        S.recover_secret(sh, 6, 2)  # n < 3t+1
    with pytest.raises(S.InvalidInput):
        S.recover_secret([], n, t)
    bad = [S.Share(s.v, s.id, s.degree) for s in sh]
    bad[3].degree = 1
    with pytest.raises(S.DegreeMismatch):
        S.recover_secret(bad, n, t)
    dup = [S.Share(s.v, s.id, s.degree) for s in sh]
    dup[1].id = 0
    with pytest.raises(S.InvalidInput):
        S.recover_secret(dup, n, t)
    with pytest.raises(S.InvalidInput):
        S.recover_secret(sh[:4], n, t)  # < degree + t + 1
    # t+1 corrupted: cannot decode
    over = [S.Share(s.v, s.id, s.degree) for s in sh]
    for i in range(t + 1):
        over[i].v = (over[i].v + 1) % R
    with pytest.raises(S.DecodingError):
        S.recover_secret(over, n, t)


def test_beaver_and_truncpr_algebra():
    # config 1 of BASELINE.json (n=4, t=1, 5 Beaver muls, tests/node_test.rs:447-453) as pure algebra
    n, t = 4, 1
    rng = S.SplitMix64(12)
    for _ in range(5):
        x, y, a, b = (rng.fr() for _ in range(4))
        sx = S.compute_shares([x, rng.fr()], n, t)
        sy = S.compute_shares([y, rng.fr()], n, t)
        sa = S.compute_shares([a, rng.fr()], n, t)
        sb = S.compute_shares([b, rng.fr()], n, t)
        sc = S.compute_shares([a * b % R, rng.fr()], n, t)
        dsh, esh = zip(*[S.beaver_open_shares(sa[i], sb[i], sx[i], sy[i]) for i in range(n)])
        d = S.recover_secret(list(dsh), n, t)[1]
        e = S.recover_secret(list(esh), n, t)[1]
        assert d == (a - x) % R and e == (b - y) % R
        z = [S.beaver_finalize(sc[i], sx[i], sy[i], d, e) for i in range(n)]
        assert S.recover_secret(z, n, t)[1] == x * y % R
    # TruncPr: d*2^m + c' - r' == a  (exact identity of truncpr.rs:215-220)
    k, m = 16, 4
    a_val = 0x1234
    sa = S.compute_shares([a_val, rng.fr()], n, t)
    bits = [rng.next() & 1 for _ in range(m)]
    sbits = [S.compute_shares([bv, rng.fr()], n, t) for bv in bits]
    rint = rng.next() & 0xFFFF
    srint = S.compute_shares([rint, rng.fr()], n, t)
    rd = [S.truncpr_rdash([sbits[j][i] for j in range(m)], m, i, t) for i in range(n)]
    op = [S.truncpr_open_share(sa[i], rd[i], srint[i], k, m) for i in range(n)]
    c = S.recover_secret(op, n, t)[1]
    rdash = sum(bv << j for j, bv in enumerate(bits))
    assert c == (a_val + (1 << (k - 1)) + (rint << m) + rdash) % R
    dsh = [S.truncpr_finalize(sa[i], rd[i], c, m) for i in range(n)]
    dval = S.recover_secret(dsh, n, t)[1]
    assert dval in (a_val >> m, (a_val >> m) + 1)
    assert S.mod_pow_2_from_field(0xABCD, 0) == 0
    assert S.mod_pow_2_from_field(0xABCD, 7) == 0xABCD & 0x7F
    assert S.mod_pow_2_from_field(0xABCD, 8) == 0xCD
    assert S.mod_pow_2_from_field(R - 1, 256) == R - 1
    with pytest.raises(S.InvalidInput):
        S.mod_pow_2_from_field(R - 1, 300)
