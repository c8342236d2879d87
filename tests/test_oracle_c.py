"""The C oracle (oracle/hbmpc_oracle.c) against the golden vectors and, on seeded random inputs,
against the Python big-int restatement (oracle/spec.py)."""
import numpy as np
import pytest

from oracle import cref as E
from oracle import spec as S
from tests import golden_util as GU

R = S.R_MOD


def test_golden_vectors():
    assert GU.run_all(E) > 100


def test_fill_random_matches_spec():
    a = E.fill_random(0xC0FFEE01, 64)
    rng = S.SplitMix64(0xC0FFEE01)
    assert E.u256_to_ints(a) == [rng.fr() for _ in range(64)]


def test_field_ops_random():
    a, b = E.fill_random(1, 512), E.fill_random(2, 512)
    ai, bi = E.u256_to_ints(a), E.u256_to_ints(b)
    assert E.u256_to_ints(E.fr_binop("mul", a, b)) == [x * y % R for x, y in zip(ai, bi)]
    assert E.u256_to_ints(E.fr_binop("add", a, b)) == [(x + y) % R for x, y in zip(ai, bi)]
    assert E.u256_to_ints(E.fr_binop("sub", a, b)) == [(x - y) % R for x, y in zip(ai, bi)]
    assert E.u256_to_ints(E.fr_inv(a[:16])) == [S.inv(x) for x in ai[:16]]
    edge = E.ints_to_u256([0, 1, R - 1, R - 2, 2, (1 << 255) % R])
    ei = E.u256_to_ints(edge)
    for x in range(len(ei)):
        rot = np.roll(edge, x, axis=0)
        ri = E.u256_to_ints(rot)
        assert E.u256_to_ints(E.fr_binop("mul", edge, rot)) == [p * q % R for p, q in zip(ei, ri)]
        assert E.u256_to_ints(E.fr_binop("add", edge, rot)) == [(p + q) % R for p, q in zip(ei, ri)]
        assert E.u256_to_ints(E.fr_binop("sub", edge, rot)) == [(p - q) % R for p, q in zip(ei, ri)]


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 7, 10, 16, 20, 31, 33, 100, 255])
def test_domain(n):
    el = E.u256_to_ints(E.domain_elements(n, n))
    assert el == [S.domain_element(n, j) for j in range(n)]
    assert 0 not in el and len(set(el)) == n


@pytest.mark.parametrize("n,d", [(4, 1), (16, 5), (31, 10), (31, 30), (255, 84), (64, 63), (7, 0)])
def test_compute_shares_random(n, d):
    B = 5
    coeffs = E.fill_random(100 + n, B * (d + 1)).reshape(B, d + 1, 4)
    rc, sh = E.compute_shares(coeffs, n, d)
    assert rc == 0
    ci = E.u256_to_ints(coeffs)
    want = [[S.compute_shares(ci[b], n, d)[j].v for b in range(B)] for j in range(n)]
    assert E.u256_to_ints(sh) == want
    rc, y = E.vandermonde_apply(coeffs, n, d)  # same linear map
    assert rc == 0 and GU.eq(y, sh)


def test_compute_shares_errors():
    c = E.fill_random(1, 6).reshape(1, 6, 4)
    assert E.compute_shares(c, 5, 5)[0] == S.InvalidInput.code
    assert E.compute_shares(c, 6, 5)[0] == 0


@pytest.mark.parametrize("n,t,d,G,nbad", [(7, 2, 2, 12, 2), (10, 3, 3, 10, 3), (16, 5, 5, 8, 5), (16, 5, 10, 6, 0),
                                          (31, 10, 10, 6, 10), (13, 4, 4, 6, 5), (4, 1, 1, 9, 1)])
def test_batch_recover_random_vs_spec(n, t, d, G, nbad):
    rng = S.SplitMix64(n * 1000 + d)
    polys = [[rng.fr() for _ in range(d + 1)] for _ in range(G)]
    ids = list(range(n))
    # deterministic shuffle
    for i in range(n - 1, 0, -1):
        j = rng.next() % (i + 1)
        ids[i], ids[j] = ids[j], ids[i]
    ev = [(i, [S.p_eval(p, S.domain_element(n, i)) for p in polys]) for i in ids]
    for c in range(G):
        k = c % (nbad + 1)  # 0..nbad corrupted senders in chunk c (nbad may exceed t: failure case)
        for pos in range(k):
            ev[(pos * 3 + c) % n][1][c] = (ev[(pos * 3 + c) % n][1][c] + 1 + pos) % R
    evals = E.ints_to_u256([v for _, v in ev])
    rc, co, nco, st = E.batch_recover(ids, evals, n, d, t)
    # spec, chunk by chunk (so that one failing chunk does not hide the others)
    first_err = 0
    for c in range(G):
        one = [(sid, [vals[c]]) for sid, vals in ev]
        try:
            want = S.batch_recover_secret(one, n, d, t)[0]
            assert list(E.u256_to_ints(co[c])) == want + [0] * (d + 1 - len(want)), c
            assert nco[c] == len(want)
            assert st[c] in (0, 1)
        except S.ShareErr as e:
            assert st[c] == e.code, c
            first_err = first_err or e.code
    assert rc == first_err


def test_recover_secret_errors():
    n, t = 7, 2
    rng = S.SplitMix64(5)
    sh = S.compute_shares([5, rng.fr(), rng.fr()], n, t)
    vals = E.ints_to_u256([s.v for s in sh])
    ids, deg = list(range(n)), [t] * n
    assert E.recover_secret(ids, deg, vals, 6, 2)[0] == 4
    assert E.recover_secret([], [], vals[:0], n, t)[0] == 4
    assert E.recover_secret(ids, [2, 2, 2, 1, 2, 2, 2], vals, n, t)[0] == 2
    assert E.recover_secret([0, 0, 2, 3, 4, 5, 6], deg, vals, n, t)[0] == 4
    assert E.recover_secret([0, 1, 2, 3, 4, 5, 7], deg, vals, n, t)[0] == 4
    assert E.recover_secret(ids[:4], deg[:4], vals[:4], n, t)[0] == 4
    rc, co, sec = E.recover_secret(ids[::-1], deg, vals[::-1].copy(), n, t)
    assert rc == 0 and E.u256_to_ints(sec) == 5 and len(co) == 3


def test_elementwise_random_vs_spec():
    N = 33
    a, b, c, d, e = (E.fill_random(s, N) for s in range(20, 25))
    ai, bi, ci, di, ei = (E.u256_to_ints(x) for x in (a, b, c, d, e))
    assert E.u256_to_ints(E.triple_local(a, b, c)[1]) == [(x * y - z) % R for x, y, z in zip(ai, bi, ci)]
    assert E.u256_to_ints(E.beaver_finalize(a, b, c, d, e)[1]) == [
        (ai[i] - di[i] * ei[i] - di[i] * ci[i] - ei[i] * bi[i]) % R for i in range(N)]
    for m in (0, 1, 7, 8, 9, 64, 100, 254, 255, 256):
        got = E.u256_to_ints(E.truncpr_finalize(a, b, c, m)[1])
        want = [S.truncpr_finalize(S.Share(ai[i], 0, 1), S.Share(bi[i], 0, 1), ci[i], m).v for i in range(N)]
        assert got == want, m
    assert E.truncpr_finalize(a, b, c, 257)[0] == 4
    assert E.truncpr_open_share(a, b, c, 0, 4)[0] == 4
