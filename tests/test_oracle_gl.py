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


class SpecEngine:
    """oracle/spec_gl.py behind the engine interface tests/golden_util.py drives (numpy uint64 in, (rc, ...) out)"""

    def __init__(self, S):
        self.S = S

    @staticmethod
    def _a(x):
        import numpy as np
        return np.array(x, dtype=np.uint64)

    def _try(self, fn):
        try:
            return (0,) + tuple(fn())
        except self.S.ShareErr as e:
            return (e.code,) + (None,) * 4

    def make_vandermonde(self, n, d):
        return 0, self._a(self.S.make_vandermonde(n, d))

    def compute_shares(self, coeffs, n, d):
        cols = [[s.v for s in self.S.compute_shares([int(c) for c in row], n, d)] for row in coeffs]
        return 0, self._a(cols).T.copy() if len(cols) else self._a([[]] * n)

    def vandermonde_apply(self, x, n, d):
        return self.compute_shares(x, n, d)

    def batch_recover(self, ids, ev, n, d, t):
        import numpy as np
        evs = [(i, [int(v) for v in ev[k]]) for k, i in enumerate(ids)]
        try:
            res = self.S.batch_recover_secret(evs, n, d, t)
        except self.S.ShareErr as e:
            return e.code, None, None, None
        return 0, self._a([r + [0] * (d + 1 - len(r)) for r in res]), [len(r) for r in res], np.zeros(len(res))

    def batch_recover_p0(self, ids, ev, n, d, t):
        rc, co, nco, st = self.batch_recover(ids, ev, n, d, t)
        return rc, (co[:, 0] if rc == 0 else None), st

    def recover_secret(self, ids, degrees, vals, n, t):
        sh = [self.S.Share(int(v), i, dg) for v, i, dg in zip(vals, ids, degrees)]
        r = self._try(lambda: self.S.recover_secret(sh, n, t))
        return (r[0], self._a(r[1]), self._a(r[2])) if r[0] == 0 else (r[0], None, None)

    def gao_rs_decode(self, received, k, n, erasures):
        r = self._try(lambda: (self.S.gao_rs_decode([int(v) for v in received], k, n, erasures),))
        return (0, self._a(r[1])) if r[0] == 0 else (r[0], None)

    def nonrobust_recover_secret(self, ids, degrees, vals, n):
        sh = [self.S.Share(int(v), i, dg) for v, i, dg in zip(vals, ids, degrees)]
        r = self._try(lambda: self.S.nonrobust_recover_secret(sh, n))
        return (r[0], self._a(r[1]), self._a(r[2])) if r[0] == 0 else (r[0], None, None)

    def triple_local(self, a, b, r2t):
        return 0, self._a([(int(x) * int(y) - int(z)) % P for x, y, z in zip(a, b, r2t)])

    def triple_finalize(self, rt, opened):
        return 0, self._a([(int(x) + int(y)) % P for x, y in zip(rt, opened)])

    def beaver_open_shares(self, a, b, x, y):
        return (0, self._a([(int(p) - int(q)) % P for p, q in zip(a, x)]),
                self._a([(int(p) - int(q)) % P for p, q in zip(b, y)]))

    def beaver_finalize(self, c, x, y, d, e):
        return 0, self._a([(int(c[i]) - int(d[i]) * int(e[i]) - int(d[i]) * int(y[i]) - int(e[i]) * int(x[i])) % P
                           for i in range(len(c))])


def test_goldilocks_golden_vectors_match_the_oracle():
    """tests/golden/hbmpc_golden_gl.json (generated by tests/golden/make_golden.py) is what oracle/spec_gl.py
    computes today: guards the committed fixtures against drift of the restatement."""
    from tests import golden_util as GU
    assert GU.run_all(SpecEngine(S), field="goldilocks") > 90
