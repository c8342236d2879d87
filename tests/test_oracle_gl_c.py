"""The C restatement over Goldilocks (oracle/hbmpc_oracle.c built with -DORACLE_GOLDILOCKS, bound by oracle/cref_gl.py)
against the Python big-int restatement (oracle/spec_gl.py): the small field now has two independent checkers, as Fr has
(VERDICT r2 item 8).  CPU only."""
import random

import numpy as np

from oracle import cref_gl as CG
from oracle.spec_gl import P, S
from tests import golden_util as GU


def test_goldilocks_golden_vectors_through_the_c_restatement():
    # the fixtures were generated from spec_gl.py (tests/golden/make_golden.py): every case, error codes included
    assert GU.run_all(CG, field="goldilocks") > 90


def test_constants_and_domain():
    assert CG.P == P
    for n in (4, 7, 16, 31, 200):
        assert [int(v) for v in CG.domain_elements(n, n)] == [S.domain_element(n, j) for j in range(n)]


def test_random_encode_decode_with_corruption_matches_spec():
    rng = random.Random(77)
    for n, t, d in ((7, 2, 2), (16, 5, 5), (16, 5, 10), (31, 10, 10), (40, 13, 5)):
        G = 9
        polys = [[rng.randrange(P) for _ in range(d + 1)] for _ in range(G)]
        x = np.array(polys, dtype=np.uint64)
        rc, y = CG.vandermonde_apply(x, n, d)
        assert rc == 0
        want = [[S.p_eval(p, S.domain_element(n, j)) for p in polys] for j in range(n)]
        assert y.tolist() == want
        # corrupt up to t senders in some chunks, drop a few senders, shuffle the arrival order
        ids = list(range(n))
        rng.shuffle(ids)
        ids = ids[: max(d + t + 1, n - 2)]
        ev = y[ids].copy()
        for g in range(0, G, 2):
            for k in rng.sample(range(len(ids)), min(t, len(ids) - (d + t + 1))):
                ev[k, g] = (int(ev[k, g]) + 1 + g) % P
        rc, co, nco, st = CG.batch_recover(ids, ev, n, d, t)
        try:
            res = S.batch_recover_secret([(i, [int(v) for v in ev[k]]) for k, i in enumerate(ids)], n, d, t)
            assert rc == 0 and [list(map(int, co[g][: nco[g]])) for g in range(G)] == res
        except S.ShareErr as e:
            assert rc == e.code


def test_elementwise_against_python_ints():
    N = 257
    a, b, c, d, e = (CG.fill_random(100 + i, N) for i in range(5))
    assert all(int(v) < P for v in a)
    ai, bi, ci, di, ei = ([int(v) for v in z] for z in (a, b, c, d, e))
    assert CG.triple_local(a, b, c)[1].tolist() == [(x * y - z) % P for x, y, z in zip(ai, bi, ci)]
    assert CG.triple_finalize(a, b)[1].tolist() == [(x + y) % P for x, y in zip(ai, bi)]
    rc, ds, es = CG.beaver_open_shares(a, b, c, d)
    assert ds.tolist() == [(x - z) % P for x, z in zip(ai, ci)] and es.tolist() == [(y - w) % P for y, w in zip(bi, di)]
    assert CG.beaver_finalize(a, b, c, d, e)[1].tolist() == [(ai[i] - di[i] * ei[i] - di[i] * ci[i] - ei[i] * bi[i]) % P for i in range(N)]
