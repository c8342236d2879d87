"""Randomised shape sweep (fixed seeds): random (n, t, d), random subsets and arrival orders of senders, random
corruption patterns, both fields -- every result compared with the big-int oracle (oracle/spec.py,
oracle/spec_gl.py), error codes included.  Complements the hand-picked shapes of test_gpu_parity.py /
test_gpu_gold.py: the kernels are dispatched by shape (pruned-FFT sizes, register-resident m <= 16, generic
fall-backs, OEC/Gao block sizes 64/128/256), so a sweep walks the dispatch boundaries."""
import random

import numpy as np
import pytest

from __graft_entry__ import load_package
from oracle import spec as SFR
from oracle.spec_gl import S as SGL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engines():
    pkg = load_package()
    e = {"fr": pkg.Engine(0), "goldilocks": pkg.Engine(0, field="goldilocks")}
    yield e
    for v in e.values():
        v.close()


def to_arr(field, rows):  # rows: nested lists of python ints
    a = np.array(rows, dtype=object)
    if field == "goldilocks":
        return a.astype(np.uint64)
    out = np.zeros(a.shape + (4,), dtype=np.uint64)
    for idx in np.ndindex(a.shape):
        v = int(a[idx])
        out[idx] = [(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)]
    return out


def to_int(field, x):
    if field == "goldilocks":
        return int(x)
    return sum(int(x[k]) << (64 * k) for k in range(4))


def rand_shape(rng):
    kind = rng.randrange(5)
    if kind == 0:
        n = rng.randint(4, 16)
    elif kind == 1:
        n = rng.randint(17, 40)
    elif kind == 2:
        n = rng.choice([63, 64, 65, 100, 127, 128, 129])
    elif kind == 3:
        n = rng.choice([200, 255])
    else:
        n = rng.randint(4, 70)
    t = rng.randint(1, (n - 1) // 3)
    d = rng.choice([t, min(2 * t, n - t - 1), rng.randint(0, n - t - 1)])
    return n, t, d


@pytest.mark.parametrize("field", ["fr", "goldilocks"])
@pytest.mark.parametrize("seed", range(24))
def test_random_encode_decode(engines, field, seed):
    eng, S = engines[field], (SFR if field == "fr" else SGL)
    P = S.R_MOD
    rng = random.Random(seed * 7919 + (1 if field == "fr" else 2))
    n, t, d = rand_shape(rng)
    G = rng.choice([1, 2, 3]) if n > 100 else rng.choice([1, 3, 17, 40])
    polys = [[rng.randrange(P) for _ in range(d + 1)] for _ in range(G)]
    for p in polys[:2]:                                       # low-degree / zero polynomials among them
        for k in range(rng.randint(0, d + 1)):
            p[d - k] = 0
    eng.set_small_batch_chunks(0)                             # the lane-per-chunk (FFT / generic) kernels ...
    rc0, y0 = eng.vandermonde_apply(to_arr(field, polys), n, d)
    eng.set_small_batch_chunks(8192)                          # ... and the default wave-per-chunk kernel: same bytes
    rc, y = eng.vandermonde_apply(to_arr(field, polys), n, d)
    assert rc == 0 and rc0 == 0 and np.array_equal(y, y0)
    for g in range(min(G, 3)):
        want = [s.v for s in S.compute_shares(polys[g], n, d)] if n > d else None
        assert [to_int(field, y[j, g]) for j in range(n)] == want
    # a random set of senders in random arrival order, possibly too few
    S_cnt = rng.choice([n, rng.randint(d + t + 1, n), rng.randint(max(1, d + t - 1), n)])
    ids = rng.sample(range(n), S_cnt)
    ev = {i: [to_int(field, y[i, g]) for g in range(G)] for i in ids}
    # random corruption: none / <= t / more than t.  The big-int oracle pays one O(n^3) OEC round per error
    # (3.5 s per round at n = 255), so large n gets at most two errors per chunk.
    kmax = t if n <= 40 else min(t, 2)
    for g in range(G):
        mode = rng.randrange(4)
        hi = min(S_cnt, 2 * t + 2)
        k = 0 if mode == 0 else (rng.randint(1, kmax) if (mode < 3 or n > 40) else (rng.randint(t + 1, hi) if hi > t else hi))
        for i in rng.sample(ids, min(k, S_cnt)):
            ev[i][g] = (ev[i][g] + rng.randrange(1, P)) % P
    eng.set_small_batch_chunks(0)                             # the lane-per-chunk kernels of the large batches and
    eng.set_second_chance(False)                              # OEC/Gao for every flagged chunk ...
    lane = eng.batch_recover(ids, to_arr(field, [ev[i] for i in ids]), n, d, t)
    # the matrix-core decode of the large batches (Fr: 2 <= d + 1 <= 15, Goldilocks: <= 16; other shapes fall through to the
    # lane kernels) and the matrix-core encode: same bytes
    eng.set_matrix_cores(True, 1)
    mf = eng.batch_recover(ids, to_arr(field, [ev[i] for i in ids]), n, d, t)
    rcm, ym = eng.vandermonde_apply(to_arr(field, polys), n, d)
    eng.set_matrix_cores(True, 65536)
    assert lane[0] == mf[0] and all(np.array_equal(u, v) for u, v in zip(lane[1:], mf[1:]))
    assert rcm == 0 and np.array_equal(ym, y0)
    eng.set_small_batch_chunks(8192)                          # ... and the defaults (wave-per-chunk kernel, second-
    eng.set_second_chance(True)                               # chance candidates before OEC/Gao): same bytes
    rc, co, nco, st = eng.batch_recover(ids, to_arr(field, [ev[i] for i in ids]), n, d, t)
    assert lane[0] == rc and all(np.array_equal(u, v) for u, v in zip(lane[1:], (co, nco, st)))
    if S_cnt < d + t + 1:                                     # robust_interpolate.rs:333-341: "Not enough evaluations"
        with pytest.raises(S.InvalidInput):
            S.batch_recover_secret([(i, ev[i][:1]) for i in ids], n, d, t)
        assert rc == 4
        return
    any_fail = False
    for g in range(G):
        shares = [S.Share(ev[i][g], i, d) for i in ids]
        try:
            want, _ = S.recover_secret(shares, n, t)
            got = [to_int(field, c) for c in (co[g][: nco[g]] if st[g] == 1 else co[g])]
            assert st[g] in (0, 1) and got == want + [0] * (len(got) - len(want)), (n, t, d, g)
        except S.ShareErr as e:
            any_fail = True
            assert st[g] == e.code, (n, t, d, g, st[g], e.code)
    assert (rc != 0) == any_fail


@pytest.mark.parametrize("field", ["fr", "goldilocks"])
def test_argument_validation_fuzz(engines, field):
    """Random mostly-INVALID arguments: the error code (and so the validation order) must be the oracle's, i.e. the
    reference's (robust_interpolate.rs:94-157 and :284-341, shamir.rs:199-239)."""
    eng, S = engines[field], (SFR if field == "fr" else SGL)
    P = S.R_MOD
    rng = random.Random(20240 if field == "fr" else 20241)
    seen = set()
    for _ in range(150):
        n = rng.randint(1, 12)
        t = rng.randint(0, 4)
        d = rng.randint(0, 5)
        S_cnt = rng.randint(0, n + 2)
        ids = [rng.randint(0, n + 1) if rng.random() < 0.15 else rng.randrange(max(n, 1)) for _ in range(S_cnt)]
        if rng.random() < 0.6 and S_cnt <= n:
            ids = rng.sample(range(n), S_cnt)                 # mostly distinct, in range
        degs = [d if rng.random() < 0.9 else d + 1 for _ in range(S_cnt)]
        poly = [rng.randrange(P) for _ in range(d + 1)]
        vals = [S.p_eval(poly, S.domain_element(max(n, 1), i % max(n, 1))) for i in ids]
        # --- RobustShare::recover_secret
        if S_cnt:
            try:
                S.recover_secret([S.Share(v, i, dg) for v, i, dg in zip(vals, ids, degs)], n, t)
                want = 0
            except S.ShareErr as e:
                want = e.code
            except (IndexError, ZeroDivisionError, ValueError):
                want = None                                    # the reference would panic: unspecified here
            if want is not None:
                rc = eng.recover_secret(ids, degs, to_arr(field, vals), n, t)[0]
                assert rc == want, ("recover_secret", n, t, d, ids, degs, rc, want)
                seen.add(("rs", want))
        # --- batch_recover_secret (two chunks)
        ev = [(i, [v, (v + 1) % P]) for i, v in zip(ids, vals)]
        try:
            S.batch_recover_secret(ev, n, d, t)
            want = 0
        except S.ShareErr as e:
            want = e.code
        except (IndexError, ZeroDivisionError, ValueError):
            want = None
        if want is not None and S_cnt:
            arr = to_arr(field, [[v, (v + 1) % P] for v in vals])
            rc = eng.batch_recover(ids, arr, n, d, t)[0]
            assert rc == want, ("batch_recover", n, t, d, ids, rc, want)
            seen.add(("br", want))
        # --- compute_shares
        try:
            S.compute_shares(poly, n, d)
            want = 0
        except S.ShareErr as e:
            want = e.code
        rc = eng.compute_shares(to_arr(field, [poly]), n, d)[0] if n > 0 else None
        if rc is not None:
            assert rc == want, ("compute_shares", n, d, rc, want)
    assert len(seen) >= 4, seen                                # several distinct outcomes were actually exercised
