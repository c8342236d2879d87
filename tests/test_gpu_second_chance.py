"""The second-chance kernel (k_second_chance: two interpolation candidates for chunks that fail the optimistic
verification, before OEC/Gao) must decide exactly like the reference's oec_decode (robust_interpolate.rs:579-628).
One batch whose chunks carry EVERY liar pattern up to t + 1 liars (n = 10), or a sample of them (n = 16, d > t),
with and without missing senders -- each chunk compared with the big-int oracle, in both modes (second chance on:
most chunks never reach OEC/Gao; off: all of them do)."""
import itertools
import random

import numpy as np
import pytest

from __graft_entry__ import load_package
from oracle import spec as SFR
from oracle.spec_gl import S as SGL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["fr", "goldilocks"])
def env(request):
    e = load_package().Engine(0, field=request.param)
    yield e, request.param, (SFR if request.param == "fr" else SGL)
    e.close()


def to_arr(field, rows):
    a = np.array(rows, dtype=object)
    if field == "goldilocks":
        return a.astype(np.uint64)
    out = np.zeros(a.shape + (4,), dtype=np.uint64)
    for idx in np.ndindex(a.shape):
        v = int(a[idx])
        out[idx] = [(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)]
    return out


def to_int(field, x):
    return int(x) if field == "goldilocks" else sum(int(x[k]) << (64 * k) for k in range(4))


@pytest.mark.parametrize("n,t,d,present,max_liars,sample", [
    (10, 3, 3, None, 4, None),                     # windows A = [0,4), B = [4,8); every pattern of <= t + 1 liars
    (10, 3, 3, [0, 1, 2, 4, 5, 6, 7, 9], 3, None), # two senders missing: S = 8, one OEC round only
    (10, 3, 2, None, 4, 300),                      # d < t
    (16, 5, 7, None, 5, 400),                      # d > t: windows A = [0,8), B = [8,16) fill the prefix exactly
    (16, 5, 10, None, 2, 60),                      # d = 2t: S = needed, no OEC round exists -> every flagged chunk fails
])
def test_every_liar_pattern(env, n, t, d, present, max_liars, sample):
    eng, field, S = env
    P = S.R_MOD
    rng = random.Random(n * 100 + t * 10 + d)
    ids = list(present) if present else list(range(n))
    rng.shuffle(ids)                                               # arrival order
    patterns = [c for k in range(max_liars + 1) for c in itertools.combinations(sorted(ids), k)]
    if sample and len(patterns) > sample:
        patterns = [()] + rng.sample(patterns[1:], sample - 1)
    G = len(patterns)
    polys = [[rng.randrange(P) for _ in range(d + 1)] for _ in range(G)]
    polys[1][d] = 0                                                # a lower-degree polynomial among them
    ev = {i: [S.p_eval(polys[g], S.domain_element(n, i)) for g in range(G)] for i in ids}
    for g, liars in enumerate(patterns):
        for i in liars:
            ev[i][g] = (ev[i][g] + rng.randrange(1, P)) % P
    arr = to_arr(field, [ev[i] for i in ids])
    results = []
    for second in (True, False):
        eng.set_second_chance(second)
        results.append(eng.batch_recover(ids, arr, n, d, t))
    # the same with the lane-per-chunk decode kernel, where the second-chance candidates are a kernel of their own
    # (the default at this size runs them inside the wave-per-chunk kernel)
    eng.set_second_chance(True)
    eng.set_small_batch_chunks(0)
    sep = eng.batch_recover(ids, arr, n, d, t)
    # ... and with the runtime-shaped kernels (Fr: k_second_chance instead of k_second_chance_m<M> for the long list)
    eng.set_force_generic(True)
    gen = eng.batch_recover(ids, arr, n, d, t)
    eng.set_force_generic(False)
    eng.set_small_batch_chunks(8192)
    (rc1, co1, nco1, st1), (rc0, co0, nco0, st0) = results
    assert sep[0] == rc1 and all(np.array_equal(u, v) for u, v in zip(sep[1:], (co1, nco1, st1)))
    assert gen[0] == rc1 and all(np.array_equal(u, v) for u, v in zip(gen[1:], (co1, nco1, st1)))
    assert rc1 == rc0 and np.array_equal(st1, st0) and np.array_equal(nco1, nco0)
    good = st1 <= 1
    assert np.array_equal(co1[good], co0[good])
    fails = 0
    for g in range(G):
        shares = [S.Share(ev[i][g], i, d) for i in ids]
        try:
            want, _ = S.recover_secret(shares, n, t)
            got = [to_int(field, c) for c in (co1[g][: nco1[g]] if st1[g] == 1 else co1[g])]
            assert st1[g] in (0, 1) and got == want + [0] * (len(got) - len(want)), (patterns[g], st1[g])
            assert (st1[g] == 0) == all(i not in patterns[g] for i in sorted(ids)[: d + t + 1]), patterns[g]
        except S.ShareErr as e:
            fails += 1
            assert st1[g] == e.code, (patterns[g], st1[g], e.code)
    assert (rc1 != 0) == (fails > 0)
    if d == 2 * t:
        assert fails == sum(1 for p in patterns if any(i in p for i in sorted(ids)[: d + t + 1]))
