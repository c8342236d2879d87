"""GPU parity tests: the HIP path, called through the C ABI, against the oracle (oracle/ is the
checker only) -- golden vectors, seeded random inputs at sizes the oracle finishes in seconds,
the reference's edge cases, and size-independent properties at BASELINE.json's full sizes.
Bar: bit-exact (all arithmetic is integer mod r)."""
import numpy as np
import pytest

from __graft_entry__ import load_package
from oracle import cref as O
from oracle import spec as S
from tests import golden_util as GU

pytestmark = pytest.mark.gpu
R = S.R_MOD


@pytest.fixture(scope="module")
def eng():
    pkg = load_package()
    e = pkg.Engine(0)
    yield e
    e.close()


@pytest.fixture(params=["u29", "u29-lane", "u29-mfma", "u29-mfma-gao", "u29-generic", "sat32", "sat32-lane"])
def eng_all(request, eng):
    """u29 / sat32: the defaults (small batches decode with the wave-per-chunk kernel); -lane: the lane-per-chunk
    kernels at every size; -mfma: the matrix-core decode the large batches use, at every size (the second: every flagged
    chunk to OEC/Gao); -generic: the runtime-shaped kernels"""
    mode = request.param
    eng.set_impl("sat32" if mode.startswith("sat32") else "u29")
    eng.set_force_generic(mode == "u29-generic")
    eng.set_small_batch_chunks(0 if mode.endswith("-lane") or "-mfma" in mode else 8192)
    eng.set_matrix_cores("-mfma" in mode, 1 if "-mfma" in mode else 65536)
    # the defaults resolve most flagged chunks in the second-chance kernel; the other modes send every flagged chunk
    # to the OEC/Gao kernel, so both fallbacks see every corruption case of this file
    eng.set_second_chance(mode in ("u29", "sat32", "u29-mfma"))
    yield eng
    eng.set_matrix_cores(True, 65536)
    eng.set_impl("u29")
    eng.set_force_generic(False)
    eng.set_small_batch_chunks(8192)
    eng.set_second_chance(True)


def rnd(seed, *shape):
    n = int(np.prod(shape))
    return O.fill_random(seed, n).reshape(*shape, 4)


def test_native_library_loaded(eng):
    import os
    maps = open("/proc/self/maps").read()
    assert "libhbmpc_hip.so" in maps and os.path.exists("/dev/kfd")


def test_golden_vectors(eng_all):
    assert GU.run_all(eng_all) > 100


def test_field_ops(eng_all):
    e = eng_all
    a, b = rnd(1, 3000), rnd(2, 3000)
    edge = O.ints_to_u256([0, 1, 2, R - 1, R - 2, (1 << 255) % R, (1 << 254), R >> 1, (R >> 1) + 1, 0xFFFFFFFF,
                           (1 << 232) - 1, (1 << 29) - 1, 1 << 29, R - (1 << 29)])
    k = len(edge)
    a[:k * k] = np.repeat(edge, k, axis=0)
    b[:k * k] = np.tile(edge, (k, 1))
    for op in ("add", "sub", "mul"):
        rc, got = e.fr_op(op, a, b)
        assert rc == 0
        assert GU.eq(got, O.fr_binop(op, a, b)), op


def test_scalar_ops(eng_all):
    """share (+, -, *) one field element and element - share (common/mod.rs:205-280; oracle/spec.py:219-232): the
    scalar-broadcast entry points against the big-int restatement, edge values on both sides"""
    e = eng_all
    edge = [0, 1, 2, R - 1, R - 2, (1 << 255) % R, 1 << 254, R >> 1, 0xFFFFFFFF, (1 << 232) - 1, (1 << 29) - 1, R - (1 << 29)]
    a = rnd(5, 700)
    a[:len(edge)] = O.ints_to_u256(edge)
    av = O.u256_to_ints(a)
    for sv in edge + [int(O.u256_to_ints(rnd(6, 1))[0])]:
        sc = O.ints_to_u256([sv])
        for op, f in (("add", S.share_add_scalar), ("sub", S.share_sub_scalar), ("mul", S.share_mul_scalar)):
            rc, got = e.fr_op_scalar(op, a, sc)
            assert rc == 0 and O.u256_to_ints(got) == [f(S.Share(v, 3, 2), sv).v % R for v in av], (op, sv)
        rc, got = e.fr_op_scalar("rsub", a, sc)
        assert rc == 0 and O.u256_to_ints(got) == [S.share_from_scalar_sub(sv, S.Share(v, 3, 2)).v % R for v in av], sv
    rc, _ = e.fr_op_scalar("add", a, O.ints_to_u256([R]))      # not canonical
    assert rc == 4


# (5, 1), (10, 3), (20, 6): the (n, t) of the reference's own benches (mpc/benches/hmpc_mul_micro_bench.rs:37)
EVAL_SHAPES = [(5, 1), (1, 0), (2, 0), (2, 1), (3, 0), (3, 2), (4, 1), (4, 3), (5, 1), (7, 2), (8, 7), (10, 3), (13, 4), (16, 0),
               (16, 5), (16, 10), (16, 15), (9, 8), (20, 6), (31, 10), (31, 20), (31, 30), (32, 31), (33, 5), (64, 21),
               (100, 31), (100, 33), (128, 15), (255, 31), (255, 84), (200, 3)]


@pytest.mark.parametrize("n,d", EVAL_SHAPES)
def test_compute_shares_vs_oracle(eng, n, d):
    G = 333 if n <= 64 else 70  # ragged tiles on purpose
    x = rnd(1000 + 7 * n + d, G, d + 1)
    x[0] = 0
    x[1] = O.ints_to_u256([R - 1] * (d + 1))
    # constant polynomials whose shares sit in the rare branch of the canonical store (top limb == r's)
    x[2] = O.ints_to_u256([R - 1] + [0] * d)
    x[3] = O.ints_to_u256([(0x73EDA7 << 232) + 5] + [0] * d)
    x[4] = O.ints_to_u256([(0x73EDA7 << 232) - 1] + [0] * d)
    rc, want = O.compute_shares(x, n, d)
    assert rc == 0
    rc, got = eng.compute_shares(x, n, d)
    assert rc == 0, eng.last_error()
    assert GU.eq(got, want)
    rc, got2 = eng.vandermonde_apply(x, n, d)
    assert rc == 0 and GU.eq(got2, want)


@pytest.mark.parametrize("n,d", [(16, 5), (31, 10), (100, 33), (7, 2)])
def test_compute_shares_other_impls(eng_all, n, d):
    x = rnd(5 + n, 130, d + 1)
    rc, want = O.compute_shares(x, n, d)
    rc, got = eng_all.compute_shares(x, n, d)
    assert rc == 0 and GU.eq(got, want)


def test_eval_errors_and_empty(eng):
    x = rnd(1, 2, 6)
    assert eng.compute_shares(x, 5, 5)[0] == 4      # n <= degree
    assert eng.compute_shares(x, 6, 5)[0] == 0
    rc, out = eng.compute_shares(x[:0], 6, 5)        # empty batch
    assert rc == 0 and out.shape == (6, 0, 4)
    rc, v = eng.make_vandermonde(4, 9)               # make_vandermonde alone allows d >= n
    rc0, v0 = O.make_vandermonde(4, 9)
    assert rc == rc0 == 0 and GU.eq(v, v0)


def _corrupt_case(n, t, d, G, seed, frac_bad, max_bad, ids=None):
    """valid codewords from the oracle, then corrupt `frac_bad` of the chunks in 1..max_bad senders"""
    x = rnd(seed, G, d + 1)
    rc, y = O.compute_shares(x, n, d)
    assert rc == 0
    ids = list(range(n)) if ids is None else ids
    ev = np.ascontiguousarray(y[ids])
    rng = np.random.default_rng(seed)
    bad_chunks = rng.choice(G, size=max(1, int(G * frac_bad)), replace=False) if frac_bad > 0 else []
    for c in bad_chunks:
        k = int(rng.integers(1, max_bad + 1))
        for pos in rng.choice(len(ids), size=k, replace=False):
            ev[pos, c, 0] ^= np.uint64(1 + int(rng.integers(0, 1000)))
            if O.u256_to_ints(ev[pos, c]) >= R:  # keep canonical
                ev[pos, c, 3] = 0
    return x, ids, ev


@pytest.mark.parametrize("n,t,d,G,frac,max_bad", [
    (4, 1, 1, 500, 0.05, 1), (7, 2, 2, 700, 0.05, 3), (10, 3, 3, 600, 0.1, 4), (16, 5, 5, 2000, 0.02, 6),
    (16, 5, 10, 1500, 0.02, 2), (31, 10, 10, 1500, 0.02, 11), (13, 4, 4, 300, 0.3, 5), (31, 10, 20, 200, 0.0, 0),
    (64, 21, 21, 150, 0.05, 10), (100, 33, 33, 40, 0.1, 5), (70, 3, 30, 60, 0.1, 3)])
def test_batch_recover_vs_oracle(eng, n, t, d, G, frac, max_bad):
    x, ids, ev = _corrupt_case(n, t, d, G, 77 + n + d, frac, max_bad)
    perm = np.random.default_rng(n).permutation(len(ids))  # arrival order
    ids = [ids[i] for i in perm]
    ev = np.ascontiguousarray(ev[perm])
    rc0, co0, nco0, st0 = O.batch_recover(ids, ev, n, d, t)
    rc, co, nco, st = eng.batch_recover(ids, ev, n, d, t)
    assert rc == rc0, (rc, rc0, eng.last_error())
    assert np.array_equal(st, st0)
    assert GU.eq(co, co0) and np.array_equal(nco, nco0)
    rc1, p0, st1 = eng.batch_recover_p0(ids, ev, n, d, t)
    assert rc1 == rc0 and np.array_equal(st1, st0) and GU.eq(p0, co0[:, 0])
    if frac > 0:
        assert (st0 != 0).any()
    ok = st0 <= 1
    assert GU.eq(co[ok], x[ok])  # recovered coefficients == the original secrets


@pytest.mark.parametrize("n,t,d", [(10, 3, 3), (16, 5, 10), (31, 10, 10), (5, 1, 1), (20, 6, 6)])
def test_batch_recover_other_impls(eng_all, n, t, d):
    x, ids, ev = _corrupt_case(n, t, d, 300, 5 + n, 0.05, t)
    rc0, co0, nco0, st0 = O.batch_recover(ids, ev, n, d, t)
    rc, co, nco, st = eng_all.batch_recover(ids, ev, n, d, t)
    assert rc == rc0 and GU.eq(co, co0) and np.array_equal(nco, nco0) and np.array_equal(st, st0)


def test_batch_recover_missing_senders_and_validation(eng):
    n, t, d = 13, 4, 4
    ids = [12, 0, 5, 7, 3, 9, 1, 10, 2, 11]  # 10 of 13 senders, arrival order
    x, _, ev = _corrupt_case(n, t, d, 200, 9, 0.1, 2, ids=ids)
    rc0, co0, nco0, st0 = O.batch_recover(ids, ev, n, d, t)
    rc, co, nco, st = eng.batch_recover(ids, ev, n, d, t)
    assert rc == rc0 and GU.eq(co, co0) and np.array_equal(st, st0) and np.array_equal(nco, nco0)
    # validation order of robust_interpolate.rs:290-341
    assert eng.batch_recover(ids, ev, 12, d, t)[0] == 4           # n < 3t+1
    assert eng.batch_recover([], ev[:0], n, d, t)[0] == 4         # no evaluations
    assert eng.batch_recover(ids, ev[:, :0], n, d, t)[0] == 4     # empty batch
    assert eng.batch_recover([0, 0] + ids[2:], ev, n, d, t)[0] == 4   # duplicate
    assert eng.batch_recover([13] + ids[1:], ev, n, d, t)[0] == 4     # out of range
    assert eng.batch_recover(ids[:8], ev[:8], n, d, t)[0] == 4        # fewer than d+t+1


def test_recover_secret_all_corruption_combinations(eng):
    # robust_interpolate.rs:827-876 (n = 7, t = 2, every subset of <= t corrupted shares) + t+1 failures
    from itertools import combinations
    n, t = 7, 2
    co = rnd(3, 1, t + 1)
    co[0, 0] = O.ints_to_u256(42)
    rc, sh = O.compute_shares(co, n, t)
    vals = sh[:, 0]
    for k in range(0, t + 2):
        for idx in combinations(range(n), k):
            v = vals.copy()
            for i in idx:
                v[i, 0] ^= np.uint64(999)
            want = O.recover_secret(list(range(n)), [t] * n, v, n, t)
            got = eng.recover_secret(list(range(n)), [t] * n, v, n, t)
            assert got[0] == want[0], (idx, got[0], want[0])
            if want[0] == 0:
                assert GU.eq(got[1], want[1]) and GU.eq(got[2], want[2])
                if k <= t:
                    assert O.u256_to_ints(got[2]) == 42


def test_elementwise_vs_oracle(eng_all):
    e = eng_all
    N = 1537
    a, b, c, d, x = (rnd(s, N) for s in range(40, 45))
    assert GU.eq(e.triple_local(a, b, c)[1], O.triple_local(a, b, c)[1])
    assert GU.eq(e.triple_finalize(a, b)[1], O.triple_finalize(a, b)[1])
    g, w = e.beaver_open_shares(a, b, c, d)[1:], O.beaver_open_shares(a, b, c, d)[1:]
    assert GU.eq(g[0], w[0]) and GU.eq(g[1], w[1])
    assert GU.eq(e.beaver_finalize(a, b, c, d, x)[1], O.beaver_finalize(a, b, c, d, x)[1])
    for m in (1, 4, 16, 29, 40):
        bits = rnd(50 + m, m, N)
        assert GU.eq(e.truncpr_rdash(bits, m)[1], O.truncpr_rdash(bits, m)[1]), m
    for k, m in ((16, 4), (32, 16), (1, 0), (250, 255), (64, 100)):
        assert GU.eq(e.truncpr_open_share(a, b, c, k, m)[1], O.truncpr_open_share(a, b, c, k, m)[1])
    for m in (0, 1, 7, 8, 9, 31, 32, 33, 64, 100, 254, 255, 256, 264):
        assert GU.eq(e.truncpr_finalize(a, b, c, m)[1], O.truncpr_finalize(a, b, c, m)[1]), m
    assert e.truncpr_finalize(a, b, c, 257)[0] == 4
    assert e.truncpr_open_share(a, b, c, 0, 4)[0] == 4


def test_beaver_mul_config1(eng):
    """BASELINE config 1: n=4, t=1, 5 Beaver multiplications (tests/node_test.rs:447-453) as the
    open/finalize algebra of all 4 parties, every step on the device."""
    n, t, K = 4, 1, 5
    rng = S.SplitMix64(12)
    xs, ys, as_, bs = ([rng.fr() for _ in range(K)] for _ in range(4))

    def share(vals):  # [n][K] shares of K secrets
        co = O.ints_to_u256([[v, rng.fr()] for v in vals])
        rc, sh = eng.compute_shares(co, n, t)
        assert rc == 0
        return sh
    sx, sy, sa, sb = share(xs), share(ys), share(as_), share(bs)
    sc = share([a * b % R for a, b in zip(as_, bs)])
    dsh = np.stack([eng.beaver_open_shares(sa[i], sb[i], sx[i], sy[i])[1] for i in range(n)])
    esh = np.stack([eng.beaver_open_shares(sa[i], sb[i], sx[i], sy[i])[2] for i in range(n)])
    rc, d_open, _ = eng.batch_recover_p0(list(range(n)), dsh, n, t, t)
    rc2, e_open, _ = eng.batch_recover_p0(list(range(n)), esh, n, t, t)
    assert rc == rc2 == 0
    assert O.u256_to_ints(d_open) == [(a - x) % R for a, x in zip(as_, xs)]
    z = np.stack([eng.beaver_finalize(sc[i], sx[i], sy[i], d_open, e_open)[1] for i in range(n)])
    rc, prod, _ = eng.batch_recover_p0(list(range(n)), z, n, t, t)
    assert rc == 0 and O.u256_to_ints(prod) == [x * y % R for x, y in zip(xs, ys)]


# ---- BASELINE.json full sizes: size-independent properties (the oracle would take minutes) -------
def _dev_roundtrip(eng, n, t, d, G, seed, p0):
    """encode on the device -> erase (drop senders) -> decode on the device -> equals the input"""
    x = rnd(seed, G, d + 1)
    xd = eng.dev_alloc(x.nbytes)
    yd = eng.dev_alloc(n * G * 32)
    eng.h2d(xd, x)
    assert eng.dev_compute_shares(xd, G, n, d, yd) == 0, eng.last_error()
    keep = list(range(n))[::-1][: d + t + 1 + (n - (d + t + 1)) // 2]  # erasures: drop the lowest ids, reversed arrival
    ow = 1 if p0 else d + 1
    od = eng.dev_alloc(G * ow * 32)
    sd = eng.dev_alloc(G)
    smd = eng.dev_alloc(16)
    # gather the kept rows into a compact [S][G] array on the host side of the API (D2H/H2D: a test)
    y = O.u256((n, G))
    eng.d2h(y, yd)
    eng.sync()
    ev = np.ascontiguousarray(y[keep])
    evd = eng.dev_alloc(ev.nbytes)
    eng.h2d(evd, ev)
    assert eng.dev_batch_recover(keep, evd, G, n, d, t, od, 0, sd, smd, p0=p0) == 0, eng.last_error()
    out = O.u256((G, ow)) if not p0 else O.u256((G,))
    st = np.zeros(G, dtype=np.uint8)
    summ = np.zeros(4, dtype=np.uint32)
    eng.d2h(out, od)
    eng.d2h(st, sd)
    eng.d2h(summ, smd)
    eng.sync()
    for p in (xd, yd, od, sd, smd, evd):
        eng.dev_free(p)
    assert summ[0] == 0 and summ[1] == 0 and not st.any()
    assert GU.eq(out, x if not p0 else x[:, 0])
    return x, y


def test_full_size_cfg2_roundtrip_and_linearity(eng):
    # config 2: n=16, t=5, 2^20 secrets
    n, t, d, G = 16, 5, 5, 1 << 20
    x, y = _dev_roundtrip(eng, n, t, d, G, 0xC0FFEE01, p0=True)
    # checksum of checksums: sum over all secrets of share j == share j of the summed polynomial
    ints = lambda a: O.u256_to_ints(a)  # noqa: E731
    sub = slice(0, 4096)
    xs = [sum(col) % R for col in zip(*[ints(row) for row in x[sub]])]
    rc, ysum = eng.compute_shares(O.ints_to_u256([xs]), n, d)
    assert rc == 0
    for j in range(n):
        assert sum(ints(y[j, sub])) % R == ints(ysum[j, 0])
    # a sample of the full-size output against the oracle
    rc, want = O.compute_shares(x[-2000:], n, d)
    assert GU.eq(y[:, -2000:], want)


def test_full_size_cfg3_roundtrip(eng):
    # config 3: n=31, t=10, d=10, 2^20 chunks (encode = apply_vandermonde, decode = batch_recover_secret)
    n, t, d, G = 31, 10, 10, 1 << 20
    x, y = _dev_roundtrip(eng, n, t, d, G, 0xC0FFEE02, p0=False)
    rc, want = O.vandermonde_apply(x[:1500], n, d)
    assert GU.eq(y[:, :1500], want)


def test_full_size_cfg4_triple_algebra(eng):
    # config 4 shape (n=16, t=5, d=2t=10) on 2^18 triples of ONE simulated party + the open of a*b-r
    n, t, G = 16, 5, 1 << 18
    a, b, r2t = rnd(1, G), rnd(2, G), rnd(3, G)
    rc, loc = eng.triple_local(a, b, r2t)
    assert rc == 0
    idx = np.random.default_rng(1).choice(G, 3000, replace=False)
    assert GU.eq(loc[idx], O.triple_local(a[idx], b[idx], r2t[idx])[1])
    rc, c = eng.triple_finalize(r2t, loc)
    assert GU.eq(c[idx], O.fr_binop("mul", a[idx], b[idx]))  # (ab - r) + r == ab


@pytest.mark.parametrize("n,t,S", [(7, 2, 5), (16, 5, 16), (16, 5, 11), (31, 10, 21)])
def test_batch_interpolate_randousha_verifier(eng, n, t, S):
    """NonRobustShare::recover_secret per column (shamir.rs:199-239) as the RanDouSha verifier uses it
    (ran_dou_sha/mod.rs:569-602): plain Lagrange through all S shares, degree of the result."""
    G = 97
    deg_t, deg_2t = t, 2 * t
    ids = [int(i) for i in np.random.default_rng(S).permutation(n)[:S]]
    for true_deg in (deg_t, min(deg_2t, S - 1), 0):
        co = rnd(300 + true_deg, G, true_deg + 1)
        co[3] = 0                                   # the zero polynomial: degree() == 0
        co[4, true_deg] = 0                         # a lower-degree column
        rc, sh = O.compute_shares(co, n, true_deg)
        ev = np.ascontiguousarray(sh[ids])
        rc, got, deg = eng.batch_interpolate(ids, ev, n)
        assert rc == 0, eng.last_error()
        for g in (0, 1, 3, 4, 50, G - 1):
            rc0, want, sec = O.nonrobust_recover_secret(ids, [S - 1] * S, ev[:, g], n)
            assert rc0 == 0
            pad = np.zeros((S, 4), dtype=np.uint64)
            pad[: len(want)] = want
            assert GU.eq(got[g], pad), (true_deg, g)
            assert deg[g] == max(len(want) - 1, 0)
        assert deg[0] == true_deg and deg[3] == 0
        assert GU.eq(got[:, : true_deg + 1], co)


MAXLIMB = (0x73EDA6 << 232) | ((1 << 232) - 1)   # canonical, every 29-bit limb below the top one is all ones


@pytest.mark.parametrize("n,t,d", [(16, 5, 5), (16, 5, 10), (16, 5, 15), (31, 10, 10), (31, 10, 20), (64, 21, 40), (10, 3, 3)])
def test_accumulator_headroom_adversarial(eng_all, n, t, d):
    """Worst-case limbs for the lazy 64-bit column accumulators: constant polynomials whose value has all-ones
    limbs make every evaluation equal to that value, so all terms of a dot product add constructively
    (a wrong fold interval only shows on such inputs, never on random data)."""
    e = eng_all
    G = 130
    x = rnd(900 + n + d, G, d + 1)
    x[: G // 2] = 0
    x[: G // 2, 0] = O.ints_to_u256(MAXLIMB)             # constant polynomials
    x[G // 2: G // 2 + 10] = O.ints_to_u256(MAXLIMB)     # every coefficient max-limb
    x[G // 2 + 10: G // 2 + 20] = O.ints_to_u256(R - 1)
    rc, y = O.compute_shares(x, n, d)
    rc, got = e.compute_shares(x, n, d)
    assert rc == 0 and GU.eq(got, y)
    if n >= 3 * t + 1 and n >= d + t + 1:
        ids = list(range(n))
        rc0, co0, nco0, st0 = O.batch_recover(ids, y, n, d, t)
        rc, co, nco, st = e.batch_recover(ids, y, n, d, t)
        assert rc == rc0 == 0 and GU.eq(co, co0) and GU.eq(co, x) and not st.any()
        y[3, :, 0] ^= np.uint64(1)                       # one corrupted sender everywhere: all chunks via OEC/Gao
        rc0, co0, nco0, st0 = O.batch_recover(ids, y, n, d, t)
        rc, co, nco, st = e.batch_recover(ids, y, n, d, t)
        assert rc == rc0 and GU.eq(co, co0) and np.array_equal(st, st0) and np.array_equal(nco, nco0)
    S = min(n, d + 6)
    rc, ci, deg = e.batch_interpolate(list(range(S)), np.ascontiguousarray(y[:S, : G // 2]), n) if False else (0, None, None)
    bits = np.repeat(O.ints_to_u256([MAXLIMB, R - 1, 1, 0])[None, :, :], 40, axis=0)   # [m=40][N=4]
    assert GU.eq(e.truncpr_rdash(bits, 40)[1], O.truncpr_rdash(bits, 40)[1])


def test_interpolate_headroom_adversarial(eng):
    n, S, G = 31, 21, 64
    x = np.zeros((G, 1, 4), dtype=np.uint64)
    x[:, 0] = O.ints_to_u256(MAXLIMB)
    x[1::2, 0] = rnd(5, G // 2)
    rc, y = O.compute_shares(x, n, 0)
    ids = list(range(S))
    rc, co, deg = eng.batch_interpolate(ids, np.ascontiguousarray(y[:S]), n)
    assert rc == 0 and GU.eq(co[:, 0], x[:, 0]) and not co[:, 1:].any() and not deg.any()


def test_one_context_many_threads(eng):
    """include/hbmpc_hip.h: 'thread-safe and re-entrant: calls on one ctx from several threads serialise on the
    ctx's stream'.  Four threads hammer ONE context with encode + corrupted decode (every chunk takes the
    flag -> OEC/Gao path, whose two kernels share per-stream scratch) and must all get exact results."""
    import threading
    n, t, d, G = 10, 3, 3, 64
    errs = []

    def worker(seed):
        try:
            x = rnd(1000 + seed, G, d + 1)
            for it in range(25):
                rc, y = eng.vandermonde_apply(x, n, d)
                assert rc == 0
                y[seed % n, :, 0] ^= np.uint64(1)            # one corrupted sender: every chunk falls back
                rc, co, nco, st = eng.batch_recover(list(range(n)), y, n, d, t)
                assert rc == 0 and np.array_equal(co, x) and (st == 1).all() and (nco == d + 1).all(), (seed, it)
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for q in th:
        q.start()
    for q in th:
        q.join()
    assert not errs, errs[:2]


def test_threads_race_on_new_sender_sets(eng):
    """Two threads decode the SAME sender set, never seen before, at the same moment -- the OEC/Gao table and its
    layout (round count, offsets) must reach the cache as one unit: a thread that finds the table cached with a
    default-constructed layout would run zero OEC rounds and report every flagged chunk as DecodingError.  Second chance
    off, so that every corrupted chunk goes through the OEC/Gao kernel that reads that layout."""
    import threading
    n, t, d, G = 13, 4, 3, 48
    eng.set_second_chance(False)
    try:
        for trial in range(24):
            rng = np.random.default_rng(9000 + trial)
            S = int(rng.integers(d + t + 2, n + 1))
            ids = [int(i) for i in rng.permutation(n)[:S]]            # a fresh subset / order almost every trial
            x = rnd(2000 + trial, G, d + 1)
            rc, y = O.vandermonde_apply(x, n, d)
            ev = np.ascontiguousarray(y[ids])
            ev[ids.index(min(ids)), :, 0] ^= np.uint64(1)             # the lowest id (inside the interpolation set) lies in every chunk
            want = O.batch_recover(ids, ev, n, d, t)
            assert want[0] == 0 and (want[3] == 1).all()
            barrier = threading.Barrier(2)
            res, errs = [None, None], []

            def worker(k):
                try:
                    barrier.wait()
                    res[k] = eng.batch_recover(ids, ev, n, d, t)
                except Exception as e:  # noqa: BLE001
                    errs.append(repr(e))

            th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
            for q in th:
                q.start()
            for q in th:
                q.join()
            assert not errs, errs
            for got in res:
                assert got[0] == want[0] and all(np.array_equal(u, v) for u, v in zip(got[1:], want[1:])), trial
    finally:
        eng.set_second_chance(True)


def test_buffers_beyond_4_gib(eng):
    """10.5 M secrets, n = 16: the share buffer is 5.4 GB, so every byte offset past 2^32 is exercised (64-bit
    indexing in the staging, the party-major stores and the decode's row addressing).  Checked by the
    encode -> erase -> decode round trip over the whole batch plus oracle comparison of chunks sampled from both
    ends of the buffers."""
    import torch
    dev = torch.device("cuda", 0)
    ts = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(ts)
    s = ts.cuda_stream
    n, t, d, G = 16, 5, 5, (1 << 23) + (1 << 21) + 37      # ragged last tile as well
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    lo = torch.randint(0, 1 << 62, (G, d + 1, 3), dtype=torch.int64, device=dev, generator=g)
    hi = torch.randint(0, 0x73EDA753299D7D48, (G, d + 1, 1), dtype=torch.int64, device=dev, generator=g)
    x = torch.cat([lo, hi], dim=-1).contiguous()
    del lo, hi
    y = torch.empty((n, G, 4), dtype=torch.int64, device=dev)
    assert y.numel() * 8 > (1 << 32)
    assert eng.dev_compute_shares(x.data_ptr(), G, n, d, y.data_ptr(), s) == 0
    keep = [15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 0]           # d + t + 1 senders, the last rows of the buffer first
    ysub = y[keep].contiguous()
    co = torch.empty((G, d + 1, 4), dtype=torch.int64, device=dev)
    st = torch.empty((G,), dtype=torch.uint8, device=dev)
    assert eng.dev_batch_recover(keep, ysub.data_ptr(), G, n, d, t, co.data_ptr(), 0, st.data_ptr(), 0, s) == 0
    torch.cuda.synchronize()
    assert bool((co == x).all()) and int(st.max()) == 0
    idx = [0, 1, G // 2, G - 2, G - 1]
    xs = x[idx].cpu().numpy().view(np.uint64)
    ys = y[:, idx].cpu().numpy().view(np.uint64)
    rc, want = O.compute_shares(xs, n, d)
    assert rc == 0 and np.array_equal(ys, want)
    torch.cuda.set_stream(torch.cuda.default_stream(dev))


def test_scrub_staging_between_calls(eng):
    """hbmpc_scrub_staging zeroes the pooled staging memory of the host-pointer calls (device buffers of a large call,
    the pinned block of a small one); calls before and after it give the same bytes"""
    for B in (3, 70000):
        x = O.fill_random(31 + B, B * 6).reshape(B, 6, 4)
        rc, want = O.compute_shares(x, 16, 5)
        rc1, y1 = eng.compute_shares(x, 16, 5)
        eng.scrub_staging()
        rc2, y2 = eng.compute_shares(x, 16, 5)
        eng.scrub_staging()
        assert rc == rc1 == rc2 == 0 and np.array_equal(y1, want) and np.array_equal(y2, want)
        got = eng.batch_recover(list(range(16)), y2, 16, 5, 5)
        assert got[0] == 0 and np.array_equal(got[1], x)
