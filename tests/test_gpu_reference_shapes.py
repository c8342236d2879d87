"""The path-touching assertions of the reference's INTEGRATION tests, restated at their literal (n, t) and batch shapes
against the HIP path.  The reference holds no stored bytes for this path, so these remain
property checks -- the oracle stays "parity unpinned" -- but they are the reference's own properties at the reference's
own shapes, with every arithmetic step a call through the C ABI:

  mpc/tests/batchrecon_test.rs:31-118    manual BatchRecon flow, n = 4, t = 1, secrets [3, 4]: recovered[..] == secrets
  mpc/tests/batchrecon_test.rs:126-210   node flow, n = 4, t = 1, secrets [3, 6]
  mpc/tests/triple_gen_test.rs:20-73     n = 13, t = 2, 5 triples; :78-80 n = 15, t = 3: a b == ab, a and b recovered
  mpc/tests/mul_test.rs:23-47,217-224    n = 10, t = 3, 10 / 8 / 3 multiplications: shares [0 ..= 2t] recover x y
  mpc/benches/hmpc_mul_micro_bench.rs:37 (n, t) in {(5, 1), (10, 3), (20, 6)}: recover_secret on honest shares and on
                                         shares with t corrupted senders (the OEC/Gao path), batch_recover_secret
"""
import numpy as np
import pytest

from __graft_entry__ import load_package
from oracle import cref as O
from oracle import spec as S

pytestmark = pytest.mark.gpu
R = S.R_MOD


@pytest.fixture(scope="module")
def eng():
    e = load_package().Engine(0)
    yield e
    e.close()


def share(eng, secrets, n, d, seed):
    """[n][len(secrets)] degree-d sharings through the library's compute_shares (random higher coefficients)"""
    N = len(secrets)
    co = O.fill_random(seed, N * (d + 1)).reshape(N, d + 1, 4)
    co[:, 0] = O.ints_to_u256(secrets)
    rc, sh = eng.compute_shares(co, n, d)
    assert rc == 0
    return sh


def recover(eng, ids, vals, n, d, t):
    rc, co, sec = eng.recover_secret(ids, [d] * len(ids), np.ascontiguousarray(vals), n, t)
    assert rc == 0
    return O.u256_to_ints(co) if len(co) else [], int(O.u256_to_ints(sec.reshape(1, 4))[0])


@pytest.mark.parametrize("secrets", [[3, 4], [3, 6]])
def test_batchrecon_flow_n4_t1(eng, secrets):
    n, t = 4, 1
    assert len(secrets) == t + 1
    sh = share(eng, secrets, n, t, 7)                       # generate_independent_shares: sh[i][k] = party i's share of secret k
    # Step 1: party i encodes its t + 1 shares with the Vandermonde matrix and sends y_j to party j (:45-62)
    ys = []
    for i in range(n):
        rc, y = eng.vandermonde_apply(np.ascontiguousarray(sh[i]).reshape(1, t + 1, 4), n, t)
        assert rc == 0
        ys.append(y[:, 0])                                   # [recipient j]
    # Steps 2-5: party j interpolates from the first 2t + 1 senders and reveals y_j (:64-88)
    reveals = []
    for j in range(n):
        senders = list(range(2 * t + 1))
        _, val = recover(eng, senders, np.stack([ys[i][j] for i in senders]), n, t, t)
        reveals.append(val)
    # Step 6: everyone reconstructs the coefficients from the first 2t + 1 revealed y_j (:91-118)
    ids = list(range(2 * t + 1))
    poly, _ = recover(eng, ids, O.ints_to_u256([reveals[j] for j in ids]), n, t, t)
    poly = (poly + [0] * (t + 1))[: t + 1]
    assert poly == secrets


@pytest.mark.parametrize("n,t,n_shares", [(13, 2, 5), (15, 3, 5)])
def test_triple_gen_shapes(eng, n, t, n_shares):
    pkg = load_package()
    m = 2 * t + 1
    N = -(-n_shares // m) * m                                # the pipeline works on whole chunks of 2t + 1
    rng = np.random.default_rng(n)
    a, b, r = O.fill_random(1, N), O.fill_random(2, N), O.fill_random(3, N)
    ai, bi = O.u256_to_ints(a), O.u256_to_ints(b)
    sa, sb = share(eng, ai, n, t, 11), share(eng, bi, n, t, 12)
    srt, sr2t = share(eng, O.u256_to_ints(r), n, t, 13), share(eng, O.u256_to_ints(r), n, 2 * t, 14)
    tg = pkg.pipelines.TripleGen(eng, n, t, N)
    tg.upload(sa, sb, sr2t, srt)
    tg.run()
    c = tg.download_c()
    tg.close()
    ids = list(range(n))
    for i in range(n_shares):                                # triple_gen_test.rs:66-73
        _, av = recover(eng, ids, sa[:, i], n, t, t)
        _, bv = recover(eng, ids, sb[:, i], n, t, t)
        _, abv = recover(eng, ids, c[:, i], n, t, t)
        assert av * bv % R == abv and av == ai[i] and bv == bi[i]
    del rng


@pytest.mark.parametrize("no_of_mul", [10, 8, 3])
def test_mul_shapes_n10_t3(eng, no_of_mul):
    n, t, N = 10, 3, no_of_mul
    rng = np.random.default_rng(no_of_mul)
    xs = [int(v) for v in rng.integers(1, 1 << 62, N)]
    ys = [int(v) for v in rng.integers(1, 1 << 62, N)]
    ta, tb = O.fill_random(21, N), O.fill_random(22, N)
    tc = O.fr_binop("mul", ta, tb)
    sx, sy = share(eng, xs, n, t, 31), share(eng, ys, n, t, 32)
    sta, stb, stc = (share(eng, O.u256_to_ints(v), n, t, 33 + k) for k, v in enumerate((ta, tb, tc)))
    ids = list(range(n))
    # Multiply::init: d = a - x, e = b - y opened; finalize_mul (multiplication.rs:417-426, 57-100)
    dsh, esh = [], []
    for p in range(n):
        rc, d_, e_ = eng.beaver_open_shares(sta[p], stb[p], sx[p], sy[p])
        assert rc == 0
        dsh.append(d_), esh.append(e_)
    rc, dop, st = eng.batch_recover_p0(ids, np.stack(dsh), n, t, t)
    assert rc == 0 and not st.any()
    rc, eop, st = eng.batch_recover_p0(ids, np.stack(esh), n, t, t)
    assert rc == 0 and not st.any()
    z = []
    for p in range(n):
        rc, zp = eng.beaver_finalize(stc[p], sx[p], sy[p], dop, eop)
        assert rc == 0
        z.append(zp)
    z = np.stack(z)
    for i in range(N):                                       # mul_test.rs:217-224: shares [0 ..= 2t] of multiplication i
        sel = list(range(2 * t + 1))
        _, zr = recover(eng, sel, z[sel, i], n, t, t)
        assert zr == xs[i] * ys[i] % R


@pytest.mark.parametrize("n,t", [(5, 1), (10, 3), (20, 6)])
def test_micro_bench_shapes(eng, n, t):
    """hmpc_mul_micro_bench.rs: recover_secret on n honest shares, on shares with t corrupted senders (forces OEC/Gao),
    batch_recover_secret over a batch; make_vandermonde + apply_vandermonde at (n, t)"""
    secret = 0x1234567890ABCDEF123456789
    sh = share(eng, [secret], n, t, 99)[:, 0]
    ids = list(range(n))
    poly, val = recover(eng, ids, sh, n, t, t)
    assert val == secret and len(poly) == t + 1
    bad = sh.copy()
    for i in range(t):
        bad[i, 0] ^= np.uint64(1 + i)
    want = O.recover_secret(ids, [t] * n, bad, n, t)
    rc, co, sec = eng.recover_secret(ids, [t] * n, bad, n, t)
    assert rc == want[0] == 0 and np.array_equal(co, want[1]) and np.array_equal(sec, want[2])
    assert int(O.u256_to_ints(sec.reshape(1, 4))[0]) == secret
    G = 64
    x = O.fill_random(5, G * (t + 1)).reshape(G, t + 1, 4)
    rc, y = eng.vandermonde_apply(x, n, t)
    rc0, y0 = O.vandermonde_apply(x, n, t)
    assert rc == rc0 == 0 and np.array_equal(y, y0)
    rc, v = eng.make_vandermonde(n, t)
    rc0, v0 = O.make_vandermonde(n, t)
    assert rc == rc0 == 0 and np.array_equal(v, v0)
    rc, co, nco, st = eng.batch_recover(ids, y, n, t, t)
    assert rc == 0 and np.array_equal(co, x) and not st.any()
