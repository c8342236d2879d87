"""GPU parity tests of the Goldilocks variants: the hbmpc_gl_* entry points,
called through the C ABI, against oracle/spec_gl.py (the same restatement as for Fr with the field constants of
mpc/src/common/math/goldilocks.rs:4-13).  Bar: bit-exact.  The cases mirror tests/test_gpu_parity.py: seeded
random inputs at sizes the big-int oracle finishes in seconds, the reference's edge cases (corruption inside and
outside the verify window, missing senders, too many errors, erasures, degree checks), field edge values, and
size-independent properties at 2^20 chunks."""
import itertools
import random

import numpy as np
import pytest

from __graft_entry__ import load_package
from oracle.spec_gl import P, S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def _eng():
    e = load_package().Engine(0, field="goldilocks")
    yield e
    e.close()


@pytest.fixture(params=["fast", "lane", "generic", "mfma", "mfma-gao"])
def eng(request, _eng):
    """fast: the pruned-FFT / register-resident kernels instantiated for Goldilocks; generic: the runtime-shaped
    Horner / re-reading kernels (what shapes outside the templates fall back to); mfma: the matrix-core kernel
    (kernels_mfma_gl.hpp) at every size it covers (2 <= d + 1 <= 16), with and without the second-chance candidates"""
    mf = request.param.startswith("mfma")
    _eng.set_force_generic(request.param == "generic")
    _eng.set_small_batch_chunks(0 if request.param == "lane" or mf else 8192)  # lane: the large-batch kernels at every size
    _eng.set_second_chance(request.param in ("fast", "mfma"))  # the other modes: every flagged chunk goes to the OEC/Gao kernel
    _eng.set_matrix_cores(mf, 1 if mf else 0)
    yield _eng
    _eng.set_force_generic(False)
    _eng.set_small_batch_chunks(8192)
    _eng.set_second_chance(True)
    _eng.set_matrix_cores(True, 65536)


def rnd(seed, *shape):
    rng = random.Random(seed)
    return np.array([rng.randrange(P) for _ in range(int(np.prod(shape)))], dtype=np.uint64).reshape(shape)


EDGE = [0, 1, 2, P - 1, P - 2, (1 << 32) - 1, 1 << 32, (1 << 63), P >> 1, (P >> 1) + 1, 0xFFFFFFFF00000000, 7]


def test_field_ops(eng):
    a, b = rnd(1, 4000), rnd(2, 4000)
    k = len(EDGE)
    a[:k * k] = np.repeat(np.array(EDGE, dtype=np.uint64), k)
    b[:k * k] = np.tile(np.array(EDGE, dtype=np.uint64), k)
    ai, bi = [int(x) for x in a], [int(x) for x in b]
    for op, fn in (("add", lambda x, y: (x + y) % P), ("sub", lambda x, y: (x - y) % P), ("mul", lambda x, y: x * y % P)):
        rc, got = eng.fr_op(op, a, b)
        assert rc == 0
        assert [int(x) for x in got] == [fn(x, y) for x, y in zip(ai, bi)], op


def test_scalar_ops(eng):
    """Add<F>, Sub<F>, Mul<F>, from_scalar_sub (common/mod.rs:205-280) over the small field"""
    a = rnd(3, 900)
    a[:len(EDGE)] = np.array(EDGE, dtype=np.uint64)
    ai = [int(x) for x in a]
    for sv in EDGE + [int(rnd(4, 1)[0])]:
        sc = np.array([sv], dtype=np.uint64)
        for op, fn in (("add", lambda x: (x + sv) % P), ("sub", lambda x: (x - sv) % P), ("mul", lambda x: x * sv % P),
                       ("rsub", lambda x: (sv - x) % P)):
            rc, got = eng.fr_op_scalar(op, a, sc)
            assert rc == 0 and [int(x) for x in got] == [fn(x) for x in ai], (op, sv)
    assert eng.fr_op_scalar("add", a, np.array([P], dtype=np.uint64))[0] == 4   # not canonical


def test_context_serves_one_field(eng):
    fr = load_package().Engine(0)
    x = np.zeros((4, 3), dtype=np.uint64)
    assert fr.L.hbmpc_gl_compute_shares(fr.ctx, x.ctypes.data, 4, 6, 2, x.ctypes.data) == 5          # TypeMismatch
    assert eng.L.hbmpc_compute_shares(eng.ctx, x.ctypes.data, 4, 6, 2, x.ctypes.data) == 5
    assert eng.L.hbmpc_set_field_impl(eng.ctx, 1) == 5
    fr.close()


@pytest.mark.parametrize("n,d", [(4, 1), (6, 2), (7, 2), (10, 3), (16, 5), (16, 10), (16, 15), (31, 10), (33, 7), (100, 33),
                                 (255, 84)])
def test_compute_shares_and_vandermonde_vs_oracle(eng, n, d):
    G = 37 if n > 64 else 193
    x = rnd(n * 1000 + d, G, d + 1)
    x[0, :] = 0
    x[1, :] = P - 1
    rc, y = eng.compute_shares(x, n, d)
    assert rc == 0 and y.shape == (n, G)
    for g in (0, 1, 2, G // 2, G - 1):
        want = [s.v for s in S.compute_shares([int(c) for c in x[g]], n, d)]
        assert [int(v) for v in y[:, g]] == want
    rc, y2 = eng.vandermonde_apply(x, n, d)
    assert rc == 0 and np.array_equal(y, y2)
    rc, v = eng.make_vandermonde(n, d)
    assert rc == 0 and [[int(c) for c in row] for row in v] == S.make_vandermonde(n, d)
    # full check of every chunk through the Vandermonde rows (numpy object arithmetic)
    V = np.array(S.make_vandermonde(n, d), dtype=object)
    want_all = (V.dot(x.astype(object).T)) % P
    assert np.array_equal(y.astype(object), want_all)


def test_eval_errors_and_empty(eng):
    x = rnd(3, 5, 3)
    assert eng.compute_shares(x, 2, 2)[0] == 4          # n <= degree: InvalidInput (robust_interpolate.rs:59-64)
    rc, y = eng.compute_shares(np.zeros((0, 3), dtype=np.uint64), 6, 2)
    assert rc == 0 and y.shape == (6, 0)


def encode(eng, x, n, d):
    rc, y = eng.vandermonde_apply(x, n, d)
    assert rc == 0
    return y


@pytest.mark.parametrize("n,t,d", [(4, 1, 1), (7, 2, 2), (10, 3, 3), (16, 5, 5), (16, 5, 10), (31, 10, 10), (64, 21, 21)])
def test_batch_recover_vs_oracle(eng, n, t, d):
    G = 120
    x = rnd(n + 77 * d, G, d + 1)
    y = encode(eng, x, n, d)
    rng = random.Random(n * 31 + d)
    # chunks 0..39 clean; 40..79: <= t errors anywhere; 80..99: errors only outside the verify window;
    # 100..109: more than t errors; 110..119: one error in the base set
    for g in range(40, 80):
        for s in rng.sample(range(n), rng.randint(1, t)):
            y[s, g] = (int(y[s, g]) + rng.randrange(1, P)) % P
    needed = d + t + 1
    for g in range(80, 100):
        if needed < n:
            for s in rng.sample(range(needed, n), min(t, n - needed)):
                y[s, g] = (int(y[s, g]) + 1) % P
    for g in range(100, 110):
        for s in rng.sample(range(n), min(n, t + 1 + (n - d - 2 * t - 1))):
            y[s, g] = (int(y[s, g]) + rng.randrange(1, P)) % P
    for g in range(110, 120):
        y[rng.randrange(d + 1), g] ^= np.uint64(1)
    ids = list(range(n))
    rng.shuffle(ids)                                      # arrival order
    rc, co, nco, st = eng.batch_recover(ids, y[ids], n, d, t)
    ev = [(i, [int(v) for v in y[i]]) for i in ids]
    for g in range(G):
        shares = [S.Share(col[g], i, d) for i, col in ev]
        try:
            coeffs, _ = S.recover_secret(shares, n, t)
            ok = True
        except S.ShareErr as e:
            ok, code = False, e.code
        if ok:
            got = [int(v) for v in co[g][: nco[g]]] if st[g] == 1 else [int(v) for v in co[g]]
            assert st[g] in (0, 1), (g, st[g])
            want = coeffs + [0] * (len(got) - len(coeffs))
            assert got == want, g
        else:
            assert st[g] == code, (g, st[g], code)
    any_fail = any(st[g] not in (0, 1) for g in range(G))
    assert (rc != 0) == any_fail
    # P(0)-only variant on the decodable part
    good = [g for g in range(G) if st[g] in (0, 1)]
    rc, sec, st0 = eng.batch_recover_p0(ids, np.ascontiguousarray(y[ids][:, good]), n, d, t)
    assert rc == 0 and [int(v) for v in sec] == [int(co[g][0]) for g in good]


def test_batch_recover_validation_and_missing_senders(eng):
    n, t, d = 10, 3, 3
    x = rnd(5, 50, d + 1)
    y = encode(eng, x, n, d)
    ids = [9, 0, 3, 4, 6, 1, 8]                            # exactly d + t + 1 = 7 senders, unsorted
    rc, co, nco, st = eng.batch_recover(ids, y[ids], n, d, t)
    assert rc == 0 and np.array_equal(co, x) and not st.any()
    assert eng.batch_recover(ids[:6], y[ids[:6]], n, d, t)[0] == 4     # not enough evaluations
    assert eng.batch_recover([0, 0, 1, 2, 3, 4, 5], y[:7], n, d, t)[0] == 4  # duplicate id
    assert eng.batch_recover([0, 1, 2, 3, 4, 5, 10], y[:7], n, d, t)[0] == 4  # id out of range
    assert eng.batch_recover(list(range(10)), y, 9, d, t)[0] == 4      # n < 3t + 1


def test_recover_secret_all_corruption_combinations(eng):
    # robust_interpolate.rs test_robust_interpolate_all_corruption_combinations, in the small field
    n, t, d = 7, 2, 2
    coeffs = [11, 22, 33]
    base = S.compute_shares(coeffs, n, d)
    for k in range(0, t + 2):
        for bad in itertools.combinations(range(n), k):
            vals = np.array([(s.v + (97 + i if i in bad else 0)) % P for i, s in enumerate(base)], dtype=np.uint64)
            rc, co, sec = eng.recover_secret(list(range(n)), [d] * n, vals, n, t)
            try:
                want, wsec = S.recover_secret([S.Share(int(v), i, d) for i, v in enumerate(vals)], n, t)
                assert rc == 0 and [int(c) for c in co] == want and int(sec) == wsec, bad
            except S.ShareErr as e:
                assert rc == e.code, (bad, rc, e.code)


def test_recover_secret_trims_and_errors(eng):
    n, t = 7, 2
    sh = S.compute_shares([5, 0, 0], n, 2)                 # degree-2 sharing of a constant polynomial
    vals = np.array([s.v for s in sh], dtype=np.uint64)
    rc, co, sec = eng.recover_secret(list(range(n)), [2] * n, vals, n, t)
    assert rc == 0 and [int(c) for c in co] == [5] and int(sec) == 5
    assert eng.recover_secret(list(range(n)), [2] * 6 + [1], vals, n, t)[0] == 2      # DegreeMismatch
    assert eng.recover_secret([0, 1, 2, 3], [2] * 4, vals[:4], n, t)[0] == 4          # not enough shares


def test_gao_rs_decode(eng):
    n, k = 10, 4
    msg = [3, 1, 4, 1]
    cw = [S.p_eval(msg, S.domain_element(n, i)) for i in range(n)]
    for erasures, errors in (([], []), ([2, 7], [0]), ([1], [3, 9]), ([0, 1, 2, 3, 4, 5], []), ([], [0, 1, 2])):
        rec = list(cw)
        for i in errors:
            rec[i] = (rec[i] + 12345) % P
        rc, co = eng.gao_rs_decode(np.array(rec, dtype=np.uint64), k, n, erasures)
        try:
            want = S.gao_rs_decode(rec, k, n, erasures)
            assert rc == 0 and [int(c) for c in co] == want, (erasures, errors)
        except S.ShareErr as e:
            assert rc == e.code, (erasures, errors, rc)


def test_nonrobust_and_batch_interpolate(eng):
    n, t = 10, 3
    ids = [7, 2, 9, 0, 4, 5, 1]
    poly = [9, 8, 7, 6]
    vals = np.array([S.p_eval(poly, S.domain_element(n, i)) for i in ids], dtype=np.uint64)
    rc, co, sec = eng.nonrobust_recover_secret(ids, [6] * len(ids), vals, n)
    assert rc == 0 and [int(c) for c in co] == poly and int(sec) == 9
    assert eng.nonrobust_recover_secret(ids, [2] * len(ids), vals, n)[0] == 2          # degree 3 > 2
    # RanDouSha verifier shape: G columns through S = 2t + 1 points, degree() per column
    S_, G = 2 * t + 1, 64
    pol = rnd(8, G, S_)
    pol[::2, t + 1:] = 0                                    # even columns have degree <= t
    pol[4, :] = 0                                           # the zero polynomial (degree() = 0)
    idl = list(range(S_))
    ev = np.array([[S.p_eval([int(c) for c in pol[g]], S.domain_element(n, i)) for g in range(G)] for i in idl],
                  dtype=np.uint64)
    rc, co, deg = eng.batch_interpolate(idl, ev, n)
    assert rc == 0 and np.array_equal(co, pol)
    assert [int(x) for x in deg] == [S.p_degree(S.p_norm([int(c) for c in pol[g]])) for g in range(G)]


def test_elementwise_vs_oracle(eng):
    N = 500
    a, b, r2t, rt, c, x, y, d, e = (rnd(20 + i, N) for i in range(9))
    a[:len(EDGE)] = np.array(EDGE, dtype=np.uint64)
    b[:len(EDGE)] = np.array(EDGE[::-1], dtype=np.uint64)
    I = lambda v: [int(q) for q in v]  # noqa: E731
    rc, out = eng.triple_local(a, b, r2t)
    assert rc == 0 and I(out) == [(p * q - r) % P for p, q, r in zip(I(a), I(b), I(r2t))]
    rc, out = eng.triple_finalize(rt, c)
    assert rc == 0 and I(out) == [(p + q) % P for p, q in zip(I(rt), I(c))]
    rc, dsh, esh = eng.beaver_open_shares(a, b, x, y)
    assert rc == 0 and I(dsh) == [(p - q) % P for p, q in zip(I(a), I(x))] and I(esh) == [(p - q) % P for p, q in zip(I(b), I(y))]
    rc, z = eng.beaver_finalize(c, x, y, d, e)
    want = [S.beaver_finalize(S.Share(cc, 0, 1), S.Share(xx, 0, 1), S.Share(yy, 0, 1), dd, ee).v
            for cc, xx, yy, dd, ee in zip(I(c), I(x), I(y), I(d), I(e))]
    assert rc == 0 and I(z) == want


def test_full_size_roundtrip_and_linearity(eng):
    """2^20 chunks, n = 16, t = 5, d = 5 (the shape of BASELINE configs[1] in the small field): device-resident
    encode -> erase t senders -> decode round trip; linearity checksum; strided decode of a sub-range."""
    import torch
    dev = torch.device("cuda", 0)
    # an explicit torch stream whose handle goes through the C ABI: torch's work and the library's kernels are then
    # ordered on ONE stream (handle 0 would mean "the context's own stream", unordered with torch's default stream)
    ts = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(ts)
    n, t, d, G = 16, 5, 5, 1 << 20
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    hi = torch.randint(0, 0xFFFFFFFF, (G, d + 1), dtype=torch.int64, device=dev, generator=g)  # < 2^32 - 1: canonical
    lo = torch.randint(0, 1 << 32, (G, d + 1), dtype=torch.int64, device=dev, generator=g)
    x = (hi << 32) | lo
    y = torch.empty((n, G), dtype=torch.int64, device=dev)
    s = ts.cuda_stream
    assert s != 0
    assert eng.dev_vandermonde_apply(x.data_ptr(), G, n, d, y.data_ptr(), s) == 0
    keep = [15, 3, 8, 0, 12, 5, 9, 1, 14, 7, 2]            # d + t + 1 = 11 of the 16 senders, arrival order
    ysub = y[keep].contiguous()
    co = torch.empty((G, d + 1), dtype=torch.int64, device=dev)
    st = torch.empty((G,), dtype=torch.uint8, device=dev)
    summ = torch.zeros((4,), dtype=torch.int32, device=dev)
    assert eng.dev_batch_recover(keep, ysub.data_ptr(), G, n, d, t, co.data_ptr(), 0, st.data_ptr(), summ.data_ptr(), s) == 0
    torch.cuda.synchronize()
    assert bool((co == x).all()) and int(st.max()) == 0 and summ.tolist()[:2] == [0, 0]
    # sampled comparison with the oracle
    xs = x[:: G // 16].cpu().numpy().view(np.uint64)
    ys = y[:, :: G // 16].cpu().numpy().view(np.uint64)
    for i in range(xs.shape[0]):
        assert [int(v) for v in ys[:, i]] == [sh.v for sh in S.compute_shares([int(c) for c in xs[i]], n, d)]
    # linearity: encode(x) + encode(x') == encode(x + x') on the device
    x2 = torch.roll(x, 1, 0)
    y2, y3, xsum = torch.empty_like(y), torch.empty_like(y), torch.empty_like(x)
    assert eng.dev_vandermonde_apply(x2.data_ptr(), G, n, d, y2.data_ptr(), s) == 0
    assert eng.dev_fr_op("add", x.data_ptr(), x2.data_ptr(), G * (d + 1), xsum.data_ptr(), s) == 0
    assert eng.dev_vandermonde_apply(xsum.data_ptr(), G, n, d, y3.data_ptr(), s) == 0
    ysum = torch.empty_like(y)
    assert eng.dev_fr_op("add", y.data_ptr(), y2.data_ptr(), n * G, ysum.data_ptr(), s) == 0
    torch.cuda.synchronize()
    assert bool((ysum == y3).all())
    # corrupt t senders in every 64th chunk: OEC/Gao fallback on the device, P(0)-only, strided rows
    yc = y.clone()
    yc[:t, ::64] ^= 1
    sec = torch.empty((G,), dtype=torch.int64, device=dev)
    assert eng.dev_batch_recover_strided(list(range(n)), yc.data_ptr(), G, G, n, d, t, sec.data_ptr(), p0=True,
                                         status_d=st.data_ptr(), summary_d=summ.data_ptr(), stream=s) == 0
    torch.cuda.synchronize()
    assert bool((sec == x[:, 0]).all()) and summ.tolist()[:2] == [G // 64, 0]
    assert int(st[::64].min()) == 1 and int(st[1::64].max()) == 0
    torch.cuda.set_stream(torch.cuda.default_stream(dev))


def test_golden_vectors(eng):
    from tests import golden_util as GU
    assert GU.run_all(eng, field="goldilocks") > 90


@pytest.mark.parametrize("n,t,groups", [(4, 1, 9), (7, 2, 30), (16, 5, 120)])
def test_small_field_triple_gen_pipeline(eng, n, t, groups):
    """TripleGenNode over GoldilocksField (PreprocNodesSmallField, honeybadger/mod.rs:316-324) as a device-resident
    replay of all n parties: local a_i b_i - r2t_i, Vandermonde encode, the n EvalBatch decodes, the RevealBatch
    decode, rt_i + opened.  [c]_t must open to a b and every share must equal the reference algebra."""
    pkg = load_package()
    N = groups * (2 * t + 1)
    rng = random.Random(1000 * n + t)
    a, b, r = ([rng.randrange(P) for _ in range(N)] for _ in range(3))

    def share_all(secrets, d):  # [n][N] degree-d sharings through the oracle
        rows = [[s.v for s in S.compute_shares([x] + [rng.randrange(P) for _ in range(d)], n, d)] for x in secrets]
        return np.array(rows, dtype=np.uint64).T.copy()

    sa, sb, srt, sr2t = share_all(a, t), share_all(b, t), share_all(r, t), share_all(r, 2 * t)
    tg = pkg.pipelines.TripleGen(eng, n, t, N)
    tg.upload(sa, sb, sr2t, srt)
    tg.run()
    c = tg.download_c()
    tg.close()
    opened = [(x * y - z) % P for x, y, z in zip(a, b, r)]
    for p in range(n):
        assert [int(v) for v in c[p]] == [(int(srt[p, i]) + opened[i]) % P for i in range(N)]
    rc, p0, st = eng.batch_recover_p0(list(range(n)), c, n, t, t)
    assert rc == 0 and [int(v) for v in p0] == [x * y % P for x, y in zip(a, b)]


@pytest.mark.parametrize("n,t,groups,tamper", [(4, 1, 7, False), (7, 2, 30, True), (10, 3, 21, False), (13, 4, 9, True), (16, 5, 120, True)])
def test_small_field_triple_gen_one_launch_equals_four(eng, n, t, groups, tamper):
    """hbmpc_gl_dev_triplegen_parties as ONE launch (a workgroup per chunk, csrc/kernels_triplegen_wg.hpp over Goldilocks) against
    the four separate launches (hbmpc_set_fused_triplegen(ctx, 0)): both arms' messages, the opened values, the output shares,
    statuses and summaries byte for byte, honest or with replaced shares."""
    import ctypes as C
    pkg = load_package()
    N = groups * (2 * t + 1)
    rng = np.random.default_rng(77 * n + t)

    def share_all(secrets, d):  # [n][N] degree-d sharings by the library's own compute_shares (checked against the oracle elsewhere)
        co = rng.integers(0, P, (N, d + 1), dtype=np.uint64)
        co[:, 0] = secrets
        rc, out = eng.compute_shares(co, n, d)
        assert rc == 0
        return out

    a, b, r = (rng.integers(0, P, N, dtype=np.uint64) for _ in range(3))
    ins = [share_all(a, t), share_all(b, t), share_all(r, 2 * t), share_all(r, t)]
    if tamper:
        ins[0][1, 3] = ins[0][2, 3]
        ins[2][0, N - 1] = ins[2][1, N - 1]
    res = {}
    try:
        for form, fused in (("one", 1 << 20), ("four", 0)):
            assert eng.L.hbmpc_set_fused_triplegen(eng.ctx, C.c_size_t(fused)) == 0
            tg = pkg.pipelines.TripleGen(eng, n, t, N)
            tg.upload(*ins)
            tg.run(check=False)
            got = {"c": tg.download_c().copy(), "Y": tg.download_named("Y", (n, n, groups)).copy(), "Z": tg.download_named("Z", (n, groups)).copy(),
                   "opened": tg.download_named("opened", (N,)).copy()}
            for nm, arr in (("status", np.zeros(n * groups, dtype=np.uint8)), ("summary", np.zeros(4, dtype=np.uint32)), ("summary_first", np.zeros(4, dtype=np.uint32))):
                eng.d2h(arr, tg.buffer(nm)[0])
                got[nm] = arr
            eng.sync()
            res[form] = got
            tg.close()
    finally:
        eng.L.hbmpc_set_fused_triplegen(eng.ctx, C.c_size_t(1024))
    for nm in res["four"]:
        assert np.array_equal(res["one"][nm], res["four"][nm]), nm
    if tamper:
        assert res["one"]["summary_first"][0] > 0
    else:
        assert res["one"]["summary"].tolist() == [0, 0, 0xffffffff, 0]
        opened = [(int(x) * int(y) - int(z)) % P for x, y, z in zip(a, b, r)]
        assert [int(v) for v in res["one"]["opened"]] == opened


def test_in_place_wire_path_small_field(_eng):
    """ark's Vec<Fp64> payloads (u64-LE length + 8-byte LE elements) written by the encode kernel and decoded where
    they arrived; a sender with a non-canonical element is dropped, another one lies in one chunk."""
    import torch
    eng = _eng
    n, t, d, G = 10, 3, 3, 501
    x = rnd(31, G, d + 1)
    stride = 8 * (G + 3)                                          # any multiple of 8 >= 8 + 8 G
    dev = torch.device("cuda:0")
    wire = torch.zeros(n * stride // 8, dtype=torch.int64, device=dev)
    xd = torch.from_numpy(x.view(np.int64)).to(dev)
    assert eng.dev_encode_fvec(xd.data_ptr(), G, n, d, wire.data_ptr(), stride) == 0
    torch.cuda.synchronize()
    raw = wire.cpu().numpy().view(np.uint64).reshape(n, stride // 8)
    rc, y = eng.vandermonde_apply(x, n, d)
    assert rc == 0
    for j in range(n):
        assert raw[j, 0] == G and np.array_equal(raw[j, 1:1 + G], y[j])      # Vec<Fp64>::serialize_compressed
    raw = raw.copy()
    raw[2, 1 + 7] ^= np.uint64(1)                                # sender 2 lies in chunk 7
    raw[6, 1 + 11] = np.uint64(P)                                # sender 6: an element == p is not deserialisable
    wire.copy_(torch.from_numpy(raw.view(np.int64).reshape(-1)))
    st = torch.zeros(n, dtype=torch.int32, device=dev)
    assert eng.dev_validate_fvec(wire.data_ptr(), stride, 8 + 8 * G, G, n, st.data_ptr()) == 0
    torch.cuda.synchronize()
    assert st.cpu().tolist() == [4 if j == 6 else 0 for j in range(n)]
    order = [j for j in range(n) if j != 6][::-1]
    co = torch.zeros((G, d + 1), dtype=torch.int64, device=dev)
    stat = torch.zeros(G, dtype=torch.uint8, device=dev)
    assert eng.dev_batch_recover_slots(order, order, wire.data_ptr() + 8, stride // 8, G, n, d, t, co.data_ptr(),
                                       status_d=stat.data_ptr()) == 0
    torch.cuda.synchronize()
    assert np.array_equal(co.cpu().numpy().view(np.uint64), x)
    assert int(stat[7]) == 1 and int(stat.sum()) == 1


@pytest.mark.parametrize("n,t,G,parties", [(4, 1, 77, 4), (7, 2, 130, 7), (10, 3, 65, 3), (13, 4, 200, 2), (16, 5, 333, 16), (20, 6, 40, 3)])
def test_small_field_triple_encode_fused_equals_two_launches(_eng, n, t, G, parties):
    """hbmpc_gl_dev_triple_encode_parties: the local product fused into the encode (k_eval_fft1_triple<Gold>; (20, 6) has no
    fused kernel and takes the workspace) against triple_local + vandermonde_apply_parties, and party 0 against big ints"""
    import torch
    eng = _eng
    d = 2 * t
    N = G * (d + 1)
    dev = torch.device("cuda", 0)
    a, b, r = (torch.from_numpy(rnd(70 + k, parties * N).view(np.int64)).to(dev) for k in range(3))
    tmp = torch.empty((parties * N,), dtype=torch.int64, device=dev)
    y1 = torch.full((parties, n, G), -1, dtype=torch.int64, device=dev)
    y2 = torch.full((parties, n, G), -1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    assert eng.dev_triple_encode_parties(a.data_ptr(), b.data_ptr(), r.data_ptr(), G, n, d, parties, tmp.data_ptr(), y1.data_ptr()) == 0
    assert eng.dev_elem("triple_local", [a.data_ptr(), b.data_ptr(), r.data_ptr(), tmp.data_ptr()], parties * N) == 0
    assert eng.dev_vandermonde_apply_parties(tmp.data_ptr(), G, n, d, parties, y2.data_ptr()) == 0
    eng.sync()
    assert torch.equal(y1, y2)
    ah, bh, rh = ([int(v) for v in t_[:N].cpu().numpy().view(np.uint64)] for t_ in (a, b, r))
    got = y1[0].cpu().numpy().view(np.uint64)
    for g in (0, 1, G // 2, G - 1):
        x = [(ah[g * (d + 1) + k] * bh[g * (d + 1) + k] - rh[g * (d + 1) + k]) % P for k in range(d + 1)]
        assert [int(got[j, g]) for j in range(n)] == [s.v for s in S.compute_shares(x, n, d)]
    if n <= 16:   # the fused kernel needs no workspace
        y3 = torch.full_like(y1, -1)
        assert eng.dev_triple_encode_parties(a.data_ptr(), b.data_ptr(), r.data_ptr(), G, n, d, parties, 0, y3.data_ptr()) == 0
        eng.sync()
        assert torch.equal(y3, y1)


@pytest.mark.parametrize("n,K,row0,rows", [(16, 1500, 10, 6), (16, 37, 0, 6), (7, 3001, 4, 3), (13, 200, 0, 5), (4, 5000, 2, 2), (3, 100, 1, 2), (16, 1500, 0, 15),
                                           (16, 700, 15, 1), (9, 65, 3, 4), (20, 300, 12, 8)])
def test_small_field_mixing_step_in_one_launch(_eng, n, K, row0, rows):
    """The producers' n x n mixing step over Goldilocks: hbmpc_gl_dev_vandermonde_apply_rows_split and _lists from the kernel that reads the
    dealt rows in place and writes the parties' lists and the party-major rows itself (k_eval_fft1_mix, domains of up to 16 points)
    against the separate passes (hbmpc_set_producer_fusion(0): transpose, encode, transposes / copies) and against the plain y[row][G],
    itself checked against the oracle; n = 20 (a 32-point domain) has no such kernel and takes the passes either way"""
    import ctypes as C
    import torch
    eng = _eng
    dev = torch.device("cuda", 0)
    G, d = n * K, n - 1
    x = rnd(1000 * n + K, G, d + 1)
    x[0] = 0
    x[1] = P - 1
    xr = torch.as_tensor(np.ascontiguousarray(x.T).view(np.int64), device=dev)          # rows: [d + 1][G]
    tmp = torch.empty((G, d + 1), dtype=torch.int64, device=dev)
    y = torch.full((n, G), -1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    assert eng.dev_vandermonde_apply_rows(xr.data_ptr(), G, G, n, d, tmp.data_ptr(), y.data_ptr()) == 0, eng.last_error()
    eng.sync()
    y_np = y.cpu().numpy().view(np.uint64)
    V = np.array(S.make_vandermonde(n, d), dtype=object)                                   # oracle on the first and the last chunks
    for sl in (slice(0, 40), slice(G - 40, G)):
        assert np.array_equal(y_np[:, sl].astype(object), V.dot(x[sl].astype(object).T) % P)
    other_rows = [r for r in range(n) if not row0 <= r < row0 + rows]
    want_others = y_np.reshape(n, n, K)[other_rows].transpose(1, 0, 2)                      # [party][r'][K]
    want_lists = y_np.reshape(n, n, K)[row0:row0 + rows].transpose(1, 2, 0)                  # [party][K][row]
    k1 = K // 3
    assert eng.apply_rows_lists_in_kernel(G, n, d) == (3 <= n <= 16)
    try:
        for fused in (1, 0):
            assert eng.L.hbmpc_set_producer_fusion(eng.ctx, C.c_int(fused)) == 0
            for with_others in (True, False):
                la = torch.full((n, k1, rows), -1, dtype=torch.int64, device=dev)
                lb = torch.full((n, K - k1 - 1 + 2, rows), -1, dtype=torch.int64, device=dev)
                others = torch.full((n, n - rows, K), -1, dtype=torch.int64, device=dev)
                y.fill_(-1)
                torch.cuda.synchronize()
                slices = [(la.data_ptr(), k1 * rows, 0, k1), (lb.data_ptr(), (K - k1 + 1) * rows, k1 + 1, K - k1 - 1)]
                rc = eng.dev_vandermonde_apply_rows_split(xr.data_ptr(), G, G, n, d, tmp.data_ptr(), y.data_ptr(), row0, rows, K, slices,
                                                          others.data_ptr() if with_others else None)
                assert rc == 0, eng.last_error()
                eng.sync()
                if with_others:
                    assert np.array_equal(others.cpu().numpy().view(np.uint64), want_others), fused
                else:
                    assert np.array_equal(y.cpu().numpy().view(np.uint64)[other_rows], y_np[other_rows]), fused
                assert np.array_equal(la.cpu().numpy().view(np.uint64), want_lists[:, :k1]), fused
                got_b = lb.cpu().numpy().view(np.uint64)
                assert np.array_equal(got_b[:, :K - k1 - 1], want_lists[:, k1 + 1:]) and np.all(got_b[:, K - k1 - 1:] == np.uint64(2**64 - 1)), fused
    finally:
        eng.L.hbmpc_set_producer_fusion(eng.ctx, C.c_int(1))


@pytest.mark.parametrize("n,t,d,G", [(16, 5, 5, 1 << 15), (16, 5, 10, 20011), (31, 10, 10, 1 << 14), (64, 21, 21, 4099)])
def test_against_the_c_restatement_at_size(_eng, n, t, d, G):
    """hbmpc_gl_* against oracle/hbmpc_oracle.c built for Goldilocks (oracle/cref_gl.py) -- the second, independent
    restatement of the small field -- at batch sizes the Python big-int oracle does not reach: every chunk of an encode
    and of a decode with corrupted, missing and failing chunks, status bytes and trimmed lengths included."""
    from oracle import cref_gl as CG
    x = CG.fill_random(0xC0FFEE02 + n + d, G * (d + 1)).reshape(G, d + 1)
    rc, y = _eng.vandermonde_apply(x, n, d)
    rc0, y0 = CG.vandermonde_apply(x, n, d)
    assert rc == rc0 == 0 and np.array_equal(y, y0)
    rc, sh = _eng.compute_shares(x, n, d)
    assert rc == 0 and np.array_equal(sh, CG.compute_shares(x, n, d)[1])
    rng = np.random.default_rng(n * 1000 + d)
    ids = list(rng.permutation(n)[: max(d + t + 1, n - 1)])           # arrival order, one sender missing when possible
    ev = np.ascontiguousarray(y[ids])
    S_ = len(ids)
    bad = rng.choice(G, G // 8, replace=False)
    for j, g in enumerate(bad):
        k = 1 + j % (t + 1)                                           # 1 .. t + 1 corrupted senders: the last kind cannot decode
        for s in rng.choice(S_, min(k, S_), replace=False):
            ev[s, g] = (int(ev[s, g]) + 1 + j) % P
    rc, co, nco, st = _eng.batch_recover([int(i) for i in ids], ev, n, d, t)
    rc0, co0, nco0, st0 = CG.batch_recover([int(i) for i in ids], ev, n, d, t)
    assert rc == rc0 and np.array_equal(st, st0) and np.array_equal(nco, nco0) and np.array_equal(co, co0)
    assert (st == 0).sum() >= G - len(bad)
    if S_ > d + t + 1:   # with exactly d + t + 1 senders there is no OEC round: a corrupted chunk can only fail
        assert (st == 1).any()
    else:
        assert (st > 1).sum() == len(bad)
