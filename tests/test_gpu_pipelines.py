"""Device-resident replays of the reference pipelines (all n parties on one GPU) against the algebra:
BASELINE config 4 (triple_gen) and config 5 (fpmul) shapes.  Every arithmetic step is an hbmpc_dev_*
call; the oracle only generates inputs and checks outputs."""
import numpy as np
import pytest

from __graft_entry__ import load_package
from oracle import cref as O
from oracle import spec as S
from tests import golden_util as GU

pytestmark = pytest.mark.gpu
R = S.R_MOD


@pytest.fixture(scope="module")
def pkg_eng():
    pkg = load_package()
    e = pkg.Engine(0)
    yield pkg, e
    e.close()


def share_all(secrets_u256, n, d, seed):
    """[n][N] degree-d sharings of N secrets (random higher coefficients), via the oracle"""
    N = secrets_u256.shape[0]
    co = O.fill_random(seed, N * (d + 1)).reshape(N, d + 1, 4)
    co[:, 0] = secrets_u256
    rc, sh = O.compute_shares(co, n, d)
    assert rc == 0
    return sh


def open_all(shares, n, d, t):
    rc, p0, st = O.batch_recover_p0(list(range(n)), shares, n, d, t)
    assert rc == 0
    return p0


@pytest.mark.parametrize("n,t,groups", [(4, 1, 11), (7, 2, 40), (16, 5, 300)])
def test_triple_gen_pipeline(pkg_eng, n, t, groups):
    pkg, eng = pkg_eng
    N = groups * (2 * t + 1)
    a, b, r = O.fill_random(1, N), O.fill_random(2, N), O.fill_random(3, N)
    sa, sb = share_all(a, n, t, 11), share_all(b, n, t, 12)
    srt, sr2t = share_all(r, n, t, 13), share_all(r, n, 2 * t, 14)
    tg = pkg.pipelines.TripleGen(eng, n, t, N)
    tg.upload(sa, sb, sr2t, srt)
    tg.run()
    c = tg.download_c()
    tg.close()
    # [c]_t opens to a*b, and every party's share equals the reference algebra rt_i + (ab - r)
    ab = O.fr_binop("mul", a, b)
    assert GU.eq(open_all(c, n, t, t), ab)
    opened = O.fr_binop("sub", ab, r)
    for p in range(n):
        assert GU.eq(c[p], O.triple_finalize(srt[p], opened)[1])


def _triplegen_forms(pkg, eng, n, t, groups, seed, tamper):
    """TripleGen through hbmpc_dev_triplegen_parties as one launch and as four: every visible buffer of both"""
    import ctypes as C
    torch = pytest.importorskip("torch")
    st = torch.cuda.Stream(device=torch.device("cuda", 0)).cuda_stream
    N = groups * (2 * t + 1)
    a, b, r = O.fill_random(seed, N), O.fill_random(seed + 1, N), O.fill_random(seed + 2, N)
    ins = {"a": share_all(a, n, t, seed + 3), "b": share_all(b, n, t, seed + 4), "r2t": share_all(r, n, 2 * t, seed + 5), "rt": share_all(r, n, t, seed + 6)}
    for nm, p, i in tamper:
        ins[nm][p, i] = ins[nm][(p + 1) % n, i]
    res = {}
    try:
        for form, fused in (("one", 1 << 20), ("four", 0)):
            assert eng.L.hbmpc_set_fused_triplegen(eng.ctx, C.c_size_t(fused)) == 0
            tg = pkg.pipelines.TripleGen(eng, n, t, N, stream=st)
            tg.upload(ins["a"], ins["b"], ins["r2t"], ins["rt"])
            tg.run(check=False)
            got = {"c": tg.download_c().copy(), "Y": tg.download_named("Y", (n, n, groups)).copy(), "Z": tg.download_named("Z", (n, groups)).copy(),
                   "opened": tg.download_named("opened", (N,)).copy()}
            for nm, arr in (("status", np.zeros(n * groups, dtype=np.uint8)), ("summary", np.zeros(4, dtype=np.uint32)), ("summary_first", np.zeros(4, dtype=np.uint32))):
                eng.d2h(arr, tg.buffer(nm)[0], st)
                got[nm] = arr
            eng.sync(st)
            tg.capture()
            eng.h2d(tg.c, np.zeros((n, N, 4), dtype=np.uint64), st)
            tg.replay()
            assert GU.eq(tg.download_c(), got["c"])
            res[form] = got
            tg.close()
    finally:
        eng.L.hbmpc_set_fused_triplegen(eng.ctx, C.c_size_t(1024))
    for nm in res["four"]:
        assert np.array_equal(res["one"][nm], res["four"][nm]), (n, t, groups, nm)
    return res["one"], (a, b, r)


@pytest.mark.parametrize("n,t,groups", [(4, 1, 5), (7, 2, 40), (10, 3, 33), (13, 4, 17), (16, 5, 100)])
def test_triplegen_one_launch_equals_four(pkg_eng, n, t, groups):
    """TripleGenNode for all parties in ONE launch (hbmpc_dev_triplegen_parties at a small batch: a workgroup per chunk,
    csrc/kernels_triplegen_wg.hpp) against the four separate launches (hbmpc_set_fused_triplegen(ctx, 0)): the messages of both
    arms, the opened values, the output shares, statuses and summaries byte for byte -- honest, then with shares replaced (the
    recipients' decodes of a chunk fail and count, the steps after them run on zero) -- eager and replayed as a graph."""
    pkg, eng = pkg_eng
    got, (a, b, r) = _triplegen_forms(pkg, eng, n, t, groups, 300 + n, [])
    assert got["summary"].tolist() == [0, 0, 0xffffffff, 0] and got["summary_first"].tolist() == [0, 0, 0xffffffff, 0]
    assert GU.eq(open_all(got["c"], n, t, t), O.fr_binop("mul", a, b))
    m = 2 * t + 1
    bad, _ = _triplegen_forms(pkg, eng, n, t, groups, 300 + n, [("a", 1, 2 * m + 1), ("r2t", 0, 0)])
    assert bad["summary_first"][0] > 0 and bad["status"][groups:].any()


@pytest.mark.parametrize("seed", range(8))
def test_triplegen_forms_random(pkg_eng, seed):
    """the same on random (t, number of chunks, a few replaced shares or none); tools/soak_triplegen.py runs many more seeds"""
    pkg, eng = pkg_eng
    rng = np.random.default_rng(0x7E + seed)
    t = int(rng.integers(1, 6))
    n, groups = 3 * t + 1, int(rng.integers(1, 130))
    N = groups * (2 * t + 1)
    tamper = [(str(rng.choice(["a", "b", "r2t", "rt"])), int(rng.integers(0, n)), int(rng.integers(0, N))) for _ in range(int(rng.integers(0, 3)))]
    _triplegen_forms(pkg, eng, n, t, groups, 9000 + 10 * seed, tamper)


@pytest.mark.parametrize("senders", ["2t+1", "n"])   # the reference opens from the first 2t+1 arrivals; n: OEC rounds available
@pytest.mark.parametrize("n,t,N,k,m", [(4, 1, 50, 16, 4), (7, 2, 64, 16, 4), (16, 5, 200, 32, 16)])
def test_fpmul_pipeline(pkg_eng, n, t, N, k, m, senders):
    pkg, eng = pkg_eng
    rng = np.random.default_rng(n)
    half = (k - 2) // 2
    xs = [int(v) for v in rng.integers(0, 1 << half, N)]
    ys = [int(v) for v in rng.integers(0, 1 << half, N)]
    x, y = O.ints_to_u256(xs), O.ints_to_u256(ys)
    ta, tb = O.fill_random(21, N), O.fill_random(22, N)
    tc = O.fr_binop("mul", ta, tb)
    bits = rng.integers(0, 2, (m, N))
    rints = [int(v) for v in rng.integers(0, 1 << 40, N)]
    sx, sy = share_all(x, n, t, 31), share_all(y, n, t, 32)
    sta, stb, stc = share_all(ta, n, t, 33), share_all(tb, n, t, 34), share_all(tc, n, t, 35)
    srint = share_all(O.ints_to_u256(rints), n, t, 36)
    sbits = np.stack([share_all(O.ints_to_u256([int(v) for v in bits[j]]), n, t, 40 + j) for j in range(m)], axis=1)  # [n][m][N]
    fp = pkg.pipelines.FpMul(eng, n, t, N, k, m, open_senders=None if senders == "2t+1" else n)
    fp.upload(sx, sy, sta, stb, stc, np.ascontiguousarray(sbits), srint)
    fp.run()
    z = fp.download("z")
    out = fp.download("out")
    fp.close()
    prod = [a * b for a, b in zip(xs, ys)]
    assert O.u256_to_ints(open_all(z, n, t, t)) == prod                       # Beaver product
    got = O.u256_to_ints(open_all(out, n, t, t))
    for v, p in zip(got, prod):
        assert v in (p >> m, (p >> m) + 1)                                    # probabilistic truncation
    # exact identity of truncpr.rs:215-220 per party against the oracle's element-wise restatement
    rdash_val = [sum(int(bits[j][i]) << j for j in range(m)) for i in range(N)]
    c_open = O.ints_to_u256([(prod[i] + (1 << (k - 1)) + (rints[i] << m) + rdash_val[i]) % R for i in range(N)])
    for p in range(n):
        rd = O.truncpr_rdash(np.ascontiguousarray(sbits[p]), m)[1]
        assert GU.eq(out[p], O.truncpr_finalize(z[p], rd, c_open, m)[1])


@pytest.mark.parametrize("n,t,N,k,m", [(16, 5, 1000, 16, 4), (16, 5, 1024, 16, 4), (7, 2, 333, 32, 16), (7, 2, 96, 16, 2), (4, 1, 2048, 16, 0), (16, 5, 3008, 16, 4), (31, 10, 256, 24, 8),
                                       (10, 3, 32, 16, 4), (40, 13, 130, 16, 4), (64, 21, 66, 16, 1)])
def test_fpmul_one_launch_equals_five(pkg_eng, n, t, N, k, m):
    """FPMulNode for all parties in ONE launch (hbmpc_dev_fpmul_parties at a small batch: a wave per element,
    csrc/kernels_fpmul_wave.hpp) and in FOUR (large batches: the first open forms its senders' shares as it loads them,
    hbmpc_set_fpmul_pair_decode) against the five separate launches (hbmpc_set_fused_fpmul(ctx, 0)): every buffer a caller can
    see -- the opened a - x | b - y, z, r', the shares TruncPr opens, the opened value, the output, the statuses -- byte for
    byte, on sharings of random field elements (the steps are algebra, not range-limited) of which one is inconsistent (its
    opens fail and count, the steps after them run on zero), eager and replayed as a graph."""
    import ctypes as C
    pkg, eng = pkg_eng
    torch = pytest.importorskip("torch")
    ts = torch.cuda.Stream(device=torch.device("cuda", 0))
    names = ("dop", "eop", "z", "rdash", "osh", "cop", "out")
    ins = [share_all(O.fill_random(900 + j, N), n, t, 910 + j) for j in range(6)]                    # x, y, a, b, c, r_int: sharings of random secrets
    bits = np.stack([share_all(O.fill_random(990 + j, N), n, t, 950 + j) for j in range(m)], axis=1) if m else np.zeros((n, 0, N, 4), dtype=np.uint64)
    ins[1][1, 5] = ins[1][2, 5]                                     # party 1's y share of element 5 is wrong: b - y fails, then the second open
    ins[0][0, 5] = ins[0][2, 5]                                     # and party 0's x share: a - x fails too
    res = {}
    try:
        never = (1 << 64) - 1
        for form, (fused, pair_min) in {"one": (1 << 20, never), "four": (0, 0), "five": (0, never)}.items():
            assert eng.L.hbmpc_set_fused_fpmul(eng.ctx, C.c_size_t(fused)) == 0
            assert eng.L.hbmpc_set_fpmul_pair_decode(eng.ctx, C.c_size_t(pair_min)) == 0
            eng.set_matrix_cores(True, 32 if form == "four" else 65536)   # four: the matrix-core decode from 32 chunks on
            fp = pkg.pipelines.FpMul(eng, n, t, N, k, m, stream=ts.cuda_stream)
            fp.upload(ins[0], ins[1], ins[2], ins[3], ins[4], np.ascontiguousarray(bits), ins[5])
            marker = O.fill_random(77, 2 * n * N).reshape(n, 2, N, 4)
            fp.upload_named("desh", marker)                         # the five launches' workspace: the one launch leaves it alone
            with pytest.raises(RuntimeError):
                fp.run(check=True)
            fp.run(check=False)
            wrote_desh = not GU.eq(fp.download_named("desh", (n, 2, N)), marker)
            # (the four-launch form covers t + 1 <= 11 and batches that are whole 32-element tiles; others run all five)
            assert wrote_desh == (form == "five" or (form == "four" and (t + 1 > 11 or N % 32 != 0))), "which form ran"
            got = {}
            for nm in names:
                got[nm] = fp.download_named(nm, (N,) if nm in ("dop", "eop", "cop") else (n, N)).copy()
            for nm, arr in (("status", np.zeros(2 * N, dtype=np.uint8)), ("summary", np.zeros(4, dtype=np.uint32)), ("summary_first", np.zeros(4, dtype=np.uint32))):
                eng.d2h(arr, fp.buffer(nm)[0], ts.cuda_stream)
                got[nm] = arr
            eng.sync(ts.cuda_stream)
            assert got["summary_first"].tolist() == [2, 2, 5, 8]      # a - x and b - y of element 5 (then d = e = 0: z = c, the second open is consistent)
            assert got["summary"].tolist() == [0, 0, 0xffffffff, 0] and got["status"][N + 5] == 8 and not got["status"][:N].any()
            fp.capture()
            eng.h2d(fp.out, np.zeros((n, N, 4), dtype=np.uint64), ts.cuda_stream)
            fp.replay()
            assert GU.eq(fp.download("out"), got["out"])
            res[form] = got
            fp.close()
        for form in ("one", "four"):
            for nm in names + ("status", "summary", "summary_first"):
                assert np.array_equal(res[form][nm], res["five"][nm]), (form, nm)
    finally:
        eng.L.hbmpc_set_fused_fpmul(eng.ctx, C.c_size_t(2048))
        eng.L.hbmpc_set_fpmul_pair_decode(eng.ctx, C.c_size_t(8192))
        eng.set_matrix_cores(True, 65536)


@pytest.mark.parametrize("seed", range(12))
def test_fpmul_forms_random(pkg_eng, seed):
    """hbmpc_dev_fpmul_parties on random shapes (n, t, batch, k, m; whole tiles or not; a few inconsistent shares or none): the
    one-, four- and five-launch forms leave the same bytes in every buffer, status and summary.  tools/soak_fpmul.py runs many
    more seeds."""
    import ctypes as C
    pkg, eng = pkg_eng
    rng = np.random.default_rng(0xF9 + seed)
    n = int(rng.integers(4, 41))
    t = int(rng.integers(1, (n - 1) // 3 + 1))
    N = int(rng.choice([rng.integers(1, 200), 32 * rng.integers(1, 40), rng.integers(200, 2500)]))
    k = int(rng.integers(2, 65))
    m = int(rng.integers(0, min(k, 12) + 1))
    ins = [share_all(O.fill_random(5000 + 10 * seed + j, N), n, t, 6000 + 10 * seed + j) for j in range(6)]
    bits = np.stack([share_all(O.fill_random(7000 + 20 * seed + j, N), n, t, 8000 + 20 * seed + j) for j in range(m)], axis=1) if m else np.zeros((n, 0, N, 4), dtype=np.uint64)
    for _ in range(int(rng.integers(0, 4))):       # a share of x, y, a, b, c or r_int among the senders is replaced by a neighbour's
        j, p, i = int(rng.integers(0, 6)), int(rng.integers(0, 2 * t + 1)), int(rng.integers(0, N))
        ins[j][p, i] = ins[j][(p + 1) % n, i]
    never = (1 << 64) - 1
    names = ("dop", "eop", "z", "rdash", "osh", "cop", "out")
    res = {}
    try:
        for form, (fused, pair_min) in {"one": (1 << 20, never), "four": (0, 0), "five": (0, never)}.items():
            eng.L.hbmpc_set_fused_fpmul(eng.ctx, C.c_size_t(fused))
            eng.L.hbmpc_set_fpmul_pair_decode(eng.ctx, C.c_size_t(pair_min))
            eng.set_matrix_cores(True, 32 if form == "four" else 65536)
            fp = pkg.pipelines.FpMul(eng, n, t, N, k, m)
            fp.upload(ins[0], ins[1], ins[2], ins[3], ins[4], np.ascontiguousarray(bits), ins[5])
            fp.run(check=False)
            got = {nm: fp.download_named(nm, (N,) if nm in ("dop", "eop", "cop") else (n, N)).copy() for nm in names}
            for nm, arr in (("status", np.zeros(2 * N, dtype=np.uint8)), ("summary", np.zeros(4, dtype=np.uint32)), ("summary_first", np.zeros(4, dtype=np.uint32))):
                eng.d2h(arr, fp.buffer(nm)[0])
                got[nm] = arr
            eng.sync()
            res[form] = got
            fp.close()
        for form in ("one", "four"):
            for nm in names + ("status", "summary", "summary_first"):
                assert np.array_equal(res[form][nm], res["five"][nm]), (seed, n, t, N, k, m, form, nm)
    finally:
        eng.L.hbmpc_set_fused_fpmul(eng.ctx, C.c_size_t(2048))
        eng.L.hbmpc_set_fpmul_pair_decode(eng.ctx, C.c_size_t(8192))
        eng.set_matrix_cores(True, 65536)


def test_triplegen_parties_rejects_bad_calls(pkg_eng):
    """hbmpc_dev_triplegen_parties: null buffers, N not a positive multiple of 2t + 1, n out of range or below 3t + 1 -- InvalidInput (4);
    a Goldilocks context is TypeMismatch (5); the optional status and summaries may be null"""
    pkg, eng = pkg_eng
    n, t, groups = 7, 2, 6
    N = groups * (2 * t + 1)
    tg = pkg.pipelines.TripleGen(eng, n, t, N)
    b = {nm: tg.buffer(nm)[0] for nm in ("a", "b", "r2t", "rt", "Y", "Z", "opened", "c", "status", "summary")}

    def call(eng_=eng, N_=N, n_=n, t_=t, **kw):
        a = dict(b, **kw)
        return eng_.dev_triplegen_parties(a["a"], a["b"], a["r2t"], a["rt"], N_, n_, t_, a["Y"], a["Z"], a["opened"], a["c"], a["status"], 0, a["summary"])

    assert call() == 0 and call(status=0, summary=0) == 0
    eng.sync()
    for nm in ("a", "b", "r2t", "rt", "Y", "Z", "opened", "c"):
        assert call(**{nm: 0}) == 4, nm
    assert call(N_=0) == 4 and call(N_=N + 1) == 4 and call(n_=0) == 4 and call(n_=256) == 4 and call(t_=3, N_=7 * 6) == 4
    gl = pkg.Engine(0, field="goldilocks")
    try:
        assert call(eng_=gl) == 5
    finally:
        gl.close()
    tg.close()


def test_fpmul_parties_rejects_bad_calls(pkg_eng):
    """hbmpc_dev_fpmul_parties validates like the calls it replaces: k = 0 (2^(k-1)), m beyond the supported range, m whose byte
    index the reference would read out of bounds, null buffers, n < 3t + 1, too few / duplicate / out-of-range senders -- all
    InvalidInput (4); a Goldilocks context is TypeMismatch (5)."""
    pkg, eng = pkg_eng
    n, t, N, k, m = 7, 2, 64, 16, 4
    fp = pkg.pipelines.FpMul(eng, n, t, N, k, m)
    b = {nm: fp.buffer(nm)[0] for nm in ("ta", "tb", "tc", "x", "y", "rbits", "rint", "desh", "dop", "z", "rdash", "osh", "cop", "out", "status", "summary")}

    def call(ids=tuple(range(2 * t + 1)), eng_=eng, **kw):
        a = dict(b)
        shape = dict(k=k, m=m, N=N, n=n, t=t)
        for key, v in kw.items():
            (shape if key in shape else a)[key] = v
        return eng_.dev_fpmul_parties(list(ids), a["ta"], a["tb"], a["tc"], a["x"], a["y"], a["rbits"], a["rint"], shape["k"], shape["m"], shape["N"],
                                      shape["n"], shape["t"], a["desh"], a["dop"], a["z"], a["rdash"], a["osh"], a["cop"], a["out"], a["status"], 0, a["summary"])

    assert call() == 0
    eng.sync()
    assert call(k=0) == 4 and call(m=4097) == 4 and call(m=257) == 4 and call(N=0) == 4 and call(n=0) == 4 and call(n=256) == 4
    for nm in ("ta", "tb", "tc", "x", "y", "rint", "rbits", "desh", "dop", "z", "rdash", "osh", "cop", "out"):
        assert call(**{nm: 0}) == 4, nm
    assert call(status=0, summary=0) == 0                      # both optional
    assert call(t=3) == 4                                       # n < 3t + 1
    assert call(ids=(0, 1, 2, 3)) == 4 and call(ids=(0, 1, 2, 3, 3)) == 4 and call(ids=(0, 1, 2, 3, 7)) == 4
    eng.sync()
    gl = pkg.Engine(0, field="goldilocks")
    try:
        assert call(eng_=gl) == 5
    finally:
        gl.close()
    fp.close()


def test_fpmul_pipeline_as_hip_graph(pkg_eng):
    """The whole fpmul call sequence captured once into a HIP graph (hbmpc_graph_*) and replayed on refilled
    buffers gives exactly what the eager calls give."""
    pkg, eng = pkg_eng
    n, t, N, k, m = 7, 2, 300, 16, 4
    # a real (non-NULL) stream for the capture: created through HIP by torch, passed as a raw handle
    torch = pytest.importorskip("torch")
    ts = torch.cuda.Stream(device=torch.device("cuda", 0))
    fp = pkg.pipelines.FpMul(eng, n, t, N, k, m, stream=ts.cuda_stream)
    rng = np.random.default_rng(99)

    def fill(seed):
        half = (k - 2) // 2
        x = O.ints_to_u256([int(v) for v in rng.integers(0, 1 << half, N)])
        y = O.ints_to_u256([int(v) for v in rng.integers(0, 1 << half, N)])
        ta, tb = O.fill_random(seed, N), O.fill_random(seed + 1, N)
        tc = O.fr_binop("mul", ta, tb)
        rint = O.ints_to_u256([int(v) for v in rng.integers(0, 1 << 40, N)])
        bits = rng.integers(0, 2, (m, N))
        sb = np.stack([share_all(O.ints_to_u256([int(v) for v in bits[j]]), n, t, seed + 10 + j) for j in range(m)], axis=1)
        fp.upload(share_all(x, n, t, seed + 2), share_all(y, n, t, seed + 3), share_all(ta, n, t, seed + 4),
                  share_all(tb, n, t, seed + 5), share_all(tc, n, t, seed + 6), np.ascontiguousarray(sb),
                  share_all(rint, n, t, seed + 7))

    fill(500)
    fp.capture()                                  # one eager run, then the same calls recorded
    for seed in (500, 600, 700):
        fill(seed)
        fp.run()                                  # eager (checked) on this data
        want = fp.download("out").copy()
        eng.h2d(fp.out, np.zeros_like(want), ts.cuda_stream)
        fp.replay()
        assert GU.eq(fp.download("out"), want), seed
    fp.close()


@pytest.mark.parametrize("n,t,G,parties", [(4, 1, 77, 4), (7, 2, 130, 7), (10, 3, 65, 3), (13, 4, 200, 13), (16, 5, 333, 16),
                                           (16, 5, 64, 1), (8, 1, 50, 2), (20, 6, 40, 3), (5, 1, 10, 5)])
def test_triple_encode_fused_equals_two_launches(pkg_eng, n, t, G, parties):
    """hbmpc_dev_triple_encode_parties (local product fused into the encode where a kernel exists; (20, 6) has none and
    takes the workspace path) against hbmpc_dev_triple_local + hbmpc_dev_vandermonde_apply_parties, and the first party
    against the oracle"""
    import torch
    pkg, eng = pkg_eng
    d = 2 * t
    N = G * (d + 1)
    dev = torch.device("cuda", 0)
    a, b, r = (torch.from_numpy(O.fill_random(60 + k, parties * N).view(np.int64)).to(dev) for k in range(3))
    tmp = torch.empty((parties * N, 4), dtype=torch.int64, device=dev)
    y1 = torch.full((parties, n, G, 4), -1, dtype=torch.int64, device=dev)
    y2 = torch.full((parties, n, G, 4), -1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    assert eng.dev_triple_encode_parties(a.data_ptr(), b.data_ptr(), r.data_ptr(), G, n, d, parties, tmp.data_ptr(), y1.data_ptr()) == 0
    assert eng.dev_elem("triple_local", [a.data_ptr(), b.data_ptr(), r.data_ptr(), tmp.data_ptr()], parties * N) == 0
    assert eng.dev_vandermonde_apply_parties(tmp.data_ptr(), G, n, d, parties, y2.data_ptr()) == 0
    eng.sync()
    assert torch.equal(y1, y2)
    ah, bh, rh = (v[:N].cpu().numpy().view(np.uint64) for v in (a, b, r))
    x = O.triple_local(ah, bh, rh)[1].reshape(G, d + 1, 4)
    rc, want = O.vandermonde_apply(x, n, d)
    assert rc == 0 and np.array_equal(y1[0].cpu().numpy().view(np.uint64), want)
    if n <= 16 and d + 1 in (3, 5, 7, 9, 11):   # the fused kernel needs no workspace
        y3 = torch.full_like(y1, -1)
        assert eng.dev_triple_encode_parties(a.data_ptr(), b.data_ptr(), r.data_ptr(), G, n, d, parties, 0, y3.data_ptr()) == 0
        eng.sync()
        assert torch.equal(y3, y1)
    else:
        assert eng.dev_triple_encode_parties(a.data_ptr(), b.data_ptr(), r.data_ptr(), G, n, d, parties, 0, y1.data_ptr()) == 4


@pytest.mark.parametrize("N,parties", [(1, 1), (77, 4), (1000, 16), (5000, 3)])
def test_beaver_open_shares_paired(pkg_eng, N, parties):
    """hbmpc_dev_beaver_open_shares_paired: de[party][0][N] = a - x, de[party][1][N] = b - y, i.e. the two outputs of
    hbmpc_dev_beaver_open_shares (multiplication.rs:417-426) interleaved per party; and the pair opens with ONE
    interpolation call over 2 N values per sender"""
    import torch
    pkg, eng = pkg_eng
    dev = torch.device("cuda", 0)
    a, b, x, y = (torch.from_numpy(O.fill_random(80 + k, parties * N).view(np.int64)).to(dev) for k in range(4))
    d1 = torch.full((parties * N, 4), -1, dtype=torch.int64, device=dev)
    e1 = torch.full((parties * N, 4), -1, dtype=torch.int64, device=dev)
    de = torch.full((parties, 2, N, 4), -1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    assert eng.dev_elem("beaver_open_shares", [a.data_ptr(), b.data_ptr(), x.data_ptr(), y.data_ptr(), d1.data_ptr(), e1.data_ptr()], parties * N) == 0
    assert eng.dev_beaver_open_shares_paired(a.data_ptr(), b.data_ptr(), x.data_ptr(), y.data_ptr(), N, parties, de.data_ptr()) == 0
    eng.sync()
    assert torch.equal(de[:, 0].reshape(parties * N, 4), d1) and torch.equal(de[:, 1].reshape(parties * N, 4), e1)
    want = O.fr_binop("sub", a[:N].cpu().numpy().view(np.uint64), x[:N].cpu().numpy().view(np.uint64))
    assert GU.eq(de[0, 0].cpu().numpy().view(np.uint64), want)


@pytest.mark.parametrize("N,parties,k,m", [(1, 1, 16, 4), (77, 4, 16, 4), (1000, 16, 32, 16), (70000, 2, 16, 0), (300, 3, 40, 33)])
def test_fpmul_middle_equals_the_three_calls(pkg_eng, N, parties, k, m):
    """hbmpc_dev_fpmul_middle = beaver_finalize_parties + truncpr_rdash_parties + truncpr_open_share, byte for byte in all
    three outputs (both the in-thread party loop of large N and the party-per-block form of small N)"""
    import torch
    pkg, eng = pkg_eng
    dev = torch.device("cuda", 0)
    c, x, y, rint = (torch.from_numpy(O.fill_random(90 + q, parties * N).view(np.int64)).to(dev) for q in range(4))
    d, e = (torch.from_numpy(O.fill_random(95 + q, N).view(np.int64)).to(dev) for q in range(2))
    bits = torch.from_numpy(O.fill_random(99, parties * max(m, 1) * N).view(np.int64)).to(dev)   # any field elements will do
    z1, rd1, o1, z2, rd2, o2 = (torch.full((parties * N, 4), -1, dtype=torch.int64, device=dev) for _ in range(6))
    torch.cuda.synchronize()
    P = lambda t_: t_.data_ptr()   # noqa: E731
    assert eng.dev_elem_parties("beaver_finalize", [P(c), P(x), P(y), P(d), P(e), P(z1)], N, parties) == 0
    assert eng.dev_elem_parties("truncpr_rdash", [P(bits), P(rd1)], N, parties, extra=(m,)) == 0
    assert eng.dev_elem("truncpr_open_share", [P(z1), P(rd1), P(rint), P(o1)], parties * N, extra=(k, m)) == 0
    assert eng.dev_fpmul_middle(P(c), P(x), P(y), P(d), P(e), P(bits), P(rint), k, m, N, parties, P(z2), P(rd2), P(o2)) == 0
    eng.sync()
    assert torch.equal(z1, z2) and torch.equal(rd1, rd2) and torch.equal(o1, o2)
