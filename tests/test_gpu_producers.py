"""The producer side of config 4 (Missing 1): RanSha, DouSha + RanDouSha as device-resident pipelines of all
n parties (mpc-protocols_amd/pipelines.py) against the oracle's restatement of share_gen/share_gen.rs,
double_share/double_share_generation.rs and ran_dou_sha/mod.rs (oracle/spec.py: ransha, randousha) on the same dealers'
polynomials -- every party's output shares bit for bit, in the reference's order, and the verifiers' verdicts; then the
whole chain dealers -> [c]_t (run_preprocessing's triple part, honeybadger/mod.rs:1239-1393)."""
import random

import numpy as np
import pytest

from __graft_entry__ import load_package
from oracle import spec as SFR
from oracle.cref import ints_to_u256, u256_to_ints
from oracle.spec_gl import S as SGL

pytestmark = pytest.mark.gpu


def _field(field):
    if field == "fr":
        return SFR, (lambda v: ints_to_u256(v)), (lambda a: [u256_to_ints(r) for r in a])
    return SGL, (lambda v: np.array(v, dtype=np.uint64)), (lambda a: [[int(x) for x in r] for r in a])


def _coeffs(S, rng, n, K, deg, secrets=None):
    return [[[secrets[p][k] if secrets else rng.randrange(S.R_MOD)] + [rng.randrange(S.R_MOD) for _ in range(deg)] for k in range(K)]
            for p in range(n)]


@pytest.mark.parametrize("field", ["fr", "goldilocks"])
@pytest.mark.parametrize("n,t,K", [(4, 1, 5), (7, 2, 33), (16, 5, 7)])
def test_ransha_matches_the_oracle(field, n, t, K):
    S, to_dev, to_int = _field(field)
    pkg = load_package()
    eng = pkg.Engine(0, field="fr" if field == "fr" else "goldilocks")
    rng = random.Random(n * 100 + K)
    co = _coeffs(S, rng, n, K, t)
    want, ok = S.ransha(co, n, t)
    rs = pkg.pipelines.RanSha(eng, n, t, K)
    try:
        rs.upload(to_dev([c for row in co for poly in row for c in poly]))
        rs.run(check=True)
        assert all(ok) and to_int(rs.download()) == want
        # all n senders to the verifiers (OEC rounds available): same outputs, same verdict
        rs2 = pkg.pipelines.RanSha(eng, n, t, K, verify_senders=n)
        rs2.upload(to_dev([c for row in co for poly in row for c in poly]))
        rs2.run(check=True)
        assert to_int(rs2.download()) == want
        rs2.close()
        # a dealer sends ONE recipient a wrong share for one batch element: the mixed sharings are no longer of degree t
        # (share_gen.rs:516-530 -> Output(false)); the oracle restatement says the same about the same tampered inputs
        rs.deal()
        k_bad, p_bad, j_bad = K // 2, n - 1, 0
        U = rs.U
        one = to_dev([1])
        cur = eng._new((1,))
        off = ((p_bad * n + j_bad) * K + k_bad) * U
        eng.d2h(cur, rs.S + off)
        eng.sync()
        v = (int(to_int([cur])[0][0]) + 1) % S.R_MOD
        eng.h2d(rs.S + off, to_dev([v]))
        with pytest.raises(RuntimeError, match="RanSha"):
            rs.finish(check=True)
        bad, first = rs._bad()
        assert bad >= 1 and first == k_bad
        del one
    finally:
        rs.close()
        eng.close()


@pytest.mark.parametrize("field", ["fr", "goldilocks"])
@pytest.mark.parametrize("n,t,K", [(4, 1, 5), (7, 2, 33), (16, 5, 7)])
def test_randousha_matches_the_oracle(field, n, t, K):
    S, to_dev, to_int = _field(field)
    pkg = load_package()
    eng = pkg.Engine(0, field="fr" if field == "fr" else "goldilocks")
    rng = random.Random(n * 1000 + K)
    ct = _coeffs(S, rng, n, K, t)
    c2t = _coeffs(S, rng, n, K, 2 * t, secrets=[[ct[p][k][0] for k in range(K)] for p in range(n)])
    want_t, want_2t, ok = S.randousha(ct, c2t, n, t)
    rd = pkg.pipelines.RanDouSha(eng, n, t, K)
    try:
        rd.upload(to_dev([c for row in ct for poly in row for c in poly]), to_dev([c for row in c2t for poly in row for c in poly]))
        rd.run(check=True)
        a, b = rd.download()
        assert all(ok) and to_int(a) == want_t and to_int(b) == want_2t
        # a dealer whose two sharings hide DIFFERENT secrets (double_share_generation.rs deals the same one): the verifiers'
        # equal-secret test (ran_dou_sha/mod.rs:588) must fail, and the oracle agrees
        c2t[2][1][0] = (c2t[2][1][0] + 1) % S.R_MOD
        _, _, ok_bad = S.randousha(ct, c2t, n, t)
        assert not any(ok_bad)
        rd.upload(to_dev([c for row in ct for poly in row for c in poly]), to_dev([c for row in c2t for poly in row for c in poly]))
        with pytest.raises(RuntimeError, match="RanDouSha"):
            rd.run(check=True)
        bad, first = rd._bad()
        assert bad == n - (t + 1) and first == 1
    finally:
        rd.close()
        eng.close()


@pytest.mark.parametrize("n,t,groups", [(4, 1, 3), (7, 2, 4), (16, 5, 2), (4, 1, 2), (7, 2, 3), (16, 5, 6)])   # the last three: N divides into
def test_preprocessing_chain_from_dealers_to_triples(n, t, groups):                                                 # whole batch elements (no copies)
    """dealers' polynomials -> RanSha -> a, b; DouSha + RanDouSha -> ([r]_t, [r]_2t); TripleGen -> [c]_t, nothing leaves the
    device in between.  Checked against the oracle end to end: every party's c share equals the restatement's, and the
    shares reconstruct to a * b."""
    S = SFR
    pkg = load_package()
    eng = pkg.Engine(0)
    N = groups * (2 * t + 1)
    pre = pkg.pipelines.Preprocessing(eng, n, t, N)
    rng = random.Random(n + 17 * groups)
    try:
        co = _coeffs(S, rng, n, pre.K_rs, t)
        ct = _coeffs(S, rng, n, pre.K_rd, t)
        c2t = _coeffs(S, rng, n, pre.K_rd, 2 * t, secrets=[[ct[p][k][0] for k in range(pre.K_rd)] for p in range(n)])
        flat = lambda cc: ints_to_u256([c for row in cc for poly in row for c in poly])
        pre.rs.upload(flat(co))
        pre.rd.upload(flat(ct), flat(c2t))
        pre.run(check=True)
        c_dev = [u256_to_ints(r) for r in pre.tg.download_c()]
        # the oracle: same producers, then triple_generation.rs per party and BatchRecon's opened values
        rs_out, ok1 = S.ransha(co, n, t)
        rt_out, r2t_out, ok2 = S.randousha(ct, c2t, n, t)
        assert all(ok1) and all(ok2)
        a = [rs_out[j][:N] for j in range(n)]
        b = [rs_out[j][N:2 * N] for j in range(n)]
        for k in range(N):
            masked = [S.Share((a[j][k] * b[j][k] - r2t_out[j][k]) % S.R_MOD, j, 2 * t) for j in range(n)]
            poly, opened = S.recover_secret(masked, n, t)        # a b - r, a degree-2t sharing opened robustly
            for j in range(n):
                assert c_dev[j][k] == (rt_out[j][k] + opened) % S.R_MOD
            # and the triple is a triple
            sa = S.recover_secret([S.Share(a[j][k], j, t) for j in range(n)], n, t)[1]
            sb = S.recover_secret([S.Share(b[j][k], j, t) for j in range(n)], n, t)[1]
            sc = S.recover_secret([S.Share(c_dev[j][k], j, t) for j in range(n)], n, t)[1]
            assert sc == sa * sb % S.R_MOD
    finally:
        pre.close()
        eng.close()


@pytest.mark.parametrize("K", [20000, 70000])   # 70 000: beyond the lane kernels' and the team kernels' ranges, dealers and verifiers still together
def test_producers_at_a_size_that_takes_the_large_batch_kernels(K):
    """K = 20 000 (70 000) batch elements per dealer, n = 16: the dealers' encodes, the n x n mixing step over the dealt rows
    (hbmpc_dev_vandermonde_apply_rows, d + 1 = 16) and the RanDouSha verifiers' full-domain interpolation with its degrees all run
    on the point-pair matrix-core kernel here (the small cases above stay below its thresholds).  Every party's outputs for sampled
    batch elements against the oracle, the verifiers' verdicts, and a tampered share caught at this size too."""
    from oracle import cref as O
    S = SFR
    n, t = 16, 5
    pkg = load_package()
    eng = pkg.Engine(0)
    rng = random.Random(99)
    sample = sorted(rng.sample(range(K), 12) + [0, K - 1])
    try:
        co = O.fill_random(1234, n * K * (t + 1)).reshape(n, K, t + 1, 4)
        rs = pkg.pipelines.RanSha(eng, n, t, K)
        rs.upload(co)
        rs.run(check=True)
        got = rs.download().reshape(n, K, n - 2 * t, 4)
        pol = [[u256_to_ints(co[p, k]) for k in sample] for p in range(n)]
        want, ok = S.ransha(pol, n, t)
        assert all(ok)
        for j in range(n):
            assert u256_to_ints(np.ascontiguousarray(got[j, sample]).reshape(-1, 4)) == want[j], j
        # one dealt share changed: caught by the verifiers, first failing column reported
        rs.deal()
        k_bad = 12345
        off = ((3 * n + 7) * K + k_bad) * 32
        cur = eng._new((1,))
        eng.d2h(cur, rs.S + off)
        eng.sync()
        eng.h2d(rs.S + off, ints_to_u256([(u256_to_ints(cur)[0] + 1) % S.R_MOD]))
        with pytest.raises(RuntimeError, match="RanSha"):
            rs.finish(check=True)
        assert rs._bad()[1] == k_bad
        rs.close()

        ct = O.fill_random(77, n * K * (t + 1)).reshape(n, K, t + 1, 4)
        c2t = O.fill_random(78, n * K * (2 * t + 1)).reshape(n, K, 2 * t + 1, 4)
        c2t[:, :, 0] = ct[:, :, 0]
        rd = pkg.pipelines.RanDouSha(eng, n, t, K)
        rd.upload(ct, c2t)
        rd.run(check=True)
        a, b = rd.download()
        a, b = a.reshape(n, K, t + 1, 4), b.reshape(n, K, t + 1, 4)
        want_t, want_2t, ok = S.randousha([[u256_to_ints(ct[p, k]) for k in sample] for p in range(n)],
                                          [[u256_to_ints(c2t[p, k]) for k in sample] for p in range(n)], n, t)
        assert all(ok)
        for j in range(n):
            assert u256_to_ints(np.ascontiguousarray(a[j, sample]).reshape(-1, 4)) == want_t[j], j
            assert u256_to_ints(np.ascontiguousarray(b[j, sample]).reshape(-1, 4)) == want_2t[j], j
        # a dealer whose degree-2t sharing hides another secret in ONE column: every verifier's equal-secret test fails there
        c2t[5, 4321, 0, 0] ^= np.uint64(1)
        rd.upload(ct, c2t)
        with pytest.raises(RuntimeError, match="RanDouSha"):
            rd.run(check=True)
        bad, first = rd._bad()
        assert bad == n - (t + 1) and first == 4321
        rd.close()
    finally:
        eng.close()


@pytest.mark.parametrize("field,n,t,K", [("goldilocks", 16, 5, 3000), ("goldilocks", 7, 2, 5001), ("goldilocks", 10, 3, 2500), ("fr", 10, 3, 4000),
                                         ("fr", 4, 1, 6000), ("goldilocks", 4, 1, 6000),
                                         ("fr", 7, 2, 2048), ("goldilocks", 7, 2, 1536), ("fr", 16, 5, 1536)])   # the node's column counts per run
def test_producers_at_the_batch_sizes_of_the_reference_node(field, n, t, K):
    """Some thousands of columns per dealer (the reference node's own batches, honeybadger/mod.rs:106-120): the dealers' encodes are one
    launch over (dealer, polynomial), the verifiers' rows are party-major and every kind of verifier is ONE decode over (verifier, column)
    chunks (capi_pipelines.hip: Producer::verifiers_together; over Goldilocks and on 4-point domains through k_rows_party_major and the
    full interpolations).  Sampled columns of every party's outputs against the oracle, the verdicts, tampering caught with its column."""
    from oracle import cref as O
    S, to_dev, to_int = _field(field)
    gl = field != "fr"
    pkg = load_package()
    eng = pkg.Engine(0, field="goldilocks" if gl else "fr")
    rng = random.Random(7 * n + K)
    nrng = np.random.default_rng(K + n)
    sample = sorted(set(rng.sample(range(K), 10) + [0, K - 1]))

    def rand(count, shape):
        if gl:
            return nrng.integers(0, S.R_MOD, size=shape, dtype=np.uint64)
        return O.fill_random(int(nrng.integers(1, 1 << 30)), count).reshape(shape + (4,))

    def ints(a):  # [..] elements -> Python ints
        return [int(x) for x in a.reshape(-1)] if gl else u256_to_ints(np.ascontiguousarray(a).reshape(-1, 4))

    try:
        co = rand(n * K * (t + 1), (n, K, t + 1))
        rs = pkg.pipelines.RanSha(eng, n, t, K)
        rs.upload(co)
        rs.run(check=True)
        got = rs.download().reshape((n, K, n - 2 * t) + ((4,) if not gl else ()))
        pol = [[ints(co[p, k]) for k in sample] for p in range(n)]
        want, ok = S.ransha(pol, n, t)
        assert all(ok)
        for j in range(n):
            assert ints(got[j, sample]) == want[j], j
        rs.deal()
        k_bad, U = K - 7, rs.U
        off = ((2 * n + 1) * K + k_bad) * U
        cur = eng._new((1,))
        eng.d2h(cur, rs.S + off)
        eng.sync()
        eng.h2d(rs.S + off, to_dev([(ints(cur)[0] + 1) % S.R_MOD]))
        with pytest.raises(RuntimeError, match="RanSha"):
            rs.finish(check=True)
        assert rs._bad()[0] >= 1 and rs._bad()[1] == k_bad
        rs.close()

        ct = rand(n * K * (t + 1), (n, K, t + 1))
        c2t = rand(n * K * (2 * t + 1), (n, K, 2 * t + 1))
        c2t[:, :, 0] = ct[:, :, 0]
        rd = pkg.pipelines.RanDouSha(eng, n, t, K)
        rd.upload(ct, c2t)
        rd.run(check=True)
        a, b = rd.download()
        tail = (4,) if not gl else ()
        a, b = a.reshape((n, K, t + 1) + tail), b.reshape((n, K, t + 1) + tail)
        want_t, want_2t, ok = S.randousha([[ints(ct[p, k]) for k in sample] for p in range(n)], [[ints(c2t[p, k]) for k in sample] for p in range(n)], n, t)
        assert all(ok)
        for j in range(n):
            assert ints(a[j, sample]) == want_t[j] and ints(b[j, sample]) == want_2t[j], j
        k_bad = K // 3
        if gl:
            c2t[3, k_bad, 0] ^= np.uint64(1)
        else:
            c2t[3, k_bad, 0, 0] ^= np.uint64(1)
        rd.upload(ct, c2t)
        with pytest.raises(RuntimeError, match="RanDouSha"):
            rd.run(check=True)
        bad, first = rd._bad()
        assert bad == n - (t + 1) and first == k_bad
        rd.close()
    finally:
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,t,K", [(16, 5, 3000), (7, 2, 9000), (13, 4, 4001)])
def test_fused_producer_steps_write_the_same_bytes(n, t, K):
    """The mixing step that writes the parties' lists itself (k_mfma_bfly<.., LISTS>) and the verifier interpolation that keeps
    only c0 and the degree (hbmpc_dev_batch_interpolate_c0) against the separate passes (hbmpc_set_producer_fusion(ctx, 0):
    every row into y, hbmpc_dev_transpose, full coefficient rows + hbmpc_dev_check_double_share): whole outputs, verdicts and
    the first failing column, for RanSha, RanDouSha and the two-slice split of Preprocessing (a | b of TripleGen)."""
    from oracle import cref as O
    import ctypes as C
    pkg = load_package()
    eng = pkg.Engine(0)
    try:
        co = O.fill_random(11 + n, n * K * (t + 1)).reshape(n, K, t + 1, 4)
        ct = O.fill_random(12 + n, n * K * (t + 1)).reshape(n, K, t + 1, 4)
        c2t = O.fill_random(13 + n, n * K * (2 * t + 1)).reshape(n, K, 2 * t + 1, 4)
        c2t[:, :, 0] = ct[:, :, 0]
        c2t[2, K // 3, 0, 1] ^= np.uint64(4)          # one column fails the equal-secret test
        ct[1, K // 2, t] = 0                           # ... one has a degree-t polynomial of lower degree (dealer 1's; the column sums still have degree t)
        res = {}
        for fused in (1, 0):
            assert eng.L.hbmpc_set_producer_fusion(eng.ctx, C.c_int(fused)) == 0
            rs = pkg.pipelines.RanSha(eng, n, t, K)
            rs.upload(co)
            rs.run(check=True)
            out = rs.download().copy()
            rs.deal()
            off = ((1 * n + 2) * K + K - 5) * 32
            cur = eng._new((1,))
            eng.d2h(cur, rs.S + off)
            eng.sync()
            cur[0, 0] ^= np.uint64(1)
            eng.h2d(rs.S + off, cur)
            rs.finish(check=False)
            res[fused] = [out, rs._bad()]
            rs.close()
            rd = pkg.pipelines.RanDouSha(eng, n, t, K)
            rd.upload(ct, c2t)
            rd.run(check=False)
            a, b = rd.download()
            res[fused] += [a.copy(), b.copy(), rd._bad()]
            rd.close()
        assert np.array_equal(res[1][0], res[0][0]) and res[1][1] == res[0][1] and res[1][1][0] >= 1 and res[1][1][1] == K - 5
        assert np.array_equal(res[1][2], res[0][2]) and np.array_equal(res[1][3], res[0][3]) and res[1][4] == res[0][4]
        assert res[1][4][0] == n - (t + 1) and res[1][4][1] == K // 3
        # Preprocessing: the lists land in TripleGen's arrays through two slices (a = the first N of a party's list, b = the next N)
        N = (2 * t + 1) * (n - 2 * t) * (t + 1) * 60
        outs = {}
        for fused in (1, 0):
            assert eng.L.hbmpc_set_producer_fusion(eng.ctx, C.c_int(fused)) == 0
            pre = pkg.pipelines.Preprocessing(eng, n, t, N)
            pre.rs.upload(O.fill_random(21, n * pre.K_rs * (t + 1)).reshape(n, pre.K_rs, t + 1, 4))
            dt = O.fill_random(22, n * pre.K_rd * (t + 1)).reshape(n, pre.K_rd, t + 1, 4)
            d2t = O.fill_random(23, n * pre.K_rd * (2 * t + 1)).reshape(n, pre.K_rd, 2 * t + 1, 4)
            d2t[:, :, 0] = dt[:, :, 0]
            pre.rd.upload(dt, d2t)
            pre.run(check=True)
            outs[fused] = [pre.tg.download_named(nm, (n, N)).copy() for nm in ("a", "b", "rt", "r2t", "c")]
            pre.close()
        for x, y in zip(outs[1], outs[0]):
            assert np.array_equal(x, y)
    finally:
        eng.L.hbmpc_set_producer_fusion(eng.ctx, C.c_int(1))
        eng.close()
