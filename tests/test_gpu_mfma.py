"""The matrix-core (int8 MFMA) decode, csrc/kernels_mfma.hpp, against the oracle and against the lane-per-chunk
kernels: every d + 1 it instantiates, role splits, ragged tails, the multi-tile loop of a wave (few workgroups),
corruption inside and outside the interpolation set, arrival orders, missing senders, P(0)-only, strided rows, and
BASELINE configs[2] at full size.  Bar: bit-exact."""
import numpy as np
import pytest

from __graft_entry__ import load_package
from oracle import cref as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = load_package().Engine(0)
    e.set_small_batch_chunks(0)
    yield e
    e.close()


def rnd(seed, *shape):
    return O.fill_random(seed, int(np.prod(shape))).reshape(*shape, 4)


def codewords(seed, G, n, d):
    x = rnd(seed, G, d + 1)
    x[0] = 0                                   # the zero polynomial
    x[1 % G, :, :] = 0
    x[1 % G, 0, 0] = 1                         # the constant 1
    if G > 2:
        x[2, :] = O.ints_to_u256([O_R - 1] * (d + 1))   # every coefficient r - 1
    rc, y = O.vandermonde_apply(x, n, d)
    assert rc == 0
    return x, y


O_R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


def both_ways(eng, ids, ev, n, d, t, p0=False):
    """the same call through the matrix-core kernels and through the lane-per-chunk kernels"""
    call = eng.batch_recover_p0 if p0 else eng.batch_recover
    eng.set_matrix_cores(True, 1)
    a = call(ids, ev, n, d, t)
    eng.set_matrix_cores(2, 1)               # without the workgroup-per-tile kernel that small batches take
    a2 = call(ids, ev, n, d, t)
    assert a[0] == a2[0] and all(np.array_equal(u, v) for u, v in zip(a[1:], a2[1:]))
    eng.set_matrix_cores(False)
    b = call(ids, ev, n, d, t)
    eng.set_matrix_cores(True, 65536)
    return a, b


SHAPES = [  # (n, d, t): d + 1 = 2 .. 15, verify rows 0 .. 10; (16, 10, 5) is config 4's decode, (31, 10, 10) config 3's
    (4, 1, 1), (7, 2, 2), (10, 3, 3), (13, 4, 4), (16, 5, 5), (19, 6, 6), (22, 7, 7), (25, 8, 8), (28, 9, 9),
    (31, 10, 10), (16, 10, 5), (34, 11, 11), (37, 12, 12), (40, 13, 13), (43, 14, 13), (16, 14, 0), (31, 4, 10),
    (5, 1, 1), (10, 3, 2), (20, 6, 6),
]


@pytest.mark.parametrize("n,d,t", SHAPES)
@pytest.mark.parametrize("wgs", [0, 8])
def test_shapes_vs_oracle(eng, n, d, t, wgs):
    G = 1500 + 37 if wgs == 0 else 9000 + 5   # wgs = 8: every wave walks several tiles (both input register sets)
    eng.set_matrix_core_workgroups(wgs)
    try:
        x, y = codewords(100 + n + d, G, n, d)
        rng = np.random.default_rng(n * 1000 + d)
        # corruption: a verify row (if any), an interpolation row, two rows at once, and a chunk beyond repair
        y[min(d + 1, n - 1), 5, 0] ^= 1
        y[0, 6, 3] ^= 1 << 40
        y[1, 7, 1] ^= 7
        y[min(d + 2, n - 1), 7, 2] ^= 9
        for j in rng.choice(n, size=min(n, 2 * t + 2), replace=False):
            y[j, 8, 0] ^= 0x55
        ids = list(range(n))
        (rc, co, nco, st), lane = both_ways(eng, ids, y, n, d, t)
        rc0, co0, nco0, st0 = O.batch_recover(ids, y, n, d, t)
        assert rc == rc0 and np.array_equal(st, st0) and np.array_equal(nco, nco0) and np.array_equal(co, co0)
        assert lane[0] == rc and all(np.array_equal(u, v) for u, v in zip(lane[1:], (co, nco, st)))
        ok = st0 == 0
        ok[5:9] = False                        # t = 0 has no verify rows: a corrupted chunk is accepted as another polynomial
        assert ok.sum() >= G - 4 and np.array_equal(co[ok], x[ok])
        (rc, p0, st), lane = both_ways(eng, ids, y, n, d, t, p0=True)
        assert rc == rc0 and np.array_equal(st, st0) and np.array_equal(p0, co0[:, 0])
        assert lane[0] == rc and np.array_equal(lane[1], p0) and np.array_equal(lane[2], st)
    finally:
        eng.set_matrix_core_workgroups(0)


@pytest.mark.parametrize("n,d,t", [(31, 10, 10), (16, 5, 5), (16, 10, 5)])
def test_arrival_orders_and_missing_senders(eng, n, d, t):
    G = 3000 + 11
    x, y = codewords(7 + n, G, n, d)
    rng = np.random.default_rng(n + d)
    for trial in range(4):
        S = n if trial == 0 else int(rng.integers(d + t + 1, n + 1))
        ids = [int(i) for i in rng.permutation(n)[:S]]
        ev = np.ascontiguousarray(y[ids])
        bad = rng.choice(G, size=20, replace=False)
        ev[int(rng.integers(0, S)), bad, 0] ^= 1     # one lying sender in 20 chunks
        (rc, co, nco, st), lane = both_ways(eng, ids, ev, n, d, t)
        rc0, co0, nco0, st0 = O.batch_recover(ids, ev, n, d, t)
        assert rc == rc0 and np.array_equal(st, st0) and np.array_equal(nco, nco0) and np.array_equal(co, co0)
        assert lane[0] == rc and all(np.array_equal(u, v) for u, v in zip(lane[1:], (co, nco, st)))


def test_full_size_config3(eng):
    """BASELINE configs[2]: n = 31, t = 10, 2^20 chunks through the device API; whole-array equality with the
    lane-per-chunk kernels, a sampled comparison with the oracle, and the round trip."""
    import torch
    n, t, d, G = 31, 10, 10, 1 << 20
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0xC0FFEE02)
    lo = torch.randint(0, 1 << 62, (G, d + 1, 3), dtype=torch.int64, device=dev, generator=gen)
    hi = torch.randint(0, 0x73EDA753299D7D48, (G, d + 1, 1), dtype=torch.int64, device=dev, generator=gen)
    x = torch.cat([lo, hi], dim=-1).contiguous()
    y = torch.empty((n, G, 4), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    assert eng.dev_vandermonde_apply(x.data_ptr(), G, n, d, y.data_ptr(), 0) == 0
    eng.sync()
    # corrupt: the t lowest senders in 1 % of the chunks, plus a verify-row lie and an unrepairable chunk
    bad = torch.randperm(G, device=dev, generator=gen)[: G // 100]
    y[:t, bad, 0] ^= 1
    y[15, 12345, 2] ^= 1 << 33
    y[:, 777, 1] ^= 3
    ids = list(range(n))
    outs = {}
    for mode in ("mfma", "lane"):
        eng.set_matrix_cores(mode == "mfma", 65536)
        co = torch.full((G, d + 1, 4), -1, dtype=torch.int64, device=dev)
        nco = torch.zeros((G,), dtype=torch.int32, device=dev)
        st = torch.zeros((G,), dtype=torch.uint8, device=dev)
        summ = torch.zeros((4,), dtype=torch.int32, device=dev)
        sec = torch.full((G, 4), -1, dtype=torch.int64, device=dev)
        st2 = torch.zeros((G,), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        assert eng.dev_batch_recover(ids, y.data_ptr(), G, n, d, t, co.data_ptr(), nco.data_ptr(), st.data_ptr(), summ.data_ptr(), 0) == 0
        assert eng.dev_batch_recover(ids, y.data_ptr(), G, n, d, t, sec.data_ptr(), 0, st2.data_ptr(), 0, 0, p0=True) == 0
        eng.sync()
        outs[mode] = (co, nco, st, summ, sec, st2)
    eng.set_matrix_cores(True, 65536)
    for u, v in zip(outs["mfma"], outs["lane"]):
        assert torch.equal(u, v)
    co, nco, st, summ, sec, st2 = outs["mfma"]
    sm = summ.cpu().numpy().view(np.uint32)
    assert int(sm[0]) == G // 100 + 2 - int((bad == 12345).sum()) - int((bad == 777).sum()) and int(sm[1]) == 1 and int(sm[2]) == 777
    good = st <= 1
    assert int(good.sum()) == G - 1 and torch.equal(co[good], x[good]) and torch.equal(sec[good], x[good][:, 0])
    # sampled chunks against the oracle: first, last, tile and workgroup boundaries, the corrupted ones
    idx = np.unique(np.concatenate([np.arange(0, 70), np.arange(G - 70, G), np.arange(31, G, 32)[:50], np.arange(32 * 3072 - 3, 32 * 3072 + 3),
                                    bad.cpu().numpy()[:300], [12345, 777]]))
    ys = y[:, torch.as_tensor(idx, device=dev)].cpu().numpy().view(np.uint64)
    rc0, co0, nco0, st0 = O.batch_recover(ids, np.ascontiguousarray(ys), n, d, t)
    ti = torch.as_tensor(idx, device=dev)
    assert np.array_equal(co[ti].cpu().numpy().view(np.uint64), co0) and np.array_equal(st[ti].cpu().numpy(), st0)
    assert np.array_equal(nco[ti].cpu().numpy().view(np.uint32), nco0)


def test_gather_party_major_two_contexts():
    """hbmpc_dev_gather_party_major: two contexts (here both on device 0; on a multi-GPU node one per device) compute the
    shares of their contiguous slices of a batch; the gather puts the party-major rows of the whole batch on the root.
    Ragged slices, strided shard rows, against the single-context result."""
    import torch
    pkg = load_package()
    e0, e1 = pkg.Engine(0), pkg.Engine(0)
    try:
        n, d, B = 16, 5, 1001
        coeffs = rnd(321, B, d + 1)
        rc, want = O.compute_shares(coeffs, n, d)
        lo = 517                                           # slices of 517 and 484 secrets
        dev = torch.device("cuda", 0)
        cz = torch.from_numpy(coeffs.view(np.int64)).to(dev)
        s0 = torch.empty((n, lo + 3, 4), dtype=torch.int64, device=dev)          # row stride 520 > 517 columns
        s1 = torch.empty((n, B - lo, 4), dtype=torch.int64, device=dev)
        out = torch.full((n, B + 5, 4), -1, dtype=torch.int64, device=dev)
        c0, c1 = cz[:lo].contiguous(), cz[lo:].contiguous()
        torch.cuda.synchronize()
        assert e0.dev_vandermonde_apply_strided(c0.data_ptr(), lo, n, d, s0.data_ptr(), lo + 3) == 0
        assert e1.dev_compute_shares(c1.data_ptr(), B - lo, n, d, s1.data_ptr()) == 0
        rc = pkg.Engine.gather_party_major([e0, e1], 0, [s0.data_ptr(), s1.data_ptr()], [lo, B - lo], [lo + 3, B - lo], n,
                                           out.data_ptr(), B + 5)
        assert rc == 0, e0.last_error()
        e0.sync()
        got = out.cpu().numpy().view(np.uint64)
        assert np.array_equal(got[:, :B], want) and (got[:, B:] == np.uint64(0xFFFFFFFFFFFFFFFF)).all()
        # the branch of device pairs WITHOUT peer access (one peer copy per row), forced on this one-device box: same bytes,
        # output pitch (B + 5) and source pitch (lo + 3) both different from the copied width
        import ctypes as C
        out.fill_(-1)
        torch.cuda.synchronize()
        assert e0.L.hbmpc_set_gather_row_copies(e0.ctx, C.c_int(1)) == 0
        rc = pkg.Engine.gather_party_major([e0, e1], 0, [s0.data_ptr(), s1.data_ptr()], [lo, B - lo], [lo + 3, B - lo], n,
                                           out.data_ptr(), B + 5)
        assert rc == 0, e0.last_error()
        e0.sync()
        got2 = out.cpu().numpy().view(np.uint64)
        assert np.array_equal(got2, got)
        assert e0.L.hbmpc_set_gather_row_copies(e0.ctx, C.c_int(0)) == 0
        assert e0.peer_access(e1) is True  # same device: direct (on a multi-GPU node: peer access over xGMI, enabled by the call)
        # argument checks: stride below count, field mismatch
        assert pkg.Engine.gather_party_major([e0, e1], 0, [s0.data_ptr(), s1.data_ptr()], [lo, B - lo], [lo - 1, B - lo], n,
                                             out.data_ptr(), B + 5) == 4
        g = pkg.Engine(0, field="goldilocks")
        assert pkg.Engine.gather_party_major([e0, g], 0, [s0.data_ptr(), s1.data_ptr()], [lo, B - lo], [lo + 3, B - lo], n,
                                             out.data_ptr(), B + 5) == 5
        g.close()
    finally:
        e0.close()
        e1.close()


ENC_SHAPES = [(31, 10), (20, 6), (17, 1), (33, 5), (64, 14), (100, 9), (255, 14), (31, 13), (40, 3)]


@pytest.mark.parametrize("n,d", ENC_SHAPES)
@pytest.mark.parametrize("wgs", [0, 8])
def test_encode_vs_oracle(eng, n, d, wgs):
    """apply_vandermonde / compute_shares on the matrix cores (domains beyond 16 points, 2 <= d + 1 <= 15): against the
    oracle and against the FFT kernels, edge polynomials included; wgs = 8 walks both input register sets"""
    G = 1200 + 41 if wgs == 0 else 7000 + 3
    eng.set_matrix_core_workgroups(wgs)
    try:
        x = rnd(50 + n + d, G, d + 1)
        x[0] = 0
        x[1] = O.ints_to_u256([O_R - 1] * (d + 1))
        x[2, :, :] = 0
        x[2, d, 0] = 1
        eng.set_matrix_cores(True, 1)
        rc, y = eng.vandermonde_apply(x, n, d)
        rc2, y2 = eng.compute_shares(x, n, d)
        eng.set_matrix_cores(False)
        rc1, y1 = eng.vandermonde_apply(x, n, d)
        eng.set_matrix_cores(True, 65536)
        rc0, y0 = O.vandermonde_apply(x, n, d)
        assert rc == rc0 == rc1 == rc2 == 0 and np.array_equal(y, y0) and np.array_equal(y1, y0) and np.array_equal(y2, y0)
    finally:
        eng.set_matrix_core_workgroups(0)
        eng.set_matrix_cores(True, 65536)


def test_full_size_config3_encode(eng):
    """BASELINE configs[2]'s encode: x[2^20][11] -> y[31][2^20], matrix cores against the FFT kernels (whole array) and
    against the oracle (sampled chunks); strided output rows (the in-place wire path) as well"""
    import torch
    n, d, G = 31, 10, 1 << 20
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0xC0FFEE06)
    lo = torch.randint(0, 1 << 62, (G, d + 1, 3), dtype=torch.int64, device=dev, generator=gen)
    hi = torch.randint(0, 0x73EDA753299D7D48, (G, d + 1, 1), dtype=torch.int64, device=dev, generator=gen)
    x = torch.cat([lo, hi], dim=-1).contiguous()
    ys = {}
    for mode in ("mfma", "fft"):
        eng.set_matrix_cores(mode == "mfma", 65536)
        y = torch.full((n, G, 4), -1, dtype=torch.int64, device=dev)
        ystr = torch.full((n, G + 8, 4), -1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        assert eng.dev_vandermonde_apply(x.data_ptr(), G, n, d, y.data_ptr(), 0) == 0
        assert eng.dev_vandermonde_apply_strided(x.data_ptr(), G, n, d, ystr.data_ptr(), G + 8, 0) == 0
        eng.sync()
        ys[mode] = (y, ystr)
    eng.set_matrix_cores(True, 65536)
    assert torch.equal(ys["mfma"][0], ys["fft"][0]) and torch.equal(ys["mfma"][1], ys["fft"][1])
    assert torch.equal(ys["mfma"][1][:, :G], ys["mfma"][0]) and bool((ys["mfma"][1][:, G:] == -1).all())
    idx = np.unique(np.concatenate([np.arange(0, 70), np.arange(G - 70, G), np.arange(31, G, 32)[:40], np.arange(32 * 3072 - 3, 32 * 3072 + 3)]))
    ti = torch.as_tensor(idx, device=dev)
    rc0, y0 = O.vandermonde_apply(np.ascontiguousarray(x[ti].cpu().numpy().view(np.uint64)), n, d)
    assert rc0 == 0 and np.array_equal(ys["mfma"][0][:, ti].cpu().numpy().view(np.uint64), y0)


def test_mid_size_decode_takes_the_matrix_cores_at_the_first_sight_of_a_sender_set():
    """default thresholds (hbmpc_set_matrix_cores doc): a 4 096 .. 65 535-chunk decode takes the matrix cores the FIRST time
    a sender set is seen -- its byte-digit table is expanded on the device (kernels_tables.hpp), so there is nothing to wait a
    second sighting for (round 2 built the table on the host and did) -- and a repeat builds nothing; a >= 4 096-chunk encode on
    a 32-point domain takes the matrix cores at once.  Same bytes whichever kernel ran."""
    e = load_package().Engine(0)
    try:
        n, d, t, G = 20, 6, 6, 20000
        x, y = codewords(77, G, n, d)
        rc, ye = e.vandermonde_apply(x, n, d)
        assert rc == 0 and np.array_equal(ye, y)
        e.set_matrix_cores(False)
        rc, yl = e.vandermonde_apply(x, n, d)
        assert rc == 0 and np.array_equal(yl, y)
        y[3, 11, 0] ^= 1                      # one flagged chunk: the fallback kernels run behind either path
        want = O.batch_recover(list(range(n)), y, n, d, t)
        ids = [5, 0, 19, 3, 7, 1, 12, 2, 9, 4, 15, 6, 8, 10, 11, 13, 14, 16, 17, 18]   # an arrival order
        perm = y[ids]
        want_p = O.batch_recover(ids, perm, n, d, t)
        # the lane kernels' tables of this set first (matrix cores off), so that the count below isolates the matrix-core table
        got = e.batch_recover(ids, perm, n, d, t)
        assert got[0] == want_p[0] and all(np.array_equal(u, v) for u, v in zip(got[1:], want_p[1:]))
        e.set_matrix_cores(True)
        tables = [e.cache_stats()["tables"]]
        for k in range(3):
            got = e.batch_recover(ids, perm, n, d, t)
            assert got[0] == want_p[0] and all(np.array_equal(u, v) for u, v in zip(got[1:], want_p[1:])), k
            tables.append(e.cache_stats()["tables"])
        # first call with the matrix cores on: + the matrix-core table; calls 2, 3: nothing new
        assert tables[1] == tables[0] + 1 and tables[3] == tables[2] == tables[1], tables
        # another sender set (sender 0 missing): everything at its first call, nothing at its second
        ids2 = list(range(1, n))
        want2 = O.batch_recover(ids2, y[1:], n, d, t)
        got = e.batch_recover(ids2, y[1:], n, d, t)
        assert got[0] == want2[0] and all(np.array_equal(u, v) for u, v in zip(got[1:], want2[1:]))
        before = e.cache_stats()["tables"]
        got = e.batch_recover(ids2, y[1:], n, d, t)
        assert all(np.array_equal(u, v) for u, v in zip(got[1:], want2[1:]))
        assert e.cache_stats()["tables"] == before
        # the same set in another arrival order is the same table
        got = e.batch_recover(list(range(n)), y, n, d, t)
        assert got[0] == want[0] and all(np.array_equal(u, v) for u, v in zip(got[1:], want[1:]))
        assert e.cache_stats()["tables"] == before
    finally:
        e.close()


@pytest.mark.parametrize("field", ["fr", "goldilocks"])
def test_party_batched_encode_on_the_matrix_cores(field):
    """hbmpc_[gl_]dev_vandermonde_apply_parties on a 32-point domain from 4 096 chunks on: one matrix-core launch per party;
    every party against the single-party call and party 0 against the oracle"""
    import torch
    e = load_package().Engine(0, field=field)
    try:
        n, d, G, parties = 31, 10, 5000 + 13, 3
        dev = torch.device("cuda", 0)
        if field == "fr":
            xh = rnd(91, parties, G, d + 1)
            width = (4,)
        else:
            from oracle.spec_gl import P
            rng = np.random.default_rng(5)
            xh = (rng.integers(0, 1 << 63, (parties, G, d + 1), dtype=np.uint64) % np.uint64(P)).astype(np.uint64)
            width = ()
        x = torch.from_numpy(xh.view(np.int64)).to(dev)
        y = torch.full((parties, n, G) + width, -1, dtype=torch.int64, device=dev)
        assert e.dev_vandermonde_apply_parties(x.data_ptr(), G, n, d, parties, y.data_ptr()) == 0
        e.sync()
        got = y.cpu().numpy().view(np.uint64)
        e.set_matrix_cores(False)
        for p in range(parties):
            rc, want = e.vandermonde_apply(xh[p], n, d)     # the FFT kernels
            assert rc == 0 and np.array_equal(got[p], want), p
        if field == "fr":
            rc, want0 = O.vandermonde_apply(xh[0], n, d)
            assert rc == 0 and np.array_equal(got[0], want0)
    finally:
        e.close()


@pytest.mark.parametrize("n,d,G", [(16, 5, 5000 + 3), (16, 10, 9000 + 1), (7, 2, 3000 + 7), (10, 3, 2049), (16, 5, 16384), (13, 4, 4096)])
def test_mid_size_encode_on_small_domains(n, d, G):
    """default thresholds: on domains up to 16 points a 2 049 .. 16 384-chunk encode (compute_shares, apply_vandermonde) takes
    the workgroup-per-tile matrix-core kernel instead of the single-pass FFT; same bytes as the FFT kernels and the oracle"""
    e = load_package().Engine(0)
    try:
        x = rnd(300 + n + d, G, d + 1)
        x[0] = 0
        x[1] = O.ints_to_u256([O_R - 1] * (d + 1))
        rc, y = e.vandermonde_apply(x, n, d)
        rc2, y2 = e.compute_shares(x, n, d)
        e.set_matrix_cores(False)
        rc3, y3 = e.vandermonde_apply(x, n, d)
        rc0, want = O.vandermonde_apply(x, n, d)
        assert rc == rc2 == rc3 == rc0 == 0
        assert np.array_equal(y, want) and np.array_equal(y2, want) and np.array_equal(y3, want)
    finally:
        e.close()
