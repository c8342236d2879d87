"""A decode that is given exactly d + t + 1 senders -- what BatchRecon passes (batch_recon.rs:371-389) -- has no OEC round:
the library runs it as ONE launch whose first kernel writes the failures itself (hbmpc_set_single_launch_decode).  Every
kernel family against the oracle and against the two-launch sequence: status, coefficients, lengths, the device summary,
and the state the call leaves behind (the next call on the stream starts from zeroed counters)."""
import ctypes as C

import numpy as np
import pytest

from __graft_entry__ import load_package
from oracle import cref as O

pytestmark = pytest.mark.gpu
DECODING_ERROR = 8   # ShareErrorCode::DecodingError (include/hbmpc_hip.h, ffi/c_bindings/share/mod.rs:18-37)


def rnd(seed, *shape):
    return O.fill_random(seed, int(np.prod(shape))).reshape(*shape, 4)


def make(seed, G, n, d, t, bad_chunks):
    x = rnd(seed, G, d + 1)
    rc, y = O.vandermonde_apply(x, n, d)
    assert rc == 0
    needed = d + t + 1
    ids = list(range(n))[-needed:]            # the HIGHEST ids: not the identity prefix of any table
    ev = np.ascontiguousarray(y[ids])
    for k, g in enumerate(bad_chunks):        # a verify row, an interpolation row, both
        ev[(d + 1 + k) % needed if t else 0, g, k % 4] ^= np.uint64(1 + k)
        if k % 3 == 2:
            ev[0, g, 0] ^= np.uint64(5)
    return ids, ev


def dev_decode(e, ids, ev, n, d, t, p0):
    G = ev.shape[1]
    ow = 1 if p0 else d + 1
    dev_ev = e.dev_alloc(ev.nbytes)
    dev_out, dev_st, dev_nc, dev_su = e.dev_alloc(G * ow * 32), e.dev_alloc(G), e.dev_alloc(G * 4), e.dev_alloc(64)
    try:
        e.h2d(dev_ev, ev)
        junk = np.full(G * ow * 4, 0xEEEEEEEEEEEEEEEE, dtype=np.uint64)
        e.h2d(dev_out, junk)
        rc = e.dev_batch_recover(ids, dev_ev, G, n, d, t, dev_out, 0 if p0 else dev_nc, dev_st, dev_su, 0, p0=p0)
        assert rc == 0, e.last_error()
        out = np.zeros((G, ow, 4), dtype=np.uint64)
        st = np.zeros(G, dtype=np.uint8)
        nc = np.zeros(G, dtype=np.uint32)
        su = np.zeros(4, dtype=np.uint32)
        e.d2h(out, dev_out)
        e.d2h(st, dev_st)
        if not p0:
            e.d2h(nc, dev_nc)
        e.d2h(su, dev_su)
        e.sync()
        return out, st, nc, su
    finally:
        for p in (dev_ev, dev_out, dev_st, dev_nc, dev_su):
            e.dev_free(p)


CASES = [  # (family, n, d, t, G): wave-per-chunk, lane-per-chunk (compile-time and run-time shapes), matrix cores
    ("wide", 16, 5, 5, 700), ("wide", 31, 10, 10, 130), ("wide", 7, 2, 2, 64), ("wide", 4, 1, 0, 50),
    ("lane", 16, 5, 5, 9000 + 13), ("lane", 31, 10, 10, 8192 + 300), ("lane", 10, 3, 3, 20000), ("lane", 40, 17, 10, 9001),
    ("generic", 16, 5, 5, 9000 + 13),
    ("mfma", 16, 5, 5, 9000 + 13), ("mfma", 31, 10, 10, 5000 + 7), ("mfma", 16, 10, 5, 70001), ("mfma", 13, 4, 4, 777),
]


@pytest.mark.parametrize("family,n,d,t,G", CASES)
@pytest.mark.parametrize("p0", [False, True])
def test_one_launch_equals_oracle_and_two_launches(family, n, d, t, G, p0):
    e = load_package().Engine(0)
    try:
        if family == "generic":
            e.set_force_generic(True)
        if family == "mfma":
            e.set_small_batch_chunks(0)
            e.set_matrix_cores(True, 1)
        else:
            e.set_matrix_cores(False)
        bad = sorted({g for g in (0, 1, 63, 64, G // 2, G - 1) if g < G} if t else set())
        ids, ev = make(1000 + n + d, G, n, d, t, bad)
        rc0, co0, nc0, st0 = O.batch_recover(ids, ev, n, d, t)
        if bad:
            assert rc0 == DECODING_ERROR and all(st0[g] == DECODING_ERROR for g in bad) and int((st0 != 0).sum()) == len(bad)
        want_out = co0[:, :1] if p0 else co0
        res = {}
        for on in (True, False):
            e.set_single_launch_decode(on)
            # twice: the second call finds whatever the first left in the stream's counters
            for rep in range(2):
                out, st, nc, su = dev_decode(e, ids, ev, n, d, t, p0)
                assert np.array_equal(st, st0), (on, rep)
                assert np.array_equal(out, want_out), (on, rep)
                if not p0:
                    assert np.array_equal(nc, nc0), (on, rep)
                nb = len(bad)
                assert list(su) == [nb, nb, bad[0] if nb else 0xFFFFFFFF, DECODING_ERROR if nb else 0], (on, rep, list(su))
            res[on] = (out, st, nc)
        assert all(np.array_equal(u, v) for u, v in zip(res[True], res[False]))
        # a clean batch afterwards on the same stream, with all n senders (OEC rounds exist again)
        x = rnd(5, 300, d + 1)
        rc, y = O.vandermonde_apply(x, n, d)
        got = e.batch_recover(list(range(n)), y, n, d, t)
        assert got[0] == 0 and np.array_equal(got[1], x) and not got[3].any()
    finally:
        e.close()


@pytest.mark.parametrize("family,n,d,t,G", [c for c in CASES if c[3] > 0])
def test_single_coefficient_decode(family, n, d, t, G):
    """hbmpc_dev_batch_recover_coeff_strided: the P(0)-shaped decode that keeps coefficient k instead of coefficient 0 (RanSha's
    verifiers need the top one) against column k of the oracle's full decode, for every k and every kernel family; chunks that
    fail the verification give zero and count; with OEC rounds available (more than d + t + 1 senders) the call is refused."""
    e = load_package().Engine(0)
    try:
        if family == "generic":
            e.set_force_generic(True)
        if family == "mfma":
            e.set_small_batch_chunks(0)
            e.set_matrix_cores(True, 1)
        else:
            e.set_matrix_cores(False)
        bad = sorted({g for g in (0, 1, 63, 64, G // 2, G - 1) if g < G})
        ids, ev = make(2000 + n + d, G, n, d, t, bad)
        rc0, co0, nc0, st0 = O.batch_recover(ids, ev, n, d, t)
        dev_ev, dev_out, dev_st, dev_su = e.dev_alloc(ev.nbytes), e.dev_alloc(G * 32), e.dev_alloc(G), e.dev_alloc(64)
        e.h2d(dev_ev, ev)
        for k in sorted({0, 1, d // 2, d - 1, d}):
            e.h2d(dev_out, np.full(G * 4, 0xEEEEEEEEEEEEEEEE, dtype=np.uint64))
            assert e.dev_batch_recover_coeff_strided(ids, dev_ev, G, G, n, d, t, k, dev_out, dev_st, dev_su) == 0, e.last_error()
            out, st, su = np.zeros((G, 4), dtype=np.uint64), np.zeros(G, dtype=np.uint8), np.zeros(4, dtype=np.uint32)
            e.d2h(out, dev_out), e.d2h(st, dev_st), e.d2h(su, dev_su)
            e.sync()
            assert np.array_equal(st, st0), k
            assert np.array_equal(out, co0[:, k]), k
            assert list(su) == [len(bad), len(bad), bad[0], DECODING_ERROR], (k, list(su))
        assert e.dev_batch_recover_coeff_strided(ids, dev_ev, G, G, n, d, t, d + 1, dev_out, dev_st, dev_su) == 4          # k > d
        if len(ids) < n:
            more = [i for i in range(n) if i not in ids][:1] + ids
            assert e.dev_batch_recover_coeff_strided(more, dev_ev, G, G, n, d, t, d, dev_out, dev_st, dev_su) == 4        # an OEC round exists
        for p in (dev_ev, dev_out, dev_st, dev_su):
            e.dev_free(p)
    finally:
        e.close()


@pytest.mark.parametrize("n,d,G", [(16, 5, 300), (16, 10, 300), (16, 5, 20000), (16, 10, 20000), (7, 2, 1000), (13, 8, 70001), (4, 1, 64), (31, 10, 9000),
                                   (20, 17, 500), (20, 17, 9000)])
@pytest.mark.parametrize("fusion", [1, 0])
def test_interpolate_degree_check(n, d, G, fusion):
    """hbmpc_dev_interpolate_degree_check_strided (RanDouSha's verifier: the degree and the constant term of the polynomial through ALL
    n shares, ran_dou_sha/mod.rs:557-602) against the oracle's full interpolation: points of a degree-d polynomial give status 0 and
    (c_0, c_d); a polynomial of lower degree gives c_d = 0; one point moved off gives DecodingError and zeros -- on the wave-per-chunk
    and matrix-core kernels (the selective decode) and, with hbmpc_set_producer_fusion(0), through the full interpolation."""
    e = load_package().Engine(0)
    try:
        assert e.L.hbmpc_set_producer_fusion(e.ctx, C.c_int(fusion)) == 0
        x = rnd(4000 + n + d, G, d + 1)
        low, off = [1, G // 3], [0, 2, G // 2, G - 1]
        for g in low:
            x[g, d] = 0                                              # degree below d
        rc, y = O.vandermonde_apply(x, n, d)
        assert rc == 0
        ids = list(range(n))
        ids = ids[1::2] + ids[0::2]                                   # an arrival order that is not sorted
        ev = np.ascontiguousarray(y[ids])
        for k, g in enumerate(off):
            ev[(d + 1 + k) % n if d + 1 < n else 0, g, 0] ^= np.uint64(1 + k)
        if d + 1 == n:
            off = []                                                  # every set of n points lies on a polynomial of degree <= n - 1
            ev = np.ascontiguousarray(y[ids])
        dev_ev, dev_ws, dev_sel, dev_st = e.dev_alloc(ev.nbytes), e.dev_alloc(G * n * 32), e.dev_alloc(G * 64), e.dev_alloc(G)
        e.h2d(dev_ev, ev)
        e.h2d(dev_sel, np.full(G * 8, 0xEEEEEEEEEEEEEEEE, dtype=np.uint64))
        assert e.dev_interpolate_degree_check_strided(ids, dev_ev, G, G, n, d, dev_ws, dev_sel, dev_st) == 0, e.last_error()
        sel, st = np.zeros((G, 2, 4), dtype=np.uint64), np.zeros(G, dtype=np.uint8)
        e.d2h(sel, dev_sel), e.d2h(st, dev_st)
        e.sync()
        want_st = np.zeros(G, dtype=np.uint8)
        want_st[off] = DECODING_ERROR
        assert np.array_equal(st, want_st)
        want = np.stack([x[:, 0], x[:, d]], axis=1)
        want[off] = 0
        assert np.array_equal(sel, want)
        assert not sel[low, 1].any()
        for p in (dev_ev, dev_ws, dev_sel, dev_st):
            e.dev_free(p)
    finally:
        e.close()


@pytest.mark.parametrize("G", [5000, 50000])   # workgroup per tile / wave per tile on the matrix cores
def test_every_chunk_fails(G):
    """the failure path is as parallel as the decode: a batch in which EVERY chunk fails"""
    e = load_package().Engine(0)
    try:
        n, d, t = 16, 5, 5
        ids, ev = make(9, G, n, d, t, [])
        ev[d + 2] ^= np.uint64(1)              # one verify row wrong everywhere
        for mf in (False, True):
            e.set_matrix_cores(mf, 1)
            out, st, nc, su = dev_decode(e, ids, ev, n, d, t, False)
            assert (st == DECODING_ERROR).all() and not out.any() and not nc.any()
            assert list(su) == [G, G, 0, DECODING_ERROR]
    finally:
        e.close()


@pytest.mark.parametrize("family,n,d,t,G", [("wide", 16, 5, 5, 300), ("lane", 16, 5, 5, 9001), ("mfma", 16, 5, 5, 9001),
                                            ("mfma", 31, 10, 10, 5003), ("mfma", 16, 10, 5, 40001), ("mfma", 7, 2, 2, 64)])
@pytest.mark.parametrize("p0", [False, True])
def test_goldilocks_one_launch(family, n, d, t, G, p0):
    """the same over the small field (hbmpc_gl_*): good chunks return the polynomial that was shared, failing chunks fail
    the way the oracle's oec_decode does with no round to run, and both launch sequences agree byte for byte"""
    import random
    from oracle.spec_gl import P, S
    e = load_package().Engine(0, field="goldilocks")
    try:
        if family == "mfma":
            e.set_small_batch_chunks(0)
            e.set_matrix_cores(True, 1)
        else:
            e.set_matrix_cores(False)
        rng = random.Random(n * 100 + d)
        x = np.array([rng.randrange(P) for _ in range(G * (d + 1))], dtype=np.uint64).reshape(G, d + 1)
        rc, y = e.vandermonde_apply(x, n, d)
        assert rc == 0
        needed = d + t + 1
        ids = list(range(n))[-needed:]
        ev = np.ascontiguousarray(y[ids])
        bad = sorted({g for g in (0, 1, 31, 32, G // 2, G - 1) if g < G}) if t else []
        for k, g in enumerate(bad):
            ev[(d + 1 + k) % needed, g] ^= np.uint64(1 + k)
        # the oracle on the chunks that were touched: DecodingError with exactly d + t + 1 shares and one of them wrong
        for g in bad[:3]:
            with pytest.raises(S.ShareErr) as err:
                S.recover_secret([S.Share(int(ev[i][g]), ids[i], d) for i in range(needed)], n, t)
            assert err.value.code == DECODING_ERROR
        ow = 1 if p0 else d + 1
        want = (x[:, :1] if p0 else x).copy()
        want[bad] = 0
        res = {}
        for on in (True, False):
            e.set_single_launch_decode(on)
            for rep in range(2):
                dev_ev, dev_out, dev_st, dev_nc, dev_su = (e.dev_alloc(ev.nbytes), e.dev_alloc(G * ow * 8), e.dev_alloc(G), e.dev_alloc(G * 4),
                                                           e.dev_alloc(64))
                try:
                    e.h2d(dev_ev, ev)
                    e.h2d(dev_out, np.full(G * ow, 0xEEEEEEEEEEEEEEEE, dtype=np.uint64))
                    rc = e.dev_batch_recover(ids, dev_ev, G, n, d, t, dev_out, 0 if p0 else dev_nc, dev_st, dev_su, 0, p0=p0)
                    assert rc == 0, e.last_error()
                    out, st, nc, su = np.zeros((G, ow), dtype=np.uint64), np.zeros(G, dtype=np.uint8), np.zeros(G, dtype=np.uint32), np.zeros(4, dtype=np.uint32)
                    e.d2h(out, dev_out)
                    e.d2h(st, dev_st)
                    if not p0:
                        e.d2h(nc, dev_nc)
                    e.d2h(su, dev_su)
                    e.sync()
                finally:
                    for p in (dev_ev, dev_out, dev_st, dev_nc, dev_su):
                        e.dev_free(p)
                assert np.array_equal(out, want), (on, rep)
                assert [int(v) for v in np.nonzero(st)[0]] == bad and all(st[g] == DECODING_ERROR for g in bad), (on, rep)
                if not p0:
                    assert all(nc[g] == 0 for g in bad) and int((nc != d + 1).sum()) == len(bad)
                nb = len(bad)
                assert list(su) == [nb, nb, bad[0] if nb else 0xFFFFFFFF, DECODING_ERROR if nb else 0], (on, rep, list(su))
            res[on] = (out, st, nc)
        assert all(np.array_equal(u, v) for u, v in zip(res[True], res[False]))
    finally:
        e.close()
