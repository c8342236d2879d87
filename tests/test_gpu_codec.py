"""Wire codec on the device against the ark-serialize 'compressed' byte
layout restated with numpy: Vec<F> = u64-LE len + 32-byte LE elements; Vec<RobustShare> = u64-LE len +
(32-byte value | u64 id | u64 degree) records (Appendix B)."""
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


def ark_vec_f(row_u256):  # Vec<F>::serialize_compressed
    return np.uint64(row_u256.shape[0]).tobytes() + np.ascontiguousarray(row_u256).tobytes()


def ark_vec_shares(vals, sid, deg):  # Vec<RobustShare<F>>::serialize_compressed
    rec = np.zeros((vals.shape[0], 6), dtype=np.uint64)
    rec[:, :4], rec[:, 4], rec[:, 5] = vals, sid, deg
    return np.uint64(vals.shape[0]).tobytes() + rec.tobytes()


def test_evalbatch_payloads_from_encode(eng):
    # encode on the device, then the n EvalBatch payloads straight from the party-major rows
    n, d, G = 7, 4, 333
    x = O.fill_random(5, G * (d + 1)).reshape(G, d + 1, 4)
    xd, yd = eng.dev_alloc(x.nbytes), eng.dev_alloc(n * G * 32)
    stride = 8 + 32 * G + 24  # padded payload stride
    pd = eng.dev_alloc(n * stride)
    eng.h2d(xd, x)
    assert eng.dev_vandermonde_apply(xd, G, n, d, yd) == 0
    assert eng.dev_pack_fvec(yd, G, G, n, pd, stride) == 0
    raw = np.zeros(n * stride, dtype=np.uint8)
    eng.d2h(raw, pd)
    eng.sync()
    rc, y = O.vandermonde_apply(x, n, d)
    for j in range(n):
        assert raw[j * stride: j * stride + 8 + 32 * G].tobytes() == ark_vec_f(y[j])
    # and back (the RevealBatch / EvalBatch receive side), with validation
    rd, sd = eng.dev_alloc(n * G * 32), eng.dev_alloc(4 * n)
    assert eng.dev_unpack_fvec(pd, stride, 8 + 32 * G, G, n, rd, G, sd) == 0
    back, st = np.zeros((n, G, 4), dtype=np.uint64), np.zeros(n, dtype=np.uint32)
    eng.d2h(back, rd)
    eng.d2h(st, sd)
    eng.sync()
    assert np.array_equal(back, y) and not st.any()
    # a wrong length prefix and a non-canonical element are InvalidData in ark -> status 4 for that sender only
    bad = raw.copy().view(np.uint64).reshape(n, stride // 8)
    bad[2, 0] = G + 1
    bad[4, 1 + 4 * 17: 1 + 4 * 17 + 4] = O.ints_to_u256(R)  # element == r
    eng.h2d(pd, bad)
    assert eng.dev_unpack_fvec(pd, stride, 8 + 32 * G, G, n, rd, G, sd) == 0
    eng.d2h(st, sd)
    eng.sync()
    assert list(st) == [0, 0, 4, 0, 4, 0, 0]
    assert eng.dev_unpack_fvec(pd, stride, 8 + 32 * G - 1, G, n, rd, G, sd) == 4  # truncated payload
    for p in (xd, yd, pd, rd, sd):
        eng.dev_free(p)


@pytest.mark.parametrize("n,t,d,G", [(7, 2, 2, 333), (16, 5, 5, 3000), (31, 10, 10, 70)])
def test_in_place_wire_path(eng, n, t, d, G):
    """encode straight into the payloads (no pack pass) and decode straight out of them (validate, no unpack pass):
    the bytes on the wire are ark's, and the receive side recovers the polynomials -- also with a lying sender and
    with a sender whose payload fails deserialisation (dropped from the sender set, as the reference would)."""
    x = O.fill_random(6, G * (d + 1)).reshape(G, d + 1, 4)
    stride = 32 * (G + 2)                                    # a multiple of 32, >= 8 + 32 G
    buf = eng.dev_alloc(n * stride + 64)
    base = (buf + 31) // 32 * 32                             # 32-byte boundary ...
    pd = base + 24                                           # ... and the payloads start 8 bytes before one
    xd, sd = eng.dev_alloc(x.nbytes), eng.dev_alloc(4 * n)
    eng.h2d(xd, x)
    assert eng.dev_encode_fvec(xd, G, n, d, pd, stride) == 0
    raw = np.zeros(n * stride, dtype=np.uint8)
    eng.d2h(raw, pd)
    eng.sync()
    rc, y = O.vandermonde_apply(x, n, d)
    for j in range(n):
        assert raw[j * stride: j * stride + 8 + 32 * G].tobytes() == ark_vec_f(y[j])
    assert eng.dev_encode_fvec(xd, G, n, d, pd + 8, stride) == 4          # misaligned payload base
    assert eng.dev_encode_fvec(xd, G, n, d, pd, 8 + 32 * G) == 4          # stride not a multiple of 32
    # receive side: sender 1 lies in chunk 5, sender 3's payload is not deserialisable (an element == r)
    w = raw.copy().view(np.uint64).reshape(n, stride // 8)
    w[1, 1 + 4 * 5] ^= np.uint64(1)
    w[3, 1 + 4 * 9: 1 + 4 * 9 + 4] = O.ints_to_u256(R)
    eng.h2d(pd, w)
    assert eng.dev_validate_fvec(pd, stride, 8 + 32 * G, G, n, sd) == 0
    st = np.zeros(n, dtype=np.uint32)
    eng.d2h(st, sd)
    eng.sync()
    assert list(st) == [4 if j == 3 else 0 for j in range(n)]
    senders = [j for j in range(n) if st[j] == 0]
    co_d, nco_d, stat_d = eng.dev_alloc(G * (d + 1) * 32), eng.dev_alloc(4 * G), eng.dev_alloc(G)
    # decode in place: payload j's body sits at pd + 8 + j * stride; the dropped sender is simply not listed, the others
    # in a scrambled arrival order with their slots
    order = senders[::-1]
    assert eng.dev_batch_recover_slots(order, order, pd + 8, stride // 32, G, n, d, t, co_d, nco_d=nco_d, status_d=stat_d) == 0
    co, stat = np.zeros((G, d + 1, 4), dtype=np.uint64), np.zeros(G, dtype=np.uint8)
    eng.d2h(co, co_d)
    eng.d2h(stat, stat_d)
    eng.sync()
    assert np.array_equal(co, x) and stat[5] == 1 and stat.sum() == 1
    for p in (buf, xd, sd, co_d, nco_d, stat_d):
        eng.dev_free(p)


def test_share_records(eng):
    N, sid, deg = 257, 3, 5
    vals = O.fill_random(9, N)
    vd, pd, od, sd = eng.dev_alloc(N * 32), eng.dev_alloc(8 + 48 * N), eng.dev_alloc(N * 32), eng.dev_alloc(4)
    eng.h2d(vd, vals)
    assert eng.dev_pack_shares(vd, N, sid, deg, pd) == 0
    raw = np.zeros(8 + 48 * N, dtype=np.uint8)
    eng.d2h(raw, pd)
    eng.sync()
    assert raw.tobytes() == ark_vec_shares(vals, sid, deg)
    st = np.zeros(1, dtype=np.uint32)
    out = np.zeros((N, 4), dtype=np.uint64)
    for want_id, want_deg, code in ((sid, deg, 0), (sid + 1, deg, 3), (sid, deg + 1, 2)):
        assert eng.dev_unpack_shares(pd, 8 + 48 * N, N, want_id, want_deg, od, sd) == 0
        eng.d2h(st, sd)
        eng.d2h(out, od)
        eng.sync()
        assert st[0] == code and np.array_equal(out, vals)
    # canonical validation (what Fr::from_bigint(..).unwrap() enforces at the reference's C boundary)
    assert eng.dev_validate_canonical(vd, N, sd) == 0
    eng.d2h(st, sd)
    eng.sync()
    assert st[0] == 0
    vals[100] = O.ints_to_u256(R + 5)
    eng.h2d(vd, vals)
    assert eng.dev_validate_canonical(vd, N, sd) == 0
    eng.d2h(st, sd)
    eng.sync()
    assert st[0] == 4
    for p in (vd, pd, od, sd):
        eng.dev_free(p)
