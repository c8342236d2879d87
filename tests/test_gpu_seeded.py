"""GPU parity tests of seeded compute_shares (hbmpc_[gl_][dev_]compute_shares_seeded, hbmpc_[gl_]dev_fill_coeffs):
the coefficients drawn on the device must equal oracle/spec.py::seeded_polynomial ("hbmpc-chacha20-v1") and the
shares must equal compute_shares of exactly those polynomials.  Both fields, through the C ABI, bit-exact."""
import random

import numpy as np
import pytest
import torch

from __graft_entry__ import load_package
from oracle import spec as SFR
from oracle import spec_gl
from oracle.cref import ints_to_u256, u256_to_ints

pytestmark = pytest.mark.gpu

FIELDS = {"fr": (SFR, SFR.R_MOD), "goldilocks": (spec_gl.S, spec_gl.P)}


@pytest.fixture(scope="module", params=["fr", "goldilocks"])
def env(request):
    e = load_package().Engine(0, field=request.param)
    yield e, request.param
    e.close()


def to_arr(field, ints, shape):
    if field == "fr":
        return ints_to_u256(ints).reshape(tuple(shape) + (4,))
    return np.array(ints, dtype=np.uint64).reshape(shape)


def to_ints(field, arr):
    if field == "fr":
        return u256_to_ints(arr.reshape(-1, 4))
    return [int(x) for x in arr.reshape(-1)]


def words(seed):
    return [int.from_bytes(seed[4 * i:4 * i + 4], "little") for i in range(8)]


@pytest.mark.parametrize("n,d,B,first", [(16, 5, 300, 0), (7, 2, 65, 1 << 40), (31, 10, 130, (1 << 64) - 200), (4, 0, 9, 3)])
def test_host_call_matches_oracle(env, n, d, B, first):
    eng, field = env
    sp, mod = FIELDS[field]
    rng = random.Random(n * 1000 + d)
    seed = bytes(rng.randrange(256) for _ in range(32))
    secrets = [rng.randrange(mod) for _ in range(B)]
    rc, sh = eng.compute_shares_seeded(seed, to_arr(field, secrets, (B,)), n, d, first_index=first)
    assert rc == 0
    polys = [sp.seeded_polynomial(words(seed), (first + b) % (1 << 64), secrets[b], d) for b in range(B)]
    got = to_ints(field, sh)
    for b in range(B):
        ys = [s.v for s in sp.compute_shares(polys[b], n, d)]
        assert [got[j * B + b] for j in range(n)] == ys, (field, b)


def test_device_calls_and_workspace(env):
    eng, field = env
    sp, mod = FIELDS[field]
    n, d, B, first = 16, 5, 2000, 77
    ew = 4 if field == "fr" else 1
    seed = bytes(range(32))
    rng = random.Random(5)
    secrets = [rng.randrange(mod) for _ in range(B)]
    sec_h = to_arr(field, secrets, (B,))
    dev = torch.device("cuda:0")
    sec = torch.from_numpy(sec_h.view(np.int64)).to(dev)
    co = torch.zeros(B * (d + 1) * ew, dtype=torch.int64, device=dev)
    out = torch.zeros(n * B * ew, dtype=torch.int64, device=dev)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        assert eng.dev_compute_shares_seeded(seed, sec.data_ptr(), B, first, n, d, co.data_ptr(), out.data_ptr(),
                                             stream=st.cuda_stream) == 0
    st.synchronize()
    co_h = co.cpu().numpy().view(np.uint64)
    got = to_ints(field, co_h)
    for b in range(0, B, 37):
        assert got[b * (d + 1):(b + 1) * (d + 1)] == sp.seeded_polynomial(words(seed), first + b, secrets[b], d)
    # the shares are compute_shares of the workspace rows (ties the seeded call to the golden-pinned one)
    rc, want = eng.compute_shares(co_h.reshape((B, d + 1, 4) if field == "fr" else (B, d + 1)), n, d)
    assert rc == 0 and np.array_equal(out.cpu().numpy().view(np.uint64).reshape(want.shape), want)
    # fill alone, split over two calls with first_index: same rows
    co2 = torch.zeros_like(co)
    half = B // 2
    with torch.cuda.stream(st):
        assert eng.dev_fill_coeffs(seed, sec.data_ptr(), half, first, d, co2.data_ptr(), stream=st.cuda_stream) == 0
        assert eng.dev_fill_coeffs(seed, sec.data_ptr() + half * ew * 8, B - half, first + half, d,
                                   co2.data_ptr() + half * (d + 1) * ew * 8, stream=st.cuda_stream) == 0
    st.synchronize()
    assert torch.equal(co, co2)


def test_uniformity_and_range_at_scale(env):
    eng, field = env
    _, mod = FIELDS[field]
    B, d = 1 << 16, 5
    ew = 4 if field == "fr" else 1
    dev = torch.device("cuda:0")
    sec = torch.zeros(B * ew, dtype=torch.int64, device=dev)
    co = torch.zeros(B * (d + 1) * ew, dtype=torch.int64, device=dev)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        assert eng.dev_fill_coeffs(b"\x07" * 32, sec.data_ptr(), B, 0, d, co.data_ptr(), stream=st.cuda_stream) == 0
    st.synchronize()
    a = co.cpu().numpy().view(np.uint64).reshape(B, d + 1, ew)
    assert not a[:, 0].any()
    top = a[:, 1:, ew - 1].reshape(-1)  # most significant 64 bits of every drawn coefficient
    assert int(top.max()) <= (mod >> (64 * (ew - 1)))
    # top byte spread: every value of the leading byte that the modulus allows occurs
    lead = (top >> np.uint64(56)).astype(np.int64)
    assert len(np.unique(lead)) == (mod >> (64 * ew - 8)) + 1
    assert len(np.unique(a[:, 1:, 0].reshape(-1))) == B * d  # no repeated low words among 327680 draws


def test_errors(env):
    eng, field = env
    x = to_arr(field, [1, 2, 3], (3,))
    rc, _ = eng.compute_shares_seeded(bytes(32), x, 3, 3)
    assert rc == 4  # InvalidInput: n <= d
    rc, sh = eng.compute_shares_seeded(bytes(32), x[:0], 4, 1)
    assert rc == 0 and sh.shape[1] == 0


def test_secrets_drawn_too(env):
    """secrets = NULL: coefficient 0 comes from the stream as well (RanSha's dealer: a random secret per sharing)"""
    eng, field = env
    sp, mod = FIELDS[field]
    n, d, B, first = 10, 3, 77, 5
    seed = bytes(range(100, 132))
    rc, sh = eng.compute_shares_seeded(seed, B, n, d, first_index=first)
    assert rc == 0
    got = to_ints(field, sh)
    for b in range(B):
        poly = sp.seeded_polynomial(words(seed), first + b, None, d)
        assert poly[0] == sp.seeded_coefficient(words(seed), first + b, 0)
        assert [got[j * B + b] for j in range(n)] == [s.v for s in sp.compute_shares(poly, n, d)]
    ew = 4 if field == "fr" else 1
    dev = torch.device("cuda:0")
    co = torch.zeros(B * (d + 1) * ew, dtype=torch.int64, device=dev)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        assert eng.dev_fill_coeffs(seed, 0, B, first, d, co.data_ptr(), stream=st.cuda_stream) == 0
    st.synchronize()
    rows = to_ints(field, co.cpu().numpy().view(np.uint64))
    assert rows[: d + 1] == sp.seeded_polynomial(words(seed), first, None, d)
