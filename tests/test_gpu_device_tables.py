"""The matrix-core byte-digit table of a sender set EXPANDED ON THE DEVICE (csrc/kernels_tables.hpp) against the host
reference (csrc/tables_mfma.hpp::build_mfma_table), byte for byte.  The host reference is what tests/test_host_tables.py
compares digit by digit with the big-int oracle under the sanitizers, so the chain oracle -> host table -> device table is
closed value by value.  Then the regime the device build exists for: decodes whose sender set has never
been seen, against the oracle."""
import ctypes as C
import random

import numpy as np
import pytest

from __graft_entry__ import load_package
from oracle import cref as O

pytestmark = pytest.mark.gpu


def _table(eng, ids, n, d, t, on_device):
    L = eng.L
    arr = (C.c_size_t * len(ids))(*ids)
    nbytes = C.c_size_t(0)
    rc = L.hbmpc_debug_mfma_table(eng.ctx, arr, C.c_size_t(len(ids)), C.c_size_t(n), C.c_size_t(d), C.c_size_t(t), C.c_int(on_device), None,
                                  C.c_size_t(0), C.byref(nbytes))
    assert rc == 0, eng.last_error()
    out = np.zeros(nbytes.value, dtype=np.uint8)
    rc = L.hbmpc_debug_mfma_table(eng.ctx, arr, C.c_size_t(len(ids)), C.c_size_t(n), C.c_size_t(d), C.c_size_t(t), C.c_int(on_device),
                                  out.ctypes.data_as(C.c_void_p), C.c_size_t(out.size), C.byref(nbytes))
    assert rc == 0, eng.last_error()
    return out


@pytest.mark.parametrize("n,t,d", [(4, 1, 1), (7, 2, 2), (16, 5, 5), (16, 5, 10), (31, 10, 10), (40, 13, 13), (64, 21, 14), (255, 84, 9)])
def test_device_table_equals_host_table(n, t, d):
    eng = load_package().Engine(0)
    rng = random.Random(n * 1000 + d)
    try:
        for trial in range(4):
            S = d + t + 1 if trial % 2 == 0 else rng.randint(d + t + 1, n)
            ids = sorted(rng.sample(range(n), S))
            host, dev = _table(eng, ids, n, d, t, 0), _table(eng, ids, n, d, t, 1)
            assert host.size == (t + d + 1) * ((d + 1) * 1024 + 128)
            assert np.array_equal(host, dev), (ids, int(np.flatnonzero(host != dev)[0]))
    finally:
        eng.close()


@pytest.mark.parametrize("n,t,d,G", [(16, 5, 5, 5000), (31, 10, 10, 4096), (31, 10, 10, 70000), (64, 21, 14, 9000)])
def test_decodes_with_sender_sets_never_seen_before(n, t, d, G):
    """every call brings a NEW sender set (and arrival order) at a batch size that takes the matrix cores: the first d + t + 1
    arrivals (BatchRecon's decode, one launch) and larger sets with corrupted chunks (fallback launches behind it)"""
    eng = load_package().Engine(0)
    rng = random.Random(n + G)
    x = O.fill_random(77 + n, G * (d + 1)).reshape(G, d + 1, 4)
    rc, y = O.vandermonde_apply(x, n, d)
    try:
        for trial in range(6):
            S = d + t + 1 if trial % 2 == 0 else rng.randint(d + t + 2, n)
            ids = rng.sample(range(n), S)
            ev = np.ascontiguousarray(y[ids])
            if trial >= 2:
                for g in rng.sample(range(G), 20):
                    ev[rng.randrange(S), g, 0] ^= np.uint64(1)
            got = eng.batch_recover(ids, ev, n, d, t)
            want = O.batch_recover(ids, ev, n, d, t)
            assert got[0] == want[0] and all(np.array_equal(u, v) for u, v in zip(got[1:], want[1:])), (trial, ids)
    finally:
        eng.close()
