"""ctypes loader for the C oracle (oracle/libhbmpc_oracle.so).  TEST INFRASTRUCTURE ONLY.

Arrays are numpy uint64 with a trailing axis of 4 limbs (least-significant first) == U256[].
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

U64P = C.POINTER(C.c_uint64)


def _cpu_has_v3() -> bool:
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        return False
    return all(f in flags for f in (" bmi2", " adx", " avx2"))


def build(force: bool = False) -> str:
    name = "libhbmpc_oracle_v3.so" if _cpu_has_v3() else "libhbmpc_oracle.so"
    so = os.path.join(_HERE, name)
    # built explicitly by __graft_entry__.build() / `make -C oracle`; here only when missing (never on
    # mtimes: several bench ranks may import this at once on a freshly copied tree)
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "-B", "all"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def u256(shape):
    return np.zeros(tuple(shape) + (4,), dtype=np.uint64)


def ints_to_u256(vals) -> np.ndarray:
    """nested list / array of python ints -> uint64[..., 4]"""
    a = np.array(vals, dtype=object)
    out = np.zeros(a.shape + (4,), dtype=np.uint64)
    it = np.nditer(a, flags=["multi_index", "refs_ok"])
    for x in it:
        v = int(x.item())
        for k in range(4):
            out[it.multi_index + (k,)] = (v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return out


def u256_to_ints(a: np.ndarray):
    a = np.asarray(a, dtype=np.uint64)
    flat = a.reshape(-1, 4)
    vals = [int(r[0]) | (int(r[1]) << 64) | (int(r[2]) << 128) | (int(r[3]) << 192) for r in flat]
    return np.array(vals, dtype=object).reshape(a.shape[:-1]).tolist() if a.ndim > 1 else vals[0]


def _sz(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint64))


def fill_random(seed: int, count: int) -> np.ndarray:
    out = u256((count,))
    lib().oracle_fill_random(C.c_uint64(seed), C.c_size_t(count), _p(out))
    return out


def compute_shares(coeffs: np.ndarray, n: int, d: int):
    coeffs = np.ascontiguousarray(coeffs)
    B = coeffs.shape[0]
    out = u256((n, B))
    rc = lib().oracle_compute_shares(_p(coeffs), C.c_size_t(B), C.c_size_t(n), C.c_size_t(d), _p(out))
    return rc, out


def make_vandermonde(n: int, d: int):
    out = u256((n, d + 1))
    rc = lib().oracle_make_vandermonde(C.c_size_t(n), C.c_size_t(d), _p(out))
    return rc, out


def vandermonde_apply(x: np.ndarray, n: int, d: int):
    x = np.ascontiguousarray(x)
    G = x.shape[0]
    out = u256((n, G))
    rc = lib().oracle_vandermonde_apply(_p(x), C.c_size_t(G), C.c_size_t(n), C.c_size_t(d), _p(out))
    return rc, out


def batch_recover(sender_ids, evals: np.ndarray, n: int, d: int, t: int):
    evals = np.ascontiguousarray(evals)
    S = len(sender_ids)
    G = evals.shape[1] if evals.ndim == 3 else 0
    ids = _sz(sender_ids)
    out = u256((G, d + 1))
    nco = np.zeros(G, dtype=np.uint32)
    status = np.zeros(G, dtype=np.uint8)
    rc = lib().oracle_batch_recover(_p(ids), C.c_size_t(S), _p(evals), C.c_size_t(G), C.c_size_t(n), C.c_size_t(d),
                                    C.c_size_t(t), _p(out), _p(nco), _p(status))
    return rc, out, nco, status


def batch_recover_p0(sender_ids, evals: np.ndarray, n: int, d: int, t: int):
    evals = np.ascontiguousarray(evals)
    S = len(sender_ids)
    G = evals.shape[1] if evals.ndim == 3 else 0
    ids = _sz(sender_ids)
    out = u256((G,))
    status = np.zeros(G, dtype=np.uint8)
    rc = lib().oracle_batch_recover_p0(_p(ids), C.c_size_t(S), _p(evals), C.c_size_t(G), C.c_size_t(n),
                                       C.c_size_t(d), C.c_size_t(t), _p(out), _p(status))
    return rc, out, status


def recover_secret(ids, degrees, vals: np.ndarray, n: int, t: int):
    vals = np.ascontiguousarray(vals)
    S = len(ids)
    cap = (int(degrees[0]) + 1) if S else 1
    out = u256((max(cap, 1),))
    nco = C.c_size_t(0)
    sec = u256((1,))
    rc = lib().oracle_recover_secret(_p(_sz(ids)), _p(_sz(degrees)), _p(vals), C.c_size_t(S), C.c_size_t(n),
                                     C.c_size_t(t), _p(out), C.byref(nco), _p(sec))
    return rc, out[: nco.value], sec[0]


def gao_rs_decode(received: np.ndarray, k: int, n: int, erasures):
    received = np.ascontiguousarray(received)
    out = u256((max(k, 1),))
    nco = C.c_size_t(0)
    er = _sz(erasures)
    rc = lib().oracle_gao_rs_decode(_p(received), C.c_size_t(k), C.c_size_t(n), _p(er), C.c_size_t(len(erasures)),
                                    _p(out), C.byref(nco))
    return rc, out[: nco.value]


def nonrobust_recover_secret(ids, degrees, vals: np.ndarray, n: int):
    vals = np.ascontiguousarray(vals)
    S = len(ids)
    out = u256((max(S, 1),))
    nco = C.c_size_t(0)
    sec = u256((1,))
    rc = lib().oracle_nonrobust_recover_secret(_p(_sz(ids)), _p(_sz(degrees)), _p(vals), C.c_size_t(S),
                                               C.c_size_t(n), _p(out), C.byref(nco), _p(sec))
    return rc, out[: nco.value], sec[0]


def _ew(name, ins, n_out=1, extra=()):
    ins = [np.ascontiguousarray(a) for a in ins]
    N = ins[0].shape[-2] if name == "oracle_truncpr_rdash" else ins[0].shape[0]
    outs = [u256((N,)) for _ in range(n_out)]
    args = [_p(a) for a in ins] + [C.c_size_t(e) for e in extra] + [C.c_size_t(N)] + [_p(o) for o in outs]
    rc = getattr(lib(), name)(*args)
    return (rc, *outs)


def triple_local(a, b, r2t):
    return _ew("oracle_triple_local", [a, b, r2t])


def triple_finalize(rt, opened):
    return _ew("oracle_triple_finalize", [rt, opened])


def beaver_open_shares(a, b, x, y):
    return _ew("oracle_beaver_open_shares", [a, b, x, y], n_out=2)


def beaver_finalize(c, x, y, d, e):
    return _ew("oracle_beaver_finalize", [c, x, y, d, e])


def truncpr_rdash(r_bits, m):
    r_bits = np.ascontiguousarray(r_bits)
    N = r_bits.shape[1]
    out = u256((N,))
    rc = lib().oracle_truncpr_rdash(_p(r_bits), C.c_size_t(m), C.c_size_t(N), _p(out))
    return rc, out


def truncpr_open_share(a, r_dash, r_int, k, m):
    return _ew("oracle_truncpr_open_share", [a, r_dash, r_int], extra=(k, m))


def truncpr_finalize(a, r_dash, c_open, m):
    return _ew("oracle_truncpr_finalize", [a, r_dash, c_open], extra=(m,))


def fr_binop(name, a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    out = u256((a.shape[0],))
    getattr(lib(), "oracle_fr_" + name)(_p(a), _p(b), C.c_size_t(a.shape[0]), _p(out))
    return out


def fr_inv(a):
    a = np.ascontiguousarray(a)
    out = u256((a.shape[0],))
    lib().oracle_fr_inv(_p(a), C.c_size_t(a.shape[0]), _p(out))
    return out


def domain_elements(n: int, count: int):
    out = u256((count,))
    lib().oracle_domain_elements(C.c_size_t(n), C.c_size_t(count), _p(out))
    return out
