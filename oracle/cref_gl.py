"""ctypes binding of oracle/libhbmpc_oracle_gl.so: oracle/hbmpc_oracle.c compiled with -DORACLE_GOLDILOCKS, i.e. the C
restatement of the reference path over GoldilocksField (common/math/goldilocks.rs:4-13).

TEST INFRASTRUCTURE ONLY (see the header of hbmpc_oracle.c).  Same function set as oracle/cref.py, so tests/golden_util.py
and the GPU parity tests drive it as an engine; elements are numpy uint64 (one word each)."""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB = None
P = (1 << 64) - (1 << 32) + 1


def build() -> str:
    subprocess.check_call(["make", "-C", _DIR, "libhbmpc_oracle_gl.so"], stdout=subprocess.DEVNULL)
    return os.path.join(_DIR, "libhbmpc_oracle_gl.so")


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _sz(v):
    return np.asarray(list(v), dtype=np.uint64)


def fill_random(seed: int, count: int) -> np.ndarray:
    out = np.zeros(count, dtype=np.uint64)
    lib().oracle_gl_fill_random(C.c_uint64(seed), C.c_size_t(count), _p(out))
    return out


def compute_shares(coeffs, n, d):
    coeffs = _u(coeffs)
    B = coeffs.shape[0]
    out = np.zeros((n, B), dtype=np.uint64)
    rc = lib().oracle_gl_compute_shares(_p(coeffs), C.c_size_t(B), C.c_size_t(n), C.c_size_t(d), _p(out))
    return rc, out


def make_vandermonde(n, d):
    out = np.zeros((n, d + 1), dtype=np.uint64)
    rc = lib().oracle_gl_make_vandermonde(C.c_size_t(n), C.c_size_t(d), _p(out))
    return rc, out


def vandermonde_apply(x, n, d):
    x = _u(x)
    G = x.shape[0]
    out = np.zeros((n, G), dtype=np.uint64)
    rc = lib().oracle_gl_vandermonde_apply(_p(x), C.c_size_t(G), C.c_size_t(n), C.c_size_t(d), _p(out))
    return rc, out


def batch_recover(sender_ids, evals, n, d, t):
    evals = _u(evals)
    S = len(sender_ids)
    G = evals.shape[1] if evals.ndim == 2 else 0
    ids = _sz(sender_ids)
    out = np.zeros((G, d + 1), dtype=np.uint64)
    nco = np.zeros(G, dtype=np.uint32)
    status = np.zeros(G, dtype=np.uint8)
    rc = lib().oracle_gl_batch_recover(_p(ids), C.c_size_t(S), _p(evals), C.c_size_t(G), C.c_size_t(n), C.c_size_t(d),
                                       C.c_size_t(t), _p(out), _p(nco), _p(status))
    return rc, out, nco, status


def batch_recover_p0(sender_ids, evals, n, d, t):
    evals = _u(evals)
    S = len(sender_ids)
    G = evals.shape[1] if evals.ndim == 2 else 0
    ids = _sz(sender_ids)
    out = np.zeros(G, dtype=np.uint64)
    status = np.zeros(G, dtype=np.uint8)
    rc = lib().oracle_gl_batch_recover_p0(_p(ids), C.c_size_t(S), _p(evals), C.c_size_t(G), C.c_size_t(n), C.c_size_t(d),
                                          C.c_size_t(t), _p(out), _p(status))
    return rc, out, status


def recover_secret(ids, degrees, vals, n, t):
    vals = _u(vals)
    S = len(ids)
    co = np.zeros(max(n, 1), dtype=np.uint64)
    nco = C.c_size_t(0)
    sec = np.zeros(1, dtype=np.uint64)
    rc = lib().oracle_gl_recover_secret(_p(_sz(ids)), _p(_sz(degrees)), _p(vals), C.c_size_t(S), C.c_size_t(n), C.c_size_t(t),
                                        _p(co), C.byref(nco), _p(sec))
    return (rc, co[: nco.value].copy(), sec[0]) if rc == 0 else (rc, None, None)


def gao_rs_decode(received, k, n, erasures):
    received = _u(received)
    er = _sz(erasures)
    co = np.zeros(max(n, 1), dtype=np.uint64)
    nco = C.c_size_t(0)
    rc = lib().oracle_gl_gao_rs_decode(_p(received), C.c_size_t(k), C.c_size_t(n), _p(er), C.c_size_t(len(erasures)), _p(co),
                                       C.byref(nco))
    return (rc, co[: nco.value].copy()) if rc == 0 else (rc, None)


def nonrobust_recover_secret(ids, degrees, vals, n):
    vals = _u(vals)
    S = len(ids)
    co = np.zeros(max(S, 1), dtype=np.uint64)
    nco = C.c_size_t(0)
    sec = np.zeros(1, dtype=np.uint64)
    rc = lib().oracle_gl_nonrobust_recover_secret(_p(_sz(ids)), _p(_sz(degrees)), _p(vals), C.c_size_t(S), C.c_size_t(n), _p(co),
                                                  C.byref(nco), _p(sec))
    return (rc, co[: nco.value].copy(), sec[0]) if rc == 0 else (rc, None, None)


def _ew(name, ins, n_out=1):
    ins = [_u(a) for a in ins]
    N = ins[0].shape[0]
    outs = [np.zeros(N, dtype=np.uint64) for _ in range(n_out)]
    rc = getattr(lib(), name)(*([_p(a) for a in ins] + [C.c_size_t(N)] + [_p(o) for o in outs]))
    return (rc, *outs)


def triple_local(a, b, r2t):
    return _ew("oracle_gl_triple_local", [a, b, r2t])


def triple_finalize(rt, opened):
    return _ew("oracle_gl_triple_finalize", [rt, opened])


def beaver_open_shares(a, b, x, y):
    return _ew("oracle_gl_beaver_open_shares", [a, b, x, y], n_out=2)


def beaver_finalize(c, x, y, d, e):
    return _ew("oracle_gl_beaver_finalize", [c, x, y, d, e])


def domain_elements(n, count):
    out = np.zeros(count, dtype=np.uint64)
    lib().oracle_gl_domain_elements(C.c_size_t(n), C.c_size_t(count), _p(out))
    return out
