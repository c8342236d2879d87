"""Runs the committed golden vectors (tests/golden/hbmpc_golden.json) against an engine.

An engine is any module/object with the function set of oracle/cref.py (numpy uint64[...,4]
arrays in, (rc, outputs...) out): the C oracle (CPU tests) or the HIP library through its C ABI
(GPU tests)."""
import json
import os

import numpy as np

from oracle.cref import ints_to_u256

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_FILES = {"fr": "hbmpc_golden.json", "goldilocks": "hbmpc_golden_gl.json"}
_FIELD = "fr"  # the field of the case set being checked (set by run_all)


def load(field="fr"):
    with open(os.path.join(_DIR, _FILES[field])) as f:
        return json.load(f)["cases"]


def U(h):
    """nested hex lists -> uint64[...,4] (Fr) or uint64[...] (Goldilocks)"""
    def conv(x):
        return [conv(y) for y in x] if isinstance(x, list) else int(x, 16)
    arr = conv(h)
    if _FIELD == "goldilocks":
        return np.array(arr, dtype=np.uint64)
    if isinstance(arr, list) and len(arr) == 0:
        return np.zeros((0, 4), dtype=np.uint64)
    return ints_to_u256(arr)


def eq(a, b):
    a, b = np.asarray(a, dtype=np.uint64), np.asarray(b, dtype=np.uint64)
    return a.shape == b.shape and bool((a == b).all())


def check_case(E, c):
    op = c["op"]
    if op == "make_vandermonde":
        rc, v = E.make_vandermonde(c["n"], c["d"])
        assert rc == 0 and eq(v, U(c["v"])), c
    elif op == "compute_shares":
        rc, sh = E.compute_shares(U(c["coeffs"]), c["n"], c["d"])
        assert rc == 0 and eq(sh, U(c["shares"])), (op, c["n"], c["d"])
    elif op == "vandermonde_apply":
        rc, y = E.vandermonde_apply(U(c["x"]), c["n"], c["d"])
        assert rc == 0 and eq(y, U(c["y"])), (op, c["n"], c["d"])
    elif op == "batch_recover":
        ev = U(c["evals"])
        rc, co, nco, st = E.batch_recover(c["ids"], ev, c["n"], c["d"], c["t"])
        assert rc == c["rc"], (c["tag"], rc, c["rc"])
        if c["rc"] == 0:
            assert eq(co, U(c["coeffs"])), c["tag"]
            assert list(nco) == c["ncoeffs"], c["tag"]
            rc2, p0, st2 = E.batch_recover_p0(c["ids"], ev, c["n"], c["d"], c["t"])
            assert rc2 == 0 and eq(p0, U(c["coeffs"])[:, 0]), c["tag"]
            assert list(st) == list(st2)
    elif op == "recover_secret":
        rc, co, sec = E.recover_secret(c["ids"], c["degrees"], U(c["vals"]), c["n"], c["t"])
        assert rc == c["rc"], c
        if rc == 0:
            assert eq(co, U(c["coeffs"])) and eq(sec, U(c["secret"])), c
    elif op == "gao_rs_decode":
        rc, co = E.gao_rs_decode(U(c["received"]), c["k"], c["n"], c["erasures"])
        assert rc == c["rc"], c
        if rc == 0:
            assert eq(co, U(c["coeffs"])), c
    elif op == "nonrobust_recover":
        rc, co, sec = E.nonrobust_recover_secret(c["ids"], c["degrees"], U(c["vals"]), c["n"])
        assert rc == c["rc"], c
        if rc == 0:
            assert eq(co, U(c["coeffs"])) and eq(sec, U(c["secret"])), c
    elif op == "triple_local":
        rc, o = E.triple_local(U(c["a"]), U(c["b"]), U(c["r2t"]))
        assert rc == 0 and eq(o, U(c["out"]))
    elif op == "triple_finalize":
        rc, o = E.triple_finalize(U(c["rt"]), U(c["opened"]))
        assert rc == 0 and eq(o, U(c["out"]))
    elif op == "beaver_open_shares":
        rc, d, e = E.beaver_open_shares(U(c["a"]), U(c["b"]), U(c["x"]), U(c["y"]))
        assert rc == 0 and eq(d, U(c["d_sh"])) and eq(e, U(c["e_sh"]))
    elif op == "beaver_finalize":
        rc, z = E.beaver_finalize(U(c["c"]), U(c["x"]), U(c["y"]), U(c["d"]), U(c["e"]))
        assert rc == 0 and eq(z, U(c["z"]))
    elif op == "truncpr_rdash":
        rc, r = E.truncpr_rdash(U(c["r_bits"]), c["m"])
        assert rc == 0 and eq(r, U(c["r_dash"])), c["m"]
    elif op == "truncpr_open_share":
        rc, o = E.truncpr_open_share(U(c["a"]), U(c["r_dash"]), U(c["r_int"]), c["k"], c["m"])
        assert rc == 0 and eq(o, U(c["open"])), (c["k"], c["m"])
    elif op == "truncpr_finalize":
        rc, o = E.truncpr_finalize(U(c["a"]), U(c["r_dash"]), U(c["c_open"]), c["m"])
        assert rc == 0 and eq(o, U(c["d"])), c["m"]
    else:
        raise AssertionError("unknown op " + op)


def run_all(E, field="fr"):
    global _FIELD
    _FIELD = field
    try:
        cases = load(field)
        for c in cases:
            check_case(E, c)
    finally:
        _FIELD = "fr"
    return len(cases)
