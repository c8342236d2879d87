#!/usr/bin/env python3
"""Generates tests/golden/hbmpc_golden.json from oracle/spec.py (Python big-ints) and
tests/golden/hbmpc_golden_gl.json from oracle/spec_gl.py (the same restatement over Goldilocks).

The reference holds no golden vectors for this path and cannot be run here,
so these vectors come from the independent big-int restatement; inputs reuse the literal inputs
of the reference's own tests where they exist (U256{3,3,22,22} / {520,86,9,18} / {16,33,44,81}
of ffi/tests/secret_share.c, f = 7+3x+5x^2 of robust_interpolate.rs:656, secret 42, 918520 ...).
Run:  python tests/golden/make_golden.py   (deterministic; rewrites the JSON in place)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import spec as SPEC_FR  # noqa: E402
from oracle.spec_gl import S as SPEC_GL  # noqa: E402


def main():
    build(SPEC_FR, "hbmpc_golden.json", 64, True)
    build(SPEC_GL, "hbmpc_golden_gl.json", 16, False)   # TruncPr is big-field-only in the reference


def build(S, fname, hexdigits, with_truncpr):
    R = S.R_MOD
    H = lambda v: format(v % R, "0%dx" % hexdigits)  # noqa: E731

    def hx(x):
        return [hx(y) for y in x] if isinstance(x, (list, tuple)) else H(x)

    rng = S.SplitMix64(0xC0FFEE00)
    out = {"modulus": format(R, "0%dx" % hexdigits), "cases": []}
    add = out["cases"].append
    lit = [S.from_limbs([3, 3, 22, 22]), S.from_limbs([520, 86, 9, 18]), S.from_limbs([16, 33, 44, 81]),
           42, 918520, 0, 1, R - 1]

    # domain + vandermonde
    for n, d in [(4, 2), (6, 2), (7, 2), (10, 3), (16, 5), (16, 10), (16, 15), (31, 10), (5, 1), (20, 6), (1, 0), (2, 1), (3, 0)]:
        add({"op": "make_vandermonde", "n": n, "d": d, "v": hx(S.make_vandermonde(n, d))})

    # compute_shares: coeffs[B][d+1] -> shares[n][B]
    for n, d, B in [(4, 1, 5), (6, 2, 8), (7, 2, 4), (10, 3, 4), (16, 5, 9), (31, 10, 3), (16, 10, 3), (5, 1, 3),
                    (20, 6, 2), (13, 4, 2), (16, 15, 2), (3, 0, 4), (1, 0, 2), (64, 21, 2), (100, 33, 1)]:
        coeffs = []
        for b in range(B):
            row = [lit[b % len(lit)]] + [rng.fr() for _ in range(d)]
            if b == B - 1:
                row = [0] * (d + 1)  # the zero polynomial
            coeffs.append(row)
        if (n, d) == (16, 5):
            coeffs[0] = [7, 3, 5, 0, 0, 0]
        sh = [[S.compute_shares(c, n, d)[j].v for c in coeffs] for j in range(n)]
        add({"op": "compute_shares", "n": n, "d": d, "coeffs": hx(coeffs), "shares": hx(sh)})

    # vandermonde_apply: x[G][d+1] -> y[n][G]
    for n, d, G in [(4, 2, 3), (4, 1, 4), (7, 2, 3), (10, 3, 4), (16, 5, 5), (16, 10, 3), (31, 10, 3), (16, 15, 2),
                    (13, 4, 2), (7, 6, 2), (31, 20, 2)]:
        x = [[rng.fr() for _ in range(d + 1)] for _ in range(G)]
        if (n, d) == (4, 2):
            x[0] = [1, 2, 3]
        v = S.make_vandermonde(n, d)
        y = [[sum(v[j][k] * x[g][k] for k in range(d + 1)) % R for g in range(G)] for j in range(n)]
        add({"op": "vandermonde_apply", "n": n, "d": d, "x": hx(x), "y": hx(y)})

    # batch_recover: honest, shuffled senders, corrupted, missing senders, failing chunk
    def brec(n, t, d, G, ids, corrupt=None, tag=""):
        polys = [[rng.fr() for _ in range(d + 1)] for _ in range(G)]
        if G > 2:
            polys[1] = [rng.fr()] + [0] * d                      # constant polynomial (trailing zeros)
            polys[2] = [0] * (d + 1)                             # zero polynomial
        ev = [(i, [S.p_eval(p, S.domain_element(n, i)) for p in polys]) for i in ids]
        if corrupt:
            for (pos, c, delta) in corrupt:
                ev[pos][1][c] = (ev[pos][1][c] + delta) % R
        case = {"op": "batch_recover", "tag": tag, "n": n, "t": t, "d": d, "ids": list(ids),
                "evals": hx([v for _, v in ev])}
        try:
            res = S.batch_recover_secret(ev, n, d, t)
            case["rc"] = 0
            case["coeffs"] = hx([r + [0] * (d + 1 - len(r)) for r in res])
            case["ncoeffs"] = [len(r) for r in res]
        except S.ShareErr as e:
            case["rc"] = e.code
        add(case)

    brec(10, 3, 3, 16, list(range(10))[::-1], tag="reference test :880 (reversed arrival)")
    brec(10, 3, 3, 8, list(range(10)), corrupt=[(b, c, (c + 1) * 7 + b) for b in range(3) for c in range(8)],
         tag="reference test :931 (first t senders corrupted)")
    brec(4, 1, 1, 5, [2, 0, 3, 1], tag="n=4 t=1")
    brec(7, 2, 2, 4, [6, 5, 4, 3, 2], tag="needed senders only, high ids")
    brec(16, 5, 5, 6, list(range(16)), tag="cfg2 shape")
    brec(16, 5, 10, 5, list(range(16)), tag="triple_gen degree 2t")
    brec(16, 5, 10, 5, list(range(16)), corrupt=[(3, 0, 5), (7, 4, 9)], tag="degree 2t: needed == n, no OEC room")
    brec(31, 10, 10, 4, list(range(31)), tag="cfg3 shape")
    brec(31, 10, 10, 6, list(range(30, -1, -1)), corrupt=[(30, 0, 1), (29, 0, 2), (0, 3, 77), (30 - 12, 5, 1)],
         tag="cfg3 with corruption in and outside the verify window")
    brec(10, 3, 3, 4, list(range(10)), corrupt=[(0, 1, 1), (1, 1, 1), (2, 1, 1), (3, 1, 1)], tag="t+1 errors: chunk 1 fails")
    brec(10, 3, 3, 3, [0, 1, 2, 3, 4, 5], tag="too few senders")
    brec(9, 3, 3, 3, list(range(9)), tag="n < 3t+1")
    brec(10, 3, 3, 3, [0, 1, 2, 3, 4, 5, 5, 6], tag="duplicate sender")
    brec(10, 3, 3, 3, [0, 1, 2, 3, 4, 5, 6, 10], tag="sender out of range")
    brec(13, 4, 4, 3, [12, 0, 5, 7, 3, 9, 1, 10, 2, 11], corrupt=[(1, 0, 3), (4, 2, 8)], tag="gaps + corruption")

    # recover_secret (single): all corruption combos n=7,t=2 + the literal cases
    from itertools import combinations
    n, t = 7, 2
    coeffs = [42, rng.fr(), rng.fr()]
    base = S.compute_shares(coeffs, n, t)
    for k in range(0, t + 2):
        for idx in list(combinations(range(n), k))[:12]:
            vals = [s.v for s in base]
            for i in idx:
                vals[i] = (vals[i] + 999) % R
            case = {"op": "recover_secret", "n": n, "t": t, "ids": list(range(n)), "degrees": [t] * n, "vals": hx(vals)}
            try:
                c, z = S.recover_secret([S.Share(v, i, t) for i, v in enumerate(vals)], n, t)
                case.update(rc=0, coeffs=hx(c), secret=H(z))
            except S.ShareErr as e:
                case["rc"] = e.code
            add(case)
    for secret in lit[:3]:
        sh = S.compute_shares([secret, rng.fr(), rng.fr()], 6, 2)
        c, z = S.recover_secret(sh, 6, 1)
        add({"op": "recover_secret", "n": 6, "t": 1, "ids": list(range(6)), "degrees": [2] * 6,
             "vals": hx([s.v for s in sh]), "rc": 0, "coeffs": hx(c), "secret": H(z)})
    sh = S.compute_shares([7, 3, 5], 16, 2)
    c, z = S.recover_secret(sh[:5], 16, 2)
    add({"op": "recover_secret", "n": 16, "t": 2, "ids": list(range(5)), "degrees": [2] * 5,
         "vals": hx([s.v for s in sh[:5]]), "rc": 0, "coeffs": hx(c), "secret": H(z)})
    sh = S.compute_shares([5, 0, 0], 7, 2)  # constant polynomial: trimmed to 1 coefficient
    c, z = S.recover_secret(sh, 7, 2)
    add({"op": "recover_secret", "n": 7, "t": 2, "ids": list(range(7)), "degrees": [2] * 7,
         "vals": hx([s.v for s in sh]), "rc": 0, "coeffs": hx(c), "secret": H(z)})

    # gao_rs_decode
    for n, t, er, errs in [(8, 2, [1, 2], []), (10, 2, [], [(2, 5), (4, 3)]), (10, 3, [], [(0, 5), (5, 3), (9, 3)]),
                           (10, 3, [7], [(1, 5), (2, 3)]), (10, 3, [], [(0, 1), (1, 1), (2, 1), (3, 1)])]:
        co = [42] + [rng.fr() for _ in range(t)]
        vals = [s.v for s in S.compute_shares(co, n, t)]
        for i in er:
            vals[i] = 0
        for i, dlt in errs:
            vals[i] = (vals[i] + dlt) % R
        case = {"op": "gao_rs_decode", "n": n, "k": t + 1, "erasures": er, "received": hx(vals)}
        try:
            case.update(rc=0, coeffs=hx(S.gao_rs_decode(vals, t + 1, n, er)))
        except S.ShareErr as e:
            case["rc"] = e.code
        add(case)

    # nonrobust recover
    sh = S.compute_shares([918520] + [rng.fr() for _ in range(5)], 6, 5)
    c, z = S.nonrobust_recover_secret(sh, 6)
    add({"op": "nonrobust_recover", "n": 6, "ids": list(range(6)), "degrees": [5] * 6, "vals": hx([s.v for s in sh]),
         "rc": 0, "coeffs": hx(c), "secret": H(z)})
    sh = S.compute_shares([1, 2, 3, 4], 8, 3)
    add({"op": "nonrobust_recover", "n": 8, "ids": list(range(8)), "degrees": [2] * 8, "vals": hx([s.v for s in sh]),
         "rc": S.DegreeMismatch.code})

    # element-wise
    N = 6
    v = lambda: [rng.fr() for _ in range(N)]  # noqa: E731
    a, b, r2t = v(), v(), v()
    a[0], b[0] = R - 1, R - 1
    add({"op": "triple_local", "a": hx(a), "b": hx(b), "r2t": hx(r2t),
         "out": hx([(x * y - z) % R for x, y, z in zip(a, b, r2t)])})
    rt, op = v(), v()
    add({"op": "triple_finalize", "rt": hx(rt), "opened": hx(op), "out": hx([(x + y) % R for x, y in zip(rt, op)])})
    a, b, x, y = v(), v(), v(), v()
    add({"op": "beaver_open_shares", "a": hx(a), "b": hx(b), "x": hx(x), "y": hx(y),
         "d_sh": hx([(p - q) % R for p, q in zip(a, x)]), "e_sh": hx([(p - q) % R for p, q in zip(b, y)])})
    c, d, e = v(), v(), v()
    add({"op": "beaver_finalize", "c": hx(c), "x": hx(x), "y": hx(y), "d": hx(d), "e": hx(e),
         "z": hx([(c[i] - d[i] * e[i] - d[i] * y[i] - e[i] * x[i]) % R for i in range(N)])})
    for m in ((1, 4, 16, 20) if with_truncpr else ()):
        bits = [v() for _ in range(m)]
        add({"op": "truncpr_rdash", "m": m, "r_bits": hx(bits),
             "r_dash": hx([sum((1 << j) * bits[j][i] for j in range(m)) % R for i in range(N)])})
    for k, m in (((16, 4), (32, 16), (1, 0), (250, 255)) if with_truncpr else ()):
        a, rd, ri, co = v(), v(), v(), v()
        add({"op": "truncpr_open_share", "k": k, "m": m, "a": hx(a), "r_dash": hx(rd), "r_int": hx(ri),
             "open": hx([(a[i] + (1 << (k - 1)) + (1 << m) * ri[i] + rd[i]) % R for i in range(N)])})
        add({"op": "truncpr_finalize", "m": m, "a": hx(a), "r_dash": hx(rd), "c_open": hx(co),
             "d": hx([S.truncpr_finalize(S.Share(a[i], 0, 1), S.Share(rd[i], 0, 1), co[i], m).v for i in range(N)])})

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), fname)
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(f"wrote {path}: {len(out['cases'])} cases, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
