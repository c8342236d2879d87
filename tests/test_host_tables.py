"""The host-side table builders of the product (mpc-protocols_amd/csrc/tables.hpp + host_fr.hpp: domain elements,
Lagrange bases, verify matrices, inverses -- the only host arithmetic in the library, functions of (n, d, t, ids)
only) compiled for the CPU with AddressSanitizer + UBSan and compared value by value with the big-int oracle, for
both fields.  Runs without a GPU."""
import os
import subprocess

import pytest

from oracle import spec as SFR
from oracle.spec_gl import S as SGL

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "host_tables_dump")


@pytest.fixture(scope="module")
def dump():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "host_tables_dump"], stdout=subprocess.DEVNULL)

    def run(field, n, d, t, ids):
        p = subprocess.run([BIN, field, str(n), str(d), str(t), ",".join(map(str, ids))], capture_output=True, text=True,
                           timeout=120)
        assert p.returncode == 0, p.stderr[-2000:]          # ASan / UBSan findings abort with a non-zero code
        out = {}
        for line in p.stdout.splitlines():
            k, _, v = line.partition(" ")
            out.setdefault(k, []).append(v)
        return out
    return run


@pytest.mark.parametrize("field", ["fr", "gl"])
@pytest.mark.parametrize("n,d,t,ids", [(4, 1, 1, [3, 0, 2, 1]), (7, 2, 2, [6, 5, 4, 3, 2]), (16, 5, 5, list(range(16))),
                                       (16, 10, 5, list(range(16))), (31, 10, 10, list(range(30, -1, -1))),
                                       (100, 33, 33, list(range(0, 100)))])
def test_tables_match_oracle(dump, field, n, d, t, ids):
    S = SFR if field == "fr" else SGL
    P = S.R_MOD
    o = dump(field, n, d, t, ids)
    h = lambda xs: [int(x, 16) for x in xs]  # noqa: E731
    assert h(o["omega"]) == [S.domain_omega(n)]
    assert h(o["alpha"]) == [S.domain_element(n, j) for j in range(n)]
    m, needed = d + 1, d + t + 1
    xs = [S.domain_element(n, i) for i in ids[:m]]
    basis = []                                               # coefficients of the Lagrange basis polynomials
    for i in range(m):
        ys = [1 if j == i else 0 for j in range(m)]
        co = S.lagrange_interpolate(xs, ys)
        basis.append(co + [0] * (m - len(co)))
    assert h(o["basis"]) == [basis[i][k] for i in range(m) for k in range(m)]
    want_v = [S.p_eval(basis[i], S.domain_element(n, ids[s])) for s in range(m, needed) for i in range(m)]
    assert h(o.get("verify", [])) == want_v
    assert h(o["inv7"]) == [pow(7, -1, P)] and h(o["pow"]) == [pow(3, 1000003, P)]
    words = [int(w, 16) for w in o["const"][0].split()]
    a1 = S.domain_element(n, 1 if n > 1 else 0)
    if field == "gl":                                         # the value itself, two 32-bit words
        assert words == [a1 & 0xFFFFFFFF, a1 >> 32]
    else:                                                     # nine 29-bit limbs of a1 * 2^261 mod r (fr_u29.hpp)
        v = a1 * pow(2, 261, P) % P
        assert words == [(v >> (29 * i)) & 0x1FFFFFFF for i in range(9)]
    # second-chance tables: for every candidate window W the rows L_i^W(x_s) over the other positions of the OEC
    # prefix, then the coefficient rows of the basis
    S_cnt = len(ids)
    rmax = min(t, S_cnt - needed) if S_cnt > needed else 0
    if rmax < 1:
        assert "second" not in o
        return
    tok = o["second"][0].split()
    Pfx, nw = int(tok[0]), int(tok[1])
    starts = [int(x) for x in tok[2:2 + nw]]
    raw = [int(x, 16) for x in tok[2 + nw:]]
    assert Pfx == needed + rmax
    want_starts = []
    for off in (0, m, Pfx - m, (m + 1) // 2):
        if off + m <= Pfx and off not in want_starts:
            want_starts.append(off)
    assert starts == want_starts
    nl = 2 if field == "gl" else 9
    rinv = 1 if field == "gl" else pow(pow(2, 261, P), -1, P)
    vals = []
    for i in range(0, len(raw), nl):
        limbs = raw[i:i + nl]
        v = (limbs[0] | (limbs[1] << 32)) if field == "gl" else sum(l << (29 * j) for j, l in enumerate(limbs))
        vals.append(v * rinv % P)
    want = []
    for ws in starts:
        wx = [S.domain_element(n, ids[ws + i]) for i in range(m)]
        wb = []
        for i in range(m):
            co = S.lagrange_interpolate(wx, [1 if j == i else 0 for j in range(m)])
            wb.append(co + [0] * (m - len(co)))
        for s in range(Pfx):
            if ws <= s < ws + m:
                continue
            want += [S.p_eval(wb[i], S.domain_element(n, ids[s])) for i in range(m)]
        want += [wb[i][k] for k in range(m) for i in range(m)]
    assert vals == want


def _row_of_digit(b):
    h, reg = b >> 4, b & 15
    return (reg & 3) + 8 * (reg >> 2) + 4 * h


@pytest.mark.parametrize("field", ["fr", "gl"])
@pytest.mark.parametrize("n,d,t,ids", [(4, 1, 1, [3, 0, 2, 1]), (7, 2, 2, [6, 5, 4, 3, 2]), (10, 3, 3, list(range(10))),
                                       (16, 5, 5, list(range(16))), (31, 10, 10, list(range(31)))])
def test_matrix_core_tables(dump, field, n, d, t, ids):
    """the byte-digit tables of the matrix-core kernels: every slab entry is a balanced digit of c * 256^a mod p for the
    coefficient c the oracle computes, and the accumulator bias is 128 * (sum of the row's entries) plus a multiple of p
    (tables_mfma.hpp, tables_mfma_gl.hpp).  Small shapes are checked digit by digit, the others are built under the
    sanitizers."""
    S = SFR if field == "fr" else SGL
    P = S.R_MOD
    o = dump(field, n, d, t, ids)
    m, needed = d + 1, d + t + 1
    assert "mfma_bytes" in o
    if "mfma" not in o:
        return
    raw = b"".join(int(w, 16).to_bytes(4, "little") for w in o["mfma"][0].split())
    sid = ids   # the dump tool takes the ids in the order given (the library sorts before it builds)
    xs = [S.domain_element(n, i) for i in sid[:m]]
    basis = []
    for i in range(m):
        co = S.lagrange_interpolate(xs, [1 if j == i else 0 for j in range(m)])
        basis.append(co + [0] * (m - len(co)))
    rows = [[S.p_eval(basis[i], S.domain_element(n, sid[s])) for i in range(m)] for s in range(m, needed)]
    rows += [[basis[i][k] for i in range(m)] for k in range(m)]
    sb = lambda v: v - 256 if v >= 128 else v   # noqa: E731
    if field == "fr":
        RB = m * 1024 + 128
        assert len(raw) == len(rows) * RB
        for r, row in enumerate(rows):
            tsum = 0
            for i, c in enumerate(row):
                for a in range(32):
                    T = sum(sb(raw[r * RB + i * 1024 + (_row_of_digit(b) + 32 * (a >> 4)) * 16 + (a & 15)]) << (8 * b) for b in range(32))
                    assert T % P == c * pow(256, a, P) % P, (r, i, a)
                    tsum += T
            bias = [int.from_bytes(raw[r * RB + m * 1024 + 4 * b: r * RB + m * 1024 + 4 * b + 4], "little", signed=True) for b in range(32)]
            assert all(0 < v < (1 << 23) for v in bias)
            assert (sum(v << (8 * b) for b, v in enumerate(bias)) - 128 * tsum) % P == 0, r
    else:
        KS = (8 * m + 31) // 32
        TR = KS * 1024 + 128
        assert len(raw) == ((len(rows) + 3) // 4) * TR
        for r, row in enumerate(rows):
            mt, h, el = r // 4, (r % 4) // 2, r % 2
            tsum = 0
            for i, c in enumerate(row):
                for a in range(8):
                    kk = 8 * i + a
                    s_, ha, j = kk // 32, (kk % 32) // 16, kk % 16
                    T = 0
                    for b in range(8):
                        reg = 8 * el + b
                        mrow = (reg & 3) + 8 * (reg >> 2) + 4 * h
                        T += sb(raw[mt * TR + s_ * 1024 + (mrow + 32 * ha) * 16 + j]) << (8 * b)
                    assert T % P == c * pow(256, a, P) % P, (r, i, a)
                    tsum += T
            off = mt * TR + KS * 1024 + (h * 16 + 8 * el) * 4
            bias = [int.from_bytes(raw[off + 4 * b: off + 4 * b + 4], "little", signed=True) for b in range(8)]
            assert all(0 < v < (1 << 23) for v in bias)
            assert (sum(v << (8 * b) for b, v in enumerate(bias)) - 128 * tsum) % P == 0, r


@pytest.mark.parametrize("n,d,t,ids", [(4, 1, 1, [3, 0, 2, 1]), (7, 2, 2, [6, 5, 4, 3, 2]), (13, 4, 4, list(range(13))),
                                       (16, 5, 5, list(range(16))), (16, 14, 0, list(range(16))), (16, 15, 0, list(range(16))), (31, 10, 10, list(range(31)))])
def test_point_pair_tables_of_the_encode(dump, n, d, t, ids):
    """tables_mfma.hpp::build_mfma_bfly_table (kernels_mfma_bfly.hpp): pair p holds the digit slabs of alpha_p^i and two
    accumulator biases.  Checked here with integers: (1) the slabs are balanced digits of alpha_p^i 256^a; (2) bE + bT is a
    bias of output p and bE - bT a bias of output p + size/2 (whose odd coefficients have the opposite sign); (3) for EVERY
    input the two digit sums stay inside [0, 0xff0000) -- the interval the kernel's carry pass assumes -- by taking the
    extreme of every product digit by digit."""
    S, P = SFR, SFR.R_MOD
    o = dump("fr", n, d, t, ids)
    m = d + 1
    assert "bfly_bytes" in o and int(o["bfly_bytes"][0]) > 0, "the builder could not prove the digit-sum bound (m = 16 only)"
    if "bfly" not in o:
        return
    raw = b"".join(int(w, 16).to_bytes(4, "little") for w in o["bfly"][0].split())
    size = 1
    while size < n:
        size <<= 1
    half = size // 2
    PB = m * 1024 + 256
    assert len(raw) == half * PB
    sb = lambda v: v - 256 if v >= 128 else v   # noqa: E731
    for p in range(half):
        alpha = S.domain_element(n, p)
        dig = [[[sb(raw[p * PB + i * 1024 + (_row_of_digit(b) + 32 * (a >> 4)) * 16 + (a & 15)]) for b in range(32)] for a in range(32)]
               for i in range(m)]  # dig[i][a][b]
        tsum = [0, 0]
        for i in range(m):
            for a in range(32):
                T = sum(v << (8 * b) for b, v in enumerate(dig[i][a]))
                assert T % P == pow(alpha, i, P) * pow(256, a, P) % P, (p, i, a)
                tsum[i & 1] += T
        bE = [int.from_bytes(raw[p * PB + m * 1024 + 4 * b: p * PB + m * 1024 + 4 * b + 4], "little", signed=True) for b in range(32)]
        bT = [int.from_bytes(raw[p * PB + m * 1024 + 128 + 4 * b: p * PB + m * 1024 + 132 + 4 * b], "little", signed=True) for b in range(32)]
        val = lambda v: sum(x << (8 * b) for b, x in enumerate(v))   # noqa: E731
        plus = [e + t_ for e, t_ in zip(bE, bT)]
        minus = [e - t_ for e, t_ in zip(bE, bT)]
        assert (val(plus) - 128 * (tsum[0] + tsum[1])) % P == 0, p
        partner = p + half < n
        if partner:
            assert (val(minus) - 128 * (tsum[0] - tsum[1])) % P == 0, p
        else:
            assert all(v == 0 for v in bT)
        # the signed data bytes s are in [-128, 127]: extremes of s * digit, digit by digit
        for b in range(32):
            lo = [0, 0]
            hi = [0, 0]
            for i in range(m):
                for a in range(32):
                    v = dig[i][a][b]
                    lo[i & 1] += min(-128 * v, 127 * v)
                    hi[i & 1] += max(-128 * v, 127 * v)
            assert 0 <= plus[b] + lo[0] + lo[1] and plus[b] + hi[0] + hi[1] < 0xff0000, (p, b)
            if partner:
                assert 0 <= minus[b] + lo[0] - hi[1] and minus[b] + hi[0] - lo[1] < 0xff0000, (p, b)
