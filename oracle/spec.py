"""Python big-int restatement of the HoneyBadgerMPC Shamir hot path (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED at the stored-bytes level: the reference (Stoffel-Labs/mpc-protocols, Rust on
ark-ff/ark-poly/ark-bls12-381 0.5, none of which is vendored) cannot be compiled or run in this
environment and holds no golden vectors for this path.  What pins this file: (i) the bls12-381 Fr
modulus, generator 7 and the derived 2^32-th root of unity (checked against the published constant
in tests/test_oracle_spec.py), (ii) every literal-input test the reference holds for the path
(SURVEY.md section 4), restated in tests/, (iii) exactness of modular integer arithmetic.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product path (mpc-protocols_amd/) never does.

Every function cites the reference file:line it follows (paths relative to /root/reference/mpc/src).
Polynomials are python lists of ints, lowest degree first, *normalised like ark-poly's
DensePolynomial*: trailing zero coefficients removed, the zero polynomial is [].
"""
from __future__ import annotations

from itertools import combinations  # noqa: F401  (used by tests)

# ark_bls12_381::Fr  (Cargo.toml:17-18,32)
R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
GENERATOR = 7
TWO_ADICITY = 32
TWO_ADIC_ROOT = pow(GENERATOR, (R_MOD - 1) >> TWO_ADICITY, R_MOD)


# Error taxonomy: ffi/c_bindings/share/mod.rs:18-37 (ShareErrorCode) -- the C-ABI reuses these codes.
class ShareErr(Exception):
    code = -1

    def __init__(self, msg=""):
        super().__init__(msg or type(self).__name__)


class InsufficientShares(ShareErr):
    code = 1


class DegreeMismatch(ShareErr):
    code = 2


class IdMismatch(ShareErr):
    code = 3


class InvalidInput(ShareErr):
    code = 4


class TypeMismatch(ShareErr):
    code = 5


class NoSuitableDomain(ShareErr):
    code = 6


class PolynomialOperationError(ShareErr):
    code = 7


class DecodingError(ShareErr):
    code = 8


# ---------------------------------------------------------------------------------------------
# field helpers
# ---------------------------------------------------------------------------------------------
def inv(a: int) -> int:
    return pow(a % R_MOD, -1, R_MOD)


def to_limbs(a: int):
    """Fr -> U256 {u64[4]} least-significant limb first (ffi/c_bindings/mod.rs:17-21,43-49)."""
    return [(a >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def from_limbs(l) -> int:
    return sum(int(x) << (64 * i) for i, x in enumerate(l))


# ---------------------------------------------------------------------------------------------
# evaluation domain  (common/mod.rs:51-68 -> ark_poly GeneralEvaluationDomain::new(n): Radix-2,
# size = n.next_power_of_two(), element(j) = omega_size^j)
# ---------------------------------------------------------------------------------------------
def domain_size(n: int) -> int:
    s = 1
    while s < n:
        s <<= 1
    return s


def domain_omega(n: int) -> int:
    size = domain_size(n)
    log = size.bit_length() - 1
    if log > TWO_ADICITY:
        raise NoSuitableDomain()
    return pow(TWO_ADIC_ROOT, 1 << (TWO_ADICITY - log), R_MOD)


def domain_element(n: int, j: int) -> int:
    return pow(domain_omega(n), j, R_MOD)


# ---------------------------------------------------------------------------------------------
# DensePolynomial semantics (ark-poly 0.5 univariate::DensePolynomial)
# ---------------------------------------------------------------------------------------------
def p_norm(c):
    c = [x % R_MOD for x in c]
    while c and c[-1] == 0:
        c.pop()
    return c


def p_degree(p) -> int:
    """DensePolynomial::degree(): 0 for the zero polynomial."""
    return 0 if not p else len(p) - 1


def p_eval(p, x) -> int:
    acc = 0
    for c in reversed(p):
        acc = (acc * x + c) % R_MOD
    return acc


def p_add(a, b):
    n = max(len(a), len(b))
    return p_norm([(a[i] if i < len(a) else 0) + (b[i] if i < len(b) else 0) for i in range(n)])


def p_sub(a, b):
    n = max(len(a), len(b))
    return p_norm([(a[i] if i < len(a) else 0) - (b[i] if i < len(b) else 0) for i in range(n)])


def p_mul(a, b):
    if not a or not b:
        return []
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            out[i + j] = (out[i + j] + x * y) % R_MOD
    return p_norm(out)


def p_scale(a, s):
    s %= R_MOD
    if not a or s == 0:
        return []
    return p_norm([x * s for x in a])


def p_divmod(a, b):
    """DenseOrSparsePolynomial::divide_with_q_and_r (robust_interpolate.rs:183-193).
    Dividing by the zero polynomial panics in ark-poly; surfaced here as PolynomialOperationError."""
    if not a:
        return [], []
    if not b:
        raise PolynomialOperationError("Dividing by zero polynomial")
    if p_degree(a) < p_degree(b):
        return [], list(a)
    q = [0] * (len(a) - len(b) + 1)
    rem = list(a)
    lead_inv = inv(b[-1])
    while rem and len(rem) >= len(b):
        cq = rem[-1] * lead_inv % R_MOD
        d = len(rem) - len(b)
        q[d] = cq
        for i, y in enumerate(b):
            rem[d + i] = (rem[d + i] - cq * y) % R_MOD
        rem = p_norm(rem)
    return p_norm(q), rem


def poly_derivative(p):
    """robust_interpolate.rs:164-178."""
    if len(p) <= 1:
        return []
    return p_norm([i * c for i, c in enumerate(p)][1:])


# ---------------------------------------------------------------------------------------------
# shares
# ---------------------------------------------------------------------------------------------
class Share:
    """ShamirShare<F,1,P> (common/mod.rs:92-99): value, id, degree."""

    __slots__ = ("v", "id", "degree")

    def __init__(self, v, id, degree):
        self.v, self.id, self.degree = v % R_MOD, id, degree

    def __repr__(self):
        return f"Share({hex(self.v)}, id={self.id}, deg={self.degree})"

    def __eq__(self, o):
        return (self.v, self.id, self.degree) == (o.v, o.id, o.degree)


def share_add(a: Share, b: Share) -> Share:  # common/mod.rs:167-188
    if a.degree != b.degree:
        raise DegreeMismatch()
    if a.id != b.id:
        raise IdMismatch()
    return Share(a.v + b.v, a.id, a.degree)


def share_sub(a: Share, b: Share) -> Share:  # common/mod.rs:220-240
    if a.degree != b.degree:
        raise DegreeMismatch()
    if a.id != b.id:
        raise IdMismatch()
    return Share(a.v - b.v, a.id, a.degree)


def share_add_scalar(a: Share, s: int) -> Share:  # :205-218
    return Share(a.v + s, a.id, a.degree)


def share_sub_scalar(a: Share, s: int) -> Share:  # :242-254
    return Share(a.v - s, a.id, a.degree)


def share_from_scalar_sub(s: int, a: Share) -> Share:  # :255-265
    return Share(s - a.v, a.id, a.degree)


def share_mul_scalar(a: Share, s: int) -> Share:  # :267-280
    return Share(a.v * s, a.id, a.degree)


def share_mul(a: Share, b: Share) -> Share:  # :282-300 (degree adds; only the id is checked)
    if a.id != b.id:
        raise IdMismatch()
    return Share(a.v * b.v, a.id, a.degree + b.degree)


# ---------------------------------------------------------------------------------------------
# a3  compute_shares  (robust_interpolate.rs:52-82; shamir.rs:158-196)
# coeffs[0] is the secret, coeffs[1..=degree] are the rng draws (inputs here: SURVEY.md section 7
# "Randomness").  domain.fft(poly) == evaluations at element(0..size); the first n are the shares.
# ---------------------------------------------------------------------------------------------
def compute_shares(coeffs, n: int, degree: int):
    if n <= degree:
        raise InvalidInput("n must be greater than degree")
    assert len(coeffs) == degree + 1
    w = domain_omega(n)
    return [Share(p_eval(coeffs, pow(w, j, R_MOD)), j, degree) for j in range(n)]


# ---------------------------------------------------------------------------------------------
# a4/a5  Vandermonde  (common/share/mod.rs:31-76)
# ---------------------------------------------------------------------------------------------
def make_vandermonde(n: int, t: int):
    w = domain_omega(n)
    m = []
    for j in range(n):
        a = pow(w, j, R_MOD)
        row, p = [], 1
        for _ in range(t + 1):
            row.append(p)
            p = p * a % R_MOD
        m.append(row)
    return m


def apply_vandermonde(vdm, shares):
    for row in vdm:
        if len(row) != len(shares):
            raise InvalidInput()
    out = []
    for row in vdm:
        acc = share_mul_scalar(shares[0], row[0])
        for a, b in list(zip(row, shares))[1:]:
            acc = share_add(acc, share_mul_scalar(b, a))
        out.append(acc)
    return out


# ---------------------------------------------------------------------------------------------
# textbook Lagrange (common/mod.rs:134-165)
# ---------------------------------------------------------------------------------------------
def lagrange_interpolate(xs, ys):
    if len(xs) != len(ys):
        raise InvalidInput()
    if len(set(xs)) != len(xs):
        raise InvalidInput()
    res = []
    for j in range(len(xs)):
        num, den = [1], 1
        for m in range(len(xs)):
            if m != j:
                num = p_mul(num, [(-xs[m]) % R_MOD, 1])
                den = den * (xs[j] - xs[m]) % R_MOD
        res = p_add(res, p_mul(num, p_norm([ys[j] * inv(den)])))
    return res


# ---------------------------------------------------------------------------------------------
# a6  robust_interpolate_fnt / recover_secret  (robust_interpolate.rs:94-157, 206-266)
# ---------------------------------------------------------------------------------------------
def robust_interpolate_fnt(t: int, n: int, shares):
    degree = shares[0].degree
    subset = shares[: degree + 1]
    xs = [domain_element(n, s.id) for s in subset]
    ys = [s.v for s in subset]
    a_poly = [1]
    for x in xs:
        a_poly = p_mul(a_poly, [(-x) % R_MOD, 1])
    a_der = poly_derivative(a_poly)
    interp = []
    for i, x_i in enumerate(xs):
        denom = p_eval(a_der, x_i)
        if denom == 0:
            raise PolynomialOperationError("zero denominator")
        scalar = ys[i] * inv(denom) % R_MOD
        basis, rem = p_divmod(a_poly, [(-x_i) % R_MOD, 1])
        if rem:
            raise PolynomialOperationError("A(x) not divisible")
        interp = p_add(interp, p_scale(basis, scalar))
    valid = sum(1 for s in shares if p_eval(interp, domain_element(n, s.id)) == s.v)
    if valid >= degree + t + 1:
        return interp
    raise DecodingError("Not enough shares matched the interpolated polynomial")


def recover_secret(shares, n: int, t: int):
    """Returns (coeffs trimmed like DensePolynomial, P(0)).  Validation order per :100-142."""
    if n < 3 * t + 1:
        raise InvalidInput("n < 3t+1")
    if not shares:
        raise InvalidInput("empty")
    degree = shares[0].degree
    if any(s.degree != degree for s in shares):
        raise DegreeMismatch()
    if len({s.id for s in shares}) != len(shares):
        raise InvalidInput("duplicate ids")
    if any(s.id >= n for s in shares):
        raise InvalidInput("id out of range")
    if len(shares) < degree + t + 1:
        raise InvalidInput("not enough shares")
    srt = sorted(shares, key=lambda s: s.id)
    try:
        poly = robust_interpolate_fnt(t, n, srt[: degree + t + 1])
        return list(poly), p_eval(poly, 0)
    except ShareErr:
        pass
    poly, at0 = oec_decode(n, t, srt)
    return list(poly), at0


# ---------------------------------------------------------------------------------------------
# a8  Gao / OEC  (robust_interpolate.rs:456-628)
# ---------------------------------------------------------------------------------------------
def compute_g0_from_domain(n: int):  # :540-565
    g0 = [1]
    for i in range(n):
        g0 = p_mul(g0, [(-domain_element(n, i)) % R_MOD, 1])
    return g0


def gao_rs_decode(received, k: int, n: int, erasure_positions):
    if k > n:
        raise InvalidInput("k > n")
    s_set = set(erasure_positions)
    s = len(s_set)
    s_poly = [1]
    for i in s_set:
        s_poly = p_mul(s_poly, [(-domain_element(n, i)) % R_MOD, 1])
    known = [(domain_element(n, i), received[i] % R_MOD) for i in range(n) if i not in s_set]
    g1 = lagrange_interpolate([x for x, _ in known], [y for _, y in known])
    g0, _ = p_divmod(compute_g0_from_domain(n), s_poly)
    threshold = (n - s + k) // 2
    r0, r1 = list(g0), list(g1)
    t0, t1 = [], [1]
    while p_degree(r1) >= threshold:
        q, _ = p_divmod(r0, r1)
        r = p_sub(r0, p_mul(q, r1))
        tt = p_sub(t0, p_mul(q, t1))
        r0, r1 = r1, r
        t0, t1 = t1, tt
    g, v = r1, t1
    quotient, _ = p_divmod(g, v)
    remainder = p_sub(g, p_mul(quotient, v))
    if not remainder and p_degree(quotient) < k:
        return list(quotient)
    raise DecodingError("Failed to recover message polynomial from g(x)/v(x)")


def oec_decode(n: int, t: int, shares):
    degree = shares[0].degree
    for r in range(1, t + 1):
        required = degree + t + 1 + r
        if len(shares) < required:
            break
        subset = shares[:required]
        received = [0] * n
        erasures = []
        for i in range(n):
            hit = next((s for s in subset if s.id == i), None)
            if hit is not None:
                received[i] = hit.v
            else:
                erasures.append(i)
        try:
            coeffs = gao_rs_decode(received, degree + 1, n, erasures)
        except ShareErr:
            continue
        poly = p_norm(coeffs)
        matched = sum(1 for s in subset if p_eval(poly, domain_element(n, s.id)) == s.v)
        if matched >= degree + t + 1:
            return poly, p_eval(poly, 0)
    raise DecodingError("Online Error Correction failed to find a valid polynomial")


# ---------------------------------------------------------------------------------------------
# a7  batch_recover_secret  (robust_interpolate.rs:284-443)
# evals_by_sender: list of (sender_id, [value per chunk]).  Returns one coefficient list per
# chunk: length degree+1 on the optimistic path (:419), *trimmed* on the fallback path (:437-438).
# ---------------------------------------------------------------------------------------------
def batch_basis(ids_sorted, n: int, degree: int, t: int):
    """The shared tables of :343-399: (basis_coeffs[m][<=m], verify_matrix[needed][m])."""
    m = degree + 1
    needed = degree + t + 1
    xs = [domain_element(n, ids_sorted[i]) for i in range(m)]
    a_poly = [1]
    for x in xs:
        a_poly = p_mul(a_poly, [(-x) % R_MOD, 1])
    a_der = poly_derivative(a_poly)
    basis = []
    for x_i in xs:
        denom = p_eval(a_der, x_i)
        if denom == 0:
            raise PolynomialOperationError("zero denominator")
        bp, rem = p_divmod(a_poly, [(-x_i) % R_MOD, 1])
        if rem:
            raise PolynomialOperationError("A(x) not divisible")
        basis.append(p_scale(bp, inv(denom)))
    vxs = [domain_element(n, ids_sorted[s]) for s in range(needed)]
    verify = [[p_eval(basis[i], vxs[s]) for i in range(m)] for s in range(needed)]
    return basis, verify


def batch_recover_secret(evals_by_sender, n: int, degree: int, t: int):
    if n < 3 * t + 1:
        raise InvalidInput("n < 3t+1")
    if not evals_by_sender:
        raise InvalidInput("No evaluations provided")
    batch_len = len(evals_by_sender[0][1])
    if batch_len == 0:
        raise InvalidInput("Empty batch")
    if any(len(v) != batch_len for _, v in evals_by_sender):
        raise InvalidInput("Inconsistent batch widths")
    srt = sorted(evals_by_sender, key=lambda e: e[0])
    seen = set()
    for sid, _ in srt:
        if sid in seen:
            raise InvalidInput("Duplicate sender id")
        seen.add(sid)
        if sid >= n:
            raise InvalidInput("Sender id out of range")
    needed = degree + t + 1
    if len(srt) < needed:
        raise InvalidInput("Not enough evaluations")
    m = degree + 1
    basis, verify = batch_basis([sid for sid, _ in srt], n, degree, t)
    results = []
    for c in range(batch_len):
        ok = True
        for s in range(needed):
            acc = sum(verify[s][i] * srt[i][1][c] for i in range(m)) % R_MOD
            if acc != srt[s][1][c] % R_MOD:
                ok = False
                break
        if ok:
            results.append([
                sum((basis[i][k] if k < len(basis[i]) else 0) * srt[i][1][c] for i in range(m)) % R_MOD
                for k in range(degree + 1)
            ])
        else:
            shares = [Share(vals[c], sid, degree) for sid, vals in srt]
            coeffs, _ = recover_secret(shares, n, t)
            results.append(coeffs)
    return results


# ---------------------------------------------------------------------------------------------
# NonRobustShare::recover_secret  (shamir.rs:199-239) -- RanDouSha verifier path (section 8(f) row 3)
# ---------------------------------------------------------------------------------------------
def nonrobust_recover_secret(shares, n: int):
    if not shares:
        raise InvalidInput()
    if len({s.id for s in shares}) != len(shares):
        raise InvalidInput()
    deg = shares[0].degree
    if any(s.degree != deg for s in shares):
        raise DegreeMismatch()
    if len(shares) < deg + 1:
        raise InsufficientShares()
    if any(s.id >= n for s in shares):
        raise InvalidInput()
    poly = lagrange_interpolate([domain_element(n, s.id) for s in shares], [s.v for s in shares])
    if p_degree(poly) > deg:
        raise DegreeMismatch()
    return list(poly), (poly[0] if poly else 0)  # reference indexes poly[0] (panics on zero poly)


# ---------------------------------------------------------------------------------------------
# Preprocessing producers: RanSha (share_gen/share_gen.rs), DouSha (double_share/double_share_generation.rs),
# RanDouSha (ran_dou_sha/mod.rs) -- the arithmetic of all n parties, messages replaced by lists.
# The dealers' polynomials are inputs (the reference draws them from each party's rng): coeffs[p][k] = coefficient list
# of dealer p's k-th polynomial, coeffs[p][k][0] the secret.
# ---------------------------------------------------------------------------------------------
def _deal_and_mix(coeffs, n: int, deg: int):
    """-> r[j][k] = the n shares [r_0 .. r_{n-1}] party j holds for batch element k: every dealer's compute_shares
    (share_gen.rs:251-259 / double_share_generation.rs:168-181), sorted by dealer id (:375-384), then
    apply_vandermonde(make_vandermonde(n, n - 1), .) per batch element (:415-419 / ran_dou_sha/mod.rs:392-403)."""
    K = len(coeffs[0])
    dealt = [[compute_shares(coeffs[p][k], n, deg) for k in range(K)] for p in range(n)]   # dealt[p][k][j]
    vdm = make_vandermonde(n, n - 1)
    return [[apply_vandermonde(vdm, [dealt[p][k][j] for p in range(n)]) for k in range(K)] for j in range(n)]


def ransha(coeffs, n: int, t: int, verify_senders=None):
    """-> (outputs, ok): outputs[j] = party j's (n - 2t) K random sharings in the reference's order (try_finalize,
    share_gen.rs:199-203); ok[i] = verifier i's verdict, i < 2t: recover_secret of every batch element from the shares of
    the first `verify_senders` parties (the handler fires at 2t + 1 received, :497) and the exact-degree test (:516-530)."""
    verify_senders = 2 * t + 1 if verify_senders is None else verify_senders
    K = len(coeffs[0])
    r = _deal_and_mix(coeffs, n, t)
    ok = []
    for i in range(2 * t):
        good = True
        for k in range(K):
            shares = [r[j][k][i] for j in range(verify_senders)]
            try:
                poly, _ = recover_secret([Share(sh.v, j, t) for j, sh in enumerate(shares)], n, t)
                if p_degree(poly) != t:
                    good = False
            except ShareErr:
                good = False
            if not good:
                break
        ok.append(good)
    outputs = [[sh.v for k in range(K) for sh in r[j][k][2 * t:]] for j in range(n)]
    return outputs, ok


def randousha(coeffs_t, coeffs_2t, n: int, t: int):
    """-> (out_t, out_2t, ok): party j's (t + 1) K double sharings (try_finalize, ran_dou_sha/mod.rs:314-331) and the
    verdicts of verifiers t + 1 .. n - 1: NonRobustShare::recover_secret of both polynomials through all n shares, exact
    degrees t and 2t, equal secrets (:569-602)."""
    K = len(coeffs_t[0])
    rt, r2t = _deal_and_mix(coeffs_t, n, t), _deal_and_mix(coeffs_2t, n, 2 * t)
    ok = []
    for i in range(t + 1, n):
        good = True
        for k in range(K):
            try:
                p1, s1 = nonrobust_recover_secret([Share(rt[j][k][i].v, j, t) for j in range(n)], n)
                p2, s2 = nonrobust_recover_secret([Share(r2t[j][k][i].v, j, 2 * t) for j in range(n)], n)
                if p_degree(p1) != t or p_degree(p2) != 2 * t or s1 != s2:
                    good = False
            except ShareErr:
                good = False
            if not good:
                break
        ok.append(good)
    out_t = [[sh.v for k in range(K) for sh in rt[j][k][: t + 1]] for j in range(n)]
    out_2t = [[sh.v for k in range(K) for sh in r2t[j][k][: t + 1]] for j in range(n)]
    return out_t, out_2t, ok


# ---------------------------------------------------------------------------------------------
# a11  triple_gen local math  (triple_gen/triple_generation.rs:333-340, 196-208)
# ---------------------------------------------------------------------------------------------
def triple_local(a: Share, b: Share, r2t: Share) -> Share:
    return share_sub(share_mul(a, b), r2t)


def triple_finalize(rt: Share, opened: int) -> Share:
    return share_add_scalar(rt, opened)


# ---------------------------------------------------------------------------------------------
# a12  Beaver mul local math  (mul/multiplication.rs:417-426, 57-100)
# ---------------------------------------------------------------------------------------------
def beaver_open_shares(a: Share, b: Share, x: Share, y: Share):
    return share_sub(a, x), share_sub(b, y)


def beaver_finalize(c: Share, x: Share, y: Share, d: int, e: int) -> Share:
    """z = c - d*e - d*[y] - e*[x]   (d = a-x opened, e = b-y opened)."""
    s = share_sub_scalar(c, d * e % R_MOD)
    s2 = share_sub(s, share_mul_scalar(y, d))
    return share_sub(s2, share_mul_scalar(x, e))


# ---------------------------------------------------------------------------------------------
# a13  TruncPr local math  (fpmul/truncpr.rs:277-297, 215-220; fpmul/mod.rs:377-406)
# ---------------------------------------------------------------------------------------------
def pow2_f(e: int) -> int:
    return pow(2, e, R_MOD)


def mod_pow_2_from_field(x: int, m: int) -> int:
    """Low m bits of the canonical integer (byte-truncate then mask, then from_le_bytes_mod_order)."""
    b = bytearray((x % R_MOD).to_bytes(32, "little"))
    full, extra = divmod(m, 8)
    if extra > 0 and full >= 32:
        raise InvalidInput("bytes[full_bytes] is out of bounds in the reference (it panics)")
    usable = full if extra == 0 else full + 1
    if len(b) > usable:
        b = b[:usable]
    if extra > 0 and len(b) > 0:
        b[full] &= (1 << extra) - 1
    return int.from_bytes(bytes(b), "little") % R_MOD


def truncpr_rdash(r_bits, m: int, party_id: int, t: int) -> Share:
    acc = Share(0, party_id, t)
    for i, bit in enumerate(r_bits[:m]):
        acc = share_add(acc, share_mul_scalar(bit, pow2_f(i)))
    return acc


def truncpr_open_share(a: Share, r_dash: Share, r_int: Share, k: int, m: int) -> Share:
    b = share_add_scalar(a, pow2_f(k - 1))
    r = share_add(share_mul_scalar(r_int, pow2_f(m)), r_dash)
    return share_add(b, r)


def truncpr_finalize(a: Share, r_dash: Share, c_open: int, m: int) -> Share:
    c_mod = mod_pow_2_from_field(c_open, m)
    a_prime = share_from_scalar_sub(c_mod, r_dash)
    return share_mul_scalar(share_sub(a, a_prime), inv(pow2_f(m)))


# ---------------------------------------------------------------------------------------------
# deterministic synthetic inputs: SplitMix64 -> 4 limbs -> mod r
# ---------------------------------------------------------------------------------------------
class SplitMix64:
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def fr(self):
        return from_limbs([self.next() for _ in range(4)]) % R_MOD


# ---------------------------------------------------------------------------------------------
# Seeded coefficient generation (this library's own contract, "hbmpc-chacha20-v1"; NOT a restatement of the
# reference: there the caller's `rng: &mut impl Rng` supplies DensePolynomial::rand, robust_interpolate.rs:68).
# The random coefficient k (1 <= k <= degree) of secret number b is a function of (seed, b, k) only, so any
# number of devices / lanes produce the same polynomial:
#   ChaCha20 (20 rounds, the original 64-bit-counter / 64-bit-nonce layout), key = the 32-byte seed,
#   nonce = b, block counter = (k << 32) + attempt;  a 64-byte block is read as little-endian candidates of
#   ELEM_BYTES bytes each (32 for Fr with bit 255 cleared, 8 for Goldilocks); the first candidate < modulus over
#   attempt = 0, 1, ... is the coefficient.
# ---------------------------------------------------------------------------------------------
def _chacha20_block(key_words, counter, nonce):
    def rotl(v, c):
        return ((v << c) & 0xFFFFFFFF) | (v >> (32 - c))

    def qr(s, a, b, c, d):
        s[a] = (s[a] + s[b]) & 0xFFFFFFFF; s[d] = rotl(s[d] ^ s[a], 16)
        s[c] = (s[c] + s[d]) & 0xFFFFFFFF; s[b] = rotl(s[b] ^ s[c], 12)
        s[a] = (s[a] + s[b]) & 0xFFFFFFFF; s[d] = rotl(s[d] ^ s[a], 8)
        s[c] = (s[c] + s[d]) & 0xFFFFFFFF; s[b] = rotl(s[b] ^ s[c], 7)

    init = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + \
           [counter & 0xFFFFFFFF, counter >> 32, nonce & 0xFFFFFFFF, nonce >> 32]
    s = list(init)
    for _ in range(10):
        qr(s, 0, 4, 8, 12); qr(s, 1, 5, 9, 13); qr(s, 2, 6, 10, 14); qr(s, 3, 7, 11, 15)
        qr(s, 0, 5, 10, 15); qr(s, 1, 6, 11, 12); qr(s, 2, 7, 8, 13); qr(s, 3, 4, 9, 14)
    return [(x + y) & 0xFFFFFFFF for x, y in zip(s, init)]


def seeded_coefficient(seed_words, b: int, k: int) -> int:
    """seed_words: eight u32 (the 32-byte seed, little-endian words)"""
    ebytes = 32 if R_MOD.bit_length() > 64 else 8
    attempt = 0
    while True:
        blk = _chacha20_block(seed_words, (k << 32) + attempt, b)
        raw = b"".join(w.to_bytes(4, "little") for w in blk)
        for c in range(64 // ebytes):
            v = int.from_bytes(raw[c * ebytes:(c + 1) * ebytes], "little")
            if ebytes == 32:
                v &= (1 << 255) - 1
            if v < R_MOD:
                return v
        attempt += 1


def seeded_polynomial(seed_words, b: int, secret, degree: int):
    """secret None: coefficient 0 is drawn from the stream too (position k = 0)"""
    c0 = seeded_coefficient(seed_words, b, 0) if secret is None else secret % R_MOD
    return [c0] + [seeded_coefficient(seed_words, b, k) for k in range(1, degree + 1)]
