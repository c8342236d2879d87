"""Device-resident replays of the reference's arithmetic pipelines for ALL n simulated parties on one
GPU (how every reference test/bench runs: n parties in one process on FakeNetwork).  Host-side
orchestration only -- every arithmetic step is an hbmpc_dev_* call; the parties' all-to-all is a
layout choice (strided sender rows), nothing is copied between "parties".

  triple_gen   TripleGenNode::init_batch + BatchReconNode (degree 2t) + try_finalize_triple_gen
               triple_gen/triple_generation.rs:304-364,164-232; batch_recon/batch_recon.rs:144-185,332-481
  fpmul        FPMulNode::init = Multiply (Beaver, RBC path) + TruncPrNode
               fpmul/fpmul.rs:61-110, mul/multiplication.rs:417-426,57-139, fpmul/truncpr.rs:185-318
  ransha       RanShaNode::init_batch + init_ransha_batch + reconstruction_handler + try_finalize: the random degree-t
               sharings triple_gen consumes as a, b            share_gen/share_gen.rs:232-289,401-454,516-530,199-203
  randousha    DouShaNode::init_batch + RanDouShaNode::init_batch + reconstruction_handler + try_finalize: the double
               sharings ([r]_t, [r]_2t)     double_share/double_share_generation.rs:151-215, ran_dou_sha/mod.rs:371-449,569-602,314-331
  preprocessing  run_preprocessing's triple part (honeybadger/mod.rs:1239-1393): ransha -> a, b; randousha -> r; triple_gen
"""
from __future__ import annotations

import numpy as np

U = 32  # bytes per element


class DeviceArena:
    """bump allocator over one hbmpc_dev_alloc block (keeps the pipelines free of per-step mallocs)"""

    def __init__(self, eng, nbytes):
        self.eng, self.size, self.off = eng, nbytes, 0
        self.base = eng.dev_alloc(nbytes)

    def take(self, nbytes):
        nbytes = (nbytes + 255) & ~255
        assert self.off + nbytes <= self.size, "arena exhausted"
        p = self.base + self.off
        self.off += nbytes
        return p

    def free(self):
        self.eng.dev_free(self.base)


def _check(rc, eng, what):
    if rc != 0:
        raise RuntimeError(f"{what} -> ShareErrorCode {rc}: {eng.last_error()}")


def _summary_ok(eng, smd, what, stream=0):
    s = np.zeros(4, dtype=np.uint32)
    eng.d2h(s, smd, stream)
    eng.sync(stream)
    if s[1] != 0:
        raise RuntimeError(f"{what}: {s[1]} chunks failed to decode (first {s[2]}, error {s[3]})")
    return int(s[0])


class _Capturable:
    """run(check=False) only enqueues hbmpc_dev_* calls; after one eager run (tables and scratch then exist) the same
    sequence can be captured into a HIP graph and replayed, which removes the per-launch overhead that dominates at
    the small batch sizes the protocols actually use."""
    graph = None

    def capture(self):
        assert self.stream, "capture needs an explicit stream"
        # two eager runs: nothing can be built during capture, and a mid-size decode builds its matrix-core table the
        # second time it sees a sender set (hbmpc_set_matrix_cores) -- the recorded launches are then the ones an eager
        # caller gets from its second call on
        self.run(check=False)
        self.run(check=False)
        self.eng.sync(self.stream)
        self.eng.graph_begin(self.stream)
        try:
            self.run(check=False)
        finally:
            self.graph = self.eng.graph_end(self.stream)

    def replay(self):
        self.eng.graph_launch(self.graph, self.stream)

    def _drop_graph(self):
        if self.graph:
            self.eng.graph_destroy(self.graph)
            self.graph = None


class TripleGen(_Capturable):
    """n parties, threshold t, N triples (N a multiple of 2t+1).  Buffers are [party][N] canonical.
    Works in either field: the reference runs TripleGenNode over Fr and, in PreprocNodesSmallField
    (honeybadger/mod.rs:316-324), over GoldilocksField -- the element size follows the engine's field."""

    def __init__(self, eng, n, t, N, stream=0):
        assert N % (2 * t + 1) == 0
        self.eng, self.n, self.t, self.N, self.stream = eng, n, t, N, stream
        self.m = 2 * t + 1
        U = 32 if eng.field == "fr" else 8  # bytes per element
        self.G = N // self.m
        G = self.G
        self.arena = DeviceArena(eng, (5 * n * N + n * n * G + n * G + N) * U + (n + 2) * G + (1 << 14))
        ar = self.arena
        self.a, self.b, self.r2t, self.rt, self.c = (ar.take(n * N * U) for _ in range(5))
        self.Y = ar.take(n * n * G * U)      # Y[p][j][g]: party p's evaluation for recipient j
        self.Z = ar.take(n * G * U)          # Z[j][g]: recipient j's opened y_j (the broadcast RevealBatch)
        self.opened = ar.take(N * U)         # [G][2t+1] == flat [N]
        self.status = ar.take(n * G)
        self.summ = ar.take(64)

    def upload(self, a, b, r2t, rt):
        for dst, src in ((self.a, a), (self.b, b), (self.r2t, r2t), (self.rt, rt)):
            self.eng.h2d(dst, np.ascontiguousarray(src), self.stream)

    def run(self, check=True):
        e, n, t, N, G, m, s = self.eng, self.n, self.t, self.N, self.G, self.m, self.stream
        d = 2 * t
        ids = list(range(n))
        # 1 + 2. every party: [ab - r]_2t = a_i * b_i - r2t_i (triple_generation.rs:333-340), Vandermonde-encoded in chunks of
        #    2t+1 -> y for each recipient (batch_recon.rs:157-165): Y[party][n][G], ONE launch for all parties -- the local
        #    products never touch HBM where the fused kernel covers the shape (c is the workspace of the two-launch path)
        _check(e.dev_triple_encode_parties(self.a, self.b, self.r2t, G, n, d, n, self.c, self.Y, s), e, "local product + encode")
        # 3. EvalBatch arm: recipient j interpolates its y_j from the senders' evaluations (needs d+t+1 = 3t+1).
        #    With Y[p][j][g] the row of sender p for "chunk" c = j G + g is Y + p (n G) + c: ONE strided decode over
        #    n G chunks is all n recipients at once, and its output Z[c] is already Z[j][g].
        _check(e.dev_batch_recover_strided(ids, self.Y, n * G, n * G, n, d, t, self.Z, p0=True, status_d=self.status,
                                           summary_d=self.summ, stream=s), e, "decode y_j (all recipients)")
        if check:
            _summary_ok(e, self.summ, "EvalBatch decode", s)
        # 4. RevealBatch arm: everyone interpolates the 2t+1 opened values per chunk from the n broadcast y_j
        _check(e.dev_batch_recover(ids, self.Z, G, n, d, t, self.opened, 0, self.status, self.summ, s), e, "decode open")
        if check:
            _summary_ok(e, self.summ, "RevealBatch decode", s)
        # 5. every party: [c]_t = rt_i + opened   (triple_generation.rs:196-208): one party-batched launch
        _check(e.dev_elem_parties("triple_finalize", [self.rt, self.opened, self.c], N, n, stream=s), e, "triple_finalize")

    def download_c(self):
        out = self.eng._new((self.n, self.N))
        self.eng.d2h(out, self.c, self.stream)
        self.eng.sync(self.stream)
        return out

    def close(self):
        self._drop_graph()
        self.arena.free()


class FpMul(_Capturable):
    """Fixed-point multiplication of N element pairs for n parties: Beaver mul (a-x, b-y opened by direct
    robust interpolation, i.e. the RBC path of Multiply::init for < t+1 leftovers that FPMulNode always
    takes) followed by TruncPr with k-bit values and m fractional bits.
    open_senders: how many parties' shares an open interpolates from.  Default 2t+1: the reference opens as soon as that
    many have arrived (multiplication.rs:388,617 `received_shares.len() >= 2 * self.t + 1`, truncpr.rs:202
    `open_buf.len() < 2 * self.t + 1`), i.e. from the first 2t+1 senders -- with d = t that is exactly d + t + 1, a decode
    with no OEC round, which the library runs as one launch.  n = every party's share (OEC rounds available)."""

    def __init__(self, eng, n, t, N, k, m, stream=0, open_senders=None):
        self.eng, self.n, self.t, self.N, self.k, self.m, self.stream = eng, n, t, N, k, m, stream
        self.open_senders = 2 * t + 1 if open_senders is None else open_senders
        assert 2 * t + 1 <= self.open_senders <= n
        self.arena = DeviceArena(eng, ((12 + m) * n * N + 4 * N) * U + 8 * N + (1 << 14))
        ar = self.arena
        (self.x, self.y, self.ta, self.tb, self.tc, self.rint, self.z, self.rdash, self.osh,
         self.out) = (ar.take(n * N * U) for _ in range(10))
        self.desh = ar.take(2 * n * N * U)    # [party][2][N]: a party's shares of a - x and of b - y side by side
        self.rbits = ar.take(n * m * N * U)   # [party][bit][N]
        self.deop = ar.take(2 * N * U)        # the opened a - x [N], then the opened b - y [N]
        self.dop, self.eop = self.deop, self.deop + N * U
        self.cop = ar.take(N * U)
        self.status = ar.take(2 * N)
        self.summ = ar.take(64)

    def upload(self, x, y, ta, tb, tc, rbits, rint):
        for dst, src in ((self.x, x), (self.y, y), (self.ta, ta), (self.tb, tb), (self.tc, tc), (self.rbits, rbits),
                         (self.rint, rint)):
            self.eng.h2d(dst, np.ascontiguousarray(src), self.stream)

    def _open(self, shares, out, what, values=None):
        e, n, t, s = self.eng, self.n, self.t, self.stream
        N = self.N if values is None else values      # values per sender row
        _check(e.dev_batch_recover(list(range(self.open_senders)), shares, N, n, t, t, out, 0, self.status, self.summ, s, p0=True), e, what)
        if self.check:
            _summary_ok(e, self.summ, what, s)

    def run(self, check=True):
        """check=False enqueues only (no copy-back, no sync): the whole pipeline is then capturable into a HIP graph
        once tables and scratch exist (after one eager run)."""
        self.check = check
        e, n, N, k, m, s = self.eng, self.n, self.N, self.k, self.m, self.stream
        # the [party][N] arrays are contiguous: one launch per step for ALL parties (public operands broadcast)
        _check(e.dev_beaver_open_shares_paired(self.ta, self.tb, self.x, self.y, N, n, self.desh, s), e,
               "open shares")                         # multiplication.rs:417-426
        # reconstruct_rbc: per-element recover_secret of a - x and of b - y (:102-139) -- ONE interpolation call over the
        # 2 N values of every sender row
        self._open(self.desh, self.deop, "open a-x, b-y", values=2 * N)
        # finalize_mul (multiplication.rs:57-100), r' and the share TruncPr opens (truncpr.rs:277-297): one launch
        _check(e.dev_fpmul_middle(self.tc, self.x, self.y, self.dop, self.eop, self.rbits, self.rint, k, m, N, n, self.z,
                                  self.rdash, self.osh, s), e, "finalize_mul + r' + truncpr open share")
        self._open(self.osh, self.cop, "open b+r")   # truncpr.rs:215
        _check(e.dev_elem_parties("truncpr_finalize", [self.z, self.rdash, self.cop, self.out], N, n, extra=(m,), stream=s), e,
               "truncpr finalize")                    # truncpr.rs:216-220

    def download(self, which="out"):
        out = np.zeros((self.n, self.N, 4), dtype=np.uint64)
        self.eng.d2h(out, getattr(self, which), self.stream)
        self.eng.sync(self.stream)
        return out

    def close(self):
        self._drop_graph()
        self.arena.free()


class _Producer(_Capturable):
    """What RanSha and RanDouSha share: every dealer p deals K secrets to the n recipients (the dealers' polynomials are
    the INPUT: coefficient rows [dealer][K][deg + 1], column 0 the secret -- filled by the host, or on the device by
    hbmpc_dev_fill_coeffs: the reference draws them from each party's rng), recipient j multiplies the vector of the n
    shares it received for batch element k by the n x n Vandermonde matrix make_vandermonde(n, n - 1), row i of the
    result goes to verifier i, and the other rows are the party's output.

    Layouts (elements; all n parties on one device):  dealt S[p][j, k]  --Vandermonde over the rows p-->  y[i][j, k]; what
    party j sends verifier i is y[i][j K .. j K + K): a strided sender row, nothing is copied."""

    def __init__(self, eng, n, t, K, stream=0):
        self.eng, self.n, self.t, self.K, self.stream = eng, n, t, K, stream
        self.U = 32 if eng.field == "fr" else 8

    def _deal(self, coeffs, deg, S):
        """coeffs [n dealers][K][deg + 1] -> S[dealer][recipient][K]: every dealer's compute_shares"""
        e, n, K, s, U = self.eng, self.n, self.K, self.stream, self.U
        for p in range(n):  # dealer p: RobustShare / NonRobustShare::compute_shares for each of its K secrets
            _check(e.dev_compute_shares(coeffs + p * K * (deg + 1) * U, K, n, deg, S + p * n * K * U, s), e, "deal")

    def _mix(self, S, x, y):
        """y[i][j, k] = sum_p alpha_i^p * S[p][j, k]: what every recipient computes from the n shares it was dealt"""
        e, n, K, s = self.eng, self.n, self.K, self.stream
        # the share of dealer p for (recipient j, element k) is row p of S: the n x n map reads the dealers' outputs in place
        # (hbmpc_dev_vandermonde_apply_rows; x is the workspace of the shapes that have to be transposed first)
        _check(e.dev_vandermonde_apply_rows(S, n * K, n * K, n, n - 1, x, y, s), e, "n x n Vandermonde over the dealt shares")

    def _bad(self):
        b = np.zeros(2, dtype=np.uint32)
        self.eng.d2h(b, self.bad, self.stream)
        self.eng.sync(self.stream)
        return int(b[0]), int(b[1])

    def _clear_bad(self):
        self.eng.h2d(self.bad, np.array([0, 0xFFFFFFFF], dtype=np.uint32), self.stream)


class RanSha(_Producer):
    """K batch elements per dealer -> (n - 2t) K random degree-t sharings per party, verified by parties 0 .. 2t - 1.
    verify_senders: how many parties' shares a verifier reconstructs from.  Default 2t + 1: the reference's handler fires
    as soon as that many have arrived (share_gen.rs:497: `received_r_shares.len() >= 2 * self.threshold + 1`), i.e. the
    first 2t + 1 senders -- with degree t that is a decode with no OEC round."""

    def __init__(self, eng, n, t, K, stream=0, verify_senders=None):
        super().__init__(eng, n, t, K, stream)
        self.verify_senders = 2 * t + 1 if verify_senders is None else verify_senders
        assert 2 * t + 1 <= self.verify_senders <= n and n > 2 * t
        U = self.U
        self.nout = (n - 2 * t) * K                      # output shares per party
        self.arena = DeviceArena(eng, (n * K * (t + 1) + 3 * n * n * K + K * (t + 1) + n * self.nout) * U + K + (1 << 14))
        ar = self.arena
        self.coeffs = ar.take(n * K * (t + 1) * U)       # [dealer][K][t + 1]
        self.S = ar.take(n * n * K * U)                  # [dealer][recipient][K]
        self.x = ar.take(n * n * K * U)                  # [recipient][K][dealer]
        self.y = ar.take(n * n * K * U)                  # [row i][party][K]
        self.poly = ar.take(K * (t + 1) * U)
        self.status = ar.take(K)
        self.summ = ar.take(64)
        self.bad = ar.take(64)
        self.out = ar.take(n * self.nout * U)            # [party][K][n - 2t]: the reference's output order (share_gen.rs:199-203)

    def upload(self, coeffs):
        self.eng.h2d(self.coeffs, np.ascontiguousarray(coeffs), self.stream)

    def deal(self):
        self._deal(self.coeffs, self.t, self.S)

    def run(self, check=True, out_split=None):
        self.deal()
        self.finish(check, out_split)

    def finish(self, check=True, out_split=None):
        """everything after the dealers' messages have arrived (tests corrupt S in between).  out_split: instead of self.out,
        write batch elements [k0, k0 + count) of every party's list to dst + party * stride: (dst, stride, k0, count), ..."""
        e, n, t, K, s, U = self.eng, self.n, self.t, self.K, self.stream, self.U
        self._mix(self.S, self.x, self.y)
        # verifier i < 2t: recover_secret of the K columns from the first verify_senders parties' shares, then the exact-degree
        # test (share_gen.rs:516-530); the verdicts stay on the device (bad[0] = chunks that failed, bad[1] = the first)
        self._clear_bad()
        ids = list(range(self.verify_senders))
        for i in range(2 * t):
            _check(e.dev_batch_recover_strided(ids, self.y + i * n * K * U, K, K, n, t, t, self.poly, status_d=self.status,
                                               summary_d=self.summ, stream=s), e, "verifier reconstruction")
            _check(e.dev_check_degree(self.poly, self.status, K, t + 1, t, self.bad, s), e, "degree test")
        # output: rows 2t .. n - 1 of every batch element, per party in the order [k][i - 2t]
        for dst, stride, k0, cnt in (out_split or [(self.out, self.nout, 0, K)]):
            _check(e.dev_transpose(self.y + (2 * t * n * K + k0) * U, n - 2 * t, cnt, n * K, dst, n - 2 * t, batch=n, src_batch_stride=K,
                                   dst_batch_stride=stride, stream=s), e, "output shares")
        if check:
            bad, first = self._bad()
            if bad:
                raise RuntimeError(f"RanSha: {bad} verifier reconstructions failed or had the wrong degree (first: column {first})")

    def download(self):
        out = self.eng._new((self.n, self.nout))
        self.eng.d2h(out, self.out, self.stream)
        self.eng.sync(self.stream)
        return out

    def close(self):
        self._drop_graph()
        self.arena.free()


class RanDouSha(_Producer):
    """K batch elements per dealer -> (t + 1) K double sharings ([r]_t, [r]_2t) per party, verified by parties t + 1 .. n - 1
    (each reconstructs both polynomials through ALL n shares: ran_dou_sha/mod.rs:557-559 waits for 2t + 1 degree-t and n
    degree-2t shares, and in one process all n of both have arrived)."""

    def __init__(self, eng, n, t, K, stream=0):
        super().__init__(eng, n, t, K, stream)
        U = self.U
        self.nout = (t + 1) * K
        self.arena = DeviceArena(eng, (n * K * (3 * t + 2) + 5 * n * n * K + 2 * K * n + 2 * n * self.nout) * U + 8 * K + (1 << 14))
        ar = self.arena
        self.coeffs_t = ar.take(n * K * (t + 1) * U)     # [dealer][K][t + 1]
        self.coeffs_2t = ar.take(n * K * (2 * t + 1) * U)  # [dealer][K][2t + 1]   (same secrets in column 0)
        self.S_t = ar.take(n * n * K * U)
        self.S_2t = ar.take(n * n * K * U)
        self.x = ar.take(n * n * K * U)
        self.y_t = ar.take(n * n * K * U)
        self.y_2t = ar.take(n * n * K * U)
        self.poly_t = ar.take(K * n * U)
        self.poly_2t = ar.take(K * n * U)
        self.deg = ar.take(4 * K)
        self.bad = ar.take(64)
        self.out_t = ar.take(n * self.nout * U)          # [party][K][t + 1]  (ran_dou_sha/mod.rs:314-331)
        self.out_2t = ar.take(n * self.nout * U)

    def upload(self, coeffs_t, coeffs_2t):
        self.eng.h2d(self.coeffs_t, np.ascontiguousarray(coeffs_t), self.stream)
        self.eng.h2d(self.coeffs_2t, np.ascontiguousarray(coeffs_2t), self.stream)

    def deal(self):
        self._deal(self.coeffs_t, self.t, self.S_t)           # DouShaNode::init_batch: both sharings of every secret
        self._deal(self.coeffs_2t, 2 * self.t, self.S_2t)

    def run(self, check=True, out_split_t=None, out_split_2t=None):
        self.deal()
        self.finish(check, out_split_t, out_split_2t)

    def finish(self, check=True, out_split_t=None, out_split_2t=None):
        """out_split_*: as RanSha.finish -- where the two output lists go instead of self.out_t / self.out_2t"""
        e, n, t, K, s, U = self.eng, self.n, self.t, self.K, self.stream, self.U
        self._mix(self.S_t, self.x, self.y_t)                 # RanDouShaNode::init_batch step 1
        self._mix(self.S_2t, self.x, self.y_2t)               # step 2
        self._clear_bad()
        ids = list(range(n))
        for i in range(t + 1, n):                                              # step 3: verifier i
            _check(e.dev_batch_interpolate(ids, self.y_t + i * n * K * U, K, K, n, self.poly_t, self.deg, s), e, "interpolate [r]_t")
            _check(e.dev_batch_interpolate(ids, self.y_2t + i * n * K * U, K, K, n, self.poly_2t, self.deg, s), e, "interpolate [r]_2t")
            _check(e.dev_check_double_share(self.poly_t, self.poly_2t, K, n, t, self.bad, s), e, "degree / equal-secret tests")
        for y, split in ((self.y_t, out_split_t or [(self.out_t, self.nout, 0, K)]),
                         (self.y_2t, out_split_2t or [(self.out_2t, self.nout, 0, K)])):      # steps 4-5: rows 0 .. t
            for dst, stride, k0, cnt in split:
                _check(e.dev_transpose(y + k0 * U, t + 1, cnt, n * K, dst, t + 1, batch=n, src_batch_stride=K, dst_batch_stride=stride,
                                       stream=s), e, "output double shares")
        if check:
            bad, first = self._bad()
            if bad:
                raise RuntimeError(f"RanDouSha: {bad} verifier checks failed (first: column {first})")

    def download(self):
        a, b = self.eng._new((self.n, self.nout)), self.eng._new((self.n, self.nout))
        self.eng.d2h(a, self.out_t, self.stream)
        self.eng.d2h(b, self.out_2t, self.stream)
        self.eng.sync(self.stream)
        return a, b

    def close(self):
        self._drop_graph()
        self.arena.free()


class Preprocessing:
    """run_preprocessing's triple part (honeybadger/mod.rs:1239-1393) for all n parties, device-resident from the dealers'
    polynomials to [c]_t: RanSha produces 2 N random sharings per party (a = the first N, b = the next N:
    take_random_shares twice, :1307-1316), RanDouSha the N double sharings, TripleGen consumes them where they lie."""

    def __init__(self, eng, n, t, N, stream=0):
        assert N % (2 * t + 1) == 0
        self.eng, self.n, self.t, self.N, self.stream = eng, n, t, N, stream
        self.K_rs = -(-2 * N // (n - 2 * t))             # RanSha batch elements per dealer: (n - 2t) K >= 2 N
        self.K_rd = -(-N // (t + 1))                     # RanDouSha: (t + 1) K >= N
        self.rs = RanSha(eng, n, t, self.K_rs, stream)
        self.rd = RanDouSha(eng, n, t, self.K_rd, stream)
        self.tg = TripleGen(eng, n, t, N, stream)

    def run(self, check=True):
        e, n, t, N, s = self.eng, self.n, self.t, self.N, self.stream
        U = self.rs.U
        if N % (n - 2 * t) == 0 and N % (t + 1) == 0:
            # whole batch elements on both sides of every cut: the producers' output slices go straight into TripleGen's
            # [party][N] arrays (a = the first N of a party's list, b = the next N) and nothing is copied
            k1, k2 = N // (n - 2 * t), N // (t + 1)
            self.rs.run(check, out_split=[(self.tg.a, N, 0, k1), (self.tg.b, N, k1, k1)])
            self.rd.run(check, out_split_t=[(self.tg.rt, N, 0, k2)], out_split_2t=[(self.tg.r2t, N, 0, k2)])
            self.tg.run(check)
            return
        self.rs.run(check)
        self.rd.run(check)
        for p in range(n):   # the parties' lists, in the reference's order, become TripleGen's [party][N] inputs
            e.d2d(self.tg.a + p * N * U, self.rs.out + p * self.rs.nout * U, N * U, s)
            e.d2d(self.tg.b + p * N * U, self.rs.out + (p * self.rs.nout + N) * U, N * U, s)
            e.d2d(self.tg.rt + p * N * U, self.rd.out_t + p * self.rd.nout * U, N * U, s)
            e.d2d(self.tg.r2t + p * N * U, self.rd.out_2t + p * self.rd.nout * U, N * U, s)
        self.tg.run(check)

    def close(self):
        self.rs.close()
        self.rd.close()
        self.tg.close()
