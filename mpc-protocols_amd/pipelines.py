"""Device-resident replays of the reference's arithmetic pipelines for ALL n simulated parties on one
GPU (how every reference test/bench runs: n parties in one process on FakeNetwork).  Host-side
orchestration only -- every arithmetic step is an hbmpc_dev_* call; the parties' all-to-all is a
layout choice (strided sender rows), nothing is copied between "parties".

  triple_gen   TripleGenNode::init_batch + BatchReconNode (degree 2t) + try_finalize_triple_gen
               triple_gen/triple_generation.rs:304-364,164-232; batch_recon/batch_recon.rs:144-185,332-481
  fpmul        FPMulNode::init = Multiply (Beaver, RBC path) + TruncPrNode
               fpmul/fpmul.rs:61-110, mul/multiplication.rs:417-426,57-139, fpmul/truncpr.rs:185-318
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
