"""Device-resident replays of the reference's arithmetic pipelines for ALL n simulated parties on one GPU (how every
reference test / bench runs: n parties in one process on FakeNetwork) -- thin wrappers over the hbmpc_pipe_* handles of
the C ABI (include/hbmpc_hip.h, csrc/capi_pipelines.hip): the call sequencing, the arena layout and the capture rules live
in the library; this file only names things for the tests and bench.py.

  TripleGen      TripleGenNode::init_batch + BatchReconNode (degree 2t) + try_finalize_triple_gen
                 triple_gen/triple_generation.rs:304-364,164-232; batch_recon/batch_recon.rs:144-185,332-481
  FpMul          FPMulNode::init = Multiply (Beaver, RBC path) + TruncPrNode
                 fpmul/fpmul.rs:61-110, mul/multiplication.rs:417-426,57-139, fpmul/truncpr.rs:185-318
  RanSha         RanShaNode::init_batch + init_ransha_batch + reconstruction_handler + try_finalize
                 share_gen/share_gen.rs:232-289,401-454,516-530,199-203
  RanDouSha      DouShaNode::init_batch + RanDouShaNode::init_batch + reconstruction_handler + try_finalize
                 double_share/double_share_generation.rs:151-215, ran_dou_sha/mod.rs:371-449,569-602,314-331
  Preprocessing  run_preprocessing's triple part (honeybadger/mod.rs:1239-1393): RanSha -> a, b; RanDouSha -> r; TripleGen
"""
from __future__ import annotations

import ctypes as C

import numpy as np


class _Pipe:
    """One hbmpc_pipe handle.  Device buffers are reached by name: `pipe.a`, `pipe.out`, ... are raw device pointers
    (hbmpc_pipe_buffer).  run(check=False) only enqueues on the stream; run(check=True) reads the summary back after every
    decode and raises where the reference's `?` would return the error."""
    _own = True

    def __init__(self, eng, handle, stream):
        self.eng, self.h, self.stream = eng, handle, stream
        self.U = eng.ebytes  # bytes per element of the engine's field
        self._ptrs = {}

    def _rc(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} -> ShareErrorCode {rc}: {self.eng.last_error()}")

    @classmethod
    def _create(cls, eng, fn, *args, stream=0):
        h = C.c_void_p()
        rc = getattr(eng.L, fn)(eng.ctx, *[C.c_size_t(a) for a in args], C.c_void_p(stream), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"{fn}{args} -> ShareErrorCode {rc}: {eng.last_error()}")
        return h

    def buffer(self, name):
        """(device pointer, elements) of a named buffer of this pipeline"""
        if name not in self._ptrs:
            p, n = C.c_void_p(), C.c_size_t()
            self._rc(self.eng.L.hbmpc_pipe_buffer(self.h, name.encode(), C.byref(p), C.byref(n)), f"buffer {name!r}")
            self._ptrs[name] = (p.value or 0, n.value)
        return self._ptrs[name]

    def __getattr__(self, name):  # pipe.a, pipe.coeffs, ...: the device pointer of that buffer
        if name.startswith("_") or name in ("eng", "h", "stream", "U"):
            raise AttributeError(name)
        try:
            return self.buffer(name)[0]
        except RuntimeError:
            raise AttributeError(name) from None

    def upload_named(self, name, arr):
        arr = np.ascontiguousarray(arr)
        self._rc(self.eng.L.hbmpc_pipe_upload(self.h, name.encode(), arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes // self.eng.ebytes)),
                 f"upload {name!r}")

    def download_named(self, name, shape):
        out = self.eng._new(shape)
        self._rc(self.eng.L.hbmpc_pipe_download(self.h, name.encode(), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.nbytes // self.eng.ebytes)),
                 f"download {name!r}")
        return out

    def _call(self, fn, check):
        self.eng.L.hbmpc_pipe_set_checked(self.h, C.c_int(1 if check else 0))
        self._rc(getattr(self.eng.L, fn)(self.h), fn)

    def run(self, check=True):
        self._call("hbmpc_pipe_run", check)

    def capture(self):
        assert self.stream, "capture needs an explicit stream"
        self._rc(self.eng.L.hbmpc_pipe_capture(self.h), "capture")

    def replay(self):
        self._rc(self.eng.L.hbmpc_pipe_replay(self.h), "replay")

    def sync(self):
        self._rc(self.eng.L.hbmpc_pipe_sync(self.h), "sync")

    def summary(self):
        s = np.zeros(4, dtype=np.uint32)
        self._rc(self.eng.L.hbmpc_pipe_summary(self.h, s.ctypes.data_as(C.c_void_p)), "summary")
        return s

    def _bad(self):
        """{verifier checks that failed, first failing batch element} of a producer (synchronises)"""
        b = np.zeros(2, dtype=np.uint32)
        self._rc(self.eng.L.hbmpc_pipe_verdict(self.h, b.ctypes.data_as(C.c_void_p)), "verdict")
        return int(b[0]), int(b[1])

    def close(self):
        if self.h and self._own:
            self.eng.L.hbmpc_pipe_destroy(self.h)
        self.h = None


class TripleGen(_Pipe):
    """n parties, threshold t, N triples (N a multiple of 2t+1).  Buffers a, b, r2t, rt, c are [party][N] canonical.
    Works in either field: the reference runs TripleGenNode over Fr and, in PreprocNodesSmallField
    (honeybadger/mod.rs:316-324), over GoldilocksField -- the element size follows the engine's field.  run() is one library call
    (hbmpc_[gl_]dev_triplegen_parties): one launch up to 1 024 chunks of 2t + 1 triples when n = 3t + 1 <= 16, four launches beyond."""

    def __init__(self, eng, n, t, N, stream=0, _handle=None):
        self.n, self.t, self.N, self.m, self.G = n, t, N, 2 * t + 1, N // (2 * t + 1)
        super().__init__(eng, _handle or self._create(eng, "hbmpc_pipe_triplegen_create", n, t, N, stream=stream), stream)

    def upload(self, a, b, r2t, rt):
        for name, src in (("a", a), ("b", b), ("r2t", r2t), ("rt", rt)):
            self.upload_named(name, src)

    def download_c(self):
        return self.download_named("c", (self.n, self.N))


class FpMul(_Pipe):
    """Fixed-point multiplication of N element pairs for n parties: Beaver mul (a-x, b-y opened by direct robust
    interpolation, i.e. the RBC path of Multiply::init for < t+1 leftovers that FPMulNode always takes) followed by TruncPr
    with k-bit values and m fractional bits.  open_senders: how many parties' shares an open interpolates from.  Default
    2t+1: the reference opens as soon as that many have arrived (multiplication.rs:388,617, truncpr.rs:202) -- with d = t
    that is exactly d + t + 1, a decode with no OEC round, which the library runs as one launch.  n = every party's share.
    run() is one library call (hbmpc_dev_fpmul_parties): ONE launch up to 2 048 elements, five launches up to 8 192, four beyond
    (the first open then forms its senders' shares itself); summary_first / summary hold the two opens' summaries."""

    def __init__(self, eng, n, t, N, k, m, stream=0, open_senders=None):
        self.n, self.t, self.N, self.k, self.m = n, t, N, k, m
        self.open_senders = 2 * t + 1 if open_senders is None else open_senders
        super().__init__(eng, self._create(eng, "hbmpc_pipe_fpmul_create", n, t, N, k, m, self.open_senders, stream=stream), stream)

    def upload(self, x, y, ta, tb, tc, rbits, rint):
        for name, src in (("x", x), ("y", y), ("ta", ta), ("tb", tb), ("tc", tc), ("rbits", rbits), ("rint", rint)):
            self.upload_named(name, src)

    def download(self, which="out"):
        return self.download_named(which, (self.n, self.N))


class _Producer(_Pipe):
    """What RanSha and RanDouSha share: every dealer p deals K secrets to the n recipients (the dealers' polynomials are the
    INPUT: coefficient rows [dealer][K][deg + 1], column 0 the secret -- filled by the host, or on the device by
    hbmpc_dev_fill_coeffs: the reference draws them from each party's rng), recipient j multiplies the vector of the n
    shares it received for batch element k by the n x n Vandermonde matrix make_vandermonde(n, n - 1), row i of the result
    goes to verifier i, and the other rows are the party's output.  deal() / finish() are the two halves of run()."""

    def deal(self):
        self._call("hbmpc_pipe_deal", False)

    def finish(self, check=True):
        self._call("hbmpc_pipe_finish", False)
        if check:
            bad, first = self._bad()
            if bad:
                raise RuntimeError(f"{type(self).__name__}: {bad} verifier checks failed (first: column {first})")

    def run(self, check=True):
        self.deal()
        self.finish(check)


class RanSha(_Producer):
    """K batch elements per dealer -> (n - 2t) K random degree-t sharings per party, verified by parties 0 .. 2t - 1.
    verify_senders: how many parties' shares a verifier reconstructs from.  Default 2t + 1: the reference's handler fires
    as soon as that many have arrived (share_gen.rs:497), i.e. the first 2t + 1 senders -- a decode with no OEC round."""

    def __init__(self, eng, n, t, K, stream=0, verify_senders=None, _handle=None):
        self.n, self.t, self.K, self.nout = n, t, K, (n - 2 * t) * K
        self.verify_senders = 2 * t + 1 if verify_senders is None else verify_senders
        super().__init__(eng, _handle or self._create(eng, "hbmpc_pipe_ransha_create", n, t, K, self.verify_senders, stream=stream), stream)

    def upload(self, coeffs):
        self.upload_named("coeffs", coeffs)

    def download(self):
        return self.download_named("out", (self.n, self.nout))


class RanDouSha(_Producer):
    """K batch elements per dealer -> (t + 1) K double sharings ([r]_t, [r]_2t) per party, verified by parties t + 1 .. n - 1
    (each reconstructs both polynomials through ALL n shares: ran_dou_sha/mod.rs:557-559 waits for 2t + 1 degree-t and n
    degree-2t shares, and in one process all n of both have arrived)."""

    def __init__(self, eng, n, t, K, stream=0, _handle=None):
        self.n, self.t, self.K, self.nout = n, t, K, (t + 1) * K
        super().__init__(eng, _handle or self._create(eng, "hbmpc_pipe_randousha_create", n, t, K, stream=stream), stream)

    def upload(self, coeffs_t, coeffs_2t):
        self.upload_named("coeffs_t", coeffs_t)
        self.upload_named("coeffs_2t", coeffs_2t)

    def download(self):
        return self.download_named("out_t", (self.n, self.nout)), self.download_named("out_2t", (self.n, self.nout))


class Preprocessing(_Pipe):
    """run_preprocessing's triple part (honeybadger/mod.rs:1239-1393) for all n parties, device-resident from the dealers'
    polynomials to [c]_t: RanSha produces 2 N random sharings per party (a = the first N, b = the next N:
    take_random_shares twice, :1307-1316), RanDouSha the N double sharings, TripleGen consumes them where they lie."""

    def __init__(self, eng, n, t, N, stream=0):
        assert N % (2 * t + 1) == 0
        self.n, self.t, self.N = n, t, N
        self.K_rs = -(-2 * N // (n - 2 * t))             # RanSha batch elements per dealer: (n - 2t) K >= 2 N
        self.K_rd = -(-N // (t + 1))                     # RanDouSha: (t + 1) K >= N
        super().__init__(eng, self._create(eng, "hbmpc_pipe_preprocessing_create", n, t, N, stream=stream), stream)
        self.rs = RanSha(eng, n, t, self.K_rs, stream, _handle=self._part("ransha"))
        self.rd = RanDouSha(eng, n, t, self.K_rd, stream, _handle=self._part("randousha"))
        self.tg = TripleGen(eng, n, t, N, stream, _handle=self._part("triplegen"))
        for part in (self.rs, self.rd, self.tg):
            part._own = False  # borrowed: destroyed with this handle

    def _part(self, name):
        h = C.c_void_p()
        self._rc(self.eng.L.hbmpc_pipe_part(self.h, name.encode(), C.byref(h)), f"part {name!r}")
        return h

    def run(self, check=True):
        super().run(check)
        if check:
            bad, first = self._bad()
            if bad:
                raise RuntimeError(f"Preprocessing: {bad} verifier checks failed (first: column {first})")
