"""ctypes binding of libhbmpc_hip.so (the C ABI in include/hbmpc_hip.h) for tests and bench.py.

This is plumbing, not a product path: the product is the shared library.  Arrays are numpy uint64
with a trailing axis of 4 limbs (least-significant first) == U256[].  Every call goes through the
C ABI; there is no fallback -- if the library or a GPU is missing, loading/creating fails loudly.

Mirrors the reference interface for the path (SecretSharingScheme / free functions):
  compute_shares        RobustShare::compute_shares        robust_interpolate.rs:52-82
  make_vandermonde / vandermonde_apply                     common/share/mod.rs:31-76
  batch_recover(_p0)    batch_recover_secret               robust_interpolate.rs:284-443
  recover_secret        RobustShare::recover_secret        robust_interpolate.rs:94-157
  gao_rs_decode                                            robust_interpolate.rs:456-538
  nonrobust_recover_secret  NonRobustShare::recover_secret common/share/shamir.rs:199-239
  triple_local/.../truncpr_finalize                        triple_gen, mul, fpmul element-wise math
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhbmpc_hip.so")
_LIB = None


class HbmpcError(RuntimeError):
    pass


def build(jobs: int = 8) -> str:
    """Compile every HIP translation unit for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), f"-j{jobs}"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise HbmpcError(f"{LIB_PATH} is missing: run `make -C {_HERE}/csrc -j8` (or __graft_entry__.build())")
        _LIB = C.CDLL(LIB_PATH)
        _LIB.hbmpc_last_error.restype = C.c_char_p
        _LIB.hbmpc_version.restype = C.c_char_p
    return _LIB


def _p(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(int(a))  # raw device pointer


def u256(shape):
    return np.zeros(tuple(shape) + (4,), dtype=np.uint64)


def _sz(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint64))


class Engine:
    """One hbmpc_ctx on one GPU.  Host-pointer calls take/return numpy U256 arrays; the
    dev_* calls take raw device pointers (ints, e.g. torch.Tensor.data_ptr()) and a stream."""

    def __init__(self, device: int = 0, impl: str | None = None, field: str = "fr"):
        """field: "fr" (bls12-381 Fr, U256 elements = numpy [..., 4] uint64) or "goldilocks" (8-byte elements =
        numpy uint64; the hbmpc_gl_* entry points)"""
        self.L = lib()
        self.ctx = C.c_void_p()
        self.field = field
        self.ebytes = {"fr": 32, "goldilocks": 8}[field]
        self._pfx = "hbmpc_" if field == "fr" else "hbmpc_gl_"
        rc = self.L.hbmpc_create(C.c_int(device), C.c_int(0 if field == "fr" else 1), C.byref(self.ctx))
        if rc != 0:
            msg = self.L.hbmpc_last_error(None)
            raise HbmpcError(f"hbmpc_create(device={device}) failed with code {rc}: {msg.decode() if msg else ''}")
        if impl is not None:
            self.set_impl(impl)

    def _f(self, name):  # the entry point of this context's field
        return getattr(self.L, self._pfx + name)

    def _new(self, shape):  # zeroed output array of field elements
        return u256(shape) if self.field == "fr" else np.zeros(tuple(shape), dtype=np.uint64)

    def close(self):
        if self.ctx:
            self.L.hbmpc_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self) -> str:
        m = self.L.hbmpc_last_error(self.ctx)
        return m.decode() if m else ""

    def set_impl(self, impl: str):
        rc = self.L.hbmpc_set_field_impl(self.ctx, C.c_int({"u29": 0, "sat32": 1}[impl]))
        assert rc == 0

    def set_small_call_staging(self, zero_copy: bool):
        assert self.L.hbmpc_set_small_call_staging(self.ctx, C.c_int(1 if zero_copy else 0)) == 0

    def set_small_batch_chunks(self, max_chunks: int):
        assert self.L.hbmpc_set_small_batch_chunks(self.ctx, C.c_size_t(max_chunks)) == 0

    @staticmethod
    def gather_party_major(engines, root, shards_d, counts, strides, n_rows, out_d, out_stride, sync_sources=True, stream=0):
        """hbmpc_dev_gather_party_major: one engine (context) per shard; device pointers as ints"""
        k = len(engines)
        ctxs = (C.c_void_p * k)(*[e.ctx for e in engines])
        ptrs = (C.c_void_p * k)(*shards_d)
        cnt = (C.c_size_t * k)(*counts)
        strd = (C.c_size_t * k)(*strides)
        return engines[root].L.hbmpc_dev_gather_party_major(ctxs, C.c_size_t(k), C.c_size_t(root), ptrs, cnt, strd, C.c_size_t(n_rows),
                                                            C.c_void_p(out_d), C.c_size_t(out_stride), C.c_int(1 if sync_sources else 0),
                                                            C.c_void_p(stream))

    def peer_access(self, source: "Engine") -> bool:
        """True when this engine's device reads `source`'s memory directly (same device or peer access over xGMI)"""
        out = C.c_int(0)
        rc = self.L.hbmpc_dev_peer_access(self.ctx, source.ctx, C.byref(out))
        if rc != 0:
            raise HbmpcError(rc, self.last_error())
        return bool(out.value)

    def set_matrix_cores(self, on: bool, min_chunks: int = 0):
        """large Fr decodes on the matrix cores (int8 MFMA); min_chunks = 0 keeps the current threshold"""
        # on: False / True, or 2 = matrix cores without the workgroup-per-tile kernel of small batches, 3 = large encodes with
        # one table row per point instead of per point pair (both A/B aids)
        assert self.L.hbmpc_set_matrix_cores(self.ctx, C.c_int(int(on)), C.c_size_t(min_chunks)) == 0

    def scrub_staging(self):
        assert self.L.hbmpc_scrub_staging(self.ctx) == 0

    def set_lazy_fallback_tables(self, on):
        """a new sender set's OEC / Gao and second-chance tables are built only when the first kernel flagged a chunk:
        0 = never, 1 = host-pointer calls (default), 2 = device-pointer calls as well"""
        assert self.L.hbmpc_set_lazy_fallback_tables(self.ctx, C.c_int(int(on))) == 0

    def set_single_launch_decode(self, on: bool):
        assert self.L.hbmpc_set_single_launch_decode(self.ctx, C.c_int(1 if on else 0)) == 0

    def set_matrix_core_workgroups(self, workgroups: int):
        assert self.L.hbmpc_set_matrix_core_workgroups(self.ctx, C.c_int(workgroups)) == 0

    def set_second_chance(self, on: bool):
        assert self.L.hbmpc_set_second_chance(self.ctx, C.c_int(1 if on else 0)) == 0

    def cache_stats(self):
        out = (C.c_size_t * 4)()
        assert self.L.hbmpc_cache_stats(self.ctx, out) == 0
        return dict(zip(("tables", "pinned", "retired", "evictions"), [int(v) for v in out]))

    def set_force_generic(self, on: bool):
        assert self.L.hbmpc_set_force_generic(self.ctx, C.c_int(1 if on else 0)) == 0

    # ---- host-pointer API ----
    def compute_shares(self, coeffs, n, d, out=None):
        coeffs = np.ascontiguousarray(coeffs)
        B = coeffs.shape[0]
        if out is None:  # a fresh array: its first-touch page faults happen inside the call's device-to-host copy
            out = self._new((n, B))
        rc = self._f("compute_shares")(self.ctx, _p(coeffs), C.c_size_t(B), C.c_size_t(n), C.c_size_t(d), _p(out))
        return rc, out

    def compute_shares_seeded(self, seed, secrets, n, d, first_index=0):
        """seed: 32 bytes ("hbmpc-chacha20-v1", include/hbmpc_hip.h); secrets: [B] elements, or an int B: the
        secrets are drawn too"""
        if isinstance(secrets, int):
            B, sp = secrets, C.c_void_p(0)
        else:
            secrets = np.ascontiguousarray(secrets)
            B, sp = secrets.shape[0], _p(secrets)
        out = self._new((n, B))
        seed = (C.c_uint8 * 32).from_buffer_copy(bytes(seed))
        rc = self._f("compute_shares_seeded")(self.ctx, seed, sp, C.c_size_t(B), C.c_uint64(first_index),
                                              C.c_size_t(n), C.c_size_t(d), _p(out))
        return rc, out

    def make_vandermonde(self, n, d):
        out = self._new((n, d + 1))
        rc = self._f("make_vandermonde")(self.ctx, C.c_size_t(n), C.c_size_t(d), _p(out))
        return rc, out

    def vandermonde_apply(self, x, n, d):
        x = np.ascontiguousarray(x)
        G = x.shape[0]
        out = self._new((n, G))
        rc = self._f("vandermonde_apply")(self.ctx, _p(x), C.c_size_t(G), C.c_size_t(n), C.c_size_t(d), _p(out))
        return rc, out

    def batch_recover(self, sender_ids, evals, n, d, t):
        evals = np.ascontiguousarray(evals)
        S = len(sender_ids)
        G = evals.shape[1] if evals.ndim == (3 if self.field == "fr" else 2) else 0
        out = self._new((G, d + 1))
        nco = np.zeros(G, dtype=np.uint32)
        st = np.zeros(G, dtype=np.uint8)
        rc = self._f("batch_recover")(self.ctx, _p(_sz(sender_ids)), C.c_size_t(S), _p(evals), C.c_size_t(G),
                                        C.c_size_t(n), C.c_size_t(d), C.c_size_t(t), _p(out), _p(nco), _p(st))
        return rc, out, nco, st

    def batch_recover_p0(self, sender_ids, evals, n, d, t):
        evals = np.ascontiguousarray(evals)
        S = len(sender_ids)
        G = evals.shape[1] if evals.ndim == (3 if self.field == "fr" else 2) else 0
        out = self._new((G,))
        st = np.zeros(G, dtype=np.uint8)
        rc = self._f("batch_recover_p0")(self.ctx, _p(_sz(sender_ids)), C.c_size_t(S), _p(evals), C.c_size_t(G),
                                           C.c_size_t(n), C.c_size_t(d), C.c_size_t(t), _p(out), _p(st))
        return rc, out, st

    def recover_secret(self, ids, degrees, vals, n, t):
        vals = np.ascontiguousarray(vals)
        S = len(ids)
        cap = (int(degrees[0]) + 1) if S else 1
        out = self._new((max(cap, 1),))
        nco = C.c_size_t(0)
        sec = self._new((1,))
        rc = self._f("recover_secret")(self.ctx, _p(_sz(ids)), _p(_sz(degrees)), _p(vals), C.c_size_t(S),
                                         C.c_size_t(n), C.c_size_t(t), _p(out), C.byref(nco), _p(sec))
        return rc, out[: nco.value], sec[0]

    def gao_rs_decode(self, received, k, n, erasures):
        received = np.ascontiguousarray(received)
        out = self._new((max(k, 1),))
        nco = C.c_size_t(0)
        rc = self._f("gao_rs_decode")(self.ctx, _p(received), C.c_size_t(k), C.c_size_t(n), _p(_sz(erasures)),
                                        C.c_size_t(len(erasures)), _p(out), C.byref(nco))
        return rc, out[: nco.value]

    def nonrobust_recover_secret(self, ids, degrees, vals, n):
        vals = np.ascontiguousarray(vals)
        S = len(ids)
        out = self._new((max(S, 1),))
        nco = C.c_size_t(0)
        sec = self._new((1,))
        rc = self._f("nonrobust_recover_secret")(self.ctx, _p(_sz(ids)), _p(_sz(degrees)), _p(vals), C.c_size_t(S),
                                                   C.c_size_t(n), _p(out), C.byref(nco), _p(sec))
        return rc, out[: nco.value], sec[0]

    def batch_interpolate(self, ids, evals, n):
        evals = np.ascontiguousarray(evals)
        S, G = len(ids), evals.shape[1]
        co = self._new((G, S))
        deg = np.zeros(G, dtype=np.uint32)
        rc = self._f("batch_interpolate")(self.ctx, _p(_sz(ids)), C.c_size_t(S), _p(evals), C.c_size_t(G),
                                            C.c_size_t(n), _p(co), _p(deg))
        return rc, co, deg

    def _ew(self, name, ins, n_out=1, extra=()):
        ins = [np.ascontiguousarray(a) for a in ins]
        N = ins[0].shape[0]
        outs = [self._new((N,)) for _ in range(n_out)]
        args = [self.ctx] + [_p(a) for a in ins] + [C.c_size_t(e) for e in extra] + [C.c_size_t(N)] + [_p(o) for o in outs]
        rc = getattr(self.L, name.replace("hbmpc_", self._pfx, 1))(*args)
        return (rc, *outs)

    def fr_op(self, op, a, b):
        a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
        out = self._new((a.shape[0],))
        rc = self._f("fr_op")(self.ctx, C.c_int({"add": 0, "sub": 1, "mul": 2}[op]), _p(a), _p(b),
                                C.c_size_t(a.shape[0]), _p(out))
        return rc, out

    def triple_local(self, a, b, r2t):
        return self._ew("hbmpc_triple_local", [a, b, r2t])

    def triple_finalize(self, rt, opened):
        return self._ew("hbmpc_triple_finalize", [rt, opened])

    def beaver_open_shares(self, a, b, x, y):
        return self._ew("hbmpc_beaver_open_shares", [a, b, x, y], n_out=2)

    def beaver_finalize(self, c, x, y, d, e):
        return self._ew("hbmpc_beaver_finalize", [c, x, y, d, e])

    def truncpr_rdash(self, r_bits, m):
        r_bits = np.ascontiguousarray(r_bits)
        N = r_bits.shape[1]
        out = self._new((N,))
        rc = self.L.hbmpc_truncpr_rdash(self.ctx, _p(r_bits), C.c_size_t(m), C.c_size_t(N), _p(out))
        return rc, out

    def truncpr_open_share(self, a, r_dash, r_int, k, m):
        return self._ew("hbmpc_truncpr_open_share", [a, r_dash, r_int], extra=(k, m))

    def truncpr_finalize(self, a, r_dash, c_open, m):
        return self._ew("hbmpc_truncpr_finalize", [a, r_dash, c_open], extra=(m,))

    # ---- device-pointer API ----
    def dev_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        rc = self.L.hbmpc_dev_alloc(self.ctx, C.c_size_t(nbytes), C.byref(p))
        if rc != 0:
            raise HbmpcError(f"hbmpc_dev_alloc({nbytes}) -> {rc}: {self.last_error()}")
        return p.value

    def dev_free(self, ptr: int):
        self.L.hbmpc_dev_free(self.ctx, C.c_void_p(ptr))

    def stream_pool_release_threshold(self) -> int:
        """Release threshold of the device's current stream-ordered pool (0 = the platform default, under which hipMallocAsync
        buffers are NOT safe to hand to the library: include/hbmpc_hip.h, 'Device buffers')."""
        v = C.c_uint64(0)
        rc = self.L.hbmpc_stream_pool_release_threshold(self.ctx, C.byref(v))
        if rc != 0:
            raise HbmpcError(f"hbmpc_stream_pool_release_threshold -> {rc}: {self.last_error()}")
        return v.value

    def stream_pool_retain(self):
        rc = self.L.hbmpc_stream_pool_retain(self.ctx)
        if rc != 0:
            raise HbmpcError(f"hbmpc_stream_pool_retain -> {rc}: {self.last_error()}")

    def h2d(self, dptr: int, arr: np.ndarray, stream=0):
        arr = np.ascontiguousarray(arr)
        rc = self.L.hbmpc_memcpy_h2d(self.ctx, C.c_void_p(dptr), _p(arr), C.c_size_t(arr.nbytes), C.c_void_p(stream))
        assert rc == 0, self.last_error()

    def d2h(self, arr: np.ndarray, dptr: int, stream=0):
        assert arr.flags["C_CONTIGUOUS"]
        rc = self.L.hbmpc_memcpy_d2h(self.ctx, _p(arr), C.c_void_p(dptr), C.c_size_t(arr.nbytes), C.c_void_p(stream))
        assert rc == 0, self.last_error()

    def d2d(self, dst: int, src: int, nbytes: int, stream=0):
        rc = self.L.hbmpc_memcpy_d2d(self.ctx, C.c_void_p(dst), C.c_void_p(src), C.c_size_t(nbytes), C.c_void_p(stream))
        assert rc == 0, self.last_error()

    def fr_op_scalar(self, op, a, scalar):
        """op: "add" (a + s), "sub" (a - s), "mul" (a * s), "rsub" (s - a); scalar: one element"""
        a = np.ascontiguousarray(a)
        sc = np.ascontiguousarray(scalar)
        out = self._new((a.shape[0],))
        rc = self._f("fr_op_scalar")(self.ctx, C.c_int({"add": 0, "sub": 1, "mul": 2, "rsub": 3}[op]), _p(a), _p(sc),
                                       C.c_size_t(a.shape[0]), _p(out))
        return rc, out

    def dev_fr_op_scalar(self, op, a_d, scalar, N, out_d, stream=0):
        sc = np.ascontiguousarray(scalar)
        return self._f("dev_fr_op_scalar")(self.ctx, C.c_int({"add": 0, "sub": 1, "mul": 2, "rsub": 3}[op]), C.c_void_p(a_d),
                                             _p(sc), C.c_size_t(N), C.c_void_p(out_d), C.c_void_p(stream))

    def dev_fr_op(self, op, a_d, b_d, N, out_d, stream=0):
        return self._f("dev_fr_op")(self.ctx, C.c_int({"add": 0, "sub": 1, "mul": 2}[op]), C.c_void_p(a_d),
                                      C.c_void_p(b_d), C.c_size_t(N), C.c_void_p(out_d), C.c_void_p(stream))

    def stream_create(self) -> int:
        st = C.c_void_p()
        rc = self.L.hbmpc_stream_create(self.ctx, C.byref(st))
        if rc != 0:
            raise HbmpcError(f"stream_create -> {rc}: {self.last_error()}")
        return st.value

    def stream_destroy(self, stream: int):
        self.L.hbmpc_stream_destroy(self.ctx, C.c_void_p(stream))

    # ---- HIP graphs: capture a sequence of dev_* calls once, replay it per refill of the buffers ----
    def graph_begin(self, stream):
        rc = self.L.hbmpc_graph_begin_capture(self.ctx, C.c_void_p(stream))
        if rc != 0:
            raise HbmpcError(f"graph_begin_capture -> {rc}: {self.last_error()}")

    def graph_end(self, stream) -> int:
        g = C.c_void_p()
        rc = self.L.hbmpc_graph_end_capture(self.ctx, C.c_void_p(stream), C.byref(g))
        if rc != 0:
            raise HbmpcError(f"graph_end_capture -> {rc}: {self.last_error()}")
        return g.value

    def graph_launch(self, graph: int, stream=0):
        rc = self.L.hbmpc_graph_launch(self.ctx, C.c_void_p(graph), C.c_void_p(stream))
        if rc != 0:
            raise HbmpcError(f"graph_launch -> {rc}: {self.last_error()}")

    def graph_destroy(self, graph: int):
        self.L.hbmpc_graph_destroy(C.c_void_p(graph))

    def sync(self, stream=0):
        rc = self.L.hbmpc_stream_sync(self.ctx, C.c_void_p(stream))
        if rc != 0:
            raise HbmpcError(f"stream sync -> {rc}: {self.last_error()}")

    def dev_compute_shares(self, coeffs_d, B, n, d, out_d, stream=0):
        return self._f("dev_compute_shares")(self.ctx, C.c_void_p(coeffs_d), C.c_size_t(B), C.c_size_t(n),
                                               C.c_size_t(d), C.c_void_p(out_d), C.c_void_p(stream))

    def dev_fill_coeffs(self, seed, secrets_d, B, first_index, d, coeffs_d, stream=0):
        seed = (C.c_uint8 * 32).from_buffer_copy(bytes(seed))
        return self._f("dev_fill_coeffs")(self.ctx, seed, C.c_void_p(secrets_d), C.c_size_t(B), C.c_uint64(first_index),
                                          C.c_size_t(d), C.c_void_p(coeffs_d), C.c_void_p(stream))

    def dev_compute_shares_seeded(self, seed, secrets_d, B, first_index, n, d, coeffs_ws_d, out_d, stream=0):
        seed = (C.c_uint8 * 32).from_buffer_copy(bytes(seed))
        return self._f("dev_compute_shares_seeded")(self.ctx, seed, C.c_void_p(secrets_d), C.c_size_t(B),
                                                    C.c_uint64(first_index), C.c_size_t(n), C.c_size_t(d),
                                                    C.c_void_p(coeffs_ws_d), C.c_void_p(out_d), C.c_void_p(stream))

    def dev_batch_interpolate(self, ids, evals_d, row_stride, G, n, coeffs_d, degree_d, stream=0):
        """plain Lagrange through ALL len(ids) sender rows (row s at evals_d + s * row_stride elements): coeffs[G][S], degree[G]"""
        return self._f("dev_batch_interpolate")(self.ctx, _p(_sz(ids)), C.c_size_t(len(ids)), C.c_void_p(evals_d), C.c_size_t(row_stride),
                                                C.c_size_t(G), C.c_size_t(n), C.c_void_p(coeffs_d), C.c_void_p(degree_d), C.c_void_p(stream))

    def dev_transpose(self, src_d, rows, cols, src_row_stride, dst_d, dst_row_stride, batch=1, src_batch_stride=0, dst_batch_stride=0,
                      stream=0):
        """dst[b][c][r] = src[b][r][c]; strides in elements of this engine's field"""
        return self.L.hbmpc_dev_transpose(self.ctx, C.c_void_p(src_d), C.c_size_t(rows), C.c_size_t(cols), C.c_size_t(src_row_stride),
                                          C.c_void_p(dst_d), C.c_size_t(dst_row_stride), C.c_size_t(batch), C.c_size_t(src_batch_stride),
                                          C.c_size_t(dst_batch_stride), C.c_void_p(stream))

    def dev_check_degree(self, coeffs_d, status_d, G, m, want_degree, bad_d, stream=0):
        return self.L.hbmpc_dev_check_degree(self.ctx, C.c_void_p(coeffs_d), C.c_void_p(status_d), C.c_size_t(G), C.c_size_t(m),
                                             C.c_size_t(want_degree), C.c_void_p(bad_d), C.c_void_p(stream))

    def dev_check_double_share(self, ct_d, c2t_d, G, m, t, bad_d, stream=0):
        return self.L.hbmpc_dev_check_double_share(self.ctx, C.c_void_p(ct_d), C.c_void_p(c2t_d), C.c_size_t(G), C.c_size_t(m),
                                                   C.c_size_t(t), C.c_void_p(bad_d), C.c_void_p(stream))

    def dev_vandermonde_apply(self, x_d, G, n, d, y_d, stream=0):
        return self._f("dev_vandermonde_apply")(self.ctx, C.c_void_p(x_d), C.c_size_t(G), C.c_size_t(n),
                                                  C.c_size_t(d), C.c_void_p(y_d), C.c_void_p(stream))

    def dev_triple_encode_parties(self, a_d, b_d, r2t_d, G, n, d, parties, tmp_d, y_d, stream=0):
        return self._f("dev_triple_encode_parties")(self.ctx, C.c_void_p(a_d), C.c_void_p(b_d), C.c_void_p(r2t_d), C.c_size_t(G),
                                                      C.c_size_t(n), C.c_size_t(d), C.c_size_t(parties), C.c_void_p(tmp_d),
                                                      C.c_void_p(y_d), C.c_void_p(stream))

    def dev_vandermonde_apply_parties(self, x_d, G, n, d, parties, y_d, stream=0):
        return self._f("dev_vandermonde_apply_parties")(self.ctx, C.c_void_p(x_d), C.c_size_t(G), C.c_size_t(n), C.c_size_t(d),
                                                        C.c_size_t(parties), C.c_void_p(y_d), C.c_void_p(stream))

    def dev_batch_recover(self, sender_ids, evals_d, G, n, d, t, out_d, nco_d=0, status_d=0, summary_d=0, stream=0,
                          p0=False):
        ids = _sz(sender_ids)
        if p0:
            return self._f("dev_batch_recover_p0")(self.ctx, _p(ids), C.c_size_t(len(sender_ids)),
                                                     C.c_void_p(evals_d), C.c_size_t(G), C.c_size_t(n), C.c_size_t(d),
                                                     C.c_size_t(t), C.c_void_p(out_d), C.c_void_p(status_d),
                                                     C.c_void_p(summary_d), C.c_void_p(stream))
        return self._f("dev_batch_recover")(self.ctx, _p(ids), C.c_size_t(len(sender_ids)), C.c_void_p(evals_d),
                                              C.c_size_t(G), C.c_size_t(n), C.c_size_t(d), C.c_size_t(t),
                                              C.c_void_p(out_d), C.c_void_p(nco_d), C.c_void_p(status_d),
                                              C.c_void_p(summary_d), C.c_void_p(stream))

    def dev_batch_recover_strided(self, sender_ids, evals_d, row_stride, G, n, d, t, out_d, p0=False, nco_d=0,
                                  status_d=0, summary_d=0, stream=0):
        ids = _sz(sender_ids)
        return self._f("dev_batch_recover_strided")(self.ctx, _p(ids), C.c_size_t(len(sender_ids)),
                                                      C.c_void_p(evals_d), C.c_size_t(row_stride), C.c_size_t(G),
                                                      C.c_size_t(n), C.c_size_t(d), C.c_size_t(t),
                                                      C.c_int(1 if p0 else 0), C.c_void_p(out_d), C.c_void_p(nco_d),
                                                      C.c_void_p(status_d), C.c_void_p(summary_d), C.c_void_p(stream))

    def dev_batch_recover_coeff_strided(self, sender_ids, evals_d, row_stride, G, n, d, t, k, out_d, status_d=0, summary_d=0, stream=0):
        """a P(0)-shaped decode that keeps coefficient k (exactly d + t + 1 senders)"""
        ids = _sz(sender_ids)
        return self._f("dev_batch_recover_coeff_strided")(self.ctx, _p(ids), C.c_size_t(len(sender_ids)), C.c_void_p(evals_d), C.c_size_t(row_stride),
                                                            C.c_size_t(G), C.c_size_t(n), C.c_size_t(d), C.c_size_t(t), C.c_size_t(k),
                                                            C.c_void_p(out_d), C.c_void_p(status_d), C.c_void_p(summary_d), C.c_void_p(stream))

    def dev_interpolate_degree_check_strided(self, ids, evals_d, row_stride, G, n, d, ws_d, sel_d, status_d, stream=0, groups=1, group_stride=0):
        """do the len(ids) points of every chunk lie on a polynomial of degree <= d?  sel[G][2] = (constant term, coefficient d), status[G]"""
        return self._f("dev_interpolate_degree_check_strided")(self.ctx, _p(_sz(ids)), C.c_size_t(len(ids)), C.c_void_p(evals_d), C.c_size_t(row_stride),
                                                                 C.c_size_t(G), C.c_size_t(n), C.c_size_t(d), C.c_size_t(groups), C.c_size_t(group_stride),
                                                                 C.c_void_p(ws_d), C.c_void_p(sel_d), C.c_void_p(status_d), C.c_void_p(stream))

    def dev_batch_recover_slots(self, sender_ids, row_slots, evals_d, row_stride, G, n, d, t, out_d, p0=False, nco_d=0,
                                status_d=0, summary_d=0, stream=0):
        return self._f("dev_batch_recover_slots")(self.ctx, _p(_sz(sender_ids)), _p(_sz(row_slots)),
                                                    C.c_size_t(len(sender_ids)), C.c_void_p(evals_d), C.c_size_t(row_stride),
                                                    C.c_size_t(G), C.c_size_t(n), C.c_size_t(d), C.c_size_t(t),
                                                    C.c_int(1 if p0 else 0), C.c_void_p(out_d), C.c_void_p(nco_d),
                                                    C.c_void_p(status_d), C.c_void_p(summary_d), C.c_void_p(stream))

    def dev_elem(self, name, ptrs, N, extra=(), stream=0):
        args = [self.ctx] + [C.c_void_p(p) for p in ptrs[: name_inputs(name)]] + [C.c_size_t(e) for e in extra] + \
               [C.c_size_t(N)] + [C.c_void_p(p) for p in ptrs[name_inputs(name):]] + [C.c_void_p(stream)]
        return getattr(self.L, self._pfx + "dev_" + name)(*args)

    def dev_elem_parties(self, name, ptrs, N, parties, extra=(), stream=0):
        """party-batched forms (hbmpc_dev_*_parties): per-party arrays [parties][N], public operands [N]"""
        args = [self.ctx] + [C.c_void_p(p) for p in ptrs[: name_inputs(name)]] + [C.c_size_t(e) for e in extra] + \
               [C.c_size_t(N), C.c_size_t(parties)] + [C.c_void_p(p) for p in ptrs[name_inputs(name):]] + [C.c_void_p(stream)]
        return getattr(self.L, self._pfx + "dev_" + name + "_parties")(*args)

    def dev_fpmul_middle(self, c_d, x_d, y_d, d_d, e_d, rbits_d, rint_d, k, m, N, parties, z_d, rdash_d, open_d, stream=0):
        """finalize_mul + r' + the share TruncPr opens, one launch (hbmpc_dev_fpmul_middle)"""
        return self.L.hbmpc_dev_fpmul_middle(self.ctx, *(C.c_void_p(p) for p in (c_d, x_d, y_d, d_d, e_d, rbits_d, rint_d)),
                                             C.c_size_t(k), C.c_size_t(m), C.c_size_t(N), C.c_size_t(parties),
                                             C.c_void_p(z_d), C.c_void_p(rdash_d), C.c_void_p(open_d), C.c_void_p(stream))

    def dev_triplegen_parties(self, a_d, b_d, r2t_d, rt_d, N, n, t, y_d, z_d, opened_d, c_d, status_d=0, summary_first_d=0, summary_d=0, stream=0):
        """TripleGenNode for all n parties of this device in one call (hbmpc_dev_triplegen_parties): one launch for a small batch, four
        otherwise; returns the ShareErrorCode"""
        return self.L.hbmpc_dev_triplegen_parties(self.ctx, *(C.c_void_p(p) for p in (a_d, b_d, r2t_d, rt_d)), C.c_size_t(N), C.c_size_t(n), C.c_size_t(t),
                                                  *(C.c_void_p(p) for p in (y_d, z_d, opened_d, c_d, status_d, summary_first_d, summary_d)), C.c_void_p(stream))

    def dev_fpmul_parties(self, sender_ids, a_d, b_d, c_d, x_d, y_d, rbits_d, rint_d, k, m, N, n, t, de_ws_d, de_d, z_d, rdash_d, osh_d,
                          cop_d, out_d, status_d=0, summary_first_d=0, summary_d=0, stream=0):
        """FPMulNode for all n parties of this device in one call (hbmpc_dev_fpmul_parties): one launch for a small batch, four or
        five otherwise; returns the ShareErrorCode"""
        ids = (C.c_size_t * len(sender_ids))(*sender_ids)
        return self.L.hbmpc_dev_fpmul_parties(self.ctx, ids, C.c_size_t(len(sender_ids)),
                                              *(C.c_void_p(p) for p in (a_d, b_d, c_d, x_d, y_d, rbits_d, rint_d)),
                                              C.c_size_t(k), C.c_size_t(m), C.c_size_t(N), C.c_size_t(n), C.c_size_t(t),
                                              *(C.c_void_p(p) for p in (de_ws_d, de_d, z_d, rdash_d, osh_d, cop_d, out_d, status_d, summary_first_d, summary_d)),
                                              C.c_void_p(stream))

    def dev_beaver_open_shares_paired(self, a_d, b_d, x_d, y_d, N, parties, de_d, stream=0):
        """de[party][0][N] = a - x, de[party][1][N] = b - y: one P(0) decode over 2 N values per sender opens both"""
        return self._f("dev_beaver_open_shares_paired")(self.ctx, C.c_void_p(a_d), C.c_void_p(b_d), C.c_void_p(x_d), C.c_void_p(y_d),
                                                          C.c_size_t(N), C.c_size_t(parties), C.c_void_p(de_d), C.c_void_p(stream))

    # ---- wire codec ----
    def dev_pack_fvec(self, rows_d, row_stride, G, n_rows, payloads_d, payload_stride_bytes, stream=0):
        return self.L.hbmpc_dev_pack_fvec(self.ctx, C.c_void_p(rows_d), C.c_size_t(row_stride), C.c_size_t(G),
                                          C.c_size_t(n_rows), C.c_void_p(payloads_d), C.c_size_t(payload_stride_bytes),
                                          C.c_void_p(stream))

    def dev_unpack_fvec(self, payloads_d, payload_stride_bytes, payload_bytes, G, n_rows, rows_d, row_stride, status_d,
                        stream=0):
        return self.L.hbmpc_dev_unpack_fvec(self.ctx, C.c_void_p(payloads_d), C.c_size_t(payload_stride_bytes),
                                            C.c_size_t(payload_bytes), C.c_size_t(G), C.c_size_t(n_rows),
                                            C.c_void_p(rows_d), C.c_size_t(row_stride), C.c_void_p(status_d),
                                            C.c_void_p(stream))

    def dev_vandermonde_apply_rows(self, x_rows_d, x_row_stride, G, n, d, tmp_d, y_d, stream=0):
        """x as d + 1 rows of G elements (x_row_stride apart) -> y[n][G]; tmp_d: G * (d + 1) elements or 0"""
        return self._f("dev_vandermonde_apply_rows")(self.ctx, C.c_void_p(x_rows_d), C.c_size_t(x_row_stride), C.c_size_t(G), C.c_size_t(n),
                                                       C.c_size_t(d), C.c_void_p(tmp_d), C.c_void_p(y_d), C.c_void_p(stream))

    def dev_vandermonde_apply_rows_split(self, x_rows_d, x_row_stride, G, n, d, tmp_d, y_d, list_row0, list_rows, K, slices, others_d, stream=0):
        """The producers' mixing step: output rows [list_row0, list_row0 + list_rows) as the parties' lists (slices: (dst_dev, party_stride,
        k0, count), at most two), the other rows party-major to others_d [(party (n - list_rows) + r') K + k]
        (hbmpc_dev_vandermonde_apply_rows_split); others_d = None: the other rows to y_d[row][G] (hbmpc_dev_vandermonde_apply_rows_lists)"""
        class _Slice(C.Structure):
            _fields_ = [("dst_dev", C.c_void_p), ("party_stride", C.c_size_t), ("k0", C.c_size_t), ("count", C.c_size_t)]
        arr = (_Slice * len(slices))(*[_Slice(*sl) for sl in slices])
        common = (self.ctx, C.c_void_p(x_rows_d), C.c_size_t(x_row_stride), C.c_size_t(G), C.c_size_t(n), C.c_size_t(d), C.c_void_p(tmp_d),
                  C.c_void_p(y_d), C.c_size_t(list_row0), C.c_size_t(list_rows), C.c_size_t(K), arr, C.c_size_t(len(slices)))
        if others_d is None:
            return self._f("dev_vandermonde_apply_rows_lists")(*common, C.c_void_p(stream))
        return self._f("dev_vandermonde_apply_rows_split")(*common, C.c_void_p(others_d), C.c_void_p(stream))

    def apply_rows_lists_in_kernel(self, G, n, d) -> bool:
        yes = C.c_int(0)
        rc = self.L.hbmpc_dev_apply_rows_lists_in_kernel(self.ctx, C.c_size_t(G), C.c_size_t(n), C.c_size_t(d), C.byref(yes))
        assert rc == 0, self.last_error()
        return bool(yes.value)

    def dev_vandermonde_apply_strided(self, x_d, G, n, d, y_d, y_row_stride, stream=0):
        return self._f("dev_vandermonde_apply_strided")(self.ctx, C.c_void_p(x_d), C.c_size_t(G), C.c_size_t(n), C.c_size_t(d),
                                                        C.c_void_p(y_d), C.c_size_t(y_row_stride), C.c_void_p(stream))

    def dev_encode_fvec(self, x_d, G, n, d, payloads_d, payload_stride_bytes, stream=0):
        return self._f("dev_encode_fvec")(self.ctx, C.c_void_p(x_d), C.c_size_t(G), C.c_size_t(n), C.c_size_t(d),
                                            C.c_void_p(payloads_d), C.c_size_t(payload_stride_bytes), C.c_void_p(stream))

    def dev_validate_fvec(self, payloads_d, payload_stride_bytes, payload_bytes, G, n_rows, status_d, stream=0):
        return self._f("dev_validate_fvec")(self.ctx, C.c_void_p(payloads_d), C.c_size_t(payload_stride_bytes),
                                              C.c_size_t(payload_bytes), C.c_size_t(G), C.c_size_t(n_rows),
                                              C.c_void_p(status_d), C.c_void_p(stream))

    def dev_pack_shares(self, values_d, N, id, degree, payload_d, stream=0):
        return self.L.hbmpc_dev_pack_shares(self.ctx, C.c_void_p(values_d), C.c_size_t(N), C.c_size_t(id),
                                            C.c_size_t(degree), C.c_void_p(payload_d), C.c_void_p(stream))

    def dev_unpack_shares(self, payload_d, payload_bytes, N, id, degree, values_d, status_d, stream=0):
        return self.L.hbmpc_dev_unpack_shares(self.ctx, C.c_void_p(payload_d), C.c_size_t(payload_bytes), C.c_size_t(N),
                                              C.c_size_t(id), C.c_size_t(degree), C.c_void_p(values_d),
                                              C.c_void_p(status_d), C.c_void_p(stream))

    def dev_validate_canonical(self, a_d, N, status_d, stream=0):
        return self.L.hbmpc_dev_validate_canonical(self.ctx, C.c_void_p(a_d), C.c_size_t(N), C.c_void_p(status_d),
                                                   C.c_void_p(stream))

    def dev_traffic_ubench(self, x_d, G, m, y_d, n, stream=0):
        """x[G][m] -> y[n][G] with no arithmetic: what the memory system delivers for an encode's loads and stores"""
        return self.L.hbmpc_dev_traffic_ubench(self.ctx, C.c_void_p(x_d), C.c_size_t(G), C.c_size_t(m), C.c_void_p(y_d), C.c_size_t(n),
                                               C.c_void_p(stream))

    def dev_modmul_ubench(self, out_d, threads, iters, stream=0):
        return self.L.hbmpc_dev_modmul_ubench(self.ctx, C.c_void_p(out_d), C.c_size_t(threads), C.c_uint32(iters),
                                              C.c_void_p(stream))


_N_IN = {"triple_local": 3, "triple_finalize": 2, "beaver_open_shares": 4, "beaver_finalize": 5, "truncpr_rdash": 1,
         "truncpr_open_share": 3, "truncpr_finalize": 3}


def name_inputs(name):
    return _N_IN[name]
