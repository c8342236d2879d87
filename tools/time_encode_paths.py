"""Large encodes (y = X * V on the domain, 2^20 and 2^16 chunks) through the three kernels the library has for them: the
matrix-core kernel with the points in pairs (kernels_mfma_bfly.hpp), with one table row per point (set_matrix_cores(3)) and the
FFT kernels (set_matrix_cores(0)).  Outputs are compared byte for byte with each other and, on a sample of chunks, with the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package
from oracle import cref
eng = load_package().Engine(0)
shapes = [(4, 1), (7, 2), (10, 3), (13, 4), (16, 5), (16, 7), (16, 10), (16, 14), (20, 6), (31, 10), (40, 13), (64, 14)]
for lg in (20, 16):
    G = 1 << lg
    for (n, d) in shapes:
        x = cref.fill_random(n * 100 + d, G * (d + 1)).reshape(G, d + 1, 4)
        x_d = eng.dev_alloc(x.nbytes); y_d = eng.dev_alloc(n * G * 32)
        eng.h2d(x_d, x)
        outs, ms = {}, {}
        for name, mode in (("pairs", 1), ("rows", 3), ("fft", 0)):
            eng.set_matrix_cores(mode)
            eng.dev_vandermonde_apply(x_d, G, n, d, y_d); eng.sync()
            y = np.empty((n, G, 4), dtype=np.uint64); eng.d2h(y, y_d); outs[name] = y
            for _ in range(5): eng.dev_vandermonde_apply(x_d, G, n, d, y_d)
            eng.sync()
            t0 = time.perf_counter()
            for _ in range(20): eng.dev_vandermonde_apply(x_d, G, n, d, y_d)
            eng.sync()
            ms[name] = (time.perf_counter() - t0) / 20 * 1e3
        eng.set_matrix_cores(1)
        same = np.array_equal(outs["pairs"], outs["fft"]) and np.array_equal(outs["rows"], outs["fft"])
        idx = np.concatenate([np.arange(8), np.random.default_rng(n).integers(0, G, 56)])
        rc, want = cref.compute_shares(np.ascontiguousarray(x[idx]), n, d)
        ok = rc == 0 and np.array_equal(outs["pairs"][:, idx], want)
        bytes_ = G * (d + 1 + n) * 32
        print(f"n={n:2d} d={d:2d} 2^{lg} chunks: pairs {ms['pairs']:.4f} ms  rows {ms['rows']:.4f}  fft {ms['fft']:.4f}   "
              f"({bytes_ / ms['pairs'] / 1e9:.2f} TB/s)  identical={same} oracle={ok}", flush=True)
        eng.dev_free(x_d); eng.dev_free(y_d)
