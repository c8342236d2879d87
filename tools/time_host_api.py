"""PCIe-inclusive rate of the host-pointer API (hbmpc_compute_shares with numpy arrays in pageable host memory)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package
from oracle import cref
eng = load_package().Engine(0)
n, d = 16, 5
for lg in (14, 17, 20):
    B = 1 << lg
    x = cref.fill_random(1, B * (d + 1)).reshape(B, d + 1, 4)
    eng.compute_shares(x[:1024], n, d)
    t0 = time.perf_counter(); reps = 3
    for _ in range(reps): rc, y = eng.compute_shares(x, n, d)
    dt = (time.perf_counter() - t0) / reps
    by = (d + 1 + n) * 32 * B
    print(f"host API compute_shares B=2^{lg}: {dt*1e3:.2f} ms  {n*B/dt:.3e} share-evals/s  {by/dt/1e9:.1f} GB/s over the boundary (incl. numpy output allocation)")
    y = np.ones((n, B, 4), dtype=np.uint64)  # pages already touched, reused across calls (what a caller that recycles its buffers sees)
    t0 = time.perf_counter()
    for _ in range(reps): rc, y = eng.compute_shares(x, n, d, out=y)
    dt = (time.perf_counter() - t0) / reps
    print(f"host API compute_shares B=2^{lg}, output buffer reused: {dt*1e3:.2f} ms  {n*B/dt:.3e} share-evals/s  {by/dt/1e9:.1f} GB/s over the boundary")
    eng.compute_shares_seeded(bytes(32), x[:1024, 0], n, d)
    sec = np.ascontiguousarray(x[:, 0])
    t0 = time.perf_counter()
    for _ in range(reps): rc, y = eng.compute_shares_seeded(bytes(32), sec, n, d)
    dt = (time.perf_counter() - t0) / reps
    print(f"host API compute_shares_seeded B=2^{lg}: {dt*1e3:.2f} ms  {n*B/dt:.3e} share-evals/s (uploads the secrets only)")
