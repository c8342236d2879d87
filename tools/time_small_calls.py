"""Latency of the smallest host-pointer calls (one polynomial): what a call site that is NOT batched pays."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package
from oracle import cref
eng = load_package().Engine(0)

def lat(fn, reps=2000):
    for _ in range(20): fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e6

for zc, n, t in ((True, 16, 5), (False, 16, 5), (True, 31, 10), (False, 31, 10)):
    d = t
    eng.set_small_call_staging(zc)
    print("small calls staged through mapped host memory" if zc else "small calls through device buffers + copies")
    x = cref.fill_random(3, d + 1).reshape(1, d + 1, 4)
    rc, y = eng.compute_shares(x, n, d)
    ids = list(range(n)); degs = [d] * n
    vals = np.ascontiguousarray(y[:, 0])
    print(f"n={n} t={t}: compute_shares B=1 {lat(lambda: eng.compute_shares(x, n, d)):.1f} us;"
          f" recover_secret {lat(lambda: eng.recover_secret(ids, degs, vals, n, t)):.1f} us;"
          f" batch_recover G=1 {lat(lambda: eng.batch_recover(ids, y, n, d, t)):.1f} us", flush=True)
    bad = vals.copy(); bad[0, 0] ^= np.uint64(1)
    print(f"   recover_secret with one lie (fallback) {lat(lambda: eng.recover_secret(ids, degs, bad, n, t), 500):.1f} us", flush=True)
    for G in (64, 1024):
        xb = cref.fill_random(5, G * (d + 1)).reshape(G, d + 1, 4)
        rc, yb = eng.compute_shares(xb, n, d)
        print(f"   batch_recover_p0 of {G} polynomials (one call): {lat(lambda: eng.batch_recover_p0(ids, yb, n, d, t), 500):.1f} us", flush=True)
    for B in (64, 1024):
        xb = cref.fill_random(4, B * (d + 1)).reshape(B, d + 1, 4)
        out = np.ones((n, B, 4), dtype=np.uint64)
        print(f"   compute_shares B={B}: {lat(lambda: eng.compute_shares(xb, n, d, out=out), 500):.1f} us", flush=True)
