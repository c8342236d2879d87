"""Latency of a decode whose sender set has not been seen before (host-side table building + upload), vs a repeat."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package
from oracle import cref
eng = load_package().Engine(0)
for (n, t, d) in ((16, 5, 5), (31, 10, 10), (64, 21, 21)):
    G = 64
    x = cref.fill_random(1, G * (d + 1)).reshape(G, d + 1, 4)
    rc, y = eng.vandermonde_apply(x, n, d)
    rng = random.Random(n)
    ids0 = list(range(n))
    eng.batch_recover(ids0, y, n, d, t)
    t0 = time.perf_counter()
    for _ in range(20): eng.batch_recover(ids0, y, n, d, t)
    rep = (time.perf_counter() - t0) / 20
    fresh = []
    for _ in range(10):
        ids = sorted(rng.sample(range(n), d + t + 1 + (n - d - t - 1) // 2))
        ev = np.ascontiguousarray(y[ids])
        t0 = time.perf_counter(); rc, co, nco, st = eng.batch_recover(ids, ev, n, d, t); fresh.append(time.perf_counter() - t0)
        assert rc == 0 and np.array_equal(co, x)
    print(f"n={n} t={t} d={d}: repeat call {rep*1e3:.3f} ms, first call with a new sender set {np.median(fresh)*1e3:.3f} ms")
