"""Where the time of a one-polynomial host call goes: tiny transfers, launches, synchronisation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package
from oracle import cref
eng = load_package().Engine(0)
n, t = 16, 5; d = t
x = cref.fill_random(3, d + 1).reshape(1, d + 1, 4)
rc, y = eng.compute_shares(x, n, d)
def lat(label, fn, reps=2000):
    for _ in range(20): fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    print(f"{label:44s} {(time.perf_counter() - t0) / reps * 1e6:7.1f} us", flush=True)
xd = eng.dev_alloc(4096); yd = eng.dev_alloc(4096); od = eng.dev_alloc(4096); sd = eng.dev_alloc(64); st = eng.dev_alloc(64)
small = np.zeros(64, dtype=np.uint64)
lat("ctypes call overhead (stream sync, idle)", lambda: eng.sync())
lat("h2d 512 B pageable (async call)", lambda: eng.h2d(xd, small))
lat("h2d 512 B + sync", lambda: (eng.h2d(xd, small), eng.sync()))
lat("d2h 512 B + sync", lambda: (eng.d2h(small, xd), eng.sync()))
lat("4x d2h 512 B + sync", lambda: (eng.d2h(small, xd), eng.d2h(small, xd), eng.d2h(small, xd), eng.d2h(small, xd), eng.sync()))
eng.h2d(xd, x)
lat("dev_compute_shares B=1 launch + sync", lambda: (eng.dev_compute_shares(xd, 1, n, d, yd), eng.sync()))
ids = list(range(n))
lat("dev_batch_recover G=1 (3 launches) + sync", lambda: (eng.dev_batch_recover(ids, yd, 1, n, d, t, od, 0, st, sd), eng.sync()))
lat("dev_batch_recover G=1 enqueue only", lambda: eng.dev_batch_recover(ids, yd, 1, n, d, t, od, 0, st, sd)); eng.sync()
lat("host compute_shares B=1", lambda: eng.compute_shares(x, n, d))
lat("host batch_recover G=1", lambda: eng.batch_recover(ids, y, n, d, t))
