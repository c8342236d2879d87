import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package
from oracle import cref
eng = load_package().Engine(0)
n, d, B = 16, 5, 1 << 14
x = cref.fill_random(1, B * (d + 1)).reshape(B, d + 1, 4)
y = np.zeros((n, B, 4), dtype=np.uint64)
def T(label, fn, reps=5):
    fn(); t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    print(f"{label:28s} {(time.perf_counter()-t0)/reps*1e3:8.3f} ms"); return r
T("full host call", lambda: eng.compute_shares(x, n, d))
T("np.zeros out", lambda: np.zeros((n, B, 4), dtype=np.uint64))
ptrs = []
def alloc2(): ptrs.append((eng.dev_alloc(x.nbytes), eng.dev_alloc(y.nbytes)))
T("2x dev_alloc", alloc2)
xd, yd = ptrs[0]
def h2d(): eng.h2d(xd, x); eng.sync()
T("h2d 3 MB + sync", h2d)
def run(): eng.dev_compute_shares(xd, B, n, d, yd); eng.sync()
T("kernel + sync", run)
def d2h(): eng.d2h(y, yd); eng.sync()
T("d2h 8 MB + sync", d2h)
def free2():
    a, b = ptrs.pop(); eng.dev_free(a); eng.dev_free(b)
T("2x dev_free", free2, reps=len(ptrs) - 2)
