"""Latency of a decode whose sender set has not been seen before (table building) against a repeat of the same call.
Three regimes per shape: exactly d + t + 1 senders (what BatchRecon issues: it decodes with the first d + t + 1 arrivals,
batch_recon.rs:371-389 -- no OEC round, only the interpolation tables), a set with OEC rounds available (needed + half of
the rest: + the Gao and second-chance tables), both at 64 chunks (host-pointer call, wave-per-chunk kernels) and at 8 192
chunks (the matrix-core kernel: + its byte-digit table, expanded on the device)."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package
from oracle import cref
eng = load_package().Engine(0)
for (n, t, d) in ((16, 5, 5), (31, 10, 10), (64, 21, 14)):
    for G in (64, 8192):
        x = cref.fill_random(1, G * (d + 1)).reshape(G, d + 1, 4)
        rc, y = eng.vandermonde_apply(x, n, d)
        rng = random.Random(n)
        for what, S in (("d+t+1 senders", d + t + 1), ("with OEC rounds", d + t + 1 + (n - d - t - 1) // 2)):
            ids0 = sorted(rng.sample(range(n), S))
            ev0 = np.ascontiguousarray(y[ids0])
            eng.batch_recover(ids0, ev0, n, d, t)
            t0 = time.perf_counter()
            for _ in range(20): eng.batch_recover(ids0, ev0, n, d, t)
            rep = (time.perf_counter() - t0) / 20
            fresh = []
            for _ in range(10):
                ids = sorted(rng.sample(range(n), S))
                ev = np.ascontiguousarray(y[ids])
                t0 = time.perf_counter(); rc, co, nco, st = eng.batch_recover(ids, ev, n, d, t); fresh.append(time.perf_counter() - t0)
                assert rc == 0 and np.array_equal(co, x)
            print(f"n={n} t={t} d={d} G={G:5d} {what:16s}: repeat call {rep*1e3:.3f} ms, first call with a new sender set {np.median(fresh)*1e3:.3f} ms "
                  f"({np.median(fresh)/rep:.1f}x)")
