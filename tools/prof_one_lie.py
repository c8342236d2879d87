"""Kernel times of a one-polynomial recover_secret with one lie (rocprofv3 --kernel-trace --stats -- python3 tools/prof_one_lie.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package
from oracle import cref
eng = load_package().Engine(0)
for n, t in ((16, 5), (31, 10)):
    d = t
    x = cref.fill_random(3, d + 1).reshape(1, d + 1, 4)
    rc, y = eng.compute_shares(x, n, d)
    ids = list(range(n)); degs = [d] * n
    bad = np.ascontiguousarray(y[:, 0]); bad[0, 0] ^= np.uint64(1)
    for _ in range(200):
        rc, co, sec = eng.recover_secret(ids, degs, bad, n, t)
        assert rc == 0
