"""Soak: tests/test_gpu_pipelines.py::test_triplegen_forms_random (hbmpc_dev_triplegen_parties: one launch and four leave the same bytes)
on many more seeds (not part of the pytest suite; run on the GPU box):  python -u tools/soak_triplegen.py FIRST LAST [MAX_SECONDS]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_pipelines as T
from __graft_entry__ import load_package
first, last = int(sys.argv[1]), int(sys.argv[2])
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 600.0
pkg = load_package()
eng = pkg.Engine(0)
fn = getattr(T.test_triplegen_forms_random, "__wrapped__", T.test_triplegen_forms_random)
t0, done = time.time(), 0
for seed in range(first, last):
    fn((pkg, eng), seed)
    done += 1
    if seed % 16 == 0:  # a mid-size batch: the four launches take the matrix-core product + encode and decodes there, the one launch is forced
        rng = np.random.default_rng(seed)
        t = int(rng.integers(1, 6))
        groups = int(rng.integers(1025, 7000))
        tamper = [(str(rng.choice(["a", "b", "r2t", "rt"])), int(rng.integers(0, 3 * t + 1)), int(rng.integers(0, groups * (2 * t + 1)))) for _ in range(int(rng.integers(0, 3)))]
        T._triplegen_forms(pkg, eng, 3 * t + 1, t, groups, 77000 + seed, tamper)
        done += 1
    if seed % 10 == 0:
        print(f"seed {seed}: {done} cases ok, {time.time() - t0:.0f} s", flush=True)
    if time.time() - t0 > budget:
        print(f"time budget reached at seed {seed}", flush=True)
        break
print(f"soak ok: {done} cases, seeds {first}..{seed}", flush=True)
