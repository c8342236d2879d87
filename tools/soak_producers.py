"""Soak: tests/test_gpu_producers.py's RanSha / RanDouSha checks against the oracle (outputs bit for bit, verdicts with a corrupted dealer)
on random (n, t, batch size) in both fields (not part of the pytest suite; run on the GPU box):
    python -u tools/soak_producers.py FIRST LAST [MAX_SECONDS]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_producers as T
first, last = int(sys.argv[1]), int(sys.argv[2])
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 600.0
rs = getattr(T.test_ransha_matches_the_oracle, "__wrapped__", T.test_ransha_matches_the_oracle)
rd = getattr(T.test_randousha_matches_the_oracle, "__wrapped__", T.test_randousha_matches_the_oracle)
mid = getattr(T.test_producers_at_the_batch_sizes_of_the_reference_node, "__wrapped__", T.test_producers_at_the_batch_sizes_of_the_reference_node)
t0, done = time.time(), 0
for seed in range(first, last):
    rng = random.Random(seed)
    n = rng.randint(4, 22)
    t = rng.randint(1, (n - 1) // 3)
    K = rng.choice([rng.randint(2, 12), rng.randint(13, 70), rng.randint(71, 300)])
    field = rng.choice(["fr", "goldilocks"])
    rs(field, n, t, K)
    rd(field, n, t, K)
    done += 2
    if seed % 4 == 0:  # a mid-size batch too (sampled columns): dealers in one launch, the verifiers' rows party-major, one decode per kind of verifier
        Km = rng.choice([rng.randint(301, 1100), rng.randint(1101, 2600), rng.randint(2601, 7000)])
        mid(field, n, t, Km)
        done += 1
    if seed % 5 == 0:
        print(f"seed {seed}: {done} cases ok, {time.time() - t0:.0f} s (last: {field} n={n} t={t} K={K})", flush=True)
    if time.time() - t0 > budget:
        print(f"time budget reached at seed {seed}", flush=True)
        break
print(f"soak ok: {done} cases, seeds {first}..{seed}", flush=True)
