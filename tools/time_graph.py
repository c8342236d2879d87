"""Small-batch latency of the fpmul pipeline (the reference's real regime: a few hundred elements per protocol
message): eager hbmpc_dev_* calls vs the same call sequence captured once into a HIP graph and replayed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
eng = pkg.Engine(0)
dev = torch.device("cuda", 0)
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts); s = ts.cuda_stream
n, t, k, m = 16, 5, 16, 4
import bench
for N in (64, 1024, 16384):
    fp = bench.setup_fpmul(eng, torch, dev, s, n, t, N, k, m)
    fp.run(check=True)                                # one checked run: every decode reports zero failures
    for _ in range(3): fp.run(check=False)          # warm: tables and scratch exist, nothing allocates any more
    torch.cuda.synchronize()
    def timeit(fn, reps=30):
        import time
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter(); e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, e0.elapsed_time(e1) / reps
    wall_e, gpu_e = timeit(lambda: fp.run(check=False))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=ts):
        fp.run(check=False)
    wall_g, gpu_g = timeit(g.replay)
    out = fp.download("out")
    print(f"fpmul n={n} N={N}: eager {wall_e:.3f} ms wall / {gpu_e:.3f} ms stream;  HIP graph replay {wall_g:.3f} ms wall / {gpu_g:.3f} ms stream")
    fp.close()
