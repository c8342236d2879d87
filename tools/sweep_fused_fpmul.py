#!/usr/bin/env python3
"""FPMulNode at small batches (all parties on one device): hbmpc_dev_fpmul_parties as ONE launch (a wave per element,
csrc/kernels_fpmul_wave.hpp) against its five separate launches, eager and as a HIP graph, over batch sizes -- where
hbmpc_set_fused_fpmul's default comes from.
    python tools/sweep_fused_fpmul.py [sizes ...] > gpurun_out/fused_fpmul.txt"""
import ctypes as C
import gc
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402


def ev_time(fn, reps=30, warm=3, warm_seconds=0.15):
    t0 = time.perf_counter()
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    while time.perf_counter() - t0 < warm_seconds:
        fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gc.disable()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    gc.enable()
    return e0.elapsed_time(e1) / reps


def main():
    sizes = [int(v) for v in sys.argv[1:]] or [64, 256, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 262144]
    dev = torch.device("cuda", 0)
    eng = load_package().Engine(0)
    ts = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    torch.cuda.set_stream(ts)
    stream = ts.cuda_stream
    n, t, k, m = 16, 5, 16, 4
    print(f"fpmul n={n} t={t} (k, f)=({k}, {m}); ms per multiplication batch: eager / replayed graph")
    print(f"{'elements':>9} {'one eager':>12} {'one graph':>12} {'five eager':>12} {'five graph':>12} {'four eager':>12} {'four graph':>12}")
    never = (1 << 64) - 1
    if True:
        for N in sizes:
            row = []
            # one launch (a wave per element) | five launches | four (the first open forms its senders' shares as it loads them)
            for fused, pair_min in ((1 << 30, never), (0, never), (0, 0)):
                if fused and N > 16384:  # a wave per element is far behind by then
                    row += [float("nan")] * 2
                    continue
                eng.L.hbmpc_set_fused_fpmul(eng.ctx, C.c_size_t(fused))
                eng.L.hbmpc_set_fpmul_pair_decode(eng.ctx, C.c_size_t(pair_min))
                fp = bench.setup_fpmul(eng, torch, dev, stream, n, t, N, k, m)
                fp.run(check=True)
                fp.run(check=False)
                e = ev_time(lambda: fp.run(check=False))
                fp.capture()
                g = ev_time(fp.replay)
                fp.close()
                row += [e, g]
            print(f"{N:9d} " + " ".join(f"{v:12.4f}" for v in row), flush=True)


if __name__ == "__main__":
    main()
