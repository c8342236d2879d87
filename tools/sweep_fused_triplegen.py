#!/usr/bin/env python3
"""TripleGenNode at small batches (all parties on one device): hbmpc_dev_triplegen_parties as ONE launch (a workgroup per chunk of
2t + 1 triples, csrc/kernels_triplegen_wg.hpp) against its four separate launches, eager and as a HIP graph, over batch sizes --
where hbmpc_set_fused_triplegen's default comes from.
    python tools/sweep_fused_triplegen.py [chunks ...] > gpurun_out/fused_triplegen.txt"""
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
    sizes = [int(v) for v in sys.argv[1:]] or [10, 50, 100, 200, 300, 500, 1000, 2000]
    dev = torch.device("cuda", 0)
    pkg = load_package()
    eng = pkg.Engine(0)
    ts = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    torch.cuda.set_stream(ts)
    stream = ts.cuda_stream
    n, t = 16, 5
    print(f"triple generation n={n} t={t}; ms per batch of `chunks` x {2 * t + 1} triples per party: eager / replayed graph")
    print(f"{'chunks':>7} {'triples':>8} {'one eager':>11} {'one graph':>11} {'four eager':>11} {'four graph':>11}")
    for groups in sizes:
        N = groups * (2 * t + 1)
        row = []
        for fused in (1 << 30, 0):
            eng.L.hbmpc_set_fused_triplegen(eng.ctx, C.c_size_t(fused))
            tg = pkg.pipelines.TripleGen(eng, n, t, N, stream)
            a, b, r = (bench._rand_fr(torch, dev, N) for _ in range(3))
            bench._share_on_device(eng, torch, dev, stream, a, n, t, tg.a)
            bench._share_on_device(eng, torch, dev, stream, b, n, t, tg.b)
            bench._share_on_device(eng, torch, dev, stream, r, n, t, tg.rt)
            bench._share_on_device(eng, torch, dev, stream, r, n, 2 * t, tg.r2t)
            tg.run(check=True)
            tg.run(check=False)
            e = ev_time(lambda: tg.run(check=False))
            tg.capture()
            g = ev_time(tg.replay)
            tg.close()
            row += [e, g]
        print(f"{groups:7d} {N:8d} " + " ".join(f"{v:11.4f}" for v in row), flush=True)


if __name__ == "__main__":
    main()
