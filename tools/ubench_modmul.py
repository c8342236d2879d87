"""Register-resident modmul chain at 1/2/4/8 waves per SIMD, both field implementations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
pkg = load_package()
eng = pkg.Engine(0)
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s); st = s.cuda_stream
for impl in ("u29", "sat32"):
    eng.set_impl(impl)
    for wps in (1, 2, 3, 4, 8):
        threads, iters = 256 * 256 * wps, 1000
        buf = torch.empty((threads, 4), dtype=torch.int64, device=dev)
        for _ in range(2): eng.dev_modmul_ubench(buf.data_ptr(), threads, iters, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): eng.dev_modmul_ubench(buf.data_ptr(), threads, iters, st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        rate = threads * (2 * iters + 2) / ms * 1e3
        print(f"{impl} waves/SIMD={wps}: {rate:.3e} modmul/s  ({2.4e9*1024*64/rate:.0f} cycles per wave-modmul per SIMD @2.4GHz)")
