"""Does splitting one compute_shares batch over two streams (two concurrent kernels) overlap memory and arithmetic
better than one launch?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
eng = load_package().Engine(0)
dev = torch.device("cuda", 0)
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
n, d, B = 16, 5, 1 << 20
x = torch.randint(0, 2**62, (B * (d + 1), 4), dtype=torch.int64, device=dev)
y = torch.empty((n, B, 4), dtype=torch.int64, device=dev)
ya = torch.empty((n, B // 2, 4), dtype=torch.int64, device=dev)
yb = torch.empty((n, B // 2, 4), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
def one():
    eng.dev_compute_shares(x.data_ptr(), B, n, d, y.data_ptr(), s1.cuda_stream)
def two():
    eng.dev_compute_shares(x.data_ptr(), B // 2, n, d, ya.data_ptr(), s1.cuda_stream)
    eng.dev_compute_shares(x.data_ptr() + (B // 2) * (d + 1) * 32, B // 2, n, d, yb.data_ptr(), s2.cuda_stream)
def four(parts=4):
    q = B // parts
    outs = four.outs
    for i in range(parts):
        eng.dev_compute_shares(x.data_ptr() + i * q * (d + 1) * 32, q, n, d, outs[i].data_ptr(), (s1 if i % 2 == 0 else s2).cuda_stream)
four.outs = [torch.empty((n, B // 4, 4), dtype=torch.int64, device=dev) for _ in range(4)]
import time
for name, fn in (("one launch 2^20", one), ("two streams 2 x 2^19", two), ("two streams 4 x 2^18", four)):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per 2^20 secrets")
