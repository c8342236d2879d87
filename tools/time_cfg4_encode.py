"""config 4's fused local product + encode (hbmpc_dev_triple_encode_parties, 16 parties x 381 300 chunks of 11) through the
matrix-core kernel with the products inside (default) and through the fused FFT kernel (set_matrix_cores(0)), same buffers, alternating."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from __graft_entry__ import load_package
eng = load_package().Engine(0)
n, t = 16, 5
d, G, P = 2 * t, 381300, 16
N = P * G * (d + 1)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(5)
def rnd():
    lo = torch.randint(0, 1 << 62, (N, 3), dtype=torch.int64, device=dev, generator=g)
    hi = torch.randint(0, 0x73EDA753299D7D48, (N, 1), dtype=torch.int64, device=dev, generator=g)
    return torch.cat([lo, hi], dim=-1).contiguous()
a, b, r = rnd(), rnd(), rnd()
ys = {}
def run(mode, reps):
    eng.set_matrix_cores(mode)
    y = torch.empty((P, n, G, 4), dtype=torch.int64, device=dev)
    for _ in range(3): assert eng.dev_triple_encode_parties(a.data_ptr(), b.data_ptr(), r.data_ptr(), G, n, d, P, 0, y.data_ptr()) == 0
    eng.sync(); t0 = time.perf_counter()
    for _ in range(reps): eng.dev_triple_encode_parties(a.data_ptr(), b.data_ptr(), r.data_ptr(), G, n, d, P, 0, y.data_ptr())
    eng.sync()
    return (time.perf_counter() - t0) / reps * 1e3, y
for _ in range(100): eng.dev_triple_encode_parties(a.data_ptr(), b.data_ptr(), r.data_ptr(), G, n, d, P, 0, torch.empty((P, n, G, 4), dtype=torch.int64, device=dev).data_ptr())
eng.sync()
for rnd_ in range(3):
    m1, y1 = run(1, 30)
    m0, y0 = run(0, 30)
    print(f"round {rnd_}: matrix cores with the products inside {m1:.3f} ms, fused FFT kernel {m0:.3f} ms, identical={torch.equal(y1, y0)}", flush=True)
