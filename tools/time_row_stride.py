"""Does the distance between the n output rows matter?  compute_shares / apply_vandermonde of BASELINE configs[1] (y[16][2^20]: rows 32 MiB apart) with
the rows at padded strides (hbmpc_dev_vandermonde_apply_strided), and config 3's decode with its sender rows at padded strides."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from __graft_entry__ import load_package
eng = load_package().Engine(0)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
def rnd(*shape):
    lo = torch.randint(0, 1 << 62, (*shape, 3), dtype=torch.int64, device=dev, generator=g)
    hi = torch.randint(0, 0x73EDA753299D7D48, (*shape, 1), dtype=torch.int64, device=dev, generator=g)
    return torch.cat([lo, hi], dim=-1).contiguous()
def timeit(fn, reps=50):
    for _ in range(200): fn()
    eng.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    eng.sync()
    return (time.perf_counter() - t0) / reps * 1e3
n, d, G = 16, 5, 1 << 20
x = rnd(G, d + 1)
for pad in (0, 8, 64, 128, 1024, 4096 + 64, 65536 + 192):
    y = torch.empty((n, G + pad, 4), dtype=torch.int64, device=dev)
    ms = timeit(lambda: eng.dev_vandermonde_apply_strided(x.data_ptr(), G, n, d, y.data_ptr(), G + pad, 0))
    print(f"encode n=16 d=5 2^20 chunks, output row stride G + {pad:6d} elements: {ms:.4f} ms", flush=True)
    del y
n, d, t = 31, 10, 10
x3 = rnd(G, d + 1)
for pad in (0, 64, 1024 + 64):
    y = torch.empty((n, G + pad, 4), dtype=torch.int64, device=dev)
    assert eng.dev_vandermonde_apply_strided(x3.data_ptr(), G, n, d, y.data_ptr(), G + pad, 0) == 0
    out = torch.empty((G, d + 1, 4), dtype=torch.int64, device=dev)
    st = torch.empty((G,), dtype=torch.uint8, device=dev)
    ids = list(range(d + t + 1))
    ms = timeit(lambda: eng.dev_batch_recover_strided(ids, y.data_ptr(), G + pad, G, n, d, t, out.data_ptr(), status_d=st.data_ptr()), 30)
    print(f"decode n=31 d=t=10 2^20 chunks, sender row stride G + {pad:6d}: {ms:.4f} ms", flush=True)
    del y
