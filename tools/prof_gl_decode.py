"""Goldilocks cfg3 decode loop for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
eng = load_package().Engine(0, field="goldilocks")
dev = torch.device("cuda:0")
st_ = torch.cuda.Stream(); torch.cuda.set_stream(st_); s = st_.cuda_stream
n, t, d, G = 31, 10, 10, 1 << 20
hi = torch.randint(0, 0xFFFFFFFF, (G, d + 1), dtype=torch.int64, device=dev)
lo = torch.randint(0, 1 << 32, (G, d + 1), dtype=torch.int64, device=dev)
x = (hi << 32) | lo
y = torch.empty((n, G), dtype=torch.int64, device=dev)
co = torch.empty((G, d + 1), dtype=torch.int64, device=dev)
st = torch.empty((G,), dtype=torch.uint8, device=dev)
summ = torch.zeros((4,), dtype=torch.int32, device=dev)
assert eng.dev_vandermonde_apply(x.data_ptr(), G, n, d, y.data_ptr(), s) == 0
for _ in range(12):
    assert eng.dev_batch_recover(list(range(n)), y.data_ptr(), G, n, d, t, co.data_ptr(), 0, st.data_ptr(), summ.data_ptr(), s) == 0
torch.cuda.synchronize()
assert bool((co == x).all())
