"""Runs cfg3 encode / decode / decode_p0 a few times on device-resident data (for rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from __graft_entry__ import load_package
from oracle import cref
pkg = load_package(); eng = pkg.Engine(0); dev = torch.device("cuda", 0)
s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s); st = s.cuda_stream
n, t, d, G = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (31, 10, 10, 1 << 20)))
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
host = cref.fill_random(0xC0FFEE02, G * (d + 1)).reshape(G, d + 1, 4)
x = torch.from_numpy(host.view(np.int64)).to(dev)
y = torch.empty((n, G, 4), dtype=torch.int64, device=dev)
co = torch.empty((G, d + 1, 4), dtype=torch.int64, device=dev)
sec = torch.empty((G, 4), dtype=torch.int64, device=dev)
stt = torch.empty((G,), dtype=torch.uint8, device=dev)
summ = torch.zeros((4,), dtype=torch.int32, device=dev)
ids = list(range(n))
def tm(fn):
    fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
print("encode ms", tm(lambda: eng.dev_vandermonde_apply(x.data_ptr(), G, n, d, y.data_ptr(), st)))
print("decode ms", tm(lambda: eng.dev_batch_recover(ids, y.data_ptr(), G, n, d, t, co.data_ptr(), 0, stt.data_ptr(), summ.data_ptr(), st)))
print("decode_p0 ms", tm(lambda: eng.dev_batch_recover(ids, y.data_ptr(), G, n, d, t, sec.data_ptr(), 0, stt.data_ptr(), summ.data_ptr(), st, p0=True)))
assert bool((co == x).all())
