"""Times hbmpc_dev_compute_shares (BASELINE configs[1]) without checking results: for A/B experiments with
deliberately incomplete kernels.  usage: time_cfg2.py [lib.so]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
pkg = load_package()
if len(sys.argv) > 1:
    pkg.hbmpc.LIB_PATH = os.path.abspath(sys.argv[1])
eng = pkg.Engine(0)
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s); st = s.cuda_stream
n, d, B = 16, 5, 1 << 20
x = torch.randint(0, 2**62, (B * (d + 1), 4), dtype=torch.int64, device=dev)
y = torch.empty((n * B, 4), dtype=torch.int64, device=dev)
for _ in range(20): eng.dev_compute_shares(x.data_ptr(), B, n, d, y.data_ptr(), st)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): eng.dev_compute_shares(x.data_ptr(), B, n, d, y.data_ptr(), st)
e1.record(); torch.cuda.synchronize()
print(f"{sys.argv[1] if len(sys.argv) > 1 else 'default'}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per launch")
