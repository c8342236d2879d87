"""What would the matrix-core decode cost with fewer verify rows?  Device time of hbmpc_dev_batch_recover at 2^20 chunks, n = 32,
d = 10, for t = 10 (21 table rows, two roles: the BASELINE shape) and t = 1, 2, 3 (12, 13, 14 rows: one role, table resident)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package

pkg = load_package()
eng = pkg.Engine(0)
dev = torch.device("cuda", 0)
ts = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(ts)
s = ts.cuda_stream
G, n, d = 1 << 20, 32, 10
x = torch.randint(0, 1 << 62, (G, d + 1, 4), dtype=torch.int64, device=dev)
x[..., 3] &= (1 << 60) - 1
y = torch.empty((n, G, 4), dtype=torch.int64, device=dev)
out = torch.empty((G, d + 1, 4), dtype=torch.int64, device=dev)
st = torch.empty((G,), dtype=torch.uint8, device=dev)
summ = torch.zeros((16,), dtype=torch.int32, device=dev)
assert eng.dev_vandermonde_apply(x.data_ptr(), G, n, d, y.data_ptr(), s) == 0
torch.cuda.synchronize()
for t in (10, 1, 2, 3, 10):
    S = d + t + 1
    ids = list(range(S))
    fn = lambda: eng.dev_batch_recover(ids, y.data_ptr(), G, n, d, t, out.data_ptr(), 0, st.data_ptr(), summ.data_ptr(), s)
    for _ in range(10):
        assert fn() == 0, eng.last_error()
    torch.cuda.synchronize()
    assert int(summ[0]) == 0 and torch.equal(out, x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 100
    byt = G * 32 * (S + d + 1)
    print(f"t={t:2d} rows={t + d + 1:2d} points={S}: {ms:.4f} ms  {byt / ms / 1e9:.2f} TB/s algorithmic", flush=True)
