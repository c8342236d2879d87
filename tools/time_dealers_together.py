"""The dealers' encodes of the producers: hbmpc_dev_vandermonde_apply_parties over all n dealers in one call against a
hbmpc_dev_compute_shares call per dealer, n = 16, degree t and 2t, K polynomials per dealer -- where Producer::dealers_together_max comes from."""
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


def ev_ms(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for n, t in ((16, 5), (7, 2)):
    for d in (t, 2 * t):
        for K in (2000, 8000, 16000, 24000, 40000, 65536, 131072, 262144):
            x = torch.randint(0, 1 << 62, (n, K, d + 1, 4), dtype=torch.int64, device=dev)
            x[..., 3] &= (1 << 60) - 1
            y1 = torch.empty((n, n, K, 4), dtype=torch.int64, device=dev)
            y2 = torch.empty((n, n, K, 4), dtype=torch.int64, device=dev)

            def one():
                assert eng.dev_vandermonde_apply_parties(x.data_ptr(), K, n, d, n, y1.data_ptr(), s) == 0, eng.last_error()

            def each():
                for p in range(n):
                    assert eng.dev_compute_shares(x[p].data_ptr(), K, n, d, y2[p].data_ptr(), s) == 0, eng.last_error()

            a, b = ev_ms(one), ev_ms(each)
            torch.cuda.synchronize()
            assert torch.equal(y1, y2)
            print(f"n={n} d={d:2d} K={K:7d}: together {a:.4f} ms   per dealer {b:.4f} ms", flush=True)
            del x, y1, y2
