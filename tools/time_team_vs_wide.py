import os, sys, gc
sys.path.insert(0, "/root/repo")
import torch
from __graft_entry__ import load_package
eng = load_package().Engine(0)
dev = torch.device("cuda", 0); ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts); s = ts.cuda_stream
gc.disable()
def ev_us(fn, reps=300):
    for _ in range(30): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
def rand_fr(*shape):
    x = torch.randint(0, 1 << 62, shape + (4,), dtype=torch.int64, device=dev); x[..., 3] &= (1 << 60) - 1; return x
for (n, t, d) in ((31, 10, 10), (16, 5, 5), (16, 5, 10), (7, 2, 2)):
    m, needed = d + 1, d + t + 1
    for G in (512, 1024, 2048, 3072, 4096, 6144, 8192):
        x = rand_fr(G, m); y = torch.empty((n, G, 4), dtype=torch.int64, device=dev); out = torch.empty((G, m, 4), dtype=torch.int64, device=dev)
        st = torch.empty((G,), dtype=torch.uint8, device=dev); summ = torch.zeros((16,), dtype=torch.int32, device=dev); nco = torch.empty((G,), dtype=torch.int32, device=dev)
        assert eng.dev_vandermonde_apply(x.data_ptr(), G, n, d, y.data_ptr(), s) == 0
        row = []
        for mode in ("team", "wide"):
            eng.set_matrix_cores(1 if mode == "team" else 0, 1)
            eng.set_small_batch_chunks(0 if mode == "team" else 8192)
            full = lambda: eng.dev_batch_recover(list(range(n)), y.data_ptr(), G, n, d, t, out.data_ptr(), nco.data_ptr(), st.data_ptr(), summ.data_ptr(), s)
            p0n = lambda: eng.dev_batch_recover(list(range(needed)), y.data_ptr(), G, n, d, t, out.data_ptr(), 0, st.data_ptr(), summ.data_ptr(), s, p0=True)
            fulln = lambda: eng.dev_batch_recover(list(range(needed)), y.data_ptr(), G, n, d, t, out.data_ptr(), nco.data_ptr(), st.data_ptr(), summ.data_ptr(), s)
            assert full() == 0 and p0n() == 0 and fulln() == 0
            torch.cuda.synchronize(); assert int(summ[0].item()) == 0
            row.append((ev_us(full), ev_us(fulln), ev_us(p0n)))
        a, b = row
        print(f"n={n} t={t} d={d} G={G:5d}: all senders: matrix cores {a[0]:5.1f} us (wave per chunk {b[0]:5.1f}) | exactly d+t+1: {a[1]:5.1f} ({b[1]:5.1f}) | P(0), exactly d+t+1: {a[2]:5.1f} ({b[2]:5.1f})", flush=True)
