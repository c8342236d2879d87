"""Device time of a decode that gets exactly d + t + 1 senders (what BatchRecon passes: it decodes as soon as that many
have arrived) against one that gets all n.  With no OEC round possible the call is two launches (decode, OEC/Gao marking
failures) instead of four.  Reports the slowest host call when it exceeds 2 ms: that is the interpreter's cyclic collector
(~40 ms with torch imported), which is why bench.py switches it off inside its timed region."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
eng = load_package().Engine(0)
dev = torch.device("cuda", 0); ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts); s = ts.cuda_stream
def ev_ms(fn, reps):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import time
    e0.record()
    worst = (0.0, -1)
    for i in range(reps):
        t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
        if dt > worst[0]: worst = (dt, i)
    e1.record(); torch.cuda.synchronize()
    if worst[0] > 2e-3: print(f"   slowest host call: {worst[0]*1e3:.1f} ms at repetition {worst[1]}; cache {eng.cache_stats()}", flush=True)
    return e0.elapsed_time(e1) / reps
def rand_fr(*shape):
    x = torch.randint(0, 1 << 62, shape + (4,), dtype=torch.int64, device=dev); x[..., 3] &= (1 << 60) - 1; return x
for (n, t, d) in ((31, 10, 10), (16, 5, 5)):
    m, needed = d + 1, d + t + 1
    for lg in (10, 12, 14, 16, 18, 20):
        G = 1 << lg
        x = rand_fr(G, m); y = torch.empty((n, G, 4), dtype=torch.int64, device=dev); out = torch.empty((G, m, 4), dtype=torch.int64, device=dev)
        st = torch.empty((G,), dtype=torch.uint8, device=dev); summ = torch.zeros((16,), dtype=torch.int32, device=dev)
        assert eng.dev_vandermonde_apply(x.data_ptr(), G, n, d, y.data_ptr(), s) == 0
        res = []
        for ids in (list(range(n)), list(range(needed))):
            f = lambda: eng.dev_batch_recover(ids, y.data_ptr(), G, n, d, t, out.data_ptr(), 0, st.data_ptr(), summ.data_ptr(), s, p0=True)
            assert f() == 0 and f() == 0
            torch.cuda.synchronize(); assert int(summ[0].item()) == 0
            res.append(ev_ms(f, 100 if lg < 18 else 30))
        print(f"n={n} t={t} G=2^{lg}: P(0) decode with all {n} senders {res[0]*1e3:7.1f} us, with exactly {needed}: {res[1]*1e3:7.1f} us", flush=True)
