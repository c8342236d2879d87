"""Where should batch_recover move from the lane-per-chunk kernels to the matrix cores?  Device time per call (HIP events
via torch, tables cached) of hbmpc_dev_batch_recover / _p0 and hbmpc_dev_vandermonde_apply for G = 2^10 .. 2^18 chunks with
the matrix-core path forced on (min_chunks = 1) and off, for the BASELINE shapes.  The threshold the library ships
(hbmpc_set_matrix_cores, default 65 536 chunks) should sit where the two curves cross."""
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


def ev_ms(fn, reps):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def rand_fr(*shape):
    x = torch.randint(0, 1 << 62, shape + (4,), dtype=torch.int64, device=dev)
    x[..., 3] &= (1 << 60) - 1   # < 2^252 < r: canonical
    return x


for (n, t, d) in ((31, 10, 10), (16, 5, 5), (16, 5, 10)):
    m = d + 1
    ids = list(range(n))
    print(f"n={n} t={t} d={d}", flush=True)
    for lg in range(10, 19):
        G = 1 << lg
        x = rand_fr(G, m)
        y = torch.empty((n, G, 4), dtype=torch.int64, device=dev)
        out = torch.empty((G, m, 4), dtype=torch.int64, device=dev)
        st = torch.empty((G,), dtype=torch.uint8, device=dev)
        summ = torch.zeros((16,), dtype=torch.int32, device=dev)
        nco = torch.empty((G,), dtype=torch.int32, device=dev)
        assert eng.dev_vandermonde_apply(x.data_ptr(), G, n, d, y.data_ptr(), s) == 0
        row = {}
        for name, on in (("mfma", True), ("lane", False), ("mfma-nowide", True)):
            if name == "mfma-nowide" and lg > 14:
                continue
            # G <= 8192 goes to the wave-per-chunk kernels before either of the others is considered: the third arm
            # switches those off to see where the matrix cores would overtake them
            eng.set_small_batch_chunks(0 if name == "mfma-nowide" else 8192)
            eng.set_matrix_cores(on, 1)
            reps = 200 if lg < 15 else 50
            full = lambda: eng.dev_batch_recover(ids, y.data_ptr(), G, n, d, t, out.data_ptr(), nco.data_ptr(), st.data_ptr(), summ.data_ptr(), s)
            p0 = lambda: eng.dev_batch_recover(ids, y.data_ptr(), G, n, d, t, out.data_ptr(), 0, st.data_ptr(), summ.data_ptr(), s, p0=True)
            enc = lambda: eng.dev_vandermonde_apply(x.data_ptr(), G, n, d, y.data_ptr(), s)
            assert full() == 0 and p0() == 0
            torch.cuda.synchronize()
            assert int(summ[0].item()) == 0, "flagged chunks in a clean batch"
            row[name] = (ev_ms(full, reps), ev_ms(p0, reps), ev_ms(enc, reps))
        eng.set_small_batch_chunks(8192)
        if "mfma-nowide" in row:
            c = row["mfma-nowide"]
            print(f"  G=2^{lg:<2}  wave-per-chunk kernels off, matrix cores on: decode {c[0]*1e3:8.1f} us | p0 {c[1]*1e3:8.1f} | encode {c[2]*1e3:8.1f}")
        a, b = row["mfma"], row["lane"]
        print(f"  G=2^{lg:<2}  decode mfma {a[0]*1e3:8.1f} us  lane {b[0]*1e3:8.1f} us | p0 mfma {a[1]*1e3:8.1f} lane {b[1]*1e3:8.1f} |"
              f" encode mfma {a[2]*1e3:8.1f} lane {b[2]*1e3:8.1f}", flush=True)
