cd $GRAFT_REPO_ROOT
python3 - <<PY
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from __graft_entry__ import load_package
eng = load_package().Engine(0, field="goldilocks")
dev = torch.device("cuda:0")
st_ = torch.cuda.Stream(); torch.cuda.set_stream(st_); s = st_.cuda_stream
def ev(fn, reps=50):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (n, t, d) in ((31, 10, 10), (16, 5, 5), (16, 5, 10)):
    G = 1 << 20
    hi = torch.randint(0, 0xFFFFFFFF, (G, d + 1), dtype=torch.int64, device=dev); lo = torch.randint(0, 1 << 32, (G, d + 1), dtype=torch.int64, device=dev)
    x = (hi << 32) | lo
    y = torch.empty((n, G), dtype=torch.int64, device=dev); co = torch.empty((G, d + 1), dtype=torch.int64, device=dev)
    st = torch.empty((G,), dtype=torch.uint8, device=dev); summ = torch.zeros((4,), dtype=torch.int32, device=dev)
    out = []
    for mf in (True, False):
        eng.set_matrix_cores(mf, 0)
        enc = lambda: eng.dev_vandermonde_apply(x.data_ptr(), G, n, d, y.data_ptr(), s)
        dec = lambda: eng.dev_batch_recover(list(range(n)), y.data_ptr(), G, n, d, t, co.data_ptr(), 0, st.data_ptr(), summ.data_ptr(), s)
        dn = lambda: eng.dev_batch_recover(list(range(d + t + 1)), y.data_ptr(), G, n, d, t, co.data_ptr(), 0, st.data_ptr(), summ.data_ptr(), s)
        p0 = lambda: eng.dev_batch_recover(list(range(n)), y.data_ptr(), G, n, d, t, co.data_ptr(), 0, st.data_ptr(), summ.data_ptr(), s, p0=True)
        assert enc() == 0 and dec() == 0
        torch.cuda.synchronize(); assert bool((co == x).all())
        out.append((ev(enc), ev(dec), ev(dn), ev(p0)))
    a, b = out
    print(f"gl n={n} t={t} d={d} 2^20: encode mfma {a[0]:.1f} us (lane {b[0]:.1f}) | decode {a[1]:.1f} ({b[1]:.1f}) | decode, exactly d+t+1 senders {a[2]:.1f} ({b[2]:.1f}) | P(0) {a[3]:.1f} ({b[3]:.1f})", flush=True)
PY
