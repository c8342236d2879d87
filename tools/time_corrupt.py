"""cfg3 decode with t lowest-id senders corrupted in a fraction of the chunks: time of the flag + OEC/Gao path"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from __graft_entry__ import load_package
from oracle import cref
eng = load_package().Engine(0)
dev = torch.device("cuda", 0)
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts); s = ts.cuda_stream
n, t, d, G = 31, 10, 10, 1 << 20
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
x = torch.from_numpy(cref.fill_random(2, G * (d + 1)).reshape(G, d + 1, 4).view(np.int64)).to(dev)
y = torch.empty((n, G, 4), dtype=torch.int64, device=dev)
co = torch.empty((G, d + 1, 4), dtype=torch.int64, device=dev)
st = torch.empty((G,), dtype=torch.uint8, device=dev); summ = torch.zeros((4,), dtype=torch.int32, device=dev)
eng.dev_vandermonde_apply(x.data_ptr(), G, n, d, y.data_ptr(), s)
bad = torch.randperm(G, device=dev)[: int(G * frac)]
nbad = int(sys.argv[2]) if len(sys.argv) > 2 else t
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # id of the first corrupted sender
if len(sys.argv) > 4: eng.set_second_chance(sys.argv[4] != "gao")
y[first:first + nbad, bad, 0] ^= 1
ids = list(range(n))
for _ in range(2): eng.dev_batch_recover(ids, y.data_ptr(), G, n, d, t, co.data_ptr(), 0, st.data_ptr(), summ.data_ptr(), s)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): eng.dev_batch_recover(ids, y.data_ptr(), G, n, d, t, co.data_ptr(), 0, st.data_ptr(), summ.data_ptr(), s)
e1.record(); torch.cuda.synchronize()
ok = bool((co == x).all())
print(f"frac={frac} corrupted_senders={nbad} from id {first} ({'OEC/Gao only' if len(sys.argv) > 4 and sys.argv[4] == 'gao' else 'second chance first'}): {e0.elapsed_time(e1)/3:.2f} ms, fallback={summ.tolist()[:2]}, correct={ok}")
