"""Per-kernel times of the config-4 triple_gen replay (rocprofv3 --kernel-trace --stats -- python3 tools/prof_cfg4.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from __graft_entry__ import load_package
pkg = load_package()
eng = pkg.Engine(0)
dev = torch.device("cuda:0")
st = torch.cuda.Stream(); torch.cuda.set_stream(st); stream = st.cuda_stream
n, t = 16, 5
N = ((1 << 22) // (2 * t + 1)) * (2 * t + 1)
tg = pkg.pipelines.TripleGen(eng, n, t, N, stream)
a, b, r = (bench._rand_fr(torch, dev, N) for _ in range(3))
bench._share_on_device(eng, torch, dev, stream, a, n, t, tg.a)
bench._share_on_device(eng, torch, dev, stream, b, n, t, tg.b)
bench._share_on_device(eng, torch, dev, stream, r, n, t, tg.rt)
bench._share_on_device(eng, torch, dev, stream, r, n, 2 * t, tg.r2t)
for _ in range(6):
    tg.run(check=False)
torch.cuda.synchronize()
