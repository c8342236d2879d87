import sys, time
sys.path.insert(0, "/root/repo")
import torch
import bench
from __graft_entry__ import load_package
pkg = load_package(); eng = pkg.Engine(0)
dev = torch.device("cuda", 0)
ts = torch.cuda.Stream(device=dev); torch.cuda.synchronize(); torch.cuda.set_stream(ts); st = ts.cuda_stream
n, t, N = 16, 5, 1100
pre = pkg.pipelines.Preprocessing(eng, n, t, N, st)
sec0 = {}
for ptr, K, deg in ((pre.rs.coeffs, pre.K_rs, t), (pre.rd.coeffs_t, pre.K_rd, t), (pre.rd.coeffs_2t, pre.K_rd, 2 * t)):
    for p in range(n):
        co = bench._rand_fr(torch, dev, K, deg + 1)
        if ptr == pre.rd.coeffs_2t: co[:, 0] = sec0[p]
        elif ptr == pre.rd.coeffs_t: sec0[p] = co[:, 0].clone()
        eng.d2d(ptr + p * K * (deg + 1) * 32, co.data_ptr(), K * (deg + 1) * 32, st)
        torch.cuda.synchronize()
pre.run(check=True)
torch.cuda.synchronize()
for _ in range(100): pre.run(check=False)
torch.cuda.synchronize()
print("K_rs", pre.K_rs, "K_rd", pre.K_rd)
