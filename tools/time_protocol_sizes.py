"""run_preprocessing's triple part at the batch sizes the reference's node uses (honeybadger/mod.rs:106-120, :1434: 4 096 triple groups
per batch, i.e. N = 4 096 (2t + 1) triples per party) for the party counts of its tests and benches: device time per run, eager and as a
HIP graph, of Preprocessing (RanSha -> a, b; DouSha + RanDouSha -> r; TripleGen) and of TripleGen alone.
    python tools/time_protocol_sizes.py [groups] [n,t ...]  (under rocprofv3 --kernel-trace --stats for the launches behind the times)
    FIELD=goldilocks python tools/time_protocol_sizes.py    the reference's small field (PreprocNodesSmallField, honeybadger/mod.rs:316-324)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package

pkg = load_package()
pl = pkg.pipelines
GL = os.environ.get("FIELD", "fr") == "goldilocks"
eng = pkg.Engine(0, field="goldilocks") if GL else pkg.Engine(0)
EB = 8 if GL else 32
dev = torch.device("cuda", 0)
ts = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(ts)
stream = ts.cuda_stream
groups = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
shapes = [tuple(map(int, a.split(","))) for a in sys.argv[2:]] or [(4, 1), (7, 2), (10, 3), (16, 5)]


def rand_fr(*shape):
    if GL:  # canonical: below 2^63 < p
        return torch.randint(0, 1 << 62, shape, dtype=torch.int64, device=dev)
    x = torch.randint(0, 1 << 62, shape + (4,), dtype=torch.int64, device=dev)
    x[..., 3] &= (1 << 60) - 1
    return x


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


for n, t in shapes:
    N = groups * (2 * t + 1)
    pre = pl.Preprocessing(eng, n, t, N, stream)
    sec0 = {}
    for ptr, K, deg in ((pre.rs.coeffs, pre.K_rs, t), (pre.rd.coeffs_t, pre.K_rd, t), (pre.rd.coeffs_2t, pre.K_rd, 2 * t)):
        for p in range(n):
            co = rand_fr(K, deg + 1)
            if ptr == pre.rd.coeffs_2t:
                co[:, 0] = sec0[p]
            elif ptr == pre.rd.coeffs_t:
                sec0[p] = co[:, 0].clone()
            eng.d2d(ptr + p * K * (deg + 1) * EB, co.data_ptr(), K * (deg + 1) * EB, stream)
            torch.cuda.synchronize()
    pre.run(check=True)
    ms_e = ev_ms(lambda: pre.run(check=False))
    ms_rs = ev_ms(lambda: pre.rs.run(check=False))
    ms_rd = ev_ms(lambda: pre.rd.run(check=False))
    ms_tg = ev_ms(lambda: pre.tg.run(check=False))
    pre.capture()
    ms_g = ev_ms(pre.replay)
    print(f"n={n:2d} t={t} N={N:6d} K_rs={pre.K_rs:6d} K_rd={pre.K_rd:6d}: preprocessing {ms_e:.3f} ms eager, {ms_g:.3f} ms graph "
          f"({N / ms_g * 1e3:.3e} triples/s); RanSha {ms_rs:.3f}  RanDouSha {ms_rd:.3f}  TripleGen {ms_tg:.3f}", flush=True)
    pre.close()


# the producers alone at the column counts the reference's node hard-codes per run: RanSha 2 048 (honeybadger/mod.rs:1434), RanDouSha 1 536 (:114-120)
for n, t in shapes:
    rs = pl.RanSha(eng, n, t, 2048, stream)
    rd = pl.RanDouSha(eng, n, t, 1536, stream)
    for p in range(n):
        co = rand_fr(2048, t + 1)
        eng.d2d(rs.coeffs + p * 2048 * (t + 1) * EB, co.data_ptr(), 2048 * (t + 1) * EB, stream)
        ct, c2t = rand_fr(1536, t + 1), rand_fr(1536, 2 * t + 1)
        c2t[:, 0] = ct[:, 0]
        eng.d2d(rd.coeffs_t + p * 1536 * (t + 1) * EB, ct.data_ptr(), 1536 * (t + 1) * EB, stream)
        eng.d2d(rd.coeffs_2t + p * 1536 * (2 * t + 1) * EB, c2t.data_ptr(), 1536 * (2 * t + 1) * EB, stream)
        torch.cuda.synchronize()
    rs.run(check=True)
    rd.run(check=True)
    ms_rs, ms_rd = ev_ms(lambda: rs.run(check=False)), ev_ms(lambda: rd.run(check=False))
    rs.capture(), rd.capture()
    print(f"n={n:2d} t={t}: RanSha 2048 columns {ms_rs:.3f} ms eager, {ev_ms(rs.replay):.3f} graph ({(n - 2 * t) * 2048 / ms_rs * 1e3:.3e} sharings/s per party); "
          f"RanDouSha 1536 columns {ms_rd:.3f} ms eager, {ev_ms(rd.replay):.3f} graph", flush=True)
    rs.close(), rd.close()
