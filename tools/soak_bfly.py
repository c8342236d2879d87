"""Soak of the point-pair matrix-core kernels (csrc/kernels_mfma_bfly.hpp) on random shapes: encode (chunk-major and from rows),
full-domain interpolation and the fused triple encode, each against the FFT / lane kernels over the WHOLE output and against the
oracle on sampled chunks.  python -u tools/soak_bfly.py FIRST LAST [MAX_SECONDS]   (run on the GPU box)"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from __graft_entry__ import load_package
from oracle import cref as O
first, last = int(sys.argv[1]), int(sys.argv[2])
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 600.0
eng = load_package().Engine(0)
eng.set_small_batch_chunks(0)
dev = torch.device("cuda", 0)
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
def T(a): return torch.as_tensor(np.ascontiguousarray(a).view(np.int64), device=dev)
def H(t): return t.cpu().numpy().view(np.uint64)
t0, done = time.time(), 0
for seed in range(first, last):
    rng = random.Random(seed)
    n = rng.choice([5, 6, 7, 8, 9, 12, 13, 15, 16, 16, 16, 17, 20, 24, 31, 32, 33, 40, 63, 64, 100, 128, 200, 255])
    d = rng.randint(1, min(15, n - 1))
    few = rng.random() < 0.7
    G = rng.randint(600, 3000) if few else rng.randint(16385, 30000)
    eng.set_matrix_core_workgroups(8 if few else 0)
    x = O.fill_random(seed, G * (d + 1)).reshape(G, d + 1, 4)
    for g in rng.sample(range(G), 6):
        x[g] = O.ints_to_u256([rng.choice([0, 1, R - 1, R - 2, (1 << 254) + 12345]) for _ in range(d + 1)])
    ys = []
    for mode in (1, 0):
        eng.set_matrix_cores(mode, 1)
        rc, y = eng.vandermonde_apply(x, n, d)
        assert rc == 0
        ys.append(y)
    assert np.array_equal(ys[0], ys[1]), ("encode", seed, n, d, G)
    idx = sorted(rng.sample(range(G), 12))
    rc0, y0 = O.vandermonde_apply(np.ascontiguousarray(x[idx]), n, d)
    assert rc0 == 0 and np.array_equal(ys[0][:, idx], y0), ("encode vs oracle", seed, n, d, G)
    # the same from rows
    eng.set_matrix_cores(1, 1)
    stride = G + rng.choice([0, 8, 40])
    xr = torch.zeros((d + 1, stride, 4), dtype=torch.int64, device=dev)
    xr[:, :G] = T(x.transpose(1, 0, 2))
    tmp = torch.empty((G, d + 1, 4), dtype=torch.int64, device=dev)
    yd = torch.zeros((n, G, 4), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()  # the library's stream does not wait for torch's (the fills above must have landed)
    assert eng.dev_vandermonde_apply_rows(xr.data_ptr(), stride, G, n, d, tmp.data_ptr(), yd.data_ptr()) == 0
    eng.sync()
    assert np.array_equal(H(yd), ys[1]), ("rows", seed, n, d, G)
    done += 2
    # full-domain interpolation
    if seed % 3 == 0:
        nn = rng.choice([8, 16])
        dd = rng.randint(0, nn - 1)
        co = O.fill_random(seed + 7, G * (dd + 1)).reshape(G, dd + 1, 4)
        co[rng.randrange(G)] = 0
        rc, sh = O.compute_shares(co, nn, dd)
        ids = list(range(nn)); rng.shuffle(ids)
        ev = T(sh[ids])
        res = []
        for mode in (1, 0):
            eng.set_matrix_cores(mode, 1)
            c = torch.zeros((G, nn, 4), dtype=torch.int64, device=dev); dg = torch.zeros((G,), dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            assert eng.dev_batch_interpolate(ids, ev.data_ptr(), G, G, nn, c.data_ptr(), dg.data_ptr()) == 0
            eng.sync(); res.append((H(c), dg.cpu().numpy()))
        assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]), ("interpolate", seed, nn, dd, G)
        assert np.array_equal(res[0][0][:, :dd + 1], co) and not res[0][0][:, dd + 1:].any()
        done += 1
    # the fused triple encode
    if seed % 4 == 0:
        nt = rng.choice([9, 12, 13, 16, 16, 20, 24, 31])
        tt = rng.randint(1, min(7, (nt - 1) // 3))
        d2 = 2 * tt
        if (1 << (nt - 1).bit_length()) // 2 * ((d2 + 1) * 1024 + 256) <= 160 * 1024:
            P = rng.choice([1, 2, 3])
            Gt = (131072 + P - 1) // P + rng.randint(0, 50)
            Nt = P * Gt * (d2 + 1)
            a, b, r = (O.fill_random(seed + 11 + k, Nt) for k in range(3))
            ad, bd, rd = T(a), T(b), T(r)
            outs = []
            for mode in (1, 0):
                eng.set_matrix_cores(mode, 65536)
                y = torch.zeros((P, nt, Gt, 4), dtype=torch.int64, device=dev)
                tw = torch.empty((Nt, 4), dtype=torch.int64, device=dev)
                torch.cuda.synchronize()
                assert eng.dev_triple_encode_parties(ad.data_ptr(), bd.data_ptr(), rd.data_ptr(), Gt, nt, d2, P, tw.data_ptr(), y.data_ptr()) == 0, eng.last_error()
                eng.sync(); outs.append(y)
            assert torch.equal(outs[0], outs[1]), ("triple", seed, nt, tt, Gt, P)
            done += 1
    if seed % 10 == 0:
        print(f"seed {seed}: {done} cases ok, {time.time() - t0:.0f} s", flush=True)
    if time.time() - t0 > budget:
        print(f"time budget reached at seed {seed}", flush=True)
        break
eng.set_matrix_core_workgroups(0)
print(f"soak ok: {done} cases, seeds {first}..{seed}", flush=True)
