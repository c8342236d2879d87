"""BASELINE configs[3] and [4] at FULL size through the device pipelines (all 16 simulated parties on one GPU):
triple_gen over 2^22 triples and fpmul over 2^18 elements with (k, f) = (16, 4) and (16, 16).  Inputs are drawn and
shared on the device (the library's own compute_shares, as bench.py does); >= 2000 sampled elements per party -- first,
last, tile and chunk boundaries, random -- are compared with the oracle's element-wise restatement of every step
(triple_generation.rs:333-340,196-208; multiplication.rs:417-426,57-100; truncpr.rs:277-297,215-220), and the sampled
results are opened."""
import numpy as np
import pytest

import bench
from __graft_entry__ import load_package
from oracle import cref as O

pytestmark = pytest.mark.gpu


def sample_indices(N, seed, unit=1):
    rng = np.random.default_rng(seed)
    edges = [0, 1, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 65535, 65536, N // 2 - 1, N // 2, N - 65, N - 2, N - 1]
    edges += [unit * k + o for k in (1, 2, 1000, N // unit - 1) for o in (-1, 0, 1)]
    idx = np.unique(np.concatenate([np.arange(0, 48), np.arange(N - 48, N), np.array([e for e in edges if 0 <= e < N]),
                                    rng.integers(0, N, 2200)]))
    return idx


def rows_at(eng, ptr, n, N, idx, stream, inner=1):
    """[n][inner][len(idx)] elements of a device array [n][inner][N] (downloaded whole, then sampled)"""
    host = np.zeros((n, inner, N, 4), dtype=np.uint64)
    eng.d2h(host, ptr, stream)
    eng.sync(stream)
    return np.ascontiguousarray(host[:, :, idx])


def open_p0(sh, n, d, t):
    rc, p0, st = O.batch_recover_p0(list(range(n)), np.ascontiguousarray(sh), n, d, t)
    assert rc == 0 and not st.any()
    return p0


def test_triple_gen_full_size():
    import torch
    pkg = load_package()
    eng = pkg.Engine(0)
    dev = torch.device("cuda", 0)
    ts = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(ts)
    s = ts.cuda_stream
    n, t = 16, 5
    m = 2 * t + 1
    N = ((1 << 22) // m) * m
    torch.manual_seed(0xC0FFEE03)
    tg = pkg.pipelines.TripleGen(eng, n, t, N, s)
    a, b, r = (bench._rand_fr(torch, dev, N) for _ in range(3))
    bench._share_on_device(eng, torch, dev, s, a, n, t, tg.a)
    bench._share_on_device(eng, torch, dev, s, b, n, t, tg.b)
    bench._share_on_device(eng, torch, dev, s, r, n, t, tg.rt)
    bench._share_on_device(eng, torch, dev, s, r, n, 2 * t, tg.r2t)
    tg.run(check=True)
    idx = sample_indices(N, 4, unit=m)
    assert len(idx) >= 2000
    ti = torch.as_tensor(idx, device=dev)
    ah, bh, rh = (v[ti].cpu().numpy().view(np.uint64) for v in (a, b, r))
    c = rows_at(eng, tg.c, n, N, idx, s)[:, 0]
    rt = rows_at(eng, tg.rt, n, N, idx, s)[:, 0]
    tg.close()
    eng.close()
    ab = O.fr_binop("mul", ah, bh)
    opened = O.fr_binop("sub", ab, rh)                       # what BatchRecon opens: a b - r
    for p in range(n):
        assert np.array_equal(c[p], O.triple_finalize(np.ascontiguousarray(rt[p]), opened)[1]), p
    assert np.array_equal(open_p0(c, n, t, t), ab)           # [c]_t opens to a b


@pytest.mark.parametrize("k,f", [(16, 4), (16, 16)])
def test_fpmul_full_size(k, f):
    import torch
    pkg = load_package()
    eng = pkg.Engine(0)
    dev = torch.device("cuda", 0)
    ts = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(ts)
    s = ts.cuda_stream
    n, t, N = 16, 5, 1 << 18
    torch.manual_seed(0xC0FFEE05 + f)
    fp = bench.setup_fpmul(eng, torch, dev, s, n, t, N, k, f)
    fp.run(check=True)
    idx = sample_indices(N, 5 + f)
    assert len(idx) >= 2000
    g = {name: rows_at(eng, getattr(fp, name), n, N, idx, s)[:, 0] for name in ("x", "y", "ta", "tb", "tc", "rint", "z", "out")}
    rbits = rows_at(eng, fp.rbits, n, N, idx, s, inner=f)    # [n][f][samples]
    fp.close()
    eng.close()
    # the oracle's restatement of every step, on the sampled columns, from the input shares alone
    dsh, esh = zip(*[O.beaver_open_shares(g["ta"][p], g["tb"][p], g["x"][p], g["y"][p])[1:] for p in range(n)])
    dop, eop = open_p0(np.stack(dsh), n, t, t), open_p0(np.stack(esh), n, t, t)
    z = np.stack([O.beaver_finalize(g["tc"][p], g["x"][p], g["y"][p], dop, eop)[1] for p in range(n)])
    assert np.array_equal(z, g["z"])
    rdash = np.stack([O.truncpr_rdash(np.ascontiguousarray(rbits[p]), f)[1] for p in range(n)])
    osh = np.stack([O.truncpr_open_share(z[p], rdash[p], g["rint"][p], k, f)[1] for p in range(n)])
    cop = open_p0(osh, n, t, t)
    out = np.stack([O.truncpr_finalize(z[p], rdash[p], cop, f)[1] for p in range(n)])
    assert np.array_equal(out, g["out"])
    # and the protocol's meaning: z opens to x y, out to floor(x y / 2^f) or that plus one
    xs, ys = O.u256_to_ints(open_p0(g["x"], n, t, t)), O.u256_to_ints(open_p0(g["y"], n, t, t))
    prod = [u * v for u, v in zip(xs, ys)]
    assert O.u256_to_ints(open_p0(z, n, t, t)) == prod
    if f < k:  # TruncPr's own precondition (m < k, truncpr.rs:185-200); (16, 16) is an arithmetic-parity case only
        for v, pr in zip(O.u256_to_ints(open_p0(out, n, t, t)), prod):
            assert v in (pr >> f, (pr >> f) + 1)
