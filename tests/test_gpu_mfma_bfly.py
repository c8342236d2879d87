"""The matrix-core encode with the domain points taken in pairs (k, k + size/2), csrc/kernels_mfma_bfly.hpp: against the
oracle, against the kernel with one table row per point (hbmpc_set_matrix_cores(ctx, 3, ..)) and against the FFT kernels.
Covers every d + 1 it instantiates, points without a partner (n below the domain size), the unrolled pair loops (roles of
4, 8 and 16 pairs) and the run-time one, several roles, ragged tails, few workgroups (a wave walks many tiles through both
input register sets), strided output rows, party-batched calls and BASELINE configs[1] at full size.  Bar: bit-exact."""
import numpy as np
import pytest

from __graft_entry__ import load_package
from oracle import cref as O

pytestmark = pytest.mark.gpu

O_R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


@pytest.fixture(scope="module")
def eng():
    e = load_package().Engine(0)
    e.set_small_batch_chunks(0)
    yield e
    e.close()


def rnd(seed, *shape):
    return O.fill_random(seed, int(np.prod(shape))).reshape(*shape, 4)


def polys(seed, G, d):
    x = rnd(seed, G, d + 1)
    x[0] = 0
    x[1] = O.ints_to_u256([O_R - 1] * (d + 1))
    x[2, :, :] = 0
    x[2, d, 0] = 1
    x[3] = O.ints_to_u256([(O_R - 1) if i & 1 else 0 for i in range(d + 1)])   # the odd half alone at its maximum
    x[4] = O.ints_to_u256([0 if i & 1 else (O_R - 1) for i in range(d + 1)])   # the even half alone
    return x


def three_ways(eng, x, n, d):
    out = []
    for mode in (1, 3, 0):
        eng.set_matrix_cores(mode, 1)
        rc, y = eng.vandermonde_apply(x, n, d)
        assert rc == 0
        out.append(y)
    eng.set_matrix_cores(1, 65536)
    return out


# (n, d): pairs per role / roles.  size <= 16 takes this kernel beyond the workgroup-per-tile range only (> 512 tiles)
SHAPES = [(5, 1), (6, 2), (7, 2), (8, 3), (9, 1), (10, 3), (11, 4), (12, 5), (13, 4), (14, 6), (15, 7), (16, 5), (16, 8), (16, 10), (16, 14),
          (16, 15), (20, 15), (31, 15), (64, 15), (17, 1), (20, 6), (24, 8), (31, 10), (31, 13), (31, 14), (32, 9), (33, 5), (40, 3), (40, 13), (63, 12), (64, 14), (64, 2), (100, 3)]


@pytest.mark.parametrize("n,d", SHAPES)
def test_point_pairs_vs_oracle_and_the_other_kernels(eng, n, d):
    G = 16384 + 32 * 7 + 5   # 520 tiles: beyond the workgroup-per-tile kernel, last tile ragged
    x = polys(1000 + 37 * n + d, G, d)
    y_pairs, y_rows, y_fft = three_ways(eng, x, n, d)
    rc0, y0 = O.vandermonde_apply(x, n, d)
    assert rc0 == 0
    assert np.array_equal(y_pairs, y0) and np.array_equal(y_rows, y0) and np.array_equal(y_fft, y0)


@pytest.mark.parametrize("n,d", [(16, 5), (31, 10), (13, 4), (20, 6), (40, 3)])
def test_few_workgroups_walk_many_tiles(eng, n, d):
    """8 (16 with two roles) workgroups: every wave runs the tile loop many times, through both input register sets and the
    counted waits between them"""
    G = 40000 + 17
    x = polys(7 * n + d, G, d)
    eng.set_matrix_core_workgroups(8)
    try:
        y_pairs, y_rows, y_fft = three_ways(eng, x, n, d)
    finally:
        eng.set_matrix_core_workgroups(0)
    rc0, y0 = O.vandermonde_apply(x, n, d)
    assert rc0 == 0 and np.array_equal(y_pairs, y0) and np.array_equal(y_rows, y0) and np.array_equal(y_fft, y0)


def test_strided_rows_and_party_batches(eng):
    """the in-place wire path (output rows G + 8 apart: nothing written between them) and x[P][G][d+1] -> y[P][n][G]"""
    import torch
    dev = torch.device("cuda", 0)
    for (n, d, G, P) in ((16, 5, 20000 + 9, 3), (31, 10, 17000 + 1, 2), (7, 2, 30000, 2)):
        x = np.stack([polys(90 + p + n, G, d) for p in range(P)])
        xd = torch.as_tensor(x.view(np.int64), device=dev).contiguous()
        ystr = torch.full((n, G + 8, 4), -1, dtype=torch.int64, device=dev)
        yp = torch.full((P, n, G, 4), -1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        eng.set_matrix_cores(1, 1)
        assert eng.dev_vandermonde_apply_strided(xd[0].data_ptr(), G, n, d, ystr.data_ptr(), G + 8, 0) == 0
        assert eng.dev_vandermonde_apply_parties(xd.data_ptr(), G, n, d, P, yp.data_ptr(), 0) == 0
        eng.sync()
        eng.set_matrix_cores(1, 65536)
        for p in range(P):
            rc0, y0 = O.vandermonde_apply(x[p], n, d)
            assert rc0 == 0 and np.array_equal(yp[p].cpu().numpy().view(np.uint64), y0), (n, d, p)
            if p == 0:
                assert np.array_equal(ystr[:, :G].cpu().numpy().view(np.uint64), y0) and bool((ystr[:, G:] == -1).all())


def test_full_size_config2(eng):
    """BASELINE configs[1]: x[2^20][6] -> y[16][2^20], the point pairs against the FFT kernel (whole array) and against the oracle
    (sampled chunks: both ends, tile boundaries, the wrap of a wave's tile walk)"""
    import torch
    n, d, G = 16, 5, 1 << 20
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0xC0FFEE12)
    lo = torch.randint(0, 1 << 62, (G, d + 1, 3), dtype=torch.int64, device=dev, generator=gen)
    hi = torch.randint(0, 0x73EDA753299D7D48, (G, d + 1, 1), dtype=torch.int64, device=dev, generator=gen)
    x = torch.cat([lo, hi], dim=-1).contiguous()
    ys = {}
    for mode in (1, 0):
        eng.set_matrix_cores(mode, 65536)
        y = torch.full((n, G, 4), -1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        assert eng.dev_vandermonde_apply(x.data_ptr(), G, n, d, y.data_ptr(), 0) == 0
        eng.sync()
        ys[mode] = y
    eng.set_matrix_cores(1, 65536)
    assert torch.equal(ys[1], ys[0])
    idx = np.unique(np.concatenate([np.arange(0, 70), np.arange(G - 70, G), np.arange(31, G, 32)[:40], np.arange(32 * 3072 - 3, 32 * 3072 + 3)]))
    ti = torch.as_tensor(idx, device=dev)
    rc0, y0 = O.vandermonde_apply(np.ascontiguousarray(x[ti].cpu().numpy().view(np.uint64)), n, d)
    assert rc0 == 0 and np.array_equal(ys[1][:, ti].cpu().numpy().view(np.uint64), y0)
    # linearity, a size-independent property: V(x + x') = V(x) + V(x') over the whole array
    x2 = torch.roll(x, 1, 0).contiguous()
    s_in = torch.empty_like(x)
    s_out = torch.empty_like(ys[1])
    y2 = torch.empty_like(ys[1])
    assert eng.dev_fr_op("add", x.data_ptr(), x2.data_ptr(), G * (d + 1), s_in.data_ptr()) == 0
    assert eng.dev_vandermonde_apply(x2.data_ptr(), G, n, d, y2.data_ptr(), 0) == 0
    assert eng.dev_fr_op("add", ys[1].data_ptr(), y2.data_ptr(), n * G, s_out.data_ptr()) == 0
    ysum = torch.empty_like(ys[1])
    assert eng.dev_vandermonde_apply(s_in.data_ptr(), G, n, d, ysum.data_ptr(), 0) == 0
    eng.sync()
    assert torch.equal(ysum, s_out)


@pytest.mark.parametrize("n,t,G,parties", [(16, 5, 9000 + 7, 16), (16, 5, 140000, 1), (13, 4, 50000 + 3, 3), (16, 2, 70000, 2), (16, 7, 33000 + 1, 4),
                                           (20, 3, 44000 + 5, 3), (24, 4, 66000, 2), (9, 1, 131072, 1),
                                           # from 2^14 chunks over all parties (the reference node's batches: 4 096 chunks x n parties)
                                           (16, 5, 1100, 16), (16, 5, 4096 + 3, 16), (13, 4, 6000 + 1, 3), (10, 3, 20000, 1), (16, 5, 1024, 16), (16, 5, 1000, 16),
                                           (20, 3, 5000 + 9, 4)])
def test_triple_generation_encode_with_the_products_inside(eng, n, t, G, parties):
    """hbmpc_dev_triple_encode_parties on large batches: the local products a b - r2t are computed inside the matrix-core encode
    (k_mfma_bfly<.., TRIPLE>).  Against hbmpc_dev_triple_local + hbmpc_dev_vandermonde_apply_parties with the matrix cores off
    (the whole array), and against the oracle on sampled chunks of the first and the last party; edge operands included"""
    import torch
    d = 2 * t
    N = G * (d + 1)
    dev = torch.device("cuda", 0)
    ah, bh, rh = (O.fill_random(160 + k + n, parties * N) for k in range(3))
    ah[:d + 1] = 0
    bh[d + 1:2 * (d + 1)] = O.ints_to_u256([O_R - 1] * (d + 1))
    ah[d + 1:2 * (d + 1)] = O.ints_to_u256([O_R - 1] * (d + 1))
    rh[2 * (d + 1):3 * (d + 1)] = O.ints_to_u256([O_R - 1] * (d + 1))
    a, b, r = (torch.from_numpy(v.view(np.int64)).to(dev) for v in (ah, bh, rh))
    tmp = torch.empty((parties * N, 4), dtype=torch.int64, device=dev)
    y1 = torch.full((parties, n, G, 4), -1, dtype=torch.int64, device=dev)
    y2 = torch.full((parties, n, G, 4), -1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    eng.set_matrix_cores(1, 65536)
    assert eng.dev_triple_encode_parties(a.data_ptr(), b.data_ptr(), r.data_ptr(), G, n, d, parties, 0, y1.data_ptr()) == 0, eng.last_error()
    eng.set_matrix_cores(0)
    try:
        assert eng.dev_elem("triple_local", [a.data_ptr(), b.data_ptr(), r.data_ptr(), tmp.data_ptr()], parties * N) == 0
        assert eng.dev_vandermonde_apply_parties(tmp.data_ptr(), G, n, d, parties, y2.data_ptr()) == 0
        eng.sync()
    finally:
        eng.set_matrix_cores(1, 65536)
    assert torch.equal(y1, y2)
    idx = np.unique(np.concatenate([np.arange(0, 40), np.arange(G - 40, G), np.random.default_rng(n).integers(0, G, 60)]))
    for p in (0, parties - 1):
        sel = (p * G + idx)[:, None] * (d + 1) + np.arange(d + 1)[None, :]
        x = O.triple_local(ah[sel.ravel()], bh[sel.ravel()], rh[sel.ravel()])[1].reshape(len(idx), d + 1, 4)
        rc, want = O.vandermonde_apply(x, n, d)
        assert rc == 0 and np.array_equal(y1[p][:, torch.as_tensor(idx, device=dev)].cpu().numpy().view(np.uint64), want), p


@pytest.mark.parametrize("n,d,G", [(16, 15, 20000 + 3), (16, 5, 40000), (31, 10, 17000 + 9), (13, 12, 30000 + 1), (16, 15, 700), (5, 2, 50000),
                                   (40, 20, 20000)])
def test_inputs_given_as_rows(eng, n, d, G):
    """hbmpc_dev_vandermonde_apply_rows: x[d + 1][stride] instead of x[G][d + 1] -- read in place by the point-pair kernel where it
    covers the shape (large batches, d + 1 <= 16), through the workspace otherwise (small batch, 4-point... domain, d + 1 > 16);
    without a workspace those shapes are refused"""
    import torch
    dev = torch.device("cuda", 0)
    x = polys(31 * n + d, G, d)
    stride = G + 24
    xr = torch.full((d + 1, stride, 4), -1, dtype=torch.int64, device=dev)
    xr[:, :G] = torch.as_tensor(np.ascontiguousarray(x.transpose(1, 0, 2)).view(np.int64), device=dev)
    tmp = torch.empty((G, d + 1, 4), dtype=torch.int64, device=dev)
    y = torch.full((n, G, 4), -1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    assert eng.dev_vandermonde_apply_rows(xr.data_ptr(), stride, G, n, d, tmp.data_ptr(), y.data_ptr()) == 0, eng.last_error()
    eng.sync()
    rc0, y0 = O.vandermonde_apply(x, n, d)
    assert rc0 == 0 and np.array_equal(y.cpu().numpy().view(np.uint64), y0)
    direct = d + 1 <= 16 and n > 4 and (G + 31) // 32 > 512
    y.fill_(-1)
    rc = eng.dev_vandermonde_apply_rows(xr.data_ptr(), stride, G, n, d, 0, y.data_ptr())
    eng.sync()
    if direct:
        assert rc == 0 and np.array_equal(y.cpu().numpy().view(np.uint64), y0)
    else:
        assert rc != 0 and "workspace" in eng.last_error()


@pytest.mark.parametrize("n,d,G,parties", [(16, 5, 1100, 16), (16, 10, 1536 + 3, 16), (7, 2, 2048 + 1, 7), (10, 3, 4000, 10), (13, 8, 7000 + 5, 3), (16, 5, 15019, 16),
                                           (5, 1, 9000, 5), (16, 5, 20000 + 1, 2), (16, 5, 130, 16), (16, 11, 3000, 16), (20, 6, 3000, 8), (4, 1, 6000, 4)])
def test_party_batched_encode_is_one_launch_with_the_same_bytes(eng, n, d, G, parties):
    """hbmpc_dev_vandermonde_apply_parties, x[P][G][d + 1] -> y[P][n][G] (the dealers' compute_shares of the producers): where the point-pair
    kernel covers the shape all parties' chunks are ONE launch (k_mfma_bfly<.., LISTS> with every row party-major) -- against the lane
    kernels (matrix cores off: the whole array) and the oracle on the first and the last party; shapes outside it (d + 1 > 11, 4- and 32-point
    domains, few chunks) take the other paths and must agree as well"""
    import torch
    dev = torch.device("cuda", 0)
    x = polys(5 * n + d + G, parties * G, d)
    xd = torch.as_tensor(x.view(np.int64), device=dev)
    y1 = torch.full((parties, n, G, 4), -1, dtype=torch.int64, device=dev)
    y2 = torch.full((parties, n, G, 4), -1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    assert eng.dev_vandermonde_apply_parties(xd.data_ptr(), G, n, d, parties, y1.data_ptr()) == 0, eng.last_error()
    eng.set_matrix_cores(0)
    try:
        assert eng.dev_vandermonde_apply_parties(xd.data_ptr(), G, n, d, parties, y2.data_ptr()) == 0, eng.last_error()
        eng.sync()
    finally:
        eng.set_matrix_cores(1, 65536)
    assert torch.equal(y1, y2)
    for p in (0, parties - 1):
        rc, want = O.vandermonde_apply(x[p * G:(p + 1) * G], n, d)
        assert rc == 0 and np.array_equal(y1[p].cpu().numpy().view(np.uint64), want), p


@pytest.mark.parametrize("n,K,row0,rows", [(16, 1500, 10, 6), (16, 1237, 0, 6), (7, 3001, 4, 3), (13, 2000, 0, 5), (4, 5000, 2, 2), (16, 40, 10, 6),
                                           (16, 1500, 0, 15), (16, 1500, 15, 1), (7, 367, 4, 3), (4, 33, 0, 2), (3, 700, 1, 2), (8, 100, 0, 3), (5, 64, 2, 3)])
def test_mixing_step_with_lists_and_party_major_other_rows(eng, n, K, row0, rows):
    """hbmpc_dev_vandermonde_apply_rows_split: the n x n mixing step of the producers over G = n K chunks (party j, batch element k) with
    output rows [row0, row0 + rows) written as the parties' lists (two slices) and every other row party-major -- from the kernel that
    computes them (k_mfma_bfly<.., LISTS>: 5 .. 16 rows, large batches) and through the copies (every other shape;
    hbmpc_set_producer_fusion(0)): both against the plain y[row][G] of hbmpc_dev_vandermonde_apply_rows, itself checked against the oracle"""
    import ctypes as C
    import torch
    dev = torch.device("cuda", 0)
    G, d = n * K, n - 1
    x = polys(17 * n + K, G, d)
    xr = torch.as_tensor(np.ascontiguousarray(x.transpose(1, 0, 2)).view(np.int64), device=dev)
    tmp = torch.empty((G, d + 1, 4), dtype=torch.int64, device=dev)
    y = torch.full((n, G, 4), -1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    assert eng.dev_vandermonde_apply_rows(xr.data_ptr(), G, G, n, d, tmp.data_ptr(), y.data_ptr()) == 0, eng.last_error()
    eng.sync()
    rc0, y0 = O.vandermonde_apply(x, n, d)
    y_np = y.cpu().numpy().view(np.uint64)
    assert rc0 == 0 and np.array_equal(y_np, y0)
    nother = n - rows
    other_rows = [r for r in range(n) if not row0 <= r < row0 + rows]
    want_others = y_np.reshape(n, n, K, 4)[other_rows].transpose(1, 0, 2, 3)            # [party][r'][K]
    want_lists = y_np.reshape(n, n, K, 4)[row0:row0 + rows].transpose(1, 2, 0, 3)         # [party][K][row]
    k1 = K // 3                                                                           # two slices: [0, k1) and [k1 + 1, K): element k1 is dropped
    in_kernel = eng.apply_rows_lists_in_kernel(G, n, d)
    assert in_kernel == (3 <= n <= 8 or (n <= 16 and (G + 31) // 32 > 512))   # up to 8 parties the lane kernel k_eval_fft1_mix serves what the matrix cores do not
    for fused in (1, 0):
        assert eng.L.hbmpc_set_producer_fusion(eng.ctx, C.c_int(fused)) == 0
        assert eng.apply_rows_lists_in_kernel(G, n, d) == (in_kernel and fused == 1)
        la = torch.full((n, k1, rows, 4), -1, dtype=torch.int64, device=dev)
        lb = torch.full((n, K - k1 - 1 + 2, rows, 4), -1, dtype=torch.int64, device=dev)  # party stride two elements of `rows` wider than the slice
        others = torch.full((n, nother, K, 4), -1, dtype=torch.int64, device=dev)
        y.fill_(-1)
        torch.cuda.synchronize()
        slices = [(la.data_ptr(), k1 * rows, 0, k1), (lb.data_ptr(), (K - k1 + 1) * rows, k1 + 1, K - k1 - 1)]
        rc = eng.dev_vandermonde_apply_rows_split(xr.data_ptr(), G, G, n, d, tmp.data_ptr(), y.data_ptr(), row0, rows, K, slices, others.data_ptr())
        assert rc == 0, eng.last_error()
        eng.sync()
        assert np.array_equal(others.cpu().numpy().view(np.uint64), want_others), fused
        assert np.array_equal(la.cpu().numpy().view(np.uint64), want_lists[:, :k1]), fused
        got_b = lb.cpu().numpy().view(np.uint64)
        assert np.array_equal(got_b[:, :K - k1 - 1], want_lists[:, k1 + 1:]) and np.all(got_b[:, K - k1 - 1:] == np.uint64(2**64 - 1)), fused
    assert eng.L.hbmpc_set_producer_fusion(eng.ctx, C.c_int(1)) == 0
    # no buffer for the other rows, every row a list row: refused
    rc = eng._f("dev_vandermonde_apply_rows_split")(eng.ctx, C.c_void_p(xr.data_ptr()), C.c_size_t(G), C.c_size_t(G), C.c_size_t(n), C.c_size_t(d),
                                                    C.c_void_p(tmp.data_ptr()), C.c_void_p(y.data_ptr()), C.c_size_t(row0), C.c_size_t(rows), C.c_size_t(K),
                                                    None, C.c_size_t(0), C.c_void_p(0), C.c_void_p(0))
    assert rc != 0


@pytest.mark.parametrize("n,G,stride_pad", [(16, 20000 + 5, 0), (16, 40000, 64), (8, 33000 + 1, 0)])
def test_full_domain_interpolation_is_the_inverse_transform(eng, n, G, stride_pad):
    """hbmpc_dev_batch_interpolate through all n shares of a full domain (the RanDouSha verifier, ran_dou_sha/mod.rs:569-602) on the
    point-pair kernel with the inverse DFT rows: against the lane kernels (matrix cores off: the whole array) and against the
    coefficients the shares were made from; arrival order shuffled, sender rows strided, degrees included"""
    import torch
    dev = torch.device("cuda", 0)
    t = (n - 1) // 3
    rng = np.random.default_rng(n + G)
    ids = [int(i) for i in rng.permutation(n)]
    stride = G + stride_pad
    for true_deg in (t, 2 * t, n - 1):
        co = polys(700 + true_deg + n, G, true_deg)
        co[5, true_deg] = 0                           # a lower-degree column
        rc, sh = O.compute_shares(co, n, true_deg)
        assert rc == 0
        ev = torch.full((n, stride, 4), -1, dtype=torch.int64, device=dev)
        ev[:, :G] = torch.as_tensor(np.ascontiguousarray(sh[ids]).view(np.int64), device=dev)
        outs = []
        for mode in (1, 0):
            eng.set_matrix_cores(mode)
            c = torch.full((G, n, 4), -1, dtype=torch.int64, device=dev)
            dg = torch.full((G,), -1, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            assert eng.dev_batch_interpolate(ids, ev.data_ptr(), stride, G, n, c.data_ptr(), dg.data_ptr()) == 0, eng.last_error()
            eng.sync()
            outs.append((c.cpu().numpy().view(np.uint64), dg.cpu().numpy()))
        eng.set_matrix_cores(1)
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
        got, deg = outs[0]
        assert np.array_equal(got[:, :true_deg + 1], co) and not got[:, true_deg + 1:].any()
        assert deg[6] == true_deg and deg[0] == 0 and deg[5] < true_deg
