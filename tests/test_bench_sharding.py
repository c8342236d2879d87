"""The N > 1 path of bench.py on CPU: world_size-2 gloo processes run the same shard / barrier /
max-over-ranks timing logic the GPU ranks run (the compute step is replaced by a stub: there is no
data-path collective to test, only the sharding and the reduction)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 1 << 20, (1 << 22) + 3):
        for world in (1, 2, 3, 4, 8):
            spans = [bench.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import time
    calls = []

    def step():
        calls.append(1)
        time.sleep(0.002 * (rank + 1))  # rank 1 is slower: the reported time must be ITS time

    def max_reduce(x):
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    secs = bench.timed_steps(step, steps=5, warmup=2, sync_fn=lambda: None, barrier_fn=dist.barrier,
                             max_reduce_fn=max_reduce)["secs"]
    lo, hi = bench.shard_range(1000, rank, world)
    tot = torch.tensor([hi - lo], dtype=torch.int64)
    dist.all_reduce(tot)
    q.put((rank, len(calls), secs, int(tot.item())))
    dist.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [7, 7]                 # W + K steps on every rank, exactly
    assert res[0][2] == res[1][2]                        # both ranks report the max-over-ranks time
    assert res[0][2] >= 5 * 0.004 * 0.9                  # ... which is the slow rank's
    assert res[0][3] == 1000                             # shards cover the batch exactly once


def _gather_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np
    from __graft_entry__ import load_package
    from oracle import cref  # stands in for the device compute in this CPU test
    load_package()
    from mpc_protocols_amd import sharding
    n, d, total = 7, 2, 101                                  # 101 over 2 ranks: ragged shards (51 + 50)
    coeffs = cref.fill_random(77, total * (d + 1)).reshape(total, d + 1, 4)
    lo, hi = sharding.shard_range(total, rank, world)
    rc, mine = cref.compute_shares(coeffs[lo:hi], n, d)      # this rank's slice of the batch: [n][hi - lo]
    assert rc == 0
    got = sharding.gather_party_major(torch.from_numpy(mine.view(np.int64)), total)
    rc, want = cref.compute_shares(coeffs, n, d)
    ok = bool(np.array_equal(got.numpy().view(np.uint64), want))
    # equal shards (100 over 2 ranks): the permuted view of the ONE receive buffer, no copy -- and the dense form made from it
    total2 = 100
    lo2, hi2 = sharding.shard_range(total2, rank, world)
    rc, mine2 = cref.compute_shares(coeffs[lo2:hi2], n, d)
    view, spans = sharding.gather_shards(torch.from_numpy(mine2.view(np.int64)), total2)
    rc, want2 = cref.compute_shares(coeffs[:total2], n, d)
    ok = ok and tuple(view.shape) == (n, world, 50, 4) and spans == [(0, 50), (50, 100)] and not view.is_contiguous()
    ok = ok and view._base is not None and view._base.numel() == world * n * 50 * 4        # a view of the one receive buffer
    for r in range(world):
        ok = ok and bool(np.array_equal(view[:, r].numpy().view(np.uint64), want2[:, spans[r][0]:spans[r][1]]))
    dense = sharding.gather_party_major(torch.from_numpy(mine2.view(np.int64)), total2)
    ok = ok and dense.is_contiguous() and bool(np.array_equal(dense.numpy().view(np.uint64), want2))
    q.put((rank, ok, tuple(got.shape)))
    dist.destroy_process_group()


def test_final_gather_world_size_2_gloo():
    """the path's only collective -- every rank ends up with party-major [n][total] rows equal to the
    single-device result, ragged shards included."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 17) % 1000
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(0, True, (7, 101, 4)), (1, True, (7, 101, 4))]


def _cfg4_worker(rank, world, port, q):
    """bench.py --workload cfg4 on CPU: the triple_gen batch sharded by chunks of 2t+1, every rank's result shares
    computed from its shard alone (the oracle stands in for the device pipeline: [c]_t = rt + (a b - r) per party,
    triple_generation.rs:196-208), then gather_ragged -- against the single-rank result of the whole batch."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np
    from __graft_entry__ import load_package
    from oracle import cref
    load_package()
    from mpc_protocols_amd import sharding
    n, t, groups = 7, 2, 13                                  # 13 chunks of 2t+1 = 5 over 2 ranks: 7 + 6
    m = 2 * t + 1
    N = groups * m
    a, b, r = cref.fill_random(1, N), cref.fill_random(2, N), cref.fill_random(3, N)
    co = cref.fill_random(13, N * (t + 1)).reshape(N, t + 1, 4)
    co[:, 0] = r
    rc, srt = cref.compute_shares(co, n, t)                  # [n][N] sharings of r

    def result(lo_e, hi_e):  # every party's [c]_t for elements [lo_e, hi_e), from those elements alone
        opened = cref.fr_binop("sub", cref.fr_binop("mul", a[lo_e:hi_e], b[lo_e:hi_e]), r[lo_e:hi_e])
        return np.stack([cref.triple_finalize(np.ascontiguousarray(srt[p, lo_e:hi_e]), opened)[1] for p in range(n)])

    lo, hi = bench.shard_range(groups, rank, world)
    mine = torch.from_numpy(result(lo * m, hi * m).view(np.int64))
    got = bench.gather_ragged(sharding, dist, mine, N, m)
    want = result(0, N)
    q.put((rank, bool(np.array_equal(got.numpy().view(np.uint64), want)), tuple(got.shape), hi - lo))
    dist.destroy_process_group()


def test_cfg4_shard_and_gather_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 31) % 1000
    procs = [ctx.Process(target=_cfg4_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(0, True, (7, 65, 4), 7), (1, True, (7, 65, 4), 6)]
