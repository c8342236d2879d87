"""The per-context table cache under churn: every new sender set is a new device table; the cache is bounded (512
tables, two-phase eviction) and tables / scratch that a captured HIP graph references are pinned.  A long-running
node keeps meeting new sender sets, so: results stay bit-exact across evictions, and a graph captured before the
churn still replays correctly after it."""
import itertools

import numpy as np
import pytest
import torch

from __graft_entry__ import load_package
from oracle import cref

pytestmark = pytest.mark.gpu


def test_graph_survives_cache_eviction():
    eng = load_package().Engine(0)
    n, t, d, G = 16, 3, 3, 64
    x = cref.fill_random(21, G * (d + 1)).reshape(G, d + 1, 4)
    rc, y = eng.compute_shares(x, n, d)
    assert rc == 0
    dev = torch.device("cuda:0")
    st = torch.cuda.Stream()
    ids_a = list(range(n))
    ev = torch.from_numpy(y.view(np.int64)).to(dev)
    out = torch.zeros((G, d + 1, 4), dtype=torch.int64, device=dev)
    stat = torch.zeros((G,), dtype=torch.uint8, device=dev)
    summ = torch.zeros((4,), dtype=torch.int32, device=dev)

    def call():
        assert eng.dev_batch_recover(ids_a, ev.data_ptr(), G, n, d, t, out.data_ptr(), 0, stat.data_ptr(), summ.data_ptr(),
                                     st.cuda_stream) == 0

    with torch.cuda.stream(st):
        call()  # eager: builds tables + scratch
        st.synchronize()
        eng.graph_begin(st.cuda_stream)
        call()
        g = eng.graph_end(st.cuda_stream)
    # a call whose table does not exist yet cannot be captured: clear error, and the context keeps working
    with torch.cuda.stream(st):
        eng.graph_begin(st.cuda_stream)
        rc = eng.dev_batch_recover(ids_a[:-1], ev.data_ptr(), G, n, d, t, out.data_ptr(), 0, stat.data_ptr(), summ.data_ptr(),
                                   st.cuda_stream)
        assert rc != 0 and "eagerly" in eng.last_error()
        eng.graph_destroy(eng.graph_end(st.cuda_stream))

    def replay_and_check():
        out.zero_()
        with torch.cuda.stream(st):
            ev[2, 5, 0] ^= 1  # refill: one lie, so the replay also walks the OEC/Gao kernel and its scratch
            eng.graph_launch(g, st.cuda_stream)
            ev[2, 5, 0] ^= 1
        st.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint64), x)
        assert int(stat.cpu().numpy().sum()) == 1 and int(stat[5]) == 1

    replay_and_check()
    # churn: > 512 tables from distinct sender sets
    sets = list(itertools.islice(itertools.combinations(range(n), 12), 700))
    for k, ids in enumerate(sets):
        rows = np.ascontiguousarray(y[list(ids)][:, :8])
        rows[1, k % 8, 0] ^= np.uint64(1)  # one lie -> fallback path and its tables too
        rc, co, nco, stt = eng.batch_recover(list(ids), rows, n, d, t)
        assert rc == 0 and np.array_equal(co, x[:8]) and int(stt.sum()) == 1, (k, ids)
    cs = eng.cache_stats()
    assert cs["evictions"] >= 1 and cs["pinned"] >= 2 and cs["tables"] <= 512, cs
    replay_and_check()  # the graph's tables were pinned, not evicted
    with torch.cuda.stream(st):
        call()
    st.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint64), x)
    # the first sets again (evicted, rebuilt): same results
    for ids in sets[:5]:
        rc, co, _, _ = eng.batch_recover(list(ids), np.ascontiguousarray(y[list(ids)][:, :8]), n, d, t)
        assert rc == 0 and np.array_equal(co, x[:8])
    eng.graph_destroy(g)
    eng.close()


def test_contexts_release_their_memory():
    """create -> use (small zero-copy calls, a large staged call, a fallback decode, a stream + scratch) -> destroy,
    twenty times: the device's free memory must come back (tables, scratch, staging pools, pinned blocks)."""
    pkg = load_package()
    n, t, d = 16, 5, 5

    def cycle():
        eng = pkg.Engine(0)
        x = cref.fill_random(5, 4096 * (d + 1)).reshape(4096, d + 1, 4)
        rc, y = eng.compute_shares(x, n, d)                      # large enough for the device-buffer path
        assert rc == 0
        rc, y1 = eng.compute_shares(x[:3], n, d)                 # zero-copy path
        ev = np.ascontiguousarray(y[:, :64])
        ev[0, 7, 0] ^= np.uint64(1)
        rc, co, nco, st = eng.batch_recover(list(range(n)), ev, n, d, t)   # fallback: scratch + OEC/Gao tables
        assert rc == 0 and np.array_equal(co, x[:64]) and st[7] == 1
        s = eng.stream_create()
        dx = eng.dev_alloc(x.nbytes)
        dy = eng.dev_alloc(n * 4096 * 32)
        eng.h2d(dx, x, s)
        assert eng.dev_compute_shares(dx, 4096, n, d, dy, s) == 0
        eng.sync(s)
        eng.dev_free(dx)
        eng.dev_free(dy)
        eng.stream_destroy(s)
        eng.close()

    cycle()  # first use: runtime-internal allocations (code objects, queues) happen once
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(20):
        cycle()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (8 << 20), (free0, free1)
