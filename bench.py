#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Shamir share-arithmetic hot path on MI355X.

Contract: python bench.py --gpus N --steps K --warmup W   (N > 1: one rank per GPU under torch.distributed.run; started
WITHOUT a launcher -- no WORLD_SIZE in the environment -- bench.py starts that launcher itself as a child process before
anything touches the GPU and relays rank 0's JSON line; --single-process drives the N devices from ONE process instead, a
context and a host thread per device, with the library's own gather: what a Rust node that owns all its devices does).  A "step" is one pass of RobustShare::compute_shares over one resident batch:
BASELINE.json configs[1] -- n=16, t=5, 2^20 secrets, 256-bit Fr -- per GPU (weak scaling: batches are
independent, there is no data-path collective).  Rank 0 prints ONE JSON line.

  value      shares/s (share evaluations), whole job (n * B * N * K / max-over-ranks time), inputs resident in HBM
  roofline   dominant kernel against the HBM roofline: algorithmic bytes per launch (704 B/secret)
             / average launch duration measured with HIP events on the launch stream
  cpu_baseline  the C restatement of the reference algorithm (oracle/, "port") timed on this box's
             host cores on a bounded sample of the same workload (single thread: the reference runs
             its arithmetic inline in one tokio task per party, no rayon).  This leg is the ONLY place
             the oracle is used; it also checks the first 512 secrets of the benchmarked buffers.
  extra      cfg3 encode/decode rates, element-wise rate, the register-resident modmul ceiling
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def encode_kernel_name(impl):
    """the kernel one compute_shares launch of BASELINE configs[1] is (rocprofv3 --kernel-trace shows it under this name): the
    matrix-core encode with the 16 domain points in 8 pairs for the default field arithmetic, the generic Horner kernel for
    the saturated-limb cross-check"""
    return "k_mfma_bfly<6,12,8>" if impl == "u29" else "k_eval_generic<Sat32>"


def shard_range(total: int, rank: int, world: int):
    """contiguous batch shard of `rank`"""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def timed_steps(step_fn, steps, warmup, sync_fn, barrier_fn, max_reduce_fn, prewarm_s=0.0, events=None, manage_gc=True):
    """Untimed pre-warm by TIME (launch until prewarm_s seconds have passed: the chip's clocks and caches need a few
    hundred ms of load to settle, which a count of W short launches does not cover), then W untimed warmups, then
    EXACTLY K steps bracketed by barrier+sync.  `events` = (ev0, ev1, record) brackets the SAME K launches with two HIP
    events on the launch stream: kernel_ms comes from the timed region itself.  Returns a dict; "secs" is the
    max-over-ranks wall time of the K steps."""
    # The interpreter's cyclic collector must not land in the timed region: with torch imported a full collection takes
    # ~40 ms (seen as ONE 38 ms host call among thousands of 15 us ones, tools/time_needed_only.py), ten times the whole
    # K = 20 region.  Collect NOW -- before the pre-warm, so that the pause does not let the chip's clocks fall again
    # right before the timed launches -- and keep it off until they are done.
    # (manage_gc = False: the caller -- one process driving several devices from threads -- holds the collector off for all of
    # them; a thread that re-enabled it on finishing would switch it on under the others' timed regions)
    gc_was_on = False
    if manage_gc:
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
    t_pre = time.perf_counter()
    pre_steps = 0
    while prewarm_s > 0 and time.perf_counter() - t_pre < prewarm_s:
        for _ in range(32):
            step_fn()
        pre_steps += 32
        sync_fn()
    prewarm_ms = (time.perf_counter() - t_pre) * 1e3 if prewarm_s > 0 else 0.0
    for _ in range(warmup):
        step_fn()
    sync_fn()
    barrier_fn()
    if events:
        events[2](events[0])
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    if events:
        events[2](events[1])
    sync_fn()
    dt = time.perf_counter() - t0
    if gc_was_on:
        gc.enable()
    barrier_fn()
    out = {"secs": max_reduce_fn(dt), "prewarm_ms": prewarm_ms, "prewarm_steps": pre_steps}
    if events:
        out["kernel_ms"] = events[0].elapsed_time(events[1]) / steps
    return out


def cpu_baseline(n, d, sample, check=None, min_seconds=10.0):
    """The only place bench.py touches oracle/: the CPU restatement timed as the reported baseline ("port"), and --
    as the checker it is -- compared once with the benchmarked GPU buffers (check = (coeffs, shares) of the first
    secrets of the timed batch, copied to the host)."""
    from oracle import cref
    if check is not None:
        rc, want = cref.compute_shares(np.ascontiguousarray(check[0]), n, d)
        assert rc == 0 and np.array_equal(check[1], want), "bench output differs from the oracle"
    x = cref.fill_random(0xC0FFEE01, sample * (d + 1)).reshape(sample, d + 1, 4)
    cref.compute_shares(x[:1024], n, d)
    # a bounded sample of the same workload: repeat the pass over `sample` secrets until >= 10 s of CPU work
    reps, dt = 0, 0.0
    t0 = time.perf_counter()
    while reps < 16 and dt < min_seconds:
        rc, _ = cref.compute_shares(x, n, d)
        assert rc == 0
        reps += 1
        dt = time.perf_counter() - t0
    return {"value": n * sample * reps / dt, "unit": "shares/s", "cores": 1, "kind": "port",
            "sample": f"compute_shares n={n} d={d}: {reps} passes over {sample} secrets, single thread, "
                      f"{os.path.basename(cref.build())}", "seconds": round(dt, 2)}


def cpu_baseline_recon(n, d, t, sample, check=None, min_seconds=5.0):
    """The recon half of the metric on one host thread: the restatement of batch_recover_secret (shared Lagrange basis,
    per-chunk verify + recover, robust_interpolate.rs:284-443) on valid codewords.  check = (evals, coeffs) of the first
    chunks of the benchmarked decode."""
    from oracle import cref
    ids = list(range(n))
    if check is not None:
        rc, want, _, st = cref.batch_recover(ids, np.ascontiguousarray(check[0]), n, d, t)
        assert rc == 0 and not st.any() and np.array_equal(check[1], want), "bench decode differs from the oracle"
    x = cref.fill_random(0xC0FFEE02, sample * (d + 1)).reshape(sample, d + 1, 4)
    rc, y = cref.vandermonde_apply(x, n, d)
    assert rc == 0
    reps, dt = 0, 0.0
    t0 = time.perf_counter()
    while reps < 64 and dt < min_seconds:
        rc, co, _, st = cref.batch_recover(ids, y, n, d, t)
        assert rc == 0
        reps += 1
        dt = time.perf_counter() - t0
    assert np.array_equal(co, x)
    return {"value": sample * reps / dt, "unit": "recons/s", "cores": 1, "kind": "port",
            "sample": f"batch_recover n={n} d={d} t={t}: {reps} passes over {sample} chunks of valid codewords, single thread",
            "seconds": round(dt, 2)}


def cpu_baseline_threads(n, d, sample, min_seconds=2.0):
    """the same restatement on every host core (the reference itself is single-threaded on this path; SURVEY 8(d)
    asks for both).  ctypes releases the GIL during the C call; every thread repeats its pass until >= 2 s."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import cref
    cores = len(os.sched_getaffinity(0))
    per = max(1024, sample // cores)
    xs = [cref.fill_random(0xC0FFEE10 + i, per * (d + 1)).reshape(per, d + 1, 4) for i in range(cores)]

    def work(x):
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < min_seconds:
            assert cref.compute_shares(x, n, d)[0] == 0
            reps += 1
        return reps

    with ThreadPoolExecutor(cores) as ex:
        t0 = time.perf_counter()
        reps = list(ex.map(work, xs))
        dt = time.perf_counter() - t0
    return {"value": n * per * sum(reps) / dt, "unit": "shares/s", "cores": cores, "kind": "port",
            "sample": f"compute_shares n={n} d={d}: {per} secrets per thread, {min(reps)}..{max(reps)} passes each",
            "seconds": round(dt, 2)}


def cpu_baseline_pipeline(kind, n, t, min_seconds=5.0, k=16, f=4):
    """configs 4 / 5 on one host thread: ONE party's step of the pipeline in the oracle's element-wise / per-chunk
    restatement (the reference runs a party's arithmetic inline in one task), on a bounded sample repeated until
    >= min_seconds.  The GPU line covers all n simulated parties per step: `value` here is one party's rate."""
    from oracle import cref
    ids = list(range(n))
    if kind == "cfg4":
        m, d = 2 * t + 1, 2 * t
        G = 1 << 11
        N = G * m
        a, b, r2t, rt = (cref.fill_random(0xC0FFEE03 + i, N) for i in range(4))
        reps, dt, t0 = 0, 0.0, time.perf_counter()
        while reps < 1024 and dt < min_seconds:
            rc, x = cref.triple_local(a, b, r2t)                              # triple_generation.rs:333-340
            rc1, y = cref.vandermonde_apply(x.reshape(G, m, 4), n, d)         # batch_recon.rs:157-165
            rc2, z, st = cref.batch_recover_p0(ids, y, n, d, t)               # EvalBatch arm :371-389 (G chunks of this recipient)
            rc3, co, _, st2 = cref.batch_recover(ids, y, n, d, t)             # RevealBatch arm :457-467
            rc4, c = cref.triple_finalize(rt, co.reshape(N, 4))               # triple_generation.rs:196-208
            assert rc == rc1 == rc2 == rc3 == rc4 == 0 and not st.any() and not st2.any()
            reps += 1
            dt = time.perf_counter() - t0
        assert np.array_equal(co.reshape(N, 4), x)
        return {"value": N * reps / dt, "unit": "triples/s", "cores": 1, "kind": "port", "seconds": round(dt, 2),
                "sample": f"one party's triple_gen step (local product, encode, the two BatchRecon decodes, finalize) over {N} triples, "
                          f"{reps} passes, single thread; the GPU line runs this for all {n} parties per step",
                "value_all_parties": N * reps / dt / n}
    N = 1 << 13
    x, y, ta, tb, tc, rint = (cref.fill_random(0xC0FFEE05 + i, N) for i in range(6))
    rbits = cref.fill_random(0xC0FFEE0B, f * N).reshape(f, N, 4)
    # what a party receives for an open: 2t+1 senders' shares of valid degree-t sharings
    sec = cref.fill_random(0xC0FFEE0C, 2 * N * (t + 1)).reshape(2 * N, t + 1, 4)
    rc, opn = cref.compute_shares(sec, n, t)
    assert rc == 0
    opn = np.ascontiguousarray(opn[: 2 * t + 1])
    oids = list(range(2 * t + 1))
    reps, dt, t0 = 0, 0.0, time.perf_counter()
    while reps < 64 and dt < min_seconds:
        rc, dsh, esh = cref.beaver_open_shares(ta, tb, x, y)                  # multiplication.rs:417-426
        rc1, de, st = cref.batch_recover_p0(oids, opn, n, t, t)               # reconstruct_rbc :102-139 (a - x and b - y: 2 N opens)
        rc2, z = cref.beaver_finalize(tc, x, y, de[:N], de[N:])               # finalize_mul :57-100
        rc3, rd = cref.truncpr_rdash(rbits, f)                                # truncpr.rs:277-297
        rc4, osh = cref.truncpr_open_share(z, rd, rint, k, f)
        rc5, cop, st2 = cref.batch_recover_p0(oids, opn[:, :N], n, t, t)      # truncpr.rs:215
        rc6, out = cref.truncpr_finalize(z, rd, cop, f)                       # :216-220
        assert rc == rc1 == rc2 == rc3 == rc4 == rc5 == rc6 == 0 and not st.any() and not st2.any()
        reps += 1
        dt = time.perf_counter() - t0
    return {"value": N * reps / dt, "unit": "fpmuls/s", "cores": 1, "kind": "port", "seconds": round(dt, 2),
            "sample": f"one party's fpmul step (Beaver open shares, 2 N opens from 2t+1 senders, finalize_mul, r', TruncPr open share, "
                      f"N opens, TruncPr finalize; (k, f) = ({k}, {f})) over {N} elements, {reps} passes, single thread; the GPU line "
                      f"runs this for all {n} parties per step", "value_all_parties": N * reps / dt / n}


def traffic_record(key):
    tr = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        return json.load(open(tr)).get(key)
    except Exception:
        return None


def launch_command(n, argv, port):
    """the child command of an N-rank run: exactly what the driver's own launcher line is"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + [a for a in argv if a != "--print-launch"]


def launch_ranks(n, argv, print_only=False):
    """N > 1 without a launcher: start `python -m torch.distributed.run ... bench.py <same arguments>` as a child, pass its
    stderr through, print the JSON line of rank 0 (the last stdout line that parses as the bench record) and return the
    child's exit code.  Never an exec: a process that may have initialised the GPU must not be replaced."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = launch_command(n, argv, port)
    if print_only:
        print(json.dumps({"launch": cmd}))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL / IPC on this stack need the dmabuf path
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    record = None
    for line in proc.stdout:
        try:
            obj = json.loads(line)
            if isinstance(obj, dict) and "metric" in obj:
                record = line.rstrip("\n")
                continue
        except ValueError:
            pass
        sys.stderr.write(line)  # anything else the ranks printed
    rc = proc.wait()
    if record is not None:
        print(record)
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks exited 0 without a bench record\n")
        rc = 1
    return rc


def bench_single_process(args, torch):
    """--single-process --gpus N: one process, one Engine (context) + one host thread per device.  Every device runs the
    two halves of the metric on its own 2^20-element shard (weak scaling, no data-path collective: the same steps as the
    process-per-GPU mode), the threads meet at a barrier before and after the K timed steps and the slowest one sets the
    time; then the shares of all devices are gathered party-major onto device 0 by hbmpc_dev_gather_party_major (one strided
    copy per shard over xGMI) -- the final gather a host that owns all its devices issues, timed separately."""
    import threading
    from __graft_entry__ import load_package
    pkg = load_package()
    N = args.gpus
    n, t, d = 16, 5, 5
    n3, t3, d3 = 31, 10, 10
    B = 1 << args.log2_batch
    devno = [0 if args.same_device else i for i in range(N)]  # --same-device: rehearsal of the control flow on one GPU
    engines = [pkg.Engine(devno[i], impl=args.impl) for i in range(N)]
    for e in engines:
        e.set_matrix_cores(args.recon_kernels == "mfma")
    peer = [engines[0].peer_access(e) for e in engines]
    bar = threading.Barrier(N)
    res = [None] * N
    keep = [None] * N
    errs = []

    def worker(i):
        try:
            torch.cuda.set_device(devno[i])
            dev = torch.device("cuda", devno[i])
            eng = engines[i]
            ts = torch.cuda.Stream(device=dev)
            torch.cuda.set_stream(ts)
            stream = ts.cuda_stream
            torch.manual_seed(0xC0FFEE01 + i)
            coeffs = _rand_fr(torch, dev, B, d + 1)
            shares = torch.empty((n, B, 4), dtype=torch.int64, device=dev)

            def step():
                rc = eng.dev_compute_shares(coeffs.data_ptr(), B, n, d, shares.data_ptr(), stream)
                if rc != 0:
                    raise RuntimeError(f"hbmpc_dev_compute_shares -> {rc}: {eng.last_error()}")

            sync = lambda: torch.cuda.synchronize(dev)
            ev = lambda: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), lambda e: e.record(ts))
            step()
            sync()
            tm = timed_steps(step, args.steps, args.warmup, sync, bar.wait, lambda x: x, prewarm_s=args.prewarm_seconds, events=ev(), manage_gc=False)
            torch.manual_seed(0xC0FFEE02 + i)
            x = _rand_fr(torch, dev, B, d3 + 1)
            y = torch.empty((n3, B, 4), dtype=torch.int64, device=dev)
            co = torch.empty((B, d3 + 1, 4), dtype=torch.int64, device=dev)
            st = torch.empty((B,), dtype=torch.uint8, device=dev)
            summ = torch.zeros((4,), dtype=torch.int32, device=dev)
            ids = list(range(n3))
            assert eng.dev_vandermonde_apply(x.data_ptr(), B, n3, d3, y.data_ptr(), stream) == 0, eng.last_error()

            def rstep():
                rc = eng.dev_batch_recover(ids, y.data_ptr(), B, n3, d3, t3, co.data_ptr(), 0, st.data_ptr(), summ.data_ptr(), stream)
                if rc != 0:
                    raise RuntimeError(f"hbmpc_dev_batch_recover -> {rc}: {eng.last_error()}")

            rstep()
            sync()
            assert bool((co == x).all()) and int(st.max()) == 0, "decode(encode(x)) != x"
            rtm = timed_steps(rstep, args.steps, args.warmup, sync, bar.wait, lambda x: x, prewarm_s=args.prewarm_seconds, events=ev(), manage_gc=False)
            res[i] = (tm, rtm)
            keep[i] = (shares, ts)
        except BaseException as e:  # a thread that dies must not leave the others at the barrier
            errs.append((i, repr(e)))
            bar.abort()

    th = [threading.Thread(target=worker, args=(i,)) for i in range(N)]
    gc.collect()
    gc_was_on = gc.isenabled()
    gc.disable()  # process-wide, for every thread's timed region
    for x in th:
        x.start()
    for x in th:
        x.join()
    if gc_was_on:
        gc.enable()
    if errs:
        raise SystemExit(f"--single-process: {errs}")
    secs = max(r[0]["secs"] for r in res)
    rsecs = max(r[1]["secs"] for r in res)
    kernel_ms = max(r[0]["kernel_ms"] for r in res)
    rkernel_ms = max(r[1]["kernel_ms"] for r in res)
    algo = (d + 1 + n) * 32 * B
    r_algo = (d3 + t3 + 1 + d3 + 1) * 32 * B
    out = {
        "metric": "shares/sec (compute_shares) + recons/sec (batch_recon), 256-bit Fr, 1/2/4/8 GPU",
        "value": n * B * N * args.steps / secs, "unit": "shares/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": secs / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"compute_shares n={n} t={t} batch=2^{args.log2_batch} secrets per GPU (BASELINE configs[1])",
                   "field": "bls12-381 Fr", "parallelism": f"batch-sharded x{N}, ONE process with a context and a host thread per device, "
                                                             "no data-path collective", "field_impl": args.impl},
        "roofline": {"bound": "hbm", "achieved": algo / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": algo / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel": encode_kernel_name(args.impl),
                     "kernel_ms": kernel_ms, "algorithmic_bytes": algo, "note": "slowest device; per-device kernel_ms in per_device"},
        "recon": {"value": B * N * args.steps / rsecs, "unit": "recons/s", "ms_per_step": rsecs / args.steps * 1e3,
                  "roofline": {"bound": "hbm", "achieved": r_algo / (rkernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": r_algo / (rkernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel_ms": rkernel_ms}},
        "per_device": [{"device": i, "kernel_ms": res[i][0]["kernel_ms"], "recon_kernel_ms": res[i][1]["kernel_ms"]} for i in range(N)],
    }
    out["recons_per_s"] = out["recon"]["value"]
    if not args.no_final_gather:
        torch.cuda.set_device(0)
        full = torch.empty((n, B * N, 4), dtype=torch.int64, device=torch.device("cuda", 0))
        ts0 = keep[0][1]
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ptrs = [keep[i][0].data_ptr() for i in range(N)]
        for rep in range(2):  # the first call enables peer access and maps the peers' memory
            g0.record(ts0)
            rc = pkg.Engine.gather_party_major(engines, 0, ptrs, [B] * N, [B] * N, n, full.data_ptr(), B * N, sync_sources=True,
                                               stream=ts0.cuda_stream)
            assert rc == 0, engines[0].last_error()
            g1.record(ts0)
            torch.cuda.synchronize()
        ms = g0.elapsed_time(g1)
        for i in range(N):
            assert torch.equal(full[:, i * B:(i + 1) * B].cpu(), keep[i][0].cpu()), f"gathered shard {i} differs"
        # the same gather source by source (root + ONE peer per call): what each link delivers, next to whether the root reads that
        # peer's memory directly (peer access over xGMI) or through a staged copy
        per_source = []
        for i in range(1, N):
            for rep in range(2):
                g0.record(ts0)
                rc = pkg.Engine.gather_party_major([engines[0], engines[i]], 0, [ptrs[0], ptrs[i]], [0, B], [B, B], n, full.data_ptr(), B * N,
                                                   sync_sources=True, stream=ts0.cuda_stream)
                assert rc == 0, engines[0].last_error()
                g1.record(ts0)
                torch.cuda.synchronize()
            per_source.append({"source_device": devno[i], "peer_access_direct": bool(peer[i]), "ms": g0.elapsed_time(g1),
                               "GBps": n * B * 32 / g0.elapsed_time(g1) / 1e6})
        out["final_gather"] = {"ms": ms, "bytes_total": n * B * N * 32, "bytes_from_peers": n * B * (N - 1) * 32,
                               "GBps_from_peers": n * B * (N - 1) * 32 / ms / 1e6,
                               "copies": "one hipMemcpy2DAsync per shard on device 0's stream (hbmpc_dev_gather_party_major)",
                               "peer_access_direct": peer, "per_source": per_source}
    if args.same_device:
        out["rehearsal"] = "every context on GPU 0: a check of the one-process control flow, not a measurement"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--log2-batch", type=int, default=20)
    ap.add_argument("--cpu-sample-log2", type=int, default=22)
    ap.add_argument("--prewarm-seconds", type=float, default=0.5,
                    help="untimed launches of the same step until this much wall time has passed (before the W warmups)")
    ap.add_argument("--workload", default="cfg2+cfg3", choices=["cfg2+cfg3", "cfg4", "cfg5"],
                    help="cfg2+cfg3 (default): the two halves of BASELINE.json's metric, 2^20 secrets / chunks per GPU "
                         "(weak scaling).  cfg4 / cfg5: the triple_gen / fpmul pipelines of BASELINE configs[3] / [4], their "
                         "fixed batch sharded over the ranks (strong scaling)")
    ap.add_argument("--with-producers", action="store_true",
                    help="--workload cfg4: time run_preprocessing's whole triple part -- RanSha (a, b), DouSha + RanDouSha ([r]_t, [r]_2t) "
                         "and the triple generation -- device-resident from the dealers' polynomials to [c]_t")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-final-gather", action="store_true",
                    help="N > 1: skip the all-gather of the shares after the timed region (the path's only collective)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the barrier / max-reduce / final gather (nccl = RCCL; gloo: rehearsal of "
                         "the N > 1 control flow where RCCL cannot run, e.g. several ranks on one GPU with --same-device)")
    ap.add_argument("--same-device", action="store_true",
                    help="rehearsal only: every rank uses GPU 0 (with --dist-backend gloo); the numbers then mean nothing")
    ap.add_argument("--single-process", action="store_true",
                    help="N > 1 from ONE process: an Engine (context) and a host thread per device, each on its shard, then "
                         "hbmpc_dev_gather_party_major to device 0 (no torch.distributed, no RCCL bootstrap)")
    ap.add_argument("--print-launch", action="store_true",
                    help="N > 1 without a launcher: print the child command line as JSON and exit (used by the CPU tests)")
    ap.add_argument("--impl", default="u29", choices=["u29", "sat32"])
    ap.add_argument("--recon-kernels", default="mfma", choices=["mfma", "lane"],
                    help="batch_recover on the matrix cores (default) or with the lane-per-chunk kernels (A/B)")
    args = ap.parse_args()

    if args.gpus > 1 and not args.single_process and "WORLD_SIZE" not in os.environ:
        # started without a launcher: become the launcher's parent.  Nothing in this process has touched the GPU (torch is
        # not even imported yet), the ranks are CHILD processes, and this process only relays and exits with their code.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:], args.print_launch))

    import torch
    import torch.distributed as dist

    if args.single_process and args.gpus > 1:
        if not torch.cuda.is_available() or (torch.cuda.device_count() < args.gpus and not args.same_device):
            raise SystemExit(f"--single-process --gpus {args.gpus}: {torch.cuda.device_count() if torch.cuda.is_available() else 0} devices visible")
        print(json.dumps(bench_single_process(args, torch)))
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    from __graft_entry__ import load_package
    pkg = load_package()
    eng = pkg.Engine(local_rank, impl=args.impl)
    eng.set_matrix_cores(args.recon_kernels == "mfma")

    # an explicit (non-default) torch stream: its handle goes through the C ABI, so the kernels, the
    # torch.cuda.Event timings and the copies are all ordered on ONE stream
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0

    def barrier():
        if world > 1:
            dist.barrier()

    def max_reduce(x):
        if world == 1:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def events():
        return (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), lambda e: e.record())

    ctx = dict(args=args, torch=torch, dist=dist, dev=dev, eng=eng, stream=stream, rank=rank, world=world,
               barrier=barrier, max_reduce=max_reduce, events=events)
    if args.workload == "cfg2+cfg3":
        out = bench_metric(ctx)
    else:
        out = bench_pipeline(ctx)
    if args.same_device:
        out["rehearsal"] = "all ranks on GPU 0 over gloo: a check of the N > 1 control flow, not a measurement"
    if world > 1:
        # how many ranks the collective layer really saw (an all-reduce of ones) and which backend carried it
        ones = torch.ones(1, dtype=torch.int64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        out["ranks_seen"] = int(ones.item())
        out["backend"] = dist.get_backend()
        assert out["ranks_seen"] == world == out["n_gpus"], (out["ranks_seen"], world)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def bench_metric(ctx):
    """BASELINE.json's metric: shares/s of compute_shares (configs[1]) -- `value` -- and recons/s of batch_recon
    (configs[2]) -- `recon` --, both on this rank's 2^20-element shard, timed with the same barrier / max-over-ranks
    contract."""
    args, torch, dist, dev, eng, stream = (ctx[k] for k in ("args", "torch", "dist", "dev", "eng", "stream"))
    rank, world = ctx["rank"], ctx["world"]
    n, t, d = 16, 5, 5
    B = 1 << args.log2_batch  # per GPU (weak scaling)
    torch.manual_seed(0xC0FFEE01 + rank)  # synthetic inputs: canonical, uniform-ish field elements drawn on the device
    coeffs = _rand_fr(torch, dev, B, d + 1)                         # [B][d+1][4] resident in HBM (this rank's shard)
    shares = torch.empty((n, B, 4), dtype=torch.int64, device=dev)  # [n][B][4]
    torch.cuda.synchronize()

    def step():
        rc = eng.dev_compute_shares(coeffs.data_ptr(), B, n, d, shares.data_ptr(), stream)
        if rc != 0:
            raise RuntimeError(f"hbmpc_dev_compute_shares -> {rc}: {eng.last_error()}")

    step()
    torch.cuda.synchronize()
    # the first secrets of the benchmarked buffers, for the oracle check inside the cpu_baseline leg
    check = (coeffs[:512].cpu().numpy().view(np.uint64), shares[:, :512].cpu().numpy().view(np.uint64))
    tm = timed_steps(step, args.steps, args.warmup, torch.cuda.synchronize, ctx["barrier"], ctx["max_reduce"],
                     prewarm_s=args.prewarm_seconds, events=ctx["events"]())
    secs, kernel_ms = tm["secs"], tm["kernel_ms"]
    value = n * B * world * args.steps / secs
    algo_bytes = (d + 1 + n) * 32 * B
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9

    out = {
        # BASELINE.json's metric, verbatim.  `value` is its compute_shares half on BASELINE configs[1] (one share = one
        # evaluation of one secret's polynomial at one party's point); the batch_recon half is `recon` / `recons_per_s`
        "metric": "shares/sec (compute_shares) + recons/sec (batch_recon), 256-bit Fr, 1/2/4/8 GPU",
        "value": value, "unit": "shares/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": secs / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "prewarm_ms": tm["prewarm_ms"], "prewarm_steps": tm["prewarm_steps"],
        "config": {"workload": f"compute_shares n={n} t={t} batch=2^{args.log2_batch} secrets per GPU (BASELINE configs[1])",
                   "field": "bls12-381 Fr", "parallelism": f"batch-sharded x{world}, no data-path collective",
                   "field_impl": args.impl},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": encode_kernel_name(args.impl), "kernel_ms": kernel_ms, "algorithmic_bytes": algo_bytes,
                     "kernel_ms_source": "HIP events around the K timed launches themselves"},
    }
    rec = traffic_record(f"compute_shares_n{n}_d{d}_B2^{args.log2_batch}_{args.impl}")
    if rec:
        out["roofline"]["traffic"] = rec["hbm_bytes_per_launch"]
        out["roofline"]["traffic_source"] = rec["source"]
    # context for `frac`, measured IN THIS RUN on the same buffers right behind the timed region (the chip is warm): the same
    # loads and stores with no arithmetic at all (hbmpc_dev_traffic_ubench: x[B][d+1] -> y[n][B], every loaded word kept live)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        assert eng.dev_traffic_ubench(coeffs.data_ptr(), B, d + 1, shares.data_ptr(), n, stream) == 0, eng.last_error()
    e0.record()
    for _ in range(20):
        eng.dev_traffic_ubench(coeffs.data_ptr(), B, d + 1, shares.data_ptr(), n, stream)
    e1.record()
    torch.cuda.synchronize()
    no_arith_ms = e0.elapsed_time(e1) / 20
    out["roofline"]["same_traffic_no_arithmetic_ms"] = no_arith_ms
    out["roofline"]["same_traffic_no_arithmetic_GBps"] = algo_bytes / (no_arith_ms * 1e-3) / 1e9
    out["roofline"]["same_traffic_no_arithmetic_source"] = ("hbmpc_dev_traffic_ubench on the benchmarked buffers, 20 launches behind the "
                                                            "timed region of this run (HIP events)")
    step()  # the shares the checks below look at
    torch.cuda.synchronize()

    final_gather = None
    if world > 1 and not args.no_final_gather:
        # SURVEY 8(e) / north_star: "RCCL over xGMI used only for the final gather" -- outside the timed region
        from mpc_protocols_amd import sharding
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dist.barrier()
        g0.record()
        full, spans = sharding.gather_shards(shares, B * world)  # [n][world][B]: a view of the one receive buffer
        g1.record()
        torch.cuda.synchronize()
        assert tuple(full.shape[:3]) == (n, world, B) and spans[rank] == (rank * B, rank * B + B) and torch.equal(full[:, rank], shares)
        final_gather = {"ms": g0.elapsed_time(g1), "bytes_per_rank": n * B * 32,
                        "GBps_per_rank_received": n * B * 32 * (world - 1) / g0.elapsed_time(g1) / 1e6,
                        "collective": "all_gather_into_tensor of the [n][B] shares of every rank into one [world][n][B] buffer "
                                      "(RCCL), read party-major through a permuted view: no copy after the collective"}
        del full
    del coeffs, shares

    # ---- the recon half: batch_recover_secret on BASELINE configs[2], this rank's 2^20 chunks ----
    n3, t3, d3, G = 31, 10, 10, 1 << args.log2_batch
    torch.manual_seed(0xC0FFEE02 + rank)
    x = _rand_fr(torch, dev, G, d3 + 1)
    y = torch.empty((n3, G, 4), dtype=torch.int64, device=dev)
    co = torch.empty((G, d3 + 1, 4), dtype=torch.int64, device=dev)
    st = torch.empty((G,), dtype=torch.uint8, device=dev)
    summ = torch.zeros((4,), dtype=torch.int32, device=dev)
    ids = list(range(n3))
    torch.cuda.synchronize()
    assert eng.dev_vandermonde_apply(x.data_ptr(), G, n3, d3, y.data_ptr(), stream) == 0, eng.last_error()

    def rstep():
        rc = eng.dev_batch_recover(ids, y.data_ptr(), G, n3, d3, t3, co.data_ptr(), 0, st.data_ptr(), summ.data_ptr(), stream)
        if rc != 0:
            raise RuntimeError(f"hbmpc_dev_batch_recover -> {rc}: {eng.last_error()}")

    rstep()
    torch.cuda.synchronize()
    assert bool((co == x).all()) and int(st.max()) == 0, "decode(encode(x)) != x"
    rcheck = (y[:, :256].cpu().numpy().view(np.uint64), co[:256].cpu().numpy().view(np.uint64))
    rtm = timed_steps(rstep, args.steps, args.warmup, torch.cuda.synchronize, ctx["barrier"], ctx["max_reduce"],
                      prewarm_s=args.prewarm_seconds, events=ctx["events"]())
    r_algo = (d3 + t3 + 1 + d3 + 1) * 32 * G
    r_ach = r_algo / (rtm["kernel_ms"] * 1e-3) / 1e9
    rk = "k_mfma_rows<11,1,12>" if args.recon_kernels == "mfma" and args.impl == "u29" else "k_batch_recover<U29,11,false>"
    out["recon"] = {
        "value": G * world * args.steps / rtm["secs"], "unit": "recons/s", "ms_per_step": rtm["secs"] / args.steps * 1e3,
        "secrets_per_s": (d3 + 1) * G * world * args.steps / rtm["secs"],
        "prewarm_ms": rtm["prewarm_ms"], "prewarm_steps": rtm["prewarm_steps"],
        "config": {"workload": f"batch_recover_secret n={n3} t={t3} d={d3}, 2^{args.log2_batch} chunks per GPU, all {n3} senders "
                               f"supplied, valid codewords (BASELINE configs[2])", "kernels": args.recon_kernels},
        "roofline": {"bound": "hbm", "achieved": r_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": r_ach / HBM_PEAK_GBS,
                     "traffic": None, "kernel": rk, "kernel_ms": rtm["kernel_ms"], "algorithmic_bytes": r_algo,
                     "note": "one decode call = the optimistic kernel + the (empty-list) fallback launches; kernel_ms covers the call"},
    }
    rec = traffic_record(f"batch_recover_n{n3}_d{d3}_t{t3}_G2^{args.log2_batch}_{args.recon_kernels}")
    if rec:
        out["recon"]["roofline"]["traffic"] = rec["hbm_bytes_per_launch"]
        out["recon"]["roofline"]["traffic_source"] = rec["source"]
    out["recons_per_s"] = out["recon"]["value"]
    out["roofline_recon"] = out["recon"]["roofline"]
    if final_gather:
        out["final_gather"] = final_gather
    del x, y, co, st

    if rank == 0 and world == 1:
        out["cpu_baseline"] = cpu_baseline(n, d, 1 << args.cpu_sample_log2, check)
        out["cpu_baseline_recon"] = cpu_baseline_recon(n3, d3, t3, 1 << 15, rcheck)
        out["recon"]["cpu_baseline"] = out["cpu_baseline_recon"]
        if not args.no_extra:
            out["cpu_baseline_all_cores"] = cpu_baseline_threads(n, d, 1 << args.cpu_sample_log2)
            out["extra"] = extra_measurements(eng, torch, dev, stream)
            # context for `frac` (report against the vendor peak AND what the chip delivers): a plain
            # device copy measured in this run (the no-arithmetic form of this kernel's own traffic is measured above)
            out["roofline"]["peak_measured_copy_GBps"] = out["extra"]["device_copy_GBps"]
    return out


def bench_pipeline(ctx):
    """BASELINE configs[3] / [4]: the triple_gen (2^22 triples) and fpmul (2^18 fixed-point muls) pipelines, n = 16,
    t = 5, all 16 simulated parties of a batch element on one GPU and the batch sharded over the ranks (strong scaling:
    independent preprocessing batches, reference honeybadger/mod.rs:1334-1393, triple_gen/triple_generation.rs:304-364);
    no data-path collective, one all-gather of the result shares after the timed region."""
    args, torch, dist, dev, eng, stream = (ctx[k] for k in ("args", "torch", "dist", "dev", "eng", "stream"))
    rank, world = ctx["rank"], ctx["world"]
    from __graft_entry__ import load_package
    pl = load_package().pipelines
    n, t = 16, 5
    torch.manual_seed(0xC0FFEE03 + rank)
    pre = None
    if args.workload == "cfg4" and args.with_producers:
        m = 2 * t + 1
        groups = (1 << 22) // m
        lo, hi = shard_range(groups, rank, world)
        N = (hi - lo) * m
        pre = pl.Preprocessing(eng, n, t, N, stream)
        tg = pre.tg
        # the dealers' polynomials (what each party's rng would draw), filled on the device; DouSha deals BOTH sharings of one secret
        sec0 = {}
        for ptr, K, deg in ((pre.rs.coeffs, pre.K_rs, t), (pre.rd.coeffs_t, pre.K_rd, t), (pre.rd.coeffs_2t, pre.K_rd, 2 * t)):
            for p in range(n):   # dealer by dealer: bounds the temporary
                co = _rand_fr(torch, dev, K, deg + 1)
                if ptr == pre.rd.coeffs_2t:
                    co[:, 0] = sec0[p]
                elif ptr == pre.rd.coeffs_t:
                    sec0[p] = co[:, 0].clone()
                eng.d2d(ptr + p * K * (deg + 1) * 32, co.data_ptr(), K * (deg + 1) * 32, stream)
                torch.cuda.synchronize()
                del co
        del sec0
        pre.run(check=True)                           # verifiers say OK, every decode reports zero failures
        step, total, unit, pipe, result_ptr = (lambda: pre.run(check=False)), groups * m, "triples/s", pre, tg.c
        what = (f"run_preprocessing's triple part n={n} t={t}: RanSha ({pre.K_rs} batch elements per dealer -> a, b), DouSha + RanDouSha "
                f"({pre.K_rd} -> [r]_t, [r]_2t), triple_gen of {groups * m} Beaver triples, {N} on this rank, from the dealers' polynomials")
    elif args.workload == "cfg4":
        m = 2 * t + 1
        groups = (1 << 22) // m                      # chunks of 2t+1 triples (BatchRecon's unit)
        lo, hi = shard_range(groups, rank, world)
        N = (hi - lo) * m
        tg = pl.TripleGen(eng, n, t, N, stream)
        a, b, r = (_rand_fr(torch, dev, N) for _ in range(3))
        _share_on_device(eng, torch, dev, stream, a, n, t, tg.a)
        _share_on_device(eng, torch, dev, stream, b, n, t, tg.b)
        _share_on_device(eng, torch, dev, stream, r, n, t, tg.rt)
        _share_on_device(eng, torch, dev, stream, r, n, 2 * t, tg.r2t)
        tg.run(check=True)                            # every decode reports zero failures
        step, total, unit, pipe, result_ptr = (lambda: tg.run(check=False)), groups * m, "triples/s", tg, tg.c
        what = f"triple_gen n={n} t={t}, {groups * m} Beaver triples (2^22 rounded to chunks of 2t+1), {N} on this rank"
    else:
        total = 1 << 18
        lo, hi = shard_range(total, rank, world)
        N, k, f = hi - lo, 16, 4
        fp = setup_fpmul(eng, torch, dev, stream, n, t, N, k, f)
        fp.run(check=True)                            # every open reports zero failures
        step, unit, pipe, result_ptr = (lambda: fp.run(check=False)), "fpmuls/s", fp, fp.out   # enqueue only, like cfg4
        what = (f"fpmul n={n} t={t} (k, f) = ({k}, {f}), 2^18 fixed-point multiplications, {N} on this rank; every open "
                f"interpolates from the first 2t+1 = {2 * t + 1} senders, as the reference does (multiplication.rs:388, truncpr.rs:202)")
    torch.cuda.synchronize()
    tm = timed_steps(step, args.steps, args.warmup, torch.cuda.synchronize, ctx["barrier"], ctx["max_reduce"],
                     prewarm_s=args.prewarm_seconds, events=ctx["events"]())
    out = {
        "metric": "shares/sec (compute_shares) + recons/sec (batch_recon), 256-bit Fr, 1/2/4/8 GPU",
        "value": total * args.steps / tm["secs"], "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": tm["secs"] / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic", "prewarm_ms": tm["prewarm_ms"], "prewarm_steps": tm["prewarm_steps"],
        "device_ms_per_step": tm["kernel_ms"],
        "config": {"workload": what + f" (BASELINE configs[{3 if args.workload == 'cfg4' else 4}])", "field": "bls12-381 Fr",
                   "parallelism": f"batch-sharded x{world}, all {n} simulated parties of a shard on one GPU, no data-path collective"},
    }
    # the dominant kernel of the step against the HBM roofline: its algorithmic bytes / its own launch duration, measured
    # live with HIP events around K launches of that call alone (same buffers, same stream)
    e0, e1, rec = ctx["events"]()
    if args.workload == "cfg4":
        d2 = 2 * t
        G = N // m
        dom = lambda: eng.dev_triple_encode_parties(tg.a, tg.b, tg.r2t, G, n, d2, n, tg.c, tg.Y, stream)
        dom_bytes = n * (3 * N + n * G) * 32          # per party: a, b, r2t read, Y[party][n][G] written
        dom_name = "k_mfma_bfly<11,8,8,TRIPLE,2>"     # the local products inside the matrix-core encode (kernels_mfma_bfly.hpp)
        step_bytes = dom_bytes + (n * n * G + n * G) * 32 + (n * G + N) * 32 + n * N * 64 + N * 32
        tkey = f"triple_encode_parties_n{n}_t{t}_N{N}"
        limiter = ("HBM on a 67 % read / 33 % write stream of 32-byte elements at a 352-byte stride: the same launch with loads and stores only "
                   "(every load kept live) takes 2.14 ms against 2.17 - 2.25 ms, profiles/r04_ablations_all_loads_live.txt")
    else:
        dom = lambda: eng.dev_fpmul_middle(fp.tc, fp.x, fp.y, fp.dop, fp.eop, fp.rbits, fp.rint, k, f, N, n, fp.z, fp.rdash, fp.osh, stream)
        dom_bytes = n * N * 32 * (7 + f) + 2 * N * 32  # c, x, y, r_int, f bit shares read, z, r', open share written per party; d, e once
        dom_name = "k_fpmul_middle<U29>"
        # the first open reads a, b, x, y of its 2t + 1 senders and forms their shares itself (four launches: hbmpc_dev_fpmul_parties);
        # the second reads the 2t + 1 senders' open shares; d, e, c out; TruncPr's last step z, r' in, the output shares out
        step_bytes = dom_bytes + (2 * t + 1) * 4 * N * 32 + (2 * t + 1) * N * 32 + 3 * N * 32 + n * N * 32 * 3 + N * 32
        tkey = f"fpmul_middle_n{n}_t{t}_N{N}_f{f}"
        limiter = "HBM (element-wise, 5 - 6 TB/s: profiles/r04_cfg5_kernel_stats.csv)"
    for _ in range(3):
        dom()
    torch.cuda.synchronize()
    rec(e0)
    for _ in range(args.steps):
        dom()
    rec(e1)
    torch.cuda.synchronize()
    dom_ms = e0.elapsed_time(e1) / args.steps
    ach = dom_bytes / (dom_ms * 1e-3) / 1e9
    out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                       "kernel": dom_name, "kernel_ms": dom_ms, "algorithmic_bytes": dom_bytes, "limiter": limiter,
                       "kernel_ms_source": "HIP events around K launches of this call alone, after the timed region",
                       "step_algorithmic_bytes": step_bytes,
                       "step_GBps": step_bytes / (tm["kernel_ms"] * 1e-3) / 1e9, "step_frac": step_bytes / (tm["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
    trec = traffic_record(tkey)
    if trec:
        out["roofline"]["traffic"] = trec["hbm_bytes_per_launch"]
        out["roofline"]["traffic_source"] = trec["source"]
    if pre is not None:
        # the producers alone (their share of the step), and their compulsory traffic
        def ev_ms(fn):
            a0, a1, r = ctx["events"]()
            fn()
            torch.cuda.synchronize()
            r(a0)
            for _ in range(args.steps):
                fn()
            r(a1)
            torch.cuda.synchronize()
            return a0.elapsed_time(a1) / args.steps
        Krs, Krd = pre.K_rs, pre.K_rd
        # deal: coefficients in, n^2 K shares out; mix: n^2 K in (the dealt rows, read in place), n^2 K out; verifiers: their
        # sender rows in, polynomials out; output slices in and out
        rs_b = (n * Krs * (t + 1) + 3 * n * n * Krs + 2 * t * ((2 * t + 1) * Krs + Krs * (t + 1)) + 2 * n * (n - 2 * t) * Krs) * 32
        rd_b = (n * Krd * (3 * t + 2) + 6 * n * n * Krd + (n - t - 1) * 4 * n * Krd + 4 * n * (t + 1) * Krd) * 32
        rs_ms, rd_ms = ev_ms(lambda: pre.rs.run(check=False)), ev_ms(lambda: pre.rd.run(check=False))
        out["producers"] = {"ransha_ms": rs_ms, "ransha_algorithmic_bytes": rs_b, "ransha_GBps": rs_b / rs_ms / 1e6,
                            "randousha_ms": rd_ms, "randousha_algorithmic_bytes": rd_b, "randousha_GBps": rd_b / rd_ms / 1e6,
                            "batch_elements_per_dealer": {"ransha": Krs, "randousha": Krd}}
        out["roofline"]["step_algorithmic_bytes"] = step_bytes + rs_b + rd_b
        out["roofline"]["step_GBps"] = (step_bytes + rs_b + rd_b) / (tm["kernel_ms"] * 1e-3) / 1e9
        out["roofline"]["step_frac"] = out["roofline"]["step_GBps"] / HBM_PEAK_GBS
    if world > 1 and not args.no_final_gather:
        from mpc_protocols_amd import sharding
        # this rank's result shares as a torch view of the pipeline's device buffer: [n][N][4]
        mine = torch.empty((n, N, 4), dtype=torch.int64, device=dev)
        eng.d2d(mine.data_ptr(), result_ptr, n * N * 32, stream)
        torch.cuda.synchronize()
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dist.barrier()
        g0.record()
        per_unit = (2 * t + 1) if args.workload == "cfg4" else 1
        full = gather_ragged(sharding, dist, mine, total, per_unit)
        g1.record()
        torch.cuda.synchronize()
        o = lo * per_unit
        assert torch.equal(full[:, o:o + N], mine)
        out["final_gather"] = {"ms": g0.elapsed_time(g1), "bytes_per_rank": n * N * 32,
                               "collective": "all_gather of the [n][N_rank] result shares (RCCL)"}
        del full, mine
    pipe.close()
    if rank == 0 and world == 1:
        out["cpu_baseline"] = cpu_baseline_pipeline(args.workload, n, t)
    return out


def gather_ragged(sharding, dist, mine, total, per_unit):
    """all-gather of party-major shards whose lengths are shard_range(total / per_unit) * per_unit"""
    n = mine.shape[0]
    view = mine.reshape(n, mine.shape[1] // per_unit, per_unit * mine.shape[2])
    full = sharding.gather_party_major(view, total // per_unit)
    return full.reshape(n, total, mine.shape[2])


def extra_measurements(eng, torch, dev, stream):
    """Other rows of the path, timed with HIP events (not part of `value`).  No oracle here: inputs are drawn on the
    device and every check is a property of the path itself (encode -> decode returns the input, corrupted chunks
    are repaired); parity with the oracle is the test suite's job."""
    res = {}

    def ev_time(fn, reps=10, warm=2, warm_seconds=0.15):
        # warm up by TIME as well as by count: these rows run after seconds of CPU-only work (the cpu_baseline legs),
        # i.e. on an idle chip at low clocks, which two launches do not bring back
        t_w = time.perf_counter()
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        while time.perf_counter() - t_w < warm_seconds:
            fn()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gc_on = gc.isenabled()
        gc.disable()  # a full collection (~40 ms with torch imported) inside a 20-launch loop reads as a 4x slower row
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        if gc_on:
            gc.enable()
        return e0.elapsed_time(e1) / reps

    # config 3: n=31, t=10, d=10, 2^20 chunks: encode (apply_vandermonde) and decode (batch_recover_secret)
    n, t, d, G = 31, 10, 10, 1 << 20
    torch.manual_seed(0xC0FFEE02)
    x = _rand_fr(torch, dev, G, d + 1)
    y = torch.empty((n, G, 4), dtype=torch.int64, device=dev)
    co = torch.empty((G, d + 1, 4), dtype=torch.int64, device=dev)
    st = torch.empty((G,), dtype=torch.uint8, device=dev)
    summ = torch.zeros((4,), dtype=torch.int32, device=dev)
    ids = list(range(n))
    ms = ev_time(lambda: eng.dev_vandermonde_apply(x.data_ptr(), G, n, d, y.data_ptr(), stream))
    res["cfg3_encode"] = {"chunks_per_s": G / ms * 1e3, "ms": ms, "GBps_algorithmic": (d + 1 + n) * 32 * G / ms / 1e6,
                          "hbm_frac": (d + 1 + n) * 32 * G / ms / 1e6 / HBM_PEAK_GBS}
    ms = ev_time(lambda: eng.dev_batch_recover(ids, y.data_ptr(), G, n, d, t, co.data_ptr(), 0, st.data_ptr(),
                                               summ.data_ptr(), stream))
    torch.cuda.synchronize()
    assert bool((co == x).all()) and int(st.max()) == 0
    res["cfg3_decode"] = {"recons_per_s": G / ms * 1e3, "secrets_per_s": (d + 1) * G / ms * 1e3, "ms": ms,
                          "GBps_algorithmic": (d + t + 1 + d + 1) * 32 * G / ms / 1e6,
                          "hbm_frac": (d + t + 1 + d + 1) * 32 * G / ms / 1e6 / HBM_PEAK_GBS}
    sec = torch.empty((G, 4), dtype=torch.int64, device=dev)
    ms = ev_time(lambda: eng.dev_batch_recover(ids, y.data_ptr(), G, n, d, t, sec.data_ptr(), 0, st.data_ptr(),
                                               summ.data_ptr(), stream, p0=True))
    res["cfg3_decode_p0"] = {"recons_per_s": G / ms * 1e3, "ms": ms,
                             "GBps_algorithmic": (d + t + 1 + 1) * 32 * G / ms / 1e6}
    # wire codec: the n EvalBatch payloads straight from the party-major encode rows, and back
    stride = 8 + 32 * G + 24
    pay = torch.empty((n * stride,), dtype=torch.uint8, device=dev)
    back = torch.empty_like(y)
    cst = torch.zeros((n,), dtype=torch.int32, device=dev)
    ms = ev_time(lambda: eng.dev_pack_fvec(y.data_ptr(), G, G, n, pay.data_ptr(), stride, stream))
    res["codec_pack_fvec"] = {"ms": ms, "GBps": 2 * 32 * G * n / ms / 1e6, "payloads": n, "elements_each": G}
    ms = ev_time(lambda: eng.dev_unpack_fvec(pay.data_ptr(), stride, 8 + 32 * G, G, n, back.data_ptr(), G, cst.data_ptr(), stream))
    torch.cuda.synchronize()
    assert bool((back == y).all()) and int(cst.abs().max()) == 0
    res["codec_unpack_fvec_validated"] = {"ms": ms, "GBps": 2 * 32 * G * n / ms / 1e6}
    del pay, back
    # the in-place wire path: encode straight into the payload bodies (no pack pass), validate without copying and
    # decode out of the payloads where they arrived (no unpack pass)
    wstride = 32 * (G + 1)
    wire = torch.empty((n * wstride + 64,), dtype=torch.uint8, device=dev)
    pd = (wire.data_ptr() + 31) // 32 * 32 + 24
    ms = ev_time(lambda: eng.dev_encode_fvec(x.data_ptr(), G, n, d, pd, wstride, stream))
    res["wire_encode_in_place"] = {"ms": ms, "note": "vandermonde_apply + pack_fvec in one pass", "payloads": n}
    ms = ev_time(lambda: eng.dev_validate_fvec(pd, wstride, 8 + 32 * G, G, n, cst.data_ptr(), stream))
    torch.cuda.synchronize()
    assert int(cst.abs().max()) == 0
    res["wire_validate_in_place"] = {"ms": ms, "GBps": 32 * G * n / ms / 1e6}
    ms = ev_time(lambda: eng.dev_batch_recover_slots(ids, ids, pd + 8, wstride // 32, G, n, d, t, co.data_ptr(), nco_d=0,
                                                     status_d=st.data_ptr(), summary_d=summ.data_ptr(), stream=stream))
    torch.cuda.synchronize()
    assert bool((co == x).all()) and int(st.max()) == 0
    res["wire_decode_in_place"] = {"ms": ms, "recons_per_s": G / ms * 1e3}
    del wire
    # the same decode with the t lowest-id senders corrupted in 1 % of the chunks (flag + OEC/Gao
    # fallback on the device); results must still be the original polynomials
    gen = torch.Generator(device=dev)
    gen.manual_seed(0xC0FFEE04)
    bad = torch.randperm(G, device=dev, generator=gen)[: G // 100]
    y[:t, bad, 0] ^= 1  # flips the lowest bit of the value: still canonical
    ms = ev_time(lambda: eng.dev_batch_recover(ids, y.data_ptr(), G, n, d, t, co.data_ptr(), 0, st.data_ptr(),
                                               summ.data_ptr(), stream), reps=5, warm=1)
    torch.cuda.synchronize()
    sm = summ.cpu().numpy().view(np.uint32)
    assert bool((co == x).all()) and int(sm[0]) == G // 100 and int(sm[1]) == 0, sm
    res["cfg3_decode_1pct_corrupted"] = {"recons_per_s": G / ms * 1e3, "ms": ms, "fallback_chunks": int(sm[0]),
                                         "note": "t lowest-id senders corrupted in 1 % of chunks"}
    # one Byzantine sender (the lowest id: inside the interpolation set) lies in EVERY chunk: the whole batch fails the
    # optimistic verification; the second-chance candidates resolve it without the OEC/Gao kernel
    y[:t, bad, 0] ^= 1
    y[0, :, 0] ^= 1
    ms = ev_time(lambda: eng.dev_batch_recover(ids, y.data_ptr(), G, n, d, t, co.data_ptr(), 0, st.data_ptr(),
                                               summ.data_ptr(), stream), reps=3, warm=1)
    torch.cuda.synchronize()
    sm = summ.cpu().numpy().view(np.uint32)
    assert bool((co == x).all()) and int(sm[0]) == G and int(sm[1]) == 0, sm
    res["cfg3_decode_byzantine_sender"] = {"recons_per_s": G / ms * 1e3, "ms": ms, "fallback_chunks": int(sm[0]),
                                           "note": "sender 0 lies in every chunk"}
    del x, y, co, sec
    # achievable HBM rate of a plain device copy (report against the vendor peak AND this)
    big = torch.empty((1 << 27,), dtype=torch.int64, device=dev)  # 1 GiB
    dst = torch.empty_like(big)
    ms = ev_time(lambda: dst.copy_(big), reps=5, warm=2)
    res["device_copy_GBps"] = 2 * big.numel() * 8 / ms / 1e6
    del big, dst
    # element-wise: triple_local on 2^22 elements (128 B/element)
    N = 1 << 22
    a, b, c = _rand_fr(torch, dev, N), _rand_fr(torch, dev, N), _rand_fr(torch, dev, N)
    o = torch.empty_like(a)
    ms = ev_time(lambda: eng.dev_elem("triple_local", [a.data_ptr(), b.data_ptr(), c.data_ptr(), o.data_ptr()], N,
                                      stream=stream))
    res["triple_local"] = {"elems_per_s": N / ms * 1e3, "ms": ms, "GBps_algorithmic": 128 * N / ms / 1e6,
                           "hbm_frac": 128 * N / ms / 1e6 / HBM_PEAK_GBS}
    del a, b, c, o
    # seeded dealer (include/hbmpc_hip.h, "hbmpc-chacha20-v1"): config 2 with the coefficients drawn on the device
    n, d, B = 16, 5, 1 << 20
    sec = _rand_fr(torch, dev, B)
    cws = torch.empty((B, d + 1, 4), dtype=torch.int64, device=dev)
    sh = torch.empty((n, B, 4), dtype=torch.int64, device=dev)
    seed = bytes(range(32))
    ms_fill = ev_time(lambda: eng.dev_fill_coeffs(seed, sec.data_ptr(), B, 0, d, cws.data_ptr(), stream))
    ms = ev_time(lambda: eng.dev_compute_shares_seeded(seed, sec.data_ptr(), B, 0, n, d, cws.data_ptr(), sh.data_ptr(), stream))
    res["cfg2_seeded"] = {"share_evals_per_s": n * B / ms * 1e3, "ms": ms, "ms_fill_coeffs": ms_fill,
                          "coeffs_per_s": d * B / ms_fill * 1e3,
                          "note": "secrets in HBM -> ChaCha20 rejection-sampled coefficients -> shares"}
    del sec, cws, sh
    res.update(pipeline_measurements(eng, torch, dev, stream, ev_time))
    res.update(goldilocks_measurements(torch, dev, stream, ev_time))
    # integer-ALU ceiling: register-resident modmul chain (2 modmuls per iteration per lane)
    for impl in ("u29", "sat32"):
        eng.set_impl(impl)
        threads, iters = 256 * 256 * 8, 2000
        buf = torch.empty((threads, 4), dtype=torch.int64, device=dev)
        ms = ev_time(lambda: eng.dev_modmul_ubench(buf.data_ptr(), threads, iters, stream), reps=3, warm=1)
        res[f"modmul_per_s_{impl}"] = threads * (2 * iters + 2) / ms * 1e3
    eng.set_impl("u29")
    return res


def goldilocks_measurements(torch, dev, stream, ev_time):
    """the small-field variants at the shapes of configs 2 and 3 (8-byte elements)."""
    from __graft_entry__ import load_package
    eng = load_package().Engine(dev.index or 0, field="goldilocks")
    res = {}
    gen = torch.Generator(device=dev)
    gen.manual_seed(0xC0FFEE05)

    def rand(*shape):  # canonical: high word < 2^32 - 1
        hi = torch.randint(0, 0xFFFFFFFF, shape, dtype=torch.int64, device=dev, generator=gen)
        lo = torch.randint(0, 1 << 32, shape, dtype=torch.int64, device=dev, generator=gen)
        return (hi << 32) | lo

    for tag, n, t, d, G in (("gl_cfg2_compute_shares", 16, 5, 5, 1 << 20), ("gl_cfg3", 31, 10, 10, 1 << 20)):
        x = rand(G, d + 1)
        y = torch.empty((n, G), dtype=torch.int64, device=dev)
        ms = ev_time(lambda: eng.dev_vandermonde_apply(x.data_ptr(), G, n, d, y.data_ptr(), stream))
        torch.cuda.synchronize()
        by = (d + 1 + n) * 8 * G
        res[tag + ("" if "shares" in tag else "_encode")] = {"share_evals_per_s": n * G / ms * 1e3, "ms": ms,
                                                             "GBps_algorithmic": by / ms / 1e6, "hbm_frac": by / ms / 1e6 / HBM_PEAK_GBS}
        if tag == "gl_cfg3":
            co = torch.empty((G, d + 1), dtype=torch.int64, device=dev)
            st = torch.empty((G,), dtype=torch.uint8, device=dev)
            summ = torch.zeros((4,), dtype=torch.int32, device=dev)
            ms = ev_time(lambda: eng.dev_batch_recover(list(range(n)), y.data_ptr(), G, n, d, t, co.data_ptr(), 0,
                                                       st.data_ptr(), summ.data_ptr(), stream))
            torch.cuda.synchronize()
            assert bool((co == x).all()) and int(st.max()) == 0
            by = (d + t + 1 + d + 1) * 8 * G
            res["gl_cfg3_decode"] = {"recons_per_s": G / ms * 1e3, "ms": ms, "GBps_algorithmic": by / ms / 1e6,
                                     "hbm_frac": by / ms / 1e6 / HBM_PEAK_GBS}
        del x, y
    # config 4's shape in the small field (TripleGenNode over GoldilocksField, PreprocNodesSmallField): all 16 parties
    n, t = 16, 5
    N = ((1 << 22) // (2 * t + 1)) * (2 * t + 1)
    tg = load_package().pipelines.TripleGen(eng, n, t, N, stream)

    def share(secrets, d, out_ptr):
        co = rand(N, d + 1)
        co[:, 0] = secrets
        assert eng.dev_compute_shares(co.data_ptr(), N, n, d, out_ptr, stream) == 0, eng.last_error()
        torch.cuda.synchronize()

    a, b, r = rand(N), rand(N), rand(N)
    share(a, t, tg.a), share(b, t, tg.b), share(r, t, tg.rt), share(r, 2 * t, tg.r2t)
    tg.run(check=True)
    ms = ev_time(lambda: tg.run(check=False), reps=5, warm=1)
    res["gl_cfg4_triple_gen_16_parties"] = {"triples_per_s": N / ms * 1e3, "ms": ms, "triples": N}
    tg.close()
    eng.close()
    # the CPU figure of the small-field rows: the C restatement built for Goldilocks (oracle/cref_gl.py, "port"), one thread,
    # a bounded sample; like every oracle use in this file it is a reported baseline and a checker, never the thing timed above
    from oracle import cref_gl
    smp = 1 << 18
    xs = cref_gl.fill_random(0xC0FFEE05, smp * 6).reshape(smp, 6)
    t0, reps = time.perf_counter(), 0
    while reps < 64 and time.perf_counter() - t0 < 2.0:
        assert cref_gl.compute_shares(xs, 16, 5)[0] == 0
        reps += 1
    dt = time.perf_counter() - t0
    res["gl_cpu_baseline_compute_shares"] = {"value": 16 * smp * reps / dt, "unit": "shares/s", "cores": 1, "kind": "port",
                                             "sample": f"compute_shares n=16 d=5 over Goldilocks: {reps} passes over {smp} secrets, single thread"}
    xr = cref_gl.fill_random(0xC0FFEE06, (1 << 14) * 11).reshape(1 << 14, 11)
    rc, yr = cref_gl.vandermonde_apply(xr, 31, 10)
    t0, reps = time.perf_counter(), 0
    while reps < 64 and time.perf_counter() - t0 < 2.0:
        rc, co, _, st = cref_gl.batch_recover(list(range(31)), yr, 31, 10, 10)
        assert rc == 0
        reps += 1
    dt = time.perf_counter() - t0
    assert np.array_equal(co, xr)
    res["gl_cpu_baseline_batch_recover"] = {"value": (1 << 14) * reps / dt, "unit": "recons/s", "cores": 1, "kind": "port",
                                            "sample": f"batch_recover n=31 d=t=10 over Goldilocks: {reps} passes over 16384 chunks, single thread"}
    return res


def _rand_fr(torch, dev, *shape):
    """uniform-ish canonical field elements generated on the device (limb 3 below r's top limb)"""
    lo = torch.randint(0, 1 << 62, shape + (3,), dtype=torch.int64, device=dev)
    hi = torch.randint(0, 0x73EDA753299D7D48, shape + (1,), dtype=torch.int64, device=dev)
    return torch.cat([lo, hi], dim=-1).contiguous()


def _share_on_device(eng, torch, dev, stream, secrets, n, d, out_ptr):
    """[n][N] degree-d sharings of `secrets` written to out_ptr, by the compute_shares kernel itself"""
    N = secrets.shape[0]
    co = _rand_fr(torch, dev, N, d + 1)
    co[:, 0] = secrets
    rc = eng.dev_compute_shares(co.data_ptr(), N, n, d, out_ptr, stream)
    assert rc == 0, eng.last_error()
    torch.cuda.synchronize()


def setup_fpmul(eng, torch, dev, stream, n, t, N, k, m):
    """an FpMul pipeline object whose device buffers hold VALID degree-t sharings (small fixed-point inputs, a Beaver
    triple, TruncPr's random bits and r_int), produced by the compute_shares kernel itself"""
    from __graft_entry__ import load_package
    fp = load_package().pipelines.FpMul(eng, n, t, N, k, m, stream)
    x = torch.zeros((N, 4), dtype=torch.int64, device=dev)
    y = torch.zeros((N, 4), dtype=torch.int64, device=dev)
    x[:, 0] = torch.randint(0, 1 << 7, (N,), device=dev)
    y[:, 0] = torch.randint(0, 1 << 7, (N,), device=dev)
    ta, tb = _rand_fr(torch, dev, N), _rand_fr(torch, dev, N)
    tc = torch.empty_like(ta)
    torch.cuda.synchronize()
    assert eng.dev_fr_op("mul", ta.data_ptr(), tb.data_ptr(), N, tc.data_ptr(), stream) == 0
    rint = torch.zeros((N, 4), dtype=torch.int64, device=dev)
    rint[:, 0] = torch.randint(0, 1 << 40, (N,), device=dev)
    for sec, ptr in ((x, fp.x), (y, fp.y), (ta, fp.ta), (tb, fp.tb), (tc, fp.tc), (rint, fp.rint)):
        _share_on_device(eng, torch, dev, stream, sec, n, t, ptr)
    tmp = torch.empty((n, N, 4), dtype=torch.int64, device=dev)
    for j in range(m):  # r_bits[party][bit][N]
        bit = torch.zeros((N, 4), dtype=torch.int64, device=dev)
        bit[:, 0] = torch.randint(0, 2, (N,), device=dev)
        _share_on_device(eng, torch, dev, stream, bit, n, t, tmp.data_ptr())
        for p in range(n):
            eng.d2d(fp.rbits + (p * m + j) * N * 32, tmp.data_ptr() + p * N * 32, N * 32, stream)
    torch.cuda.synchronize()
    return fp


def pipeline_measurements(eng, torch, dev, stream, ev_time):
    """BASELINE configs 4 and 5 as device-resident replays of all n parties on ONE GPU (mpc-protocols_amd/pipelines.py)."""
    from __graft_entry__ import load_package
    pl = load_package().pipelines
    res = {}
    # config 4: triple_gen, n=16, t=5, 2^22 triples (rounded down to a multiple of 2t+1)
    n, t = 16, 5
    N = ((1 << 22) // (2 * t + 1)) * (2 * t + 1)
    tg = pl.TripleGen(eng, n, t, N, stream)
    a, b, r = (_rand_fr(torch, dev, N) for _ in range(3))
    _share_on_device(eng, torch, dev, stream, a, n, t, tg.a)
    _share_on_device(eng, torch, dev, stream, b, n, t, tg.b)
    _share_on_device(eng, torch, dev, stream, r, n, t, tg.rt)
    _share_on_device(eng, torch, dev, stream, r, n, 2 * t, tg.r2t)
    tg.run(check=True)  # one checked run: every decode reports zero failures
    ms = ev_time(lambda: tg.run(check=False), reps=5, warm=1)
    tg.capture()  # the same launches recorded once into a HIP graph
    ms_graph = ev_time(tg.replay, reps=5, warm=1)
    res["cfg4_triple_gen_16_parties"] = {"triples_per_s": N / min(ms, ms_graph) * 1e3, "ms": ms, "ms_hip_graph": ms_graph,
                                         "triples": N,
                                         "note": "all 16 simulated parties on one GPU: local mul, encode, 16 P(0) decodes, reveal decode, finalize"}
    tg.close()
    del a, b, r
    # the same at a protocol-sized batch: 1100 and 11000 triples per party (100 / 1000 chunks of 2t+1)
    for groups in (100, 1000):
        Ns = groups * (2 * t + 1)
        tg = pl.TripleGen(eng, n, t, Ns, stream)
        a, b, r = (_rand_fr(torch, dev, Ns) for _ in range(3))
        _share_on_device(eng, torch, dev, stream, a, n, t, tg.a)
        _share_on_device(eng, torch, dev, stream, b, n, t, tg.b)
        _share_on_device(eng, torch, dev, stream, r, n, t, tg.rt)
        _share_on_device(eng, torch, dev, stream, r, n, 2 * t, tg.r2t)
        tg.run(check=True)
        tg.run(check=False)
        ms_eager = ev_time(lambda: tg.run(check=False), reps=20, warm=2)
        tg.capture()
        ms_graph = ev_time(tg.replay, reps=20, warm=2)
        res[f"cfg4_triple_gen_{Ns}_triples"] = {"triples": Ns, "parties": n, "ms_eager": ms_eager, "ms_hip_graph": ms_graph,
                                                "triples_per_s_hip_graph": Ns / ms_graph * 1e3}
        tg.close()
        del a, b, r
    # the producers in front of it at the same protocol-sized batch: run_preprocessing's triple part from the dealers' polynomials
    # (RanSha -> a, b; DouSha + RanDouSha -> r; TripleGen).  Launch-bound: ~100 launches, the verifiers' loops most of them.
    for groups_pre in (100, 4096):  # 4096 triple groups per batch: the reference node's own batch size (honeybadger/mod.rs:106-112)
        Ns = groups_pre * (2 * t + 1)
        pre = pl.Preprocessing(eng, n, t, Ns, stream)
        sec0 = {}
        for ptr, K, deg in ((pre.rs.coeffs, pre.K_rs, t), (pre.rd.coeffs_t, pre.K_rd, t), (pre.rd.coeffs_2t, pre.K_rd, 2 * t)):
            for p in range(n):
                co = _rand_fr(torch, dev, K, deg + 1)
                if ptr == pre.rd.coeffs_2t:
                    co[:, 0] = sec0[p]
                elif ptr == pre.rd.coeffs_t:
                    sec0[p] = co[:, 0].clone()
                eng.d2d(ptr + p * K * (deg + 1) * 32, co.data_ptr(), K * (deg + 1) * 32, stream)
                torch.cuda.synchronize()
        pre.run(check=True)
        ms_eager = ev_time(lambda: pre.run(check=False), reps=10, warm=2)
        pre.capture()
        ms_graph = ev_time(pre.replay, reps=10, warm=2)
        res[f"cfg4_preprocessing_{Ns}_triples"] = {"triples": Ns, "parties": n, "ms_eager": ms_eager, "ms_hip_graph": ms_graph, "triples_per_s": Ns / min(ms_eager, ms_graph) * 1e3,
                                                  "note": "dealers' polynomials -> [c]: RanSha, DouSha + RanDouSha, TripleGen" + ("; launch-bound at this size" if groups_pre == 100 else "; the reference node's batch size (4096 triple groups)")}
        pre.close()
        del sec0
    # config 5: fpmul, n=16, t=5, 2^18 elements, (k, f) = (16, 4)
    N, k, m = 1 << 18, 16, 4
    fp = setup_fpmul(eng, torch, dev, stream, n, t, N, k, m)
    torch.cuda.synchronize()
    fp.run(check=True)                                 # every open reports zero failures (three copy-backs and syncs)
    ms = ev_time(lambda: fp.run(check=False), reps=10, warm=2)   # the pipeline itself: enqueue only
    fp.capture()
    ms_graph = ev_time(fp.replay, reps=10, warm=2)
    res["cfg5_fpmul_16_parties"] = {"fpmuls_per_s": N / min(ms, ms_graph) * 1e3, "ms": ms, "ms_hip_graph": ms_graph, "elements": N, "k": k, "f": m}
    fp.close()
    # the same with f = 16 fractional bits ("and f=16"): 16 bit shares per element and party instead of 4
    fp = setup_fpmul(eng, torch, dev, stream, n, t, N, 32, 16)
    torch.cuda.synchronize()
    fp.run(check=True)
    ms16 = ev_time(lambda: fp.run(check=False), reps=10, warm=2)
    res["cfg5_fpmul_16_parties_f16"] = {"fpmuls_per_s": N / ms16 * 1e3, "ms": ms16, "elements": N, "k": 32, "f": 16,
                                        "algorithmic_GBps": (n * N * 32 * (7 + 16 + 3) + (2 * t + 1) * 5 * N * 32) / (ms16 * 1e-3) / 1e9}
    fp.close()
    # the regime the protocols actually run in: small batches, where the ~110 launches of one fpmul are launch-bound.
    # Eager hbmpc_dev_* calls vs the same sequence captured once into a HIP graph (hbmpc_graph_*) and replayed.
    Ns = 1024
    fp = setup_fpmul(eng, torch, dev, stream, n, t, Ns, k, m)
    fp.run(check=True)
    ms_eager = ev_time(lambda: fp.run(check=False), reps=20, warm=2)
    fp.capture()
    ms_graph = ev_time(fp.replay, reps=20, warm=2)
    fp.close()
    # the same as the five separate launches (one per step) instead of one
    import ctypes as _C
    eng.L.hbmpc_set_fused_fpmul(eng.ctx, _C.c_size_t(0))
    fp = setup_fpmul(eng, torch, dev, stream, n, t, Ns, k, m)
    fp.run(check=True)
    ms_eager5 = ev_time(lambda: fp.run(check=False), reps=20, warm=2)
    fp.capture()
    ms_graph5 = ev_time(fp.replay, reps=20, warm=2)
    fp.close()
    eng.L.hbmpc_set_fused_fpmul(eng.ctx, _C.c_size_t(2048))
    res["cfg5_fpmul_small_batch"] = {"elements": Ns, "parties": n, "ms_eager": ms_eager, "ms_hip_graph": ms_graph,
                                     "fpmuls_per_s_hip_graph": Ns / ms_graph * 1e3, "launches": 1,
                                     "five_launches": {"ms_eager": ms_eager5, "ms_hip_graph": ms_graph5}}
    # in between: 8192 elements -- each of the three opens is ONE matrix-core launch (workgroup per tile) once its sender
    # set has been seen twice
    Nm = 8192
    fp = setup_fpmul(eng, torch, dev, stream, n, t, Nm, k, m)
    fp.run(check=True)
    fp.run(check=False)
    ms_eager = ev_time(lambda: fp.run(check=False), reps=20, warm=2)
    fp.capture()
    ms_graph = ev_time(fp.replay, reps=20, warm=2)
    res["cfg5_fpmul_mid_batch"] = {"elements": Nm, "parties": n, "ms_eager": ms_eager, "ms_hip_graph": ms_graph,
                                   "fpmuls_per_s_hip_graph": Nm / ms_graph * 1e3}
    fp.close()
    return res


if __name__ == "__main__":
    main()
