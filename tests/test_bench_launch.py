"""bench.py --gpus N started WITHOUT a launcher ("Next round" item 2): before anything touches the GPU it
starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process, relays rank 0's JSON line and
exits with the child's code.  No GPU here: the command line is asserted, and a real spawn is followed as far as the
ranks' own "no GPU" exit."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_child_command_line():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "cfg4", "--print-launch"],
                         capture_output=True, text=True, env=_env(), timeout=120)
    assert out.returncode == 0, out.stderr
    cmd = json.loads(out.stdout)["launch"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 0 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(BENCH)
    # the ranks get exactly the arguments this process got (minus the test switch)
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "cfg4"]


def test_one_gpu_needs_no_launcher():
    # N = 1 (the default) must not spawn anything: without a GPU it ends in bench.py's own message
    out = subprocess.run([sys.executable, BENCH, "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=_env(), timeout=300)
    assert out.returncode != 0 and "needs an MI355X" in out.stderr and "torch.distributed" not in out.stderr


def test_spawns_the_ranks_and_returns_their_exit_code():
    # a real spawn: the ranks start, find no GPU and exit non-zero (the launcher stops the others as soon as one has failed,
    # so one message is guaranteed, not two); the parent relays their stderr and their exit code
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                         env=_env(), timeout=600)
    assert out.returncode != 0
    assert out.stderr.count("needs an MI355X") >= 1, out.stderr[-2000:]
    assert out.stdout.strip() == ""  # no bench record


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["cfg2+cfg3", "cfg4"])
def test_two_ranks_on_one_gpu_report_what_the_collective_layer_saw(workload):
    """The N > 1 path end to end on the one GPU of the test box: `bench.py --gpus 2 --same-device --dist-backend gloo` starts its
    own launcher, both ranks run their shard on GPU 0, barrier + max-reduce timing, the final gather, and rank 0's line says
    how many ranks the collective layer really saw (an all-reduce of ones) and which backend carried it."""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--same-device", "--dist-backend", "gloo", "--steps", "2", "--warmup", "1",
                          "--log2-batch", "14", "--no-extra", "--prewarm-seconds", "0.05", "--workload", workload],
                         capture_output=True, text=True, env=_env(), timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["backend"] == "gloo"
    assert "rehearsal" in rec and rec["final_gather"]["ms"] > 0
