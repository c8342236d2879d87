"""Runs the C++ restatement of the reference's own unit tests (tests/cpp/test_reference_units.cpp) against
the C++ host mirror include/hbmpc_shares.hpp -- the reference is compiled code, so the host side above the
C ABI exists in C++ as well as in the ctypes mirror the other tests use."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "test_reference_units")
CBIN = os.path.join(ROOT, "tests", "cpp", "test_c_abi")
PBIN = os.path.join(ROOT, "tests", "cpp", "test_pipelines")


def test_cpp_mirror_builds():
    # host-only g++ compile + link against the in-tree library (no GPU needed)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")], stdout=subprocess.DEVNULL)
    assert os.path.exists(BIN) and os.path.exists(CBIN) and os.path.exists(PBIN)


def test_header_is_plain_c99():
    subprocess.check_call(["gcc", "-x", "c", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only",
                           os.path.join(ROOT, "include", "hbmpc_hip.h")])


@pytest.mark.gpu
def test_c99_caller_round_trips():
    if not os.path.exists(CBIN):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")], stdout=subprocess.DEVNULL)
    p = subprocess.run([CBIN], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "C ABI round trips passed" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


@pytest.mark.gpu
def test_reference_unit_tests_in_cpp():
    if not os.path.exists(BIN):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")], stdout=subprocess.DEVNULL)
    p = subprocess.run([BIN], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "all reference unit tests passed" in p.stdout


@pytest.mark.gpu
def test_cpp_pipelines_eager_and_graph():
    if not os.path.exists(PBIN):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")], stdout=subprocess.DEVNULL)
    p = subprocess.run([PBIN], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "pipelines passed" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


@pytest.mark.gpu
def test_pool_allocated_caller_buffers():
    """hipMallocAsync memory for evals / out / ncoeffs / status / summary, every chunk flagged, buffers re-allocated and
    first used after a pause in each of 25 episodes per kernel family (tests/cpp/test_pool_buffers.hip)"""
    bin_ = os.path.join(ROOT, "tests", "cpp", "test_pool_buffers")
    if not os.path.exists(bin_):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")], stdout=subprocess.DEVNULL)
    p = subprocess.run([bin_], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "pool buffers passed" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]
    # the pool was put into the supported configuration by the library's own call (hbmpc_stream_pool_retain), from the platform's default
    assert "release threshold of the device's pool raised from 0:" in p.stdout, p.stdout[:500]


@pytest.mark.gpu
@pytest.mark.xfail(strict=False, reason="UNSUPPORTED configuration (include/hbmpc_hip.h, 'Device buffers'): a hipMallocAsync pool at its default "
                                        "release threshold returns memory to the driver at every synchronisation; on ROCm 7.2 / gfx950 blocks it "
                                        "re-acquires are then read through stale cache lines (every second episode wrong, "
                                        "profiles/r02_pool_buffers_default_threshold.txt); "
                                        "hbmpc_stream_pool_release_threshold reports that configuration and hbmpc_stream_pool_retain repairs it")
def test_pool_allocated_caller_buffers_default_release_threshold():
    """the same episodes with the pool's DEFAULT release threshold: recorded as an expected failure (an XPASS means the
    platform no longer shows the stale lines), so that the outcome is in the test report instead of a print"""
    bin_ = os.path.join(ROOT, "tests", "cpp", "test_pool_buffers")
    if not os.path.exists(bin_):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")], stdout=subprocess.DEVNULL)
    q = subprocess.run([bin_], capture_output=True, text=True, timeout=600, env=dict(os.environ, POOL_DEFAULT_THRESHOLD="1"))
    assert q.returncode == 0, q.stdout.strip().splitlines()[-1] if q.stdout.strip() else q.stderr[-500:]
