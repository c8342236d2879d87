"""mpc-protocols_amd: MI355X-native Shamir share arithmetic for HoneyBadgerMPC (hot path only).

csrc/   hand-written HIP kernels for gfx950 + the C ABI of include/hbmpc_hip.h  -> libhbmpc_hip.so
hbmpc.py  ctypes binding used by tests/ and bench.py (plumbing; every call goes through the C ABI)

The directory name carries a hyphen (it is the reference's repository name), so import it through
`__graft_entry__.load_package()` which registers it as the module `mpc_protocols_amd`.
"""
from . import hbmpc, pipelines  # noqa: F401
# `sharding` (torch.distributed helpers of the multi-GPU layout) is imported on demand: it needs torch
from .hbmpc import Engine, HbmpcError, build, lib  # noqa: F401
