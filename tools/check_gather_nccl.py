"""RCCL smoke of the final gather on whatever GPUs this box has (launch with torch.distributed.run, one rank per GPU):
   python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29531 tools/check_gather_nccl.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from __graft_entry__ import load_package
load_package()
from mpc_protocols_amd import sharding
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", device_id=dev)
n, total = 16, 1000 * world + 3
lo, hi = sharding.shard_range(total, rank, world)
full = torch.arange(n * total * 4, dtype=torch.int64, device=dev).reshape(n, total, 4)
got = sharding.gather_party_major(full[:, lo:hi].contiguous(), total)
torch.cuda.synchronize()
assert torch.equal(got, full)
if rank == 0:
    print(f"final gather over {world} rank(s): ok")
dist.destroy_process_group()
