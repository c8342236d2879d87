"""Multi-GPU layout of the path: the batch index shards across ranks, every rank runs the
identical kernel sequence on its slice, and the ONLY collective is the optional final gather of the outputs
(BASELINE.json north_star: "RCCL over xGMI used only for the final gather").  One process per GPU;
`torch.distributed` is the transport (backend "nccl" = RCCL on the GPU box, "gloo" in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int):
    """contiguous batch shard [lo, hi) of `rank`; sizes differ by at most one"""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_party_major(local: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """local: this rank's party-major output [n][B_rank][...] for its shard_range(total, rank, world) of the batch.
    Returns [n][total][...] on every rank: party j's row is the concatenation of the ranks' column ranges.
    Ranks may hold different B_rank (ragged shards): shorter ones are padded for the fixed-size all_gather."""
    world = dist.get_world_size(group)
    n = local.shape[0]
    spans = [shard_range(total, r, world) for r in range(world)]
    bmax = max(hi - lo for lo, hi in spans)
    tail = tuple(local.shape[2:])
    send = local
    if local.shape[1] != bmax:
        send = torch.zeros((n, bmax) + tail, dtype=local.dtype, device=local.device)
        send[:, : local.shape[1]] = local
    recv = [torch.empty((n, bmax) + tail, dtype=local.dtype, device=local.device) for _ in range(world)]
    dist.all_gather(recv, send.contiguous(), group=group)  # one collective; RCCL runs it as a ring over xGMI
    out = torch.empty((n, total) + tail, dtype=local.dtype, device=local.device)
    for r, (lo, hi) in enumerate(spans):
        out[:, lo:hi] = recv[r][:, : hi - lo]
    return out
