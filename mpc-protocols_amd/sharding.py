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


def gather_shards(local: torch.Tensor, total: int, group=None):
    """ONE collective into ONE receive buffer: returns (view, spans) with view = [n][world][bmax][...] -- party j's row as
    `world` column blocks, block r holding rank r's shard_range (its first spans[r][1] - spans[r][0] columns; ragged shards are
    padded to the longest for the fixed-size collective).  A permuted VIEW of the receive buffer: nothing is copied after the
    all_gather_into_tensor, which RCCL runs as a ring over xGMI."""
    world = dist.get_world_size(group)
    n = local.shape[0]
    spans = [shard_range(total, r, world) for r in range(world)]
    bmax = max(hi - lo for lo, hi in spans)
    tail = tuple(local.shape[2:])
    send = local
    if local.shape[1] != bmax:
        send = torch.zeros((n, bmax) + tail, dtype=local.dtype, device=local.device)
        send[:, : local.shape[1]] = local
    # the concatenated form [world * n][bmax] (what every backend accepts; gloo rejects the stacked one), read as [world][n][bmax]
    recv = torch.empty((world * n, bmax) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    recv = recv.view((world, n, bmax) + tail)
    return recv.permute(1, 0, 2, *range(3, recv.dim())), spans


def gather_party_major(local: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """local: this rank's party-major output [n][B_rank][...] for its shard_range(total, rank, world) of the batch.
    Returns the dense [n][total][...] on every rank: party j's row is the concatenation of the ranks' column ranges.
    gather_shards + ONE strided copy when the shards are equal (total divisible by world: BASELINE configs 2 and 3 always),
    one copy per rank range when they are ragged.  A consumer that can address (rank, column) takes gather_shards' view
    and skips the copy and the second full-size buffer."""
    view, spans = gather_shards(local, total, group)
    n, world, bmax = view.shape[:3]
    tail = tuple(view.shape[3:])
    if all(hi - lo == bmax for lo, hi in spans):
        return view.reshape((n, total) + tail)  # the permuted view does not flatten for free: this is the one copy
    out = torch.empty((n, total) + tail, dtype=local.dtype, device=local.device)
    for r, (lo, hi) in enumerate(spans):
        out[:, lo:hi] = view[:, r, : hi - lo]
    return out
