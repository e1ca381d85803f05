"""Multi-GPU plumbing for the one exchange step of the path: the final match-list gather.

Queries are independent given their group, so a level's groups are sharded across ranks with no data-path
collective; each rank then owns a variable-length byte blob (packed pga_match_t records + CIGAR pool) and rank 0
collects them: an all_gather of the blob sizes followed by one send per rank to the owner.  With the `nccl`
backend this is RCCL over xGMI (point-to-point sends to the owning rank; the CIGAR pool dominates the payload).  The same code runs on CPU tensors under `gloo` (tests/test_dist_cpu.py, world_size 2).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_groups(n_groups: int, rank: int, world: int) -> List[int]:
    """Contiguous, balanced ranges: rank r owns groups [r*n/world, (r+1)*n/world)."""
    lo = rank * n_groups // world
    hi = (rank + 1) * n_groups // world
    return list(range(lo, hi))


def gather_blobs(blob, device: torch.device, dst: int = 0, as_bytes: bool = True):
    """Variable-length gather of one byte blob per rank (bytes, or a uint8 numpy array / torch tensor, which is not copied on
    the host); returns the list on `dst` (bytes, or host uint8 tensors with as_bytes=False), None elsewhere.
    Sizes travel by all_gather; the payload goes point to point to `dst` only -- a level's CIGAR pool is hundreds of MB per
    rank, and nobody but the owner of the graph needs it."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    if isinstance(blob, (bytes, bytearray)):
        mine = torch.frombuffer(bytearray(blob), dtype=torch.uint8) if len(blob) else torch.zeros(0, dtype=torch.uint8)
    elif isinstance(blob, torch.Tensor):
        mine = blob
    else:
        mine = torch.from_numpy(blob) if len(blob) else torch.zeros(0, dtype=torch.uint8)
    n = torch.tensor([mine.numel()], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    if rank != dst:
        if sizes[rank] > 0:
            dist.send(mine.to(device), dst)
        return None
    outs = []
    for r in range(world):
        if r == rank:
            outs.append(mine)
        elif sizes[r] > 0:
            t = torch.empty(sizes[r], dtype=torch.uint8, device=device)
            dist.recv(t, r)
            outs.append(t.cpu())
        else:
            outs.append(torch.zeros(0, dtype=torch.uint8))
    if as_bytes:
        return [bytes(o.cpu().numpy().tobytes()) for o in outs]
    return outs


def max_over_ranks(x: float, device: torch.device) -> float:
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x: float, device: torch.device) -> float:
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
