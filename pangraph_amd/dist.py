"""Multi-GPU plumbing for the one exchange step of the path: the final match-list gather.

Queries are independent given their group, so a level's groups are sharded across ranks with no data-path
collective; each rank then owns a variable-length byte blob (packed pga_match_t records + CIGAR pool) and rank 0
collects them: an all_gather of the blob sizes followed by a padded all_gather of the blobs.  With the `nccl`
backend this is RCCL over xGMI; the payload is ~100 B per alignment, so the step is latency-bound and a ring is
unnecessary.  The same code runs on CPU tensors under `gloo` (tests/test_dist_cpu.py, world_size 2).
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


def gather_blobs(blob: bytes, device: torch.device, dst: int = 0) -> Optional[List[bytes]]:
    """Variable-length gather of one bytes object per rank; returns the list on `dst`, None elsewhere."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    n = torch.tensor([len(blob)], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    buf = torch.zeros(mx, dtype=torch.uint8, device=device)
    if blob:
        buf[: len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    outs = [torch.empty(mx, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(outs, buf)
    if rank != dst:
        return None
    return [bytes(o[:s].cpu().numpy().tobytes()) for o, s in zip(outs, sizes)]


def max_over_ranks(x: float, device: torch.device) -> float:
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x: float, device: torch.device) -> float:
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
