"""Multi-GPU plumbing for the one exchange step of the path: the match-list gather.

Queries are independent given their group, so the groups of a wave (one tree level, one self-merge round) are sharded across
ranks with no data-path collective -- balanced by their base counts (`shard_groups_balanced`) -- and each rank ends up with a
variable-length list of packed `pga_match_t` records plus a CIGAR pool.  Rank 0 (the owner of the graph) collects them:
an all_gather of the two blob sizes, then one point-to-point send per rank to the owner (`gather_blobs`; with the `nccl`
backend this is RCCL over xGMI; the CIGAR pool dominates the payload and nobody but the owner needs it).  `gather_matches`
puts the pieces together: group ids become global again and CIGAR offsets point into the concatenated pool, records ordered by
(group, query, the aligner's own order inside a query) -- the order a single rank would have produced.
The same code runs on CPU tensors under `gloo` (tests/test_dist_cpu.py, world_size 2).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

# include/pga_align.h: pga_match_t (88 bytes)
MATCH_DTYPE = np.dtype([("group", "<i4"), ("qry", "<i4"), ("ref", "<i4"), ("qry_len", "<i4"), ("qry_start", "<i4"), ("qry_end", "<i4"),
                        ("ref_len", "<i4"), ("ref_start", "<i4"), ("ref_end", "<i4"), ("matches", "<i4"), ("length", "<i4"), ("quality", "<i4"),
                        ("reverse", "<i4"), ("align", "<i4"), ("n_ambi", "<i4"), ("inv", "<i4"), ("divergence", "<f8"), ("cigar_off", "<u8"),
                        ("n_cigar", "<u4"), ("pad", "<u4")])
assert MATCH_DTYPE.itemsize == 88


def shard_groups(n_groups: int, rank: int, world: int) -> List[int]:
    """Contiguous, balanced ranges: rank r owns groups [r*n/world, (r+1)*n/world)."""
    lo = rank * n_groups // world
    hi = (rank + 1) * n_groups // world
    return list(range(lo, hi))


def shard_groups_balanced(weights: Sequence[float], world: int) -> List[List[int]]:
    """Groups dealt to `world` ranks, heaviest first, each to the currently lightest rank (LPT); every rank's list is in
    ascending group order.  Deterministic, so every rank computes the same plan without talking."""
    order = sorted(range(len(weights)), key=lambda g: (-weights[g], g))
    load = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for g in order:
        r = min(range(world), key=lambda i: (load[i], i))
        out[r].append(g)
        load[r] += weights[g]
    for v in out:
        v.sort()
    return out


def gather_blobs(blob, device: torch.device, dst: int = 0, as_bytes: bool = True):
    """Variable-length gather of one byte blob per rank (bytes, or a uint8 numpy array / torch tensor, which is not copied on
    the host); returns the list on `dst` (bytes, or host uint8 tensors with as_bytes=False), None elsewhere.
    Sizes travel by all_gather; the payload goes point to point to `dst` only."""
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


def merge_match_lists(parts_m: Sequence[np.ndarray], parts_c: Sequence[np.ndarray], local_to_global: Sequence[Sequence[int]]):
    """parts_m[r]: the pga_match_t records of rank r (uint8 or structured), parts_c[r]: its CIGAR pool (uint8 or uint32),
    local_to_global[r][g]: the global index of rank r's local group g.  Returns (records, cigar pool) of the whole wave as one
    rank would have produced them."""
    recs, pools, base = [], [], 0
    for m, c, l2g in zip(parts_m, parts_c, local_to_global):
        m = np.asarray(m)
        m = m.view(MATCH_DTYPE).copy() if m.dtype != MATCH_DTYPE else m.copy()
        c = np.asarray(c)
        c = c.view(np.uint32) if c.dtype != np.uint32 else c
        if len(m):
            m["group"] = np.asarray(l2g, dtype=np.int32)[m["group"]]
            m["cigar_off"] += np.uint64(base)
        recs.append(m)
        pools.append(c)
        base += len(c)
    rec = np.concatenate(recs) if recs else np.zeros(0, MATCH_DTYPE)
    pool = np.concatenate(pools) if pools else np.zeros(0, np.uint32)
    # a rank's records are already in (group, query, own order) order: a stable sort by (group, query) restores the single-rank order,
    # also when the queries of ONE group were split over the ranks (pga_batch_align_shard)
    key = rec["group"].astype(np.int64) << 32 | rec["qry"].astype(np.int64)
    rec = rec[np.argsort(key, kind="stable")]
    return rec, pool


def gather_matches(raw_matches, raw_cigars, my_groups: Sequence[int], plan: Sequence[Sequence[int]], device: torch.device, dst: int = 0):
    """The exchange step of a wave.  Every rank passes its packed records / CIGAR pool (uint8 views of the native result) and the
    sharding plan; `dst` gets (records, pool) with global group ids, the others None."""
    pm = gather_blobs(raw_matches, device, dst=dst, as_bytes=False)
    pc = gather_blobs(raw_cigars, device, dst=dst, as_bytes=False)
    if pm is None:
        return None
    return merge_match_lists([t.numpy() for t in pm], [t.numpy() for t in pc], plan)


def max_over_ranks(x: float, device: torch.device) -> float:
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x: float, device: torch.device) -> float:
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


# ---- the same host logic behind the C-ABI (include/pga_align.h: what a host that is not Python calls either side of its own transport) ----
def native_merge_match_lists(parts_m: Sequence[np.ndarray], parts_c: Sequence[np.ndarray], local_to_global: Optional[Sequence[Optional[Sequence[int]]]] = None):
    """merge_match_lists through pga_merge_match_lists"""
    import ctypes as C
    from .batch import lib, pga_match_t
    d = lib()
    d.pga_merge_match_lists.restype = C.c_int
    n = len(parts_m)
    ms = [np.ascontiguousarray(np.asarray(m).view(np.uint8)) for m in parts_m]
    cs = [np.ascontiguousarray(np.asarray(c).view(np.uint8)).view(np.uint32) if len(c) else np.zeros(0, np.uint32) for c in parts_c]
    nm = np.asarray([len(m) // MATCH_DTYPE.itemsize for m in ms], dtype=np.int64)
    nc = np.asarray([len(c) for c in cs], dtype=np.int64)
    mp = (C.c_void_p * max(1, n))(*[m.ctypes.data if len(m) else None for m in ms])
    cp = (C.c_void_p * max(1, n))(*[c.ctypes.data if len(c) else None for c in cs])
    tabs, lp, nl = [], None, None
    if local_to_global is not None:
        tabs = [None if t is None else np.ascontiguousarray(t, dtype=np.int32) for t in local_to_global]
        lp = (C.c_void_p * max(1, n))(*[None if t is None or not len(t) else t.ctypes.data for t in tabs])
        nl = np.asarray([0 if t is None else len(t) for t in tabs], dtype=np.int32)
    out_m = np.zeros(int(nm.sum()), MATCH_DTYPE)
    out_c = np.zeros(int(nc.sum()), np.uint32)
    rc = d.pga_merge_match_lists(C.c_int32(n), mp, nm.ctypes.data_as(C.c_void_p), cp, nc.ctypes.data_as(C.c_void_p), lp,
                                 None if nl is None else nl.ctypes.data_as(C.c_void_p), out_m.ctypes.data_as(C.c_void_p), out_c.ctypes.data_as(C.c_void_p))
    if rc != 0:
        d.pga_sched_error.restype = C.c_char_p
        raise ValueError(d.pga_sched_error().decode())
    return out_m, out_c


def native_shard_groups_balanced(weights: Sequence[float], world: int) -> List[List[int]]:
    """shard_groups_balanced through pga_shard_groups_balanced"""
    import ctypes as C
    from .batch import lib
    w = np.ascontiguousarray(weights, dtype=np.float64)
    owner = np.zeros(max(1, len(w)), dtype=np.int32)
    d = lib()
    d.pga_shard_groups_balanced.restype = None
    d.pga_shard_groups_balanced(C.c_int32(len(w)), w.ctypes.data_as(C.c_void_p), C.c_int32(world), owner.ctypes.data_as(C.c_void_p))
    out: List[List[int]] = [[] for _ in range(world)]
    for g in range(len(w)):
        out[int(owner[g])].append(g)
    return out
