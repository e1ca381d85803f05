"""The N>1 path on CPU: two gloo ranks shard a level's groups and gather their (fake) match blobs to rank 0."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pangraph_amd.dist import shard_groups, gather_blobs, max_over_ranks, sum_over_ranks


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    mine = shard_groups(7, rank, world)
    blob = b"".join(bytes([g]) * (g + 1) for g in mine)           # variable length per rank
    got = gather_blobs(blob, dev, dst=0)
    import numpy as np
    got_np = gather_blobs(np.frombuffer(blob, dtype=np.uint8), dev, dst=0, as_bytes=False)      # zero-copy input, tensor output (bench.py's path)
    if got_np is not None:
        assert [bytes(t.numpy().tobytes()) for t in got_np] == got
    mx = max_over_ranks(float(rank + 1), dev)
    sm = sum_over_ranks(float(len(mine)), dev)
    q.put((rank, mine, got, mx, sm))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, g0, got0, mx0, sm0), (r1, g1, got1, mx1, sm1) = res
    assert g0 + g1 == list(range(7)) and len(g0) in (3, 4)
    assert got1 is None and len(got0) == 2
    assert got0[0] == b"".join(bytes([g]) * (g + 1) for g in g0)
    assert got0[1] == b"".join(bytes([g]) * (g + 1) for g in g1)
    assert mx0 == mx1 == 2.0 and sm0 == sm1 == 7.0


def test_shard_groups_covers_everything():
    for n in (0, 1, 5, 8, 1000):
        for world in (1, 2, 4, 8):
            allg = [g for r in range(world) for g in shard_groups(n, r, world)]
            assert allg == list(range(n))
            sizes = [len(shard_groups(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
