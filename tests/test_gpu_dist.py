"""The N > 1 path end to end on ONE device: two gloo ranks shard the groups of a wave (balanced plan), align their shards on
GPU 0 and gather the match lists to rank 0, which checks them record for record against its own single-rank run of the whole
wave (what bench.py --gpus N does per wave, with RCCL in place of gloo)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _records(res):
    from pangraph_amd.dist import MATCH_DTYPE
    m = np.array(res.raw_matches, copy=True).view(MATCH_DTYPE)
    c = np.array(res.raw_cigars, copy=True).view(np.uint32)
    return m, c


def _worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from pangraph_amd import batch
        from pangraph_amd.dist import gather_matches, shard_groups_balanced
        from pangraph_amd.levels import Population, Rates
        batch.set_device(0)
        waves = Population(5, 8, 150_000, Rates(ev_min=300, ev_max=6000)).build_waves()
        ok, n_rec = True, 0
        for label, groups, names in waves[:4]:
            plan = shard_groups_balanced([sum(len(s) for s in g) for g in groups], world)
            ids = plan[rank]
            z = np.zeros(0, np.uint8)
            res = batch.ResidentBatch(batch.PreparedBatch([groups[i] for i in ids], [names[i] for i in ids])).align(sensitivity=10, want_raw=True) if ids else None
            got = gather_matches(res.raw_matches if res else z, res.raw_cigars if res else z, ids, plan, torch.device("cpu"), dst=0)
            if rank == 0:
                full = batch.ResidentBatch(batch.PreparedBatch(groups, names)).align(sensitivity=10, want_raw=True)
                wm, wc = _records(full)
                gm, gc = got
                ok &= len(gm) == len(wm)
                for a, b in zip(gm, wm):
                    same = all(a[f] == b[f] for f in a.dtype.names if f not in ("cigar_off", "pad"))
                    same &= bool((gc[int(a["cigar_off"]):int(a["cigar_off"]) + int(a["n_cigar"])] == wc[int(b["cigar_off"]):int(b["cigar_off"]) + int(b["n_cigar"])]).all())
                    ok &= bool(same)
                n_rec += len(wm)
        if rank == 0:
            q.put((bool(ok), n_rec))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        if rank == 0:
            q.put((False, repr(e)))
        raise


def test_two_ranks_one_device_gather_equals_single_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    ok, n = q.get(timeout=600)
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ok is True, n
    assert n > 20
