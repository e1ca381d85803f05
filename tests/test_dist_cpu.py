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


# ---- the match-list exchange proper: records of a sharded wave == records of one rank ----
def _fake_wave(n_groups, seed=3):
    """per group a list of (record fields, cigar ops) as one rank would emit them"""
    import numpy as np
    from pangraph_amd.dist import MATCH_DTYPE
    rng = np.random.default_rng(seed)
    out = []
    for g in range(n_groups):
        recs = []
        for q in range(int(rng.integers(0, 4))):
            for _ in range(int(rng.integers(0, 3))):
                recs.append((q, int(rng.integers(0, 9)), rng.integers(1, 1000, size=int(rng.integers(1, 6))).astype(np.uint32)))
        out.append(recs)
    return out


def _pack(groups, ids):
    """packed pga_match_t[] + CIGAR pool of the groups `ids` (local group index = position in ids)"""
    import numpy as np
    from pangraph_amd.dist import MATCH_DTYPE
    m, pool = [], []
    off = 0
    for lg, g in enumerate(ids):
        for q, r, cg in groups[g]:
            rec = np.zeros(1, MATCH_DTYPE)
            rec["group"], rec["qry"], rec["ref"], rec["cigar_off"], rec["n_cigar"], rec["matches"] = lg, q, r, off, len(cg), 1000 * g + q
            m.append(rec); pool.append(cg); off += len(cg)
    return (np.concatenate(m) if m else np.zeros(0, MATCH_DTYPE)), (np.concatenate(pool) if pool else np.zeros(0, np.uint32))


def _gather_worker(rank, world, port, q):
    import numpy as np
    from pangraph_amd.dist import gather_matches, shard_groups_balanced
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    groups = _fake_wave(11)
    weights = [1 + sum(len(c) for _, _, c in g) for g in groups]
    plan = shard_groups_balanced(weights, world)
    m, c = _pack(groups, plan[rank])
    got = gather_matches(m.view(np.uint8), c.view(np.uint8), plan[rank], plan, torch.device("cpu"), dst=0)
    if rank == 0:
        rec, pool = got
        want_m, want_c = _pack(groups, list(range(len(groups))))
        ok = len(rec) == len(want_m)
        for a, b in zip(rec, want_m):
            ok &= all(a[f] == b[f] for f in ("group", "qry", "ref", "n_cigar", "matches"))
            ok &= (pool[int(a["cigar_off"]):int(a["cigar_off"]) + int(a["n_cigar"])] == want_c[int(b["cigar_off"]):int(b["cigar_off"]) + int(b["n_cigar"])]).all()
        q.put(bool(ok) and len(rec) > 5)
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_wave_gather_equals_single_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    ps = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    assert q.get(timeout=120) is True
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0


def test_shard_groups_balanced():
    from pangraph_amd.dist import shard_groups_balanced
    w = [10, 1, 1, 1, 9, 2, 2, 8, 3, 3]
    for world in (1, 2, 3, 8):
        plan = shard_groups_balanced(w, world)
        assert sorted(g for p in plan for g in p) == list(range(len(w)))
        assert all(p == sorted(p) for p in plan)
        loads = [sum(w[g] for g in p) for p in plan]
        assert max(loads) <= sum(w) / world + max(w)


def test_native_merge_and_sharding_equal_the_python_host(product_so):
    """pga_merge_match_lists / pga_shard_groups_balanced (include/pga_align.h: what a host that is not Python calls either side of its transport)
    against dist.merge_match_lists / shard_groups_balanced: whole groups per rank, the queries of single groups split over ranks
    (pga_batch_align_shard), empty parts, records that already carry global ids."""
    import numpy as np
    import pytest
    from pangraph_amd import dist as pd
    for seed in (3, 4, 5):
        groups = _fake_wave(23, seed)
        weights = [1 + sum(len(c) for _, _, c in g) for g in groups]
        for world in (1, 2, 3, 8):
            plan = pd.shard_groups_balanced(weights, world)
            assert pd.native_shard_groups_balanced(weights, world) == plan
            parts = [_pack(groups, ids) for ids in plan]
            a = pd.merge_match_lists([m for m, _ in parts], [c for _, c in parts], plan)
            b = pd.native_merge_match_lists([m.view(np.uint8) for m, _ in parts], [c.view(np.uint8) for _, c in parts], plan)
            assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes() and (world == 1 or len(a[0]) > 5)
            want_m, want_c = _pack(groups, list(range(len(groups))))
            assert [int(x) for x in b[0]["group"]] == [int(x) for x in want_m["group"]] and [int(x) for x in b[0]["qry"]] == [int(x) for x in want_m["qry"]]
        # one group over all ranks: rank r holds the records of the queries q with q % world == r, global ids already in place
        full_m, full_c = _pack(groups, list(range(len(groups))))
        for world in (2, 3):
            parts = []
            for r in range(world):
                sel = full_m[full_m["qry"] % world == r].copy()
                pool, off = [], 0
                for rec in sel:
                    cg = full_c[int(rec["cigar_off"]):int(rec["cigar_off"]) + int(rec["n_cigar"])]
                    rec["cigar_off"] = off; pool.append(cg); off += len(cg)
                parts.append((sel, np.concatenate(pool) if pool else np.zeros(0, np.uint32)))
            ident = [list(range(len(groups)))] * world
            a = pd.merge_match_lists([m for m, _ in parts], [c for _, c in parts], ident)
            b = pd.native_merge_match_lists([m for m, _ in parts], [c for _, c in parts], None)
            assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes()
            for x, y in zip(b[0], full_m):
                assert all(x[f] == y[f] for f in ("group", "qry", "ref", "n_cigar", "matches"))
                assert (b[1][int(x["cigar_off"]):int(x["cigar_off"]) + int(x["n_cigar"])] == full_c[int(y["cigar_off"]):int(y["cigar_off"]) + int(y["n_cigar"])]).all()
    m, c = _pack(_fake_wave(5), [0, 1, 2, 3, 4])
    with pytest.raises(ValueError, match="outside its table"):
        pd.native_merge_match_lists([m], [c], [[7, 8]])
    e = pd.native_merge_match_lists([], [], [])
    assert len(e[0]) == 0 and len(e[1]) == 0
