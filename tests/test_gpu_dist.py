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


def _worker_c4(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from pangraph_amd import batch
        from pangraph_amd.dist import gather_matches, shard_groups_balanced
        from pangraph_amd.levels import Population
        from levels_util import digest, records_to_lists
        from util import load_golden
        batch.set_device(0)
        e = load_golden("builds_expected.json.gz")["c4"]
        p = e["params"]
        waves = Population(p["seed"], p["n"], p["length"]).build_waves()
        bad, n_rec, n_groups = [], 0, 0
        if len(waves) != len(e["waves"]):
            bad.append(("waves", len(waves), len(e["waves"])))
        for (label, groups, names), ew in zip(waves, e["waves"]):
            plan = shard_groups_balanced([sum(len(s) for s in g) for g in groups], world)
            ids = plan[rank]
            z = np.zeros(0, np.uint8)
            res = batch.ResidentBatch(batch.PreparedBatch([groups[i] for i in ids], [names[i] for i in ids])).align(sensitivity=10, want_raw=True) if ids else None
            got = gather_matches(res.raw_matches if res else z, res.raw_cigars if res else z, ids, plan, torch.device("cpu"), dst=0)
            if rank == 0:
                rows = records_to_lists(got[0], got[1], names)
                for g, (r, x) in enumerate(zip(rows, ew["groups"])):
                    if (len(r), digest(r)) != (x["n"], x["sha256"]):
                        bad.append((label, g, len(r), x["n"]))
                n_rec += sum(len(r) for r in rows)
                n_groups += len(groups)
        if rank == 0:
            q.put((not bad, (bad[:5], n_groups, n_rec)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        if rank == 0:
            q.put((False, repr(e)))
        raise


def test_c4_every_wave_two_ranks_gather_vs_reference_digests():
    """config C4 ("klebs-like": 16 x 5.3 Mbp, seed 3; BASELINE.json configs[3]): every wave of the build sharded over TWO ranks (gloo) that
    share the one device of the box, match lists gathered to rank 0 (pangraph_amd.dist.gather_matches, the exchange step of SURVEY 8e) and
    digested per group against what the compiled reference returned in the build container (tests/golden/make_golden_builds.py)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000
    ps = [ctx.Process(target=_worker_c4, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    ok, info = q.get(timeout=900)
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ok is True, info
    assert info[1] == 30 and info[2] > 1000, info


def _worker_split(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from pangraph_amd import batch
        from pangraph_amd.dist import gather_matches
        from pangraph_amd.levels import Population, Rates
        batch.set_device(0)
        waves = Population(9, 6, 200_000, Rates(ev_min=300, ev_max=6000)).build_waves()
        ok, n_rec, n_mine = True, 0, 0
        for label, groups, names in (waves[0], waves[-2], waves[-1]):          # leaf pairs, and the root merge: ONE group, both self-merge rounds
            ids = list(range(len(groups)))
            rb = batch.ResidentBatch(batch.PreparedBatch(groups, names))
            res = rb.align(sensitivity=10, want_raw=True, shard=(rank, world))  # every rank: all groups, its share of the queries of each
            mine = _records(res)[0]
            n_mine += len(mine)
            got = gather_matches(res.raw_matches, res.raw_cigars, ids, [ids] * world, torch.device("cpu"), dst=0)
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
            ok &= 0 < n_mine < n_rec                                               # the work really was split
            q.put((bool(ok), (n_rec, n_mine)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        if rank == 0:
            q.put((False, repr(e)))
        raise


def test_one_group_split_over_two_ranks_equals_single_rank():
    """SURVEY 8e / align_with_minimap2_lib.rs:64-74: waves with fewer groups than ranks -- every rank indexes the whole group and maps a
    contiguous range of its queries (pga_batch_align_shard); the gathered list is the single-rank list, record for record."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 37500 + os.getpid() % 2000
    ps = [ctx.Process(target=_worker_split, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    ok, n = q.get(timeout=600)
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ok is True, n
    assert n[0] > 20


def _worker_nccl(q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(39500 + os.getpid() % 2000)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        from pangraph_amd.dist import gather_blobs, gather_matches, max_over_ranks, sum_over_ranks, MATCH_DTYPE
        blob = np.arange(1000, dtype=np.uint8)
        got = gather_blobs(blob, dev, dst=0)
        ok = got == [blob.tobytes()]
        ok &= max_over_ranks(2.5, dev) == 2.5 and sum_over_ranks(4.0, dev) == 4.0
        m = np.zeros(3, MATCH_DTYPE); m["group"] = [0, 1, 1]; m["qry"] = [2, 1, 0]; m["n_cigar"] = 1; m["cigar_off"] = [0, 1, 2]
        rec, pool = gather_matches(m.view(np.uint8), np.arange(3, dtype=np.uint32).view(np.uint8), [0, 1], [[0, 1]], dev, dst=0)
        ok &= list(rec["qry"]) == [2, 0, 1] and len(pool) == 3
        dist.barrier()
        dist.destroy_process_group()
        q.put((bool(ok), ""))
    except Exception as e:  # noqa: BLE001
        q.put((False, repr(e)))
        raise


def test_nccl_backend_runs_the_gather_on_one_gpu():
    """bench.py --gpus N uses the `nccl` backend (RCCL): at least once, on the one GPU of the test box, the process group comes up and the
    collectives of pangraph_amd.dist (all_gather of sizes, all_reduce) run through it"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_nccl, args=(q,))
    p.start()
    ok, msg = q.get(timeout=300)
    p.join(timeout=120)
    assert ok is True, msg


def test_bench_two_ranks_on_one_device_gathers_every_match():
    """bench.py --gpus 2 (ranks over gloo, both on GPU 0): subtrees on their ranks, the merges above the cut by both ranks together with the
    queries of every group split -- the gathered match count equals the single-rank build's"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PGA_BENCH_SINGLE_DEVICE="1")
    common = ["--genomes", "12", "--length", "150000", "--steps", "1", "--warmup", "0", "--cpu-budget", "0", "--no-next-rows"]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + common, env=env, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"] + common, env=env, capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-2000:]
    a = json.loads(one.stdout.strip().splitlines()[-1]); b = json.loads(two.stdout.strip().splitlines()[-1])
    assert a["n_matches_gathered"] == b["n_matches_gathered"] > 50
    assert b["n_gpus"] == 2 and abs(a["config"]["genomes"] - b["config"]["genomes"]) == 0


def test_bench_c5_two_ranks_on_one_device_every_call_vs_reference_digests():
    """bench.py's own multi-rank path at the BASELINE configuration: two ranks (gloo) on GPU 0 -- subtrees per rank under the ready-set schedule,
    one gather, the merges above the cut by both ranks together (queries split, pga_batch_align_shard), one more gather -- and bench.py digests
    the gathered list of the step it timed: all 1998 calls against the compiled reference's digests, or it prints no value (exit 3)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PGA_BENCH_SINGLE_DEVICE="1")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--cpu-budget", "0", "--no-next-rows",
           "--detail", os.path.join(root, "gpurun_out", "bench_detail_2ranks.json")]
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    two = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert two.returncode == 0, (two.stdout[-1500:], two.stderr[-2000:])
    line = two.stdout.strip().splitlines()[-1]
    assert len(line) < 4096
    b = json.loads(line)
    assert b["n_gpus"] == 2 and b["parity_checked_calls"] == 1998 and b["n_matches_gathered"] == 104928, b


def test_bench_eight_ranks_on_one_device_equals_the_single_rank_build():
    """N = 8 without an 8-GPU node: eight gloo ranks share GPU 0 (PGA_BENCH_SINGLE_DEVICE=1, an eighth of the memory each) on a tree too small for
    eight heavy subtrees -- ranks that own little or nothing in phase 1, most merges above the cut and mapped by all ranks together (queries of
    every group split eight ways), both gathers -- and the digest of the whole build's record lists equals the single-rank build's."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PGA_BENCH_SINGLE_DEVICE="1")
    common = ["--genomes", "40", "--length", "60000", "--steps", "1", "--warmup", "0", "--cpu-budget", "0", "--no-next-rows", "--no-resident-rate"]
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    outs = {}
    for n in (1, 8):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--detail", os.path.join(root, "gpurun_out", f"bench_detail_{n}ranks_small.json")] + common,
                           env=env, capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, (n, r.stdout[-1500:], r.stderr[-2500:])
        outs[n] = json.loads(r.stdout.strip().splitlines()[-1])
    a, b = outs[1], outs[8]
    assert b["n_gpus"] == 8 and a["n_matches_gathered"] == b["n_matches_gathered"] > 100
    assert a["build_sha256"] and a["build_sha256"] == b["build_sha256"]


def test_bench_c5_eight_ranks_on_one_device_every_call_vs_reference_digests_and_phases():
    """N = 8 at the BASELINE configuration without an 8-GPU node: eight gloo ranks share GPU 0 and the box's 16 usable cores (two per rank, what
    an 8-GPU node with this grant gives a rank; four helper threads per batch).  Phase 1 (125 leaf groups and the subtrees above them per rank),
    the gather, phase 2 (the merges above the cut, queries split eight ways), the second gather; bench.py digests the gathered list -- all 1998
    calls against the compiled reference's digests, or it prints no value -- and writes the phases of rank 0 beside schedule.predict_scaling's
    (`phases_rank0` in the detail file: on ONE device the measured times say how the phases relate, not how fast eight devices are)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PGA_BENCH_SINGLE_DEVICE="1")
    det = os.path.join(root, "gpurun_out", "bench_detail_8ranks_c5.json")
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0", "--cpu-budget", "0", "--no-next-rows", "--detail", det]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2000:])
    b = json.loads(r.stdout.strip().splitlines()[-1])
    assert b["n_gpus"] == 8 and b["parity_checked_calls"] == 1998 and b["n_matches_gathered"] == 104928, b
    ph = json.load(open(det))["phases_rank0"]
    print("phases of rank 0, measured (eight ranks on one device) against the model (eight devices):", json.dumps(ph))
    m = ph["measured_s"]
    assert m["phase1_s"] > 0 and m["phase2_s"] > 0 and ph["model_s"]["phase1_s"] > 0 and ph["model_s"]["calls_above_the_cut"] > 0
    assert 1e-3 * b["ms_per_step"] >= m["phase1_s"] + m["phase2_s"]


@pytest.mark.gpu
@pytest.mark.parametrize("world,genomes,length", [(2, 12, 150_000), (5, 12, 150_000), (8, 40, 400_000)], ids=["2_ranks", "5_ranks_small_tree", "8_ranks_40_genomes"])
def test_compiled_host_several_ranks_equal_the_single_rank_build(gpu_lib, tmp_path, world, genomes, length):
    """pangraph_amd/host/build_driver.cpp as several ranks WITHOUT Python or torch in the ranks (PGA_RANK / PGA_WORLD / PGA_XDIR; here all on one device, a
    share of its memory each): subtrees per rank under each rank's own ready-set schedule, one exchange, the calls above the cut by all ranks with the queries
    of every group split (pga_batch_align_shard), a second exchange, pga_merge_match_lists on rank 0.  Every field of every record and every CIGAR of every
    call equal the single-rank driver's (whose records are held against the Python host and, through it, against the reference elsewhere).  Five ranks on a
    twelve-genome tree: ranks that own little or nothing, most calls above the cut; eight ranks on forty genomes of 0.4 Mbp: every rank owns subtrees."""
    import subprocess
    from conftest import ROOT
    from pangraph_amd import schedule as sched
    from pangraph_amd.levels import Population
    exe = os.path.join(ROOT, "pangraph_amd", "host", "build_driver")
    pop = Population(5, genomes, length)
    tasks = sched.build_tasks(pop)
    tf, o1, ow, xd = str(tmp_path / "tasks.bin"), str(tmp_path / "one.bin"), str(tmp_path / "many.bin"), tmp_path / "x"
    xd.mkdir()
    sched.write_task_file(tasks, tf, sensitivity=10, n_threads=4, pop=pop)
    share = "%.3f" % (0.6 / (world + 1))
    r = subprocess.run([exe, tf, o1, "6"], env=dict(os.environ, PGA_MEM_SHARE=share), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    procs = [subprocess.Popen([exe, tf, ow if k == 0 else str(tmp_path / f"unused{k}.bin"), "3"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(os.environ, PGA_MEM_SHARE=share, PGA_RANK=str(k), PGA_WORLD=str(world), PGA_XDIR=str(xd), PGA_DEVICE="0")) for k in range(world)]
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[0][-500:] + o[1][-1500:] for o in outs]
    one, _ = sched.read_driver_results(o1)
    many, _ = sched.read_driver_results(ow)
    assert len(one) == len(many) == len(tasks)
    fields = [f for f in one[0][0].dtype.names if f not in ("cigar_off", "pad")]
    n_rec = 0
    for tid, ((m1, c1), (m2, c2)) in enumerate(zip(one, many)):
        assert len(m1) == len(m2), (tid, len(m1), len(m2))
        for a, b in zip(m1, m2):
            assert all(a[f] == b[f] for f in fields), (tid, a, b)
            assert (c1[int(a["cigar_off"]):int(a["cigar_off"]) + int(a["n_cigar"])] == c2[int(b["cigar_off"]):int(b["cigar_off"]) + int(b["n_cigar"])]).all(), tid
        n_rec += len(m1)
    owner, _ = sched.partition_subtrees(pop, tasks, world)
    assert n_rec > len(tasks) and any(o < 0 for o in owner) and f"{world} ranks: {len(tasks)} calls ({sum(1 for o in owner if o < 0)} above the cut)" in outs[0][0]
